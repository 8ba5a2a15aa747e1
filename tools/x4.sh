#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/x4
E=raytracinggpu_amd/exp
REPS=2 STEPS=40 tools/ab_variants.sh x4_pad "RT_LIB=$E/p_none.so --large-steps 0" "RT_LIB=$E/p_1x4.so --large-steps 0" "RT_LIB=$E/p_2x4.so --large-steps 0" "RT_LIB=$E/p_4x4.so --large-steps 0" "RT_LIB=$E/p_4x2.so --large-steps 0" "RT_LIB=$E/p_4x1.so --large-steps 0" "RT_LIB=$E/p_2x1.so --large-steps 0" > gpurun_out/x4/pad.txt 2>&1
cat gpurun_out/x4/pad.txt
