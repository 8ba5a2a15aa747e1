#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/x17
E=raytracinggpu_amd/exp
RT_LIB=$E/r96.so RT_TRAVQ_R=96 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "direct_lighting or bounces or work_counters or headline or synthetic" > gpurun_out/x17/pytest_r96.log 2>&1; tail -3 gpurun_out/x17/pytest_r96.log
REPS=3 STEPS=40 tools/ab_variants.sh x17_r "RT_LIB=$E/r96.so --large-steps 0" "RT_LIB=$E/r96.so RT_TRAVQ_R=96 --large-steps 0" "RT_LIB=$E/r96.so RT_TRAVQ_R=96 RT_TRAVQ_LOW=64 --large-steps 0" "RT_LIB=$E/r96.so RT_TRAVQ_R=96 RT_TRAVQ_MINFREE=16 --large-steps 0" "RT_LIB=$E/r96.so RT_TRAVQ_R=96 RT_TRAVQ_MINFREE=32 RT_TRAVQ_LOW=64 --large-steps 0" > gpurun_out/x17/r.txt 2>&1
cat gpurun_out/x17/r.txt
