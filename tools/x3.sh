#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/x3
E=raytracinggpu_amd/exp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py tests/test_gpu_bvh_build.py -m gpu -x -q > gpurun_out/x3/pytest.log 2>&1; tail -5 gpurun_out/x3/pytest.log
REPS=3 STEPS=40 tools/ab_variants.sh x3_top "RT_LIB=$E/top0.so --large-steps 0" "RT_LIB=$E/top2.so --large-steps 0" "RT_LIB=$E/top0.so RT_PARTS=1 --large-steps 0" "RT_LIB=$E/top2.so RT_PARTS=1 --large-steps 0" > gpurun_out/x3/top.txt 2>&1
cat gpurun_out/x3/top.txt
