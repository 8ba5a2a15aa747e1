#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/x10
RT_TRAVQ_R=128 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py tests/test_gpu_lbvh.py -m gpu -x -q > gpurun_out/x10/pytest_r128.log 2>&1; tail -4 gpurun_out/x10/pytest_r128.log
REPS=2 STEPS=40 tools/ab_variants.sh x10_r "--large-steps 0" "RT_TRAVQ_R=128 --large-steps 0" "RT_TRAVQ_R=128 RT_TRAVQ_LOW=32 --large-steps 0" "RT_TRAVQ_R=128 RT_TRAVQ_LOW=64 --large-steps 0" "RT_TRAVQ_R=128 RT_TRAVQ_MINFREE=16 --large-steps 0" "RT_TRAVQ_R=128 RT_TRAVQ_MINFREE=64 --large-steps 0" "RT_PARTS=1 --large-steps 0" "RT_TRAVQ_R=128 RT_PARTS=1 --large-steps 0" > gpurun_out/x10/r.txt 2>&1
cat gpurun_out/x10/r.txt
