#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/x5
RT_TRAVQ_TOPLDS=128 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py -m gpu -x -q > gpurun_out/x5/pytest.log 2>&1; tail -5 gpurun_out/x5/pytest.log
REPS=2 STEPS=40 tools/ab_variants.sh x5_tl "--large-steps 0" "RT_TRAVQ_TOPLDS=32 --large-steps 0" "RT_TRAVQ_TOPLDS=64 --large-steps 0" "RT_TRAVQ_TOPLDS=96 --large-steps 0" "RT_TRAVQ_TOPLDS=128 --large-steps 0" "RT_TRAVQ_TOPLDS=160 --large-steps 0" "RT_TRAVQ_TOPLDS=256 --large-steps 0" "RT_PARTS=1 --large-steps 0" "RT_TRAVQ_TOPLDS=128 RT_PARTS=1 --large-steps 0" > gpurun_out/x5/tl.txt 2>&1
cat gpurun_out/x5/tl.txt
