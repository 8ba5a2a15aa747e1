#!/bin/bash
# final validation batch: the whole GPU suite, the big-mesh table, the round profile (kernel stats, PMC, bench) -- on the final sources
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/x16
python -m pytest tests -m gpu -x -q > gpurun_out/x16/pytest.log 2>&1; tail -4 gpurun_out/x16/pytest.log
python3 tools/big_mesh_bench.py > gpurun_out/x16/big.txt 2>&1; grep -c triangles gpurun_out/x16/big.txt
bash tools/round_profile.sh round5 > gpurun_out/x16/profile.log 2>&1; tail -3 gpurun_out/x16/profile.log | cut -c1-600
