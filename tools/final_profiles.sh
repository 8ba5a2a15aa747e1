#!/bin/bash
# GPU box: every table of profiles/<round>/ that depends on the kernels, on the final sources (after tools/validate_gpu.sh).  usage: tools/final_profiles.sh <dir under gpurun_out>
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-final}
mkdir -p $out
timeout -k 10 600 python3 tools/grid_bench.py > $out/grid_bench.md 2> $out/grid.err || { tail -3 $out/grid.err; exit 1; }
cp gpurun_out/grid_bench.json $out/ 2>/dev/null
timeout -k 10 300 python3 tools/share_scaling.py > $out/share_scaling.txt 2> $out/ss.err || { tail -3 $out/ss.err; exit 1; }
timeout -k 10 300 python3 tools/share_frames.py > $out/share_frames.txt 2> $out/sf.err || { tail -3 $out/sf.err; exit 1; }
timeout -k 10 300 python3 tools/spp_bench.py > $out/spp.txt 2> $out/spp.err || { tail -3 $out/spp.err; exit 1; }
bash tools/kernel_timeline.sh ${1:-final}_tl > $out/timeline.log 2>&1 || { tail -3 $out/timeline.log; exit 1; }
bash tools/pmc_memory_pipe.sh ${1:-final}_mp > $out/mempipe.log 2>&1 || { tail -3 $out/mempipe.log; exit 1; }
tail -4 $out/share_scaling.txt; tail -3 $out/spp.txt
