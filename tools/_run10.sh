cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5b
echo "== before (all chains of sub-frame 0 issued first)"; RT_LIB=raytracinggpu_amd/exp/qw_s460.so RT_TRAVQ_QW=1 timeout -k 10 200 python3 tools/spp_bench.py 2>&1 | grep spp
echo "== after (chains interleaved over the sub-frames)"; timeout -k 10 200 python3 tools/spp_bench.py 2>&1 | grep spp
