cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5b
timeout -k 10 300 python3 tools/spp_slope.py > gpurun_out/r5b/spp_slope_after.txt 2>&1; grep -E "^A|^B|after" gpurun_out/r5b/spp_slope_after.txt | cut -c1-150
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sample or spp or jitter or progressive or grid or stochastic" 2>&1 | tail -3
