cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5b
ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>/dev/null | head -30
timeout -k 10 300 python3 tools/spp_slope.py > gpurun_out/r5b/spp_slope.txt 2>&1; cat gpurun_out/r5b/spp_slope.txt
