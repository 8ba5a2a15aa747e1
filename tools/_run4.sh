cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5a
for mg in 16 32 64 128 256; do for os_ in 2 1; do echo "== RT_TRAV_MIN_GROUPS=$mg RT_TRAVQ_OVERSUB=$os_"; RT_TRAV_MIN_GROUPS=$mg RT_TRAVQ_OVERSUB=$os_ KS=2,4 WORLDS=8 timeout -k 10 200 python3 tools/share_frames.py 2>&1 | tail -1; done; done > gpurun_out/r5a/share_mg.txt
cat gpurun_out/r5a/share_mg.txt
