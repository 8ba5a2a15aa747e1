cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5a
for mg in 32 48 64 96; do echo "== RT_TRAV_MIN_GROUPS=$mg"; RT_TRAV_MIN_GROUPS=$mg KS=2,3,4,5,6,8 WORLDS=8,4,2 timeout -k 10 300 python3 tools/share_frames.py 2>&1 | tail -3; done > gpurun_out/r5a/share_mg2.txt
for mg in 16 64; do echo "== 3840x2160 RT_TRAV_MIN_GROUPS=$mg"; SIZE=3840x2160 RT_TRAV_MIN_GROUPS=$mg KS=2,4 WORLDS=8 timeout -k 10 300 python3 tools/share_frames.py 2>&1 | tail -2; done >> gpurun_out/r5a/share_mg2.txt
cat gpurun_out/r5a/share_mg2.txt
