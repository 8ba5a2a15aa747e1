cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5a
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5a/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r5a/pytest.log
for q in 0 1; do echo "== RT_TRAVQ_QW=$q"; RT_TRAVQ_QW=$q BIG_N=161,513 timeout -k 10 300 python3 tools/big_mesh_bench.py 2>&1 | tail -6; done > gpurun_out/r5a/big_mesh.txt
cat gpurun_out/r5a/big_mesh.txt
for q in 0 1; do RT_TRAVQ_QW=$q timeout -k 10 200 python3 tools/share_frames.py 2>&1 | tail -5; done > gpurun_out/r5a/share_frames.txt
RT_TRAVQ_QW=1 GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python3 tools/share_frames.py 2>&1 | tail -5 >> gpurun_out/r5a/share_frames.txt
cat gpurun_out/r5a/share_frames.txt
