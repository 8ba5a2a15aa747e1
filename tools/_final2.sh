cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/x17
python -m pytest tests -m gpu -x -q > gpurun_out/x17/pytest.log 2>&1; tail -3 gpurun_out/x17/pytest.log
bash tools/round_profile.sh round5 > gpurun_out/x17/profile.log 2>&1; tail -2 gpurun_out/x17/profile.log | cut -c1-300
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
