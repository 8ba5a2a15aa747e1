#!/bin/bash
# GPU box, final validation batch of a round: the whole GPU suite, the compute side of the N-rank job, the launcher's timing laps, then the round's profile (kernel stats, PMC, bench)
# on the final sources.  usage: bash tools/validate_gpu.sh <round tag, e.g. round6>   (copy gpurun_out/<tag>/ and gpurun_out/<tag>_v/ into profiles/<tag>/ afterwards)
cd "$GRAFT_REPO_ROOT"
tag=${1:-round6}
v=gpurun_out/${tag}_v
mkdir -p $v
timeout -k 10 900 python -m pytest tests -m gpu -q > $v/pytest.log 2>&1; tail -4 $v/pytest.log | cut -c1-300
timeout -k 10 300 python tools/share_batch.py > $v/share_batch.txt 2> $v/share_batch.err; cut -c1-330 $v/share_batch.txt
timeout -k 10 200 python tools/launcher_timing.py > $v/launcher_timing.txt 2>&1; grep -E "run |scene:|rt_scene_upload" $v/launcher_timing.txt | cut -c1-200 | head -12
bash tools/round_profile.sh $tag > $v/profile.log 2>&1; tail -3 $v/profile.log | cut -c1-700
