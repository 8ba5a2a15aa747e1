#!/bin/bash
# round 6, first measurement batch: the whole GPU suite, the bench line, the launcher's timing laps, the compute side of the N-rank job with frame batches
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r6/pytest_c.log 2>&1; tail -30 gpurun_out/r6/pytest_c.log | cut -c1-300
python bench.py > gpurun_out/r6/bench_c.json 2> gpurun_out/r6/bench_c.err
python -c "
import json;d=json.loads(open('gpurun_out/r6/bench_c.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'])"
python tools/launcher_timing.py > gpurun_out/r6/launcher_timing.txt 2>&1; grep -E "scene:|rt_scene_upload|run " gpurun_out/r6/launcher_timing.txt | cut -c1-200
python tools/share_batch.py > gpurun_out/r6/share_batch.txt 2>&1; cat gpurun_out/r6/share_batch.txt | cut -c1-400
