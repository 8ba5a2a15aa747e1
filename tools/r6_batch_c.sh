#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r6/pytest_e.log 2>&1; tail -5 gpurun_out/r6/pytest_e.log | cut -c1-300
SIZES=1920x1080 python tools/share_batch.py > gpurun_out/r6/share_batch2.txt 2>&1; cat gpurun_out/r6/share_batch2.txt | cut -c1-420
python bench.py --no-cpu-baseline --no-end-to-end > gpurun_out/r6/bench_e.json 2> gpurun_out/r6/bench_e.err
python -c "
import json;d=json.loads(open('gpurun_out/r6/bench_e.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'],d['config'].get('ms_per_step_one_frame_in_flight'))"
python bench.py --gpus 2 --share-gpu --no-cpu-baseline --large-steps 0 --steps 24 > gpurun_out/r6/bench_share2.json 2> gpurun_out/r6/bench_share2.err; tail -c 1800 gpurun_out/r6/bench_share2.json; tail -3 gpurun_out/r6/bench_share2.err
