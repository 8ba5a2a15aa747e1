#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
python tools/small_frame_parts.py > gpurun_out/r6/small_frame_parts.txt 2>&1; cat gpurun_out/r6/small_frame_parts.txt | cut -c1-250
python tools/launcher_timing.py > gpurun_out/r6/launcher_timing_b.txt 2>&1; grep -E "timing:|run " gpurun_out/r6/launcher_timing_b.txt | cut -c1-200
RT_PARTS_MIN_ITEMS=400000 python tools/launcher_timing.py > gpurun_out/r6/launcher_timing_c.txt 2>&1; grep -E "enqueue|run |render_rgb8, total" gpurun_out/r6/launcher_timing_c.txt | cut -c1-200
python bench.py --no-cpu-baseline --no-end-to-end > gpurun_out/r6/bench_d.json 2> gpurun_out/r6/bench_d.err
python -c "
import json;d=json.loads(open('gpurun_out/r6/bench_d.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'],d['config'].get('ms_per_step_one_frame_in_flight'))"
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r6/pytest_d.log 2>&1; tail -5 gpurun_out/r6/pytest_d.log | cut -c1-300
