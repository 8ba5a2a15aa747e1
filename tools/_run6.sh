cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5b
python3 tools/launcher_timing.py > gpurun_out/r5b/launcher_timing.txt 2>&1; cat gpurun_out/r5b/launcher_timing.txt
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r5b/bench_n1.json 2> gpurun_out/r5b/bench_n1.err || tail -5 gpurun_out/r5b/bench_n1.err
python3 -c "
import json; d=json.load(open('gpurun_out/r5b/bench_n1.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], d['config'].get('ms_per_step_one_frame_in_flight'), d['config'].get('end_to_end'))
print({k: r[k] for k in ('kernel','kernel_ms','frac','frac_unweighted','box_step_lane_occupancy','tri_step_lane_occupancy','steps_per_frame','valu_wave_insts_per_launch')})
print(d['cpu_baseline'])
"
timeout -k 10 600 python3 -m pytest tests/test_capi_loads.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
