#!/bin/bash
# build container: copy what tools/validate_gpu.sh + tools/final_profiles.sh <tag> left under gpurun_out/ into profiles/<round>/ and rebuild its summary.json.  usage: tools/collect_profiles.sh round5 final5
set -e
round=${1:-round5}; tag=${2:-final5}
cd "$(dirname "$0")/.."
for f in bench_kernel_stats.csv single_stream_kernel_stats.csv pmc_wf_travq.json pmc_wf_advance.json bench_n1.json bench_under_rocprof.json single_stream_bench.json; do cp gpurun_out/$round/$f profiles/$round/$f; done
cp gpurun_out/x16/big.txt profiles/$round/big_mesh_bench.txt
cp gpurun_out/$tag/grid_bench.md gpurun_out/$tag/grid_bench.json gpurun_out/$tag/share_scaling.txt profiles/$round/
cp gpurun_out/${tag}_mp/pmc_memory_pipe.json profiles/$round/pmc_memory_pipe_wf_travq.json
(head -2 profiles/$round/kernel_timeline.txt; cat gpurun_out/${tag}_tl/timeline.txt) > /tmp/_tl.txt && cp /tmp/_tl.txt profiles/$round/kernel_timeline.txt
python3 tools/make_profile_summary.py profiles/$round > /dev/null
python3 -c "
import json, bench; d=json.load(open('profiles/$round/summary.json')); print('hash', d['code_hash'], bench.code_hash()); print(d['instruction_estimate_check'])
b=json.loads(open('profiles/$round/bench_n1.json').read().strip().splitlines()[-1]); print(b['value'], b['ms_per_step'], b['config'].get('ms_per_step_one_frame_in_flight'), b['config']['large']['ms_per_step'], b['config']['end_to_end']['program_s'], b['roofline']['frac'], b['cpu_baseline']['value'])"
