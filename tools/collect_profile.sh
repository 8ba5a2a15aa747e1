#!/bin/bash
# Build container, after `gpurun bash tools/validate_gpu.sh <tag>`: copy what the call left under gpurun_out/ into profiles/<tag>/ and rebuild summary.json.
# (share_batch.txt and launcher_timing.txt keep their hand-written header lines; their bodies are replaced.)   usage: bash tools/collect_profile.sh round6
tag=${1:-round6}
cd "$(dirname "$0")/.."
for f in bench_kernel_stats.csv bench_n1.json bench_under_rocprof.json pmc_wf_advance.json pmc_wf_travq.json single_stream_bench.json single_stream_kernel_stats.csv; do cp gpurun_out/$tag/$f profiles/$tag/$f; done
[ -f gpurun_out/${tag}_v/ab_anyhit_steps.txt ] && cp gpurun_out/${tag}_v/ab_anyhit_steps.txt profiles/$tag/
(grep '^#' profiles/$tag/share_batch.txt; grep -v '^#' gpurun_out/${tag}_v/share_batch.txt) > /tmp/_sb.txt && cp /tmp/_sb.txt profiles/$tag/share_batch.txt
python tools/make_profile_summary.py profiles/$tag > /dev/null
python - <<PY
import bench, json
s = json.load(open('profiles/$tag/summary.json')); b = json.load(open('profiles/$tag/bench_n1.json'))
k = s['kernels']['wf_travq']
print('code hash', bench.code_hash(), '== summary', s['code_hash'])
print('wf_travq', k['rocprof_avg_us_two_streams'], 'us beside its twin,', k['rocprof_avg_us_single_stream'], 'us alone,', k['valu_wave_insts_per_launch'], 'vector wave-instructions = frac', round(k['valu_wave_insts_per_launch'] / k['rocprof_avg_us_single_stream'] / 1e3 / 1228.8, 4))
print('bench', b['ms_per_step'], 'ms', b['value'], 'Mrays/s; one frame in flight', b['config']['ms_per_step_one_frame_in_flight'], '; 7680x4320', b['config']['large']['ms_per_step'], '; frac', b['roofline']['frac'])
PY
grep -v '^#' profiles/$tag/share_batch.txt | cut -c1-150 | sed -n 1,4p
