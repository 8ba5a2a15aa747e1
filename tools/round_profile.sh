#!/bin/bash
# GPU box: the round's profile evidence in one go (copy gpurun_out/<tag>/ to profiles/<round>/ afterwards).
#   1. rocprofv3 --kernel-trace --stats of the default bench command            -> bench_kernel_stats.csv, bench.json
#   2. the same with RT_PARTS=1 (one sub-frame: every launch owns the chip)     -> single_stream_kernel_stats.csv
#   3. PMC passes (kernel-trace only, one counter group per pass) with RT_PARTS=1 for wf_travq and wf_advance
#      -> pmc_wf_travq.json, pmc_wf_advance.json      usage: tools/round_profile.sh <tag>
set -e
tag=${1:-round}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t1 -o trace -- python3 bench.py --no-cpu-baseline --no-end-to-end --large-steps 0 --steps 30 --warmup 5 > $out/bench_under_rocprof.json 2> $out/t1.err || { tail -5 $out/t1.err; exit 1; }
cp $(find $out/t1 -name "*kernel_stats.csv" | head -1) $out/bench_kernel_stats.csv
RT_PARTS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t2 -o trace -- python3 bench.py --no-cpu-baseline --no-end-to-end --large-steps 0 --steps 30 --warmup 5 > $out/single_stream_bench.json 2> $out/t2.err || { tail -5 $out/t2.err; exit 1; }
cp $(find $out/t2 -name "*kernel_stats.csv" | head -1) $out/single_stream_kernel_stats.csv
i=0
for grp in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32" \
  "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32" \
  "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  RT_PARTS=1 timeout -k 5 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -o pmc -- python3 bench.py --no-cpu-baseline --no-end-to-end --large-steps 0 --prewarm-ms 0 --steps 4 --warmup 1 > $out/p$i.json 2> $out/p$i.err || { echo "pass $i ($grp) failed"; tail -5 $out/p$i.err; }
done
python3 tools/pmc_summary.py $out "wf_travq<false" > $out/pmc_wf_travq.json
python3 tools/pmc_summary.py $out "wf_advance<false, false>" > $out/pmc_wf_advance.json
python3 bench.py --steps 30 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err
rm -rf $out/t1 $out/t2 $out/p[0-9]*
head -6 $out/bench_kernel_stats.csv; head -6 $out/single_stream_kernel_stats.csv
python3 -c "
import json
for k in ('wf_travq','wf_advance'):
    d=json.load(open('$out/pmc_%s.json'%k)); print(k, json.dumps(d['derived']))
"
cat $out/bench_n1.json
