#!/bin/bash
# GPU box: the core counter groups of wf_travq for two settings of the environment (A/B), RT_PARTS=1.  usage: tools/pmc_ab.sh <tag> "<ENV=..>" "<ENV=..>"
set -e
tag=${1:-pmcab}; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
k=0
for spec in "$@"; do
  k=$((k+1)); out=gpurun_out/$tag/s$k; rm -rf $out; mkdir -p $out; echo "$spec" > $out/spec.txt
  i=0
  for grp in \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
    "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32" \
    "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAIT_INST_LDS" \
    "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum" \
    "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    env $spec RT_PARTS=1 timeout -k 5 90 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -o pmc -- python3 bench.py --no-cpu-baseline --no-end-to-end --large-steps 0 --prewarm-ms 0 --steps 4 --warmup 1 > $out/p$i.json 2> $out/p$i.err || { echo "pass $i ($grp) failed"; grep -m3 -i "error\|abort\|fail\|invalid" $out/p$i.err || true; }
  done
  python3 tools/pmc_summary.py $out "wf_travq<false" > $out/pmc_wf_travq.json
  rm -rf $out/p[0-9]*
  echo "== $spec"; python3 -c "
import json; d=json.load(open('$out/pmc_wf_travq.json')); c=d['counters_avg_per_dispatch']
for k in sorted(c): print('  %-34s %16.0f' % (k, c[k]))
print(json.dumps(d['derived'], indent=1))"
done
