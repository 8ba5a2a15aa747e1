#!/bin/bash
# GPU box: vector-memory-pipe counters of one kernel.  usage: tools/pmc_mem.sh <tag> [bench args]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcmem_$tag
rm -rf $out && mkdir -p $out
i=0
for grp in "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_TA_TCP_STATE_READ_sum" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY" \
           "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -o pmc -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 "$@" > $out/p$i.json 2> $out/p$i.err || { echo "pass $i ($grp) failed"; tail -3 $out/p$i.err; }
done
python3 tools/pmc_summary.py $out "${PMC_KERNEL:-wf_trav}" > $out/summary.json
python3 -c "
import json; d=json.load(open('$out/summary.json'))['counters_avg_per_dispatch']
for k in sorted(d): print(k, d[k])"
