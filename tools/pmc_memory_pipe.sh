#!/bin/bash
# GPU box: counters of the vector-memory pipeline (address unit TA, L1 = TCP, data return TD, address translation) for the traversal
# kernel, one counter group per pass, RT_PARTS=1.  usage: tools/pmc_memory_pipe.sh <tag> [kernel-substring]
set -e
tag=${1:-mempipe}
kern=${2:-wf_travq<false}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
rm -rf $out && mkdir -p $out
i=0
for grp in \
  "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
  "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
  "TD_TD_BUSY_sum TD_TC_STALL_sum" "TD_LOAD_WAVEFRONT_sum TD_SPI_STALL_sum" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
  "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  RT_PARTS=1 timeout -k 5 ${PASS_TIMEOUT:-60} rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -o pmc -- python3 bench.py --no-cpu-baseline --no-end-to-end --large-steps 0 --prewarm-ms 0 --steps 4 --warmup 1 > $out/p$i.json 2> $out/p$i.err || { echo "pass $i ($grp) failed"; grep -m3 -i "error\|abort\|fail\|invalid" $out/p$i.err || true; }
done
python3 tools/pmc_summary.py $out "$kern" > $out/pmc_memory_pipe.json
rm -rf $out/p[0-9]*
cat $out/pmc_memory_pipe.json
