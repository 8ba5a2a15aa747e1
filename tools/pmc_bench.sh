#!/bin/bash
# GPU box: hardware counters of the render kernel, one rocprofv3 --pmc pass per counter group
# (kernel-trace only; never combined with other trace domains).  usage: tools/pmc_bench.sh <tag> [bench args]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
rm -rf $out && mkdir -p $out
i=0
for grp in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32" \
  "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32" \
  "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -o pmc -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 "$@" > $out/p$i.json 2> $out/p$i.err || { echo "pass $i ($grp) failed"; tail -5 $out/p$i.err; }
done
python3 tools/pmc_summary.py $out "${PMC_KERNEL:-render_}" > $out/summary.json
cat $out/summary.json
