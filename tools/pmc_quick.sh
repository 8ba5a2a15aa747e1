#!/bin/bash
# GPU box: the three SQ counter groups only (instruction mix, busy / wait cycles).  usage: PMC_KERNEL=wf_path tools/pmc_quick.sh <tag> [bench args]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcq_$tag
rm -rf $out && mkdir -p $out
i=0
for grp in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32" \
  "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32" \
  "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -o pmc -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 "$@" > $out/p$i.json 2> $out/p$i.err || { echo "pass $i ($grp) failed"; tail -5 $out/p$i.err; }
done
python3 tools/pmc_summary.py $out "${PMC_KERNEL:-wf_path}" > $out/summary.json
python3 - <<PY
import json
d = json.load(open("$out/summary.json"))
c = d["counters_avg_per_dispatch"]
for k in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_VALU_TRANS_F32", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE", "_VGPR_Count", "_SGPR_Count", "_Scratch_Size"):
    print("%-28s %s" % (k, c.get(k)))
print(json.dumps(d["derived"], indent=1))
PY
