#!/bin/bash
# GPU box, round 4 run 1: baseline + sensitivity pads + cache-policy variants + LDS-node variants + unpack ubench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/x1
E=raytracinggpu_amd/exp
tools/ubench/unpack_rate > gpurun_out/x1/unpack_rate.jsonl 2>&1
REPS=2 STEPS=40 tools/ab_variants.sh x1_nt "RT_LIB=$E/base.so --large-steps 0" "RT_LIB=$E/nt1.so --large-steps 0" "RT_LIB=$E/nt2.so --large-steps 0" "RT_LIB=$E/nt4.so --large-steps 0" "RT_LIB=$E/nt8.so --large-steps 0" "RT_LIB=$E/nt15.so --large-steps 0" > gpurun_out/x1/nt.txt 2>&1
echo nt done
REPS=2 STEPS=40 tools/ab_variants.sh x1_pad "RT_LIB=$E/dbg_none.so --large-steps 0" "RT_LIB=$E/pad_valu.so --large-steps 0" "RT_LIB=$E/pad_salu.so --large-steps 0" "RT_LIB=$E/pad_lds.so --large-steps 0" "RT_LIB=$E/pad_vmem.so --large-steps 0" "RT_LIB=$E/pad_vmem2.so --large-steps 0" > gpurun_out/x1/pad.txt 2>&1
echo pad done
REPS=2 STEPS=40 tools/ab_variants.sh x1_lds "RT_LIB=$E/base.so --large-steps 0" "RT_LIB=$E/base.so RT_TRAVQ_LDS=12 --large-steps 0" "RT_LIB=$E/scap256.so RT_TRAVQ_LDS=13 --large-steps 0" "RT_LIB=$E/scap256.so RT_TRAVQ_LDS=12 --large-steps 0" "RT_LIB=$E/base.so RT_PARTS=1 --large-steps 0" "RT_LIB=$E/base.so RT_TRAVQ_LDS=12 RT_PARTS=1 --large-steps 0"  "RT_LIB=$E/scap256.so RT_TRAVQ_LDS=13 RT_PARTS=1 --large-steps 0" > gpurun_out/x1/lds.txt 2>&1
echo lds done
cat gpurun_out/x1/nt.txt gpurun_out/x1/pad.txt gpurun_out/x1/lds.txt
