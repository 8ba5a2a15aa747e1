#!/bin/bash
# Build container: variants of the library with 32 extra scalar / vector / 4 extra LDS instructions per BOX step of wf_travq
# (sensitivity experiment: which issue port bounds the kernel).  The GPU box then runs: REPS=2 tools/ab_variants.sh pad \
#   "--variant auto" "RT_LIB=gpurun_out/pad/salu.so --variant auto" "RT_LIB=gpurun_out/pad/valu.so ..." "RT_LIB=gpurun_out/pad/lds.so ..."
set -e
cd "$(dirname "$0")/.."
mkdir -p raytracinggpu_amd/pad
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -fPIC -shared"
for v in SALU VALU LDS VMEM; do
  /opt/rocm/bin/hipcc $F -DRT_DEBUG -DRT_PAD_$v -o raytracinggpu_amd/pad/$(echo $v | tr A-Z a-z).so raytracinggpu_amd/csrc/rt_capi.hip &
done
wait
ls -la raytracinggpu_amd/pad
