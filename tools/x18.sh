#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/x18
E=raytracinggpu_amd/exp
REPS=2 STEPS=40 tools/ab_variants.sh x18_r "RT_LIB=$E/r96.so --large-steps 0" "RT_LIB=$E/r96.so RT_TRAVQ_R=96 RT_TRAVQ_LOW=80 --large-steps 0" "RT_LIB=$E/r96.so RT_TRAVQ_R=96 RT_TRAVQ_LOW=96 --large-steps 0" "RT_LIB=$E/r96.so RT_TRAVQ_R=96 RT_TRAVQ_LOW=128 --large-steps 0" "RT_LIB=$E/r96.so RT_TRAVQ_R=96 RT_TRAVQ_LOW=96 RT_TRAVQ_MINFREE=8 --large-steps 0" "RT_LIB=$E/r96.so RT_TRAVQ_R=96 RT_TRAVQ_LOW=128 RT_TRAVQ_MINFREE=8 --large-steps 0" "RT_LIB=$E/r96.so RT_TRAVQ_R=96 RT_TRAVQ_LOW=96 RT_TRAVQ_MINFREE=48 --large-steps 0" > gpurun_out/x18/r.txt 2>&1
cat gpurun_out/x18/r.txt
python3 - <<'PY'
import json
cur=None; seen=set()
for l in open('gpurun_out/ab_x18_r.log'):
    if l.startswith('=='): cur=l.strip(); continue
    if l.startswith('{') and cur not in seen:
        seen.add(cur)
        d=json.loads(l); s=d['roofline']['steps_per_frame']
        print(cur[34:90], d['ms_per_step'], {k:s[k] for k in ('iterations','refill_passes','refill_rounds','box_steps','serial_drains')})
PY
