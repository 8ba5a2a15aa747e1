#!/bin/bash
# GPU box: A/B several builds of libraytrace_hip.so on the default bench (swap the .so, run bench, restore)
cd "$GRAFT_REPO_ROOT/raytracinggpu_amd"
cp libraytrace_hip.so /tmp/lib_orig.so
for lib in "$@"; do
  cp $lib libraytrace_hip.so
  (cd .. && timeout -k 10 120 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', d['value'], d['ms_per_step'], 'trav', r['kernel_ms'], 'frac', r['frac'])")
done
cp /tmp/lib_orig.so libraytrace_hip.so
