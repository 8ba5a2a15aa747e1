#!/bin/bash
# GPU box: bench lines of several variants / env settings, one after the other.  usage: tools/ab_variants.sh <tag> "<ENV=.. ENV=..> --variant x" ...
tag=$1; shift
mkdir -p gpurun_out
i=0
for spec in "$@"; do
  i=$((i+1))
  envs=""; args=""
  for w in $spec; do case $w in *=*) envs="$envs $w";; *) args="$args $w";; esac; done
  echo "== $spec" >> gpurun_out/ab_$tag.log
  env $envs timeout -k 10 240 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 $args >> gpurun_out/ab_$tag.log 2>> gpurun_out/ab_$tag.err || echo "FAILED: $spec" >> gpurun_out/ab_$tag.log
done
python3 - <<PY
import json
for l in open("gpurun_out/ab_$tag.log"):
    if l.startswith("=="): print(l.strip())
    elif l.startswith("{"):
        d = json.loads(l); r = d.get("roofline", {})
        print("   ms/frame %.4f  Mrays/s %.1f  trav kernel_ms %.4f x %s" % (d["ms_per_step"], d["value"], r.get("kernel_ms", 0), r.get("launches_per_frame")))
    else: print(l.strip())
PY
