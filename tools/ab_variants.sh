#!/bin/bash
# GPU box: bench lines of several variants / env settings, REPS rounds interleaved.  usage: tools/ab_variants.sh <tag> "<ENV=.. ENV=..> --variant x" ...
tag=$1; shift
mkdir -p gpurun_out
rm -f gpurun_out/ab_$tag.log
for rep in $(seq 1 ${REPS:-2}); do
for spec in "$@"; do
  envs=""; args=""
  for w in $spec; do case $w in *=*) envs="$envs $w";; *) args="$args $w";; esac; done
  echo "== $spec" >> gpurun_out/ab_$tag.log
  env $envs timeout -k 10 240 python3 bench.py --no-cpu-baseline --no-end-to-end --steps ${STEPS:-20} --warmup 3 $args >> gpurun_out/ab_$tag.log 2>> gpurun_out/ab_$tag.err || { echo "FAILED: $spec" >> gpurun_out/ab_$tag.log; tail -3 gpurun_out/ab_$tag.err; echo "stopping after the first failure"; exit 1; }
done
done
python3 - <<PY
import json, collections
res = collections.OrderedDict(); cur = None
for l in open("gpurun_out/ab_$tag.log"):
    if l.startswith("=="): cur = l.strip()[3:]; res.setdefault(cur, [])
    elif l.startswith("{"):
        d = json.loads(l); r = d.get("roofline", {})
        res[cur].append((d["ms_per_step"], r.get("kernel_ms", 0), r.get("launches_per_frame")))
    else: res[cur].append(("FAILED", 0, 0))
for k, v in res.items():
    print("%-70s ms/frame %s | trav ms %s x %s" % (k, " ".join("%.4f" % x[0] if x[0] != "FAILED" else "FAILED" for x in v), " ".join("%.4f" % x[1] for x in v), v[0][2]))
PY
