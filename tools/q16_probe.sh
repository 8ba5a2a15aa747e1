#!/bin/bash
# GPU box: the 16-bit fixed-point BOX step (RT_TRAVQ_Q16=1) against the default: ms/frame, single-context launch times, and the frames compared word for word
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/${1:-q16}; mkdir -p $out
for rep in 1 2; do for q in 0 1; do
  RT_TRAVQ_Q16=$q timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-end-to-end --large-steps 0 --dump-frame $out/f$q.npy > $out/b$q.json 2> $out/b$q.err || { tail -3 $out/b$q.err; exit 1; }
  python3 - <<P
import json; b=json.load(open("$out/b$q.json")); print("q16=$q", b["ms_per_step"], b["config"].get("ms_per_step_one_frame_in_flight"), [(k["kernel"], k["kernel_ms"]) for k in b["roofline"]["kernels"]])
P
done; done
python3 - <<P
import numpy as np
a=np.load("$out/f0.npy"); b=np.load("$out/f1.npy")
d=(a.view(np.uint32)!=b.view(np.uint32))
print("words differing:", int(d.sum()), "of", d.size, "pixels:", int(d.any(axis=-1).sum()), "max abs diff", float(np.abs(a-b).max()))
P
