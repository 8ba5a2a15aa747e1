#!/bin/bash
# GPU box: the 4-wide fixed-point BOX step (RT_TRAVQ_QW=1) against the default: parity tests under the knob, then ms per frame interleaved, then step counters of both kernels
cd "$GRAFT_REPO_ROOT"; tag=${1:-qw}; out=gpurun_out/$tag; mkdir -p $out
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  RT_TRAVQ_QW=1 timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py -x -q -m gpu ${PYTEST_K:+-k "$PYTEST_K"} > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
  tail -3 $out/pytest.log
fi
for rep in 1 2; do for q in 0 1; do
  RT_TRAVQ_QW=$q RT_TRAVQ_QW_COUNT=$q timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-end-to-end --large-steps 0 --dump-frame $out/f$q.npy > $out/b$q.json 2> $out/b$q.err || { tail -3 $out/b$q.err; exit 1; }
  python3 - <<P
import json; b=json.load(open("$out/b$q.json")); r=b["roofline"]; s=r.get("steps_per_frame",{})
print("qw=$q", b["ms_per_step"], b["config"].get("ms_per_step_one_frame_in_flight"), [(k["kernel"], k["kernel_ms"]) for k in r["kernels"]], "box_steps", s.get("box_steps"), "tri_steps", s.get("tri_steps"), "serial", s.get("serial_drains"), "box occ", r.get("box_step_lane_occupancy"), "grid", b["config"].get("grid_blocks"))
P
done; done
python3 - <<P
import numpy as np
a=np.load("$out/f0.npy"); b=np.load("$out/f1.npy")
d=(a.view(np.uint32)!=b.view(np.uint32))
print("words differing:", int(d.sum()), "of", d.size, "pixels:", int(d.any(axis=-1).sum()), "max abs diff", float(np.abs(a-b).max()))
P
