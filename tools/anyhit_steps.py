"""GPU box: the work-stack kernel's own step counters for the headline frame with any-hit on and off (RT_TRAVQ_ANYHIT; RT_TRAVQ_QW_COUNT=1 makes rt_count_work run the
production kernel's counting instantiation), then bench.py both ways, three times each.  usage: python tools/anyhit_steps.py > profiles/roundN/ab_anyhit_steps.txt"""
import os, sys, json
sys.path.insert(0, os.getcwd())
os.environ["RT_TRAVQ_QW_COUNT"] = "1"
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib
v, t = rt.scenes.load_cat_arrays()
mesh = hostlib.build_mesh(v, t, object_slot=6)
for a in ("1", "0"):
    os.environ["RT_TRAVQ_ANYHIT"] = a
    c = rt.Context(0)
    c.scene_upload(rt.scenes.spheres("cpu"), mesh)
    p = rt.make_params(1920, 1080, 1, 3, **rt.scenes.CPU_LAUNCHER)
    w = c.count_work(p, detail=True)
    print("ANYHIT", a, json.dumps(w))

import subprocess
for r in range(3):
    for a in ("1", "0"):
        env = dict(os.environ, RT_TRAVQ_ANYHIT=a)
        env.pop("RT_TRAVQ_QW_COUNT", None)
        out = subprocess.run([sys.executable, "bench.py", "--steps", "60", "--warmup", "5", "--large-steps", "0"], env=env, capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        print("bench RT_TRAVQ_ANYHIT=%s: %.4f ms per frame, %.0f Mrays/s" % (a, d["ms_per_step"], d["value"]), flush=True)
