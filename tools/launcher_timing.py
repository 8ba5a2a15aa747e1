"""GPU box: where the wall time of `rt_launcher 8 3` (the reference's own CLI and size) goes: the launcher's --timing 1 breakdown, three runs, next to the
program's own `Rendering time`.  The cat is written as an OBJ into a temporary directory the way bench.py's end_to_end does.
usage: python tools/launcher_timing.py [extra launcher args ...] [> profiles/roundN/launcher_timing.txt]"""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracinggpu_amd as rt
exe = os.path.join(ROOT, "raytracinggpu_amd", "rt_launcher")
g = np.load(rt.scenes.CAT_FIXTURE, allow_pickle=False)
with tempfile.TemporaryDirectory() as d:
    od = os.path.join(d, "cadnav.com_model", "Models_F0202A090")
    os.makedirs(od)
    with open(os.path.join(od, "cat.obj"), "w") as f:
        for v in g["vertices"]:
            f.write("v %.9g %.9g %.9g 1 1 1\r\n" % tuple(float(x) for x in v))
        for t in g["tri_obj_order"]:
            f.write("f %d/1/1 %d/1/1 %d/1/1\r\n" % tuple(int(x) + 1 for x in t))
    for run in range(int(os.environ.get("RUNS", "3"))):
        t0 = time.perf_counter()
        r = subprocess.run([exe, "8", "3", "--timing", "1", *sys.argv[1:]], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
        wall = time.perf_counter() - t0
        print(f"--- run {run}: exit {r.returncode}, wall clock around the process {wall * 1e3:.1f} ms; {r.stdout.strip()}")
        print(r.stderr.rstrip())
