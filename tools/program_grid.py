"""GPU box: the reference's own harness semantics (benchmark.py:10-33: wall clock around the whole PROCESS, num_rays in 1..256 x num_bounce in 1..10, 512x512) for `rt_launcher`, and -- on a
subset of the grid, because it takes seconds to minutes per point -- for the reference program itself (oracle/_ref/cpu, built from /root/reference/cpu_launcher.cpp in the build container,
OpenMP on the box's host cores).  One run per point for the launcher's full grid (REPS for more), wall clock by time.time() around subprocess.run as the reference's script does.
usage: python tools/program_grid.py [> profiles/roundN/program_grid.md]"""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracinggpu_amd as rt
from bench import host_cores

exe = os.path.join(ROOT, "raytracinggpu_amd", "rt_launcher")
ref = os.path.join(ROOT, "oracle", "_ref", "cpu")
rays = [2 ** i for i in range(9)]
bounces = list(range(1, 11))
reps = int(os.environ.get("REPS", "1"))
ref_points = [(s, b) for s in (1, 8, 64, 256) for b in (1, 3, 10)]
g = np.load(rt.scenes.CAT_FIXTURE, allow_pickle=False)
threads = host_cores()


def wall(cmd, cwd, env=None):
    t0 = time.time()
    r = subprocess.run(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1800)
    dt = time.time() - t0
    if r.returncode != 0:
        raise SystemExit(f"{cmd}: exit {r.returncode}: {r.stderr[-300:]}")
    return dt


with tempfile.TemporaryDirectory() as d:
    od = os.path.join(d, "cadnav.com_model", "Models_F0202A090")
    os.makedirs(od)
    with open(os.path.join(od, "cat.obj"), "w") as f:
        for v in g["vertices"]:
            f.write("v %.9g %.9g %.9g 1 1 1\r\n" % tuple(float(x) for x in v))
        for t in g["tri_obj_order"]:
            f.write("f %d/1/1 %d/1/1 %d/1/1\r\n" % tuple(int(x) + 1 for x in t))
    wall([exe, "1", "1"], d)                                          # first process after a pause
    res = np.zeros((len(rays), len(bounces)))
    for i, s in enumerate(rays):
        for j, b in enumerate(bounces):
            res[i, j] = min(wall([exe, str(s), str(b)], d) for _ in range(reps))
        print("# rt_launcher num_rays %d done" % s, file=sys.stderr, flush=True)
    print("`rt_launcher num_rays num_bounce` at 512x512 on one MI355X: wall-clock seconds around the whole process (benchmark.py:19-23), rows = num_rays, columns = num_bounce 1..10\n")
    print("| num_rays | " + " | ".join(str(b) for b in bounces) + " |")
    print("|---|" + "---|" * len(bounces))
    for i, s in enumerate(rays):
        print(f"| {s} | " + " | ".join("%.3f" % res[i, j] for j in range(len(bounces))) + " |")
    if os.path.exists(ref):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads))
        print(f"\nThe reference program (`oracle/_ref/cpu` = g++ -O3 -fopenmp of /root/reference/cpu_launcher.cpp, {threads} threads on the same box), the same way, on a subset of the grid; last column = reference / rt_launcher\n")
        print("| num_rays | num_bounce | reference s | rt_launcher s | ratio |")
        print("|---|---|---|---|---|")
        for s, b in ref_points:
            tr = wall([ref, str(s), str(b)], d, env)
            tg = res[rays.index(s), bounces.index(b)]
            print(f"| {s} | {b} | {tr:.3f} | {tg:.3f} | {tr / tg:.1f} |", flush=True)
    else:
        print("\n(oracle/_ref/cpu is not in this snapshot: reference column skipped)")
