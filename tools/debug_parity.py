"""Debug helper (GPU box): per-pixel ulp differences between the HIP path and the oracle."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracinggpu_amd as rt
from oracle import oracle_py as orc

scene = sys.argv[1] if len(sys.argv) > 1 else "cpu"
W, H, spp, b = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (512, 512, 1, 0)
g = np.load(rt.scenes.CAT_FIXTURE, allow_pickle=False)
ctx = rt.Context(0)
mesh = None
if scene == "cpu":
    mesh = dict(vertices=g["vertices"], indices=g["tri_bvh_order"], bvh_arr10=g["bvh_arr10"], albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
ctx.scene_upload(rt.scenes.spheres(scene), mesh)
got = ctx.render(rt.make_params(W, H, spp, b, **rt.scenes.CPU_LAUNCHER))
om = orc.Mesh.from_arrays(g["vertices"], g["tri_obj_order"]).build_bvh()
sc = orc.Scene.preset(scene, om if scene == "cpu" else None)
exp, _, _ = sc.render(W, H, spp, b, want_rgb8=False)
a = got[..., :3].view(np.uint32).astype(np.int64); e = exp[..., :3].view(np.uint32).astype(np.int64)
d = a - e
print("mismatch frac", (d != 0).mean(), "max ulp", np.abs(d).max(), "rays equal", (got[..., 3] == exp[..., 3]).all())
print("ulp histogram", np.unique(np.clip(np.abs(d), 0, 8), return_counts=True))
z = np.float32(-W / (2 * np.float32(np.tan(np.float64(np.float32(np.float32(np.pi / 3) / 2))))))
ids = np.zeros((H, W), int)
bad = np.argwhere((d != 0).any(axis=2))
for (i, j) in bad[:: max(1, len(bad) // 12)][:12]:
    u = np.array([np.float32(j) - np.float32(W) / 2 + 0.5, np.float32(H) / 2 - i - 0.5, z], np.float32)
    n = np.sqrt(np.float32(np.float32(u[0] * u[0] + u[1] * u[1]) + u[2] * u[2])); u = (u / n).astype(np.float32)
    hit, oid, P, N = sc.intersect_all([0, 0, 55], u)
    print(i, j, "obj", oid, "got", got[i, j, :3], "exp", exp[i, j, :3], "ulp", d[i, j])
# which objects do mismatching pixels hit
cnt = {}
for (i, j) in bad[:: max(1, len(bad) // 400)]:
    u = np.array([np.float32(j) - np.float32(W) / 2 + 0.5, np.float32(H) / 2 - i - 0.5, z], np.float32)
    n = np.sqrt(np.float32(np.float32(u[0] * u[0] + u[1] * u[1]) + u[2] * u[2])); u = (u / n).astype(np.float32)
    hit, oid, P, N = sc.intersect_all([0, 0, 55], u)
    cnt[oid] = cnt.get(oid, 0) + 1
print("objects hit by mismatching pixels (sampled):", cnt)
