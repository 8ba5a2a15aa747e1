"""GPU box: step counters of the counting instantiation of wf_travq for the headline frame (rt_count_work, detail)."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
ctx = rt.Context(0)
verts, tris = rt.scenes.load_cat_arrays()
ctx.scene_upload(rt.scenes.spheres("cpu"), hostlib.build_mesh(verts, tris, albedo=rt.scenes.CAT_ALBEDO, object_slot=rt.scenes.mesh_slot("cpu")))
w = ctx.count_work(rt.make_params(W, H, 1, 3, **rt.scenes.CPU_LAUNCHER), detail=True)
s = w["steps"]
w["box_lane_occupancy"] = (w["box_tests"] - w["rays"]) / (128.0 * max(s["box_steps"], 1))     # the root-box tests belong to the uniform kernel
w["tri_lane_occupancy"] = w["tri_tests"] / (128.0 * max(s["tri_steps"], 1))
print(json.dumps(w))
