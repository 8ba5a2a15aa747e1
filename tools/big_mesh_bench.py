"""GPU box: the headline frame settings on meshes far larger than the cat (displaced grids of 51 200 and 524 288 triangles): host BVH
build time, upload, frame time, rays/s, traversal work per ray; device rebuild time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib, tiling
ctx = rt.Context(0)
for n in [int(x) for x in os.environ.get("BIG_N", "45,161,513,1025").split(",")]:
    rng = np.random.default_rng(11)
    gx, gz = np.meshgrid(np.linspace(-18, 18, n), np.linspace(-14, 22, n), indexing="ij")
    gy = -9.0 + 3.0 * np.sin(gx * 0.45) * np.cos(gz * 0.38) + 0.15 * rng.standard_normal((n, n))
    v = np.stack([gx, gy, gz], -1).reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    a = (i * n + j).reshape(-1)
    t = np.concatenate([np.stack([a, a + 1, a + n], 1), np.stack([a + 1, a + n + 1, a + n], 1)]).astype(np.int32)
    t0 = time.perf_counter(); mesh = hostlib.build_mesh(v, t, albedo=rt.scenes.CAT_ALBEDO, object_slot=6); t_build = time.perf_counter() - t0
    t0 = time.perf_counter(); ctx.scene_upload(rt.scenes.spheres("cpu"), mesh); t_up = time.perf_counter() - t0
    W, H = 1920, 1080
    p = rt.make_params(W, H, 1, 3, **rt.scenes.CPU_LAUNCHER)
    rows, _ = rt.interleaved_rows(H, 8, 0, 1)
    buf = tiling.local_buffer(H, W, 1, "cuda:0")
    for _ in range(20):
        ctx.render_device(p, rows, buf.data_ptr())
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.render_device(p, rows, buf.data_ptr())
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    rays = float(buf[..., 3].double().sum().item())
    w = ctx.count_work(p)
    t0 = time.perf_counter(); arr, order = ctx.mesh_rebuild(len(t)); t_re = time.perf_counter() - t0
    print("%7d triangles, %6d nodes: host build %.2f s, upload %.2f s, frame %.3f ms = %.0f Mrays/s, per ray %.1f box tests / %.1f triangle tests; device rebuild %.2f s" %
          (len(t), len(mesh["bvh_arr10"]), t_build, t_up, ms, rays / ms / 1e3, w["box_tests"] / w["rays"], w["tri_tests"] / w["rays"], t_re), flush=True)
    # the same mesh on the LBVH tree (rt_mesh_rebuild_mode(RT_BVH_LBVH): Morton sort + parallel hierarchy, leaves cut by the surface-area heuristic: at most 32 triangles)
    t0 = time.perf_counter(); arr, order = ctx.mesh_rebuild(len(t), mode="lbvh"); t_lb = time.perf_counter() - t0
    st = ctx.build_stats()
    for _ in range(20):
        ctx.render_device(p, rows, buf.data_ptr())
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.render_device(p, rows, buf.data_ptr())
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    rays2 = float(buf[..., 3].double().sum().item())
    assert rays2 == rays
    w = ctx.count_work(p)
    print("        LBVH: %6d nodes, leaves <= %d, depth %d: device build %.2f ms (+ install %.0f ms, call %.2f s), frame %.3f ms = %.0f Mrays/s, per ray %.1f box tests / %.1f triangle tests" %
          (st["n_nodes"], st["max_leaf_tris"], st["max_depth"], st["device_build_ms"], st["install_ms"], t_lb, ms, rays / ms / 1e3, w["box_tests"] / w["rays"], w["tri_tests"] / w["rays"]), flush=True)
