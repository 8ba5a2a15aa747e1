"""GPU box: randomized soak of rt_ctx_set_pipelining.  Frames of several sizes / parameter sets rendered back to back on one or two
streams into rotating buffers, a tone-mapping consumer behind every frame, pipelining switched on and off on the way, async frames
mixed in; every 8-bit image compared with the image of a frame rendered alone.  usage: python3 tools/pipeline_stress.py [frames] [seed]"""
import os, sys, random
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = rt.Context(0)
v, t = rt.scenes.load_cat_arrays()
ctx.scene_upload(rt.scenes.spheres("cpu"), hostlib.build_mesh(v, t, object_slot=6))
cfgs = [(640, 360, 1, 3), (640, 360, 2, 1), (1920, 1080, 1, 3), (320, 200, 1, 2), (2560, 1440, 1, 1)]   # the last one is rendered in chunks
ps = [rt.make_params(W, H, spp, b, **rt.scenes.CPU_LAUNCHER) for (W, H, spp, b) in cfgs]
refs = [ctx.render_rgb8(p) for p in ps]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
bufs = {i: [torch.zeros((cfgs[i][1], cfgs[i][0], 4), dtype=torch.float32, device="cuda:0") for _ in range(3)] for i in range(len(cfgs))}
pend = []      # (config, image tensor)
bad = 0
cur_stream, cur_cfg, n_buf, k_buf = 0, 0, 2, 0
ctx.set_pipelining(True)
for k in range(n_frames):
    r = rng.random()
    if r < 0.08: cur_cfg = rng.randrange(len(cfgs))
    elif r < 0.12: n_buf = rng.choice([1, 2, 3])
    elif r < 0.15:
        torch.cuda.synchronize(); cur_stream ^= 1          # (a caller orders its own streams)
    elif r < 0.18:
        ctx.set_pipelining(rng.random() < 0.7)
    elif r < 0.21:                                          # an asynchronous host frame in between (its own stream and slots)
        torch.cuda.synchronize()
        W, H = cfgs[cur_cfg][:2]
        pin = rt.PinnedArray((H, W, 3), dtype=np.uint8)
        ctx.render_async(ps[cur_cfg], pin.array, slot=k & 1, rgb8=True); ctx.wait(k & 1)
        if not (pin.array == refs[cur_cfg]).all(): bad += 1; print("async frame", k, "differs")
        pin.close()
    W, H = cfgs[cur_cfg][:2]
    rows, _ = rt.interleaved_rows(H, 8, 0, 1)
    st = streams[cur_stream]
    buf = bufs[cur_cfg][k_buf % n_buf]; k_buf += 1
    img = torch.empty((H * W * 3 + 16,), dtype=torch.uint8, device="cuda:0")
    with torch.cuda.stream(st):
        ctx.render_device(ps[cur_cfg], rows, buf.data_ptr(), st.cuda_stream)
        ctx.tonemap_device(buf.data_ptr(), H * W, img.data_ptr(), st.cuda_stream)
    pend.append((cur_cfg, img, k))
    if len(pend) >= 24 or k == n_frames - 1:
        torch.cuda.synchronize()
        for c, im, kk in pend:
            W2, H2 = cfgs[c][:2]
            if not (im[:H2 * W2 * 3].cpu().numpy().reshape(H2, W2, 3) == refs[c]).all():
                bad += 1; print("frame", kk, "config", c, "differs")
        pend = []
ctx.set_pipelining(False)
ctx.selfcheck()
print("frames %d, differing %d" % (n_frames, bad))
sys.exit(1 if bad else 0)
