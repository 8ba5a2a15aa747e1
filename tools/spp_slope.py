"""GPU box: why a 256-sample 1080p frame costs more per sample than a 32-sample one.  Three measurements with the shader clock sampled beside them
(sysfs, every 5 ms, a thread of this process):
  A  N one-sample frames back to back (two frames in flight, as bench.py runs them), per-frame GPU time over the run: if the SAME frames get slower as the run gets longer,
     the cause is the device (clock / power management under sustained load), not the multi-sample path;
  B  one frame of 32 / 64 / 128 / 256 samples: ms per sample;
  C  B again after a 0.5 s pause vs straight after a 300 ms busy period (is the first part of a long frame the fast part?).
usage: python tools/spp_slope.py [> profiles/roundN/spp_slope.txt]"""
import glob, os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib, tiling


def clock_files():
    out = []
    for pat in ("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input", "/sys/class/hwmon/hwmon*/freq1_input"):
        out += glob.glob(pat)
    return sorted(set(out))


class ClockSampler:
    def __init__(self):
        self.files = clock_files()
        self.samples, self.on = [], False
        self.power = sorted(set(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")))

    def _run(self):
        while self.on:
            t = time.perf_counter()
            vals = []
            for f in self.files[:1]:
                try:
                    vals.append(int(open(f).read()) / 1e6)
                except (OSError, ValueError):
                    pass
            pw = None
            for f in self.power[:1]:
                try:
                    pw = int(open(f).read()) / 1e6
                except (OSError, ValueError):
                    pass
            self.samples.append((t, vals[0] if vals else None, pw))
            time.sleep(0.005)

    def __enter__(self):
        self.samples, self.on = [], True
        self.th = threading.Thread(target=self._run, daemon=True); self.th.start()
        return self

    def __exit__(self, *a):
        self.on = False; self.th.join()

    def summary(self, t0, t1, bins=6):
        s = [(t, c, p) for t, c, p in self.samples if t0 <= t <= t1 and c is not None]
        if not s:
            return "no clock readings (sysfs not readable: %d files)" % len(self.files)
        edges = np.linspace(t0, t1, bins + 1)
        out = []
        for a, b in zip(edges[:-1], edges[1:]):
            cs = [c for t, c, p in s if a <= t < b]
            ps = [p for t, c, p in s if a <= t < b and p is not None]
            out.append("%s MHz%s" % ("%.0f" % np.mean(cs) if cs else "-", (" %.0f W" % np.mean(ps)) if ps else ""))
        return "shader clock over the run, %d equal slices: " % bins + " | ".join(out)


ctx = rt.Context(0)
v, t = rt.scenes.load_cat_arrays()
ctx.scene_upload(rt.scenes.spheres("cpu"), hostlib.build_mesh(v, t, albedo=rt.scenes.CAT_ALBEDO, object_slot=6))
W, H = 1920, 1080
rows, _ = rt.interleaved_rows(H, 8, 0, 1)
st = torch.cuda.Stream()
bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(2)]
print("clock files:", clock_files()[:2])
# A: one-sample frames, sustained
p1 = rt.make_params(W, H, 1, 3, **rt.scenes.CPU_LAUNCHER)
ctx.set_pipelining(True)
for N in (40, 400):
    torch.cuda.synchronize(); time.sleep(0.5)
    with ClockSampler() as cs, torch.cuda.stream(st):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
        t0 = time.perf_counter()
        ev[0].record()
        for k in range(N):
            ctx.render_device(p1, rows, bufs[k & 1].data_ptr(), st.cuda_stream)
            ev[k + 1].record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    per = np.array([ev[k].elapsed_time(ev[k + 1]) for k in range(N)])
    q = max(N // 8, 1)
    print("A: %d one-sample frames back to back after a 0.5 s pause: %.1f ms in all; ms per frame by eighth of the run: %s" % (N, (t1 - t0) * 1e3, " ".join("%.3f" % per[i:i + q].mean() for i in range(0, N, q))))
    print("   " + cs.summary(t0, t1, 8))
ctx.set_pipelining(False)
# B / C: multi-sample frames
buf = bufs[0]
for spp in (32, 64, 128, 256):
    p = rt.make_params(W, H, spp, 3, **rt.scenes.CPU_LAUNCHER)
    ctx.render_device(p, rows, buf.data_ptr()); ctx.synchronize()
    res = []
    for mode in ("after a 0.5 s pause", "straight after 300 ms of one-sample frames"):
        if mode.startswith("after"):
            time.sleep(0.5)
        else:
            for k in range(330):
                ctx.render_device(p1, rows, bufs[1].data_ptr())
            ctx.synchronize()
        with ClockSampler() as cs:
            t0 = time.perf_counter()
            ctx.render_device(p, rows, buf.data_ptr()); ctx.synchronize()
            t1 = time.perf_counter()
        ms = ctx.stats()["kernel_ms"]
        res.append("%s: %.2f ms = %.4f per sample [%s]" % (mode, ms, ms / spp, cs.summary(t0, t1, 4)))
    print("B: %3d samples: " % spp + "\n               ".join(res), flush=True)
