#!/usr/bin/env python3
"""Static instruction counts per kernel of the gfx950 code object: `hipcc -S --cuda-device-only` of rt_capi.hip with the build's flags, one line per kernel whose
demangled name matches the pattern (total instructions, vector instructions, issue-weighted vector instructions as tools/static_counts.py weighs them, VGPRs, scratch bytes).
    python3 tools/kernel_static.py [pattern] [source-root]        e.g.  python3 tools/kernel_static.py wf_advance /tmp/head
"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import static_counts as sc   # noqa: E402

pat = sys.argv[1] if len(sys.argv) > 1 else "wf_"
root = sys.argv[2] if len(sys.argv) > 2 else sc.ROOT
sys.path.insert(0, sc.ROOT)
import __graft_entry__ as g   # noqa: E402
flags = [f for f in g.HIP_FLAGS if f not in ("-shared", "-fPIC")]
src = os.path.join(root, "raytracinggpu_amd", "csrc", "rt_capi.hip")
asm = subprocess.run([g.HIPCC, *flags, "-S", "--cuda-device-only", "-o", "-", src], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
names = re.findall(r"^(_Z\w+):", asm, re.M)
dem = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True).stdout.splitlines()
for m, d in zip(names, dem):
    if not re.search(pat, d):
        continue
    lines = sc.kernel_text(asm, m)
    c = sc.count(lines)
    meta = asm[asm.index(m + ":"):]
    vg = re.search(r"; NumVgprs: (\d+)", meta)
    scr = re.search(r"; ScratchSize: (\d+)", meta)
    print(f"{c['valu'] + c['salu'] + c['branch'] + c['lds'] + c['vmem'] + c['smem'] + c['other']:6d} insts  {c['valu']:5d} valu (weight {c['valu_weight']:5d})  {c['salu']:5d} salu  "
          f"{c['vmem']:4d} vmem  vgprs {vg.group(1) if vg else '?':>3}  scratch {scr.group(1) if scr else '?':>4}  {d[:110]}")
