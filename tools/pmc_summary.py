"""Averages rocprofv3 --pmc counter rows of the render kernel over its dispatches (GPU box helper)."""
import csv, glob, json, os, sys
out = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "render_"
acc, n = {}, {}
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if pat not in k:
            continue
        if "<true" in k:
            continue
        c = row["Counter_Name"]; v = float(row["Counter_Value"])
        acc[c] = acc.get(c, 0.0) + v; n[c] = n.get(c, 0) + 1
        for extra in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Accum_VGPR_Count"):
            if extra in row:
                acc["_" + extra] = float(row[extra]); n["_" + extra] = 1
res = {c: acc[c] / n[c] for c in sorted(acc)}
d = {}
g = res.get
if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU"):
    d["valu_lane_utilization"] = g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64)
if g("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA"):
        if g(k) is not None:
            d[k + "/WAVE_CYCLES"] = g(k) / g("SQ_WAVE_CYCLES")
if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
    d["valu_insts_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")
if g("GRBM_GUI_ACTIVE"):
    cyc = g("GRBM_GUI_ACTIVE") / 8.0                      # rocprofv3 reports the sum over the 8 XCDs (MI355X_MICROARCH.md, DVFS)
    d["kernel_cycles"] = cyc
    if g("SQ_INSTS_VALU"):
        d["valu_issue_per_simd_cycle"] = g("SQ_INSTS_VALU") / (cyc * 1024)           # wave-instructions per SIMD and cycle (1024 SIMDs)
    if g("SQ_ACTIVE_INST_VALU"):
        d["valu_pipe_busy_frac"] = g("SQ_ACTIVE_INST_VALU") * 4 / (cyc * 1024)       # SQ_ACTIVE_INST_* count quad-cycles
    if g("SQ_INSTS_SALU"):
        d["salu_issue_per_cu_cycle"] = g("SQ_INSTS_SALU") / (cyc * 256)              # one scalar unit per CU
if g("FETCH_SIZE") is not None:
    d["hbm_read_bytes_corrected"] = g("FETCH_SIZE") * 1024 * 2      # gfx950: FETCH_SIZE reads 1/2 (MI355X_MICROARCH.md HBM)
if g("WRITE_SIZE") is not None:
    d["hbm_write_bytes"] = g("WRITE_SIZE") * 1024
if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
    d["l2_hit_rate"] = g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1)
print(json.dumps({"counters_avg_per_dispatch": res, "derived": d}, indent=1))
