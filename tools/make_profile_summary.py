"""Build container: condense a round's rocprofv3 files (tools/round_profile.sh, copied to profiles/<round>/) into
profiles/<round>/summary.json -- the per-kernel figures bench.py quotes in its `roofline.binding` object.
usage: python tools/make_profile_summary.py profiles/round2"""
import csv, json, os, sys

d = sys.argv[1]
HBM_PEAK = 8.0e12


def stats(path):
    out = {}
    for r in csv.DictReader(open(path)):
        out[r["Name"]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "share_of_gpu_time": float(r["Percentage"]) / 100}
    return out


two = stats(os.path.join(d, "bench_kernel_stats.csv"))
one = stats(os.path.join(d, "single_stream_kernel_stats.csv"))
res = {"source": {"bench_kernel_stats": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 (two concurrent sub-frames)",
                  "single_stream_kernel_stats": "the same with RT_PARTS=1: one sub-frame, every launch owns the chip",
                  "pmc": "rocprofv3 --kernel-trace --pmc <group> (one group per pass), RT_PARTS=1; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950",
                  "files": sorted(os.listdir(d))},
       "kernels": {}}
for key, pat in (("wf_travq", "wf_travq<false"), ("wf_advance", "wf_advance<false, false>")):
    pm = json.load(open(os.path.join(d, "pmc_%s.json" % key)))
    c, dv = pm["counters_avg_per_dispatch"], pm["derived"]
    n2 = next(v for k, v in two.items() if pat in k)
    n1 = next(v for k, v in one.items() if pat in k)
    hbm = dv.get("hbm_read_bytes_corrected", 0) + dv.get("hbm_write_bytes", 0)
    res["kernels"][key] = {
        "rocprof_avg_us_two_streams": round(n2["avg_us"], 2), "share_of_gpu_time": round(n2["share_of_gpu_time"], 4),
        "rocprof_avg_us_single_stream": round(n1["avg_us"], 2),
        "hbm_read_bytes_per_launch": int(dv.get("hbm_read_bytes_corrected", 0)), "hbm_write_bytes_per_launch": int(dv.get("hbm_write_bytes", 0)),
        "hbm_GBps_single_stream": round(hbm / (n1["avg_us"] * 1e-6) / 1e9, 1), "hbm_frac_of_8TBps": round(hbm / (n1["avg_us"] * 1e-6) / HBM_PEAK, 4),
        "l2_hit_rate": round(dv.get("l2_hit_rate", 0), 4),
        "valu_wave_insts_per_launch": int(c.get("SQ_INSTS_VALU", 0)), "salu_wave_insts_per_launch": int(c.get("SQ_INSTS_SALU", 0)),
        "lds_wave_insts_per_launch": int(c.get("SQ_INSTS_LDS", 0)), "vmem_rd_wave_insts_per_launch": int(c.get("SQ_INSTS_VMEM_RD", 0)),
        "valu_pipe_busy_frac": round(dv.get("valu_pipe_busy_frac", 0), 4), "valu_issue_per_simd_cycle": round(dv.get("valu_issue_per_simd_cycle", 0), 4),
        "salu_issue_per_cu_cycle": round(dv.get("salu_issue_per_cu_cycle", 0), 4),
        "valu_lane_utilization": round(dv.get("valu_lane_utilization", 0), 4),
        "wave_cycles_waiting_frac": round(dv.get("SQ_WAIT_ANY/WAVE_CYCLES", 0), 4), "wave_cycles_issue_stalled_frac": round(dv.get("SQ_WAIT_INST_ANY/WAVE_CYCLES", 0), 4),
        "waves_per_launch": int(c.get("SQ_WAVES", 0)), "vgprs": c.get("_VGPR_Count"), "sgprs": c.get("_SGPR_Count"),
    }
    if key == "wf_advance":                                           # one-wave workgroups, one lane per path: bench.py scales the measured bytes to its own launch size
        res["kernels"][key]["paths_per_launch"] = int(c.get("SQ_WAVES", 0)) * 64
# the code the counters were taken from (bench.py quotes `traffic` only when its own sources hash to the same value) and the check of
# bench.py's in-run instruction estimate (step counters x static per-step counts) against the hardware counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
res["code_hash"] = bench.code_hash()
try:
    line = json.loads([l for l in open(os.path.join(d, "bench_n1.json")) if l.startswith("{")][-1])
    r = line["roofline"]
    est, pmc = r["valu_wave_insts_per_launch"], res["kernels"]["wf_travq"]["valu_wave_insts_per_launch"]
    ests, pmcs = r["salu_wave_insts_per_launch"], res["kernels"]["wf_travq"]["salu_wave_insts_per_launch"]
    res["instruction_estimate_check"] = {"bench_estimate_valu_per_launch": est, "pmc_SQ_INSTS_VALU_per_launch": pmc, "ratio": round(est / pmc, 4),
                                         "bench_estimate_salu_per_launch": ests, "pmc_SQ_INSTS_SALU_per_launch": pmcs, "salu_ratio": round(ests / pmcs, 4),
                                         "bench_frac": r["frac"], "driver_style_ms_per_step": line["ms_per_step"]}
except Exception as e:
    res["instruction_estimate_check"] = {"skipped": str(e)}
json.dump(res, open(os.path.join(d, "summary.json"), "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
