#!/bin/bash
# GPU box: HBM bytes per launch of wf_advance / wf_travq only (two PMC passes) + the single-context launch times.   usage: tools/pmc_bytes.sh <tag>
set -e
tag=${1:-bytes}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
rm -rf $out && mkdir -p $out
i=3
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  RT_PARTS=1 timeout -k 5 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -o pmc -- python3 bench.py --no-cpu-baseline --no-end-to-end --large-steps 0 --prewarm-ms 0 --steps 4 --warmup 1 > $out/p$i.json 2> $out/p$i.err || { echo "pass $i ($grp) failed"; tail -5 $out/p$i.err; }
done
python3 tools/pmc_summary.py $out "wf_advance<false, false>" > $out/pmc_wf_advance.json || true
python3 - <<P
import json
d=json.load(open("$out/pmc_wf_advance.json"))["derived"]; print({k: d[k] for k in d if "hbm" in k})
P
python3 bench.py --no-cpu-baseline --no-end-to-end --large-steps 0 > $out/bench.json 2> $out/bench.err
python3 - <<P
import json
b=json.load(open("$out/bench.json")); print(b["value"], b["ms_per_step"], [(k["kernel"], k["kernel_ms"]) for k in b["roofline"]["kernels"]])
P
python3 - <<P
import csv, glob
for f in sorted(glob.glob("$out/p*/**/*counter_collection.csv", recursive=True)):
    rows=[r for r in csv.DictReader(open(f)) if "wf_advance<false" in r["Kernel_Name"] or "wf_travq<false" in r["Kernel_Name"]]
    rows.sort(key=lambda r:int(r["Dispatch_Id"]))
    print(f.split("/")[-3] if "/" in f else f, [(r["Kernel_Name"][8:18], r["Counter_Name"], round(float(r["Counter_Value"])/1024*(2 if r["Counter_Name"]=="FETCH_SIZE" else 1),1)) for r in rows[-20:]])
P
rm -rf $out/p[0-9]
