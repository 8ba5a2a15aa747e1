#!/bin/bash
# GPU box: kernel-trace stats of the default bench (two sub-frames) and of RT_PARTS=1 (every launch owns the chip).  usage: tools/quick_profile.sh <tag>
set -e
tag=${1:-quick}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t1 -o trace -- python3 bench.py --no-cpu-baseline --large-steps 0 --steps 30 --warmup 5 > $out/bench_under_rocprof.json 2> $out/t1.err || { tail -5 $out/t1.err; exit 1; }
cp $(find $out/t1 -name "*kernel_stats.csv" | head -1) $out/bench_kernel_stats.csv
RT_PARTS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t2 -o trace -- python3 bench.py --no-cpu-baseline --large-steps 0 --steps 30 --warmup 5 > $out/single_stream_bench.json 2> $out/t2.err || { tail -5 $out/t2.err; exit 1; }
cp $(find $out/t2 -name "*kernel_stats.csv" | head -1) $out/single_stream_kernel_stats.csv
rm -rf $out/t1 $out/t2
head -5 $out/bench_kernel_stats.csv | cut -c1-200; head -5 $out/single_stream_kernel_stats.csv | cut -c1-200
