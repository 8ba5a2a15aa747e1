#!/bin/bash
# GPU box: rocprofv3 kernel trace of the default bench command; copies the stats summary to profiles/.
# usage: tools/profile_bench.sh <tag> [bench args]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- python3 bench.py --no-cpu-baseline "$@" > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
cat $out/bench.json
head -8 $out/kernel_stats.csv
