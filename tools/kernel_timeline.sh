#!/bin/bash
# GPU box: kernel trace (start / end of every launch) of a few frames of the default bench, reduced to one frame's timeline:
# which launches of the two sub-frames' streams ran beside each other.  usage: tools/kernel_timeline.sh <tag>
set -e
tag=${1:-timeline}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/t -o trace -- python3 bench.py --no-cpu-baseline --no-end-to-end --large-steps 0 --steps 6 --warmup 3 > $out/bench.json 2> $out/t.err || { tail -5 $out/t.err; exit 1; }
cp $(find $out/t -name "*kernel_trace.csv" | head -1) $out/kernel_trace.csv
rm -rf $out/t
python3 tools/kernel_timeline.py $out/kernel_trace.csv > $out/timeline.txt
tail -40 $out/timeline.txt
