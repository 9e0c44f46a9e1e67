#!/bin/bash
# dev: kernel stats of N graph replays of one model's train step: tools/gpu_kprof.sh datt|narre|deepconn [tag]
set -o pipefail
cd "$(dirname "$0")/.."
m=${1:-datt}; tag=${2:-$m}
O=gpurun_out/kprof_$tag; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r -- python3 tools/dev_count_launches.py $m 60 > $O/log.txt 2>&1
f=$(find $O -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/kstats_$tag.csv && python tools/kstats.py gpurun_out/kstats_$tag.csv 60
find $O -name '*kernel_trace.csv' | head -1 | xargs -I{} cp {} gpurun_out/ktrace_$tag.csv
rm -rf $O
exit 0
