#!/bin/bash
# dev: step timelines (graph replays) of deepconn cfg2 and datt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for m in deepconn datt; do
O=gpurun_out/kp_$m; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r -- python3 tools/dev_count_launches.py $m 40 > $O/log.txt 2>&1
f=$(find $O -name '*kernel_trace.csv' | head -1); python tools/step_timeline.py $f > gpurun_out/timeline_$m.txt
rm -rf $O
done
cut -c1-150 gpurun_out/timeline_deepconn.txt
