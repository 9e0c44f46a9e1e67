#!/bin/bash
# dev: step timeline of a model under an env setting: tools/gpu_kprof2.sh narre "RBR_BWD_OVERLAP=0" tag
cd "$(dirname "$0")/.."
m=$1; envs=$2; tag=$3
O=gpurun_out/kprof_$tag; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
export $envs
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r -- python3 tools/dev_count_launches.py $m 40 > $O/log.txt 2>&1
find $O -name '*kernel_trace.csv' | head -1 | xargs -I{} cp {} gpurun_out/ktrace_$tag.csv
rm -rf $O
python tools/step_timeline.py gpurun_out/ktrace_$tag.csv | cut -c1-120 | grep -v "XXsanitize\|zero_regions\|mark_scan\|PackJob\|dropout\|attn_\|head_\|mse_"
