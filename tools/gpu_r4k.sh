#!/bin/bash
# dev: D-ATT backward over two streams
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_random_gpu.py tests/test_siamese_gpu.py tests/test_narre_datt_gpu.py tests/test_datt_pair_gpu.py tests/test_fused_step_gpu.py tests/test_graph_step_gpu.py tests/test_trainer_gpu.py -m gpu -q -x > gpurun_out/tests_k.log 2>&1
rc=$?; tail -4 gpurun_out/tests_k.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/bench_models.py datt 2>/dev/null | tail -1 | cut -c1-600
O=gpurun_out/kp_d; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r -- python3 tools/dev_count_launches.py datt 40 > $O/log.txt 2>&1
f=$(find $O -name '*kernel_trace.csv' | head -1); python tools/step_timeline.py $f > gpurun_out/timeline_datt.txt
rm -rf $O
head -1 gpurun_out/timeline_datt.txt
