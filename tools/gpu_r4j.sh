#!/bin/bash
# dev: bf16-class backward of NARRE
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_precision_gpu.py tests/test_narre_datt_gpu.py tests/test_fused_step_gpu.py -m gpu -q -x > gpurun_out/tests_j.log 2>&1
rc=$?; tail -3 gpurun_out/tests_j.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/bench_models.py narre --precision=bf16 2>/dev/null | tail -1 | cut -c1-700
timeout -k 10 300 python tools/bench_models.py narre 2>/dev/null | tail -1 | cut -c1-400
O=gpurun_out/kp_nb; rm -rf $O; mkdir -p $O
RBR_PROD_PRECISION=bf16 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r -- python3 tools/dev_count_launches.py narre 40 > $O/log.txt 2>&1
f=$(find $O -name '*kernel_trace.csv' | head -1); python tools/step_timeline.py $f | cut -c1-130 > gpurun_out/timeline_narre_bf16.txt
rm -rf $O
cat gpurun_out/timeline_narre_bf16.txt
