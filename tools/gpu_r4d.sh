#!/bin/bash
# dev (round 4): NARRE tests + models bench + kernel stats, one call
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest ${TESTS:-tests/test_fused_step_gpu.py tests/test_narre_datt_gpu.py tests/test_precision_gpu.py} -m gpu -q -x --maxfail=3 > gpurun_out/tests_b.log 2>&1
rc=$?; grep -E "^(FAILED|ERROR)|passed|failed|Error|error" gpurun_out/tests_b.log | tail -15; echo "tests rc=$rc"
[ $rc -ne 0 ] && tail -40 gpurun_out/tests_b.log && exit 0
timeout -k 10 300 python tools/bench_models.py narre > gpurun_out/models_narre.log 2>&1; tail -1 gpurun_out/models_narre.log | cut -c1-300
bash tools/gpu_kprof.sh narre | cut -c1-150 | head -${KLINES:-30}
exit 0
