#!/bin/bash
# dev: full gpu suite + the driver's default bench command + 2-rank rehearsal of bench.py over gloo on one device
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/tests.log 2>&1
rc=$?; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests.log | tail -15; echo "tests rc=$rc"
[ $rc -eq 124 ] && exit $rc
timeout -k 10 400 python bench.py > gpurun_out/bench_full.log 2>&1; rc=$?
tail -1 gpurun_out/bench_full.log | cut -c1-3000; echo "bench rc=$rc"
[ $rc -eq 124 ] && exit $rc
RBR_BENCH_SINGLE_DEVICE=1 RBR_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline ${DP2_ARGS} > gpurun_out/bench_dp2.log 2>&1; rc=$?
tail -1 gpurun_out/bench_dp2.log | cut -c1-1500; echo "dp2 rc=$rc"
exit 0
