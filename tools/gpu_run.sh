#!/bin/bash
# dev: one gpurun call = tests subset + bench + kernel trace; outputs under gpurun_out/
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
T="${TESTS:-tests/test_fused_step_gpu.py tests/test_deepconn_gpu.py tests/test_graph_step_gpu.py tests/test_optim_gpu.py}"
if [ "$T" != "none" ]; then
  timeout -k 10 900 python -m pytest $T -m gpu -q --maxfail=8 > gpurun_out/tests.log 2>&1
  rc=$?; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests.log | tail -15; echo "tests rc=$rc"
  [ $rc -eq 124 ] && exit $rc
fi
timeout -k 10 300 python bench.py --no-cpu-baseline --no-variants ${BENCH_ARGS} > gpurun_out/bench.log 2>&1; rc=$?
tail -3 gpurun_out/bench.log | cut -c1-1500; echo "bench rc=$rc"
[ $rc -ne 0 ] && exit $rc
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r -- python3 bench.py --no-cpu-baseline --no-variants ${BENCH_ARGS} > gpurun_out/prof.log 2>&1; rc=$?
echo "prof rc=$rc"
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/kernel_stats.csv && python tools/kstats.py gpurun_out/kernel_stats.csv 2>/dev/null | head -40
exit 0
