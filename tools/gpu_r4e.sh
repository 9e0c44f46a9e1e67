#!/bin/bash
# dev (round 4): the rows-stationary GEMM -- edge tests, D-ATT tests, D-ATT bench with / without it
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_textcnn_edges_gpu.py tests/test_narre_datt_gpu.py tests/test_datt_pair_gpu.py -m gpu -q -x > gpurun_out/tests_e.log 2>&1
rc=$?; tail -5 gpurun_out/tests_e.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
for f in 1 0; do
  RBR_GEMM_ROWS_STATIONARY=$f timeout -k 10 300 python tools/bench_models.py datt > gpurun_out/datt_stat$f.log 2>&1 || exit 1
  tail -1 gpurun_out/datt_stat$f.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('stationary=$f', d['train_graph_ms'], d['fwd_ms'], d['kernels_ms'])"
done
