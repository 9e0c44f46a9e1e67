#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_textcnn_edges_gpu.py -m gpu -q -x > gpurun_out/tests_e.log 2>&1; tail -2 gpurun_out/tests_e.log
for k in 0 1; do
  RBR_DBG_K=$k timeout -k 10 300 python tools/bench_models.py datt --no-graph 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dbg=$k', d['kernels_ms'])"
done
RBR_DBG_K=3 timeout -k 10 300 python tools/bench_models.py datt --no-graph > gpurun_out/dbgk.out 2> gpurun_out/dbgk.err
grep -c DBGK gpurun_out/dbgk.err
