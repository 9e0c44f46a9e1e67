#!/bin/bash
# dev (round 4): full gpu suite + the driver's bench line (with configs)
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/tests.log 2>&1
rc=$?; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests.log | tail -15; echo "tests rc=$rc"
[ $rc -eq 124 ] && exit $rc
( time timeout -k 10 500 python bench.py --no-cpu-baseline ) > gpurun_out/bench_cfgs.log 2> gpurun_out/bench_cfgs.err; rc=$?
tail -1 gpurun_out/bench_cfgs.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('headline', d['ms_per_step'], d['value'])
for k,v in d.get('configs',{}).items():
    print(k, json.dumps({a:b for a,b in v.items() if a not in ('workload','fixture','launch','kernels_ms_note')})[:1500])
"
tail -5 gpurun_out/bench_cfgs.err
echo "bench rc=$rc"
exit 0
