#!/bin/bash
# dev: A/B bench runs on ONE box: each line of $AB_CONFIGS is "name ENV=.. ENV=.." ; bench + rocprof kernel trace per config
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -n "$TESTS" ] && [ "$TESTS" != "none" ]; then
  timeout -k 10 900 python -m pytest $TESTS -m gpu -q --maxfail=10 > gpurun_out/tests.log 2>&1
  rc=$?; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests.log | tail -15; echo "tests rc=$rc"
  [ $rc -eq 124 ] && exit $rc
fi
while IFS= read -r line; do
  [ -z "$line" ] && continue
  name=$(echo "$line" | cut -d' ' -f1); envs=$(echo "$line" | cut -s -d' ' -f2-)
  echo "=== $name [$envs]"
  env $envs timeout -k 10 200 python bench.py --no-cpu-baseline --no-variants ${BENCH_ARGS} > gpurun_out/bench_$name.log 2>&1; rc=$?
  python - "$name" <<'PY'
import json,sys
name=sys.argv[1]
try:
    l=[x for x in open(f"gpurun_out/bench_{name}.log") if x.startswith("{")][-1]; d=json.loads(l)
    print(name, "ms_per_step", d["ms_per_step"], "min", d["ms_per_step_min"], "pairs/s", d["value"], "kernels_ms", d.get("kernels_ms"))
except Exception as e:
    print(name, "no bench line:", e); print(open(f"gpurun_out/bench_{name}.log").read()[-1500:])
PY
  [ $rc -eq 124 ] && exit $rc
  if [ "${PROF:-1}" = "1" ]; then
    rm -rf gpurun_out/prof_$name
    # (the program itself after --: no env / bash hop under the profiler)
    export $envs 2>/dev/null
    timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/prof_$name -o r -- python3 bench.py --no-cpu-baseline --no-variants --steps 20 ${BENCH_ARGS} > gpurun_out/prof_$name.log 2>&1; rc=$?
    for e in $envs; do unset "${e%%=*}"; done
    python tools/prof_db.py gpurun_out/prof_$name --timeline 20 > gpurun_out/prof_$name.txt 2>&1; head -50 gpurun_out/prof_$name.txt
    rm -rf gpurun_out/prof_$name
    [ $rc -eq 124 ] && exit $rc
  fi
done <<< "$AB_CONFIGS"
exit 0
