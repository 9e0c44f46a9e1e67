#!/bin/bash
# dev: tools/bench_models.py under several env configs on ONE box; each line of $AB_CONFIGS: "name model [ENV=..]..."
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
while IFS= read -r line; do
  [ -z "$line" ] && continue
  name=$(echo "$line" | cut -d' ' -f1); model=$(echo "$line" | cut -d' ' -f2); envs=$(echo "$line" | cut -s -d' ' -f3-)
  echo "=== $name $model [$envs]"
  env $envs timeout -k 10 300 python tools/bench_models.py $model ${MODEL_ARGS} > gpurun_out/models_$name.log 2>&1; rc=$?
  grep '^{' gpurun_out/models_$name.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['model'][:28], 'graph_ms', d['train_graph_ms'], 'eager_ms', d['train_ms'], 'fwd_ms', d['fwd_ms'])" || tail -5 gpurun_out/models_$name.log
  [ $rc -eq 124 ] && exit $rc
  if [ "${PROF:-0}" = "1" ]; then
    export $envs 2>/dev/null
    rm -rf gpurun_out/mprof_$name
    timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/mprof_$name -o r -- python3 tools/bench_models.py $model ${MODEL_ARGS} > gpurun_out/mprof_$name.log 2>&1
    for e in $envs; do unset "${e%%=*}"; done
    python tools/prof_db.py gpurun_out/mprof_$name --timeline 130 > gpurun_out/mprof_$name.txt 2>&1; head -4 gpurun_out/mprof_$name.txt | cut -c1-120
    rm -rf gpurun_out/mprof_$name
  fi
done <<< "$AB_CONFIGS"
exit 0
