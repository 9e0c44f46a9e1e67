#!/bin/bash
# dev: A/B of two builds of the library on ONE box: tools/gpu_ab2.sh <path of build A> [bench args]; B = the in-tree build
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
A=$1; shift
for r in 1 2; do
  for v in A B; do
    if [ $v = A ]; then export RBR_LIB_PATH=$A; else unset RBR_LIB_PATH; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-variants --no-configs "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['ms_per_step_min'], d['value'])"
  done
done
