#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout -k 10 300 python tools/dev_soak.py 3000 2>&1 | tail -4 && timeout -k 10 300 python tools/dev_soak_models.py datt 1800 2>&1 | tail -3 && timeout -k 10 300 python tools/dev_soak_models.py narre 3000 2>&1 | tail -3
