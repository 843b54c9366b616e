#!/bin/bash
run() { echo "== $*"; env "$@" timeout -k 10 150 python bench.py --batch ${BATCH:-8192} --steps 300 --warmup 20 --no-cpu-baseline --no-sweep --no-loader-compare 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["ms_per_step"], d["value"])' || exit 1; }
run SCA_GAMMA_TABLE=0
run SCA_GAMMA_TABLE=1
run SCA_GAMMA_TABLE=0
run SCA_GAMMA_TABLE=1
BATCH=4096 run SCA_GAMMA_TABLE=0
BATCH=4096 run SCA_GAMMA_TABLE=1
