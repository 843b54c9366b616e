#!/bin/bash
# r03_ab6.sh LAUNCH BATCH [VAR=VAL ...] : one 300-step bench run, prints ms/step, cells/s, host issue ms
L=$1; B=$2; shift 2
echo "== $L $B $*"
env "$@" timeout -k 10 150 python bench.py --batch $B --launch $L --steps 300 --warmup 20 --no-cpu-baseline --no-sweep --no-loader-compare --no-families 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(round(d["ms_per_step"],4), round(d["value"]), round(d["host_issue_ms_per_step"],3))'
