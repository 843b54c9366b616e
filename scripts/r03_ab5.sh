#!/bin/bash
# generic env A/B: r03_ab5.sh "VAR=VAL ..." "VAR=VAL ..." ...   (BATCH env selects the batch size)
run() { echo "== B=${BATCH:-8192} $*"; env $* timeout -k 10 150 python bench.py --batch ${BATCH:-8192} --steps 300 --warmup 20 --no-cpu-baseline --no-sweep --no-loader-compare --no-families 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["ms_per_step"], d["value"], {k.replace("sca_",""): round(v["avg_us"],1) for k,v in d["in_step_kernel_us"].items()})' || exit 1; }
for cfg in "$@"; do run "$cfg"; done
