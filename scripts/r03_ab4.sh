#!/bin/bash
# A/B of the two fused-middle kernel families inside the step: SCA_MID_DIRECT bit 0 = forward, bit 1 = backward direct
run() { echo "== B=${BATCH:-8192} $*"; env "$@" timeout -k 10 150 python bench.py --batch ${BATCH:-8192} --steps 300 --warmup 20 --no-cpu-baseline --no-sweep --no-loader-compare --no-families 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["ms_per_step"], d["value"], {k: round(v["avg_us"],1) for k,v in d["in_step_kernel_us"].items() if "mid" in k})' || exit 1; }
for m in 0 3 2 0 3; do run SCA_MID_DIRECT=$m; done
for b in 128 1024 4096 16384; do for m in 0 3; do BATCH=$b run SCA_MID_DIRECT=$m; done; done
