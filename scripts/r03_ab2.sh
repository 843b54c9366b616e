#!/bin/bash
# round-3 A/B no. 2: fused output layer + likelihood vs the unfused pair, with the MMD fork variants
run() { echo "== $*"; env "$@" timeout -k 10 150 python bench.py --batch ${BATCH:-8192} --steps 200 --warmup 20 --no-cpu-baseline --no-sweep --no-loader-compare 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["ms_per_step"], d["value"])' || exit 1; }
run SCA_OUT_FUSED=0
run SCA_OUT_FUSED=1
run SCA_OUT_FUSED=1 SCA_MMD_AFTER_OUT=1
run SCA_OUT_FUSED=1 SCA_MMD_AFTER_OUT=0 SCA_MMD_LDS_PAD=0
run SCA_OUT_FUSED=1 SCA_MMD_AFTER_OUT=1 SCA_MMD_LDS_PAD=0
run SCA_OUT_FUSED=1 SCA_MMD_AFTER_OUT=0 SCA_MMD_LDS_PAD=24000
BATCH=4096 run SCA_OUT_FUSED=0
BATCH=4096 run SCA_OUT_FUSED=1
