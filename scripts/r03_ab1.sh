#!/bin/bash
# round-3 A/B no. 1: baseline vs the MMD as a low-occupancy background kernel (few long workgroups, no LDS reservation)
run() { echo "== $*"; env "$@" timeout -k 10 150 python bench.py --batch 8192 --steps 200 --warmup 20 --no-cpu-baseline --no-sweep --no-loader-compare 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["ms_per_step"], d["value"])' || exit 1; }
run A=0
run SCA_MMD_ROWS_JR=16 SCA_MMD_LDS_PAD=0 SCA_MMD_AFTER_OUT=0
run SCA_MMD_ROWS_JR=32 SCA_MMD_LDS_PAD=0 SCA_MMD_AFTER_OUT=0
run SCA_MMD_ROWS_JR=8 SCA_MMD_LDS_PAD=0 SCA_MMD_AFTER_OUT=0
run SCA_MMD_ROWS_JR=16 SCA_MMD_LDS_PAD=0 SCA_MMD_AFTER_OUT=1
run SCA_MMD_ROWS_JR=32 SCA_MMD_LDS_PAD=24000 SCA_MMD_AFTER_OUT=0
