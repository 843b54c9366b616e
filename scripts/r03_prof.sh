#!/bin/bash
# usage: r03_prof.sh TAG [VAR=VALUE ...]  -> gpurun_out/TAG_kernel_stats.csv, TAG_step_timeline.txt, TAG_under_rocprof.json
# kernel trace of the headline pass of bench.py (program directly behind `--`; knobs are exported, not passed by env)
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$OUT" -- python3 bench.py --batch ${BATCH:-8192} --steps 30 --warmup 10 --no-cpu-baseline --no-sweep --no-loader-compare --no-families > gpurun_out/${TAG}_under_rocprof.json 2> gpurun_out/${TAG}_prof.err || { tail -5 gpurun_out/${TAG}_prof.err; exit 1; }
DB=$(find "$OUT" -name '*.db' | head -1)
python3 scripts/rocpd_export.py "$DB" gpurun_out/$TAG && rm -rf "$OUT"
cat gpurun_out/${TAG}_step_timeline.txt
