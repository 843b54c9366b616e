#!/bin/bash
# usage: r03_pmc.sh [BATCH]  -> gpurun_out/r03_pmc_summary_B<BATCH>.json
# Three separate `rocprofv3 --pmc` passes (counters only, no traces) of the bench command, eager launches so that every
# kernel is its own dispatch; aggregated per kernel by scripts/pmc_summary.py.  FETCH_SIZE / WRITE_SIZE are in KB.
B=${1:-8192}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
CMD="python3 bench.py --batch $B --steps 5 --warmup 2 --launch eager --no-cpu-baseline --no-sweep --no-loader-compare --no-families"
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c
timeout -s KILL 150 rocprofv3 --output-format csv --pmc FETCH_SIZE -d gpurun_out/pmc_a -- $CMD > gpurun_out/pmc_a.log 2>&1 || { tail -5 gpurun_out/pmc_a.log; exit 1; }
timeout -s KILL 150 rocprofv3 --output-format csv --pmc WRITE_SIZE -d gpurun_out/pmc_b -- $CMD > gpurun_out/pmc_b.log 2>&1 || { tail -5 gpurun_out/pmc_b.log; exit 1; }
timeout -s KILL 150 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmc_c -- $CMD > gpurun_out/pmc_c.log 2>&1 || { tail -5 gpurun_out/pmc_c.log; exit 1; }
python3 scripts/pmc_summary.py gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c > gpurun_out/r03_pmc_summary_B$B.json && rm -rf gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c
python3 - <<PY
import json
d = json.load(open("gpurun_out/r03_pmc_summary_B$B.json"))
for k, v in sorted(d.items()):
    print(k[:48].ljust(48), {c: round(x["mean"]) for c, x in v.items() if c in ("SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")})
PY
