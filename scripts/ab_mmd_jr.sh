#!/bin/bash
# A/B of the column chunks per workgroup of the row-block MMD inside the bench step.
for b in ${BATCHES:-8192 4096}; do for jr in ${JRS:-0 2 4 16 0 32}; do
  echo "B=$b SCA_MMD_ROWS_JR=$jr $(SCA_MMD_ROWS_JR=$jr timeout -k 10 120 python bench.py --batch $b --steps 200 --warmup 20 --no-cpu-baseline --no-sweep | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["ms_per_step"], d["value"])')" || exit 1
done; done
