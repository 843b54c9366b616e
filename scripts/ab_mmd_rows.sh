#!/bin/bash
# A/B of the whole-minibatch MMD dispatch (symmetric kernel vs row-block instantiation) inside the bench step.
for b in ${BATCHES:-8192 4096}; do for m in 0 16384 0 16384; do
  echo "B=$b SCA_MMD_ROWS_MAX=$m $(SCA_MMD_ROWS_MAX=$m timeout -k 10 120 python bench.py --batch $b --steps 200 --warmup 20 --no-cpu-baseline --no-sweep | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["ms_per_step"], d["value"])')" || exit 1
done; done
