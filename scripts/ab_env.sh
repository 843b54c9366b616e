#!/bin/bash
# A/B of one environment knob inside the bench step.  usage: VAR=NAME VALS="a b c" BATCHES="8192" ab_env.sh
for b in ${BATCHES:-8192}; do for v in $VALS; do
  echo "B=$b $VAR=$v $(env $VAR=$v timeout -k 10 120 python bench.py --batch $b --steps 200 --warmup 20 --no-cpu-baseline --no-sweep | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["ms_per_step"], d["value"])')" || exit 1
done; done
