#!/bin/bash
# Diagnosis aid: repeats bench.py's two-rank rehearsal (gloo, both ranks on GPU 0) until the first failure and keeps
# that run's complete stderr (both ranks' faulthandler stacks).  usage: rehearsal_probe.sh <reps> <launch> <dp-mmd>
reps=${1:-5}; launch=${2:-auto}; mmd=${3:-local}
export SCA_BENCH_REHEARSAL=1 SCA_BENCH_HANG_S=40 HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=2
for i in $(seq 1 $reps); do
  port=$((29500 + RANDOM % 2000))
  timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port $port bench.py --gpus 2 --steps 4 --warmup 2 --batch 1024 --cells 40000 --no-cpu-baseline --no-sweep \
    --dp-mmd $mmd --launch $launch > gpurun_out/reh_out.log 2> gpurun_out/reh_err.log
  rc=$?
  echo "rep=$i launch=$launch mmd=$mmd rc=$rc"
  if [ $rc -ne 0 ]; then exit 1; fi
done
