#!/bin/bash
# native RCCL transport vs torch.distributed/ProcessGroupNCCL at world size 1 (the data-parallel code path on one GPU),
# interleaved on one box:  bash tools/ab_comm.sh [repeats]
cd "$(dirname "$0")/.."
AB_BENCH_ARGS="--ddp --steps 16 --warmup 4 --mode eager --no-cpu-baseline --no-split-probe --no-roofline"
AB_FMT="'%.2f ms/step  %.1f img/s  host enqueue %.1f ms  %s' % (j['ms_per_step'], j['value'], j['config']['host_enqueue_ms_per_step'], j['config']['gradient_transport'])"
. tools/ab_common.sh
for i in $(seq 1 ${1:-2}); do
  run MMIDET_COMM=native
  run MMIDET_COMM=torch
done
run MMIDET_COMM=native MMIDET_DDP_DEBUG=noreduce
