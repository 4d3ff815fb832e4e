#!/bin/bash
# where the native transport's 3 ms come from: the same step with the reducer's hooks off (nothing of the reducer runs)
cd "$(dirname "$0")/.."
AB_FMT="'%.2f ms/step  %.1f img/s  host enqueue %.1f ms  %s' % (j['ms_per_step'], j['value'], j['config']['host_enqueue_ms_per_step'], j['config']['gradient_transport'])"
AB_BENCH_ARGS="--ddp --steps 16 --warmup 4 --mode eager --no-cpu-baseline --no-split-probe --no-roofline"
. tools/ab_common.sh
run MMIDET_COMM=native MMIDET_DDP_DEBUG=nohooks
run MMIDET_COMM=torch MMIDET_DDP_DEBUG=nohooks
run MMIDET_COMM=native MMIDET_DDP_DEBUG=nohooks
run MMIDET_COMM=torch MMIDET_DDP_DEBUG=nohooks
AB_BENCH_ARGS="--steps 16 --warmup 4 --mode eager --no-cpu-baseline --no-split-probe --no-roofline"
run X=1
