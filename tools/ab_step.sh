#!/bin/bash
# A/B of the round-2 launch fusions inside the full training step: one bench.py run per switch setting.
# usage (on the GPU box): bash tools/ab_step.sh > gpurun_out/ab.txt
cd "$(dirname "$0")/.."
AB_BENCH_ARGS="--steps 12 --warmup 3 --mode eager --no-cpu-baseline --no-split-probe --no-roofline"
. tools/ab_common.sh
run X=1
run MMIDET_BN_FOLD=0
run MMIDET_PACK_C3=0
run MMIDET_SKIP_FUSE=0
run MMIDET_WGRAD_FOLD=0
run MMIDET_WGRAD_TABLE=0
run MMIDET_BN_FOLD=0 MMIDET_PACK_C3=0 MMIDET_SKIP_FUSE=0 MMIDET_WGRAD_FOLD=0
run X=1
