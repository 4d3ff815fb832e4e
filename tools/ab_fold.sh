#!/bin/bash
# A/B of the wgrad split-K fold tree inside the full training step and on the conv micro-bench.
# usage (on the GPU box): bash tools/ab_fold.sh > gpurun_out/ab_fold.txt
cd "$(dirname "$0")/.."
AB_BENCH_ARGS="--steps 12 --warmup 3 --mode eager --no-cpu-baseline --no-split-probe --no-roofline"
. tools/ab_common.sh
run MMIDET_WGRAD_FOLD_MAX=256
run MMIDET_WGRAD_FOLD_MAX=4
run MMIDET_WGRAD_FOLD_MAX=16
run MMIDET_WGRAD_FOLD_MAX=64
run MMIDET_WGRAD_FOLD_MAX=256
run MMIDET_WGRAD_FOLD_MAX=4
for m in 4 16 256; do
  echo "== conv micro-bench, MMIDET_WGRAD_FOLD_MAX=$m"
  ab_run_cmd "bench_conv fold_max=$m" env MMIDET_WGRAD_FOLD_MAX=$m python tools/bench_conv.py
done
