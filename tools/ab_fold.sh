#!/bin/bash
# A/B of the wgrad split-K fold tree inside the full training step and on the conv micro-bench.
# usage (on the GPU box): bash tools/ab_fold.sh > gpurun_out/ab_fold.txt
cd "$(dirname "$0")/.."
run() {
  echo "== $*"
  env "$@" python bench.py --steps 12 --warmup 3 --mode eager --no-cpu-baseline --no-split-probe --no-roofline 2>/dev/null |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.2f ms/step  %.1f img/s  host enqueue %.1f ms' % (j['ms_per_step'], j['value'], j['config']['host_enqueue_ms_per_step']))"
}
run MMIDET_WGRAD_FOLD_MAX=256
run MMIDET_WGRAD_FOLD_MAX=4
run MMIDET_WGRAD_FOLD_MAX=16
run MMIDET_WGRAD_FOLD_MAX=64
run MMIDET_WGRAD_FOLD_MAX=256
run MMIDET_WGRAD_FOLD_MAX=4
for m in 4 16 256; do
  echo "== conv micro-bench, MMIDET_WGRAD_FOLD_MAX=$m"
  MMIDET_WGRAD_FOLD_MAX=$m python tools/bench_conv.py 2>/dev/null
done
