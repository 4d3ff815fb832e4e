#!/bin/bash
# A/B of the round-2 launch fusions inside the full training step: one bench.py run per switch setting.
# usage (on the GPU box): bash tools/ab_step.sh > gpurun_out/ab.txt
cd "$(dirname "$0")/.."
run() {
  echo "== $*"
  env "$@" python bench.py --steps 12 --warmup 3 --mode eager --no-cpu-baseline --no-split-probe --no-roofline 2>/dev/null |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.2f ms/step  %.1f img/s  host enqueue %.1f ms' % (j['ms_per_step'], j['value'], j['config']['host_enqueue_ms_per_step']))"
}
run X=1
run MMIDET_BN_FOLD=0
run MMIDET_PACK_C3=0
run MMIDET_SKIP_FUSE=0
run MMIDET_WGRAD_FOLD=0
run MMIDET_WGRAD_TABLE=0
run MMIDET_WGRAD_SPLIT_PENALTY=0.005
run MMIDET_WGRAD_SPLIT_PENALTY=0.01
run MMIDET_WGRAD_SPLIT_PENALTY=0.02
run MMIDET_WGRAD_SPLIT_PENALTY=0.01 MMIDET_WGRAD_FOLD_MAX=8
run MMIDET_BN_FOLD=0 MMIDET_PACK_C3=0 MMIDET_SKIP_FUSE=0 MMIDET_WGRAD_FOLD=0
run X=1
