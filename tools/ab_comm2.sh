#!/bin/bash
# where the native transport's 3 ms come from: the same step with the reducer's hooks off (nothing of the reducer runs)
cd "$(dirname "$0")/.."
run() {
  echo "== $*"
  env "$@" python bench.py $EXTRA --steps 16 --warmup 4 --mode eager --no-cpu-baseline --no-split-probe --no-roofline 2>/dev/null |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.2f ms/step  %.1f img/s  host enqueue %.1f ms  %s' % (j['ms_per_step'], j['value'], j['config']['host_enqueue_ms_per_step'], j['config']['gradient_transport']))"
}
EXTRA=--ddp
run MMIDET_COMM=native MMIDET_DDP_DEBUG=nohooks
run MMIDET_COMM=torch MMIDET_DDP_DEBUG=nohooks
run MMIDET_COMM=native MMIDET_DDP_DEBUG=nohooks
run MMIDET_COMM=torch MMIDET_DDP_DEBUG=nohooks
EXTRA=
run X=1
