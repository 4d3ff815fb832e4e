#!/bin/bash
# host sensitivity of the step: a busy-wait of n microseconds after every library call (about 3000 per step)
cd "$(dirname "$0")/.."
run() {
  echo "== $*"
  env "$@" python bench.py --steps 16 --warmup 4 --mode eager --no-cpu-baseline --no-split-probe --no-roofline 2>/dev/null |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.2f ms/step  %.1f img/s  host enqueue %.1f ms' % (j['ms_per_step'], j['value'], j['config']['host_enqueue_ms_per_step']))"
}
run MMIDET_HOST_SPIN_US=0
run MMIDET_HOST_SPIN_US=2
run MMIDET_HOST_SPIN_US=4
run MMIDET_HOST_SPIN_US=0
run MMIDET_HOST_SPIN_US=6
