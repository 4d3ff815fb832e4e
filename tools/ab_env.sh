#!/bin/bash
# A/B of one environment switch inside the full training step, interleaved on one box:  bash tools/ab_env.sh VAR=VALUE [repeats]
cd "$(dirname "$0")/.."
run() {
  echo "== $*"
  env "$@" python bench.py --steps 16 --warmup 4 --mode eager --no-cpu-baseline --no-split-probe --no-roofline 2>/dev/null |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.2f ms/step  %.1f img/s  host enqueue %.1f ms' % (j['ms_per_step'], j['value'], j['config']['host_enqueue_ms_per_step']))"
}
for i in $(seq 1 ${2:-3}); do
  run X=1
  run "$1"
done
