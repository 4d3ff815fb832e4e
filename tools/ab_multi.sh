#!/bin/bash
# several environment settings against the default, interleaved on one box:  bash tools/ab_multi.sh "A=1" "B=2 C=3" ...
cd "$(dirname "$0")/.."
run() {
  echo "== $*"
  env $* python bench.py --steps 16 --warmup 4 --mode eager --no-cpu-baseline --no-split-probe --no-roofline 2>/dev/null |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.2f ms/step  %.1f img/s  host enqueue %.1f ms' % (j['ms_per_step'], j['value'], j['config']['host_enqueue_ms_per_step']))"
}
for rep in 1 2; do
  run X=1
  for cfg in "$@"; do run "$cfg"; done
done
