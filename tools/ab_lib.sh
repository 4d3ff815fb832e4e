#!/bin/bash
# A/B of two builds of the kernel library: conv micro-bench and the full step, interleaved.  bash tools/ab_lib.sh /path/to/other.so
cd "$(dirname "$0")/.."
OTHER=$1
run() {
  echo "== $*"
  env "$@" python bench.py --steps 16 --warmup 4 --mode eager --no-cpu-baseline --no-split-probe --no-roofline 2>/dev/null |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.2f ms/step  %.1f img/s  host enqueue %.1f ms' % (j['ms_per_step'], j['value'], j['config']['host_enqueue_ms_per_step']))"
}
echo "== conv micro-bench, this build"; python tools/bench_conv.py 2>/dev/null
echo "== conv micro-bench, $OTHER"; MMIDET_HIP_LIB=$OTHER python tools/bench_conv.py 2>/dev/null
for i in 1 2 3; do
  run X=1
  run MMIDET_HIP_LIB=$OTHER
done
