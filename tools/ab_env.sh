#!/bin/bash
# A/B of one environment switch inside the full training step, interleaved on one box:  bash tools/ab_env.sh VAR=VALUE [repeats]
cd "$(dirname "$0")/.."
. tools/ab_common.sh
for i in $(seq 1 ${2:-3}); do
  run X=1
  run "$1"
done
