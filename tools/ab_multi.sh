#!/bin/bash
# several environment settings against the default, interleaved on one box:  bash tools/ab_multi.sh "A=1" "B=2 C=3" ...
cd "$(dirname "$0")/.."
. tools/ab_common.sh
for rep in 1 2; do
  run X=1
  for cfg in "$@"; do run $cfg; done
done
