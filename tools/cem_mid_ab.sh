#!/bin/bash
# A/B of the CEM backward forms within one box (module time at 16 x 640 x 640)
export BENCH_CEM_FUSED_ONLY=1
for v in "MMIDET_CEM_BWD_BN=0" "MMIDET_CEM_BWD_BN=1" "MMIDET_CEM_BWD_BN=0" "MMIDET_CEM_BWD_BN=1"; do
  echo -n "$v: "
  env $v python tools/bench_cem.py 2>&1 | grep "fused forward"
done
