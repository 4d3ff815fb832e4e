#!/bin/bash
# A/B of cem_bwd_mid's grid size within one box (module time, shipped configuration)
export BENCH_CEM_FUSED_ONLY=1
for v in 1024 768 512 384 256 1024 512; do
  echo -n "blocks $v: "
  MMIDET_CEM_MID_BLOCKS=$v python tools/bench_cem.py 2>&1 | grep "fused forward"
done
