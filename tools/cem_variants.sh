#!/bin/bash
# A/B of the fused CEM forward's forms (one process each: the switches are read once)
export BENCH_CEM_FUSED_ONLY=1
for v in "2 0" "3 0" "4 0" "4 1"; do
  set -- $v
  echo "MMIDET_CEM_OB=$1 MMIDET_CEM_FORM=$2"
  MMIDET_CEM_OB=$1 MMIDET_CEM_FORM=$2 python tools/bench_cem.py 2>&1 | grep fused
done
