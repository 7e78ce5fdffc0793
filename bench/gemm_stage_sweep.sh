#!/bin/bash
# Sweep of the GEMM tuning knobs (LDS stages, forced tile, LDS padding = blocks per CU) on the first-layer shapes.
set -e
export SHAPE=${SHAPE:-65536,784,128}
for cfg in "0 0 0" "16 0 0" "16 0 40" "16 0 32" "16 0 52" "0 128 0" "16 128 0" "16 128 52" "16 128 40" "0 6428 80" "0 64 0" "16 64 0"; do
  set -- $cfg
  echo "== DBG=$1 TILE=$2 LDS_KB=$3"
  if [ "$3" = "0" ]; then unset DTF_GEMM_LDS_KB; else export DTF_GEMM_LDS_KB=$3; fi
  DTF_GEMM_DBG=$1 DTF_GEMM_TILE=$2 timeout -k 10 120 python bench/gemm_bench.py
done
