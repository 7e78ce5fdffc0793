#!/bin/bash
# wgrad (tile, batch splits) sweep for one product; SHAPE=m,k,n as in gemm_bench.py
set -e
export SHAPE=${SHAPE:-65536,784,128}
for plan in "" 128,73 128,36 128,18 6428,59 6428,39 6428,29 6428,19 64,78 64,39 64,19; do
  echo "== DTF_DW_PLAN=$plan"
  if [ -z "$plan" ]; then unset DTF_DW_PLAN; else export DTF_DW_PLAN=$plan; fi
  timeout -k 10 120 python bench/gemm_bench.py | grep "^dw"
done
