#!/bin/bash
# wgrad (tile, batch splits) sweep for one product; SHAPE=m,k,n as in gemm_bench.py; PLANS overrides the list
set -e
export SHAPE=${SHAPE:-131072,784,128}
PLANS=${PLANS:-"default 128,73 128,110 128,146 6428,59 6428,88 6428,118 6428,177 6428,236 64,78 64,156"}
for plan in $PLANS; do
  echo "== DTF_DW_PLAN=$plan"
  if [ "$plan" = "default" ]; then unset DTF_DW_PLAN; else export DTF_DW_PLAN=$plan; fi
  timeout -k 10 120 python bench/gemm_bench.py | grep "^dw"
done
