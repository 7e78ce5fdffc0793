#!/bin/bash
# One rocprofv3 counter pass (8 SQ + 1 GRBM counters) over a short bench.py run: bash bench/pmc_one.sh <name> <rows> <bench flags...>
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT
export TMPDIR=/tmp
name=$1; rows=$2; shift 2
cd /tmp
timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT/$name -o $name --output-format csv -- python3 $ROOT/bench.py "$@" > $OUT/$name.log 2>&1
cd $ROOT
csv=$(ls $OUT/$name/*counter_collection.csv $OUT/$name/*/*counter_collection.csv 2>/dev/null | head -1)
python3 bench/pmc_summary.py $csv $rows > $OUT/${name}_pmc.txt
rm -rf $OUT/$name
echo "done $name"
