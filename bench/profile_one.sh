#!/bin/bash
# rocprofv3 kernel trace of one bench.py configuration: bash bench/profile_one.sh <name> <timeline rows> <bench flags...>
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_models
mkdir -p $OUT
export TMPDIR=/tmp
name=$1; rows=$2; shift 2
cd /tmp
timeout -k 10 170 rocprofv3 --kernel-trace --stats -d $OUT/$name -o $name --output-format csv -- python3 $ROOT/bench.py "$@" > $OUT/$name.log 2>&1
cd $ROOT
trace=$(ls $OUT/$name/*kernel_trace.csv $OUT/$name/*/*kernel_trace.csv 2>/dev/null | head -1)
stats=$(ls $OUT/$name/*kernel_stats.csv $OUT/$name/*/*kernel_stats.csv 2>/dev/null | head -1)
python3 bench/trace_summary.py $trace $rows > $OUT/${name}_timeline.txt
head -25 $stats > $OUT/${name}_kernel_stats.csv
rm -rf $OUT/$name
echo "done $name"
