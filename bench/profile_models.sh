#!/bin/bash
# rocprofv3 kernel trace + statistics of the benchmark step for the headline MLP (default batch), the CNN and the
# widened MLP on the bf16-operand kernels; summaries land in gpurun_out/prof_models/.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_models
mkdir -p $OUT
export TMPDIR=/tmp
prof() {   # name, rows of the timeline, bench flags...
    name=$1; rows=$2; shift 2
    cd /tmp
    timeout -k 10 170 rocprofv3 --kernel-trace --stats -d $OUT/$name -o $name --output-format csv -- python3 $ROOT/bench.py "$@" > $OUT/$name.log 2>&1
    cd $ROOT
    trace=$(ls $OUT/$name/*kernel_trace.csv $OUT/$name/*/*kernel_trace.csv 2>/dev/null | head -1)
    stats=$(ls $OUT/$name/*kernel_stats.csv $OUT/$name/*/*kernel_stats.csv 2>/dev/null | head -1)
    python3 bench/trace_summary.py $trace $rows > $OUT/${name}_timeline.txt
    head -25 $stats > $OUT/${name}_kernel_stats.csv
    tail -1 $OUT/$name.log
    rm -rf $OUT/$name
    echo "done $name"
}
prof mnist_mlp_b131072 12 --steps 20 --warmup 5 --graph 0
prof mnist_cnn_b1024 60 --model mnist_cnn --steps 20 --warmup 5 --graph 0
prof wide_mlp_bf16 24 --model wide_mlp --matmul bf16 --steps 5 --warmup 2 --graph 0
