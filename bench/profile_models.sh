#!/bin/bash
# rocprofv3 kernel trace + statistics of the benchmark step for the headline MLP (default batch), the CNN and the
# widened MLP on the bf16-operand kernels; summaries land in gpurun_out/prof_models/ (see profile_one.sh).
set -e
bash bench/profile_one.sh mnist_mlp_b131072 12 --steps 20 --warmup 5 --graph 0
bash bench/profile_one.sh mnist_cnn_b1024 60 --model mnist_cnn --steps 20 --warmup 5 --graph 0
bash bench/profile_one.sh wide_mlp_bf16 24 --model wide_mlp --matmul bf16 --steps 5 --warmup 2 --graph 0
