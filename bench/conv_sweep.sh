# tuning sweep: second-convolution wgrad of the MNIST CNN under split-plan variants
set -e
r() { env LABEL="$1" ONLY=wgrad $1 timeout -k 10 100 python bench/conv_gemm_bench.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/conv_sweep.txt; }
: > gpurun_out/conv_sweep.txt
for sp in 49 64 78 90 96 97 98 99 100 104 112 128 147 196 256 392; do r "DTF_DW_PLAN=64,$sp"; done
cat gpurun_out/conv_sweep.txt
