// Pure v_mfma_f32_32x32x2_f32 issue-rate probe: register operands only, random (non-zero) data.
// Prints achieved TFLOP/s and the in-kernel shader clock (s_memtime ticks per s_memrealtime tick).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void probe(const float* in, float* out, int iters, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[lane + 64 * i]; b[i] = in[256 + lane + 64 * i]; }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int n = 0; n < NACC; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[(t + n) & 3], acc[n], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
void run(int blocks, int iters, const float* din, float* dout, unsigned long long* dclk) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(256), 0, 0, din, dout, iters, dclk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(256), 0, 0, din, dout, iters, dclk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> clk(2 * blocks);
    hipMemcpy(clk.data(), dclk, clk.size() * 8, hipMemcpyDeviceToHost);
    double ghz = (double)clk[0] / (double)clk[1] * 0.1;
    double flop = (double)blocks * 4 * iters * 4 * NACC * 32.0 * 32 * 2 * 2;
    printf("blocks=%d (waves/SIMD=%d) nacc=%d  %.3f ms  %.1f TF  clock %.2f GHz  cycles/MFMA/SIMD %.1f\n", blocks, blocks / 256,
           NACC, ms, flop / ms / 1e9, ghz, (double)clk[0] / ((double)iters * 4 * NACC) / (blocks / 256));
}

int main() {
    float* din; float* dout; unsigned long long* dclk;
    std::vector<float> h(512);
    for (auto& x : h) x = (float)rand() / RAND_MAX * 2 - 1;
    hipMalloc(&din, 512 * 4); hipMalloc(&dout, 2048 * 256 * 4); hipMalloc(&dclk, 2048 * 16);
    hipMemcpy(din, h.data(), 512 * 4, hipMemcpyHostToDevice);
    run<1>(256, 20000, din, dout, dclk);
    run<2>(256, 10000, din, dout, dclk);
    run<4>(256, 5000, din, dout, dclk);
    run<2>(512, 10000, din, dout, dclk);
    run<4>(512, 5000, din, dout, dclk);
    run<2>(1024, 5000, din, dout, dclk);
    return 0;
}
