// Memory-bound elementwise kernels: activation gradient, stand-alone dropout, the
// device-side XOR data generator and small utilities.  16 bytes per lane, grid-stride.
#include "common.h"

namespace {

inline int ew_grid(long long n4) {
    long long g = (n4 + 255) / 256;
    if (g > 2048) g = 2048;
    return g < 1 ? 1 : (int)g;
}

// dZ = dY * act'(Y).  relu: (Y > 0) * scale (scale carries a fused dropout's 1/keep);
// sigmoid: Y (1 - Y); tanh: 1 - Y^2; none: copy * scale.
__global__ __launch_bounds__(256) void act_grad_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                                       float* __restrict__ dZ, long long n, int act, float scale) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float g = dY[i], y = Y[i];
        float r;
        if (act == ACT_RELU) r = (y > 0.f) ? g * scale : 0.f;
        else if (act == ACT_SIGMOID) r = g * y * (1.0f - y);
        else if (act == ACT_TANH) r = g * (1.0f - y * y);
        else r = g * scale;
        dZ[i] = r;
    }
}

// Inverted dropout (Keras Dropout at /root/reference/example.py:151,153): keep with
// probability `keep`, scale kept values by 1/keep.  The mask is a pure function of
// (seed, step, index); the backward pass calls the same kernel on the gradient.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ X, float* __restrict__ Y,
                                                      long long n, float keep, uint32_t seed,
                                                      const long long* step_ptr) {
    const uint32_t key = dtf_dropout_key(seed, step_ptr);
    const float inv = 1.0f / keep;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float u = dtf_uniform(key, (unsigned long long)i);
        Y[i] = (u < keep) ? X[i] * inv : 0.f;
    }
}

// Synthetic task of /root/reference/example.py:24-48: 2*bits random input bits per
// sample, label bit k = x[k] XOR x[k + bits].  One thread per sample.
__global__ __launch_bounds__(256) void xor_data_kernel(float* __restrict__ X, float* __restrict__ Y, int n,
                                                       int bits, uint32_t seed) {
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        float* x = X + (long long)s * 2 * bits;
        float* y = Y + (long long)s * bits;
        for (int k = 0; k < bits; ++k) {
            const uint32_t r = dtf_mix32(dtf_mix32((uint32_t)s * 0x9E3779B1u + seed) + (uint32_t)k * 0x85ebca6bU);
            const int a = (r >> 9) & 1, b = (r >> 21) & 1;
            x[k] = (float)a;
            x[k + bits] = (float)b;
            y[k] = (float)(a ^ b);
        }
    }
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long long n, float s) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] *= s;
}

}  // namespace

DTF_API int dtf_act_grad(const float* dY, const float* Y, float* dZ, long long n, int act, float scale,
                         hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(act_grad_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, dY, Y, dZ, n, act, scale);
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_dropout(const float* X, float* Y, long long n, float keep, uint32_t seed,
                        const long long* step_ptr, hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(dropout_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, X, Y, n, keep, seed, step_ptr);
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_xor_data(float* X, float* Y, int n, int bits, uint32_t seed, hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(xor_data_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, X, Y, n, bits, seed);
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_scale(float* x, long long n, float s, hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(scale_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, x, n, s);
    DTF_CHECK_LAUNCH();
    return 0;
}

namespace {
__global__ __launch_bounds__(256) void activation_kernel(const float* __restrict__ X, float* __restrict__ Y,
                                                         long long n, int act) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float x = X[i];
        Y[i] = (act == ACT_RELU) ? fmaxf(x, 0.f) : (act == ACT_SIGMOID ? dtf_sigmoid(x) : (act == ACT_TANH ? tanhf(x) : x));
    }
}
}  // namespace

DTF_API int dtf_activation(const float* X, float* Y, long long n, int act, hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(activation_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, X, Y, n, act);
    DTF_CHECK_LAUNCH();
    return 0;
}

// Variable initialisers on the device: x[i] = lo + (hi - lo) * u(seed, i) with u a counter-based uniform in [0, 1)
// (two avalanche rounds over the 64-bit element index and the seed), so a billion-parameter model is initialised
// at HBM speed instead of drawing on the host and copying (Keras' glorot_uniform / he_uniform behind the layers of
// /root/reference/example.py:150-154 are uniform distributions; only their bounds differ).
__global__ __launch_bounds__(256) void init_uniform_kernel(float* __restrict__ x, long long n, uint32_t seed, float lo, float hi) {
#pragma clang fp contract(off)      // multiply and add round separately: the same bits as the CPU implementation
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float span = hi - lo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t a = dtf_mix32((uint32_t)i * 0x9E3779B1u + seed);
        const uint32_t r = dtf_mix32(a ^ ((uint32_t)((unsigned long long)i >> 32) * 0x85EBCA6Bu + 0x27D4EB2Fu));
        const float u = (float)(r >> 8) * (1.0f / 16777216.0f);
        const float su = span * u;
        x[i] = lo + su;
    }
}

DTF_API int dtf_init_uniform(float* x, long long n, uint32_t seed, float lo, float hi, hipStream_t stream) {
    if (n <= 0) return 0;
    long long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(init_uniform_kernel, dim3((unsigned)g), dim3(256), 0, stream, x, n, seed, lo, hi);
    DTF_CHECK_LAUNCH();
    return 0;
}
