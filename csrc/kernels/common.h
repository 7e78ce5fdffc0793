// Shared device helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DTF_API extern "C" __attribute__((visibility("default")))

// Activation codes shared with the Python side (ops/_native.py).
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2,
       ACT_TANH = 3 };   // tanh: stand-alone activation / activation-gradient kernels only (the GEMM epilogues fuse relu and sigmoid)

#define DTF_CHECK_LAUNCH()                               \
    do {                                                 \
        hipError_t e__ = hipGetLastError();              \
        if (e__ != hipSuccess) return (int)e__;          \
    } while (0)

// 32-bit finaliser with full avalanche ("lowbias32").
__device__ __forceinline__ uint32_t dtf_mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

// Per-launch dropout key: one value per (layer seed, training step).  The step
// lives in device memory so a captured hipGraph draws a fresh mask on every replay.
__device__ __forceinline__ uint32_t dtf_dropout_key(uint32_t seed, const long long* step_ptr) {
    unsigned long long s = step_ptr ? (unsigned long long)(*step_ptr) : 0ull;
    uint32_t k = dtf_mix32((uint32_t)s + 0x9E3779B9u);
    k = dtf_mix32(k ^ ((uint32_t)(s >> 32) * 0x85ebca6bU) ^ seed);
    return k;
}

// Counter-based uniform in [0,1): a pure function of (key, element index), so the
// backward pass can regenerate a mask instead of storing it.  One avalanche round over
// index * golden-ratio + key (the key is itself a two-round mix of seed and step): 32-bit
// integer multiplies issue at quarter rate, and a layer's dropout decisions are made inside
// GEMM epilogues where vector issue slots are what limits the kernel, so the hash is kept to three.
__device__ __forceinline__ float dtf_uniform(uint32_t key, unsigned long long idx) {
    const uint32_t hi = (uint32_t)(idx >> 32);
    const uint32_t r = dtf_mix32(((uint32_t)idx * 0x9E3779B1u + key) ^ ((hi << 16) | (hi >> 16)));
    return (float)(r >> 8) * (1.0f / 16777216.0f);
}

// The dropout decision "dtf_uniform(key, idx) < keep" without leaving the integers: (float)(r >> 8) * 2^-24 < keep
// <=> (r >> 8) < ceil(keep * 2^24) <=> r < ceil(keep * 2^24) << 8 (keep * 2^24 is exact in fp32).  Bit-identical
// masks for half the vector instructions; valid for keep < 1.
__device__ __forceinline__ uint32_t dtf_keep_threshold(float keep) {
    return ((uint32_t)ceilf(keep * 16777216.0f)) << 8;
}
// element indices below 2^32 (the high word of dtf_uniform's index is zero and drops out)
__device__ __forceinline__ bool dtf_keep32(uint32_t key, uint32_t idx, uint32_t threshold) {
    return dtf_mix32(idx * 0x9E3779B1u + key) < threshold;
}
__device__ __forceinline__ bool dtf_keep64(uint32_t key, unsigned long long idx, uint32_t threshold) {
    const uint32_t hi = (uint32_t)(idx >> 32);
    return dtf_mix32(((uint32_t)idx * 0x9E3779B1u + key) ^ ((hi << 16) | (hi >> 16))) < threshold;
}

__device__ __forceinline__ float dtf_sigmoid(float z) { return 1.0f / (1.0f + __expf(-z)); }

__device__ __forceinline__ float dtf_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
