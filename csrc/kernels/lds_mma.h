// Building blocks for kernels that keep both GEMM operands in LDS (the one-launch step kernel and the fused
// network-tail kernel): register-staged global->LDS block copies and a 32x32 MFMA tile product on LDS operands.
#pragma once
#include "common.h"

typedef unsigned int dtf_u32x4 __attribute__((vector_size(16)));

// A [ROWS x COLS] block of a row-major matrix is ROWS*COLS/4 16-byte vectors; thread `tid` of NT owns vectors tid,
// tid+NT, ...  Loads go through a buffer resource of `bytes` bytes: reads past it return zero, which pads a ragged
// batch without a branch.  issue() only requests the data; commit() writes the LDS image [ROWS][COLS+4] - so a
// block requested before a compute phase lands under it.
template <int NT, int ROWS, int COLS>
struct LdsStage {
    static constexpr int NV = ROWS * COLS / 4 / NT;
    static_assert(NV * NT * 4 == ROWS * COLS, "block must split evenly over the workgroup");
    f32x4 v[NV];

    // `ld`: row stride of the source in floats; SC1: L2-served load (for data another launch phase rewrote)
    template <bool SC1>
    __device__ __forceinline__ void issue(const float* base, unsigned bytes, int ld, int tid) {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + NT * i;
            const int row = idx / (COLS / 4), c = (idx % (COLS / 4)) * 4;
            // whole-vector bit_cast only: indexing the builtin's result element-wise makes hipcc (ROCm 7.2) emit
            // a single-dword load
            const dtf_u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(r, (row * ld + c) * 4, 0, SC1 ? 16 : 0);
            v[i] = __builtin_bit_cast(f32x4, raw);
        }
    }
    __device__ __forceinline__ void commit(float* dst, int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + NT * i;
            const int row = idx / (COLS / 4), c = (idx % (COLS / 4)) * 4;
            *reinterpret_cast<f32x4*>(dst + row * (COLS + 4) + c) = v[i];
        }
    }
};

// One 32x32 output tile: acc += sum_k A(m,k) B(k,n), both operands in LDS, on v_mfma_f32_32x32x2_f32.
//   A_KC: A[m][k] row-major (k contiguous)   else A[k][m]
//   B_KC: B[n][k] row-major (k contiguous)   else B[k][n]
// Lane (li = lane & 31, h = lane >> 5) feeds row/column li; within a group of 8 k-values half h takes k = 4h..4h+3,
// so a k-contiguous operand is one 16-byte LDS read per group (row strides with stride/4 odd are conflict-free).
// The fragments of k-group g+1 are requested after the first MFMA of group g and before its other three (two
// register sets; the scheduling fences keep the compiler from moving the reads next to their consumers): a read
// issued right before the MFMA that needs it idles the matrix pipe for the LDS latency in every group.
//
// `filler(g)` is independent vector-ALU work of about VALU_PER_GROUP instructions (for example the dropout hash of
// output element g): it is spread between the MFMAs of k-group g, where the wave would otherwise only wait for the
// matrix pipe, instead of running as a separate pass after the product.
template <bool A_KC, bool B_KC, int K, int VALU_PER_GROUP, class Filler>
__device__ __forceinline__ void lds_tile_mma_fill(f32x16& acc, const float* A, int lda, int m0, const float* B, int ldb,
                                                  int n0, int lane, Filler filler) {
    static_assert(K % 8 == 0, "k extent is a multiple of the 8-deep fragment group");
    const int li = lane & 31, h = lane >> 5;
    const float* pa = A_KC ? A + (m0 + li) * lda + 4 * h : A + (4 * h) * lda + m0 + li;
    const float* pb = B_KC ? B + (n0 + li) * ldb + 4 * h : B + (4 * h) * ldb + n0 + li;
    auto read = [&](int kk, f32x4& a, f32x4& b) {
        if (A_KC) {
            a = *reinterpret_cast<const f32x4*>(pa + kk);
        } else {
            const float* q = pa + kk * lda;
            a[0] = q[0]; a[1] = q[lda]; a[2] = q[2 * lda]; a[3] = q[3 * lda];
        }
        if (B_KC) {
            b = *reinterpret_cast<const f32x4*>(pb + kk);
        } else {
            const float* q = pb + kk * ldb;
            b[0] = q[0]; b[1] = q[ldb]; b[2] = q[2 * ldb]; b[3] = q[3 * ldb];
        }
    };
    f32x4 a[2], b[2];
    read(0, a[0], b[0]);
    constexpr int NREADS = (A_KC ? 1 : 4) + (B_KC ? 1 : 4);      // LDS read instructions per k-group
#pragma unroll
    for (int g = 0; g < K / 8; ++g) {
        const int cur = g & 1;
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][0], b[cur][0], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < K / 8) read((g + 1) * 8, a[cur ^ 1], b[cur ^ 1]);
        filler(g);
#pragma unroll
        for (int t = 1; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][t], b[cur][t], acc, 0, 0, 0);
        // The next group's reads are dealt out over the gaps behind MFMAs 0, 1 and 2 (a burst of five to eight LDS
        // reads in one gap costs the wave ~50 cycles of issue before its next MFMA; one or two per gap are free), the
        // filler's vector instructions likewise.
#pragma unroll
        for (int t = 1; t < 4; ++t) {
            if (g + 1 < K / 8) __builtin_amdgcn_sched_group_barrier(0x100, (NREADS + 2) / 3, 0);
            if (VALU_PER_GROUP > 0) __builtin_amdgcn_sched_group_barrier(0x002, (VALU_PER_GROUP + 2) / 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <bool A_KC, bool B_KC, int K>
__device__ __forceinline__ void lds_tile_mma(f32x16& acc, const float* A, int lda, int m0, const float* B, int ldb, int n0,
                                             int lane) {
    lds_tile_mma_fill<A_KC, B_KC, K, 0>(acc, A, lda, m0, B, ldb, n0, lane, [](int) {});
}

// One 16x16 output tile on v_mfma_f32_16x16x4_f32 (half the cycles of a 32x32 tile for outputs that are only 16 wide:
// class logits, their weight gradient).  Lane (i = lane & 15, q = lane >> 4) feeds row / column i; within a group of 16
// k-values quarter q takes k = 4q .. 4q+3, one per MFMA step, so a k-contiguous operand is one 16-byte read per group.
// C/D: column lane & 15, rows 4 * (lane >> 4) + r for register r = 0..3.
template <bool A_KC, bool B_KC, int K>
__device__ __forceinline__ void lds_tile16_mma(f32x4& acc, const float* A, int lda, int m0, const float* B, int ldb, int n0,
                                               int lane) {
    static_assert(K % 16 == 0, "k extent is a multiple of the 16-deep fragment group");
    const int i = lane & 15, q = lane >> 4;
    const float* pa = A_KC ? A + (m0 + i) * lda + 4 * q : A + (4 * q) * lda + m0 + i;
    const float* pb = B_KC ? B + (n0 + i) * ldb + 4 * q : B + (4 * q) * ldb + n0 + i;
    auto read = [&](int kk, f32x4& a, f32x4& b) {
        if (A_KC) {
            a = *reinterpret_cast<const f32x4*>(pa + kk);
        } else {
            const float* p = pa + kk * lda;
            a[0] = p[0]; a[1] = p[lda]; a[2] = p[2 * lda]; a[3] = p[3 * lda];
        }
        if (B_KC) {
            b = *reinterpret_cast<const f32x4*>(pb + kk);
        } else {
            const float* p = pb + kk * ldb;
            b[0] = p[0]; b[1] = p[ldb]; b[2] = p[2 * ldb]; b[3] = p[3 * ldb];
        }
    };
    f32x4 a[2], b[2];
    read(0, a[0], b[0]);
#pragma unroll
    for (int g = 0; g < K / 16; ++g) {
        const int cur = g & 1;
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][0], b[cur][0], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < K / 16) read((g + 1) * 16, a[cur ^ 1], b[cur ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 1; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][t], b[cur][t], acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void lds_zero(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// row of accumulator register r inside a 32x32 tile for lane half h (column = lane & 31)
__device__ __forceinline__ int lds_tile_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
