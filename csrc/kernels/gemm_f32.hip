// Exact-fp32 dense-layer GEMMs for gfx950 on v_mfma_f32_32x32x2_f32 (wave64 MFMA).
//
// One templated LDS-tiled kernel serves the three products a dense layer needs
// (what TF runs as MatMul / BiasAdd / Relu / Sigmoid / dropout and their gradients
// behind /root/reference/example.py:150-154 and :170):
//
//   forward   Y[M,N]  = dropout(act(X[M,K] . W[K,N] + b))          A k-contig, B n-contig
//   dgrad     dX[M,K] = (dZ[M,N] . W[K,N]^T) * act'(Yprev)         A k-contig, B k-contig
//   wgrad     dW[K,N] = X[M,K]^T . dZ[M,N],  db[N] = colsum(dZ)     A m-contig, B n-contig
//
// Block = 256 threads = 4 waves; each wave owns a TMxTN grid of 32x32 MFMA tiles.
// K is consumed in 32-deep tiles through a double-buffered LDS image: the global loads of
// tile t+1 are issued before the MFMAs of tile t, land in registers meanwhile, and are
// written to the other LDS buffer afterwards, so one barrier per tile suffices.
//
// Global loads are raw buffer loads: an out-of-range lane gets an out-of-range offset and
// the hardware returns zeros, so tails need no branches (a branchy loader made hipcc wait
// for vmcnt(0) in front of the MFMAs).
//
// Inside each group of 8 k-values the lane half h = lane>>5 takes k = 8*kk + 4*h + t at
// MFMA step t: a GEMM sums over k in any order, and this order lets a k-contiguous operand
// be fetched with one 16-byte ds_read_b128 per four MFMAs.  Row stride of k-contiguous LDS
// tiles is 36 floats (144 B = 9 x 16 B, an odd multiple), which makes those reads
// conflict-free.
#include "common.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;  // row stride (floats) of a k-contiguous LDS tile
constexpr int LDKB = BK + 8; // row stride (bf16 elements, 80 bytes) of a bf16 LDS tile: 16-byte fragment reads of 8
                             // consecutive rows fall into distinct bank groups

// Element offset of (row R, k) in a bf16 LDS image [rows][LDKB].  The four 16-byte chunks of a row are XOR-swizzled
// with bits 4-5 of the row: fragment reads (16 consecutive rows, one chunk) stay conflict-free, and the transposing
// loader's 8-byte stores, whose 32 lanes hit rows 4 apart (bank stride 16 of 64), spread over all 16 chunk slots
// (2-way instead of 8-way conflicts).
__device__ __forceinline__ int bf16_lds_off(int R, int k) {
    return R * LDKB + ((((k >> 3) ^ (R >> 4)) & 3) << 3) + (k & 7);
}

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { EPI_FWD = 0, EPI_DX = 1, EPI_DW = 2 };


struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    int lda, ldb, ldc;
    int M, N, K;
    int k_chunk;              // K range handled by one blockIdx.z (multiple of BK)
    long long c_split_stride; // elements between successive split slabs of C
    // forward epilogue
    const float* bias;
    int act;
    float keep;               // dropout keep probability; >= 1 disables dropout
    uint32_t seed;
    const long long* step_ptr;
    unsigned char* pool_arg;  // forward, rows ordered four-per-2x2-pooling-window (see conv_rowtab_masked_kernel): C is the
                              // POOLED output [M/4, N] and pool_arg[M/4, N] receives the arg-max position 0..3 of every
                              // window; the full-resolution activation is never written.  Null: plain forward.
    // dgrad epilogue
    const float* yprev;       // output of the producing layer (post activation/dropout)
    int ldy;
    float scale_prev;         // 1/keep of the producing layer's dropout (1 if none)
    // wgrad extras
    float* colsum;            // partial bias gradients of split 0 (may be null)
    long long colsum_split_stride;  // elements between successive splits' colsum rows
    int colsum_mfma;          // the column sums ride on the matrix cores: the loader sets the first padding row of A (row M,
                              // inside the last, partly filled 32-row sub-tile) to ones, so row M of the product IS
                              // colsum(B) and the epilogue routes it to `colsum`.  Needs M % 32 != 0.  Otherwise the block
                              // of row tile 0 sums the B tile's columns from LDS (15 % longer than its neighbours).
    int single_lds;           // one LDS stage instead of two (tuning aid DTF_GEMM_SINGLE): half the LDS per block, so twice
                              // the resident blocks, for a second barrier per k tile
    int dbg;                  // DTF_GEMM_DBG bit 8: every block stamps its main-loop time over its first outputs; 0 in production
    int xcd_remap;            // walk the grid so that the blocks one XCD receives (every eighth in dispatch order) are
                              // NEIGHBOURS in the grid: they share operand panels, and an XCD has its own L2
    // implicit-GEMM convolution (GATHER kernels): operand A is the patch matrix of a zero-padded NHWC image that is
    // never materialised.  Patch row m starts at element g_rowtab[m] of the image; window position w = k / g_C
    // adds g_shift[w]; the channel k % g_C is contiguous.
    const int* g_rowtab;      // [g_rows]
    const unsigned* g_masktab; // [g_rows] or null.  Null: the image is zero-padded and every tap of every row is inside it.
                              // Otherwise the image is NOT padded: bit w of g_masktab[m] tells whether window position w
                              // of patch row m lies inside the image (taps outside read as zero), and g_rowtab[m] - the
                              // origin of the window - may be negative for rows at the image border.
    const int* g_shift;       // [KH*KW]
    int g_rows;
    int g_C;                  // channels per window position (multiple of 4; multiples of 32 take the uniform fast path)
    long long g_elems;        // elements of the padded image
};

typedef unsigned int u32x4 __attribute__((vector_size(16)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, long long bytes) {
    const unsigned n = bytes <= 0 ? 0u : (bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0u : (unsigned)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)n, 0x00020000);
}

constexpr unsigned OOB = 0xFFFFFFF8u;  // >= any num_records: the load returns zeros

// Exact split of an fp32 value into three bf16 limbs, x = hi + mid + lo: each limb is the bf16 rounding of what the
// previous limbs left over, and both subtractions are exact in fp32 (3 x 8 significant bits cover the 24 of fp32).
__device__ __forceinline__ void split3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = (__bf16)x;
    float r = x - (float)hi;
    mid = (__bf16)r;
    r -= (float)mid;
    lo = (__bf16)r;
}

// Tile of ROWS (M- or N-dim) x BK (k-dim) held by 256 threads as NV float4 each.
//   KC  (k is the contiguous global dim): LDS image [ROWS][LDK]
//   !KC (the M/N index is contiguous)   : LDS image [BK][ROWS+4]
template <int ROWS, bool KC>
struct TileLoader {
    static constexpr int NV = ROWS * BK / 4 / 256;
    static constexpr int LDS_FLOATS = KC ? ROWS * LDK : BK * (ROWS + 4);

    __amdgpu_buffer_rsrc_t rsrc;
    int ld;
    int mn_left;   // valid rows/cols of the M/N dimension from this block's origin
    unsigned voff[NV];   // byte offset of this thread's vectors inside the first k tile; OOB if its row/column is outside
    f32x4 v[NV];
    bool ones_row = false;   // M/N-contiguous images only: the first index past the operand (mn_left) is stored as 1.0

    // base points at element (mn0, kbeg) of the operand; rows_total/k_total bound the allocation
    __device__ __forceinline__ void init(const float* base, int ld_, int mn0, int mn_dim, int kbeg, int k_total) {
        ld = ld_;
        mn_left = mn_dim - mn0;
        if (KC) {
            // rows = mn index: base + mn0*ld ; buffer ends after row mn_dim-1
            rsrc = make_rsrc(base + (long long)mn0 * ld, ((long long)mn_left * ld) * 4);
        } else {
            // rows = k index: base + kbeg*ld + mn0 ; buffer ends after row k_total-1
            rsrc = make_rsrc(base + (long long)kbeg * ld + mn0, (((long long)(k_total - kbeg)) * ld - mn0) * 4);
        }
        const int tid = threadIdx.x;
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const int id = tid + 256 * s;
            if (KC) {
                const int r = id >> 3, c = (id & 7) * 4;
                voff[s] = (r < mn_left) ? ((unsigned)r * (unsigned)ld + (unsigned)c) * 4u : OOB;
            } else {
                const int c = id / (ROWS / 4), r = (id % (ROWS / 4)) * 4;
                voff[s] = (r < mn_left) ? ((unsigned)c * (unsigned)ld + (unsigned)r) * 4u : OOB;
            }
        }
    }

    // Issue the tile's loads (k0: absolute k of the tile start; k_left = kend - k0).  Nothing here
    // consumes the data, so the wait lands in store(), after the MFMAs of the previous tile.
    // A full tile costs one add and one select per vector: the per-thread part of the address is fixed, the tile
    // adds a uniform offset, and rows / columns outside the operand were given the out-of-range offset once.
    __device__ __forceinline__ void load(int k0, int kbeg, int k_left, int tid) {
        const unsigned koff = KC ? (unsigned)k0 * 4u : (unsigned)(k0 - kbeg) * (unsigned)ld * 4u;
        const bool full = k_left >= BK;
        if (full) {
            // whole tile: the uniform tile offset rides in the instruction's scalar offset, the per-thread offset
            // (already out of range for rows / columns outside the operand) is used as it is - no vector arithmetic
#pragma unroll
            for (int s = 0; s < NV; ++s) {
                const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff[s], (int)koff, 0);
                v[s] = __builtin_bit_cast(f32x4, raw);
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const int id = tid + 256 * s;
            const int c = KC ? (id & 7) * 4 : id / (ROWS / 4);       // k index inside the tile
            const bool ok = (voff[s] != OOB) && (full || c < k_left);
            // whole-vector bit_cast only: indexing the builtin's result element-wise makes hipcc
            // (ROCm 7.2) emit a single-dword load
            const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(ok ? voff[s] + koff : OOB), 0, 0);
            v[s] = __builtin_bit_cast(f32x4, raw);
        }
    }

    // Write the LDS image.  Only a vector that straddles the end of K (k-contiguous tiles) or of the M/N range
    // needs its tail elements zeroed; that is decided per tile with a uniform test.
    __device__ __forceinline__ void store(float* lds, int k_left, int tid) const {
        const bool plain = KC ? (k_left >= BK) : (mn_left >= ROWS);
        if (plain) {
#pragma unroll
            for (int s = 0; s < NV; ++s) {
                const int id = tid + 256 * s;
                const int off = KC ? (id >> 3) * LDK + (id & 7) * 4
                                   : (id / (ROWS / 4)) * (ROWS + 4) + (id % (ROWS / 4)) * 4;
                *reinterpret_cast<f32x4*>(lds + off) = v[s];
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const int id = tid + 256 * s;
            int off, lim;
            if (KC) {
                off = (id >> 3) * LDK + (id & 7) * 4;
                lim = k_left - (id & 7) * 4;
            } else {
                off = (id / (ROWS / 4)) * (ROWS + 4) + (id % (ROWS / 4)) * 4;
                lim = mn_left - (id % (ROWS / 4)) * 4;
            }
            f32x4 x = v[s];
#pragma unroll
            for (int j = 1; j < 4; ++j) x[j] = (j < lim) ? x[j] : 0.f;
            if (!KC && ones_row) {
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = (j == lim) ? 1.0f : x[j];
            }
            *reinterpret_cast<f32x4*>(lds + off) = x;
        }
    }

    // bf16 image for the bf16-operand kernel, ALWAYS k-contiguous: [ROWS][LDKB] of bf16 (round to nearest even).
    // A k-contiguous source writes its 4 values as one 8-byte store; an M/N-contiguous source is transposed on the
    // way in (4 two-byte stores to 4 consecutive rows).
    __device__ __forceinline__ void store_bf16(unsigned short* lds, int k_left, int tid) const {
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const int id = tid + 256 * s;
            f32x4 x = v[s];
            if (KC) {
                const int r = id >> 3, c = (id & 7) * 4;
                const int lim = k_left - c;
#pragma unroll
                for (int j = 1; j < 4; ++j) x[j] = (j < lim) ? x[j] : 0.f;
                bf16x4 y;
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = (__bf16)x[j];
                *reinterpret_cast<bf16x4*>(lds + bf16_lds_off(r, c)) = y;
            } else {
                const int c = id / (ROWS / 4), r = (id % (ROWS / 4)) * 4;      // c = k index, r = first of 4 rows
                const int lim = mn_left - r;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const __bf16 y = (__bf16)((j == 0 || j < lim) ? x[j] : 0.f);
                    lds[bf16_lds_off(r + j, c)] = __builtin_bit_cast(unsigned short, y);
                }
            }
        }
    }

    // Three bf16 limb planes (hi | mid | lo, `plane` elements apart) of a k-contiguous tile whose K is a multiple of 4.
    __device__ __forceinline__ void store_split(unsigned short* lds, int plane, int tid) const {
        static_assert(KC, "M/N-contiguous operands go through TransposeLoader");
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const int id = tid + 256 * s;
            bf16x4 hi, mid, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __bf16 a, b, c;
                split3(v[s][j], a, b, c);
                hi[j] = a; mid[j] = b; lo[j] = c;
            }
            unsigned short* d = lds + bf16_lds_off(id >> 3, (id & 7) * 4);
            *reinterpret_cast<bf16x4*>(d) = hi;
            *reinterpret_cast<bf16x4*>(d + plane) = mid;
            *reinterpret_cast<bf16x4*>(d + 2 * plane) = lo;
        }
    }
};

// Loader of an M/N-contiguous operand for the bf16 kernel.  The LDS image must be k-contiguous, so the tile is
// transposed on the way in; a thread therefore owns 4 CONSECUTIVE k rows of the same 4 M/N columns (4 float4 loads)
// and writes each column's 4 k values as one 8-byte store - a quarter of the LDS stores of a per-element scatter.
template <int ROWS>
struct TransposeLoader {
    static constexpr int NV = 4;
    static_assert(ROWS == 128, "256 threads x (4 k rows x 4 columns) cover a 32 x 128 tile");
    __amdgpu_buffer_rsrc_t rsrc;
    int ld, mn_left;
    unsigned voff[NV];
    f32x4 v[NV];

    __device__ __forceinline__ void init(const float* base, int ld_, int mn0, int mn_dim, int kbeg, int k_total) {
        ld = ld_;
        mn_left = mn_dim - mn0;
        rsrc = make_rsrc(base + (long long)kbeg * ld + mn0, (((long long)(k_total - kbeg)) * ld - mn0) * 4);
        const int tid = threadIdx.x;
        const int r = (tid & 31) * 4, c0 = (tid >> 5) * 4;
#pragma unroll
        for (int s = 0; s < NV; ++s)
            voff[s] = (r < mn_left) ? ((unsigned)(c0 + s) * (unsigned)ld + (unsigned)r) * 4u : OOB;
    }
    __device__ __forceinline__ void load(int k0, int kbeg, int k_left, int tid) {
        const unsigned koff = (unsigned)(k0 - kbeg) * (unsigned)ld * 4u;
        const int c0 = (tid >> 5) * 4;
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const bool ok = (voff[s] != OOB) && (c0 + s < k_left);
            const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(ok ? voff[s] + koff : OOB), 0, 0);
            v[s] = __builtin_bit_cast(f32x4, raw);
        }
    }
    __device__ __forceinline__ void store_bf16(unsigned short* lds, int k_left, int tid) const {
        const int r = (tid & 31) * 4, c0 = (tid >> 5) * 4;
        const int lim = mn_left - r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x4 y;
#pragma unroll
            for (int s = 0; s < NV; ++s) y[s] = (__bf16)((j == 0 || j < lim) ? v[s][j] : 0.f);
            *reinterpret_cast<bf16x4*>(lds + bf16_lds_off(r + j, c0)) = y;
        }
    }
    // limb planes as in TileLoader::store_split; the M/N extent is a multiple of 4, so a loaded vector is all inside
    // the operand or all zeros
    __device__ __forceinline__ void store_split(unsigned short* lds, int plane, int tid) const {
        const int r = (tid & 31) * 4, c0 = (tid >> 5) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x4 hi, mid, lo;
#pragma unroll
            for (int s = 0; s < NV; ++s) {
                __bf16 a, b, c;
                split3(v[s][j], a, b, c);
                hi[s] = a; mid[s] = b; lo[s] = c;
            }
            unsigned short* d = lds + bf16_lds_off(r + j, c0);
            *reinterpret_cast<bf16x4*>(d) = hi;
            *reinterpret_cast<bf16x4*>(d + plane) = mid;
            *reinterpret_cast<bf16x4*>(d + 2 * plane) = lo;
        }
    }
    // column sums of the tile this thread holds (columns r .. r+3 over its 4 k rows)
    __device__ __forceinline__ void add_colsum(f32x4& cs) const {
#pragma unroll
        for (int s = 0; s < NV; ++s) cs += v[s];
    }
};

template <int ROWS, bool KC>
struct PickBf16Loader { using type = TileLoader<ROWS, true>; };
template <int ROWS>
struct PickBf16Loader<ROWS, false> { using type = TransposeLoader<ROWS>; };

// Operand-A loaders of the implicit-GEMM convolution: same LDS images and the same load()/store() protocol as
// TileLoader, but the global address of patch element (m, k) is rowtab[m] + shift[k / C] + k % C in the padded image.
template <int ROWS, bool KC>
struct GatherLoader;

// rows = patch rows m, k contiguous inside one window position (forward, and dgrad as a convolution of dz)
template <int ROWS>
struct GatherLoader<ROWS, true> {
    static constexpr int NV = ROWS * BK / 4 / 256;
    static constexpr int LDS_FLOATS = ROWS * LDK;
    __amdgpu_buffer_rsrc_t rsrc;
    const int* shift;
    int C;
    int rb[NV];          // image offset of this thread's patch rows
    unsigned taps[NV];   // bit w: window position w of the row is inside the image (0: row past M)
    f32x4 v[NV];

    __device__ __forceinline__ void init_gather(const GemmArgs& p, int mn0, int kbeg, int tid) {
        rsrc = make_rsrc(p.A, p.g_elems * 4);
        shift = p.g_shift;
        C = p.g_C;
        __amdgpu_buffer_rsrc_t tab = make_rsrc(reinterpret_cast<const float*>(p.g_rowtab), (long long)p.g_rows * 4);
        __amdgpu_buffer_rsrc_t mtab = make_rsrc(reinterpret_cast<const float*>(p.g_masktab), p.g_masktab ? (long long)p.g_rows * 4 : 0);
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const int m = mn0 + ((tid + 256 * s) >> 3);
            rb[s] = (int)__builtin_amdgcn_raw_buffer_load_b32(tab, m * 4, 0, 0);
            const unsigned mk = p.g_masktab ? __builtin_amdgcn_raw_buffer_load_b32(mtab, m * 4, 0, 0) : 0xFFFFFFFFu;
            taps[s] = (m < p.M) ? mk : 0u;
        }
    }
    __device__ __forceinline__ void load(int k0, int kbeg, int k_left, int tid) {
        // All vectors of a thread sit in the same k column c = (tid & 7) * 4 of the tile.  With C a multiple of the tile
        // depth the whole tile lies inside one window position (w and the shift are uniform); otherwise C is a multiple
        // of 4, a vector still lies inside one position, and every thread finds its own (one division per thread and tile).
        const int c = (tid & 7) * 4;
        int w, sh;
        if (C % BK == 0) {
            w = k0 / C;
            sh = shift[w] + (k0 % C) + c;
        } else {
            const int k = min(k0 + c, k0 + k_left - 1);      // past the end of K: any valid position, the load is masked
            w = k / C;
            sh = shift[w] + (k - w * C);
        }
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const bool ok = ((taps[s] >> w) & 1u) && (c < k_left);
            const unsigned off = (unsigned)(rb[s] + sh) * 4u;
            const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(ok ? off : OOB), 0, 0);
            v[s] = __builtin_bit_cast(f32x4, raw);
        }
    }
    __device__ __forceinline__ void store(float* lds, int k_left, int tid) const {
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const int id = tid + 256 * s;
            *reinterpret_cast<f32x4*>(lds + (id >> 3) * LDK + (id & 7) * 4) = v[s];    // K is a multiple of 4: vectors past it were loaded as zeros
        }
    }
};

// wgrad: GEMM rows = patch columns kin (contiguous inside a window position), GEMM k = patch rows m
template <int ROWS>
struct GatherLoader<ROWS, false> {
    static constexpr int NV = ROWS * BK / 4 / 256;
    static constexpr int LDS_FLOATS = BK * (ROWS + 4);
    __amdgpu_buffer_rsrc_t rsrc, tab, mtab;
    int ko[NV];       // image offset of this thread's patch columns relative to the row origin (-1: past the patch)
    int wpos[NV];     // window position of those columns
    int rb_next[NV];  // row origin of the NEXT tile's patch row, requested one tile ahead
    unsigned mk_next[NV];   // and which of its window positions are inside the image
    bool masked;
    f32x4 v[NV];

    __device__ __forceinline__ void init_gather(const GemmArgs& p, int mn0, int kbeg, int tid) {
        rsrc = make_rsrc(p.A, p.g_elems * 4);
        tab = make_rsrc(reinterpret_cast<const float*>(p.g_rowtab), (long long)p.g_rows * 4);
        masked = p.g_masktab != nullptr;
        mtab = make_rsrc(reinterpret_cast<const float*>(p.g_masktab), masked ? (long long)p.g_rows * 4 : 0);
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const int id = tid + 256 * s;
            const int kin = mn0 + (id % (ROWS / 4)) * 4;
            wpos[s] = (kin < p.M) ? kin / p.g_C : 0;
            ko[s] = (kin < p.M) ? p.g_shift[wpos[s]] + kin % p.g_C : -1;
            rb_next[s] = (int)__builtin_amdgcn_raw_buffer_load_b32(tab, (kbeg + id / (ROWS / 4)) * 4, 0, 0);
            mk_next[s] = masked ? __builtin_amdgcn_raw_buffer_load_b32(mtab, (kbeg + id / (ROWS / 4)) * 4, 0, 0) : 0xFFFFFFFFu;
        }
    }
    __device__ __forceinline__ void load(int k0, int kbeg, int k_left, int tid) {
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const int c = (tid + 256 * s) / (ROWS / 4);
            const bool ok = (ko[s] >= 0) && (c < k_left) && ((mk_next[s] >> wpos[s]) & 1u);
            const unsigned off = (unsigned)(rb_next[s] + ko[s]) * 4u;
            const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(ok ? off : OOB), 0, 0);
            v[s] = __builtin_bit_cast(f32x4, raw);
        }
#pragma unroll
        for (int s = 0; s < NV; ++s) {    // past the table the loads return 0 and the data load is masked by k_left
            const int row = (k0 + BK + (tid + 256 * s) / (ROWS / 4)) * 4;
            rb_next[s] = (int)__builtin_amdgcn_raw_buffer_load_b32(tab, row, 0, 0);
            if (masked) mk_next[s] = __builtin_amdgcn_raw_buffer_load_b32(mtab, row, 0, 0);
        }
    }
    __device__ __forceinline__ void store(float* lds, int k_left, int tid) const {
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const int id = tid + 256 * s;
            *reinterpret_cast<f32x4*>(lds + (id / (ROWS / 4)) * (ROWS + 4) + (id % (ROWS / 4)) * 4) = v[s];
        }
    }
};

// Fragment for MFMA steps t=0..3 of k-group kk: element t belongs to k = 8*kk + 4*h + t.
template <int ROWS, bool KC>
__device__ __forceinline__ f32x4 read_frag(const float* lds, int mn, int kk, int h) {
    if (KC) {
        return *reinterpret_cast<const f32x4*>(lds + mn * LDK + kk * 8 + 4 * h);
    } else {
        const float* p = lds + (kk * 8 + 4 * h) * (ROWS + 4) + mn;
        f32x4 r;
        r[0] = p[0];
        r[1] = p[ROWS + 4];
        r[2] = p[2 * (ROWS + 4)];
        r[3] = p[3 * (ROWS + 4)];
        return r;
    }
}

template <bool GATHER, int ROWS, bool KC>
struct PickLoader { using type = TileLoader<ROWS, KC>; };
template <int ROWS, bool KC>
struct PickLoader<true, ROWS, KC> { using type = GatherLoader<ROWS, KC>; };

// Epilogue shared by the fp32 and the bf16-operand kernels (the C/D layout of the 32x32 MFMAs does not depend on the
// operand type): col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
template <int TM, int TN, int BM, int BN, int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int li,
                                              int h, int bz) {
    float* C = p.C + (long long)bz * p.c_split_stride;
    if (EPI == EPI_FWD && p.pool_arg) {
        // 2x2 max pooling in registers: in the C/D layout of the 32x32 MFMA the four registers 4q..4q+3 of a lane are
        // four CONSECUTIVE rows, and the row order of the patch table makes those the four pixels of one pooling
        // window - bias, ReLU, max and arg-max never leave the lane (first maximum wins, like the stand-alone kernel).
        const float floor_v = (p.act == ACT_RELU) ? 0.f : -INFINITY;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + (wn * TN + j) * 32 + li;
                if (col >= p.N) continue;
                const float bv = p.bias ? p.bias[col] : 0.f;
                const int row0 = m0 + (wm * TM + i) * 32 + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = row0 + 8 * q;
                    if (row >= p.M) continue;
                    float best = fmaxf(acc[i][j][4 * q] + bv, floor_v);
                    int bi = 0;
#pragma unroll
                    for (int k = 1; k < 4; ++k) {
                        const float v = fmaxf(acc[i][j][4 * q + k] + bv, floor_v);
                        if (v > best) { best = v; bi = k; }
                    }
                    const long long o = (long long)(row >> 2) * p.ldc + col;
                    C[o] = best;
                    p.pool_arg[o] = (unsigned char)bi;
                }
            }
        }
        return;
    }
    uint32_t key = 0;
    const bool drop = (EPI == EPI_FWD) && (p.keep < 1.0f);
    float inv_keep = 1.0f;
    uint32_t keep_thr = 0;
    if (drop) {
        key = dtf_dropout_key(p.seed, p.step_ptr);
        inv_keep = 1.0f / p.keep;
        keep_thr = dtf_keep_threshold(p.keep);
    }
    // A tile that lies entirely inside C (all of them except the last row / column of tiles) skips the per-element
    // bounds tests: with them the epilogue is several thousand instructions and costs about a tenth of a short-K GEMM.
    const float relu_floor = (p.act == ACT_RELU) ? 0.f : -INFINITY;
    auto write_tile = [&](auto checked, auto simple, auto idx32) {
        constexpr bool CHECK = decltype(checked)::value;
        constexpr bool SIMPLE = decltype(simple)::value;   // forward: bias and ReLU / identity only; dgrad: ReLU mask or none
        constexpr bool IDX32 = decltype(idx32)::value;     // every element index of C fits 32 bits (dropout hashing)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + (wn * TN + j) * 32 + li;
                if (CHECK && col >= p.N) continue;
                float bv = 0.f;
                if (EPI == EPI_FWD && p.bias) bv = p.bias[col];
                const int row0 = m0 + (wm * TM + i) * 32 + 4 * h;
                float* Cp = C + (long long)row0 * p.ldc + col;
                const float* Yp = (EPI == EPI_DX && p.yprev) ? p.yprev + (long long)row0 * p.ldy + col : nullptr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (CHECK && EPI == EPI_DW && p.colsum_mfma && row0 + dr == p.M && p.colsum) {
                        p.colsum[(long long)bz * p.colsum_split_stride + col] = acc[i][j][r];    // ones row . B
                        continue;
                    }
                    if (CHECK && row0 + dr >= p.M) continue;
                    float v = acc[i][j][r];
                    if (EPI == EPI_FWD) {
                        v += bv;
                        if (SIMPLE) {
                            v = fmaxf(v, relu_floor);
                        } else {
                            if (p.act == ACT_SIGMOID) v = dtf_sigmoid(v);
                            else v = fmaxf(v, relu_floor);
                            if (drop) {
                                const bool kept = IDX32 ? dtf_keep32(key, (uint32_t)(row0 + dr) * (uint32_t)p.N + (uint32_t)col, keep_thr)
                                                        : dtf_keep64(key, (unsigned long long)(row0 + dr) * (unsigned)p.N + col, keep_thr);
                                v = kept ? v * inv_keep : 0.f;
                            }
                        }
                    } else if (EPI == EPI_DX) {
                        if (SIMPLE) {
                            if (Yp) v = (Yp[(long long)dr * p.ldy] > 0.f) ? v * p.scale_prev : 0.f;
                        } else if (Yp) {
                            const float y = Yp[(long long)dr * p.ldy];
                            if (p.act == ACT_RELU) v = (y > 0.f) ? v * p.scale_prev : 0.f;
                            else if (p.act == ACT_SIGMOID) v = v * y * (1.0f - y);
                        }
                    }
                    Cp[(long long)dr * p.ldc] = v;
                }
            }
        }
    };
    const bool simple = (EPI == EPI_FWD) ? (!drop && p.act != ACT_SIGMOID)
                                         : (EPI == EPI_DX ? (p.yprev == nullptr || p.act == ACT_RELU) : true);
    const bool small_idx = (long long)p.M * p.N < (1ll << 32);
    if (m0 + BM <= p.M && n0 + BN <= p.N) {
        if (simple) write_tile(std::false_type{}, std::true_type{}, std::true_type{});
        else if (small_idx) write_tile(std::false_type{}, std::false_type{}, std::true_type{});
        else write_tile(std::false_type{}, std::false_type{}, std::false_type{});
    } else {
        write_tile(std::true_type{}, std::false_type{}, std::false_type{});
    }
}

// Register budgets (waves per SIMD the compiler must leave room for).  A wave that owns a single 32x32 sub-tile: six
// (80 registers; it took 82-88 on its own, which is five waves) - the small-tile kernels, thin outputs and convolution
// gathers, hide their load latency behind resident waves, not behind long MFMA runs.  A wave with four sub-tiles: three
// (168 registers instead of 195), so that with one LDS stage three 128x128 blocks share a CU instead of two (headline
// step -1.6 %, widened MLP -3 %).  Eight waves for the small tiles (64 registers) spill and run 10-70 % slower.
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, int EPI, bool GATHER = false>
__global__ __launch_bounds__(256, (BM / WM == 32 && BN / WN == 32) ? 6 : ((BM / WM) * (BN / WN) == 4096 ? 3 : 1))
void gemm_f32_kernel(const GemmArgs p) {
    static_assert(WM * WN == 4, "block is four waves");
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold one 32x32 MFMA tile");
    using LoadA = typename PickLoader<GATHER, BM, A_KC>::type;
    using LoadB = TileLoader<BN, B_KC>;
    constexpr int STAGE = LoadA::LDS_FLOATS + LoadB::LDS_FLOATS;

    extern __shared__ __attribute__((aligned(16))) float smem[];   // two stages of [A tile | B tile]
    unsigned long long dbg_entry = 0;
    if (p.dbg & 8) dbg_entry = __builtin_amdgcn_s_memrealtime();

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, h = lane >> 5;

    // Which tile / K split this block computes.  Workgroups are dealt to the 8 XCDs round-robin in dispatch order
    // (x fastest, then y, then z), and every XCD has its own L2: with the plain mapping the blocks that share an
    // operand panel - the row tiles of one K split of a wgrad all read the same slab of dZ - sit on eight different
    // L2s and each fetches the panel for itself.  Remapped, the blocks with equal (dispatch index mod 8) take a
    // contiguous run of the grid (bijective also when the block count is no multiple of 8).
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_remap) {
        const int gx = gridDim.x, gy = gridDim.y;
        const int nwg = gx * gy * (int)gridDim.z;
        const int lin = bx + gx * (by + gy * bz);
        const int q = nwg >> 3, r = nwg & 7, xcd = lin & 7;
        const int to = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
        bx = to % gx;
        by = (to / gx) % gy;
        bz = to / (gx * gy);
    }
    const int m0 = by * BM;
    const int n0 = bx * BN;
    const int kbeg = bz * p.k_chunk;
    const int kend = min(p.K, kbeg + p.k_chunk);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // does this wave own any output inside C?  (TM x TN sub-tiles starting at its first row / column)
    const bool wave_live = (m0 + wm * TM * 32 < p.M) && (n0 + wn * TN * 32 < p.N);
    float csum = 0.f;  // wgrad: partial bias gradient of column n0 + tid
    const bool do_colsum = (EPI == EPI_DW) && (p.colsum != nullptr) && (by == 0) && !p.colsum_mfma;

    LoadA la;
    LoadB lb;
    if constexpr (GATHER) la.init_gather(p, m0, kbeg, tid);
    else {
        la.init(p.A, p.lda, m0, p.M, kbeg, p.K);
        if constexpr (EPI == EPI_DW && !A_KC) la.ones_row = p.colsum_mfma && p.colsum != nullptr;
    }
    lb.init(p.B, p.ldb, n0, p.N, kbeg, p.K);
    la.load(kbeg, kbeg, kend - kbeg, tid);
    lb.load(kbeg, kbeg, kend - kbeg, tid);
    la.store(smem, kend - kbeg, tid);
    lb.store(smem + LoadA::LDS_FLOATS, kend - kbeg, tid);
    __syncthreads();

    unsigned long long dbg_t0 = 0, dbg_r0 = 0;
    if (p.dbg & 8) { dbg_t0 = __builtin_amdgcn_s_memtime(); dbg_r0 = __builtin_amdgcn_s_memrealtime(); }
    // Main loop.  The fragments of k-group kk+1 are requested from LDS after the first TM*TN MFMAs of k-group kk and
    // before its other 3*TM*TN (two fragment register sets; scheduling fences keep the compiler from sinking the
    // reads next to their use): a read issued right before its consumer leaves the matrix pipe idle for the LDS
    // latency once per k-group.  In the last k-group of a tile the same slot holds the store of the next tile, the
    // barrier and the request for that tile's first fragments, so these are covered by 3*TM*TN MFMAs as well.
    int stage = 0;
    const int stage_stride = p.single_lds ? 0 : STAGE;
    f32x4 fa[2][TM], fb[2][TN];
    {
        const float* As = smem;
        const float* Bs = As + LoadA::LDS_FLOATS;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = read_frag<BM, A_KC>(As, (wm * TM + i) * 32 + li, 0, h);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = read_frag<BN, B_KC>(Bs, (wn * TN + j) * 32 + li, 0, h);
    }
    // The loop exists in two compiled versions: with MFMAs, and - for a wave whose 32-row / 32-column sub-tiles lie
    // entirely past M / N (half the waves of the last row tile of a 784-row wgrad) - without: such a wave keeps loading
    // and synchronising with its block but leaves the matrix pipe to the co-resident waves.  (A run-time test around
    // the MFMAs inside ONE loop costs the compiler its schedule: measured +40 % on the wgrad kernel.)
    auto main_loop = [&](auto live_tag) {
    constexpr bool LIVE = decltype(live_tag)::value;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = (k0 + BK) < kend;
        const float* As = smem + stage * stage_stride;
        const float* Bs = As + LoadA::LDS_FLOATS;
        if (more && !(p.dbg & 1)) {     // bit 1 (diagnostic): no global loads after the first tile
            la.load(k0 + BK, kbeg, kend - k0 - BK, tid);
            lb.load(k0 + BK, kbeg, kend - k0 - BK, tid);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            constexpr int LAST = BK / 8 - 1;
            const int cur = kk & 1, nxt = cur ^ 1;
            // step t = 0 first: whatever wait precedes it only concerns reads issued a whole k-group ago
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (LIVE) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][0], fb[cur][j][0], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kk < LAST) {
                if (!(p.dbg & 16)) {      // bit 16 (diagnostic): no fragment reads inside a tile
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[nxt][i] = read_frag<BM, A_KC>(As, (wm * TM + i) * 32 + li, kk + 1, h);
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[nxt][j] = read_frag<BN, B_KC>(Bs, (wn * TN + j) * 32 + li, kk + 1, h);
                } else {
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[nxt][i] = fa[cur][i];
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[nxt][j] = fb[cur][j];
                }
            } else {
                if (EPI == EPI_DW) {
                    if (do_colsum && tid < BN) {
                        // B tile is n-contiguous here: Bs[k][n].  Rows past kend were staged as zeros.
#pragma unroll 8
                        for (int k = 0; k < BK; ++k) csum += Bs[k * (BN + 4) + tid];
                    }
                }
                // Unconditional barrier and reads (after the last tile they fetch stale data nobody uses): one path
                // through here lets the compiler count outstanding LDS operations exactly instead of draining them.
                float* nx = smem + (stage ^ 1) * stage_stride;
                if (p.single_lds) __syncthreads();      // every wave has its fragments of this tile in registers
                if (more && !(p.dbg & 4)) {     // bit 4 (diagnostic): no LDS stores after the first tile
                    la.store(nx, kend - k0 - BK, tid);
                    lb.store(nx + LoadA::LDS_FLOATS, kend - k0 - BK, tid);
                }
                if (!(p.dbg & 2)) __syncthreads();      // bit 2 (diagnostic): no barrier inside the loop
                const float* An = nx;
                const float* Bn = nx + LoadA::LDS_FLOATS;
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[nxt][i] = read_frag<BM, A_KC>(An, (wm * TM + i) * 32 + li, 0, h);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[nxt][j] = read_frag<BN, B_KC>(Bn, (wn * TN + j) * 32 + li, 0, h);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (LIVE) {
#pragma unroll
                for (int t = 1; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][t], fb[cur][j][t], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        stage ^= 1;
    }
    };
    if (wave_live) main_loop(std::true_type{});
    else main_loop(std::false_type{});

    unsigned long long dbg_t1 = 0;
    if (p.dbg & 8) dbg_t1 = __builtin_amdgcn_s_memtime();
    gemm_epilogue<TM, TN, BM, BN, EPI>(p, acc, m0, n0, wm, wn, li, h, bz);
    if ((p.dbg & 8) && tid == 0) {   // timing-only diagnostic: loop cycles and 100 MHz ticks into the tile's first row
        float* C = p.C + (long long)bz * p.c_split_stride;
        C[(long long)m0 * p.ldc + n0] = (float)(__builtin_amdgcn_s_memtime() - dbg_t0);
        C[(long long)m0 * p.ldc + n0 + 1] = (float)(__builtin_amdgcn_s_memrealtime() - dbg_r0);
        C[(long long)m0 * p.ldc + n0 + 2] = (float)(dbg_t1 - dbg_t0);
        C[(long long)m0 * p.ldc + n0 + 3] = (float)(dbg_entry & 0xFFFFF);                         // block entry, 10 ns ticks
        C[(long long)m0 * p.ldc + n0 + 4] = (float)(__builtin_amdgcn_s_memrealtime() & 0xFFFFF);  // block exit
        // where wave 0 ran: HW_ID (wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]) and XCC_ID
        C[(long long)m0 * p.ldc + n0 + 5] = (float)(__builtin_amdgcn_s_getreg(4 | (31 << 11)) & 0xFFFF);
        C[(long long)m0 * p.ldc + n0 + 6] = (float)(__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xF);
    }
    if (EPI == EPI_DW) {
        if (do_colsum && tid < BN && (n0 + tid) < p.N)
            p.colsum[(long long)bz * p.colsum_split_stride + n0 + tid] = csum;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16-operand variant for wide layers (opt-in mixed precision): operands are read as fp32, rounded to bf16 on the way
// into LDS, multiplied on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate) and accumulated in fp32; outputs, bias,
// activations, dropout and the optimizer stay fp32.  128x128 tile, four waves of 64x64, 32-deep k tiles (two
// 16-deep MFMA steps); both LDS images are k-contiguous so every fragment is one 16-byte read: lane (r = lane & 31,
// h = lane >> 5) holds A[row r][k = 8h .. 8h+7] and B[k = 8h .. 8h+7][col r].
template <bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const GemmArgs p) {
    constexpr int BM = 128, BN = 128, TM = 2, TN = 2;
    using LoadA = typename PickBf16Loader<BM, A_KC>::type;
    using LoadB = typename PickBf16Loader<BN, B_KC>::type;
    constexpr int TILE = BM * LDKB;                 // bf16 elements per operand image (BM == BN)
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * 2 * TILE];   // two stages of [A | B]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * p.k_chunk;
    const int kend = min(p.K, kbeg + p.k_chunk);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    LoadA la;
    LoadB lb;
    la.init(p.A, p.lda, m0, p.M, kbeg, p.K);
    lb.init(p.B, p.ldb, n0, p.N, kbeg, p.K);
    la.load(kbeg, kbeg, kend - kbeg, tid);
    lb.load(kbeg, kbeg, kend - kbeg, tid);
    la.store_bf16(smem, kend - kbeg, tid);
    lb.store_bf16(smem + TILE, kend - kbeg, tid);
    __syncthreads();

    int stage = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = (k0 + BK) < kend;
        const unsigned short* As = smem + stage * 2 * TILE;
        const unsigned short* Bs = As + TILE;
        if (more) {
            la.load(k0 + BK, kbeg, kend - k0 - BK, tid);
            lb.load(k0 + BK, kbeg, kend - k0 - BK, tid);
        }
        bf16x8 a[2][TM], b[2][TN];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[ks][i] = *reinterpret_cast<const bf16x8*>(As + bf16_lds_off((wm * TM + i) * 32 + li, ks * 16 + 8 * h));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[ks][j] = *reinterpret_cast<const bf16x8*>(Bs + bf16_lds_off((wn * TN + j) * 32 + li, ks * 16 + 8 * h));
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
        if (more) {
            unsigned short* nx = smem + (stage ^ 1) * 2 * TILE;
            la.store_bf16(nx, kend - k0 - BK, tid);
            lb.store_bf16(nx + TILE, kend - k0 - BK, tid);
        }
        __syncthreads();
        stage ^= 1;
    }
    gemm_epilogue<TM, TN, BM, BN, EPI>(p, acc, m0, n0, wm, wn, li, h, blockIdx.z);
}

template <bool A_KC, bool B_KC, int EPI>
int launch_bf16(const GemmArgs& a, hipStream_t s) {
    dim3 grid((a.N + 127) / 128, (a.M + 127) / 128, 1);
    hipLaunchKernelGGL((gemm_bf16_kernel<A_KC, B_KC, EPI>), grid, dim3(256), 0, s, a);
    DTF_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 products on the bf16 matrix cores.  gfx950 multiplies bf16 sixteen times faster than fp32, so each fp32 operand
// element is split EXACTLY into three bf16 limbs on its way into LDS (split3) and the product is accumulated in fp32
// from the partial products of the limbs, most significant last:
//     hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi, mid.lo, lo.mid   (NPROD = 8; lo.lo, 2^-32 of a product, is dropped;
//                                                                    NPROD = 6 also drops the two 2^-24 terms)
// Every partial product of two bf16 values is exact in the fp32 accumulator's input, so the result differs from an
// fp32 FMA chain only in rounding order (measured against fp64: the same or a smaller error than the fp32 MFMA
// kernel's, for 6, 8 and 9 products alike).  128x128 tile, four waves of 64x64, one 32-deep k tile in LDS as six
// [128][LDKB] limb planes (60 KB: two blocks per CU); the next tile's global loads fly during the MFMAs.
// The chip is power-limited here: with random operands the MFMA phase alone runs at ~1.6 GHz (4430 cycles per k tile
// and block pair, 92 % of the pipe), the split phase at ~2.3 GHz, and their times add up whether or not the two
// resident blocks are staggered - so fewer limb products, not more overlap, is what buys time
// (DTF_GEMM_DBG bits 8 / 4 / 16 with bench/gemm_split_probe.py measure the phases).
template <bool A_KC, bool B_KC, int EPI, int NPROD>
__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(const GemmArgs p) {
    constexpr int BM = 128, BN = 128, TM = 2, TN = 2;
    using LoadA = typename PickBf16Loader<BM, A_KC>::type;
    using LoadB = typename PickBf16Loader<BN, B_KC>::type;
    constexpr int TILE = BM * LDKB;                 // bf16 elements per limb plane (BM == BN)
    __shared__ __attribute__((aligned(16))) unsigned short smem[6 * TILE];   // A hi|mid|lo, B hi|mid|lo

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * p.k_chunk;
    const int kend = min(p.K, kbeg + p.k_chunk);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr bool CAN_COLSUM = (EPI == EPI_DW) && !B_KC;
    const bool do_colsum = CAN_COLSUM && p.colsum != nullptr && blockIdx.y == 0;
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};

    LoadA la;
    LoadB lb;
    la.init(p.A, p.lda, m0, p.M, kbeg, p.K);
    lb.init(p.B, p.ldb, n0, p.N, kbeg, p.K);
    la.load(kbeg, kbeg, kend - kbeg, tid);
    lb.load(kbeg, kbeg, kend - kbeg, tid);
    unsigned long long dbg_t0 = 0, dbg_r0 = 0;
    if (p.dbg & 8) { dbg_t0 = __builtin_amdgcn_s_memtime(); dbg_r0 = __builtin_amdgcn_s_memrealtime(); }

    // limb pairs in order of significance
    constexpr int PA[9] = {0, 0, 1, 1, 0, 2, 1, 2, 2};
    constexpr int PB[9] = {0, 1, 0, 1, 2, 0, 2, 1, 2};

    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        if (!(p.dbg & 4) || k0 == kbeg) {
        la.store_split(smem, TILE, tid);
        lb.store_split(smem + 3 * TILE, TILE, tid);
        }
        if constexpr (CAN_COLSUM) {
            if (do_colsum) lb.add_colsum(cs);
        }
        __syncthreads();
        if (k0 + BK < kend) {
            la.load(k0 + BK, kbeg, kend - k0 - BK, tid);
            lb.load(k0 + BK, kbeg, kend - k0 - BK, tid);
        }
        if (!(p.dbg & 16))
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[3][TM], b[3][TN];
#pragma unroll
            for (int l = 0; l < 3; ++l) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[l][i] = *reinterpret_cast<const bf16x8*>(smem + l * TILE + bf16_lds_off((wm * TM + i) * 32 + li, ks * 16 + 8 * h));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[l][j] = *reinterpret_cast<const bf16x8*>(smem + (3 + l) * TILE + bf16_lds_off((wn * TN + j) * 32 + li, ks * 16 + 8 * h));
            }
#pragma unroll
            for (int q = NPROD - 1; q >= 0; --q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]][i], b[PB[q]][j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    if constexpr (CAN_COLSUM) {
        if (do_colsum) {         // uniform per block; the 8 k-row groups of a column meet in LDS (free after the last barrier)
            float* red = reinterpret_cast<float*>(smem);
            *reinterpret_cast<f32x4*>(red + (tid >> 5) * BN + (tid & 31) * 4) = cs;
            __syncthreads();
            if (tid < BN && n0 + tid < p.N) {
                float t = 0.f;
#pragma unroll
                for (int g = 0; g < 8; ++g) t += red[g * BN + tid];
                p.colsum[(long long)blockIdx.z * p.colsum_split_stride + n0 + tid] = t;
            }
        }
    }
    gemm_epilogue<TM, TN, BM, BN, EPI>(p, acc, m0, n0, wm, wn, li, h, blockIdx.z);
    if ((p.dbg & 8) && tid == 0) {   // timing-only diagnostic: block cycles and 100 MHz ticks over the tile's first outputs
        float* C = p.C + (long long)blockIdx.z * p.c_split_stride;
        C[(long long)m0 * p.ldc + n0] = (float)(__builtin_amdgcn_s_memtime() - dbg_t0);
        C[(long long)m0 * p.ldc + n0 + 1] = (float)(__builtin_amdgcn_s_memrealtime() - dbg_r0);
    }
}

inline int env_dbg();

// Number of limb products the fp32 GEMMs use on the bf16 matrix cores; 0 = native fp32 MFMA (the default).
int g_split_products = -1;
inline int split_products() {
    if (g_split_products < 0) {
        const char* e = getenv("DTF_GEMM_SPLIT");
        const int v = e ? atoi(e) : 0;
        g_split_products = (v == 6 || v == 8 || v == 9) ? v : 0;
    }
    return g_split_products;
}

// the split kernel wants whole float4 vectors on every operand and enough 128x128 tiles to be worth its 60 KB blocks
inline bool split_shape_ok(const GemmArgs& a, int splits) {
    if (((a.M | a.N | a.K | a.lda | a.ldb | a.k_chunk) & 3) != 0) return false;
    if (a.M < 64 || a.N < 64) return false;
    return (long long)((a.M + 127) / 128) * ((a.N + 127) / 128) * splits >= 64;
}

template <bool A_KC, bool B_KC, int EPI>
int launch_x3(const GemmArgs& a, int splits, int nprod, hipStream_t s) {
    dim3 grid((a.N + 127) / 128, (a.M + 127) / 128, splits);
    GemmArgs args = a;
    args.dbg = env_dbg();
    if (nprod == 6) hipLaunchKernelGGL((gemm_x3_kernel<A_KC, B_KC, EPI, 6>), grid, dim3(256), 0, s, args);
    else if (nprod == 9) hipLaunchKernelGGL((gemm_x3_kernel<A_KC, B_KC, EPI, 9>), grid, dim3(256), 0, s, args);
    else hipLaunchKernelGGL((gemm_x3_kernel<A_KC, B_KC, EPI, 8>), grid, dim3(256), 0, s, args);
    DTF_CHECK_LAUNCH();
    return 0;
}

// db[n] = sum over m of dZ[m][n]: one thread per 4 columns and row block, row blocks meet through atomics on a zeroed db
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dZ, float* __restrict__ db, int M, int N,
                                                     int rows_per_block) {
    const int c4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c4 >= N) return;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < r1; ++r) s0 += *reinterpret_cast<const f32x4*>(dZ + (long long)r * N + c4);
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(db + c4 + e, s0[e]);
}

inline int env_dbg() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("DTF_GEMM_DBG");
        v = e ? atoi(e) : 0;
    }
    return v;
}

// One LDS stage instead of two (half the LDS per block: more resident blocks per CU, for a second barrier per k tile)?
// Default: the convolution gathers and the 128x128 / 64x128 dense tiles (three instead of two, four instead of three
// blocks per CU); the small dense tiles gain nothing.  DTF_GEMM_SINGLE=0/1 forces it for every kernel;
// DTF_GEMM_SINGLE_TILES is a mask of dense tiles that take one stage (1: 128x128, 2: 64x128, 4: the others).
inline int gemm_single_stage(int bm, int bn, bool gather) {
    static const int forced = getenv("DTF_GEMM_SINGLE") ? atoi(getenv("DTF_GEMM_SINGLE")) : -1;
    static const int mask = getenv("DTF_GEMM_SINGLE_TILES") ? atoi(getenv("DTF_GEMM_SINGLE_TILES")) : 3;
    if (forced >= 0) return forced;
    if (gather) return 1;
    const int bit = (bm == 128 && bn == 128) ? 1 : ((bm == 64 && bn == 128) ? 2 : 4);
    return (mask & bit) ? 1 : 0;
}

template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, int EPI, bool GATHER = false>
int launch_cfg(const GemmArgs& a, int splits, hipStream_t s) {
    using LoadA = TileLoader<BM, A_KC>;     // the gather loaders build the same LDS images
    using LoadB = TileLoader<BN, B_KC>;
    const int single = gemm_single_stage(BM, BN, GATHER);
    const size_t lds = (single ? 1 : 2) * (LoadA::LDS_FLOATS + LoadB::LDS_FLOATS) * sizeof(float);
    auto kern = gemm_f32_kernel<BM, BN, WM, WN, A_KC, B_KC, EPI, GATHER>;
    if (lds > 48 * 1024) {
        static bool once = false;  // per instantiation
        if (!once) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            once = true;
        }
    }
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, splits);
    GemmArgs args = a;
    args.dbg = env_dbg();
    args.single_lds = single;
    // DTF_GEMM_XCD (tuning aid) is a mask of the kernels that walk their grid XCD by XCD: 1 dense wgrad, 2 the
    // convolution gathers, 4 dense forward / dgrad
    static const int xcd_mask = getenv("DTF_GEMM_XCD") ? atoi(getenv("DTF_GEMM_XCD")) : 0;
    const int xcd_bit = GATHER ? 2 : (EPI == EPI_DW ? 1 : 4);
    args.xcd_remap = (xcd_mask & xcd_bit) && (long long)grid.x * grid.y * grid.z >= 16 ? 1 : 0;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, args);
    DTF_CHECK_LAUNCH();
    return 0;
}

// Tile choice by a small cost model.  A CU holds as many blocks as its 160 KB of LDS allow; resident blocks share
// the four matrix pipes, so a "round" of resident blocks takes (blocks per CU) x (one block's MFMA cycles), and a
// grid that does not fill its last round evenly pays for that round as if it were as full as its fullest CU.
// N <= 32 uses the narrow tile so thin output layers (10 / 32 columns) do not pay for 128 columns.
inline long long nblocks(const GemmArgs& a, int bm, int bn, int splits) {
    return (long long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn) * splits;
}

inline size_t tile_lds_bytes(int bm, int bn, bool a_kc, bool b_kc) {
    const size_t fa = a_kc ? (size_t)bm * LDK : (size_t)BK * (bm + 4);
    const size_t fb = b_kc ? (size_t)bn * LDK : (size_t)BK * (bn + 4);
    return 2 * (fa + fb) * sizeof(float);
}

constexpr double BLOCK_OVERHEAD_CYCLES = 6000.0;   // first-tile latency + epilogue of one block (measured ~3 us)

inline double tile_cost(long long M, long long N, long long kc, int splits, int bm, int bn, bool a_kc, bool b_kc,
                        long long res_known = 0) {
    const long long nb = ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * splits;
    long long res = res_known > 0 ? res_known : (long long)((160 * 1024) / tile_lds_bytes(bm, bn, a_kc, b_kc));
    if (res_known <= 0 && res > 4) res = 4;
    const long long slots = 256 * res;
    const long long full = nb / slots, rem = nb % slots;
    const long long seq = full * res + (rem + 255) / 256;          // blocks the busiest CU runs one after another
    double c = (double)seq * ((double)bm * bn * (double)kc / 128.0 + BLOCK_OVERHEAD_CYCLES);
    if (full == 0 && rem <= 256) c *= 1.2;     // a lone block per CU cannot hide its LDS-store and barrier phases
    if (bm * bn <= 64 * 64) c *= 1.1;          // 32x32 wave tiles: twice the LDS fragment traffic per MFMA (measured)
    return c;
}

// Forward / dgrad tile choice.  Blocks that share a CU overlap their first-tile latency and epilogues, so a round of
// resident blocks pays the per-block overhead once: round time = (blocks per CU) x (one block's MFMA cycles) +
// overhead.  Residency as built: three 128x128 blocks (one LDS stage, 168 registers), four 64x128, four 64x64.  Few
// co-resident blocks hide their barrier phases badly (one: +20 %, two: +10 %); smaller wave tiles read more fragments
// per MFMA (+3 % / +5 %).  Measured anchors: 32768 x 128 x 128 runs 24 us as 256 blocks of 128x128 and 14 us as 1024
// blocks of 64x64; 262144 x 784 x 128 is equal on 128x128 and 64x128.
inline double fwd_tile_cost(long long M, long long N, long long kc, int splits, int bm, int bn) {
    const long long nb = ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * splits;
    const long long res = (bm == 128 && bn == 128) ? 3 : 4;
    const long long slots = 256 * res;
    const long long full = nb / slots, rem = nb % slots;
    const double w = (double)bm * bn * (double)kc / 128.0;
    double c = (double)full * ((double)res * w + BLOCK_OVERHEAD_CYCLES);
    if (rem > 0) {
        const long long on_cu = (rem + 255) / 256;
        double last = (double)on_cu * w + BLOCK_OVERHEAD_CYCLES;
        if (full == 0) last *= on_cu == 1 ? 1.2 : (on_cu == 2 ? 1.1 : 1.0);
        c += last;
    }
    if (bm * bn <= 64 * 64) c *= 1.05;
    else if (bm * bn <= 64 * 128) c *= 1.03;
    return c;
}

// 128 (=128x128), 6428 (=64x128) or 64 (=64x64)
inline int pick_tile(const GemmArgs& a, int splits, bool a_kc, bool b_kc) {
    if (a.N <= 64 || a.M <= 64) return 64;
    const long long kc = a.k_chunk < a.K ? a.k_chunk : a.K;
    int best = 128;
    double cost = fwd_tile_cost(a.M, a.N, kc, splits, 128, 128);
    const double c1 = fwd_tile_cost(a.M, a.N, kc, splits, 64, 128);
    if (c1 < cost) { cost = c1; best = 6428; }
    const double c2 = fwd_tile_cost(a.M, a.N, kc, splits, 64, 64);
    if (c2 < cost) { cost = c2; best = 64; }
    return best;
}

inline int env_tile() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("DTF_GEMM_TILE");   // tuning aid: 64 / 128 / 6428 (=64x128) forces a tile
        v = e ? atoi(e) : 0;
    }
    return v;
}

template <bool A_KC, bool B_KC, int EPI, bool GATHER>
int launch_tiled(const GemmArgs& a, int splits, hipStream_t s, int tile_hint) {
    if constexpr (!GATHER) {
        if (split_products() && split_shape_ok(a, splits)) return launch_x3<A_KC, B_KC, EPI>(a, splits, split_products(), s);
    }
    // Tile codes: 128 (=128x128), 6428 (=64x128), 64 (=64x64), 12864 (=128x64: two 32x32 sub-tiles per wave on a
    // 64-column output, three fragment reads per eight MFMAs where the 64x64 tile needs two per four).
    int t = env_tile() ? env_tile() : tile_hint;
    if (a.N <= 32) return launch_cfg<128, 32, 4, 1, A_KC, B_KC, EPI, GATHER>(a, splits, s);
    if (a.N <= 64 || a.M <= 64) {
        if (t == 12864 && a.M >= 128) return launch_cfg<128, 64, 2, 2, A_KC, B_KC, EPI, GATHER>(a, splits, s);
        return launch_cfg<64, 64, 2, 2, A_KC, B_KC, EPI, GATHER>(a, splits, s);
    }
    if (t != 64 && t != 128 && t != 6428) t = pick_tile(a, splits, A_KC, B_KC);
    if (t == 128) return launch_cfg<128, 128, 2, 2, A_KC, B_KC, EPI, GATHER>(a, splits, s);
    if (t == 6428) return launch_cfg<64, 128, 2, 2, A_KC, B_KC, EPI, GATHER>(a, splits, s);
    return launch_cfg<64, 64, 2, 2, A_KC, B_KC, EPI, GATHER>(a, splits, s);
}

// Implicit-GEMM convolutions use the same rule.
template <bool A_KC, bool B_KC, int EPI>
int launch_gather(const GemmArgs& a, int splits, hipStream_t s, int tile_hint = 0) {
    return launch_tiled<A_KC, B_KC, EPI, true>(a, splits, s, tile_hint);
}

template <bool A_KC, bool B_KC, int EPI>
int launch_gemm(const GemmArgs& a, int splits, hipStream_t s, int tile_hint = 0) {
    return launch_tiled<A_KC, B_KC, EPI, false>(a, splits, s, tile_hint);
}

// out[i] = sum_s slabs[s * slab_stride + i].  A 1024-thread block covers 64 float4 columns with 16
// split-groups (one wave each); every thread keeps 4 independent loads in flight, then the 16
// partial sums meet in LDS.  Element i >= n_main goes to out_tail (the bias gradient that rides
// behind the weight gradient in every slab).
__global__ __launch_bounds__(1024) void reduce_slabs_kernel(const float* __restrict__ slabs, long long slab_stride,
                                                            int splits, float* __restrict__ out_main,
                                                            long long n_main, float* __restrict__ out_tail,
                                                            long long n_total) {
    __shared__ f32x4 part[16][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long long i4 = (long long)blockIdx.x * 64 + c;   // float4 column
    const long long i = i4 * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (i < n_total) {
        const bool vec = (i + 3 < n_total) && ((slab_stride & 3) == 0);
        f32x4 a0 = acc, a1 = acc, a2 = acc, a3 = acc;
        int s = g;
        if (vec) {
            for (; s + 48 < splits; s += 64) {
                a0 += *reinterpret_cast<const f32x4*>(slabs + (long long)s * slab_stride + i);
                a1 += *reinterpret_cast<const f32x4*>(slabs + (long long)(s + 16) * slab_stride + i);
                a2 += *reinterpret_cast<const f32x4*>(slabs + (long long)(s + 32) * slab_stride + i);
                a3 += *reinterpret_cast<const f32x4*>(slabs + (long long)(s + 48) * slab_stride + i);
            }
            for (; s < splits; s += 16) a0 += *reinterpret_cast<const f32x4*>(slabs + (long long)s * slab_stride + i);
        } else {
            for (; s < splits; s += 16)
                for (int j = 0; j < 4; ++j)
                    if (i + j < n_total) a0[j] += slabs[(long long)s * slab_stride + i + j];
        }
        acc = (a0 + a1) + (a2 + a3);
    }
    part[g][c] = acc;
    __syncthreads();
    if (g == 0 && i < n_total) {
        f32x4 t = part[0][c];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += part[k][c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long e = i + j;
            if (e < n_main) out_main[e] = t[j];
            else if (e < n_total) out_tail[e - n_main] = t[j];
        }
    }
}

// The same sum for a handful of slabs: one thread per float4 column walks the slabs itself (the 16-group kernel above
// leaves most of its waves idle below 16 slabs).  slab_stride and n_main are multiples of 4 here.
__global__ __launch_bounds__(256) void reduce_few_slabs_kernel(const float* __restrict__ slabs, long long slab_stride, int splits,
                                                               float* __restrict__ out_main, long long n_main,
                                                               float* __restrict__ out_tail, long long n_total) {
    const long long total4 = (n_total + 3) / 4;
    for (long long i4 = (long long)blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += (long long)gridDim.x * 256) {
        const long long i = i4 * 4;
        if (i + 3 < n_total) {
            f32x4 acc = *reinterpret_cast<const f32x4*>(slabs + i);
            for (int s = 1; s < splits; ++s) acc += *reinterpret_cast<const f32x4*>(slabs + (long long)s * slab_stride + i);
            if (i + 3 < n_main) {
                *reinterpret_cast<f32x4*>(out_main + i) = acc;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const long long e = i + j;
                    if (e < n_main) out_main[e] = acc[j];
                    else out_tail[e - n_main] = acc[j];
                }
            }
        } else {
            for (int j = 0; j < 4 && i + j < n_total; ++j) {
                const long long e = i + j;
                float t = 0.f;
                for (int s = 0; s < splits; ++s) t += slabs[(long long)s * slab_stride + e];
                if (e < n_main) out_main[e] = t;
                else out_tail[e - n_main] = t;
            }
        }
    }
}

inline void launch_reduce_slabs(const float* slabs, long long stride, int splits, float* out_main, long long n_main,
                                float* out_tail, long long n_total, hipStream_t stream) {
    if (splits <= 8 && (stride & 3) == 0 && (n_main & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(out_main)) & 15) == 0) {
        long long blocks = ((n_total + 3) / 4 + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(reduce_few_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, slabs, stride, splits, out_main,
                           n_main, out_tail, n_total);
    } else {
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n_total + 255) / 256)), dim3(1024), 0, stream, slabs, stride, splits,
                           out_main, n_main, out_tail, n_total);
    }
}

}  // namespace

// 0: fp32 GEMMs on the fp32 MFMA (default); 6 / 8 / 9: on the bf16 matrix cores from that many limb products
// (gemm_x3_kernel).  Workspace sizes depend on it: set it before querying dtf_dense_bwd_dw_splits.
DTF_API void dtf_gemm_set_split_products(int n) { g_split_products = (n == 6 || n == 8 || n == 9) ? n : 0; }
DTF_API int dtf_gemm_get_split_products() { return split_products(); }

// Y = dropout(act(X.W + b)).  keep >= 1 disables dropout.
DTF_API int dtf_dense_fwd(const float* X, const float* W, const float* bias, float* Y, int M, int N, int K,
                          int act, float keep, uint32_t seed, const long long* step_ptr,
                          hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    GemmArgs a = {};
    a.A = X; a.lda = K;
    a.B = W; a.ldb = N;
    a.C = Y; a.ldc = N;
    a.M = M; a.N = N; a.K = K;
    a.k_chunk = ((K + BK - 1) / BK) * BK;
    a.bias = bias; a.act = act; a.keep = keep; a.seed = seed; a.step_ptr = step_ptr;
    return launch_gemm<true, false, EPI_FWD>(a, 1, stream);
}

// Forward products with few output tiles and a long K (a wide layer at a small batch, e.g. 1024 x 3136 -> 512: 128
// tiles for 256 CUs) split K over blockIdx.z into slabs; this kernel adds the slabs, the bias, the activation and the dropout mask.
__global__ __launch_bounds__(256) void reduce_fwd_slabs_kernel(const float* __restrict__ slabs, long long slab_stride,
                                                               int splits, const float* __restrict__ bias, int act,
                                                               float keep, uint32_t seed, const long long* step_ptr,
                                                               float* __restrict__ Y, int N, long long total4) {
    const bool drop = keep < 1.0f;
    const uint32_t key = drop ? dtf_dropout_key(seed, step_ptr) : 0u;
    const float inv_keep = drop ? 1.0f / keep : 1.0f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        f32x4 acc = *reinterpret_cast<const f32x4*>(slabs + 4 * i);
        for (int s = 1; s < splits; ++s) acc += *reinterpret_cast<const f32x4*>(slabs + (long long)s * slab_stride + 4 * i);
        const int col = (int)((4 * i) % N);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[e] + (bias ? bias[col + e] : 0.f);
            if (act == ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == ACT_SIGMOID) v = dtf_sigmoid(v);
            if (drop) v = dtf_keep64(key, (unsigned long long)(4 * i + e), dtf_keep_threshold(keep)) ? v * inv_keep : 0.f;   // index = row * N + col
            acc[e] = v;
        }
        *reinterpret_cast<f32x4*>(Y + 4 * i) = acc;
    }
}

// K splits of the forward product (1 = no split).  Only products that cannot fill the chip with output tiles split.
DTF_API int dtf_dense_fwd_splits(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K < 1024 || N % 4 != 0 || N <= 32) return 1;
    static const int forced = getenv("DTF_FWD_SPLITS") ? atoi(getenv("DTF_FWD_SPLITS")) : 0;   // tuning aid
    if (forced > 0) return forced;
    const long long tiles = (long long)((M + 63) / 64) * ((N + 63) / 64);     // the smallest tile gives the most blocks
    if (tiles >= 256) return 1;
    long long s = 512 / tiles;
    const long long by_k = K / 256;                                           // at least 8 k-tiles per split
    if (s > by_k) s = by_k;
    return s < 2 ? 1 : (int)s;
}

// dtf_dense_fwd with scratch for the K-split path: workspace holds splits * M * N floats.
DTF_API int dtf_dense_fwd_ws(const float* X, const float* W, const float* bias, float* Y, int M, int N, int K, int act,
                             float keep, uint32_t seed, const long long* step_ptr, float* workspace,
                             long long workspace_floats, hipStream_t stream) {
    const int splits = dtf_dense_fwd_splits(M, N, K);
    const long long slab = (long long)M * N;
    if (splits <= 1 || !workspace || workspace_floats < splits * slab)
        return dtf_dense_fwd(X, W, bias, Y, M, N, K, act, keep, seed, step_ptr, stream);
    GemmArgs a = {};
    a.A = X; a.lda = K;
    a.B = W; a.ldb = N;
    a.C = workspace; a.ldc = N; a.c_split_stride = slab;
    a.M = M; a.N = N; a.K = K;
    int chunk = (K + splits - 1) / splits;
    chunk = ((chunk + BK - 1) / BK) * BK;
    a.k_chunk = chunk;
    const int eff = (K + chunk - 1) / chunk;
    int rc = launch_gemm<true, false, EPI_DW>(a, eff, stream);      // plain stores of the partial products
    if (rc) return rc;
    const long long total4 = slab / 4;
    long long blocks = (total4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(reduce_fwd_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, workspace, slab, eff, bias, act, keep,
                       seed, step_ptr, Y, N, total4);
    DTF_CHECK_LAUNCH();
    return 0;
}

// dX[M,K] = dZ[M,N] . W[K,N]^T, then (optionally) the producing layer's activation and
// dropout gradient: relu -> (Yprev > 0) * scale_prev, sigmoid -> Yprev (1 - Yprev).
DTF_API int dtf_dense_bwd_dx(const float* dZ, const float* W, float* dX, int M, int N, int K,
                             const float* Yprev, int act_prev, float scale_prev, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    GemmArgs a = {};
    a.A = dZ; a.lda = N;
    a.B = W; a.ldb = N;       // B(k=n, col=kin) = W[kin*N + n]  -> k-contiguous
    a.C = dX; a.ldc = K;
    a.M = M; a.N = K; a.K = N;
    a.k_chunk = ((N + BK - 1) / BK) * BK;
    a.yprev = Yprev; a.ldy = K; a.act = act_prev; a.scale_prev = scale_prev;
    return launch_gemm<true, true, EPI_DX>(a, 1, stream);
}

// wgrad plan: tile and number of batch splits are chosen together.  For every tile the split count that fills the
// CUs' resident-block slots once is priced with the same model as above plus the slab traffic of the split-batch
// reduction (each split writes, and the reducer reads, a K x N slab).
struct DwPlan { int tile; int splits; };

// Workgroups of a wgrad kernel instantiation one CU holds at once, as the runtime computes it from the kernel's
// registers and LDS (the 64x64 gather kernel: 82 registers -> 5 waves per SIMD, although its 17 KB of LDS would
// allow nine).  The split count is chosen to fill exactly these slots: one block more and a second round starts
// (measured on the MNIST CNN's second convolution: 98 splits 807 us, 99 splits 945 us).
template <int BM, int BN, int WM, int WN, bool GATHER>
static int wgrad_resident_blocks() {
    static int cached = 0;
    if (cached) return cached;
    using LoadA = TileLoader<BM, false>;
    using LoadB = TileLoader<BN, false>;
    const size_t lds = (gemm_single_stage(BM, BN, GATHER) ? 1 : 2) * (LoadA::LDS_FLOATS + LoadB::LDS_FLOATS) * sizeof(float);
    auto kern = gemm_f32_kernel<BM, BN, WM, WN, false, false, EPI_DW, GATHER>;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 256, lds) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        n = (int)((160 * 1024) / lds);
        if (n > 4) n = 4;
    }
    cached = n > 8 ? 8 : n;
    return cached;
}

static int wgrad_resident(int tile, bool gather) {
    if (gather) {
        if (tile == 128) return wgrad_resident_blocks<128, 128, 2, 2, true>();
        if (tile == 6428) return wgrad_resident_blocks<64, 128, 2, 2, true>();
        if (tile == 32) return wgrad_resident_blocks<128, 32, 4, 1, true>();
        return wgrad_resident_blocks<64, 64, 2, 2, true>();
    }
    if (tile == 128) return wgrad_resident_blocks<128, 128, 2, 2, false>();
    if (tile == 6428) return wgrad_resident_blocks<64, 128, 2, 2, false>();
    if (tile == 32) return wgrad_resident_blocks<128, 32, 4, 1, false>();
    return wgrad_resident_blocks<64, 64, 2, 2, false>();
}

static DwPlan dw_plan(int M, int N, int K, bool gather = false) {
    if (const char* e = getenv("DTF_DW_PLAN")) {     // tuning aid: "tile,splits"
        int t = 0, sp = 0;
        if (sscanf(e, "%d,%d", &t, &sp) == 2 && sp >= 1) return {t, sp};
    }
    int max_by_m = (M + 8 * BK - 1) / (8 * BK);      // at least 256 rows of batch per split
    if (max_by_m < 1) max_by_m = 1;
    if (split_products() && ((M | N | K) & 3) == 0 && N >= 64 && K >= 64) {     // split kernel: 128x128 only, 2 blocks per CU
        const int tiles = ((K + 127) / 128) * ((N + 127) / 128);
        int want = 512 / tiles;
        if (want < 1) want = 1;
        return {128, want < max_by_m ? want : max_by_m};
    }
    if (N <= 32) {
        const int tiles = ((K + 127) / 128) * ((N + 31) / 32);
        int want = 512 / tiles;
        if (want < 1) want = 1;
        return {32, want < max_by_m ? want : max_by_m};
    }
    const int cand[3][3] = {{128, 128, 128}, {64, 128, 6428}, {64, 64, 64}};
    DwPlan best = {64, 1};
    double best_cost = -1.0;
    for (int c = 0; c < 3; ++c) {
        const int bm = cand[c][0], bn = cand[c][1];
        if ((K <= 64 || N <= 64) && cand[c][2] != 64) continue;
        const long long tiles = (long long)((K + bm - 1) / bm) * ((N + bn - 1) / bn);
        const long long res = wgrad_resident(cand[c][2], gather);
        long long want = (256 * res) / tiles;
        if (want < 1) want = 1;
        int sp = (int)(want < max_by_m ? want : max_by_m);
        int chunk = (((M + sp - 1) / sp + BK - 1) / BK) * BK;
        sp = (M + chunk - 1) / chunk;
        double cost = tile_cost(K, N, chunk, sp, bm, bn, false, false, res);
        if (sp > 1) cost += 2.0 * sp * (double)K * N * 4.0 * 0.0007;     // slab bytes at ~3 TB/s, in 2.1 GHz cycles
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = {cand[c][2], sp}; }
    }
    return best;
}

static int dw_tile(int N, int K, int M, bool gather = false) { return dw_plan(M, N, K, gather).tile; }

// Number of batch splits the wgrad kernel uses for (M,N,K); the workspace must hold
// splits * round_up(K*N + N, 4) floats when splits > 1.
DTF_API int dtf_dense_bwd_dw_splits(int M, int N, int K) { return dw_plan(M, N, K).splits; }
// the same for the implicit-GEMM convolution's weight gradient (its gather kernels hold other numbers of blocks per CU)
DTF_API int dtf_conv_wgrad_splits(int M, int N, int K) { return dw_plan(M, N, K, true).splits; }

// dW[K,N] = X[M,K]^T . dZ[M,N];  db[N] = sum_m dZ[m,:]  (db may be null).
DTF_API int dtf_dense_bwd_dw(const float* X, const float* dZ, float* dW, float* db, int M, int N, int K,
                             float* workspace, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int splits = dtf_dense_bwd_dw_splits(M, N, K);
    const int hint = dw_tile(N, K, M);
    GemmArgs a = {};
    a.A = X; a.lda = K;       // A(i=kin, k=m) = X[m*K + kin] -> i-contiguous
    a.B = dZ; a.ldb = N;
    a.M = K; a.N = N; a.K = M;
    a.ldc = N;
    int chunk = (M + splits - 1) / splits;
    chunk = ((chunk + BK - 1) / BK) * BK;
    a.k_chunk = chunk;
    const int eff_splits = (M + chunk - 1) / chunk;
    static const int colsum_on_mfma = getenv("DTF_COLSUM_MFMA") ? atoi(getenv("DTF_COLSUM_MFMA")) : 1;
    a.colsum_mfma = (colsum_on_mfma && db && K % 32 != 0 && !split_products()) ? 1 : 0;
    if (eff_splits <= 1) {
        a.C = dW; a.c_split_stride = 0; a.colsum = db;
        return launch_gemm<false, false, EPI_DW>(a, 1, stream, hint);
    }
    // slab s = [ dW partial (K*N) | db partial (N) ], rows padded to a multiple of 4 floats
    const long long kn = (long long)K * N;
    const long long stride = ((kn + N + 3) / 4) * 4;
    a.C = workspace; a.c_split_stride = stride;
    a.colsum = db ? workspace + kn : nullptr;
    a.colsum_split_stride = stride;
    int rc = launch_gemm<false, false, EPI_DW>(a, eff_splits, stream, hint);
    if (rc) return rc;
    const long long n_total = db ? kn + N : kn;
    launch_reduce_slabs(workspace, stride, eff_splits, dW, kn, db, n_total, stream);
    DTF_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution (stride 1, NHWC, zero-padded input image prepared by the caller; see conv.hip):
// patch row m lives at rowtab[m] in the padded image, window position w at + shift[w], channels contiguous.

static bool gather_ok(const int* rowtab, const int* shift, int C, int K, bool a_kc) {
    return rowtab && shift && C > 0 && K % C == 0 && C % 4 == 0;
}

// Y[M,N] = act(patches(Xpad)[M,K] . W[K,N] + b), K = KH*KW*C
DTF_API int dtf_conv_gemm_fwd(const float* Xpad, long long xpad_elems, const int* rowtab, const int* shift, int C,
                              const float* W, const float* bias, float* Y, int M, int N, int K, int act,
                              const unsigned* masktab, unsigned char* pool_arg, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if (!gather_ok(rowtab, shift, C, K, true)) return -3;
    if (pool_arg && (M % 4 != 0 || (act != ACT_NONE && act != ACT_RELU))) return -3;
    GemmArgs a = {};
    a.A = Xpad; a.lda = 0; a.B = W; a.ldb = N; a.C = Y; a.ldc = N;
    a.M = M; a.N = N; a.K = K; a.k_chunk = K;
    a.bias = bias; a.act = act; a.keep = 1.0f;
    a.g_rowtab = rowtab; a.g_shift = shift; a.g_rows = M; a.g_C = C; a.g_elems = xpad_elems; a.g_masktab = masktab;
    a.pool_arg = pool_arg;
    return launch_gather<true, false, EPI_FWD>(a, 1, stream);
}

// dX[M,N] = (patches(dZpad)[M,K] . Wflip[K,N]) * act'(Yprev): the data gradient of a stride-1 convolution is the
// convolution of the zero-padded output gradient with the spatially flipped, channel-transposed filter.
// K = KH*KW*Cout, N = Cin.
DTF_API int dtf_conv_gemm_dgrad(const float* dZpad, long long dzpad_elems, const int* rowtab, const int* shift, int Cout,
                                const float* Wflip, float* dX, int M, int N, int K, const float* Yprev, int act_prev,
                                const unsigned* masktab, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if (!gather_ok(rowtab, shift, Cout, K, true)) return -3;
    GemmArgs a = {};
    a.A = dZpad; a.B = Wflip; a.ldb = N; a.C = dX; a.ldc = N;
    a.M = M; a.N = N; a.K = K; a.k_chunk = K;
    a.yprev = Yprev; a.ldy = N; a.act = act_prev; a.scale_prev = 1.0f;
    a.g_rowtab = rowtab; a.g_shift = shift; a.g_rows = M; a.g_C = Cout; a.g_elems = dzpad_elems; a.g_masktab = masktab;
    return launch_gather<true, false, EPI_DX>(a, 1, stream);
}

// dW[K,N] = patches(Xpad)[M,K]^T . dZ[M,N], db = colsum(dZ); workspace as for dtf_dense_bwd_dw(M, N, K)
DTF_API int dtf_conv_gemm_wgrad(const float* Xpad, long long xpad_elems, const int* rowtab, const int* shift, int C,
                                const float* dZ, float* dW, float* db, int M, int N, int K, float* workspace,
                                const unsigned* masktab, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if (!gather_ok(rowtab, shift, C, K, false)) return -3;
    const int splits = dtf_conv_wgrad_splits(M, N, K);
    const int hint = dw_tile(N, K, M, true);
    GemmArgs a = {};
    a.A = Xpad; a.B = dZ; a.ldb = N;
    a.M = K; a.N = N; a.K = M; a.ldc = N;
    a.g_rowtab = rowtab; a.g_shift = shift; a.g_rows = M; a.g_C = C; a.g_elems = xpad_elems; a.g_masktab = masktab;
    int chunk = (M + splits - 1) / splits;
    chunk = ((chunk + BK - 1) / BK) * BK;
    a.k_chunk = chunk;
    const int eff_splits = (M + chunk - 1) / chunk;
    if (eff_splits <= 1) {
        a.C = dW; a.c_split_stride = 0; a.colsum = db;
        return launch_gather<false, false, EPI_DW>(a, 1, stream, hint);
    }
    const long long kn = (long long)K * N;
    const long long stride = ((kn + N + 3) / 4) * 4;
    a.C = workspace; a.c_split_stride = stride;
    a.colsum = db ? workspace + kn : nullptr;
    a.colsum_split_stride = stride;
    int rc = launch_gather<false, false, EPI_DW>(a, eff_splits, stream, hint);
    if (rc) return rc;
    const long long n_total = db ? kn + N : kn;
    launch_reduce_slabs(workspace, stride, eff_splits, dW, kn, db, n_total, stream);
    DTF_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// bf16-operand products (see gemm_bf16_kernel).  Shapes must give the kernel whole float4 vectors; products too small
// to fill the chip with 128x128 tiles are refused (-5) and the caller stays on the fp32 kernels.
static bool bf16_shape_ok(long long rows, long long cols, int k) {
    return cols % 4 == 0 && k % 4 == 0 && ((rows + 127) / 128) * ((cols + 127) / 128) >= 128;
}

DTF_API int dtf_dense_bf16_supported(int M, int N, int K) {
    return (bf16_shape_ok(M, N, K) && bf16_shape_ok(M, K, N) && bf16_shape_ok(K, N, M)) ? 1 : 0;
}

DTF_API int dtf_dense_fwd_bf16(const float* X, const float* W, const float* bias, float* Y, int M, int N, int K, int act,
                               float keep, uint32_t seed, const long long* step_ptr, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if (!bf16_shape_ok(M, N, K)) return -5;
    GemmArgs a = {};
    a.A = X; a.lda = K; a.B = W; a.ldb = N; a.C = Y; a.ldc = N;
    a.M = M; a.N = N; a.K = K; a.k_chunk = ((K + BK - 1) / BK) * BK;
    a.bias = bias; a.act = act; a.keep = keep; a.seed = seed; a.step_ptr = step_ptr;
    return launch_bf16<true, false, EPI_FWD>(a, stream);
}

DTF_API int dtf_dense_bwd_dx_bf16(const float* dZ, const float* W, float* dX, int M, int N, int K, const float* Yprev,
                                  int act_prev, float scale_prev, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if (!bf16_shape_ok(M, K, N)) return -5;
    GemmArgs a = {};
    a.A = dZ; a.lda = N; a.B = W; a.ldb = N; a.C = dX; a.ldc = K;
    a.M = M; a.N = K; a.K = N; a.k_chunk = ((N + BK - 1) / BK) * BK;
    a.yprev = Yprev; a.ldy = K; a.act = act_prev; a.scale_prev = scale_prev;
    return launch_bf16<true, true, EPI_DX>(a, stream);
}

DTF_API int dtf_dense_bwd_dw_bf16(const float* X, const float* dZ, float* dW, float* db, int M, int N, int K,
                                  hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if (!bf16_shape_ok(K, N, M)) return -5;
    GemmArgs a = {};
    a.A = X; a.lda = K; a.B = dZ; a.ldb = N; a.C = dW; a.ldc = N;
    a.M = K; a.N = N; a.K = M; a.k_chunk = ((M + BK - 1) / BK) * BK;
    int rc = launch_bf16<false, false, EPI_DW>(a, stream);
    if (rc) return rc;
    if (db) {                                        // the bias gradient stays an fp32 sum of the unrounded dZ
        if (hipMemsetAsync(db, 0, (size_t)N * sizeof(float), stream) != hipSuccess) return -6;
        const int rows_per_block = 256;
        dim3 grid((N / 4 + 255) / 256, (M + rows_per_block - 1) / rows_per_block);
        hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, stream, dZ, db, M, N, rows_per_block);
        DTF_CHECK_LAUNCH();
    }
    return 0;
}
