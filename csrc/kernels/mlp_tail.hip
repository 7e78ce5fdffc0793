// Fused "network tail": everything after the first (wide-input) layer of an MLP classifier in one kernel.
//
//   h1 [M,128] (first layer's output, after its ReLU and dropout)
//     -> h2 = dropout(relu(h1 W2 + b2))            [M,128]
//     -> z  = h2 W3 + b3                           [M,C], C <= 16
//     -> softmax cross-entropy vs. sparse labels    loss / accuracy statistics, dz3
//     -> dW3, db3, dz2 = (dz3 W3^T) * relu'(h2) * dropout
//     -> dW2, db2, dz1 = (dz2 W2^T) * relu'(h1) * dropout
//
// With 128-wide hidden layers each of these GEMMs is a few microseconds of MFMA work; as separate kernels
// (three forward, loss, two dgrad, two wgrad, two slab reductions) their launch / fill / drain costs dominate and
// h2, z, dz3, dz2 make round trips through HBM.  Here a workgroup owns 64-row slabs of the batch: a slab's
// activations never leave LDS, W2 and W3 are loaded into LDS once per workgroup, the weight gradients
// accumulate in registers across the workgroup's slabs and leave as one partial slab per workgroup, and the only
// per-row HBM traffic is h1 in and dz1 out.  Dropout masks are the counter-based function of
// (seed, global_step, row, column) the layer-wise kernels use, so both paths draw identical masks.
#include "lds_mma.h"

namespace {

constexpr int NT = 512;                 // 8 waves, two per SIMD
constexpr int BM = 64;                  // rows per slab
constexpr int H = 128;                  // hidden width of both tail layers
constexpr int CP = 16;                  // classes padded
constexpr int LH = H + 4, LC = CP + 4;  // LDS row strides (floats), stride/4 odd
constexpr int SW3_FLOATS = H * LC + 32; // + slack: 32-lane reads of a 20-float row run past the last row
constexpr int LDS_FLOATS = 2 * BM * LH + H * LH + SW3_FLOATS + 2 * BM * LC + 32 /*slack*/ + H + CP + 32;
static_assert(LDS_FLOATS * 4 <= 160 * 1024, "slab + weights must fit one CU's LDS");

struct TailArgs {
    const float* H1;
    const float* W2;
    const float* b2;
    const float* W3;
    const float* b3;
    const long long* labels;
    float* dZ1;
    float* slabs;            // [gridDim.x][slab_stride]: dW2 | db2 | dW3 | db3
    long long slab_stride;
    int M, C;
    float keep2;
    uint32_t seed2;
    const long long* step_ptr;
    float scale1;            // 1 / keep-probability of the first layer's dropout
    float grad_scale;        // 1 / global batch
    float* stats;            // [>=3]: loss sum, correct count, example count
    long long* prof;         // [32] shader-clock stamps of workgroup 0's second slab (steady state), or null (profiling aid)
};

#define TAIL_PROF(i)                                                                                    \
    do {                                                                                                \
        if (p.prof && tid0 == 0 && slab == (nslabs > (int)gridDim.x ? (int)gridDim.x : 0)) p.prof[i] = (long long)__builtin_readcyclecounter();      \
    } while (0)

__global__ __launch_bounds__(NT, 1) void mlp_tail_kernel(const TailArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sH1 = lds;
    float* sH2 = sH1 + BM * LH;
    float* sW2 = sH2 + BM * LH;
    float* sW3 = sW2 + H * LH;
    float* sPa = sW3 + SW3_FLOATS;       // logits k-half 0, then dz3
    float* sPb = sPa + BM * LC;          // logits k-half 1
    float* sB2 = sPb + BM * LC + 32;
    float* sB3 = sB2 + H;
    float* sRed = sB3 + CP;

    const int tid0 = threadIdx.x;
    int tid = tid0;
    const int C = p.C;
    const bool drop2 = p.keep2 < 1.0f;
    const float scale2 = drop2 ? 1.0f / p.keep2 : 1.0f;
    const uint32_t key2 = dtf_dropout_key(p.seed2, p.step_ptr);
    const uint32_t thr2 = dtf_keep_threshold(p.keep2);
    const bool idx32 = (long long)p.M * H < (1ll << 32);      // every element index of h2 fits 32 bits
    const int nslabs = (p.M + BM - 1) / BM;

    // ---- once per workgroup: weights and biases into LDS
    {
        LdsStage<NT, H, H> w2;
        w2.template issue<false>(p.W2, (unsigned)(H * H * 4), H, tid);
        for (int e = tid; e < SW3_FLOATS; e += NT) {
            const int k = e / LC, c = e - k * LC;
            sW3[e] = (k < H && c < C) ? p.W3[k * C + c] : 0.f;
        }
        if (tid < H) sB2[tid] = p.b2[tid];
        if (tid < CP) sB3[tid] = tid < C ? p.b3[tid] : 0.f;
        for (int e = tid; e < 2 * BM * LC + 32; e += NT) sPa[e] = 0.f;
        w2.commit(sW2, tid);
    }
    if (p.prof && blockIdx.x == 0 && (tid0 & 63) == 0)      // which SIMD each wave of workgroup 0 landed on (HW_ID[5:4])
        p.prof[16 + (tid0 >> 6)] = (long long)((__builtin_amdgcn_s_getreg(0xF804) >> 4) & 3);
    LdsStage<NT, BM, H> ph;
    int slab = blockIdx.x;
    if (slab < nslabs) {
        const int rows = min(BM, p.M - slab * BM);
        ph.template issue<false>(p.H1 + (long long)slab * BM * H, (unsigned)rows * H * 4, H, tid);
    }
    f32x16 g2[2];
    f32x4 g3 = {0.f, 0.f, 0.f, 0.f};     // dW3 rows 16 * wave .. + 15 (C/D layout of the 16x16 MFMA)
    lds_zero(g2[0]); lds_zero(g2[1]);
    float db2r = 0.f, db3r = 0.f, loss = 0.f, correct = 0.f, count = 0.f;

    for (; slab < nslabs; slab += gridDim.x) {
        // The thread index is made opaque once per slab: the addresses below are loop-invariant and hoisting them
        // all would cost more registers than recomputing them.
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int li = lane & 31, h = lane >> 5;
        const int tm = wave >> 2, tn = wave & 3;
        const int col = tn * 32 + li;
        const int row0 = slab * BM;
        const int rows = min(BM, p.M - row0);

        TAIL_PROF(0);
        // this thread's softmax row label, requested now so that the HBM round trip passes under the layer-2 products
        const int label = ((tid >> 3) < rows) ? (int)p.labels[row0 + (tid >> 3)] : 0;
        ph.commit(sH1, tid);
        __syncthreads();
        TAIL_PROF(1);
        if (slab + (int)gridDim.x < nslabs) {
            const int nslab = slab + gridDim.x;
            const int nrows = min(BM, p.M - nslab * BM);
            ph.template issue<false>(p.H1 + (long long)nslab * BM * H, (unsigned)nrows * H * 4, H, tid);
        }
        // ---- h2 = dropout(relu(h1 W2 + b2)): one 32x32 tile per wave.  The dropout decision of output element g
        //      (a counter hash of its index) is computed between the MFMAs of k-group g: H / 8 = 16 groups, 16 elements.
        {
            static_assert(H / 8 == 16, "one k-group per accumulator register");
            f32x16 acc; lds_zero(acc);
            float mult[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) mult[r] = 1.0f;
            if (drop2 && idx32) {
                // the hashing is what bounds this phase (vector issue, not the matrix pipe): 32-bit indices and an
                // integer threshold keep it to a dozen instructions per element
                const uint32_t e0 = (uint32_t)(row0 + tm * 32 + 4 * h) * (uint32_t)H + (uint32_t)col;
                lds_tile_mma_fill<true, false, H, 14>(acc, sH1, LH, tm * 32, sW2, LH, tn * 32, lane, [&](int g) {
                    mult[g] = dtf_keep32(key2, e0 + (uint32_t)(((g & 3) + 8 * (g >> 2)) * H), thr2) ? scale2 : 0.f;
                });
            } else if (drop2) {
                const unsigned long long e0 = (unsigned long long)(row0 + tm * 32) * (unsigned)H + col;
                lds_tile_mma_fill<true, false, H, 30>(acc, sH1, LH, tm * 32, sW2, LH, tn * 32, lane, [&](int g) {
                    mult[g] = dtf_keep64(key2, e0 + (unsigned long long)lds_tile_row(g, h) * (unsigned)H, thr2) ? scale2 : 0.f;
                });
            } else {
                lds_tile_mma<true, false, H>(acc, sH1, LH, tm * 32, sW2, LH, tn * 32, lane);
            }
            TAIL_PROF(9);
            if (p.prof && blockIdx.x == 0 && lane == 0 && slab == (nslabs > (int)gridDim.x ? (int)gridDim.x : 0))
                p.prof[24 + wave] = (long long)__builtin_readcyclecounter();     // every wave's end of the h2 products
            const float bv = sB2[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tm * 32 + lds_tile_row(r, h);
                sH2[row * LH + col] = fmaxf(acc[r] + bv, 0.f) * mult[r];
            }
            TAIL_PROF(10);
        }
        __syncthreads();
        TAIL_PROF(2);
        // ---- logits: the padded class dimension is 16 wide, so the products run as 16x16 tiles (half the matrix-pipe
        //      time of 32x32 tiles with 16 dead columns): four row tiles x two k-halves on the eight waves; the halves
        //      meet in the softmax pass (fixed order)
        {
            static_assert(CP == 16 && BM == 64 && NT == 512, "eight 16x16 logit jobs, one per wave");
            const int rt = wave & 3, kh = wave >> 2;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            lds_tile16_mma<true, false, 64>(acc, sH2 + kh * 64, LH, rt * 16, sW3 + kh * 64 * LC, LC, 0, lane);
            float* dst = kh ? sPb : sPa;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(rt * 16 + 4 * (lane >> 4) + r) * LC + (lane & 15)] = acc[r];
        }
        __syncthreads();
        TAIL_PROF(3);
        // ---- softmax cross-entropy: eight threads per row, two classes each (the row's maximum, sum and label term
        //      meet through lane exchanges); dz3 replaces the logits (zero in rows past the batch and in the padded
        //      classes)
        {
            static_assert(CP == 16 && NT == 8 * BM, "eight threads x two classes cover one row of padded logits");
            const int row = tid >> 3, c0 = 2 * (tid & 7);
            float* z = sPa + row * LC;
            if (row < rows) {
                const float* zb = sPb + row * LC;
                const bool k0 = c0 < C, k1 = c0 + 1 < C;
                const float v0 = k0 ? z[c0] + zb[c0] + sB3[c0] : -INFINITY;
                const float v1 = k1 ? z[c0 + 1] + zb[c0 + 1] + sB3[c0 + 1] : -INFINITY;
                float mx = v0;
                int arg = c0;
                if (v1 > v0) { mx = v1; arg = c0 + 1; }
#pragma unroll
                for (int d = 1; d < 8; d <<= 1) {            // first maximum wins, as in a left-to-right scan
                    const float mo = __shfl_xor(mx, d);
                    const int ao = __shfl_xor(arg, d);
                    if (mo > mx || (mo == mx && ao < arg)) { mx = mo; arg = ao; }
                }
                const float e0 = k0 ? __expf(v0 - mx) : 0.f, e1 = k1 ? __expf(v1 - mx) : 0.f;
                float se = e0 + e1;
                float zl = (c0 == label) ? e0 : ((c0 + 1 == label) ? e1 : 0.f);
#pragma unroll
                for (int d = 1; d < 8; d <<= 1) {
                    se += __shfl_xor(se, d);
                    zl += __shfl_xor(zl, d);
                }
                const float inv = 1.0f / se;
                z[c0] = k0 ? (e0 * inv - (c0 == label ? 1.f : 0.f)) * p.grad_scale : 0.f;
                z[c0 + 1] = k1 ? (e1 * inv - (c0 + 1 == label ? 1.f : 0.f)) * p.grad_scale : 0.f;
                if ((tid & 7) == 0) {
                    loss += -__logf(zl * inv);
                    correct += (arg == label) ? 1.f : 0.f;
                    count += 1.f;
                }
            } else {
                z[c0] = 0.f;
                z[c0 + 1] = 0.f;
            }
        }
        __syncthreads();
        TAIL_PROF(4);
        // ---- dW3 += h2^T dz3 (16x16 tiles: hidden units 16 * wave .. + 15 on every wave); dz2 products; both read h2 / dz3
        static_assert(H == 16 * (NT / 64), "one 16-unit block of dW3 per wave");
        lds_tile16_mma<false, false, BM>(g3, sH2, LH, wave * 16, sPa, LC, 0, lane);
        f32x16 az; lds_zero(az);
        lds_tile_mma<true, true, CP>(az, sPa, LC, tm * 32, sW3, LC, tn * 32, lane);
        {   // db3: thread = (class tid & 15, row pair tid >> 4); the partial sums meet after the last slab
            static_assert(NT == CP * (BM / 2), "one row pair per thread");
            const float* z = sPa + (2 * (tid >> 4)) * LC + (tid & 15);
            db3r += z[0] + z[LC];
        }
        __syncthreads();
        TAIL_PROF(5);
        // ---- dz2 = (dz3 W3^T) * (h2 > 0) / keep2 in place of h2
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tm * 32 + lds_tile_row(r, h);
            const float y = sH2[row * LH + col];
            sH2[row * LH + col] = (y > 0.f) ? az[r] * scale2 : 0.f;
        }
        __syncthreads();
        TAIL_PROF(6);
        // ---- dW2 += h1^T dz2: tiles t = wave, wave + 8 (input block t >> 2, output block t & 3); db2
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int t = wave + 8 * q;
            lds_tile_mma<false, false, BM>(g2[q], sH1, LH, (t >> 2) * 32, sH2, LH, (t & 3) * 32, lane);
        }
        {   // db2: thread = (unit tid & 127, row quarter tid >> 7)
            static_assert(NT == 4 * H && BM % 4 == 0, "four row quarters per hidden unit");
            const float* z = sH2 + ((tid >> 7) * (BM / 4)) * LH + (tid & (H - 1));
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int r = 0; r < BM / 4; r += 2) {
                s0 += z[r * LH];
                s1 += z[(r + 1) * LH];
            }
            db2r += s0 + s1;
        }
        TAIL_PROF(7);
        // ---- dz1 = (dz2 W2^T) * (h1 > 0) / keep1 -> HBM
        {
            f32x16 a1; lds_zero(a1);
            lds_tile_mma<true, true, H>(a1, sH2, LH, tm * 32, sW2, LH, tn * 32, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tm * 32 + lds_tile_row(r, h);
                if (row < rows) {
                    const float y = sH1[row * LH + col];
                    p.dZ1[(long long)(row0 + row) * H + col] = (y > 0.f) ? a1[r] * p.scale1 : 0.f;
                }
            }
        }
        __syncthreads();     // the next slab overwrites h1 / h2
        TAIL_PROF(8);
    }

    // ---- this workgroup's partial gradients
    {
        const int lane = tid0 & 63, wave = tid0 >> 6, li = lane & 31, h = lane >> 5;
        float* base = p.slabs + (long long)blockIdx.x * p.slab_stride;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int t = wave + 8 * q;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                base[((t >> 2) * 32 + lds_tile_row(r, h)) * H + (t & 3) * 32 + li] = g2[q][r];
        }
        sH1[tid0] = db2r;            // the slab buffers are free now: partial bias sums meet here
        sH1[NT + tid0] = db3r;
        if ((lane & 15) < C) {
#pragma unroll
            for (int r = 0; r < 4; ++r) base[H * H + H + (wave * 16 + 4 * (lane >> 4) + r) * C + (lane & 15)] = g3[r];
        }
        // loss / correct / example counts ride in the slab too: thousands of atomic adds on three addresses from
        // every wave of the grid would serialise in L2 for longer than the kernel runs
        const float l = dtf_wave_sum(loss), a = dtf_wave_sum(correct), n = dtf_wave_sum(count);
        if (lane == 0) {
            sRed[wave * 3 + 0] = l;
            sRed[wave * 3 + 1] = a;
            sRed[wave * 3 + 2] = n;
        }
        __syncthreads();
        if (tid0 < 3) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) t += sRed[w * 3 + tid0];
            base[H * H + H + H * C + C + tid0] = t;
        }
        if (tid0 < H) base[H * H + tid0] = (sH1[tid0] + sH1[H + tid0]) + (sH1[2 * H + tid0] + sH1[3 * H + tid0]);
        if (tid0 >= NT - CP && tid0 - (NT - CP) < C) {       // a different wave than the db2 sums
            const int c = tid0 - (NT - CP);
            float t = 0.f;
#pragma unroll 8
            for (int k = 0; k < BM / 2; ++k) t += sH1[NT + k * CP + c];
            base[H * H + H + H * C + c] = t;
        }
    }
}

// out[i] = sum over the workgroups' slabs, scattered to the four gradient tensors.  A 1024-thread block covers 64
// float4 columns with 16 slab-groups (one wave each); the 16 partial sums meet in LDS.
__global__ __launch_bounds__(1024) void tail_reduce_kernel(const float* __restrict__ slabs, long long slab_stride, int nslabs,
                                                           float* __restrict__ dW2, float* __restrict__ db2,
                                                           float* __restrict__ dW3, float* __restrict__ db3,
                                                           float* __restrict__ stats, int C) {
    __shared__ f32x4 part[16][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int n_total = H * H + H + H * C + C + 3;       // ... + loss sum, correct count, example count
    const int i = (blockIdx.x * 64 + c) * 4;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    if (i < n_total) {                        // slab_stride is a multiple of 4 and >= n_total rounded up: in bounds
        int s = g;
        for (; s + 16 < nslabs; s += 32) {
            a0 += *reinterpret_cast<const f32x4*>(slabs + (long long)s * slab_stride + i);
            a1 += *reinterpret_cast<const f32x4*>(slabs + (long long)(s + 16) * slab_stride + i);
        }
        for (; s < nslabs; s += 16) a0 += *reinterpret_cast<const f32x4*>(slabs + (long long)s * slab_stride + i);
    }
    part[g][c] = a0 + a1;
    __syncthreads();
    if (g == 0 && i < n_total) {
        f32x4 t = part[0][c];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += part[k][c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int e = i + j;
            if (e >= n_total) break;
            if (e < H * H) { dW2[e] = t[j]; continue; }
            e -= H * H;
            if (e < H) { db2[e] = t[j]; continue; }
            e -= H;
            if (e < H * C) { dW3[e] = t[j]; continue; }
            e -= H * C;
            if (e < C) { db3[e] = t[j]; continue; }
            if (stats) stats[e - C] += t[j];              // running totals of the epoch; this thread alone owns them here
        }
    }
}

inline long long slab_stride_for(int C) {
    const long long n = (long long)H * H + H + (long long)H * C + C + 3;
    return (n + 3) / 4 * 4;
}
inline int grid_for(int M) {
    const int nslabs = (M + BM - 1) / BM;
    return nslabs < 256 ? nslabs : 256;     // one persistent workgroup per CU
}

}  // namespace

long long* g_tail_prof = nullptr;

// Profiling aid: nine shader-clock stamps of workgroup 0's first slab go to `device_buffer` (null turns it off).
DTF_API int dtf_mlp_tail_set_profile_buffer(long long* device_buffer) {
    g_tail_prof = device_buffer;
    return 0;
}

DTF_API int dtf_mlp_tail_supported(int H1, int H2, int C) { return (H1 == H && H2 == H && C >= 1 && C <= CP) ? 1 : 0; }

DTF_API long long dtf_mlp_tail_workspace_floats(int M, int C) { return (long long)grid_for(M) * slab_stride_for(C); }

// Forward from h1 through the loss and backward to dz1 / dW2 / db2 / dW3 / db3 (see the file header).
DTF_API int dtf_mlp_tail(const float* H1, const float* W2, const float* b2, const float* W3, const float* b3,
                         const long long* labels, float* dZ1, float* dW2, float* db2, float* dW3, float* db3,
                         float* workspace, long long workspace_floats, int M, int H1dim, int H2dim, int C, float keep2,
                         uint32_t seed2, const long long* step_ptr, float scale1, float grad_scale, float* stats,
                         hipStream_t stream) {
    if (!dtf_mlp_tail_supported(H1dim, H2dim, C)) return -4;
    if (M <= 0) return 0;
    const int grid = grid_for(M);
    const long long stride = slab_stride_for(C);
    if (workspace_floats < (long long)grid * stride) return -7;
    TailArgs a = {};
    a.H1 = H1; a.W2 = W2; a.b2 = b2; a.W3 = W3; a.b3 = b3; a.labels = labels; a.dZ1 = dZ1;
    a.slabs = workspace; a.slab_stride = stride; a.M = M; a.C = C; a.keep2 = keep2; a.seed2 = seed2;
    a.step_ptr = step_ptr; a.scale1 = scale1; a.grad_scale = grad_scale; a.stats = stats; a.prof = g_tail_prof;
    constexpr size_t lds = sizeof(float) * LDS_FLOATS;
    static bool once = false;
    if (!once) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_tail_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        once = true;
    }
    hipLaunchKernelGGL(mlp_tail_kernel, dim3(grid), dim3(NT), lds, stream, a);
    DTF_CHECK_LAUNCH();
    const int n_total = H * H + H + H * C + C + 3;
    hipLaunchKernelGGL(tail_reduce_kernel, dim3((n_total + 255) / 256), dim3(1024), 0, stream, workspace, stride, grid, dW2,
                       db2, dW3, db3, stats, C);
    DTF_CHECK_LAUNCH();
    return 0;
}
