// Whole-training-step kernel for the reference network at small batch:
// 64 -> 128 relu -> dropout -> 128 relu -> dropout -> 32 sigmoid, MSE, Adam, batch 50
// (/root/reference/example.py:149-170,207-213).
//
// At batch 50 a step is ~9 MFLOP: a chain of layer kernels is pure launch latency (eleven
// dependent launches at ~1.5 us per boundary plus their fill/drain).  Here ONE workgroup of eight
// waves (two per SIMD, so one wave's LDS / VALU / memory waits hide under the other's MFMAs) on
// one CU runs forward, loss, backward and the optimizer update for `nsteps` consecutive
// mini-batches in a single launch:
//   * activations live in LDS for the whole step (x, h1, h2, p; dz2/dz1 overwrite h2/h1 in place);
//   * weights stream L2 -> registers -> LDS in chunks, each chunk's loads issued one phase ahead so
//     their latency hides under the previous phase's MFMAs; W1 never leaves LDS between steps (the
//     W1 update writes its LDS image as well);
//   * every GEMM runs on v_mfma_f32_32x32x2_f32 from LDS operands with the same k-permuted 16-byte
//     fragment reads as the big GEMM kernel; each wave owns one 32x32 output tile per phase;
//   * weight gradients never leave the accumulators: the wgrad epilogue applies Adam (or SGD)
//     straight to the variable and its slots, whose old values were requested before the tile's
//     MFMA loop.  Every layer's gradients are taken before that layer's update is visible
//     (TensorFlow's compute-then-apply order);
//   * loss / accuracy accumulate in registers and reach the statistics record once per launch;
//   * dropout masks are the same counter-based function of (seed, global_step, index) as in the
//     layer-wise kernels, so both paths draw identical masks; all sums have a fixed order.
// Variables are read with sc1 (L2-served) loads: the CU re-reads values it stored one step earlier.
//
// Asynchronous workers inside one GPU (MULTI instantiation).  The reference's training mode is several workers
// updating shared variables without locks or barriers, with one shared global_step as the budget
// (/root/reference/example.py:136-139,168-170,187: plain AdamOptimizer, use_locking=False).  The same semantics
// on one chip: up to `workers` workgroups, one CU each, run this step loop at once against the SAME variables and
// Adam slots in HBM/L2.  A step is claimed with one returning atomic add on a ticket word (step number = ticket,
// mini-batch = ticket, so every mini-batch is consumed exactly once and the launch ends after `nsteps` steps in
// total); W1 and the biases are re-read from L2 at the start of every step because other workers have written them;
// updates are plain read-modify-write (a lost update is the reference's own race).  All workers must share one
// L2 for their sc1 loads to see each other's stores, and the per-XCD L2s are not coherent: the first workgroup
// to arrive elects its XCD with a compare-and-swap, workgroups on other XCDs leave at once, the first `workers`
// arrivals on the elected XCD work.  Eight times as many workgroups as workers are launched (they are dealt
// round-robin over the XCDs); whatever the placement turns out to be, at least the electing workgroup works, so
// placement decides speed, never the result.
#include "common.h"
#include "lds_mma.h"

namespace {

typedef unsigned int u32x4 __attribute__((vector_size(16)));

constexpr int NT = 512;                      // threads: 8 waves
constexpr int BM = 64;                       // batch rows held in LDS (batch <= 64)
constexpr int D0 = 64, D1 = 128, D2 = 128, D3 = 32;
constexpr int LX = D0 + 4, L1 = D1 + 4, L2 = D2 + 4, L3 = D3 + 4;   // LDS row strides (floats): stride/4 odd
constexpr int LT = 64 + 4;                   // row stride of a 64-column slice of W2
// Staging area images: W1 / a 64-row chunk of W2 as [64][132]; W3 as [128][36] followed by a [64][36] partial-sum
// scratch; a 64-column slice of W2 as [128][68].
constexpr int SW_FLOATS = 128 * LT;
constexpr int W3_SCRATCH = D2 * L3;
static_assert(64 * L2 <= SW_FLOATS && W3_SCRATCH + BM * L3 <= SW_FLOATS, "staging images must fit");
constexpr int LDS_FLOATS = BM * LX + BM * L3 + BM * L1 + BM * L2 + BM * L3 + SW_FLOATS + 32 + D1 + D2 + D3;
static_assert(LDS_FLOATS * 4 <= 160 * 1024, "activations + weight staging must fit one CU's LDS");

struct FusedArgs {
    const float* X;      // [nsteps * B, D0]
    const float* Y;      // [nsteps * B, D3]
    float* params;       // flat variable buffer
    float* slot_m;       // Adam m (null for SGD)
    float* slot_v;       // Adam v
    long long off[6];    // element offsets of W1, b1, W2, b2, W3, b3 in the flat buffers
    int B;
    int nsteps;
    int training;        // 1: dropout on
    int want_cat;        // also count argmax(p) == argmax(y) rows into stats[3] (Keras 'accuracy' on a wide output)
    float keep1, keep2;
    uint32_t seed1, seed2;
    float lr, beta1, beta2, eps;
    long long* step_ptr;
    float* beta_powers;  // [2] or null
    float* stats;        // [4] running statistics or null
    long long* prof;     // [16] shader-clock timestamps of the last step's phases, or null (profiling aid)
    unsigned int* ctl;   // MULTI: {elected XCD (0xFFFFFFFF = none yet), workers admitted, workgroups left, step tickets}
    int workers;         // MULTI: workgroups that may train at once
    int stagger;         // MULTI: worker number i starts i * stagger ticks of the 100 MHz wall clock late
    // Gradient-only launches (OPT 2) of a worker whose variables live with parameter-server tasks: variable i is read
    // from var_ptr[i] and its gradient written to grad_ptr[i] - peer-mapped memory of the owning task (its variable
    // plane, this worker's inbox slot there) - and the launch ends by ringing the owners' doorbells, so a worker's
    // whole step is this one kernel: no pull, no push, no copy.  Null pointers: params + off[i] / slot_m + off[i].
    const float* var_ptr[6];
    float* grad_ptr[6];
    unsigned long long* ring_rep[64];        // every rank's replica of the control blocks, as this GPU addresses it
    int ring_world, ring_self;               // number of replicas; the one in this rank's own memory
    long long ring_block_word[8];            // owner i: position of its block inside a replica (64-bit words)
    long long ring_lane_word[8];             // owner i: position of this worker's lane in that block, inside a replica
    int ring_count, ring_slot;
    unsigned long long ring_seq, ring_clock;
};

// Phase timestamp i of the launch's last step (thread 0 only; a scalar branch when profiling is off).
#define DTF_PROF(i)                                                                                        \
    do {                                                                                                   \
        if (p.prof && tid0 == 0 && s == p.nsteps - 1) p.prof[i] = (long long)__builtin_readcyclecounter(); \
    } while (0)

__device__ __forceinline__ float ld1_sc1(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld1_sys(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- global -> register -> LDS staging -------------------------------------------------------
// A [ROWS x COLS] block of a row-major matrix is ROWS*COLS/4 16-byte vectors; thread `tid` owns vectors tid,
// tid+NT, ...  Loads go through a buffer resource of `bytes` bytes: reads past it return zero, which pads the batch
// to BM rows without a branch.
template <int ROWS, int COLS>
struct Stage {
    static constexpr int NV = ROWS * COLS / 4 / NT;
    static_assert(NV * NT * 4 == ROWS * COLS, "block must split evenly over the workgroup");
    f32x4 v[NV];

    // `ld`: row stride of the source in floats; AUX: cache-policy bits of the load (0 plain, 16 = sc1: served by this
    // GPU's L2, 17 = sc0 sc1: system scope, for variables that live in a peer's memory)
    template <int AUX>
    __device__ __forceinline__ void issue(const float* base, unsigned bytes, int ld, int tid) {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + NT * i;
            const int row = idx / (COLS / 4), c = (idx % (COLS / 4)) * 4;
            const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(r, (row * ld + c) * 4, 0, AUX);
            v[i] = __builtin_bit_cast(f32x4, raw);
        }
    }
    // dst is [ROWS][COLS + 4]
    __device__ __forceinline__ void commit(float* dst, int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + NT * i;
            const int row = idx / (COLS / 4), c = (idx % (COLS / 4)) * 4;
            *reinterpret_cast<f32x4*>(dst + row * (COLS + 4) + c) = v[i];
        }
    }
};

// One 32x32 output tile: acc += sum_k A(m,k) B(k,n), both operands in LDS (software-pipelined fragment reads).
template <bool A_KC, bool B_KC, int K>
__device__ __forceinline__ void tile_mma(f32x16& acc, const float* A, int lda, int m0, const float* B, int ldb, int n0,
                                         int lane) {
    lds_tile_mma<A_KC, B_KC, K>(acc, A, lda, m0, B, ldb, n0, lane);
}

__device__ __forceinline__ void zero(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

__device__ __forceinline__ int tile_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

struct Opt {
    float lr, lr_t, b1, b2, eps;
};

// TF's Adam: var -= lr_t * m / (sqrt(v) + eps).  sqrt and reciprocal use the hardware approximations (1 ulp each).
template <int OPT>
__device__ __forceinline__ float opt_update(const Opt& o, float w, float& m, float& v, float g) {
    if (OPT == 0) {
        m = o.b1 * m + (1.0f - o.b1) * g;
        v = o.b2 * v + (1.0f - o.b2) * g * g;
        return w - o.lr_t * m * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + o.eps);
    }
    return w - o.lr * g;
}

// The flat variable buffer and its slot buffers as buffer resources: 32-bit byte offsets instead of 64-bit
// pointers per element (half the address registers and address arithmetic of flat accesses).
struct VarBufs {
    __amdgpu_buffer_rsrc_t w, m, v;
    float* wptr;          // the variables once more as a plain pointer (atomic adds of the asynchronous workers)
};
// Several workers: the step is ADDED to the variable with a hardware float atomic, so concurrent updates of one
// element all land whatever their timing (a plain read-modify-write would keep one of them: workers that start
// together run in lockstep and would lose seven updates of eight).  The Adam slots stay plain stores: a lost slot
// update drops one sample from a running average.
template <int OPT>
__device__ __forceinline__ void opt_update_shared(const Opt& o, float* wptr, int byte_off, float& m, float& v, float g) {
    float step;
    if (OPT == 0) {
        m = o.b1 * m + (1.0f - o.b1) * g;
        v = o.b2 * v + (1.0f - o.b2) * g * g;
        step = o.lr_t * m * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + o.eps);
    } else {
        step = o.lr * g;
    }
    unsafeAtomicAdd(wptr + (byte_off >> 2), -step);
}
__device__ __forceinline__ float vb_load(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 16));   // sc1: L2-served
}
__device__ __forceinline__ void vb_store(__amdgpu_buffer_rsrc_t r, int byte_off, float x) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, x), r, byte_off, 0, 0);
}

// Old values of one 32x32 variable tile in the MFMA accumulator layout, requested ahead of use.
template <int OPT>
struct TileState {
    float w[16], m[16], v[16];

    __device__ __forceinline__ void issue(const VarBufs& vb, long long off, int N, int i0, int n0, int lane) {
        const int li = lane & 31, h = lane >> 5;
        const int base = ((int)off + (i0 + 4 * h) * N + n0 + li) * 4;
        if (OPT == 2) return;       // gradients only: no old values needed
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int bo = base + ((r & 3) + 8 * (r >> 2)) * N * 4;
            w[r] = vb_load(vb.w, bo);
            if (OPT == 0) {
                m[r] = vb_load(vb.m, bo);
                v[r] = vb_load(vb.v, bo);
            }
        }
    }
    // apply the gradient tile; `mirror` (optional) receives the new values as an LDS image [rows][ldm]
    template <bool SHARED = false>
    __device__ __forceinline__ void finish(const VarBufs& vb, const Opt& o, const f32x16& g, long long off, int N, int i0,
                                           int n0, int lane, float* mirror, int ldm, float* gdst = nullptr) {
        const int li = lane & 31, h = lane >> 5;
        const int base = ((int)off + (i0 + 4 * h) * N + n0 + li) * 4;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2);
            const int bo = base + rr * N * 4;
            if (OPT == 2) {             // gradients only: to the variable's gradient tensor, the variables stay as they are
                gdst[(i0 + 4 * h + rr) * N + n0 + li] = g[r];
                continue;
            }
            if (SHARED) {
                opt_update_shared<OPT>(o, vb.wptr, bo, m[r], v[r], g[r]);
                if (OPT == 0) {
                    vb_store(vb.m, bo, m[r]);
                    vb_store(vb.v, bo, v[r]);
                }
                continue;
            }
            const float nw = opt_update<OPT>(o, w[r], m[r], v[r], g[r]);
            vb_store(vb.w, bo, nw);
            if (OPT == 0) {
                vb_store(vb.m, bo, m[r]);
                vb_store(vb.v, bo, v[r]);
            }
            if (mirror) mirror[(i0 + 4 * h + rr) * ldm + n0 + li] = nw;
        }
    }
};

// One bias element per thread (tid < N): old values requested early, gradient = column sum of dz (rows >= B are 0).
template <int OPT>
struct BiasState {
    float w, m, v;
    __device__ __forceinline__ void issue(const VarBufs& vb, long long off, int N, int tid) {
        w = m = v = 0.f;
        if (tid < N && OPT != 2) {
            const int bo = ((int)off + tid) * 4;
            w = vb_load(vb.w, bo);
            if (OPT == 0) {
                m = vb_load(vb.m, bo);
                v = vb_load(vb.v, bo);
            }
        }
    }
    // `mirror`: LDS copy of the bias that the next step's forward epilogue reads
    template <bool SHARED = false>
    __device__ __forceinline__ void finish(const VarBufs& vb, const Opt& o, const float* dz, int ld, long long off, int N,
                                           int tid, float* mirror, float* gdst = nullptr) {
        if (tid < N) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 4
            for (int r = 0; r < BM; r += 4) {
                s0 += dz[r * ld + tid];
                s1 += dz[(r + 1) * ld + tid];
                s2 += dz[(r + 2) * ld + tid];
                s3 += dz[(r + 3) * ld + tid];
            }
            const int bo = ((int)off + tid) * 4;
            if (OPT == 2) {
                gdst[tid] = (s0 + s1) + (s2 + s3);
                return;
            }
            if (SHARED) {
                opt_update_shared<OPT>(o, vb.wptr, bo, m, v, (s0 + s1) + (s2 + s3));
                if (OPT == 0) {
                    vb_store(vb.m, bo, m);
                    vb_store(vb.v, bo, v);
                }
                return;
            }
            const float nw = opt_update<OPT>(o, w, m, v, (s0 + s1) + (s2 + s3));
            vb_store(vb.w, bo, nw);
            if (OPT == 0) {
                vb_store(vb.m, bo, m);
                vb_store(vb.v, bo, v);
            }
            mirror[tid] = nw;
        }
    }
};

template <int OPT, bool MULTI>
__global__ __launch_bounds__(NT, 1) void mlp3_fused_kernel(const FusedArgs p) {
    constexpr int WAUX = OPT == 2 ? 17 : 16;    // variable loads: a gradient-only launch may read them from a peer's memory
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sX = lds;
    float* sY = sX + BM * LX;
    float* sH1 = sY + BM * L3;
    float* sH2 = sH1 + BM * L1;
    float* sP = sH2 + BM * L2;
    float* sW = sP + BM * L3;
    float* sRed = sW + SW_FLOATS;   // 32 floats
    float* sB1 = sRed + 32;         // current biases (D1 + D2 + D3 floats)
    float* sB2 = sB1 + D1;
    float* sB3 = sB2 + D2;

    const int tid0 = threadIdx.x;
    int tid = tid0;
    const int B = p.B;
    const long long step0 = *p.step_ptr;
    int* sTicket = reinterpret_cast<int*>(sRed) + 24;      // [0] this step's ticket, [1] the next one
    bool worker = true;
    if (MULTI) {
        // elect an XCD, admit the first `workers` workgroups that run on it, hand the first ticket to each of them
        if (tid0 == 0) {
            const unsigned int xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xFu;   // HW_REG_XCC_ID[3:0]
            const unsigned int seen = atomicCAS(p.ctl + 0, 0xFFFFFFFFu, xcc);
            int ticket = -1;
            if (seen == 0xFFFFFFFFu || seen == xcc) {
                const unsigned int slot = atomicAdd(p.ctl + 1, 1u);
                if (slot < (unsigned)p.workers) {
                    // Workers that start together stay in lockstep (same code, same step time): all would compute
                    // their gradients from the same variables and apply them at once - one step of k times the
                    // batch and k times the rate.  Spread over the step period, every worker reads variables that
                    // already hold the others' latest updates: delayed-gradient SGD at the nominal rate.
                    const unsigned long long until = wall_clock64() + (unsigned long long)slot * (unsigned)p.stagger;
                    while (wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
                    ticket = (int)atomicAdd(p.ctl + 3, 1u);
                }
            }
            sTicket[0] = (ticket >= 0 && ticket < p.nsteps) ? ticket : -1;
        }
        __syncthreads();
        worker = sTicket[0] >= 0;
    }
    const int ticket = MULTI ? sTicket[0] : 0;
    int steps_done = 0;
    float loss_acc = 0.f, acc_acc = 0.f, cat_acc = 0.f;
    float b1_pow = 1.f, b2_pow = 1.f;
    const bool drop1 = p.training && p.keep1 < 1.0f, drop2 = p.training && p.keep2 < 1.0f;
    const float sc1 = drop1 ? 1.0f / p.keep1 : 1.0f, sc2 = drop2 ? 1.0f / p.keep2 : 1.0f;
    const unsigned W_BYTES = 0x7FFFFFF0u;
    VarBufs vb;
    vb.w = __builtin_amdgcn_make_buffer_rsrc(p.params, 0, (int)W_BYTES, 0x00020000);
    vb.m = __builtin_amdgcn_make_buffer_rsrc(OPT == 0 ? p.slot_m : p.params, 0, (int)W_BYTES, 0x00020000);
    vb.v = __builtin_amdgcn_make_buffer_rsrc(OPT == 0 ? p.slot_v : p.params, 0, (int)W_BYTES, 0x00020000);
    vb.wptr = p.params;
    // where each variable is read and (gradient-only launches) where its gradient goes
    const float* vp[6];
    float* gp[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        vp[k] = (OPT == 2 && p.var_ptr[k]) ? p.var_ptr[k] : p.params + p.off[k];
        gp[k] = OPT == 2 ? (p.grad_ptr[k] ? p.grad_ptr[k] : p.slot_m + p.off[k]) : nullptr;
    }
    unsigned long long pulled_version = 0;
    if (OPT == 2 && tid0 < p.ring_count)      // word 0 of owner tid0's block (this rank's replica): its version
        pulled_version = __hip_atomic_load(p.ring_rep[p.ring_self] + p.ring_block_word[tid0], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_SYSTEM);

    Stage<64, 128> pw;             // W1, a 64-row chunk of W2 (forward)
    Stage<128, 64> pt;             // a 64-column slice of W2 (dgrad)
    Stage<128, 32> pw3;            // W3
    Stage<BM, D0> px;              // next batch
    Stage<BM, D3> py;

    if (worker) {
    px.template issue<0>(p.X + (long long)ticket * B * D0, (unsigned)B * D0 * 4, D0, tid);
    py.template issue<0>(p.Y + (long long)ticket * B * D3, (unsigned)B * D3 * 4, D3, tid);
    pw.template issue<WAUX>(vp[0], W_BYTES, D1, tid);
    pw.commit(sW, tid);            // W1 image; later steps of a lone worker find it already there
    if (tid < D1) sB1[tid] = OPT == 2 ? ld1_sys(vp[1] + tid) : ld1_sc1(vp[1] + tid);
    if (tid < D2) sB2[tid] = OPT == 2 ? ld1_sys(vp[3] + tid) : ld1_sc1(vp[3] + tid);
    if (tid < D3) sB3[tid] = OPT == 2 ? ld1_sys(vp[5] + tid) : ld1_sc1(vp[5] + tid);

    for (int s = MULTI ? ticket : 0; s >= 0 && s < p.nsteps;) {
        // The thread index is made opaque once per step: every address and dropout-hash term below is step-invariant,
        // and hoisting them all out of this loop would cost far more registers than recomputing them.
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int li = lane & 31, h = lane >> 5;
        const int tm = wave >> 2, tn = wave & 3;       // this wave's tile of a [64 x 128] output
        const int col = tn * 32 + li;
        const long long step = step0 + s;
        DTF_PROF(0);
        Opt o;
        o.lr = p.lr; o.b1 = p.beta1; o.b2 = p.beta2; o.eps = p.eps;
        if (MULTI || (s & 63) == 0) {  // beta^t as running products, re-based every 64 steps: powf costs hundreds of
            b1_pow = powf(p.beta1, (float)step);    // instructions (a worker among several takes steps out of sequence)
            b2_pow = powf(p.beta2, (float)step);
        }
        b1_pow *= p.beta1;
        b2_pow *= p.beta2;
        o.lr_t = p.lr * sqrtf(1.0f - b2_pow) / (1.0f - b1_pow);
        const uint32_t key1 = dtf_dropout_key(p.seed1, &step), key2 = dtf_dropout_key(p.seed2, &step);

        // ---- batch -> LDS
        px.commit(sX, tid);
        py.commit(sY, tid);
        __syncthreads();
        DTF_PROF(1);
        const float bf1 = sB1[col], bf2 = sB2[col], bf3 = sB3[tid & 31];

        // ---- L1: h1 = dropout(relu(x W1 + b1))
        pw.template issue<WAUX>(vp[2], W_BYTES, D2, tid);                 // W2 rows 0..63
        {
            f32x16 acc; zero(acc);
            tile_mma<true, false, D0>(acc, sX, LX, tm * 32, sW, L1, tn * 32, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tm * 32 + tile_row(r, h);
                float v = fmaxf(acc[r] + bf1, 0.f);
                if (drop1) v = (dtf_uniform(key1, (unsigned long long)row * (unsigned)D1 + col) < p.keep1) ? v * sc1 : 0.f;
                sH1[row * L1 + col] = v;
            }
        }
        __syncthreads();
        DTF_PROF(2);
        // ---- L2: h2 = dropout(relu(h1 W2 + b2)); two k-chunks of 64
        pw.commit(sW, tid);
        pw.template issue<WAUX>(vp[2] + 64 * D2, W_BYTES, D2, tid);       // W2 rows 64..127
        __syncthreads();
        f32x16 acc2; zero(acc2);
        tile_mma<true, false, 64>(acc2, sH1, L1, tm * 32, sW, L2, tn * 32, lane);
        __syncthreads();
        DTF_PROF(3);
        pw.commit(sW, tid);
        pw3.template issue<WAUX>(vp[4], W_BYTES, D3, tid);                // W3
        __syncthreads();
        tile_mma<true, false, 64>(acc2, sH1 + 64, L1, tm * 32, sW, L2, tn * 32, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tm * 32 + tile_row(r, h);
            float v = fmaxf(acc2[r] + bf2, 0.f);
            if (drop2) v = (dtf_uniform(key2, (unsigned long long)row * (unsigned)D2 + col) < p.keep2) ? v * sc2 : 0.f;
            sH2[row * L2 + col] = v;
        }
        __syncthreads();
        DTF_PROF(4);
        // ---- L3: p = sigmoid(h2 W3 + b3).  The W3 image [128][36] serves forward (B[k][n]) and dgrad (B[n'][k'],
        //      n' = input unit, k' = output unit) alike.  Two row tiles x two k-halves on waves 0..3; the halves meet
        //      in the elementwise pass below (fixed order).
        pw3.commit(sW, tid);
        __syncthreads();
        if (wave < 4) {
            const int rt = wave & 1, kh = wave >> 1;
            f32x16 acc; zero(acc);
            tile_mma<true, false, 64>(acc, sH2 + kh * 64, L2, rt * 32, sW + kh * 64 * L3, L3, 0, lane);
            float* dst = kh ? sW + W3_SCRATCH : sP;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(rt * 32 + tile_row(r, h)) * L3 + li] = acc[r];
        }
        __syncthreads();
        DTF_PROF(5);
        // ---- sigmoid, loss, metrics, dz3 (in place of p); rows >= B become zero.  Element e = tid + NT*i:
        //      row = e / 32, column = tid & 31.
        {
            float pv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (tid >> 5) + 16 * i, c = tid & 31;
                pv[i] = dtf_sigmoid(sP[row * L3 + c] + sW[W3_SCRATCH + row * L3 + c] + bf3);
            }
            if (p.want_cat) {
#pragma unroll
                for (int i = 0; i < 4; ++i) sP[((tid >> 5) + 16 * i) * L3 + (tid & 31)] = pv[i];
                __syncthreads();
                if (tid < B) {
                    int bp = 0, by = 0;
                    float vp = sP[tid * L3], vy = sY[tid * L3];
                    for (int c = 1; c < D3; ++c) {
                        const float a = sP[tid * L3 + c], b = sY[tid * L3 + c];
                        if (a > vp) { vp = a; bp = c; }
                        if (b > vy) { vy = b; by = c; }
                    }
                    cat_acc += (bp == by) ? 1.f : 0.f;
                }
                __syncthreads();
            }
            const float gs = 2.0f / ((float)B * (float)D3);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (tid >> 5) + 16 * i, c = tid & 31;
                float dz = 0.f;
                if (row < B) {
                    const float yv = sY[row * L3 + c], d = pv[i] - yv;
                    loss_acc += d * d;
                    acc_acc += (rintf(pv[i]) == rintf(yv)) ? 1.f : 0.f;
                    dz = gs * d * pv[i] * (1.0f - pv[i]);
                }
                sP[row * L3 + c] = dz;
            }
        }
        if (MULTI && tid == 0) sTicket[1] = (int)atomicAdd(p.ctl + 3, 1u);     // the step this worker takes next
        __syncthreads();
        DTF_PROF(6);

        // ---- dW3 = h2^T dz3 (waves 0..3: input-unit block = wave) and the dz2 products, both reading h2
        TileState<OPT> ts;
        BiasState<OPT> bs3, bs2, bs1;
        if (wave < 4) ts.issue(vb, p.off[4], D3, wave * 32, 0, lane);
        bs3.issue(vb, p.off[5], D3, tid);
        bs2.issue(vb, p.off[3], D2, tid);
        bs1.issue(vb, p.off[1], D1, tid);
        pt.template issue<WAUX>(vp[2], W_BYTES, D2, tid);                 // W2 columns 0..63
        f32x16 g3; zero(g3);
        if (wave < 4) tile_mma<false, false, BM>(g3, sH2, L2, wave * 32, sP, L3, 0, lane);
        f32x16 az; zero(az);
        tile_mma<true, true, D3>(az, sP, L3, tm * 32, sW, L3, tn * 32, lane);
        __syncthreads();                                                                // every reader of h2 is done
        DTF_PROF(7);
        // ---- dz2 = (dz3 W3^T) * (h2 > 0) / keep2, in place of h2; apply W3 and b3
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tm * 32 + tile_row(r, h);
            const float y = sH2[row * L2 + col];
            sH2[row * L2 + col] = (y > 0.f) ? az[r] * sc2 : 0.f;
        }
        __syncthreads();                                                                // dz2 complete, W3 image free
        DTF_PROF(8);

        // ---- dW2 = h1^T dz2 held in registers: tiles t = wave, wave + 8 (input block t >> 2, output block t & 3).
        //      Waves 0..3 first apply W3 / b3 (VALU + memory) while their SIMD partners 4..7 already run MFMAs.
        pt.commit(sW, tid);
        pt.template issue<WAUX>(vp[2] + 64, W_BYTES, D2, tid);            // W2 columns 64..127
        if (wave < 4) ts.template finish<MULTI>(vb, o, g3, p.off[4], D3, wave * 32, 0, lane, nullptr, 0, gp[4]);
        bs3.template finish<MULTI>(vb, o, sP, L3, p.off[5], D3, tid, sB3, gp[5]);
        f32x16 g2[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            zero(g2[q]);
            const int t = wave + 8 * q;
            tile_mma<false, false, BM>(g2[q], sH1, L1, (t >> 2) * 32, sH2, L2, (t & 3) * 32, lane);
        }
        bs2.template finish<MULTI>(vb, o, sH2, L2, p.off[3], D2, tid, sB2, gp[3]);
        __syncthreads();                                                                // h1 readers done; slice 0 visible
        DTF_PROF(9);
        // ---- dz1 = (dz2 W2^T) * (h1 > 0) / keep1 in place of h1, summed over two 64-wide slices of W2's columns
        {
            f32x16 a1; zero(a1);
            tile_mma<true, true, 64>(a1, sH2, L2, tm * 32, sW, LT, tn * 32, lane);
            __syncthreads();
            pt.commit(sW, tid);
            __syncthreads();
            tile_mma<true, true, 64>(a1, sH2 + 64, L2, tm * 32, sW, LT, tn * 32, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tm * 32 + tile_row(r, h);
                const float y = sH1[row * L1 + col];
                sH1[row * L1 + col] = (y > 0.f) ? a1[r] * sc1 : 0.f;
            }
        }
        __syncthreads();        // dz1 complete; every wave's W2 loads have landed (each committed its own); sW is free
        DTF_PROF(10);
        // ---- apply W2 from the held accumulators, and dW1 = x^T dz1 applied directly (input block tm, output block
        //      tn; new W1 mirrored into its LDS image).  The two waves of a SIMD take the two jobs in opposite order,
        //      so one's update arithmetic and memory traffic overlap the other's MFMAs.
        const int next = MULTI ? sTicket[1] : s + 1;       // (written several barriers ago)
        if (next < p.nsteps) {
            px.template issue<0>(p.X + (long long)next * B * D0, (unsigned)B * D0 * 4, D0, tid);
            py.template issue<0>(p.Y + (long long)next * B * D3, (unsigned)B * D3 * 4, D3, tid);
        }
        {
            const int i2a = (wave >> 2) * 32, i2b = ((wave + 8) >> 2) * 32, n2 = (wave & 3) * 32;
            TileState<OPT> t1, t2;
            if (wave < 4) {
                t2.issue(vb, p.off[2], D2, i2a, n2, lane);
                t1.issue(vb, p.off[0], D1, tm * 32, tn * 32, lane);
                t2.template finish<MULTI>(vb, o, g2[0], p.off[2], D2, i2a, n2, lane, nullptr, 0, gp[2]);
                t2.issue(vb, p.off[2], D2, i2b, n2, lane);
                f32x16 acc; zero(acc);
                tile_mma<false, false, BM>(acc, sX, LX, tm * 32, sH1, L1, tn * 32, lane);
                t2.template finish<MULTI>(vb, o, g2[1], p.off[2], D2, i2b, n2, lane, nullptr, 0, gp[2]);
                t1.template finish<MULTI>(vb, o, acc, p.off[0], D1, tm * 32, tn * 32, lane, sW, L1, gp[0]);
            } else {
                t1.issue(vb, p.off[0], D1, tm * 32, tn * 32, lane);
                t2.issue(vb, p.off[2], D2, i2a, n2, lane);
                f32x16 acc; zero(acc);
                tile_mma<false, false, BM>(acc, sX, LX, tm * 32, sH1, L1, tn * 32, lane);
                t2.template finish<MULTI>(vb, o, g2[0], p.off[2], D2, i2a, n2, lane, nullptr, 0, gp[2]);
                t2.issue(vb, p.off[2], D2, i2b, n2, lane);
                t1.template finish<MULTI>(vb, o, acc, p.off[0], D1, tm * 32, tn * 32, lane, sW, L1, gp[0]);
                t2.template finish<MULTI>(vb, o, g2[1], p.off[2], D2, i2b, n2, lane, nullptr, 0, gp[2]);
            }
        }
        DTF_PROF(11);
        bs1.template finish<MULTI>(vb, o, sH1, L1, p.off[1], D1, tid, sB1, gp[1]);
        // every wave's variable stores must have reached L2 before the next step requests them again
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        DTF_PROF(12);
        ++steps_done;
        if (MULTI) {
            // other workers have updated the variables meanwhile: W1 and the biases come from L2 again (the other
            // weights are streamed every step anyway); the barrier at the top of the next step publishes the images
            if (next < p.nsteps) {
                pw.template issue<WAUX>(vp[0], W_BYTES, D1, tid);
                pw.commit(sW, tid);
                if (tid < D1) sB1[tid] = ld1_sc1(vp[1] + tid);
                if (tid < D2) sB2[tid] = ld1_sc1(vp[3] + tid);
                if (tid < D3) sB3[tid] = ld1_sc1(vp[5] + tid);
            }
            s = next;
        } else {
            ++s;
        }
    }
    }   // worker

    // ---- gradient-only launch of a parameter-server worker: every gradient store has left this CU; tell the owners
    if (OPT == 2 && p.ring_count > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __threadfence_system();
        __syncthreads();
        if (tid0 < p.ring_count) {      // thread i rings owner i, in every replica it was given: pull_version[slot], then the door
            const long long lw = p.ring_lane_word[tid0];
            for (int r = 0; r < p.ring_world; ++r)      // (the clock is published by owner 0 with its acknowledgement)
                __hip_atomic_store(p.ring_rep[r] + lw + 2 + p.ring_slot, pulled_version, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
            for (int r = 0; r < p.ring_world; ++r)
                __hip_atomic_store(p.ring_rep[r] + lw, p.ring_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }

    // ---- once per launch: statistics, global_step, beta powers
    const int lane = tid0 & 63, wave = tid0 >> 6;
    if (p.stats && worker) {
        const float l = dtf_wave_sum(loss_acc), a = dtf_wave_sum(acc_acc), c = dtf_wave_sum(cat_acc);
        if (lane == 0) { sRed[wave] = l; sRed[8 + wave] = a; sRed[16 + wave] = c; }
        __syncthreads();
        if (tid0 == 0) {
            float sl = 0.f, sa = 0.f, sc = 0.f;
            for (int w = 0; w < NT / 64; ++w) { sl += sRed[w]; sa += sRed[8 + w]; sc += sRed[16 + w]; }
            const float inv = 1.0f / (float)D3;
            atomicAdd(p.stats + 0, sl * inv);
            atomicAdd(p.stats + 1, sa * inv);
            atomicAdd(p.stats + 2, (float)B * (float)steps_done);
            if (p.want_cat) atomicAdd(p.stats + 3, sc);
        }
    }
    bool last = true;
    if (MULTI) {
        // the workgroup that leaves last (every worker has finished its steps by then) closes the launch and re-arms
        // the control words for the next one
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid0 == 0) {
            last = atomicAdd(p.ctl + 2, 1u) == gridDim.x - 1;
            if (last) {
                __hip_atomic_store(p.ctl + 0, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.ctl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.ctl + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.ctl + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (tid0 == 0 && last && OPT != 2) {
        *p.step_ptr = step0 + p.nsteps;
        if (p.beta_powers && OPT == 0) {
            p.beta_powers[0] *= powf(p.beta1, (float)p.nsteps);
            p.beta_powers[1] *= powf(p.beta2, (float)p.nsteps);
        }
    }
}

long long* g_prof_buffer = nullptr;

}  // namespace

// Profiling aid: device buffer of 16 int64 that receives shader-clock timestamps of the phases of each launch's
// last step (null switches it off).
DTF_API int dtf_mlp3_fused_set_profile_buffer(long long* device_buffer) {
    g_prof_buffer = device_buffer;
    return 0;
}

// Supported shape of this build: (64, 128, 128, 32) - the reference network - at batch <= 64.
DTF_API int dtf_mlp3_fused_supported(int d0, int d1, int d2, int d3, int B) {
    return (d0 == D0 && d1 == D1 && d2 == D2 && d3 == D3 && B >= 1 && B <= BM) ? 1 : 0;
}

// nsteps full training steps (relu, relu, sigmoid; MSE; Adam when opt == 0, SGD when opt == 1) on consecutive
// mini-batches X[k*B:(k+1)*B], Y[k*B:(k+1)*B].
namespace {
int launch_fused(FusedArgs a, int opt, hipStream_t stream) {
    constexpr size_t lds = sizeof(float) * LDS_FLOATS;
    static bool once = false;
    if (!once) {
        const void* fns[5] = {reinterpret_cast<const void*>(mlp3_fused_kernel<0, false>),
                              reinterpret_cast<const void*>(mlp3_fused_kernel<1, false>),
                              reinterpret_cast<const void*>(mlp3_fused_kernel<0, true>),
                              reinterpret_cast<const void*>(mlp3_fused_kernel<1, true>),
                              reinterpret_cast<const void*>(mlp3_fused_kernel<2, false>)};
        for (const void* f : fns) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        once = true;
    }
    if (a.workers > 1) {
        // eight workgroups per worker: dealt round-robin over the eight XCDs, about `workers` of them land on the XCD
        // that gets elected (the others leave at once)
        const dim3 grid((unsigned)(8 * a.workers));
        if (opt == 0) hipLaunchKernelGGL((mlp3_fused_kernel<0, true>), grid, dim3(NT), lds, stream, a);
        else hipLaunchKernelGGL((mlp3_fused_kernel<1, true>), grid, dim3(NT), lds, stream, a);
    } else {
        if (opt == 0) hipLaunchKernelGGL((mlp3_fused_kernel<0, false>), dim3(1), dim3(NT), lds, stream, a);
        else if (opt == 2) hipLaunchKernelGGL((mlp3_fused_kernel<2, false>), dim3(1), dim3(NT), lds, stream, a);
        else hipLaunchKernelGGL((mlp3_fused_kernel<1, false>), dim3(1), dim3(NT), lds, stream, a);
    }
    DTF_CHECK_LAUNCH();
    return 0;
}
}  // namespace

// The same with up to `workers` asynchronous workers (workgroups) sharing the variables, lock-free; `ctl` is a device
// array of four 32-bit words initialised to {0xFFFFFFFF, 0, 0, 0} that every launch leaves in that state.
DTF_API int dtf_mlp3_fused_train_async(const float* X, const float* Y, float* params, float* slot_m, float* slot_v,
                                       const long long* offsets6, int d0, int d1, int d2, int d3, int B, int nsteps,
                                       int training, int opt, int want_cat, float keep1, float keep2, uint32_t seed1,
                                       uint32_t seed2, float lr, float beta1, float beta2, float eps, long long* step_ptr,
                                       float* beta_powers, float* stats, unsigned int* ctl, int workers,
                                       hipStream_t stream) {
    if (!dtf_mlp3_fused_supported(d0, d1, d2, d3, B)) return -4;
    if (opt == 0 && (!slot_m || !slot_v)) return -5;
    if (opt != 0 && opt != 1) return -6;
    if (workers < 1 || workers > 32 || (workers > 1 && !ctl)) return -7;
    if (nsteps <= 0) return 0;
    FusedArgs a = {};
    a.X = X; a.Y = Y; a.params = params; a.slot_m = slot_m; a.slot_v = slot_v;
    for (int i = 0; i < 6; ++i) a.off[i] = offsets6[i];
    a.B = B; a.nsteps = nsteps; a.training = training; a.want_cat = want_cat;
    a.keep1 = keep1; a.keep2 = keep2; a.seed1 = seed1; a.seed2 = seed2;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.step_ptr = step_ptr; a.beta_powers = beta_powers; a.stats = stats; a.prof = workers > 1 ? nullptr : g_prof_buffer;
    a.ctl = ctl; a.workers = workers;
    a.stagger = 3700 / workers;      // a step takes ~37 us = 3700 ticks: the workers' phases tile one step period
    return launch_fused(a, opt, stream);
}

DTF_API int dtf_mlp3_fused_train(const float* X, const float* Y, float* params, float* slot_m, float* slot_v,
                                 const long long* offsets6, int d0, int d1, int d2, int d3, int B, int nsteps,
                                 int training, int opt, int want_cat, float keep1, float keep2, uint32_t seed1,
                                 uint32_t seed2, float lr, float beta1, float beta2, float eps, long long* step_ptr,
                                 float* beta_powers, float* stats, hipStream_t stream) {
    if (!dtf_mlp3_fused_supported(d0, d1, d2, d3, B)) return -4;
    if (opt == 0 && (!slot_m || !slot_v)) return -5;
    if (opt != 0 && opt != 1) return -6;
    if (nsteps <= 0) return 0;
    FusedArgs a = {};
    a.X = X; a.Y = Y; a.params = params; a.slot_m = slot_m; a.slot_v = slot_v;
    for (int i = 0; i < 6; ++i) a.off[i] = offsets6[i];
    a.B = B; a.nsteps = nsteps; a.training = training; a.want_cat = want_cat;
    a.keep1 = keep1; a.keep2 = keep2; a.seed1 = seed1; a.seed2 = seed2;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.step_ptr = step_ptr; a.beta_powers = beta_powers; a.stats = stats; a.prof = g_prof_buffer;
    a.ctl = nullptr; a.workers = 1;
    return launch_fused(a, opt, stream);
}

// Forward, loss and backward of ONE mini-batch in one launch: the gradients of W1, b1, W2, b2, W3, b3 land in `grads`
// at the variables' offsets, nothing else changes (no update, no step increment; the dropout masks are those of the
// step `*step_ptr`).  What a replica of a cluster runs before it hands its gradients to the strategy - an asynchronous
// worker's push to the parameter servers (/root/reference/example.py:168-170), a synchronous replica's all-reduce.
DTF_API int dtf_mlp3_fused_grad(const float* X, const float* Y, const float* params, float* grads, const long long* offsets6,
                                int d0, int d1, int d2, int d3, int B, int training, int want_cat, float keep1, float keep2,
                                uint32_t seed1, uint32_t seed2, const long long* step_ptr, float* stats, hipStream_t stream) {
    if (!dtf_mlp3_fused_supported(d0, d1, d2, d3, B)) return -4;
    if (!grads || !params || !step_ptr) return -5;
    FusedArgs a = {};
    a.X = X; a.Y = Y; a.params = const_cast<float*>(params); a.slot_m = grads; a.slot_v = nullptr;
    for (int i = 0; i < 6; ++i) a.off[i] = offsets6[i];
    a.B = B; a.nsteps = 1; a.training = training; a.want_cat = want_cat;
    a.keep1 = keep1; a.keep2 = keep2; a.seed1 = seed1; a.seed2 = seed2;
    a.lr = 0.f; a.beta1 = 0.9f; a.beta2 = 0.999f; a.eps = 1e-8f;
    a.step_ptr = const_cast<long long*>(step_ptr); a.beta_powers = nullptr; a.stats = stats; a.prof = nullptr;
    a.ctl = nullptr; a.workers = 1;
    return launch_fused(a, 2, stream);
}


// The same for a worker whose variables live with parameter-server tasks: variable i is read from var_ptr6[i], its
// gradient goes to grad_ptr6[i] (both peer-mapped memory of the owning task), `step_ptr` is the shared global step as
// the GPU addresses it, and the launch ends by publishing gradient number `seq` (inbox slot `slot`) to `count` <= 8
// owners (count == 0: the caller rings the doorbells behind this launch): `replicas` are the control areas of all
// `world` ranks as this GPU addresses them, `self_rank` the one in this rank's own memory, `block_words[i]` /
// `lane_words[i]` the positions of owner i's block and of this worker's lane in it inside an area.  The owners'
// version words are read when the launch starts (what the gradient counts as computed from); `clock` goes to owner
// 0's block.
DTF_API int dtf_mlp3_fused_grad_direct(const float* X, const float* Y, const void* const* var_ptr6, void* const* grad_ptr6,
                                       int d0, int d1, int d2, int d3, int B, int training, int want_cat, float keep1,
                                       float keep2, uint32_t seed1, uint32_t seed2, const long long* step_ptr, float* stats,
                                       void* const* replicas, int world, int self_rank, const long long* block_words,
                                       const long long* lane_words, int count, int slot, unsigned long long seq,
                                       unsigned long long clock, hipStream_t stream) {
    if (!dtf_mlp3_fused_supported(d0, d1, d2, d3, B)) return -4;
    if (!var_ptr6 || !grad_ptr6 || !step_ptr || count < 0 || count > 8 || slot < 0 || slot > 3) return -5;
    if (count > 0 && (!replicas || world < 1 || world > 64 || self_rank < 0 || self_rank >= world || !block_words || !lane_words))
        return -5;
    FusedArgs a = {};
    a.X = X; a.Y = Y;
    for (int i = 0; i < 6; ++i) {
        if (!var_ptr6[i] || !grad_ptr6[i]) return -5;
        a.var_ptr[i] = static_cast<const float*>(var_ptr6[i]);
        a.grad_ptr[i] = static_cast<float*>(grad_ptr6[i]);
    }
    if (count > 0) {
        for (int r = 0; r < world; ++r) {
            if (!replicas[r]) return -5;
            a.ring_rep[r] = static_cast<unsigned long long*>(replicas[r]);
        }
        a.ring_world = world; a.ring_self = self_rank;
        for (int i = 0; i < count; ++i) {
            a.ring_block_word[i] = block_words[i];
            a.ring_lane_word[i] = lane_words[i];
        }
    }
    a.ring_count = count; a.ring_slot = slot; a.ring_seq = seq; a.ring_clock = clock;
    a.B = B; a.nsteps = 1; a.training = training; a.want_cat = want_cat;
    a.keep1 = keep1; a.keep2 = keep2; a.seed1 = seed1; a.seed2 = seed2;
    a.lr = 0.f; a.beta1 = 0.9f; a.beta2 = 0.999f; a.eps = 1e-8f;
    a.step_ptr = const_cast<long long*>(step_ptr); a.stats = stats;
    a.workers = 1;
    return launch_fused(a, 2, stream);
}
