// Loss / metric kernels with the gradient fused into the same pass.
//
//  * sigmoid-output mean-squared-error + rounded-bit accuracy: the loss and metric of
//    /root/reference/example.py:157-164 (Keras mean_squared_error = mean over the last
//    axis, then reduce_mean over the batch; accuracy = mean(round(p) == round(y)) with
//    round-half-to-even).
//  * binary cross-entropy on the same sigmoid outputs (Keras binary_crossentropy: probabilities clipped to
//    [1e-7, 1 - 1e-7], mean over the last axis), the loss such a head is usually trained with.
//  * softmax cross-entropy + argmax accuracy for MNIST-shaped classifiers.
//
// Every kernel ADDS into a 4-float running-statistics record
//   stats[0] += sum over examples of the per-example loss
//   stats[1] += sum over examples of the per-example accuracy (0..1)
//   stats[2] += number of examples
// so the host can read means every N steps instead of synchronising every step.
#include "common.h"

namespace {

__device__ __forceinline__ void block_add3(float a, float b, float c, float* stats) {
    __shared__ float red[3][4];
    a = dtf_wave_sum(a);
    b = dtf_wave_sum(b);
    c = dtf_wave_sum(c);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wave] = a;
        red[1][wave] = b;
        red[2][wave] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float sa = 0.f, sb = 0.f, sc = 0.f;
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 0; w < nw; ++w) {
            sa += red[0][w];
            sb += red[1][w];
            sc += red[2][w];
        }
        atomicAdd(stats + 0, sa);
        atomicAdd(stats + 1, sb);
        atomicAdd(stats + 2, sc);
    }
}

// Elementwise over the flat [M*N] range, 16 bytes per lane.  P are probabilities (the dense
// layer already applied the sigmoid).  G = dLoss/dP * (P (1-P) if through_sigmoid) for
// Loss = mean_{M,N} (P - Y)^2.  Each element carries weight 1/N into the per-example sums.
//
// BCE: Loss = mean_{M,N} -(Y log Pc + (1-Y) log(1-Pc)) with Pc = clip(P, 1e-7, 1 - 1e-7); G = (Pc - Y) / (Pc (1 - Pc))
// (just Pc - Y through the sigmoid), and zero where the clip is active - the gradient Keras' clip passes on.
constexpr float BCE_EPS = 1e-7f;

template <bool BCE>
__device__ __forceinline__ void loss_element(float p, float y, float grad_scale, int through_sigmoid, float& loss,
                                             float& acc, float& g) {
    acc += (rintf(p) == rintf(y)) ? 1.f : 0.f;
    if (BCE) {
        const float pc = fminf(fmaxf(p, BCE_EPS), 1.0f - BCE_EPS);
        loss -= y * logf(pc) + (1.0f - y) * logf(1.0f - pc);
        g = (pc == p) ? (through_sigmoid ? (pc - y) : (pc - y) / (pc * (1.0f - pc))) * grad_scale : 0.0f;
    } else {
        const float d = p - y;
        loss += d * d;
        g = 2.0f * d * grad_scale;
        if (through_sigmoid) g *= p * (1.0f - p);
    }
}

template <bool BCE>
__global__ __launch_bounds__(256) void mse_flat_kernel(const float* __restrict__ P, const float* __restrict__ Y,
                                                       float* __restrict__ G, long long total, float inv_n,
                                                       float grad_scale, int through_sigmoid, float* stats) {
    float loss = 0.f, acc = 0.f, cnt = 0.f;
    const long long n4 = total >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const f32x4* P4 = reinterpret_cast<const f32x4*>(P);
    const f32x4* Y4 = reinterpret_cast<const f32x4*>(Y);
    f32x4* G4 = reinterpret_cast<f32x4*>(G);
    for (long long i = tid; i < n4; i += stride) {
        const f32x4 p = P4[i], y = Y4[i];
        f32x4 g;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gj;
            loss_element<BCE>(p[j], y[j], grad_scale, through_sigmoid, loss, acc, gj);
            g[j] = gj;
        }
        cnt += 4.f;
        if (G) G4[i] = g;
    }
    for (long long i = (n4 << 2) + tid; i < total; i += stride) {
        float gj;
        loss_element<BCE>(P[i], Y[i], grad_scale, through_sigmoid, loss, acc, gj);
        cnt += 1.f;
        if (G) G[i] = gj;
    }
    if (stats) block_add3(loss * inv_n, acc * inv_n, cnt * inv_n, stats);
}

// Softmax cross-entropy, 256 rows per block.  The [256, C] logit tile is staged through LDS
// with coalesced loads (row stride C+1 floats, so the per-row walk is bank-conflict free), each
// thread then owns one row, and the gradient tile leaves through LDS with coalesced stores.
// labels: int64 class ids (sparse) or one-hot / probability rows (dense).
__global__ __launch_bounds__(256) void softmax_xent_kernel(const float* __restrict__ Z,
                                                           const long long* __restrict__ sparse_labels,
                                                           const float* __restrict__ dense_labels,
                                                           float* __restrict__ G, int M, int C, float grad_scale,
                                                           float* stats) {
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [256][C+1] logits (+ [256][C+1] dense labels)
    const int ld = C + 1;
    float* zt = tile;
    float* yt = tile + 256 * ld;
    float loss = 0.f, acc = 0.f, cnt = 0.f;
    const int tid = threadIdx.x;
    for (long long row0 = (long long)blockIdx.x * 256; row0 < M; row0 += (long long)gridDim.x * 256) {
        const int rows = (int)min((long long)256, M - row0);
        const int elems = rows * C;
        const float* zsrc = Z + row0 * C;
        for (int e = tid; e < elems; e += 256) {
            const int r = e / C, c = e - r * C;
            zt[r * ld + c] = zsrc[e];
            if (dense_labels) yt[r * ld + c] = dense_labels[row0 * C + e];
        }
        __syncthreads();
        if (tid < rows) {
            float* z = zt + tid * ld;
            float mx = z[0];
            int arg = 0;
            for (int j = 1; j < C; ++j) {
                const float v = z[j];
                if (v > mx) {
                    mx = v;
                    arg = j;
                }
            }
            float se = 0.f;
            for (int j = 0; j < C; ++j) se += __expf(z[j] - mx);
            const float lse = mx + __logf(se);
            const float inv = 1.0f / se;
            float l = 0.f;
            int label_arg = 0;
            if (sparse_labels) {
                label_arg = (int)sparse_labels[row0 + tid];
                l = lse - z[label_arg];
                for (int j = 0; j < C; ++j)
                    z[j] = (__expf(z[j] - mx) * inv - (j == label_arg ? 1.f : 0.f)) * grad_scale;
            } else {
                const float* y = yt + tid * ld;
                float ymax = y[0], ysum = 0.f;
                for (int j = 0; j < C; ++j) {
                    const float yj = y[j];
                    if (yj > ymax) {
                        ymax = yj;
                        label_arg = j;
                    }
                    ysum += yj;
                    l += yj * (lse - z[j]);
                }
                for (int j = 0; j < C; ++j) z[j] = (__expf(z[j] - mx) * inv * ysum - y[j]) * grad_scale;
            }
            loss += l;
            acc += (arg == label_arg) ? 1.f : 0.f;
            cnt += 1.f;
        }
        __syncthreads();
        if (G) {
            float* gdst = G + row0 * C;
            for (int e = tid; e < elems; e += 256) {
                const int r = e / C, c = e - r * C;
                gdst[e] = zt[r * ld + c];
            }
        }
        __syncthreads();
    }
    if (stats) block_add3(loss, acc, cnt, stats);
}

// argmax(P) == argmax(Y) per row: Keras "categorical_accuracy", which is what
// metrics=['accuracy'] resolves to at /root/reference/example2.py:165.
__global__ __launch_bounds__(256) void categorical_acc_kernel(const float* __restrict__ P,
                                                              const float* __restrict__ Y, int M, int N,
                                                              float* acc_sum) {
    float acc = 0.f;
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < M; row += gridDim.x * blockDim.x) {
        const float* p = P + (long long)row * N;
        const float* y = Y + (long long)row * N;
        int ap = 0, ay = 0;
        for (int j = 1; j < N; ++j) {
            if (p[j] > p[ap]) ap = j;
            if (y[j] > y[ay]) ay = j;
        }
        acc += (ap == ay) ? 1.f : 0.f;
    }
    acc = dtf_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(acc_sum, acc);
}

// Row softmax (one thread per row): class-probability output of a softmax head.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ Z, float* __restrict__ P,
                                                           int M, int C) {
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < M; row += gridDim.x * blockDim.x) {
        const float* z = Z + (long long)row * C;
        float mx = z[0];
        for (int j = 1; j < C; ++j) mx = fmaxf(mx, z[j]);
        float se = 0.f;
        for (int j = 0; j < C; ++j) se += __expf(z[j] - mx);
        const float inv = 1.0f / se;
        for (int j = 0; j < C; ++j) P[(long long)row * C + j] = __expf(z[j] - mx) * inv;
    }
}

inline int row_grid(int M) {
    int g = (M + 255) / 256;
    if (g > 512) g = 512;        // the kernels are grid-stride; every workgroup ends in same-address atomics (see dtf_mse_loss)
    return g < 1 ? 1 : g;
}

}  // namespace

static int launch_flat_loss(bool bce, const float* P, const float* Y, float* G, int M, int N, float grad_scale,
                            int through_sigmoid, float* stats, hipStream_t stream) {
    if (M <= 0 || N <= 0) return 0;
    const long long total = (long long)M * N;
    // one workgroup per CU at most: every workgroup ends with three atomic adds on the same three statistics words, and
    // those serialise in L2 at ~14 ns each (2048 workgroups: 43 us of atomics behind 5 us of streaming)
    long long g = (total / 4 + 255) / 256;
    if (g > 256) g = 256;
    if (g < 1) g = 1;
    if (bce)
        hipLaunchKernelGGL(mse_flat_kernel<true>, dim3((int)g), dim3(256), 0, stream, P, Y, G, total, 1.0f / (float)N,
                           grad_scale, through_sigmoid, stats);
    else
        hipLaunchKernelGGL(mse_flat_kernel<false>, dim3((int)g), dim3(256), 0, stream, P, Y, G, total, 1.0f / (float)N,
                           grad_scale, through_sigmoid, stats);
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_mse_loss(const float* P, const float* Y, float* G, int M, int N, float grad_scale,
                         int through_sigmoid, float* stats, hipStream_t stream) {
    return launch_flat_loss(false, P, Y, G, M, N, grad_scale, through_sigmoid, stats, stream);
}

DTF_API int dtf_bce_loss(const float* P, const float* Y, float* G, int M, int N, float grad_scale,
                         int through_sigmoid, float* stats, hipStream_t stream) {
    return launch_flat_loss(true, P, Y, G, M, N, grad_scale, through_sigmoid, stats, stream);
}

DTF_API int dtf_softmax_xent(const float* Z, const long long* sparse_labels, const float* dense_labels, float* G,
                             int M, int C, float grad_scale, float* stats, hipStream_t stream) {
    if (M <= 0 || C <= 0) return 0;
    const size_t lds = (size_t)256 * (C + 1) * sizeof(float) * (dense_labels ? 2 : 1);
    if (lds > 150 * 1024) return -3;  // class count beyond what one block stages (about 70 dense / 145 sparse)
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(softmax_xent_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(softmax_xent_kernel, dim3(row_grid(M)), dim3(256), lds, stream, Z, sparse_labels,
                       dense_labels, G, M, C, grad_scale, stats);
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_categorical_accuracy(const float* P, const float* Y, int M, int N, float* acc_sum,
                                     hipStream_t stream) {
    if (M <= 0 || N <= 0) return 0;
    hipLaunchKernelGGL(categorical_acc_kernel, dim3(row_grid(M)), dim3(256), 0, stream, P, Y, M, N, acc_sum);
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_softmax_rows(const float* Z, float* P, int M, int C, hipStream_t stream) {
    if (M <= 0 || C <= 0) return 0;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(row_grid(M)), dim3(256), 0, stream, Z, P, M, C);
    DTF_CHECK_LAUNCH();
    return 0;
}
