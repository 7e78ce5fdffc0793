// Loss / metric kernels with the gradient fused into the same pass.
//
//  * sigmoid-output mean-squared-error + rounded-bit accuracy: the loss and metric of
//    /root/reference/example.py:157-164 (Keras mean_squared_error = mean over the last
//    axis, then reduce_mean over the batch; accuracy = mean(round(p) == round(y)) with
//    round-half-to-even).
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

// One thread per example row (N is a classifier/bit-vector width: tens of columns).
// P are probabilities (the dense layer already applied the sigmoid).  Writes
// G = dLoss/dP * (P (1-P) if through_sigmoid) for Loss = mean_{M,N} (P - Y)^2.
__global__ __launch_bounds__(256) void mse_rows_kernel(const float* __restrict__ P, const float* __restrict__ Y,
                                                       float* __restrict__ G, int M, int N, float grad_scale,
                                                       int through_sigmoid, float* stats) {
    float loss = 0.f, acc = 0.f, cnt = 0.f;
    const float inv_n = 1.0f / (float)N;
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < M; row += gridDim.x * blockDim.x) {
        const float* p = P + (long long)row * N;
        const float* y = Y + (long long)row * N;
        float l = 0.f, a = 0.f;
        for (int j = 0; j < N; ++j) {
            const float pj = p[j], d = pj - y[j];
            l += d * d;
            a += (rintf(pj) == rintf(y[j])) ? 1.f : 0.f;
            if (G) {
                float g = 2.0f * d * grad_scale;
                if (through_sigmoid) g *= pj * (1.0f - pj);
                G[(long long)row * N + j] = g;
            }
        }
        loss += l * inv_n;
        acc += a * inv_n;
        cnt += 1.f;
    }
    if (stats) block_add3(loss, acc, cnt, stats);
}

// One thread per row.  labels: int64 class ids (sparse) or one-hot/probability rows (dense).
__global__ __launch_bounds__(256) void softmax_xent_kernel(const float* __restrict__ Z,
                                                           const long long* __restrict__ sparse_labels,
                                                           const float* __restrict__ dense_labels,
                                                           float* __restrict__ G, int M, int C, float grad_scale,
                                                           float* stats) {
    float loss = 0.f, acc = 0.f, cnt = 0.f;
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < M; row += gridDim.x * blockDim.x) {
        const float* z = Z + (long long)row * C;
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
            label_arg = (int)sparse_labels[row];
            l = lse - z[label_arg];
            if (G) {
                for (int j = 0; j < C; ++j) {
                    const float sm = __expf(z[j] - mx) * inv;
                    G[(long long)row * C + j] = (sm - (j == label_arg ? 1.f : 0.f)) * grad_scale;
                }
            }
        } else {
            const float* y = dense_labels + (long long)row * C;
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
            if (G) {
                for (int j = 0; j < C; ++j) {
                    const float sm = __expf(z[j] - mx) * inv;
                    G[(long long)row * C + j] = (sm * ysum - y[j]) * grad_scale;
                }
            }
        }
        loss += l;
        acc += (arg == label_arg) ? 1.f : 0.f;
        cnt += 1.f;
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
    if (g > 2048) g = 2048;
    return g < 1 ? 1 : g;
}

}  // namespace

DTF_API int dtf_mse_loss(const float* P, const float* Y, float* G, int M, int N, float grad_scale,
                         int through_sigmoid, float* stats, hipStream_t stream) {
    if (M <= 0 || N <= 0) return 0;
    hipLaunchKernelGGL(mse_rows_kernel, dim3(row_grid(M)), dim3(256), 0, stream, P, Y, G, M, N, grad_scale,
                       through_sigmoid, stats);
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_softmax_xent(const float* Z, const long long* sparse_labels, const float* dense_labels, float* G,
                             int M, int C, float grad_scale, float* stats, hipStream_t stream) {
    if (M <= 0 || C <= 0) return 0;
    hipLaunchKernelGGL(softmax_xent_kernel, dim3(row_grid(M)), dim3(256), 0, stream, Z, sparse_labels,
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
