// Batch normalisation over the last axis of a row-major [R, C] activation (NHWC feature maps flattened to rows, or
// the output of a dense layer): Keras' BatchNormalization(axis=-1) in training and inference mode, with the gradient.
//
//   training   mean_c, var_c over the R rows (biased);  y = gamma (x - mean) / sqrt(var + eps) + beta
//              moving_mean <- m moving_mean + (1 - m) mean,  moving_variance likewise
//   inference  the same affine map from the moving statistics
//   backward   dbeta = sum dy;  dgamma = sum dy xhat;  dx = gamma inv_std (dy - dbeta / R - xhat dgamma / R)
//
// Both directions are memory-bound and run as: per-channel partial sums over row blocks (16-byte loads when C is a
// multiple of 4, rows dealt to the lanes a block has left after its channels) -> one small kernel that adds the
// partials in fp64 and in a fixed order (results do not depend on the launch shape's timing) and derives the
// per-channel coefficients -> one elementwise pass.  X is read twice per direction, nothing else is materialised.
#include "common.h"

namespace {

constexpr int BN_MAX_BLOCKS = 2048;      // row blocks of the partial-sum pass (eight per CU)

// slab[g][0][c] / slab[g][1][c]: forward  sum x, sum x^2;  backward  sum dy, sum dy xhat
template <int VEC, bool BWD>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ X, const float* __restrict__ DY,
                                                         const float* __restrict__ mean, const float* __restrict__ inv_std,
                                                         float* __restrict__ slab, long long R, int C, long long rows_per_block) {
    __shared__ float red[2][256 * VEC];
    const int CV = C / VEC;                      // vector columns
    const int cw = CV < 256 ? CV : 256;          // vector columns a pass of the block covers
    const int rl = 256 / cw;                     // row lanes
    const int tid = threadIdx.x;
    const int cl = tid % cw, rlane = tid / cw;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    float* out = slab + (long long)blockIdx.x * 2 * C;
    for (int cbase = 0; cbase < CV; cbase += cw) {
        const int cv = cbase + cl;
        const bool live = rlane < rl && cv < CV;
        float a0[VEC], a1[VEC], mu[VEC], is[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            a0[e] = a1[e] = 0.f;
            mu[e] = (BWD && live) ? mean[cv * VEC + e] : 0.f;
            is[e] = (BWD && live) ? inv_std[cv * VEC + e] : 0.f;
        }
        if (live) {
            auto one_row = [&](const float (&x)[VEC], const float (&g)[VEC]) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    if (BWD) {
                        a0[e] += g[e];
                        a1[e] += g[e] * (x[e] - mu[e]) * is[e];
                    } else {
                        a0[e] += x[e];
                        a1[e] += x[e] * x[e];
                    }
                }
            };
            auto fetch = [&](long long r, float (&x)[VEC], float (&g)[VEC]) {
                if (VEC == 4) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(X + r * C + cv * 4);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) x[e] = xv[e];
                    if (BWD) {
                        const f32x4 gv = *reinterpret_cast<const f32x4*>(DY + r * C + cv * 4);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) g[e] = gv[e];
                    }
                } else {
                    x[0] = X[r * C + cv];
                    if (BWD) g[0] = DY[r * C + cv];
                }
            };
            long long r = r0 + rlane;
            // four rows per trip: their loads are in flight together (a thread's rows are rl apart)
            for (; r + 3ll * rl < r1; r += 4ll * rl) {
                float x[4][VEC], g[4][VEC];
#pragma unroll
                for (int u = 0; u < 4; ++u) fetch(r + (long long)u * rl, x[u], g[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) one_row(x[u], g[u]);
            }
            for (; r < r1; r += rl) {
                float x[VEC], g[VEC];
                fetch(r, x, g);
                one_row(x, g);
            }
        }
        __syncthreads();          // the previous pass has been read out
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            red[0][tid * VEC + e] = a0[e];
            red[1][tid * VEC + e] = a1[e];
        }
        __syncthreads();
        if (rlane == 0 && cv < CV) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float s0 = 0.f, s1 = 0.f;
                for (int k = 0; k < rl; ++k) {          // fixed order
                    s0 += red[0][(k * cw + cl) * VEC + e];
                    s1 += red[1][(k * cw + cl) * VEC + e];
                }
                out[cv * VEC + e] = s0;
                out[C + cv * VEC + e] = s1;
            }
        }
    }
}

// Sum of the G partial pairs of one channel, in fp64 and in a fixed order.  A block of 256 threads takes 8 channels:
// thirty-two lanes per channel add every thirty-second partial (the chain of dependent additions is what this small
// kernel's time consists of: G / 32 links), then lane 0 adds the thirty-two.  Returns the channel for the thread
// that holds the result, -1 for all others.
constexpr int BN_FIN_CH = 8, BN_FIN_LANES = 32;

__device__ __forceinline__ int bn_sum_partials(const float* __restrict__ slab, int G, int C, double& s0, double& s1) {
    __shared__ double part[2][BN_FIN_LANES][BN_FIN_CH];
    const int cl = threadIdx.x % BN_FIN_CH, gl = threadIdx.x / BN_FIN_CH;
    const int c = blockIdx.x * BN_FIN_CH + cl;
    double a0 = 0.0, a1 = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int g = gl; g < G; g += BN_FIN_LANES) {
            a0 += (double)slab[(long long)g * 2 * C + c];
            a1 += (double)slab[(long long)g * 2 * C + C + c];
        }
    }
    part[0][gl][cl] = a0;
    part[1][gl][cl] = a1;
    __syncthreads();
    if (gl != 0 || c >= C) return -1;
    s0 = s1 = 0.0;
#pragma unroll
    for (int k = 0; k < BN_FIN_LANES; ++k) {
        s0 += part[0][k][cl];
        s1 += part[1][k][cl];
    }
    return c;
}

// Forward: statistics, coefficients of y = x scale + shift, moving averages.
__global__ __launch_bounds__(256) void bn_finish_fwd_kernel(const float* __restrict__ slab, int G, int C, long long R,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, float momentum, float* __restrict__ moving_mean,
                                                            float* __restrict__ moving_var, float* __restrict__ save_mean,
                                                            float* __restrict__ save_inv_std, float* __restrict__ scale,
                                                            float* __restrict__ shift) {
    double s0, s1;
    const int c = bn_sum_partials(slab, G, C, s0, s1);
    if (c < 0) return;
    const double m = s0 / (double)R;
    double v = s1 / (double)R - m * m;
    if (v < 0.0) v = 0.0;
    const float inv = (float)(1.0 / sqrt(v + (double)eps));
    const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
    save_mean[c] = (float)m;
    save_inv_std[c] = inv;
    scale[c] = ga * inv;
    shift[c] = be - (float)m * ga * inv;
    moving_mean[c] = momentum * moving_mean[c] + (1.0f - momentum) * (float)m;
    moving_var[c] = momentum * moving_var[c] + (1.0f - momentum) * (float)v;
}

// Inference: the coefficients from the moving statistics.
__global__ __launch_bounds__(256) void bn_infer_coeff_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, const float* __restrict__ moving_mean,
                                                             const float* __restrict__ moving_var, float* __restrict__ scale,
                                                             float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float inv = 1.0f / sqrtf(moving_var[c] + eps);
    const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
    scale[c] = ga * inv;
    shift[c] = be - moving_mean[c] * ga * inv;
}

// Backward: dgamma, dbeta and the coefficients of dx = a (dy - b - xhat c).
__global__ __launch_bounds__(256) void bn_finish_bwd_kernel(const float* __restrict__ slab, int G, int C, long long R,
                                                            const float* __restrict__ gamma, const float* __restrict__ inv_std,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ coef /* [3][C]: a, b, c */) {
    double s0, s1;
    const int c = bn_sum_partials(slab, G, C, s0, s1);
    if (c < 0) return;
    if (dbeta) dbeta[c] = (float)s0;
    if (dgamma) dgamma[c] = (float)s1;
    coef[c] = (gamma ? gamma[c] : 1.0f) * inv_std[c];
    coef[C + c] = (float)(s0 / (double)R);
    coef[2 * C + c] = (float)(s1 / (double)R);
}

// FWD: Y = X scale + shift.  BWD: DX = a (DY - b - xhat c), xhat = (X - mean) inv_std.
template <int VEC, bool BWD>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ X, const float* __restrict__ DY,
                                                       float* __restrict__ OUT, long long total_v, int C,
                                                       const float* __restrict__ p0, const float* __restrict__ p1,
                                                       const float* __restrict__ mean, const float* __restrict__ inv_std,
                                                       const float* __restrict__ coef) {
    const int CV = C / VEC;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total_v; i += stride) {
        const int cv = (int)(i % CV);
        if (VEC == 4) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(X + i * 4);
            f32x4 o;
            if (BWD) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(DY + i * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = cv * 4 + e;
                    const float xhat = (x[e] - mean[c]) * inv_std[c];
                    o[e] = coef[c] * (g[e] - coef[C + c] - xhat * coef[2 * C + c]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = x[e] * p0[cv * 4 + e] + p1[cv * 4 + e];
            }
            *reinterpret_cast<f32x4*>(OUT + i * 4) = o;
        } else {
            const float x = X[i];
            if (BWD) {
                const float xhat = (x - mean[cv]) * inv_std[cv];
                OUT[i] = coef[cv] * (DY[i] - coef[C + cv] - xhat * coef[2 * C + cv]);
            } else {
                OUT[i] = x * p0[cv] + p1[cv];
            }
        }
    }
}

inline int bn_blocks(long long R) {
    // enough row blocks to fill the chip, each with at least 64 rows
    long long g = (R + 63) / 64;
    if (g > BN_MAX_BLOCKS) g = BN_MAX_BLOCKS;
    return g < 1 ? 1 : (int)g;
}

inline int elementwise_grid(long long total_v) {
    long long g = (total_v + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    return g < 1 ? 1 : (int)g;
}

}  // namespace

// floats of scratch a call needs: the partial sums, then [scale | shift] (forward) or [a | b | c] (backward)
DTF_API long long dtf_batchnorm_workspace_floats(long long R, int C) {
    return (long long)bn_blocks(R) * 2 * C + 3ll * C;
}

// training != 0: batch statistics (saved in save_mean / save_inv_std for the backward pass) and moving-average update;
// else the moving statistics.  gamma / beta may be null (scale=False / center=False).
DTF_API int dtf_batchnorm_fwd(const float* X, float* Y, long long R, int C, const float* gamma, const float* beta,
                              float* moving_mean, float* moving_var, float momentum, float eps, int training,
                              float* save_mean, float* save_inv_std, float* workspace, hipStream_t stream) {
    if (R <= 0 || C <= 0) return 0;
    const bool vec = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15) == 0;
    const int G = bn_blocks(R);
    float* slab = workspace;
    float* scale = workspace + (long long)G * 2 * C;
    float* shift = scale + C;
    const int cgrid = (C + 255) / 256;
    if (training) {
        const long long rpb = (R + G - 1) / G;
        if (vec)
            hipLaunchKernelGGL((bn_partial_kernel<4, false>), dim3(G), dim3(256), 0, stream, X, nullptr, nullptr, nullptr, slab, R,
                               C, rpb);
        else
            hipLaunchKernelGGL((bn_partial_kernel<1, false>), dim3(G), dim3(256), 0, stream, X, nullptr, nullptr, nullptr, slab, R,
                               C, rpb);
        DTF_CHECK_LAUNCH();
        hipLaunchKernelGGL(bn_finish_fwd_kernel, dim3((C + BN_FIN_CH - 1) / BN_FIN_CH), dim3(256), 0, stream, slab, G, C, R, gamma, beta, eps, momentum,
                           moving_mean, moving_var, save_mean, save_inv_std, scale, shift);
    } else {
        hipLaunchKernelGGL(bn_infer_coeff_kernel, dim3(cgrid), dim3(256), 0, stream, C, gamma, beta, eps, moving_mean,
                           moving_var, scale, shift);
    }
    DTF_CHECK_LAUNCH();
    const long long total = R * C;
    if (vec)
        hipLaunchKernelGGL((bn_apply_kernel<4, false>), dim3(elementwise_grid(total / 4)), dim3(256), 0, stream, X, nullptr, Y,
                           total / 4, C, scale, shift, nullptr, nullptr, nullptr);
    else
        hipLaunchKernelGGL((bn_apply_kernel<1, false>), dim3(elementwise_grid(total)), dim3(256), 0, stream, X, nullptr, Y, total,
                           C, scale, shift, nullptr, nullptr, nullptr);
    DTF_CHECK_LAUNCH();
    return 0;
}

// dgamma / dbeta / DX may each be null (DX: the layer is the first of its model).
DTF_API int dtf_batchnorm_bwd(const float* X, const float* DY, float* DX, long long R, int C, const float* gamma,
                              const float* save_mean, const float* save_inv_std, float* dgamma, float* dbeta,
                              float* workspace, hipStream_t stream) {
    if (R <= 0 || C <= 0) return 0;
    const bool vec = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(DY) |
                                       reinterpret_cast<uintptr_t>(DX)) & 15) == 0;
    const int G = bn_blocks(R);
    float* slab = workspace;
    float* coef = workspace + (long long)G * 2 * C;
    const long long rpb = (R + G - 1) / G;
    if (vec)
        hipLaunchKernelGGL((bn_partial_kernel<4, true>), dim3(G), dim3(256), 0, stream, X, DY, save_mean, save_inv_std, slab, R, C,
                           rpb);
    else
        hipLaunchKernelGGL((bn_partial_kernel<1, true>), dim3(G), dim3(256), 0, stream, X, DY, save_mean, save_inv_std, slab, R, C,
                           rpb);
    DTF_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_finish_bwd_kernel, dim3((C + BN_FIN_CH - 1) / BN_FIN_CH), dim3(256), 0, stream, slab, G, C, R, gamma, save_inv_std,
                       dgamma, dbeta, coef);
    DTF_CHECK_LAUNCH();
    if (!DX) return 0;
    const long long total = R * C;
    if (vec)
        hipLaunchKernelGGL((bn_apply_kernel<4, true>), dim3(elementwise_grid(total / 4)), dim3(256), 0, stream, X, DY, DX,
                           total / 4, C, nullptr, nullptr, save_mean, save_inv_std, coef);
    else
        hipLaunchKernelGGL((bn_apply_kernel<1, true>), dim3(elementwise_grid(total)), dim3(256), 0, stream, X, DY, DX, total, C,
                           nullptr, nullptr, save_mean, save_inv_std, coef);
    DTF_CHECK_LAUNCH();
    return 0;
}
