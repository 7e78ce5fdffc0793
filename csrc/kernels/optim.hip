// Apply-gradients kernels over a FLAT parameter range.
//
// Variables, gradients and optimizer slots of a model live in contiguous flat fp32
// buffers, so "apply all variables" and "apply my 1/N shard" are the same kernel on a
// different [begin, end) range: no pointer tables, 16-byte vector access, grid-stride.
//
// Adam follows the TensorFlow ApplyAdam formula used by tf.train.AdamOptimizer at
// /root/reference/example.py:168-170 (epsilon OUTSIDE the bias correction):
//     lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t),  t = global_step + 1
//     m   <- b1 m + (1 - b1) g
//     v   <- b2 v + (1 - b2) g^2
//     var <- var - lr_t m / (sqrt(v) + eps)
// global_step is device-resident (int64) so a captured hipGraph advances by itself.
//
// RMSProp and Adagrad are the other two rules a script of this family picks (tf.train.RMSPropOptimizer /
// tf.train.AdagradOptimizer, Keras 'rmsprop' / 'adagrad'); both keep a running sum of squared gradients per element:
//     RMSProp   a <- rho a + (1 - rho) g^2     u = lr g / D(a)     mom <- momentum mom + u ; var <- var - mom
//     Adagrad   a <- a + g^2                   var <- var - lr g / D(a)
//     Adadelta  a <- rho a + (1 - rho) g^2     u = g sqrt(d + eps) / sqrt(a + eps);  var <- var - lr u;
//               d <- rho d + (1 - rho) u^2     (TensorFlow's ApplyAdadelta and Keras' Adadelta agree)
// D(a) = sqrt(a + eps) in TensorFlow's kernels and sqrt(a) + eps in Keras' Python formulas: `eps_inside` selects.
#include "common.h"

namespace {

// Optional tail of an apply kernel: the LAST workgroup to finish advances global_step (and TF's beta-power
// variables), which saves a separate one-thread launch per training step.  Every workgroup has read *step_ptr
// before it takes its ticket, so the increment cannot be seen by a workgroup of the same launch.
struct Advance {
    long long* step;        // null: do not advance
    float* beta_powers;     // [2] or null
    unsigned int* ticket;   // device counter, zero between launches
    float b1, b2;
};

__device__ __forceinline__ void advance_if_last(const Advance& a) {
    if (!a.step) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        // No fence: nothing a workgroup wrote has to be visible to the last one - only the order "every workgroup
        // read the step, then the step changes" matters, and the ticket atomic after the barrier gives that.  (An
        // agent-scope release here made every workgroup write back its XCD's L2: 70 us on a 1.6 M-parameter model.)
        const unsigned int t = atomicAdd(a.ticket, 1u);
        if (t == gridDim.x - 1) {
            *a.ticket = 0u;
            *a.step += 1;
            if (a.beta_powers) {
                a.beta_powers[0] *= a.b1;
                a.beta_powers[1] *= a.b2;
            }
        }
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ var, float* __restrict__ m,
                                                   float* __restrict__ v, const float* __restrict__ g,
                                                   long long n, float lr, float b1, float b2, float eps,
                                                   float grad_scale, const long long* step_ptr, const Advance adv) {
    // lr_t once per workgroup: two powf cost more than the update of the few elements a thread owns
    __shared__ float lr_t_shared;
    if (threadIdx.x == 0) {
        const float t = (float)(*step_ptr + 1);
        lr_t_shared = lr * sqrtf(1.0f - powf(b2, t)) / (1.0f - powf(b1, t));
    }
    __syncthreads();
    const float lr_t = lr_t_shared;
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4* var4 = reinterpret_cast<f32x4*>(var);
    f32x4* m4 = reinterpret_cast<f32x4*>(m);
    f32x4* v4 = reinterpret_cast<f32x4*>(v);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    for (long long i = tid; i < n4; i += stride) {
        f32x4 gg = g4[i], mm = m4[i], vv = v4[i], w = var4[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gj = gg[j] * grad_scale;
            mm[j] = b1 * mm[j] + omb1 * gj;
            vv[j] = b2 * vv[j] + omb2 * gj * gj;
            w[j] -= lr_t * mm[j] / (sqrtf(vv[j]) + eps);
        }
        m4[i] = mm;
        v4[i] = vv;
        var4[i] = w;
    }
    for (long long i = (n4 << 2) + tid; i < n; i += stride) {
        const float gj = g[i] * grad_scale;
        const float mj = b1 * m[i] + omb1 * gj;
        const float vj = b2 * v[i] + omb2 * gj * gj;
        m[i] = mj;
        v[i] = vj;
        var[i] -= lr_t * mj / (sqrtf(vj) + eps);
    }
    advance_if_last(adv);
}

// var <- var - lr * (momentum-filtered) g.  mom == nullptr: plain SGD (tf.train.GradientDescentOptimizer).
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ var, float* __restrict__ mom,
                                                  const float* __restrict__ g, long long n, float lr,
                                                  float momentum, float grad_scale, const Advance adv) {
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4* var4 = reinterpret_cast<f32x4*>(var);
    f32x4* mo4 = reinterpret_cast<f32x4*>(mom);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    for (long long i = tid; i < n4; i += stride) {
        f32x4 gg = g4[i], w = var4[i];
        if (mom) {
            f32x4 a = mo4[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = momentum * a[j] + gg[j] * grad_scale;
                w[j] -= lr * a[j];
            }
            mo4[i] = a;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] -= lr * gg[j] * grad_scale;
        }
        var4[i] = w;
    }
    for (long long i = (n4 << 2) + tid; i < n; i += stride) {
        float gj = g[i] * grad_scale;
        if (mom) {
            gj = momentum * mom[i] + gj;
            mom[i] = gj;
        }
        var[i] -= lr * gj;
    }
    advance_if_last(adv);
}

// RULE 0: RMSProp (mom may be null: no momentum slot), RULE 1: Adagrad (rho, momentum, mom unused), RULE 2: Adadelta
// (mom holds the accumulated squared updates d).
template <int RULE>
__global__ __launch_bounds__(256) void accum_rule_kernel(float* __restrict__ var, float* __restrict__ acc,
                                                         float* __restrict__ mom, const float* __restrict__ g,
                                                         long long n, float lr, float rho, float momentum, float eps,
                                                         int eps_inside, float grad_scale, const Advance adv) {
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const float omr = 1.0f - rho;
    auto one = [&](float gj, float& a, float& mo, float& w) {
        gj *= grad_scale;
        if (RULE == 2) {
            a = rho * a + omr * gj * gj;
            const float u = gj * sqrtf(mo + eps) / sqrtf(a + eps);
            mo = rho * mo + omr * u * u;
            w -= lr * u;
            return;
        }
        a = RULE == 0 ? rho * a + omr * gj * gj : a + gj * gj;
        const float d = eps_inside ? sqrtf(a + eps) : sqrtf(a) + eps;
        float u = lr * gj / d;
        if (RULE == 0 && mom) {
            mo = momentum * mo + u;
            u = mo;
        }
        w -= u;
    };
    f32x4* var4 = reinterpret_cast<f32x4*>(var);
    f32x4* a4 = reinterpret_cast<f32x4*>(acc);
    f32x4* mo4 = reinterpret_cast<f32x4*>(mom);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    for (long long i = tid; i < n4; i += stride) {
        f32x4 gg = g4[i], aa = a4[i], w = var4[i], mm = {0.f, 0.f, 0.f, 0.f};
        if (RULE != 1 && mom) mm = mo4[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = aa[j], mo = mm[j], wj = w[j];
            one(gg[j], a, mo, wj);
            aa[j] = a;
            mm[j] = mo;
            w[j] = wj;
        }
        a4[i] = aa;
        if (RULE != 1 && mom) mo4[i] = mm;
        var4[i] = w;
    }
    for (long long i = (n4 << 2) + tid; i < n; i += stride) {
        float a = acc[i], w = var[i], mo = (RULE != 1 && mom) ? mom[i] : 0.f;
        one(g[i], a, mo, w);
        acc[i] = a;
        if (RULE != 1 && mom) mom[i] = mo;
        var[i] = w;
    }
    advance_if_last(adv);
}

// global_step += 1 and the TF non-slot variables beta1_power / beta2_power *= beta.
__global__ void step_advance_kernel(long long* step_ptr, float* beta_powers, float b1, float b2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        *step_ptr += 1;
        if (beta_powers) {
            beta_powers[0] *= b1;
            beta_powers[1] *= b2;
        }
    }
}

inline int flat_grid(long long n) {
    // Every workgroup ends with one returning atomic on the ticket word (~11 ns each, serialised in L2): models of a
    // few million parameters are done streaming before 2048 tickets are, so they get 512 workgroups (two per CU).
    long long g = (n / 4 + 255) / 256;
    const long long cap = n > (1ll << 26) ? 2048 : 512;
    if (g > cap) g = cap;
    return g < 1 ? 1 : (int)g;
}

}  // namespace

// advance_step (nullable) is incremented once by the launch's last workgroup; it needs `ticket`, a zeroed device
// counter that only these kernels touch.
DTF_API int dtf_adam_apply(float* var, float* m, float* v, const float* g, long long n, float lr, float b1,
                           float b2, float eps, float grad_scale, const long long* step_ptr, long long* advance_step,
                           float* beta_powers, unsigned int* ticket, hipStream_t stream) {
    if (n <= 0) return 0;
    if (advance_step && !ticket) return -3;
    const Advance adv = {advance_step, beta_powers, ticket, b1, b2};
    if ((reinterpret_cast<uintptr_t>(var) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
         reinterpret_cast<uintptr_t>(g)) & 15)
        return -2;  // flat buffers and shard boundaries are 16-byte aligned by construction
    hipLaunchKernelGGL(adam_kernel, dim3(flat_grid(n)), dim3(256), 0, stream, var, m, v, g, n, lr, b1, b2, eps,
                       grad_scale, step_ptr, adv);
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_sgd_apply(float* var, float* mom, const float* g, long long n, float lr, float momentum,
                          float grad_scale, long long* advance_step, unsigned int* ticket, hipStream_t stream) {
    if (n <= 0) return 0;
    if (advance_step && !ticket) return -3;
    const Advance adv = {advance_step, nullptr, ticket, 0.f, 0.f};
    if ((reinterpret_cast<uintptr_t>(var) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(mom)) & 15)
        return -2;
    hipLaunchKernelGGL(sgd_kernel, dim3(flat_grid(n)), dim3(256), 0, stream, var, mom, g, n, lr, momentum,
                       grad_scale, adv);
    DTF_CHECK_LAUNCH();
    return 0;
}

// mom may be null (momentum 0: the slot is not kept)
DTF_API int dtf_rmsprop_apply(float* var, float* ms, float* mom, const float* g, long long n, float lr, float rho,
                              float momentum, float eps, int eps_inside, float grad_scale, long long* advance_step,
                              unsigned int* ticket, hipStream_t stream) {
    if (n <= 0) return 0;
    if (advance_step && !ticket) return -3;
    const Advance adv = {advance_step, nullptr, ticket, 0.f, 0.f};
    if ((reinterpret_cast<uintptr_t>(var) | reinterpret_cast<uintptr_t>(ms) | reinterpret_cast<uintptr_t>(mom) |
         reinterpret_cast<uintptr_t>(g)) & 15)
        return -2;
    hipLaunchKernelGGL(accum_rule_kernel<0>, dim3(flat_grid(n)), dim3(256), 0, stream, var, ms, mom, g, n, lr, rho,
                       momentum, eps, eps_inside, grad_scale, adv);
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_adadelta_apply(float* var, float* acc, float* acc_update, const float* g, long long n, float lr, float rho,
                               float eps, float grad_scale, long long* advance_step, unsigned int* ticket, hipStream_t stream) {
    if (n <= 0) return 0;
    if (advance_step && !ticket) return -3;
    if (!acc_update) return -3;
    const Advance adv = {advance_step, nullptr, ticket, 0.f, 0.f};
    if ((reinterpret_cast<uintptr_t>(var) | reinterpret_cast<uintptr_t>(acc) | reinterpret_cast<uintptr_t>(acc_update) |
         reinterpret_cast<uintptr_t>(g)) & 15)
        return -2;
    hipLaunchKernelGGL(accum_rule_kernel<2>, dim3(flat_grid(n)), dim3(256), 0, stream, var, acc, acc_update, g, n, lr, rho,
                       0.f, eps, 1, grad_scale, adv);
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_adagrad_apply(float* var, float* acc, const float* g, long long n, float lr, float eps, int eps_inside,
                              float grad_scale, long long* advance_step, unsigned int* ticket, hipStream_t stream) {
    if (n <= 0) return 0;
    if (advance_step && !ticket) return -3;
    const Advance adv = {advance_step, nullptr, ticket, 0.f, 0.f};
    if ((reinterpret_cast<uintptr_t>(var) | reinterpret_cast<uintptr_t>(acc) | reinterpret_cast<uintptr_t>(g)) & 15)
        return -2;
    hipLaunchKernelGGL(accum_rule_kernel<1>, dim3(flat_grid(n)), dim3(256), 0, stream, var, acc, nullptr, g, n, lr,
                       0.f, 0.f, eps, eps_inside, grad_scale, adv);
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_step_advance(long long* step_ptr, float* beta_powers, float b1, float b2, hipStream_t stream) {
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, stream, step_ptr, beta_powers, b1, b2);
    DTF_CHECK_LAUNCH();
    return 0;
}
