// Average pooling of NHWC tensors (Keras AveragePooling2D / GlobalAveragePooling2D): any window and strides, VALID or
// SAME padding.  As TensorFlow does, a window that hangs over the border averages over the positions inside the
// image.  Memory-bound elementwise kernels: one thread per 4 channels of an output pixel (forward) or of an input pixel
// (backward, which gathers from the windows covering it - no atomics); a scalar twin for channel counts that are not
// multiples of 4.
#include "common.h"

namespace {

struct AvgGeom {
    int B, H, W, C, PH, PW, SH, SW, OH, OW, pt, pl;
};

template <int VEC>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, AvgGeom g,
                                                          long long total) {
    const int CV = g.C / VEC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        long long pix = i / CV;
        const int ow = (int)(pix % g.OW);
        pix /= g.OW;
        const int oh = (int)(pix % g.OH);
        const int b = (int)(pix / g.OH);
        const int h0 = max(oh * g.SH - g.pt, 0), h1 = min(oh * g.SH - g.pt + g.PH, g.H);
        const int w0 = max(ow * g.SW - g.pl, 0), w1 = min(ow * g.SW - g.pl + g.PW, g.W);
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) {
                const float* p = x + (((long long)b * g.H + h) * g.W + w) * g.C + cv * VEC;
                if (VEC == 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] += v[e];
                } else {
                    acc[0] += p[0];
                }
            }
        const int cnt = (h1 - h0) * (w1 - w0);
        const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.f;
        float* q = y + i * VEC;
        if (VEC == 4) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = acc[e] * inv;
            *reinterpret_cast<f32x4*>(q) = o;
        } else {
            q[0] = acc[0] * inv;
        }
    }
}

__device__ __forceinline__ int ceil_div_pos(int a, int b) { return a <= 0 ? 0 : (a + b - 1) / b; }

template <int VEC>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, AvgGeom g,
                                                          long long total) {
    const int CV = g.C / VEC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        long long pix = i / CV;
        const int w = (int)(pix % g.W);
        pix /= g.W;
        const int h = (int)(pix % g.H);
        const int b = (int)(pix / g.H);
        // windows oh with oh*SH - pt <= h < oh*SH - pt + PH
        const int oh0 = ceil_div_pos(h + g.pt - g.PH + 1, g.SH), oh1 = min((h + g.pt) / g.SH, g.OH - 1);
        const int ow0 = ceil_div_pos(w + g.pl - g.PW + 1, g.SW), ow1 = min((w + g.pl) / g.SW, g.OW - 1);
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        for (int oh = oh0; oh <= oh1; ++oh) {
            const int ch = min(oh * g.SH - g.pt + g.PH, g.H) - max(oh * g.SH - g.pt, 0);
            for (int ow = ow0; ow <= ow1; ++ow) {
                const int cw = min(ow * g.SW - g.pl + g.PW, g.W) - max(ow * g.SW - g.pl, 0);
                const float inv = 1.0f / (float)(ch * cw);
                const float* p = dy + (((long long)b * g.OH + oh) * g.OW + ow) * g.C + cv * VEC;
                if (VEC == 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] += v[e] * inv;
                } else {
                    acc[0] += p[0] * inv;
                }
            }
        }
        float* q = dx + i * VEC;
        if (VEC == 4) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = acc[e];
            *reinterpret_cast<f32x4*>(q) = o;
        } else {
            q[0] = acc[0];
        }
    }
}

inline int avg_grid(long long total) {
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    return blocks < 1 ? 1 : (int)blocks;
}

inline bool avg_ok(const AvgGeom& g) {
    return g.B > 0 && g.H > 0 && g.W > 0 && g.C > 0 && g.PH > 0 && g.PW > 0 && g.SH > 0 && g.SW > 0 && g.OH > 0 && g.OW > 0 &&
           g.pt >= 0 && g.pl >= 0 && g.pt < g.PH && g.pl < g.PW;
}

}  // namespace

DTF_API int dtf_avgpool_fwd(const float* x, float* y, int B, int H, int W, int C, int PH, int PW, int SH, int SW, int OH, int OW,
                            int pt, int pl, hipStream_t stream) {
    const AvgGeom g = {B, H, W, C, PH, PW, SH, SW, OH, OW, pt, pl};
    if (!avg_ok(g)) return -3;
    const long long n = (long long)B * OH * OW * C;
    const bool vec = C % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    if (vec) hipLaunchKernelGGL(avgpool_fwd_kernel<4>, dim3(avg_grid(n / 4)), dim3(256), 0, stream, x, y, g, n / 4);
    else hipLaunchKernelGGL(avgpool_fwd_kernel<1>, dim3(avg_grid(n)), dim3(256), 0, stream, x, y, g, n);
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_avgpool_bwd(const float* dy, float* dx, int B, int H, int W, int C, int PH, int PW, int SH, int SW, int OH,
                            int OW, int pt, int pl, hipStream_t stream) {
    const AvgGeom g = {B, H, W, C, PH, PW, SH, SW, OH, OW, pt, pl};
    if (!avg_ok(g)) return -3;
    const long long n = (long long)B * H * W * C;
    const bool vec = C % 4 == 0 && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
    if (vec) hipLaunchKernelGGL(avgpool_bwd_kernel<4>, dim3(avg_grid(n / 4)), dim3(256), 0, stream, dy, dx, g, n / 4);
    else hipLaunchKernelGGL(avgpool_bwd_kernel<1>, dim3(avg_grid(n)), dim3(256), 0, stream, dy, dx, g, n);
    DTF_CHECK_LAUNCH();
    return 0;
}
