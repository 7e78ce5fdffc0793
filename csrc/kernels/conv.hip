// Convolution support kernels (NHWC): patch matrix build / scatter-back and max pooling.
//
// conv2d runs as "patch matrix x filter matrix" on the MFMA GEMM kernels: a Keras/TF filter variable
// [KH, KW, Cin, Cout] IS the row-major [KH*KW*Cin, Cout] GEMM operand, and the GEMM output
// [B*OH*OW, Cout] IS the NHWC activation, so no transposes are ever materialised and the fused
// bias / activation / bias-gradient epilogues of the dense path apply unchanged.  With 288 GB of HBM
// the patch matrix is kept for the backward pass (wgrad reads it again) instead of being rebuilt.
#include "common.h"

namespace {

struct ConvGeom {
    int B, H, W, C;        // input
    int KH, KW, OH, OW;    // window, output
    int pad_t, pad_l, sh, sw;
};

// cols[m][(kh*KW + kw)*C + c] = x[b][oh*sh - pad_t + kh][ow*sw - pad_l + kw][c]  (zero outside the image)
template <int V>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, float* __restrict__ cols,
                                                     const ConvGeom g, long long total) {
    const int K = g.KH * g.KW * g.C;
    const int KV = K / V;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long m = idx / KV;
        const int k = (int)(idx - m * KV) * V;
        const int c = k % g.C;
        const int kw = (k / g.C) % g.KW;
        const int kh = k / (g.C * g.KW);
        const int ow = (int)(m % g.OW);
        const int oh = (int)((m / g.OW) % g.OH);
        const int b = (int)(m / ((long long)g.OW * g.OH));
        const int ih = oh * g.sh - g.pad_t + kh, iw = ow * g.sw - g.pad_l + kw;
        const bool in = ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
        const long long src = (((long long)b * g.H + ih) * g.W + iw) * g.C + c;
        if (V == 4) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (in) v = *reinterpret_cast<const f32x4*>(x + src);
            *reinterpret_cast<f32x4*>(cols + m * K + k) = v;
        } else {
            cols[m * K + k] = in ? x[src] : 0.f;
        }
    }
}

// Single-channel images (the first convolution of an MNIST network): one thread per (output pixel, kernel row) writes
// the KW consecutive patch entries of that row, so the pixel decode - integer divisions - is paid once per KW elements.
__global__ __launch_bounds__(256) void im2col_c1_kernel(const float* __restrict__ x, float* __restrict__ cols, const ConvGeom g,
                                                        long long total) {
    const int K = g.KH * g.KW;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long m = idx / g.KH;
        const int kh = (int)(idx - m * g.KH);
        const int ow = (int)(m % g.OW);
        const int oh = (int)((m / g.OW) % g.OH);
        const int b = (int)(m / ((long long)g.OW * g.OH));
        const int ih = oh * g.sh - g.pad_t + kh;
        const int iw0 = ow * g.sw - g.pad_l;
        const bool row_in = ih >= 0 && ih < g.H;
        const float* src = x + ((long long)b * g.H + ih) * g.W;
        float* dst = cols + m * K + kh * g.KW;
        for (int kw = 0; kw < g.KW; ++kw) {
            const int iw = iw0 + kw;
            dst[kw] = (row_in && iw >= 0 && iw < g.W) ? src[iw] : 0.f;
        }
    }
}

// dx[b][ih][iw][c] = sum over (kh, kw) of dcols[(b, oh, ow)][(kh*KW + kw)*C + c] with oh*sh - pad_t + kh == ih (same for w);
// optionally times the producing layer's ReLU gradient (yprev > 0).
template <int V>
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcols, float* __restrict__ dx,
                                                     const float* __restrict__ yprev, const ConvGeom g, long long total) {
    const int K = g.KH * g.KW * g.C;
    const int CV = g.C / V;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % CV) * V;
        const long long pix = idx / CV;
        const int iw = (int)(pix % g.W);
        const int ih = (int)((pix / g.W) % g.H);
        const int b = (int)(pix / ((long long)g.W * g.H));
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
        for (int kh = 0; kh < g.KH; ++kh) {
            const int th = ih + g.pad_t - kh;
            if (th < 0 || th % g.sh) continue;
            const int oh = th / g.sh;
            if (oh >= g.OH) continue;
            for (int kw = 0; kw < g.KW; ++kw) {
                const int tw = iw + g.pad_l - kw;
                if (tw < 0 || tw % g.sw) continue;
                const int ow = tw / g.sw;
                if (ow >= g.OW) continue;
                const long long m = ((long long)b * g.OH + oh) * g.OW + ow;
                const float* src = dcols + m * K + (kh * g.KW + kw) * g.C + c;
                if (V == 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] += v[j];
                } else {
                    acc[0] += src[0];
                }
            }
        }
        const long long dst = pix * g.C + c;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float r = acc[j];
            if (yprev) r = yprev[dst + j] > 0.f ? r : 0.f;
            dx[dst + j] = r;
        }
    }
}

struct PoolGeom {
    int B, H, W, C, PH, PW, OH, OW;   // non-overlapping windows (stride == window), VALID padding
};

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          unsigned char* __restrict__ arg, const PoolGeom g, long long total) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % g.C);
        const long long pix = idx / g.C;
        const int ow = (int)(pix % g.OW);
        const int oh = (int)((pix / g.OW) % g.OH);
        const int b = (int)(pix / ((long long)g.OW * g.OH));
        float best = -INFINITY;
        int bi = 0;
        for (int i = 0; i < g.PH; ++i)
            for (int j = 0; j < g.PW; ++j) {
                const float v = x[(((long long)b * g.H + oh * g.PH + i) * g.W + ow * g.PW + j) * g.C + c];
                if (v > best) { best = v; bi = i * g.PW + j; }
            }
        y[idx] = best;
        arg[idx] = (unsigned char)bi;
    }
}

// dx = dy routed to each window's arg-max position; optionally times the ReLU gradient of the pooled tensor (x > 0)
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg,
                                                          const float* __restrict__ x_relu, float* __restrict__ dx,
                                                          const PoolGeom g, long long total) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % g.C);
        const long long pix = idx / g.C;
        const int iw = (int)(pix % g.W);
        const int ih = (int)((pix / g.W) % g.H);
        const int b = (int)(pix / ((long long)g.W * g.H));
        const int oh = ih / g.PH, ow = iw / g.PW;
        float r = 0.f;
        if (oh < g.OH && ow < g.OW) {
            const long long o = (((long long)b * g.OH + oh) * g.OW + ow) * g.C + c;
            if ((int)arg[o] == (ih - oh * g.PH) * g.PW + (iw - ow * g.PW)) r = dy[o];
        }
        if (x_relu && !(x_relu[idx] > 0.f)) r = 0.f;
        dx[idx] = r;
    }
}


// General max pooling: any window, any strides, VALID or SAME padding (TensorFlow's rule: padded positions never win,
// the extra padding goes to the bottom / right).  Forward: one thread per output element, arg = window position of the
// maximum.  Backward: one thread per INPUT element sums the gradients of the windows that cover it and chose it
// (overlapping windows when a stride is smaller than its window) - a gather, so no atomics and a fixed summation order.
struct PoolGeomS {
    int B, H, W, C, PH, PW, SH, SW, OH, OW, pad_t, pad_l;
};

__global__ __launch_bounds__(256) void maxpool_strided_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  unsigned char* __restrict__ arg, const PoolGeomS g,
                                                                  long long total) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % g.C);
        const long long pix = idx / g.C;
        const int ow = (int)(pix % g.OW);
        const int oh = (int)((pix / g.OW) % g.OH);
        const int b = (int)(pix / ((long long)g.OW * g.OH));
        float best = -INFINITY;
        int bi = 0;
        for (int i = 0; i < g.PH; ++i) {
            const int ih = oh * g.SH - g.pad_t + i;
            if (ih < 0 || ih >= g.H) continue;
            for (int j = 0; j < g.PW; ++j) {
                const int iw = ow * g.SW - g.pad_l + j;
                if (iw < 0 || iw >= g.W) continue;
                const float v = x[(((long long)b * g.H + ih) * g.W + iw) * g.C + c];
                if (v > best) { best = v; bi = i * g.PW + j; }
            }
        }
        y[idx] = best;
        arg[idx] = (unsigned char)bi;
    }
}

__global__ __launch_bounds__(256) void maxpool_strided_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg,
                                                                  const float* __restrict__ x_relu, float* __restrict__ dx,
                                                                  const PoolGeomS g, long long total) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % g.C);
        const long long pix = idx / g.C;
        const int iw = (int)(pix % g.W);
        const int ih = (int)((pix / g.W) % g.H);
        const int b = (int)(pix / ((long long)g.W * g.H));
        // windows (oh, ow) with oh * SH - pad_t <= ih < oh * SH - pad_t + PH
        const int py = ih + g.pad_t, px = iw + g.pad_l;
        const int oh_lo = py - g.PH + 1 > 0 ? (py - g.PH + g.SH) / g.SH : 0;
        const int ow_lo = px - g.PW + 1 > 0 ? (px - g.PW + g.SW) / g.SW : 0;
        const int oh_hi = min(py / g.SH, g.OH - 1), ow_hi = min(px / g.SW, g.OW - 1);
        float r = 0.f;
        for (int oh = oh_lo; oh <= oh_hi; ++oh)
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const long long o = (((long long)b * g.OH + oh) * g.OW + ow) * g.C + c;
                if ((int)arg[o] == (py - oh * g.SH) * g.PW + (px - ow * g.SW)) r += dy[o];
            }
        if (x_relu && !(x_relu[idx] > 0.f)) r = 0.f;
        dx[idx] = r;
    }
}

// Four channels per thread (C % 4 == 0): 16-byte accesses and a quarter of the index arithmetic.
__global__ __launch_bounds__(256) void maxpool_fwd4_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           unsigned char* __restrict__ arg, const PoolGeom g, long long total) {
    const int CV = g.C / 4;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % CV) * 4;
        const long long pix = idx / CV;
        const int ow = (int)(pix % g.OW);
        const int oh = (int)((pix / g.OW) % g.OH);
        const int b = (int)(pix / ((long long)g.OW * g.OH));
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
        for (int i = 0; i < g.PH; ++i)
            for (int j = 0; j < g.PW; ++j) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(
                    x + (((long long)b * g.H + oh * g.PH + i) * g.W + ow * g.PW + j) * g.C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] > best[e]) { best[e] = v[e]; bi[e] = i * g.PW + j; }
            }
        *reinterpret_cast<f32x4*>(y + pix * g.C + c) = best;
        *reinterpret_cast<uchar4*>(arg + pix * g.C + c) =
            make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd4_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg,
                                                           const float* __restrict__ x_relu, float* __restrict__ dx,
                                                           const PoolGeom g, long long total, int mask_pooled) {
    const int CV = g.C / 4;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % CV) * 4;
        const long long pix = idx / CV;
        const int iw = (int)(pix % g.W);
        const int ih = (int)((pix / g.W) % g.H);
        const int b = (int)(pix / ((long long)g.W * g.H));
        const int oh = ih / g.PH, ow = iw / g.PW;
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (oh < g.OH && ow < g.OW) {
            const long long o = (((long long)b * g.OH + oh) * g.OW + ow) * g.C + c;
            const int me = (ih - oh * g.PH) * g.PW + (iw - ow * g.PW);
            const uchar4 a = *reinterpret_cast<const uchar4*>(arg + o);
            const f32x4 d = *reinterpret_cast<const f32x4*>(dy + o);
            r[0] = (int)a.x == me ? d[0] : 0.f;
            r[1] = (int)a.y == me ? d[1] : 0.f;
            r[2] = (int)a.z == me ? d[2] : 0.f;
            r[3] = (int)a.w == me ? d[3] : 0.f;
        }
        const long long dst = pix * g.C + c;
        if (x_relu) {
            // the ReLU mask: from the full-resolution ReLU output, or (mask_pooled) from the POOLED tensor - at the
            // arg-max position, the only one with a gradient, the two hold the same value
            const bool inside = oh < g.OH && ow < g.OW;
            const long long src = mask_pooled ? (inside ? (((long long)b * g.OH + oh) * g.OW + ow) * g.C + c : -1) : dst;
            if (src >= 0) {
                const f32x4 xr = *reinterpret_cast<const f32x4*>(x_relu + src);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (!(xr[e] > 0.f)) r[e] = 0.f;
            }
        }
        *reinterpret_cast<f32x4*>(dx + dst) = r;
    }
}

// 2x2 windows that tile the image exactly (the usual case): one thread per 4 channels of a POOLED pixel reads the
// gradient, the arg-max and the ReLU mask once and writes the four pixels of its window; 32-bit index arithmetic (the
// per-output-pixel kernel above spends most of its time in 64-bit divisions: 77 us for 80 MB on the MNIST CNN).
__global__ __launch_bounds__(256) void maxpool_bwd4_2x2_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg,
                                                               const float* __restrict__ x_relu, float* __restrict__ dx, unsigned OW,
                                                               unsigned C, unsigned total, int mask_pooled) {
    const unsigned CV = C / 4;
    const long long rs = 2ll * OW * C;      // one row of the full-resolution image
    for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < total; p += gridDim.x * 256u) {
        const unsigned q = p / CV, c = (p - q * CV) * 4;      // pooled pixel, first channel
        const unsigned t = q / OW, ow = q - t * OW;           // t = b * OH + oh: full-resolution row 2t
        const long long o = (long long)q * C + c;
        const uchar4 a = *reinterpret_cast<const uchar4*>(arg + o);
        f32x4 d = *reinterpret_cast<const f32x4*>(dy + o);
        if (x_relu && mask_pooled) {
            const f32x4 xr = *reinterpret_cast<const f32x4*>(x_relu + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = (xr[e] > 0.f) ? d[e] : 0.f;
        }
        const long long base = (2ll * t * (2ll * OW) + 2ll * ow) * C + c;
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            const long long dst = base + (pos >> 1) * rs + (pos & 1) * (long long)C;
            f32x4 r;
            r[0] = (int)a.x == pos ? d[0] : 0.f;
            r[1] = (int)a.y == pos ? d[1] : 0.f;
            r[2] = (int)a.z == pos ? d[2] : 0.f;
            r[3] = (int)a.w == pos ? d[3] : 0.f;
            if (x_relu && !mask_pooled) {
                const f32x4 xr = *reinterpret_cast<const f32x4*>(x_relu + dst);
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = (xr[e] > 0.f) ? r[e] : 0.f;
            }
            *reinterpret_cast<f32x4*>(dx + dst) = r;
        }
    }
}

inline int grid_for(long long total) {
    long long blocks = (total + 255) / 256;
    const long long cap = 256 * 32;      // grid-stride beyond 32 blocks per CU
    return (int)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

bool geom_ok(const ConvGeom& g) {
    // every image access is bounds-checked in the kernels; the caller sizes cols as [B*OH*OW, KH*KW*C]
    return g.B > 0 && g.H > 0 && g.W > 0 && g.C > 0 && g.KH > 0 && g.KW > 0 && g.OH > 0 && g.OW > 0 && g.sh > 0 &&
           g.sw > 0 && g.pad_t >= 0 && g.pad_l >= 0;
}

}  // namespace

DTF_API int dtf_im2col(const float* x, float* cols, int B, int H, int W, int C, int KH, int KW, int OH, int OW, int pad_t,
                       int pad_l, int sh, int sw, hipStream_t stream) {
    const ConvGeom g = {B, H, W, C, KH, KW, OH, OW, pad_t, pad_l, sh, sw};
    if (!geom_ok(g)) return -3;
    const long long M = (long long)B * OH * OW, K = (long long)KH * KW * C;
    if (C % 4 == 0) {
        const long long total = M * (K / 4);
        hipLaunchKernelGGL(im2col_kernel<4>, dim3(grid_for(total)), dim3(256), 0, stream, x, cols, g, total);
    } else if (C == 1) {
        const long long total = M * KH;
        hipLaunchKernelGGL(im2col_c1_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, cols, g, total);
    } else {
        const long long total = M * K;
        hipLaunchKernelGGL(im2col_kernel<1>, dim3(grid_for(total)), dim3(256), 0, stream, x, cols, g, total);
    }
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_col2im(const float* dcols, float* dx, const float* yprev, int B, int H, int W, int C, int KH, int KW, int OH,
                       int OW, int pad_t, int pad_l, int sh, int sw, hipStream_t stream) {
    const ConvGeom g = {B, H, W, C, KH, KW, OH, OW, pad_t, pad_l, sh, sw};
    if (!geom_ok(g)) return -3;
    if (C % 4 == 0) {
        const long long total = (long long)B * H * W * (C / 4);
        hipLaunchKernelGGL(col2im_kernel<4>, dim3(grid_for(total)), dim3(256), 0, stream, dcols, dx, yprev, g, total);
    } else {
        const long long total = (long long)B * H * W * C;
        hipLaunchKernelGGL(col2im_kernel<1>, dim3(grid_for(total)), dim3(256), 0, stream, dcols, dx, yprev, g, total);
    }
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_maxpool_strided_fwd(const float* x, float* y, unsigned char* arg, int B, int H, int W, int C, int PH, int PW,
                                    int SH, int SW, int OH, int OW, int pad_t, int pad_l, hipStream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || PH <= 0 || PW <= 0 || SH <= 0 || SW <= 0 || OH <= 0 || OW <= 0 ||
        PH * PW > 255 || pad_t < 0 || pad_l < 0)
        return -3;
    const PoolGeomS g = {B, H, W, C, PH, PW, SH, SW, OH, OW, pad_t, pad_l};
    const long long total = (long long)B * OH * OW * C;
    hipLaunchKernelGGL(maxpool_strided_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, y, arg, g, total);
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_maxpool_strided_bwd(const float* dy, const unsigned char* arg, const float* x_relu, float* dx, int B, int H,
                                    int W, int C, int PH, int PW, int SH, int SW, int OH, int OW, int pad_t, int pad_l,
                                    hipStream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || PH <= 0 || PW <= 0 || SH <= 0 || SW <= 0 || OH <= 0 || OW <= 0 ||
        PH * PW > 255 || pad_t < 0 || pad_l < 0)
        return -3;
    const PoolGeomS g = {B, H, W, C, PH, PW, SH, SW, OH, OW, pad_t, pad_l};
    const long long total = (long long)B * H * W * C;
    hipLaunchKernelGGL(maxpool_strided_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dy, arg, x_relu, dx, g, total);
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_maxpool_fwd(const float* x, float* y, unsigned char* arg, int B, int H, int W, int C, int PH, int PW,
                            hipStream_t stream) {
    if (B <= 0 || H < PH || W < PW || C <= 0 || PH <= 0 || PW <= 0 || PH * PW > 255) return -3;
    const PoolGeom g = {B, H, W, C, PH, PW, H / PH, W / PW};
    const long long total = (long long)B * g.OH * g.OW * C;
    if (C % 4 == 0)
        hipLaunchKernelGGL(maxpool_fwd4_kernel, dim3(grid_for(total / 4)), dim3(256), 0, stream, x, y, arg, g, total / 4);
    else
        hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, y, arg, g, total);
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_maxpool_bwd(const float* dy, const unsigned char* arg, const float* x_relu, float* dx, int B, int H, int W,
                            int C, int PH, int PW, int mask_pooled, hipStream_t stream) {
    if (B <= 0 || H < PH || W < PW || C <= 0 || PH <= 0 || PW <= 0 || PH * PW > 255) return -3;
    if (mask_pooled && x_relu && C % 4 != 0) return -3;
    const PoolGeom g = {B, H, W, C, PH, PW, H / PH, W / PW};
    const long long total = (long long)B * H * W * C;
    if (C % 4 == 0 && PH == 2 && PW == 2 && H % 2 == 0 && W % 2 == 0 && total / 16 < 0xFFFF0000ll) {
        const long long pooled = total / 16;      // 4-channel vectors of pooled pixels
        hipLaunchKernelGGL(maxpool_bwd4_2x2_kernel, dim3(grid_for(pooled)), dim3(256), 0, stream, dy, arg, x_relu, dx,
                           (unsigned)(W / 2), (unsigned)C, (unsigned)pooled, mask_pooled);
    } else if (C % 4 == 0)
        hipLaunchKernelGGL(maxpool_bwd4_kernel, dim3(grid_for(total / 4)), dim3(256), 0, stream, dy, arg, x_relu, dx, g,
                           total / 4, mask_pooled);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dy, arg, x_relu, dx, g, total);
    DTF_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Helpers of the implicit-GEMM convolution path (stride 1): the GEMM kernels gather patch rows straight from a
// zero-padded copy of the image, so the only per-step extra traffic is that copy (1.3-1.7x the image) instead of
// the KH*KW-times larger patch matrix.
namespace {

// dst[b][h + pt][w + pl][c] = src[b][h][w][c], zero elsewhere; one thread per 4 channels of a padded pixel
__global__ __launch_bounds__(256) void pad_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H,
                                                       int W, int C, int pt, int pl, int Hp, int Wp, long long total) {
    const int CV = C / 4;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % CV) * 4;
        const long long pix = idx / CV;
        const int wp = (int)(pix % Wp);
        const int hp = (int)((pix / Wp) % Hp);
        const int b = (int)(pix / ((long long)Wp * Hp));
        const int h = hp - pt, w = wp - pl;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (h >= 0 && h < H && w >= 0 && w < W) v = *reinterpret_cast<const f32x4*>(src + (((long long)b * H + h) * W + w) * C + c);
        *reinterpret_cast<f32x4*>(dst + pix * C + c) = v;
    }
}

// the same for channel counts that are not multiples of 4 (a 1- or 3-channel input image): one thread per element
__global__ __launch_bounds__(256) void pad_nhwc_scalar_kernel(const float* __restrict__ src, float* __restrict__ dst, int B,
                                                              int H, int W, int C, int pt, int pl, int Hp, int Wp,
                                                              long long total) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const long long pix = idx / C;
        const int wp = (int)(pix % Wp);
        const int hp = (int)((pix / Wp) % Hp);
        const int b = (int)(pix / ((long long)Wp * Hp));
        const int h = hp - pt, w = wp - pl;
        dst[idx] = (h >= 0 && h < H && w >= 0 && w < W) ? src[(((long long)b * H + h) * W + w) * C + c] : 0.f;
    }
}

// rowtab[(b*OH + oh)*OW + ow] = ((b*Hp + oh*sy)*Wp + ow*sx) * C : origin of the patch of output pixel (b, oh, ow) of a
// convolution with strides (sy, sx)
__global__ __launch_bounds__(256) void conv_rowtab_kernel(int* __restrict__ rowtab, int B, int OH, int OW, int Hp, int Wp,
                                                          int C, int sy, int sx, long long total) {
    for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < total; m += (long long)gridDim.x * 256) {
        const int ow = (int)(m % OW);
        const int oh = (int)((m / OW) % OH);
        const int b = (int)(m / ((long long)OW * OH));
        rowtab[m] = (int)((((long long)b * Hp + oh * sy) * Wp + ow * sx) * C);
    }
}

// Tables for gathering patches from an UNPADDED image: rowtab[m] = ((b*H + oh*sy - pt)*W + (ow*sx - pl)) * C, the origin
// of the window of output pixel m = (b, oh, ow) - negative at the top / left border - and bit ky*KW + kx of masktab[m]
// set iff input pixel (oh*sy - pt + ky, ow*sx - pl + kx) exists.  Taps outside the image then read as zero inside the GEMM loader,
// and no zero-padded copy of the image is ever written.
// pool_order: rows are numbered window by window of a 2x2 pooling over the output, m = ((b*OH/2 + py)*OW/2 + px)*4 +
// 2*dy + dx for output pixel (2py + dy, 2px + dx), so that four consecutive GEMM rows are one pooling window.
__global__ __launch_bounds__(256) void conv_rowtab_masked_kernel(int* __restrict__ rowtab, unsigned* __restrict__ masktab, int B,
                                                                 int OH, int OW, int H, int W, int KH, int KW, int pt, int pl,
                                                                 int C, int pool_order, int sy, int sx, long long total) {
    for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < total; m += (long long)gridDim.x * 256) {
        int ow, oh, b;
        if (pool_order) {
            const int pos = (int)(m & 3);
            const long long win = m >> 2;
            const int PW2 = OW / 2, PH2 = OH / 2;
            ow = (int)(win % PW2) * 2 + (pos & 1);
            oh = (int)((win / PW2) % PH2) * 2 + (pos >> 1);
            b = (int)(win / ((long long)PW2 * PH2));
        } else {
            ow = (int)(m % OW);
            oh = (int)((m / OW) % OH);
            b = (int)(m / ((long long)OW * OH));
        }
        rowtab[m] = (int)((((long long)b * H + (oh * sy - pt)) * W + (ow * sx - pl)) * C);
        unsigned mk = 0;
        for (int ky = 0; ky < KH; ++ky)
            for (int kx = 0; kx < KW; ++kx) {
                const int h = oh * sy - pt + ky, w = ow * sx - pl + kx;
                if (h >= 0 && h < H && w >= 0 && w < W) mk |= 1u << (ky * KW + kx);
            }
        masktab[m] = mk;
    }
}

// wflip[((KH-1-kh)*KW + (KW-1-kw))*N + n][c] = w[kh][kw][c][n]
__global__ __launch_bounds__(256) void conv_flip_kernel(const float* __restrict__ w, float* __restrict__ wflip, int KH, int KW,
                                                        int C, int N, int total) {
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int n = idx % N;
        const int c = (idx / N) % C;
        const int kw = (idx / (N * C)) % KW;
        const int kh = idx / (N * C * KW);
        wflip[(((KH - 1 - kh) * KW + (KW - 1 - kw)) * N + n) * C + c] = w[idx];
    }
}

}  // namespace

DTF_API int dtf_conv_pad(const float* src, float* dst, int B, int H, int W, int C, int pt, int pl, int Hp, int Wp,
                         hipStream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || pt < 0 || pl < 0 || Hp < H + pt || Wp < W + pl) return -3;
    if ((long long)B * Hp * Wp * C >= (1ll << 31)) return -8;        // offsets are 32-bit element indices
    if (C % 4) {
        const long long n = (long long)B * Hp * Wp * C;
        hipLaunchKernelGGL(pad_nhwc_scalar_kernel, dim3(grid_for(n)), dim3(256), 0, stream, src, dst, B, H, W, C, pt, pl, Hp, Wp, n);
        DTF_CHECK_LAUNCH();
        return 0;
    }
    const long long total = (long long)B * Hp * Wp * (C / 4);
    hipLaunchKernelGGL(pad_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, stream, src, dst, B, H, W, C, pt, pl, Hp, Wp, total);
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_conv_rowtab_masked(int* rowtab, unsigned* masktab, int B, int OH, int OW, int H, int W, int KH, int KW, int pt,
                                   int pl, int C, int pool_order, int sy, int sx, hipStream_t stream) {
    if (B <= 0 || OH <= 0 || OW <= 0 || H <= 0 || W <= 0 || C <= 0 || KH <= 0 || KW <= 0 || KH * KW > 32) return -3;
    if (sy < 1 || sx < 1) return -3;
    if (pool_order && (OH % 2 != 0 || OW % 2 != 0)) return -3;
    if ((long long)B * H * W * C >= (1ll << 30)) return -8;      // byte offsets of the gather are 32-bit
    const long long total = (long long)B * OH * OW;
    hipLaunchKernelGGL(conv_rowtab_masked_kernel, dim3(grid_for(total)), dim3(256), 0, stream, rowtab, masktab, B, OH, OW, H, W,
                       KH, KW, pt, pl, C, pool_order, sy, sx, total);
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_conv_rowtab(int* rowtab, int B, int OH, int OW, int Hp, int Wp, int C, int sy, int sx, hipStream_t stream) {
    if (B <= 0 || OH <= 0 || OW <= 0 || sy < 1 || sx < 1 || Hp < (OH - 1) * sy + 1 || Wp < (OW - 1) * sx + 1 || C <= 0) return -3;
    if ((long long)B * Hp * Wp * C >= (1ll << 31)) return -8;
    const long long total = (long long)B * OH * OW;
    hipLaunchKernelGGL(conv_rowtab_kernel, dim3(grid_for(total)), dim3(256), 0, stream, rowtab, B, OH, OW, Hp, Wp, C, sy, sx, total);
    DTF_CHECK_LAUNCH();
    return 0;
}

DTF_API int dtf_conv_flip_weights(const float* w, float* wflip, int KH, int KW, int C, int N, hipStream_t stream) {
    if (KH <= 0 || KW <= 0 || C <= 0 || N <= 0) return -3;
    const int total = KH * KW * C * N;
    hipLaunchKernelGGL(conv_flip_kernel, dim3(grid_for(total)), dim3(256), 0, stream, w, wflip, KH, KW, C, N, total);
    DTF_CHECK_LAUNCH();
    return 0;
}
