// First-layer convolution fused with its 2x2 max pooling (NHWC, stride 1, few input channels: KH*KW*Cin < 32).
//
// A network's first convolution (e.g. 5x5x1 -> 32 on 28x28 images) has a tiny K and a huge, short-lived output: the
// unfused chain writes a patch matrix, the full-resolution activation, the pooled activation, and on the way back the
// full-resolution gradient - hundreds of MB per step for a few MFLOP per pixel.  Here one wave owns 8 pooled pixels
// (= 32 convolution pixels) x 32 output channels as ONE 32x32 MFMA tile:
//
//   forward   rows = conv pixels ordered so that the 4 members of a pooling window are rows 4q .. 4q+3, which the
//             32x32 C/D layout puts into 4 consecutive registers of one lane: bias, ReLU, max and argmax are
//             register-only, and only the pooled tensor and a 2-bit argmax (stored as a byte) leave the chip.
//             A operands are gathered from the image one value per lane per k step (the image stays in L1/L2).
//   wgrad     D[k][c] += patch[k][pixel] * dz[pixel][c], two pixels per v_mfma_f32_32x32x2_f32; dz is rebuilt in
//             registers from the pooled gradient, the argmax and the ReLU mask (pooled output > 0), so the
//             full-resolution gradient never exists.  Row K of the A operand is all ones: its result row is the
//             bias gradient.  Every block writes one partial [K+1][F] slab; a small kernel adds the slabs.
//
// Both kernels exist in two forms.  The *staged* form (the default whenever a few images fit in LDS) copies a group of
// whole images into LDS once - zero padding included, so 'same' convolutions need no padded copy of the input and no
// bounds test per tap - and gathers every A operand from there: a gather of 64 scattered floats costs a couple of LDS
// cycles instead of a trip through the vector L1 per lane, and the position of every pooled pixel comes from a table
// built once per block instead of divisions per tile.  The *direct* form gathers from global memory and takes any size.
//
// This is what TF runs as Conv2D + BiasAdd + Relu + MaxPool and their gradients for a first layer; the reference
// network itself is dense only (/root/reference/example.py:149-154), the CNN is this framework's MNIST-shaped model.
#include "common.h"

namespace {

struct ConvPoolGeom {
    int B, H, W, C;         // input image
    int KH, KW, pad_t, pad_l;
    int PH, PW;             // pooled output (conv output is 2*PH x 2*PW)
    int F;                  // output channels, multiple of 32
    int K;                  // KH*KW*C, < 32
    unsigned taps[32];      // tap k packed on the host: (dy + 64) | (dx + 64) << 8 | channel << 16 | valid << 31
};

// tap k of the patch -> (row offset, column offset, channel); k >= K is marked invalid
struct Tap { int dy, dx, ci; bool ok; };
__device__ __forceinline__ Tap unpack_tap(unsigned u) {
    Tap t;
    t.dy = (int)(u & 0xFF) - 64;
    t.dx = (int)((u >> 8) & 0xFF) - 64;
    t.ci = (int)((u >> 16) & 0x7FFF);
    t.ok = (u >> 31) != 0;
    return t;
}
__host__ __device__ __forceinline__ Tap tap_of(int k, const ConvPoolGeom& g) {
    Tap t;
    t.ok = k < g.K;
    const int kk = t.ok ? k : 0;
    t.ci = kk % g.C;
    const int w = kk / g.C;
    t.dx = w % g.KW - g.pad_l;
    t.dy = w / g.KW - g.pad_t;
    return t;
}

// CHECK = false: the caller guarantees every tap of every window lies inside the image (VALID convolution, e.g. of
// an image that was zero-padded beforehand), so a tap costs one address add instead of bounds tests and selects.
template <bool CHECK>
__global__ __launch_bounds__(256) void conv_pool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ ypool,
                                                            unsigned char* __restrict__ arg, const ConvPoolGeom g, int act,
                                                            int n_pooled) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int c = blockIdx.y * 32 + r;                      // this lane's output channel (C/D column and B column)
    const int ks = (g.K + 1) / 2;                           // MFMA steps of two taps
    // B operand: lane (r, h) holds w[k = 2s + h][c]; tap of step s: offset inside the image relative to the pixel
    float bw[16];
    Tap tp[16];
    int toff[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        tp[s] = unpack_tap(h ? g.taps[2 * s + 1] : g.taps[2 * s]);
        toff[s] = tp[s].ok ? (tp[s].dy * g.W + tp[s].dx) * g.C + tp[s].ci : 0;
        bw[s] = tp[s].ok ? w[(long long)(2 * s + h) * g.F + c] : 0.f;
    }
    const float bv = bias ? bias[c] : 0.f;
    const int tiles = (n_pooled + 7) / 8;
    const int HW = g.PH * g.PW;
    for (int t = blockIdx.x * 4 + wave; t < tiles; t += gridDim.x * 4) {
        // A operand: lane (r, h) holds patch[row r][k = 2s + h]; row r = pooled pixel r >> 2, window member r & 3.
        // The tile's first pooled pixel is decoded once (wave-uniform), a lane steps at most 7 pixels on from it.
        int b = (t * 8) / HW;
        const int rem = t * 8 - b * HW;
        int py = rem / g.PW, px = rem - py * g.PW + (r >> 2);
        while (px >= g.PW) {
            px -= g.PW;
            if (++py == g.PH) { py = 0; ++b; }
        }
        const bool pv = t * 8 + (r >> 2) < n_pooled;
        if (!pv) { b = 0; py = 0; px = 0; }
        const int oy = 2 * py + ((r >> 1) & 1), ox = 2 * px + (r & 1);
        const float* pix = x + ((long long)b * g.H * g.W + oy * g.W + ox) * g.C;
        float a[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (CHECK) {
                const int iy = oy + tp[s].dy, ix = ox + tp[s].dx;
                const bool in = pv && tp[s].ok && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W && s < ks;
                const float v = pix[in ? toff[s] : -((oy * g.W + ox) * g.C)];     // unconditional load (image start if outside)
                a[s] = in ? v : 0.f;
            } else {
                a[s] = pix[toff[s]];                        // taps past K multiply a zero filter row
            }
        }
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s)
            if (s < ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bw[s], acc, 0, 0, 0);
        // rows of register i: (i & 3) + 8 * (i >> 2) + 4 * h  ->  pooled pixel 2 * (i >> 2) + h, member i & 3
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float m = acc[4 * q] + bv;
            int am = 0;
#pragma unroll
            for (int e = 1; e < 4; ++e) {
                const float v = acc[4 * q + e] + bv;
                if (v > m) { m = v; am = e; }
            }
            if (act == ACT_RELU) m = fmaxf(m, 0.f);
            const int Pq = t * 8 + 2 * q + h;
            if (Pq < n_pooled) {
                ypool[(long long)Pq * g.F + c] = m;
                arg[(long long)Pq * g.F + c] = (unsigned char)am;
            }
        }
    }
}

// one partial [K + 1][F] slab per block: rows 0 .. K-1 the filter gradient, row K the bias gradient
template <bool CHECK>
__global__ __launch_bounds__(256) void conv_pool_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dpool,
                                                              const float* __restrict__ ypool,
                                                              const unsigned char* __restrict__ arg, float* __restrict__ slabs,
                                                              long long slab_stride, const ConvPoolGeom g, int act, int n_pooled) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int c = blockIdx.y * 32 + r;                      // B column: output channel
    const Tap tp = tap_of(r, g);                            // A row: tap r (row K is the ones row)
    const bool ones = r == g.K;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int tiles = (n_pooled + 7) / 8;
    const int HW = g.PH * g.PW;
    for (int t = blockIdx.x * 4 + wave; t < tiles; t += gridDim.x * 4) {
        // MFMA step 2q + e covers the conv pixels of pooled pixel q, window row e; lane half h is the window column
        float dzq[8];
        int argq[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int P = t * 8 + q;
            const bool pv = P < n_pooled;                    // loads are unconditional (clamped) so all 24 fly together
            const long long o = (long long)(pv ? P : n_pooled - 1) * g.F + c;
            float d = dpool[o];
            const float yv = ypool[o];
            int am = arg[o];
            if (act == ACT_RELU && !(yv > 0.f)) d = 0.f;
            if (!pv) { d = 0.f; am = -1; }
            dzq[q] = d;
            argq[q] = am;
        }
        // image position of the tile's first pooled pixel, then one step per pooled pixel (all wave-uniform)
        int b = (t * 8) / HW;
        int rem = t * 8 - b * HW;
        int py = rem / g.PW, px = rem - py * g.PW;
        float a[16];                                         // all gathers of the tile in flight before the first MFMA
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool pv = t * 8 + q < n_pooled;
            const float* img = x + (long long)(pv ? b : 0) * g.H * g.W * g.C;
#pragma unroll
            for (int e = 0; e < 2; ++e) {                    // window row e: conv pixels (2py + e, 2px + h)
                const int iy = 2 * py + e + tp.dy, ix = 2 * px + h + tp.dx;
                const bool in = CHECK ? (pv && tp.ok && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) : (pv && tp.ok);
                const float v = img[in ? (iy * g.W + ix) * g.C + tp.ci : 0];      // unconditional load, then select
                a[2 * q + e] = ones ? 1.f : (in ? v : 0.f);
            }
            if (++px == g.PW) {
                px = 0;
                if (++py == g.PH) { py = 0; ++b; }
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float bz = (argq[s >> 1] == 2 * (s & 1) + h) ? dzq[s >> 1] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bz, acc, 0, 0, 0);
        }
    }
    // the block's four waves meet in LDS; one slab per block
    __shared__ float red[4][16][64];
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
    float* slab = slabs + (long long)blockIdx.x * slab_stride;
    for (int idx = threadIdx.x; idx < 1024; idx += 256) {
        const int i = idx >> 6, l = idx & 63;
        const float v = (red[0][i][l] + red[1][i][l]) + (red[2][i][l] + red[3][i][l]);
        const int k = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
        if (k <= g.K) slab[(long long)k * g.F + blockIdx.y * 32 + (l & 31)] = v;
    }
}

// out[i] = sum over slabs; 1024 threads = 16 outputs x 64 slab groups
__global__ __launch_bounds__(1024) void conv_pool_reduce_kernel(const float* __restrict__ slabs, long long slab_stride, int n_slabs,
                                                                float* __restrict__ dW, int n_main, float* __restrict__ db,
                                                                int n_total) {
    __shared__ float part[64][16];
    const int e = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + e;
    float s = 0.f;
    if (i < n_total)
        for (int k = grp; k < n_slabs; k += 64) s += slabs[(long long)k * slab_stride + i];
    part[grp][e] = s;
    __syncthreads();
    if (grp == 0 && i < n_total) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 64; ++k) t += part[k][e];
        if (i < n_main) dW[i] = t;
        else if (db) db[i - n_main] = t;
    }
}


// ---- staged form ---------------------------------------------------------------------------------------------------
// LDS image of a group of G images: [G][HP][WS] floats, HP = rows a pooled output needs (2*PH + KH - 1), WS = row stride
// in floats (>= (2*PW + KW - 1) * C, a multiple of 4 whose quarter is odd: rows two apart do not share banks).  Pixels
// outside the source image are zero.  `ptab[P]`: LDS offset of the window origin (tap (0,0) of conv pixel (2py, 2px)) of
// pooled pixel P of the group.
struct StageGeom {
    int G, HP, WS, img_floats;     // img_floats = HP * WS
    int tiles;                     // ceil(G * PH * PW / 8)
};

__device__ __forceinline__ void stage_group(const float* __restrict__ x, float* simg, int* ptab, const ConvPoolGeom& g,
                                            const StageGeom& sg, int first_image, int n_images) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int total = sg.G * sg.img_floats;
    for (int i = tid; i < total; i += nt) simg[i] = 0.f;
    const int HW = g.PH * g.PW;
    for (int P = tid; P < sg.G * HW; P += nt) {
        const int img = P / HW, rem = P - img * HW;
        const int py = rem / g.PW, px = rem - py * g.PW;
        ptab[P] = (img * sg.HP + 2 * py) * sg.WS + 2 * px * g.C;
    }
    __syncthreads();
    // the group's images are contiguous in memory: 16-byte loads, scattered to their padded places
    const int row = g.W * g.C, per_img = g.H * row;
    const long long base = (long long)first_image * per_img;
    const int n = n_images * per_img;
    for (int i = tid * 4; i < n; i += nt * 4) {
        float v[4];
        if (i + 3 < n && ((base + i) & 3) == 0) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(x + base + i);
            v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = i + j < n ? x[base + i + j] : 0.f;
        }
        int img = i / per_img, rem = i - img * per_img;
        int y = rem / row, xc = rem - y * row;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (i + j < n) simg[(img * sg.HP + y + g.pad_t) * sg.WS + xc + g.pad_l * g.C] = v[j];
            if (++xc == row) {
                xc = 0;
                if (++y == g.H) { y = 0; ++img; }
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void conv_pool_fwd_staged_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ bias, float* __restrict__ ypool,
                                                                   unsigned char* __restrict__ arg, const ConvPoolGeom g,
                                                                   const StageGeom sg, int act) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* simg = lds;
    int* ptab = reinterpret_cast<int*>(lds + sg.G * sg.img_floats);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int c = blockIdx.y * 32 + r;
    const int ks = (g.K + 1) / 2;
    float bw[16];
    int toff[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const Tap tp = unpack_tap(h ? g.taps[2 * s + 1] : g.taps[2 * s]);
        // tap offsets inside the padded LDS image are never negative: (dy + pad_t, dx + pad_l)
        toff[s] = tp.ok ? (tp.dy + g.pad_t) * sg.WS + (tp.dx + g.pad_l) * g.C + tp.ci : 0;
        bw[s] = tp.ok ? w[(long long)(2 * s + h) * g.F + c] : 0.f;
    }
    const float bv = bias ? bias[c] : 0.f;
    const int HW = g.PH * g.PW;
    const int member = ((r >> 1) & 1) * sg.WS + (r & 1) * g.C;        // this row's conv pixel inside its pooling window
    const int groups = (g.B + sg.G - 1) / sg.G;
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int first = grp * sg.G;
        const int n_img = min(sg.G, g.B - first);
        if (grp != (int)blockIdx.x) __syncthreads();                 // every wave is done with the previous group's image
        stage_group(x, simg, ptab, g, sg, first, n_img);
        const int n_pooled = n_img * HW;
        const long long out0 = (long long)first * HW;
        for (int t = wave; t * 8 < n_pooled; t += 4) {
            const int P = t * 8 + (r >> 2);
            const int base = (P < n_pooled ? ptab[P] : 0) + member;
            float a[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) a[s] = simg[base + toff[s]];   // taps past K multiply a zero filter row
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s)
                if (s < ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bw[s], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float m = acc[4 * q] + bv;
                int am = 0;
#pragma unroll
                for (int e = 1; e < 4; ++e) {
                    const float v = acc[4 * q + e] + bv;
                    if (v > m) { m = v; am = e; }
                }
                if (act == ACT_RELU) m = fmaxf(m, 0.f);
                const int Pq = t * 8 + 2 * q + h;
                if (Pq < n_pooled) {
                    ypool[(out0 + Pq) * g.F + c] = m;
                    arg[(out0 + Pq) * g.F + c] = (unsigned char)am;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void conv_pool_wgrad_staged_kernel(const float* __restrict__ x, const float* __restrict__ dpool,
                                                                     const float* __restrict__ ypool,
                                                                     const unsigned char* __restrict__ arg,
                                                                     float* __restrict__ slabs, long long slab_stride,
                                                                     const ConvPoolGeom g, const StageGeom sg, int act) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* simg = lds;
    int* ptab = reinterpret_cast<int*>(lds + sg.G * sg.img_floats);
    float* red = lds + sg.G * sg.img_floats + ((sg.G * g.PH * g.PW + 3) & ~3);     // [4][16][64]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int c = blockIdx.y * 32 + r;
    const Tap tp = tap_of(r, g);
    const bool ones = r == g.K;
    // this lane's part of every gather address: tap r, window column h
    const int lofs = tp.ok ? (tp.dy + g.pad_t) * sg.WS + (h + tp.dx + g.pad_l) * g.C + tp.ci : 0;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int HW = g.PH * g.PW;
    const int groups = (g.B + sg.G - 1) / sg.G;
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int first = grp * sg.G;
        const int n_img = min(sg.G, g.B - first);
        if (grp != (int)blockIdx.x) __syncthreads();
        stage_group(x, simg, ptab, g, sg, first, n_img);
        const int n_pooled = n_img * HW;
        const long long out0 = (long long)first * HW;
        for (int t = wave; t * 8 < n_pooled; t += 4) {
            float dzq[8];
            int argq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int P = t * 8 + q;
                const bool pv = P < n_pooled;
                const long long o = (out0 + (pv ? P : n_pooled - 1)) * g.F + c;
                float d = dpool[o];
                const float yv = ypool[o];
                int am = arg[o];
                if (act == ACT_RELU && !(yv > 0.f)) d = 0.f;
                if (!pv) { d = 0.f; am = -1; }
                dzq[q] = d;
                argq[q] = am;
            }
            float a[16];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int P = t * 8 + q;
                const int origin = ptab[P < n_pooled ? P : 0];         // wave-uniform address: one broadcast read
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float v = simg[origin + e * sg.WS + lofs];
                    a[2 * q + e] = ones ? 1.f : (tp.ok ? v : 0.f);
                }
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float bz = (argq[s >> 1] == 2 * (s & 1) + h) ? dzq[s >> 1] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bz, acc, 0, 0, 0);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[i];
    __syncthreads();
    float* slab = slabs + (long long)blockIdx.x * slab_stride;
    for (int idx = threadIdx.x; idx < 1024; idx += 256) {
        const int i = idx >> 6, l = idx & 63;
        const float v = (red[(0 * 16 + i) * 64 + l] + red[(1 * 16 + i) * 64 + l]) + (red[(2 * 16 + i) * 64 + l] + red[(3 * 16 + i) * 64 + l]);
        const int k = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
        if (k <= g.K) slab[(long long)k * g.F + blockIdx.y * 32 + (l & 31)] = v;
    }
}


// Filter gradient on the vector ALUs.  Of the four conv pixels of a pooling window only the arg-max member has a
// gradient, and which one differs per channel: as a matrix product three quarters of dz are zeros.  Here a lane owns one
// (pooled pixel, channel): it reads that channel's arg-max member, and adds d * patch[k] for the K taps of that ONE
// pixel's patch, read from the staged image (a wave = 2 pooled pixels x 32 channels; at most four distinct addresses
// per pixel and tap, in different banks).  A quarter of the multiply-adds of the MFMA form and no operand shuffling;
// the next item's gradient, mask and arg-max are requested before the current one is accumulated.  KMAX: the number of
// taps rounded up to a multiple of four (the tap loop is unrolled without tests; surplus taps re-read the pixel itself).
template <int KMAX>
__global__ __launch_bounds__(256) void conv_pool_wgrad_valu_kernel(const float* __restrict__ x, const float* __restrict__ dpool,
                                                                   const float* __restrict__ ypool,
                                                                   const unsigned char* __restrict__ arg,
                                                                   float* __restrict__ slabs, long long slab_stride,
                                                                   const ConvPoolGeom g, const StageGeom sg, int act) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* simg = lds;
    int* ptab = reinterpret_cast<int*>(lds + sg.G * sg.img_floats);
    float* red = lds + sg.G * sg.img_floats + ((sg.G * g.PH * g.PW + 3) & ~3);     // [4 waves][KMAX + 1][32]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int c = blockIdx.y * 32 + r;
    int toff[KMAX];                                         // wave-uniform: LDS offset of tap k from a conv pixel's origin
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const Tap tp = unpack_tap(g.taps[k]);
        toff[k] = tp.ok ? (tp.dy + g.pad_t) * sg.WS + (tp.dx + g.pad_l) * g.C + tp.ci : 0;
    }
    float acc[KMAX], accb = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
    const int HW = g.PH * g.PW;
    const int groups = (g.B + sg.G - 1) / sg.G;
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int first = grp * sg.G;
        const int n_img = min(sg.G, g.B - first);
        if (grp != (int)blockIdx.x) __syncthreads();
        stage_group(x, simg, ptab, g, sg, first, n_img);
        const int n_pooled = n_img * HW;
        const long long out0 = (long long)first * HW;
        // item P = 2 * (wave + 4 * it) + h
        int P = 2 * wave + h;
        float d = 0.f, yv = 1.f;
        int am = 0;
        if (P < n_pooled) {
            const long long o = (out0 + P) * g.F + c;
            d = dpool[o]; yv = ypool[o]; am = arg[o];
        }
        while (P - h < n_pooled) {                          // (wave-uniform: both halves run the same trip count)
            const int Pn = P + 8;
            float dn = 0.f, yn = 1.f;
            int an = 0;
            if (Pn < n_pooled) {
                const long long o = (out0 + Pn) * g.F + c;
                dn = dpool[o]; yn = ypool[o]; an = arg[o];
            }
            if (P < n_pooled) {
                if (act == ACT_RELU && !(yv > 0.f)) d = 0.f;
                const int base = ptab[P] + (am >> 1) * sg.WS + (am & 1) * g.C;
                accb += d;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) acc[k] = fmaf(simg[base + toff[k]], d, acc[k]);   // taps >= K: unused sums
            }
            P = Pn; d = dn; yv = yn; am = an;
        }
    }
    // the two pooled pixels of a wave, then the block's four waves; one [K + 1][F] slab per block
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] += __shfl_down(acc[k], 32, 64);
    accb += __shfl_down(accb, 32, 64);
    __syncthreads();
    if (h == 0) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) red[(wave * (KMAX + 1) + k) * 32 + r] = acc[k];
        red[(wave * (KMAX + 1) + KMAX) * 32 + r] = accb;
    }
    __syncthreads();
    float* slab = slabs + (long long)blockIdx.x * slab_stride;
    for (int idx = threadIdx.x; idx < (g.K + 1) * 32; idx += 256) {
        const int k = idx >> 5, l = idx & 31;
        const int kk = k < g.K ? k : KMAX;                   // row K of the slab is the bias gradient
        const float v = (red[(0 * (KMAX + 1) + kk) * 32 + l] + red[(1 * (KMAX + 1) + kk) * 32 + l]) +
                        (red[(2 * (KMAX + 1) + kk) * 32 + l] + red[(3 * (KMAX + 1) + kk) * 32 + l]);
        slab[(long long)k * g.F + blockIdx.y * 32 + l] = v;
    }
}

// Staging plan for a geometry: how many images per group (a whole number of 8-pixel tiles where possible) and whether
// the group fits; false = use the direct kernels.
bool stage_plan(const ConvPoolGeom& g, StageGeom& sg, size_t* lds_bytes, bool wgrad) {
    const int HP = 2 * g.PH + g.KH - 1;
    int WS = ((2 * g.PW + g.KW - 1) * g.C + 3) & ~3;
    if (((WS >> 2) & 1) == 0) WS += 4;
    // the source image must fit inside the padded frame
    if (g.H + g.pad_t > HP || (g.W + g.pad_l) * g.C > WS) return false;
    const int HW = g.PH * g.PW;
    int G = 1;
    while (G < 8 && (G * HW) % 8 != 0) ++G;              // G * HW a multiple of 8: no partly filled tile per group
    if ((G * HW) % 8 != 0) G = 1;
    while (G * 2 <= 8 && G * 2 * HW <= 512 && G * 2 <= g.B) G *= 2;     // ~50 tiles per block of four waves
    sg.G = G; sg.HP = HP; sg.WS = WS; sg.img_floats = HP * WS;
    sg.tiles = (G * HW + 7) / 8;
    size_t floats = (size_t)G * sg.img_floats + (((size_t)G * HW + 3) & ~(size_t)3) + (wgrad ? 4 * 16 * 64 : 0);
    *lds_bytes = floats * 4;
    return *lds_bytes <= 64 * 1024;
}

void fill_taps(ConvPoolGeom& g) {
    for (int k = 0; k < 32; ++k) {
        const Tap t = tap_of(k, g);
        g.taps[k] = (unsigned)(t.dy + 64) | ((unsigned)(t.dx + 64) << 8) | ((unsigned)t.ci << 16) | (t.ok ? 0x80000000u : 0u);
    }
}

// every tap of every pooling window inside the image?
bool taps_inside(const ConvPoolGeom& g) {
    return g.pad_t == 0 && g.pad_l == 0 && 2 * g.PH + g.KH - 1 <= g.H && 2 * g.PW + g.KW - 1 <= g.W;
}

bool geom_ok(const ConvPoolGeom& g) {
    return g.B > 0 && g.H > 0 && g.W > 0 && g.C > 0 && g.K == g.KH * g.KW * g.C && g.K < 32 && g.F > 0 && g.F % 32 == 0 &&
           g.KH <= 32 && g.KW <= 32 && g.PH > 0 && g.PW > 0 && (long long)g.B * g.PH * g.PW < (1ll << 31) - 8 &&
           (long long)g.B * g.H * g.W * g.C < (1ll << 31);
}

constexpr int WGRAD_BLOCKS = 1024;    // partial slabs; 4 waves per SIMD keep the gathers of several tiles in flight
int g_wgrad_form = 0;                 // staged filter gradient: 0 = vector ALUs (one arg-max pixel per item), 1 = MFMA tiles
bool g_use_staged = true;             // DTF_CONV_POOL_STAGED=0 (read by the Python side) switches the staged form off

}  // namespace

// Switches the staged (LDS) form on or off (on by default); the direct kernels then take every shape.
DTF_API int dtf_conv_pool_set_staged(int on) {
    g_use_staged = on != 0;
    g_wgrad_form = on == 2 ? 1 : 0;      // 2: staged, with the filter gradient on MFMA tiles (measurement aid)
    return 0;
}

// 1 when the staged kernels take this geometry: they read the UNPADDED image and pad while staging
DTF_API int dtf_conv_pool_staged_fits(int B, int H, int W, int C, int KH, int KW, int pad_t, int pad_l, int PH, int PW, int F) {
    ConvPoolGeom g = {B, H, W, C, KH, KW, pad_t, pad_l, PH, PW, F, KH * KW * C, {}};
    StageGeom sg;
    size_t lds = 0;
    return (g_use_staged && geom_ok(g) && stage_plan(g, sg, &lds, true)) ? 1 : 0;
}

// 1 when the fused first-layer kernels take this shape
DTF_API int dtf_conv_pool_supported(int C, int KH, int KW, int F) {
    return (C > 0 && KH > 0 && KW > 0 && KH * KW * C < 32 && F > 0 && F % 32 == 0) ? 1 : 0;
}

// floats of scratch dtf_conv_pool_wgrad needs
DTF_API long long dtf_conv_pool_workspace_floats(int C, int KH, int KW, int F) {
    const long long stride = (((long long)(KH * KW * C + 1) * F + 3) / 4) * 4;
    return stride * WGRAD_BLOCKS;
}

// ypool[B, PH, PW, F] = maxpool2x2(act(conv(x[B, H, W, C], w[KH*KW*C, F]) + bias)); arg = window member (dy*2+dx) of the max
DTF_API int dtf_conv_pool_fwd(const float* x, const float* w, const float* bias, float* ypool, unsigned char* arg, int B, int H,
                              int W, int C, int KH, int KW, int pad_t, int pad_l, int PH, int PW, int F, int act,
                              hipStream_t stream) {
    ConvPoolGeom g = {B, H, W, C, KH, KW, pad_t, pad_l, PH, PW, F, KH * KW * C, {}};
    if (!geom_ok(g)) return -3;
    fill_taps(g);
    const int n_pooled = B * PH * PW;
    StageGeom sg;
    size_t lds = 0;
    if (g_use_staged && stage_plan(g, sg, &lds, false)) {
        int groups = (B + sg.G - 1) / sg.G;
        if (groups > 4096) groups = 4096;
        hipLaunchKernelGGL(conv_pool_fwd_staged_kernel, dim3(groups, F / 32), dim3(256), lds, stream, x, w, bias, ypool, arg,
                           g, sg, act);
        DTF_CHECK_LAUNCH();
        return 0;
    }
    const int tiles = (n_pooled + 7) / 8;
    int blocks = (tiles + 3) / 4;
    if (blocks > 2048) blocks = 2048;          // a wave amortises its filter registers over several tiles
    if (taps_inside(g))
        hipLaunchKernelGGL(conv_pool_fwd_kernel<false>, dim3(blocks, F / 32), dim3(256), 0, stream, x, w, bias, ypool, arg, g, act, n_pooled);
    else
        hipLaunchKernelGGL(conv_pool_fwd_kernel<true>, dim3(blocks, F / 32), dim3(256), 0, stream, x, w, bias, ypool, arg, g, act, n_pooled);
    DTF_CHECK_LAUNCH();
    return 0;
}

// dW[KH*KW*C, F], db[F] (may be null) from the pooled gradient dpool[B, PH, PW, F]
DTF_API int dtf_conv_pool_wgrad(const float* x, const float* dpool, const float* ypool, const unsigned char* arg, float* dW,
                                float* db, int B, int H, int W, int C, int KH, int KW, int pad_t, int pad_l, int PH, int PW,
                                int F, int act, float* workspace, hipStream_t stream) {
    ConvPoolGeom g = {B, H, W, C, KH, KW, pad_t, pad_l, PH, PW, F, KH * KW * C, {}};
    if (!geom_ok(g) || !workspace) return -3;
    fill_taps(g);
    const int n_pooled = B * PH * PW;
    const long long stride = (((long long)(g.K + 1) * F + 3) / 4) * 4;
    StageGeom sg;
    size_t lds = 0;
    int n_slabs = WGRAD_BLOCKS;
    if (g_use_staged && stage_plan(g, sg, &lds, true)) {
        const int groups = (B + sg.G - 1) / sg.G;
        n_slabs = groups < WGRAD_BLOCKS ? groups : WGRAD_BLOCKS;
        if (g_wgrad_form == 0) {
#define DTF_WGRAD_VALU(KU)                                                                                                   \
    hipLaunchKernelGGL(conv_pool_wgrad_valu_kernel<KU>, dim3(n_slabs, F / 32), dim3(256), lds, stream, x, dpool, ypool, arg, \
                       workspace, stride, g, sg, act)
            switch ((g.K + 3) / 4) {
                case 1: DTF_WGRAD_VALU(4); break;
                case 2: DTF_WGRAD_VALU(8); break;
                case 3: DTF_WGRAD_VALU(12); break;
                case 4: DTF_WGRAD_VALU(16); break;
                case 5: DTF_WGRAD_VALU(20); break;
                case 6: DTF_WGRAD_VALU(24); break;
                case 7: DTF_WGRAD_VALU(28); break;
                default: DTF_WGRAD_VALU(31); break;
            }
#undef DTF_WGRAD_VALU
        } else
            hipLaunchKernelGGL(conv_pool_wgrad_staged_kernel, dim3(n_slabs, F / 32), dim3(256), lds, stream, x, dpool, ypool,
                               arg, workspace, stride, g, sg, act);
    } else if (taps_inside(g))
        hipLaunchKernelGGL(conv_pool_wgrad_kernel<false>, dim3(WGRAD_BLOCKS, F / 32), dim3(256), 0, stream, x, dpool, ypool, arg,
                           workspace, stride, g, act, n_pooled);
    else
        hipLaunchKernelGGL(conv_pool_wgrad_kernel<true>, dim3(WGRAD_BLOCKS, F / 32), dim3(256), 0, stream, x, dpool, ypool, arg,
                           workspace, stride, g, act, n_pooled);
    DTF_CHECK_LAUNCH();
    const int n_main = g.K * F, n_total = (g.K + 1) * F;
    hipLaunchKernelGGL(conv_pool_reduce_kernel, dim3((n_total + 15) / 16), dim3(1024), 0, stream, workspace, stride,
                       n_slabs, dW, n_main, db, n_total);
    DTF_CHECK_LAUNCH();
    return 0;
}
