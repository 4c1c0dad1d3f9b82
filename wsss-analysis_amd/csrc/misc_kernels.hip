// misc_kernels.hip -- the small HBM-bound kernels around the conv stack.
#include "common.h"

namespace {

// float32 NCHW [N][3][H][W] -> bf16 NHWC4 [N][H][W][4] (4th channel zero), optional lo plane.
// Replaces img.cuda() + the layout torch/cuDNN picks internally (make_cam.py:48).
__global__ void nchw_to_nhwc4_kernel(const float *__restrict__ x, int N, int HW, bf16_t *__restrict__ y,
                                     bf16_t *__restrict__ y_lo, int fmt) {
    const long long total = (long long)N * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / HW;
        const int p = (int)(i - n * HW);
        const float *s = x + n * 3 * HW + p;
        const float c0 = s[0], c1 = s[HW], c2 = s[2 * HW];
        const bf16_t h0 = f32_to_h16(c0, fmt), h1 = f32_to_h16(c1, fmt), h2 = f32_to_h16(c2, fmt);
        reinterpret_cast<uint2 *>(y)[i] = make_uint2((uint32_t)h0 | ((uint32_t)h1 << 16), (uint32_t)h2);
        if (y_lo != nullptr) {
            const bf16_t l0 = f32_to_h16(c0 - h16_to_f32(h0, fmt), fmt), l1 = f32_to_h16(c1 - h16_to_f32(h1, fmt), fmt),
                         l2 = f32_to_h16(c2 - h16_to_f32(h2, fmt), fmt);
            reinterpret_cast<uint2 *>(y_lo)[i] = make_uint2((uint32_t)l0 | ((uint32_t)l1 << 16), (uint32_t)l2);
        }
    }
}

// The same with a zero border baked in: out [N][Hp][Wp][4], pixel (py, px) = input (py - pad, px - pad) or zeros.  The f16x3
// stem reads this buffer without bounds tests (conv_igemm.hip, small_cin == 3): the 8-pixel window of a kernel row is 64
// contiguous, 16-byte aligned bytes of either plane.
__global__ void nchw_to_nhwc4_pad_kernel(const float *__restrict__ x, int N, int H, int W, int Hp, int Wp, int pad,
                                         bf16_t *__restrict__ y, bf16_t *__restrict__ y_lo, int fmt) {
    const long long total = (long long)N * Hp * Wp;
    const long long HW = (long long)H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int px = (int)(i % Wp);
        const long long r = i / Wp;
        const int py = (int)(r % Hp);
        const long long n = r / Hp;
        const int iy = py - pad, ix = px - pad;
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
            const float *s = x + n * 3 * HW + (long long)iy * W + ix;
            c0 = s[0]; c1 = s[HW]; c2 = s[2 * HW];
        }
        const bf16_t h0 = f32_to_h16(c0, fmt), h1 = f32_to_h16(c1, fmt), h2 = f32_to_h16(c2, fmt);
        reinterpret_cast<uint2 *>(y)[i] = make_uint2((uint32_t)h0 | ((uint32_t)h1 << 16), (uint32_t)h2);
        if (y_lo != nullptr) {
            const bf16_t l0 = f32_to_h16(c0 - h16_to_f32(h0, fmt), fmt), l1 = f32_to_h16(c1 - h16_to_f32(h1, fmt), fmt),
                         l2 = f32_to_h16(c2 - h16_to_f32(h2, fmt), fmt);
            reinterpret_cast<uint2 *>(y_lo)[i] = make_uint2((uint32_t)l0 | ((uint32_t)l1 << 16), (uint32_t)l2);
        }
    }
}

// Strided pixel gather into a channel range of a wider tensor: y[(n, ho, wo)][0 .. C) = x[n][ho * stride][wo * stride][0 .. C),
// rows of y `ldy` elements apart -- the shortcut input of a ResNet stage's first block, placed beside conv2's output so that
// conv3 and the 1x1 projection are ONE GEMM over the concatenated channels (net.hip build_resnet50_backbone).  16 bytes per thread.
__global__ __launch_bounds__(256) void gather_strided_kernel(const bf16_t *__restrict__ x, const bf16_t *__restrict__ x_lo, int H, int W,
                                                             int C8, int stride, int Ho, int Wo, bf16_t *__restrict__ y,
                                                             bf16_t *__restrict__ y_lo, int ldy, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        long long m = i / C8;
        const int wo = (int)(m % Wo);
        const long long r = m / Wo;
        const int ho = (int)(r % Ho);
        const long long n = r / Ho;
        const long long src = (((n * H + (long long)ho * stride) * W + (long long)wo * stride) * C8 + c8) * 8;
        const long long dst = m * ldy + c8 * 8;
        *reinterpret_cast<uint4 *>(y + dst) = *reinterpret_cast<const uint4 *>(x + src);
        if (x_lo != nullptr) *reinterpret_cast<uint4 *>(y_lo + dst) = *reinterpret_cast<const uint4 *>(x_lo + src);
    }
}

// nn.MaxPool2d(k, stride, pad) on NHWC bf16, 8 channels per thread.
// resnet50.py:64 (3x3 s2 p1), common_cnn.py:131-132 (2x2 s2).
__global__ void maxpool_kernel(const bf16_t *__restrict__ x, const bf16_t *__restrict__ x_lo, int N, int H,
                               int W, int C, int k, int stride, int pad, int Ho, int Wo,
                               bf16_t *__restrict__ y, bf16_t *__restrict__ y_lo, int fmt) {
    const int C8 = C >> 3;
    const long long total = (long long)N * Ho * Wo * C8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        long long pix = i / C8;
        const int wo = (int)(pix % Wo);
        pix /= Wo;
        const int ho = (int)(pix % Ho);
        const int n = (int)(pix / Ho);
        float best[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) best[j] = -3.0e38f;
        for (int dy = 0; dy < k; ++dy) {
            const int hi = ho * stride - pad + dy;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int dx = 0; dx < k; ++dx) {
                const int wi = wo * stride - pad + dx;
                if ((unsigned)wi >= (unsigned)W) continue;
                const long long o = (((long long)n * H + hi) * W + wi) * C + c8 * 8;
                const uint4 v = *reinterpret_cast<const uint4 *>(x + o);
                const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
                float f[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f[2 * j] = h16_to_f32((bf16_t)(vw[j] & 0xffffu), fmt);
                    f[2 * j + 1] = h16_to_f32((bf16_t)(vw[j] >> 16), fmt);
                }
                if (x_lo != nullptr) {
                    const uint4 l = *reinterpret_cast<const uint4 *>(x_lo + o);
                    const uint32_t lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f[2 * j] += h16_to_f32((bf16_t)(lw[j] & 0xffffu), fmt);
                        f[2 * j + 1] += h16_to_f32((bf16_t)(lw[j] >> 16), fmt);
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) best[j] = fmaxf(best[j], f[j]);
            }
        }
        uint32_t hw[4], lw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16_t h0 = f32_to_h16(best[2 * j], fmt), h1 = f32_to_h16(best[2 * j + 1], fmt);
            hw[j] = (uint32_t)h0 | ((uint32_t)h1 << 16);
            const bf16_t l0 = f32_to_h16(best[2 * j] - h16_to_f32(h0, fmt), fmt);
            const bf16_t l1 = f32_to_h16(best[2 * j + 1] - h16_to_f32(h1, fmt), fmt);
            lw[j] = (uint32_t)l0 | ((uint32_t)l1 << 16);
        }
        const long long oo = (((long long)n * Ho + ho) * Wo + wo) * C + c8 * 8;
        *reinterpret_cast<uint4 *>(y + oo) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        if (y_lo != nullptr) *reinterpret_cast<uint4 *>(y_lo + oo) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
}

// The same pooling for IEEE-half activations with one precision plane (the default path): the maximum of halves is taken
// on the halves themselves (v_pk_max_f16, two channels per instruction -- exact, no conversion), one block row per output
// row (blockIdx.y = n * Ho + ho: no 64-bit index divisions).  8 channels = one 16-byte load per tap.
__global__ __launch_bounds__(256) void maxpool_f16_kernel(const bf16_t *__restrict__ x, int H, int W, int C, int k, int stride,
                                                          int pad, int Ho, int Wo, bf16_t *__restrict__ y) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const int C8 = C >> 3;
    const int row = blockIdx.y; // n * Ho + ho
    const int n = row / Ho, ho = row - n * Ho;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Wo * C8; i += gridDim.x * blockDim.x) {
        const int wo = i / C8, c8 = i - wo * C8;
        const h2_t lowest = {(_Float16)-65504.f, (_Float16)-65504.f};
        h2_t best[4] = {lowest, lowest, lowest, lowest};
        bool any = false;
        for (int dy = 0; dy < k; ++dy) {
            const int hi = ho * stride - pad + dy;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int dx = 0; dx < k; ++dx) {
                const int wi = wo * stride - pad + dx;
                if ((unsigned)wi >= (unsigned)W) continue;
                const uint4 v = *reinterpret_cast<const uint4 *>(x + ((((long long)n * H + hi) * W + wi) * C + c8 * 8));
                const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) best[j] = __builtin_elementwise_max(best[j], __builtin_bit_cast(h2_t, vw[j]));
                any = true;
            }
        }
        (void)any;
        *reinterpret_cast<uint4 *>(y + (((long long)row * Wo + wo) * C + c8 * 8)) =
            make_uint4(__builtin_bit_cast(uint32_t, best[0]), __builtin_bit_cast(uint32_t, best[1]),
                       __builtin_bit_cast(uint32_t, best[2]), __builtin_bit_cast(uint32_t, best[3]));
    }
}

// The same mapping for the two-plane half activations of the f16x3 mode: the value of a tap is hi + lo (exact in fp32: 22
// bits), the maximum is taken on the values and split again (value-exact: the planes of the maximum are re-derived, the
// value is one of the inputs').  The generic kernel above spends a 64-bit index division chain per 8 channels and ran the
// ResNet stem's pool at 3.75 TB/s (142 us for 532 MB).
__global__ __launch_bounds__(256) void maxpool_f16x2_kernel(const bf16_t *__restrict__ x, const bf16_t *__restrict__ x_lo, int H, int W,
                                                            int C, int k, int stride, int pad, int Ho, int Wo, bf16_t *__restrict__ y,
                                                            bf16_t *__restrict__ y_lo) {
    const int C8 = C >> 3;
    const int row = blockIdx.y; // n * Ho + ho
    const int n = row / Ho, ho = row - n * Ho;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Wo * C8; i += gridDim.x * blockDim.x) {
        const int wo = i / C8, c8 = i - wo * C8;
        float best[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) best[j] = -3.0e38f;
        for (int dy = 0; dy < k; ++dy) {
            const int hi = ho * stride - pad + dy;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int dx = 0; dx < k; ++dx) {
                const int wi = wo * stride - pad + dx;
                if ((unsigned)wi >= (unsigned)W) continue;
                const long long o = (((long long)n * H + hi) * W + wi) * C + c8 * 8;
                const uint4 v = *reinterpret_cast<const uint4 *>(x + o), l = *reinterpret_cast<const uint4 *>(x_lo + o);
                const uint32_t vw[4] = {v.x, v.y, v.z, v.w}, lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    best[2 * j] = fmaxf(best[2 * j], f16_to_f32((bf16_t)(vw[j] & 0xffffu)) + f16_to_f32((bf16_t)(lw[j] & 0xffffu)));
                    best[2 * j + 1] = fmaxf(best[2 * j + 1], f16_to_f32((bf16_t)(vw[j] >> 16)) + f16_to_f32((bf16_t)(lw[j] >> 16)));
                }
            }
        }
        uint32_t hw[4], lw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16_t h0 = f32_to_f16(best[2 * j]), h1 = f32_to_f16(best[2 * j + 1]);
            hw[j] = (uint32_t)h0 | ((uint32_t)h1 << 16);
            const bf16_t l0 = f32_to_f16(best[2 * j] - f16_to_f32(h0)), l1 = f32_to_f16(best[2 * j + 1] - f16_to_f32(h1));
            lw[j] = (uint32_t)l0 | ((uint32_t)l1 << 16);
        }
        const long long oo = ((long long)row * Wo + wo) * C + c8 * 8;
        *reinterpret_cast<uint4 *>(y + oo) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        *reinterpret_cast<uint4 *>(y_lo + oo) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
}

// x = relu(head); cam = x[0] + x[1].flip(-1)   (resnet50_cam.py:66-68, vgg16_cam.py:49-50)
// head: fp32 [2B][h][w][Cs] (NHWC, first C channels valid) -> cam fp32 [B][C][h][w]
__global__ void flip_add_kernel(const float *__restrict__ head, int B, int h, int w, int C, int Cs,
                                float *__restrict__ cam) {
    const long long total = (long long)B * C * h * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % w);
        long long r = i / w;
        const int yy = (int)(r % h);
        r /= h;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        const float a = head[(((long long)(2 * b) * h + yy) * w + xx) * Cs + c];
        const float f = head[(((long long)(2 * b + 1) * h + yy) * w + (w - 1 - xx)) * Cs + c];
        cam[i] = fmaxf(a, 0.f) + fmaxf(f, 0.f);
    }
}

// Classifier branch of vgg16_cam.py:34-36 on sample 0 of each image:
// score[b][c] = sigmoid(bias[c] + sum_f Wc[c][f] * mean_hw feat[sample_stride*b][.][f])
// Global pooling + Linear + Sigmoid of the classifier branch, in two kernels: (1) a block per (image, 64 channels), lane =
// channel, the positions dealt to the block's GAP_WAVES waves in contiguous runs: every wave sums its run in order (fp32, eight
// independent loads in flight) and wave 0 adds the GAP_WAVES partial sums in run order -- a fixed summation order, whatever
// the batch.  (One wave per (image, 64 channels) walked all 1600 positions of a 40 x 40 map alone: 200 dependent round
// trips, 450 us for 16 images; one block per image before that: 1.9 ms.)  (2) one block per image for the C dot products.
// feat NHWC half / bf16 (+ lo plane).  hw < 0 selects the global max of m7 (m7_cam.py:32-35: MaxPool 2x2 then
// AdaptiveMaxPool2d((1,1))).
constexpr int GAP_WAVES = 16;
__global__ __launch_bounds__(64 * GAP_WAVES) void gap_kernel(const bf16_t *__restrict__ feat, const bf16_t *__restrict__ feat_lo,
                                                             int hw, int F, float *__restrict__ gap, int fmt, int sample_stride) {
    __shared__ float part[GAP_WAVES][64];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int f = blockIdx.x * 64 + lane;
    const bool use_max = hw < 0; // m7: AdaptiveMaxPool2d((1,1)) instead of the average
    const int npix = use_max ? -hw : hw;
    const int run = (npix + GAP_WAVES - 1) / GAP_WAVES;
    const int pb = wv * run, pe = min(npix, pb + run);
    float s = use_max ? -3.0e38f : 0.f;
    if (f < F) {
        const long long img = (long long)(sample_stride * b) * npix * F;
        const bf16_t *f0 = feat + img + f;
        const bf16_t *l0 = feat_lo ? feat_lo + img + f : nullptr;
        int p = pb;
        for (; p + 8 <= pe; p += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = h16_to_f32(f0[(long long)(p + u) * F], fmt);
                if (l0) v[u] += h16_to_f32(l0[(long long)(p + u) * F], fmt);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s = use_max ? fmaxf(s, v[u]) : s + v[u];
        }
        for (; p < pe; ++p) {
            float v = h16_to_f32(f0[(long long)p * F], fmt);
            if (l0) v += h16_to_f32(l0[(long long)p * F], fmt);
            s = use_max ? fmaxf(s, v) : s + v;
        }
    }
    part[wv][lane] = s;
    __syncthreads();
    if (wv == 0 && f < F) {
        float t = part[0][lane];
        for (int k = 1; k < GAP_WAVES; ++k) t = use_max ? fmaxf(t, part[k][lane]) : t + part[k][lane];
        gap[(long long)b * F + f] = use_max ? t : t / (float)npix;
    }
}

__global__ void linear_sigmoid_kernel(const float *__restrict__ gapbuf, int F, const float *__restrict__ Wc,
                                      const float *__restrict__ bias, int C, float *__restrict__ score) {
    extern __shared__ float gap[]; // F floats
    const int b = blockIdx.x;
    for (int f = threadIdx.x; f < F; f += blockDim.x) gap[f] = gapbuf[(long long)b * F + f];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int c = wv; c < C; c += nw) {
        float s = 0.f;
        for (int f = lane; f < F; f += 64) s += gap[f] * Wc[(long long)c * F + f];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) score[(long long)b * C + c] = 1.f / (1.f + expf(-(s + (bias ? bias[c] : 0.f))));
    }
}

__global__ void bf16_to_f32_kernel(const bf16_t *__restrict__ x, const bf16_t *__restrict__ x_lo, size_t n,
                                   float *__restrict__ y, int fmt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = h16_to_f32(x[i], fmt);
        if (x_lo) v += h16_to_f32(x_lo[i], fmt);
        y[i] = v;
    }
}

// generic layout changes for the single-layer entry point (wsc_conv2d_nchw)
__global__ void nchw_to_nhwc_kernel(const float *__restrict__ x, int N, int C, int HW, bf16_t *__restrict__ y,
                                    bf16_t *__restrict__ y_lo, int fmt) {
    const long long total = (long long)N * C * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long r = i / C;
        const int p = (int)(r % HW);
        const long long n = r / HW;
        const float v = x[(n * C + c) * HW + p];
        const bf16_t h = f32_to_h16(v, fmt);
        y[i] = h;
        if (y_lo) y_lo[i] = f32_to_h16(v - h16_to_f32(h, fmt), fmt);
    }
}
__global__ void nhwc_to_nchw_kernel(const bf16_t *__restrict__ x, const bf16_t *__restrict__ x_lo, int N, int C,
                                    int HW, float *__restrict__ y, int fmt) {
    const long long total = (long long)N * C * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const long long r = i / HW;
        const int c = (int)(r % C);
        const long long n = r / C;
        const long long src = (n * HW + p) * C + c;
        float v = h16_to_f32(x[src], fmt);
        if (x_lo) v += h16_to_f32(x_lo[src], fmt);
        y[i] = v;
    }
}

inline int grid_for(long long total, int block = 256, int cap = 256 * 16) {
    long long g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

} // namespace

int launch_nchw_to_nhwc4(wsc_ctx *ctx, const float *x, int N, int H, int W, bf16_t *y, bf16_t *y_lo, int fmt) {
    const long long total = (long long)N * H * W;
    WscKernelTimer timer(ctx, WSC_K_POOL_MISC, (double)total * (12 + 8));
    hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, x, N, H * W, y,
                       y_lo, fmt);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

int launch_nchw_to_nhwc4_pad(wsc_ctx *ctx, const float *x, int N, int H, int W, int Hp, int Wp, int pad, bf16_t *y, bf16_t *y_lo,
                             int fmt) {
    const long long total = (long long)N * Hp * Wp;
    WscKernelTimer timer(ctx, WSC_K_POOL_MISC, (double)N * H * W * 12 + (double)total * (y_lo ? 16 : 8));
    hipLaunchKernelGGL(nchw_to_nhwc4_pad_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, x, N, H, W, Hp, Wp, pad, y,
                       y_lo, fmt);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

int launch_gather_strided(wsc_ctx *ctx, const bf16_t *x, const bf16_t *x_lo, int N, int H, int W, int C, int stride, int Ho, int Wo,
                          bf16_t *y, bf16_t *y_lo, int ldy) {
    WSC_CHECK(C % 8 == 0 && ldy % 8 == 0, WSC_ERR_INVALID, "gather: C=%d, pitch=%d not multiples of 8", C, ldy);
    const long long total = (long long)N * Ho * Wo * (C / 8);
    WscKernelTimer timer(ctx, WSC_K_POOL_MISC, (double)total * 32 * (x_lo ? 2 : 1));
    hipLaunchKernelGGL(gather_strided_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, x, x_lo, H, W, C / 8, stride, Ho, Wo, y,
                       y_lo, ldy, total);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

int launch_maxpool(wsc_ctx *ctx, const bf16_t *x, const bf16_t *x_lo, int N, int H, int W, int C, int k,
                   int stride, int pad, int Ho, int Wo, bf16_t *y, bf16_t *y_lo, int fmt) {
    WSC_CHECK(C % 8 == 0, WSC_ERR_INVALID, "maxpool: C=%d not a multiple of 8", C);
    const long long total = (long long)N * Ho * Wo * (C / 8);
    WscKernelTimer timer(ctx, WSC_K_POOL_MISC, ((double)N * H * W * C + (double)N * Ho * Wo * C) * 2);
    if (fmt == 1 && x_lo == nullptr && y_lo == nullptr && (long long)N * Ho <= 65535) {
        const int per_row = Wo * (C / 8);
        hipLaunchKernelGGL(maxpool_f16_kernel, dim3((unsigned)((per_row + 255) / 256), (unsigned)(N * Ho)), dim3(256), 0, ctx->stream,
                           x, H, W, C, k, stride, pad, Ho, Wo, y);
        WSC_HIP(hipGetLastError());
        return WSC_OK;
    }
    if (fmt == 1 && x_lo != nullptr && y_lo != nullptr && (long long)N * Ho <= 65535) {
        const int per_row = Wo * (C / 8);
        hipLaunchKernelGGL(maxpool_f16x2_kernel, dim3((unsigned)((per_row + 255) / 256), (unsigned)(N * Ho)), dim3(256), 0, ctx->stream,
                           x, x_lo, H, W, C, k, stride, pad, Ho, Wo, y, y_lo);
        WSC_HIP(hipGetLastError());
        return WSC_OK;
    }
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, x, x_lo, N, H, W, C, k,
                       stride, pad, Ho, Wo, y, y_lo, fmt);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

int launch_flip_add(wsc_ctx *ctx, const float *head, int B, int h, int w, int C, int Cs, float *cam) {
    const long long total = (long long)B * C * h * w;
    WscKernelTimer timer(ctx, WSC_K_POOL_MISC, (double)total * 12);
    hipLaunchKernelGGL(flip_add_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, head, B, h, w, C, Cs,
                       cam);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

int launch_gap_linear_sigmoid(wsc_ctx *ctx, const bf16_t *feat, const bf16_t *feat_lo, int B, int hw, int F,
                              const float *Wc, const float *bias, int C, float *score, int fmt, int sample_stride) {
    float *gapbuf = nullptr;
    WSC_TRY(wsc_ctx_cached_alloc(ctx, sizeof(float) * (size_t)B * F, (void **)&gapbuf));
    WscCachedGuard gapbuf_guard(ctx, gapbuf);
    hipLaunchKernelGGL(gap_kernel, dim3((unsigned)((F + 63) / 64), (unsigned)B), dim3(64 * GAP_WAVES), 0, ctx->stream, feat, feat_lo, hw, F, gapbuf,
                       fmt, sample_stride);
    hipLaunchKernelGGL(linear_sigmoid_kernel, dim3(B), dim3(256), F * sizeof(float), ctx->stream, (const float *)gapbuf, F, Wc, bias, C,
                       score);
    WSC_HIP(hipGetLastError());
    gapbuf_guard.free_now(); // stream-ordered reuse
    return WSC_OK;
}

int launch_bf16_to_f32(wsc_ctx *ctx, const bf16_t *x, const bf16_t *x_lo, size_t n, float *y, int fmt) {
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for((long long)n)), dim3(256), 0, ctx->stream, x, x_lo, n,
                       y, fmt);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

int launch_nchw_to_nhwc(wsc_ctx *ctx, const float *x, int N, int C, int HW, bf16_t *y, bf16_t *y_lo, int fmt) {
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((long long)N * C * HW)), dim3(256), 0, ctx->stream, x, N, C,
                       HW, y, y_lo, fmt);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

int launch_nhwc_to_nchw(wsc_ctx *ctx, const bf16_t *x, const bf16_t *x_lo, int N, int C, int HW, float *y, int fmt) {
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((long long)N * C * HW)), dim3(256), 0, ctx->stream, x, x_lo,
                       N, C, HW, y, fmt);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}
