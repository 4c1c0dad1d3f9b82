// cam_tail.hip -- the tail of make_cam._work for a batch of images
// (03b_irn/step/make_cam.py:41-42, 62-76):
//   strided_cam = F.interpolate(cam, strided_size, 'bilinear', align_corners=False)[valid_cat]
//   highres_cam = F.interpolate(cam, strided_up_size, ...)[valid_cat][:, :H0, :W0]
//   x /= F.adaptive_max_pool2d(x, (1,1)) + 1e-5      (per channel)
// HBM-bound on the output write (K*(h4*w4 + H0*W0)*4 bytes per image); the 21x21 source
// map of a job lives in LDS.  Two launches: (1) per-channel spatial max of the upsampled
// maps, (2) recompute + divide + one coalesced write.  Nothing but the final result is
// ever written to HBM.
//
// Bilinear arithmetic follows torch's CPU kernel for align_corners=False:
//   scale = in/out (fp32); src = scale*(dst+0.5)-0.5, clamped at 0; i0=(int)src;
//   i1 = i0 + (i0 < in-1); l1 = src - i0; l0 = 1 - l1;
//   out = lh0*(lw0*a + lw1*b) + lh1*(lw0*c + lw1*d)
#include "common.h"

#include <algorithm>
#include <cstring>

namespace {

struct TailJob {
    long long cam_off;     // float offset of cam[b][key] (h*w floats)
    long long strided_off; // float offset of this channel's strided output
    long long highres_off; // float offset of this channel's high_res output
    int H0, W0, h4, w4, Hu, Wu;
};

constexpr int PIX_PER_BLOCK = 4096;

__device__ __forceinline__ void src_index(int dst, float scale, int in, int &i0, int &i1, float &l0, float &l1) {
    float s = __builtin_fmaf(scale, (float)dst + 0.5f, -0.5f);
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
    l0 = 1.f - l1;
}

__device__ __forceinline__ float bilerp(const float *src, int w, int y0, int y1, float ly0, float ly1, int x0, int x1,
                                        float lx0, float lx1) {
    // the fused multiply-adds are spelled out (not left to -ffp-contract): every kernel that samples a map --
    // cam_tail_kernel, cam_max_kernel, cam_unary_kernel, bilinear_kernel -- then computes the same bits
    const float top = __builtin_fmaf(lx0, src[y0 * w + x0], lx1 * src[y0 * w + x1]);
    const float bot = __builtin_fmaf(lx0, src[y1 * w + x0], lx1 * src[y1 * w + x1]);
    return __builtin_fmaf(ly0, top, ly1 * bot);
}

// -log(clip(p, 1e-5, 1)) of pydensecrf.utils.unary_from_softmax (03c_hsn/utilities.py:431): the clip is one median, the
// logarithm the hardware log2 times ln 2 (|error| <= ~2e-6 on values up to 11.5; the tests hold the unaries to 2e-5 x
// max(1, |ref|)).  One helper for every kernel that writes unaries, so the fused and the two-step paths agree bit for bit.
__device__ __forceinline__ float neg_log_clip(float p) { return -__logf(__builtin_amdgcn_fmed3f(p, 1e-5f, 1.f)); }

// Order-preserving float <-> uint code for atomicMax over floats of EITHER sign: the ADP background channel
// (bg - max exception CAM, common_cam.py:57-75, no ReLU) can be <= 0 everywhere, and the reference then divides by
// (negative max + 1e-5).  Code 0 (the memset value) is below every real number; it decodes to 0 (empty map).
__device__ __forceinline__ unsigned int ord_enc(float f) {
    const unsigned int b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord_dec(unsigned int u) {
    if (u == 0u) return 0.f;
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// WRITE = false: atomicMax the per-job maxima; WRITE = true: write v / (max + 1e-5).
template <bool WRITE>
__global__ __launch_bounds__(256) void cam_tail_kernel(const float *__restrict__ cam, const TailJob *__restrict__ jobs,
                                                       int h, int w, unsigned int *__restrict__ mx,
                                                       float *__restrict__ strided, float *__restrict__ highres) {
    extern __shared__ float src[]; // h*w
    const TailJob job = jobs[blockIdx.y];
    const int n_hi = job.H0 * job.W0;
    const int n_st = job.h4 * job.w4;
    const int start = blockIdx.x * PIX_PER_BLOCK;
    if (start >= n_hi + n_st) return;
    for (int i = threadIdx.x; i < h * w; i += blockDim.x) src[i] = cam[job.cam_off + i];
    __syncthreads();

    const float sh_hi = (float)h / (float)job.Hu, sw_hi = (float)w / (float)job.Wu;
    const float sh_st = (float)h / (float)job.h4, sw_st = (float)w / (float)job.w4;
    float d_hi = 1.f, d_st = 1.f;
    if (WRITE) {
        d_hi = ord_dec(mx[2 * blockIdx.y]) + 1e-5f;
        d_st = ord_dec(mx[2 * blockIdx.y + 1]) + 1e-5f;
    }
    float m_hi = -3.0e38f, m_st = -3.0e38f;
    const int end = min(start + PIX_PER_BLOCK, n_hi + n_st);
    for (int i = start + threadIdx.x; i < end; i += blockDim.x) {
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        if (i < n_hi) {
            const int yy = i / job.W0, xx = i - yy * job.W0;
            src_index(yy, sh_hi, h, y0, y1, ly0, ly1);
            src_index(xx, sw_hi, w, x0, x1, lx0, lx1);
            const float v = bilerp(src, w, y0, y1, ly0, ly1, x0, x1, lx0, lx1);
            if (WRITE) highres[job.highres_off + i] = v / d_hi;
            else m_hi = fmaxf(m_hi, v);
        } else {
            const int k = i - n_hi;
            const int yy = k / job.w4, xx = k - yy * job.w4;
            src_index(yy, sh_st, h, y0, y1, ly0, ly1);
            src_index(xx, sw_st, w, x0, x1, lx0, lx1);
            const float v = bilerp(src, w, y0, y1, ly0, ly1, x0, x1, lx0, lx1);
            if (WRITE) strided[job.strided_off + k] = v / d_st;
            else m_st = fmaxf(m_st, v);
        }
    }
    if (!WRITE) {
        for (int o = 32; o > 0; o >>= 1) {
            m_hi = fmaxf(m_hi, __shfl_down(m_hi, o, 64));
            m_st = fmaxf(m_st, __shfl_down(m_st, o, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            if (m_hi > -3.0e38f) atomicMax(&mx[2 * blockIdx.y], ord_enc(m_hi));
            if (m_st > -3.0e38f) atomicMax(&mx[2 * blockIdx.y + 1], ord_enc(m_st));
        }
    }
}

// The maxima of cam_tail_kernel<false> in the column form of cam_max_kernel: gridDim.y blocks per job (each a band of output
// rows), a thread per output column, the row taps of the band's output rows in an LDS table, a column's horizontal taps recomputed only when the source rows change.
// Same bilerp operands per sample as the per-pixel form: the same maxima, without a division and two src_index calls per sample.
__global__ __launch_bounds__(512) void cam_tail_max_kernel(const float *__restrict__ cam, const TailJob *__restrict__ jobs, int h,
                                                           int w, unsigned int *__restrict__ mx) {
    extern __shared__ float src[]; // h*w, then 16 wave maxima, then row taps {y0*w, y1*w, ly0, ly1} of max(H0, h4) output rows
    const TailJob job = jobs[blockIdx.x];
    float4 *rows = reinterpret_cast<float4 *>(src + ((h * w + 16 + 3) & ~3));
    for (int i = threadIdx.x; i < h * w; i += blockDim.x) src[i] = cam[job.cam_off + i];
    float result[2];
    for (int part = 0; part < 2; ++part) { // 0: high_res (H0 x W0 of the Hu x Wu upsampling), 1: strided (h4 x w4)
        const int Hall = part == 0 ? job.H0 : job.h4, Wo = part == 0 ? job.W0 : job.w4;
        const int band = (Hall + (int)gridDim.y - 1) / (int)gridDim.y;
        const int yb = (int)blockIdx.y * band, Ho = max(0, min(band, Hall - yb)); // this block's rows [yb, yb + Ho)
        const float sh = (float)h / (float)(part == 0 ? job.Hu : job.h4), sw = (float)w / (float)(part == 0 ? job.Wu : job.w4);
        __syncthreads(); // (the source map is staged / the previous part is done with the table)
        for (int yy = threadIdx.x; yy < Ho; yy += blockDim.x) {
            int y0, y1;
            float ly0, ly1;
            src_index(yb + yy, sh, h, y0, y1, ly0, ly1);
            rows[yy] = make_float4(__int_as_float(y0 * w), __int_as_float(y1 * w), ly0, ly1);
        }
        __syncthreads();
        float m = -3.0e38f;
        for (int xx = threadIdx.x; xx < Wo && Ho > 0; xx += blockDim.x) {
            int x0, x1;
            float lx0, lx1;
            src_index(xx, sw, w, x0, x1, lx0, lx1);
            int o0 = -1, o1 = -1;
            float top = 0.f, bot = 0.f;
            for (int yy = 0; yy < Ho; ++yy) {
                const float4 r = rows[yy];
                const int n0 = __float_as_int(r.x), n1 = __float_as_int(r.y);
                if (n0 != o0 || n1 != o1) {
                    const float *r0 = src + n0, *r1 = src + n1;
                    top = __builtin_fmaf(lx0, r0[x0], lx1 * r0[x1]);
                    bot = __builtin_fmaf(lx0, r1[x0], lx1 * r1[x1]);
                    o0 = n0;
                    o1 = n1;
                }
                m = fmaxf(m, __builtin_fmaf(r.z, top, r.w * bot));
            }
        }
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
        float *wm = src + h * w;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, wm[i]);
            result[part] = m;
        }
    }
    if (threadIdx.x == 0) { // (mx is cleared by the launch function; 0 = no value)
        if (result[0] > -3.0e38f) atomicMax(&mx[2 * blockIdx.x], ord_enc(result[0]));
        if (result[1] > -3.0e38f) atomicMax(&mx[2 * blockIdx.x + 1], ord_enc(result[1]));
    }
}

// plain F.interpolate(bilinear, align_corners=False): [C][h][w] -> [C][H][W]
__global__ void bilinear_kernel(const float *__restrict__ src, int C, int h, int w, float *__restrict__ dst, int H,
                                int W) {
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const long long total = (long long)C * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W);
        const long long r = i / W;
        const int yy = (int)(r % H);
        const int c = (int)(r / H);
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        src_index(yy, sh, h, y0, y1, ly0, ly1);
        src_index(xx, sw, w, x0, x1, lx0, lx1);
        dst[i] = bilerp(src + (long long)c * h * w, w, y0, y1, ly0, ly1, x0, x1, lx0, lx1);
    }
}

// U[b][m][p] = -log(clip(v_m / sum_m v_m, 1e-5, 1)),  v_0 = bg_value, v_{c+1} = maps[b][c][p]
// (eval_cam.py:49-51 pads the max-normalised CAMs with a constant background channel;
//  pydensecrf.utils.unary_from_softmax, 03c_hsn/utilities.py:431, turns probabilities into unaries)
__global__ void unary_from_maps_kernel(const float *__restrict__ maps, float bg, int C, int N, long long total,
                                       float *__restrict__ unary) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / N;
        const int p = (int)(i - b * N);
        const float *src = maps + b * C * N + p;
        float sum = bg;
        for (int c = 0; c < C; ++c) sum += src[(long long)c * N];
        float *dst = unary + b * (C + 1) * N + p;
        const float inv = 1.f / sum;
        dst[0] = neg_log_clip(bg * inv);
        for (int c = 0; c < C; ++c) dst[(long long)(c + 1) * N] = neg_log_clip(src[(long long)c * N] * inv);
    }
}

// Fused form of cam_tail (high_res of ALL C classes at one size) + unary_from_maps: nothing but the unaries is
// written.  Pass 1: per (image, class) maximum of the upsampled, cropped map; pass 2: a thread owns one pixel,
// recomputes the C bilinear samples from the image's C source maps in LDS (C*h*w floats), divides by
// (max + 1e-5), and applies the unary formula.  Same float operations in the same order as the two-step
// path (cam_tail_kernel<true> then unary_from_maps_kernel), so the results are bit-identical.
__global__ __launch_bounds__(1024) void cam_max_kernel(const float *__restrict__ cam, int C, int h, int w, int H0, int W0,
                                                       int Hu, int Wu, unsigned int *__restrict__ mx) {
    // One block per (image, class), a thread per output COLUMN: the column's source taps are computed once, the row's
    // taps are uniform over the block, and a sample costs four LDS reads + bilerp -- the per-sample index arithmetic (a
    // division and two src_index calls) made the first version instruction-bound at 190-360 us for 640 maps at 321 x 321.
    // bilerp gets the same operands as in cam_unary_kernel: the maximum is the maximum of exactly the values written there.
    extern __shared__ float src[]; // h*w, then 16 wave maxima, then the row taps {y0*w, y1*w, ly0, ly1} of every output row
    const int bc = blockIdx.x;     // b*C + c
    float4 *rows = reinterpret_cast<float4 *>(src + ((h * w + 16 + 3) & ~3));
    for (int i = threadIdx.x; i < h * w; i += blockDim.x) src[i] = cam[(long long)bc * h * w + i];
    const float sh = (float)h / (float)Hu, sw = (float)w / (float)Wu;
    for (int yy = threadIdx.x; yy < H0; yy += blockDim.x) {
        int y0, y1;
        float ly0, ly1;
        src_index(yy, sh, h, y0, y1, ly0, ly1);
        rows[yy] = make_float4(__int_as_float(y0 * w), __int_as_float(y1 * w), ly0, ly1);
    }
    __syncthreads();
    float m = -3.0e38f;
    for (int xx = threadIdx.x; xx < W0; xx += blockDim.x) {
        int x0, x1;
        float lx0, lx1;
        src_index(xx, sw, w, x0, x1, lx0, lx1);
        // the two source rows of an output row change every Hu / h (= 16) output rows: the column's horizontal taps
        // `top` / `bot` are recomputed only then (same values as per row: the maximum is unchanged, a sixteenth of the LDS reads)
        int o0 = -1, o1 = -1;
        float top = 0.f, bot = 0.f;
#pragma unroll 4
        for (int yy = 0; yy < H0; ++yy) {
            const float4 r = rows[yy]; // uniform over the block
            // bilerp(src, w, y0, y1, ly0, ly1, x0, x1, lx0, lx1) with the row offsets taken from the table
            const int n0 = __float_as_int(r.x), n1 = __float_as_int(r.y);
            if (n0 != o0 || n1 != o1) {
                const float *r0 = src + n0, *r1 = src + n1;
                top = __builtin_fmaf(lx0, r0[x0], lx1 * r0[x1]);
                bot = __builtin_fmaf(lx0, r1[x0], lx1 * r1[x1]);
                o0 = n0;
                o1 = n1;
            }
            m = fmaxf(m, __builtin_fmaf(r.z, top, r.w * bot));
        }
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    float *wm = src + h * w;
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, wm[i]);
        mx[bc] = m > -3.0e38f ? ord_enc(m) : 0u;
    }
}

template <int CMAX, bool PM>
__global__ __launch_bounds__(256) void cam_unary_kernel(const float *__restrict__ cam, int C, int h, int w, int H0, int W0,
                                                        int Hu, int Wu, const unsigned int *__restrict__ mx, float bg,
                                                        float *__restrict__ unary) {
    extern __shared__ float src[]; // C*h*w maps of this image, then C divisors
    const int b = blockIdx.y;
    const int hw = h * w;
    float *div = src + C * hw;
    if (((C * hw) & 3) == 0) { // 16-byte staging loads (20 x 21 x 21 floats = 2205 float4: 9 per thread instead of 35)
        const float4 *g4 = reinterpret_cast<const float4 *>(cam + (long long)b * C * hw);
        float4 *s4 = reinterpret_cast<float4 *>(src);
        for (int i = threadIdx.x; i < (C * hw) >> 2; i += blockDim.x) s4[i] = g4[i];
    } else {
        for (int i = threadIdx.x; i < C * hw; i += blockDim.x) src[i] = cam[(long long)b * C * hw + i];
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) div[c] = ord_dec(mx[b * C + c]) + 1e-5f;
    __syncthreads();
    const float sh = (float)h / (float)Hu, sw = (float)w / (float)Wu;
    const int n = H0 * W0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int yy = i / W0, xx = i - yy * W0;
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        src_index(yy, sh, h, y0, y1, ly0, ly1);
        src_index(xx, sw, w, x0, x1, lx0, lx1);
        float v[CMAX]; // the pixel's C normalised class values stay in registers (fully unrolled)
        float sum = bg;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            v[c] = 0.f;
            if (c < C) {
                v[c] = bilerp(src + c * hw, w, y0, y1, ly0, ly1, x0, x1, lx0, lx1) / div[c];
                sum += v[c];
            }
        }
        const float inv = 1.f / sum;
        if (PM) {
            // pixel-major, rows padded to Mp = 4 * ceil((C+1)/4) floats: the layout the mean-field loop reads (no transpose pass)
            const int Mp = (C + 1 + 3) / 4 * 4;
            float uu[CMAX + 4];
            uu[0] = neg_log_clip(bg * inv);
#pragma unroll
            for (int c = 0; c < CMAX + 3; ++c)
                uu[c + 1] = (c < CMAX && c < C) ? neg_log_clip(v[c < CMAX ? c : 0] * inv) : 0.f;
            float4 *dst = reinterpret_cast<float4 *>(unary + ((long long)b * n + i) * Mp);
#pragma unroll
            for (int q = 0; q < (CMAX + 4) / 4; ++q)
                if (4 * q < Mp) dst[q] = make_float4(uu[4 * q], uu[4 * q + 1], uu[4 * q + 2], uu[4 * q + 3]);
        } else {
        float *dst = unary + (long long)b * (C + 1) * n + i;
        dst[0] = neg_log_clip(bg * inv);
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) dst[(long long)(c + 1) * n] = neg_log_clip(v[c] * inv);
        }
    }
}

// eval_cam (03b_irn/step/eval_cam.py:48-62 + chainercv's calc_semantic_segmentation_confusion):
//   cams = pad(high_res, bg channel = thres); cls = pad(keys+1, 0)[argmax(cams, 0)]
//   confusion[gt][cls] += 1 for every pixel whose gt != ignore_label
// Integer work: per-block LDS histogram, one 64-bit global atomic per touched cell.
struct EvalJob {
    long long highres_off; // float offset of this image's [K][H0*W0] block
    long long pix_off;     // offset of this image in the packed gt / pred arrays
    int npix, K, key_base;
};

__global__ __launch_bounds__(256) void cam_eval_kernel(const float *__restrict__ highres,
                                                       const EvalJob *__restrict__ jobs,
                                                       const int32_t *__restrict__ keys, float thres,
                                                       const uint8_t *__restrict__ gt, int n_class, int ignore_label,
                                                       uint8_t *__restrict__ pred,
                                                       unsigned long long *__restrict__ confusion,
                                                       unsigned *__restrict__ n_bad) {
    extern __shared__ unsigned hist[]; // n_class * n_class
    const EvalJob job = jobs[blockIdx.y];
    const int cells = n_class * n_class;
    for (int i = threadIdx.x; i < cells; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < job.npix; p += gridDim.x * blockDim.x) {
        float best = thres;
        int idx = 0;
        for (int k = 0; k < job.K; ++k) {
            const float v = highres[job.highres_off + (long long)k * job.npix + p];
            if (v > best) { // strict: np.argmax keeps the first maximum (the background channel comes first)
                best = v;
                idx = k + 1;
            }
        }
        const int cls = idx == 0 ? 0 : keys[job.key_base + idx - 1] + 1;
        if (pred != nullptr) pred[job.pix_off + p] = (uint8_t)cls;
        const int g = gt != nullptr ? (int)gt[job.pix_off + p] : ignore_label;
        if (g != ignore_label) {
            // chainercv's calc_semantic_segmentation_confusion raises on labels outside [0, n_class): count them, the
            // host turns a non-zero count into an error instead of a plausible but wrong mIoU
            if (g < n_class && cls < n_class) atomicAdd(&hist[g * n_class + cls], 1u);
            else atomicAdd(n_bad, 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cells; i += blockDim.x)
        if (hist[i]) atomicAdd(&confusion[i], (unsigned long long)hist[i]);
}

// ADP / DeepGlobe branch of eval_cam (eval_cam.py:53-63): no background padding, pred = keys[argmax(maps)] taken at the
// maps' own size, then cv2.resize(pred, outsize, INTER_NEAREST) -- evaluated per OUTPUT pixel: source pixel
// (min(floor(Y * inv_y), h - 1), min(floor(X * inv_x), w - 1)) with inv = 1 / (out / src) in double, cv2's rule.
struct EvalNNJob {
    long long maps_off; // float offset of this image's [K][h*w] block
    long long pix_off;  // offset of this image in the packed gt / pred arrays (out_h * out_w each)
    int h, w, out_h, out_w, K, key_base;
    double inv_y, inv_x;
};

// LABELS: the source is an int32 label map (K = 0; `maps` reinterpreted) instead of K class maps -- the evaluation tail of
// 03c_hsn/demo.py:386-408, whose predictions are the CRF's label maps.
template <bool LABELS>
__global__ __launch_bounds__(256) void cam_eval_nn_kernel(const float *__restrict__ maps, const EvalNNJob *__restrict__ jobs,
                                                          const int32_t *__restrict__ keys, const uint8_t *__restrict__ gt,
                                                          int n_class, int ignore_label, uint8_t *__restrict__ pred,
                                                          unsigned long long *__restrict__ confusion,
                                                          unsigned *__restrict__ n_bad) {
    extern __shared__ unsigned hist[]; // n_class * n_class
    const EvalNNJob job = jobs[blockIdx.y];
    const int cells = n_class * n_class;
    for (int i = threadIdx.x; i < cells; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const long long n_out = (long long)job.out_h * job.out_w, n_src = (long long)job.h * job.w;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n_out; p += (long long)gridDim.x * blockDim.x) {
        const int Y = (int)(p / job.out_w), X = (int)(p - (long long)Y * job.out_w);
        const int sy = min((int)floor((double)Y * job.inv_y), job.h - 1);
        const int sx = min((int)floor((double)X * job.inv_x), job.w - 1);
        const long long sp = (long long)sy * job.w + sx;
        int cls;
        if (LABELS) {
            cls = reinterpret_cast<const int32_t *>(maps)[job.maps_off + sp];
        } else {
            float best = maps[job.maps_off + sp];
            int idx = 0;
            for (int k = 1; k < job.K; ++k) {
                const float v = maps[job.maps_off + (long long)k * n_src + sp];
                if (v > best) { // strict: np.argmax keeps the first maximum
                    best = v;
                    idx = k;
                }
            }
            cls = keys[job.key_base + idx];
        }
        if (pred != nullptr) pred[job.pix_off + p] = (uint8_t)cls;
        const int g = gt != nullptr ? (int)gt[job.pix_off + p] : ignore_label;
        if (g != ignore_label) {
            if (g < n_class && (unsigned)cls < (unsigned)n_class) atomicAdd(&hist[g * n_class + cls], 1u);
            else atomicAdd(n_bad, 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cells; i += blockDim.x)
        if (hist[i]) atomicAdd(&confusion[i], (unsigned long long)hist[i]);
}

// make_sem_seg_labels tail (03b_irn/step/make_sem_seg_labels.py:74-79, :91-96, :113-118): the random-walk maps of an image
//   rw_up = F.interpolate(rw, size, 'bilinear', align_corners=False)[..., :H0, :W0];  rw_up /= rw_up.max()
//   [voc12: rw_up = pad(rw_up, bg channel = sem_seg_bg_thres)];  pred = keys[argmax(rw_up, 0)]
// WRITE = false: the image's maximum over all maps and pixels (ordered-uint atomicMax, one per block);
// WRITE = true: the label map.  A zero maximum gives NaN maps in the reference: argmax then picks the first of them.
struct SemSegJob {
    long long rw_off;  // float offset of the image's [K][h*w] maps
    long long pix_off; // offset of the image's labels in the packed output (H0 * W0 each)
    int K, h, w, Hu, Wu, H0, W0, key_base;
};

template <bool WRITE>
__global__ __launch_bounds__(256) void sem_seg_finish_kernel(const float *__restrict__ rw, const SemSegJob *__restrict__ jobs,
                                                             const int32_t *__restrict__ keys, int has_bg, float bg_thres,
                                                             unsigned int *__restrict__ mx, uint8_t *__restrict__ label) {
    const SemSegJob job = jobs[blockIdx.y];
    const int n = job.H0 * job.W0, hw = job.h * job.w;
    const float sh = (float)job.h / (float)job.Hu, sw = (float)job.w / (float)job.Wu;
    const float *src = rw + job.rw_off;
    float m = -3.0e38f, d = 1.f;
    bool degenerate = false;
    if (WRITE) {
        d = ord_dec(mx[blockIdx.y]);
        degenerate = d == 0.f; // 0 / 0 everywhere: NaN maps
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int yy = i / job.W0, xx = i - yy * job.W0;
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        src_index(yy, sh, job.h, y0, y1, ly0, ly1);
        src_index(xx, sw, job.w, x0, x1, lx0, lx1);
        if (!WRITE) {
            for (int k = 0; k < job.K; ++k) m = fmaxf(m, bilerp(src + (long long)k * hw, job.w, y0, y1, ly0, ly1, x0, x1, lx0, lx1));
        } else {
            int idx = has_bg ? 0 : -1;
            float best = has_bg ? bg_thres : -3.0e38f;
            if (degenerate) {
                idx = has_bg ? 1 : 0; // torch.argmax / np.argmax return the first NaN
            } else {
                for (int k = 0; k < job.K; ++k) {
                    const float v = bilerp(src + (long long)k * hw, job.w, y0, y1, ly0, ly1, x0, x1, lx0, lx1) / d;
                    if (idx < 0 || v > best) { // strict: the first maximum wins (the background channel comes first)
                        best = v;
                        idx = k + (has_bg ? 1 : 0);
                    }
                }
            }
            label[job.pix_off + i] = (uint8_t)keys[job.key_base + idx];
        }
    }
    if (!WRITE) {
        __shared__ float wmax[4];
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            if (m > -3.0e38f) atomicMax(&mx[blockIdx.y], ord_enc(m));
        }
    }
}

} // namespace

extern "C" {

int wsc_cam_eval_confusion(wsc_ctx *ctx, const float *highres_dev, int B, const int32_t *size_hw_host,
                           const int32_t *keys_host, const int32_t *key_off_host, const int64_t *highres_off_host,
                           float bg_thres, const uint8_t *gt_dev, int n_class, int ignore_label, uint8_t *pred_dev,
                           int64_t *confusion_dev) {
    WSC_CHECK(ctx && highres_dev && size_hw_host && key_off_host && highres_off_host && confusion_dev, WSC_ERR_INVALID,
              "wsc_cam_eval_confusion: null argument");
    WSC_CHECK(B > 0 && n_class > 0 && n_class <= 64, WSC_ERR_INVALID, "wsc_cam_eval_confusion: B=%d n_class=%d", B,
              n_class);
    WSC_HIP(hipSetDevice(ctx->device));
    std::vector<EvalJob> jobs(B);
    long long pix = 0;
    int max_pix = 0;
    for (int b = 0; b < B; ++b) {
        const int H0 = size_hw_host[2 * b], W0 = size_hw_host[2 * b + 1];
        WSC_CHECK(H0 > 0 && W0 > 0, WSC_ERR_INVALID, "image %d has size %dx%d", b, H0, W0);
        jobs[b].highres_off = highres_off_host[b];
        jobs[b].pix_off = pix;
        jobs[b].npix = H0 * W0;
        jobs[b].K = key_off_host[b + 1] - key_off_host[b];
        jobs[b].key_base = key_off_host[b];
        pix += (long long)H0 * W0;
        max_pix = std::max(max_pix, H0 * W0);
    }
    const int nkeys = key_off_host[B];
    for (int i = 0; i < nkeys; ++i)
        WSC_CHECK(keys_host[i] >= 0 && keys_host[i] + 1 < n_class, WSC_ERR_INVALID,
                  "wsc_cam_eval_confusion: key %d outside [0, %d) (n_class counts the background)", keys_host[i], n_class - 1);
    const size_t jb = (jobs.size() * sizeof(EvalJob) + 15) / 16 * 16,
                 kb = ((size_t)std::max(nkeys, 1) * sizeof(int32_t) + 15) / 16 * 16;
    char *d = nullptr;
    WSC_TRY(wsc_ctx_cached_alloc(ctx, jb + kb + 16, (void **)&d));
    WscCachedGuard d_guard(ctx, d);
    std::vector<char> stage(jb + kb + 16, 0); // the last 16 bytes: out-of-range counter, zeroed
    memcpy(stage.data(), jobs.data(), jobs.size() * sizeof(EvalJob));
    if (nkeys > 0) memcpy(stage.data() + jb, keys_host, (size_t)nkeys * sizeof(int32_t));
    WSC_TRY(wsc_ctx_upload_small(ctx, d, stage.data(), stage.size()));
    const dim3 grid((unsigned)std::min((max_pix + 255) / 256, 64), (unsigned)B);
    hipLaunchKernelGGL(cam_eval_kernel, grid, dim3(256), (size_t)n_class * n_class * sizeof(unsigned), ctx->stream,
                       highres_dev, (const EvalJob *)d, (const int32_t *)(d + jb), bg_thres, gt_dev, n_class,
                       ignore_label, pred_dev, (unsigned long long *)confusion_dev, (unsigned *)(d + jb + kb));
    WSC_HIP(hipGetLastError());
    unsigned n_bad = 0;
    WSC_HIP(hipMemcpyAsync(&n_bad, d + jb + kb, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    WSC_HIP(hipStreamSynchronize(ctx->stream));
    d_guard.free_now();
    WSC_CHECK(n_bad == 0, WSC_ERR_INVALID,
              "wsc_cam_eval_confusion: %u pixels carry a ground-truth label outside [0, %d) (and != ignore_label %d)", n_bad,
              n_class, ignore_label);
    return WSC_OK;
}

int wsc_cam_eval_confusion_nn(wsc_ctx *ctx, const float *maps_dev, int B, const int32_t *src_hw_host,
                              const int32_t *out_hw_host, const int32_t *keys_host, const int32_t *key_off_host,
                              const int64_t *maps_off_host, const uint8_t *gt_dev, int n_class, int ignore_label,
                              uint8_t *pred_dev, int64_t *confusion_dev) {
    WSC_CHECK(ctx && maps_dev && src_hw_host && out_hw_host && keys_host && key_off_host && maps_off_host && confusion_dev,
              WSC_ERR_INVALID, "wsc_cam_eval_confusion_nn: null argument");
    WSC_CHECK(B > 0 && n_class > 0 && n_class <= 64, WSC_ERR_INVALID, "wsc_cam_eval_confusion_nn: B=%d n_class=%d", B, n_class);
    WSC_HIP(hipSetDevice(ctx->device));
    std::vector<EvalNNJob> jobs(B);
    long long pix = 0, max_pix = 0;
    for (int b = 0; b < B; ++b) {
        EvalNNJob &j = jobs[b];
        j.h = src_hw_host[2 * b]; j.w = src_hw_host[2 * b + 1];
        j.out_h = out_hw_host[2 * b]; j.out_w = out_hw_host[2 * b + 1];
        WSC_CHECK(j.h > 0 && j.w > 0 && j.out_h > 0 && j.out_w > 0, WSC_ERR_INVALID, "image %d: %dx%d -> %dx%d", b, j.h, j.w,
                  j.out_h, j.out_w);
        j.K = key_off_host[b + 1] - key_off_host[b];
        WSC_CHECK(j.K >= 1, WSC_ERR_INVALID, "wsc_cam_eval_confusion_nn: image %d has no class map (np.argmax of an empty stack)", b);
        j.key_base = key_off_host[b];
        j.maps_off = maps_off_host[b];
        j.pix_off = pix;
        // cv2.resize: inv_scale = 1. / (dsize / ssize), both in double
        j.inv_y = 1.0 / ((double)j.out_h / (double)j.h);
        j.inv_x = 1.0 / ((double)j.out_w / (double)j.w);
        pix += (long long)j.out_h * j.out_w;
        max_pix = std::max(max_pix, (long long)j.out_h * j.out_w);
    }
    const int nkeys = key_off_host[B];
    for (int i = 0; i < nkeys; ++i)
        WSC_CHECK(keys_host[i] >= 0 && keys_host[i] < n_class, WSC_ERR_INVALID,
                  "wsc_cam_eval_confusion_nn: key %d outside [0, %d)", keys_host[i], n_class);
    const size_t jb = (jobs.size() * sizeof(EvalNNJob) + 15) / 16 * 16, kb = ((size_t)nkeys * sizeof(int32_t) + 15) / 16 * 16;
    char *d = nullptr;
    WSC_TRY(wsc_ctx_cached_alloc(ctx, jb + kb + 16, (void **)&d));
    WscCachedGuard d_guard(ctx, d);
    std::vector<char> stage(jb + kb + 16, 0); // the last 16 bytes: out-of-range counter, zeroed
    memcpy(stage.data(), jobs.data(), jobs.size() * sizeof(EvalNNJob));
    memcpy(stage.data() + jb, keys_host, (size_t)nkeys * sizeof(int32_t));
    WSC_TRY(wsc_ctx_upload_small(ctx, d, stage.data(), stage.size()));
    const dim3 grid((unsigned)std::min<long long>((max_pix + 255) / 256, 1024), (unsigned)B);
    hipLaunchKernelGGL(cam_eval_nn_kernel<false>, grid, dim3(256), (size_t)n_class * n_class * sizeof(unsigned), ctx->stream, maps_dev,
                       (const EvalNNJob *)d, (const int32_t *)(d + jb), gt_dev, n_class, ignore_label, pred_dev,
                       (unsigned long long *)confusion_dev, (unsigned *)(d + jb + kb));
    WSC_HIP(hipGetLastError());
    unsigned n_bad = 0;
    WSC_HIP(hipMemcpyAsync(&n_bad, d + jb + kb, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    WSC_HIP(hipStreamSynchronize(ctx->stream));
    d_guard.free_now();
    WSC_CHECK(n_bad == 0, WSC_ERR_INVALID,
              "wsc_cam_eval_confusion_nn: %u pixels carry a ground-truth label outside [0, %d) (and != ignore_label %d)", n_bad,
              n_class, ignore_label);
    return WSC_OK;
}

int wsc_label_confusion_nn(wsc_ctx *ctx, const int32_t *labels_dev, int B, const int32_t *src_hw_host, const int32_t *out_hw_host,
                           const int64_t *labels_off_host, const uint8_t *gt_dev, int n_class, int ignore_label, uint8_t *pred_dev,
                           int64_t *confusion_dev) {
    WSC_CHECK(ctx && labels_dev && src_hw_host && out_hw_host && labels_off_host && confusion_dev, WSC_ERR_INVALID,
              "wsc_label_confusion_nn: null argument");
    WSC_CHECK(B > 0 && n_class > 0 && n_class <= 64, WSC_ERR_INVALID, "wsc_label_confusion_nn: B=%d n_class=%d", B, n_class);
    WSC_HIP(hipSetDevice(ctx->device));
    std::vector<EvalNNJob> jobs(B);
    long long pix = 0, max_pix = 0;
    for (int b = 0; b < B; ++b) {
        EvalNNJob &j = jobs[b];
        j.h = src_hw_host[2 * b]; j.w = src_hw_host[2 * b + 1];
        j.out_h = out_hw_host[2 * b]; j.out_w = out_hw_host[2 * b + 1];
        WSC_CHECK(j.h > 0 && j.w > 0 && j.out_h > 0 && j.out_w > 0, WSC_ERR_INVALID, "image %d: %dx%d -> %dx%d", b, j.h, j.w,
                  j.out_h, j.out_w);
        j.K = 0;
        j.key_base = 0;
        j.maps_off = labels_off_host[b];
        j.pix_off = pix;
        j.inv_y = 1.0 / ((double)j.out_h / (double)j.h); // cv2.resize: inv_scale = 1. / (dsize / ssize), both in double
        j.inv_x = 1.0 / ((double)j.out_w / (double)j.w);
        pix += (long long)j.out_h * j.out_w;
        max_pix = std::max(max_pix, (long long)j.out_h * j.out_w);
    }
    const size_t jb = (jobs.size() * sizeof(EvalNNJob) + 15) / 16 * 16;
    char *d = nullptr;
    WSC_TRY(wsc_ctx_cached_alloc(ctx, jb + 16, (void **)&d));
    WscCachedGuard d_guard(ctx, d);
    std::vector<char> stage(jb + 16, 0); // the last 16 bytes: out-of-range counter, zeroed
    memcpy(stage.data(), jobs.data(), jobs.size() * sizeof(EvalNNJob));
    WSC_TRY(wsc_ctx_upload_small(ctx, d, stage.data(), stage.size()));
    const dim3 grid((unsigned)std::min<long long>((max_pix + 255) / 256, 1024), (unsigned)B);
    hipLaunchKernelGGL(cam_eval_nn_kernel<true>, grid, dim3(256), (size_t)n_class * n_class * sizeof(unsigned), ctx->stream,
                       reinterpret_cast<const float *>(labels_dev), (const EvalNNJob *)d, (const int32_t *)nullptr, gt_dev, n_class,
                       ignore_label, pred_dev, (unsigned long long *)confusion_dev, (unsigned *)(d + jb));
    WSC_HIP(hipGetLastError());
    unsigned n_bad = 0;
    WSC_HIP(hipMemcpyAsync(&n_bad, d + jb, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    WSC_HIP(hipStreamSynchronize(ctx->stream));
    d_guard.free_now();
    WSC_CHECK(n_bad == 0, WSC_ERR_INVALID,
              "wsc_label_confusion_nn: %u pixels carry a label or a ground-truth label outside [0, %d) (and != ignore_label %d)", n_bad,
              n_class, ignore_label);
    return WSC_OK;
}

int wsc_sem_seg_finish(wsc_ctx *ctx, const float *rw_dev, int B, const int64_t *rw_off_host, const int32_t *khw_host,
                       const int32_t *up_hw_host, const int32_t *out_hw_host, const int32_t *keys_host, const int32_t *key_off_host,
                       int has_bg, float bg_thres, uint8_t *label_dev) {
    WSC_CHECK(ctx && rw_dev && rw_off_host && khw_host && up_hw_host && out_hw_host && keys_host && key_off_host && label_dev,
              WSC_ERR_INVALID, "wsc_sem_seg_finish: null argument");
    WSC_CHECK(B > 0 && B <= 65535, WSC_ERR_INVALID, "wsc_sem_seg_finish: B=%d", B);
    WSC_HIP(hipSetDevice(ctx->device));
    std::vector<SemSegJob> jobs(B);
    long long pix = 0;
    int max_pix = 0;
    for (int b = 0; b < B; ++b) {
        SemSegJob &j = jobs[b];
        j.K = khw_host[3 * b]; j.h = khw_host[3 * b + 1]; j.w = khw_host[3 * b + 2];
        j.Hu = up_hw_host[2 * b]; j.Wu = up_hw_host[2 * b + 1];
        j.H0 = out_hw_host[2 * b]; j.W0 = out_hw_host[2 * b + 1];
        WSC_CHECK(j.K >= 1 && j.h > 0 && j.w > 0 && j.Hu >= j.H0 && j.Wu >= j.W0 && j.H0 > 0 && j.W0 > 0, WSC_ERR_INVALID,
                  "wsc_sem_seg_finish: image %d: K=%d %dx%d -> %dx%d crop %dx%d", b, j.K, j.h, j.w, j.Hu, j.Wu, j.H0, j.W0);
        WSC_CHECK(key_off_host[b + 1] - key_off_host[b] == j.K + (has_bg ? 1 : 0), WSC_ERR_INVALID,
                  "wsc_sem_seg_finish: image %d has %d keys for %d maps%s", b, key_off_host[b + 1] - key_off_host[b], j.K,
                  has_bg ? " + background" : "");
        j.key_base = key_off_host[b];
        j.rw_off = rw_off_host[b];
        j.pix_off = pix;
        pix += (long long)j.H0 * j.W0;
        max_pix = std::max(max_pix, j.H0 * j.W0);
    }
    const int nkeys = key_off_host[B];
    for (int i = 0; i < nkeys; ++i)
        WSC_CHECK(keys_host[i] >= 0 && keys_host[i] <= 255, WSC_ERR_INVALID, "wsc_sem_seg_finish: key %d does not fit a uint8 label", keys_host[i]);
    const size_t jb = (jobs.size() * sizeof(SemSegJob) + 15) / 16 * 16, kb = ((size_t)nkeys * sizeof(int32_t) + 15) / 16 * 16,
                 mb = ((size_t)B * sizeof(unsigned) + 15) / 16 * 16;
    char *d = nullptr;
    WSC_TRY(wsc_ctx_cached_alloc(ctx, jb + kb + mb, (void **)&d));
    WscCachedGuard d_guard(ctx, d);
    std::vector<char> stage(jb + kb + mb, 0); // the maxima start at code 0 = "below every real number"
    memcpy(stage.data(), jobs.data(), jobs.size() * sizeof(SemSegJob));
    memcpy(stage.data() + jb, keys_host, (size_t)nkeys * sizeof(int32_t));
    WSC_TRY(wsc_ctx_upload_small(ctx, d, stage.data(), stage.size()));
    const dim3 grid((unsigned)std::min((max_pix + 255) / 256, 256), (unsigned)B);
    WscKernelTimer timer(ctx, WSC_K_CAM_TAIL, (double)pix * 4);
    hipLaunchKernelGGL(sem_seg_finish_kernel<false>, grid, dim3(256), 0, ctx->stream, rw_dev, (const SemSegJob *)d,
                       (const int32_t *)(d + jb), has_bg, bg_thres, (unsigned int *)(d + jb + kb), label_dev);
    hipLaunchKernelGGL(sem_seg_finish_kernel<true>, grid, dim3(256), 0, ctx->stream, rw_dev, (const SemSegJob *)d,
                       (const int32_t *)(d + jb), has_bg, bg_thres, (unsigned int *)(d + jb + kb), label_dev);
    WSC_HIP(hipGetLastError());
    d_guard.free_now(); // stream-ordered reuse
    return WSC_OK;
}

int wsc_unary_from_maps(wsc_ctx *ctx, const float *maps_dev, int B, int C, int N, float bg_value, float *unary_dev) {
    WSC_CHECK(ctx && maps_dev && unary_dev, WSC_ERR_INVALID, "wsc_unary_from_maps: null argument");
    WSC_CHECK(B > 0 && C > 0 && N > 0 && bg_value > 0.f, WSC_ERR_INVALID, "wsc_unary_from_maps: bad argument");
    WSC_HIP(hipSetDevice(ctx->device));
    const long long total = (long long)B * N;
    long long g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    WscKernelTimer timer(ctx, WSC_K_CAM_TAIL, (double)B * N * (2.0 * C + 1) * 4);
    hipLaunchKernelGGL(unary_from_maps_kernel, dim3((unsigned)g), dim3(256), 0, ctx->stream, maps_dev, bg_value, C, N,
                       total, unary_dev);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

static int cam_unary_impl(wsc_ctx *ctx, const float *cam_dev, int B, int C, int h, int w, int H0, int W0, float bg_value,
                          float *unary_dev, bool pixel_major);

// out[b] = sum over the n_scales consecutive maps of image b, added in scale order (fp32)
__global__ void cam_sum_scales_kernel(const float *__restrict__ cam, long long n_out, int n_scales, long long elems,
                                      float *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / elems, e = i - b * elems;
        const float *src = cam + (b * n_scales) * elems + e;
        float acc = src[0];
        for (int sidx = 1; sidx < n_scales; ++sidx) acc += src[(long long)sidx * elems];
        out[i] = acc;
    }
}

int wsc_cam_sum_scales(wsc_ctx *ctx, const float *cam_dev, int n_images, int n_scales, long long map_elems, float *out_dev) {
    WSC_CHECK(ctx && cam_dev && out_dev, WSC_ERR_INVALID, "wsc_cam_sum_scales: null argument");
    WSC_CHECK(n_images > 0 && n_scales > 0 && map_elems > 0, WSC_ERR_INVALID, "wsc_cam_sum_scales: bad argument");
    WSC_CHECK(cam_dev != out_dev || n_scales == 1, WSC_ERR_INVALID, "wsc_cam_sum_scales: in-place call with several scales");
    WSC_HIP(hipSetDevice(ctx->device));
    const long long n_out = (long long)n_images * map_elems;
    long long g = (n_out + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(cam_sum_scales_kernel, dim3((unsigned)g), dim3(256), 0, ctx->stream, cam_dev, n_out, n_scales, map_elems,
                       out_dev);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

int wsc_cam_unary(wsc_ctx *ctx, const float *cam_dev, int B, int C, int h, int w, int H0, int W0, float bg_value,
                  float *unary_dev) {
    return cam_unary_impl(ctx, cam_dev, B, C, h, w, H0, W0, bg_value, unary_dev, false);
}

int wsc_cam_unary_pm(wsc_ctx *ctx, const float *cam_dev, int B, int C, int h, int w, int H0, int W0, float bg_value,
                     float *unary_pm_dev) {
    return cam_unary_impl(ctx, cam_dev, B, C, h, w, H0, W0, bg_value, unary_pm_dev, true);
}

static int cam_unary_impl(wsc_ctx *ctx, const float *cam_dev, int B, int C, int h, int w, int H0, int W0, float bg_value,
                          float *unary_dev, bool pixel_major) {
    WSC_CHECK(ctx && cam_dev && unary_dev, WSC_ERR_INVALID, "wsc_cam_unary: null argument");
    WSC_CHECK(B > 0 && C > 0 && h > 0 && w > 0 && H0 > 0 && W0 > 0 && bg_value > 0.f, WSC_ERR_INVALID,
              "wsc_cam_unary: bad argument");
    WSC_CHECK((long long)H0 * W0 < (1ll << 31) && B <= 65535 && (long long)B * C <= 65535, WSC_ERR_INVALID,
              "wsc_cam_unary: batch too large for one call");
    WSC_CHECK(C <= 32, WSC_ERR_INVALID, "wsc_cam_unary: C=%d > 32 classes (use wsc_cam_postprocess + wsc_unary_from_maps)", C);
    const size_t lds = ((size_t)C * h * w + C) * sizeof(float);
    WSC_CHECK(lds <= 64 * 1024, WSC_ERR_INVALID, "wsc_cam_unary: %d maps of %dx%d do not fit the 64 KB LDS tile", C, h, w);
    WSC_HIP(hipSetDevice(ctx->device));
    // misc.imutils.get_strided_up_size(size, 16)
    const int Hu = ((H0 - 1) / 16 + 1) * 16, Wu = ((W0 - 1) / 16 + 1) * 16;
    unsigned int *mx = nullptr;
    WSC_TRY(wsc_ctx_cached_alloc(ctx, sizeof(unsigned int) * (size_t)B * C, (void **)&mx));
    WscCachedGuard mx_guard(ctx, mx);
    WSC_HIP(hipMemsetAsync(mx, 0, sizeof(unsigned int) * (size_t)B * C, ctx->stream));
    const int n = H0 * W0;
    WscKernelTimer timer(ctx, WSC_K_CAM_TAIL, (double)B * (C + 1) * n * 4);
    hipLaunchKernelGGL(cam_max_kernel, dim3((unsigned)(B * C)), dim3((unsigned)std::min(1024, (W0 + 63) / 64 * 64)),
                       ((size_t)h * w + 20 + 4 * (size_t)H0) * sizeof(float), ctx->stream, cam_dev, C, h, w, H0, W0, Hu, Wu, mx);
    // each block re-reads its image's C source maps (35 KB for 20 x 21 x 21): a few pixels per thread amortise that
    const dim3 ugrid((unsigned)std::min((n + 2047) / 2048, 512), (unsigned)B);
    if (C <= 20 && !pixel_major)
        hipLaunchKernelGGL((cam_unary_kernel<20, false>), ugrid, dim3(256), lds, ctx->stream, cam_dev, C, h, w, H0, W0, Hu, Wu,
                           (const unsigned int *)mx, bg_value, unary_dev);
    else if (C <= 20)
        hipLaunchKernelGGL((cam_unary_kernel<20, true>), ugrid, dim3(256), lds, ctx->stream, cam_dev, C, h, w, H0, W0, Hu, Wu,
                           (const unsigned int *)mx, bg_value, unary_dev);
    else if (!pixel_major)
        hipLaunchKernelGGL((cam_unary_kernel<32, false>), ugrid, dim3(256), lds, ctx->stream, cam_dev, C, h, w, H0, W0, Hu, Wu,
                           (const unsigned int *)mx, bg_value, unary_dev);
    else
        hipLaunchKernelGGL((cam_unary_kernel<32, true>), ugrid, dim3(256), lds, ctx->stream, cam_dev, C, h, w, H0, W0, Hu, Wu,
                           (const unsigned int *)mx, bg_value, unary_dev);
    WSC_HIP(hipGetLastError());
    mx_guard.free_now(); // stream-ordered reuse
    return WSC_OK;
}

int wsc_cam_postprocess(wsc_ctx *ctx, const float *cam_dev, int B, int C, int h, int w,
                        const int32_t *size_hw_host, const int32_t *keys_host, const int32_t *key_off_host,
                        const int64_t *strided_off_host, const int64_t *highres_off_host, float *strided_dev,
                        float *highres_dev) {
    WSC_CHECK(ctx && cam_dev && size_hw_host && key_off_host && strided_off_host && highres_off_host,
              WSC_ERR_INVALID, "wsc_cam_postprocess: null argument");
    WSC_CHECK(B > 0 && C > 0 && h > 0 && w > 0, WSC_ERR_INVALID, "wsc_cam_postprocess: bad shape");
    WSC_CHECK((size_t)h * w * sizeof(float) <= 64 * 1024, WSC_ERR_INVALID, "CAM %dx%d too large for the LDS tile", h, w);
    WSC_HIP(hipSetDevice(ctx->device));
    std::vector<TailJob> jobs;
    int max_pix = 0;
    for (int b = 0; b < B; ++b) {
        const int H0 = size_hw_host[2 * b], W0 = size_hw_host[2 * b + 1];
        WSC_CHECK(H0 > 0 && W0 > 0, WSC_ERR_INVALID, "image %d has size %dx%d", b, H0, W0);
        // misc.imutils.get_strided_size(size, 4) / get_strided_up_size(size, 16)
        const int h4 = (H0 - 1) / 4 + 1, w4 = (W0 - 1) / 4 + 1;
        const int Hu = ((H0 - 1) / 16 + 1) * 16, Wu = ((W0 - 1) / 16 + 1) * 16;
        const int K = key_off_host[b + 1] - key_off_host[b];
        WSC_CHECK(K >= 0, WSC_ERR_INVALID, "key_off must be non-decreasing");
        WSC_CHECK((long long)H0 * W0 + (long long)h4 * w4 < (1ll << 31), WSC_ERR_INVALID, "image %d too large", b);
        for (int j = 0; j < K; ++j) {
            const int key = keys_host[key_off_host[b] + j];
            WSC_CHECK(key >= 0 && key < C, WSC_ERR_INVALID, "image %d: class key %d outside [0,%d)", b, key, C);
            TailJob t;
            t.cam_off = ((long long)b * C + key) * h * w;
            t.strided_off = strided_off_host[b] + (long long)j * h4 * w4;
            t.highres_off = highres_off_host[b] + (long long)j * H0 * W0;
            t.H0 = H0; t.W0 = W0; t.h4 = h4; t.w4 = w4; t.Hu = Hu; t.Wu = Wu;
            jobs.push_back(t);
            max_pix = std::max(max_pix, H0 * W0 + h4 * w4);
        }
    }
    if (jobs.empty()) return WSC_OK;
    WSC_CHECK(strided_dev && highres_dev, WSC_ERR_INVALID, "wsc_cam_postprocess: null output");
    WSC_CHECK(jobs.size() <= 65535, WSC_ERR_INVALID, "too many (image, class) jobs in one call: %zu", jobs.size());
    // descriptor + maxima live in a small block of the ctx's stream-ordered cache
    const size_t jb = jobs.size() * sizeof(TailJob), mb = jobs.size() * 2 * sizeof(unsigned int);
    char *d = nullptr;
    WSC_TRY(wsc_ctx_cached_alloc(ctx, jb + mb, (void **)&d));
    WscCachedGuard d_guard(ctx, d);
    WSC_TRY(wsc_ctx_upload_small(ctx, d, jobs.data(), jb)); // through pinned staging: no host sync
    WSC_HIP(hipMemsetAsync(d + jb, 0, mb, ctx->stream));
    const dim3 grid((max_pix + PIX_PER_BLOCK - 1) / PIX_PER_BLOCK, (unsigned)jobs.size());
    const size_t lds = (size_t)h * w * sizeof(float);
    double out_bytes = 0;
    for (const TailJob &t : jobs) out_bytes += ((double)t.H0 * t.W0 + (double)t.h4 * t.w4) * 4;
    WscKernelTimer timer(ctx, WSC_K_CAM_TAIL, out_bytes);
    // pass 1, the maxima: column form (one block per job) while its row-tap table fits the LDS, else the per-pixel form
    int max_rows = 1;
    for (const TailJob &t : jobs) max_rows = std::max(max_rows, std::max(t.H0, t.h4));
    const int bands = 8; // row bands per job: a few hundred blocks for the ~50 jobs of a 32-image batch
    const size_t lds_max = (((size_t)h * w + 16 + 3) & ~(size_t)3) * sizeof(float) +
                           (size_t)((max_rows + bands - 1) / bands) * sizeof(float4);
    if (lds_max <= 64 * 1024)
        hipLaunchKernelGGL(cam_tail_max_kernel, dim3((unsigned)jobs.size(), (unsigned)bands), dim3(512), lds_max, ctx->stream,
                           cam_dev, (const TailJob *)d, h, w, (unsigned int *)(d + jb));
    else
        hipLaunchKernelGGL(cam_tail_kernel<false>, grid, dim3(256), lds, ctx->stream, cam_dev, (const TailJob *)d, h, w,
                           (unsigned int *)(d + jb), strided_dev, highres_dev);
    hipLaunchKernelGGL(cam_tail_kernel<true>, grid, dim3(256), lds, ctx->stream, cam_dev, (const TailJob *)d, h, w,
                       (unsigned int *)(d + jb), strided_dev, highres_dev);
    WSC_HIP(hipGetLastError());
    d_guard.free_now(); // stream-ordered reuse
    return WSC_OK;
}

int wsc_bilinear_resize(wsc_ctx *ctx, const float *src_dev, int C, int h, int w, float *dst_dev, int H, int W) {
    WSC_CHECK(ctx && src_dev && dst_dev, WSC_ERR_INVALID, "wsc_bilinear_resize: null argument");
    WSC_CHECK(C > 0 && h > 0 && w > 0 && H > 0 && W > 0, WSC_ERR_INVALID, "wsc_bilinear_resize: bad shape");
    WSC_HIP(hipSetDevice(ctx->device));
    const long long total = (long long)C * H * W;
    long long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(bilinear_kernel, dim3((unsigned)g), dim3(256), 0, ctx->stream, src_dev, C, h, w, dst_dev, H, W);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

} // extern "C"
