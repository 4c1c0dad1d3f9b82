// hsn.hip -- HistoSegNet post-processing between the Grad-CAM head and the dense CRF, on the device
// (reference: 03c_hsn/utilities.py:231-397, driven by 03c_hsn/demo.py:318-380; twins in 02_cues/adp_cues.py:244-302
// and 03b_irn/net/common_cam.py:31-92).  The reference does all of this in numpy / scipy / cv2 on the host, image by
// image; here the Grad-CAM stack never leaves HBM between wsc_net_forward_gradcam and wsc_crf_inference:
//   wsc_hsn_gradcam_post   per (image, class) bilinear upsample + max(., 0), division by the per-image maximum over
//                          all classes, gating by score x pass                                  (utilities.py:262-277)
//   wsc_hsn_background     0.75 * expit(4 * (mean_rgb - 240)), separable Gaussian sigma = 2 (scipy.ndimage.
//                          gaussian_filter: truncate 4 sigma, mode 'reflect', axis 0 then axis 1)  (:341-347)
//   wsc_hsn_cs_gradcam     valid-class stack, Background / Other channels (modify_by_htt :348-363), class-specific
//                          Grad-CAM = top1 - top2 margin on the arg-max class (get_cs_gradcam :367-397), per
//                          (image, class) "has positive mass" flags (dcrf_process :425)
//   wsc_hsn_gather_unary   U = -log(clip(p, 1e-5, 1)) of each image's passing classes (unary_from_softmax, :431)
// The maps are fp32 (the reference carries float64 arrays of float32-born values); the background activation and the
// per-pixel class comparisons are float64 (see hsn_bg_sigmoid_kernel).  HBM-bound elementwise / stencil kernels.
#include "common.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

__device__ __forceinline__ void src_index_hp(int dst, float scale, int n, int &i0, int &i1, float &l0, float &l1) {
    // half-pixel centres with clamping: cv2.resize INTER_LINEAR == F.interpolate(bilinear, align_corners=False)
    float s = ((float)dst + 0.5f) * scale - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > n - 1) i0 = n - 1;
    i1 = i0 < n - 1 ? i0 + 1 : i0;
    l1 = s - (float)i0;
    l0 = 1.f - l1;
}

// One block per (image, class, band of output rows): the class's h x w source map (NHWC-strided in memory) is staged in LDS,
// a thread owns an output column and walks down the band.  The two source rows of an output row change every S / h (= 8)
// output rows: the column's horizontal taps are recomputed only then, and nothing is divided per sample (the per-pixel
// form spent its time on one division and two tap computations per sample: 337 + 216 us for 16 x 31 maps at 321 x 321).
// The bilinear sample is spelled with explicit fused multiply-adds so that the max pass and the write pass -- two kernels --
// compute the same bits (the normalised maximum is then exactly what the division makes of it).
template <bool WRITE>
__global__ __launch_bounds__(256) void hsn_gradcam_post_kernel(const float *__restrict__ cams, int h, int w, int C, int S,
                                                               const float *__restrict__ gate, unsigned int *__restrict__ mx,
                                                               float *__restrict__ out, int Ctot, int c0, int band) {
    extern __shared__ float src[]; // h*w
    const int bc = blockIdx.y, b = bc / C, c = bc - b * C;
    const float *base = cams + (long long)b * h * w * C + c;
    for (int i = threadIdx.x; i < h * w; i += blockDim.x) src[i] = base[(long long)i * C];
    __syncthreads();
    const float sh = (float)h / (float)S, sw = (float)w / (float)S;
    const int n = S * S;
    const int yb = blockIdx.x * band, ye = min(S, yb + band);
    float m = 0.f, scale = 0.f;
    if (WRITE) scale = gate[bc] / fmaxf(__uint_as_float(mx[b]), 1e-7f);
    float *dst = WRITE ? out + ((long long)b * Ctot + c0 + c) * n : nullptr;
    for (int xx = threadIdx.x; xx < S; xx += blockDim.x) {
        int x0, x1;
        float lx0, lx1;
        src_index_hp(xx, sw, w, x0, x1, lx0, lx1);
        int o0 = -1, o1 = -1;
        float top = 0.f, bot = 0.f;
        for (int yy = yb; yy < ye; ++yy) {
            int y0, y1;
            float ly0, ly1;
            src_index_hp(yy, sh, h, y0, y1, ly0, ly1); // (uniform over the block)
            if (y0 != o0 || y1 != o1) {
                top = __builtin_fmaf(lx0, src[y0 * w + x0], lx1 * src[y0 * w + x1]);
                bot = __builtin_fmaf(lx0, src[y1 * w + x0], lx1 * src[y1 * w + x1]);
                o0 = y0;
                o1 = y1;
            }
            const float v = fmaxf(__builtin_fmaf(ly0, top, ly1 * bot), 0.f);
            if (WRITE) dst[yy * S + xx] = v * scale;
            else m = fmaxf(m, v);
        }
    }
    if (!WRITE) {
        // one atomic per BLOCK, and only when it can raise the image's maximum: 127 k waves hammering 16 addresses made this
        // pass 1.2 ms
        __shared__ float wmax[4];
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            // values >= 0: uint order = float order.  (The plain read may be stale; the atomic decides.)
            if (m > 0.f && __float_as_uint(m) > mx[b]) atomicMax(&mx[b], __float_as_uint(m));
        }
    }
}

// bg0 = 0.75 * expit(4 * (mean_rgb - 240))
// (float64 like the reference: on tissue the activation is 1e-100 ... 1e-40 -- zero in fp32 -- yet it still beats an
// all-zero class stack, and dcrf_process keeps the Background class exactly when that happens somewhere, :425)
__global__ void hsn_bg_sigmoid_kernel(const uint8_t *__restrict__ rgb, long long total, double *__restrict__ bg) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const uint8_t *p = rgb + i * 3;
        const double mean = (double)((int)p[0] + (int)p[1] + (int)p[2]) / 3.0;
        bg[i] = 0.75 / (1.0 + exp(-4.0 * (mean - 240.0)));
    }
}

struct GaussTaps {
    double w[17]; // radius 8 = int(4 * 2 + 0.5)
};

// one 1-D pass of scipy.ndimage.gaussian_filter1d(sigma = 2), mode 'reflect' (d c b a | a b c d | d c b a)
__global__ void hsn_gauss1d_kernel(const double *__restrict__ in, int H, int W, int axis, long long total, GaussTaps t,
                                   double *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long img = i / ((long long)H * W);
        const int r = (int)(i - img * H * W);
        const int y = r / W, x = r - y * W;
        const double *src = in + img * H * W;
        const int n = axis == 0 ? H : W;
        const int pos = axis == 0 ? y : x;
        double acc = 0.0;
#pragma unroll
        for (int k = -8; k <= 8; ++k) {
            int q = pos + k;
            // reflect about the half-sample edges, repeatedly for tiny images
            while (q < 0 || q >= n) q = q < 0 ? -q - 1 : 2 * n - 1 - q;
            acc += t.w[k + 8] * (axis == 0 ? src[q * W + x] : src[y * W + q]);
        }
        out[i] = acc;
    }
}

// cv2.resize(bg, (Wo, Ho)) of the float64 activation (modify_by_htt :345-347 when the image is larger than the CAM grid)
__global__ void hsn_resize_f64_kernel(const double *__restrict__ in, int H, int W, int Ho, int Wo, long long total,
                                      double *__restrict__ out) {
    const double sh = (double)H / (double)Ho, sw = (double)W / (double)Wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long img = i / ((long long)Ho * Wo);
        const int r = (int)(i - img * Ho * Wo);
        const int yy = r / Wo, xx = r - yy * Wo;
        double fy = ((double)yy + 0.5) * sh - 0.5, fx = ((double)xx + 0.5) * sw - 0.5;
        fy = fy < 0.0 ? 0.0 : (fy > (double)(H - 1) ? (double)(H - 1) : fy);
        fx = fx < 0.0 ? 0.0 : (fx > (double)(W - 1) ? (double)(W - 1) : fx);
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 < H - 1 ? y0 + 1 : y0, x1 = x0 < W - 1 ? x0 + 1 : x0;
        const double wy = fy - (double)y0, wx = fx - (double)x0;
        const double *src = in + img * H * W;
        out[i] = (src[y0 * W + x0] * (1.0 - wx) + src[y0 * W + x1] * wx) * (1.0 - wy) +
                 (src[y1 * W + x0] * (1.0 - wx) + src[y1 * W + x1] * wx) * wy;
    }
}

// ---- ADP background / 'other' channels of the 03b_irn CAM networks (net/common_cam.py:31-92) ---------------------------------
struct AdpModArgs {
    const float *cam;  // [B][n_sc][C][hw]
    const double *bg;  // [B][n_sc][hw] smoothed + resized background activation (wsc_hsn_background)
    float *out;        // [B][n_use + 1 + mode][hw]
    int n_sc, C, hw, mode, n_use, n_adip, n_exc;
    int use[64], adip[4], exc[4];
};
// One thread per (image, pixel); the scales of an image are added in scale order (make_cam.py:62-69 sums the modified stacks).
// fp32 like the reference's torch tensors (the float64 activation is rounded once, where torch.from_numpy meets the fp32 CAM).
__global__ __launch_bounds__(256) void cam_adp_modify_kernel(AdpModArgs a) {
    const int b = blockIdx.y;
    const int Cout = a.n_use + 1 + a.mode;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < a.hw; p += gridDim.x * blockDim.x) {
        float o_bg = 0.f, o_other = 0.f;
        float *ob = a.out + (long long)b * Cout * a.hw + p;
        for (int s = 0; s < a.n_sc; ++s) {
            const float *cb = a.cam + ((long long)b * a.n_sc + s) * a.C * a.hw + p;
            const float bgv = (float)a.bg[((long long)b * a.n_sc + s) * a.hw + p];
            float adip = -3.0e38f;
            for (int k = 0; k < a.n_adip; ++k) adip = fmaxf(adip, cb[(long long)a.adip[k] * a.hw]);
            if (a.mode == 0) { // _adp_modify_morph :31-55: relu(bg - max adipose)
                o_bg += fmaxf(bgv - adip, 0.f);
            } else {           // _adp_modify_func :57-92: bg - max exception (no relu), then the 'other' channel
                float ex = -3.0e38f;
                for (int k = 0; k < a.n_exc; ++k) ex = fmaxf(ex, cb[(long long)a.exc[k] * a.hw]);
                const float b0 = bgv - ex;
                float moh = b0;
                for (int i = 0; i < a.n_use; ++i) moh = fmaxf(moh, cb[(long long)a.use[i] * a.hw]);
                o_bg += b0;
                o_other += fmaxf(0.05f * (1.f - moh), adip);
            }
        }
        ob[0] = o_bg;
        if (a.mode) ob[a.hw] = o_other;
        for (int i = 0; i < a.n_use; ++i) {
            float v = 0.f;
            for (int s = 0; s < a.n_sc; ++s) v += a.cam[(((long long)b * a.n_sc + s) * a.C + a.use[i]) * a.hw + p];
            ob[(long long)(1 + a.mode + i) * a.hw] = v;
        }
    }
}

struct CsArgs {
    const float *H;      // [B][C_all][N] gated Grad-CAMs
    const double *bg;    // [B][N] smoothed background activation (float64)
    float *cs;           // [B][Cv][N]
    float *y;            // [B][Cv][N] modified stack (or null)
    unsigned int *mass;  // [B][Cv] 1 if the class has positive mass in the image
    int C_all, Cv, N;
    int bg_ind, other_ind; // other_ind < 0: morphological types
    int src_of[32];      // valid class -> channel of H (-1: synthesised)
    int exc[4], n_exc;   // background exception classes (valid-class indices)
    int adip[4], n_adip; // adipose channels of H (functional types)
};

// per pixel: Y = valid-class stack; Y[bg] = bg - max Y[exc]; functional: Y[other] = max(0.05 (1 - max_c Y), max adipose);
// cs[c] = (top1 - top2) [argmax == c], Other passes through
__global__ __launch_bounds__(256) void hsn_cs_kernel(CsArgs a) {
    const int b = blockIdx.y;
    const int Cv = a.Cv;
    __shared__ unsigned int flags[32];
    if (threadIdx.x < 32) flags[threadIdx.x] = 0;
    __syncthreads();
    const float *Hb = a.H + (long long)b * a.C_all * a.N;
    unsigned int pos_mask = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < a.N; p += gridDim.x * blockDim.x) {
        // float64 per pixel like the reference's arrays: the comparisons that decide arg-max, margin and "positive mass"
        // involve background values far below the fp32 range
        double y[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            y[c] = 0.0;
            if (c < Cv && a.src_of[c] >= 0) y[c] = (double)Hb[(long long)a.src_of[c] * a.N + p];
        }
        if (a.bg != nullptr) {
            double ex = -1.0e300;
#pragma unroll
            for (int c = 0; c < 32; ++c)
                if (c < Cv) {
                    bool is_exc = false;
                    for (int k = 0; k < a.n_exc; ++k) is_exc = is_exc || a.exc[k] == c;
                    if (is_exc) ex = fmax(ex, y[c]);
                }
            const double bgv = a.bg[(long long)b * a.N + p] - ex;
#pragma unroll
            for (int c = 0; c < 32; ++c)
                if (c == a.bg_ind) y[c] = bgv;
            if (a.other_ind >= 0) {
                double moh = -1.0e300;
#pragma unroll
                for (int c = 0; c < 32; ++c)
                    if (c < Cv) moh = fmax(moh, y[c]);
                double other = 0.05 * (1.0 - moh);
                for (int k = 0; k < a.n_adip; ++k) other = fmax(other, (double)Hb[(long long)a.adip[k] * a.N + p]);
#pragma unroll
                for (int c = 0; c < 32; ++c)
                    if (c == a.other_ind) y[c] = other;
            }
        }
        // top-2 with np.argmax's tie rule (first maximum)
        double t1 = -1.0e300, t2 = -1.0e300;
        int am = 0;
#pragma unroll
        for (int c = 0; c < 32; ++c)
            if (c < Cv) {
                const double v = y[c];
                if (v > t1) {
                    t2 = t1;
                    t1 = v;
                    am = c;
                } else if (v > t2) {
                    t2 = v;
                }
            }
        const double diff = Cv > 1 ? t1 - t2 : 0.0;
#pragma unroll
        for (int c = 0; c < 32; ++c)
            if (c < Cv) {
                const double v = c == a.other_ind ? y[c] : (c == am ? diff : 0.0);
                a.cs[((long long)b * Cv + c) * a.N + p] = (float)v; // the CRF clips to [1e-5, 1]: fp32 is enough from here on
                if (a.y) a.y[((long long)b * Cv + c) * a.N + p] = (float)y[c];
                if (v > 0.0) pos_mask |= 1u << c;
            }
    }
    for (int c = 0; c < Cv; ++c)
        if (pos_mask >> c & 1u) flags[c] = 1u; // benign race: every writer stores 1
    __syncthreads();
    if ((int)threadIdx.x < Cv && flags[threadIdx.x]) a.mass[b * Cv + threadIdx.x] = 1u;
}

// U[g][m][p] = -log(clip(src[chan(g, m)][p], 1e-5, 1)); chan = float offset / N into `maps`
__global__ void hsn_gather_unary_kernel(const float *__restrict__ maps, const long long *__restrict__ chan_off, int N,
                                        long long total, float *__restrict__ unary) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long gm = i / N;
        const int p = (int)(i - gm * N);
        const float v = maps[chan_off[gm] + p];
        unary[i] = -logf(fminf(fmaxf(v, 1e-5f), 1.f));
    }
}

// VOC background channel of 03c_hsn/demo.py:145-147 (Q6: the maximum runs over the WHOLE batch):
//   X_bg = sum_c H_bg[b][c];  Y[b][0] = 0.15 * expit(max_batch(X_bg) - X_bg)
__global__ void hsn_sum_max_kernel(const float *__restrict__ Hbg, int Cb, int N, long long total, float *__restrict__ xbg,
                                   unsigned int *__restrict__ mx) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / N;
        const int p = (int)(i - b * N);
        const float *src = Hbg + b * Cb * N + p;
        float acc = 0.f;
        for (int c = 0; c < Cb; ++c) acc += src[(long long)c * N];
        xbg[i] = acc;
        m = fmaxf(m, acc);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(mx, __float_as_uint(m)); // X_bg >= 0: uint order = float order
}
__global__ void hsn_voc_bg_kernel(const float *__restrict__ xbg, const unsigned int *__restrict__ mx, int Ctot, int N,
                                  long long total, float *__restrict__ y) {
    const float m = __uint_as_float(*mx);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / N;
        const int p = (int)(i - b * N);
        y[b * Ctot * N + p] = 0.15f / (1.0f + expf(-(m - xbg[i])));
    }
}
// mass[b][c] = 1 if maps[b][c] has a positive entry (all entries >= 0: the sum of dcrf_process :425 is > 0 exactly then)
__global__ __launch_bounds__(256) void hsn_mass_kernel(const float *__restrict__ maps, int N, unsigned int *__restrict__ mass) {
    const float *src = maps + (long long)blockIdx.y * N;
    bool any = false;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < N; p += gridDim.x * blockDim.x) any = any || src[p] > 0.f;
    if (__ballot(any) != 0ull && (threadIdx.x & 63) == 0) mass[blockIdx.y] = 1u;
}

inline int grid_for(long long total, int cap = 8192) {
    long long g = (total + 255) / 256;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

} // namespace

extern "C" {

int wsc_hsn_gradcam_post(wsc_ctx *ctx, const float *cams_nhwc_dev, int B, int h, int w, int C, int S, const float *gate_dev,
                         float *out_dev, int out_channels, int out_first) {
    WSC_CHECK(ctx && cams_nhwc_dev && gate_dev && out_dev, WSC_ERR_INVALID, "wsc_hsn_gradcam_post: null argument");
    if (out_channels <= 0) { out_channels = C; out_first = 0; }
    WSC_CHECK(out_first >= 0 && out_first + C <= out_channels, WSC_ERR_INVALID,
              "wsc_hsn_gradcam_post: channels [%d, %d) do not fit a stack of %d", out_first, out_first + C, out_channels);
    WSC_CHECK(B > 0 && h > 0 && w > 0 && C > 0 && S > 0 && (long long)B * C <= 65535, WSC_ERR_INVALID,
              "wsc_hsn_gradcam_post: bad shape B=%d h=%d w=%d C=%d S=%d", B, h, w, C, S);
    const size_t lds = (size_t)h * w * sizeof(float);
    WSC_CHECK(lds <= 64 * 1024, WSC_ERR_INVALID, "wsc_hsn_gradcam_post: a %dx%d map does not fit the 64 KB LDS tile", h, w);
    WSC_HIP(hipSetDevice(ctx->device));
    unsigned int *mx = nullptr;
    WSC_TRY(wsc_ctx_cached_alloc(ctx, sizeof(unsigned int) * (size_t)B, (void **)&mx));
    WscCachedGuard mx_guard(ctx, mx);
    WSC_HIP(hipMemsetAsync(mx, 0, sizeof(unsigned int) * (size_t)B, ctx->stream));
    // bands of output rows: a few blocks per CU over the B * C maps
    int nb = std::max(1, std::min(S, (4 * ctx->num_cus + B * C - 1) / (B * C)));
    const int band = (S + nb - 1) / nb;
    nb = (S + band - 1) / band;
    const dim3 grid((unsigned)nb, (unsigned)(B * C));
    WscKernelTimer timer(ctx, WSC_K_CAM_TAIL, (double)B * C * S * S * 4);
    hipLaunchKernelGGL(hsn_gradcam_post_kernel<false>, grid, dim3(256), lds, ctx->stream, cams_nhwc_dev, h, w, C, S, gate_dev, mx,
                       out_dev, out_channels, out_first, band);
    hipLaunchKernelGGL(hsn_gradcam_post_kernel<true>, grid, dim3(256), lds, ctx->stream, cams_nhwc_dev, h, w, C, S, gate_dev, mx,
                       out_dev, out_channels, out_first, band);
    WSC_HIP(hipGetLastError());
    mx_guard.free_now(); // stream-ordered reuse
    return WSC_OK;
}

int wsc_hsn_background(wsc_ctx *ctx, const uint8_t *rgb_dev, int B, int H, int W, int Ho, int Wo, double *bg_dev) {
    WSC_CHECK(ctx && rgb_dev && bg_dev, WSC_ERR_INVALID, "wsc_hsn_background: null argument");
    WSC_CHECK(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, WSC_ERR_INVALID, "wsc_hsn_background: B=%d H=%d W=%d -> %dx%d", B, H, W,
              Ho, Wo);
    WSC_HIP(hipSetDevice(ctx->device));
    const long long total = (long long)B * H * W;
    const bool resize = Ho != H || Wo != W;
    double *t1 = nullptr, *t2 = nullptr;
    WSC_TRY(wsc_ctx_cached_alloc(ctx, sizeof(double) * (size_t)total, (void **)&t1));
    WscCachedGuard t1_guard(ctx, t1);
    WSC_TRY(wsc_ctx_cached_alloc(ctx, sizeof(double) * (size_t)total, (void **)&t2));
    WscCachedGuard t2_guard(ctx, t2);
    GaussTaps t; // scipy.ndimage._filters._gaussian_kernel1d(sigma = 2, order = 0, radius = 8)
    double ws[17], sum = 0.0;
    for (int k = -8; k <= 8; ++k) {
        ws[k + 8] = std::exp(-0.5 / 4.0 * (double)(k * k));
        sum += ws[k + 8];
    }
    for (int k = 0; k < 17; ++k) t.w[k] = ws[k] / sum;
    double *last = resize ? t1 : bg_dev;
    hipLaunchKernelGGL(hsn_bg_sigmoid_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, rgb_dev, total, t1);
    hipLaunchKernelGGL(hsn_gauss1d_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, (const double *)t1, H, W, 0, total, t, t2);
    hipLaunchKernelGGL(hsn_gauss1d_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, (const double *)t2, H, W, 1, total, t,
                       last);
    if (resize) {
        const long long to = (long long)B * Ho * Wo;
        hipLaunchKernelGGL(hsn_resize_f64_kernel, dim3(grid_for(to)), dim3(256), 0, ctx->stream, (const double *)t1, H, W, Ho, Wo, to,
                           bg_dev);
    }
    WSC_HIP(hipGetLastError());
    t1_guard.free_now();
    t2_guard.free_now();
    return WSC_OK;
}

int wsc_cam_adp_modify(wsc_ctx *ctx, const float *cam_dev, int B, int n_sc, int C, int hw, const double *bg_dev, int mode,
                       const int32_t *use_host, int n_use, const int32_t *adipose_host, int n_adip, const int32_t *exc_host,
                       int n_exc, float *out_dev) {
    WSC_CHECK(ctx && cam_dev && bg_dev && out_dev && use_host && adipose_host, WSC_ERR_INVALID, "wsc_cam_adp_modify: null argument");
    WSC_CHECK(B > 0 && B <= 65535 && n_sc > 0 && C > 0 && hw > 0 && (mode == 0 || mode == 1), WSC_ERR_INVALID,
              "wsc_cam_adp_modify: B=%d n_sc=%d C=%d hw=%d mode=%d", B, n_sc, C, hw, mode);
    WSC_CHECK(n_use > 0 && n_use <= 64 && n_adip > 0 && n_adip <= 4 && n_exc >= 0 && n_exc <= 4 && (mode == 0 || (exc_host && n_exc > 0)),
              WSC_ERR_INVALID, "wsc_cam_adp_modify: n_use=%d n_adip=%d n_exc=%d", n_use, n_adip, n_exc);
    WSC_HIP(hipSetDevice(ctx->device));
    AdpModArgs a;
    memset(&a, 0, sizeof(a));
    a.cam = cam_dev; a.bg = bg_dev; a.out = out_dev;
    a.n_sc = n_sc; a.C = C; a.hw = hw; a.mode = mode; a.n_use = n_use; a.n_adip = n_adip; a.n_exc = mode ? n_exc : 0;
    auto in_range = [&](int c) { return c >= 0 && c < C; };
    for (int i = 0; i < n_use; ++i) {
        WSC_CHECK(in_range(use_host[i]), WSC_ERR_INVALID, "wsc_cam_adp_modify: use channel %d out of range", use_host[i]);
        a.use[i] = use_host[i];
    }
    for (int k = 0; k < n_adip; ++k) {
        WSC_CHECK(in_range(adipose_host[k]), WSC_ERR_INVALID, "wsc_cam_adp_modify: adipose channel %d out of range", adipose_host[k]);
        a.adip[k] = adipose_host[k];
    }
    for (int k = 0; k < a.n_exc; ++k) {
        WSC_CHECK(in_range(exc_host[k]), WSC_ERR_INVALID, "wsc_cam_adp_modify: exception channel %d out of range", exc_host[k]);
        a.exc[k] = exc_host[k];
    }
    WscKernelTimer timer(ctx, WSC_K_CAM_TAIL, (double)B * hw * 4 * ((double)n_sc * (n_use + 6) + n_use + 2));
    const dim3 grid((unsigned)std::min((hw + 255) / 256, 256), (unsigned)B);
    hipLaunchKernelGGL(cam_adp_modify_kernel, grid, dim3(256), 0, ctx->stream, a);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

int wsc_hsn_cs_gradcam(wsc_ctx *ctx, const float *H_dev, int B, int C_all, int N, const double *bg_dev,
                       const int32_t *src_of_valid_host, int Cv, int bg_ind, int other_ind,
                       const int32_t *exception_inds_host, int n_exc, const int32_t *adipose_src_host, int n_adip,
                       float *cs_dev, float *y_dev, uint32_t *mass_dev) {
    WSC_CHECK(ctx && H_dev && src_of_valid_host && cs_dev && mass_dev, WSC_ERR_INVALID,
              "wsc_hsn_cs_gradcam: null argument");
    WSC_CHECK(B > 0 && B <= 65535 && C_all > 0 && N > 0 && Cv > 0 && Cv <= 32, WSC_ERR_INVALID,
              "wsc_hsn_cs_gradcam: B=%d C_all=%d N=%d Cv=%d (at most 32 valid classes)", B, C_all, N, Cv);
    WSC_CHECK(bg_ind >= 0 && bg_ind < Cv && other_ind < Cv && n_exc >= 0 && n_exc <= 4 && n_adip >= 0 && n_adip <= 4,
              WSC_ERR_INVALID, "wsc_hsn_cs_gradcam: bad class bookkeeping");
    WSC_CHECK(other_ind < 0 || bg_dev == nullptr || (adipose_src_host && n_adip > 0), WSC_ERR_INVALID,
              "wsc_hsn_cs_gradcam: functional types need the adipose channels (03c_hsn/utilities.py:335-336)");
    WSC_HIP(hipSetDevice(ctx->device));
    CsArgs a;
    memset(&a, 0, sizeof(a));
    a.H = H_dev; a.bg = bg_dev; a.cs = cs_dev; a.y = y_dev; a.mass = mass_dev;
    a.C_all = C_all; a.Cv = Cv; a.N = N; a.bg_ind = bg_ind; a.other_ind = other_ind;
    for (int c = 0; c < 32; ++c) a.src_of[c] = -1;
    for (int c = 0; c < Cv; ++c) {
        WSC_CHECK(src_of_valid_host[c] < C_all, WSC_ERR_INVALID, "wsc_hsn_cs_gradcam: source channel %d out of range",
                  src_of_valid_host[c]);
        a.src_of[c] = src_of_valid_host[c];
    }
    a.n_exc = n_exc;
    for (int k = 0; k < n_exc; ++k) a.exc[k] = exception_inds_host[k];
    a.n_adip = n_adip;
    for (int k = 0; k < n_adip; ++k) {
        WSC_CHECK(adipose_src_host[k] >= 0 && adipose_src_host[k] < C_all, WSC_ERR_INVALID, "adipose channel out of range");
        a.adip[k] = adipose_src_host[k];
    }
    WSC_HIP(hipMemsetAsync(mass_dev, 0, sizeof(uint32_t) * (size_t)B * Cv, ctx->stream));
    const dim3 grid((unsigned)std::min((N + 255) / 256, 128), (unsigned)B);
    WscKernelTimer timer(ctx, WSC_K_CAM_TAIL, (double)B * N * 4 * (2.0 * Cv + 1));
    hipLaunchKernelGGL(hsn_cs_kernel, grid, dim3(256), 0, ctx->stream, a);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

int wsc_hsn_voc_background(wsc_ctx *ctx, const float *Hbg_dev, int B, int Cb, int N, float *y_dev, int Ctot) {
    WSC_CHECK(ctx && Hbg_dev && y_dev, WSC_ERR_INVALID, "wsc_hsn_voc_background: null argument");
    WSC_CHECK(B > 0 && Cb > 0 && N > 0 && Ctot > 0, WSC_ERR_INVALID, "wsc_hsn_voc_background: bad shape");
    WSC_HIP(hipSetDevice(ctx->device));
    const long long total = (long long)B * N;
    float *xbg = nullptr;
    WSC_TRY(wsc_ctx_cached_alloc(ctx, sizeof(float) * (size_t)total + 256, (void **)&xbg));
    WscCachedGuard xbg_guard(ctx, xbg);
    unsigned int *mx = reinterpret_cast<unsigned int *>(xbg + total);
    WSC_HIP(hipMemsetAsync(mx, 0, sizeof(unsigned int), ctx->stream));
    hipLaunchKernelGGL(hsn_sum_max_kernel, dim3(grid_for(total, 2048)), dim3(256), 0, ctx->stream, Hbg_dev, Cb, N, total, xbg, mx);
    hipLaunchKernelGGL(hsn_voc_bg_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, (const float *)xbg,
                       (const unsigned int *)mx, Ctot, N, total, y_dev);
    WSC_HIP(hipGetLastError());
    xbg_guard.free_now();
    return WSC_OK;
}

int wsc_hsn_class_mass(wsc_ctx *ctx, const float *maps_dev, int n_maps, int N, uint32_t *mass_dev) {
    WSC_CHECK(ctx && maps_dev && mass_dev, WSC_ERR_INVALID, "wsc_hsn_class_mass: null argument");
    WSC_CHECK(n_maps > 0 && n_maps <= 65535 && N > 0, WSC_ERR_INVALID, "wsc_hsn_class_mass: n_maps=%d N=%d", n_maps, N);
    WSC_HIP(hipSetDevice(ctx->device));
    WSC_HIP(hipMemsetAsync(mass_dev, 0, sizeof(uint32_t) * (size_t)n_maps, ctx->stream));
    hipLaunchKernelGGL(hsn_mass_kernel, dim3((unsigned)std::min((N + 255) / 256, 64), (unsigned)n_maps), dim3(256), 0, ctx->stream,
                       maps_dev, N, mass_dev);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

int wsc_hsn_gather_unary(wsc_ctx *ctx, const float *maps_dev, const int64_t *chan_off_host, int n_chan, int N, float *unary_dev) {
    WSC_CHECK(ctx && maps_dev && chan_off_host && unary_dev, WSC_ERR_INVALID, "wsc_hsn_gather_unary: null argument");
    WSC_CHECK(n_chan > 0 && N > 0, WSC_ERR_INVALID, "wsc_hsn_gather_unary: n_chan=%d N=%d", n_chan, N);
    WSC_HIP(hipSetDevice(ctx->device));
    long long *d = nullptr;
    WSC_TRY(wsc_ctx_cached_alloc(ctx, sizeof(long long) * (size_t)n_chan, (void **)&d));
    WscCachedGuard d_guard(ctx, d);
    WSC_TRY(wsc_ctx_upload_small(ctx, d, chan_off_host, sizeof(long long) * (size_t)n_chan));
    const long long total = (long long)n_chan * N;
    WscKernelTimer timer(ctx, WSC_K_CAM_TAIL, (double)total * 8);
    hipLaunchKernelGGL(hsn_gather_unary_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, maps_dev, (const long long *)d, N,
                       total, unary_dev);
    WSC_HIP(hipGetLastError());
    d_guard.free_now();
    return WSC_OK;
}

} // extern "C"
