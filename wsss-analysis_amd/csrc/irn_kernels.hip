// irn_kernels.hip -- the non-GEMM pieces of the IRNet EdgeDisplacement heads.
//
// Reference: 03b_irn/net/resnet50_irn.py:22-92 / vgg16_irn.py:28-178 / m7_irn.py -- every head is
//   nn.Conv2d(Cin, Cout, 1, bias=False[, stride=2]) -> nn.GroupNorm(G, Cout) [-> nn.Upsample(scale, 'bilinear',
//   align_corners=False)] -> nn.ReLU, its output cropped ([..., :h, :w]) and concatenated along channels
//   (Net.forward, resnet50_irn.py:110-132); EdgeDisplacement.forward :218-232 ends with
//   sigmoid(edge[0]/2 + edge[1].flip(-1)/2) and dp[0] (MeanShift subtracts running_mean in eval, :96-108).
// The 1x1 convolutions run on conv_igemm.hip with an fp32 NHWC output; the kernels here do
//   gn_partial / gn_finish : GroupNorm statistics per (sample, group), double accumulation, fixed reduction
//                            order (bit-reproducible run to run)
//   gn_apply               : (x - mean) * rstd * gamma + beta, bilinear x`up` upsample (GroupNorm is a
//                            per-channel affine map, so it commutes with the interpolation), crop, ReLU and
//                            the write into the channel slice of the NHWC concat buffer (16-bit planes)
//   edge_finish            : the flip-average + sigmoid, the crop to the feature size, dp[0]
// All of them are small HBM-bound passes (two samples at 128 x 128 x <= 256 channels).
#include "common.h"

namespace {

constexpr int GN_CHUNK = 1024; // pixels per partial block

// x: fp32 NHWC [N][HW][C]; partial[(n*G + g)*nchunk + chunk] = {sum, sumsq} over the chunk's pixels x Cg
__global__ __launch_bounds__(256) void gn_partial_kernel(const float *__restrict__ x, int HW, int C, int G, int nchunk,
                                                         double2 *__restrict__ partial) {
    const int ng = blockIdx.x; // n * G + g
    const int chunk = blockIdx.y;
    const int n = ng / G, g = ng - n * G;
    const int Cg = C / G;
    const int p0 = chunk * GN_CHUNK;
    const int p1 = min(p0 + GN_CHUNK, HW);
    double s = 0.0, ss = 0.0;
    const long long total = (long long)(p1 - p0) * Cg;
    for (long long i = threadIdx.x; i < total; i += 256) {
        const int p = p0 + (int)(i / Cg);
        const int c = (int)(i - (long long)(p - p0) * Cg);
        const double v = (double)x[((long long)n * HW + p) * C + g * Cg + c];
        s += v;
        ss += v * v;
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(long long)ng * nchunk + chunk] = make_double2(sh[0][0], sh[1][0]);
}

// The same sums with 16-byte loads and 32-bit index math (Cg % 4 == 0): a thread owns (pixel, channel quad) items of the
// chunk.  Double accumulation in a fixed order as above (another order than the scalar kernel's: the statistics agree to
// ~1e-16 relative, not bit for bit).  The scalar kernel divided a 64-bit index per element and ran at a third of this rate.
__global__ __launch_bounds__(256) void gn_partial4_kernel(const float *__restrict__ x, int HW, int C, int G, int nchunk,
                                                          double2 *__restrict__ partial) {
    const int ng = blockIdx.x, chunk = blockIdx.y;
    const int n = ng / G, g = ng - n * G;
    const int Cg = C / G, Q = Cg >> 2;
    const int p0 = chunk * GN_CHUNK, p1 = min(p0 + GN_CHUNK, HW);
    const float *xb = x + ((long long)n * HW + p0) * C + g * Cg;
    double s = 0.0, ss = 0.0;
    const unsigned total = (unsigned)(p1 - p0) * (unsigned)Q;
    for (unsigned i = threadIdx.x; i < total; i += 256) {
        const unsigned p = i / (unsigned)Q, q = i - p * (unsigned)Q;
        const float4 v = *reinterpret_cast<const float4 *>(xb + (size_t)p * C + q * 4);
        const double a0 = v.x, a1 = v.y, a2 = v.z, a3 = v.w;
        s += a0; ss += a0 * a0;
        s += a1; ss += a1 * a1;
        s += a2; ss += a2 * a2;
        s += a3; ss += a3 * a3;
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(long long)ng * nchunk + chunk] = make_double2(sh[0][0], sh[1][0]);
}

// stats[ng] = {mean, rstd}; biased variance, eps inside the sqrt (torch.nn.GroupNorm)
__global__ void gn_finish_kernel(const double2 *__restrict__ partial, int NG, int nchunk, double count, float eps,
                                 float2 *__restrict__ stats) {
    const int ng = blockIdx.x * blockDim.x + threadIdx.x;
    if (ng >= NG) return;
    double s = 0.0, ss = 0.0;
    for (int c = 0; c < nchunk; ++c) {
        const double2 v = partial[(long long)ng * nchunk + c];
        s += v.x;
        ss += v.y;
    }
    const double mean = s / count;
    double var = ss / count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[ng] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
}

struct GnApplyArgs {
    const float *x;      // fp32 NHWC [N][H][W][C]
    const float2 *stats; // [N*G]
    const float *gamma, *beta;
    bf16_t *y, *y_lo;    // NHWC [N][Hd][Wd][Ctot], this head writes channels [coff, coff + C)
    int N, H, W, C, G, up, relu, Hd, Wd, Ctot, coff, fmt, split;
};

__global__ __launch_bounds__(256) void gn_apply_kernel(GnApplyArgs a) {
    const long long total = (long long)a.N * a.Hd * a.Wd * a.C;
    const int Cg = a.C / a.G;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % a.C);
        long long t = i / a.C;
        const int wo = (int)(t % a.Wd);
        t /= a.Wd;
        const int ho = (int)(t % a.Hd);
        const int n = (int)(t / a.Hd);
        float v;
        const float *xn = a.x + (long long)n * a.H * a.W * a.C + c;
        if (a.up == 1) {
            v = xn[((long long)ho * a.W + wo) * a.C];
        } else {
            // torch upsample_bilinear2d, align_corners=False, scale_factor given: src = (dst + 0.5) / up - 0.5
            const float inv = 1.0f / (float)a.up;
            float sy = ((float)ho + 0.5f) * inv - 0.5f, sx = ((float)wo + 0.5f) * inv - 0.5f;
            sy = sy < 0.f ? 0.f : sy;
            sx = sx < 0.f ? 0.f : sx;
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + (y0 < a.H - 1 ? 1 : 0), x1 = x0 + (x0 < a.W - 1 ? 1 : 0);
            const float ly = sy - (float)y0, lx = sx - (float)x0;
            const float hy = 1.f - ly, hx = 1.f - lx;
            const float v00 = xn[((long long)y0 * a.W + x0) * a.C], v01 = xn[((long long)y0 * a.W + x1) * a.C];
            const float v10 = xn[((long long)y1 * a.W + x0) * a.C], v11 = xn[((long long)y1 * a.W + x1) * a.C];
            v = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
        }
        const float2 st = a.stats[n * a.G + c / Cg];
        v = (v - st.x) * st.y * a.gamma[c] + a.beta[c];
        if (a.relu) v = fmaxf(v, 0.f);
        const long long o = (((long long)n * a.Hd + ho) * a.Wd + wo) * a.Ctot + a.coff + c;
        const bf16_t h = f32_to_h16(v, a.fmt);
        a.y[o] = h;
        if (a.split) a.y_lo[o] = f32_to_h16(v - h16_to_f32(h, a.fmt), a.fmt); // the lo plane has the hi plane's format
    }
}

// The same map, eight channels of one output pixel per thread (C, Ctot, coff multiples of 8; fewer than 2^31 items): 32-byte
// loads per tap, one 16-byte store per 16-bit plane, 32-bit index math.  Every element goes through the expressions of the
// scalar kernel above (same bits); that kernel -- three 64-bit divisions and two 2-byte stores per ELEMENT -- took 0.61 ms per
// head at 32 VOC-sized images, 10 % of the IRNet pass.
template <bool UP>
__global__ __launch_bounds__(256) void gn_apply8_kernel(GnApplyArgs a) {
    const unsigned C8 = (unsigned)a.C >> 3;
    const unsigned total = (unsigned)a.N * (unsigned)a.Hd * (unsigned)a.Wd * C8;
    const int Cg = a.C / a.G;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned c8 = i % C8;
        unsigned t = i / C8;
        const int wo = (int)(t % (unsigned)a.Wd);
        t /= (unsigned)a.Wd;
        const int ho = (int)(t % (unsigned)a.Hd);
        const int n = (int)(t / (unsigned)a.Hd);
        const int c0 = (int)c8 * 8;
        const float *xn = a.x + (long long)n * a.H * a.W * a.C + c0;
        float v[8];
        if (!UP) {
            const float4 *p = reinterpret_cast<const float4 *>(xn + ((long long)ho * a.W + wo) * a.C);
            const float4 lo4 = p[0], hi4 = p[1];
            v[0] = lo4.x; v[1] = lo4.y; v[2] = lo4.z; v[3] = lo4.w; v[4] = hi4.x; v[5] = hi4.y; v[6] = hi4.z; v[7] = hi4.w;
        } else {
            const float inv = 1.0f / (float)a.up;
            float sy = ((float)ho + 0.5f) * inv - 0.5f, sx = ((float)wo + 0.5f) * inv - 0.5f;
            sy = sy < 0.f ? 0.f : sy;
            sx = sx < 0.f ? 0.f : sx;
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + (y0 < a.H - 1 ? 1 : 0), x1 = x0 + (x0 < a.W - 1 ? 1 : 0);
            const float ly = sy - (float)y0, lx = sx - (float)x0;
            const float hy = 1.f - ly, hx = 1.f - lx;
            float t00[8], t01[8], t10[8], t11[8];
            const float *q[4] = {xn + ((long long)y0 * a.W + x0) * a.C, xn + ((long long)y0 * a.W + x1) * a.C,
                                 xn + ((long long)y1 * a.W + x0) * a.C, xn + ((long long)y1 * a.W + x1) * a.C};
            float *dst[4] = {t00, t01, t10, t11};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 lo4 = reinterpret_cast<const float4 *>(q[k])[0], hi4 = reinterpret_cast<const float4 *>(q[k])[1];
                dst[k][0] = lo4.x; dst[k][1] = lo4.y; dst[k][2] = lo4.z; dst[k][3] = lo4.w;
                dst[k][4] = hi4.x; dst[k][5] = hi4.y; dst[k][6] = hi4.z; dst[k][7] = hi4.w;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = hy * (hx * t00[j] + lx * t01[j]) + ly * (hx * t10[j] + lx * t11[j]);
        }
        uint16_t hh[8], ll[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + j;
            const float2 st = a.stats[n * a.G + c / Cg];
            float r = (v[j] - st.x) * st.y * a.gamma[c] + a.beta[c];
            if (a.relu) r = fmaxf(r, 0.f);
            hh[j] = f32_to_h16(r, a.fmt);
            ll[j] = a.split ? f32_to_h16(r - h16_to_f32(hh[j], a.fmt), a.fmt) : (uint16_t)0;
        }
        const long long o = (((long long)n * a.Hd + ho) * a.Wd + wo) * a.Ctot + a.coff + c0;
        uint4 ph, pw;
        ph.x = hh[0] | ((unsigned)hh[1] << 16); ph.y = hh[2] | ((unsigned)hh[3] << 16);
        ph.z = hh[4] | ((unsigned)hh[5] << 16); ph.w = hh[6] | ((unsigned)hh[7] << 16);
        *reinterpret_cast<uint4 *>(a.y + o) = ph;
        if (a.split) {
            pw.x = ll[0] | ((unsigned)ll[1] << 16); pw.y = ll[2] | ((unsigned)ll[3] << 16);
            pw.z = ll[4] | ((unsigned)ll[5] << 16); pw.w = ll[6] | ((unsigned)ll[7] << 16);
            *reinterpret_cast<uint4 *>(a.y_lo + o) = pw;
        }
    }
}

// e: fp32 [2B][He][We] (Cout = 1), d: fp32 [2B][Hd][Wd][2] (the M7 net has its edge map at twice the resolution)
__global__ void edge_finish_kernel(const float *__restrict__ e, int He, int We, const float *__restrict__ d, int Hd,
                                   int Wd, int B, int fh, int fw, float ms0, float ms1, float *__restrict__ edge,
                                   float *__restrict__ dp) {
    const long long total = (long long)B * fh * fw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % fw);
        long long t = i / fw;
        const int y = (int)(t % fh);
        const int b = (int)(t / fh);
        const float e0 = e[((long long)(2 * b) * He + y) * We + x];
        const float e1 = e[((long long)(2 * b + 1) * He + y) * We + (fw - 1 - x)]; // crop to fw, then flip(-1)
        const float z = e0 / 2.f + e1 / 2.f;
        edge[i] = 1.f / (1.f + expf(-z));
        const float *dd = d + (((long long)(2 * b) * Hd + y) * Wd + x) * 2;
        dp[((long long)b * 2 + 0) * fh * fw + (long long)y * fw + x] = dd[0] - ms0;
        dp[((long long)b * 2 + 1) * fh * fw + (long long)y * fw + x] = dd[1] - ms1;
    }
}

inline int grid_for(long long total) {
    long long g = (total + 255) / 256;
    if (g > 65535) g = 65535;
    if (g < 1) g = 1;
    return (int)g;
}

} // namespace

// x fp32 NHWC [N][H][W][C]; `partial` needs N*G*ceil(H*W/1024) double2, `stats` N*G float2
int launch_group_norm_stats(wsc_ctx *ctx, const float *x, int N, int H, int W, int C, int G, float eps, void *partial,
                            void *stats) {
    WSC_CHECK(G > 0 && C % G == 0, WSC_ERR_INVALID, "GroupNorm: %d channels in %d groups", C, G);
    const int HW = H * W;
    const int nchunk = (HW + GN_CHUNK - 1) / GN_CHUNK;
    if ((C / G) % 4 == 0 && ((uintptr_t)x & 15) == 0)
        hipLaunchKernelGGL(gn_partial4_kernel, dim3(N * G, nchunk), dim3(256), 0, ctx->stream, x, HW, C, G, nchunk,
                           (double2 *)partial);
    else
        hipLaunchKernelGGL(gn_partial_kernel, dim3(N * G, nchunk), dim3(256), 0, ctx->stream, x, HW, C, G, nchunk,
                           (double2 *)partial);
    hipLaunchKernelGGL(gn_finish_kernel, dim3((N * G + 63) / 64), dim3(64), 0, ctx->stream, (const double2 *)partial,
                       N * G, nchunk, (double)HW * (C / G), eps, (float2 *)stats);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}
size_t group_norm_partial_bytes(int N, int H, int W, int G) {
    return sizeof(double2) * (size_t)N * G * ((H * W + GN_CHUNK - 1) / GN_CHUNK);
}

int launch_group_norm_apply(wsc_ctx *ctx, const float *x, const void *stats, const float *gamma, const float *beta,
                            int N, int H, int W, int C, int G, int up, int relu, bf16_t *y, bf16_t *y_lo, int Hd, int Wd,
                            int Ctot, int coff, int fmt) {
    WSC_CHECK(Hd <= H * up && Wd <= W * up, WSC_ERR_INVALID, "GroupNorm apply: crop %dx%d larger than %dx%d", Hd, Wd,
              H * up, W * up);
    GnApplyArgs a;
    a.x = x; a.stats = (const float2 *)stats; a.gamma = gamma; a.beta = beta; a.y = y; a.y_lo = y_lo;
    a.N = N; a.H = H; a.W = W; a.C = C; a.G = G; a.up = up; a.relu = relu; a.Hd = Hd; a.Wd = Wd; a.Ctot = Ctot;
    a.coff = coff; a.fmt = fmt; a.split = y_lo != nullptr;
    const long long items8 = (long long)N * Hd * Wd * (C / 8);
    const bool vec = C % 8 == 0 && Ctot % 8 == 0 && coff % 8 == 0 && items8 < (1ll << 31) && ((uintptr_t)x & 15) == 0 &&
                     ((uintptr_t)y & 15) == 0 && (!y_lo || ((uintptr_t)y_lo & 15) == 0);
    if (vec && up == 1)
        hipLaunchKernelGGL(gn_apply8_kernel<false>, dim3(grid_for(items8)), dim3(256), 0, ctx->stream, a);
    else if (vec)
        hipLaunchKernelGGL(gn_apply8_kernel<true>, dim3(grid_for(items8)), dim3(256), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(gn_apply_kernel, dim3(grid_for((long long)N * Hd * Wd * C)), dim3(256), 0, ctx->stream, a);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

int launch_edge_finish(wsc_ctx *ctx, const float *e, int He, int We, const float *d, int Hd, int Wd, int B, int fh, int fw,
                       float ms0, float ms1, float *edge, float *dp) {
    WSC_CHECK(fh <= He && fw <= We && fh <= Hd && fw <= Wd, WSC_ERR_INVALID,
              "edge: feature size %dx%d exceeds the maps %dx%d / %dx%d", fh, fw, He, We, Hd, Wd);
    hipLaunchKernelGGL(edge_finish_kernel, dim3(grid_for((long long)B * fh * fw)), dim3(256), 0, ctx->stream, e, He, We,
                       d, Hd, Wd, B, fh, fw, ms0, ms1, edge, dp);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}
