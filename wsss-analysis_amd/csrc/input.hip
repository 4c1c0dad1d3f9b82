// input.hip -- the MSF dataset transform on the device: decoded uint8 images in, network input out.
//
// Reference (host, numpy / cv2, per image): 03b_irn/voc12/dataloader.py:68-106, 225-246 --
//   TorchvisionResize: np.asarray(img, 'float64'); cv2.resize(img, outsize) (bilinear) if the shape differs
//   TorchvisionNormalize('int'): float32(img)[..., c] = (img[..., c] - mean[c]) / std[c]   (mean 104/117/123 on R,G,B: Q3)
//   HWC_to_CHW, np.stack([x, np.flip(x, -1)])                                             -> float32 (2, 3, S, S)
// ADP / DeepGlobe use other constants (adp/dataloader.py:64-80, deepglobe/dataloader.py:60-66); 02_cues / 03c_hsn
// normalise a plain batch without the flip pair (02_cues/utilities.py:146-181, adp_cues.py:130).
// The float64 bilinear resize is a few ms of numpy per image on the host and the float32 pair is 8x the bytes of the
// decoded image over PCIe; here one HBM-bound kernel does resize + normalise + layout + flip.
//
// This file is compiled with -ffp-contract=off: the interpolation is the same sequence of float64 products and sums
// as the host expression (top * (1 - wy) + bot * wy with top = a * (1 - wx) + b * wx), so the result is bit-identical
// to wsscam.voc12.dataloader.resize_bilinear_f64 + TorchvisionNormalize.
#include "common.h"

#include <cstring>
#include <vector>

namespace {

struct InJob {
    long long src_off; // byte offset of the image in the packed uint8 buffer
    int H0, W0;
};

__global__ __launch_bounds__(256) void msf_input_kernel(const uint8_t *__restrict__ src, const InJob *__restrict__ jobs, int S,
                                                        float m0, float m1, float m2, float s0, float s1, float s2,
                                                        int pre_div255, int pair, float *__restrict__ out) {
    const InJob job = jobs[blockIdx.y];
    const uint8_t *im = src + job.src_off;
    const int H = job.H0, W = job.W0;
    const bool same = H == S && W == S;
    const int n = S * S;
    float *dst = out + (long long)blockIdx.y * (pair ? 2 : 1) * 3 * n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int yy = i / S, xx = i - yy * S;
        double v[3];
        if (same) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = (double)im[((long long)yy * W + xx) * 3 + c];
        } else {
            double fy = ((double)yy + 0.5) * (double)H / (double)S - 0.5;
            double fx = ((double)xx + 0.5) * (double)W / (double)S - 0.5;
            fy = fy < 0.0 ? 0.0 : (fy > (double)(H - 1) ? (double)(H - 1) : fy);
            fx = fx < 0.0 ? 0.0 : (fx > (double)(W - 1) ? (double)(W - 1) : fx);
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = y0 + 1 < H ? y0 + 1 : H - 1, x1 = x0 + 1 < W ? x0 + 1 : W - 1;
            const double wy = fy - (double)y0, wx = fx - (double)x0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double a = (double)im[((long long)y0 * W + x0) * 3 + c], b = (double)im[((long long)y0 * W + x1) * 3 + c];
                const double d = (double)im[((long long)y1 * W + x0) * 3 + c], e = (double)im[((long long)y1 * W + x1) * 3 + c];
                const double top = a * (1.0 - wx) + b * wx;
                const double bot = d * (1.0 - wx) + e * wx;
                v[c] = top * (1.0 - wy) + bot * wy;
            }
        }
        const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float f = (float)v[c];
            if (pre_div255) f = f / 255.0f;         // norm_mode 'float': (img / 255 - mean) / std
            f = (f - mean[c]) / sd[c];
            dst[(long long)c * n + i] = f;
            if (pair) dst[(long long)(3 + c) * n + yy * S + (S - 1 - xx)] = f; // np.flip(x, -1)
        }
    }
}

// cv2.resize(uint8 image, dsize) with the default INTER_LINEAR (read_batch of 02_cues/utilities.py:172-176 and
// 03c_hsn/utilities.py:176-181: the batch is `np.empty(..., dtype='uint8')`, so the network and the CRF see OpenCV's
// 8-bit result).  OpenCV's 8U path is FIXED POINT (public algorithm of modules/imgproc/src/resize.cpp, restated):
//   scale = 1 / (dst / src) in double; per output index f = (float)((d + 0.5) * scale - 0.5), s = floor(f), f -= s;
//   columns: s < 0 -> (s, f) = (0, 0); s >= src - 1 -> (src - 1, 0); rows: both taps clamped to [0, src - 1];
//   coefficients short(rint(c * 2048)) (round half to even) of c = 1.f - f and f;
//   horizontal pass in int: a * alpha0 + b * alpha1;  vertical: (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
//   Exactly 2 x 2 decimation takes OpenCV's INTER_AREA shortcut: (a + b + c + d + 2) >> 2.
// Integer arithmetic: bit-identical to wsscam.voc12.dataloader.resize_bilinear_u8 (numpy) and oracle/hsn_ref.py's loops.
__device__ __forceinline__ void cv2_coef(int d, int src, int dst, bool clamp_taps, int &s, int &c0, int &c1) {
    const double scale = 1.0 / ((double)dst / (double)src);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f -= (float)s;
    if (clamp_taps) {
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= src - 1) { s = src - 1; f = 0.f; }
    }
    c0 = (int)rintf((1.f - f) * 2048.f);
    c1 = (int)rintf(f * 2048.f);
}

__global__ __launch_bounds__(256) void resize_u8_cv2_kernel(const uint8_t *__restrict__ src, const InJob *__restrict__ jobs, int OH,
                                                            int OW, uint8_t *__restrict__ out) {
    const InJob job = jobs[blockIdx.y];
    const uint8_t *im = src + job.src_off;
    const int H = job.H0, W = job.W0;
    const int n = OH * OW;
    uint8_t *dst = out + (long long)blockIdx.y * n * 3;
    const bool same = H == OH && W == OW, half = H == 2 * OH && W == 2 * OW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int yy = i / OW, xx = i - yy * OW;
        if (same) {
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[(long long)i * 3 + c] = im[(long long)i * 3 + c];
            continue;
        }
        if (half) {
            const uint8_t *p = im + ((long long)(2 * yy) * W + 2 * xx) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[(long long)i * 3 + c] = (uint8_t)((p[c] + p[3 + c] + p[W * 3 + c] + p[W * 3 + 3 + c] + 2) >> 2);
            continue;
        }
        int sx, a0, a1, sy, b0, b1;
        cv2_coef(xx, W, OW, true, sx, a0, a1);
        cv2_coef(yy, H, OH, false, sy, b0, b1);
        const int x1 = sx + 1 < W ? sx + 1 : W - 1; // only reached with a1 == 0 when clamped
        const int y0 = sy < 0 ? 0 : (sy < H ? sy : H - 1), y1 = sy + 1 < 0 ? 0 : (sy + 1 < H ? sy + 1 : H - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int S0 = (int)im[((long long)y0 * W + sx) * 3 + c] * a0 + (int)im[((long long)y0 * W + x1) * 3 + c] * a1;
            const int S1 = (int)im[((long long)y1 * W + sx) * 3 + c] * a0 + (int)im[((long long)y1 * W + x1) * 3 + c] * a1;
            int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            dst[(long long)i * 3 + c] = (uint8_t)v;
        }
    }
}

} // namespace

extern "C" {

int wsc_msf_input_u8(wsc_ctx *ctx, const uint8_t *images_dev, int B, const int32_t *size_hw_host, const int64_t *offset_host,
                     int S, const float *mean3_host, const float *std3_host, int pre_div255, int pair, float *x_dev) {
    WSC_CHECK(ctx && images_dev && size_hw_host && offset_host && mean3_host && std3_host && x_dev, WSC_ERR_INVALID,
              "wsc_msf_input_u8: null argument");
    WSC_CHECK(B > 0 && B <= 65535 && S > 0, WSC_ERR_INVALID, "wsc_msf_input_u8: B=%d S=%d", B, S);
    WSC_HIP(hipSetDevice(ctx->device));
    std::vector<InJob> jobs(B);
    for (int b = 0; b < B; ++b) {
        WSC_CHECK(size_hw_host[2 * b] > 0 && size_hw_host[2 * b + 1] > 0, WSC_ERR_INVALID, "image %d has size %dx%d", b,
                  size_hw_host[2 * b], size_hw_host[2 * b + 1]);
        jobs[b].src_off = offset_host[b];
        jobs[b].H0 = size_hw_host[2 * b];
        jobs[b].W0 = size_hw_host[2 * b + 1];
    }
    InJob *d = nullptr;
    WSC_TRY(wsc_ctx_cached_alloc(ctx, sizeof(InJob) * (size_t)B, (void **)&d));
    int st = wsc_ctx_upload_small(ctx, d, jobs.data(), sizeof(InJob) * (size_t)B);
    if (st == WSC_OK) {
        const dim3 grid((unsigned)std::min((S * S + 255) / 256, 64), (unsigned)B);
        WscKernelTimer timer(ctx, WSC_K_POOL_MISC, (double)B * S * S * 3 * (pair ? 8 : 4));
        hipLaunchKernelGGL(msf_input_kernel, grid, dim3(256), 0, ctx->stream, images_dev, (const InJob *)d, S, mean3_host[0],
                           mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2], pre_div255, pair, x_dev);
        if (hipGetLastError() != hipSuccess) {
            wsc_set_error("wsc_msf_input_u8: launch failed");
            st = WSC_ERR_HIP;
        }
    }
    wsc_ctx_cached_free(ctx, d);
    return st;
}

int wsc_resize_u8(wsc_ctx *ctx, const uint8_t *images_dev, int B, const int32_t *size_hw_host, const int64_t *offset_host, int OH,
                  int OW, uint8_t *out_dev) {
    WSC_CHECK(ctx && images_dev && size_hw_host && offset_host && out_dev, WSC_ERR_INVALID, "wsc_resize_u8: null argument");
    WSC_CHECK(B > 0 && B <= 65535 && OH > 0 && OW > 0, WSC_ERR_INVALID, "wsc_resize_u8: B=%d out=%dx%d", B, OH, OW);
    WSC_HIP(hipSetDevice(ctx->device));
    std::vector<InJob> jobs(B);
    for (int b = 0; b < B; ++b) {
        WSC_CHECK(size_hw_host[2 * b] > 0 && size_hw_host[2 * b + 1] > 0, WSC_ERR_INVALID, "image %d has size %dx%d", b,
                  size_hw_host[2 * b], size_hw_host[2 * b + 1]);
        jobs[b].src_off = offset_host[b];
        jobs[b].H0 = size_hw_host[2 * b];
        jobs[b].W0 = size_hw_host[2 * b + 1];
    }
    InJob *d = nullptr;
    WSC_TRY(wsc_ctx_cached_alloc(ctx, sizeof(InJob) * (size_t)B, (void **)&d));
    int st = wsc_ctx_upload_small(ctx, d, jobs.data(), sizeof(InJob) * (size_t)B);
    if (st == WSC_OK) {
        const dim3 grid((unsigned)std::min((OH * OW + 255) / 256, 64), (unsigned)B);
        WscKernelTimer timer(ctx, WSC_K_POOL_MISC, (double)B * OH * OW * 3 * 5);
        hipLaunchKernelGGL(resize_u8_cv2_kernel, grid, dim3(256), 0, ctx->stream, images_dev, (const InJob *)d, OH, OW, out_dev);
        if (hipGetLastError() != hipSuccess) {
            wsc_set_error("wsc_resize_u8: launch failed");
            st = WSC_ERR_HIP;
        }
    }
    wsc_ctx_cached_free(ctx, d);
    return st;
}

} // extern "C"
