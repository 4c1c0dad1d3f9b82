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
    WSC_TRY(wsc_ctx_upload_small(ctx, d, jobs.data(), sizeof(InJob) * (size_t)B));
    const dim3 grid((unsigned)std::min((S * S + 255) / 256, 64), (unsigned)B);
    WscKernelTimer timer(ctx, WSC_K_POOL_MISC, (double)B * S * S * 3 * (pair ? 8 : 4));
    hipLaunchKernelGGL(msf_input_kernel, grid, dim3(256), 0, ctx->stream, images_dev, (const InJob *)d, S, mean3_host[0],
                       mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2], pre_div255, pair, x_dev);
    WSC_HIP(hipGetLastError());
    wsc_ctx_cached_free(ctx, d);
    return WSC_OK;
}

} // extern "C"
