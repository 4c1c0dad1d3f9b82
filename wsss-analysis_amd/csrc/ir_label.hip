// ir_label.hip -- cam_to_ir_label on the device (03b_irn/step/cam_to_ir_label.py:26-75).
//   wsc_label_unary_from_cam  labels = argmax([thres | high_res], axis 0) (np.pad + np.argmax, :30-31 / :45-51 / :65-67)
//                             and pydensecrf.utils.unary_from_labels(labels, n_labels, gt_prob, zero_unsure=False) as
//                             imutils.crf_inference_label calls it: energy -log(gt_prob) for the pixel's label,
//                             -log((1 - gt_prob) / (n_labels - 1)) for every other label
//   wsc_ir_label_combine      conf = keys[fg_pred]; VOC: conf[fg == 0] = 255, conf[bg + fg == 0] = 0 (:54-57);
//                             ADP / DeepGlobe (keys[0] = -1): conf[fg == -1] = 255 (:38-40, :71-73)
// Batches of B images of one size and one class count share the calls (and the lattices of one wsc_crf).
#include "common.h"

#include <cmath>

namespace {

__global__ void label_unary_kernel(const float *__restrict__ maps, int K, int N, long long total, float thres, float p_energy,
                                   float n_energy, float *__restrict__ unary, int32_t *__restrict__ labels) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / N;
        const int p = (int)(i - b * N);
        const float *src = maps + b * K * N + p;
        float best = thres;
        int idx = 0;
        for (int k = 0; k < K; ++k) {
            const float v = src[(long long)k * N];
            if (v > best) { // strict: np.argmax keeps the first maximum (the padded channel comes first)
                best = v;
                idx = k + 1;
            }
        }
        if (labels) labels[i] = idx;
        float *dst = unary + b * (K + 1) * N + p;
        for (int m = 0; m <= K; ++m) dst[(long long)m * N] = m == idx ? p_energy : n_energy;
    }
}

__global__ void ir_combine_kernel(const int32_t *__restrict__ fg, const int32_t *__restrict__ bg, const int32_t *__restrict__ keys,
                                  int M, int N, long long total, int voc, uint8_t *__restrict__ conf) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / N;
        const int32_t *kb = keys + b * M;
        const int f = kb[fg[i]];
        int c = f;
        if (voc) {
            const int g = kb[bg[i]];
            if (f == 0) c = 255;
            if (g + f == 0) c = 0;
        } else if (f == -1) {
            c = 255;
        }
        conf[i] = (uint8_t)c;
    }
}

inline int grid_for(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

} // namespace

extern "C" {

int wsc_label_unary_from_cam(wsc_ctx *ctx, const float *highres_dev, int B, int K, int N, float thres, float gt_prob,
                             float *unary_dev, int32_t *labels_dev) {
    WSC_CHECK(ctx && highres_dev && unary_dev, WSC_ERR_INVALID, "wsc_label_unary_from_cam: null argument");
    WSC_CHECK(B > 0 && K >= 1 && N > 0 && gt_prob > 0.f && gt_prob < 1.f, WSC_ERR_INVALID,
              "wsc_label_unary_from_cam: B=%d K=%d N=%d gt_prob=%g (`gt_prob must be in (0,1)`, n_labels = K + 1 >= 2)", B, K, N,
              (double)gt_prob);
    WSC_HIP(hipSetDevice(ctx->device));
    const double n_energy = -std::log((1.0 - (double)gt_prob) / (double)K); // n_labels - 1 = K
    const double p_energy = -std::log((double)gt_prob);
    const long long total = (long long)B * N;
    WscKernelTimer timer(ctx, WSC_K_CAM_TAIL, (double)total * 4 * (2.0 * K + 1));
    hipLaunchKernelGGL(label_unary_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, highres_dev, K, N, total, thres,
                       (float)p_energy, (float)n_energy, unary_dev, labels_dev);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

int wsc_ir_label_combine(wsc_ctx *ctx, const int32_t *fg_pred_dev, const int32_t *bg_pred_dev, const int32_t *keys_host, int B, int M,
                         int N, uint8_t *conf_dev) {
    WSC_CHECK(ctx && fg_pred_dev && keys_host && conf_dev, WSC_ERR_INVALID, "wsc_ir_label_combine: null argument");
    WSC_CHECK(B > 0 && M > 0 && N > 0, WSC_ERR_INVALID, "wsc_ir_label_combine: B=%d M=%d N=%d", B, M, N);
    WSC_HIP(hipSetDevice(ctx->device));
    int32_t *k = nullptr;
    WSC_TRY(wsc_ctx_cached_alloc(ctx, sizeof(int32_t) * (size_t)B * M, (void **)&k));
    WscCachedGuard k_guard(ctx, k);
    WSC_TRY(wsc_ctx_upload_small(ctx, k, keys_host, sizeof(int32_t) * (size_t)B * M));
    const long long total = (long long)B * N;
    hipLaunchKernelGGL(ir_combine_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, fg_pred_dev, bg_pred_dev,
                       (const int32_t *)k, M, N, total, bg_pred_dev != nullptr ? 1 : 0, conf_dev);
    WSC_HIP(hipGetLastError());
    k_guard.free_now();
    return WSC_OK;
}

} // extern "C"
