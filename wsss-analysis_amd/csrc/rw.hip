// rw.hip -- IRNet random-walk propagation of CAM scores along edge-derived affinities.
//
// Reference: misc.indexing.propagate_to_edge(x, edge, radius=5, beta=10, exp_times=8), called by
// 03b_irn/step/make_sem_seg_labels.py:59,76,93.  The `misc` package is not in the reference tree; the
// algorithm restated here is upstream IRNet's (jiwoon-ahn/irn, misc/indexing.py), whose affinity step the
// reference does carry in-tree as AffinityDisplacementLoss.to_affinity (net/vgg16_irn.py:247-261):
//   for every search direction d = (dy, dx) inside the radius and every pixel p with q = p + d on the grid
//       aff(p, q) = 1 - max over the pixels of the straight path p -> q of edge
//   A = symmetric (hw x hw) matrix of those affinities with ones on the diagonal
//   T = A^beta with every COLUMN divided by its sum;  T <- T @ T, exp_times times;  rw = (x * (1 - edge)) @ T
//
// The reference materialises T (138 M entries at 94 x 125) and squares it 8 times: 8 x 2 hw^3 = 26 TFLOP per
// image.  x @ T^(2^e) is the same as 2^e applications of the 69-point stencil T to the K score maps:
//   v <- (v + sum_d S_d (.) shift(v, +d) + shift(S_d (.) v, -d)) / colsum,      S_d(p) = aff(p, p + d)^beta
// 2^e x K x hw x 69 MACs = 9 GFLOP -- four orders of magnitude less work and no hw x hw matrix.  The state is
// kept in fp64: 256 sequential fp32 steps drift by 1.4e-4 from the exact value of the reference's expression
// (its own 8 fp32 squarings by 4.5e-5); in fp64 the stencil reproduces that value to 1e-12, the arithmetic is
// still negligible (MI355X runs fp64 vector math at half the fp32 rate) and the fixed summation order makes it
// bit-reproducible.  Every step is one small L2-bound kernel over K x hw values; the affinity maps S (34 x hw)
// stay resident in L2.
#include "common.h"

#include <cmath>
#include <cstring>

namespace {

// S[d][p] = aff(p, p + dir_d)^beta, 0 when p + dir_d is off the grid
__global__ void rw_affinity_kernel(const float *__restrict__ edge, int h, int w, const int32_t *__restrict__ dirs,
                                   const int32_t *__restrict__ path_start, const int32_t *__restrict__ path_yx, int D,
                                   float beta, double *__restrict__ S) {
    const int hw = h * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)D * hw;
         i += (long long)gridDim.x * blockDim.x) {
        const int d = (int)(i / hw);
        const int p = (int)(i - (long long)d * hw);
        const int y = p / w, x = p - y * w;
        const int qy = y + dirs[2 * d], qx = x + dirs[2 * d + 1];
        double s = 0.0;
        if (qy >= 0 && qy < h && qx >= 0 && qx < w) {
            float m = -3.0e38f; // the straight path between two grid pixels stays on the grid
            for (int c = path_start[d]; c < path_start[d + 1]; ++c)
                m = fmaxf(m, edge[(y + path_yx[2 * c]) * w + (x + path_yx[2 * c + 1])]);
            s = pow((double)(1.f - m), (double)beta); // 1 - edge is an fp32 value in the reference as well
        }
        S[i] = s;
    }
}

// inv_col[j] = 1 / (1 + sum over neighbours i of j of A[i][j]^beta)
__global__ void rw_colsum_kernel(const double *__restrict__ S, int h, int w, const int32_t *__restrict__ dirs, int D,
                                 double *__restrict__ inv_col) {
    const int hw = h * w;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < hw; j += gridDim.x * blockDim.x) {
        const int y = j / w, x = j - y * w;
        double c = 1.0;
        for (int d = 0; d < D; ++d) {
            const int dy = dirs[2 * d], dx = dirs[2 * d + 1];
            c += S[(long long)d * hw + j]; // 0 when j + d is off the grid
            const int py = y - dy, px = x - dx;
            if (py >= 0 && py < h && px >= 0 && px < w) c += S[(long long)d * hw + py * w + px];
        }
        inv_col[j] = 1.0 / c;
    }
}

// one application of T to K maps: out[k][j] = (in[k][j] + sum_d S_d[j] in[k][j+d] + S_d[j-d] in[k][j-d]) * inv_col[j]
__global__ __launch_bounds__(256) void rw_step_kernel(const double *__restrict__ in, const double *__restrict__ S,
                                                      const double *__restrict__ inv_col, int K, int h, int w,
                                                      const int32_t *__restrict__ dirs, int D, double *__restrict__ out,
                                                      float *__restrict__ out_f32) {
    const int hw = h * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)K * hw;
         i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i / hw);
        const int j = (int)(i - (long long)k * hw);
        const int y = j / w, x = j - y * w;
        const double *v = in + (long long)k * hw;
        double acc = v[j];
        for (int d = 0; d < D; ++d) {
            const int dy = dirs[2 * d], dx = dirs[2 * d + 1];
            const int qy = y + dy, qx = x + dx;
            if (qy < h && qx >= 0 && qx < w) acc += S[(long long)d * hw + j] * v[qy * w + qx];
            const int py = y - dy, px = x - dx;
            if (py >= 0 && px >= 0 && px < w) {
                const int pj = py * w + px;
                acc += S[(long long)d * hw + pj] * v[pj];
            }
        }
        acc *= inv_col[j];
        if (out_f32) out_f32[i] = (float)acc; // the last step writes the result
        else out[i] = acc;
    }
}

__global__ void rw_mask_kernel(const float *__restrict__ x, const float *__restrict__ edge, int K, int hw,
                               double *__restrict__ v, float *__restrict__ v_f32) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)K * hw;
         i += (long long)gridDim.x * blockDim.x)
    {
        const float m = x[i] * (1.f - edge[i % hw]); // fp32 product, as the reference forms it
        v[i] = (double)m;
        if (v_f32) v_f32[i] = m;
    }
}

inline int grid_for(long long total) {
    long long g = (total + 255) / 256;
    if (g > 65535) g = 65535;
    if (g < 1) g = 1;
    return (int)g;
}

} // namespace

extern "C" int wsc_rw_propagate(wsc_ctx *ctx, const float *x_dev, const float *edge_dev, int K, int h, int w,
                                const int32_t *dirs_host, const int32_t *path_start_host, const int32_t *path_yx_host,
                                int D, float beta, int n_steps, float *rw_dev) {
    WSC_CHECK(ctx && x_dev && edge_dev && rw_dev && dirs_host && path_start_host && path_yx_host, WSC_ERR_INVALID,
              "wsc_rw_propagate: null argument");
    WSC_CHECK(K > 0 && h > 0 && w > 0 && D > 0 && D <= 1024 && n_steps >= 0, WSC_ERR_INVALID,
              "wsc_rw_propagate: K=%d h=%d w=%d D=%d n_steps=%d", K, h, w, D, n_steps);
    WSC_HIP(hipSetDevice(ctx->device));
    const int hw = h * w;
    const int n_path = path_start_host[D];
    for (int d = 0; d < D; ++d) {
        WSC_CHECK(path_start_host[d] <= path_start_host[d + 1], WSC_ERR_INVALID, "wsc_rw_propagate: path table");
        // direction order of PathIndex: dy > 0, or dy == 0 and dx > 0 (each unordered pair once)
        WSC_CHECK(dirs_host[2 * d] > 0 || (dirs_host[2 * d] == 0 && dirs_host[2 * d + 1] > 0), WSC_ERR_INVALID,
                  "wsc_rw_propagate: direction %d = (%d, %d) is not in the upper half plane", d, dirs_host[2 * d],
                  dirs_host[2 * d + 1]);
    }
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t tbl = al(sizeof(int32_t) * (2 * (size_t)D + (D + 1) + 2 * (size_t)n_path));
    const size_t sb = al(sizeof(double) * (size_t)D * hw), cb = al(sizeof(double) * hw), vb = al(sizeof(double) * (size_t)K * hw);
    void *ws;
    WSC_TRY(wsc_ctx_workspace(ctx, tbl + sb + cb + 2 * vb, &ws));
    char *p = (char *)ws;
    int32_t *dirs = (int32_t *)p;
    int32_t *pstart = dirs + 2 * D;
    int32_t *pyx = pstart + (D + 1);
    p += tbl;
    double *S = (double *)p; p += sb;
    double *inv_col = (double *)p; p += cb;
    double *va = (double *)p; p += vb;
    double *vbuf = (double *)p; p += vb;
    std::vector<int32_t> packed(2 * (size_t)D + (D + 1) + 2 * (size_t)n_path);
    memcpy(packed.data(), dirs_host, sizeof(int32_t) * 2 * D);
    memcpy(packed.data() + 2 * D, path_start_host, sizeof(int32_t) * (D + 1));
    memcpy(packed.data() + 2 * D + D + 1, path_yx_host, sizeof(int32_t) * 2 * n_path);
    WSC_TRY(wsc_ctx_upload_small(ctx, dirs, packed.data(), packed.size() * sizeof(int32_t)));

    hipLaunchKernelGGL(rw_affinity_kernel, dim3(grid_for((long long)D * hw)), dim3(256), 0, ctx->stream, edge_dev, h, w,
                       dirs, pstart, pyx, D, beta, S);
    hipLaunchKernelGGL(rw_colsum_kernel, dim3(grid_for(hw)), dim3(256), 0, ctx->stream, S, h, w, dirs, D, inv_col);
    hipLaunchKernelGGL(rw_mask_kernel, dim3(grid_for((long long)K * hw)), dim3(256), 0, ctx->stream, x_dev, edge_dev, K,
                       hw, va, n_steps == 0 ? rw_dev : (float *)nullptr);
    double *cur = va, *nxt = vbuf;
    for (int s = 0; s < n_steps; ++s) {
        hipLaunchKernelGGL(rw_step_kernel, dim3(grid_for((long long)K * hw)), dim3(256), 0, ctx->stream, cur, S, inv_col,
                           K, h, w, dirs, D, nxt, s == n_steps - 1 ? rw_dev : (float *)nullptr);
        double *t = cur; cur = nxt; nxt = t;
    }
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}
