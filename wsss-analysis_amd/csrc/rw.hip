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

#include <algorithm>
#include <cmath>
#include <cstring>

namespace {

// one image of a batch: K score maps on an h x w grid; offsets into the packed buffers
struct RwImg {
    int K, h, w, hw;
    long long x_off;   // floats into x / rw / the fp64 state buffers (K*hw each)
    long long e_off;   // floats into edge / the fp64 column-sum buffer (hw each)
    long long s_off;   // doubles into S (D*hw each)
};

// S[d][p] = aff(p, p + dir_d)^beta, 0 when p + dir_d is off the grid        (grid: x over D*hw, y = image)
__global__ void rw_affinity_kernel(const RwImg *__restrict__ imgs, const float *__restrict__ edge_all,
                                   const int32_t *__restrict__ dirs, const int32_t *__restrict__ path_start,
                                   const int32_t *__restrict__ path_yx, int D, float beta, float *__restrict__ S_all) {
    const RwImg im = imgs[blockIdx.y];
    const int h = im.h, w = im.w, hw = im.hw;
    const float *edge = edge_all + im.e_off;
    float *S = S_all + im.s_off;
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
        S[i] = (float)s; // the weights are stored in fp32 (they derive from one fp32 value); the sums run in fp64
    }
}

// inv_col[j] = 1 / (1 + sum over neighbours i of j of A[i][j]^beta)
__global__ void rw_colsum_kernel(const RwImg *__restrict__ imgs, const float *__restrict__ S_all,
                                 const int32_t *__restrict__ dirs, int D, double *__restrict__ inv_col_all) {
    const RwImg im = imgs[blockIdx.y];
    const int h = im.h, w = im.w, hw = im.hw;
    const float *S = S_all + im.s_off;
    double *inv_col = inv_col_all + im.e_off;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < hw; j += gridDim.x * blockDim.x) {
        const int y = j / w, x = j - y * w;
        double c = 1.0;
        for (int d = 0; d < D; ++d) {
            const int dy = dirs[2 * d], dx = dirs[2 * d + 1];
            c += (double)S[(long long)d * hw + j]; // 0 when j + d is off the grid
            const int py = y - dy, px = x - dx;
            if (py >= 0 && py < h && px >= 0 && px < w) c += (double)S[(long long)d * hw + py * w + px];
        }
        inv_col[j] = 1.0 / c;
    }
}

// one application of T to the K maps of every image:
//   out[k][j] = (in[k][j] + sum_d S_d[j] in[k][j+d] + S_d[j-d] in[k][j-d]) * inv_col[j]
// One thread per (map, pixel).  (One thread per pixel looping over the maps re-uses the weight loads and was
// 13 % faster on a 32-image pass, but halves the parallelism of a single-image call: 4.0 -> 7.2 ms.)
__global__ __launch_bounds__(256) void rw_step_kernel(const RwImg *__restrict__ imgs, const double *__restrict__ in_all,
                                                      const float *__restrict__ S_all,
                                                      const double *__restrict__ inv_col_all,
                                                      const int32_t *__restrict__ dirs, int D,
                                                      double *__restrict__ out_all, float *__restrict__ out_f32_all) {
    const RwImg im = imgs[blockIdx.y];
    const int h = im.h, w = im.w, hw = im.hw;
    const float *S = S_all + im.s_off;
    const double *inv_col = inv_col_all + im.e_off;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)im.K * hw;
         i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i / hw);
        const int j = (int)(i - (long long)k * hw);
        const int y = j / w, x = j - y * w;
        const double *v = in_all + im.x_off + (long long)k * hw;
        double acc = v[j];
        for (int d = 0; d < D; ++d) {
            const int dy = dirs[2 * d], dx = dirs[2 * d + 1];
            const int qy = y + dy, qx = x + dx;
            if (qy < h && qx >= 0 && qx < w) acc += (double)S[(long long)d * hw + j] * v[qy * w + qx];
            const int py = y - dy, px = x - dx;
            if (py >= 0 && px >= 0 && px < w) {
                const int pj = py * w + px;
                acc += (double)S[(long long)d * hw + pj] * v[pj];
            }
        }
        acc *= inv_col[j];
        if (out_f32_all) out_f32_all[im.x_off + i] = (float)acc; // the last step writes the result
        else out_all[im.x_off + i] = acc;
    }
}

// The same step on 16 x 16 pixel tiles: the tile's values with a halo of RW_R pixels are staged in LDS (float64, RW_KC
// maps at a time), a thread owns one pixel and walks the directions once per group of maps -- the two weights of a direction
// are loaded once for the group, the neighbour values come from LDS at offsets that are uniform over the block, and nothing
// is divided per value.  Per (map, pixel) the additions run in the order of rw_step_kernel (direction by direction, +d then
// -d, the same predicates): bit-identical.  Directions reach at most RW_R pixels (the host checks; IRNet's radius is 5).
// RW_KC: maps per group -- 2, 4 or 8 by the largest map count of the batch (a group costs one walk over the directions
// whatever it holds; 5.4 KB of LDS per map).
constexpr int RW_T = 16, RW_R = 5, RW_W = RW_T + 2 * RW_R;
template <int RW_KC>
__global__ __launch_bounds__(RW_T * RW_T) void rw_step_tile_kernel(const RwImg *__restrict__ imgs, const double *__restrict__ in_all,
                                                                    const float *__restrict__ S_all,
                                                                    const double *__restrict__ inv_col_all,
                                                                    const int32_t *__restrict__ dirs, int D,
                                                                    double *__restrict__ out_all, float *__restrict__ out_f32_all,
                                                                    int max_tiles, int xcd_map) {
    __shared__ double vt[RW_KC][RW_W * RW_W];
    // 1-D grid of n_img * max_tiles blocks.  Blocks go round-robin over the 8 XCDs (block b -> XCD b % 8): with the plain order
    // every XCD's 4 MB L2 sees tiles of every image, i.e. all 34 weight maps of the whole batch (51 MB at 32 VOC images) plus
    // the halo overlap.  The bijective remap gives an XCD a contiguous range of logical ids = whole images (4 of 32), whose
    // weights and state it then re-reads from its own L2 for all 2^8 steps.
    int lb = (int)blockIdx.x;
    if (xcd_map) {
        const int nb = (int)gridDim.x, xcd = lb & 7, q = nb >> 3, r = nb & 7;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lb >> 3);
    }
    const int img = lb / max_tiles, tile = lb - img * max_tiles;
    const RwImg im = imgs[img];
    const int h = im.h, w = im.w, hw = im.hw;
    const int ntx = (w + RW_T - 1) / RW_T, nty = (h + RW_T - 1) / RW_T;
    if (tile >= ntx * nty) return;
    const int ty0 = (tile / ntx) * RW_T, tx0 = (tile % ntx) * RW_T;
    const int ty = (int)threadIdx.x / RW_T, tx = (int)threadIdx.x % RW_T;
    const int y = ty0 + ty, x = tx0 + tx;
    const bool ok = y < h && x < w;
    const int j = y * w + x;
    const float *S = S_all + im.s_off;
    const double ic = ok ? inv_col_all[im.e_off + j] : 0.0;
    const int lc = (ty + RW_R) * RW_W + tx + RW_R; // the pixel's slot in the staged tile
    for (int k0 = 0; k0 < im.K; k0 += RW_KC) {
        const int kc = min(RW_KC, im.K - k0);
        __syncthreads(); // (the previous group's reads are done)
        for (int i = threadIdx.x; i < kc * RW_W * RW_W; i += RW_T * RW_T) {
            const int k = i / (RW_W * RW_W), r = i - k * (RW_W * RW_W);
            const int yy = ty0 - RW_R + r / RW_W, xx = tx0 - RW_R + r % RW_W;
            double v = 0.0;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) v = in_all[im.x_off + (long long)(k0 + k) * hw + yy * w + xx];
            vt[k][r] = v;
        }
        __syncthreads();
        if (ok) {
            double acc[RW_KC];
#pragma unroll
            for (int k = 0; k < RW_KC; ++k) acc[k] = vt[k][lc];
            // (requesting the weights of four directions together before using them -- 8 loads in flight per thread instead
            // of 2 -- measured 7 % slower: 36 VGPRs instead of 12 and the predicates kept twice, 0.269 -> 0.287 ms per image;
            // unrolling this loop by 2 / 4: 0.267 / 0.265 -- the step is not waiting for its weight loads)
            for (int d = 0; d < D; ++d) {
                const int dy = dirs[2 * d], dx = dirs[2 * d + 1]; // (uniform)
                const int qy = y + dy, qx = x + dx;
                if (qy < h && qx >= 0 && qx < w) {
                    const double s1 = (double)S[(long long)d * hw + j];
                    const int o = lc + dy * RW_W + dx;
#pragma unroll
                    for (int k = 0; k < RW_KC; ++k)
                        if (k < kc) acc[k] += s1 * vt[k][o];
                }
                const int py = y - dy, px = x - dx;
                if (py >= 0 && px >= 0 && px < w) {
                    const double s2 = (double)S[(long long)d * hw + py * w + px];
                    const int o = lc - dy * RW_W - dx;
#pragma unroll
                    for (int k = 0; k < RW_KC; ++k)
                        if (k < kc) acc[k] += s2 * vt[k][o];
                }
            }
#pragma unroll
            for (int k = 0; k < RW_KC; ++k)
                if (k < kc) {
                    const double r = acc[k] * ic;
                    const long long o = im.x_off + (long long)(k0 + k) * hw + j;
                    if (out_f32_all) out_f32_all[o] = (float)r; // the last step writes the result
                    else out_all[o] = r;
                }
        }
    }
}

__global__ void rw_mask_kernel(const RwImg *__restrict__ imgs, const float *__restrict__ x_all,
                               const float *__restrict__ edge_all, double *__restrict__ v_all,
                               float *__restrict__ v_f32_all) {
    const RwImg im = imgs[blockIdx.y];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)im.K * im.hw;
         i += (long long)gridDim.x * blockDim.x) {
        const float m = x_all[im.x_off + i] * (1.f - edge_all[im.e_off + i % im.hw]); // fp32, as the reference forms it
        v_all[im.x_off + i] = (double)m;
        if (v_f32_all) v_f32_all[im.x_off + i] = m;
    }
}

inline int grid_for(long long total) {
    long long g = (total + 255) / 256;
    if (g > 65535) g = 65535;
    if (g < 1) g = 1;
    return (int)g;
}

} // namespace

extern "C" int wsc_rw_propagate_batch(wsc_ctx *ctx, int n_img, const int32_t *K_host, const int32_t *h_host,
                                      const int32_t *w_host, const float *x_dev, const float *edge_dev,
                                      const int32_t *dirs_host, const int32_t *path_start_host,
                                      const int32_t *path_yx_host, int D, float beta, int n_steps, float *rw_dev) {
    WSC_CHECK(ctx && K_host && h_host && w_host && x_dev && edge_dev && rw_dev && dirs_host && path_start_host &&
                  path_yx_host,
              WSC_ERR_INVALID, "wsc_rw_propagate: null argument");
    WSC_CHECK(n_img > 0 && n_img <= 65535 && D > 0 && D <= 1024 && n_steps >= 0, WSC_ERR_INVALID,
              "wsc_rw_propagate: n_img=%d D=%d n_steps=%d", n_img, D, n_steps);
    WSC_HIP(hipSetDevice(ctx->device));
    const int n_path = path_start_host[D];
    for (int d = 0; d < D; ++d) {
        WSC_CHECK(path_start_host[d] <= path_start_host[d + 1], WSC_ERR_INVALID, "wsc_rw_propagate: path table");
        // direction order of PathIndex: dy > 0, or dy == 0 and dx > 0 (each unordered pair once)
        WSC_CHECK(dirs_host[2 * d] > 0 || (dirs_host[2 * d] == 0 && dirs_host[2 * d + 1] > 0), WSC_ERR_INVALID,
                  "wsc_rw_propagate: direction %d = (%d, %d) is not in the upper half plane", d, dirs_host[2 * d],
                  dirs_host[2 * d + 1]);
    }
    std::vector<RwImg> imgs(n_img);
    long long xo = 0, eo = 0, so = 0, max_khw = 0, max_hw = 0;
    for (int b = 0; b < n_img; ++b) {
        WSC_CHECK(K_host[b] > 0 && h_host[b] > 0 && w_host[b] > 0, WSC_ERR_INVALID,
                  "wsc_rw_propagate: image %d has K=%d h=%d w=%d", b, K_host[b], h_host[b], w_host[b]);
        RwImg &im = imgs[b];
        im.K = K_host[b]; im.h = h_host[b]; im.w = w_host[b]; im.hw = im.h * im.w;
        im.x_off = xo; im.e_off = eo; im.s_off = so;
        xo += (long long)im.K * im.hw;
        eo += im.hw;
        so += (long long)D * im.hw;
        max_khw = std::max(max_khw, (long long)im.K * im.hw);
        max_hw = std::max(max_hw, (long long)im.hw);
    }
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t tbl_ints = 2 * (size_t)D + (D + 1) + 2 * (size_t)n_path;
    const size_t tbl = al(sizeof(int32_t) * tbl_ints + 8) + al(sizeof(RwImg) * (size_t)n_img);
    const size_t sb = al(sizeof(float) * (size_t)so), cb = al(sizeof(double) * (size_t)eo), vb = al(sizeof(double) * (size_t)xo);
    void *ws;
    WSC_TRY(wsc_ctx_workspace(ctx, tbl + sb + cb + 2 * vb, &ws));
    char *p = (char *)ws;
    int32_t *dirs = (int32_t *)p;
    int32_t *pstart = dirs + 2 * D;
    int32_t *pyx = pstart + (D + 1);
    RwImg *imgs_dev = (RwImg *)(p + al(sizeof(int32_t) * tbl_ints + 8));
    p += tbl;
    float *S = (float *)p; p += sb;
    double *inv_col = (double *)p; p += cb;
    double *va = (double *)p; p += vb;
    double *vbuf = (double *)p; p += vb;
    std::vector<int32_t> packed(tbl_ints);
    memcpy(packed.data(), dirs_host, sizeof(int32_t) * 2 * D);
    memcpy(packed.data() + 2 * D, path_start_host, sizeof(int32_t) * (D + 1));
    memcpy(packed.data() + 2 * D + D + 1, path_yx_host, sizeof(int32_t) * 2 * n_path);
    WSC_TRY(wsc_ctx_upload_small(ctx, dirs, packed.data(), packed.size() * sizeof(int32_t)));
    WSC_TRY(wsc_ctx_upload_small(ctx, imgs_dev, imgs.data(), imgs.size() * sizeof(RwImg)));

    const dim3 g_khw(grid_for(max_khw), n_img), g_hw(grid_for(max_hw), n_img), g_dhw(grid_for(max_hw * D), n_img);
    hipLaunchKernelGGL(rw_affinity_kernel, g_dhw, dim3(256), 0, ctx->stream, imgs_dev, edge_dev, dirs, pstart, pyx, D,
                       beta, S);
    hipLaunchKernelGGL(rw_colsum_kernel, g_hw, dim3(256), 0, ctx->stream, imgs_dev, S, dirs, D, inv_col);
    hipLaunchKernelGGL(rw_mask_kernel, g_khw, dim3(256), 0, ctx->stream, imgs_dev, x_dev, edge_dev, va,
                       n_steps == 0 ? rw_dev : (float *)nullptr);
    double *cur = va, *nxt = vbuf;
    // tiled step (values through LDS) when every direction stays inside its halo; WSC_RW_TILED=0 keeps the flat kernel
    bool tiled = ctx->opt[WSC_OPT_RW_TILED] != 0;
    for (int d = 0; d < D; ++d) tiled = tiled && dirs_host[2 * d] <= RW_R && std::abs(dirs_host[2 * d + 1]) <= RW_R;
    int max_tiles = 1, max_K = 1;
    long long all_tiles = 0;
    for (int b = 0; b < n_img; ++b) {
        max_K = std::max(max_K, imgs[b].K);
        const int t = ((imgs[b].h + RW_T - 1) / RW_T) * ((imgs[b].w + RW_T - 1) / RW_T);
        max_tiles = std::max(max_tiles, t);
        all_tiles += t;
    }
    // a thread of the tiled kernel owns ALL maps of a pixel, so a small call has too few threads for it.  With the
    // XCD-contiguous block order (round 6) it wins from 4 VOC-sized images on -- ms per image, flat / tiled, 94 x 125, K = 2,
    // 2^8 steps: 1 image (48 tiles) 3.95 / 4.95, 2: 2.03 / 2.46, 4: 1.53 / 1.22, 8: 0.91 / 0.62, 16: 0.50 / 0.35, 24: 0.61 / 0.34,
    // 32: - / 0.27 (0.31 before the block order) -- smaller calls keep the flat kernel
    if (ctx->opt[WSC_OPT_RW_TILED] < 0) tiled = tiled && all_tiles * 8 >= 5ll * ctx->num_cus;
#ifdef WSC_AB_KNOBS
    static const int xcd_map = [] { const char *e = getenv("WSC_RW_XCD"); return e ? atoi(e) : 1; }();
#else
    constexpr int xcd_map = 1;
#endif
    for (int s = 0; s < n_steps; ++s) {
        if (tiled) {
            const dim3 tg((unsigned)max_tiles * (unsigned)n_img);
            float *of = s == n_steps - 1 ? rw_dev : (float *)nullptr;
            if (max_K <= 2)
                hipLaunchKernelGGL(rw_step_tile_kernel<2>, tg, dim3(RW_T * RW_T), 0, ctx->stream, imgs_dev, cur, S, inv_col, dirs, D, nxt, of, max_tiles, xcd_map);
            else if (max_K <= 4)
                hipLaunchKernelGGL(rw_step_tile_kernel<4>, tg, dim3(RW_T * RW_T), 0, ctx->stream, imgs_dev, cur, S, inv_col, dirs, D, nxt, of, max_tiles, xcd_map);
            else
                hipLaunchKernelGGL(rw_step_tile_kernel<8>, tg, dim3(RW_T * RW_T), 0, ctx->stream, imgs_dev, cur, S, inv_col, dirs, D, nxt, of, max_tiles, xcd_map);
        }
        else
        hipLaunchKernelGGL(rw_step_kernel, g_khw, dim3(256), 0, ctx->stream, imgs_dev, cur, S, inv_col, dirs, D, nxt,
                           s == n_steps - 1 ? rw_dev : (float *)nullptr);
        double *t = cur; cur = nxt; nxt = t;
    }
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

extern "C" int wsc_rw_propagate(wsc_ctx *ctx, const float *x_dev, const float *edge_dev, int K, int h, int w,
                                const int32_t *dirs_host, const int32_t *path_start_host, const int32_t *path_yx_host,
                                int D, float beta, int n_steps, float *rw_dev) {
    const int32_t k = K, hh = h, ww = w;
    return wsc_rw_propagate_batch(ctx, 1, &k, &hh, &ww, x_dev, edge_dev, dirs_host, path_start_host, path_yx_host, D,
                                  beta, n_steps, rw_dev);
}
