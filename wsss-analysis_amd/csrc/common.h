// common.h -- shared declarations of libwsscam (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/wsscam.h"

typedef uint16_t bf16_t; // raw bfloat16 bits

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// ---- bf16 helpers (host + device), round-to-nearest-even --------------------
__host__ __device__ inline bf16_t f32_to_bf16(float f) {
    union { float f; uint32_t u; } v;
    v.f = f;
    uint32_t u = v.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40); // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__host__ __device__ inline float bf16_to_f32(bf16_t h) {
    union { float f; uint32_t u; } v;
    v.u = ((uint32_t)h) << 16;
    return v.f;
}

// ---- IEEE half helpers: round-to-nearest-even, SATURATING at +-65504 (no infinities) -------
__host__ __device__ inline uint16_t f32_to_f16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    const _Float16 h = (_Float16)fminf(fmaxf(f, -65504.f), 65504.f);
    return __builtin_bit_cast(uint16_t, h);
#else
    union { float f; uint32_t u; } v, magic;
    v.f = f;
    const uint32_t sign = v.u & 0x80000000u;
    v.u ^= sign;
    uint16_t o;
    if (v.u >= ((127u + 16u) << 23)) {
        o = v.u > (255u << 23) ? 0x7e00 : 0x7bff; // NaN stays NaN, overflow saturates
    } else if (v.u < (113u << 23)) { // subnormal half or zero
        magic.u = ((127u - 15u) + (23u - 10u) + 1u) << 23;
        v.f += magic.f;
        o = (uint16_t)(v.u - magic.u);
    } else {
        const uint32_t mant_odd = (v.u >> 13) & 1u;
        v.u += ((15u - 127u) << 23) + 0xfffu;
        v.u += mant_odd;
        o = (uint16_t)(v.u >> 13);
        if (o >= 0x7c00) o = 0x7bff;
    }
    return (uint16_t)(o | (sign >> 16));
#endif
}
__host__ __device__ inline float f16_to_f32(uint16_t h) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (float)__builtin_bit_cast(_Float16, h);
#else
    union { float f; uint32_t u; } o, magic;
    magic.u = 113u << 23;
    const uint32_t shifted_exp = 0x7c00u << 13;
    o.u = ((uint32_t)h & 0x7fffu) << 13;
    const uint32_t exp = shifted_exp & o.u;
    o.u += (127u - 15u) << 23;
    if (exp == shifted_exp) o.u += (128u - 16u) << 23;
    else if (exp == 0) { o.u += 1u << 23; o.f -= magic.f; }
    o.u |= ((uint32_t)h & 0x8000u) << 16;
    return o.f;
#endif
}
// 16-bit storage format of activations / weights: 0 = bfloat16, 1 = IEEE half
__host__ __device__ inline uint16_t f32_to_h16(float f, int fmt) { return fmt ? f32_to_f16(f) : f32_to_bf16(f); }
__host__ __device__ inline float h16_to_f32(uint16_t h, int fmt) { return fmt ? f16_to_f32(h) : bf16_to_f32(h); }

// ---- error plumbing -----------------------------------------------------------
void wsc_set_error(const char *fmt, ...);
#define WSC_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            wsc_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,                   \
                          hipGetErrorString(_e));                                              \
            return WSC_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)
#define WSC_CHECK(cond, code, ...)                                                             \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            wsc_set_error(__VA_ARGS__);                                                        \
            return (code);                                                                     \
        }                                                                                      \
    } while (0)
#define WSC_TRY(expr)                                                                          \
    do {                                                                                       \
        int _s = (expr);                                                                       \
        if (_s != WSC_OK) return _s;                                                           \
    } while (0)

// ---- per-kernel-class timing (wsc_profile_begin / _end) --------------------------------------------
enum WscKernelClass {
    WSC_K_CONV256 = 0,   // conv_igemm_kernel, 256x128 tile, 3-stage LDS-DMA pipeline
    WSC_K_CONV128,       // conv_igemm_kernel, 128x128 tile, LDS-DMA staging
    WSC_K_CONV64,        // conv_igemm_kernel, 128x64 tile
    WSC_K_CONV_SMALLCIN, // conv_igemm_kernel, stem / first layer (register staging); stem_pool_kernel (f16x3 ResNet stem + max-pool)
    WSC_K_POOL_MISC,     // maxpool, layout changes, flip-add, classifier branch
    WSC_K_CAM_TAIL,      // cam_tail_kernel (both passes) + unary_from_maps
    WSC_K_CRF_BUILD,     // every kernel of wsc_crf_create
    WSC_K_GAUSS_MSG,     // gauss_msg_kernel: Gaussian lattice combine + three blur passes + slice into E, per pixel tile, in LDS
    WSC_K_BLUR,          // combine4_kernel + blur4_kernel (bilateral) + blur3_tile_kernel (Gaussian, when its message is not formed on chip)
    WSC_K_SLICE_UPDATE,  // update_splat_kernel: slice + mean-field update + splat of the result
    WSC_K_CRF_MISC,      // init_q / finish
    WSC_K_COUNT
};
struct WscProfRecord {
    int cls;
    double work; // algorithmic FLOPs (conv classes) or bytes (everything else) of this launch
    hipEvent_t e0, e1;
};

// ---- context --------------------------------------------------------------------
struct wsc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 0;
    std::string arch;
    // grow-only workspace arena (activations, CRF scratch); never freed before destroy
    void *ws = nullptr;
    size_t ws_bytes = 0;
    // small pinned staging buffers for descriptor uploads: a ring of slots, so that an upload waits for an earlier one only
    // when the ring has gone round (a copy is ordered behind everything enqueued on the stream before it: with one slot the
    // host blocked on the previous call's whole device work)
    static constexpr int PIN_SLOTS = 8;
    struct PinSlot {
        void *p = nullptr;
        size_t bytes = 0;
        hipEvent_t ev = nullptr; // completion of the last copy out of this slot
        bool busy = false;
    };
    PinSlot pin_ring[PIN_SLOTS];
    int pin_next = 0;
    // path selectors (wsc_ctx_set_option): every one picks between two paths that both exist for some inputs and give the
    // same bits -- the tests hold them to that -- with ONE exception: WSC_OPT_CAM_HEAD_STREAM.  cam_head_kernel sums K in four
    // per-wave quarters, a different fp32 summation order from the tiled kernel: equal to fp32 round-off (test bound 2e-6
    // relative), not bit-identical.  Defaults: wsc_option in include/wsscam.h.
    int opt[WSC_OPT_COUNT] = {1, 1, 1, 0, 0, -1, 1, 1, 1, 1};
    void *pinned = nullptr; // (legacy single buffer: unused)
    size_t pinned_bytes = 0;
    void *zero_page = nullptr; // 256 bytes of zeros in HBM (source of padded conv taps)
    // range guard of the IEEE-half conv modes: one word of mapped, page-locked host memory that a conv epilogue stores to
    // (plain store of a non-zero value, no atomic needed: every writer writes "raised") when an activation saturates at the
    // half ceiling; read by the host after a stream synchronisation (wsc_sync, wsc_memcpy_d2h, wsc_ctx_range_status)
    unsigned *range_host = nullptr, *range_dev = nullptr;
    hipEvent_t pinned_ev = nullptr; // completion of the last copy out of `pinned`
    bool pinned_busy = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t join_ev = nullptr; // wsc_ctx_wait
    hipEvent_t marks[8] = {};     // wsc_ctx_mark / wsc_ctx_wait_mark (created on first use)
    bool mark_set[8] = {};
    // side stream for work that is independent inside one call (crf.hip: the bilateral lattice's combine + blur passes run
    // beside the Gaussian lattice's fused blur); forked from / joined into `stream` with the two events
    hipStream_t aux_stream = nullptr;
    hipEvent_t fork_ev = nullptr, aux_done_ev = nullptr;
    bool profiling = false;
    std::vector<WscProfRecord> prof;
    std::vector<hipEvent_t> prof_pool;
    // stream-ordered caching allocator: blocks released by wsc_ctx_cached_free are reused by later
    // requests of the same ctx (all work of a ctx is on one stream, so reuse is ordered after the
    // previous user) instead of going through hipFree/hipMalloc (both synchronise the device).
    std::multimap<size_t, void *> free_blocks;
    std::unordered_map<void *, size_t> live_blocks;
    // host-side objects owned by the ctx (crf.hip keeps its per-image-size Gaussian lattices here);
    // their device arrays are cached-alloc blocks, released with everything else at destroy
    std::vector<std::pair<void *, void (*)(void *)>> attachments;
};
int wsc_ctx_workspace(wsc_ctx *ctx, size_t bytes, void **out);
// WSC_ERR_RANGE (with the error text) when the ctx's range flag is raised; the stream must have been synchronised
int wsc_ctx_range_check(wsc_ctx *ctx);
// One packed pair of IEEE halves as the saturation test of an epilogue: bit 15 / 31 of the result is set iff the low / high
// half's magnitude is >= 0x7bff (65504: the value the saturating conversion stores for anything beyond, and NaN / inf)
__device__ __forceinline__ unsigned half2_at_ceiling(unsigned hw) { return (hw & 0x7fff7fffu) + 0x04010401u; }
// Brackets the launches enqueued during its lifetime with a pair of HIP events on the ctx stream
// when profiling is on (no-op otherwise).
struct WscKernelTimer {
    wsc_ctx *ctx;
    int idx = -1;
    WscKernelTimer(wsc_ctx *c, int cls, double work);
    ~WscKernelTimer();
};
int wsc_ctx_cached_alloc(wsc_ctx *ctx, size_t bytes, void **out);
void wsc_ctx_cached_free(wsc_ctx *ctx, void *p);
// Scope guard of a cached block: an early `return` of WSC_TRY / WSC_HIP / WSC_CHECK gives the block back too
// (free_now() at the usual place keeps the stream-ordered reuse exactly where it was).
struct WscCachedGuard {
    wsc_ctx *ctx;
    void *p;
    WscCachedGuard(wsc_ctx *c, void *q) : ctx(c), p(q) {}
    WscCachedGuard(const WscCachedGuard &) = delete;
    WscCachedGuard &operator=(const WscCachedGuard &) = delete;
    void free_now() {
        if (p) wsc_ctx_cached_free(ctx, p);
        p = nullptr;
    }
    ~WscCachedGuard() { free_now(); }
};
// copies `bytes` of host data to dst_dev through the ctx's pinned staging buffer, asynchronously
int wsc_ctx_upload_small(wsc_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);

// ---- conv (implicit GEMM) -----------------------------------------------------------
// One conv layer as the kernel sees it.  Activations are NHWC bf16; in split
// precision every activation has a second ("lo") plane.
struct ConvLaunch {
    const bf16_t *x, *x_lo;     // input  [N][H][W][Cin]   (Cin = 4 in small-Cin mode)
    const bf16_t *w;            // packed [CoutPad][Kw] 16-bit, K order (cin/64, kh, kw, cin%64); split 1 (and the small-Cin layers of
                                // split 2): [hi K | lo K]; split 2 generic: K order (cin/32, kh, kw) x [32 hi | 32 lo]
    const float *s1, *b1;       // y = acc*s1 + b1 (folded BN, or conv bias with s1 = 1)
    const float *s2, *b2;       // optional post-ReLU affine (VGG's conv->ReLU->BN order), or null
    const bf16_t *res, *res_lo; // optional residual [M][Cout]
    bf16_t *y, *y_lo;           // output [M][Cout] bf16 (may be null when y_f32 is set)
    float *y_f32;               // optional fp32 output [M][Cout]
    int N, H, W, Cin, Ho, Wo, Cout, CoutPad;
    int kh, kw, stride, pad;
    int relu;
    int small_cin; // 0: generic (Cin % 64 == 0); else log2(slots per kernel row): 2 = 7x7 stem, 1 = 3x3 Cin<=4;
                   // 3: f16x3 stem on a zero-padded NHWC4 input (H, W = the padded size, pad = 0; conv_igemm.hip)
    int split;     // 0: one plane; 1: bf16x3 (three K segments); 2: f16x3 (hi + lo staged once per K-step).  The lo plane has the hi plane's format
    int fmt;       // 16-bit operand format: 0 bf16, 1 f16 (split 1 requires bf16, split 2 f16)
    int generic;   // 1: keep the generic kernel variants (testing: the FAST variants give the same bits)
    int ldy;       // row pitch of y / y_lo in elements; 0 = Cout (wider: the output is a channel range of a concatenated tensor)
    // optional second input of a 1x1 / stride 1 layer: the last C2 of the Cin input channels of output pixel (n, ho, wo) come from
    // x2[n][ho * stride2][wo * stride2][0 .. C2) instead of x (which then holds Cin - C2 channels per pixel); null: one input
    const bf16_t *x2, *x2_lo;
    int H2, W2, C2, stride2;
};
int conv_igemm_launch(wsc_ctx *ctx, const ConvLaunch &p);
// cam_head.hip: the 1x1 head with <= 32 output channels as a streaming GEMM (IEEE-half planes, fp32 [M][C] output)
int launch_cam_head(wsc_ctx *ctx, const bf16_t *x, const bf16_t *x_lo, int M, int K, const bf16_t *w, int Kw, int CoutPad,
                    const float *s1, const float *b1, int C, int relu, float *y);

// ---- misc kernels ---------------------------------------------------------------------
int launch_nchw_to_nhwc4(wsc_ctx *ctx, const float *x, int N, int H, int W, bf16_t *y, bf16_t *y_lo, int fmt);
// stem_pool.hip: conv 7x7/2 + BN + ReLU + MaxPool 3x3/2/1 of the f16x3 ResNet stem in one kernel
void stem_pool_input_dims(int H, int W, int *Hp, int *Wp);
int launch_stem_pool(wsc_ctx *ctx, const bf16_t *x, const bf16_t *x_lo, int N, int H, int W, const bf16_t *w, int Kw, const float *s1,
                     const float *b1, int relu, bf16_t *y, bf16_t *y_lo);
// x[n][ho * stride][wo * stride][0 .. C) -> y[(n, ho, wo)][0 .. C) with row pitch ldy (both planes; y points at the first channel)
int launch_gather_strided(wsc_ctx *ctx, const bf16_t *x, const bf16_t *x_lo, int N, int H, int W, int C, int stride, int Ho, int Wo,
                          bf16_t *y, bf16_t *y_lo, int ldy);
// [N][Hp][Wp][4] with a zero border of `pad` pixels on the top / left (and whatever Hp, Wp leave on the bottom / right)
int launch_nchw_to_nhwc4_pad(wsc_ctx *ctx, const float *x, int N, int H, int W, int Hp, int Wp, int pad, bf16_t *y, bf16_t *y_lo,
                             int fmt);
int launch_maxpool(wsc_ctx *ctx, const bf16_t *x, const bf16_t *x_lo, int N, int H, int W, int C, int k,
                   int stride, int pad, int Ho, int Wo, bf16_t *y, bf16_t *y_lo, int fmt);
// cam[b][c][y][x] = relu(head[2b][y][x][c]) + relu(head[2b+1][y][w-1-x][c])   (head fp32 NHWC, stride Cs)
int launch_flip_add(wsc_ctx *ctx, const float *head, int B, int h, int w, int C, int Cs, float *cam);
// score[b][c] = sigmoid(sum_f mean_hw(feat[2b])[f] * Wc[c][f] + bias[c])
int launch_gap_linear_sigmoid(wsc_ctx *ctx, const bf16_t *feat, const bf16_t *feat_lo, int B, int hw, int F,
                              const float *Wc, const float *bias, int C, float *score, int fmt,
                              int sample_stride);
int launch_bf16_to_f32(wsc_ctx *ctx, const bf16_t *x, const bf16_t *x_lo, size_t n, float *y, int fmt);
int launch_nchw_to_nhwc(wsc_ctx *ctx, const float *x, int N, int C, int HW, bf16_t *y, bf16_t *y_lo, int fmt);
int launch_nhwc_to_nchw(wsc_ctx *ctx, const bf16_t *x, const bf16_t *x_lo, int N, int C, int HW, float *y, int fmt);

// ---- irn_kernels.hip ---------------------------------------------------------------------
int launch_group_norm_stats(wsc_ctx *ctx, const float *x, int N, int H, int W, int C, int G, float eps, void *partial,
                            void *stats);
size_t group_norm_partial_bytes(int N, int H, int W, int G);
int launch_group_norm_apply(wsc_ctx *ctx, const float *x, const void *stats, const float *gamma, const float *beta,
                            int N, int H, int W, int C, int G, int up, int relu, bf16_t *y, bf16_t *y_lo, int Hd, int Wd,
                            int Ctot, int coff, int fmt);
int launch_edge_finish(wsc_ctx *ctx, const float *e, int He, int We, const float *d, int Hd, int Wd, int B, int fh, int fw,
                       float ms0, float ms1, float *edge, float *dp);

