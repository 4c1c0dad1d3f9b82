// The ResNet stem of the f16x3 mode as ONE kernel: conv 7x7 / 2 (3 -> 64) + BatchNorm + ReLU + MaxPool 3x3 / 2 / 1
// (reference: network/resnet50.py:54-64 -- conv1, bn1, relu, maxpool -- the first four modules of every ResNet50 variant).
//
// Why its own kernel.  As an implicit GEMM (conv_igemm.hip, small_cin == 3) the stem spends its time moving bytes, not
// multiplying: the 128 x 224 A tile of a block is a 4 x amplified copy of the 29 KB of input the block really covers
// (2.2 GB of LDS-DMA per 64 x 321^2 batch), the epilogue writes 424 MB of two-plane activations and the pool that follows
// reads them back (ablation in profiles/README.md: 133 of the stem's 207 us remain with neither DMA nor MFMA).  Here a block
// owns a 16 x 16 patch of conv outputs = 7 x 7 pooled outputs:
//   * the 37 x 38 pixel input patch (zero-padded NHWC4, hi and lo planes) is brought to LDS ONCE (28 KB) and the MFMA A
//     fragments are read straight out of it: output pixel (t, u), kernel row r needs the 8 pixels x 4 channels at patch
//     row 2t + r, pixels 2u .. 2u + 7 -- 64 contiguous bytes, i.e. the four 16-byte k-groups of the two MFMA slices.
//     A row pitch of 384 B (a multiple of 128) makes those reads conflict-free without a swizzle;
//   * the weights ([64][7 K-steps x (32 hi | 32 lo)], the small_cin == 3 packing of net.hip make_conv) stream per kernel
//     row through a ring of three 8 KB stages by LDS-DMA, swizzled like the B tiles of conv_igemm.hip;
//   * BN + ReLU are applied to the accumulators, the tile goes to LDS as fp32 (nine tile rows at a time: 39 KB, three blocks
//     per CU) and the 3 x 3 / 2 maximum is taken there: only the pooled planes (1/4 of the pixels) are written.
// Arithmetic: the per-accumulator MFMA sequence (kernel row, slice, lo*hi, hi*lo, hi*hi) is the one conv_igemm.hip runs; the
// two-launch form rounds every conv output to its (hi, lo) pair and the pool re-splits the maximum of those values -- the
// rounding is monotone, so here both roundings are applied to the maximum only and the pooled planes are BIT-IDENTICAL to
// conv_igemm + maxpool_f16x2_kernel (tests/test_gpu_edge.py::test_stem_pool_fused_equals_unfused).
#include "common.h"
#include <type_traits>

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int TP = 16;              // conv outputs per tile side
constexpr int PQ = 7;               // pooled outputs per tile side
constexpr int PROWS = 2 * (TP - 1) + 7;  // 37 patch rows
constexpr int PPIX = 2 * (TP - 1) + 8;   // 38 patch pixels per row (8-pixel window per kernel row)
constexpr int PITCH = 384;          // bytes per patch row in LDS (48 pixels x 8 B; 2 * PITCH % 256 == 0: conflict-free reads)
constexpr int PLANE = 14 * 1024;    // one patch plane = 14 DMA pieces of 1 KiB (37 * 384 = 14208 B used)
constexpr int B_AT = 2 * PLANE;     // weight stages: NSTB x 8 KiB
constexpr int B_BYTES = 8192;
constexpr int NSTB = 3;
constexpr int CT_STRIDE = 68;       // fp32 tile rows of 64 + 4
constexpr int CT_BYTES = 9 * TP * CT_STRIDE * 4; // 39168: nine tile rows at a time
constexpr int LDS_BYTES = B_AT + NSTB * B_BYTES; // 53248: three blocks per CU
static_assert(LDS_BYTES >= CT_BYTES, "the epilogue tile aliases the pipeline's LDS");

struct StemPoolArgs {
    const bf16_t *x, *x_lo; // [N][Hp][Wp][4] zero-padded by 5 (3 of the conv + 2 of the first tile's pool border)
    const bf16_t *w;        // [64][Kw]
    const float *s1, *b1;
    bf16_t *y, *y_lo;       // [N][Hq][Wq][64]
    int Hp, Wp, Kw, Ho, Wo, Hq, Wq, nti, ntj, relu;
    int debug; // A/B builds only (-DWSC_AB_KNOBS): phases switched off for the ablations of profiles/README.md
    unsigned *range; // the ctx's range flag (common.h): raised when a pooled value sits at the half ceiling
};
#ifdef WSC_AB_KNOBS
#define WSC_SDBG(p, bit) ((p).debug & (bit))
#else
#define WSC_SDBG(p, bit) false
#endif

__device__ __forceinline__ int b_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

__global__ __launch_bounds__(256, 3) void stem_pool_kernel(StemPoolArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, kgrp = lane >> 5;
    int tile = blockIdx.x;
    const int tj = tile % p.ntj;
    tile /= p.ntj;
    const int ti = tile % p.nti;
    const int n = tile / p.nti;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

    // ---- the input patch: 28 pieces of 1 KiB (14 per plane), 7 per wave; lanes past a row's 304 bytes re-read its start
    const long long porg = (((long long)n * p.Hp + 28 * ti) * p.Wp + 28 * tj) * 4;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int q = wv + 4 * i, plane = q / 14, pq = q - plane * 14;
        const int o = pq * 1024 + lane * 16;
        const int row = o / PITCH, c16 = (o - row * PITCH) >> 4;
        const bool ok = row < PROWS && c16 < (PPIX * 8) / 16;
        const bf16_t *g = (plane ? p.x_lo : p.x) + porg + (ok ? ((long long)row * p.Wp * 4 + c16 * 8) : 0ll);
        if (!WSC_SDBG(p, 1))
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)(smem + plane * PLANE + pq * 1024), 16, 0, 0);
    }
    // ---- weights of kernel row r: 64 rows x 128 B = 8 pieces, 2 per wave, swizzled on the source side
    const bf16_t *wsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wv * 2 + i) * 8 + (lane >> 3);
        const int ks = (lane & 7) ^ ((row >> 1) & 7);
        wsrc[i] = p.w + (long long)row * p.Kw + ks * 8;
    }
    auto issue_b = [&](int r, int buf) {
        if (WSC_SDBG(p, 2)) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc[i] + r * 64),
                                             (__attribute__((address_space(3))) void *)(smem + B_AT + buf * B_BYTES + (wv * 2 + i) * 1024),
                                             16, 0, 0);
    };
    // three kernel rows' weights are requested up front; row r + 3 takes the stage of row r when every wave is done with it
#pragma unroll
    for (int r = 0; r < NSTB; ++r) issue_b(r, r);

    f32x16_t acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    float sc[2], sh[2]; // BatchNorm scale / shift of the lane's two channels (ni * 32 + l31)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        sc[ni] = p.s1[ni * 32 + l31];
        sh[ni] = p.b1[ni * 32 + l31];
    }

    // fragment addresses.  Wave wv owns tile rows 4 wv .. 4 wv + 3; MFMA tile mi holds rows 4 wv + 2 mi + (l31 >> 4), all 16 columns.
    unsigned offA[2]; // per mi: the hi plane's k-group of slice 0, kernel row 0 (plane, kernel row and slice by immediate)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int tr = 4 * wv + 2 * mi + (l31 >> 4), u = l31 & 15;
        offA[mi] = lds0 + 2 * tr * PITCH + u * 16 + kgrp * 16;
    }
    // B: [slice][hi / lo] of row l31 in stage 0 (row + 32 keeps (row >> 1) & 7, so tile ni = 1 is the same offset + 4096)
    unsigned offBs[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        offBs[s][0] = lds0 + B_AT + b_off(l31, 2 * s + kgrp);
        offBs[s][1] = lds0 + B_AT + b_off(l31, 4 + 2 * s + kgrp);
    }

    auto mfma = [&](const u32x4_t &a, const u32x4_t &b, f32x16_t &c) {
        if (WSC_SDBG(p, 4)) return;
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    };

    // Fragment sets are requested one slice ahead (two register sets); the weights of kernel row r + 1 are checked in -- and the
    // stage of row r handed to row r + 3 -- in the middle of slice (r, 1), once the fragments of that slice are in registers.
    u32x4_t fa[2][2][2], fb[2][2][2]; // [set][hi / lo][mi | ni]
    auto rd = [&](auto set_c, auto r_c, auto s_c) __attribute__((always_inline)) {
        constexpr int set = decltype(set_c)::value, r = decltype(r_c)::value, sl = decltype(s_c)::value, buf = r % NSTB;
        if (WSC_SDBG(p, 8)) return;
        // inline asm: compiler-visible reads would be fenced by an s_waitcnt vmcnt(0) against the DMA in flight (conv_igemm.hip)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[set][1][mi]) : "v"(offA[mi]), "n"(PLANE + r * PITCH + sl * 32) : "memory");
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[set][0][ni]) : "v"(offBs[sl][0]), "n"(buf * B_BYTES + ni * 4096) : "memory");
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[set][0][mi]) : "v"(offA[mi]), "n"(r * PITCH + sl * 32) : "memory");
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[set][1][ni]) : "v"(offBs[sl][1]), "n"(buf * B_BYTES + ni * 4096) : "memory");
    };
    auto products = [&](int set) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) mfma(fa[set][1][mi], fb[set][0][ni], acc[mi][ni]); // lo * hi
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) mfma(fa[set][0][mi], fb[set][1][ni], acc[mi][ni]); // hi * lo
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) mfma(fa[set][0][mi], fb[set][0][ni], acc[mi][ni]); // hi * hi
        __builtin_amdgcn_sched_barrier(0);
    };
    static_assert(NSTB == 3, "the vmcnt counts below are those of a three-stage ring");
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); // the patch and row 0 of the weights (rows 1, 2 in flight)
    __builtin_amdgcn_s_barrier(); // (s_barrier, not __syncthreads(): its fence would wait for every request in flight)
    rd(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    auto krow = [&](auto r_c) __attribute__((always_inline)) {
        constexpr int r = decltype(r_c)::value;
        rd(std::integral_constant<int, 1>{}, r_c, std::integral_constant<int, 1>{});
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); // slice 0 is in registers, slice 1 on its way
        products(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // slice 1 is in registers: this wave is done with stage r % 3
        if (r + 1 < 7) {
            // this wave's pieces of row r + 1 have landed (one later row may stay in flight) ...
            if (r + 2 < 7) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier(); // ... and every other wave's; every wave is done with stage r % 3
            if (r + NSTB < 7) issue_b(r + NSTB, r % NSTB);
            rd(std::integral_constant<int, 0>{}, std::integral_constant<int, (r + 1 < 7 ? r + 1 : 0)>{}, std::integral_constant<int, 0>{});
        }
        products(1);
    };
    krow(std::integral_constant<int, 0>{});
    krow(std::integral_constant<int, 1>{});
    krow(std::integral_constant<int, 2>{});
    krow(std::integral_constant<int, 3>{});
    krow(std::integral_constant<int, 4>{});
    krow(std::integral_constant<int, 5>{});
    krow(std::integral_constant<int, 6>{});
    __syncthreads(); // the LDS is the epilogue's

    // ---- BN + ReLU, tile to LDS as fp32, 3 x 3 / 2 maximum there ---------------------------------------------------------------
    // Two groups of tile rows (0 .. 8 -> pooled rows 0 .. 3; 8 .. 15 -> pooled rows 4 .. 6) so that the fp32 tile needs 39 KB, not
    // 70 KB, and three blocks fit a CU.  The rounding to the (hi, lo) pair is applied to the MAXIMUM only: v -> hi(v) + lo(v) is
    // monotone (a value just below the midpoint of two halves rounds to at most the midpoint, one just above to at least it), so
    // max(round(v_i)) = round(max(v_i)) -- the bits of maxpool_f16x2_kernel over the conv's rounded planes, at 1/5 of the roundings.
    float *ct = reinterpret_cast<float *>(smem);
    if (WSC_SDBG(p, 16)) return;
    const int c0 = 2 * PQ * ti - 1, u0 = 2 * PQ * tj - 1; // conv coordinates of tile position (0, 0)
    const float sat_lo = p.relu ? 0.f : -65504.f;
    // Tile positions outside the conv output (row / column -1 of the first tiles, the far side of the last ones) hold finite
    // garbage; the pool never reads them: a window's positions are CLAMPED to the valid range (a duplicate leaves a maximum
    // alone, and the window's centre is always valid).  Per-element masks cost more instructions than the arithmetic.
    const int tr_lo = c0 < 0 ? -c0 : 0, tr_hi = min(TP - 1, p.Ho - 1 - c0);
    const int u_lo = u0 < 0 ? -u0 : 0, u_hi = min(TP - 1, p.Wo - 1 - u0);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (g) __syncthreads(); // the first group's reads are done
        const int row0 = g ? 8 : 0;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                const int tr = 4 * wv + 2 * mi + rh; // wave-uniform
                if (g ? tr < 8 : tr > 8) continue;
                float *dst = ct + ((tr - row0) * TP + 4 * kgrp) * CT_STRIDE + l31;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr) // accumulator register rh * 8 + rr = tile column (rr & 3) + 8 (rr >> 2) + 4 kgrp
                        dst[((rr & 3) + 8 * (rr >> 2)) * CT_STRIDE + ni * 32] =
                            __builtin_amdgcn_fmed3f(acc[mi][ni][rh * 8 + rr] * sc[ni] + sh[ni], sat_lo, 65504.f);
            }
        __syncthreads();
        const int nk = g ? 3 : 4; // pooled rows of this group
        if (t < nk * PQ * 8) {
            const int k = t / (PQ * 8), rem = t - k * (PQ * 8), j = rem >> 3, c8 = rem & 7;
            const int kk = (g ? 4 : 0) + k;
            const int qi = PQ * ti + kk, qj = PQ * tj + j;
            if (qi < p.Hq && qj < p.Wq) {
                int ro[3], co[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    ro[d] = (min(max(2 * kk + d, tr_lo), tr_hi) - row0) * (TP * CT_STRIDE);
                    co[d] = min(max(2 * j + d, u_lo), u_hi) * CT_STRIDE + c8 * 8;
                }
                f32x4_t b0, b1;
                {
                    const float *src = ct + ro[0] + co[0];
                    b0 = *reinterpret_cast<const f32x4_t *>(src);
                    b1 = *reinterpret_cast<const f32x4_t *>(src + 4);
                }
#pragma unroll
                for (int q = 1; q < 9; q += 2) { // two more window positions per v_max3_f32
                    const float *sa = ct + ro[q / 3] + co[q % 3], *sb = ct + ro[(q + 1) / 3] + co[(q + 1) % 3];
                    const f32x4_t a0 = *reinterpret_cast<const f32x4_t *>(sa), a1 = *reinterpret_cast<const f32x4_t *>(sa + 4);
                    const f32x4_t c0v = *reinterpret_cast<const f32x4_t *>(sb), c1v = *reinterpret_cast<const f32x4_t *>(sb + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        b0[e] = __builtin_fmaxf(__builtin_fmaxf(b0[e], a0[e]), c0v[e]);
                        b1[e] = __builtin_fmaxf(__builtin_fmaxf(b1[e], a1[e]), c1v[e]);
                    }
                }
                const float best[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                uint32_t hw[4], lw[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // the conv's rounding of the maximum, then the pool's own split of that value (a lo part of exactly half an
                    // ulp may move the hi part: same value, other planes -- the planes of the two-launch form are the contract)
                    const _Float16 g0 = (_Float16)best[2 * e], g1 = (_Float16)best[2 * e + 1];
                    const float f0 = (float)g0 + (float)(_Float16)(best[2 * e] - (float)g0);
                    const float f1 = (float)g1 + (float)(_Float16)(best[2 * e + 1] - (float)g1);
                    const _Float16 h0 = (_Float16)f0, h1 = (_Float16)f1;
                    const _Float16 l0 = (_Float16)(f0 - (float)h0), l1 = (_Float16)(f1 - (float)h1);
                    hw[e] = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
                    lw[e] = (uint32_t)__builtin_bit_cast(uint16_t, l0) | ((uint32_t)__builtin_bit_cast(uint16_t, l1) << 16);
                }
                if ((half2_at_ceiling(hw[0]) | half2_at_ceiling(hw[1]) | half2_at_ceiling(hw[2]) | half2_at_ceiling(hw[3])) & 0x80008000u)
                    *p.range = 64u;
                const long long oo = (((long long)n * p.Hq + qi) * p.Wq + qj) * 64 + c8 * 8;
                if (!WSC_SDBG(p, 32)) {
                    *reinterpret_cast<uint4 *>(p.y + oo) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
                    *reinterpret_cast<uint4 *>(p.y_lo + oo) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
                }
            }
        }
    }
}

} // namespace

// Size of the zero-padded NHWC4 input the fused stem reads for an H x W image: the image sits at (5, 5).
void stem_pool_input_dims(int H, int W, int *Hp, int *Wp) {
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    const int Hq = (Ho + 2 - 3) / 2 + 1, Wq = (Wo + 2 - 3) / 2 + 1;
    const int nti = (Hq + PQ - 1) / PQ, ntj = (Wq + PQ - 1) / PQ;
    *Hp = 28 * (nti - 1) + PROWS;
    *Wp = 28 * (ntj - 1) + PPIX;
    if (*Hp < H + 5) *Hp = H + 5;
    if (*Wp < W + 5) *Wp = W + 5;
    *Wp += *Wp & 1; // 16-byte rows
}

// x / x_lo: the padded input (stem_pool_input_dims, launch_nchw_to_nhwc4_pad with pad 5); H, W: the image size.
int launch_stem_pool(wsc_ctx *ctx, const bf16_t *x, const bf16_t *x_lo, int N, int H, int W, const bf16_t *w, int Kw, const float *s1,
                     const float *b1, int relu, bf16_t *y, bf16_t *y_lo) {
    StemPoolArgs a;
    a.x = x; a.x_lo = x_lo; a.w = w; a.s1 = s1; a.b1 = b1; a.y = y; a.y_lo = y_lo;
    stem_pool_input_dims(H, W, &a.Hp, &a.Wp);
    a.Kw = Kw; a.relu = relu; a.debug = 0;
    a.range = ctx->range_dev;
#ifdef WSC_AB_KNOBS
    static const int dbg = [] { const char *e = getenv("WSC_STEM_DEBUG"); return e ? atoi(e) : 0; }();
    a.debug = dbg;
#endif
    a.Ho = (H + 6 - 7) / 2 + 1; a.Wo = (W + 6 - 7) / 2 + 1;
    a.Hq = (a.Ho + 2 - 3) / 2 + 1; a.Wq = (a.Wo + 2 - 3) / 2 + 1;
    a.nti = (a.Hq + PQ - 1) / PQ; a.ntj = (a.Wq + PQ - 1) / PQ;
    WSC_CHECK(x_lo != nullptr && y_lo != nullptr && Kw == 7 * 64, WSC_ERR_INVALID, "stem_pool: f16x3 planes and the 7-row packing only");
    const long long nblk = (long long)N * a.nti * a.ntj;
    WSC_CHECK(nblk < (1ll << 31), WSC_ERR_SHAPE, "stem_pool: too many tiles");
    static bool attr_set[64] = {};
    const int dev = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
    if (!attr_set[dev]) {
        WSC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(stem_pool_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set[dev] = true;
    }
    // algorithmic FLOPs of the conv (the pool adds none): 2 * M * Cout * kh * kw * Cin_real
    WscKernelTimer timer(ctx, WSC_K_CONV_SMALLCIN, 2.0 * N * a.Ho * a.Wo * 64 * 7 * 7 * 3);
    hipLaunchKernelGGL(stem_pool_kernel, dim3((unsigned)nblk), dim3(256), LDS_BYTES, ctx->stream, a);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}
