// conv_igemm.hip -- direct convolution as an implicit GEMM on the gfx950 matrix cores.
//
// Replaces every nn.Conv2d (+ FixedBatchNorm + ReLU + residual add) of the
// reference CAM networks: 03b_irn/net/resnet50.py:17-54,57-108 (Bottleneck /
// ResNet), net/common_cnn.py:128-141 (make_layers: conv -> ReLU -> BatchNorm),
// and the 1x1 CAM head F.conv2d(x, classifier.weight) (resnet50_cam.py:65,
// vgg16_cam.py:48).
//
// GEMM view:  M = N*Ho*Wo output pixels, N = Cout, K = kh*kw*Cin.
//   A[m][k]  gathered on the fly from the NHWC bf16 activation (zero padding by predicate)
//   B[n][k]  packed weights, K contiguous
//   C        fp32 accumulators in registers (v_mfma_f32_32x32x16_bf16 / _f16)
// Block tile 128 x BN x 64, 4 waves (2x2), each wave 64 x BN/2 as 32x32 MFMA tiles.
// LDS tiles are [row][64 k] bf16 (128 B rows) with the 16-byte slot XOR-swizzled by
// (row>>1)&7 so both the ds_write_b128 staging stores and the ds_read_b128 fragment
// loads are bank-conflict free.  Staging is double-buffered, one barrier per K-step, one K-step
// ahead: generic layers use the LDS DMA (global_load_lds_dwordx4: no staging VGPRs, no
// ds_write pass -- the ds_write_b128 path tops out at ~79 B/clk/CU and together with the fragment
// reads made the first version of this kernel LDS-bound at ~300 TFLOP/s); the swizzle is applied
// to the per-lane SOURCE address because a DMA's LDS destination is lane-linear, and padded taps
// read a zero page.  The small-Cin layers (stem) keep global -> register -> LDS staging.
// Epilogue: accumulators go through LDS as an fp32 tile so that each thread owns 8
// consecutive channels of one pixel: folded-BN scale/shift, residual add, ReLU,
// optional post-ReLU affine, then one 16-byte coalesced store.
//
// Split precision (bf16x3): activations and weights carry a second bf16 plane with the
// rounding remainder; the K loop runs three segments (x_hi*w_hi, x_lo*w_hi, x_hi*w_lo)
// into the same accumulators, giving ~2^-16 relative operand error with the same
// MFMA instruction.
#include "common.h"

#include <cstdlib>
#include <type_traits>

// A/B knobs (tile overrides, timing-only ablations that produce wrong results) exist only in builds with -DWSC_AB_KNOBS
// (profiles/conv_ab.sh); the shipped library has one code path per decision.
#ifdef WSC_AB_KNOBS
#define WSC_DBG(p, bit) ((p).debug & (bit))
#else
#define WSC_DBG(p, bit) 0
#endif

namespace {

constexpr int BK = 64;

struct ConvKArgs {
    const bf16_t *x, *x_lo, *w;
    const float *s1, *b1, *s2, *b2;
    const bf16_t *res, *res_lo;
    bf16_t *y, *y_lo;
    float *y_f32;
    int H, W, Cin, Ho, Wo, Cout;
    int ldy;            // row pitch of y / y_lo in elements (Cout, or wider: the output is a channel range of a concatenated tensor)
    int kh, kw, stride, pad, relu;
    int M, HoWo;
    int m_base, m_end; // rows [m_base, m_end) of the M = N*Ho*Wo output rows are this launch's (a layer may be cut in two)
    // exact unsigned division by HoWo / Wo as multiply-high + shifts (Granlund-Montgomery): the prologue
    // decodes 4 output rows per thread and a hardware-less 32-bit division costs ~25 VALU instructions
    unsigned div_howo_mul, div_howo_s1, div_howo_s2, div_wo_mul, div_wo_s1, div_wo_s2;
    int cchunks;     // Cin / 64 (generic mode)
    int ntaps;       // kh * kw
    int ksteps_base; // K-steps of one precision segment
    int nk;          // total K-steps (x3 in split mode)
    int Kw;          // packed weight row length in elements
    int Kbase;       // elements of one weight plane per row
    int ntiles_n, nblocks;
    const bf16_t *zero; // >= 16 bytes of zeros in HBM: source of padded taps for the LDS-DMA path
    int fast;           // host side only: FAST variant of the kernel this launch may use (0 = generic)
    int debug;          // WSC_CONV_DEBUG ablations (timing only, results are wrong): 1 = no DMA after the
                        // first two stages, 2 = no fragment reads / MFMAs
    long long lo_delta; // SPLIT 2: x_lo - x in elements (both planes live in one workspace block)
    // second A source of a 1 x 1 layer (a ResNet stage's first block: conv3 and the projection shortcut as one GEMM, net.hip):
    // channel chunks [0, cc2) come from x (dense [M][ldx]), chunks [cc2, cchunks) from x2 = [N][H2][W2][C2] at pixel
    // (ho * stride2, wo * stride2) of the output pixel.  x2 == nullptr: one source.
    const bf16_t *x2;
    long long lo_delta2; // SPLIT 2: x2_lo - x2
    int cc2, H2, W2, C2, stride2;
    int ldx;            // pixel pitch of x in elements (Cin; the first source's channel count when there are two)
    // LDS input window of a 3 x 3 / stride 1 / pad 1 layer (WPT > 0 variants): positions of the input raster [N][H][W]
    // starting at index (first output pixel of the block) - W - 1; win_npix = N H W (pieces outside the tensor load zeros)
    int win_npix;
    int stem_rows;      // host side only: the padded-input stem form (a K-step = one kernel row of 8 pixels x 4 channels)
    int kw_real;        // host side only: kernel width of the layer (FLOP accounting; kw is 1 in the stem form)
    unsigned *range;    // the ctx's range flag (common.h): raised by an IEEE-half epilogue that stores a value at the half ceiling
};

__device__ __forceinline__ int lds_off(int row, int slot) {
    return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
}

// BM x BN x 64 tile, BM/32 waves (BM/64 along M x 2 along N, 64 x BN/2 per wave), STAGES LDS buffers.
//   128-row tile: 4 waves, 2 stages, 2 blocks per CU.
//   256-row tile: 8 waves, 3 stages (144 KB), 1 block per CU: 0.73x the L2->LDS bytes per FLOP and a
//   prefetch distance of two K-steps, with counted s_waitcnt vmcnt + raw s_barrier so a stage stays in
//   flight across the barrier (a __syncthreads() would drain the LDS DMA every K-step).
//
// FAST (f16, single precision plane, LDS-DMA layers only) removes per-element case handling the common layers do not need:
//   bit 0  epilogue: fp16 output only, every column tile full (Cout % BN == 0), 32-bit output
//          offsets -- scale / shift as float4 loads, no per-channel predicates, residual rows read at a clamped row
//          (no predicate), only the final store is masked by the row bound.  Same arithmetic, operation for operation.
//   bit 1  pointwise: 1x1, stride 1, no padding -- output row m reads activation row m: no (n, ho, wo) decode in the
//          prologue, no tap bounds test per DMA piece (rows past the end are clamped to the last row; their results
//          are never stored).
// One-K-step FAST tiles are compiled for 4 waves per SIMD (<= 128 VGPRs) so that 4 blocks of 34 KB share a CU.
//
// SPLIT: 0 = one precision plane.  1 = two planes, the K loop runs three segments x_hi*w_hi + x_lo*w_hi + x_hi*w_lo, each
// staging its own tiles (bf16x3, and the small-Cin layers of f16x3).  2 = two planes staged ONCE (f16x3, generic layers):
// a K-step is one (tap, 32-channel chunk); its 128-byte LDS row holds the chunk's 32 hi values in 16-byte slots 0-3 and
// its 32 lo values in slots 4-7 (weights packed the same way), so one K-step's 4 + NB DMA pieces feed 2 k-slices x 3
// MFMA products: 1.5x the matrix work per LDS byte of the one-plane kernel instead of 3x its staging.
//
// WPT > 0 (LDS input WINDOW, f16x3 single-staged 128-row tiles, 3 x 3 / stride 1 / pad 1): the A operand of all nine taps of a
// 32-channel chunk is ONE window of the input raster [N][H][W] -- output pixel m of a same-size convolution sits at raster index m,
// so the block's 128 consecutive output pixels read, for tap (r, s), the window positions (m - m0) + r W + s of the window that
// starts at raster index m0 - W - 1 -- staged once per chunk (WPT x 32 positions x [32 hi | 32 lo], same 128-byte rows and XOR
// swizzle as an A tile, so the fragment reads only change their base address per tap) instead of nine per-tap A tiles: 2.7-5.6x
// fewer A bytes through L2 -> LDS.  Round 6: the raster is NOT padded.  Where a tap leaves the image (left / right column, top /
// bottom row) the position holds the neighbouring row's or image's pixel, and the lane reads the window's ZERO ROW instead (a
// 9-bit per-lane tap mask decided once; one v_cndmask per fragment address).  The positions of 32 consecutive output pixels are
// then consecutive for every tap, which is what the swizzle needs to be conflict-free: round 5's zero-padded raster jumped by
// two positions at a row's end and lost 21-30 % of its LDS cycles to bank conflicts there.  It is also smaller: 127 + 2 W + 4
// positions (21 x 21: 173, 41 x 41: 213, 81 x 81: 293) against 236 / 312 / 468.  The next chunk's
// window travels in registers (WPT 16-byte loads per thread, issued at the chunk's first tap) and is written to LDS between
// the chunk's last tap and the next one's first; the weight tiles keep their per-K-step LDS-DMA double buffer.  Same MFMA
// sequence on the same operands as the per-tap kernel: bit-identical results (tests/test_gpu_conv.py).
template <int BM, int BN, int MODE, int SPLIT, int ET, bool GLDS, int STAGES, int WMT = 64, int FAST = 0, int WPT = 0>
__global__ __launch_bounds__(BM * 2, (STAGES == 1 && FAST != 0) ? 4 : 2) void conv_igemm_kernel(ConvKArgs p) {
    constexpr bool FEPI = (FAST & 1) != 0, PW = (FAST & 2) != 0;
    constexpr bool WIN = WPT > 0;
    static_assert(!WIN || (SPLIT == 2 && BM == 128 && WMT == 64 && STAGES == 2 && FAST == 1), "LDS window: single-staged split, 128-row tile");
    static_assert(FAST == 0 || (SPLIT != 1 && ET == 1 && ((GLDS && MODE == 0) || FAST == 1)),
                  "FAST paths: f16, one plane or the single-staged split; the pointwise prologue belongs to the LDS-DMA layers");
    static_assert(SPLIT != 2 || (GLDS && MODE == 0 && STAGES == 2), "single-staged split: LDS-DMA layers, two LDS buffers");
    constexpr int CK = SPLIT == 2 ? 32 : 64; // channels of one K-step
    constexpr int NT = BM * 2;   // threads
    constexpr int NW = BM / 32;  // waves
    // waves are laid out WR (along M) x WC (along N); a wave owns a WMT x WN tile = MI x NI MFMA tiles.
    // WMT = 64 (default): WC = 2, 64 x BN/2 per wave.  WMT = 128 with a 256 x 256 block: 2 x 4 waves of 128 x 64 --
    // half the LDS-DMA bytes per FLOP of the 128 x 128 block and 0.75x the fragment reads per MFMA.
    constexpr int WR = BM / WMT;
    constexpr int WC = NW / WR;
    constexpr int WN = BN / WC;
    constexpr int MI = WMT / 32;
    constexpr int NI = WN / 32;
    static_assert(WR * WC == NW && WN % 32 == 0 && (WMT == 64 || WMT == 128), "wave layout");
    constexpr int NB = BN * 8 / NT; // B 16-byte slots per thread per K-step
    constexpr int A_BYTES = BM * BK * 2;
    constexpr int B_BYTES = BN * BK * 2;
    constexpr int WIN_BYTES = WPT * 4096;                           // WPT x 32 window positions of 128 bytes
    constexpr int A_REGION = WIN ? WIN_BYTES : STAGES * A_BYTES;    // bytes in front of the B buffers (two-buffer LDS-DMA path)
    constexpr int CT_STRIDE = BN + 4;
    static_assert(GLDS || (BM == 128 && STAGES <= 2), "register staging exists for the 128-row tile only");
    static_assert(STAGES >= 1 && STAGES <= 3, "1 (single K-step layers), 2 or 3 LDS buffers");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6); // wave-uniform: LDS DMA destinations stay in SGPRs
    const int wm = wv / WC, wn = wv - wm * WC;

    // XCD-aware, bijective block -> tile map: consecutive tiles (which share the A rows)
    // stay on one XCD's L2.
    int tile;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7;
        const int q = p.nblocks >> 3, r = p.nblocks & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mt = tile / p.ntiles_n;
    const int nt = tile - mt * p.ntiles_n;
    const int m0 = p.m_base + mt * BM;
    const int n0 = nt * BN;

    // ---- per-thread A gather state: 4 rows, one 16-byte slot each -------------
    // Staging ownership.  Register path: thread t stages rows (t>>3) + 32 i, k-slot t&7.
    // LDS-DMA path (GLDS): one global_load_lds_dwordx4 wave-instruction fills 1 KiB of LDS in lane
    // order = 8 tile rows x 8 slot positions, so wave w's i-th instruction owns rows
    // w*32 + i*8 + (lane>>3) and lane position lane&7; the XOR swizzle is applied to the SOURCE
    // k-slot (the LDS destination of a DMA is always lane-linear).
    const int slot = t & 7;
    const int lrow = GLDS ? (wv * 32 + (lane >> 3)) : (t >> 3);
    constexpr int RSTEP = GLDS ? 8 : 32;
    int hb[4], wb[4];
    long long base[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + lrow + RSTEP * i;
        if (PW) {
            const int mc = m < p.m_end ? m : p.m_end - 1;
            hb[i] = 0;
            wb[i] = 0;
            base[i] = (long long)mc * p.ldx;
        } else if (m < p.m_end) {
            const unsigned t1 = __umulhi(p.div_howo_mul, (unsigned)m);
            const int n = (int)((t1 + (((unsigned)m - t1) >> p.div_howo_s1)) >> p.div_howo_s2);
            const int rem = m - n * p.HoWo;
            const unsigned t2 = __umulhi(p.div_wo_mul, (unsigned)rem);
            const int ho = (int)((t2 + (((unsigned)rem - t2) >> p.div_wo_s1)) >> p.div_wo_s2);
            const int wo = rem - ho * p.Wo;
            hb[i] = ho * p.stride - p.pad;
            wb[i] = wo * p.stride - p.pad;
            base[i] = (((long long)n * p.H + hb[i]) * p.W + wb[i]) * (long long)p.ldx;
        } else {
            hb[i] = -(1 << 28);
            wb[i] = 0;
            base[i] = 0;
        }
    }
    // B rows: register path (t>>3) + 32 i; DMA path wave w owns rows w*(BN/NW) + i*8 + (lane>>3)
    const int brow0 = GLDS ? (wv * (BN / NW) + (lane >> 3)) : (t >> 3);
    const bf16_t *wrow[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int r = brow0 + RSTEP * i;
        const int ks = GLDS ? (slot ^ ((r >> 1) & 7)) : slot;
        wrow[i] = p.w + (long long)(n0 + r) * p.Kw + ks * 8;
    }

    // LDS-DMA path: per-row element offset of the lane's 16-byte source slot (row base + swizzled k-slot)
    long long aoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = lrow + RSTEP * i;
        const int ks = slot ^ ((r >> 1) & 7);
        if (SPLIT == 2) aoff[i] = base[i] + (ks & 3) * 8 + ((ks & 4) ? p.lo_delta : 0ll);
        else aoff[i] = base[i] + ks * 8;
    }

    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t ra[4], rb[NB]; // native vectors: with HIP's uint4 struct the B registers were kept in scratch

    // LDS-DMA issue of one K-step's A and B tiles into buffer `buf` (MODE 0 only)
    auto issue_dma = [&](int kt, int buf) {
        int seg = 0, ktl = kt;
        if (SPLIT) {
            seg = kt / p.ksteps_base;
            ktl = kt - seg * p.ksteps_base;
        }
        const bf16_t *src = (SPLIT && seg == 1) ? p.x_lo : p.x;
        const int cc = ktl / p.ntaps;
        const int tap = ktl - cc * p.ntaps;
        const int khi = tap / p.kw;
        const int kwi = tap - khi * p.kw;
        const long long tap_off = ((long long)khi * p.W + kwi) * p.ldx + cc * 64;
        char *sa = smem + buf * A_BYTES + wv * 4096;
        char *sb = smem + STAGES * A_BYTES + buf * B_BYTES + wv * (NB * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lrow + RSTEP * i;
            const int ks = slot ^ ((r >> 1) & 7);
            const int hi = hb[i] + khi, wi = wb[i] + kwi;
            const bool ok = PW || ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W);
            const bf16_t *g = ok ? src + base[i] + tap_off + ks * 8 : p.zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)(sa + i * 1024), 16, 0, 0);
        }
        const int wk = ((SPLIT && seg == 2) ? p.Kbase : 0) + ktl * 64;
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wrow[i] + wk),
                                             (__attribute__((address_space(3))) void *)(sb + i * 1024), 16, 0, 0);
    };

    auto load_tile = [&](int kt) __attribute__((always_inline)) {
        int seg = 0, ktl = kt;
        if (SPLIT) {
            seg = kt / p.ksteps_base;
            ktl = kt - seg * p.ksteps_base;
        }
        const bf16_t *src = (SPLIT && seg == 1) ? p.x_lo : p.x;
        if (MODE == 0) {
            const int cc = ktl / p.ntaps;
            const int tap = ktl - cc * p.ntaps;
            const int khi = tap / p.kw;
            const int kwi = tap - khi * p.kw;
            const long long tap_off = ((long long)khi * p.W + kwi) * p.ldx + cc * 64 + slot * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hi = hb[i] + khi, wi = wb[i] + kwi;
                const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                ra[i] = ok ? *reinterpret_cast<const u32x4_t *>(src + base[i] + tap_off) : u32x4_t{0u, 0u, 0u, 0u};
            }
        } else {
            // small-Cin mode: activation is [N][H][W][4]; one kernel row = 2^MODE slots of
            // 2 pixels (8 bf16) each; weights are packed to match, zero in the padding.
            const int g = ktl * 8 + slot;
            const int khi = g >> MODE;
            const int px = (g & ((1 << MODE) - 1)) * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hi = hb[i] + khi, wi = wb[i] + px;
                const bool okh = khi < p.kh && (unsigned)hi < (unsigned)p.H;
                const uint2 *q = reinterpret_cast<const uint2 *>(src + base[i] + ((long long)khi * p.W + px) * 4);
                uint2 v0 = make_uint2(0, 0), v1 = make_uint2(0, 0);
                if (okh && (unsigned)wi < (unsigned)p.W) v0 = q[0];
                if (okh && (unsigned)(wi + 1) < (unsigned)p.W) v1 = q[1];
                ra[i] = u32x4_t{v0.x, v0.y, v1.x, v1.y};
            }
        }
        const int wk = ((SPLIT && seg == 2) ? p.Kbase : 0) + ktl * 64;
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const u32x4_t *>(wrow[i] + wk);
    };

    auto store_lds = [&](int buf) __attribute__((always_inline)) {
        char *sa = smem + buf * A_BYTES;
        char *sb = smem + STAGES * A_BYTES + buf * B_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4_t *>(sa + lds_off(lrow + 32 * i, slot)) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<u32x4_t *>(sb + lds_off(lrow + 32 * i, slot)) = rb[i];
    };

    f32x16_t acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int l31 = lane & 31;
    const int kgrp = lane >> 5;

    auto mfma = [&](const u32x4_t &a, const u32x4_t &b, f32x16_t &c) {
        if (WSC_DBG(p, 4)) return; // (ablation: no matrix work)
        if (ET == 0)
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                        c, 0, 0, 0);
        else
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b),
                                                       c, 0, 0, 0);
    };
    auto mfma0 = [&](const u32x4_t &a, const u32x4_t &b) -> f32x16_t {
        const f32x16_t z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ET == 0)
            return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), z, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), z, 0, 0, 0);
    };
    // LDS byte address of the (only) LDS object
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

    auto compute = [&](int buf) {
        if (GLDS) {
            // Fragment reads in inline asm.  With compiler-visible ds_reads hipcc puts an
            // s_waitcnt vmcnt(0) in front of the first read of every K-step (it cannot prove that the
            // in-flight LDS DMA of the NEXT stage does not alias the buffer being read), which drains
            // the DMA before the MFMAs start and removes all overlap inside a block.  The reads of
            // k-slice ks+1 are issued before the MFMAs of ks; LDS returns in order, so lgkmcnt(4)
            // means "all but the 4 newest reads have landed".
            const unsigned sa = lds0 + buf * A_BYTES, sb = lds0 + STAGES * A_BYTES + buf * B_BYTES;
            u32x4_t fa[2][MI], fb[2][NI];
            auto rd = [&](int set, int ks) {
                const int sl = ks * 2 + kgrp;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(fa[set][mi]) : "v"(sa + lds_off(wm * WMT + mi * 32 + l31, sl)) : "memory");
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(fb[set][ni]) : "v"(sb + lds_off(wn * WN + ni * 32 + l31, sl)) : "memory");
            };
            rd(0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int set = ks & 1;
                if (ks < 3) {
                    rd(set ^ 1, ks + 1);
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MI + NI) : "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) mfma(fa[set][mi], fb[set][ni], acc[mi][ni]);
            }
            return;
        }
        const char *sa = smem + buf * A_BYTES;
        const char *sb = smem + STAGES * A_BYTES + buf * B_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int sl = ks * 2 + kgrp;
            u32x4_t af[MI], bfr[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[mi] = *reinterpret_cast<const u32x4_t *>(sa + lds_off(wm * WMT + mi * 32 + l31, sl));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                bfr[ni] = *reinterpret_cast<const u32x4_t *>(sb + lds_off(wn * WN + ni * 32 + l31, sl));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) mfma(af[mi], bfr[ni], acc[mi][ni]);
        }
    };

    // ---- main loop ------------------------------------------------------------------
    const int nk = STAGES == 1 ? 1 : p.nk; // (the single-buffer variant is launched for one-K-step layers only)
    if (GLDS && MODE == 0 && STAGES == 3) {
        // three LDS buffers, prefetch distance two K-steps.  Every wave issues PER = 4 + NB DMA
        // instructions per stage; vmcnt(PER) therefore means "everything but the newest stage has
        // landed".  The raw barrier after it makes the other waves' DMA of that stage visible too and
        // doubles as the WAR fence for the buffer the next iteration refills.
        constexpr int PER = 4 + NB;
        issue_dma(0, 0);
        if (nk > 1) {
            issue_dma(1, 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const bool ahead = kt + 2 < nk;
            if (ahead && !WSC_DBG(p, 1)) issue_dma(kt + 2, cur == 0 ? 2 : cur - 1);
            if (!WSC_DBG(p, 2)) compute(cur);
            if (ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            cur = cur == 2 ? 0 : cur + 1;
        }
    } else if (GLDS && MODE == 0) {
        // Two LDS buffers.  The 4 + NB LDS-DMA pieces of K-step kt+1 are issued at the top of K-step kt, before its
        // fragment reads and MFMAs (interleaving one piece after every 1 ... 4 MFMAs was measured: no layer gained,
        // profiles/README.md).  The (segment, tap, channel-chunk) decode of the next K-step is carried incrementally in
        // SGPRs (no per-step integer divisions).  The barrier at the end of a step carries the vmcnt(0) for the pieces
        // in flight.
        int n_seg = 0, n_khi = 0, n_kwi = 0, n_cc = 0, n_ktl = 0; // decode of the next K-step to issue
        const bf16_t *n_src = p.x;
        long long n_tap = 0;
        int n_wk = 0;
        auto prep = [&]() {
            n_src = (SPLIT == 1 && n_seg == 1) ? p.x_lo : p.x;
            n_tap = ((long long)n_khi * p.W + n_kwi) * p.ldx + n_cc * CK;
            if (SPLIT != 1 && p.x2 != nullptr && n_cc >= p.cc2) { // second source (1 x 1 layers only): its own chunk count
                n_src = p.x2;
                n_tap = (n_cc - p.cc2) * CK;
            }
            n_wk = ((SPLIT == 1 && n_seg == 2) ? p.Kbase : 0) + n_ktl * 64;
        };
        // The K loop crosses into the second source once per block: the lanes' row offsets are decoded again for ITS geometry
        // (same registers -- every piece of the first source has been issued by then)
        auto second_source = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + lrow + RSTEP * i;
                const unsigned mc = (unsigned)(m < p.m_end ? m : p.m_end - 1);
                const unsigned t1 = __umulhi(p.div_howo_mul, mc);
                const int n = (int)((t1 + ((mc - t1) >> p.div_howo_s1)) >> p.div_howo_s2);
                const int rem = (int)mc - n * p.HoWo;
                const unsigned t2 = __umulhi(p.div_wo_mul, (unsigned)rem);
                const int ho = (int)((t2 + (((unsigned)rem - t2) >> p.div_wo_s1)) >> p.div_wo_s2);
                const int wo = rem - ho * p.Wo;
                const long long b2 = (((long long)n * p.H2 + ho * p.stride2) * p.W2 + wo * p.stride2) * (long long)p.C2;
                const int r = lrow + RSTEP * i;
                const int ks = slot ^ ((r >> 1) & 7);
                if (SPLIT == 2) aoff[i] = b2 + (ks & 3) * 8 + ((ks & 4) ? p.lo_delta2 : 0ll);
                else aoff[i] = b2 + ks * 8;
            }
        };
        auto advance = [&]() {
            // K order (channel chunk, kh, kw): the taps of one 64-channel chunk are consecutive K-steps
            ++n_ktl;
            if (++n_kwi == p.kw) {
                n_kwi = 0;
                if (++n_khi == p.kh) {
                    n_khi = 0;
                    if (SPLIT != 1 && p.x2 != nullptr && n_cc + 1 == p.cc2) second_source();
                    if (++n_cc == p.cchunks) {
                        n_cc = 0;
                        n_ktl = 0;
                        ++n_seg;
                    }
                }
            }
        };
        auto issue_a = [&](int i, int buf) {
            const int hi = hb[i] + n_khi, wi = wb[i] + n_kwi;
            const bool ok = PW || ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W);
            const bf16_t *g = ok ? n_src + (aoff[i] + n_tap) : p.zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)(smem + buf * A_BYTES + wv * 4096 + i * 1024),
                                             16, 0, 0);
        };
        auto issue_b = [&](int i, int buf) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wrow[i] + n_wk),
                                             (__attribute__((address_space(3))) void *)(smem + A_REGION + buf * B_BYTES + wv * (NB * 1024) + i * 1024),
                                             16, 0, 0);
        };
        // ROLL (single-staged split on the 256 x 256 block): the fragment reads roll across the K-steps, see below
        constexpr bool ROLL = SPLIT == 2 && WMT == 128;
        // ---- LDS window (WIN) state -----------------------------------------------------------------------------------------
        constexpr int WP = WIN ? WPT : 1;
        int woff[WP];          // per thread and piece: source offset in 16-byte units (plane and 8-channel group included)
        unsigned wvalid = 0;   // bit i: piece i is a pixel of the image (else halo / outside: zeros)
        u32x4_t wreg[WP];      // the next chunk's window pieces on their way to LDS
        int qm[MI];            // window position of the lane's rows for tap (0, 0)
        int c_khi = 0, c_kwi = 0, c_cc = 0; // decode of the K-step being computed
        auto win_load = [&](int cc) {
#pragma unroll
            for (int i = 0; i < WP; ++i) {
                const u32x4_t *g = reinterpret_cast<const u32x4_t *>(p.x + ((long long)woff[i] * 8 + cc * 32));
                wreg[i] = ((wvalid >> i) & 1u) ? *g : u32x4_t{0u, 0u, 0u, 0u};
            }
        };
        auto win_store = [&]() {
#pragma unroll
            for (int i = 0; i < WP; ++i) {
                const int g = i * NT + t;
                *reinterpret_cast<u32x4_t *>(smem + lds_off(g >> 3, g & 7)) = wreg[i];
            }
        };
        unsigned tapmask[MI]; // bit (3 r + s): tap (r, s) of the lane's output pixel lies outside the image
        constexpr int WIN_ZERO = WPT * 32 - 1; // the window's last position: a row of zeros (no piece of the tensor lands there)
        if constexpr (WIN) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int mr = m0 + wm * WMT + mi * 32 + l31;
                const int m = mr < p.m_end ? mr : p.m_end - 1; // (rows past the end: a valid position, results never stored)
                const unsigned t1 = __umulhi(p.div_howo_mul, (unsigned)m);
                const int n = (int)((t1 + (((unsigned)m - t1) >> p.div_howo_s1)) >> p.div_howo_s2);
                const int rem = m - n * p.HoWo;
                const unsigned t2 = __umulhi(p.div_wo_mul, (unsigned)rem);
                const int ho = (int)((t2 + (((unsigned)rem - t2) >> p.div_wo_s1)) >> p.div_wo_s2);
                const int wo = rem - ho * p.Wo;
                qm[mi] = m - m0; // window position of the pixel's tap (0, 0)
                const unsigned left = wo == 0 ? 0x49u : 0u, right = wo == p.W - 1 ? 0x124u : 0u;   // s = 0: bits 0, 3, 6; s = 2: 2, 5, 8
                const unsigned top = ho == 0 ? 0x7u : 0u, bottom = ho == p.H - 1 ? 0x1c0u : 0u;     // r = 0: bits 0-2; r = 2: bits 6-8
                tapmask[mi] = left | right | top | bottom;
            }
            const int u0 = m0 - p.W - 1; // raster index of window position 0
#pragma unroll
            for (int i = 0; i < WP; ++i) {
                const int g = i * NT + t;
                const int wpos = g >> 3, sl = g & 7;
                const int u = u0 + wpos;
                const bool ok = u >= 0 && u < p.win_npix && wpos != WIN_ZERO;
                const long long e = (long long)u * (long long)p.ldx + (sl & 3) * 8 + ((sl & 4) ? p.lo_delta : 0ll);
                woff[i] = ok ? (int)(e >> 3) : 0;
                wvalid |= ok ? (1u << i) : 0u;
            }
            win_load(0);
        }
        prep();
        if constexpr (!WIN) {
#pragma unroll
            for (int i = 0; i < 4; ++i) issue_a(i, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) issue_b(i, 0);
        advance();
        if constexpr (WIN) win_store();
        if (ROLL && nk > 1) { // two stages ahead; the first one has landed when all but the newest 4 + NB pieces have
            prep();
#pragma unroll
            for (int i = 0; i < 4; ++i) issue_a(i, 1);
#pragma unroll
            for (int i = 0; i < NB; ++i) issue_b(i, 1);
            advance();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + NB) : "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            __syncthreads();
        }
        // Fragment read addresses: lds_off(row + 32 mi, sl) = lds_off(row, sl) + 4096 mi (the swizzle term (row >> 1) & 7 does not
        // see multiples of 32), so one VGPR per k-slice and operand serves every mi / ni through the instruction's immediate
        // offset, which also carries the buffer (the K loop is unrolled by two so that `cur` is a constant): no address VALU
        // inside the loop (it was 20 v_add per K-step).
        unsigned offA[4], offB[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            offA[ks] = lds0 + lds_off(wm * WMT + l31, ks * 2 + kgrp);
            offB[ks] = lds0 + A_REGION + lds_off(wn * WN + l31, ks * 2 + kgrp);
        }
        auto kstep = [&](auto cur_c, int kt) __attribute__((always_inline)) {
            constexpr int cur = decltype(cur_c)::value;
            const unsigned(&oA)[4] = offA, (&oB)[4] = offB; // (named here: the nested lambda below must not be the first use)
            const bool more = kt + 1 < nk && !WSC_DBG(p, 1);
            if (more) {
                prep();
                if (SPLIT != 2) { // (the single-staged split spreads its pieces over the MFMAs, below; spreading them in the
                                  // one-plane kernels too was measured in round 4 and lost 3-4 %: ResNet50 f16 3.17 -> 3.29 ms)
#pragma unroll
                    for (int i = 0; i < 4; ++i) issue_a(i, cur ^ 1);
#pragma unroll
                    for (int i = 0; i < NB; ++i) issue_b(i, cur ^ 1);
                }
            }
            if constexpr (SPLIT == 2) {
                // slot pairs 0, 1 = hi halves of k-slices 0, 1; slot pairs 2, 3 = their lo halves
                u32x4_t fah[2][MI], fal[2][MI], fbh[2][NI], fbl[2][NI];
                // the next K-step's 4 + NB DMA pieces go out one at a time between this K-step's 6 MI NI MFMAs (issued back to
                // back at the top of the step they queue in front of the texture addresser and hold the wave in its issue stage)
                constexpr int NAP = WIN ? 0 : 4; // per-K-step A pieces (none with the LDS window)
                constexpr int EVERY = (6 * MI * NI) / (NAP + NB);
                static_assert(EVERY >= 1, "a K-step has room for every DMA piece");
                auto piece = [&](int j) {
                    if (!more || j % EVERY != 0) return;
                    const int q = j / EVERY;
                    if (q < NAP) {
                        if (!(WSC_DBG(p, 32) && (n_khi | n_kwi) != 0)) issue_a(q, cur ^ 1); // (ablation 32: A staged for the first tap of a chunk only)
                    } else if (q < NAP + NB) issue_b(q - NAP, cur ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                };
                // LDS window: the first tap of a chunk sends for the NEXT chunk's window (it has the whole chunk to arrive); the
                // fragment addresses of this tap are the lane's window positions + r (W + 2) + s, swizzled by the position
                unsigned aw[MI][4];
                if constexpr (WIN) {
                    if ((c_khi | c_kwi) == 0 && c_cc + 1 < p.cchunks) win_load(c_cc + 1);
                    const int toff = c_khi * p.W + c_kwi;
                    const unsigned tbit = 1u << (c_khi * 3 + c_kwi);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const int pos = (tapmask[mi] & tbit) ? WIN_ZERO : qm[mi] + toff;
                        const unsigned b0 = lds0 + (unsigned)pos * 128u, sw = (unsigned)(pos >> 1) & 7u;
#pragma unroll
                        for (int pr = 0; pr < 4; ++pr) aw[mi][pr] = b0 + ((((unsigned)(pr * 2 + kgrp)) ^ sw) << 4);
                    }
                }
                auto rd = [&](int set, int sl) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        if constexpr (WIN) {
                            asm volatile("ds_read_b128 %0, %1" : "=v"(fah[set][mi]) : "v"(aw[mi][sl]) : "memory");
                            asm volatile("ds_read_b128 %0, %1" : "=v"(fal[set][mi]) : "v"(aw[mi][2 + sl]) : "memory");
                        } else {
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fah[set][mi]) : "v"(oA[sl]), "n"(cur * A_BYTES + mi * 4096) : "memory");
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fal[set][mi]) : "v"(oA[2 + sl]), "n"(cur * A_BYTES + mi * 4096) : "memory");
                        }
                    }
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fbh[set][ni]) : "v"(oB[sl]), "n"(cur * B_BYTES + ni * 4096) : "memory");
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fbl[set][ni]) : "v"(oB[2 + sl]), "n"(cur * B_BYTES + ni * 4096) : "memory");
                    }
                };
                rd(0, 0);
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    if (sl == 0) {
                        rd(1, 1);
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (MI + NI)) : "memory");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // the two correction products first, then the main one; each product runs over all MI x NI accumulators
                    // before the next touches them again (an accumulator's MFMAs are MI * NI issue slots apart)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            mfma(fal[sl][mi], fbh[sl][ni], acc[mi][ni]);
                            piece((sl * 3 + 0) * MI * NI + mi * NI + ni);
                        }
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            mfma(fah[sl][mi], fbl[sl][ni], acc[mi][ni]);
                            piece((sl * 3 + 1) * MI * NI + mi * NI + ni);
                        }
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            mfma(fah[sl][mi], fbh[sl][ni], acc[mi][ni]);
                            piece((sl * 3 + 2) * MI * NI + mi * NI + ni);
                        }
                }
            } else if (STAGES == 1 || !WSC_DBG(p, 2)) {
                u32x4_t fa[2][MI], fb[2][NI];

                auto rd = [&](int set, int ks) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[set][mi]) : "v"(oA[ks]), "n"(cur * A_BYTES + mi * 4096) : "memory");
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[set][ni]) : "v"(oB[ks]), "n"(cur * B_BYTES + ni * 4096) : "memory");
                };
                rd(0, 0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int set = ks & 1;
                    if (ks < 3) {
                        rd(set ^ 1, ks + 1);
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MI + NI) : "memory");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            // one-K-step layers: the first k-slice takes C = 0 as an inline constant (no 64 v_mov per wave)
                            if (STAGES == 1 && ks == 0) acc[mi][ni] = mfma0(fa[set][mi], fb[set][ni]);
                            else mfma(fa[set][mi], fb[set][ni], acc[mi][ni]);
                        }
                }
            }
            if (more) advance();
            __syncthreads();
            if constexpr (WIN) {
                // every read of the window has landed (barrier above).  Was this the chunk's last tap?  Then the next chunk's
                // window, which has been travelling in registers since the chunk's first tap, replaces it.
                if (++c_kwi == p.kw) {
                    c_kwi = 0;
                    if (++c_khi == p.kh) {
                        c_khi = 0;
                        ++c_cc;
                        if (more) {
                            win_store();
                            __syncthreads();
                        }
                    }
                }
            }
        };
        static_assert(A_BYTES + (MI - 1) * 4096 < 65536 && B_BYTES + (NI - 1) * 4096 < 65536, "ds_read immediate offset range");
        if constexpr (ROLL) {
            // 128 x 64 per wave: 128 accumulators leave room for ONE set of fragments (lo(A), hi(A): MI, lo(B), hi(B): NI
            // registers x 4), so a slice's operands are re-requested for the NEXT slice as soon as the last MFMA that reads
            // them has been issued -- lo(A) after the first product, lo(B) after the second, the hi pair after the third --
            // and the requests roll on across the K-steps: the ONE barrier of a K-step sits between its two k-slices, when
            // every read of its LDS buffer has been issued.  Behind it the buffer is refilled (K-step kt + 2) and the second
            // slice's MFMAs cover the first reads of the next buffer, whose DMA was issued a whole K-step earlier.
            // (With the barrier at the end of the K-step all eight waves waited together for 12 KB of fragments each.)
            // LDS returns in order: lgkmcnt(n) = "all but the n newest reads have landed".
            u32x4_t fah[MI], fal[MI], fbh[NI], fbl[NI];
            // `dma` (the K-step's second slice): the 4 + NB LDS-DMA pieces of K-step kt + 2 go out one at a time between the MFMAs
            // (one piece per three MFMAs).  Issued back to back behind the barrier, the 64 pieces of the eight waves queued
            // in front of the texture addresser and every wave sat in its issue stage meanwhile -- no wave fed the matrix
            // pipe (DMA and MFMA time ADDED up: ablations in profiles/README.md).
            auto slice = [&](auto buf_c, auto pair_c, bool more, bool dma, int dbuf) __attribute__((always_inline)) {
                constexpr int nbuf = decltype(buf_c)::value, npair = decltype(pair_c)::value; // where the NEXT slice lives
                const unsigned(&oA)[4] = offA, (&oB)[4] = offB;
                auto piece = [&](int j) { // after MFMA j of the slice (0 .. 3 MI NI - 1)
                    if (!dma || j % 3 != 0) return;
                    const int q = j / 3;
                    if (q < 4) issue_a(q, dbuf);
                    else if (q < 4 + NB) issue_b(q - 4, dbuf);
                    __builtin_amdgcn_sched_barrier(0);
                };
                static_assert(3 * MI * NI >= 3 * (4 + NB), "a slice has room for every DMA piece");
                auto rdA = [&](u32x4_t(&f)[MI], int pair) {
                    if (WSC_DBG(p, 16)) return; // (ablation: no fragment reads)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[mi]) : "v"(oA[pair]), "n"(nbuf * A_BYTES + mi * 4096) : "memory");
                };
                auto rdB = [&](u32x4_t(&f)[NI], int pair) {
                    if (WSC_DBG(p, 16)) return;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[ni]) : "v"(oB[pair]), "n"(nbuf * B_BYTES + ni * 4096) : "memory");
                };
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MI) : "memory"); // lo(A), lo(B), hi(B) are here, hi(A) may be in flight
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        mfma(fal[mi], fbh[ni], acc[mi][ni]);
                        piece(mi * NI + ni);
                    }
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    rdA(fal, 2 + npair);
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MI) : "memory"); // hi(A)
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        mfma(fah[mi], fbl[ni], acc[mi][ni]);
                        piece(MI * NI + mi * NI + ni);
                    }
                __builtin_amdgcn_sched_barrier(0);
                if (more) rdB(fbl, 2 + npair);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        mfma(fah[mi], fbh[ni], acc[mi][ni]);
                        piece(2 * MI * NI + mi * NI + ni);
                    }
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    rdB(fbh, npair);
                    rdA(fah, npair);
                }
            };
            {   // first slice of K-step 0, in the order the rolling requests have: lo(A), lo(B), hi(B), hi(A)
                const unsigned(&oA)[4] = offA, (&oB)[4] = offB;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fal[mi]) : "v"(oA[2]), "n"(mi * 4096) : "memory");
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fbl[ni]) : "v"(oB[2]), "n"(ni * 4096) : "memory");
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fbh[ni]) : "v"(oB[0]), "n"(ni * 4096) : "memory");
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fah[mi]) : "v"(oA[0]), "n"(mi * 4096) : "memory");
            }
            auto rstep = [&](auto cur_c, int kt) __attribute__((always_inline)) {
                constexpr int cur = decltype(cur_c)::value;
                slice(std::integral_constant<int, cur>{}, std::integral_constant<int, 1>{}, true, false, 0); // slice 0; requests slice 1
                if (WSC_DBG(p, 8)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); // (ablation: no barrier)
                else
                __syncthreads(); // every read of buffer `cur` has landed, for every wave; so has the DMA of K-step kt + 1
                const bool refill = kt + 2 < nk && !WSC_DBG(p, 1);
                if (refill) prep();
                // slice 1; requests the next K-step's slice 0 and refills buffer `cur`
                slice(std::integral_constant<int, cur ^ 1>{}, std::integral_constant<int, 0>{}, kt + 1 < nk, refill, cur);
                if (refill) advance();
            };
            for (int kt = 0; kt < nk; kt += 2) {
                rstep(std::integral_constant<int, 0>{}, kt);
                if (kt + 1 < nk) rstep(std::integral_constant<int, 1>{}, kt + 1);
            }
            __syncthreads(); // the epilogue reuses the LDS
        } else
        for (int kt = 0; kt < nk; kt += 2) {
            kstep(std::integral_constant<int, 0>{}, kt);
            if (kt + 1 < nk) kstep(std::integral_constant<int, 1>{}, kt + 1);
        }
    } else {
        load_tile(0);
        store_lds(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            const bool more = kt + 1 < nk;
            if (more) load_tile(kt + 1);
            compute(cur);
            if (more) store_lds(cur ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue --------------------------------------------------------------------
    // Thread (r0, c8) finishes 8 consecutive channels of rows r0 + pass*RPP.  The residual rows are
    // requested FIRST (they are the only long-latency loads of the epilogue) so that they travel while
    // the accumulators go through LDS; the per-pass work is fully unrolled so the 16-byte stores of all
    // passes are issued back to back.
    constexpr int TPR = BN / 8;   // threads per row
    constexpr int RPP = NT / TPR; // rows per pass
    constexpr int NPASS = BM / RPP;
    const int c8 = t % TPR;
    const int r0 = t / TPR;
    const int c = n0 + c8 * 8;
    const bool cok = c < p.Cout;
    const bool full = c + 8 <= p.Cout;
    const bool has_res = p.res != nullptr && full;
    // (the 256 x 256 tile requests them per 128-row group: 16 passes of residual rows next to 128 accumulator
    // registers would not fit the 256-VGPR budget of 2 waves per SIMD)
    // (the one-K-step FAST tile, compiled for 128 VGPRs, requests them per 64-row group as well)
    constexpr bool PRE_ALL = WMT == 64 && !(STAGES == 1 && FEPI);
    constexpr int NHALF_R = STAGES == 1 ? BM / 64 : (WMT == 128 ? BM / 128 : 1);
    constexpr int NRES = PRE_ALL ? NPASS : NPASS / NHALF_R;
    uint4 rres[NRES], rres_lo[NRES];
    auto fetch_res = [&](int first_pass) {
#pragma unroll
        for (int i = 0; i < NRES; ++i) {
            const int m = m0 + (first_pass + i) * RPP + r0;
            rres[i] = make_uint4(0, 0, 0, 0);
            rres_lo[i] = make_uint4(0, 0, 0, 0);
            if (FEPI) {
                if (has_res) {
                    const int mc = m < p.m_end ? m : p.m_end - 1; // clamped: always a valid row, never stored past the end
                    rres[i] = *reinterpret_cast<const uint4 *>(p.res + ((unsigned)mc * (unsigned)p.Cout + (unsigned)c));
                    if (SPLIT) rres_lo[i] = *reinterpret_cast<const uint4 *>(p.res_lo + ((unsigned)mc * (unsigned)p.Cout + (unsigned)c));
                }
            } else if (has_res && m < p.m_end) {
                const long long o = (long long)m * p.Cout + c;
                rres[i] = *reinterpret_cast<const uint4 *>(p.res + o);
                if (SPLIT) rres_lo[i] = *reinterpret_cast<const uint4 *>(p.res_lo + o);
            }
        }
    };
    if (PRE_ALL) fetch_res(0);
    const bool fpost = FEPI && p.s2 != nullptr;
    const float sat_lo = (p.relu && !fpost) ? 0.f : -65504.f;
    float s1[8], b1[8], s2[FEPI ? 1 : 8], b2[FEPI ? 1 : 8];
    const bool post = !FEPI && p.s2 != nullptr;
    if (FEPI) {
        const f32x4_t sa = *reinterpret_cast<const f32x4_t *>(p.s1 + c), sb = *reinterpret_cast<const f32x4_t *>(p.s1 + c + 4);
        const f32x4_t ba = *reinterpret_cast<const f32x4_t *>(p.b1 + c), bb = *reinterpret_cast<const f32x4_t *>(p.b1 + c + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s1[j] = sa[j];
            s1[4 + j] = sb[j];
            b1[j] = ba[j];
            b1[4 + j] = bb[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s1[j] = cok ? p.s1[c + j] : 0.f;
            b1[j] = cok ? p.b1[c + j] : 0.f;
            s2[j] = (cok && post) ? p.s2[c + j] : 1.f;
            b2[j] = (cok && post) ? p.b2[c + j] : 0.f;
        }
    }

    // fp32 transpose through LDS.  Multi-K-step variants move the whole BM x BN tile at once (one barrier).
    // The single-buffer variant (one-K-step layers: 1x1 convs with 64 input channels, bound by bytes in
    // flight, not by MFMA) moves it 64 rows at a time so that a block needs 34 KB of LDS instead of 68 KB
    // and a third block fits the CU: layer1.conv3 of the ResNet50 stack went from 190-205 us to 128 us.
    // (Splitting everywhere costs more than it gains: stem 179 -> 219 us, layer2.0.conv1 88 -> 105 us.)
    float *ct = reinterpret_cast<float *>(smem);
    // The 256 x 256 tile goes in two 128-row groups (130 KB of fp32; the whole tile would need 260 KB).
    constexpr int NHALF = STAGES == 1 ? BM / 64 : (WMT == 128 ? BM / 128 : 1);
    constexpr int GR = BM / NHALF;     // tile rows per group
    constexpr int PPH = NPASS / NHALF; // passes per group
    static_assert(NPASS % NHALF == 0 && PPH * RPP == GR && GR % WMT == 0, "epilogue pass layout");
    const int grp = (wm * WMT) / GR, roff = wm * WMT - grp * GR;
    unsigned ovf = 0u;
#pragma unroll
    for (int half = 0; half < NHALF; ++half) {
        if (half > 0) __syncthreads(); // the previous group's reads are done
        if (!PRE_ALL) fetch_res(half * PPH);
        if (grp == half) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = roff + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kgrp;
                        const int col = wn * WN + ni * 32 + l31;
                        ct[row * CT_STRIDE + col] = acc[mi][ni][r];
                    }
        }
        __syncthreads();
        if (!cok) continue;
#pragma unroll
        for (int pp = 0; pp < PPH; ++pp) {
            const int pass = half * PPH + pp;
            const int lrow_e = pp * RPP + r0; // row inside the group
            const int m = m0 + half * GR + lrow_e;
            if (FEPI) {
                float v[8];
                const f32x4_t q0 = *reinterpret_cast<const f32x4_t *>(ct + lrow_e * CT_STRIDE + c8 * 8);
                const f32x4_t q1 = *reinterpret_cast<const f32x4_t *>(ct + lrow_e * CT_STRIDE + c8 * 8 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = q0[j] * s1[j] + b1[j];
                    v[4 + j] = q1[j] * s1[4 + j] + b1[4 + j];
                }
                if (has_res) {
                    const int ri = PRE_ALL ? pass : pp;
                    const uint32_t rw[4] = {rres[ri].x, rres[ri].y, rres[ri].z, rres[ri].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[2 * j] += f16_to_f32((bf16_t)(rw[j] & 0xffffu));
                        v[2 * j + 1] += f16_to_f32((bf16_t)(rw[j] >> 16));
                    }
                    if (SPLIT) {
                        const uint32_t lw[4] = {rres_lo[ri].x, rres_lo[ri].y, rres_lo[ri].z, rres_lo[ri].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[2 * j] += f16_to_f32((bf16_t)(lw[j] & 0xffffu));
                            v[2 * j + 1] += f16_to_f32((bf16_t)(lw[j] >> 16));
                        }
                    }
                }
                // ReLU and the saturation of f32_to_f16 (+-65504, no infinities) in ONE median: med3(v, lo, 65504) with
                // lo = 0 under ReLU (max(v, 0) then min(., 65504)) and -65504 otherwise -- same values as fmaxf + the clamp,
                // a quarter of the instructions (each fmaxf / fminf costs a canonicalising v_max on top)
                if (fpost) {
                    // conv -> ReLU -> BatchNorm layers (net/common_cnn.py make_layers): ReLU, then the affine; its scale / shift
                    // are re-read per pass (L1 hits) instead of holding 16 more VGPRs through the K loop
                    const f32x4_t s2a = *reinterpret_cast<const f32x4_t *>(p.s2 + c), s2b = *reinterpret_cast<const f32x4_t *>(p.s2 + c + 4);
                    const f32x4_t b2a = *reinterpret_cast<const f32x4_t *>(p.b2 + c), b2b = *reinterpret_cast<const f32x4_t *>(p.b2 + c + 4);
                    const float rl = p.relu ? 0.f : -3.0e38f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = __builtin_amdgcn_fmed3f(v[j], rl, 3.0e38f) * s2a[j] + b2a[j];
                        v[4 + j] = __builtin_amdgcn_fmed3f(v[4 + j], rl, 3.0e38f) * s2b[j] + b2b[j];
                    }
                }
                uint32_t hw[4], lw[SPLIT ? 4 : 1];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float c0 = __builtin_amdgcn_fmed3f(v[2 * j], sat_lo, 65504.f), c1 = __builtin_amdgcn_fmed3f(v[2 * j + 1], sat_lo, 65504.f);
                    const _Float16 h0 = (_Float16)c0;
                    const _Float16 h1 = (_Float16)c1;
                    hw[j] = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
                    if (SPLIT) {
                        // lo = half(value - hi): |value - hi| <= 2^-11 |value|, exact in fp32, then rounded to a (sub)normal half
                        const _Float16 l0 = (_Float16)(c0 - (float)h0), l1 = (_Float16)(c1 - (float)h1);
                        lw[j] = (uint32_t)__builtin_bit_cast(uint16_t, l0) | ((uint32_t)__builtin_bit_cast(uint16_t, l1) << 16);
                    }
                }
                if (m < p.m_end) {
                    // range guard (wsc_ctx_range_status): a stored half at the ceiling = a value the clamp above cut (or NaN)
                    ovf |= half2_at_ceiling(hw[0]) | half2_at_ceiling(hw[1]) | half2_at_ceiling(hw[2]) | half2_at_ceiling(hw[3]);
                    *reinterpret_cast<uint4 *>(p.y + ((unsigned)m * (unsigned)p.ldy + (unsigned)c)) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
                    if (SPLIT) *reinterpret_cast<uint4 *>(p.y_lo + ((unsigned)m * (unsigned)p.ldy + (unsigned)c)) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
                }
                continue;
            }
            if (m < p.m_end) {
                float v[8];
                const f32x4_t q0 = *reinterpret_cast<const f32x4_t *>(ct + lrow_e * CT_STRIDE + c8 * 8);
                const f32x4_t q1 = *reinterpret_cast<const f32x4_t *>(ct + lrow_e * CT_STRIDE + c8 * 8 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = q0[j];
                    v[4 + j] = q1[j];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = v[j] * s1[j] + b1[j];
                const long long o = (long long)m * p.Cout + c;
                if (has_res) {
                    const int ri = PRE_ALL ? pass : pp;
                    const uint32_t rw[4] = {rres[ri].x, rres[ri].y, rres[ri].z, rres[ri].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[2 * j] += h16_to_f32((bf16_t)(rw[j] & 0xffffu), ET);
                        v[2 * j + 1] += h16_to_f32((bf16_t)(rw[j] >> 16), ET);
                    }
                    if (SPLIT) {
                        const uint32_t lw[4] = {rres_lo[ri].x, rres_lo[ri].y, rres_lo[ri].z, rres_lo[ri].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[2 * j] += h16_to_f32((bf16_t)(lw[j] & 0xffffu), ET);
                            v[2 * j + 1] += h16_to_f32((bf16_t)(lw[j] >> 16), ET);
                        }
                    }
                }
                if (p.relu) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
                }
                if (post) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = v[j] * s2[j] + b2[j];
                }
                if (p.y_f32 != nullptr) {
                    if (full && (p.Cout & 3) == 0) {
                        f32x4_t o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                        *reinterpret_cast<f32x4_t *>(p.y_f32 + o) = o0;
                        *reinterpret_cast<f32x4_t *>(p.y_f32 + o + 4) = o1;
                    } else {
                        for (int j = 0; j < 8 && c + j < p.Cout; ++j) p.y_f32[o + j] = v[j];
                    }
                }
                if (p.y != nullptr && full) {
                    uint32_t hw[4], lw[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bf16_t h0 = f32_to_h16(v[2 * j], ET), h1 = f32_to_h16(v[2 * j + 1], ET);
                        hw[j] = (uint32_t)h0 | ((uint32_t)h1 << 16);
                        if (SPLIT) {
                            // the lo plane has the hi plane's format; a value beyond the half range saturates as a whole (hi = +-65504, lo = 0)
                            const float c0 = ET ? fminf(fmaxf(v[2 * j], -65504.f), 65504.f) : v[2 * j];
                            const float c1 = ET ? fminf(fmaxf(v[2 * j + 1], -65504.f), 65504.f) : v[2 * j + 1];
                            const bf16_t l0 = f32_to_h16(c0 - h16_to_f32(h0, ET), ET);
                            const bf16_t l1 = f32_to_h16(c1 - h16_to_f32(h1, ET), ET);
                            lw[j] = (uint32_t)l0 | ((uint32_t)l1 << 16);
                        }
                    }
                    if (ET) ovf |= half2_at_ceiling(hw[0]) | half2_at_ceiling(hw[1]) | half2_at_ceiling(hw[2]) | half2_at_ceiling(hw[3]);
                    const long long oy = (long long)m * p.ldy + c;
                    *reinterpret_cast<uint4 *>(p.y + oy) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
                    if (SPLIT) *reinterpret_cast<uint4 *>(p.y_lo + oy) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
                }
            }
        }
    }
    if (ET && (ovf & 0x80008000u)) *p.range = (unsigned)p.Cout;
}

template <int BM, int BN, int MODE, int SPLIT, int ET, int STAGES, int WMT = 64, int FAST = 0, int WPT = 0>
int launch_stages(wsc_ctx *ctx, const ConvKArgs &a) {
    constexpr bool GLDS = MODE == 0; // LDS-DMA staging for every generic layer; small-Cin layers stage via registers
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int PIPE = WPT > 0 ? WPT * 4096 + STAGES * B_BYTES : STAGES * (A_BYTES + B_BYTES);
    // fp32 transpose: 64-row groups in the single-buffer variant, 128-row groups in the 256 x 256 tile
    constexpr int EPI = (STAGES == 1 ? 64 : (WMT == 128 ? 128 : BM)) * (BN + 4) * 4;
    constexpr int LDS = PIPE > EPI ? PIPE : EPI;
    static_assert(LDS <= 160 * 1024, "LDS budget of a CU");
    // the attribute belongs to the (function, device) pair: a process may hold contexts on several GPUs
    static bool attr_set[64] = {};
    auto kern = conv_igemm_kernel<BM, BN, MODE, SPLIT, ET, GLDS, STAGES, WMT, FAST, WPT>;
    const int dev = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
    int lds_req = LDS;
#ifdef WSC_AB_KNOBS
    // A/B: pad the LDS request of the multi-K-step tiles (caps the blocks per CU: room for another stream's workgroups)
    static const int lds_pad = [] { const char *e = getenv("WSC_CONV_LDS_PAD"); return e ? atoi(e) : 0; }();
    if (STAGES == 2 && lds_pad > lds_req && lds_pad <= 160 * 1024) lds_req = lds_pad;
#endif
    if (!attr_set[dev] || ctx->device != dev) {
        WSC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds_req > LDS ? 160 * 1024 : LDS));
        attr_set[dev] = true;
    }
    // algorithmic FLOPs: 2 * M * Cout * (kh*kw*Cin_real), x1 regardless of the precision mode
    const double flops = 2.0 * (a.m_end - a.m_base) * a.Cout *
                         ((MODE == 0 && !a.stem_rows) ? (double)a.kh * a.kw * a.Cin : (double)a.kh * (a.stem_rows ? a.kw_real : a.kw) * 3);
    WscKernelTimer timer(ctx, (MODE != 0 || a.stem_rows) ? WSC_K_CONV_SMALLCIN : (BM == 256 ? WSC_K_CONV256 : (BN == 128 ? WSC_K_CONV128 : WSC_K_CONV64)), flops);
    hipLaunchKernelGGL(kern, dim3(a.nblocks), dim3(BM * 2), lds_req, ctx->stream, a);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

// 3 x 3 / stride 1 / pad 1 layer of the f16x3 mode on the LDS-window variant (WPT x 32 window positions)
template <int WPT, int BN = 128>
int launch_window(wsc_ctx *ctx, const ConvKArgs &a) {
    return launch_stages<128, BN, 0, 2, 1, 2, 64, 1, WPT>(ctx, a);
}

template <int BM, int BN, int MODE, int SPLIT, int ET, int FAST = 0>
int launch_variant(wsc_ctx *ctx, const ConvKArgs &a) {
    if constexpr (BM == 256 && BN == 256) {
        return launch_stages<BM, BN, MODE, SPLIT, ET, 2, 128, FAST>(ctx, a);
    } else if constexpr (BM == 256) {
        return launch_stages<BM, BN, MODE, SPLIT, ET, 3, 64, FAST>(ctx, a);
    } else if constexpr (MODE == 0) {
        // a one-K-step layer (1x1 conv, 64 input channels) needs one LDS buffer: 34 KB per block, 4 blocks per CU
        // (the single-staged split has 32-channel K-steps: never fewer than two)
        if constexpr (SPLIT != 2)
            if (a.nk == 1) return launch_stages<BM, BN, MODE, SPLIT, ET, 1, 64, FAST>(ctx, a);
        return launch_stages<BM, BN, MODE, SPLIT, ET, 2, 64, FAST>(ctx, a);
    } else {
        // one-K-step small-Cin layer (3x3 on <= 4 channels: VGG16 / M7 first conv): single LDS buffer, 64-row epilogue
        if (a.nk == 1) return launch_stages<BM, BN, MODE, SPLIT, ET, 1, 64, FAST>(ctx, a);
        return launch_stages<BM, BN, MODE, SPLIT, ET, 2, 64, FAST>(ctx, a);
    }
}

// f16 LDS-DMA layer on its FAST variant (fast = 1: epilogue, 3: epilogue + pointwise)
template <int BM, int BN, int SPLIT = 0>
int launch_fast(wsc_ctx *ctx, const ConvKArgs &a, int fast) {
    if (fast == 3) return launch_variant<BM, BN, 0, SPLIT, 1, 3>(ctx, a);
    return launch_variant<BM, BN, 0, SPLIT, 1, 1>(ctx, a);
}

template <int BN>
int launch_bn(wsc_ctx *ctx, const ConvKArgs &a, int small_cin, int split, int fmt) {
    if (split == 2) { // f16x3: generic layers single-staged, small-Cin layers in three segments on half planes
        if (small_cin == 0 && a.fast == 3) return launch_variant<128, BN, 0, 2, 1, 3>(ctx, a);
        if (small_cin == 0 && a.fast) return launch_variant<128, BN, 0, 2, 1, 1>(ctx, a);
        if (small_cin == 0) return launch_variant<128, BN, 0, 2, 1>(ctx, a);
        if (small_cin == 1) return launch_variant<128, BN, 1, 1, 1>(ctx, a);
        return launch_variant<128, BN, 2, 1, 1>(ctx, a);
    }
    if (split) {
        if (small_cin == 0) return launch_variant<128, BN, 0, true, 0>(ctx, a);
        if (small_cin == 1) return launch_variant<128, BN, 1, true, 0>(ctx, a);
        return launch_variant<128, BN, 2, true, 0>(ctx, a);
    }
    if (fmt) {
        if (small_cin == 0 && a.fast) return launch_fast<128, BN>(ctx, a, a.fast);
        if (small_cin == 1 && a.fast) return launch_variant<128, BN, 1, false, 1, 1>(ctx, a);
        if (small_cin == 2 && a.fast) return launch_variant<128, BN, 2, false, 1, 1>(ctx, a);
        if (small_cin == 0) return launch_variant<128, BN, 0, false, 1>(ctx, a);
        if (small_cin == 1) return launch_variant<128, BN, 1, false, 1>(ctx, a);
        return launch_variant<128, BN, 2, false, 1>(ctx, a);
    }
    if (small_cin == 0) return launch_variant<128, BN, 0, false, 0>(ctx, a);
    if (small_cin == 1) return launch_variant<128, BN, 1, false, 0>(ctx, a);
    return launch_variant<128, BN, 2, false, 0>(ctx, a);
}

// 256 x 128 tile (generic layers only)
int launch_big(wsc_ctx *ctx, const ConvKArgs &a, int split, int fmt) {
    if (split) return launch_variant<256, 128, 0, true, 0>(ctx, a);
    if (fmt && a.fast) return launch_fast<256, 128>(ctx, a, a.fast);
    if (fmt) return launch_variant<256, 128, 0, false, 1>(ctx, a);
    return launch_variant<256, 128, 0, false, 0>(ctx, a);
}
// 256 x 256 tile, 128 x 64 per wave (generic layers with CoutPad % 256 == 0 only)
int launch_square(wsc_ctx *ctx, const ConvKArgs &a, int split, int fmt) {
    if (split == 2) return a.fast ? launch_fast<256, 256, 2>(ctx, a, a.fast) : launch_variant<256, 256, 0, 2, 1>(ctx, a);
    if (split) return launch_variant<256, 256, 0, true, 0>(ctx, a);
    if (fmt && a.fast) return launch_fast<256, 256>(ctx, a, a.fast);
    if (fmt) return launch_variant<256, 256, 0, false, 1>(ctx, a);
    return launch_variant<256, 256, 0, false, 0>(ctx, a);
}

} // namespace

int conv_igemm_launch(wsc_ctx *ctx, const ConvLaunch &p) {
    ConvKArgs a;
    a.x = p.x; a.x_lo = p.x_lo; a.w = p.w;
    a.s1 = p.s1; a.b1 = p.b1; a.s2 = p.s2; a.b2 = p.b2;
    a.res = p.res; a.res_lo = p.res_lo;
    a.y = p.y; a.y_lo = p.y_lo; a.y_f32 = p.y_f32;
    a.H = p.H; a.W = p.W; a.Cin = p.Cin; a.Ho = p.Ho; a.Wo = p.Wo; a.Cout = p.Cout;
    a.ldy = p.ldy > 0 ? p.ldy : p.Cout;
    a.win_npix = 0;
    a.ldx = p.Cin; a.x2 = nullptr; a.lo_delta2 = 0; a.cc2 = 0; a.H2 = a.W2 = a.C2 = 0; a.stride2 = 1;
    a.kh = p.kh; a.kw = p.kw; a.stride = p.stride; a.pad = p.pad; a.relu = p.relu;
    a.M = p.N * p.Ho * p.Wo;
    a.m_base = 0;
    a.m_end = a.M;
    a.HoWo = p.Ho * p.Wo;
    auto fastdiv = [](unsigned d, unsigned &mul, unsigned &s1, unsigned &s2) {
        unsigned l = 0;
        while ((1ull << l) < d) ++l; // ceil(log2 d)
        mul = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
        s1 = l < 1 ? l : 1;
        s2 = l > 0 ? l - 1 : 0;
    };
    fastdiv((unsigned)(a.HoWo > 0 ? a.HoWo : 1), a.div_howo_mul, a.div_howo_s1, a.div_howo_s2);
    fastdiv((unsigned)(p.Wo > 0 ? p.Wo : 1), a.div_wo_mul, a.div_wo_s1, a.div_wo_s2);
    WSC_CHECK(p.split >= 0 && p.split <= 2 && !(p.split == 1 && p.fmt) && !(p.split == 2 && !p.fmt), WSC_ERR_INVALID,
              "conv: split mode %d with operand format %d (bf16x3 = split 1 on bf16 planes, f16x3 = split 2 on half planes)", p.split, p.fmt);
    // small_cin == 3: the stem of the f16x3 mode on a zero-PADDED NHWC4 input (net.hip run_backbone): the 8-pixel x 4-channel
    // window of kernel row r of an output pixel is 32 contiguous elements of either plane, 16-byte aligned (stride 2: the
    // window starts at an even pixel), so a K-step = one kernel row = 32 hi + 32 lo values -- exactly the single-staged
    // split's K-step.  The layer then IS a generic single-staged layer with kh K-steps: no bounds tests (the padding is
    // in the buffer), LDS-DMA staging, the FAST epilogue (round 3 ran it in three register-staged K segments: 342 us).
    const bool stem_rows = p.small_cin == 3;
    if (stem_rows)
        WSC_CHECK(p.split == 2 && p.Cin == 4 && p.pad == 0 && p.kw <= 7 && (p.stride & 1) == 0, WSC_ERR_INVALID,
                  "conv: the padded-stem form needs split 2, a 4-channel padded input, kw <= 7 and an even stride");
    const int small_cin_eff = stem_rows ? 0 : p.small_cin;
    a.stem_rows = stem_rows ? 1 : 0;
    a.kw_real = p.kw;
    const bool single_staged = p.split == 2 && small_cin_eff == 0;
    if (stem_rows) {
        a.kw = 1;
        a.cchunks = 1;
        a.ntaps = p.kh;
        a.ksteps_base = p.kh;
    } else if (p.small_cin == 0) {
        WSC_CHECK(p.Cin % 64 == 0, WSC_ERR_INVALID, "conv: Cin=%d not a multiple of 64", p.Cin);
        a.cchunks = p.Cin / (single_staged ? 32 : 64);
        a.ntaps = p.kh * p.kw;
        a.ksteps_base = p.kh * p.kw * a.cchunks;
    } else {
        WSC_CHECK(p.Cin == 4, WSC_ERR_INVALID, "conv: small-Cin mode needs a 4-channel activation");
        a.cchunks = 1;
        a.ntaps = p.kh * p.kw;
        // kh kernel rows, 2^small_cin slots each, 8 slots per K-step
        a.ksteps_base = ((p.kh << p.small_cin) + 7) / 8;
    }
    a.Kbase = a.ksteps_base * 64; // (single-staged split: a K-step's 64 weight elements are 32 hi + 32 lo)
    a.Kw = a.Kbase * ((p.split && !single_staged) ? 2 : 1);
    a.nk = a.ksteps_base * ((p.split && !single_staged) ? 3 : 1);
    a.lo_delta = single_staged ? (long long)(p.x_lo - p.x) : 0;
    if (p.x2 != nullptr) { // two A sources: [x (Cin - C2 channels, dense) | x2 at (ho, wo) * stride2 (C2 channels)]
        const int K1 = p.Cin - p.C2;
        WSC_CHECK(p.split != 1 && small_cin_eff == 0 && p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0 && K1 > 0 &&
                  K1 % 64 == 0 && p.C2 % 64 == 0 && p.stride2 >= 1 && (p.Ho - 1) * p.stride2 < p.H2 && (p.Wo - 1) * p.stride2 < p.W2 &&
                  (!single_staged || p.x2_lo != nullptr),
                  WSC_ERR_INVALID, "conv: a second input needs a 1x1 / stride 1 layer, channel counts in multiples of 64 and no bf16x3");
        a.x2 = p.x2;
        a.lo_delta2 = single_staged ? (long long)(p.x2_lo - p.x2) : 0;
        a.cc2 = K1 / (single_staged ? 32 : 64);
        a.H2 = p.H2; a.W2 = p.W2; a.C2 = p.C2; a.stride2 = p.stride2;
        a.ldx = K1;
    }
    const int BN = p.CoutPad % 128 == 0 ? 128 : 64;
    WSC_CHECK(p.CoutPad % 64 == 0, WSC_ERR_INVALID, "conv: CoutPad=%d not a multiple of 64", p.CoutPad);
    a.ntiles_n = p.CoutPad / BN;
    a.zero = (const bf16_t *)ctx->zero_page;
    a.range = ctx->range_dev;
#ifdef WSC_AB_KNOBS
    // timing-only ablations (wrong results): 1 = no DMA after the prologue, 2 = no fragment reads / MFMAs, 4 = no MFMAs
    static const int debug = [] { const char *e = getenv("WSC_CONV_DEBUG"); return e ? atoi(e) : 0; }();
    a.debug = debug;
#else
    a.debug = 0;
#endif
    // FAST variants (see the kernel): f16, one precision plane (or the single-staged split), fp16 output only, full column
    // tiles, no post-ReLU affine.  p.generic (wsc_conv2d_nchw's WSC_CONV_GENERIC flag) keeps the generic variants: a test
    // holds the two to the same bits.
    const int nofast = p.generic;
    a.fast = 0;
    if (!nofast && p.fmt && (p.split == 0 || single_staged) && p.y != nullptr && p.y_f32 == nullptr &&
        p.Cout == p.CoutPad && (long long)a.M * a.ldy < (1ll << 31)) {
        a.fast = 1;
        if (small_cin_eff == 0 && p.kh == 1 && p.kw == 1 && p.pad == 0 && p.stride == 1) a.fast = 3;
    }
    if (a.M == 0) return WSC_OK;
    // tile choice.  Measured on the ResNet50-CAM stack (64 samples @321^2, f16): 128-row tiles 4.31 ms,
    // 256-row 3-stage tiles on the K >= 512 layers 4.37 ms, everywhere 4.45 ms -- the stack is bound by
    // per-block memory latency with 1-2 blocks per CU (an ablation without DMA and without MFMAs still
    // takes 2.2 ms), not by L2->LDS bytes per FLOP, so the 128-row tile stays the default.  128x64 tiles
    // (3 blocks per CU) on the K <= 128 / 256 / 512 layers were also measured: no layer gained, layer1.conv3
    // lost 10 %.
    // WSC_CONV_TILE (A/B runs): 256 selects the 256x128 tile wherever it applies, -1 where K >= 512, 512 the
    // 256x256 tile wherever CoutPad % 256 == 0, 1 (any other value) the 128-row tiles everywhere.
#ifdef WSC_AB_KNOBS
    static const int force = [] { const char *e = getenv("WSC_CONV_TILE"); return e ? atoi(e) : 0; }();
#else
    constexpr int force = 0;
#endif
    const long long blocks256 = ((a.M + 255) / 256) * (long long)a.ntiles_n;
    bool big = false;
    if (force == -1) big = small_cin_eff == 0 && BN == 128 && a.Kbase >= 512 && blocks256 >= 200;
    if (force == 256) big = small_cin_eff == 0 && BN == 128;
    // 256 x 256 tile: half the L2->LDS bytes per FLOP of the 128 x 128 tile; needs enough K-steps to amortise
    // its 130 KB prologue/epilogue and enough tiles to fill 256 CUs at one block per CU.  VGG16 @321, 64
    // samples: 806 -> 920 TFLOP/s on the 11 layers it takes (stack 11.5 -> 10.8 ms).  Thresholds (>= 4 K-steps, >= one
    // round of tiles, later 3/4 of a round) from a sweep on ResNet50: (8, 768) 3.96 ms, (8, 256) 3.96, (4, 768) 3.94, (4, 256) 3.92; VGG16 +-0.
    // A grid of 192+ tiles (3/4 of a round: layer4's 3x3 and 1x1 -> 512 convs, 222 tiles) also wins: 4.11 -> 3.97 ms.
    const long long blocks_sq = ((a.M + 255) / 256) * (long long)(p.CoutPad / 256);
    bool square = small_cin_eff == 0 && p.CoutPad % 256 == 0 && a.nk >= 4 && blocks_sq >= 192;
    if (force == 512) square = small_cin_eff == 0 && p.CoutPad % 256 == 0;
    if (force != 0 && force != 512) square = false;
    if (p.split == 2) big = false; // (the three-buffer 256 x 128 tile has no single-staged variant)
    if (p.x2 != nullptr) big = false; // (nor a second A source)
    if (p.split == 2 && !a.fast) square = false; // (its generic epilogue next to 128 accumulators + both planes' fragments spills)
    // f16x3 K-steps are 32 channels: with fewer than 16 of them (K < 512: layer2's 128 -> 512 and 256 -> 512 convs) the 256 x 256
    // block's prologue + two-group epilogue outweigh its smaller staging traffic (sweep: 109 vs 127 us, 149 vs 162 us)
    if (p.split == 2 && force == 0 && a.nk < 16) square = false;
    if (square) {
        const int ntn = p.CoutPad / 256;
        // One block per CU: a grid of r * 256 + rem tiles takes r + 1 rounds.  When the last round would be less
        // than half full, the square tiles take whole rounds only and the remaining rows go to the 128 x 128 kernel
        // (<= 512 tiles = one round at 2 blocks per CU, ~0.56 of a square round): VGG16 conv4 (840 tiles) 4 -> 3.6
        // rounds.  The two launches write disjoint output rows.
        const long long rounds = blocks_sq / ctx->num_cus, rem = blocks_sq - rounds * ctx->num_cus;
#ifdef WSC_AB_KNOBS
        static const int nosplit = [] { const char *e = getenv("WSC_CONV_NOSPLIT"); return e ? atoi(e) : 0; }();
#else
        constexpr int nosplit = 0;
#endif
        if (!nosplit && rounds >= 1 && rem > 0 && rem * 2 <= ctx->num_cus && ctx->num_cus > 0) {
            const int big_rows = (int)((rounds * ctx->num_cus) / ntn); // 256-row tile rows given to the square kernel
            ConvKArgs b = a;
            b.ntiles_n = ntn;
            b.m_end = big_rows * 256;
            b.nblocks = big_rows * ntn;
            WSC_TRY(launch_square(ctx, b, p.split, p.fmt));
            a.m_base = big_rows * 256;
            a.nblocks = ((a.M - a.m_base + 127) / 128) * a.ntiles_n; // BN = 128 here (CoutPad % 256 == 0)
            return launch_bn<128>(ctx, a, small_cin_eff, p.split, p.fmt);
        }
        a.ntiles_n = ntn;
        a.nblocks = (int)blocks_sq;
        return launch_square(ctx, a, p.split, p.fmt);
    }
    const int BMsel = big ? 256 : 128;
    a.nblocks = ((a.M + BMsel - 1) / BMsel) * a.ntiles_n;
    // LDS input window (north star: "3x3 convolutions as MFMA-tiled direct convs with LDS-staged input windows"): the f16x3
    // 3 x 3 / stride 1 / pad 1 layers on 128-row tiles whose window -- the raster positions from (first output pixel - W - 1) to
    // (last output pixel + W + 1), plus the zero row -- fits 256 or 320 positions (32 / 40 KB next to the weight tiles' 32 / 16 KB:
    // two blocks per CU): ResNet50 @321 layer2 / layer3 conv2 (41 x 41: 213, 21 x 21: 173 positions) and, since round 6's
    // unpadded raster, layer1 conv2 (81 x 81: 293).
    if (!big && single_staged && a.fast == 1 && p.kh == 3 && p.kw == 3 && p.stride == 1 && p.pad == 1 && p.x2 == nullptr &&
        p.Ho == p.H && p.Wo == p.W && ctx->opt[WSC_OPT_CONV_WINDOW] != 0) {
        // positions of a block's window: its 128 output pixels' raster span, one row + one pixel before and after, and the
        // zero row at the window's last position
        const long long need = 127 + 2ll * p.W + 3 + 1;
        const long long npix = (long long)p.N * p.H * p.W;
        // (64-column tiles -- Cout = 64: ResNet50 layer1 conv2 at 81 x 81 -- take the 320-position window: 40 KB + 16 KB of weight
        // tiles, two blocks per CU instead of three -- measured below; round 5's padded raster needed 480 positions there: rejected)
        // (window source offsets are non-negative 32-bit counts of 16-byte units measured from x, the lo plane's distance included:
        // a lo plane BELOW x keeps the per-tap tiles -- ADVICE r5)
        if (need <= 320 && npix < (1ll << 30) && a.lo_delta >= 0 && npix * p.Cin + a.lo_delta < (1ll << 33) && (a.lo_delta & 7) == 0) {
            a.win_npix = (int)npix;
            if (BN == 128) return need <= 256 ? launch_window<8>(ctx, a) : launch_window<10>(ctx, a);
            return need <= 256 ? launch_window<8, 64>(ctx, a) : launch_window<10, 64>(ctx, a);
        }
    }
    if (big) return launch_big(ctx, a, p.split, p.fmt);
    if (BN == 128) return launch_bn<128>(ctx, a, small_cin_eff, p.split, p.fmt);
    return launch_bn<64>(ctx, a, small_cin_eff, p.split, p.fmt);
}
