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

namespace {

constexpr int BM = 128;
constexpr int BK = 64;

struct ConvKArgs {
    const bf16_t *x, *x_lo, *w;
    const float *s1, *b1, *s2, *b2;
    const bf16_t *res, *res_lo;
    bf16_t *y, *y_lo;
    float *y_f32;
    int H, W, Cin, Ho, Wo, Cout;
    int kh, kw, stride, pad, relu;
    int M, HoWo;
    int cchunks;     // Cin / 64 (generic mode)
    int ksteps_base; // K-steps of one precision segment
    int nk;          // total K-steps (x3 in split mode)
    int Kw;          // packed weight row length in elements
    int Kbase;       // elements of one weight plane per row
    int ntiles_n, nblocks;
    const bf16_t *zero; // >= 16 bytes of zeros in HBM: source of padded taps for the LDS-DMA path
};

__device__ __forceinline__ int lds_off(int row, int slot) {
    return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
}

template <int BN, int MODE, bool SPLIT, int ET, bool GLDS>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvKArgs p) {
    constexpr int WN = BN / 2;
    constexpr int NI = WN / 32;
    constexpr int NB = BN / 32; // B rows per thread
    constexpr int A_BYTES = BM * BK * 2;
    constexpr int B_BYTES = BN * BK * 2;
    constexpr int CT_STRIDE = BN + 4;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wv = t >> 6;
    const int wm = wv >> 1, wn = wv & 1;

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
    const int m0 = mt * BM;
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
        if (m < p.M) {
            const int n = m / p.HoWo;
            const int rem = m - n * p.HoWo;
            const int ho = rem / p.Wo;
            const int wo = rem - ho * p.Wo;
            hb[i] = ho * p.stride - p.pad;
            wb[i] = wo * p.stride - p.pad;
            base[i] = (((long long)n * p.H + hb[i]) * p.W + wb[i]) * (long long)p.Cin;
        } else {
            hb[i] = -(1 << 28);
            wb[i] = 0;
            base[i] = 0;
        }
    }
    // B rows: register path (t>>3) + 32 i; DMA path wave w owns rows w*(BN/4) + i*8 + (lane>>3)
    const int brow0 = GLDS ? (wv * (BN / 4) + (lane >> 3)) : (t >> 3);
    const bf16_t *wrow[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int r = brow0 + RSTEP * i;
        const int ks = GLDS ? (slot ^ ((r >> 1) & 7)) : slot;
        wrow[i] = p.w + (long long)(n0 + r) * p.Kw + ks * 8;
    }

    uint4 ra[4], rb[NB];

    // LDS-DMA issue of one K-step's A and B tiles into buffer `buf` (MODE 0 only)
    auto issue_dma = [&](int kt, int buf) {
        int seg = 0, ktl = kt;
        if (SPLIT) {
            seg = kt / p.ksteps_base;
            ktl = kt - seg * p.ksteps_base;
        }
        const bf16_t *src = (SPLIT && seg == 1) ? p.x_lo : p.x;
        const int tap = ktl / p.cchunks;
        const int cc = ktl - tap * p.cchunks;
        const int khi = tap / p.kw;
        const int kwi = tap - khi * p.kw;
        const long long tap_off = ((long long)khi * p.W + kwi) * p.Cin + cc * 64;
        char *sa = smem + buf * A_BYTES + wv * 4096;
        char *sb = smem + 2 * A_BYTES + buf * B_BYTES + wv * (NB * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lrow + RSTEP * i;
            const int ks = slot ^ ((r >> 1) & 7);
            const int hi = hb[i] + khi, wi = wb[i] + kwi;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
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

    auto load_tile = [&](int kt) {
        int seg = 0, ktl = kt;
        if (SPLIT) {
            seg = kt / p.ksteps_base;
            ktl = kt - seg * p.ksteps_base;
        }
        const bf16_t *src = (SPLIT && seg == 1) ? p.x_lo : p.x;
        if (MODE == 0) {
            const int tap = ktl / p.cchunks;
            const int cc = ktl - tap * p.cchunks;
            const int khi = tap / p.kw;
            const int kwi = tap - khi * p.kw;
            const long long tap_off = ((long long)khi * p.W + kwi) * p.Cin + cc * 64 + slot * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hi = hb[i] + khi, wi = wb[i] + kwi;
                const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                ra[i] = ok ? *reinterpret_cast<const uint4 *>(src + base[i] + tap_off) : make_uint4(0, 0, 0, 0);
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
                ra[i] = make_uint4(v0.x, v0.y, v1.x, v1.y);
            }
        }
        const int wk = ((SPLIT && seg == 2) ? p.Kbase : 0) + ktl * 64;
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const uint4 *>(wrow[i] + wk);
    };

    auto store_lds = [&](int buf) {
        char *sa = smem + buf * A_BYTES;
        char *sb = smem + 2 * A_BYTES + buf * B_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4 *>(sa + lds_off(lrow + 32 * i, slot)) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<uint4 *>(sb + lds_off(lrow + 32 * i, slot)) = rb[i];
    };

    f32x16_t acc[2][NI];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int l31 = lane & 31;
    const int kgrp = lane >> 5;

    auto compute = [&](int buf) {
        const char *sa = smem + buf * A_BYTES;
        const char *sb = smem + 2 * A_BYTES + buf * B_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int sl = ks * 2 + kgrp;
            uint4 af[2], bfr[NI];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                af[mi] = *reinterpret_cast<const uint4 *>(sa + lds_off(wm * 64 + mi * 32 + l31, sl));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                bfr[ni] = *reinterpret_cast<const uint4 *>(sb + lds_off(wn * WN + ni * 32 + l31, sl));
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if (ET == 0)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8_t, af[mi]), __builtin_bit_cast(bf16x8_t, bfr[ni]), acc[mi][ni], 0,
                            0, 0);
                    else
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(f16x8_t, af[mi]), __builtin_bit_cast(f16x8_t, bfr[ni]), acc[mi][ni], 0, 0,
                            0);
                }
        }
    };

    // ---- main loop ------------------------------------------------------------------
    const int nk = p.nk;
    if (GLDS && MODE == 0) {
        // the DMA for K-step kt+1 is issued before the MFMAs of K-step kt and has landed when the
        // barrier (which carries the vmcnt(0) for the in-flight DMA) releases
        issue_dma(0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) issue_dma(kt + 1, cur ^ 1);
            compute(cur);
            __syncthreads();
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
    float *ct = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kgrp;
                const int col = wn * WN + ni * 32 + l31;
                ct[row * CT_STRIDE + col] = acc[mi][ni][r];
            }
    __syncthreads();

    constexpr int TPR = BN / 8;    // threads per row
    constexpr int RPP = 256 / TPR; // rows per pass
    const int c8 = t % TPR;
    const int r0 = t / TPR;
    const int c = n0 + c8 * 8;
    if (c < p.Cout) {
        float s1[8], b1[8], s2[8], b2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s1[j] = p.s1[c + j];
            b1[j] = p.b1[c + j];
        }
        const bool post = p.s2 != nullptr;
        if (post) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s2[j] = p.s2[c + j];
                b2[j] = p.b2[c + j];
            }
        }
        const bool full = c + 8 <= p.Cout;
        for (int pass = 0; pass < BM / RPP; ++pass) {
            const int row = pass * RPP + r0;
            const int m = m0 + row;
            if (m >= p.M) break;
            float v[8];
            const f32x4_t q0 = *reinterpret_cast<const f32x4_t *>(ct + row * CT_STRIDE + c8 * 8);
            const f32x4_t q1 = *reinterpret_cast<const f32x4_t *>(ct + row * CT_STRIDE + c8 * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = q0[j];
                v[4 + j] = q1[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] * s1[j] + b1[j];
            const long long o = (long long)m * p.Cout + c;
            if (p.res != nullptr && full) {
                const uint4 rv = *reinterpret_cast<const uint4 *>(p.res + o);
                const uint32_t rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[2 * j] += h16_to_f32((bf16_t)(rw[j] & 0xffffu), ET);
                    v[2 * j + 1] += h16_to_f32((bf16_t)(rw[j] >> 16), ET);
                }
                if (SPLIT) {
                    const uint4 lv = *reinterpret_cast<const uint4 *>(p.res_lo + o);
                    const uint32_t lw[4] = {lv.x, lv.y, lv.z, lv.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[2 * j] += bf16_to_f32((bf16_t)(lw[j] & 0xffffu));
                        v[2 * j + 1] += bf16_to_f32((bf16_t)(lw[j] >> 16));
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
                        const bf16_t l0 = f32_to_bf16(v[2 * j] - bf16_to_f32(h0));
                        const bf16_t l1 = f32_to_bf16(v[2 * j + 1] - bf16_to_f32(h1));
                        lw[j] = (uint32_t)l0 | ((uint32_t)l1 << 16);
                    }
                }
                *reinterpret_cast<uint4 *>(p.y + o) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
                if (SPLIT) *reinterpret_cast<uint4 *>(p.y_lo + o) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
            }
        }
    }
}

template <int BN, int MODE, bool SPLIT, int ET>
int launch_variant(wsc_ctx *ctx, const ConvKArgs &a) {
    constexpr bool GLDS = MODE == 0; // LDS-DMA staging for every generic layer; small-Cin layers stage via registers
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int PIPE = 2 * (A_BYTES + B_BYTES);
    constexpr int EPI = BM * (BN + 4) * 4;
    constexpr int LDS = PIPE > EPI ? PIPE : EPI;
    static bool attr_set = false;
    auto kern = conv_igemm_kernel<BN, MODE, SPLIT, ET, GLDS>;
    if (!attr_set) {
        WSC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    // algorithmic FLOPs: 2 * M * Cout * (kh*kw*Cin_real), x1 regardless of the precision mode
    const double flops = 2.0 * a.M * a.Cout * (MODE == 0 ? (double)a.kh * a.kw * a.Cin : (double)a.kh * a.kw * 3);
    WscKernelTimer timer(ctx, MODE != 0 ? WSC_K_CONV_SMALLCIN : (BN == 128 ? WSC_K_CONV128 : WSC_K_CONV64), flops);
    hipLaunchKernelGGL(kern, dim3(a.nblocks), dim3(256), LDS, ctx->stream, a);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

template <int BN>
int launch_bn(wsc_ctx *ctx, const ConvKArgs &a, int small_cin, int split, int fmt) {
    if (split) {
        if (small_cin == 0) return launch_variant<BN, 0, true, 0>(ctx, a);
        if (small_cin == 1) return launch_variant<BN, 1, true, 0>(ctx, a);
        return launch_variant<BN, 2, true, 0>(ctx, a);
    }
    if (fmt) {
        if (small_cin == 0) return launch_variant<BN, 0, false, 1>(ctx, a);
        if (small_cin == 1) return launch_variant<BN, 1, false, 1>(ctx, a);
        return launch_variant<BN, 2, false, 1>(ctx, a);
    }
    if (small_cin == 0) return launch_variant<BN, 0, false, 0>(ctx, a);
    if (small_cin == 1) return launch_variant<BN, 1, false, 0>(ctx, a);
    return launch_variant<BN, 2, false, 0>(ctx, a);
}

} // namespace

int conv_igemm_launch(wsc_ctx *ctx, const ConvLaunch &p) {
    ConvKArgs a;
    a.x = p.x; a.x_lo = p.x_lo; a.w = p.w;
    a.s1 = p.s1; a.b1 = p.b1; a.s2 = p.s2; a.b2 = p.b2;
    a.res = p.res; a.res_lo = p.res_lo;
    a.y = p.y; a.y_lo = p.y_lo; a.y_f32 = p.y_f32;
    a.H = p.H; a.W = p.W; a.Cin = p.Cin; a.Ho = p.Ho; a.Wo = p.Wo; a.Cout = p.Cout;
    a.kh = p.kh; a.kw = p.kw; a.stride = p.stride; a.pad = p.pad; a.relu = p.relu;
    a.M = p.N * p.Ho * p.Wo;
    a.HoWo = p.Ho * p.Wo;
    if (p.small_cin == 0) {
        WSC_CHECK(p.Cin % 64 == 0, WSC_ERR_INVALID, "conv: Cin=%d not a multiple of 64", p.Cin);
        a.cchunks = p.Cin / 64;
        a.ksteps_base = p.kh * p.kw * a.cchunks;
    } else {
        WSC_CHECK(p.Cin == 4, WSC_ERR_INVALID, "conv: small-Cin mode needs a 4-channel activation");
        a.cchunks = 1;
        // kh kernel rows, 2^small_cin slots each, 8 slots per K-step
        a.ksteps_base = ((p.kh << p.small_cin) + 7) / 8;
    }
    a.Kbase = a.ksteps_base * 64;
    a.Kw = a.Kbase * (p.split ? 2 : 1);
    a.nk = a.ksteps_base * (p.split ? 3 : 1);
    const int BN = p.CoutPad % 128 == 0 ? 128 : 64;
    WSC_CHECK(p.CoutPad % 64 == 0, WSC_ERR_INVALID, "conv: CoutPad=%d not a multiple of 64", p.CoutPad);
    a.ntiles_n = p.CoutPad / BN;
    a.zero = (const bf16_t *)ctx->zero_page;
    a.nblocks = ((a.M + BM - 1) / BM) * a.ntiles_n;
    if (a.M == 0) return WSC_OK;
    WSC_CHECK(!(p.split && p.fmt), WSC_ERR_INVALID, "conv: split precision requires bf16 planes");
    if (BN == 128) return launch_bn<128>(ctx, a, p.small_cin, p.split, p.fmt);
    return launch_bn<64>(ctx, a, p.small_cin, p.split, p.fmt);
}
