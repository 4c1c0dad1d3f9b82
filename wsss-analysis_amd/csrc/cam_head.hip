// cam_head.hip -- the 1x1 CAM / Grad-CAM head as a streaming GEMM: out[m][c] = act(sum_k x[m][k] w[c][k] * s1[c] + b1[c]).
//
// Reference: F.conv2d(x, classifier.weight) of 03b_irn/net/resnet50_cam.py:65 / vgg16_cam.py:48, and the Grad-CAM contraction
// np.einsum('ijkl,lm->ijkm', conv_val, weights) of 02_cues/utilities.py:133 / 03c_hsn/utilities.py:258 (m7_cam.py:45-46).
//
// The head has C = 20 ... 31 output channels over K = 256 ... 2048 inputs: through the tiled implicit-GEMM kernel it is a
// 128 x 64 tile whose columns are two thirds padding, 221 blocks for 256 CUs, each walking 64 K-steps behind one barrier per
// step -- 82 us for a 231 MB read (2.85 TB/s, r04_pmc_conv.txt: matrix pipe 10.8 %).  It is an HBM stream, so it runs as one:
//   * a block owns 64 rows (441 blocks at 64 samples x 21 x 21), its four waves each a QUARTER of K for all 64 rows;
//   * the MFMA operands never touch LDS: lane (row l & 31, k-group l >> 5) of v_mfma_f32_32x32x16_f16 holds 8 consecutive k of
//     one row = 16 contiguous bytes of the activation row (A) or of the packed weight row (B), loaded straight from memory, four
//     k-slices (one 128-byte line per row and plane) per lane in flight ahead of the MFMAs;
//   * f16x3: the same three products per k-slice as conv_igemm (lo*hi, hi*lo, hi*hi) into fp32 accumulators;
//   * the four K-quarters meet in LDS (32 KB of fp32) and are added in wave order 0, 1, 2, 3 -- a fixed order: the result does
//     not depend on scheduling or on what else is in the batch -- then scale / shift / ReLU and one fp32 store per value.
// Weights: the conv_igemm packing of net.hip::make_conv (f16x3: per 32-channel chunk [32 hi | 32 lo]; f16: plain K order),
// 164 KB for C = 20, K = 2048: L2-resident, read through L1 by every block.
#include "common.h"

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

struct HeadArgs {
    const bf16_t *x, *x_lo; // [M][K] IEEE half planes
    const bf16_t *w;        // packed [CoutPad][Kw]
    const float *s1, *b1;   // [CoutPad]
    float *y;               // [M][C] fp32
    int M, K, C, Kw, relu;
};

template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void cam_head_kernel(HeadArgs p) {
    __shared__ float red[4][64][33];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int l31 = lane & 31, kgrp = lane >> 5;
    const int m0 = blockIdx.x * 64;
    const int kq = p.K >> 2;           // the wave's quarter of K (a multiple of 64: K % 256 == 0 -- whole trips of U slices)
    const int k_begin = wv * kq;
    const int nsl = kq >> 4;           // k-slices of 16
    // A rows of the lane: m0 + l31 and m0 + 32 + l31 (clamped: rows past the end are computed and never stored)
    const int r0 = min(m0 + l31, p.M - 1), r1 = min(m0 + 32 + l31, p.M - 1);
    const bf16_t *a0 = p.x + (long long)r0 * p.K + k_begin + kgrp * 8;
    const bf16_t *a1 = p.x + (long long)r1 * p.K + k_begin + kgrp * 8;
    const long long lo_d = SPLIT ? (long long)(p.x_lo - p.x) : 0ll;
    // B row of the lane: output channel l31 (rows >= C of the packed matrix are zero padding up to CoutPad >= 32)
    const bf16_t *bw = p.w + (long long)l31 * p.Kw;
    auto b_at = [&](int k) -> const bf16_t * {  // k: absolute input channel of the slice's first element for this lane
        if (SPLIT) return bw + (k >> 5) * 64 + (k & 31);
        return bw + k;
    };
    f32x16_t acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    auto mfma = [](const u32x4_t &a, const u32x4_t &b, f32x16_t &c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    };
    constexpr int U = 4; // k-slices per trip: 4 x 16 channels = one 128-byte line of each row and plane
    for (int s = 0; s < nsl; s += U) {
        u32x4_t ah0[U], ah1[U], al0[U], al1[U], bh[U], bl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ko = (s + u) * 16;
            ah0[u] = *reinterpret_cast<const u32x4_t *>(a0 + ko);
            ah1[u] = *reinterpret_cast<const u32x4_t *>(a1 + ko);
            if (SPLIT) {
                al0[u] = *reinterpret_cast<const u32x4_t *>(a0 + lo_d + ko);
                al1[u] = *reinterpret_cast<const u32x4_t *>(a1 + lo_d + ko);
            }
            const bf16_t *bp = b_at(k_begin + ko + kgrp * 8);
            bh[u] = *reinterpret_cast<const u32x4_t *>(bp);
            if (SPLIT) bl[u] = *reinterpret_cast<const u32x4_t *>(bp + 32);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (SPLIT) {
                mfma(al0[u], bh[u], acc0);
                mfma(al1[u], bh[u], acc1);
                mfma(ah0[u], bl[u], acc0);
                mfma(ah1[u], bl[u], acc1);
            }
            mfma(ah0[u], bh[u], acc0);
            mfma(ah1[u], bh[u], acc1);
        }
    }
    // accumulator (32 x 32 tile): lane holds column l31, rows (r & 3) + 8 (r >> 2) + 4 kgrp
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kgrp;
        red[wv][row][l31] = acc0[r];
        red[wv][32 + row][l31] = acc1[r];
    }
    __syncthreads();
    for (int i = t; i < 64 * 32; i += 256) {
        const int row = i >> 5, c = i & 31;
        const int m = m0 + row;
        if (c < p.C && m < p.M) {
            float v = ((red[0][row][c] + red[1][row][c]) + red[2][row][c]) + red[3][row][c];
            v = v * p.s1[c] + p.b1[c];
            if (p.relu) v = fmaxf(v, 0.f);
            p.y[(long long)m * p.C + c] = v;
        }
    }
}

} // namespace

// x / x_lo: IEEE-half planes [M][K] (x_lo null: one plane, f16 mode); w: conv_igemm packing with CoutPad >= 32 rows of Kw elements
// (x_lo given: the f16x3 interleaving, per 32-channel chunk [32 hi | 32 lo], Kw = 2 K; else Kw = K).
// Returns WSC_ERR_INVALID for anything the streaming form does not take (the caller keeps the tiled kernel for those): the kernel
// reads weight rows 0 .. 31 unconditionally, indexes the packed rows with (k >> 5) * 64, and moves 16-byte pieces of both planes
// through one per-lane offset -- all of that is checked HERE, not left to the caller.
int launch_cam_head(wsc_ctx *ctx, const bf16_t *x, const bf16_t *x_lo, int M, int K, const bf16_t *w, int Kw, int CoutPad,
                    const float *s1, const float *b1, int C, int relu, float *y) {
    WSC_CHECK(M > 0 && K > 0 && K % 256 == 0 && C > 0 && C <= 32, WSC_ERR_INVALID, "cam head: M=%d K=%d C=%d (K in multiples of 256)", M, K, C);
    WSC_CHECK(CoutPad >= 32 && Kw == K * (x_lo ? 2 : 1), WSC_ERR_INVALID, "cam head: CoutPad=%d Kw=%d for K=%d (%s packing)", CoutPad, Kw, K,
              x_lo ? "f16x3" : "f16");
    const auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    WSC_CHECK(x && w && y && s1 && b1 && al16(x) && al16(w) && (x_lo == nullptr || (al16(x_lo) && ((x_lo - x) & 7) == 0)), WSC_ERR_INVALID,
              "cam head: operands must be 16-byte aligned (x %p, x_lo %p, w %p)", (const void *)x, (const void *)x_lo, (const void *)w);
    HeadArgs a;
    a.x = x; a.x_lo = x_lo; a.w = w; a.s1 = s1; a.b1 = b1; a.y = y;
    a.M = M; a.K = K; a.C = C; a.Kw = Kw; a.relu = relu;
    const int blocks = (M + 63) / 64;
    // class: with the 128 x 64 tiles it replaces; algorithmic FLOPs as for every conv class
    WscKernelTimer timer(ctx, WSC_K_CONV64, 2.0 * M * (double)C * K);
    if (x_lo) hipLaunchKernelGGL(cam_head_kernel<true>, dim3(blocks), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(cam_head_kernel<false>, dim3(blocks), dim3(256), 0, ctx->stream, a);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}
