// crf.hip -- dense-CRF mean-field inference on permutohedral lattices (pydensecrf replacement).
//
// Reference call sites: 03c_hsn/utilities.py:427-444 (dcrf_process), the
// misc.imutils.crf_inference_label calls of 03b_irn/step/cam_to_ir_label.py:35-67 and
// lib.crf.crf_inference of 03a_sec-dsrg (SEC.py:275, DSRG.py:328, model.py:689-693).
// Algorithm: Kraehenbuehl & Koltun's DenseCRF with Adams/Baek/Davis permutohedral
// filtering, NORMALIZE_SYMMETRIC kernels, Potts compatibility (see oracle/densecrf_ref.c
// for the CPU restatement this file is checked against).
//
// Data layout in HBM (whole batch of B images of H x W, N = H*W pixels each):
//   lattice vertex rows of ALL images share one index space: row 0 is a permanent zero row
//   (a missing blur neighbour points at it), image 0's vertices follow in first-touch
//   raster order (the CPU reference's insertion order), then image 1's, ...
//   offset[e], bary[e]   e = (b*N + n)*(d+1) + r : row id / barycentric weight of the r-th
//                        enclosing simplex vertex of pixel n
//   csr_start[row], csr_pix[], csr_w[] : per row, the pixels splatting into it (gather form
//                        of the splat: no float atomics in the iteration loop)
//   nbr[j][row] = (n1, n2) : blur neighbours along axis j
//   norm[b*N+n]          : 1/sqrt(Lattice(1)+1e-20)
//   val[row][M]          : lattice values, class-minor (M contiguous floats per row)
//   Q[pixel][M], U[pixel][M] : pixel-major, class-minor
// Every iteration kernel is a flat, fully coalesced pass over these arrays; all are
// HBM-bound (SURVEY.md section 8d gives the algorithmic byte count).
//
// This file is compiled with -ffp-contract=off: the simplex search compares rounded
// float expressions and must not be re-associated into FMAs, or pixels near a cell
// boundary land in a different (equally valid) simplex than the CPU reference picks.
#include "common.h"

#include <atomic>
#include <cmath>
#include <cstring>

namespace {

constexpr unsigned long long EMPTY_KEY = 0xFFFFFFFFFFFFFFFFull;
typedef float f32x2_t __attribute__((ext_vector_type(2)));

struct LatticeDev {
    int d = 0;
    // rep > 1: the index arrays below describe ONE image and are shared by `rep` images whose value
    // rows / pixels are laid out back to back (replica k uses rows [k*rows, (k+1)*rows) and pixels
    // [k*n_pix, (k+1)*n_pix)).  The Gaussian lattice depends on (H, W, sxy) only, so the whole batch
    // shares one copy; the bilateral lattice is per image content (rep = 1, arrays cover the batch).
    int rep = 1;
    int rows = 0;               // rows of one replica incl. its zero row (V + 1)
    int32_t *offset = nullptr;  // [B*N*(d+1)]
    float *bary = nullptr;      // [B*N*(d+1)]
    float *norm = nullptr;      // [B*N]
    int2 *nbr = nullptr;          // [(d+1)][rows]
    // Splat tables, pixel-tile major (see "pixel tiles" below).  The splat of a tile's pixels is a gather per
    // SLOT: a slot is a run of <= SLOT_ENT entries (pixel of the tile, weight) that go to one lattice row, in a
    // fixed (sorted) order, so plain fp32 sums are reproducible; every slot produces one partial row, and a row's
    // value is the (order-independent, fixed-point) sum of its slots' partials.
    int32_t *tslot_start = nullptr;  // [n_tiles + 1] first slot of each tile
    int2 *slot_desc = nullptr;       // [n_slots] {first entry (relative to the tile) | entry count << 16, index of its partial row}
    // entries, tile-major, grouped by row: weight w * norm[pixel] and the pixel's index inside its tile (TILE_PIX <= 256: one
    // byte) as two arrays -- 5 bytes per entry instead of a uint2's 8 (the update re-reads them every iteration)
    float *tent_w = nullptr;         // [B*N*(d+1)]
    uint8_t *tent_p = nullptr;       // [B*N*(d+1)]
    // partial rows are laid out ROW-major: the slots of row r write partial rows [row_slot_start[r], row_slot_start[r+1])
    int32_t *row_slot_start = nullptr; // [rows + 1]
    int n_tiles = 0, n_slots = 0;
    bool sorted_dest = false; // partial rows of a row in deterministic order: plain fp32 combine (else fixed point)
    long long n_pix = 0; // pixels of one replica (B*N when rep == 1)
    int M_cur = 0;       // class count of the inference in flight (algorithmic byte accounting)
    float alpha = 0.f;
    // Gaussian lattice only (d = 2): tiles of the dense (i, j) index space for the fused three-pass blur
    int32_t *tile_rows = nullptr; // [n_tiles][GBI * GBJ] row id of every point of the tile's halo box (0 = absent)
    int2 *tile_pstart = nullptr;  // [n_tiles][GBI * GBJ] {first partial row, count} of that row
    int32_t *tile_list = nullptr; // [n_tiles_occ] tiles with at least one interior vertex
    int n_tiles_occ = 0;
    // Gaussian lattice only: per PIXEL tile, the closed vertex set the update kernel blurs on chip (gauss_fuse_tables):
    // the tile's own vertices first, then the neighbours the three blur passes reach (axis 2, then 1, then 0)
    int4 *gt_cnt = nullptr;      // [tpi] sizes of the nested sets: T, T + nbr_2, T + nbr_2 + nbr_1, all (passes 2 / 1 / 0 / load)
    int2 *gt_rows = nullptr;     // [tpi][gt_stride] {first partial row, partial rows} of local vertex v (padding: {0, 0})
    uint4 *gt_nbr = nullptr;     // [tpi][gt_stride] local ids of the blur neighbours: x / y / z = axis 0 / 1 / 2 as n1 | n2 << 16
    uint4 *gt_pix = nullptr;     // [N] per pixel {local ids of its 3 vertices (10 bits each), bary[r] * norm (3 floats)}
    int gt_stride = 0;           // max vertices of a tile's set + 1; local id gt_stride - 1 is every set's zero row
    std::vector<int32_t> v_per_image;
    // per-image lattices (rep == 1): rows [img_row[b], img_row[b + 1]) are image b's (row 0, the zero row, is nobody's)
    int32_t *img_row = nullptr; // [B + 1]
    int max_img_rows = 0;
    uint32_t *part_row = nullptr; // [n_slots] per partial row: its lattice row | min(partial rows of that row, 255) << 24
    // blur_lds_kernel: every image takes the kernel variant its vertex count admits (BL_VAR); the workgroup table lists
    // {image, variant, class group} for the class count of the call
    bool bl_ok = false;           // every image fits a variant
    int2 *bl_blk = nullptr;                      // device [bl_cap] workgroup table {image | variant << 24, class group} (-1: idle),
    int bl_cap = 0;                              // written per class count by blur_lds()
    mutable std::vector<int2> bl_blk_host;       // (kept alive for the asynchronous upload)
    mutable int bl_M = -1, bl_nblk = 0;          // class count the table was built for, its length
    mutable size_t bl_lds = 0;
};

// ---- pixel tiles ------------------------------------------------------------------------------------
// The iteration kernels walk the image in 2-D pixel tiles of at most TILE_H x TILE_W pixels (balanced split: an
// image of W columns is cut into ceil(W / TILE_W) tile columns whose widths differ by at most one).  Neighbouring
// pixels share lattice vertices (a 16 x 16 tile of a 321 x 321 image touches ~120 of the image's ~10 000 bilateral
// vertices and ~160 Gaussian ones), so a tile's contribution to a vertex is summed on chip and leaves the CU once.
#ifndef WSC_TILE_W
#define WSC_TILE_W 16
#endif
#ifndef WSC_TILE_H
#define WSC_TILE_H 16
#endif
constexpr int TILE_W = WSC_TILE_W, TILE_H = WSC_TILE_H, TILE_PIX = TILE_W * TILE_H;
static_assert(TILE_PIX <= 256, "an entry's pixel index is stored in one byte");
constexpr int SLOT_ENT = 32;       // entries per slot (bounds the serial chain of one lane group)
constexpr int SORT_MAX = 2048;     // >= TILE_PIX * 6 entries of a bilateral tile, power of two
static_assert(TILE_PIX * 6 <= SORT_MAX && SORT_MAX <= 65535, "tile too large for the in-LDS grouping");

struct TileGeom {
    int H, W, ntx, nty, tpi; // tile columns / rows / tiles per image
};
__host__ __device__ inline TileGeom make_geom(int H, int W) {
    TileGeom g;
    g.H = H; g.W = W;
    g.ntx = (W + TILE_W - 1) / TILE_W;
    g.nty = (H + TILE_H - 1) / TILE_H;
    g.tpi = g.ntx * g.nty;
    return g;
}
struct TileBox {
    int x0, y0, cw, ch; // origin and size of the tile
    int ebase;          // pixels of the image in tiles before this one (tile-major pixel order)
};
__host__ __device__ inline TileBox tile_box(const TileGeom &g, int j) {
    // 32-bit arithmetic (tx * W < 2^32): this runs in every block of the iteration kernels
    const unsigned ty = (unsigned)j / (unsigned)g.ntx, tx = (unsigned)j - ty * (unsigned)g.ntx;
    TileBox b;
    b.x0 = (int)(tx * (unsigned)g.W / (unsigned)g.ntx);
    b.y0 = (int)(ty * (unsigned)g.H / (unsigned)g.nty);
    b.cw = (int)((tx + 1) * (unsigned)g.W / (unsigned)g.ntx) - b.x0;
    b.ch = (int)((ty + 1) * (unsigned)g.H / (unsigned)g.nty) - b.y0;
    b.ebase = b.y0 * g.W + b.x0 * b.ch;
    return b;
}
// t / cw for t < TILE_PIX, cw <= TILE_W as a multiply and a shift (exact for cw <= 26, t < 4096)
static_assert(TILE_W <= 26 && TILE_PIX <= 4096, "tile_div_magic range");
__host__ __device__ inline unsigned tile_div_magic(int cw) { return (65536u + (unsigned)cw - 1u) / (unsigned)cw; }

} // namespace

struct wsc_crf {
    wsc_ctx *ctx = nullptr;
    int B = 0, H = 0, W = 0, N = 0;
    LatticeDev lat[2]; // 0: Gaussian (d=2), 1: bilateral (d=5)
    // per pixel, 20 dwords = five 16-byte loads: offG[3] offB[6] baryG[3] baryB[6] normG normB
    uint4 *pix_rec = nullptr;
    // per pixel, 13 dwords (52 bytes): the bilateral part alone -- offB[6] baryB[6] normB -- for the updates that start from
    // the Gaussian message kernel's E (GF variants); built when the Gaussian lattice has its tile vertex sets
    uint32_t *pix_rec_b = nullptr;
    std::vector<void *> allocs;
    void *gauss_entry = nullptr;        // the ctx's GaussCache entry lat[0] aliases (its `users` count is ours to drop)
    std::vector<void *> persist_allocs; // blocks of a cached Gaussian lattice under construction (freed if the build fails)
    bool persist = false; // allocations made while set belong to the ctx (cached Gaussian lattice)
    // set by wsc_crf_inference on a ctx other than the build ctx: the last loop's completion on that stream
    hipEvent_t use_ev = nullptr;
    bool used_elsewhere = false;
};

namespace {

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

// Pack the d hashed lattice coordinates into 64 bits.  d=2: 32 bits each; d=5: 12 bits each.
template <int D>
__device__ __forceinline__ bool pack_key(const int *key, unsigned long long &out) {
    if (D == 2) {
        out = ((unsigned long long)(unsigned)(key[0] + 0x40000000) << 32) | (unsigned)(key[1] + 0x40000000);
        return true;
    }
    unsigned long long k = 0;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const int v = key[i] + 2048;
        ok = ok && v >= 0 && v < 4096;
        k = (k << 12) | (unsigned long long)(v & 4095);
    }
    out = k;
    return ok;
}
template <int D>
__device__ __forceinline__ void unpack_key(unsigned long long k, int *key) {
    if (D == 2) {
        key[0] = (int)(unsigned)(k >> 32) - 0x40000000;
        key[1] = (int)(unsigned)(k & 0xffffffffu) - 0x40000000;
        return;
    }
#pragma unroll
    for (int i = D - 1; i >= 0; --i) {
        key[i] = (int)(k & 4095) - 2048;
        k >>= 12;
    }
}

// Linear probing, at most HASH_MAX_PROBE slots: returns -1 when the run is longer (the table is too
// full -- the caller flags it and the lattice is rebuilt with a worst-case-sized table).
constexpr int HASH_MAX_PROBE = 256;
__device__ __forceinline__ int hash_insert(unsigned long long *table, unsigned mask, unsigned long long key) {
    unsigned slot = (unsigned)mix64(key) & mask;
    for (int probe = 0; probe < HASH_MAX_PROBE; ++probe) {
        const unsigned long long prev = atomicCAS(&table[slot], EMPTY_KEY, key);
        if (prev == EMPTY_KEY || prev == key) return (int)slot;
        slot = (slot + 1) & mask;
    }
    return -1;
}
__device__ __forceinline__ int hash_lookup(const unsigned long long *table, unsigned mask, unsigned long long key) {
    unsigned slot = (unsigned)mix64(key) & mask;
    for (;;) {
        const unsigned long long cur = table[slot];
        if (cur == key) return (int)slot;
        if (cur == EMPTY_KEY) return -1;
        slot = (slot + 1) & mask;
    }
}

// ---- wave-aggregated atomics ---------------------------------------------------------------
// Neighbouring pixels share lattice vertices (57 pixels per bilateral vertex on the bench images),
// so the lanes of a wave mostly hit the same few table slots / rows.  wave_match returns, for
// every active lane, the mask of active lanes holding the same (key, tag); the lowest lane of
// a mask acts for the group: one CAS / atomic per distinct key per wave instead of one per lane.
__device__ __forceinline__ unsigned long long wave_match(unsigned long long key, int tag) {
    unsigned long long remaining = __ballot(1);
    unsigned long long mine = 0;
    while (remaining) {
        const int leader = __ffsll((long long)remaining) - 1;
        const unsigned klo = __shfl((unsigned)key, leader, 64);
        const unsigned khi = __shfl((unsigned)(key >> 32), leader, 64);
        const int t = __shfl(tag, leader, 64);
        const bool same = (unsigned)key == klo && (unsigned)(key >> 32) == khi && tag == t;
        const unsigned long long m = __ballot(same);
        if (same) mine = m;
        remaining &= ~m;
    }
    return mine;
}
// 32-bit keys (row ids): one shuffle per distinct key instead of three
__device__ __forceinline__ unsigned long long wave_match32(unsigned key) {
    unsigned long long remaining = __ballot(1);
    unsigned long long mine = 0;
    while (remaining) {
        const int leader = __ffsll((long long)remaining) - 1;
        const bool same = key == (unsigned)__shfl((int)key, leader, 64);
        const unsigned long long m = __ballot(same);
        if (same) mine = m;
        remaining &= ~m;
    }
    return mine;
}
// (Matching only RUNS of adjacent lanes -- three shuffles and two ballots instead of one loop trip per
// distinct key -- was measured: the lattice build went from 2.6 to 4.5 ms per 32-image batch; equal keys
// are interleaved across the wave, not adjacent, and the extra atomics cost far more than the loop.)
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// XCD-contiguous work split for the gather kernels.  Blocks are dispatched round-robin over the 8
// XCDs (block b -> XCD b % 8); a plain grid-stride loop therefore makes every XCD touch every
// image, and the lattice rows a pixel gathers never stay in that XCD's 4 MB L2.  Here block b gets
// the logical id that puts the blocks of one XCD next to each other, and each block owns ONE
// contiguous range of items, so an XCD sweeps a contiguous window of pixels / lattice rows whose
// neighbours are in its own L2 (and, within a block, in the CU's L1).  Pure speed: any placement
// computes the same result.
__device__ __forceinline__ void xcd_range(long long total, long long &begin, long long &end) {
    const int nb = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nb >> 3, r = nb & 7;
    const long long lb = (xcd < r ? (long long)xcd * (q + 1) : (long long)r * (q + 1) + (long long)(xcd - r) * q) + (bid >> 3);
    const long long per = (total + nb - 1) / nb;
    begin = lb * per;
    end = begin + per < total ? begin + per : total;
}

// logical block id that puts the blocks of one XCD next to each other (see xcd_range): for kernels with one block per pixel
// tile, the tiles of an image -- which share its lattice rows, hash table and slot -> row map -- then run on one XCD's L2
__device__ __forceinline__ int xcd_block_id() {
    const int nb = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nb >> 3, r = nb & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// Replicated form: the grid is `rep` equal groups of blocks (gridDim.x % rep == 0); group k works on
// replica k's [0, n_local) items.  The replica is uniform over the block, so its pointer offsets are
// scalar and the loop bodies are the same as in the unreplicated case (rep = 1 reduces to xcd_range).
__device__ __forceinline__ void xcd_range_rep(long long n_local, int rep, int &k, long long &begin, long long &end) {
    const int nb = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nb >> 3, r = nb & 7;
    const int lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int bpr = nb / rep;
    k = lb / bpr;
    const int j = lb - k * bpr;
    const long long per = (n_local + bpr - 1) / bpr;
    begin = (long long)j * per;
    end = begin + per < n_local ? begin + per : n_local;
}

struct EmbedArgs {
    const uint8_t *rgb; // [B][N][3]
    int B, H, W, N;
    float inv_sxy_den, inv_srgb_den; // sxy, srgb (divisors, applied with '/')
    float scale[5];
    unsigned long long *table; // [B][cap]
    unsigned cap_mask;
    long long cap;
    float *bary;     // [B*N*(d+1)]
    int32_t *first;  // [B*cap] smallest entry index touching the slot
    int *err;        // key range error flag
};

// Elevate a pixel's feature vector, find its simplex and barycentric weights (exactly as Permutohedral::init does):
// the d+1 packed vertex keys and weights of pixel gp = (x, y).  Returns false when a key leaves the packed range.
template <int D>
__device__ __forceinline__ bool embed_pixel(const EmbedArgs &a, long long gp, int x, int y, unsigned long long *pk_out,
                                            float *bary_out) {

    float f[D];
    f[0] = (float)x / a.inv_sxy_den;
    f[1] = (float)y / a.inv_sxy_den;
    if (D == 5) {
        const uint8_t *px = a.rgb + gp * 3;
        f[2] = (float)(int)px[0] / a.inv_srgb_den;
        f[3] = (float)(int)px[1] / a.inv_srgb_den;
        f[4] = (float)(int)px[2] / a.inv_srgb_den;
    }
    float elevated[D + 1], rem0[D + 1];
    int rank[D + 1];
    float sm = 0.f;
#pragma unroll
    for (int j = D; j > 0; --j) {
        const float cf = f[j - 1] * a.scale[j - 1];
        elevated[j] = sm - (float)j * cf;
        sm += cf;
    }
    elevated[0] = sm;

    const float down_factor = 1.0f / (float)(D + 1);
    const float up_factor = (float)(D + 1);
    int sum = 0;
#pragma unroll
    for (int i = 0; i <= D; ++i) {
        const float v = down_factor * elevated[i];
        const float up = ceilf(v) * up_factor;
        const float down = floorf(v) * up_factor;
        int rd2;
        if (up - elevated[i] < elevated[i] - down) rd2 = (int)(short)up;
        else rd2 = (int)(short)down;
        rem0[i] = (float)rd2;
        sum = (int)((float)sum + (float)rd2 * down_factor); // int += float, truncating
    }
#pragma unroll
    for (int i = 0; i <= D; ++i) rank[i] = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const float di = elevated[i] - rem0[i];
#pragma unroll
        for (int j = i + 1; j <= D; ++j) {
            if (di < elevated[j] - rem0[j]) rank[i]++;
            else rank[j]++;
        }
    }
#pragma unroll
    for (int i = 0; i <= D; ++i) {
        rank[i] += sum;
        if (rank[i] < 0) {
            rank[i] += D + 1;
            rem0[i] += (float)(D + 1);
        } else if (rank[i] > D) {
            rank[i] -= D + 1;
            rem0[i] -= (float)(D + 1);
        }
    }
    float bary[D + 2];
#pragma unroll
    for (int i = 0; i <= D + 1; ++i) bary[i] = 0.f;
#pragma unroll
    for (int i = 0; i <= D; ++i) {
        const float v = (elevated[i] - rem0[i]) * down_factor;
        // bary[D - rank[i]] += v; bary[D - rank[i] + 1] -= v;   (static indexing via selects)
#pragma unroll
        for (int s = 0; s <= D + 1; ++s) {
            if (s == D - rank[i]) bary[s] += v;
            if (s == D - rank[i] + 1) bary[s] -= v;
        }
    }
    bary[0] = (float)((double)bary[0] + (1.0 + (double)bary[D + 1]));

    bool ok = true;
#pragma unroll
    for (int r = 0; r <= D; ++r) {
        int key[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            // canonical[r][rank[i]] = rank[i] <= D - r ? r : r - (D+1)
            const int can = rank[i] <= D - r ? r : r - (D + 1);
            key[i] = (int)(short)(rem0[i] + (float)can);
        }
        ok = pack_key<D>(key, pk_out[r]) && ok;
        bary_out[r] = bary[r];
    }
    return ok;
}

// Vertex ids in first-touch raster order (the CPU reference's insertion order): every occupied table slot knows the smallest
// entry index that touches its vertex (`first`, an atomicMin of the embed pass).  One bit per ENTRY marks the first touchers
// (set from the table side: 4 M slots instead of 19.8 M entries), a popcount scan over the 32-bit words ranks them, and a
// second pass over the table hands every vertex its row = 1 + rank(first).  (Round 1 flagged, scanned and re-read all
// entries: three 80 MB passes where these move ~70 MB in total.)
__global__ void first_bits_kernel(const int32_t *__restrict__ first, long long cap, int per_img, unsigned *__restrict__ bitmap) {
    const int b = blockIdx.y;
    const int32_t *fi = first + (long long)b * cap;
    for (long long sl = (long long)blockIdx.x * blockDim.x + threadIdx.x; sl < cap; sl += (long long)gridDim.x * blockDim.x) {
        const int f = fi[sl];
        if (f != 0x7fffffff) {
            const long long g = (long long)b * per_img + f;
            atomicOr(&bitmap[g >> 5], 1u << (unsigned)(g & 31));
        }
    }
}
__global__ void popc_words_kernel(const unsigned *__restrict__ bitmap, long long nw, unsigned *__restrict__ cnt) {
    for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += (long long)gridDim.x * blockDim.x)
        cnt[w] = (unsigned)__popc(bitmap[w]);
}
__device__ __forceinline__ unsigned first_rank(const unsigned *bitmap, const unsigned *wprefix, long long g) {
    return wprefix[g >> 5] + (unsigned)__popc(bitmap[g >> 5] & ((1u << (unsigned)(g & 31)) - 1u));
}
// vertices of the images before image b (bound[b]), one thread per image
__global__ void image_bounds_kernel(const unsigned *__restrict__ bitmap, const unsigned *__restrict__ wprefix, int per_img, int B,
                                    unsigned *__restrict__ bound) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) bound[b] = first_rank(bitmap, wprefix, (long long)b * per_img);
}
// row ranges of the images: rows [1 + bound[b], 1 + bound[b + 1]) (row 0 is the zero row), the last one ends at `rows`
__global__ void image_rows_kernel(const unsigned *__restrict__ bound, int B, int rows, int32_t *__restrict__ img_row) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) img_row[b] = 1 + (int)bound[b];
    if (b == B) img_row[b] = rows;
}
// occupied slots publish their row id, key and image
__global__ void assign_rows_kernel(const int32_t *__restrict__ first, const unsigned long long *__restrict__ table, long long cap,
                                   int per_img, const unsigned *__restrict__ bitmap, const unsigned *__restrict__ wprefix,
                                   int32_t *__restrict__ slot2row, unsigned long long *__restrict__ rowkey,
                                   int32_t *__restrict__ rowimg) {
    const int b = blockIdx.y;
    for (long long sl = (long long)blockIdx.x * blockDim.x + threadIdx.x; sl < cap; sl += (long long)gridDim.x * blockDim.x) {
        const long long s_ = (long long)b * cap + sl;
        const int f = first[s_];
        if (f == 0x7fffffff) continue;
        const int row = 1 + (int)first_rank(bitmap, wprefix, (long long)b * per_img + f);
        slot2row[s_] = row;
        rowkey[row] = table[s_];
        rowimg[row] = b;
    }
}

// ---- flat exclusive scan (3 kernels), unsigned 32-bit -------------------------------
constexpr int SCAN_CHUNK = 4096; // elements per block (256 threads x 16)

__device__ __forceinline__ unsigned block_exclusive_scan_256(unsigned v, unsigned *lds, unsigned &total) {
    // inclusive wave scan
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned x = v;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) lds[wv] = x;
    __syncthreads();
    unsigned base = 0;
    for (int i = 0; i < wv; ++i) base += lds[i];
    total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + x - v;
}

__global__ __launch_bounds__(256) void scan_reduce_kernel(const unsigned *__restrict__ in, long long n,
                                                          unsigned *__restrict__ sums) {
    __shared__ unsigned lds[4];
    const long long base = (long long)blockIdx.x * SCAN_CHUNK;
    unsigned s = 0;
    for (int i = 0; i < 16; ++i) {
        const long long idx = base + i * 256 + threadIdx.x;
        if (idx < n) s += in[idx];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

// single block: exclusive scan of sums[0..m) in place; writes the grand total to sums[m]
__global__ __launch_bounds__(256) void scan_sums_kernel(unsigned *__restrict__ sums, int m) {
    __shared__ unsigned lds[4];
    unsigned carry = 0;
    for (int base = 0; base < m; base += 256) {
        const int idx = base + threadIdx.x;
        const unsigned v = idx < m ? sums[idx] : 0u;
        unsigned total;
        const unsigned ex = block_exclusive_scan_256(v, lds, total);
        if (idx < m) sums[idx] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) sums[m] = carry;
}

__global__ __launch_bounds__(256) void scan_apply_kernel(const unsigned *__restrict__ in, long long n,
                                                         const unsigned *__restrict__ sums, unsigned *__restrict__ out) {
    __shared__ unsigned lds[4];
    const long long base = (long long)blockIdx.x * SCAN_CHUNK;
    unsigned carry = sums[blockIdx.x];
    // thread t owns 16 consecutive elements [t*16, t*16+16)
    unsigned v[16];
    unsigned s = 0;
    const long long t0 = base + (long long)threadIdx.x * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        v[i] = (t0 + i < n) ? in[t0 + i] : 0u;
        s += v[i];
    }
    unsigned total;
    unsigned ex = carry + block_exclusive_scan_256(s, lds, total);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (t0 + i < n) out[t0 + i] = ex;
        ex += v[i];
    }
}

// scan_apply_kernel that forms its block's carry from the RAW block sums itself (nblk <= 256: one value per thread) -- the
// scans of a lattice build have 4 ... 152 blocks, so the single-block pass over the sums (a launch of its own) is not needed.
// Block 0 leaves the grand total in sums[nblk], where the three-kernel form puts it.
__global__ __launch_bounds__(256) void scan_apply_fused_kernel(const unsigned *__restrict__ in, long long n,
                                                               unsigned *__restrict__ sums, int nblk, unsigned *__restrict__ out) {
    __shared__ unsigned lds[4];
    __shared__ unsigned red[8];
    const long long base = (long long)blockIdx.x * SCAN_CHUNK;
    const unsigned mine = (int)threadIdx.x < nblk ? sums[threadIdx.x] : 0u;
    unsigned c = (int)threadIdx.x < (int)blockIdx.x ? mine : 0u, tot = mine;
    for (int o = 32; o > 0; o >>= 1) {
        c += __shfl_down(c, o, 64);
        tot += __shfl_down(tot, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = c;
        red[4 + (threadIdx.x >> 6)] = tot;
    }
    __syncthreads();
    const unsigned carry = red[0] + red[1] + red[2] + red[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) sums[nblk] = red[4] + red[5] + red[6] + red[7];
    unsigned v[16];
    unsigned s = 0;
    const long long t0 = base + (long long)threadIdx.x * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        v[i] = (t0 + i < n) ? in[t0 + i] : 0u;
        s += v[i];
    }
    unsigned total;
    unsigned ex = carry + block_exclusive_scan_256(s, lds, total);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (t0 + i < n) out[t0 + i] = ex;
        ex += v[i];
    }
}

int exclusive_scan(wsc_ctx *ctx, const unsigned *in, long long n, unsigned *out, unsigned *sums /* nblk+1 */) {
    const int nblk = (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nblk), dim3(256), 0, ctx->stream, in, n, sums);
    if (nblk <= 256) {
        hipLaunchKernelGGL(scan_apply_fused_kernel, dim3(nblk), dim3(256), 0, ctx->stream, in, n, sums, nblk, out);
        WSC_HIP(hipGetLastError());
        return WSC_OK;
    }
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, ctx->stream, sums, nblk);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nblk), dim3(256), 0, ctx->stream, in, n, sums, out);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

// ---- splat tables of a pixel tile (lattice build) -------------------------------------------------------
// Block-wide inclusive scan of an int array in LDS (n <= SORT_MAX, 256 threads, 8 consecutive elements per
// thread).  MAXOP: running maximum instead of running sum.
template <bool MAXOP, class T = int>
__device__ __forceinline__ void block_scan_lds(T *v, int n, int *wtot /* [4] */) {
    constexpr int PER = SORT_MAX / 256;
    const int t0 = threadIdx.x * PER;
    int loc[PER];
    int run = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int x = t0 + i < n ? (int)v[t0 + i] : 0;
        run = MAXOP ? max(run, x) : run + x;
        loc[i] = run;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int x = run;
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x = MAXOP ? max(x, y) : x + y;
    }
    if (lane == 63) wtot[wv] = x;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < wv; ++i) base = MAXOP ? max(base, wtot[i]) : base + wtot[i];
    int prev = __shfl_up(x, 1, 64); // inclusive result of the previous thread of this wave
    if (lane == 0) prev = 0;
    const int carry = MAXOP ? max(base, prev) : base + prev;
#pragma unroll
    for (int i = 0; i < PER; ++i)
        if (t0 + i < n) v[t0 + i] = (T)(MAXOP ? max(carry, loc[i]) : carry + loc[i]);
    __syncthreads();
}

// Grouped rows of a tile in LDS -> flag[i] = 1 where a slot starts (a new row, or SLOT_ENT entries into a row's
// run); on return aux[i] holds the inclusive count of slot starts (aux[ne-1] = slots of the tile) and seg[i] the
// first entry of i's run of equal rows.
__device__ __forceinline__ void tile_slot_flags(const int *srow, int ne, unsigned short *aux, unsigned short *flag, unsigned short *seg,
                                                int *wtot) {
    for (int i = threadIdx.x; i < ne; i += 256) seg[i] = (unsigned short)((i == 0 || srow[i] != srow[i - 1]) ? i : 0);
    __syncthreads();
    block_scan_lds<true>(seg, ne, wtot); // seg[i] = start of the run of equal rows containing i
    for (int i = threadIdx.x; i < ne; i += 256) {
        flag[i] = ((i - (int)seg[i]) % SLOT_ENT == 0) ? 1 : 0;
        aux[i] = flag[i];
    }
    __syncthreads();
    block_scan_lds<false>(aux, ne, wtot);
}

// Pass 1 of the lattice build, one block per pixel tile (one thread per pixel): embed the tile's pixels, bring their
// d+1 entries each (pixel t of the tile, vertex rank r; index e = t*(d+1)+r) into groups of equal lattice vertex, keeping
// the entries of a group in index order, write them out tile-major, and insert every DISTINCT vertex of the tile into
// the image's global hash table once (a 16 x 16 tile touches ~120 bilateral vertices with its 1536 entries: 13x fewer
// global compare-and-swaps / atomicMins than one per entry).  The index order inside a group makes the summation
// order of every splat slot a function of the image alone -- not of the batch it is in, nor of an atomic's arrival
// order.  (The ORDER OF THE GROUPS within the tile is that of an LDS hash table and may differ from run to run: it only
// decides where a slot's partial row lives, never a value.)
//   A  keys -> slots of an LDS hash table (64-bit compare-and-swap), per slot the smallest global entry index
//   B  stable rank of every entry inside its group.  A group holds at most ONE entry per pixel (the d+1 vertices of a simplex
//      are distinct), so index order inside a group is pixel order and rank = number of lower pixels of the tile that touch
//      the vertex: every group gets a TILE_PIX-bit pixel mask (LDS atomicOr: the result does not depend on arrival order),
//      rank = popcount of the mask below the pixel.  ~120 instructions per wave instead of the ~1800 of the ballot-matching
//      walk (wave q walks the q-th quarter of the index range, 64 entries at a time, equal slots matched with ballots),
//      which stays as the path for tiles with more than RANK_GMAX distinct vertices (noise images) or WSC_CRF_RANK_BALLOT=1
//   C  per occupied slot: group size; global insert + atomicMin of the first-touch index (vertex ids are assigned in
//      first-touch raster order, the CPU reference's insertion order)
//   D  scan -> group starts; position = start + entries of the group in earlier quarters + rank
// The LDS hash table has GROUP_HT slots in two sizes.  Everything a block does besides its pixels is a pass over the table
// (clear, compact ids, group sizes, scan), and a 16 x 16 tile of a natural image touches ~120 distinct vertices: the
// first launch runs every tile on a 512-slot table (19 KB of LDS instead of 53: eight blocks per CU instead of three) and
// a tile with more than 0.75 * 512 distinct vertices -- noise -- quits before it has written anything and puts itself on a
// list that a second launch works off with the full 2048-slot table (load <= 0.75 even when every entry has its own vertex).
#ifndef WSC_EMBED_HT
#define WSC_EMBED_HT 512
#endif
constexpr int GROUP_HT_SMALL = WSC_EMBED_HT, GROUP_HT_FULL = SORT_MAX;
constexpr int RANK_MW = (TILE_PIX + 31) / 32;            // mask words per group
template <int D, int GROUP_HT>
__device__ __forceinline__ void tile_embed_body(const EmbedArgs &a, const TileGeom &tg, float *__restrict__ tent_w,
                                                uint8_t *__restrict__ tent_p, int32_t *__restrict__ sslot_out,
                                                unsigned *__restrict__ tile_nslots, int force_ballot, int tile,
                                                int32_t *__restrict__ redo_list, unsigned *__restrict__ redo_count) {
    constexpr int RANK_GMAX = (4 * GROUP_HT * 2 + SORT_MAX * 2) / (RANK_MW * 4); // groups whose masks fit the wcnt + erank storage
    constexpr bool SMALL = GROUP_HT < GROUP_HT_FULL;
    constexpr int dp1 = D + 1;
    __shared__ unsigned long long table[GROUP_HT]; // vertex key, later the vertex's global hash slot
    __shared__ int start[GROUP_HT];                // first-touch entry index, then group size, then group start
    // ballot path: wcnt[4][GROUP_HT] + erank[SORT_MAX] (unsigned short); mask path: RANK_GMAX pixel masks in the same bytes
    __shared__ __attribute__((aligned(16))) unsigned short rank_store[4 * GROUP_HT + SORT_MAX];
    unsigned short(*wcnt)[GROUP_HT] = reinterpret_cast<unsigned short(*)[GROUP_HT]>(rank_store);
    unsigned short *erank = rank_store + 4 * GROUP_HT;
    unsigned *gmask = reinterpret_cast<unsigned *>(rank_store); // [group][RANK_MW]
    __shared__ unsigned short eslot[SORT_MAX];
    __shared__ unsigned short cid[GROUP_HT]; // compact group id of an occupied slot (mask path)
    __shared__ int wtot[4];
    __shared__ int n_groups_s;
    __shared__ int full_s; // SMALL: a probe walked the whole table
    const int b = tile / tg.tpi, j = tile - b * tg.tpi;
    const TileBox tb = tile_box(tg, j);
    const int N = tg.H * tg.W;
    const int np = tb.cw * tb.ch, ne = np * dp1;
    for (int i = threadIdx.x; i < GROUP_HT; i += 256) {
        table[i] = EMPTY_KEY;
        start[i] = 0x7fffffff;
    }
    for (int i = threadIdx.x; i < (4 * GROUP_HT + SORT_MAX) / 2; i += 256) gmask[i] = 0u; // wcnt + erank / the pixel masks
    if (threadIdx.x == 0) full_s = 0;
    __syncthreads();
    // ---- A: this thread's pixel
    const int t = threadIdx.x;
    const bool have = t < np;
    const int ty = have ? t / tb.cw : 0, tx = have ? t - ty * tb.cw : 0;
    const int y = tb.y0 + ty, x = tb.x0 + tx;
    const int n = y * tg.W + x;
    const long long gp = (long long)b * N + n;
    unsigned long long pk[dp1];
    float bary[dp1];
    if (have) {
        if (!embed_pixel<D>(a, gp, x, y, pk, bary)) atomicOr(a.err, 1);
#pragma unroll
        for (int r = 0; r < dp1; ++r) {
            unsigned sl = (unsigned)(mix64(pk[r]) >> 40) & (GROUP_HT - 1);
            for (int probes = 0;; ++probes) {
                const unsigned long long old = atomicCAS(&table[sl], EMPTY_KEY, pk[r]);
                if (old == EMPTY_KEY || old == pk[r]) break;
                if (SMALL && probes >= GROUP_HT) { // (the full table always has a free slot: 1536 entries at most)
                    full_s = 1;
                    break;
                }
                sl = (sl + 1) & (GROUP_HT - 1);
            }
            eslot[t * dp1 + r] = (unsigned short)sl;
            atomicMin(&start[sl], n * dp1 + r);
        }
    }
    __syncthreads();
    // ---- compact ids of the occupied slots (thread i owns slots [8 i, 8 i + 8): count, block prefix, number)
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    {
        constexpr int PER = GROUP_HT / 256;
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) cnt += table[threadIdx.x * PER + k] != EMPTY_KEY ? 1 : 0;
        int x = cnt;
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wtot[q] = x;
        __syncthreads();
        int base = x - cnt;
        for (int i = 0; i < q; ++i) base += wtot[i];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int sl = threadIdx.x * PER + k;
            const bool occ = table[sl] != EMPTY_KEY;
            cid[sl] = (unsigned short)base;
            base += occ ? 1 : 0;
        }
        if (threadIdx.x == 255) n_groups_s = base;
        __syncthreads();
    }
    if (SMALL && (full_s || n_groups_s > GROUP_HT * 3 / 4)) { // nothing written yet: the tile goes to the full-table launch
        if (threadIdx.x == 0) redo_list[atomicAdd(redo_count, 1u)] = tile;
        return;
    }
    const bool use_mask = !force_ballot && n_groups_s <= RANK_GMAX;
    const int Q = ((ne + 3) / 4 + 63) / 64 * 64; // ballot path: entries per quarter, whole wave trips
    // ---- B: stable ranks
    if (use_mask) {
        if (have) {
#pragma unroll
            for (int r = 0; r < dp1; ++r)
                atomicOr(&gmask[(unsigned)cid[eslot[t * dp1 + r]] * RANK_MW + (t >> 5)], 1u << (t & 31));
        }
    } else {
        for (int c = q * Q; c < min(ne, (q + 1) * Q); c += 64) {
            const int e = c + lane;
            const bool valid = e < ne;
            const unsigned sl = valid ? eslot[e] : 0u;
            unsigned long long remaining = __ballot(valid);
            unsigned long long mine = 0;
            while (remaining) {
                const int leader = __ffsll((long long)remaining) - 1;
                const bool same = valid && sl == (unsigned)__shfl((int)sl, leader, 64);
                const unsigned long long m = __ballot(same);
                if (same) mine = m;
                remaining &= ~m;
            }
            if (valid) {
                const int leader = __ffsll((long long)mine) - 1;
                int base = 0;
                if (lane == leader) {
                    base = wcnt[q][sl];
                    wcnt[q][sl] = (unsigned short)(base + __popcll(mine));
                }
                base = __shfl(base, leader, 64);
                erank[e] = (unsigned short)(base + __popcll(mine & ((1ull << lane) - 1ull)));
            }
        }
    }
    __syncthreads();
    // ---- C: groups -> global table
    unsigned long long *gtable = a.table + (long long)b * a.cap;
    int32_t *gfirst = a.first + (long long)b * a.cap;
    int my_slots = 0;
    for (int i = threadIdx.x; i < GROUP_HT; i += 256) {
        int tot = 0, c0 = 0, c1 = 0, c2 = 0;
        if (use_mask) {
            if (table[i] != EMPTY_KEY) {
#pragma unroll
                for (int w = 0; w < RANK_MW; ++w) tot += __popc(gmask[(unsigned)cid[i] * RANK_MW + w]);
            }
        } else {
            c0 = wcnt[0][i]; c1 = wcnt[1][i]; c2 = wcnt[2][i];
            tot = c0 + c1 + c2 + wcnt[3][i];
        }
        if (tot > 0) {
            int gs = hash_insert(gtable, a.cap_mask, table[i]);
            if (gs >= 0) atomicMin(&gfirst[gs], start[i]);
            else {
                atomicOr(a.err, 2); // table too small: the build is repeated with the worst-case size
                gs = 0;
            }
            table[i] = (unsigned long long)(unsigned)gs;
        }
        start[i] = tot;
        if (!use_mask) {
            wcnt[0][i] = 0;
            wcnt[1][i] = (unsigned short)c0;
            wcnt[2][i] = (unsigned short)(c0 + c1);
            wcnt[3][i] = (unsigned short)(c0 + c1 + c2);
        }
        my_slots += (tot + SLOT_ENT - 1) / SLOT_ENT;
    }
    for (int o = 32; o > 0; o >>= 1) my_slots += __shfl_down(my_slots, o, 64);
    __syncthreads();
    if (lane == 0) wtot[q] = my_slots;
    __syncthreads();
    if (threadIdx.x == 0) tile_nslots[tile] = (unsigned)(wtot[0] + wtot[1] + wtot[2] + wtot[3]);
    __syncthreads();
    // ---- D: positions
    block_scan_lds<false>(start, GROUP_HT, wtot); // inclusive
    if (have) {
        const long long ebase = ((long long)b * N + tb.ebase) * dp1;
#pragma unroll
        for (int r = 0; r < dp1; ++r) {
            const int e = t * dp1 + r;
            const int sl = eslot[e];
            const int prev = sl > 0 ? start[sl - 1] : 0; // exclusive start of the group
            int rank;
            if (use_mask) {
                // pixels of the tile below t that touch this vertex
                const unsigned *m = gmask + (unsigned)cid[sl] * RANK_MW;
                rank = __popc(m[t >> 5] & ((1u << (t & 31)) - 1u));
                for (int w = 0; w < (t >> 5); ++w) rank += __popc(m[w]);
            } else {
                rank = (int)wcnt[e / Q][sl] + (int)erank[e];
            }
            const int pos = prev + rank;
            const int gs = (int)(unsigned)table[sl];
            tent_w[ebase + pos] = bary[r];
            tent_p[ebase + pos] = (uint8_t)t;
            // global hash slot (< 2^28, checked by the host) | the entry's index r among its pixel's d+1 vertices: the slot pass
            // turns both into offset[pixel][r] = row (no pixel-major copy of the slots, no separate remap pass)
            sslot_out[ebase + pos] = gs | (r << 28);
            a.bary[gp * dp1 + r] = bary[r];
        }
    }
}
template <int D>
__global__ __launch_bounds__(256) void tile_embed_kernel(EmbedArgs a, TileGeom tg, float *__restrict__ tent_w, uint8_t *__restrict__ tent_p,
                                                         int32_t *__restrict__ sslot_out, unsigned *__restrict__ tile_nslots,
                                                         int force_ballot, int32_t *__restrict__ redo_list,
                                                         unsigned *__restrict__ redo_count) {
    tile_embed_body<D, GROUP_HT_SMALL>(a, tg, tent_w, tent_p, sslot_out, tile_nslots, force_ballot, (int)blockIdx.x, redo_list,
                                       redo_count);
}
// the tiles the first launch gave up on (usually none: the blocks leave at once), or every tile when the list is null
template <int D>
__global__ __launch_bounds__(256) void tile_embed_full_kernel(EmbedArgs a, TileGeom tg, float *__restrict__ tent_w,
                                                              uint8_t *__restrict__ tent_p, int32_t *__restrict__ sslot_out,
                                                              unsigned *__restrict__ tile_nslots, int force_ballot,
                                                              const int32_t *__restrict__ redo_list,
                                                              const unsigned *__restrict__ redo_count, int n_tiles) {
    const int count = redo_list ? (int)*redo_count : n_tiles;
    for (int i = blockIdx.x; i < count; i += gridDim.x) {
        tile_embed_body<D, GROUP_HT_FULL>(a, tg, tent_w, tent_p, sslot_out, tile_nslots, force_ballot,
                                          redo_list ? redo_list[i] : i, nullptr, nullptr);
        __syncthreads(); // (the body's LDS arrays are reused by the next tile)
    }
}

// Pass 2, one block per tile: slot descriptors at their final (compact) index; slots per row counted.
__global__ __launch_bounds__(256) void tile_slots_kernel(const int32_t *__restrict__ sslot_in, const uint8_t *__restrict__ tent_p,
                                                         int32_t *__restrict__ offset, const int32_t *__restrict__ slot2row,
                                                         long long cap, int dp1, TileGeom tg,
                                                         const int32_t *__restrict__ tslot_start, int2 *__restrict__ slot_desc,
                                                         int32_t *__restrict__ slot_row, unsigned *__restrict__ slot_key,
                                                         unsigned *__restrict__ row_nslots) {
    // entry indices and counts of a tile fit 16 bits (SORT_MAX = 2048): 24 KB of LDS per block instead of 41 (6 blocks per CU)
    __shared__ int rows_s[SORT_MAX];
    __shared__ unsigned short aux[SORT_MAX];
    __shared__ unsigned short flag[SORT_MAX];
    __shared__ unsigned short seg[SORT_MAX];
    __shared__ unsigned short pos[SORT_MAX + 2];
    __shared__ int wtot[4];
    const int tile = xcd_block_id(); // (round 5: an image's tiles on one XCD -- its slot -> row map is read once, not by all eight L2s)
    const int b = tile / tg.tpi, j = tile - b * tg.tpi;
    const TileBox tb = tile_box(tg, j);
    const int N = tg.H * tg.W;
    const int ne = tb.cw * tb.ch * dp1;
    const long long ebase = ((long long)b * N + tb.ebase) * dp1;
    const unsigned cw_magic = tile_div_magic(tb.cw);
    // (round 6, measured and rejected: a thread's <= 6 entries requested together, then their rows together -- 155 -> 175 us: the
    // registers cost a block per CU; slice_norm_tile_kernel and slot_ones_kernel did gain from the same change: 160 -> 136, 36 -> 28 us)
    for (int i = threadIdx.x; i < ne; i += 256) {
        const int ss = sslot_in[ebase + i];
        const int row = slot2row[(long long)b * cap + (ss & 0x0fffffff)];
        rows_s[i] = row;
        // offset[pixel][r] = row of the pixel's r-th vertex (Permutohedral::init's offset_ array, pixel-major)
        const unsigned t = tent_p[ebase + i];
        const unsigned ty = (t * cw_magic) >> 16, tx = t - ty * (unsigned)tb.cw;
        const long long p = (long long)b * N + (long long)(tb.y0 + (int)ty) * tg.W + tb.x0 + (int)tx;
        offset[p * dp1 + (ss >> 28)] = row;
    }
    __syncthreads();
    tile_slot_flags(rows_s, ne, aux, flag, seg, wtot);
    const int ns = aux[ne - 1];
    for (int i = threadIdx.x; i < ne; i += 256)
        if (flag[i]) pos[aux[i] - 1] = (unsigned short)i;
    if (threadIdx.x == 0) pos[ns] = (unsigned short)ne;
    __syncthreads();
    // The tile's slots are stored longest first: the lane groups of a wave of the update kernel take consecutive
    // slots and wait for the longest of them (the order of a tile's slots is free -- it only decides which
    // descriptor index a slot gets).
    __shared__ int hist[SLOT_ENT + 1], hcur[SLOT_ENT + 1];
    if (threadIdx.x <= SLOT_ENT) { hist[threadIdx.x] = 0; hcur[threadIdx.x] = 0; }
    __syncthreads();
    for (int s = threadIdx.x; s < ns; s += 256) atomicAdd(&hist[pos[s + 1] - pos[s]], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int n = SLOT_ENT; n >= 0; --n) {
            const int c = hist[n];
            hist[n] = run;
            run += c;
        }
    }
    __syncthreads();
    const int s0 = tslot_start[tile];
    for (int s = threadIdx.x; s < ns; s += 256) {
        const int i = pos[s], n = pos[s + 1] - i;
        const int row = rows_s[i];
        const int dst = s0 + hist[n] + atomicAdd(&hcur[n], 1);
        slot_desc[dst] = make_int2(i | (n << 16), 0);
        slot_row[dst] = row;
        // (tile, run of SLOT_ENT entries inside the tile's group of this row): the slot's identity, whatever index it got
        if (slot_key) slot_key[dst] = (unsigned)tile * 64u + (unsigned)((i - seg[i]) / SLOT_ENT);
        atomicAdd(&row_nslots[row], 1u);
    }
}

// every slot gets a partial row inside its row's range (any order: the combine adds fixed-point integers)
__global__ void slot_dest_kernel(const int32_t *__restrict__ slot_row, int n_slots, const int32_t *__restrict__ row_slot_start,
                                 unsigned *__restrict__ cursor, int2 *__restrict__ slot_desc, uint32_t *__restrict__ part_row) {
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n_slots; s += gridDim.x * blockDim.x) {
        const int row = slot_row[s];
        const int sb = row_slot_start[row];
        const int dst = sb + (int)atomicAdd(&cursor[row], 1u);
        slot_desc[s].y = dst;
        if (part_row) part_row[dst] = (uint32_t)row | ((uint32_t)min(row_slot_start[row + 1] - sb, 255) << 24);
    }
}

// Deterministic partial-row order inside every row (by slot identity): a lattice built this way can sum a row's
// partials in plain fp32, in index order.  Used for the Gaussian lattice (few slots per row, built once per size).
__global__ void dest_inverse_kernel(const int2 *__restrict__ slot_desc, int n_slots, int32_t *__restrict__ dest_slot) {
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n_slots; s += gridDim.x * blockDim.x) dest_slot[slot_desc[s].y] = s;
}
__global__ void row_sort_dest_kernel(const unsigned *__restrict__ slot_key, int32_t *__restrict__ dest_slot,
                                     const int32_t *__restrict__ row_slot_start, int rows, int2 *__restrict__ slot_desc) {
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += gridDim.x * blockDim.x) {
        const int sb = row_slot_start[row], se = row_slot_start[row + 1];
        for (int i = sb + 1; i < se; ++i) { // insertion sort of the row's slots by key
            const int s = dest_slot[i];
            const unsigned key = slot_key[s];
            int j = i - 1;
            while (j >= sb && slot_key[dest_slot[j]] > key) {
                dest_slot[j + 1] = dest_slot[j];
                --j;
            }
            dest_slot[j + 1] = s;
        }
        for (int i = sb; i < se; ++i) slot_desc[dest_slot[i]].y = i;
    }
}

// entry weights: barycentric weight -> weight * norm[pixel]  (the splat input is norm * Q)
__global__ __launch_bounds__(256) void tile_scale_entries_kernel(const float *__restrict__ norm, int dp1, TileGeom tg,
                                                                 float *__restrict__ tent_w, const uint8_t *__restrict__ tent_p) {
    const int tile = blockIdx.x;
    const int b = tile / tg.tpi, j = tile - b * tg.tpi;
    const TileBox tb = tile_box(tg, j);
    const int N = tg.H * tg.W;
    const int ne = tb.cw * tb.ch * dp1;
    const long long ebase = ((long long)b * N + tb.ebase) * dp1;
    const unsigned cw_magic = tile_div_magic(tb.cw);
    for (int i = threadIdx.x; i < ne; i += 256) {
        const int t = (int)tent_p[ebase + i];
        const int ty = (int)(((unsigned)t * cw_magic) >> 16), tx = t - ty * tb.cw;
        const long long p = (long long)b * N + (long long)(tb.y0 + ty) * tg.W + tb.x0 + tx;
        tent_w[ebase + i] = tent_w[ebase + i] * norm[p];
    }
}

// blur neighbours of every row along every axis (Permutohedral::init, second half).  The two neighbours of a vertex along an
// axis sit at key + delta and key - delta, so n1(r) = r' implies n2(r') = r: every row looks up ONE key (a probe of the
// image's hash table + the slot's row: two dependent random reads) and writes its own n1 and the other row's n2; rows
// whose n2 does not exist keep the zero the array was cleared to.
template <int D>
__global__ void neighbors_kernel(const unsigned long long *__restrict__ rowkey, const int32_t *__restrict__ rowimg,
                                 const unsigned long long *__restrict__ table, const int32_t *__restrict__ slot2row,
                                 long long cap, unsigned cap_mask, int rows, int2 *__restrict__ nbr) {
    const int j = blockIdx.y; // blur axis
    // XCD-contiguous row ranges (round 5): rows are numbered image by image, and a row's lookups go to ITS image's hash table
    // (1 MB at 321 x 321) and slot -> row map; with a plain grid-stride loop every XCD probed every image's table -- 417 MB
    // fetched per build for 42 MB of neighbour ids (r05_pmc_hbm_traffic.txt) -- now an XCD's blocks sweep a contiguous window
    // of rows, i.e. a few images whose tables stay in that XCD's 4 MB L2.  Any placement writes the same ids.
    long long rbeg, rend;
    xcd_range(rows, rbeg, rend);
    for (int row = (int)rbeg + (int)threadIdx.x; row < (int)rend; row += (int)blockDim.x) {
        const long long i = (long long)j * rows + row;
        int *nb = reinterpret_cast<int *>(nbr);
        if (row == 0) {
            nb[2 * i] = 0; // (row 0 is nobody's neighbour: its n2 stays cleared)
            continue;
        }
        int key[D], k1[D];
        unpack_key<D>(rowkey[row], key);
#pragma unroll
        for (int k = 0; k < D; ++k) k1[k] = key[k] - 1;
        if (j < D) {
            // static indexing
#pragma unroll
            for (int k = 0; k < D; ++k)
                if (k == j) k1[k] = key[k] + D;
        }
        const int b = rowimg[row];
        const unsigned long long *tb = table + (long long)b * cap;
        const int32_t *s2r = slot2row + (long long)b * cap;
        unsigned long long p1;
        int r1 = 0;
        if (pack_key<D>(k1, p1)) {
            const int s = hash_lookup(tb, cap_mask, p1);
            if (s >= 0) r1 = s2r[s];
        }
        nb[2 * i] = r1;
        if (r1 > 0) nb[2 * ((long long)j * rows + r1) + 1] = row; // this row is r1's neighbour on the other side
    }
}

// ---- iteration kernels ------------------------------------------------------------------

// Splat, tile-gather form: val[row][m] = sum over the pixels p touching the row of w * (norm[p] * Q[p][m]).
// The pixels of a tile are on chip (LDS) when their contributions are needed -- the update kernel below
// has just computed them -- and every SLOT (<= SLOT_ENT entries of one row, fixed order) is summed by one lane
// group in plain fp32 and written out as one partial row.  A row's value is the sum of its slots' partials;
// that sum runs over a list filled through an atomic cursor (arbitrary order), so it is taken in FIXED POINT:
// every partial is rounded to a multiple of 2^-24 (|partial| <= SLOT_ENT * 2.5 = 80 < 2^7 because
// norm <= 1/sqrt(alpha/(d+1)), so it fits an int32) and added as a 64-bit integer.  Integer addition is
// associative: the result does not depend on the order of the list, and the iteration loop is
// bit-reproducible from run to run without sorting and without float atomics.
constexpr float PFIX_SCALE = 16777216.0f;       // 2^24
constexpr float PFIX_INV = 1.0f / 16777216.0f;  // 2^-24 (exact)
static_assert(SLOT_ENT * 2.5f < 127.0f, "slot partials must fit the 2^24 fixed-point int32");

// sum of the partial rows [sb, se) (float4 l of each), 4 loads in flight.  PLAIN: the lattice's partial rows are in
// a deterministic order (row_sort_dest_kernel) and are added in plain fp32, in that order; otherwise in fixed point.
template <bool PLAIN>
__device__ __forceinline__ f32x4_t combine_slots4(const f32x4_t *__restrict__ part, int sb, int se, unsigned LP, unsigned l) {
    f32x4_t first = part[(unsigned)sb * LP + l];
    if (se - sb == 1) return first;
    if (PLAIN) {
        for (int i = sb + 1; i < se; ++i) {
            const f32x4_t v = part[(unsigned)i * LP + l];
#pragma unroll
            for (int k = 0; k < 4; ++k) first[k] += v[k];
        }
        return first;
    }
    long long acc[4] = {0, 0, 0, 0};
    for (int i = sb; i < se; i += 4) {
        f32x4_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = part[(unsigned)min(i + u, se - 1) * LP + l];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u < se) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] += (long long)__float2int_rn(v[u][k] * PFIX_SCALE);
            }
    }
    f32x4_t o = {(float)acc[0] * PFIX_INV, (float)acc[1] * PFIX_INV, (float)acc[2] * PFIX_INV, (float)acc[3] * PFIX_INV};
    return o;
}

// Normalisation pass (splat of the all-ones vector, one value per row): slot partials, then rows.
__global__ __launch_bounds__(256) void slot_ones_kernel(const int32_t *__restrict__ tslot_start, const int2 *__restrict__ slot_desc,
                                                        const float *__restrict__ tent_w, int dp1, TileGeom tg,
                                                        float *__restrict__ part) {
    const int tile = blockIdx.x;
    const int b = tile / tg.tpi, j = tile - b * tg.tpi;
    const TileBox tb = tile_box(tg, j);
    const long long ebase = ((long long)b * tg.H * tg.W + tb.ebase) * dp1;
    // the tile's entry weights through LDS (one coalesced pass): a thread then walks its slot's entries at LDS latency
    // instead of <= 32 dependent trips to L2 (a tile has ~120 slots: half the block's threads each own a chain)
    __shared__ float w[TILE_PIX * 6];
    const int ne = tb.cw * tb.ch * dp1;
    {
        constexpr int EN = (TILE_PIX * 6 + 255) / 256;
        float ew[EN];
#pragma unroll
        for (int k = 0; k < EN; ++k) {
            const int i = (int)threadIdx.x + k * 256;
            ew[k] = i < ne ? tent_w[ebase + i] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < EN; ++k) {
            const int i = (int)threadIdx.x + k * 256;
            if (i < ne) w[i] = ew[k];
        }
    }
    __syncthreads();
    for (int s = tslot_start[tile] + threadIdx.x; s < tslot_start[tile + 1]; s += 256) {
        const int2 d = slot_desc[s];
        const int i0 = d.x & 0xffff, n = d.x >> 16;
        float acc = 0.f;
        for (int i = 0; i < n; ++i) acc += w[i0 + i];
        part[d.y] = acc;
    }
}
__global__ void combine1_kernel(const float *__restrict__ part, const int32_t *__restrict__ row_slot_start, int rows,
                                float *__restrict__ val) {
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += gridDim.x * blockDim.x) {
        long long acc = 0;
        for (int i = row_slot_start[row]; i < row_slot_start[row + 1]; ++i)
            acc += (long long)__float2int_rn(part[i] * PFIX_SCALE);
        val[row] = (float)acc * PFIX_INV;
    }
}

// Rows from slot partials, iteration form (bilateral lattice; the Gaussian lattice combines inside its fused
// blur).  LP lanes per row, rows padded to Mp = 4*LP floats; a row's partials are consecutive.
template <bool PLAIN>
__global__ __launch_bounds__(256) void combine4_kernel(const f32x4_t *__restrict__ part, const int32_t *__restrict__ row_slot_start,
                                                       int LP, int rows_local, int n_slots, int rep, f32x4_t *__restrict__ val) {
    const int rpb = 256 / LP;
    const int tr = threadIdx.x / LP;
    const int l = threadIdx.x - tr * LP;
    if (tr >= rpb) return;
    long long rbeg, rend;
    int rk;
    xcd_range_rep(rows_local, rep, rk, rbeg, rend);
    part += (size_t)rk * n_slots * LP;
    val += (size_t)rk * rows_local * LP;
    for (long long row = rbeg + tr; row < rend; row += rpb) {
        const int sb = row_slot_start[row], se = row_slot_start[row + 1];
        f32x4_t o = {0.f, 0.f, 0.f, 0.f};
        if (se > sb) o = combine_slots4<PLAIN>(part, sb, se, (unsigned)LP, (unsigned)l);
        val[(unsigned)row * (unsigned)LP + l] = o;
    }
}

// The same rows from the same partials, balanced over the PARTIALS instead of the rows (bilateral lattice: ~5 partials per
// row on average, 1 ... 30+ per row -- with a lane group per row a wave waits for its longest row).  A block owns CB_ROWS
// consecutive rows, i.e. one contiguous run of partial rows; its lane groups walk that run in stride (four independent
// 96-byte loads in flight each), find a partial's row by bisection in the block's slice of row_slot_start, and add it into
// the row's 64-bit fixed-point accumulators in LDS (integer atomics: order-independent, so the result is the one of
// combine_slots4<false>, bit for bit; a row's single partial is passed through unconverted as there).
#ifndef WSC_CB_ROWS
#define WSC_CB_ROWS 32 // (A/B on the VOC batch: 128 / 64 / 32 / 16 / 8 rows per block -> 85 / 75 / 67 / 67 / 77 us per launch)
#endif
#ifndef WSC_CB_U
#define WSC_CB_U 4
#endif
constexpr int CB_ROWS = WSC_CB_ROWS;
// part_row != null: a partial row's lattice row (and whether it is the row's only partial) comes from a table written at
// build time -- one coalesced 4-byte load issued together with the partial row itself -- instead of a bisection in the
// block's slice of row_slot_start, and the partial loads no longer wait for that slice to be staged in LDS.
__global__ __launch_bounds__(256) void combine4_balanced_kernel(const f32x4_t *__restrict__ part, const int32_t *__restrict__ row_slot_start,
                                                                const uint32_t *__restrict__ part_row, int LP, int rows,
                                                                f32x4_t *__restrict__ val) {
    __shared__ int rss_l[CB_ROWS + 1];
    __shared__ unsigned long long acc[CB_ROWS * 32];
    const int r0 = blockIdx.x * CB_ROWS;
    const int nr = min(CB_ROWS, rows - r0);
    const int P0 = row_slot_start[r0], P1 = row_slot_start[r0 + nr]; // (uniform: scalar loads)
    for (int i = threadIdx.x; i <= nr; i += 256) rss_l[i] = row_slot_start[r0 + i];
    for (int i = threadIdx.x; i < CB_ROWS * 32; i += 256) acc[i] = 0ull;
    const int gpb = 256 / LP;
    const int tr = threadIdx.x / LP, l = threadIdx.x - tr * LP;
    constexpr int U = WSC_CB_U;
    // the first batch of partial rows travels while the accumulators are cleared
    f32x4_t v[U];
    uint32_t pr[U];
    auto fetch = [&](int p) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pp = p + u * gpb;
            v[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            pr[u] = 0u;
            if (tr < gpb && pp < P1) {
                v[u] = part[(unsigned)pp * (unsigned)LP + l];
                if (part_row) pr[u] = part_row[pp];
            }
        }
    };
    fetch(P0 + tr);
    __syncthreads();
    if (tr < gpb) {
        for (int p = P0 + tr; p < P1; p += U * gpb) {
            if (p != P0 + tr) fetch(p);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pp = p + u * gpb;
                if (pp < P1) {
                    int lo;
                    bool single;
                    if (part_row) {
                        lo = (int)(pr[u] & 0xffffffu) - r0;
                        single = (pr[u] >> 24) == 1u;
                    } else {
                        lo = 0;
                        int hi = nr; // largest row with rss_l[row] <= pp (rows without partials are skipped by the <=)
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (rss_l[mid] <= pp) lo = mid;
                            else hi = mid;
                        }
                        single = rss_l[lo + 1] - rss_l[lo] == 1;
                    }
                    if (single) {
                        val[(unsigned)(r0 + lo) * (unsigned)LP + l] = v[u];
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            atomicAdd(&acc[lo * 32 + 4 * l + k], (unsigned long long)(long long)__float2int_rn(v[u][k] * PFIX_SCALE));
                    }
                }
            }
        }
    }
    __syncthreads();
    if (tr < gpb) {
        for (int row = tr; row < nr; row += gpb) {
            const int n = rss_l[row + 1] - rss_l[row];
            if (n == 1) continue; // written above
            f32x4_t o = {0.f, 0.f, 0.f, 0.f};
            if (n > 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = (float)(long long)acc[row * 32 + 4 * l + k] * PFIX_INV;
            }
            val[(unsigned)(r0 + row) * (unsigned)LP + l] = o;
        }
    }
}

// (Round 6, measured and removed: CHUNKS of 4 / 8 / 16 consecutive partial rows per lane group, summed in registers, a row whose
// partials lie inside one chunk stored directly -- a third of the LDS atomics and no same-address conflicts between
// neighbouring lane groups: 126.8 / 130.7 / 138.9 us against 127.0 us for this kernel + the on-chip blur on the A/B batch
// (profiles/README.md).  The same-address atomics that SQ_LDS_BANK_CONFLICT counts here are not what bounds the kernel.)

// One blur pass along one lattice axis: out[row] = in[row] + 0.5*(in[n1] + in[n2]).
// Scalar form for the normalisation pass (one value per row).
__global__ __launch_bounds__(256) void blur1_kernel(const float *__restrict__ in, const int2 *__restrict__ nbr,
                                                    int rows, float *__restrict__ out) {
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += gridDim.x * blockDim.x) {
        const int2 nb = nbr[row];
        out[row] = row == 0 ? 0.f : in[row] + 0.5f * (in[nb.x] + in[nb.y]);
    }
}

// Iteration form on padded rows: a lane owns one float4 of a row, a block covers floor(256/LP)
// consecutive rows (one contiguous run of `in` / `out`), the two neighbour rows are gathered as
// 16-byte loads.  Each block owns an XCD-contiguous range of rows.
__global__ __launch_bounds__(256) void blur4_kernel(const f32x4_t *__restrict__ in, const int2 *__restrict__ nbr,
                                                    int LP, int rows_local, int rep, f32x4_t *__restrict__ out) {
    const int rpb = 256 / LP;
    const int tr = threadIdx.x / LP;
    const int l = threadIdx.x - tr * LP;
    if (tr >= rpb) return;
    constexpr int U = 4;
    long long rbeg, rend;
    int rk; // replica k: rows [k*rows_local, (k+1)*rows_local) of in / out, shared neighbour table
    xcd_range_rep(rows_local, rep, rk, rbeg, rend);
    in += (size_t)rk * rows_local * LP;
    out += (size_t)rk * rows_local * LP;
    for (long long row0 = rbeg + tr; row0 < rend; row0 += U * rpb) {
        int2 nb[U];
        f32x4_t c[U], a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long row = row0 + u * rpb;
            nb[u] = row < rend ? nbr[row] : make_int2(0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long row = row0 + u * rpb;
            c[u] = in[(unsigned)(row < rend ? row : 0) * (unsigned)LP + l];
            a[u] = in[(unsigned)nb[u].x * (unsigned)LP + l];
            b[u] = in[(unsigned)nb[u].y * (unsigned)LP + l];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long row = row0 + u * rpb;
            if (row < rend) {
                f32x4_t o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = row == 0 ? 0.f : c[u][k] + 0.5f * (a[u][k] + b[u][k]);
                out[(unsigned)row * (unsigned)LP + l] = o;
            }
        }
    }
}

// All d+1 passes of a per-image lattice (the bilateral one) in ONE launch, on chip.  The passes of an image depend on each
// other through the whole image but never on another image or another class, so a workgroup takes ONE image and GW
// classes: the image's rows of those classes as GW float planes in LDS (10.8 k vertices x 3 classes = 130 KB at 321 x 321,
// M = 21: 7 workgroups per image, 224 for a batch of 32 -- one round of the chip), every pass gathers the two neighbours
// of a row from the planes (4-byte LDS reads), the new values wait in registers across a barrier and overwrite the planes.
// The arithmetic per value is blur4_kernel's (c + 0.5 (x1 + x2), same operand order): bit-identical.  HBM sees one read
// and one write of the rows instead of six of each plus twelve neighbour gathers (~27 us a pass at the fabric's rate).
// The workgroups of an image sit on one XCD (they read the same row lines and the same neighbour tables: L2 hits).
constexpr int BL_THREADS = 1024;
// variants {classes per workgroup GW, rows per thread RPT}: an image of `rows` vertices takes the first one with
// rows <= RPT * 1024 and (rows + 1) * GW * 4 bytes <= 160 KB of LDS -- the most classes its rows leave room for (the new
// values of a pass wait in RPT * GW registers next to RPT packed neighbour pairs: no spills at 128 registers)
constexpr int BL_NVAR = 5;
constexpr int BL_VAR[BL_NVAR][2] = {{4, 4}, {4, 10}, {3, 13}, {2, 20}, {1, 26}};
inline int blur_lds_variant(int rows) {
    int v0 = 0;
#ifdef WSC_AB_KNOBS
    if (const char *e = getenv("WSC_BLUR_LDS_MINVAR")) v0 = atoi(e); // A/B: start at a later variant (fewer classes per workgroup)
#endif
    for (int v = v0; v < BL_NVAR; ++v)
        if (rows <= BL_VAR[v][1] * BL_THREADS && (size_t)(rows + 1) * BL_VAR[v][0] * 4 <= 160 * 1024 && rows < 65535) return v;
    return -1;
}
struct BlurLdsArgs {
    const int2 *nbr;        // [(d+1)][rows] global row ids (0 = absent)
    const int32_t *img_row; // [B + 1]
    const int2 *blk;        // [gridDim.x] {image | variant << 24, class group}; x < 0: nothing to do
    float *val;             // [rows][Mp], blurred in place
    int Mp, M, rows, npass;
};
// GW consecutive floats at a 4-byte-aligned address as ONE access (a wave's 64 rows are 64 different cache lines: one
// request per row instead of GW)
template <int GW> struct FloatRun { float v[GW]; };
typedef float f32x3u_t __attribute__((ext_vector_type(3), aligned(4))); // three floats at a 4-byte-aligned address
template <int GW>
__device__ __forceinline__ FloatRun<GW> load_run(const float *q) { // q: 4 GW-byte aligned for GW = 1, 2, 4; 4-byte aligned for GW = 3
    FloatRun<GW> r;
    if constexpr (GW == 4) {
        const f32x4_t x = *reinterpret_cast<const f32x4_t *>(q);
        r.v[0] = x[0]; r.v[1] = x[1]; r.v[2] = x[2]; r.v[3] = x[3];
    } else if constexpr (GW == 3) {
        const f32x3u_t x = *reinterpret_cast<const f32x3u_t *>(q);
        r.v[0] = x[0]; r.v[1] = x[1]; r.v[2] = x[2];
    } else if constexpr (GW == 2) {
        const f32x2_t x = *reinterpret_cast<const f32x2_t *>(q);
        r.v[0] = x[0]; r.v[1] = x[1];
    } else {
        r.v[0] = q[0];
    }
    return r;
}
template <int GW>
__device__ __forceinline__ void store_run(float *q, const FloatRun<GW> &r) {
    if constexpr (GW == 4) *reinterpret_cast<f32x4_t *>(q) = f32x4_t{r.v[0], r.v[1], r.v[2], r.v[3]};
    else if constexpr (GW == 3) {
        f32x3u_t x;
        x[0] = r.v[0]; x[1] = r.v[1]; x[2] = r.v[2];
        *reinterpret_cast<f32x3u_t *>(q) = x;
    } else if constexpr (GW == 2) *reinterpret_cast<f32x2_t *>(q) = f32x2_t{r.v[0], r.v[1]};
    else q[0] = r.v[0];
}
template <int GW, int RPT>
__device__ __forceinline__ void blur_lds_body(const BlurLdsArgs &p, float *pl, int img, int grp) {
    const int r0 = p.img_row[img], nr = p.img_row[img + 1] - r0;
    const int stride = nr + 1; // plane g, local row r: pl[g * stride + r]; index nr of every plane is the zero row
    const int c0 = grp * GW;
    const int tid = (int)threadIdx.x;
    float *vb = p.val + (size_t)r0 * p.Mp + c0;
    unsigned nb[RPT]; // the two neighbours as local row ids (16 bits each; absent -> the zero row nr)
    auto load_idx = [&](int pass) {
        const int2 *q = p.nbr + (size_t)pass * p.rows + r0;
        constexpr int CH = RPT < 10 ? RPT : 10; // raw pairs in flight (registers)
#pragma unroll
        for (int i0 = 0; i0 < RPT; i0 += CH) {
            int2 raw[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int r = tid + (i0 + i) * BL_THREADS;
                raw[i] = (i0 + i < RPT && r < nr) ? q[r] : make_int2(0, 0);
            }
#pragma unroll
            for (int i = 0; i < CH; ++i)
                if (i0 + i < RPT)
                    nb[i0 + i] = (unsigned)(raw[i].x ? raw[i].x - r0 : nr) | ((unsigned)(raw[i].y ? raw[i].y - r0 : nr) << 16);
        }
    };
    load_idx(0);
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = tid + i * BL_THREADS;
        if (r < nr) { // (classes past M are padding: read where the row has them, and dropped)
            FloatRun<GW> run;
            if (GW == 3 && c0 + GW > p.Mp) { // the last class group of a row whose length is no multiple of 3 (Mp = 20: classes
                                             // 18, 19): a 3-float access would run into the next row -- uniform over the workgroup
#pragma unroll
                for (int g = 0; g < GW; ++g) run.v[g] = c0 + g < p.Mp ? vb[(size_t)r * p.Mp + g] : 0.f;
            } else {
                run = load_run<GW>(vb + (size_t)r * p.Mp);
            }
#pragma unroll
            for (int g = 0; g < GW; ++g) pl[g * stride + r] = c0 + g < p.M ? run.v[g] : 0.f;
        }
    }
    if (tid < GW) pl[tid * stride + nr] = 0.f;
    __syncthreads();
    float o[RPT][GW];
    for (int pass = 0; pass < p.npass; ++pass) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = tid + i * BL_THREADS;
            const int rc = r < nr ? r : nr;
            const int l1 = (int)(nb[i] & 0xffffu), l2 = (int)(nb[i] >> 16);
#pragma unroll
            for (int g = 0; g < GW; ++g) o[i][g] = pl[g * stride + rc] + 0.5f * (pl[g * stride + l1] + pl[g * stride + l2]);
        }
        if (pass + 1 < p.npass) load_idx(pass + 1); // travels across the barriers
        else break;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = tid + i * BL_THREADS;
            if (r < nr) {
#pragma unroll
                for (int g = 0; g < GW; ++g) pl[g * stride + r] = o[i][g];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = tid + i * BL_THREADS;
        if (r < nr) {
            if (c0 + GW <= p.M) {
                FloatRun<GW> run;
#pragma unroll
                for (int g = 0; g < GW; ++g) run.v[g] = o[i][g];
                store_run<GW>(vb + (size_t)r * p.Mp, run);
            } else {
#pragma unroll
                for (int g = 0; g < GW; ++g)
                    if (c0 + g < p.M) vb[(size_t)r * p.Mp + g] = o[i][g];
            }
        }
    }
}
__global__ __launch_bounds__(BL_THREADS) void blur_lds_kernel(BlurLdsArgs p) {
    extern __shared__ float pl[];
    const int2 w = p.blk[blockIdx.x];
    if (w.x < 0) return;
    const int img = w.x & 0xffffff;
    switch (w.x >> 24) { // (uniform over the workgroup)
    case 0: blur_lds_body<BL_VAR[0][0], BL_VAR[0][1]>(p, pl, img, w.y); break;
    case 1: blur_lds_body<BL_VAR[1][0], BL_VAR[1][1]>(p, pl, img, w.y); break;
    case 2: blur_lds_body<BL_VAR[2][0], BL_VAR[2][1]>(p, pl, img, w.y); break;
    case 3: blur_lds_body<BL_VAR[3][0], BL_VAR[3][1]>(p, pl, img, w.y); break;
    default: blur_lds_body<BL_VAR[4][0], BL_VAR[4][1]>(p, pl, img, w.y); break;
    }
}

// ---- fused three-pass blur of the Gaussian lattice -----------------------------------------------------
// For d = 2 the lattice points (k0, k1, -k0-k1), k0 = k1 (mod 3), are the integer pairs
//   i = (2 k0 + k1) / 3,  j = (k0 + 2 k1) / 3        (k0 = 2i - j, k1 = 2j - i)
// and the blur neighbours along the three axes (neighbors_kernel) are (i +- 1, j), (i, j +- 1), (i -+ 1, j -+ 1).
// The (i, j) plane is cut into GTI x GTJ tiles; a block loads a tile with a halo of 2 (a GBI x GBJ box of
// points, absent ones as zeros -- exactly what a neighbour pointer to the zero row reads) into LDS, runs the
// three passes there with fixed local offsets (no neighbour table), and writes the GTI x GTJ interior: one read
// and one write of the value array instead of three of each.  Per pass the arithmetic is blur4_kernel's
// (c + 0.5f * (a + b), a + b commutes), so the result is bit-identical.
#ifndef WSC_GTI
#define WSC_GTI 12
#endif
#ifndef WSC_GTJ
#define WSC_GTJ 12
#endif
#ifndef WSC_GLH
#define WSC_GLH 6
#endif
constexpr int GTI = WSC_GTI, GTJ = WSC_GTJ, GBI = GTI + 4, GBJ = GTJ + 4; // tile interior / halo box in i and j

__global__ void gauss_ij_kernel(const unsigned long long *__restrict__ rowkey, int rows, int2 *__restrict__ ij,
                                int *__restrict__ bbox /* imin jmin imax jmax err */) {
    for (int row = 1 + blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += gridDim.x * blockDim.x) {
        int key[2];
        unpack_key<2>(rowkey[row], key);
        const int a = 2 * key[0] + key[1], b = key[0] + 2 * key[1];
        if (a % 3 != 0 || b % 3 != 0) atomicOr(&bbox[4], 1);
        const int i = a / 3, j = b / 3;
        ij[row] = make_int2(i, j);
        atomicMin(&bbox[0], i);
        atomicMin(&bbox[1], j);
        atomicMax(&bbox[2], i);
        atomicMax(&bbox[3], j);
    }
}

__global__ void gauss_tile_fill_kernel(const int2 *__restrict__ ij, int rows, int imin, int jmin, int nti, int ntj,
                                       int32_t *__restrict__ tile_rows, int32_t *__restrict__ occ) {
    for (int row = 1 + blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += gridDim.x * blockDim.x) {
        const int i = ij[row].x - imin, j = ij[row].y - jmin;
        occ[(i / GTI) * ntj + j / GTJ] = 1; // interior owner
        // every tile whose halo box [ti*GTI - 2, ti*GTI + GTI + 2) x [tj*GTJ - 2, ...) contains the point
        for (int ti = (i - GTI - 1 >= 0 ? (i - GTI - 1) / GTI : 0); ti <= (i + 2) / GTI && ti < nti; ++ti) {
            const int li = i - ti * GTI + 2;
            if (li < 0 || li >= GBI) continue;
            for (int tj = (j - GTJ - 1 >= 0 ? (j - GTJ - 1) / GTJ : 0); tj <= (j + 2) / GTJ && tj < ntj; ++tj) {
                const int lj = j - tj * GTJ + 2;
                if (lj < 0 || lj >= GBJ) continue;
                tile_rows[((long long)ti * ntj + tj) * (GBI * GBJ) + li * GBJ + lj] = row;
            }
        }
    }
}

__global__ void gauss_tile_pstart_kernel(const int32_t *__restrict__ tile_rows, const int32_t *__restrict__ row_slot_start,
                                         long long total, int2 *__restrict__ tile_pstart) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int row = tile_rows[i];
        tile_pstart[i] = row ? make_int2(row_slot_start[row], row_slot_start[row + 1] - row_slot_start[row]) : make_int2(0, 0);
    }
}

#ifndef WSC_BLUR3_INPLACE
#define WSC_BLUR3_INPLACE 1
#endif
template <int LH>
__global__ __launch_bounds__(GBI * GBJ) void blur3_tile_kernel(const f32x4_t *__restrict__ in, const int32_t *__restrict__ tile_rows,
                                                         const int32_t *__restrict__ tile_list, int n_occ, int LP,
                                                         int rows_local, int rep, f32x4_t *__restrict__ out,
                                                         const f32x4_t *__restrict__ part, const int2 *__restrict__ tile_pstart,
                                                         int n_slots) {
    // part != null: the input rows are not in memory yet -- every box point sums its row from the slot partials
    // of the splat (combine_slots4, consecutive partial rows) while loading; `in` is unused
    constexpr int P = GBI * GBJ; // one thread per point of the halo box
    // plane stride P + 1: the row-wise load / store phases address (point, float4) with the float4 index fastest, and a
    // stride of exactly P float4s (4096 B) would put a row's LH float4s on one bank
    constexpr int PS = P + 1;
#if WSC_BLUR3_INPLACE
    __shared__ f32x4_t b0[LH * PS];
    f32x4_t *const b1 = b0; // (unused by the in-place passes)
#else
    __shared__ f32x4_t b0[LH * PS], b1[LH * PS];
#endif
    // XCD-contiguous logical block id: neighbouring tiles of one replica (which share halo rows) on one L2
    const int nb = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, qq = nb >> 3, rr = nb & 7;
    const int lb = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int k = lb / n_occ, t = lb - k * n_occ;
    if (in) in += (size_t)k * rows_local * LP;
    if (part) part += (size_t)k * n_slots * LP;
    out += (size_t)k * rows_local * LP;
    const int p = threadIdx.x;
    const int li = p / GBJ, lj = p - li * GBJ;
    const long long tbase = (long long)tile_list[t] * P;
    const int row = tile_rows[tbase + p];
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    if (t == 0 && p < LP) out[p] = zero; // the permanent zero row of this replica
    const bool r0 = row && li >= 1 && li < GBI - 1;      // pass 0 region
    const bool r1 = r0 && lj >= 1 && lj < GBJ - 1;       // pass 1 region
    const bool r2 = row && li >= 2 && li < GBI - 2 && lj >= 2 && lj < GBJ - 2; // interior
    // Memory side: LH consecutive lanes move the LH consecutive float4s of ONE row (a wave instruction touches
    // 64/LH rows of LH*16 contiguous bytes; with a lane per point it touched 64 different rows, 16 bytes each, and
    // the kernel was bound by the L1's line rate).  Item i of thread tid is float4 ll = idx % LH of point
    // pp = idx / LH, idx = i*P + tid; the three passes below stay one thread per point.
    int prow[LH], psb[LH], pse[LH];
#pragma unroll
    for (int i = 0; i < LH; ++i) {
        const int pp = (i * P + p) / LH;
        const int pli = pp / GBJ, plj = pp - pli * GBJ;
        const int r = tile_rows[tbase + pp];
        psb[i] = 0; pse[i] = 0;
        if (part && r) {
            const int2 ps = tile_pstart[tbase + pp];
            psb[i] = ps.x;
            pse[i] = ps.x + ps.y;
        }
        // negative: the point is not written back (halo, or absent)
        prow[i] = (r && pli >= 2 && pli < GBI - 2 && plj >= 2 && plj < GBJ - 2) ? r : (r ? -r : 0);
    }
    for (int lbase = 0; lbase < LP; lbase += LH) {
        if (lbase > 0) __syncthreads(); // the previous group's reads of b0 / b1 are done
        {
            f32x4_t v[LH];
#pragma unroll
            for (int i = 0; i < LH; ++i) {
                const int idx = i * P + p, pp = idx / LH, ll = idx - pp * LH;
                const int r = prow[i] < 0 ? -prow[i] : prow[i];
                v[i] = zero;
                if (r && lbase + ll < LP) {
                    if (part) {
                        // first slot of the row (independent loads for all items); further slots -- tile-border
                        // vertices, ~20 % of the rows -- are added below, in index order
                        if (pse[i] > psb[i]) v[i] = part[(unsigned)psb[i] * (unsigned)LP + lbase + ll];
                    } else {
                        v[i] = in[(unsigned)r * (unsigned)LP + lbase + ll];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < LH; ++i) {
                const int idx = i * P + p, pp = idx / LH, ll = idx - pp * LH;
                if (part && pse[i] - psb[i] > 1 && lbase + ll < LP)
                    v[i] = combine_slots4<true>(part, psb[i], pse[i], (unsigned)LP, (unsigned)(lbase + ll));
                b0[ll * PS + pp] = v[i];
            }
        }
        __syncthreads();
#if WSC_BLUR3_INPLACE
        // ONE LDS plane set: a pass reads its three taps into registers, a barrier, then writes them back in place (two
        // barriers per pass instead of one, but 24.7 KB of LDS per block instead of 49.3: six blocks per CU instead of three
        // for the load phase's dependent gathers to hide behind)
        f32x4_t o[LH];
        // pass 0, axis 0: (i +- 1, j) = p +- GBJ
#pragma unroll
        for (int l = 0; l < LH; ++l) {
            o[l] = zero;
            if (r0) {
                const f32x4_t c = b0[l * PS + p], a = b0[l * PS + p + GBJ], b = b0[l * PS + p - GBJ];
#pragma unroll
                for (int q = 0; q < 4; ++q) o[l][q] = c[q] + 0.5f * (a[q] + b[q]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int l = 0; l < LH; ++l) b0[l * PS + p] = o[l];
        __syncthreads();
        // pass 1, axis 1: (i, j +- 1) = p +- 1
#pragma unroll
        for (int l = 0; l < LH; ++l) {
            o[l] = zero;
            if (r1) {
                const f32x4_t c = b0[l * PS + p], a = b0[l * PS + p - 1], b = b0[l * PS + p + 1];
#pragma unroll
                for (int q = 0; q < 4; ++q) o[l][q] = c[q] + 0.5f * (a[q] + b[q]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int l = 0; l < LH; ++l) b0[l * PS + p] = o[l];
        __syncthreads();
        // pass 2, axis 2: (i -+ 1, j -+ 1) = p -+ (GBJ + 1); interior only
        if (r2) {
#pragma unroll
            for (int l = 0; l < LH; ++l) {
                const f32x4_t c = b0[l * PS + p], a = b0[l * PS + p - GBJ - 1], b = b0[l * PS + p + GBJ + 1];
#pragma unroll
                for (int q = 0; q < 4; ++q) o[l][q] = c[q] + 0.5f * (a[q] + b[q]);
            }
        }
        __syncthreads();
        if (r2) {
#pragma unroll
            for (int l = 0; l < LH; ++l) b0[l * PS + p] = o[l];
        }
#else
        // pass 0, axis 0: (i +- 1, j) = p +- GBJ
#pragma unroll
        for (int l = 0; l < LH; ++l) {
            f32x4_t o = zero;
            if (r0) {
                const f32x4_t c = b0[l * PS + p], a = b0[l * PS + p + GBJ], b = b0[l * PS + p - GBJ];
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = c[q] + 0.5f * (a[q] + b[q]);
            }
            b1[l * PS + p] = o;
        }
        __syncthreads();
        // pass 1, axis 1: (i, j +- 1) = p +- 1
#pragma unroll
        for (int l = 0; l < LH; ++l) {
            f32x4_t o = zero;
            if (r1) {
                const f32x4_t c = b1[l * PS + p], a = b1[l * PS + p - 1], b = b1[l * PS + p + 1];
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = c[q] + 0.5f * (a[q] + b[q]);
            }
            b0[l * PS + p] = o;
        }
        __syncthreads();
        // pass 2, axis 2: (i -+ 1, j -+ 1) = p -+ (GBJ + 1); interior only; through b1 to the row-wise store
        if (r2) {
#pragma unroll
            for (int l = 0; l < LH; ++l) {
                const f32x4_t c = b0[l * PS + p], a = b0[l * PS + p - GBJ - 1], b = b0[l * PS + p + GBJ + 1];
                f32x4_t o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = c[q] + 0.5f * (a[q] + b[q]);
                b1[l * PS + p] = o;
            }
        }
#endif
        __syncthreads();
#pragma unroll
        for (int i = 0; i < LH; ++i) {
            const int idx = i * P + p, pp = idx / LH, ll = idx - pp * LH;
            if (prow[i] > 0 && lbase + ll < LP) out[(unsigned)prow[i] * (unsigned)LP + lbase + ll] = (WSC_BLUR3_INPLACE ? b0 : b1)[ll * PS + pp];
        }
    }
}

// Slice of the ones-filter and norm = 1/sqrt(x + 1e-20)   (DenseKernel::initLattice)
// rec != null (bilateral lattice, dp1 = 6): the pixel's 13-dword record of the GF updates {6 row ids, 6 barycentric weights,
// norm} is written here as well -- the values are in registers, the block's 256 records go through LDS so that the stores are
// coalesced -- instead of by a second kernel that reads the three arrays back (pack_pixels_b_kernel: ~90-130 us per build).
__global__ __launch_bounds__(256) void slice_norm_kernel(const int32_t *__restrict__ offset, const float *__restrict__ bary,
                                                          int dp1, float alpha, const float *__restrict__ val, long long npix,
                                                          float *__restrict__ norm, uint32_t *__restrict__ rec) {
    __shared__ uint32_t lrec[256 * 13];
    for (long long p0 = (long long)blockIdx.x * 256; p0 < npix; p0 += (long long)gridDim.x * 256) {
        const long long p = p0 + threadIdx.x;
        if (p < npix) {
            float acc = 0.f;
            for (int r = 0; r < dp1; ++r) {
                const int o = offset[p * dp1 + r];
                const float w = bary[p * dp1 + r];
                acc += w * val[o] * alpha;
                if (rec) {
                    lrec[threadIdx.x * 13 + r] = (uint32_t)o;
                    lrec[threadIdx.x * 13 + 6 + r] = __float_as_uint(w);
                }
            }
            const float nv = (float)(1.0 / sqrt((double)acc + 1e-20));
            norm[p] = nv;
            if (rec) lrec[threadIdx.x * 13 + 12] = __float_as_uint(nv);
        }
        if (rec) {
            __syncthreads();
            const int n13 = (int)min(256ll, npix - p0) * 13;
            for (int i = threadIdx.x; i < n13; i += 256) rec[p0 * 13 + i] = lrec[i];
            __syncthreads();
        }
    }
}

// The same per PIXEL TILE (one block per tile, a thread per pixel), followed by the tile's entry scaling
// (tile_scale_entries_kernel: weight -> weight * norm[pixel]) from the norms the block has just computed -- the per-image
// lattices of a batch (round 4: one launch and one pass over the norms less per build; same arithmetic per pixel and entry).
__global__ __launch_bounds__(256) void slice_norm_tile_kernel(const int32_t *__restrict__ offset, const float *__restrict__ bary,
                                                               int dp1, float alpha, const float *__restrict__ val, TileGeom tg,
                                                               float *__restrict__ norm, uint32_t *__restrict__ rec,
                                                               float *__restrict__ tent_w, const uint8_t *__restrict__ tent_p) {
    __shared__ uint32_t lrec[256 * 13];
    __shared__ float lnorm[256];
    const int tile = xcd_block_id();
    const int b = tile / tg.tpi, j = tile - b * tg.tpi;
    const TileBox tb = tile_box(tg, j);
    const int N = tg.H * tg.W;
    const int np = tb.cw * tb.ch;
    const unsigned cw_magic = tile_div_magic(tb.cw);
    const int t = threadIdx.x;
    const int ty = (int)(((unsigned)t * cw_magic) >> 16), tx = t - ty * tb.cw;
    const long long p = (long long)b * N + (long long)(tb.y0 + ty) * tg.W + tb.x0 + tx;
    if (t < np) {
        // (round 6: the d+1 row ids and weights first, then the d+1 row values -- two memory round trips instead of a chain of
        // 2 (d+1) dependent ones per pixel; same products, added in the same order)
        constexpr int DMAX = 6;
        int o[DMAX];
        float w[DMAX], v[DMAX];
#pragma unroll
        for (int r = 0; r < DMAX; ++r) {
            o[r] = r < dp1 ? offset[p * dp1 + r] : 0;
            w[r] = r < dp1 ? bary[p * dp1 + r] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < DMAX; ++r) v[r] = r < dp1 ? val[o[r]] : 0.f;
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < DMAX; ++r)
            if (r < dp1) {
                acc += w[r] * v[r] * alpha;
                if (rec) {
                    lrec[t * 13 + r] = (uint32_t)o[r];
                    lrec[t * 13 + 6 + r] = __float_as_uint(w[r]);
                }
            }
        const float nv = (float)(1.0 / sqrt((double)acc + 1e-20));
        norm[p] = nv;
        lnorm[t] = nv;
        if (rec) lrec[t * 13 + 12] = __float_as_uint(nv);
    }
    __syncthreads();
    if (rec) { // a tile row's records are contiguous in the pixel-major record array: cw * 13 dwords per row
        const int rowlen = tb.cw * 13;
        for (int i = t; i < np * 13; i += 256) {
            const int ry = i / rowlen, k = i - ry * rowlen;
            rec[((long long)b * N + (long long)(tb.y0 + ry) * tg.W + tb.x0) * 13 + k] = lrec[i];
        }
    }
    const int ne = np * dp1;
    const long long ebase = ((long long)b * N + tb.ebase) * dp1;
    {   // entry weights x norm[pixel]: all of a thread's entries requested before the first is used
        constexpr int EN = (TILE_PIX * 6 + 255) / 256;
        float ew[EN];
        int ep[EN];
#pragma unroll
        for (int k = 0; k < EN; ++k) {
            const int i = t + k * 256;
            ew[k] = i < ne ? tent_w[ebase + i] : 0.f;
            ep[k] = i < ne ? (int)tent_p[ebase + i] : 0;
        }
#pragma unroll
        for (int k = 0; k < EN; ++k) {
            const int i = t + k * 256;
            if (i < ne) tent_w[ebase + i] = ew[k] * lnorm[ep[k]];
        }
    }
}

// everything the update kernel needs to know about a pixel in 80 contiguous bytes: one thread per 16-byte piece
// (coalesced stores; blockIdx.y = image, so no per-pixel 64-bit modulo for the shared Gaussian lattice's pixel)
__global__ void pack_pixels_kernel(const int32_t *__restrict__ off_g, const float *__restrict__ bary_g,
                                   const float *__restrict__ norm_g, const int32_t *__restrict__ off_b,
                                   const float *__restrict__ bary_b, const float *__restrict__ norm_b, int N,
                                   int g_shared, uint4 *__restrict__ rec) {
    const int b = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N * 5; i += gridDim.x * blockDim.x) {
        const int n = i / 5, piece = i - n * 5;
        const long long p = (long long)b * N + n;
        const long long pg = g_shared ? (long long)n : p; // the Gaussian lattice arrays describe one image
        uint32_t w[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = piece * 4 + c; // word of the record: 3 + 6 row ids, 3 + 6 barycentric weights, 2 norms
            uint32_t v;
            if (k < 3) v = (uint32_t)off_g[pg * 3 + k];
            else if (k < 9) v = (uint32_t)off_b[p * 6 + (k - 3)];
            else if (k < 12) v = __float_as_uint(bary_g[pg * 3 + (k - 9)]);
            else if (k < 18) v = __float_as_uint(bary_b[p * 6 + (k - 12)]);
            else if (k == 18) v = __float_as_uint(norm_g[pg]);
            else v = __float_as_uint(norm_b[p]);
            w[c] = v;
        }
        rec[p * 5 + piece] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// the bilateral part of the record alone: 13 dwords per pixel (one thread per dword: coalesced)
__global__ void pack_pixels_b_kernel(const int32_t *__restrict__ off_b, const float *__restrict__ bary_b,
                                     const float *__restrict__ norm_b, long long npix, uint32_t *__restrict__ rec) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix * 13; i += (long long)gridDim.x * blockDim.x) {
        const long long p = i / 13;
        const int k = (int)(i - p * 13);
        rec[i] = k < 6 ? (uint32_t)off_b[p * 6 + k] : (k < 12 ? __float_as_uint(bary_b[p * 6 + (k - 6)]) : __float_as_uint(norm_b[p]));
    }
}

constexpr int GM_THREADS = 512; // block size of gauss_msg_kernel and of the update kernel's FG variants
struct SplatTab { // splat tables of one lattice as the update kernel sees them
    const int32_t *tslot_start;
    const int2 *slot_desc;
    const float *tent_w;
    const uint8_t *tent_p;
    float *part;        // [slots][Mp] partial rows out
    int n_slots;        // per replica
    int shared;         // 1: tables describe one image (tile index j), replica k writes part + k*n_slots*Mp
    int dp1;
};

struct UpdateArgs {
    const uint4 *pix_rec; // [pixel][5]
    const uint32_t *pix_rec_b; // [pixel][13] bilateral part alone (GF variants)
    const float *val_g, *val_b;
    const float *u; // [pixel][Mp]
    float *q;       // [pixel][Mp] or null (only the last iteration's Q leaves the chip)
    int32_t *argmax; // [pixel] or null: labels-only call, the last iteration writes the arg-max instead of Q
    float alpha_g, alpha_b, compat_g, compat_b;
    int M, LP;
    int B;
    TileGeom tg;
    unsigned g_pix, g_rows; // shared Gaussian lattice: pixels / rows per replica (g_rows = 0: not shared)
    SplatTab sg, sb;
    // FG variants (the Gaussian message formed inside this kernel, E never in HBM): the tile vertex sets of gauss_fuse_tables
    // and the Gaussian slot partials of the PREVIOUS splat (a different array from sg.part, which this launch writes: a
    // neighbouring tile's block may already be splatting while this one still sums its closed vertex set)
    const int4 *gt_cnt;
    const int2 *gt_rows;
    const uint4 *gt_nbr;
    const uint4 *gt_pix;
    const float *part_g_in;
    int gt_stride;
    unsigned long long *tl; // A/B builds: per block 8 shader-clock stamps at the phase boundaries (null: off)
};
#ifdef WSC_AB_KNOBS
#define WSC_TL(a, i)                                                                                    \
    do {                                                                                                \
        if ((a).tl != nullptr && threadIdx.x == 0) (a).tl[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define WSC_TL(a, i) \
    do {             \
    } while (0)
#endif
constexpr int SPLAT_TAB_BYTES = (int)(sizeof(uint2) * TILE_PIX * 6 + sizeof(int2) * 256); // entries + slot descriptors

// load through a uniform base + 32-bit byte offset: the compiler can use the SGPR-base addressing form and the
// per-lane address arithmetic stays 32-bit (the 64-bit pointer adds were a fifth of the update loop's VALU work)
template <class T>
__device__ __forceinline__ T ld_off(const void *base, unsigned byte_off) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
constexpr int GATHER_SB = 256; // slot descriptors staged per batch
constexpr int GATHER_ENT = TILE_PIX * 6;

// partial[slot] = sum over the slot's entries of w * stage[pixel] in the entries' order, one slot per lane group per
// trip.  The tile's entries (contiguous in memory) and slot descriptors are copied into LDS first: the per-slot chains
// (descriptor -> entries -> Q rows) then run on LDS latency, not on three dependent trips to L2 / HBM per slot.
#ifndef WSC_SPLAT_ABL
#define WSC_SPLAT_ABL 0 // timing-only ablations of the splat phase (wrong results): 1 no partial-row stores, 2 no gather, 4 no entry copy
#endif
// The splat tables of one (tile, lattice) held in registers between their request and their use: the tile's entries
// (TILE_PIX * (d+1) of them: EN per thread), the thread's slot descriptor of the first batch and the tile's slot range.
// Requested early (the Gaussian lattice's before the trips, the bilateral one's before the Gaussian gather), they arrive
// under other work; the loop form (load -> LDS per entry, one after the other) had every block wait ~6 dependent HBM
// round trips per update.
template <int EN>
struct SplatRegs {
    float w[EN];
    unsigned px[EN];
    int2 desc;
    int s_beg, s_end;
};
template <int EN>
__device__ __forceinline__ void splat_fetch(const SplatTab &T, int tile, long long ebase_pix, int np, SplatRegs<EN> &r) {
    r.s_beg = T.tslot_start[tile];
    r.s_end = T.tslot_start[tile + 1];
    const int ne = (WSC_SPLAT_ABL & 4) ? 0 : np * T.dp1;
    const float *sw = T.tent_w + ebase_pix * T.dp1;
    const uint8_t *sp = T.tent_p + ebase_pix * T.dp1;
#pragma unroll
    for (int i = 0; i < EN; ++i) {
        const int e = (int)threadIdx.x + i * (int)blockDim.x;
        r.w[i] = 0.f;
        r.px[i] = 0u;
        if (e < ne) {
            r.w[i] = sw[e];
            r.px[i] = sp[e];
        }
    }
    r.desc = make_int2(0, 0);
    if ((int)threadIdx.x < r.s_end - r.s_beg && threadIdx.x < GATHER_SB) r.desc = T.slot_desc[r.s_beg + threadIdx.x];
}
template <int EN>
__device__ __forceinline__ void splat_commit(const SplatTab &T, int np, const SplatRegs<EN> &r, uint2 *lent, int2 *ldesc, int LP) {
    const int ne = (WSC_SPLAT_ABL & 4) ? 0 : np * T.dp1;
#pragma unroll
    for (int i = 0; i < EN; ++i) {
        const int e = (int)threadIdx.x + i * (int)blockDim.x;
        // {float4 index of the pixel's row in `stage`, weight bits}
        if (e < ne) lent[e] = make_uint2(r.px[i] * (unsigned)LP, __float_as_uint(r.w[i]));
    }
    if (threadIdx.x < GATHER_SB) ldesc[threadIdx.x] = r.desc;
}
// gather of the committed tables (first batch of slot descriptors already in LDS; a barrier separates commit and gather)
__device__ __forceinline__ void tile_gather(const SplatTab &T, int s_beg, int s_end, int rep_k,
                                            const f32x4_t *stage, uint2 *lent, int2 *ldesc, int LP, int l, int g, int gpw,
                                            bool act) {
    f32x4_t *part = reinterpret_cast<f32x4_t *>(T.part) + (T.shared ? (size_t)rep_k * T.n_slots * LP : (size_t)0);
    const int nw = (int)(blockDim.x >> 6), wv = (int)(threadIdx.x >> 6);
    for (int sb0 = s_beg; sb0 < s_end; sb0 += GATHER_SB) {
        const int nsb = min(GATHER_SB, s_end - sb0);
        if (sb0 > s_beg) { // further batches (tiles with more than GATHER_SB slots: noise images)
            if ((int)threadIdx.x < nsb) ldesc[threadIdx.x] = T.slot_desc[sb0 + threadIdx.x];
            __syncthreads();
        }
        for (int s0 = wv * gpw; s0 < nsb; s0 += nw * gpw) {
            const int s = s0 + g;
            const bool ok = act && s < nsb;
            const int2 d = ok ? ldesc[s] : make_int2(0, 0);
            const int i0 = d.x & 0xffff, n = (WSC_SPLAT_ABL & 2) ? 0 : d.x >> 16;
            // explicit packed FMAs (this file is compiled with -ffp-contract=off for the simplex search): the gather
            // is VALU/LDS-issue bound, 8 v_pk_fma_f32 per 4 entries instead of 16 mul + 16 add
            f32x2_t a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
            // whole groups of four entries without any per-entry predicate (no index clamp, no weight mask: 3 VALU per
            // entry next to the two packed FMAs instead of 5), then at most one masked group for the remainder; the
            // entries are summed in the same order as before
            const uint2 *le = lent + i0;
            int i = 0;
            for (; i + 4 <= n; i += 4) {
                uint2 en[4];
                f32x4_t in[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) en[jj] = le[i + jj];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) in[jj] = stage[en[jj].x + l];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float w = __uint_as_float(en[jj].y); // w * norm[pixel]
                    const f32x2_t w2 = {w, w}, lo = {in[jj][0], in[jj][1]}, hi = {in[jj][2], in[jj][3]};
                    a01 = __builtin_elementwise_fma(w2, lo, a01);
                    a23 = __builtin_elementwise_fma(w2, hi, a23);
                }
            }
            if (i < n) {
                uint2 en[3];
                f32x4_t in[3];
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) en[jj] = le[min(i + jj, n - 1)];
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) in[jj] = stage[en[jj].x + l];
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    const float w = i + jj < n ? __uint_as_float(en[jj].y) : 0.f;
                    const f32x2_t w2 = {w, w}, lo = {in[jj][0], in[jj][1]}, hi = {in[jj][2], in[jj][3]};
                    a01 = __builtin_elementwise_fma(w2, lo, a01);
                    a23 = __builtin_elementwise_fma(w2, hi, a23);
                }
            }
            const f32x4_t acc = {a01[0], a01[1], a23[0], a23[1]};
            if (ok && !(WSC_SPLAT_ABL & 1)) part[(unsigned)d.y * (unsigned)LP + l] = acc;
        }
        __syncthreads();
    }
}

// One mean-field update of a pixel tile, and the splat of the result into both lattices:
//   E = -U - (-wG * normG * sliceG) - (-wB * normB * sliceB);  Q = expAndNormalize(E)       (DenseCRF::inference loop body)
//   partial rows of (norm * Q) for the tile's Gaussian and bilateral slots                    (Permutohedral::compute, splat)
// SLICE = false: the messages are zero (Q = softmax(-U): the state before the first iteration).
// SPLAT = false: last iteration, Q goes to memory instead.
// LP lanes per pixel (4 classes each, 16-byte gathers), floor(64/LP) pixels per wave; the max / sum over a
// pixel's classes are reduced inside the lane, then across the pixel's LP lanes by shuffle-down with a segment
// bound and a broadcast from the segment's first lane.  Between iterations Q exists only as the tile's LDS copy.
// GF (with SLICE): the Gaussian lattice's message is not gathered here.  gauss_msg_kernel (below) has already summed the
// Gaussian slot partials of every pixel tile's closed vertex set, blurred them in LDS and left E = -U + (Gaussian message)
// in a pixel-major buffer of U's layout -- with the very FMAs, in the very order, this kernel would have used -- so a.u
// points at E, the energy starts from +E instead of -U, and only the six bilateral rows are gathered.  Bit-identical to
// the unfused path; the Gaussian lattice's value rows never exist, and the message kernel runs beside the bilateral
// lattice's combine + blur chain between two updates.
// DMA (GF variants): the tile's E rows and 52-byte records are streamed into LDS by LDS-DMA loads (global_load_lds: no
// registers, every byte of the tile in flight at once) before the trips start -- E straight into the Q-stage slots its lanes
// overwrite later, the records (padded to 56 bytes) into the region the splat tables use afterwards.  The trips then read
// both from LDS and only the six bilateral row gathers (L2) remain in a trip's dependent chain; without it a trip waits
// for its record from HBM before it can request its rows, and the kernel is bound by that latency at 4 waves per SIMD.
constexpr int REC_LDS_BYTES = 56; // 13 dwords + 1 pad: 8-byte aligned records, TILE_PIX of them fit the splat-table region
static_assert(TILE_PIX * REC_LDS_BYTES <= (int)(sizeof(uint2) * GATHER_ENT + sizeof(int2) * GATHER_SB), "record stage");
template <bool SLICE, bool SPLAT, bool GF = false, bool DMA = false, int FG = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) void update_splat_kernel(UpdateArgs a) {
    static_assert(!DMA || (SLICE && GF) || (!SLICE && SPLAT), "the LDS-DMA staging belongs to the GF updates and the first update");
    static_assert(FG == 0 || (SLICE && GF && DMA), "the in-kernel Gaussian message belongs to the DMA-staged GF updates");
    extern __shared__ f32x4_t stage[]; // [TILE_PIX][LP]
    const int LP = a.LP;
    const int gpw = 64 / LP;
    const int lane = threadIdx.x & 63;
    const int g = lane / LP;
    const int l = lane - g * LP;
    const bool act = g < gpw;
    const int seg0 = g * LP;
    // XCD-contiguous logical block id: the tiles of one image (which share lattice rows) on one L2
    const int nb = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, qq = nb >> 3, rr = nb & 7;
    const int lb = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int k = lb / a.tg.tpi, j = lb - k * a.tg.tpi; // image, tile of the image
    const TileBox tb = tile_box(a.tg, j);
    const int np = tb.cw * tb.ch;
    const int N = a.tg.H * a.tg.W;
    const int ppt = (int)(blockDim.x >> 6) * gpw; // pixels per trip
    // Addressing: a uniform (SGPR) base per array -- the tile's first pixel, the image's replica of the Gaussian rows --
    // plus a 32-bit per-lane byte offset formed with 24-bit multiplies (v_mad_u32_u24, full rate; a 32-bit v_mul_lo_u32
    // is a quarter-rate instruction and there were 15 of them per trip): the tile-local pixel offset is below N < 2^24
    // and a row id below 2^24 (both checked by the host).
    const unsigned cw_magic = tile_div_magic(tb.cw);
    const size_t pix0 = (size_t)k * (size_t)N + (size_t)tb.y0 * (size_t)a.tg.W + (size_t)tb.x0;
    const unsigned LP16 = (unsigned)LP * 16u, l16 = (unsigned)l * 16u;
    constexpr unsigned REC_BYTES = GF ? 52u : 80u; // GF: the 13-dword bilateral record (ids 0..5, bary 6..11, norm 12)
    constexpr int REC_ID = GF ? 0 : 3, REC_BARY = GF ? 6 : 12, REC_NORM = GF ? 12 : 19;
    const char *rec_b = (GF ? reinterpret_cast<const char *>(a.pix_rec_b) : reinterpret_cast<const char *>(a.pix_rec)) + pix0 * REC_BYTES;
    const char *u_b = reinterpret_cast<const char *>(a.u) + pix0 * LP16;
    char *q_b = reinterpret_cast<char *>(a.q) + pix0 * LP16;
    const char *vg_b = reinterpret_cast<const char *>(a.val_g) + (a.g_rows ? (size_t)k * a.g_rows * LP16 : (size_t)0);
    const char *vb_b = reinterpret_cast<const char *>(a.val_b);
    // pixel of the lane group in trip t0 as an offset from the tile's first pixel (clamped to the tile: idle lanes
    // re-read pixel 0)
    auto pixel_of = [&](int t0) -> unsigned {
        const int t = t0 + g;
        const unsigned tc = (act && t < np) ? (unsigned)t : 0u;
        const unsigned ty = (tc * cw_magic) >> 16, tx = tc - __umul24(ty, (unsigned)tb.cw);
        return __umul24(ty, (unsigned)a.tg.W) + tx;
    };
    // Software pipeline over the trips of a wave, two deep: at the top of trip t the lattice rows of trip t+1 are
    // requested (its record arrived during trip t-1) together with the record + unary of trip t+2; trip t's softmax runs
    // on an energy that is already complete; at the bottom the rows of trip t+1 are folded into ITS energy.  record ->
    // rows is a dependent chain and the kernel runs at 4 waves per SIMD: a wave now keeps a trip of row gathers in
    // flight across the whole update of the trip before it (same FMA order per pixel as before: bit-identical).
    uint4 rq[5];          // record of the trip whose rows are not requested yet
    f32x4_t unn;          // unary of the trip after the next one to be folded
    f32x2_t e01, e23;     // energy of the current trip
    const float cag = a.compat_g * a.alpha_g, cab = a.compat_b * a.alpha_b;
    const char *lrec = reinterpret_cast<const char *>(stage + TILE_PIX * LP); // DMA: the tile's records in LDS
    auto tile_idx = [&](int t0) -> unsigned { // the lane group's pixel of trip t0 inside the tile (clamped like pixel_of)
        const int t = t0 + g;
        return (act && t < np) ? (unsigned)t : 0u;
    };
    auto load_u = [&](int t0) -> f32x4_t {
        if (DMA) return stage[tile_idx(t0) * (unsigned)LP + (unsigned)l];
        return ld_off<f32x4_t>(u_b, __umul24(pixel_of(t0), LP16) + l16);
    };
    auto load_rec = [&](int t0, uint4(&r)[5]) {
        if (DMA) { // 6 x 8 bytes + 4 from the LDS copy
            const char *q = lrec + tile_idx(t0) * (unsigned)REC_LDS_BYTES;
            uint2 w[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i] = *reinterpret_cast<const uint2 *>(q + 8 * i);
            r[0] = make_uint4(w[0].x, w[0].y, w[1].x, w[1].y);
            r[1] = make_uint4(w[2].x, w[2].y, w[3].x, w[3].y);
            r[2] = make_uint4(w[4].x, w[4].y, w[5].x, w[5].y);
            r[3] = make_uint4(*reinterpret_cast<const uint32_t *>(q + 48), 0u, 0u, 0u);
            r[4] = make_uint4(0u, 0u, 0u, 0u);
            return;
        }
        const unsigned p = pixel_of(t0);
        if (GF) { // 3 x 16 bytes + 4 (dword-aligned 16-byte loads)
#pragma unroll
            for (int i = 0; i < 3; ++i) r[i] = ld_off<uint4>(rec_b, __umul24(p, REC_BYTES) + 16u * i);
            r[3] = make_uint4(ld_off<uint32_t>(rec_b, __umul24(p, REC_BYTES) + 48u), 0u, 0u, 0u);
            r[4] = make_uint4(0u, 0u, 0u, 0u);
        } else {
#pragma unroll
            for (int i = 0; i < 5; ++i) r[i] = ld_off<uint4>(rec_b, __umul24(p, REC_BYTES) + 16u * i);
        }
    };
    // the pixel's record: 5 x 16 bytes, identical for the LP lanes of the pixel (broadcast loads)
    auto issue_rows = [&](const uint4(&r)[5], f32x4_t(&g3)[3], f32x4_t(&b6)[6], float(&w9)[9]) {
        uint32_t rc[20];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            rc[4 * i] = r[i].x; rc[4 * i + 1] = r[i].y; rc[4 * i + 2] = r[i].z; rc[4 * i + 3] = r[i].w;
        }
        if (!GF) {
#pragma unroll
            for (int i = 0; i < 3; ++i) g3[i] = ld_off<f32x4_t>(vg_b, __umul24(rc[i], LP16) + l16);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) b6[i] = ld_off<f32x4_t>(vb_b, __umul24(rc[REC_ID + i], LP16) + l16);
        const float wb = cab * __uint_as_float(rc[REC_NORM]);
        if (!GF) { // (bary * norm) * (compat * alpha): the product the GF path reads precomputed from gt_pix
#pragma unroll
            for (int i = 0; i < 3; ++i) w9[i] = (__uint_as_float(rc[9 + i]) * __uint_as_float(rc[18])) * cag;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) w9[3 + i] = __uint_as_float(rc[REC_BARY + i]) * wb;
    };
    // E = -U + sum_r (compat * alpha * norm * bary_r) * row_r : nine packed FMAs per class pair (the weights are formed
    // once per pixel, the row sums run as v_pk_fma_f32)
    auto fold = [&](const f32x4_t &u, const f32x4_t(&g3)[3], const f32x4_t(&b6)[6], const float(&w9)[9], f32x2_t &o01, f32x2_t &o23) {
        if (SLICE && GF) { // `u` is E = -U + Gaussian message
            o01 = f32x2_t{u[0], u[1]};
            o23 = f32x2_t{u[2], u[3]};
        } else {
            o01 = f32x2_t{-u[0], -u[1]};
            o23 = f32x2_t{-u[2], -u[3]};
        }
        if (SLICE) {
            if (!GF) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const f32x2_t w2 = {w9[r], w9[r]}, lo = {g3[r][0], g3[r][1]}, hi = {g3[r][2], g3[r][3]};
                    o01 = __builtin_elementwise_fma(w2, lo, o01);
                    o23 = __builtin_elementwise_fma(w2, hi, o23);
                }
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const f32x2_t w2 = {w9[3 + r], w9[3 + r]}, lo = {b6[r][0], b6[r][1]}, hi = {b6[r][2], b6[r][3]};
                o01 = __builtin_elementwise_fma(w2, lo, o01);
                o23 = __builtin_elementwise_fma(w2, hi, o23);
            }
        }
    };
    const int t_first = (int)(threadIdx.x >> 6) * gpw;
    // splat tables of the Gaussian lattice: requested now, used after the trips (8 registers across the loop)
    constexpr int EN_G = (TILE_PIX * 3 + 255) / 256, EN_B = (TILE_PIX * 6 + 255) / 256; // blocks of >= 256 threads (update_threads)
    SplatRegs<EN_G> rg;
    WSC_TL(a, 0); // block start
    if (SPLAT && FG == 0) splat_fetch<EN_G>(a.sg, a.sg.shared ? j : lb, (a.sg.shared ? 0ll : (long long)k * N) + tb.ebase, np, rg);
    // FG: the descriptors of the tile's closed vertex set {first partial row, count} are requested before the LDS-DMA pieces
    // (loads return in order: behind them the first wait of the Gaussian phase would sit behind the whole tile's stream)
    int4 cnt = make_int4(0, 0, 0, 0);
    int2 ps[FG > 0 ? FG : 1];
    uint4 tn0 = make_uint4(0, 0, 0, 0);
    if (FG > 0) {
        cnt = a.gt_cnt[j];
        const unsigned lpm = (65536u + (unsigned)LP - 1u) / (unsigned)LP;
        const int2 *trow = a.gt_rows + (size_t)j * a.gt_stride;
#pragma unroll
        for (int it = 0; it < (FG > 0 ? FG : 1); ++it) {
            const int i = (int)threadIdx.x + it * GM_THREADS;
            ps[it] = make_int2(0, 0);
            if (i < a.gt_stride * LP) ps[it] = trow[((unsigned)i * lpm) >> 16];
        }
        if ((int)threadIdx.x < a.gt_stride) tn0 = a.gt_nbr[(size_t)j * a.gt_stride + threadIdx.x]; // (neighbour words: to LDS below)
    }
    if (DMA) {
        // E rows -> stage[t][l] (16-byte units u = t * LP + l), records -> lrec[t][14 dwords] (dword units d = t * 14 + k);
        // a wave instruction fills 64 consecutive units from per-lane source addresses (units past the tile re-read its
        // last one: the destination is lane-linear and has room for whole wave instructions)
        const int wvu = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwv = (int)(blockDim.x >> 6);
        const unsigned lp_magic = (65536u + (unsigned)LP - 1u) / (unsigned)LP; // u / LP for u < 8192, LP <= 8
        const int nE = np * LP;
        for (int u0 = wvu * 64; u0 < nE; u0 += nwv * 64) {
            const unsigned u = min((unsigned)(u0 + lane), (unsigned)(nE - 1));
            const unsigned t = (u * lp_magic) >> 16, ll = u - t * (unsigned)LP;
            const unsigned ty = (t * cw_magic) >> 16, tx = t - __umul24(ty, (unsigned)tb.cw);
            const char *src = u_b + (__umul24(__umul24(ty, (unsigned)a.tg.W) + tx, LP16) + ll * 16u);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(stage) + u0 * 16), 16, 0, 0);
        }
        const int nR = SLICE ? np * 14 : 0; // (the first update has no records: U rows only)
        for (int d0 = wvu * 64; d0 < nR; d0 += nwv * 64) {
            const unsigned d = min((unsigned)(d0 + lane), (unsigned)(nR - 1));
            const unsigned t = (d * 4682u) >> 16; // d / 14 for d < 4096
            const unsigned kk = min(d - t * 14u, 12u); // the pad dword re-reads the norm
            const unsigned ty = (t * cw_magic) >> 16, tx = t - __umul24(ty, (unsigned)tb.cw);
            // (staging straight from the lattice's row-id / weight / norm arrays instead of the packed record was measured:
            // 322 vs 311 us per launch -- three source streams per pixel -- for the ~90 us of pack_pixels_b_kernel per build)
            const char *src = rec_b + (__umul24(__umul24(ty, (unsigned)a.tg.W) + tx, REC_BYTES) + kk * 4u);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(const_cast<char *>(lrec) + d0 * 4), 4, 0, 0);
        }
        if (FG == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    WSC_TL(a, 1); // DMA issued (FG) / landed (two-launch form)
    if constexpr (FG > 0) {
        // ---- Gaussian message of the tile, on chip (round 6): the body of gauss_msg_kernel with E = -U + message left in the
        // Q stage instead of HBM.  blockDim.x == GM_THREADS.  The U rows and the records are already travelling (LDS-DMA above);
        // this phase sums the Gaussian slot partials of the tile's closed vertex set into LDS, blurs them in place and slices
        // them into the staged U rows -- the same loads, the same FMAs in the same order as the two-kernel form: identical bits.
        constexpr int NIT = FG, NPI = FG == 4 ? 3 : 4;
        f32x4_t *gl = reinterpret_cast<f32x4_t *>(reinterpret_cast<char *>(stage + TILE_PIX * LP) + SPLAT_TAB_BYTES);
        const int stride = a.gt_stride, tid = (int)threadIdx.x;
        unsigned *lnb = reinterpret_cast<unsigned *>(gl + (size_t)stride * (size_t)LP);
        const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
        const f32x4_t *partg = reinterpret_cast<const f32x4_t *>(a.part_g_in) + (a.sg.shared ? (size_t)k * a.sg.n_slots * LP : (size_t)0);
        const int nitems_all = stride * LP;
        const unsigned lp_magic = (65536u + (unsigned)LP - 1u) / (unsigned)LP;
        const uint4 *tn = a.gt_nbr + (size_t)j * stride;
        const unsigned gpix0 = (unsigned)tb.y0 * (unsigned)a.tg.W + (unsigned)tb.x0;
        {
            f32x4_t val[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = tid + it * GM_THREADS;
                const unsigned v = ((unsigned)i * lp_magic) >> 16, ll = (unsigned)i - v * (unsigned)LP;
                val[it] = zero;
                if (ps[it].y > 0) val[it] = partg[(unsigned)ps[it].x * (unsigned)LP + ll];
            }
            WSC_TL(a, 2); // descriptors arrived, first partials requested
            if (tid < stride) reinterpret_cast<uint4 *>(lnb)[tid] = tn0;
            for (int v = tid + GM_THREADS; v < stride; v += GM_THREADS) reinterpret_cast<uint4 *>(lnb)[v] = tn[v];
            // further partials of the rows that have them (tile-border vertices: up to four tiles touch one): the second, third
            // and fourth partial of every item are requested TOGETHER (round 6: one memory round trip instead of one per
            // partial -- the phase timeline showed 2.4 us here) and added in index order as before: first + second + ...
            int maxc = 0;
#pragma unroll
            for (int it = 0; it < NIT; ++it) maxc = max(maxc, ps[it].y);
            constexpr int UN = NIT <= 4 ? 3 : 2;
            if (maxc > 1) {
                f32x4_t more[UN][NIT];
#pragma unroll
                for (int c = 1; c <= UN; ++c)
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const int i = tid + it * GM_THREADS;
                        const unsigned v = ((unsigned)i * lp_magic) >> 16, ll = (unsigned)i - v * (unsigned)LP;
                        more[c - 1][it] = zero;
                        if (c < ps[it].y) more[c - 1][it] = partg[(unsigned)(ps[it].x + c) * (unsigned)LP + ll];
                    }
#pragma unroll
                for (int c = 1; c <= UN; ++c)
#pragma unroll
                    for (int it = 0; it < NIT; ++it)
                        if (c < ps[it].y) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) val[it][q] += more[c - 1][it][q];
                        }
            }
            for (int c = UN + 1; c < maxc; ++c) {
                f32x4_t more[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int i = tid + it * GM_THREADS;
                    const unsigned v = ((unsigned)i * lp_magic) >> 16, ll = (unsigned)i - v * (unsigned)LP;
                    more[it] = zero;
                    if (c < ps[it].y) more[it] = partg[(unsigned)(ps[it].x + c) * (unsigned)LP + ll];
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it)
                    if (c < ps[it].y) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) val[it][q] += more[it][q];
                    }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = tid + it * GM_THREADS;
                if (i < nitems_all) gl[i] = val[it];
            }
        }
        // every load of this wave so far has been consumed -- the partial rows were requested after the LDS-DMA pieces and
        // loads return in order, so the wave's pieces (U rows, records) have landed too; the barrier below publishes them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WSC_TL(a, 3); // all partial rows summed into LDS
        // the pixels' Gaussian records (L2: one table for all images) travel under the blur passes
        uint4 gpx[NPI];
#pragma unroll
        for (int it = 0; it < NPI; ++it) {
            const int i = tid + it * GM_THREADS;
            gpx[it] = make_uint4(0, 0, 0, 0);
            if (i < np * LP) {
                const unsigned t = ((unsigned)i * lp_magic) >> 16;
                const unsigned ty = (t * cw_magic) >> 16, tx = t - ty * (unsigned)tb.cw;
                gpx[it] = a.gt_pix[gpix0 + ty * (unsigned)a.tg.W + tx];
            }
        }
        __syncthreads();
        // (Round 6, measured and removed: the six bilateral rows of the wave's first trip requested here, under the blur passes:
        // trips 3.54 -> 2.89 us, blur + slice 4.30 -> 4.75 us, Gaussian splat 2.59 -> 2.84 us -- the block's lifetime did not
        // move (18.4 -> 18.6 us).  What one phase gives up another takes: the kernel is bound by throughput, not by a latency.)
#pragma unroll
        for (int axis = 0; axis < 3; ++axis) {
            // (Round 6, measured and removed: an item's own value carried through the passes in registers instead of re-read
            // from LDS -- a quarter of the blur's LDS reads less, 118 registers, no spills: 463.6 -> 505.7 us per launch, the
            // blur + slice phase 4.33 -> 5.00 us.)
            const int nitems = (axis == 0 ? cnt.z : (axis == 1 ? cnt.y : cnt.x)) * LP;
            f32x4_t o[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = tid + it * GM_THREADS;
                o[it] = zero;
                if (i < nitems) {
                    const unsigned v = ((unsigned)i * lp_magic) >> 16, ll = (unsigned)i - v * (unsigned)LP;
                    const unsigned w = lnb[v * 4 + axis];
                    const f32x4_t c = gl[i], x1 = gl[(w & 0xffffu) * (unsigned)LP + ll], x2 = gl[(w >> 16) * (unsigned)LP + ll];
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[it][q] = c[q] + 0.5f * (x1[q] + x2[q]);
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = tid + it * GM_THREADS;
                if (i < nitems) gl[i] = o[it];
            }
            __syncthreads();
        }
        // slice into the staged U rows: E = -U, then the three FMAs (weights (bary * norm) * (compat * alpha)), in place
#pragma unroll
        for (int it = 0; it < NPI; ++it) {
            const int i = tid + it * GM_THREADS;
            if (i < np * LP) {
                const unsigned t = ((unsigned)i * lp_magic) >> 16, ll = (unsigned)i - t * (unsigned)LP;
                const uint4 gp = gpx[it];
                const f32x4_t uu = stage[i];
                f32x2_t o01 = {-uu[0], -uu[1]}, o23 = {-uu[2], -uu[3]};
                const float wr[3] = {__uint_as_float(gp.y) * cag, __uint_as_float(gp.z) * cag, __uint_as_float(gp.w) * cag};
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const f32x4_t row = gl[((gp.x >> (10 * r)) & 1023u) * (unsigned)LP + ll];
                    const f32x2_t w2 = {wr[r], wr[r]}, lo = {row[0], row[1]}, hi = {row[2], row[3]};
                    o01 = __builtin_elementwise_fma(w2, lo, o01);
                    o23 = __builtin_elementwise_fma(w2, hi, o23);
                }
                stage[i] = f32x4_t{o01[0], o01[1], o23[0], o23[1]};
            }
        }
        if (SPLAT) splat_fetch<EN_G>(a.sg, a.sg.shared ? j : lb, (a.sg.shared ? 0ll : (long long)k * N) + tb.ebase, np, rg);
        __syncthreads();
    }
    WSC_TL(a, 4); // E in the Q stage: the trips start
    // softmax of the current trip's energy (e01, e23) -> Q into the tile's stage (and to memory / the arg-max in the last update)
    auto emit = [&](int t0) {
        const int t = t0 + g;
        const bool ok = act && t < np;
        const unsigned p = pixel_of(t0);
        float e[4] = {e01[0], e01[1], e23[0], e23[1]};
        float mx = -3.0e38f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            e[kk] = (ok && 4 * l + kk < a.M) ? e[kk] : -3.0e38f;
            mx = fmaxf(mx, e[kk]);
        }
        for (int o = 4; o > 0; o >>= 1) {
            const float other = __shfl_down(mx, o, 64);
            if (l + o < LP) mx = fmaxf(mx, other);
        }
        mx = __shfl(mx, seg0, 64);
        float ex[4], sum = 0.f;
        const float mxl = mx * 1.44269504088896341f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            // exp(e - mx) as exp2(e * log2(e) - mx * log2(e)); masked classes give exp2(-huge) = 0
            ex[kk] = __builtin_amdgcn_exp2f(__builtin_fmaf(e[kk], 1.44269504088896341f, -mxl));
            sum += ex[kk];
        }
        for (int o = 4; o > 0; o >>= 1) {
            const float other = __shfl_down(sum, o, 64);
            if (l + o < LP) sum += other;
        }
        sum = __shfl(sum, seg0, 64);
        const float rs = __builtin_amdgcn_rcpf(sum);
        const f32x4_t o4 = {ex[0] * rs, ex[1] * rs, ex[2] * rs, ex[3] * rs};
        if (ok) {
            if (SPLAT) stage[t * LP + l] = o4;
            if (a.q) *reinterpret_cast<f32x4_t *>(q_b + (__umul24(p, LP16) + l16)) = o4;
        }
        if (!SPLAT && a.argmax != nullptr) {
            // np.argmax(Q, axis=0) on the very values a full call would have written: first maximum in class order
            float bv = o4[0];
            int bi = 4 * l;
#pragma unroll
            for (int kk = 1; kk < 4; ++kk)
                if (4 * l + kk < a.M && o4[kk] > bv) {
                    bv = o4[kk];
                    bi = 4 * l + kk;
                }
            for (int o = 4; o > 0; o >>= 1) {
                const float ov = __shfl_down(bv, o, 64);
                const int oi = __shfl_down(bi, o, 64);
                if (l + o < LP && ov > bv) { // the other lane holds HIGHER classes: it wins only when strictly larger
                    bv = ov;
                    bi = oi;
                }
            }
            if (ok && l == 0) a.argmax[pix0 + p] = bi;
        }
    };
    // (Round 6, measured and removed: the six bilateral rows of a trip requested TWO trips before they are folded -- two sets of
    // row registers, the loop unrolled by two, records and E rows re-read from LDS at fold time: 128 registers, no spills,
    // identical bits.  The trips of a block got 0.55 us shorter (8.41 -> 7.86 us, profiles/upd_timeline.py) and the wait at the
    // next barrier as much longer: 302.0 -> 301.1 us per launch.  The trips are not bound by the L2 gather latency.)
    {
        {
            const f32x4_t u0 = load_u(t_first);
            f32x4_t g3[3], b6[6];
            float w9[9];
            if (SLICE) {
                load_rec(t_first, rq);
                issue_rows(rq, g3, b6, w9);
            }
            unn = u0;
            if (t_first + ppt < np) {
                if (SLICE) load_rec(t_first + ppt, rq);
                unn = load_u(t_first + ppt);
            }
            fold(u0, g3, b6, w9, e01, e23);
        }
        for (int t0 = t_first; t0 < np; t0 += ppt) {
            // next trip's rows (its record was requested one trip ago), then the record + unary of the trip after it
            const bool has_next = t0 + ppt < np;
            f32x4_t vgn[3], vbn[6];
            float wrn[9];
            const f32x4_t un1 = unn;
            if (SLICE && has_next) issue_rows(rq, vgn, vbn, wrn);
            if (t0 + 2 * ppt < np) {
                if (SLICE) load_rec(t0 + 2 * ppt, rq);
                unn = load_u(t0 + 2 * ppt);
            }
            emit(t0);
            if (has_next) fold(un1, vgn, vbn, wrn, e01, e23); // the next trip's energy: its rows have had this trip to arrive
        }
    }
    WSC_TL(a, 5); // this wave's trips done
    if (SPLAT) {
        uint2 *lent = reinterpret_cast<uint2 *>(stage + TILE_PIX * LP);
        int2 *ldesc = reinterpret_cast<int2 *>(lent + GATHER_ENT);
        if (DMA) __syncthreads(); // every wave is done with the records: their region becomes the splat tables
        splat_commit<EN_G>(a.sg, np, rg, lent, ldesc, LP);
        // the bilateral lattice's tables travel while the Gaussian lattice's slots are gathered
        SplatRegs<EN_B> rb;
        splat_fetch<EN_B>(a.sb, a.sb.shared ? j : lb, (a.sb.shared ? 0ll : (long long)k * N) + tb.ebase, np, rb);
        __syncthreads(); // Q stage + Gaussian tables complete
        tile_gather(a.sg, rg.s_beg, rg.s_end, k, stage, lent, ldesc, LP, l, g, gpw, act); // (ends with a barrier)
        WSC_TL(a, 6); // Gaussian slots written
        splat_commit<EN_B>(a.sb, np, rb, lent, ldesc, LP);
        __syncthreads();
        tile_gather(a.sb, rb.s_beg, rb.s_end, k, stage, lent, ldesc, LP, l, g, gpw, act);
    }
    WSC_TL(a, 7); // block end
}

// ---- Gaussian message of a pixel tile, on chip ------------------------------------------------------------------------
// E[pixel] = -U[pixel] + (compat * alpha * norm) * slice(blur(splat values)) for the pixels of one tile, from the Gaussian
// lattice's slot partials: the block sums the partial rows of the tile's closed vertex set (gauss_fuse_tables) into LDS,
// runs the three blur passes there in place (taps -> registers, barrier, write back: the arithmetic of blur4_kernel /
// blur3_tile_kernel, value for value) and folds the three vertex rows of every pixel into -U with the FMAs, in the order,
// update_splat_kernel uses -- which then starts from E (its GF variants) and gathers the bilateral rows only.  One read of
// the partial rows, no value rows in HBM, and the launch runs beside the bilateral lattice's combine + blur chain.
#ifndef WSC_GF_ABL
#define WSC_GF_ABL 0 // timing-only ablations (wrong results): 1 no blur passes, 2 no partial-row loads, 4 no slice
#endif
struct GaussMsgArgs {
    const int4 *gt_cnt;
    const int2 *gt_rows;
    const uint4 *gt_nbr;
    const uint4 *gt_pix;
    const float *part; // [rep][n_slots][Mp] slot partials of the last splat
    const float *u;    // [pixel][Mp]
    float *e;          // [pixel][Mp] out
    int gt_stride, n_slots, shared, LP, B;
    float cag;         // compat * alpha
    TileGeom tg;
};
// NIT: (vertex, float4) items per thread in a blur pass; NPI: (pixel, float4) items per thread in the slice
template <int NIT, int NPI>
__global__ __launch_bounds__(GM_THREADS) __attribute__((amdgpu_waves_per_eu(4))) void gauss_msg_kernel(GaussMsgArgs a) {
    extern __shared__ f32x4_t gl[]; // [stride][LP] splat values, blurred in place; then [stride][4] neighbour words
    const int LP = a.LP, stride = a.gt_stride, tid = (int)threadIdx.x;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    const int nb = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, qq = nb >> 3, rr = nb & 7;
    const int lb = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int k = lb / a.tg.tpi, j = lb - k * a.tg.tpi; // image, tile of the image
    const TileBox tb = tile_box(a.tg, j);
    const int np = tb.cw * tb.ch;
    const int N = a.tg.H * a.tg.W;
    unsigned *lnb = reinterpret_cast<unsigned *>(gl + (size_t)stride * (size_t)LP);
    const f32x4_t *partg = reinterpret_cast<const f32x4_t *>(a.part) + (a.shared ? (size_t)k * a.n_slots * LP : (size_t)0);
    // items = (vertex, float4) over the PADDED set: padding vertices have no partial rows (value 0) and the last one is the
    // zero row absent neighbours point at, so nothing waits for a per-tile count
    const int nitems = stride * LP;
    const unsigned lp_magic = (65536u + (unsigned)LP - 1u) / (unsigned)LP; // i / LP for i < 8192, LP <= 8
    const unsigned cw_magic = tile_div_magic(tb.cw);
    const int2 *trow = a.gt_rows + (size_t)j * stride;
    const uint4 *tn = a.gt_nbr + (size_t)j * stride;
    const unsigned gpix0 = (unsigned)tb.y0 * (unsigned)a.tg.W + (unsigned)tb.x0;
    const size_t pix0 = (size_t)k * (size_t)N + gpix0;
    const f32x4_t *u4 = reinterpret_cast<const f32x4_t *>(a.u) + pix0 * LP;
    f32x4_t *e4 = reinterpret_cast<f32x4_t *>(a.e) + pix0 * LP;
    // every load that depends on nothing first: the set sizes, row descriptors, the pixels' records and unaries, the
    // neighbour words
    const int4 cnt = a.gt_cnt[j];
    int2 ps[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + it * GM_THREADS;
        ps[it] = make_int2(0, 0);
        if (i < nitems) ps[it] = trow[((unsigned)i * lp_magic) >> 16];
    }
    // the pixels' records and unaries: requested up front when the registers allow it (the small variant), else after the blur
    constexpr bool EARLY = false; // (up front they cost 30 registers: two blocks per CU instead of four, 186 -> 241 us)
    uint4 gpx[NPI];
    f32x4_t uu[NPI];
    unsigned poff[NPI];
    auto load_pixels = [&]() {
    #pragma unroll
        for (int it = 0; it < NPI; ++it) {
            const int i = tid + it * GM_THREADS;
            gpx[it] = make_uint4(0, 0, 0, 0);
            uu[it] = zero;
            poff[it] = 0;
            if (i < np * LP) {
                const unsigned t = ((unsigned)i * lp_magic) >> 16, ll = (unsigned)i - t * (unsigned)LP;
                const unsigned ty = (t * cw_magic) >> 16, tx = t - ty * (unsigned)tb.cw;
                const unsigned po = ty * (unsigned)a.tg.W + tx;
                poff[it] = po * (unsigned)LP + ll;
                gpx[it] = a.gt_pix[gpix0 + po];
                uu[it] = u4[poff[it]];
            }
        }
    };
    if (EARLY) load_pixels();
#pragma unroll 2
    for (int v = tid; v < stride; v += GM_THREADS) reinterpret_cast<uint4 *>(lnb)[v] = tn[v];
    {
        // all first partials (independent loads: one round trip for the whole set), then the rows with several partials
        // (tile-border vertices, ~20 %), summed in index order as combine_slots4 does everywhere
        f32x4_t val[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * GM_THREADS;
            const unsigned v = ((unsigned)i * lp_magic) >> 16, ll = (unsigned)i - v * (unsigned)LP;
            val[it] = zero;
            if (!(WSC_GF_ABL & 2) && ps[it].y > 0) val[it] = partg[(unsigned)ps[it].x * (unsigned)LP + ll];
        }
        // further partials of the rows that have them, all items of the thread per step (static register indices;
        // every row still adds its partials in index order: first + second + ...)
        int maxc = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) maxc = max(maxc, ps[it].y);
        if (WSC_GF_ABL & 2) maxc = 0;
        // (round 6: requesting the second ... fourth partials of every item together -- what the FG variants of
        // update_splat_kernel do -- costs this kernel 32 registers and a block per CU: 186 -> 220 us; one partial per step stays)
        for (int c = 1; c < maxc; ++c) {
            f32x4_t more[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = tid + it * GM_THREADS;
                const unsigned v = ((unsigned)i * lp_magic) >> 16, ll = (unsigned)i - v * (unsigned)LP;
                more[it] = zero;
                if (c < ps[it].y) more[it] = partg[(unsigned)(ps[it].x + c) * (unsigned)LP + ll];
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if (c < ps[it].y) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) val[it][q] += more[it][q];
                }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * GM_THREADS;
            if (i < nitems) gl[i] = val[it];
        }
    }
    __syncthreads();
#pragma unroll
    for (int axis = 0; axis < ((WSC_GF_ABL & 1) ? 0 : 3); ++axis) {
        // pass `axis` only has to be right on the vertices the later passes and the slice read: the nested prefixes of the
        // set (gauss_fuse_tables); the rest keeps its old value, which nothing reads any more
        const int nitems = (axis == 0 ? cnt.z : (axis == 1 ? cnt.y : cnt.x)) * LP;
        f32x4_t o[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * GM_THREADS;
            o[it] = zero;
            if (i < nitems) {
                const unsigned v = ((unsigned)i * lp_magic) >> 16, ll = (unsigned)i - v * (unsigned)LP;
                const unsigned w = lnb[v * 4 + axis];
                const f32x4_t c = gl[i], x1 = gl[(w & 0xffffu) * (unsigned)LP + ll], x2 = gl[(w >> 16) * (unsigned)LP + ll];
#pragma unroll
                for (int q = 0; q < 4; ++q) o[it][q] = c[q] + 0.5f * (x1[q] + x2[q]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * GM_THREADS;
            if (i < nitems) gl[i] = o[it];
        }
        __syncthreads();
    }
    // slice into the energy: E = -U, then the three FMAs of update_splat_kernel, same order, same weights
    // ((bary * norm) * (compat * alpha))
    if (!EARLY) load_pixels();
#pragma unroll
    for (int it = 0; it < NPI; ++it) {
        const int i = tid + it * GM_THREADS;
        if (i < np * LP) {
            const unsigned t = ((unsigned)i * lp_magic) >> 16, ll = (unsigned)i - t * (unsigned)LP;
            const uint4 gp = gpx[it];
            f32x2_t o01 = {-uu[it][0], -uu[it][1]}, o23 = {-uu[it][2], -uu[it][3]};
            const float wr[3] = {__uint_as_float(gp.y) * a.cag, __uint_as_float(gp.z) * a.cag, __uint_as_float(gp.w) * a.cag};
#pragma unroll
            for (int r = 0; r < ((WSC_GF_ABL & 4) ? 0 : 3); ++r) {
                const f32x4_t row = gl[((gp.x >> (10 * r)) & 1023u) * (unsigned)LP + ll];
                const f32x2_t w2 = {wr[r], wr[r]}, lo = {row[0], row[1]}, hi = {row[2], row[3]};
                o01 = __builtin_elementwise_fma(w2, lo, o01);
                o23 = __builtin_elementwise_fma(w2, hi, o23);
            }
            e4[poff[it]] = f32x4_t{o01[0], o01[1], o23[0], o23[1]};
        }
    }
}

constexpr int TP = 128; // pixels per block of the layout-changing kernels

// unary [B][M][N] (class-major) -> U [pixel][M] and Q = softmax(-U); TP pixels per block.
__global__ __launch_bounds__(TP) void init_q_kernel(const float *__restrict__ unary, int M, int Mp, int N,
                                                     float *__restrict__ u, float *__restrict__ q) {
    extern __shared__ float tile[]; // [M][TP+1] twice
    float *tu = tile, *tq = tile + (size_t)M * (TP + 1);
    const int b = blockIdx.y;
    const int n0 = blockIdx.x * TP;
    const int np = min(TP, N - n0);
    const float *src = unary + (long long)b * M * N;
    for (int m = 0; m < M; ++m)
        if ((int)threadIdx.x < np) tu[m * (TP + 1) + threadIdx.x] = src[(long long)m * N + n0 + threadIdx.x];
    __syncthreads();
    if ((int)threadIdx.x < np) {
        float mx = -3.0e38f;
        for (int m = 0; m < M; ++m) mx = fmaxf(mx, -tu[m * (TP + 1) + threadIdx.x]);
        float s = 0.f;
        for (int m = 0; m < M; ++m) {
            const float e = expf(-tu[m * (TP + 1) + threadIdx.x] - mx);
            tq[m * (TP + 1) + threadIdx.x] = e;
            s += e;
        }
        for (int m = 0; m < M; ++m) tq[m * (TP + 1) + threadIdx.x] = tq[m * (TP + 1) + threadIdx.x] / s;
    }
    __syncthreads();
    const long long obase = ((long long)b * N + n0) * Mp;
    for (int i = threadIdx.x; i < np * Mp; i += TP) {
        const int n = i / Mp, m = i - n * Mp;
        u[obase + i] = m < M ? tu[m * (TP + 1) + n] : 0.f; // padding classes: masked in slice_update
        if (q) q[obase + i] = m < M ? tq[m * (TP + 1) + n] : 0.f;
    }
}

// Ragged batch (wsc_crf_v): the images of one (H, W) group bring their own class-major unaries [M_b][N] and class counts;
// the group runs at Mg = max M_b.  U[pixel][Mp]: the image's own classes, then classes M_b .. Mg - 1 with a unary of 1e30
// (probability 0: exp2(-huge) = 0 in every soft-max, so they never carry mass -- the Potts model couples a class only to
// itself, and adding exact zeros in the soft-max's fixed reduction tree changes no bit of the other classes), then zeros.
struct UnaryVJob {
    const float *u; // [M][N]
    int M;
};
__global__ __launch_bounds__(256) void gather_unary_v_kernel(const UnaryVJob *__restrict__ jobs, int Mg, int Mp, int N,
                                                             float *__restrict__ out) {
    const UnaryVJob job = jobs[blockIdx.y];
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float *dst = out + ((long long)blockIdx.y * N + n) * Mp;
    for (int m0 = 0; m0 < Mp; m0 += 4) {
        f32x4_t v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = m0 + k;
            v[k] = m < job.M ? job.u[(long long)m * N + n] : (m < Mg ? 1.0e30f : 0.f);
        }
        *reinterpret_cast<f32x4_t *>(dst + m0) = v;
    }
}

// Q [pixel][M] -> q_out [B][M][N] and/or argmax [B][N]
__global__ __launch_bounds__(TP) void finish_kernel(const float *__restrict__ q, int M, int Mp, int N,
                                                    float *__restrict__ q_out, int32_t *__restrict__ argmax) {
    extern __shared__ float tile[]; // [M][TP+1]
    const int b = blockIdx.y;
    const int n0 = blockIdx.x * TP;
    const int np = min(TP, N - n0);
    const long long ibase = ((long long)b * N + n0) * Mp;
    for (int i = threadIdx.x; i < np * Mp; i += TP) {
        const int n = i / Mp, m = i - n * Mp;
        if (m < M) tile[m * (TP + 1) + n] = q[ibase + i];
    }
    __syncthreads();
    if ((int)threadIdx.x < np) {
        if (q_out)
            for (int m = 0; m < M; ++m)
                q_out[((long long)b * M + m) * N + n0 + threadIdx.x] = tile[m * (TP + 1) + threadIdx.x];
        if (argmax) {
            int best = 0;
            float bv = tile[threadIdx.x];
            for (int m = 1; m < M; ++m) {
                const float v = tile[m * (TP + 1) + threadIdx.x];
                if (v > bv) {
                    bv = v;
                    best = m;
                }
            }
            argmax[(long long)b * N + n0 + threadIdx.x] = best;
        }
    }
}

__global__ void fill_u32_kernel(unsigned *p, unsigned v, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        p[i] = v;
}
// the hash table (EMPTY_KEY) and its first-toucher array (0x7fffffff) in one pass
__global__ void fill_tables_kernel(unsigned long long *__restrict__ table, unsigned *__restrict__ first, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        table[i] = EMPTY_KEY;
        first[i] = 0x7fffffffu;
    }
}
__global__ void fill_u64_kernel(unsigned long long *p, unsigned long long v, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        p[i] = v;
}

inline int grid1d(long long total, int per_block = 256, int cap = 256 * 32) {
    long long g = (total + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// grid of the replicated kernels: a multiple of rep (xcd_range_rep), at most 256*64 blocks
inline int grid_rep(long long total, int per_block, int rep) {
    const int g = grid1d(total, per_block, 256 * 64);
    return g <= rep ? rep : g / rep * rep;
}

int crf_alloc(wsc_crf *crf, size_t bytes, void **out) {
    void *p = nullptr;
    WSC_TRY(wsc_ctx_cached_alloc(crf->ctx, bytes, &p));
    if (!crf->persist) crf->allocs.push_back(p);
    else crf->persist_allocs.push_back(p);
    *out = p;
    return WSC_OK;
}

struct TempBuf { // build-time scratch handed back to the ctx cache at the end of a lattice build
    wsc_ctx *ctx;
    std::vector<void *> ptrs;
    explicit TempBuf(wsc_ctx *c) : ctx(c) {}
    int alloc(size_t bytes, void **out) {
        void *p = nullptr;
        WSC_TRY(wsc_ctx_cached_alloc(ctx, bytes, &p));
        ptrs.push_back(p);
        *out = p;
        return WSC_OK;
    }
    ~TempBuf() {
        for (void *p : ptrs) wsc_ctx_cached_free(ctx, p); // stream-ordered reuse: no sync needed
    }
};

// the normalisation pass: val = Lattice-splat of the all-ones vector (one value per row)
void splat_ones(wsc_ctx *ctx, const LatticeDev &L, const TileGeom &tg, float *val, float *part) {
    hipLaunchKernelGGL(slot_ones_kernel, dim3((unsigned)L.n_tiles), dim3(256), 0, ctx->stream, L.tslot_start, L.slot_desc,
                       L.tent_w, L.d + 1, tg, part);
    hipLaunchKernelGGL(combine1_kernel, dim3(grid1d(L.rows)), dim3(256), 0, ctx->stream, part, L.row_slot_start, L.rows, val);
}

// rows of a lattice from the slot partials of the splat
void combine4(wsc_ctx *ctx, hipStream_t st, const LatticeDev &L, const float *part, int LP, float *val) {
    WscKernelTimer timer(ctx, WSC_K_BLUR, ((double)L.n_slots + L.rows) * L.rep * L.M_cur * 4);
#ifdef WSC_AB_KNOBS
    const char *be = getenv("WSC_CRF_COMBINE_BALANCED"); // A/B: 0 keeps the lane-group-per-row kernel
#else
    const char *be = nullptr;
#endif
    if (L.sorted_dest)
        hipLaunchKernelGGL(combine4_kernel<true>, dim3(grid_rep((long long)L.rows * L.rep, 256 / LP, L.rep)), dim3(256), 0,
                           st, (const f32x4_t *)part, L.row_slot_start, LP, L.rows, L.n_slots, L.rep, (f32x4_t *)val);
    else if (L.rep == 1 && !(be && atoi(be) == 0))
        hipLaunchKernelGGL(combine4_balanced_kernel, dim3((unsigned)((L.rows + CB_ROWS - 1) / CB_ROWS)), dim3(256), 0, st,
                           (const f32x4_t *)part, L.row_slot_start, (const uint32_t *)L.part_row, LP, L.rows, (f32x4_t *)val);
    else
        hipLaunchKernelGGL(combine4_kernel<false>, dim3(grid_rep((long long)L.rows * L.rep, 256 / LP, L.rep)), dim3(256), 0,
                           st, (const f32x4_t *)part, L.row_slot_start, LP, L.rows, L.n_slots, L.rep, (f32x4_t *)val);
}

// d+1 blur passes of the one-value-per-row normalisation lattice, ping-pong between a and b
float *blur_all1(wsc_ctx *ctx, const LatticeDev &L, float *a, float *b) {
    for (int j = 0; j <= L.d; ++j) {
        hipLaunchKernelGGL(blur1_kernel, dim3(grid1d(L.rows)), dim3(256), 0, ctx->stream, a,
                           L.nbr + (long long)j * L.rows, L.rows, b);
        float *t = a; a = b; b = t;
    }
    return a;
}

// blur_lds_kernel on the rows in `val` when the lattice is per image and every image's vertex count admits a variant
// (BL_VAR: GW classes of its rows in a workgroup's LDS, RPT rows per thread); returns false when the per-pass launches have
// to run.  WSC_CRF_BLUR_LDS=0 (read per call: a test compares the two paths) switches it off.
// The workgroup table of blur_lds_kernel for a class count, written on the ctx's MAIN stream at the start of an inference
// call (through the ctx's page-locked staging buffer: the host vector may be rebuilt by the next call at once), so that
// whichever stream runs the blur later in the call is ordered behind it.
int blur_lds_prepare(wsc_ctx *ctx, const LatticeDev &L, int M) {
    if (L.rep != 1 || !L.img_row || !L.bl_ok) return WSC_OK;
    const int B = (int)L.v_per_image.size();
    if (L.bl_M != M) {
        // workgroup table for this class count: the images are dealt to the 8 XCDs (workgroup b runs on XCD b % 8), an
        // image's class groups follow each other on its XCD (they read the same row lines and neighbour tables: L2 hits)
        std::vector<std::vector<int2>> q(8);
        size_t lds = 0;
        for (int b = 0; b < B; ++b) {
            const int v = blur_lds_variant(L.v_per_image[b]), gw = BL_VAR[v][0];
            lds = std::max(lds, (size_t)(L.v_per_image[b] + 1) * gw * 4);
            for (int g = 0; g * gw < M; ++g) q[b & 7].push_back(make_int2(b | (v << 24), g));
        }
        size_t len = 0;
        for (int x = 0; x < 8; ++x) len = std::max(len, q[x].size());
        std::vector<int2> &h = L.bl_blk_host;
        h.assign(len * 8, make_int2(-1, 0));
        for (int x = 0; x < 8; ++x)
            for (size_t i = 0; i < q[x].size(); ++i) h[i * 8 + x] = q[x][i];
        L.bl_M = -1;
        if ((int)h.size() > L.bl_cap) return WSC_OK; // (sized for M <= 32 at one class per workgroup: the per-pass launches run)
        WSC_TRY(wsc_ctx_upload_small(ctx, L.bl_blk, h.data(), sizeof(int2) * h.size()));
        L.bl_nblk = (int)h.size();
        L.bl_lds = lds;
        L.bl_M = M;
    }
    return WSC_OK;
}
bool blur_lds(wsc_ctx *ctx, hipStream_t st, const LatticeDev &L, int LP, float *val) {
    const int M = L.M_cur;
    if (!ctx->opt[WSC_OPT_CRF_BLUR_ON_CHIP] || L.rep != 1 || !L.img_row || !L.bl_ok || L.bl_M != M) return false;
    if (L.bl_nblk == 0) return true;
    static bool attr_set[64] = {};
    const int dev = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(blur_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return false;
        attr_set[dev] = true;
    }
    BlurLdsArgs ba;
    ba.nbr = L.nbr; ba.img_row = L.img_row; ba.blk = L.bl_blk; ba.val = val; ba.Mp = 4 * LP; ba.M = M; ba.rows = L.rows;
    ba.npass = L.d + 1;
    WscKernelTimer timer(ctx, WSC_K_BLUR, 2.0 * (L.d + 1) * L.rows * M * 4);
    hipLaunchKernelGGL(blur_lds_kernel, dim3((unsigned)L.bl_nblk), dim3(BL_THREADS), L.bl_lds, st, ba);
    return true;
}

// Rows from the splat's slot partials (`part`), then the d+1 blur passes; a / b: two row buffers.  Returns the
// buffer holding the result.  The launches go to `st` (the ctx's main or side stream): the shared ctx is never
// re-pointed, so an error return or another thread's wsc_sync always sees ctx->stream = the main stream.  The timers are
// live only while profiling, which runs without the fork (st == ctx->stream then).
float *combine_blur_all4(wsc_ctx *ctx, hipStream_t st, const LatticeDev &L, int LP, const float *part, float *a, float *b) {
    // WSC_OPT_CRF_FUSED_BLUR = 0 keeps the three separate passes
    const bool fused_off = !ctx->opt[WSC_OPT_CRF_FUSED_BLUR];
    if (L.d == 2 && L.tile_rows && L.tile_pstart && L.n_tiles_occ > 0 && !fused_off) {
        // one read of the partials + one write of the value rows (the halo re-reads come out of L2).  Whole rows
        // per group (6 float4 = 49 KB of LDS, 3 blocks per CU) beat 3 / 2 / 1 float4 per group at 6+ blocks per CU
        WscKernelTimer timer(ctx, WSC_K_BLUR, ((double)L.n_slots + L.rows) * L.rep * L.M_cur * 4);
#ifdef WSC_AB_KNOBS
        const char *ge = getenv("WSC_CRF_GLH");
        const int glh = ge ? atoi(ge) : WSC_GLH;
#else
        constexpr int glh = WSC_GLH;
#endif
        if (glh == 3)
            hipLaunchKernelGGL(blur3_tile_kernel<3>, dim3((unsigned)(L.n_tiles_occ * L.rep)), dim3(GBI * GBJ), 0, st,
                               (const f32x4_t *)nullptr, L.tile_rows, L.tile_list, L.n_tiles_occ, LP, L.rows, L.rep, (f32x4_t *)b,
                               (const f32x4_t *)part, L.tile_pstart, L.n_slots);
        else if (glh == 2)
            hipLaunchKernelGGL(blur3_tile_kernel<2>, dim3((unsigned)(L.n_tiles_occ * L.rep)), dim3(GBI * GBJ), 0, st,
                               (const f32x4_t *)nullptr, L.tile_rows, L.tile_list, L.n_tiles_occ, LP, L.rows, L.rep, (f32x4_t *)b,
                               (const f32x4_t *)part, L.tile_pstart, L.n_slots);
        else
        hipLaunchKernelGGL(blur3_tile_kernel<WSC_GLH>, dim3((unsigned)(L.n_tiles_occ * L.rep)), dim3(GBI * GBJ), 0, st,
                           (const f32x4_t *)nullptr, L.tile_rows, L.tile_list, L.n_tiles_occ, LP, L.rows, L.rep, (f32x4_t *)b,
                           (const f32x4_t *)part, L.tile_pstart, L.n_slots);
        return b;
    }
    combine4(ctx, st, L, part, LP, a);
    if (blur_lds(ctx, st, L, LP, a)) return a;
    for (int j = 0; j <= L.d; ++j) {
        WscKernelTimer timer(ctx, WSC_K_BLUR, 2.0 * L.rows * L.rep * L.M_cur * 4); // read + write every row once
        hipLaunchKernelGGL(blur4_kernel, dim3(grid_rep((long long)L.rows * L.rep, (256 / LP) * 4, L.rep)), dim3(256),
                           0, st, (const f32x4_t *)a, L.nbr + (long long)j * L.rows, LP, L.rows, L.rep,
                           (f32x4_t *)b);
        float *t = a; a = b; b = t;
    }
    return a;
}

// ---- Gaussian blur on chip: per-pixel-tile vertex sets ------------------------------------------------------------
// The update kernel can run the three blur passes of the Gaussian lattice itself: a pixel tile's slice reads only the
// ~120-160 vertices its pixels touch, and their blurred values depend on the splat values of the vertices within one step
// along axis 2, then axis 1, then axis 0 of them (the passes run 0, 1, 2: Permutohedral::compute).  This host pass (one-off
// per cached Gaussian lattice, i.e. per image size and sxy) lists, for every pixel tile, that closed set: the tile's own
// vertices T first, then nbr_2(T), then nbr_1 of all so far, then nbr_0 of all so far -- pass 0 is then exact on T + nbr_2 +
// nbr_1, pass 1 on T + nbr_2, pass 2 on T, whatever the values outside the set are taken to be (zero).  A neighbour that
// does not exist in the lattice or lies outside the set points at the set's zero row (the last, padding, local index).
// Tables: see LatticeDev::gt_*.  Sets of more than GT_MAX_LOCAL vertices (tiny sxy: every pixel its own simplex) switch
// the fusion off for the lattice (gt_rows stays null) and the separate blur kernel runs.
constexpr int GT_MAX_LOCAL = 1022; // local ids are packed in 10 bits; the set's zero row takes one more id
int gauss_fuse_tables(wsc_crf *crf, LatticeDev &L, const TileGeom &tg) {
    wsc_ctx *ctx = crf->ctx;
    const int N = crf->N, rows = L.rows, W = crf->W;
    std::vector<int32_t> off((size_t)N * 3), rss((size_t)rows + 1);
    std::vector<float> bary((size_t)N * 3), norm((size_t)N);
    std::vector<int2> nbr((size_t)3 * rows);
    WSC_HIP(hipMemcpyAsync(off.data(), L.offset, sizeof(int32_t) * off.size(), hipMemcpyDeviceToHost, ctx->stream));
    WSC_HIP(hipMemcpyAsync(bary.data(), L.bary, sizeof(float) * bary.size(), hipMemcpyDeviceToHost, ctx->stream));
    WSC_HIP(hipMemcpyAsync(norm.data(), L.norm, sizeof(float) * norm.size(), hipMemcpyDeviceToHost, ctx->stream));
    WSC_HIP(hipMemcpyAsync(rss.data(), L.row_slot_start, sizeof(int32_t) * rss.size(), hipMemcpyDeviceToHost, ctx->stream));
    WSC_HIP(hipMemcpyAsync(nbr.data(), L.nbr, sizeof(int2) * nbr.size(), hipMemcpyDeviceToHost, ctx->stream));
    WSC_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<int> loc((size_t)rows, -1);
    // pass 1: the sets (row ids in local order) and their sizes
    std::vector<std::vector<int>> sets((size_t)tg.tpi);
    std::vector<int4> cnt((size_t)tg.tpi);
    int nvmax = 0;
    for (int j = 0; j < tg.tpi; ++j) {
        const TileBox tb = tile_box(tg, j);
        std::vector<int> &list = sets[j];
        auto add = [&](int R) {
            if (R > 0 && R < rows && loc[R] < 0) {
                loc[R] = (int)list.size();
                list.push_back(R);
            }
        };
        for (int ty = 0; ty < tb.ch; ++ty)
            for (int tx = 0; tx < tb.cw; ++tx) {
                const size_t n = (size_t)(tb.y0 + ty) * W + tb.x0 + tx;
                for (int r = 0; r < 3; ++r) add(off[n * 3 + r]);
            }
        int4 &c = cnt[j];
        c.x = (int)list.size(); // T: the tile's own vertices -- where pass 2 has to be right
        for (int axis = 2; axis >= 0; --axis) {
            const size_t cur = list.size();
            for (size_t i = 0; i < cur; ++i) {
                const int2 nb = nbr[(size_t)axis * rows + list[i]];
                add(nb.x);
                add(nb.y);
            }
            if (axis == 2) c.y = (int)list.size(); // T + nbr_2: where pass 1 has to be right
            if (axis == 1) c.z = (int)list.size(); // ... + nbr_1: where pass 0 has to be right
        }
        c.w = (int)list.size();
        for (int R : list) loc[R] = -1;
        if ((int)list.size() > nvmax) nvmax = (int)list.size();
    }
    if (nvmax > GT_MAX_LOCAL || nvmax == 0) return WSC_OK; // no fusion for this lattice
    // pass 2: the tables, padded to `stride` entries per tile; local id stride - 1 is every tile's zero row (a padding
    // entry: no partial rows), so the kernel needs no per-tile count
    const int stride = nvmax + 1, zr = stride - 1;
    std::vector<int2> trows((size_t)tg.tpi * stride, make_int2(0, 0));
    std::vector<uint4> tnbr((size_t)tg.tpi * stride, make_uint4((unsigned)zr | ((unsigned)zr << 16), (unsigned)zr | ((unsigned)zr << 16),
                                                                (unsigned)zr | ((unsigned)zr << 16), 0u));
    std::vector<uint4> pix((size_t)N);
    for (int j = 0; j < tg.tpi; ++j) {
        const TileBox tb = tile_box(tg, j);
        const std::vector<int> &list = sets[j];
        const int nv = (int)list.size();
        for (int v = 0; v < nv; ++v) loc[list[v]] = v;
        auto lid = [&](int R) { return (R > 0 && R < rows && loc[R] >= 0) ? loc[R] : zr; };
        for (int v = 0; v < nv; ++v) {
            const int R = list[v];
            trows[(size_t)j * stride + v] = make_int2(rss[R], rss[R + 1] - rss[R]);
            unsigned w[3];
            for (int axis = 0; axis < 3; ++axis) {
                const int2 nb = nbr[(size_t)axis * rows + R];
                w[axis] = (unsigned)lid(nb.x) | ((unsigned)lid(nb.y) << 16);
            }
            tnbr[(size_t)j * stride + v] = make_uint4(w[0], w[1], w[2], 0u);
        }
        for (int ty = 0; ty < tb.ch; ++ty)
            for (int tx = 0; tx < tb.cw; ++tx) {
                const size_t n = (size_t)(tb.y0 + ty) * W + tb.x0 + tx;
                unsigned ids = 0;
                uint32_t bw[3];
                for (int r = 0; r < 3; ++r) {
                    ids |= (unsigned)lid(off[n * 3 + r]) << (10 * r);
                    const float bn = bary[n * 3 + r] * norm[n]; // the slice weight up to compat * alpha: (bary * norm) * (compat * alpha)
                    memcpy(&bw[r], &bn, sizeof(float));
                }
                pix[n] = make_uint4(ids, bw[0], bw[1], bw[2]);
            }
        for (int R : list) loc[R] = -1;
    }
    WSC_TRY(crf_alloc(crf, sizeof(int4) * cnt.size(), (void **)&L.gt_cnt));
    WSC_HIP(hipMemcpyAsync(L.gt_cnt, cnt.data(), sizeof(int4) * cnt.size(), hipMemcpyHostToDevice, ctx->stream));
    WSC_TRY(crf_alloc(crf, sizeof(int2) * trows.size(), (void **)&L.gt_rows));
    WSC_TRY(crf_alloc(crf, sizeof(uint4) * tnbr.size(), (void **)&L.gt_nbr));
    WSC_TRY(crf_alloc(crf, sizeof(uint4) * pix.size(), (void **)&L.gt_pix));
    WSC_HIP(hipMemcpyAsync(L.gt_rows, trows.data(), sizeof(int2) * trows.size(), hipMemcpyHostToDevice, ctx->stream));
    WSC_HIP(hipMemcpyAsync(L.gt_nbr, tnbr.data(), sizeof(uint4) * tnbr.size(), hipMemcpyHostToDevice, ctx->stream));
    WSC_HIP(hipMemcpyAsync(L.gt_pix, pix.data(), sizeof(uint4) * pix.size(), hipMemcpyHostToDevice, ctx->stream));
    WSC_HIP(hipStreamSynchronize(ctx->stream)); // the sources are pageable host vectors
    L.gt_stride = stride;
    return WSC_OK;
}

constexpr int WSC_RETRY_FULL_TABLE = 1; // internal status of build_lattice: the right-sized hash table overflowed

template <int D>
int build_lattice(wsc_crf *crf, LatticeDev &L, const uint8_t *rgb_dev, float sxy, float srgb, bool shared,
                  bool full_table) {
    wsc_ctx *ctx = crf->ctx;
    // shared: build the lattice of ONE image and let all crf->B images use it (position-only features)
    const int B = shared ? 1 : crf->B, N = crf->N, dp1 = D + 1;
    L.rep = shared ? crf->B : 1;
    const long long npix = (long long)B * N;
    const long long total = npix * dp1;
    WSC_CHECK(total < (1ll << 31), WSC_ERR_CAPACITY, "CRF batch too large: %lld lattice entries", total);
    L.d = D;
    L.n_pix = npix;
    L.alpha = 1.0f / (1.0f + powf(2.0f, -(float)D));
    // Hash table slots per image.  Worst case (every pixel owns its d+1 vertices) needs 2 N (d+1); natural
    // images have 1-2 orders of magnitude fewer vertices (10.8 k for 103 k pixels on the bench set), so the
    // first attempt uses N (d+1) / 8 slots rounded up to a power of two (1 MB instead of 16 MB per image at
    // 321^2: the probes stay in L2 and the table fill drops from 0.8 GB to 50 MB per batch).  An insert that
    // probes more than HASH_MAX_PROBE slots flags the build, which is then repeated with the worst-case size.
    long long cap = 1;
    const long long want = full_table ? 2ll * N * dp1 : (long long)N * dp1 / 8;
    while (cap < want || cap < 1024) cap <<= 1;
    WSC_CHECK(B * cap < (1ll << 31), WSC_ERR_CAPACITY, "CRF batch too large for the hash tables");
    WSC_CHECK(cap <= (1ll << 28), WSC_ERR_CAPACITY, "CRF image too large: hash slots do not fit 28 bits"); // (tile pass: slot | r << 28)

    TempBuf tmp(ctx);
    unsigned long long *table;
    int32_t *first, *slot2row, *rowimg;
    unsigned *bitmap, *wcount, *wprefix, *sums, *bound_dev;
    unsigned long long *rowkey;
    int *err;
    WSC_TRY(tmp.alloc(sizeof(unsigned long long) * B * cap, (void **)&table));
    WSC_TRY(tmp.alloc(sizeof(int32_t) * B * cap, (void **)&first));
    WSC_TRY(tmp.alloc(sizeof(int32_t) * B * cap, (void **)&slot2row));
    const long long nw = (total + 31) / 32 + 1; // first-toucher bitmap over the entries
    WSC_TRY(tmp.alloc(sizeof(unsigned) * nw, (void **)&bitmap));
    WSC_TRY(tmp.alloc(sizeof(unsigned) * nw, (void **)&wcount));
    WSC_TRY(tmp.alloc(sizeof(unsigned) * (nw + 1), (void **)&wprefix));
    WSC_TRY(tmp.alloc(sizeof(unsigned) * (size_t)(B + 1), (void **)&bound_dev));
    const int nblk = (int)((nw + SCAN_CHUNK - 1) / SCAN_CHUNK);
    WSC_TRY(tmp.alloc(sizeof(unsigned) * (nblk + 2), (void **)&sums));
    // zero-initialised scratch in ONE block (one memset instead of three): err | redo_count | tile_nslots[n_tiles + 1]
    const TileGeom tg0 = make_geom(crf->H, crf->W);
    const int n_tiles0 = B * tg0.tpi;
    unsigned *zblk;
    WSC_TRY(tmp.alloc(sizeof(unsigned) * (size_t)(n_tiles0 + 1 + 4), (void **)&zblk));
    err = reinterpret_cast<int *>(zblk);
    WSC_TRY(crf_alloc(crf, sizeof(int32_t) * total, (void **)&L.offset));
    WSC_TRY(crf_alloc(crf, sizeof(float) * total, (void **)&L.bary));
    WSC_TRY(crf_alloc(crf, sizeof(float) * npix, (void **)&L.norm));
    WSC_TRY(crf_alloc(crf, sizeof(float) * total, (void **)&L.tent_w));
    WSC_TRY(crf_alloc(crf, sizeof(uint8_t) * total, (void **)&L.tent_p));

    hipLaunchKernelGGL(fill_tables_kernel, dim3(grid1d(B * cap)), dim3(256), 0, ctx->stream, table, (unsigned *)first,
                       (long long)B * cap);
    WSC_HIP(hipMemsetAsync(zblk, 0, sizeof(unsigned) * (size_t)(n_tiles0 + 1 + 4), ctx->stream));

    EmbedArgs ea;
    ea.rgb = rgb_dev; ea.B = B; ea.H = crf->H; ea.W = crf->W; ea.N = N;
    ea.inv_sxy_den = sxy; ea.inv_srgb_den = srgb;
    {   // Permutohedral::init: scale_factor[i] = 1/sqrt((i+2)(i+1)) * sqrt(2/3)*(d+1)
        const float inv_std_dev = (float)(std::sqrt(2.0 / 3.0) * (D + 1));
        for (int i = 0; i < 5; ++i)
            ea.scale[i] = i < D ? (float)(1.0 / std::sqrt((double)((i + 2) * (i + 1))) * inv_std_dev) : 0.f;
    }
    ea.table = table; ea.cap_mask = (unsigned)(cap - 1); ea.cap = cap;
    ea.bary = L.bary; ea.first = first; ea.err = err;
    const TileGeom tg = make_geom(crf->H, crf->W);
    L.n_tiles = B * tg.tpi;
    int32_t *sslot;
    unsigned *tile_nslots;
    WSC_TRY(tmp.alloc(sizeof(int32_t) * total, (void **)&sslot));
    tile_nslots = zblk + 4; // (zeroed above)
    {
        // small-table launch for every tile, full-table launch for the ones that gave up (WSC_OPT_CRF_EMBED_FULL: full table
        // for every tile, the single-launch form); WSC_OPT_CRF_RANK_BALLOT: the ballot-matching rank walk for every tile
        const int fb = ctx->opt[WSC_OPT_CRF_RANK_BALLOT] ? 1 : 0;
        if (ctx->opt[WSC_OPT_CRF_EMBED_FULL]) {
            hipLaunchKernelGGL(tile_embed_full_kernel<D>, dim3((unsigned)L.n_tiles), dim3(256), 0, ctx->stream, ea, tg, L.tent_w,
                               L.tent_p, sslot, tile_nslots, fb, (const int32_t *)nullptr, (const unsigned *)nullptr, L.n_tiles);
        } else {
            int32_t *redo_list;
            unsigned *redo_count;
            WSC_TRY(tmp.alloc(sizeof(int32_t) * (size_t)L.n_tiles, (void **)&redo_list));
            redo_count = zblk + 1; // (zeroed above)
            hipLaunchKernelGGL(tile_embed_kernel<D>, dim3((unsigned)L.n_tiles), dim3(256), 0, ctx->stream, ea, tg, L.tent_w, L.tent_p,
                               sslot, tile_nslots, fb, redo_list, redo_count);
            // (the redo list is usually empty or short: three blocks per CU -- what the 53 KB table admits -- walk it; 2048 blocks
            // that mostly found nothing to do took 23 us to come and go)
            hipLaunchKernelGGL(tile_embed_full_kernel<D>, dim3((unsigned)std::min(L.n_tiles, 3 * ctx->num_cus)), dim3(256), 0, ctx->stream, ea,
                               tg, L.tent_w, L.tent_p, sslot, tile_nslots, fb, (const int32_t *)redo_list,
                               (const unsigned *)redo_count, L.n_tiles);
        }
    }
    const int per_img = N * dp1; // entries of one image (total < 2^31 checked above)
    const dim3 grid_tab((unsigned)grid1d(cap, 256, B >= 32 ? 256 : 8192 / (B > 0 ? B : 1)), (unsigned)B);
    WSC_HIP(hipMemsetAsync(bitmap, 0, sizeof(unsigned) * nw, ctx->stream));
    hipLaunchKernelGGL(first_bits_kernel, grid_tab, dim3(256), 0, ctx->stream, first, cap, per_img, bitmap);
    hipLaunchKernelGGL(popc_words_kernel, dim3(grid1d(nw)), dim3(256), 0, ctx->stream, bitmap, nw, wcount);
    WSC_TRY(exclusive_scan(ctx, wcount, nw, wprefix, sums));
    hipLaunchKernelGGL(image_bounds_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, ctx->stream, bitmap, wprefix, per_img, B,
                       bound_dev);
    // vertex counts: grand total and per-image boundaries
    int herr = 0;
    unsigned vtot = 0;
    WSC_HIP(hipMemcpyAsync(&vtot, sums + nblk, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    WSC_HIP(hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    std::vector<unsigned> bound(B, 0);
    WSC_HIP(hipMemcpyAsync(bound.data(), bound_dev, sizeof(unsigned) * (size_t)B, hipMemcpyDeviceToHost, ctx->stream));
    WSC_HIP(hipStreamSynchronize(ctx->stream));
    WSC_CHECK((herr & 1) == 0, WSC_ERR_KEY_RANGE,
              "CRF lattice coordinate outside the packed-key range (sxy=%g srgb=%g too small for this image size)",
              (double)sxy, (double)srgb);
    if (herr & 2) {
        WSC_CHECK(!full_table, WSC_ERR_CAPACITY, "CRF lattice hash table overflow at worst-case size");
        return WSC_RETRY_FULL_TABLE; // crf_alloc'ed arrays of this attempt are released with the crf / reused
    }
    L.rows = (int)vtot + 1;
    L.v_per_image.resize(crf->B);
    if (shared)
        for (int b = 0; b < crf->B; ++b) L.v_per_image[b] = (int)vtot;
    else
        for (int b = 0; b < B; ++b) L.v_per_image[b] = (int)((b + 1 < B ? bound[b + 1] : vtot) - bound[b]);
    if (!shared) { // row range of every image (blur_lds_kernel takes one image's rows into a workgroup's LDS)
        WSC_TRY(crf_alloc(crf, sizeof(int32_t) * (size_t)(B + 1), (void **)&L.img_row));
        hipLaunchKernelGGL(image_rows_kernel, dim3((unsigned)(B / 64 + 1)), dim3(64), 0, ctx->stream, bound_dev, B, (int)vtot + 1,
                           L.img_row);
        L.max_img_rows = 0;
        for (int b = 0; b < B; ++b) L.max_img_rows = std::max(L.max_img_rows, (int)L.v_per_image[b]);
        L.bl_ok = B < (1 << 24);
        for (int b = 0; b < B; ++b)
            if (blur_lds_variant(L.v_per_image[b]) < 0) L.bl_ok = false; // an image with too many vertices for any variant
        if (L.bl_ok) { // the workgroup table: at most 32 class groups per image, the images dealt to 8 XCD queues
            L.bl_cap = 8 * ((B + 7) / 8) * 32;
            WSC_TRY(crf_alloc(crf, sizeof(int2) * (size_t)L.bl_cap, (void **)&L.bl_blk));
        }
    }

    WSC_TRY(tmp.alloc(sizeof(unsigned long long) * L.rows, (void **)&rowkey));
    WSC_TRY(tmp.alloc(sizeof(int32_t) * L.rows, (void **)&rowimg));
    WSC_TRY(crf_alloc(crf, sizeof(int2) * (size_t)dp1 * L.rows, (void **)&L.nbr));

    hipLaunchKernelGGL(assign_rows_kernel, grid_tab, dim3(256), 0, ctx->stream, first, table, cap, per_img, bitmap, wprefix, slot2row,
                       rowkey, rowimg);
    {   // splat tables: slots of the grouped tile entries, partial rows of each lattice row
        int32_t *slot_row;
        unsigned *sums2, *row_nslots, *cursor, *sums3;
        const int nb2 = (L.n_tiles + 1 + SCAN_CHUNK - 1) / SCAN_CHUNK, nb3 = (L.rows + 1 + SCAN_CHUNK - 1) / SCAN_CHUNK;
        WSC_TRY(tmp.alloc(sizeof(unsigned) * (nb2 + 2), (void **)&sums2));
        WSC_TRY(tmp.alloc(sizeof(unsigned) * 2 * (size_t)(L.rows + 1), (void **)&row_nslots)); // row_nslots | cursor: one memset
        cursor = row_nslots + (L.rows + 1);
        WSC_TRY(tmp.alloc(sizeof(unsigned) * (nb3 + 2), (void **)&sums3));
        WSC_TRY(crf_alloc(crf, sizeof(int32_t) * (L.n_tiles + 2), (void **)&L.tslot_start));
        WSC_TRY(crf_alloc(crf, sizeof(int32_t) * (L.rows + 2), (void **)&L.row_slot_start));
        WSC_HIP(hipMemsetAsync(row_nslots, 0, sizeof(unsigned) * 2 * (size_t)(L.rows + 1), ctx->stream));
        WSC_TRY(exclusive_scan(ctx, tile_nslots, L.n_tiles + 1, (unsigned *)L.tslot_start, sums2));
        unsigned ns = 0;
        WSC_HIP(hipMemcpyAsync(&ns, sums2 + nb2, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        WSC_HIP(hipStreamSynchronize(ctx->stream));
        L.n_slots = (int)ns;
        WSC_TRY(crf_alloc(crf, sizeof(int2) * (size_t)(L.n_slots + 1), (void **)&L.slot_desc));
        WSC_TRY(tmp.alloc(sizeof(int32_t) * (size_t)(L.n_slots + 1), (void **)&slot_row));
        unsigned *slot_key = nullptr;
        int32_t *dest_slot = nullptr;
        if (D == 2) {
            WSC_TRY(tmp.alloc(sizeof(unsigned) * (size_t)(L.n_slots + 1), (void **)&slot_key));
            WSC_TRY(tmp.alloc(sizeof(int32_t) * (size_t)(L.n_slots + 1), (void **)&dest_slot));
        }
        hipLaunchKernelGGL(tile_slots_kernel, dim3((unsigned)L.n_tiles), dim3(256), 0, ctx->stream, sslot, L.tent_p, L.offset, slot2row,
                           cap, dp1, tg,
                           L.tslot_start, L.slot_desc, slot_row, slot_key, row_nslots);
        WSC_TRY(exclusive_scan(ctx, row_nslots, L.rows + 1, (unsigned *)L.row_slot_start, sums3));
        if (!shared) WSC_TRY(crf_alloc(crf, sizeof(uint32_t) * (size_t)(L.n_slots + 1), (void **)&L.part_row));
        hipLaunchKernelGGL(slot_dest_kernel, dim3(grid1d(L.n_slots)), dim3(256), 0, ctx->stream, slot_row, L.n_slots,
                           L.row_slot_start, cursor, L.slot_desc, L.part_row);
        if (D == 2) { // Gaussian lattice (built once per image size): deterministic partial-row order, plain fp32 combine
            hipLaunchKernelGGL(dest_inverse_kernel, dim3(grid1d(L.n_slots)), dim3(256), 0, ctx->stream, L.slot_desc, L.n_slots,
                               dest_slot);
            hipLaunchKernelGGL(row_sort_dest_kernel, dim3(grid1d(L.rows)), dim3(256), 0, ctx->stream, slot_key, dest_slot,
                               L.row_slot_start, L.rows, L.slot_desc);
            L.sorted_dest = true;
        }
    }
    WSC_HIP(hipMemsetAsync(L.nbr, 0, sizeof(int2) * (size_t)dp1 * L.rows, ctx->stream));
    hipLaunchKernelGGL(neighbors_kernel<D>, dim3((unsigned)grid1d(L.rows, 256, 4096), (unsigned)dp1), dim3(256), 0, ctx->stream, rowkey,
                       rowimg, table, slot2row, cap, (unsigned)(cap - 1), L.rows, L.nbr);
    WSC_HIP(hipGetLastError());
    if (D == 2) { // tile tables of the fused blur (one-off per cached Gaussian lattice: the host syncs are fine)
        int2 *ij;
        int *bbox;
        WSC_TRY(tmp.alloc(sizeof(int2) * L.rows, (void **)&ij));
        WSC_TRY(tmp.alloc(sizeof(int) * 5, (void **)&bbox));
        const int init[5] = {0x7fffffff, 0x7fffffff, -0x7fffffff, -0x7fffffff, 0};
        WSC_TRY(wsc_ctx_upload_small(ctx, bbox, init, sizeof(init)));
        hipLaunchKernelGGL(gauss_ij_kernel, dim3(grid1d(L.rows)), dim3(256), 0, ctx->stream, rowkey, L.rows, ij, bbox);
        int hb[5] = {0, 0, -1, -1, 1};
        WSC_HIP(hipMemcpyAsync(hb, bbox, sizeof(hb), hipMemcpyDeviceToHost, ctx->stream));
        WSC_HIP(hipStreamSynchronize(ctx->stream));
        const long long nti = hb[2] >= hb[0] ? ((long long)hb[2] - hb[0]) / GTI + 1 : 0;
        const long long ntj = hb[3] >= hb[1] ? ((long long)hb[3] - hb[1]) / GTJ + 1 : 0;
        // a lattice that is not the expected dense (i, j) plane, or a degenerate / huge box: keep the three passes
        if (hb[4] == 0 && nti > 0 && ntj > 0 && nti * ntj <= 4ll * L.rows + 64) {
            const long long nt = nti * ntj;
            int32_t *occ;
            WSC_TRY(tmp.alloc(sizeof(int32_t) * nt, (void **)&occ));
            WSC_TRY(crf_alloc(crf, sizeof(int32_t) * nt * GBI * GBJ, (void **)&L.tile_rows));
            WSC_HIP(hipMemsetAsync(L.tile_rows, 0, sizeof(int32_t) * nt * GBI * GBJ, ctx->stream));
            WSC_HIP(hipMemsetAsync(occ, 0, sizeof(int32_t) * nt, ctx->stream));
            hipLaunchKernelGGL(gauss_tile_fill_kernel, dim3(grid1d(L.rows)), dim3(256), 0, ctx->stream, ij, L.rows, hb[0],
                               hb[1], (int)nti, (int)ntj, L.tile_rows, occ);
            std::vector<int32_t> hocc(nt), list;
            WSC_HIP(hipMemcpyAsync(hocc.data(), occ, sizeof(int32_t) * nt, hipMemcpyDeviceToHost, ctx->stream));
            WSC_HIP(hipStreamSynchronize(ctx->stream));
            for (long long t = 0; t < nt; ++t)
                if (hocc[t]) list.push_back((int32_t)t);
            L.n_tiles_occ = (int)list.size();
            if (L.n_tiles_occ > 0) {
                WSC_TRY(crf_alloc(crf, sizeof(int32_t) * list.size(), (void **)&L.tile_list));
                WSC_HIP(hipMemcpyAsync(L.tile_list, list.data(), sizeof(int32_t) * list.size(), hipMemcpyHostToDevice,
                                       ctx->stream));
                WSC_HIP(hipStreamSynchronize(ctx->stream)); // `list` is pageable host memory
                WSC_TRY(crf_alloc(crf, sizeof(int2) * nt * GBI * GBJ, (void **)&L.tile_pstart));
                hipLaunchKernelGGL(gauss_tile_pstart_kernel, dim3(grid1d(nt * GBI * GBJ)), dim3(256), 0, ctx->stream,
                                   L.tile_rows, L.row_slot_start, nt * GBI * GBJ, L.tile_pstart);
            }
        }
    }
    // norm = 1/sqrt(Lattice(1) + 1e-20)
    float *va, *vb, *vp;
    WSC_TRY(tmp.alloc(sizeof(float) * L.rows, (void **)&va));
    WSC_TRY(tmp.alloc(sizeof(float) * L.rows, (void **)&vb));
    WSC_TRY(tmp.alloc(sizeof(float) * (size_t)(L.n_slots + 1), (void **)&vp));
    splat_ones(ctx, L, make_geom(crf->H, crf->W), va, vp);
    float *res = blur_all1(ctx, L, va, vb);
    // the bilateral lattice of a crf whose updates start from the on-chip Gaussian message: their 52-byte record is
    // written by the same pass (crf_bilateral_records finds it done)
    uint32_t *rec_b = nullptr;
    if (D == 5 && !shared && crf->lat[0].gt_rows && !crf->pix_rec_b) {
        WSC_TRY(crf_alloc(crf, sizeof(uint32_t) * 13 * (size_t)npix, (void **)&crf->pix_rec_b));
        rec_b = crf->pix_rec_b;
    }
    if (!shared && D == 5) { // per-image lattices of a batch: norm + record + entry scaling in one pass per pixel tile
        hipLaunchKernelGGL(slice_norm_tile_kernel, dim3((unsigned)L.n_tiles), dim3(256), 0, ctx->stream, L.offset, L.bary, dp1,
                           L.alpha, res, make_geom(crf->H, crf->W), L.norm, rec_b, L.tent_w, L.tent_p);
    } else {
        hipLaunchKernelGGL(slice_norm_kernel, dim3(grid1d(npix)), dim3(256), 0, ctx->stream, L.offset, L.bary, dp1,
                           L.alpha, res, npix, L.norm, rec_b);
        hipLaunchKernelGGL(tile_scale_entries_kernel, dim3((unsigned)L.n_tiles), dim3(256), 0, ctx->stream, L.norm, dp1,
                           make_geom(crf->H, crf->W), L.tent_w, L.tent_p);
    }
    WSC_HIP(hipGetLastError());
    // (the tile vertex sets of the on-chip Gaussian message -- gauss_fuse_tables: a device -> host copy of the lattice and a
    // host pass, 3-4 ms -- are added when a cached lattice is used a SECOND time: wsc_crf_create)
    return WSC_OK;
}

struct GaussCache {
    int H, W;
    float sxy;
    LatticeDev L;
    std::vector<void *> blocks; // its device arrays (cached-alloc blocks of the ctx): handed back when the entry is evicted
    int users = 0;              // live wsc_crf objects whose lat[0] aliases L
    unsigned long long last_use = 0;
    bool fuse_pending = true;   // the tile vertex sets of the on-chip message are built at the entry's second use
};
// distinct image sizes kept per ctx (~25 MB each at 375 x 500).  When the table is full the least recently used entry that
// no live wsc_crf refers to makes room (round 3 had no eviction: after 64 sizes every later size was rebuilt per call,
// without the on-chip message path -- cam_to_ir_label walks hundreds of sizes).  Only when all entries are in use is a
// lattice built for the one call, without the host-built tile vertex sets of the on-chip message path.
constexpr int GAUSS_CACHE_MAX = 64;
std::atomic<unsigned long long> g_gauss_use_clock{0}; // (contexts of several host threads share it: only the order matters)
void gauss_cache_delete(void *p) { delete static_cast<GaussCache *>(p); }

// slice: messages of both lattices are read (false before the first iteration); splat: the result is splatted
// (false in the last iteration, whose Q is written to a.q instead)
// block size of the update kernel (WSC_CRF_UPD_THREADS: A/B runs)
int update_threads() {
#ifdef WSC_AB_KNOBS
    const char *te = getenv("WSC_CRF_UPD_THREADS");
    const int nthr = te ? atoi(te) : 256;
    return nthr == 512 ? 512 : 256; // (the kernel's per-thread table registers cover a tile with >= 256 threads)
#else
    return 256;
#endif
}
size_t update_splat_lds(int LP) { return sizeof(f32x4_t) * TILE_PIX * LP + sizeof(uint2) * GATHER_ENT + sizeof(int2) * GATHER_SB; }
size_t gauss_msg_lds(int LP, int gt_stride) { return ((size_t)LP * sizeof(f32x4_t) + sizeof(uint4)) * (size_t)gt_stride; }
// can the Gaussian message be formed on chip (gauss_msg_kernel) for this call?  The tile vertex sets must exist (sets within
// the local-id range) and fit the kernel's LDS and its per-thread item bound.
bool update_gf_ok(const wsc_ctx *ctx, const LatticeDev &G, int LP) {
    if (!ctx->opt[WSC_OPT_CRF_GAUSS_ON_CHIP]) return false;
    if (!G.gt_rows || G.rep < 1 || G.gt_stride <= 0) return false;
    if ((long long)G.gt_stride * LP > 6ll * GM_THREADS || (long long)G.gt_stride * LP >= 8192) return false;
    return gauss_msg_lds(LP, G.gt_stride) <= 64 * 1024;
}

int launch_gauss_msg(wsc_ctx *ctx, hipStream_t st, const GaussMsgArgs &g, double bytes) {
    WscKernelTimer timer(ctx, WSC_K_GAUSS_MSG, bytes);
    const dim3 grid((unsigned)(g.B * g.tg.tpi)), block(GM_THREADS);
    size_t lds = gauss_msg_lds(g.LP, g.gt_stride);
#ifdef WSC_AB_KNOBS
    const char *le = getenv("WSC_CRF_GM_LDS"); // A/B: pad the LDS request (bytes) to cap the blocks per CU
    if (le && (size_t)atoi(le) > lds && atoi(le) <= 64 * 1024) lds = (size_t)atoi(le);
#endif
    if ((long long)g.gt_stride * g.LP <= 4ll * GM_THREADS && g.LP <= 6)
        hipLaunchKernelGGL((gauss_msg_kernel<4, 3>), grid, block, lds, st, g);
    else
        hipLaunchKernelGGL((gauss_msg_kernel<6, 4>), grid, block, lds, st, g);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

// gf: the energy starts from the E buffer of gauss_msg_kernel (a.u points at it), no Gaussian rows are gathered
// LDS of the FG variants: Q stage + splat tables / records + the tile's closed vertex set (rows + neighbour words)
size_t update_fg_lds(int LP, int gt_stride) { return update_splat_lds(LP) + gauss_msg_lds(LP, gt_stride); }
// can the Gaussian message be formed INSIDE the update kernel?  Two blocks of GM_THREADS threads per CU must fit
bool update_fg_ok(const wsc_ctx *ctx, const LatticeDev &G, int LP) {
    if (!ctx->opt[WSC_OPT_CRF_MSG_IN_UPDATE] || !update_gf_ok(ctx, G, LP)) return false;
    return update_fg_lds(LP, G.gt_stride) <= 80 * 1024;
}
// fg: the Gaussian message is formed inside the kernel (a.u = U, a.part_g_in = the previous splat's Gaussian partials)
int launch_update(wsc_ctx *ctx, const UpdateArgs &a, bool slice, bool splat, bool gf, bool fg = false, double fg_bytes = 0.0) {
    const double npix = (double)a.B * a.tg.H * a.tg.W;
    // algorithmic bytes: read U, write Q; slice: index+weight of both lattices (9 entries of 8 bytes) and the
    // two messages the reference materialises (N*M*4 each); splat: read Q for both lattices + index+weight
    // (gf: the Gaussian lattice's slice -- 3 of the 9 entries and one of the two messages -- is gauss_msg_kernel's work and
    // is accounted there)
    const double by = npix * (2.0 * a.M * 4 + (slice ? (gf ? 6 * 8 + 1.0 * a.M * 4 : 9 * 8 + 2.0 * a.M * 4) : 0.0) +
                              (splat ? 9 * 8 + 2.0 * a.M * 4 : 0.0));
    WscKernelTimer timer(ctx, WSC_K_SLICE_UPDATE, by + (fg ? fg_bytes : 0.0));
    const dim3 grid((unsigned)(a.B * a.tg.tpi)), block(update_threads());
    size_t lds = splat ? update_splat_lds(a.LP) : 0;
    if (fg) {
        const size_t l = update_fg_lds(a.LP, a.gt_stride);
        const bool small = (long long)a.gt_stride * a.LP <= 4ll * GM_THREADS && a.LP <= 6;
        auto set_lds = [&](const void *fn) {
            static bool done[64][4] = {};
            const int dev = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0, vi = (small ? 0 : 2) + (splat ? 0 : 1);
            if (!done[dev][vi] && l > 48 * 1024) {
                (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
                done[dev][vi] = true;
            }
        };
        if (small) {
            if (splat) {
                set_lds(reinterpret_cast<const void *>(update_splat_kernel<true, true, true, true, 4>));
                hipLaunchKernelGGL((update_splat_kernel<true, true, true, true, 4>), grid, dim3(GM_THREADS), l, ctx->stream, a);
            } else {
                set_lds(reinterpret_cast<const void *>(update_splat_kernel<true, false, true, true, 4>));
                hipLaunchKernelGGL((update_splat_kernel<true, false, true, true, 4>), grid, dim3(GM_THREADS), l, ctx->stream, a);
            }
        } else {
            if (splat) {
                set_lds(reinterpret_cast<const void *>(update_splat_kernel<true, true, true, true, 6>));
                hipLaunchKernelGGL((update_splat_kernel<true, true, true, true, 6>), grid, dim3(GM_THREADS), l, ctx->stream, a);
            } else {
                set_lds(reinterpret_cast<const void *>(update_splat_kernel<true, false, true, true, 6>));
                hipLaunchKernelGGL((update_splat_kernel<true, false, true, true, 6>), grid, dim3(GM_THREADS), l, ctx->stream, a);
            }
        }
        WSC_HIP(hipGetLastError());
        return WSC_OK;
    }
#ifdef WSC_AB_KNOBS
    const char *de = getenv("WSC_CRF_UPD_DMA"); // A/B: 0 keeps the register loads of E and the records
#else
    const char *de = nullptr;
#endif
    const bool dma = slice && gf && !(de && atoi(de) == 0);
    if (dma) lds = update_splat_lds(a.LP); // the last update stages E + records too
    if (dma) {
        if (splat) hipLaunchKernelGGL((update_splat_kernel<true, true, true, true>), grid, block, lds, ctx->stream, a);
        else hipLaunchKernelGGL((update_splat_kernel<true, false, true, true>), grid, block, lds, ctx->stream, a);
    } else if (slice && gf) {
        if (splat) hipLaunchKernelGGL((update_splat_kernel<true, true, true>), grid, block, lds, ctx->stream, a);
        else hipLaunchKernelGGL((update_splat_kernel<true, false, true>), grid, block, lds, ctx->stream, a);
    } else if (slice && splat) hipLaunchKernelGGL((update_splat_kernel<true, true>), grid, block, lds, ctx->stream, a);
    else if (slice) hipLaunchKernelGGL((update_splat_kernel<true, false>), grid, block, lds, ctx->stream, a);
    else if (splat && !(de && atoi(de) == 0)) // the first update: U rows staged by LDS-DMA like the E rows of the later ones
        hipLaunchKernelGGL((update_splat_kernel<false, true, false, true>), grid, block, lds, ctx->stream, a);
    else if (splat) hipLaunchKernelGGL((update_splat_kernel<false, true>), grid, block, lds, ctx->stream, a);
    else hipLaunchKernelGGL((update_splat_kernel<false, false>), grid, block, lds, ctx->stream, a);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

// the 80-byte per-pixel records (pack_pixels_kernel) of a crf, built on `st`.  The block comes from the BUILD ctx's
// stream-ordered cache: when `st` is another stream, it first waits for everything enqueued on the build stream so far
// (the block's previous user), and the build stream waits for this launch before it can hand the block on (crf->use_ev).
int crf_full_records(wsc_crf *crf, hipStream_t st) {
    if (crf->pix_rec) return WSC_OK;
    wsc_ctx *bctx = crf->ctx;
    WSC_TRY(crf_alloc(crf, sizeof(uint4) * 5 * (size_t)crf->B * crf->N, (void **)&crf->pix_rec));
    if (st != bctx->stream) {
        WSC_HIP(hipEventRecord(bctx->join_ev, bctx->stream));
        WSC_HIP(hipStreamWaitEvent(st, bctx->join_ev, 0));
    }
    hipLaunchKernelGGL(pack_pixels_kernel, dim3((unsigned)grid1d((long long)crf->N * 5, 256, 2048), (unsigned)crf->B), dim3(256), 0, st,
                       crf->lat[0].offset, crf->lat[0].bary, crf->lat[0].norm, crf->lat[1].offset, crf->lat[1].bary,
                       crf->lat[1].norm, crf->N, crf->lat[0].rep > 1 ? 1 : 0, crf->pix_rec);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

// the 52-byte bilateral records (pack_pixels_b_kernel) of the GF updates; same stream rules as crf_full_records
int crf_bilateral_records(wsc_crf *crf, hipStream_t st) {
    if (crf->pix_rec_b) return WSC_OK;
    wsc_ctx *bctx = crf->ctx;
    WSC_TRY(crf_alloc(crf, sizeof(uint32_t) * 13 * (size_t)crf->B * crf->N, (void **)&crf->pix_rec_b));
    if (st != bctx->stream) {
        WSC_HIP(hipEventRecord(bctx->join_ev, bctx->stream));
        WSC_HIP(hipStreamWaitEvent(st, bctx->join_ev, 0));
    }
    hipLaunchKernelGGL(pack_pixels_b_kernel, dim3((unsigned)grid1d((long long)crf->B * crf->N * 13, 256, 8192)), dim3(256), 0, st,
                       crf->lat[1].offset, crf->lat[1].bary, crf->lat[1].norm, (long long)crf->B * crf->N, crf->pix_rec_b);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

} // namespace

extern "C" {

int wsc_crf_create(wsc_ctx *ctx, const uint8_t *rgb_dev, int B, int H, int W, float g_sxy, float bi_sxy,
                   float bi_srgb, wsc_crf **out) {
    WSC_CHECK(ctx && rgb_dev && out, WSC_ERR_INVALID, "wsc_crf_create: null argument");
    WSC_CHECK(B > 0 && H > 0 && W > 0, WSC_ERR_INVALID, "wsc_crf_create: B=%d H=%d W=%d", B, H, W);
    WSC_CHECK(g_sxy > 0.f && bi_sxy > 0.f && bi_srgb > 0.f, WSC_ERR_INVALID,
              "wsc_crf_create: kernel widths must be positive");
    WSC_HIP(hipSetDevice(ctx->device));
    wsc_crf *crf = new wsc_crf();
    crf->ctx = ctx; crf->B = B; crf->H = H; crf->W = W; crf->N = H * W;
    WscKernelTimer timer(ctx, WSC_K_CRF_BUILD, (double)B * H * W * (3.0 * 16 + 6.0 * 16));
    // The Gaussian lattice is a function of (H, W, sxy) alone: one copy per ctx serves every batch of
    // that image size (the reference rebuilds it per image, addPairwiseGaussian in dcrf_process /
    // crf_inference_label, and gets the same table every time).
    int st = WSC_OK;
    GaussCache *hit = nullptr, *lru = nullptr;
    int n_cached = 0;
    size_t lru_at = 0;
    for (size_t i = 0; i < ctx->attachments.size(); ++i) {
        auto &a = ctx->attachments[i];
        if (a.second == &gauss_cache_delete) {
            ++n_cached;
            GaussCache *g = static_cast<GaussCache *>(a.first);
            if (g->H == H && g->W == W && g->sxy == g_sxy) hit = g;
            if (g->users == 0 && (!lru || g->last_use < lru->last_use)) {
                lru = g;
                lru_at = i;
            }
        }
    }
    if (!hit && n_cached >= GAUSS_CACHE_MAX && lru) {
        // evict: its blocks go back to the build ctx's stream-ordered cache.  No live wsc_crf refers to it, and every
        // wsc_crf that did has been destroyed -- which made this stream wait for loops it ran on other streams.
        for (void *p : lru->blocks) wsc_ctx_cached_free(ctx, p);
        delete lru;
        ctx->attachments.erase(ctx->attachments.begin() + (long)lru_at);
        --n_cached;
    }
    if (hit) {
        if (hit->fuse_pending) {
            // second use of this image size: now the host-built tile vertex sets pay (a size seen once -- cam_to_ir_label walks
            // hundreds of native VOC sizes -- never pays the 3-4 ms; its loop takes the blur-kernel path, same bits)
            hit->fuse_pending = false;
            crf->persist = true;
            const int st_f = gauss_fuse_tables(crf, hit->L, make_geom(H, W));
            hit->blocks.insert(hit->blocks.end(), crf->persist_allocs.begin(), crf->persist_allocs.end());
            crf->persist_allocs.clear();
            crf->persist = false;
            if (st_f != WSC_OK) { // (the lattice itself stays valid without the tables)
                hit->L.gt_cnt = nullptr; hit->L.gt_rows = nullptr; hit->L.gt_nbr = nullptr; hit->L.gt_pix = nullptr;
                hit->L.gt_stride = 0;
            }
        }
        crf->lat[0] = hit->L;
        hit->users += 1;
        hit->last_use = ++g_gauss_use_clock;
        crf->gauss_entry = hit;
    } else {
        crf->persist = n_cached < GAUSS_CACHE_MAX;
        st = build_lattice<2>(crf, crf->lat[0], rgb_dev, g_sxy, 1.f, true, true); // built once: worst-case table
        if (st == WSC_OK && crf->persist) {
            GaussCache *g = new GaussCache{H, W, g_sxy, crf->lat[0], crf->persist_allocs, 1, ++g_gauss_use_clock};
            ctx->attachments.emplace_back(g, &gauss_cache_delete);
            crf->gauss_entry = g;
        }
        if (st != WSC_OK) // a failed build must not leave its arrays with the ctx for the rest of its life
            for (void *p : crf->persist_allocs) wsc_ctx_cached_free(ctx, p);
        crf->persist_allocs.clear();
        crf->persist = false;
    }
    crf->lat[0].rep = B;
    crf->lat[0].v_per_image.assign(B, crf->lat[0].rows - 1);
    if (st == WSC_OK) {
        const size_t mark = crf->allocs.size();
        st = build_lattice<5>(crf, crf->lat[1], rgb_dev, bi_sxy, bi_srgb, false, false);
        if (st == WSC_RETRY_FULL_TABLE) { // noise-like image: more vertices than the right-sized table holds
            for (size_t i = mark; i < crf->allocs.size(); ++i) wsc_ctx_cached_free(ctx, crf->allocs[i]);
            crf->allocs.resize(mark);
            crf->pix_rec_b = nullptr; // (allocated by the failed attempt: handed back with its other blocks)
            crf->lat[1] = LatticeDev();
            st = build_lattice<5>(crf, crf->lat[1], rgb_dev, bi_sxy, bi_srgb, false, true);
        }
    }
    // The 80-byte record (both lattices) serves the updates that gather the Gaussian rows themselves.  When the Gaussian
    // message can be formed on chip for every class count (the tile vertex sets fit at LP = 8), it is not built here: a
    // call that still wants it (WSC_CRF_NO_GFUSE=1) builds it on first use (crf_full_records).
    if (st == WSC_OK && !(crf->lat[0].gt_rows && update_gf_ok(ctx, crf->lat[0], 8))) st = crf_full_records(crf, ctx->stream);
    if (st == WSC_OK && crf->lat[0].gt_rows) st = crf_bilateral_records(crf, ctx->stream);
    if (st != WSC_OK) {
        wsc_crf_destroy(crf);
        return st;
    }
    *out = crf;
    return WSC_OK;
}

void wsc_crf_destroy(wsc_crf *crf) {
    if (!crf) return;
    // The blocks go back to the build ctx's stream-ordered cache.  A mean-field loop enqueued on ANOTHER ctx may still
    // be reading them: make the build stream wait for it before anything it launches later can reuse the memory.
    if (crf->used_elsewhere && crf->use_ev) (void)hipStreamWaitEvent(crf->ctx->stream, crf->use_ev, 0);
    if (crf->gauss_entry) static_cast<GaussCache *>(crf->gauss_entry)->users -= 1;
    for (void *p : crf->allocs) wsc_ctx_cached_free(crf->ctx, p); // reused in stream order
    if (crf->use_ev) (void)hipEventDestroy(crf->use_ev);
    delete crf;
}

int wsc_crf_lattice_sizes(wsc_ctx *ctx, const wsc_crf *crf, int32_t *v_gauss_host, int32_t *v_bilat_host) {
    WSC_CHECK(ctx && crf, WSC_ERR_INVALID, "wsc_crf_lattice_sizes: null argument");
    for (int b = 0; b < crf->B; ++b) {
        if (v_gauss_host) v_gauss_host[b] = crf->lat[0].v_per_image[b];
        if (v_bilat_host) v_bilat_host[b] = crf->lat[1].v_per_image[b];
    }
    return WSC_OK;
}

int wsc_crf_gaussian_on_chip(const wsc_crf *crf, int M) {
    if (!crf || M < 1 || M > 32) return 0;
    return update_gf_ok(crf->ctx, crf->lat[0], (M + 3) / 4) ? 1 : 0;
}

static int crf_inference_impl(wsc_ctx *ctx, wsc_crf *crf, const float *unary_dev, bool pixel_major, int M, float g_compat,
                              float bi_compat, int n_iters, float *q_dev, int32_t *argmax_dev);

int wsc_crf_inference(wsc_ctx *ctx, wsc_crf *crf, const float *unary_dev, int M, float g_compat, float bi_compat,
                      int n_iters, float *q_dev, int32_t *argmax_dev) {
    return crf_inference_impl(ctx, crf, unary_dev, false, M, g_compat, bi_compat, n_iters, q_dev, argmax_dev);
}

int wsc_crf_inference_pm(wsc_ctx *ctx, wsc_crf *crf, const float *unary_pm_dev, int M, float g_compat, float bi_compat,
                         int n_iters, float *q_dev, int32_t *argmax_dev) {
    return crf_inference_impl(ctx, crf, unary_pm_dev, true, M, g_compat, bi_compat, n_iters, q_dev, argmax_dev);
}

static int crf_inference_impl(wsc_ctx *ctx, wsc_crf *crf, const float *unary_dev, bool pixel_major, int M, float g_compat,
                              float bi_compat, int n_iters, float *q_dev, int32_t *argmax_dev) {
    WSC_CHECK(ctx && crf && unary_dev, WSC_ERR_INVALID, "wsc_crf_inference: null argument");
    WSC_CHECK(M >= 1 && M <= 32, WSC_ERR_INVALID, "wsc_crf_inference: M=%d outside [1,32]", M);
    WSC_CHECK(n_iters >= 0, WSC_ERR_INVALID, "wsc_crf_inference: n_iters=%d", n_iters);
    WSC_HIP(hipSetDevice(ctx->device));
    const int B = crf->B, N = crf->N;
    const long long npix = (long long)B * N;
    const LatticeDev &G = crf->lat[0], &Bl = crf->lat[1];
    const int LP = (M + 3) / 4, Mp = 4 * LP; // rows padded to 16-byte multiples
    const long long g_rows = (long long)G.rows * G.rep, g_slots = (long long)G.n_slots * G.rep;
    // the iteration kernels address with 32-bit BYTE offsets: every array below 4 GiB
    WSC_CHECK(npix * Mp < (1ll << 30) && g_rows * Mp < (1ll << 30) && (long long)Bl.rows * Mp < (1ll << 30) &&
                  g_slots * Mp < (1ll << 30) && (long long)Bl.n_slots * Mp < (1ll << 30) && npix * 80 < (1ll << 32),
              WSC_ERR_CAPACITY, "CRF batch too large for 32-bit byte offsets (B*N*Mp = %lld elements; split the batch)", npix * Mp);
    // ... and forms them with 24-bit multiplies: pixels of one image and row ids below 2^24
    WSC_CHECK(N < (1 << 24) && G.rows < (1 << 24) && Bl.rows < (1 << 24), WSC_ERR_CAPACITY,
              "CRF image / lattice too large for 24-bit row arithmetic (N = %d, rows = %d / %d)", N, G.rows, Bl.rows);
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t qb = al(sizeof(float) * npix * Mp);
    const size_t pg = al(sizeof(float) * (size_t)g_slots * Mp), pb = al(sizeof(float) * (size_t)Bl.n_slots * Mp);
    const size_t vg = al(sizeof(float) * (size_t)g_rows * Mp), vb = al(sizeof(float) * (size_t)Bl.rows * Mp);
    void *ws;
    WSC_TRY(wsc_ctx_workspace(ctx, 2 * qb + 2 * vg + 2 * vb + 2 * pg + pb, &ws));
    char *p = (char *)ws;
    float *u = (float *)p; p += qb;
    float *q = (float *)p; p += qb;
    float *vg0 = (float *)p; p += vg;
    float *vg1 = (float *)p; p += vg;
    float *vb0 = (float *)p; p += vb;
    float *vb1 = (float *)p; p += vb;
    float *partg = (float *)p; p += pg;
    float *partg2 = (float *)p; p += pg; // FG updates: the splat writes one array while the blocks still read the other
    float *partb = (float *)p; p += pb;

    crf->lat[0].M_cur = M;
    crf->lat[1].M_cur = M;
    if (n_iters > 0) WSC_TRY(blur_lds_prepare(ctx, crf->lat[1], M));
    const dim3 tgrid((N + TP - 1) / TP, B);
    if (!pixel_major) {
        WscKernelTimer timer(ctx, WSC_K_CRF_MISC, (double)npix * M * 12);
        hipLaunchKernelGGL(init_q_kernel, tgrid, dim3(TP), 2 * (size_t)M * (TP + 1) * sizeof(float), ctx->stream,
                           unary_dev, M, Mp, N, u, n_iters == 0 ? q : (float *)nullptr);
    } else {
        u = const_cast<float *>(unary_dev); // already [pixel][Mp] (wsc_cam_unary_pm): read in place, never written
    }
    // The side stream pays off only for the per-pass bilateral launches (seven short kernels beside the Gaussian message);
    // with the on-chip blur (two launches whose workgroups fill the CUs' LDS) one stream and two are the same to +-0.5 %
    // (profiles/README.md), so that path stays on one stream.  WSC_CRF_NO_FORK=1 / 0 forces either.
#ifdef WSC_AB_KNOBS
    const char *nf = getenv("WSC_CRF_NO_FORK");
#else
    const char *nf = nullptr;
#endif
    const bool lds_blur = Bl.bl_ok && Bl.img_row && ctx->opt[WSC_OPT_CRF_BLUR_ON_CHIP];
    const bool no_fork = (nf ? atoi(nf) != 0 : lds_blur) || ctx->profiling; // per-kernel timing wants the launches one after the other
    if (!no_fork && !ctx->aux_stream) {
        // (default priority: a high-priority side stream, or any other priority split between the stages of a pipelined
        // caller, was measured and lost 3-20 %: profiles/README.md)
        WSC_HIP(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
        WSC_HIP(hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming));
        WSC_HIP(hipEventCreateWithFlags(&ctx->aux_done_ev, hipEventDisableTiming));
    }
    UpdateArgs a;
    a.pix_rec = crf->pix_rec; a.pix_rec_b = crf->pix_rec_b; a.val_g = nullptr; a.val_b = nullptr;
    a.u = u; a.q = nullptr; a.argmax = nullptr;
    a.alpha_g = G.alpha; a.alpha_b = Bl.alpha; a.compat_g = g_compat; a.compat_b = bi_compat;
    a.M = M; a.LP = LP; a.B = B;
    a.tg = make_geom(crf->H, crf->W);
    a.g_pix = (unsigned)G.n_pix; a.g_rows = G.rep > 1 ? (unsigned)G.rows : 0u;
    a.sg.tslot_start = G.tslot_start; a.sg.slot_desc = G.slot_desc; a.sg.tent_w = G.tent_w; a.sg.tent_p = G.tent_p; a.sg.part = partg;
    a.sg.n_slots = G.n_slots; a.sg.shared = G.rep > 1 ? 1 : 0; a.sg.dp1 = 3;
    a.sb.tslot_start = Bl.tslot_start; a.sb.slot_desc = Bl.slot_desc; a.sb.tent_w = Bl.tent_w; a.sb.tent_p = Bl.tent_p; a.sb.part = partb;
    a.sb.n_slots = Bl.n_slots; a.sb.shared = 0; a.sb.dp1 = 6;
    // Gaussian message on chip (tile vertex sets that fit the LDS): no Gaussian value rows, no Gaussian blur launch --
    // gauss_msg_kernel turns the Gaussian slot partials into E = -U + message (in the Q buffer: an update reads its slot of
    // E before it writes Q there) beside the bilateral lattice's combine + six passes, and the update starts from E
    const bool gf = update_gf_ok(ctx, G, LP);
    const bool fg = gf && update_fg_ok(ctx, G, LP);
    a.gt_cnt = G.gt_cnt; a.gt_rows = G.gt_rows; a.gt_nbr = G.gt_nbr; a.gt_pix = G.gt_pix; a.gt_stride = G.gt_stride;
    a.part_g_in = nullptr;
    a.tl = nullptr;
#ifdef WSC_AB_KNOBS
    if (const char *te = getenv("WSC_CRF_UPD_TIMELINE")) { // A/B: file the stamps of the LAST splatting update of this call go to
        static unsigned long long *tl_dev = nullptr;
        static size_t tl_cap = 0;
        const size_t need = (size_t)B * a.tg.tpi * 8;
        if (need > tl_cap) {
            if (tl_dev) (void)hipFree(tl_dev);
            WSC_HIP(hipMalloc((void **)&tl_dev, need * sizeof(unsigned long long)));
            tl_cap = need;
        }
        WSC_HIP(hipMemsetAsync(tl_dev, 0, need * sizeof(unsigned long long), ctx->stream));
        a.tl = tl_dev;
        (void)te;
    }
    unsigned long long *const tl_keep = a.tl;
#endif
    if (gf && n_iters > 0) { // (built by wsc_crf_create whenever the lattice has its tile vertex sets; here for completeness)
        WSC_TRY(crf_bilateral_records(crf, ctx->stream));
        a.pix_rec_b = crf->pix_rec_b;
    }
    if (!gf && n_iters > 0) {
        WSC_TRY(crf_full_records(crf, ctx->stream));
        a.pix_rec = crf->pix_rec;
    }
    GaussMsgArgs gm;
    gm.gt_cnt = G.gt_cnt; gm.gt_rows = G.gt_rows; gm.gt_nbr = G.gt_nbr; gm.gt_pix = G.gt_pix; gm.part = partg; gm.u = u; gm.e = q;
    gm.gt_stride = G.gt_stride; gm.n_slots = G.n_slots; gm.shared = G.rep > 1 ? 1 : 0; gm.LP = LP; gm.B = B;
    gm.cag = g_compat * G.alpha; gm.tg = a.tg;
    // SURVEY 8d terms of the Gaussian kernel this launch covers: the blur (2 (d+1) V M 4) and the slice (index + weight of the
    // 3 vertices per pixel, the message N M 4)
    const double gf_bytes = 2.0 * 3.0 * (double)(G.rows - 1) * G.rep * M * 4 + (double)npix * (3 * 8 + 1.0 * M * 4);
    // Q(0) = softmax(-U) is splatted straight from the kernel that computes it; iteration t slices the blurred
    // lattices, forms Q(t) and splats it for iteration t+1; the last iteration writes Q(T) instead.
    if (n_iters == 0 && pixel_major) { // Q = softmax(-U): the update kernel without messages and without the splat
        a.q = q;
        WSC_TRY(launch_update(ctx, a, false, false, false));
    }
    for (int it = 0; it <= n_iters && n_iters > 0; ++it) {
        const bool last = it == n_iters;
        const bool labels_only = last && q_dev == nullptr && argmax_dev != nullptr;
        a.q = last && !labels_only ? q : nullptr;
        a.argmax = labels_only ? argmax_dev : nullptr;
        a.u = (gf && !fg && it > 0) ? q : u;
#ifdef WSC_AB_KNOBS
        if (last || it == 0) a.tl = nullptr; // (the stamps are those of the last SPLATTING update with messages)
        else a.tl = tl_keep;
#endif
        if (fg) { // iteration `it` reads the Gaussian partials of splat it - 1 and writes those of splat `it`
            a.part_g_in = (it & 1) ? partg : partg2;
            a.sg.part = (it & 1) ? partg2 : partg;
        }
        WSC_TRY(launch_update(ctx, a, it > 0, !last, gf, fg && it > 0, gf_bytes));
        if (last) break;
        // The two lattices are independent until the next update: the bilateral one (seven short launches on ~1 MB per
        // image) runs on the ctx's side stream beside the Gaussian lattice's fused blur.
        const bool fork = !no_fork;
        hipStream_t main_stream = ctx->stream;
        if (fork) {
            WSC_HIP(hipEventRecord(ctx->fork_ev, main_stream));
            WSC_HIP(hipStreamWaitEvent(ctx->aux_stream, ctx->fork_ev, 0));
        }
        a.val_b = combine_blur_all4(ctx, fork ? ctx->aux_stream : main_stream, Bl, LP, partb, vb0, vb1);
        if (fork) WSC_HIP(hipEventRecord(ctx->aux_done_ev, ctx->aux_stream));
        if (fg) {
        } else if (gf) WSC_TRY(launch_gauss_msg(ctx, main_stream, gm, gf_bytes));
        else a.val_g = combine_blur_all4(ctx, main_stream, G, LP, partg, vg0, vg1);
        if (fork) WSC_HIP(hipStreamWaitEvent(main_stream, ctx->aux_done_ev, 0));
    }
    {
        WscKernelTimer ftimer(ctx, WSC_K_CRF_MISC, (double)npix * M * 8);
        if (q_dev || (argmax_dev && n_iters == 0))
            hipLaunchKernelGGL(finish_kernel, tgrid, dim3(TP), (size_t)M * (TP + 1) * sizeof(float), ctx->stream, q, M, Mp,
                               N, q_dev, argmax_dev);
    }
    WSC_HIP(hipGetLastError());
#ifdef WSC_AB_KNOBS
    if (tl_keep != nullptr) {
        const size_t need = (size_t)B * a.tg.tpi * 8;
        std::vector<unsigned long long> h(need);
        WSC_HIP(hipMemcpyAsync(h.data(), tl_keep, need * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        WSC_HIP(hipStreamSynchronize(ctx->stream));
        if (FILE *f = fopen(getenv("WSC_CRF_UPD_TIMELINE"), "wb")) {
            fwrite(h.data(), sizeof(unsigned long long), need, f);
            fclose(f);
        }
    }
#endif
    if (ctx != crf->ctx) { // destroy must not hand the lattices back to the build ctx's cache before this loop is done
        if (!crf->use_ev) WSC_HIP(hipEventCreateWithFlags(&crf->use_ev, hipEventDisableTiming));
        WSC_HIP(hipEventRecord(crf->use_ev, ctx->stream));
        crf->used_elsewhere = true;
    }
    return WSC_OK;
}


// ---- ragged batch: per-image (H, W, M) in one object ---------------------------------------------------------------------
// 03b_irn/step/cam_to_ir_label.py:25-58 and 03c_hsn/utilities.py:420-445 run the CRF one image at a time, every image with
// its own size and its own class count.  A wsc_crf_v takes such a list as it comes: images of equal (H, W) form a group
// (one lattice build, one mean-field loop per group, issued back to back on the ctx's stream with no host synchronisation
// between them), and inside a group every image keeps its own class count -- the loop runs at the group's largest M with
// the missing classes of the smaller images at probability zero, which changes no bit of their results (see
// gather_unary_v_kernel).
struct CrfVGroup {
    wsc_crf *crf = nullptr;
    std::vector<int> members; // image indices, in input order
    uint8_t *rgb = nullptr;   // [members][H][W][3] (cached block of the build ctx), or null when the inputs were contiguous
};
} // extern "C"
struct wsc_crf_v {
    wsc_ctx *ctx = nullptr;
    int B = 0;
    std::vector<int> H, W;
    std::vector<CrfVGroup> groups;
};
extern "C" {

void wsc_crf_v_destroy(wsc_crf_v *cv) {
    if (!cv) return;
    for (CrfVGroup &g : cv->groups) {
        if (g.crf) wsc_crf_destroy(g.crf);
        if (g.rgb) wsc_ctx_cached_free(cv->ctx, g.rgb);
    }
    delete cv;
}

int wsc_crf_v_create(wsc_ctx *ctx, const uint8_t *const *rgb_dev, const int32_t *hw_host, int B, float g_sxy, float bi_sxy,
                     float bi_srgb, wsc_crf_v **out) {
    WSC_CHECK(ctx && rgb_dev && hw_host && out, WSC_ERR_INVALID, "wsc_crf_v_create: null argument");
    WSC_CHECK(B > 0, WSC_ERR_INVALID, "wsc_crf_v_create: B=%d", B);
    WSC_HIP(hipSetDevice(ctx->device));
    wsc_crf_v *cv = new wsc_crf_v();
    cv->ctx = ctx; cv->B = B;
    for (int b = 0; b < B; ++b) {
        const int H = hw_host[2 * b], W = hw_host[2 * b + 1];
        if (!(H > 0 && W > 0 && rgb_dev[b])) {
            wsc_crf_v_destroy(cv);
            wsc_set_error("wsc_crf_v_create: image %d: %dx%d, rgb %p", b, H, W, (const void *)rgb_dev[b]);
            return WSC_ERR_INVALID;
        }
        cv->H.push_back(H); cv->W.push_back(W);
        size_t gi = 0;
        for (; gi < cv->groups.size(); ++gi)
            if (cv->H[cv->groups[gi].members[0]] == H && cv->W[cv->groups[gi].members[0]] == W) break;
        if (gi == cv->groups.size()) cv->groups.emplace_back();
        cv->groups[gi].members.push_back(b);
    }
    for (CrfVGroup &g : cv->groups) {
        const int H = cv->H[g.members[0]], W = cv->W[g.members[0]];
        const size_t ib = (size_t)H * W * 3;
        const int nb = (int)g.members.size();
        bool contiguous = true;
        for (int i = 1; i < nb; ++i) contiguous = contiguous && rgb_dev[g.members[i]] == rgb_dev[g.members[0]] + (size_t)i * ib;
        const uint8_t *src = rgb_dev[g.members[0]];
        int st = WSC_OK;
        if (!contiguous) {
            st = wsc_ctx_cached_alloc(ctx, ib * nb, (void **)&g.rgb);
            for (int i = 0; i < nb && st == WSC_OK; ++i)
                if (hipMemcpyAsync(g.rgb + (size_t)i * ib, rgb_dev[g.members[i]], ib, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
                    wsc_set_error("wsc_crf_v_create: device copy of image %d failed", g.members[i]);
                    st = WSC_ERR_HIP;
                }
            src = g.rgb;
        }
        if (st == WSC_OK) st = wsc_crf_create(ctx, src, nb, H, W, g_sxy, bi_sxy, bi_srgb, &g.crf);
        if (st != WSC_OK) {
            wsc_crf_v_destroy(cv);
            return st;
        }
    }
    *out = cv;
    return WSC_OK;
}

int wsc_crf_v_num_groups(const wsc_crf_v *cv) { return cv ? (int)cv->groups.size() : 0; }

int wsc_crf_v_inference(wsc_ctx *ctx, wsc_crf_v *cv, const float *const *unary_dev, const int32_t *M_host, float g_compat,
                        float bi_compat, int n_iters, float *const *q_dev, int32_t *const *argmax_dev) {
    WSC_CHECK(ctx && cv && unary_dev && M_host, WSC_ERR_INVALID, "wsc_crf_v_inference: null argument");
    WSC_CHECK(q_dev || argmax_dev, WSC_ERR_INVALID, "wsc_crf_v_inference: neither q_dev nor argmax_dev");
    WSC_HIP(hipSetDevice(ctx->device));
    for (int b = 0; b < cv->B; ++b)
        WSC_CHECK(M_host[b] >= 1 && M_host[b] <= 32 && unary_dev[b], WSC_ERR_INVALID, "wsc_crf_v_inference: image %d: M=%d, unary %p", b,
                  M_host[b], (const void *)unary_dev[b]);
    for (CrfVGroup &g : cv->groups) {
        const int nb = (int)g.members.size();
        const int N = cv->H[g.members[0]] * cv->W[g.members[0]];
        int Mg = 0;
        for (int i : g.members) Mg = std::max(Mg, (int)M_host[i]);
        const int Mp = (Mg + 3) / 4 * 4;
        // per group: job table, pixel-major unaries, and the group's outputs (class-major Q / labels) before they are dealt out
        const bool want_q = q_dev != nullptr;
        const size_t jb = ((size_t)nb * sizeof(UnaryVJob) + 255) / 256 * 256, ub = ((size_t)nb * N * Mp * 4 + 255) / 256 * 256;
        const size_t qb = want_q ? ((size_t)nb * N * Mg * 4 + 255) / 256 * 256 : 0, ab = argmax_dev ? ((size_t)nb * N * 4 + 255) / 256 * 256 : 0;
        char *blk = nullptr;
        WSC_TRY(wsc_ctx_cached_alloc(ctx, jb + ub + qb + ab, (void **)&blk));
        std::vector<UnaryVJob> jobs(nb);
        for (int i = 0; i < nb; ++i) jobs[i] = UnaryVJob{unary_dev[g.members[i]], (int)M_host[g.members[i]]};
        int st = wsc_ctx_upload_small(ctx, blk, jobs.data(), jobs.size() * sizeof(UnaryVJob));
        float *u_pm = (float *)(blk + jb), *q_g = want_q ? (float *)(blk + jb + ub) : nullptr;
        int32_t *a_g = argmax_dev ? (int32_t *)(blk + jb + ub + qb) : nullptr;
        if (st == WSC_OK) {
            WscKernelTimer timer(ctx, WSC_K_CRF_MISC, (double)nb * N * (Mg + Mp) * 4);
            hipLaunchKernelGGL(gather_unary_v_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)nb), dim3(256), 0, ctx->stream,
                               (const UnaryVJob *)blk, Mg, Mp, N, u_pm);
            if (hipGetLastError() != hipSuccess) st = WSC_ERR_HIP;
        }
        if (st == WSC_OK) st = crf_inference_impl(ctx, g.crf, u_pm, true, Mg, g_compat, bi_compat, n_iters, q_g, a_g);
        for (int i = 0; i < nb && st == WSC_OK; ++i) {
            const int b = g.members[i];
            if (want_q && q_dev[b] &&
                hipMemcpyAsync(q_dev[b], q_g + (size_t)i * Mg * N, (size_t)M_host[b] * N * 4, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
                st = WSC_ERR_HIP;
            if (argmax_dev && argmax_dev[b] &&
                hipMemcpyAsync(argmax_dev[b], a_g + (size_t)i * N, (size_t)N * 4, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
                st = WSC_ERR_HIP;
        }
        wsc_ctx_cached_free(ctx, blk); // stream-ordered: reused only by work enqueued later on this ctx
        if (st == WSC_ERR_HIP) wsc_set_error("wsc_crf_v_inference: a launch or device copy failed");
        if (st != WSC_OK) return st;
    }
    return WSC_OK;
}

} // extern "C"
