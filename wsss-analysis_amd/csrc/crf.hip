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

#include <cmath>
#include <cstring>

namespace {

constexpr unsigned long long EMPTY_KEY = 0xFFFFFFFFFFFFFFFFull;

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
    int32_t *csr_start = nullptr; // [rows + 1]
    int32_t *csr_pix = nullptr;   // [B*N*(d+1)]  global pixel index
    float *csr_w = nullptr;       // [B*N*(d+1)]
    uint2 *csr_ent = nullptr;     // [B*N*(d+1)] {pixel, bits(w * norm[pixel])}: one 8-byte load per gathered pixel
    int2 *nbr = nullptr;          // [(d+1)][rows]
    // splat work items: every row is cut into chunks of <= SPLAT_CHUNK gathered pixels (64: 32 -> 294.5 us, 64 -> 288.6, 128 -> 322.9 per splat; with the final kernels 8 / 16 / 32 / 64 -> 473 / 358 / 299 / 269 us)
    int32_t *chunk_base = nullptr; // [rows + 1] first chunk of each row
    int32_t *chunk_row = nullptr;  // [n_chunks] owning row
    int4 *chunk_desc = nullptr;    // [n_chunks] {first entry, entry count, row, 1 if the row's only chunk}
    int32_t *long_rows = nullptr;  // [n_long] rows with more than one chunk
    int n_chunks = 0, n_long = 0;
    long long n_pix = 0; // pixels of one replica (B*N when rep == 1)
    int M_cur = 0;       // class count of the inference in flight (algorithmic byte accounting)
    float alpha = 0.f;
    // Gaussian lattice only (d = 2): tiles of the dense (i, j) index space for the fused three-pass blur
    int32_t *tile_rows = nullptr; // [n_tiles][GBI * GBJ] row id of every point of the tile's halo box (0 = absent)
    int32_t *tile_list = nullptr; // [n_tiles_occ] tiles with at least one interior vertex
    int n_tiles_occ = 0;
    std::vector<int32_t> v_per_image;
};

} // namespace

struct wsc_crf {
    wsc_ctx *ctx = nullptr;
    int B = 0, H = 0, W = 0, N = 0;
    LatticeDev lat[2]; // 0: Gaussian (d=2), 1: bilateral (d=5)
    // per pixel, 20 dwords = five 16-byte loads: offG[3] offB[6] baryG[3] baryB[6] normG normB
    uint4 *pix_rec = nullptr;
    std::vector<void *> allocs;
    bool persist = false; // allocations made while set belong to the ctx (cached Gaussian lattice)
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
    int32_t *eslot;  // [B*N*(d+1)] table slot of each entry
    float *bary;     // [B*N*(d+1)]
    int32_t *first;  // [B*cap] smallest entry index touching the slot
    int *err;        // key range error flag
};

// Step 1 of the lattice build: elevate every pixel's feature vector, find its simplex and
// barycentric weights (exactly as Permutohedral::init does), insert the d+1 vertex keys
// into the image's hash table.
template <int D>
__global__ __launch_bounds__(256) void lattice_embed_kernel(EmbedArgs a) {
    const long long gp = (long long)blockIdx.x * blockDim.x + threadIdx.x; // global pixel
    if (gp >= (long long)a.B * a.N) return;
    const int b = (int)(gp / a.N);
    const int n = (int)(gp - (long long)b * a.N);
    const int y = n / a.W, x = n - y * a.W;

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

    unsigned long long *table = a.table + (long long)b * a.cap;
    int32_t *first = a.first + (long long)b * a.cap;
    const long long e0 = gp * (D + 1);
#pragma unroll
    for (int r = 0; r <= D; ++r) {
        int key[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            // canonical[r][rank[i]] = rank[i] <= D - r ? r : r - (D+1)
            const int can = rank[i] <= D - r ? r : r - (D + 1);
            key[i] = (int)(short)(rem0[i] + (float)can);
        }
        unsigned long long pk;
        if (!pack_key<D>(key, pk)) atomicOr(a.err, 1);
        int slot;
        if (true) {
            // lanes = consecutive pixels: the group leader (lowest lane = lowest pixel) inserts for all
            const unsigned long long grp = wave_match(pk, b);
            const int leader = __ffsll((long long)grp) - 1;
            slot = 0;
            if (lane_id() == leader) {
                slot = hash_insert(table, a.cap_mask, pk);
                if (slot >= 0) atomicMin(&first[slot], n * (D + 1) + r);
                else atomicOr(a.err, 2); // table too small
            }
            slot = __shfl(slot, leader, 64);
        } else {
            slot = hash_insert(table, a.cap_mask, pk);
            if (slot >= 0) atomicMin(&first[slot], n * (D + 1) + r);
            else atomicOr(a.err, 2);
        }
        if (slot < 0) slot = 0; // keeps the bookkeeping kernels in range; the build is discarded
        a.eslot[e0 + r] = slot;
        a.bary[e0 + r] = bary[r];
    }
}

// flag[e] = 1 if entry e is the first (raster order) toucher of its vertex
__global__ void flag_first_kernel(const int32_t *__restrict__ eslot, const int32_t *__restrict__ first, long long cap,
                                  int N, int dp1, long long total, unsigned *__restrict__ flag) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long per_img = (long long)N * dp1;
        const int b = (int)(e / per_img);
        const int local = (int)(e - b * per_img);
        flag[e] = first[(long long)b * cap + eslot[e]] == local ? 1u : 0u;
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

int exclusive_scan(wsc_ctx *ctx, const unsigned *in, long long n, unsigned *out, unsigned *sums /* nblk+1 */) {
    const int nblk = (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nblk), dim3(256), 0, ctx->stream, in, n, sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, ctx->stream, sums, nblk);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nblk), dim3(256), 0, ctx->stream, in, n, sums, out);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

// first-toucher entries publish the row id of their slot and the slot's key
__global__ void assign_ids_kernel(const int32_t *__restrict__ eslot, const unsigned *__restrict__ flag,
                                  const unsigned *__restrict__ prefix, const unsigned long long *__restrict__ table,
                                  long long cap, int N, int dp1, long long total, int32_t *__restrict__ slot2row,
                                  unsigned long long *__restrict__ rowkey, int32_t *__restrict__ rowimg) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        if (!flag[e]) continue;
        const int b = (int)(e / ((long long)N * dp1));
        const long long s = (long long)b * cap + eslot[e];
        const int row = 1 + (int)prefix[e];
        slot2row[s] = row;
        rowkey[row] = table[s];
        rowimg[row] = b;
    }
}

// offset[e] = row of entry e; count entries per row.  One thread per pixel, r in the loop, so the
// lanes of a wave are neighbouring pixels and share rows: one atomicAdd per distinct row per wave.
__global__ __launch_bounds__(256) void remap_count_kernel(const int32_t *__restrict__ eslot,
                                                          const int32_t *__restrict__ slot2row, long long cap, int N,
                                                          int dp1, long long npix, int32_t *__restrict__ offset,
                                                          unsigned *__restrict__ count) {
    const long long gp = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gp >= npix) return;
    const int b = (int)(gp / N);
    for (int r = 0; r < dp1; ++r) {
        const long long e = gp * dp1 + r;
        const int row = slot2row[(long long)b * cap + eslot[e]];
        offset[e] = row;
        const unsigned long long grp = wave_match32((unsigned)row);
        if (lane_id() == __ffsll((long long)grp) - 1) atomicAdd(&count[row], (unsigned)__popcll(grp));
    }
}

__global__ __launch_bounds__(256) void csr_fill_kernel(const int32_t *__restrict__ offset,
                                                       const float *__restrict__ bary, int dp1, long long npix,
                                                       const unsigned *__restrict__ start,
                                                       unsigned *__restrict__ cursor, int32_t *__restrict__ csr_pix,
                                                       float *__restrict__ csr_w) {
    const long long gp = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gp >= npix) return;
    const int lane = lane_id();
    for (int r = 0; r < dp1; ++r) {
        const long long e = gp * dp1 + r;
        const int row = offset[e];
        const unsigned long long grp = wave_match32((unsigned)row);
        const int leader = __ffsll((long long)grp) - 1;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(&cursor[row], (unsigned)__popcll(grp));
        base = __shfl(base, leader, 64);
        const unsigned pos = start[row] + base + (unsigned)__popcll(grp & ((1ull << lane) - 1ull));
        csr_pix[pos] = (int32_t)gp;
        csr_w[pos] = bary[e];
    }
}

// blur neighbours of every row along every axis (Permutohedral::init, second half)
template <int D>
__global__ void neighbors_kernel(const unsigned long long *__restrict__ rowkey, const int32_t *__restrict__ rowimg,
                                 const unsigned long long *__restrict__ table, const int32_t *__restrict__ slot2row,
                                 long long cap, unsigned cap_mask, int rows, int2 *__restrict__ nbr) {
    const long long total = (long long)rows * (D + 1);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i / rows);
        const int row = (int)(i - (long long)j * rows);
        if (row == 0) {
            nbr[i] = make_int2(0, 0);
            continue;
        }
        int key[D], k1[D], k2[D];
        unpack_key<D>(rowkey[row], key);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            k1[k] = key[k] - 1;
            k2[k] = key[k] + 1;
        }
        if (j < D) {
            // static indexing
#pragma unroll
            for (int k = 0; k < D; ++k)
                if (k == j) {
                    k1[k] = key[k] + D;
                    k2[k] = key[k] - D;
                }
        }
        const int b = rowimg[row];
        const unsigned long long *tb = table + (long long)b * cap;
        const int32_t *s2r = slot2row + (long long)b * cap;
        unsigned long long p1, p2;
        int r1 = 0, r2 = 0;
        if (pack_key<D>(k1, p1)) {
            const int s = hash_lookup(tb, cap_mask, p1);
            if (s >= 0) r1 = s2r[s];
        }
        if (pack_key<D>(k2, p2)) {
            const int s = hash_lookup(tb, cap_mask, p2);
            if (s >= 0) r2 = s2r[s];
        }
        nbr[i] = make_int2(r1, r2);
    }
}

// ---- iteration kernels ------------------------------------------------------------------

// Splat in gather form: val[row][m] = sum over the row's pixels of w * (norm[p] * Q[p][m]).
// Work item = one chunk (<= SPLAT_CHUNK consecutive entries of one row's pixel list) x M lanes;
// a wave carries floor(64/M) chunks (M = 21 -> 63 of 64 lanes busy).  Rows of flat image regions
// gather thousands of pixels; cutting them bounds the serial chain per work item.  Entries are
// fetched 8 at a time so that 8 independent Q-row gathers are in flight per lane.
//
// The sum is accumulated in 64-bit FIXED POINT: every fp32 term w*norm*Q is rounded to a multiple
// of 2^-28 (|term| <= 2.5 because norm <= 1/sqrt(alpha/(d+1)), so it fits an int32) and added as
// an integer.  Integer addition is associative, so the result does not depend on the order in
// which csr_fill's atomic cursors laid the row out -- the iteration loop is bit-reproducible from
// run to run without sorting the lists and without float atomics.  Quantisation error per term is
// <= 1.9e-9 absolute, far inside the 1e-3 parity budget on Q.
// A single-chunk row writes val directly; chunks of long rows write int64 partials that
// splat_combine_kernel adds up.
// ONES: splat of the all-ones vector (M = 1) for the normalisation pass.
constexpr int SPLAT_CHUNK = 64;
constexpr int SPLAT_CU = 1; // chunks per lane group per trip of splat4_kernel
constexpr float FIX_SCALE = 268435456.0f;        // 2^28
constexpr float FIX_INV = 1.0f / 268435456.0f;   // 2^-28 (exact)

// Scalar form, used for the normalisation pass only (splat of the all-ones vector, M = 1).
__global__ __launch_bounds__(256) void splat_ones_kernel(const unsigned *__restrict__ start,
                                                         const int32_t *__restrict__ chunk_base,
                                                         const int32_t *__restrict__ chunk_row,
                                                         const float *__restrict__ csr_w, int n_chunks,
                                                         float *__restrict__ val, long long *__restrict__ part) {
    // 8 lanes per chunk (coalesced 32-byte runs of the weight list), integer partial sums folded by shuffles; the
    // 8 lanes of a group share c, so they enter and leave the loop together
    const int sub = threadIdx.x & 7;
    const long long nthr8 = (long long)gridDim.x * blockDim.x / 8;
    for (long long c = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / 8; c < n_chunks; c += nthr8) {
        const int row = chunk_row[c];
        const int cb = chunk_base[row];
        const bool single = chunk_base[row + 1] - cb == 1;
        const unsigned s = start[row] + (unsigned)(c - cb) * SPLAT_CHUNK;
        const unsigned e = min(s + SPLAT_CHUNK, start[row + 1]);
        long long acc = 0;
        for (unsigned i = s + sub; i < e; i += 8) acc += (long long)__float2int_rn(csr_w[i] * FIX_SCALE);
        for (int o = 4; o > 0; o >>= 1) {
            const unsigned lo = __shfl_down((unsigned)acc, o, 8), hi = __shfl_down((unsigned)(acc >> 32), o, 8);
            acc += (long long)(((unsigned long long)hi << 32) | lo);
        }
        if (sub == 0) {
            if (single) val[row] = (float)acc * FIX_INV;
            else part[c] = acc;
        }
    }
}

// Iteration form.  Q / U / val rows are padded to Mp = 4*LP floats (16-byte aligned rows) and a
// lane owns 4 consecutive classes: every gather is a 16-byte load.  (The L1 serves one access per
// clock whatever its width; with 4-byte lanes the gathers of this kernel ran at 1.04 L1 accesses
// per clock per CU -- L1-issue bound -- for 3.8 useful bytes each: profiles/r01_pmc_crf.txt.)
// LP lanes per chunk, floor(64/LP) chunks per wave.
__global__ __launch_bounds__(256) void splat4_kernel(const int4 *__restrict__ chunk_desc,
                                                     const uint2 *__restrict__ csr_ent,
                                                     const float *__restrict__ q, int LP, int n_local,
                                                     int rep, unsigned pix_stride, unsigned row_stride,
                                                     float *__restrict__ val, long long *__restrict__ part) {
    // n_local chunks per replica; replica k reads pixels + k*pix_stride and writes rows + k*row_stride
    const int gpw = 64 / LP;
    const int lane = threadIdx.x & 63;
    const int g = lane / LP;
    const int l = lane - g * LP;
    if (g >= gpw) return;
    long long cbeg, cend;
    int rk;
    xcd_range_rep(n_local, rep, rk, cbeg, cend);
    const f32x4_t *q4 = reinterpret_cast<const f32x4_t *>(q) + (size_t)rk * pix_stride * LP;
    val += (size_t)rk * row_stride * LP * 4;
    part += (size_t)rk * n_local * LP * 4;
    // descriptor -> entries -> Q rows is a chain of three dependent memory latencies; it is hidden by
    // occupancy, not by per-wave parallelism: one chunk per lane group per trip keeps the kernel at 63
    // VGPRs = 8 waves per SIMD.  Measured splat + combine per iteration (G + B, 32 images, M = 21):
    // 1 chunk 273 us, 2 chunks 278 us, 3 chunks 289 us, 4 chunks (161 VGPRs, 3 waves) 296 us.  16 entries
    // per batch (288 us) and 32-bit accumulators at 2^-23 (272 us) change nothing: the kernel is bound by
    // the gather path (L1/L2 requests), neither by VALU nor by exposed latency.
    constexpr int CU_ = SPLAT_CU;
    for (long long c0 = cbeg + (threadIdx.x >> 6) * gpw * CU_ + g; c0 < cend; c0 += (int)(blockDim.x >> 6) * gpw * CU_) {
        int4 d[CU_];
#pragma unroll
        for (int u = 0; u < CU_; ++u) {
            const long long c = c0 + (long long)u * gpw;
            d[u] = c < cend ? chunk_desc[c] : make_int4(0, 0, 0, 1);
        }
        long long acc[CU_][4];
#pragma unroll
        for (int u = 0; u < CU_; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[u][k] = 0;
        int maxlen = 0;
#pragma unroll
        for (int u = 0; u < CU_; ++u) maxlen = max(maxlen, d[u].y);
        for (int i = 0; i < maxlen; i += 8) {
            uint2 en[CU_][8];
            f32x4_t in[CU_][8];
#pragma unroll
            for (int u = 0; u < CU_; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    en[u][j] = i + j < d[u].y ? csr_ent[(unsigned)d[u].x + i + j] : make_uint2(0, 0); // weight 0
#pragma unroll
            for (int u = 0; u < CU_; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) in[u][j] = q4[en[u][j].x * (unsigned)LP + l];
#pragma unroll
            for (int u = 0; u < CU_; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float w = __uint_as_float(en[u][j].y); // w * norm[pixel]
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[u][k] += (long long)__float2int_rn((w * in[u][j][k]) * FIX_SCALE);
                }
        }
#pragma unroll
        for (int u = 0; u < CU_; ++u) {
            const long long c = c0 + (long long)u * gpw;
            if (c >= cend) continue;
            if (d[u].w) {
                f32x4_t o = {(float)acc[u][0] * FIX_INV, (float)acc[u][1] * FIX_INV, (float)acc[u][2] * FIX_INV,
                             (float)acc[u][3] * FIX_INV};
                reinterpret_cast<f32x4_t *>(val)[(unsigned)d[u].z * (unsigned)LP + l] = o;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) part[(c * LP + l) * 4 + k] = acc[u][k];
            }
        }
    }
}

// sums the int64 partials of multi-chunk rows; Mp = values per row (1 for the ones pass)
__global__ __launch_bounds__(256) void splat_combine_kernel(const int32_t *__restrict__ long_rows, int n_long,
                                                            const int32_t *__restrict__ chunk_base,
                                                            const long long *__restrict__ part, int Mp, int rep,
                                                            int chunk_stride, int row_stride,
                                                            float *__restrict__ val) {
    const long long total = (long long)n_long * rep * Mp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int gi = (int)(i / Mp);
        const int m = (int)(i - (long long)gi * Mp);
        const int k = gi / n_long; // replica
        const int row = long_rows[gi - k * n_long];
        const int cb = chunk_base[row], ce = chunk_base[row + 1];
        const long long pbase = (long long)k * chunk_stride;
        long long acc = 0;
        for (int c = cb; c < ce; ++c) acc += part[(pbase + c) * Mp + m];
        val[((long long)k * row_stride + row) * Mp + m] = (float)acc * FIX_INV;
    }
}

// chunk bookkeeping (lattice build)
__global__ void count_chunks_kernel(const unsigned *__restrict__ start, int rows, unsigned *__restrict__ nch) {
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += gridDim.x * blockDim.x) {
        const unsigned len = start[row + 1] - start[row];
        nch[row] = len <= SPLAT_CHUNK ? 1u : (len + SPLAT_CHUNK - 1) / SPLAT_CHUNK;
    }
}
__global__ void fill_chunks_kernel(const unsigned *__restrict__ start, const int32_t *__restrict__ chunk_base,
                                   int rows, int32_t *__restrict__ chunk_row, int4 *__restrict__ chunk_desc,
                                   unsigned *__restrict__ n_long, int32_t *__restrict__ long_rows) {
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += gridDim.x * blockDim.x) {
        const int cb = chunk_base[row], ce = chunk_base[row + 1];
        const unsigned s = start[row], e = start[row + 1];
        for (int c = cb; c < ce; ++c) {
            chunk_row[c] = row;
            const unsigned b = s + (unsigned)(c - cb) * SPLAT_CHUNK;
            chunk_desc[c] = make_int4((int)b, (int)(min(b + SPLAT_CHUNK, e) - b), row, ce - cb == 1 ? 1 : 0);
        }
        if (ce - cb > 1) long_rows[atomicAdd(n_long, 1u)] = row;
    }
}

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

template <int LH>
__global__ __launch_bounds__(GBI * GBJ) void blur3_tile_kernel(const f32x4_t *__restrict__ in, const int32_t *__restrict__ tile_rows,
                                                         const int32_t *__restrict__ tile_list, int n_occ, int LP,
                                                         int rows_local, int rep, f32x4_t *__restrict__ out) {
    // thread p owns point p of the 16 x 16 box and walks the row's float4s, LH at a time (rows wider than LH float4s
    // take several groups); LDS layout [l][p] (conflict-free 16-byte accesses, neighbours at fixed offsets in p)
    constexpr int P = GBI * GBJ; // one thread per point of the halo box
    __shared__ f32x4_t b0[LH * P], b1[LH * P];
    // XCD-contiguous logical block id: neighbouring tiles of one replica (which share halo rows) on one L2
    const int nb = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, qq = nb >> 3, rr = nb & 7;
    const int lb = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int k = lb / n_occ, t = lb - k * n_occ;
    in += (size_t)k * rows_local * LP;
    out += (size_t)k * rows_local * LP;
    const int p = threadIdx.x;
    const int li = p / GBJ, lj = p - li * GBJ;
    const int row = tile_rows[(long long)tile_list[t] * P + p];
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    if (t == 0 && p < LP) out[p] = zero; // the permanent zero row of this replica
    const bool r0 = row && li >= 1 && li < GBI - 1;      // pass 0 region
    const bool r1 = r0 && lj >= 1 && lj < GBJ - 1;       // pass 1 region
    const bool r2 = row && li >= 2 && li < GBI - 2 && lj >= 2 && lj < GBJ - 2; // interior
    for (int lbase = 0; lbase < LP; lbase += LH) {
        if (lbase > 0) __syncthreads(); // the previous group's pass-2 reads of b0 are done
#pragma unroll
        for (int l = 0; l < LH; ++l) b0[l * P + p] = (row && lbase + l < LP) ? in[(unsigned)row * (unsigned)LP + lbase + l] : zero;
        __syncthreads();
        // pass 0, axis 0: (i +- 1, j) = p +- GBJ
#pragma unroll
        for (int l = 0; l < LH; ++l) {
            f32x4_t o = zero;
            if (r0) {
                const f32x4_t c = b0[l * P + p], a = b0[l * P + p + GBJ], b = b0[l * P + p - GBJ];
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = c[q] + 0.5f * (a[q] + b[q]);
            }
            b1[l * P + p] = o;
        }
        __syncthreads();
        // pass 1, axis 1: (i, j +- 1) = p +- 1
#pragma unroll
        for (int l = 0; l < LH; ++l) {
            f32x4_t o = zero;
            if (r1) {
                const f32x4_t c = b1[l * P + p], a = b1[l * P + p - 1], b = b1[l * P + p + 1];
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = c[q] + 0.5f * (a[q] + b[q]);
            }
            b0[l * P + p] = o;
        }
        __syncthreads();
        // pass 2, axis 2: (i -+ 1, j -+ 1) = p -+ (GBJ + 1); interior only
        if (r2) {
#pragma unroll
            for (int l = 0; l < LH; ++l) {
                if (lbase + l >= LP) break;
                const f32x4_t c = b0[l * P + p], a = b0[l * P + p - GBJ - 1], b = b0[l * P + p + GBJ + 1];
                f32x4_t o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = c[q] + 0.5f * (a[q] + b[q]);
                out[(unsigned)row * (unsigned)LP + lbase + l] = o;
            }
        }
    }
}

// Slice of the ones-filter and norm = 1/sqrt(x + 1e-20)   (DenseKernel::initLattice)
__global__ void slice_norm_kernel(const int32_t *__restrict__ offset, const float *__restrict__ bary, int dp1,
                                  float alpha, const float *__restrict__ val, long long npix,
                                  float *__restrict__ norm) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix;
         p += (long long)gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int r = 0; r < dp1; ++r) acc += bary[p * dp1 + r] * val[offset[p * dp1 + r]] * alpha;
        norm[p] = (float)(1.0 / sqrt((double)acc + 1e-20));
    }
}

// csr_ent[i] = {pixel, w * norm[pixel]}: the splat then needs one 8-byte load per gathered pixel
__global__ void pack_entries_kernel(const int32_t *__restrict__ csr_pix, const float *__restrict__ csr_w,
                                    const float *__restrict__ norm, long long total, uint2 *__restrict__ ent) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int p = csr_pix[i];
        ent[i] = make_uint2((unsigned)p, __float_as_uint(csr_w[i] * norm[p]));
    }
}

// everything slice_update needs to know about a pixel in 80 contiguous bytes
__global__ void pack_pixels_kernel(const int32_t *__restrict__ off_g, const float *__restrict__ bary_g,
                                   const float *__restrict__ norm_g, const int32_t *__restrict__ off_b,
                                   const float *__restrict__ bary_b, const float *__restrict__ norm_b, long long npix,
                                   long long g_pix, uint32_t *__restrict__ rec) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix;
         p += (long long)gridDim.x * blockDim.x) {
        uint32_t *r = rec + p * 20;
        const long long pg = p % g_pix; // the Gaussian lattice arrays describe one image
        for (int i = 0; i < 3; ++i) r[i] = (uint32_t)off_g[pg * 3 + i];
        for (int i = 0; i < 6; ++i) r[3 + i] = (uint32_t)off_b[p * 6 + i];
        for (int i = 0; i < 3; ++i) r[9 + i] = __float_as_uint(bary_g[pg * 3 + i]);
        for (int i = 0; i < 6; ++i) r[12 + i] = __float_as_uint(bary_b[p * 6 + i]);
        r[18] = __float_as_uint(norm_g[pg]);
        r[19] = __float_as_uint(norm_b[p]);
    }
}

struct UpdateArgs {
    const uint4 *pix_rec; // [pixel][5]
    const float *val_g, *val_b;
    const float *u; // [pixel][Mp]
    float *q;       // [pixel][Mp]
    float alpha_g, alpha_b, compat_g, compat_b;
    int M, LP;
    long long npix;
    unsigned g_pix, g_rows; // shared Gaussian lattice: pixels / rows per replica (g_rows = 0: not shared)
};

// Slice both lattices, add the unary, softmax over classes (DenseCRF::inference loop body):
//   E = -U - (-wG * normG * sliceG) - (-wB * normB * sliceB);  Q = expAndNormalize(E)
// LP lanes per pixel (4 classes each, 16-byte gathers), floor(64/LP) pixels per wave; the max / sum
// over a pixel's classes are reduced inside the lane, then across the pixel's LP lanes by
// shuffle-down with a segment bound and a broadcast from the segment's first lane.
__global__ __launch_bounds__(256) void slice_update_kernel(UpdateArgs a) {
    const int LP = a.LP;
    const int gpw = 64 / LP;
    const int lane = threadIdx.x & 63;
    const int g = lane / LP;
    const int l = lane - g * LP;
    const bool act = g < gpw;
    const int seg0 = g * LP;
    const f32x4_t *vg4 = reinterpret_cast<const f32x4_t *>(a.val_g);
    const f32x4_t *vb4 = reinterpret_cast<const f32x4_t *>(a.val_b);
    const f32x4_t *u4 = reinterpret_cast<const f32x4_t *>(a.u);
    f32x4_t *q4 = reinterpret_cast<f32x4_t *>(a.q);
    long long pbeg, pend;
    xcd_range(a.npix, pbeg, pend);
    constexpr int U = 1; // pixels per lane group per trip (2 measured slower: 362 vs 332 us)
    for (long long p0 = pbeg + (long long)(threadIdx.x >> 6) * gpw * U; p0 < pend; p0 += (long long)(blockDim.x >> 6) * gpw * U) {
        long long pp[U];
        bool ok[U];
        uint32_t rc[U][20];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pp[u] = p0 + (long long)u * gpw + g;
            ok[u] = act && pp[u] < pend;
            const long long p = ok[u] ? pp[u] : pbeg;
            // the pixel's record: 5 x 16 bytes, identical for the LP lanes of the pixel (broadcast loads)
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const uint4 t = a.pix_rec[p * 5 + i];
                rc[u][4 * i] = t.x; rc[u][4 * i + 1] = t.y; rc[u][4 * i + 2] = t.z; rc[u][4 * i + 3] = t.w;
            }
        }
        f32x4_t vg[U][3], vb[U][6], un[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long p = ok[u] ? pp[u] : pbeg;
            const unsigned gofs = a.g_rows ? ((unsigned)p / a.g_pix) * a.g_rows : 0u;
#pragma unroll
            for (int r = 0; r < 3; ++r) vg[u][r] = vg4[(rc[u][r] + gofs) * (unsigned)LP + l];
#pragma unroll
            for (int r = 0; r < 6; ++r) vb[u][r] = vb4[rc[u][3 + r] * (unsigned)LP + l];
            un[u] = u4[p * LP + l];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float ng = __uint_as_float(rc[u][18]), nb = __uint_as_float(rc[u][19]);
            float e[4];
            float mx = -3.0e38f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float sg = 0.f, sb = 0.f;
#pragma unroll
                for (int r = 0; r < 3; ++r) sg += __uint_as_float(rc[u][9 + r]) * vg[u][r][k] * a.alpha_g;
#pragma unroll
                for (int r = 0; r < 6; ++r) sb += __uint_as_float(rc[u][12 + r]) * vb[u][r][k] * a.alpha_b;
                float ek = -un[u][k];
                ek -= -a.compat_g * (sg * ng);
                ek -= -a.compat_b * (sb * nb);
                const bool valid = ok[u] && 4 * l + k < a.M;
                e[k] = valid ? ek : -3.0e38f;
                mx = fmaxf(mx, e[k]);
            }
            for (int o = 4; o > 0; o >>= 1) {
                const float other = __shfl_down(mx, o, 64);
                if (l + o < LP) mx = fmaxf(mx, other);
            }
            mx = __shfl(mx, seg0, 64);
            float ex[4], sum = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ex[k] = (ok[u] && 4 * l + k < a.M) ? expf(e[k] - mx) : 0.f;
                sum += ex[k];
            }
            for (int o = 4; o > 0; o >>= 1) {
                const float other = __shfl_down(sum, o, 64);
                if (l + o < LP) sum += other;
            }
            sum = __shfl(sum, seg0, 64);
            if (ok[u]) {
                f32x4_t o4 = {ex[0] / sum, ex[1] / sum, ex[2] / sum, ex[3] / sum};
                q4[pp[u] * LP + l] = o4;
            }
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
        q[obase + i] = m < M ? tq[m * (TP + 1) + n] : 0.f;
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
void splat_ones(wsc_ctx *ctx, const LatticeDev &L, float *val, long long *part) {
    hipLaunchKernelGGL(splat_ones_kernel, dim3(grid1d((long long)L.n_chunks * 8)), dim3(256), 0, ctx->stream,
                       (const unsigned *)L.csr_start, L.chunk_base, L.chunk_row, L.csr_w, L.n_chunks, val, part);
    if (L.n_long > 0)
        hipLaunchKernelGGL(splat_combine_kernel, dim3(grid1d(L.n_long, 256, 4096)), dim3(256), 0, ctx->stream,
                           L.long_rows, L.n_long, L.chunk_base, part, 1, 1, 0, 0, val);
}

// part: scratch of n_chunks * Mp int64 (partials of multi-chunk rows)
void splat4(wsc_ctx *ctx, const LatticeDev &L, const float *q, int LP, float *val, long long *part) {
    const int gpw = 64 / LP;
    // algorithmic bytes: read the batch's Q once + (pixel index, weight) per gathered pixel + write the rows
    const double npix = (double)L.n_pix * L.rep, rows = (double)L.rows * L.rep;
    WscKernelTimer timer(ctx, WSC_K_SPLAT, npix * L.M_cur * 4 + npix * (L.d + 1) * 8 + rows * L.M_cur * 4);
    // 256 threads per block: 128 / 64 measured 289 / 320 us against 270 (the opposite of slice_update)
    hipLaunchKernelGGL(splat4_kernel, dim3(grid_rep((long long)L.n_chunks * L.rep, 4 * SPLAT_CU * gpw, L.rep)), dim3(256), 0,
                       ctx->stream, L.chunk_desc, L.csr_ent, q, LP, L.n_chunks, L.rep, (unsigned)L.n_pix,
                       (unsigned)L.rows, val, part);
    if (L.n_long > 0)
        hipLaunchKernelGGL(splat_combine_kernel, dim3(grid1d((long long)L.n_long * L.rep * 4 * LP, 256, 4096)),
                           dim3(256), 0, ctx->stream, L.long_rows, L.n_long, L.chunk_base, part, 4 * LP, L.rep,
                           L.n_chunks, L.rows, val);
}

// d+1 blur passes, ping-pong between a and b; returns the buffer holding the result
float *blur_all1(wsc_ctx *ctx, const LatticeDev &L, float *a, float *b) {
    for (int j = 0; j <= L.d; ++j) {
        hipLaunchKernelGGL(blur1_kernel, dim3(grid1d(L.rows)), dim3(256), 0, ctx->stream, a,
                           L.nbr + (long long)j * L.rows, L.rows, b);
        float *t = a; a = b; b = t;
    }
    return a;
}
float *blur_all4(wsc_ctx *ctx, const LatticeDev &L, int LP, float *a, float *b) {
    // WSC_CRF_NO_FUSED_BLUR=1 (read per call, so a test can flip it) keeps the three separate passes
    const char *fe = getenv("WSC_CRF_NO_FUSED_BLUR");
    const bool fused_off = fe && atoi(fe) != 0;
    if (L.d == 2 && L.tile_rows && L.n_tiles_occ > 0 && !fused_off) {
        // one read + one write of the value rows (the halo re-reads come out of L2).  Whole rows per group
        // (6 float4 = 49 KB of LDS, 3 blocks per CU) beat 3 / 2 / 1 float4 per group at 6+ blocks per CU:
        // blur 2.37 vs 2.42 / 2.72 / 3.15 ms per step.
        WscKernelTimer timer(ctx, WSC_K_BLUR, 2.0 * L.rows * L.rep * L.M_cur * 4);
        hipLaunchKernelGGL(blur3_tile_kernel<WSC_GLH>, dim3((unsigned)(L.n_tiles_occ * L.rep)), dim3(GBI * GBJ), 0, ctx->stream,
                           (const f32x4_t *)a, L.tile_rows, L.tile_list, L.n_tiles_occ, LP, L.rows, L.rep, (f32x4_t *)b);
        return b;
    }
    for (int j = 0; j <= L.d; ++j) {
        WscKernelTimer timer(ctx, WSC_K_BLUR, 2.0 * L.rows * L.rep * L.M_cur * 4); // read + write every row once
        hipLaunchKernelGGL(blur4_kernel, dim3(grid_rep((long long)L.rows * L.rep, (256 / LP) * 4, L.rep)), dim3(256),
                           0, ctx->stream, (const f32x4_t *)a, L.nbr + (long long)j * L.rows, LP, L.rows, L.rep,
                           (f32x4_t *)b);
        float *t = a; a = b; b = t;
    }
    return a;
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

    TempBuf tmp(ctx);
    unsigned long long *table;
    int32_t *first, *eslot, *slot2row, *rowimg;
    unsigned *flag, *prefix, *sums, *count, *cursor;
    unsigned long long *rowkey;
    int *err;
    WSC_TRY(tmp.alloc(sizeof(unsigned long long) * B * cap, (void **)&table));
    WSC_TRY(tmp.alloc(sizeof(int32_t) * B * cap, (void **)&first));
    WSC_TRY(tmp.alloc(sizeof(int32_t) * B * cap, (void **)&slot2row));
    WSC_TRY(tmp.alloc(sizeof(int32_t) * total, (void **)&eslot));
    WSC_TRY(tmp.alloc(sizeof(unsigned) * total, (void **)&flag));
    WSC_TRY(tmp.alloc(sizeof(unsigned) * (total + 1), (void **)&prefix));
    const int nblk = (int)((total + SCAN_CHUNK - 1) / SCAN_CHUNK);
    WSC_TRY(tmp.alloc(sizeof(unsigned) * (nblk + 2), (void **)&sums));
    WSC_TRY(tmp.alloc(sizeof(int), (void **)&err));
    WSC_TRY(crf_alloc(crf, sizeof(int32_t) * total, (void **)&L.offset));
    WSC_TRY(crf_alloc(crf, sizeof(float) * total, (void **)&L.bary));
    WSC_TRY(crf_alloc(crf, sizeof(float) * npix, (void **)&L.norm));
    WSC_TRY(crf_alloc(crf, sizeof(int32_t) * total, (void **)&L.csr_pix));
    WSC_TRY(crf_alloc(crf, sizeof(float) * total, (void **)&L.csr_w));

    hipLaunchKernelGGL(fill_u64_kernel, dim3(grid1d(B * cap)), dim3(256), 0, ctx->stream, table, EMPTY_KEY,
                       (long long)B * cap);
    hipLaunchKernelGGL(fill_u32_kernel, dim3(grid1d(B * cap)), dim3(256), 0, ctx->stream, (unsigned *)first,
                       0x7fffffffu, (long long)B * cap);
    WSC_HIP(hipMemsetAsync(err, 0, sizeof(int), ctx->stream));

    EmbedArgs ea;
    ea.rgb = rgb_dev; ea.B = B; ea.H = crf->H; ea.W = crf->W; ea.N = N;
    ea.inv_sxy_den = sxy; ea.inv_srgb_den = srgb;
    {   // Permutohedral::init: scale_factor[i] = 1/sqrt((i+2)(i+1)) * sqrt(2/3)*(d+1)
        const float inv_std_dev = (float)(std::sqrt(2.0 / 3.0) * (D + 1));
        for (int i = 0; i < 5; ++i)
            ea.scale[i] = i < D ? (float)(1.0 / std::sqrt((double)((i + 2) * (i + 1))) * inv_std_dev) : 0.f;
    }
    ea.table = table; ea.cap_mask = (unsigned)(cap - 1); ea.cap = cap;
    ea.eslot = eslot; ea.bary = L.bary; ea.first = first; ea.err = err;
    hipLaunchKernelGGL(lattice_embed_kernel<D>, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, ctx->stream, ea);
    hipLaunchKernelGGL(flag_first_kernel, dim3(grid1d(total)), dim3(256), 0, ctx->stream, eslot, first, cap, N, dp1,
                       total, flag);
    WSC_TRY(exclusive_scan(ctx, flag, total, prefix, sums));
    // vertex counts: grand total and per-image boundaries
    int herr = 0;
    unsigned vtot = 0;
    WSC_HIP(hipMemcpyAsync(&vtot, sums + nblk, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    WSC_HIP(hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    std::vector<unsigned> bound(B, 0);
    for (int b = 1; b < B; ++b)
        WSC_HIP(hipMemcpyAsync(&bound[b], prefix + (long long)b * N * dp1, sizeof(unsigned), hipMemcpyDeviceToHost,
                               ctx->stream));
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

    WSC_TRY(tmp.alloc(sizeof(unsigned long long) * L.rows, (void **)&rowkey));
    WSC_TRY(tmp.alloc(sizeof(int32_t) * L.rows, (void **)&rowimg));
    WSC_TRY(tmp.alloc(sizeof(unsigned) * (L.rows + 1), (void **)&count));
    WSC_TRY(tmp.alloc(sizeof(unsigned) * (L.rows + 1), (void **)&cursor));
    WSC_TRY(crf_alloc(crf, sizeof(int32_t) * (L.rows + 2), (void **)&L.csr_start));
    WSC_TRY(crf_alloc(crf, sizeof(int2) * (size_t)dp1 * L.rows, (void **)&L.nbr));
    WSC_HIP(hipMemsetAsync(count, 0, sizeof(unsigned) * (L.rows + 1), ctx->stream));
    WSC_HIP(hipMemsetAsync(cursor, 0, sizeof(unsigned) * (L.rows + 1), ctx->stream));

    hipLaunchKernelGGL(assign_ids_kernel, dim3(grid1d(total)), dim3(256), 0, ctx->stream, eslot, flag, prefix, table,
                       cap, N, dp1, total, slot2row, rowkey, rowimg);
    hipLaunchKernelGGL(remap_count_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, ctx->stream, eslot,
                       slot2row, cap, N, dp1, npix, L.offset, count);
    {
        unsigned *sums2;
        const int nb2 = (L.rows + 1 + SCAN_CHUNK - 1) / SCAN_CHUNK;
        WSC_TRY(tmp.alloc(sizeof(unsigned) * (nb2 + 2), (void **)&sums2));
        WSC_TRY(exclusive_scan(ctx, count, L.rows + 1, (unsigned *)L.csr_start, sums2));
    }
    hipLaunchKernelGGL(csr_fill_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, ctx->stream, L.offset,
                       L.bary, dp1, npix, (const unsigned *)L.csr_start, cursor, L.csr_pix, L.csr_w);
    hipLaunchKernelGGL(neighbors_kernel<D>, dim3(grid1d((long long)L.rows * dp1)), dim3(256), 0, ctx->stream, rowkey,
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
            }
        }
    }
    {   // splat chunk tables
        unsigned *nch, *sums3, *n_long_dev;
        const int nb3 = (L.rows + 1 + SCAN_CHUNK - 1) / SCAN_CHUNK;
        WSC_TRY(tmp.alloc(sizeof(unsigned) * (L.rows + 1), (void **)&nch));
        WSC_TRY(tmp.alloc(sizeof(unsigned) * (nb3 + 2), (void **)&sums3));
        WSC_TRY(tmp.alloc(sizeof(unsigned), (void **)&n_long_dev));
        WSC_TRY(crf_alloc(crf, sizeof(int32_t) * (L.rows + 2), (void **)&L.chunk_base));
        WSC_HIP(hipMemsetAsync(nch, 0, sizeof(unsigned) * (L.rows + 1), ctx->stream));
        WSC_HIP(hipMemsetAsync(n_long_dev, 0, sizeof(unsigned), ctx->stream));
        hipLaunchKernelGGL(count_chunks_kernel, dim3(grid1d(L.rows)), dim3(256), 0, ctx->stream,
                           (const unsigned *)L.csr_start, L.rows, nch);
        WSC_TRY(exclusive_scan(ctx, nch, L.rows + 1, (unsigned *)L.chunk_base, sums3));
        unsigned tc = 0;
        WSC_HIP(hipMemcpyAsync(&tc, sums3 + nb3, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        WSC_HIP(hipStreamSynchronize(ctx->stream));
        L.n_chunks = (int)tc;
        WSC_TRY(crf_alloc(crf, sizeof(int32_t) * (size_t)L.n_chunks, (void **)&L.chunk_row));
        WSC_TRY(crf_alloc(crf, sizeof(int4) * (size_t)L.n_chunks, (void **)&L.chunk_desc));
        WSC_TRY(crf_alloc(crf, sizeof(int32_t) * (size_t)L.rows, (void **)&L.long_rows));
        hipLaunchKernelGGL(fill_chunks_kernel, dim3(grid1d(L.rows)), dim3(256), 0, ctx->stream,
                           (const unsigned *)L.csr_start, L.chunk_base, L.rows, L.chunk_row, L.chunk_desc, n_long_dev,
                           L.long_rows);
        unsigned nl = 0;
        WSC_HIP(hipMemcpyAsync(&nl, n_long_dev, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        WSC_HIP(hipStreamSynchronize(ctx->stream));
        L.n_long = (int)nl;
    }

    // norm = 1/sqrt(Lattice(1) + 1e-20)
    float *va, *vb;
    long long *vp;
    WSC_TRY(tmp.alloc(sizeof(float) * L.rows, (void **)&va));
    WSC_TRY(tmp.alloc(sizeof(float) * L.rows, (void **)&vb));
    WSC_TRY(tmp.alloc(sizeof(long long) * L.n_chunks, (void **)&vp));
    splat_ones(ctx, L, va, vp);
    float *res = blur_all1(ctx, L, va, vb);
    hipLaunchKernelGGL(slice_norm_kernel, dim3(grid1d(npix)), dim3(256), 0, ctx->stream, L.offset, L.bary, dp1,
                       L.alpha, res, npix, L.norm);
    WSC_TRY(crf_alloc(crf, sizeof(uint2) * total, (void **)&L.csr_ent));
    hipLaunchKernelGGL(pack_entries_kernel, dim3(grid1d(total)), dim3(256), 0, ctx->stream, L.csr_pix, L.csr_w, L.norm,
                       total, L.csr_ent);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

struct GaussCache {
    int H, W;
    float sxy;
    LatticeDev L;
};
constexpr int GAUSS_CACHE_MAX = 16; // distinct image sizes kept per ctx; later sizes are rebuilt per call
void gauss_cache_delete(void *p) { delete static_cast<GaussCache *>(p); }

void launch_update(wsc_ctx *ctx, const UpdateArgs &a) {
    const int gpw = 64 / a.LP;
    // algorithmic bytes: slice index+weight of both lattices, read U, write Q (+ the two
    // messages the reference materialises: N*M*4 each)
    WscKernelTimer timer(ctx, WSC_K_SLICE_UPDATE, (double)a.npix * (9 * 8 + 4.0 * a.M * 4));
    // one trip (4 * gpw pixels) per block: with the grid capped at 16 384 / 65 536 blocks the kernel took 324 / 353 us,
    // uncapped (82 k blocks for 32 images at 321^2) 299 us -- whole trips per block, no ragged per-block ranges
    // ... and 128-thread blocks (20 pixels): 302 -> 289 us (64 threads: 291)
    hipLaunchKernelGGL(slice_update_kernel, dim3(grid1d(a.npix, 2 * gpw, 1 << 22)), dim3(128), 0, ctx->stream, a);
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
    GaussCache *hit = nullptr;
    int n_cached = 0;
    for (auto &a : ctx->attachments)
        if (a.second == &gauss_cache_delete) {
            ++n_cached;
            GaussCache *g = static_cast<GaussCache *>(a.first);
            if (g->H == H && g->W == W && g->sxy == g_sxy) hit = g;
        }
    if (hit) {
        crf->lat[0] = hit->L;
    } else {
        crf->persist = n_cached < GAUSS_CACHE_MAX;
        st = build_lattice<2>(crf, crf->lat[0], rgb_dev, g_sxy, 1.f, true, true); // built once: worst-case table
        if (st == WSC_OK && crf->persist) {
            GaussCache *g = new GaussCache{H, W, g_sxy, crf->lat[0]};
            ctx->attachments.emplace_back(g, &gauss_cache_delete);
        }
        // on failure the partially built arrays stay with the ctx and are released at wsc_ctx_destroy
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
            crf->lat[1] = LatticeDev();
            st = build_lattice<5>(crf, crf->lat[1], rgb_dev, bi_sxy, bi_srgb, false, true);
        }
    }
    if (st == WSC_OK) st = crf_alloc(crf, sizeof(uint4) * 5 * (size_t)B * crf->N, (void **)&crf->pix_rec);
    if (st == WSC_OK) {
        const long long npix = (long long)B * crf->N;
        hipLaunchKernelGGL(pack_pixels_kernel, dim3(grid1d(npix)), dim3(256), 0, ctx->stream, crf->lat[0].offset,
                           crf->lat[0].bary, crf->lat[0].norm, crf->lat[1].offset, crf->lat[1].bary, crf->lat[1].norm,
                           npix, (long long)crf->lat[0].n_pix, (uint32_t *)crf->pix_rec);
    }
    if (st != WSC_OK) {
        wsc_crf_destroy(crf);
        return st;
    }
    *out = crf;
    return WSC_OK;
}

void wsc_crf_destroy(wsc_crf *crf) {
    if (!crf) return;
    for (void *p : crf->allocs) wsc_ctx_cached_free(crf->ctx, p); // reused in stream order
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

int wsc_crf_inference(wsc_ctx *ctx, wsc_crf *crf, const float *unary_dev, int M, float g_compat, float bi_compat,
                      int n_iters, float *q_dev, int32_t *argmax_dev) {
    WSC_CHECK(ctx && crf && unary_dev, WSC_ERR_INVALID, "wsc_crf_inference: null argument");
    WSC_CHECK(M >= 1 && M <= 32, WSC_ERR_INVALID, "wsc_crf_inference: M=%d outside [1,32]", M);
    WSC_CHECK(n_iters >= 0, WSC_ERR_INVALID, "wsc_crf_inference: n_iters=%d", n_iters);
    WSC_HIP(hipSetDevice(ctx->device));
    const int B = crf->B, N = crf->N;
    const long long npix = (long long)B * N;
    const LatticeDev &G = crf->lat[0], &Bl = crf->lat[1];
    const int LP = (M + 3) / 4, Mp = 4 * LP; // rows padded to 16-byte multiples
    const long long g_rows = (long long)G.rows * G.rep, g_chunks = (long long)G.n_chunks * G.rep;
    WSC_CHECK(npix * Mp < (1ll << 31) && g_rows * Mp < (1ll << 31) && (long long)Bl.rows * Mp < (1ll << 31),
              WSC_ERR_CAPACITY, "CRF batch too large for 32-bit element indices (B*N*Mp = %lld)", npix * Mp);
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t qb = al(sizeof(float) * npix * Mp);
    const size_t vg = al(sizeof(float) * (size_t)g_rows * Mp), vb = al(sizeof(float) * (size_t)Bl.rows * Mp);
    const size_t pg = al(sizeof(long long) * (size_t)g_chunks * Mp), pb = al(sizeof(long long) * (size_t)Bl.n_chunks * Mp);
    void *ws;
    WSC_TRY(wsc_ctx_workspace(ctx, 2 * qb + 2 * vg + 2 * vb + pg + pb, &ws));
    char *p = (char *)ws;
    float *u = (float *)p; p += qb;
    float *q = (float *)p; p += qb;
    float *vg0 = (float *)p; p += vg;
    float *vg1 = (float *)p; p += vg;
    float *vb0 = (float *)p; p += vb;
    float *vb1 = (float *)p; p += vb;
    long long *partg = (long long *)p; p += pg;
    long long *partb = (long long *)p; p += pb;

    crf->lat[0].M_cur = M;
    crf->lat[1].M_cur = M;
    const dim3 tgrid((N + TP - 1) / TP, B);
    {
        WscKernelTimer timer(ctx, WSC_K_CRF_MISC, (double)npix * M * 12);
        hipLaunchKernelGGL(init_q_kernel, tgrid, dim3(TP), 2 * (size_t)M * (TP + 1) * sizeof(float), ctx->stream,
                           unary_dev, M, Mp, N, u, q);
    }
    for (int it = 0; it < n_iters; ++it) {
        splat4(ctx, G, q, LP, vg0, partg);
        float *rg = blur_all4(ctx, G, LP, vg0, vg1);
        splat4(ctx, Bl, q, LP, vb0, partb);
        float *rb = blur_all4(ctx, Bl, LP, vb0, vb1);
        UpdateArgs a;
        a.pix_rec = crf->pix_rec; a.val_g = rg; a.val_b = rb;
        a.u = u; a.q = q;
        a.alpha_g = G.alpha; a.alpha_b = Bl.alpha; a.compat_g = g_compat; a.compat_b = bi_compat;
        a.M = M; a.LP = LP; a.npix = npix;
        a.g_pix = (unsigned)G.n_pix; a.g_rows = G.rep > 1 ? (unsigned)G.rows : 0u;
        launch_update(ctx, a);
    }
    WscKernelTimer ftimer(ctx, WSC_K_CRF_MISC, (double)npix * M * 8);
    if (q_dev || argmax_dev)
        hipLaunchKernelGGL(finish_kernel, tgrid, dim3(TP), (size_t)M * (TP + 1) * sizeof(float), ctx->stream, q, M, Mp,
                           N, q_dev, argmax_dev);
    WSC_HIP(hipGetLastError());
    return WSC_OK;
}

} // extern "C"
