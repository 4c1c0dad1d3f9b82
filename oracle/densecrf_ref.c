/*
 * densecrf_ref.c -- TEST INFRASTRUCTURE ONLY (oracle).  Nothing in the product
 * path may call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg load it.
 *
 * PARITY UNPINNED: the arithmetic this file restates lives in the third-party
 * package pydensecrf (github.com/lucasb-eyer/pydensecrf, pulled at git HEAD by
 * the reference's requirements.txt:11), which wraps Kraehenbuehl & Koltun's
 * DenseCRF v2 C++ (Eigen).  Neither pydensecrf nor its sources are under
 * /root/reference, it is not installed in this image, and the reference holds
 * no golden vectors for it.  This is a plain-C, single-threaded restatement of
 * the published algorithm (Adams, Baek & Davis 2010, "Fast high-dimensional
 * filtering using the permutohedral lattice"; Kraehenbuehl & Koltun NIPS'11),
 * following the structure of that public implementation so results line up
 * with it as closely as possible: float arithmetic throughout, short lattice
 * keys, int `sum` accumulator fed by float products, values stored at
 * offset+1 so a missing neighbour reads a permanent zero row, blur axes
 * 0..d in forward order, alpha = 1/(1+2^-d), NORMALIZE_SYMMETRIC, Potts
 * compatibility.  It is pinned instead by tests/test_crf_oracle.py against an
 * exact O(N^2) Gaussian mean-field and algebraic invariants.
 *
 * Reference call sites this stands in for:
 *   03c_hsn/utilities.py:427-444  (dcrf_process: DenseCRF2D, setUnaryEnergy,
 *       addPairwiseGaussian, addPairwiseBilateral, inference, argmax)
 *   03b_irn/step/cam_to_ir_label.py:35,47,52,67 (imutils.crf_inference_label)
 *   03a_sec-dsrg/SEC.py:275, DSRG.py:328, model.py:689,693 (lib.crf.crf_inference)
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -o libdensecrf_ref.so densecrf_ref.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- hash table over short keys, linear probing (insertion order = vertex id) ---- */
typedef struct {
    int key_size;
    size_t filled, capacity;
    short *keys;
    int *table;
} hash_t;

static void hash_init(hash_t *h, int key_size, size_t n_elements) {
    h->key_size = key_size;
    h->filled = 0;
    h->capacity = 2 * n_elements + 16;
    h->keys = (short *)malloc(sizeof(short) * (h->capacity / 2 + 10) * key_size);
    h->table = (int *)malloc(sizeof(int) * h->capacity);
    for (size_t i = 0; i < h->capacity; ++i) h->table[i] = -1;
}
static void hash_free(hash_t *h) {
    free(h->keys);
    free(h->table);
}
static size_t hash_fn(const hash_t *h, const short *k) {
    size_t r = 0;
    for (int i = 0; i < h->key_size; ++i) {
        r += (size_t)(long)k[i];
        r *= 1664525u;
    }
    return r;
}
static int hash_find(hash_t *h, const short *k, int create) {
    size_t p = hash_fn(h, k) % h->capacity;
    for (;;) {
        int e = h->table[p];
        if (e == -1) {
            if (!create) return -1;
            for (int i = 0; i < h->key_size; ++i) h->keys[h->filled * h->key_size + i] = k[i];
            h->table[p] = (int)h->filled;
            return (int)h->filled++;
        }
        int good = 1;
        for (int i = 0; i < h->key_size && good; ++i)
            if (h->keys[(size_t)e * h->key_size + i] != k[i]) good = 0;
        if (good) return e;
        if (++p == h->capacity) p = 0;
    }
}

/* ---- permutohedral lattice ---- */
typedef struct {
    int N, M, d;       /* pixels, lattice vertices, feature dimension */
    int *offset;       /* [N][d+1] vertex id */
    float *barycentric;/* [N][d+1] */
    int *n1, *n2;      /* [d+1][M] blur neighbours (-1 = absent) */
} lattice_t;

static void lattice_free(lattice_t *L) {
    free(L->offset);
    free(L->barycentric);
    free(L->n1);
    free(L->n2);
}

/* feature: [N][d] (pixel-major) */
static void lattice_init(lattice_t *L, const float *feature, int N, int d) {
    L->N = N;
    L->d = d;
    hash_t ht;
    hash_init(&ht, d, (size_t)N * (d + 1));
    L->offset = (int *)malloc(sizeof(int) * (size_t)N * (d + 1));
    L->barycentric = (float *)malloc(sizeof(float) * (size_t)N * (d + 1));

    float scale_factor[16], elevated[17], rem0[17], barycentric[18];
    short rank[17], canonical[17 * 17], key[17];

    for (int i = 0; i <= d; ++i) {
        for (int j = 0; j <= d - i; ++j) canonical[i * (d + 1) + j] = (short)i;
        for (int j = d - i + 1; j <= d; ++j) canonical[i * (d + 1) + j] = (short)(i - (d + 1));
    }
    /* expected standard deviation of the filter (p.6 of Adams et al.) */
    float inv_std_dev = sqrt(2.0 / 3.0) * (d + 1);
    for (int i = 0; i < d; ++i) scale_factor[i] = 1.0 / sqrt((double)((i + 2) * (i + 1))) * inv_std_dev;

    for (int k = 0; k < N; ++k) {
        const float *f = feature + (size_t)k * d;
        /* elevate: y = E p */
        float sm = 0;
        for (int j = d; j > 0; --j) {
            float cf = f[j - 1] * scale_factor[j - 1];
            elevated[j] = sm - j * cf;
            sm += cf;
        }
        elevated[0] = sm;

        /* closest 0-coloured simplex by rounding */
        float down_factor = 1.0f / (d + 1);
        float up_factor = (float)(d + 1);
        int sum = 0;
        for (int i = 0; i <= d; ++i) {
            int rd2;
            float v = down_factor * elevated[i];
            float up = ceilf(v) * up_factor;
            float down = floorf(v) * up_factor;
            if (up - elevated[i] < elevated[i] - down)
                rd2 = (short)up;
            else
                rd2 = (short)down;
            rem0[i] = (float)rd2;
            sum += rd2 * down_factor; /* int += float: truncates on every add */
        }

        /* rank of each coordinate's residual */
        for (int i = 0; i <= d; ++i) rank[i] = 0;
        for (int i = 0; i < d; ++i) {
            double di = elevated[i] - rem0[i];
            for (int j = i + 1; j <= d; ++j)
                if (di < elevated[j] - rem0[j])
                    rank[i]++;
                else
                    rank[j]++;
        }
        /* bring the point back onto the plane if sum != 0 */
        for (int i = 0; i <= d; ++i) {
            rank[i] += sum;
            if (rank[i] < 0) {
                rank[i] += d + 1;
                rem0[i] += d + 1;
            } else if (rank[i] > d) {
                rank[i] -= d + 1;
                rem0[i] -= d + 1;
            }
        }
        /* barycentric coordinates (p.10 of Adams et al.) */
        for (int i = 0; i <= d + 1; ++i) barycentric[i] = 0;
        for (int i = 0; i <= d; ++i) {
            float v = (elevated[i] - rem0[i]) * down_factor;
            barycentric[d - rank[i]] += v;
            barycentric[d - rank[i] + 1] -= v;
        }
        barycentric[0] += 1.0 + barycentric[d + 1];

        for (int remainder = 0; remainder <= d; ++remainder) {
            for (int i = 0; i < d; ++i) key[i] = (short)(rem0[i] + canonical[remainder * (d + 1) + rank[i]]);
            L->offset[(size_t)k * (d + 1) + remainder] = hash_find(&ht, key, 1);
            L->barycentric[(size_t)k * (d + 1) + remainder] = barycentric[remainder];
        }
    }

    L->M = (int)ht.filled;
    L->n1 = (int *)malloc(sizeof(int) * (size_t)(d + 1) * L->M);
    L->n2 = (int *)malloc(sizeof(int) * (size_t)(d + 1) * L->M);
    short n1[17], n2[17];
    for (int j = 0; j <= d; ++j)
        for (int i = 0; i < L->M; ++i) {
            const short *kk = ht.keys + (size_t)i * d;
            for (int k = 0; k < d; ++k) {
                n1[k] = (short)(kk[k] - 1);
                n2[k] = (short)(kk[k] + 1);
            }
            if (j < d) { /* the (d+1)-th coordinate is not stored/hashed */
                n1[j] = (short)(kk[j] + d);
                n2[j] = (short)(kk[j] - d);
            }
            L->n1[(size_t)j * L->M + i] = hash_find(&ht, n1, 0);
            L->n2[(size_t)j * L->M + i] = hash_find(&ht, n2, 0);
        }
    hash_free(&ht);
}

/* in/out: [N][vs] pixel-major, class-minor (Eigen column-major M x N) */
static void lattice_compute(const lattice_t *L, float *out, const float *in, int vs) {
    const int d = L->d, N = L->N, M = L->M;
    size_t sz = (size_t)(M + 2) * vs;
    float *values = (float *)calloc(sz, sizeof(float));
    float *new_values = (float *)calloc(sz, sizeof(float));
    /* splat */
    for (int i = 0; i < N; ++i)
        for (int j = 0; j <= d; ++j) {
            int o = L->offset[(size_t)i * (d + 1) + j] + 1;
            float w = L->barycentric[(size_t)i * (d + 1) + j];
            for (int k = 0; k < vs; ++k) values[(size_t)o * vs + k] += w * in[(size_t)i * vs + k];
        }
    /* blur along each of the d+1 axes */
    for (int j = 0; j <= d; ++j) {
        for (int i = 0; i < M; ++i) {
            const float *old_val = values + (size_t)(i + 1) * vs;
            float *new_val = new_values + (size_t)(i + 1) * vs;
            int n1 = L->n1[(size_t)j * M + i] + 1;
            int n2 = L->n2[(size_t)j * M + i] + 1;
            const float *n1_val = values + (size_t)n1 * vs;
            const float *n2_val = values + (size_t)n2 * vs;
            for (int k = 0; k < vs; ++k) new_val[k] = old_val[k] + 0.5f * (n1_val[k] + n2_val[k]);
        }
        float *t = values;
        values = new_values;
        new_values = t;
    }
    /* slice */
    float alpha = 1.0f / (1 + powf(2, -d));
    for (int i = 0; i < N; ++i) {
        for (int k = 0; k < vs; ++k) out[(size_t)i * vs + k] = 0;
        for (int j = 0; j <= d; ++j) {
            int o = L->offset[(size_t)i * (d + 1) + j] + 1;
            float w = L->barycentric[(size_t)i * (d + 1) + j];
            for (int k = 0; k < vs; ++k) out[(size_t)i * vs + k] += w * values[(size_t)o * vs + k] * alpha;
        }
    }
    free(values);
    free(new_values);
}

/* ---- dense kernel with NORMALIZE_SYMMETRIC ---- */
typedef struct {
    lattice_t L;
    float *norm; /* [N] */
} dkernel_t;

static void dkernel_init(dkernel_t *K, const float *feature, int N, int d) {
    lattice_init(&K->L, feature, N, d);
    K->norm = (float *)malloc(sizeof(float) * N);
    float *ones = (float *)malloc(sizeof(float) * N);
    for (int i = 0; i < N; ++i) ones[i] = 1.f;
    lattice_compute(&K->L, K->norm, ones, 1);
    for (int i = 0; i < N; ++i) K->norm[i] = 1.0 / sqrt(K->norm[i] + 1e-20);
    free(ones);
}
static void dkernel_free(dkernel_t *K) {
    lattice_free(&K->L);
    free(K->norm);
}
/* out = norm * Lattice(norm * in) */
static void dkernel_apply(const dkernel_t *K, float *out, const float *in, int vs, float *tmp) {
    const int N = K->L.N;
    for (int i = 0; i < N; ++i)
        for (int k = 0; k < vs; ++k) tmp[(size_t)i * vs + k] = in[(size_t)i * vs + k] * K->norm[i];
    lattice_compute(&K->L, out, tmp, vs);
    for (int i = 0; i < N; ++i)
        for (int k = 0; k < vs; ++k) out[(size_t)i * vs + k] *= K->norm[i];
}

static void exp_and_normalize(float *out, const float *in, int N, int M) {
    for (int i = 0; i < N; ++i) {
        const float *b = in + (size_t)i * M;
        float *o = out + (size_t)i * M;
        float mx = b[0];
        for (int k = 1; k < M; ++k)
            if (b[k] > mx) mx = b[k];
        float s = 0;
        for (int k = 0; k < M; ++k) {
            o[k] = expf(b[k] - mx);
            s += o[k];
        }
        for (int k = 0; k < M; ++k) o[k] = o[k] / s;
    }
}

/*
 * One image.  rgb: uint8 [H][W][3]; unary: float [M][H*W] (= -log p, class-major as
 * unary_from_softmax returns); q_out: float [M][H*W] or NULL; argmax_out: int32 [H*W] or
 * NULL; lattice_sizes: int[2] (Gaussian, bilateral vertex counts) or NULL.
 * Pairwise terms with compat == 0 AND sxy <= 0 are skipped entirely.
 * Returns 0.
 */
int densecrf_ref_inference(const uint8_t *rgb, int H, int W, const float *unary, int M, float g_sxy,
                           float g_compat, float bi_sxy, float bi_srgb, float bi_compat, int n_iters,
                           float *q_out, int32_t *argmax_out, int *lattice_sizes) {
    const int N = H * W;
    dkernel_t KG, KB;
    int use_g = g_sxy > 0, use_b = bi_sxy > 0 && bi_srgb > 0;
    if (use_g) { /* DenseCRF2D::addPairwiseGaussian */
        float *f = (float *)malloc(sizeof(float) * (size_t)N * 2);
        for (int j = 0; j < H; ++j)
            for (int i = 0; i < W; ++i) {
                f[(size_t)(j * W + i) * 2 + 0] = i / g_sxy;
                f[(size_t)(j * W + i) * 2 + 1] = j / g_sxy;
            }
        dkernel_init(&KG, f, N, 2);
        free(f);
    }
    if (use_b) { /* DenseCRF2D::addPairwiseBilateral */
        float *f = (float *)malloc(sizeof(float) * (size_t)N * 5);
        for (int j = 0; j < H; ++j)
            for (int i = 0; i < W; ++i) {
                size_t p = (size_t)(j * W + i);
                f[p * 5 + 0] = i / bi_sxy;
                f[p * 5 + 1] = j / bi_sxy;
                f[p * 5 + 2] = rgb[p * 3 + 0] / bi_srgb;
                f[p * 5 + 3] = rgb[p * 3 + 1] / bi_srgb;
                f[p * 5 + 4] = rgb[p * 3 + 2] / bi_srgb;
            }
        dkernel_init(&KB, f, N, 5);
        free(f);
    }
    if (lattice_sizes) {
        lattice_sizes[0] = use_g ? KG.L.M : 0;
        lattice_sizes[1] = use_b ? KB.L.M : 0;
    }
    size_t sz = (size_t)N * M;
    float *U = (float *)malloc(sizeof(float) * sz);    /* pixel-major */
    float *Q = (float *)malloc(sizeof(float) * sz);
    float *tmp1 = (float *)malloc(sizeof(float) * sz);
    float *tmp2 = (float *)malloc(sizeof(float) * sz);
    float *scratch = (float *)malloc(sizeof(float) * sz);
    for (int m = 0; m < M; ++m)
        for (int i = 0; i < N; ++i) U[(size_t)i * M + m] = unary[(size_t)m * N + i];

    /* DenseCRF::inference */
    for (size_t i = 0; i < sz; ++i) tmp1[i] = -U[i];
    exp_and_normalize(Q, tmp1, N, M);
    for (int it = 0; it < n_iters; ++it) {
        for (size_t i = 0; i < sz; ++i) tmp1[i] = -U[i];
        if (use_g) {
            dkernel_apply(&KG, tmp2, Q, M, scratch);
            /* PottsCompatibility::apply: out = -w * Q ; tmp1 -= out */
            for (size_t i = 0; i < sz; ++i) tmp1[i] -= -g_compat * tmp2[i];
        }
        if (use_b) {
            dkernel_apply(&KB, tmp2, Q, M, scratch);
            for (size_t i = 0; i < sz; ++i) tmp1[i] -= -bi_compat * tmp2[i];
        }
        exp_and_normalize(Q, tmp1, N, M);
    }
    if (q_out)
        for (int m = 0; m < M; ++m)
            for (int i = 0; i < N; ++i) q_out[(size_t)m * N + i] = Q[(size_t)i * M + m];
    if (argmax_out)
        for (int i = 0; i < N; ++i) {
            int best = 0;
            for (int m = 1; m < M; ++m)
                if (Q[(size_t)i * M + m] > Q[(size_t)i * M + best]) best = m;
            argmax_out[i] = best;
        }
    free(U); free(Q); free(tmp1); free(tmp2); free(scratch);
    if (use_g) dkernel_free(&KG);
    if (use_b) dkernel_free(&KB);
    return 0;
}

/* Standalone lattice filter for tests: out = Lattice(in) without normalisation.
 * feature [N][d], in/out [N][vs]. Returns the vertex count. */
int densecrf_ref_lattice_filter(const float *feature, int N, int d, const float *in, float *out, int vs) {
    lattice_t L;
    lattice_init(&L, feature, N, d);
    lattice_compute(&L, out, in, vs);
    int M = L.M;
    lattice_free(&L);
    return M;
}
