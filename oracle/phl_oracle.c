/*
 * phl_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-thread CPU restatement of the reference's permutohedral
 * lattice filter (mfinzi/depth-estimation, crf/lattice/lite/permutohedral.h).
 * It exists to CHECK the HIP product path (tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg).  Nothing under depth-estimation_amd/ may
 * import, link or call it.
 *
 * Parity status: PINNED.  oracle/gen_golden.py checks this file bit-for-bit
 * (keys, vertex count, replay offsets, weights, post-splat / post-blur vertex
 * values and the output) against oracle/_ref/libphl_ref.so, which is the
 * reference's own engine compiled from /root/reference (oracle/build_ref.sh),
 * and commits the resulting vectors under tests/golden/.
 *
 * Structure differs from the reference on purpose (init-once / filter-many,
 * precomputed blur-neighbour table, int32 arithmetic with an explicit int16
 * range check); arithmetic and its ORDER are the reference's, cited per
 * function, so that fp32 results are bit-identical on an IEEE host compiled
 * without FMA contraction.
 *
 * The four semantic deltas of this reference versus upstream Adams et al.
 *  are all reproduced:
 *   (1) no homogeneous channel, slice does not normalise   (:265-276)
 *   (2) blur weights doubled: 2*(1/4, 1/2, 1/4)            (:526)
 *   (3) slice divides every term by (1 + 2^-d)             (:480)
 *   (4) `sum *= 1/(d+1)` through float instead of `/=`     (:403)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#define PHLO_MAX_D 64

typedef struct phlo_lattice {
    int64_t n;       /* pixels */
    int d;           /* feature dimensions */
    int64_t M;       /* occupied lattice vertices, numbered in first-touch (insertion) order */
    int16_t *keys;   /* [M][d]   first d of the d+1 lattice coordinates (:39, :446-447) */
    int32_t *rvid;   /* [n][d+1] vertex id per (pixel, remainder)  == ReplayEntry.offset / vd (:458) */
    float *rw;       /* [n][d+1] barycentric weight                == ReplayEntry.weight      (:459) */
    int32_t *nbr;    /* [d+1][M][2] blur neighbours (vm1, vp1) per axis, -1 = absent (:504-522) */
    int status;      /* 0 ok, 1 = a lattice coordinate left the int16 range of the reference's keys */
    /* open-addressing table: slot -> vertex id */
    int32_t *slots;
    uint64_t cap;    /* power of two */
    int faithful;    /* reproduce the reference's stale-slot-after-grow defect (see table_lookup) */
} phlo_lattice;

/* ---- hash table over int16[d] keys -------------------------------------------------------
 * permutohedral.h:29-169: vertices are stored densely in insertion order and found through an
 * open-addressing table (linear probe, capacity 2^15 doubling when filled >= capacity/2 - 1,
 * hash k = (k + key[i]) * 2531011 in size_t).  Only "vertex id == insertion rank" is
 * observable -- with ONE exception, which is a latent defect of the reference:
 *
 *   lookup() computes h = hash(k) % capacity (:102) and THEN lookupOffset() may grow() the
 *   table (:62) and keeps probing from the stale h (old capacity).  The one key being looked
 *   up at each doubling is therefore probed/filed from the wrong slot whenever bit
 *   log2(old capacity) of its hash is set; it is not found (or later lookups do not find it),
 *   so the reference ends up with two vertices carrying the same key, one of them invisible
 *   to blur()'s neighbour lookups until the next doubling re-files it.  At most one such
 *   orphan per doubling, first doubling at M = 16383.
 *
 * faithful != 0 reproduces that behaviour exactly (used ONLY to pin this oracle bit-for-bit
 * against oracle/_ref at sizes that grow the table).  faithful == 0 (the default everywhere
 * else, and what the HIP path is compared against) recomputes h after growing, i.e. the
 * algorithm as written minus the stale-slot defect.  For M < 16383 both modes coincide. */
static uint64_t key_hash(const int16_t *k, int d)
{
    uint64_t h = 0;
    for (int i = 0; i < d; i++) {
        h += (uint64_t)(int64_t)k[i];
        h *= 2531011;
    }
    return h;
}

static void table_grow(phlo_lattice *L)
{
    uint64_t oldcap = L->cap;
    int32_t *old = L->slots;
    L->cap = oldcap * 2;
    L->slots = (int32_t *)malloc(sizeof(int32_t) * L->cap);
    memset(L->slots, 0xFF, sizeof(int32_t) * L->cap);
    for (uint64_t i = 0; i < oldcap; i++) { /* slot order, as grow() :144-152 */
        if (old[i] < 0) continue;
        uint64_t h = key_hash(L->keys + (int64_t)old[i] * L->d, L->d) % L->cap;
        while (L->slots[h] >= 0) { h++; if (h == L->cap) h = 0; }
        L->slots[h] = old[i];
    }
    free(old);
}

/* returns vertex id or -1; create appends a new vertex (lookup :101-106, lookupOffset :59-91) */
static int32_t table_lookup(phlo_lattice *L, const int16_t *key, int create)
{
    const int d = L->d;
    uint64_t h = key_hash(key, d) % L->cap;
    if ((uint64_t)L->M >= (L->cap / 2) - 1) {
        table_grow(L);
        if (!L->faithful) h = key_hash(key, d) % L->cap;
    }
    for (;;) {
        int32_t v = L->slots[h];
        if (v < 0) {
            if (!create) return -1;
            memcpy(L->keys + L->M * d, key, sizeof(int16_t) * d);
            L->slots[h] = (int32_t)L->M;
            return (int32_t)(L->M++);
        }
        if (memcmp(L->keys + (int64_t)v * d, key, sizeof(int16_t) * d) == 0) return v;
        h++;
        if (h == L->cap) h = 0;
    }
}

/* scaleFactor, PermutohedralLattice ctor :354-371 (same float expression order). */
void phlo_scale_factors(int d, float *sf)
{
    for (int i = 0; i < d; i++) {
        sf[i] = 1.0f / (sqrtf((float)(i + 1) * (i + 2)));
        sf[i] *= (d + 1) * sqrtf(2.0 / 3);
    }
}

/* Per-pixel simplex: elevate (:380-384), nearest remainder-0 point (:392-403),
 * rank (:407-411), hyperplane fix-up (:413-433), barycentric weights (:436-441).
 * Outputs greedy[d+1] (int32), rank[d+1], bary[d+2]. */
static void simplex_of(const float *position, int d, const float *sf,
                       int32_t *greedy, int *rank, float *bary)
{
    float elevated[PHLO_MAX_D + 1];
    elevated[d] = -d * position[d - 1] * sf[d - 1];
    for (int i = d - 1; i > 0; i--)
        elevated[i] = (elevated[i + 1] - i * position[i - 1] * sf[i - 1] + (i + 2) * position[i] * sf[i]);
    elevated[0] = elevated[1] + 2 * position[0] * sf[0];

    float scale = 1.0f / (d + 1);
    int sum = 0;
    for (int i = 0; i <= d; i++) {
        float v = elevated[i] * scale;
        float up = ceilf(v) * (d + 1);
        float down = floorf(v) * (d + 1);
        if (up - elevated[i] < elevated[i] - down) greedy[i] = (int32_t)up;
        else greedy[i] = (int32_t)down;
        sum += greedy[i];
    }
    sum *= scale; /* delta (4): int -> float multiply -> truncate (:403) */

    for (int i = 0; i <= d; i++) rank[i] = 0;
    for (int i = 0; i < d; i++)
        for (int j = i + 1; j <= d; j++) {
            if (elevated[i] - greedy[i] < elevated[j] - greedy[j]) rank[i]++;
            else rank[j]++;
        }

    if (sum > 0) {
        for (int i = 0; i <= d; i++) {
            if (rank[i] >= d + 1 - sum) { greedy[i] -= d + 1; rank[i] += sum - (d + 1); }
            else rank[i] += sum;
        }
    } else if (sum < 0) {
        for (int i = 0; i <= d; i++) {
            if (rank[i] < -sum) { greedy[i] += d + 1; rank[i] += (d + 1) + sum; }
            else rank[i] += sum;
        }
    }

    for (int i = 0; i <= d + 1; i++) bary[i] = 0.0f;
    for (int i = 0; i <= d; i++) {
        bary[d - rank[i]] += (elevated[i] - greedy[i]) * scale;
        bary[d + 1 - rank[i]] -= (elevated[i] - greedy[i]) * scale;
    }
    bary[0] += 1.0f + bary[d + 1];
}

/* blur-neighbour table for all current vertices (:502-522) */
static void compute_neighbors(phlo_lattice *L)
{
    const int d = L->d;
    int64_t M = L->M;
    free(L->nbr);
    L->nbr = (int32_t *)malloc(sizeof(int32_t) * (M > 0 ? M : 1) * (d + 1) * 2);
    int16_t n1[PHLO_MAX_D + 1], n2[PHLO_MAX_D + 1];
    for (int j = 0; j <= d; j++) {
        for (int64_t v = 0; v < M; v++) {
            const int16_t *k = L->keys + v * d;
            for (int i = 0; i < d; i++) { n1[i] = k[i] + 1; n2[i] = k[i] - 1; }
            /* for j == d the written coordinate is the implied (d+1)-th one, outside the
             * hashed key, so the neighbour is key +- 1 in all d stored coordinates (:508-509) */
            if (j < d) { n1[j] = k[j] - d; n2[j] = k[j] + d; }
            L->nbr[((int64_t)j * M + v) * 2 + 0] = table_lookup(L, n1, 0);
            L->nbr[((int64_t)j * M + v) * 2 + 1] = table_lookup(L, n2, 0);
        }
    }
}

/* Build: the geometry half of splat() for every pixel in order (:376-462 minus the
 * value accumulate), then the blur-neighbour table (:502-522).
 * ref is [n][d] with element strides (rs, cs) so that NCHW views work like the
 * reference's accessor copy (:221-226). */
phlo_lattice *phlo_build(const float *ref, int64_t n, int d, int64_t rs, int64_t cs, int faithful)
{
    if (d < 1 || d > PHLO_MAX_D || n < 0) return NULL;
    phlo_lattice *L = (phlo_lattice *)calloc(1, sizeof(*L));
    L->n = n;
    L->d = d;
    int64_t cap_rows = n * (d + 1) > 0 ? n * (d + 1) : 1;
    L->keys = (int16_t *)malloc(sizeof(int16_t) * cap_rows * d);
    L->rvid = (int32_t *)malloc(sizeof(int32_t) * cap_rows);
    L->rw = (float *)malloc(sizeof(float) * cap_rows);
    L->faithful = faithful;
    L->cap = 1 << 15; /* :36 */
    L->slots = (int32_t *)malloc(sizeof(int32_t) * L->cap);
    memset(L->slots, 0xFF, sizeof(int32_t) * L->cap);

    float sf[PHLO_MAX_D];
    phlo_scale_factors(d, sf);
    float position[PHLO_MAX_D];
    int32_t greedy[PHLO_MAX_D + 1];
    int rank[PHLO_MAX_D + 1];
    float bary[PHLO_MAX_D + 2];
    int16_t key[PHLO_MAX_D];

    for (int64_t p = 0; p < n; p++) {
        for (int c = 0; c < d; c++) position[c] = ref[p * rs + c * cs];
        simplex_of(position, d, sf, greedy, rank, bary);
        for (int r = 0; r <= d; r++) {
            for (int i = 0; i < d; i++) {
                /* canonical[r*(d+1)+k] = r for k <= d-r, else r-(d+1)  (:346-351) */
                int32_t c = greedy[i] + (rank[i] <= d - r ? r : r - (d + 1));
                if (c < -32768 || c > 32767) L->status = 1;
                key[i] = (int16_t)c;
            }
            L->rvid[p * (d + 1) + r] = table_lookup(L, key, 1);
            L->rw[p * (d + 1) + r] = bary[r];
        }
    }

    compute_neighbors(L);
    return L;
}

/* Row-band support (tests of the multi-GPU decomposition): append vertices owned by a
 * neighbouring band as ghosts; vid_out[i] = local id of key i.  Mirrors phl_add_vertices. */
void phlo_add_vertices(phlo_lattice *L, const int16_t *keys, int64_t count, int32_t *vid_out)
{
    const int d = L->d;
    L->keys = (int16_t *)realloc(L->keys, sizeof(int16_t) * (size_t)(L->M + count + 1) * d);
    for (int64_t i = 0; i < count; i++) vid_out[i] = table_lookup(L, keys + i * d, 1);
    compute_neighbors(L);
}

void phlo_free(phlo_lattice *L)
{
    if (!L) return;
    free(L->keys); free(L->rvid); free(L->rw); free(L->nbr); free(L->slots); free(L);
}

int64_t phlo_num_vertices(const phlo_lattice *L) { return L->M; }
int phlo_status(const phlo_lattice *L) { return L->status; }
void phlo_get_keys(const phlo_lattice *L, int16_t *out) { memcpy(out, L->keys, sizeof(int16_t) * L->M * L->d); }
void phlo_get_replay(const phlo_lattice *L, int32_t *vid, float *w)
{
    memcpy(vid, L->rvid, sizeof(int32_t) * L->n * (L->d + 1));
    memcpy(w, L->rw, sizeof(float) * L->n * (L->d + 1));
}
void phlo_get_neighbors(const phlo_lattice *L, int32_t *out) { memcpy(out, L->nbr, sizeof(int32_t) * L->M * (L->d + 1) * 2); }

static double now_s(void)
{
    struct timeval t;
    gettimeofday(&t, NULL);
    return t.tv_sec + t.tv_usec * 1e-6;
}

/* value half of splat(): vert[M][vd] = sum over pixels in order of w*src (:236-238, :454-455) */
void phlo_splat(const phlo_lattice *L, const float *src, int vd, int64_t s_rs, int64_t s_cs, float *vert)
{
    const int d = L->d;
    memset(vert, 0, sizeof(float) * (size_t)L->M * vd);
    for (int64_t p = 0; p < L->n; p++)
        for (int r = 0; r <= d; r++) {
            float *v = vert + (int64_t)L->rvid[p * (d + 1) + r] * vd;
            float w = L->rw[p * (d + 1) + r];
            for (int c = 0; c < vd; c++) v[c] += (w * src[p * s_rs + c * s_cs]);
        }
}

/* blur() (:486-548): axes 0..d, Jacobi ping-pong, absent neighbour = 0; in place on vert */
void phlo_blur(const phlo_lattice *L, float *vert, int vd)
{
    const int d = L->d;
    const int64_t M = L->M;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)(M > 0 ? M : 1) * vd);
    float *oldv = vert, *newv = tmp;
    for (int j = 0; j <= d; j++) {
        for (int64_t v = 0; v < M; v++) {
            int32_t a = L->nbr[((int64_t)j * M + v) * 2 + 0];
            int32_t b = L->nbr[((int64_t)j * M + v) * 2 + 1];
            const float *o = oldv + v * vd;
            float *w = newv + v * vd;
            for (int c = 0; c < vd; c++) {
                float vm1 = a >= 0 ? oldv[(int64_t)a * vd + c] : 0.0f;
                float vp1 = b >= 0 ? oldv[(int64_t)b * vd + c] : 0.0f;
                w[c] = 2 * (0.25f * vm1 + 0.5f * o[c] + 0.25f * vp1);
            }
        }
        float *t = newv; newv = oldv; oldv = t;
    }
    if (oldv != vert) memcpy(vert, oldv, sizeof(float) * (size_t)M * vd);
    free(tmp);
}

/* slice() (:473-483): out[p] = sum_r w_r * vert[v_r] / (1 + 2^-d), term by term */
void phlo_slice(const phlo_lattice *L, const float *vert, int vd, float *out, int64_t o_rs, int64_t o_cs)
{
    const int d = L->d;
    float col[4096];
    float *colp = vd <= 4096 ? col : (float *)malloc(sizeof(float) * vd);
    for (int64_t p = 0; p < L->n; p++) {
        for (int c = 0; c < vd; c++) colp[c] = 0;
        for (int r = 0; r <= d; r++) {
            const float *v = vert + (int64_t)L->rvid[p * (d + 1) + r] * vd;
            float w = L->rw[p * (d + 1) + r];
            for (int c = 0; c < vd; c++) colp[c] += w * v[c] / (1 + powf(2, -d));
        }
        for (int c = 0; c < vd; c++) out[p * o_rs + c * o_cs] = colp[c];
    }
    if (colp != col) free(colp);
}

/* splat -> blur -> slice.  src/out are [n][vd] with element strides; dumps (may be NULL) are
 * dense [M][vd].  t_stage (may be NULL) receives seconds for {splat, blur, slice}. */
int phlo_filter(const phlo_lattice *L, const float *src, int vd, int64_t s_rs, int64_t s_cs,
                float *out, int64_t o_rs, int64_t o_cs,
                float *splat_dump, float *blur_dump, double *t_stage)
{
    const int64_t M = L->M;
    if (vd < 1) return 2;
    float *val = (float *)malloc(sizeof(float) * (size_t)(M > 0 ? M : 1) * vd);
    double t0 = now_s();
    phlo_splat(L, src, vd, s_rs, s_cs, val);
    if (splat_dump) memcpy(splat_dump, val, sizeof(float) * M * vd);
    double t1 = now_s();
    phlo_blur(L, val, vd);
    if (blur_dump) memcpy(blur_dump, val, sizeof(float) * M * vd);
    double t2 = now_s();
    phlo_slice(L, val, vd, out, o_rs, o_cs);
    double t3 = now_s();
    if (t_stage) { t_stage[0] = t1 - t0; t_stage[1] = t2 - t1; t_stage[2] = t3 - t2; }
    free(val);
    return L->status;
}

/* One-shot convenience == reference lattice.filter(src, ref) (lattice.cpp:6-10): rebuilds
 * the lattice on every call, as the reference does. */
int phlo_filter_once(const float *ref, const float *src, int64_t n, int d, int vd, float *out)
{
    phlo_lattice *L = phlo_build(ref, n, d, d, 1, 0);
    if (!L) return 3;
    int st = phlo_filter(L, src, vd, vd, 1, out, vd, 1, NULL, NULL, NULL);
    phlo_free(L);
    return st;
}
