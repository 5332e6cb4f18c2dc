// phl_build.hip -- lattice construction on gfx950 (runs once per `ref`, then cached).
//
// Replaces, for all n pixels at once, the geometry half of the reference's sequential
// splat() loop (crf/lattice/lite/permutohedral.h:376-447,458-460), its insertion-ordered hash
// table (:29-169) and the per-axis neighbour lookups of blur() (:504-522).
//
// Pipeline (one kernel each, all on the caller's stream):
//   elevate    per pixel: elevate to H_d, nearest remainder-0 point, rank, barycentric
//              weights -> a record the d+1 candidate keys are rebuilt from + weights (:380-447)
//   insert     lock-free open addressing: slot <- atomicCAS(EMPTY, e), equal keys fold to the
//              MINIMUM candidate index with atomicMin  => deterministic representative;
//              runs of equal keys along a wavefront (neighbouring pixels) probe once
//   flag+scan  representative candidates as a bit mask, scan of its word counts => vertex id =
//              first-touch rank, i.e. exactly the reference's insertion order (:70-77)
//   assign     vertex keys [M][d], first touches; table now maps key -> vertex id
//   final_vid  (from phl_tiles_build, once the locality numbering is known) replay[].vid, written once:
//              candidate -> table slot -> clean vertex -> [reference vertex] -> row
//   (on demand) count/scan/fill/sort   transpose of the replay matrix: per vertex the (pixel,
//              weight) list in ascending pixel order for the reference-exact gather splat
//              (phl_ensure_csr); the default chunk kernels do not need it
//   neighbors  [d+1][M][2] blur neighbour ids, -1 where the vertex does not exist (:516-522)
//
// The whole translation unit is compiled with -ffp-contract=off: elevate must round exactly
// like the reference's scalar C++ (mul and add separately) so that keys, ranks and weights
// are bit-identical to the CPU path.
#include <cstring>
#include <mutex>

#include "phl_device_utils.h"

namespace {

struct sf_t {
    float v[PHL_MAX_D];
};

// ------------------------------------------------------------------------------------------
// Per-pixel record the candidate keys are rebuilt from: the first D coordinates of the rounded point (`greedy`, as
// shorts: every key coordinate is one of them plus a canonical offset, and r = 0 has offset 0, so they are in range
// whenever the keys are) and their ranks (0..D, five bits each, six to a word).
template <int D>
struct pix_rec {
    static constexpr int GW = (D + 1) / 2;            // words of packed shorts
    static constexpr int W = GW + (D + 5) / 6;        // + words of packed ranks
    static __device__ __forceinline__ void load(const uint32_t *__restrict__ p, uint32_t (&w)[W])
    {
        if constexpr (W == 4) {
            const uint4 v = *reinterpret_cast<const uint4 *>(p);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else if constexpr (W == 2) {
            const uint2 v = *reinterpret_cast<const uint2 *>(p);
            w[0] = v.x; w[1] = v.y;
        } else {
#pragma unroll
            for (int j = 0; j < W; j++) w[j] = p[j];
        }
    }
    static __device__ __forceinline__ void store(uint32_t *__restrict__ p, const uint32_t (&w)[W])
    {
        if constexpr (W == 4) *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
        else if constexpr (W == 2) *reinterpret_cast<uint2 *>(p) = make_uint2(w[0], w[1]);
        else {
#pragma unroll
            for (int j = 0; j < W; j++) p[j] = w[j];
        }
    }
    // candidate key r of the pixel, as GW words of packed shorts (the unused half of the last word is zero)
    static __device__ __forceinline__ void key(const uint32_t (&w)[W], int r, uint32_t (&k)[GW])
    {
#pragma unroll
        for (int j = 0; j < GW; j++) k[j] = 0u;
#pragma unroll
        for (int i = 0; i < D; i++) {
            const int g = (int)(int16_t)(uint16_t)(w[i >> 1] >> (16 * (i & 1)));
            const int rk = (int)((w[GW + i / 6] >> (5 * (i % 6))) & 31u);
            const int c = g + (rk <= D - r ? r : r - (D + 1));
            k[i >> 1] |= (uint32_t)(uint16_t)(int16_t)c << (16 * (i & 1));
        }
    }
};

// ------------------------------------------------------------------------------------------
// elevate: one thread per pixel, everything in registers (D is a template parameter so that
// all loops unroll and no array is runtime-indexed).
template <int D>
__global__ __launch_bounds__(256) void k_elevate(const float *__restrict__ ref, int64_t rs, int64_t cs, int64_t n,
                                                 sf_t sf, uint32_t *__restrict__ recs,
                                                 phl_replay_t *__restrict__ replay, int *__restrict__ err,
                                                 float *__restrict__ mm /* [grid][D][2] feature ranges of the block */)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = p < n;
    float pos[D];
#pragma unroll
    for (int i = 0; i < D; i++) pos[i] = live ? ref[p * rs + i * cs] : 0.f;
    {   // feature ranges (the chunk grid of phl_tiles_build is laid over the two widest features): the values are in
        // registers here, a second pass over `ref` would read them again.  Wavefront reduction on DPP row shifts and
        // broadcasts (VALU only); lane 63 holds the result.
        __shared__ float smin[4][D], smax[4][D];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#define PHL_DPP_F(x, ctrl, rowmask, OP, IDENT) x = OP(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(IDENT), __float_as_int(x), ctrl, rowmask, 0xF, false)))
#define PHL_WAVE_REDUCE(x, OP, IDENT)                                                                              \
    PHL_DPP_F(x, 0x111, 0xF, OP, IDENT); PHL_DPP_F(x, 0x112, 0xF, OP, IDENT); PHL_DPP_F(x, 0x114, 0xF, OP, IDENT); \
    PHL_DPP_F(x, 0x118, 0xF, OP, IDENT); PHL_DPP_F(x, 0x142, 0xA, OP, IDENT); PHL_DPP_F(x, 0x143, 0xC, OP, IDENT);
#pragma unroll
        for (int i = 0; i < D; i++) {
            float a = live ? pos[i] : INFINITY, b = live ? pos[i] : -INFINITY;
            PHL_WAVE_REDUCE(a, fminf, INFINITY)
            PHL_WAVE_REDUCE(b, fmaxf, -INFINITY)
            if (lane == 63) { smin[w][i] = a; smax[w][i] = b; }
        }
#undef PHL_WAVE_REDUCE
#undef PHL_DPP_F
        __syncthreads();
        if ((int)threadIdx.x < D) {
            float a = smin[0][threadIdx.x], b = smax[0][threadIdx.x];
#pragma unroll
            for (int k = 1; k < 4; k++) { a = fminf(a, smin[k][threadIdx.x]); b = fmaxf(b, smax[k][threadIdx.x]); }
            mm[((int64_t)blockIdx.x * D + threadIdx.x) * 2 + 0] = a;
            mm[((int64_t)blockIdx.x * D + threadIdx.x) * 2 + 1] = b;
        }
    }
    if (!live) return;

    // permutohedral.h:380-384 (expression order kept)
    float el[D + 1];
    el[D] = (float)(-D) * pos[D - 1] * sf.v[D - 1];
#pragma unroll
    for (int i = D - 1; i > 0; i--)
        el[i] = (el[i + 1] - (float)i * pos[i - 1] * sf.v[i - 1]) + (float)(i + 2) * pos[i] * sf.v[i];
    el[0] = el[1] + 2.0f * pos[0] * sf.v[0];

    // :387-403
    const float scale = 1.0f / (float)(D + 1);
    int greedy[D + 1];
    int sum = 0;
#pragma unroll
    for (int i = 0; i <= D; i++) {
        float v = el[i] * scale;
        float up = ceilf(v) * (float)(D + 1);
        float down = floorf(v) * (float)(D + 1);
        greedy[i] = (up - el[i] < el[i] - down) ? (int)up : (int)down;
        sum += greedy[i];
    }
    sum = (int)((float)sum * scale);

    // :407-411  (ties go to the later index)
    int rank[D + 1];
#pragma unroll
    for (int i = 0; i <= D; i++) rank[i] = 0;
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
        for (int j = i + 1; j <= D; j++) {
            bool lt = (el[i] - (float)greedy[i]) < (el[j] - (float)greedy[j]);
            rank[i] += lt ? 1 : 0;
            rank[j] += lt ? 0 : 1;
        }

    // :413-433
    if (sum > 0) {
#pragma unroll
        for (int i = 0; i <= D; i++) {
            if (rank[i] >= D + 1 - sum) { greedy[i] -= D + 1; rank[i] += sum - (D + 1); }
            else rank[i] += sum;
        }
    } else if (sum < 0) {
#pragma unroll
        for (int i = 0; i <= D; i++) {
            if (rank[i] < -sum) { greedy[i] += D + 1; rank[i] += (D + 1) + sum; }
            else rank[i] += sum;
        }
    }

    // :436-441  barycentric; selects instead of runtime-indexed stores keep it in registers
    float bary[D + 2];
#pragma unroll
    for (int k = 0; k <= D + 1; k++) bary[k] = 0.0f;
#pragma unroll
    for (int i = 0; i <= D; i++) {
        float t = (el[i] - (float)greedy[i]) * scale;
#pragma unroll
        for (int k = 0; k <= D + 1; k++) {
            if (k == D - rank[i]) bary[k] = bary[k] + t;
            if (k == D + 1 - rank[i]) bary[k] = bary[k] - t;
        }
    }
    bary[0] = bary[0] + (1.0f + bary[D + 1]);

    // :444-447, :458-460.  The d+1 candidate keys of the pixel differ only by the canonical offsets (:346-351):
    // key_r[i] = greedy[i] + (rank[i] <= D - r ? r : r - (D+1)).  Stored once per pixel as a record (pix_rec<D>)
    // instead of d+1 keys of d shorts each: 16 bytes instead of 60 at d = 5.
    bool bad = false;
#pragma unroll
    for (int r = 0; r <= D; r++)
#pragma unroll
        for (int i = 0; i < D; i++) {
            const int c = greedy[i] + (rank[i] <= D - r ? r : r - (D + 1));
            bad |= (c < -32768) | (c > 32767);
        }
    uint32_t rw[pix_rec<D>::W];
#pragma unroll
    for (int j = 0; j < pix_rec<D>::W; j++) rw[j] = 0u;
#pragma unroll
    for (int i = 0; i < D; i++) {
        rw[i >> 1] |= (uint32_t)(uint16_t)(int16_t)greedy[i] << (16 * (i & 1));
        rw[pix_rec<D>::GW + i / 6] |= (uint32_t)rank[i] << (5 * (i % 6));
    }
    pix_rec<D>::store(recs + p * pix_rec<D>::W, rw);
    // whole entries (vertex id 0 until k_final_vid writes it): full 16-byte stores instead of every other word of the rows
    phl_replay_t *rout = replay + p * (D + 1);
    if constexpr ((D + 1) % 2 == 0) {
#pragma unroll
        for (int r = 0; r <= D; r += 2)
            reinterpret_cast<uint4 *>(rout)[r >> 1] = make_uint4(0u, __float_as_uint(bary[r]), 0u, __float_as_uint(bary[r + 1]));
    } else {
#pragma unroll
        for (int r = 0; r <= D; r++) reinterpret_cast<uint2 *>(rout)[r] = make_uint2(0u, __float_as_uint(bary[r]));
    }
    if (bad) atomicOr(err, 1);
}

// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix_begin() { return 0x811C9DC5u; }
__device__ __forceinline__ uint32_t mix_step(uint32_t h, int c) { return (h ^ (uint32_t)(uint16_t)c) * 0x01000193u; }
__device__ __forceinline__ uint32_t mix_end(uint32_t h)
{
    h ^= h >> 15; h *= 0x2C1B3C6Du;
    h ^= h >> 12; h *= 0x297A2D39u;
    h ^= h >> 15;
    return h;
}

// One thread per candidate (pixel, remainder).  table[slot] ends up holding the SMALLEST
// candidate index among all candidates with that key.  The records were written by the previous
// launch, and the index read back from the atomic is always a candidate of a pixel whose record is
// there, so no in-kernel hand-off of plain data is needed.
template <int D>
__global__ __launch_bounds__(256) void k_insert(const uint32_t *__restrict__ recs, int n, int *table,
                                                uint32_t mask, int *__restrict__ slot_of, int max_probe, int *__restrict__ err)
{
    // A wavefront takes 64 CONSECUTIVE pixels of ONE remainder: neighbouring pixels mostly lie
    // in the same simplex, so the same key repeats along the lanes and only the first lane of
    // each run probes the table (its candidate index is the smallest of the run, which is what
    // the atomicMin wants anyway).
    using R = pix_rec<D>;
    constexpr int dp1 = D + 1;
    const int lane = threadIdx.x & 63;
    // the small table has been found too small: the host will repeat the insertion, nothing of this launch is kept
    if (max_probe != 0x7FFFFFFF && (*reinterpret_cast<volatile int *>(err) & 2)) return;
    // (the wavefront's index and remainder are scalars; the hash runs over the key's packed words)
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));   // < 2^31
    const unsigned wq = wave / (unsigned)dp1;
    const int r = (int)(wave - wq * (unsigned)dp1);
    const int p = (int)(wq * 64u) + lane;                  // (n(d+1) < 2^31)
    const bool active = p < n;
    const int pc = active ? p : n - 1;
    const int e = pc * dp1 + r;
    uint32_t rec[R::W], key[R::GW];
    R::load(recs + (int64_t)pc * R::W, rec);
    R::key(rec, r, key);
    uint32_t h = mix_begin();
#pragma unroll
    for (int j = 0; j < R::GW; j++) h = (h ^ key[j]) * 0x01000193u;
    h = mix_end(h);
    // same key as the previous lane?
    bool same_prev = active && lane > 0;
#pragma unroll
    for (int j = 0; j < R::GW; j++) same_prev &= ((uint32_t)__shfl_up((int)key[j], 1) == key[j]);
    const unsigned long long heads = __ballot(active && !same_prev);
    const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    const int head_lane = 63 - __clzll(below ? below : 1ull);
    int slot = 0;
    if (active && lane == head_lane) {
        h &= mask;
        // max_probe: the table is sized for the vertex counts images have, not for the worst case (every candidate its
        // own vertex); a probe sequence this long means it is too small -- flag it, the host repeats with the full size
        for (int steps = 0;; steps++) {
            if (steps >= max_probe) {
                atomicOr(err, 2);
                break;
            }
            const int prev = atomicCAS(&table[h], PHL_EMPTY, e);
            if (prev == PHL_EMPTY) break;
            const int p2 = prev / dp1;
            uint32_t rec2[R::W], key2[R::GW];
            R::load(recs + (int64_t)p2 * R::W, rec2);
            R::key(rec2, prev - p2 * dp1, key2);
            bool same = true;
#pragma unroll
            for (int j = 0; j < R::GW; j++) same &= (key2[j] == key[j]);
            if (same) {
                if (e < prev) atomicMin(&table[h], e);
                break;
            }
            h = (h + 1) & mask;
        }
        slot = (int)h;
    }
    slot = __shfl(slot, head_lane);
    if (active) slot_of[e] = slot;
}

// First-touch flags as a bit mask: bit e is set iff candidate e is the representative (smallest index) of its key,
// i.e. iff some table slot holds e -- set from the table (one coalesced pass over the slots, a few hundred thousand
// atomics) rather than by asking for every candidate.  The vertex id of a representative is its rank among the set
// bits: the scan of the N / 64 word counts plus a popcount -- not a scan over all N candidates.
__global__ __launch_bounds__(256) void k_flag_bits(const int *__restrict__ table, int64_t cap, unsigned long long *__restrict__ bits)
{
    const int64_t s0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (s0 >= cap) return;                                 // cap is a multiple of 4
    const int4 t = *reinterpret_cast<const int4 *>(table + s0);
    const int e[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (e[j] != PHL_EMPTY) atomicOr(&bits[e[j] >> 6], 1ull << (e[j] & 63));
}

__global__ __launch_bounds__(256) void k_word_counts(const unsigned long long *__restrict__ bits, int NW, int *__restrict__ wcount)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < NW) wcount[w] = __popcll(bits[w]);
}

// One thread per table slot: the slot's representative candidate becomes a vertex.
template <int D>
__global__ __launch_bounds__(256) void k_assign(const unsigned long long *__restrict__ bits, const int *__restrict__ wrank,
                                                const uint32_t *__restrict__ recs, int64_t cap, int *__restrict__ table,
                                                int16_t *__restrict__ vkeys, int *__restrict__ vfirst)
{
    using R = pix_rec<D>;
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cap) return;
    const int e = table[s];
    if (e == PHL_EMPTY) return;
    const unsigned long long m = bits[e >> 6];
    const int vid = wrank[e >> 6] + __popcll(m & ((1ull << (e & 63)) - 1ull));
    const int p = e / (D + 1);
    uint32_t rec[R::W], key[R::GW];
    R::load(recs + (int64_t)p * R::W, rec);
    R::key(rec, e - p * (D + 1), key);
#pragma unroll
    for (int i = 0; i < D; i++) vkeys[(int64_t)vid * D + i] = (int16_t)(uint16_t)(key[i >> 1] >> (16 * (i & 1)));
    table[s] = -(vid + 1);
    vfirst[vid] = e;     // first-touch candidate of the vertex (its pixel tells the renumbering where the vertex lives)
}

// replay[e].vid, written once: clean vertex of the candidate's table slot -> reference vertex (remap; for a key with
// several vertices the last segment that starts at or before e) -> locality numbering (int_of_ft).
// (the kernel is three dependent gathers per candidate -- 96 % of its wave cycles are parked, profiles/r04h_build_pmc.json --
// so a thread walks FV_PER candidates, 256 apart, with the gathers of each level issued together)
constexpr int FV_PER = 4;
__global__ __launch_bounds__(256) void k_final_vid(const int *__restrict__ table, const int *__restrict__ slot_of, int N,
                                                   const int *__restrict__ remap, const int *__restrict__ dup_ptr,
                                                   const int *__restrict__ seg_e, const int *__restrict__ seg_id,
                                                   const int *__restrict__ int_of_ft, phl_replay_t *__restrict__ replay)
{
    const int64_t e0 = (int64_t)blockIdx.x * (256 * FV_PER) + threadIdx.x;      // (64 bits: N may come within 1024 of 2^31)
    int s[FV_PER], r[FV_PER];
#pragma unroll
    for (int j = 0; j < FV_PER; j++) {
        const int64_t e = e0 + j * 256;
        s[j] = slot_of[e < N ? e : (int64_t)N - 1];
    }
#pragma unroll
    for (int j = 0; j < FV_PER; j++) r[j] = -(table[s[j]] + 1);
    if (remap) {
#pragma unroll
        for (int j = 0; j < FV_PER; j++) r[j] = remap[r[j]];
#pragma unroll
        for (int j = 0; j < FV_PER; j++) {
            if (r[j] < 0) {
                const int64_t e = e0 + j * 256;
                const int k = -r[j] - 1;
                int id = seg_id[dup_ptr[k]];
                for (int q = dup_ptr[k]; q < dup_ptr[k + 1] && seg_e[q] <= e; q++) id = seg_id[q];
                r[j] = id;
            }
        }
    }
    if (int_of_ft) {
#pragma unroll
        for (int j = 0; j < FV_PER; j++) r[j] = int_of_ft[r[j]];
    }
#pragma unroll
    for (int j = 0; j < FV_PER; j++) {
        const int64_t e = e0 + j * 256;
        if (e < N) replay[e].vid = r[j];
    }
}

__global__ __launch_bounds__(256) void k_count_vid(const phl_replay_t *__restrict__ replay, int N, int *cnt)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < N) atomicAdd(&cnt[replay[e].vid], 1);
}

__global__ __launch_bounds__(256) void k_replay_vids(const phl_replay_t *__restrict__ replay, int N, int *__restrict__ vid)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < N) vid[e] = replay[e].vid;
}

// candidates in (vertex, candidate index) order -> (pixel, weight)
__global__ __launch_bounds__(256) void k_fill_sorted(const phl_replay_t *__restrict__ replay, const int *__restrict__ perm,
                                                     int N, int dp1, phl_contrib_t *__restrict__ csr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int e = perm[i];
    phl_contrib_t c;
    c.pixel = e / dp1;
    c.w = replay[e].w;
    csr[i] = c;
}

// Vertex keys as packed words (two shorts a word, rows padded to a vector load: 16 bytes for d = 5..8): a key
// comparison in the neighbour search is one divergent load instead of d two-byte loads.
template <int D>
struct packed_key {
    static constexpr int KW = (D + 1) / 2;
    static constexpr int PW = KW == 3 ? 4 : KW;          // row stride in words
    static __device__ __forceinline__ void load(const uint32_t *__restrict__ p, uint32_t (&w)[KW])
    {
        if constexpr (PW == 4) {
            const uint4 v = *reinterpret_cast<const uint4 *>(p);
            w[0] = v.x; w[1] = v.y; w[2] = v.z;
            if constexpr (KW == 4) w[3] = v.w;
        } else if constexpr (PW == 2) {
            const uint2 v = *reinterpret_cast<const uint2 *>(p);
            w[0] = v.x; w[1] = v.y;
        } else {
#pragma unroll
            for (int j = 0; j < KW; j++) w[j] = p[j];
        }
    }
};

template <int D>
__global__ __launch_bounds__(256) void k_pack_keys(const int16_t *__restrict__ vkeys, int M, uint32_t *__restrict__ out)
{
    using K = packed_key<D>;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= M) return;
    uint32_t w[K::PW];
#pragma unroll
    for (int j = 0; j < K::PW; j++) w[j] = 0u;
#pragma unroll
    for (int i = 0; i < D; i++) w[i >> 1] |= (uint32_t)(uint16_t)vkeys[(int64_t)v * D + i] << (16 * (i & 1));
#pragma unroll
    for (int j = 0; j < K::PW; j++) out[(int64_t)v * K::PW + j] = w[j];
}

// Thread per (axis, vertex): neighbour keys are key +- 1 in every stored coordinate, with
// coordinate `axis` set to key[axis] -+ d; for axis == d the touched coordinate is the implied
// (d+1)-th one, i.e. all d stored coordinates move by +-1 (permutohedral.h:504-509).
// The two sides' probe sequences run interleaved (two independent chains of dependent loads in flight).
template <int D>
__global__ __launch_bounds__(256) void k_neighbors(const uint32_t *__restrict__ vkp, int M,
                                                   const int *__restrict__ table, uint32_t mask,
                                                   int *__restrict__ nbr)
{
    using K = packed_key<D>;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * (D + 1)) return;
    const int axis = (int)(idx / M);
    const int v = (int)(idx - (int64_t)axis * M);
    uint32_t own[K::KW];
    K::load(vkp + (int64_t)v * K::PW, own);
    uint32_t want[2][K::KW];
    uint32_t h[2];
#pragma unroll
    for (int side = 0; side < 2; side++) {
        const int step = side == 0 ? 1 : -1;  // side 0 = neighbor1 (vm1), side 1 = neighbor2 (vp1)
#pragma unroll
        for (int j = 0; j < K::KW; j++) want[side][j] = 0u;
        uint32_t hh = mix_begin();
#pragma unroll
        for (int i = 0; i < D; i++) {
            const int k = (int)(int16_t)(uint16_t)(own[i >> 1] >> (16 * (i & 1)));
            const int c = (i == axis) ? k - step * D : k + step;
            hh = mix_step(hh, (int16_t)c);
            want[side][i >> 1] |= (uint32_t)(uint16_t)(int16_t)c << (16 * (i & 1));
        }
        h[side] = mix_end(hh) & mask;
    }
    int res[2] = {-1, -1};
    bool open[2] = {true, true};
    while (open[0] || open[1]) {
        int t[2];
#pragma unroll
        for (int side = 0; side < 2; side++) t[side] = open[side] ? table[h[side]] : PHL_EMPTY;
        uint32_t other[2][K::KW];
#pragma unroll
        for (int side = 0; side < 2; side++) {
            if (t[side] == PHL_EMPTY) { open[side] = false; continue; }
            K::load(vkp + (int64_t)(-(t[side] + 1)) * K::PW, other[side]);
        }
#pragma unroll
        for (int side = 0; side < 2; side++) {
            if (!open[side]) continue;
            bool same = true;
#pragma unroll
            for (int j = 0; j < K::KW; j++) same &= (other[side][j] == want[side][j]);
            if (same) { res[side] = -(t[side] + 1); open[side] = false; }
            else h[side] = (h[side] + 1) & mask;
        }
    }
    nbr[idx * 2 + 0] = res[0];
    nbr[idx * 2 + 1] = res[1];
}

template <int D>
int launch_neighbors(const int16_t *vkeys, int M, const int *table, uint32_t mask, int *nbr, void **scratch_out, hipStream_t st)
{
    using K = packed_key<D>;
    uint32_t *vkp;
    PHL_HIP(phl_dev_malloc((void **)&vkp, sizeof(uint32_t) * ((size_t)M * K::PW + 4)));
    *scratch_out = vkp;
    hipLaunchKernelGGL(k_pack_keys<D>, dim3((M + 255) / 256), dim3(256), 0, st, vkeys, M, vkp);
    const int64_t tot = (int64_t)M * (D + 1);
    hipLaunchKernelGGL(k_neighbors<D>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, vkp, M, table, mask, nbr);
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}

// Neighbour ids for blurring along axes a = 2p and b = 2p+1 in ONE pass (k_blur2):
//   out[v] = blur_b(blur_a(in))[v] needs in[] at  a-(u), u, a+(u)  for u in { b-(v), v, b+(v) }.
// Stored per vertex: { a-(b-), b-, a+(b-), a-(v) | a+(v), a-(b+), b+, a+(b+) }, -1 = absent.
__global__ __launch_bounds__(256) void k_compose_pairs(const int2 *__restrict__ nbr, int M, int npairs, int4 *__restrict__ nbr2)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * npairs) return;
    const int p = (int)(idx / M);
    const int v = (int)(idx - (int64_t)p * M);
    const int2 *na = nbr + (int64_t)(2 * p) * M;
    const int2 nb = nbr[(int64_t)(2 * p + 1) * M + v];
    const int2 none = make_int2(-1, -1);
    const int2 am = nb.x >= 0 ? na[nb.x] : none;
    const int2 a0 = na[v];
    const int2 ap = nb.y >= 0 ? na[nb.y] : none;
    nbr2[idx * 2 + 0] = make_int4(am.x, nb.x, am.y, a0.x);
    nbr2[idx * 2 + 1] = make_int4(a0.y, ap.x, nb.y, ap.y);
}

struct hidden_t {
    int n;
    int32_t id[PHL_MAX_HIDDEN];
};

// (re)build the persistent key -> vertex table from the distinct vertex keys (reference-table mode: the
// duplicates the reference's own table cannot reach stay out)
__global__ __launch_bounds__(256) void k_table_insert(const int16_t *__restrict__ vkeys, int d, int M, int *table,
                                                      uint32_t mask, hidden_t hidden, const int *__restrict__ int_of_ft)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= M) return;
    for (int i = 0; i < hidden.n; i++)          // hidden ids are first-touch ids
        if ((int_of_ft ? int_of_ft[hidden.id[i]] : hidden.id[i]) == v) return;
    const int16_t *key = vkeys + (int64_t)v * d;
    uint32_t h = mix_begin();
    for (int i = 0; i < d; i++) h = mix_step(h, key[i]);
    h = mix_end(h) & mask;
    while (atomicCAS(&table[h], PHL_EMPTY, -(v + 1)) != PHL_EMPTY) h = (h + 1) & mask;
}

// vid of each query key, or -1
__global__ __launch_bounds__(256) void k_table_lookup(const int16_t *__restrict__ qkeys, int d, int K,
                                                      const int16_t *__restrict__ vkeys, const int *__restrict__ table,
                                                      uint32_t mask, int *__restrict__ vid_out, int *__restrict__ missing)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= K) return;
    const int16_t *key = qkeys + (int64_t)q * d;
    uint32_t h = mix_begin();
    for (int i = 0; i < d; i++) h = mix_step(h, key[i]);
    h = mix_end(h) & mask;
    int found = -1;
    for (;;) {
        const int t = table[h];
        if (t == PHL_EMPTY) break;
        const int16_t *other = vkeys + (int64_t)(-(t + 1)) * d;
        bool same = true;
        for (int i = 0; i < d; i++) same &= (other[i] == key[i]);
        if (same) { found = -(t + 1); break; }
        h = (h + 1) & mask;
    }
    vid_out[q] = found;
    missing[q] = found < 0 ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_append_missing(const int16_t *__restrict__ qkeys, int d, int K,
                                                        const int *__restrict__ missing_rank, int M_old,
                                                        int16_t *__restrict__ vkeys, int *__restrict__ vid_io)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= K || vid_io[q] >= 0) return;
    const int v = M_old + missing_rank[q];
    for (int i = 0; i < d; i++) vkeys[(int64_t)v * d + i] = qkeys[(int64_t)q * d + i];
    vid_io[q] = v;
}

template <int D>
void launch_elevate(const float *ref, int64_t rs, int64_t cs, int64_t n, const sf_t &sf, uint32_t *recs,
                    phl_replay_t *replay, int *err, float *mm, hipStream_t st)
{
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_elevate<D>, dim3(blocks), dim3(256), 0, st, ref, rs, cs, n, sf, recs, replay, err, mm);
}

// [nb][d][2] block ranges -> [gridDim][d][2]: a thread takes whole blocks' records, a workgroup reduces its threads
__global__ __launch_bounds__(256) void k_minmax_stage(const float *__restrict__ mm, int nb, int d, float *__restrict__ out)
{
    __shared__ float smin[256], smax[256];
    for (int i = 0; i < d; i++) {
        float a = INFINITY, b = -INFINITY;
        for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += gridDim.x * blockDim.x) {
            a = fminf(a, mm[((int64_t)k * d + i) * 2 + 0]);
            b = fmaxf(b, mm[((int64_t)k * d + i) * 2 + 1]);
        }
        smin[threadIdx.x] = a;
        smax[threadIdx.x] = b;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) {
                smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + o]);
                smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + o]);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            out[((int64_t)blockIdx.x * d + i) * 2 + 0] = smin[0];
            out[((int64_t)blockIdx.x * d + i) * 2 + 1] = smax[0];
        }
        __syncthreads();
    }
}

// [nb][d][2] ranges -> [d][2]; then the build's scalars into the host's mailbox: {M, err, lo/hi per feature}
__global__ __launch_bounds__(256) void k_build_info(const float *__restrict__ mm, int nb, int d, const int *__restrict__ m_total,
                                                    const int *__restrict__ err, int *__restrict__ info)
{
    __shared__ float smin[256], smax[256];
    for (int i = 0; i < d; i++) {
        float a = INFINITY, b = -INFINITY;
        for (int k = threadIdx.x; k < nb; k += 256) {
            a = fminf(a, mm[((int64_t)k * d + i) * 2 + 0]);
            b = fmaxf(b, mm[((int64_t)k * d + i) * 2 + 1]);
        }
        smin[threadIdx.x] = a;
        smax[threadIdx.x] = b;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) {
                smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + o]);
                smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + o]);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            info[2 + 2 * i] = __float_as_int(smin[0]);
            info[3 + 2 * i] = __float_as_int(smax[0]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        info[0] = *m_total;
        info[1] = *err;
    }
}

template <int D>
void launch_insert(const uint32_t *recs, int64_t n, int *table, uint32_t mask, int *slot_of, int max_probe, int *err, hipStream_t st)
{
    const int64_t waves = ((n + 63) / 64) * (D + 1);
    hipLaunchKernelGGL(k_insert<D>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, recs, (int)n, table, mask, slot_of, max_probe, err);
}

template <int D>
void launch_assign(const unsigned long long *bits, const int *wrank, const uint32_t *recs, int64_t cap, int *table,
                   int16_t *vkeys, int *vfirst, hipStream_t st)
{
    hipLaunchKernelGGL(k_assign<D>, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, st, bits, wrank, recs, cap, table, vkeys, vfirst);
}

constexpr int rec_words(int d) { return (d + 1) / 2 + (d + 5) / 6; }

#define PHL_FOR_D(X)                                                                                                     \
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)

__global__ __launch_bounds__(256) void k_iota_from(int *p, int n, int first)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = first + i;
}

// nbr[axis 0][first-touch vertex 0][side 0] = value (first-touch id or -1): the one neighbour a table doubling
// inside the reference's blur() decides (phl_reftable.hip)
__global__ void k_override_nbr00(int *nbr, const int *__restrict__ int_of_ft, int value)
{
    const int row = int_of_ft ? int_of_ft[0] : 0;
    nbr[(int64_t)row * 2] = (value >= 0 && int_of_ft) ? int_of_ft[value] : value;
}

}  // namespace

// Persistent compact table (capacity >= 2M) + blur neighbour ids for ALL current vertices.
// *scratch_out: a device block the launches read (packed keys); the caller releases it with phl_dev_free once the
// stream has been synchronised (the block cache may hand a freed block to another thread's build at once).
int phl_rebuild_table_and_neighbors(phl_lattice *lat, hipStream_t st, void **scratch_out)
{
    *scratch_out = nullptr;
    const int d = lat->d;
    const int M = (int)lat->M;
    if (lat->table) PHL_HIP(phl_dev_free(lat->table));
    if (lat->nbr) PHL_HIP(phl_dev_free(lat->nbr));
    if (lat->nbr2) PHL_HIP(phl_dev_free(lat->nbr2));
    lat->table = nullptr;
    lat->nbr = nullptr;
    lat->nbr2 = nullptr;
    uint64_t cap = 1024;
    while (cap < (uint64_t)M * 2) cap <<= 1;
    lat->table_mask = (uint32_t)(cap - 1);
    PHL_HIP(phl_dev_malloc((void **)&lat->table, sizeof(int) * cap));
    PHL_HIP(phl_dev_malloc((void **)&lat->nbr, sizeof(int32_t) * (size_t)(M ? M : 1) * (d + 1) * 2));
    hipLaunchKernelGGL(k_fill_i32, dim3(1024), dim3(256), 0, st, lat->table, (int64_t)cap, PHL_EMPTY);
    if (M > 0) {
        hidden_t hidden;
        hidden.n = lat->n_hidden;
        for (int i = 0; i < lat->n_hidden; i++) hidden.id[i] = lat->hidden[i];
        hipLaunchKernelGGL(k_table_insert, dim3((M + 255) / 256), dim3(256), 0, st, lat->vkeys, d, M, lat->table,
                           lat->table_mask, hidden, lat->int_of_ft);
        int rcn = PHL_OK;
        switch (d) {
#define PHL_CASE(D) case D: rcn = launch_neighbors<D>(lat->vkeys, M, lat->table, lat->table_mask, lat->nbr, scratch_out, st); break;
            PHL_FOR_D(PHL_CASE)
#undef PHL_CASE
        }
        if (rcn) return rcn;
        if (lat->nbr00_override != -2)    // a table doubling inside the reference's blur(): phl_reftable.hip
            hipLaunchKernelGGL(k_override_nbr00, dim3(1), dim3(1), 0, st, lat->nbr, lat->int_of_ft, lat->nbr00_override);
        const int npairs = (d + 1) / 2;
        PHL_HIP(phl_dev_malloc((void **)&lat->nbr2, sizeof(int32_t) * (size_t)M * npairs * 8));
        const int64_t totp = (int64_t)M * npairs;
        hipLaunchKernelGGL(k_compose_pairs, dim3((unsigned)((totp + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<const int2 *>(lat->nbr), M, npairs, reinterpret_cast<int4 *>(lat->nbr2));
    }
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}

void phl_release_build_tables(phl_lattice *lat)
{
    // (bt_seg_e / bt_seg_id point into bt_dup_ptr's block: phl_apply_reference_table)
    int32_t **p[] = {&lat->bt_slot_of, &lat->bt_table, &lat->bt_remap, &lat->bt_dup_ptr, &lat->bt_cell};
    for (int32_t **q : p) {
        if (*q) (void)phl_dev_free(*q);
        *q = nullptr;
    }
    lat->bt_seg_e = lat->bt_seg_id = nullptr;
}

int phl_write_final_vids(phl_lattice *lat, hipStream_t st)
{
    if (!lat->bt_slot_of || !lat->bt_table) { phl_set_error("phl_write_final_vids: no build tables"); return PHL_ERR_INVALID; }
    const int N = (int)lat->N;
    if (N > 0) {
        hipLaunchKernelGGL(k_final_vid, dim3((unsigned)(((int64_t)N + 256 * FV_PER - 1) / (256 * FV_PER))), dim3(256), 0, st, lat->bt_table, lat->bt_slot_of, N,
                           lat->bt_remap, lat->bt_dup_ptr, lat->bt_seg_e, lat->bt_seg_id, lat->int_of_ft, lat->replay);
        PHL_HIP(hipGetLastError());
    }
    return PHL_OK;
}

namespace {
// side stream of the early pixel order (below), one per (thread, device)
struct side_t {
    int dev = -1;
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
constexpr int MAX_SIDE_DEV = 64;
thread_local side_t t_sides[MAX_SIDE_DEV];
}  // namespace

// test hook: side streams the calling thread holds (one per device it has built reference-table lattices on)
extern "C" int phl_debug_side_streams(void)
{
    int k = 0;
    for (int i = 0; i < MAX_SIDE_DEV; i++) k += t_sides[i].dev >= 0;
    return k;
}

int phl_build_device(phl_lattice *lat, const float *ref, int64_t rs, int64_t cs, hipStream_t st)
{
    const int d = lat->d;
    const int64_t n = lat->n;
    const int64_t N64 = n * (d + 1);
    lat->N = N64;
    lat->M = 0;
    lat->M_local = 0;
    lat->n_hidden = 0;
    lat->nbr00_override = -2;
    if (n == 0) return PHL_OK;            // (tables: phl_tiles_build)
    const int N = (int)N64;

    // scaleFactor exactly as the reference computes it on the host (permutohedral.h:354-371)
    sf_t sf;
    for (int i = 0; i < PHL_MAX_D; i++) sf.v[i] = 0.f;
    for (int i = 0; i < d; i++) {
        sf.v[i] = 1.0f / (sqrtf((float)(i + 1) * (i + 2)));
        sf.v[i] *= (d + 1) * sqrtf(2.0 / 3);
    }

    // Candidate table.  Worst case every one of the N candidates is a vertex (iid features): 2N slots.  Images have
    // M <= n/2 or so, and a table sized for N = n(d+1) candidates (268 MB at C3) makes every probe, flag and
    // vertex-id lookup a miss in L2 and the Infinity Cache: start with 2n slots (holds M <= n at load 1/2; 33 MB at C3)
    // and repeat the insertion with the full size only if a probe sequence runs long (k_insert's max_probe: 128 steps,
    // which at load 1/2 does not happen; waves that start after the flag is up leave at once, so finding out is cheap).
    uint64_t cap_full = 1024;
    while (cap_full < (uint64_t)N * 2) cap_full <<= 1;
    uint64_t cap = 1 << 16;
    while (cap < (uint64_t)n * 2) cap <<= 1;
    static const bool small_table = !(getenv("PHL_SMALL_TABLE") && atoi(getenv("PHL_SMALL_TABLE")) == 0);
    if (cap > cap_full || !small_table) cap = cap_full;
    uint32_t mask = (uint32_t)(cap - 1);

    temp_pool tmp;
    uint32_t *recs;                                  // per-pixel records the candidate keys are rebuilt from (pix_rec)
    int *table, *slot_of, *wcount, *wrank, *tile_sums, *err;
    unsigned long long *fbits;
    const unsigned gN = (unsigned)((N + 255) / 256);
    const int NW = (int)gN * 4;                      // 64-candidate words of the first-touch mask
    PHL_HIP(tmp.get(&recs, (size_t)n * rec_words(d) + 4));
    phl_release_build_tables(lat);
    PHL_HIP(phl_dev_malloc((void **)&lat->bt_table, sizeof(int) * (size_t)cap));       // (outlive this function: bt_*)
    PHL_HIP(phl_dev_malloc((void **)&lat->bt_slot_of, sizeof(int) * (size_t)N));
    table = lat->bt_table;
    slot_of = lat->bt_slot_of;
    PHL_HIP(tmp.get(&fbits, (size_t)NW));
    PHL_HIP(tmp.get(&wcount, (size_t)NW));
    PHL_HIP(tmp.get(&wrank, (size_t)NW + 1));
    PHL_HIP(tmp.get(&tile_sums, (size_t)NW / SCAN_TILE + 2));
    PHL_HIP(tmp.get(&err, 1));
    float *mm, *mm2;                                 // [blocks of k_elevate][d][2], reduced to [MM2][d][2]
    const int nblk = (int)((n + 255) / 256);
    constexpr int MM2 = 64;
    PHL_HIP(tmp.get(&mm, (size_t)nblk * d * 2));
    PHL_HIP(tmp.get(&mm2, (size_t)MM2 * d * 2));
    int *info_dev;                                   // {M, err, lo/hi per feature}: written by k_build_info
    constexpr int INFO_N = 2 + 2 * PHL_MAX_D;
    phl_pinned_reset();
    int *info_pinned = (int *)phl_pinned_alloc(sizeof(int) * INFO_N);
    int info_host[INFO_N];
    PHL_HIP(tmp.get(&info_dev, (size_t)INFO_N));
    PHL_HIP(phl_dev_malloc((void **)&lat->replay, sizeof(phl_replay_t) * (size_t)N));
    PHL_HIP(hipMemsetAsync(err, 0, sizeof(int), st));

    switch (d) {
#define PHL_CASE(D) case D: launch_elevate<D>(ref, rs, cs, n, sf, recs, lat->replay, err, mm, st); break;
        PHL_FOR_D(PHL_CASE)
#undef PHL_CASE
        default: phl_set_error("d=%d unsupported (1..%d)", d, PHL_MAX_D); return PHL_ERR_UNSUPPORTED;
    }
    int host[2] = {0, 0};
    int rc = PHL_OK;
    for (;;) {
        hipLaunchKernelGGL(k_fill_i32, dim3(2048), dim3(256), 0, st, table, (int64_t)cap, PHL_EMPTY);
        switch (d) {
#define PHL_CASE(D) case D: launch_insert<D>(recs, n, table, mask, slot_of, cap == cap_full ? 0x7FFFFFFF : 128, err, st); break;
            PHL_FOR_D(PHL_CASE)
#undef PHL_CASE
        }
        PHL_HIP(hipMemsetAsync(fbits, 0, sizeof(unsigned long long) * (size_t)NW, st));
        hipLaunchKernelGGL(k_flag_bits, dim3((unsigned)((cap / 4 + 255) / 256)), dim3(256), 0, st, table, (int64_t)cap, fbits);
        hipLaunchKernelGGL(k_word_counts, dim3((unsigned)((NW + 255) / 256)), dim3(256), 0, st, fbits, NW, wcount);
        PHL_HIP(hipGetLastError());
        rc = exclusive_scan(wcount, wrank, NW, tile_sums, st);
        if (rc) return rc;
        // one read-back: the kernel writes straight into pinned host memory where there is some
        if (nblk > 4 * MM2) {
            hipLaunchKernelGGL(k_minmax_stage, dim3(MM2), dim3(256), 0, st, mm, nblk, d, mm2);
            hipLaunchKernelGGL(k_build_info, dim3(1), dim3(256), 0, st, mm2, MM2, d, wrank + NW, err, info_pinned ? info_pinned : info_dev);
        } else {
            hipLaunchKernelGGL(k_build_info, dim3(1), dim3(256), 0, st, mm, nblk, d, wrank + NW, err, info_pinned ? info_pinned : info_dev);
        }
        PHL_HIP(hipGetLastError());
        if (!info_pinned) PHL_HIP(hipMemcpyAsync(info_host, info_dev, sizeof(int) * INFO_N, hipMemcpyDeviceToHost, st));
        PHL_HIP(hipStreamSynchronize(st));
        const int *info = info_pinned ? info_pinned : info_host;
        host[0] = info[0];
        host[1] = info[1];
        for (int i = 0; i < d; i++) {
            memcpy(&lat->feat_lo[i], &info[2 + 2 * i], sizeof(float));
            memcpy(&lat->feat_hi[i], &info[3 + 2 * i], sizeof(float));
        }
        lat->feat_range_valid = 1;
        if ((host[1] & 1) || !(host[1] & 2) || cap == cap_full) break;
        // the small table overflowed (little sharing: most candidates are vertices of their own): full size
        cap = cap_full;
        mask = (uint32_t)(cap - 1);
        PHL_HIP(phl_dev_free(lat->bt_table));             // (the stream has been synchronised)
        lat->bt_table = nullptr;
        PHL_HIP(phl_dev_malloc((void **)&lat->bt_table, sizeof(int) * (size_t)cap));
        table = lat->bt_table;
        PHL_HIP(hipMemsetAsync(err, 0, sizeof(int), st));
    }
    if (host[1] & 1) {
        phl_set_error("lattice coordinate outside int16 (reference keys are `short`, permutohedral.h:39,398); "
                      "rescale the features");
        return PHL_ERR_KEY_RANGE;
    }
    const int M = host[0];
    lat->M = M;

    PHL_HIP(phl_dev_malloc((void **)&lat->vkeys, sizeof(int16_t) * (size_t)M * d));
    if (lat->vfirst) PHL_HIP(phl_dev_free(lat->vfirst));
    lat->vfirst = nullptr;
    lat->vfirst_valid_for_M = 0;
    PHL_HIP(phl_dev_malloc((void **)&lat->vfirst, sizeof(int) * ((size_t)M + 1)));
    switch (d) {
#define PHL_CASE(D) case D: launch_assign<D>(fbits, wrank, recs, (int64_t)cap, table, lat->vkeys, lat->vfirst, st); break;
        PHL_FOR_D(PHL_CASE)
#undef PHL_CASE
    }
    // (replay[].vid is written at the end of phl_tiles_build, through the renamings that follow: phl_write_final_vids)
    PHL_HIP(hipGetLastError());
    if (lat->build_flags & PHL_BUILD_REFERENCE_TABLE) {
        void *arena = nullptr;
        char *arena_c = nullptr;
        const size_t arena_bytes = phl_reftable_scratch_bytes(M);
        if (tmp.get(&arena_c, arena_bytes) == hipSuccess) arena = arena_c;
        else (void)hipGetLastError();
        // the pixel order of the chunk build does not depend on the table: its launches go in under the replay, which is
        // the host waiting for device answers a few bytes at a time
        struct under_t {
            phl_lattice *lat;
            const float *ref;
            int64_t rs, cs;
            void *arena;
            size_t bytes;
            hipStream_t st, aux;
            hipEvent_t fork, join;
            bool launched;
        } under = {lat, ref, rs, cs, nullptr, 0, st, nullptr, nullptr, nullptr, false};
        static const bool early = !(getenv("PHL_EARLY_PIXEL_ORDER") && atoi(getenv("PHL_EARLY_PIXEL_ORDER")) == 0);
        char *arena2 = nullptr;
        under.bytes = phl_tiles_pixel_order_scratch_bytes(n);
        // (a second stream: on the caller's the replay's device queries would queue behind these launches)
        // one side stream per (thread, device): a thread that deals builds over several GPUs (phl.batched_filter, the NCHW
        // mean field) comes back to each device's own stream instead of dropping and re-creating it
        // (never destroyed: a thread's exit may come after the runtime has shut down)
        side_t *const sides = t_sides;
        side_t none;
        side_t *sidep = &none;
        if (early) {
            int dev = -1;
            if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_SIDE_DEV) {
                sidep = &sides[dev];
                if (sidep->dev != dev) {                 // first build of this thread on this device
                    side_t fresh;
                    if (hipStreamCreateWithFlags(&fresh.s, hipStreamNonBlocking) == hipSuccess &&
                        hipEventCreateWithFlags(&fresh.fork, hipEventDisableTiming) == hipSuccess &&
                        hipEventCreateWithFlags(&fresh.join, hipEventDisableTiming) == hipSuccess) {
                        fresh.dev = dev;
                        *sidep = fresh;
                    } else {                             // nothing half-made is kept
                        (void)hipGetLastError();
                        if (fresh.s) (void)hipStreamDestroy(fresh.s);
                        if (fresh.fork) (void)hipEventDestroy(fresh.fork);
                        if (fresh.join) (void)hipEventDestroy(fresh.join);
                    }
                }
            } else {
                (void)hipGetLastError();
            }
            side_t &side = *sidep;
            if (side.dev == dev && dev >= 0 && tmp.get(&arena2, under.bytes) == hipSuccess) {
                under.arena = arena2;
                under.aux = side.s;
                under.fork = side.fork;
                under.join = side.join;
            } else {
                (void)hipGetLastError();
            }
        }
        auto hook = [](void *a) -> int {
            under_t *u = (under_t *)a;
            PHL_HIP(hipEventRecord(u->fork, u->st));                 // everything enqueued so far (vertex keys, first touches)
            PHL_HIP(hipStreamWaitEvent(u->aux, u->fork, 0));
            const int rcp = phl_tiles_pixel_order(u->lat, u->ref, u->rs, u->cs, u->arena, u->bytes, u->aux);
            PHL_HIP(hipEventRecord(u->join, u->aux));
            u->launched = true;
            return rcp;
        };
        rc = phl_apply_reference_table(lat, st, arena, arena ? arena_bytes : 0, under.arena ? +hook : nullptr, &under);
        if (under.launched) {
            if (rc) (void)hipStreamSynchronize(under.aux);      // (their temporaries die with this function)
            else PHL_HIP(hipStreamWaitEvent(st, under.join, 0));   // the caller's stream carries on behind them
        }
        if (rc) return rc;
    }
    if (lat->M != M && lat->vfirst_valid_for_M != lat->M) {     // duplicate vertices inserted without a list of their own
        PHL_HIP(phl_dev_free(lat->vfirst));
        lat->vfirst = nullptr;
    }
    lat->M_local = lat->M;
    // (locality renumbering, key -> vertex table and blur neighbours follow in phl_tiles_build)
    PHL_HIP(hipGetLastError());
    PHL_HIP(hipStreamSynchronize(st));  // temporaries are freed on return
    return PHL_OK;
}

// Pixel-sorted contribution lists per vertex (the transpose of `replay`).  Only the
// reference-exact gather splat (PHL_FILTER_EXACT / fallback shapes) and the introspection calls
// need them, so they are built on first use, from `replay` alone.
static int build_csr(phl_lattice *lat, hipStream_t st);
int phl_ensure_csr(phl_lattice *lat, hipStream_t st)
{
    std::lock_guard<std::mutex> once(*phl_csr_mutex(lat));     // several threads may filter through one lattice
    if (!(lat->csr_ptr && lat->csr)) {
        const int rc = build_csr(lat, st);
        if (rc) return rc;
    }
    return phl_tiles_ensure_vorder(lat, st);  // (the gather splat's vertex order: same first use, same lock)
}

static int build_csr(phl_lattice *lat, hipStream_t st)
{
    const int M = (int)lat->M, N = (int)lat->N, dp1 = lat->d + 1;
    if (lat->csr_ptr) PHL_HIP(phl_dev_free(lat->csr_ptr));
    if (lat->csr) PHL_HIP(phl_dev_free(lat->csr));
    lat->csr_ptr = nullptr;
    lat->csr = nullptr;
    PHL_HIP(phl_dev_malloc((void **)&lat->csr_ptr, sizeof(int32_t) * ((size_t)M + 1)));
    PHL_HIP(phl_dev_malloc((void **)&lat->csr, sizeof(phl_contrib_t) * (size_t)(N ? N : 1)));
    if (N == 0 || M == 0) {
        PHL_HIP(hipMemsetAsync(lat->csr_ptr, 0, sizeof(int32_t) * ((size_t)M + 1), st));
        PHL_HIP(hipStreamSynchronize(st));
        return PHL_OK;
    }
    temp_pool tmp;
    int *cnt, *tile_sums, *vkey, *perm;
    PHL_HIP(tmp.get(&cnt, (size_t)M));
    PHL_HIP(tmp.get(&tile_sums, (size_t)M / SCAN_TILE + 2));
    PHL_HIP(tmp.get(&vkey, (size_t)N));
    PHL_HIP(tmp.get(&perm, (size_t)N));
    PHL_HIP(hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)M, st));
    const unsigned gN = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(k_count_vid, dim3(gN), dim3(256), 0, st, lat->replay, N, cnt);
    PHL_HIP(hipGetLastError());
    int rc = exclusive_scan(cnt, lat->csr_ptr, M, tile_sums, st);
    if (rc) return rc;
    // a pixel holds a vertex at most once (the d+1 vertices of a simplex are distinct), so a STABLE sort of the
    // candidates by vertex id leaves every list in ascending pixel order; O(N) for any list lengths
    hipLaunchKernelGGL(k_replay_vids, dim3(gN), dim3(256), 0, st, lat->replay, N, vkey);
    PHL_HIP(hipGetLastError());
    rc = stable_sort_perm(vkey, N, M, perm, tmp, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_fill_sorted, dim3(gN), dim3(256), 0, st, lat->replay, perm, N, dp1, lat->csr);
    PHL_HIP(hipGetLastError());
    PHL_HIP(hipStreamSynchronize(st));  // temporaries go back to the scratch cache
    return PHL_OK;
}
// Append vertices that exist in a neighbouring row band ("ghosts": no local contributions, so
// their splat lists are empty) and return the local id of every queried key.  Keys must be
// distinct.  Used by the row-band multi-GPU path; rebuilds the table and the neighbour ids.
int phl_add_vertices_device(phl_lattice *lat, const int16_t *keys_host, int64_t count, int32_t *vid_host, hipStream_t st)
{
    void *nbr_scratch = nullptr;
    if (count == 0) return PHL_OK;
    const int d = lat->d;
    const int K = (int)count;
    const int M_old = (int)lat->M;
    temp_pool tmp;
    int16_t *qkeys;
    int *vid, *missing, *mrank, *tile_sums;
    PHL_HIP(tmp.get(&qkeys, (size_t)K * d));
    PHL_HIP(tmp.get(&vid, (size_t)K));
    PHL_HIP(tmp.get(&missing, (size_t)K));
    PHL_HIP(tmp.get(&mrank, (size_t)K + 1));
    PHL_HIP(tmp.get(&tile_sums, (size_t)K / SCAN_TILE + 2));
    PHL_HIP(hipMemcpyAsync(qkeys, keys_host, sizeof(int16_t) * (size_t)K * d, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_table_lookup, dim3((K + 255) / 256), dim3(256), 0, st, qkeys, d, K, lat->vkeys, lat->table,
                       lat->table_mask, vid, missing);
    PHL_HIP(hipGetLastError());
    int rc = exclusive_scan(missing, mrank, K, tile_sums, st);
    if (rc) return rc;
    int n_new = 0;
    PHL_HIP(hipMemcpyAsync(&n_new, mrank + K, sizeof(int), hipMemcpyDeviceToHost, st));
    PHL_HIP(hipStreamSynchronize(st));
    if (n_new > 0) {
        const int M_new = M_old + n_new;
        int16_t *vkeys_new;
        PHL_HIP(phl_dev_malloc((void **)&vkeys_new, sizeof(int16_t) * (size_t)M_new * d));
        if (M_old > 0)
            PHL_HIP(hipMemcpyAsync(vkeys_new, lat->vkeys, sizeof(int16_t) * (size_t)M_old * d, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_append_missing, dim3((K + 255) / 256), dim3(256), 0, st, qkeys, d, K, mrank, M_old, vkeys_new,
                           vid);
        PHL_HIP(hipGetLastError());
        PHL_HIP(hipStreamSynchronize(st));
        if (lat->int_of_ft) {        // ghosts are appended at the end in both numberings
            int32_t *a, *b;
            PHL_HIP(phl_dev_malloc((void **)&a, sizeof(int32_t) * (size_t)M_new));
            PHL_HIP(phl_dev_malloc((void **)&b, sizeof(int32_t) * (size_t)M_new));
            PHL_HIP(hipMemcpyAsync(a, lat->ft_of_int, sizeof(int32_t) * (size_t)M_old, hipMemcpyDeviceToDevice, st));
            PHL_HIP(hipMemcpyAsync(b, lat->int_of_ft, sizeof(int32_t) * (size_t)M_old, hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(k_iota_from, dim3((n_new + 255) / 256), dim3(256), 0, st, a + M_old, n_new, M_old);
            hipLaunchKernelGGL(k_iota_from, dim3((n_new + 255) / 256), dim3(256), 0, st, b + M_old, n_new, M_old);
            PHL_HIP(hipGetLastError());
            PHL_HIP(hipStreamSynchronize(st));
            PHL_HIP(phl_dev_free(lat->ft_of_int));
            PHL_HIP(phl_dev_free(lat->int_of_ft));
            lat->ft_of_int = a;
            lat->int_of_ft = b;
        }
        if (lat->vkeys) PHL_HIP(phl_dev_free(lat->vkeys));
        // the per-vertex contribution lists are indexed by M: drop them, they are rebuilt on demand
        // (ghosts get empty lists)
        if (lat->csr_ptr) PHL_HIP(phl_dev_free(lat->csr_ptr));
        if (lat->csr) PHL_HIP(phl_dev_free(lat->csr));
        lat->csr_ptr = nullptr;
        lat->csr = nullptr;
        lat->vkeys = vkeys_new;
        lat->M = M_new;
        rc = phl_rebuild_table_and_neighbors(lat, st, &nbr_scratch);
        if (rc) return rc;
        // (the value workspaces are sized by M: phl_add_vertices drops them)
    }
    PHL_HIP(hipMemcpyAsync(vid_host, vid, sizeof(int32_t) * (size_t)K, hipMemcpyDeviceToHost, st));
    PHL_HIP(hipStreamSynchronize(st));
    if (nbr_scratch) PHL_HIP(phl_dev_free(nbr_scratch));
    return PHL_OK;
}
