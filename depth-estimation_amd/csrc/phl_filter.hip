// phl_filter.hip -- blur (all modes) and the gather forms of splat / slice on gfx950.
//
// k_blur is the blur of every filter call.  k_splat / k_slice are the plain gather kernels:
// they carry PHL_FILTER_EXACT (reference-exact summation order) and every case the LDS-staged
// chunk kernels of phl_tiles.hip do not take (vd % 4 != 0, unaligned rows, chunks that share
// too few vertices).  The default fast path for splat and slice lives in phl_tiles.hip.
//
// All stages are HBM/L2-bound row gathers over fp32 rows of `vd` channels (no MFMA:
// arithmetic intensity < 1 flop/byte).  Common shape of every kernel here:
//
//   * a row (one lattice vertex or one pixel, vd channels) is owned by a group of LPR lanes,
//     each lane moving VEC=4 consecutive channels with one 16-byte global_load/store_dwordx4
//     (VEC=1 fallback for vd % 4 != 0 or unaligned rows).  vd = 256 -> one 64-lane wavefront
//     reads a whole 1 KiB row per instruction, the widest coalesced access gfx950 has;
//   * a wavefront owns 64/LPR rows at a time and keeps several independent row loads in
//     flight before the first use (the only latency hiding a gather has);
//   * index data (contribution lists, neighbour ids, replay entries) is read once per row
//     group, wave-uniform where LPR == 64 so it goes through the scalar cache.
//
// Numerics: compiled with -ffp-contract=off and written so that each stage rounds exactly
// like the reference's scalar loops (crf/lattice/lite/permutohedral.h):
//   splat  vert[v] += w * src[p]   in ascending pixel order           (:236-238, :454-455)
//   blur   2*(0.25*a + 0.5*s + 0.25*b), Jacobi per axis               (:526, :530-532)
//   slice  col += w * vert / (1 + 2^-d) per term                      (:480)
// so with PHL_FILTER_EXACT the device results are bit-identical to the CPU path (the splat lists
// are pixel-sorted, so the summation order is the reference's everywhere).  Without the flag
// slice uses fma + one final multiply by 1/(1+2^-d).
#include <stdlib.h>

#include <type_traits>

#include "phl_internal.h"

namespace {

template <int VEC> struct vec_of;
template <> struct vec_of<1> { using type = float; };
template <> struct vec_of<4> { using type = float4; };

template <int VEC> __device__ __forceinline__ typename vec_of<VEC>::type vzero();
template <> __device__ __forceinline__ float vzero<1>() { return 0.f; }
template <> __device__ __forceinline__ float4 vzero<4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }

__device__ __forceinline__ float vload(const float *p, float) { return *p; }
__device__ __forceinline__ float4 vload(const float *p, float4) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void vstore(float *p, float v) { *p = v; }
__device__ __forceinline__ void vstore(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// acc + w*q, mul and add rounded separately (reference: val[i] += (barycentric*value[i]))
__device__ __forceinline__ float mac(float acc, float w, float q) { return acc + w * q; }
__device__ __forceinline__ float4 mac(float4 acc, float w, float4 q)
{
    return make_float4(acc.x + w * q.x, acc.y + w * q.y, acc.z + w * q.z, acc.w + w * q.w);
}
__device__ __forceinline__ float fmac(float acc, float w, float q) { return __builtin_fmaf(w, q, acc); }
__device__ __forceinline__ float4 fmac(float4 acc, float w, float4 q)
{
    return make_float4(__builtin_fmaf(w, q.x, acc.x), __builtin_fmaf(w, q.y, acc.y), __builtin_fmaf(w, q.z, acc.z),
                       __builtin_fmaf(w, q.w, acc.w));
}

// reference blur arithmetic, literally (:526)
__device__ __forceinline__ float blur3(float a, float s, float b) { return 2 * (0.25f * a + 0.5f * s + 0.25f * b); }
__device__ __forceinline__ float4 blur3(float4 a, float4 s, float4 b)
{
    return make_float4(blur3(a.x, s.x, b.x), blur3(a.y, s.y, b.y), blur3(a.z, s.z, b.z), blur3(a.w, s.w, b.w));
}

// correctly rounded t / c for c = 1 + 2^-d with rc = RN(1/c): multiply, exact residual, one
// correction (Markstein).  Equals the IEEE quotient the reference computes in every normal-
// range case (checked exhaustively-by-sampling in tests/test_exact_divide.py).
__device__ __forceinline__ float div_c(float t, float c, float rc)
{
    float q = t * rc;
    float rem = __builtin_fmaf(-q, c, t);
    return __builtin_fmaf(rem, rc, q);
}
// acc + (w*v)/c
__device__ __forceinline__ float slice_term(float acc, float w, float v, float c, float rc)
{
    return acc + div_c(w * v, c, rc);
}
__device__ __forceinline__ float4 slice_term(float4 acc, float w, float4 v, float c, float rc)
{
    return make_float4(slice_term(acc.x, w, v.x, c, rc), slice_term(acc.y, w, v.y, c, rc),
                       slice_term(acc.z, w, v.z, c, rc), slice_term(acc.w, w, v.w, c, rc));
}
__device__ __forceinline__ float vscale(float a, float s) { return a * s; }
__device__ __forceinline__ float4 vscale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float vsub(float a, float b) { return a - b; }
__device__ __forceinline__ float4 vsub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// wave id inside the grid, as a scalar
__device__ __forceinline__ int wave_in_grid()
{
    return (int)blockIdx.x * (int)(blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}
__device__ __forceinline__ int waves_in_grid() { return (int)gridDim.x * (int)(blockDim.x >> 6); }

// ------------------------------------------------------------------------------------------
// splat: vert[v][:] = sum_{(p,w) in list(v), ascending p}  w * src[p][:]
// Gather form of the reference's scatter: every vertex sums its own contribution list, so no
// float atomics, and the accumulation order is the reference's pixel order.
template <int VEC, int LPR>
__global__ __launch_bounds__(256) void k_splat(const float *__restrict__ src, int64_t src_rs, int vd,
                                               const int *__restrict__ ptr, const phl_contrib_t *__restrict__ csr,
                                               int M, float *__restrict__ vert, const int *__restrict__ vorder,
                                               int xcd_chunk)
{
    using V = typename vec_of<VEC>::type;
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR;
    const int l = lane % LPR;
    // vertices are visited in chunk-major order (vorder) and every XCD gets a contiguous part of
    // that order (see k_blur): vertices summed together then share pixel rows through one L2
    const int lb = xcd_chunk > 0 ? (int)(blockIdx.x & 7) * xcd_chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int wv = lb * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    {
        const int64_t idx = (int64_t)wv * G + sub;
        if ((int64_t)wv * G >= M) return;
        const bool act = idx < M;
        const int64_t v = act ? (vorder ? vorder[idx] : idx) : 0;
        int beg = 0, end = 0;
        if (act) { beg = ptr[v]; end = ptr[v + 1]; }
        for (int c = l * VEC; c < vd; c += LPR * VEC) {
            V acc = vzero<VEC>();
            int e = beg;
            for (; e + 8 <= end; e += 8) {      // 8 independent row loads in flight, summed in list order
                phl_contrib_t cc[8];
                V qq[8];
#pragma unroll
                for (int u = 0; u < 8; u++) cc[u] = csr[e + u];
#pragma unroll
                for (int u = 0; u < 8; u++) qq[u] = vload(src + (int64_t)cc[u].pixel * src_rs + c, V());
#pragma unroll
                for (int u = 0; u < 8; u++) acc = mac(acc, cc[u].w, qq[u]);
            }
            for (; e + 4 <= end; e += 4) {
                const phl_contrib_t c0 = csr[e], c1 = csr[e + 1], c2 = csr[e + 2], c3 = csr[e + 3];
                const V q0 = vload(src + (int64_t)c0.pixel * src_rs + c, V());
                const V q1 = vload(src + (int64_t)c1.pixel * src_rs + c, V());
                const V q2 = vload(src + (int64_t)c2.pixel * src_rs + c, V());
                const V q3 = vload(src + (int64_t)c3.pixel * src_rs + c, V());
                acc = mac(acc, c0.w, q0);
                acc = mac(acc, c1.w, q1);
                acc = mac(acc, c2.w, q2);
                acc = mac(acc, c3.w, q3);
            }
            for (; e < end; e++) {
                const phl_contrib_t c0 = csr[e];
                acc = mac(acc, c0.w, vload(src + (int64_t)c0.pixel * src_rs + c, V()));
            }
            if (act) vstore(vert + v * vd + c, acc);
        }
    }
}

// Rows a blur launch computes: logical index j in [0, total) -> row, over up to three ascending row ranges (a row-band
// lattice computes, per axis, only the rows whose output something later reads: phl_set_blur_rows).  Unrestricted:
// one range [0, M).  M stays the array extent (neighbour ids are rows of the whole array).
struct blur_rows_t {
    int n0, n01, total;     // rows in range 0, in ranges 0 + 1, in all
    int b0, b1, b2;         // first row of each range
    __device__ __forceinline__ int64_t row(int64_t j) const { return j < n0 ? b0 + j : (j < n01 ? b1 + (j - n0) : b2 + (j - n01)); }
};

// ------------------------------------------------------------------------------------------
// blur along one lattice axis (Jacobi): out[v] = 2*(1/4 in[n1(v)] + 1/2 in[v] + 1/4 in[n2(v)]),
// absent neighbour = 0 (the reference never creates vertices in blur, :516-522).
template <int VEC, int LPR>
__global__ __launch_bounds__(256) void k_blur(const float *__restrict__ vin, float *__restrict__ vout,
                                              const int2 *__restrict__ nbr, int M, int vd, int xcd_chunk, const blur_rows_t rr)
{
    using V = typename vec_of<VEC>::type;
    constexpr int G = 64 / LPR;
    constexpr int U = 4;  // row groups in flight per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR;
    const int l = lane % LPR;
    // XCD-aware block order: workgroups are dealt round-robin to the 8 XCDs, so blocks b and
    // b+8 share an L2.  Giving every XCD one CONTIGUOUS eighth of the vertex range makes a
    // vertex's blur neighbours (a few thousand rows away at most in first-touch order) land in
    // the same L2 instead of being re-fetched by another XCD.  Speed only, never correctness.
    const int lb = xcd_chunk > 0 ? (int)(blockIdx.x & 7) * xcd_chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int wv = lb * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    {
        const int64_t j0 = (int64_t)wv * G * U;
        if (j0 >= rr.total) return;
        int64_t v[U];                 // row of each group's vertex; M = none (past the end)
        int2 nb[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t j = j0 + u * G + sub;
            v[u] = j < rr.total ? rr.row(j) : (int64_t)M;
            nb[u] = v[u] < M ? nbr[v[u]] : make_int2(-1, -1);
        }
        for (int c = l * VEC; c < vd; c += LPR * VEC) {
            // unconditional loads from clamped rows + selects: a load under a divergent `if`
            // makes hipcc wait vmcnt(0) per load and serialises the gather
            V a[U], s[U], b[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t vc = v[u] < M ? v[u] : (int64_t)M - 1;
                s[u] = vload(vin + vc * vd + c, V());
                // absent neighbours (-1) issue no load when no vertex of the wave has one (wave-uniform test, see
                // k_blur2); otherwise the vertex's own row is loaded and discarded
                a[u] = b[u] = vzero<VEC>();
                if (__ballot(nb[u].x >= 0) != 0ull) a[u] = vload(vin + (nb[u].x >= 0 ? (int64_t)nb[u].x : vc) * vd + c, V());
                if (__ballot(nb[u].y >= 0) != 0ull) b[u] = vload(vin + (nb[u].y >= 0 ? (int64_t)nb[u].y : vc) * vd + c, V());
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (nb[u].x < 0) a[u] = vzero<VEC>();
                if (nb[u].y < 0) b[u] = vzero<VEC>();
                if (v[u] < M) vstore(vout + v[u] * vd + c, blur3(a[u], s[u], b[u]));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Two consecutive blur axes (a = 2p, then b = 2p+1) in one pass: every value the second blur reads
// is recomputed from the input with the first blur's own expression, so the result is bit-identical
// to two k_blur passes while the vertex array is read and written once instead of twice.
// ids per vertex (k_compose_pairs): { a-(b-), b-, a+(b-), a-(v) | a+(v), a-(b+), b+, a+(b+) }.
template <int VEC, int LPR, int U>
__global__ __launch_bounds__(256) void k_blur2(const float *__restrict__ vin, float *__restrict__ vout,
                                               const int4 *__restrict__ nb2, int M, int vd, int xcd_chunk, const blur_rows_t rr)
{
    using V = typename vec_of<VEC>::type;
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR;
    const int l = lane % LPR;
    const int lb = xcd_chunk > 0 ? (int)(blockIdx.x & 7) * xcd_chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int wv = lb * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t j0 = (int64_t)wv * G * U;
    if (j0 >= rr.total) return;
    int64_t v[U];                     // row of each group's vertex; M = none (past the end)
    int id[U][8];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int64_t j = j0 + u * G + sub;
        v[u] = j < rr.total ? rr.row(j) : (int64_t)M;
        const int64_t vc = v[u] < M ? v[u] : (int64_t)M - 1;
        const int4 i0 = nb2[vc * 2], i1 = nb2[vc * 2 + 1];
        id[u][0] = i0.x; id[u][1] = i0.y; id[u][2] = i0.z; id[u][3] = i0.w;
        id[u][4] = i1.x; id[u][5] = i1.y; id[u][6] = i1.z; id[u][7] = i1.w;
    }
    for (int c = l * VEC; c < vd; c += LPR * VEC) {
        V x[U][8], s[U];
#pragma unroll
        for (int u = 0; u < U; u++) {   // unconditional loads from clamped rows (see k_blur)
            const int64_t vc = v[u] < M ? v[u] : (int64_t)M - 1;
            s[u] = vload(vin + vc * vd + c, V());
#pragma unroll
            // Absent neighbours (id < 0) are more than half of the stencil on a sparse lattice: 2.7 / 4.2 / 5.1 of
            // the 8 at C3 for the three axis pairs.  Loading a dummy row for them (round 1) kept the code
            // branch-free but made the pass issue-bound on loads whose results it throws away: skipping them took the
            // three passes from 0.459 to 0.385 ms.  The test must be WAVE-UNIFORM (a load under a divergent `if`
            // makes hipcc drain vmcnt per load).
            for (int k = 0; k < 8; k++) {
                if constexpr (LPR == 64) {
                    // one vertex per wave: the ids are wave-uniform, so an absent neighbour is skipped by a SCALAR
                    // branch (no divergence, the loads that are issued still go out back to back)
                    const int idk = __builtin_amdgcn_readfirstlane(id[u][k]);
                    x[u][k] = vzero<VEC>();
                    if (idk >= 0) x[u][k] = vload(vin + (int64_t)idk * vd + c, V());
                } else {
                    // several vertices per wave: skip the load when NONE of them has this neighbour (still a
                    // wave-uniform test); lanes of a vertex without it load its own row and discard it
                    x[u][k] = vzero<VEC>();
                    if (__ballot(id[u][k] >= 0) != 0ull)
                        x[u][k] = vload(vin + (id[u][k] >= 0 ? (int64_t)id[u][k] : vc) * vd + c, V());
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (id[u][k] < 0) x[u][k] = vzero<VEC>();
            // absent b-neighbour: the reference reads a row of zeros there (:516-522)
            const V tm = id[u][1] >= 0 ? blur3(x[u][0], x[u][1], x[u][2]) : vzero<VEC>();
            const V t0 = blur3(x[u][3], s[u], x[u][4]);
            const V tp = id[u][6] >= 0 ? blur3(x[u][5], x[u][6], x[u][7]) : vzero<VEC>();
            if (v[u] < M) vstore(vout + v[u] * vd + c, blur3(tm, t0, tp));
        }
    }
}

// ------------------------------------------------------------------------------------------
// slice: out[p][:] = sum_{r<=d} w_r * vert[v_r][:] / (1 + 2^-d)   [ - sub[p][:] ]
template <int VEC, int LPR, bool EXACT>
__global__ __launch_bounds__(256) void k_slice(const float *__restrict__ vert, int vd,
                                               const phl_replay_t *__restrict__ replay, int dp1, int64_t n,
                                               float *__restrict__ out, int64_t out_rs,
                                               const float *__restrict__ sub_src, int64_t sub_rs, float cdiv,
                                               float rcdiv)
{
    using V = typename vec_of<VEC>::type;
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR;
    const int l = lane % LPR;
    const int64_t stride = (int64_t)waves_in_grid() * G;
    for (int64_t p0 = (int64_t)wave_in_grid() * G; p0 < n; p0 += stride) {
        const int64_t p = p0 + sub;
        if (p >= n) continue;
        const phl_replay_t *rp = replay + p * dp1;
        for (int c = l * VEC; c < vd; c += LPR * VEC) {
            V acc = vzero<VEC>();
            int r = 0;
            for (; r + 3 <= dp1; r += 3) {
                const phl_replay_t r0 = rp[r], r1 = rp[r + 1], r2 = rp[r + 2];
                const V q0 = vload(vert + (int64_t)r0.vid * vd + c, V());
                const V q1 = vload(vert + (int64_t)r1.vid * vd + c, V());
                const V q2 = vload(vert + (int64_t)r2.vid * vd + c, V());
                if (EXACT) {
                    acc = slice_term(acc, r0.w, q0, cdiv, rcdiv);
                    acc = slice_term(acc, r1.w, q1, cdiv, rcdiv);
                    acc = slice_term(acc, r2.w, q2, cdiv, rcdiv);
                } else {
                    acc = fmac(acc, r0.w, q0);
                    acc = fmac(acc, r1.w, q1);
                    acc = fmac(acc, r2.w, q2);
                }
            }
            for (; r < dp1; r++) {
                const phl_replay_t r0 = rp[r];
                const V q0 = vload(vert + (int64_t)r0.vid * vd + c, V());
                if (EXACT) acc = slice_term(acc, r0.w, q0, cdiv, rcdiv);
                else acc = fmac(acc, r0.w, q0);
            }
            if (!EXACT) acc = vscale(acc, rcdiv);
            if (sub_src) acc = vsub(acc, vload(sub_src + p * sub_rs + c, V()));
            vstore(out + p * out_rs + c, acc);
        }
    }
}

// ------------------------------------------------------------------------------------------
// strided 2-D copy through a padded 64x64 LDS tile: both sides coalesced when either stride
// of each side is 1 (NCHW <-> pixel-major staging for BatchedAdjacency-style views,
// gaussian_matrix.py:348-349).
__global__ __launch_bounds__(256) void k_copy2d(const float *__restrict__ src, int64_t srs, int64_t scs,
                                                float *__restrict__ dst, int64_t drs, int64_t dcs, int64_t rows,
                                                int cols)
{
    __shared__ float tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    // read: make the unit-stride source dimension the fast one
    const bool src_col_fast = (scs == 1) || (srs != 1);
    for (int k = ty; k < 64; k += 4) {
        int rr = src_col_fast ? k : tx, cc = src_col_fast ? tx : k;
        if (r0 + rr < rows && c0 + cc < cols) tile[rr][cc] = src[(r0 + rr) * srs + (int64_t)(c0 + cc) * scs];
    }
    __syncthreads();
    const bool dst_col_fast = (dcs == 1) || (drs != 1);
    for (int k = ty; k < 64; k += 4) {
        int rr = dst_col_fast ? k : tx, cc = dst_col_fast ? tx : k;
        if (r0 + rr < rows && c0 + cc < cols) dst[(r0 + rr) * drs + (int64_t)(c0 + cc) * dcs] = tile[rr][cc];
    }
}

// Both sides pixel-major (unit column stride), any row strides and any 4-byte-aligned base: rows copied in 16-byte pieces
// that need no 16-byte alignment (gfx950 takes dword-aligned dwordx4 accesses) -- the repacking of rows that are off the
// 16-byte grid (phl_filter: channel counts that are not a multiple of 4, column slices) runs at the streaming rate instead
// of through the 64x64 transpose tile.  A wave takes four rows at a time, a lane one piece of each.
struct __attribute__((packed, aligned(4))) phl_f4u { float x, y, z, w; };
__global__ __launch_bounds__(256) void k_copy_rows(const float *__restrict__ src, int64_t srs, float *__restrict__ dst, int64_t drs,
                                                   int64_t rows, int cols)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    const int pieces = cols >> 2, rem = cols & 3;
    for (int64_t r0 = wave * 4; r0 < rows; r0 += nw * 4) {
        for (int c = lane; c < pieces; c += 64) {
            phl_f4u v[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (r0 + u < rows) v[u] = *reinterpret_cast<const phl_f4u *>(src + (r0 + u) * srs + 4 * c);
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (r0 + u < rows) *reinterpret_cast<phl_f4u *>(dst + (r0 + u) * drs + 4 * c) = v[u];
        }
        if (lane < 4 * rem) {                    // the last cols % 4 floats of the four rows
            const int u = lane / rem, c = 4 * pieces + lane % rem;
            if (r0 + u < rows) dst[(r0 + u) * drs + c] = src[(r0 + u) * srs + c];
        }
    }
}

// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// Row-band exchange helpers (phl/rowtile.py): pack the boundary vertices' rows into one send
// buffer / add the rows received from the neighbouring bands.  One row per LPR lanes.
template <int LPR>
__global__ __launch_bounds__(256) void k_gather_rows(const float *__restrict__ vert, int vd, const int64_t *__restrict__ idx,
                                                     int64_t k, float *__restrict__ out, int64_t out_rs)
{
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * G + lane / LPR;
    if (r >= k) return;
    const float *src = vert + idx[r] * vd;
    for (int c = (lane % LPR) * 4; c < vd; c += LPR * 4) vstore(out + r * out_rs + c, vload(src + c, float4()));
}

template <int LPR>
__global__ __launch_bounds__(256) void k_scatter_add_rows(float *__restrict__ vert, int vd, const int64_t *__restrict__ idx,
                                                          int64_t k, const float *__restrict__ in, int64_t in_rs)
{
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * G + lane / LPR;
    if (r >= k) return;
    float *dst = vert + idx[r] * vd;
    for (int c = (lane % LPR) * 4; c < vd; c += LPR * 4) {
        const float4 a = vload(dst + c, float4()), b = vload(in + r * in_rs + c, float4());
        vstore(dst + c, make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w));
    }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline int pick_lpr(int vd, int vec)
{
    const int need = (vd + vec - 1) / vec;
    if (need >= 64) return 64;
    if (need >= 32) return 32;      // vd = 128: one pass per row instead of two half-width ones
    if (need >= 16) return 16;
    if (need >= 4) return 4;
    return 1;
}

inline unsigned grid_for(int64_t rows, int rows_per_wave)
{
    int64_t waves = (rows + rows_per_wave - 1) / rows_per_wave;
    int64_t blocks = (waves + 3) / 4;
    const int64_t cap = 256 * 8;  // 8 blocks of 4 waves per CU: full occupancy, grid-stride the rest
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

template <typename F>
inline void dispatch_lpr(int lpr, F &&f)
{
    switch (lpr) {
        case 64: f(std::integral_constant<int, 64>{}); break;
        case 32: f(std::integral_constant<int, 32>{}); break;
        case 16: f(std::integral_constant<int, 16>{}); break;
        case 4: f(std::integral_constant<int, 4>{}); break;
        default: f(std::integral_constant<int, 1>{}); break;
    }
}

}  // namespace

int phl_launch_splat(phl_lattice *lat, const float *src, int64_t src_rs, int vd, float *vert, hipStream_t st)
{
    const int M = (int)lat->M;
    if (M == 0 || vd == 0) return PHL_OK;
    const int rc_csr = phl_ensure_csr(lat, st);
    if (rc_csr) return rc_csr;
    const bool v4 = (vd % 4 == 0) && (src_rs % 4 == 0) && aligned16(src) && aligned16(vert);
    const int lpr = pick_lpr(vd, v4 ? 4 : 1);
    const int rows_per_block = (64 / lpr) * 4;
    int64_t blocks = ((int64_t)M + rows_per_block - 1) / rows_per_block;
    static const bool xcd = !(getenv("PHL_XCD") && atoi(getenv("PHL_XCD")) == 0);
    int xcd_chunk = 0;
    if (xcd && blocks >= 64) {
        blocks = (blocks + 7) / 8 * 8;
        xcd_chunk = (int)(blocks / 8);
    }
    const unsigned grid = (unsigned)blocks;
    dispatch_lpr(lpr, [&](auto L) {
        constexpr int LPR = decltype(L)::value;
        if (v4) k_splat<4, LPR><<<dim3(grid), dim3(256), 0, st>>>(src, src_rs, vd, lat->csr_ptr, lat->csr, M, vert, lat->vorder, xcd_chunk);
        else k_splat<1, LPR><<<dim3(grid), dim3(256), 0, st>>>(src, src_rs, vd, lat->csr_ptr, lat->csr, M, vert, lat->vorder, xcd_chunk);
    });
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}

// the rows a pass whose last axis is `axis` computes
static blur_rows_t rows_of(const phl_lattice *lat, int axis, bool restricted)
{
    blur_rows_t r;
    const int M = (int)lat->M;
    if (!restricted || !lat->blur_rows_set) {
        r.n0 = r.n01 = r.total = M;
        r.b0 = r.b1 = r.b2 = 0;
        return r;
    }
    const int32_t(*g)[2] = lat->blur_rows[axis];
    r.b0 = g[0][0]; r.b1 = g[1][0]; r.b2 = g[2][0];
    r.n0 = g[0][1] - g[0][0];
    r.n01 = r.n0 + (g[1][1] - g[1][0]);
    r.total = r.n01 + (g[2][1] - g[2][0]);
    return r;
}

int phl_launch_blur(const phl_lattice *lat, int axis, const float *vin, float *vout, int vd, hipStream_t st, bool restricted)
{
    const int M = (int)lat->M;
    if (M == 0 || vd == 0) return PHL_OK;
    const blur_rows_t rr = rows_of(lat, axis, restricted);
    if (rr.total == 0) return PHL_OK;
    const int2 *nbr = reinterpret_cast<const int2 *>(lat->nbr) + (int64_t)axis * M;
    const bool v4 = (vd % 4 == 0) && aligned16(vin) && aligned16(vout);
    const int lpr = pick_lpr(vd, v4 ? 4 : 1);
    const int rows_per_block = (64 / lpr) * 4 * 4;
    int64_t blocks = ((int64_t)rr.total + rows_per_block - 1) / rows_per_block;
    static const bool xcd = !(getenv("PHL_XCD") && atoi(getenv("PHL_XCD")) == 0);
    int xcd_chunk = 0;
    if (xcd && blocks >= 64) {
        blocks = (blocks + 7) / 8 * 8;
        xcd_chunk = (int)(blocks / 8);
    }
    const unsigned grid = (unsigned)blocks;
    dispatch_lpr(lpr, [&](auto L) {
        constexpr int LPR = decltype(L)::value;
        if (v4) k_blur<4, LPR><<<dim3(grid), dim3(256), 0, st>>>(vin, vout, nbr, M, vd, xcd_chunk, rr);
        else k_blur<1, LPR><<<dim3(grid), dim3(256), 0, st>>>(vin, vout, nbr, M, vd, xcd_chunk, rr);
    });
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}

int phl_launch_rows(bool scatter, float *vert, int vd, const int64_t *idx, int64_t k, float *buf, int64_t buf_rs, hipStream_t st)
{
    if (k == 0 || vd == 0) return PHL_OK;
    if (vd % 4 || buf_rs % 4 || !aligned16(vert) || !aligned16(buf)) {
        phl_set_error("row gather/scatter: needs vd %% 4 == 0 and 16-byte aligned rows");
        return PHL_ERR_UNSUPPORTED;
    }
    const int lpr = pick_lpr(vd, 4);
    const unsigned grid = (unsigned)((k + (64 / lpr) * 4 - 1) / ((64 / lpr) * 4));
    dispatch_lpr(lpr, [&](auto L) {
        constexpr int LPR = decltype(L)::value;
        if (scatter) k_scatter_add_rows<LPR><<<dim3(grid), dim3(256), 0, st>>>(vert, vd, idx, k, buf, buf_rs);
        else k_gather_rows<LPR><<<dim3(grid), dim3(256), 0, st>>>(vert, vd, idx, k, buf, buf_rs);
    });
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}

int phl_launch_blur2(const phl_lattice *lat, int pair, const float *vin, float *vout, int vd, hipStream_t st, bool restricted)
{
    const int M = (int)lat->M;
    if (M == 0 || vd == 0) return PHL_OK;
    const blur_rows_t rr = rows_of(lat, 2 * pair + 1, restricted);      // out = blur_b(blur_a(in)): the rows axis b's output is read at
    if (rr.total == 0) return PHL_OK;
    const int4 *nb2 = reinterpret_cast<const int4 *>(lat->nbr2) + (int64_t)pair * M * 2;
    const bool v4 = (vd % 4 == 0) && aligned16(vin) && aligned16(vout);
    const int lpr = pick_lpr(vd, v4 ? 4 : 1);
    constexpr int U2 = 1;   // row groups in flight per wave: 9 row loads each; 1 measured best (2: +3 %, 4: +15 %)
    const int rows_per_block = (64 / lpr) * U2 * 4;
    int64_t blocks = ((int64_t)rr.total + rows_per_block - 1) / rows_per_block;
    static const bool xcd = !(getenv("PHL_XCD") && atoi(getenv("PHL_XCD")) == 0);
    int xcd_chunk = 0;
    if (xcd && blocks >= 64) {
        blocks = (blocks + 7) / 8 * 8;
        xcd_chunk = (int)(blocks / 8);
    }
    const unsigned grid = (unsigned)blocks;
    dispatch_lpr(lpr, [&](auto L) {
        constexpr int LPR = decltype(L)::value;
        if (v4) k_blur2<4, LPR, U2><<<dim3(grid), dim3(256), 0, st>>>(vin, vout, nb2, M, vd, xcd_chunk, rr);
        else k_blur2<1, LPR, U2><<<dim3(grid), dim3(256), 0, st>>>(vin, vout, nb2, M, vd, xcd_chunk, rr);
    });
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}

int phl_launch_slice(const phl_lattice *lat, const float *vert, int vd, float *out, int64_t out_rs, const float *sub,
                     int64_t sub_rs, unsigned flags, hipStream_t st)
{
    const int64_t n = lat->n;
    if (n == 0 || vd == 0) return PHL_OK;
    const int dp1 = lat->d + 1;
    // the reference's constant, computed as it computes it: 1 + powf(2, -d)   (:480)
    const float cdiv = 1 + powf(2, -lat->d);
    const float rcdiv = 1.0f / cdiv;
    const bool v4 = (vd % 4 == 0) && (out_rs % 4 == 0) && aligned16(vert) && aligned16(out) &&
                    (!sub || ((sub_rs % 4 == 0) && aligned16(sub)));
    const int lpr = pick_lpr(vd, v4 ? 4 : 1);
    const unsigned grid = grid_for(n, 64 / lpr);
    const bool exact = (flags & PHL_FILTER_EXACT) != 0;
    dispatch_lpr(lpr, [&](auto L) {
        constexpr int LPR = decltype(L)::value;
#define PHL_SLICE_ARGS vert, vd, lat->replay, dp1, n, out, out_rs, sub, sub_rs, cdiv, rcdiv
        if (v4 && exact) k_slice<4, LPR, true><<<dim3(grid), dim3(256), 0, st>>>(PHL_SLICE_ARGS);
        else if (v4) k_slice<4, LPR, false><<<dim3(grid), dim3(256), 0, st>>>(PHL_SLICE_ARGS);
        else if (exact) k_slice<1, LPR, true><<<dim3(grid), dim3(256), 0, st>>>(PHL_SLICE_ARGS);
        else k_slice<1, LPR, false><<<dim3(grid), dim3(256), 0, st>>>(PHL_SLICE_ARGS);
#undef PHL_SLICE_ARGS
    });
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}

// Test hook (phl_debug_slow_copy): a copy by a FEW workgroups -- long-running, next to no HBM bandwidth, a handful of wave
// slots -- the footprint of a point-to-point transfer kernel, for rehearsing how much of an exchange a schedule hides.
__global__ __launch_bounds__(256) void k_slow_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, int64_t n4, int repeat)
{
    for (int r = 0; r < repeat; r++)
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

extern "C" int phl_debug_slow_copy(const float *src, float *dst, int64_t n_floats, int workgroups, int repeat, phl_stream stream)
{
    if (!src || !dst || n_floats < 0 || n_floats % 4 || workgroups < 1 || repeat < 1) { phl_set_error("phl_debug_slow_copy: bad arguments"); return PHL_ERR_INVALID; }
    if (n_floats == 0) return PHL_OK;
    hipLaunchKernelGGL(k_slow_copy, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4 *>(src),
                       reinterpret_cast<float4 *>(dst), n_floats / 4, repeat);
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}

int phl_launch_copy2d(const float *src, int64_t srs, int64_t scs, float *dst, int64_t drs, int64_t dcs, int64_t rows,
                      int cols, hipStream_t st)
{
    if (rows == 0 || cols == 0) return PHL_OK;
    if (scs == 1 && dcs == 1 && cols >= 4) {     // rows to rows: no transpose tile
        int64_t blocks = (rows + 15) / 16;
        if (blocks > 256 * 32) blocks = 256 * 32;
        hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)blocks), dim3(256), 0, st, src, srs, dst, drs, rows, cols);
        PHL_HIP(hipGetLastError());
        return PHL_OK;
    }
    dim3 grid((unsigned)((rows + 63) / 64), (unsigned)((cols + 63) / 64));
    hipLaunchKernelGGL(k_copy2d, grid, dim3(256), 0, st, src, srs, scs, dst, drs, dcs, rows, cols);
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}
