// phl_tiles.hip -- pixel chunks ("tiles") and the LDS-staged splat / slice kernels.
//
// Why: in the plain gather kernels (phl_filter.hip) every pixel row is re-read by its d+1
// vertices (splat) and every vertex row by all pixels of its cell (slice).  Those re-reads
// come out of L2 / the Infinity Cache at fabric rate and bound the kernels (round-1 profile:
// splat moved 19 GB for 3.2 GB of input).  A chunk is a set of <= P pixels that are close in
// feature space, so they share most of their lattice vertices; a workgroup stages the chunk's
// rows ONCE in LDS (channel slab by channel slab) and all reuse happens there:
//
//   k_splat_tiled   stage Q[chunk pixels][slab] -> every local vertex sums its pixel-sorted
//                   segment out of LDS -> the vertex row if this chunk is the vertex's only
//                   contributor, else a partial row; k_splat_reduce then adds a vertex's
//                   partial rows in ascending chunk order (deterministic, no float atomics)
//   k_slice_tiled   stage vert[chunk's local vertices][slab] -> every pixel gathers its d+1
//                   rows from LDS with the reference's per-term arithmetic (bit-exact)
//
// Chunks need no image geometry: pixels are ordered by the cell of a uniform 2-D grid laid over
// the two feature dimensions with the widest range (for an image: x/sigma, y/sigma ->
// sqrt(P) x sqrt(P) pixel tiles), then cut into runs of P.  Any other data still works, only
// with less sharing; the host falls back to the gather kernels when sharing is poor.
//
// Reference semantics are those of crf/lattice/lite/permutohedral.h:454-455 (splat
// accumulate) and :473-483 (slice); summation ORDER of multi-chunk vertices differs from the
// reference's pixel order (partial sums), hence results agree to fp32 rounding (~1e-7), not
// bit for bit; PHL_FILTER_EXACT selects the pixel-ordered gather splat instead.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#include "phl_device_utils.h"

namespace {

// ---- feature ranges --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_minmax(const float *__restrict__ ref, int64_t rs, int64_t cs, int64_t n, int d,
                                                float *__restrict__ out /* [grid][d][2] */)
{
    // one pass over the pixels, all d features of a pixel by the same thread (pixel-major features: every cache line is
    // touched once, not d times)
    __shared__ float smin[4][PHL_MAX_D], smax[4][PHL_MAX_D];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float lo[PHL_MAX_D], hi[PHL_MAX_D];
#pragma unroll
    for (int i = 0; i < PHL_MAX_D; i++) { lo[i] = INFINITY; hi[i] = -INFINITY; }
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int i = 0; i < PHL_MAX_D; i++)
            if (i < d) {
                const float v = ref[p * rs + i * cs];
                lo[i] = fminf(lo[i], v);
                hi[i] = fmaxf(hi[i], v);
            }
    }
#pragma unroll
    for (int i = 0; i < PHL_MAX_D; i++)
        if (i < d) {
            float a = lo[i], b = hi[i];
            for (int o = 32; o > 0; o >>= 1) {
                a = fminf(a, __shfl_xor(a, o));
                b = fmaxf(b, __shfl_xor(b, o));
            }
            if (lane == 0) { smin[w][i] = a; smax[w][i] = b; }
        }
    __syncthreads();
    if ((int)threadIdx.x < d) {
        float a = smin[0][threadIdx.x], b = smax[0][threadIdx.x];
        for (int k = 1; k < 4; k++) { a = fminf(a, smin[k][threadIdx.x]); b = fmaxf(b, smax[k][threadIdx.x]); }
        out[((int64_t)blockIdx.x * d + threadIdx.x) * 2 + 0] = a;
        out[((int64_t)blockIdx.x * d + threadIdx.x) * 2 + 1] = b;
    }
}

__global__ __launch_bounds__(256) void k_cell_ids(const float *__restrict__ ref, int64_t rs, int64_t cs, int64_t n, int da,
                                                  int db, float lo_a, float lo_b, float inv_t, int nca, int ncb,
                                                  int *__restrict__ cell)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = p < n;
    const int64_t pc = active ? p : n - 1;
    int ca = (int)((ref[pc * rs + da * cs] - lo_a) * inv_t);
    ca = min(max(ca, 0), nca - 1);
    int cb = 0;
    if (db >= 0) {
        cb = (int)((ref[pc * rs + db * cs] - lo_b) * inv_t);
        cb = min(max(cb, 0), ncb - 1);
    }
    const int c = cb * nca + ca;   // the wider dimension runs fastest inside a row of cells
    if (active) cell[p] = c;
}

// ---- locality renumbering of the vertices ----------------------------------------------------------------
// First-touch ids follow the pixel order: for an image, raster order, so a vertex and the blur neighbours a few
// pixels above / below it are a few image ROWS apart in every [M][vd] array (2 MB at 2048 pixels per row) and
// the 9-row stencil of a blur pass outruns the 4 MiB L2 of an XCD (58 % hits).  Internally the vertices are
// therefore numbered strip by strip: the grid of the two widest feature dimensions is cut into 8 strips along
// the wider one, a vertex's home cell is the cell of its first-touch pixel, and vertices are
// ordered by (strip, cell row, cell inside the strip), first-touch order inside a cell (stable sort).  A blur
// launch gives every XCD a contiguous eighth of the ids = about one strip, walked row by row, so that both the
// own-row stream and the stencil stay local.  Public introspection keeps the reference's first-touch numbering
// (phl_get_keys & co translate); rows of caller-visible vertex buffers are in internal order
// (phl_get_vertex_order).
__global__ __launch_bounds__(256) void k_vertex_home(const phl_replay_t *__restrict__ replay, int N, int dp1,
                                                     const int *__restrict__ cell, int *vhome)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const int v = replay[e].vid, c = cell[e / dp1];
    // the value only ever decreases, so a stale read can at worst cause a redundant atomic; in pixel order the
    // first toucher usually already holds the minimum and most candidates skip the atomic
    if (c < *reinterpret_cast<volatile int *>(&vhome[v])) atomicMin(&vhome[v], c);
}

// home cell from the first-touch candidate (the common case: one thread per vertex, no atomics)
__global__ __launch_bounds__(256) void k_vertex_home_first(const int *__restrict__ vfirst, int M, int dp1,
                                                           const int *__restrict__ cell, int *__restrict__ vhome)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < M) vhome[v] = cell[vfirst[v] / dp1];
}

__global__ __launch_bounds__(256) void k_strip_key(const int *__restrict__ vhome, int M, int nca, int ncb, int stripw,
                                                   int *__restrict__ key)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= M) return;
    const int c = vhome[v];
    const int ca = c % nca, cb = c / nca;
    key[v] = ((ca / stripw) * ncb + cb) * stripw + ca % stripw;
}

__global__ __launch_bounds__(256) void k_iota_tail(int *__restrict__ p, int first, int end)
{
    const int i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < end) p[i] = i;
}

__global__ __launch_bounds__(256) void k_invert_perm(const int *__restrict__ perm, int M, int *__restrict__ inv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) inv[perm[i]] = i;
}

__global__ __launch_bounds__(256) void k_permute_keys(const int16_t *__restrict__ in, const int *__restrict__ ft_of_int, int M,
                                                      int d, int16_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * d) return;
    const int v = (int)(i / d), c = (int)(i - (int64_t)v * d);
    out[i] = in[(int64_t)ft_of_int[v] * d + c];
}

__global__ __launch_bounds__(256) void k_relabel_replay(phl_replay_t *__restrict__ replay, int N, const int *__restrict__ int_of_ft)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < N) replay[e].vid = int_of_ft[replay[e].vid];
}

// ---- per-chunk structure: one workgroup groups the chunk's entries by vertex in LDS ----------------
// entry e = k*(d+1)+r of the chunk (k-th pixel in chunk order, remainder r).  Wanted: the entries grouped by vertex
// with ascending e inside a group, i.e. ascending pixel: exactly the per-vertex segment the splat needs.
//   1. the chunk's distinct vertices get dense ids 0..nv-1 through an LDS hash table (slot order);
//   2. a STABLE least-significant-digit radix sort of the entries by dense id, four bits a pass -- ceil(log2 nv)/4
//      passes: two for the ~50-250 local vertices of an image chunk, where a comparison sort of (vertex, entry) keys
//      took 66 compare-exchange stages.  A thread owns PER consecutive entries; its digit histogram is a packed
//      64-bit register (sixteen 4-bit counts), the workgroup-wide prefix per (digit, thread) a wavefront scan on DPP
//      row shifts over four words of four 16-bit fields.
// WRITE=false only counts the distinct vertices.
#define PHL_DPP_ADD(x, ctrl, rowmask) x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)(x), ctrl, rowmask, 0xF, false)
__device__ __forceinline__ unsigned wave_inclusive_scan_u32(unsigned x)
{
    PHL_DPP_ADD(x, 0x111, 0xF);     // row_shr:1
    PHL_DPP_ADD(x, 0x112, 0xF);     // row_shr:2
    PHL_DPP_ADD(x, 0x114, 0xF);     // row_shr:4
    PHL_DPP_ADD(x, 0x118, 0xF);     // row_shr:8
    PHL_DPP_ADD(x, 0x142, 0xA);     // row_bcast:15 into rows 1 and 3
    PHL_DPP_ADD(x, 0x143, 0xC);     // row_bcast:31 into rows 2 and 3
    return x;
}
#undef PHL_DPP_ADD

template <int SORTN, bool WRITE>
__global__ __launch_bounds__(256) void k_chunk_group(const int *__restrict__ pix_order, int n, int P, int dp1,
                                                     const phl_replay_t *__restrict__ replay, int *__restrict__ nv_out,
                                                     const int *__restrict__ vptr, int stride, int *__restrict__ slot_vert,
                                                     int2 *__restrict__ seg_rng, phl_contrib_t *__restrict__ seg,
                                                     unsigned short *__restrict__ lidx, const int *__restrict__ nv_known,
                                                     int skip_le)
{
    // nv_known: the chunks' vertex counts from k_chunk_masks -- chunks with at most skip_le are done already
    if (nv_known && nv_known[blockIdx.x] <= skip_le) return;
    // vptr != null: slots go to their final place vptr[c] + local index.  vptr == null (first and
    // normally only pass): slots go to a scratch area with a fixed `stride` per chunk (local indices
    // beyond it are dropped -- the host then repeats the pass with the real offsets), and the number of
    // local vertices is reported in nv_out.
    constexpr int PER = SORTN / 256;       // consecutive entries owned by a thread
    constexpr int HT = 2 * SORTN;          // hash slots (load <= 1/2)
    constexpr int HB = SORTN == 2048 ? 12 : (SORTN == 1024 ? 11 : 10);
    static_assert(PER <= 8, "the per-thread digit histogram has 4-bit counts");
    __shared__ unsigned keys[SORTN];       // (dense id << 11) | entry
    __shared__ int tab[HT + 8];            // slot -> vertex id, then slot -> dense id; later hpos | newidx (shorts)
    __shared__ int lvid[SORTN];            // dense id -> vertex id
    __shared__ unsigned wtot[4][8];
    __shared__ int lbin[258];              // histogram over segment lengths 1..P (P <= 256)
    const int c = blockIdx.x;
    const int base = c * P;
    const int cnt = min(P, n - base);
    const int E = cnt * dp1;
    const int i0 = threadIdx.x * PER;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int j = threadIdx.x; j < HT; j += 256) tab[j] = -1;
    int vid[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int e = i0 + u;
        vid[u] = -1;
        if (e < E) {
            const int k = e / dp1, rr = e - k * dp1;
            const int p = pix_order[base + k];
            vid[u] = replay[(int64_t)p * dp1 + rr].vid;
        }
    }
    __syncthreads();
    int slot[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) {
        slot[u] = 0;
        if (vid[u] >= 0) {
            unsigned h = ((unsigned)vid[u] * 2654435761u) >> (32 - HB);
            for (;;) {
                const int prev = atomicCAS(&tab[h], -1, vid[u]);
                if (prev == -1 || prev == vid[u]) break;
                h = (h + 1) & (HT - 1);
            }
            slot[u] = (int)h;
        }
    }
    __syncthreads();
    int nv;
    {
        constexpr int SPT = HT / 256;      // slots owned by a thread
        int occ = 0;
#pragma unroll
        for (int j = 0; j < SPT; j++) occ += tab[threadIdx.x * SPT + j] >= 0 ? 1 : 0;
        int id = block_exclusive_scan(occ, &nv);
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            const int sidx = threadIdx.x * SPT + j;
            const int v = tab[sidx];
            if (v >= 0) {
                lvid[id] = v;
                tab[sidx] = id++;
            }
        }
    }
    __syncthreads();
    if (nv_out && threadIdx.x == 0) nv_out[c] = nv;
    if (!WRITE) return;
    unsigned r[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) r[u] = vid[u] >= 0 ? (((unsigned)tab[slot[u]] << 11) | (unsigned)(i0 + u)) : ~0u;
    const int bits = nv > 1 ? 32 - __clz(nv - 1) : 0;
    for (int sh = 11; sh < 11 + bits; sh += 4) {
        // digit histogram of the thread's entries (4-bit counts) and every entry's rank among the thread's equal digits
        unsigned long long hist = 0;
        int lr[PER], dg[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) {
            dg[u] = (int)((r[u] >> sh) & 15u);
            lr[u] = (int)((hist >> (4 * dg[u])) & 15ull);
            if (r[u] != ~0u) hist += 1ull << (4 * dg[u]);
        }
        // eight words of two 16-bit counts (digits 2j, 2j+1); inclusive scan over the workgroup's threads
        unsigned own[8], inc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned x = (unsigned)(hist >> (8 * j)) & 0xFFu;
            own[j] = (x & 15u) | ((x >> 4) << 16);
            inc[j] = wave_inclusive_scan_u32(own[j]);
        }
        if (lane == 63)                                   // (the previous pass's reads of wtot lie behind its last barrier)
#pragma unroll
            for (int j = 0; j < 8; j++) wtot[wv][j] = inc[j];
        __syncthreads();
        unsigned pos[8];
        unsigned run = 0;                                 // exclusive scan over the digits (counts < 2^16)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            unsigned before = 0, tot = 0;
#pragma unroll
            for (int w2 = 0; w2 < 4; w2++) {
                const unsigned t = wtot[w2][j];
                tot += t;
                if (w2 < wv) before += t;
            }
            const unsigned lo = run, hi = run + (tot & 0xFFFFu);
            run = hi + (tot >> 16);
            pos[j] = (inc[j] - own[j]) + before + (lo | (hi << 16));
        }
#pragma unroll
        for (int u = 0; u < PER; u++) {
            if (r[u] == ~0u) continue;
            unsigned q = pos[0];
#pragma unroll
            for (int j = 1; j < 8; j++) q = (dg[u] >> 1) == j ? pos[j] : q;
            const int rank = (int)((q >> (16 * (dg[u] & 1))) & 0xFFFFu) + lr[u];
            keys[rank] = r[u];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; u++) r[u] = (i0 + u) < E ? keys[i0 + u] : ~0u;
    }
    if (bits == 0) {
#pragma unroll
        for (int u = 0; u < PER; u++)
            if (i0 + u < E) keys[i0 + u] = r[u];
    }
    __syncthreads();                                      // (tab is dead from here: hpos | newidx take its place)
    // Local vertices are renumbered by DESCENDING segment length (counting sort in LDS): the
    // splat kernel hands neighbouring local vertices to the lane groups of one wavefront, which
    // then run loops of nearly equal length, and takes groups longest-first.
    unsigned short *hpos = reinterpret_cast<unsigned short *>(tab);   // [nv + 1] start of the segment of dense id j
    unsigned short *newidx = hpos + SORTN + 2;                         // [nv] dense id -> length-order index
    const int total = nv;
    const int64_t vbase = vptr ? (int64_t)vptr[c] : (int64_t)c * stride;
    const int vcap = vptr ? SORTN : stride;
    const int64_t ebase = (int64_t)base * dp1;
    for (int j = threadIdx.x; j < 258; j += 256) lbin[j] = 0;
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int i = i0 + u;
        if (i < E && (i == 0 || (keys[i] >> 11) != (keys[i - 1] >> 11))) hpos[keys[i] >> 11] = (unsigned short)i;
    }
    if (threadIdx.x == 0) hpos[total] = (unsigned short)E;
    __syncthreads();
    for (int j = threadIdx.x; j < total; j += 256) atomicAdd(&lbin[256 - min(hpos[j + 1] - hpos[j], 256)], 1);   // bin 0 = longest
    __syncthreads();
    if (threadIdx.x < 64) {           // exclusive scan of the 257 bins by one wavefront
        int carry = 0;
        for (int b0 = 0; b0 < 257; b0 += 64) {
            const int b = b0 + (int)threadIdx.x;
            const int x = b < 257 ? lbin[b] : 0;
            int incl = x;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int y = __shfl_up(incl, o);
                if ((int)threadIdx.x >= o) incl += y;
            }
            if (b < 257) lbin[b] = carry + incl - x;
            carry += __shfl(incl, 63);
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < total; j += 256)
        newidx[j] = (unsigned short)atomicAdd(&lbin[256 - min(hpos[j + 1] - hpos[j], 256)], 1);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int i = i0 + u;
        if (i >= E) break;
        const int li = (int)(keys[i] >> 11);
        const int e = (int)(keys[i] & 2047u);
        const bool head = (i == 0) || li != (int)(keys[i - 1] >> 11);
        const int k = e / dp1, rr = e - k * dp1;
        const int p = pix_order[base + k];
        if (head && newidx[li] < vcap) {
            const int64_t sl = vbase + newidx[li];
            slot_vert[sl] = lvid[li];
            seg_rng[sl] = make_int2((int)(ebase + i), (int)(ebase + hpos[li + 1]));
        }
        phl_contrib_t sg;
        sg.pixel = k;
        sg.w = replay[(int64_t)p * dp1 + rr].w;
        seg[ebase + i] = sg;
        lidx[ebase + e] = (unsigned short)newidx[li];
    }
}

// The same grouping without a sort, for chunks with at most NVC distinct vertices (every chunk of an image): a pixel
// holds a vertex at most once (the d+1 vertices of a simplex are distinct), so a local vertex's segment is a SET of chunk
// pixels -- one bit per pixel, eight words per vertex (P <= 256), set with one LDS atomic per entry.  An entry's place in
// its segment is the number of set bits below its pixel (a per-word prefix per vertex + one popcount), the segment's
// start the scan of the segment lengths.  ~400 vector instructions a wavefront where the radix passes of k_chunk_group
// take ~1,700 (the kernel is bound by instruction issue).  A chunk with more vertices only reports its count; the host
// then runs k_chunk_group on those chunks.
constexpr int NVC = 256;               // (the kernel waits on LDS / L2 round trips: a small footprint buys workgroups per CU)

template <int SORTN, int HTX>
__global__ __launch_bounds__(256) void k_chunk_masks(const int *__restrict__ pix_order, int n, int P, int dp1,
                                                     const phl_replay_t *__restrict__ replay, int *__restrict__ nv_out,
                                                     const int *__restrict__ vptr, int stride, int *__restrict__ slot_vert,
                                                     int2 *__restrict__ seg_rng, phl_contrib_t *__restrict__ seg,
                                                     unsigned short *__restrict__ lidx)
{
    constexpr int PER = SORTN / 256;       // consecutive entries owned by a thread
    constexpr int HT = HTX * SORTN;        // hash slots: load <= 3/4 (the host picks HTX = 2 where P(d+1) > 3/4 SORTN)
    constexpr int HB = (SORTN == 2048 ? 11 : (SORTN == 1024 ? 10 : 9)) + (HTX == 2 ? 1 : 0);
    constexpr int TABN = (HT > NVC * 8 ? HT : NVC * 8) + 8;
    __shared__ __attribute__((aligned(16))) int tab[TABN];   // slot -> vertex id, then slot -> dense id; then the pixel masks [nv][8]
    __shared__ int lvid[NVC];              // dense id -> vertex id
    __shared__ __attribute__((aligned(8))) unsigned char cum[NVC][8];   // set bits of a vertex's mask below word w (<= 224)
    __shared__ unsigned short startv[NVC + 2];   // segment start of dense id j (entries), [nv] = E
    __shared__ unsigned short newidx[NVC]; // dense id -> length-order index
    __shared__ int lbin[258];              // histogram over segment lengths 1..P (P <= 256)
    const int c = blockIdx.x;
    const int base = c * P;
    const int cnt = min(P, n - base);
    const int E = cnt * dp1;
    const int i0 = threadIdx.x * PER;
    for (int j = threadIdx.x; j < HT; j += 256) tab[j] = -1;
    int vid[PER], kk[PER], rrr[PER];
    float wgt[PER];
    {
        int k = i0 / dp1, rr = i0 - k * dp1;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            vid[u] = -1;
            wgt[u] = 0.f;
            kk[u] = k;
            rrr[u] = rr;
            if (i0 + u < E) {
                const int p = pix_order[base + k];
                const phl_replay_t rp = replay[(int64_t)p * dp1 + rr];
                vid[u] = rp.vid;
                wgt[u] = rp.w;
            }
            if (++rr == dp1) { rr = 0; k++; }
        }
    }
    __syncthreads();
    int slot[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) {
        slot[u] = 0;
        if (vid[u] >= 0) {
            unsigned h = ((unsigned)vid[u] * 2654435761u) >> (32 - HB);
            for (;;) {
                const int prev = atomicCAS(&tab[h], -1, vid[u]);
                if (prev == -1 || prev == vid[u]) break;
                h = (h + 1) & (HT - 1);
            }
            slot[u] = (int)h;
        }
    }
    __syncthreads();
    int nv;
    {
        constexpr int SPT = HT / 256;      // slots owned by a thread
        int occ = 0;
#pragma unroll
        for (int j = 0; j < SPT; j++) occ += tab[threadIdx.x * SPT + j] >= 0 ? 1 : 0;
        int id = block_exclusive_scan(occ, &nv);
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            const int sidx = threadIdx.x * SPT + j;
            const int v = tab[sidx];
            if (v >= 0) {
                if (id < NVC) lvid[id] = v;
                tab[sidx] = id++;
            }
        }
    }
    __syncthreads();
    if (nv_out && threadIdx.x == 0) nv_out[c] = nv;
    if (nv > NVC) return;                  // (workgroup-uniform) left to k_chunk_group
    int id[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) id[u] = vid[u] >= 0 ? tab[slot[u]] : 0;
    __syncthreads();                       // tab is dead: the masks take its place
    unsigned *mask = reinterpret_cast<unsigned *>(tab);
    for (int j = threadIdx.x; j < nv * 8; j += 256) mask[j] = 0u;
    for (int j = threadIdx.x; j < 258; j += 256) lbin[j] = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; u++)
        if (vid[u] >= 0) atomicOr(&mask[id[u] * 8 + (kk[u] >> 5)], 1u << (kk[u] & 31));
    __syncthreads();
    // per vertex (thread v): prefix of set bits per word, segment length
    int len = 0;
    if ((int)threadIdx.x < nv) {
        const int v = threadIdx.x;
        const uint4 m0 = *reinterpret_cast<const uint4 *>(mask + v * 8), m1 = *reinterpret_cast<const uint4 *>(mask + v * 8 + 4);
        const unsigned mw[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
        unsigned c4[2] = {0u, 0u};
#pragma unroll
        for (int w = 0; w < 8; w++) {
            c4[w >> 2] |= (unsigned)len << (8 * (w & 3));
            len += __popc(mw[w]);
        }
        *reinterpret_cast<uint2 *>(&cum[v][0]) = make_uint2(c4[0], c4[1]);
        atomicAdd(&lbin[256 - min(len, 256)], 1);          // bin 0 = longest
    }
    {
        int tot;
        const int ex = block_exclusive_scan(len, &tot);
        if ((int)threadIdx.x < nv) startv[threadIdx.x] = (unsigned short)ex;
        if (threadIdx.x == 0) startv[nv] = (unsigned short)E;
    }
    __syncthreads();
    // Local vertices are renumbered by DESCENDING segment length (counting sort in LDS): the
    // splat kernel hands neighbouring local vertices to the lane groups of one wavefront, which
    // then run loops of nearly equal length, and takes groups longest-first.
    if (threadIdx.x < 64) {           // exclusive scan of the 257 bins by one wavefront
        int carry = 0;
        for (int b0 = 0; b0 < 257; b0 += 64) {
            const int b = b0 + (int)threadIdx.x;
            const int x = b < 257 ? lbin[b] : 0;
            int incl = x;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int y = __shfl_up(incl, o);
                if ((int)threadIdx.x >= o) incl += y;
            }
            if (b < 257) lbin[b] = carry + incl - x;
            carry += __shfl(incl, 63);
        }
    }
    __syncthreads();
    const int64_t vbase = vptr ? (int64_t)vptr[c] : (int64_t)c * stride;
    const int vcap = vptr ? SORTN : stride;
    const int64_t ebase = (int64_t)base * dp1;
    if ((int)threadIdx.x < nv) {
        const int v = threadIdx.x;
        const int ni = atomicAdd(&lbin[256 - min(len, 256)], 1);
        newidx[v] = (unsigned short)ni;
        if (ni < vcap) {
            const int64_t sl = vbase + ni;
            slot_vert[sl] = lvid[v];
            seg_rng[sl] = make_int2((int)(ebase + startv[v]), (int)(ebase + startv[v] + len));
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; u++) {
        if (vid[u] < 0) continue;
        const int v = id[u], k = kk[u];
        const int pos = (int)startv[v] + (int)cum[v][k >> 5] + __popc(mask[v * 8 + (k >> 5)] & ((1u << (k & 31)) - 1u));
        phl_contrib_t sg;
        sg.pixel = k;
        sg.w = wgt[u];
        seg[ebase + pos] = sg;
        lidx[ebase + i0 + u] = newidx[v];
    }
}

// scratch slot records [chunk][stride] -> compact [vptr[chunk] + i]
__global__ __launch_bounds__(256) void k_compact_slots(const int *__restrict__ vptr, int nchunks, int stride,
                                                       const int *__restrict__ t_vert, const int2 *__restrict__ t_rng,
                                                       int *__restrict__ slot_vert, int2 *__restrict__ seg_rng)
{
    const int c = blockIdx.x;
    const int b = vptr[c], nv = vptr[c + 1] - b;
    for (int i = threadIdx.x; i < nv; i += 256) {
        slot_vert[b + i] = t_vert[(int64_t)c * stride + i];
        seg_rng[b + i] = t_rng[(int64_t)c * stride + i];
    }
}

__global__ __launch_bounds__(256) void k_count_slots(const int *__restrict__ slot_vert, int S, int *cnt)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S) atomicAdd(&cnt[slot_vert[s]], 1);
}

// the vertex -> slots list entries from the sort's permutation, and the slots' marks: bit 31 of slot_vert = this chunk is
// the vertex's only contributor (sole); the others are flagged for the partial buffer
__global__ __launch_bounds__(256) void k_contrib_and_sole(const int *__restrict__ perm, int *__restrict__ slot_vert, int S,
                                                          const int *__restrict__ vs_ptr, phl_contrib_t *__restrict__ vs,
                                                          int *__restrict__ multi)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    phl_contrib_t c;
    c.pixel = perm[s];
    c.w = 0.f;
    vs[s] = c;
    const int v = slot_vert[s];
    const bool sole = (vs_ptr[v + 1] - vs_ptr[v]) == 1;
    if (sole) slot_vert[s] = v | (int)0x80000000;
    multi[s] = sole ? 0 : 1;
}

// vertices fed by more than `long_list` chunks, appended in any order (k_splat_reduce_long gives each its own
// workgroup; the order only decides which starts first -- lists of more than 64 are sorted by length afterwards)
__global__ __launch_bounds__(256) void k_append_long(const int *__restrict__ vs_ptr, int M, int long_list, int *__restrict__ vlong,
                                                     int *__restrict__ count)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < M && vs_ptr[v + 1] - vs_ptr[v] > long_list) vlong[atomicAdd(count, 1)] = v;
}

// ---- hot kernels -----------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float div_c(float t, float c, float rc)
{
    float q = t * rc;
    float rem = __builtin_fmaf(-q, c, t);
    return __builtin_fmaf(rem, rc, q);
}
__device__ __forceinline__ float4 term4(float4 acc, float w, float4 v, float c, float rc)   // acc + (w*v)/c  (:480)
{
    return make_float4(acc.x + div_c(w * v.x, c, rc), acc.y + div_c(w * v.y, c, rc), acc.z + div_c(w * v.z, c, rc),
                       acc.w + div_c(w * v.w, c, rc));
}
__device__ __forceinline__ float4 fma4(float4 acc, float w, float4 v)
{
    return make_float4(__builtin_fmaf(w, v.x, acc.x), __builtin_fmaf(w, v.y, acc.y), __builtin_fmaf(w, v.z, acc.z),
                       __builtin_fmaf(w, v.w, acc.w));
}

// lane K of every 16-lane DPP row, to all lanes of that row (v_mov_b32_dpp row_newbcast:K -- VALU, no LDS traffic)
#define PHL_ROW_BCAST(x, K) ((unsigned)__builtin_amdgcn_update_dpp(0, (int)(x), 0x150 + (K), 0xF, 0xF, false))

// Eight consecutive entries of a segment, held one per lane in lanes K0..K0+7 of each DPP row (e.x = byte offset
// of the pixel's LDS row, e.y = weight bits): broadcast each, issue the eight 16-byte row reads back to
// back (one LDS round trip for all of them), then accumulate in entry order.
template <int K0>
__device__ __forceinline__ float4 sum8(float4 acc, const uint2 e, const char *rbase)
{
    float4 q[8];
    unsigned w[8];
#define PHL_E(k) w[k] = PHL_ROW_BCAST(e.y, K0 + k); q[k] = *reinterpret_cast<const float4 *>(rbase + PHL_ROW_BCAST(e.x, K0 + k));
    PHL_E(0) PHL_E(1) PHL_E(2) PHL_E(3) PHL_E(4) PHL_E(5) PHL_E(6) PHL_E(7)
#undef PHL_E
#pragma unroll
    for (int k = 0; k < 8; k++) acc = fma4(acc, __uint_as_float(w[k]), q[k]);
    return acc;
}

// One workgroup (TPB threads) per chunk.  LPRS lanes own one row of a channel slab of
// SL = 4*LPRS floats.  LDS: [rows x SL] staged rows | per-entry {LDS byte offset, weight} |
// chunk pixel ids | local segment pointers | destination of each local vertex.  Index data
// is staged ONCE per chunk; inside the slab loop global memory is touched only for value rows,
// and the rows of slab s+1 are prefetched into registers while slab s is being summed.
// Each of a wavefront's 64/LPRS lane groups owns one local vertex and sums its pixel-sorted
// segment sequentially out of LDS: a segmented reduction with one segment per lane group,
// deterministic, no atomics (details at the loop).
// splat: segments of at least this many entries are summed by a whole wavefront (see k_splat_tiled).  Two workgroups
// share a CU, so imbalance inside one is mostly absorbed by the other and what counts is total work: the
// cooperative form pads a segment to a multiple of 64 entries and pays a cross-row combine, so it is reserved for
// segments that would otherwise be the whole critical path (flat image regions: all 256 pixels in one vertex).
inline int long_seg()
{
    static const int v = getenv("PHL_LONG_SEG") ? atoi(getenv("PHL_LONG_SEG")) : 128;
    return v < 16 ? 16 : v;
}
constexpr int TPB = 512;     // slice workgroup
constexpr int TPB_S = 512;   // splat workgroup (1024 threads and 1 workgroup per CU measured no better)

template <int LPRS>
__global__ __launch_bounds__(TPB_S) void k_splat_tiled(const float *__restrict__ src, int64_t src_rs, int vd, int n, int P,
                                                     int dp1, int nv_cap, const int *__restrict__ pix_order,
                                                     const int *__restrict__ vptr, const int *__restrict__ slot_vert,
                                                     const int *__restrict__ slot_pidx, const int2 *__restrict__ seg_rng,
                                                     const phl_contrib_t *__restrict__ seg, float *__restrict__ vert,
                                                     float *__restrict__ partial, int nchunks, int xcd_chunk,
                                                     unsigned long long *__restrict__ tl, const int *__restrict__ chunk_list,
                                                     int nv_lo, int nv_hi, int long_seg, int nsets,
                                                     const float *__restrict__ fref, int64_t fref_rs, int64_t fref_cs,
                                                     int64_t out_rs)
{
    // nsets > 1 (the wide splat of the gradient w.r.t. the features, phl_filter_grad): every staged slab is summed
    // nsets times, set 0 with the barycentric weights w and set 1+k with w * fref[pixel][k] -- the splat of
    // src (x) ref[:, k] without ever forming that product in memory -- into channel block `set` of rows that are
    // out_rs = nsets * vd floats long.  nsets == 1: the plain splat (fref unused, out_rs = vd).
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SL = LPRS * 4;
    constexpr int G = TPB_S / LPRS;          // row groups of the workgroup (staging)
    constexpr int Q = 64 / LPRS;           // lane groups of a wavefront (entry-parallel)
    constexpr int NW = TPB_S / 64;
    constexpr int PF = (256 / G) < 8 ? (256 / G) : 8;   // rows prefetched per thread
    const int g = threadIdx.x / LPRS, l = threadIdx.x % LPRS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = lane / LPRS;
    // XCD-aware chunk order (see k_blur): neighbouring chunks share boundary vertices.  With a chunk list
    // (phl_splat_part: a subset of the chunks, `nchunks` = its length) the same order runs over list positions.
    const int ci = xcd_chunk > 0 ? (int)(blockIdx.x & 7) * xcd_chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (ci >= nchunks) return;
    const int c = chunk_list ? chunk_list[ci] : ci;
    // debug timeline (PHL_TIMELINE=file): 100 MHz wall-clock stamps per workgroup, tl == nullptr in normal runs
    unsigned long long *tlb = tl ? tl + (size_t)blockIdx.x * 8 : nullptr;
    if (tlb && threadIdx.x == 0) {
        tlb[0] = wall_clock64();
        tlb[7] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
    const int base = c * P;
    const int cnt = min(P, n - base);
    const int E = cnt * dp1;
    const int64_t ebase = (int64_t)base * dp1;
    const int vbase = vptr[c], nv = vptr[c + 1] - vbase;
    // chunk classes (phl_launch_splat_tiled): this launch's LDS is sized for chunks with nv_lo < nv <= nv_hi local
    // vertices; the others belong to the launch of another class
    if (nv <= nv_lo || nv > nv_hi) return;
    float *rows = lds;                                             // [P][SL] staged pixel rows + one row of zeros
    uint2 *ent = reinterpret_cast<uint2 *>(lds + (size_t)(P + 1) * SL);
    int2 *meta = reinterpret_cast<int2 *>(ent + P * dp1);          // [nv_cap] {seg begin | seg end << 16, destination row}
    int *pixl = reinterpret_cast<int *>(meta + nv_cap);           // [P]
    int *ctr = pixl + P;                                           // two work counters, used by alternate slabs; [2] = #long segments
    // Loads are issued UNCONDITIONALLY from clamped (always valid) addresses and only the LDS
    // stores are predicated: a load under a divergent `if` makes hipcc wait vmcnt(0) per load.
    const int kclamp = cnt - 1;
    const char *rbase = reinterpret_cast<const char *>(rows) + l * 16;
    float4 pf[PF];
    // Prologue, arranged so that the chunk pays ONE dependent HBM round trip, not two: every
    // thread fetches the pixel ids of its own first PF rows itself and launches slab 0's row loads
    // on them, while the index data (entries, pixel ids, per-vertex records) streams into LDS.
    int prow[PF];
#pragma unroll
    for (int u = 0; u < PF; u++) prow[u] = pix_order[base + min(g + u * G, kclamp)];
    for (int e = threadIdx.x; e < E; e += TPB_S) {
        const phl_contrib_t s = seg[ebase + e];
        ent[e] = make_uint2((unsigned)(s.pixel * SL * 4), __float_as_uint(s.w));
    }
    for (int k = threadIdx.x; k < cnt; k += TPB_S) pixl[k] = pix_order[base + k];
    for (int i = threadIdx.x; i < nv; i += TPB_S) {
        const int2 rg = seg_rng[vbase + i];
        // destination: the vertex row (bit 31 set) if this chunk is the vertex's only contributor, else a partial row
        const int sv = slot_vert[vbase + i];
        meta[i] = make_int2((int)(rg.x - ebase) | ((int)(rg.y - ebase) << 16), sv < 0 ? sv : slot_pidx[vbase + i]);
        // Local vertices come in descending segment length (k_chunk_group), so the LONG ones (>= long_seg entries:
        // summed by a whole wavefront, below) are a prefix; its length is written by exactly one thread.
        const bool lng = rg.y - rg.x >= long_seg;
        if (i + 1 < nv) {
            const int2 rn = seg_rng[vbase + i + 1];
            if (lng && rn.y - rn.x < long_seg) ctr[2] = i + 1;
        } else if (lng) {
            ctr[2] = nv;
        }
        if (i == 0 && !lng) ctr[2] = 0;
    }
    if (nv == 0 && threadIdx.x == 0) ctr[2] = 0;
    if (threadIdx.x < 2) ctr[threadIdx.x] = NW;
    if (threadIdx.x < LPRS) st4(rows + (size_t)P * SL + threadIdx.x * 4, make_float4(0.f, 0.f, 0.f, 0.f));
    {
        const bool chok = l * 4 < vd;
        const int chc = chok ? l * 4 : 0;
#pragma unroll
        for (int u = 0; u < PF; u++) pf[u] = ld4(src + (int64_t)prow[u] * src_rs + chc);
#pragma unroll
        for (int u = 0; u < PF; u++)
            if (chok && g + u * G < cnt) st4(rows + (g + u * G) * SL + l * 4, pf[u]);
        if (PF * G < 256) {                // wide slabs: more rows per thread than the prefetch depth
            __syncthreads();               // pixl is needed for the remaining rows
            for (int k0 = g + PF * G; k0 < cnt; k0 += PF * G) {
#pragma unroll
                for (int u = 0; u < PF; u++) pf[u] = ld4(src + (int64_t)pixl[min(k0 + u * G, kclamp)] * src_rs + chc);
#pragma unroll
                for (int u = 0; u < PF; u++)
                    if (chok && k0 + u * G < cnt) st4(rows + (k0 + u * G) * SL + l * 4, pf[u]);
            }
        }
    }
    for (int c0 = 0; c0 < vd; c0 += SL) {
        const int ch = c0 + l * 4;
        const bool chok = ch < vd;
        __syncthreads();                   // rows of this slab are in LDS
        const int slab = c0 / SL;
        if (tlb && threadIdx.x == 0 && slab < 2) tlb[1 + 2 * slab] = wall_clock64();
        const int chn = ch + SL;
        const bool more = c0 + SL < vd;    // wave-uniform
        const bool chnok = more && chn < vd;
        const int chnc = chnok ? chn : 0;
        if (more) {
#pragma unroll
            for (int u = 0; u < PF; u++) pf[u] = ld4(src + (int64_t)pixl[min(g + u * G, kclamp)] * src_rs + chnc);
        }
        // Each of the wavefront's Q lane groups sums ONE local vertex (its pixel-sorted segment,
        // sequentially, out of LDS).  Local vertices are numbered by descending segment length
        // (k_chunk_group), so the Q vertices of a group have nearly equal loops, and the waves take
        // groups longest-first from a shared counter: no cross-lane combine, no padding, and the
        // per-vertex bookkeeping is paid once per Q vertices.
        for (int set = 0; set < nsets; set++) {
        const int phase = slab * nsets + set;
        if (nsets > 1) {
            if (set > 0) __syncthreads();                 // the previous set's sums are done with the entry weights
            for (int e = threadIdx.x; e < E; e += TPB_S) {
                const phl_contrib_t sg = seg[ebase + e];
                const float f = set ? fref[(int64_t)pixl[sg.pixel] * fref_rs + (int64_t)(set - 1) * fref_cs] : 1.f;
                ent[e].y = __float_as_uint(sg.w * f);
            }
        }
        if (threadIdx.x == 0) ctr[(phase + 1) & 1] = NW;  // re-arm the other counter for the next phase
        if (nsets > 1) __syncthreads();                   // weights of this set are in LDS
        int *slab_ctr = ctr + (phase & 1);
        // Work items, handed out longest-first: items [0, nlong) are single LONG vertices (LPRS >= 16 only), summed
        // by all Q lane groups of the wave together -- group q takes the 16-entry batches q, q+Q, ... of the
        // segment and the Q partial sums are combined across the DPP rows in a fixed order -- so that one 256-entry
        // segment (a flat image region: every pixel of the chunk in one vertex) costs a wave 4 batches, not 16;
        // the remaining items are groups of Q vertices, one per lane group.
        const int nlong = (LPRS >= 16 && Q > 1) ? ctr[2] : 0;
        const int nitems = nlong + (nv - nlong + Q - 1) / Q;
        for (int gi = wave; gi < nitems;) {
            const bool coop = gi < nlong;                       // wave-uniform
            const int i = coop ? gi : nlong + (gi - nlong) * Q + q;
            int2 m = make_int2(0, 0);
            if (i < nv) m = meta[i];
            int nxt = 0;
            if (lane == 0) nxt = atomicAdd(slab_ctr, 1);       // next item, fetched under this one's work
            const int s1 = (int)((unsigned)m.x >> 16);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int s = m.x & 0xFFFF;
            if constexpr (LPRS >= 16) {
                // A lane group is one or more whole 16-lane DPP rows.  Each row keeps SIXTEEN entries of its
                // vertex's segment in registers, one per lane (a single 8-byte LDS read per lane), and hands
                // them round with row broadcasts; the row reads of eight entries are in flight together.
                // The trip count is the longest of the wave's segments, so the loop is wave-uniform; a
                // shorter segment pads with weight 0 on the row of zeros.
                const int len = s1 - s;
                int lmax = __builtin_amdgcn_readlane(len, 0);
                if (Q > 1) lmax = max(lmax, __builtin_amdgcn_readlane(len, 32));
                if (Q > 2) lmax = max(max(lmax, __builtin_amdgcn_readlane(len, 16)), __builtin_amdgcn_readlane(len, 48));
                const int r16 = lane & 15;
                const unsigned zoff = (unsigned)P * SL * 4;
                const int elast = E - 1;
                // cooperative item: group q starts at batch q and strides over Q batches
                const int step = coop ? 16 * Q : 16;
                if (coop) s += 16 * q;
                uint2 e = ent[min(max(s + r16, 0), elast)];
                if (s + r16 >= s1) e = make_uint2(zoff, 0u);
                for (int b = 0; b < lmax; b += step) {
                    const int nx = s + b + step + r16;
                    uint2 en = ent[min(nx, elast)];              // next sixteen, under this batch's work
                    if (nx >= s1) en = make_uint2(zoff, 0u);
                    acc = sum8<0>(acc, e, rbase);
                    if (b + 8 < lmax) acc = sum8<8>(acc, e, rbase);
                    e = en;
                }
                if (Q > 1 && coop) {
                    // ((g0 + g1) + (g2 + g3)): every lane adds its partner's value, so all groups end with the same bits
                    acc = make_float4(acc.x + __shfl_xor(acc.x, LPRS), acc.y + __shfl_xor(acc.y, LPRS),
                                      acc.z + __shfl_xor(acc.z, LPRS), acc.w + __shfl_xor(acc.w, LPRS));
                    if (Q > 2)
                        acc = make_float4(acc.x + __shfl_xor(acc.x, 2 * LPRS), acc.y + __shfl_xor(acc.y, 2 * LPRS),
                                          acc.z + __shfl_xor(acc.z, 2 * LPRS), acc.w + __shfl_xor(acc.w, 2 * LPRS));
                }
            } else {
            // software pipeline: the index reads of batch b+1 are issued before the row reads of
            // batch b are consumed, so a batch costs one LDS round trip instead of two (LDS returns
            // in order: waiting for the rows leaves the next indices in flight)
            if (s + 4 <= s1) {
                uint2 e0 = ent[s], e1 = ent[s + 1], e2 = ent[s + 2], e3 = ent[s + 3];
                for (; s + 8 <= s1; s += 4) {
                    const float4 q0 = *reinterpret_cast<const float4 *>(rbase + e0.x);
                    const float4 q1 = *reinterpret_cast<const float4 *>(rbase + e1.x);
                    const float4 q2 = *reinterpret_cast<const float4 *>(rbase + e2.x);
                    const float4 q3 = *reinterpret_cast<const float4 *>(rbase + e3.x);
                    const uint2 n0 = ent[s + 4], n1 = ent[s + 5], n2 = ent[s + 6], n3 = ent[s + 7];
                    acc = fma4(acc, __uint_as_float(e0.y), q0);
                    acc = fma4(acc, __uint_as_float(e1.y), q1);
                    acc = fma4(acc, __uint_as_float(e2.y), q2);
                    acc = fma4(acc, __uint_as_float(e3.y), q3);
                    e0 = n0; e1 = n1; e2 = n2; e3 = n3;
                }
                const float4 q0 = *reinterpret_cast<const float4 *>(rbase + e0.x);
                const float4 q1 = *reinterpret_cast<const float4 *>(rbase + e1.x);
                const float4 q2 = *reinterpret_cast<const float4 *>(rbase + e2.x);
                const float4 q3 = *reinterpret_cast<const float4 *>(rbase + e3.x);
                acc = fma4(acc, __uint_as_float(e0.y), q0);
                acc = fma4(acc, __uint_as_float(e1.y), q1);
                acc = fma4(acc, __uint_as_float(e2.y), q2);
                acc = fma4(acc, __uint_as_float(e3.y), q3);
                s += 4;
            }
            for (; s < s1; s++) {
                const uint2 e0 = ent[s];
                acc = fma4(acc, __uint_as_float(e0.y), *reinterpret_cast<const float4 *>(rbase + e0.x));
            }
            }
            if (i < nv && chok && !(coop && q != 0)) {
                float *dst = m.y < 0 ? vert + (int64_t)(m.y & 0x7FFFFFFF) * out_rs : partial + (int64_t)m.y * out_rs;
                st4(dst + (int64_t)set * vd + ch, acc);
            }
            gi = __builtin_amdgcn_readfirstlane(nxt);
        }
        }   // sets
        __syncthreads();                   // everyone is done reading this slab
        if (tlb && threadIdx.x == 0 && slab < 2) tlb[2 + 2 * slab] = wall_clock64();
        if (tlb && threadIdx.x == 0 && !more) tlb[5] = wall_clock64();
        if (more) {
#pragma unroll
            for (int u = 0; u < PF; u++)
                if (chnok && g + u * G < cnt) st4(rows + (g + u * G) * SL + l * 4, pf[u]);
            for (int k0 = g + PF * G; k0 < cnt; k0 += PF * G) {   // rows beyond the prefetch depth (wide slabs)
#pragma unroll
                for (int u = 0; u < PF; u++) pf[u] = ld4(src + (int64_t)pixl[min(k0 + u * G, kclamp)] * src_rs + chnc);
#pragma unroll
                for (int u = 0; u < PF; u++)
                    if (chnok && k0 + u * G < cnt) st4(rows + (k0 + u * G) * SL + l * 4, pf[u]);
            }
        }
    }
}

// ---- wide splat, one sum phase per slab -------------------------------------------------------------------------------
// phl_filter_grad's first stage: vertex rows [set][vd], set 0 = splat of src, set 1+k = splat of src (x) fref[:, k]
// (NS = d+1 sets).  k_splat_tiled's nsets > 1 mode runs its sum phase NS times per slab, re-staging the entry weights in
// between: NS times the LDS row reads, which is what bounds it (1.56 ms at C2 against 0.42 for a plain splat).  Here
// every LDS row read feeds NS accumulators: a lane keeps its entry's barycentric weight AND its pixel's d features
// (gathered from the L2-resident feature rows one batch ahead), and per entry the weights w, w*f_0 ... w*f_{d-1} are
// formed from row broadcasts.  Same products, same summation order, same combine as the nsets > 1 mode: bitwise the
// same rows.  64-channel slabs (16 lanes per row) only; other widths keep the multi-phase mode.
// one entry (lane K of every DPP row) into all NS accumulators; K must be a literal for the DPP control field
template <int K, int NS>
__device__ __forceinline__ void wide_entry(float4 (&acc)[NS], const float4 q, const float w, const float (&f)[NS > 1 ? NS - 1 : 1])
{
    acc[0] = fma4(acc[0], w, q);
#define PHL_WJ(j)                                                                                            \
    if constexpr (j + 1 < NS) {                                                                              \
        const float wj = w * __uint_as_float(PHL_ROW_BCAST(__float_as_uint(f[j < NS - 1 ? j : 0]), K));      \
        acc[j + 1 < NS ? j + 1 : 0] = fma4(acc[j + 1 < NS ? j + 1 : 0], wj, q);                              \
    }
    PHL_WJ(0) PHL_WJ(1) PHL_WJ(2) PHL_WJ(3) PHL_WJ(4) PHL_WJ(5) PHL_WJ(6)
#undef PHL_WJ
}

// four consecutive entries K0..K0+3: the four row reads in flight together, then the NS x 4 accumulations
template <int K0, int NS>
__device__ __forceinline__ void sum4w(float4 (&acc)[NS], const uint2 e, const float (&f)[NS > 1 ? NS - 1 : 1], const char *rbase)
{
    const float w0 = __uint_as_float(PHL_ROW_BCAST(e.y, K0 + 0)), w1 = __uint_as_float(PHL_ROW_BCAST(e.y, K0 + 1));
    const float w2 = __uint_as_float(PHL_ROW_BCAST(e.y, K0 + 2)), w3 = __uint_as_float(PHL_ROW_BCAST(e.y, K0 + 3));
    const float4 q0 = *reinterpret_cast<const float4 *>(rbase + PHL_ROW_BCAST(e.x, K0 + 0));
    const float4 q1 = *reinterpret_cast<const float4 *>(rbase + PHL_ROW_BCAST(e.x, K0 + 1));
    const float4 q2 = *reinterpret_cast<const float4 *>(rbase + PHL_ROW_BCAST(e.x, K0 + 2));
    const float4 q3 = *reinterpret_cast<const float4 *>(rbase + PHL_ROW_BCAST(e.x, K0 + 3));
    wide_entry<K0 + 0, NS>(acc, q0, w0, f);
    wide_entry<K0 + 1, NS>(acc, q1, w1, f);
    wide_entry<K0 + 2, NS>(acc, q2, w2, f);
    wide_entry<K0 + 3, NS>(acc, q3, w3, f);
}

template <int NS>
__global__ __launch_bounds__(TPB_S) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_splat_wide(const float *__restrict__ src, int64_t src_rs, int vd, int n, int P,
                                                    int nv_cap, const int *__restrict__ pix_order, const int *__restrict__ vptr,
                                                    const int *__restrict__ slot_vert, const int *__restrict__ slot_pidx,
                                                    const int2 *__restrict__ seg_rng, const phl_contrib_t *__restrict__ seg,
                                                    float *__restrict__ vert, float *__restrict__ partial, int nchunks, int xcd_chunk,
                                                    const int *__restrict__ chunk_list, int nv_lo, int nv_hi, int long_seg,
                                                    const float *__restrict__ fref, int64_t fref_rs, int64_t fref_cs)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LPRS = 16, SL = 64, G = TPB_S / LPRS, Q = 4, NW = TPB_S / 64, PF = 8, NF = NS - 1, dp1 = NS;
    const int g = threadIdx.x / LPRS, l = threadIdx.x % LPRS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = lane / LPRS;
    const int ci = xcd_chunk > 0 ? (int)(blockIdx.x & 7) * xcd_chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (ci >= nchunks) return;
    const int c = chunk_list ? chunk_list[ci] : ci;
    const int base = c * P;
    const int cnt = min(P, n - base);
    const int E = cnt * dp1;
    const int64_t ebase = (int64_t)base * dp1;
    const int vbase = vptr[c], nv = vptr[c + 1] - vbase;
    if (nv <= nv_lo || nv > nv_hi) return;
    const int64_t out_rs = (int64_t)NS * vd;
    // LDS layout and prologue as in k_splat_tiled<16>
    float *rows = lds;
    uint2 *ent = reinterpret_cast<uint2 *>(lds + (size_t)(P + 1) * SL);
    int2 *meta = reinterpret_cast<int2 *>(ent + P * dp1);
    int *pixl = reinterpret_cast<int *>(meta + nv_cap);
    int *ctr = pixl + P;
    const int kclamp = cnt - 1;
    const char *rbase = reinterpret_cast<const char *>(rows) + l * 16;
    float4 pf[PF];
    int prow[PF];
#pragma unroll
    for (int u = 0; u < PF; u++) prow[u] = pix_order[base + min(g + u * G, kclamp)];
    for (int e = threadIdx.x; e < E; e += TPB_S) {
        const phl_contrib_t sg = seg[ebase + e];
        ent[e] = make_uint2((unsigned)(sg.pixel * SL * 4), __float_as_uint(sg.w));
    }
    for (int k = threadIdx.x; k < cnt; k += TPB_S) pixl[k] = pix_order[base + k];
    for (int i = threadIdx.x; i < nv; i += TPB_S) {
        const int2 rg = seg_rng[vbase + i];
        const int sv = slot_vert[vbase + i];
        meta[i] = make_int2((int)(rg.x - ebase) | ((int)(rg.y - ebase) << 16), sv < 0 ? sv : slot_pidx[vbase + i]);
        const bool lng = rg.y - rg.x >= long_seg;
        if (i + 1 < nv) {
            const int2 rn = seg_rng[vbase + i + 1];
            if (lng && rn.y - rn.x < long_seg) ctr[2] = i + 1;
        } else if (lng) {
            ctr[2] = nv;
        }
        if (i == 0 && !lng) ctr[2] = 0;
    }
    if (nv == 0 && threadIdx.x == 0) ctr[2] = 0;
    if (threadIdx.x < 2) ctr[threadIdx.x] = NW;
    if (threadIdx.x < LPRS) st4(rows + (size_t)P * SL + threadIdx.x * 4, make_float4(0.f, 0.f, 0.f, 0.f));
    {
        const bool chok = l * 4 < vd;
        const int chc = chok ? l * 4 : 0;
#pragma unroll
        for (int u = 0; u < PF; u++) pf[u] = ld4(src + (int64_t)prow[u] * src_rs + chc);
#pragma unroll
        for (int u = 0; u < PF; u++)
            if (chok && g + u * G < cnt) st4(rows + (g + u * G) * SL + l * 4, pf[u]);
    }
    const int r16 = lane & 15;
    const unsigned zoff = (unsigned)P * SL * 4;
    const int elast = E - 1;
    // an entry and the features of its pixel (weight-0 padding entries point at the row of zeros; any pixel's features do)
    auto fetch = [&](int pos, int s1, uint2 &e, float (&f)[NF > 0 ? NF : 1]) {
        e = ent[min(max(pos, 0), elast)];
        if (pos >= s1) e = make_uint2(zoff, 0u);
        const int64_t px = pixl[min((int)(e.x / (SL * 4)), kclamp)];
#pragma unroll
        for (int j = 0; j < NF; j++) f[j] = fref[px * fref_rs + (int64_t)j * fref_cs];
    };
    for (int c0 = 0; c0 < vd; c0 += SL) {
        const int ch = c0 + l * 4;
        const bool chok = ch < vd;
        __syncthreads();                   // rows of this slab are in LDS
        const int slab = c0 / SL;
        if (threadIdx.x == 0) ctr[(slab + 1) & 1] = NW;
        const int chn = ch + SL;
        const bool more = c0 + SL < vd;
        const bool chnok = more && chn < vd;
        const int chnc = chnok ? chn : 0;
        int *slab_ctr = ctr + (slab & 1);
        const int nlong = ctr[2];
        const int nitems = nlong + (nv - nlong + Q - 1) / Q;
        for (int gi = wave; gi < nitems;) {
            const bool coop = gi < nlong;
            const int i = coop ? gi : nlong + (gi - nlong) * Q + q;
            int2 m = make_int2(0, 0);
            if (i < nv) m = meta[i];
            int nxt = 0;
            if (lane == 0) nxt = atomicAdd(slab_ctr, 1);
            const int s1 = (int)((unsigned)m.x >> 16);
            int s = m.x & 0xFFFF;
            float4 acc[NS];
#pragma unroll
            for (int t = 0; t < NS; t++) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int len = s1 - s;
            int lmax = __builtin_amdgcn_readlane(len, 0);
            lmax = max(lmax, __builtin_amdgcn_readlane(len, 32));
            lmax = max(max(lmax, __builtin_amdgcn_readlane(len, 16)), __builtin_amdgcn_readlane(len, 48));
            const int step = coop ? 16 * Q : 16;
            if (coop) s += 16 * q;
            uint2 e;
            float f[NF > 0 ? NF : 1];
            fetch(s + r16, s1, e, f);
            for (int b = 0; b < lmax; b += step) {
                uint2 en;
                float fn[NF > 0 ? NF : 1];
                fetch(s + b + step + r16, s1, en, fn);       // next sixteen, under this batch's work
                sum4w<0, NS>(acc, e, f, rbase);
                sum4w<4, NS>(acc, e, f, rbase);
                if (b + 8 < lmax) {
                    sum4w<8, NS>(acc, e, f, rbase);
                    sum4w<12, NS>(acc, e, f, rbase);
                }
                e = en;
#pragma unroll
                for (int j = 0; j < NF; j++) f[j] = fn[j];
            }
            if (coop) {
#pragma unroll
                for (int t = 0; t < NS; t++) {
                    float4 a = acc[t];
                    a = make_float4(a.x + __shfl_xor(a.x, LPRS), a.y + __shfl_xor(a.y, LPRS), a.z + __shfl_xor(a.z, LPRS),
                                    a.w + __shfl_xor(a.w, LPRS));
                    a = make_float4(a.x + __shfl_xor(a.x, 2 * LPRS), a.y + __shfl_xor(a.y, 2 * LPRS), a.z + __shfl_xor(a.z, 2 * LPRS),
                                    a.w + __shfl_xor(a.w, 2 * LPRS));
                    acc[t] = a;
                }
            }
            if (i < nv && chok && !(coop && q != 0)) {
                float *dst = m.y < 0 ? vert + (int64_t)(m.y & 0x7FFFFFFF) * out_rs : partial + (int64_t)m.y * out_rs;
#pragma unroll
                for (int t = 0; t < NS; t++) st4(dst + (int64_t)t * vd + ch, acc[t]);
            }
            gi = __builtin_amdgcn_readfirstlane(nxt);
        }
        __syncthreads();                   // everyone is done reading this slab
        if (more) {
            // next slab's rows: loaded here, not prefetched into registers under the sums (32 VGPRs that decide between one
            // and two workgroups per CU; the other workgroup's sums cover this round trip)
#pragma unroll
            for (int u = 0; u < PF; u++) pf[u] = ld4(src + (int64_t)pixl[min(g + u * G, kclamp)] * src_rs + chnc);
#pragma unroll
            for (int u = 0; u < PF; u++)
                if (chnok && g + u * G < cnt) st4(rows + (g + u * G) * SL + l * 4, pf[u]);
        }
    }
}

// vertices with != 1 contributing chunk: sum their partial rows in ascending chunk order
// (0 chunks = ghost vertex of a neighbouring row band: zeros).  One lane group per vertex, independent waves.
// Vertices with more than `long_list` rows are left to k_splat_reduce_long.
constexpr int LONG_LIST = 24;
template <int LPR>
__global__ __launch_bounds__(256) void k_splat_reduce(const float *__restrict__ partial, const int *__restrict__ vs_ptr,
                                                      const phl_contrib_t *__restrict__ vs,
                                                      const int *__restrict__ slot_pidx, int M, int vd,
                                                      float *__restrict__ vert, const int *__restrict__ vlist, int long_list,
                                                      const int *__restrict__ pack_pos, float *__restrict__ pack, int64_t pack_rs)
{
    // vlist (optional): only these M vertex rows (phl_splat_part); pack_pos (optional, with vlist): listed row i is also
    // written to pack[pack_pos[i]] -- the row-band exchange's send buffer filled by the kernel that completes the rows
    constexpr int Gw = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, l = lane % LPR;
    const int wave = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * 4 * Gw;
    auto add4 = [](float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
    for (int64_t v0 = (int64_t)wave * Gw; v0 < M; v0 += stride) {
        if (v0 + sub >= M) continue;
        const int64_t v = vlist ? vlist[v0 + sub] : v0 + sub;
        const int beg = vs_ptr[v], end = vs_ptr[v + 1];
        const int pp = pack_pos ? pack_pos[v0 + sub] : -1;
        if (end - beg > long_list) continue;
        if (end - beg == 1) {
            // the chunk kernel wrote the row itself (launches before this one): only the copy into the send buffer is left
            if (pp >= 0)
                for (int ch = l * 4; ch < vd; ch += LPR * 4) st4(pack + (int64_t)pp * pack_rs + ch, ld4(vert + v * vd + ch));
            continue;
        }
        for (int ch = l * 4; ch < vd; ch += LPR * 4) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int e = beg;
            for (; e + 4 <= end; e += 4) {
                const float4 q0 = ld4(partial + (int64_t)__float_as_int(vs[e].w) * vd + ch);
                const float4 q1 = ld4(partial + (int64_t)__float_as_int(vs[e + 1].w) * vd + ch);
                const float4 q2 = ld4(partial + (int64_t)__float_as_int(vs[e + 2].w) * vd + ch);
                const float4 q3 = ld4(partial + (int64_t)__float_as_int(vs[e + 3].w) * vd + ch);
                acc = add4(add4(add4(add4(acc, q0), q1), q2), q3);
            }
            for (; e < end; e++) acc = add4(acc, ld4(partial + (int64_t)__float_as_int(vs[e].w) * vd + ch));
            st4(vert + v * vd + ch, acc);
            if (pp >= 0) st4(pack + (int64_t)pp * pack_rs + ch, acc);
        }
    }
}

// A vertex fed by MANY chunks (a flat image region spans hundreds of 16x16 tiles: natural images at M/n ~ 0.01
// have vertices with 100-600 partial rows, which one wave would add up one dependent load after the other) is
// summed by a whole workgroup: lane group j of the 256/LPR groups adds rows j, j+NG, ... (four loads in flight
// each) and the NG sums are added in ascending j -- a fixed order, so the result does not depend on scheduling.
// One workgroup per entry of `list` (the lattice's long vertices, or the caller's rows: then short ones exit).
template <int LPR>
__global__ __launch_bounds__(256) void k_splat_reduce_long(const float *__restrict__ partial, const int *__restrict__ vs_ptr,
                                                           const phl_contrib_t *__restrict__ vs,
                                                           const int *__restrict__ slot_pidx, int vd,
                                                           float *__restrict__ vert, const int *__restrict__ list, int long_list,
                                                           const int *__restrict__ pack_pos, float *__restrict__ pack, int64_t pack_rs)
{
    constexpr int NG = 256 / LPR;
    __shared__ float4 red[256];
    const int l = (int)threadIdx.x % LPR, jg = (int)threadIdx.x / LPR;
    const int64_t v = list[blockIdx.x];
    const int b0 = vs_ptr[v], b1 = vs_ptr[v + 1];
    if (b1 - b0 <= long_list) return;                       // workgroup-uniform
    auto add4 = [](float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
    for (int c0 = 0; c0 < vd; c0 += LPR * 4) {              // uniform trip count: there are barriers inside
        const bool chok = c0 + l * 4 < vd;
        const int ch = chok ? c0 + l * 4 : 0;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int e = b0 + jg;
        for (; e + 7 * NG < b1; e += 8 * NG) {               // eight row loads in flight per lane group
            float4 q[8];
#pragma unroll
            for (int u = 0; u < 8; u++) q[u] = ld4(partial + (int64_t)__float_as_int(vs[e + u * NG].w) * vd + ch);
#pragma unroll
            for (int u = 0; u < 8; u++) acc = add4(acc, q[u]);
        }
        for (; e < b1; e += NG) acc = add4(acc, ld4(partial + (int64_t)__float_as_int(vs[e].w) * vd + ch));
        red[threadIdx.x] = acc;
        __syncthreads();
        if (jg == 0 && chok) {
            float4 t = red[l];
            for (int k = 1; k < NG; k++) t = add4(t, red[k * LPR + l]);
            st4(vert + v * vd + ch, t);
            if (pack_pos && pack_pos[blockIdx.x] >= 0) st4(pack + (int64_t)pack_pos[blockIdx.x] * pack_rs + ch, t);
        }
        __syncthreads();
    }
}

// One workgroup per chunk; the SLAB WIDTH IS CHOSEN BY THE WORKGROUP from its own chunk: 2^LSH lanes (of 4 floats)
// per row, the widest (up to `lsh_max`) whose nv staged vertex rows fit next to the index data in the `lds_bytes`
// the launch was given.  A typical chunk of a natural image has 20-60 local vertices and takes all 256 channels in
// one slab, a textured one with 300 takes 32-channel slabs, and neither decides for the other (round 2 sized every
// chunk for the worst one).  A chunk whose rows do not fit even the narrowest slab gathers them from global memory
// (DIRECT: pixels that share next to nothing -- there is nothing to stage).  The width is a template parameter of
// the body (a workgroup-uniform switch in the kernel): with a run-time width the same loops ran 40 % slower.
struct slice_args {
    const float *vert;
    int vd, dp1, cnt, nv;
    float *out;
    int64_t out_rs;
    const float *sub_src;
    int64_t sub_rs;
    float cdiv, rcdiv;
};

template <int LSH, bool EXACT, bool DIRECT>
__device__ __forceinline__ void slice_chunk(const slice_args &a, const uint2 *__restrict__ ent, const int *__restrict__ pixl,
                                            const int *__restrict__ vl, float *__restrict__ rows)
{
    constexpr int LPRS = 1 << LSH;
    constexpr int SL = LPRS * 4;
    constexpr int G = TPB / LPRS;
    const int g = threadIdx.x / LPRS, l = threadIdx.x % LPRS;
    const int vd = a.vd, dp1 = a.dp1, cnt = a.cnt, nv = a.nv;
    for (int c0 = 0; c0 < vd; c0 += SL) {
        const int ch = c0 + l * 4;
        const bool chok = ch < vd;
        const int chc = chok ? ch : 0;
        if (!DIRECT) {
            const int iclamp = nv - 1;
            for (int i0 = g; i0 < nv; i0 += 8 * G) {
                float4 q[8];
#pragma unroll
                for (int u = 0; u < 8; u++) q[u] = ld4(a.vert + (int64_t)vl[min(i0 + u * G, iclamp)] * vd + chc);
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (chok && i0 + u * G < nv) st4(rows + (i0 + u * G) * SL + l * 4, q[u]);
            }
            __syncthreads();
        }
        const char *rbase = reinterpret_cast<const char *>(rows) + l * 16;
        const float *gbase = a.vert + chc;
        auto row = [&](unsigned x) -> float4 {
            if (DIRECT) return ld4(gbase + (int64_t)x * vd);
            return *reinterpret_cast<const float4 *>(rbase + x);
        };
        for (int k = g; k < cnt; k += G) {
            const uint2 *ek = ent + k * dp1;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int r = 0;
            for (; r + 3 <= dp1; r += 3) {
                const uint2 e0 = ek[r], e1 = ek[r + 1], e2 = ek[r + 2];
                const float4 q0 = row(e0.x), q1 = row(e1.x), q2 = row(e2.x);
                if (EXACT) {
                    acc = term4(acc, __uint_as_float(e0.y), q0, a.cdiv, a.rcdiv);
                    acc = term4(acc, __uint_as_float(e1.y), q1, a.cdiv, a.rcdiv);
                    acc = term4(acc, __uint_as_float(e2.y), q2, a.cdiv, a.rcdiv);
                } else {
                    acc = fma4(acc, __uint_as_float(e0.y), q0);
                    acc = fma4(acc, __uint_as_float(e1.y), q1);
                    acc = fma4(acc, __uint_as_float(e2.y), q2);
                }
            }
            for (; r < dp1; r++) {
                const uint2 e0 = ek[r];
                const float4 q0 = row(e0.x);
                if (EXACT) acc = term4(acc, __uint_as_float(e0.y), q0, a.cdiv, a.rcdiv);
                else acc = fma4(acc, __uint_as_float(e0.y), q0);
            }
            if (chok) {
                const int p = pixl[k];
                if (!EXACT) acc = make_float4(acc.x * a.rcdiv, acc.y * a.rcdiv, acc.z * a.rcdiv, acc.w * a.rcdiv);
                if (a.sub_src) {
                    const float4 s = ld4(a.sub_src + (int64_t)p * a.sub_rs + ch);
                    acc = make_float4(acc.x - s.x, acc.y - s.y, acc.z - s.z, acc.w - s.w);
                }
                st4(a.out + (int64_t)p * a.out_rs + ch, acc);
            }
        }
        if (!DIRECT) __syncthreads();
    }
}

template <bool EXACT>
__global__ __launch_bounds__(TPB) void k_slice_tiled(const float *__restrict__ vert, int vd, int n, int P, int dp1,
                                                     int lsh_max, int lds_bytes, const int *__restrict__ pix_order,
                                                     const int *__restrict__ vptr, const int *__restrict__ slot_vert,
                                                     const unsigned short *__restrict__ lidx,
                                                     const phl_replay_t *__restrict__ replay, float *__restrict__ out,
                                                     int64_t out_rs, const float *__restrict__ sub_src, int64_t sub_rs,
                                                     float cdiv, float rcdiv, int nchunks, int xcd_chunk,
                                                     const int *__restrict__ heavy_list, int heavy_n, int heavy_pad, int heavy_thr)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // Heavy chunks first: the first heavy_pad blocks take the heavy_n chunks with more than heavy_thr local vertices
    // (narrow slabs, several times the work of a typical chunk) so that they run under the bulk instead of forming
    // the launch's tail; the other blocks walk all chunks in the XCD-aware order (see k_blur: neighbouring chunks
    // share boundary vertices) and skip those.  heavy_pad is a multiple of 8: the XCD of a chunk does not move.
    int c;
    const bool heavy_block = (int)blockIdx.x < heavy_pad;
    if (heavy_block) {
        if ((int)blockIdx.x >= heavy_n) return;
        c = heavy_list[blockIdx.x];
    } else {
        const int b = (int)blockIdx.x - heavy_pad;
        c = xcd_chunk > 0 ? (b & 7) * xcd_chunk + (b >> 3) : b;
        if (c >= nchunks) return;
    }
    const int base = c * P;
    const int cnt = min(P, n - base);
    const int E = cnt * dp1;
    const int64_t ebase = (int64_t)base * dp1;
    const int vbase = vptr[c], nv = vptr[c + 1] - vbase;
    if (!heavy_block && nv > heavy_thr) return;
    // index data first (its size does not depend on the slab width), rows behind it
    uint2 *ent = reinterpret_cast<uint2 *>(lds);                    // [P*dp1] {row offset or vertex id, weight}
    int *pixl = reinterpret_cast<int *>(ent + P * dp1);             // [P]
    int *vl = pixl + P;                                             // [nv]
    const int fixed = (P * dp1 * 8 + P * 4 + nv * 4 + 15) & ~15;
    float *rows = reinterpret_cast<float *>(reinterpret_cast<char *>(lds) + fixed);
    int lsh = lsh_max;                                              // workgroup-uniform
    while (lsh > 2 && fixed + (int64_t)nv * (16 << lsh) > lds_bytes) lsh--;
    const bool direct = fixed + (int64_t)nv * (16 << lsh) > lds_bytes;
    if (direct) lsh = lsh_max;
    for (int k = threadIdx.x; k < cnt; k += TPB) pixl[k] = pix_order[base + k];
    if (!direct)
        for (int i = threadIdx.x; i < nv; i += TPB) vl[i] = slot_vert[vbase + i] & 0x7FFFFFFF;
    for (int e = threadIdx.x; e < E; e += TPB) {
        const int k = e / dp1, r = e - k * dp1;
        const int p = pix_order[base + k];
        const unsigned li = lidx[ebase + e];
        // staged: byte offset of the local vertex's LDS row; direct: the vertex id itself
        ent[e] = make_uint2(direct ? (unsigned)(slot_vert[vbase + li] & 0x7FFFFFFF) : li << (lsh + 4),
                            __float_as_uint(replay[(int64_t)p * dp1 + r].w));
    }
    __syncthreads();
    slice_args a;
    a.vert = vert; a.vd = vd; a.dp1 = dp1; a.cnt = cnt; a.nv = nv;
    a.out = out; a.out_rs = out_rs; a.sub_src = sub_src; a.sub_rs = sub_rs; a.cdiv = cdiv; a.rcdiv = rcdiv;
    if (direct) {
        switch (lsh) {
            case 6: slice_chunk<6, EXACT, true>(a, ent, pixl, vl, rows); break;
            case 5: slice_chunk<5, EXACT, true>(a, ent, pixl, vl, rows); break;
            case 4: slice_chunk<4, EXACT, true>(a, ent, pixl, vl, rows); break;
            case 3: slice_chunk<3, EXACT, true>(a, ent, pixl, vl, rows); break;
            default: slice_chunk<2, EXACT, true>(a, ent, pixl, vl, rows); break;
        }
    } else {
        switch (lsh) {
            case 6: slice_chunk<6, EXACT, false>(a, ent, pixl, vl, rows); break;
            case 5: slice_chunk<5, EXACT, false>(a, ent, pixl, vl, rows); break;
            case 4: slice_chunk<4, EXACT, false>(a, ent, pixl, vl, rows); break;
            case 3: slice_chunk<3, EXACT, false>(a, ent, pixl, vl, rows); break;
            default: slice_chunk<2, EXACT, false>(a, ent, pixl, vl, rows); break;
        }
    }
}

// ---- gradient w.r.t. the features: slice of the WIDE vertex buffer, contracted on the fly ---------------------------
// crf/gaussian_matrix.py:450-463 filters [g, g(x)ref, src, src(x)ref] (2L(1+d) channels) and contracts the result to
// [n, d].  Here one pass (x, y) of phl_filter_grad holds the blurred wide vertex rows [M][NS*L] (block 0 = splat of x,
// block 1+k = splat of x (x) ref[:, k]; NS = d+1) and this kernel evaluates, per pixel i,
//     T_ik = -2 / (1 + 2^-d) * ( f_ik * sum_l y_il (Wx)_il  -  sum_l y_il (W(x f_k))_il )
// without writing any sliced row.  Per slab of 4*LG channels the chunk's vertex rows are staged for ALL NS blocks at
// once ([nv][NS][4*LG] floats), so a chunk passes L / (4 LG) barriers-bounded phases, not NS times as many; an
// LG-lane group owns 256 / (512 / LG) pixels and keeps their NS running dot products in registers.  y is read once,
// the vertex array about twice (chunks overlap), [n, d] is written; (Wx) itself is written on request (it is the
// gradient w.r.t. the source when x = g).  The workgroup picks LG = 8 (32-channel slabs) if its rows fit the LDS it
// was given, else 4; DIRECT: chunks that fit neither gather their rows from global memory.
template <int NS, int LG, bool DIRECT>
__device__ __forceinline__ void slice_grad_chunk(const float *__restrict__ vertw, int L, int cnt, int nv,
                                                 const uint2 *ent, const int *pixl, const int *vl, float *rows,   // (one LDS block:
                                                 // NOT restrict -- with it the compiler hoists every pixel's entries and
                                                 // all NS*NS row addresses out of the slab loop and spills 800 bytes per lane)
                                                 const float *__restrict__ y, int64_t y_rs, const float *__restrict__ ref,
                                                 int64_t ref_rs, int64_t ref_cs, float *__restrict__ grad_ref, int accumulate,
                                                 float *__restrict__ wx_out, int64_t wx_rs, float rcdiv)
{
    constexpr int SLG = LG * 4;              // channels per slab
    constexpr int G = TPB / LG;              // lane groups
    constexpr int PPG = 256 / G;             // pixels per group (P <= 256)
    constexpr int PITCH = NS * SLG;          // floats per staged local vertex
    const int g = threadIdx.x / LG, l = threadIdx.x % LG;
    const int64_t vdw = (int64_t)NS * L;
    float acc[PPG][NS];
#pragma unroll
    for (int u = 0; u < PPG; u++)
#pragma unroll
        for (int t = 0; t < NS; t++) acc[u][t] = 0.f;
    const int kclamp = cnt - 1;
    for (int c0 = 0; c0 < L; c0 += SLG) {
        const int ch = c0 + l * 4;
        const bool chok = ch < L;
        const int chc = chok ? ch : 0;
        float4 yv[PPG];
#pragma unroll
        for (int u = 0; u < PPG; u++) yv[u] = ld4(y + (int64_t)pixl[min(g + u * G, kclamp)] * y_rs + chc);
        if (!DIRECT) {
            // [nv][NS][LG] 16-byte pieces, LG consecutive threads on LG consecutive pieces of one (vertex, block)
            const int total = nv * NS * LG;
            for (int i0 = threadIdx.x; i0 < total; i0 += 4 * TPB) {
                float4 q[4];
                int dsto[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int idx = min(i0 + u * TPB, total - 1);
                    const int i = idx / (NS * LG), rem = idx - i * (NS * LG);
                    const int set = rem / LG, ll = rem - set * LG;
                    const int cc = c0 + ll * 4 < L ? c0 + ll * 4 : 0;
                    q[u] = ld4(vertw + (int64_t)vl[i] * vdw + (int64_t)set * L + cc);
                    dsto[u] = i * PITCH + set * SLG + ll * 4;
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (i0 + u * TPB < total) st4(rows + dsto[u], q[u]);
            }
            __syncthreads();
        }
        const char *rbase = reinterpret_cast<const char *>(rows) + l * 16;
#pragma unroll
        for (int u = 0; u < PPG; u++) {
            const int k = min(g + u * G, kclamp);
            const bool live = chok && g + u * G < cnt;
            uint2 e[NS];                       // the pixel's d+1 = NS simplex vertices {row offset | vertex id, weight}
            const char *pr[NS];
#pragma unroll
            for (int r = 0; r < NS; r++) {
                e[r] = ent[k * NS + r];
                pr[r] = rbase + e[r].x;         // block `set` of the row: a constant offset from here
            }
#pragma unroll
            for (int set = 0; set < NS; set++) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int r = 0; r < NS; r++) {
                    const float4 q0 = DIRECT ? ld4(vertw + (int64_t)e[r].x * vdw + (int64_t)set * L + chc)
                                             : *reinterpret_cast<const float4 *>(pr[r] + set * (SLG * 4));
                    v = fma4(v, __uint_as_float(e[r].y), q0);
                }
                const float dot = yv[u].x * v.x + yv[u].y * v.y + yv[u].z * v.z + yv[u].w * v.w;
                acc[u][set] += live ? dot : 0.f;
                if (set == 0 && wx_out && live)
                    st4(wx_out + (int64_t)pixl[k] * wx_rs + ch, make_float4(v.x * rcdiv, v.y * rcdiv, v.z * rcdiv, v.w * rcdiv));
                // one (pixel, block) at a time: left alone, the compiler issues all NS*NS row reads of a pixel first and
                // spills them (800 bytes of scratch per lane); sched_barrier did not stop that, this does
                asm volatile("" : "+v"(acc[u][set])::"memory");
            }
        }
        if (!DIRECT) __syncthreads();
    }
    // the LG lanes of a group hold a slab's channels between them: add their dot products
#pragma unroll
    for (int u = 0; u < PPG; u++)
#pragma unroll
        for (int t = 0; t < NS; t++) {
            float a = acc[u][t];
#pragma unroll
            for (int o = LG / 2; o > 0; o >>= 1) a += __shfl_xor(a, o, LG);
            acc[u][t] = a;
        }
    if (l == 0) {
#pragma unroll
        for (int u = 0; u < PPG; u++) {
            const int k = g + u * G;
            if (k >= cnt) continue;
            const int64_t p = pixl[k];
#pragma unroll
            for (int t = 1; t < NS; t++) {
                const float f = ref[p * ref_rs + (int64_t)(t - 1) * ref_cs];
                const float val = -2.f * rcdiv * (f * acc[u][0] - acc[u][t]);
                float *dst = grad_ref + p * (NS - 1) + (t - 1);
                *dst = accumulate ? *dst + val : val;
            }
        }
    }
}

template <int NS>
__global__ __launch_bounds__(TPB) void k_slice_grad(const float *__restrict__ vertw, int L, int n, int P, int lds_bytes,
                                                    const int *__restrict__ pix_order, const int *__restrict__ vptr,
                                                    const int *__restrict__ slot_vert, const unsigned short *__restrict__ lidx,
                                                    const phl_replay_t *__restrict__ replay, const float *__restrict__ y,
                                                    int64_t y_rs, const float *__restrict__ ref, int64_t ref_rs, int64_t ref_cs,
                                                    float *__restrict__ grad_ref, int accumulate, float *__restrict__ wx_out,
                                                    int64_t wx_rs, float rcdiv, int nchunks, int xcd_chunk)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int dp1 = NS;
    const int c = xcd_chunk > 0 ? (int)(blockIdx.x & 7) * xcd_chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (c >= nchunks) return;
    const int base = c * P;
    const int cnt = min(P, n - base);
    const int E = cnt * dp1;
    const int64_t ebase = (int64_t)base * dp1;
    const int vbase = vptr[c], nv = vptr[c + 1] - vbase;
    uint2 *ent = reinterpret_cast<uint2 *>(lds);
    int *pixl = reinterpret_cast<int *>(ent + P * dp1);
    int *vl = pixl + P;
    const int fixed = (P * dp1 * 8 + P * 4 + nv * 4 + 15) & ~15;
    float *rows = reinterpret_cast<float *>(reinterpret_cast<char *>(lds) + fixed);
    // workgroup-uniform choice: 32-channel slabs, 16-channel slabs, or no staging
    const int mode = fixed + (int64_t)nv * (NS * 128) <= lds_bytes ? 8 : (fixed + (int64_t)nv * (NS * 64) <= lds_bytes ? 4 : 0);
    const unsigned pitch_bytes = (unsigned)NS * (mode == 8 ? 128u : 64u);
    for (int k = threadIdx.x; k < cnt; k += TPB) pixl[k] = pix_order[base + k];
    if (mode)
        for (int i = threadIdx.x; i < nv; i += TPB) vl[i] = slot_vert[vbase + i] & 0x7FFFFFFF;
    for (int e = threadIdx.x; e < E; e += TPB) {
        const int k = e / dp1, r = e - k * dp1;
        const int p = pix_order[base + k];
        const unsigned li = lidx[ebase + e];
        ent[e] = make_uint2(mode ? li * pitch_bytes : (unsigned)(slot_vert[vbase + li] & 0x7FFFFFFF),
                            __float_as_uint(replay[(int64_t)p * dp1 + r].w));
    }
    __syncthreads();
    if (mode == 8)
        slice_grad_chunk<NS, 8, false>(vertw, L, cnt, nv, ent, pixl, vl, rows, y, y_rs, ref, ref_rs, ref_cs, grad_ref, accumulate,
                                       wx_out, wx_rs, rcdiv);
    else if (mode == 4)
        slice_grad_chunk<NS, 4, false>(vertw, L, cnt, nv, ent, pixl, vl, rows, y, y_rs, ref, ref_rs, ref_cs, grad_ref, accumulate,
                                       wx_out, wx_rs, rcdiv);
    else
        slice_grad_chunk<NS, 8, true>(vertw, L, cnt, nv, ent, pixl, vl, rows, y, y_rs, ref, ref_rs, ref_cs, grad_ref, accumulate,
                                      wx_out, wx_rs, rcdiv);
}

// vs[e].w <- partial-buffer row of the slot (bit pattern of an int): saves the reduce kernels one dependent load
__global__ __launch_bounds__(256) void k_fill_vs_rows(phl_contrib_t *__restrict__ vs, int S, const int *__restrict__ slot_pidx)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < S) vs[e].w = __int_as_float(slot_pidx[vs[e].pixel]);
}

// sort key of a long vertex: lists in descending length
__global__ __launch_bounds__(256) void k_long_keys(const int *__restrict__ vlong, int n, const int *__restrict__ vs_ptr, int kmax,
                                                   int *__restrict__ key)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key[i] = kmax - min(vs_ptr[vlong[i] + 1] - vs_ptr[vlong[i]], kmax);
}

__global__ __launch_bounds__(256) void k_gather_i32(const int *__restrict__ src, const int *__restrict__ perm, int n, int *__restrict__ dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}

// vertex processing order for the gather splat: vertices sorted by the first chunk that touches
// them, so that vertices summed at the same time read the same few chunks' pixel rows (L2 hits
// instead of Infinity-Cache traffic).  first[s] = 1 iff slot s is the first slot of its vertex.
__global__ __launch_bounds__(256) void k_first_slot(const int *__restrict__ slot_vert, int S, const int *__restrict__ vs_ptr,
                                                    const phl_contrib_t *__restrict__ vs, int *__restrict__ first)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const int v = slot_vert[s] & 0x7FFFFFFF;
    first[s] = (vs[vs_ptr[v]].pixel == s) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_fill_vorder(const int *__restrict__ slot_vert, int S, const int *__restrict__ first,
                                                     const int *__restrict__ rank, int M_local, int M,
                                                     int *__restrict__ vorder)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < S && first[i]) vorder[rank[i]] = slot_vert[i] & 0x7FFFFFFF;
    if (i >= M_local && i < M) vorder[i] = i;      // ghost vertices (no local contributions) go last
}

__global__ __launch_bounds__(256) void k_strip_marks(int *slot_vert, int S)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S) slot_vert[s] &= 0x7FFFFFFF;
}

template <typename F>
inline void dispatch_lprs(int lprs, F &&f)
{
    switch (lprs) {
        case 64: f(std::integral_constant<int, 64>{}); break;
        case 32: f(std::integral_constant<int, 32>{}); break;
        case 16: f(std::integral_constant<int, 16>{}); break;
        case 8: f(std::integral_constant<int, 8>{}); break;
        default: f(std::integral_constant<int, 4>{}); break;
    }
}

// LDS per workgroup: default 80 KiB -> two workgroups per CU (160 KiB LDS per CU on gfx950)
int lds_budget()
{
    static int b = [] {
        const char *e = getenv("PHL_TILE_LDS");
        int v = e ? atoi(e) : 80 * 1024;
        return v < 8192 ? 8192 : (v > 160 * 1024 ? 160 * 1024 : v);
    }();
    return b;
}

// bytes of index data staged next to the rows (entries, pixel ids, local pointers)
inline int64_t lds_extra(int P, int dp1, int nv_max) { return (int64_t)P * dp1 * 8 + (int64_t)P * 4 + ((int64_t)nv_max + 2) * 8 + 16 + 256; }
// k_slice_tiled keeps less: entries, pixel ids, one vertex id per local vertex

// Launch configuration of a chunk kernel: lanes (of 4 floats) per slab row and LDS bytes per workgroup.
// Candidates in order of preference: every slab width, widest first, at two workgroups per CU (80 KiB); only then
// the same widths at one workgroup per CU.  Measured on C3: the splat with 128-channel slabs and one workgroup per
// CU takes 0.96 ms against 0.78 with 64-channel slabs and two; at M/n = 0.5 64 channels / one workgroup 2.0 ms
// against 1.47 with 32 channels / two -- a second workgroup hides the staging round trips of the first.
struct tile_cfg {
    int lprs = -1;
    size_t lds = 0;
    int cap = 0;        // most local vertices a chunk may have under this configuration
};
inline int64_t lds_big() { return lds_budget() > 160 * 1024 ? lds_budget() : 160 * 1024; }

// k-th candidate for `vd` channels (k = 0 is the preferred one); returns false past the last
inline bool cfg_candidate(int vd, int k, int *lprs, int64_t *budget)
{
    const int need = (vd + 3) / 4;
    int w = 4;
    while (w < 64 && w < need) w <<= 1;
    static const int max_w = getenv("PHL_MAX_LPRS_SPLAT") ? atoi(getenv("PHL_MAX_LPRS_SPLAT")) : 64;   // experiments
    while (w > 4 && w > max_w) w >>= 1;
    int nw = 0;
    for (int x = w; x >= 4; x >>= 1) nw++;
    if (k >= 2 * nw) return false;
    *lprs = w >> (k % nw);
    *budget = k / nw ? lds_big() : lds_budget();
    return true;
}

inline int64_t splat_lds(int P, int dp1, int lprs, int nv) { return (int64_t)(P + 1) * lprs * 16 + lds_extra(P, dp1, nv); }

// first candidate that holds a chunk of `nv` local vertices, sized for exactly that many
inline tile_cfg pick_cfg(const phl_lattice *lat, int vd, int nv)
{
    tile_cfg c;
    int lprs;
    int64_t budget;
    const int P = lat->P, dp1 = lat->d + 1;
    for (int k = 0; cfg_candidate(vd, k, &lprs, &budget); k++) {
        const int64_t need = splat_lds(P, dp1, lprs, nv);
        if (need > budget) continue;
        c.lprs = lprs;
        c.cap = nv;
        c.lds = (size_t)need;
        return c;
    }
    return c;
}

// Chunk classes.  A launch's LDS layout is sized by the most local vertices a chunk of it may have.  Sizing every
// chunk for the worst one (round 2) lets a single textured 16x16 tile narrow the slabs of the whole image: natural
// images at M/n ~ 0.01 have 30 vertices in a typical chunk and 300 in the worst (0.3 % of the chunks above 158).
// So the chunks are cut into classes by their vertex count, one launch per class: walking the candidate
// configurations in order of preference, each one takes the chunks it can hold that no better one took (a class of
// fewer than MIN_CLASS chunks is left to the next candidate).  The largest class runs over the whole grid in
// chunk order (chunks of other classes exit at once); the others are contiguous ranges of `chunk_by_nv` (chunk
// ids by descending vertex count, built with the lattice).
struct tile_class {
    tile_cfg cfg;
    int lo, hi;         // chunks with lo < nv <= hi
    int begin, count;   // range of chunk_by_nv
    bool full_grid;
};
struct tile_plan {
    tile_class cls[12];
    int n = 0;
};
constexpr int MIN_CLASS = 64;
// chunks with more than nv local vertices
inline int chunks_above(const phl_lattice *lat, int nv) { return nv >= lat->nv_max ? 0 : lat->nchunks - lat->nv_cum[nv < 0 ? 0 : nv]; }

inline tile_plan plan_tiles(const phl_lattice *lat, int vd)
{
    tile_plan p;
    const tile_cfg all = pick_cfg(lat, vd, lat->nv_max);
    if (all.lprs < 0) return p;
    static const bool classes = !(getenv("PHL_CLASSES") && atoi(getenv("PHL_CLASSES")) == 0);
    const int P = lat->P, dp1 = lat->d + 1;
    int covered = -1;                                         // chunks with nv <= covered are taken
    if (classes && lat->nv_cum && lat->chunk_by_nv) {
        int lprs;
        int64_t budget;
        for (int k = 0; covered < lat->nv_max && cfg_candidate(vd, k, &lprs, &budget); k++) {
            const int64_t base = splat_lds(P, dp1, lprs, 0);
            if (base > budget) continue;
            const int64_t per_v = 8;
            int cap = (int)((budget - base) / per_v);
            if (cap > lat->nv_max) cap = lat->nv_max;
            if (cap <= covered) continue;
            const int count = chunks_above(lat, covered) - chunks_above(lat, cap);
            if (cap < lat->nv_max && count < MIN_CLASS) continue;
            if (count == 0) { covered = cap; continue; }
            // a class of at most one chunk per CU is all tail: occupancy buys nothing, half the slabs halve its time
            if (count <= 256) {
                int wmax = 4;
                while (wmax < 64 && wmax < (vd + 3) / 4) wmax <<= 1;
                for (int x = wmax; x > lprs; x >>= 1)
                    if (splat_lds(P, dp1, x, cap) <= lds_big()) { lprs = x; break; }
            }
            tile_class &c = p.cls[p.n++];
            c.cfg.lprs = lprs;
            c.cfg.cap = cap;
            c.cfg.lds = (size_t)splat_lds(P, dp1, lprs, cap);
            c.lo = covered;
            c.hi = cap;
            c.begin = chunks_above(lat, cap);
            c.count = count;
            c.full_grid = false;
            covered = cap;
            if (p.n == 11) break;
        }
    }
    if (covered < lat->nv_max) {                              // no classes, or candidates exhausted: `all` takes the rest
        tile_class &c = p.cls[p.n++];
        c.cfg = all;
        c.lo = covered;
        c.hi = 0x7FFFFFFF;
        c.begin = 0;
        c.count = chunks_above(lat, covered);
        c.full_grid = false;
    }
    int big = 0;
    for (int i = 1; i < p.n; i++)
        if (p.cls[i].count > p.cls[big].count) big = i;
    p.cls[big].full_grid = true;
    if (p.n == 1) { p.cls[0].lo = -1; p.cls[0].hi = 0x7FFFFFFF; }
    static const bool dbg = getenv("PHL_DEBUG") != nullptr;
    if (dbg)
        for (int i = 0; i < p.n; i++)
            fprintf(stderr, "[phl] splat vd=%d class %d/%d: %d < nv <= %d, %d chunks, %d lanes, %zu B LDS%s\n",
                    vd, i, p.n, p.cls[i].lo, p.cls[i].hi, p.cls[i].count, p.cls[i].cfg.lprs, p.cls[i].cfg.lds,
                    p.cls[i].full_grid ? " (whole grid)" : "");
    return p;
}

// The slice picks its slab width per workgroup (k_slice_tiled); the launch only fixes the widest slab worth
// having (as wide as vd) and the LDS per workgroup: what the worst chunk needs at that width, at most the budget
// of two workgroups per CU (the index data alone, P*(8(d+1)+4) bytes, is always below it).
struct slice_cfg {
    int lsh_max;
    size_t lds;
};
inline slice_cfg plan_slice(const phl_lattice *lat, int vd)
{
    const int need = (vd + 3) / 4;
    int lsh = 2;
    while (lsh < 6 && (1 << lsh) < need) lsh++;
    static const int max_w = getenv("PHL_MAX_LPRS_SLICE") ? atoi(getenv("PHL_MAX_LPRS_SLICE")) : 64;   // experiments
    while (lsh > 2 && (1 << lsh) > max_w) lsh--;
    const int P = lat->P, dp1 = lat->d + 1;
    const int64_t fixed = (((int64_t)P * dp1 * 8 + (int64_t)P * 4 + (int64_t)lat->nv_max * 4 + 15) & ~(int64_t)15);
    int64_t lds = fixed + (int64_t)lat->nv_max * (16 << lsh);
    if (lds > lds_budget()) lds = lds_budget();     // chunks that do not fit take narrower slabs, or go direct
    slice_cfg c;
    c.lsh_max = lsh;
    c.lds = (size_t)lds;
    return c;
}

template <typename K>
inline int allow_lds(K kernel, size_t bytes, int threads = 512)
{
    if (bytes > 64 * 1024) {
        // once per (device, kernel, size): the attribute call is not free and this runs on every launch
        static std::mutex mu;
        static std::map<std::pair<int, const void *>, size_t> allowed;
        int dev = 0;
        PHL_HIP(hipGetDevice(&dev));
        const void *fn = reinterpret_cast<const void *>(kernel);
        std::lock_guard<std::mutex> lk(mu);
        size_t &have = allowed[std::make_pair(dev, fn)];
        if (bytes > have) {
            PHL_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            have = bytes;
        }
    }
    static const bool dbg = getenv("PHL_DEBUG") != nullptr;
    if (dbg) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kernel), threads, bytes);
        fprintf(stderr, "[phl] chunk kernel: %d threads, %zu B LDS -> %d workgroups per CU (occupancy API)\n", threads, bytes, nb);
    }
    return PHL_OK;
}

// grid size and per-XCD chunk count for the XCD-aware chunk order
inline void chunk_grid(int nchunks, unsigned *grid, int *xcd_chunk)
{
    static const bool xcd = !(getenv("PHL_XCD") && atoi(getenv("PHL_XCD")) == 0);
    if (xcd && nchunks >= 64) {
        *grid = (unsigned)((nchunks + 7) / 8 * 8);
        *xcd_chunk = (int)(*grid / 8);
    } else {
        *grid = (unsigned)nchunks;
        *xcd_chunk = 0;
    }
}

inline int pick_lpr_row(int vd)
{
    const int need = (vd + 3) / 4;
    return need >= 64 ? 64 : need >= 16 ? 16 : need >= 4 ? 4 : 1;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
int phl_tiles_free(phl_lattice *lat)
{
    void *ptrs[] = {lat->pix_order, lat->chunk_vptr, lat->slot_vert, lat->slot_pidx, lat->seg_rng, lat->seg,
                    lat->lidx, lat->vs_ptr, lat->vs, lat->vorder, lat->chunk_by_nv, lat->vlong};
    for (void *p : ptrs)
        if (p) (void)phl_dev_free(p);
    free(lat->nv_cum);
    lat->nv_cum = nullptr;
    lat->chunk_by_nv = nullptr;
    lat->vlong = nullptr;
    lat->n_long = 0;
    lat->pix_order = lat->chunk_vptr = lat->slot_vert = lat->slot_pidx = lat->vs_ptr = nullptr;
    lat->seg_rng = nullptr;
    lat->vorder = nullptr;
    lat->seg = lat->vs = nullptr;
    lat->lidx = nullptr;
    lat->nchunks = 0;
    lat->S = lat->S_multi = 0;
    lat->nv_max = 0;
    return PHL_OK;
}

// vertex -> slots lists (ascending slot = ascending chunk), sole marks and partial-row indices.
// Also called after ghost vertices were appended (M grew, the slots did not change).
// Vertex processing order of the gather splat (exact arithmetic, shapes the chunk kernels do not take): made on first
// use, under the same lock as the contribution lists (phl_ensure_csr).
int phl_tiles_ensure_vorder(phl_lattice *lat, hipStream_t st)
{
    if (lat->vorder || !lat->vs_ptr || !lat->vs || !lat->slot_vert) return PHL_OK;
    const int M = (int)lat->M, S = (int)lat->S;
    if (M == 0 || S == 0) return PHL_OK;
    temp_pool tmp;
    int *first, *frank, *tile_sums;
    PHL_HIP(tmp.get(&first, (size_t)S + 1));
    PHL_HIP(tmp.get(&frank, (size_t)S + 2));
    PHL_HIP(tmp.get(&tile_sums, (size_t)S / SCAN_TILE + 2));
    int *vorder = nullptr;
    PHL_HIP(phl_dev_malloc((void **)&vorder, sizeof(int) * ((size_t)M + 1)));
    const unsigned gS = (unsigned)((S + 255) / 256);
    hipLaunchKernelGGL(k_first_slot, dim3(gS), dim3(256), 0, st, lat->slot_vert, S, lat->vs_ptr, lat->vs, first);
    PHL_HIP(hipGetLastError());
    const int rc = exclusive_scan(first, frank, S, tile_sums, st);
    if (rc) return rc;
    const int span = S > M ? S : M;
    hipLaunchKernelGGL(k_fill_vorder, dim3((span + 255) / 256), dim3(256), 0, st, lat->slot_vert, S, first, frank,
                       (int)lat->M_local, M, vorder);
    PHL_HIP(hipGetLastError());
    PHL_HIP(hipStreamSynchronize(st));      // temporaries go back to the scratch cache
    lat->vorder = vorder;
    return PHL_OK;
}

// ---- side streams for launches that are independent of their neighbours in a call ---------------------------------------
// A chunk class of a few heavy chunks (textured 16x16 tiles of a natural image: 35 us for 0.6 MB at C3) used to run as its
// own launch BEHIND the main grid, the whole chip waiting for a handful of workgroups.  It now runs BESIDE the main grid, on
// a high-priority side stream forked from the caller's (event fork / join: legal inside a stream capture too).  Slots are
// pooled per device; a slot is taken for the duration of the host call only -- later users of the same stream are ordered
// behind the earlier work, and an event wait refers to the record that preceded it, so re-recording an event is safe.
namespace {
struct fork_slot {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool busy = false;
};
std::mutex g_fork_mu;
std::map<int, std::vector<fork_slot *>> g_fork_pool;        // (never destroyed: the runtime may be gone at exit)

fork_slot *fork_acquire()
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lk(g_fork_mu);
    for (fork_slot *f : g_fork_pool[dev])
        if (!f->busy) { f->busy = true; return f; }
    if (g_fork_pool[dev].size() >= 16) return nullptr;
    fork_slot *f = new fork_slot();
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);        // hi = the numerically lowest value = the highest priority
    if (hipStreamCreateWithPriority(&f->s, hipStreamNonBlocking, hi) != hipSuccess ||
        hipEventCreateWithFlags(&f->fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&f->join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        if (f->s) (void)hipStreamDestroy(f->s);
        if (f->fork) (void)hipEventDestroy(f->fork);
        if (f->join) (void)hipEventDestroy(f->join);
        delete f;
        return nullptr;
    }
    f->busy = true;
    g_fork_pool[dev].push_back(f);
    return f;
}
void fork_release(fork_slot *f)
{
    if (!f) return;
    std::lock_guard<std::mutex> lk(g_fork_mu);
    f->busy = false;
}
// fork from `st` (work enqueued on the slot's stream starts behind everything enqueued on st so far)
fork_slot *fork_from(hipStream_t st)
{
    fork_slot *f = fork_acquire();
    if (f && (hipEventRecord(f->fork, st) != hipSuccess || hipStreamWaitEvent(f->s, f->fork, 0) != hipSuccess)) {
        (void)hipGetLastError();
        fork_release(f);
        f = nullptr;
    }
    return f;
}
// A forked chain inside a host function with temporaries: on any exit the side stream is drained before they are released.
struct fork_guard {
    fork_slot *f = nullptr;
    ~fork_guard()
    {
        if (!f) return;
        (void)hipStreamSynchronize(f->s);
        fork_release(f);
    }
    // make `st` wait for the chain, give the slot back
    hipError_t join(hipStream_t st)
    {
        if (!f) return hipSuccess;
        hipError_t e = hipEventRecord(f->join, f->s);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, f->join, 0);
        if (e == hipSuccess) {
            fork_release(f);
            f = nullptr;
        }
        return e;
    }
};
}  // namespace

int phl_tiles_link_vertices(phl_lattice *lat, hipStream_t st)
{
    const int M = (int)lat->M, S = (int)lat->S;
    phl_pinned_reset();                       // (callers have synchronised the stream: nothing of the arena is in flight)
    if (lat->vs_ptr) PHL_HIP(phl_dev_free(lat->vs_ptr));
    if (lat->vs) PHL_HIP(phl_dev_free(lat->vs));
    if (lat->slot_pidx) PHL_HIP(phl_dev_free(lat->slot_pidx));
    lat->vs_ptr = nullptr;
    lat->vs = nullptr;
    lat->slot_pidx = nullptr;
    lat->S_multi = 0;
    PHL_HIP(phl_dev_malloc((void **)&lat->vs_ptr, sizeof(int) * ((size_t)M + 1)));
    PHL_HIP(phl_dev_malloc((void **)&lat->vs, sizeof(phl_contrib_t) * ((size_t)S + 1)));
    PHL_HIP(phl_dev_malloc((void **)&lat->slot_pidx, sizeof(int) * ((size_t)S + 1)));
    if (S == 0) {
        PHL_HIP(hipMemsetAsync(lat->vs_ptr, 0, sizeof(int) * ((size_t)M + 1), st));
        PHL_HIP(hipStreamSynchronize(st));
        return PHL_OK;
    }
    temp_pool tmp;
    int *cnt, *multi, *tile_sums, *sperm;
    PHL_HIP(tmp.get(&cnt, (size_t)M + 1));
    PHL_HIP(tmp.get(&multi, (size_t)S + 1));
    PHL_HIP(tmp.get(&tile_sums, (size_t)(S > M ? S : M) / SCAN_TILE + 2));
    PHL_HIP(tmp.get(&sperm, (size_t)S));
    PHL_HIP(hipMemsetAsync(cnt, 0, sizeof(int) * ((size_t)M + 1), st));
    const unsigned gS = (unsigned)((S + 255) / 256);
    hipLaunchKernelGGL(k_strip_marks, dim3(gS), dim3(256), 0, st, lat->slot_vert, S);
    hipLaunchKernelGGL(k_count_slots, dim3(gS), dim3(256), 0, st, lat->slot_vert, S, cnt);
    PHL_HIP(hipGetLastError());
    int rc = exclusive_scan(cnt, lat->vs_ptr, M, tile_sums, st);
    if (rc) return rc;
    // slots grouped by vertex, ascending slot (= ascending chunk) inside a vertex: a stable sort by vertex id (the
    // marks are stripped: slot_vert itself is the key array)
    rc = stable_sort_perm(lat->slot_vert, S, M, sperm, tmp, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_contrib_and_sole, dim3(gS), dim3(256), 0, st, sperm, lat->slot_vert, S, lat->vs_ptr, lat->vs, multi);
    PHL_HIP(hipGetLastError());
    rc = exclusive_scan(multi, lat->slot_pidx, S, tile_sums, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_fill_vs_rows, dim3(gS), dim3(256), 0, st, lat->vs, S, lat->slot_pidx);
    PHL_HIP(hipGetLastError());
    int pageable_counts[2] = {0, 0};
    int *counts = (int *)phl_pinned_alloc(sizeof(int) * 2);          // {S_multi, n_long}
    if (!counts) counts = pageable_counts;
    PHL_HIP(hipMemcpyAsync(&counts[0], lat->slot_pidx + S, sizeof(int), hipMemcpyDeviceToHost, st));
    // vertices with long slot lists (k_splat_reduce_long)
    int n_long = 0;
    if (lat->vlong) PHL_HIP(phl_dev_free(lat->vlong));
    lat->vlong = nullptr;
    lat->n_long = 0;
    {
        int *lcount;
        PHL_HIP(tmp.get(&lcount, 1));
        PHL_HIP(hipMemsetAsync(lcount, 0, sizeof(int), st));
        PHL_HIP(phl_dev_malloc((void **)&lat->vlong, sizeof(int) * ((size_t)M + 1)));      // worst case; usually almost empty
        hipLaunchKernelGGL(k_append_long, dim3((M + 255) / 256), dim3(256), 0, st, lat->vs_ptr, M, LONG_LIST, lat->vlong, lcount);
        PHL_HIP(hipGetLastError());
        PHL_HIP(hipMemcpyAsync(&counts[1], lcount, sizeof(int), hipMemcpyDeviceToHost, st));
    }
    // (the chunk-major vertex order of the gather splat is made on first use: phl_tiles_ensure_vorder)
    if (lat->vorder) PHL_HIP(phl_dev_free(lat->vorder));
    lat->vorder = nullptr;
    PHL_HIP(hipStreamSynchronize(st));
    const int s_multi = counts[0];
    n_long = counts[1];
    lat->S_multi = s_multi;
    lat->n_long = n_long;
    if (n_long > 64) {
        // longest lists first: k_splat_reduce_long runs one workgroup per vertex, and a 600-row list that starts
        // last is the launch's tail
        int *lkey, *lperm, *lsorted;
        PHL_HIP(tmp.get(&lkey, (size_t)n_long));
        PHL_HIP(tmp.get(&lperm, (size_t)n_long));
        PHL_HIP(tmp.get(&lsorted, (size_t)n_long));
        const unsigned gl = (unsigned)((n_long + 255) / 256);
        const int kmax = lat->nchunks + 1;
        hipLaunchKernelGGL(k_long_keys, dim3(gl), dim3(256), 0, st, lat->vlong, n_long, lat->vs_ptr, kmax, lkey);
        PHL_HIP(hipGetLastError());
        rc = stable_sort_perm(lkey, n_long, (int64_t)kmax + 1, lperm, tmp, st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_gather_i32, dim3(gl), dim3(256), 0, st, lat->vlong, lperm, n_long, lsorted);
        PHL_HIP(hipGetLastError());
        PHL_HIP(hipMemcpyAsync(lat->vlong, lsorted, sizeof(int) * (size_t)n_long, hipMemcpyDeviceToDevice, st));
        PHL_HIP(hipStreamSynchronize(st));
    }
    return PHL_OK;
}

namespace {
int tile_pixels(int dp1)
{
    int P = 2048 / dp1;
    if (P > 256) P = 256;
    P &= ~15;
    if (const char *e = getenv("PHL_TILE_P")) {   // experiments: smaller chunks leave LDS headroom
        const int v = atoi(e) & ~15;
        if (v >= 16 && v <= P) P = v;
    }
    return P;
}

// 1. feature ranges -> the two widest dimensions -> uniform grid with ~P pixels per cell
// 2. pixels in cell-major order (ascending pixel inside a cell): a stable sort of the pixels by cell id.
//    O(n) whatever the features look like -- a constant or heavily clustered `ref` puts (nearly) all
//    pixels into one cell
// Launches only (given the ranges): lat->pix_order, `cell` [n], the grid's cell counts.
template <typename Pool>
int pixel_order(phl_lattice *lat, const float *ref, int64_t rs, int64_t cs, int P, const float *lo, const float *hi, int *cell,
                int *nca_out, int *ncb_out, Pool &tmp, hipStream_t st)
{
    const int d = lat->d, n = (int)lat->n;
    int da = 0, db = -1;
    for (int i = 1; i < d; i++)
        if (hi[i] - lo[i] > hi[da] - lo[da]) da = i;
    for (int i = 0; i < d; i++)
        if (i != da && (db < 0 || hi[i] - lo[i] > hi[db] - lo[db])) db = i;
    double ra = (double)hi[da] - lo[da], rb = db >= 0 ? (double)hi[db] - lo[db] : 0.0;
    if (!(ra > 0) || !isfinite(ra)) ra = 0;
    if (!(rb > 0) || !isfinite(rb)) { rb = 0; db = -1; }
    int nca = 1, ncb = 1;
    float inv_t = 0.f;
    if (ra > 0 && n > P) {
        double T = rb > 0 ? sqrt((double)P * ra * rb / n) : (double)P * ra / n;
        if (T > 0 && isfinite(T)) {
            nca = (int)fmin(ra / T, 32767.0) + 1;
            ncb = rb > 0 ? (int)fmin(rb / T, 32767.0) + 1 : 1;
            while ((int64_t)nca * ncb > (int64_t)4 * n + 1024) {
                T *= 1.5;
                nca = (int)fmin(ra / T, 32767.0) + 1;
                ncb = rb > 0 ? (int)fmin(rb / T, 32767.0) + 1 : 1;
            }
            inv_t = (float)(1.0 / T);
        }
    }
    const int ncell = nca * ncb;
    const unsigned gn = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_cell_ids, dim3(gn), dim3(256), 0, st, ref, rs, cs, (int64_t)n, da, db, lo[da],
                       db >= 0 ? lo[db] : 0.f, inv_t, nca, ncb, cell);
    PHL_HIP(hipGetLastError());
    PHL_HIP(phl_dev_malloc((void **)&lat->pix_order, sizeof(int) * (size_t)n));
    *nca_out = nca;
    *ncb_out = ncb;
    return stable_sort_perm(cell, n, ncell, lat->pix_order, tmp, st);
}
}  // namespace

size_t phl_tiles_pixel_order_scratch_bytes(int64_t n)
{
    // stable_sort_perm: two [256][blocks] histograms, a scan workspace, three [n] arrays (+ alignment slack)
    const size_t nblocks = ((size_t)n + RS_TILE - 1) / RS_TILE;
    return sizeof(int) * (2 * 256 * nblocks + 256 * nblocks / SCAN_TILE + 3 * (size_t)n) + 16 * 1024;
}

int phl_tiles_pixel_order(phl_lattice *lat, const float *ref, int64_t rs, int64_t cs, void *arena, size_t arena_bytes, hipStream_t st)
{
    const int n = (int)lat->n;
    if (n == 0 || !lat->feat_range_valid || lat->pix_order || lat->bt_cell) return PHL_OK;     // (phl_tiles_build does it)
    arena_pool tmp(arena, arena_bytes);
    PHL_HIP(phl_dev_malloc((void **)&lat->bt_cell, sizeof(int) * (size_t)n));
    return pixel_order(lat, ref, rs, cs, tile_pixels(lat->d + 1), lat->feat_lo, lat->feat_hi, lat->bt_cell, &lat->grid_nca,
                       &lat->grid_ncb, tmp, st);
}

int phl_tiles_build(phl_lattice *lat, const float *ref, int64_t rs, int64_t cs, hipStream_t st)
{
    int *pix_ready = lat->bt_cell ? lat->pix_order : nullptr;        // made ahead (phl_tiles_pixel_order): keep it
    if (pix_ready) lat->pix_order = nullptr;
    phl_tiles_free(lat);
    lat->pix_order = pix_ready;
    const int d = lat->d, dp1 = d + 1;
    const int n = (int)lat->n;
    const int P = tile_pixels(dp1);
    lat->P = P;
    if (n == 0) {
        void *scratch = nullptr;
        const int rc0 = phl_rebuild_table_and_neighbors(lat, st, &scratch);
        const int rc1 = rc0 ? rc0 : phl_tiles_link_vertices(lat, st);      // (synchronises the stream)
        if (scratch) { (void)hipStreamSynchronize(st); (void)phl_dev_free(scratch); }
        return rc1;
    }
    int sortn = 512;
    while (sortn < P * dp1) sortn <<= 1;

    int rc = PHL_OK;
    int nchunks = 0;
    int64_t S = 0;
    const int64_t N = lat->N;
    // arrays replaced while launches that read them may still be in flight: released only behind the stream
    // synchronisation at the end of this phase (the block cache may hand a freed block to another thread's build)
    struct deferred_t {
        void *p[3] = {nullptr, nullptr, nullptr};
        ~deferred_t() { for (void *q : p) if (q) (void)phl_dev_free(q); }
    } deferred;
    fork_guard nbr_chain;                            // (declared behind `deferred`: drained before those blocks are released)
    {   // temporaries of the chunk build go back to the scratch cache before the vertex lists are linked
    temp_pool tmp;
    // 1.-2. the pixel order (pixel_order above), unless it has been made under the table replay already
    int nca = 1, ncb = 1;
    int *cell, *tile_sums;
    PHL_HIP(tmp.get(&tile_sums, (size_t)n / SCAN_TILE + 2));
    if (pix_ready) {
        cell = lat->bt_cell;
        nca = lat->grid_nca;
        ncb = lat->grid_ncb;
    } else {
        std::vector<float> lo(d, INFINITY), hi(d, -INFINITY);
        if (lat->feat_range_valid) {          // found while elevating (phl_build_device)
            for (int i = 0; i < d; i++) { lo[i] = lat->feat_lo[i]; hi[i] = lat->feat_hi[i]; }
        } else {
            constexpr int MMB = 1024;
            float *mm_dev;
            PHL_HIP(tmp.get(&mm_dev, (size_t)MMB * d * 2));
            hipLaunchKernelGGL(k_minmax, dim3(MMB), dim3(256), 0, st, ref, rs, cs, (int64_t)n, d, mm_dev);
            PHL_HIP(hipGetLastError());
            std::vector<float> mm((size_t)MMB * d * 2);
            PHL_HIP(hipMemcpyAsync(mm.data(), mm_dev, sizeof(float) * mm.size(), hipMemcpyDeviceToHost, st));
            PHL_HIP(hipStreamSynchronize(st));
            for (int b = 0; b < MMB; b++)
                for (int i = 0; i < d; i++) {
                    lo[i] = fminf(lo[i], mm[((size_t)b * d + i) * 2]);
                    hi[i] = fmaxf(hi[i], mm[((size_t)b * d + i) * 2 + 1]);
                }
        }
        PHL_HIP(tmp.get(&cell, (size_t)n));
        rc = pixel_order(lat, ref, rs, cs, P, lo.data(), hi.data(), cell, &nca, &ncb, tmp, st);
        if (rc) return rc;
    }
    const int ncell = nca * ncb;

    // 2b. internal vertex numbering (see k_vertex_home), then the key -> vertex table and the blur neighbours
    {
        static const bool renumber = !(getenv("PHL_RENUMBER") && atoi(getenv("PHL_RENUMBER")) == 0);
        const int M_all = (int)lat->M;
        // a band cut out of the whole image's lattice (phl_sub_lattice) comes with ghost vertices behind its own ones:
        // only the own vertices are renumbered, the ghosts keep their rows (and the caller's order)
        const int M = (lat->M_local > 0 && lat->M_local < lat->M) ? (int)lat->M_local : M_all;
        // fresh build (phl_build_device's tables are there): the candidates' vertex ids are written once, below,
        // through the locality numbering -- unless the vertices' homes have to be read off replay[] first
        bool from_tables = lat->bt_slot_of != nullptr;
        bool wrote_vids = false;
        if (from_tables && !lat->vfirst) {
            rc = phl_write_final_vids(lat, st);      // (int_of_ft is null here: first-touch / reference ids)
            if (rc) return rc;
            from_tables = false;
        }
        if (renumber && ncell > 8 && M > 1) {
            int *vhome, *vkey;
            PHL_HIP(tmp.get(&vhome, (size_t)M));
            PHL_HIP(tmp.get(&vkey, (size_t)M));
            if (lat->vfirst) {
                hipLaunchKernelGGL(k_vertex_home_first, dim3((M + 255) / 256), dim3(256), 0, st, lat->vfirst, M, dp1, cell, vhome);
            } else {        // reference-table mode with duplicates: smallest cell among the touching pixels
                hipLaunchKernelGGL(k_fill_i32, dim3(256), dim3(256), 0, st, vhome, (int64_t)M, 0x7FFFFFFF);
                hipLaunchKernelGGL(k_vertex_home, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, lat->replay, (int)N, dp1, cell, vhome);
            }
            static const int nstrips = getenv("PHL_STRIPS") ? atoi(getenv("PHL_STRIPS")) : 8;
            const int stripw = (nca + nstrips - 1) / nstrips;
            hipLaunchKernelGGL(k_strip_key, dim3((M + 255) / 256), dim3(256), 0, st, vhome, M, nca, ncb, stripw, vkey);
            PHL_HIP(hipGetLastError());
            PHL_HIP(phl_dev_malloc((void **)&lat->ft_of_int, sizeof(int) * (size_t)M_all));
            PHL_HIP(phl_dev_malloc((void **)&lat->int_of_ft, sizeof(int) * (size_t)M_all));
            rc = stable_sort_perm(vkey, M, (int64_t)(nstrips + 1) * ncb * stripw, lat->ft_of_int, tmp, st);
            if (rc) return rc;
            if (M_all > M)        // ghosts: identity
                hipLaunchKernelGGL(k_iota_tail, dim3((M_all - M + 255) / 256), dim3(256), 0, st, lat->ft_of_int, M, M_all);
            hipLaunchKernelGGL(k_invert_perm, dim3((M_all + 255) / 256), dim3(256), 0, st, lat->ft_of_int, M_all, lat->int_of_ft);
            int16_t *vkeys_new;
            PHL_HIP(phl_dev_malloc((void **)&vkeys_new, sizeof(int16_t) * (size_t)M_all * d));
            hipLaunchKernelGGL(k_permute_keys, dim3((unsigned)(((int64_t)M_all * d + 255) / 256)), dim3(256), 0, st, lat->vkeys,
                               lat->ft_of_int, M_all, d, vkeys_new);
            if (from_tables) {
                rc = phl_write_final_vids(lat, st);
                if (rc) return rc;
                wrote_vids = true;
            } else {
                hipLaunchKernelGGL(k_relabel_replay, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, lat->replay, (int)N,
                                   lat->int_of_ft);
            }
            PHL_HIP(hipGetLastError());
            deferred.p[0] = lat->vkeys;              // still being read by the gather above
            lat->vkeys = vkeys_new;
        }
        if (from_tables && !wrote_vids) {            // no locality numbering: first-touch (reference) ids as they are
            rc = phl_write_final_vids(lat, st);
            if (rc) return rc;
        }
        deferred.p[1] = lat->vfirst;                 // build-time only
        lat->vfirst = nullptr;
        // key -> vertex table, packed keys, blur neighbours, composed pairs: ~100 us of small dependent launches that
        // nothing in the chunk build below depends on -- on a forked stream beside it, joined before this phase ends
        static const bool side_nbr = !(getenv("PHL_SIDE_NEIGHBORS") && atoi(getenv("PHL_SIDE_NEIGHBORS")) == 0);
        if (side_nbr) nbr_chain.f = fork_from(st);
        rc = phl_rebuild_table_and_neighbors(lat, nbr_chain.f ? nbr_chain.f->s : st, &deferred.p[2]);
        if (rc) return rc;
    }

    // 3. per-chunk local vertex lists, segments and local indices
    nchunks = (n + P - 1) / P;
    lat->nchunks = nchunks;
    int *nv;
    PHL_HIP(tmp.get(&nv, (size_t)nchunks + 1));
    PHL_HIP(phl_dev_malloc((void **)&lat->chunk_vptr, sizeof(int) * ((size_t)nchunks + 1)));
#define PHL_CHUNK_MASKS_X(HTX_, ...)                                                                                    \
    switch (sortn) {                                                                                                     \
        case 512: hipLaunchKernelGGL((k_chunk_masks<512, HTX_>), dim3(nchunks), dim3(256), 0, st, __VA_ARGS__); break;    \
        case 1024: hipLaunchKernelGGL((k_chunk_masks<1024, HTX_>), dim3(nchunks), dim3(256), 0, st, __VA_ARGS__); break;  \
        default: hipLaunchKernelGGL((k_chunk_masks<2048, HTX_>), dim3(nchunks), dim3(256), 0, st, __VA_ARGS__); break;    \
    }
#define PHL_CHUNK_MASKS(...)                                           \
    if (P * dp1 * 4 > sortn * 3) { PHL_CHUNK_MASKS_X(2, __VA_ARGS__) } \
    else { PHL_CHUNK_MASKS_X(1, __VA_ARGS__) }
#define PHL_CHUNK_SORT(WRITE_, ...)                                                                                      \
    switch (sortn) {                                                                                                      \
        case 512: hipLaunchKernelGGL((k_chunk_group<512, WRITE_>), dim3(nchunks), dim3(256), 0, st, __VA_ARGS__); break;   \
        case 1024: hipLaunchKernelGGL((k_chunk_group<1024, WRITE_>), dim3(nchunks), dim3(256), 0, st, __VA_ARGS__); break; \
        default: hipLaunchKernelGGL((k_chunk_group<2048, WRITE_>), dim3(nchunks), dim3(256), 0, st, __VA_ARGS__); break;   \
    }
    // One grouping pass: segments and local indices go to their final arrays, the per-chunk slot records to a
    // scratch area with a fixed stride; they are compacted once the chunk offsets are known.  Only when a
    // chunk has more local vertices than the stride (pixels that share next to nothing) the pass is repeated.
    constexpr int SLOT_STRIDE = 384;
    int *t_vert;
    int2 *t_rng;
    PHL_HIP(tmp.get(&t_vert, (size_t)nchunks * SLOT_STRIDE));
    PHL_HIP(tmp.get(&t_rng, (size_t)nchunks * SLOT_STRIDE));
    PHL_HIP(phl_dev_malloc((void **)&lat->seg, sizeof(phl_contrib_t) * (size_t)N));
    PHL_HIP(phl_dev_malloc((void **)&lat->lidx, sizeof(unsigned short) * (size_t)N));
    // k_chunk_masks does every chunk with at most NVC local vertices; heavier ones only report their count here and
    // are done by k_chunk_group below, once the counts are on the host
    static const bool use_masks = !(getenv("PHL_CHUNK_MASKS") && atoi(getenv("PHL_CHUNK_MASKS")) == 0);
    if (use_masks) {
        PHL_CHUNK_MASKS(lat->pix_order, n, P, dp1, lat->replay, nv, (const int *)nullptr, SLOT_STRIDE, t_vert, t_rng, lat->seg,
                        lat->lidx)
    } else {
        PHL_CHUNK_SORT(true, lat->pix_order, n, P, dp1, lat->replay, nv, (const int *)nullptr, SLOT_STRIDE, t_vert, t_rng,
                       lat->seg, lat->lidx, (const int *)nullptr, 0)
    }
    PHL_HIP(hipGetLastError());
    rc = exclusive_scan(nv, lat->chunk_vptr, nchunks, tile_sums, st);
    if (rc) return rc;
    // (pinned staging where there is some: a copy into pageable memory would block the host twice)
    std::vector<int> nv_pageable, by_nv_pageable;
    int *nv_host = (int *)phl_pinned_alloc(sizeof(int) * (size_t)nchunks);
    int *by_nv = (int *)phl_pinned_alloc(sizeof(int) * (size_t)nchunks);
    if (!nv_host) { nv_pageable.resize((size_t)nchunks); nv_host = nv_pageable.data(); }
    if (!by_nv) { by_nv_pageable.resize((size_t)nchunks); by_nv = by_nv_pageable.data(); }
    PHL_HIP(hipMemcpyAsync(nv_host, nv, sizeof(int) * (size_t)nchunks, hipMemcpyDeviceToHost, st));
    PHL_HIP(hipStreamSynchronize(st));
    int nv_max = 0;
    for (int c = 0; c < nchunks; c++) {
        const int v = nv_host[c];
        S += v;
        if (v > nv_max) nv_max = v;
    }
    lat->S = S;
    lat->nv_max = nv_max;
    // chunk classes (plan_tiles): cumulative histogram of the vertex counts on the host, chunk ids by descending
    // vertex count on the device (counting sort; ascending chunk id among equals)
    {
        lat->nv_cum = (int *)malloc(sizeof(int) * ((size_t)nv_max + 2));
        if (!lat->nv_cum) { phl_set_error("phl_tiles_build: out of host memory"); return PHL_ERR_HIP; }
        std::vector<int> start((size_t)nv_max + 2, 0);
        for (int c = 0; c < nchunks; c++) start[(size_t)(nv_max - nv_host[c]) + 1]++;         // bin 0 = heaviest
        for (int b = 0; b <= nv_max; b++) start[(size_t)b + 1] += start[(size_t)b];
        for (int x = 0; x <= nv_max; x++) lat->nv_cum[x] = nchunks - start[(size_t)(nv_max - x)];   // #chunks with nv <= x
        for (int c = 0; c < nchunks; c++) by_nv[start[(size_t)(nv_max - nv_host[c])]++] = c;
        PHL_HIP(phl_dev_malloc((void **)&lat->chunk_by_nv, sizeof(int) * ((size_t)nchunks + 1)));
        PHL_HIP(hipMemcpyAsync(lat->chunk_by_nv, by_nv, sizeof(int) * (size_t)nchunks, hipMemcpyHostToDevice, st));
    }
    PHL_HIP(phl_dev_malloc((void **)&lat->slot_vert, sizeof(int) * ((size_t)S + 1)));
    PHL_HIP(phl_dev_malloc((void **)&lat->seg_rng, sizeof(int2) * ((size_t)S + 1)));
    if (nv_max <= SLOT_STRIDE) {
        if (use_masks && nv_max > NVC) {     // the chunks k_chunk_masks left out (first-pass form: slot records to the scratch)
            PHL_CHUNK_SORT(true, lat->pix_order, n, P, dp1, lat->replay, (int *)nullptr, (const int *)nullptr, SLOT_STRIDE, t_vert,
                           t_rng, lat->seg, lat->lidx, (const int *)nv, NVC)
        }
        hipLaunchKernelGGL(k_compact_slots, dim3(nchunks), dim3(256), 0, st, lat->chunk_vptr, nchunks, SLOT_STRIDE, t_vert,
                           t_rng, lat->slot_vert, lat->seg_rng);
    } else {
        if (use_masks) {
            PHL_CHUNK_MASKS(lat->pix_order, n, P, dp1, lat->replay, (int *)nullptr, lat->chunk_vptr, 0, lat->slot_vert,
                            lat->seg_rng, lat->seg, lat->lidx)
            if (nv_max > NVC) {
                PHL_CHUNK_SORT(true, lat->pix_order, n, P, dp1, lat->replay, (int *)nullptr, lat->chunk_vptr, 0, lat->slot_vert,
                               lat->seg_rng, lat->seg, lat->lidx, (const int *)nv, NVC)
            }
        } else {
            PHL_CHUNK_SORT(true, lat->pix_order, n, P, dp1, lat->replay, (int *)nullptr, lat->chunk_vptr, 0, lat->slot_vert,
                           lat->seg_rng, lat->seg, lat->lidx, (const int *)nullptr, 0)
        }
    }
#undef PHL_CHUNK_SORT
#undef PHL_CHUNK_MASKS
#undef PHL_CHUNK_MASKS_X
    PHL_HIP(hipGetLastError());
    PHL_HIP(nbr_chain.join(st));
    PHL_HIP(hipStreamSynchronize(st));
    }
    phl_release_build_tables(lat);                 // (read by k_final_vid: behind the synchronisation)
    rc = phl_tiles_link_vertices(lat, st);
    if (rc) return rc;
    lat->table_bytes = (int64_t)(sizeof(int16_t) * (size_t)lat->M * d + sizeof(phl_replay_t) * (size_t)N +
                                 sizeof(int32_t) * (size_t)lat->M * (d + 1) * 2 + sizeof(int32_t) * (size_t)lat->M * ((d + 1) / 2) * 8 +
                                 sizeof(int) * ((size_t)lat->table_mask + 1) + (lat->int_of_ft ? 2 * sizeof(int) * (size_t)lat->M : 0));
    lat->tile_bytes = (int64_t)(sizeof(int) * ((size_t)n + nchunks + 1 + 3 * ((size_t)S + 1) + (size_t)lat->M + 1) +
                                sizeof(phl_contrib_t) * ((size_t)N + S + 1) + sizeof(unsigned short) * (size_t)N);
    return PHL_OK;
}

// Is the LDS-staged path available for this channel count?
int phl_tiles_lprs(const phl_lattice *lat, int vd, int for_slice)
{
    if (lat->nchunks == 0 || vd % 4 != 0) return -1;
    if (for_slice) return 1 << plan_slice(lat, vd).lsh_max;      // any chunk: the workgroup picks its own slab width
    return pick_cfg(lat, vd, lat->nv_max).lprs;
}

// which chunks hold a slot of any of the listed vertex rows
__global__ __launch_bounds__(256) void k_mark_chunks(const int64_t *__restrict__ rows, int64_t k, const int *__restrict__ vs_ptr,
                                                     const phl_contrib_t *__restrict__ vs, const int *__restrict__ chunk_vptr,
                                                     int nchunks, int *__restrict__ mask)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const int64_t v = rows[i];
    for (int e = vs_ptr[v]; e < vs_ptr[v + 1]; e++) {
        const int slot = vs[e].pixel;
        int lo = 0, hi = nchunks - 1;                     // chunk_vptr[c] <= slot < chunk_vptr[c + 1]
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (chunk_vptr[mid] <= slot) lo = mid;
            else hi = mid - 1;
        }
        mask[lo] = 1;
    }
}

int phl_tiles_chunks_touching(phl_lattice *lat, const int64_t *rows_dev, int64_t k, int32_t *mask_host, hipStream_t st)
{
    const int nchunks = lat->nchunks;
    if (nchunks == 0) return PHL_OK;
    temp_pool tmp;
    int *mask;
    PHL_HIP(tmp.get(&mask, (size_t)nchunks));
    PHL_HIP(hipMemsetAsync(mask, 0, sizeof(int) * (size_t)nchunks, st));
    if (k > 0) {
        hipLaunchKernelGGL(k_mark_chunks, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, st, rows_dev, k, lat->vs_ptr, lat->vs,
                           lat->chunk_vptr, nchunks, mask);
        PHL_HIP(hipGetLastError());
    }
    PHL_HIP(hipMemcpyAsync(mask_host, mask, sizeof(int) * (size_t)nchunks, hipMemcpyDeviceToHost, st));
    PHL_HIP(hipStreamSynchronize(st));
    return PHL_OK;
}

int phl_launch_splat_tiled(phl_lattice *lat, const float *src, int64_t src_rs, int vd, float *vert, float *partial,
                           hipStream_t st, bool subset, const int *chunk_list, int nlist, const int *vlist, int64_t nvl,
                           const phl_splat_wide *wide, const int *pack_pos, float *pack, int64_t pack_rs)
{
    if (pack_pos && (!subset || wide)) { phl_set_error("tiled splat: row packing needs a row list and a plain splat"); return PHL_ERR_INVALID; }
    // wide: vertex / partial rows of wide->nsets * vd floats, block k+1 = splat of src (x) fref[:, k] (k_splat_tiled)
    const int64_t out_rs = (int64_t)vd * (wide ? wide->nsets : 1);
    // subset: run only the listed chunks, then complete only the listed vertex rows (either list may be empty)
    const int M = subset ? (int)nvl : (int)lat->M;
    if (lat->M == 0 || vd == 0) return PHL_OK;
    const tile_plan plan = plan_tiles(lat, vd);
    if (plan.n == 0) {
        phl_set_error("tiled splat: chunk does not fit LDS");
        return PHL_ERR_UNSUPPORTED;
    }
    int rc = PHL_OK;
    const int nrun = subset ? nlist : lat->nchunks;
    static const char *tl_path = getenv("PHL_TIMELINE");     // debug: dump per-workgroup time stamps of the main launch
    unsigned long long *tl = nullptr;
    size_t tl_n = 0;
    // the small classes go first, on a forked high-priority stream, and run beside the main grid (see fork_slot)
    static const bool side_classes = !(getenv("PHL_SIDE_CLASSES") && atoi(getenv("PHL_SIDE_CLASSES")) == 0);
    fork_slot *fk = (plan.n > 1 && !subset && !tl_path && side_classes) ? fork_from(st) : nullptr;
    hipStream_t const st_main = st;
    int order[12];
    int no = 0;
    for (int ci = 0; ci < plan.n; ci++)
        if (!plan.cls[ci].full_grid) order[no++] = ci;
    for (int ci = 0; ci < plan.n; ci++)
        if (plan.cls[ci].full_grid) order[no++] = ci;
    for (int oi = 0; oi < plan.n && rc == PHL_OK && nrun > 0; oi++) {
        const int ci = fk ? order[oi] : oi;
        const tile_class &c = plan.cls[ci];
        st = (fk && !c.full_grid) ? fk->s : st_main;
        // the largest class (and every class of a caller's subset) walks the whole grid / list in chunk order and
        // filters by vertex count in the kernel; the others run exactly their range of chunk_by_nv
        const bool filtered = subset || c.full_grid;
        const int *list = subset ? chunk_list : (c.full_grid ? nullptr : lat->chunk_by_nv + c.begin);
        const int cnt = filtered ? nrun : c.count;
        if (cnt == 0) continue;
        unsigned cgrid;
        int xcd_chunk;
        chunk_grid(cnt, &cgrid, &xcd_chunk);
        unsigned long long *tlc = nullptr;
        if (tl_path && c.full_grid && !tl) {
            tl_n = (size_t)cgrid * 8;
            PHL_HIP(phl_dev_malloc((void **)&tl, tl_n * 8));
            PHL_HIP(hipMemsetAsync(tl, 0, tl_n * 8, st));
            tlc = tl;
        }
        const char *env1 = getenv("PHL_WIDE_ONE_PHASE");            // read per call: the parity test toggles it
        const bool one_phase = !(env1 && atoi(env1) == 0);
        if (wide && one_phase && c.cfg.lprs == 16 && wide->nsets >= 2 && wide->nsets <= 8 && lat->P <= 256) {
            // every LDS row read feeds all nsets accumulators (k_splat_wide)
#define PHL_SW(NS_)                                                                                                           \
    case NS_:                                                                                                                 \
        if ((rc = allow_lds(k_splat_wide<NS_>, c.cfg.lds)) != PHL_OK) break;                                                  \
        k_splat_wide<NS_><<<dim3(cgrid), dim3(TPB_S), c.cfg.lds, st>>>(                                                       \
            src, src_rs, vd, (int)lat->n, lat->P, c.cfg.cap, lat->pix_order, lat->chunk_vptr, lat->slot_vert, lat->slot_pidx, \
            lat->seg_rng, lat->seg, vert, partial, cnt, xcd_chunk, list, c.lo, c.hi, long_seg(), wide->fref, wide->rs,        \
            wide->cs);                                                                                                        \
        break;
            switch (wide->nsets) { PHL_SW(2) PHL_SW(3) PHL_SW(4) PHL_SW(5) PHL_SW(6) PHL_SW(7) PHL_SW(8) }
#undef PHL_SW
            continue;
        }
        dispatch_lprs(c.cfg.lprs, [&](auto L) {
            constexpr int LPRS = decltype(L)::value;
            if ((rc = allow_lds(k_splat_tiled<LPRS>, c.cfg.lds)) != PHL_OK) return;
            k_splat_tiled<LPRS><<<dim3(cgrid), dim3(TPB_S), c.cfg.lds, st>>>(
                src, src_rs, vd, (int)lat->n, lat->P, lat->d + 1, c.cfg.cap, lat->pix_order, lat->chunk_vptr, lat->slot_vert,
                lat->slot_pidx, lat->seg_rng, lat->seg, vert, partial, cnt, xcd_chunk, tlc, list, c.lo, c.hi, long_seg(),
                wide ? wide->nsets : 1, wide ? wide->fref : nullptr, wide ? wide->rs : 0, wide ? wide->cs : 0, out_rs);
        });
    }
    st = st_main;
    if (fk) {
        hipError_t e = hipEventRecord(fk->join, fk->s);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, fk->join, 0);      // the reduction below needs every class's sums
        fork_release(fk);
        if (e != hipSuccess) return phl_hip_fail(e, "joining the side stream of the chunk classes", __FILE__, __LINE__);
    }
    if (tl) {                                                 // debug only
        std::vector<unsigned long long> h(tl_n);
        PHL_HIP(hipMemcpyAsync(h.data(), tl, tl_n * 8, hipMemcpyDeviceToHost, st));
        PHL_HIP(hipStreamSynchronize(st));
        (void)phl_dev_free(tl);
        if (FILE *f = fopen(tl_path, "wb")) {
            fwrite(h.data(), 8, tl_n, f);
            fclose(f);
        }
    }
    if (rc) return rc;
    if (M == 0) return PHL_OK;
    const int lpr = pick_lpr_row((int)out_rs);
    int64_t waves = ((int64_t)M + (64 / lpr) - 1) / (64 / lpr);
    int64_t blocks = (waves + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    // long slot lists: the lattice's own list of such vertices, or (subset) the caller's rows, short ones exiting
    // (a subset call walks its own rows only if the lattice has long vertices at all: one workgroup per listed row is
    //  tens of thousands of empty workgroups on the row-band path otherwise)
    // (with a send buffer to fill, the long kernel walks the caller's rows even when ... see above: n_long == 0 means no
    //  listed row can be long, so nothing is lost by skipping it)
    const int *llist = subset ? vlist : lat->vlong;
    const int64_t nl = lat->n_long == 0 ? 0 : (subset ? nvl : lat->n_long);
    const int long_list = llist ? LONG_LIST : 0x7FFFFFFF;
#define PHL_RED(LPR_)                                                                                                  \
    k_splat_reduce<LPR_><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(partial, lat->vs_ptr, lat->vs, lat->slot_pidx, \
                                                                       M, (int)out_rs, vert, vlist, long_list,       \
                                                                       pack_pos, pack, pack_rs);                     \
    if (llist && nl > 0)                                                                                               \
        k_splat_reduce_long<LPR_><<<dim3((unsigned)nl), dim3(256), 0, st>>>(partial, lat->vs_ptr, lat->vs,            \
                                                                            lat->slot_pidx, (int)out_rs, vert, llist, long_list, \
                                                                            subset ? pack_pos : nullptr, pack, pack_rs)
    switch (lpr) {
        case 64: PHL_RED(64); break;
        case 16: PHL_RED(16); break;
        case 4: PHL_RED(4); break;
        default: PHL_RED(1); break;
    }
#undef PHL_RED
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}

int phl_launch_slice_tiled(const phl_lattice *lat, const float *vert, int vd, float *out, int64_t out_rs, const float *sub,
                           int64_t sub_rs, unsigned flags, hipStream_t st)
{
    if (lat->n == 0 || vd == 0) return PHL_OK;
    const slice_cfg cfg = plan_slice(lat, vd);
    const float cdiv = 1 + powf(2, -lat->d);  // permutohedral.h:480
    const float rcdiv = 1.0f / cdiv;
    const bool exact = (flags & PHL_FILTER_EXACT) != 0;
    int rc = PHL_OK;
    unsigned cgrid;
    int xcd_chunk;
    chunk_grid(lat->nchunks, &cgrid, &xcd_chunk);
    // heavy chunks = those that cannot take at least half the widest slab; worth a head start only if they are few
    int heavy_thr = 0x7FFFFFFF, heavy_n = 0;
    if (lat->nv_cum && lat->chunk_by_nv && cfg.lsh_max > 2) {
        const int64_t fixed0 = (int64_t)lat->P * (lat->d + 1) * 8 + (int64_t)lat->P * 4 + 16;
        const int thr = (int)(((int64_t)cfg.lds - fixed0) / ((16 << (cfg.lsh_max - 1)) + 4));
        const int above = chunks_above(lat, thr);
        if (above > 0 && (int64_t)above * 16 <= lat->nchunks) { heavy_thr = thr; heavy_n = above; }
    }
    const int heavy_pad = (heavy_n + 7) & ~7;
    cgrid += (unsigned)heavy_pad;
    if (exact) {
        if ((rc = allow_lds(k_slice_tiled<true>, cfg.lds)) != PHL_OK) return rc;
        k_slice_tiled<true><<<dim3(cgrid), dim3(TPB), cfg.lds, st>>>(
            vert, vd, (int)lat->n, lat->P, lat->d + 1, cfg.lsh_max, (int)cfg.lds, lat->pix_order, lat->chunk_vptr, lat->slot_vert,
            lat->lidx, lat->replay, out, out_rs, sub, sub_rs, cdiv, rcdiv, lat->nchunks, xcd_chunk, lat->chunk_by_nv, heavy_n,
            heavy_pad, heavy_thr);
    } else {
        if ((rc = allow_lds(k_slice_tiled<false>, cfg.lds)) != PHL_OK) return rc;
        k_slice_tiled<false><<<dim3(cgrid), dim3(TPB), cfg.lds, st>>>(
            vert, vd, (int)lat->n, lat->P, lat->d + 1, cfg.lsh_max, (int)cfg.lds, lat->pix_order, lat->chunk_vptr, lat->slot_vert,
            lat->lidx, lat->replay, out, out_rs, sub, sub_rs, cdiv, rcdiv, lat->nchunks, xcd_chunk, lat->chunk_by_nv, heavy_n,
            heavy_pad, heavy_thr);
    }
    if (rc) return rc;
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}

// phl_filter_grad's last stage (see k_slice_grad).  d <= 7; P <= 256 (eight pixels per 16-lane group).
int phl_launch_slice_grad(const phl_lattice *lat, const float *vertw, int L, const float *y, int64_t y_rs, const float *ref,
                          int64_t ref_rs, int64_t ref_cs, float *grad_ref, int accumulate, float *wx_out, int64_t wx_rs,
                          hipStream_t st)
{
    if (lat->n == 0 || L == 0) return PHL_OK;
    const int dp1 = lat->d + 1;
    if (dp1 < 2 || dp1 > 8 || lat->P > 256 || lat->nchunks == 0) {
        phl_set_error("fused feature gradient: d = %d not supported (1..7)", lat->d);
        return PHL_ERR_UNSUPPORTED;
    }
    const int64_t fixed = (((int64_t)lat->P * dp1 * 8 + (int64_t)lat->P * 4 + (int64_t)lat->nv_max * 4 + 15) & ~(int64_t)15);
    int64_t lds = fixed + (int64_t)lat->nv_max * dp1 * 128;      // every chunk on 32-channel slabs, if the budget allows
    if (lds > lds_budget()) lds = lds_budget();
    const float rcdiv = 1.0f / (1 + powf(2, -lat->d));      // permutohedral.h:480
    unsigned cgrid;
    int xcd_chunk;
    chunk_grid(lat->nchunks, &cgrid, &xcd_chunk);
    int rc = PHL_OK;
#define PHL_SG(NS_)                                                                                                          \
    case NS_:                                                                                                                \
        if ((rc = allow_lds(k_slice_grad<NS_>, (size_t)lds)) != PHL_OK) return rc;                                           \
        k_slice_grad<NS_><<<dim3(cgrid), dim3(TPB), (size_t)lds, st>>>(vertw, L, (int)lat->n, lat->P, (int)lds,              \
            lat->pix_order, lat->chunk_vptr, lat->slot_vert, lat->lidx, lat->replay, y, y_rs, ref, ref_rs, ref_cs, grad_ref, \
            accumulate, wx_out, wx_rs, rcdiv, lat->nchunks, xcd_chunk);                                                      \
        break;
    switch (dp1) {
        PHL_SG(2) PHL_SG(3) PHL_SG(4) PHL_SG(5) PHL_SG(6) PHL_SG(7) PHL_SG(8)
    }
#undef PHL_SG
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}
