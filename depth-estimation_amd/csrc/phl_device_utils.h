// Device utilities shared by the phl translation units (each TU gets its own copy: everything
// here lives in an anonymous namespace).
#pragma once
#include <vector>

#include "phl_internal.h"

namespace {

// Run aggregation for atomics whose keys repeat in CONSECUTIVE lanes (neighbouring pixels hit the
// same lattice vertex / grid cell): only the first lane of a run touches memory.
// head_of_run: lane index of the first lane of my run; run_len (valid on head lanes): its length.
__device__ __forceinline__ void wave_runs(int key, bool active, int *head_lane, int *run_len)
{
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(key, 1);
    const bool prev_active = __shfl_up((int)active, 1) != 0;
    const bool head = active && (lane == 0 || !prev_active || prev != key);
    const unsigned long long heads = __ballot(head);
    const unsigned long long act = __ballot(active);
    const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    *head_lane = 63 - __clzll(below ? below : 1ull);
    // end of run = next head above me, or the end of the active lanes
    const unsigned long long above = (lane == 63) ? 0ull : (heads >> (lane + 1));
    const int next_head = above ? lane + 1 + __ffsll((long long)above) - 1 : 64;
    const int last_active = act ? 64 - __clzll(act) : 0;   // one past the highest active lane
    *run_len = min(next_head, last_active) - lane;
}

// counters[key] += (run length); returns the value before the add plus my rank inside the run
__device__ __forceinline__ int run_atomic_add(int *counters, int key, bool active)
{
    int head_lane, run_len;
    wave_runs(key, active, &head_lane, &run_len);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (active && lane == head_lane) base = atomicAdd(&counters[key], run_len);
    base = __shfl(base, head_lane);
    return base + (lane - head_lane);
}

__attribute__((unused)) __global__ void k_fill_i32(int *p, int64_t n, int value)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = value;
}

// ------------------------------------------------------------------------------------------
// exclusive scan of int32 (three launches; out has n+1 entries, out[n] = total)
constexpr int SCAN_T = 256, SCAN_I = 8, SCAN_TILE = SCAN_T * SCAN_I;

__device__ __forceinline__ int block_exclusive_scan(int x, int *total)
{
    __shared__ int wsum[SCAN_T / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_T / 64; i++) {
        if (i < w) base += wsum[i];
        tot += wsum[i];
    }
    __syncthreads();
    *total = tot;
    return base + incl - x;
}

__global__ __launch_bounds__(SCAN_T) void k_scan_tile(const int *__restrict__ in, int *__restrict__ out,
                                                      int *__restrict__ tile_sums, int n)
{
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_I;
    int v[SCAN_I], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_I; i++) {
        v[i] = (base + i) < n ? in[base + i] : 0;
        s += v[i];
    }
    int tot;
    int ex = block_exclusive_scan(s, &tot);
#pragma unroll
    for (int i = 0; i < SCAN_I; i++) {
        if ((base + i) < n) out[base + i] = ex;
        ex += v[i];
    }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_T) void k_scan_sums(int *tile_sums, int ntiles, int *total_out)
{
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < ntiles; base += SCAN_T) {
        const int i = base + threadIdx.x;
        const int x = i < ntiles ? tile_sums[i] : 0;
        int tot;
        const int ex = block_exclusive_scan(x, &tot);
        const int carry = carry_s;
        if (i < ntiles) tile_sums[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry_s;
}

__global__ __launch_bounds__(SCAN_T) void k_scan_add(int *__restrict__ out, const int *__restrict__ tile_sums, int n)
{
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_I;
    const int add = tile_sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_I; i++)
        if ((base + i) < n) out[base + i] += add;
}

// Up to SCAN_FUSE_TILES tiles (2 M entries) the middle launch is not needed: a consumer of the tile-local scan adds up
// the tile sums in front of its tile itself (four per thread and one block scan).  The build is a chain of dependent
// few-microsecond launches, each of which costs the dispatch-to-dispatch minimum (profiles/r04g_build_timeline.txt):
// every launch less is ~4.6 us off the critical path.  Integer sums: the results are the same bits.
constexpr int SCAN_FUSE_TILES = 4 * SCAN_T;

// exclusive prefix of tile_sums[0 .. ntiles), ntiles <= SCAN_FUSE_TILES, into pre[] (LDS); all SCAN_T threads call
__device__ __forceinline__ void tile_prefix_to_lds(const int *__restrict__ tile_sums, int ntiles, int *pre)
{
    const int b4 = threadIdx.x * 4;
    int v[4], s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        v[i] = (b4 + i) < ntiles ? tile_sums[b4 + i] : 0;
        s += v[i];
    }
    int tot;
    int ex = block_exclusive_scan(s, &tot);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        pre[b4 + i] = ex;
        ex += v[i];
    }
    __syncthreads();
}

// k_scan_sums + k_scan_add in one launch (ntiles <= SCAN_FUSE_TILES): tile_sums stays as k_scan_tile wrote it
__global__ __launch_bounds__(SCAN_T) void k_scan_add_tiles(int *__restrict__ out, const int *__restrict__ tile_sums, int n)
{
    int s = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += SCAN_T) s += tile_sums[i];
    int add;
    (void)block_exclusive_scan(s, &add);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = add + tile_sums[blockIdx.x];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_I;
#pragma unroll
    for (int i = 0; i < SCAN_I; i++)
        if ((base + i) < n) out[base + i] += add;
}

__attribute__((unused)) int exclusive_scan(const int *in, int *out /* n+1 */, int n, int *tile_sums, hipStream_t st)
{
    if (n <= 0) {
        PHL_HIP(hipMemsetAsync(out, 0, sizeof(int), st));
        return PHL_OK;
    }
    const int ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_tile, dim3(ntiles), dim3(SCAN_T), 0, st, in, out, tile_sums, n);
    if (ntiles <= SCAN_FUSE_TILES) {
        hipLaunchKernelGGL(k_scan_add_tiles, dim3(ntiles), dim3(SCAN_T), 0, st, out, tile_sums, n);
        PHL_HIP(hipGetLastError());
        return PHL_OK;
    }
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_T), 0, st, tile_sums, ntiles, out + n);
    hipLaunchKernelGGL(k_scan_add, dim3(ntiles), dim3(SCAN_T), 0, st, out, tile_sums, n);
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}

// Temporaries of one build phase.  Requests are carved out of a cached, grow-only device scratch
// block (phl_scratch_*, phl_api.hip) so that a warm build performs no hipMalloc/hipFree for its
// ~25 work arrays; whatever does not fit falls back to hipMalloc, and the block is re-sized to the
// phase's total on release so the next build fits.
struct temp_pool {
    std::vector<void *> ptrs;
    char *base = nullptr;
    size_t cap = 0, off = 0, wanted = 0;
    bool have_cache = false;
    temp_pool() { have_cache = phl_scratch_acquire((void **)&base, &cap); }
    ~temp_pool()
    {
        for (void *p : ptrs) (void)hipFree(p);
        if (have_cache) phl_scratch_release(wanted);
    }
    template <typename T>
    hipError_t get(T **out, size_t count)
    {
        const size_t bytes = (((count ? count : 1) * sizeof(T)) + 255) & ~(size_t)255;
        wanted += bytes;
        if (have_cache && off + bytes <= cap) {
            *out = (T *)(base + off);
            off += bytes;
            return hipSuccess;
        }
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) ptrs.push_back(p);
        *out = (T *)p;
        return e;
    }
};

// ------------------------------------------------------------------------------------------
// Stable sort of n int32 keys in [0, key_bound): returns the permutation (perm[i] = original index of the
// i-th smallest key; equal keys keep their original order).  LSD radix sort, 8 bits per pass, O(n) work for
// ANY key distribution -- the per-vertex / per-cell lists built from it have no length limit (a constant
// feature tensor puts every pixel into one list).  Per pass: block histograms [digit][block] -> one exclusive
// scan -> stable scatter (a block walks its 2048 keys in rounds of 256; inside a round a lane's rank among
// the lanes of its wavefront with the same digit comes from eight ballots).
constexpr int RS_TILE = 2048;

__attribute__((unused)) __global__ __launch_bounds__(256) void k_radix_hist(const int *__restrict__ keys, int n, int shift, int nblocks,
                                                    int *__restrict__ hist)
{
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * RS_TILE;
    for (int i = threadIdx.x; i < RS_TILE; i += 256) {
        const int e = base + i;
        if (e < n) atomicAdd(&h[(keys[e] >> shift) & 255], 1);
    }
    __syncthreads();
    hist[threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

__attribute__((unused)) __global__ __launch_bounds__(256) void k_radix_scatter(const int *__restrict__ keys, const int *__restrict__ idx, int n,
                                                       int shift, int nblocks, const int *__restrict__ base,
                                                       const int *__restrict__ tile_sums, int ntiles,
                                                       int *__restrict__ keys_out, int *__restrict__ idx_out)
{
    __shared__ int run[256];        // next output position of each digit for this block
    __shared__ int wcnt[4][256];    // digit counts of each wavefront in the current round
    __shared__ int pre[SCAN_FUSE_TILES];
    // tile_sums given: `base` is the tile-local scan of k_scan_tile, the tiles in front are added here (no k_scan_sums /
    // k_scan_add launch in a pass)
    const int g = threadIdx.x * nblocks + blockIdx.x;
    if (tile_sums) {
        tile_prefix_to_lds(tile_sums, ntiles, pre);
        run[threadIdx.x] = base[g] + pre[g / SCAN_TILE];
    } else {
        run[threadIdx.x] = base[g];
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = 0; r < RS_TILE; r += 256) {
        for (int j = threadIdx.x; j < 4 * 256; j += 256) (&wcnt[0][0])[j] = 0;
        __syncthreads();
        const int e = blockIdx.x * RS_TILE + r + threadIdx.x;
        const bool act = e < n;
        const int key = act ? keys[e] : 0;
        const int dg = (key >> shift) & 255;
        unsigned long long peers = __ballot(act);       // active lanes of my wavefront with my digit
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((dg >> b) & 1);
            peers &= ((dg >> b) & 1) ? m : ~m;
        }
        const int rank = __popcll(peers & ((1ull << lane) - 1ull));
        if (act && rank == 0) wcnt[w][dg] = __popcll(peers);
        __syncthreads();
        int pos = 0;
        if (act) {
            pos = run[dg] + rank;
            for (int k = 0; k < w; k++) pos += wcnt[k][dg];
        }
        __syncthreads();
        run[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
        if (act) {
            keys_out[pos] = key;
            idx_out[pos] = idx ? idx[e] : e;
        }
        __syncthreads();
    }
}

__attribute__((unused)) __global__ __launch_bounds__(256) void k_iota(int *p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

// Requests served from a caller's arena first (device bytes out of the caller's temp_pool, handed across translation
// units as a plain pointer), from a pool of its own when that is missing or full -- which means hipMalloc / hipFree
// while the caller holds the scratch block.
struct arena_pool {
    char *base;
    size_t cap, off = 0;
    temp_pool fallback;
    arena_pool(void *b, size_t c) : base((char *)b), cap(b ? c : 0) {}
    template <typename T>
    hipError_t get(T **out, size_t count)
    {
        const size_t bytes = (((count ? count : 1) * sizeof(T)) + 255) & ~(size_t)255;
        if (base && off + bytes <= cap) {
            *out = (T *)(base + off);
            off += bytes;
            return hipSuccess;
        }
        return fallback.get(out, count);
    }
};

template <typename Pool>
__attribute__((unused)) int stable_sort_perm(const int *keys, int n, int64_t key_bound, int *perm_out, Pool &tmp, hipStream_t st)
{
    if (n <= 0) return PHL_OK;
    int passes = 0;
    for (int64_t b = key_bound > 1 ? key_bound - 1 : 0; b > 0; b >>= 8) passes++;
    if (passes == 0) {
        hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, st, perm_out, n);
        PHL_HIP(hipGetLastError());
        return PHL_OK;
    }
    const int nblocks = (n + RS_TILE - 1) / RS_TILE;
    int *hist, *base, *tile_sums, *kA, *kB, *iA;
    PHL_HIP(tmp.get(&hist, (size_t)256 * nblocks));
    PHL_HIP(tmp.get(&base, (size_t)256 * nblocks + 1));
    PHL_HIP(tmp.get(&tile_sums, (size_t)256 * nblocks / SCAN_TILE + 2));
    PHL_HIP(tmp.get(&kA, (size_t)n));
    PHL_HIP(tmp.get(&kB, (size_t)n));
    PHL_HIP(tmp.get(&iA, (size_t)n));
    const int *kin = keys, *iin = nullptr;
    for (int p = 0; p < passes; p++) {
        int *kout = (p & 1) ? kB : kA;
        int *iout = ((passes - 1 - p) & 1) ? iA : perm_out;     // the last pass lands in perm_out
        hipLaunchKernelGGL(k_radix_hist, dim3(nblocks), dim3(256), 0, st, kin, n, 8 * p, nblocks, hist);
        PHL_HIP(hipGetLastError());
        const int nh = 256 * nblocks, ntiles = (nh + SCAN_TILE - 1) / SCAN_TILE;
        if (ntiles <= SCAN_FUSE_TILES) {        // three launches a pass: histogram, tile-local scan, scatter
            hipLaunchKernelGGL(k_scan_tile, dim3(ntiles), dim3(SCAN_T), 0, st, (const int *)hist, base, tile_sums, nh);
            hipLaunchKernelGGL(k_radix_scatter, dim3(nblocks), dim3(256), 0, st, kin, iin, n, 8 * p, nblocks, (const int *)base,
                               (const int *)tile_sums, ntiles, kout, iout);
        } else {
            const int rc = exclusive_scan(hist, base, nh, tile_sums, st);
            if (rc) return rc;
            hipLaunchKernelGGL(k_radix_scatter, dim3(nblocks), dim3(256), 0, st, kin, iin, n, 8 * p, nblocks, (const int *)base,
                               (const int *)nullptr, 0, kout, iout);
        }
        PHL_HIP(hipGetLastError());
        kin = kout;
        iin = iout;
    }
    return PHL_OK;
}

}  // namespace
