// Device utilities shared by the phl translation units (each TU gets its own copy: everything
// here lives in an anonymous namespace).
#pragma once
#include <vector>

#include "phl_internal.h"

namespace {

// One wave per vertex: rank-sort its contribution list by pixel index (pixels are distinct
// inside a list because the d+1 vertices of one pixel's simplex are distinct).
__global__ __launch_bounds__(256) void k_sort_lists(const phl_contrib_t *__restrict__ tmp, const int *__restrict__ ptr,
                                                    int M, phl_contrib_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int v = wave; v < M; v += nwaves) {
        const int beg = ptr[v], k = ptr[v + 1] - beg;
        for (int i0 = 0; i0 < k; i0 += 64) {
            const bool valid = (i0 + lane) < k;
            phl_contrib_t mine;
            mine.pixel = 0x7FFFFFFF;
            mine.w = 0.f;
            if (valid) mine = tmp[beg + i0 + lane];
            int rank = 0;
            for (int j0 = 0; j0 < k; j0 += 64) {
                const int other = (j0 + lane) < k ? tmp[beg + j0 + lane].pixel : 0x7FFFFFFF;
                const int cnt = min(64, k - j0);
                for (int t = 0; t < cnt; t++) rank += (__shfl(other, t) < mine.pixel) ? 1 : 0;
            }
            if (valid) out[beg + rank] = mine;
        }
    }
}

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

int exclusive_scan(const int *in, int *out /* n+1 */, int n, int *tile_sums, hipStream_t st)
{
    if (n <= 0) {
        PHL_HIP(hipMemsetAsync(out, 0, sizeof(int), st));
        return PHL_OK;
    }
    const int ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_tile, dim3(ntiles), dim3(SCAN_T), 0, st, in, out, tile_sums, n);
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

}  // namespace
