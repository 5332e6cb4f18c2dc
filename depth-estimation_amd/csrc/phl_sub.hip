// phl_sub.hip -- a row band's lattice CUT OUT OF the whole image's lattice (phl_sub_lattice).
//
// Row-band multi-GPU runs used to build one defect-free lattice per band from the band's own pixels.  The reference's
// hash table, however, files the key in flight at each of its doublings from a stale slot (permutohedral.h:59-62,
// 101-103): a handful of keys get a second vertex, and which of the two a (pixel, remainder) lookup resolves to depends
// on the insertion history of the WHOLE image.  A band that is to return the reference's results therefore takes its
// vertices from the whole image's reference-table lattice (which every rank can build: 1.4 ms at C3) instead of
// re-deriving them:
//   * its pixels' replay entries are the global ones, re-indexed;
//   * its vertex set is a caller-chosen selection of global vertices -- the ones its own pixels touch first (`n_own`
//     of them), then the ghosts of the neighbouring bands in the caller's order -- duplicates of a key included, with the
//     global lattice's hidden flags (the vertices blur's neighbour lookups cannot see);
//   * neighbour tables, chunk structures and the locality numbering of the OWN vertices are built the usual way
//     (phl_tiles_build); ghost rows keep the caller's order behind them.
// No reference counterpart (the reference is single-process).
#include <vector>

#include "phl_device_utils.h"

namespace {
__global__ __launch_bounds__(256) void k_mark_rows_of_pixels(const phl_replay_t *__restrict__ replay, int64_t e0, int64_t e1,
                                                             const int *__restrict__ ft_of_int, unsigned char *__restrict__ mask_ft)
{
    const int64_t e = e0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= e1) return;
    const int row = replay[e].vid;
    mask_ft[ft_of_int ? ft_of_int[row] : row] = 1;
}

__global__ __launch_bounds__(256) void k_sub_select(const int *__restrict__ sel_ft, int K, const int *__restrict__ int_of_ft,
                                                    const int16_t *__restrict__ vkeys_g, int d, int *__restrict__ g2l,
                                                    int16_t *__restrict__ vkeys_l)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const int row = int_of_ft ? int_of_ft[sel_ft[i]] : sel_ft[i];
    g2l[row] = i;
    for (int c = 0; c < d; c++) vkeys_l[(int64_t)i * d + c] = vkeys_g[(int64_t)row * d + c];
}

__global__ __launch_bounds__(256) void k_sub_replay(const phl_replay_t *__restrict__ rg, int64_t e0, int N, const int *__restrict__ g2l,
                                                    int n_own, phl_replay_t *__restrict__ rl, int *__restrict__ err)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const phl_replay_t r = rg[e0 + e];
    const int v = g2l[r.vid];
    if (v < 0 || v >= n_own) atomicOr(err, 1);      // a pixel of the band touches a vertex that is not among its own
    phl_replay_t o;
    o.vid = v < 0 ? 0 : v;
    o.w = r.w;
    rl[e] = o;
}

struct device_sel {
    int prev = -1;
    bool ok = false;
    explicit device_sel(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (prev == dev) || (hipSetDevice(dev) == hipSuccess);
    }
    ~device_sel()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
}  // namespace

extern "C" {

int phl_vertices_of_pixels(phl_lattice *lat, int64_t p0, int64_t p1, unsigned char *mask_host, phl_stream stream)
{
    if (!lat || !mask_host || p0 < 0 || p1 < p0 || p1 > lat->n) { phl_set_error("phl_vertices_of_pixels: bad arguments"); return PHL_ERR_INVALID; }
    const int64_t M = lat->M;
    if (M == 0) return PHL_OK;
    device_sel g(lat->device);
    hipStream_t st = (hipStream_t)stream;
    temp_pool tmp;
    unsigned char *mask;
    PHL_HIP(tmp.get(&mask, (size_t)M));
    PHL_HIP(hipMemsetAsync(mask, 0, (size_t)M, st));
    const int dp1 = lat->d + 1;
    const int64_t e0 = p0 * dp1, e1 = p1 * dp1;
    if (e1 > e0) {
        hipLaunchKernelGGL(k_mark_rows_of_pixels, dim3((unsigned)((e1 - e0 + 255) / 256)), dim3(256), 0, st, lat->replay, e0, e1,
                           lat->ft_of_int, mask);
        PHL_HIP(hipGetLastError());
    }
    PHL_HIP(hipMemcpyAsync(mask_host, mask, (size_t)M, hipMemcpyDeviceToHost, st));
    PHL_HIP(hipStreamSynchronize(st));
    return PHL_OK;
}

int phl_sub_lattice(phl_lattice **out, phl_lattice *g, int64_t p0, int64_t p1, const int32_t *sel_host, int64_t n_sel, int64_t n_own,
                    const float *ref_dev, int64_t rs, int64_t cs, phl_stream stream)
{
    if (!out) { phl_set_error("phl_sub_lattice: out is NULL"); return PHL_ERR_INVALID; }
    *out = nullptr;
    if (!g || p0 < 0 || p1 <= p0 || p1 > g->n || !sel_host || n_own < 0 || n_sel < n_own || n_sel > g->M || !ref_dev) {
        phl_set_error("phl_sub_lattice: bad arguments");
        return PHL_ERR_INVALID;
    }
    if (g->nbr00_override != -2) {
        phl_set_error("phl_sub_lattice: the whole image's table doubles inside blur() (M = 2^k - 1 exactly): not carried over to bands");
        return PHL_ERR_UNSUPPORTED;
    }
    {   // the selection must name distinct vertices of the whole lattice (a repeated id would silently drop a row)
        std::vector<unsigned char> seen((size_t)g->M, 0);
        for (int64_t i = 0; i < n_sel; i++) {
            const int32_t v = sel_host[i];
            if (v < 0 || v >= g->M || seen[(size_t)v]) {
                phl_set_error("phl_sub_lattice: selection entry %lld = %d is out of range or repeated", (long long)i, (int)v);
                return PHL_ERR_INVALID;
            }
            seen[(size_t)v] = 1;
        }
    }
    device_sel guard(g->device);
    hipStream_t st = (hipStream_t)stream;
    const int d = g->d, dp1 = d + 1;
    const int64_t n = p1 - p0, N = n * dp1;
    const int K = (int)n_sel;

    phl_lattice *lat = nullptr;
    int rc = phl_lattice_blank(&lat, g->device, d, n);
    if (rc) return rc;
    lat->N = N;
    lat->M = K;
    lat->M_local = n_own;
    lat->build_flags = g->build_flags;
    auto fail = [&](int code) {
        (void)hipStreamSynchronize(st);
        phl_destroy(lat);
        return code;
    };
#define SUB_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e__ = (call);                                                       \
        if (e__ != hipSuccess) return fail(phl_hip_fail(e__, #call, __FILE__, __LINE__)); \
    } while (0)
    {
        temp_pool tmp;
        int *sel_dev, *g2l, *err;
        SUB_HIP(tmp.get(&sel_dev, (size_t)K + 1));
        SUB_HIP(tmp.get(&g2l, (size_t)g->M + 1));
        SUB_HIP(tmp.get(&err, 1));
        SUB_HIP(hipMemsetAsync(err, 0, sizeof(int), st));
        SUB_HIP(hipMemsetAsync(g2l, 0xFF, sizeof(int) * (size_t)g->M, st));       // -1
        SUB_HIP(hipMemcpyAsync(sel_dev, sel_host, sizeof(int) * (size_t)K, hipMemcpyHostToDevice, st));
        SUB_HIP(phl_dev_malloc((void **)&lat->vkeys, sizeof(int16_t) * (size_t)(K ? K : 1) * d));
        SUB_HIP(phl_dev_malloc((void **)&lat->replay, sizeof(phl_replay_t) * (size_t)N));
        if (K > 0)
            hipLaunchKernelGGL(k_sub_select, dim3((K + 255) / 256), dim3(256), 0, st, sel_dev, K, g->int_of_ft, g->vkeys, d, g2l, lat->vkeys);
        hipLaunchKernelGGL(k_sub_replay, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, g->replay, p0 * dp1, (int)N, g2l, (int)n_own,
                           lat->replay, err);
        SUB_HIP(hipGetLastError());
        int err_host = 0;
        SUB_HIP(hipMemcpyAsync(&err_host, err, sizeof(int), hipMemcpyDeviceToHost, st));
        SUB_HIP(hipStreamSynchronize(st));
        if (err_host) {
            phl_set_error("phl_sub_lattice: a pixel of [%lld, %lld) touches a vertex outside the first %lld selected ones", (long long)p0,
                          (long long)p1, (long long)n_own);
            return fail(PHL_ERR_INVALID);
        }
    }
    // hidden vertices (first-touch ids of the whole lattice) that were selected, in the band's own first-touch numbering
    // (= position in the selection)
    lat->n_hidden = 0;
    for (int h = 0; h < g->n_hidden; h++)
        for (int64_t i = 0; i < n_sel; i++)
            if (sel_host[i] == g->hidden[h]) {
                if (lat->n_hidden < PHL_MAX_HIDDEN) lat->hidden[lat->n_hidden++] = (int32_t)i;
                break;
            }
    rc = phl_tiles_build(lat, ref_dev, rs, cs, st);
    if (rc) return fail(rc);
    *out = lat;
    return PHL_OK;
#undef SUB_HIP
}

}  // extern "C"
