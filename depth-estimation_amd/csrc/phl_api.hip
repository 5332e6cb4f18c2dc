// phl_api.hip -- host side of the C ABI declared in include/phl.h.
//
// No CPU fallback lives here on purpose: without a HIP device every entry point that would
// compute returns PHL_ERR_NO_DEVICE / PHL_ERR_HIP.  The CPU restatement under oracle/ is test
// infrastructure and is never linked into this library.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <chrono>
#include <mutex>
#include <utility>
#include <vector>

#include "phl_internal.h"

namespace {
thread_local char g_err[512] = "";

struct device_guard {
    int prev = -1;
    bool ok = false;
    explicit device_guard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (prev == dev) || (hipSetDevice(dev) == hipSuccess);
    }
    ~device_guard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int grow(float **p, int64_t *cap, int64_t need)
{
    if (need <= *cap) return PHL_OK;
    if (*p) {
        PHL_HIP(hipDeviceSynchronize());      // (rare) the old buffer may still be read by launches in flight
        PHL_HIP(phl_dev_free(*p));
    }
    *p = nullptr;
    *cap = 0;
    PHL_HIP(phl_dev_malloc((void **)p, sizeof(float) * (size_t)need));
    *cap = need;
    return PHL_OK;
}
}  // namespace

// ---- device block cache --------------------------------------------------------------------------------
// hipMalloc / hipFree cost 50-300 us each and hipFree synchronises the device; a lattice owns ~20 device arrays,
// so the reference-shaped call (build, filter, destroy per call: phl_filter_once) spent as long in the allocator
// as in its kernels.  Blocks released by a lattice are kept, by device and size class (sizes rounded up to 1/8 of
// their power of two, so that the arrays of a slightly different image fit the same blocks), and handed to the
// next build.  A caller of phl_dev_free guarantees that no launch still uses the block, or that its next user
// is ordered behind it (phl_destroy synchronises the device first, as hipFree itself would).
// PHL_CACHE_MAX_MB (default 8192, 0 = off) caps the cached bytes; phl_trim_scratch() releases everything.
namespace {
struct dev_cache_t {
    std::mutex mu;
    std::map<std::pair<int, size_t>, std::vector<void *>> free_blocks;
    std::map<void *, std::pair<int, size_t>> live;      // every block handed out: device, class size
    size_t cached = 0;
};
dev_cache_t &dev_cache()
{
    static dev_cache_t *c = new dev_cache_t();          // never destroyed: the HIP runtime may be gone at exit
    return *c;
}
size_t cache_limit()
{
    static const size_t lim = [] {
        const char *e = getenv("PHL_CACHE_MAX_MB");
        return (size_t)(e ? atoll(e) : 8192) << 20;
    }();
    return lim;
}
size_t size_class(size_t bytes)
{
    if (bytes < 4096) return 4096;
    size_t p2 = 1;
    while (p2 * 2 <= bytes) p2 *= 2;
    const size_t step = p2 / 8;
    return (bytes + step - 1) / step * step;
}
void cache_flush_locked(dev_cache_t &c)
{
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (auto &kv : c.free_blocks) {
        (void)hipSetDevice(kv.first.first);
        for (void *p : kv.second) (void)hipFree(p);
    }
    c.free_blocks.clear();
    c.cached = 0;
    if (prev >= 0) (void)hipSetDevice(prev);
}
}  // namespace

hipError_t phl_dev_malloc(void **p, size_t bytes)
{
    dev_cache_t &c = dev_cache();
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const size_t cls = size_class(bytes);
    {
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.free_blocks.find({dev, cls});
        if (it != c.free_blocks.end() && !it->second.empty()) {
            *p = it->second.back();
            it->second.pop_back();
            c.cached -= cls;
            c.live[*p] = {dev, cls};
            return hipSuccess;
        }
    }
    e = hipMalloc(p, cls);
    if (e != hipSuccess) {                               // out of memory: give the cached blocks back and retry once
        (void)hipGetLastError();
        std::lock_guard<std::mutex> lk(c.mu);
        cache_flush_locked(c);
        e = hipMalloc(p, cls);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lk(c.mu);
    c.live[*p] = {dev, cls};
    return hipSuccess;
}

hipError_t phl_dev_free(void *p)
{
    if (!p) return hipSuccess;
    dev_cache_t &c = dev_cache();
    std::lock_guard<std::mutex> lk(c.mu);
    auto it = c.live.find(p);
    if (it == c.live.end()) return hipFree(p);           // not ours
    const std::pair<int, size_t> key = it->second;
    c.live.erase(it);
    if (cache_limit() == 0 || key.second > cache_limit()) return hipFree(p);
    if (c.cached + key.second > cache_limit()) cache_flush_locked(c);
    c.free_blocks[key].push_back(p);
    c.cached += key.second;
    return hipSuccess;
}

// ---- value workspaces --------------------------------------------------------------------------------
// SURVEY 8(b), threading row: "re-entrant; autograd may call from any thread".  A lattice's tables are
// read-only after the build; what a filter call WRITES (the [M][vd] ping-pong pair, partial rows, staging
// copies) lives in a workspace the call takes for the duration of its launches.  A workspace is handed out
// again when (a) it has never carried work, (b) the caller enqueues on the stream that used it last (stream
// order protects it), or (c) the event recorded behind its last launch has completed.  Otherwise a new one is
// allocated: two threads on two streams run concurrently on separate buffers.  A workspace whose launches were
// captured into a HIP graph stays bound to that stream and is never resized (the graph holds its pointers).
struct phl_shared {
    std::mutex mu;                       // guards `ws` and the workspaces' state flags
    std::vector<phl_workspace *> ws;

    std::mutex csr_mu;                   // one-time build of the pixel-sorted lists (phl_ensure_csr)
};

namespace {
void ws_free_buffers(phl_workspace *w)
{
    void *ptrs[] = {w->buf[0], w->buf[1], w->partial, w->stage_in, w->stage_out};
    for (void *p : ptrs)
        if (p) (void)phl_dev_free(p);
    w->buf[0] = w->buf[1] = w->partial = w->stage_in = w->stage_out = nullptr;
    w->buf_elems = w->partial_elems = w->stage_elems = 0;
}

// (no fence needed: the launches it stands behind finished before it started -- stream order -- and the store
//  itself is released to the host at the kernel's end; tools/mark_probe.hip: 2.7 us in-stream, against 3.5 us with
//  a system fence and 4.4 us for hipStreamWriteValue64)
__global__ void k_mark_done(unsigned long long *word, unsigned long long ticket)
{
    *reinterpret_cast<volatile unsigned long long *>(word) = ticket;
}

bool ws_drained(const phl_workspace *w)
{
    return w->done_word && *reinterpret_cast<volatile unsigned long long *>(w->done_word) == w->ticket;
}

bool is_capturing(hipStream_t st)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return cs != hipStreamCaptureStatusNone;
}
}  // namespace

std::mutex *phl_csr_mutex(phl_lattice *lat) { return &lat->shared->csr_mu; }

int phl_ws_acquire(phl_lattice *lat, hipStream_t st, int64_t buf_elems, int64_t partial_elems, int64_t stage_elems,
                   phl_workspace **out)
{
    phl_shared *sh = lat->shared;
    phl_workspace *w = nullptr;
    {
        std::lock_guard<std::mutex> lk(sh->mu);
        phl_workspace *fits = nullptr, *any = nullptr;
        for (phl_workspace *c : sh->ws) {
            if (c->in_enqueue) continue;
            if (c->stream_bound && c->last_stream != st) {
                if (c->captured || !ws_drained(c)) continue;
                c->stream_bound = false;       // its work has drained
            }
            const bool big = c->buf_elems >= buf_elems && c->partial_elems >= partial_elems && c->stage_elems >= stage_elems;
            if (big && !fits) fits = c;
            if (!c->captured && !any) any = c;  // may be grown
        }
        w = fits ? fits : any;
        if (!w) {
            w = new phl_workspace();
            memset(w, 0, sizeof(*w));
            // pinned, device-visible host word for the completion mark (cannot be allocated during a capture:
            // such a workspace then simply stays bound to its stream)
            void *hw = nullptr;
            if (hipHostMalloc(&hw, sizeof(unsigned long long), hipHostMallocMapped) == hipSuccess) {
                w->done_word = (unsigned long long *)hw;
                *w->done_word = 0;
            } else {
                (void)hipGetLastError();
            }
            sh->ws.push_back(w);
        }
        w->in_enqueue = true;
    }
    int rc = PHL_OK;
    if (buf_elems > w->buf_elems) {
        int64_t c0 = w->buf_elems, c1 = w->buf_elems;
        rc = grow(&w->buf[0], &c0, buf_elems);
        if (!rc) rc = grow(&w->buf[1], &c1, buf_elems);
        w->buf_elems = rc ? 0 : buf_elems;
    }
    if (!rc && partial_elems > w->partial_elems) {
        rc = grow(&w->partial, &w->partial_elems, partial_elems);
    }
    if (!rc && stage_elems > w->stage_elems) {
        int64_t c0 = w->stage_elems, c1 = w->stage_elems;
        rc = grow(&w->stage_in, &c0, stage_elems);
        if (!rc) rc = grow(&w->stage_out, &c1, stage_elems);
        w->stage_elems = rc ? 0 : stage_elems;
    }
    if (rc) {
        std::lock_guard<std::mutex> lk(sh->mu);
        w->in_enqueue = false;
        return rc;
    }
    *out = w;
    return PHL_OK;
}

void phl_ws_release(phl_lattice *lat, phl_workspace *w, hipStream_t st, bool idle)
{
    if (!idle) {
        // a captured call binds its workspace to the stream for good (the graph keeps the pointers), so nobody will
        // ever ask whether its work has drained: no completion mark inside a graph (one launch of seven saved per
        // replayed Tsukuba-sized filter, where launches are what is being paid for)
        const bool capturing = is_capturing(st);
        if (capturing) w->captured = true;
        w->ticket++;
        if (w->done_word && !capturing) {
            hipLaunchKernelGGL(k_mark_done, dim3(1), dim3(1), 0, st, w->done_word, w->ticket);
            if (hipGetLastError() != hipSuccess) w->ticket += 1ull << 32;   // never reads as drained
        }
    }
    std::lock_guard<std::mutex> lk(lat->shared->mu);
    if (!idle) {
        w->last_stream = st;
        w->stream_bound = true;
    }
    w->in_enqueue = false;
}

// ---- cached scratch block for build temporaries -------------------------------------------------
namespace {
std::mutex g_scratch_mu;
void *g_scratch = nullptr;
size_t g_scratch_cap = 0;
int g_scratch_dev = -1;
bool g_scratch_busy = false;
}  // namespace

namespace {
struct pinned_arena {
    char *base = nullptr;
    size_t cap = 0, off = 0, wanted = 0;
};
thread_local pinned_arena t_pinned;     // (never freed: a thread's exit may come after the runtime has shut down;
                                        //  portable: one thread may build on several devices in turn)
}  // namespace

void phl_pinned_reset()
{
    pinned_arena &a = t_pinned;
    if (a.wanted > a.cap || !a.base) {
        size_t want = a.wanted > ((size_t)1 << 20) ? a.wanted + (a.wanted >> 2) : ((size_t)1 << 20);
        if (want <= ((size_t)256 << 20)) {
            if (a.base) (void)hipHostFree(a.base);
            a.base = nullptr;
            a.cap = 0;
            void *p = nullptr;
            if (hipHostMalloc(&p, want, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess) {
                a.base = (char *)p;
                a.cap = want;
            } else {
                (void)hipGetLastError();
            }
        }
    }
    a.off = 0;
    a.wanted = 0;
}

void *phl_pinned_alloc(size_t bytes)
{
    pinned_arena &a = t_pinned;
    bytes = (bytes + 255) & ~(size_t)255;
    a.wanted += bytes;
    if (!a.base || a.off + bytes > a.cap) return nullptr;
    void *p = a.base + a.off;
    a.off += bytes;
    return p;
}

bool phl_scratch_acquire(void **base, size_t *cap)
{
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_busy) return false;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (g_scratch && dev != g_scratch_dev) {   // lattices are being built on another device now
        int prev = dev;
        (void)hipSetDevice(g_scratch_dev);
        (void)hipFree(g_scratch);
        (void)hipSetDevice(prev);
        g_scratch = nullptr;
        g_scratch_cap = 0;
    }
    g_scratch_dev = dev;
    g_scratch_busy = true;
    *base = g_scratch;
    *cap = g_scratch_cap;
    return true;
}

void phl_scratch_release(size_t wanted)
{
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    g_scratch_busy = false;
    static const size_t limit = [] {
        const char *e = getenv("PHL_SCRATCH_MAX_MB");   // 0 disables the cache
        return (size_t)(e ? atoll(e) : 4096) << 20;
    }();
    if (wanted > g_scratch_cap && wanted <= limit) {
        if (g_scratch) (void)hipFree(g_scratch);
        g_scratch = nullptr;
        g_scratch_cap = 0;
        const size_t want = wanted + (wanted >> 3);
        if (hipMalloc(&g_scratch, want) == hipSuccess) g_scratch_cap = want;
        else (void)hipGetLastError();
    }
}

void phl_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int phl_hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    const char *base = strrchr(file, '/');
    phl_set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, base ? base + 1 : file, line);
    (void)hipGetLastError();  // clear sticky state
    return e == hipErrorNoDevice || e == hipErrorInvalidDevice ? PHL_ERR_NO_DEVICE : PHL_ERR_HIP;
}

// an empty handle (phl_sub_lattice fills it): same initial state phl_build_ex gives one
int phl_lattice_blank(phl_lattice **out, int device, int d, int64_t n)
{
    phl_lattice *lat = new phl_lattice();
    memset(lat, 0, sizeof(*lat));
    lat->device = device;
    lat->d = d;
    lat->n = n;
    lat->nbr00_override = -2;
    lat->shared = new phl_shared();
    *out = lat;
    return PHL_OK;
}

extern "C" {

int phl_version(void) { return PHL_VERSION; }
const char *phl_last_error(void) { return g_err; }

const char *phl_status_string(int s)
{
    switch (s) {
        case PHL_OK: return "ok";
        case PHL_ERR_INVALID: return "invalid argument";
        case PHL_ERR_SHAPE: return "incompatible shapes";
        case PHL_ERR_HIP: return "HIP runtime error";
        case PHL_ERR_NO_DEVICE: return "no HIP device";
        case PHL_ERR_KEY_RANGE: return "lattice key outside int16";
        case PHL_ERR_TOO_LARGE: return "problem too large for int32 indexing";
        case PHL_ERR_UNSUPPORTED: return "unsupported";
        default: return "unknown";
    }
}

int phl_trim_scratch(void)
{
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_busy) return PHL_OK;
    if (g_scratch) {
        int prev = -1;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(g_scratch_dev);
        (void)hipFree(g_scratch);
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    g_scratch = nullptr;
    g_scratch_cap = 0;
    {
        dev_cache_t &c = dev_cache();
        std::lock_guard<std::mutex> lk2(c.mu);
        cache_flush_locked(c);
    }
    return PHL_OK;
}

int phl_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int phl_build(phl_lattice **out, const float *ref_dev, int64_t n, int d, int64_t rs, int64_t cs, int device,
              phl_stream stream)
{
    return phl_build_ex(out, ref_dev, n, d, rs, cs, device, stream, PHL_BUILD_DEFAULT);
}

int phl_build_ex(phl_lattice **out, const float *ref_dev, int64_t n, int d, int64_t rs, int64_t cs, int device,
                 phl_stream stream, unsigned build_flags)
{
    if (!out) { phl_set_error("phl_build: out is NULL"); return PHL_ERR_INVALID; }
    *out = nullptr;
    if (n < 0 || d < 1 || (n > 0 && !ref_dev)) { phl_set_error("phl_build: bad arguments (n=%lld d=%d)", (long long)n, d); return PHL_ERR_INVALID; }
    if (d > PHL_MAX_D) { phl_set_error("phl_build: d=%d exceeds PHL_MAX_D=%d", d, PHL_MAX_D); return PHL_ERR_UNSUPPORTED; }
    if (n * (int64_t)(d + 1) > (int64_t)1 << 30) { phl_set_error("phl_build: n*(d+1)=%lld exceeds 2^30", (long long)(n * (d + 1))); return PHL_ERR_TOO_LARGE; }
    if (phl_device_count() <= device || device < 0) { phl_set_error("phl_build: HIP device %d not available (no CPU fallback)", device); return PHL_ERR_NO_DEVICE; }
    device_guard g(device);
    if (!g.ok) { phl_set_error("phl_build: cannot select device %d", device); return PHL_ERR_NO_DEVICE; }

    phl_lattice *lat = new phl_lattice();
    memset(lat, 0, sizeof(*lat));
    lat->device = device;
    lat->d = d;
    lat->n = n;
    lat->build_flags = build_flags;
    lat->nbr00_override = -2;
    lat->shared = new phl_shared();
    static const bool dbg_t = getenv("PHL_DEBUG") != nullptr;
    const auto tb0 = std::chrono::steady_clock::now();
    int rc = phl_build_device(lat, ref_dev, rs, cs, (hipStream_t)stream);
    const auto tb1 = std::chrono::steady_clock::now();
    if (rc == PHL_OK) rc = phl_tiles_build(lat, ref_dev, rs, cs, (hipStream_t)stream);
    if (dbg_t)
        fprintf(stderr, "[phl] build: vertices %.2f ms, chunks %.2f ms\n", std::chrono::duration<double, std::milli>(tb1 - tb0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb1).count());
    if (rc != PHL_OK) {
        phl_destroy(lat);
        return rc;
    }
    *out = lat;
    return PHL_OK;
}

int phl_destroy(phl_lattice *lat)
{
    if (!lat) return PHL_OK;
    device_guard g(lat->device);
    (void)hipDeviceSynchronize();      // what hipFree would do: nothing may still be using the arrays (they are cached, not freed)
    phl_tiles_free(lat);
    phl_release_build_tables(lat);                 // (only a build that failed half-way leaves any)
    void *ptrs[] = {lat->vkeys, lat->replay, lat->csr_ptr, lat->csr, lat->nbr, lat->nbr2, lat->table, lat->ft_of_int, lat->int_of_ft, lat->vfirst};
    for (void *p : ptrs)
        if (p) (void)phl_dev_free(p);
    if (lat->shared) {
        for (phl_workspace *w : lat->shared->ws) {
            ws_free_buffers(w);
            if (w->done_word) (void)hipHostFree(w->done_word);
            delete w;
        }
        delete lat->shared;
    }
    delete lat;
    return PHL_OK;
}

int64_t phl_num_pixels(const phl_lattice *lat) { return lat ? lat->n : -1; }
int64_t phl_num_vertices(const phl_lattice *lat) { return lat ? lat->M : -1; }
int phl_num_dims(const phl_lattice *lat) { return lat ? lat->d : -1; }
int phl_device(const phl_lattice *lat) { return lat ? lat->device : -1; }
int64_t phl_device_bytes(const phl_lattice *lat)
{
    if (!lat) return -1;
    int64_t b = lat->table_bytes + lat->tile_bytes;
    std::lock_guard<std::mutex> lk(lat->shared->mu);
    for (const phl_workspace *w : lat->shared->ws)
        b += (w->partial_elems + 2 * w->buf_elems + 2 * w->stage_elems) * (int64_t)sizeof(float);
    return b;
}

int64_t phl_num_local_vertices(const phl_lattice *lat) { return lat ? lat->M_local : -1; }

int phl_add_vertices(phl_lattice *lat, const int16_t *keys_host, int64_t count, int32_t *vid_host, phl_stream stream)
{
    if (!lat || count < 0 || (count > 0 && (!keys_host || !vid_host))) { phl_set_error("phl_add_vertices: bad arguments"); return PHL_ERR_INVALID; }
    if (lat->M + count > ((int64_t)1 << 30)) { phl_set_error("phl_add_vertices: too many vertices"); return PHL_ERR_TOO_LARGE; }
    device_guard g(lat->device);
    const int64_t M_before = lat->M;
    int rc = phl_add_vertices_device(lat, keys_host, count, vid_host, (hipStream_t)stream);
    lat->blur_rows_set = 0;            // row ranges are rows of the old vertex array
    if (rc == PHL_OK && lat->M != M_before) {
        rc = phl_tiles_link_vertices(lat, (hipStream_t)stream);
        // the value workspaces are sized by M and S_multi: drop them (not a concurrent operation: the lattice is
        // still being assembled)
        std::lock_guard<std::mutex> lk(lat->shared->mu);
        for (phl_workspace *w : lat->shared->ws) ws_free_buffers(w);
    }
    return rc;
}

int phl_set_blur_rows(phl_lattice *lat, const int64_t *ranges, int naxes)
{
    if (!lat) { phl_set_error("phl_set_blur_rows: bad arguments"); return PHL_ERR_INVALID; }
    if (!ranges || naxes == 0) {        // back to all rows on every axis
        lat->blur_rows_set = 0;
        return PHL_OK;
    }
    if (naxes != lat->d + 1) { phl_set_error("phl_set_blur_rows: %d axes given, the lattice has %d", naxes, lat->d + 1); return PHL_ERR_INVALID; }
    for (int a = 0; a < naxes; a++) {
        int64_t prev_end = 0;
        for (int k = 0; k < 3; k++) {
            const int64_t b = ranges[(a * 3 + k) * 2], e = ranges[(a * 3 + k) * 2 + 1];
            if (b < prev_end || e < b || e > lat->M) {
                phl_set_error("phl_set_blur_rows: axis %d range %d = [%lld, %lld) is not ascending / inside [0, %lld)", a, k, (long long)b,
                              (long long)e, (long long)lat->M);
                return PHL_ERR_INVALID;
            }
            prev_end = e;
        }
    }
    for (int a = 0; a < naxes; a++)
        for (int k = 0; k < 3; k++) {
            lat->blur_rows[a][k][0] = (int32_t)ranges[(a * 3 + k) * 2];
            lat->blur_rows[a][k][1] = (int32_t)ranges[(a * 3 + k) * 2 + 1];
        }
    lat->blur_rows_set = 1;
    return PHL_OK;
}

namespace {
bool use_tiled_splat(const phl_lattice *lat, int vd, unsigned flags, const void *a, const void *b, int64_t rs)
{
    if (flags & (PHL_FILTER_EXACT | PHL_FILTER_NO_TILES)) return false;
    if (phl_tiles_lprs(lat, vd, 0) < 0) return false;
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15 || rs % 4) return false;
    // Partial rows cost 2 * S_multi row transfers (written, re-read); staging saves d of the d+1 reads the gather
    // splat makes of every pixel row.  Break-even is far out (S_multi = d/2 * n); beyond 2n the chunks share so
    // little that the gather form is taken (the iid stress case sits at ~0.9n and is still 1.3x faster staged).
    return lat->S_multi <= 2 * lat->n;
}
bool use_tiled_slice(const phl_lattice *lat, int vd, unsigned flags, const void *a, const void *b, const void *c, int64_t rs,
                     int64_t rs2)
{
    if (flags & PHL_FILTER_NO_TILES) return false;
    if (phl_tiles_lprs(lat, vd, 1) < 0) return false;
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15 || rs % 4 || rs2 % 4) return false;
    return lat->S <= 3 * lat->n;
}
}  // namespace

int phl_tile_stats(const phl_lattice *lat, int vd, int64_t out[7])
{
    if (!lat || !out) { phl_set_error("phl_tile_stats: bad arguments"); return PHL_ERR_INVALID; }
    out[0] = lat->P; out[1] = lat->nchunks; out[2] = lat->nv_max; out[3] = lat->S; out[4] = lat->S_multi;
    out[5] = use_tiled_splat(lat, vd, 0, nullptr, nullptr, 0);
    out[6] = use_tiled_slice(lat, vd, 0, nullptr, nullptr, nullptr, 0, 0);
    return PHL_OK;
}

namespace {
// workspace sizes of a filter call over vd channels
int64_t need_partial(const phl_lattice *lat, int vd, unsigned flags)
{
    if (flags & (PHL_FILTER_EXACT | PHL_FILTER_NO_TILES)) return 0;
    if (phl_tiles_lprs(lat, vd, 0) < 0 || lat->S_multi > 2 * lat->n) return 0;
    return lat->S_multi * (int64_t)vd;
}
}  // namespace

int phl_reserve_ex(phl_lattice *lat, int vd, unsigned reserve_flags)
{
    if (!lat || vd < 0) { phl_set_error("phl_reserve: bad arguments"); return PHL_ERR_INVALID; }
    device_guard g(lat->device);
    const unsigned fflags = (reserve_flags & PHL_RESERVE_EXACT) ? PHL_FILTER_EXACT : 0;
    // what phl_filter will do with this width: channel counts >= 128 that are not a multiple of 4 run staged at the width
    // rounded up (both sides); rows off the 16-byte grid at a multiple of 4 cannot be known here -- PHL_RESERVE_STRIDED_IO
    // sizes the staging buffers for those as for channel-major views
    int vdw = vd;
    bool staged = (reserve_flags & PHL_RESERVE_STRIDED_IO) != 0;
    if (vd >= 128 && vd % 4 && !(fflags & PHL_FILTER_EXACT)) {
        vdw = (vd + 3) & ~3;
        staged = true;
    }
    const int64_t partial = need_partial(lat, vdw, fflags);
    // the gather splat (exact mode, or shapes the chunk splat does not take) needs its pixel-sorted lists
    const bool gather = partial == 0 && !use_tiled_splat(lat, vdw, fflags, nullptr, nullptr, 0);
    if (gather) {
        const int rc = phl_ensure_csr(lat, nullptr);
        if (rc) return rc;
    }
    phl_workspace *w = nullptr;
    const int64_t stage = staged ? lat->n * (int64_t)vdw : 0;
    int rc = phl_ws_acquire(lat, nullptr, lat->M * (int64_t)vdw, partial, stage, &w);
    if (rc) return rc;
    phl_ws_release(lat, w, nullptr, /*idle=*/true);      // sized, carries no work: the next call on any stream takes it
    return PHL_OK;
}

int phl_reserve(phl_lattice *lat, int vd) { return phl_reserve_ex(lat, vd, 0); }

int phl_splat(phl_lattice *lat, const float *src, int vd, int64_t src_rs, float *vert, unsigned flags, phl_stream st)
{
    if (!lat || vd < 0 || (lat->n > 0 && vd > 0 && (!src || !vert))) { phl_set_error("phl_splat: bad arguments"); return PHL_ERR_INVALID; }
    device_guard g(lat->device);
    if (use_tiled_splat(lat, vd, flags, src, vert, src_rs)) {
        phl_workspace *w = nullptr;       // only the partial rows come from the workspace here
        int rc = phl_ws_acquire(lat, (hipStream_t)st, 0, need_partial(lat, vd, flags), 0, &w);
        if (rc) return rc;
        rc = phl_launch_splat_tiled(lat, src, src_rs, vd, vert, w->partial, (hipStream_t)st);
        phl_ws_release(lat, w, (hipStream_t)st, false);
        return rc;
    }
    return phl_launch_splat(lat, src, src_rs, vd, vert, (hipStream_t)st);
}

int64_t phl_num_chunks(const phl_lattice *lat) { return lat ? lat->nchunks : -1; }
int64_t phl_partial_rows(const phl_lattice *lat) { return lat ? lat->S_multi : -1; }

int phl_chunks_touching(phl_lattice *lat, const int64_t *rows_dev, int64_t k, int32_t *mask_host, phl_stream st)
{
    if (!lat || k < 0 || (k > 0 && !rows_dev) || !mask_host) { phl_set_error("phl_chunks_touching: bad arguments"); return PHL_ERR_INVALID; }
    device_guard g(lat->device);
    return phl_tiles_chunks_touching(lat, rows_dev, k, mask_host, (hipStream_t)st);
}

int phl_splat_part(phl_lattice *lat, const float *src, int vd, int64_t src_rs, float *vert, float *partial_dev,
                   const int32_t *chunks_dev, int64_t nchunks_sel, const int32_t *rows_dev, int64_t nrows, phl_stream st)
{
    if (!lat || vd < 0 || nchunks_sel < 0 || nrows < 0 || !src || !vert || (nchunks_sel > 0 && !chunks_dev) || (nrows > 0 && !rows_dev) ||
        (lat->S_multi > 0 && !partial_dev)) {
        phl_set_error("phl_splat_part: bad arguments");
        return PHL_ERR_INVALID;
    }
    device_guard g(lat->device);
    if (!use_tiled_splat(lat, vd, 0, src, vert, src_rs) || (reinterpret_cast<uintptr_t>(partial_dev) & 15)) {
        phl_set_error("phl_splat_part: the chunk splat is not available for this shape (vd %% 4, alignment, sharing)");
        return PHL_ERR_UNSUPPORTED;
    }
    return phl_launch_splat_tiled(lat, src, src_rs, vd, vert, partial_dev, (hipStream_t)st, /*subset=*/true, chunks_dev,
                                  (int)nchunks_sel, rows_dev, nrows);
}

int phl_splat_part_pack(phl_lattice *lat, const float *src, int vd, int64_t src_rs, float *vert, float *partial_dev,
                        const int32_t *chunks_dev, int64_t nchunks_sel, const int32_t *rows_dev, int64_t nrows,
                        const int32_t *pack_pos_dev, float *pack_dev, int64_t pack_rs, phl_stream st)
{
    if (!lat || vd < 0 || nchunks_sel < 0 || nrows < 0 || !vert || (nchunks_sel > 0 && (!chunks_dev || !src)) || (nrows > 0 && !rows_dev) ||
        (lat->S_multi > 0 && !partial_dev) || (pack_pos_dev && (!pack_dev || pack_rs < vd || pack_rs % 4 ||
                                                                   (reinterpret_cast<uintptr_t>(pack_dev) & 15)))) {
        phl_set_error("phl_splat_part_pack: bad arguments");
        return PHL_ERR_INVALID;
    }
    device_guard g(lat->device);
    if (!use_tiled_splat(lat, vd, 0, src, vert, src_rs) || (reinterpret_cast<uintptr_t>(partial_dev) & 15)) {
        phl_set_error("phl_splat_part_pack: the chunk splat is not available for this shape (vd %% 4, alignment, sharing)");
        return PHL_ERR_UNSUPPORTED;
    }
    return phl_launch_splat_tiled(lat, src, src_rs, vd, vert, partial_dev, (hipStream_t)st, /*subset=*/true, chunks_dev,
                                  (int)nchunks_sel, rows_dev, nrows, nullptr, pack_pos_dev, pack_dev, pack_rs);
}

int phl_blur_axis(phl_lattice *lat, int axis, const float *vin, float *vout, int vd, phl_stream st)
{
    if (!lat || axis < 0 || axis > lat->d || vd < 0 || vin == vout) { phl_set_error("phl_blur_axis: bad arguments"); return PHL_ERR_INVALID; }
    device_guard g(lat->device);
    return phl_launch_blur(lat, axis, vin, vout, vd, (hipStream_t)st);
}

// all blur axes, two per pass where a pair is left; returns the index (0/1) of the buffer holding the result
static int blur_all(const phl_lattice *lat, float *const buf[2], int vd, hipStream_t st, int *cur_out)
{
    static const bool pairs = !(getenv("PHL_BLUR_PAIRS") && atoi(getenv("PHL_BLUR_PAIRS")) == 0);
    int cur = 0;
    for (int axis = 0; axis <= lat->d;) {  // axis order 0..d, Jacobi ping-pong (:498, :530-532)
        int rc;
        if (pairs && axis + 1 <= lat->d && lat->nbr2) {
            rc = phl_launch_blur2(lat, axis / 2, buf[cur], buf[cur ^ 1], vd, st, /*restricted=*/true);
            axis += 2;
        } else {
            rc = phl_launch_blur(lat, axis, buf[cur], buf[cur ^ 1], vd, st, /*restricted=*/true);
            axis += 1;
        }
        if (rc) return rc;
        cur ^= 1;
    }
    *cur_out = cur;
    return PHL_OK;
}

int phl_blur(phl_lattice *lat, float *vert_a, float *vert_b, int vd, int *result_in_b, phl_stream st)
{
    if (!lat || vd < 0 || !result_in_b || vert_a == vert_b || (lat->M > 0 && vd > 0 && (!vert_a || !vert_b))) {
        phl_set_error("phl_blur: bad arguments");
        return PHL_ERR_INVALID;
    }
    device_guard g(lat->device);
    float *const buf[2] = {vert_a, vert_b};
    return blur_all(lat, buf, vd, (hipStream_t)st, result_in_b);
}

int phl_gather_rows(const float *vert, int vd, const int64_t *idx_dev, int64_t k, float *out, int64_t out_rs, phl_stream st)
{
    if (vd < 0 || k < 0 || (k > 0 && vd > 0 && (!vert || !idx_dev || !out))) { phl_set_error("phl_gather_rows: bad arguments"); return PHL_ERR_INVALID; }
    return phl_launch_rows(false, const_cast<float *>(vert), vd, idx_dev, k, out, out_rs, (hipStream_t)st);
}

int phl_scatter_add_rows(float *vert, int vd, const int64_t *idx_dev, int64_t k, const float *in, int64_t in_rs, phl_stream st)
{
    if (vd < 0 || k < 0 || (k > 0 && vd > 0 && (!vert || !idx_dev || !in))) { phl_set_error("phl_scatter_add_rows: bad arguments"); return PHL_ERR_INVALID; }
    return phl_launch_rows(true, vert, vd, idx_dev, k, const_cast<float *>(in), in_rs, (hipStream_t)st);
}

int phl_slice(phl_lattice *lat, const float *vert, int vd, float *out, int64_t out_rs, const float *sub, int64_t sub_rs,
              unsigned flags, phl_stream st)
{
    if (!lat || vd < 0) { phl_set_error("phl_slice: bad arguments"); return PHL_ERR_INVALID; }
    device_guard g(lat->device);
    if (use_tiled_slice(lat, vd, flags, vert, out, sub, out_rs, sub ? sub_rs : 0))
        return phl_launch_slice_tiled(lat, vert, vd, out, out_rs, sub, sub_rs, flags, (hipStream_t)st);
    return phl_launch_slice(lat, vert, vd, out, out_rs, sub, sub_rs, flags, (hipStream_t)st);
}

// vdw: the width the kernels run at -- vd, or vd rounded up to a multiple of 4 when both sides are staged for that (the
// extra channels hold whatever the staging buffer held: channels never mix, and they are not copied out)
static int filter_on(phl_lattice *lat, phl_workspace *w, const float *src, int vd_user, int vdw, int64_t src_rs, int64_t src_cs, float *out,
                     int64_t out_rs, int64_t out_cs, unsigned flags, bool stage_src, bool stage_dst, hipStream_t st)
{
    const int64_t n = lat->n;
    const float *src_eff = src;
    int64_t src_eff_rs = src_rs;
    float *out_eff = out;
    int64_t out_eff_rs = out_rs;
    int rc = PHL_OK;
    const int vd = vdw;
    if (stage_src) {
        rc = phl_launch_copy2d(src, src_rs, src_cs, w->stage_in, vdw, 1, n, vd_user, st);
        if (rc) return rc;
        src_eff = w->stage_in;
        src_eff_rs = vdw;
    }
    if (stage_dst) {
        out_eff = w->stage_out;
        out_eff_rs = vdw;
    }
    if (use_tiled_splat(lat, vd, flags, src_eff, w->buf[0], src_eff_rs)) rc = phl_launch_splat_tiled(lat, src_eff, src_eff_rs, vd, w->buf[0], w->partial, st);
    else rc = phl_launch_splat(lat, src_eff, src_eff_rs, vd, w->buf[0], st);
    if (rc) return rc;
    int cur = 0;
    rc = blur_all(lat, w->buf, vd, st, &cur);
    if (rc) return rc;
    const float *sub = (flags & PHL_FILTER_SUBTRACT_INPUT) ? src_eff : nullptr;
    if (use_tiled_slice(lat, vd, flags, w->buf[cur], out_eff, sub, out_eff_rs, sub ? src_eff_rs : 0))
        rc = phl_launch_slice_tiled(lat, w->buf[cur], vd, out_eff, out_eff_rs, sub, src_eff_rs, flags, st);
    else
        rc = phl_launch_slice(lat, w->buf[cur], vd, out_eff, out_eff_rs, sub, src_eff_rs, flags, st);
    if (rc) return rc;
    if (stage_dst) rc = phl_launch_copy2d(w->stage_out, vdw, 1, out, out_rs, out_cs, n, vd_user, st);
    return rc;
}

int phl_filter(phl_lattice *lat, const float *src, int vd, int64_t src_rs, int64_t src_cs, float *out, int64_t out_rs,
               int64_t out_cs, unsigned flags, phl_stream stream)
{
    if (!lat || vd < 0) { phl_set_error("phl_filter: bad arguments"); return PHL_ERR_INVALID; }
    const int64_t n = lat->n;
    if (n == 0 || vd == 0) return PHL_OK;
    if (!src || !out) { phl_set_error("phl_filter: NULL src/out"); return PHL_ERR_INVALID; }
    if (src == out) { phl_set_error("phl_filter: out may not alias src"); return PHL_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    device_guard g(lat->device);

    // pixel-major rows are consumed in place; anything else (e.g. the [n,L] view of an NCHW
    // tensor, gaussian_matrix.py:348) is staged through one coalesced transpose
    bool stage_src = (src_cs != 1) && vd > 1;
    bool stage_dst = (out_cs != 1) && vd > 1;
    // Rows the chunk kernels cannot take -- a channel count that is not a multiple of 4 (the reference's max_disp = w // 6:
    // 231 at 1390 columns), a row stride or base address off the 16-byte grid (a column slice of a wider tensor) -- used to
    // fall back to the gather kernels (2.4x slower at 1390x1110x231).  From 128 channels on they are staged instead: one
    // copy into / out of a buffer with 16-byte rows, the kernels run at the width rounded up to a multiple of 4.  (Below
    // that the two extra passes over the volume cost more than the gather kernels lose: tools/odd_narrow.py, 1390x1110 --
    // 9 channels 0.46 vs 0.27 ms, 50: 0.87 vs 0.83, 110: 1.29 vs 1.20, 130: 1.48 vs 1.59, 231: 2.59 vs 3.14.)
    int vdw = vd;
    if (vd >= 128 && !(flags & (PHL_FILTER_EXACT | PHL_FILTER_NO_TILES))) {
        auto off_grid = [](const void *p, int64_t rs) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0 || rs % 4 != 0; };
        if (vd % 4) {
            vdw = (vd + 3) & ~3;
            stage_src = stage_dst = true;
        } else {
            if (!stage_src && off_grid(src, src_rs)) stage_src = true;
            if (!stage_dst && off_grid(out, out_rs)) stage_dst = true;
        }
    }
    phl_workspace *w = nullptr;
    int rc = phl_ws_acquire(lat, st, lat->M * (int64_t)vdw, need_partial(lat, vdw, flags),
                            (stage_src || stage_dst) ? n * (int64_t)vdw : 0, &w);
    if (rc) return rc;
    rc = filter_on(lat, w, src, vd, vdw, src_rs, src_cs, out, out_rs, out_cs, flags, stage_src, stage_dst, st);
    phl_ws_release(lat, w, st, false);
    return rc;
}

int phl_filter_grad(phl_lattice *lat, const float *src, int64_t src_rs, const float *g, int64_t g_rs, int L, const float *ref,
                    int64_t ref_rs, int64_t ref_cs, float *grad_ref, float *grad_src, int64_t grad_src_rs, phl_stream stream)
{
    if (!lat || L < 0 || !grad_ref) { phl_set_error("phl_filter_grad: bad arguments"); return PHL_ERR_INVALID; }
    const int64_t n = lat->n;
    const int d = lat->d;
    if (n == 0) return PHL_OK;
    if (!src || !g || !ref) { phl_set_error("phl_filter_grad: NULL src / g / ref"); return PHL_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    device_guard guard(lat->device);
    if (L == 0) {
        PHL_HIP(hipMemsetAsync(grad_ref, 0, sizeof(float) * (size_t)n * d, st));
        return PHL_OK;
    }
    const int64_t vdw = (int64_t)L * (d + 1);
    if (d > 7 || L % 4 != 0 || !use_tiled_splat(lat, L, 0, src, g, src_rs) || g_rs % 4 || vdw > (1 << 20) ||
        (grad_src && ((reinterpret_cast<uintptr_t>(grad_src) & 15) || grad_src_rs % 4))) {
        phl_set_error("phl_filter_grad: shape not covered by the fused path (d <= 7, L %% 4 == 0, 16-byte aligned pixel-major rows, "
                      "chunk splat available); filter the 2L(1+d)-channel operand instead");
        return PHL_ERR_UNSUPPORTED;
    }
    // Workspace: the wide vertex rows (two ping-pong buffers) and the wide partial rows, (2M + S_multi) * (1+d) * Lg floats
    // for a group of Lg channels.  The contraction is a sum over channels, so the call runs in CHANNEL GROUPS whenever
    // all L at once would exceed the budget (PHL_GRAD_WS_MB, default 24576): features with little sharing push S_multi
    // towards 2n, and (2M + S_multi)(1+d)L floats would then be as large as the reference's wide operand that this
    // path exists to avoid.  Groups are multiples of 64 channels (the wide splat's slab) where L allows.
    const char *ws_env = getenv("PHL_GRAD_WS_MB");      // (read per call: a backward pass is milliseconds, and tests vary it)
    const int64_t ws_budget = (int64_t)(ws_env && atoll(ws_env) > 0 ? atoll(ws_env) : 24576) << 20;
    const int64_t rows_ws = 2 * lat->M + lat->S_multi;
    int Lg = L;
    if (rows_ws * vdw * (int64_t)sizeof(float) > ws_budget) {
        const int64_t per_ch = rows_ws * (d + 1) * (int64_t)sizeof(float);
        int64_t fit = ws_budget / (per_ch > 0 ? per_ch : 1);
        fit = fit >= 64 ? fit / 64 * 64 : fit / 4 * 4;
        if (fit < 4) {
            phl_set_error("phl_filter_grad: %lld vertex + partial rows do not fit the workspace budget even four channels at a time "
                          "(PHL_GRAD_WS_MB)", (long long)rows_ws);
            return PHL_ERR_UNSUPPORTED;
        }
        Lg = (int)(fit < L ? fit : L);
    }
    const int64_t vdg = (int64_t)Lg * (d + 1);
    phl_workspace *w = nullptr;
    int rc = phl_ws_acquire(lat, st, lat->M * vdg, lat->S_multi * vdg, 0, &w);
    if (rc) return rc;
    // Two passes of the same three stages, each over (1+d) L channels instead of the reference's 2 (1+d) L at once:
    //   pass 1  x = g,   y = src:  T  = -2 sum_l src (f Wg - W(g f))      (+ Wg itself = the gradient w.r.t. src)
    //   pass 2  x = src, y = g:    T += -2 sum_l g   (f Ws - W(s f))
    const float *xs[2] = {g, src}, *ys[2] = {src, g};
    const int64_t xrs[2] = {g_rs, src_rs}, yrs[2] = {src_rs, g_rs};
    for (int c0 = 0; c0 < L && rc == PHL_OK; c0 += Lg) {
        const int Lc = L - c0 < Lg ? L - c0 : Lg;
        for (int pass = 0; pass < 2 && rc == PHL_OK; pass++) {
            phl_splat_wide wide = {d + 1, ref, ref_rs, ref_cs};
            rc = phl_launch_splat_tiled(lat, xs[pass] + c0, xrs[pass], Lc, w->buf[0], w->partial, st, false, nullptr, 0, nullptr, 0, &wide);
            if (rc) break;
            int cur = 0;
            rc = blur_all(lat, w->buf, Lc * (d + 1), st, &cur);
            if (rc) break;
            rc = phl_launch_slice_grad(lat, w->buf[cur], Lc, ys[pass] + c0, yrs[pass], ref, ref_rs, ref_cs, grad_ref, (pass || c0) ? 1 : 0,
                                       (pass == 0 && grad_src) ? grad_src + c0 : nullptr, grad_src_rs, st);
        }
    }
    phl_ws_release(lat, w, st, false);
    return rc;
}

int phl_copy2d(const float *src, int64_t src_rs, int64_t src_cs, float *dst, int64_t dst_rs, int64_t dst_cs, int64_t rows,
               int cols, phl_stream stream)
{
    if (rows < 0 || cols < 0 || (rows > 0 && cols > 0 && (!src || !dst))) { phl_set_error("phl_copy2d: bad arguments"); return PHL_ERR_INVALID; }
    if (rows == 0 || cols == 0) return PHL_OK;
    return phl_launch_copy2d(src, src_rs, src_cs, dst, dst_rs, dst_cs, rows, cols, (hipStream_t)stream);
}

int phl_filter_once(const float *src, int vd, int64_t src_rs, int64_t src_cs, const float *ref, int d, int64_t ref_rs,
                    int64_t ref_cs, int64_t n, float *out, int64_t out_rs, int64_t out_cs, unsigned flags, int device,
                    phl_stream stream)
{
    phl_lattice *lat = nullptr;
    int rc = phl_build_ex(&lat, ref, n, d, ref_rs, ref_cs, device, stream,
                          (flags & PHL_FILTER_CLEAN_TABLE) ? PHL_BUILD_DEFAULT : PHL_BUILD_REFERENCE_TABLE);
    if (rc) return rc;
    rc = phl_filter(lat, src, vd, src_rs, src_cs, out, out_rs, out_cs, flags & ~(unsigned)PHL_FILTER_CLEAN_TABLE, stream);
    if (rc == PHL_OK) {
        device_guard g(device);
        hipError_t e = hipStreamSynchronize((hipStream_t)stream);  // workspace dies with the lattice
        if (e != hipSuccess) rc = phl_hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__);
    }
    phl_destroy(lat);
    return rc;
}

// first-touch <-> internal maps on the host (empty vectors: the numberings coincide)
static int fetch_maps(phl_lattice *lat, std::vector<int32_t> &ft_of_int, std::vector<int32_t> &int_of_ft)
{
    ft_of_int.clear();
    int_of_ft.clear();
    if (!lat->int_of_ft || lat->M == 0) return PHL_OK;
    ft_of_int.resize((size_t)lat->M);
    int_of_ft.resize((size_t)lat->M);
    PHL_HIP(hipMemcpy(ft_of_int.data(), lat->ft_of_int, sizeof(int32_t) * (size_t)lat->M, hipMemcpyDeviceToHost));
    PHL_HIP(hipMemcpy(int_of_ft.data(), lat->int_of_ft, sizeof(int32_t) * (size_t)lat->M, hipMemcpyDeviceToHost));
    return PHL_OK;
}

int phl_get_vertex_order(phl_lattice *lat, int32_t *row_of_vertex)
{
    if (!lat || !row_of_vertex) { phl_set_error("phl_get_vertex_order: bad arguments"); return PHL_ERR_INVALID; }
    device_guard g(lat->device);
    PHL_HIP(hipDeviceSynchronize());
    if (lat->int_of_ft) PHL_HIP(hipMemcpy(row_of_vertex, lat->int_of_ft, sizeof(int32_t) * (size_t)lat->M, hipMemcpyDeviceToHost));
    else
        for (int64_t v = 0; v < lat->M; v++) row_of_vertex[v] = (int32_t)v;
    return PHL_OK;
}

int phl_get_pixel_order(phl_lattice *lat, int32_t *pix_order)
{
    if (!lat || !pix_order) { phl_set_error("phl_get_pixel_order: bad arguments"); return PHL_ERR_INVALID; }
    if (lat->n == 0) return PHL_OK;
    if (!lat->pix_order) { phl_set_error("phl_get_pixel_order: the lattice has no chunks"); return PHL_ERR_UNSUPPORTED; }
    device_guard g(lat->device);
    PHL_HIP(hipDeviceSynchronize());
    PHL_HIP(hipMemcpy(pix_order, lat->pix_order, sizeof(int32_t) * (size_t)lat->n, hipMemcpyDeviceToHost));
    return PHL_OK;
}

int phl_get_keys(phl_lattice *lat, int16_t *keys)
{
    if (!lat || !keys) { phl_set_error("phl_get_keys: bad arguments"); return PHL_ERR_INVALID; }
    if (lat->M == 0) return PHL_OK;
    device_guard g(lat->device);
    PHL_HIP(hipDeviceSynchronize());
    std::vector<int32_t> f, r;
    int rc = fetch_maps(lat, f, r);
    if (rc) return rc;
    const int d = lat->d;
    if (r.empty()) {
        PHL_HIP(hipMemcpy(keys, lat->vkeys, sizeof(int16_t) * (size_t)lat->M * d, hipMemcpyDeviceToHost));
        return PHL_OK;
    }
    std::vector<int16_t> h((size_t)lat->M * d);
    PHL_HIP(hipMemcpy(h.data(), lat->vkeys, sizeof(int16_t) * h.size(), hipMemcpyDeviceToHost));
    for (int64_t v = 0; v < lat->M; v++) memcpy(keys + v * d, h.data() + (size_t)r[v] * d, sizeof(int16_t) * d);
    return PHL_OK;
}

int phl_get_replay(phl_lattice *lat, int32_t *vid, float *w)
{
    if (!lat || !vid || !w) { phl_set_error("phl_get_replay: bad arguments"); return PHL_ERR_INVALID; }
    if (lat->N == 0) return PHL_OK;
    device_guard g(lat->device);
    std::vector<phl_replay_t> h((size_t)lat->N);
    PHL_HIP(hipDeviceSynchronize());
    PHL_HIP(hipMemcpy(h.data(), lat->replay, sizeof(phl_replay_t) * h.size(), hipMemcpyDeviceToHost));
    std::vector<int32_t> f, r;
    int rc = fetch_maps(lat, f, r);
    if (rc) return rc;
    for (size_t i = 0; i < h.size(); i++) { vid[i] = f.empty() ? h[i].vid : f[h[i].vid]; w[i] = h[i].w; }
    return PHL_OK;
}

int phl_get_neighbors(phl_lattice *lat, int32_t *nbr)
{
    if (!lat || !nbr) { phl_set_error("phl_get_neighbors: bad arguments"); return PHL_ERR_INVALID; }
    if (lat->M == 0) return PHL_OK;
    device_guard g(lat->device);
    PHL_HIP(hipDeviceSynchronize());
    std::vector<int32_t> f, r;
    int rc = fetch_maps(lat, f, r);
    if (rc) return rc;
    const size_t M = (size_t)lat->M;
    const int axes = lat->d + 1;
    if (r.empty()) {
        PHL_HIP(hipMemcpy(nbr, lat->nbr, sizeof(int32_t) * M * axes * 2, hipMemcpyDeviceToHost));
        return PHL_OK;
    }
    std::vector<int32_t> h(M * axes * 2);
    PHL_HIP(hipMemcpy(h.data(), lat->nbr, sizeof(int32_t) * h.size(), hipMemcpyDeviceToHost));
    for (int a = 0; a < axes; a++)
        for (size_t v = 0; v < M; v++)
            for (int s2 = 0; s2 < 2; s2++) {
                const int32_t x = h[((size_t)a * M + r[v]) * 2 + s2];
                nbr[((size_t)a * M + v) * 2 + s2] = x < 0 ? x : f[x];
            }
    return PHL_OK;
}

int phl_get_splat_lists(phl_lattice *lat, int32_t *ptr, int32_t *pixel, float *w)
{
    if (!lat || !ptr || !pixel || !w) { phl_set_error("phl_get_splat_lists: bad arguments"); return PHL_ERR_INVALID; }
    if (lat->N == 0) { ptr[0] = 0; return PHL_OK; }
    device_guard g(lat->device);
    const int rc0 = phl_ensure_csr(lat, nullptr);
    if (rc0) return rc0;
    std::vector<phl_contrib_t> h((size_t)lat->N);
    std::vector<int32_t> p((size_t)lat->M + 1);
    PHL_HIP(hipDeviceSynchronize());
    PHL_HIP(hipMemcpy(p.data(), lat->csr_ptr, sizeof(int32_t) * p.size(), hipMemcpyDeviceToHost));
    PHL_HIP(hipMemcpy(h.data(), lat->csr, sizeof(phl_contrib_t) * h.size(), hipMemcpyDeviceToHost));
    std::vector<int32_t> f, r;
    int rc = fetch_maps(lat, f, r);
    if (rc) return rc;
    int64_t o = 0;
    for (int64_t v = 0; v < lat->M; v++) {          // lists in first-touch vertex order
        const int64_t row = r.empty() ? v : r[v];
        ptr[v] = (int32_t)o;
        for (int32_t e = p[row]; e < p[row + 1]; e++, o++) { pixel[o] = h[e].pixel; w[o] = h[e].w; }
    }
    ptr[lat->M] = (int32_t)o;
    return PHL_OK;
}

}  // extern "C"
