// Internal declarations shared by the phl translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "phl.h"

#include <vector>

#define PHL_WAVE 64
#define PHL_MAX_HIDDEN 64   // duplicate vertices of the reference-table mode (<= 2 per table doubling)
#define PHL_EMPTY 0x7FFFFFFF  // empty hash slot (larger than any candidate index)

// (vertex, weight) per (pixel, remainder): the sparse n x M splat matrix, d+1 entries per
// row.  Same content as the reference's ReplayEntry (permutohedral.h:558-561) with a vertex
// id instead of a value offset.
struct phl_replay_t {
    int32_t vid;
    float w;
};
// (pixel, weight): the transpose of the above, grouped by vertex and sorted by pixel.
struct phl_contrib_t {
    int32_t pixel;
    float w;
};

struct phl_lattice {
    int device;
    int d;
    int64_t n;
    int64_t M;
    int64_t N;  // n*(d+1) (pixel, remainder) pairs

    int16_t *vkeys;         // [M][d]
    phl_replay_t *replay;   // [n][d+1]
    int32_t *csr_ptr;       // [M+1]
    phl_contrib_t *csr;     // [N] grouped by vertex, pixel-ascending inside a group
    int32_t *nbr;           // [d+1][M][2]
    int32_t *nbr2;          // [(d+1)/2][M][8] composed neighbour ids of axis pairs (2p, 2p+1), see k_blur2
    int *table;             // open-addressing key -> -(vid+1), PHL_EMPTY = free; kept for phl_add_vertices
    uint32_t table_mask;
    int64_t M_local;        // vertices created by this lattice's own pixels (ghosts come after)
    // Internal vertex numbering (phl_renumber_vertices): rows of every [M][..] array are in LOCALITY order, not in
    // first-touch order.  Both maps are null when the two coincide.  The public introspection calls translate.
    int32_t *vfirst;        // [M] first-touch candidate (pixel*(d+1)+remainder) per vertex: build-time only, may be null
    int64_t vfirst_valid_for_M;   // the reference-table replay wrote vfirst for this many vertices (0: it did not)
    // Build-time only (null outside phl_build_device ... phl_tiles_build): every candidate's slot in the build's key
    // table and the table itself (slot -> -(clean vertex + 1)); the reference-table step's renaming (clean -> reference
    // vertex, per-candidate segments for keys with several vertices).  replay[].vid is written ONCE, at the end, through
    // all of them and the locality numbering (k_final_vid) instead of once per renaming.
    int32_t *bt_slot_of, *bt_table;
    int32_t *bt_remap, *bt_dup_ptr, *bt_seg_e, *bt_seg_id;
    // ... and the grid cell of every pixel + the grid the chunk order was made on, when the pixel order has been made
    // ahead of phl_tiles_build (phl_tiles_pixel_order, launched under the host's table replay); else null
    int32_t *bt_cell;
    int grid_nca, grid_ncb;
    int32_t *ft_of_int;     // [M] first-touch id of internal vertex i
    int32_t *int_of_ft;     // [M] internal id (row) of first-touch vertex v

    // pixel chunks ("tiles") for the LDS-staged splat / slice (phl_tiles.hip)
    int P;                  // pixels per chunk
    int nchunks;
    int nv_max;             // max local vertices of any chunk
    int64_t S;              // total (chunk, local vertex) slots
    int64_t S_multi;        // slots whose vertex has contributions from several chunks
    int32_t *pix_order;     // [n] pixels in cell-major order; chunk c = pix_order[c*P ...)
    int32_t *chunk_vptr;    // [nchunks+1] slot range of each chunk
    int32_t *slot_vert;     // [S] vertex id (bit 31: this chunk is the vertex's only contributor)
    int32_t *slot_pidx;     // [S+1] row of the slot in the partial buffer (multi-chunk vertices)
    int2 *seg_rng;          // [S] {begin, end} of each slot's segment in seg[]
    phl_contrib_t *seg;     // [N] {pixel index inside the chunk, weight}, ascending pixel per slot
    unsigned short *lidx;   // [N] local vertex index per (chunk pixel, remainder)
    int32_t *vs_ptr;        // [M+1] slots of each vertex ...
    phl_contrib_t *vs;      // [S]   ... ascending (slot index in .pixel)
    int32_t *vorder;        // [M] vertices in chunk-major order (gather splat locality); may be null
    int32_t *chunk_by_nv;   // [nchunks] chunk ids by descending local-vertex count (heavy class = a prefix)
    int *nv_cum;            // HOST [nv_max+1]: number of chunks with at most x local vertices
    int32_t *vlong;         // [n_long] vertices fed by more than LONG_LIST chunks (k_splat_reduce_long)
    int64_t n_long;
    // feature ranges found while elevating (phl_build_device): the chunk grid is laid over the two widest
    float feat_lo[PHL_MAX_D], feat_hi[PHL_MAX_D];
    int feat_range_valid;
    int64_t tile_bytes;

    // value workspaces (phl_api.hip): a filter call takes one for the duration of its launches, so any number
    // of host threads / streams may filter through one lattice at the same time
    struct phl_shared *shared;

    int64_t table_bytes;    // device bytes of the persistent tables

    // PHL_BUILD_REFERENCE_TABLE (phl_reftable.hip): vertices the reference's final hash table cannot reach
    // (duplicates of a key; never anybody's blur neighbour), and the one neighbour entry a doubling inside
    // blur() decides (-2 = none)
    // Row-band lattices (phl_set_blur_rows): for every blur axis the rows whose OUTPUT of that axis anything later
    // reads, as up to three ascending row ranges {begin, end}; blur_rows_set = 0: all M rows on every axis.
    int blur_rows_set;
    int32_t blur_rows[PHL_MAX_D + 1][3][2];
    unsigned build_flags;
    int n_hidden;
    int32_t hidden[PHL_MAX_HIDDEN];
    int32_t nbr00_override;
};

// host replay of the reference's hash table (phl_reftable.hip)
struct phl_reftable_query {
    virtual int vid_at(int64_t candidate) = 0;                          // clean vertex id of a candidate
    virtual int64_t next_occurrence(int clean_vid, int64_t after) = 0;  // next candidate with that key, or -1
    // A linear-probing table of `cap` slots holds the clean vertices [0, n_clean), one more entry for every key of
    // extra_clean, and (counted in addition: a superset of the real occupancy) one entry filed from the home under
    // cap / 2 for every key of stale_clean.  For every key of `check`: is there an empty slot between the key's home
    // and the table's last slot, i.e. do the key's entries lie in index order along their probe path?  1 = yes for
    // all (certain), 0 = not for all, or it cannot be told.  Depends on the occupancy only, not on the insertion order.
    // Questions are SUBMITTED as the replay goes (the device answers them in stream order, without a round trip each)
    // and the conjunction of the answers is collected once at the end.
    virtual void probe_paths_submit(int64_t n_clean, const std::vector<int32_t> &extra_clean,
                                    const std::vector<int32_t> &stale_clean, uint64_t cap,
                                    const std::vector<int32_t> &check) = 0;
    virtual int probe_paths_all_ok() = 0;
    // The replay needs the key of a handful of vertices only (its hash: which half of a doubled table the key lands in).
    // With the keys on the host (keys_clean != NULL in phl_reference_table_fast) it hashes them itself; without, it asks:
    virtual uint64_t key_hash(int clean_vid) { (void)clean_vid; return 0; }
    // ... and it announces the vid_at() / key_hash() questions it is likely to ask (candidate, or a vertex already known:
    // known[i] >= 0), so that an implementation with a slow round trip can answer them in one
    virtual void prefetch(const std::vector<int64_t> &cands, const std::vector<int32_t> &known) { (void)cands; (void)known; }
    int probe_paths_do_not_wrap(int64_t n_clean, const std::vector<int32_t> &extra_clean, const std::vector<int32_t> &stale_clean,
                                uint64_t cap, const std::vector<int32_t> &check)
    {
        probe_paths_submit(n_clean, extra_clean, stale_clean, cap, check);
        return probe_paths_all_ok();
    }
    virtual ~phl_reftable_query() {}
};
struct phl_reftable_result {
    int64_t M_ref;
    std::vector<int16_t> keys;      // [M_ref][d] in the reference's insertion order
    std::vector<int32_t> remap;     // [M_clean] reference vertex, or -(k+1) for tracked key k
    std::vector<int32_t> dup_clean, dup_ptr, seg_e, seg_id;
    std::vector<int32_t> hidden;
    bool blur_grow;
    int32_t blur_first_nbr;
    // compact form (analytic replay): the reference order is the clean order with a few extra creations inserted --
    // ex_id ascending reference ids, ex_clean the clean vertex each one repeats.  keys / remap are then left empty and
    // are produced where they are needed (on the device: phl_apply_reference_table; phl_reftable_expand on the host).
    bool compact = false;
    std::vector<int32_t> ex_id, ex_clean;
};
void phl_reftable_expand(const int16_t *keys_clean, int64_t M, int d, phl_reftable_result &R);
int phl_reference_table_sim(const int16_t *keys_clean, const int32_t *efirst, int64_t M, int d, int64_t N,
                            phl_reftable_query &q, phl_reftable_result &out);
// the same result without simulating the table (phl_reftable.hip, "analytic replay"); 1 = not applicable here, use the sim
int phl_reference_table_fast(const int16_t *keys_clean, const int32_t *efirst, int64_t M, int d, int64_t N,
                             phl_reftable_query &q, phl_reftable_result &out, bool compact);
// (arena: device bytes from the caller's pool of build temporaries, phl_reftable_scratch_bytes(M) of them -- a second
//  pool would find the scratch block taken and hipMalloc / hipFree each of its requests; may be null)
size_t phl_reftable_scratch_bytes(int64_t M);
int phl_apply_reference_table(phl_lattice *lat, hipStream_t st, void *arena, size_t arena_bytes,
                              int (*under_replay)(void *), void *under_replay_arg);
// (under_replay: called once the first-touch / key copies are enqueued, before the host waits for them: launches that
//  do not depend on the replay run under it)
// replay[].vid from the build-time tables (see bt_* above), mapped through int_of_ft if there is one; releases nothing
int phl_write_final_vids(phl_lattice *lat, hipStream_t st);
void phl_release_build_tables(phl_lattice *lat);          // (after a stream synchronisation)
// steps 1-2 of phl_tiles_build (grid over the two widest features, pixels in cell-major order), launches only:
// lat->pix_order, bt_cell, grid_*.  Temporaries out of `arena` (phl_tiles_pixel_order_scratch_bytes(n) device bytes).
size_t phl_tiles_pixel_order_scratch_bytes(int64_t n);
int phl_tiles_pixel_order(phl_lattice *lat, const float *ref, int64_t rs, int64_t cs, void *arena, size_t arena_bytes, hipStream_t st);
int phl_tiles_ensure_vorder(phl_lattice *lat, hipStream_t st);   // caller holds the lattice's list lock (phl_ensure_csr)
// Pinned, device-visible host memory for the build's read-backs (thread-local bump arena).  hipMemcpyAsync into pageable
// memory blocks the host until the copy has run -- a stream synchronisation per read-back; into this it does not, and
// kernels may write into it directly.  phl_pinned_reset() at the start of a build; phl_pinned_alloc() returns null when
// the arena is too small (the caller then reads back into pageable memory; the next reset grows the arena).
void phl_pinned_reset();
void *phl_pinned_alloc(size_t bytes);

// Buffers one filter call writes: the [M][vd] Jacobi ping-pong pair, the partial rows of the chunk splat and
// the staging copies of non pixel-major inputs / outputs.  Grown on demand, reused in stream order.
struct phl_workspace {
    float *buf[2];
    int64_t buf_elems;      // capacity of each buffer in floats
    float *partial;         // [S_multi][vd] partial splat sums
    int64_t partial_elems;
    float *stage_in;        // [n][vd]
    float *stage_out;
    int64_t stage_elems;
    hipStream_t last_stream;
    bool stream_bound;      // work that uses the buffers may still be pending on last_stream
    // completion mark: a one-thread kernel behind the last launch stores `ticket` into this pinned host word, so
    // "has its work drained?" is a plain host load -- legal at any time, also while some stream is being captured
    // (event queries are not)
    unsigned long long *done_word;
    unsigned long long ticket;
    bool in_enqueue;        // a host thread is issuing launches on it right now
    bool captured;          // a HIP graph holds its pointers: never resized or handed to another stream
};
int phl_ws_acquire(phl_lattice *lat, hipStream_t st, int64_t buf_elems, int64_t partial_elems, int64_t stage_elems,
                   phl_workspace **out);
void phl_ws_release(phl_lattice *lat, phl_workspace *ws, hipStream_t st, bool idle);

// thread-local error message
void phl_set_error(const char *fmt, ...);
int phl_hip_fail(hipError_t e, const char *what, const char *file, int line);
#define PHL_HIP(call)                                                              \
    do {                                                                           \
        hipError_t e__ = (call);                                                   \
        if (e__ != hipSuccess) return phl_hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

// cached device blocks for the arrays a lattice owns (phl_api.hip)
hipError_t phl_dev_malloc(void **p, size_t bytes);
hipError_t phl_dev_free(void *p);

// cached device scratch for build temporaries (phl_api.hip); one user at a time per process
bool phl_scratch_acquire(void **base, size_t *cap);
void phl_scratch_release(size_t wanted_bytes);

int phl_lattice_blank(phl_lattice **out, int device, int d, int64_t n);      // phl_api.hip (extern "C" there)

// ---- launchers implemented in phl_build.hip ----
int phl_build_device(phl_lattice *lat, const float *ref, int64_t rs, int64_t cs, hipStream_t st);
int phl_ensure_csr(phl_lattice *lat, hipStream_t st);  // pixel-sorted lists, built on first use
namespace std { class mutex; }
std::mutex *phl_csr_mutex(phl_lattice *lat);
int phl_add_vertices_device(phl_lattice *lat, const int16_t *keys_host, int64_t count, int32_t *vid_host, hipStream_t st);
int phl_rebuild_table_and_neighbors(phl_lattice *lat, hipStream_t st, void **scratch_out);

// ---- implemented in phl_tiles.hip ----
int phl_tiles_build(phl_lattice *lat, const float *ref, int64_t rs, int64_t cs, hipStream_t st);   // + renumbering, tables
int phl_tiles_link_vertices(phl_lattice *lat, hipStream_t st);
int phl_tiles_free(phl_lattice *lat);
int phl_tiles_lprs(const phl_lattice *lat, int vd, int for_slice);  // -1: LDS-staged path unavailable
// wide splat (phl_filter_grad): block 0 = splat(src), block 1+k = splat(src (x) fref[:, k]); rows of nsets * vd floats
struct phl_splat_wide {
    int nsets;
    const float *fref;
    int64_t rs, cs;
};
int phl_launch_splat_tiled(phl_lattice *lat, const float *src, int64_t src_rs, int vd, float *vert, float *partial,
                           hipStream_t st, bool subset = false, const int *chunk_list = nullptr, int nlist = 0,
                           const int *vlist = nullptr, int64_t nvl = 0, const phl_splat_wide *wide = nullptr,
                           const int *pack_pos = nullptr, float *pack = nullptr, int64_t pack_rs = 0);
// (pack_pos / pack: subset calls only -- listed row i is also written to pack[pack_pos[i]] (pack_pos[i] < 0: not))
// slice of a wide vertex buffer contracted to the feature gradient (phl_tiles.hip, k_slice_grad)
int phl_launch_slice_grad(const phl_lattice *lat, const float *vertw, int L, const float *y, int64_t y_rs, const float *ref,
                          int64_t ref_rs, int64_t ref_cs, float *grad_ref, int accumulate, float *wx_out, int64_t wx_rs,
                          hipStream_t st);
int phl_tiles_chunks_touching(phl_lattice *lat, const int64_t *rows_dev, int64_t k, int32_t *mask_host, hipStream_t st);
int phl_launch_slice_tiled(const phl_lattice *lat, const float *vert, int vd, float *out, int64_t out_rs, const float *sub,
                           int64_t sub_rs, unsigned flags, hipStream_t st);

// ---- launchers implemented in phl_filter.hip ----
int phl_launch_splat(phl_lattice *lat, const float *src, int64_t src_rs, int vd, float *vert, hipStream_t st);
// (restricted: only the rows phl_set_blur_rows named for the pass's last axis are computed)
int phl_launch_blur(const phl_lattice *lat, int axis, const float *vin, float *vout, int vd, hipStream_t st, bool restricted = false);
int phl_launch_blur2(const phl_lattice *lat, int pair, const float *vin, float *vout, int vd, hipStream_t st, bool restricted = false);
int phl_launch_rows(bool scatter, float *vert, int vd, const int64_t *idx, int64_t k, float *buf, int64_t buf_rs, hipStream_t st);
int phl_launch_slice(const phl_lattice *lat, const float *vert, int vd, float *out, int64_t out_rs,
                     const float *sub, int64_t sub_rs, unsigned flags, hipStream_t st);
// generic strided 2-D copy dst[r*drs + c*dcs] = src[r*srs + c*scs], rows x cols
int phl_launch_copy2d(const float *src, int64_t srs, int64_t scs, float *dst, int64_t drs, int64_t dcs,
                      int64_t rows, int cols, hipStream_t st);
