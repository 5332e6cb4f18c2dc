/*
 * phl.h -- C ABI of the MI355X-native permutohedral-lattice filter ("phl").
 *
 * This is the drop-in boundary for the dense-CRF mean-field message-passing step of
 * mfinzi/depth-estimation.  Every entry point names the reference interface it replaces
 * (paths relative to the reference repo).  Plain pointers and sizes only: no torch types.
 *
 * Reference boundary being replaced
 *   python : latticefilter = lattice.filter            crf/gaussian_matrix.py:15-16
 *   C++    : at::Tensor filter(at::Tensor src, at::Tensor ref)
 *                                                      crf/lattice/lite/lattice.cpp:6-15
 *   engine : PermutohedralLattice::filter / splat / blur / slice
 *                                                      crf/lattice/lite/permutohedral.h:199-548
 *
 * Design difference (MI355X-first): the reference rebuilds the lattice inside every filter()
 * call although `ref` never changes during mean-field inference.  Here the
 * lattice is an object: build once per `ref` (phl_build), filter many value sets
 * (phl_filter).  phl_filter_once() keeps the reference's one-shot call shape.
 *
 * All pointers named *_dev are DEVICE pointers to fp32 (the reference is fp32 only,
 * permutohedral.h:16-19).  Strides are in ELEMENTS, so permuted NCHW views
 * (gaussian_matrix.py:348-349) are accepted without a host copy.  All work is enqueued on
 * the caller's HIP stream (NULL = default stream); functions that return host data
 * synchronise that stream.  Functions return PHL_OK or an error code and never abort;
 * phl_last_error() gives the message for the calling thread.
 */
#ifndef PHL_H
#define PHL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHL_VERSION 100 /* major*10000 + minor*100 + patch */

typedef struct phl_lattice phl_lattice; /* opaque, owns device memory */
typedef void *phl_stream;               /* hipStream_t */

enum phl_status {
    PHL_OK = 0,
    PHL_ERR_INVALID = 1,     /* bad argument (NULL, negative size, d out of range, ...) */
    PHL_ERR_SHAPE = 2,       /* rows of src and ref differ: reference asserts
                                "Incompatible shapes {}, and {}" (gaussian_matrix.py:429-430,
                                permutohedral.h:204) */
    PHL_ERR_HIP = 3,         /* a HIP runtime call failed (message has hipGetErrorString) */
    PHL_ERR_NO_DEVICE = 4,   /* no gfx950 device: the product path has NO CPU fallback */
    PHL_ERR_KEY_RANGE = 5,   /* a lattice coordinate left int16: the reference stores keys as
                                `short` (permutohedral.h:39,398) and silently wraps */
    PHL_ERR_TOO_LARGE = 6,   /* n*(d+1) does not fit int32 indexing */
    PHL_ERR_UNSUPPORTED = 7
};

enum phl_filter_flags {
    PHL_FILTER_DEFAULT = 0,
    /* out = filter(src) - src : LatticeGaussian.forward, gaussian_matrix.py:302-303, and
       BatchedAdjacency.forward :352, fused into the slice epilogue */
    PHL_FILTER_SUBTRACT_INPUT = 1,
    /* Reference-exact arithmetic: splat sums every vertex in ascending pixel order (gather
       kernel) and slice divides every term by (1 + 2^-d) (permutohedral.h:455, :480), so the
       result is BIT-IDENTICAL to the reference's CPU path.  The default instead uses the
       LDS-staged chunk splat (per-chunk partial sums) and one final multiply by 1/(1+2^-d):
       same algorithm, fp32 rounding differences of ~1e-7 relative, ~1.3x faster. */
    PHL_FILTER_EXACT = 4,
    /* do not use the LDS-staged chunk kernels at all (plain gather splat and slice) */
    PHL_FILTER_NO_TILES = 8,
    /* phl_filter_once only (it builds its own lattice): the defect-free table, one vertex per key, instead of the
       reference's table behaviour (PHL_BUILD_REFERENCE_TABLE), which is what the one-shot call -- the drop-in for
       lattice.filter(src, ref) -- builds by default */
    PHL_FILTER_CLEAN_TABLE = 16
};

enum phl_build_flags {
    PHL_BUILD_DEFAULT = 0,
    /* Reproduce the observable behaviour of the reference's hash table across its doublings
     * (permutohedral.h:59-62, :101-103: lookup() hashes with the capacity it had BEFORE lookupOffset() grows
     * the table, so the one key in flight at each doubling -- first at M = 16383 -- is probed from a stale
     * slot and usually gets a SECOND vertex; blur then sees only one of the two).  Default: every key has one
     * vertex (the algorithm without that defect; identical to the reference below M = 16383 and on all but
     * 0.3-4 % of rows above).  With this flag vertex numbering, duplicate vertices and their visibility are
     * exactly the reference's and PHL_FILTER_EXACT is bit-identical to the reference's CPU path at any size.
     * Costs one host replay of the table per build (csrc/phl_reftable.hip). */
    PHL_BUILD_REFERENCE_TABLE = 1
};

/* Limits. */
#define PHL_MAX_D 16 /* feature dimensions supported by the device build */

int phl_version(void);
const char *phl_last_error(void);
const char *phl_status_string(int status);
/* Number of visible HIP devices (0 if none); never initialises a device context. */
int phl_device_count(void);

/* ---- lattice construction ---------------------------------------------------------------
 * Replaces PermutohedralLattice(d, vd, n) + the geometry half of splat() for all n pixels
 * (permutohedral.h:328-372, :376-447, :458-460) + the neighbour lookups of blur() (:504-522).
 * ref_dev: [n][d] fp32 with element strides (row_stride, col_stride).
 * Vertices are numbered in first-touch order of the (pixel, remainder) sequence -- the same
 * numbering the reference's insertion-ordered hash table produces (permutohedral.h:70-77). */
int phl_build(phl_lattice **out, const float *ref_dev, int64_t n, int d, int64_t ref_row_stride,
              int64_t ref_col_stride, int device, phl_stream stream);
/* phl_build with phl_build_flags. */
int phl_build_ex(phl_lattice **out, const float *ref_dev, int64_t n, int d, int64_t ref_row_stride,
                 int64_t ref_col_stride, int device, phl_stream stream, unsigned build_flags);
/* Frees device memory (hipFree): like any free it must not run while a stream capture is in
 * progress on the device (the Python binding parks handles that die during a capture). */
int phl_destroy(phl_lattice *lat);

int64_t phl_num_pixels(const phl_lattice *lat);
int64_t phl_num_vertices(const phl_lattice *lat); /* == hashTable.size() after splat */
int phl_num_dims(const phl_lattice *lat);
int phl_device(const phl_lattice *lat);
/* Device bytes held by the lattice (tables + value workspace). */
int64_t phl_device_bytes(const phl_lattice *lat);

/* phl_build keeps one grow-only device scratch block for its temporaries (so that building a
 * lattice per video frame performs no hipMalloc for work arrays; PHL_SCRATCH_MAX_MB caps it,
 * 0 disables).  phl_trim_scratch() gives the block back. */
int phl_trim_scratch(void);

/* ---- row-band multi-GPU support -------------------------------------------------------------
 * Append vertices that exist in a NEIGHBOURING row band of the same image ("ghost" vertices:
 * no local pixel splats into them) so that blur() sees the same neighbourhood it would see in
 * the whole-image lattice.  keys_host: [count][d] int16, distinct.  vid_host[i] receives the
 * local vertex id of key i (an existing vertex when this band already has it, else a new id
 * >= phl_num_local_vertices()).  No reference counterpart: the reference is single-process. */
int phl_add_vertices(phl_lattice *lat, const int16_t *keys_host, int64_t count, int32_t *vid_host,
                     phl_stream stream);   /* vid_host: ROW of each key in the [M][vd] vertex buffers */
int64_t phl_num_local_vertices(const phl_lattice *lat); /* vertices created by this lattice's own pixels */
/* A band's lattice CUT OUT OF the whole image's lattice `whole` (built with PHL_BUILD_REFERENCE_TABLE, so that the band
 * inherits the reference's vertices -- the duplicates its hash table creates at its doublings, permutohedral.h:59-62,101-103,
 * and which of them every lookup resolved to -- instead of re-deriving a defect-free lattice from its own pixels).
 * Pixels [p0, p1) of `whole` become pixels 0 .. p1-p0-1; sel_host lists the n_sel vertices of `whole` to keep (first-touch
 * ids, distinct): the first n_own must cover every vertex those pixels touch, the rest are ghosts (vertices of the
 * neighbouring bands within blur's reach), kept in the caller's order behind the own vertices.  The new lattice's
 * first-touch numbering is the position in sel_host.  ref_dev: the band's own features [p1-p0][d] (chunk grid).
 * phl_vertices_of_pixels: mask_host[v] = 1 iff a pixel of [p0, p1) touches first-touch vertex v (else 0), [phl_num_vertices].
 * No reference counterpart (the reference is single-process). */
int phl_vertices_of_pixels(phl_lattice *lat, int64_t p0, int64_t p1, unsigned char *mask_host, phl_stream stream);
int phl_sub_lattice(phl_lattice **out, phl_lattice *whole, int64_t p0, int64_t p1, const int32_t *sel_host, int64_t n_sel,
                    int64_t n_own, const float *ref_dev, int64_t ref_row_stride, int64_t ref_col_stride, phl_stream stream);

/* Pre-size everything phl_filter(vd) needs (the [M][vd] ping-pong buffers, the partial-row buffer
 * of the chunk splat or the contribution lists of the gather splat) so that the NEXT call, on any stream,
 * allocates nothing and never synchronises: required before hipGraph capture.
 *
 * Threading: a lattice is re-entrant.  Its tables are read-only after the build; the buffers a filter call
 * writes live in workspaces handed out per call (reused in stream order, a second one is allocated when two
 * streams are in flight at once), so any number of host threads / streams may call phl_filter, phl_splat,
 * phl_blur, phl_slice on one handle concurrently.  phl_add_vertices and phl_destroy are not concurrent with
 * anything. */
int phl_reserve(phl_lattice *lat, int vd);
/* phl_reserve with options: PHL_RESERVE_STRIDED_IO also sizes the staging copies that channel-major (NCHW) views -- and
 * pixel-major rows whose stride or base address is off the 16-byte grid, from 128 channels on -- go through (channel
 * counts >= 128 that are not a multiple of 4 are always staged, at the width rounded up: sized without the flag),
 * PHL_RESERVE_EXACT prepares for PHL_FILTER_EXACT calls (builds the pixel-sorted lists). */
enum phl_reserve_flags { PHL_RESERVE_STRIDED_IO = 1, PHL_RESERVE_EXACT = 2 };
int phl_reserve_ex(phl_lattice *lat, int vd, unsigned reserve_flags);

/* ---- the hot path -----------------------------------------------------------------------
 * out = slice(blur(splat(src)))  == lattice.filter(src, ref) of the reference
 * (lattice.cpp:6-10 -> permutohedral.h:236-238, :260, :264-276).
 * src_dev/out_dev: [n][vd] fp32, element strides; out may not alias src. */
int phl_filter(phl_lattice *lat, const float *src_dev, int vd, int64_t src_row_stride,
               int64_t src_col_stride, float *out_dev, int64_t out_row_stride,
               int64_t out_col_stride, unsigned flags, phl_stream stream);

/* One-shot call with the reference's argument order (src first, ref second):
 * builds, filters, destroys -- what lattice.filter(src, ref) does on every call.  The lattice is built with
 * PHL_BUILD_REFERENCE_TABLE (results are the reference's at any size) unless flags has PHL_FILTER_CLEAN_TABLE. */
int phl_filter_once(const float *src_dev, int vd, int64_t src_row_stride, int64_t src_col_stride,
                    const float *ref_dev, int d, int64_t ref_row_stride, int64_t ref_col_stride,
                    int64_t n, float *out_dev, int64_t out_row_stride, int64_t out_col_stride,
                    unsigned flags, int device, phl_stream stream);

/* Backward of one filter call w.r.t. the FEATURES (and the source), replacing the body of
 * LatticeFilter.backward, crf/gaussian_matrix.py:435-468: for out = filter(src, ref) and an incoming gradient g [n][L],
 *     grad_ref[i][k] = -2 sum_l ( src_il f_ik (Wg)_il - src_il (W(g f_k))_il + g_il f_ik (Ws)_il - g_il (W(s f_k))_il )
 * (:463; W = this lattice's splat-blur-slice, f = ref) and, if grad_src != NULL, grad_src = W g (:446 -- the
 * operator is symmetric).  The reference materialises the 2L(1+d)-channel operand [g, g(x)ref, src, src(x)ref],
 * filters it and contracts the 2L(1+d)-channel result; here the products are formed on the weights inside the
 * splat and the contraction inside the slice, so HBM sees src and g twice, the wide vertex buffer
 * [M][(1+d)L] (two passes) and [n][d] -- never a wide per-pixel tensor.  ref must be the array the lattice was
 * built from.  src, g, grad_src: pixel-major rows (unit channel stride), 16-byte aligned, L % 4 == 0; d <= 7.
 * grad_ref: dense [n][d].  PHL_ERR_UNSUPPORTED for shapes outside that (callers then filter the wide operand). */
int phl_filter_grad(phl_lattice *lat, const float *src_dev, int64_t src_row_stride, const float *g_dev,
                    int64_t g_row_stride, int L, const float *ref_dev, int64_t ref_row_stride, int64_t ref_col_stride,
                    float *grad_ref_dev, float *grad_src_dev, int64_t grad_src_row_stride, phl_stream stream);

/* ---- stage-level entry points (profiling, roofline measurement, parity of intermediates) --
 * vert buffers are dense [M][vd] fp32 device arrays owned by the caller. */
/* value half of splat(): vert[v] = sum over (pixel,weight) of w*src[pixel]   (:454-455) */
int phl_splat(phl_lattice *lat, const float *src_dev, int vd, int64_t src_row_stride,
              float *vert_dev, unsigned flags, phl_stream stream);
/* one blur axis, Jacobi: dst[v] = 2*(1/4 src[n1] + 1/2 src[v] + 1/4 src[n2])   (:498-533) */
int phl_blur_axis(phl_lattice *lat, int axis, const float *vert_src_dev, float *vert_dst_dev, int vd,
                  phl_stream stream);
/* blur(): all d+1 axes in order 0..d (:486-548), ping-ponging between the caller's two [M][vd]
 * buffers (input in vert_a).  Consecutive axes are taken two per pass (same bits as two
 * phl_blur_axis calls, half the traffic over the vertex array).  *result_in_b = 1 if the result
 * ends up in vert_b, 0 if in vert_a. */
int phl_blur(phl_lattice *lat, float *vert_a_dev, float *vert_b_dev, int vd, int *result_in_b, phl_stream stream);
/* Row-band exchange helpers (no reference counterpart): out[r] = vert[idx[r]] and vert[idx[r]] += in[r]
 * for k rows of vd channels (vd % 4 == 0, 16-byte aligned); idx_dev: int64 on the device, DISTINCT
 * within one scatter call (no atomics: the sum order stays fixed).  Current device, given stream. */
int phl_gather_rows(const float *vert_dev, int vd, const int64_t *idx_dev, int64_t k, float *out_dev,
                    int64_t out_row_stride, phl_stream stream);
int phl_scatter_add_rows(float *vert_dev, int vd, const int64_t *idx_dev, int64_t k, const float *in_dev,
                         int64_t in_row_stride, phl_stream stream);
/* The chunk splat in two (or more) parts, for overlapping the row-band exchange with compute: run only the listed
 * pixel chunks, then complete exactly the listed vertex rows.  A rank first splats the chunks that touch its
 * boundary vertices (phl_chunks_touching) and completes those rows -- they can travel -- and then the interior
 * chunks and all other rows while the exchange is in flight.  partial_dev: caller-owned [phl_partial_rows()][vd]
 * scratch shared by the parts of one splat (rows shared between an early and a late chunk are summed by the late
 * part).  Every chunk must appear in exactly one part and every vertex row in exactly one part that runs after all
 * chunks touching it.  No reference counterpart. */
int64_t phl_num_chunks(const phl_lattice *lat);
int64_t phl_partial_rows(const phl_lattice *lat);
/* mask_host[c] = 1 iff pixel chunk c contributes to any of the k listed vertex rows (device int64 array) */
int phl_chunks_touching(phl_lattice *lat, const int64_t *rows_dev, int64_t k, int32_t *mask_host /* [phl_num_chunks] */,
                        phl_stream stream);
int phl_splat_part(phl_lattice *lat, const float *src_dev, int vd, int64_t src_row_stride, float *vert_dev,
                   float *partial_dev, const int32_t *chunks_dev, int64_t nchunks_sel, const int32_t *rows_dev,
                   int64_t nrows, phl_stream stream);
/* phl_splat_part that also fills the exchange's send buffer: listed row i is written to pack_dev[pack_pos_dev[i]]
 * (rows of pack_row_stride floats; pack_pos < 0: not packed) by the kernel that completes it, instead of a separate
 * gather afterwards.  src_dev may be NULL when no chunks are listed.  pack_pos_dev == NULL: plain phl_splat_part. */
int phl_splat_part_pack(phl_lattice *lat, const float *src_dev, int vd, int64_t src_row_stride, float *vert_dev,
                        float *partial_dev, const int32_t *chunks_dev, int64_t nchunks_sel, const int32_t *rows_dev,
                        int64_t nrows, const int32_t *pack_pos_dev, float *pack_dev, int64_t pack_row_stride,
                        phl_stream stream);
/* Row-band lattices: per blur axis a, the rows whose OUTPUT of axis a is read by anything later (the next axes' stencils,
 * finally slice) as three ascending, non-overlapping row ranges ranges[a][k] = {begin, end} (empty ranges allowed).  The
 * blur of phl_blur / phl_filter then computes only those rows (a pass over the axis pair (2p, 2p+1) computes the rows
 * named for axis 2p+1); the others keep stale values nobody reads.  A band carries ghost vertices 2-3 lattice steps deep
 * for the FIRST axes' stencils; later axes need fewer of them and slice none (phl/rowtile.py derives the sets from the
 * neighbour tables).  ranges == NULL: all rows again.  phl_add_vertices resets it.  Like phl_add_vertices it belongs to
 * the assembly of a band's lattice: not concurrent with filter calls on the handle.  No reference counterpart. */
int phl_set_blur_rows(phl_lattice *lat, const int64_t *ranges /* [d+1][3][2] */, int naxes);
/* slice(): out[p] = sum_i w_i * vert[v_i] / (1 + 2^-d)                        (:473-483) */
int phl_slice(phl_lattice *lat, const float *vert_dev, int vd, float *out_dev, int64_t out_row_stride,
              const float *sub_dev /* NULL or src to subtract */, int64_t sub_row_stride, unsigned flags,
              phl_stream stream);

/* ---- mean-field elementwise steps around the filter ----------------------
 * out[p,:] = softmax(-(E0[p,:] + G[p,:])) over the L labels of every pixel; G may be NULL.
 * Fuses `E = E_0 + (W@Q)@Mu; Q = softmax(-E, dim=1)` of mean_field_infer (crf/crf_module.py:49-52)
 * into one pass.  Rows are pixel-major with unit channel stride; row strides in elements. */
int phl_softmax_neg_add(const float *E0_dev, int64_t e0_row_stride, const float *G_dev, int64_t g_row_stride,
                        float *out_dev, int64_t out_row_stride, int64_t n, int L, phl_stream stream);
/* out[p,:] = softmax(-(E0[p,:] + X[p,:] @ Mu)): the whole non-lattice half of a mean-field iteration
 * (`E = E_0 + (W@Q)@Mu; Q = softmax(-E)`, crf/crf_module.py:51-52, with X = W@Q) in ONE kernel -- the
 * compatibility product on the fp32-input matrix cores (exact f32, as the reference computes), +E0 and the row
 * softmax applied to the accumulators, so G and E never exist in memory.
 * mu_t_dev is Mu TRANSPOSED and zero-padded to the tile width Lp = 32*ceil(L/32): a dense [Lp][Lp] array with
 * mu_t[c*Lp + k] = Mu[k][c] for c, k < L and 0 elsewhere (Charbonnier / Potts compatibilities are symmetric: their
 * own transpose).  Needs L % 4 == 0, L <= 256, all three row strides % 4 == 0 and 16-byte aligned E0 / X / out /
 * mu_t (else PHL_ERR_UNSUPPORTED: the caller keeps its GEMM + phl_softmax_neg_add).  Rows have unit channel stride.
 * flags: PHL_COMPAT_LOGITS writes -(E0 + X @ Mu) instead of its softmax (CRFasRNN returns the logits of the last
 * iteration, crf_module.py:103). */
enum phl_compat_flags { PHL_COMPAT_SOFTMAX = 0, PHL_COMPAT_LOGITS = 1 };
int phl_compat_softmax(const float *E0_dev, int64_t e0_row_stride, const float *X_dev, int64_t x_row_stride,
                       const float *mu_t_dev, float *out_dev, int64_t out_row_stride, int64_t n, int L, unsigned flags,
                       phl_stream stream);
/* The same step for 128 < L <= 256 (one 256-label tile, padded above L) on the bf16 matrix cores (sixteen times the f32
 * rate; it pays from L ~ 176 on, below that the f32 kernel is bound by its bytes as well) with both operands split
 * into three bf16 addends each -- x = h + m + l, eight significant bits apiece, exact -- and the six partial products
 * that matter (hH + hM + mH + hL + lH + mM; the dropped ones are below 2^-23 |x mu|).  Products of bf16 pairs are exact
 * in f32 and the matrix cores accumulate in f32: the result carries the rounding of an f32 dot product (measured next
 * to phl_compat_softmax against float64: tests/test_gpu_meanfield.py), at 3/8 of the matrix time, which makes the step
 * the streaming pass over E0, X and Q that its bytes say.  The compatibility matrix is PREPARED once per Mu:
 *   phl_compat_planes_bytes(L)   size of the prepared planes, 0 if the split kernel does not take this L
 *   phl_compat_prepare           mu_t_dev ([Lp][Lp], as for phl_compat_softmax) -> planes_dev (16-byte aligned)
 *   phl_compat_softmax_split     the step; same arguments and flags as phl_compat_softmax plus the planes (mu_t_dev
 *                                serves the last n % 128 rows, which run the f32 chain)
 * PHL_ERR_UNSUPPORTED for other L / alignments (the caller stays with phl_compat_softmax). */
size_t phl_compat_planes_bytes(int L);
int phl_compat_prepare(const float *mu_t_dev, int L, void *planes_dev, phl_stream stream);
int phl_compat_softmax_split(const float *E0_dev, int64_t e0_row_stride, const float *X_dev, int64_t x_row_stride,
                             const float *mu_t_dev, const void *planes_dev, float *out_dev, int64_t out_row_stride,
                             int64_t n, int L, unsigned flags, phl_stream stream);
/* The same step for compatibility matrices of the Potts family, Mu = alpha*J + beta*I (J all ones; the reference's
 * `potts` layer is alpha = 1, beta = -1, crf_module.py:55-64): X @ Mu = alpha*rowsum(X) + beta*X, so
 * out[p,:] = softmax(-(E0[p,:] + alpha*sum_c X[p,c] + beta*X[p,:])) is one streaming pass, no matrix product.
 * Needs L % 4 == 0, L <= 1024, row strides % 4 == 0 and 16-byte aligned E0 / X / out (else PHL_ERR_UNSUPPORTED).
 * flags: phl_compat_flags. */
int phl_uniform_compat_softmax(const float *E0_dev, int64_t e0_row_stride, const float *X_dev, int64_t x_row_stride,
                               float alpha, float beta, float *out_dev, int64_t out_row_stride, int64_t n, int L,
                               unsigned flags, phl_stream stream);
/* out[p] = sum_c Q[p,c]*labels[c] : the expected disparity `mf @ labels`
 * (Experiments/DenseCrf.ipynb cell 11). */
int phl_expected_value(const float *Q_dev, int64_t q_row_stride, const float *labels_dev, float *out_dev,
                       int64_t n, int L, phl_stream stream);

/* ---- caller side of the path: the unary cost volume E_0 ---------------------
 * disparity_badness(img1, img2, window_size, criterion), crf/depth.py:36-53, on the device:
 *   out[(y*w+x)*out_row_stride + k] = sum over the window x window neighbourhood (scipy 'reflect'
 *   borders on the cost array, :51-52) of  sum_ch criterion(img1[y,x,ch], img2[y,x-k,ch])  with
 *   img2 = 0 left of the image (:44-50), k = 0..max_disp-1.
 * img1/img2: [h][w][channels] fp32 device arrays (1..4 channels), window odd <= 17,
 * criterion 0 = AD (:26-27), 1 = SD (:24-25), 2 = nprod (:28-29).  The reference takes
 * max_disp = w // 6 (:40); here it is the caller's argument.  Output is pixel-major fp32, the
 * layout phl_filter reads. */
int phl_cost_volume(const float *img1_dev, const float *img2_dev, int h, int w, int channels, int max_disp,
                    int window, int criterion, float *out_dev, int64_t out_row_stride, phl_stream stream);

/* Plain float4 streaming copy dst <- src (n_floats % 4 == 0, 16-byte aligned): measures the
 * HBM read+write ceiling of the box that the roofline fractions are compared with. */
int phl_stream_copy(const float *src_dev, float *dst_dev, int64_t n_floats, phl_stream stream);
/* dst[r*dst_rs + c*dst_cs] = src[r*src_rs + c*src_cs] for rows x cols floats (LDS-tiled: both sides coalesced): the
 * transpose between the reference's NCHW tensors (BatchedAdjacency's channel-major [n, L] views,
 * crf/gaussian_matrix.py:345-349) and the pixel-major rows the kernels work on.  Current device, given stream. */
int phl_copy2d(const float *src_dev, int64_t src_row_stride, int64_t src_col_stride, float *dst_dev, int64_t dst_row_stride,
               int64_t dst_col_stride, int64_t rows, int cols, phl_stream stream);

/* Chunk ("tile") statistics of the LDS-staged path: out[0]=pixels per chunk, [1]=#chunks,
 * [2]=max local vertices per chunk, [3]=(chunk,vertex) slots S, [4]=slots of vertices fed by
 * several chunks, [5]=1 if the staged splat / [6]=slice would be chosen for this vd. */
int phl_tile_stats(const phl_lattice *lat, int vd, int64_t out[7]);

/* ---- introspection for parity tests (synchronous device->host copies) ---------------------
 * Vertex ids in these calls are the reference's: first-touch (insertion) order, the numbering its hash table
 * produces.  ROWS of the [M][vd] vertex buffers the stage-level calls take are in the library's internal
 * (locality) order instead: row_of_vertex[v] = row that holds first-touch vertex v; phl_add_vertices returns rows.
 * Ghost vertices come last in both numberings. */
int phl_get_vertex_order(phl_lattice *lat, int32_t *row_of_vertex_host /* [M] */);
/* Pixels in chunk order: chunk c of the LDS-staged kernels holds pixels pix_order[c*P ... (c+1)*P), P = phl_tile_stats()[0]. */
int phl_get_pixel_order(phl_lattice *lat, int32_t *pix_order_host /* [n] */);
int phl_get_keys(phl_lattice *lat, int16_t *keys_host /* [M][d] */);
int phl_get_replay(phl_lattice *lat, int32_t *vid_host /* [n][d+1] */, float *w_host /* [n][d+1] */);
int phl_get_neighbors(phl_lattice *lat, int32_t *nbr_host /* [d+1][M][2], -1 = absent */);
int phl_get_splat_lists(phl_lattice *lat, int32_t *ptr_host /* [M+1] */, int32_t *pixel_host /* [n(d+1)] */,
                        float *w_host /* [n(d+1)] */);

/* Test hook, host only (no device needed): the table replay behind PHL_BUILD_REFERENCE_TABLE on host arrays.
 * keys_clean [M][d] = distinct keys in first-touch order, cand_vid [N] = index into them for every
 * (pixel, remainder) candidate in order.  Returns the reference's key list (insertion order, duplicates
 * included), the vertex every candidate's lookup resolved to, the vertices its final table cannot reach and
 * the neighbour a doubling inside blur() decides (-2: no such doubling). */
int phl_debug_reference_table(const int16_t *keys_clean, const int32_t *cand_vid, int64_t M, int d, int64_t N,
                              int16_t *keys_ref_out, int64_t keys_ref_cap, int64_t *M_ref_out,
                              int32_t *cand_ref_vid_out, int32_t *hidden_out, int hidden_cap, int *n_hidden_out,
                              int *blur_first_nbr_out);

/* Test hook: the one assumption of the analytic table replay -- no tracked key's probe path in the reference's table
 * (linear probing, permutohedral.h:84-106, capacity `cap` = a power of two >= 2^15) runs past the last slot -- checked
 * for a caller-made key set on the device (on_device = 1: the kernels the build uses) or on the host (0).
 * keys_clean [n_clean][d]; extra_clean / stale_clean / check index into it (entries filed a second time under `cap`,
 * entries filed under cap / 2, keys whose paths are asked about).  *result_out = 1 if none wraps. */
int phl_debug_probe_paths(const int16_t *keys_clean, int64_t n_clean, int d, const int32_t *extra_clean, int n_extra,
                          const int32_t *stale_clean, int n_stale, uint64_t cap, const int32_t *check, int n_check,
                          int on_device, int *result_out);

/* Test hook: dst <- src (n_floats % 4 == 0, 16-byte aligned) copied `repeat` times by `workgroups` workgroups of 256 threads:
 * a long-running kernel with the footprint of a point-to-point transfer (a few wave slots, next to no HBM bandwidth), for
 * rehearsing on one GPU how much of an exchange of a given duration a schedule hides (tools/band_time.py wire variants). */
int phl_debug_slow_copy(const float *src_dev, float *dst_dev, int64_t n_floats, int workgroups, int repeat, phl_stream stream);
/* Test hook: side streams the calling thread holds for the reference-table build (one per device the thread has
 * built such a lattice on; building on devices A, B, A, B ... must not create more than two). */
int phl_debug_side_streams(void);

#ifdef __cplusplus
}
#endif
#endif /* PHL_H */
