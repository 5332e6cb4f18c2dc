// phl_meanfield.hip -- the elementwise half of one mean-field iteration, fused.
//
// Reference (crf/crf_module.py:49-52):   E = E_0 + (W@Q)@Mu ;  Q = softmax(-E, dim=1)
// torch runs that as add, neg, and a 2-3 pass softmax: ~7 sweeps over an [n, L] tensor that is
// 3.2 GB at 2048x1536x256.  Here it is one read of E_0, one read of G = (W@Q)@Mu, one write of Q:
// a wavefront owns a row, every lane keeps its L/64 values in registers between the max, the
// sum and the normalisation.  HBM-bound elementwise work, no MFMA.
//   k_softmax_neg_add   Q[p,:] = softmax(-(E0[p,:] + G[p,:]))          (G optional)
//   k_expected_value    out[p] = sum_c Q[p,c] * labels[c]                (Experiments/DenseCrf.ipynb cell 11)
#include <math.h>

#include "phl_internal.h"

namespace {

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// NV = float4 per lane held in registers: rows up to NV*256 channels in one pass
template <int NV>
__global__ __launch_bounds__(256) void k_softmax_neg_add(const float *__restrict__ E0, int64_t e_rs,
                                                         const float *__restrict__ G, int64_t g_rs,
                                                         float *__restrict__ out, int64_t o_rs, int64_t n, int L)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t p = wave; p < n; p += nw) {
        float4 x[NV];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int c = (j * 64 + lane) * 4;
            if (c < L) {
                float4 e = *reinterpret_cast<const float4 *>(E0 + p * e_rs + c);
                if (G) {
                    const float4 g = *reinterpret_cast<const float4 *>(G + p * g_rs + c);
                    e.x += g.x; e.y += g.y; e.z += g.z; e.w += g.w;
                }
                x[j] = make_float4(-e.x, -e.y, -e.z, -e.w);
                m = fmaxf(fmaxf(m, fmaxf(x[j].x, x[j].y)), fmaxf(x[j].z, x[j].w));
            }
        }
        m = wave_max(m);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int c = (j * 64 + lane) * 4;
            if (c < L) {
                x[j] = make_float4(expf(x[j].x - m), expf(x[j].y - m), expf(x[j].z - m), expf(x[j].w - m));
                s += (x[j].x + x[j].y) + (x[j].z + x[j].w);
            }
        }
        s = wave_sum(s);
        const float inv = 1.0f / s;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int c = (j * 64 + lane) * 4;
            if (c < L)
                *reinterpret_cast<float4 *>(out + p * o_rs + c) = make_float4(x[j].x * inv, x[j].y * inv, x[j].z * inv, x[j].w * inv);
        }
    }
}

// any L / alignment: three passes over the row, scalar accesses
__global__ __launch_bounds__(256) void k_softmax_neg_add_generic(const float *__restrict__ E0, int64_t e_rs,
                                                                 const float *__restrict__ G, int64_t g_rs,
                                                                 float *__restrict__ out, int64_t o_rs, int64_t n, int L)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t p = wave; p < n; p += nw) {
        float m = -INFINITY;
        for (int c = lane; c < L; c += 64) m = fmaxf(m, -(E0[p * e_rs + c] + (G ? G[p * g_rs + c] : 0.f)));
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < L; c += 64) s += expf(-(E0[p * e_rs + c] + (G ? G[p * g_rs + c] : 0.f)) - m);
        s = wave_sum(s);
        const float inv = 1.0f / s;
        for (int c = lane; c < L; c += 64) out[p * o_rs + c] = expf(-(E0[p * e_rs + c] + (G ? G[p * g_rs + c] : 0.f)) - m) * inv;
    }
}

__global__ __launch_bounds__(256) void k_expected_value(const float *__restrict__ Q, int64_t q_rs,
                                                        const float *__restrict__ labels, float *__restrict__ out,
                                                        int64_t n, int L)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t p = wave; p < n; p += nw) {
        float s = 0.f;
        for (int c = lane; c < L; c += 64) s += Q[p * q_rs + c] * labels[c];
        s = wave_sum(s);
        if (lane == 0) out[p] = s;
    }
}

// float4 streaming copy: the measured HBM ceiling the roofline fractions are quoted against.  Form = the best of
// the sweep in tools/copy_probe.hip on this pool's boxes (grid sizes 2 Ki .. 64 Ki blocks, 4 or 8 loads in
// flight, temporal / non-temporal): many short workgroups (64 Ki blocks of 256 threads, 4 loads in flight each)
// with non-temporal loads and stores, 5.2 TB/s R+W on a 3.2 GB volume against 4.6 for a 2 Ki-block grid-stride
// loop and 4.6 for hipMemcpyAsync.
typedef float vf4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream_copy(const vf4 *__restrict__ src, vf4 *__restrict__ dst, int64_t n4)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const vf4 a = __builtin_nontemporal_load(&src[i]), b = __builtin_nontemporal_load(&src[i + stride]),
                  c = __builtin_nontemporal_load(&src[i + 2 * stride]), d = __builtin_nontemporal_load(&src[i + 3 * stride]);
        __builtin_nontemporal_store(a, &dst[i]);
        __builtin_nontemporal_store(b, &dst[i + stride]);
        __builtin_nontemporal_store(c, &dst[i + 2 * stride]);
        __builtin_nontemporal_store(d, &dst[i + 3 * stride]);
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}


// ------------------------------------------------------------------------------------------------------------
// k_compat_softmax: out[p,:] = softmax(-(E0[p,:] + X[p,:] @ Mu)) -- everything of a mean-field iteration that is
// not the lattice filter (crf/crf_module.py:51-52 with X = W@Q), in ONE kernel: the [n,L]x[L,L] compatibility
// product on the fp32-input matrix cores (v_mfma_f32_32x32x2_f32: exact f32, bitwise an fmaf chain in k order;
// gfx950 has no xf32/TF32 path and the reference computes in f32), with +E0, negate and the row softmax done on
// the accumulators.  Neither G = X@Mu nor E ever exists in HBM: the iteration's non-lattice half moves 3 x [n,L]
// (read X, read E0, write Q) instead of the 7 x [n,L] of GEMM + fused softmax, and is MFMA-bound
// (2 n L^2 flop at ~155 TF f32: 2.7 ms for 2048x1536x256).
//
// Tiling.  L = 32*NT <= 256, so a pixel's whole label row fits one workgroup: 256 threads = 4 waves, a wave owns
// 32 pixels x all L labels = NT accumulator tiles of 32x32 (16 VGPRs each, 128 at L = 256), two workgroups per CU
// so that one wave's epilogue (exp, stores) runs under its SIMD partner's MFMAs.  K = L is walked in chunks of
// 32: Mu's chunk (32 k x L labels, 32 KiB) is shared by the four waves through LDS, double buffered; X comes
// straight from global memory (each value is used by one wave only).
//
// Operand maps (32x32x2: lane l = (i = l&31, h = l>>5) supplies A[i][k-slot h] and B[k-slot h][j = i]).  The k
// order of a contraction is free as long as A and B agree, so a lane loads 16 B = X[row i][8q+4h .. 8q+4h+3]
// and the four values feed MFMAs u = 0..3 of group q: MFMA (q,u) contracts k = 8q+u (lower half-wave) and
// k = 8q+4+u (upper).  B must follow: lane (j,h) takes Mu[8q+4h+u][32t+j], u = 0..3 -- 16 contiguous bytes of
// the TRANSPOSED compatibility matrix, which is what the caller passes (MuT[c][k]) and what the LDS image
// holds: Bt[label][k], 128-byte rows whose 16-byte slots are XOR-swizzled (see the chunk loader).
// 16 bytes global -> LDS without a register in between (global_load_lds_dwordx4): the LDS address is the
// wave-uniform `l` plus lane*16, the global address is per lane
//
// The DMA is issued through inline assembly, NOT __builtin_amdgcn_global_load_lds: with the builtin the compiler
// knows an LDS write is in flight, cannot tell it from the buffer the ds_reads below use, and puts
// `s_waitcnt vmcnt(0)` in front of the first ds_read after every prefetch -- each chunk then waits out a full
// memory round trip with its MFMAs idle (measured: 60 us per tile phase against 28 us of MFMA work).  The
// kernel's own protocol makes the wait unnecessary (a buffer is only read behind the barrier that follows its
// DMA), so the kernel waits itself: dma_drain() before each barrier.
// Addressing: scalar base (SGPR pair) + 32-bit per-lane byte offset, so that a whole chunk's DMAs share two VGPRs.
__device__ __forceinline__ void glds16(const float *sbase, unsigned voff, unsigned lds_addr)
{
    unsigned keep;                              // m0 is the compiler's: hand it back as found
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr));
}
__device__ __forceinline__ void dma_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// VALU helpers of the epilogue.  v_mfma_f32_32x32x2_f32 and ordinary VALU instructions do NOT overlap on gfx950
// (tools/mfma_probe.hip: every v_fma slipped between two MFMAs costs its own issue time plus a ~10-cycle bubble,
// SQ_VALU_MFMA_COEXEC_CYCLES reads 0), so every epilogue instruction is paid for in matrix-pipe time: minima
// without the compiler's NaN canonicalisation (v_max x,x before every v_min), cross-lane steps as DPP modifiers
// of the min / add itself instead of ds_bpermute round trips.
__device__ __forceinline__ float vmin3(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// (s_nop 1: a DPP source needs two wait states behind the VALU write of that register, and the compiler's hazard
// recogniser does not look inside inline assembly)
#define PHL_DPP_ROR(op, x, n)                                                                                             \
    asm("s_nop 1\n\t" op " %0, %1, %1 row_ror:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(x) : "v"(x))
// v_permlane16_swap_b32 a, b: the odd 16-lane rows of a trade places with the even rows of b.  Inline assembly
// because __builtin_amdgcn_permlane16_swap hands back its FIRST result twice in this compiler (ROCm 7.2; checked
// with tools/dpp_probe.hip).
#define PHL_ROW_SWAP(a, b) asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b))
// all-reduce of 16 values at once over the 32 lanes of a half-wave (two DPP rows of 16): rotations inside the row,
// then the row swap.  Step-major, so that consecutive instructions are independent (a DPP operand must not have
// been written by the instruction just before it).
__device__ __forceinline__ void half_wave_min16(float (&x)[16])
{
#pragma unroll
    for (int r = 0; r < 16; r++) PHL_DPP_ROR("v_min_f32_dpp", x[r], 8);
#pragma unroll
    for (int r = 0; r < 16; r++) PHL_DPP_ROR("v_min_f32_dpp", x[r], 4);
#pragma unroll
    for (int r = 0; r < 16; r++) PHL_DPP_ROR("v_min_f32_dpp", x[r], 2);
#pragma unroll
    for (int r = 0; r < 16; r++) PHL_DPP_ROR("v_min_f32_dpp", x[r], 1);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        float a = x[r], b = x[r];       // -> (row0,row0,row2,row2), (row1,row1,row3,row3)
        PHL_ROW_SWAP(a, b);
        asm("v_min_f32 %0, %1, %2" : "=v"(x[r]) : "v"(a), "v"(b));
    }
}
__device__ __forceinline__ void half_wave_sum16(float (&x)[16])
{
#pragma unroll
    for (int r = 0; r < 16; r++) PHL_DPP_ROR("v_add_f32_dpp", x[r], 8);
#pragma unroll
    for (int r = 0; r < 16; r++) PHL_DPP_ROR("v_add_f32_dpp", x[r], 4);
#pragma unroll
    for (int r = 0; r < 16; r++) PHL_DPP_ROR("v_add_f32_dpp", x[r], 2);
#pragma unroll
    for (int r = 0; r < 16; r++) PHL_DPP_ROR("v_add_f32_dpp", x[r], 1);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        float a = x[r], b = x[r];
        PHL_ROW_SWAP(a, b);
        x[r] = a + b;
    }
}

#ifdef PHL_COMPAT_TIMELINE
// debug build only (make EXTRA=-DPHL_COMPAT_TIMELINE): per wave group, 100 MHz stamps of the first 16 phase starts
__device__ unsigned long long *g_cs_timeline;
__device__ int g_cs_exp;
#define CS_EXP(bit) (g_cs_exp & (bit))
#define CS_STAMP(p) do { if (g_cs_timeline && lane == 0 && w4 == 0 && (p) < 16) g_cs_timeline[((size_t)blockIdx.x * 2 + grp) * 40 + (p)] = wall_clock64(); } while (0)
#define CS_ARRIVE(k) do { if (g_cs_timeline && lane == 0 && w4 == 0 && it == 2) g_cs_timeline[((size_t)blockIdx.x * 2 + grp) * 40 + 16 + (k)] = wall_clock64(); } while (0)
#else
#define CS_STAMP(p) do { } while (0)
#define CS_EXP(bit) 0
#define CS_ARRIVE(k) do { } while (0)
#endif

template <int NT, bool LOGITS, bool PAD>
__global__ __launch_bounds__(512) void k_compat_softmax(const float *__restrict__ E0, int64_t e_rs,
                                                        const float *__restrict__ X, int64_t x_rs,
                                                        const float *__restrict__ MuT, float *__restrict__ out,
                                                        int64_t o_rs, int64_t n, int Lr)
{
    // Lr = the real label count (a multiple of 4, <= L): columns Lr..L-1 are padding -- MuT is zero there, E0 reads
    // as +inf (so exp gives 0 and the row minimum ignores them), X reads as 0, nothing is stored.
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int L = 32 * NT;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 x [L labels][32 k], 16-byte slots XOR-swizzled
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, w4 = wave & 3;     // wave group (0/1) and wave within it
    const int i = lane & 31, h = lane >> 5;
    const int64_t ntiles = (n + 127) / 128;
    const int64_t G2 = 2 * (int64_t)gridDim.x;

    // Chunk loader (LDS-DMA, no staging registers): a chunk is L labels x 8 slots of 16 B; one wave-instruction
    // fills 64 consecutive slots = 8 labels.  LDS stays linear (that is all the DMA can write); the bank
    // swizzle lives in WHICH 16 bytes a lane fetches: slot s of label r holds k-part s ^ ((r >> 1) & 7), and the
    // reads below apply the same XOR (the 16 lanes a ds_read_b128 serves at a time then hit 16 different
    // 16-byte bank groups).  The per-lane part of the address only depends on the parity of the
    // wave-instruction's index gi = w4*NT + r (label = 8 gi + lane/8, so (label >> 1) & 7 = (4 (gi & 1) + lane/16) & 7):
    // two byte offsets serve all of them.
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds;
    const unsigned voff0 = ((lane >> 3) * L + 4 * ((lane & 7) ^ (lane >> 4))) * 4;
    const unsigned voff1 = ((lane >> 3) * L + 4 * ((lane & 7) ^ (4 + (lane >> 4)))) * 4;
    auto load_mu = [&](int kc, int buf) {        // issued by the four waves of ONE group
#pragma unroll
        for (int r = 0; r < NT; r++) {
            const int gi = w4 * NT + r;          // wave-instruction index: 64 slots = 8 labels
            glds16(MuT + (int64_t)gi * 8 * L + 32 * kc, (gi & 1) ? voff1 : voff0, lds_base + (buf * (L * 32) + gi * 256) * 4);
        }
    };
    auto x_row = [&](int64_t tile) {             // this lane's X row of a tile (clamped: loads stay in bounds)
        return X + min(tile * 128 + w4 * 32 + i, n - 1) * x_rs + 4 * h;
    };
    // E0 / out addressing: a register r of the C/D map is pixel row rr(r) + 4h of the wave's 32, label 32t + lane&31:
    // wave-uniform row pointer (scalar registers) + one per-lane byte offset for the whole kernel + 128 t -- no
    // vector address arithmetic.  Rows beyond n (last tile only) are masked per lane, never clamped.
    // Padding is applied where a value is CONSUMED, not where it is loaded (a select on a fresh load would make the
    // wave wait for it): the last label tile's loads are clamped into the row, pad_e0() / pad_x() overwrite the
    // padding afterwards.
    const int h4 = 4 * h;
    const unsigned lo_e = (unsigned)(h4 * e_rs + i) * 4u, lo_o = (unsigned)(h4 * o_rs + i) * 4u;
    const unsigned lo_e_last = PAD ? (unsigned)(h4 * e_rs + min(32 * (NT - 1) + i, Lr - 1) - 32 * (NT - 1)) * 4u : lo_e;
    const bool lane_stores_last = !PAD || 32 * (NT - 1) + i < Lr;
    auto rr = [](int r) { return (r & 3) + 8 * (r >> 2); };
    auto rows_left = [&](int64_t tile) { return (int)min((int64_t)32, n - (tile * 128 + w4 * 32)); };   // of this wave (<= 0: none)
    auto e0_load_row = [&](f32x16 *acc, int r, const float *wave_rows, int left) {       // wave_rows: E0 row of the wave's first pixel
        if (rr(r) + h4 < left) {
            const char *p = reinterpret_cast<const char *>(wave_rows + rr(r) * e_rs);
#pragma unroll
            for (int t = 0; t < NT; t++)
                acc[t][r] = *reinterpret_cast<const float *>(p + 128 * t + ((PAD && t == NT - 1) ? lo_e_last : lo_e));
        }
    };
    auto x_at = [&](const float *xrow, int k0) {      // 4 contraction values from column k0 + 4h (clamped under PAD)
        return *reinterpret_cast<const float4 *>(xrow + (PAD ? min(k0, Lr - 4 - 4 * h) : k0));
    };
    auto pad_e0 = [&](f32x16 &last) {                 // only the last label tile holds padding (L - Lr < 32)
        if (PAD && 32 * (NT - 1) + i >= Lr) {
#pragma unroll
            for (int r = 0; r < 16; r++) last[r] = INFINITY;
        }
    };
    auto pad_x = [&](float4 &v, int k0) {
        if (PAD && k0 + 4 * h >= Lr) v = make_float4(0.f, 0.f, 0.f, 0.f);
    };

    // Tiles of this group: 2 (b + k gridDim) + grp, k = 0, 1, ...  Every wave alternates the MFMA half of a tile (NT
    // slots, one per K chunk) with its epilogue half (NT slots as well); a slot ends in the workgroup barrier, and
    // group 1 starts one half late (NT bare barriers), group 0 ends with them: whenever one group is on the matrix
    // cores, the other one -- its SIMD partner -- is storing Q and fetching E0.  Both groups run the same number of
    // iterations (group 0 never has fewer tiles); a group without a tile left still keeps the LDS ring fed.
    const int64_t iters = (ntiles - 2 * (int64_t)blockIdx.x + G2 - 1) / G2;          // tiles 2b, 2b + G2, ... < ntiles
    int64_t tile = 2 * (int64_t)blockIdx.x + grp;
    bool valid = tile < ntiles;                  // wave-uniform
    f32x16 acc[NT];
    float4 a_cur[4], a_nxt[4];
    if (valid) {
        const float *erows = E0 + (tile * 128 + w4 * 32) * e_rs;
        const int left = rows_left(tile);
#pragma unroll
        for (int r = 0; r < 16; r++) e0_load_row(acc, r, erows, left);
        const float *xrow = x_row(tile);
#pragma unroll
        for (int q = 0; q < 4; q++) a_nxt[q] = x_at(xrow, 8 * q);
    }
    int slot = 0;                                // chunks since the start: chunk `slot` sits in LDS buffer slot & 1
    if (grp == 0) {
        load_mu(0, 0);
        dma_drain();
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) {
        for (int s = 0; s < NT; s++, slot++) __builtin_amdgcn_s_barrier();
    }
    const int sw = (i >> 1) & 7;                 // the swizzle of the labels this lane reads (32t + i)

    for (int64_t it = 0; it < iters; it++) {
        CS_STAMP(2 * it);
        // =========== MFMA half: E = E0 + X @ Mu on `tile`; the group in this half feeds the LDS ring ===========
        {
            const float *xrow = x_row(valid ? tile : 0);
            pad_e0(acc[NT - 1]);
            for (int kc = 0; kc < NT; kc++, slot++) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    a_cur[q] = a_nxt[q];
                    if (kc == NT - 1) pad_x(a_cur[q], 32 * kc + 8 * q);
                }
                // next chunk (of this tile, or chunk 0 for the other group's tile): its buffer was last read in the
                // previous slot, behind the barrier
                load_mu(kc + 1 < NT ? kc + 1 : 0, (slot + 1) & 1);
                if (valid && kc + 1 < NT && !CS_EXP(8)) {
#pragma unroll
                    for (int q = 0; q < 4; q++) a_nxt[q] = x_at(xrow, 32 * (kc + 1) + 8 * q);
                }
                // keep the prefetch HERE: left alone, the scheduler sinks these loads to the end of the chunk
                __builtin_amdgcn_sched_barrier(0);
                if (valid) {
                    const float *bbase = lds + (slot & 1) * (L * 32) + i * 32;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        float4 b[NT];
                        const int sl = ((2 * q + h) ^ sw) * 4;
#pragma unroll
                        for (int t = 0; t < NT; t++) b[t] = *reinterpret_cast<const float4 *>(bbase + t * 32 * 32 + sl);
#pragma unroll
                        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].x, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].y, b[t].y, acc[t], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].z, b[t].z, acc[t], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].w, b[t].w, acc[t], 0, 0, 0);
                    }
                }
                CS_ARRIVE(16 + kc);              // (before the drain)
                dma_drain();                     // the next chunk's DMA: issued a whole chunk of MFMAs ago
                CS_ARRIVE(kc);
                __builtin_amdgcn_s_barrier();
            }
        }
        CS_STAMP(2 * it + 1);
        // =========== epilogue half: softmax and store of `tile`, E0 / X of the group's next tile in ===============
        // One piece per slot of the other group's MFMA half: the 16 pixel rows a lane holds go out over the first
        // NT-1 slots, the last slot only lets the loads land.  Nothing here waits on memory: a stored row's registers
        // take the next tile's E0 at once (the MFMA's C input, so E = E0 + X @ Mu comes out of the matrix pipe).
        // C/D map: column = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5): for one register the 32 lanes of a
        // half-wave hold 32 consecutive labels of one pixel (128 B per access).
        {
            const int64_t nxt = tile + G2;
            const bool has_next = nxt < ntiles;  // uniform
            float *orows = out + (tile * 128 + w4 * 32) * o_rs;                        // wave-uniform
            const float *erows = E0 + ((has_next ? nxt : tile) * 128 + w4 * 32) * e_rs;
            const int left_o = valid ? rows_left(tile) : 0, left_e = has_next ? rows_left(nxt) : 0;
            float scale[16];
            if (valid && !LOGITS && !CS_EXP(4)) {
                float m[16];
#pragma unroll
                for (int r = 0; r < 16; r++) {   // softmax(-E): shift by the row MINIMUM of E
                    m[r] = acc[0][r];
#pragma unroll
                    for (int t = 1; t + 1 < NT; t += 2) m[r] = vmin3(m[r], acc[t][r], acc[t + 1][r]);
                    if (NT % 2 == 0) m[r] = vmin3(m[r], acc[NT - 1][r], acc[NT - 1][r]);
                }
                // the 16 rows are reduced TOGETHER: sixteen independent chains per step
                half_wave_min16(m);
#pragma unroll
                for (int r = 0; r < 16; r++) m[r] *= 1.4426950408889634f;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    float s0 = 0.f, s1 = 0.f;
#pragma unroll
                    for (int t = 0; t < NT; t++) {   // exp(-(E - min)) = exp2(min log2e - E log2e)
                        acc[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[t][r], -1.4426950408889634f, m[r]));
                        if (t & 1) s1 += acc[t][r]; else s0 += acc[t][r];
                    }
                    scale[r] = s0 + s1;
                }
                half_wave_sum16(scale);
#pragma unroll
                for (int r = 0; r < 16; r++) scale[r] = __builtin_amdgcn_rcpf(scale[r]);
            }
#pragma unroll
            for (int s = 0; s < NT; s++) {
                constexpr int D = NT > 1 ? NT - 1 : 1;
                const int r_lo = s < D ? (16 * s + D - 1) / D : 16, r_hi = s + 1 < D ? (16 * (s + 1) + D - 1) / D : 16;
#pragma unroll
                for (int r = 0; r < 16; r++) {   // (constant trip counts: both loops unroll and the test folds)
                    if (r < r_lo || r >= r_hi) continue;
                    if (rr(r) + h4 < left_o && !CS_EXP(1)) {
                        const float sc = LOGITS ? -1.0f : scale[r];     // CRFasRNN returns -E (crf_module.py:103)
                        char *p = reinterpret_cast<char *>(orows + rr(r) * o_rs);
#pragma unroll
                        for (int t = 0; t < NT; t++)
                            if (!PAD || t < NT - 1 || lane_stores_last) *reinterpret_cast<float *>(p + 128 * t + lo_o) = acc[t][r] * sc;
                    }
                    // this row's registers are free: the next tile's E0 goes in
                    if (!CS_EXP(2)) e0_load_row(acc, r, erows, left_e);
                }
                if (s == NT - 1 && has_next) {
                    const float *xn = x_row(nxt);
#pragma unroll
                    for (int q = 0; q < 4; q++) a_nxt[q] = x_at(xn, 8 * q);
                }
                CS_ARRIVE(NT + s);
                __builtin_amdgcn_s_barrier();
                slot++;
            }
            tile = nxt;
            valid = has_next;
        }
    }
    if (grp == 0) {
        for (int s = 0; s < NT; s++) __builtin_amdgcn_s_barrier();
    }
}

inline unsigned rows_grid(int64_t n)
{
    int64_t b = (n + 3) / 4;
    if (b > 256 * 8) b = 256 * 8;
    return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" {

int phl_softmax_neg_add(const float *E0, int64_t e_rs, const float *G, int64_t g_rs, float *out, int64_t o_rs, int64_t n,
                        int L, phl_stream stream)
{
    if (n < 0 || L < 1 || (n > 0 && (!E0 || !out))) { phl_set_error("phl_softmax_neg_add: bad arguments"); return PHL_ERR_INVALID; }
    if (n == 0) return PHL_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool v4 = L % 4 == 0 && e_rs % 4 == 0 && o_rs % 4 == 0 && (!G || g_rs % 4 == 0) &&
                    ((reinterpret_cast<uintptr_t>(E0) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(G)) & 15) == 0;
    const unsigned grid = rows_grid(n);
    if (v4 && L <= 256) k_softmax_neg_add<1><<<dim3(grid), dim3(256), 0, st>>>(E0, e_rs, G, g_rs, out, o_rs, n, L);
    else if (v4 && L <= 512) k_softmax_neg_add<2><<<dim3(grid), dim3(256), 0, st>>>(E0, e_rs, G, g_rs, out, o_rs, n, L);
    else if (v4 && L <= 1024) k_softmax_neg_add<4><<<dim3(grid), dim3(256), 0, st>>>(E0, e_rs, G, g_rs, out, o_rs, n, L);
    else k_softmax_neg_add_generic<<<dim3(grid), dim3(256), 0, st>>>(E0, e_rs, G, g_rs, out, o_rs, n, L);
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}

int phl_compat_softmax(const float *E0, int64_t e_rs, const float *X, int64_t x_rs, const float *MuT, float *out, int64_t o_rs,
                       int64_t n, int L, unsigned flags, phl_stream stream)
{
    if (n < 0 || L < 1 || (n > 0 && (!E0 || !X || !MuT || !out))) { phl_set_error("phl_compat_softmax: bad arguments"); return PHL_ERR_INVALID; }
    if (n == 0) return PHL_OK;
    if (L % 4 || L > 256 || x_rs % 4 || e_rs % 4 || o_rs % 4 ||
        ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(MuT) | reinterpret_cast<uintptr_t>(E0) | reinterpret_cast<uintptr_t>(out)) & 15)) {
        phl_set_error("phl_compat_softmax: needs L %% 4 == 0, L <= 256 and 16-byte aligned E0 / X / out rows (L=%d)", L);
        return PHL_ERR_UNSUPPORTED;
    }
    const int Lp = (L + 31) / 32 * 32;       // the tile width: mu_t is [Lp][Lp], zero beyond L
    const bool pad = Lp != L;
    hipStream_t st = (hipStream_t)stream;
    // persistent workgroups, one per CU (256 CUs): each walks pairs of 128-pixel tiles (one per wave group)
    const int64_t npairs = ((n + 127) / 128 + 1) / 2;
    const unsigned grid = (unsigned)(npairs < 256 ? npairs : 256);
    const size_t lds = (size_t)2 * Lp * 32 * sizeof(float);
    const bool logits = (flags & PHL_COMPAT_LOGITS) != 0;     // LDS is 2*L*128 B <= 64 KiB: no attribute needed
#ifdef PHL_COMPAT_TIMELINE
    static unsigned long long *tl_buf = nullptr;
    if (const char *path = getenv("PHL_COMPAT_TIMELINE")) {
        if (tl_buf) {                            // dump the previous launch
            hipDeviceSynchronize();
            std::vector<unsigned long long> hbuf(512 * 40);
            hipMemcpy(hbuf.data(), tl_buf, hbuf.size() * 8, hipMemcpyDeviceToHost);
            if (FILE *f = fopen(path, "wb")) { fwrite(hbuf.data(), 8, hbuf.size(), f); fclose(f); }
        } else {
            hipMalloc(&tl_buf, 512 * 40 * 8);
            hipMemcpyToSymbol(HIP_SYMBOL(g_cs_timeline), &tl_buf, sizeof(tl_buf));
            const int exp_mode = getenv("PHL_CS_EXP") ? atoi(getenv("PHL_CS_EXP")) : 0;
            hipMemcpyToSymbol(HIP_SYMBOL(g_cs_exp), &exp_mode, sizeof(exp_mode));
        }
        hipMemset(tl_buf, 0, 512 * 40 * 8);
    }
#endif
#define PHL_CS(NT_)                                                                                                       \
    case NT_:                                                                                                             \
        if (logits && pad) k_compat_softmax<NT_, true, true><<<dim3(grid), dim3(512), lds, st>>>(E0, e_rs, X, x_rs, MuT, out, o_rs, n, L);        \
        else if (logits) k_compat_softmax<NT_, true, false><<<dim3(grid), dim3(512), lds, st>>>(E0, e_rs, X, x_rs, MuT, out, o_rs, n, L);         \
        else if (pad) k_compat_softmax<NT_, false, true><<<dim3(grid), dim3(512), lds, st>>>(E0, e_rs, X, x_rs, MuT, out, o_rs, n, L);            \
        else k_compat_softmax<NT_, false, false><<<dim3(grid), dim3(512), lds, st>>>(E0, e_rs, X, x_rs, MuT, out, o_rs, n, L);                    \
        break;
    switch (Lp / 32) {
        PHL_CS(1) PHL_CS(2) PHL_CS(3) PHL_CS(4) PHL_CS(5) PHL_CS(6) PHL_CS(7) PHL_CS(8)
    }
#undef PHL_CS
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}

int phl_stream_copy(const float *src, float *dst, int64_t n_floats, phl_stream stream)
{
    if (n_floats < 0 || n_floats % 4 || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15)) { phl_set_error("phl_stream_copy: needs 16-byte aligned buffers, n % 4 == 0"); return PHL_ERR_INVALID; }
    if (n_floats == 0) return PHL_OK;
    int64_t blocks = (n_floats / 4 + 256 * 4 - 1) / (256 * 4);
    if (blocks > 65536) blocks = 65536;
    k_stream_copy<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(reinterpret_cast<const vf4 *>(src), reinterpret_cast<vf4 *>(dst), n_floats / 4);
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}

int phl_expected_value(const float *Q, int64_t q_rs, const float *labels, float *out, int64_t n, int L, phl_stream stream)
{
    if (n < 0 || L < 1 || (n > 0 && (!Q || !labels || !out))) { phl_set_error("phl_expected_value: bad arguments"); return PHL_ERR_INVALID; }
    if (n == 0) return PHL_OK;
    k_expected_value<<<dim3(rows_grid(n)), dim3(256), 0, (hipStream_t)stream>>>(Q, q_rs, labels, out, n, L);
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}

}  // extern "C"
