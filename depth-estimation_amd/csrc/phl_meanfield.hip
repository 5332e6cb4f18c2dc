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
#include <type_traits>

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
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
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
#define CS_STAMP(p) do { if (g_cs_timeline && lane == 0 && w4 == 0 && (p) < 16) g_cs_timeline[((size_t)blockIdx.x * 2 + grp) * 64 + (p)] = wall_clock64(); } while (0)
#ifdef PHL_CS_FINE
#define CS_ARRIVE(k) do { if (g_cs_timeline && lane == 0 && w4 == 0 && tile / G2 == 2) g_cs_timeline[((size_t)blockIdx.x * 2 + grp) * 64 + 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define CS_ARRIVE(k) do { } while (0)
#endif
#else
#define CS_STAMP(p) do { } while (0)
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
    const int64_t ntiles = n / 128;               // WHOLE tiles only: the launcher hands the last n % 128 pixels to k_compat_tail
    const int64_t G2 = 2 * (int64_t)gridDim.x;

    // RULE OF THIS KERNEL: no vector-ALU instruction inside the MFMA stream.  On gfx950 a VALU instruction between two
    // v_mfma_f32_32x32x2_f32 does not overlap with them (tools/mfma_probe.hip: 16 register copies per 128 MFMAs cost
    // 3 cycles per MFMA), so every address below is "scalar base + per-lane offset fixed for the whole kernel +
    // immediate": the scalar unit does the arithmetic.

    // Chunk loader (LDS-DMA, no staging registers): a chunk is L labels x 8 slots of 16 B; one wave-instruction
    // fills 64 consecutive slots = 8 labels.  LDS stays linear (that is all the DMA can write); the bank
    // swizzle lives in WHICH 16 bytes a lane fetches: slot s of label r holds k-part s ^ ((r >> 1) & 7), and the
    // reads below apply the same XOR (the 16 lanes a ds_read_b128 serves at a time then hit 16 different
    // 16-byte bank groups).  The per-lane part of the address only depends on the parity of the
    // wave-instruction's index gi = w4*NT + r (label = 8 gi + lane/8, so (label >> 1) & 7 = (4 (gi & 1) + lane/16) & 7):
    // two byte offsets serve all of them, picked once per wave (even r / odd r).
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds;
    const unsigned voff0 = ((lane >> 3) * L + 4 * ((lane & 7) ^ (lane >> 4))) * 4;
    const unsigned voff1 = ((lane >> 3) * L + 4 * ((lane & 7) ^ (4 + (lane >> 4)))) * 4;
    const bool odd0 = (w4 * NT) & 1;
    const unsigned voff_even = odd0 ? voff1 : voff0, voff_odd = odd0 ? voff0 : voff1;    // for even / odd r
    auto load_mu_piece = [&](int kc, int buf, int r) {       // piece r of NT: 64 slots = 8 labels
        const int gi = w4 * NT + r;              // wave-instruction index within the chunk (the group's four waves share it)
        glds16(MuT + (int64_t)gi * 8 * L + 32 * kc, (r & 1) ? voff_odd : voff_even, lds_base + (buf * (L * 32) + gi * 256) * 4);
    };
    auto load_mu = [&](int kc, int buf) {        // issued by the four waves of ONE group
#pragma unroll
        for (int r = 0; r < NT; r++) load_mu_piece(kc, buf, r);
    };
    // B-operand reads: lane (i, h) takes k-part 2q + h of label 32t + i -> slot ((2q + h) ^ sw) of LDS row 32t + i
    const int sw = (i >> 1) & 7;
    const float *brow[4];
#pragma unroll
    for (int q = 0; q < 4; q++) brow[q] = lds + i * 32 + ((2 * q + h) ^ sw) * 4;

    // E0 / out / X addressing: a register r of the C/D map is pixel row rr(r) + 4h of the wave's 32, label
    // 32t + lane&31; the A operand of lane (i, h) is X[row i][8q + 4h ..].  Wave-uniform row pointer (scalar
    // registers) + one per-lane byte offset for the whole kernel + an immediate.  Every tile is whole, so there is
    // no row masking or clamping anywhere (a lane-dependent branch around a load would also make the compiler treat
    // the whole 16-register accumulator as that load's destination and wait for it at every later touch).
    // Padding is applied where a value is CONSUMED, not where it is loaded (a select on a fresh load would make the
    // wave wait for it): loads of padded columns are clamped into the row, pad_e0() / pad_x() overwrite them later.
    const int h4 = 4 * h;
    const unsigned lo_e = (unsigned)(h4 * e_rs + i) * 4u, lo_o = (unsigned)(h4 * o_rs + i) * 4u;
    const unsigned lo_e_last = PAD ? (unsigned)(h4 * e_rs + min(32 * (NT - 1) + i, Lr - 1) - 32 * (NT - 1)) * 4u : lo_e;
    const unsigned lo_x = (unsigned)(i * x_rs + h4) * 4u;
    const bool lane_stores_last = !PAD || 32 * (NT - 1) + i < Lr;
    auto rr = [](int r) { return (r & 3) + 8 * (r >> 2); };
#define PHL_E0_LOAD_ROW(r, wave_rows)  /* wave_rows: E0 row of the wave's first pixel */                                    \
    do {                                                                                                                  \
        const char *p_ = reinterpret_cast<const char *>((wave_rows) + rr(r) * e_rs);                                      \
        _Pragma("unroll") for (int t = 0; t < NT; t++)                                                                    \
            acc[t][r] = *reinterpret_cast<const float *>(p_ + 128 * t + ((PAD && t == NT - 1) ? lo_e_last : lo_e));       \
    } while (0)
    // A fragment q of chunk kc: columns 32 kc + 8q + 4h .. +3 of this lane's X row
    auto x_load_q = [&](float4 &a, const float *wave_rows, int kc, int q) {
        if (PAD && kc == NT - 1)                 // padded columns: clamp into the row (per-lane offsets, last chunk only)
            a = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(wave_rows) +
                                                  (unsigned)(i * x_rs + min(32 * kc + 8 * q + h4, Lr - 4)) * 4u);
        else
            a = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(wave_rows + 32 * kc) + 32 * q + lo_x);
    };
    auto x_load = [&](float4 (&a)[4], const float *wave_rows, int kc) {
#pragma unroll
        for (int q = 0; q < 4; q++) x_load_q(a[q], wave_rows, kc, q);
    };
    auto pad_e0 = [&](f32x16 &last) {                 // only the last label tile holds padding (L - Lr < 32)
        if (PAD && 32 * (NT - 1) + i >= Lr) {
#pragma unroll
            for (int r = 0; r < 16; r++) last[r] = INFINITY;
        }
    };
    auto pad_x = [&](float4 &v, int k0) {
        if (PAD && k0 + h4 >= Lr) v = make_float4(0.f, 0.f, 0.f, 0.f);
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
    float4 a[2][4];                              // A fragments: chunk kc lives in a[kc & 1] (no copies between chunks)
    if (valid) {
#pragma unroll
        for (int r = 0; r < 16; r++) PHL_E0_LOAD_ROW(r, E0 + (tile * 128 + w4 * 32) * e_rs);
        x_load(a[0], X + (tile * 128 + w4 * 32) * x_rs, 0);
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

    // One K chunk of the MFMA half.  PAR = kc & 1 picks the A registers; for even NT it is also the LDS buffer (a
    // half is NT slots, so `slot` is even whenever a half starts) and every LDS offset is an immediate.
    auto chunk = [&](auto par, int kc, const float *xrows) {
        constexpr int PAR = decltype(par)::value;
        const int buf = (NT % 2 == 0) ? PAR : (slot & 1);
        // The prefetch for the NEXT slot -- Mu's chunk (of this tile, or chunk 0 for the other group's tile; its buffer
        // was last read in the previous slot, behind the barrier) and this tile's next A fragments -- is issued in
        // pieces BETWEEN the MFMA steps: an LDS-DMA instruction takes 100-200 cycles to issue, and the 8 of them + 4
        // loads in one block ahead of the MFMAs kept the matrix pipe idle for a quarter of the chunk (in-kernel
        // stamps: 2.7k of 10.9k cycles).
        const int kc_next = kc + 1 < NT ? kc + 1 : 0;
        CS_ARRIVE(kc * 4 + 0);
        if (valid) {
            if (PAD && kc == NT - 1) {
#pragma unroll
                for (int q = 0; q < 4; q++) pad_x(a[PAR][q], 32 * kc + 8 * q);
            }
            // B operands in half-groups of NT/2 label tiles, software-pipelined by hand: the LDS reads of step s+1 are
            // issued BEFORE the MFMAs of step s (only one wave per SIMD is on the matrix cores at a time, so an LDS round
            // trip left exposed is matrix-pipe time lost).
            constexpr int HT = (NT + 1) / 2, STEPS = NT > 1 ? 8 : 4;     // NT == 1: whole groups
            float4 b[2][HT];
            auto read_b = [&](int st, float4 (&dst)[HT]) {
                const int q = NT > 1 ? st >> 1 : st, t0 = NT > 1 ? (st & 1) * HT : 0;
#pragma unroll
                for (int t = 0; t < HT; t++)
                    if (t0 + t < NT) dst[t] = *reinterpret_cast<const float4 *>(brow[q] + buf * (L * 32) + (t0 + t) * 32 * 32);
            };
            read_b(0, b[0]);
#pragma unroll
            for (int st = 0; st < STEPS; st++) {
                const int q = NT > 1 ? st >> 1 : st, t0 = NT > 1 ? (st & 1) * HT : 0;
                if (st + 1 < STEPS) read_b(st + 1, b[(st + 1) & 1]);
#pragma unroll
                for (int r = 0; r < NT; r++)
                    if (r * STEPS / NT == st) load_mu_piece(kc_next, buf ^ 1, r);
                if (kc + 1 < NT) {
                    if (STEPS == 8 && (st & 1)) x_load_q(a[PAR ^ 1][st >> 1], xrows, kc + 1, st >> 1);
                    if (STEPS == 4) x_load_q(a[PAR ^ 1][st], xrows, kc + 1, st);
                }
                __builtin_amdgcn_sched_barrier(0);
                float4(&bc)[HT] = b[st & 1];
                const float4 av = a[PAR][q];
#pragma unroll
                for (int t = 0; t < HT; t++) if (t0 + t < NT) acc[t0 + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bc[t].x, acc[t0 + t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < HT; t++) if (t0 + t < NT) acc[t0 + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bc[t].y, acc[t0 + t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < HT; t++) if (t0 + t < NT) acc[t0 + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bc[t].z, acc[t0 + t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < HT; t++) if (t0 + t < NT) acc[t0 + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bc[t].w, acc[t0 + t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            load_mu(kc_next, buf ^ 1);           // (a group out of tiles only keeps the LDS ring fed)
        }
        CS_ARRIVE(kc * 4 + 2);
        dma_drain();                             // the next chunk's DMA
        CS_ARRIVE(kc * 4 + 3);
        __builtin_amdgcn_s_barrier();
        slot++;
    };

    for (int64_t it = 0; it < iters; it++) {
        CS_STAMP(2 * it);
        // =========== MFMA half: E = E0 + X @ Mu on `tile`; the group in this half feeds the LDS ring ===========
        {
            const float *xrows = X + ((valid ? tile : 0) * 128 + w4 * 32) * x_rs;
            // every E0 load has landed by now (they were issued at least a slot ago); saying so HERE, with a use of
            // the accumulator that was loaded last, keeps the compiler from guarding single registers later
#ifdef __HIP_DEVICE_COMPILE__                    /* (the host pass has no "v" registers) */
            asm volatile("" ::"v"(acc[NT - 1]));
#endif
            pad_e0(acc[NT - 1]);
            for (int kc = 0; kc < NT; kc += 2) {
                chunk(std::integral_constant<int, 0>(), kc, xrows);
                if (kc + 1 < NT) chunk(std::integral_constant<int, 1>(), kc + 1, xrows);
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
            float scale[16];
#ifdef PHL_CS_NO_EPILOGUE            /* timing experiment: the MFMA halves alone */
            if (false) {
#else
            if (valid && !LOGITS) {
#endif
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
                // Q = exp / sum, in place and BEFORE the first E0 load of the next tile is issued: a write to one
                // register of a 16-register accumulator while a load into another one is in flight makes the
                // compiler wait for that load (it tracks the accumulator as one unit)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    scale[r] = __builtin_amdgcn_rcpf(scale[r]);
#pragma unroll
                    for (int t = 0; t < NT; t++) acc[t][r] *= scale[r];
                }
            } else if (valid) {                  // CRFasRNN returns -E of the last iteration, not Q (crf_module.py:103)
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = -acc[t];
            }
#pragma unroll
            for (int s = 0; s < NT; s++) {
                constexpr int D = NT > 1 ? NT - 1 : 1;
                const int r_lo = s < D ? (16 * s + D - 1) / D : 16, r_hi = s + 1 < D ? (16 * (s + 1) + D - 1) / D : 16;
#pragma unroll
                for (int r = 0; r < 16; r++) {   // (constant trip counts: both loops unroll and the test folds)
                    if (r < r_lo || r >= r_hi) continue;
#ifdef PHL_CS_NO_EPILOGUE
                    if (n != -12345) continue;     /* (never true: keeps the accumulators alive) */
#endif
                    if (valid) {
                        char *p = reinterpret_cast<char *>(orows + rr(r) * o_rs);
#pragma unroll
                        for (int t = 0; t < NT; t++)
                            if (!PAD || t < NT - 1 || lane_stores_last) *reinterpret_cast<float *>(p + 128 * t + lo_o) = acc[t][r];
                    }
                    // this row's registers are free: the next tile's E0 goes in
                    if (has_next) PHL_E0_LOAD_ROW(r, erows);
                }
                if (s == NT - 1 && has_next) x_load(a[0], X + (nxt * 128 + w4 * 32) * x_rs, 0);
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

// The last n % 128 pixels of phl_compat_softmax (the tile kernel takes whole tiles only): one workgroup per pixel,
// thread c owns label c -- an fmaf chain over k straight from the transposed compatibility matrix, then the row
// softmax through LDS.  At most 127 pixels: its speed does not matter, its arithmetic is the tile kernel's
// (f32 fma chain from E0, exp2 of the log2e-scaled difference to the row minimum).
template <bool LOGITS>
__global__ __launch_bounds__(256) void k_compat_tail(const float *__restrict__ E0, int64_t e_rs, const float *__restrict__ X,
                                                     int64_t x_rs, const float *__restrict__ MuT, int Lp,
                                                     float *__restrict__ out, int64_t o_rs, int64_t p0, int L)
{
    __shared__ float xs[256], red[256];
    const int64_t p = p0 + blockIdx.x;
    const int c = threadIdx.x;
    xs[c] = c < L ? X[p * x_rs + c] : 0.f;
    __syncthreads();
    float e = INFINITY;
    if (c < L) {
        e = E0[p * e_rs + c];
        for (int k = 0; k < L; k++) e = __builtin_fmaf(xs[k], MuT[(int64_t)c * Lp + k], e);
    }
    if (LOGITS) {
        if (c < L) out[p * o_rs + c] = -e;
        return;
    }
    red[c] = e;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (c < o) red[c] = fminf(red[c], red[c + o]);
        __syncthreads();
    }
    const float m = red[0] * 1.4426950408889634f;
    __syncthreads();
    const float v = c < L ? __builtin_amdgcn_exp2f(__builtin_fmaf(e, -1.4426950408889634f, m)) : 0.f;
    red[c] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (c < o) red[c] += red[c + o];
        __syncthreads();
    }
    if (c < L) out[p * o_rs + c] = v * __builtin_amdgcn_rcpf(red[0]);
}

#undef PHL_E0_LOAD_ROW
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
    if (L % 4 || L > 256 || x_rs % 4 || e_rs % 4 || o_rs % 4 || x_rs >= (1 << 24) || e_rs >= (1 << 24) || o_rs >= (1 << 24) ||
        ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(MuT) | reinterpret_cast<uintptr_t>(E0) | reinterpret_cast<uintptr_t>(out)) & 15)) {
        phl_set_error("phl_compat_softmax: needs L %% 4 == 0, L <= 256 and 16-byte aligned E0 / X / out rows (L=%d)", L);
        return PHL_ERR_UNSUPPORTED;
    }
    const int Lp = (L + 31) / 32 * 32;       // the tile width: mu_t is [Lp][Lp], zero beyond L
    const bool pad = Lp != L;
    hipStream_t st = (hipStream_t)stream;
    // persistent workgroups, one per CU (256 CUs): each walks pairs of whole 128-pixel tiles (one per wave group);
    // the last n % 128 pixels go to k_compat_tail
    const int64_t n_main = n / 128 * 128, npairs = (n / 128 + 1) / 2;
    const unsigned grid = (unsigned)(npairs < 256 ? npairs : 256);
    const size_t lds = (size_t)2 * Lp * 32 * sizeof(float);
    const bool logits = (flags & PHL_COMPAT_LOGITS) != 0;     // LDS is 2*L*128 B <= 64 KiB: no attribute needed
#ifdef PHL_COMPAT_TIMELINE
    static unsigned long long *tl_buf = nullptr;
    if (const char *path = getenv("PHL_COMPAT_TIMELINE")) {
        if (tl_buf) {                            // dump the previous launch
            hipDeviceSynchronize();
            std::vector<unsigned long long> hbuf(512 * 64);
            hipMemcpy(hbuf.data(), tl_buf, hbuf.size() * 8, hipMemcpyDeviceToHost);
            if (FILE *f = fopen(path, "wb")) { fwrite(hbuf.data(), 8, hbuf.size(), f); fclose(f); }
        } else {
            hipMalloc(&tl_buf, 512 * 64 * 8);
            hipMemcpyToSymbol(HIP_SYMBOL(g_cs_timeline), &tl_buf, sizeof(tl_buf));
        }
        hipMemset(tl_buf, 0, 512 * 64 * 8);
    }
#endif
#define PHL_CS(NT_)                                                                                                       \
    case NT_:                                                                                                             \
        if (logits && pad) k_compat_softmax<NT_, true, true><<<dim3(grid), dim3(512), lds, st>>>(E0, e_rs, X, x_rs, MuT, out, o_rs, n, L);        \
        else if (logits) k_compat_softmax<NT_, true, false><<<dim3(grid), dim3(512), lds, st>>>(E0, e_rs, X, x_rs, MuT, out, o_rs, n, L);         \
        else if (pad) k_compat_softmax<NT_, false, true><<<dim3(grid), dim3(512), lds, st>>>(E0, e_rs, X, x_rs, MuT, out, o_rs, n, L);            \
        else k_compat_softmax<NT_, false, false><<<dim3(grid), dim3(512), lds, st>>>(E0, e_rs, X, x_rs, MuT, out, o_rs, n, L);                    \
        break;
    if (n_main > 0) {
        switch (Lp / 32) {
            PHL_CS(1) PHL_CS(2) PHL_CS(3) PHL_CS(4) PHL_CS(5) PHL_CS(6) PHL_CS(7) PHL_CS(8)
        }
    }
    if (n > n_main) {
        if (logits) k_compat_tail<true><<<dim3((unsigned)(n - n_main)), dim3(256), 0, st>>>(E0, e_rs, X, x_rs, MuT, Lp, out, o_rs, n_main, L);
        else k_compat_tail<false><<<dim3((unsigned)(n - n_main)), dim3(256), 0, st>>>(E0, e_rs, X, x_rs, MuT, Lp, out, o_rs, n_main, L);
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
