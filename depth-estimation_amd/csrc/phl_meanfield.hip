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
#include <atomic>

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

// Compatibility matrices of the Potts family, Mu = alpha J + beta I (J all ones: `potts` is alpha = 1, beta = -1,
// crf_module.py:55-64): X @ Mu = alpha rowsum(X) + beta X, so the whole non-lattice half of the iteration is one
// streaming pass over E0, X and Q -- no matrix product at all.  A wave per pixel row, NV float4 per lane.
template <int NV, bool LOGITS>
__global__ __launch_bounds__(256) void k_uniform_compat_softmax(const float *__restrict__ E0, int64_t e_rs,
                                                                const float *__restrict__ X, int64_t x_rs, float alpha,
                                                                float beta, float *__restrict__ out, int64_t o_rs,
                                                                int64_t n, int L)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t p = wave; p < n; p += nw) {
        float4 e[NV], x[NV];
        float xs = 0.f;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int c = (j * 64 + lane) * 4;
            if (c < L) {
                e[j] = *reinterpret_cast<const float4 *>(E0 + p * e_rs + c);
                x[j] = *reinterpret_cast<const float4 *>(X + p * x_rs + c);
                xs += (x[j].x + x[j].y) + (x[j].z + x[j].w);
            }
        }
        const float base = alpha * wave_sum(xs);
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int c = (j * 64 + lane) * 4;
            if (c < L) {            // x := -E = -(E0 + alpha sum + beta x)
                x[j] = make_float4(-(e[j].x + (base + beta * x[j].x)), -(e[j].y + (base + beta * x[j].y)),
                                   -(e[j].z + (base + beta * x[j].z)), -(e[j].w + (base + beta * x[j].w)));
                m = fmaxf(fmaxf(m, fmaxf(x[j].x, x[j].y)), fmaxf(x[j].z, x[j].w));
            }
        }
        float inv = 1.f;
        if (!LOGITS) {
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
            inv = 1.0f / wave_sum(s);
        }
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int c = (j * 64 + lane) * 4;
            if (c < L)
                *reinterpret_cast<float4 *>(out + p * o_rs + c) = make_float4(x[j].x * inv, x[j].y * inv, x[j].z * inv, x[j].w * inv);
        }
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
// product on the fp32-input matrix cores (v_mfma_f32_16x16x4_f32: exact f32, bitwise an fmaf chain in k order;
// gfx950 has no xf32/TF32 path and the reference computes in f32), with +E0, negate and the row softmax done on
// the accumulators.  Neither G = X@Mu nor E ever exists in HBM: the iteration's non-lattice half moves 3 x [n,L]
// (read X, read E0, write Q) instead of the 7 x [n,L] of GEMM + fused softmax, and is MFMA-bound
// (2 n L^2 flop at ~150 TF f32: 2.75 ms for 2048x1536x256).
//
// Tiling.  L = 32*NT <= 256, so a pixel's whole label row fits one wave: a wave owns 32 pixels x all L labels = 2 x 2NT
// accumulator tiles of 16x16 (4 VGPRs each, 128 at L = 256).  One persistent 512-thread workgroup per CU = two
// GROUPS of four waves, one wave of each per SIMD, working in anti-phase on alternate 128-pixel tiles: while one
// group walks K = L in NT chunks of 32 on the matrix cores ("MFMA half", NT slots), the other one is in the
// "epilogue half" of its previous tile -- stores Q, fetches the next tile's E0 straight into the accumulators (the
// MFMA's C input) -- and FEEDS the first: both operands of the MFMA stream come through LDS rings (Mu's chunk,
// 32 k x L labels = 32 KiB shared by the four waves; the tile's X chunk, 16 KiB), filled by LDS-DMA two slots
// ahead.  Every slot ends in the workgroup barrier.  The product is computed transposed (labels = MFMA rows), so
// that a lane holds four consecutive labels of a pixel in the four registers of an accumulator (16-byte E0 / Q
// accesses, 64 contiguous bytes per pixel) and a pixel's row sits in four lanes (softmax reductions: in-lane + two
// lane swaps).
//
// What shaped it (all measured on gfx950; tools/mfma_probe.hip, tools/compat_timeline.py):
//  * an f32 MFMA stream (measured on v_mfma_f32_32x32x2_f32) does not overlap with anything else the SIMD issues: a VALU instruction slipped between
//    two MFMAs costs its own time plus a bubble (SQ_VALU_MFMA_COEXEC_CYCLES = 0), and the OTHER wave of the SIMD --
//    VALU, scalar or memory instructions alike -- only gets issue slots in the stream's gaps.  Two independent
//    workgroups per CU therefore do not hide an epilogue behind a partner's MFMAs; they take turns anyway, and
//    unsynchronised they also spend time with both in their epilogue and the matrix pipe idle.
//  * a vector-memory instruction takes 100-200 cycles to ISSUE (an LDS-DMA piece as much as a load): the 8 + 4 of
//    them a wave needed per chunk kept the pipe idle for a quarter of the chunk, wherever in the chunk they stood.
//    Hence the loader duty of the epilogue group: the waves on the matrix cores issue MFMAs and LDS reads, nothing else.
//  * a wave can have only 63 vector-memory operations in flight; with a dword per lane (the untransposed C/D map)
//    that cap, at ~5 us loaded latency, paced the epilogue.  Hence 16-byte accesses.
//  * LDS-DMA pieces issued right behind the barrier delay the LDS reads the other group's first MFMAs wait for
//    (0.8 us per slot when the GPU is not at full clocks): the feeder sleeps 256 cycles first.
//
//  * what the CU's memory pipe charges for is cache lines per instruction: on 32x32x2 tiles a 16-byte access of the
//    transposed map touches 32 lines (32 pixels x 32 bytes), on 16x16x4 tiles 16 (16 pixels x 64 bytes) -- same
//    matrix rate, 3.97 -> 3.67 ms.
//
// Operand maps (16x16x4: lane l = (i = l&15, g = l>>4) supplies A[i][k-slot g] and B[k-slot g][j = i]).  A = Mu^T
// (rows = labels 16T + i), B = X^T (columns = pixels 16 pg + i of the wave's 32).  The k order of a contraction is
// free as long as A and B agree, so a lane reads 16 B = X[pixel][16q+4g .. 16q+4g+3] and the four values feed MFMAs
// u = 0..3 of group q: MFMA (q,u) contracts k = 16q + 4g' + u over the four k-slots g'; for A the lane reads
// Mu^T[16T+i][16q+4g .. +3] -- 16 contiguous bytes of the TRANSPOSED compatibility matrix, which is what the caller
// passes (MuT[c][k]).  Both LDS images are [row][32 k] with 128-byte rows whose 16-byte slots are XOR-swizzled (see
// the feeder).  16 bytes global -> LDS without a register in between (global_load_lds_dwordx4): the LDS address is
// wave-uniform (M0) plus lane*16, the global address is per lane.
//
// The DMA is issued through inline assembly, NOT __builtin_amdgcn_global_load_lds: with the builtin the compiler
// knows an LDS write is in flight, cannot tell it from the buffer the ds_reads use, and puts `s_waitcnt vmcnt(0)`
// in front of the first ds_read after every prefetch.  The kernel's own protocol makes that wait unnecessary (a
// buffer is only read behind the barrier that follows the counted wait for its DMA).
// Addressing: scalar base (SGPR pair) + 32-bit per-lane byte offset, so that a whole chunk's DMAs share two VGPRs.
__device__ __forceinline__ void glds16(const float *sbase, unsigned voff, unsigned lds_addr)
{
    unsigned keep;                              // m0 is the compiler's: hand it back as found
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr));
}
__device__ __forceinline__ void dma_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// VALU helpers of the epilogue.  f32 MFMAs and ordinary VALU instructions do NOT overlap on gfx950
// (tools/mfma_probe.hip: every v_fma slipped between two MFMAs costs its own issue time plus a ~10-cycle bubble,
// SQ_VALU_MFMA_COEXEC_CYCLES reads 0), so every epilogue instruction is paid for in matrix-pipe time: minima
// three at a time and without the compiler's NaN canonicalisation (v_max x,x before every v_min), the one
// cross-lane step as a lane-half swap instead of a ds_bpermute round trip.  (The swap is inline assembly because
// __builtin_amdgcn_permlane16/32_swap hands back its FIRST result twice in this compiler -- ROCm 7.2, checked with
// tools/dpp_probe.hip.)
__device__ __forceinline__ float vmin3(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// v_permlane32_swap_b32 a, b: the upper 32 lanes of a trade places with the lower 32 lanes of b (checked on the GPU:
// with a = b = x on entry, a holds x[lane & 31] and b holds x[32 + (lane & 31)] in every lane afterwards)
#define PHL_HALF_SWAP(a, b) asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b))
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

// compile-time loop: f(integral_constant<int, I>) for I = 0 .. N-1 (slot-dependent wait counts must be immediates)
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>());
        static_for<N, I + 1>(f);
    }
}

// v_permlane16_swap_b32 a, b: the odd 16-lane rows of a trade places with the even rows of b
#define PHL_ROW_SWAP(a, b) asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b))
template <int NT, bool LOGITS, bool PAD>
__global__ __launch_bounds__(512) void k_compat_softmax(const float *__restrict__ E0, int64_t e_rs,
                                                        const float *__restrict__ X, int64_t x_rs,
                                                        const float *__restrict__ MuT, float *__restrict__ out,
                                                        int64_t o_rs, int64_t n, int Lr)
{
    // Lr = the real label count (a multiple of 4, <= L): columns Lr..L-1 are padding -- MuT is zero there, E0 reads
    // as +inf (so exp gives 0 and the row minimum ignores them), X reads as 0, nothing is stored.
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int NL = 2 * NT;                     // label tiles of 16
    constexpr int L = 32 * NT;
    constexpr int MU_FLOATS = L * 32;             // one chunk of Mu^T: [L labels][32 k]
    constexpr int X_FLOATS = 4 * 1024;            // one chunk of the tile's X: 4 waves x [32 pixels][32 k]
    // LDS: Mu ring [3][MU_FLOATS], then X ring [3][X_FLOATS]; slot s reads buffers s % 3, and both are requested TWO
    // slots ahead (a request then has a whole slot beyond its own to land: with one slot the DMA's issue + latency
    // was the critical path of every slot); rows of 128 B, 16-byte slots XOR-swizzled
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, w4 = wave & 3;     // wave group (0/1) and wave within it
    const int i = lane & 15, g4 = lane >> 4;      // MFMA 16x16x4: column i, k-slot / row group g4
    const int64_t ntiles = n / 128;               // WHOLE tiles only: the launcher hands the last n % 128 pixels to k_compat_tail
    const int64_t G2 = 2 * (int64_t)gridDim.x;

    // RULES OF THIS KERNEL (each one measured, tools/mfma_probe.hip and the in-kernel stamps of the debug build):
    //  * the waves on the matrix cores issue NOTHING but MFMAs and LDS reads.  A VALU instruction between two
    //    f32 MFMAs do not overlap with them on gfx950, and a vector-memory instruction takes 100-200
    //    cycles to ISSUE (an LDS-DMA piece as much as a load): 8 + 4 of them per chunk kept the pipe idle for a
    //    quarter of the time.  So both operands come through LDS, and the OTHER wave group -- the one in its epilogue
    //    half, whose SIMD partner is busy with MFMAs anyway -- does all the fetching (loader duty).
    //  * every address is "scalar base + per-lane offset fixed for the whole kernel + immediate": the scalar unit
    //    does the arithmetic.

    // Loader duty (LDS-DMA, no staging registers).  A Mu chunk is L labels x 8 slots of 16 B; one wave-instruction
    // ("piece") fills 64 consecutive slots = 8 labels.  An X chunk is, per wave of the consuming group, 32 pixels x 8
    // slots = 4 pieces.  LDS stays linear (that is all the DMA can write); the bank swizzle lives in WHICH 16 bytes a
    // lane fetches: slot s of row r holds k-part s ^ ((r >> 1) & 7), and the reads apply the same XOR (the 16 lanes
    // a ds_read_b128 serves at a time then hit 16 different 16-byte bank groups).  The per-lane part of the address
    // only depends on the parity of the piece (row = 8 piece + lane/8, so (row >> 1) & 7 = (4 (piece & 1) + lane/16) & 7):
    // two byte offsets per matrix serve every piece.
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds;
    const int part0 = (lane & 7) ^ (lane >> 4), part1 = (lane & 7) ^ (4 + (lane >> 4));
    const unsigned mvoff0 = ((lane >> 3) * L + 4 * part0) * 4, mvoff1 = ((lane >> 3) * L + 4 * part1) * 4;
    const bool odd0 = (w4 * NT) & 1;              // piece index of Mu within the chunk: w4 * NT + r
    const unsigned mvoff_even = odd0 ? mvoff1 : mvoff0, mvoff_odd = odd0 ? mvoff0 : mvoff1;    // for even / odd r
    const unsigned xvoff0 = (unsigned)((lane >> 3) * x_rs + 4 * part0) * 4u, xvoff1 = (unsigned)((lane >> 3) * x_rs + 4 * part1) * 4u;
    // feed(S), called in slot S by the group that is NOT on the matrix cores (four waves): Mu's chunk and the X chunk
    // for slot S+2.  Who multiplies in a slot follows from its number alone: half ph = slot / NT belongs
    // to group ph & 1, which is then on its (ph >> 1)-th tile (group 1 started one half late), chunk slot % NT.
    // Returns the number of X pieces issued (0 if that group has run out of tiles) for the counted wait.
    const int64_t b2 = 2 * (int64_t)blockIdx.x;
    auto feed = [&](int S, int s3) -> bool {     // s3 = S mod 3 (S >= -2)
        const int ph = (S + 2) / NT, kc = (S + 2) - ph * NT, xb = s3 == 0 ? 2 : s3 - 1;     // (S + 2) % 3
#pragma unroll
        for (int r = 0; r < NT; r++) {
            const int gi = w4 * NT + r;
            glds16(MuT + (int64_t)gi * 8 * L + 32 * kc, (r & 1) ? mvoff_odd : mvoff_even, lds_base + (xb * MU_FLOATS + gi * 256) * 4);
        }
        const int64_t tile = (ph & 1) ? b2 + 1 + (int64_t)((ph - 1) >> 1) * G2 : b2 + (int64_t)(ph >> 1) * G2;
        const bool ok = tile < ntiles;
        if (ok) {
            const float *xw = X + (tile * 128 + w4 * 32) * x_rs;          // the consuming wave's 32 rows (same w4)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                unsigned vo = (j & 1) ? xvoff1 : xvoff0;
                if (PAD && kc == NT - 1)         // padded columns: clamp into the row (pad_x() zeroes them when consumed)
                    vo = (unsigned)((lane >> 3) * x_rs + min(4 * ((j & 1) ? part1 : part0), Lr - 4 - 32 * kc)) * 4u;
                glds16(xw + (int64_t)8 * j * x_rs + 32 * kc, vo, lds_base + (3 * MU_FLOATS + xb * X_FLOATS + w4 * 1024 + j * 256) * 4);
            }
        }
        asm volatile("" ::: "memory");           // the row traffic below stays behind the DMAs (counted wait)
        return ok;
    };
    // wait until all but this slot's own requests -- NT + 4 xp DMA pieces and `rows` accesses behind them, the wave's
    // youngest vector-memory operations -- are done (immediates only)
#define PHL_WAIT_BUT(xp, rows)                                                                                            \
    do {                                                                                                                  \
        if (xp) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((rows) + NT + 4 < 63 ? (rows) + NT + 4 : 63) : "memory");        \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((rows) + NT < 63 ? (rows) + NT : 63) : "memory");                   \
    } while (0)
    // operand reads (16x16x4: lane (i, g4) supplies A[i][k-slot g4] and B[k-slot g4][i]): the lane takes k-part 4q + g4
    // of row i (label 16T + i of Mu^T, pixel 16 pg + i of X) -> slot ((4q + g4) ^ sw) of that row; MFMA (q, u)
    // contracts k = 16q + 4 g4 + u
    const int sw = (i >> 1) & 7;
    const float *brow[2], *arow[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        brow[q] = lds + i * 32 + ((4 * q + g4) ^ sw) * 4;
        arow[q] = brow[q] + 3 * MU_FLOATS + w4 * 1024;
    }

    // The product is computed TRANSPOSED (D^T = Mu^T X^T: labels are the MFMA's rows, pixels its columns) on 16x16x4
    // tiles: lane (i, g4) holds pixels i and 16 + i of the wave's 32 (pixel groups pg = 0, 1); the four registers of
    // accumulator (pg, T) are labels 16T + 4 g4 .. +3.  An E0 / Q access is 16 bytes per lane and 64 contiguous bytes
    // per pixel: 16 lines per instruction (the 32x32x2 map: 32 pixels x 32 bytes = 32 lines, and what the CU's memory
    // pipe charges for is lines).  A pixel's row sits in four lanes (i + 16 g4).
    const unsigned lo_e = (unsigned)(i * e_rs + 4 * g4) * 4u, lo_o = (unsigned)(i * o_rs + 4 * g4) * 4u;
#define PHL_E0_LOAD_UNIT(pg, T, wave_rows)  /* wave_rows: E0 row of the wave's first pixel */                              \
    do {                                                                                                                  \
        const char *p_ = reinterpret_cast<const char *>((wave_rows) + (pg) * 16 * e_rs + 16 * (T));                       \
        unsigned lo_ = lo_e;                                                                                              \
        if (PAD && (T) >= NL - 2) {              /* clamp padded columns into the row */                                  \
            p_ = reinterpret_cast<const char *>((wave_rows) + (pg) * 16 * e_rs);                                          \
            lo_ = (unsigned)(i * e_rs + min(16 * (T) + 4 * g4, Lr - 4)) * 4u;                                             \
        }                                                                                                                 \
        const float4 v_ = *reinterpret_cast<const float4 *>(p_ + lo_);                                                    \
        acc[pg][T] = f32x4{v_.x, v_.y, v_.z, v_.w};                                                                       \
    } while (0)
    auto pad_x = [&](float4 &v, int k0) {
        if (PAD && k0 + 4 * g4 >= Lr) v = make_float4(0.f, 0.f, 0.f, 0.f);
    };

    // Tiles of this group: 2 (b + k gridDim) + grp, k = 0, 1, ...  Every wave alternates the MFMA half of a tile (NT
    // slots, one per K chunk) with its epilogue half (NT slots as well); a slot ends in the workgroup barrier.  Group 1
    // starts one half late (it spends that half feeding group 0), group 0 ends with NT bare barriers: whenever one
    // group is on the matrix cores, the other one -- its SIMD partner -- is storing Q, fetching E0 and feeding it.
    // Both groups run the same number of iterations (group 0 never has fewer tiles).
    const int64_t iters = (ntiles - 2 * (int64_t)blockIdx.x + G2 - 1) / G2;          // tiles 2b, 2b + G2, ... < ntiles
    int64_t tile = 2 * (int64_t)blockIdx.x + grp;
    bool valid = tile < ntiles;                  // wave-uniform
    f32x4 acc[2][NL];
    auto pad_e0 = [&]() {                             // only the last two label tiles can hold padding (L - Lr < 32)
        if (PAD) {
#pragma unroll
            for (int T = NL - 2; T < NL; T++)
                if (16 * T + 4 * g4 >= Lr) {
                    acc[0][T] = f32x4{INFINITY, INFINITY, INFINITY, INFINITY};
                    acc[1][T] = acc[0][T];
                }
        }
    };
    if (valid) {
#pragma unroll
        for (int pg = 0; pg < 2; pg++)
#pragma unroll
            for (int T = 0; T < NL; T++) PHL_E0_LOAD_UNIT(pg, T, E0 + (tile * 128 + w4 * 32) * e_rs);
    }
#ifdef __HIP_DEVICE_COMPILE__                    /* the first tile's E0 has landed before the loop is entered: with that known on
                                                    every path into it, the loop body needs no compiler-made vmcnt wait */
    asm volatile("" ::"v"(acc[1][NL - 1]));
#endif
    int slot = 0, slot3 = 0;                     // slots since the start, and that number mod 3
    auto next_slot = [&]() { slot++; slot3 = slot3 == 2 ? 0 : slot3 + 1; };
    if (grp == 1) {
        // group 0's first MFMA half is fed by group 1, which has nothing else to do yet: first what slots 0 and 1
        // need (feed(-1) and feed(-2) in the numbering above), then slot by slot
        feed(-2, 1);
        feed(-1, 2);
        dma_drain();
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s < NT; s++) {
            const bool xp = feed(slot, slot3);
            PHL_WAIT_BUT(xp, 0);
            __builtin_amdgcn_s_barrier();
            next_slot();
        }
    } else {
        __builtin_amdgcn_s_barrier();
    }

    // The softmax arithmetic, run by a wave at the END of its MFMA half (inside the last slot, before the barrier):
    // next to the other group's MFMA stream a wave gets a VALU issue slot about once per MFMA -- the ~400 instructions
    // below took 17k cycles there, and the slot waited for them; here they take their own 3-4k, with the matrix pipe
    // idle and the other group in the one slot of its epilogue half that has nothing else to do.
    auto softmax_in_place = [&]() {
    if (valid && !LOGITS) {
        // softmax(-E) of this lane's two pixels: 4 NL registers each here, the rest of the rows in lanes i + 16 g4.
        // Shift by the row MINIMUM of E; exp(-(E - min)) = exp2(min log2e - E log2e)
#pragma unroll
        for (int pg = 0; pg < 2; pg++) {
            float m = acc[pg][0][0];
            m = vmin3(m, acc[pg][0][1], acc[pg][0][2]);
#pragma unroll
            for (int T = 1; T < NL; T++) {
                m = vmin3(m, acc[pg][T][0], acc[pg][T][1]);
                m = vmin3(m, acc[pg][T][2], acc[pg][T][3]);
            }
            m = vmin3(m, acc[pg][0][3], acc[pg][0][3]);
            {
                float ma = m, mb = m;
                PHL_ROW_SWAP(ma, mb);
                m = vmin3(ma, mb, mb);
                ma = m; mb = m;
                PHL_HALF_SWAP(ma, mb);
                m = vmin3(ma, mb, mb) * 1.4426950408889634f;
            }
            f32x4 vs = 0.f;
#pragma unroll
            for (int T = 0; T < NL; T++) {
                acc[pg][T] = __builtin_elementwise_fma(acc[pg][T], (f32x4)(-1.4426950408889634f), (f32x4)m);
#pragma unroll
                for (int r = 0; r < 4; r++) acc[pg][T][r] = __builtin_amdgcn_exp2f(acc[pg][T][r]);
                vs += acc[pg][T];
            }
            float sum = (vs[0] + vs[1]) + (vs[2] + vs[3]);
            {
                float sa = sum, sb = sum;
                PHL_ROW_SWAP(sa, sb);
                sum = sa + sb;
                sa = sum; sb = sum;
                PHL_HALF_SWAP(sa, sb);
                sum = sa + sb;
            }
            const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
            for (int T = 0; T < NL; T++) acc[pg][T] *= inv;
        }
    } else if (valid) {                  // CRFasRNN returns -E of the last iteration, not Q (crf_module.py:103)
#pragma unroll
        for (int pg = 0; pg < 2; pg++)
#pragma unroll
            for (int T = 0; T < NL; T++) acc[pg][T] = -acc[pg][T];
    }
    };

    // One K chunk of the MFMA half: operands from LDS only (ring position slot3)
    auto chunk = [&](int kc) {
        if (valid) {
            float4 xb[2][2];                     // B operands: [pixel group][q]
            const float *bb[2];                  // (the ring position is the only vector arithmetic of the chunk)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                xb[0][q] = *reinterpret_cast<const float4 *>(arow[q] + slot3 * X_FLOATS);
                xb[1][q] = *reinterpret_cast<const float4 *>(arow[q] + slot3 * X_FLOATS + 16 * 32);
                bb[q] = brow[q] + slot3 * MU_FLOATS;
            }
            // A operands in blocks of TBS label tiles, software-pipelined by hand: the LDS reads of step s+1 are issued
            // BEFORE the MFMAs of step s (only one wave per SIMD is on the matrix cores at a time, so an LDS round trip
            // left exposed is matrix-pipe time lost)
            constexpr int NB = NL < 4 ? NL : 4, TBS = (NL + NB - 1) / NB, STEPS = 2 * NB;
            float4 ma[2][TBS];
            auto read_a = [&](int st, float4 (&dst)[TBS]) {
                const int q = st / NB, T0 = (st % NB) * TBS;
#pragma unroll
                for (int j = 0; j < TBS; j++)
                    if (T0 + j < NL) dst[j] = *reinterpret_cast<const float4 *>(bb[q] + (T0 + j) * 16 * 32);
            };
            read_a(0, ma[0]);
            if (PAD && kc == NT - 1) {
#pragma unroll
                for (int q = 0; q < 2; q++) { pad_x(xb[0][q], 32 * kc + 16 * q); pad_x(xb[1][q], 32 * kc + 16 * q); }
            }
#pragma unroll
            for (int st = 0; st < STEPS; st++) {
                const int q = st / NB, T0 = (st % NB) * TBS;
                if (st + 1 < STEPS) read_a(st + 1, ma[(st + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                float4(&mc)[TBS] = ma[st & 1];
                const float4 x0 = xb[0][q], x1 = xb[1][q];
#define PHL_MF16(comp)                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < TBS; j++) if (T0 + j < NL) {                                                  \
        acc[0][T0 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(mc[j].comp, x0.comp, acc[0][T0 + j], 0, 0, 0);            \
        acc[1][T0 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(mc[j].comp, x1.comp, acc[1][T0 + j], 0, 0, 0);            \
    }
                PHL_MF16(x) PHL_MF16(y) PHL_MF16(z) PHL_MF16(w)
#undef PHL_MF16
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (kc == NT - 1) softmax_in_place();
        // the DMAs this wave issued in the last slot of its epilogue half are what slot 1 of this half reads: they had this
        // whole slot to land, now make it certain (nothing else of this wave is in flight; in later slots: nothing at all)
        dma_drain();                             // (in every slot: a branch here upsets the compiler's own wait placement)
        CS_ARRIVE(kc);
        asm volatile("" ::: "memory");           // LDS reads of this slot stay on this side of the barrier
        __builtin_amdgcn_s_barrier();
        next_slot();
    };

    for (int64_t it = 0; it < iters; it++) {
        CS_STAMP(2 * it);
        // =========== MFMA half: E = E0 + X @ Mu on `tile` =======================================================
        {
#ifdef __HIP_DEVICE_COMPILE__                    /* (see the last epilogue slot: no wait is generated here, but without this
                                                    use the compiler guards single accumulator registers in the epilogue) */
            asm volatile("" ::"v"(acc[1][NL - 1]));
#endif
            pad_e0();
            for (int kc = 0; kc < NT; kc++) chunk(kc);
        }
        CS_STAMP(2 * it + 1);
        // =========== epilogue half: softmax and store of `tile`, E0 of the group's next tile in, loader duty =======
        // One piece per slot of the other group's MFMA half.  A slot first does its loader duty (feed(): what the
        // matrix cores read two slots from now), then moves its share of the 4 NT units of 4 registers a lane holds: a
        // stored unit's registers take the next tile's E0 at once (the MFMA's C input, so E = E0 + X @ Mu comes out of
        // the matrix pipe).  Before the barrier it waits for the DMAs that the NEXT slot reads, nothing younger (a
        // counted wait); the last slot moves no units, so that the E0 loads have landed when the MFMA half begins.
        {
            const int64_t nxt = tile + G2;
            const bool has_next = nxt < ntiles;  // uniform
            float *orows = out + (tile * 128 + w4 * 32) * o_rs;                        // wave-uniform
            const float *erows = E0 + ((has_next ? nxt : tile) * 128 + w4 * 32) * e_rs;
            static_for<NT>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if (s == NT - 1) {
                    // the units of this half are out and the next tile's E0 was requested at least a slot ago: the
                    // compiler waits for it HERE (a use of the accumulator loaded last), before this slot's DMAs
                    // are in flight, instead of draining them at the first MFMA
#ifdef __HIP_DEVICE_COMPILE__                    /* (the host pass has no "v" registers) */
                    asm volatile("" ::"v"(acc[1][NL - 1]));
#endif
                }
                // not at once: right behind the barrier the other group issues the LDS reads its first MFMAs wait for,
                // and a burst of DMA pieces ahead of them in the queue cost it 0.8 us per slot on a GPU that was
                // not at full clocks (at full clocks the delay neither helps nor hurts: 3.67 ms with 0, 128 or 512 cycles)
                __builtin_amdgcn_s_sleep(4);
                const bool xp = feed(slot, slot3);
                // this slot's share of the 4 NT units: store Q, and the freed registers take the next tile's E0 at
                // once (the MFMA's C input, so E = E0 + X @ Mu comes out of the matrix pipe)
                constexpr int U = 4 * NT, D = NT > 1 ? NT - 1 : 1;
                constexpr int u_lo = s < D ? (U * s + D - 1) / D : U, u_hi = s + 1 < D ? (U * (s + 1) + D - 1) / D : U;
#pragma unroll
                for (int u = 0; u < U; u++) {    // (constant trip count: unrolls, and the test folds)
                    if (u < u_lo || u >= u_hi) continue;
                    const int pg = u / NL, T = u % NL;
                    if (valid && (!PAD || T < NL - 2 || 16 * T + 4 * g4 < Lr))
                        *reinterpret_cast<float4 *>(reinterpret_cast<char *>(orows + pg * 16 * o_rs + 16 * T) + lo_o) =
                            make_float4(acc[pg][T][0], acc[pg][T][1], acc[pg][T][2], acc[pg][T][3]);
                    if (has_next) PHL_E0_LOAD_UNIT(pg, T, erows);
                }
                // What slot S+1 reads was requested in slot S-1 and must have landed.  vmcnt counts in issue order, so
                // everything younger may stay in flight: the units of slot S-1 (issued behind its DMAs), this slot's
                // DMAs and this slot's units -- E0 loads and Q stores get two slots to complete, not one (with one,
                // the slot length was the loaded memory latency)
                // (counted: loads always issue; under PAD a store of the last label tile may have no lane left and be
                // skipped, so those are not counted -- allowing fewer in flight only waits longer)
                constexpr int p_lo = s >= 1 && s - 1 < D ? (U * (s - 1) + D - 1) / D : U, p_hi = s >= 1 && s < D ? (U * s + D - 1) / D : U;
                constexpr int FIRM = PAD ? 0 : U;                        // (PAD: stores of the last two label tiles of either pixel group may be skipped; count none)
                constexpr int LOADS = (u_hi > u_lo ? u_hi - u_lo : 0) + (p_hi > p_lo ? p_hi - p_lo : 0);
                constexpr int STORES = ((u_hi < FIRM ? u_hi : FIRM) > u_lo ? (u_hi < FIRM ? u_hi : FIRM) - u_lo : 0) +
                                       ((p_hi < FIRM ? p_hi : FIRM) > p_lo ? (p_hi < FIRM ? p_hi : FIRM) - p_lo : 0);
                if (valid && has_next) PHL_WAIT_BUT(xp, LOADS + STORES);
                else if (has_next) PHL_WAIT_BUT(xp, LOADS);
                else if (valid) PHL_WAIT_BUT(xp, STORES);
                else PHL_WAIT_BUT(xp, 0);
                CS_ARRIVE(NT + s);
                __builtin_amdgcn_s_barrier();
                next_slot();
            });
            tile = nxt;
            valid = has_next;
        }
    }
    if (grp == 0) {
        for (int s = 0; s < NT; s++) __builtin_amdgcn_s_barrier();
    }
    dma_drain();                                 // the last feeds nobody reads: landed before the LDS is given back
}
#undef PHL_E0_LOAD_UNIT
#undef PHL_WAIT_BUT

// ------------------------------------------------------------------------------------------------------------
// k_compat_split: the same fused step for L in (128, 256] (a 256-label tile, padded above L) on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16, sixteen
// times the f32 rate), with BOTH operands split into three bf16 addends -- x = h + m + l, eight significant bits each,
// by truncation: the split is exact, h + m + l == x bit for bit -- and the six products that matter:
//     x mu  =  hH + hM + mH + hL + lH + mM   ( + mL + lM + lL  <=  2^-23 |x mu|, dropped )
// Every partial product of two bf16 numbers is exact in f32 and the matrix core accumulates in f32, so the result
// carries the error of an f32 dot product (measured against float64 next to the f32-MFMA kernel: tests and
// tools/compat_time.py), at 6/16 x 1/2 = 3/8 of its matrix time: the kernel becomes what the step is by its bytes,
// a streaming pass over E0, X and Q.
//
// Same skeleton as k_compat_softmax (two wave groups in anti-phase, the group in its epilogue half feeds the one on the
// matrix cores through LDS rings, every slot ends in the workgroup barrier; transposed product, 16x16 tiles, softmax on
// the accumulators) with these differences:
//  * the compatibility matrix arrives PREPARED (k_compat_planes): three bf16 planes in exactly the order the matrix
//    cores read them.  A slot is (K chunk of 32, label half): its piece of the planes is 128 labels x 32 k x 3 planes
//    = 24 KiB, [label tile][plane][lane] x 16 B -- a DMA piece is 64 consecutive 16-byte units, an operand read is
//    lane-linear (no swizzle, no bank conflict).  16 slots per half; pieces are requested THREE slots ahead into a ring
//    of four (96 KiB), X chunks (f32, 16 KiB, as before) four slots = two chunks ahead into a ring of three: the slots are
//    a fifth as long as the f32 kernel's, the latency to cover is the same.
//  * the waves on the matrix cores split their X operands themselves (8 values a lane and pixel group per chunk:
//    and / sub / and / sub + three byte permutes per pair), once per chunk, kept across the chunk's two slots.
//  * row traffic (Q stores, next E0 loads) is spread over the first EPI_D = 14 of the 16 epilogue slots; the E0 loads go
//    into the accumulators through inline assembly and the kernel waits for them itself at the top of the matrix half
//    (PHL_E0_LOAD_HIDDEN: a compiler-made wait would drain the DMA queue with them).  They are unconditional -- without a
//    next tile they re-read rows that exist -- so that they sit in straight-line code (tools/check_split_isa.py).
//  * one instance serves every L in (128, 256]: columns above L are padding (PAD: planes zero, E0 +inf, X fetches clamped
//    into the row and zeroed when consumed, nothing stored); the clamp arithmetic works on an opaque copy of L per slot so
//    that it is not hoisted out of the slot loop (its registers would spill).
//  * a wave that goes from its epilogue half to the matrix cores still has the DMAs of its last two slots in flight
//    (they feed slots 1 and 2 of the half it enters): counted waits at the end of its first two slots there.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// two f32 -> the packed bf16 pairs of their three addends (truncation: top 8 significant bits, the next 8, the last 8)
__device__ __forceinline__ void split3(float x0, float x1, unsigned &h, unsigned &m, unsigned &l)
{
    const unsigned b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
    const float r0 = x0 - __uint_as_float(b0 & 0xFFFF0000u), r1 = x1 - __uint_as_float(b1 & 0xFFFF0000u);
    const unsigned c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
    const float s0 = r0 - __uint_as_float(c0 & 0xFFFF0000u), s1 = r1 - __uint_as_float(c1 & 0xFFFF0000u);
    h = __builtin_amdgcn_perm(b1, b0, 0x07060302u);        // {hi16(x1), hi16(x0)}
    m = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
    l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

constexpr int CSP_PIECE = 24576;                 // bytes of one slot's piece of the planes
#ifndef PHL_CSP_XR
#define PHL_CSP_XR 3
#endif
#ifndef PHL_CSP_TB
#define PHL_CSP_TB 1
#endif
constexpr int CSP_XR = PHL_CSP_XR, CSP_TB = PHL_CSP_TB;
constexpr int CSP_PLANES_BYTES = 16 * CSP_PIECE; // 16 pieces: (K chunk 0..7) x (label half 0..1)

// MuT [Lp][Lp] f32 (Lp = the label count rounded up to 32, zero beyond the real label count; read as zero beyond Lp) ->
// planes: piece (kc, hf), label tile Tl of the half, plane P, lane (i, g): eight bf16 = plane P of
// MuT[16 (8 hf + Tl) + i][32 kc + 8 g .. + 7]
__global__ __launch_bounds__(256) void k_compat_planes(const float *__restrict__ MuT, int Lp, u32x4 *__restrict__ planes)
{
    const int t = blockIdx.x * 256 + threadIdx.x;           // one thread per (piece, Tl, lane): 16 * 8 * 64
    if (t >= 16 * 8 * 64) return;
    const int lane = t & 63, Tl = (t >> 6) & 7, piece = t >> 9, kc = piece >> 1, hf = piece & 1;
    const int label = 16 * (8 * hf + Tl) + (lane & 15), k0 = 32 * kc + 8 * (lane >> 4);
    const bool in = label < Lp && k0 < Lp;                  // (Lp is a multiple of 32: a lane's eight values are all in or all out)
    const float *src = MuT + (size_t)(in ? label : 0) * Lp + (in ? k0 : 0);
    u32x4 h, m, l;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        unsigned a, b, c;
        split3(in ? src[2 * j] : 0.f, in ? src[2 * j + 1] : 0.f, a, b, c);
        h[j] = a; m[j] = b; l[j] = c;
    }
    u32x4 *dst = planes + (size_t)piece * (CSP_PIECE / 16) + (size_t)Tl * 3 * 64 + lane;
    dst[0] = h;
    dst[64] = m;
    dst[128] = l;
}

template <bool LOGITS, bool PAD>
__global__ __launch_bounds__(512) void k_compat_split(const float *__restrict__ E0, int64_t e_rs,
                                                      const float *__restrict__ X, int64_t x_rs,
                                                      const unsigned char *__restrict__ planes, float *__restrict__ out,
                                                      int64_t o_rs, int64_t n, int Lr)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int NT = 8, NL = 16, NS = 16;        // K chunks, label tiles of 16, slots per half
    constexpr int X_FLOATS = 4 * 1024;            // one chunk of the tile's X: 4 waves x [32 pixels][32 k]
    constexpr int X_BASE = 4 * CSP_PIECE;         // bytes: the X ring [XR][X_FLOATS] sits behind the ring of four pieces
    constexpr int XR = CSP_XR, XA = 2 * (XR - 1); // X ring depth; an X chunk is requested XA slots = XR - 1 chunks ahead
#ifndef PHL_CSP_EPI_D
#define PHL_CSP_EPI_D 14
#endif
#ifndef PHL_CSP_SLEEP
#define PHL_CSP_SLEEP 2
#endif
    constexpr int EPI_D = PHL_CSP_EPI_D;          // epilogue slots that move rows
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    const int i = lane & 15, g4 = lane >> 4;
    const int64_t ntiles = n / 128;
    const int64_t G2 = 2 * (int64_t)gridDim.x;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds;
    const char *ldsb = reinterpret_cast<const char *>(lds);

    // X feed: as in k_compat_softmax (rows of 128 B, 16-byte slots XOR-swizzled by (row >> 1) & 7)
    const int part0 = (lane & 7) ^ (lane >> 4), part1 = (lane & 7) ^ (4 + (lane >> 4));
    const unsigned xvoff0 = (unsigned)((lane >> 3) * x_rs + 4 * part0) * 4u, xvoff1 = (unsigned)((lane >> 3) * x_rs + 4 * part1) * 4u;
    const unsigned mvoff = (unsigned)lane * 16u;
    const int64_t b2 = 2 * (int64_t)blockIdx.x;
    // feed(s, ph, xb): the requests of slot S = 16 ph + s (s >= -4 before the first slot): the piece of the planes for
    // slot S + 3 and, in even slots, the X chunk of slots S + XA and S + XA + 1 into X buffer xb = ((S + XA) >> 1) % XR.
    // Returns whether X pieces were issued (the consuming group may have run out of tiles).
    auto feed = [&](auto sc, int ph, int xb) -> bool {
        constexpr int s = decltype(sc)::value;
        int lr_now = Lr;                           // (opaque copy: the clamps of a padded tile are computed where they are used --
        if (PAD) asm volatile("" : "+s"(lr_now));  //  hoisted out of the slot loop they would cost some twenty registers)
        if constexpr (s + 3 >= 0) {
            constexpr int t = s + 3;
            const unsigned char *src = planes + (size_t)(t & 15) * CSP_PIECE + (size_t)w4 * 6 * 1024;
            const unsigned dst = lds_base + (t & 3) * CSP_PIECE + w4 * 6 * 1024;
#pragma unroll
            for (int r = 0; r < 6; r++) glds16(reinterpret_cast<const float *>(src + r * 1024), mvoff, dst + r * 1024);
        }
        bool ok = false;
        if constexpr (s + XA >= 0 && (s & 1) == 0) {
            constexpr int t = s + XA;
            const int php = ph + (t >= NS ? 1 : 0);
            constexpr int kc = (t & 15) >> 1;
            const int64_t tile = (php & 1) ? b2 + 1 + (int64_t)((php - 1) >> 1) * G2 : b2 + (int64_t)(php >> 1) * G2;
            ok = tile < ntiles;
            if (ok) {
                const float *xw = X + (tile * 128 + w4 * 32) * x_rs;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const unsigned dst = lds_base + X_BASE + (xb * X_FLOATS + w4 * 1024 + j * 256) * 4;
                    if (PAD && kc >= NT / 2 && 32 * kc + 32 > lr_now) {  // a chunk with padded columns: every fetch clamped into the row
                        const int col = min(32 * kc + 4 * ((j & 1) ? part1 : part0), lr_now - 4);   // (the consumer zeroes what is padding)
                        glds16(xw + (int64_t)8 * j * x_rs, (unsigned)((lane >> 3) * x_rs + col) * 4u, dst);
                    } else {
                        glds16(xw + (int64_t)8 * j * x_rs + 32 * kc, (j & 1) ? xvoff1 : xvoff0, dst);
                    }
                }
            }
        }
        asm volatile("" ::: "memory");
        return ok;
    };
#define PHL_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((N) < 63 ? (N) : 63) : "memory")

    const unsigned lo_e = (unsigned)(i * e_rs + 4 * g4) * 4u, lo_o = (unsigned)(i * o_rs + 4 * g4) * 4u;
#define PHL_E0_LOAD_UNIT(pg, T, wave_rows)                                                                                \
    do {                                                                                                                  \
        const char *p_ = reinterpret_cast<const char *>((wave_rows) + (pg) * 16 * e_rs + 16 * (T));                       \
        unsigned lo_ = lo_e;                                                                                              \
        if (PAD && (T) >= NL / 2 && 16 * (T) + 16 > lr_now) {                                                             \
            p_ = reinterpret_cast<const char *>((wave_rows) + (pg) * 16 * e_rs);                                          \
            lo_ = (unsigned)(i * e_rs + min(16 * (T) + 4 * g4, lr_now - 4)) * 4u;                                         \
        }                                                                                                                 \
        const float4 v_ = *reinterpret_cast<const float4 *>(p_ + lo_);                                                    \
        acc[pg][T] = f32x4{v_.x, v_.y, v_.z, v_.w};                                                                       \
    } while (0)

    // The same load, invisible to the compiler's wait counting (inline assembly): the loads of the NEXT tile's E0 go
    // straight into the accumulators in the first EPI_D slots of an epilogue half, and a compiler-made wait for them
    // would also drain the DMA pieces issued behind them (it cannot count those) -- a stalled slot per half.  The wait is
    // ours: a counted one at the start of the matrix-core half.  Sound only while the compiler leaves these registers alone
    // between load and wait (no copy, spill or AGPR move): tools/check_split_isa.py verifies that on the generated code.
#define PHL_E0_LOAD_HIDDEN(pg, T, wave_rows)                                                                              \
    do {                                                                                                                  \
        const float *p_ = (wave_rows) + (pg) * 16 * e_rs + 16 * (T);                                                      \
        unsigned lo_ = lo_e;                                                                                              \
        if (PAD && (T) >= NL / 2 && 16 * (T) + 16 > lr_now) {                                                             \
            p_ = (wave_rows) + (pg) * 16 * e_rs;                                                                          \
            lo_ = (unsigned)(i * e_rs + min(16 * (T) + 4 * g4, lr_now - 4)) * 4u;                                         \
        }                                                                                                                 \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(acc[pg][T]) : "v"(lo_), "s"(p_) : "memory");                 \
    } while (0)

    const int64_t iters = (ntiles - 2 * (int64_t)blockIdx.x + G2 - 1) / G2;
    int64_t tile = 2 * (int64_t)blockIdx.x + grp;
    bool valid = tile < ntiles;
    f32x4 acc[2][NL];
    auto pad_e0 = [&]() {
        if (PAD) {
            int lr_now = Lr;
            asm volatile("" : "+s"(lr_now));
#pragma unroll
            for (int T = NL / 2; T < NL; T++)
                if (16 * T + 4 * g4 >= lr_now) {
                    acc[0][T] = f32x4{INFINITY, INFINITY, INFINITY, INFINITY};
                    acc[1][T] = acc[0][T];
                }
        }
    };
    if (valid) {
        const int lr_now = Lr;
#pragma unroll
        for (int pg = 0; pg < 2; pg++)
#pragma unroll
            for (int T = 0; T < NL; T++) PHL_E0_LOAD_UNIT(pg, T, E0 + (tile * 128 + w4 * 32) * e_rs);
    }
#ifdef __HIP_DEVICE_COMPILE__
    asm volatile("" ::"v"(acc[1][NL - 1]));
#endif
    int ph = 0, x3 = 0;                            // half index of the current slot, (slot >> 1) % XR
    auto next_chunk = [&]() { x3 = x3 == XR - 1 ? 0 : x3 + 1; };
    if (grp == 1) {
        // group 0's first half is fed by group 1: first what its slots 0..3 need, then slot by slot
        static_for<XA>([&](auto ic) {
            constexpr int s = decltype(ic)::value - XA;          // slots -XA .. -1
            feed(std::integral_constant<int, s>(), 0, ((s + XA) >> 1) % XR);
        });
        dma_drain();
        __builtin_amdgcn_s_barrier();
        bool xe = false;
        static_for<NS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            const bool xp = feed(sc, 0, x3 == 0 ? XR - 1 : x3 - 1);
            if ((s & 1) == 0) xe = xp;
            // what slot s + 1 reads was requested in slot s - 2: the requests of slots s - 1 and s may stay in flight
            if (xe) PHL_VMCNT((s >= 1 ? 6 : 0) + 6 + 4);
            else PHL_VMCNT((s >= 1 ? 6 : 0) + 6);
            __builtin_amdgcn_s_barrier();
            if (s & 1) next_chunk();
        });
        ph = 1;
    } else {
        __builtin_amdgcn_s_barrier();
    }

    auto softmax_in_place = [&]() {
    if (valid && !LOGITS) {
#pragma unroll
        for (int pg = 0; pg < 2; pg++) {
            float m = acc[pg][0][0];
            m = vmin3(m, acc[pg][0][1], acc[pg][0][2]);
#pragma unroll
            for (int T = 1; T < NL; T++) {
                m = vmin3(m, acc[pg][T][0], acc[pg][T][1]);
                m = vmin3(m, acc[pg][T][2], acc[pg][T][3]);
            }
            m = vmin3(m, acc[pg][0][3], acc[pg][0][3]);
            {
                float ma = m, mb = m;
                PHL_ROW_SWAP(ma, mb);
                m = vmin3(ma, mb, mb);
                ma = m; mb = m;
                PHL_HALF_SWAP(ma, mb);
                m = vmin3(ma, mb, mb) * 1.4426950408889634f;
            }
            f32x4 vs = 0.f;
#pragma unroll
            for (int T = 0; T < NL; T++) {
                acc[pg][T] = __builtin_elementwise_fma(acc[pg][T], (f32x4)(-1.4426950408889634f), (f32x4)m);
#pragma unroll
                for (int r = 0; r < 4; r++) acc[pg][T][r] = __builtin_amdgcn_exp2f(acc[pg][T][r]);
                vs += acc[pg][T];
            }
            float sum = (vs[0] + vs[1]) + (vs[2] + vs[3]);
            {
                float sa = sum, sb = sum;
                PHL_ROW_SWAP(sa, sb);
                sum = sa + sb;
                sa = sum; sb = sum;
                PHL_HALF_SWAP(sa, sb);
                sum = sa + sb;
            }
            const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
            for (int T = 0; T < NL; T++) acc[pg][T] *= inv;
        }
    } else if (valid) {
#pragma unroll
        for (int pg = 0; pg < 2; pg++)
#pragma unroll
            for (int T = 0; T < NL; T++) acc[pg][T] = -acc[pg][T];
    }
    };

    // operand addresses: the planes of label tile Tl of the slot's piece at (3 Tl + P) KiB + lane * 16; the lane's X values
    // are k-parts 2 g4 and 2 g4 + 1 of row 16 pg + i
    const int sw = (i >> 1) & 7;
    const char *a_lane = ldsb + lane * 16;
    const float *x_lane0 = lds + X_BASE / 4 + w4 * 1024 + i * 32 + ((2 * g4) ^ sw) * 4;
    const float *x_lane1 = lds + X_BASE / 4 + w4 * 1024 + i * 32 + ((2 * g4 + 1) ^ sw) * 4;
    u32x4 xh[2], xm[2], xl[2];                     // the chunk's X operands: [pixel group], split
    bool x_late = true;                            // the X feeds behind the last E0 load of an epilogue half were all issued

    for (int64_t it = 0; it < iters; it++) {
        // =========== matrix-core half ===========================================================================
        // the tile's E0 (hidden loads of the epilogue's first EPI_D slots) has landed when at most the DMAs of the slots
        // behind them are in flight: six pieces of the planes a slot, four X pieces in the even ones if they were issued
        {
            constexpr int LATE = NS - EPI_D, LATE_EVEN = NS / 2 - (EPI_D + 1) / 2;     // slots EPI_D .. NS-1, the even ones among them
            if (it > 0) {
                if (x_late) PHL_VMCNT(6 * LATE + 4 * LATE_EVEN);
                else PHL_VMCNT(6 * LATE);
            }
        }
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll
        for (int T = 0; T < NL; T++) asm volatile("" : "+v"(acc[0][T]), "+v"(acc[1][T]));
        asm volatile("; CSP_E0_LANDED (tools/check_split_isa.py: accumulators may be touched from here to the slot's barrier)");
#endif
        pad_e0();
        static_for<NS>([&](auto sc) {
            constexpr int s = decltype(sc)::value, kc = s >> 1, hf = s & 1;
            if (valid) {
                const char *ab = a_lane + (s & 3) * CSP_PIECE;
                constexpr int TB = CSP_TB, STEPS = 8 / TB;
                u32x4 pa[2][TB][3];
                auto read_a = [&](int st, u32x4 (&dst)[TB][3]) {
#pragma unroll
                    for (int j = 0; j < TB; j++)
#pragma unroll
                        for (int P = 0; P < 3; P++)
                            dst[j][P] = *reinterpret_cast<const u32x4 *>(ab + ((st * TB + j) * 3 + P) * 1024);
                };
                read_a(0, pa[0]);
                if (hf == 0) {
#pragma unroll
                    for (int pg = 0; pg < 2; pg++) {
                        float4 v0 = *reinterpret_cast<const float4 *>(x_lane0 + x3 * X_FLOATS + pg * 16 * 32);
                        float4 v1 = *reinterpret_cast<const float4 *>(x_lane1 + x3 * X_FLOATS + pg * 16 * 32);
                        if (PAD && kc >= NT / 2) {
                            int lr_now = Lr;
                            asm volatile("" : "+s"(lr_now));
                            if (32 * kc + 8 * g4 >= lr_now) v0 = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (32 * kc + 8 * g4 + 4 >= lr_now) v1 = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                        unsigned h_[4], m_[4], l_[4];
                        split3(v0.x, v0.y, h_[0], m_[0], l_[0]);
                        split3(v0.z, v0.w, h_[1], m_[1], l_[1]);
                        split3(v1.x, v1.y, h_[2], m_[2], l_[2]);
                        split3(v1.z, v1.w, h_[3], m_[3], l_[3]);
                        xh[pg] = u32x4{h_[0], h_[1], h_[2], h_[3]};
                        xm[pg] = u32x4{m_[0], m_[1], m_[2], m_[3]};
                        xl[pg] = u32x4{l_[0], l_[1], l_[2], l_[3]};
                    }
                }
#pragma unroll
                for (int st = 0; st < STEPS; st++) {
                    if (st + 1 < STEPS) read_a(st + 1, pa[(st + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    u32x4(&pc)[TB][3] = pa[st & 1];
#define PHL_MFB(P, xop)                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < TB; j++) {                                                                      \
        acc[0][8 * hf + st * TB + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                           \
            __builtin_bit_cast(bf16x8, pc[j][P]), __builtin_bit_cast(bf16x8, xop[0]), acc[0][8 * hf + st * TB + j], 0, 0, 0); \
        acc[1][8 * hf + st * TB + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                           \
            __builtin_bit_cast(bf16x8, pc[j][P]), __builtin_bit_cast(bf16x8, xop[1]), acc[1][8 * hf + st * TB + j], 0, 0, 0); \
    }
                    // smallest terms first: L h, M m, H l, M h, H m, H h
                    PHL_MFB(2, xh) PHL_MFB(1, xm) PHL_MFB(0, xl) PHL_MFB(1, xh) PHL_MFB(0, xm) PHL_MFB(0, xh)
#undef PHL_MFB
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (s == NS - 1) softmax_in_place();
            // this wave's requests from the last two slots of its epilogue half feed slots 1 and 2 of this half
            if (s == 0) PHL_VMCNT(6);
            if (s == 1) PHL_VMCNT(0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s & 1) next_chunk();
        });
        ph++;
        // =========== epilogue half: stores of `tile`, E0 of the next tile, loader duty =============================
        {
            const int64_t nxt = tile + G2;
            const bool has_next = nxt < ntiles;
            float *orows = out + (tile * 128 + w4 * 32) * o_rs;
            // (the loads are unconditional: without a next tile they fetch rows that certainly exist -- this tile's, or tile 0's
            // if this group's tile itself lies beyond the end -- into registers nobody reads)
            const float *erows = E0 + ((has_next ? nxt : (valid ? tile : 0)) * 128 + w4 * 32) * e_rs;
            bool xe = false;
            x_late = true;
            static_for<NS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                __builtin_amdgcn_s_sleep(PHL_CSP_SLEEP);
                const bool xp = feed(sc, ph, x3 == 0 ? XR - 1 : x3 - 1);
                if ((s & 1) == 0) xe = xp;
                if (s >= EPI_D && (s & 1) == 0) x_late = x_late && xp;
                constexpr int U = 4 * NT, D = EPI_D;
                constexpr int u_lo = s < D ? (U * s + D - 1) / D : U, u_hi = s + 1 < D ? (U * (s + 1) + D - 1) / D : U;
                int lr_now = Lr;
                if (PAD) asm volatile("" : "+s"(lr_now));
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (u < u_lo || u >= u_hi) continue;
                    const int pg = u / NL, T = u % NL;
                    if (valid && (!PAD || T < NL / 2 || 16 * T + 4 * g4 < lr_now))
                        *reinterpret_cast<float4 *>(reinterpret_cast<char *>(orows + pg * 16 * o_rs + 16 * T) + lo_o) =
                            make_float4(acc[pg][T][0], acc[pg][T][1], acc[pg][T][2], acc[pg][T][3]);
                    PHL_E0_LOAD_HIDDEN(pg, T, erows);       // (no next tile: this one's rows again, never used -- no branch around the load)
                }
                // What slot s + 1 reads was requested in slot s - 2 or earlier.  vmcnt counts in issue order, so what this
                // wave issued behind the DMAs of slot s - 2 may stay in flight: the units of slot s - 2, the DMAs and units
                // of slots s - 1 and s.  (Under PAD a store may be skipped: none is counted -- waiting for fewer is safe.)
                constexpr int un0 = s < D ? u_hi - u_lo : 0;
                constexpr int un1 = (s >= 1 && s - 1 < D) ? ((s < D ? (U * s + D - 1) / D : U) - (U * (s - 1) + D - 1) / D) : 0;
                constexpr int un2 = (s >= 2 && s - 2 < D) ? ((s - 1 < D ? (U * (s - 1) + D - 1) / D : U) - (U * (s - 2) + D - 1) / D) : 0;
                constexpr int UN = un0 + un1 + un2;
                constexpr int DM = 6 + (s >= 1 ? 6 : 0);
                constexpr int ST = PAD ? 0 : UN;
                if (xe) {
                    if (valid) PHL_VMCNT(DM + 4 + UN + ST);
                    else PHL_VMCNT(DM + 4 + UN);
                } else {
                    if (valid) PHL_VMCNT(DM + UN + ST);
                    else PHL_VMCNT(DM + UN);
                }
                __builtin_amdgcn_s_barrier();
                if (s & 1) next_chunk();
            });
            ph++;
            tile = nxt;
            valid = has_next;
        }
    }
    if (grp == 0) {
        for (int s = 0; s < NS; s++) __builtin_amdgcn_s_barrier();
    }
    dma_drain();
}
#undef PHL_E0_LOAD_UNIT
#undef PHL_E0_LOAD_HIDDEN
#undef PHL_VMCNT
#undef PHL_ROW_SWAP

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

int phl_uniform_compat_softmax(const float *E0, int64_t e_rs, const float *X, int64_t x_rs, float alpha, float beta, float *out,
                               int64_t o_rs, int64_t n, int L, unsigned flags, phl_stream stream)
{
    if (n < 0 || L < 1 || (n > 0 && (!E0 || !X || !out))) { phl_set_error("phl_uniform_compat_softmax: bad arguments"); return PHL_ERR_INVALID; }
    if (n == 0) return PHL_OK;
    if (L % 4 || L > 1024 || x_rs % 4 || e_rs % 4 || o_rs % 4 ||
        ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(E0) | reinterpret_cast<uintptr_t>(out)) & 15)) {
        phl_set_error("phl_uniform_compat_softmax: needs L %% 4 == 0, L <= 1024 and 16-byte aligned E0 / X / out rows (L=%d)", L);
        return PHL_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = rows_grid(n);
    const bool logits = (flags & PHL_COMPAT_LOGITS) != 0;
#define PHL_UC(NV_)                                                                                                       \
    do {                                                                                                                  \
        if (logits) k_uniform_compat_softmax<NV_, true><<<dim3(grid), dim3(256), 0, st>>>(E0, e_rs, X, x_rs, alpha, beta, out, o_rs, n, L);  \
        else k_uniform_compat_softmax<NV_, false><<<dim3(grid), dim3(256), 0, st>>>(E0, e_rs, X, x_rs, alpha, beta, out, o_rs, n, L);        \
    } while (0)
    if (L <= 256) PHL_UC(1);
    else if (L <= 512) PHL_UC(2);
    else PHL_UC(4);
#undef PHL_UC
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
    const size_t lds = ((size_t)3 * Lp * 32 + 3 * 4 * 1024) * sizeof(float);     // Mu ring + X ring: 144 KiB at L = 256
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
    // more than 64 KiB of dynamic LDS needs the attribute, once per device and instance (not per launch: the call is
    // not a stream operation, and a launch may sit inside a stream capture)
    int dev = 0;
    PHL_HIP(hipGetDevice(&dev));
#define PHL_CS_LAUNCH(NT_, LG_, PD_)                                                                                    \
    do {                                                                                                                  \
        static std::atomic<unsigned long long> ready{0};                                                                  \
        const unsigned long long bit = 1ull << (dev & 63);                                                                \
        if (lds > 64 * 1024 && !(ready.load(std::memory_order_acquire) & bit)) {                                          \
            PHL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_compat_softmax<NT_, LG_, PD_>),                 \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
            ready.fetch_or(bit, std::memory_order_release);                                                               \
        }                                                                                                                 \
        k_compat_softmax<NT_, LG_, PD_><<<dim3(grid), dim3(512), lds, st>>>(E0, e_rs, X, x_rs, MuT, out, o_rs, n, L);     \
    } while (0)
#define PHL_CS(NT_)                                                                                                       \
    case NT_:                                                                                                             \
        if (logits && pad) PHL_CS_LAUNCH(NT_, true, true);                                                                \
        else if (logits) PHL_CS_LAUNCH(NT_, true, false);                                                                 \
        else if (pad) PHL_CS_LAUNCH(NT_, false, true);                                                                    \
        else PHL_CS_LAUNCH(NT_, false, false);                                                                            \
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
#undef PHL_CS_LAUNCH
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}

size_t phl_compat_planes_bytes(int L)
{
    return (L > 128 && L <= 256 && L % 4 == 0) ? (size_t)CSP_PLANES_BYTES : 0;
}

int phl_compat_prepare(const float *MuT, int L, void *planes, phl_stream stream)
{
    if (!phl_compat_planes_bytes(L)) { phl_set_error("phl_compat_prepare: the split kernel takes 128 < L <= 256, L %% 4 == 0 (L=%d)", L); return PHL_ERR_UNSUPPORTED; }
    if (!MuT || !planes || ((reinterpret_cast<uintptr_t>(MuT) | reinterpret_cast<uintptr_t>(planes)) & 15)) { phl_set_error("phl_compat_prepare: bad arguments"); return PHL_ERR_INVALID; }
    k_compat_planes<<<dim3(16 * 8 * 64 / 256), dim3(256), 0, (hipStream_t)stream>>>(MuT, (L + 31) / 32 * 32, reinterpret_cast<u32x4 *>(planes));
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}

int phl_compat_softmax_split(const float *E0, int64_t e_rs, const float *X, int64_t x_rs, const float *MuT, const void *planes,
                             float *out, int64_t o_rs, int64_t n, int L, unsigned flags, phl_stream stream)
{
    if (n < 0 || L < 1 || (n > 0 && (!E0 || !X || !MuT || !planes || !out))) { phl_set_error("phl_compat_softmax_split: bad arguments"); return PHL_ERR_INVALID; }
    if (n == 0) return PHL_OK;
    if (!phl_compat_planes_bytes(L) || x_rs % 4 || e_rs % 4 || o_rs % 4 || x_rs >= (1 << 24) || e_rs >= (1 << 24) || o_rs >= (1 << 24) ||
        ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(MuT) | reinterpret_cast<uintptr_t>(planes) | reinterpret_cast<uintptr_t>(E0) |
          reinterpret_cast<uintptr_t>(out)) & 15)) {
        phl_set_error("phl_compat_softmax_split: needs 128 < L <= 256, L %% 4 == 0 and 16-byte aligned E0 / X / out rows (L=%d)", L);
        return PHL_ERR_UNSUPPORTED;
    }
    const bool pad = L != 256;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_main = n / 128 * 128, npairs = (n / 128 + 1) / 2;
    const unsigned grid = (unsigned)(npairs < 256 ? npairs : 256);
    const size_t lds = (size_t)4 * CSP_PIECE + CSP_XR * 4 * 1024 * sizeof(float);     // ring of four pieces + X ring: 144 KiB
    const bool logits = (flags & PHL_COMPAT_LOGITS) != 0;
    int dev = 0;
    PHL_HIP(hipGetDevice(&dev));
#define PHL_CSP_LAUNCH(LG_, PD_)                                                                                          \
    do {                                                                                                                  \
        static std::atomic<unsigned long long> ready{0};                                                                  \
        const unsigned long long bit = 1ull << (dev & 63);                                                                \
        if (!(ready.load(std::memory_order_acquire) & bit)) {                                                             \
            PHL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_compat_split<LG_, PD_>),                        \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
            ready.fetch_or(bit, std::memory_order_release);                                                               \
        }                                                                                                                 \
        k_compat_split<LG_, PD_><<<dim3(grid), dim3(512), lds, st>>>(E0, e_rs, X, x_rs, reinterpret_cast<const unsigned char *>(planes), out, o_rs, n, L); \
    } while (0)
    if (n_main > 0) {
        if (logits && pad) PHL_CSP_LAUNCH(true, true);
        else if (logits) PHL_CSP_LAUNCH(true, false);
        else if (pad) PHL_CSP_LAUNCH(false, true);
        else PHL_CSP_LAUNCH(false, false);
    }
#undef PHL_CSP_LAUNCH
    if (n > n_main) {        // the last n % 128 pixels: the f32 chain of the other kernel's tail (at most 127 rows)
        const int Lp = (L + 31) / 32 * 32;
        if (logits) k_compat_tail<true><<<dim3((unsigned)(n - n_main)), dim3(256), 0, st>>>(E0, e_rs, X, x_rs, MuT, Lp, out, o_rs, n_main, L);
        else k_compat_tail<false><<<dim3((unsigned)(n - n_main)), dim3(256), 0, st>>>(E0, e_rs, X, x_rs, MuT, Lp, out, o_rs, n_main, L);
    }
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
