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
__device__ __forceinline__ void glds16(const float *g, float *l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

template <int NT, bool LOGITS, bool PAD>
__global__ __launch_bounds__(256, 2) void k_compat_softmax(const float *__restrict__ E0, int64_t e_rs,
                                                           const float *__restrict__ X, int64_t x_rs,
                                                           const float *__restrict__ MuT, float *__restrict__ out,
                                                           int64_t o_rs, int64_t n, int Lr)
{
    // Lr = the real label count (a multiple of 4, <= L): columns Lr..L-1 are padding -- MuT is zero there, E0 reads
    // as +inf (so exp gives 0 and the row minimum ignores them), X reads as 0, nothing is stored.
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int L = 32 * NT;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 x [L labels][32 k], 16-byte slots XOR-swizzled
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int64_t ntiles = (n + 127) / 128;

    // Chunk loader (LDS-DMA, no staging registers): a chunk is L labels x 8 slots of 16 B; one wave-instruction
    // fills 64 consecutive slots = 8 labels.  LDS stays linear (that is all the DMA can write); the bank
    // swizzle lives in WHICH 16 bytes a lane fetches: slot s of label r holds k-part s ^ ((r >> 1) & 7), and the
    // reads below apply the same XOR (the 16 lanes a ds_read_b128 serves at a time then hit 16 different
    // 16-byte bank groups).
    auto load_mu = [&](int kc, int buf) {
#pragma unroll
        for (int r = 0; r < NT; r++) {
            const int g0 = (wave * NT + r) * 64;             // first slot of this wave-instruction
            const int g = g0 + lane;
            const int lab = g >> 3, part = (g & 7) ^ ((lab >> 1) & 7);
            glds16(MuT + (int64_t)lab * L + 32 * kc + 4 * part, lds + buf * (L * 32) + g0 * 4);
        }
    };
    auto x_row = [&](int64_t tile) {             // this lane's X row of a tile (clamped: loads stay in bounds)
        return X + min(tile * 128 + wave * 32 + i, n - 1) * x_rs + 4 * h;
    };
    auto e0_at = [&](int64_t tile, int r, int t) {
        const int64_t pr = min(tile * 128 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, n - 1);
        const int col = 32 * t + i;
        if (!PAD) return E0[pr * e_rs + col];
        const float e = E0[pr * e_rs + min(col, Lr - 1)];
        return col < Lr ? e : INFINITY;
    };
    auto x_at = [&](const float *xrow, int k0) {      // 4 contraction values from column k0 + 4h (0 beyond Lr)
        if (!PAD) return *reinterpret_cast<const float4 *>(xrow + k0);
        const float4 v = *reinterpret_cast<const float4 *>(xrow + min(k0, Lr - 4 - 4 * h));
        return (k0 + 4 * h) < Lr ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    };

    // PERSISTENT workgroups (two per CU) walk the 128-pixel tiles; everything the NEXT tile needs before its first
    // MFMA is requested while the current tile is in its epilogue: Mu's chunk 0 (both LDS buffers are idle then),
    // the X fragments of chunk 0, and -- register by register, as soon as a row has been stored -- the E0 tile,
    // which goes straight into the accumulators (the MFMA's C input: E = E0 + X @ Mu comes out of the matrix pipe
    // itself and the epilogue has no loads to wait for).  C/D map: column = lane&31,
    // row = (reg&3) + 8*(reg>>2) + 4*(lane>>5): for one register the 32 lanes of a half-wave hold 32 consecutive
    // labels of one pixel (128 B per access).
    int64_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    f32x16 acc[NT];
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t][r] = e0_at(tile, r, t);
    float4 a_cur[4], a_nxt[4];
    {
        const float *xrow = x_row(tile);
#pragma unroll
        for (int q = 0; q < 4; q++) a_nxt[q] = x_at(xrow, 8 * q);
    }
    load_mu(0, 0);
    __syncthreads();
    const int sw = (i >> 1) & 7;            // the swizzle of the labels this lane reads (32t + i)
    for (; tile < ntiles; tile += gridDim.x) {
        const float *xrow = x_row(tile);
        const int64_t row0 = tile * 128 + wave * 32;
        for (int kc = 0; kc < NT; kc++) {
#pragma unroll
            for (int q = 0; q < 4; q++) a_cur[q] = a_nxt[q];
            const bool more = kc + 1 < NT;       // uniform
            if (more) {
                // next chunk: the other buffer was last read in chunk kc-1, behind the previous barrier
                load_mu(kc + 1, (kc + 1) & 1);
#pragma unroll
                for (int q = 0; q < 4; q++) a_nxt[q] = x_at(xrow, 32 * (kc + 1) + 8 * q);
            }
            // keep the prefetch HERE: left alone, the scheduler sinks these loads to the end of the chunk (their
            // results are not needed before the next one) and the wave then waits out a full HBM miss per chunk
            __builtin_amdgcn_sched_barrier(0);
            const float *bbase = lds + (kc & 1) * (L * 32) + i * 32;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float4 b[NT];
                const int slot = ((2 * q + h) ^ sw) * 4;
#pragma unroll
                for (int t = 0; t < NT; t++) b[t] = *reinterpret_cast<const float4 *>(bbase + t * 32 * 32 + slot);
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].x, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].y, b[t].y, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].z, b[t].z, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].w, b[t].w, acc[t], 0, 0, 0);
            }
            __syncthreads();                     // (drains the DMA of the next chunk: issued 8k MFMA cycles ago)
        }

        // ---- next tile's inputs, requested before this tile's epilogue -------------------------------------
        const int64_t nxt = tile + gridDim.x;
        const bool has_next = nxt < ntiles;      // uniform
        if (has_next) {
            load_mu(0, 0);                       // every wave is past the last barrier: both buffers are idle
            const float *xn = x_row(nxt);
#pragma unroll
            for (int q = 0; q < 4; q++) a_nxt[q] = x_at(xn, 8 * q);
        }

        // ---- epilogue on the accumulators (acc = E now) ------------------------------------------------------
        // The 16 pixel rows a lane holds are reduced TOGETHER: sixteen independent butterfly chains per step, so the
        // cross-lane latency is paid once per step, not per row.
        if (LOGITS) {              // CRFasRNN returns -E of the last iteration, not Q (crf_module.py:103)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t prow = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (prow < n) {
#pragma unroll
                    for (int t = 0; t < NT; t++)
                        if (!PAD || 32 * t + i < Lr) out[prow * o_rs + 32 * t + i] = -acc[t][r];
                }
                if (has_next) {
#pragma unroll
                    for (int t = 0; t < NT; t++) acc[t][r] = e0_at(nxt, r, t);
                }
            }
        } else {
            float m[16], sum[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {       // softmax(-E): shift by the row MINIMUM of E
                m[r] = INFINITY;
#pragma unroll
                for (int t = 0; t < NT; t++) m[r] = fminf(m[r], acc[t][r]);
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1)     // the 32 lanes of this half-wave
#pragma unroll
                for (int r = 0; r < 16; r++) m[r] = fminf(m[r], __shfl_xor(m[r], o));
#pragma unroll
            for (int r = 0; r < 16; r++) {
                sum[r] = 0.f;
#pragma unroll
                for (int t = 0; t < NT; t++) {   // exp(-(E - min)) = exp2((min - E) * log2 e)
                    acc[t][r] = __builtin_amdgcn_exp2f((m[r] - acc[t][r]) * 1.4426950408889634f);
                    sum[r] += acc[t][r];
                }
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1)
#pragma unroll
                for (int r = 0; r < 16; r++) sum[r] += __shfl_xor(sum[r], o);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t prow = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float inv = 1.0f / sum[r];
                if (prow < n) {
#pragma unroll
                    for (int t = 0; t < NT; t++)
                        if (!PAD || 32 * t + i < Lr) out[prow * o_rs + 32 * t + i] = acc[t][r] * inv;
                }
                if (has_next) {                  // this row's registers are free: the next tile's E0 goes in
#pragma unroll
                    for (int t = 0; t < NT; t++) acc[t][r] = e0_at(nxt, r, t);
                }
            }
        }
        __syncthreads();                         // next tile's chunk 0 has landed in LDS (and its E0 / X in registers)
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
    if (L % 4 || L > 256 || x_rs % 4 || (reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(MuT) & 15)) {
        phl_set_error("phl_compat_softmax: needs L %% 4 == 0, L <= 256 and 16-byte aligned X rows (L=%d)", L);
        return PHL_ERR_UNSUPPORTED;
    }
    const int Lp = (L + 31) / 32 * 32;       // the tile width: mu_t is [Lp][Lp], zero beyond L
    const bool pad = Lp != L;
    hipStream_t st = (hipStream_t)stream;
    // persistent workgroups: two per CU (256 CUs), each walking tiles blockIdx.x, blockIdx.x + grid, ...
    const int64_t ntiles = (n + 127) / 128;
    const unsigned grid = (unsigned)(ntiles < 512 ? ntiles : 512);
    const size_t lds = (size_t)2 * Lp * 32 * sizeof(float);
    const bool logits = (flags & PHL_COMPAT_LOGITS) != 0;     // LDS is 2*L*128 B <= 64 KiB: no attribute needed
#define PHL_CS(NT_)                                                                                                       \
    case NT_:                                                                                                             \
        if (logits && pad) k_compat_softmax<NT_, true, true><<<dim3(grid), dim3(256), lds, st>>>(E0, e_rs, X, x_rs, MuT, out, o_rs, n, L);        \
        else if (logits) k_compat_softmax<NT_, true, false><<<dim3(grid), dim3(256), lds, st>>>(E0, e_rs, X, x_rs, MuT, out, o_rs, n, L);         \
        else if (pad) k_compat_softmax<NT_, false, true><<<dim3(grid), dim3(256), lds, st>>>(E0, e_rs, X, x_rs, MuT, out, o_rs, n, L);            \
        else k_compat_softmax<NT_, false, false><<<dim3(grid), dim3(256), lds, st>>>(E0, e_rs, X, x_rs, MuT, out, o_rs, n, L);                    \
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
