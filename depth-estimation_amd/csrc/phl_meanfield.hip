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

// float4 streaming copy: the measured HBM ceiling the roofline fractions are quoted against
__global__ __launch_bounds__(256) void k_stream_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, int64_t n4)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n4; i += stride) dst[i] = src[i];
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

int phl_stream_copy(const float *src, float *dst, int64_t n_floats, phl_stream stream)
{
    if (n_floats < 0 || n_floats % 4 || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15)) { phl_set_error("phl_stream_copy: needs 16-byte aligned buffers, n % 4 == 0"); return PHL_ERR_INVALID; }
    if (n_floats == 0) return PHL_OK;
    k_stream_copy<<<dim3(256 * 8), dim3(256), 0, (hipStream_t)stream>>>(reinterpret_cast<const float4 *>(src), reinterpret_cast<float4 *>(dst), n_floats / 4);
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
