// How fast does v_mfma_f32_32x32x2_f32 issue from one / two waves per SIMD?  (calibration for phl_compat_softmax)
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; t++) for (int r = 0; r < 16; r++) acc[t][r] = (float)(threadIdx.x + t);
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int t = 0; t < NACC; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0;
    for (int t = 0; t < NACC; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters)
{
    float *d; hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<NACC><<<blocks, 256>>>(d, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<NACC><<<blocks, 256>>>(d, iters, 1.f, 2.f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double mfmas = (double)blocks * 4 * iters * 4 * NACC;      // per wave: iters*4*NACC
    const double per_simd = mfmas / 1024;                              // 256 CUs x 4 SIMDs
    printf("NACC=%d blocks=%4d (%.1f waves/SIMD): %.3f ms, %.1f TFLOP/s, %.1f ns per MFMA per SIMD (64 cycles = %.1f ns at 2.4 GHz)\n",
           NACC, blocks, blocks * 4 / 1024.0, ms, mfmas * 4096 / ms / 1e9, ms * 1e6 / per_simd, 64 / 2.4);
    hipFree(d);
}
// Same-wave interleave: KV independent v_fma behind every MFMA of a single wave per SIMD
template <int KV>
__global__ __launch_bounds__(256) void k_mix(float *out, int iters)
{
    f32x16 acc[8];
    for (int t = 0; t < 8; t++) for (int r = 0; r < 16; r++) acc[t][r] = (float)(threadIdx.x + t);
    float v[16];
    for (int r = 0; r < 16; r++) v[r] = (float)(threadIdx.x + r);
    float a = 1.f + threadIdx.x, b = 2.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 8; t++) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < KV; r++) v[r] = __builtin_fmaf(v[r], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int t = 0; t < 8; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
    for (int r = 0; r < 16; r++) s += v[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KV>
void mix()
{
    float *d; hipMalloc(&d, 256 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 8192;
    k_mix<KV><<<256, 256>>>(d, iters); hipDeviceSynchronize();
    hipEventRecord(a); k_mix<KV><<<256, 256>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("same wave, %2d v_fma per MFMA: %.1f cycles per MFMA (at 2.4 GHz)\n", KV, ms * 1e-3 * 2.4e9 / (iters * 8.0));
    hipFree(d);
}

// The compat kernel's inner loop in isolation: per q-group 8 ds_read_b128 (B operands, swizzled like the kernel's)
// and 32 MFMAs on 8 accumulators; MODE 0 = B from registers only, 1 = with the LDS reads, 2 = reads + a barrier
// every 4 q-groups.
__device__ __forceinline__ float hashf(unsigned x)      // pseudo-random float in [-1, 1)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (float)(int)x * (1.0f / 2147483648.0f);
}
// LB = the launch bound: 256 lets the compiler keep the accumulators in AccVGPRs (512 registers per lane at one
// wave per SIMD), 512 halves the budget and they end up in architectural VGPRs next to the A/B operands
template <int MODE, int LB = 256>
__global__ __launch_bounds__(LB) void k_loop(float *out, int iters, int random)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * 256 * 32];
    for (int j = threadIdx.x; j < 2 * 256 * 32; j += 256) lds[j] = random ? hashf(j * 7919u + blockIdx.x) : (float)(j & 15);
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5, sw = (i >> 1) & 7;
    f32x16 acc[8];
    for (int t = 0; t < 8; t++) for (int r = 0; r < 16; r++) acc[t][r] = (float)(threadIdx.x + t);
    float4 a = make_float4(1.f + lane, 2.f, 3.f, 4.f);
    float4 aq[4], an[4];
    for (int q = 0; q < 4; q++) { aq[q] = make_float4(1.f + lane + q, 2.f, 3.f + q, 4.f); an[q] = aq[q]; }
    float4 b[8];
    for (int t = 0; t < 8; t++) b[t] = make_float4(1.f + t, 2.f, 3.f, 4.f);
    if (random) {
        a = make_float4(hashf(threadIdx.x), hashf(threadIdx.x + 999), hashf(threadIdx.x + 77777), hashf(threadIdx.x + 31337));
        for (int t = 0; t < 8; t++) b[t] = make_float4(hashf(lane * 8 + t), hashf(lane * 8 + t + 4096), hashf(lane * 8 + t + 9999), hashf(lane + t * 131));
        for (int t = 0; t < 8; t++) for (int r = 0; r < 16; r++) acc[t][r] = hashf(threadIdx.x * 128 + t * 16 + r);
    }
    for (int it = 0; it < iters; it++) {
        const float *bbase = lds + (it & 1) * (256 * 32) + i * 32;
        if (MODE >= 3) {
#pragma unroll
            for (int q = 0; q < 4; q++) { aq[q] = an[q]; asm volatile("" : "+v"(an[q].x), "+v"(an[q].y), "+v"(an[q].z), "+v"(an[q].w)); }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (MODE >= 3) a = aq[q];
            if (MODE >= 1) {
                const int sl = ((2 * q + h) ^ sw) * 4;
#pragma unroll
                for (int t = 0; t < 8; t++) b[t] = *reinterpret_cast<const float4 *>(bbase + t * 32 * 32 + sl);
            }
#pragma unroll
            for (int t = 0; t < 8; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 8; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[t].y, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 8; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[t].z, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 8; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[t].w, acc[t], 0, 0, 0);
        }
        if (MODE >= 2) __builtin_amdgcn_s_barrier();
    }
    float s = 0;
    for (int t = 0; t < 8; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int LB = 256>
void loop_probe(int blocks, int random = 0)
{
    float *d; hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2048;
    k_loop<MODE, LB><<<blocks, 256>>>(d, iters, random); hipDeviceSynchronize();
    hipEventRecord(a); k_loop<MODE, LB><<<blocks, 256>>>(d, iters, random); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("inner loop mode %d%s bound %d, %d waves/SIMD: %.1f cycles per MFMA per SIMD (at 2.4 GHz), %.1f TFLOP/s\n", MODE, random ? " RANDOM data" : "", LB, blocks / 256,
           ms * 1e-3 * 2.4e9 / (iters * 128.0 * (blocks / 256)), (double)blocks * 4 * iters * 128 * 4096 / ms / 1e9);
    hipFree(d);
}

// Co-execution: one 512-thread workgroup per CU = 2 waves per SIMD.  Waves 0-3 issue MFMAs, waves 4-7 run `mode`:
// 1 = independent v_fma chains, 2 = streaming global loads + stores (dword per lane, 128-byte segments).
// Each group's duration comes from its own 100 MHz stamps (max over the grid).
__global__ __launch_bounds__(512) void k_coexec(float *out, float *mem, unsigned long long *stamps, int mfma_iters, int other_iters, int mode)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long t0 = wall_clock64();
    float s = 0;
    if (mode >= 16 && wave >= 4) __builtin_amdgcn_s_setprio(3);     // mode + 16: the non-MFMA waves run at high priority
    mode &= 15;
    if (wave < 4) {
        f32x16 acc[8];
        for (int t = 0; t < 8; t++) for (int r = 0; r < 16; r++) acc[t][r] = (float)(threadIdx.x + t);
        float a = 1.f + threadIdx.x, b = 2.f;
        for (int it = 0; it < mfma_iters; it++) {
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int t = 0; t < 8; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
        for (int t = 0; t < 8; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
    } else if (mode == 1) {
        float v[16];
        for (int r = 0; r < 16; r++) v[r] = (float)(lane + r);
        for (int it = 0; it < other_iters; it++) {
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = __builtin_fmaf(v[r], 1.0001f, 0.5f);
        }
        for (int r = 0; r < 16; r++) s += v[r];
    } else if (mode == 2) {
        // rows of 1 KiB; a wave walks its own region: 8 loads + 8 stores of 256 B per step
        float *base = mem + ((size_t)blockIdx.x * 4 + (wave - 4)) * ((size_t)other_iters * 2048) + lane;
        for (int it = 0; it < other_iters; it++) {
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = base[(size_t)it * 2048 + u * 64];
#pragma unroll
            for (int u = 0; u < 8; u++) base[(size_t)it * 2048 + 1024 + u * 64] = x[u] + 1.f;
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (lane == 0) { stamps[(blockIdx.x * 8 + wave) * 2] = t0; stamps[(blockIdx.x * 8 + wave) * 2 + 1] = t1; }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
void coexec(int mfma_iters, int other_iters, int mode, float *mem)
{
    const int blocks = 256;
    float *d; hipMalloc(&d, blocks * 512 * 4);
    unsigned long long *st; hipMalloc(&st, blocks * 16 * 8);
    for (int rep = 0; rep < 2; rep++) { k_coexec<<<blocks, 512>>>(d, mem, st, mfma_iters, other_iters, mode); hipDeviceSynchronize(); }
    unsigned long long h[256 * 16];
    hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    double dm = 0, dother = 0;
    for (int b = 0; b < blocks; b++) for (int w = 0; w < 8; w++) {
        const double us = (h[(b * 8 + w) * 2 + 1] - h[(b * 8 + w) * 2]) / 100.0;
        if (w < 4) dm += us / (blocks * 4); else dother += us / (blocks * 4);
    }
    printf("mfma_iters=%5d other(mode %d)_iters=%6d : MFMA waves %8.1f us (%.1f cycles/MFMA at 2.4 GHz), other waves %8.1f us",
           mfma_iters, mode & 15, other_iters, dm, mfma_iters ? dm * 2400 / (mfma_iters * 32.0) : 0.0, dother);
    if (mode >= 16) printf(" [prio]");
    mode &= 15;
    if (mode == 1 && other_iters) printf(" (%.2f cycles per v_fma)", dother * 2400 / (other_iters * 128.0));
    if (mode == 2 && other_iters) printf(" (%.1f GB/s per CU, %.2f TB/s)", other_iters * 4 * 4096.0 / dother / 1e3, other_iters * 4 * 4096.0 / dother / 1e6 * 256);
    printf("\n");
    hipFree(d); hipFree(st);
}
// sustained rate: the same launch 24 times back to back, each timed (clock ramp-up from idle, power limits)
void sustain()
{
    float *d; hipMalloc(&d, 512 * 256 * 4);
    hipEvent_t ev[25];
    for (auto &e : ev) hipEventCreate(&e);
    for (int random = 0; random < 2; random++) {
        hipEventRecord(ev[0]);
        for (int k = 0; k < 24; k++) { k_loop<1, 256><<<512, 256>>>(d, 1024, random); hipEventRecord(ev[k + 1]); }
        hipDeviceSynchronize();
        printf("inner loop (LDS operands%s), 2 waves/SIMD, 24 launches back to back, TFLOP/s:", random ? ", random data" : "");
        for (int k = 0; k < 24; k++) { float ms; hipEventElapsedTime(&ms, ev[k], ev[k + 1]); printf(" %.0f", 512.0 * 4 * 1024 * 128 * 4096 / ms / 1e9); }
        printf("\n");
    }
    hipFree(d);
}
int main(int argc, char **argv)
{
    if (argc > 1 && argv[1][0] == 's') { sustain(); return 0; }
    if (argc > 1) {
        float *mem; const size_t bytes = (size_t)256 * 4 * 1024 * 8192;    // other_iters <= 1024
        hipMalloc(&mem, bytes); hipMemset(mem, 0, bytes);
        loop_probe<3, 512>(256); loop_probe<3, 512>(512); loop_probe<3, 256>(256); loop_probe<1, 512>(256); loop_probe<0>(256, 1); loop_probe<1>(256, 1); loop_probe<1>(512, 1); loop_probe<0>(256); loop_probe<1>(256); loop_probe<2>(256); loop_probe<0>(512); loop_probe<1>(512); loop_probe<2>(512);
        mix<0>(); mix<2>(); mix<4>(); mix<8>(); mix<12>(); mix<16>();
        coexec(2048, 0, 1, mem); coexec(0, 4096, 1, mem); coexec(2048, 4096, 1, mem); coexec(2048, 1024, 1, mem);
        coexec(0, 1024, 2, mem); coexec(2048, 1024, 2, mem); coexec(2048, 256, 2, mem);
        coexec(2048, 4096, 17, mem); coexec(2048, 1024, 17, mem); coexec(2048, 1024, 18, mem); coexec(2048, 256, 18, mem);
        return 0;
    }
    for (int blocks : {256, 512, 1024}) { run<8>(blocks, 4096); run<4>(blocks, 8192); run<2>(blocks, 16384); run<1>(blocks, 32768); }
    return 0;
}
