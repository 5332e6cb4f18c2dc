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
int main()
{
    for (int blocks : {256, 512, 1024}) { run<8>(blocks, 4096); run<4>(blocks, 8192); run<2>(blocks, 16384); run<1>(blocks, 32768); }
    return 0;
}
