// Streaming-copy probe: which access form reaches the highest R+W rate on this MI355X box?
// build: hipcc --offload-arch=gfx950 -O3 tools/copy_probe.hip -o tools/copy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float vf4 __attribute__((ext_vector_type(4)));

template <int MODE, int UNROLL>
__global__ __launch_bounds__(256) void k_copy(const vf4 *__restrict__ src, vf4 *__restrict__ dst, long n4)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        vf4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (MODE & 1) v[u] = __builtin_nontemporal_load(&src[i + u * stride]);
            else v[u] = src[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (MODE & 2) __builtin_nontemporal_store(v[u], &dst[i + u * stride]);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}

template <int MODE, int UNROLL>
void run(const char *name, const vf4 *s, vf4 *d, long n4, int blocks)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_copy<MODE, UNROLL><<<blocks, 256>>>(s, d, n4);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) k_copy<MODE, UNROLL><<<blocks, 256>>>(s, d, n4);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s blocks=%6d unroll=%d : %7.1f GB/s (R+W)\n", name, blocks, UNROLL, 2.0 * n4 * 16 * 5 / (ms * 1e-3) / 1e9);
}

int main()
{
    const long n4 = (long)3145728 * 256 / 4;   // the benchmark's 3.2 GB value volume
    vf4 *s, *d;
    hipMalloc(&s, n4 * 16); hipMalloc(&d, n4 * 16);
    hipMemset(s, 1, n4 * 16);
    for (int blocks : {2048, 4096, 16384, 65536}) {
        run<0, 4>("plain", s, d, n4, blocks);
        run<2, 4>("nt store", s, d, n4, blocks);
        run<3, 4>("nt load + nt store", s, d, n4, blocks);
        run<0, 8>("plain", s, d, n4, blocks);
        run<2, 8>("nt store", s, d, n4, blocks);
    }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) hipMemcpyAsync(d, s, n4 * 16, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s : %7.1f GB/s (R+W)\n", "hipMemcpyAsync D2D", 2.0 * n4 * 16 * 5 / (ms * 1e-3) / 1e9);
    return 0;
}
