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

// read-only (sum into one float per thread) and write-only (fill) streams: how far is a one-directional
// stream from the copy rate?
template <int UNROLL>
__global__ __launch_bounds__(256) void k_read(const vf4 *__restrict__ src, float *__restrict__ sink, long n4)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    vf4 acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        vf4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u];
    }
    for (; i < n4; i += stride) acc += src[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;   // never true: keeps the loads alive
}

__global__ __launch_bounds__(256) void k_fill(vf4 *__restrict__ dst, long n4)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    const vf4 v = {1.f, 2.f, 3.f, 4.f};
    for (; i < n4; i += stride) dst[i] = v;
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
    for (int blocks : {2048, 8192, 65536}) {
        float ms;
        k_read<8><<<blocks, 256>>>(s, (float *)d, n4);
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int r = 0; r < 5; r++) k_read<8><<<blocks, 256>>>(s, (float *)d, n4);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("read only   blocks=%6d : %7.1f GB/s\n", blocks, 1.0 * n4 * 16 * 5 / (ms * 1e-3) / 1e9);
        hipEventRecord(a);
        for (int r = 0; r < 5; r++) k_fill<<<blocks, 256>>>(d, n4);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("write only  blocks=%6d : %7.1f GB/s\n", blocks, 1.0 * n4 * 16 * 5 / (ms * 1e-3) / 1e9);
    }
    // size sweep of the best form: is a higher published copy rate (MI355X_MICROARCH.md quotes 6.29 TB/s) a
    // matter of footprint (buffers that fit the 256 MiB Infinity Cache) rather than of the access form?
    for (long mb : {32L, 64L, 128L, 256L, 512L, 1024L, 3072L}) {
        const long m4 = mb * 1024 * 1024 / 16;
        const int blocks = (int)((m4 + 1023) / 1024 < 65536 ? (m4 + 1023) / 1024 : 65536);
        float ms;
        k_copy<3, 4><<<blocks, 256>>>(s, d, m4);
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int r = 0; r < 20; r++) k_copy<3, 4><<<blocks, 256>>>(s, d, m4);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("copy of %5ld MiB (nt, %d blocks): %7.1f GB/s (R+W)\n", mb, blocks, 2.0 * m4 * 16 * 20 / (ms * 1e-3) / 1e9);
    }
    // relative placement of source and destination: both on 2 MiB boundaries vs the destination shifted
    {
        vf4 *big;
        hipMalloc(&big, n4 * 16 + (64 << 20));
        for (long off : {0L, 256L, 1024L, 4096L, 16384L, 65536L, 1L << 20, (1L << 20) + 4096}) {
            vf4 *dd = (vf4 *)((char *)big + off);
            float ms;
            k_copy<3, 4><<<65536, 256>>>(s, dd, n4);
            hipDeviceSynchronize();
            hipEventRecord(a);
            for (int r = 0; r < 5; r++) k_copy<3, 4><<<65536, 256>>>(s, dd, n4);
            hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
            printf("copy, dst shifted by %8ld B from a 2 MiB boundary: %7.1f GB/s (R+W)   (src %p dst %p)\n", off, 2.0 * n4 * 16 * 5 / (ms * 1e-3) / 1e9, (void *)s, (void *)dd);
        }
        hipFree(big);
    }
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) hipMemcpyAsync(d, s, n4 * 16, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s : %7.1f GB/s (R+W)\n", "hipMemcpyAsync D2D", 2.0 * n4 * 16 * 5 / (ms * 1e-3) / 1e9);
    return 0;
}
