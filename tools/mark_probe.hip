#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void k_mark(unsigned long long *w, unsigned long long t) { __threadfence_system(); *(volatile unsigned long long *)w = t; }
__global__ void k_mark_nofence(unsigned long long *w, unsigned long long t) { *(volatile unsigned long long *)w = t; }
__global__ void k_busy(float *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
int main() {
    unsigned long long *hw = nullptr;
    hipError_t e = hipHostMalloc((void **)&hw, 8, hipHostMallocMapped);
    printf("hostmalloc %d\n", (int)e);
    *hw = 0;
    hipStream_t st; hipStreamCreate(&st);
    float *d; hipMalloc(&d, 1 << 24);
    // 1. stream write value on pinned host memory
    e = hipStreamWriteValue64(st, hw, 7ull, 0);
    printf("hipStreamWriteValue64 on pinned host: rc %d (%s)\n", (int)e, hipGetErrorString(e));
    hipStreamSynchronize(st);
    printf("value now %llu\n", *(volatile unsigned long long *)hw);
    // ordering: busy kernel then write value: value must appear only after kernel end
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int variant = 0; variant < 3; variant++) {
        hipDeviceSynchronize();
        hipEventRecord(a, st);
        for (int r = 0; r < 200; r++) {
            k_busy<<<64, 256, 0, st>>>(d, 16384);
            if (variant == 0) k_mark<<<1, 1, 0, st>>>(hw, 100 + r);
            else if (variant == 1) k_mark_nofence<<<1, 1, 0, st>>>(hw, 100 + r);
            else hipStreamWriteValue64(st, hw, 100 + r, 0);
        }
        hipEventRecord(b, st);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("variant %d: %.2f us per (busy + mark), final value %llu\n", variant, ms * 1000 / 200, *(volatile unsigned long long *)hw);
    }
    hipDeviceSynchronize();
    hipEventRecord(a, st);
    for (int r = 0; r < 200; r++) k_busy<<<64, 256, 0, st>>>(d, 16384);
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("busy alone: %.2f us\n", ms * 1000 / 200);
    // signal memory
    unsigned long long *sg = nullptr;
    e = hipExtMallocWithFlags((void **)&sg, 8, hipMallocSignalMemory);
    printf("signal memory alloc rc %d\n", (int)e);
    return 0;
}
