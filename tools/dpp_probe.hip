// Semantics check of the lane-swap instructions the compat kernel's softmax uses (gfx950):
//   v_permlane16_swap_b32 a, b : the odd 16-lane rows of a trade places with the even rows of b
//   v_permlane32_swap_b32 a, b : the upper 32 lanes of a trade places with the lower 32 lanes of b
// and of the compiler builtin, which returns its FIRST result twice (ROCm 7.2) -- hence inline assembly in the kernel.
// build: hipcc --offload-arch=gfx950 -O3 tools/dpp_probe.hip -o tools/dpp_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const float *in, float *o16a, float *o16b, float *o32a, float *o32b, float *ob0, float *ob1)
{
    const float x = in[threadIdx.x];
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    o16a[threadIdx.x] = a; o16b[threadIdx.x] = b;
    a = x; b = x;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    o32a[threadIdx.x] = a; o32b[threadIdx.x] = b;
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    ob0[threadIdx.x] = __builtin_bit_cast(float, r[0]); ob1[threadIdx.x] = __builtin_bit_cast(float, r[1]);
}
int main()
{
    float h[64], *d, *o; hipMalloc(&d, 256); hipMalloc(&o, 6 * 256);
    for (int i = 0; i < 64; i++) h[i] = (float)i;
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, o + 64, o + 128, o + 192, o + 256, o + 320);
    float r[384]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int bad16 = 0, bad32 = 0, builtin_same = 0;
    for (int l = 0; l < 64; l++) {
        const int row = l >> 4, c = l & 15;
        bad16 += r[l] != (float)((row & ~1) * 16 + c) || r[64 + l] != (float)((row | 1) * 16 + c);
        bad32 += r[128 + l] != (float)(l & 31) || r[192 + l] != (float)(32 + (l & 31));
        builtin_same += r[256 + l] == r[320 + l];
    }
    printf("v_permlane16_swap (asm): %d mismatches; v_permlane32_swap (asm): %d mismatches; builtin: both results equal in %d of 64 lanes\n",
           bad16, bad32, builtin_same);
    return bad16 || bad32;
}
