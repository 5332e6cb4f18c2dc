// scratch test of the half-wave all-reduce helpers (DPP rotations + v_permlane16_swap)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#define PHL_DPP_ROR(op, x, n) asm("s_nop 1\n\t" op " %0, %1, %1 row_ror:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(x) : "v"(x))
__global__ void k(const float *in, float *omin, float *osum, float *oswap0, float *oswap1)
{
    float x = in[threadIdx.x], y = x;
    PHL_DPP_ROR("v_min_f32_dpp", x, 8); PHL_DPP_ROR("v_min_f32_dpp", x, 4); PHL_DPP_ROR("v_min_f32_dpp", x, 2); PHL_DPP_ROR("v_min_f32_dpp", x, 1);
    float xa = x, xb = x;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(xa), "+v"(xb));
    oswap0[threadIdx.x] = xa; oswap1[threadIdx.x] = xb;
    omin[threadIdx.x] = fminf(xa, xb);
    PHL_DPP_ROR("v_add_f32_dpp", y, 8); PHL_DPP_ROR("v_add_f32_dpp", y, 4); PHL_DPP_ROR("v_add_f32_dpp", y, 2); PHL_DPP_ROR("v_add_f32_dpp", y, 1);
    float ya = y, yb = y;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(ya), "+v"(yb));
    osum[threadIdx.x] = ya + yb;
}
int main()
{
    float h[64], *d, *o; hipMalloc(&d, 256); hipMalloc(&o, 4 * 256);
    for (int i = 0; i < 64; i++) h[i] = (float)((i * 37) % 64) + 1;
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, o + 64, o + 128, o + 192);
    float r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    for (int hf = 0; hf < 2; hf++) {
        float m = 1e9, s = 0; for (int i = 0; i < 32; i++) { m = fminf(m, h[hf * 32 + i]); s += h[hf * 32 + i]; }
        printf("half %d expect min %g sum %g | got", hf, m, s);
        for (int i = 0; i < 32; i += 5) printf(" [%g %g]", r[hf * 32 + i], r[64 + hf * 32 + i]);
        printf("\n");
    }
    printf("row mins after rotations (lanes 0,16,32,48): swap0 %g %g %g %g  swap1 %g %g %g %g\n", r[128], r[144], r[160], r[176], r[192], r[208], r[224], r[240]);
    return 0;
}
