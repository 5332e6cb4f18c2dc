// phl_costvol.hip -- the unary cost volume E_0 of the stereo CRF, produced on the device in the
// layout the lattice filter reads (pixel-major [h*w][L] fp32).
//
// Reference (caller side of the hot path, numpy + scipy on the CPU): crf/depth.py:36-53
//   disparity_badness(img1, img2, window_size, criterion):
//     cost[y,x,k] = sum_ch criterion(img1[y,x,ch], img2[y,x-k,ch])      img2 zero for x-k < 0  (:45-50)
//     out[y,x,k]  = sum over the ws x ws window of cost[.,.,k]           (:51-52)
//   with scipy.ndimage's default border rule 'reflect' (d c b a | a b c d | d c b a) on the COST array.
// criterion: AD |a-b| (:26-27), SD (a-b)^2 (:24-25), nprod -a*b (:28-29).
//
// One workgroup makes a TY x TX pixel tile for DC consecutive disparities: the image rows it needs
// go to LDS once, every thread then owns one (row, disparity) and forms the horizontal window sums
// in registers, the vertical sums are read back from LDS, and a wavefront stores 128 contiguous
// bytes per pixel.  Separable RUNNING sums (window enters with one add, leaves with one subtract,
// restarted every tile): ~4 adds per output instead of ws^2; the kernel's only HBM traffic of size
// is the 4*h*w*L-byte result.  VALU-bound (the raw costs), not HBM-bound.
#include <math.h>

#include "phl_internal.h"

namespace {

constexpr int TX = 16, TY = 16, DC = 32, CMAX = 4;   // threads = (TY + 2R) rows x DC disparities

__device__ __forceinline__ int reflect(int i, int n)
{
    // scipy 'reflect': -1 -> 0, -2 -> 1, n -> n-1, n+1 -> n-2 (period 2n)
    if (i >= 0 && i < n) return i;
    if (i < 0 && i >= -n) return -i - 1;          // one fold: the common border case, no division
    if (i >= n && i < 2 * n) return 2 * n - 1 - i;
    const int p = 2 * n;                          // windows larger than the image
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

__device__ __forceinline__ float4 ld_pixel(const float *img, int64_t pix, int C)
{
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);   // channels padded with zeros: every criterion gives 0 on (0, 0)
    const float *p = img + pix * C;
    v.x = p[0];
    if (C > 1) v.y = p[1];
    if (C > 2) v.z = p[2];
    if (C > 3) v.w = p[3];
    return v;
}

template <int CRIT> __device__ __forceinline__ float crit(float a, float b);
template <> __device__ __forceinline__ float crit<0>(float a, float b) { return fabsf(a - b); }
template <> __device__ __forceinline__ float crit<1>(float a, float b) { return (a - b) * (a - b); }
template <> __device__ __forceinline__ float crit<2>(float a, float b) { return -1.0f * a * b; }

template <int R, int CRIT>
__global__ __launch_bounds__((TY + 2 * R) * DC) void k_cost_volume(const float *__restrict__ img1, const float *__restrict__ img2, int h,
                                                    int w, int C, int L, float *__restrict__ out, int64_t out_rs)
{
    constexpr int ROWS = TY + 2 * R, COLS = TX + 2 * R, W2 = COLS + DC;   // staged extents
    constexpr int NT = ROWS * DC;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float4 *i1s = reinterpret_cast<float4 *>(lds);    // [ROWS][COLS]  img1 pixel (<= 4 channels) at reflected (row, col)
    float4 *i2s = i1s + ROWS * COLS;                  // [ROWS][W2]    img2 pixel, actual columns base2 .. base2+W2-1, 0 left of the image
    float *hs = reinterpret_cast<float *>(i2s + ROWS * W2);      // [ROWS][TX][DC]    horizontal window sums
    int *xr = reinterpret_cast<int *>(hs + ROWS * TX * DC);      // [COLS] reflected column of each tile column, as index into a row of i2s
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY, d0 = blockIdx.z * DC;
    // the tile's reflected columns fall on a contiguous range [cmin, cmax] of actual columns (at most COLS wide)
    int cmin = w;
    for (int xx = 0; xx < COLS; xx++) cmin = min(cmin, reflect(x0 - R + xx, w));   // tiny, uniform over the workgroup
    const int base2 = cmin - (d0 + DC - 1);           // leftmost img2 column any (column, disparity) pair reads
    for (int xx = threadIdx.x; xx < COLS; xx += NT) xr[xx] = reflect(x0 - R + xx, w) - d0 - base2;
    for (int e = threadIdx.x; e < ROWS * COLS; e += NT) {
        const int rr = e / COLS, xx = e - rr * COLS;
        const int y = reflect(y0 - R + rr, h), x = reflect(x0 - R + xx, w);
        i1s[e] = ld_pixel(img1, (int64_t)y * w + x, C);
    }
    for (int e = threadIdx.x; e < ROWS * W2; e += NT) {
        const int rr = e / W2, cc = e - rr * W2;
        const int y = reflect(y0 - R + rr, h), x = base2 + cc;
        i2s[e] = (x >= 0 && x < w) ? ld_pixel(img2, (int64_t)y * w + x, C) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // phase 1: thread = (row, disparity); cost of the row's COLS columns -> horizontal running window sums
    {
        const int k = threadIdx.x % DC, rr = threadIdx.x / DC;
        float c[COLS];
        const float4 *arow = i1s + rr * COLS;
        const float4 *brow = i2s + rr * W2 - k;
#pragma unroll
        for (int xx = 0; xx < COLS; xx++) {
            const float4 a = arow[xx], b = brow[xr[xx]];
            c[xx] = ((crit<CRIT>(a.x, b.x) + crit<CRIT>(a.y, b.y)) + crit<CRIT>(a.z, b.z)) + crit<CRIT>(a.w, b.w);
        }
        float s = c[0];
#pragma unroll
        for (int t = 1; t <= 2 * R; t++) s += c[t];
        hs[(rr * TX + 0) * DC + k] = s;
#pragma unroll
        for (int x = 1; x < TX; x++) {
            s = s + c[x + 2 * R] - c[x - 1];
            hs[(rr * TX + x) * DC + k] = s;
        }
    }
    __syncthreads();
    // phase 2: thread = (column, disparity); vertical running window sums, 128 contiguous bytes per pixel per wave half
    if (threadIdx.x < TX * DC) {
        const int k = threadIdx.x % DC, x = threadIdx.x / DC;
        const int gx = x0 + x, gk = d0 + k;
        if (gx < w && gk < L) {
            float s = hs[(0 * TX + x) * DC + k];
#pragma unroll
            for (int t = 1; t <= 2 * R; t++) s += hs[(t * TX + x) * DC + k];
            out[((int64_t)y0 * w + gx) * out_rs + gk] = s;
            for (int oy = 1; oy < TY && y0 + oy < h; oy++) {
                s = s + hs[((oy + 2 * R) * TX + x) * DC + k] - hs[((oy - 1) * TX + x) * DC + k];
                out[((int64_t)(y0 + oy) * w + gx) * out_rs + gk] = s;
            }
        }
    }
}

template <int R, int CRIT>
int launch(const float *img1, const float *img2, int h, int w, int C, int L, float *out, int64_t out_rs, hipStream_t st)
{
    constexpr int ROWS = TY + 2 * R, COLS = TX + 2 * R, W2 = COLS + DC;
    const size_t lds = sizeof(float) * (4 * (size_t)ROWS * COLS + 4 * (size_t)ROWS * W2 + COLS + (size_t)ROWS * TX * DC);
    if (lds > 64 * 1024)
        PHL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cost_volume<R, CRIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid((unsigned)((w + TX - 1) / TX), (unsigned)((h + TY - 1) / TY), (unsigned)((L + DC - 1) / DC));
    k_cost_volume<R, CRIT><<<grid, dim3(ROWS * DC), lds, st>>>(img1, img2, h, w, C, L, out, out_rs);
    PHL_HIP(hipGetLastError());
    return PHL_OK;
}

template <int CRIT>
int launch_r(int R, const float *img1, const float *img2, int h, int w, int C, int L, float *out, int64_t out_rs, hipStream_t st)
{
    switch (R) {
        case 0: return launch<0, CRIT>(img1, img2, h, w, C, L, out, out_rs, st);
        case 1: return launch<1, CRIT>(img1, img2, h, w, C, L, out, out_rs, st);
        case 2: return launch<2, CRIT>(img1, img2, h, w, C, L, out, out_rs, st);
        case 3: return launch<3, CRIT>(img1, img2, h, w, C, L, out, out_rs, st);
        case 4: return launch<4, CRIT>(img1, img2, h, w, C, L, out, out_rs, st);
        case 5: return launch<5, CRIT>(img1, img2, h, w, C, L, out, out_rs, st);
        case 6: return launch<6, CRIT>(img1, img2, h, w, C, L, out, out_rs, st);
        case 7: return launch<7, CRIT>(img1, img2, h, w, C, L, out, out_rs, st);
        default: return launch<8, CRIT>(img1, img2, h, w, C, L, out, out_rs, st);
    }
}

}  // namespace

extern "C" int phl_cost_volume(const float *img1, const float *img2, int h, int w, int channels, int max_disp, int window,
                               int criterion, float *out, int64_t out_rs, phl_stream stream)
{
    if (h < 1 || w < 1 || max_disp < 0 || !img1 || !img2 || (max_disp > 0 && !out) || out_rs < max_disp) {
        phl_set_error("phl_cost_volume: bad arguments");
        return PHL_ERR_INVALID;
    }
    if (channels < 1 || channels > CMAX || window < 1 || window % 2 == 0 || window > 17 || criterion < 0 || criterion > 2) {
        phl_set_error("phl_cost_volume: supports 1..%d channels, odd windows up to 17, criterion 0 (AD) / 1 (SD) / 2 (nprod); got c=%d ws=%d crit=%d",
                      CMAX, channels, window, criterion);
        return PHL_ERR_UNSUPPORTED;
    }
    if (max_disp == 0) return PHL_OK;
    const int R = window / 2;
    hipStream_t st = (hipStream_t)stream;
    switch (criterion) {
        case 0: return launch_r<0>(R, img1, img2, h, w, channels, max_disp, out, out_rs, st);
        case 1: return launch_r<1>(R, img1, img2, h, w, channels, max_disp, out, out_rs, st);
        default: return launch_r<2>(R, img1, img2, h, w, channels, max_disp, out, out_rs, st);
    }
}
