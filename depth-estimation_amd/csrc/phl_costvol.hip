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
// bytes per pixel.  Separable sums: 2*ws adds per output instead of ws^2; the kernel's only HBM
// traffic of size is the 4*h*w*L-byte result.
#include <math.h>

#include "phl_internal.h"

namespace {

constexpr int TX = 16, TY = 8, DC = 32, NT = 512, CMAX = 4;

__device__ __forceinline__ int reflect(int i, int n)
{
    // scipy 'reflect': -1 -> 0, -2 -> 1, n -> n-1, n+1 -> n-2 (period 2n)
    if (n == 1) return 0;
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

template <int CRIT> __device__ __forceinline__ float crit(float a, float b);
template <> __device__ __forceinline__ float crit<0>(float a, float b) { return fabsf(a - b); }
template <> __device__ __forceinline__ float crit<1>(float a, float b) { return (a - b) * (a - b); }
template <> __device__ __forceinline__ float crit<2>(float a, float b) { return -1.0f * a * b; }

template <int R, int CRIT>
__global__ __launch_bounds__(NT) void k_cost_volume(const float *__restrict__ img1, const float *__restrict__ img2, int h,
                                                    int w, int C, int L, float *__restrict__ out, int64_t out_rs)
{
    constexpr int ROWS = TY + 2 * R, COLS = TX + 2 * R, W2 = COLS + DC;   // staged extents
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *i1s = lds;                                 // [C][ROWS][COLS]   img1 at reflected (row, col); planar: lanes walk columns
    float *i2s = i1s + C * ROWS * COLS;               // [C][ROWS][W2]     img2, actual columns base2 .. base2+W2-1, 0 left of the image
    int *xr = reinterpret_cast<int *>(i2s + C * ROWS * W2);      // [COLS] reflected column of each tile column
    float *hs = reinterpret_cast<float *>(xr + COLS);            // [ROWS][TX][DC]    horizontal window sums
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY, d0 = blockIdx.z * DC;
    // the tile's reflected columns fall on a contiguous range [cmin, cmax] of actual columns (at most COLS wide)
    int cmin = w;
    for (int xx = 0; xx < COLS; xx++) cmin = min(cmin, reflect(x0 - R + xx, w));   // tiny, uniform over the workgroup
    const int base2 = cmin - (d0 + DC - 1);           // leftmost img2 column any (column, disparity) pair reads
    for (int xx = threadIdx.x; xx < COLS; xx += NT) xr[xx] = reflect(x0 - R + xx, w);
    for (int e = threadIdx.x; e < ROWS * COLS; e += NT) {
        const int rr = e / COLS, xx = e - rr * COLS;
        const int y = reflect(y0 - R + rr, h), x = reflect(x0 - R + xx, w);
        for (int ch = 0; ch < C; ch++) i1s[ch * ROWS * COLS + e] = img1[((int64_t)y * w + x) * C + ch];
    }
    for (int e = threadIdx.x; e < ROWS * W2; e += NT) {
        const int rr = e / W2, cc = e - rr * W2;
        const int y = reflect(y0 - R + rr, h), x = base2 + cc;
        const bool in = x >= 0 && x < w;
        for (int ch = 0; ch < C; ch++) i2s[ch * ROWS * W2 + e] = in ? img2[((int64_t)y * w + x) * C + ch] : 0.f;
    }
    __syncthreads();
    // phase 1: thread = (row lane, disparity); cost of the row's COLS columns -> horizontal window sums
    {
        const int k = threadIdx.x % DC;
        for (int rr = threadIdx.x / DC; rr < ROWS; rr += NT / DC) {
            float c[COLS];
#pragma unroll
            for (int xx = 0; xx < COLS; xx++) {
                const float *a = i1s + rr * COLS + xx;
                const float *b = i2s + rr * W2 + (xr[xx] - (d0 + k) - base2);
                float s = 0.f;
                for (int ch = 0; ch < C; ch++) s += crit<CRIT>(a[ch * ROWS * COLS], b[ch * ROWS * W2]);
                c[xx] = s;
            }
#pragma unroll
            for (int x = 0; x < TX; x++) {
                float s = c[x];
#pragma unroll
                for (int t = 1; t <= 2 * R; t++) s += c[x + t];
                hs[(rr * TX + x) * DC + k] = s;
            }
        }
    }
    __syncthreads();
    // phase 2: thread = (column, disparity); vertical window sums, 128 contiguous bytes per pixel per wave half
    {
        const int k = threadIdx.x % DC, x = threadIdx.x / DC;
        const int gx = x0 + x, gk = d0 + k;
        if (gx < w && gk < L) {
            for (int oy = 0; oy < TY && y0 + oy < h; oy++) {
                float s = hs[(oy * TX + x) * DC + k];
#pragma unroll
                for (int t = 1; t <= 2 * R; t++) s += hs[((oy + t) * TX + x) * DC + k];
                out[((int64_t)(y0 + oy) * w + gx) * out_rs + gk] = s;
            }
        }
    }
}

template <int R, int CRIT>
int launch(const float *img1, const float *img2, int h, int w, int C, int L, float *out, int64_t out_rs, hipStream_t st)
{
    constexpr int ROWS = TY + 2 * R, COLS = TX + 2 * R, W2 = COLS + DC;
    const size_t lds = sizeof(float) * ((size_t)C * ROWS * COLS + (size_t)C * ROWS * W2 + COLS + (size_t)ROWS * TX * DC);
    if (lds > 64 * 1024)
        PHL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cost_volume<R, CRIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid((unsigned)((w + TX - 1) / TX), (unsigned)((h + TY - 1) / TY), (unsigned)((L + DC - 1) / DC));
    k_cost_volume<R, CRIT><<<grid, dim3(NT), lds, st>>>(img1, img2, h, w, C, L, out, out_rs);
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
