/* A torch-free client of the C ABI (include/phl.h), in plain C: what a non-Python host of the reference's
 * lattice.filter(src, ref) would do.  Reads n, d, vd and the two arrays from a raw file, runs
 * phl_filter_once on device 0 and phl_build + phl_filter twice (init-once / filter-many), writes the
 * outputs.  Built and driven by tests/test_gpu_cabi_client.py.
 *   file in : int32 n, d, vd; float ref[n*d]; float src[n*vd]
 *   file out: float out_once[n*vd]; float out_many[n*vd]; int64 M
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "phl.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_PHL(x) do { int rc_ = (x); if (rc_ != PHL_OK) { fprintf(stderr, "phl %d (%s): %s\n", rc_, phl_status_string(rc_), phl_last_error()); return 3; } } while (0)

int main(int argc, char **argv)
{
    if (argc != 3) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    int32_t hdr[3];
    if (fread(hdr, sizeof(int32_t), 3, f) != 3) return 1;
    const int64_t n = hdr[0];
    const int d = hdr[1], vd = hdr[2];
    float *ref = (float *)malloc(sizeof(float) * n * d), *src = (float *)malloc(sizeof(float) * n * vd);
    float *out = (float *)malloc(sizeof(float) * n * vd);
    if (fread(ref, sizeof(float), n * d, f) != (size_t)(n * d) || fread(src, sizeof(float), n * vd, f) != (size_t)(n * vd)) return 1;
    fclose(f);
    if (phl_version() < 100 || phl_device_count() < 1) { fprintf(stderr, "no device / old library\n"); return 4; }

    float *dref, *dsrc, *dout;
    hipStream_t st;
    CHECK_HIP(hipSetDevice(0));
    CHECK_HIP(hipStreamCreate(&st));
    CHECK_HIP(hipMalloc((void **)&dref, sizeof(float) * n * d));
    CHECK_HIP(hipMalloc((void **)&dsrc, sizeof(float) * n * vd));
    CHECK_HIP(hipMalloc((void **)&dout, sizeof(float) * n * vd));
    CHECK_HIP(hipMemcpy(dref, ref, sizeof(float) * n * d, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dsrc, src, sizeof(float) * n * vd, hipMemcpyHostToDevice));

    f = fopen(argv[2], "wb");
    if (!f) return 1;
    /* one-shot call, the reference's call shape */
    CHECK_PHL(phl_filter_once(dsrc, vd, vd, 1, dref, d, d, 1, n, dout, vd, 1, 0, 0, (phl_stream)st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipMemcpy(out, dout, sizeof(float) * n * vd, hipMemcpyDeviceToHost));
    fwrite(out, sizeof(float), n * vd, f);
    /* init once, filter many */
    phl_lattice *lat = NULL;
    CHECK_PHL(phl_build(&lat, dref, n, d, d, 1, 0, (phl_stream)st));
    const int64_t M = phl_num_vertices(lat);
    for (int it = 0; it < 2; it++) CHECK_PHL(phl_filter(lat, dsrc, vd, vd, 1, dout, vd, 1, 0, (phl_stream)st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipMemcpy(out, dout, sizeof(float) * n * vd, hipMemcpyDeviceToHost));
    fwrite(out, sizeof(float), n * vd, f);
    fwrite(&M, sizeof(int64_t), 1, f);
    fclose(f);
    /* error path: codes, never abort */
    if (phl_filter(lat, dsrc, vd, vd, 1, dsrc, vd, 1, 0, (phl_stream)st) != PHL_ERR_INVALID) { fprintf(stderr, "aliasing not rejected\n"); return 5; }
    CHECK_PHL(phl_destroy(lat));
    (void)phl_trim_scratch();
    hipFree(dref); hipFree(dsrc); hipFree(dout);
    printf("ok M=%lld\n", (long long)M);
    return 0;
}
