// TEST INFRASTRUCTURE ONLY -- never linked into or called from the product path.
//
// Torch-free driver around the reference's own lattice engine
//   /root/reference/crf/lattice/lite/permutohedral.h
// The reference header is NOT copied into this repo.  oracle/build_ref.sh
// compiles this file against a temporary, line-filtered view of that header
// (everything except `#include <torch/torch.h>` and the at::Tensor
// marshalling function at :199-321, which no longer compiles on torch 2.10 --
// SURVEY.md section 8c) and writes only oracle/_ref/libphl_ref.so.
//
// What this harness does is exactly what PermutohedralLattice::filter
// (permutohedral.h:199-321) does between its tensor copies: construct the
// lattice (:208), splat every row (:236-238), blur (:260), beginSlice +
// slice every row (:264-276).  Stage dumps are read from the engine's own
// members after each stage.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#define printf(...) ((void)0)      /* the engine chats on stdout (:123,:499,:543) */
#define private public             /* replay[] (:558-562) is private; we only read it */
#include "perm_nofilter.h"
#undef private
#undef printf

extern "C" {

// Full filter.  Optional dumps (any may be NULL):
//   M_out            number of lattice vertices after splat
//   keys_out         [M*d]  int16 vertex keys in insertion order        (cap: n*(d+1) rows)
//   splat_out        [M*vd] vertex values after splat                   (cap: n*(d+1) rows)
//   blur_out         [M*vd] vertex values after blur
//   replay_off_out   [n*(d+1)] vertex index (offset/vd) per (pixel, remainder)
//   replay_w_out     [n*(d+1)] barycentric weight per (pixel, remainder)
int ref_filter(const float *ref, const float *src, int n, int d, int vd,
               float *out, int *M_out, short *keys_out, float *splat_out,
               float *blur_out, int *replay_off_out, float *replay_w_out)
{
    PermutohedralLattice lattice(d, vd, n);
    float *pos = new float[d];
    float *val = new float[vd];
    for (int i = 0; i < n; ++i) {
        memcpy(pos, ref + (size_t)i * d, sizeof(float) * d);
        memcpy(val, src + (size_t)i * vd, sizeof(float) * vd);
        lattice.splat(pos, val);
    }
    int M = lattice.hashTable.size();
    if (M_out) *M_out = M;
    if (keys_out) memcpy(keys_out, lattice.hashTable.getKeys(), sizeof(short) * (size_t)M * d);
    if (splat_out) memcpy(splat_out, lattice.hashTable.getValues(), sizeof(float) * (size_t)M * vd);
    if (replay_off_out || replay_w_out) {
        for (size_t e = 0; e < (size_t)n * (d + 1); ++e) {
            if (replay_off_out) replay_off_out[e] = lattice.replay[e].offset / vd;
            if (replay_w_out) replay_w_out[e] = lattice.replay[e].weight;
        }
    }
    lattice.blur();
    if (blur_out) memcpy(blur_out, lattice.hashTable.getValues(), sizeof(float) * (size_t)M * vd);
    lattice.beginSlice();
    for (int i = 0; i < n; ++i) lattice.slice(out + (size_t)i * vd);
    delete[] pos;
    delete[] val;
    return 0;
}

// Per-stage wall clock (seconds) of the reference engine, for the CPU baseline.
int ref_filter_timed(const float *ref, const float *src, int n, int d, int vd,
                     float *out, int *M_out, double *t_stage /* [4] init,splat,blur,slice */)
{
    struct timeval t[5];
    gettimeofday(t + 0, NULL);
    PermutohedralLattice lattice(d, vd, n);
    gettimeofday(t + 1, NULL);
    for (int i = 0; i < n; ++i)
        lattice.splat(const_cast<float *>(ref) + (size_t)i * d, const_cast<float *>(src) + (size_t)i * vd);
    gettimeofday(t + 2, NULL);
    lattice.blur();
    gettimeofday(t + 3, NULL);
    lattice.beginSlice();
    for (int i = 0; i < n; ++i) lattice.slice(out + (size_t)i * vd);
    gettimeofday(t + 4, NULL);
    if (M_out) *M_out = lattice.hashTable.size();
    for (int i = 0; i < 4; ++i)
        t_stage[i] = (t[i + 1].tv_sec - t[i].tv_sec) + (t[i + 1].tv_usec - t[i].tv_usec) / 1e6;
    return 0;
}

}  // extern "C"
