// Launchers of the fused ZoomOut iteration (dm_zoomfuse.hip), called by dm_zoomout.hip.
#pragma once
#include "dm_internal.h"

template <typename TR>
struct zo_embed_args {
    const TR* Phi1; long long s1; int ld1; int N1;
    const double* C; long long strideC; int ldc;        // (B, >= 16 nrb, ldc): zero outside the current k x k block
    int k, nrb;                                         // map size, ceil(k / 16)
    int D, ldS, R1;                                     // halves written per source row (>= 32 nrb, zero beyond), row stride, rows per pair
    _Float16* Fy; float* bias; double* n1; int N1pad; double* embr; int Kpad;
    const double* amaxT; int nT;                        // partial maxima of the target operand (its scale sx)
    const unsigned long long* amax_prev;                // (B) bits of max |emb1| the scale sy of this call is derived from
    unsigned long long* amax_cur;                       // (B) bits of max |emb1| of this call (atomicMax; zeroed by the caller)
    unsigned int* bmax;                                 // (B) max |bias| (float bits, atomicMax; zeroed by the caller)
    int only_max;                                       // the pre-pass: amax_cur only
    int dbg;                                            // DM_EXPERIMENTS builds only (0 in the product): ablations, WRONG results
};
template <typename TR>
int dm_zo_embed_split(dm_ctx* ctx, int B, const zo_embed_args<TR>& a);

template <typename TR>
struct zo_mx_args {
    dm_simnn_queue q;                                // pb / pj / ps, nparts, pw, Npad of the tile pass (the flag_* members unused)
    const float* tnorm2; const unsigned int* smax2; const unsigned int* bmax; float tau_scale;
    const unsigned long long* amax_prev; const unsigned long long* amax_cur;
    const TR* Phi2; int ld2;
    const double* embr; int Kpad; const double* n1; int N1pad;
    int K, N2, N1;
    int32_t* nn;
    int* qrow; float* qthr;                          // (B N2) the iteration's queue: rows whose margin is inside the bound + their thresholds
    unsigned int* qcount; int qcap;                  // its length (zeroed by the caller), its capacity B N2
    int dbg;                                         // DM_EXPERIMENTS builds only (WRONG results): 1 no exact launch work, 2 first two kept blocks of a row only, 4 every block reads the pair's first rows
};
template <typename TR>
int dm_zo_merge_exact(dm_ctx* ctx, int B, const zo_mx_args<TR>& a);

// amax[b * nch + chunk] = max |Phi[b][i][c]|, rows i = chunk (mod nch), c < k
template <typename TR>
int dm_zo_absmax_rows(dm_ctx* ctx, int B, int N, int k, const TR* Phi, int ld, int nch, double* amax);
// dst[b][r][c] = src[b][r][c] for r < rows, c < cols (float64 matrices with their own row / pair strides)
int dm_zo_copy_mat(dm_ctx* ctx, int B, int rows, int cols, const double* src, int lds, long long ss, double* dst, int ldd, long long sd);
