// Column statistics of row-major matrices in ONE launch (round 6): BatchNorm batch statistics, BatchNorm backward sums, bias gradients.
//   s[k][c] = sum_r f_k(r, c),  k < NST,  fp64, fixed order:  rows r = chunk start + lane, + lane count, ... per row lane; row lanes in order; chunks in
//   groups of consecutive chunks, the groups in order -- a result depends on (R, C, the launch shape) only, never on timing.
// grid (chunks, column groups), 256 threads.  A thread owns W consecutive columns (W = 4: one 16-byte load per operand and row; W = 1 for widths /
// strides / bases that are not multiples of 4) of one row lane; a block covers up to 64 such units, the 256 threads fold into as many row lanes as fit
// (C = 32, W = 4: 8 units x 32 row lanes -- the round-5 form gave a workgroup 64 columns x 4 row lanes whatever C was, scalar loads, 64 workgroups: a
// 32-wide layer over 80 000 rows ran ~30 us at 2 % of the memory system; 354 + 246 + 246 such launches per training step).  The workgroup of a column
// group that arrives last (common.h: etch_last_block) folds the chunks' partials (part[chunk][k][C]) and hands each column's sums to `fin(c, s)`.
#pragma once
#include "common.h"

#define CS_MAX_CHUNKS 64
static inline int colstat_chunks(long R) {
    long n = R / 64;
    return (int)(n < 1 ? 1 : (n > CS_MAX_CHUNKS ? CS_MAX_CHUNKS : n));
}
template <int W> static inline int colstat_groups(int C) { return (C + 64 * W - 1) / (64 * W); }

// F: struct with  template <int W> __device__ void load(long r, int c, double (&v)[NST][W]) const;   FIN: __device__ void operator()(int c, const double (&s)[NST])
template <int NST, int W, class F, class FIN>
__device__ __forceinline__ void colstat_run(long R, int C, double* part, unsigned* counters, const F& f, const FIN& fin) {
    __shared__ double red[NST * W * 256];
    const int tid = threadIdx.x, nch = gridDim.x;
    const int units = (C + W - 1) / W - 64 * blockIdx.y;                 // units of this column group (> 0 by the grid's construction)
    int ub = 1;
    while (ub < units && ub < 64) ub <<= 1;                              // power of two <= 64
    const int rlanes = 256 / ub, u = tid & (ub - 1), rl = tid / ub;
    const int c = (blockIdx.y * 64 + u) * W;
    const bool live = u < units && c < C;
    const long r0 = R * blockIdx.x / nch, r1 = R * (blockIdx.x + 1) / nch;
    double s[NST][W];
#pragma unroll
    for (int k = 0; k < NST; ++k)
#pragma unroll
        for (int j = 0; j < W; ++j) s[k][j] = 0.0;
    if (live)
        for (long r = r0 + rl; r < r1; r += 2 * rlanes) {
            double va[NST][W], vb[NST][W];
            const bool two = r + rlanes < r1;
            f.template load<W>(r, c, va);
            if (two) f.template load<W>(r + rlanes, c, vb);
#pragma unroll
            for (int k = 0; k < NST; ++k)
#pragma unroll
                for (int j = 0; j < W; ++j) { s[k][j] += va[k][j]; if (two) s[k][j] += vb[k][j]; }
        }
#pragma unroll
    for (int k = 0; k < NST; ++k)
#pragma unroll
        for (int j = 0; j < W; ++j) red[(k * W + j) * 256 + tid] = s[k][j];
    __syncthreads();
    // row lanes in order: thread (u, j-th column, k) -- ub * W * NST <= 512 sums of `rlanes` terms, spread over the 256 threads
    for (int e = tid; e < ub * W * NST; e += 256) {
        const int uu = e % ub, j = (e / ub) % W, k = e / (ub * W);
        const int cc = (blockIdx.y * 64 + uu) * W + j;
        if (uu < units && cc < C) {
            double t = 0.0;
            for (int l = 0; l < rlanes; ++l) t += red[(k * W + j) * 256 + l * ub + uu];
            part[((size_t)blockIdx.x * NST + k) * C + cc] = t;
        }
    }
    if (nch > 1 && !etch_last_block(counters + blockIdx.y, (unsigned)nch)) return;       // (one chunk: its partials are this workgroup's own writes)
    // the chunks: `lanes` groups of consecutive chunks per column (loads batched: after the acquire every partial comes from memory), groups in order
    const int ncols = (units < 64 ? units : 64) * W;                     // columns of this group (the tail clipped by cc < C below)
    int lanes = 1;
    while (lanes < 8 && 2 * lanes * ncols <= 256) lanes <<= 1;          // 1 / 2 / 4 / 8 chunk groups per column
    const int per = (nch + lanes - 1) / lanes;
    __syncthreads();
    for (int base = 0; base < ncols; base += 256 / lanes) {
        const int col = base + tid % (256 / lanes), ln = tid / (256 / lanes);
        const int cc = blockIdx.y * 64 * W + col;
        double t[NST];
#pragma unroll
        for (int k = 0; k < NST; ++k) t[k] = 0.0;
        if (col < ncols && cc < C && ln < lanes) {
            const int k0 = ln * per, k1 = (k0 + per < nch ? k0 + per : nch);
            for (int kk = k0; kk < k1; kk += 8) {
                double v[NST][8];
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int k = 0; k < NST; ++k) v[k][q] = kk + q < k1 ? part[((size_t)(kk + q) * NST + k) * C + cc] : 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int k = 0; k < NST; ++k) t[k] += v[k][q];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NST; ++k) red[k * 256 + tid] = t[k];
        __syncthreads();
        if (col < ncols && cc < C && ln == 0) {
            double sfin[NST];
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                double a = 0.0;
                for (int l = 0; l < lanes; ++l) a += red[k * 256 + l * (256 / lanes) + tid];
                sfin[k] = a;
            }
            fin(cc, sfin);
        }
    }
}

// 16-byte loads are possible when the width, the leading dimensions and the bases are multiples of 4 floats
static inline bool colstat_vec_ok(int C, long ld, const void* p0, const void* p1 = nullptr, const void* p2 = nullptr, long ld2 = 0) {
    return (C & 3) == 0 && (ld & 3) == 0 && (ld2 & 3) == 0 && ((uintptr_t)p0 & 15) == 0 && ((uintptr_t)p1 & 15) == 0 && ((uintptr_t)p2 & 15) == 0;
}
