// Training-side kernels of the two Point-Transformer nets: what /root/reference/src/train.py:77-101 obtains from autograd through
// /root/reference/src/models/pointtransformer_seg.py in train() mode.
//   etch_bn_stats              mean[c], var[c] (biased) of the rows of x (R,C)                         BatchNorm1d batch statistics (fp64 sums)
//   etch_bn_apply              y = act((x - mean[c]) * scale[c] + beta[c])                            BatchNorm1d applied (+ ReLU)
//   etch_bn_backward           dx, dgamma, dbeta of y = act(gamma * xhat + beta), train or eval mode
//   etch_rows_maxpool_backward d/dx of MaxPool1d(ns) over consecutive row groups (TransitionDown :66): the gradient goes to the FIRST maximum
//   etch_pt_softmax_agg        out[i, s*cs + j] = sum_k softmax_k(logit[i, k, j]) v[i, k, s*cs + j]   (pointtransformer_seg.py:34-36) and its backward
// Reductions are fixed-order (no atomics): gradients are reproducible run to run.
#include "common.h"
#include "colstat.h"

#define TO_CHUNKS 64

// part[chunk][stat][C]: per-chunk fp64 sums of NST statistics produced by `f(row, col, out[NST])`; block = 64 columns x 4 row lanes
template <int NST, class F>
__device__ __forceinline__ void colstat_partial(long R, int C, double* __restrict__ part, F f) {
    __shared__ double red[NST][256];
    const int tid = threadIdx.x, cl = tid & 63, rl = tid >> 6;
    const int c = blockIdx.y * 64 + cl;
    const long r0 = R * blockIdx.x / TO_CHUNKS, r1 = R * (blockIdx.x + 1) / TO_CHUNKS;
    double s[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) s[k] = 0.0;
    if (c < C)
        for (long r = r0 + rl; r < r1; r += 4) {
            double v[NST];
            f(r, c, v);
#pragma unroll
            for (int k = 0; k < NST; ++k) s[k] += v[k];
        }
#pragma unroll
    for (int k = 0; k < NST; ++k) red[k][tid] = s[k];
    __syncthreads();
    if (rl == 0 && c < C) {
#pragma unroll
        for (int k = 0; k < NST; ++k) part[((size_t)blockIdx.x * NST + k) * C + c] = ((red[k][cl] + red[k][64 + cl]) + red[k][128 + cl]) + red[k][192 + cl];
    }
}

__global__ void __launch_bounds__(256) bn_stats_partial_kernel(long R, int C, const float* __restrict__ x, long ldx, double* __restrict__ part) {
    colstat_partial<2>(R, C, part, [&](long r, int c, double (&v)[2]) {
        const double t = (double)x[r * ldx + c];
        v[0] = t; v[1] = t * t;
    });
}
__global__ void bn_stats_final_kernel(long R, int C, const double* __restrict__ part, float* __restrict__ mean, float* __restrict__ var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < TO_CHUNKS; ++k) { s1 += part[((size_t)k * 2 + 0) * C + c]; s2 += part[((size_t)k * 2 + 1) * C + c]; }
    const double m = s1 / (double)R;
    double v = s2 / (double)R - m * m;
    if (v < 0.0) v = 0.0;
    mean[c] = (float)m; var[c] = (float)v;
}

// Round 6: one launch for the statistics AND everything torch.nn.functional.batch_norm derives from them in train() mode -- the last workgroup of a
// column group (common.h: etch_last_block) sums the group's partials in chunk order (= bn_stats_final_kernel's sums, bit for bit) and writes mean,
// rstd = 1 / sqrt(var + eps), scale = gamma rstd, the running statistics (momentum < 0: the cumulative average 1 / num_batches_tracked) and
// num_batches_tracked.  Before: 2 launches here + 9 element-wise torch launches per BatchNorm call, 246 calls per training step.
struct BnStatsLoad {
    const float* x; long ldx;
    template <int W> __device__ __forceinline__ void load(long r, int c, double (&v)[2][W]) const {
        float t[W];
        if (W == 4) { const float4 q = *reinterpret_cast<const float4*>(x + r * ldx + c); t[0] = q.x; t[1] = q.y; t[2] = q.z; t[W - 1] = q.w; }
        else t[0] = x[r * ldx + c];
#pragma unroll
        for (int j = 0; j < W; ++j) { v[0][j] = (double)t[j]; v[1][j] = (double)t[j] * (double)t[j]; }
    }
};
template <int W>
__global__ void __launch_bounds__(256) bn_train_stats_kernel(long R, int C, const float* __restrict__ x, long ldx, double* part, unsigned* counters,
                                                             const float* __restrict__ gamma, float eps, float momentum, float* running_mean,
                                                             float* running_var, const long long* num_batches, float* __restrict__ mean,
                                                             float* __restrict__ rstd, float* __restrict__ scale) {
    colstat_run<2, W>(R, C, part, counters, BnStatsLoad{x, ldx}, [&](int c, const double (&s)[2]) {
        const double m = s[0] / (double)R;
        double v = s[1] / (double)R - m * m;
        if (v < 0.0) v = 0.0;
        const float mf = (float)m, vf = (float)v, rs = rsqrtf(vf + eps);
        mean[c] = mf; rstd[c] = rs; scale[c] = gamma[c] * rs;
        if (running_mean && running_var) {
            // num_batches is read only here (every column sees the value of before this call); the apply kernel behind this one counts the call
            const long long nb = num_batches ? *num_batches + 1 : 1;
            const float mom = momentum >= 0.f ? momentum : (float)(1.0 / (double)(nb > 0 ? nb : 1));
            const float unbiased = vf * (float)((double)R / (double)(R > 1 ? R - 1 : 1));
            running_mean[c] = fmaf(mom, mf, running_mean[c] * (1.f - mom));
            running_var[c] = fmaf(mom, unbiased, running_var[c] * (1.f - mom));
        }
    });
}

// centred before scaling: x * scale + (beta - mean * scale) cancels when |x - mean| << |x|
__global__ void __launch_bounds__(256) bn_apply_kernel(long n, int C, const float* __restrict__ x, long ldx, const float* __restrict__ mean,
                                                       const float* __restrict__ scale, const float* __restrict__ beta, int relu, float* __restrict__ y,
                                                       long long* count_call) {
    if (count_call && blockIdx.x == 0 && threadIdx.x == 0) *count_call += 1;          // num_batches_tracked (etch_bn_train_forward)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        const float v = fmaf(x[r * ldx + c] - mean[c], scale[c], beta[c]);
        y[i] = relu ? fmaxf(v, 0.f) : v;
    }
}

// g = dy * [y > 0] (relu) ; s1 = sum_r g ; s2 = sum_r g * xhat
__global__ void __launch_bounds__(256) bn_bwd_partial_kernel(long R, int C, const float* __restrict__ x, long ldx, const float* __restrict__ y,
                                                             const float* __restrict__ dy, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, int relu, double* __restrict__ part) {
    colstat_partial<2>(R, C, part, [&](long r, int c, double (&v)[2]) {
        float g = dy[r * C + c];
        if (relu && !(y[r * C + c] > 0.f)) g = 0.f;
        const float xh = (x[r * ldx + c] - mean[c]) * rstd[c];
        v[0] = (double)g; v[1] = (double)g * (double)xh;
    });
}
__global__ void bn_bwd_final_kernel(int C, const double* __restrict__ part, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < TO_CHUNKS; ++k) { s1 += part[((size_t)k * 2 + 0) * C + c]; s2 += part[((size_t)k * 2 + 1) * C + c]; }
    dbeta[c] = (float)s1; dgamma[c] = (float)s2;
}
// the two kernels above in one launch (the last workgroup of a column group sums its partials in chunk order)
struct BnBwdLoad {
    const float *x, *y, *dy, *mean, *rstd; long ldx; int C, relu;
    template <int W> __device__ __forceinline__ void load(long r, int c, double (&v)[2][W]) const {
        float g[W], xv[W], yv[W];
        if (W == 4) {
            const float4 q = *reinterpret_cast<const float4*>(dy + r * C + c), p = *reinterpret_cast<const float4*>(x + r * ldx + c);
            g[0] = q.x; g[1] = q.y; g[2] = q.z; g[W - 1] = q.w; xv[0] = p.x; xv[1] = p.y; xv[2] = p.z; xv[W - 1] = p.w;
            if (relu) { const float4 o = *reinterpret_cast<const float4*>(y + r * C + c); yv[0] = o.x; yv[1] = o.y; yv[2] = o.z; yv[W - 1] = o.w; }
        } else {
            g[0] = dy[r * C + c]; xv[0] = x[r * ldx + c];
            if (relu) yv[0] = y[r * C + c];
        }
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const float gg = relu && !(yv[j] > 0.f) ? 0.f : g[j];
            const float xh = (xv[j] - mean[c + j]) * rstd[c + j];
            v[0][j] = (double)gg; v[1][j] = (double)gg * (double)xh;
        }
    }
};
template <int W>
__global__ void __launch_bounds__(256) bn_bwd_stats_kernel(long R, int C, const float* __restrict__ x, long ldx, const float* __restrict__ y,
                                                           const float* __restrict__ dy, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, int relu, double* part, unsigned* counters,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta) {
    colstat_run<2, W>(R, C, part, counters, BnBwdLoad{x, y, dy, mean, rstd, ldx, C, relu}, [&](int c, const double (&s)[2]) {
        dbeta[c] = (float)s[0]; dgamma[c] = (float)s[1];
    });
}
// train: dx = gamma rstd (g - s1/R - xhat s2/R);  eval (statistics are constants): dx = gamma rstd g
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(long R, int C, const float* __restrict__ x, long ldx, const float* __restrict__ y,
                                                           const float* __restrict__ dy, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                                           const float* __restrict__ dbeta, int relu, int train, float* __restrict__ dx) {
    const long n = R * C;
    const float invR = (float)(1.0 / (double)R);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        float g = dy[i];
        if (relu && !(y[i] > 0.f)) g = 0.f;
        const float k = gamma[c] * rstd[c];
        if (train) {
            const float xh = (x[r * ldx + c] - mean[c]) * rstd[c];
            dx[i] = k * ((g - dbeta[c] * invR) - xh * (dgamma[c] * invR));
        } else {
            dx[i] = k * g;
        }
    }
}

// dy (m*ns, c): dout[i, ch] at the first row of group i where y attains its maximum, zero elsewhere
__global__ void __launch_bounds__(256) rows_maxpool_bwd_kernel(long m, int ns, int c, const float* __restrict__ y, const float* __restrict__ dout,
                                                               float* __restrict__ dy) {
    const long n = m * c;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const long i = e / c;
        const int ch = (int)(e - i * c);
        const float* yr = y + (size_t)i * ns * c + ch;
        float best = yr[0];
        int bj = 0;
        for (int j = 1; j < ns; ++j) {
            const float v = yr[(size_t)j * c];
            if (v > best) { best = v; bj = j; }
        }
        const float g = dout[e];
        for (int j = 0; j < ns; ++j) dy[((size_t)i * ns + j) * c + ch] = j == bj ? g : 0.f;
    }
}

// softmax over the ns neighbours per (point, shared channel j) + weighted aggregation; thread = (point, channel ch), j = ch % cs
__global__ void __launch_bounds__(256) pt_softmax_agg_kernel(long n, int ns, int c, int cs, const float* __restrict__ logit, const float* __restrict__ v,
                                                             float* __restrict__ sm, float* __restrict__ out) {
    const long tot = n * c;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long)gridDim.x * 256) {
        const long i = e / c;
        const int ch = (int)(e - i * c), j = ch % cs;
        const float* lg = logit + (size_t)i * ns * cs + j;
        float mx = lg[0];
        for (int k = 1; k < ns; ++k) mx = fmaxf(mx, lg[(size_t)k * cs]);
        float den = 0.f;
        for (int k = 0; k < ns; ++k) den += expf(lg[(size_t)k * cs] - mx);
        float acc = 0.f;
        for (int k = 0; k < ns; ++k) {
            const float p = expf(lg[(size_t)k * cs] - mx) / den;
            if (ch < cs) sm[((size_t)i * ns + k) * cs + j] = p;
            acc = fmaf(p, v[((size_t)i * ns + k) * c + ch], acc);
        }
        out[e] = acc;
    }
}
// dv[i,k,ch] = dout[i,ch] sm[i,k,j];  dsm[i,k,j] = sum_s dout[i,s cs+j] v[i,k,s cs+j];  dlogit = sm (dsm - sum_k sm dsm).  thread = (point, j)
__global__ void __launch_bounds__(256) pt_softmax_agg_bwd_kernel(long n, int ns, int c, int cs, const float* __restrict__ sm, const float* __restrict__ v,
                                                                 const float* __restrict__ dout, float* __restrict__ dlogit, float* __restrict__ dv) {
    const long tot = n * cs;
    const int share = c / cs;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long)gridDim.x * 256) {
        const long i = e / cs;
        const int j = (int)(e - i * cs);
        float dot = 0.f;
        for (int k = 0; k < ns; ++k) {
            const size_t row = (size_t)i * ns + k;
            const float p = sm[row * cs + j];
            float d = 0.f;
            for (int s = 0; s < share; ++s) {
                const int ch = s * cs + j;
                const float g = dout[(size_t)i * c + ch];
                d = fmaf(g, v[row * c + ch], d);
                dv[row * c + ch] = g * p;
            }
            dlogit[row * cs + j] = d;          // dsm for now
            dot = fmaf(p, d, dot);
        }
        for (int k = 0; k < ns; ++k) {
            const size_t o = ((size_t)i * ns + k) * cs + j;
            dlogit[o] = sm[o] * (dlogit[o] - dot);
        }
    }
}

static inline unsigned to_grid(long n) {
    long g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

extern "C" {

int etch_bn_stats(long R, int C, const float* x, long ldx, double* workspace, float* mean, float* var, void* stream) {
    if (R <= 0 || C <= 0 || !x || !workspace || !mean || !var || ldx < C) return ETCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(TO_CHUNKS, (C + 63) / 64), dim3(256), 0, st, R, C, x, ldx, workspace);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + 255) / 256), dim3(256), 0, st, R, C, workspace, mean, var);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_bn_apply(long R, int C, const float* x, long ldx, const float* mean, const float* scale, const float* beta, int relu, float* y, void* stream) {
    if (R <= 0 || C <= 0) return ETCH_OK;
    if (!x || !mean || !scale || !beta || !y || ldx < C) return ETCH_EINVAL;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(to_grid(R * C)), dim3(256), 0, (hipStream_t)stream, R * C, C, x, ldx, mean, scale, beta, relu, y, (long long*)nullptr);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_bn_backward(long R, int C, const float* x, long ldx, const float* y, const float* dy, const float* mean, const float* rstd, const float* gamma,
                     int relu, int train, double* workspace, float* dx, float* dgamma, float* dbeta, void* stream) {
    if (R <= 0 || C <= 0 || !x || !dy || !mean || !rstd || !gamma || !workspace || !dgamma || !dbeta || ldx < C || (relu && !y)) return ETCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(TO_CHUNKS, (C + 63) / 64), dim3(256), 0, st, R, C, x, ldx, y, dy, mean, rstd, relu, workspace);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + 255) / 256), dim3(256), 0, st, C, workspace, dgamma, dbeta);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    if (dx) {
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(to_grid(R * C)), dim3(256), 0, st, R, C, x, ldx, y, dy, mean, rstd, gamma, dgamma, dbeta, relu, train, dx);
        ETCH_RETURN_IF_LAUNCH_FAILED();
    }
    return ETCH_OK;
}

int etch_bn_train_forward(long R, int C, const float* x, long ldx, const float* gamma, const float* beta, float eps, float momentum,
                          float* running_mean, float* running_var, long long* num_batches, int relu, double* workspace, unsigned* counters,
                          float* mean, float* rstd, float* scale, float* y, void* stream) {
    if (R <= 0 || C <= 0 || !x || !gamma || !beta || !workspace || !counters || !mean || !rstd || !scale || !y || ldx < C) return ETCH_EINVAL;
    if ((C + 63) / 64 > ETCH_REDUCE_COUNTERS) return ETCH_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (colstat_vec_ok(C, ldx, x))
        hipLaunchKernelGGL(bn_train_stats_kernel<4>, dim3(colstat_chunks(R), colstat_groups<4>(C)), dim3(256), 0, st, R, C, x, ldx, workspace, counters, gamma, eps,
                           momentum, running_mean, running_var, num_batches, mean, rstd, scale);
    else
        hipLaunchKernelGGL(bn_train_stats_kernel<1>, dim3(colstat_chunks(R), colstat_groups<1>(C)), dim3(256), 0, st, R, C, x, ldx, workspace, counters, gamma, eps,
                           momentum, running_mean, running_var, num_batches, mean, rstd, scale);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(bn_apply_kernel, dim3(to_grid(R * C)), dim3(256), 0, st, R * C, C, x, ldx, mean, scale, beta, relu, y, num_batches);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_bn_backward_fused(long R, int C, const float* x, long ldx, const float* y, const float* dy, const float* mean, const float* rstd,
                           const float* gamma, int relu, int train, double* workspace, unsigned* counters, float* dx, float* dgamma, float* dbeta,
                           void* stream) {
    if (R <= 0 || C <= 0 || !x || !dy || !mean || !rstd || !gamma || !workspace || !counters || !dgamma || !dbeta || ldx < C || (relu && !y)) return ETCH_EINVAL;
    if ((C + 63) / 64 > ETCH_REDUCE_COUNTERS) return ETCH_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (colstat_vec_ok(C, ldx, x, dy, relu ? y : nullptr))
        hipLaunchKernelGGL(bn_bwd_stats_kernel<4>, dim3(colstat_chunks(R), colstat_groups<4>(C)), dim3(256), 0, st, R, C, x, ldx, y, dy, mean, rstd, relu, workspace,
                           counters, dgamma, dbeta);
    else
        hipLaunchKernelGGL(bn_bwd_stats_kernel<1>, dim3(colstat_chunks(R), colstat_groups<1>(C)), dim3(256), 0, st, R, C, x, ldx, y, dy, mean, rstd, relu, workspace,
                           counters, dgamma, dbeta);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    if (dx) {
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(to_grid(R * C)), dim3(256), 0, st, R, C, x, ldx, y, dy, mean, rstd, gamma, dgamma, dbeta, relu, train, dx);
        ETCH_RETURN_IF_LAUNCH_FAILED();
    }
    return ETCH_OK;
}

int etch_rows_maxpool_backward(long m, int ns, int c, const float* y, const float* dout, float* dy, void* stream) {
    if (m <= 0 || c <= 0) return ETCH_OK;
    if (ns <= 0 || !y || !dout || !dy) return ETCH_EINVAL;
    hipLaunchKernelGGL(rows_maxpool_bwd_kernel, dim3(to_grid(m * c)), dim3(256), 0, (hipStream_t)stream, m, ns, c, y, dout, dy);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_pt_softmax_agg(long n, int ns, int c, int cs, const float* logit, const float* v, float* sm, float* out, void* stream) {
    if (n <= 0) return ETCH_OK;
    if (ns <= 0 || c <= 0 || cs <= 0 || c % cs || !logit || !v || !sm || !out) return ETCH_EINVAL;
    hipLaunchKernelGGL(pt_softmax_agg_kernel, dim3(to_grid(n * c)), dim3(256), 0, (hipStream_t)stream, n, ns, c, cs, logit, v, sm, out);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_pt_softmax_agg_backward(long n, int ns, int c, int cs, const float* sm, const float* v, const float* dout, float* dlogit, float* dv, void* stream) {
    if (n <= 0) return ETCH_OK;
    if (ns <= 0 || c <= 0 || cs <= 0 || c % cs || !sm || !v || !dout || !dlogit || !dv) return ETCH_EINVAL;
    hipLaunchKernelGGL(pt_softmax_agg_bwd_kernel, dim3(to_grid(n * cs)), dim3(256), 0, (hipStream_t)stream, n, ns, c, cs, sm, v, dout, dlogit, dv);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

}  // extern "C"
