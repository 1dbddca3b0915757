/*
 * oracle/discrete_ops.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's native index kernels (the CUDA sources cannot
 * be built or run in this image: no nvcc, no CUDA device).  Only tests/, bench.py's
 * cpu_baseline leg, __graft_entry__.smoke() and oracle/ref_harness may load this library.
 *
 * Arithmetic contract ("bit-defined fp32"): every squared distance is
 *     ((dx*dx) + (dy*dy)) + (dz*dz)   with dx = a - b, each op rounded to fp32,
 * i.e. the C expression of the reference with NO fused multiply-add.  Build with
 * -ffp-contract=off (see oracle/Makefile).  The HIP kernels use the same contract.
 *
 * Functions and the reference lines they follow (paths relative to /root/reference):
 *   orc_ball_query       external/vgtk/vgtk/cuda/grouping_cuda_kernel.cu:68-113, grouping_cuda.cpp:71-86
 *   orc_fps_vgtk         external/vgtk/vgtk/cuda/grouping_cuda_kernel.cu:340-466, grouping_cuda.cpp:158-173
 *   orc_gather_points    external/vgtk/vgtk/cuda/gathering_cuda_kernel.cu:43-68
 *   orc_knnquery         external/pointops/src/knnquery/knnquery_cuda_kernel.cu:21-108
 *   orc_fps_pointops     external/pointops/src/sampling/sampling_cuda_kernel.cu:5-171
 *   orc_opt_n_threads    grouping_cuda_kernel.cu:29-33 == pointops/src/cuda_utils.h:11-14
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ORC_FMA selects the contraction of a*a + b*b + c*c (default 0 = none, the contract of record).  The reference binary is
 * built by nvcc, whose default -fmad=true MAY contract these expressions; which form it picks is not knowable here, so the two
 * plausible ones are provided ONLY for tests/test_oracle_ops.py::test_fma_contraction_sensitivity, which counts how many index
 * decisions depend on the choice:  1: fma(c,c, fma(b,b, a*a))   2: fma(c,c, fma(a,a, b*b)). */
#ifndef ORC_FMA
#define ORC_FMA 0
#endif
static inline float sumsq3(float a, float b, float c) {
#if ORC_FMA == 1
    return fmaf(c, c, fmaf(b, b, a * a));
#elif ORC_FMA == 2
    return fmaf(c, c, fmaf(a, a, b * b));
#else
    float xx = a * a, yy = b * b, zz = c * c;
    float s = xx + yy;
    return s + zz;
#endif
}

static inline float sqdist(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return sumsq3(dx, dy, dz);
}

/* min(1024, 2^floor(log2 n)) computed exactly the way the host code does (double log ratio). */
int orc_opt_n_threads(int work_size) {
    const int pow_2 = (int)(log((double)work_size) / log(2.0));
    int v = 1 << pow_2;
    if (v > 1024) v = 1024;
    if (v < 1) v = 1;
    return v;
}

/* new_xyz (b,3,m), xyz (b,3,n) -> idx (b,m,nsample); idx is zero-initialised by the caller
 * in the reference (grouping_cuda.cpp:80-82) -- done here. */
void orc_ball_query(int b, int n, int m, float radius, int nsample,
                    const float* new_xyz, const float* xyz, int32_t* idx) {
    memset(idx, 0, sizeof(int32_t) * (size_t)b * m * nsample);
    const float radius2 = radius * radius;
    for (int bi = 0; bi < b; ++bi) {
        const float* X = xyz + (size_t)bi * 3 * n;
        const float* Q = new_xyz + (size_t)bi * 3 * m;
        int32_t* I = idx + (size_t)bi * m * nsample;
        for (int j = 0; j < m; ++j) {
            const float qx = Q[j], qy = Q[m + j], qz = Q[2 * m + j];
            int cnt = 0;
            for (int k = 0; k < n && cnt < nsample; ++k) {
                /* reference: (new_x - x)*(new_x - x) + ... */
                float d2 = sqdist(qx, qy, qz, X[k], X[n + k], X[2 * n + k]);
                if (d2 < radius2) { I[j * nsample + cnt] = k; ++cnt; }
            }
            if (cnt < nsample - 1) {
                for (int k = 0; k + cnt < nsample; ++k) I[j * nsample + k + cnt] = I[j * nsample + k];
            }
        }
    }
}

/* literal emulation of one FPS block: per-thread strided scan + LDS tree reduce */
static int fps_tree_reduce(float* dists, int* dists_i, int bs) {
    for (int s = bs / 2; s >= 1; s >>= 1) {
        for (int t = 0; t < s; ++t) {
            float v1 = dists[t], v2 = dists[t + s];
            int i1 = dists_i[t], i2 = dists_i[t + s];
            dists[t] = v1 > v2 ? v1 : v2;         /* max(v1, v2) */
            dists_i[t] = v2 > v1 ? i2 : i1;
        }
    }
    return dists_i[0];
}

/* xyz (b,3,n) -> idx (b,m).  temp starts at 1e10, points with |p|^2 <= 1e-3 are skipped
 * (the comparison is float-vs-double literal, i.e. done in double). */
void orc_fps_vgtk(int b, int n, int m, const float* xyz, int32_t* idx) {
    if (m <= 0) return;
    const int bs = orc_opt_n_threads(n);
    float* temp = (float*)malloc(sizeof(float) * n);
    float* dists = (float*)malloc(sizeof(float) * bs);
    int* dists_i = (int*)malloc(sizeof(int) * bs);
    for (int bi = 0; bi < b; ++bi) {
        const float* X = xyz + (size_t)bi * 3 * n;
        int32_t* out = idx + (size_t)bi * m;
        for (int k = 0; k < n; ++k) temp[k] = 1e10f;
        int old = 0;
        out[0] = 0;
        for (int j = 1; j < m; ++j) {
            const float x1 = X[old], y1 = X[n + old], z1 = X[2 * n + old];
            for (int tid = 0; tid < bs; ++tid) {
                int besti = 0; float best = -1.0f;
                for (int k = tid; k < n; k += bs) {
                    float x2 = X[k], y2 = X[n + k], z2 = X[2 * n + k];
                    float mag = sumsq3(x2, y2, z2);
                    if ((double)mag <= 1e-3) continue;
                    float d = sqdist(x2, y2, z2, x1, y1, z1);
                    float d2 = d < temp[k] ? d : temp[k];   /* min(d, temp[k]) */
                    temp[k] = d2;
                    besti = d2 > best ? k : besti;
                    best = d2 > best ? d2 : best;
                }
                dists[tid] = best; dists_i[tid] = besti;
            }
            old = fps_tree_reduce(dists, dists_i, bs);
            out[j] = old;
        }
    }
    free(temp); free(dists); free(dists_i);
}

/* Same sampling expressed with the closed-form tie rule the HIP kernel uses:
 * winner = max d2; ties -> smallest bit-reversed (k mod bs), then smallest k.
 * Kept in the oracle to prove (tests/test_oracle_ops.py) that the closed form equals the
 * literal tree emulation above, including under massive ties. */
static unsigned bitrev(unsigned v, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}
void orc_fps_vgtk_keyed(int b, int n, int m, const float* xyz, int32_t* idx) {
    if (m <= 0) return;
    const int bs = orc_opt_n_threads(n);
    int bits = 0; while ((1 << bits) < bs) ++bits;
    float* temp = (float*)malloc(sizeof(float) * n);
    for (int bi = 0; bi < b; ++bi) {
        const float* X = xyz + (size_t)bi * 3 * n;
        int32_t* out = idx + (size_t)bi * m;
        for (int k = 0; k < n; ++k) temp[k] = 1e10f;
        int old = 0; out[0] = 0;
        for (int j = 1; j < m; ++j) {
            const float x1 = X[old], y1 = X[n + old], z1 = X[2 * n + old];
            uint64_t bestkey = 0; int besti = 0;
            for (int k = 0; k < n; ++k) {
                float x2 = X[k], y2 = X[n + k], z2 = X[2 * n + k];
                float mag = sumsq3(x2, y2, z2);
                if ((double)mag <= 1e-3) continue;
                float d = sqdist(x2, y2, z2, x1, y1, z1);
                float d2 = d < temp[k] ? d : temp[k];
                temp[k] = d2;
                uint32_t fb; memcpy(&fb, &d2, 4);
                uint32_t tie = (bitrev((unsigned)(k % bs), bits) << 16) | (unsigned)(k / bs);
                uint64_t key = ((uint64_t)fb << 32) | (uint64_t)(0xFFFFFFFFu - tie);
                if (key > bestkey) { bestkey = key; besti = k; }
            }
            old = besti; out[j] = old;
        }
    }
    free(temp);
}

/* points (b,c,n), idx (b,m) -> out (b,c,m) */
void orc_gather_points(int b, int c, int n, int m, const float* points, const int32_t* idx, float* out) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci)
            for (int j = 0; j < m; ++j)
                out[((size_t)bi * c + ci) * m + j] = points[((size_t)bi * c + ci) * n + idx[(size_t)bi * m + j]];
}

/* grad_out (b,c,m), idx (b,m) -> grad_points (b,c,n): external/vgtk/vgtk/cuda/gathering_cuda_kernel.cu:73-98 (atomicAdd scatter into a
 * zeroed buffer, gathering_cuda.cpp:54-59); sequential here, i.e. repeated indices are summed in ascending m order */
void orc_gather_points_backward(int b, int c, int n, int m, const float* grad_out, const int32_t* idx, float* grad_points) {
    for (size_t e = 0; e < (size_t)b * c * n; ++e) grad_points[e] = 0.f;
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci)
            for (int j = 0; j < m; ++j)
                grad_points[((size_t)bi * c + ci) * n + idx[(size_t)bi * m + j]] += grad_out[((size_t)bi * c + ci) * m + j];
}

/* ---- pointops kNN: literal max-heap ---- */
static void reheap(float* dist, int* idx, int k) {
    int root = 0, child = 1;
    while (child < k) {
        if (child + 1 < k && dist[child + 1] > dist[child]) child++;
        if (dist[root] > dist[child]) return;
        float tf = dist[root]; dist[root] = dist[child]; dist[child] = tf;
        int ti = idx[root]; idx[root] = idx[child]; idx[child] = ti;
        root = child; child = root * 2 + 1;
    }
}
static void heap_sort(float* dist, int* idx, int k) {
    for (int i = k - 1; i > 0; i--) {
        float tf = dist[0]; dist[0] = dist[i]; dist[i] = tf;
        int ti = idx[0]; idx[0] = idx[i]; idx[i] = ti;
        reheap(dist, idx, i);
    }
}
/* xyz (n,3), new_xyz (m,3), offset (b), new_offset (b) -> idx (m,nsample), dist2 (m,nsample) */
void orc_knnquery(int b, int m, int nsample, const float* xyz, const float* new_xyz,
                  const int32_t* offset, const int32_t* new_offset, int32_t* idx, float* dist2) {
    float best_dist[100]; int best_idx[100];
    (void)b;
    for (int pt = 0; pt < m; ++pt) {
        int bt = 0; while (!(pt < new_offset[bt])) bt++;
        int start = bt == 0 ? 0 : offset[bt - 1];
        int end = offset[bt];
        const float qx = new_xyz[pt * 3], qy = new_xyz[pt * 3 + 1], qz = new_xyz[pt * 3 + 2];
        for (int i = 0; i < nsample; ++i) { best_dist[i] = 1e10f; best_idx[i] = start; }
        for (int i = start; i < end; ++i) {
            float d2 = sqdist(qx, qy, qz, xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]);
            if (d2 < best_dist[0]) {
                best_dist[0] = d2; best_idx[0] = i;
                reheap(best_dist, best_idx, nsample);
            }
        }
        heap_sort(best_dist, best_idx, nsample);
        for (int i = 0; i < nsample; ++i) { idx[(size_t)pt * nsample + i] = best_idx[i]; dist2[(size_t)pt * nsample + i] = best_dist[i]; }
    }
}

/* xyz (n,3), offset(b), new_offset(b), n_max = largest segment -> idx (new_offset[b-1]).
 * tmp (n) starts at 1e10 (pointops.py:22). */
void orc_fps_pointops(int b, int n_max, int n_total, const float* xyz, const int32_t* offset,
                      const int32_t* new_offset, int32_t* idx) {
    const int bs = orc_opt_n_threads(n_max);
    float* tmp = (float*)malloc(sizeof(float) * (n_total > 0 ? n_total : 1));
    float* dists = (float*)malloc(sizeof(float) * bs);
    int* dists_i = (int*)malloc(sizeof(int) * bs);
    for (int k = 0; k < n_total; ++k) tmp[k] = 1e10f;
    for (int bid = 0; bid < b; ++bid) {
        int start_n = bid == 0 ? 0 : offset[bid - 1], end_n = offset[bid];
        int start_m = bid == 0 ? 0 : new_offset[bid - 1], end_m = new_offset[bid];
        int old = start_n;
        /* the reference writes idx[start_m] unconditionally (sampling_cuda_kernel.cu:41) */
        if (start_m < new_offset[b - 1]) idx[start_m] = start_n;
        for (int j = start_m + 1; j < end_m; ++j) {
            const float x1 = xyz[old * 3], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
            for (int tid = 0; tid < bs; ++tid) {
                int besti = start_n; float best = -1.0f;
                for (int k = start_n + tid; k < end_n; k += bs) {
                    float d = sqdist(xyz[k * 3], xyz[k * 3 + 1], xyz[k * 3 + 2], x1, y1, z1);
                    float d2 = d < tmp[k] ? d : tmp[k];
                    tmp[k] = d2;
                    besti = d2 > best ? k : besti;
                    best = d2 > best ? d2 : best;
                }
                dists[tid] = best; dists_i[tid] = besti;
            }
            old = fps_tree_reduce(dists, dists_i, bs);
            idx[j] = old;
        }
    }
    free(tmp); free(dists); free(dists_i);
}
