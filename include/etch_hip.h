/*
 * etch_hip.h -- C ABI of libetch_hip.so, the MI355X (gfx950) implementation of the ETCH hot path.
 *
 * Every entry point takes plain device pointers + sizes, an explicit HIP stream (hipStream_t passed
 * as void*) and returns an int status: 0 = ok, >0 = hipError_t of the failed launch, -1 = invalid
 * argument, -2 = unsupported size.  The caller owns all memory; kernels are asynchronous on `stream`.
 * No torch types appear in any signature.
 *
 * Each function cites the reference interface it replaces (paths relative to the ETCH repository).
 * INTEGRATION.md shows the binding a reference maintainer would add.
 */
#ifndef ETCH_HIP_H
#define ETCH_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- epn_grouping / epn_gathering (external/vgtk/vgtk/cuda) ------------------------------------ */

/* Replaces epn_grouping.ball_query: grouping_cuda.cpp:71-86 -> ball_query_cuda_kernel
 * grouping_cuda_kernel.cu:68-113.  new_xyz (b,3,m) f32, xyz (b,3,n) f32 -> idx (b,m,nsample) i32.
 * First `nsample` support indices (ascending) with d2 < radius^2; rows with cnt < nsample-1 are padded
 * cyclically with their own prefix, the last slot stays 0 when cnt == nsample-1 (reference quirk). */
int etch_ball_query(int b, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz,
                    int* idx, void* stream);

/* Replaces epn_grouping.furthest_point_sampling: grouping_cuda.cpp:158-173 ->
 * furthest_point_sampling_cuda_kernel grouping_cuda_kernel.cu:352-466.  xyz (b,3,n) f32 -> idx (b,m) i32.
 * Start index 0, points with |p|^2 <= 1e-3 never selected, reference tie-breaking reproduced exactly. */
int etch_furthest_point_sampling(int b, int n, int m, const float* xyz, int* idx, void* stream);

/* Replaces epn_gathering.gather_points_forward: gathering_cuda.cpp:29-46 -> gathering_cuda_kernel.cu:43-68.
 * points (b,c,n) f32, idx (b,m) i32 -> out (b,c,m) f32. */
int etch_gather_points(int b, int c, int n, int m, const float* points, const int* idx, float* out, void* stream);

/* ---- pointops_cuda (external/pointops/src) ------------------------------------------------------ */

/* Replaces knnquery_cuda_launcher: knnquery/knnquery_cuda_kernel.h:10-16, kernel .cu:65-108.
 * xyz (n,3), new_xyz (m,3), offset/new_offset (b) cumulative i32 -> idx (m,nsample) i32, dist (m,nsample) f32.
 * `m_max` = largest number of queries in one segment (grid sizing; known to the host that built the
 * offsets).  write_sqrt != 0 stores sqrt(d2) (what pointops.py:43 returns), else d2.  nsample <= 28. */
int etch_knnquery(int b, int m_max, int nsample, const float* xyz, const float* new_xyz, const int* offset,
                  const int* new_offset, int* idx, float* dist, int write_sqrt, void* stream);

/* Replaces furthestsampling_cuda_launcher: sampling/sampling_cuda_kernel.h:10-16, kernel .cu:15-129.
 * xyz (n,3), offset/new_offset (b) -> idx (new_offset[b-1]) i32 (global indices).  n_max = largest segment.
 * The reference's caller-provided `tmp` scratch is not needed (distances live in registers). */
int etch_furthestsampling(int b, int n_max, const float* xyz, const int* offset, const int* new_offset, int* idx,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ETCH_HIP_H */
