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

/* The same samplers with every scan split over G workgroups (2 <= G <= 8; large scans: one compute unit's distance update bounds the
 * round): bit-identical picks (unique 64-bit keys, grouping_cuda_kernel.cu:352-466 / sampling_cuda_kernel.cu:15-129 tie-breaking).  The
 * G workgroups of a scan exchange one candidate per round through `workspace` (etch_fps_split_workspace_bytes(nseg, G) bytes, zeroed by the
 * call) and must be co-resident: callers keep nseg * G at or below the number of compute units.  A workgroup that waits longer than the
 * spin limit sets the fail word (etch_fps_split_failed, synchronises the stream) and the scan's remaining indices become INT_MIN.
 * etch_fps_split_debug(spin_limit, drop_group): test hook (0 / -1 = defaults). */
int etch_fps_split_workspace_bytes(int nseg, int G);
int etch_furthest_point_sampling_split(int b, int n, int m, const float* xyz, int* idx, int G, void* workspace, void* stream);
int etch_furthestsampling_split(int b, int n_max, const float* xyz, const int* offset, const int* new_offset, int* idx, int G, void* workspace,
                                void* stream);
int etch_fps_split_failed(const void* workspace, int* failed, void* stream);
int etch_fps_split_debug(unsigned spin_limit, int drop_group);
/* A/B switch of the one-workgroup kernels' branch-free slot loop (default 1; 0 = the 64-bit-key loop everywhere): same picks either way. */
int etch_fps_fast(int enable);

/* Replaces epn_gathering.gather_points_forward: gathering_cuda.cpp:29-46 -> gathering_cuda_kernel.cu:43-68.
 * points (b,c,n) f32, idx (b,m) i32 -> out (b,c,m) f32. */
int etch_gather_points(int b, int c, int n, int m, const float* points, const int* idx, float* out, void* stream);

/* Replaces epn_gathering.gather_points_backward: gathering_cuda.cpp:45-60 -> gathering_cuda_kernel.cu:73-98 (the
 * first of the training-side ops): grad_out (b,c,m), idx (b,m) -> grad_points (b,c,n), every element written (no zero-fill needed);
 * contributions of a repeated index are summed in ascending m order (the reference's atomicAdd order is unspecified). */
int etch_gather_points_backward(int b, int c, int n, int m, const float* grad_out, const int* idx, float* grad_points, void* stream);

/* ---- pointops_cuda (external/pointops/src) ------------------------------------------------------ */

/* Replaces knnquery_cuda_launcher: knnquery/knnquery_cuda_kernel.h:10-16, kernel .cu:65-108.
 * xyz (n,3), new_xyz (m,3), offset/new_offset (b) cumulative i32 -> idx (m,nsample) i32, dist (m,nsample) f32.
 * `m_max` = largest number of queries in one segment, `m_total` = total number of queries (grid sizing; known to the
 * host that built the offsets).  m_total > 0 selects the wave-per-query kernel (64-wide distance evaluation, the
 * reference's sequential heap updates preserved); m_total == 0 the thread-per-query kernel.  Results are identical.
 * write_sqrt != 0 stores sqrt(d2) (what pointops.py:43 returns), else d2.  nsample <= 100 (the reference's best_dist[100],
 * knnquery_cuda_kernel.cu:86-87) on the wave-per-query kernel, <= 28 on the thread-per-query form. */
int etch_knnquery(int b, int m_max, int m_total, int nsample, const float* xyz, const float* new_xyz, const int* offset,
                  const int* new_offset, int* idx, float* dist, int write_sqrt, void* stream);

/* Replaces furthestsampling_cuda_launcher: sampling/sampling_cuda_kernel.h:10-16, kernel .cu:15-129.
 * xyz (n,3), offset/new_offset (b) -> idx (new_offset[b-1]) i32 (global indices).  n_max = largest segment.
 * The reference's caller-provided `tmp` scratch is not needed (distances live in registers). */
int etch_furthestsampling(int b, int n_max, const float* xyz, const int* offset, const int* new_offset, int* idx,
                          void* stream);
/* Round 6: the vgtk FPS of a batch (grouping_cuda_kernel.cu:352-466, as etch_furthest_point_sampling) and the pointops FPS of the SAME scans
 * (sampling_cuda_kernel.cu:15-129, as etch_furthestsampling; every segment n points) in one launch of 2 b workgroups -- both chains are one workgroup per
 * scan and depend on the coordinates only; bit-identical picks. */
int etch_fps_pair(int b, int n, int m, const float* xyz_b3n, int* idx_a, const float* xyz_packed, const int* offset, const int* new_offset, int* idx_b,
                  void* stream);

/* etch_knnquery without any host knowledge of the offsets: every query finds its segment on the device by scanning new_offset the way
 * the reference kernel does (knnquery_cuda_kernel.cu:52-62 get_bt_idx, :74-80), the grid is sized from m alone.  Same results as
 * etch_knnquery; this is the form a binding calls when the offsets only exist on the device (no .tolist() / .item() sync). */
int etch_knnquery_dev(int m, int nsample, const float* xyz, const float* new_xyz, const int* offset, const int* new_offset, int* idx,
                      float* dist, int write_sqrt, void* stream);
/* The same with the number of segments given (entries of offset / new_offset: known on the host without a device read): the per-query segment
 * scan is clamped to it.  etch_knnquery_dev keeps the reference launcher's contract (m == new_offset[nseg - 1]). */
int etch_knnquery_dev_bounded(int m, int nsample, int nseg, const float* xyz, const float* new_xyz, const int* offset, const int* new_offset,
                              int* idx, float* dist, int write_sqrt, void* stream);

/* ---- the reference's own launcher symbols --------------------------------------------------------------
 * Exactly the names and signatures the reference's host wrappers link against, so knnquery_cuda.cpp / sampling_cuda.cpp (and a
 * pointer-level wrapper of the vgtk kernels) build against libetch_hip.so unchanged.  Like the reference they launch on the NULL
 * stream (`<<<blocks, threads, 0>>>`), return void, and report a failed launch on stderr.  dist2 receives SQUARED distances. */

/* external/pointops/src/knnquery/knnquery_cuda_kernel.h:14 (definition knnquery_cuda_kernel.cu:111-116). */
void knnquery_cuda_launcher(int m, int nsample, const float* xyz, const float* new_xyz, const int* offset, const int* new_offset, int* idx,
                            float* dist2);
/* external/pointops/src/sampling/sampling_cuda_kernel.h:14 (definition sampling_cuda_kernel.cu:131-171).  n = largest segment
 * (pointops.py:20-24); tmp (n_total) is accepted and left untouched (the running minimum distances live in registers). */
void furthestsampling_cuda_launcher(int b, int n, const float* xyz, const int* offset, const int* new_offset, float* tmp, int* idx);
/* Pointer-level forms of the vgtk launches, argument order of the kernels they start:
 * ball_query_cuda_kernel<<<b, opt_n_threads(m)>>>(b,n,m,radius,nsample,new_xyz,xyz,idx)   grouping_cuda_kernel.cu:68-73,476-481
 * furthest_point_sampling_cuda_kernel<<<nb, n_threads>>>(nb,nq,m,source,temp,idx)         grouping_cuda_kernel.cu:352-354,637-642
 * gather_points_forward_kernel / gather_points_backward_kernel                            gathering_cuda_kernel.cu:43-98,103-165 */
void ball_query_cuda_launcher(int b, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz, int* idx);
void furthest_point_sampling_cuda_launcher(int b, int n, int m, const float* dataset, float* temp, int* idxs);
void gather_points_forward_cuda_launcher(int b, int c, int n, int m, const float* points, const int* idx, float* out);
void gather_points_backward_cuda_launcher(int b, int c, int n, int m, const float* grad_out, const int* idx, float* grad_points);

/* ---- dense per-point layers ----------------------------------------------------------------------- */

/* Y[r,o] = epi(sum_k X[rowmap(r),k] * W[o,k]) on the fp32 matrix cores.  Replaces the un-fused ATen calls
 * torch.nn.Linear / Conv1d(k=1) / Conv2d(1x1) + eval BatchNorm + ReLU used all over the path
 * (src/models/direction_backbones.py:33-35,53-57, src/models/pointtransformer_seg.py:15-22,44-49,66,104-122,144-145,
 * src/models/so3conv.py:166,181).  epi: v = acc + bias[o]; if scale: v = v*scale[o] + shift[o];
 * res_mode 1: v += res[r,o]; act (0 none, 1 relu, 2 leaky 0.01); res_mode 2: v += res[r,o].
 * rowmap: identity when row_idx == NULL; else q = r / grp, src = ((q / p_out) * p_in + row_idx[q]) * grp + r % grp
 * (grp = 1, p_in = 0 gives a plain global row gather).  X, W need not be padded: any K / ld is accepted. */
int etch_linear(int R, int K, int O, const float* X, long ldx, const int* row_idx, int grp, int p_in, int p_out,
                const float* W, long ldw, const float* bias, const float* scale, const float* shift, int act,
                const float* res, long ldr, int res_mode, float* Y, long ldy, void* stream);

/* ---- EPN encoder (channels-last activations F[b][p][60][c]) -------------------------------------------- */

/* Fused inter-SO(3) convolution.  Replaces inter_so3conv_grouping_anchor + inter_so3conv_feat_grouping
 * (external/vgtk/vgtk/so3conv/functional.py:286-324, 61-67) + BasicSO3Conv (modules.py:33-39): the
 * [b,p2,60,24,nn] kernel-weight tensor is never materialised.
 * xyz (b,3,p1), new_xyz (b,3,p2), ball_idx (b,p2,nn) i32, feats (b,p1,60,cin), rk (60,24,3) = anchors @ kernels^T,
 * W (cout, cin*24) reference layout, Wp = etch fragment order of W (used when cin % 16 == 0), bias (cout)
 * -> out (b,p2,60,cout), pre-normalisation. */
int etch_inter_so3conv(int b, int cin, int cout, int p1, int p2, int nn, float sigma, const float* xyz,
                       const float* new_xyz, const int* ball_idx, const float* feats, const float* rk, const float* W,
                       const float* Wp, const float* bias, float* out, void* stream);

/* The same convolution with an explicit processing order of the output points (a scheduling hint: results are identical).
 * order (b,p2) int32 = a permutation of 0..p2-1 per scan, e.g. from etch_spatial_order; NULL = index order.  Workgroups walk
 * `order`, one contiguous eighth per XCD, so that the workgroups sharing an L2 gather from the same source rows.
 * stat_part (b,p2,2,cout) fp64 or NULL: per output point the sum and the sum of squares of its 60 x cout outputs, i.e. the
 * InstanceNorm2d statistics of so3conv.py:96-99 without a second pass over the output (etch_instnorm_from_partials finishes them);
 * accumulated in fp64 so that a channel whose mean dominates its spread keeps its variance.
 * Round 6: cin == 1 accepts feats == NULL for ALL-ONES features (the encoder's occupancy input, so3conv.py:7-16): the neighbours' feature rows are then
 * neither gathered nor multiplied in. */
int etch_inter_so3conv_ordered(int b, int cin, int cout, int p1, int p2, int nn, float sigma, const float* xyz, const float* new_xyz,
                               const int* ball_idx, const float* feats, const float* rk, const float* W, const float* Wp,
                               const float* bias, float* out, const int* order, double* stat_part, void* stream);

#ifdef ETCH_BUILD_EXPERIMENTS
/* The same convolution on the 32x32x2 fp32 MFMA (0.986 of the matrix peak on this chip against 0.85 for 16x16x4,
 * profiles/r03_mfma_issue_rate.txt), two output points per workgroup; (cin, cout) in {(32,32), (32,64), (64,64)}, nn <= 64.
 * Wp32[slice = 3 h + g][mt][kp][u][lane][s] = W[32 mt + lane % 32][(32 h + kp * NU + u) * 24 + 8 g + 4 (lane / 32) + s] with
 * NKP = 8 / (cout / 32) K-shares kp of NU = 32 / NKP steps u (etch_amd/ops.py inter_weight_frag32).  order / stat_part as above. */
int etch_inter_so3conv32(int b, int cin, int cout, int p1, int p2, int nn, float sigma, const float* xyz, const float* new_xyz,
                         const int* ball_idx, const float* feats, const float* rk, const float* Wp32, const float* bias, float* out,
                         const int* order, double* stat_part, void* stream);
#endif

/* The same convolution with its second contraction (W x X1, 1/2 - 2/3 of the flops) on the bf16 matrix cores: every fp32 operand split exactly
 * into three bf16 values (8 + 8 + 8 mantissa bits), the six largest cross products accumulated in fp32 -- the error against fp64 of the fp32
 * MFMA (profiles/r03_bf16x3_split.txt) at 2.3 x its rate.  cin, cout multiples of 16.  Wq: 3 * cout * cin * 24 bf16 =
 * [chunk of 32 kappas][o tile][plane hi / mid / lo][lane][8] in the kernel's contraction order (etch_amd/ops.py inter_weight_split). */
int etch_inter_so3conv_split(int b, int cin, int cout, int p1, int p2, int nn, float sigma, const float* xyz, const float* new_xyz,
                             const int* ball_idx, const float* feats, const float* rk, const void* Wq, const float* bias, float* out,
                             const int* order, double* stat_part, void* stream);

/* 1 if (cin, cout, nn) is a shape the planes kernels (etch_inter_so3conv_planes_kq; with ETCH_BUILD_EXPERIMENTS also the round-4 forms) are built for. */
int etch_inter_so3conv_planes_supported(int cin, int cout, int nn);

#ifdef ETCH_BUILD_EXPERIMENTS      /* round-4 kernels, retired from the default build in round 6 (a sibling instantiation mis-executed for a cause that was never named: profiles/r05_x32_cin32_root_cause.txt) */
/* The same convolution with BOTH contractions on the bf16 matrix cores (round 4, etch_amd/csrc/so3conv_x.hip): the neighbour contraction
 * X1[k, c] = sum_n w[a, k, n] F[idx_n, a, c] (functional.py:61-67 with the weights of :286-324 generated in registers) runs as split-operand
 * products too.  The gathered rows come as three bf16 planes written once by their producer: feats_planes (b, p1, 60, 3, cin) bf16 =
 * hi / mid / lo of the fp32 features (exact: hi + mid + lo == value; etch_split3_planes or etch_instnorm_act_add_planes write them), loaded
 * global -> LDS directly and read back as matrix-core fragments with the transposing LDS read.  (cin, cout) in {(32,32), (32,64), (64,64)},
 * nn in {32, 64} (etch_inter_so3conv_planes_supported).  Wq as for etch_inter_so3conv_split but in W's NATURAL column order
 * (etch_amd/ops.py inter_weight_split(natural=True)).  order / stat_part as above.  Same result as the fp32 kernels up to the order of the
 * fp32 sums. */
int etch_inter_so3conv_planes(int b, int cin, int cout, int p1, int p2, int nn, float sigma, const float* xyz, const float* new_xyz,
                              const int* ball_idx, const void* feats_planes, const float* rk, const void* Wq, const float* bias, float* out,
                              const int* order, double* stat_part, void* stream);
/* The same on v_mfma_f32_32x32x16_bf16, the bf16 shape that issues at the matrix peak on this chip (profiles/r04_mfma_bf16_issue_rates.txt; the 16x16x32
 * shape of the entry above reaches 0.55 - 0.6 of it).  Wq32: 3 * cout * cin * 24 bf16 = [K step of 16][o tile of 32][plane][lane][8] in the kernel's
 * physical contraction order (etch_amd/ops.py inter_weight_split32).  Same shapes, same result up to the order of the fp32 sums. */
int etch_inter_so3conv_planes32(int b, int cin, int cout, int p1, int p2, int nn, float sigma, const float* xyz, const float* new_xyz,
                                const int* ball_idx, const void* feats_planes, const float* rk, const void* Wq32, const float* bias, float* out,
                                const int* order, double* stat_part, void* stream);
#endif

/* Round 5 (etch_amd/csrc/so3conv_y.hip): the inter conv of functional.py:286-324, :61-67 + modules.py:33-39 rebuilt around three measurements of
 * this round (profiles/r05_inter_conv_latency_bound.txt, r05_f16_two_plane_split.txt):
 *   (1) kernel weights off the MATRIX cores: the pre-activation 1 - |g_n - R_a kappa_k|^2 / sigma = [a_n, 1, G_n] . [1, b_k, r_ak] is a rank-5
 *       bilinear form whose six largest exactly-split bf16 cross terms fill one pair of v_mfma_f32_32x32x16_bf16 per (anchor, 32 neighbours);
 *   (2) both contractions on v_mfma_f32_32x32x16_f16 with TWO fp16 planes per operand (x = h + l, h = fp16(x), l = fp16(x - h), both to nearest:
 *       22 significant bits, absolute floor 2^-24; three cross products): for the O(1) operands of this path the error of the fp32 MFMA, with half
 *       the matrix instructions, 4 instead of 6 bytes per gathered element and 40 % of the split work of the three-plane bf16 split;
 *   (3) gathered rows through a register ring (plain loads, several chunks in flight per wave) instead of LDS-direct loads.
 * feats_planes (b, p1, 60, 2, cin) fp16 = etch_split2_planes_f16 / etch_instnorm_act_add_planes_f16 of the fp32 features; kq =
 * etch_inter_kpoint_operand(sigma, rk): [60 anchors][2 K steps][64 lanes][8] bf16 (122 880 bytes), the kernel-point factor in B-fragment order, built
 * once per layer from rk [60][24][3] = anchors @ kernel points (functional.py:296); Wq: 2 * cout * cin * 24 fp16 = the two planes of W, every row (output
 * channel) times the power of two that puts its maximum into [8, 16), in the physical contraction order of etch_inter_so3conv_planes32,
 * [K step of 16][o tile of 32][plane][lane][8]; wsc (cout floats) = the rows' inverse powers (etch_amd/ops.py inter_weight_split32_f16 and its `.wsc`).
 * fsc: NULL (the planes are those of the features as they stand: InstanceNorm outputs, unit scale by construction) or b floats, scan s's outputs
 * (before the bias) are multiplied by fsc[s] (etch_split2_planes_f16_scaled).  Round 6: persistent workgroups (one grid of 2 x CUs workgroups walks the
 * b * p2 output points, XCD-aware, dynamic beyond the first two points of a workgroup) -- results are bit for bit those of one workgroup per point.
 * Covers every shape of etch_inter_so3conv_planes_supported; order / stat_part as for etch_inter_so3conv.  Same result as the fp32 kernels to
 * ~1e-6 of the output scale (tests/test_gpu_r05.py holds it to the fp32 kernel AND to fp64 under the entitled-error rule). */
int etch_inter_kpoint_operand(float sigma, const float* rk, void* kq, void* stream);
int etch_inter_so3conv_planes_kq(int b, int cin, int cout, int p1, int p2, int nn, float sigma, const float* xyz, const float* new_xyz,
                                 const int* ball_idx, const void* feats_planes, const void* kq, const void* Wq32, const float* wsc, const float* fsc,
                                 const float* bias, float* out, const int* order, double* stat_part, void* stream);

/* x (rows, C) fp32 -> planes (rows, 2, C) fp16: h = fp16(x), l = fp16(x - h), both to nearest (the gather format of etch_inter_so3conv_planes_kq). */
int etch_split2_planes_f16(long rows, int C, const float* x, void* planes, void* stream);
/* Features of unknown scale (callers of the operator API, vgtk so3conv/modules.py:92-128 has no domain restriction): x (b, rows, C) fp32 -> planes
 * (b, rows, 2, C) fp16 of x[s] * 2^k(s), k(s) = the power of two that brings scan s's largest magnitude into [8, 16) (exact), and fsc[s] = 2^-k(s): pass
 * `fsc` to etch_inter_so3conv_planes_kq, whose epilogue multiplies scan s's outputs by it.  mx: b unsigned words of workspace, zeroed by the caller. */
int etch_split2_planes_f16_scaled(int b, long rows, int C, const float* x, void* mx, void* planes, float* fsc, void* stream);

/* x (rows, C) fp32 -> planes (rows, 3, C) bf16: the exact split x = hi + mid + lo (8 + 8 + 8 mantissa bits, by truncation). */
int etch_split3_planes(long rows, int C, const float* x, void* planes, void* stream);

/* mean / rstd (b,C) of InstanceNorm2d(affine=False, eps 1e-5) from per-part partial sums: partial (b,nparts,2,C) fp64, each over `count`
 * values.  Same result as etch_instnorm_stats over the full tensor (fp64 accumulation, fixed order). */
int etch_instnorm_from_partials(int b, int nparts, int C, int count, const double* partial, float* mean, float* rstd, void* stream);

/* Morton (Z-curve) order of each scan's points on its own bounding box, ties by index: xyz (b,3,n) -> order (b,n) int32, a
 * permutation of 0..n-1 per scan.  n <= 16384: 10 bits per axis; larger scans: 5 bits per axis, sorted in independent slices of
 * 32768 points.  No counterpart in the reference: it only schedules etch_inter_so3conv_ordered / etch_prop_interp_ordered. */
int etch_spatial_order(int b, int n, const float* xyz, int* order, void* stream);

/* Fused intra-SO(3) convolution.  Replaces intra_so3conv_grouping (functional.py:331-378) + BasicSO3Conv
 * (modules.py:150-153).  X (b,p,60,c); if mean != NULL the input is first normalised per (b,c) and passed
 * through leaky_relu(0.01) on load (InstanceNorm of the preceding inter block, src/models/so3conv.py:96-99).
 * intra_idx (60,12) i32; Wp = fragment order of W2[o][tap*c + ch] = W[o][ch*12 + tap]; -> Y (b,p,60,cout). */
int etch_intra_so3conv(int b, int c, int cout, int p, const float* X, const float* mean, const float* rstd,
                       const int* intra_idx, const float* Wp, const float* bias, float* Y, void* stream);
/* The same with the InstanceNorm partial sums of Y from the epilogue: stat_part (b * p/2, 2, cout) = per workgroup (2 points) the sum and
 * the sum of squares per output channel; finish with etch_instnorm_from_partials(b, p/2, cout, 120, ...).  p must be even; NULL = plain. */
int etch_intra_so3conv_stats(int b, int c, int cout, int p, const float* X, const float* mean, const float* rstd,
                             const int* intra_idx, const float* Wp, const float* bias, float* Y, double* stat_part, void* stream);
/* The same convolution on the 32x32x2 fp32 MFMA (0.986 of the matrix peak on this chip against 0.85 for 16x16x4, profiles/r03_mfma_issue_rate.txt);
 * c = cout in {32, 64}; Wp32[t][mt][lane][s] = W2[32 mt + lane % 32][8 t + 4 (lane / 32) + s] with W2 as above.  stat_part as above (p even) or NULL. */
int etch_intra_so3conv32(int b, int c, int cout, int p, const float* X, const float* mean, const float* rstd, const int* intra_idx,
                         const float* Wp32, const float* bias, float* Y, double* stat_part, void* stream);

/* The same convolution, weight-stationary on the bf16 matrix cores (csrc/so3conv_ws.hip): fp32 operands split exactly into three bf16 values, six
 * cross products accumulated in fp32 (the fp32 MFMA's error against fp64, profiles/r03_bf16x3_split.txt); W lives in registers for the whole launch,
 * persistent workgroups walk point pairs.  c = cout in {32, 64}; Wq = etch_amd/ops.py intra_weight_split:
 * [mt][kq][K step][plane hi / mid / lo][lane][8 bf16] of W2 above.  stat_part as etch_intra_so3conv_stats (p even) or NULL. */
int etch_intra_so3conv_split(int b, int c, int cout, int p, const float* X, const float* mean, const float* rstd, const int* intra_idx,
                             const void* Wq, const float* bias, float* Y, double* stat_part, void* stream);

/* Round 5: the same kernel on v_mfma_f32_32x32x16_f16 with TWO fp16 planes per operand (h = fp16(x), l = fp16(x - h), both to nearest; three cross
 * products; for this path's unit-scale operands the fp32 MFMA's error against fp64, profiles/r05_f16_two_plane_split.txt).  Wqh = etch_amd/ops.py
 * intra_weight_split_f16: [mt][kq][K step][plane h / l][lane][8 fp16] of W2 with every row (output channel) times its own power of two; wsc (cout
 * floats) = the inverse powers, applied by the kernel's epilogue (exact). */
int etch_intra_so3conv_f16(int b, int c, int cout, int p, const float* X, const float* mean, const float* rstd, const int* intra_idx,
                           const void* Wqh, const float* wsc, const float* bias, float* Y, double* stat_part, void* stream);

/* InstanceNorm2d(affine=False, eps=1e-5) statistics over (p,a) per (b,c) (src/models/so3conv.py:24,85,168).
 * x (b,rows,C) -> mean (b,C), rstd (b,C).  workspace: etch_instnorm_stats_workspace_bytes(b, C) bytes. */
int etch_instnorm_stats(int b, int rows, int C, const float* x, double* workspace, float* mean, float* rstd, void* stream);
int etch_instnorm_stats_workspace_bytes(int b, int C);

/* out = leaky_relu((x1-m1)*r1) [+ leaky_relu((x2-m2)*r2)]: norm + activation + skip add of
 * SeparableSO3ConvBlock.forward (src/models/so3conv.py:178-182).  x2 may be NULL. */
int etch_instnorm_act_add(int b, int rows, int C, const float* x1, const float* m1, const float* r1, const float* x2,
                          const float* m2, const float* r2, float* out, void* stream);
/* The same, additionally writing the result as three bf16 planes (b, rows, 3, C) for etch_inter_so3conv_planes (planes may be NULL). */
int etch_instnorm_act_add_planes(int b, int rows, int C, const float* x1, const float* m1, const float* r1, const float* x2,
                                 const float* m2, const float* r2, float* out, void* planes, void* stream);
/* The same with two fp16 planes (b, rows, 2, C) for etch_inter_so3conv_planes_kq. */
int etch_instnorm_act_add_planes_f16(int b, int rows, int C, const float* x1, const float* m1, const float* r1, const float* x2,
                                     const float* m2, const float* r2, float* out, void* planes, void* stream);
/* Round 6 -- the same with the second branch a ONE-channel 1x1 conv + InstanceNorm folded in (the skip branch of the encoder's first block,
 * so3conv.py:171-183 with dim_in = 1): f [b][rows] holds the branch's input value per row; slope / offset [b][C] = w_c rstd_c and (bias_c - mean_c) rstd_c
 * of its conv output s = w_c f + bias_c (mean_c = w_c mean(f) + bias_c, var_c = w_c^2 var(f): the conv output and its statistics pass are never made):
 *   out = lrelu((x1 - m1) r1) + lrelu(f slope_c + offset_c);  planes (may be NULL): the two fp16 planes of out. */
int etch_instnorm_act_add_k1_planes_f16(int b, int rows, int C, const float* x1, const float* m1, const float* r1, const float* f, const float* slope,
                                        const float* offset, float* out, void* planes, void* stream);

/* ---- feature propagation + direction head --------------------------------------------------------------- */

/* 3-NN of PointFeatPropagation (src/models/pointnet2_utils.py:45-70): xyz1 (B,N,3), xyz2 (B,3,S) ->
 * idx (B,N,3) i32 (ascending distance), weight (B,N,3) = normalised 1/(d2+1e-8) with d2 from the
 * reference's expansion formula -2xy + |x|^2 + |y|^2 in fp32. */
int etch_prop3nn(int B, int N, int S, const float* xyz1, const float* xyz2, int* idx, float* weight, void* stream);

/* Weighted gather (pointnet2_utils.py:71) + anchor mean (models_pointcloud.py:184): feats (B,S,A,C) channels-last
 * -> out (B,N,A,C), inv (B,N,C) = mean over A.  C in {32,64,128}. */
int etch_prop_interp(int B, int N, int S, int A, int C, const float* feats, const int* idx, const float* weight,
                     float* out, float* inv, void* stream);
/* The same with a processing order of the fine points (scheduling hint, results identical): order (B,N) int32 from
 * etch_spatial_order, NULL = index order. */
int etch_prop_interp_ordered(int B, int N, int S, int A, int C, const float* feats, const int* idx, const float* weight,
                             float* out, float* inv, const int* order, void* stream);

/* DotProdAttention of MultiHeadAttention (src/models/direction_backbones.py:102-129,160-194) for 60 tokens,
 * 8 heads x 8 dims: rows [T*60][ld] hold q/k/v at column offsets qoff/koff/voff -> out rows [T*60][ldo] (64 cols). */
int etch_mhsa_attention(long T, const float* qkv, long ld, int qoff, int koff, int voff, float* out, long ldo, void* stream);
/* The same attention core for the other encoder depths (models_pointcloud.py:34-48: embedding_dim 32 / 64 / 128 / 256 -> 8 heads of
 * 4 / 8 / 16 / 32, direction_backbones.py:151; logits / sqrt(head size), :125).  out (T*60, ldo) = concatenated heads. */
int etch_mhsa_attention_dim(long T, int embedding_dim, const float* qkv, long ld, int qoff, int koff, int voff, float* out, long ldo,
                            void* stream);

/* One whole MultiHeadAttention layer (direction_backbones.py:132-194; 64 dims, 8 heads x 8, 60 tokens per point) in a single
 * kernel: q/k/v transforms (no bias, weights (64,64) row-major as in key/query/value_transform.weight) -> per-head softmax
 * attention -> head_combine (Wc (64,64), bc (64)).  X, out: (T*60, 64) contiguous.
 * mode 0: out = X + att Wc^T + bc  (the residual form StackedMHSA.forward :216-221 applies to all but the last layer)
 * mode 1: out = att Wc^T + bc       mode 2: out = att (concatenated heads; Wc/bc unused -- head_combine folded downstream) */
int etch_mhsa_layer(long T, const float* X, const float* Wq, const float* Wk, const float* Wv, const float* Wc, const float* bc,
                    int mode, float* out, void* stream);
/* Round 5: the LAST attention layer of the direction head with the folded tail inside (models_pointcloud.py:115-117): out (T, 60) =
 * v . relu(Wf att + bf) + c per token, the anchor weights that so3_mean consumes -- the (T*60, 64) attention output and the (T*60, 128) hidden layer
 * never reach HBM.  Wfq = the two fp16 planes of Wf with every hidden unit's row times its own power of two, the powers folded into tab (ops.dirtail_weight_split /
 * dirtail_constants) (Wf 128 x 64 = direction_predictor.net[0] o head_combine, folded on the host) as matrix-core A
 * fragments [4 waves][4 K steps][2 planes][64 lanes][8] (etch_amd/ops.py dirtail_weight_split); tab = [bf (128) | v (128) | c] fp32 (257 floats). */
int etch_mhsa_layer_dirtail(long T, const float* X, const float* Wq, const float* Wk, const float* Wv, const void* Wfq, const float* tab,
                            float* out, void* stream);

/* The first MultiHeadAttention layer of the direction head applied to 3-NN INTERPOLATED tokens (PointFeatPropagation,
 * pointnet2_utils.py:45-74, feeding StackedMHSA, direction_backbones.py:216-221) without writing them out: F (B,S,60,64) coarse
 * tokens, idx / weight (B,N,3) from etch_prop3nn, order (B,N) int32 or NULL (scheduling only), weights as etch_mhsa_layer
 * -> out (B*N*60, 64) = X + att Wc^T + bc with X = the interpolated tokens (mode 0 of etch_mhsa_layer on etch_prop_interp's output).
 * sched: workspace of B*N*8 int32 (16-byte aligned).
 * etch_token_mean: X (T,A,C) -> mean over the A tokens (T,C) (the anchor mean of models_pointcloud.py:184 at the coarse points). */
int etch_mhsa_interp_layer(int B, int N, int S, const float* F, const int* idx, const float* weight, const int* order, const float* Wq,
                           const float* Wk, const float* Wv, const float* Wc, const float* bc, float* out, int* sched, void* stream);
int etch_token_mean(long T, int A, int C, const float* X, float* mean, void* stream);

/* y[r] = x[r,:K] . w + bias: so3_reg Conv1d(128,1,1) (src/models/models_pointcloud.py:54,117). */
int etch_rowdot(long R, int K, const float* x, long ldx, const float* w, float bias, float* y, void* stream);

/* so3_mean (src/models/so3conv.py:186-225) + R @ [0,0,1] (models_pointcloud.py:120-124): w (T,60), anchors (60,3,3)
 * -> dir (T,3); optional R (T,3,3) and singular values sv (T,3) (NULL to skip). */
int etch_so3_mean_dir(long T, int A, const float* w, const float* anchors, float* dir, float* R, float* sv, void* stream);
/* Backward of etch_so3_mean_dir (autograd through so3conv.py:186-225 + models_pointcloud.py:120-124 in train.py:77-85): ddir (T,3) = dL/d dir
 * -> dw (T,A) = dL/d w.  Derivative of the polar factor through the 3x3 Sylvester equation in V's basis, fp64 per point. */
int etch_so3_mean_dir_backward(long T, int A, const float* w, const float* anchors, const float* ddir, float* dw, void* stream);
/* Backward of the 3-NN propagation (pointnet2_utils.py:45-74; forward etch_prop_interp) and of any weighted row gather:
 * dst (nseg,C)[q] = sum_{k in [seg[q], seg[q+1])} wgt[perm[k]] * src[perm[k] / fan], perm = stable sort of the flattened (rows, fan) index
 * list by target row, seg its offsets (int64): scatter-add in a fixed order.  C % 4 == 0. */
int etch_weighted_segment_sum_rows(long nseg, int C, int fan, const float* src, const float* wgt, const long long* perm, const long long* seg,
                                   float* dst, void* stream);

/* ---- Point-Transformer heads (src/models/pointtransformer_seg.py) ------------------------------------------ */

/* PointTransformerLayer.forward after the q/k/v Linear layers (pointtransformer_seg.py:28-36; grouping as
 * src/models/pointops.py:79-100).  p (n,3); xq/xk/xv rows with leading dimension ldq; idx (n,ns) kNN indices;
 * params = 14 device pointers: linear_p[0] W(3x3), b(3); BN(3) folded scale, shift; linear_p[3] W(c,3), b(c);
 * linear_w[0] BN(c) scale, shift; linear_w[2] W TRANSPOSED (c, c/8), b(c/8); linear_w[3] BN(c/8) scale, shift;
 * linear_w[5] W(c/8,c/8), b(c/8); then an optional output BN(c) scale, shift (+ReLU) = block.bn2 (:117), NULL to skip
 * (16 pointers in total).  -> out (n,c) rows with leading dimension ldo. */
int etch_pt_attention(int n, int c, int ns, const float* p, const float* xq, const float* xk, const float* xv, long ldq,
                      const int* idx, const float* const* params, float* out, long ldo, void* stream);

/* The same layer split around the matrix cores (the schedule the model uses): prep builds the attention-MLP input
 * w_in[(i,j),:] = relu(bn(x_k[idx] - x_q + p_r)) (pointtransformer_seg.py:31-33 up to the first Linear), the two Linear
 * layers of linear_w run through etch_linear over all n*ns rows, aggregate applies softmax over the neighbours and the
 * shared-plane weighted sum (:34-36) [+ block.bn2 + ReLU].  params: the same 16 pointers as etch_pt_attention. */
int etch_pt_attn_prep(int n, int c, int ns, const float* p, const float* xq, const float* xk, long ldq, const int* idx,
                      const float* const* params, float* w_in, void* stream);
int etch_pt_attn_aggregate(int n, int c, int ns, const float* p, const float* xv, long ldq, const int* idx, const float* logits,
                           const float* const* params, float* out, long ldo, void* stream);

/* The same attention core in one matrix-core kernel: the c -> c/8 -> c/8 MLP of linear_w runs as two chained MFMA products on
 * w_in rows built in registers from the gathered k rows; softmax over the neighbours and the aggregation of (v + p_r) follow in
 * the same wave.  W2 = linear_w[2].weight (c/8, c) row-major (params[8] holds its transpose for the other variants).
 * (c, ns) in {(64,8), (128,8), (64,16), (128,16), (256,16), (512,16)}, otherwise ETCH_EUNSUPPORTED (-2). */
int etch_pt_attention_mfma(int n, int c, int ns, const float* p, const float* xq, const float* xk, const float* xv, long ldq,
                           const int* idx, const float* const* params, const float* W2, float* out, long ldo, void* stream);

#ifdef ETCH_BUILD_EXPERIMENTS      /* lab records, measured slower than the default path; built only with ETCH_BUILD_EXPERIMENTS=1 (etch_amd/build.py) */
/* PointTransformerBlock (src/models/pointtransformer_seg.py:101-122) in two kernels, consecutive blocks of a level chained:
 * K1: qkv (n,3c) = relu(bn1(x W1^T)) Wqkv^T + bqkv   -- linear1 (no bias) -> bn1 (folded s1,t1) -> ReLU -> linear_q|k|v of transformer2.
 * c in {64,128,256,512}. */
int etch_pt_block_k1(int n, int c, const float* x, long ldx, const float* W1, const float* s1, const float* t1, const float* Wqkv,
                     const float* bqkv, float* qkv, long ldq, void* stream);
/* K2: out (n,c) = relu(bn3(relu(bn2(vector attention(q|k|v, idx))) W3^T) + x)  -- transformer2's attention core (as
 * etch_pt_attention_mfma; params[14], params[15] = bn2 folded, required), bn2, ReLU, linear3, bn3, residual, ReLU; the attention output
 * stays in LDS.  tail = {W3 (c,c), s3, t3, x (residual, row stride ldx), W1', s1', t1', Wqkv', bqkv'}: when tail[4] != NULL the K1 of the
 * NEXT block runs on the output tile and fills qkv_next (n,3c).  (c, ns) in {(64,8),(128,8),(128,16),(256,16),(512,16)}. */
int etch_pt_block_k2(int n, int c, int ns, const float* p, const float* xq, const float* xk, const float* xv, long ldq, const int* idx,
                     const float* const* params, const float* W2, const float* const* tail, long ldx, float* out, long ldo,
                     float* qkv_next, long ldqn, void* stream);
#endif

/* queryandgroup(use_xyz=True) rows for TransitionDown (pointtransformer_seg.py:61, pointops.py:90-98):
 * out[(i*ns+j)] = [p[idx[i,j]] - new_p[i] | x[idx[i,j]] | 0...], row stride ldo >= 3+c (padding columns zeroed). */
int etch_pt_group(int m, int ns, int c, const float* p, const float* new_p, const float* x, long ldx, const int* idx,
                  float* out, long ldo, void* stream);

/* Row gather n_p = p[idx] (pointtransformer_seg.py:60): out[i,:c] = x[idx[i],:c]. */
int etch_gather_rows(int m, int c, const float* x, long ldx, const int* idx, float* out, void* stream);

/* TransitionDown with stride (pointtransformer_seg.py:40-68) without the m*ns grouped rows: ux = x Wx^T per SOURCE point (n, co),
 * Wx = linear.weight[:, 3:]; this kernel adds the coordinate part Wp (p_j - p_i), Wp = linear.weight[:, :3] (co,3), applies the
 * folded BatchNorm + ReLU and max-pools over the ns neighbours:  out[i,o] = max_j relu(bn(ux[idx[i,j],o] + Wp[o].(p[idx[i,j]] - new_p[i]))). */
int etch_pt_down_gather_max(int m, int ns, int co, const float* ux, long ldu, const float* p, const float* new_p, const int* idx,
                            const float* Wp, const float* scale, const float* shift, float* out, void* stream);

/* MaxPool1d(nsample) over consecutive row groups (pointtransformer_seg.py:63): x (m*ns, c) -> out (m, c). */
int etch_rows_maxpool(int m, int ns, int c, const float* x, float* out, void* stream);

/* TransitionUp tail (pointtransformer_seg.py:97, pointops.py:164-178): out = a + sum_k w_k f[idx_k],
 * w = (1/(dist+1e-8))/sum with the NON-squared kNN distances.  a,out (n,c); f (m,c); idx,dist (n,3). */
int etch_pt_interp_add(int n, int c, const float* a, const float* f, const int* idx, const float* dist, float* out, void* stream);

/* TransitionUp head branch (pointtransformer_seg.py:83-93): per-segment mean, and [x | g[segment]] concat. */
int etch_seg_mean(int nseg, int c, const float* x, const int* offset, float* mean, void* stream);
int etch_concat_bcast(int n, int c, int nseg, const float* x, const float* g, const int* offset, float* out, void* stream);

/* confi[2]: Conv1d(G*J, G, 1, groups=G) (pointtransformer_seg.py:145): out[r,g] = h[r, g*J:(g+1)*J] . w[g] + b[g]. */
int etch_grouped_dot(long R, int G, int J, const float* h, long ldh, const float* w, const float* bias, float* out, long ldo,
                     void* stream);

/* Fused Conv1d(K, G*J, 1) -> ReLU -> Conv1d(G*J, G, 1, groups=G) of the confidence head (pointtransformer_seg.py:145,
 * applied at :183-189; G = k markers, J = K = 128) and of the direction-head tail MLP.net[0] -> ReLU -> (net[2] o so3_reg)
 * (models_pointcloud.py:115-117, direction_backbones.py:38-75; G = 1, K = 64, J = 128):
 *   out[r,g] = b2[g] + sum_{j<J} relu(X[r,:] . W[g*J+j,:] + b1[g*J+j]) * w2[g*J+j]
 * The (R x G*J) hidden activations never leave the chip.  X (R,K) row stride ldx; W (G*J,K) row stride ldw; Wp = optional
 * copy of W in MFMA fragment order [G][K/16][8][64][4] (Wp[g][t][s][l][e] = W[g*J+16s+l%16][16t+4(l/16)+e]; NULL -> W is
 * read directly); out (R,G) row stride ldo.  J must be 128, K 64 or 128; ldx % 4 == 0, 16-byte aligned pointers. */
int etch_linear_relu_dot(long R, int K, int G, int J, const float* X, long ldx, const float* W, long ldw, const float* Wp,
                         const float* b1, const float* w2, const float* b2, float* out, long ldo, void* stream);

/* The same chain on the bf16 matrix cores with exactly split fp32 operands (three bf16 values per fp32 value, six cross products accumulated in
 * fp32: the fp32 MFMA's error against fp64, profiles/r03_bf16x3_split.txt).  K in {32, 64, 128, 256}, J = 128.  Wq = etch_amd/ops.py
 * lrd_weight_split: [g][K/32][8 strips][plane hi / mid / lo][lane][8 bf16]. */
int etch_linear_relu_dot_split(long R, int K, int G, int J, const float* X, long ldx, const void* Wq, const float* b1, const float* w2,
                               const float* b2, float* out, long ldo, void* stream);

/* Round 5: the weight-stationary form of the same chain on v_mfma_f32_16x16x32_f16 with TWO fp16 planes per operand (three cross terms; half the matrix
 * instructions).  Neither operand carries a known scale: each row of X is staged times the power of two that puts its maximum into [8, 16), each row of
 * W (hidden unit) likewise on the host; both powers leave in the epilogue's fmaf with the bias (exact).  Wqh = etch_amd/ops.py lrd_weight_split_f16;
 * wsc (G * J floats) = the weight rows' inverse powers.  K in {32, 64, 128}, G = 1 or G >= 8, b1 / w2 / wsc 16-byte aligned; other shapes:
 * ETCH_EUNSUPPORTED (the caller keeps etch_linear_relu_dot_split / etch_linear_relu_dot). */
int etch_linear_relu_dot_f16(long R, int K, int G, int J, const float* X, long ldx, const void* Wqh, const float* wsc, const float* b1, const float* w2,
                             const float* b2, float* out, long ldo, void* stream);

/* confidence = sum_g softmax(logits)_g * v_g (pointtransformer_seg.py:183-189): (R,G),(R,G) -> (R). */
int etch_softmax_dot(long R, int G, const float* logits, const float* v, float* out, void* stream);

/* so3_mean with the reference's own signature (src/models/so3conv.py:186-225): Rs (T,A,3,3) [rs_stride = A*9] or one rotation set
 * shared by all rows [rs_stride = 0], weights w (T,A) or NULL (uniform) -> chordal-L2 mean rotation R (T,3,3). */
int etch_so3_mean(long T, int A, const float* Rs, long rs_stride, const float* w, float* R, void* stream);

/* ---- un-fused operator forms of the reference's functional API (never on the hot path; csrc/functional_ops.hip) ---------- */

/* inter_so3conv_grouping_anchor (external/vgtk/vgtk/so3conv/functional.py:286-324): grouped_xyz (b,3,p,nn), rotated kernel points
 * (na,ks,3) = anchors @ kernels^T -> w (b,p,na,ks,nn) = relu(1 - |g - R_a kappa_k|^2 / sigma). */
int etch_inter_kernel_weights(int b, int p, int nn, int na, int ks, const float* grouped_xyz, const float* rotated_kernels, float sigma,
                              float* w, void* stream);

/* inter_so3conv_feat_grouping (functional.py:61-67): idx (b,p,nn) i32 into the q rows of feats (b,c,q,na) [q includes the shadow
 * row of add_shadow_feature, :101-105], w (b,p,na,ks,nn) -> (b,c,ks,p,na). */
int etch_inter_feat_grouping(int b, int c, int q, int p, int nn, int na, int ks, const int* idx, const float* w, const float* feats,
                             float* out, void* stream);

/* intra_so3conv_grouping (functional.py:331-378): intra_idx (na,nt) i64, feat (b,c,p,na) -> (b,c,nt,p,na). */
int etch_intra_grouping(int b, int c, int p, int na, int nt, const long long* intra_idx, const float* feat, float* out, void* stream);

/* square_distance (src/models/pointnet2_utils.py:4-23): src (B,N,C), dst (B,M,C) -> (B,N,M), expansion formula in the reference's
 * operation order. */
int etch_square_distance(int B, int N, int M, int C, const float* src, const float* dst, float* out, void* stream);

/* index_points (pointnet2_utils.py:26-43): points (B,N,C), idx (B,S) i64 -> (B,S,C). */
int etch_index_points(int B, int N, long S, int C, const float* points, const long long* idx, float* out, void* stream);

/* ---- backward kernels of the encoder's native ops (csrc/backward.hip).  The reference gets these gradients from
 * torch.autograd through the un-fused forms (vgtk/so3conv/functional.py:224-324, :61-67; modules.py:33-39, :131-153; src/models/
 * so3conv.py:24-44; call sites src/train.py:77-101).  All reductions run in a fixed order (no atomics). ---------------------------- */

/* Grouped features of the inter conv, recomputed for output points [p_begin, p_begin + pc): x1 (b, pc, 60, cin*24) with
 * x1[.., a, c*24+k] = sum_n feats[b, idx[b,p,n], a, c] * w[p,a,k,n]  (feats channels-last (b,p1,60,cin); w regenerated from xyz). */
int etch_inter_x1_rows(int b, int cin, int p1, int p2, int p_begin, int pc, int nn, float sigma, const float* xyz, const float* new_xyz,
                       const int* ball_idx, const float* feats, const float* rk, float* x1, void* stream);

/* C (M,N) (+)= A (R,M)^T B (R,N) (dW = dY^T X1: the weight gradients, sums over every row of the batch) on the fp64 matrix cores
 * (v_mfma_f64_16x16x4_f64: fp32 operands widened exactly, exact products, fp64 accumulation -- these sums cancel to ~1e-7 of their terms for the
 * direction head, profiles/r05_weight_gradient_accumulation.txt): split over R, fp64 partial tiles in `workspace`
 * (etch_gemm_tn_workspace_floats(R, M, N) floats), summed in split order, rounded to fp32 once. */
int etch_gemm_tn_workspace_floats(long R, int M, int N);
int etch_gemm_tn(long R, int M, int N, const float* A, long lda, const float* B, long ldb, float* C, int accumulate, float* workspace,
                 void* stream);
/* etch_gemm_tn in ONE launch: the workgroup of a 64 x 64 tile of C that finishes last sums the tile's partials in split order (bit for bit
 * etch_gemm_tn's result).  counters: 64 unsigned, zero before the first call, left zero (see etch_bn_train_forward); more than 64 tiles or more
 * than 64 row ranges per tile (long-and-narrow products): the two-launch form. */
int etch_gemm_tn_fused(long R, int M, int N, const float* A, long lda, const float* B, long ldb, float* C, int accumulate, float* workspace,
                       unsigned* counters, void* stream);

/* d feats of the inter conv from d x1 of the output points [p_begin, p_begin + pc): dfeats (b,p1,60,cin) (+)= the gather-side sum over
 * the neighbour slots (p,n) with idx[b,p,n] == q, taken in slot order.  cin in {4..64}, multiple of 4. */
int etch_inter_dfeat(int b, int cin, int p1, int p2, int p_begin, int pc, int nn, float sigma, const float* xyz, const float* new_xyz,
                     const int* ball_idx, const float* rk, const float* dx1, float* dfeats, int accumulate, void* stream);
/* The same gradient from the target side (round 6): contrib [b][pc][nn][60][cin] receives, per neighbour slot (p, n) of the chunk, the row
 * sum_k w[p,a,k,n] dX1[(p,a), c*24+k]; d feats[b, q] = the sum of the rows of the slots with ball_idx == q, which the caller takes with
 * etch_segment_sum_rows over a stable sort of the slots by source point (fixed order: reproducible).  Reads every dX1 tile once instead of nn times.
 * nn <= 64; cin as for etch_inter_dfeat.  A slot with ball_idx < 0 gets a zero row. */
int etch_inter_dfeat_slots(int b, int cin, int p1, int p2, int p_begin, int pc, int nn, float sigma, const float* xyz, const float* new_xyz,
                           const int* ball_idx, const float* rk, const float* dx1, float* contrib, void* stream);

/* Gathered operand of the intra conv's weight gradient: x (points,60,C) -> xg (points,60,nt,C) with xg[p,a,t,:] = x[p,intra_idx[a,t],:]
 * (functional.py:331-378 in channels-last rows). */
int etch_intra_rows(long points, int C, int nt, const int* intra_idx, const float* x, float* xg, void* stream);

/* Backward of the vector-attention core of etch_pt_attention (pointtransformer_seg.py:28-36, eval-mode BatchNorm as folded constants, no
 * output BN): arguments as etch_pt_attention (params[0..13]) + dout (n,c) rows with leading dimension lddo.  outs = 10 device pointers,
 * rows indexed by e = i*ns + j:  GV (E,c) gradient of (x_v[idx] + p_r);  GU (E,c) gradient of x_k[idx] - x_q + p_r;  A (E,c), DZ (E,c/8)
 * input / output-gradient rows of linear_w[2];  G (E,c/8), DL (E,c/8) the same for linear_w[5];  H4, DH4, R4 (E,4) hidden, output
 * gradient of linear_p[0] and relative positions (3 used);  dxq (n,c) = -sum_j GU.  The caller closes the sums over rows with
 * etch_gemm_tn / etch_colsum and scatters GV / GU to the source points with etch_segment_sum_rows (etch_amd/autograd.py). */
int etch_pt_attention_backward(int n, int c, int ns, const float* p, const float* xq, const float* xk, const float* xv, long ldq, const int* idx,
                               const float* const* params, const float* dout, long lddo, float* const* outs, void* stream);
/* dst[q,:] = sum of src[perm[k],:] for k in [seg[q], seg[q+1]) in that order (perm = stable sort of an index list by target row, seg its
 * segment offsets, int64): the reproducible form of scatter-add.  C % 4 == 0. */
int etch_segment_sum_rows(long nseg, int C, const float* src, const long long* perm, const long long* seg, float* dst, void* stream);

/* ---- training-side kernels of the Point-Transformer nets (train.py:77-101 through pointtransformer_seg.py in train() mode; etch_amd/autograd_pt.py) */
/* BatchNorm1d batch statistics of the rows of x (R,C; leading dimension ldx): mean[c], biased var[c], summed in fp64 in a fixed order.
 * workspace: 64 * 2 * C doubles. */
int etch_bn_stats(long R, int C, const float* x, long ldx, double* workspace, float* mean, float* var, void* stream);
/* y (R,C) = act((x - mean[c]) * scale[c] + beta[c]) with scale = gamma / sqrt(var + eps); act = ReLU if relu != 0. */
int etch_bn_apply(long R, int C, const float* x, long ldx, const float* mean, const float* scale, const float* beta, int relu, float* y, void* stream);
/* Backward of y = act(gamma * (x - mean) * rstd + beta): dgamma, dbeta and (dx != NULL) dx.  train != 0: mean / rstd are the batch
 * statistics of x (torch.nn.BatchNorm1d in train() mode); train == 0: constants (running statistics).  y is read only when relu != 0.
 * workspace: 64 * 2 * C doubles. */
int etch_bn_backward(long R, int C, const float* x, long ldx, const float* y, const float* dy, const float* mean, const float* rstd, const float* gamma,
                     int relu, int train, double* workspace, float* dx, float* dgamma, float* dbeta, void* stream);
/* Round 6 -- the same reductions in ONE launch each (the workgroup that finishes last sums the partials in their fixed order: results bit for bit those
 * of the two-launch forms above).  counters: ETCH_REDUCE_COUNTERS (64) unsigned, ZERO before the first call and left zero by every call; calls that
 * share `counters` / `workspace` must be ordered on one stream.
 * etch_bn_train_forward = torch.nn.functional.batch_norm(training=True) (+ ReLU) on rows (pointtransformer_seg.py's nn.BatchNorm1d in train() mode,
 * train.py:77-101): batch statistics -> mean, rstd = 1/sqrt(var + eps), scale = gamma rstd (kept for the backward), y = act((x - mean) scale + beta),
 * and -- when running_mean / running_var are given -- running = (1 - momentum) running + momentum {mean, unbiased var} (momentum < 0: the cumulative
 * average over *num_batches + 1 calls); *num_batches += 1 when given.  2 launches.  workspace: 64 * 2 * C doubles; C <= 4096. */
int etch_bn_train_forward(long R, int C, const float* x, long ldx, const float* gamma, const float* beta, float eps, float momentum,
                          float* running_mean, float* running_var, long long* num_batches, int relu, double* workspace, unsigned* counters,
                          float* mean, float* rstd, float* scale, float* y, void* stream);
/* etch_bn_backward with the column sums and their final reduction in one launch (2 launches with dx). */
int etch_bn_backward_fused(long R, int C, const float* x, long ldx, const float* y, const float* dy, const float* mean, const float* rstd,
                           const float* gamma, int relu, int train, double* workspace, unsigned* counters, float* dx, float* dgamma, float* dbeta,
                           void* stream);
/* Backward of etch_rows_maxpool (nn.MaxPool1d(nsample), pointtransformer_seg.py:66): dy (m*ns,c) = dout at the first maximum of each group. */
int etch_rows_maxpool_backward(long m, int ns, int c, const float* y, const float* dout, float* dy, void* stream);
/* The tail of PointTransformerLayer.forward (pointtransformer_seg.py:34-36): softmax over the ns neighbours of logit (n*ns, cs), then
 * out[i, s*cs + j] = sum_k sm[i,k,j] v[i,k,s*cs + j] for v (n*ns, c); sm (n*ns, cs) is kept for the backward. */
int etch_pt_softmax_agg(long n, int ns, int c, int cs, const float* logit, const float* v, float* sm, float* out, void* stream);
int etch_pt_softmax_agg_backward(long n, int ns, int c, int cs, const float* sm, const float* v, const float* dout, float* dlogit, float* dv, void* stream);

/* Backward of the 8-head dot-product attention over a point's 60 tokens (direction_backbones.py:102-129; autograd through it in
 * train.py:77-101): qkv rows [T*60][ld] with q / k / v at column offsets qoff / koff / voff (the layout of etch_mhsa_attention), dO rows
 * [T*60][ldo] = gradient of the concatenated head outputs -> dqkv rows [T*60][ld] at the same offsets.  Fixed summation order. */
int etch_mhsa_attention_backward(long T, const float* qkv, long ld, int qoff, int koff, int voff, const float* dO, long ldo, float* dqkv,
                                 void* stream);
/* The same for the other encoder depths: embedding_dim in {32, 64, 128, 256} = 8 heads of width 4 / 8 / 16 / 32 (models_pointcloud.py:34-48). */
int etch_mhsa_attention_backward_dim(long T, int embedding_dim, const float* qkv, long ld, int qoff, int koff, int voff, const float* dO, long ldo,
                                     float* dqkv, void* stream);

/* Column sums s[c] = sum_r x[r,c] (bias gradients): fp64, two levels, fixed order.  workspace: 64*C doubles. */
int etch_colsum(long R, int C, const float* x, double* workspace, float* out, void* stream);

/* The same in one coalesced launch (x with leading dimension ldx; counters / ordering as etch_bn_train_forward); C > 4096: the two-launch form
 * (which needs ldx == C). */
int etch_colsum_fused(long R, int C, const float* x, long ldx, double* workspace, unsigned* counters, float* out, void* stream);

/* d/dx of leaky_relu(InstanceNorm2d(x), slope) (so3conv.py:36-44): x, dy (b,rows,C) channels-last, mean / rstd (b,C) of the forward. */
int etch_instnorm_act_backward_workspace_bytes(int b, int C);
int etch_instnorm_act_backward(int b, int rows, int C, const float* x, const float* dy, const float* mean, const float* rstd, float slope,
                               double* workspace, float* dx, void* stream);

/* ---- stage 2: markers + SMPL Levenberg-Marquardt fit ------------------------------------------------------ */

/* torch.max(part_labels, -1) of predict_smpl (src/inference_demo.py:52-53): logits (R,G) -> int64 labels (R). */
int etch_argmax_rows(long R, int G, const float* logits, long long* out, void* stream);

/* pred_inner_points = points - direction * magnitude / scale_magnitude (src/inference_demo.py:58-59): (n,3),(n,3),(n) -> (n,3). */
int etch_inner_points(long n, const float* pts, const float* dir, const float* mag, float scale, float* out, void* stream);

/* get_markers (src/models/fit_SMPL.py:17-62): pts (B,K,3), labels (B,K) i64, conf (B,K,1) -> markers (B,M,3),
 * valid as float (B,M) and/or as bool bytes (B,M) (either may be NULL).  Per (scan,label): top-3 confidences,
 * weights conf^20, weighted centre; empty label -> zeros + invalid. */
int etch_get_markers(int B, int K, int M, const float* pts, const long long* labels, const float* conf, float* markers,
                     float* valid_f, unsigned char* valid_b, void* stream);

/* Per-scan status of the aggregated markers (NaN check on the marker weights at the boundary): status (B) i32,
 * bit 0 = a valid marker of the scan is non-finite -- every one of a label's top-3 confidences underflowed in conf**20, so the
 * weighted centre is 0/0 exactly as fit_SMPL.py:52-57 computes it, and the fit of that scan is NaN; bit 1 = no valid marker. */
int etch_marker_status(int B, int M, const float* markers, const float* valid_f, int* status, void* stream);

/* fit_smpl's two Levenberg-Marquardt stages (src/models/fit_SMPL.py:161-249; Theseus LM + smplx LBS upstream).
 * Body model: nj joints, nb shape coefficients -- (24, 10) = SMPL, the reference's model (fit_SMPL.py:100); (55, 20) = an
 * SMPL-X-sized model (BASELINE configs[4]); other sizes return ETCH_EUNSUPPORTED.  P = 9 (nj - 1) pose-feature rows.
 * consts = 7 device pointers {J0 (nj,3) = J_regressor @ v_template, Jd (nj,3,nb) = J_regressor @ shapedirs, parents (nj) i32, and for
 * the M <= 96 marker vertices: v_template rows (M,3), shapedirs rows (M,3,nb), posedirs columns as (M, nj-1, 28): per joint k >= 1 its 9 x 3 block (pose-feature row e, component a at 3 e + a) zero-padded to 28 floats,
 * lbs_weights rows (M,nj)}.
 * markers (B,M,3), valid (B,M) float mask.  Variable vector x (B, 3 nj + nb + 3) = pose | betas | global_orient | transl
 * (fit_SMPL.py:174,225).  Stage 0: it0 iterations, step0, damping damp0, betas[:2] only (:161-200); stage 1: it1, step1, damp1, all
 * betas (:219-249).  x_stage0 / err_trace (B, it0+it1+2) / phase_ticks (B,8) may be NULL. */
int etch_smpl_lm_fit(int B, int M, int nj, int nb, const void* const* consts, const float* markers, const float* valid, int it0, float step0,
                     float damp0, int it1, float step1, float damp1, float* x_out, float* x_stage0, float* err_trace, long long* phase_ticks,
                     void* stream);
/* The same fit with every scan's linearisation split over G workgroups (latency regime: a handful of scans leave most of the chip idle):
 * workgroup g takes the marker chunks g, g + G, ...; the partial tiles of [J | r]^T [J | r] are exchanged through `workspace` once per
 * linearisation and added in a fixed order, all G copies of a scan's state stay bit-identical and solve redundantly.  Results may differ
 * from G = 1 in the last bits of the fp64 normal matrix (a different, still fixed, summation order).  B * G must not exceed the CU count
 * (the workgroups of a scan wait for each other); workspace: etch_smpl_lm_split_workspace_bytes(B, nj, nb, G) bytes, contents arbitrary. */
int etch_smpl_lm_fit_split(int B, int M, int nj, int nb, const void* const* consts, const float* markers, const float* valid, int it0, float step0,
                           float damp0, int it1, float step1, float damp1, float* x_out, float* x_stage0, float* err_trace, long long* phase_ticks,
                           int G, void* workspace, void* stream);
/* Failure path of the split fit.  A workgroup that waits longer than its spin limit for its partners (they were not co-resident: another kernel, a CU
 * partition) flags the scan and every workgroup of the scan abandons the fit: the scan's x_out / x_stage0 / err_trace rows are NaN -- never a fit of
 * partial sums -- and the flag stays set in the workspace.  etch_smpl_lm_split_failed counts the flagged scans (synchronises the stream).
 * etch_smpl_lm_debug is the test hook: spin limit in polls (0 = default, ~10 s) and a group index that returns at launch (-1 = none). */
int etch_smpl_lm_split_failed(int B, const void* workspace, int* n_failed, void* stream);
int etch_smpl_lm_debug(unsigned spin_limit, int drop_group);
/* LDS bytes one scan's fit holds for its whole duration (one workgroup per scan); ETCH_EUNSUPPORTED for unknown (nj, nb). */
int etch_smpl_lm_workspace_bytes(int nj, int nb);

/* fit_smpl of the first-order variant (src/models/fit_SMPL_Adam.py:68-225): torch.optim.Adam semantics (bias-corrected, betas / eps as
 * given; the reference uses the defaults 0.9 / 0.999 / 1e-8 and lr 1e-2) on the batch-mean squared marker error over the valid marker
 * coordinates; it0 steps on betas[:2] (:104-160), then it1 steps with a fresh optimizer state on all betas (:166-218).  Same consts /
 * markers / valid / x layout as etch_smpl_lm_fit.  x_last (optional) = the parameters of the last forward pass (before the final
 * step): the reference builds its output meshes from that forward.  loss_trace (B, it0+it1, optional): this scan's share of L. */
int etch_smpl_adam_fit(int B, int M, int nj, int nb, const void* const* consts, const float* markers, const float* valid, int it0, int it1,
                       float lr, float beta1, float beta2, float eps, float* x_out, float* x_last, float* loss_trace, void* stream);

/* Diagnostics of the LM kernel (tests): ONE linearisation at a caller-given x (B,DOF) with nb_active betas -> residual
 * (B,3M) = mask * (target - markers(x)) (fit_SMPL.py:127-131), the analytic Jacobian d resid / d x (B,3M,DOF) the fit uses in place
 * of the reference's autograd Jacobian (AutoDiffCostFunction, fit_SMPL.py:176-183, 227-234), and (optional) the normal equations
 * as the fit accumulates them on the fp64 matrix cores: (B, DOF+1, DOF+1) fp64, lower triangle = J^T J, row DOF = -J^T r. */
int etch_smpl_lm_linearize(int B, int M, int nj, int nb_model, int nb_active, const void* const* consts, const float* x, const float* markers,
                           const float* valid, float* resid, float* jac, double* normal, void* stream);

/* batch_rodrigues as the fit evaluates it (in-tree copy src/data_utils/GT_dataloader_mixed.py:29-64, angle = |theta + 1e-8|):
 * theta (n,3) -> R (n,9) fp64 and dR/dtheta_q (n,3,9). */
int etch_rodrigues(int n, const float* theta, double* R, float* dR, void* stream);

/* Final smpl_model(...) (fit_SMPL.py:258-259, smplx.SMPL.forward upstream): x (B, 3 nj + nb + 3) -> verts (B,V,3), joints
 * (B, nj + n_extra, 3) = regressed joints + n_extra vertex-picked joints.  consts = 8 device pointers {v_template (V,3), shapedirs
 * (V,3,nb), posedirs (9 (nj-1), V*3), lbs_weights (V,nj), J0, Jd, parents, extra_vids (n_extra) i32}.  nj <= 55, nb <= 20. */
int etch_smpl_lbs(int B, int V, int nj, int nb, int n_extra, const void* const* consts, const float* x, float* verts, float* joints, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ETCH_HIP_H */
