// Backward kernels of the EPN encoder's native ops for gfx950 (what train.py needs on MI355X next to
// etch_gather_points_backward).  The reference obtains these gradients from torch.autograd through the UN-FUSED forms
// (/root/reference/external/vgtk/vgtk/so3conv/functional.py:224-324 grouping + :61-67 einsum, modules.py:33-39 GEMM, :131-153 intra
// gather; src/models/so3conv.py:24-44,85-99 InstanceNorm2d + leaky_relu; call sites src/train.py:77-101).  Here every gradient is a
// hand-written kernel; kernel weights are regenerated from the coordinates (never stored), and every reduction runs in a FIXED order
// (no atomics): results are reproducible run to run, like the gather backward.
//
//   etch_inter_x1_rows        X1[(b,p,a), c*24+k] = sum_n F[b,idx[p,n],a,c] w[p,a,k,n]         (the conv's grouped features, recomputed)
//   etch_gemm_tn              C[M,N] (+)= A[R,M]^T B[R,N]                                      dW = dY^T X1 on the fp32 matrix cores
//   etch_inter_dfeat          dF[b,q,a,c] (+)= sum_{(p,n): idx[p,n]=q} sum_k w[p,a,k,n] dX1[(b,p,a), c*24+k]   gather-side, index order
//   etch_intra_rows           Xg[(b,p,a), t*C+c] = X[b,p,intra_idx[a,t],c]                     operand of the intra conv's dW
//   etch_colsum               s[c] = sum_r x[r,c]                                              bias gradients
//   etch_instnorm_act_backward  d/dx of leaky_relu(InstanceNorm(x)) given dy, mean, rstd
// Activations are channels-last ([b, p, 60, c]) as in the forward kernels.
#include "common.h"
#include "colstat.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NA 60
#define KS 24

// ---------------------------------------------------------------------------------------------- inter conv: grouped features
// One workgroup per output point.  thread <-> (anchor a, kernel point k) pairs (1440 of them, 6 passes of 240 threads... 256);
// the neighbour terms (2g/sigma, 1-|g|^2/sigma) and the source rows' offsets are staged in LDS once.
__global__ void __launch_bounds__(256) inter_x1_rows_kernel(int cin, int p1, int p2, int p_begin, int nn, float inv_sigma,
                                                            const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                            const int* __restrict__ ball_idx, const float* __restrict__ feats,
                                                            const float* __restrict__ rk, float* __restrict__ x1) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* GA = reinterpret_cast<float4*>(smem);        // [nn]
    int* qi = reinterpret_cast<int*>(smem + 4 * nn);     // [nn]  (-1: padded slot)
    const int tid = threadIdx.x, b = blockIdx.y, pl = blockIdx.x, p = p_begin + pl;
    const float* X = xyz + (size_t)b * 3 * p1;
    for (int n = tid; n < nn; n += 256) {
        const int q = ball_idx[((size_t)b * p2 + p) * nn + n];
        qi[n] = q;
        const float x = X[q] - new_xyz[((size_t)b * 3 + 0) * p2 + p], y = X[p1 + q] - new_xyz[((size_t)b * 3 + 1) * p2 + p],
                    z = X[2 * p1 + q] - new_xyz[((size_t)b * 3 + 2) * p2 + p];
        GA[n] = make_float4(2.0f * inv_sigma * x, 2.0f * inv_sigma * y, 2.0f * inv_sigma * z, 1.0f - (x * x + y * y + z * z) * inv_sigma);
    }
    __syncthreads();
    const float* Fb = feats + (size_t)b * p1 * NA * cin;
    const int kk = cin * KS;
    float* out = x1 + ((size_t)b * gridDim.x + pl) * NA * kk;
    for (int e = tid; e < NA * KS; e += 256) {
        const int a = e / KS, k = e - a * KS;
        const float rx = rk[(a * KS + k) * 3], ry = rk[(a * KS + k) * 3 + 1], rz = rk[(a * KS + k) * 3 + 2];
        const float rb = -(rx * rx + ry * ry + rz * rz) * inv_sigma;
        for (int c0 = 0; c0 < cin; c0 += 16) {
            float acc[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] = 0.f;
            for (int n = 0; n < nn; ++n) {
                const float4 g = GA[n];
                const float w = fmaxf(0.f, fmaf(g.z, rz, fmaf(g.y, ry, fmaf(g.x, rx, g.w + rb))));
                if (w > 0.f) {
                    const float* fr = Fb + ((size_t)qi[n] * NA + a) * cin + c0;
#pragma unroll
                    for (int c = 0; c < 16; ++c)
                        if (c0 + c < cin) acc[c] = fmaf(fr[c], w, acc[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c0 + c < cin) out[(size_t)a * kk + (c0 + c) * KS + k] = acc[c];
        }
    }
}

// ---------------------------------------------------------------------------------------------- C (+)= A^T B on the matrix cores
// grid (N/64, M/64, splits): a workgroup owns a 64x64 tile of C and a contiguous range of the R rows; 32 rows at a time go through
// LDS row-major (coalesced loads), the MFMA fragments are read transposed (row stride = 16 mod 64 floats: conflict-free).
// Partial tiles [split][M][N] are summed in split order by gemm_tn_reduce_kernel (deterministic).
// Round 5: on v_mfma_f64_16x16x4_f64 -- fp32 operands widened exactly, products exact (48 < 53 bits), fp64 accumulation and fp64 partial tiles.  These
// are the weight gradients: sums over every row (token) of the batch, and for the direction head's layers they cancel to ~1e-7 of the sum of the
// terms' magnitudes (profiles/r05_weight_gradient_accumulation.txt): fp32 accumulation, though exact to 2e-8 of that sum, left 12 % error in
// d(net[2].weight).  The fp64 matrix pipe runs at half the fp32 one's rate and these GEMMs are ~2 % of a training step.
// Round 6: (a) tiles of 32 or 64 rows / columns of C (the Point-Transformer nets' layers are 3 .. 64 wide mostly: a 64 x 64 tile spent 3/4 .. 15/16 of
// its fp64 MFMAs on padding), (b) row ranges of ~256 rows per workgroup instead of 2048 (a 32-wide layer over 80 000 rows ran on 40 of 256 CUs, each
// bound by its own fp64 matrix pipe: 354 launches of ~90 us were 32 of a training step's 125 ms of kernel time), (c) the next 32 rows' loads in flight
// while the current ones are multiplied.  The partial tiles are still summed in split order: a result depends on (R, M, N) only.
typedef double f64x4 __attribute__((ext_vector_type(4)));
// FUSED (round 6): the workgroup of a tile that finishes last (common.h: etch_last_block, one counter per tile) sums the tile's partials in split order
// -- gemm_tn_reduce_kernel's sums, bit for bit -- so a weight gradient is one launch instead of two (177 per training step).
// VEC: M, N, lda, ldb multiples of 4 and 16-byte aligned bases -- one 16-byte load per operand piece instead of four predicated 4-byte ones.
template <int TM, int TN, bool FUSED, bool VEC>
__global__ void __launch_bounds__(256) gemm_tn_kernel(long R, int M, int N, const float* __restrict__ A, long lda, const float* __restrict__ B,
                                                      long ldb, double* part, unsigned* counters, float* C, int accumulate) {
    constexpr int LDA = TM + 16, LDB = TN + 16;                    // row stride = 16 mod 32 / 64 floats: the transposed fragment reads are conflict-free
    constexpr int NI = TM / 32, NJ = TN / 32;                      // 16 x 16 blocks per wave (waves 2 x 2 over the tile)
    constexpr int QA = TM / 4, QB = TN / 4, NLA = 32 * QA / 256 > 0 ? 32 * QA / 256 : 1, NLB = 32 * QB / 256 > 0 ? 32 * QB / 256 : 1;
    __shared__ __attribute__((aligned(16))) float As[32 * LDA], Bs[32 * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * TN, m0 = blockIdx.y * TM, sp = blockIdx.z, nsp = gridDim.z;
    const long r_begin = R * sp / nsp, r_end = R * (sp + 1) / nsp;
    const int wm = (wave >> 1) * (TM / 2), wn = (wave & 1) * (TN / 2);
    f64x4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f64x4){0, 0, 0, 0};
    float4 va[NLA], vb[NLB];
    auto fetch = [&](long r0) {
#pragma unroll
        for (int l = 0; l < NLA; ++l) {
            const int e = tid + 256 * l, r = e / QA, c4 = (e % QA) * 4;
            va[l] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < 32 * QA && r0 + r < r_end) {
                const float* ar = A + (r0 + r) * lda + m0 + c4;
                if (VEC) { if (m0 + c4 < M) va[l] = *reinterpret_cast<const float4*>(ar); }
                else va[l] = make_float4(m0 + c4 < M ? ar[0] : 0.f, m0 + c4 + 1 < M ? ar[1] : 0.f, m0 + c4 + 2 < M ? ar[2] : 0.f, m0 + c4 + 3 < M ? ar[3] : 0.f);
            }
        }
#pragma unroll
        for (int l = 0; l < NLB; ++l) {
            const int e = tid + 256 * l, r = e / QB, c4 = (e % QB) * 4;
            vb[l] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < 32 * QB && r0 + r < r_end) {
                const float* br = B + (r0 + r) * ldb + n0 + c4;
                if (VEC) { if (n0 + c4 < N) vb[l] = *reinterpret_cast<const float4*>(br); }
                else vb[l] = make_float4(n0 + c4 < N ? br[0] : 0.f, n0 + c4 + 1 < N ? br[1] : 0.f, n0 + c4 + 2 < N ? br[2] : 0.f, n0 + c4 + 3 < N ? br[3] : 0.f);
            }
        }
    };
    if (r_begin < r_end) fetch(r_begin);
    for (long r0 = r_begin; r0 < r_end; r0 += 32) {
        __syncthreads();
#pragma unroll
        for (int l = 0; l < NLA; ++l) {
            const int e = tid + 256 * l;
            if (e < 32 * QA) *reinterpret_cast<float4*>(&As[(e / QA) * LDA + (e % QA) * 4]) = va[l];
        }
#pragma unroll
        for (int l = 0; l < NLB; ++l) {
            const int e = tid + 256 * l;
            if (e < 32 * QB) *reinterpret_cast<float4*>(&Bs[(e / QB) * LDB + (e % QB) * 4]) = vb[l];
        }
        __syncthreads();
        if (r0 + 32 < r_end) fetch(r0 + 32);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            double a[NI], bq[NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i) a[i] = (double)As[(4 * t + fg) * LDA + wm + 16 * i + fr];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bq[j] = (double)Bs[(4 * t + fg) * LDB + wn + 16 * j + fr];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bq[j], acc[i][j], 0, 0, 0);
        }
    }
    if (FUSED && nsp == 1) {                                       // one row range: the tile is complete -- no partials, no hand-over (the same bits: 0 + acc)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = m0 + wm + 16 * i + 4 * q + fg, n = n0 + wn + 16 * j + fr;
                    if (m < M && n < N) C[(size_t)m * N + n] = (float)((accumulate ? (double)C[(size_t)m * N + n] : 0.0) + acc[i][j][q]);
                }
        return;
    }
    double* P = part + (size_t)sp * M * N;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + wm + 16 * i + 4 * q + fg, n = n0 + wn + 16 * j + fr;       // D layout of the fp64 MFMA: row 4 q + lane / 16 (the fp32 one's: 4 (lane / 16) + q)
                if (m < M && n < N) P[(size_t)m * N + n] = acc[i][j][q];
            }
    if (FUSED) {
        if (!etch_last_block(counters + blockIdx.y * gridDim.x + blockIdx.x, (unsigned)nsp)) return;
        const size_t MN = (size_t)M * N;
        for (int e = tid; e < TM * TN; e += 256) {
            const int m = m0 + e / TN, n = n0 + e % TN;
            if (m >= M || n >= N) continue;
            const size_t i = (size_t)m * N + n;
            double s = accumulate ? (double)C[i] : 0.0;
            for (int k = 0; k < nsp; k += 8) {                      // eight loads in flight (after the acquire they come from memory), added in split order
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = k + q < nsp ? part[(size_t)(k + q) * MN + i] : 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q) if (k + q < nsp) s += v[q];
            }
            C[i] = (float)s;
        }
    }
}

__global__ void __launch_bounds__(256) gemm_tn_reduce_kernel(long MN, int splits, const double* __restrict__ part, float* __restrict__ C, int accumulate) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < MN; i += (long)gridDim.x * 256) {
        double s = accumulate ? (double)C[i] : 0.0;
        for (int k = 0; k < splits; ++k) s += part[(size_t)k * MN + i];
        C[i] = (float)s;
    }
}

// ---------------------------------------------------------------------------------------------- inter conv: d feats
// One workgroup per SOURCE point q (gather side): the chunk's index list is scanned 256 slots at a time in index order; a matching
// slot (p, n) contributes  sum_k w[p,a,k,n] dX1[(p,a), c*24+k]  to dF[q,a,c] with w regenerated from g = xyz_q - new_xyz_p.
// thread <-> (anchor a, channel group): 240 threads = 60 anchors x 4 groups of cin/4 channels.
__global__ void __launch_bounds__(256) inter_dfeat_kernel(int cin, int cbase, int cw, int p1, int p2, int p_begin, int pc, int nn, float inv_sigma,
                                                          const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                          const int* __restrict__ ball_idx, const float* __restrict__ rk,
                                                          const float* __restrict__ dx1, float* __restrict__ dfeats, int accumulate) {
    __shared__ int match[256];
    __shared__ int nmatch_w[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y, q = blockIdx.x;
    const float* X = xyz + (size_t)b * 3 * p1;
    const float qx = X[q], qy = X[p1 + q], qz = X[2 * p1 + q];
    const int a = tid >> 2, cg = tid & 3, cper = cw >> 2, kk = cin * KS;      // this launch: channels cbase .. cbase + cw - 1 (cw <= 64)
    const bool worker = tid < NA * 4;
    float acc[16];                                        // cin / 4 <= 16 channels per thread
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    const int* idx = ball_idx + ((size_t)b * p2 + p_begin) * nn;
    const long nslot = (long)pc * nn;
    for (long s0 = 0; s0 < nslot; s0 += 256) {
        const long sl = s0 + tid;
        const bool hit = sl < nslot && idx[sl] == q;
        const unsigned long long m = __ballot(hit);
        if (lane == 0) nmatch_w[wave] = __popcll(m);
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += nmatch_w[w];
        const int total = nmatch_w[0] + nmatch_w[1] + nmatch_w[2] + nmatch_w[3];
        if (hit) match[base + __popcll(m & ((1ull << lane) - 1ull))] = (int)sl;      // compacted in slot order
        __syncthreads();
        if (worker) {
            for (int e = 0; e < total; ++e) {
                const int slot = match[e];
                const int pl = slot / nn;
                const int p = p_begin + pl;
                const float gx = qx - new_xyz[((size_t)b * 3 + 0) * p2 + p], gy = qy - new_xyz[((size_t)b * 3 + 1) * p2 + p],
                            gz = qz - new_xyz[((size_t)b * 3 + 2) * p2 + p];
                const float ga = 1.0f - (gx * gx + gy * gy + gz * gz) * inv_sigma;
                // Round 6: the 24 weights of this (match, anchor) once into registers, then every channel's 24 gradients as SIX 16-byte loads (a channel's
                // row of dX1 is 24 consecutive floats, 96 bytes, 16-byte aligned: kk and KS are multiples of 4) -- the first form read them one float at a
                // time with a stride of 24 between a thread's channels: 26 % of a training step (profiles/r06_train_kernel_stats.txt).  Same products,
                // same order of the sum over k for every channel (k ascending; zero weights contribute +0 instead of being skipped: bitwise the same sum
                // unless an accumulator is -0).
                float wv[KS];
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    const float rx = rk[(a * KS + k) * 3], ry = rk[(a * KS + k) * 3 + 1], rz = rk[(a * KS + k) * 3 + 2];
                    wv[k] = fmaxf(0.f, fmaf(2.0f * inv_sigma * gz, rz, fmaf(2.0f * inv_sigma * gy, ry, fmaf(2.0f * inv_sigma * gx, rx,
                                                ga - (rx * rx + ry * ry + rz * rz) * inv_sigma))));
                }
                const float4* drow = reinterpret_cast<const float4*>(dx1 + (((size_t)b * pc + pl) * NA + a) * kk + (size_t)(cbase + cg * cper) * KS);
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (c < cper) {
                        float s_ = acc[c];
#pragma unroll
                        for (int k4 = 0; k4 < KS / 4; ++k4) {
                            const float4 d = drow[c * (KS / 4) + k4];
                            s_ = fmaf(wv[4 * k4], d.x, s_); s_ = fmaf(wv[4 * k4 + 1], d.y, s_); s_ = fmaf(wv[4 * k4 + 2], d.z, s_); s_ = fmaf(wv[4 * k4 + 3], d.w, s_);
                        }
                        acc[c] = s_;
                    }
            }
        }
        __syncthreads();
    }
    if (worker) {
        float* dst = dfeats + (((size_t)b * p1 + q) * NA + a) * cin + cbase + cg * cper;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < cper) dst[c] = (accumulate ? dst[c] : 0.f) + acc[c];
    }
}

// Round 6: the same gradient from the TARGET side.  inter_dfeat_kernel above reads the whole dX1 tile of a target point (60 anchors x cin x 24 floats: 184 /
// 368 KB) once per neighbour SLOT that points at its source -- nn times per target point, 12 GB per 1 024-point chunk, 9.5 ms of a training step.  Here a
// workgroup owns a target point, reads its tile once (one anchor row at a time, transposed into LDS as [k][channel]) and writes the contribution of every
// one of its nn slots:  contrib[b, p, n][a][c] = sum_k w[p, a, k, n] dX1[(p, a), c * 24 + k];  the caller then sums the rows of each SOURCE point in the
// order of a stable sort of the slots by source (etch_segment_sum_rows: the reproducible scatter-add the Point-Transformer gathers already use).
// thread <-> (4 channels, neighbour n0 + j * (256 / (cw / 4))): a slot's 4-channel pieces are contiguous in a row of contrib (cw bytes x 4 per 16 lanes).
__global__ void __launch_bounds__(256) inter_dfeat_slots_kernel(int cin, int cbase, int cw, int p1, int p2, int p_begin, int pc, int nn, float inv_sigma,
                                                                const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                                const int* __restrict__ ball_idx, const float* __restrict__ rk,
                                                                const float* __restrict__ dx1, float* __restrict__ contrib) {
    __shared__ float4 gs[64];                                 // per neighbour: 2 g / sigma, 1 - |g|^2 / sigma   (nn <= 64)
    __shared__ __attribute__((aligned(16))) float dXs[KS * 64];          // this anchor's gradients, [k][channel of the window]
    const int tid = threadIdx.x, b = blockIdx.y, pl = blockIdx.x, p = p_begin + pl, kk = cin * KS;
    const float* X = xyz + (size_t)b * 3 * p1;
    if (tid < nn) {
        const int q = ball_idx[((size_t)b * p2 + p) * nn + tid], qq = q < 0 ? 0 : q;
        const float gx = X[qq] - new_xyz[((size_t)b * 3 + 0) * p2 + p], gy = X[p1 + qq] - new_xyz[((size_t)b * 3 + 1) * p2 + p],
                    gz = X[2 * p1 + qq] - new_xyz[((size_t)b * 3 + 2) * p2 + p];
        gs[tid] = make_float4(2.0f * inv_sigma * gx, 2.0f * inv_sigma * gy, 2.0f * inv_sigma * gz, q < 0 ? -1e30f : 1.0f - (gx * gx + gy * gy + gz * gz) * inv_sigma);
    }
    const int cq = cw >> 2, c4 = tid % cq, n0 = tid / cq, nstep = 256 / cq;           // cw in {4 .. 64}, a power of two times 4 is not required: tid / cq >= nn idles
    for (int a = 0; a < NA; ++a) {
        __syncthreads();                                      // the previous anchor's readers of dXs are done (first pass: gs is written)
        const float4* drow = reinterpret_cast<const float4*>(dx1 + (((size_t)b * pc + pl) * NA + a) * kk + (size_t)cbase * KS);
        for (int e = tid; e < cw * (KS / 4); e += 256) {
            const int c = e / (KS / 4), k4 = e - c * (KS / 4);
            const float4 v = drow[e];
            dXs[(4 * k4) * cw + c] = v.x; dXs[(4 * k4 + 1) * cw + c] = v.y; dXs[(4 * k4 + 2) * cw + c] = v.z; dXs[(4 * k4 + 3) * cw + c] = v.w;
        }
        __syncthreads();
        for (int n = n0; n < nn; n += nstep) {
            const float4 g = gs[n];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const float rx = rk[(a * KS + k) * 3], ry = rk[(a * KS + k) * 3 + 1], rz = rk[(a * KS + k) * 3 + 2];
                const float w = fmaxf(0.f, fmaf(g.z, rz, fmaf(g.y, ry, fmaf(g.x, rx, g.w - (rx * rx + ry * ry + rz * rz) * inv_sigma))));
                const float4 d = *reinterpret_cast<const float4*>(&dXs[k * cw + 4 * c4]);
                acc.x = fmaf(w, d.x, acc.x); acc.y = fmaf(w, d.y, acc.y); acc.z = fmaf(w, d.z, acc.z); acc.w = fmaf(w, d.w, acc.w);
            }
            *reinterpret_cast<float4*>(contrib + ((((size_t)b * pc + pl) * nn + n) * NA + a) * cin + cbase + 4 * c4) = acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------- intra conv: gathered rows
__global__ void __launch_bounds__(256) intra_rows_kernel(long total, int C, int nt, const int* __restrict__ intra_idx, const float* __restrict__ x,
                                                         float* __restrict__ xg) {
    // total = points * 60 * nt * C;  xg[((pt*60 + a) * nt + t) * C + c] = x[(pt*60 + idx[a,t]) * C + c]
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long r = i / C;
        const int t = (int)(r % nt); r /= nt;
        const int a = (int)(r % NA);
        const long pt = r / NA;
        xg[i] = x[(pt * NA + intra_idx[a * nt + t]) * C + c];
    }
}

// ---------------------------------------------------------------------------------------------- column sums (bias gradients)
__global__ void __launch_bounds__(256) colsum_partial_kernel(long R, int C, const float* __restrict__ x, double* __restrict__ part) {
    __shared__ double red[256];
    const int tid = threadIdx.x, c = blockIdx.y, nb = gridDim.x;
    const long r_begin = R * blockIdx.x / nb, r_end = R * (blockIdx.x + 1) / nb;
    double s = 0.0;
    for (long r = r_begin + tid; r < r_end; r += 256) s += (double)x[r * C + c];
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) part[(size_t)blockIdx.x * C + c] = red[0];
}
__global__ void colsum_final_kernel(int C, int nparts, const double* __restrict__ part, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int k = 0; k < nparts; ++k) s += part[(size_t)k * C + c];
    out[c] = (float)s;
}

// Round 6: one launch (colstat.h: 16-byte loads, as many row lanes as the width leaves room for, the last workgroup folds the chunks).  The two
// kernels above give one column to a workgroup -- a stride-C walk.
struct ColsumLoad {
    const float* x; long ldx;
    template <int W> __device__ __forceinline__ void load(long r, int c, double (&v)[1][W]) const {
        if (W == 4) { const float4 q = *reinterpret_cast<const float4*>(x + r * ldx + c); v[0][0] = q.x; v[0][1] = q.y; v[0][2] = q.z; v[0][W - 1] = q.w; }
        else v[0][0] = (double)x[r * ldx + c];
    }
};
template <int W>
__global__ void __launch_bounds__(256) colsum_fused_kernel(long R, int C, const float* __restrict__ x, long ldx, double* part, unsigned* counters,
                                                           float* __restrict__ out) {
    colstat_run<1, W>(R, C, part, counters, ColsumLoad{x, ldx}, [&](int c, const double (&s)[1]) { out[c] = (float)s[0]; });
}

// ---------------------------------------------------------------------------------------------- InstanceNorm + LeakyReLU backward
// y = lrelu(xh), xh = (x - mean) * rstd over the rows of one (scan, channel):  g = dy * lrelu'(xh),
// dx = rstd * (g - mean_rows(g) - xh * mean_rows(g * xh)).  Pass 1: per-chunk fp64 partial sums of g and g*xh; pass 2: apply.
#define INB_CHUNKS 64
__global__ void __launch_bounds__(256) instnorm_bwd_partial_kernel(int rows, int C, const float* __restrict__ x, const float* __restrict__ dy,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   float slope, double* __restrict__ part) {
    __shared__ double red[2][256];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int rpp = 256 / C > 0 ? 256 / C : 1;            // rows per pass (C <= 256, C divides 256)
    const int c = tid % C, rsub = tid / C;
    const int r_begin = (int)(((long)rows * chunk) / INB_CHUNKS), r_end = (int)(((long)rows * (chunk + 1)) / INB_CHUNKS);
    const float mu = mean[(size_t)b * C + c], rs = rstd[(size_t)b * C + c];
    double s1 = 0.0, s2 = 0.0;
    if (rsub < rpp)
        for (int r = r_begin + rsub; r < r_end; r += rpp) {
            const size_t o = ((size_t)b * rows + r) * C + c;
            const float xh = (x[o] - mu) * rs;
            const float g = dy[o] * (xh > 0.f ? 1.0f : slope);
            s1 += (double)g; s2 += (double)g * (double)xh;
        }
    red[0][tid] = s1; red[1][tid] = s2;
    __syncthreads();
    if (rsub == 0) {
        for (int k = 1; k < rpp; ++k) { s1 += red[0][k * C + c]; s2 += red[1][k * C + c]; }
        part[(((size_t)b * INB_CHUNKS + chunk) * 2 + 0) * C + c] = s1;
        part[(((size_t)b * INB_CHUNKS + chunk) * 2 + 1) * C + c] = s2;
    }
}
__global__ void __launch_bounds__(256) instnorm_bwd_apply_kernel(long n, int rows, int C, const float* __restrict__ x, const float* __restrict__ dy,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd, float slope,
                                                                 const double* __restrict__ part, float* __restrict__ dx) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long b = i / ((long)rows * C);
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < INB_CHUNKS; ++k) {
            s1 += part[((b * INB_CHUNKS + k) * 2 + 0) * C + c];
            s2 += part[((b * INB_CHUNKS + k) * 2 + 1) * C + c];
        }
        const float mu = mean[b * C + c], rs = rstd[b * C + c];
        const float xh = (x[i] - mu) * rs;
        const float g = dy[i] * (xh > 0.f ? 1.0f : slope);
        dx[i] = rs * (float)((double)g - s1 / rows - (double)xh * (s2 / rows));
    }
}

static inline unsigned grid_for(long total) {
    long blocks = (total + 255) / 256;
    if (blocks > 65535l * 16) blocks = 65535l * 16;
    return (unsigned)blocks;
}

// Row ranges: ~256 rows per workgroup and no more workgroups than ~8 per CU in all; at most 256 per tile.  Up to 64 ranges the last workgroup of a tile sums
// them (one launch); beyond -- the long-and-narrow products of the direction head and the encoder, 300 000 rows into 2 x 2 tiles, which ran one workgroup
// per CU with every 32-row step's load latency exposed -- the reduction is a launch of its own over all of C.
static inline void gemm_tn_shape(long R, int M, int N, int& tm, int& tn, int& gx, int& gy, int& splits) {
    tm = M <= 32 ? 32 : 64; tn = N <= 32 ? 32 : 64;
    gx = (N + tn - 1) / tn; gy = (M + tm - 1) / tm;
    long s = (R + 255) / 256;
    const long cap = 2048 / ((long)gx * gy) > 0 ? 2048 / ((long)gx * gy) : 1;
    if (s > cap) s = cap;
    if (s > 256) s = 256;
    if (s < 1) s = 1;
    splits = (int)s;
}
#define GEMM_TN_FUSED_MAX_SPLITS 64
template <bool FUSED>
static int gemm_tn_launch(long R, int M, int N, const float* A, long lda, const float* B, long ldb, double* part, unsigned* counters, float* C,
                          int accumulate, hipStream_t st) {
    int tm, tn, gx, gy, splits;
    gemm_tn_shape(R, M, N, tm, tn, gx, gy, splits);
    const dim3 grid(gx, gy, splits);
    const bool vec = (M & 3) == 0 && (N & 3) == 0 && (lda & 3) == 0 && (ldb & 3) == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0;
#define TN_GO(TM_, TN_) do { if (vec) hipLaunchKernelGGL((gemm_tn_kernel<TM_, TN_, FUSED, true>), grid, dim3(256), 0, st, R, M, N, A, lda, B, ldb, part, counters, C, accumulate); \
                             else hipLaunchKernelGGL((gemm_tn_kernel<TM_, TN_, FUSED, false>), grid, dim3(256), 0, st, R, M, N, A, lda, B, ldb, part, counters, C, accumulate); } while (0)
    if (tm == 32 && tn == 32) TN_GO(32, 32);
    else if (tm == 32) TN_GO(32, 64);
    else if (tn == 32) TN_GO(64, 32);
    else TN_GO(64, 64);
#undef TN_GO
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return splits;
}

extern "C" {

int etch_inter_x1_rows(int b, int cin, int p1, int p2, int p_begin, int pc, int nn, float sigma, const float* xyz, const float* new_xyz,
                       const int* ball_idx, const float* feats, const float* rk, float* x1, void* stream) {
    if (b <= 0 || pc <= 0) return ETCH_OK;
    if (cin <= 0 || nn <= 0 || p_begin < 0 || p_begin + pc > p2 || b > 65535) return ETCH_EINVAL;
    hipLaunchKernelGGL(inter_x1_rows_kernel, dim3(pc, b), dim3(256), (size_t)5 * nn * sizeof(float), (hipStream_t)stream, cin, p1, p2, p_begin, nn,
                       1.0f / sigma, xyz, new_xyz, ball_idx, feats, rk, x1);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_gemm_tn_workspace_floats(long R, int M, int N) {
    int tm, tn, gx, gy, splits;
    gemm_tn_shape(R, M, N, tm, tn, gx, gy, splits);
    return 2 * splits * M * N + 2;      // fp64 partial tiles (+ 2: room to align the base to 8 bytes)
}

int etch_gemm_tn(long R, int M, int N, const float* A, long lda, const float* B, long ldb, float* C, int accumulate, float* workspace,
                 void* stream) {
    if (M <= 0 || N <= 0) return ETCH_OK;
    if (R < 0 || lda < M || ldb < N) return ETCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    double* part = reinterpret_cast<double*>(((uintptr_t)workspace + 7) & ~(uintptr_t)7);
    const int splits = gemm_tn_launch<false>(R, M, N, A, lda, B, ldb, part, nullptr, nullptr, 0, st);
    if (splits < 0) return splits;
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(grid_for((long)M * N)), dim3(256), 0, st, (long)M * N, splits, part, C, accumulate);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_gemm_tn_fused(long R, int M, int N, const float* A, long lda, const float* B, long ldb, float* C, int accumulate, float* workspace,
                       unsigned* counters, void* stream) {
    if (M <= 0 || N <= 0) return ETCH_OK;
    if (R < 0 || lda < M || ldb < N || !counters) return ETCH_EINVAL;
    int tm, tn, gx, gy, splits;
    gemm_tn_shape(R, M, N, tm, tn, gx, gy, splits);
    if (gx * gy > ETCH_REDUCE_COUNTERS || splits > GEMM_TN_FUSED_MAX_SPLITS) return etch_gemm_tn(R, M, N, A, lda, B, ldb, C, accumulate, workspace, stream);
    double* part = reinterpret_cast<double*>(((uintptr_t)workspace + 7) & ~(uintptr_t)7);
    const int rc = gemm_tn_launch<true>(R, M, N, A, lda, B, ldb, part, counters, C, accumulate, (hipStream_t)stream);
    return rc < 0 ? rc : ETCH_OK;
}

int etch_inter_dfeat(int b, int cin, int p1, int p2, int p_begin, int pc, int nn, float sigma, const float* xyz, const float* new_xyz,
                     const int* ball_idx, const float* rk, const float* dx1, float* dfeats, int accumulate, void* stream) {
    if (b <= 0 || p1 <= 0) return ETCH_OK;
    if (cin <= 0 || (cin & 3) || (cin > 64 && (cin & 63)) || nn <= 0 || p_begin < 0 || p_begin + pc > p2 || b > 65535) return ETCH_EUNSUPPORTED;
    for (int cbase = 0; cbase < cin; cbase += 64) {         // wider inputs (encoder depths 3 / 4: 128 / 256 channels) in windows of 64 channels
        const int cw = cin - cbase < 64 ? cin - cbase : 64;
        hipLaunchKernelGGL(inter_dfeat_kernel, dim3(p1, b), dim3(256), 0, (hipStream_t)stream, cin, cbase, cw, p1, p2, p_begin, pc, nn, 1.0f / sigma, xyz,
                           new_xyz, ball_idx, rk, dx1, dfeats, accumulate);
        ETCH_RETURN_IF_LAUNCH_FAILED();
    }
    return ETCH_OK;
}

int etch_inter_dfeat_slots(int b, int cin, int p1, int p2, int p_begin, int pc, int nn, float sigma, const float* xyz, const float* new_xyz,
                           const int* ball_idx, const float* rk, const float* dx1, float* contrib, void* stream) {
    if (b <= 0 || pc <= 0) return ETCH_OK;
    if (cin <= 0 || (cin & 3) || (cin > 64 && (cin & 63)) || nn <= 0 || nn > 64 || p_begin < 0 || p_begin + pc > p2 || b > 65535) return ETCH_EUNSUPPORTED;
    if (!xyz || !new_xyz || !ball_idx || !rk || !dx1 || !contrib || sigma <= 0.f) return ETCH_EINVAL;
    for (int cbase = 0; cbase < cin; cbase += 64) {
        const int cw = cin - cbase < 64 ? cin - cbase : 64;
        hipLaunchKernelGGL(inter_dfeat_slots_kernel, dim3(pc, b), dim3(256), 0, (hipStream_t)stream, cin, cbase, cw, p1, p2, p_begin, pc, nn, 1.0f / sigma, xyz,
                           new_xyz, ball_idx, rk, dx1, contrib);
        ETCH_RETURN_IF_LAUNCH_FAILED();
    }
    return ETCH_OK;
}

int etch_intra_rows(long points, int C, int nt, const int* intra_idx, const float* x, float* xg, void* stream) {
    const long total = points * NA * nt * C;
    if (total <= 0) return ETCH_OK;
    hipLaunchKernelGGL(intra_rows_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, total, C, nt, intra_idx, x, xg);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_colsum(long R, int C, const float* x, double* workspace, float* out, void* stream);
int etch_colsum_fused(long R, int C, const float* x, long ldx, double* workspace, unsigned* counters, float* out, void* stream) {
    if (C <= 0) return ETCH_OK;
    if (R < 0 || !x || !workspace || !counters || !out || ldx < C) return ETCH_EINVAL;
    if ((C + 63) / 64 > ETCH_REDUCE_COUNTERS) return ldx == C ? etch_colsum(R, C, x, workspace, out, stream) : ETCH_EUNSUPPORTED;
    if (colstat_vec_ok(C, ldx, x))
        hipLaunchKernelGGL(colsum_fused_kernel<4>, dim3(colstat_chunks(R), colstat_groups<4>(C)), dim3(256), 0, (hipStream_t)stream, R, C, x, ldx, workspace, counters, out);
    else
        hipLaunchKernelGGL(colsum_fused_kernel<1>, dim3(colstat_chunks(R), colstat_groups<1>(C)), dim3(256), 0, (hipStream_t)stream, R, C, x, ldx, workspace, counters, out);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_colsum(long R, int C, const float* x, double* workspace, float* out, void* stream) {
    if (C <= 0) return ETCH_OK;
    if (C > 65535) return ETCH_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(64, C), dim3(256), 0, st, R, C, x, workspace);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, st, C, 64, workspace, out);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_instnorm_act_backward(int b, int rows, int C, const float* x, const float* dy, const float* mean, const float* rstd, float slope,
                               double* workspace, float* dx, void* stream) {
    if (b <= 0 || rows <= 0) return ETCH_OK;
    if (C <= 0 || C > 256 || (256 % C) != 0 || b > 65535) return ETCH_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(instnorm_bwd_partial_kernel, dim3(INB_CHUNKS, b), dim3(256), 0, st, rows, C, x, dy, mean, rstd, slope, workspace);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    const long n = (long)b * rows * C;
    hipLaunchKernelGGL(instnorm_bwd_apply_kernel, dim3(grid_for(n)), dim3(256), 0, st, n, rows, C, x, dy, mean, rstd, slope, workspace, dx);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_instnorm_act_backward_workspace_bytes(int b, int C) { return (int)((size_t)INB_CHUNKS * b * 2 * C * sizeof(double)); }

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Backward of DotProdAttention inside MultiHeadAttention (direction_backbones.py:102-129,160-194) for the ETCH head: 60 tokens per
// point, 8 heads x 8 dims, scores scaled by 1/sqrt(8).  Inputs: qkv rows [T*60][ld] (q | k | v at column offsets qoff / koff / voff,
// head h in columns 8h..8h+7 of each) as the forward's fused QKV GEMM produces them, dO rows [T*60][ldo] (gradient of the concatenated
// head outputs).  Output: dqkv rows [T*60][ld] at the same offsets.
//   P = softmax(S), S = Q K^T / sqrt(8);  dP = dO V^T;  D_i = sum_j dP_ij P_ij;  dS = P * (dP - D) / sqrt(8)
//   dQ = dS K;  dK = dS^T Q;  dV = P^T dO
// One workgroup (4 waves) per point, wave = head (two rounds of 4 heads), lane = token.  Pass 1 (lane = query i): softmax statistics,
// D_i and dQ_i; pass 2 (lane = key j): the scores are formed again column-wise, dK_j and dV_j accumulate in the lane -- no atomics and
// no cross-lane reductions, every sum runs over the 60 tokens in index order: reproducible bit for bit.
// ------------------------------------------------------------------------------------------------
#define MB_L 60
// MB_HD = head width (embedding_dim / 8): 8 for the released encoder depth, 4 / 16 / 32 for EPN_layer_num 1 / 3 / 4 (models_pointcloud.py:34-48)
template <int MB_HD>
__global__ void __launch_bounds__(256) mhsa_attention_backward_kernel(const float* __restrict__ qkv, long ld, int qoff, int koff, int voff,
                                                                      const float* __restrict__ dO, long ldo, float* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) float mb_smem[];
    typedef float (*mb_tile)[MB_L][MB_HD];
    mb_tile qs = reinterpret_cast<mb_tile>(mb_smem), ks = qs + 4, vs = ks + 4, gs = vs + 4;      // [4 waves][60 tokens][head width] each
    typedef float (*mb_stat)[64];
    mb_stat st_m = reinterpret_cast<mb_stat>(mb_smem + 16 * MB_L * MB_HD), st_l = st_m + 4, st_d = st_l + 4;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t row0 = (size_t)blockIdx.x * MB_L;
    const float scale = MB_HD == 8 ? 0.35355339059327373f : MB_HD == 4 ? 0.5f : MB_HD == 16 ? 0.25f : 0.17677669529663687f;      // 1 / sqrt(head width)
    for (int round = 0; round < 2; ++round) {
        const int h = round * 4 + w;
        for (int e = lane; e < MB_L * (MB_HD / 4); e += 64) {
            const int r = e / (MB_HD / 4), half = (e % (MB_HD / 4)) * 4;
            const float* base = qkv + (row0 + r) * ld + h * MB_HD + half;
            *reinterpret_cast<float4*>(&qs[w][r][half]) = *reinterpret_cast<const float4*>(base + qoff);
            *reinterpret_cast<float4*>(&ks[w][r][half]) = *reinterpret_cast<const float4*>(base + koff);
            *reinterpret_cast<float4*>(&vs[w][r][half]) = *reinterpret_cast<const float4*>(base + voff);
            *reinterpret_cast<float4*>(&gs[w][r][half]) = *reinterpret_cast<const float4*>(dO + (row0 + r) * ldo + h * MB_HD + half);
        }
        __syncthreads();
        const int i = lane < MB_L ? lane : MB_L - 1;           // lanes 60..63 shadow token 59 and store nothing
        // ---- pass 1: lane = query i
        {
            float q[MB_HD], g[MB_HD];
#pragma unroll
            for (int d = 0; d < MB_HD; ++d) { q[d] = qs[w][i][d]; g[d] = gs[w][i][d]; }
            float m = -INFINITY;
            for (int j = 0; j < MB_L; ++j) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < MB_HD; ++d) s += q[d] * ks[w][j][d];
                m = fmaxf(m, s * scale);
            }
            float l = 0.f, D = 0.f;
            for (int j = 0; j < MB_L; ++j) {
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < MB_HD; ++d) { s += q[d] * ks[w][j][d]; dp += g[d] * vs[w][j][d]; }
                const float e = __expf(s * scale - m);
                l += e;
                D += e * dp;
            }
            const float il = 1.0f / l;
            D *= il;
            float dq[MB_HD];
#pragma unroll
            for (int d = 0; d < MB_HD; ++d) dq[d] = 0.f;
            for (int j = 0; j < MB_L; ++j) {
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < MB_HD; ++d) { s += q[d] * ks[w][j][d]; dp += g[d] * vs[w][j][d]; }
                const float ds = __expf(s * scale - m) * il * (dp - D) * scale;
#pragma unroll
                for (int d = 0; d < MB_HD; ++d) dq[d] += ds * ks[w][j][d];
            }
            st_m[w][lane] = m; st_l[w][lane] = il; st_d[w][lane] = D;
            if (lane < MB_L) {
                float* o = dqkv + (row0 + lane) * ld + qoff + h * MB_HD;
#pragma unroll
                for (int d = 0; d < MB_HD; d += 4) *reinterpret_cast<float4*>(o + d) = make_float4(dq[d], dq[d + 1], dq[d + 2], dq[d + 3]);
            }
        }
        __syncthreads();
        // ---- pass 2: lane = key j
        {
            float k[MB_HD], v[MB_HD], dk[MB_HD], dv[MB_HD];
#pragma unroll
            for (int d = 0; d < MB_HD; ++d) { k[d] = ks[w][i][d]; v[d] = vs[w][i][d]; dk[d] = 0.f; dv[d] = 0.f; }
            for (int t = 0; t < MB_L; ++t) {                   // query t: broadcast reads
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < MB_HD; ++d) { s += qs[w][t][d] * k[d]; dp += gs[w][t][d] * v[d]; }
                const float p = __expf(s * scale - st_m[w][t]) * st_l[w][t];
                const float ds = p * (dp - st_d[w][t]) * scale;
#pragma unroll
                for (int d = 0; d < MB_HD; ++d) { dk[d] += ds * qs[w][t][d]; dv[d] += p * gs[w][t][d]; }
            }
            if (lane < MB_L) {
                float* ok = dqkv + (row0 + lane) * ld + koff + h * MB_HD;
                float* ov = dqkv + (row0 + lane) * ld + voff + h * MB_HD;
#pragma unroll
                for (int d = 0; d < MB_HD; d += 4) {
                    *reinterpret_cast<float4*>(ok + d) = make_float4(dk[d], dk[d + 1], dk[d + 2], dk[d + 3]);
                    *reinterpret_cast<float4*>(ov + d) = make_float4(dv[d], dv[d + 1], dv[d + 2], dv[d + 3]);
                }
            }
        }
        __syncthreads();
    }
}

template <int HD>
static int launch_mhsa_attention_backward(long T, const float* qkv, long ld, int qoff, int koff, int voff, const float* dO, long ldo, float* dqkv, hipStream_t st) {
    const size_t lds = (size_t)(16 * MB_L * HD + 12 * 64) * sizeof(float);
    auto kern = mhsa_attention_backward_kernel<HD>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)T), dim3(256), lds, st, qkv, ld, qoff, koff, voff, dO, ldo, dqkv);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

extern "C" int etch_mhsa_attention_backward_dim(long T, int embedding_dim, const float* qkv, long ld, int qoff, int koff, int voff, const float* dO, long ldo,
                                                float* dqkv, void* stream) {
    if (T <= 0) return ETCH_OK;
    if (!qkv || !dO || !dqkv) return ETCH_EINVAL;
    if ((ld & 3) || (ldo & 3) || (qoff & 3) || (koff & 3) || (voff & 3) || (((uintptr_t)qkv | (uintptr_t)dO | (uintptr_t)dqkv) & 15)) return ETCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    switch (embedding_dim) {
        case 32: return launch_mhsa_attention_backward<4>(T, qkv, ld, qoff, koff, voff, dO, ldo, dqkv, st);
        case 64: return launch_mhsa_attention_backward<8>(T, qkv, ld, qoff, koff, voff, dO, ldo, dqkv, st);
        case 128: return launch_mhsa_attention_backward<16>(T, qkv, ld, qoff, koff, voff, dO, ldo, dqkv, st);
        case 256: return launch_mhsa_attention_backward<32>(T, qkv, ld, qoff, koff, voff, dO, ldo, dqkv, st);
    }
    return ETCH_EUNSUPPORTED;
}

extern "C" int etch_mhsa_attention_backward(long T, const float* qkv, long ld, int qoff, int koff, int voff, const float* dO, long ldo,
                                            float* dqkv, void* stream) {
    return etch_mhsa_attention_backward_dim(T, 64, qkv, ld, qoff, koff, voff, dO, ldo, dqkv, stream);
}

// ------------------------------------------------------------------------------------------------
// Backward of the 3-NN feature propagation (src/models/pointnet2_utils.py:45-74; forward = prop_interp_kernel):
//   out[n, :] = sum_k w[n,k] * feats[idx[n,k], :]      =>      dfeats[s, :] = sum over { (n,k) : idx[n,k] == s } of w[n,k] * dout[n, :]
// Gather-side and in a FIXED order: `perm` is the stable sort of the flattened (n,k) list by coarse row, `seg` its segment offsets
// (int64, as for etch_segment_sum_rows); one workgroup per coarse row adds its entries in that order -- no atomics, reproducible.
// The interpolation weights and indices are functions of the coordinates only: no gradient flows into them (as in the reference, where
// `dists.sort` / the reciprocal weights do carry autograd history to xyz, but xyz is an input without grad).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) weighted_segment_sum_rows_kernel(long nseg, int C, const float* __restrict__ src, const float* __restrict__ wgt,
                                                                        int fan, const long long* __restrict__ perm,
                                                                        const long long* __restrict__ seg, float* __restrict__ dst) {
    const long q = blockIdx.x;
    if (q >= nseg) return;
    const int c4 = C >> 2;
    const long long k0 = seg[q], k1 = seg[q + 1];
    for (int ch = threadIdx.x; ch < c4; ch += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (long long k = k0; k < k1; ++k) {
            const long long e = perm[k];                   // flat index into the (rows, fan) index / weight lists
            const float w = wgt[e];
            const float4 v = reinterpret_cast<const float4*>(src + (size_t)(e / fan) * C)[ch];
            acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
        }
        reinterpret_cast<float4*>(dst + (size_t)q * C)[ch] = acc;
    }
}

// dst (nseg, C)[q] = sum_{k in [seg[q], seg[q+1])} wgt[perm[k]] * src[perm[k] / fan]   (src (rows, C), wgt (rows * fan))
extern "C" int etch_weighted_segment_sum_rows(long nseg, int C, int fan, const float* src, const float* wgt, const long long* perm, const long long* seg,
                                              float* dst, void* stream) {
    if (nseg <= 0) return ETCH_OK;
    if (C <= 0 || (C & 3) || fan <= 0 || !src || !wgt || !perm || !seg || !dst) return ETCH_EINVAL;
    if (nseg > 0x7fffffffL) return ETCH_EUNSUPPORTED;
    hipLaunchKernelGGL(weighted_segment_sum_rows_kernel, dim3((unsigned)nseg), dim3(256), 0, (hipStream_t)stream, nseg, C, src, wgt, fan, perm, seg, dst);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}
