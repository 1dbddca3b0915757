// Fused SO(3)-equivariant point convolutions of the EPN encoder for gfx950.
//
// Activations live channels-last in HBM:  F[b][p][a][c]  (a = 60 icosahedral anchors), so a
// neighbour gather (q, a) is one contiguous row of c floats.
//
//   etch_inter_so3conv   replaces inter_so3conv_grouping_anchor + inter_so3conv_feat_grouping + BasicSO3Conv
//                        (/root/reference/external/vgtk/vgtk/so3conv/functional.py:286-324, :61-67, modules.py:33-39):
//                        the [b,p,60,24,nn] kernel-weight tensor (921 MB / scan at b0c0) is NEVER materialised;
//                        weights are generated in registers as the B operand of the first MFMA contraction.
//   etch_intra_so3conv   replaces intra_so3conv_grouping + BasicSO3Conv (functional.py:331-378, modules.py:150-153):
//                        the 12x gathered tensor is never written; optional InstanceNorm+LeakyReLU applied on load.
//   etch_instnorm_stats / etch_instnorm_act_add
//                        InstanceNorm2d(affine=False, eps=1e-5) + leaky_relu(0.01) (+ residual branch)
//                        (/root/reference/src/models/so3conv.py:36-44,96-99,178-182).
#include "common.h"
#include "split_bf16.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NA 60
#define KS 24

// ------------------------------------------------------------------------------------------------
// inter conv, CIN in {16,32,64} (+ 128 / 256 in passes of 64 input channels): one workgroup (4 waves) per output point.
//   step 1 (per anchor, per wave):  X1[c,k] = sum_n F[idx[n], a, c] * w[a,k,n]      MFMA M=c, N=k(24->32), K=n
//       w[a,k,n] = relu(1 - |g_n - R_a kappa_k|^2 / sigma) generated per lane as the B fragment
//   step 2 (16 anchors at a time):  Y[o,col] = sum_kappa W[o,kappa] * X1[col][kappa] + bias      MFMA M=o, N=col, K=CIN*24
//       X1 goes through LDS ([col][kappa], row stride CIN*24+4), W is read pre-permuted in fragment order.
// Channel order: a lane gathers VEC = CIN/16 CONSECUTIVE channels of a neighbour row with ONE 4/8/16-byte load (a (q,a) row is CIN
// contiguous floats) and feeds one of them to each of the VEC c-tiles, i.e. row r of c-tile mi is channel VEC*r + mi.  The
// contraction index of step 2 follows that order: kappa' = (half h, cc = r*MTH + (mi - h*MTH), k); the host permutes the columns
// of W accordingly before the fragment permutation (ops.inter_weight_frag) -- the result is the same sum in a different order.
// ------------------------------------------------------------------------------------------------
// BX: step 2 on the bf16 matrix cores with fp32 operands split exactly into three bf16 values each (hi / mid / lo mantissa bytes) and the six
// largest cross products accumulated in fp32 -- the same error against fp64 as the fp32 MFMA (profiles/r03_bf16x3_split.txt) at 2.3 x its rate, and
// beside the VALU instead of on it.  W comes pre-split from the host (ops.inter_weight_split); X1 stays fp32 in LDS and is split by the wave that
// consumes it (each X1 element is read by exactly one wave), in the shadow of that wave's bf16 MFMAs.  Step 1 stays on the fp32 MFMA: its
// weights are generated per use, a split per use would cost more VALU work than the matrix cores save.
// waves per SIMD the BX instantiations are compiled for: the 32 -> 32 channel kernel fits four workgroups per CU in LDS (5.30 -> 5.09 ms with 128 registers)
// W-fragment look-ahead of the BX step 2 (output widths >= 64: an even number of o-tile batches per chunk): the next batch's six fragments are
// requested before the current batch is multiplied, through inline-asm loads + counted waits (plain loads are sunk in front of their MFMAs; fully
// unrolled forms cost 100+ registers).  Two register sets alternate (even / odd batch); the chunk loop stays rolled and issues unconditionally
// (the last chunk re-requests its own first batch: a conditional asm load would turn a set into a phi of the asm's output and its old value, and
// the merging copy reads the register before the load lands), the one set still in flight after the loop is drained with the set as operand.
// Step 2 issues no other vector-memory instruction, so the counts are exact; older loads in flight only make a wait stricter.
__device__ __forceinline__ void bx_wload(f32x4& dst, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p)); }
template <int N> __device__ __forceinline__ void bx_wwait6(f32x4 (&v)[2][3]) {
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[0][2]), "+v"(v[1][0]), "+v"(v[1][1]), "+v"(v[1][2]) : "n"(N));
}
// on for the 64-channel-and-wider inputs (6.55 -> 6.22 ms at 64 -> 64: the look-ahead's 48 registers fit beside two workgroups per CU); off at 32 input
// channels (4.90 -> 5.32 ms at 32 -> 64: the third workgroup per CU it costs was hiding more latency than the look-ahead does)
#ifndef INTER_BX_RING
#define INTER_BX_RING(CIN, COUT) ((CIN) >= 64)
#endif
#ifndef INTER_BX_WPE
#define INTER_BX_WPE(CIN, COUT) ((CIN) <= 32 && (COUT) <= 32 ? 4 : 2)
#endif
template <int CIN, int COUT, int MAXT, int PD, bool BX>     // MAXT = ceil(nn / 16) neighbour chunks held in registers (nn <= 16 * MAXT); PD = gather prefetch distance (chunk-steps)
__global__ void __launch_bounds__(256, BX ? INTER_BX_WPE(CIN, COUT) : 2) inter_so3conv_kernel(
    int p1, int p2, int nn, float inv_sigma, const float* __restrict__ xyz, const float* __restrict__ new_xyz,
    const int* __restrict__ ball_idx, const float* __restrict__ feats, const float* __restrict__ rk,
    const float* __restrict__ Wp, const float* __restrict__ bias, float* __restrict__ out, const int* __restrict__ order,
    double* __restrict__ stat_part) {
    // CIN > 64 (encoder depths 3 / 4: 128 / 256 channels): the input channels are processed in NCC passes of CCH = 64 channels -- step 1
    // on that slice of the gathered rows (kernel weights regenerated per pass: 5 VALU ops against 2 * MT1 MFMAs), step 2 accumulating
    // Y over the slice's part of the contraction; registers and LDS stay those of the 64-channel kernel.  NCC = 1: the code below folds
    // to the single-pass form.
    constexpr int CCH = CIN > 64 ? 64 : CIN;
    constexpr int NCC = CIN / CCH;
    constexpr int MT1 = CCH / 16;          // c tiles in step 1 (per pass)
    constexpr int MT2 = COUT / 16;         // o tiles in step 2
    constexpr int KK = CCH * KS;           // contraction length of step 2 (per pass)
    // For CIN = 64 the X1 tile goes through LDS in two channel halves (the second half waits in registers), which
    // halves the LDS footprint and lets two workgroups share a CU (2 waves / SIMD hide the gather latency of step 1).
    constexpr int HALVES = CCH >= 32 ? 2 : 1;
    constexpr int MTH = MT1 / HALVES;      // c tiles per half
    constexpr int KH = KK / HALVES;        // contraction length per half
    constexpr int S = KH + 40;             // LDS row stride (floats): S/4 = 10 (mod 16) keeps the ds_read_b128 B-fragment reads conflict-free
    constexpr int PS = COUT + 4;           // partial-tile row stride
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X1s = smem;                     // [16][S]
    float* part = smem + 16 * S;           // [4 waves][16 cols][PS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.y;
    int p = blockIdx.x;
    if (order) {
        // spatially ordered schedule: `order` lists the output points of a scan along a space-filling curve.  Workgroup ids go
        // round-robin over the 8 XCDs (grid.x is a multiple of 8), so XCD x gets the x-th contiguous eighth of the curve: the
        // workgroups that share an L2 at any moment are spatial neighbours and gather from the same source rows.
        const int per = gridDim.x >> 3;
        const int slot = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (slot >= p2) return;
        p = order[(size_t)b * p2 + slot];
    }
    const int nchunk = (nn + 15) >> 4;

    // neighbour table of this output point in LDS, one entry per neighbour slot n < 16 * MAXT (shared by all lanes and waves: the
    // matrix-core steps read it with broadcast ds_reads instead of holding 5 * 4 * MAXT registers per lane):
    //   nbt[n] = (2 g / sigma, 1 - |g|^2 / sigma)   for  w = relu(1 - |g - r|^2 / sigma) = relu(nbt.w + rb + nbt.xyz . r)  (5 VALU ops per weight)
    //   noff[n] = float offset of the neighbour's anchor-0 feature row;  padded slots: weight forced to 0, row 0
    constexpr int VEC = MT1;               // consecutive channels a lane gathers per neighbour row
    float4* nbt = reinterpret_cast<float4*>(part + 4 * 16 * PS);       // [16 * MAXT]
    unsigned* noff = reinterpret_cast<unsigned*>(nbt + 16 * MAXT);      // [16 * MAXT]
    if (tid < 16 * MAXT) {
        const int n = tid;
        const int* row = ball_idx + ((size_t)b * p2 + p) * nn;
        int q = row[n < nn ? n : nn - 1];                                // branch-free: clamped read, masked below
        q = n < nn ? q : -1;
        const int qq = q < 0 ? 0 : q;
        const float* X = xyz + (size_t)b * 3 * p1;
        const float x = X[qq] - new_xyz[((size_t)b * 3 + 0) * p2 + p], y = X[p1 + qq] - new_xyz[((size_t)b * 3 + 1) * p2 + p],
                    z = X[2 * p1 + qq] - new_xyz[((size_t)b * 3 + 2) * p2 + p];
        nbt[n] = make_float4(2.0f * inv_sigma * x, 2.0f * inv_sigma * y, 2.0f * inv_sigma * z,
                             q < 0 ? -1e30f : 1.0f - (x * x + y * y + z * z) * inv_sigma);
        noff[n] = (unsigned)qq * (unsigned)(NA * CIN);
    }
    __syncthreads();
    const float* Fb = feats + (size_t)b * p1 * NA * CIN;
    float* outp = out + ((size_t)b * p2 + p) * NA * COUT;
    const bool k1ok = fr < 8;
    const int k1 = k1ok ? 16 + fr : 0;

    // Gathered neighbour rows travel through a register ring PD chunk-steps (4 neighbours each) ahead of the matrix cores, across
    // anchors and across the anchor groups (the first chunks of the next group are in flight during step 2); the kernel points of
    // the next anchor one anchor ahead.  A wave's own MFMAs then cover most of the L2 / HBM latency of its gathers.
    typedef float VecT __attribute__((ext_vector_type(VEC)));
    constexpr int NCS = 4 * MAXT;          // chunk-steps of one wave in one anchor group (4 anchors x MAXT chunks)
    constexpr int NRING = PD + 1;
    static_assert(NCS % NRING == 0, "ring slots must line up across anchor groups");
    VecT ring[NRING][4];
    float rkn[6];
    auto issue = [&](int it_, int cs, VecT (&dst)[4]) {    // it_ = anchor group * NCC + channel pass
        const int j = cs / MAXT, t = cs % MAXT;
        int a = (it_ / NCC) * 16 + wave * 4 + j;
        a = a < NA ? a : NA - 1;
        const float* Fa = Fb + (size_t)a * CIN + (it_ % NCC) * CCH;         // wave-uniform base + 32-bit lane offset
#pragma unroll
#ifdef BX_ABL_NOGATHER
        for (int s = 0; s < 4; ++s) dst[s] = *reinterpret_cast<const VecT*>(Fa + (unsigned)(VEC * fr) + 0 * noff[16 * t + 4 * fg + s]);      // timing experiment: every gather hits one row
#else
        for (int s = 0; s < 4; ++s) dst[s] = *reinterpret_cast<const VecT*>(Fa + (noff[16 * t + 4 * fg + s] + (unsigned)(VEC * fr)));
#endif
    };
    auto issue_rk = [&](int a) {
        a = a < NA ? a : NA - 1;
        const float* rka = rk + (size_t)a * KS * 3;
        rkn[0] = rka[fr * 3]; rkn[1] = rka[fr * 3 + 1]; rkn[2] = rka[fr * 3 + 2];
        rkn[3] = rka[k1 * 3]; rkn[4] = rka[k1 * 3 + 1]; rkn[5] = rka[k1 * 3 + 2];
    };
    issue_rk(wave * 4);
#pragma unroll
    for (int c = 0; c < PD; ++c) issue(0, c, ring[c % NRING]);

    // InstanceNorm statistics of the output, fused: this thread's output channel is fixed (256 % COUT == 0), so it keeps the sum
    // and the sum of squares of everything it writes; reduced per workgroup at the end (stat_part [b][p2][2][COUT]).  In fp64: squares
    // formed in fp32 lose (mean/std)^2 * 6e-8 of the variance to rounding when a channel's mean dominates its spread (32 fp64 ops per
    // thread and point against ~10^4 matrix-core cycles)
    double st_s = 0.0, st_q = 0.0;
    f32x4 y[MT2];
#pragma unroll 1
    for (int it = 0; it < 4 * NCC; ++it) {
        const int ag = it / NCC, cc = it % NCC;
        f32x4 keep[4][HALVES > 1 ? MTH : 1][2];        // second channel half of the wave's 4 anchors (HALVES == 2 only)
        // ---------------- step 1: 4 anchors per wave, MAXT chunk-steps each
        f32x4 acc[MT1][2];
        float r0x = 0.f, r0y = 0.f, r0z = 0.f, r1x = 0.f, r1y = 0.f, r1z = 0.f, rb0 = 0.f, rb1 = 0.f;
#pragma unroll
        for (int cs = 0; cs < NCS; ++cs) {
            const int j = cs / MAXT, t = cs % MAXT;
            const int col = wave * 4 + j;
            const int a = ag * 16 + col;
            if (t == 0) {                                   // anchor start
                asm volatile("" ::: "memory");              // keeps the (loop-invariant) neighbour-table reads in LDS instead of 80 hoisted registers
#pragma unroll
                for (int mi = 0; mi < MT1; ++mi) { acc[mi][0] = (f32x4){0, 0, 0, 0}; acc[mi][1] = (f32x4){0, 0, 0, 0}; }
                r0x = rkn[0]; r0y = rkn[1]; r0z = rkn[2]; r1x = rkn[3]; r1y = rkn[4]; r1z = rkn[5];
                rb0 = -(r0x * r0x + r0y * r0y + r0z * r0z) * inv_sigma;
                rb1 = k1ok ? -(r1x * r1x + r1y * r1y + r1z * r1z) * inv_sigma : -1e30f;   // k >= 24: weight 0
                issue_rk(j < 3 ? a + 1 : ((it + 1) / NCC) * 16 + wave * 4);                // next anchor's kernel points
            }
            {
                constexpr int dummy = 0; (void)dummy;
                const int nc = cs + PD;
                if (nc < NCS) issue(it, nc, ring[nc % NRING]);
                else if (it < 4 * NCC - 1) issue(it + 1, nc - NCS, ring[nc % NRING]);     // wave-uniform
            }
            if (a < NA && t < nchunk) {                     // wave-uniform
                VecT (&cur)[4] = ring[cs % NRING];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float4 g = nbt[16 * t + 4 * fg + s];
                    const float w0 = fmaxf(0.f, fmaf(g.z, r0z, fmaf(g.y, r0y, fmaf(g.x, r0x, g.w + rb0))));
                    const float w1 = fmaxf(0.f, fmaf(g.z, r1z, fmaf(g.y, r1y, fmaf(g.x, r1x, g.w + rb1))));
#pragma unroll
                    for (int mi = 0; mi < MT1; ++mi) {
                        const float av = cur[s][mi];
#ifdef BX_ABL_NOS1MFMA
                        asm volatile("" :: "v"(w0), "v"(w1), "v"(av));      // timing experiment: step 1 without its MFMAs
#else
                        acc[mi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, av, acc[mi][0], 0, 0, 0);      // D[k][c]: weights as the row operand
                        acc[mi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, av, acc[mi][1], 0, 0, 0);
#endif
                    }
                }
            }
            if (t == MAXT - 1) {                            // anchor end
                // D[row = kernel point 4fg + q (+ 16)][col r = fr]: column r of c-tile mi = channel VEC*r + mi -> position cc = 16 mi + r of the
                // half; a lane's 4 accumulator registers are 4 CONSECUTIVE kernel points of one channel: one 16-byte LDS store per tile
                // (the transposed product -- channels as rows -- needed 4-byte stores, 2-way bank conflicts: 8 % of the LDS cycles)
                float* xcol = X1s + col * S;
#pragma unroll
                for (int mi = 0; mi < MTH; ++mi) {
                    float* xr = xcol + (mi * 16 + fr) * KS + 4 * fg;
                    *reinterpret_cast<float4*>(xr) = make_float4(acc[mi][0][0], acc[mi][0][1], acc[mi][0][2], acc[mi][0][3]);
                    if (fg < 2) *reinterpret_cast<float4*>(xr + 16) = make_float4(acc[mi][1][0], acc[mi][1][1], acc[mi][1][2], acc[mi][1][3]);
                }
                if (HALVES > 1) {
#pragma unroll
                    for (int mi = 0; mi < MTH; ++mi) { keep[j][mi][0] = acc[MTH + mi][0]; keep[j][mi][1] = acc[MTH + mi][1]; }
                }
            }
        }
        if (cc == 0) {
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) y[mt] = (f32x4){0, 0, 0, 0};
        }
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
            if (h > 0) {
                __syncthreads();                            // every wave finished reading the first half
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float* xcol = X1s + (wave * 4 + j) * S;
#pragma unroll
                    for (int mi = 0; mi < MTH; ++mi) {
                        float* xr = xcol + (mi * 16 + fr) * KS + 4 * fg;
                        *reinterpret_cast<float4*>(xr) = make_float4(keep[j][mi][0][0], keep[j][mi][0][1], keep[j][mi][0][2], keep[j][mi][0][3]);
                        if (fg < 2) *reinterpret_cast<float4*>(xr + 16) = make_float4(keep[j][mi][1][0], keep[j][mi][1][1], keep[j][mi][1][2], keep[j][mi][1][3]);
                    }
                }
            }
            __syncthreads();
            if constexpr (BX) {
                // ---------------- step 2 on the bf16 matrix cores: chunk t = 32 kappas, K split over the 4 waves; per chunk and o tile six
                // v_mfma_f32_16x16x32_bf16 (smallest cross products first), term-major so that consecutive MFMAs are independent
                const bf16x8* Wq = reinterpret_cast<const bf16x8*>(Wp);
                if constexpr (INTER_BX_RING(CIN, COUT) && MT2 >= 4 && (MT2 / 2) % 2 == 0) {
                    constexpr int NM = MT2 / 2;                  // o-tile batches (of two tiles) per chunk: even
                    constexpr int NTW = KH / 32 / 4;             // chunks per wave
                    f32x4 ra[2][2][3];                           // [set][o tile of the batch][plane]
                    const int tg0 = (cc * HALVES + h) * (KH / 32) + wave;
                    auto issue_w = [&](int c, int m, f32x4 (&dst)[2][3]) {
                        const size_t base = ((size_t)(tg0 + 4 * c) * MT2 + 2 * m) * 3 * 64 + lane;
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl) bx_wload(dst[mt][pl], Wq + base + (mt * 3 + pl) * 64);
                    };
                    issue_w(0, 0, ra[0]);
#pragma unroll 1
                    for (int c = 0; c < NTW; ++c) {
                        const float* xr = &X1s[fr * S + (wave + 4 * c) * 32 + fg * 8];
                        bf16x8 bq[3];
                        split3_pack8(*reinterpret_cast<const float4*>(xr), *reinterpret_cast<const float4*>(xr + 4), bq[0], bq[1], bq[2]);
#pragma unroll
                        for (int m = 0; m < NM; ++m) {
                            // next batch: (c, m + 1), or the first of the next chunk (after the last chunk: its own first again)
                            if (m + 1 < NM) issue_w(c, m + 1, ra[(m + 1) & 1]);
                            else issue_w(c + 1 < NTW ? c + 1 : c, 0, ra[0]);
                            bx_wwait6<6>(ra[m & 1]);             // behind this batch's six loads: the six just requested
                            f32x4 (&ac)[2][3] = ra[m & 1];
#define BX_TERM(PA, PB) _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) y[2 * m + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ac[mt][PA]), bq[PB], y[2 * m + mt], 0, 0, 0);
                            BX_TERM(2, 0) BX_TERM(0, 2) BX_TERM(1, 1) BX_TERM(1, 0) BX_TERM(0, 1) BX_TERM(0, 0)
#undef BX_TERM
                        }
                    }
                    bx_wwait6<0>(ra[0]);                         // the redundant last request
                } else
#pragma unroll 1
                for (int t = wave; t < KH / 32; t += 4) {
                    const float* xr = &X1s[fr * S + t * 32 + fg * 8];
                    bf16x8 bq[3];
                    split3_pack8(*reinterpret_cast<const float4*>(xr), *reinterpret_cast<const float4*>(xr + 4), bq[0], bq[1], bq[2]);
#ifdef BX_ABL_SAMEW
                    const int tg = 0;       // timing experiment: every chunk reads the same fragments (L1 hits)
#else
                    const int tg = (cc * HALVES + h) * (KH / 32) + t;
#endif
                    constexpr int MB = MT2 > 2 ? 2 : MT2;       // o tiles at a time: 2 x 3 planes x 4 registers of W fragments
#pragma unroll
                    for (int m0 = 0; m0 < MT2; m0 += MB) {
                        bf16x8 aq[MB][3];
#pragma unroll
                        for (int mt = 0; mt < MB; ++mt)
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl) aq[mt][pl] = Wq[(((size_t)tg * MT2 + m0 + mt) * 3 + pl) * 64 + lane];
#ifdef BX_ABL_NOMFMA
#define BX_TERM(PA, PB) _Pragma("unroll") for (int mt = 0; mt < MB; ++mt) asm volatile("" :: "v"(aq[mt][PA]), "v"(bq[PB]));
#else
#define BX_TERM(PA, PB) _Pragma("unroll") for (int mt = 0; mt < MB; ++mt) y[m0 + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[mt][PA], bq[PB], y[m0 + mt], 0, 0, 0);
#endif
                        BX_TERM(2, 0) BX_TERM(0, 2) BX_TERM(1, 1) BX_TERM(1, 0) BX_TERM(0, 1) BX_TERM(0, 0)
#undef BX_TERM
                    }
                }
            } else
            // ---------------- step 2: K split over the 4 waves (chunk t of 16 kappas -> wave t & 3).  (An explicitly double-buffered form of
            // this loop measured 5 % slower: profiles/r02_inter_conv_experiments.txt.)
            for (int t = wave; t < KH / 16; t += 4) {
                const float4 bv = *reinterpret_cast<const float4*>(&X1s[fr * S + t * 16 + fg * 4]);
                const int tg = (cc * HALVES + h) * (KH / 16) + t;           // chunk index in the kernel's contraction order
                constexpr int MB = MT2 > 8 ? 4 : MT2;       // W fragments in registers at a time (COUT = 256: four batches of 4 tiles, the accumulators alone take 64 registers)
#pragma unroll
                for (int m0 = 0; m0 < MT2; m0 += MB) {
                    float4 av[MB];
#pragma unroll
                    for (int mt = 0; mt < MB; ++mt) av[mt] = *reinterpret_cast<const float4*>(&Wp[(((size_t)tg * MT2 + m0 + mt) * 64 + lane) * 4]);
#pragma unroll
                    for (int mt = 0; mt < MB; ++mt) y[m0 + mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt].x, bv.x, y[m0 + mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MB; ++mt) y[m0 + mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt].y, bv.y, y[m0 + mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MB; ++mt) y[m0 + mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt].z, bv.z, y[m0 + mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MB; ++mt) y[m0 + mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt].w, bv.w, y[m0 + mt], 0, 0, 0);
                }
            }
        }
        if (NCC > 1 && cc < NCC - 1) {
            __syncthreads();                                // every wave finished reading this pass's X1 tile before the next pass rewrites it
            continue;
        }
        // y[mt][q] = Y[o = 16mt + 4fg + q][col = fr]
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
            *reinterpret_cast<float4*>(&part[(wave * 16 + fr) * PS + mt * 16 + fg * 4]) = make_float4(y[mt][0], y[mt][1], y[mt][2], y[mt][3]);
        __syncthreads();
        for (int e = tid; e < 16 * COUT; e += 256) {
            const int col = e / COUT, o = e - col * COUT;
            const int a = ag * 16 + col;
            if (a < NA) {
                float v = part[(0 * 16 + col) * PS + o] + part[(1 * 16 + col) * PS + o];
                v += part[(2 * 16 + col) * PS + o] + part[(3 * 16 + col) * PS + o];
                v += bias[o];
                outp[(size_t)a * COUT + o] = v;
                st_s += (double)v; st_q += (double)v * (double)v;
            }
        }
        // next group's X1s / partial writes are ordered behind the barriers above
    }
    if (stat_part) {
        static_assert(256 % COUT == 0, "a thread must keep one output channel");
        __syncthreads();                                    // the partial tile is free again
        double* dred = reinterpret_cast<double*>(part);      // 512 doubles <= 4 * 16 * PS floats
        dred[tid] = st_s; dred[256 + tid] = st_q;
        __syncthreads();
        if (tid < COUT) {
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int k = 0; k < 256 / COUT; ++k) { a0 += dred[k * COUT + tid]; a1 += dred[256 + k * COUT + tid]; }
            double* sp = stat_part + ((size_t)b * p2 + p) * 2 * COUT;
            sp[tid] = a0; sp[COUT + tid] = a1;
        }
    }
}

// inter conv for tiny CIN (the first layer: CIN = 1): VALU only, one workgroup per output point.
// X1[a][c][k] in LDS, then Y[a][o] = sum_{c,k} W[o, c*24+k] X1[a][c][k] + bias.
__global__ void __launch_bounds__(256) inter_so3conv_small_kernel(
    int cin, int cout, int p1, int p2, int nn, float inv_sigma, const float* __restrict__ xyz,
    const float* __restrict__ new_xyz, const int* __restrict__ ball_idx, const float* __restrict__ feats,
    const float* __restrict__ rk, const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* g = smem;                       // [nn][3]
    int* qi = (int*)(smem + 3 * nn);       // [nn]
    float* X1 = smem + 4 * nn;             // [NA][cin*KS]
    const int tid = threadIdx.x, b = blockIdx.y, p = blockIdx.x;
    const float* X = xyz + (size_t)b * 3 * p1;
    for (int n = tid; n < nn; n += 256) {
        const int q = ball_idx[((size_t)b * p2 + p) * nn + n];
        qi[n] = q;
        g[n * 3 + 0] = X[q] - new_xyz[((size_t)b * 3 + 0) * p2 + p];
        g[n * 3 + 1] = X[p1 + q] - new_xyz[((size_t)b * 3 + 1) * p2 + p];
        g[n * 3 + 2] = X[2 * p1 + q] - new_xyz[((size_t)b * 3 + 2) * p2 + p];
    }
    __syncthreads();
    const float* Fb = feats + (size_t)b * p1 * NA * cin;
    const int kk = cin * KS;
    if (cin == 1) {
        // first EPN layer: one accumulator per (anchor, kernel point); neighbour features staged per anchor row on the fly
        for (int e = tid; e < NA * KS; e += 256) {
            const int a = e / KS, k = e - a * KS;
            const float rx = rk[(a * KS + k) * 3], ry = rk[(a * KS + k) * 3 + 1], rz = rk[(a * KS + k) * 3 + 2];
            float acc = 0.f;
            for (int n = 0; n < nn; ++n) {
                const float dx = g[n * 3] - rx, dy = g[n * 3 + 1] - ry, dz = g[n * 3 + 2] - rz;
                const float w = fmaxf(0.f, 1.0f - (dx * dx + dy * dy + dz * dz) * inv_sigma);
                acc = fmaf(Fb[(size_t)qi[n] * NA + a], w, acc);
            }
            X1[a * KS + k] = acc;
        }
    } else {
        for (int e = tid; e < NA * KS; e += 256) {
            const int a = e / KS, k = e - a * KS;
            const float rx = rk[(a * KS + k) * 3], ry = rk[(a * KS + k) * 3 + 1], rz = rk[(a * KS + k) * 3 + 2];
            for (int c = 0; c < cin; ++c) X1[a * kk + c * KS + k] = 0.f;
            for (int n = 0; n < nn; ++n) {
                const float dx = g[n * 3] - rx, dy = g[n * 3 + 1] - ry, dz = g[n * 3 + 2] - rz;
                const float w = fmaxf(0.f, 1.0f - (dx * dx + dy * dy + dz * dz) * inv_sigma);
                if (w > 0.f) {
                    const float* fr = Fb + ((size_t)qi[n] * NA + a) * cin;
                    for (int c = 0; c < cin; ++c) X1[a * kk + c * KS + k] += fr[c] * w;
                }
            }
        }
    }
    __syncthreads();
    float* outp = out + ((size_t)b * p2 + p) * NA * cout;
    for (int e = tid; e < NA * cout; e += 256) {
        const int a = e / cout, o = e - a * cout;
        float acc = 0.f;
        const float* wr = W + (size_t)o * kk;
        const float* xr = X1 + a * kk;
        for (int i = 0; i < kk; ++i) acc += wr[i] * xr[i];
        outp[e] = acc + bias[o];
    }
}

// First EPN layer (CIN = 1, functional.py:286-324 with a single input channel): X1[a][k] = sum_n F[idx_n, a] * w[a,k,n].
// There is no channel dimension to put on the matrix cores (the weights depend on the anchor), so this is VALU work:
// 92 160 weights per output point at 5 ops each (the relu(ga_n + rb_k + G_n . r_k) form of the MFMA kernel above) + 1 FMA.
// One workgroup per output point; thread = (anchor, group of 6 kernel points): 240 of 256 threads busy in one pass, the
// per-neighbour terms (G_n, ga_n) and the gathered features F[idx_n, :] are staged once in LDS (one broadcast
// ds_read_b128 + one ds_read_b32 per neighbour feed 36 VALU ops); two kernel points per instruction (v_pk_fma_f32).
// UNIT (round 6): feats == NULL stands for the all-ones occupancy features the encoder feeds this conv (so3conv.py:7-16, functional.py:70-89): no gather of
// the neighbours' feature rows (nn x 60 floats per point through LDS), no product with them.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool UNIT>
__global__ void __launch_bounds__(256) inter_so3conv_c1_kernel(
    int cout, int p1, int p2, int nn, float inv_sigma, const float* __restrict__ xyz, const float* __restrict__ new_xyz,
    const int* __restrict__ ball_idx, const float* __restrict__ feats, const float* __restrict__ rk, const float* __restrict__ W,
    const float* __restrict__ bias, float* __restrict__ out, double* __restrict__ stat_part) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* GA = reinterpret_cast<float4*>(smem);          // [nn]  (2/sigma * g, 1 - |g|^2/sigma)
    float* Fn = smem + 4 * nn;                              // [nn][NA]
    float* X1 = Fn + nn * NA;                               // [NA][KS]
    float* Ws = X1 + NA * KS;                               // [cout][KS]
    int* qi = reinterpret_cast<int*>(Ws + cout * KS);       // [nn]
    const int tid = threadIdx.x, b = blockIdx.y, p = blockIdx.x;
    const float* X = xyz + (size_t)b * 3 * p1;
    for (int n = tid; n < nn; n += 256) {
        const int q = ball_idx[((size_t)b * p2 + p) * nn + n];
        qi[n] = q;
        const float x = X[q] - new_xyz[((size_t)b * 3 + 0) * p2 + p], y = X[p1 + q] - new_xyz[((size_t)b * 3 + 1) * p2 + p],
                    z = X[2 * p1 + q] - new_xyz[((size_t)b * 3 + 2) * p2 + p];
        GA[n] = make_float4(2.0f * inv_sigma * x, 2.0f * inv_sigma * y, 2.0f * inv_sigma * z, 1.0f - (x * x + y * y + z * z) * inv_sigma);
    }
    for (int e = tid; e < cout * KS; e += 256) Ws[e] = W[e];
    __syncthreads();
    if (!UNIT) {
        const float* Fb = feats + (size_t)b * p1 * NA;
        for (int e = tid; e < nn * NA; e += 256) {
            const int n = e / NA, a = e - n * NA;
            Fn[e] = Fb[(size_t)qi[n] * NA + a];
        }
        __syncthreads();
    }
    if (tid < NA * 4) {
        const int a = tid >> 2, k0 = (tid & 3) * 6;
        f32x2 rx[3], ry[3], rz[3], rb[3], acc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float* r0 = rk + ((size_t)a * KS + k0 + 2 * j) * 3;
            rx[j] = (f32x2){r0[0], r0[3]}; ry[j] = (f32x2){r0[1], r0[4]}; rz[j] = (f32x2){r0[2], r0[5]};
            rb[j] = -(rx[j] * rx[j] + ry[j] * ry[j] + rz[j] * rz[j]) * inv_sigma;
            acc[j] = (f32x2){0.f, 0.f};
        }
        for (int n = 0; n < nn; ++n) {
            const float4 g = GA[n];
            const float f = UNIT ? 1.0f : Fn[n * NA + a];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                f32x2 w = rz[j] * g.z + (ry[j] * g.y + (rx[j] * g.x + (rb[j] + g.w)));
                w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f);
                acc[j] += w * f;
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) { X1[a * KS + k0 + 2 * j] = acc[j].x; X1[a * KS + k0 + 2 * j + 1] = acc[j].y; }
    }
    __syncthreads();
    float* outp = out + ((size_t)b * p2 + p) * NA * cout;
    // Step 2 (32 output channels, the path's first conv): thread = (anchor, block of 8 channels) keeps its X1 row in registers and reads the weights as
    // 16-byte broadcasts -- 54 LDS instructions per thread instead of 360 scalar ones (two per FMA: the launch was as much LDS- as VALU-bound); the
    // 60 x 32 outputs go through the (now free) gathered-feature tile so that the store / statistics loop below keeps its thread mapping, i.e. the
    // same FMA order per output and the same fp64 summation order: results bit for bit those of the scalar form.
#ifdef C1_SCALAR_STEP2
    const bool tile2 = false;                               // (A/B build: the scalar form)
#else
    const bool tile2 = cout == 32 && nn >= 32;
#endif
    float* Yt = Fn;                                         // [NA][32]
    if (tile2) {
        if (tid < NA * 4) {
            const int a = tid % NA, ob = tid / NA;
            float x[KS];
#pragma unroll
            for (int i = 0; i < KS / 4; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(&X1[a * KS + 4 * i]);
                x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int o = 8 * ob + j;
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < KS / 4; ++i) {
                    const float4 w4 = *reinterpret_cast<const float4*>(&Ws[o * KS + 4 * i]);
                    acc = fmaf(w4.x, x[4 * i], acc); acc = fmaf(w4.y, x[4 * i + 1], acc); acc = fmaf(w4.z, x[4 * i + 2], acc); acc = fmaf(w4.w, x[4 * i + 3], acc);
                }
                Yt[a * 32 + o] = acc + bias[o];
            }
        }
        __syncthreads();
    }
    double st_s = 0.0, st_q = 0.0;                          // fused InstanceNorm statistics in fp64 (needs 256 % cout == 0: fixed channel per thread)
    for (int e = tid; e < NA * cout; e += 256) {
        const int a = e / cout, o = e - a * cout;
        float acc = 0.f;
        if (tile2) {
            acc = Yt[e];
        } else {
#pragma unroll
            for (int i = 0; i < KS; ++i) acc = fmaf(Ws[o * KS + i], X1[a * KS + i], acc);
            acc += bias[o];
        }
        outp[e] = acc;
        st_s += (double)acc; st_q += (double)acc * (double)acc;
    }
    if (stat_part) {
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem + ((4 * nn + 1) & ~1));   // the gathered-feature tile (>= 1024 floats: nn * 60), no longer needed
        red[tid] = st_s; red[256 + tid] = st_q;
        __syncthreads();
        if (tid < cout) {
            double a0 = 0.0, a1 = 0.0;
            for (int k = 0; k < 256 / cout; ++k) { a0 += red[k * cout + tid]; a1 += red[256 + k * cout + tid]; }
            double* sp = stat_part + ((size_t)b * p2 + p) * 2 * cout;
            sp[tid] = a0; sp[cout + tid] = a1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// intra conv: Y[b,p,a,o] = sum_{tap<12} sum_c act(X[b,p,intra_idx[a,tap],c]) * W[o, c*12+tap] + bias[o]
// One workgroup = PTS points; the point's [60][C] tile is staged in LDS (optionally normalised +
// leaky-relu'ed on the way in), each wave owns 16 anchors (N tile) x all output tiles.
// K order inside the kernel: kappa = tap*C + c; W is pre-permuted on the host into fragment order.
// ------------------------------------------------------------------------------------------------
template <int C, int COUT, int PTS>     // C >= 128 (encoder depths 3 / 4): PTS = 1, the input tile in dynamic LDS (71 KB at C = 256), no fused statistics
__global__ void __launch_bounds__(256) intra_so3conv_kernel(int npts_total, int pts_per_batch, const float* __restrict__ X,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const int* __restrict__ intra_idx, const float* __restrict__ Wp,
                                                            const float* __restrict__ bias, float* __restrict__ Y,
                                                            double* __restrict__ stat_part) {
    constexpr int MT = COUT / 16;
    constexpr int LD = C == 16 ? 56 : C + 40;   // row stride (floats) with LD/4 = 10 or 14 (mod 16): conflict-free ds_read_b128 of the gathered rows
    constexpr int NT = 12 * C / 16;        // K chunks
    constexpr bool STATIC_TILE = (size_t)PTS * NA * LD * sizeof(float) <= 60 * 1024;
    __shared__ __attribute__((aligned(16))) float Xs_static[STATIC_TILE ? PTS * NA * LD : 4];
    extern __shared__ __attribute__((aligned(16))) float Xs_dynamic[];
    float* Xs = STATIC_TILE ? Xs_static : Xs_dynamic;
    __shared__ int iidx[NA * 12];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int pt0 = blockIdx.x * PTS;
    for (int e = tid; e < NA * 12; e += 256) iidx[e] = intra_idx[e];
    for (int e = tid; e < PTS * NA * (C / 4); e += 256) {
        const int c4 = e % (C / 4), row = e / (C / 4);          // row = pt*60 + a
        const int pt = pt0 + row / NA;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pt < npts_total) {
            v = *reinterpret_cast<const float4*>(X + ((size_t)pt0 * NA + row) * C + c4 * 4);
            if (mean) {
                const int bb = pt / pts_per_batch;
                const float4 m = *reinterpret_cast<const float4*>(mean + (size_t)bb * C + c4 * 4);
                const float4 r = *reinterpret_cast<const float4*>(rstd + (size_t)bb * C + c4 * 4);
                v.x = (v.x - m.x) * r.x; v.y = (v.y - m.y) * r.y; v.z = (v.z - m.z) * r.z; v.w = (v.w - m.w) * r.w;
                v.x = v.x > 0.f ? v.x : 0.01f * v.x; v.y = v.y > 0.f ? v.y : 0.01f * v.y;
                v.z = v.z > 0.f ? v.z : 0.01f * v.z; v.w = v.w > 0.f ? v.w : 0.01f * v.w;
            }
        }
        *reinterpret_cast<float4*>(&Xs[row * LD + c4 * 4]) = v;
    }
    __syncthreads();
    const int a = wave * 16 + fr;
    const int aa = a < NA ? a : 0;
    f32x4 acc[PTS][MT];
#pragma unroll
    for (int pi = 0; pi < PTS; ++pi)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[pi][mt] = (f32x4){0, 0, 0, 0};
    for (int t = 0; t < NT; ++t) {
        const int tap = t / (C / 16), cb = (t - tap * (C / 16)) * 16;
        const int src = iidx[aa * 12 + tap];
        float4 bv[PTS];
#pragma unroll
        for (int pi = 0; pi < PTS; ++pi) bv[pi] = *reinterpret_cast<const float4*>(&Xs[(pi * NA + src) * LD + cb + fg * 4]);
        float4 av[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const float4*>(&Wp[(((size_t)t * MT + mt) * 64 + lane) * 4]);
        // k-slice outermost so consecutive MFMAs use different accumulators (40-cycle dependent latency)
#define I_STEP(C)                                                                                         \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int pi = 0; pi < PTS; ++pi)  \
        acc[pi][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt].C, bv[pi].C, acc[pi][mt], 0, 0, 0);
        I_STEP(x) I_STEP(y) I_STEP(z) I_STEP(w)
#undef I_STEP
    }
    // InstanceNorm statistics of the output, fused (stat_part != NULL; the host guarantees pts_per_batch % PTS == 0, so a workgroup's
    // points belong to one sample): per lane the sum / sum of squares of its 4 channels per tile over the workgroup's points
    // (in fp64: a channel whose mean dominates its spread loses (mean/std)^2 * 6e-8 of its variance to fp32 squares, and 2 * mean * 6e-8 * mean
    // to an fp32 sum)
    double ss[MT][4], sq[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) ss[mt][q] = sq[mt][q] = 0.0;
    if (a < NA) {
#pragma unroll
        for (int pi = 0; pi < PTS; ++pi) {
            const int pt = pt0 + pi;
            if (pt >= npts_total) continue;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int o = mt * 16 + fg * 4;
                const float4 bs = *reinterpret_cast<const float4*>(bias + o);
                const float4 v = make_float4(acc[pi][mt][0] + bs.x, acc[pi][mt][1] + bs.y, acc[pi][mt][2] + bs.z, acc[pi][mt][3] + bs.w);
                *reinterpret_cast<float4*>(Y + ((size_t)pt * NA + a) * COUT + o) = v;
                ss[mt][0] += (double)v.x; ss[mt][1] += (double)v.y; ss[mt][2] += (double)v.z; ss[mt][3] += (double)v.w;
                sq[mt][0] += (double)v.x * v.x; sq[mt][1] += (double)v.y * v.y; sq[mt][2] += (double)v.z * v.z; sq[mt][3] += (double)v.w * v.w;
            }
        }
    }
    constexpr int RS = 2 * COUT + 4;                        // per anchor slot: COUT doubles (+ pad)
    if constexpr (NA * RS <= PTS * NA * LD) if (stat_part) {
        // [60 anchor slots][COUT fp64] through the (now free) input tile, once for the sums and once for the sums of squares; one thread per
        // channel adds the 60 slots in order (the host only asks for it where the staging fits the input tile)
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            __syncthreads();
            if (a < NA) {
                double* red = reinterpret_cast<double*>(Xs + a * RS);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) red[mt * 16 + fg * 4 + q] = which == 0 ? ss[mt][q] : sq[mt][q];
            }
            __syncthreads();
            if (tid < COUT) {
                double t = 0.0;
                for (int k = 0; k < NA; ++k) t += reinterpret_cast<const double*>(Xs + k * RS)[tid];
                stat_part[(size_t)blockIdx.x * 2 * COUT + which * COUT + tid] = t;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same intra conv on v_mfma_f32_32x32x2_f32.  On this chip the 16x16x4 fp32 MFMA never issues faster than 0.85 of the matrix peak
// (~37 cycles instead of 32), the 32x32x2 form runs at 0.986 of it (profiles/r03_mfma_issue_rate.txt).  Tile = 32 output channels x 32 anchors,
// K = 2 per instruction: wave w owns anchor half h = w & 1 (anchors 32 h .. 32 h + 31; 60..63 are padding) and
//   COUT = 64: output-channel tile mt = w >> 1 for BOTH points of the workgroup (each W fragment feeds two MFMAs),
//   COUT = 32: the single channel tile for point w >> 1.
// K order: step t covers kappa = 8 t .. 8 t + 7 of the tap-major contraction (tap = 8 t / C); lane half kk = lane >> 5 takes kappa = 8 t + 4 kk + s
// in MFMA s (one 16-byte LDS read and one 16-byte weight load feed four MFMAs).  Wp32[t][mt][lane][s] = W2[32 mt + lane % 32][8 t + 4 (lane / 32) + s]
// (ops.permute_weight_frag32).  D[i][j]: lane holds column j = lane % 32 (anchor), rows i = 8 (v / 4) + 4 (lane / 32) + v % 4 (channel) in v = 0..15.
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

// The weight-fragment ring of the kernel below is built from inline-asm loads and waits: written as plain loads the compiler splits every
// 16-byte load into dwords and sinks them in front of their MFMAs (global_load_dword + s_waitcnt vmcnt(0) per MFMA pair: 7.2 -> 8.9 ms), as volatile
// loads it waits for vmcnt(0) at every step (7.5 ms).  Rules that keep this safe: (1) no load is issued whose result is not consumed inside the loop
// (a load still in flight after the loop would land in registers the compiler has already reused); (2) the wait names the ring register as an in/out
// operand, so the consuming MFMAs cannot be scheduled above it; (3) vmcnt counts the compiler's own loads too, and loads return in order, so the
// compiler's waits can only become stricter, never too weak.
__device__ __forceinline__ f32x4 intra32_wload(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void intra32_wwait(f32x4& v) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N)); }

template <int C, int COUT>
__global__ void __launch_bounds__(256) intra_so3conv32_kernel(int npts_total, int pts_per_batch, const float* __restrict__ X,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const int* __restrict__ intra_idx, const float* __restrict__ Wp,
                                                              const float* __restrict__ bias, float* __restrict__ Y,
                                                              double* __restrict__ stat_part) {
    constexpr int PTS = 2;
    constexpr int MT = COUT / 32;
    constexpr int WPP = MT == 1 ? 1 : 2;   // points per wave
    constexpr int LD = C + 40;             // row stride (floats): LD / 4 = 10 (mod 16) as in the 16-wide kernel
    constexpr int NT = 12 * C / 8;         // K steps of 8
    static_assert(MT == 1 || MT == 2, "32 or 64 output channels");
    __shared__ __attribute__((aligned(16))) float Xs[PTS * NA * LD];
    __shared__ int iidx[NA * 12];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, kk = lane >> 5;
    const int pt0 = blockIdx.x * PTS;
    for (int e = tid; e < NA * 12; e += 256) iidx[e] = intra_idx[e];
    for (int e = tid; e < PTS * NA * (C / 4); e += 256) {
        const int c4 = e % (C / 4), row = e / (C / 4);          // row = pt*60 + a
        const int pt = pt0 + row / NA;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pt < npts_total) {
            v = *reinterpret_cast<const float4*>(X + ((size_t)pt0 * NA + row) * C + c4 * 4);
            if (mean) {
                const int bb = pt / pts_per_batch;
                const float4 m = *reinterpret_cast<const float4*>(mean + (size_t)bb * C + c4 * 4);
                const float4 r = *reinterpret_cast<const float4*>(rstd + (size_t)bb * C + c4 * 4);
                v.x = (v.x - m.x) * r.x; v.y = (v.y - m.y) * r.y; v.z = (v.z - m.z) * r.z; v.w = (v.w - m.w) * r.w;
                v.x = v.x > 0.f ? v.x : 0.01f * v.x; v.y = v.y > 0.f ? v.y : 0.01f * v.y;
                v.z = v.z > 0.f ? v.z : 0.01f * v.z; v.w = v.w > 0.f ? v.w : 0.01f * v.w;
            }
        }
        *reinterpret_cast<float4*>(&Xs[row * LD + c4 * 4]) = v;
    }
    __syncthreads();
    const int h = wave & 1;
    const int mt = MT == 1 ? 0 : (wave >> 1);
    const int p_first = MT == 1 ? (wave >> 1) : 0;
    const int a = 32 * h + j;
    const int aa = a < NA ? a : 0;
    f32x16 acc[WPP];
#pragma unroll
    for (int pi = 0; pi < WPP; ++pi)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[pi][v] = 0.f;
    // weight fragments travel through a register ring PF steps ahead of the matrix cores (L2 latency ~ 1 us against 0.25 us of MFMAs per step)
    constexpr int PF = 4;
    static_assert(NT % PF == 0, "ring slots must line up");
    f32x4 avq[PF];
#pragma unroll
    for (int q = 0; q < PF - 1; ++q) avq[q] = intra32_wload(&Wp[(((size_t)q * MT + mt) * 64 + lane) * 4]);
    constexpr int SPT = C / 8;             // K steps per tap
    static_assert(SPT % PF == 0, "a tap's steps must fill whole ring turns");
    // (the last tap is a separate instantiation: a load under a run-time condition would make the ring register a phi of the asm's output and
    // its old value, and the copy that merges them reads the register before the load has landed)
    auto do_tap = [&](int tap, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int src = iidx[aa * 12 + tap];                    // one dependent LDS read per tap, not per step
        const float* xrow = &Xs[src * LD + 4 * kk];
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int q = i % PF;
            const int t = tap * SPT + i;
            constexpr int dummy = 0; (void)dummy;
            const bool tail = LAST && i + PF - 1 >= SPT;               // compile-time per unrolled step: the last PF - 1 steps issue no load
            if (!tail) avq[(q + PF - 1) % PF] = intra32_wload(&Wp[(((size_t)(t + PF - 1) * MT + mt) * 64 + lane) * 4]);
            float4 bv[WPP];
#pragma unroll
            for (int pi = 0; pi < WPP; ++pi) bv[pi] = *reinterpret_cast<const float4*>(xrow + (p_first + pi) * NA * LD + 8 * i);
            // outstanding ring loads here: PF (this step's, the oldest, + PF - 1 younger ones), fewer in the tail
            if (!tail) intra32_wwait<PF - 1>(avq[q]);
            else if (SPT - 1 - i == 2) intra32_wwait<2>(avq[q]);
            else if (SPT - 1 - i == 1) intra32_wwait<1>(avq[q]);
            else intra32_wwait<0>(avq[q]);
            const f32x4 av = avq[q];
#define I32_STEP(CMP) _Pragma("unroll") for (int pi = 0; pi < WPP; ++pi) acc[pi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.CMP, bv[pi].CMP, acc[pi], 0, 0, 0);
            I32_STEP(x) I32_STEP(y) I32_STEP(z) I32_STEP(w)
#undef I32_STEP
        }
    };
#pragma unroll 1
    for (int tap = 0; tap < 11; ++tap) do_tap(tap, std::false_type{});
    do_tap(11, std::true_type{});
    // epilogue: bias, stores (16 bytes = 4 consecutive channels per register group), statistics
    double ss[16], sq[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) ss[v] = sq[v] = 0.0;
    if (a < NA) {
#pragma unroll
        for (int pi = 0; pi < WPP; ++pi) {
            const int pt = pt0 + p_first + pi;
            if (pt >= npts_total) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int o = 32 * mt + 8 * g + 4 * kk;
                const float4 bs = *reinterpret_cast<const float4*>(bias + o);
                const float4 v = make_float4(acc[pi][4 * g] + bs.x, acc[pi][4 * g + 1] + bs.y, acc[pi][4 * g + 2] + bs.z, acc[pi][4 * g + 3] + bs.w);
                *reinterpret_cast<float4*>(Y + ((size_t)pt * NA + a) * COUT + o) = v;
                ss[4 * g] += (double)v.x; ss[4 * g + 1] += (double)v.y; ss[4 * g + 2] += (double)v.z; ss[4 * g + 3] += (double)v.w;
                sq[4 * g] += (double)v.x * v.x; sq[4 * g + 1] += (double)v.y * v.y; sq[4 * g + 2] += (double)v.z * v.z; sq[4 * g + 3] += (double)v.w * v.w;
            }
        }
    }
    if (stat_part) {
        // [slot][COUT fp64] through the (now free) input tile, once for the sums and once for the sums of squares; slot = anchor (COUT = 64: a lane
        // already holds both points) or (point, anchor) (COUT = 32); one thread per channel adds the slots in order
        constexpr int NSLOT = MT == 1 ? 2 * NA : NA;
        constexpr int RS = 2 * COUT + 4;
        static_assert(NSLOT * RS <= PTS * NA * LD, "statistics staging must fit the input tile");
        const int slot = (MT == 1 ? p_first * NA : 0) + a;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            __syncthreads();
            if (a < NA) {
                double* red = reinterpret_cast<double*>(Xs + slot * RS);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int q = 0; q < 4; ++q) red[32 * mt + 8 * g + 4 * kk + q] = which == 0 ? ss[4 * g + q] : sq[4 * g + q];
            }
            __syncthreads();
            if (tid < COUT) {
                double t = 0.0;
                for (int k = 0; k < NSLOT; ++k) t += reinterpret_cast<const double*>(Xs + k * RS)[tid];
                stat_part[(size_t)blockIdx.x * 2 * COUT + which * COUT + tid] = t;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// InstanceNorm statistics over (p, a) per (b, c): deterministic two-level reduction in fp64.
//   x [b][rows][C]  ->  mean[b][C], rstd[b][C] = 1/sqrt(var_biased + eps)
// ------------------------------------------------------------------------------------------------
#define IN_CHUNKS 64
__global__ void __launch_bounds__(256) instnorm_partial_kernel(int rows, int C, const float* __restrict__ x,
                                                               double* __restrict__ partial) {
    // grid (IN_CHUNKS, b); thread -> channel quad (tid % (C/4)) and row slice; 16-byte loads, two rows in flight per thread
    extern __shared__ __attribute__((aligned(16))) double sred[];   // [2][256][4]
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int tpr = C >> 2;                    // threads per row (C % 4 == 0, C/4 divides 256)
    const int rpp = 256 / tpr;                 // rows per pass
    const int cq = tid % tpr, rsub = tid / tpr;
    const int r_begin = (int)(((long)rows * chunk) / IN_CHUNKS), r_end = (int)(((long)rows * (chunk + 1)) / IN_CHUNKS);
    double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
    const float4* xb = reinterpret_cast<const float4*>(x + (size_t)b * rows * C) + cq;
    int r = r_begin + rsub;
    for (; r + rpp < r_end; r += 2 * rpp) {
        const float4 u = xb[(size_t)r * tpr], v = xb[(size_t)(r + rpp) * tpr];
        s[0] += (double)u.x; ss[0] += (double)u.x * u.x; s[1] += (double)u.y; ss[1] += (double)u.y * u.y;
        s[2] += (double)u.z; ss[2] += (double)u.z * u.z; s[3] += (double)u.w; ss[3] += (double)u.w * u.w;
        s[0] += (double)v.x; ss[0] += (double)v.x * v.x; s[1] += (double)v.y; ss[1] += (double)v.y * v.y;
        s[2] += (double)v.z; ss[2] += (double)v.z * v.z; s[3] += (double)v.w; ss[3] += (double)v.w * v.w;
    }
    if (r < r_end) {
        const float4 u = xb[(size_t)r * tpr];
        s[0] += (double)u.x; ss[0] += (double)u.x * u.x; s[1] += (double)u.y; ss[1] += (double)u.y * u.y;
        s[2] += (double)u.z; ss[2] += (double)u.z * u.z; s[3] += (double)u.w; ss[3] += (double)u.w * u.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { sred[tid * 4 + j] = s[j]; sred[1024 + tid * 4 + j] = ss[j]; }
    __syncthreads();
    if (rsub == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double a = s[j], q = ss[j];
            for (int k = 1; k < rpp; ++k) { a += sred[(k * tpr + cq) * 4 + j]; q += sred[1024 + (k * tpr + cq) * 4 + j]; }
            partial[(((size_t)b * IN_CHUNKS + chunk) * 2 + 0) * C + cq * 4 + j] = a;
            partial[(((size_t)b * IN_CHUNKS + chunk) * 2 + 1) * C + cq * 4 + j] = q;
        }
    }
}

__global__ void instnorm_final_kernel(int rows, int C, float eps, const double* __restrict__ partial,
                                      float* __restrict__ mean, float* __restrict__ rstd) {
    const int b = blockIdx.x, c = threadIdx.x;
    if (c >= C) return;
    double s = 0.0, ss = 0.0;
    for (int k = 0; k < IN_CHUNKS; ++k) {
        s += partial[(((size_t)b * IN_CHUNKS + k) * 2 + 0) * C + c];
        ss += partial[(((size_t)b * IN_CHUNKS + k) * 2 + 1) * C + c];
    }
    const double m = s / rows;
    double var = ss / rows - m * m;
    if (var < 0.0) var = 0.0;
    mean[(size_t)b * C + c] = (float)m;
    rstd[(size_t)b * C + c] = (float)(1.0 / sqrt(var + (double)eps));
}

// mean / rstd from per-workgroup partial sums written by the fused inter conv: part [b][nparts][2][C] (sum, sum of squares over
// `count` values each) -> the same statistics as instnorm_partial/final over nparts * count values; fp64 accumulation in a fixed order
__global__ void __launch_bounds__(1024) instnorm_from_partials_kernel(int nparts, int C, int count, float eps, const double* __restrict__ part,
                                                                      float* __restrict__ mean, float* __restrict__ rstd) {
    __shared__ double red[2][1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int c4 = C >> 2;                                 // thread -> channel quad and slice of the parts; 16-byte loads
    const int cq = tid % c4, sl = tid / c4, nsl = 1024 / c4;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    const double* pb = part + (size_t)b * nparts * 2 * C;
    for (int k = sl; k < nparts; k += nsl) {
        const double2* u = reinterpret_cast<const double2*>(pb + (size_t)k * 2 * C + cq * 4);
        const double2* v = reinterpret_cast<const double2*>(pb + (size_t)k * 2 * C + C + cq * 4);
        const double2 u0 = u[0], u1 = u[1], v0 = v[0], v1 = v[1];
        s[0] += u0.x; s[1] += u0.y; s[2] += u1.x; s[3] += u1.y;
        q[0] += v0.x; q[1] += v0.y; q[2] += v1.x; q[3] += v1.y;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __syncthreads();
        red[0][tid] = s[j]; red[1][tid] = q[j];
        __syncthreads();
        if (sl == 0) {
            double a = s[j], d = q[j];
            for (int k = 1; k < nsl; ++k) { a += red[0][k * c4 + cq]; d += red[1][k * c4 + cq]; }
            const double n = (double)nparts * count, m = a / n;
            double var = d / n - m * m;
            if (var < 0.0) var = 0.0;
            mean[(size_t)b * C + cq * 4 + j] = (float)m;
            rstd[(size_t)b * C + cq * 4 + j] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
}

// out = lrelu((x1 - m1) * r1) [+ lrelu((x2 - m2) * r2)]     (elementwise, channels-last, float4)
// Round 6: blockIdx.y = sample, 32-bit indices inside a sample, channel / row from shifts (C is a power of two on this path): the first form divided a
// 64-bit element index three times per float4 (~150 VALU instructions around four loads) and ran at 5.2 of 8 TB/s; two float4 per thread and iteration.
// K1 (round 6): the second branch is a ONE-channel 1x1 conv folded in -- x2 holds one value f per row, (m2, r2) hold per (sample, channel) the slope and
// offset of  (w_c f + bias_c - mean_c) rstd_c  (the skip branch of the encoder's first block: its [rows][C] conv output and that tensor's statistics pass
// are never made: 0.31 + 0.18 ms and 0.6 GB less to read here; etch_instnorm_act_add_k1_planes_f16).
template <bool F16, bool POW2, bool K1 = false>
__global__ void __launch_bounds__(256) instnorm_act_add_kernel(unsigned per4, int rows, int C, int cshift, const float* __restrict__ x1,
                                                               const float* __restrict__ m1, const float* __restrict__ r1,
                                                               const float* __restrict__ x2, const float* __restrict__ m2,
                                                               const float* __restrict__ r2, float* __restrict__ out,
                                                               unsigned short* __restrict__ planes) {
    const int b = blockIdx.y;
    const size_t base4 = (size_t)b * per4;
    const float4* X1 = reinterpret_cast<const float4*>(x1) + base4;
    const float4* X2 = x2 && !K1 ? reinterpret_cast<const float4*>(x2) + base4 : nullptr;
    const float* F2 = K1 ? x2 + (size_t)b * rows : nullptr;
    float4* O = reinterpret_cast<float4*>(out) + base4;
    unsigned short* P = planes ? planes + base4 * 4 * (F16 ? 2 : 3) : nullptr;
    const float* M1 = m1 + (size_t)b * C;
    const float* R1 = r1 + (size_t)b * C;
    const float* M2 = m2 ? m2 + (size_t)b * C : nullptr;
    const float* R2 = r2 ? r2 + (size_t)b * C : nullptr;
    auto one = [&](unsigned i, const float4 v, const float4 v2) {
        const unsigned e = i * 4u;
        const unsigned c = POW2 ? (e & (unsigned)(C - 1)) : (e % (unsigned)C);
        const unsigned row = POW2 ? (e >> cshift) : (e / (unsigned)C);
        const float4 m = *reinterpret_cast<const float4*>(M1 + c);
        const float4 r = *reinterpret_cast<const float4*>(R1 + c);
        float o[4] = {(v.x - m.x) * r.x, (v.y - m.y) * r.y, (v.z - m.z) * r.z, (v.w - m.w) * r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = o[k] > 0.f ? o[k] : 0.01f * o[k];
        if (K1) {
            const float f = F2[row];
            const float4 aa = *reinterpret_cast<const float4*>(M2 + c);
            const float4 bb = *reinterpret_cast<const float4*>(R2 + c);
            const float o2[4] = {fmaf(f, aa.x, bb.x), fmaf(f, aa.y, bb.y), fmaf(f, aa.z, bb.z), fmaf(f, aa.w, bb.w)};
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] += o2[k] > 0.f ? o2[k] : 0.01f * o2[k];
        } else if (X2) {
            const float4 mm = *reinterpret_cast<const float4*>(M2 + c);
            const float4 rr = *reinterpret_cast<const float4*>(R2 + c);
            float o2[4] = {(v2.x - mm.x) * rr.x, (v2.y - mm.y) * rr.y, (v2.z - mm.z) * rr.z, (v2.w - mm.w) * rr.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] += o2[k] > 0.f ? o2[k] : 0.01f * o2[k];
        }
        O[i] = make_float4(o[0], o[1], o[2], o[3]);
        if (P) {
            // the same values split for the next conv's gathers, written by the producer once instead of being split by every gather of the row
            if constexpr (F16) {          // two fp16 planes [row][2][C] (split_bf16.h: split2h): the operand format of etch_inter_so3conv_planes_kq
                uint2 h, l;
                split2h_pack4(make_float4(o[0], o[1], o[2], o[3]), h, l);
                unsigned short* pr = P + (size_t)row * 2 * C + c;
                *reinterpret_cast<uint2*>(pr) = h; *reinterpret_cast<uint2*>(pr + C) = l;
            } else {                      // three bf16 planes [row][3][C] (exact split)
                uint2 hi, mid, lo;
                split3_pack4(make_float4(o[0], o[1], o[2], o[3]), hi, mid, lo);
                unsigned short* pr = P + (size_t)row * 3 * C + c;
                *reinterpret_cast<uint2*>(pr) = hi; *reinterpret_cast<uint2*>(pr + C) = mid; *reinterpret_cast<uint2*>(pr + 2 * C) = lo;
            }
        }
    };
    const unsigned stride = gridDim.x * 256u;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned i = blockIdx.x * 256u + threadIdx.x;
    for (; i + stride < per4; i += 2 * stride) {          // two elements per iteration: all four loads in flight before the first use
        const float4 va = X1[i], vb = X1[i + stride];
        const float4 wa = X2 ? X2[i] : z4, wb = X2 ? X2[i + stride] : z4;
        one(i, va, wa);
        one(i + stride, vb, wb);
    }
    if (i < per4) one(i, X1[i], X2 ? X2[i] : z4);
}

// ------------------------------------------------------------------------------------------------ C ABI
// gather prefetch distance in chunk-steps (PD + 1 must divide 4 * MAXT: 1 or 3; 3 measured 3-8 % slower: more registers, fewer waves)
#ifndef INTER_PD
#define INTER_PD(CIN, MAXT) 1
#endif
template <int CIN, int COUT, int MAXT, bool BX>
static int launch_inter_t(int b, int p1, int p2, int nn, float sigma, const float* xyz, const float* new_xyz, const int* idx,
                          const float* feats, const float* rk, const float* Wp, const float* bias, float* out, const int* order,
                          double* stat_part, hipStream_t st) {
    constexpr int CCH = CIN > 64 ? 64 : CIN;
    const size_t lds = (size_t)(16 * (CCH * KS / (CCH >= 32 ? 2 : 1) + 40) + 4 * 16 * (COUT + 4) + 16 * MAXT * 5) * sizeof(float);
    constexpr int PD = INTER_PD(CIN, MAXT);
    auto kern = inter_so3conv_kernel<CIN, COUT, MAXT, PD, BX>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const unsigned gx = order ? 8u * (unsigned)((p2 + 7) / 8) : (unsigned)p2;
    hipLaunchKernelGGL(kern, dim3(gx, b), dim3(256), lds, st, p1, p2, nn, 1.0f / sigma, xyz, new_xyz, idx, feats, rk, Wp, bias, out, order,
                       stat_part);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

template <int CIN, int COUT, bool BX>
static int launch_inter(int b, int p1, int p2, int nn, float sigma, const float* xyz, const float* new_xyz, const int* idx,
                        const float* feats, const float* rk, const float* Wp, const float* bias, float* out, const int* order,
                        double* stat_part, hipStream_t st) {
    if (nn <= 16) return launch_inter_t<CIN, COUT, 1, BX>(b, p1, p2, nn, sigma, xyz, new_xyz, idx, feats, rk, Wp, bias, out, order, stat_part, st);
    if (nn <= 32) return launch_inter_t<CIN, COUT, 2, BX>(b, p1, p2, nn, sigma, xyz, new_xyz, idx, feats, rk, Wp, bias, out, order, stat_part, st);
    return launch_inter_t<CIN, COUT, 4, BX>(b, p1, p2, nn, sigma, xyz, new_xyz, idx, feats, rk, Wp, bias, out, order, stat_part, st);
}

template <int C, int COUT>
static int launch_intra(int npts, int ppb, const float* X, const float* mean, const float* rstd, const int* intra_idx,
                        const float* Wp, const float* bias, float* Y, double* stat_part, hipStream_t st) {
    constexpr int PTS = C >= 128 ? 1 : 2;
    constexpr int LD = C == 16 ? 56 : C + 40;
    if (stat_part && ((ppb % PTS) != 0 || C >= 128)) return ETCH_EUNSUPPORTED;      // a workgroup's points must belong to one sample
    const size_t tile = (size_t)PTS * NA * LD * sizeof(float);
    const size_t dyn = tile <= 60 * 1024 ? 0 : tile;
    auto kern = intra_so3conv_kernel<C, COUT, PTS>;
    if (dyn + NA * 12 * sizeof(int) > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((npts + PTS - 1) / PTS), dim3(256), dyn, st, npts, ppb, X, mean,
                       rstd, intra_idx, Wp, bias, Y, stat_part);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

template <int C, int COUT>
static int launch_intra32(int npts, int ppb, const float* X, const float* mean, const float* rstd, const int* intra_idx,
                          const float* Wp32, const float* bias, float* Y, double* stat_part, hipStream_t st) {
    if (stat_part && (ppb % 2) != 0) return ETCH_EUNSUPPORTED;
    hipLaunchKernelGGL((intra_so3conv32_kernel<C, COUT>), dim3((npts + 1) / 2), dim3(256), 0, st, npts, ppb, X, mean, rstd, intra_idx, Wp32, bias, Y,
                       stat_part);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

extern "C" {

// The 32x32x2 form (c = cout in {32, 64}): Wp32 = ops.permute_weight_frag32 order.  Same result as etch_intra_so3conv_stats up to the order of the
// fp32 sums inside an output (the contraction is split over the two lane halves differently).
int etch_intra_so3conv32(int b, int c, int cout, int p, const float* X, const float* mean, const float* rstd, const int* intra_idx, const float* Wp32,
                         const float* bias, float* Y, double* stat_part, void* stream) {
    if (b <= 0 || p <= 0) return ETCH_OK;
    hipStream_t st = (hipStream_t)stream;
    if (c == 32 && cout == 32) return launch_intra32<32, 32>(b * p, p, X, mean, rstd, intra_idx, Wp32, bias, Y, stat_part, st);
    if (c == 64 && cout == 64) return launch_intra32<64, 64>(b * p, p, X, mean, rstd, intra_idx, Wp32, bias, Y, stat_part, st);
    return ETCH_EUNSUPPORTED;
}

// W layouts: `W` = reference layout [cout][cin*24] (index c*24+k); `Wp` = fragment order produced by
// etch_permute_weight_frag (only the MFMA path, cin % 16 == 0, reads it).
int etch_inter_so3conv_ordered(int b, int cin, int cout, int p1, int p2, int nn, float sigma, const float* xyz, const float* new_xyz,
                               const int* ball_idx, const float* feats, const float* rk, const float* W, const float* Wp,
                               const float* bias, float* out, const int* order, double* stat_part, void* stream);

int etch_inter_so3conv(int b, int cin, int cout, int p1, int p2, int nn, float sigma, const float* xyz, const float* new_xyz,
                       const int* ball_idx, const float* feats, const float* rk, const float* W, const float* Wp,
                       const float* bias, float* out, void* stream) {
    return etch_inter_so3conv_ordered(b, cin, cout, p1, p2, nn, sigma, xyz, new_xyz, ball_idx, feats, rk, W, Wp, bias, out, nullptr, nullptr, stream);
}

int etch_inter_so3conv_ordered(int b, int cin, int cout, int p1, int p2, int nn, float sigma, const float* xyz, const float* new_xyz,
                               const int* ball_idx, const float* feats, const float* rk, const float* W, const float* Wp,
                               const float* bias, float* out, const int* order, double* stat_part, void* stream) {
    if (b <= 0 || p2 <= 0) return ETCH_OK;
    if (nn <= 0 || nn > 64 || sigma <= 0.f) return ETCH_EINVAL;
    if (!feats && cin != 1) return ETCH_EINVAL;            // feats == NULL: all-ones features, the one-channel conv only
    hipStream_t st = (hipStream_t)stream;
#define INTER_CASE(CI, CO) \
    if (cin == CI && cout == CO) return launch_inter<CI, CO, false>(b, p1, p2, nn, sigma, xyz, new_xyz, ball_idx, feats, rk, Wp, bias, out, order, stat_part, st);
    INTER_CASE(16, 16) INTER_CASE(16, 32) INTER_CASE(32, 32) INTER_CASE(32, 64) INTER_CASE(64, 64)
    INTER_CASE(64, 128) INTER_CASE(128, 128) INTER_CASE(128, 256) INTER_CASE(256, 256)       // encoder depths 3 / 4 (models_pointcloud.py:34-48)
#undef INTER_CASE
    if (cin == 1 && cout <= 64 && (!stat_part || (256 % cout == 0 && nn * NA >= 1026))) {
        const size_t lds = ((size_t)4 * nn + (size_t)nn * NA + NA * KS + (size_t)cout * KS + nn) * sizeof(float);
        if (lds <= 64 * 1024) {
            if (feats) hipLaunchKernelGGL(inter_so3conv_c1_kernel<false>, dim3(p2, b), dim3(256), lds, st, cout, p1, p2, nn, 1.0f / sigma, xyz, new_xyz, ball_idx,
                                          feats, rk, W, bias, out, stat_part);
            else hipLaunchKernelGGL(inter_so3conv_c1_kernel<true>, dim3(p2, b), dim3(256), lds, st, cout, p1, p2, nn, 1.0f / sigma, xyz, new_xyz, ball_idx,
                                    feats, rk, W, bias, out, stat_part);
            ETCH_RETURN_IF_LAUNCH_FAILED();
            return ETCH_OK;
        }
    }
    if (stat_part || !feats) return ETCH_EUNSUPPORTED;     // the generic small-CIN kernel has no fused statistics (and reads its features)
    if (cin <= 8) {
        const size_t lds = (size_t)(4 * nn + NA * cin * KS) * sizeof(float);
        hipLaunchKernelGGL(inter_so3conv_small_kernel, dim3(p2, b), dim3(256), lds, st, cin, cout, p1, p2, nn, 1.0f / sigma, xyz,
                           new_xyz, ball_idx, feats, rk, W, bias, out);
        ETCH_RETURN_IF_LAUNCH_FAILED();
        return ETCH_OK;
    }
    return ETCH_EUNSUPPORTED;
}

// The same convolution with step 2 on the bf16 matrix cores (split fp32 operands, see inter_so3conv_kernel BX).  Wq = ops.inter_weight_split:
// [chunk of 32 kappas][o tile][plane hi / mid / lo][lane][8 bf16] in the kernel's contraction order.
int etch_inter_so3conv_split(int b, int cin, int cout, int p1, int p2, int nn, float sigma, const float* xyz, const float* new_xyz,
                             const int* ball_idx, const float* feats, const float* rk, const void* Wq, const float* bias, float* out,
                             const int* order, double* stat_part, void* stream) {
    if (b <= 0 || p2 <= 0) return ETCH_OK;
    if (nn <= 0 || nn > 64 || sigma <= 0.f || !Wq) return ETCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const float* Wp = reinterpret_cast<const float*>(Wq);
#define INTER_CASE(CI, CO) \
    if (cin == CI && cout == CO) return launch_inter<CI, CO, true>(b, p1, p2, nn, sigma, xyz, new_xyz, ball_idx, feats, rk, Wp, bias, out, order, stat_part, st);
    INTER_CASE(16, 16) INTER_CASE(16, 32) INTER_CASE(32, 32) INTER_CASE(32, 64) INTER_CASE(64, 64)
    INTER_CASE(64, 128) INTER_CASE(128, 128) INTER_CASE(128, 256) INTER_CASE(256, 256)
#undef INTER_CASE
    return ETCH_EUNSUPPORTED;
}

int etch_intra_so3conv_stats(int b, int c, int cout, int p, const float* X, const float* mean, const float* rstd,
                             const int* intra_idx, const float* Wp, const float* bias, float* Y, double* stat_part, void* stream) {
    if (b <= 0 || p <= 0) return ETCH_OK;
    hipStream_t st = (hipStream_t)stream;
#define INTRA_CASE(CI, CO) \
    if (c == CI && cout == CO) return launch_intra<CI, CO>(b * p, p, X, mean, rstd, intra_idx, Wp, bias, Y, stat_part, st);
    INTRA_CASE(16, 16) INTRA_CASE(32, 32) INTRA_CASE(64, 64) INTRA_CASE(128, 128) INTRA_CASE(256, 256)
#undef INTRA_CASE
    return ETCH_EUNSUPPORTED;
}

int etch_intra_so3conv(int b, int c, int cout, int p, const float* X, const float* mean, const float* rstd,
                       const int* intra_idx, const float* Wp, const float* bias, float* Y, void* stream) {
    return etch_intra_so3conv_stats(b, c, cout, p, X, mean, rstd, intra_idx, Wp, bias, Y, nullptr, stream);
}

// workspace: IN_CHUNKS * b * 2 * C doubles
int etch_instnorm_stats(int b, int rows, int C, const float* x, double* workspace, float* mean, float* rstd, void* stream) {
    if (b <= 0) return ETCH_OK;
    if (C < 4 || C > 256 || (C & 3) || (256 % (C >> 2)) != 0 || ((uintptr_t)x & 15)) return ETCH_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(instnorm_partial_kernel, dim3(IN_CHUNKS, b), dim3(256), 2 * 1024 * sizeof(double), st, rows, C, x, workspace);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(instnorm_final_kernel, dim3(b), dim3(256), 0, st, rows, C, 1e-5f, workspace, mean, rstd);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_instnorm_from_partials(int b, int nparts, int C, int count, const double* partial, float* mean, float* rstd, void* stream) {
    if (b <= 0) return ETCH_OK;
    if (C < 4 || C > 256 || (C & 3) || (1024 % (C >> 2)) != 0 || nparts <= 0 || count <= 0 || ((uintptr_t)partial & 15)) return ETCH_EUNSUPPORTED;
    hipLaunchKernelGGL(instnorm_from_partials_kernel, dim3(b), dim3(1024), 0, (hipStream_t)stream, nparts, C, count, 1e-5f, partial, mean, rstd);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_instnorm_stats_workspace_bytes(int b, int C) { return (int)((size_t)IN_CHUNKS * b * 2 * C * sizeof(double)); }

static int instnorm_act_add_launch(bool f16, int b, int rows, int C, const float* x1, const float* m1, const float* r1, const float* x2,
                                   const float* m2, const float* r2, float* out, void* planes, void* stream, bool k1 = false) {
    if (b <= 0 || rows <= 0) return ETCH_OK;
    if (k1 && (!f16 || !x2 || !m2 || !r2)) return ETCH_EINVAL;
    if ((C & 3) || ((uintptr_t)planes & 7)) return ETCH_EUNSUPPORTED;
    const long per4l = (long)rows * C / 4;
    if (per4l >= (1L << 30) || b > 65535) return ETCH_EUNSUPPORTED;      // 32-bit element indices inside a sample
    const unsigned per4 = (unsigned)per4l;
    long bx = (per4l + 511) / 512;
    const long cap = (256L * 16 + b - 1) / b;
    if (bx > cap) bx = cap;
    if (bx < 1) bx = 1;
    const bool pow2 = (C & (C - 1)) == 0;
    int cshift = 0;
    while ((1 << cshift) < C) ++cshift;
    const dim3 grid((unsigned)bx, (unsigned)b);
    unsigned short* pl = reinterpret_cast<unsigned short*>(planes);
    hipStream_t st = (hipStream_t)stream;
#define INA_LAUNCH(F, P2) hipLaunchKernelGGL((instnorm_act_add_kernel<F, P2>), grid, dim3(256), 0, st, per4, rows, C, cshift, x1, m1, r1, x2, m2, r2, out, pl)
    if (k1) {
        if (pow2) hipLaunchKernelGGL((instnorm_act_add_kernel<true, true, true>), grid, dim3(256), 0, st, per4, rows, C, cshift, x1, m1, r1, x2, m2, r2, out, pl);
        else hipLaunchKernelGGL((instnorm_act_add_kernel<true, false, true>), grid, dim3(256), 0, st, per4, rows, C, cshift, x1, m1, r1, x2, m2, r2, out, pl);
    } else if (f16) { if (pow2) INA_LAUNCH(true, true); else INA_LAUNCH(true, false); }
    else { if (pow2) INA_LAUNCH(false, true); else INA_LAUNCH(false, false); }
#undef INA_LAUNCH
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_instnorm_act_add_planes(int b, int rows, int C, const float* x1, const float* m1, const float* r1, const float* x2,
                                 const float* m2, const float* r2, float* out, void* planes, void* stream) {
    return instnorm_act_add_launch(false, b, rows, C, x1, m1, r1, x2, m2, r2, out, planes, stream);
}

int etch_instnorm_act_add_planes_f16(int b, int rows, int C, const float* x1, const float* m1, const float* r1, const float* x2,
                                     const float* m2, const float* r2, float* out, void* planes, void* stream) {
    return instnorm_act_add_launch(true, b, rows, C, x1, m1, r1, x2, m2, r2, out, planes, stream);
}

int etch_instnorm_act_add_k1_planes_f16(int b, int rows, int C, const float* x1, const float* m1, const float* r1, const float* f, const float* slope,
                                        const float* offset, float* out, void* planes, void* stream) {
    return instnorm_act_add_launch(true, b, rows, C, x1, m1, r1, f, slope, offset, out, planes, stream, true);
}

int etch_instnorm_act_add(int b, int rows, int C, const float* x1, const float* m1, const float* r1, const float* x2,
                          const float* m2, const float* r2, float* out, void* stream) {
    return etch_instnorm_act_add_planes(b, rows, C, x1, m1, r1, x2, m2, r2, out, nullptr, stream);
}

}  // extern "C"
