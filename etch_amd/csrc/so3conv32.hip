// Fused inter-SO(3) convolution on v_mfma_f32_32x32x2_f32, the 32-wide sibling of inter_so3conv_kernel (so3conv.hip).
//
// Why a second kernel: on this MI355X the 16x16x4 fp32 MFMA never issues faster than 0.85 of the matrix peak (~37 instead of 32 cycles per
// instruction; 125 - 134 TFLOP/s), the 32x32x2 form runs at 0.986 of it (155 TFLOP/s) -- profiles/r03_mfma_issue_rate.txt.  Both steps of the
// 16-wide kernel already run at 117 - 124 TFLOP/s of issued MFMAs, i.e. at that instruction's ceiling.
//
// A 32-wide N needs 32 columns in step 2: a workgroup (8 waves) therefore owns TWO output points (neighbours on the Morton curve), the columns
// of an anchor group are 16 anchors of point A | 16 anchors of point B, and every W fragment serves both points (half the L2 weight stream per
// point).  Replaces, like the 16-wide kernel, inter_so3conv_grouping_anchor + inter_so3conv_feat_grouping + BasicSO3Conv
// (/root/reference/external/vgtk/vgtk/so3conv/functional.py:286-324, :61-67, modules.py:33-39); nothing is materialised.
//
// Per anchor group (16 anchors x 2 points) and 32-channel tile h of the input channels:
//   step 1  wave w: point w >> 2, anchors 4 (w & 3) .. + 3:  X1[k][c] = sum_n w[a,k,n] F[idx_n, a, 32 h + c]
//           MFMA M = 32 kernel-point rows (24 used), N = 32 channels, K = 2 neighbours; the weights relu(1 - |g_n - R_a kappa_k|^2 / sigma) are
//           generated in registers as the A operand (one per lane and MFMA), the gathered features are the B operand (lane = channel: a wave
//           reads two contiguous 128-byte runs per load).  D: lane holds channel j = lane % 32, kernel points 8 g + 4 (lane / 32) + q in
//           registers 4 g + q -- the kernel-point groups g = 0, 1, 2 are the K-slices of step 2.
//   step 2  per slice (h, g): the 4 x 2 waves store their anchors' [32 ch][8 k] blocks as columns of the X1 tile [32 cols][256] in LDS (one
//           16-byte store per anchor), then wave w multiplies output-channel tile mt = w % MT by its share kp = w / MT of the slice's
//           contraction: Y[o][col] += W[o][kappa] X1[col][kappa], W streamed from L2 in fragment order (ops.inter_weight_frag32).
//   after the last slice: the K-split partial tiles meet in LDS, bias, output, InstanceNorm statistics (fp64) as in the 16-wide kernel.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define NA 60
#define KS 24
#define I32_SLD 260          // X1 tile row stride (floats): 256 + 4
#define I32_BUF(COUT) ((8 / ((COUT) / 32)) * 32 * ((COUT) + 4) > 32 * I32_SLD ? (8 / ((COUT) / 32)) * 32 * ((COUT) + 4) : 32 * I32_SLD)

template <int CIN, int COUT, int MAXC, int PD, int WGS>     // MAXC = ceil(nn / 8) neighbour chunks; PD = gather prefetch distance (chunk-steps); WGS = workgroups per CU
__global__ void __launch_bounds__(512, 2 * WGS) inter_so3conv32_kernel(
    int p1, int p2, int nn, float inv_sigma, const float* __restrict__ xyz, const float* __restrict__ new_xyz,
    const int* __restrict__ ball_idx, const float* __restrict__ feats, const float* __restrict__ rk,
    const float* __restrict__ Wp, const float* __restrict__ bias, float* __restrict__ out, const int* __restrict__ order,
    double* __restrict__ stat_part) {
    constexpr int NTIL = CIN / 32;         // 32-channel tiles of the input
    constexpr int MT = COUT / 32;          // 32-row tiles of the output channels
    constexpr int NKP = 8 / MT;            // K-split of a slice over the waves
    constexpr int KP = 256 / NKP;          // contraction length per wave and slice
    constexpr int NU = KP / 8;             // k-steps (8 kappas) per wave and slice
    constexpr int PS = COUT + 4;           // partial-tile row stride
    constexpr int NBR = 8 * MAXC;          // neighbour slots
    constexpr int BUF = I32_BUF(COUT);     // floats per LDS buffer: an X1 slice [32 cols][I32_SLD] or the partial tiles [NKP][32 cols][PS]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* nbt = reinterpret_cast<float4*>(smem + 2 * BUF);         // [2 points][NBR]
    unsigned* noff = reinterpret_cast<unsigned*>(nbt + 2 * NBR);     // [2 points][NBR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, kk = lane >> 5;
    const int b = blockIdx.y;
    // two consecutive slots of the (spatially ordered) schedule; workgroup ids go round-robin over the 8 XCDs
    const int per = gridDim.x >> 3;
    const int pair = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (2 * pair >= p2) return;
    int pts[2];
    pts[0] = order ? order[(size_t)b * p2 + 2 * pair] : 2 * pair;
    const bool haveB = 2 * pair + 1 < p2;
    pts[1] = haveB ? (order ? order[(size_t)b * p2 + 2 * pair + 1] : 2 * pair + 1) : pts[0];

    // neighbour tables of both points: nbt[n] = (2 g / sigma, 1 - |g|^2 / sigma), noff[n] = float offset of the neighbour's anchor-0 row
    if (tid < 2 * NBR) {
        const int pt = tid / NBR, n = tid - pt * NBR, p = pts[pt];
        const int* row = ball_idx + ((size_t)b * p2 + p) * nn;
        int q = row[n < nn ? n : nn - 1];
        q = n < nn ? q : -1;
        const int qq = q < 0 ? 0 : q;
        const float* X = xyz + (size_t)b * 3 * p1;
        const float x = X[qq] - new_xyz[((size_t)b * 3 + 0) * p2 + p], y = X[p1 + qq] - new_xyz[((size_t)b * 3 + 1) * p2 + p],
                    z = X[2 * p1 + qq] - new_xyz[((size_t)b * 3 + 2) * p2 + p];
        nbt[tid] = make_float4(2.0f * inv_sigma * x, 2.0f * inv_sigma * y, 2.0f * inv_sigma * z,
                               q < 0 ? -1e30f : 1.0f - (x * x + y * y + z * z) * inv_sigma);
        noff[tid] = (unsigned)qq * (unsigned)(NA * CIN);
    }
    __syncthreads();
    const int pt = wave >> 2, wv = wave & 3;                   // step 1: this wave's point and anchor quad
    const float4* mynbt = nbt + pt * NBR;
    const unsigned* mynoff = noff + pt * NBR;
    const float* Fb = feats + (size_t)b * p1 * NA * CIN + j;   // this lane's channel of a 32-channel tile
    const int mt = wave % MT, kp = wave / MT;                  // step 2: this wave's output-channel tile and K share
    const int nch = (nn + 7) >> 3;
    const bool kvalid = j < KS;                                // rows 24..31 of the kernel-point tile are padding
    const int jk = kvalid ? j : 0;
    double st_s[2] = {0.0, 0.0}, st_q[2] = {0.0, 0.0};         // fused InstanceNorm statistics of point A / B (this thread's channel is fixed)

    // Gathers run PD chunk-steps ahead through a register ring, across anchors and tile passes (the first chunks of the next pass are in
    // flight during step 2); the rotated kernel points one anchor pair ahead.  A chunk-step = 8 neighbours of one anchor: 4 dword loads
    // (a wave reads two 128-byte runs per load) and 4 MFMAs; two anchors are interleaved (two independent accumulator chains).
    constexpr int NCS = 4 * MAXC;          // chunk-steps per wave and tile pass, order: (anchor pair, chunk, anchor of the pair)
    constexpr int NRING = PD + 2;          // two chunk-steps are consumed and two issued per iteration: the issue lands in the previous iteration's slots
    static_assert(PD % 2 == 0 && NCS % NRING == 0, "ring slots must line up across tile passes");
    float ring[NRING][4];
    float rkn[2][3];
    auto issue = [&](int tp_, int cs, float (&dst)[4]) {
        const int pr = cs / (2 * MAXC), c = (cs % (2 * MAXC)) >> 1, ja = 2 * pr + (cs & 1);
        int a = (tp_ / NTIL) * 16 + wv * 4 + ja;
        a = a < NA ? a : NA - 1;
        const float* Fa = Fb + (size_t)a * CIN + 32 * (tp_ % NTIL);      // wave-uniform base + 32-bit lane offset
#pragma unroll
        for (int s = 0; s < 4; ++s) dst[s] = Fa[mynoff[8 * c + 2 * s + kk]];
    };
    auto issue_rk = [&](int q, int a) {
        a = a < NA ? a : NA - 1;
        const float* rka = rk + ((size_t)a * KS + jk) * 3;
        rkn[q][0] = rka[0]; rkn[q][1] = rka[1]; rkn[q][2] = rka[2];
    };
    issue_rk(0, wv * 4); issue_rk(1, wv * 4 + 1);
#pragma unroll
    for (int c = 0; c < PD; ++c) issue(0, c, ring[c % NRING]);

    float* wbuf = smem;                    // the buffer the next slice is written to
    float* obuf = smem + BUF;
    f32x16 yacc;
#pragma unroll 1
    for (int tp = 0; tp < 4 * NTIL; ++tp) {
        const int ag = tp / NTIL, h = tp % NTIL;
        if (h == 0) {
#pragma unroll
            for (int v = 0; v < 16; ++v) yacc[v] = 0.f;
        }
        // ---------------- step 1: 4 anchors (2 interleaved pairs), channel tile h
        f32x4 keep[4][2];                                      // kernel-point groups 1, 2 of the 4 anchors (group 0 goes to LDS at once)
        f32x16 acc[2];
        float rx[2], ry[2], rz[2], rb[2];
#pragma unroll
        for (int cp = 0; cp < NCS / 2; ++cp) {
            const int pr = cp / MAXC, c = cp % MAXC;
            const int a0 = ag * 16 + wv * 4 + 2 * pr;           // anchors a0, a0 + 1
            if (c == 0) {                                       // pair start
                asm volatile("" ::: "memory");                  // keeps the (loop-invariant) neighbour-table reads in LDS instead of hoisted registers
#pragma unroll
                for (int q = 0; q < 2; ++q) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[q][v] = 0.f;
                    rx[q] = rkn[q][0]; ry[q] = rkn[q][1]; rz[q] = rkn[q][2];
                    rb[q] = kvalid ? -(rx[q] * rx[q] + ry[q] * ry[q] + rz[q] * rz[q]) * inv_sigma : -1e30f;
                    issue_rk(q, (pr == 0 ? a0 + 2 : ((tp + 1) / NTIL) * 16 + wv * 4) + q);     // next pair's kernel points
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int nc = 2 * cp + q + PD;
                if (nc < NCS) issue(tp, nc, ring[nc % NRING]);
                else if (tp < 4 * NTIL - 1) issue(tp + 1, nc - NCS, ring[nc % NRING]);          // wave-uniform
            }
            if (a0 < NA && c < nch) {                           // wave-uniform (anchors come in fours: a0 < 60 <=> a0 + 1 < 60)
                float (&f0)[4] = ring[(2 * cp) % NRING];
                float (&f1)[4] = ring[(2 * cp + 1) % NRING];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float4 g = mynbt[8 * c + 2 * s + kk];
                    const float w0 = fmaxf(0.f, fmaf(g.z, rz[0], fmaf(g.y, ry[0], fmaf(g.x, rx[0], g.w + rb[0]))));
                    const float w1 = fmaxf(0.f, fmaf(g.z, rz[1], fmaf(g.y, ry[1], fmaf(g.x, rx[1], g.w + rb[1]))));
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, f0[s], acc[0], 0, 0, 0);      // D[k][c]
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, f1[s], acc[1], 0, 0, 0);
                }
            }
            if (c == MAXC - 1) {                                // pair end: lane = channel j, registers 4 g + q = kernel points 8 g + 4 kk + q
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int ja = 2 * pr + q;
                    *reinterpret_cast<f32x4*>(&wbuf[(16 * pt + 4 * wv + ja) * I32_SLD + j * 8 + 4 * kk]) = (f32x4){acc[q][0], acc[q][1], acc[q][2], acc[q][3]};
                    keep[ja][0] = (f32x4){acc[q][4], acc[q][5], acc[q][6], acc[q][7]};
                    keep[ja][1] = (f32x4){acc[q][8], acc[q][9], acc[q][10], acc[q][11]};
                }
            }
        }
        // ---------------- step 2: the three kernel-point slices of this channel tile, one barrier each (the X1 slice is double-buffered:
        // slice s + 1 is written after the barrier of slice s, which every wave passes after it finished reading slice s - 1)
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            if (g > 0) {
#pragma unroll
                for (int ja = 0; ja < 4; ++ja)
                    *reinterpret_cast<f32x4*>(&wbuf[(16 * pt + 4 * wv + ja) * I32_SLD + j * 8 + 4 * kk]) = keep[ja][g - 1];
            }
            __syncthreads();
            const int sl = h * 3 + g;
            const float* wbase = Wp + ((((size_t)sl * MT + mt) * NKP + kp) * NU) * 256 + lane * 4;
            const float* xbase = &wbuf[j * I32_SLD + kp * KP + 4 * kk];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const float4 av = *reinterpret_cast<const float4*>(wbase + u * 256);
                const float4 bv = *reinterpret_cast<const float4*>(xbase + 8 * u);
                yacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, yacc, 0, 0, 0);
                yacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, yacc, 0, 0, 0);
                yacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, yacc, 0, 0, 0);
                yacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, yacc, 0, 0, 0);
            }
            float* t_ = wbuf; wbuf = obuf; obuf = t_;
        }
        if (h < NTIL - 1) continue;
        // ---------------- the K-split partial tiles meet (in the buffer nobody reads any more: the slice before last);
        // yacc[v] = Y[o = 32 mt + 8 (v / 4) + 4 kk + v % 4][col = j]
        float* part = wbuf;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(&part[(kp * 32 + j) * PS + 32 * mt + 8 * g + 4 * kk]) = (f32x4){yacc[4 * g], yacc[4 * g + 1], yacc[4 * g + 2], yacc[4 * g + 3]};
        __syncthreads();                                        // also: every wave finished reading the last slice (obuf)
        constexpr int NIT = 32 * COUT / 512;                    // the first NIT / 2 iterations are point A's columns, the rest point B's
#pragma unroll
        for (int itr = 0; itr < NIT; ++itr) {
            const int e = tid + 512 * itr;
            const int col = e / COUT, o = e - col * COUT;
            const int which = itr >= NIT / 2 ? 1 : 0, a = ag * 16 + (col & 15);
            if (a < NA && (which == 0 || haveB)) {
                float v = part[col * PS + o];
#pragma unroll
                for (int q = 1; q < NKP; ++q) v += part[(q * 32 + col) * PS + o];
                v += bias[o];
                out[(((size_t)b * p2 + pts[which]) * NA + a) * COUT + o] = v;
                st_s[which] += (double)v; st_q[which] += (double)v * (double)v;
            }
        }
        { float* t_ = wbuf; wbuf = obuf; obuf = t_; }            // the next slice goes to the last slice's buffer; the partial tiles are overwritten
                                                                // one barrier later, behind every wave's output loop
    }
    if (stat_part) {
        static_assert(512 % COUT == 0, "a thread must keep one output channel");
        double* dred = reinterpret_cast<double*>(smem);      // 4 x 512 doubles = 16 KB
        __syncthreads();
        dred[tid] = st_s[0]; dred[512 + tid] = st_q[0]; dred[1024 + tid] = st_s[1]; dred[1536 + tid] = st_q[1];
        __syncthreads();
        if (tid < 2 * COUT) {
            const int which = tid / COUT, o = tid - which * COUT;
            if (which == 0 || haveB) {
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int k = 0; k < 512 / COUT; ++k) { a0 += dred[which * 1024 + k * COUT + o]; a1 += dred[which * 1024 + 512 + k * COUT + o]; }
                double* sp = stat_part + ((size_t)b * p2 + pts[which]) * 2 * COUT;
                sp[o] = a0; sp[COUT + o] = a1;
            }
        }
    }
}

template <int CIN, int COUT, int MAXC, int PD, int WGS>
static int launch_inter32_t(int b, int p1, int p2, int nn, float sigma, const float* xyz, const float* new_xyz, const int* idx, const float* feats,
                            const float* rk, const float* Wp32, const float* bias, float* out, const int* order, double* stat_part, hipStream_t st) {
    const size_t lds = (size_t)(2 * I32_BUF(COUT) + 2 * 8 * MAXC * 5) * sizeof(float);
    auto kern = inter_so3conv32_kernel<CIN, COUT, MAXC, PD, WGS>;
    static bool ready = false;
    if (!ready) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        ready = true;
    }
    const unsigned pairs = (unsigned)((p2 + 1) / 2);
    hipLaunchKernelGGL(kern, dim3(8u * ((pairs + 7) / 8), b), dim3(512), lds, st, p1, p2, nn, 1.0f / sigma, xyz, new_xyz, idx, feats, rk, Wp32, bias, out,
                       order, stat_part);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

#ifndef I32_PD
#define I32_PD 2
#endif
#ifndef I32_WGS
#define I32_WGS 1
#endif
template <int CIN, int COUT>
static int launch_inter32(int b, int p1, int p2, int nn, float sigma, const float* xyz, const float* new_xyz, const int* idx, const float* feats,
                          const float* rk, const float* Wp32, const float* bias, float* out, const int* order, double* stat_part, hipStream_t st) {
    if (nn <= 16) return launch_inter32_t<CIN, COUT, 2, I32_PD, I32_WGS>(b, p1, p2, nn, sigma, xyz, new_xyz, idx, feats, rk, Wp32, bias, out, order, stat_part, st);
    if (nn <= 32) return launch_inter32_t<CIN, COUT, 4, I32_PD, I32_WGS>(b, p1, p2, nn, sigma, xyz, new_xyz, idx, feats, rk, Wp32, bias, out, order, stat_part, st);
    return launch_inter32_t<CIN, COUT, 8, I32_PD, I32_WGS>(b, p1, p2, nn, sigma, xyz, new_xyz, idx, feats, rk, Wp32, bias, out, order, stat_part, st);
}

// Wp32 = ops.inter_weight_frag32 order: [slice = 3 h + g][mt][kp][u][lane][4] with
//   Wp32[...][lane][s] = W[32 mt + lane % 32][(32 h + c) * 24 + 8 g + 4 (lane / 32) + s],  c = (kp * KP + 8 u) / 8.
extern "C" int etch_inter_so3conv32(int b, int cin, int cout, int p1, int p2, int nn, float sigma, const float* xyz, const float* new_xyz,
                                    const int* ball_idx, const float* feats, const float* rk, const float* Wp32, const float* bias, float* out,
                                    const int* order, double* stat_part, void* stream) {
    if (b <= 0 || p2 <= 0) return ETCH_OK;
    if (nn <= 0 || nn > 64 || sigma <= 0.f || !Wp32) return ETCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define INTER32_CASE(CI, CO) \
    if (cin == CI && cout == CO) return launch_inter32<CI, CO>(b, p1, p2, nn, sigma, xyz, new_xyz, ball_idx, feats, rk, Wp32, bias, out, order, stat_part, st);
    INTER32_CASE(32, 32) INTER32_CASE(32, 64) INTER32_CASE(64, 64)
#undef INTER32_CASE
    return ETCH_EUNSUPPORTED;
}
