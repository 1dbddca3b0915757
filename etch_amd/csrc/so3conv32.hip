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

// The operand ring.  Everything the matrix core consumes from global memory -- the gathered feature dwords of step 1, the weight fragments
// of step 2 -- comes through ONE register ring in consumption order, LA groups ahead (a group = what four MFMAs need: 4 gathered dwords, or one
// 16-byte weight fragment), built from inline-asm loads and counted waits: plain loads get sunk in front of their MFMAs by the compiler
// (s_waitcnt vmcnt(1) per MFMA group in step 2: the L2 latency of every weight fragment exposed, 52 - 60 % matrix-pipe occupancy).
// Rules that keep this safe (as for intra_so3conv32_kernel, so3conv.hip): no load is left in flight whose registers the compiler has
// released (the ring is drained after the loop with every slot as an operand); a wait names its slot as an in/out operand, so the consuming
// MFMAs cannot move above it; loads return in order and vmcnt counts every vector-memory instruction, so foreign loads / stores in flight
// only make a counted wait stricter.  The loop itself issues no other vector-memory load (kernel points and neighbour tables sit in LDS,
// the bias in a register).
__device__ __forceinline__ void i32_gload(float& dst, unsigned voff, const float* sbase) {
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase));
}
__device__ __forceinline__ void i32_wload(f32x4& dst, unsigned voff, const float* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase));
}
template <int N> __device__ __forceinline__ void i32_gwait(float (&v)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "n"(N));
}
template <int N> __device__ __forceinline__ void i32_wwait(f32x4& v) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N)); }
// ng = gather groups among the LA groups issued behind the awaited one: 4 ng + (LA - ng) load instructions
template <int LA, int K = 0> __device__ __forceinline__ void i32_gwait_ng(int ng, float (&v)[4]) {
    if (ng == K) i32_gwait<LA + 3 * K>(v);
    else if constexpr (K < LA) i32_gwait_ng<LA, K + 1>(ng, v);
}
template <int LA, int K = 0> __device__ __forceinline__ void i32_wwait_ng(int ng, f32x4& v) {
    if (ng == K) i32_wwait<LA + 3 * K>(v);
    else if constexpr (K < LA) i32_wwait_ng<LA, K + 1>(ng, v);
}

#ifndef I32_ABL
#define I32_ABL 0      // timing experiments only (wrong results): 1 no step-1 MFMAs, 2 no step-2 MFMAs, 4 no gathers, 8 no weight loads, 16 no barriers
#endif
#if I32_ABL & 16
#define I32_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define I32_BARRIER() __syncthreads()
#endif
template <int CIN, int COUT, int MAXC, int NRING>     // MAXC = ceil(nn / 8) neighbour chunks; NRING = ring slots (look-ahead NRING - 2 groups)
__global__ void __launch_bounds__(512, 2) inter_so3conv32_kernel(
    int p1, int p2, int nn, float inv_sigma, const float* __restrict__ xyz, const float* __restrict__ new_xyz,
    const int* __restrict__ ball_idx, const float* __restrict__ feats, const float* __restrict__ rk,
    const float* __restrict__ Wp, const float* __restrict__ bias, float* __restrict__ out, const int* __restrict__ order,
    double* __restrict__ stat_part) {
    constexpr int NTIL = CIN / 32;         // 32-channel tiles of the input
    constexpr int NTP = 4 * NTIL;          // tile passes: (anchor group, channel tile)
    constexpr int MT = COUT / 32;          // 32-row tiles of the output channels
    constexpr int NKP = 8 / MT;            // K-split of a slice over the waves
    constexpr int KP = 256 / NKP;          // contraction length per wave and slice
    constexpr int NU = KP / 8;             // k-steps (8 kappas) per wave and slice
    constexpr int PS = COUT + 4;           // partial-tile row stride
    constexpr int NBR = 8 * MAXC;          // neighbour slots
    constexpr int BUF = I32_BUF(COUT);     // floats per LDS buffer: an X1 slice [32 cols][I32_SLD] or the partial tiles [NKP][32 cols][PS]
    constexpr int NCS = 4 * MAXC;          // gather groups per wave and tile pass, order: (anchor pair, chunk, anchor of the pair)
    constexpr int NG = NCS + 3 * NU;       // + the weight groups of the three slices
    constexpr int LA = NRING - 2;          // look-ahead in groups; one slot of slack: step 1 consumes two groups (an anchor pair's) at a time
    static_assert(NG % NRING == 0 && LA >= 1 && LA < NG, "ring slots must line up across tile passes");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* nbt = reinterpret_cast<float4*>(smem + 2 * BUF);         // [2 points][NBR]
    float4* rkt = nbt + 2 * NBR;                                     // [60][24]: rotated kernel point, -|r|^2 / sigma
    uint4* noffq = reinterpret_cast<uint4*>(rkt + NA * KS);          // [2 points][MAXC chunks][2 kk]: byte offsets of neighbours 8 c + 2 s + kk, s = 0..3
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

    // neighbour tables of both points: nbt[n] = (2 g / sigma, 1 - |g|^2 / sigma), noffq = byte offsets of the neighbours' anchor-0 rows
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
        reinterpret_cast<unsigned*>(noffq)[pt * NBR + (n >> 3) * 8 + (n & 1) * 4 + ((n & 7) >> 1)] = (unsigned)qq * (unsigned)(NA * CIN * 4);
    }
    for (int e = tid; e < NA * KS; e += 512) {
        const float x = rk[e * 3], y = rk[e * 3 + 1], z = rk[e * 3 + 2];
        rkt[e] = make_float4(x, y, z, -(x * x + y * y + z * z) * inv_sigma);
    }
    const float bo = bias[tid % COUT];                         // this thread's output channel is fixed (512 % COUT == 0)
    __syncthreads();
    const int pt = wave >> 2, wv = wave & 3;                   // step 1: this wave's point and anchor quad
    const float4* mynbt = nbt + pt * NBR;
    const uint4* mynoffq = noffq + pt * 2 * MAXC + kk;
    const float* Fb = feats + (size_t)b * p1 * NA * CIN;
    const int mt = wave % MT, kp = wave / MT;                  // step 2: this wave's output-channel tile and K share
    const float* Wme = Wp + (size_t)(mt * NKP + kp) * NU * 256;
    const bool kvalid = j < KS;                                // rows 24..31 of the kernel-point tile are padding
    const int jk = kvalid ? j : 0;
    const unsigned lane16 = lane * 16, j4 = j * 4;
    double st_s[2] = {0.0, 0.0}, st_q[2] = {0.0, 0.0};         // fused InstanceNorm statistics of point A / B

    float gr[NRING][4];                    // ring slots as gather groups ...
    f32x4 wr[NRING];                       // ... and as weight groups (a position uses one of the two)
    // group gx of tile pass P into slot sl
    auto issue = [&](int P, int gx, int sl) {
        if (gx < NCS) {
            const int pr = gx / (2 * MAXC), c = (gx % (2 * MAXC)) >> 1, ja = 2 * pr + (gx & 1);
            int a = (P / NTIL) * 16 + wv * 4 + ja;
            a = a < NA ? a : NA - 1;
            const float* sb = Fb + a * CIN + 32 * (P % NTIL);               // wave-uniform
            const uint4 o = mynoffq[2 * c];
#if I32_ABL & 4
            asm volatile("v_mov_b32 %0, %1" : "=v"(gr[sl][0]) : "v"(o.x)); asm volatile("v_mov_b32 %0, %1" : "=v"(gr[sl][1]) : "v"(o.y));
            asm volatile("v_mov_b32 %0, %1" : "=v"(gr[sl][2]) : "v"(o.z)); asm volatile("v_mov_b32 %0, %1" : "=v"(gr[sl][3]) : "v"(o.w)); (void)sb;
#else
            i32_gload(gr[sl][0], o.x + j4, sb); i32_gload(gr[sl][1], o.y + j4, sb); i32_gload(gr[sl][2], o.z + j4, sb); i32_gload(gr[sl][3], o.w + j4, sb);
#endif
        } else {
            const int w = gx - NCS;                                         // = g * NU + u; slices are contiguous in Wp32
#if I32_ABL & 8
            asm volatile("v_mov_b32 %0, %1" : "=v"(wr[sl].x) : "v"(lane16)); wr[sl].y = wr[sl].x; wr[sl].z = wr[sl].x; wr[sl].w = wr[sl].x;
#else
            i32_wload(wr[sl], lane16, Wme + (size_t)(((P % NTIL) * 3 + w / NU) * 8 * NU + w % NU) * 256);
#endif
        }
    };
    // look-ahead of position i of pass tp (next pass tpn), and the number of gather groups among the LA groups behind i
    auto ahead = [&](int tp, int tpn, int i) { const int gi = i + LA; issue(gi < NG ? tp : tpn, gi % NG, gi % NRING); };
    auto ngather = [&](int i) { int n = 0; for (int k = 1; k <= LA; ++k) n += ((i + k) % NG) < NCS ? 1 : 0; return n; };
#pragma unroll
    for (int i = 0; i < LA; ++i) issue(0, i, i % NRING);

    float* wbuf = smem;                    // the buffer the next slice is written to
    float* obuf = smem + BUF;
    f32x16 yacc[2];
#pragma unroll 1
    for (int tp = 0; tp < NTP; ++tp) {
        const int tpn = tp + 1 < NTP ? tp + 1 : 0;            // (the last pass looks ahead into pass 0 again: loads nobody uses, drained below)
        const int ag = tp / NTIL, h = tp % NTIL;
        if (h == 0) {
#pragma unroll
            for (int v = 0; v < 16; ++v) { yacc[0][v] = 0.f; yacc[1][v] = 0.f; }
        }
        // ---------------- step 1: 4 anchors (2 interleaved pairs), channel tile h
        f32x4 keep[4][2];                                      // kernel-point groups 1, 2 of the 4 anchors (group 0 goes to LDS at once)
        f32x16 acc[2];
        float4 r[2];
#pragma unroll
        for (int cp = 0; cp < NCS / 2; ++cp) {
            const int pr = cp / MAXC, c = cp % MAXC;
            const int a0 = ag * 16 + wv * 4 + 2 * pr;           // anchors a0, a0 + 1
            if (c == 0) {                                       // pair start
#pragma unroll
                for (int q = 0; q < 2; ++q) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[q][v] = 0.f;
                    r[q] = rkt[(a0 + q < NA ? a0 + q : NA - 1) * KS + jk];
                    r[q].w = kvalid ? r[q].w : -1e30f;
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int i = 2 * cp + q;
                ahead(tp, tpn, i);
                i32_gwait_ng<LA>(ngather(i), gr[i % NRING]);
            }
            {                                                   // (anchors 60..63 -- waves 3 and 7 of the last group -- recompute anchor 59: never stored,
                                                                // and these waves would only wait at the barrier otherwise)
                float (&f0)[4] = gr[(2 * cp) % NRING];
                float (&f1)[4] = gr[(2 * cp + 1) % NRING];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float4 g = mynbt[8 * c + 2 * s + kk];
                    const float w0 = fmaxf(0.f, fmaf(g.z, r[0].z, fmaf(g.y, r[0].y, fmaf(g.x, r[0].x, g.w + r[0].w))));
                    const float w1 = fmaxf(0.f, fmaf(g.z, r[1].z, fmaf(g.y, r[1].y, fmaf(g.x, r[1].x, g.w + r[1].w))));
#if I32_ABL & 1
                    asm volatile("" :: "v"(w0), "v"(w1), "v"(f0[s]), "v"(f1[s]));
#else
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, f0[s], acc[0], 0, 0, 0);      // D[k][c]
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, f1[s], acc[1], 0, 0, 0);
#endif
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
            I32_BARRIER();
            const float* xbase = &wbuf[j * I32_SLD + kp * KP + 4 * kk];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int i = NCS + g * NU + u;
                ahead(tp, tpn, i);
                const float4 bv = *reinterpret_cast<const float4*>(xbase + 8 * u);
                i32_wwait_ng<LA>(ngather(i), wr[i % NRING]);
                const f32x4 av = wr[i % NRING];
                f32x16& y = yacc[u & 1];
#if I32_ABL & 2
                asm volatile("" :: "v"(av), "v"(bv.x), "v"(bv.y), "v"(bv.z), "v"(bv.w)); (void)y;
#else
                y = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, y, 0, 0, 0);
                y = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, y, 0, 0, 0);
                y = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, y, 0, 0, 0);
                y = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, y, 0, 0, 0);
#endif
            }
            float* t_ = wbuf; wbuf = obuf; obuf = t_;
        }
        if (h < NTIL - 1) continue;
        // ---------------- the K-split partial tiles meet (in the buffer nobody reads any more: the slice before last);
        // y[v] = Y[o = 32 mt + 8 (v / 4) + 4 kk + v % 4][col = j]
        float* part = wbuf;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(&part[(kp * 32 + j) * PS + 32 * mt + 8 * g + 4 * kk]) =
                (f32x4){yacc[0][4 * g] + yacc[1][4 * g], yacc[0][4 * g + 1] + yacc[1][4 * g + 1], yacc[0][4 * g + 2] + yacc[1][4 * g + 2], yacc[0][4 * g + 3] + yacc[1][4 * g + 3]};
        I32_BARRIER();                                        // also: every wave finished reading the last slice (obuf)
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
                v += bo;
                out[(((size_t)b * p2 + pts[which]) * NA + a) * COUT + o] = v;
                st_s[which] += (double)v; st_q[which] += (double)v * (double)v;
            }
        }
        { float* t_ = wbuf; wbuf = obuf; obuf = t_; }            // the next slice goes to the last slice's buffer; the partial tiles are overwritten
                                                                // one barrier later, behind every wave's output loop
    }
    // drain the look-ahead of the last pass: every slot is an operand, so no slot's registers are reused while a load is in flight
#pragma unroll
    for (int sl = 0; sl < NRING; ++sl) { i32_gwait<0>(gr[sl]); i32_wwait<0>(wr[sl]); }
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

template <int CIN, int COUT, int MAXC, int NRING>
static int launch_inter32_t(int b, int p1, int p2, int nn, float sigma, const float* xyz, const float* new_xyz, const int* idx, const float* feats,
                            const float* rk, const float* Wp32, const float* bias, float* out, const int* order, double* stat_part, hipStream_t st) {
    const size_t lds = (size_t)(2 * I32_BUF(COUT) + 2 * 8 * MAXC * 5 + NA * KS * 4) * sizeof(float);
    auto kern = inter_so3conv32_kernel<CIN, COUT, MAXC, NRING>;
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

// ring slots: the largest of 8 / 7 / 5 / 4 that divides the groups of a tile pass (4 always does)
constexpr int i32_nring(int ng) { return ng % 8 == 0 ? 8 : ng % 7 == 0 ? 7 : ng % 5 == 0 ? 5 : 4; }
#ifndef I32_NRING
#define I32_NRING(NG) i32_nring(NG)
#endif
template <int CIN, int COUT>
static int launch_inter32(int b, int p1, int p2, int nn, float sigma, const float* xyz, const float* new_xyz, const int* idx, const float* feats,
                          const float* rk, const float* Wp32, const float* bias, float* out, const int* order, double* stat_part, hipStream_t st) {
    switch ((nn + 7) / 8) {
#define I32_CASE(MC) case MC: return launch_inter32_t<CIN, COUT, MC, I32_NRING(4 * MC + 3 * (32 / (8 / (COUT / 32))))>(b, p1, p2, nn, sigma, xyz, new_xyz, idx, feats, rk, Wp32, bias, out, order, stat_part, st);
        I32_CASE(1) I32_CASE(2) I32_CASE(3) I32_CASE(4) I32_CASE(5) I32_CASE(6) I32_CASE(7) I32_CASE(8)
#undef I32_CASE
    }
    return ETCH_EINVAL;
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
