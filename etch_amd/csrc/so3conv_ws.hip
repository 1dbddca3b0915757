// Intra-SO(3) convolution, weight-stationary on the bf16 matrix cores (replaces intra_so3conv_grouping + BasicSO3Conv,
// /root/reference/external/vgtk/vgtk/so3conv/functional.py:331-378, modules.py:150-153, like etch_intra_so3conv).
//
// Arithmetic: every fp32 operand is split EXACTLY into three bf16 values (8 + 8 + 8 mantissa bits, by truncation: x = hi + mid + lo) and a
// product is the fp32-accumulated sum of its six largest cross products on v_mfma_f32_32x32x16_bf16 -- the error against fp64 of the fp32 MFMA
// (profiles/r03_bf16x3_split.txt) at 2.3 x its rate, and beside the VALU instead of on its datapath.  That only pays where an operand is
// split once and used many times, which decides the layout:
//   W   (C x 12 C, static)           split on the host (ops.intra_weight_split), loaded ONCE per workgroup into registers: wave (mt, kq) keeps the
//                                    fragments of output-channel tile mt (32 rows) x the taps 3 kq .. 3 kq + 2 for the whole launch (72 / 144 VGPRs)
//                                    -> no weight stream from L2 at all (the 32x32x2 kernel re-read 197 KB of W per point pair);
//   X   (2 points x 60 anchors x C)  normalised + LeakyReLU'd + split when it is staged in LDS (three bf16 planes), read 12 x (once per tap).
// A workgroup is persistent: it walks point pairs (grid = CUs), the next pair's rows are in flight (registers) while the matrix cores work.
// Per pair and (point, anchor half): wave (mt, kq) forms the partial tile Y[32 mt .. + 31][32 anchors] over its three taps; the four K-parts meet
// in LDS (double-buffered: one barrier per phase), bias, output, fp64 InstanceNorm partial sums per pair as etch_intra_so3conv32.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define NA 60

// 4 consecutive fp32 values -> 3 planes x 4 bf16 (2 dwords each); exact: v = hi + mid + lo
__device__ __forceinline__ void ws_split3_pack4(const float4 v4, uint2& hi, uint2& mid, uint2& lo) {
    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = __float_as_uint(v[i]);
        const float r = v[i] - __uint_as_float(h[i] & 0xffff0000u);
        m[i] = __float_as_uint(r);
        l[i] = __float_as_uint(r - __uint_as_float(m[i] & 0xffff0000u));
    }
    // v_perm_b32: bytes 2, 3 of the even element below bytes 2, 3 of the odd one
    hi = make_uint2(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u));
    mid = make_uint2(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u));
    lo = make_uint2(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u));
}

#ifndef WS_ABL
#define WS_ABL 0       // timing experiments only (wrong results): 1 no MFMAs, 2 no LDS operand reads, 4 no output loop, 8 no staging of the next pair
#endif
template <int C>
struct WsShape {
    static constexpr int NT = C * 8;              // threads: 4 (C = 32) or 8 (C = 64) waves
    static constexpr int MT = C / 32;             // output-channel tiles
    static constexpr int KQ = 4;                  // K parts (3 taps each)
    static constexpr int NKS = 3 * C / 16;        // K steps of 16 per wave
    static constexpr int LDB = C + 8;             // bf16 row stride of a plane: LDB / 8 odd -> the 16-byte B reads of consecutive rows spread over the banks
    static constexpr int PS = C + 4;              // partial-tile row stride (floats)
    static constexpr int PLANE = 2 * NA * LDB;    // bf16 elements of one plane (2 points)
    static constexpr int NPRE = (2 * NA * (C / 4) + NT - 1) / NT;      // float4 loads per thread and pair
    static constexpr size_t lds_bytes = (size_t)3 * PLANE * 2 + (size_t)2 * KQ * 32 * PS * 4 + NA * 12 * 4 + (size_t)2 * NT * 8;
};

template <int C>
__global__ void __launch_bounds__(C * 8) intra_so3conv_ws_kernel(int npts_total, int pts_per_batch, const float* __restrict__ X,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const int* __restrict__ intra_idx, const bf16x8* __restrict__ Wq,
                                                                 const float* __restrict__ bias, float* __restrict__ Y,
                                                                 double* __restrict__ stat_part) {
    using S = WsShape<C>;
    constexpr int NT = S::NT, MT = S::MT, KQ = S::KQ, NKS = S::NKS, LDB = S::LDB, PS = S::PS, PLANE = S::PLANE, NPRE = S::NPRE;
    constexpr int SPT = C / 16;                   // K steps per tap
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned short* planes = reinterpret_cast<unsigned short*>(smem_raw);                    // [3][2 * 60][LDB]
    float* part = reinterpret_cast<float*>(planes + 3 * PLANE);                              // [2 buffers][KQ][32 cols][PS]
    int* iidx = reinterpret_cast<int*>(part + 2 * KQ * 32 * PS);                              // [60][12]
    double* dred = reinterpret_cast<double*>(iidx + NA * 12);                                 // [2][NT]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kk = lane >> 5;
    const int mt = wave % MT, kq = wave / MT;
    const int npairs = (npts_total + 1) >> 1;

    // ---- the wave's weight fragments, for the whole launch
    bf16x8 aq[NKS][3];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) aq[ks][pl] = Wq[((((size_t)mt * KQ + kq) * NKS + ks) * 3 + pl) * 64 + lane];
    for (int e = tid; e < NA * 12; e += NT) iidx[e] = intra_idx[e];
    const int o_out = tid % C;                    // the output channel this thread writes in every phase (NT % C == 0)
    const float bo = bias[o_out];
    __syncthreads();
    // source rows of this lane's anchors (both halves) for the wave's three taps
    int srow[2][3];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int a = 32 * half + j;
#pragma unroll
        for (int tl = 0; tl < 3; ++tl) srow[half][tl] = iidx[(a < NA ? a : 0) * 12 + 3 * kq + tl] * LDB;
    }

    // ---- staging: rows of a pair -> registers (raw) -> normalise, LeakyReLU, split -> three bf16 planes
    float4 pre[NPRE];
    auto prefetch = [&](int pp) {
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int e = tid + i * NT;
            const int row = e / (C / 4), c4 = e % (C / 4);
            const int pt = 2 * pp + row / NA;
            pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < 2 * NA * (C / 4) && pt < npts_total) pre[i] = *reinterpret_cast<const float4*>(X + ((size_t)2 * pp * NA + row) * C + c4 * 4);
        }
    };
    auto stage = [&](int pp) {
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int e = tid + i * NT;
            if (e >= 2 * NA * (C / 4)) continue;
            const int row = e / (C / 4), c4 = e % (C / 4);
            const int pt = 2 * pp + row / NA;
            float4 v = pre[i];
            if (mean && pt < npts_total) {
                const int bb = pt / pts_per_batch;
                const float4 m = *reinterpret_cast<const float4*>(mean + (size_t)bb * C + c4 * 4);
                const float4 r = *reinterpret_cast<const float4*>(rstd + (size_t)bb * C + c4 * 4);
                v.x = (v.x - m.x) * r.x; v.y = (v.y - m.y) * r.y; v.z = (v.z - m.z) * r.z; v.w = (v.w - m.w) * r.w;
                v.x = v.x > 0.f ? v.x : 0.01f * v.x; v.y = v.y > 0.f ? v.y : 0.01f * v.y;
                v.z = v.z > 0.f ? v.z : 0.01f * v.z; v.w = v.w > 0.f ? v.w : 0.01f * v.w;
            }
            uint2 ph, pm, pl;
            ws_split3_pack4(v, ph, pm, pl);
            unsigned short* dst = planes + row * LDB + c4 * 4;
            *reinterpret_cast<uint2*>(dst) = ph; *reinterpret_cast<uint2*>(dst + PLANE) = pm; *reinterpret_cast<uint2*>(dst + 2 * PLANE) = pl;
        }
    };

    int pp = blockIdx.x;
    if (pp < npairs) prefetch(pp);
    if (pp < npairs) stage(pp);
    __syncthreads();
#pragma unroll 1
    while (pp < npairs) {
        const int next = pp + gridDim.x;
        if (next < npairs) prefetch(next);        // in flight during the four phases below
        double st_s = 0.0, st_q = 0.0;            // InstanceNorm partial sums of this pair, channel o_out
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int pt = ph >> 1, half = ph & 1;
            f32x16 acc;
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
            for (int tl = 0; tl < 3; ++tl) {
                const unsigned short* xrow = planes + pt * NA * LDB + srow[half][tl] + 8 * kk;
#pragma unroll
                for (int s = 0; s < SPT; ++s) {
                    const int ks = tl * SPT + s;
                    bf16x8 bq[3];
#if WS_ABL & 2
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) bq[pl] = aq[ks][pl];
#else
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) bq[pl] = *reinterpret_cast<const bf16x8*>(xrow + pl * PLANE + 16 * s);
#endif
#if WS_ABL & 1
                    asm volatile("" :: "v"(bq[0]), "v"(bq[1]), "v"(bq[2]));
#else
                    // smallest cross products first
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks][2], bq[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks][0], bq[2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks][1], bq[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks][1], bq[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks][0], bq[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks][0], bq[0], acc, 0, 0, 0);
#endif
                }
            }
            // acc[v] = partial Y[o = 32 mt + 8 (v / 4) + 4 kk + v % 4][anchor 32 half + j]
            float* pb = part + (ph & 1) * KQ * 32 * PS;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(&pb[(kq * 32 + j) * PS + 32 * mt + 8 * g + 4 * kk]) = (f32x4){acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            __syncthreads();
            const int ptg = 2 * pp + pt;
            if (ptg < npts_total && !(WS_ABL & 4)) {
#pragma unroll
                for (int it = 0; it < 32 * C / NT; ++it) {
                    const int col = (tid + it * NT) / C;
                    const int a = 32 * half + col;
                    if (a < NA) {
                        float v = pb[col * PS + o_out];
#pragma unroll
                        for (int q = 1; q < KQ; ++q) v += pb[(q * 32 + col) * PS + o_out];
                        v += bo;
                        Y[((size_t)ptg * NA + a) * C + o_out] = v;
                        st_s += (double)v; st_q += (double)v * (double)v;
                    }
                }
            }
            // (the buffer this phase read is rewritten two phases on, behind the next phase's barrier)
        }
        // the planes are free (every wave passed the last phase's barrier behind its MFMAs): the next pair goes in
        if (stat_part) { dred[tid] = st_s; dred[NT + tid] = st_q; }
        if (next < npairs && !(WS_ABL & 8)) stage(next);
        __syncthreads();
        if (stat_part && tid < C) {
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int k = 0; k < NT / C; ++k) { a0 += dred[k * C + tid]; a1 += dred[NT + k * C + tid]; }
            stat_part[(size_t)pp * 2 * C + tid] = a0; stat_part[(size_t)pp * 2 * C + C + tid] = a1;
        }
        pp = next;
    }
}

template <int C>
static int launch_intra_ws(int npts, int ppb, const float* X, const float* mean, const float* rstd, const int* intra_idx, const void* Wq,
                           const float* bias, float* Y, double* stat_part, hipStream_t st) {
    using S = WsShape<C>;
    if (stat_part && (ppb % 2) != 0) return ETCH_EUNSUPPORTED;          // a pair's points must belong to one sample
    auto kern = intra_so3conv_ws_kernel<C>;
    static bool ready = false;
    if (!ready) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::lds_bytes);
        if (e != hipSuccess) return (int)e;
        ready = true;
    }
    const int npairs = (npts + 1) / 2;
    const int wgs_per_cu = C == 32 ? 2 : 1;                              // C = 32: 73 KB of LDS and 4 waves per workgroup
    int grid = etch_cu_count() * wgs_per_cu;
    if (grid > npairs) grid = npairs;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(S::NT), S::lds_bytes, st, npts, ppb, X, mean, rstd, intra_idx, reinterpret_cast<const bf16x8*>(Wq), bias, Y,
                       stat_part);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

// Wq = ops.intra_weight_split: [mt][kq][K step][plane hi / mid / lo][lane][8 bf16],
//   [lane][e] = plane of W2[32 mt + lane % 32][(3 kq) C + 16 ks + 8 (lane / 32) + e],  W2[o][tap * C + ch] = W[o][ch * 12 + tap].
// stat_part (b * p/2, 2, cout) fp64 or NULL as etch_intra_so3conv_stats (p even).
extern "C" int etch_intra_so3conv_split(int b, int c, int cout, int p, const float* X, const float* mean, const float* rstd, const int* intra_idx,
                                        const void* Wq, const float* bias, float* Y, double* stat_part, void* stream) {
    if (b <= 0 || p <= 0) return ETCH_OK;
    if (c != cout || !Wq) return ETCH_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (c == 32) return launch_intra_ws<32>(b * p, p, X, mean, rstd, intra_idx, Wq, bias, Y, stat_part, st);
    if (c == 64) return launch_intra_ws<64>(b * p, p, X, mean, rstd, intra_idx, Wq, bias, Y, stat_part, st);
    return ETCH_EUNSUPPORTED;
}
