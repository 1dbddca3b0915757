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
// Round 5, template flag F16 (C ABI etch_intra_so3conv_f16): the same kernel on v_mfma_f32_32x32x16_f16 with TWO fp16 planes per operand (split_bf16.h:
// h = fp16(x), l = fp16(x - h), both to nearest; three cross terms) -- the operands here are at unit scale (InstanceNorm + LeakyReLU outputs; W arrives
// as planes of its rows times their own powers of two, the epilogue multiplies by the inverse power), where that split carries the fp32 MFMA's error (profiles/r05_f16_two_plane_split.txt):
// half the matrix instructions, two thirds of the LDS plane traffic, 48 / 96 instead of 72 / 144 registers of weight fragments.
#include "common.h"
#include "split_bf16.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
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

template <int C, bool F16 = false>
struct WsShape {
    static constexpr int NPL = F16 ? 2 : 3;       // operand planes
    static constexpr int NT = C * 8;              // threads: 4 (C = 32) or 8 (C = 64) waves
    static constexpr int MT = C / 32;             // output-channel tiles
    static constexpr int KQ = 4;                  // K parts (3 taps each)
    static constexpr int NKS = 3 * C / 16;        // K steps of 16 per wave
    static constexpr int LDB = C + 8;             // bf16 row stride of a plane: LDB / 8 odd -> the 16-byte B reads of consecutive rows spread over the banks
    static constexpr int PS = C + 4;              // partial-tile row stride (floats)
    static constexpr int PLANE = NA * LDB;        // bf16 elements of one plane of one point
    static constexpr int NPRE = (NA * (C / 4) + NT - 1) / NT;          // float4 loads per thread and point
    // [2 points][3 planes] + [2 buffers][KQ][32 cols][PS] partial tiles + anchor table + statistics staging
    static constexpr size_t lds_bytes = (size_t)2 * NPL * PLANE * 2 + (size_t)2 * KQ * 32 * PS * 4 + NA * 12 * 4 + (size_t)2 * NT * 8;
};

// The kernel is a pipeline of PHASES (point, anchor half).  In phase (pt, half) a wave
//   * multiplies: its 3 taps x the point's planes -> partial tile, written to part[half] at the end of the phase (one barrier per phase);
//   * and, in the same straight-line block -- independent work the scheduler interleaves with the (dependent) MFMA chain, and which the bf16
//     matrix cores do not compete with --
//       - sums and stores the PREVIOUS phase's partial tiles (bias, output, statistics),
//       - half 0: normalises / splits the next point's rows (registers) into the other plane buffer,
//       - half 1: requests the rows of the point after that.
// Nothing in a phase is conditional (rows and stores of points past the end go through zero-sized buffer resources: loads return 0, stores
// are dropped; anchors 60..63 of a tile fall outside the point's 60 x C records), so a phase is ONE basic block.
#define SROW_REGS_C(C) ((C) <= 32)
template <int C, bool NORM, bool STATS, bool F16 = false>     // NORM: mean / rstd given (InstanceNorm + LeakyReLU on load); STATS: stat_part given
__global__ void __launch_bounds__(C * 8) intra_so3conv_ws_kernel(int npts_total, int pts_per_batch, const float* __restrict__ X,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const int* __restrict__ intra_idx, const bf16x8* __restrict__ Wq,
                                                                 const float* __restrict__ bias, float* __restrict__ Y,
                                                                 double* __restrict__ stat_part, unsigned* __restrict__ ctr, const float* __restrict__ wsc) {
    __shared__ unsigned s_grab;
    using S = WsShape<C, F16>;
    constexpr int NT = S::NT, MT = S::MT, KQ = S::KQ, NKS = S::NKS, LDB = S::LDB, PS = S::PS, PLANE = S::PLANE, NPRE = S::NPRE, NPL = S::NPL;
    constexpr int SPT = C / 16;                   // K steps per tap
    constexpr int PBYTES = NA * C * 4;            // bytes of one point's rows
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned short* planes = reinterpret_cast<unsigned short*>(smem_raw);                    // [2 points][NPL][60][LDB]
    float* part = reinterpret_cast<float*>(planes + 2 * NPL * PLANE);                          // [2 buffers][KQ][32 cols][PS]
    int* iidx = reinterpret_cast<int*>(part + 2 * KQ * 32 * PS);                              // [60][12]
    double* dred = reinterpret_cast<double*>(iidx + NA * 12);                                 // [2][NT]
    int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kk = lane >> 5;
    const int mt = wave % MT, kq = wave / MT;
    const int npairs = (npts_total + 1) >> 1;

    // ---- the wave's weight fragments, for the whole launch
    bf16x8 aq[NKS][NPL];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) aq[ks][pl] = Wq[((((size_t)mt * KQ + kq) * NKS + ks) * NPL + pl) * 64 + lane];
    for (int e = tid; e < NA * 12; e += NT) iidx[e] = intra_idx[e];
    int o_out = tid % C;                          // the output channel this thread writes in every phase (NT % C == 0)
    const float bo = bias[o_out];
    const float oscale = F16 ? wsc[o_out] : 1.0f;         // F16: the weight planes carry this output channel's row times its own power of two (exact)
    __syncthreads();
    // source rows of this lane's anchors (both halves) for the wave's three taps
    // (C = 64: 144 registers of weight fragments leave no room for the six offsets -- 8 spilled registers -- so they are re-read from the LDS
    // table per phase: three ds_read_b32 beside 72 MFMAs)
    constexpr bool SROW_REGS = SROW_REGS_C(C);
    int srow[SROW_REGS ? 2 : 1][3];
    if constexpr (SROW_REGS) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int a = 32 * half + j;
#pragma unroll
            for (int tl = 0; tl < 3; ++tl) srow[half][tl] = iidx[(a < NA ? a : 0) * 12 + 3 * kq + tl] * LDB;
        }
    }
    const int* irow_lo = iidx + j * 12 + 3 * kq;                          // anchor j
    const int* irow_hi = iidx + (32 + j < NA ? 32 + j : 0) * 12 + 3 * kq;  // anchor 32 + j (60..63: anchor 0, results dropped)

    // ---- staging: rows of a point -> registers (raw) -> normalise, LeakyReLU, split -> three bf16 planes
    float4 pre[NPRE];
    int pre_pt = 0;                               // the point `pre` holds (for its sample's statistics)
    auto prefetch = [&](int pt) {
        pre_pt = pt;
        const bool ok = pt < npts_total;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X) + (size_t)(ok ? pt : 0) * (NA * C), 0, ok ? PBYTES : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < NPRE; ++i) pre[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + i * NT) * 16, 0, 0));     // past the point's rows: 0
    };
    auto stage = [&](unsigned short* P) {
        const unsigned bb = (unsigned)(pre_pt < npts_total ? pre_pt : 0) / (unsigned)pts_per_batch;      // (32-bit: a 64-bit division brings branches)
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int e = tid + i * NT;
            const int row = e / (C / 4), c4 = e % (C / 4);
            float4 v = pre[i];
            if (NORM) {
                const float4 m = *reinterpret_cast<const float4*>(mean + (size_t)bb * C + c4 * 4);
                const float4 r = *reinterpret_cast<const float4*>(rstd + (size_t)bb * C + c4 * 4);
                v.x = (v.x - m.x) * r.x; v.y = (v.y - m.y) * r.y; v.z = (v.z - m.z) * r.z; v.w = (v.w - m.w) * r.w;
                v.x = v.x > 0.f ? v.x : 0.01f * v.x; v.y = v.y > 0.f ? v.y : 0.01f * v.y;
                v.z = v.z > 0.f ? v.z : 0.01f * v.z; v.w = v.w > 0.f ? v.w : 0.01f * v.w;
            }
            // (threads past the 60 rows write into the 8-element row padding region of row 59 .. never read: clamp instead of a branch)
            unsigned short* dst = P + (e < NA * (C / 4) ? row * LDB + c4 * 4 : (NA - 1) * LDB + C);
            if constexpr (F16) {
                uint2 ph, pl;
                split2h_pack4(v, ph, pl);
                *reinterpret_cast<uint2*>(dst) = ph; *reinterpret_cast<uint2*>(dst + PLANE) = pl;
            } else {
                uint2 ph, pm, pl;
                ws_split3_pack4(v, ph, pm, pl);
                *reinterpret_cast<uint2*>(dst) = ph; *reinterpret_cast<uint2*>(dst + PLANE) = pm; *reinterpret_cast<uint2*>(dst + 2 * PLANE) = pl;
            }
        }
    };
    // ---- one phase's matrix work: planes P of the point, anchor half -> partial tile into pb
    auto multiply = [&](const unsigned short* P, int half, float* pb) {
        f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
        for (int tl = 0; tl < 3; ++tl) {
            const unsigned short* xrow = P + (SROW_REGS ? srow[SROW_REGS ? half : 0][tl] : (half ? irow_hi : irow_lo)[tl] * LDB) + 8 * kk;
#pragma unroll
            for (int s_ = 0; s_ < SPT; ++s_) {
                const int ks = tl * SPT + s_;
                bf16x8 bq[NPL];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) bq[pl] = *reinterpret_cast<const bf16x8*>(xrow + pl * PLANE + 16 * s_);
                // smallest cross products first
                if constexpr (F16) {
#define WS_H(a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, aq[ks][a]), __builtin_bit_cast(f16x8, bq[b]), acc, 0, 0, 0)
                    WS_H(1, 0); WS_H(0, 1); WS_H(0, 0);
#undef WS_H
                    continue;
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks][2 % NPL], bq[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks][0], bq[2 % NPL], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks][1], bq[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks][1], bq[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks][0], bq[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks][0], bq[0], acc, 0, 0, 0);
            }
        }
        // acc[v] = partial Y[o = 32 mt + 8 (v / 4) + 4 kk + v % 4][anchor 32 half + j]
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(&pb[(kq * 32 + j) * PS + 32 * mt + 8 * g + 4 * kk]) = (f32x4){acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    };
    // ---- the previous phase's partial tiles -> bias, output (point ptg, anchor half), statistics
    double st_s = 0.0, st_q = 0.0;                // InstanceNorm partial sums of the pair being written, channel o_out
    auto emit = [&](const float* pb, int ptg, int half) {
        const bool ok = ptg >= 0 && ptg < npts_total;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Y + (size_t)(ok ? ptg : 0) * (NA * C), 0, ok ? PBYTES : 0, 0x00020000);
#pragma unroll
        for (int it = 0; it < 32 * C / NT; ++it) {
            const int col = (tid + it * NT) / C;
            const int a = 32 * half + col;
            float v = pb[col * PS + o_out];
#pragma unroll
            for (int q = 1; q < KQ; ++q) v += pb[(q * 32 + col) * PS + o_out];
            v = v * oscale + bo;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (a * C + o_out) * 4, 0, 0);      // anchors 60..63 / points past the end: dropped
            const double dv = (ok && a < NA) ? (double)v : 0.0;
            st_s += dv; st_q += dv * dv;
        }
    };
    auto finish_stats = [&](int pp) {               // after the barrier that follows the dred writes
        if (STATS && pp >= 0 && tid < C) {
            int t2 = tid;
            asm volatile("" : "+v"(t2));            // the store address is formed here, not hoisted out of the point loop as a 64-bit register pair (spilled at C = 64)
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int k = 0; k < NT / C; ++k) { a0 += dred[k * C + t2]; a1 += dred[NT + k * C + t2]; }
            stat_part[(size_t)pp * 2 * C + t2] = a0; stat_part[(size_t)pp * 2 * C + C + t2] = a1;
        }
    };

    // the matrix-core chain of a phase is dependent (one accumulator): between two MFMAs the wave has ~32 idle cycles -- the scheduler is told
    // to put the phase's other instructions there (one LDS access, a few VALU ops, now and then a global access per MFMA)
    auto interleave = [&]() {
#pragma unroll
        for (int i = 0; i < 6 * NKS; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);      // 1 LDS access
            __builtin_amdgcn_sched_group_barrier(0x002, C == 32 ? 5 : 3, 0);      // VALU
            if (i % 4 == 3) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    // 1 global access
        }
    };
    unsigned short* P0 = planes;
    unsigned short* P1 = planes + NPL * PLANE;
    float* part0 = part;
    float* part1 = part + KQ * 32 * PS;
    int pp = blockIdx.x;
    if (pp >= npairs) return;
    prefetch(2 * pp); stage(P0);
    prefetch(2 * pp + 1);
    __syncthreads();
    int prev_pp = -1;                             // the pair whose last phase is still to be written (-1: none -- its stores are dropped)
    // pairs: the first two of a workgroup are static, every further one comes from the launch's work counter (common.h: etch_work_counter_slot),
    // asked for by thread 0 at the top of a pair and handed over through LDS behind the pair's last barrier
    int nxt = pp + (int)gridDim.x;
#pragma unroll 1
    for (; pp < npairs;) {
        const int next = nxt < npairs ? nxt : npairs;      // (past the end: zero-sized resources)

        // phase (point 0, half 0): previous pair's last phase goes out; point 1 of this pair is staged
        emit(part1, 2 * prev_pp + 1, 1);
        if (STATS) { dred[tid] = st_s; dred[NT + tid] = st_q; }
        st_s = 0.0; st_q = 0.0;
        stage(P1);
        multiply(P0, 0, part0);
        interleave();
        __syncthreads();
        // phase (0, 1)
        finish_stats(prev_pp);
        emit(part0, 2 * pp, 0);
        prefetch(2 * next);
        multiply(P0, 1, part1);
        interleave();
        __syncthreads();
        // phase (1, 0): the next pair's point 0 is staged (its rows were requested a phase ago)
        emit(part1, 2 * pp, 1);
        stage(P0);
        multiply(P1, 0, part0);
        interleave();
        __syncthreads();
        // phase (1, 1)  (+ the request for the pair after next: its answer has this phase's 72 MFMAs to arrive)
        unsigned grabbed = 0u;
        if (ctr && tid == 0) grabbed = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        emit(part0, 2 * pp + 1, 0);
        prefetch(2 * next + 1);
        multiply(P1, 1, part1);
        interleave();
        if (ctr && tid == 0) s_grab = grabbed;
        __syncthreads();
        prev_pp = pp;
        pp = next;
        nxt = ctr ? 2 * (int)gridDim.x + __builtin_amdgcn_readfirstlane((int)s_grab) : nxt + (int)gridDim.x;      // wave-uniform: scalar registers
    }
    // drain: the last phase's tiles, the last pair's statistics.  (The laundering keeps the drain's address arithmetic out of the registers the
    // loop needs: values derived from tid / o_out for use down here were spilled across the whole loop at C = 64.)
    asm volatile("" : "+v"(tid), "+v"(o_out));
    emit(part1, 2 * prev_pp + 1, 1);
    if (STATS) { dred[tid] = st_s; dred[NT + tid] = st_q; }
    __syncthreads();
    finish_stats(prev_pp);
}

template <int C, bool NORM, bool STATS, bool F16>
static int launch_intra_ws_t(int npts, int ppb, const float* X, const float* mean, const float* rstd, const int* intra_idx, const void* Wq,
                             const float* bias, float* Y, double* stat_part, hipStream_t st, const float* wsc) {
    using S = WsShape<C, F16>;
    auto kern = intra_so3conv_ws_kernel<C, NORM, STATS, F16>;
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
                       stat_part, etch_work_counter_slot(st), wsc);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}
template <int C, bool F16>
static int launch_intra_ws(int npts, int ppb, const float* X, const float* mean, const float* rstd, const int* intra_idx, const void* Wq,
                           const float* bias, float* Y, double* stat_part, hipStream_t st, const float* wsc = nullptr) {
    if (stat_part && (ppb % 2) != 0) return ETCH_EUNSUPPORTED;          // a pair's points must belong to one sample
    if ((mean == nullptr) != (rstd == nullptr)) return ETCH_EINVAL;
#define WS_GO(N, S_) return launch_intra_ws_t<C, N, S_, F16>(npts, ppb, X, mean, rstd, intra_idx, Wq, bias, Y, stat_part, st, wsc)
    if (mean) { if (stat_part) WS_GO(true, true); WS_GO(true, false); }
    if (stat_part) WS_GO(false, true);
    WS_GO(false, false);
#undef WS_GO
}

// Wq = ops.intra_weight_split: [mt][kq][K step][plane hi / mid / lo][lane][8 bf16],
//   [lane][e] = plane of W2[32 mt + lane % 32][(3 kq) C + 16 ks + 8 (lane / 32) + e],  W2[o][tap * C + ch] = W[o][ch * 12 + tap].
// stat_part (b * p/2, 2, cout) fp64 or NULL as etch_intra_so3conv_stats (p even).
extern "C" int etch_intra_so3conv_split(int b, int c, int cout, int p, const float* X, const float* mean, const float* rstd, const int* intra_idx,
                                        const void* Wq, const float* bias, float* Y, double* stat_part, void* stream) {
    if (b <= 0 || p <= 0) return ETCH_OK;
    if (c != cout || !Wq) return ETCH_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (c == 32) return launch_intra_ws<32, false>(b * p, p, X, mean, rstd, intra_idx, Wq, bias, Y, stat_part, st);
    if (c == 64) return launch_intra_ws<64, false>(b * p, p, X, mean, rstd, intra_idx, Wq, bias, Y, stat_part, st);
    return ETCH_EUNSUPPORTED;
}

// The same on the fp16 matrix cores with two planes per operand.  Wqh = ops.intra_weight_split_f16: the layout above with two fp16 planes of W2, every
// row (output channel) times its own power of two; wsc (cout floats) = the inverse powers.
extern "C" int etch_intra_so3conv_f16(int b, int c, int cout, int p, const float* X, const float* mean, const float* rstd, const int* intra_idx,
                                      const void* Wqh, const float* wsc, const float* bias, float* Y, double* stat_part, void* stream) {
    if (b <= 0 || p <= 0) return ETCH_OK;
    if (c != cout || !Wqh || !wsc) return ETCH_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (c == 32) return launch_intra_ws<32, true>(b * p, p, X, mean, rstd, intra_idx, Wqh, bias, Y, stat_part, st, wsc);
    if (c == 64) return launch_intra_ws<64, true>(b * p, p, X, mean, rstd, intra_idx, Wqh, bias, Y, stat_part, st, wsc);
    return ETCH_EUNSUPPORTED;
}
