// Inter SO(3) conv, round 5: the kernel weights come off the MATRIX pipe.
//
//   etch_inter_so3conv_planes_kq   replaces inter_so3conv_grouping_anchor + inter_so3conv_feat_grouping + BasicSO3Conv
//                                  (/root/reference/external/vgtk/vgtk/so3conv/functional.py:286-324, :61-67, modules.py:33-39)
//   etch_inter_kpoint_operand      rotated kernel points (functional.py:296) -> the B operand of the pre-activation product
//
// inter_so3conv_x32_kernel (so3conv_x.hip) generates every kernel weight w[a, k, n] = relu(1 - |g_n - R_a kappa_k|^2 / sigma) on the VALU: four
// packed fp32 instructions per weight pair behind two LDS table reads, as a dependent chain per pair, laid between the matrix instructions in
// program order -- profiles/r04_inter_x_counters_and_ablations.txt: with the gathers, step 1's MFMAs and step 2 compiled out the kernels still take
// 2.2 / 1.5 / 2.2 ms, i.e. ~1 800 cycles per (anchor, chunk) step for ~150 instructions; packed fp32 instructions never hide behind an MFMA
// (profiles/r04_mfma_bf16_issue_rates.txt).  Here the pre-activation
//         P[n, k] = a_n + b_k + G_n . r_ak,    a_n = 1 - |g_n|^2 / sigma,  G_n = 2 g_n / sigma,  b_k = -|r_ak|^2 / sigma,  r_ak = R_a kappa_k
// is a rank-5 bilinear form [a_n, 1, G_n] . [1, b_k, r_ak]; both factors are split exactly into three bf16 planes and the six largest cross terms
// of the five products fill 30 of the 32 K slots of ONE pair of v_mfma_f32_32x32x16_bf16 per (anchor, 32-neighbour chunk): rows = neighbours,
// columns = kernel points, so a lane's 16 accumulators are one kernel point x 16 neighbours -- exactly the B operand of step 1.  What is left on
// the VALU per weight is the clamp and the exact split (v_med3, 2 and, 2 sub, 1.5 perm: plain instructions that hide beside MFMAs), written as
// 16 independent chains per lane and interleaved with the step's matrix instructions by sched_group_barrier.
//   * neighbour factor: built once per output point into an LDS table [nn][32 slots] bf16 and held in registers (8 VGPRs per chunk);
//   * kernel-point factor: a table of the layer, [60 anchors][2][64 lanes][8] bf16 (122 880 bytes, L2-resident), two 16-byte loads per anchor;
//   * the neighbour order of the product's rows is the staging tile's row order with bits 2 and 3 swapped, which makes the accumulator layout
//     (rows 8 (v / 4) + 4 (lane / 32) + v % 4) land on the K slots 8 (lane / 32) + e of the two K steps of step 1.
// Step 1 (transposed product on the gathered bf16 planes), the X1 tile, step 2 and the epilogue are those of inter_so3conv_x32_kernel.
// No inline-asm LDS reads are left in the loop (see etch_amd/isa_lint.py for why that matters); the W stream of step 2 keeps its asm ring.
#include "so3conv_x.h"

#ifndef INTER_Y_WPE
#define INTER_Y_WPE(CIN) ((CIN) <= 32 ? 2 : 1)
#endif
#define Y_PAD(CIN) ((CIN) <= 32 ? 12 : 44)
typedef unsigned y_u32x4 __attribute__((ext_vector_type(4)));

// K slot s of the pre-activation product: term t = s / 5 (s < 30), component c = s % 5.  Planes (0 hi, 1 mid, 2 lo) of the two factors per term:
//   term            0        1         2         3        4        5
//   neighbour side  hi       hi        mid       hi       lo       mid
//   kernel side     hi       mid       hi        lo       hi       mid
__host__ __device__ constexpr int y_geo_plane(int t) { return t == 2 || t == 5 ? 1 : (t == 4 ? 2 : 0); }
__host__ __device__ constexpr int y_kp_plane(int t) { return t == 1 || t == 5 ? 1 : (t == 3 ? 2 : 0); }

struct Y3 { unsigned h, m, l; };          // bf16 bit patterns (in the low 16 bits) of the exact split of one fp32 value
__device__ __forceinline__ Y3 y_split3(float v) {
    const unsigned h = __float_as_uint(v);
    const float r = v - __uint_as_float(h & 0xffff0000u);
    const unsigned m = __float_as_uint(r);
    const float q = r - __uint_as_float(m & 0xffff0000u);
    return {h >> 16, m >> 16, __float_as_uint(q) >> 16};
}
__device__ __forceinline__ constexpr unsigned y_pick(const Y3& s, int plane) { return plane == 0 ? s.h : (plane == 1 ? s.m : s.l); }

// [60][24][3] rotated kernel points -> kq [60][2 K steps][64 lanes][8] bf16: lane = 32 kg + k holds the slots 16 j + 8 kg + i of kernel point k
// (components [1, b_k, r_x, r_y, r_z]; k >= 24: b = -1e30, r = 0: weight 0)
__global__ void __launch_bounds__(128) inter_kpoint_operand_kernel(float inv_sigma, const float* __restrict__ rk, unsigned short* __restrict__ kq) {
    const int a = blockIdx.x, j = threadIdx.x >> 6, lane = threadIdx.x & 63, k = lane & 31, kg = lane >> 5;
    float comp[5] = {1.0f, -1e30f, 0.f, 0.f, 0.f};
    if (k < KS) {
        const float* r = rk + ((size_t)a * KS + k) * 3;
        comp[1] = -(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * inv_sigma;
        comp[2] = r[0]; comp[3] = r[1]; comp[4] = r[2];
    }
    const Y3 sp[5] = {y_split3(comp[0]), y_split3(comp[1]), y_split3(comp[2]), y_split3(comp[3]), y_split3(comp[4])};
    unsigned short* dst = kq + (((size_t)a * 2 + j) * 64 + lane) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int s = 16 * j + 8 * kg + i;
        dst[i] = (unsigned short)(s < 30 ? y_pick(sp[s % 5], y_kp_plane(s / 5)) : 0u);
    }
}

// step 2 of the 32x32x16 kernels on the fp16 matrix cores: Y[o][a] += sum_kappa W[o][kappa] X1[a][kappa] with both operands as two fp16 planes (W pre-split
// with every output channel's row times its own power of two, ops.inter_weight_split32_f16 -- undone per channel in the epilogue; the X1 row split by the wave that reads it) and the three largest cross products.
// W fragments: [K step of 16][o tile of 32][plane][lane][8], streamed from L2 one batch ahead (inline-asm loads + counted waits, see X32Step2).
// Round 6: the look-ahead ring holds RD batches.  With ONE batch ahead (round 5) a batch's loads were requested 3 MFMAs (~100 cycles) before their wait --
// an L2 round trip is several times that, and the 64-channel kernel (one wave per SIMD: nothing else to run meanwhile) spent step 2 waiting for W.
template <int CIN, int COUT, int PAD, int RD>
struct H32Step2 {
    static constexpr int MT2 = COUT / 32, CH = CIN / 2, KH = CH * KS, S = KH + PAD;
    static constexpr int NSW = KH / 16 / 4;         // K steps per wave and half
    static constexpr int NB = NSW * MT2;            // batches (one K step x one o tile: two planes) per half
    static_assert(RD >= 2 && RD - 1 <= 2 * NB, "ring depth");
    f32x4 ra[RD][2];
    __device__ __forceinline__ void issue(int i, const bf16x8* __restrict__ Wq, int wave, int lane) {
        const int h = i / NB, c = (i % NB) / MT2, mt = i % MT2;
#ifdef Y_ABL_WCACHED
        const char* b0 = reinterpret_cast<const char*>(Wq) + (size_t)(i & 1) * 2 * 1024;      // timing experiment: every W fragment is one of two cached ones
#else
        const char* b0 = reinterpret_cast<const char*>(Wq) + ((size_t)(h * (KH / 16) + wave + 4 * c) * MT2 + mt) * 2 * 1024;
#endif
        const unsigned vo = (unsigned)lane * 16u;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) x_wload_s(ra[i % RD][pl], vo, b0, pl);
    }
    // the first RD - 1 batches (before the barrier in front of the first half)
    __device__ __forceinline__ void prime(const bf16x8* __restrict__ Wq, int wave, int lane) {
#pragma unroll
        for (int i = 0; i < RD - 1; ++i) issue(i, Wq, wave, lane);
    }
    template <int N> __device__ __forceinline__ void wait(f32x4 (&v)[2]) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(v[0]), "+v"(v[1]) : "n"(N)); }
    // PF (the 64-channel kernel: one wave per SIMD, registers to spare): the X1 row of K step c + 1 is read while step c computes -- read where it is
    // split, every K step waited out an LDS round trip with nothing else to issue
#ifndef Y_S2_PF
#define Y_S2_PF(CIN, COUT) ((CIN) >= 64)
#endif
    static constexpr bool PF = Y_S2_PF(CIN, COUT);
    template <int H>
    __device__ __forceinline__ void half(f32x16 (&y)[MT2], const float* X1s, const bf16x8* __restrict__ Wq, int wave, int lane) {
        const int an = lane & 31, kg = lane >> 5;
        f16x8 bq[2];
        float4 xa = make_float4(0.f, 0.f, 0.f, 0.f), xb = xa;
        if (PF) {
            const float* xr = &X1s[an * S + wave * 16 + kg * 8];
            xa = *reinterpret_cast<const float4*>(xr); xb = *reinterpret_cast<const float4*>(xr + 4);
        }
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            const int i = H * NB + r, c = r / MT2, mt = r % MT2;
            if (mt == 0) {
                asm volatile("" ::: "memory");             // keeps the X1 reads (and their splits) of later steps from being hoisted
                if (PF) {
                    split2h_pack8(xa, xb, bq[0], bq[1]);
                    if (c + 1 < NSW) {
                        const float* xr = &X1s[an * S + (wave + 4 * (c + 1)) * 16 + kg * 8];
                        xa = *reinterpret_cast<const float4*>(xr); xb = *reinterpret_cast<const float4*>(xr + 4);
                    }
                } else {
                    const float* xr = &X1s[an * S + (wave + 4 * c) * 16 + kg * 8];
                    split2h_pack8(*reinterpret_cast<const float4*>(xr), *reinterpret_cast<const float4*>(xr + 4), bq[0], bq[1]);
                }
            }
            // batch i + RD - 1 goes into the entry batch i - 1 left; then all but the younger batches' loads must have landed (two loads per batch, in order)
            if (i + RD - 1 < 2 * NB) issue(i + RD - 1, Wq, wave, lane);
            constexpr int TOT = 2 * NB;
            const int younger = (i + RD - 1 < TOT ? RD - 1 : TOT - 1 - i);
            static_assert(RD <= 8, "wait table");
            if (younger >= 7) wait<14>(ra[i % RD]);
            else if (younger == 6) wait<12>(ra[i % RD]);
            else if (younger == 5) wait<10>(ra[i % RD]);
            else if (younger == 4) wait<8>(ra[i % RD]);
            else if (younger == 3) wait<6>(ra[i % RD]);
            else if (younger == 2) wait<4>(ra[i % RD]);
            else if (younger == 1) wait<2>(ra[i % RD]);
            else wait<0>(ra[i % RD]);
            f32x4 (&ac)[2] = ra[i % RD];
            // smallest cross products first: l * h, h * l, h * h
            y[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ac[1]), bq[0], y[mt], 0, 0, 0);
            y[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ac[0]), bq[1], y[mt], 0, 0, 0);
            y[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ac[0]), bq[0], y[mt], 0, 0, 0);
        }
    }
};

#ifndef Y_S2_DEPTH
#define Y_S2_DEPTH(CIN) ((CIN) <= 32 ? 2 : 6)
#endif

// Round 6: PERSISTENT workgroups.  profiles/r05_inter_conv_latency_bound.txt priced 8.3 of the 22 us a workgroup spends on a point as
// per-point fixed cost: workgroup launch, ball_idx -> row offsets -> barrier -> first gathers / coordinates -> neighbour factor -> barrier, pipeline fill.
// Now a workgroup walks a list of points and its software pipeline never drains between them:
//   * the point after the current one is known a point ahead (its index in the processing order is resolved during pass 1 of the previous point);
//   * its neighbour list is requested when pass 0 of the current point starts; its row offsets go to the OTHER slot of a double-buffered LDS table and its
//     coordinates are requested between the first two anchors of pass 0, its neighbour factor is built two anchors later (wave 0's lanes < NN, between
//     anchors, where the pass's register pressure is lowest); the barriers of pass 0's step 2 publish the tables;
//   * in pass 1 the wave re-reads its row offsets / neighbour factor IN PLACE from the next point's slot exactly where the pipeline crosses the pass
//     boundary (the loads of chunk g + 1 + D >= NSTEP, the pre-activations of step g + 2 >= NSTEP), so the last steps of a point gather, stage, read and
//     split the first chunks of the NEXT point -- the steps that used to fetch a clamped anchor 59 for nothing.  No setup barrier, no dependent round
//     trip and no pipeline fill is left between points; the first point of a workgroup takes the old prologue.
// Work: XCD x (= blockIdx.x % 8) owns a contiguous eighth of every scan's processing order (the Morton order the encoder passes: neighbouring points share
// most of their gathered rows in the XCD's L2); its workgroups take the first two items statically and every further one from the launch's per-XCD work
// counter (common.h), asked for two points ahead -- under the multi-stream pipeline other kernels share the compute units unevenly.  Items are independent
// and a point's arithmetic does not depend on its place in a workgroup's list: results are bitwise independent of the distribution and of the order.
//
// Kernel arguments as ONE struct, read through the kernarg segment: the persistent loop carries ~50 scalars, and thirteen pointers held in SGPR pairs
// from the kernel's entry pushed the register allocator into spilling scalars to VGPR lanes (and VGPRs to scratch: 256 of them are all a workgroup of
// the 32-channel kernels may use).  The pointers of the point setup / epilogue (used once per point) are re-read where they are used -- a scalar load
// from the constant cache -- through a laundered copy of the segment pointer, so they occupy no register in between; the pointers of the inner loop
// are read once.
struct YArgs {
    int nb, p1, p2;
    float inv_sigma;
    const float* xyz; const float* new_xyz; const int* ball_idx;
    const unsigned short* Fq; const bf16x8* kq; const bf16x8* Wq;
    const float* wsc; const float* fsc; const float* bias;
    float* out; const int* order; double* stat_part; unsigned* ctr;
};
typedef const YArgs __attribute__((address_space(4)))* y_args_ptr;
#define Y_RARE(field) ([&] { y_args_ptr a_ = A_; asm volatile("" : "+s"(a_)); return a_->field; }())

template <int CIN, int COUT, int NCH, int D>
__global__ void __launch_bounds__(256, INTER_Y_WPE(CIN)) inter_so3conv_y_kernel(YArgs args_) {
    const y_args_ptr A_ = (y_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();      // the struct is the kernel's only explicit argument: offset 0
    const int p1 = A_->p1, p2 = A_->p2;
    const unsigned short* __restrict__ Fq = A_->Fq;
    const bf16x8* __restrict__ kq = A_->kq;
    const bf16x8* __restrict__ Wq = A_->Wq;
    const bool has_ctr = A_->ctr != nullptr;
    constexpr int NN = 32 * NCH;
    constexpr int AG = 32, NJ = 8, NG = 2;         // anchors per pass, per wave and pass; passes per point
    constexpr int NT32 = CIN / 32;                 // 32-channel tiles of step 1
    constexpr int MT2 = COUT / 32;                 // 32-wide output tiles of step 2
    constexpr int CH = CIN / 2;                    // channels per X1 half
    constexpr int NKR = CH / 2;                    // accumulator registers of a lane per half
    constexpr int KH = CH * KS;                    // contraction length of step 2 per half
    constexpr int S = KH + Y_PAD(CIN);             // X1s row stride (floats), S / 4 odd (see inter_so3conv_x32_kernel); 32 input channels: the tile,
                                                   // the tables and the staging tiles of TWO workgroups must fit the 160 KB of a compute unit
    static_assert((S / 4) % 2 == 1, "row stride");
    constexpr int PS = COUT + 4;
    static_assert(4 * PS <= S, "the partial table must fit the X1 tile it aliases");
    constexpr int ROWB = 2 * CIN * 2;              // bytes of one (q, a) row: two planes of CIN fp16
    constexpr int PPR = CIN / 8, RPI = 64 / PPR, NRB = 32 / RPI;
    constexpr int PLB = 32 * CIN * 2, STG = 2 * PLB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X1s = smem;                             // [32][S]
    float* part = smem;                            // [4 waves][32 cols][PS], aliases X1s between the last product of a pass and the next pass
    unsigned* geo = reinterpret_cast<unsigned*>(smem + AG * S);        // [2 slots][NN][16 dwords]: the neighbour factor, 32 bf16 slots per neighbour
    unsigned* noffs = geo + 2 * NN * 16;                                // [2 slots][NN] byte offset of the neighbour's anchor-0 row
    float* dump = reinterpret_cast<float*>(noffs + 2 * NN);             // [64 lanes][4]: where the lanes of the unused kernel points 24 .. 31 store
    unsigned* s_next = reinterpret_cast<unsigned*>(dump + 256);         // the work item after next (from the counter)
    double* s_stat = reinterpret_cast<double*>(dump + 260);             // [256 threads][2]: the statistics' partial sums of pass 0, parked over pass 1's step 1
    __shared__ __attribute__((aligned(16))) char stage[4 * STG];       // [4 waves][STG]: its own LDS object (see inter_so3conv_x_kernel)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- work items of this workgroup: XCD xcd walks the slots [xcd * per, xcd * per + cnt) of every scan's processing order, item m = (scan m / cnt, slot m % cnt)
    const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int per = (p2 + 7) >> 3;
    int cnt = p2 - xcd * per;
    cnt = cnt < 0 ? 0 : (cnt > per ? per : cnt);
    const int total = Y_RARE(nb) * cnt;
    if (wi >= total) return;
    // item -> (scan, output point): the processing order is one dependent, wave-uniform load; the value is made scalar (readfirstlane) only where it is
    // consumed, so the load of the point after next stays in flight over a whole anchor
    auto resolve = [&](int mm, int& bb, int& pp) {
        bb = mm / cnt;
        const int sl_ = xcd * per + (mm - bb * cnt);
        const int* ord_ = Y_RARE(order);
        pp = ord_ ? ord_[(size_t)bb * p2 + sl_] : sl_;
    };
    int m = wi, b, p, b_n, p_n;
    resolve(m, b, p);
    p = __builtin_amdgcn_readfirstlane(p);
    bool have_next = m + wpx < total;
    b_n = b; p_n = p;
    if (have_next) { resolve(m + wpx, b_n, p_n); p_n = __builtin_amdgcn_readfirstlane(p_n); }
    int slot = 0;                                   // LDS slot of the current point's tables

    // setup of the FIRST point, part 1: the neighbour list -> row offsets.  The first gathers of the point need nothing else: they are requested below, BEFORE the
    // coordinates are fetched and the neighbour factor is built (the point's two dependent memory round trips overlap instead of adding up)
    int q_nb = 0;
    if (tid < NN) {
        q_nb = Y_RARE(ball_idx)[((size_t)b * p2 + p) * NN + tid];
        noffs[tid] = (unsigned)(q_nb < 0 ? 0 : q_nb) * (unsigned)(NA * ROWB);
    }
    __syncthreads();
    const int kp = lane & 31, kg = lane >> 5;      // this lane's kernel point (weights: column of B; X1: column of D) and neighbour group / channel sub-block
    const bool kok = kp < KS;
    // staging image of one plane of a chunk (see inter_so3conv_x32_kernel): load side -- lane = (row rl of the row block, 16-byte piece sl)
    const int rl = lane / PPR, sl = lane % PPR;
    const unsigned pieceoff = NT32 == 1 ? (unsigned)(sl * 16) : (unsigned)((4 * (((sl >> 2) - (rl >> 1)) & 1) + (sl & 3)) * 16);
    unsigned roff[NCH][NRB];
    auto read_roff = [&](int t, const unsigned* tab) {       // row offsets of chunk position t from a slot of the offset table
#pragma unroll
#ifdef Y_ABL_SAMEROW
        for (int rb = 0; rb < NRB; ++rb) roff[t][rb] = 0 * tab[32 * t + RPI * rb + rl] + pieceoff;     // timing experiment: every gather hits one row (cache-resident)
#else
        for (int rb = 0; rb < NRB; ++rb) roff[t][rb] = tab[32 * t + RPI * rb + rl] + pieceoff;
#endif
    };
#pragma unroll
    for (int t = 0; t < NCH; ++t) read_roff(t, noffs);
    const char* Fb = reinterpret_cast<const char*>(Fq) + (size_t)b * p1 * NA * ROWB;
    const char* Fb_n = reinterpret_cast<const char*>(Fq) + (size_t)b_n * p1 * NA * ROWB;
    char* stg = stage + wave * STG;
    char* stg_w = stg + lane * 16;                 // write side: the lane's 16-byte piece of every 1-KiB row block
    // read side: lane group g = lane / 16 -> channels 16 (g & 1) .. of the tile, rows 8 (g >> 1) + (i >> 2) + {0, 4} of the 16-row K step; i = lane % 16
    unsigned toff[NT32];
    {
        const int g = lane >> 4, i = lane & 15;
        const int r0 = 8 * (g >> 1) + (i >> 2);
#pragma unroll
        for (int ct = 0; ct < NT32; ++ct)
            toff[ct] = (unsigned)(r0 * CIN * 2 + (NT32 == 1 ? 0 : 64 * ((ct + (r0 >> 1)) & 1)) + 32 * (g & 1) + 8 * (i & 3));
    }
    static_assert(256 % COUT == 0, "a thread's epilogue elements e = tid + 256 j all belong to output channel tid % COUT");
    const float ws_t = Y_RARE(wsc)[tid % COUT], b_t = Y_RARE(bias)[tid % COUT];

    // ---- the software pipeline of one wave.  Step g = 0 .. NSTEP - 1 of a pass: anchor j = g / NCH of the wave's NJ anchors of the pass, chunk g % NCH.
    //   gathered rows   global -> registers (ring of D chunks, plain loads: in order, tracked by the compiler) -> staging tile (ds_write) -> fragments
    //   during step g   the MFMAs of step g (fragments and split weights prepared during step g - 1) are interleaved, slot by slot, with
    //                   * the ring's chunk g + 1 going to the staging tile, the loads of chunk g + 1 + D into the freed registers,
    //                   * the fragment reads of chunk g + 1 into the other fragment set,
    //                   * the clamp / split of the weights of step g + 1 (their pre-activation MFMAs close step g - 1).
    //   Chunks / steps >= NSTEP belong to the NEXT pass: the second half of this point's anchors (pass 0) or the first half of the next point's (pass 1).
    //   Every slot ends with sched_barrier(0): the order below IS the schedule (left alone, the scheduler clusters the VALU work; asked with
    //   sched_group_barrier over a whole pass, its solver takes minutes).  No LDS-direct loads: with one staging tile per wave they bound a step
    //   from below by the memory round trip (profiles/r05_inter_conv_latency_bound.txt), and any LDS access behind one is drained by the compiler.
    constexpr int NSTEP = NJ * NCH;                // steps per pass and wave
    static_assert(NSTEP % D == 0 && NSTEP % 2 == 0 && D + 1 + NCH <= NSTEP, "ring depth");
    constexpr int NMF = 6 * NT32;                  // step-1 MFMAs (= slots) per step: 2 K steps x 3 cross terms x channel tiles
    constexpr int NF = 4 * NT32;                   // operand fragments of the gathered rows per step (K step, plane, channel tile; two transposing reads each)
    constexpr int NL = 2 * NRB;                    // 16-byte loads per lane and chunk
    constexpr int NW = NL / 2;                     // slots that stage two pieces each
    static_assert(NW + NL <= NMF && NW + NF <= NMF, "slot plan");
    f32x4 ring[D][NL];
    // the neighbour factor of this lane: resident for one chunk per anchor (re-read only where the pipeline crosses the pass boundary); with two chunks
    // per anchor ONE buffer that every step re-reads for the chunk it forms pre-activations for (2 of the step's ~20 LDS reads; 8 registers less)
    constexpr bool GEO_RES = NCH == 1;
    bf16x8 geo_r[GEO_RES ? NCH : 1][2];
    f16x8 bf[2][NT32][2][2] = {};                   // [step parity][channel tile][K step][plane]
    y_u32x4 aws[2][2][2] = {};                      // [step parity][K step of the chunk][plane]
    bf16x8 kpn[2][2] = {};                          // kernel-point factor of this wave's current / next anchor: [anchor parity][K step]
    f32x16 acc[NT32];
    // anchor of sequence index q = 0 .. 15 of a point (pass q / NJ, the wave's anchor q % NJ; larger q wrap into the next point's sequence; anchors past 59
    // are clamped by the loaders: harmless repeats of anchor 59, the pipeline stays branch-free)
    auto anchor_of = [&](int q) { const int qq = q & (NG * NJ - 1); const int a = (qq / NJ) * AG + wave * NJ + (qq % NJ); return a < NA ? a : NA - 1; };
    auto issue_kp = [&](int q, bf16x8 (&dst)[2]) {
#ifdef Y_ABL_NOPRE
        return;
#endif
        const bf16x8* src = kq + (size_t)anchor_of(q) * 128 + lane;
        dst[0] = src[0]; dst[1] = src[64];
    };
    // piece k = (plane, row block) of chunk gl of the pass that starts at global step G0 (gl >= NSTEP: the next pass, whose rows are `base`'s)
    auto load_piece = [&](int G0, int gl, const char* base, int k, f32x4& dst) {
        const char* src = base + (size_t)anchor_of((G0 + gl) / NCH) * ROWB + roff[gl % NCH][k % NRB] + (k / NRB) * CIN * 2;
#ifdef Y_ABL_NOLOAD
        asm volatile("" : "=v"(dst) : "v"(src));
#else
        dst = *reinterpret_cast<const f32x4*>(src);
#endif
    };
    auto stage_piece = [&](int k, const f32x4& v) { *reinterpret_cast<f32x4*>(stg_w + (k / NRB) * PLB + (k % NRB) * 1024) = v; };
    auto read_frag = [&](int r, f16x8 (&dst)[NT32][2][2]) {      // fragment r = (K step h2, plane, channel tile), in the order the MFMAs want them
        const int ct = r % NT32, pl = (r / NT32) % 2, h2 = r / (2 * NT32);
        const char* pa = stg + pl * PLB + toff[ct] + 16 * h2 * CIN * 2;
        const bf16x4 lo4 = x_tr16(pa), hi4 = x_tr16(pa + 4 * CIN * 2);
        dst[ct][h2][pl] = __builtin_bit_cast(f16x8, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto gen_pre = [&](int t, const bf16x8 (&kpa)[2]) {          // pre-activations of one (anchor, chunk): P[v] = row 8 (v / 4) + 4 kg + v % 4 of column kp
#ifdef Y_ABL_NOPRE
        f32x16 P = zero16;
        asm volatile("" : "+v"(P) : "v"(kpa[0]));
        return P;
#else
        f32x16 P = __builtin_amdgcn_mfma_f32_32x32x16_bf16(geo_r[GEO_RES ? t : 0][0], kpa[0], zero16, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(geo_r[GEO_RES ? t : 0][1], kpa[1], P, 0, 0, 0);
#endif
    };
    // clamp to [0, 1] (the weight's mathematical range) + two-plane fp16 split of the pair k = (P[2 k], P[2 k + 1]) as a software pipeline over the
    // pairs: tick tau runs stage 1 of pair tau (clamp), stage 2 of pair tau - 1 (h = fp16 pair), stage 3 of pair tau - 2 (l = w - h:
    // v_fma_mix_f32 reads the fp16 half directly), stage 4 of pair tau - 3 (l as fp16 pair) -- inside a tick nothing depends on anything (a VALU
    // instruction that reads its predecessor's result issues ~4 cycles late, profiles/r05_valu_rates_f16.txt).  K step h2 = k / 4 of step 1,
    // dword d = k % 4 of its fragment.
    // (Round 6 tried a three-instruction form -- v_cvt_pkrtz_f16_f32 clamp, v_fma_mixlo_f16 clamp, v_fma_mixhi_f16 clamp: 29 % fewer VALU instructions in
    // step 1 -- and measured NO gain (9.90 against 9.68 ms for the three launches) before its semantics were even right: the step is not bound by its VALU
    // count; profiles/r06_inter_conv_persistent.txt.)
    struct Split { float w[16]; unsigned h[8]; float l[16]; };
    constexpr int NTICK = 11;
    auto split_tick = [&](int k, const f32x16& P, Split& S_, y_u32x4 (&aw)[2][2]) {
#ifdef Y_ABL_NOSPLIT
        if (k == 0) asm volatile("" :: "v"(P));
        return;
#endif
        if (k < 8) {
            S_.w[2 * k] = __builtin_amdgcn_fmed3f(P[2 * k], 0.0f, 1.0f);
            S_.w[2 * k + 1] = __builtin_amdgcn_fmed3f(P[2 * k + 1], 0.0f, 1.0f);
        }
        if (k >= 1 && k < 9) {
            const int pr = k - 1;
            S_.h[pr] = __builtin_bit_cast(unsigned, (f16x2){(_Float16)S_.w[2 * pr], (_Float16)S_.w[2 * pr + 1]});      // both planes rounded to nearest (split_bf16.h)
            aw[pr / 4][0][pr % 4] = S_.h[pr];
        }
        if (k >= 2 && k < 10) {
            const int pr = k - 2;
#ifdef Y_NO_FMA_MIX
            const f16x2 h = __builtin_bit_cast(f16x2, S_.h[pr]);
            S_.l[2 * pr] = S_.w[2 * pr] - (float)h[0]; S_.l[2 * pr + 1] = S_.w[2 * pr + 1] - (float)h[1];
#else
            // l = fp16-half * (-1) + w in one instruction (exact: w - h has at most 13 significant bits)
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(S_.l[2 * pr]) : "v"(S_.h[pr]), "v"(S_.w[2 * pr]));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(S_.l[2 * pr + 1]) : "v"(S_.h[pr]), "v"(S_.w[2 * pr + 1]));
#endif
        }
        if (k >= 3) {
            const int pr = k - 3;
            aw[pr / 4][1][pr % 4] = __builtin_bit_cast(unsigned, (f16x2){(_Float16)S_.l[2 * pr], (_Float16)S_.l[2 * pr + 1]});
        }
    };
    // X1 store addresses of this lane (LDS byte addresses): column 0 of the wave, kernel point kp, channel blocks (2 q + kg) ^ sw(kp), q < NKR / 4;
    // the lanes of the kernel points 24 .. 31 store into the dump slot (no branch).  Column j adds j * xmul -- formed per anchor from an opaque
    // copy: left to itself the compiler precomputes all NJ * NKR / 4 addresses, spills them, and every reload waits for the whole load ring.
    typedef __attribute__((address_space(3))) f32x4* x_lds_f4;
    unsigned xaddr[NKR / 4];
#pragma unroll
    for (int q = 0; q < NKR / 4; ++q)
        xaddr[q] = kok ? (unsigned)(uintptr_t)(X1s + wave * NJ * S + kp * CH + 4 * (((2 * q + kg) ^ (CH == 16 ? (kp >> 1) & 3 : kp & 7)))) : (unsigned)(uintptr_t)(dump + 4 * lane);
    const unsigned xmul = kok ? (unsigned)(S * 4) : 0u;
    auto x1_store = [&](int j, int q4, float a0, float a1, float a2, float a3) {
        unsigned m_ = xmul;
        asm volatile("" : "+v"(m_));
        *(x_lds_f4)(uintptr_t)(xaddr[q4] + (unsigned)j * m_) = (f32x4){a0, a1, a2, a3};
    };
    // the neighbour factor of neighbour n = tid (relative coordinates x, y, z; q < 0: a padded slot) -> 16 dwords of a slot of the geo table.  slot s = 5 t + c:
    // plane y_geo_plane(t) of component c of [a_n, 1, G_x, G_y, G_z]
    auto write_geo = [&](unsigned* gtab, int q, float x, float y, float z) {
        // Two call sites (the first point of a workgroup, every further point): the arithmetic is spelled out, fused operations included, and the compiler may
        // not contract it any other way -- left to itself it fused the two inlined copies differently, and a point's last bits depended on whether it was
        // a workgroup's first point (caught by the ordered-vs-plain test: 1e-7 of the output scale)
#pragma clang fp contract(off)
        const float inv_sigma = Y_RARE(inv_sigma);
        const float d2 = fmaf(z, z, x * x) + y * y;
        const float g2 = inv_sigma + inv_sigma;
        const Y3 s0 = y_split3(q < 0 ? -1e30f : fmaf(-inv_sigma, d2, 1.0f)), s1 = {0x3f80u, 0u, 0u},
                 s2 = y_split3(g2 * x), s3 = y_split3(g2 * y), s4 = y_split3(g2 * z);
#define Y_SLOT(s) ((s) >= 30 ? 0u : y_pick((s) % 5 == 0 ? s0 : (s) % 5 == 1 ? s1 : (s) % 5 == 2 ? s2 : (s) % 5 == 3 ? s3 : s4, y_geo_plane((s) / 5)))
#define Y_DW(d) (Y_SLOT(2 * (d)) | (Y_SLOT(2 * (d) + 1) << 16))
        const unsigned dw[16] = {Y_DW(0), Y_DW(1), Y_DW(2), Y_DW(3), Y_DW(4), Y_DW(5), Y_DW(6), Y_DW(7), Y_DW(8), Y_DW(9), Y_DW(10), Y_DW(11), Y_DW(12), Y_DW(13), Y_DW(14), Y_DW(15)};
#undef Y_DW
#undef Y_SLOT
        y_u32x4* gr = reinterpret_cast<y_u32x4*>(gtab + tid * 16);
#pragma unroll
        for (int d4 = 0; d4 < 4; ++d4) gr[d4] = (y_u32x4){dw[4 * d4], dw[4 * d4 + 1], dw[4 * d4 + 2], dw[4 * d4 + 3]};
    };
    // the neighbour factor of this lane: row mrow = lane % 32 of the pre-activation product = the neighbour in staging row swap23(mrow)
    const int nloc = (kp & ~12) | ((kp & 4) << 1) | ((kp & 8) >> 1);
    auto read_geo = [&](int t, const unsigned* gtab) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            geo_r[GEO_RES ? t : 0][j] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(gtab) + (32 * t + nloc) * 64 + 32 * j + 16 * kg);
    };

    // prologue of the first point: the first D chunks requested, chunk 0 staged and read, the weights of step 0
    issue_kp(0, kpn[0]);
    issue_kp(1, kpn[1]);
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int k = 0; k < NL; ++k) load_piece(0, d, Fb, k, ring[d][k]);
    // setup, part 2 (the gathers above are in flight): neighbour coordinates -> the neighbour factor of the weights' pre-activation
    if (tid < NN) {
        const int qq = q_nb < 0 ? 0 : q_nb;
        const float* X = Y_RARE(xyz) + (size_t)b * 3 * p1;
        const float* NX = Y_RARE(new_xyz);
        write_geo(geo, q_nb, X[qq] - NX[((size_t)b * 3 + 0) * p2 + p], X[p1 + qq] - NX[((size_t)b * 3 + 1) * p2 + p], X[2 * p1 + qq] - NX[((size_t)b * 3 + 2) * p2 + p]);
    }
    __syncthreads();
    if (GEO_RES) {
#pragma unroll
        for (int t = 0; t < NCH; ++t) read_geo(t, geo);
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) stage_piece(k, ring[0][k]);
#pragma unroll
    for (int k = 0; k < NL; ++k) load_piece(0, D, Fb, k, ring[0][k]);
#pragma unroll
    for (int r = 0; r < NF; ++r) read_frag(r, bf[0]);
    f32x16 P;                                       // pre-activations of the step AFTER the current one (formed at the end of the step before it)
    {
        if (!GEO_RES) read_geo(0, geo);
        P = gen_pre(0, kpn[0]);
        Split S_;
#pragma unroll
        for (int k = 0; k < NTICK; ++k) split_tick(k, P, S_, aws[0]);
        if (!GEO_RES) read_geo(1 % NCH, geo);
        P = gen_pre(1 % NCH, kpn[(1 / NCH) & 1]);
    }

    f32x16 y[MT2];
    float keep[NJ][NT32 == 1 ? 8 : 16];            // second channel half of the wave's anchors of a pass
#pragma unroll 1
    for (int pass = 0;; ++pass) {
        const int ag = pass & 1;
        // (declared per pass: as loop-carried variables that only some lanes / passes assign they would stay live through the whole pass)
        float cx = 0.f, cy = 0.f, cz = 0.f;        // the next point's neighbour coordinates (lanes < NN of wave 0), in flight over two anchors of pass 0
        float ox = 0.f, oy = 0.f, oz = 0.f;        // ... and its own
        unsigned grabbed = 0u;
        int b_2 = 0, p_2 = 0;                      // the point after next (resolved during pass 1)
        bool have_2 = false;
        q_nb = 0;
        const int G0 = ag * NSTEP;
        // what the pipeline reaches when it crosses the end of this pass: the second half of this point's anchors, or the next point
        const char* Fb_np = ag ? Fb_n : Fb;
        const unsigned* noffs_np = noffs + (ag ? (slot ^ 1) : slot) * NN;
        const unsigned* geo_np = geo + (ag ? (slot ^ 1) : slot) * (NN * 16);
        const unsigned* geo_cur = geo + slot * (NN * 16);
        if (ag == 0) {
            // the next point: its neighbour list (consumed an anchor later), and the request for the work item after it
            if (tid < NN) q_nb = Y_RARE(ball_idx)[((size_t)b_n * p2 + p_n) * NN + tid];
            if (has_ctr && tid == 0) grabbed = __hip_atomic_fetch_add(Y_RARE(ctr) + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // the point after next: its index in the processing order (a dependent load; made scalar an anchor later, when it has long arrived)
            const int m2 = has_ctr ? (int)__builtin_amdgcn_readfirstlane(*s_next) : m + 2 * wpx;
            have_2 = have_next && m2 < total;
            if (have_2) resolve(m2, b_2, p_2);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = ag * NJ + j;
            // the next point's setup rides in the FIRST anchors of pass 0, where the pass's register pressure is lowest (`keep` fills up towards its end);
            // wave 0's lanes < NN only, between two anchors (not inside the slot schedule).  Every load consumed here was issued at least an anchor ago and is
            // older than the ring's loads the wave has waited for since: no exposed latency.
            if (j == 1) {
                if (ag == 0) {
                    // row offsets -> the other slot of the table (published by the barriers of this pass's step 2, read back in pass 1); coordinates requested
                    if (tid < NN) {
                        noffs[(slot ^ 1) * NN + tid] = (unsigned)(q_nb < 0 ? 0 : q_nb) * (unsigned)(NA * ROWB);
                        const int qq = q_nb < 0 ? 0 : q_nb;
                        const float* X = Y_RARE(xyz) + (size_t)b_n * 3 * p1;
                        cx = X[qq]; cy = X[p1 + qq]; cz = X[2 * p1 + qq];
                        // the point's own coordinates: requested here as well (every lane the same address: one cache line), two anchors before their use --
                        // asked for where they are used, they were a vector load with its wait right behind it: one exposed round trip per point for wave 0
                        const float* NX = Y_RARE(new_xyz);
                        ox = NX[((size_t)b_n * 3 + 0) * p2 + p_n]; oy = NX[((size_t)b_n * 3 + 1) * p2 + p_n]; oz = NX[((size_t)b_n * 3 + 2) * p2 + p_n];
                    }
                    if (has_ctr && tid == 0) *s_next = 2u * (unsigned)wpx + grabbed;
                } else if (have_2) {
                    p_2 = __builtin_amdgcn_readfirstlane(p_2);
                }
            }
            if (j == 3 && ag == 0 && tid < NN)          // the next point's neighbour factor -> the other slot
                write_geo(geo + (slot ^ 1) * (NN * 16), q_nb, cx - ox, cy - oy, cz - oz);
#pragma unroll
            for (int t = 0; t < NCH; ++t) {
                const int g = j * NCH + t, sp = g & 1;                 // local step and its parity (fragment set, weight set)
                const int rs = (g + 1) % D;                             // ring entry of chunk g + 1
                // this anchor's kernel-point factor was last used a step ago (the pre-activations of its last chunk, formed two steps ahead of their
                // use): the anchor after next goes into its set
                if (t == 0) issue_kp(q + 2, kpn[j & 1]);
                // the pass boundary: the row offsets of the chunks this step starts to request / the neighbour factor of the pre-activations it forms, in place
                if (g + 1 + D >= NSTEP && g + 1 + D < NSTEP + NCH) read_roff((g + 1 + D) % NCH, noffs_np);
                if (!GEO_RES) read_geo((g + 2) % NCH, g + 2 >= NSTEP ? geo_np : geo_cur);
                else if (g + 2 >= NSTEP && g + 2 < NSTEP + NCH) read_geo((g + 2) % NCH, geo_np);
                __builtin_amdgcn_sched_barrier(0);
                Split S_;
                constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};     // (weight plane, feature plane): l * h, h * l, h * h
#pragma unroll
                for (int i = 0; i < NMF; ++i) {
                    {
                        const int ct = i % NT32, term = (i / NT32) % 3, h2 = i / (3 * NT32);
#ifndef Y_ABL_NOMFMA1
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[sp][ct][h2][PB[term]], __builtin_bit_cast(f16x8, aws[sp][h2][PA[term]]),
                                                                         (t == 0 && i < NT32) ? zero16 : acc[ct], 0, 0, 0);
#else
                        asm volatile("" :: "v"(aws[sp][h2][PA[term]]), "v"(bf[sp][ct][h2][PB[term]]));
                        if (t == 0 && i < NT32) acc[ct] = zero16;
#endif
                    }
                    // chunk g + 1: ring -> staging tile (two pieces per slot), then its fragments (one per slot); the freed registers take chunk g + 1 + D
#ifndef Y_ABL_NOSTAGE
                    if (i < NW) { stage_piece(2 * i, ring[rs][2 * i]); stage_piece(2 * i + 1, ring[rs][2 * i + 1]); }
#else
                    if (i < NW) asm volatile("" :: "v"(ring[rs][2 * i]), "v"(ring[rs][2 * i + 1]));
#endif
                    if (i >= NW && i < NW + NL) load_piece(G0, g + 1 + D, g + 1 + D >= NSTEP ? Fb_np : Fb, i - NW, ring[rs][i - NW]);
#ifndef Y_ABL_NOFRAG
                    if (i >= NW && i < NW + NF) read_frag(i - NW, bf[sp ^ 1]);
#endif
                    // the next step's weights
                    if (NT32 == 2) { if (i >= 1 && i < 1 + NTICK) split_tick(i - 1, P, S_, aws[sp ^ 1]); }
                    else { split_tick(2 * i, P, S_, aws[sp ^ 1]); if (2 * i + 1 < NTICK) split_tick(2 * i + 1, P, S_, aws[sp ^ 1]); }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the pre-activations of step g + 2 (P is free: the last tick above consumed those of step g + 1); they complete across the step boundary
                P = gen_pre((t + 2) % NCH, kpn[(j + (t + 2) / NCH) & 1]);
                if (t == NCH - 1) {
                    // anchor end.  D[c][k]: this lane = kernel point kp, channels 8 (v / 4) + 4 kg + v % 4 of each 32-channel tile: first half -> X1 tile, second half parked
#pragma unroll
                    for (int q4 = 0; q4 < NKR / 4; ++q4) x1_store(j, q4, acc[0][4 * q4], acc[0][4 * q4 + 1], acc[0][4 * q4 + 2], acc[0][4 * q4 + 3]);
#pragma unroll
                    for (int v = 0; v < NKR; ++v) keep[j][v] = NT32 == 1 ? acc[0][NKR + v] : acc[NT32 - 1][v];
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
            for (int v = 0; v < 16; ++v) y[mt][v] = 0.f;
#ifndef Y_ABL_NOSTEP2
        H32Step2<CIN, COUT, Y_PAD(CIN), Y_S2_DEPTH(CIN)> s2;
        const bf16x8* Wq_g = Wq;
        asm volatile("" : "+s"(Wq_g));
        s2.prime(Wq_g, wave, lane);
        __syncthreads();
        s2.template half<0>(y, X1s, Wq_g, wave, lane);
#endif
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int q4 = 0; q4 < NKR / 4; ++q4) x1_store(j, q4, keep[j][4 * q4], keep[j][4 * q4 + 1], keep[j][4 * q4 + 2], keep[j][4 * q4 + 3]);
        }
        __syncthreads();
#ifndef Y_ABL_NOSTEP2
        s2.template half<1>(y, X1s, Wq_g, wave, lane);
#endif
        __syncthreads();                                // every wave finished reading X1s: the partial table may overwrite it
        // y[mt][v] = 2^kw(o) Y[o = 32 mt + 8 (v / 4) + 4 kg + v % 4][anchor column = lane % 32]
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
                *reinterpret_cast<float4*>(&part[(wave * AG + kp) * PS + 32 * mt + 8 * q4 + 4 * kg]) = make_float4(y[mt][4 * q4], y[mt][4 * q4 + 1], y[mt][4 * q4 + 2], y[mt][4 * q4 + 3]);
        __syncthreads();
        {
            // this thread's sums of its output channel (fp64): pass 0 starts them, pass 1 continues from LDS (8 registers that would be live through step 1)
            double st_s = 0.0, st_q = 0.0;
            if (ag == 1) { const double2 t_ = *reinterpret_cast<const double2*>(s_stat + 2 * tid); st_s = t_.x; st_q = t_.y; }
            float* outp = Y_RARE(out) + ((size_t)b * p2 + p) * NA * COUT;
            const float* fsc = Y_RARE(fsc);
            const float wsf = fsc ? ws_t * fsc[b] : ws_t;      // the weight planes carry every output channel's row times its own power of two, the feature planes (when the
                                                               // caller scaled them: fsc) their scan's -- both exact, both taken out here; o == tid % COUT
            // (all of the thread's partial sums are read before the first is used: read element by element, each element waited out its own LDS round trip)
            constexpr int NE = AG * COUT / 256;
            const int o = tid % COUT, col0 = tid / COUT;
            float pv[NE][4];
#pragma unroll
            for (int e = 0; e < NE; ++e)
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) pv[e][w4] = part[(w4 * AG + col0 + e * (256 / COUT)) * PS + o];
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int a = ag * AG + col0 + e * (256 / COUT);
                if (a < NA) {
                    float v = pv[e][0] + pv[e][1];
                    v += pv[e][2] + pv[e][3];
                    v = v * wsf + b_t;
                    outp[(size_t)a * COUT + o] = v;
                    st_s += (double)v; st_q += (double)v * (double)v;
                }
            }
            *reinterpret_cast<double2*>(s_stat + 2 * tid) = make_double2(st_s, st_q);
        }
        __syncthreads();                                // the partial table is read: the next pass may write X1s; the next point's tables are visible
        if (ag == 1) {
            double* stat_part = Y_RARE(stat_part);
            if (stat_part) {
                static_assert(256 % COUT == 0, "a thread must keep one output channel");
                // the threads' sums are in s_stat (written before the barrier above); the 256 / COUT threads of a channel are added in thread order.  The
                // next write of s_stat (pass 0 of the next point, behind its step-2 barriers) cannot overtake these reads
                if (tid < COUT) {
                    double a0 = 0.0, a1 = 0.0;
#pragma unroll
                    for (int k = 0; k < 256 / COUT; ++k) { a0 += s_stat[2 * (k * COUT + tid)]; a1 += s_stat[2 * (k * COUT + tid) + 1]; }
                    double* sp_ = stat_part + ((size_t)b * p2 + p) * 2 * COUT;
                    sp_[tid] = a0; sp_[COUT + tid] = a1;
                }
            }
            if (!have_next) break;
            // the points rotate: the tables of the next point are in the other slot, its first chunks in the ring, its first weights split
            m = has_ctr ? 0 : m + wpx;                  // (with a counter the static successor is not used)
            b = b_n; p = p_n; Fb = Fb_n;
            slot ^= 1;
            have_next = have_2;
            if (have_2) { b_n = b_2; p_n = p_2; Fb_n = reinterpret_cast<const char*>(Fq) + (size_t)b_n * p1 * NA * ROWB; }
        }
    }
}

#ifndef INTER_Y_DEPTH
#define INTER_Y_DEPTH(CIN) 2
#endif

template <int CIN, int COUT, int NCH>
static int launch_y(int b, int p1, int p2, float sigma, const float* xyz, const float* new_xyz, const int* idx, const void* Fq, const void* kq,
                    const void* Wq, const float* wsc, const float* fsc, const float* bias, float* out, const int* order, double* stat_part, hipStream_t st) {
    constexpr int NN = 32 * NCH;
    const size_t lds = (size_t)(32 * ((CIN / 2) * KS + Y_PAD(CIN)) + 2 * 17 * NN + 256 + 4 + 1024) * sizeof(float);
    auto kern = inter_so3conv_y_kernel<CIN, COUT, NCH, INTER_Y_DEPTH(CIN)>;
    {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    // persistent grid: every compute unit's share of resident workgroups, a multiple of the 8 XCDs (blockIdx.x % 8 = XCD, round-robin dispatch)
    const long items = (long)b * p2;
    long wpx = (long)((etch_cu_count() + 7) / 8) * INTER_Y_WPE(CIN);
    if (wpx > (items + 7) / 8) wpx = (items + 7) / 8;
    if (wpx < 1) wpx = 1;
    const YArgs ka = {b, p1, p2, 1.0f / sigma, xyz, new_xyz, idx, reinterpret_cast<const unsigned short*>(Fq), reinterpret_cast<const bf16x8*>(kq),
                      reinterpret_cast<const bf16x8*>(Wq), wsc, fsc, bias, out, order, stat_part, etch_work_counter_slot(st)};
    hipLaunchKernelGGL(kern, dim3((unsigned)(8 * wpx)), dim3(256), lds, st, ka);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

// x [rows][C] fp32 -> planes [rows][2][C] fp16 (split2h); thread = 4 consecutive channels
__global__ void __launch_bounds__(256) split2_planes_f16_kernel(long n4, int C, const float* __restrict__ x, unsigned short* __restrict__ planes) {
    const int c4 = C >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long row = i / c4;
        const int c = (int)(i - row * c4) * 4;
        uint2 h, l;
        split2h_pack4(reinterpret_cast<const float4*>(x)[i], h, l);
        unsigned short* pr = planes + (size_t)row * 2 * C + c;
        *reinterpret_cast<uint2*>(pr) = h; *reinterpret_cast<uint2*>(pr + C) = l;
    }
}

// ---- features of unknown scale: a caller of the operator API may hand the conv anything; the fp16 planes carry 23 bits only near
// [2^-2, 2^4) and overflow above 65 504.  Per scan: the maximum magnitude (order-independent: atomic max on the bit patterns of non-negative floats) -> the
// power of two k that puts it into [8, 16) -> planes of x 2^k (exact) and the scan's output factor 2^-k, which the conv's epilogue applies with the
// output channel's own power (the conv is linear in its features; |X1| <= nn x 16 stays inside fp16's range).  Per SCAN, not per call: a scan's result
// never depends on its batch neighbours.
__global__ void __launch_bounds__(256) absmax_per_scan_kernel(long n4, const float* __restrict__ x, unsigned* __restrict__ mx) {
    const float4* xs = reinterpret_cast<const float4*>(x) + (size_t)blockIdx.y * n4;
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) m = etch_max4abs(xs[i], m);
    m = etch_wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(mx + blockIdx.y, __float_as_uint(m));      // NaN never wins (m > 0 is false for it): NaN inputs stay NaN in the planes
}
__global__ void __launch_bounds__(256) split2_planes_f16_scaled_kernel(long n4, int C, const float* __restrict__ x, const unsigned* __restrict__ mx,
                                                                       unsigned short* __restrict__ planes, float* __restrict__ fsc) {
    const int c4 = C >> 2;
    const int k = etch_scale_exp(__uint_as_float(mx[blockIdx.y]));
    const float sc = ldexpf(1.0f, k);
    if (blockIdx.x == 0 && threadIdx.x == 0) fsc[blockIdx.y] = ldexpf(1.0f, -k);
    const float4* xs = reinterpret_cast<const float4*>(x) + (size_t)blockIdx.y * n4;
    unsigned short* ps = planes + (size_t)blockIdx.y * n4 * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long row = i / c4;
        const int c = (int)(i - row * c4) * 4;
        const float4 v = xs[i];
        uint2 h, l;
        split2h_pack4(make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc), h, l);
        unsigned short* pr = ps + (size_t)row * 2 * C + c;
        *reinterpret_cast<uint2*>(pr) = h; *reinterpret_cast<uint2*>(pr + C) = l;
    }
}

extern "C" {

// x (b, rows, C) fp32 -> planes (b, rows, 2, C) fp16 of x[s] 2^k(s), k(s) = the power of two that puts scan s's maximum magnitude into [8, 16), and
// fsc[s] = 2^-k(s) (the `fsc` argument of etch_inter_so3conv_planes_kq).  mx (b unsigned) is workspace, ZEROED by the caller.
int etch_split2_planes_f16_scaled(int b, long rows, int C, const float* x, void* mx, void* planes, float* fsc, void* stream) {
    if (b <= 0 || rows <= 0) return ETCH_OK;
    if (C <= 0 || (C & 3) || ((uintptr_t)x & 15) || ((uintptr_t)planes & 7) || !mx || !fsc) return ETCH_EUNSUPPORTED;
    const long n4 = rows * (C / 4);
    long blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16 / (b < 16 ? b : 16)) blocks = 256 * 16 / (b < 16 ? b : 16);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(absmax_per_scan_kernel, dim3((unsigned)blocks, b), dim3(256), 0, (hipStream_t)stream, n4, x, reinterpret_cast<unsigned*>(mx));
    ETCH_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(split2_planes_f16_scaled_kernel, dim3((unsigned)blocks, b), dim3(256), 0, (hipStream_t)stream, n4, C, x, reinterpret_cast<const unsigned*>(mx),
                       reinterpret_cast<unsigned short*>(planes), fsc);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_split2_planes_f16(long rows, int C, const float* x, void* planes, void* stream) {
    if (rows <= 0) return ETCH_OK;
    if (C <= 0 || (C & 3) || ((uintptr_t)x & 15) || ((uintptr_t)planes & 7)) return ETCH_EUNSUPPORTED;
    const long n4 = rows * (C / 4);
    long blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(split2_planes_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n4, C, x, reinterpret_cast<unsigned short*>(planes));
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

// rk [60][24][3] fp32 (anchors @ kernel points, functional.py:296) + sigma -> kq [60][2][64][8] bf16 (122 880 bytes), once per layer
int etch_inter_kpoint_operand(float sigma, const float* rk, void* kq, void* stream) {
    if (sigma <= 0.f || !rk || !kq || ((uintptr_t)kq & 15)) return ETCH_EINVAL;
    hipLaunchKernelGGL(inter_kpoint_operand_kernel, dim3(NA), dim3(128), 0, (hipStream_t)stream, 1.0f / sigma, rk, reinterpret_cast<unsigned short*>(kq));
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

// etch_inter_so3conv_planes32 with the kernel weights formed on the matrix cores.  kq = etch_inter_kpoint_operand(sigma, rk); Wq32 and feats_planes as there.
int etch_inter_so3conv_planes_kq(int b, int cin, int cout, int p1, int p2, int nn, float sigma, const float* xyz, const float* new_xyz,
                                 const int* ball_idx, const void* feats_planes, const void* kq, const void* Wq32, const float* wsc, const float* fsc,
                                 const float* bias, float* out, const int* order, double* stat_part, void* stream) {
    if (b <= 0 || p2 <= 0) return ETCH_OK;
    if (sigma <= 0.f || !Wq32 || !wsc || !feats_planes || !kq) return ETCH_EINVAL;
    if (((uintptr_t)feats_planes & 15) || ((uintptr_t)Wq32 & 15) || ((uintptr_t)kq & 15)) return ETCH_EINVAL;
    if ((size_t)p1 * NA * 2 * cin * 2 >= ((size_t)1 << 32)) return ETCH_EUNSUPPORTED;      // 32-bit byte offsets inside a scan
    hipStream_t st = (hipStream_t)stream;
#define Y_CASE(CI, CO, NC) \
    if (cin == CI && cout == CO && nn == 32 * NC) return launch_y<CI, CO, NC>(b, p1, p2, sigma, xyz, new_xyz, ball_idx, feats_planes, kq, Wq32, wsc, fsc, bias, out, order, stat_part, st);
    Y_CASE(32, 32, 1) Y_CASE(32, 32, 2) Y_CASE(32, 64, 1) Y_CASE(32, 64, 2) Y_CASE(64, 64, 1) Y_CASE(64, 64, 2)
#undef Y_CASE
    return ETCH_EUNSUPPORTED;
}

}  // extern "C"
