// Dense layer -> ReLU -> grouped dot, fused on the gfx950 fp32 matrix cores:
//     out[r, g] = b2[g] + sum_{j < 128} relu( X[r, :] . W[g*128 + j, :] + b1[g*128 + j] ) * w2[g*128 + j]
// Two call sites of the path end in exactly this chain and previously round-tripped the (R x G*128) hidden
// activations through HBM:
//   * the confidence head  Conv1d(128, 128k, 1) -> ReLU -> Conv1d(128k, k, 1, groups=k)   (G = k = 86, K = 128;
//     /root/reference/src/models/pointtransformer_seg.py:145,183-189), hidden = 160 000 x 11 008 floats per batch;
//   * the tail of the direction head  MLP.net[0] -> ReLU -> (net[2] o so3_reg folded)    (G = 1, K = 64;
//     /root/reference/src/models/models_pointcloud.py:115-117), hidden = 9.6 M x 128 floats per batch.
//
// One workgroup = 128 (K = 64) or 64 (K = 128) rows of X, resident in LDS for all G groups (X is read from HBM exactly once); 8 waves, wave w
// owns the 16-column strip w of the current group's 128 hidden columns: 8 / 4 row tiles x 1 column tile of 16x16x4 f32
// MFMAs, B fragments (weights) streamed straight from L2 in fragment order with a one-step register prefetch, A fragments
// (X) read from LDS with one conflict-free ds_read_b128 per four MFMAs (interleaved-K order, see gemm.hip).  The epilogue
// applies bias / ReLU / w2 in registers, sums the strip's 16 columns with DPP row reductions, and the 8 strips through a
// double-buffered LDS table (one barrier per group, fixed summation order: results are run-to-run reproducible).
#include "common.h"
#include "split_bf16.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FD_J 128
#define FD_PAD 40      // LDS row = K + 40 floats: (K+40)/4 = 10 (mod 16) for K in {64,128} -> conflict-free b128 fragment reads

template <int CTRL>
__device__ __forceinline__ float fd_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row; every lane of the row ends up with the total
__device__ __forceinline__ float fd_row_sum16(float v) {
    v += fd_dpp<0xB1>(v);      // quad_perm [1,0,3,2]
    v += fd_dpp<0x4E>(v);      // quad_perm [2,3,0,1]
    v += fd_dpp<0x141>(v);     // row_half_mirror
    v += fd_dpp<0x140>(v);     // row_mirror
    return v;
}

template <int K, bool PERM, int FD_ROWS>
__global__ void __launch_bounds__(512) linear_relu_dot_kernel(long R, int G, const float* __restrict__ X, long ldx,
                                                              const float* __restrict__ W, long ldw, const float* __restrict__ b1,
                                                              const float* __restrict__ w2, const float* __restrict__ b2,
                                                              float* __restrict__ out, long ldo) {
    constexpr int S = K + FD_PAD, KT = K / 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Xs = lds;                         // [FD_ROWS][S]
    float* red = lds + FD_ROWS * S;          // [2][8][FD_ROWS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    // Persistent workgroup over the row tiles: the NEXT tile's rows travel through registers while the current tile is multiplied (with
    // one group -- the direction head's tail -- a tile is loaded for 2 us and multiplied for 2 us: un-overlapped, the matrix cores idled
    // 40 % of the time)
    constexpr int C4 = K / 4;
    constexpr int XL = FD_ROWS * C4 / 512;   // float4 per thread and tile
    static_assert(FD_ROWS * C4 % 512 == 0, "tile / thread geometry");
    const long ntiles = (R + FD_ROWS - 1) / FD_ROWS;
    float4 xn[XL];
    auto fetch = [&](long tile) {
        const long rt = tile * FD_ROWS;
#pragma unroll
        for (int h = 0; h < XL; ++h) {
            const int e = tid + 512 * h;
            const int row = e / C4, c = (e - row * C4) * 4;
            xn[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tile < ntiles && rt + row < R) xn[h] = *reinterpret_cast<const float4*>(X + (rt + row) * ldx + c);   // rows past R are zero
        }
    };
    // weight fragment of (group g, k-step t) for this wave's strip.  PERM: W is pre-permuted to fragment order
    // Wp[g][t][strip][lane][4] (one contiguous 1 KiB per wave load); else the plain row-major [G*128][ldw] matrix.
    auto wfrag = [&](int g, int t) -> float4 {
        if (PERM) return *reinterpret_cast<const float4*>(W + ((((long)g * KT + t) * 8 + wave) * 64 + lane) * 4);
        return *reinterpret_cast<const float4*>(W + ((long)g * FD_J + wave * 16 + fr) * ldw + t * 16 + fg * 4);
    };
    fetch(blockIdx.x);
    int buf = 0;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long r0 = tile * FD_ROWS;
    __syncthreads();                         // the previous tile (and its last reduction table) is consumed
#pragma unroll
    for (int h = 0; h < XL; ++h) {
        const int e = tid + 512 * h;
        const int row = e / C4, c = (e - row * C4) * 4;
        *reinterpret_cast<float4*>(&Xs[row * S + c]) = xn[h];
    }
    float4 bn = wfrag(0, 0);
    __syncthreads();
    fetch(tile + gridDim.x);                 // in flight during all G groups of this tile

    for (int g = 0; g < G; ++g) {
        constexpr int RT = FD_ROWS / 16;
        f32x4 acc[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const float4 b = bn;
            // prefetch the next fragment (next k-step, or the first of the next group; the last prefetch re-reads a valid one)
            if (t + 1 < KT) bn = wfrag(g, t + 1);
            else bn = wfrag(g + 1 < G ? g + 1 : g, 0);
            float4 a[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i) a[i] = *reinterpret_cast<const float4*>(&Xs[(i * 16 + fr) * S + t * 16 + fg * 4]);
#define FD_STEP(C) _Pragma("unroll") for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].C, b.C, acc[i], 0, 0, 0);
            FD_STEP(x) FD_STEP(y) FD_STEP(z) FD_STEP(w)
#undef FD_STEP
        }
        // epilogue: D[row = 16 i + 4 fg + q][col = fr]
        const int col = g * FD_J + wave * 16 + fr;
        const float bs = b1[col], ww = w2[col];
        float* rp = red + (buf * 8 + wave) * FD_ROWS;
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float s = fd_row_sum16(fmaxf(acc[i][q] + bs, 0.f) * ww);
                if (fr == ((i * 4 + q) & 15)) rp[i * 16 + fg * 4 + q] = s;    // spread the 32 stores over the row's lanes
            }
        __syncthreads();
        if (tid < FD_ROWS && r0 + tid < R) {
            const float* rr = red + buf * 8 * FD_ROWS + tid;
            float s = rr[0];
#pragma unroll
            for (int w = 1; w < 8; ++w) s += rr[w * FD_ROWS];
            out[(r0 + tid) * ldo + g] = s + b2[g];
        }
        buf ^= 1;
    }
    }
}

// ------------------------------------------------------------------------------------------------
// The same chain on the bf16 matrix cores with exactly split fp32 operands (x = hi + mid + lo, 8 + 8 + 8 mantissa bits; six cross products
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16: the error against fp64 of the fp32 MFMA, profiles/r03_bf16x3_split.txt, at 2.3 x its rate and
// beside the VALU -- the epilogue's bias / ReLU / w2 / DPP row sums now run in the matrix cores' shadow).  X is split once when a row tile is
// staged in LDS (three bf16 planes), W comes pre-split from the host (ops.lrd_weight_split).  Wave w = (row half w >> 2, strip pair w & 3): every
// X fragment read from LDS feeds two 16-column strips (with one strip per wave the planes' reads alone would saturate the LDS), every W fragment
// is read by the two row halves (the second read hits in L1).  128-row tiles for every K: the weight stream per row is what bounds the
// confidence head (G = 86: 8.4 MB of split weights per tile).
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void fd_split(const float v, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(v);
    const float r = v - __uint_as_float(h & 0xffff0000u);
    m = __float_as_uint(r);
    l = __float_as_uint(r - __uint_as_float(m & 0xffff0000u));
}
template <int K, int FD_ROWS>
__global__ void __launch_bounds__(512) linear_relu_dot_bx_kernel(long R, int G, const float* __restrict__ X, long ldx, const bf16x8* __restrict__ Wq,
                                                                 const float* __restrict__ b1, const float* __restrict__ w2,
                                                                 const float* __restrict__ b2, float* __restrict__ out, long ldo) {
    constexpr int SB = K + 8, KT = K / 32;       // bf16 row stride of a plane (SB / 8 odd: conflict-free 16-byte fragment reads); K steps
    constexpr int PLANE = FD_ROWS * SB;
    constexpr int RT = FD_ROWS / 32;             // 16-row tiles per wave (its row half)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned short* Xp = reinterpret_cast<unsigned short*>(lds);            // [3][FD_ROWS][SB]
    float* red = lds + 3 * PLANE / 2;                                        // [2][8 strips][FD_ROWS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int rh = wave >> 2, sp = wave & 3;     // row half, strip pair (strips 2 sp, 2 sp + 1)
    constexpr int C4 = K / 4;
    constexpr int XL = FD_ROWS * C4 / 512;       // float4 per thread and tile
    static_assert(FD_ROWS * C4 % 512 == 0, "tile / thread geometry");
    const long ntiles = (R + FD_ROWS - 1) / FD_ROWS;
    float4 xn[XL];
    auto fetch = [&](long tile) {
        const long rt = tile * FD_ROWS;
#pragma unroll
        for (int h = 0; h < XL; ++h) {
            const int e = tid + 512 * h;
            const int row = e / C4, c = (e - row * C4) * 4;
            xn[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tile < ntiles && rt + row < R) xn[h] = *reinterpret_cast<const float4*>(X + (rt + row) * ldx + c);   // rows past R are zero
        }
    };
    // W fragments of (group g, K step t, strip s): [g][t][strip][plane][lane] x 16 bytes
    auto wfrag = [&](int g, int t, int s, bf16x8 (&dst)[3]) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) dst[pl] = Wq[((((long)g * KT + t) * 8 + s) * 3 + pl) * 64 + lane];
    };
    fetch(blockIdx.x);
    int buf = 0;
    bf16x8 bn[2][3];                             // the NEXT K step's W fragments (one step of register look-ahead, across groups and tiles:
                                                 // the L2 latency of the weight stream is what an 8-wave workgroup cannot hide otherwise)
    wfrag(0, 0, 2 * sp, bn[0]); wfrag(0, 0, 2 * sp + 1, bn[1]);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long r0 = tile * FD_ROWS;
        __syncthreads();                         // the previous tile (and its last reduction table) is consumed
#pragma unroll
        for (int h = 0; h < XL; ++h) {
            const int e = tid + 512 * h;
            const int row = e / C4, c = (e - row * C4) * 4;
            const float v[4] = {xn[h].x, xn[h].y, xn[h].z, xn[h].w};
            unsigned hh[4], mm[4], ll[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fd_split(v[i], hh[i], mm[i], ll[i]);
            unsigned short* d = Xp + row * SB + c;
            *reinterpret_cast<uint2*>(d) = make_uint2(__builtin_amdgcn_perm(hh[1], hh[0], 0x07060302u), __builtin_amdgcn_perm(hh[3], hh[2], 0x07060302u));
            *reinterpret_cast<uint2*>(d + PLANE) = make_uint2(__builtin_amdgcn_perm(mm[1], mm[0], 0x07060302u), __builtin_amdgcn_perm(mm[3], mm[2], 0x07060302u));
            *reinterpret_cast<uint2*>(d + 2 * PLANE) = make_uint2(__builtin_amdgcn_perm(ll[1], ll[0], 0x07060302u), __builtin_amdgcn_perm(ll[3], ll[2], 0x07060302u));
        }
        __syncthreads();
        fetch(tile + gridDim.x);                 // in flight during all G groups of this tile

#pragma unroll 1
        for (int g = 0; g < G; ++g) {
            f32x4 acc[RT][2];
#pragma unroll
            for (int i = 0; i < RT; ++i) { acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[i][1] = acc[i][0]; }
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                bf16x8 b[2][3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) { b[0][pl] = bn[0][pl]; b[1][pl] = bn[1][pl]; }
                {   // next step: (g, t + 1), or the first of the next group (after the last group: group 0 again, for the next tile)
                    const int gn = t + 1 < KT ? g : (g + 1 < G ? g + 1 : 0), tn = t + 1 < KT ? t + 1 : 0;
                    wfrag(gn, tn, 2 * sp, bn[0]); wfrag(gn, tn, 2 * sp + 1, bn[1]);
                }
                bf16x8 a[RT][3];
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) a[i][pl] = *reinterpret_cast<const bf16x8*>(Xp + pl * PLANE + (rh * (FD_ROWS / 2) + i * 16 + fr) * SB + t * 32 + fg * 8);
                // smallest cross products first; term-major: consecutive MFMAs are independent
#if defined(LRD_ABL) && (LRD_ABL & 1)
#define FD_T(PA, PB) _Pragma("unroll") for (int i = 0; i < RT; ++i) asm volatile("" :: "v"(a[i][PA]), "v"(b[0][PB]), "v"(b[1][PB]));
#else
#define FD_T(PA, PB) _Pragma("unroll") for (int i = 0; i < RT; ++i) { \
        acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][PA], b[0][PB], acc[i][0], 0, 0, 0); \
        acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][PA], b[1][PB], acc[i][1], 0, 0, 0); }
#endif
                FD_T(2, 0) FD_T(0, 2) FD_T(1, 1) FD_T(1, 0) FD_T(0, 1) FD_T(0, 0)
#undef FD_T
            }
            // epilogue: D[row = 16 i + 4 fg + q][col = fr] of strips 2 sp, 2 sp + 1
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int strip = 2 * sp + s2;
                const int col = g * FD_J + strip * 16 + fr;
                const float bs = b1[col], ww = w2[col];
                float* rp = red + (buf * 8 + strip) * FD_ROWS + rh * (FD_ROWS / 2);
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#if defined(LRD_ABL) && (LRD_ABL & 2)
                        const float s = acc[i][s2][q] + bs + ww;       // timing experiment: no row sums
#else
                        const float s = fd_row_sum16(fmaxf(acc[i][s2][q] + bs, 0.f) * ww);
#endif
                        if (fr == ((i * 4 + q) & 15)) rp[i * 16 + fg * 4 + q] = s;    // spread the stores over the row's lanes
                    }
            }
            __syncthreads();
            if (tid < FD_ROWS && r0 + tid < R) {
                const float* rr = red + buf * 8 * FD_ROWS + tid;
                float s = rr[0];
#pragma unroll
                for (int w = 1; w < 8; ++w) s += rr[w * FD_ROWS];
                out[(r0 + tid) * ldo + g] = s + b2[g];
            }
            buf ^= 1;
        }
    }
}

template <int K>
static int launch_lrd_bx(long R, int G, const float* X, long ldx, const void* Wq, const float* b1, const float* w2, const float* b2, float* out,
                         long ldo, hipStream_t st) {
    constexpr int FD_ROWS = K <= 128 ? 128 : 64;          // K = 256: 128 rows of planes (203 KB) do not fit the LDS
    const size_t lds = (size_t)3 * FD_ROWS * (K + 8) * 2 + (size_t)2 * 8 * FD_ROWS * sizeof(float);
    const long ntiles = (R + FD_ROWS - 1) / FD_ROWS;
    auto kern = linear_relu_dot_bx_kernel<K, FD_ROWS>;
    static int blocks_resident = 0;
    if (blocks_resident == 0) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 512, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        blocks_resident = etch_cu_count() * per_cu;
    }
    long blocks = blocks_resident;
    if (blocks > ntiles) blocks = ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, st, R, G, X, ldx, reinterpret_cast<const bf16x8*>(Wq), b1, w2, b2, out, ldo);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

// ------------------------------------------------------------------------------------------------
// Many groups (the confidence head: G = 86): weight-stationary.  The bx kernel above streams all 8.4 MB of split weights from L2 for every
// 128-row tile (10.5 GB per launch: 1.3 ms of L2 time that its 8 waves do not overlap with 1.2 ms of matrix work).  Here a workgroup owns a
// COLUMN BLOCK of two groups (256 hidden columns) and a row block: wave w keeps the fragments of strips 2 w, 2 w + 1 of the block (16 strips)
// in registers for the whole launch (K = 128: 96 VGPRs) and walks its row block's 128-row tiles; X (82 MB, re-read by the 43 column blocks
// out of L2 / MALL) is split into LDS planes per tile as before.  The product is formed TRANSPOSED (hidden units as accumulator rows, the
// tile's rows as columns): a lane holds 4 hidden values of ONE row, so bias / ReLU / w2 and the sum over the strip's hidden units are in-lane
// adds plus two lane-group swaps per row tile instead of four DPP reductions per accumulator register.
// SPW = strips per wave: 2 (a column block = two groups, 16 strips over the 8 waves) or 1 (a column block = one group: G = 1, the direction
// head's tail, whose 49 KB of weights stream from L2 for nothing otherwise and whose 9.6 M rows make the epilogue the larger half of the work).
// F16 (round 5, etch_linear_relu_dot_f16): the same kernel on v_mfma_f32_16x16x32_f16 with TWO fp16 planes per operand and three cross terms -- half
// the matrix instructions of the bf16 split (the confidence head's launch is matrix-bound: pipe busy 0.61).  X has no known scale (Point-Transformer
// features), so every ROW is multiplied by the power of two that puts its maximum into [8, 16) (its 128 channels sit in 32 consecutive lanes when the
// tile is staged: a DPP maximum, no barrier) and the power leaves again in the epilogue's fmaf with the bias; Wq arrives as planes of W with every row (hidden unit) times its own power
// of two, `wsc` holds the inverse powers (an operand scaled per matrix only would leave rows far below the matrix maximum with the planes' absolute floor).
template <int K, int SPW, bool F16 = false>
__global__ void __launch_bounds__(512) linear_relu_dot_ws_kernel(long R, int G, const float* __restrict__ X, long ldx,
                                                                 const bf16x8* __restrict__ Wq, const float* __restrict__ b1,
                                                                 const float* __restrict__ w2, const float* __restrict__ b2,
                                                                 float* __restrict__ out, long ldo, unsigned* __restrict__ ctr, const float* __restrict__ wsc) {
    __shared__ unsigned s_grab[2];
    // 64-row tiles, planes and reduction table double-buffered: ONE barrier per tile, and the next tile's split + LDS stores sit between the two
    // halves of this tile's MFMA stream (VALU work beside the bf16 matrix cores is free; beside a barrier it is not)
    constexpr int FD_ROWS = 64, SB = K + 8, KT = K / 32, PLANE = FD_ROWS * SB, RT = FD_ROWS / 16;
    constexpr int NPL = F16 ? 2 : 3;              // operand planes
    constexpr int GPB = SPW;                      // groups per column block
    constexpr int WPG = 8 / GPB;                  // waves per group
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned short* Xp = reinterpret_cast<unsigned short*>(lds);            // [2 buffers][NPL][FD_ROWS][SB]
    float* red = lds + 2 * NPL * PLANE / 2;                                  // [2 buffers][8 waves][FD_ROWS]
    float* rsc = red + 2 * 8 * FD_ROWS;                                      // F16: [2 buffers][FD_ROWS] the rows' epilogue factors 2^-kx
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int cb = blockIdx.x, rb = blockIdx.y;
    // this wave's strips: SPW w .. of the column block = strips (SPW w) % 8 .. of group GPB cb + w / WPG (clamped: an odd G leaves the last
    // block's second group empty -- its waves recompute the last group and never store)
    const int gq = GPB * cb + wave / WPG;
    const int g = gq < G ? gq : G - 1;
    const int s0 = (SPW * wave) & 7;
    bf16x8 wf[KT][SPW][NPL];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int s2 = 0; s2 < SPW; ++s2)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) wf[t][s2][pl] = Wq[((((long)g * KT + t) * 8 + s0 + s2) * NPL + pl) * 64 + lane];
    // hidden units of this lane: strip s0 + s2, rows 4 fg .. 4 fg + 3
    float4 bs[SPW], ww[SPW];
#pragma unroll
    for (int s2 = 0; s2 < SPW; ++s2) {
        bs[s2] = *reinterpret_cast<const float4*>(b1 + g * FD_J + (s0 + s2) * 16 + 4 * fg);
        ww[s2] = *reinterpret_cast<const float4*>(w2 + g * FD_J + (s0 + s2) * 16 + 4 * fg);
        if constexpr (F16) {
            // every hidden unit's weight row carries its own power of two 2^kw; wsc = 2^-kw.  relu(2^-kw a + b) w2 = relu(a + 2^kw b) (2^-kw w2): the powers
            // move into the bias and w2 once per workgroup (exact), the epilogue stays as it is
            const float4 k4 = *reinterpret_cast<const float4*>(wsc + g * FD_J + (s0 + s2) * 16 + 4 * fg);
            bs[s2] = make_float4(bs[s2].x / k4.x, bs[s2].y / k4.y, bs[s2].z / k4.z, bs[s2].w / k4.w);
            ww[s2] = make_float4(ww[s2].x * k4.x, ww[s2].y * k4.y, ww[s2].z * k4.z, ww[s2].w * k4.w);
        }
    }
    const float b2a = b2[GPB * cb < G ? GPB * cb : G - 1], b2b = b2[GPB * cb + 1 < G ? GPB * cb + 1 : G - 1];

    constexpr int C4 = K / 4;
    constexpr int XL = (FD_ROWS * C4 + 511) / 512;
    // row tiles of the column block: the first three of a workgroup are static (rb, rb + nrb, rb + 2 nrb), every further one comes from the column
    // block's work counter (common.h: etch_work_counter_slot) -- asked for two tiles ahead, handed over through LDS (double-buffered: the loop has one
    // barrier per tile); without a counter the round-robin continues
    const long row_hi = R, ntiles = (R + FD_ROWS - 1) / FD_ROWS;
    const int nrb = gridDim.y;
    long t0 = rb, t1 = t0 + nrb, t2 = t1 + nrb;
    float4 xn[XL];
    auto fetch = [&](long r0) {
#pragma unroll
        for (int h = 0; h < XL; ++h) {
            const int e = tid + 512 * h;
            const int row = e / C4, c = (e - row * C4) * 4;
            xn[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < FD_ROWS * C4 && r0 + row < row_hi) xn[h] = *reinterpret_cast<const float4*>(X + (r0 + row) * ldx + c);   // rows past the block are zero
        }
    };
    auto stage = [&](unsigned short* P, float* rs) {
#pragma unroll
        for (int h = 0; h < XL; ++h) {
            const int e = tid + 512 * h;
            if (e >= FD_ROWS * C4) continue;
            const int row = e / C4, c = (e - row * C4) * 4;
            if constexpr (F16) {
                static_assert(C4 == 8 || C4 == 16 || C4 == 32, "a row's float4s sit in one aligned lane group");
                const int kx = etch_scale_exp(etch_group_max<C4>(etch_max4abs(xn[h], 0.f)));      // the row's power of two
                const float sx = ldexpf(1.0f, kx);
                uint2 ph, pl;
                split2h_pack4(make_float4(xn[h].x * sx, xn[h].y * sx, xn[h].z * sx, xn[h].w * sx), ph, pl);
                unsigned short* d = P + row * SB + c;
                *reinterpret_cast<uint2*>(d) = ph;
                *reinterpret_cast<uint2*>(d + PLANE) = pl;
                if (c == 0) rs[row] = ldexpf(1.0f, -kx);
                continue;
            }
            const float v[4] = {xn[h].x, xn[h].y, xn[h].z, xn[h].w};
            unsigned hh[4], mm[4], ll[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fd_split(v[i], hh[i], mm[i], ll[i]);
            unsigned short* d = P + row * SB + c;
            *reinterpret_cast<uint2*>(d) = make_uint2(__builtin_amdgcn_perm(hh[1], hh[0], 0x07060302u), __builtin_amdgcn_perm(hh[3], hh[2], 0x07060302u));
            *reinterpret_cast<uint2*>(d + PLANE) = make_uint2(__builtin_amdgcn_perm(mm[1], mm[0], 0x07060302u), __builtin_amdgcn_perm(mm[3], mm[2], 0x07060302u));
            *reinterpret_cast<uint2*>(d + 2 * PLANE) = make_uint2(__builtin_amdgcn_perm(ll[1], ll[0], 0x07060302u), __builtin_amdgcn_perm(ll[3], ll[2], 0x07060302u));
        }
    };
    // two row tiles: D[hidden 4 fg + q of strip s2][row 16 i + fr] -> sums over the wave's hidden units into rt[]
    auto half_tile = [&](const unsigned short* P, int i0, float* rt, const float* rs) {
        f32x4 acc[2][SPW];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s2 = 0; s2 < SPW; ++s2) acc[i][s2] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            bf16x8 x[2][NPL];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) x[i][pl] = *reinterpret_cast<const bf16x8*>(P + pl * PLANE + ((i0 + i) * 16 + fr) * SB + t * 32 + fg * 8);
            if constexpr (F16) {
#define FD_T(PA, PB) _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int s2 = 0; s2 < SPW; ++s2) \
        acc[i][s2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[t][s2][PA]), __builtin_bit_cast(f16x8, x[i][PB]), acc[i][s2], 0, 0, 0);
                FD_T(1, 0) FD_T(0, 1) FD_T(0, 0)
#undef FD_T
                continue;
            }
#define FD_T(PA, PB) _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int s2 = 0; s2 < SPW; ++s2) \
        acc[i][s2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][s2][PA % NPL], x[i][PB % NPL], acc[i][s2], 0, 0, 0);
            FD_T(2, 0) FD_T(0, 2) FD_T(1, 1) FD_T(1, 0) FD_T(0, 1) FD_T(0, 0)
#undef FD_T
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float tsum = 0.f;
            const float sc = F16 ? rs[(i0 + i) * 16 + fr] : 1.0f;       // this lane's row: the power of two of its staging and of W (exact)
#pragma unroll
            for (int s2 = 0; s2 < SPW; ++s2) {
                if constexpr (F16) {
                    tsum += fmaxf(fmaf(acc[i][s2][0], sc, bs[s2].x), 0.f) * ww[s2].x; tsum += fmaxf(fmaf(acc[i][s2][1], sc, bs[s2].y), 0.f) * ww[s2].y;
                    tsum += fmaxf(fmaf(acc[i][s2][2], sc, bs[s2].z), 0.f) * ww[s2].z; tsum += fmaxf(fmaf(acc[i][s2][3], sc, bs[s2].w), 0.f) * ww[s2].w;
                    continue;
                }
                tsum += fmaxf(acc[i][s2][0] + bs[s2].x, 0.f) * ww[s2].x; tsum += fmaxf(acc[i][s2][1] + bs[s2].y, 0.f) * ww[s2].y;
                tsum += fmaxf(acc[i][s2][2] + bs[s2].z, 0.f) * ww[s2].z; tsum += fmaxf(acc[i][s2][3] + bs[s2].w, 0.f) * ww[s2].w;
            }
            // sum over the 4 lane groups (the other hidden rows of the strips): v_permlane16_swap / v_permlane32_swap
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(tsum), __float_as_uint(tsum), false, false);
            tsum = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            r = __builtin_amdgcn_permlane32_swap(__float_as_uint(tsum), __float_as_uint(tsum), false, false);
            tsum = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            if (fg == 0) rt[(i0 + i) * 16 + fr] = tsum;
        }
    };
    if (t0 >= ntiles) return;
    fetch(t0 * FD_ROWS);
    stage(Xp, rsc);
    fetch(t1 * FD_ROWS);                          // (tiles past the end: rows >= R are fetched as zeros)
    __syncthreads();
    int buf = 0;
    for (; t0 < ntiles; buf ^= 1) {
        const long r0 = t0 * FD_ROWS;
        const unsigned short* P = Xp + buf * NPL * PLANE;
        float* rtab = red + buf * 8 * FD_ROWS;
        unsigned grabbed = 0u;
        if (ctr && tid == 0) grabbed = __hip_atomic_fetch_add(ctr + (cb & 63), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        half_tile(P, 0, rtab + wave * FD_ROWS, rsc + buf * FD_ROWS);
        stage(Xp + (buf ^ 1) * NPL * PLANE, rsc + (buf ^ 1) * FD_ROWS);       // the next tile (its buffer's last readers finished before the previous barrier)
        fetch(t2 * FD_ROWS);
        half_tile(P, 2, rtab + wave * FD_ROWS, rsc + buf * FD_ROWS);
        if (ctr && tid == 0) s_grab[buf] = grabbed;
        __syncthreads();
        t0 = t1; t1 = t2;
        t2 = ctr ? 3L * nrb + __builtin_amdgcn_readfirstlane((int)s_grab[buf]) : t2 + nrb;
        if (tid < GPB * FD_ROWS) {               // thread = (group of the block, row): the group's waves in wave order
            const int gi = tid / FD_ROWS, row = tid % FD_ROWS;
            if (GPB * cb + gi < G && r0 + row < row_hi) {
                const float* rr = rtab + gi * WPG * FD_ROWS + row;
                float s_ = rr[0];
#pragma unroll
                for (int w = 1; w < WPG; ++w) s_ += rr[w * FD_ROWS];
                out[(r0 + row) * ldo + GPB * cb + gi] = s_ + (gi == 0 ? b2a : b2b);
            }
        }
        // (the table read here is rewritten two tiles on, behind the next tile's barrier)
    }
}

template <int K, int SPW, bool F16 = false>
static int launch_lrd_ws(long R, int G, const float* X, long ldx, const void* Wq, const float* b1, const float* w2, const float* b2, float* out,
                         long ldo, hipStream_t st, const float* wsc = nullptr) {
    constexpr int FD_ROWS = 64;
    const size_t lds = (size_t)2 * (F16 ? 2 : 3) * FD_ROWS * (K + 8) * 2 + (size_t)2 * 8 * FD_ROWS * sizeof(float) + (size_t)2 * FD_ROWS * sizeof(float);
    auto kern = linear_relu_dot_ws_kernel<K, SPW, F16>;
    static int per_cu = 0;
    if (per_cu == 0) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)kern, 512, lds) != hipSuccess || n < 1) n = 1;
        per_cu = n;
    }
    const int ncb = (G + SPW - 1) / SPW;
    int nrb = etch_cu_count() * per_cu / ncb;    // resident workgroups: row blocks per column block
    if (nrb < 1) nrb = 1;
    const long ntiles = (R + FD_ROWS - 1) / FD_ROWS;
    if (nrb > ntiles) nrb = (int)ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)ncb, (unsigned)nrb), dim3(512), lds, st, R, G, X, ldx, reinterpret_cast<const bf16x8*>(Wq), b1, w2, b2, out, ldo,
                       // one column block only (the direction tail).  With many column blocks (the confidence head: 43) the workgroups of a row position
                       // walk the same X tiles in lock-step and share them through L2; per-block counters let them drift apart (measured: 706 against
                       // 744 scans/s, and unstable) -- those launches keep the static round-robin
                       ncb == 1 ? etch_work_counter_slot(st) : nullptr, wsc);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

// The weight-stationary kernel on the fp16 matrix cores (two planes per operand).  Wqh = ops.lrd_weight_split_f16: lrd_weight_split's order with two fp16
// planes of W, every ROW (hidden unit) times its own power of two; wsc (G * J floats) = the inverse powers.  Shapes the weight-stationary kernel covers:
// K in {32, 64, 128}, G = 1 or G >= 8, b1 / w2 / wsc 16-byte aligned; else ETCH_EUNSUPPORTED.
extern "C" int etch_linear_relu_dot_f16(long R, int K, int G, int J, const float* X, long ldx, const void* Wqh, const float* wsc, const float* b1, const float* w2,
                                        const float* b2, float* out, long ldo, void* stream) {
    if (R <= 0 || G <= 0) return ETCH_OK;
    if (!X || !Wqh || !wsc || !b1 || !w2 || !b2 || !out) return ETCH_EINVAL;
    if ((ldx & 3) || ((uintptr_t)X & 15) || ((uintptr_t)Wqh & 15) || ((uintptr_t)wsc & 15)) return ETCH_EINVAL;
    if (J != FD_J || ((((uintptr_t)b1 | (uintptr_t)w2) & 15) != 0)) return ETCH_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (G >= 8) {
        if (K == 32) return launch_lrd_ws<32, 2, true>(R, G, X, ldx, Wqh, b1, w2, b2, out, ldo, st, wsc);
        if (K == 64) return launch_lrd_ws<64, 2, true>(R, G, X, ldx, Wqh, b1, w2, b2, out, ldo, st, wsc);
        if (K == 128) return launch_lrd_ws<128, 2, true>(R, G, X, ldx, Wqh, b1, w2, b2, out, ldo, st, wsc);
    }
    if (G == 1) {
        if (K == 32) return launch_lrd_ws<32, 1, true>(R, G, X, ldx, Wqh, b1, w2, b2, out, ldo, st, wsc);
        if (K == 64) return launch_lrd_ws<64, 1, true>(R, G, X, ldx, Wqh, b1, w2, b2, out, ldo, st, wsc);
        if (K == 128) return launch_lrd_ws<128, 1, true>(R, G, X, ldx, Wqh, b1, w2, b2, out, ldo, st, wsc);
    }
    return ETCH_EUNSUPPORTED;
}

// Wq = ops.lrd_weight_split: [g][K step of 32][strip of 16 hidden columns][plane hi / mid / lo][lane = 16 * (k / 8) + column][8 bf16]
extern "C" int etch_linear_relu_dot_split(long R, int K, int G, int J, const float* X, long ldx, const void* Wq, const float* b1, const float* w2,
                                          const float* b2, float* out, long ldo, void* stream) {
    if (R <= 0 || G <= 0) return ETCH_OK;
    if (!X || !Wq || !b1 || !w2 || !b2 || !out) return ETCH_EINVAL;
    if ((ldx & 3) || ((uintptr_t)X & 15) || ((uintptr_t)Wq & 15)) return ETCH_EINVAL;
    if (J != FD_J) return ETCH_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
#ifndef LRD_WS_MIN_G
#define LRD_WS_MIN_G 8        // from this many groups on the weight-stationary kernel with two groups per workgroup
#endif
    // (the choice depends on the layer's shape only, never on the row count: a scan's result may not depend on its batch neighbours, bit for bit)
    // the weight-stationary kernel reads b1 and w2 with 16-byte loads: a bias / w2 VIEW at an offset that is not a multiple of 16 bytes goes to the
    // streaming kernel below (scalar reads) instead of faulting
    const bool ws_aligned = (((uintptr_t)b1 | (uintptr_t)w2) & 15) == 0;
    if (K <= 128 && ws_aligned) {
        if (G >= LRD_WS_MIN_G) {
            if (K == 32) return launch_lrd_ws<32, 2>(R, G, X, ldx, Wq, b1, w2, b2, out, ldo, st);
            if (K == 64) return launch_lrd_ws<64, 2>(R, G, X, ldx, Wq, b1, w2, b2, out, ldo, st);
            if (K == 128) return launch_lrd_ws<128, 2>(R, G, X, ldx, Wq, b1, w2, b2, out, ldo, st);
        }
        if (G == 1) {
            if (K == 32) return launch_lrd_ws<32, 1>(R, G, X, ldx, Wq, b1, w2, b2, out, ldo, st);
            if (K == 64) return launch_lrd_ws<64, 1>(R, G, X, ldx, Wq, b1, w2, b2, out, ldo, st);
            if (K == 128) return launch_lrd_ws<128, 1>(R, G, X, ldx, Wq, b1, w2, b2, out, ldo, st);
        }
    }
    if (K == 32) return launch_lrd_bx<32>(R, G, X, ldx, Wq, b1, w2, b2, out, ldo, st);
    if (K == 64) return launch_lrd_bx<64>(R, G, X, ldx, Wq, b1, w2, b2, out, ldo, st);
    if (K == 128) return launch_lrd_bx<128>(R, G, X, ldx, Wq, b1, w2, b2, out, ldo, st);
    if (K == 256) return launch_lrd_bx<256>(R, G, X, ldx, Wq, b1, w2, b2, out, ldo, st);
    return ETCH_EUNSUPPORTED;
}

template <int K, int FD_ROWS>
static int launch_lrd(long R, int G, const float* X, long ldx, const float* W, long ldw, const float* Wp, const float* b1,
                      const float* w2, const float* b2, float* out, long ldo, hipStream_t st) {
    const size_t lds = ((size_t)FD_ROWS * (K + FD_PAD) + 2 * 8 * FD_ROWS) * sizeof(float);
    const long ntiles = (R + FD_ROWS - 1) / FD_ROWS;
    auto go = [&](auto kern, const float* Wk, long ldk) -> int {
        // per kernel instantiation, once per process (never inside a HIP-graph capture: the first call of a shape happens in the warm-up
        // passes): the dynamic-LDS attribute, the resident workgroups per CU and the CU count of the current device
        static int blocks_resident = 0;
        if (blocks_resident == 0) {
            if (lds > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return (int)e;
            }
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 512, lds) != hipSuccess || per_cu < 1) per_cu = 1;
            blocks_resident = etch_cu_count() * per_cu;
        }
        long blocks = blocks_resident;           // persistent: every resident workgroup walks the tiles with this stride
        if (blocks > ntiles) blocks = ntiles;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, st, R, G, X, ldx, Wk, ldk, b1, w2, b2, out, ldo);
        return ETCH_OK;
    };
    const int rc = Wp ? go(linear_relu_dot_kernel<K, true, FD_ROWS>, Wp, 0L) : go(linear_relu_dot_kernel<K, false, FD_ROWS>, W, ldw);
    if (rc != ETCH_OK) return rc;
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

extern "C" int etch_linear_relu_dot(long R, int K, int G, int J, const float* X, long ldx, const float* W, long ldw, const float* Wp,
                                    const float* b1, const float* w2, const float* b2, float* out, long ldo, void* stream) {
    if (R <= 0 || G <= 0) return ETCH_OK;
    if (!X || !(W || Wp) || !b1 || !w2 || !b2 || !out) return ETCH_EINVAL;
    if ((ldx & 3) || ((uintptr_t)X & 15) || (!Wp && ((ldw & 3) || ((uintptr_t)W & 15))) || (Wp && ((uintptr_t)Wp & 15))) return ETCH_EINVAL;
    if (J != FD_J) return ETCH_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    // row-tile height: 128 for K = 64 (61 KB of LDS: 2 workgroups per CU); 64 for K = 128 (47 KB: 3 per CU instead of one 94 KB
    // workgroup -- 4.07 -> 3.90 ms on the confidence head although every workgroup streams the weights)
    if (K == 64) return launch_lrd<64, 128>(R, G, X, ldx, W, ldw, Wp, b1, w2, b2, out, ldo, st);
    if (K == 128) return launch_lrd<128, 64>(R, G, X, ldx, W, ldw, Wp, b1, w2, b2, out, ldo, st);
    if (K == 32) return launch_lrd<32, 128>(R, G, X, ldx, W, ldw, Wp, b1, w2, b2, out, ldo, st);      // direction tail of encoder depth 1
    if (K == 256) return launch_lrd<256, 32>(R, G, X, ldx, W, ldw, Wp, b1, w2, b2, out, ldo, st);     // ... of depth 4
    return ETCH_EUNSUPPORTED;
}
