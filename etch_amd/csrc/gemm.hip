// Generic fp32 NT GEMM with fused epilogue on the gfx950 matrix cores (v_mfma_f32_16x16x4_f32):
//     Y[r, o] = epi( sum_k X[rowmap(r), k] * W[o, k] )        r < R, o < O, k < K
// Serves every dense per-point layer of the hot path (SURVEY 8 rows a9 skip conv, a13 linears,
// a15 linears / heads): torch.nn.Linear / Conv1d(k=1) / Conv2d(1x1) with eval-mode BatchNorm folded
// into (scale, shift).  fp32 inputs, fp32 accumulate: the f32 MFMA is an exact fmaf chain.
//
// Tiling: 256 threads = 4 waves, block tile 64 rows x 64 outputs, K step 32 staged through LDS
// with 16-byte loads; each wave owns a 32x32 sub-tile (2x2 MFMA tiles).  K is consumed in an
// interleaved order (lane group g = lane>>4 takes k = 16t + 4g + s in MFMA step s) so that one
// ds_read_b128 per operand feeds four MFMAs; A and B use the same order, so the sum is unchanged.
#include "common.h"
#include "split_bf16.h"
#include <cstdlib>
#include <cstring>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
    int R, K, O;
    const float* X; long ldx; const int* row_idx; int grp, p_in, p_out;
    const float* W; long ldw;
    const float* bias; const float* scale; const float* shift;
    int act;            // 0 none, 1 relu, 2 leaky_relu(0.01)
    const float* res; long ldr; int res_mode;   // 0 none, 1 add before act, 2 add after act
    float* Y; long ldy;
};

#define G_BM 64
#define G_BN 64
#define G_BK 32
#define G_LD 40   // padded LDS row (floats): 160 B keeps 16-B alignment and makes the ds_read_b128 fragment reads conflict-free (36 was 2-way)

template <bool VEC>
__device__ __forceinline__ float4 ld4_guard(const float* row, int k, int K, bool row_ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (VEC) {
        if (row_ok && k < K) v = *reinterpret_cast<const float4*>(row + k);   // K % 4 == 0, 16-B aligned rows
    } else if (row_ok) {
        if (k < K) v.x = row[k];
        if (k + 1 < K) v.y = row[k + 1];
        if (k + 2 < K) v.z = row[k + 2];
        if (k + 3 < K) v.w = row[k + 3];
    }
    return v;
}

template <bool VECX, bool VECW, int BN, int BM = 64, int WAVES = 4>   // block tile BM rows x BN outputs, WAVES waves (2 along M)
__global__ void __launch_bounds__(WAVES * 64) gemm_nt_kernel(GemmArgs a) {
    constexpr int THREADS = WAVES * 64;
    constexpr int NWN = WAVES / 2;             // waves along the output dimension
    constexpr int WNS = BN / NWN;              // outputs per wave
    constexpr int NJ = WNS / 16;               // 16-wide output tiles per wave
    constexpr int MI = BM / 32;                // 16-high row tiles per wave (wave tile = BM/2 rows)
    constexpr int RPP = THREADS / 8;           // rows staged per pass (8 float4 per 32-float row)
    constexpr int WL = BN / RPP;               // W float4 staging loads per thread
    constexpr int XL = BM / RPP;               // X float4 staging loads per thread
    static_assert(WNS % 16 == 0 && BN % RPP == 0 && BM % RPP == 0, "tile / thread geometry");
    __shared__ __attribute__((aligned(16))) float Xs[BM * G_LD];
    __shared__ __attribute__((aligned(16))) float Ws[BN * G_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * BM, o0 = blockIdx.y * BN;
    const int wm = wave / NWN, wn = wave % NWN;
    // staging coordinates: thread -> (row = tid/8 [+RPP*h], float4 column = tid%8)
    const int srow = tid >> 3, sc4 = (tid & 7) * 4;
    const float* xrow[XL];
    bool xok[XL];
    const float* wrow[WL];
    bool wok[WL];
#pragma unroll
    for (int h = 0; h < XL; ++h) {
        const int r = r0 + srow + RPP * h;
        xok[h] = r < a.R;
        long src = 0;
        if (xok[h]) {
            src = r;
            if (a.row_idx) {
                const int q = r / a.grp;
                src = ((long)(q / a.p_out) * a.p_in + a.row_idx[q]) * a.grp + (r - q * a.grp);
            }
        }
        xrow[h] = a.X + src * a.ldx;
    }
#pragma unroll
    for (int h = 0; h < WL; ++h) {
        const int o = o0 + srow + RPP * h;
        wok[h] = o < a.O;
        wrow[h] = a.W + (long)(wok[h] ? o : 0) * a.ldw;
    }
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    float4 xv[XL], wv[WL];
#pragma unroll
    for (int h = 0; h < XL; ++h) xv[h] = ld4_guard<VECX>(xrow[h], sc4, a.K, xok[h]);
#pragma unroll
    for (int h = 0; h < WL; ++h) wv[h] = ld4_guard<VECW>(wrow[h], sc4, a.K, wok[h]);
    for (int k0 = 0; k0 < a.K; k0 += G_BK) {
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int h = 0; h < XL; ++h) *reinterpret_cast<float4*>(&Xs[(srow + RPP * h) * G_LD + sc4]) = xv[h];
#pragma unroll
        for (int h = 0; h < WL; ++h) *reinterpret_cast<float4*>(&Ws[(srow + RPP * h) * G_LD + sc4]) = wv[h];
        __syncthreads();
        if (k0 + G_BK < a.K) {      // register prefetch of the next K tile: in flight while this tile is multiplied
#pragma unroll
            for (int h = 0; h < XL; ++h) xv[h] = ld4_guard<VECX>(xrow[h], k0 + G_BK + sc4, a.K, xok[h]);
#pragma unroll
            for (int h = 0; h < WL; ++h) wv[h] = ld4_guard<VECW>(wrow[h], k0 + G_BK + sc4, a.K, wok[h]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float4 af[MI], bf[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const float4*>(&Xs[(wm * (BM / 2) + i * 16 + fr) * G_LD + t * 16 + fg * 4]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[j] = *reinterpret_cast<const float4*>(&Ws[(wn * WNS + j * 16 + fr) * G_LD + t * 16 + fg * 4]);
            // k-slice outermost: consecutive MFMAs hit different accumulators (a 16x16x4 f32 MFMA has a 40-cycle
            // dependent latency vs a 32-cycle issue interval), so no accumulator is touched twice in a row
#define G_STEP(C)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j)        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].C, bf[j].C, acc[i][j], 0, 0, 0);
            G_STEP(x) G_STEP(y) G_STEP(z) G_STEP(w)
#undef G_STEP
        }
    }
    // epilogue: D[row = fg*4 + reg][col = fr]
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int o = o0 + wn * WNS + j * 16 + fr;
        if (o >= a.O) continue;
        const float bs = a.bias ? a.bias[o] : 0.f;
        const float sc = a.scale ? a.scale[o] : 1.f;
        const float sh = a.shift ? a.shift[o] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + wm * (BM / 2) + i * 16 + fg * 4 + q;
                if (r >= a.R) continue;
                float v = acc[i][j][q] + bs;
                if (a.scale) v = v * sc + sh;
                if (a.res_mode == 1) v += a.res[(long)r * a.ldr + o];
                if (a.act == 1) v = fmaxf(v, 0.f);
                else if (a.act == 2) v = v > 0.f ? v : 0.01f * v;
                if (a.res_mode == 2) v += a.res[(long)r * a.ldr + o];
                a.Y[(long)r * a.ldy + o] = v;
            }
    }
}

template <int BN, int BM = 64, int WAVES = 4>
static void launch_tiled(const GemmArgs& a, bool vx, bool vw, hipStream_t st) {
    dim3 grid((a.R + BM - 1) / BM, (a.O + BN - 1) / BN);
    if (vx && vw) hipLaunchKernelGGL((gemm_nt_kernel<true, true, BN, BM, WAVES>), grid, dim3(WAVES * 64), 0, st, a);
    else if (vx) hipLaunchKernelGGL((gemm_nt_kernel<true, false, BN, BM, WAVES>), grid, dim3(WAVES * 64), 0, st, a);
    else if (vw) hipLaunchKernelGGL((gemm_nt_kernel<false, true, BN, BM, WAVES>), grid, dim3(WAVES * 64), 0, st, a);
    else hipLaunchKernelGGL((gemm_nt_kernel<false, false, BN, BM, WAVES>), grid, dim3(WAVES * 64), 0, st, a);
}

// ------------------------------------------------------------------------------------------------
// Skinny-K variant for the per-token layers (K in {64,128}, O*K*4 <= 64 KiB): the whole weight matrix sits in LDS in
// MFMA fragment order for the lifetime of a persistent workgroup; every wave streams its own 16-row tiles straight
// from HBM into A fragments (each input byte is read exactly once, no LDS round trip, no inter-wave barrier in the
// loop) and owns all O/16 accumulators, so X is read once and Y written once: HBM-roofline for K = 64.
// ------------------------------------------------------------------------------------------------
template <int K, int NT>
__global__ void __launch_bounds__(256, 2) gemm_wres_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float Wl[];      // [K/16][NT][64 lanes][4]
    constexpr int KT = K / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    for (int e = tid; e < KT * NT * 64; e += 256) {
        const int l = e & 63, nt = (e >> 6) % NT, t = (e >> 6) / NT;
        const int o = nt * 16 + (l & 15);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (o < a.O) v = *reinterpret_cast<const float4*>(a.W + (long)o * a.ldw + t * 16 + (l >> 4) * 4);
        *reinterpret_cast<float4*>(&Wl[e * 4]) = v;
    }
    __syncthreads();
    const int ntiles = (a.R + 15) >> 4;
    float4 an[KT];
    auto load_tile = [&](int tile) {
        const int r = tile * 16 + fr;
        const bool ok = r < a.R;
        long src = 0;
        if (ok) {
            src = r;
            if (a.row_idx) {
                const int q = r / a.grp;
                src = ((long)(q / a.p_out) * a.p_in + a.row_idx[q]) * a.grp + (r - q * a.grp);
            }
        }
        const float* xr = a.X + src * a.ldx + fg * 4;
#pragma unroll
        for (int t = 0; t < KT; ++t) an[t] = ok ? *reinterpret_cast<const float4*>(xr + t * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    int tile = blockIdx.x * 4 + wave;
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += gridDim.x * 4) {
        float4 ac[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) ac[t] = an[t];
        const int nxt = tile + gridDim.x * 4;
        if (nxt < ntiles) load_tile(nxt);                   // prefetch the next tile's fragments
        const int r0 = tile * 16;
        constexpr int NG = NT < 4 ? NT : 4;                 // output tiles per pass: bounds the live B fragments / accumulators
#pragma unroll 1
        for (int ng = 0; ng < NT; ng += NG) {
            f32x4 acc[NG];
#pragma unroll
            for (int j = 0; j < NG; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < KT; ++t) {
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    const int nt = ng + j < NT ? ng + j : NT - 1;
                    const float4 b = *reinterpret_cast<const float4*>(&Wl[((t * NT + nt) * 64 + lane) * 4]);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[t].x, b.x, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[t].y, b.y, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[t].z, b.z, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[t].w, b.w, acc[j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                const int o = (ng + j) * 16 + fr;
                if (ng + j >= NT || o >= a.O) continue;
                const float bs = a.bias ? a.bias[o] : 0.f;
                const float sc = a.scale ? a.scale[o] : 1.f;
                const float sh = a.shift ? a.shift[o] : 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = r0 + fg * 4 + q;
                    if (r >= a.R) continue;
                    float v = acc[j][q] + bs;
                    if (a.scale) v = v * sc + sh;
                    if (a.res_mode == 1) v += a.res[(long)r * a.ldr + o];
                    if (a.act == 1) v = fmaxf(v, 0.f);
                    else if (a.act == 2) v = v > 0.f ? v : 0.01f * v;
                    if (a.res_mode == 2) v += a.res[(long)r * a.ldr + o];
                    a.Y[(long)r * a.ldy + o] = v;
                }
            }
        }
    }
}

template <int K, int NT>
static int launch_wres(const GemmArgs& a, hipStream_t st) {
    const size_t lds = (size_t)(K / 16) * NT * 64 * 4 * sizeof(float);
    auto kern = gemm_wres_kernel<K, NT>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const int ntiles = (a.R + 15) / 16;
    // NOT sized to the chip: the kernel usually runs next to kernels of other HIP streams (heads on their own streams, the
    // 2-deep pipeline), where a grid of exactly CUs x occupancy blocks with a static share of the rows each ran 6x longer
    // (blocks that do not fit immediately start late and finish late).  8 row tiles per wave amortise the weight staging;
    // the dispatcher balances the rest.
    int blocks = (ntiles + 31) / 32;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, st, a);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

// K == 1 (the skip conv of the first EPN block: one input channel): Y[r, o] = epi(x[rowmap(r)] * W[o, 0]) is an outer product,
// i.e. a pure store stream (614 MB at 4.8 M rows x 32 outputs); thread = (row, 4 outputs), 16-byte stores.
__global__ void __launch_bounds__(256) linear_k1_kernel(GemmArgs a) {
    const int o4n = (a.O + 3) >> 2;
    const long total = (long)a.R * o4n;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / o4n;
        const int o0 = (int)(e - r * o4n) * 4;
        long src = r;
        if (a.row_idx) {
            const long q = r / a.grp;
            src = ((q / a.p_out) * a.p_in + a.row_idx[q]) * a.grp + (r - q * a.grp);
        }
        const float x = a.X[src * a.ldx];
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int o = o0 + u;
            float t = 0.f;
            if (o < a.O) {
                t = x * a.W[(long)o * a.ldw] + (a.bias ? a.bias[o] : 0.f);
                if (a.scale) t = t * a.scale[o] + a.shift[o];
                if (a.res_mode == 1) t += a.res[r * a.ldr + o];
                if (a.act == 1) t = fmaxf(t, 0.f);
                else if (a.act == 2) t = t > 0.f ? t : 0.01f * t;
                if (a.res_mode == 2) t += a.res[r * a.ldr + o];
            }
            v[u] = t;
        }
        float* y = a.Y + r * a.ldy + o0;
        if (o0 + 3 < a.O && !(a.ldy & 3) && !((uintptr_t)a.Y & 15)) *reinterpret_cast<float4*>(y) = make_float4(v[0], v[1], v[2], v[3]);
        else
            for (int u = 0; u < 4 && o0 + u < a.O; ++u) y[u] = v[u];
    }
}

// ------------------------------------------------------------------------------------------------
// Weight-stationary GEMM on the bf16 matrix cores for the layers with a small weight matrix and many rows (the Point-Transformer nets' linear
// layers at the two large levels, the encoder's skip convs): fp32 operands split EXACTLY into three bf16 values (8 + 8 + 8 mantissa bits), six
// cross products accumulated in fp32 by v_mfma_f32_16x16x32_bf16 -- the fp32 MFMA's error against fp64 (profiles/r03_bf16x3_split.txt).  The
// tiled kernel above reads X once per 64-column output tile and spends 37 - 50 % of the fp32 matrix rate on these shapes; here a workgroup
// keeps ALL of W in registers (split once per workgroup: wave = (strip group, row part), SPW strips of 16 output channels each), streams
// 64-row tiles of X through double-buffered LDS planes (split at staging, one barrier per tile) and writes Y once: X and Y cross HBM once.
// The product is formed transposed (output channels as accumulator rows): a lane holds 4 consecutive channels of one row = one 16-byte store.
// ------------------------------------------------------------------------------------------------
typedef short g_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void g_split(const float v, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(v);
    const float r = v - __uint_as_float(h & 0xffff0000u);
    m = __float_as_uint(r);
    l = __float_as_uint(r - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ void g_split8(const float4 v0, const float4 v1, g_bf16x8 (&o)[3]) {
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) g_split(v[i], h[i], m[i], l[i]);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define G_PK(a) (u32x4){__builtin_amdgcn_perm(a[1], a[0], 0x07060302u), __builtin_amdgcn_perm(a[3], a[2], 0x07060302u), \
                        __builtin_amdgcn_perm(a[5], a[4], 0x07060302u), __builtin_amdgcn_perm(a[7], a[6], 0x07060302u)}
    const u32x4 ph = G_PK(h), pm = G_PK(m), pl = G_PK(l);
#undef G_PK
    o[0] = __builtin_bit_cast(g_bf16x8, ph); o[1] = __builtin_bit_cast(g_bf16x8, pm); o[2] = __builtin_bit_cast(g_bf16x8, pl);
}

// O = 16 * SPW * (8 / RP) output channels; RP row parts (the 4 row tiles of a 64-row tile divided among them)
// F16 (round 5): two fp16 planes per operand on v_mfma_f32_16x16x32_f16, three cross terms.  Neither operand has a known scale here (Point-Transformer
// features, trained weights): every row of X is staged times the power of two that puts its maximum into [8, 16) (its K / 4 float4s sit in one aligned
// lane group: DPP maximum, no barrier), every ROW of W (output channel) likewise, once per workgroup; both powers leave in the epilogue's
// first fmaf (exact).  Opt-in (ETCH_LINEAR_SPLIT=f16): see the dispatch in etch_linear.
template <int K, int SPW, int RP, bool F16 = false>
__global__ void __launch_bounds__(512) gemm_ws_split_kernel(GemmArgs a, long rows_per_block) {
    constexpr int FD_ROWS = 64, SB = K + 8, KT = K / 32, PLANE = FD_ROWS * SB;
    constexpr int NPL = F16 ? 2 : 3;
    constexpr int SG = 8 / RP;                    // strip groups (waves along the output channels)
    constexpr int RTW = (FD_ROWS / 16) / RP;      // row tiles per wave
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned short* Xp = reinterpret_cast<unsigned short*>(lds);            // [2 buffers][NPL][FD_ROWS][SB]
    float* rsc = lds + 2 * NPL * PLANE / 2;                                 // F16: [2 buffers][FD_ROWS] the rows' factors 2^-kx
    float* wtab = rsc + 2 * FD_ROWS;                                        // F16: [8 waves][SPW][16] the weight rows' factors 2^-kw
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int sg = wave % SG, rp = wave / SG;
    // this wave's weight fragments: strips SPW sg .. + SPW - 1, rows (output channels) 16 strip + fr, k = 32 t + 8 fg + e
    g_bf16x8 wf[KT][SPW][NPL];
    float4 wk4[SPW];                              // F16: 2^-kw of this lane's four output channels (rows 4 fg .. of the strip)
    // epilogue constants: resident next to the weight fragments where both fit the 256-register budget of 2 waves/SIMD, re-read per row tile
    // (36 of them, L1 hits) where the fragments alone take 144 (K = 128, 3 strips: 22 spilled registers otherwise)
    constexpr bool EPI_REGS = KT * SPW * 12 <= 96;
    float4 bs[EPI_REGS ? SPW : 1], sc[EPI_REGS ? SPW : 1], sh[EPI_REGS ? SPW : 1];
    auto epi_load = [&](int o, float4& b_, float4& s_, float4& h_) {
        b_ = a.bias ? make_float4(a.bias[o + 4 * fg], a.bias[o + 4 * fg + 1], a.bias[o + 4 * fg + 2], a.bias[o + 4 * fg + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        s_ = a.scale ? make_float4(a.scale[o + 4 * fg], a.scale[o + 4 * fg + 1], a.scale[o + 4 * fg + 2], a.scale[o + 4 * fg + 3]) : make_float4(1.f, 1.f, 1.f, 1.f);
        h_ = a.shift ? make_float4(a.shift[o + 4 * fg], a.shift[o + 4 * fg + 1], a.shift[o + 4 * fg + 2], a.shift[o + 4 * fg + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
#pragma unroll
    for (int s2 = 0; s2 < SPW; ++s2) {
        const int o = 16 * (SPW * sg + s2);
        wk4[s2] = make_float4(1.f, 1.f, 1.f, 1.f);
        if constexpr (F16) {
            float4 w0[KT], w1[KT];
            float m = 0.f;
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                const float* p = a.W + (long)(o + fr) * a.ldw + 32 * t + 8 * fg;
                w0[t] = *reinterpret_cast<const float4*>(p); w1[t] = *reinterpret_cast<const float4*>(p + 4);
                m = etch_max4abs(w1[t], etch_max4abs(w0[t], m));
            }
            // the maximum of THIS lane's weight row (output channel o + fr): its K / 8 slices sit in the four lanes fr, fr + 16, fr + 32, fr + 48
            {
                typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
                u32x2_ r_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                m = fmaxf(__uint_as_float(r_[0]), __uint_as_float(r_[1]));
                r_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                m = fmaxf(__uint_as_float(r_[0]), __uint_as_float(r_[1]));
            }
            const int kw = etch_scale_exp(m);
            const float sw = ldexpf(1.0f, kw);
            if (fg == 0) wtab[(wave * SPW + s2) * 16 + fr] = ldexpf(1.0f, -kw);
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                f16x8 h, l;
                split2h_pack8(make_float4(w0[t].x * sw, w0[t].y * sw, w0[t].z * sw, w0[t].w * sw), make_float4(w1[t].x * sw, w1[t].y * sw, w1[t].z * sw, w1[t].w * sw), h, l);
                wf[t][s2][0] = __builtin_bit_cast(g_bf16x8, h); wf[t][s2][1] = __builtin_bit_cast(g_bf16x8, l);
            }
        } else {
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                const float* p = a.W + (long)(o + fr) * a.ldw + 32 * t + 8 * fg;
                g_bf16x8 q3[3];
                g_split8(*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4), q3);
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) wf[t][s2][pl] = q3[pl % 3];
            }
        }
        // epilogue constants of the lane's 4 channels o + 4 fg .. + 3
        if constexpr (EPI_REGS) epi_load(o, bs[s2], sc[s2], sh[s2]);
    }
    constexpr int C4 = K / 4;
    constexpr int XL = (FD_ROWS * C4 + 511) / 512;
    const long row_lo = (long)blockIdx.x * rows_per_block, row_hi = row_lo + rows_per_block < a.R ? row_lo + rows_per_block : a.R;
    float4 xn[XL];
    // element e = tid + 512 h of a tile: 512 is a multiple of C4 (K = 32 / 64 / 128), so h only moves the row -- one address per thread, h as an
    // immediate offset (the generic e / C4 per h cost an address register per h, one of them spilled at K = 128)
    static_assert(512 % C4 == 0 && (FD_ROWS * C4) % 512 == 0, "tile staging assumes whole 512-element passes");
    constexpr int RSTEP = 512 / C4;
    const int row0 = tid / C4, c0 = (tid % C4) * 4;
    auto fetch = [&](long r0) {
        const float* src = a.X + (r0 + row0) * a.ldx + c0;
#pragma unroll
        for (int h = 0; h < XL; ++h) {
            xn[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + row0 + h * RSTEP < row_hi) xn[h] = *reinterpret_cast<const float4*>(src + (long)(h * RSTEP) * a.ldx);
        }
    };
    auto stage = [&](unsigned short* P, float* rs) {
        unsigned short* d0 = P + row0 * SB + c0;
#pragma unroll
        for (int h = 0; h < XL; ++h) {
            if constexpr (F16) {
                const int kx = etch_scale_exp(etch_group_max<C4>(etch_max4abs(xn[h], 0.f)));      // the row's power of two (its C4 float4s: one lane group)
                const float sx = ldexpf(1.0f, kx);
                uint2 ph, pl;
                split2h_pack4(make_float4(xn[h].x * sx, xn[h].y * sx, xn[h].z * sx, xn[h].w * sx), ph, pl);
                unsigned short* d = d0 + h * RSTEP * SB;
                *reinterpret_cast<uint2*>(d) = ph;
                *reinterpret_cast<uint2*>(d + PLANE) = pl;
                if (c0 == 0) rs[row0 + h * RSTEP] = ldexpf(1.0f, -kx);
                continue;
            }
            const float v[4] = {xn[h].x, xn[h].y, xn[h].z, xn[h].w};
            unsigned hh[4], mm[4], ll[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) g_split(v[i], hh[i], mm[i], ll[i]);
            unsigned short* d = d0 + h * RSTEP * SB;
            *reinterpret_cast<uint2*>(d) = make_uint2(__builtin_amdgcn_perm(hh[1], hh[0], 0x07060302u), __builtin_amdgcn_perm(hh[3], hh[2], 0x07060302u));
            *reinterpret_cast<uint2*>(d + PLANE) = make_uint2(__builtin_amdgcn_perm(mm[1], mm[0], 0x07060302u), __builtin_amdgcn_perm(mm[3], mm[2], 0x07060302u));
            *reinterpret_cast<uint2*>(d + 2 * PLANE) = make_uint2(__builtin_amdgcn_perm(ll[1], ll[0], 0x07060302u), __builtin_amdgcn_perm(ll[3], ll[2], 0x07060302u));
        }
    };
    // row tile i of the 64-row tile at r0: D[channel 4 fg + q of strip s2][row 16 i + fr] -> epilogue -> Y
    auto row_tile = [&](const unsigned short* P, int i, long r0, const float* rs) {
        f32x4 acc[SPW];
#pragma unroll
        for (int s2 = 0; s2 < SPW; ++s2) acc[s2] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            g_bf16x8 x[NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) x[pl] = *reinterpret_cast<const g_bf16x8*>(P + pl * PLANE + (i * 16 + fr) * SB + t * 32 + fg * 8);
            if constexpr (F16) {
#define G_T(PA, PB) _Pragma("unroll") for (int s2 = 0; s2 < SPW; ++s2) \
        acc[s2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[t][s2][PA]), __builtin_bit_cast(f16x8, x[PB]), acc[s2], 0, 0, 0);
                G_T(1, 0) G_T(0, 1) G_T(0, 0)
#undef G_T
                continue;
            }
#define G_T(PA, PB) _Pragma("unroll") for (int s2 = 0; s2 < SPW; ++s2) acc[s2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][s2][PA % NPL], x[PB % NPL], acc[s2], 0, 0, 0);
            G_T(2, 0) G_T(0, 2) G_T(1, 1) G_T(1, 0) G_T(0, 1) G_T(0, 0)
#undef G_T
        }
        const long r = r0 + i * 16 + fr;
        const float rsv = F16 ? rs[i * 16 + fr] : 1.0f;      // this lane's row: the power of two of its staging
        if (r < row_hi) {
#pragma unroll
            for (int s2 = 0; s2 < SPW; ++s2) {
                const int o = 16 * (SPW * sg + s2) + 4 * fg;
                float4 b_, s_, h_;
                if constexpr (EPI_REGS) { b_ = bs[s2]; s_ = sc[s2]; h_ = sh[s2]; }
                else epi_load(o - 4 * fg, b_, s_, h_);
                float4 v;
                if constexpr (F16) {
                    // 2^-(kx + kw) of this row and these channels: exact
                    v = make_float4(fmaf(acc[s2][0], rsv * wk4[s2].x, b_.x), fmaf(acc[s2][1], rsv * wk4[s2].y, b_.y), fmaf(acc[s2][2], rsv * wk4[s2].z, b_.z),
                                    fmaf(acc[s2][3], rsv * wk4[s2].w, b_.w));
                } else {
                    v = make_float4(acc[s2][0] + b_.x, acc[s2][1] + b_.y, acc[s2][2] + b_.z, acc[s2][3] + b_.w);
                }
                if (a.scale) { v.x = v.x * s_.x + h_.x; v.y = v.y * s_.y + h_.y; v.z = v.z * s_.z + h_.z; v.w = v.w * s_.w + h_.w; }
                float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.res_mode != 0) rs = *reinterpret_cast<const float4*>(a.res + r * a.ldr + o);
                if (a.res_mode == 1) { v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w; }
                if (a.act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                else if (a.act == 2) { v.x = v.x > 0.f ? v.x : 0.01f * v.x; v.y = v.y > 0.f ? v.y : 0.01f * v.y; v.z = v.z > 0.f ? v.z : 0.01f * v.z; v.w = v.w > 0.f ? v.w : 0.01f * v.w; }
                if (a.res_mode == 2) { v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w; }
                *reinterpret_cast<float4*>(a.Y + r * a.ldy + o) = v;
            }
        }
    };
    if (row_lo >= row_hi) return;
    fetch(row_lo);
    stage(Xp, rsc);
    fetch(row_lo + FD_ROWS);
    __syncthreads();
    if constexpr (F16) {
#pragma unroll
        for (int s2 = 0; s2 < SPW; ++s2) wk4[s2] = *reinterpret_cast<const float4*>(&wtab[(wave * SPW + s2) * 16 + 4 * fg]);
    }
    int buf = 0;
    for (long r0 = row_lo; r0 < row_hi; r0 += FD_ROWS, buf ^= 1) {
        const unsigned short* P = Xp + buf * NPL * PLANE;
        row_tile(P, rp * RTW, r0, rsc + buf * FD_ROWS);
        stage(Xp + (buf ^ 1) * NPL * PLANE, rsc + (buf ^ 1) * FD_ROWS);       // the next tile (its buffer's last readers finished before the previous barrier)
        if constexpr (EPI_REGS) fetch(r0 + 2 * FD_ROWS);
#pragma unroll
        for (int i = 1; i < RTW; ++i) row_tile(P, rp * RTW + i, r0, rsc + buf * FD_ROWS);
        if constexpr (!EPI_REGS) fetch(r0 + 2 * FD_ROWS);      // register-bound shape: the prefetch registers are not live across the row tiles
        __syncthreads();
    }
}

template <int K, int SPW, int RP, bool F16 = false>
static int launch_ws_split(const GemmArgs& a, hipStream_t st) {
    constexpr int FD_ROWS = 64;
    const size_t lds = (size_t)2 * (F16 ? 2 : 3) * FD_ROWS * (K + 8) * 2 + (size_t)2 * FD_ROWS * sizeof(float) + (size_t)8 * SPW * 16 * sizeof(float);
    auto kern = gemm_ws_split_kernel<K, SPW, RP, F16>;
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
    long nb = (long)etch_cu_count() * per_cu;
    long rpb = (a.R + nb - 1) / nb;
#ifndef GEMM_WS_MIN_TILES
#define GEMM_WS_MIN_TILES 4       // a workgroup pays for splitting W once: at least this many 64-row tiles each
#endif
    if (rpb < GEMM_WS_MIN_TILES * FD_ROWS) rpb = GEMM_WS_MIN_TILES * FD_ROWS;
    rpb = (rpb + FD_ROWS - 1) / FD_ROWS * FD_ROWS;
    nb = (a.R + rpb - 1) / rpb;
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(512), lds, st, a, rpb);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

extern "C" int etch_linear(int R, int K, int O, const float* X, long ldx, const int* row_idx, int grp, int p_in, int p_out,
                           const float* W, long ldw, const float* bias, const float* scale, const float* shift, int act,
                           const float* res, long ldr, int res_mode, float* Y, long ldy, void* stream) {
    if (R <= 0 || O <= 0) return ETCH_OK;
    if (K <= 0) return ETCH_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return ETCH_EINVAL;
    if (res_mode != 0 && res == nullptr) return ETCH_EINVAL;
    if (row_idx && (grp <= 0 || p_out <= 0)) return ETCH_EINVAL;
    GemmArgs a{R, K, O, X, ldx, row_idx, grp, p_in, p_out, W, ldw, bias, scale, shift, act, res, ldr, res_mode, Y, ldy};
    const bool vx = !(K & 3) && !(ldx & 3) && !((uintptr_t)X & 15);
    const bool vw = !(K & 3) && !(ldw & 3) && !((uintptr_t)W & 15);
    hipStream_t st = (hipStream_t)stream;
    if (K == 1) {
        long blocks = ((long)R * ((O + 3) / 4) + 255) / 256;
        if (blocks > 65535L * 16) blocks = 65535L * 16;
        hipLaunchKernelGGL(linear_k1_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
        ETCH_RETURN_IF_LAUNCH_FAILED();
        return ETCH_OK;
    }
    // weight-stationary split kernel: chosen by the layer's shape only (never by R: a scan's result may not depend on its batch neighbours)
    static const bool no_split = getenv("ETCH_LINEAR_SPLIT") != nullptr && !strcmp(getenv("ETCH_LINEAR_SPLIT"), "0");
    const bool al = vx && vw && !row_idx && !(ldy & 3) && !((uintptr_t)Y & 15) && (!res || (!(ldr & 3) && !((uintptr_t)res & 15)));
    // ETCH_LINEAR_SPLIT=f16: the two-plane fp16 form (built and parity-tested in round 5; serially 5 % faster over the step's ~40 launches, but these layers
    // run on the side streams NEXT TO the VALU-bound kernels of the main stream, whose matrix pipe is two thirds idle: trading MFMAs for VALU work is the
    // wrong direction there -- the bench did not gain, the three-plane form stays the default)
    static const bool split_f16 = getenv("ETCH_LINEAR_SPLIT") != nullptr && !strcmp(getenv("ETCH_LINEAR_SPLIT"), "f16");
    if (!no_split && al) {
#define WS_CASE(KK, OO, SPW, RP) if (K == KK && O == OO) return split_f16 ? launch_ws_split<KK, SPW, RP, true>(a, st) : launch_ws_split<KK, SPW, RP, false>(a, st);
        WS_CASE(32, 32, 1, 4) WS_CASE(32, 64, 1, 2) WS_CASE(32, 128, 1, 1)
        WS_CASE(64, 64, 1, 2) WS_CASE(64, 128, 1, 1) WS_CASE(64, 192, 3, 2) WS_CASE(64, 256, 2, 1)
        WS_CASE(128, 128, 1, 1) WS_CASE(128, 256, 2, 1) WS_CASE(128, 384, 3, 1)
#undef WS_CASE
    }
    static const bool no_wres = getenv("ETCH_GEMM_NO_WRES") != nullptr;   // diagnostics: force the tiled kernel
    if (!no_wres && vx && vw && R >= 8192 && O <= 16) {   // weight-resident streaming kernel: narrow outputs only (measured)
        if (K == 64) return launch_wres<64, 1>(a, st);
        if (K == 128) return launch_wres<128, 1>(a, st);
    }
    // output-tile width 64: measured faster than 128 / 192 on every shape of the path (the X re-reads of the
    // narrower tile hit L2 / Infinity Cache; the wider tiles lose occupancy) -- see profiles/r01_gemm_shapes.txt
    static const char* tile_env = getenv("ETCH_GEMM_TILE");    // diagnostics: "128x64" / "128x128"
    if (tile_env && !strcmp(tile_env, "128x64")) launch_tiled<64, 128>(a, vx, vw, st);
    else if (tile_env && !strcmp(tile_env, "128x128")) launch_tiled<128, 128>(a, vx, vw, st);
    else if (tile_env && !strcmp(tile_env, "64x192w8") && O == 192) launch_tiled<192, 64, 8>(a, vx, vw, st);
    else if (tile_env && !strcmp(tile_env, "64x128w8") && O >= 128) launch_tiled<128, 64, 8>(a, vx, vw, st);
    else if (tile_env && !strcmp(tile_env, "128x128w8") && O >= 128) launch_tiled<128, 128, 8>(a, vx, vw, st);
    else launch_tiled<64>(a, vx, vw, st);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}
