// One whole MultiHeadAttention layer of the direction head (/root/reference/src/models/direction_backbones.py:132-194:
// key/query/value_transform -> 8-head DotProdAttention over the 60 anchor tokens of a point (:102-129) -> head_combine,
// and the residual add of StackedMHSA.forward :216-221) as ONE kernel on the gfx950 fp32 matrix cores.  The per-token
// q|k|v rows (9.6 M x 192 floats per batch) and the attention output never touch HBM: a point's 60 x 64 tokens are read
// once, everything else lives in registers / LDS.
//
// Persistent workgroup = 4 waves, one point at a time; wave w owns heads 2w and 2w+1 end to end:
//   A  Q^T, K^T (16 channels x 64 tokens) and V (64 tokens x 16 channels) tiles of its two heads from the X tile in LDS;
//      the weight fragments sit in registers for the lifetime of the workgroup.  The 16 rows of the Q/K weight tile are
//      ordered so that accumulator register r of lane group g holds dim 2g + (r&1) of head 2w + (r>>1): the accumulators
//      ARE the MFMA operands of the score product (no LDS round trip, no zero padding of the 8-wide heads).
//   B  S^T = K Q^T per (head, 16-query tile): the lane owning query i holds 16 of its 64 scores; softmax needs two
//      cross-lane-group exchanges; the normalised P accumulators ARE the A operand of P V, V's accumulators the B operand.
//      Both heads share the V tile: head 2w's result lands in output columns 0-7, head 2w+1's in columns 8-15.
//   C  head_combine (+ bias + residual) from the attention tile in LDS, float4 stores.
// Keys 60..63 (tile padding) are masked to -inf; rows 60..63 are never stored.
#include "common.h"
#include "split_bf16.h"
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ML_TOK 60
#define ML_C 64
#define ML_S 104     // LDS row stride (floats): 104/4 = 10 (mod 16) -> conflict-free ds_read_b128 fragment reads

#define ML_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// reductions over the 4 lane groups (lanes l, l^16, l^32, l^48) with the gfx950 row / half swaps: v_permlane16_swap exchanges the odd
// 16-lane rows of one register with the even rows of another, v_permlane32_swap the upper half of one with the lower half of the other;
// fed the same value twice they put x[l] and x[l^16] (resp. x[l^32]) side by side in every lane -- one VALU op instead of a
// ds_bpermute round trip through LDS (8 of them sat on the softmax's critical path per 16-query tile)
typedef unsigned ml_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float ml_xmax(float v) {
    ml_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float ml_xsum(float v) {
    ml_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// softmax over the 64 (60 valid) keys of one query, scores spread over the 4 lane groups x 16 registers; returns P / sum
__device__ __forceinline__ void ml_softmax(f32x4 (&s)[4], int fg) {
    const float c = 0.35355339059327373f * 1.4426950408889634f;     // 1/sqrt(8) * log2(e)
    // keys 60..63 do not exist: their scores arrive as -1e30 (the accumulators of the last key tile start there in lane group 3)
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) m = fmaxf(fmaxf(fmaxf(s[j][0], s[j][1]), fmaxf(s[j][2], s[j][3])), m);
    m = ml_xmax(m);
    const float mc = m * c;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(fmaf(s[j][r], c, -mc));
            s[j][r] = p;
            sum += p;
        }
    sum = ml_xsum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] *= inv;
}

// ---- the four dense products of the layer (q / k / v transforms, head_combine: 57 % of its MFMAs) on the bf16 matrix cores with exactly split
// fp32 operands: x = hi + mid + lo (8 + 8 + 8 mantissa bits, by truncation), the six largest cross products accumulated in fp32 by
// v_mfma_f32_16x16x32_bf16 -- the error against fp64 of the fp32 MFMA (profiles/r03_bf16x3_split.txt) at 2.3 x its rate, and beside the VALU
// instead of on its datapath.  Their C/D layout is that of v_mfma_f32_16x16x4_f32, so the attention phase -- whose operands ARE the
// projections' accumulators -- is untouched (it stays on the fp32 MFMA: its operands are produced per use).  The weights are split once per
// workgroup (registers), the token tile when it is staged in LDS, the attention output when it is stored for head_combine.
#define ML_PL (64 * 64)      // bf16 elements of one plane of a 64 x 64 tile: rows of 128 bytes, 16-byte units XOR-swizzled with the row (no padding:
                             // three planes of a tile take 24.6 KB against 26.6 KB of the padded fp32 tile, so three workgroups still share a CU)
__device__ __forceinline__ int ml_sw(int row, int k) { return row * 64 + ((((k) >> 3) ^ (row & 7)) << 3) + (k & 7); }
#define ML_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void ml_split(const float v, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(v);
    const float r = v - __uint_as_float(h & 0xffff0000u);
    m = __float_as_uint(r);
    l = __float_as_uint(r - __uint_as_float(m & 0xffff0000u));
}
// 8 consecutive fp32 values -> 3 planes x 8 bf16
__device__ __forceinline__ void ml_split8(const float4 v0, const float4 v1, bf16x8 (&o)[3]) {
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ml_split(v[i], h[i], m[i], l[i]);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // v_perm_b32: bytes 2, 3 of the even element below bytes 2, 3 of the odd one
#define ML_PK(a) (u32x4){__builtin_amdgcn_perm(a[1], a[0], 0x07060302u), __builtin_amdgcn_perm(a[3], a[2], 0x07060302u), \
                         __builtin_amdgcn_perm(a[5], a[4], 0x07060302u), __builtin_amdgcn_perm(a[7], a[6], 0x07060302u)}
    const u32x4 ph = ML_PK(h), pm = ML_PK(m), pl = ML_PK(l);
#undef ML_PK
    o[0] = __builtin_bit_cast(bf16x8, ph); o[1] = __builtin_bit_cast(bf16x8, pm); o[2] = __builtin_bit_cast(bf16x8, pl);
}
// float4 number e (row e >> 4, channels 4 (e & 15) ..) of a token tile -> the three planes
__device__ __forceinline__ void ml_stage4(unsigned short* P, int e, const float4 v4) {
    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ml_split(v[i], h[i], m[i], l[i]);
    unsigned short* d = P + ml_sw(e >> 4, (e & 15) * 4);
    *reinterpret_cast<uint2*>(d) = make_uint2(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u));
    *reinterpret_cast<uint2*>(d + ML_PL) = make_uint2(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u));
    *reinterpret_cast<uint2*>(d + 2 * ML_PL) = make_uint2(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u));
}
// six cross products of one K = 32 step, smallest first
#define ML_BX6(ACC, A, B)                                                                                   \
    ACC = ML_MFMA16(A[2], B[0], ACC); ACC = ML_MFMA16(A[0], B[2], ACC); ACC = ML_MFMA16(A[1], B[1], ACC);  \
    ACC = ML_MFMA16(A[1], B[0], ACC); ACC = ML_MFMA16(A[0], B[1], ACC); ACC = ML_MFMA16(A[0], B[0], ACC);

struct MlWeights {          // fragments of this wave's tiles, K step ks = channels 32 ks + 8 fg .. + 7 of row / column fr
    bf16x8 q[2][3], k[2][3], v[2][3], c[2][3];
};
template <bool COMBINE>
__device__ __forceinline__ void ml_load_weights(MlWeights& W, const float* __restrict__ Wq, const float* __restrict__ Wk, const float* __restrict__ Wv,
                                                const float* __restrict__ Wc, int w, int fr, int fg) {
    const int chq = 8 * (2 * w + ((fr & 3) >> 1)) + 2 * (fr >> 2) + (fr & 1);    // Q/K tile row fr -> original channel
    const int chv = 16 * w + fr;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int o = ks * 32 + fg * 8;
        ml_split8(*reinterpret_cast<const float4*>(Wq + chq * ML_C + o), *reinterpret_cast<const float4*>(Wq + chq * ML_C + o + 4), W.q[ks]);
        ml_split8(*reinterpret_cast<const float4*>(Wk + chq * ML_C + o), *reinterpret_cast<const float4*>(Wk + chq * ML_C + o + 4), W.k[ks]);
        ml_split8(*reinterpret_cast<const float4*>(Wv + chv * ML_C + o), *reinterpret_cast<const float4*>(Wv + chv * ML_C + o + 4), W.v[ks]);
        if (COMBINE) ml_split8(*reinterpret_cast<const float4*>(Wc + chv * ML_C + o), *reinterpret_cast<const float4*>(Wc + chv * ML_C + o + 4), W.c[ks]);
    }
}
// A: projections of this wave's two heads from the token planes
__device__ __forceinline__ void ml_project(const unsigned short* Xp, const MlWeights& W, f32x4 (&Q)[4], f32x4 (&Kt)[4], f32x4 (&V)[4], int fr, int fg) {
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        f32x4 q = {0.f, 0.f, 0.f, 0.f}, k = q, v = q;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 x[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) x[pl] = *reinterpret_cast<const bf16x8*>(Xp + pl * ML_PL + ml_sw(tt * 16 + fr, ks * 32 + fg * 8));
            // term-major over q / k / v: consecutive MFMAs are independent
#define ML_T(PA, PB) q = ML_MFMA16(W.q[ks][PA], x[PB], q); k = ML_MFMA16(W.k[ks][PA], x[PB], k); v = ML_MFMA16(x[PB], W.v[ks][PA], v);
            ML_T(2, 0) ML_T(0, 2) ML_T(1, 1) ML_T(1, 0) ML_T(0, 1) ML_T(0, 0)
#undef ML_T
        }
        Q[tt] = q; Kt[tt] = k; V[tt] = v;
    }
}
// B: attention of heads 2w (registers 0,1 / output columns 0-7) and 2w+1 (registers 2,3 / columns 8-15): scores on the fp32 MFMA (their operands are the
// projections' accumulators as they stand; 8-dim heads fill only a quarter of a bf16 K step) and so does P V (see ML_PV_BF16); the output tile goes to
// LDS as fp32 (As: the layer's output in MODE 2) or as the three planes of head_combine's operand (Ap)
template <bool PLANES>
__device__ __forceinline__ void ml_attention(const f32x4 (&Q)[4], const f32x4 (&Kt)[4], const f32x4 (&V)[4], float* As, unsigned short* Ap, int w, int fr, int fg) {
#ifdef ML_PV_BF16
    bf16x8 Vq[2][3];                                                   // the V accumulators as split operands, once per point
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        ml_split8(make_float4(V[2 * ks][0], V[2 * ks][1], V[2 * ks][2], V[2 * ks][3]),
                  make_float4(V[2 * ks + 1][0], V[2 * ks + 1][1], V[2 * ks + 1][2], V[2 * ks + 1][3]), Vq[ks]);
#endif
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        f32x4 sa[4], sb[4];
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const float pad = fg == 3 ? -1e30f : 0.f;                     // padded keys 60..63 (tile 3, lane group 3): exp2 -> 0
        const f32x4 zpad = {pad, pad, pad, pad};
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            sa[jt] = ML_MFMA(Kt[jt][0], Q[it][0], jt == 3 ? zpad : z);
            sb[jt] = ML_MFMA(Kt[jt][2], Q[it][2], jt == 3 ? zpad : z);
        }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            sa[jt] = ML_MFMA(Kt[jt][1], Q[it][1], sa[jt]);
            sb[jt] = ML_MFMA(Kt[jt][3], Q[it][3], sb[jt]);
        }
        ml_softmax(sa, fg);
        ml_softmax(sb, fg);
#ifdef ML_PV_BF16      // measured SLOWER (interp layer 4.54 -> 4.77 ms, mode 2 3.68 -> 3.76): splitting 32 probabilities per lane and query tile costs more
                       // VALU time than the 32 -> 24 cheaper MFMAs give back; kept as the record of the experiment (tests pass with it)
        // P V on the bf16 matrix cores.  A lane's 16 probabilities and its 16 V accumulators belong to the SAME keys (16 jt + 4 fg + r), so they
        // are the lane's slices of the 16x16x32 operands as they stand: K step ks = key tiles 2 ks, 2 ks + 1, element e = 4 (jt - 2 ks) + r.
        f32x4 o4[2][2] = {{z, z}, {z, z}};                             // [head][ks]: four independent chains
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 pa[3], pb[3];
            ml_split8(make_float4(sa[2 * ks][0], sa[2 * ks][1], sa[2 * ks][2], sa[2 * ks][3]),
                      make_float4(sa[2 * ks + 1][0], sa[2 * ks + 1][1], sa[2 * ks + 1][2], sa[2 * ks + 1][3]), pa);
            ml_split8(make_float4(sb[2 * ks][0], sb[2 * ks][1], sb[2 * ks][2], sb[2 * ks][3]),
                      make_float4(sb[2 * ks + 1][0], sb[2 * ks + 1][1], sb[2 * ks + 1][2], sb[2 * ks + 1][3]), pb);
#define ML_T(PA, PB) o4[0][ks] = ML_MFMA16(pa[PA], Vq[ks][PB], o4[0][ks]); o4[1][ks] = ML_MFMA16(pb[PA], Vq[ks][PB], o4[1][ks]);
            ML_T(2, 0) ML_T(0, 2) ML_T(1, 1) ML_T(1, 0) ML_T(0, 1) ML_T(0, 0)
#undef ML_T
        }
        const f32x4 oa = o4[0][0] + o4[0][1], ob = o4[1][0] + o4[1][1];
#else
        f32x4 oa = z, ob = z;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                oa = ML_MFMA(sa[jt][r], V[jt][r], oa);
                ob = ML_MFMA(sb[jt][r], V[jt][r], ob);
            }
#endif
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float o = fr < 8 ? oa[r] : ob[r];
            const int row = it * 16 + fg * 4 + r, col = 16 * w + fr;
            if (PLANES) {
                unsigned h, m, l;
                ml_split(o, h, m, l);
                unsigned short* d = Ap + ml_sw(row, col);
                d[0] = (unsigned short)(h >> 16); d[ML_PL] = (unsigned short)(m >> 16); d[2 * ML_PL] = (unsigned short)(l >> 16);
            } else {
                As[row * ML_S + col] = o;
            }
        }
    }
}
// C: head_combine, transposed (rows = this wave's 16 output channels, columns = tokens)
// wc: the wave's head_combine fragments [ks][plane] -- registers (W.c) or, where registers are short, its 6 x 1 KB slice of an LDS copy
template <class WC>
__device__ __forceinline__ void ml_combine(const unsigned short* Ap, WC wc, f32x4 (&y)[4], int fr, int fg) {
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) y[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 a[4][3], c[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) c[pl] = wc(ks, pl);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[tt][pl] = *reinterpret_cast<const bf16x8*>(Ap + pl * ML_PL + ml_sw(tt * 16 + fr, ks * 32 + fg * 8));
#define ML_T(PA, PB) _Pragma("unroll") for (int tt = 0; tt < 4; ++tt) y[tt] = ML_MFMA16(c[PA], a[tt][PB], y[tt]);
        ML_T(2, 0) ML_T(0, 2) ML_T(1, 1) ML_T(1, 0) ML_T(0, 1) ML_T(0, 0)
#undef ML_T
    }
}

// MODE 0: out = X + att Wc^T + bc (residual layer);  1: out = att Wc^T + bc;  2: out = att (head_combine folded downstream);
// MODE 3 (round 5): the direction tail folded in -- out[token] = v . relu(Wf att[token] + bf) + c (models_pointcloud.py:115-117 with the linear
//   chains folded on the host: Wf = net[0] o head_combine, v = so3_reg o net[2]): the 60 x 64 attention tile is in LDS when the layer ends, the
//   hidden layer (64 -> 128 -> 1, 1 MFLOP per point) runs on v_mfma_f32_32x32x16_f16 from two fp16 planes per operand (split_bf16.h; Wf arrives as
//   the planes of 2^6 Wf in fragment order, `Wc`; `bc` = [bf (128) | v (128) | c]); 240 bytes per point leave the kernel instead of 15 KB, and
//   linear_relu_dot_ws_kernel<64, 1> (1.74 ms, 2.46 GB read) leaves the path.
// waves per SIMD the layer is compiled for: the split weight fragments (72 - 96 registers) no longer fit three (168 registers, 116 - 268 bytes
// of scratch: mode 0 / mode 2 = 5.02 / 3.93 ms); two (no scratch): 4.13 / 3.68 ms
#ifndef ML_LAYER_WPE
#define ML_LAYER_WPE 2
#endif
template <int MODE>
__global__ void __launch_bounds__(256, ML_LAYER_WPE) mhsa_layer_kernel(long T, const float* __restrict__ X, const float* __restrict__ Wq,
                                                            const float* __restrict__ Wk, const float* __restrict__ Wv,
                                                            const float* __restrict__ Wc, const float* __restrict__ bc,
                                                            float* __restrict__ out, unsigned* __restrict__ ctr) {
    __shared__ unsigned s_grab;
    __shared__ __attribute__((aligned(16))) unsigned short Xp[3 * ML_PL];                     // token tile, three planes
    __shared__ __attribute__((aligned(16))) float Asm[MODE >= 2 ? 64 * ML_S : 3 * ML_PL / 2];   // attention tile: fp32 (MODE 2, 3) or three planes
    __shared__ __attribute__((aligned(16))) float tail_tab[MODE == 3 ? 256 : 4];              // MODE 3: bf | v
    __shared__ float tail_part[MODE == 3 ? 4 * 64 : 4];                                         // MODE 3: the four waves' shares of the 64 token sums
    float* As = Asm;
    unsigned short* Ap = reinterpret_cast<unsigned short*>(Asm);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;

    MlWeights W;
    ml_load_weights<(MODE < 2)>(W, Wq, Wk, Wv, Wc, w, fr, fg);
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE < 2 && bc) bias = *reinterpret_cast<const float4*>(bc + 16 * w + 4 * fg);
    // MODE 3: this wave's 32 hidden units of Wf as A fragments (K step ks = channels 16 ks + 8 (lane / 32) .., two planes), resident in registers
    f16x8 Wt[4][2];
    float tail_c = 0.f;
    if (MODE == 3) {
        const f16x8* wf = reinterpret_cast<const f16x8*>(Wc) + (size_t)w * 4 * 2 * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) Wt[ks][pl] = wf[(ks * 2 + pl) * 64];
        tail_tab[tid] = bc[tid];
        tail_c = bc[256];
    }
    // token rows 60..63 stay zero for the whole kernel (all planes): 3 x 4 rows x 64 bf16 = 384 dwords
    for (int e = tid; e < 384; e += 256) reinterpret_cast<unsigned*>(Xp + (e / 128) * ML_PL + ML_TOK * 64)[e % 128] = 0u;

    // next point's tokens: 960 float4 over 256 threads, four named registers (an indexed array captured by a lambda ended up
    // in scratch memory: +5 GB of HBM traffic per launch)
    float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0, x2 = x0, x3 = x0;
#define ML_GLOAD(P)                                                                          \
    {                                                                                        \
        const float4* src_ = reinterpret_cast<const float4*>(X + (P) * (ML_TOK * ML_C));     \
        x0 = src_[tid]; x1 = src_[tid + 256]; x2 = src_[tid + 512];                          \
        if (tid < ML_TOK * ML_C / 4 - 768) x3 = src_[tid + 768];                             \
    }
    // Work distribution.  The workgroups are persistent; under the 2-deep pipeline other streams' kernels share the compute units unevenly, and a
    // static round-robin makes the whole launch wait for the workgroups that were slowed down.  With `ctr` (zeroed by the launcher) the first two
    // points of a workgroup are static and every further one is taken from a device-wide counter: thread 0 asks for the point after next while the
    // current one computes (the atomic's latency disappears behind the layer), the answer crosses LDS behind the loop's last barrier.
    long pt = blockIdx.x, nxt = (long)blockIdx.x + gridDim.x;
    if (pt < T) ML_GLOAD(pt)
    for (; pt < T;) {
        ml_stage4(Xp, tid, x0); ml_stage4(Xp, tid + 256, x1); ml_stage4(Xp, tid + 512, x2);
        if (tid < ML_TOK * ML_C / 4 - 768) ml_stage4(Xp, tid + 768, x3);
        __syncthreads();
        if (nxt < T) ML_GLOAD(nxt)                            // next point's tokens: in flight during the whole layer
        unsigned grabbed = 0u;
        if (ctr && tid == 0) grabbed = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        f32x4 Q[4], Kt[4], V[4];
        ml_project(Xp, W, Q, Kt, V, fr, fg);
        ml_attention<(MODE < 2)>(Q, Kt, V, As, Ap, w, fr, fg);
        __syncthreads();

        float* dst = out + pt * (ML_TOK * ML_C);
        if (MODE == 3) {
            // hidden[h][token] = sum_c Wf[h][c] att[token][c]: A = Wf (rows = this wave's hidden units), B = the attention tile (column = token, 8 channels
            // per lane and K step, split into two fp16 planes by the wave that reads them); three cross terms, smallest first
            const int tl = lane & 31, kg = lane >> 5;
            typedef float f32x16_ __attribute__((ext_vector_type(16)));
            f32x16_ d[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int v = 0; v < 16; ++v) d[tt][v] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const float* xr = &As[(32 * tt + tl) * ML_S + 16 * ks + 8 * kg];
                    f16x8 bh, bl;
                    split2h_pack8(*reinterpret_cast<const float4*>(xr), *reinterpret_cast<const float4*>(xr + 4), bh, bl);
                    d[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wt[ks][1], bh, d[tt], 0, 0, 0);
                    d[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wt[ks][0], bl, d[tt], 0, 0, 0);
                    d[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wt[ks][0], bh, d[tt], 0, 0, 0);
                }
            // d[tt][v] = 2^6 hidden[32 w + 8 (v / 4) + 4 kg + v % 4][token 32 tt + tl]: bias, ReLU, . v, summed over this lane's 16 hidden units,
            // then over the two lane halves and (through LDS, fixed order) over the four waves
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                float sacc = 0.f;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 b4 = *reinterpret_cast<const float4*>(&tail_tab[32 * w + 8 * q4 + 4 * kg]);
                    const float4 v4 = *reinterpret_cast<const float4*>(&tail_tab[128 + 32 * w + 8 * q4 + 4 * kg]);
                    sacc = fmaf(fmaxf(fmaf(d[tt][4 * q4 + 0], 0.015625f, b4.x), 0.f), v4.x, sacc);
                    sacc = fmaf(fmaxf(fmaf(d[tt][4 * q4 + 1], 0.015625f, b4.y), 0.f), v4.y, sacc);
                    sacc = fmaf(fmaxf(fmaf(d[tt][4 * q4 + 2], 0.015625f, b4.z), 0.f), v4.z, sacc);
                    sacc = fmaf(fmaxf(fmaf(d[tt][4 * q4 + 3], 0.015625f, b4.w), 0.f), v4.w, sacc);
                }
                const ml_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(sacc), __float_as_uint(sacc), false, false);
                if (kg == 0) tail_part[w * 64 + 32 * tt + tl] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
            __syncthreads();
            if (tid < ML_TOK) out[pt * ML_TOK + tid] = ((tail_part[tid] + tail_part[64 + tid]) + (tail_part[128 + tid] + tail_part[192 + tid])) + tail_c;
        } else if (MODE == 2) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int e = tid + 256 * h;
                if (e < ML_TOK * ML_C / 4)
                    reinterpret_cast<float4*>(dst)[e] = *reinterpret_cast<const float4*>(&As[(e >> 4) * ML_S + (e & 15) * 4]);
            }
        } else {
            f32x4 y[4];
            ml_combine(Ap, [&](int ks, int pl) { return W.c[ks][pl]; }, y, fr, fg);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int tok = tt * 16 + fr;
                if (tok < ML_TOK) {
                    float4 o = make_float4(y[tt][0] + bias.x, y[tt][1] + bias.y, y[tt][2] + bias.z, y[tt][3] + bias.w);
                    if (MODE == 0) {                       // the residual: the point's own rows again (L2: they were read a layer ago)
                        const float4 rx = *reinterpret_cast<const float4*>(X + pt * (ML_TOK * ML_C) + tok * ML_C + 16 * w + 4 * fg);
                        o.x += rx.x; o.y += rx.y; o.z += rx.z; o.w += rx.w;
                    }
                    *reinterpret_cast<float4*>(dst + tok * ML_C + 16 * w + 4 * fg) = o;
                }
            }
        }
        if (ctr && tid == 0) s_grab = grabbed;
        __syncthreads();      // Xp / As are rewritten by the next point
        pt = nxt;
        nxt = ctr ? 2L * gridDim.x + s_grab : nxt + gridDim.x;
    }
}

// ------------------------------------------------------------------------------------------------
// First layer of the direction head on INTERPOLATED tokens without a separate interpolation pass: the tokens of a scan point are the
// 3-NN blend X = w0 F[i0] + w1 F[i1] + w2 F[i2] of coarse token tiles (pointnet2_utils.py:45-74); mhsa_interp_layer_kernel forms the
// tile on its way into LDS (same arithmetic as prop_interp_kernel) and runs the layer on it.  The interpolated tokens (2.46 GB
// written and read back per batch) and the interpolation kernel disappear; the workgroups of an XCD walk one contiguous eighth of
// the scans' spatial order so that the three coarse rows of a point (shared with its neighbours) hit in that XCD's L2.
// Also measured (profiles/scripts/dirhead_time.py, 32 x 5000 points): the q / k / v transforms are linear and bias-free, so they can be
// evaluated once per COARSE point (4x fewer) and blended per scan point in the attention phase's register layout.  Blending all three
// removes 768 of 1 760 MFMAs per point but reads 192 KB per point from L2 (68 % hit rate): 4.93 ms + 0.71 ms projection; q and k
// only 4.96 + 0.53; q only 5.17 + 0.29 -- against 5.56 ms for this form, i.e. the same within noise: the layer is bound by
// its dependent phases at 2 - 3 waves per SIMD, not by the matrix-core count, so the simple form is kept.
#ifndef MHSA_INTERP_WGS
#define MHSA_INTERP_WGS 2
#endif
__device__ __forceinline__ f32x4 ml_blend(const float4 a, const float4 b, const float4 c, float w0, float w1, float w2) {
#pragma clang fp contract(off)      // the arithmetic of prop_interp_kernel, bit for bit
    f32x4 r;
    r[0] = (a.x * w0 + b.x * w1) + c.x * w2;
    r[1] = (a.y * w0 + b.y * w1) + c.y * w2;
    r[2] = (a.z * w0 + b.z * w1) + c.z * w2;
    r[3] = (a.w * w0 + b.w * w1) + c.w * w2;
    return r;
}

// sched[slot] = {output row b N + n, coarse rows b S + idx[0..2], weights[0..2], 0} for slot = position of (b, n) in the processing order
__global__ void __launch_bounds__(256) interp_schedule_kernel(int B, int N, int S, const int* __restrict__ idx, const float* __restrict__ wgt,
                                                              const int* __restrict__ order, int4* __restrict__ sched) {
    const long slot = (long)blockIdx.x * 256 + threadIdx.x;
    if (slot >= (long)B * N) return;
    const long b = slot / N, s = slot - b * N;
    const long pt = b * N + (order ? order[slot] : s);
    const int cb = (int)(b * S);
    sched[2 * slot] = make_int4((int)pt, cb + idx[pt * 3], cb + idx[pt * 3 + 1], cb + idx[pt * 3 + 2]);
    sched[2 * slot + 1] = make_int4(__float_as_int(wgt[pt * 3]), __float_as_int(wgt[pt * 3 + 1]), __float_as_int(wgt[pt * 3 + 2]), 0);
}

// out[b,n] = X + att Wc^T + bc with X = blend of three rows of F (B,S,60,64), sched from interp_schedule_kernel.  Grid = multiple of 8
// workgroups.  Per scan point: its three coarse token rows were requested during the previous point (12 float4 in registers) and are
// blended into LDS (fp32 for the residual, three bf16 planes for the projections); the rows of the NEXT point are requested before the
// layer's three phases start.  LDS: 24.6 + 24.6 + 24.6 KB (dynamic).
#define ML_INTERP_LDS (2 * 3 * ML_PL * 2 + 4 * 2 * 3 * 64 * 16)
__global__ void __launch_bounds__(256, MHSA_INTERP_WGS) mhsa_interp_layer_kernel(int B, int N, int S, const float* __restrict__ F,
                                                                   const int* __restrict__ sched, const float* __restrict__ Wq,
                                                                   const float* __restrict__ Wk, const float* __restrict__ Wv,
                                                                   const float* __restrict__ Wc, const float* __restrict__ bc,
                                                                   float* __restrict__ out, unsigned* __restrict__ ctr) {
    __shared__ unsigned s_grab;
    extern __shared__ __attribute__((aligned(16))) float ml_dyn[];
    unsigned short* Xp = reinterpret_cast<unsigned short*>(ml_dyn);                // token tile, three planes (the residual is rebuilt from them: exact)
    unsigned short* Ap = Xp + 3 * ML_PL;                                           // attention tile, three planes
    bf16x8* Wcl = reinterpret_cast<bf16x8*>(Ap + 3 * ML_PL);                       // head_combine fragments [wave][ks][plane][lane]: 24 registers the
                                                                                   // prefetched coarse rows (48) leave no room for
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    MlWeights W;
    ml_load_weights<false>(W, Wq, Wk, Wv, Wc, w, fr, fg);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 c[3];
        const float* p = Wc + (16 * w + fr) * ML_C + ks * 32 + fg * 8;
        ml_split8(*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4), c);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) Wcl[((w * 2 + ks) * 3 + pl) * 64 + lane] = c[pl];
    }
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bc) bias = *reinterpret_cast<const float4*>(bc + 16 * w + 4 * fg);
    for (int e = tid; e < 384; e += 256) reinterpret_cast<unsigned*>(Xp + (e / 128) * ML_PL + ML_TOK * 64)[e % 128] = 0u;      // token rows 60..63: zero

    const long T = (long)B * N;
    const long share = (T + 7) >> 3;                    // contiguous slots of the spatial order per XCD
    const int per = gridDim.x >> 3, xcd = blockIdx.x & 7;
    const long lim = share < T - xcd * share ? share : T - xcd * share;
    // positions in the XCD's share of the spatial order: the first three of a workgroup are static, every further one comes from the XCD's counter
    // (see mhsa_layer_kernel; per-XCD counters keep the coarse rows of neighbouring points in that XCD's L2)
    long q = blockIdx.x >> 3, q1 = q + (gridDim.x >> 3), q2 = q1 + (gridDim.x >> 3);
    // Scalar state of a point = one 32-byte record of `sched` (output row, three coarse rows, three weights), read two points ahead:
    // the record of point k+2 is requested while point k computes, so that the token rows of k+1 can be requested at the top of k
    // without waiting for a scalar load (order -> idx -> rows would be three dependent latencies per point).
    const int4* rec = reinterpret_cast<const int4*>(sched) + 2 * (xcd * share);
    int4 c_i = make_int4(0, 0, 0, 0), c_w = c_i, n_i = c_i, n_w = c_i;
    if (q < lim) { c_i = rec[2 * q]; c_w = rec[2 * q + 1]; }
    if (q1 < lim) { n_i = rec[2 * q1]; n_w = rec[2 * q1 + 1]; }
    // raw token rows of a point: element e = tid + 256 h of each of its three coarse rows
    float4 xa[4], xb[4], xc[4];
#define ML_XLOAD(R)                                                                        \
    {                                                                                      \
        const float4* r0_ = reinterpret_cast<const float4*>(F + (size_t)(R).y * (ML_TOK * ML_C));  \
        const float4* r1_ = reinterpret_cast<const float4*>(F + (size_t)(R).z * (ML_TOK * ML_C));  \
        const float4* r2_ = reinterpret_cast<const float4*>(F + (size_t)(R).w * (ML_TOK * ML_C));  \
        _Pragma("unroll") for (int h = 0; h < 4; ++h) {                                    \
            const int e = tid + 256 * h;                                                   \
            if (h < 3 || e < ML_TOK * ML_C / 4) { xa[h] = r0_[e]; xb[h] = r1_[e]; xc[h] = r2_[e]; }  \
        }                                                                                  \
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) xa[h] = xb[h] = xc[h] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < lim) ML_XLOAD(c_i)
    for (; q < lim;) {
        const long cpt = c_i.x;
        const float a0 = __int_as_float(c_w.x), a1 = __int_as_float(c_w.y), a2 = __int_as_float(c_w.z);
        // the token tile itself (projections, residual)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int e = tid + 256 * h;
            if (h < 3 || e < ML_TOK * ML_C / 4) {
                const f32x4 v = ml_blend(xa[h], xb[h], xc[h], a0, a1, a2);
                ml_stage4(Xp, e, make_float4(v[0], v[1], v[2], v[3]));
            }
        }
        // the next point's token rows: in flight during the whole layer
        if (q1 < lim) ML_XLOAD(n_i)
        c_i = n_i; c_w = n_w;
        if (q2 < lim) { n_i = rec[2 * q2]; n_w = rec[2 * q2 + 1]; }
        unsigned grabbed = 0u;
        if (ctr && tid == 0) grabbed = __hip_atomic_fetch_add(ctr + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();

        f32x4 Q[4], Kt[4], V[4];
        ml_project(Xp, W, Q, Kt, V, fr, fg);
        ml_attention<true>(Q, Kt, V, nullptr, Ap, w, fr, fg);
        __syncthreads();

        // ---- C: head_combine + bias + residual
        f32x4 y[4];
        ml_combine(Ap, [&](int ks, int pl) { return Wcl[((w * 2 + ks) * 3 + pl) * 64 + lane]; }, y, fr, fg);
        float* dst = out + cpt * (ML_TOK * ML_C);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int tok = tt * 16 + fr;
            if (tok < ML_TOK) {
                // the residual X[tok][16 w + 4 fg ..] = hi + mid + lo of the planes: (hi + mid) + lo is exact in fp32
                const unsigned short* xp = Xp + ml_sw(tok, 16 * w + 4 * fg);
                const uint2 ph = *reinterpret_cast<const uint2*>(xp), pm = *reinterpret_cast<const uint2*>(xp + ML_PL), pl = *reinterpret_cast<const uint2*>(xp + 2 * ML_PL);
#define ML_LO(u) __uint_as_float((u) << 16)
#define ML_HI(u) __uint_as_float((u) & 0xffff0000u)
                const float4 rx = make_float4((ML_LO(ph.x) + ML_LO(pm.x)) + ML_LO(pl.x), (ML_HI(ph.x) + ML_HI(pm.x)) + ML_HI(pl.x),
                                              (ML_LO(ph.y) + ML_LO(pm.y)) + ML_LO(pl.y), (ML_HI(ph.y) + ML_HI(pm.y)) + ML_HI(pl.y));
#undef ML_LO
#undef ML_HI
                *reinterpret_cast<float4*>(dst + tok * ML_C + 16 * w + 4 * fg) =
                    make_float4(y[tt][0] + bias.x + rx.x, y[tt][1] + bias.y + rx.y, y[tt][2] + bias.z + rx.z, y[tt][3] + bias.w + rx.w);
            }
        }
        if (ctr && tid == 0) s_grab = grabbed;
        __syncthreads();      // Xp / Ap are rewritten by the next point
        q = q1; q1 = q2;
        q2 = ctr ? 3L * per + s_grab : q2 + per;
    }
#undef ML_XLOAD
}

// mean over the A tokens of each point: X (T, A, C) -> mean (T, C); C <= 256, C % 4 == 0
__global__ void __launch_bounds__(256) token_mean_kernel(long T, int A, int C, const float* __restrict__ X, float* __restrict__ mean) {
    __shared__ float4 part[256];
    const int C4 = C >> 2, c4 = threadIdx.x % C4, a0 = threadIdx.x / C4, AP = 256 / C4;
    for (long pt = blockIdx.x; pt < T; pt += gridDim.x) {
        const float4* src = reinterpret_cast<const float4*>(X + (size_t)pt * A * C);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a0 < AP)
            for (int a = a0; a < A; a += AP) {
                const float4 v = src[a * C4 + c4];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        part[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x < C4) {
            float4 t = part[threadIdx.x];
            for (int k = 1; k < AP; ++k) {
                const float4 u = part[k * C4 + threadIdx.x];
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            const float ia = 1.0f / (float)A;
            reinterpret_cast<float4*>(mean + (size_t)pt * C)[threadIdx.x] = make_float4(t.x * ia, t.y * ia, t.z * ia, t.w * ia);
        }
        __syncthreads();
    }
}

template <int MODE>
static int launch_layer(long T, const float* X, const float* Wq, const float* Wk, const float* Wv, const float* Wc, const float* bc,
                        float* out, hipStream_t st) {
    auto kern = mhsa_layer_kernel<MODE>;
    static int per_cu = 0;
    if (per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)kern, 256, 0) != hipSuccess || n < 1) n = 2;
        per_cu = n;
    }
    long blocks = (long)etch_cu_count() * per_cu;
    if (blocks > T) blocks = T;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, st, T, X, Wq, Wk, Wv, Wc, bc, out, etch_work_counter_slot(st));
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

// MODE 3: attention heads of the last layer + the folded direction tail.  Wfq = ops.dirtail_weight_split(Wf): [4 waves][4 K steps][2 planes][64 lanes][8] fp16 of
// 2^6 Wf (Wf 128 x 64 = direction_predictor.net[0] o head_combine); tab = [bf (128) | v (128) | c] fp32; out (T, 60) = the anchor weights of so3_mean
extern "C" int etch_mhsa_layer_dirtail(long T, const float* X, const float* Wq, const float* Wk, const float* Wv, const void* Wfq, const float* tab,
                                       float* out, void* stream) {
    if (T <= 0) return ETCH_OK;
    if (!X || !Wq || !Wk || !Wv || !Wfq || !tab || !out) return ETCH_EINVAL;
    if (((uintptr_t)X | (uintptr_t)Wq | (uintptr_t)Wk | (uintptr_t)Wv | (uintptr_t)Wfq | (uintptr_t)tab) & 15) return ETCH_EINVAL;
    return launch_layer<3>(T, X, Wq, Wk, Wv, reinterpret_cast<const float*>(Wfq), tab, out, (hipStream_t)stream);
}

extern "C" int etch_mhsa_layer(long T, const float* X, const float* Wq, const float* Wk, const float* Wv, const float* Wc,
                               const float* bc, int mode, float* out, void* stream) {
    if (T <= 0) return ETCH_OK;
    if (!X || !Wq || !Wk || !Wv || !out || (mode != 2 && !Wc)) return ETCH_EINVAL;
    if (((uintptr_t)X | (uintptr_t)Wq | (uintptr_t)Wk | (uintptr_t)Wv | (uintptr_t)Wc | (uintptr_t)bc | (uintptr_t)out) & 15) return ETCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) return launch_layer<0>(T, X, Wq, Wk, Wv, Wc, bc, out, st);
    if (mode == 1) return launch_layer<1>(T, X, Wq, Wk, Wv, Wc, bc, out, st);
    if (mode == 2) return launch_layer<2>(T, X, Wq, Wk, Wv, Wc, bc, out, st);
    return ETCH_EINVAL;
}

extern "C" int etch_mhsa_interp_layer(int B, int N, int S, const float* F, const int* idx, const float* weight,
                                      const int* order, const float* Wq, const float* Wk, const float* Wv, const float* Wc, const float* bc,
                                      float* out, int* sched, void* stream) {
    if (B <= 0 || N <= 0) return ETCH_OK;
    if (S <= 0 || !F || !Wq || !idx || !weight || !Wk || !Wv || !Wc || !out || !sched) return ETCH_EINVAL;
    if ((long)B * N > 0x7fffffffL || (long)B * S > 0x7fffffffL) return ETCH_EUNSUPPORTED;
    if (((uintptr_t)F | (uintptr_t)Wq | (uintptr_t)Wk | (uintptr_t)Wv | (uintptr_t)Wc | (uintptr_t)bc | (uintptr_t)out | (uintptr_t)sched) & 15)
        return ETCH_EINVAL;
    static int per_cu = 0;
    if (per_cu == 0) {
        hipError_t e = hipFuncSetAttribute((const void*)mhsa_interp_layer_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ML_INTERP_LDS);
        if (e != hipSuccess) return (int)e;
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)mhsa_interp_layer_kernel, 256, ML_INTERP_LDS) != hipSuccess || n < 1) n = 2;
        per_cu = n;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(interp_schedule_kernel, dim3((unsigned)(((long)B * N + 255) / 256)), dim3(256), 0, st, B, N, S, idx, weight, order,
                       reinterpret_cast<int4*>(sched));
    hipLaunchKernelGGL(mhsa_interp_layer_kernel, dim3((unsigned)(((etch_cu_count() + 7) / 8) * 8 * per_cu)), dim3(256), ML_INTERP_LDS, st, B, N, S, F, sched, Wq, Wk, Wv, Wc, bc, out, etch_work_counter_slot(st));
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

extern "C" int etch_token_mean(long T, int A, int C, const float* X, float* mean, void* stream) {
    if (T <= 0) return ETCH_OK;
    if (A <= 0 || C <= 0 || C > 256 || (C & 3) || 256 % (C >> 2) || !X || !mean) return ETCH_EINVAL;
    long blocks = T < 256L * 16 ? T : 256L * 16;
    hipLaunchKernelGGL(token_mean_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, T, A, C, X, mean);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}
