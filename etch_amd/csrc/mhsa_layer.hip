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

// ---- Round 5: every product of the layer on the fp16 matrix cores (v_mfma_f32_16x16x32_f16) from TWO fp16 planes per operand (split_bf16.h:
// h = fp16(x), l = fp16(x - h): 22 mantissa bits; cross terms l h + h l + h h, the score product also l l), fp32 accumulation.  Per wave and point:
// 72 + 32 + 48 MFMAs instead of 144 bf16 (six cross terms of the three-plane split) + 192 fp32 16x16x4 ones -- the attention phase's fp32 MFMAs were
// 73 % of the layer's matrix time (profiles/r05_mhsa_f16.txt).
//   Range: fp16 planes carry 22 bits only for |h| >= 2^-3, so nothing here relies on the operands' natural scale -- every operand is brought to a
//   known power-of-two range first and the powers are taken out again where a scalar is applied anyway:
//     token tile   x 2^kx, max |x| 2^kx in [8, 16) (the tile's maximum is found while the previous point computes);
//     weights      x 2^kw likewise (once per workgroup, in the kernel);  |q|, |k|, |v| accumulators <= 64 * 16 * 16 = 2^14 -- inside fp16's range, so
//                  they are split as they stand; the scores' power 2^-(2 kx + kq + kk) rides on the softmax's 1/sqrt(8) log2(e) factor;
//     softmax      P x 2^13 (the factor of the normalisation), attention output x 2^-13 x 2^-(kx + kv) on its way out (or later: head_combine and the
//                  fused tail take it at <= 2^14 and apply the power with their own epilogue factor).
//   All powers of two: exact.  Elements below 2^-3 after scaling (2^-7 of the tile's maximum) keep an absolute error of 2^-25 -- 2^-29 of the maximum,
//   under the fp32 MFMA's own rounding of the sums they enter.
// The C/D layout of v_mfma_f32_16x16x32_f16 is that of v_mfma_f32_16x16x4_f32: the projections' accumulators are still the attention phase's operands
// (lane (fr, fg) of the Q^T / K^T tiles holds dims 2 fg, 2 fg + 1 of both heads for token fr = its own slots 8 fg .. 8 fg + 7 of a K = 32 step).
#define ML_PL (64 * 64)      // fp16 elements of one plane of a 64 x 64 tile: rows of 128 bytes, 16-byte units XOR-swizzled with the row
__device__ __forceinline__ int ml_sw(int row, int k) { return row * 64 + ((((k) >> 3) ^ (row & 7)) << 3) + (k & 7); }
#define ML_MFMAH(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
typedef unsigned ml_u32x4 __attribute__((ext_vector_type(4)));

// maximum over the wave of a non-negative value: quad / half-row / row mirrors inside the rows of 16 (DPP), swaps across the lane groups
__device__ __forceinline__ float ml_wave_max(float v) {
#define ML_DPP(C) v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), (C), 0xf, 0xf, true)));
    ML_DPP(0xB1) ML_DPP(0x4E) ML_DPP(0x141) ML_DPP(0x140)
#undef ML_DPP
    return ml_xmax(v);
}
// v_max3_f32 as written (fmaxf chains carry a canonicalising v_max_f32 x, x per operand that comes from memory or an MFMA: 240 instead of 64 operations per point)
// (in place -- the result overwrites the first operand's register: see split_bf16.h on inline-asm results and in-flight MFMAs)
__device__ __forceinline__ float ml_max3(float a, float b, float c) { asm("v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); return a; }
__device__ __forceinline__ float ml_max4(const float4 v, float m) { return etch_max4abs(v, m); }
// k with m 2^k in [8, 16) (m = 0 or subnormal: 0; capped so that 2^k is a float)
__device__ __forceinline__ int ml_scale_exp(float m) {
    const int e = (int)((__float_as_uint(m) >> 23) & 0xffu);
    return e == 0 ? 0 : (130 - e < 120 ? 130 - e : 120);
}

// float4 number e (row e >> 4, channels 4 (e & 15) ..) of a token tile, times the tile's power of two -> the two planes
__device__ __forceinline__ void ml_stage4(unsigned short* P, int e, const float4 v, float s) {
    uint2 ph, pl;
    split2h_pack4(make_float4(v.x * s, v.y * s, v.z * s, v.w * s), ph, pl);
    unsigned short* d = P + ml_sw(e >> 4, (e & 15) * 4);
    *reinterpret_cast<uint2*>(d) = ph;
    *reinterpret_cast<uint2*>(d + ML_PL) = pl;
}
// three cross products of one K = 32 step, smallest first
#define ML_HX3(ACC, A, B) ACC = ML_MFMAH(A[1], B[0], ACC); ACC = ML_MFMAH(A[0], B[1], ACC); ACC = ML_MFMAH(A[0], B[0], ACC);

struct MlWeights {          // fragments of this wave's tiles (x 2^kw), K step ks = channels 32 ks + 8 fg .. + 7 of row / column fr; planes h, l
    f16x8 q[2][2], k[2][2], v[2][2], c[2][2];
    int kq, kk, kv, kc;     // the powers
};
// red: 16 floats of LDS; ends with a barrier
template <bool COMBINE>
__device__ __forceinline__ void ml_load_weights(MlWeights& W, const float* __restrict__ Wq, const float* __restrict__ Wk, const float* __restrict__ Wv,
                                                const float* __restrict__ Wc, float* red, int tid, int w, int fr, int fg) {
    const float* Ws[4] = {Wq, Wk, Wv, Wc};
#pragma unroll
    for (int j = 0; j < (COMBINE ? 4 : 3); ++j) {
        float m = 0.f;
#pragma unroll
        for (int h = 0; h < 4; ++h) m = ml_max4(reinterpret_cast<const float4*>(Ws[j])[tid + 256 * h], m);
        m = ml_wave_max(m);
        if ((tid & 63) == 0) red[4 * j + w] = m;
    }
    __syncthreads();
    int kw[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < (COMBINE ? 4 : 3); ++j) kw[j] = ml_scale_exp(fmaxf(fmaxf(red[4 * j], red[4 * j + 1]), fmaxf(red[4 * j + 2], red[4 * j + 3])));
    W.kq = kw[0]; W.kk = kw[1]; W.kv = kw[2]; W.kc = kw[3];
    const int chq = 8 * (2 * w + ((fr & 3) >> 1)) + 2 * (fr >> 2) + (fr & 1);    // Q/K tile row fr -> original channel
    const int chv = 16 * w + fr;
#define ML_W8(P, K, OUT)                                                                                           \
    {                                                                                                              \
        const float s_ = ldexpf(1.0f, (K));                                                                        \
        const float4 a_ = *reinterpret_cast<const float4*>(P), b_ = *reinterpret_cast<const float4*>((P) + 4);      \
        split2h_pack8(make_float4(a_.x * s_, a_.y * s_, a_.z * s_, a_.w * s_), make_float4(b_.x * s_, b_.y * s_, b_.z * s_, b_.w * s_), OUT[0], OUT[1]); \
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int o = ks * 32 + fg * 8;
        ML_W8(Wq + chq * ML_C + o, W.kq, W.q[ks])
        ML_W8(Wk + chq * ML_C + o, W.kk, W.k[ks])
        ML_W8(Wv + chv * ML_C + o, W.kv, W.v[ks])
        if (COMBINE) ML_W8(Wc + chv * ML_C + o, W.kc, W.c[ks])
    }
}
// A: projections of this wave's two heads from the token planes
__device__ __forceinline__ void ml_project(const unsigned short* Xp, const MlWeights& W, f32x4 (&Q)[4], f32x4 (&Kt)[4], f32x4 (&V)[4], int fr, int fg) {
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        f32x4 q = {0.f, 0.f, 0.f, 0.f}, k = q, v = q;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 x[2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) x[pl] = *reinterpret_cast<const f16x8*>(Xp + pl * ML_PL + ml_sw(tt * 16 + fr, ks * 32 + fg * 8));
            // term-major over q / k / v: consecutive MFMAs are independent
#define ML_T(PA, PB) q = ML_MFMAH(W.q[ks][PA], x[PB], q); k = ML_MFMAH(W.k[ks][PA], x[PB], k); v = ML_MFMAH(x[PB], W.v[ks][PA], v);
            ML_T(1, 0) ML_T(0, 1) ML_T(0, 0)
#undef ML_T
        }
        Q[tt] = q; Kt[tt] = k; V[tt] = v;
    }
}
// softmax over the 64 (60 valid) keys of one query, scores spread over the 4 lane groups x 16 registers, c = the scores' factor (1/sqrt(8) log2(e) and
// their power of two); returns 2^13 P
__device__ __forceinline__ void ml_softmax(f32x4 (&s)[4], float c) {
    // keys 60..63 do not exist: their scores arrive as -inf (the accumulators of the last key tile start there in lane group 3)
    float m = ml_max3(s[0][0], s[0][1], s[0][2]);
    m = ml_max3(m, s[0][3], s[1][0]);
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        m = ml_max3(m, s[j][1], s[j][2]);
        m = ml_max3(m, s[j][3], s[j < 3 ? j + 1 : 0][0]);
    }
    m = ml_xmax(m);
    const float mc = m * c;
    // the layer is VALU-bound since round 5: the exponent's fma and the running sum as PACKED fp32 instructions (v_pk_fma_f32 / v_pk_add_f32: two values
    // per issue slot; beside an idle matrix pipe that is half the instructions) -- two partial sums, added at the end
    typedef float ml_f32x2 __attribute__((ext_vector_type(2)));
    const ml_f32x2 c2 = {c, c}, nmc2 = {-mc, -mc};
    ml_f32x2 sum2 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            const ml_f32x2 t = __builtin_elementwise_fma((ml_f32x2){s[j][r], s[j][r + 1]}, c2, nmc2);
            const ml_f32x2 p = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
            s[j][r] = p[0]; s[j][r + 1] = p[1];
            sum2 += p;
        }
    float sum = ml_xsum(sum2[0] + sum2[1]);
    float inv = __builtin_amdgcn_rcpf(sum);          // + one Newton step: the quotient to an ulp in 4 operations instead of the division's 10
    inv = fmaf(fmaf(-sum, inv, 1.0f), inv, inv) * 8192.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] *= inv;
}
// B: attention of heads 2w (registers 0,1 / output columns 0-7) and 2w+1 (registers 2,3 / columns 8-15).  Score product: ONE MFMA per (key tile, query
// tile, head) -- the lane's two dims of a head, as planes, fill its eight slots of the K = 32 step with all four cross terms:
// K side [h h' | l l' | h h' | l l'], Q side [l l' | h h' | h h' | l l'].  P V: a lane's 16 probabilities and its 16 V accumulators belong to the SAME
// keys (16 jt + 4 fg + r), so they are the lane's slices of the operands as they stand: K step ks = key tiles 2 ks, 2 ks + 1, element 4 (jt - 2 ks) + r.
// cs: the scores' factor; the output tile, times os, goes to LDS as fp32 (As) or as the two planes of head_combine's operand (Ap)
// the scores' factor 1/sqrt(8) log2(e) 2^-k (k = the powers of two the projections' operands carry).  The exponent is held at >= -125 so that the
// factor stays a NORMAL float: a factor that underflowed to 0 met the padded keys' -inf start value as (-inf) * 0 = NaN (k > 149 needs a token
// tile below ~1e-20).  With the clamp the padded keys stay at -inf (probability 0) and the real scores, all ~0, give the uniform softmax over the 60
// keys that the reference returns for such a tile.
__device__ __forceinline__ float ml_score_factor(int k) { return ldexpf(0.35355339059327373f * 1.4426950408889634f, k < 125 ? -k : -125); }
template <bool PLANES>
__device__ __forceinline__ void ml_attention(const f32x4 (&Q)[4], const f32x4 (&Kt)[4], const f32x4 (&V)[4], float* As, unsigned short* Ap, float cs, float os,
                                             int w, int fr, int fg) {
    f16x8 KB[4][2], Vq[2][2];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int hd = 0; hd < 2; ++hd) {
            unsigned h, l;
            split2h_pair(Kt[tt][2 * hd], Kt[tt][2 * hd + 1], h, l);
            KB[tt][hd] = __builtin_bit_cast(f16x8, (ml_u32x4){h, l, h, l});
        }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        split2h_pack8(make_float4(V[2 * ks][0], V[2 * ks][1], V[2 * ks][2], V[2 * ks][3]),
                      make_float4(V[2 * ks + 1][0], V[2 * ks + 1][1], V[2 * ks + 1][2], V[2 * ks + 1][3]), Vq[ks][0], Vq[ks][1]);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        f32x4 sa[4], sb[4];
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const float pad = fg == 3 ? -INFINITY : 0.f;                     // padded keys 60..63 (tile 3, lane group 3): exp2 -> 0
        const f32x4 zpad = {pad, pad, pad, pad};
        f16x8 QA[2];                                                   // this query tile's operands (the fp32 accumulators stay: 16 instead of 32 registers)
#pragma unroll
        for (int hd = 0; hd < 2; ++hd) {
            unsigned h, l;
            split2h_pair(Q[it][2 * hd], Q[it][2 * hd + 1], h, l);
            QA[hd] = __builtin_bit_cast(f16x8, (ml_u32x4){l, h, h, l});
        }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            sa[jt] = ML_MFMAH(KB[jt][0], QA[0], jt == 3 ? zpad : z);
            sb[jt] = ML_MFMAH(KB[jt][1], QA[1], jt == 3 ? zpad : z);
        }
        // ml_softmax's first reads of the scores are inline asm (v_max3_f32), which the compiler's hazard recogniser does not see: a VALU read of an MFMA
        // result needs up to 19 wait states (CDNA3 ISA 4.5) and the hardware does not interlock.  This block depends on all eight accumulators and
        // everything after depends on it; MFMAs retire in order, so 20 wait states behind the last one cover them all (once per query tile: 80 cycles per point).
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(sa[0]), "+v"(sa[1]), "+v"(sa[2]), "+v"(sa[3]), "+v"(sb[0]), "+v"(sb[1]), "+v"(sb[2]), "+v"(sb[3]));
        ml_softmax(sa, cs);
        ml_softmax(sb, cs);
        f32x4 o4[2][2] = {{z, z}, {z, z}};                             // [head][ks]: four independent chains
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 pa[2], pb[2];
            split2h_pack8(make_float4(sa[2 * ks][0], sa[2 * ks][1], sa[2 * ks][2], sa[2 * ks][3]),
                          make_float4(sa[2 * ks + 1][0], sa[2 * ks + 1][1], sa[2 * ks + 1][2], sa[2 * ks + 1][3]), pa[0], pa[1]);
            split2h_pack8(make_float4(sb[2 * ks][0], sb[2 * ks][1], sb[2 * ks][2], sb[2 * ks][3]),
                          make_float4(sb[2 * ks + 1][0], sb[2 * ks + 1][1], sb[2 * ks + 1][2], sb[2 * ks + 1][3]), pb[0], pb[1]);
#define ML_T(PA, PB) o4[0][ks] = ML_MFMAH(pa[PA], Vq[ks][PB], o4[0][ks]); o4[1][ks] = ML_MFMAH(pb[PA], Vq[ks][PB], o4[1][ks]);
            ML_T(1, 0) ML_T(0, 1) ML_T(0, 0)
#undef ML_T
        }
        const f32x4 oa = o4[0][0] + o4[0][1], ob = o4[1][0] + o4[1][1];
        const int col = 16 * w + fr;
        if (PLANES) {
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                unsigned h, l;
                split2h_pair((fr < 8 ? oa[r] : ob[r]) * os, (fr < 8 ? oa[r + 1] : ob[r + 1]) * os, h, l);
                unsigned short* d = Ap + ml_sw(it * 16 + fg * 4 + r, col);           // rows r, r + 1: the same 16-byte unit only if their low bits agree -> two addresses
                unsigned short* d1 = Ap + ml_sw(it * 16 + fg * 4 + r + 1, col);
                d[0] = (unsigned short)(h & 0xffffu); d[ML_PL] = (unsigned short)(l & 0xffffu);
                d1[0] = (unsigned short)(h >> 16); d1[ML_PL] = (unsigned short)(l >> 16);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) As[(it * 16 + fg * 4 + r) * ML_S + col] = (fr < 8 ? oa[r] : ob[r]) * os;
        }
    }
}
// C: head_combine, transposed (rows = this wave's 16 output channels, columns = tokens)
// wc: the wave's head_combine fragments [ks][plane] -- registers (W.c) or, where registers are short, its slice of an LDS copy
template <class WC>
__device__ __forceinline__ void ml_combine(const unsigned short* Ap, WC wc, f32x4 (&y)[4], int fr, int fg) {
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) y[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        f16x8 a[4][2], c[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) c[pl] = wc(ks, pl);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) a[tt][pl] = *reinterpret_cast<const f16x8*>(Ap + pl * ML_PL + ml_sw(tt * 16 + fr, ks * 32 + fg * 8));
#define ML_T(PA, PB) _Pragma("unroll") for (int tt = 0; tt < 4; ++tt) y[tt] = ML_MFMAH(c[PA], a[tt][PB], y[tt]);
        ML_T(1, 0) ML_T(0, 1) ML_T(0, 0)
#undef ML_T
    }
}

// MODE 0: out = X + att Wc^T + bc (residual layer);  1: out = att Wc^T + bc;  2: out = att (head_combine folded downstream);
// MODE 3 (round 5): the direction tail folded in -- out[token] = v . relu(Wf att[token] + bf) + c (models_pointcloud.py:115-117 with the linear
//   chains folded on the host: Wf = net[0] o head_combine, v = so3_reg o net[2]): the 60 x 64 attention tile is in LDS when the layer ends, the
//   hidden layer (64 -> 128 -> 1, 1 MFLOP per point) runs on v_mfma_f32_32x32x16_f16 from two fp16 planes per operand (split_bf16.h; Wf arrives as
//   the planes of Wf (rows times their own powers of two, folded into bf / v on the host) in fragment order, `Wc`; `bc` = [bf (128) | v (128) | c]); 240 bytes per point leave the kernel instead of 15 KB, and
//   linear_relu_dot_ws_kernel<64, 1> (1.74 ms, 2.46 GB read) leaves the path.
// waves per SIMD the layer is compiled for: two (the weight fragments, the score operands of all four token tiles and the V planes are resident)
#ifndef ML_LAYER_WPE
#define ML_LAYER_WPE 2
#endif
template <int MODE>
__global__ void __launch_bounds__(256, ML_LAYER_WPE) mhsa_layer_kernel(long T, const float* __restrict__ X, const float* __restrict__ Wq,
                                                            const float* __restrict__ Wk, const float* __restrict__ Wv,
                                                            const float* __restrict__ Wc, const float* __restrict__ bc,
                                                            float* __restrict__ out, unsigned* __restrict__ ctr) {
    __shared__ unsigned s_grab;
    __shared__ __attribute__((aligned(16))) unsigned short Xp[2 * ML_PL];                     // token tile, two planes
    __shared__ __attribute__((aligned(16))) float Asm[MODE >= 2 ? 64 * ML_S : ML_PL];         // attention tile: fp32 (MODE 2, 3) or two planes
    __shared__ float s_red[16];                                                                 // the weights' maxima
    __shared__ float s_tmax[4];                                                                 // the four waves' shares of the next tile's maximum
    __shared__ __attribute__((aligned(16))) float tail_tab[MODE == 3 ? 256 : 4];              // MODE 3: bf | v
    __shared__ float tail_part[MODE == 3 ? 4 * 64 : 4];                                         // MODE 3: the four waves' shares of the 64 token sums
    float* As = Asm;
    unsigned short* Ap = reinterpret_cast<unsigned short*>(Asm);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;

    MlWeights W;
    ml_load_weights<(MODE < 2)>(W, Wq, Wk, Wv, Wc, s_red, tid, w, fr, fg);
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
    // token rows 60..63 stay zero for the whole kernel (both planes): 2 x 4 rows x 64 fp16 = 256 dwords
    reinterpret_cast<unsigned*>(Xp + (tid / 128) * ML_PL + ML_TOK * 64)[tid % 128] = 0u;

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
    // the maximum of the tile in x0..x3: this wave's share to s_tmax (read behind the next barrier)
#define ML_TILE_MAX()                                                                        \
    {                                                                                        \
        const float m_ = ml_wave_max(ml_max4(x0, ml_max4(x1, ml_max4(x2, ml_max4(x3, 0.f))))); \
        if (lane == 0) s_tmax[w] = m_;                                                       \
    }
    long pt = blockIdx.x, nxt = (long)blockIdx.x + gridDim.x;
    if (pt < T) ML_GLOAD(pt)
    ML_TILE_MAX()
    __syncthreads();
    for (; pt < T;) {
        const int kx = ml_scale_exp(fmaxf(fmaxf(s_tmax[0], s_tmax[1]), fmaxf(s_tmax[2], s_tmax[3])));
        const float sx = ldexpf(1.0f, kx);
        ml_stage4(Xp, tid, x0, sx); ml_stage4(Xp, tid + 256, x1, sx); ml_stage4(Xp, tid + 512, x2, sx);
        if (tid < ML_TOK * ML_C / 4 - 768) ml_stage4(Xp, tid + 768, x3, sx);
        __syncthreads();
        if (nxt < T) ML_GLOAD(nxt)                            // next point's tokens: in flight during the whole layer
        unsigned grabbed = 0u;
        if (ctr && tid == 0) grabbed = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        f32x4 Q[4], Kt[4], V[4];
        ml_project(Xp, W, Q, Kt, V, fr, fg);
        // MODE 2: the attention output itself leaves the kernel; otherwise it stays at <= 2^14 and its power is applied by the consumer's epilogue
        ml_attention<(MODE < 2)>(Q, Kt, V, As, Ap, ml_score_factor(2 * kx + W.kq + W.kk),
                                 ldexpf(1.0f, MODE == 2 ? -13 - kx - W.kv : -13), w, fr, fg);
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
            const float tf = ldexpf(1.0f, -(kx + W.kv));        // (the hidden units' own powers of two are folded into bf and v on the host: dirtail_weight_split)
            // d[tt][v] = 2^(6 + kx + kv) hidden[32 w + 8 (v / 4) + 4 kg + v % 4][token 32 tt + tl]: bias, ReLU, . v, summed over this lane's 16 hidden units,
            // then over the two lane halves and (through LDS, fixed order) over the four waves
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                float sacc = 0.f;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 b4 = *reinterpret_cast<const float4*>(&tail_tab[32 * w + 8 * q4 + 4 * kg]);
                    const float4 v4 = *reinterpret_cast<const float4*>(&tail_tab[128 + 32 * w + 8 * q4 + 4 * kg]);
                    sacc = fmaf(fmaxf(fmaf(d[tt][4 * q4 + 0], tf, b4.x), 0.f), v4.x, sacc);
                    sacc = fmaf(fmaxf(fmaf(d[tt][4 * q4 + 1], tf, b4.y), 0.f), v4.y, sacc);
                    sacc = fmaf(fmaxf(fmaf(d[tt][4 * q4 + 2], tf, b4.z), 0.f), v4.z, sacc);
                    sacc = fmaf(fmaxf(fmaf(d[tt][4 * q4 + 3], tf, b4.w), 0.f), v4.w, sacc);
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
            const float ys = ldexpf(1.0f, -(kx + W.kv + W.kc));
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int tok = tt * 16 + fr;
                if (tok < ML_TOK) {
                    float4 o = make_float4(fmaf(y[tt][0], ys, bias.x), fmaf(y[tt][1], ys, bias.y), fmaf(y[tt][2], ys, bias.z), fmaf(y[tt][3], ys, bias.w));
                    if (MODE == 0) {                       // the residual: the point's own rows again (L2: they were read a layer ago)
                        const float4 rx = *reinterpret_cast<const float4*>(X + pt * (ML_TOK * ML_C) + tok * ML_C + 16 * w + 4 * fg);
                        o.x += rx.x; o.y += rx.y; o.z += rx.z; o.w += rx.w;
                    }
                    *reinterpret_cast<float4*>(dst + tok * ML_C + 16 * w + 4 * fg) = o;
                }
            }
        }
        if (ctr && tid == 0) s_grab = grabbed;
        ML_TILE_MAX()         // the next point's tokens have had the whole layer to arrive
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
// blended when the previous point's layer ends (16 registers across the barrier instead of 48; the tile's maximum is taken there), then staged
// in LDS as two fp16 planes for the projections and as fp32 for the residual; the rows of the NEXT point are requested before the layer's three
// phases start.  LDS: 16.4 (token planes) + 16.4 (attention planes) + 16.4 (head_combine fragments) + 16.4 KB (fp32 tokens), dynamic.
#define ML_INTERP_LDS (2 * 2 * ML_PL * 2 + 4 * 2 * 2 * 64 * 16 + 64 * 64 * 4)
__device__ __forceinline__ int ml_swf(int row, int c4) { return row * 64 + ((c4 ^ (row & 15)) << 2); }      // fp32 tile: 16-byte units XOR-swizzled with the row
__global__ void __launch_bounds__(256, MHSA_INTERP_WGS) mhsa_interp_layer_kernel(int B, int N, int S, const float* __restrict__ F,
                                                                   const int* __restrict__ sched, const float* __restrict__ Wq,
                                                                   const float* __restrict__ Wk, const float* __restrict__ Wv,
                                                                   const float* __restrict__ Wc, const float* __restrict__ bc,
                                                                   float* __restrict__ out, unsigned* __restrict__ ctr) {
    __shared__ unsigned s_grab;
    extern __shared__ __attribute__((aligned(16))) float ml_dyn[];
    __shared__ float s_red[16];
    __shared__ float s_tmax[4];
    unsigned short* Xp = reinterpret_cast<unsigned short*>(ml_dyn);                // token tile, two planes
    unsigned short* Ap = Xp + 2 * ML_PL;                                           // attention tile, two planes
    f16x8* Wcl = reinterpret_cast<f16x8*>(Ap + 2 * ML_PL);                         // head_combine fragments [wave][ks][plane][lane]: 16 registers the
                                                                                   // prefetched coarse rows (48) leave no room for
    float* Xf = reinterpret_cast<float*>(Wcl + 4 * 2 * 2 * 64);                    // the tokens in fp32 (the residual)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    MlWeights W;
    ml_load_weights<true>(W, Wq, Wk, Wv, Wc, s_red, tid, w, fr, fg);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            Wcl[((w * 2 + ks) * 2 + pl) * 64 + lane] = W.c[ks][pl];
            W.c[ks][pl] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};           // dead from here on
        }
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bc) bias = *reinterpret_cast<const float4*>(bc + 16 * w + 4 * fg);
    reinterpret_cast<unsigned*>(Xp + (tid / 128) * ML_PL + ML_TOK * 64)[tid % 128] = 0u;      // token rows 60..63: zero (both planes)

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
    // the blended tile of the point whose rows are in xa / xb / xc (element e = tid + 256 h), and this wave's share of its maximum
    f32x4 xv[4];
#define ML_BLEND(WR)                                                                                       \
    {                                                                                                      \
        const float a0_ = __int_as_float((WR).x), a1_ = __int_as_float((WR).y), a2_ = __int_as_float((WR).z);  \
        float m_ = 0.f;                                                                                    \
        _Pragma("unroll") for (int h = 0; h < 4; ++h) {                                                    \
            xv[h] = ml_blend(xa[h], xb[h], xc[h], a0_, a1_, a2_);                                          \
            m_ = ml_max4(make_float4(xv[h][0], xv[h][1], xv[h][2], xv[h][3]), m_);                         \
        }                                                                                                  \
        m_ = ml_wave_max(m_);                                                                              \
        if (lane == 0) s_tmax[w] = m_;                                                                     \
    }
    ML_BLEND(c_w)
    __syncthreads();
    for (; q < lim;) {
        const long cpt = c_i.x;
        const int kx = ml_scale_exp(fmaxf(fmaxf(s_tmax[0], s_tmax[1]), fmaxf(s_tmax[2], s_tmax[3])));
        const float sx = ldexpf(1.0f, kx);
        // the token tile itself (projections, residual)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int e = tid + 256 * h;
            if (h < 3 || e < ML_TOK * ML_C / 4) {
                ml_stage4(Xp, e, make_float4(xv[h][0], xv[h][1], xv[h][2], xv[h][3]), sx);
                *reinterpret_cast<f32x4*>(&Xf[ml_swf(e >> 4, e & 15)]) = xv[h];
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
        ml_attention<true>(Q, Kt, V, nullptr, Ap, ml_score_factor(2 * kx + W.kq + W.kk), ldexpf(1.0f, -13), w, fr, fg);
        __syncthreads();

        // ---- C: head_combine + bias + residual
        f32x4 y[4];
        ml_combine(Ap, [&](int ks, int pl) { return Wcl[((w * 2 + ks) * 2 + pl) * 64 + lane]; }, y, fr, fg);
        const float ys = ldexpf(1.0f, -(kx + W.kv + W.kc));
        float* dst = out + cpt * (ML_TOK * ML_C);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int tok = tt * 16 + fr;
            if (tok < ML_TOK) {
                const f32x4 rx = *reinterpret_cast<const f32x4*>(&Xf[ml_swf(tok, 4 * w + fg)]);      // the residual X[tok][16 w + 4 fg ..]
                *reinterpret_cast<float4*>(dst + tok * ML_C + 16 * w + 4 * fg) =
                    make_float4(fmaf(y[tt][0], ys, bias.x) + rx[0], fmaf(y[tt][1], ys, bias.y) + rx[1], fmaf(y[tt][2], ys, bias.z) + rx[2], fmaf(y[tt][3], ys, bias.w) + rx[3]);
            }
        }
        if (ctr && tid == 0) s_grab = grabbed;
        ML_BLEND(c_w)         // the next point's rows have had the whole layer to arrive (c_w: its weights since the top of this iteration)
        __syncthreads();      // Xp / Ap / Xf are rewritten by the next point
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
// row-scaled Wf (Wf 128 x 64 = direction_predictor.net[0] o head_combine); tab = [bf (128) | v (128) | c] fp32; out (T, 60) = the anchor weights of so3_mean
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
