// Exact split of fp32 values into three bf16 values (hi / mid / lo = the top, middle and bottom 8 mantissa bits, by truncation: v = hi + mid + lo
// with no rounding anywhere) -- the operand format of the "fp32 products on the bf16 matrix cores" kernels.  Shared by
// so3conv.hip (step 2 of the inter conv) and so3conv_x.hip (both steps).
#pragma once
#include <hip/hip_runtime.h>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

// 8 consecutive fp32 values -> 3 planes x 8 bf16
__device__ __forceinline__ void split3_pack8(const float4 v0, const float4 v1, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = __float_as_uint(v[i]);
        const float r = v[i] - __uint_as_float(h[i] & 0xffff0000u);
        m[i] = __float_as_uint(r);
        l[i] = __float_as_uint(r - __uint_as_float(m[i] & 0xffff0000u));
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // v_perm_b32: bytes 2, 3 of the even element below bytes 2, 3 of the odd one
#define ETCH_PK(a) (u32x4){__builtin_amdgcn_perm(a[1], a[0], 0x07060302u), __builtin_amdgcn_perm(a[3], a[2], 0x07060302u), \
                           __builtin_amdgcn_perm(a[5], a[4], 0x07060302u), __builtin_amdgcn_perm(a[7], a[6], 0x07060302u)}
    const u32x4 ph = ETCH_PK(h), pm = ETCH_PK(m), pl = ETCH_PK(l);
#undef ETCH_PK
    hi = __builtin_bit_cast(bf16x8, ph); mid = __builtin_bit_cast(bf16x8, pm); lo = __builtin_bit_cast(bf16x8, pl);
}

// 4 consecutive fp32 values -> 3 planes x 4 bf16 (8 bytes each)
__device__ __forceinline__ void split3_pack4(const float4 v4, uint2& hi, uint2& mid, uint2& lo) {
    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = __float_as_uint(v[i]);
        const float r = v[i] - __uint_as_float(h[i] & 0xffff0000u);
        m[i] = __float_as_uint(r);
        l[i] = __float_as_uint(r - __uint_as_float(m[i] & 0xffff0000u));
    }
    hi = make_uint2(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u));
    mid = make_uint2(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u));
    lo = make_uint2(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u));
}

// The same split with the two residual subtractions of a PAIR of values as one packed instruction (v_pk_add_f32 with negated second operand): 4.5
// instead of 5.5 VALU instructions per value.  pairs[i] = (a_i, b_i): the a's and the b's are split into their own fragments (two 8-element
// fragments at once, e.g. the two kernel-point tiles of a weight chunk).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pack8_pairs(const f32x2 (&v)[8], bf16x8 (&a)[3], bf16x8 (&b)[3]) {
    unsigned ha[8], ma[8], la[8], hb[8], mb[8], lb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        ha[i] = __float_as_uint(v[i].x); hb[i] = __float_as_uint(v[i].y);
        const f32x2 r = v[i] - (f32x2){__uint_as_float(ha[i] & 0xffff0000u), __uint_as_float(hb[i] & 0xffff0000u)};
        ma[i] = __float_as_uint(r.x); mb[i] = __float_as_uint(r.y);
        const f32x2 q = r - (f32x2){__uint_as_float(ma[i] & 0xffff0000u), __uint_as_float(mb[i] & 0xffff0000u)};
        la[i] = __float_as_uint(q.x); lb[i] = __float_as_uint(q.y);
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define ETCH_PK(x) (u32x4){__builtin_amdgcn_perm(x[1], x[0], 0x07060302u), __builtin_amdgcn_perm(x[3], x[2], 0x07060302u), \
                           __builtin_amdgcn_perm(x[5], x[4], 0x07060302u), __builtin_amdgcn_perm(x[7], x[6], 0x07060302u)}
    a[0] = __builtin_bit_cast(bf16x8, ETCH_PK(ha)); a[1] = __builtin_bit_cast(bf16x8, ETCH_PK(ma)); a[2] = __builtin_bit_cast(bf16x8, ETCH_PK(la));
    b[0] = __builtin_bit_cast(bf16x8, ETCH_PK(hb)); b[1] = __builtin_bit_cast(bf16x8, ETCH_PK(mb)); b[2] = __builtin_bit_cast(bf16x8, ETCH_PK(lb));
#undef ETCH_PK
}
// 8 consecutive values of one row, residuals of neighbouring values paired
__device__ __forceinline__ void split3_pack8p(const float4 v0, const float4 v1, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    const f32x2 v[4] = {{v0.x, v0.y}, {v0.z, v0.w}, {v1.x, v1.y}, {v1.z, v1.w}};
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[2 * i] = __float_as_uint(v[i].x); h[2 * i + 1] = __float_as_uint(v[i].y);
        const f32x2 r = v[i] - (f32x2){__uint_as_float(h[2 * i] & 0xffff0000u), __uint_as_float(h[2 * i + 1] & 0xffff0000u)};
        m[2 * i] = __float_as_uint(r.x); m[2 * i + 1] = __float_as_uint(r.y);
        const f32x2 q = r - (f32x2){__uint_as_float(m[2 * i] & 0xffff0000u), __uint_as_float(m[2 * i + 1] & 0xffff0000u)};
        l[2 * i] = __float_as_uint(q.x); l[2 * i + 1] = __float_as_uint(q.y);
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define ETCH_PK(x) (u32x4){__builtin_amdgcn_perm(x[1], x[0], 0x07060302u), __builtin_amdgcn_perm(x[3], x[2], 0x07060302u), \
                           __builtin_amdgcn_perm(x[5], x[4], 0x07060302u), __builtin_amdgcn_perm(x[7], x[6], 0x07060302u)}
    hi = __builtin_bit_cast(bf16x8, ETCH_PK(h)); mid = __builtin_bit_cast(bf16x8, ETCH_PK(m)); lo = __builtin_bit_cast(bf16x8, ETCH_PK(l));
#undef ETCH_PK
}

// ---- two-plane fp16 split (round 5): x = h + l with h = fp16(x) and l = fp16(x - h), BOTH rounded to nearest (v_cvt_pk_f16_f32): 23 significant bits
// above 2^-13 and an absolute floor of 2^-25 below (fp16 subnormals, which the gfx950 matrix cores keep: profiles/r05_f16_two_plane_split.txt) -- for
// O(1) operands a product (l*h + h*l + h*h, three v_mfma_f32_32x32x16_f16) carries the error of the fp32 MFMA, with half the matrix instructions, 4
// instead of 6 bytes per element and 4 instead of 5.5 VALU operations per split value of the exact three-plane bf16 split above.
// Rounding to nearest, not truncation: a truncated residual makes every represented value SMALLER in magnitude (2^-23 on average), a bias that is
// coherent over a whole batch -- invisible in any single product, but the direction loss's weight gradients cancel to 1e-7 of their terms and a
// coherent 2e-7 came out of them as a 10 - 50 % error (profiles/r05_weight_gradient_accumulation.txt).  Rounded planes have zero-mean residuals.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2h_pair(float a, float b, unsigned& h, unsigned& l) {
    h = __builtin_bit_cast(unsigned, (f16x2){(_Float16)a, (_Float16)b});
    // l = fp16 half * (-1) + x in one instruction each (v_fma_mix_f32 reads the half directly; exact: x - h has at most 14 significant bits) instead of
    // a conversion and a subtraction: 4 instead of 6 VALU operations per pair
    // IN PLACE (the result overwrites the register that holds x): the compiler's hazard recogniser does not see inline asm, and a fresh output register
    // could be one that an MFMA issued a moment ago is still reading as its accumulator input (write-after-read, up to 7 wait states for an 8-pass
    // MFMA, no hardware interlock; etch_amd/isa_lint.py reported exactly that in csrc/mhsa_layer.hip).  A register that holds a live value cannot be
    // one, and where x is needed afterwards the copy is a compiler-emitted move, which gets its wait states.  Reads: x was read by the conversion above.
#ifndef ETCH_SPLIT2H_MIXF16
    float la = a, lb = b;
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(la) : "v"(h));
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lb) : "v"(h));
    l = __builtin_bit_cast(unsigned, (f16x2){(_Float16)la, (_Float16)lb});
#else
    // Round 6, measured and NOT adopted: v_fma_mixlo_f16 / v_fma_mixhi_f16 round the product-sum once to fp16 straight into the low / high half of the
    // packed pair -- three instructions per pair instead of four, bit for bit the same planes (2^20 random pairs incl. subnormal residuals, +-0, 65 504:
    // scratch/mix/mixtest.hip) -- and SLOWER: the attention layers 3.03 / 2.75 against 2.96 / 2.63 ms, the inter conv 9.13 against 8.99 ms on the same
    // box.  The second instruction merges into the half-written register of the first: a dependent pair per value pair where the form above has two
    // independent instructions (profiles/r06_inter_conv_persistent.txt, section 3).
    float la = a;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(la) : "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(la) : "v"(h), "v"(b));
    l = __builtin_bit_cast(unsigned, la);
#endif
}
// 4 consecutive fp32 values -> 2 planes x 4 fp16 (8 bytes each)
__device__ __forceinline__ void split2h_pack4(const float4 v, uint2& h, uint2& l) {
    split2h_pair(v.x, v.y, h.x, l.x);
    split2h_pair(v.z, v.w, h.y, l.y);
}
// 8 consecutive fp32 values -> 2 planes x 8 fp16
__device__ __forceinline__ void split2h_pack8(const float4 v0, const float4 v1, f16x8& h, f16x8& l) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    unsigned hh[4], ll[4];
    split2h_pair(v0.x, v0.y, hh[0], ll[0]); split2h_pair(v0.z, v0.w, hh[1], ll[1]);
    split2h_pair(v1.x, v1.y, hh[2], ll[2]); split2h_pair(v1.z, v1.w, hh[3], ll[3]);
    h = __builtin_bit_cast(f16x8, (u32x4_){hh[0], hh[1], hh[2], hh[3]}); l = __builtin_bit_cast(f16x8, (u32x4_){ll[0], ll[1], ll[2], ll[3]});
}

// ---- bringing an operand into the fp16 planes' range (round 5).  The planes carry 23 bits only where |h| >= 2^-2, so a kernel whose operands have no
// known scale multiplies a tile / a row by the power of two that puts its maximum into [8, 16) and takes the power out again where a scalar is applied
// anyway (exact).  v_max3_f32 as written: fmaxf chains carry a canonicalising v_max_f32 x, x per operand that comes from memory.
// (in place, like the splits above: an inline-asm result never lands in a register the compiler considers free)
__device__ __forceinline__ float etch_max3abs(float a, float b, float c) { asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(c) : "v"(a), "v"(b)); return c; }
__device__ __forceinline__ float etch_max4abs(const float4 v, float m) { return etch_max3abs(v.z, v.w, etch_max3abs(v.x, v.y, m)); }
// k with m 2^k in [8, 16) (m = 0 or subnormal: 0; capped so that 2^k is a float)
__device__ __forceinline__ int etch_scale_exp(float m) {
    const int e = (int)((__float_as_uint(m) >> 23) & 0xffu);
    return e == 0 ? 0 : (130 - e < 120 ? 130 - e : 120);
}
// maximum of a non-negative value over aligned groups of W = 8 / 16 / 32 consecutive lanes (DPP mirrors inside rows of 16, a row swap for 32)
template <int W>
__device__ __forceinline__ float etch_group_max(float v) {
#define ETCH_DPP(C) v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), (C), 0xf, 0xf, true)));
    ETCH_DPP(0xB1) ETCH_DPP(0x4E) ETCH_DPP(0x141)
    if (W >= 16) { ETCH_DPP(0x140) }
#undef ETCH_DPP
    if (W >= 32) {
        typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
        const u32x2_ r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    return v;
}
// maximum of a non-negative value over the whole wave
__device__ __forceinline__ float etch_wave_max(float v) {
    v = etch_group_max<32>(v);
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    const u32x2_ r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
