// Shared by so3conv_x.hip and so3conv_y.hip: the inline-asm load helpers with counted waits and step 2 of the 32x32x16 kernels.
#pragma once
#include "common.h"
#include "split_bf16.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NA 60
#define KS 24

typedef const void __attribute__((address_space(1)))* x_gptr;
typedef void __attribute__((address_space(3)))* x_lptr;

// Neighbour-table reads as inline asm + counted waits.  Written as plain LDS loads, the compiler orders every one of them behind s_waitcnt vmcnt(0) when
// an LDS-direct load is in flight (it cannot prove that the table and the staging tile are disjoint, not even as separate LDS objects): the rows requested
// at the top of a chunk-step were drained a few instructions later, and the gather latency was paid in full every step.  LDS operations complete in
// order, so "all but the N youngest" is exact; operations the compiler issues in between only make a wait stricter.
#define X_LDS_READ128(dst, addr, off) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"((addr) + (unsigned)(off)))
template <int N> __device__ __forceinline__ void x_lds_wait(f32x4& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N> __device__ __forceinline__ void x_lds_wait2(f32x4& a, f32x4& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }

__device__ __forceinline__ bf16x4 x_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3)))*)p);
}
__device__ __forceinline__ void x_wload(f32x4& dst, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p)); }
// the same with a wave-uniform base (SGPR pair) + 32-bit lane offset + k KiB immediate
__device__ __forceinline__ void x_wload_s(f32x4& dst, unsigned voff, const void* sbase, int k) {
    if (k == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase));
    else if (k == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(dst) : "v"(voff), "s"(sbase));
    else asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(dst) : "v"(voff), "s"(sbase));
}
template <int N> __device__ __forceinline__ void x_wwait6(f32x4 (&v)[2][3]) {
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[0][2]), "+v"(v[1][0]), "+v"(v[1][1]), "+v"(v[1][2]) : "n"(N));
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CIN, int COUT, int PAD = 44>      // PAD: (KH + PAD) / 4 must be odd (the 32 rows of a B-fragment read start in different 16-byte slots)
struct X32Step2 {
    static constexpr int MT2 = COUT / 32, CH = CIN / 2, KH = CH * KS, S = KH + PAD;
    static constexpr int NSW = KH / 16 / 4;         // K steps per wave and half
    f32x4 ra[2][MT2][3];
    __device__ __forceinline__ void issue(int i, const bf16x8* __restrict__ Wq, int wave, int lane) {
        const int h = i / NSW, c = i % NSW;
        const char* b0 = reinterpret_cast<const char*>(Wq) + ((size_t)(h * (KH / 16) + wave + 4 * c) * MT2) * 3 * 1024;
        const unsigned vo = (unsigned)lane * 16u;
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) x_wload_s(ra[i & 1][mt][pl], vo, b0 + mt * 3 * 1024, pl);
    }
    template <int N> __device__ __forceinline__ void wait(f32x4 (&v)[MT2][3]) {
        if constexpr (MT2 == 2) asm volatile("s_waitcnt vmcnt(%6)" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[0][2]), "+v"(v[1][0]), "+v"(v[1][1]), "+v"(v[1][2]) : "n"(N));
        else asm volatile("s_waitcnt vmcnt(%3)" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[0][2]) : "n"(N));
    }
    template <int H>
    __device__ __forceinline__ void half(f32x16 (&y)[MT2], const float* X1s, const bf16x8* __restrict__ Wq, int wave, int lane) {
        const int an = lane & 31, kg = lane >> 5;
#pragma unroll
        for (int c = 0; c < NSW; ++c) {
            const int i = H * NSW + c;
            asm volatile("" ::: "memory");                 // keeps the X1 reads (and their splits) of later steps from being hoisted
            const float* xr = &X1s[an * S + (wave + 4 * c) * 16 + kg * 8];
            bf16x8 bq[3];
            split3_pack8p(*reinterpret_cast<const float4*>(xr), *reinterpret_cast<const float4*>(xr + 4), bq[0], bq[1], bq[2]);
            if (i + 1 < 2 * NSW) { issue(i + 1, Wq, wave, lane); wait<3 * MT2>(ra[i & 1]); }
            else wait<0>(ra[i & 1]);
            f32x4 (&ac)[MT2][3] = ra[i & 1];
#define X_TERM(PA, PB) _Pragma("unroll") for (int mt = 0; mt < MT2; ++mt) \
    y[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ac[mt][PA]), bq[PB], y[mt], 0, 0, 0);
            X_TERM(2, 0) X_TERM(0, 2) X_TERM(1, 1) X_TERM(1, 0) X_TERM(0, 1) X_TERM(0, 0)
#undef X_TERM
        }
    }
};

