// Issue cost of the instructions of the two-plane fp16 split (round 5) on this MI355X, alone and beside v_mfma_f32_32x32x16_f16: per loop iteration 4
// independent MFMAs, each followed by NV instructions of one kind over 8 independent registers.  SIMD cycles per slot at 2.4 GHz nominal.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/vrf profiles/scripts/valu_rates_f16.hip && /tmp/vrf
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NV, int NOM, int DEP = 0>
__global__ void __launch_bounds__(256) kern(int iters, float* out) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f + i); b[i] = (_Float16)(blockIdx.x * 1e-3f + i); }
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    unsigned u[8]; for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 77u + i;
    const float c = blockIdx.x * 1e-4f + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!NOM) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int q = DEP ? (i * NV + k) % DEP : (i * NV + k) % 8;
                if (KIND == 0) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[q]) : "v"(v[q]), "v"(c));
                if (KIND == 1) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[q]) : "v"(u[q]));
                if (KIND == 2) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(v[q]) : "v"(u[q]));
                if (KIND == 3) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(v[q]) : "v"(u[q]));
                if (KIND == 4) asm volatile("v_max_f32 %0, %0, %0 clamp" : "+v"(v[q]));
                if (KIND == 5) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[q]) : "v"(c));
                if (KIND == 6) asm volatile("v_med3_f32 %0, %0, 0, 1.0" : "+v"(v[q]));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    for (int i = 0; i < 8; ++i) s += v[i] + (float)u[i];
    if (s == 123.456f) out[0] = s;
}
template <class K>
static double run(K k, int wgs_per_cu, int iters, float* d) {
    int cus = 0; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    hipLaunchKernelGGL(k, dim3(cus * wgs_per_cu), dim3(256), 0, 0, 16, d);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    hipLaunchKernelGGL(k, dim3(cus * wgs_per_cu), dim3(256), 0, 0, iters, d);
    (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms = 0; (void)hipEventElapsedTime(&ms, s, e);
    return ms * 1e-3 * 2.4e9 / ((double)iters * 4 * wgs_per_cu);
}
#define ROW(KIND, NAME) { printf("%-22s", NAME); \
    printf(" | no MFMA, w/SIMD 1: %5.1f %5.1f  2: %5.1f %5.1f", run(kern<KIND, 4, 1>, 1, it, d), run(kern<KIND, 8, 1>, 1, it, d), run(kern<KIND, 4, 1>, 2, it, d), run(kern<KIND, 8, 1>, 2, it, d)); \
    printf(" | with MFMA, w/SIMD 1: %5.1f %5.1f %5.1f %5.1f  2: %5.1f %5.1f %5.1f %5.1f\n", run(kern<KIND, 0, 0>, 1, it, d), run(kern<KIND, 4, 0>, 1, it, d), run(kern<KIND, 8, 0>, 1, it, d), run(kern<KIND, 12, 0>, 1, it, d), \
           run(kern<KIND, 0, 0>, 2, it, d), run(kern<KIND, 4, 0>, 2, it, d), run(kern<KIND, 8, 0>, 2, it, d), run(kern<KIND, 12, 0>, 2, it, d)); }
int main() {
    float* d; (void)hipMalloc(&d, 64);
    const int it = 20000;
    printf("SIMD cycles (2.4 GHz nominal) per slot.  no MFMA: NV = 4, 8.  with v_mfma_f32_32x32x16_f16: NV = 0, 4, 8, 12\n");
    // dependent chains: the NV instructions of a slot rotate over DEP registers (DEP = 1: every instruction reads the previous one's result)
    for (int rep = 0; rep < 1; ++rep) {
        printf("v_sub_f32, no MFMA, NV = 8, 1 wave/SIMD: rotating over 1 / 2 / 4 / 8 registers: %5.1f %5.1f %5.1f %5.1f   2 waves/SIMD: %5.1f %5.1f %5.1f %5.1f\n",
               run(kern<5, 8, 1, 1>, 1, it, d), run(kern<5, 8, 1, 2>, 1, it, d), run(kern<5, 8, 1, 4>, 1, it, d), run(kern<5, 8, 1, 8>, 1, it, d),
               run(kern<5, 8, 1, 1>, 2, it, d), run(kern<5, 8, 1, 2>, 2, it, d), run(kern<5, 8, 1, 4>, 2, it, d), run(kern<5, 8, 1, 8>, 2, it, d));
        printf("v_max_f32 clamp, same:                                                            %5.1f %5.1f %5.1f %5.1f   2 waves/SIMD: %5.1f %5.1f %5.1f %5.1f\n",
               run(kern<4, 8, 1, 1>, 1, it, d), run(kern<4, 8, 1, 2>, 1, it, d), run(kern<4, 8, 1, 4>, 1, it, d), run(kern<4, 8, 1, 8>, 1, it, d),
               run(kern<4, 8, 1, 1>, 2, it, d), run(kern<4, 8, 1, 2>, 2, it, d), run(kern<4, 8, 1, 4>, 2, it, d), run(kern<4, 8, 1, 8>, 2, it, d));
        printf("v_fma_mix_f32, same:                                                              %5.1f %5.1f %5.1f %5.1f   2 waves/SIMD: %5.1f %5.1f %5.1f %5.1f\n",
               run(kern<3, 8, 1, 1>, 1, it, d), run(kern<3, 8, 1, 2>, 1, it, d), run(kern<3, 8, 1, 4>, 1, it, d), run(kern<3, 8, 1, 8>, 1, it, d),
               run(kern<3, 8, 1, 1>, 2, it, d), run(kern<3, 8, 1, 2>, 2, it, d), run(kern<3, 8, 1, 4>, 2, it, d), run(kern<3, 8, 1, 8>, 2, it, d));
    }
    ROW(0, "v_cvt_pkrtz_f16_f32") ROW(1, "v_cvt_f32_f16") ROW(2, "v_cvt_f32_f16 sdwa") ROW(3, "v_fma_mix_f32") ROW(4, "v_max_f32 clamp") ROW(6, "v_med3_f32") ROW(5, "v_sub_f32")
    return 0;
}
