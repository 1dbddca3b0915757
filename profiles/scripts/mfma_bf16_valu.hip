// What does VALU work cost beside v_mfma_f32_16x16x32_bf16 on this MI355X?  Per loop iteration: 4 independent MFMAs, each followed by NV VALU
// instructions of one kind (independent chains over 8 registers).  Reports SIMD cycles per MFMA (at 2.4 GHz nominal; the chip clocks lower
// under load) for 1, 2 and 4 waves per SIMD.  Flat in NV = the VALU work hides behind the matrix pipe.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mbv profiles/scripts/mfma_bf16_valu.hip && /tmp/mbv
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int NV, int NOM>
__global__ void __launch_bounds__(256) kern(int iters, float* out) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(blockIdx.x + i); }
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    f32x2 pv[8]; for (int i = 0; i < 8; ++i) pv[i] = (f32x2){v[i], v[i] + 1.f};
    const float c = blockIdx.x * 1e-4f + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!NOM) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int q = (i * NV + k) % 8;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[q]) : "v"(c));
                if (KIND == 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(v[q]));
                if (KIND == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(c), "v"(0x07060302u));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pv[q]) : "v"(pv[(q + 1) % 8]));
                if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pv[q]) : "v"(pv[(q + 1) % 8]));
                if (KIND == 5) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[q]) : "v"(c));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    for (int i = 0; i < 8; ++i) s += v[i] + pv[i][0] + pv[i][1];
    if (s == 123.456f) out[0] = s;
}
template <class K>
static double run(K k, int wgs_per_cu, int iters, float* d) {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(k, dim3(cus * wgs_per_cu), dim3(256), 0, 0, 16, d);
    hipDeviceSynchronize();
    hipEventRecord(s);
    hipLaunchKernelGGL(k, dim3(cus * wgs_per_cu), dim3(256), 0, 0, iters, d);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms = 0; hipEventElapsedTime(&ms, s, e);
    return ms * 1e-3 * 2.4e9 / ((double)iters * 4 * wgs_per_cu);      // SIMD cycles per (MFMA + NV VALU) slot
}
#define ROW(KIND, NAME) { printf("%-14s", NAME); \
    printf(" | w/SIMD 1:"); printf(" %5.1f", run(kern<KIND, 0, 0>, 1, it, d)); printf(" %5.1f", run(kern<KIND, 2, 0>, 1, it, d)); printf(" %5.1f", run(kern<KIND, 3, 0>, 1, it, d)); printf(" %5.1f", run(kern<KIND, 4, 0>, 1, it, d)); printf(" %5.1f", run(kern<KIND, 6, 0>, 1, it, d)); printf(" %5.1f", run(kern<KIND, 8, 0>, 1, it, d)); \
    printf(" | 2:"); printf(" %5.1f", run(kern<KIND, 0, 0>, 2, it, d)); printf(" %5.1f", run(kern<KIND, 2, 0>, 2, it, d)); printf(" %5.1f", run(kern<KIND, 3, 0>, 2, it, d)); printf(" %5.1f", run(kern<KIND, 4, 0>, 2, it, d)); printf(" %5.1f", run(kern<KIND, 6, 0>, 2, it, d)); printf(" %5.1f", run(kern<KIND, 8, 0>, 2, it, d)); \
    printf(" | 4:"); printf(" %5.1f", run(kern<KIND, 0, 0>, 4, it, d)); printf(" %5.1f", run(kern<KIND, 3, 0>, 4, it, d)); printf(" %5.1f", run(kern<KIND, 6, 0>, 4, it, d)); \
    printf(" | no MFMA, 2 w/SIMD:"); printf(" %5.1f", run(kern<KIND, 3, 1>, 2, it, d)); printf(" %5.1f", run(kern<KIND, 6, 1>, 2, it, d)); printf("\n"); }
int main() {
    float* d; hipMalloc(&d, 64);
    const int it = 20000;
    printf("SIMD cycles (2.4 GHz nominal) per slot = 1 x v_mfma_f32_16x16x32_bf16 + NV VALU instructions; NV = 0 2 3 4 6 8 (1 and 2 waves/SIMD), 0 3 6 (4 waves/SIMD); last: NV = 3 6 without the MFMA\n");
    ROW(0, "v_fma_f32") ROW(1, "v_and_b32") ROW(2, "v_perm_b32") ROW(5, "v_sub_f32") ROW(3, "v_pk_add_f32") ROW(4, "v_pk_fma_f32")
    return 0;
}
