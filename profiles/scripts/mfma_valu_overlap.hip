// Does VALU work issue in the shadow of the matrix core on this MI355X?  Per loop iteration: NACC independent MFMAs + NV independent v_fma_f32
// (or NV/2 v_pk_fma_f32) + optionally NL ds_read_b128; TFLOP/s of the MFMAs alone is reported.  Flat in NV = the VALU work hides.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mvo profiles/scripts/mfma_valu_overlap.hip && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int W32, int NACC, int NV, int PK, int NL, int DEP>
__global__ void __launch_bounds__(256) kern(int iters, float* out) {
    __shared__ float4 lds[1024];
    lds[threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    f32x16 acc32[NACC]; f32x4 acc16[NACC];
    for (int i = 0; i < NACC; ++i) { for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f; acc16[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f + 1.f;
    float v[16]; for (int i = 0; i < 16; ++i) v[i] = a + i;
    f32x2 pv[8]; for (int i = 0; i < 8; ++i) pv[i] = (f32x2){a + i, b + i};
    float4 lacc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            // DEP: the MFMA's A operand comes out of the VALU work of this iteration (the inter conv's weights)
            const float aa = DEP ? (PK ? pv[i % 8][0] : v[i % 16]) : a;
            if (W32) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa, b, acc32[i], 0, 0, 0);
            else acc16[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa, b, acc16[i], 0, 0, 0);
            if (PK) {
#pragma unroll
                for (int k = 0; k < NV / 2 / NACC; ++k) { const int q = (i * (NV / 2 / NACC) + k) % 8; asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pv[q]) : "v"(pv[(q + 1) % 8])); }
            } else {
#pragma unroll
                for (int k = 0; k < NV / NACC; ++k) { const int q = (i * (NV / NACC) + k) % 16; asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[q]) : "v"(b)); }
            }
#pragma unroll
            for (int k = 0; k < NL / NACC; ++k) { const float4 t = lds[(threadIdx.x + 64 * (k + i) + it) & 1023]; lacc.x += t.x; }
        }
    }
    float s = lacc.x;
    for (int i = 0; i < NACC; ++i) { for (int j = 0; j < 16; ++j) s += acc32[i][j]; s += acc16[i][0]; }
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 8; ++i) s += pv[i][0] + pv[i][1];
    if (s == 123.456f) out[0] = s;
}
template <class K>
static void run(const char* name, K k, int wgs_per_cu, int nacc, double flop_per_mfma, int iters, float* d) {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(k, dim3(cus * wgs_per_cu), dim3(256), 0, 0, 16, d);
    hipDeviceSynchronize();
    hipEventRecord(s);
    hipLaunchKernelGGL(k, dim3(cus * wgs_per_cu), dim3(256), 0, 0, iters, d);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms = 0; hipEventElapsedTime(&ms, s, e);
    const double flops = (double)cus * wgs_per_cu * 4 * iters * nacc * flop_per_mfma;
    const double cyc_per_mfma_per_simd = ms * 1e-3 * 2.4e9 / ((double)iters * nacc * wgs_per_cu);
    printf("%-44s waves/SIMD %d  %8.2f ms  %7.1f TFLOP/s   %.1f cycles (2.4 GHz) of SIMD time per MFMA\n", name, wgs_per_cu, ms, flops / ms / 1e9, cyc_per_mfma_per_simd);
}
#define RUN(W32, NACC, NV, PK, NL, DEP, WG) run(#W32 " acc " #NACC " valu " #NV " pk " #PK " lds " #NL " dep " #DEP, kern<W32, NACC, NV, PK, NL, DEP>, WG, NACC, W32 ? F32 : F16, iters, d)
int main() {
    float* d; hipMalloc(&d, 64);
    const double F16 = 2.0 * 16 * 16 * 4, F32 = 2.0 * 32 * 32 * 2;
    const int iters = 20000;
    printf("32x32x2 (64 cycles), 2 accumulators per wave: VALU ops per iteration (= per 2 MFMAs)\n");
    RUN(1, 2, 0, 0, 0, 0, 1); RUN(1, 2, 8, 0, 0, 0, 1); RUN(1, 2, 16, 0, 0, 0, 1); RUN(1, 2, 24, 0, 0, 0, 1); RUN(1, 2, 32, 0, 0, 0, 1); RUN(1, 2, 48, 0, 0, 0, 1);
    RUN(1, 2, 0, 0, 0, 0, 2); RUN(1, 2, 8, 0, 0, 0, 2); RUN(1, 2, 16, 0, 0, 0, 2); RUN(1, 2, 24, 0, 0, 0, 2); RUN(1, 2, 32, 0, 0, 0, 2); RUN(1, 2, 48, 0, 0, 0, 2);
    printf("the same with the MFMA's A operand produced by that VALU work\n");
    RUN(1, 2, 8, 0, 0, 1, 1); RUN(1, 2, 16, 0, 0, 1, 1); RUN(1, 2, 8, 0, 0, 1, 2); RUN(1, 2, 16, 0, 0, 1, 2); RUN(1, 2, 32, 0, 0, 1, 2);
    printf("packed fp32 (v_pk_fma_f32: two lanes' worth per instruction)\n");
    RUN(1, 2, 16, 1, 0, 0, 1); RUN(1, 2, 32, 1, 0, 0, 1); RUN(1, 2, 16, 1, 0, 0, 2); RUN(1, 2, 32, 1, 0, 0, 2);
    printf("LDS reads (ds_read_b128) per iteration\n");
    RUN(1, 2, 0, 0, 2, 0, 1); RUN(1, 2, 0, 0, 4, 0, 1); RUN(1, 2, 0, 0, 2, 0, 2); RUN(1, 2, 0, 0, 4, 0, 2); RUN(1, 2, 16, 0, 2, 0, 2);
    printf("16x16x4 (32 cycles), 4 accumulators per wave: VALU ops per iteration (= per 4 MFMAs)\n");
    RUN(0, 4, 0, 0, 0, 0, 2); RUN(0, 4, 8, 0, 0, 0, 2); RUN(0, 4, 16, 0, 0, 0, 2); RUN(0, 4, 32, 0, 0, 0, 2); RUN(0, 4, 0, 0, 0, 0, 4); RUN(0, 4, 16, 0, 0, 0, 4); RUN(0, 4, 32, 0, 0, 0, 4);
    return 0;
}
