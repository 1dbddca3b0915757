// Issue rate of v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16 vs the number of independent accumulators per wave and waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mbr profiles/scripts/mfma_bf16_rate.hip && /tmp/mbr
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC, int W32>
__global__ void __launch_bounds__(256) kern(int iters, float* out) {
    f32x4 acc[NACC]; f32x16 acc2[NACC];
    for (int i = 0; i < NACC; ++i) { acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; for (int j = 0; j < 16; ++j) acc2[i][j] = 0.f; }
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(blockIdx.x + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (W32) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc2[i][0];
    if (s == 123.456f) out[0] = s;
}
template <class K>
static void run(const char* name, K k, int nacc, int wgs_per_cu, float* d, double flop) {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int iters = 40000;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(k, dim3(cus * wgs_per_cu), dim3(256), 0, 0, 2000, d);
    hipDeviceSynchronize();
    hipEventRecord(s);
    hipLaunchKernelGGL(k, dim3(cus * wgs_per_cu), dim3(256), 0, 0, iters, d);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms = 0; hipEventElapsedTime(&ms, s, e);
    printf("%s acc %2d waves/SIMD %d: %6.1f nominal cycles per MFMA per SIMD, %7.1f TFLOP/s\n", name, nacc, wgs_per_cu, ms * 1e-3 * 2.4e9 / ((double)iters * nacc * wgs_per_cu),
           (double)cus * 4 * wgs_per_cu * iters * nacc * flop / ms / 1e9);
}
int main() {
    float* d; hipMalloc(&d, 64);
    const double F16 = 2.0 * 16 * 16 * 32, F32 = 2.0 * 32 * 32 * 16;
    run("16x16x32", kern<2, 0>, 2, 1, d, F16); run("16x16x32", kern<4, 0>, 4, 1, d, F16); run("16x16x32", kern<8, 0>, 8, 1, d, F16); run("16x16x32", kern<16, 0>, 16, 1, d, F16);
    run("16x16x32", kern<4, 0>, 4, 2, d, F16); run("16x16x32", kern<8, 0>, 8, 2, d, F16); run("16x16x32", kern<4, 0>, 4, 4, d, F16); run("16x16x32", kern<8, 0>, 8, 4, d, F16);
    run("32x32x16", kern<1, 1>, 1, 1, d, F32); run("32x32x16", kern<2, 1>, 2, 1, d, F32); run("32x32x16", kern<4, 1>, 4, 1, d, F32); run("32x32x16", kern<2, 1>, 2, 2, d, F32); run("32x32x16", kern<4, 1>, 4, 2, d, F32);
    return 0;
}
