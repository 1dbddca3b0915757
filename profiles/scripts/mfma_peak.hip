// Achievable fp32 matrix-core rate of this MI355X: independent v_mfma_f32_16x16x4_f32 / 32x32x2 chains in registers, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak profiles/scripts/mfma_peak.hip && /tmp/mfma_peak
// Prints TFLOP/s for several (waves per SIMD, accumulators per wave) shapes and run lengths (short launches run at a higher clock than the
// sustained one: the long runs are the ones that bound a 5 - 8 ms kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) k16(int iters, float* out) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[0] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) k32(int iters, float* out) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 123.456f) out[0] = s;
}
template <class K>
static void run(const char* name, K kern, int wgs_per_cu, int nacc, double flop_per_mfma, int iters, float* d) {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(kern, dim3(cus * wgs_per_cu), dim3(256), 0, 0, 16, d);
    hipDeviceSynchronize();
    hipEventRecord(s);
    hipLaunchKernelGGL(kern, dim3(cus * wgs_per_cu), dim3(256), 0, 0, iters, d);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms = 0; hipEventElapsedTime(&ms, s, e);
    const double flops = (double)cus * wgs_per_cu * 4 * iters * nacc * flop_per_mfma;
    printf("%-10s waves/SIMD %d  acc/wave %d  %8.2f ms  %7.1f TFLOP/s  (%.0f MHz-equivalent at 256 FLOP/clk/CU)\n", name, wgs_per_cu, nacc, ms, flops / ms / 1e9,
           flops / ms / 1e3 / (256.0 * cus));
}
int main() {
    float* d; hipMalloc(&d, 64);
    const double F16 = 2.0 * 16 * 16 * 4, F32 = 2.0 * 32 * 32 * 2;
    for (int iters : {2000, 20000, 100000}) {
        printf("-- %d iterations\n", iters);
        run("16x16x4", k16<4>, 1, 4, F16, iters, d);
        run("16x16x4", k16<8>, 1, 8, F16, iters, d);
        run("16x16x4", k16<4>, 2, 4, F16, iters, d);
        run("16x16x4", k16<8>, 2, 8, F16, iters, d);
        run("16x16x4", k16<4>, 4, 4, F16, iters, d);
        run("32x32x2", k32<2>, 1, 2, F32, iters, d);
        run("32x32x2", k32<4>, 2, 4, F32, iters, d);
    }
    return 0;
}
