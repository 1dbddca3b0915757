// v_mfma_f32_4x4x1_16b_f32 on gfx950: operand / result lane layout and issue rate.
// Hypothesis checked: 16 independent blocks of a 4x4 outer product; lane l supplies A[block l/4][row l%4] and B[block l/4][col l%4];
// result register r of lane l = D[block l/4][row r][col l%4].
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/m441 profiles/scripts/mfma_4x4x1.hip && /tmp/m441
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}
template <int CH>
__global__ void rate(int iters, float* out, long long* cyc) {
    const int l = threadIdx.x & 63;
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = l * 0.001f, b = 1.0f;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int CH> static void run_rate(int waves, float* d_o, long long* d_c) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<CH>, dim3(256), dim3(64 * waves), 0, 0, 100, d_o, d_c);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<CH>, dim3(256), dim3(64 * waves), 0, 0, iters, d_o, d_c);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 8 * CH;        // MFMAs per wave
    // waves per SIMD = waves / 4 (one workgroup per CU)
    const double per_simd = n * (waves >= 4 ? waves / 4.0 : 1.0);
    printf("chains %d, waves/CU %2d: %.3f ms -> %.2f ns per MFMA per SIMD = %.1f cycles at 2.4 GHz\n", CH, waves, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}
int main() {
    float ha[64], hb[64], hd[256];
    for (int i = 0; i < 64; ++i) { ha[i] = 1.0f + i; hb[i] = 100.0f + 3 * i; }
    float *a, *b, *d; long long* c;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1 << 22); hipMalloc(&c, 8);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) bad += hd[l * 4 + r] != ha[4 * (l / 4) + r] * hb[l];
    printf("layout hypothesis D[l][r] = A[4 (l/4) + r] * B[l]: %d mismatches of 256\n", bad);
    if (bad) for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, hd[l * 4], hd[l * 4 + 1], hd[l * 4 + 2], hd[l * 4 + 3]);
    run_rate<1>(4, d, c); run_rate<2>(4, d, c); run_rate<4>(4, d, c); run_rate<8>(4, d, c);
    run_rate<4>(8, d, c); run_rate<4>(16, d, c);
    return 0;
}
