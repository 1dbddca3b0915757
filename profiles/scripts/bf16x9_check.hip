// fp32 products on the bf16 matrix cores: every fp32 operand split EXACTLY into three bf16 values (8 + 8 + 8 mantissa bits, by truncation), the nine
// (or the six largest) cross products accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Part 1: error of a 32 x 32 x K product against fp64, next to the
// fp32 MFMA (v_mfma_f32_32x32x2_f32) and a host fmaf chain.  Part 2: issue rate of the bf16 MFMA with VALU fillers beside it.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bx9 profiles/scripts/bf16x9_check.hip && /tmp/bx9
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float a, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    const unsigned ua = __float_as_uint(a);
    const float fh = __uint_as_float(ua & 0xffff0000u);
    const float r = a - fh;
    const unsigned ur = __float_as_uint(r);
    const float fm = __uint_as_float(ur & 0xffff0000u);
    const float l = r - fm;
    hi = ua >> 16; mid = ur >> 16; lo = __float_as_uint(l) >> 16;
}
// one wave: C[32][32] = A[32][K] * B[K][32]; A row-major, B row-major
template <int NTERM>
__global__ void gemm_bx(const float* A, const float* B, float* C, int K) {
    const int l = threadIdx.x, i = l & 31, kg = l >> 5;
    f32x16 acc; for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 a[3], b[3];
        for (int e = 0; e < 8; ++e) {
            unsigned short h, m, lo;
            split3(A[i * K + k0 + 8 * kg + e], h, m, lo); a[0][e] = h; a[1][e] = m; a[2][e] = lo;
            split3(B[(k0 + 8 * kg + e) * 32 + i], h, m, lo); b[0][e] = h; b[1][e] = m; b[2][e] = lo;
        }
        // small terms first
        if (NTERM >= 9) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], acc, 0, 0, 0); }
        if (NTERM >= 8) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[1], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[2], acc, 0, 0, 0); }
        if (NTERM >= 6) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
    for (int v = 0; v < 16; ++v) C[((v & 3) + 8 * (v >> 2) + 4 * kg) * 32 + i] = acc[v];
}
__global__ void gemm_f32(const float* A, const float* B, float* C, int K) {
    const int l = threadIdx.x, i = l & 31, kg = l >> 5;
    f32x16 acc; for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k0 + kg], B[(k0 + kg) * 32 + i], acc, 0, 0, 0);
    for (int v = 0; v < 16; ++v) C[((v & 3) + 8 * (v >> 2) + 4 * kg) * 32 + i] = acc[v];
}
// issue rate: NACC independent bf16 MFMA chains + NV v_fma_f32 fillers per MFMA
template <int NACC, int NV>
__global__ void __launch_bounds__(256) rate(int iters, float* out) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    bf16x8 a, b; for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3f80 + threadIdx.x % 7); b[e] = (short)(0x3f00 + e); }
    float f[8]; for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 1e-3f + i;
    const float c = blockIdx.x * 1e-4f + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[(i * NV + k) % 8]) : "v"(c));
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int v = 0; v < 16; ++v) s += acc[i][v];
    for (int i = 0; i < 8; ++i) s += f[i];
    if (s == 123.456f) out[0] = s;
}
template <class K> static void run_rate(const char* name, K k, int wgs, int nacc, int iters, float* d) {
    int cus = 0; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    hipLaunchKernelGGL(k, dim3(cus * wgs), dim3(256), 0, 0, 16, d); (void)hipDeviceSynchronize();
    (void)hipEventRecord(s); hipLaunchKernelGGL(k, dim3(cus * wgs), dim3(256), 0, 0, iters, d); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms = 0; (void)hipEventElapsedTime(&ms, s, e);
    const double flops = (double)cus * wgs * 4 * iters * nacc * 2.0 * 32 * 32 * 16;
    printf("%-28s waves/SIMD %d  %7.2f ms  %7.1f TFLOP/s bf16  = %6.1f TFLOP/s of fp32 products at 9 terms, %6.1f at 6   (%.1f cycles per MFMA and SIMD at 2.4 GHz)\n", name, wgs, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 9, flops / ms / 1e9 / 6, ms * 1e-3 * 2.4e9 / ((double)iters * nacc * wgs));
}
int main() {
    const int K = 768;
    std::vector<float> A(32 * K), B(K * 32), C(32 * 32);
    float *dA, *dB, *dC; (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, C.size() * 4);
    for (int trial = 0; trial < 3; ++trial) {
        srand(17 + trial);
        // trial 0: N(0,1)-like; 1: wide dynamic range (exponents over 2^-20 .. 2^20); 2: positive (no cancellation)
        for (auto* v : {&A, &B}) for (auto& x : *v) {
            const float u = (rand() / (float)RAND_MAX) * 2.f - 1.f;
            x = trial == 0 ? u : trial == 1 ? u * std::ldexp(1.f, rand() % 41 - 20) : std::fabs(u) + 0.1f;
        }
        std::vector<double> R(32 * 32); std::vector<float> H(32 * 32);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0; float h = 0.f;
            for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * (double)B[k * 32 + j]; h = std::fmaf(A[i * K + k], B[k * 32 + j], h); }
            R[i * 32 + j] = s; H[i * 32 + j] = h;
        }
        (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        auto report = [&](const char* name, const float* c) {
            double num = 0, den = 0, mx = 0;
            for (int e = 0; e < 1024; ++e) { const double d = c[e] - R[e]; num += d * d; den += R[e] * R[e]; mx = std::fmax(mx, std::fabs(d)); }
            double scale = 0; for (int e = 0; e < 1024; ++e) scale = std::fmax(scale, std::fabs(R[e]));
            printf("  trial %d  %-34s rel L2 error %.3e   max |error| / max |C| %.3e\n", trial, name, std::sqrt(num / den), mx / scale);
        };
        report("host fmaf chain (fp32)", H.data());
        hipLaunchKernelGGL(gemm_f32, dim3(1), dim3(64), 0, 0, dA, dB, dC, K); (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost); report("v_mfma_f32_32x32x2_f32", C.data());
        hipLaunchKernelGGL(gemm_bx<9>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K); (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost); report("bf16 x 9 terms", C.data());
        hipLaunchKernelGGL(gemm_bx<8>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K); (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost); report("bf16 x 8 terms (no lo*lo)", C.data());
        hipLaunchKernelGGL(gemm_bx<6>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K); (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost); report("bf16 x 6 terms (no lo*lo, lo*mid)", C.data());
        hipLaunchKernelGGL(gemm_bx<3>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K); (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost); report("bf16 x 3 terms (hi*hi, hi*mid, mid*hi)", C.data());
    }
    float* d; (void)hipMalloc(&d, 64);
    const int iters = 40000;
    run_rate("bf16 32x32x16, 0 fillers", rate<4, 0>, 1, 4, iters, d);
    run_rate("bf16 32x32x16, 2 fillers", rate<4, 2>, 1, 4, iters, d);
    run_rate("bf16 32x32x16, 4 fillers", rate<4, 4>, 1, 4, iters, d);
    run_rate("bf16 32x32x16, 6 fillers", rate<4, 6>, 1, 4, iters, d);
    run_rate("bf16 32x32x16, 8 fillers", rate<4, 8>, 1, 4, iters, d);
    run_rate("bf16 32x32x16, 0 fillers", rate<4, 0>, 2, 4, iters, d);
    run_rate("bf16 32x32x16, 4 fillers", rate<4, 4>, 2, 4, iters, d);
    run_rate("bf16 32x32x16, 6 fillers", rate<4, 6>, 2, 4, iters, d);
    run_rate("bf16 32x32x16, 8 fillers", rate<4, 8>, 2, 4, iters, d);
    return 0;
}
