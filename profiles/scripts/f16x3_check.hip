// fp32 products on the fp16 matrix cores with TWO planes per operand: x = h + l, h = fp16(x) by truncation (v_cvt_pkrtz_f16_f32), l = fp16(x - h);
// the three largest cross products (l*h, h*l, h*h) accumulated in fp32 by v_mfma_f32_32x32x16_f16 -- half the matrix instructions, a third less
// operand bytes and less than half the split work of the exact three-plane bf16 split (bf16x9_check.hip).  Questions answered here:
//   1. does the fp16 MFMA of gfx950 keep SUBNORMAL inputs (the l planes of O(1) values sit at and below 2^-14)?
//   2. error of a 32 x 32 x K product against fp64 next to the six-term bf16 split and the fp32 MFMA, for unit-scale data, data scaled to 1e-2,
//      weights in [0, 1] x normalised features (the inter conv's step 1), and small weights x large sums (its step 2, with and without a 2^6 pre-scale).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/f16x3 profiles/scripts/f16x3_check.hip && /tmp/f16x3
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split3(float a, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    const unsigned ua = __float_as_uint(a);
    const float r = a - __uint_as_float(ua & 0xffff0000u);
    const unsigned ur = __float_as_uint(r);
    const float l = r - __uint_as_float(ur & 0xffff0000u);
    hi = ua >> 16; mid = ur >> 16; lo = __float_as_uint(l) >> 16;
}
__device__ __forceinline__ void split2h(float a, float b, f16x2& h, f16x2& l) {
    h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
    l = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a - (float)h[0], b - (float)h[1]));
}
__global__ void gemm_bx6(const float* A, const float* B, float* C, int K) {
    const int l = threadIdx.x, i = l & 31, kg = l >> 5;
    f32x16 acc; for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 a[3], b[3];
        for (int e = 0; e < 8; ++e) {
            unsigned short h, m, lo;
            split3(A[i * K + k0 + 8 * kg + e], h, m, lo); a[0][e] = h; a[1][e] = m; a[2][e] = lo;
            split3(B[(k0 + 8 * kg + e) * 32 + i], h, m, lo); b[0][e] = h; b[1][e] = m; b[2][e] = lo;
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
    for (int v = 0; v < 16; ++v) C[((v & 3) + 8 * (v >> 2) + 4 * kg) * 32 + i] = acc[v];
}
template <int NTERM>
__global__ void gemm_hx(const float* A, const float* B, float* C, int K) {
    const int l = threadIdx.x, i = l & 31, kg = l >> 5;
    f32x16 acc; for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        f16x8 ah, al, bh, bl;
        for (int e = 0; e < 8; e += 2) {
            f16x2 h, lo;
            split2h(A[i * K + k0 + 8 * kg + e], A[i * K + k0 + 8 * kg + e + 1], h, lo); ah[e] = h[0]; ah[e + 1] = h[1]; al[e] = lo[0]; al[e + 1] = lo[1];
            split2h(B[(k0 + 8 * kg + e) * 32 + i], B[(k0 + 8 * kg + e + 1) * 32 + i], h, lo); bh[e] = h[0]; bh[e + 1] = h[1]; bl[e] = lo[0]; bl[e + 1] = lo[1];
        }
        if (NTERM >= 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    }
    for (int v = 0; v < 16; ++v) C[((v & 3) + 8 * (v >> 2) + 4 * kg) * 32 + i] = acc[v];
}
__global__ void gemm_f32(const float* A, const float* B, float* C, int K) {
    const int l = threadIdx.x, i = l & 31, kg = l >> 5;
    f32x16 acc; for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k0 + kg], B[(k0 + kg) * 32 + i], acc, 0, 0, 0);
    for (int v = 0; v < 16; ++v) C[((v & 3) + 8 * (v >> 2) + 4 * kg) * 32 + i] = acc[v];
}
// subnormal probe: A[i][0] = a (all rows), B[0][j] = b, everything else 0 -> C = a * b if the operands survive
__global__ void denorm_probe(float a, float b, float* out) {
    const int l = threadIdx.x;
    f16x8 x, y; for (int e = 0; e < 8; ++e) { x[e] = 0; y[e] = 0; }
    if (l < 32) { x[0] = (_Float16)a; y[0] = (_Float16)b; }
    f32x16 acc; for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc, 0, 0, 0);
    if (l == 0) { out[0] = acc[0]; out[1] = (float)x[0]; out[2] = (float)y[0]; }
}
int main() {
    float* dout; (void)hipMalloc(&dout, 64);
    float ho[3];
    const float probes[4][2] = {{3.0e-6f, 1.0f}, {6.0e-8f, 1024.0f}, {1.0e-5f, 1.0e-5f}, {5.0e-5f, 0.5f}};
    for (auto& pr : probes) {
        hipLaunchKernelGGL(denorm_probe, dim3(1), dim3(64), 0, 0, pr[0], pr[1], dout); (void)hipMemcpy(ho, dout, 12, hipMemcpyDeviceToHost);
        printf("subnormal probe: fp16(%g) = %.9g  x  fp16(%g) = %.9g  ->  MFMA %.9g   (exact product of the fp16 values %.9g)\n", pr[0], ho[1], pr[1], ho[2], ho[0], (double)ho[1] * ho[2]);
    }
    const int K = 768;
    std::vector<float> A(32 * K), B(K * 32), C(32 * 32);
    float *dA, *dB, *dC; (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, C.size() * 4);
    const char* names[6] = {"unit-scale x unit-scale", "1e-2-scale x 1e-2-scale", "weights in [0,1] (70% zero) x normalised features", "xavier weights (0.05) x sums (~30)",
                            "xavier weights * 2^6 x sums (~30)", "wide exponents 2^-12..2^12"};
    for (int trial = 0; trial < 6; ++trial) {
        srand(17 + trial);
        auto u = [] { return (rand() / (float)RAND_MAX) * 2.f - 1.f; };
        auto g = [&] { float s = 0; for (int i = 0; i < 6; ++i) s += u(); return s * 0.7071f; };
        for (auto& x : A) x = trial == 0 ? u() : trial == 1 ? 1e-2f * u() : trial == 2 ? std::fmax(0.f, u() - 0.4f) / 0.6f : trial == 3 ? 0.05f * g() : trial == 4 ? 64.f * 0.05f * g() : u() * std::ldexp(1.f, rand() % 25 - 12);
        for (auto& x : B) x = trial == 0 ? u() : trial == 1 ? 1e-2f * u() : trial == 2 ? g() * 1.5f : trial <= 4 ? 30.f * g() : u() * std::ldexp(1.f, rand() % 25 - 12);
        std::vector<double> R(32 * 32);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[k * 32 + j]; R[i * 32 + j] = s; }
        (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        printf("%s\n", names[trial]);
        auto report = [&](const char* name, const float* c) {
            double num = 0, den = 0, mx = 0, scale = 0;
            for (int e = 0; e < 1024; ++e) { const double d = c[e] - R[e]; num += d * d; den += R[e] * R[e]; mx = std::fmax(mx, std::fabs(d)); scale = std::fmax(scale, std::fabs(R[e])); }
            printf("    %-40s rel L2 error %.3e   max |error| / max |C| %.3e\n", name, std::sqrt(num / den), mx / scale);
        };
        hipLaunchKernelGGL(gemm_f32, dim3(1), dim3(64), 0, 0, dA, dB, dC, K); (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost); report("v_mfma_f32_32x32x2_f32", C.data());
        hipLaunchKernelGGL(gemm_bx6, dim3(1), dim3(64), 0, 0, dA, dB, dC, K); (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost); report("bf16, 3 planes, 6 terms (exact split)", C.data());
        hipLaunchKernelGGL(gemm_hx<3>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K); (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost); report("fp16, 2 planes, 3 terms", C.data());
        hipLaunchKernelGGL(gemm_hx<4>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K); (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost); report("fp16, 2 planes, 4 terms", C.data());
    }
    return 0;
}
