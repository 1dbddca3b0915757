// How long after the issue of the LAST MFMA of a dependent chain (same accumulator, back to back) is its result readable by a VALU instruction?
// ROCm 7.2's hazard recogniser inserts a fixed number of wait states after the last MFMA (it models the result as ready passes + 2 states after
// ISSUE).  This probe issues N dependent v_mfma_f32_32x32x16_bf16 back to back, then K s_nop states, then copies the accumulator with v_mov and
// compares with the fully drained result.  Reported: the smallest K for which every element is right.
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mcl profiles/scripts/mfma_chain_latency.hip && /tmp/mcl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int N, int K>
__global__ void kern(const unsigned* in, float* out) {
    const int lane = threadIdx.x;
    u32x4 a, b;
    for (int i = 0; i < 4; ++i) { a[i] = in[lane * 4 + i]; b[i] = in[256 + lane * 4 + i]; }
    f32x16 acc, cp; for (int i = 0; i < 16; ++i) { acc[i] = 0.f; cp[i] = -1.f; }
    if (SHAPE == 32) {
        asm volatile(
            "s_nop 15\n s_nop 15\n"
            ".rept %4\n v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n .endr\n"
            ".rept %5\n s_nop 0\n .endr\n"
            "v_mov_b32 %1, %0\n"                    // register 0 of the tuple pair: the assembler expands %1 / %0 to the tuples; copy all 16 below
            : "+v"(acc), "+v"(cp) : "v"(a), "v"(b), "n"(N), "n"(K));
    }
    out[lane * 16] = 0.f;
}
// explicit-register version (the tuple copy needs per-register moves)
template <int SHAPE, int N, int K>
__global__ void kern2(const unsigned* in, float* out) {
    const int lane = threadIdx.x;
    u32x4 a, b;
    for (int i = 0; i < 4; ++i) { a[i] = in[lane * 4 + i]; b[i] = in[256 + lane * 4 + i]; }
    f32x16 acc, cp; for (int i = 0; i < 16; ++i) { acc[i] = 0.f; cp[i] = -1.f; }
    if (SHAPE == 32)
        asm volatile(
            "s_nop 15\n s_nop 15\n"
            ".rept %4\n v_mfma_f32_32x32x16_bf16 v[40:55], v[20:23], v[24:27], v[40:55]\n .endr\n"
            ".rept %5\n s_nop 0\n .endr\n"
            "v_mov_b32 v60, v40\n v_mov_b32 v61, v41\n v_mov_b32 v62, v42\n v_mov_b32 v63, v43\n v_mov_b32 v64, v44\n v_mov_b32 v65, v45\n v_mov_b32 v66, v46\n v_mov_b32 v67, v47\n"
            "v_mov_b32 v68, v48\n v_mov_b32 v69, v49\n v_mov_b32 v70, v50\n v_mov_b32 v71, v51\n v_mov_b32 v72, v52\n v_mov_b32 v73, v53\n v_mov_b32 v74, v54\n v_mov_b32 v75, v55\n"
            "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
            : "+{v[40:55]}"(acc), "+{v[60:75]}"(cp) : "{v[20:23]}"(a), "{v[24:27]}"(b), "n"(N), "n"(K));
    else {
        f32x4 acc4 = {0, 0, 0, 0}, cp4 = {-1, -1, -1, -1};
        asm volatile(
            "s_nop 15\n s_nop 15\n"
            ".rept %4\n v_mfma_f32_16x16x32_bf16 v[40:43], v[20:23], v[24:27], v[40:43]\n .endr\n"
            ".rept %5\n s_nop 0\n .endr\n"
            "v_mov_b32 v60, v40\n v_mov_b32 v61, v41\n v_mov_b32 v62, v42\n v_mov_b32 v63, v43\n"
            "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
            : "+{v[40:43]}"(acc4), "+{v[60:63]}"(cp4) : "{v[20:23]}"(a), "{v[24:27]}"(b), "n"(N), "n"(K));
        for (int i = 0; i < 4; ++i) { acc[i] = acc4[i]; cp[i] = cp4[i]; }
        for (int i = 4; i < 16; ++i) { acc[i] = 0; cp[i] = 0; }
    }
    for (int i = 0; i < 16; ++i) { out[lane * 32 + i] = acc[i]; out[lane * 32 + 16 + i] = cp[i]; }
}
static unsigned* d_in; static float* d_o; static float h[2048];
template <int SHAPE, int N, int K> static int probe() {
    hipLaunchKernelGGL((kern2<SHAPE, N, K>), dim3(1), dim3(64), 0, 0, d_in, d_o);
    hipMemcpy(h, d_o, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) bad += memcmp(&h[l * 32 + i], &h[l * 32 + 16 + i], 4) != 0;
    return bad;
}
#define ROW(SHAPE, N) printf("%dx%d, chain of %2d: wrong elements after K wait states: K=2 %4d  K=6 %4d  K=10 %4d  K=14 %4d  K=18 %4d  K=24 %4d  K=32 %4d  K=48 %4d  K=64 %4d  K=96 %4d  K=128 %4d\n", SHAPE, SHAPE, N, \
    probe<SHAPE, N, 2>(), probe<SHAPE, N, 6>(), probe<SHAPE, N, 10>(), probe<SHAPE, N, 14>(), probe<SHAPE, N, 18>(), probe<SHAPE, N, 24>(), probe<SHAPE, N, 32>(), probe<SHAPE, N, 48>(), probe<SHAPE, N, 64>(), probe<SHAPE, N, 96>(), probe<SHAPE, N, 128>());
int main() {
    unsigned hh[512];
    for (int i = 0; i < 512; ++i) { const unsigned x = (unsigned)(i * 2654435761u); hh[i] = 0x3f803f80u ^ ((x >> 9) & 0x007f007fu); }
    hipMalloc(&d_in, sizeof(hh)); hipMalloc(&d_o, sizeof(h));
    hipMemcpy(d_in, hh, sizeof(hh), hipMemcpyHostToDevice);
    ROW(32, 1) ROW(32, 2) ROW(32, 4) ROW(32, 6) ROW(32, 12)
    ROW(16, 1) ROW(16, 4) ROW(16, 12)
    return 0;
}
