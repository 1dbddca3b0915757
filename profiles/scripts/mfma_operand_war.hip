// Is it safe to overwrite a VGPR of an MFMA's A / B operand right after the MFMA has been issued?  (ROCm 7.2's compiler does it: it has no
// write-after-read hazard for SrcA / SrcB in its model.)  One wave: acc = mfma(a, b, acc) [optionally preceded by NDEP dependent MFMAs on the same
// accumulator, so that the probed one waits in the matrix pipe], then -- after PAD s_nop slots -- a v_mov that clobbers register R of A or B.
// The result is compared with the same sequence without the clobber.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mow profiles/scripts/mfma_operand_war.hip && /tmp/mow
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int WHICH, int R, int PAD, int NDEP, int CLOB>
__global__ void kern(const unsigned* in, float* out) {
    const int lane = threadIdx.x;
    u32x4 a, b;
    for (int i = 0; i < 4; ++i) { a[i] = in[lane * 4 + i]; b[i] = in[256 + lane * 4 + i]; }
    u32x4 a0 = a, b0 = b;
    f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    f32x4 acc4 = {0, 0, 0, 0};
    // everything below in ONE asm block with FIXED registers so that the distances and the clobbered register are exact:
    // a = v[20:23], b = v[24:27], the dependent chain's operands a0 = v[28:31], b0 = v[32:35], acc = v[40:55] / v[40:43]
    if (SHAPE == 32) {
        asm volatile(
            ".rept %4\n v_mfma_f32_32x32x16_bf16 v[40:55], v[28:31], v[32:35], v[40:55]\n .endr\n"
            "v_mfma_f32_32x32x16_bf16 v[40:55], v[20:23], v[24:27], v[40:55]\n"
            ".rept %3\n s_nop 0\n .endr\n"
            ".if %5\n v_mov_b32 v[%6], 0x3f803f80\n .endif\n"
            "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
            : "+{v[40:55]}"(acc), "+{v[20:23]}"(a), "+{v[24:27]}"(b) : "n"(PAD), "n"(NDEP), "n"(CLOB), "n"(20 + 4 * WHICH + R), "{v[28:31]}"(a0), "{v[32:35]}"(b0));
    } else {
        asm volatile(
            ".rept %4\n v_mfma_f32_16x16x32_bf16 v[40:43], v[28:31], v[32:35], v[40:43]\n .endr\n"
            "v_mfma_f32_16x16x32_bf16 v[40:43], v[20:23], v[24:27], v[40:43]\n"
            ".rept %3\n s_nop 0\n .endr\n"
            ".if %5\n v_mov_b32 v[%6], 0x3f803f80\n .endif\n"
            "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
            : "+{v[40:43]}"(acc4), "+{v[20:23]}"(a), "+{v[24:27]}"(b) : "n"(PAD), "n"(NDEP), "n"(CLOB), "n"(20 + 4 * WHICH + R), "{v[28:31]}"(a0), "{v[32:35]}"(b0));
    }
    for (int i = 0; i < 16; ++i) out[lane * 16 + i] = SHAPE == 32 ? acc[i] : (i < 4 ? acc4[i] : 0.f);
}
static unsigned* d_in; static float *d_o1, *d_o2; static float h1[1024], h2[1024];
template <int SHAPE, int WHICH, int R, int PAD, int NDEP>
static int probe() {
    hipLaunchKernelGGL((kern<SHAPE, WHICH, R, PAD, NDEP, 0>), dim3(1), dim3(64), 0, 0, d_in, d_o1);
    hipLaunchKernelGGL((kern<SHAPE, WHICH, R, PAD, NDEP, 1>), dim3(1), dim3(64), 0, 0, d_in, d_o2);
    hipMemcpy(h1, d_o1, sizeof(h1), hipMemcpyDeviceToHost); hipMemcpy(h2, d_o2, sizeof(h2), hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += memcmp(&h1[i], &h2[i], 4) != 0;
    return bad;
}
#define ROW(SHAPE, WHICH, NDEP) { printf("%dx%d %s, %d dependent MFMAs ahead: ", SHAPE, SHAPE, WHICH ? "B" : "A", NDEP); \
    printf(" reg0 pad0 %4d  reg1 pad0 %4d  reg2 pad0 %4d  reg3 pad0 %4d | reg3 pad1 %4d pad2 %4d pad4 %4d pad8 %4d pad16 %4d\n", \
           probe<SHAPE, WHICH, 0, 0, NDEP>(), probe<SHAPE, WHICH, 1, 0, NDEP>(), probe<SHAPE, WHICH, 2, 0, NDEP>(), probe<SHAPE, WHICH, 3, 0, NDEP>(), \
           probe<SHAPE, WHICH, 3, 1, NDEP>(), probe<SHAPE, WHICH, 3, 2, NDEP>(), probe<SHAPE, WHICH, 3, 4, NDEP>(), probe<SHAPE, WHICH, 3, 8, NDEP>(), probe<SHAPE, WHICH, 3, 16, NDEP>()); }
int main() {
    unsigned h[512];
    for (int i = 0; i < 512; ++i) { const unsigned x = (unsigned)(i * 2654435761u); h[i] = 0x3f803f80u ^ ((x >> 9) & 0x007f007fu); }     // pairs of bf16 in [1, 2)
    hipMalloc(&d_in, sizeof(h)); hipMalloc(&d_o1, 4096); hipMalloc(&d_o2, 4096);
    hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
    printf("number of result elements (of 1024 / 256) that change when a VGPR of the operand is overwritten `pad` s_nop slots after the MFMA issue\n");
    ROW(32, 0, 0) ROW(32, 1, 0) ROW(32, 0, 1) ROW(32, 1, 1) ROW(32, 0, 3) ROW(32, 1, 3)
    ROW(16, 0, 0) ROW(16, 1, 0) ROW(16, 0, 3) ROW(16, 1, 3)
    return 0;
}
