// Price of one all-to-all exchange round between the G workgroups of a split scan (the seam a split furthest-point-sampling kernel would add to
// every round): each workgroup publishes 5 data-tagged 8-byte granules (64-bit key + 3 coordinates), one wave sweeps the 5 G granules of the round
// (double-buffered by round parity), LDS broadcast + one workgroup barrier.  S scans run concurrently (S x G workgroups).
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/xwg profiles/scripts/xwg_exchange.hip && /tmp/xwg
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(1))) unsigned long long gu64;
template <int G>
__global__ void __launch_bounds__(1024) xchg(int rounds, unsigned long long* gran, int* fail, float* out) {
    __shared__ unsigned res[5];
    const int scan = blockIdx.x / G, g = blockIdx.x % G, tid = threadIdx.x, lane = tid & 63;
    gu64* base = (gu64*)(gran + (size_t)scan * 2 * 5 * G * 8);          // granules 64 bytes apart
    float acc = 0.f;
    for (int r = 1; r <= rounds; ++r) {
        gu64* buf = base + (r & 1) * 5 * G * 8;
        if (tid < 64) {
            if (lane < 5) __hip_atomic_store(buf + (g * 5 + lane) * 8, ((unsigned long long)r << 32) | (unsigned)(g * 1000 + r + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned v = 0; bool ok; unsigned spins = 0;
            do {
                ok = true;
                if (lane < 5 * G) { const unsigned long long x = __hip_atomic_load(buf + lane * 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v = (unsigned)x; ok = (x >> 32) == (unsigned)r; }
                if (++spins > (1u << 22)) { if (lane == 0) *fail = 1; break; }
            } while (!__all(ok));
            // "fixed-order max": pick workgroup (r mod G)'s record
            const int pick = r % G;
            if (lane >= pick * 5 && lane < pick * 5 + 5) res[lane - pick * 5] = v;
        }
        __syncthreads();
        acc += (float)res[tid % 5];
        __syncthreads();
    }
    out[blockIdx.x * blockDim.x + tid] = acc;
}
template <int G> static void run(int S, int threads, unsigned long long* gran, int* fail, float* out) {
    const int rounds = 10000;
    hipMemset(gran, 0, (size_t)S * 2 * 5 * G * 64); hipMemset(fail, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(xchg<G>, dim3(S * G), dim3(threads), 0, 0, rounds, gran, fail, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int f; hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
    printf("G %d, scans %2d, %4d threads: %.3f us per round%s\n", G, S, threads, ms * 1e3 / rounds, f ? "  (SPIN TIMEOUT)" : "");
}
int main() {
    unsigned long long* gran; int* fail; float* out;
    hipMalloc(&gran, 64 * 2 * 5 * 8 * 64); hipMalloc(&fail, 4); hipMalloc(&out, 64 * 8 * 1024 * 4);
    for (int S : {1, 8, 32}) for (int th : {256, 1024}) {
        run<2>(S, th, gran, fail, out); run<4>(S, th, gran, fail, out); run<8>(S, th, gran, fail, out);
    }
    run<2>(64, 1024, gran, fail, out); run<4>(64, 1024, gran, fail, out);
    return 0;
}
