// Shared helpers for the gfx950 kernels of etch_amd.  MI355X only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ETCH_WAVE 64

#define ETCH_RETURN_IF_LAUNCH_FAILED()            \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

// status codes returned by every C-ABI entry point (0 = ok, >0 = hipError_t, <0 = argument error)
// compute units of the current device, queried once (256 on MI355X; partitioned or smaller parts report fewer): persistent kernels size
// their grids from it instead of assuming 256
static inline int etch_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        cus = n;
    }
    return cus;
}

#define ETCH_OK 0
#define ETCH_EINVAL (-1)
#define ETCH_EUNSUPPORTED (-2)

// "bit-defined fp32" squared distance: ((dx*dx)+(dy*dy))+(dz*dz), every op rounded, NO fma.
// Same contract as oracle/discrete_ops.c:sqdist.  The TU is also built with -ffp-contract=off.
__device__ __forceinline__ float etch_sqdist(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    float s = xx + yy;
    return s + zz;
}

__device__ __forceinline__ unsigned long long etch_wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned long long o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// 64-bit wave max on the VALU (DPP row shifts + row broadcasts, no LDS traffic); result valid in every lane
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ unsigned long long etch_dpp_max_u64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v & 0xFFFFFFFFull), CTRL, ROW_MASK, 0xf, BOUND);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xf, BOUND);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    return o > v ? o : v;
}
__device__ __forceinline__ unsigned long long etch_wave_max_u64_dpp(unsigned long long v) {
    v = etch_dpp_max_u64<0x111, 0xf, true>(v);     // row_shr:1  (out-of-row lanes read 0, the identity of max)
    v = etch_dpp_max_u64<0x112, 0xf, true>(v);     // row_shr:2
    v = etch_dpp_max_u64<0x114, 0xf, true>(v);     // row_shr:4
    v = etch_dpp_max_u64<0x118, 0xf, true>(v);     // row_shr:8  -> lane 15 of each row holds the row max
    v = etch_dpp_max_u64<0x142, 0xa, false>(v);    // row_bcast:15 into rows 1 and 3
    v = etch_dpp_max_u64<0x143, 0xc, false>(v);    // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave max
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xFFFFFFFFull), 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

// "Last workgroup done" (round 6: the two-level fixed-order reductions of the training kernels in ONE launch).  Every workgroup of a group writes its
// partial result, then arrives at the group's counter; the workgroup that arrives last sums the partials IN INDEX ORDER (so the result does not depend
// on which workgroup that was) and leaves the counter at zero for the next launch.  The fences are the release / acquire pair of the hand-over (on
// gfx950 they write back / invalidate the XCD's L2: the partials of workgroups on other XCDs are read from memory).  `counter` must be zero when the
// launch starts; launches that share counters must be ordered on one stream.
__device__ __forceinline__ bool etch_last_block(unsigned* counter, unsigned total) {
    __shared__ unsigned s_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const unsigned t = atomicAdd(counter, 1u);
        s_last = t == total - 1u ? 1u : 0u;
        if (t == total - 1u) atomicExch(counter, 0u);
    }
    __syncthreads();
    const bool last = s_last != 0u;
    if (last) __threadfence();
    return last;
}
#define ETCH_REDUCE_COUNTERS 64          /* counters a fused reduction may touch (one per column group / output tile) */

__device__ __forceinline__ float etch_wave_sum_f32(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float etch_wave_max_f32(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// ---- work counters of the persistent kernels (mhsa layers, intra conv, fused dense head).  Under the multi-stream pipeline other streams' kernels share
// the compute units unevenly; a static round-robin over persistent workgroups then makes a launch wait for the workgroups that were slowed down.  The
// kernels take their first items statically and every further one from a device counter.  A counter = one slot (64 words: one per XCD where a kernel keeps
// per-XCD lists, one per column block of the fused dense head), re-zeroed on the launch's stream in front of the kernel by a one-wave kernel -- not by
// hipMemsetAsync: a memset goes through the runtime's blit / copy path, which under the multi-stream pipeline occasionally stalled a step.
//   * one pool PER DEVICE (keyed by hipGetDevice at the launch: a process that drives several GPUs never hands a kernel another device's memory),
//     created under a mutex, slot numbers from atomics (ctypes releases the GIL: two host threads may launch at once);
//   * eager launches rotate through a ring of RING slots: two launches share a slot only if RING counter-using launches of this translation unit are
//     enqueued while the first is still running (a pipeline step holds ~90 in flight);
//   * launches recorded into a HIP graph take their slot from a second region that is never reused (a replayed graph re-zeroes and uses the slot it
//     was captured with, whenever it runs -- it must not be one that eager launches rotate through); when that region is used up, or the pool
//     cannot be created while a capture is in progress, the launch gets no counter = the static round-robin (same results, no dynamic distribution).
// No counter at all with ETCH_DYNAMIC_WORK=0 / ETCH_MHSA_DYNAMIC=0.  One pool set per translation unit.
#include <atomic>
#include <cstdlib>
#include <mutex>
static __global__ void etch_zero_counters_kernel(unsigned* slot) {
    slot[threadIdx.x] = 0u;
}
static inline unsigned* etch_work_counter_slot(hipStream_t st) {
    constexpr int RING = 2048, GRAPH = 2048, MAXDEV = 32;
    struct Pool { unsigned* mem = nullptr; std::atomic<int> state{0}; std::atomic<unsigned> next{0}; std::atomic<unsigned> next_graph{0}; };      // state: 0 untried, 1 ready, -1 unavailable
    static Pool pools[MAXDEV];
    static std::mutex mu;
    static std::atomic<int> disabled{0};                 // 0 unknown, 1 off (environment), 2 on
    if (disabled.load(std::memory_order_acquire) == 0) {
        const char* e1 = getenv("ETCH_DYNAMIC_WORK");
        const char* e2 = getenv("ETCH_MHSA_DYNAMIC");
        disabled.store(((e1 && e1[0] == '0') || (e2 && e2[0] == '0')) ? 1 : 2, std::memory_order_release);
    }
    if (disabled.load(std::memory_order_acquire) == 1) return nullptr;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
    Pool& P = pools[dev];
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
    if (P.state.load(std::memory_order_acquire) == 0) {
        if (capturing) return nullptr;                   // no allocation inside a capture: this launch runs static, a later one creates the pool
        std::lock_guard<std::mutex> lock(mu);
        if (P.state.load(std::memory_order_relaxed) == 0) {
            unsigned* m = nullptr;
            const bool ok = hipMalloc((void**)&m, (size_t)(RING + GRAPH) * 64 * sizeof(unsigned)) == hipSuccess;
            P.mem = ok ? m : nullptr;
            P.state.store(ok ? 1 : -1, std::memory_order_release);
        }
    }
    if (P.state.load(std::memory_order_acquire) != 1) return nullptr;
    unsigned idx;
    if (capturing) {
        idx = P.next_graph.fetch_add(1u, std::memory_order_relaxed);
        if (idx >= (unsigned)GRAPH) return nullptr;
        idx += RING;
    } else {
        idx = P.next.fetch_add(1u, std::memory_order_relaxed) % (unsigned)RING;
    }
    unsigned* slot = P.mem + (size_t)idx * 64;
    hipLaunchKernelGGL(etch_zero_counters_kernel, dim3(1), dim3(64), 0, st, slot);
    if (hipGetLastError() != hipSuccess) return nullptr;
    return slot;
}
