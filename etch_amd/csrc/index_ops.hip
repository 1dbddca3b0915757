// Index kernels of the ETCH hot path for gfx950.
// Replaces, behind the same semantics:
//   ball_query_cuda_kernel               /root/reference/external/vgtk/vgtk/cuda/grouping_cuda_kernel.cu:68-113
//   furthest_point_sampling_cuda_kernel  .../grouping_cuda_kernel.cu:352-466
//   gather_points_forward_kernel         .../gathering_cuda_kernel.cu:43-68
//   knnquery_cuda_kernel                 /root/reference/external/pointops/src/knnquery/knnquery_cuda_kernel.cu:65-108
//   furthestsampling_cuda_kernel         /root/reference/external/pointops/src/sampling/sampling_cuda_kernel.cu:15-129
// Results are bit-identical to oracle/discrete_ops.c (integer outputs equal, distances equal).
// Built with -ffp-contract=off: distance arithmetic is "bit-defined fp32" (common.h).
#include "common.h"
#include <cmath>
#include <cstdio>

// ------------------------------------------------------------------------------------ ball query
// One WAVE per query: the 64 lanes test 64 consecutive support points per step (coalesced SoA
// loads), a ballot gives the in-ball set in index order, so "first nsample in index order" is
// exact and the early exit is per query.  The row is staged in LDS so the reference's padding
// rule (cyclic prefix when cnt < nsample-1, zero slot when cnt == nsample-1) is applied once and
// the row is written with one coalesced store.
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) ball_query_kernel(int n, int m, float radius2, int nsample,
                                                               const float* __restrict__ new_xyz,
                                                               const float* __restrict__ xyz, int* __restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) int smem_i[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const float* X = xyz + (size_t)b * 3 * n;
    const float* Q = new_xyz + (size_t)b * 3 * m;
    int* row = smem_i + wave * nsample;
    for (int j = blockIdx.x * WAVES + wave; j < m; j += gridDim.x * WAVES) {
        const float qx = Q[j], qy = Q[m + j], qz = Q[2 * m + j];
        int cnt = 0;
        for (int k0 = 0; k0 < n && cnt < nsample; k0 += 64) {
            const int k = k0 + lane;
            bool in = false;
            if (k < n) {
                float d2 = etch_sqdist(qx, qy, qz, X[k], X[n + k], X[2 * n + k]);
                in = d2 < radius2;
            }
            const unsigned long long mask = __ballot(in);
            const int before = __popcll(mask & ((1ull << lane) - 1ull));
            if (in && cnt + before < nsample) row[cnt + before] = k;
            cnt += __popcll(mask);
        }
        if (cnt > nsample) cnt = nsample;
        __builtin_amdgcn_wave_barrier();
        int* out = idx + ((size_t)b * m + j) * nsample;
        for (int t = lane; t < nsample; t += 64) {
            int v;
            if (t < cnt) v = row[t];
            else if (cnt < nsample - 1 && cnt > 0) v = row[t % cnt];
            else v = 0;
            out[t] = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------ FPS
// One workgroup per scan / segment, distances and validity live in registers (PPT points per
// thread), one 64-bit max-reduction per round:
//   key = (bits(d2) << 32) | (0xFFFFFFFF - tie),  tie = (bitrev(k_local mod bs) << 16) | (k_local / bs)
// which reproduces the reference's per-thread strict '>' scan + LDS tree reduce tie-breaking
// exactly (proved against the literal emulation in tests/test_oracle_ops.py).
// Addressing: coordinate c of local point k is src[c * cs + k * ps].
// CARRY: the candidate's coordinates travel with its key through the reduction (scans too large for an LDS copy of the coordinates:
// saves the dependent global read of the winner, 16 -> 2.7 us per round at 20 000 points); otherwise the winner is read from the LDS copy.
// FAST (REGS kernels whose tie-key block size equals THREADS, i.e. every scan of >= 1024 points): a thread's slots share the bit-reversed part of
// the tie key and differ by the slot number only, so the thread's winner is tracked as (distance, slot) with a strict '>' over ascending slots
// -- the same order as the 64-bit keys -- and the key is composed once per round; invalid slots carry distance -2 (< the initial -1) instead of an
// exec-mask branch per slot.  10 instead of 13 VALU instructions + a branch per slot: 2.29 -> 2.18 us per round at 20 000 points; slower where
// few of a thread's slots are occupied (5 000 points: 1.11 -> 1.24), so only the large-scan (CARRY) variants use it (profiles/r04_fps_split.txt).
// One sampling problem (either flavour).  A launch may carry TWO of them over the same point count (round 6, etch_fps_pair): the workgroups
// [0, nseg0) work on the first, the rest on the second -- the encoder's FPS and the first FPS level of the Point-Transformer nets depend on the same
// coordinates only, run one workgroup per scan each, and used to queue behind each other on the index stream (34 of the 58 ms of a step of the dense
// 8-scan shard of configs[4]); a second stream for one of them loses more to hardware-queue sharing than it gains (profiles/r06_config4_fps_pair.txt).
struct FpsProb {
    const float* xyz; long cs, ps, batch_stride; int n_fixed, m_fixed; const int* offset; const int* new_offset; int skip_origin; int* idx;
};
template <int THREADS, int PPT, bool REGS, bool CARRY, bool FAST = false>   // lds_xyz = capacity (points) of the dynamic-LDS coordinate copy, 0 = none
__global__ void __launch_bounds__(THREADS) fps_kernel(int lds_xyz, FpsProb A0, FpsProb A1, int nseg0, int bs, int bs_bits) {
    // per-wave partial maxima, double-buffered
    struct Cand { unsigned long long key; float x, y, z, pad; };
    __shared__ Cand red[2 * (THREADS / 64)];
    extern __shared__ __attribute__((aligned(16))) float fps_xyz[];     // [3][lds_xyz]: non-REGS variants read the winner from here
    const int CAP = lds_xyz;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool second = (int)blockIdx.x >= nseg0;                       // (wave-uniform: the selects below are scalar)
    const float* __restrict__ xyz = second ? A1.xyz : A0.xyz;
    const long cs = second ? A1.cs : A0.cs, ps = second ? A1.ps : A0.ps, batch_stride = second ? A1.batch_stride : A0.batch_stride;
    const int n_fixed = second ? A1.n_fixed : A0.n_fixed, m_fixed = second ? A1.m_fixed : A0.m_fixed;
    const int* __restrict__ offset = second ? A1.offset : A0.offset;
    const int* __restrict__ new_offset = second ? A1.new_offset : A0.new_offset;
    const int skip_origin = second ? A1.skip_origin : A0.skip_origin;
    int* __restrict__ idx = second ? A1.idx : A0.idx;
    const int seg = (int)blockIdx.x - (second ? nseg0 : 0);
    int start_n, n, start_m, m;
    const float* src;
    if (offset == nullptr) {                 // vgtk flavour: dense (b,3,n) batches, local indices
        start_n = 0; n = n_fixed; start_m = seg * m_fixed; m = m_fixed;
        src = xyz + (size_t)seg * batch_stride;
    } else {                                 // pointops flavour: packed (n,3) with cumulative offsets, global indices
        start_n = seg == 0 ? 0 : offset[seg - 1];
        n = offset[seg] - start_n;
        start_m = seg == 0 ? 0 : new_offset[seg - 1];
        m = new_offset[seg] - start_m;
        src = xyz + (size_t)start_n * ps;
    }
    // REGS: coordinates and tie keys cached in registers; otherwise re-read (coalesced, L2-resident) every round.
    float px[REGS ? PPT : 1], py[REGS ? PPT : 1], pz[REGS ? PPT : 1];
    unsigned tie[REGS ? PPT : 1];
    float temp[PPT];
    unsigned valid = 0u;                     // bit i: point tid + i*THREADS is a candidate
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = tid + i * THREADS;
        temp[i] = 1e10f;
        if (REGS) {
            px[i] = py[i] = pz[i] = 0.f;
            const unsigned kl = (unsigned)k & (unsigned)(bs - 1);
            const unsigned rev = bs_bits ? (__brev(kl) >> (32 - bs_bits)) : 0u;
            tie[i] = 0xFFFFFFFFu - ((rev << 16) | ((unsigned)k / (unsigned)bs));
        }
        if (k < n) {
            const float x = src[k * ps], y = src[cs + k * ps], z = src[2 * cs + k * ps];
            if (REGS) { px[i] = x; py[i] = y; pz[i] = z; }
            if (!CARRY && lds_xyz) { fps_xyz[k] = x; fps_xyz[CAP + k] = y; fps_xyz[2 * CAP + k] = z; }
            bool ok = true;
            if (skip_origin) {
#pragma clang fp contract(off)
                float mag = ((x * x) + (y * y)) + (z * z);
                ok = !((double)mag <= 1e-3);
            }
            if (ok) valid |= 1u << i;
        }
    }
    if (FAST) {
#pragma unroll
        for (int i = 0; i < PPT; ++i)
            if (!(valid & (1u << i))) temp[i] = -2.0f;    // never a candidate: below the initial best of -1
    }
    if (tid == 0 && m > 0) idx[start_m] = start_n;
    if (!CARRY && lds_xyz) __syncthreads();
    int old = 0;  // local index of the last selected point
    float x1 = 0.f, y1 = 0.f, z1 = 0.f;
    if (n > 0) { x1 = src[0]; y1 = src[cs]; z1 = src[2 * cs]; }
    for (int j = 1; j < m; ++j) {
        if (!CARRY) {
            if (lds_xyz) { x1 = fps_xyz[old]; y1 = fps_xyz[CAP + old]; z1 = fps_xyz[2 * CAP + old]; }
            else { x1 = src[old * ps]; y1 = src[cs + old * ps]; z1 = src[2 * cs + old * ps]; }
        }
        unsigned long long best = 0ull;
        if constexpr (FAST) {
            float bestd = -1.0f;
            int besti = -1;
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const float d = etch_sqdist(px[i], py[i], pz[i], x1, y1, z1);
                float d2;                                  // v_min_f32 as it stands (the builtin adds a canonicalising v_max per operand); no NaNs here
                asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d), "v"(temp[i]));
                temp[i] = d2;
                const bool gt = d2 > bestd;
                bestd = gt ? d2 : bestd;
                besti = gt ? i : besti;
            }
            // tie[i] = tie[0] - i: the slots of a thread are THREADS = bs points apart
            if (besti >= 0) best = ((unsigned long long)__float_as_uint(bestd) << 32) | (tie[0] - (unsigned)besti);
        } else
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            if (valid & (1u << i)) {
                const int k = tid + i * THREADS;
                float x, y, z;
                if (REGS) { x = px[i]; y = py[i]; z = pz[i]; }
                else { x = src[k * ps]; y = src[cs + k * ps]; z = src[2 * cs + k * ps]; }
                float d = etch_sqdist(x, y, z, x1, y1, z1);
                float d2 = d < temp[i] ? d : temp[i];
                temp[i] = d2;
                unsigned t;
                if (REGS) t = tie[i];
                else {
                    const unsigned kl = (unsigned)k & (unsigned)(bs - 1);
                    const unsigned rev = bs_bits ? (__brev(kl) >> (32 - bs_bits)) : 0u;
                    t = 0xFFFFFFFFu - ((rev << 16) | ((unsigned)k / (unsigned)bs));
                }
                const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | t;
                const bool gt = key > best;
                best = gt ? key : best;
            }
        }
        const unsigned long long wbest = etch_wave_max_u64_dpp(best);
        // one barrier per round: partial maxima are double-buffered and every wave reduces them redundantly
        Cand* rb = red + (j & 1) * (THREADS / 64);
        if (CARRY) {
            // keys are unique per point, so exactly one lane holds the wave's best (or none when the wave has no candidate).  Its coordinates are
            // fetched from that lane's registers AFTER the scan: the key names the point (k = slot * THREADS + thread), the slot is wave-uniform,
            // so one scalar-selected v_readlane per coordinate replaces three selects per point and round (60 of ~300 VALU ops per lane at 20 000 points)
            const unsigned long long hit = __ballot(best == wbest && wbest != 0ull);
            const int srcl = hit ? (int)__builtin_ctzll(hit) : 0;
            float wx = 0.f, wy = 0.f, wz = 0.f;
            if (REGS && hit) {                                 // wave-uniform
                const unsigned t = 0xFFFFFFFFu - (unsigned)(wbest & 0xFFFFFFFFull);
                const unsigned kl = bs_bits ? (__brev(t >> 16) >> (32 - bs_bits)) : 0u;
                const int slot = (int)((t & 0xFFFFu) * (unsigned)bs + kl) / THREADS;
#pragma unroll
                for (int i = 0; i < PPT; ++i)
                    if (slot == i) {                           // wave-uniform: a scalar branch
                        wx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, px[i]), srcl));
                        wy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, py[i]), srcl));
                        wz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pz[i]), srcl));
                    }
            }
            if (lane == 0) { rb[wave].key = wbest; rb[wave].x = wx; rb[wave].y = wy; rb[wave].z = wz; }
        } else if (lane == 0) rb[wave].key = wbest;
        __syncthreads();
        const unsigned long long mine = lane < THREADS / 64 ? rb[lane].key : 0ull;
        const unsigned long long v = etch_wave_max_u64_dpp(mine);
        int w = 0;  // no candidate at all -> local index 0 (reference: besti default)
        if (v != 0ull) {
            const unsigned t = 0xFFFFFFFFu - (unsigned)(v & 0xFFFFFFFFull);
            const unsigned rev = t >> 16, hi = t & 0xFFFFu;
            const unsigned kl = bs_bits ? (__brev(rev) >> (32 - bs_bits)) : 0u;
            w = (int)(hi * (unsigned)bs + kl);
            if (CARRY) {
                const int wl = (int)__builtin_ctzll(__ballot(mine == v));
                x1 = rb[wl].x; y1 = rb[wl].y; z1 = rb[wl].z;
            }
        } else if (CARRY && n > 0) { x1 = src[0]; y1 = src[cs]; z1 = src[2 * cs]; }
        if (tid == 0) idx[start_m + j] = start_n + w;
        old = w;
    }
}

// ------------------------------------------------------------------------------------ FPS, one scan split over G workgroups
// Large scans (>= ~10 000 points) spend their round in the per-point distance update of ONE compute unit (2.7 us per round at 20 000 points:
// 20 points per thread).  Here G workgroups (on G compute units) own interleaved 1024-point chunks of the scan in registers; every round each
// publishes its best candidate -- the 64-bit key and the winner's coordinates as five data-tagged 8-byte granules {tag = round, value}, one
// 64-byte record per workgroup -- and ONE wave sweeps the G records of the round (double-buffered by round parity: a workgroup can only be
// one round ahead of the slowest, which has then finished reading the other parity).  Keys are unique per point, so the maximum over the G
// candidates is the maximum over all points whatever the order: the picks are bit-identical to fps_kernel's (same key, same tie-breaking).
// Price of the seam, measured (profiles/scripts/xwg_exchange.hip): 0.75 us per round for one scan, 1.2 us with 32 - 64 scans in flight.
// Every spin is bounded: on a timeout (a workgroup of the scan not resident, e.g. more workgroups than the device holds) the fail word of the
// workspace is set, the scan's remaining indices are poisoned with INT_MIN and every workgroup of the scan gives up.
typedef __attribute__((address_space(1))) unsigned long long fps_gu64;
template <int PPT>
__global__ void __launch_bounds__(1024) fps_split_kernel(int G, const float* __restrict__ xyz, long cs, long ps, long batch_stride, int n_fixed,
                                                         int m_fixed, const int* __restrict__ offset, const int* __restrict__ new_offset, int bs,
                                                         int bs_bits, int skip_origin, unsigned spin_limit, int drop_group,
                                                         unsigned long long* ws, int* __restrict__ idx) {
    constexpr int THREADS = 1024, NW = THREADS / 64;
    struct Cand { unsigned long long key; float x, y, z, pad; };
    __shared__ Cand red[NW];
    __shared__ float res[4];
    __shared__ int res_w[2];                              // winner's local index, failure flag
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seg = blockIdx.x / G, g = blockIdx.x - seg * G;
    if (g == drop_group) return;                          // test hook: a workgroup that never arrives
    int start_n, n, start_m, m;
    const float* src;
    if (offset == nullptr) {
        start_n = 0; n = n_fixed; start_m = seg * m_fixed; m = m_fixed;
        src = xyz + (size_t)seg * batch_stride;
    } else {
        start_n = seg == 0 ? 0 : offset[seg - 1];
        n = offset[seg] - start_n;
        start_m = seg == 0 ? 0 : new_offset[seg - 1];
        m = new_offset[seg] - start_m;
        src = xyz + (size_t)start_n * ps;
    }
    fps_gu64* slots = (fps_gu64*)(ws + 8) + (size_t)seg * 2 * G * 8;       // [parity][G] records of 8 granules (5 used)
    float px[PPT], py[PPT], pz[PPT], temp[PPT];
    unsigned tie[PPT];
    unsigned valid = 0u;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = (i * G + g) * THREADS + tid;        // chunk i*G + g of the scan
        temp[i] = 1e10f;
        px[i] = py[i] = pz[i] = 0.f;
        const unsigned kl = (unsigned)k & (unsigned)(bs - 1);
        const unsigned rev = bs_bits ? (__brev(kl) >> (32 - bs_bits)) : 0u;
        tie[i] = 0xFFFFFFFFu - ((rev << 16) | ((unsigned)k / (unsigned)bs));
        if (k < n) {
            const float x = src[k * ps], y = src[cs + k * ps], z = src[2 * cs + k * ps];
            px[i] = x; py[i] = y; pz[i] = z;
            bool ok = true;
            if (skip_origin) {
#pragma clang fp contract(off)
                float mag = ((x * x) + (y * y)) + (z * z);
                ok = !((double)mag <= 1e-3);
            }
            if (ok) valid |= 1u << i;
        }
    }
    if (g == 0 && tid == 0 && m > 0) idx[start_m] = start_n;
    if (tid == 0) res_w[1] = 0;
    float x1 = 0.f, y1 = 0.f, z1 = 0.f;
    if (n > 0) { x1 = src[0]; y1 = src[cs]; z1 = src[2 * cs]; }
    __syncthreads();
    for (int j = 1; j < m; ++j) {
        unsigned long long best = 0ull;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            if (valid & (1u << i)) {
                const float d = etch_sqdist(px[i], py[i], pz[i], x1, y1, z1);
                const float d2 = d < temp[i] ? d : temp[i];
                temp[i] = d2;
                const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | tie[i];
                best = key > best ? key : best;
            }
        }
        const unsigned long long wbest = etch_wave_max_u64_dpp(best);
        {
            const unsigned long long hit = __ballot(best == wbest && wbest != 0ull);
            const int srcl = hit ? (int)__builtin_ctzll(hit) : 0;
            float wx = 0.f, wy = 0.f, wz = 0.f;
            if (hit) {                                     // wave-uniform
                const unsigned t = 0xFFFFFFFFu - (unsigned)(wbest & 0xFFFFFFFFull);
                const unsigned kl = bs_bits ? (__brev(t >> 16) >> (32 - bs_bits)) : 0u;
                const int slot = (int)((t & 0xFFFFu) * (unsigned)bs + kl) / THREADS / G;
#pragma unroll
                for (int i = 0; i < PPT; ++i)
                    if (slot == i) {
                        wx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, px[i]), srcl));
                        wy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, py[i]), srcl));
                        wz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pz[i]), srcl));
                    }
            }
            if (lane == 0) { red[wave].key = wbest; red[wave].x = wx; red[wave].y = wy; red[wave].z = wz; }
        }
        __syncthreads();
        if (wave == 0) {
            // this workgroup's candidate
            const unsigned long long mine = lane < NW ? red[lane].key : 0ull;
            const unsigned long long v = etch_wave_max_u64_dpp(mine);
            const int wl = v != 0ull ? (int)__builtin_ctzll(__ballot(mine == v)) : 0;
            fps_gu64* buf = slots + (size_t)(j & 1) * G * 8;
            if (lane < 5) {
                unsigned val;
                if (lane == 0) val = (unsigned)(v & 0xFFFFFFFFull);
                else if (lane == 1) val = (unsigned)(v >> 32);
                else val = __float_as_uint(lane == 2 ? red[wl].x : (lane == 3 ? red[wl].y : red[wl].z));
                __hip_atomic_store(buf + g * 8 + lane, ((unsigned long long)(unsigned)j << 32) | val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // all G candidates of the round
            unsigned val = 0u;
            bool failed = false;
            const int rec = lane / 5, fld = lane - rec * 5;
            for (unsigned spins = 0;;) {
                bool ok = true;
                if (lane < 5 * G) {
                    const unsigned long long x = __hip_atomic_load(buf + rec * 8 + fld, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    val = (unsigned)x;
                    ok = (unsigned)(x >> 32) == (unsigned)j;
                }
                if (__all(ok)) break;
                if (++spins > spin_limit || __hip_atomic_load((__attribute__((address_space(1))) unsigned*)ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    failed = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (failed) {
                if (lane == 0) {
                    __hip_atomic_store((__attribute__((address_space(1))) unsigned*)ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    res_w[1] = 1;
                }
            } else {
                unsigned long long bk = 0ull;
                int bg = 0;
                for (int gg = 0; gg < G; ++gg) {
                    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)val, gg * 5), hi = (unsigned)__builtin_amdgcn_readlane((int)val, gg * 5 + 1);
                    const unsigned long long key = ((unsigned long long)hi << 32) | lo;
                    if (key > bk) { bk = key; bg = gg; }
                }
                int w = 0;
                float wx, wy, wz;
                if (bk != 0ull) {
                    const unsigned t = 0xFFFFFFFFu - (unsigned)(bk & 0xFFFFFFFFull);
                    const unsigned kl = bs_bits ? (__brev(t >> 16) >> (32 - bs_bits)) : 0u;
                    w = (int)((t & 0xFFFFu) * (unsigned)bs + kl);
                    wx = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)val, bg * 5 + 2));
                    wy = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)val, bg * 5 + 3));
                    wz = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)val, bg * 5 + 4));
                } else {                                   // no candidate at all -> local index 0 (reference: besti default)
                    wx = n > 0 ? src[0] : 0.f; wy = n > 0 ? src[cs] : 0.f; wz = n > 0 ? src[2 * cs] : 0.f;
                }
                if (lane == 0) { res[0] = wx; res[1] = wy; res[2] = wz; res_w[0] = w; }
            }
        }
        __syncthreads();
        if (res_w[1]) {                                    // give up: poison what is left of this scan
            if (g == 0)
                for (int t = j + tid; t < m; t += THREADS) idx[start_m + t] = (int)0x80000000;
            return;
        }
        x1 = res[0]; y1 = res[1]; z1 = res[2];
        if (g == 0 && tid == 0) idx[start_m + j] = start_n + res_w[0];
        // no third barrier: `red` is rewritten by waves that passed the barrier above (wave 0 read it before), `res` by wave 0 after the NEXT
        // round's first barrier (every thread has read it by then)
    }
}

// ------------------------------------------------------------------------------------ gather
// out[b,c,j] = points[b,c,idx[b,j]]   (b,c,n) x (b,m) -> (b,c,m); one thread per (j), loop over c.
__global__ void __launch_bounds__(256) gather_points_kernel(int c, int n, int m, const float* __restrict__ points,
                                                            const int* __restrict__ idx, float* __restrict__ out) {
    const int b = blockIdx.y;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x) {
        const int a = idx[(size_t)b * m + j];
        for (int ci = 0; ci < c; ++ci) out[((size_t)b * c + ci) * m + j] = points[((size_t)b * c + ci) * n + a];
    }
}

// grad_points[b, ci, a] = sum over { j : idx[b, j] == a } of grad_out[b, ci, j], summed in ascending j.
// The reference scatters with atomicAdd (gathering_cuda_kernel.cu:73-98), whose rounding depends on the arrival order when an
// index repeats; this is the gather-side formulation of the same sum: one thread per destination point scans the index list,
// so the result is reproducible (and equals the reference's exactly whenever the indices are unique, e.g. FPS output).
__global__ void __launch_bounds__(256) gather_points_backward_kernel(int c, int n, int m, const float* __restrict__ grad_out,
                                                                     const int* __restrict__ idx, float* __restrict__ grad_points) {
    const int b = blockIdx.y;
    const int* ib = idx + (size_t)b * m;
    for (int a = blockIdx.x * 256 + threadIdx.x; a < n; a += gridDim.x * 256) {
        for (int ci = 0; ci < c; ++ci) {
            const float* g = grad_out + ((size_t)b * c + ci) * m;
            float acc = 0.f;
            for (int j = 0; j < m; ++j)
                if (ib[j] == a) acc += g[j];
            grad_points[((size_t)b * c + ci) * n + a] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------ kNN
// One thread per query, support points of the query's segment streamed through LDS tiles
// (coalesced AoS loads -> SoA LDS, broadcast reads).  The k-slot max-heap of the reference is
// emulated literally (same strict '<' insert, same reheap / heap_sort) in LDS columns
// [slot][thread] so that even the order among exactly tied distances is identical.
#define KNN_THREADS 256
#define KNN_TILE 512
__device__ __forceinline__ void knn_reheap(float* hd, int* hi, int k) {
    int root = 0, child = 1;
    while (child < k) {
        if (child + 1 < k && hd[(child + 1) * KNN_THREADS] > hd[child * KNN_THREADS]) child++;
        if (hd[root * KNN_THREADS] > hd[child * KNN_THREADS]) return;
        float tf = hd[root * KNN_THREADS]; hd[root * KNN_THREADS] = hd[child * KNN_THREADS]; hd[child * KNN_THREADS] = tf;
        int ti = hi[root * KNN_THREADS]; hi[root * KNN_THREADS] = hi[child * KNN_THREADS]; hi[child * KNN_THREADS] = ti;
        root = child; child = root * 2 + 1;
    }
}

__global__ void __launch_bounds__(KNN_THREADS) knn_kernel(int nsample, const float* __restrict__ xyz,
                                                          const float* __restrict__ new_xyz, const int* __restrict__ offset,
                                                          const int* __restrict__ new_offset, int* __restrict__ idx,
                                                          float* __restrict__ dist2, int write_sqrt) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* tx = smem_f;                      // [KNN_TILE]
    float* ty = tx + KNN_TILE;
    float* tz = ty + KNN_TILE;
    float* hd = tz + KNN_TILE + threadIdx.x; // heap distances, column of this thread: hd[slot * KNN_THREADS]
    int* hi = (int*)(tz + KNN_TILE + nsample * KNN_THREADS) + threadIdx.x;
    const int seg = blockIdx.y;
    const int start = seg == 0 ? 0 : offset[seg - 1], end = offset[seg];
    const int qs = seg == 0 ? 0 : new_offset[seg - 1], qe = new_offset[seg];
    const int q = qs + blockIdx.x * KNN_THREADS + threadIdx.x;
    if (qs + blockIdx.x * KNN_THREADS >= qe) return;  // whole block idle (uniform)
    const bool active = q < qe;
    float qx = 0, qy = 0, qz = 0;
    if (active) { qx = new_xyz[q * 3]; qy = new_xyz[q * 3 + 1]; qz = new_xyz[q * 3 + 2]; }
    for (int i = 0; i < nsample; ++i) { hd[i * KNN_THREADS] = 1e10f; hi[i * KNN_THREADS] = start; }
    float top = 1e10f;
    for (int t0 = start; t0 < end; t0 += KNN_TILE) {
        const int cnt = min(KNN_TILE, end - t0);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt * 3; e += KNN_THREADS) {
            const float v = xyz[(size_t)t0 * 3 + e];
            const int pnt = e / 3, c = e - pnt * 3;
            (c == 0 ? tx : (c == 1 ? ty : tz))[pnt] = v;
        }
        __syncthreads();
        if (active) {
            for (int i = 0; i < cnt; ++i) {
                const float d2 = etch_sqdist(qx, qy, qz, tx[i], ty[i], tz[i]);
                if (d2 < top) {
                    hd[0] = d2; hi[0] = t0 + i;
                    knn_reheap(hd, hi, nsample);
                    top = hd[0];
                }
            }
        }
    }
    if (active) {
        for (int i = nsample - 1; i > 0; i--) {  // heap_sort
            float tf = hd[0]; hd[0] = hd[i * KNN_THREADS]; hd[i * KNN_THREADS] = tf;
            int ti = hi[0]; hi[0] = hi[i * KNN_THREADS]; hi[i * KNN_THREADS] = ti;
            knn_reheap(hd, hi, i);
        }
        for (int i = 0; i < nsample; ++i) {
            idx[(size_t)q * nsample + i] = hi[i * KNN_THREADS];
            const float d = hd[i * KNN_THREADS];
            dist2[(size_t)q * nsample + i] = write_sqrt ? sqrtf(d) : d;
        }
    }
}


// kNN, one WAVE per query (the path the model uses; knn_kernel above is the thread-per-query form kept for tiny inputs):
// the 64 lanes test 64 consecutive support points per step (coalesced AoS loads); lanes whose distance beats the
// current maximum are taken IN INDEX ORDER and re-tested against the maximum as it evolves -- the reference's sequential
// scan (same strict '<'), only the distance evaluations run 64-wide.  Literal form: candidates staged by ballot, inserted by
// lane 0 into a max-heap in LDS (same reheap / heap_sort as knnquery_cuda_kernel.cu:21-48).
#define KNNW_WAVES 4
// the literal scan of ONE query by one wave (heap in LDS): identical to the reference down to the order of tied results
__device__ __forceinline__ void knn_literal_query(int q, int start, int end, int nsample, int lane, const float* __restrict__ xyz,
                                                  const float* __restrict__ new_xyz, float* hd, int* hi, float* cand_d, int* cand_i,
                                                  int* __restrict__ idx, float* __restrict__ dist2, int write_sqrt) {
        const float qx = new_xyz[(size_t)q * 3], qy = new_xyz[(size_t)q * 3 + 1], qz = new_xyz[(size_t)q * 3 + 2];
        for (int i = lane; i < nsample; i += 64) { hd[i] = 1e10f; hi[i] = start; }
        __builtin_amdgcn_wave_barrier();
        float top = 1e10f;
        for (int k0 = start; k0 < end; k0 += 64) {
            const int k = k0 + lane;
            float d2 = 3e38f;
            if (k < end) d2 = etch_sqdist(qx, qy, qz, xyz[(size_t)k * 3], xyz[(size_t)k * 3 + 1], xyz[(size_t)k * 3 + 2]);
            const bool c = d2 < top;
            const unsigned long long mask = __ballot(c);
            if (mask == 0ull) continue;                            // wave-uniform
            const int pos = __popcll(mask & ((1ull << lane) - 1ull));
            if (c) { cand_d[pos] = d2; cand_i[pos] = k; }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                const int nc = __popcll(mask);
                float t = top;
                for (int e = 0; e < nc; ++e) {
                    const float d = cand_d[e];
                    if (d < t) {                                   // re-test: the top shrinks as candidates go in
                        hd[0] = d; hi[0] = cand_i[e];
                        int root = 0, child = 1;                   // reheap (knnquery_cuda_kernel.cu:21-37)
                        while (child < nsample) {
                            if (child + 1 < nsample && hd[child + 1] > hd[child]) child++;
                            if (hd[root] > hd[child]) break;
                            const float tf = hd[root]; hd[root] = hd[child]; hd[child] = tf;
                            const int ti = hi[root]; hi[root] = hi[child]; hi[child] = ti;
                            root = child; child = root * 2 + 1;
                        }
                        t = hd[0];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            top = hd[0];
        }
        if (lane == 0) {
            for (int i = nsample - 1; i > 0; i--) {                // heap_sort (:39-48)
                float tf = hd[0]; hd[0] = hd[i]; hd[i] = tf;
                int ti = hi[0]; hi[0] = hi[i]; hi[i] = ti;
                int root = 0, child = 1;
                while (child < i) {
                    if (child + 1 < i && hd[child + 1] > hd[child]) child++;
                    if (hd[root] > hd[child]) break;
                    const float tf2 = hd[root]; hd[root] = hd[child]; hd[child] = tf2;
                    const int ti2 = hi[root]; hi[root] = hi[child]; hi[child] = ti2;
                    root = child; child = root * 2 + 1;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < nsample; i += 64) {
            idx[(size_t)q * nsample + i] = hi[i];
            const float d = hd[i];
            dist2[(size_t)q * nsample + i] = write_sqrt ? sqrtf(d) : d;
        }
        __builtin_amdgcn_wave_barrier();
}

// One wave per query.  Fast path (nsample <= 16): the running k best live SORTED in the registers of lanes 0..k-1; a candidate
// (taken in index order, re-tested with the same strict '<' against the current maximum as the reference's heap top) is placed
// with one ballot + one DPP shift instead of an LDS reheap by a single lane -- the kernel was insertion-bound (k = 16 cost 5x
// k = 1).  The SET of results evolves exactly like the reference's heap; its ascending output order is unique unless two results
// have equal distances, and exactly then the query is redone by the literal heap scan above, so the output is the reference's in
// every case.
__global__ void __launch_bounds__(KNNW_WAVES * 64) knn_wave_kernel(int nsample, int nseg, const float* __restrict__ xyz,
                                                                    const float* __restrict__ new_xyz, const int* __restrict__ offset,
                                                                    const int* __restrict__ new_offset, int m_total, int* __restrict__ idx,
                                                                    float* __restrict__ dist2, int write_sqrt) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* hd = smem_f + wave * 2 * nsample;          // heap distances [k]   (literal path)
    int* hi = (int*)(hd + nsample);                   // heap indices   [k]
    float* cand_d = smem_f + KNNW_WAVES * 2 * nsample + wave * 128;   // candidates of the current chunk
    int* cand_i = (int*)(cand_d + 64);
    for (int q = blockIdx.x * KNNW_WAVES + wave; q < m_total; q += gridDim.x * KNNW_WAVES) {
        int seg = 0;
        while (seg < nseg - 1 && q >= new_offset[seg]) ++seg;
        const int start = seg == 0 ? 0 : offset[seg - 1], end = offset[seg];
        if (nsample > 16) {
            knn_literal_query(q, start, end, nsample, lane, xyz, new_xyz, hd, hi, cand_d, cand_i, idx, dist2, write_sqrt);
            continue;
        }
        const float qx = new_xyz[(size_t)q * 3], qy = new_xyz[(size_t)q * 3 + 1], qz = new_xyz[(size_t)q * 3 + 2];
        float bd = 1e10f;                             // lane l < k: l-th smallest distance so far (heap initialisation: (1e10, start))
        int bi = start;
        float top = 1e10f;
        bool ambiguous = false;                       // wave-uniform: an eviction happened while two DIFFERENT points tied for the maximum
        for (int k0 = start; k0 < end; k0 += 64) {
            const int k = k0 + lane;
            float d2 = 3e38f;
            if (k < end) d2 = etch_sqdist(qx, qy, qz, xyz[(size_t)k * 3], xyz[(size_t)k * 3 + 1], xyz[(size_t)k * 3 + 2]);
            unsigned long long mask = __ballot(d2 < top);
            while (mask) {                                                   // wave-uniform: candidates in index order
                const int e = __builtin_ctzll(mask);
                mask &= mask - 1ull;
                const float cd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d2), e));
                if (cd < top) {                                              // re-test: the maximum shrinks as candidates go in
                    if (nsample > 1) {
                        // the sorted list evicts its LAST entry; the reference's reheap evicts the heap root.  They are the same point
                        // unless two different points tie for the maximum -- then which one survives depends on the heap's shape, and
                        // the survivor may be the only trace left at the end: remember it and redo the query literally.
                        const float t2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bd), nsample - 2));
                        if (t2 == top && __builtin_amdgcn_readlane(bi, nsample - 1) != __builtin_amdgcn_readlane(bi, nsample - 2)) ambiguous = true;
                    }
                    const int pos = __popcll(__ballot(lane < nsample && bd <= cd));      // sorted list: a prefix
                    const float pd = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bd), 0x111, 0xf, 0xf, true));   // row_shr:1
                    const int pi = __builtin_amdgcn_update_dpp(0, bi, 0x111, 0xf, 0xf, true);
                    if (lane == pos) { bd = cd; bi = k0 + e; }
                    else if (lane > pos) { bd = pd; bi = pi; }
                    top = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bd), nsample - 1));
                }
            }
        }
        // equal distances among the results (with different points): the order of the reference's heap_sort is not the sorted one
        const float prev = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bd), 0x111, 0xf, 0xf, true));
        const int previ = __builtin_amdgcn_update_dpp(0, bi, 0x111, 0xf, 0xf, true);
        const bool tie = lane > 0 && lane < nsample && bd == prev && bi != previ;
        if (ambiguous || __ballot(tie) != 0ull) {
            knn_literal_query(q, start, end, nsample, lane, xyz, new_xyz, hd, hi, cand_d, cand_i, idx, dist2, write_sqrt);
            continue;
        }
        if (lane < nsample) {
            idx[(size_t)q * nsample + lane] = bi;
            dist2[(size_t)q * nsample + lane] = write_sqrt ? sqrtf(bd) : bd;
        }
    }
}

// ------------------------------------------------------------------------------------ C ABI
static inline int ilog2_floor_host(int v) { int b = 0; while ((1 << (b + 1)) <= v) ++b; return b; }

// opt_n_threads(): min(1024, 2^floor(log2 n)) computed like the reference host code (double log ratio)
static int opt_n_threads_host(int work_size) {
    const int pow_2 = (int)(std::log((double)work_size) / std::log(2.0));
    int v = 1 << pow_2;
    if (v > 1024) v = 1024;
    if (v < 1) v = 1;
    return v;
}

static int g_fps_fast = 1;     // etch_fps_split_debug's third knob: 0 = the 64-bit-key form everywhere (A/B timing, tests)

// nseg1 > 0: a second problem over the same n_max in the same launch (the workgroups behind the first problem's)
static int launch_fps2(int nseg0, int nseg1, int n_max, const FpsProb& p0, const FpsProb& p1, hipStream_t st) {
    const int bs = opt_n_threads_host(n_max);
    const int bits = ilog2_floor_host(bs);
    const bool fits_lds = (size_t)3 * n_max * sizeof(float) <= 62 * 1024;
    const int nseg = nseg0 + nseg1;
#define FPS_CASE(T, P, R, C, COND)                                                                                     \
    if (n_max <= T * P && (COND)) {                                                                                    \
        const int use_lds = !C && fits_lds ? n_max : 0;                                                                \
        if (R && C && T == 1024 && bs == T && g_fps_fast)     /* large scans only: at <= 8 slots the branchy loop skips the empty ones */ \
            hipLaunchKernelGGL((fps_kernel<T, P, R, C, (R && C && T == 1024)>), dim3(nseg), dim3(T), (size_t)3 * use_lds * sizeof(float), st,   \
                               use_lds, p0, p1, nseg0, bs, bits);                                                      \
        else                                                                                                           \
            hipLaunchKernelGGL((fps_kernel<T, P, R, C>), dim3(nseg), dim3(T), (size_t)3 * use_lds * sizeof(float), st, \
                               use_lds, p0, p1, nseg0, bs, bits);                                                      \
        ETCH_RETURN_IF_LAUNCH_FAILED();                                                                                \
        return ETCH_OK;                                                                                                \
    }
    FPS_CASE(64, 1, true, false, true) FPS_CASE(64, 4, true, false, true) FPS_CASE(256, 2, true, false, true) FPS_CASE(256, 8, true, false, true)
    FPS_CASE(1024, 4, true, false, true) FPS_CASE(1024, 8, true, false, fits_lds) FPS_CASE(1024, 8, true, true, true)
    FPS_CASE(1024, 12, true, true, true) FPS_CASE(1024, 20, true, true, true) FPS_CASE(1024, 32, false, false, true)
#undef FPS_CASE
    return ETCH_EUNSUPPORTED;  // more than 32768 points per segment
}

static int launch_fps(int nseg, int n_max, const float* xyz, long cs, long ps, long bstride, int n_fixed, int m_fixed,
                      const int* offset, const int* new_offset, int skip_origin, int* idx, hipStream_t st) {
    const FpsProb p = {xyz, cs, ps, bstride, n_fixed, m_fixed, offset, new_offset, skip_origin, idx};
    return launch_fps2(nseg, 0, n_max, p, p, st);
}

static unsigned g_fps_spin_limit = 1u << 22;
static int g_fps_drop_group = -1;

static int launch_fps_split(int nseg, int n_max, int G, const float* xyz, long cs, long ps, long bstride, int n_fixed, int m_fixed, const int* offset,
                            const int* new_offset, int skip_origin, int* idx, void* workspace, hipStream_t st) {
    if (G < 2 || G > 8 || !workspace) return ETCH_EINVAL;
    const int bs = opt_n_threads_host(n_max);
    const int bits = ilog2_floor_host(bs);
    const int chunks = (n_max + 1023) / 1024, need = (chunks + G - 1) / G;
    const size_t bytes = 64 + (size_t)nseg * 2 * G * 64;
    hipError_t e = hipMemsetAsync(workspace, 0, bytes, st);                    // tags and the fail word: zero before EVERY launch
    if (e != hipSuccess) return (int)e;
#define FPS_SPLIT_CASE(P)                                                                                                                  \
    if (need <= P) {                                                                                                                       \
        hipLaunchKernelGGL((fps_split_kernel<P>), dim3(nseg * G), dim3(1024), 0, st, G, xyz, cs, ps, bstride, n_fixed, m_fixed, offset,    \
                           new_offset, bs, bits, skip_origin, g_fps_spin_limit, g_fps_drop_group, (unsigned long long*)workspace, idx);    \
        ETCH_RETURN_IF_LAUNCH_FAILED();                                                                                                    \
        return ETCH_OK;                                                                                                                    \
    }
    FPS_SPLIT_CASE(2) FPS_SPLIT_CASE(3) FPS_SPLIT_CASE(4) FPS_SPLIT_CASE(5) FPS_SPLIT_CASE(6) FPS_SPLIT_CASE(8)
#undef FPS_SPLIT_CASE
    return ETCH_EUNSUPPORTED;      // more than 8 x 1024 x G points per segment
}

extern "C" {

int etch_fps_split_workspace_bytes(int nseg, int G) { return nseg > 0 && G > 0 ? 64 + nseg * 2 * G * 64 : 64; }

int etch_furthest_point_sampling_split(int b, int n, int m, const float* xyz, int* idx, int G, void* workspace, void* stream) {
    if (b <= 0 || m <= 0) return ETCH_OK;
    if (n <= 0) return ETCH_EINVAL;
    return launch_fps_split(b, n, G, xyz, (long)n, 1, (long)3 * n, n, m, nullptr, nullptr, 1, idx, workspace, (hipStream_t)stream);
}

int etch_furthestsampling_split(int b, int n_max, const float* xyz, const int* offset, const int* new_offset, int* idx, int G, void* workspace,
                                void* stream) {
    if (b <= 0) return ETCH_OK;
    if (n_max <= 0) return ETCH_EINVAL;
    return launch_fps_split(b, n_max, G, xyz, 1, 3, 0, 0, 0, offset, new_offset, 0, idx, workspace, (hipStream_t)stream);
}

int etch_fps_split_failed(const void* workspace, int* failed, void* stream) {
    if (!workspace || !failed) return ETCH_EINVAL;
    unsigned v = 0;
    hipError_t e = hipMemcpyAsync(&v, workspace, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    *failed = v != 0;
    return ETCH_OK;
}

int etch_fps_split_debug(unsigned spin_limit, int drop_group) {
    g_fps_spin_limit = spin_limit ? spin_limit : (1u << 22);
    g_fps_drop_group = drop_group;
    return ETCH_OK;
}

int etch_fps_fast(int enable) {
    g_fps_fast = enable != 0;
    return ETCH_OK;
}

int etch_ball_query(int b, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz, int* idx,
                    void* stream) {
    if (b <= 0 || m <= 0) return ETCH_OK;
    if (n <= 0 || nsample <= 0 || nsample > 4096) return ETCH_EINVAL;
    constexpr int WAVES = 4;
    const float r2 = radius * radius;
    int gx = (m + WAVES - 1) / WAVES;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL((ball_query_kernel<WAVES>), dim3(gx, b), dim3(WAVES * 64), WAVES * nsample * sizeof(int),
                       (hipStream_t)stream, n, m, r2, nsample, new_xyz, xyz, idx);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_furthest_point_sampling(int b, int n, int m, const float* xyz, int* idx, void* stream) {
    if (b <= 0 || m <= 0) return ETCH_OK;
    if (n <= 0) return ETCH_EINVAL;
    return launch_fps(b, n, xyz, (long)n, 1, (long)3 * n, n, m, nullptr, nullptr, 1, idx, (hipStream_t)stream);
}

int etch_furthestsampling(int b, int n_max, const float* xyz, const int* offset, const int* new_offset, int* idx,
                          void* stream) {
    if (b <= 0) return ETCH_OK;
    if (n_max <= 0) return ETCH_EINVAL;
    return launch_fps(b, n_max, xyz, 1, 3, 0, 0, 0, offset, new_offset, 0, idx, (hipStream_t)stream);
}

// Both samplings of one batch of b equally sized scans in ONE launch of 2 b workgroups: etch_furthest_point_sampling(b, n, m, xyz_b3n, idx_a) and
// etch_furthestsampling(b, n, xyz_packed, offset, new_offset, idx_b) -- same picks as the two separate launches, bit for bit (the kernel is the same; a
// workgroup reads one problem or the other).
int etch_fps_pair(int b, int n, int m, const float* xyz_b3n, int* idx_a, const float* xyz_packed, const int* offset, const int* new_offset, int* idx_b,
                  void* stream) {
    if (b <= 0) return ETCH_OK;
    if (n <= 0 || m <= 0 || !xyz_b3n || !xyz_packed || !offset || !new_offset || !idx_a || !idx_b) return ETCH_EINVAL;
    const FpsProb p0 = {xyz_b3n, (long)n, 1, (long)3 * n, n, m, nullptr, nullptr, 1, idx_a};
    const FpsProb p1 = {xyz_packed, 1, 3, 0, 0, 0, offset, new_offset, 0, idx_b};
    return launch_fps2(b, b, n, p0, p1, (hipStream_t)stream);
}

int etch_gather_points(int b, int c, int n, int m, const float* points, const int* idx, float* out, void* stream) {
    if (b <= 0 || m <= 0 || c <= 0) return ETCH_OK;
    int gx = (m + 255) / 256;
    if (gx > 8192) gx = 8192;
    hipLaunchKernelGGL(gather_points_kernel, dim3(gx, b), dim3(256), 0, (hipStream_t)stream, c, n, m, points, idx, out);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_gather_points_backward(int b, int c, int n, int m, const float* grad_out, const int* idx, float* grad_points, void* stream) {
    if (b <= 0 || n <= 0 || c <= 0) return ETCH_OK;
    if (m < 0) return ETCH_EINVAL;
    int gx = (n + 255) / 256;
    if (gx > 8192) gx = 8192;
    hipLaunchKernelGGL(gather_points_backward_kernel, dim3(gx, b), dim3(256), 0, (hipStream_t)stream, c, n, m, grad_out, idx, grad_points);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_knnquery(int b, int m_max, int m_total, int nsample, const float* xyz, const float* new_xyz, const int* offset,
                  const int* new_offset, int* idx, float* dist, int write_sqrt, void* stream) {
    if (b <= 0 || m_max <= 0) return ETCH_OK;
    if (nsample <= 0) return ETCH_EINVAL;
    if (m_total > 0) {            // wave-per-query kernel: nsample <= 16 on the register list, up to the reference's 100 (knnquery_cuda_kernel.cu:86-87) on the LDS heap
        if (nsample > 100) return ETCH_EUNSUPPORTED;
        const size_t lds = (size_t)(KNNW_WAVES * 2 * nsample + KNNW_WAVES * 128) * 4;
        long blocks = ((long)m_total + KNNW_WAVES - 1) / KNNW_WAVES;
        if (blocks > 256 * 64) blocks = 256 * 64;
        hipLaunchKernelGGL(knn_wave_kernel, dim3((unsigned)blocks), dim3(KNNW_WAVES * 64), lds, (hipStream_t)stream, nsample, b, xyz,
                           new_xyz, offset, new_offset, m_total, idx, dist, write_sqrt);
        ETCH_RETURN_IF_LAUNCH_FAILED();
        return ETCH_OK;
    }
    if (nsample > 28) return ETCH_EUNSUPPORTED;  // thread-per-query form: heap columns must fit 64 KiB of LDS
    const size_t lds = (size_t)(3 * KNN_TILE + 2 * nsample * KNN_THREADS) * 4;
    hipLaunchKernelGGL(knn_kernel, dim3((m_max + KNN_THREADS - 1) / KNN_THREADS, b), dim3(KNN_THREADS), lds,
                       (hipStream_t)stream, nsample, xyz, new_xyz, offset, new_offset, idx, dist, write_sqrt);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

// Segment lookup on the device (knnquery_cuda_kernel.cu:52-62 get_bt_idx: the scan over new_offset ends because q < m = new_offset[b-1]):
// the caller needs neither the number of segments nor the largest one -- no host read of the offsets.
// nseg = number of segments (entries of offset / new_offset) when the caller knows it -- a tensor's element count, no device read: the segment
// scan of a query is then clamped to it, so a query index past new_offset[nseg - 1] (m larger than the offsets cover) cannot run off the arrays.
int etch_knnquery_dev_bounded(int m, int nsample, int nseg, const float* xyz, const float* new_xyz, const int* offset, const int* new_offset,
                              int* idx, float* dist, int write_sqrt, void* stream) {
    if (m <= 0) return ETCH_OK;
    if (nsample <= 0 || nseg <= 0) return ETCH_EINVAL;
    if (nsample > 100) return ETCH_EUNSUPPORTED;          // the reference's best_dist[100] (knnquery_cuda_kernel.cu:86-87)
    const size_t lds = (size_t)(KNNW_WAVES * 2 * nsample + KNNW_WAVES * 128) * 4;
    long blocks = ((long)m + KNNW_WAVES - 1) / KNNW_WAVES;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(knn_wave_kernel, dim3((unsigned)blocks), dim3(KNNW_WAVES * 64), lds, (hipStream_t)stream, nsample, nseg, xyz,
                       new_xyz, offset, new_offset, m, idx, dist, write_sqrt);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

// The reference launcher's contract (no segment count in its signature): the scan ends because q < m = new_offset[b - 1].
int etch_knnquery_dev(int m, int nsample, const float* xyz, const float* new_xyz, const int* offset, const int* new_offset, int* idx,
                      float* dist, int write_sqrt, void* stream) {
    return etch_knnquery_dev_bounded(m, nsample, 0x7fffffff, xyz, new_xyz, offset, new_offset, idx, dist, write_sqrt, stream);
}

// ---- the reference's launchers under their own names and signatures (null stream, void, errors printed like grouping_cuda_kernel.cu:486-488)
static void report_launcher(const char* name, int rc) {
    if (rc != ETCH_OK) fprintf(stderr, "Error: %s failed (%d%s%s)\n", name, rc, rc > 0 ? ": " : "", rc > 0 ? hipGetErrorString((hipError_t)rc) : "");
}

void knnquery_cuda_launcher(int m, int nsample, const float* xyz, const float* new_xyz, const int* offset, const int* new_offset, int* idx,
                            float* dist2) {
    report_launcher("knnquery_cuda_launcher", etch_knnquery_dev(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, 0, nullptr));
}

void furthestsampling_cuda_launcher(int b, int n, const float* xyz, const int* offset, const int* new_offset, float* tmp, int* idx) {
    (void)tmp;      // the reference's running minimum distances (caller-filled with 1e10, sampling_cuda_kernel.cu:24-60) live in registers here
    report_launcher("furthestsampling_cuda_launcher", etch_furthestsampling(b, n, xyz, offset, new_offset, idx, nullptr));
}

void ball_query_cuda_launcher(int b, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz, int* idx) {
    report_launcher("ball_query_cuda_launcher", etch_ball_query(b, n, m, radius, nsample, new_xyz, xyz, idx, nullptr));
}

void furthest_point_sampling_cuda_launcher(int b, int n, int m, const float* dataset, float* temp, int* idxs) {
    (void)temp;     // grouping_cuda_kernel.cu:352-466 keeps the running minimum distances in `temp` [b, n]; registers here
    report_launcher("furthest_point_sampling_cuda_launcher", etch_furthest_point_sampling(b, n, m, dataset, idxs, nullptr));
}

void gather_points_forward_cuda_launcher(int b, int c, int n, int m, const float* points, const int* idx, float* out) {
    report_launcher("gather_points_forward_cuda_launcher", etch_gather_points(b, c, n, m, points, idx, out, nullptr));
}

void gather_points_backward_cuda_launcher(int b, int c, int n, int m, const float* grad_out, const int* idx, float* grad_points) {
    report_launcher("gather_points_backward_cuda_launcher", etch_gather_points_backward(b, c, n, m, grad_out, idx, grad_points, nullptr));
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Spatial processing order of a scan's points (a scheduling aid, not part of the reference's semantics): 30-bit Morton keys on the
// scan's bounding box, sorted together with the point index (ties -> lower index) by a bitonic network in LDS, one workgroup
// per scan.  The fused inter conv walks its output points in this order, one contiguous eighth of the curve per XCD, so the
// workgroups sharing an L2 gather from the same source rows: HBM fetch traffic of those kernels drops 3x (12 -> 4 GB per launch).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned etch_spread10(unsigned v) {
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void __launch_bounds__(1024) spatial_order_kernel(int n, int np2, const float* __restrict__ xyz, int* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];     // [np2]
    __shared__ float red[6][16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* X = xyz + (size_t)b * 3 * n;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < n; i += 1024)
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = X[(size_t)a * n + i]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = lo[a], h = hi[a];
        for (int o = 32; o > 0; o >>= 1) { l = fminf(l, __shfl_xor(l, o)); h = fmaxf(h, __shfl_xor(h, o)); }
        if (lane == 0) { red[a][wave] = l; red[3 + a][wave] = h; }
    }
    __syncthreads();
    float mn[3], sc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = red[a][0], h = red[3 + a][0];
        for (int w = 1; w < 16; ++w) { l = fminf(l, red[a][w]); h = fmaxf(h, red[3 + a][w]); }
        mn[a] = l;
        sc[a] = h > l ? 1023.0f / (h - l) : 0.f;
    }
    for (int i = tid; i < np2; i += 1024) {
        unsigned long long k = ~0ull;                               // padding sorts to the end
        if (i < n) {
            unsigned q[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float t = (X[(size_t)a * n + i] - mn[a]) * sc[a];
                q[a] = (unsigned)fminf(fmaxf(t, 0.f), 1023.f);
            }
            const unsigned m = etch_spread10(q[0]) | (etch_spread10(q[1]) << 1) | (etch_spread10(q[2]) << 2);
            k = ((unsigned long long)m << 32) | (unsigned)i;
        }
        keys[i] = k;
    }
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (np2 >> 1); i += 1024) {
                const int lowi = ((i / stride) * (stride << 1)) + (i % stride), highi = lowi + stride;
                const bool up = ((lowi & size) == 0);
                const unsigned long long x = keys[lowi], y = keys[highi];
                if ((x > y) == up) { keys[lowi] = y; keys[highi] = x; }
            }
            __syncthreads();
        }
    for (int i = tid; i < n; i += 1024) order[(size_t)b * n + i] = (int)(unsigned)(keys[i] & 0xFFFFFFFFull);
}

// Scans of more than 16 384 points (BASELINE configs[4]: 20 000): 32-bit keys = 15-bit Morton code (32 cells per axis of the scan's
// bounding box) | 17-bit index within a 32 768-point slice, one workgroup per (scan, slice) -- 128 KiB of LDS again.  Points of one
// cell keep their index order; slices are sorted independently along the same curve (a scan of > 32 768 points is walked as that
// many interleaved curves).  Like the 64-bit form this is a scheduling hint only: any permutation gives the same results.
#define SO32_SLICE 32768
__global__ void __launch_bounds__(1024) spatial_order32_kernel(int n, const float* __restrict__ xyz, int* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned keys32[];            // [np2 of the slice]
    __shared__ float red[6][16];
    const int b = blockIdx.x, base = blockIdx.y * SO32_SLICE, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cnt = min(n - base, SO32_SLICE);
    int np2 = 2;
    while (np2 < cnt) np2 <<= 1;
    const float* X = xyz + (size_t)b * 3 * n;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < n; i += 1024)
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = X[(size_t)a * n + i]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = lo[a], h = hi[a];
        for (int o = 32; o > 0; o >>= 1) { l = fminf(l, __shfl_xor(l, o)); h = fmaxf(h, __shfl_xor(h, o)); }
        if (lane == 0) { red[a][wave] = l; red[3 + a][wave] = h; }
    }
    __syncthreads();
    float mn[3], sc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = red[a][0], h = red[3 + a][0];
        for (int w = 1; w < 16; ++w) { l = fminf(l, red[a][w]); h = fmaxf(h, red[3 + a][w]); }
        mn[a] = l;
        sc[a] = h > l ? 31.0f / (h - l) : 0.f;
    }
    for (int i = tid; i < np2; i += 1024) {
        unsigned k = ~0u;                                            // padding sorts to the end
        if (i < cnt) {
            unsigned q[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float t = (X[(size_t)a * n + base + i] - mn[a]) * sc[a];
                q[a] = (unsigned)fminf(fmaxf(t, 0.f), 31.f);
            }
            const unsigned m = etch_spread10(q[0]) | (etch_spread10(q[1]) << 1) | (etch_spread10(q[2]) << 2);     // 15 bits
            k = (m << 17) | (unsigned)i;
        }
        keys32[i] = k;
    }
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (np2 >> 1); i += 1024) {
                const int lowi = ((i / stride) * (stride << 1)) + (i % stride), highi = lowi + stride;
                const bool up = ((lowi & size) == 0);
                const unsigned x = keys32[lowi], y = keys32[highi];
                if ((x > y) == up) { keys32[lowi] = y; keys32[highi] = x; }
            }
            __syncthreads();
        }
    for (int i = tid; i < cnt; i += 1024) order[(size_t)b * n + base + i] = base + (int)(keys32[i] & 0x1FFFFu);
}

extern "C" int etch_spatial_order(int b, int n, const float* xyz, int* order, void* stream) {
    if (b <= 0 || n <= 0) return ETCH_OK;
    int np2 = 2;
    while (np2 < n) np2 <<= 1;
    if (np2 > 16384) {
        const int slices = (n + SO32_SLICE - 1) / SO32_SLICE;
        if (b > 65535 || slices > 65535) return ETCH_EUNSUPPORTED;
        const int lds = SO32_SLICE * (int)sizeof(unsigned);           // 128 KiB
        hipError_t e = hipFuncSetAttribute((const void*)spatial_order32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(spatial_order32_kernel, dim3(b, slices), dim3(1024), lds, (hipStream_t)stream, n, xyz, order);
        ETCH_RETURN_IF_LAUNCH_FAILED();
        return ETCH_OK;
    }
    const size_t lds = (size_t)np2 * sizeof(unsigned long long);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)spatial_order_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(spatial_order_kernel, dim3(b), dim3(1024), lds, (hipStream_t)stream, n, np2, xyz, order);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}
