// One whole MultiHeadAttention layer of the direction head (/root/reference/src/models/direction_backbones.py:132-194:
// key/query/value_transform -> 8-head DotProdAttention over the 60 anchor tokens of a point (:102-129) -> head_combine,
// and the residual add of StackedMHSA.forward :216-221) as ONE kernel on the gfx950 fp32 matrix cores.  The per-token
// q|k|v rows (9.6 M x 192 floats per batch) and the attention output never touch HBM: a point's 60 x 64 tokens are read
// once, everything else lives in registers / LDS.
//
// Persistent workgroup = 4 waves, one point at a time; wave w owns heads 2w and 2w+1 end to end:
//   A  Q^T, K^T (16 channels x 64 tokens) and V (64 tokens x 16 channels) tiles of its two heads from the X tile in LDS;
//      the weight fragments sit in registers for the lifetime of the workgroup.  The 16 rows of the Q/K weight tile are
//      ordered so that accumulator register r of lane group g holds dim 2g + (r&1) of head 2w + (r>>1): the accumulators
//      ARE the MFMA operands of the score product (no LDS round trip, no zero padding of the 8-wide heads).
//   B  S^T = K Q^T per (head, 16-query tile): the lane owning query i holds 16 of its 64 scores; softmax needs two
//      cross-lane-group exchanges; the normalised P accumulators ARE the A operand of P V, V's accumulators the B operand.
//      Both heads share the V tile: head 2w's result lands in output columns 0-7, head 2w+1's in columns 8-15.
//   C  head_combine (+ bias + residual) from the attention tile in LDS, float4 stores.
// Keys 60..63 (tile padding) are masked to -inf; rows 60..63 are never stored.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ML_TOK 60
#define ML_C 64
#define ML_S 104     // LDS row stride (floats): 104/4 = 10 (mod 16) -> conflict-free ds_read_b128 fragment reads

#define ML_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// softmax over the 64 (60 valid) keys of one query, scores spread over the 4 lane groups x 16 registers; returns P / sum
__device__ __forceinline__ void ml_softmax(f32x4 (&s)[4], int fg) {
    const float c = 0.35355339059327373f * 1.4426950408889634f;     // 1/sqrt(8) * log2(e)
    if (fg == 3) s[3] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};   // keys 60..63 do not exist
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) m = fmaxf(fmaxf(fmaxf(s[j][0], s[j][1]), fmaxf(s[j][2], s[j][3])), m);
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    const float mc = m * c;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(fmaf(s[j][r], c, -mc));
            s[j][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] *= inv;
}

// MODE 0: out = X + att Wc^T + bc (residual layer);  1: out = att Wc^T + bc;  2: out = att (head_combine folded downstream)
template <int MODE>
__global__ void __launch_bounds__(256, MODE == 2 ? 3 : 2) mhsa_layer_kernel(long T, const float* __restrict__ X, const float* __restrict__ Wq,
                                                            const float* __restrict__ Wk, const float* __restrict__ Wv,
                                                            const float* __restrict__ Wc, const float* __restrict__ bc,
                                                            float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float Xs[64 * ML_S];
    __shared__ __attribute__((aligned(16))) float As[64 * ML_S];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;

    // weight fragments (interleaved-K: lane group fg holds k = 16t + 4fg + s of its row), resident in registers
    float4 wq[4], wk[4], wv[4], wc[4];
    {
        const int chq = 8 * (2 * w + ((fr & 3) >> 1)) + 2 * (fr >> 2) + (fr & 1);    // Q/K tile row fr -> original channel
        const int chv = 16 * w + fr;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            wq[t] = *reinterpret_cast<const float4*>(Wq + chq * ML_C + t * 16 + fg * 4);
            wk[t] = *reinterpret_cast<const float4*>(Wk + chq * ML_C + t * 16 + fg * 4);
            wv[t] = *reinterpret_cast<const float4*>(Wv + chv * ML_C + t * 16 + fg * 4);
            if (MODE != 2) wc[t] = *reinterpret_cast<const float4*>(Wc + chv * ML_C + t * 16 + fg * 4);
        }
    }
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE != 2 && bc) bias = *reinterpret_cast<const float4*>(bc + 16 * w + 4 * fg);
    Xs[(ML_TOK + (tid >> 6)) * ML_S + (tid & 63)] = 0.f;      // token rows 60..63 stay zero for the whole kernel

    // next point's tokens: 960 float4 over 256 threads, four named registers (an indexed array captured by a lambda ended up
    // in scratch memory: +5 GB of HBM traffic per launch)
    float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0, x2 = x0, x3 = x0;
#define ML_GLOAD(P)                                                                          \
    {                                                                                        \
        const float4* src_ = reinterpret_cast<const float4*>(X + (P) * (ML_TOK * ML_C));     \
        x0 = src_[tid]; x1 = src_[tid + 256]; x2 = src_[tid + 512];                          \
        if (tid < ML_TOK * ML_C / 4 - 768) x3 = src_[tid + 768];                             \
    }
    long pt = blockIdx.x;
    if (pt < T) ML_GLOAD(pt)
    for (; pt < T; pt += gridDim.x) {
        *reinterpret_cast<float4*>(&Xs[(tid >> 4) * ML_S + (tid & 15) * 4]) = x0;
        *reinterpret_cast<float4*>(&Xs[((tid >> 4) + 16) * ML_S + (tid & 15) * 4]) = x1;
        *reinterpret_cast<float4*>(&Xs[((tid >> 4) + 32) * ML_S + (tid & 15) * 4]) = x2;
        if (tid < ML_TOK * ML_C / 4 - 768) *reinterpret_cast<float4*>(&Xs[((tid >> 4) + 48) * ML_S + (tid & 15) * 4]) = x3;
        __syncthreads();
        if (pt + gridDim.x < T) ML_GLOAD(pt + gridDim.x)      // next point's tokens: in flight during the whole layer

        // ---- A: projections of this wave's two heads
        f32x4 Q[4], Kt[4], V[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            float4 xf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) xf[t] = *reinterpret_cast<const float4*>(&Xs[(tt * 16 + fr) * ML_S + t * 16 + fg * 4]);
            f32x4 q = {0.f, 0.f, 0.f, 0.f}, k = q, v = q;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#define ML_STEP(C)                     \
    q = ML_MFMA(wq[t].C, xf[t].C, q);  \
    k = ML_MFMA(wk[t].C, xf[t].C, k);  \
    v = ML_MFMA(xf[t].C, wv[t].C, v);
                ML_STEP(x) ML_STEP(y) ML_STEP(z) ML_STEP(w)
#undef ML_STEP
            }
            Q[tt] = q; Kt[tt] = k; V[tt] = v;
        }

        // ---- B: attention of heads 2w (registers 0,1 / output columns 0-7) and 2w+1 (registers 2,3 / columns 8-15)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            f32x4 sa[4], sb[4];
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                sa[jt] = ML_MFMA(Kt[jt][0], Q[it][0], z);
                sb[jt] = ML_MFMA(Kt[jt][2], Q[it][2], z);
            }
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                sa[jt] = ML_MFMA(Kt[jt][1], Q[it][1], sa[jt]);
                sb[jt] = ML_MFMA(Kt[jt][3], Q[it][3], sb[jt]);
            }
            ml_softmax(sa, fg);
            ml_softmax(sb, fg);
            f32x4 oa = z, ob = z;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    oa = ML_MFMA(sa[jt][r], V[jt][r], oa);
                    ob = ML_MFMA(sb[jt][r], V[jt][r], ob);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) As[(it * 16 + fg * 4 + r) * ML_S + 16 * w + fr] = fr < 8 ? oa[r] : ob[r];
        }
        __syncthreads();

        float* dst = out + pt * (ML_TOK * ML_C);
        if (MODE == 2) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int e = tid + 256 * h;
                if (e < ML_TOK * ML_C / 4)
                    reinterpret_cast<float4*>(dst)[e] = *reinterpret_cast<const float4*>(&As[(e >> 4) * ML_S + (e & 15) * 4]);
            }
        } else {
            // ---- C: head_combine, transposed (rows = this wave's 16 output channels, columns = tokens) -> float4 stores
            f32x4 y[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) y[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float4 af[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) af[tt] = *reinterpret_cast<const float4*>(&As[(tt * 16 + fr) * ML_S + t * 16 + fg * 4]);
#define ML_STEP(C) _Pragma("unroll") for (int tt = 0; tt < 4; ++tt) y[tt] = ML_MFMA(wc[t].C, af[tt].C, y[tt]);
                ML_STEP(x) ML_STEP(y) ML_STEP(z) ML_STEP(w)
#undef ML_STEP
            }
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int tok = tt * 16 + fr;
                if (tok < ML_TOK) {
                    float4 o = make_float4(y[tt][0] + bias.x, y[tt][1] + bias.y, y[tt][2] + bias.z, y[tt][3] + bias.w);
                    if (MODE == 0) {
                        const float4 rx = *reinterpret_cast<const float4*>(&Xs[tok * ML_S + 16 * w + 4 * fg]);
                        o.x += rx.x; o.y += rx.y; o.z += rx.z; o.w += rx.w;
                    }
                    *reinterpret_cast<float4*>(dst + tok * ML_C + 16 * w + 4 * fg) = o;
                }
            }
        }
        __syncthreads();      // Xs / As are rewritten by the next point
    }
}

template <int MODE>
static int launch_layer(long T, const float* X, const float* Wq, const float* Wk, const float* Wv, const float* Wc, const float* bc,
                        float* out, hipStream_t st) {
    auto kern = mhsa_layer_kernel<MODE>;
    static int per_cu = 0;
    if (per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)kern, 256, 0) != hipSuccess || n < 1) n = 2;
        per_cu = n;
    }
    long blocks = 256L * per_cu;
    if (blocks > T) blocks = T;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, st, T, X, Wq, Wk, Wv, Wc, bc, out);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

extern "C" int etch_mhsa_layer(long T, const float* X, const float* Wq, const float* Wk, const float* Wv, const float* Wc,
                               const float* bc, int mode, float* out, void* stream) {
    if (T <= 0) return ETCH_OK;
    if (!X || !Wq || !Wk || !Wv || !out || (mode != 2 && !Wc)) return ETCH_EINVAL;
    if (((uintptr_t)X | (uintptr_t)Wq | (uintptr_t)Wk | (uintptr_t)Wv | (uintptr_t)Wc | (uintptr_t)bc | (uintptr_t)out) & 15) return ETCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) return launch_layer<0>(T, X, Wq, Wk, Wv, Wc, bc, out, st);
    if (mode == 1) return launch_layer<1>(T, X, Wq, Wk, Wv, Wc, bc, out, st);
    if (mode == 2) return launch_layer<2>(T, X, Wq, Wk, Wv, Wc, bc, out, st);
    return ETCH_EINVAL;
}
