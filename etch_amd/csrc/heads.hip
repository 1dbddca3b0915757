// Feature propagation and the direction head of GT_network_equiv for gfx950.
//   etch_prop3nn        3-NN search of PointFeatPropagation (/root/reference/src/models/pointnet2_utils.py:45-70)
//   etch_prop_interp    weighted gather of the 3 coarse feature rows (:71) + anchor mean (models_pointcloud.py:184)
//   etch_mhsa_attention DotProdAttention inside MultiHeadAttention (src/models/direction_backbones.py:102-129,160-194)
//   etch_rowdot         so3_reg = Conv1d(128, 1, 1) (models_pointcloud.py:54,117)
//   etch_so3_mean_dir   so3_mean + R @ [0,0,1] (src/models/so3conv.py:186-225, models_pointcloud.py:120-124)
#include "common.h"

// ------------------------------------------------------------------------------------------------
// 3-NN with the reference's expansion formula  d = ((-2 * dot) + |x|^2) + |y|^2  (pointnet2_utils.py:20-22):
// dot is the K = 3 fma chain of an fp32 GEMM, the squared norms are plain sums.  One thread per
// fine point, coarse points staged in LDS (SoA + norm).  Output: idx (B,N,3) i32 and the normalised
// inverse-distance weights w = (1/(d+1e-8)) / sum  (B,N,3).
// ------------------------------------------------------------------------------------------------
#define P3_TILE 1024
__global__ void __launch_bounds__(256) prop3nn_kernel(int N, int S, const float* __restrict__ xyz1 /*B,N,3*/,
                                                      const float* __restrict__ xyz2 /*B,3,S*/, int* __restrict__ idx,
                                                      float* __restrict__ weight) {
    __shared__ float sx[P3_TILE], sy[P3_TILE], sz[P3_TILE], sn[P3_TILE];
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool act = i < N;
    float x = 0, y = 0, z = 0, nx = 0;
    if (act) {
        const float* p = xyz1 + ((size_t)b * N + i) * 3;
        x = p[0]; y = p[1]; z = p[2];
        {
#pragma clang fp contract(off)
            nx = (x * x + y * y) + z * z;
        }
    }
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    int i0 = 0, i1 = 0, i2 = 0;
    const float* c = xyz2 + (size_t)b * 3 * S;
    for (int s0 = 0; s0 < S; s0 += P3_TILE) {
        const int cnt = min(P3_TILE, S - s0);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt; e += 256) {
            const float a = c[s0 + e], bb = c[S + s0 + e], cc = c[2 * S + s0 + e];
            sx[e] = a; sy[e] = bb; sz[e] = cc;
            {
#pragma clang fp contract(off)
                sn[e] = (a * a + bb * bb) + cc * cc;
            }
        }
        __syncthreads();
        if (act) {
            for (int e = 0; e < cnt; ++e) {
                const float dot = fmaf(z, sz[e], fmaf(y, sy[e], x * sx[e]));
                float d;
                {
#pragma clang fp contract(off)
                    d = (-2.0f * dot + nx) + sn[e];
                }
                const int id = s0 + e;
                if (d < d2) {          // strict: earlier index wins ties (stable ascending order)
                    if (d < d1) {
                        d2 = d1; i2 = i1;
                        if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = id; }
                        else { d1 = d; i1 = id; }
                    } else { d2 = d; i2 = id; }
                }
            }
        }
    }
    if (act) {
        const float r0 = 1.0f / (d0 + 1e-8f), r1 = 1.0f / (d1 + 1e-8f), r2 = 1.0f / (d2 + 1e-8f);
        const float nrm = (r0 + r1) + r2;
        const size_t o = ((size_t)b * N + i) * 3;
        idx[o] = i0; idx[o + 1] = i1; idx[o + 2] = i2;
        weight[o] = r0 / nrm; weight[o + 1] = r1 / nrm; weight[o + 2] = r2 / nrm;
    }
}

// out[b,n,a,c] = (f0*w0 + f1*w1) + f2*w2 over the coarse rows idx[b,n,0..2];  inv[b,n,c] = mean_a out[b,n,a,c].
// One workgroup per fine point; feats coarse (B,S,A,C) channels-last.
template <int C>
__global__ void __launch_bounds__(256) prop_interp_kernel(int N, int S, int A, const float* __restrict__ feats,
                                                          const int* __restrict__ idx, const float* __restrict__ weight,
                                                          float* __restrict__ out, float* __restrict__ inv, const int* __restrict__ order) {
    // grid (N or 8*ceil(N/8), B).  With `order` (a spatial order of each scan's fine points) the workgroups walk it, one contiguous
    // eighth per XCD (workgroup ids go round-robin over the XCDs): neighbouring fine points interpolate from the same coarse rows,
    // which then hit in the XCD's L2 instead of being fetched again (5.8 -> 1.3 GB read per launch)
    const int b = blockIdx.y;
    int n = blockIdx.x;
    if (order) {
        const int per = gridDim.x >> 3;
        const int slot = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (slot >= N) return;
        n = order[(size_t)b * N + slot];
    }
    const size_t pt = (size_t)b * N + n;
    const int i0 = idx[pt * 3], i1 = idx[pt * 3 + 1], i2 = idx[pt * 3 + 2];
    const float w0 = weight[pt * 3], w1 = weight[pt * 3 + 1], w2 = weight[pt * 3 + 2];
    const float4* f0 = reinterpret_cast<const float4*>(feats + ((size_t)b * S + i0) * A * C);
    const float4* f1 = reinterpret_cast<const float4*>(feats + ((size_t)b * S + i1) * A * C);
    const float4* f2 = reinterpret_cast<const float4*>(feats + ((size_t)b * S + i2) * A * C);
    float4* o = reinterpret_cast<float4*>(out + pt * A * C);
    constexpr int C4 = C / 4;
    // thread -> fixed channel quad (tid % C4), strided anchors: 256 / C4 anchors per pass
    const int c4 = threadIdx.x % C4, a0 = threadIdx.x / C4;
    constexpr int AP = 256 / C4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = a0; a < A; a += AP) {
        const int e = a * C4 + c4;
        const float4 u = f0[e], v = f1[e], w = f2[e];
        float4 r;
        {
#pragma clang fp contract(off)
            r.x = (u.x * w0 + v.x * w1) + w.x * w2; r.y = (u.y * w0 + v.y * w1) + w.y * w2;
            r.z = (u.z * w0 + v.z * w1) + w.z * w2; r.w = (u.w * w0 + v.w * w1) + w.w * w2;
        }
        o[e] = r;
        acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
    }
    // reduce the AP partial sums per channel quad through LDS
    __shared__ float4 part[256];
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < C4) {
        float4 s = part[threadIdx.x];
        for (int k = 1; k < AP; ++k) {
            const float4 t = part[k * C4 + threadIdx.x];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        const float ia = 1.0f / (float)A;
        reinterpret_cast<float4*>(inv + pt * C)[threadIdx.x] = make_float4(s.x * ia, s.y * ia, s.z * ia, s.w * ia);
    }
}

// ------------------------------------------------------------------------------------------------
// Multi-head self-attention core over the 60 anchor tokens of one point: per head h (dim 8)
//   out_i = sum_j softmax_j(q_i . k_j / sqrt(8)) v_j
// qkv rows: [T*60][ld] with q at column qoff, k at koff, v at voff (one fused QKV GEMM output).
// One workgroup (512 threads = 8 waves) per point, wave = head, lane = query token.
// ------------------------------------------------------------------------------------------------
#define MH_L 60
#define MH_HD 8
__global__ void __launch_bounds__(512) mhsa_attention_kernel(const float* __restrict__ qkv, long ld, int qoff, int koff, int voff,
                                                             float inv_sqrt_dk, float* __restrict__ out, long ldo) {
    __shared__ float ks[8][MH_L][MH_HD];
    __shared__ float vs[8][MH_L][MH_HD];
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
    const size_t row0 = (size_t)blockIdx.x * MH_L;
    // stage K and V of this head: 60 rows x 8 floats = 2 float4 per row
    for (int e = lane; e < MH_L * 2; e += 64) {
        const int r = e >> 1, half = e & 1;
        const float* base = qkv + (row0 + r) * ld + h * MH_HD + half * 4;
        *reinterpret_cast<float4*>(&ks[h][r][half * 4]) = *reinterpret_cast<const float4*>(base + koff);
        *reinterpret_cast<float4*>(&vs[h][r][half * 4]) = *reinterpret_cast<const float4*>(base + voff);
    }
    __syncthreads();
    if (lane < MH_L) {
        const float* qp = qkv + (row0 + lane) * ld + qoff + h * MH_HD;
        const float4 qa = *reinterpret_cast<const float4*>(qp), qb = *reinterpret_cast<const float4*>(qp + 4);
        float lg[MH_L];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < MH_L; ++j) {
            const float4 ka = *reinterpret_cast<const float4*>(&ks[h][j][0]), kb = *reinterpret_cast<const float4*>(&ks[h][j][4]);
            float s = qa.x * ka.x;
            s = fmaf(qa.y, ka.y, s); s = fmaf(qa.z, ka.z, s); s = fmaf(qa.w, ka.w, s);
            s = fmaf(qb.x, kb.x, s); s = fmaf(qb.y, kb.y, s); s = fmaf(qb.z, kb.z, s); s = fmaf(qb.w, kb.w, s);
            s *= inv_sqrt_dk;
            lg[j] = s;
            mx = fmaxf(mx, s);
        }
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < MH_L; ++j) { lg[j] = __expf(lg[j] - mx); den += lg[j]; }
        const float inv = 1.0f / den;
        float o[MH_HD] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < MH_L; ++j) {
            const float pj = lg[j] * inv;
            const float4 va = *reinterpret_cast<const float4*>(&vs[h][j][0]), vb = *reinterpret_cast<const float4*>(&vs[h][j][4]);
            o[0] = fmaf(pj, va.x, o[0]); o[1] = fmaf(pj, va.y, o[1]); o[2] = fmaf(pj, va.z, o[2]); o[3] = fmaf(pj, va.w, o[3]);
            o[4] = fmaf(pj, vb.x, o[4]); o[5] = fmaf(pj, vb.y, o[5]); o[6] = fmaf(pj, vb.z, o[6]); o[7] = fmaf(pj, vb.w, o[7]);
        }
        float* op = out + (row0 + lane) * ldo + h * MH_HD;
        *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(op + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
}

// The same attention core for the other encoder depths (models_pointcloud.py:34-48: embedding 32 / 128 / 256 -> 8 heads of HD = 4 / 16 / 32,
// direction_backbones.py:151 head_size = embedding_dim // num_heads; logits scaled by 1/sqrt(HD), :125).  Same geometry: wave = head,
// lane = query token, K / V of the 8 heads staged in (dynamic) LDS: 8 x 60 x HD x 2 floats = 15 / 61 / 123 KB.
template <int HD>
__global__ void __launch_bounds__(512) mhsa_attention_hd_kernel(const float* __restrict__ qkv, long ld, int qoff, int koff, int voff,
                                                                float inv_sqrt_dk, float* __restrict__ out, long ldo) {
    extern __shared__ __attribute__((aligned(16))) float mh_lds[];
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
    float* ks = mh_lds + (size_t)h * MH_L * HD;                    // [60][HD] of this head
    float* vs = mh_lds + (size_t)8 * MH_L * HD + (size_t)h * MH_L * HD;
    const size_t row0 = (size_t)blockIdx.x * MH_L;
    constexpr int Q4 = HD / 4;
    for (int e = lane; e < MH_L * Q4; e += 64) {
        const int r = e / Q4, c = (e - r * Q4) * 4;
        const float* base = qkv + (row0 + r) * ld + h * HD + c;
        *reinterpret_cast<float4*>(&ks[r * HD + c]) = *reinterpret_cast<const float4*>(base + koff);
        *reinterpret_cast<float4*>(&vs[r * HD + c]) = *reinterpret_cast<const float4*>(base + voff);
    }
    __syncthreads();
    if (lane < MH_L) {
        const float* qp = qkv + (row0 + lane) * ld + qoff + h * HD;
        float q[HD];
#pragma unroll
        for (int c = 0; c < HD; c += 4) {
            const float4 t = *reinterpret_cast<const float4*>(qp + c);
            q[c] = t.x; q[c + 1] = t.y; q[c + 2] = t.z; q[c + 3] = t.w;
        }
        float lg[MH_L];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < MH_L; ++j) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < HD; c += 4) {
                const float4 kk = *reinterpret_cast<const float4*>(&ks[j * HD + c]);       // broadcast read: every lane reads key j
                s = c == 0 ? q[0] * kk.x : fmaf(q[c], kk.x, s);
                s = fmaf(q[c + 1], kk.y, s); s = fmaf(q[c + 2], kk.z, s); s = fmaf(q[c + 3], kk.w, s);
            }
            s *= inv_sqrt_dk;
            lg[j] = s;
            mx = fmaxf(mx, s);
        }
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < MH_L; ++j) { lg[j] = __expf(lg[j] - mx); den += lg[j]; }
        const float inv = 1.0f / den;
        float o[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) o[c] = 0.f;
#pragma unroll
        for (int j = 0; j < MH_L; ++j) {
            const float pj = lg[j] * inv;
#pragma unroll
            for (int c = 0; c < HD; c += 4) {
                const float4 vv = *reinterpret_cast<const float4*>(&vs[j * HD + c]);
                o[c] = fmaf(pj, vv.x, o[c]); o[c + 1] = fmaf(pj, vv.y, o[c + 1]); o[c + 2] = fmaf(pj, vv.z, o[c + 2]); o[c + 3] = fmaf(pj, vv.w, o[c + 3]);
            }
        }
        float* op = out + (row0 + lane) * ldo + h * HD;
#pragma unroll
        for (int c = 0; c < HD; c += 4) *reinterpret_cast<float4*>(op + c) = make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]);
    }
}

template <int HD>
static int launch_mhsa_hd(long T, const float* qkv, long ld, int qoff, int koff, int voff, float* out, long ldo, hipStream_t st) {
    const size_t lds = (size_t)2 * 8 * MH_L * HD * sizeof(float);
    auto kern = mhsa_attention_hd_kernel<HD>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)T), dim3(512), lds, st, qkv, ld, qoff, koff, voff, (float)(1.0 / sqrt((double)HD)), out, ldo);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

// y[r] = x[r,:] . w + bias   (K % 4 == 0): 16 lanes per row, float4 loads, 4 rows per wave
__global__ void __launch_bounds__(256) rowdot_kernel(long R, int K, const float* __restrict__ x, long ldx,
                                                     const float* __restrict__ w, float bias, float* __restrict__ y) {
    const int sub = threadIdx.x & 15;
    for (long r = (long)blockIdx.x * 16 + (threadIdx.x >> 4); r < R; r += (long)gridDim.x * 16) {
        const float* xr = x + r * ldx;
        float s = 0.f;
        for (int k = sub * 4; k < K; k += 64) {
            const float4 a = *reinterpret_cast<const float4*>(xr + k), b = *reinterpret_cast<const float4*>(w + k);
            s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); s = fmaf(a.w, b.w, s);
        }
        s += __shfl_xor(s, 8, 16); s += __shfl_xor(s, 4, 16); s += __shfl_xor(s, 2, 16); s += __shfl_xor(s, 1, 16);
        if (sub == 0) y[r] = s + bias;
    }
}

// ------------------------------------------------------------------------------------------------
// so3_mean: Ce = sum_a w_a R_a ; R = U diag(1,1,det(U V^T)) V^T with Ce = U S V^T (singular values
// descending, as torch.svd); direction = R[:, 2].  3x3 one-sided Jacobi SVD in fp64, one thread per point.
// ------------------------------------------------------------------------------------------------
__device__ inline void jacobi_svd3(const double A_in[9], double U[9], double S[3], double V[9]) {
    double A[9];
    for (int i = 0; i < 9; ++i) A[i] = A_in[i];
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < 3; ++k) {
                    alpha += A[k * 3 + p] * A[k * 3 + p];
                    beta += A[k * 3 + q] * A[k * 3 + q];
                    gamma += A[k * 3 + p] * A[k * 3 + q];
                }
                if (gamma == 0.0) continue;
                off += fabs(gamma) / (sqrt(alpha * beta) + 1e-300);
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int k = 0; k < 3; ++k) {
                    const double ap = A[k * 3 + p], aq = A[k * 3 + q];
                    A[k * 3 + p] = c * ap - s * aq; A[k * 3 + q] = s * ap + c * aq;
                    const double vp = V[k * 3 + p], vq = V[k * 3 + q];
                    V[k * 3 + p] = c * vp - s * vq; V[k * 3 + q] = s * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    double sv[3];
    for (int j = 0; j < 3; ++j) sv[j] = sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (sv[ord[j]] < sv[ord[j + 1]]) { int t = ord[j]; ord[j] = ord[j + 1]; ord[j + 1] = t; }
    double Vs[9];
    for (int j = 0; j < 3; ++j) {
        const int src = ord[j];
        S[j] = sv[src];
        for (int k = 0; k < 3; ++k) {
            Vs[k * 3 + j] = V[k * 3 + src];
            U[k * 3 + j] = sv[src] > 1e-300 ? A[k * 3 + src] / sv[src] : 0.0;
        }
    }
    for (int i = 0; i < 9; ++i) V[i] = Vs[i];
    // complete U for (near-)zero singular values so it stays orthonormal
    const double tiny = 1e-12 * (S[0] > 0 ? S[0] : 1.0);
    if (S[1] <= tiny) {   // rank <= 1: pick any unit vector orthogonal to u0
        double u0[3] = {U[0], U[3], U[6]};
        if (S[0] <= 1e-300) { u0[0] = 1; u0[1] = 0; u0[2] = 0; U[0] = 1; U[3] = 0; U[6] = 0; }
        int m = fabs(u0[0]) < fabs(u0[1]) ? (fabs(u0[0]) < fabs(u0[2]) ? 0 : 2) : (fabs(u0[1]) < fabs(u0[2]) ? 1 : 2);
        double e[3] = {0, 0, 0}; e[m] = 1.0;
        double u1[3] = {u0[1] * e[2] - u0[2] * e[1], u0[2] * e[0] - u0[0] * e[2], u0[0] * e[1] - u0[1] * e[0]};
        const double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
        U[1] = u1[0] / n1; U[4] = u1[1] / n1; U[7] = u1[2] / n1;
    }
    if (S[2] <= tiny) {   // third column = +-(u0 x u1); sign chosen as det(U) = det(V) so det(U V^T) = +1
        double cx = U[3] * U[7] - U[6] * U[4], cy = U[6] * U[1] - U[0] * U[7], cz = U[0] * U[4] - U[3] * U[1];
        const double detV = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) + V[2] * (V[3] * V[7] - V[4] * V[6]);
        const double sgn = detV < 0 ? -1.0 : 1.0;
        U[2] = sgn * cx; U[5] = sgn * cy; U[8] = sgn * cz;
    }
}

__global__ void __launch_bounds__(256) so3_mean_dir_kernel(long T, int A, const float* __restrict__ w, const float* __restrict__ anchors,
                                                           float* __restrict__ dir, float* __restrict__ Rout, float* __restrict__ sv) {
    __shared__ float sa[60 * 9];
    for (int e = threadIdx.x; e < A * 9; e += 256) sa[e] = anchors[e];
    __syncthreads();
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    double Ce[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float* wr = w + t * A;
    for (int a = 0; a < A; ++a) {
        const double wa = (double)wr[a];
#pragma unroll
        for (int k = 0; k < 9; ++k) Ce[k] += wa * (double)sa[a * 9 + k];
    }
    double U[9], S[3], V[9];
    jacobi_svd3(Ce, U, S, V);
    const double detU = U[0] * (U[4] * U[8] - U[5] * U[7]) - U[1] * (U[3] * U[8] - U[5] * U[6]) + U[2] * (U[3] * U[7] - U[4] * U[6]);
    const double detV = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) + V[2] * (V[3] * V[7] - V[4] * V[6]);
    const double d = detU * detV;            // det(U V^T)
    double R[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i * 3 + j] = U[i * 3] * V[j * 3] + U[i * 3 + 1] * V[j * 3 + 1] + d * U[i * 3 + 2] * V[j * 3 + 2];
    dir[t * 3] = (float)R[2]; dir[t * 3 + 1] = (float)R[5]; dir[t * 3 + 2] = (float)R[8];
    if (Rout) for (int k = 0; k < 9; ++k) Rout[t * 9 + k] = (float)R[k];
    if (sv) { sv[t * 3] = (float)S[0]; sv[t * 3 + 1] = (float)S[1]; sv[t * 3 + 2] = (float)S[2]; }
}

// Backward of so3_mean_dir (src/models/so3conv.py:186-225 followed by R @ [0,0,1], models_pointcloud.py:120-124): dL/dw[t,a] from dL/d dir[t].
// R = polar factor of Ce = U diag(s) V^T, R = U D V^T, D = diag(1, 1, det(U V^T)).  With M = Ce = R S, S = V diag(sigma) V^T,
// sigma = (s0, s1, det * s2):  dR = R X, X skew with  X S + S X = R^T dM - dM^T R  (a 3x3 Sylvester equation, diagonal in V's basis).
// Adjoint:  dL/dM = U D Z V^T,  Z_ij = (A_ij - A_ji) / (sigma_i + sigma_j),  A = D U^T G V,  G = dL/dR (only its third column is
// non-zero here).  fp64 per point like the forward; denominators are floored at 1e-12 * s0 (the projection is not differentiable where
// sigma_i + sigma_j = 0; the reference's torch.svd backward is infinite there).
__global__ void __launch_bounds__(256) so3_mean_dir_backward_kernel(long T, int A, const float* __restrict__ w, const float* __restrict__ anchors,
                                                                    const float* __restrict__ ddir, float* __restrict__ dw) {
    __shared__ float sa[60 * 9];
    for (int e = threadIdx.x; e < A * 9; e += 256) sa[e] = anchors[e];
    __syncthreads();
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    double Ce[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float* wr = w + t * A;
    for (int a = 0; a < A; ++a) {
        const double wa = (double)wr[a];
#pragma unroll
        for (int k = 0; k < 9; ++k) Ce[k] += wa * (double)sa[a * 9 + k];
    }
    double U[9], S[3], V[9];
    jacobi_svd3(Ce, U, S, V);
    const double detU = U[0] * (U[4] * U[8] - U[5] * U[7]) - U[1] * (U[3] * U[8] - U[5] * U[6]) + U[2] * (U[3] * U[7] - U[4] * U[6]);
    const double detV = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) + V[2] * (V[3] * V[7] - V[4] * V[6]);
    const double d = detU * detV;
    const double sig[3] = {S[0], S[1], d * S[2]};
    const double Dg[3] = {1.0, 1.0, d};
    const double g[3] = {(double)ddir[t * 3], (double)ddir[t * 3 + 1], (double)ddir[t * 3 + 2]};     // G[i][2] = g[i], other columns 0
    // A = D U^T G V:  (U^T G)[i][j] = (sum_k U[k][i] g[k]) * [j == 2]  ->  A[i][j] = Dg[i] * ug[i] * V[2][j]
    double ug[3], Am[9], Z[9];
    for (int i = 0; i < 3; ++i) ug[i] = U[i] * g[0] + U[3 + i] * g[1] + U[6 + i] * g[2];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Am[i * 3 + j] = Dg[i] * ug[i] * V[2 * 3 + j];
    const double floor_ = 1e-12 * (S[0] > 0 ? S[0] : 1.0);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double den = sig[i] + sig[j];
            if (fabs(den) < floor_) den = den < 0 ? -floor_ : floor_;
            Z[i * 3 + j] = i == j ? 0.0 : (Am[i * 3 + j] - Am[j * 3 + i]) / den;
        }
    // dCe = U D Z V^T
    double UDZ[9], dCe[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) UDZ[i * 3 + j] = U[i * 3] * Dg[0] * Z[j] + U[i * 3 + 1] * Dg[1] * Z[3 + j] + U[i * 3 + 2] * Dg[2] * Z[6 + j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) dCe[i * 3 + j] = UDZ[i * 3] * V[j * 3] + UDZ[i * 3 + 1] * V[j * 3 + 1] + UDZ[i * 3 + 2] * V[j * 3 + 2];
    float* dwr = dw + t * A;
    for (int a = 0; a < A; ++a) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc += dCe[k] * (double)sa[a * 9 + k];
        dwr[a] = (float)acc;
    }
}

// so3_mean with the reference's general signature (so3conv.py:186-225): Rs (T,A,3,3) per row (rs_stride = A*9) or one shared set
// (rs_stride = 0), weights (T,A) or none (all ones) -> R (T,3,3).  Same fp64 accumulation / Jacobi SVD / det fix as above.
__global__ void __launch_bounds__(256) so3_mean_general_kernel(long T, int A, const float* __restrict__ Rs, long rs_stride,
                                                               const float* __restrict__ w, float* __restrict__ Rout) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    double Ce[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float* rs = Rs + t * rs_stride;
    for (int a = 0; a < A; ++a) {
        const double wa = w ? (double)w[t * A + a] : 1.0;
#pragma unroll
        for (int k = 0; k < 9; ++k) Ce[k] += wa * (double)rs[a * 9 + k];
    }
    double U[9], S[3], V[9];
    jacobi_svd3(Ce, U, S, V);
    const double detU = U[0] * (U[4] * U[8] - U[5] * U[7]) - U[1] * (U[3] * U[8] - U[5] * U[6]) + U[2] * (U[3] * U[7] - U[4] * U[6]);
    const double detV = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) + V[2] * (V[3] * V[7] - V[4] * V[6]);
    const double d = detU * detV;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rout[t * 9 + i * 3 + j] = (float)(U[i * 3] * V[j * 3] + U[i * 3 + 1] * V[j * 3 + 1] + d * U[i * 3 + 2] * V[j * 3 + 2]);
}

extern "C" {

int etch_prop3nn(int B, int N, int S, const float* xyz1, const float* xyz2, int* idx, float* weight, void* stream) {
    if (B <= 0 || N <= 0) return ETCH_OK;
    if (S < 3) return ETCH_EINVAL;
    hipLaunchKernelGGL(prop3nn_kernel, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, N, S, xyz1, xyz2, idx, weight);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_prop_interp_ordered(int B, int N, int S, int A, int C, const float* feats, const int* idx, const float* weight, float* out,
                             float* inv, const int* order, void* stream) {
    if (B <= 0 || N <= 0) return ETCH_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(order ? 8u * (unsigned)((N + 7) / 8) : (unsigned)N, (unsigned)B);
    if (C == 64) hipLaunchKernelGGL(prop_interp_kernel<64>, grid, dim3(256), 0, st, N, S, A, feats, idx, weight, out, inv, order);
    else if (C == 32) hipLaunchKernelGGL(prop_interp_kernel<32>, grid, dim3(256), 0, st, N, S, A, feats, idx, weight, out, inv, order);
    else if (C == 128) hipLaunchKernelGGL(prop_interp_kernel<128>, grid, dim3(256), 0, st, N, S, A, feats, idx, weight, out, inv, order);
    else if (C == 256) hipLaunchKernelGGL(prop_interp_kernel<256>, grid, dim3(256), 0, st, N, S, A, feats, idx, weight, out, inv, order);
    else return ETCH_EUNSUPPORTED;
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_prop_interp(int B, int N, int S, int A, int C, const float* feats, const int* idx, const float* weight, float* out,
                     float* inv, void* stream) {
    return etch_prop_interp_ordered(B, N, S, A, C, feats, idx, weight, out, inv, nullptr, stream);
}

int etch_mhsa_attention(long T, const float* qkv, long ld, int qoff, int koff, int voff, float* out, long ldo, void* stream);

int etch_mhsa_attention_dim(long T, int embedding_dim, const float* qkv, long ld, int qoff, int koff, int voff, float* out, long ldo, void* stream) {
    if (T <= 0) return ETCH_OK;
    if ((ld & 3) || (ldo & 3) || (qoff & 3) || (koff & 3) || (voff & 3)) return ETCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    switch (embedding_dim) {
        case 32: return launch_mhsa_hd<4>(T, qkv, ld, qoff, koff, voff, out, ldo, st);
        case 64: return etch_mhsa_attention(T, qkv, ld, qoff, koff, voff, out, ldo, stream);
        case 128: return launch_mhsa_hd<16>(T, qkv, ld, qoff, koff, voff, out, ldo, st);
        case 256: return launch_mhsa_hd<32>(T, qkv, ld, qoff, koff, voff, out, ldo, st);
    }
    return ETCH_EUNSUPPORTED;
}

int etch_mhsa_attention(long T, const float* qkv, long ld, int qoff, int koff, int voff, float* out, long ldo, void* stream) {
    if (T <= 0) return ETCH_OK;
    if ((ld & 3) || (ldo & 3) || (qoff & 3) || (koff & 3) || (voff & 3)) return ETCH_EINVAL;
    hipLaunchKernelGGL(mhsa_attention_kernel, dim3((unsigned)T), dim3(512), 0, (hipStream_t)stream, qkv, ld, qoff, koff, voff,
                       (float)(1.0 / sqrt(8.0)), out, ldo);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_rowdot(long R, int K, const float* x, long ldx, const float* w, float bias, float* y, void* stream) {
    if (R <= 0) return ETCH_OK;
    if ((K & 3) || (ldx & 3)) return ETCH_EINVAL;
    long blocks = (R + 15) / 16;
    if (blocks > 65535 * 8) blocks = 65535 * 8;
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, R, K, x, ldx, w, bias, y);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_so3_mean_dir(long T, int A, const float* w, const float* anchors, float* dir, float* R, float* sv, void* stream) {
    if (T <= 0) return ETCH_OK;
    if (A != 60) return ETCH_EUNSUPPORTED;
    hipLaunchKernelGGL(so3_mean_dir_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream, T, A, w, anchors, dir, R, sv);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_so3_mean_dir_backward(long T, int A, const float* w, const float* anchors, const float* ddir, float* dw, void* stream) {
    if (T <= 0) return ETCH_OK;
    if (A <= 0 || A > 60 || !w || !anchors || !ddir || !dw) return ETCH_EINVAL;
    hipLaunchKernelGGL(so3_mean_dir_backward_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream, T, A, w, anchors, ddir, dw);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_so3_mean(long T, int A, const float* Rs, long rs_stride, const float* w, float* R, void* stream) {
    if (T <= 0) return ETCH_OK;
    if (A <= 0 || (rs_stride != 0 && rs_stride != (long)A * 9)) return ETCH_EINVAL;
    hipLaunchKernelGGL(so3_mean_general_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream, T, A, Rs, rs_stride, w, R);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

}  // extern "C"
