// Point-Transformer kernels of the confidence / magnitude heads for gfx950 (Appendix B).
// Dense Linear(+BN+ReLU) layers go through etch_linear (gemm.hip); this file holds the neighbourhood parts:
//   etch_pt_attention      PointTransformerLayer.forward after the q/k/v projections
//                          (/root/reference/src/models/pointtransformer_seg.py:28-36, src/models/pointops.py:79-100)
//   etch_pt_group          queryandgroup(use_xyz=True) rows for TransitionDown (:61, pointops.py:90-98)
//   etch_rows_maxpool      MaxPool1d(nsample) (:63)
//   etch_pt_interp_add     linear1(x1) + pointops.interpolation(...) of TransitionUp (:97, pointops.py:164-178)
//   etch_seg_mean / etch_concat_bcast   TransitionUp head branch (:83-93)
//   etch_grouped_dot       confi[2] = Conv1d(128*k, k, 1, groups=k) (:145)
//   etch_softmax_dot       confidence = sum_k softmax(cls)_k * confi_k (:183-189)
// Eval-mode BatchNorm is folded on the host into per-channel (scale, shift).
#include "common.h"

struct PtAttnParams {
    const float* p;       // [n,3]
    const float* xq; const float* xk; const float* xv; long ldq;   // rows of the fused q|k|v GEMM output
    const int* idx;       // [n, ns]
    const float* W0; const float* b0; const float* s_p; const float* t_p;     // linear_p[0] (3x3), BN(3) folded
    const float* W3; const float* b3;                                         // linear_p[3] [c,3], [c]
    const float* s_w0; const float* t_w0;                                     // linear_w[0] BN(c)
    const float* W2T; const float* b2;                                        // linear_w[2] transposed [c][cs], [cs]
    const float* s_w3; const float* t_w3;                                     // linear_w[3] BN(cs)
    const float* W5; const float* b5;                                         // linear_w[5] [cs][cs], [cs]
    const float* s_out; const float* t_out;                                   // optional BN(c)+ReLU on the output (block.bn2)
    float* out; long ldo;
    int n, c, ns;
};

__global__ void __launch_bounds__(128) pt_attention_kernel(PtAttnParams a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int c = a.c, ns = a.ns, cs = c >> 3;
    float* s_pr = sm;                    // [ns][c]
    float* s_w = s_pr + ns * c;          // [ns][c]
    float* s_h = s_w + ns * c;           // [ns][cs]
    float* s_l = s_h + ns * cs;          // [ns][cs]
    float* s_p3 = s_l + ns * cs;         // [ns][4]
    int* s_idx = (int*)(s_p3 + ns * 4);  // [ns]
    const int i = blockIdx.x, tid = threadIdx.x;
    if (tid < ns) {
        const int j = a.idx[(size_t)i * ns + tid];
        s_idx[tid] = j;
        const float rx = a.p[(size_t)j * 3] - a.p[(size_t)i * 3], ry = a.p[(size_t)j * 3 + 1] - a.p[(size_t)i * 3 + 1],
                    rz = a.p[(size_t)j * 3 + 2] - a.p[(size_t)i * 3 + 2];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            float h = a.W0[o * 3] * rx + a.W0[o * 3 + 1] * ry + a.W0[o * 3 + 2] * rz + a.b0[o];
            h = h * a.s_p[o] + a.t_p[o];
            s_p3[tid * 4 + o] = fmaxf(h, 0.f);
        }
    }
    __syncthreads();
    for (int e = tid; e < ns * c; e += 128) {
        const int j = e / c, ch = e - j * c;
        const float pr = a.W3[ch * 3] * s_p3[j * 4] + a.W3[ch * 3 + 1] * s_p3[j * 4 + 1] + a.W3[ch * 3 + 2] * s_p3[j * 4 + 2] + a.b3[ch];
        s_pr[e] = pr;
        float w = a.xk[(size_t)s_idx[j] * a.ldq + ch] - a.xq[(size_t)i * a.ldq + ch] + pr;
        w = w * a.s_w0[ch] + a.t_w0[ch];
        s_w[e] = fmaxf(w, 0.f);
    }
    __syncthreads();
    for (int e = tid; e < ns * cs; e += 128) {
        const int j = e / cs, t = e - j * cs;
        float acc = 0.f;
        const float* wr = s_w + j * c;
        for (int ch = 0; ch < c; ++ch) acc = fmaf(a.W2T[(size_t)ch * cs + t], wr[ch], acc);
        acc += a.b2[t];
        acc = acc * a.s_w3[t] + a.t_w3[t];
        s_h[e] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    for (int e = tid; e < ns * cs; e += 128) {
        const int j = e / cs, t = e - j * cs;
        float acc = 0.f;
        for (int u = 0; u < cs; ++u) acc = fmaf(a.W5[t * cs + u], s_h[j * cs + u], acc);
        s_l[e] = acc + a.b5[t];
    }
    __syncthreads();
    for (int t = tid; t < cs; t += 128) {      // softmax over the ns neighbours, per shared channel t
        float mx = -INFINITY;
        for (int j = 0; j < ns; ++j) mx = fmaxf(mx, s_l[j * cs + t]);
        float den = 0.f;
        for (int j = 0; j < ns; ++j) { const float ev = __expf(s_l[j * cs + t] - mx); s_l[j * cs + t] = ev; den += ev; }
        const float inv = 1.0f / den;
        for (int j = 0; j < ns; ++j) s_l[j * cs + t] *= inv;
    }
    __syncthreads();
    for (int ch = tid; ch < c; ch += 128) {
        const int t = ch % cs;
        float acc = 0.f;
        for (int j = 0; j < ns; ++j) acc += (a.xv[(size_t)s_idx[j] * a.ldq + ch] + s_pr[j * c + ch]) * s_l[j * cs + t];
        if (a.s_out) acc = fmaxf(acc * a.s_out[ch] + a.t_out[ch], 0.f);
        a.out[(size_t)i * a.ldo + ch] = acc;
    }
}


// ------------------------------------------------------------------------------------------------
// GEMM-based split of the vector-attention layer (used for every level): the c -> c/8 -> c/8 weight MLP runs on the
// fp32 matrix cores (etch_linear) over all (point, neighbour) rows; the two kernels below are the streaming ends.
//   prep:       w_in[(i,j), ch] = relu(bn0( x_k[idx[i,j], ch] - x_q[i, ch] + p_r[(i,j), ch] ))
//   aggregate:  out[i, ch] = sum_j (x_v[idx[i,j], ch] + p_r[(i,j), ch]) * softmax_j(logits[(i,j), ch % cs])   (+ BN + ReLU)
// p_r = linear_p(p[idx] - p[i]) is recomputed on the fly in both (3 -> 3 -> c, 12 FMAs per value).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pt_rel_hidden(const PtAttnParams& a, int i, int j, float h[3]) {
    const float rx = a.p[(size_t)j * 3] - a.p[(size_t)i * 3], ry = a.p[(size_t)j * 3 + 1] - a.p[(size_t)i * 3 + 1],
                rz = a.p[(size_t)j * 3 + 2] - a.p[(size_t)i * 3 + 2];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        float v = a.W0[o * 3] * rx + a.W0[o * 3 + 1] * ry + a.W0[o * 3 + 2] * rz + a.b0[o];
        h[o] = fmaxf(v * a.s_p[o] + a.t_p[o], 0.f);
    }
}

__global__ void __launch_bounds__(256) pt_attn_prep_kernel(PtAttnParams a, float* __restrict__ w_in) {
    const int c = a.c, ns = a.ns;
    const int c4 = c >> 2;
    const size_t total = (size_t)a.n * ns * c4;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t row = e / c4;
        const int ch = (int)(e - row * c4) * 4;
        const int i = (int)(row / ns);
        const int j = a.idx[row];
        float h[3];
        pt_rel_hidden(a, i, j, h);
        const float4 k4 = *reinterpret_cast<const float4*>(a.xk + (size_t)j * a.ldq + ch);
        const float4 q4 = *reinterpret_cast<const float4*>(a.xq + (size_t)i * a.ldq + ch);
        const float kv[4] = {k4.x, k4.y, k4.z, k4.w}, qv[4] = {q4.x, q4.y, q4.z, q4.w};
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cc = ch + u;
            const float pr = a.W3[cc * 3] * h[0] + a.W3[cc * 3 + 1] * h[1] + a.W3[cc * 3 + 2] * h[2] + a.b3[cc];
            const float w = (kv[u] - qv[u] + pr) * a.s_w0[cc] + a.t_w0[cc];
            o[u] = fmaxf(w, 0.f);
        }
        *reinterpret_cast<float4*>(w_in + row * c + ch) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// one wave per point; lane handles channels ch = lane, lane + 64, ...
__global__ void __launch_bounds__(256) pt_attn_aggregate_kernel(PtAttnParams a, const float* __restrict__ logits) {
    const int c = a.c, ns = a.ns, cs = c >> 3;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= a.n) return;
    const int* nb = a.idx + (size_t)i * ns;
    for (int ch = lane; ch < c; ch += 64) {
        const int t = ch % cs;
        float mx = -INFINITY;
        for (int j = 0; j < ns; ++j) mx = fmaxf(mx, logits[((size_t)i * ns + j) * cs + t]);
        float den = 0.f, acc = 0.f;
        const float w30 = a.W3[ch * 3], w31 = a.W3[ch * 3 + 1], w32 = a.W3[ch * 3 + 2], b3 = a.b3[ch];
        for (int j = 0; j < ns; ++j) {
            const int q = nb[j];
            float h[3];
            pt_rel_hidden(a, i, q, h);
            const float pr = w30 * h[0] + w31 * h[1] + w32 * h[2] + b3;
            const float ev = __expf(logits[((size_t)i * ns + j) * cs + t] - mx);
            den += ev;
            acc += (a.xv[(size_t)q * a.ldq + ch] + pr) * ev;
        }
        float v = acc / den;
        if (a.s_out) v = fmaxf(v * a.s_out[ch] + a.t_out[ch], 0.f);
        a.out[(size_t)i * a.ldo + ch] = v;
    }
}

// rows[(i*ns + j)] = [ p[idx[i,j]] - new_p[i]  (3) | x[idx[i,j]] (c) | 0 ... ],   row stride ld >= 3 + c (padding columns are zeroed,
// so a consumer may treat the row as ld wide: ld % 4 == 0 keeps the following GEMM on its 16-byte load path)
__global__ void __launch_bounds__(256) pt_group_kernel(int m, int ns, int c, const float* __restrict__ p, const float* __restrict__ new_p,
                                                       const float* __restrict__ x, long ldx, const int* __restrict__ idx,
                                                       float* __restrict__ out, int ld) {
    const size_t total = (size_t)m * ns * ld;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t row = e / ld;
        const int col = (int)(e - row * ld);
        const int i = (int)(row / ns);
        const int j = idx[row];
        out[e] = col < 3 ? p[(size_t)j * 3 + col] - new_p[(size_t)i * 3 + col] : (col < 3 + c ? x[(size_t)j * ldx + col - 3] : 0.f);
    }
}

// out[i, ch] = max_j x[i*ns + j, ch]
__global__ void __launch_bounds__(256) rows_maxpool_kernel(int m, int ns, int c, const float* __restrict__ x, float* __restrict__ out) {
    const size_t total = (size_t)m * c;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t i = e / c;
        const int ch = (int)(e - i * c);
        float v = -INFINITY;
        for (int j = 0; j < ns; ++j) v = fmaxf(v, x[(i * ns + j) * c + ch]);
        out[e] = v;
    }
}

// out[i,ch] = a[i,ch] + ((f[idx0]*w0) + f[idx1]*w1) + f[idx2]*w2,  w = (1/(d+1e-8)) / sum   (d = NON-squared distance)
__global__ void __launch_bounds__(256) pt_interp_add_kernel(int n, int c, const float* __restrict__ a, const float* __restrict__ f,
                                                            const int* __restrict__ idx, const float* __restrict__ dist,
                                                            float* __restrict__ out) {
    const size_t total = (size_t)n * c;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t i = e / c;
        const int ch = (int)(e - i * c);
        const float r0 = 1.0f / (dist[i * 3] + 1e-8f), r1 = 1.0f / (dist[i * 3 + 1] + 1e-8f), r2 = 1.0f / (dist[i * 3 + 2] + 1e-8f);
        const float nrm = (r0 + r1) + r2;
        float v = f[(size_t)idx[i * 3] * c + ch] * (r0 / nrm);
        v += f[(size_t)idx[i * 3 + 1] * c + ch] * (r1 / nrm);
        v += f[(size_t)idx[i * 3 + 2] * c + ch] * (r2 / nrm);
        out[e] = a[e] + v;
    }
}

// mean[b, ch] = sum_{rows of segment b} x[row, ch] / cnt      (one workgroup per segment, thread per channel)
__global__ void __launch_bounds__(256) seg_mean_kernel(int c, const float* __restrict__ x, const int* __restrict__ offset,
                                                       float* __restrict__ mean) {
    const int b = blockIdx.x;
    const int s = b == 0 ? 0 : offset[b - 1], e = offset[b];
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        float acc = 0.f;
        for (int r = s; r < e; ++r) acc += x[(size_t)r * c + ch];
        mean[(size_t)b * c + ch] = acc / (float)(e - s);
    }
}

// out[row] = [ x[row] (c) | g[seg(row)] (c) ]
__global__ void __launch_bounds__(256) concat_bcast_kernel(int n, int c, int nseg, const float* __restrict__ x, const float* __restrict__ g,
                                                           const int* __restrict__ offset, float* __restrict__ out) {
    const size_t total = (size_t)n * 2 * c;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t row = e / (2 * c);
        const int col = (int)(e - row * 2 * c);
        if (col < c) out[e] = x[row * c + col];
        else {
            int b = 0;
            while (b < nseg - 1 && (int)row >= offset[b]) ++b;
            out[e] = g[(size_t)b * c + col - c];
        }
    }
}

// out[r, g] = sum_j h[r, g*J + j] * w[g, j] + b[g]       (16 lanes per (r,g), J % 64 == 0)
__global__ void __launch_bounds__(256) grouped_dot_kernel(long R, int G, int J, const float* __restrict__ h, long ldh,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ out, long ldo) {
    const int sub = threadIdx.x & 15;
    const long total = R * G;
    for (long e = (long)blockIdx.x * 16 + (threadIdx.x >> 4); e < total; e += (long)gridDim.x * 16) {
        const long r = e / G;
        const int g = (int)(e - r * G);
        const float* hr = h + r * ldh + (long)g * J;
        const float* wr = w + (long)g * J;
        float s = 0.f;
        for (int k = sub * 4; k < J; k += 64) {
            const float4 x = *reinterpret_cast<const float4*>(hr + k), y = *reinterpret_cast<const float4*>(wr + k);
            s = fmaf(x.x, y.x, s); s = fmaf(x.y, y.y, s); s = fmaf(x.z, y.z, s); s = fmaf(x.w, y.w, s);
        }
        s += __shfl_xor(s, 8, 16); s += __shfl_xor(s, 4, 16); s += __shfl_xor(s, 2, 16); s += __shfl_xor(s, 1, 16);
        if (sub == 0) out[r * ldo + g] = s + bias[g];
    }
}

// conf[r] = sum_g softmax(logits[r, :])_g * v[r, g]       (one wave per row)
__global__ void __launch_bounds__(256) softmax_dot_kernel(long R, int G, const float* __restrict__ logits, const float* __restrict__ v,
                                                          float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < R; r += (long)gridDim.x * 4) {
        float mx = -INFINITY;
        for (int g = lane; g < G; g += 64) mx = fmaxf(mx, logits[r * G + g]);
        mx = etch_wave_max_f32(mx);
        float den = 0.f, num = 0.f;
        for (int g = lane; g < G; g += 64) {
            const float ev = __expf(logits[r * G + g] - mx);
            den += ev; num += ev * v[r * G + g];
        }
        den = etch_wave_sum_f32(den); num = etch_wave_sum_f32(num);
        if (lane == 0) out[r] = num / den;
    }
}

// out[i, :c] = x[idx[i], :c]
__global__ void __launch_bounds__(256) gather_rows_kernel(int m, int c, const float* __restrict__ x, long ldx, const int* __restrict__ idx,
                                                          float* __restrict__ out) {
    const size_t total = (size_t)m * c;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t i = e / c;
        out[e] = x[(size_t)idx[i] * ldx + (e - i * c)];
    }
}

static inline unsigned grid_for(size_t total, int per_block) {
    size_t b = (total + per_block - 1) / per_block;
    if (b > 65535u * 16u) b = 65535u * 16u;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// ------------------------------------------------------------------------------------------------
// Backward of the vector-attention core (the function pt_attention_kernel computes, without the optional output BN; eval-mode BatchNorm
// = the folded per-channel (scale, shift), treated as constants): autograd through pointtransformer_seg.py:28-36 as train.py:77-101
// obtains it.  One workgroup per point i recomputes the forward in LDS and writes, per (point, neighbour) row e = i ns + j:
//   GV[e][c]  = dout[c] w_j[c % cs]                       gradient of (x_v[idx] + p_r)         -> scattered to dx_v by idx
//   GU[e][c]  = gradient of u = x_k[idx] - x_q + p_r      -> scattered to dx_k by idx;  dx_q[i] = - sum_j GU
//   A[e][c], DZ[e][cs]     input / output-gradient rows of linear_w[2]:  dW2 = DZ^T A,  db2 = colsum DZ
//   G[e][cs], DL[e][cs]    input / output-gradient rows of linear_w[5]:  dW5 = DL^T G,  db5 = colsum DL
//   H4, DH4, R4 [e][4]     linear_p: hidden (post ReLU), gradient at linear_p[0]'s output, relative position (3 used + 0):
//                          dW3 = (GV + GU)^T H4, db3 = colsum(GV + GU);  dW0 = DH4^T R4, db0 = colsum DH4
// The sums over rows (matrix-core GEMMs, column sums, per-source-point segment sums) run in fixed orders: reproducible bit for bit.
// ------------------------------------------------------------------------------------------------
struct PtAttnBwdOut {
    float* GV; float* GU; float* A; float* DZ; float* G; float* DL; float* H4; float* DH4; float* R4; float* dxq;
};

__global__ void __launch_bounds__(128) pt_attention_backward_kernel(PtAttnParams a, const float* __restrict__ dout, long lddo, PtAttnBwdOut o) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int c = a.c, ns = a.ns, cs = c >> 3;
    float* s_pr = sm;                    // [ns][c]   p_r, later d p_r
    float* s_a = s_pr + ns * c;          // [ns][c]   a = relu(bn0(u))
    float* s_gv = s_a + ns * c;          // [ns][c]   GV
    float* s_g = s_gv + ns * c;          // [ns][cs]  g = relu(bn3(z))
    float* s_w = s_g + ns * cs;          // [ns][cs]  logits -> softmax weights
    float* s_d = s_w + ns * cs;          // [ns][cs]  dw -> dl -> dz
    float* s_h = s_d + ns * cs;          // [ns][4]   h (post ReLU)
    float* s_r = s_h + ns * 4;           // [ns][4]   relative position
    int* s_idx = (int*)(s_r + ns * 4);   // [ns]
    const int i = blockIdx.x, tid = threadIdx.x;
    const size_t e0 = (size_t)i * ns;
    if (tid < ns) {
        const int j = a.idx[e0 + tid];
        s_idx[tid] = j;
        const float r[3] = {a.p[(size_t)j * 3] - a.p[(size_t)i * 3], a.p[(size_t)j * 3 + 1] - a.p[(size_t)i * 3 + 1],
                            a.p[(size_t)j * 3 + 2] - a.p[(size_t)i * 3 + 2]};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float h = a.W0[q * 3] * r[0] + a.W0[q * 3 + 1] * r[1] + a.W0[q * 3 + 2] * r[2] + a.b0[q];
            h = h * a.s_p[q] + a.t_p[q];
            s_h[tid * 4 + q] = fmaxf(h, 0.f);
            s_r[tid * 4 + q] = r[q];
        }
        s_h[tid * 4 + 3] = 0.f; s_r[tid * 4 + 3] = 0.f;
    }
    __syncthreads();
    // ---- forward (as pt_attention_kernel)
    for (int e = tid; e < ns * c; e += 128) {
        const int j = e / c, ch = e - j * c;
        const float pr = a.W3[ch * 3] * s_h[j * 4] + a.W3[ch * 3 + 1] * s_h[j * 4 + 1] + a.W3[ch * 3 + 2] * s_h[j * 4 + 2] + a.b3[ch];
        s_pr[e] = pr;
        float u = a.xk[(size_t)s_idx[j] * a.ldq + ch] - a.xq[(size_t)i * a.ldq + ch] + pr;
        u = u * a.s_w0[ch] + a.t_w0[ch];
        s_a[e] = fmaxf(u, 0.f);
    }
    __syncthreads();
    for (int e = tid; e < ns * cs; e += 128) {
        const int j = e / cs, t = e - j * cs;
        float acc = 0.f;
        const float* ar = s_a + j * c;
        for (int ch = 0; ch < c; ++ch) acc = fmaf(a.W2T[(size_t)ch * cs + t], ar[ch], acc);
        acc += a.b2[t];
        acc = acc * a.s_w3[t] + a.t_w3[t];
        s_g[e] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    for (int e = tid; e < ns * cs; e += 128) {
        const int j = e / cs, t = e - j * cs;
        float acc = 0.f;
        for (int u = 0; u < cs; ++u) acc = fmaf(a.W5[t * cs + u], s_g[j * cs + u], acc);
        s_w[e] = acc + a.b5[t];
    }
    __syncthreads();
    for (int t = tid; t < cs; t += 128) {
        float mx = -INFINITY;
        for (int j = 0; j < ns; ++j) mx = fmaxf(mx, s_w[j * cs + t]);
        float den = 0.f;
        for (int j = 0; j < ns; ++j) { const float ev = __expf(s_w[j * cs + t] - mx); s_w[j * cs + t] = ev; den += ev; }
        const float inv = 1.0f / den;
        for (int j = 0; j < ns; ++j) s_w[j * cs + t] *= inv;
    }
    __syncthreads();
    // ---- backward
    // GV and dw[j][t] = sum over the 8 shared planes of dout * (x_v + p_r)
    for (int e = tid; e < ns * c; e += 128) {
        const int j = e / c, ch = e - j * c;
        const float gv = dout[(size_t)i * lddo + ch] * s_w[j * cs + ch % cs];
        s_gv[e] = gv;
        o.GV[(e0 + j) * c + ch] = gv;
    }
    for (int e = tid; e < ns * cs; e += 128) {
        const int j = e / cs, t = e - j * cs;
        float acc = 0.f;
        for (int sh = 0; sh < 8; ++sh) {
            const int ch = sh * cs + t;
            acc += dout[(size_t)i * lddo + ch] * (a.xv[(size_t)s_idx[j] * a.ldq + ch] + s_pr[j * c + ch]);
        }
        s_d[e] = acc;
    }
    __syncthreads();
    // softmax backward over the neighbours: dl = w (dw - sum_j dw w)
    for (int t = tid; t < cs; t += 128) {
        float D = 0.f;
        for (int j = 0; j < ns; ++j) D += s_d[j * cs + t] * s_w[j * cs + t];
        for (int j = 0; j < ns; ++j) {
            const float dl = s_w[j * cs + t] * (s_d[j * cs + t] - D);
            s_w[j * cs + t] = dl;                                       // s_w now holds dl
            o.DL[(e0 + j) * cs + t] = dl;
            o.G[(e0 + j) * cs + t] = s_g[j * cs + t];
        }
    }
    __syncthreads();
    // dz = (W5^T dl) * s_w3 * [g > 0]
    for (int e = tid; e < ns * cs; e += 128) {
        const int j = e / cs, u = e - j * cs;
        float acc = 0.f;
        for (int t = 0; t < cs; ++t) acc = fmaf(a.W5[t * cs + u], s_w[j * cs + t], acc);
        const float dz = s_g[e] > 0.f ? acc * a.s_w3[u] : 0.f;
        s_d[e] = dz;
        o.DZ[(e0 + j) * cs + u] = dz;
    }
    __syncthreads();
    // du = (W2^T dz) * s_w0 * [a > 0];  d p_r = GV + du
    for (int e = tid; e < ns * c; e += 128) {
        const int j = e / c, ch = e - j * c;
        float acc = 0.f;
        for (int u = 0; u < cs; ++u) acc = fmaf(a.W2T[(size_t)ch * cs + u], s_d[j * cs + u], acc);
        const float av = s_a[e];
        const float du = av > 0.f ? acc * a.s_w0[ch] : 0.f;
        o.GU[(e0 + j) * c + ch] = du;
        o.A[(e0 + j) * c + ch] = av;
        s_a[e] = du;                                                    // s_a now holds du
        s_pr[e] = s_gv[e] + du;                                         // s_pr now holds d p_r
    }
    __syncthreads();
    for (int ch = tid; ch < c; ch += 128) {
        float acc = 0.f;
        for (int j = 0; j < ns; ++j) acc += s_a[j * c + ch];
        o.dxq[(size_t)i * c + ch] = -acc;
    }
    // dh = W3^T d p_r;  gradient at linear_p[0]'s output = dh * s_p * [h > 0]
    for (int e = tid; e < ns * 4; e += 128) {
        const int j = e >> 2, q = e & 3;
        float v = 0.f;
        if (q < 3) {
            float acc = 0.f;
            for (int ch = 0; ch < c; ++ch) acc = fmaf(a.W3[ch * 3 + q], s_pr[j * c + ch], acc);
            v = s_h[e] > 0.f ? acc * a.s_p[q] : 0.f;
        }
        o.DH4[(e0 + j) * 4 + q] = v;
        o.H4[(e0 + j) * 4 + q] = s_h[e];
        o.R4[(e0 + j) * 4 + q] = s_r[e];
    }
}

// dst[q][:] = sum over the rows perm[seg[q] .. seg[q+1]) of src, in that order (a stable sort of the index list by source point):
// the deterministic form of scatter-add
__global__ void __launch_bounds__(256) segment_sum_rows_kernel(long nseg, int C, const float* __restrict__ src, const long long* __restrict__ perm,
                                                               const long long* __restrict__ seg, float* __restrict__ dst) {
    const int c4 = C >> 2;
    const long total = nseg * c4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long q = e / c4;
        const int ch = (int)(e - q * c4) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (long long k = seg[q]; k < seg[q + 1]; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)perm[k] * C + ch);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(dst + (size_t)q * C + ch) = acc;
    }
}

extern "C" {

// params: 16 device pointers in the order of PtAttnParams (W0,b0,s_p,t_p,W3,b3,s_w0,t_w0,W2T,b2,s_w3,t_w3,W5,b5,s_out,t_out)
int etch_pt_attention(int n, int c, int ns, const float* p, const float* xq, const float* xk, const float* xv, long ldq,
                      const int* idx, const float* const* params, float* out, long ldo, void* stream) {
    if (n <= 0) return ETCH_OK;
    if (c <= 0 || (c & 7) || ns <= 0 || ns > 128) return ETCH_EINVAL;
    PtAttnParams a;
    a.p = p; a.xq = xq; a.xk = xk; a.xv = xv; a.ldq = ldq; a.idx = idx;
    a.W0 = params[0]; a.b0 = params[1]; a.s_p = params[2]; a.t_p = params[3]; a.W3 = params[4]; a.b3 = params[5];
    a.s_w0 = params[6]; a.t_w0 = params[7]; a.W2T = params[8]; a.b2 = params[9]; a.s_w3 = params[10]; a.t_w3 = params[11];
    a.W5 = params[12]; a.b5 = params[13]; a.s_out = params[14]; a.t_out = params[15];
    a.out = out; a.ldo = ldo; a.n = n; a.c = c; a.ns = ns;
    const int cs = c / 8;
    const size_t lds = (size_t)(2 * ns * c + 2 * ns * cs + ns * 4 + ns) * sizeof(float);
    if (lds > 160 * 1024) return ETCH_EUNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)pt_attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(pt_attention_kernel, dim3(n), dim3(128), lds, (hipStream_t)stream, a);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

static void fill_pt_params(PtAttnParams& a, int n, int c, int ns, const float* p, const float* xq, const float* xk, const float* xv, long ldq,
                           const int* idx, const float* const* params, float* out, long ldo) {
    a.p = p; a.xq = xq; a.xk = xk; a.xv = xv; a.ldq = ldq; a.idx = idx;
    a.W0 = params[0]; a.b0 = params[1]; a.s_p = params[2]; a.t_p = params[3]; a.W3 = params[4]; a.b3 = params[5];
    a.s_w0 = params[6]; a.t_w0 = params[7]; a.W2T = params[8]; a.b2 = params[9]; a.s_w3 = params[10]; a.t_w3 = params[11];
    a.W5 = params[12]; a.b5 = params[13]; a.s_out = params[14]; a.t_out = params[15];
    a.out = out; a.ldo = ldo; a.n = n; a.c = c; a.ns = ns;
}

int etch_pt_attn_prep(int n, int c, int ns, const float* p, const float* xq, const float* xk, long ldq, const int* idx,
                      const float* const* params, float* w_in, void* stream) {
    if (n <= 0) return ETCH_OK;
    if (c <= 0 || (c & 7) || (ldq & 3)) return ETCH_EINVAL;
    PtAttnParams a;
    fill_pt_params(a, n, c, ns, p, xq, xk, nullptr, ldq, idx, params, nullptr, 0);
    hipLaunchKernelGGL(pt_attn_prep_kernel, dim3(grid_for((size_t)n * ns * (c / 4), 256)), dim3(256), 0, (hipStream_t)stream, a, w_in);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_pt_attn_aggregate(int n, int c, int ns, const float* p, const float* xv, long ldq, const int* idx, const float* logits,
                           const float* const* params, float* out, long ldo, void* stream) {
    if (n <= 0) return ETCH_OK;
    if (c <= 0 || (c & 7)) return ETCH_EINVAL;
    PtAttnParams a;
    fill_pt_params(a, n, c, ns, p, nullptr, nullptr, xv, ldq, idx, params, out, ldo);
    hipLaunchKernelGGL(pt_attn_aggregate_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, logits);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_pt_group(int m, int ns, int c, const float* p, const float* new_p, const float* x, long ldx, const int* idx, float* out,
                  long ldo, void* stream) {
    if (m <= 0) return ETCH_OK;
    if (ldo < 3 + c) return ETCH_EINVAL;
    hipLaunchKernelGGL(pt_group_kernel, dim3(grid_for((size_t)m * ns * ldo, 256)), dim3(256), 0, (hipStream_t)stream, m, ns, c, p,
                       new_p, x, ldx, idx, out, (int)ldo);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_gather_rows(int m, int c, const float* x, long ldx, const int* idx, float* out, void* stream) {
    if (m <= 0) return ETCH_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((size_t)m * c, 256)), dim3(256), 0, (hipStream_t)stream, m, c, x, ldx, idx, out);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_rows_maxpool(int m, int ns, int c, const float* x, float* out, void* stream) {
    if (m <= 0) return ETCH_OK;
    hipLaunchKernelGGL(rows_maxpool_kernel, dim3(grid_for((size_t)m * c, 256)), dim3(256), 0, (hipStream_t)stream, m, ns, c, x, out);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_pt_interp_add(int n, int c, const float* a, const float* f, const int* idx, const float* dist, float* out, void* stream) {
    if (n <= 0) return ETCH_OK;
    hipLaunchKernelGGL(pt_interp_add_kernel, dim3(grid_for((size_t)n * c, 256)), dim3(256), 0, (hipStream_t)stream, n, c, a, f, idx, dist, out);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_seg_mean(int nseg, int c, const float* x, const int* offset, float* mean, void* stream) {
    if (nseg <= 0) return ETCH_OK;
    hipLaunchKernelGGL(seg_mean_kernel, dim3(nseg), dim3(256), 0, (hipStream_t)stream, c, x, offset, mean);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_concat_bcast(int n, int c, int nseg, const float* x, const float* g, const int* offset, float* out, void* stream) {
    if (n <= 0) return ETCH_OK;
    hipLaunchKernelGGL(concat_bcast_kernel, dim3(grid_for((size_t)n * 2 * c, 256)), dim3(256), 0, (hipStream_t)stream, n, c, nseg, x, g,
                       offset, out);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_grouped_dot(long R, int G, int J, const float* h, long ldh, const float* w, const float* bias, float* out, long ldo,
                     void* stream) {
    if (R <= 0) return ETCH_OK;
    if ((J & 3) || (ldh & 3)) return ETCH_EINVAL;
    hipLaunchKernelGGL(grouped_dot_kernel, dim3(grid_for((size_t)R * G, 16)), dim3(256), 0, (hipStream_t)stream, R, G, J, h, ldh, w, bias,
                       out, ldo);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_softmax_dot(long R, int G, const float* logits, const float* v, float* out, void* stream) {
    if (R <= 0) return ETCH_OK;
    hipLaunchKernelGGL(softmax_dot_kernel, dim3(grid_for((size_t)R, 4)), dim3(256), 0, (hipStream_t)stream, R, G, logits, v, out);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_pt_attention_backward(int n, int c, int ns, const float* p, const float* xq, const float* xk, const float* xv, long ldq, const int* idx,
                               const float* const* params, const float* dout, long lddo, float* const* outs, void* stream) {
    if (n <= 0) return ETCH_OK;
    if (c <= 0 || (c & 7) || ns <= 0 || ns > 128 || !params || !outs || !dout) return ETCH_EINVAL;
    PtAttnParams a;
    a.p = p; a.xq = xq; a.xk = xk; a.xv = xv; a.ldq = ldq; a.idx = idx;
    a.W0 = params[0]; a.b0 = params[1]; a.s_p = params[2]; a.t_p = params[3]; a.W3 = params[4]; a.b3 = params[5];
    a.s_w0 = params[6]; a.t_w0 = params[7]; a.W2T = params[8]; a.b2 = params[9]; a.s_w3 = params[10]; a.t_w3 = params[11];
    a.W5 = params[12]; a.b5 = params[13]; a.s_out = nullptr; a.t_out = nullptr;
    a.out = nullptr; a.ldo = 0; a.n = n; a.c = c; a.ns = ns;
    PtAttnBwdOut o{outs[0], outs[1], outs[2], outs[3], outs[4], outs[5], outs[6], outs[7], outs[8], outs[9]};
    for (int k = 0; k < 10; ++k) if (!outs[k]) return ETCH_EINVAL;
    const int cs = c / 8;
    const size_t lds = (size_t)(3 * ns * c + 3 * ns * cs + ns * 8 + ns) * sizeof(float);
    if (lds > 160 * 1024) return ETCH_EUNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)pt_attention_backward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(pt_attention_backward_kernel, dim3(n), dim3(128), lds, (hipStream_t)stream, a, dout, lddo, o);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_segment_sum_rows(long nseg, int C, const float* src, const long long* perm, const long long* seg, float* dst, void* stream) {
    if (nseg <= 0) return ETCH_OK;
    if (C <= 0 || (C & 3) || !src || !perm || !seg || !dst) return ETCH_EINVAL;
    long blocks = (nseg * (C >> 2) + 255) / 256;
    if (blocks > 65535L * 16) blocks = 65535L * 16;
    hipLaunchKernelGGL(segment_sum_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, nseg, C, src, perm, seg, dst);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// PointTransformerLayer attention core (pointtransformer_seg.py:28-36) as ONE kernel on the matrix cores:
//   w_in[j]  = relu(bn(x_k[idx_j] - x_q[i] + p_r[j]))                 (ns x c, built in registers from the gathered rows)
//   hid^T    = relu(bn(W2 . w_in^T + b2))                             (cs x ns, cs = c/8)   MFMA, A = W2 rows, B = w_in rows
//   logit^T  = W5 . hid^T + b5                                        (cs x ns)             MFMA, B = the accumulators of hid^T
//   out[i,ch] = sum_j softmax_j(logit[j, ch % cs]) * (x_v[idx_j, ch] + p_r[j, ch])   (+ optional bn + relu)
// One wave = 16 (point, neighbour) rows = 16/ns points; lane (fg, fr): row fr, channel quad 4fg.  Computing the first product
// transposed makes its accumulators (row = hidden unit 4fg + r, column = neighbour fr) the B operand of the second one, and
// leaves every logit of a point's neighbours in one 16-lane DPP row: softmax and the neighbour sum are DPP row reductions.
// The n*ns x c / n*ns x cs intermediates of the split path (prep -> 2 GEMMs -> aggregate) never exist.
// ------------------------------------------------------------------------------------------------
typedef float pf32x4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float pt_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int NS>
__device__ __forceinline__ float pt_group_sum(float v) {        // sum over the NS (8 or 16) lanes of a neighbour group
    v += pt_dpp<0xB1>(v);          // quad_perm [1,0,3,2]
    v += pt_dpp<0x4E>(v);          // quad_perm [2,3,0,1]
    v += pt_dpp<0x141>(v);         // row_half_mirror
    if (NS == 16) v += pt_dpp<0x140>(v);   // row_mirror
    return v;
}
template <int NS>
__device__ __forceinline__ float pt_group_max(float v) {
    v = fmaxf(v, pt_dpp<0xB1>(v));
    v = fmaxf(v, pt_dpp<0x4E>(v));
    v = fmaxf(v, pt_dpp<0x141>(v));
    if (NS == 16) v = fmaxf(v, pt_dpp<0x140>(v));
    return v;
}

#ifndef PT_MFMA_WGS
#define PT_MFMA_WGS 4
#endif
// Per-channel constants (linear_p[3], its bias, the BN of linear_w[0], the output BN), linear_w's small tensors and -- up to c = 256 --
// the c/8 x c matrix W2 are staged in LDS once per (persistent) workgroup: read from global per 16-channel step they were 8 of the 10
// vector-memory instructions of the inner loop and the kernel ran at 60 % of the CU's texture-address rate, not on its MFMAs.
template <int C, int NS>
__global__ void __launch_bounds__(256, PT_MFMA_WGS) pt_attention_mfma_kernel(PtAttnParams a, const float* __restrict__ W2, long ntiles) {
    constexpr int CS = C / 8;                      // hidden width of linear_w
    constexpr int MT = CS <= 16 ? 1 : CS / 16;     // 16-row tiles of the transposed products
    constexpr int KT = C / 16;
    constexpr int PPW = 16 / NS;                   // points per wave
    constexpr bool W2_LDS = C <= 256;
    constexpr int LDW = C + 4;                     // W2 row stride in LDS: (C + 4) / 4 odd -> conflict-free ds_read_b128 over the 16 rows
    extern __shared__ __attribute__((aligned(16))) float cst[];
    float* W3s = cst;                              // [C][3]
    float* b3s = W3s + 3 * C;                      // [C]
    float* scs = b3s + C;                          // [C]  s_w0
    float* shs = scs + C;                          // [C]  t_w0
    float* sos = shs + C;                          // [C]  s_out
    float* tos = sos + C;                          // [C]  t_out
    float* b2s = tos + C;                          // [CS]
    float* s3s = b2s + CS;                         // [CS]
    float* t3s = s3s + CS;                         // [CS]
    float* b5s = t3s + CS;                         // [CS]
    float* W5s = b5s + CS;                         // [CS][CS]
    float* W2s = W5s + CS * CS;                    // [CS][LDW]  (W2_LDS only)
    const int tid = threadIdx.x;
    for (int e = tid; e < 3 * C; e += 256) W3s[e] = a.W3[e];
    for (int e = tid; e < C; e += 256) {
        b3s[e] = a.b3[e]; scs[e] = a.s_w0[e]; shs[e] = a.t_w0[e];
        sos[e] = a.s_out ? a.s_out[e] : 1.f; tos[e] = a.s_out ? a.t_out[e] : 0.f;
    }
    for (int e = tid; e < CS; e += 256) { b2s[e] = a.b2[e]; s3s[e] = a.s_w3[e]; t3s[e] = a.t_w3[e]; b5s[e] = a.b5[e]; }
    for (int e = tid; e < CS * CS; e += 256) W5s[e] = a.W5[e];
    if (W2_LDS)
        for (int e = tid; e < CS * C / 4; e += 256) {
            const int row = e / (C / 4), c4 = e - row * (C / 4);
            *reinterpret_cast<float4*>(&W2s[row * LDW + c4 * 4]) = *reinterpret_cast<const float4*>(W2 + (size_t)row * C + c4 * 4);
        }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int pw = fr / NS, jn = fr % NS;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    asm volatile("" ::: "memory");                 // the staged constants stay in LDS: hoisted out of this loop they took 150+ registers
    const long base = (tile * 4 + wave) * PPW;
    const bool valid = base + pw < a.n;
    const int i = valid ? (int)(base + pw) : a.n - 1;
    const int j = a.idx[(size_t)i * NS + jn];
    float h[3];
    pt_rel_hidden(a, i, j, h);
    const float* kr = a.xk + (size_t)j * a.ldq;
    const float* qr = a.xq + (size_t)i * a.ldq;
    const float* vr = a.xv + (size_t)j * a.ldq;

    // ---- hid^T = W2 . w_in^T
    pf32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (pf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int t = 0; t < KT; ++t) {
        const int ch0 = t * 16 + fg * 4;
        const float4 k4 = *reinterpret_cast<const float4*>(kr + ch0), q4 = *reinterpret_cast<const float4*>(qr + ch0);
        const float4 wa = *reinterpret_cast<const float4*>(W3s + ch0 * 3), wb = *reinterpret_cast<const float4*>(W3s + ch0 * 3 + 4),
                     wc = *reinterpret_cast<const float4*>(W3s + ch0 * 3 + 8);
        const float4 b3 = *reinterpret_cast<const float4*>(b3s + ch0), sc = *reinterpret_cast<const float4*>(scs + ch0),
                     sh = *reinterpret_cast<const float4*>(shs + ch0);
        float w[4];
        w[0] = fmaxf((k4.x - q4.x + (wa.x * h[0] + wa.y * h[1] + wa.z * h[2] + b3.x)) * sc.x + sh.x, 0.f);
        w[1] = fmaxf((k4.y - q4.y + (wa.w * h[0] + wb.x * h[1] + wb.y * h[2] + b3.y)) * sc.y + sh.y, 0.f);
        w[2] = fmaxf((k4.z - q4.z + (wb.z * h[0] + wb.w * h[1] + wc.x * h[2] + b3.z)) * sc.z + sh.z, 0.f);
        w[3] = fmaxf((k4.w - q4.w + (wc.y * h[0] + wc.z * h[1] + wc.w * h[2] + b3.w)) * sc.w + sh.w, 0.f);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = mt * 16 + fr;                       // hidden unit
            float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < CS) f = W2_LDS ? *reinterpret_cast<const float4*>(W2s + row * LDW + ch0) : *reinterpret_cast<const float4*>(W2 + (size_t)row * C + ch0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.x, w[0], acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.y, w[1], acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.z, w[2], acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w, w[3], acc[mt], 0, 0, 0);
        }
    }
    // bias + BN + ReLU of linear_w[2..4]; acc[mt][r] = hid[neighbour fr][unit 16mt + 4fg + r]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int u = mt * 16 + fg * 4 + r;
            float v = 0.f;
            if (u < CS) v = fmaxf((acc[mt][r] + b2s[u]) * s3s[u] + t3s[u], 0.f);
            acc[mt][r] = v;
        }
    // ---- logit^T = W5 . hid^T + b5, then softmax over the point's neighbours (one DPP row / half row)
    pf32x4 sm[MT];
#pragma unroll
    for (int mo = 0; mo < MT; ++mo) {
        pf32x4 lg = {0.f, 0.f, 0.f, 0.f};
        const int row = mo * 16 + fr;
#pragma unroll
        for (int mu = 0; mu < MT; ++mu) {
            const int u0 = mu * 16 + fg * 4;
            float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < CS && u0 < CS) f = *reinterpret_cast<const float4*>(W5s + row * CS + u0);
            lg = __builtin_amdgcn_mfma_f32_16x16x4f32(f.x, acc[mu][0], lg, 0, 0, 0);
            lg = __builtin_amdgcn_mfma_f32_16x16x4f32(f.y, acc[mu][1], lg, 0, 0, 0);
            lg = __builtin_amdgcn_mfma_f32_16x16x4f32(f.z, acc[mu][2], lg, 0, 0, 0);
            lg = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w, acc[mu][3], lg, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tt = mo * 16 + fg * 4 + r;
            const float l = lg[r] + (tt < CS ? b5s[tt] : 0.f);
            const float m = pt_group_max<NS>(l);
            const float e = __expf(l - m);
            sm[mo][r] = e / pt_group_sum<NS>(e);
        }
    }
    // ---- aggregation: channel ch uses the softmax of hidden unit ch % CS; this lane owns units 16mo + 4fg + r
    float* orow = a.out + (size_t)i * a.ldo;
#pragma unroll
    for (int mo = 0; mo < MT; ++mo) {
        if (mo * 16 + fg * 4 < CS) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int ch0 = m * CS + mo * 16 + fg * 4;
                const float4 v4 = *reinterpret_cast<const float4*>(vr + ch0);
                const float4 wa = *reinterpret_cast<const float4*>(W3s + ch0 * 3), wb = *reinterpret_cast<const float4*>(W3s + ch0 * 3 + 4),
                             wc = *reinterpret_cast<const float4*>(W3s + ch0 * 3 + 8);
                const float4 b3 = *reinterpret_cast<const float4*>(b3s + ch0);
                float o0 = (v4.x + (wa.x * h[0] + wa.y * h[1] + wa.z * h[2] + b3.x)) * sm[mo][0];
                float o1 = (v4.y + (wa.w * h[0] + wb.x * h[1] + wb.y * h[2] + b3.y)) * sm[mo][1];
                float o2 = (v4.z + (wb.z * h[0] + wb.w * h[1] + wc.x * h[2] + b3.z)) * sm[mo][2];
                float o3 = (v4.w + (wc.y * h[0] + wc.z * h[1] + wc.w * h[2] + b3.w)) * sm[mo][3];
                o0 = pt_group_sum<NS>(o0); o1 = pt_group_sum<NS>(o1); o2 = pt_group_sum<NS>(o2); o3 = pt_group_sum<NS>(o3);
                if (jn == 0 && valid) {
                    if (a.s_out) {
                        const float4 so = *reinterpret_cast<const float4*>(sos + ch0), to = *reinterpret_cast<const float4*>(tos + ch0);
                        o0 = fmaxf(o0 * so.x + to.x, 0.f); o1 = fmaxf(o1 * so.y + to.y, 0.f);
                        o2 = fmaxf(o2 * so.z + to.z, 0.f); o3 = fmaxf(o3 * so.w + to.w, 0.f);
                    }
                    *reinterpret_cast<float4*>(orow + ch0) = make_float4(o0, o1, o2, o3);
                }
            }
        }
    }
    }
}

template <int C_, int NS_>
static int launch_pt_mfma(const PtAttnParams& a, const float* W2, hipStream_t st) {
    constexpr int CS = C_ / 8, PPW = 16 / NS_;
    const size_t lds = (size_t)(8 * C_ + 4 * CS + CS * CS + (C_ <= 256 ? CS * (C_ + 4) : 0)) * sizeof(float);
    auto kern = pt_attention_mfma_kernel<C_, NS_>;
    static int per_cu = 0;
    if (per_cu == 0) {
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ETCH_EUNSUPPORTED;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 256, lds) != hipSuccess || nb < 1) nb = 1;
        per_cu = nb > 8 ? 8 : nb;
    }
    const long ntiles = ((long)a.n + 4 * PPW - 1) / (4 * PPW);
    long blocks = (long)etch_cu_count() * per_cu;
    if (blocks > ntiles) blocks = ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, a, W2, ntiles);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

extern "C" int etch_pt_attention_mfma(int n, int c, int ns, const float* p, const float* xq, const float* xk, const float* xv, long ldq,
                                      const int* idx, const float* const* params, const float* W2, float* out, long ldo, void* stream) {
    if (n <= 0) return ETCH_OK;
    if ((ldq & 3) || (ldo & 3) || !W2) return ETCH_EINVAL;
    PtAttnParams a;
    fill_pt_params(a, n, c, ns, p, xq, xk, xv, ldq, idx, params, out, ldo);
    hipStream_t st = (hipStream_t)stream;
#define PT_MFMA_CASE(C_, NS_) \
    if (c == C_ && ns == NS_) return launch_pt_mfma<C_, NS_>(a, W2, st);
    PT_MFMA_CASE(64, 8) PT_MFMA_CASE(128, 8) PT_MFMA_CASE(64, 16) PT_MFMA_CASE(128, 16) PT_MFMA_CASE(256, 16) PT_MFMA_CASE(512, 16)
#undef PT_MFMA_CASE
    return ETCH_EUNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------
// TransitionDown (pointtransformer_seg.py:40-68, stride != 1) without the grouped rows: the reference builds m*ns rows
// [p_j - p_i | x_j], applies Linear(3 + c -> c_out, bias=False) + BN + ReLU and max-pools over the ns neighbours.  The linear map
// splits as  W [p_j - p_i | x_j] = Wp (p_j - p_i) + Wx x_j : the x part is a per-SOURCE-point product ux = x Wx^T (n rows instead
// of m*ns = 4n), the coordinate part is 3 FMAs on the same difference the reference forms (no cancellation introduced).
//   out[i, o] = max_j relu(bn(ux[idx[i,j], o] + Wp[o,:] . (p[idx[i,j]] - new_p[i])))
// thread = (output point, 4 output channels).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pt_down_gather_max_kernel(int m, int ns, int co, const float* __restrict__ ux, long ldu,
                                                                 const float* __restrict__ p, const float* __restrict__ new_p,
                                                                 const int* __restrict__ idx, const float* __restrict__ Wp,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 float* __restrict__ out) {
    const int c4 = co >> 2;
    const size_t total = (size_t)m * c4;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e / c4), o0 = (int)(e - (size_t)i * c4) * 4;
        const float qx = new_p[(size_t)i * 3], qy = new_p[(size_t)i * 3 + 1], qz = new_p[(size_t)i * 3 + 2];
        float w[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) { w[u][0] = Wp[(o0 + u) * 3]; w[u][1] = Wp[(o0 + u) * 3 + 1]; w[u][2] = Wp[(o0 + u) * 3 + 2]; }
        const float4 sc = *reinterpret_cast<const float4*>(scale + o0), sh = *reinterpret_cast<const float4*>(shift + o0);
        float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int j = 0; j < ns; ++j) {
            const int q = idx[(size_t)i * ns + j];
            const float rx = p[(size_t)q * 3] - qx, ry = p[(size_t)q * 3 + 1] - qy, rz = p[(size_t)q * 3 + 2] - qz;
            const float4 v = *reinterpret_cast<const float4*>(ux + (size_t)q * ldu + o0);
            best.x = fmaxf(best.x, fmaxf((v.x + (w[0][0] * rx + w[0][1] * ry + w[0][2] * rz)) * sc.x + sh.x, 0.f));
            best.y = fmaxf(best.y, fmaxf((v.y + (w[1][0] * rx + w[1][1] * ry + w[1][2] * rz)) * sc.y + sh.y, 0.f));
            best.z = fmaxf(best.z, fmaxf((v.z + (w[2][0] * rx + w[2][1] * ry + w[2][2] * rz)) * sc.z + sh.z, 0.f));
            best.w = fmaxf(best.w, fmaxf((v.w + (w[3][0] * rx + w[3][1] * ry + w[3][2] * rz)) * sc.w + sh.w, 0.f));
        }
        *reinterpret_cast<float4*>(out + (size_t)i * co + o0) = best;
    }
}

extern "C" int etch_pt_down_gather_max(int m, int ns, int co, const float* ux, long ldu, const float* p, const float* new_p, const int* idx,
                                       const float* Wp, const float* scale, const float* shift, float* out, void* stream) {
    if (m <= 0) return ETCH_OK;
    if (co <= 0 || (co & 3) || (ldu & 3) || ns <= 0 || !scale || !shift) return ETCH_EINVAL;
    size_t blocks = ((size_t)m * (co / 4) + 255) / 256;
    if (blocks > 65535u * 16u) blocks = 65535u * 16u;
    hipLaunchKernelGGL(pt_down_gather_max_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, m, ns, co, ux, ldu, p, new_p, idx,
                       Wp, scale, shift, out);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

#ifdef ETCH_BUILD_EXPERIMENTS      // measured 1.0 - 3.4 x slower than the four-kernel block (profiles/r03_pt_block_fusion.txt): a lab record, built only on request
// ------------------------------------------------------------------------------------------------
// PointTransformerBlock (pointtransformer_seg.py:101-122) in TWO kernels instead of four, and consecutive blocks of a level chained:
//   K1  pt_block_k1_kernel      qkv = (relu(bn1(x W1^T))) Wqkv^T + bqkv                              (linear1 -> bn1 -> ReLU -> linear_q|k|v)
//   K2  pt_block_k2_kernel      att = relu(bn2(vector attention(qkv, kNN idx)))                       (transformer2, bn2, ReLU)
//                               out = relu(bn3(att W3^T) + x)                                        (linear3, bn3, residual, ReLU)
//                    CHAIN:     qkv' = K1 of the NEXT block on the out tile                          (its linear1 / q|k|v)
// A workgroup owns a tile of 16 points.  Everything between the gathered q|k|v rows and the block output stays in LDS: the
// n x c tensors y1 = relu(bn1(linear1 x)) and att never exist in HBM (4 of the 7 n x c round trips of a block), and a level with b
// blocks takes 1 + b launches instead of 4 b.  The 16-row GEMMs run on the fp32 matrix cores with the weight rows streamed from L2
// (c x c floats per 16 points), wave w owning the output column tiles w, w + 4, ...
// ------------------------------------------------------------------------------------------------
struct PtBlockTail {
    const float* W3l; const float* s3; const float* t3;      // linear3 [c][c], bn3 folded
    const float* xres; long ldx;                              // block input (residual)
    float* out; long ldo;
    // CHAIN: the next block's K1
    const float* W1n; const float* s1n; const float* t1n;     // linear1 [c][c], bn1 folded
    const float* Wqkvn; const float* bqkvn;                   // [3c][c], [3c]
    float* qkvn; long ldqn;
};

// D[16][O] = A[16][K] . W[O][K]^T; A in LDS (row stride lda), W row-major in global memory.  epi(col, acc): acc[r] = D[4 fg + r][col].
template <int K, int NB, class Epi>
__device__ __forceinline__ void pt_tile_gemm(const float* As, int lda, const float* __restrict__ W, int O, int wave, int fr, int fg, Epi epi) {
    const int nct = O >> 4;
    for (int c0 = wave; c0 < nct; c0 += 4 * NB) {
        pf32x4 acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = (pf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int t = 0; t < K / 16; ++t) {
            const float4 af = *reinterpret_cast<const float4*>(As + fr * lda + t * 16 + fg * 4);
            float4 wf[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int ct = c0 + 4 * b;
                wf[b] = ct < nct ? *reinterpret_cast<const float4*>(W + (size_t)(ct * 16 + fr) * K + t * 16 + fg * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, wf[b].x, acc[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, wf[b].y, acc[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, wf[b].z, acc[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, wf[b].w, acc[b], 0, 0, 0);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (c0 + 4 * b < nct) epi((c0 + 4 * b) * 16 + fr, acc[b]);
    }
}

// K1 on one 16-row tile: Xt (LDS, [16][C + 4]) -> Yt = relu(bn1(Xt W1^T)) (LDS) -> qkv rows (global).  Barriers inside.
template <int C>
__device__ __forceinline__ void pt_tile_k1(const float* Xt, float* Yt, const float* __restrict__ W1, const float* __restrict__ s1,
                                           const float* __restrict__ t1, const float* __restrict__ Wqkv, const float* __restrict__ bqkv,
                                           float* __restrict__ qkv, long ldq, long row0, int n, int wave, int fr, int fg) {
    constexpr int LD = C + 4;
    pt_tile_gemm<C, (C >= 128 ? 2 : 1)>(Xt, LD, W1, C, wave, fr, fg, [&](int col, const pf32x4& acc) {
        const float sc = s1[col], sh = t1[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) Yt[(4 * fg + r) * LD + col] = fmaxf(acc[r] * sc + sh, 0.f);
    });
    __syncthreads();
    pt_tile_gemm<C, (C >= 128 ? 4 : 3)>(Yt, LD, Wqkv, 3 * C, wave, fr, fg, [&](int col, const pf32x4& acc) {
        const float bs = bqkv[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long row = row0 + 4 * fg + r;
            if (row < n) qkv[row * ldq + col] = acc[r] + bs;
        }
    });
}

template <int C>
__global__ void __launch_bounds__(256) pt_block_k1_kernel(int n, const float* __restrict__ x, long ldx, const float* __restrict__ W1,
                                                          const float* __restrict__ s1, const float* __restrict__ t1,
                                                          const float* __restrict__ Wqkv, const float* __restrict__ bqkv,
                                                          float* __restrict__ qkv, long ldq) {
    constexpr int LD = C + 4;
    extern __shared__ __attribute__((aligned(16))) float cst[];
    float* Xt = cst;                 // [16][LD]
    float* Yt = cst + 16 * LD;       // [16][LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const long ntiles = ((long)n + 15) >> 4;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * 16;
        for (int e = tid; e < 16 * (C / 4); e += 256) {
            const int r = e / (C / 4), c4 = e - r * (C / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + r < n) v = *reinterpret_cast<const float4*>(x + (row0 + r) * ldx + c4 * 4);
            *reinterpret_cast<float4*>(&Xt[r * LD + c4 * 4]) = v;
        }
        __syncthreads();
        pt_tile_k1<C>(Xt, Yt, W1, s1, t1, Wqkv, bqkv, qkv, ldq, row0, n, wave, fr, fg);
        __syncthreads();
    }
}

template <int C, int NS, bool CHAIN>
__global__ void __launch_bounds__(256, 2) pt_block_k2_kernel(PtAttnParams a, const float* __restrict__ W2, PtBlockTail tl, long ntiles) {
    constexpr int CS = C / 8;
    constexpr int MT = CS <= 16 ? 1 : CS / 16;
    constexpr int KT = C / 16;
    constexpr int PPW = 16 / NS;                   // points per wave and pass
    constexpr int NPASS = 16 / (4 * PPW);          // passes of the 4 waves over the tile's 16 points
    constexpr bool W2_LDS = C <= 256;
    constexpr int LDW = C + 4;
    constexpr int LD = C + 4;                      // tile row stride
    extern __shared__ __attribute__((aligned(16))) float cst[];
    float* W3s = cst;                              // [C][3]   linear_p[3]
    float* b3s = W3s + 3 * C;
    float* scs = b3s + C;
    float* shs = scs + C;
    float* sos = shs + C;
    float* tos = sos + C;
    float* b2s = tos + C;
    float* s3s = b2s + CS;
    float* t3s = s3s + CS;
    float* b5s = t3s + CS;
    float* W5s = b5s + CS;                         // [CS][CS]
    float* W2s = W5s + CS * CS;                    // [CS][LDW]  (W2_LDS only)
    float* At = W2s + (W2_LDS ? CS * LDW : 0);     // [16][LD]  attention output tile (after bn2 + ReLU); CHAIN: y1 of the next block
    float* Ot = At + 16 * LD;                      // [16][LD]  block output tile (CHAIN only)
    const int tid = threadIdx.x;
    for (int e = tid; e < 3 * C; e += 256) W3s[e] = a.W3[e];
    for (int e = tid; e < C; e += 256) {
        b3s[e] = a.b3[e]; scs[e] = a.s_w0[e]; shs[e] = a.t_w0[e];
        sos[e] = a.s_out[e]; tos[e] = a.t_out[e];
    }
    for (int e = tid; e < CS; e += 256) { b2s[e] = a.b2[e]; s3s[e] = a.s_w3[e]; t3s[e] = a.t_w3[e]; b5s[e] = a.b5[e]; }
    for (int e = tid; e < CS * CS; e += 256) W5s[e] = a.W5[e];
    if (W2_LDS)
        for (int e = tid; e < CS * C / 4; e += 256) {
            const int row = e / (C / 4), c4 = e - row * (C / 4);
            *reinterpret_cast<float4*>(&W2s[row * LDW + c4 * 4]) = *reinterpret_cast<const float4*>(W2 + (size_t)row * C + c4 * 4);
        }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int pw = fr / NS, jn = fr % NS;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * 16;
#pragma unroll 1
        for (int pass = 0; pass < NPASS; ++pass) {
            asm volatile("" ::: "memory");
            const int lrow = (pass * 4 + wave) * PPW + pw;            // row of the tile this lane's point owns
            const long gi = row0 + lrow;
            const bool valid = gi < a.n;
            const int i = valid ? (int)gi : a.n - 1;
            const int j = a.idx[(size_t)i * NS + jn];
            float h[3];
            pt_rel_hidden(a, i, j, h);
            const float* kr = a.xk + (size_t)j * a.ldq;
            const float* qr = a.xq + (size_t)i * a.ldq;
            const float* vr = a.xv + (size_t)j * a.ldq;
            pf32x4 acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = (pf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int t = 0; t < KT; ++t) {
                const int ch0 = t * 16 + fg * 4;
                const float4 k4 = *reinterpret_cast<const float4*>(kr + ch0), q4 = *reinterpret_cast<const float4*>(qr + ch0);
                const float4 wa = *reinterpret_cast<const float4*>(W3s + ch0 * 3), wb = *reinterpret_cast<const float4*>(W3s + ch0 * 3 + 4),
                             wc = *reinterpret_cast<const float4*>(W3s + ch0 * 3 + 8);
                const float4 b3 = *reinterpret_cast<const float4*>(b3s + ch0), sc = *reinterpret_cast<const float4*>(scs + ch0),
                             sh = *reinterpret_cast<const float4*>(shs + ch0);
                float w[4];
                w[0] = fmaxf((k4.x - q4.x + (wa.x * h[0] + wa.y * h[1] + wa.z * h[2] + b3.x)) * sc.x + sh.x, 0.f);
                w[1] = fmaxf((k4.y - q4.y + (wa.w * h[0] + wb.x * h[1] + wb.y * h[2] + b3.y)) * sc.y + sh.y, 0.f);
                w[2] = fmaxf((k4.z - q4.z + (wb.z * h[0] + wb.w * h[1] + wc.x * h[2] + b3.z)) * sc.z + sh.z, 0.f);
                w[3] = fmaxf((k4.w - q4.w + (wc.y * h[0] + wc.z * h[1] + wc.w * h[2] + b3.w)) * sc.w + sh.w, 0.f);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int row = mt * 16 + fr;
                    float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row < CS) f = W2_LDS ? *reinterpret_cast<const float4*>(W2s + row * LDW + ch0) : *reinterpret_cast<const float4*>(W2 + (size_t)row * C + ch0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.x, w[0], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.y, w[1], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.z, w[2], acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w, w[3], acc[mt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int u = mt * 16 + fg * 4 + r;
                    float v = 0.f;
                    if (u < CS) v = fmaxf((acc[mt][r] + b2s[u]) * s3s[u] + t3s[u], 0.f);
                    acc[mt][r] = v;
                }
            pf32x4 sm[MT];
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) {
                pf32x4 lg = {0.f, 0.f, 0.f, 0.f};
                const int row = mo * 16 + fr;
#pragma unroll
                for (int mu = 0; mu < MT; ++mu) {
                    const int u0 = mu * 16 + fg * 4;
                    float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row < CS && u0 < CS) f = *reinterpret_cast<const float4*>(W5s + row * CS + u0);
                    lg = __builtin_amdgcn_mfma_f32_16x16x4f32(f.x, acc[mu][0], lg, 0, 0, 0);
                    lg = __builtin_amdgcn_mfma_f32_16x16x4f32(f.y, acc[mu][1], lg, 0, 0, 0);
                    lg = __builtin_amdgcn_mfma_f32_16x16x4f32(f.z, acc[mu][2], lg, 0, 0, 0);
                    lg = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w, acc[mu][3], lg, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int tt = mo * 16 + fg * 4 + r;
                    const float l = lg[r] + (tt < CS ? b5s[tt] : 0.f);
                    const float m = pt_group_max<NS>(l);
                    const float e = __expf(l - m);
                    sm[mo][r] = e / pt_group_sum<NS>(e);
                }
            }
            float* arow = At + lrow * LD;
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) {
                if (mo * 16 + fg * 4 < CS) {
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const int ch0 = m * CS + mo * 16 + fg * 4;
                        const float4 v4 = *reinterpret_cast<const float4*>(vr + ch0);
                        const float4 wa = *reinterpret_cast<const float4*>(W3s + ch0 * 3), wb = *reinterpret_cast<const float4*>(W3s + ch0 * 3 + 4),
                                     wc = *reinterpret_cast<const float4*>(W3s + ch0 * 3 + 8);
                        const float4 b3 = *reinterpret_cast<const float4*>(b3s + ch0);
                        float o0 = (v4.x + (wa.x * h[0] + wa.y * h[1] + wa.z * h[2] + b3.x)) * sm[mo][0];
                        float o1 = (v4.y + (wa.w * h[0] + wb.x * h[1] + wb.y * h[2] + b3.y)) * sm[mo][1];
                        float o2 = (v4.z + (wb.z * h[0] + wb.w * h[1] + wc.x * h[2] + b3.z)) * sm[mo][2];
                        float o3 = (v4.w + (wc.y * h[0] + wc.z * h[1] + wc.w * h[2] + b3.w)) * sm[mo][3];
                        o0 = pt_group_sum<NS>(o0); o1 = pt_group_sum<NS>(o1); o2 = pt_group_sum<NS>(o2); o3 = pt_group_sum<NS>(o3);
                        if (jn == 0) {
                            const float4 so = *reinterpret_cast<const float4*>(sos + ch0), to = *reinterpret_cast<const float4*>(tos + ch0);
                            float4 o = make_float4(fmaxf(o0 * so.x + to.x, 0.f), fmaxf(o1 * so.y + to.y, 0.f), fmaxf(o2 * so.z + to.z, 0.f),
                                                   fmaxf(o3 * so.w + to.w, 0.f));
                            if (!valid) o = make_float4(0.f, 0.f, 0.f, 0.f);
                            *reinterpret_cast<float4*>(arow + ch0) = o;
                        }
                    }
                }
            }
        }
        __syncthreads();
        // ---- out = relu(bn3(att W3^T) + x)
        pt_tile_gemm<C, (C >= 128 ? 2 : 1)>(At, LD, tl.W3l, C, wave, fr, fg, [&](int col, const pf32x4& acc) {
            const float sc = tl.s3[col], sh = tl.t3[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = row0 + 4 * fg + r;
                float v = 0.f;
                if (row < a.n) {
                    v = fmaxf(acc[r] * sc + sh + tl.xres[row * tl.ldx + col], 0.f);
                    tl.out[row * tl.ldo + col] = v;
                }
                if (CHAIN) Ot[(4 * fg + r) * LD + col] = v;
            }
        });
        __syncthreads();
        if (CHAIN) {
            pt_tile_k1<C>(Ot, At, tl.W1n, tl.s1n, tl.t1n, tl.Wqkvn, tl.bqkvn, tl.qkvn, tl.ldqn, row0, a.n, wave, fr, fg);
            __syncthreads();
        }
    }
}

template <int C_>
static int launch_pt_k1(int n, const float* x, long ldx, const float* W1, const float* s1, const float* t1, const float* Wqkv, const float* bqkv,
                        float* qkv, long ldq, hipStream_t st) {
    const long ntiles = ((long)n + 15) >> 4;
    const size_t lds = (size_t)2 * 16 * (C_ + 4) * sizeof(float);
    auto kern = pt_block_k1_kernel<C_>;
    static bool ready = false;
    if (!ready) {
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ETCH_EUNSUPPORTED;
        ready = true;
    }
    long blocks = (long)etch_cu_count() * 4;
    if (blocks > ntiles) blocks = ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, n, x, ldx, W1, s1, t1, Wqkv, bqkv, qkv, ldq);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

template <int C_, int NS_, bool CHAIN>
static int launch_pt_k2(const PtAttnParams& a, const float* W2, const PtBlockTail& tl, hipStream_t st) {
    constexpr int CS = C_ / 8;
    const size_t lds = (size_t)(8 * C_ + 4 * CS + CS * CS + (C_ <= 256 ? CS * (C_ + 4) : 0) + (CHAIN ? 2 : 1) * 16 * (C_ + 4)) * sizeof(float);
    auto kern = pt_block_k2_kernel<C_, NS_, CHAIN>;
    static int per_cu = 0;
    if (per_cu == 0) {
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ETCH_EUNSUPPORTED;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 256, lds) != hipSuccess || nb < 1) nb = 1;
        per_cu = nb > 4 ? 4 : nb;
    }
    const long ntiles = ((long)a.n + 15) >> 4;
    long blocks = (long)etch_cu_count() * per_cu;
    if (blocks > ntiles) blocks = ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, a, W2, tl, ntiles);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

extern "C" int etch_pt_block_k1(int n, int c, const float* x, long ldx, const float* W1, const float* s1, const float* t1, const float* Wqkv,
                                const float* bqkv, float* qkv, long ldq, void* stream) {
    if (n <= 0) return ETCH_OK;
    if ((ldx & 3) || (ldq & 3) || !x || !W1 || !s1 || !t1 || !Wqkv || !bqkv || !qkv) return ETCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    switch (c) {
        case 64: return launch_pt_k1<64>(n, x, ldx, W1, s1, t1, Wqkv, bqkv, qkv, ldq, st);
        case 128: return launch_pt_k1<128>(n, x, ldx, W1, s1, t1, Wqkv, bqkv, qkv, ldq, st);
        case 256: return launch_pt_k1<256>(n, x, ldx, W1, s1, t1, Wqkv, bqkv, qkv, ldq, st);
        case 512: return launch_pt_k1<512>(n, x, ldx, W1, s1, t1, Wqkv, bqkv, qkv, ldq, st);
    }
    return ETCH_EUNSUPPORTED;
}

// tail = {W3l, s3, t3, xres, W1n, s1n, t1n, Wqkvn, bqkvn} (the last five NULL: no chained K1); params as etch_pt_attention (16 entries,
// [14], [15] = bn2 folded, required)
extern "C" int etch_pt_block_k2(int n, int c, int ns, const float* p, const float* xq, const float* xk, const float* xv, long ldq,
                                const int* idx, const float* const* params, const float* W2, const float* const* tail, long ldx,
                                float* out, long ldo, float* qkv_next, long ldqn, void* stream) {
    if (n <= 0) return ETCH_OK;
    if ((ldq & 3) || (ldo & 3) || (ldx & 3) || !W2 || !params || !tail || !params[14] || !params[15] || !out) return ETCH_EINVAL;
    for (int k = 0; k < 4; ++k) if (!tail[k]) return ETCH_EINVAL;
    const bool chain = tail[4] != nullptr;
    if (chain && (!tail[5] || !tail[6] || !tail[7] || !tail[8] || !qkv_next || (ldqn & 3))) return ETCH_EINVAL;
    PtAttnParams a;
    fill_pt_params(a, n, c, ns, p, xq, xk, xv, ldq, idx, params, nullptr, 0);
    PtBlockTail tl;
    tl.W3l = tail[0]; tl.s3 = tail[1]; tl.t3 = tail[2]; tl.xres = tail[3]; tl.ldx = ldx; tl.out = out; tl.ldo = ldo;
    tl.W1n = tail[4]; tl.s1n = tail[5]; tl.t1n = tail[6]; tl.Wqkvn = tail[7]; tl.bqkvn = tail[8]; tl.qkvn = qkv_next; tl.ldqn = ldqn;
    hipStream_t st = (hipStream_t)stream;
#define PT_K2_CASE(C_, NS_) \
    if (c == C_ && ns == NS_) return chain ? launch_pt_k2<C_, NS_, true>(a, W2, tl, st) : launch_pt_k2<C_, NS_, false>(a, W2, tl, st);
    PT_K2_CASE(64, 8) PT_K2_CASE(128, 8) PT_K2_CASE(128, 16) PT_K2_CASE(256, 16) PT_K2_CASE(512, 16)
#undef PT_K2_CASE
    return ETCH_EUNSUPPORTED;
}
#endif  // ETCH_BUILD_EXPERIMENTS
