// The un-fused operator forms of the reference's functional API, as HIP kernels for gfx950.
//
// The hot path never calls these: the fused kernels of so3conv.hip / heads.hip regenerate the kernel weights in registers and
// never materialise the [b,p,60,24,nn] / [b,c,12,p,60] tensors.  They exist so that callers of the reference's operator names
// get the reference's tensors (shapes, layouts, values), and as an independent cross-check of the fused kernels:
//
//   etch_inter_kernel_weights   vgtk/so3conv/functional.py:286-324  inter_so3conv_grouping_anchor
//   etch_inter_feat_grouping    vgtk/so3conv/functional.py:61-67    inter_so3conv_feat_grouping (on shadow-padded feats, :101-105)
//   etch_intra_grouping         vgtk/so3conv/functional.py:331-378  intra_so3conv_grouping
//   etch_square_distance        src/models/pointnet2_utils.py:4-23  square_distance (expansion formula, same op order)
//   etch_index_points           src/models/pointnet2_utils.py:26-43 index_points
//   etch_so3_mean               src/models/so3conv.py:186-225 lives in heads.hip (so3_mean_dir_kernel, want_R)
// (paths relative to /root/reference/external/vgtk or /root/reference)
#include "common.h"

// w[b,p,a,k,n] = relu(1 - sum_xyz (g[b,:,p,n] - rk[:,a,k])^2 / sigma);  g (b,3,p,nn), rk (na,ks,3) = anchors @ kernels^T
__global__ void __launch_bounds__(256) inter_kernel_weights_kernel(long total, int p, int nn, int na, int ks, const float* __restrict__ g,
                                                                    const float* __restrict__ rk, float sigma, float* __restrict__ w) {
#pragma clang fp contract(off)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i % nn);
        long r = i / nn;
        const int k = (int)(r % ks); r /= ks;
        const int a = (int)(r % na); r /= na;
        const int pi = (int)(r % p);
        const long b = r / p;
        const float* gb = g + ((b * 3) * p + pi) * (long)nn + n;
        const float* q = rk + ((long)a * ks + k) * 3;
        const float dx = gb[0] - q[0], dy = gb[(long)p * nn] - q[1], dz = gb[2l * p * nn] - q[2];
        const float d = (dx * dx + dy * dy) + dz * dz;
        const float v = 1.0f - d / sigma;
        w[i] = v > 0.f ? v : 0.f;
    }
}

// out[b,c,k,p,a] = sum_n feats[b,c,idx[b,p,n],a] * w[b,p,a,k,n];  feats (b,c,q,na) reference layout (q includes the shadow row)
// one workgroup per (b, p): the nn x na gathered rows of one channel are staged in LDS, w tile read once per channel group
__global__ void __launch_bounds__(256) inter_feat_grouping_kernel(int c, int q, int p, int nn, int na, int ks, const int* __restrict__ idx,
                                                                   const float* __restrict__ w, const float* __restrict__ feats,
                                                                   float* __restrict__ out) {
    extern __shared__ float sm[];                    // [nn][na] gathered features of the current channel
    const int pi = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int* id = idx + ((long)b * p + pi) * nn;
    const float* wb = w + ((long)b * p + pi) * (long)na * ks * nn;
    for (int ch = 0; ch < c; ++ch) {
        const float* fb = feats + ((long)b * c + ch) * (long)q * na;
        __syncthreads();
        for (int e = tid; e < nn * na; e += 256) { const int n = e / na, a = e - n * na; sm[e] = fb[(long)id[n] * na + a]; }
        __syncthreads();
        for (int e = tid; e < ks * na; e += 256) {
            const int k = e / na, a = e - k * na;
            const float* wr = wb + ((long)a * ks + k) * nn;
            float acc = 0.f;
            for (int n = 0; n < nn; ++n) acc += sm[n * na + a] * wr[n];
            out[((((long)b * c + ch) * ks + k) * p + pi) * na + a] = acc;
        }
    }
}

// out[b,c,t,p,a] = feat[b,c,p,intra_idx[a,t]]
__global__ void __launch_bounds__(256) intra_grouping_kernel(long total, int p, int na, int nt, const long long* __restrict__ intra_idx,
                                                              const float* __restrict__ feat, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int a = (int)(i % na);
        long r = i / na;
        const int pi = (int)(r % p); r /= p;
        const int t = (int)(r % nt);
        const long bc = r / nt;
        out[i] = feat[(bc * p + pi) * na + (int)intra_idx[a * nt + t]];
    }
}

// dist[b,n,m] = (-2 * <src_n, dst_m>) + |src_n|^2 + |dst_m|^2, in the reference's operation order (matmul, then two in-place adds)
__global__ void __launch_bounds__(256) square_distance_kernel(long total, int N, int M, int C, const float* __restrict__ src,
                                                               const float* __restrict__ dst, float* __restrict__ out) {
#pragma clang fp contract(off)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i % M);
        const long r = i / M;
        const int n = (int)(r % N);
        const long b = r / N;
        const float* s = src + (b * N + n) * C;
        const float* d = dst + (b * M + m) * C;
        float dot = 0.f, ss = 0.f, dd = 0.f;
        for (int c = 0; c < C; ++c) { dot += s[c] * d[c]; ss += s[c] * s[c]; dd += d[c] * d[c]; }
        out[i] = (-2.0f * dot + ss) + dd;
    }
}

// out[b,s,:] = points[b, idx[b,s], :]   (idx int64, any trailing index shape flattened to S)
__global__ void __launch_bounds__(256) index_points_kernel(long total, int N, long S, int C, const float* __restrict__ points,
                                                            const long long* __restrict__ idx, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long r = i / C;
        const long b = r / S;
        out[i] = points[(b * N + idx[r]) * C + c];
    }
}

static inline unsigned grid_for(long total) {
    long blocks = (total + 255) / 256;
    if (blocks > 65535l * 16) blocks = 65535l * 16;
    return (unsigned)blocks;
}

extern "C" {

int etch_inter_kernel_weights(int b, int p, int nn, int na, int ks, const float* grouped_xyz, const float* rotated_kernels, float sigma,
                              float* w, void* stream) {
    const long total = (long)b * p * na * ks * nn;
    if (total <= 0) return ETCH_OK;
    hipLaunchKernelGGL(inter_kernel_weights_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, total, p, nn, na, ks, grouped_xyz,
                       rotated_kernels, sigma, w);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_inter_feat_grouping(int b, int c, int q, int p, int nn, int na, int ks, const int* idx, const float* w, const float* feats,
                             float* out, void* stream) {
    if (b <= 0 || p <= 0 || c <= 0) return ETCH_OK;
    const size_t lds = sizeof(float) * (size_t)nn * na;
    if (lds > 64 * 1024 || b > 65535) return ETCH_EUNSUPPORTED;
    hipLaunchKernelGGL(inter_feat_grouping_kernel, dim3(p, b), dim3(256), lds, (hipStream_t)stream, c, q, p, nn, na, ks, idx, w, feats, out);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_intra_grouping(int b, int c, int p, int na, int nt, const long long* intra_idx, const float* feat, float* out, void* stream) {
    const long total = (long)b * c * nt * p * na;
    if (total <= 0) return ETCH_OK;
    hipLaunchKernelGGL(intra_grouping_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, total, p, na, nt, intra_idx, feat, out);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_square_distance(int B, int N, int M, int C, const float* src, const float* dst, float* out, void* stream) {
    const long total = (long)B * N * M;
    if (total <= 0) return ETCH_OK;
    hipLaunchKernelGGL(square_distance_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, total, N, M, C, src, dst, out);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_index_points(int B, int N, long S, int C, const float* points, const long long* idx, float* out, void* stream) {
    const long total = (long)B * S * C;
    if (total <= 0) return ETCH_OK;
    hipLaunchKernelGGL(index_points_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, total, N, S, C, points, idx, out);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

}  // extern "C"
