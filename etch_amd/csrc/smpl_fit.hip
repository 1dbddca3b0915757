// Stage 2 of ETCH on gfx950: marker aggregation, the two-stage Levenberg-Marquardt body-model fit and the final
// full-mesh LBS.
//
//   etch_argmax_rows   torch.max(part_labels, -1) of predict_smpl (/root/reference/src/inference_demo.py:52-53)
//   etch_get_markers   get_markers (/root/reference/src/models/fit_SMPL.py:17-62)
//   etch_smpl_lm_fit   fit_smpl's two Theseus LevenbergMarquardt stages (fit_SMPL.py:161-249) with the residual of
//                      marker_error_fn_{0,1} (:111-152).  The reference differentiates the FULL-mesh LBS with autograd
//                      (~2.2 GFLOP / scan / iteration); here the forward and an ANALYTIC Jacobian are restricted to the
//                      marker vertices (Appendix C): one persistent workgroup per scan keeps the whole fit on chip
//                      through all iterations.
//   etch_smpl_lbs      the final smpl_model(...) call (fit_SMPL.py:258-259): vertices (B,V,3) and the joints.
//
// Body models: the kernels are templated on (joints NJ, shape coefficients NB) and instantiated for SMPL (24, 10: the
// reference's model, fit_SMPL.py:100) and for an SMPL-X-sized model (55, 20: BASELINE configs[4]).  Variable
// vector x = pose[3(NJ-1)] | betas[NB] | global_orient[3] | transl[3] (the reference's order, fit_SMPL.py:174,225); stage 0
// optimises the first 2 betas only (:161-165), stage 1 all of them.
//
// [upstream, not in the reference tree] LBS = smplx.lbs.lbs, LM = theseus.LevenbergMarquardt (dense Cholesky, fixed
// damping, no step rejection, per-sample freeze on |d err| < 1e-10 or |d err|/err < 1e-8).  batch_rodrigues follows
// the in-tree copy src/data_utils/GT_dataloader_mixed.py:29-64 (angle = |theta + 1e-8|).
//
// Arithmetic: kinematics, J^T J / J^T r (fp64 matrix cores, v_mfma_f64_16x16x4_f64, on the fp32 Jacobian rows -- the
// reference's Jacobian is fp32 too) and the Cholesky solve in fp64.
//
// LDS plan (one workgroup = 12 waves per scan).  The Jacobian is never held whole: markers are linearised 12 at a time
// (one per wave) into a double-buffered 36-row chunk, and every wave accumulates its share of the 16x16 tiles of the lower
// triangle of [J | r]^T [J | r] in matrix-core accumulators across the chunks.  The packed normal matrix (fp64, 30 KB for
// SMPL, 144 KB for the 188-DoF model) ALIASES the linearisation scratch: it is written from the accumulators after the last
// chunk and consumed by the Cholesky solve before the next linearisation.
#include "common.h"
#include <cstdlib>

#ifndef LM_ELIM_COLS
#define LM_ELIM_COLS 3    // pivots eliminated per barrier in the 85-DoF solve (2: the two-column form, kept for A/B)
#endif
#ifndef LM_TIMERS
#define LM_TIMERS 1      // per-phase s_memrealtime breakdown in `phase_out` (profiles/scripts/lm_time.py); measured cost: none (82.4 vs 82.8 us / iteration without)
#endif

#define LM_MAXM 96                  // markers per scan
#define NB_STAGE0 2                 // fit_SMPL.py:161-165: stage 0 optimises betas[:2]

template <int NJ_, int NB_, int THREADS_>
struct Body {
    static constexpr int NJ = NJ_, NB = NB_;
    static constexpr int THREADS = THREADS_, WAVES = THREADS_ / 64;   // one workgroup per scan
    static constexpr int MC = THREADS_ / NJ_;           // markers linearised per chunk: one (marker, joint) item per thread (SMPL 32, 188-DoF model 9)
    static constexpr int CHUNK_ROWS = (3 * MC + 3) & ~3; // Jacobian rows of a chunk, padded to a multiple of 4 (MFMA K-steps); padding rows stay zero
    static constexpr int NPOSE = 3 * (NJ - 1);          // body pose variables
    static constexpr int NPF = 9 * (NJ - 1);            // pose-feature entries vec(R_k - I)
    static constexpr int DOF = NPOSE + NB + 6;
    static constexpr int NT = (DOF + 1 + 15) / 16;      // 16-column tiles of [J | r]
    static constexpr int LDJ = NT * 16;
    static constexpr int LDJS = LDJ + 16;               // row stride of a Jacobian chunk: the 4 rows of an MFMA K-step hit distinct banks
    static constexpr int NTILES = NT * (NT + 1) / 2;    // lower triangle
    static constexpr int TPW = (NTILES + WAVES - 1) / WAVES;
    static constexpr int NPACK = (DOF + 1) * (DOF + 2) / 2;
    static constexpr int VPARTS = 8;                    // lanes sharing one posed-vertex dot product
    static constexpr int EPP = (NPF + VPARTS - 1) / VPARTS;
    static constexpr int NQ = (3 * NB + 15) & ~15;      // beta-gradient columns, padded to MFMA tiles
    static constexpr int EPU = EPP <= 32 ? (EPP + 1) / 2 : 16;   // posedirs rows loaded back to back per wait (3 floats each)
};

// lm_linearize / lm_solve are out-of-line functions on purpose.  Their call frames -- ~40 callee-saved registers saved / restored ONCE per call,
// i.e. once per LM iteration -- are the 172 - 252 bytes of scratch the fit kernels report; inlined (-DLM_PHASE_INLINE=always_inline) the register
// allocator spills 104 (SMPL) / 42 (SMPL-X) registers INSIDE the phases instead, and the file takes 3.5 minutes to compile.
#ifndef LM_PHASE_INLINE
#define LM_PHASE_INLINE noinline
#endif
struct SmplConsts {
    const float* J0;      // [NJ][3]      J_regressor @ v_template
    const float* Jd;      // [NJ][3][NB]  J_regressor @ shapedirs
    const int* parents;   // [NJ]
    // marker-restricted tables (M markers)
    const float* mk_vt;   // [M][3]
    const float* mk_S;    // [M][3][NB]
    const float* mk_P;    // [M][NJ-1][28]: per joint k >= 1 the 9 x 3 posedirs block of the marker (entry (e, a) at 3 e + a), padded to 28 floats
                          //                so that a thread fetches its block with seven 16-byte loads
    const float* mk_W;    // [M][NJ]
};

// ---------------------------------------------------------------------------------------------- helpers
__device__ inline void rodrigues_d(const double th[3], double R[9], float dR[3][9], bool want_d) {
    // R = I + sin(a) K(n) + (1 - cos a) K(n)^2,  a = |theta + 1e-8|, n = theta / a
    const double e = 1e-8;
    const double a = sqrt((th[0] + e) * (th[0] + e) + (th[1] + e) * (th[1] + e) + (th[2] + e) * (th[2] + e));
    const double n[3] = {th[0] / a, th[1] / a, th[2] / a};
    const double s = sin(a), c = cos(a);
    double K[9] = {0, -n[2], n[1], n[2], 0, -n[0], -n[1], n[0], 0};
    double K2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) K2[i * 3 + j] = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + s * K[i] + (1.0 - c) * K2[i];
    if (!want_d) return;
    for (int q = 0; q < 3; ++q) {
        const double da = (th[q] + e) / a;
        double dn[3];
        for (int i = 0; i < 3; ++i) dn[i] = ((i == q ? 1.0 : 0.0) * a - th[i] * da) / (a * a);
        const double dK[9] = {0, -dn[2], dn[1], dn[2], 0, -dn[0], -dn[1], dn[0], 0};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double dKK = 0, KdK = 0;
                for (int m = 0; m < 3; ++m) { dKK += dK[i * 3 + m] * K[m * 3 + j]; KdK += K[i * 3 + m] * dK[m * 3 + j]; }
                dR[q][i * 3 + j] = (float)(c * da * K[i * 3 + j] + s * dK[i * 3 + j] + s * da * K2[i * 3 + j] + (1.0 - c) * (dKK + KdK));
            }
    }
}

// x layout: pose[3(nj-1)] | betas[nb] | orient[3] | transl[3]
__device__ inline void joint_theta(const double* x, int j, int npose, int nb, double th[3]) {
    const double* p = j == 0 ? x + npose + nb : x + 3 * (j - 1);
    th[0] = p[0]; th[1] = p[1]; th[2] = p[2];
}

// Forward kinematics by one thread: Rw_j = Rw_p R_j, tw_j = Rw_p (J_j - J_p) + tw_p
__device__ inline void fk_chain(int nj, const int* parents, const double* R, const double* Jj, double* Rw, double* tw) {
    for (int i = 0; i < 9; ++i) Rw[i] = R[i];
    for (int i = 0; i < 3; ++i) tw[i] = Jj[i];
    for (int j = 1; j < nj; ++j) {
        const int p = parents[j];
        const double* Rp = Rw + p * 9;
        const double* Rj = R + j * 9;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) Rw[j * 9 + a * 3 + b] = Rp[a * 3] * Rj[b] + Rp[a * 3 + 1] * Rj[3 + b] + Rp[a * 3 + 2] * Rj[6 + b];
        const double d[3] = {Jj[j * 3] - Jj[p * 3], Jj[j * 3 + 1] - Jj[p * 3 + 1], Jj[j * 3 + 2] - Jj[p * 3 + 2]};
        for (int a = 0; a < 3; ++a) tw[j * 3 + a] = Rp[a * 3] * d[0] + Rp[a * 3 + 1] * d[1] + Rp[a * 3 + 2] * d[2] + tw[p * 3 + a];
    }
}

// ---------------------------------------------------------------------------------------------- LM fit
template <class BM>
struct LmLin {                        // linearisation scratch (dead during the solve)
    double R[BM::NJ * 9], Rw[BM::NJ * 9], tw[BM::NJ * 3], Jj[BM::NJ * 3];
    double Ab[BM::NJ * 3];            // tw_j - Rw_j J_j
    double vp[LM_MAXM][3];            // posed marker vertices  v_t + S beta + P^T pf
    double Tbd[LM_MAXM][12];          // blended skinning transform sum_j W_vj [Rw_j | tw_j - Rw_j J_j]: rows a = 3 rotation entries + translation
    float Wq[LM_MAXM][BM::NQ];        // sum_j W_vj d(tw_j - Rw_j J_j)/d beta_l  (column l*3 + a)
    float dR[BM::NJ][3][9];
    float omega[BM::NJ][3][4];
    float twd[BM::NB][BM::NJ][3];
    float pf[BM::NPF + 1];
    float Jd[BM::NJ * 3 * BM::NB];    // LDS copy of the joint shape basis
    float J0[BM::NJ * 3];
    float Ay[BM::MC][BM::NJ][4];      // per chunk: W_vj * y_vj (xyz) and W_vj
    float Jc[2][BM::CHUNK_ROWS * BM::LDJS];   // double-buffered Jacobian chunk: 3 rows per marker, column DOF = residual
};

// A scan's linearisation split over G workgroups (latency regime: B <= 8 scans leave most of the chip idle): workgroup g of a scan takes the
// marker chunks g, g + G, ...; the partial tiles of [J | r]^T [J | r] meet in global memory once per linearisation and every workgroup adds
// them in the SAME order (g = 0 .. G-1), so all G copies of the scan's state stay bit-identical and solve redundantly -- no second exchange.
// ws: [2 parities][G][SLOTS = WAVES * TPW][64 lanes][4] fp64 (double-buffered by the linearisation count), ctr: arrival counter.
struct LmSplit {
    int g, G;
    double* ws;
    unsigned* ctr;          // ctr[0]: arrivals; ctr[1]: the scan's give-up flag (set by the first workgroup that timed out, read by all at every exchange)
    unsigned count;         // linearisations exchanged so far (uniform over the scan's workgroups)
    unsigned spin_limit;    // polls before a workgroup gives up waiting for its partners
};

template <class BM>
struct LmShared {
    union {
        LmLin<BM> lin;
        double A[BM::NPACK];          // packed lower triangle of [J^T J + lambda I ; g^T] (row DOF = rhs), then L and y
    };
    double x[BM::DOF];
    double delta[BM::DOF + 3];
    double rdiag[BM::DOF + 3];        // 1 / L_ii
    double rpiv;                      // 1 / A_kk of the column being eliminated
    double err;
    float target[3 * LM_MAXM], mask[LM_MAXM], resid[3 * LM_MAXM];
    int parents[BM::NJ];
    unsigned long long sub[BM::NJ];   // bit j of sub[k]: joint j lies in the subtree of joint k
    int lorder[BM::NJ], lstart[BM::NJ + 2], nlev;   // joints sorted by depth in the kinematic tree: level l = lorder[lstart[l] .. lstart[l+1])
    long long phase[8];               // s_memtime cycles per phase (thread 0), optional diagnostics
    SmplConsts consts;                // the kernel's table pointers and the split state live HERE, not in a by-reference struct of the noinline
    LmSplit split;                    // phases' caller (that is a private-memory copy: 88 + 32 bytes of scratch per thread)
    int split_failed;                 // a partner workgroup of a split fit never arrived (or had given up): this scan's fit is abandoned, its results are NaN
};

__device__ inline double& Apk(double* A, int i, int j) { return A[i * (i + 1) / 2 + j]; }   // i >= j

// wave-wide fp64 sum on the VALU (DPP row shifts + row broadcasts, no LDS traffic); result broadcast to all lanes
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ double dpp_add_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, BOUND);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, BOUND);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v = dpp_add_f64<0x111, 0xf, true>(v);     // row_shr:1
    v = dpp_add_f64<0x112, 0xf, true>(v);     // row_shr:2
    v = dpp_add_f64<0x114, 0xf, true>(v);     // row_shr:4
    v = dpp_add_f64<0x118, 0xf, true>(v);     // row_shr:8   -> lane 15 of every row holds the row sum
    v = dpp_add_f64<0x142, 0xa, false>(v);    // row_bcast:15 into rows 1 and 3
    v = dpp_add_f64<0x143, 0xc, false>(v);    // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// per-scan constants into LDS: parents, targets, subtree membership masks
template <class BM>
__device__ void lm_setup(LmShared<BM>& s, const SmplConsts& C, int M, const float* target, const float* mask) {
    constexpr int NJ = BM::NJ;
    const int tid = threadIdx.x;
    if (tid < 8) s.phase[tid] = 0;
    if (tid == 0) s.consts = C;                // what the noinline phases read (see LmShared::consts)
    if (tid < NJ) s.parents[tid] = C.parents[tid];
    if (tid < M * 3) s.target[tid] = target[tid];
    if (tid < M) s.mask[tid] = mask[tid];
    __syncthreads();
    if (tid < NJ) {                            // subtree membership masks
        unsigned long long m = 0ull;
        for (int j = 0; j < NJ; ++j) {
            int a = j;
            while (a > tid) a = s.parents[a];
            if (a == tid) m |= 1ull << j;
        }
        s.sub[tid] = m;
    }
    if (tid == 0) {                            // kinematic levels (parents precede their children: depth by one pass)
        int* depth = reinterpret_cast<int*>(s.A);      // scratch of this one-thread pass in the (still unused) matrix area: dynamically indexed
        int* cnt = depth + NJ;                          // private arrays would be scratch memory
        int maxd = 0;
        for (int j = 0; j < NJ; ++j) { depth[j] = j == 0 ? 0 : depth[s.parents[j]] + 1; maxd = depth[j] > maxd ? depth[j] : maxd; }
        for (int l = 0; l <= maxd + 1; ++l) cnt[l] = 0;
        for (int j = 0; j < NJ; ++j) ++cnt[depth[j] + 1];
        for (int l = 0; l <= maxd; ++l) cnt[l + 1] += cnt[l];
        for (int l = 0; l <= maxd + 1; ++l) s.lstart[l] = cnt[l];
        for (int j = 0; j < NJ; ++j) s.lorder[cnt[depth[j]]++] = j;
        s.nlev = maxd + 1;
    }
    __syncthreads();
}

// residual + normal equations at s.x.  nb = number of active betas (2 in stage 0, NB in stage 1).  On return s.A holds the packed
// lower triangle of J^T J (no damping yet) with the right-hand side -J^T r as row DOF, s.resid / s.err the residual and 0.5 |r|^2.
// jac_out (diagnostics): the marker rows of J (3M x DOF) are also written to global memory.  grad_only (first-order fitter): only the
// tile row that holds -J^T r is accumulated; the J^T J entries of s.A are then undefined.
template <class BM>
__device__ __attribute__((LM_PHASE_INLINE)) void lm_linearize(LmShared<BM>& s, int M, int nb, float* __restrict__ jac_out, bool grad_only = false,
                                                       bool use_split = false) {
    const SmplConsts& C = s.consts;
    LmSplit* sp = use_split ? &s.split : nullptr;
    const int grp_g = sp ? sp->g : 0, grp_G = sp ? sp->G : 1;
    constexpr int NJ = BM::NJ, NB = BM::NB, NPOSE = BM::NPOSE, NPF = BM::NPF, DOF = BM::DOF, LDJ = BM::LDJ, LDJS = BM::LDJS;
    LmLin<BM>& L = s.lin;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: tile indices stay in SGPRs
    for (int i = tid; i < NJ * 3 * NB; i += BM::THREADS) L.Jd[i] = C.Jd[i];      // the solve overwrote this region: reload (L2-resident)
    if (tid < NJ * 3) L.J0[tid] = C.J0[tid];
    __syncthreads();
    if (tid < NJ) {
        double th[3];
        joint_theta(s.x, tid, NPOSE, NB, th);
        rodrigues_d(th, L.R + tid * 9, L.dR[tid], true);
        for (int c = 0; c < 3; ++c) {
            double v = L.J0[tid * 3 + c];
            for (int l = 0; l < NB; ++l) v += (double)L.Jd[(tid * 3 + c) * NB + l] * s.x[NPOSE + l];
            L.Jj[tid * 3 + c] = v;
        }
    }
    __syncthreads();
    long long t0 = 0;
#if LM_TIMERS
    if (tid == 0) t0 = wall_clock64();
#endif
    for (int e = tid; e < NPF; e += BM::THREADS) {            // pose feature vec(R_k - I), k = 1..NJ-1
        const int k = 1 + e / 9, q = e - (k - 1) * 9;
        L.pf[e] = (float)(L.R[k * 9 + q] - ((q % 4 == 0) ? 1.0 : 0.0));
    }
    // forward kinematics and the d tw_j / d beta_l chain, one tree LEVEL per barrier (9 levels for SMPL instead of 23 dependent joint
    // steps): per joint of the level 12 threads (9 rotation entries + 3 translation) + 3 NB threads (one per (beta, component))
    if (tid < 9) L.Rw[tid] = L.R[tid];
    else if (tid < 12) L.tw[tid - 9] = L.Jj[tid - 9];
    else if (tid >= 64 && tid < 64 + 3 * NB) { const int l = (tid - 64) / 3, a = (tid - 64) - 3 * l; L.twd[l][0][a] = L.Jd[(0 * 3 + a) * NB + l]; }
    __syncthreads();
    for (int lv = 1; lv < s.nlev; ++lv) {
        const int lb = s.lstart[lv], cnt = s.lstart[lv + 1] - lb;
        constexpr int PER = 12 + 3 * NB;
        for (int it = tid; it < cnt * PER; it += BM::THREADS) {
            const int js = it / PER, e = it - js * PER;
            const int j = s.lorder[lb + js], p = s.parents[j];
            const double* Rp = L.Rw + p * 9;
            if (e < 9) {
                const int a = e / 3, b = e - a * 3;
                L.Rw[j * 9 + e] = Rp[a * 3] * L.R[j * 9 + b] + Rp[a * 3 + 1] * L.R[j * 9 + 3 + b] + Rp[a * 3 + 2] * L.R[j * 9 + 6 + b];
            } else if (e < 12) {
                const int a = e - 9;
                L.tw[j * 3 + a] = Rp[a * 3] * (L.Jj[j * 3] - L.Jj[p * 3]) + Rp[a * 3 + 1] * (L.Jj[j * 3 + 1] - L.Jj[p * 3 + 1]) +
                                  Rp[a * 3 + 2] * (L.Jj[j * 3 + 2] - L.Jj[p * 3 + 2]) + L.tw[p * 3 + a];
            } else {
                const int l = (e - 12) / 3, a = (e - 12) - 3 * l;
                const double d[3] = {(double)L.Jd[(j * 3 + 0) * NB + l] - L.Jd[(p * 3 + 0) * NB + l], (double)L.Jd[(j * 3 + 1) * NB + l] - L.Jd[(p * 3 + 1) * NB + l],
                                     (double)L.Jd[(j * 3 + 2) * NB + l] - L.Jd[(p * 3 + 2) * NB + l]};
                L.twd[l][j][a] = (float)(Rp[a * 3] * d[0] + Rp[a * 3 + 1] * d[1] + Rp[a * 3 + 2] * d[2] + L.twd[l][p][a]);
            }
        }
        __syncthreads();
    }
    if (tid < NJ * 3) {                       // omega_kc = Rw_parent(k) * axial(dR_kc R_k^T)
        const int k = tid / 3, c = tid - k * 3;
        const float* d = L.dR[k][c];
        const double* Rk = L.R + k * 9;
        double Sk[9];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) Sk[a * 3 + b] = d[a * 3] * Rk[b * 3] + d[a * 3 + 1] * Rk[b * 3 + 1] + d[a * 3 + 2] * Rk[b * 3 + 2];
        const double ax[3] = {0.5 * (Sk[7] - Sk[5]), 0.5 * (Sk[2] - Sk[6]), 0.5 * (Sk[3] - Sk[1])};
        if (k == 0) { for (int a = 0; a < 3; ++a) L.omega[k][c][a] = (float)ax[a]; }
        else {
            const double* Rp = L.Rw + s.parents[k] * 9;
            for (int a = 0; a < 3; ++a) L.omega[k][c][a] = (float)(Rp[a * 3] * ax[0] + Rp[a * 3 + 1] * ax[1] + Rp[a * 3 + 2] * ax[2]);
        }
    }
    __syncthreads();
    for (int e = tid; e < NJ * 3; e += BM::THREADS) {        // translation column of the skinning transform of joint j
        const int j = e / 3, a = e - j * 3;
        const double* Rj = L.Rw + j * 9 + a * 3;
        L.Ab[e] = L.tw[e] - (Rj[0] * L.Jj[j * 3] + Rj[1] * L.Jj[j * 3 + 1] + Rj[2] * L.Jj[j * 3 + 2]);
    }
    for (int e = tid; e < NB * NJ * 3; e += BM::THREADS) {   // Q[l][j] = d tw_j/d beta_l - Rw_j Jd_j[:,l]  (in place)
        const int l = e / (NJ * 3), r = e - l * NJ * 3, j = r / 3, a = r - j * 3;
        const double* Rj = L.Rw + j * 9;
        L.twd[l][j][a] -= (float)(Rj[a * 3] * L.Jd[(j * 3 + 0) * NB + l] + Rj[a * 3 + 1] * L.Jd[(j * 3 + 1) * NB + l] + Rj[a * 3 + 2] * L.Jd[(j * 3 + 2) * NB + l]);
    }
    __syncthreads();
#if LM_TIMERS
    if (tid == 0) { const long long t1 = wall_clock64(); s.phase[0] += t1 - t0; t0 = t1; }
#endif

    // ---- markers as flat passes of the whole workgroup:
    //   A   (all markers) posed vertices v_p = v_t + S beta + P^T pf; 8 lanes stream one marker's contiguous posedirs block
    //   A2  (all markers, matrix cores) blended skinning transform Tbd = W [Rw | tw - Rw J] (fp64) and Wq = W Q (the joint-chain part of
    //       the beta columns): the skinning weights are the A operand of both products
    //   A3  residuals from Tbd and v_p (fp64)
    //   per chunk of MC markers (one (marker, joint) item per thread):
    //   B   per-joint contributions Ay = W_vj (Rw_j (v_p - J_j) + tw_j); the joint's posedirs block and the shapedirs entries needed in C
    //       are requested here
    //   C   the chunk's Jacobian rows: thread (m, k) writes joint k's three rotation columns (subtree sum + posedirs block), thread
    //       (m, u) a beta column or the translation / residual / padding columns
    //   then every wave adds the chunk's contribution to its tiles of [J | r]^T [J | r] on the fp64 matrix cores (double-buffered chunk).
    constexpr int MC = BM::MC, NQ = BM::NQ;
    long long tp = 0;
#if LM_TIMERS
    if (tid == 0) tp = wall_clock64();
#endif
    for (int it = tid; it < M * BM::VPARTS; it += BM::THREADS) {
        const int part = it % BM::VPARTS, v = it / BM::VPARTS;
        const float* Pg = C.mk_P + (size_t)v * 28 * (NJ - 1);    // row e = 9 (k-1) + q of the pose feature sits at 28 (k-1) + 3 q
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        const int e0 = part * BM::EPP;
#pragma unroll 1
        for (int i0 = 0; i0 < BM::EPP; i0 += BM::EPU) {            // EPU rows (3 floats each) in flight per wait
            float pv[BM::EPU][3];
#pragma unroll
            for (int i = 0; i < BM::EPU; ++i) {
                const int e = e0 + i0 + i;
                const int ec = e < NPF && i0 + i < BM::EPP ? e : 0;
                const float* q = Pg + (ec / 9) * 28 + (ec % 9) * 3;
                pv[i][0] = q[0]; pv[i][1] = q[1]; pv[i][2] = q[2];
            }
#pragma unroll
            for (int i = 0; i < BM::EPU; ++i) {
                const int e = e0 + i0 + i;
                const double f = e < NPF && i0 + i < BM::EPP ? (double)L.pf[e] : 0.0;
                s0 += f * (double)pv[i][0]; s1 += f * (double)pv[i][1]; s2 += f * (double)pv[i][2];
            }
        }
        if (part < 3) {                                       // lane `part` adds component `part` of v_t + S beta
            double t = (double)C.mk_vt[v * 3 + part];
            const float* Sg = C.mk_S + ((size_t)v * 3 + part) * NB;
            for (int l = 0; l < NB; ++l) t += (double)Sg[l] * s.x[NPOSE + l];
            s0 += part == 0 ? t : 0.0; s1 += part == 1 ? t : 0.0; s2 += part == 2 ? t : 0.0;
        }
#pragma unroll
        for (int o = 1; o < BM::VPARTS; o <<= 1) {
            s0 += __hiloint2double(__shfl_xor(__double2hiint(s0), o, 64), __shfl_xor(__double2loint(s0), o, 64));
            s1 += __hiloint2double(__shfl_xor(__double2hiint(s1), o, 64), __shfl_xor(__double2loint(s1), o, 64));
            s2 += __hiloint2double(__shfl_xor(__double2hiint(s2), o, 64), __shfl_xor(__double2loint(s2), o, 64));
        }
        if (part == 0) { L.vp[v][0] = s0; L.vp[v][1] = s1; L.vp[v][2] = s2; }
    }
    // ---- A2 (independent of A): one 16-marker row tile per wave and product
    {
        const int fr = lane & 15, fg = lane >> 4;
        const int nrt = (M + 15) >> 4;
        constexpr int KJ = (NJ + 3) / 4;                      // K-steps over the joints
        for (int tile = wave; tile < nrt * (1 + NQ / 16); tile += BM::WAVES) {
            const int rt = tile % nrt, ct = tile / nrt;       // ct = 0: Tbd (fp64);  ct >= 1: Wq column tile ct - 1 (fp32)
            const int v = 16 * rt + fr;
            const float* Wg = C.mk_W + (size_t)(v < M ? v : 0) * NJ;
            float wreg[KJ];                                   // this lane's skinning weights W[v][4 t + fg]: all loads in flight before the first MFMA
#pragma unroll
            for (int t = 0; t < KJ; ++t) { const int j = 4 * t + fg; wreg[t] = j < NJ && v < M ? Wg[j] : 0.f; }
            if (ct == 0) {
                f64x4 d = {0.0, 0.0, 0.0, 0.0};
                const int a = fr >> 2, c = fr & 3;
#pragma unroll
                for (int t = 0; t < KJ; ++t) {
                    const int j = 4 * t + fg;
                    const bool jo = j < NJ;
                    const double av = (double)wreg[t];
                    const double bv = !jo || fr >= 12 ? 0.0 : (c < 3 ? L.Rw[j * 9 + a * 3 + c] : L.Ab[j * 3 + a]);
                    d = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, d, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {                 // D[row = fg + 4 q][col = fr]
                    const int vr = 16 * rt + fg + 4 * q;
                    if (vr < M && fr < 12) L.Tbd[vr][fr] = d[q];
                }
            } else {
                f32x4 d = {0.f, 0.f, 0.f, 0.f};
                const int n = 16 * (ct - 1) + fr, l = n / 3, a = n - 3 * l;
#pragma unroll
                for (int t = 0; t < KJ; ++t) {
                    const int j = 4 * t + fg;
                    const bool jo = j < NJ;
                    const float av = wreg[t];
                    const float bv = jo && l < NB ? L.twd[l][j][a] : 0.f;
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, d, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {                 // D[row = 4 fg + q][col = fr]
                    const int vr = 16 * rt + 4 * fg + q;
                    if (vr < M) L.Wq[vr][n] = d[q];
                }
            }
        }
    }
    __syncthreads();
    // ---- A3
    for (int it = tid; it < M * 3; it += BM::THREADS) {
        const int v = it / 3, a = it - 3 * v;
        const double* tb = L.Tbd[v] + 4 * a;
        const double xv = tb[0] * L.vp[v][0] + tb[1] * L.vp[v][1] + tb[2] * L.vp[v][2] + tb[3];
        s.resid[it] = (float)((double)s.mask[v] * ((double)s.target[it] - (xv + s.x[NPOSE + NB + 3 + a])));
    }
    if (BM::CHUNK_ROWS > 3 * MC)                                 // K-step padding rows of both chunk buffers
        for (int e = tid; e < 2 * (BM::CHUNK_ROWS - 3 * MC) * LDJS; e += BM::THREADS) {
            const int bsel = e / ((BM::CHUNK_ROWS - 3 * MC) * LDJS), r = e - bsel * (BM::CHUNK_ROWS - 3 * MC) * LDJS;
            L.Jc[bsel][3 * MC * LDJS + r] = 0.f;
        }
#if LM_TIMERS
    if (tid == 0) { const long long t1 = wall_clock64(); s.phase[5] += t1 - tp; tp = t1; }
#endif
    f64x4 acc[BM::TPW];
    int tmi[BM::TPW], tnj[BM::TPW];
#pragma unroll
    for (int t = 0; t < BM::TPW; ++t) {
        acc[t] = (f64x4){0.0, 0.0, 0.0, 0.0};
        const int tile = wave + BM::WAVES * t;
        int mi = 0;
        while ((mi + 1) * (mi + 2) / 2 <= tile) ++mi;
        tmi[t] = mi; tnj[t] = tile - mi * (mi + 1) / 2;            // nj <= mi
    }
    const int nchunk = (M + MC - 1) / MC;
    constexpr int NU = NB + 1;                                     // (marker, u) items of pass C besides the joints
    float wnext = 0.f;                                             // skinning weight of this thread's (marker, joint) item of the NEXT chunk
    if (tid < MC * NJ && grp_g * MC + tid / NJ < M) wnext = C.mk_W[(size_t)(grp_g * MC + tid / NJ) * NJ + tid % NJ];
    for (int ch = grp_g, ci = 0; ch < nchunk; ch += grp_G, ++ci) {       // this workgroup's chunks (all of them when the scan is not split)
        const int v0 = ch * MC;
        float* Jb = L.Jc[ci & 1];
#if LM_TIMERS
        if (tid == 0) tp = wall_clock64();
#endif
        // ---- B: thread (m, j)
        const int jm = tid / NJ, jj = tid - jm * NJ, jv = v0 + jm;
        const bool jlive = tid < MC * NJ && jv < M;
        float Pk[28];
        if (tid < MC * NJ) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            const float wcur = wnext;
            if (v0 + grp_G * MC + jm < M) wnext = C.mk_W[(size_t)(v0 + grp_G * MC + jm) * NJ + jj];
            if (jlive) {
                const double wj = (double)wcur;
                if (jj >= 1) {
                    const float4* P = reinterpret_cast<const float4*>(C.mk_P + ((size_t)jv * (NJ - 1) + (jj - 1)) * 28);
#pragma unroll
                    for (int e = 0; e < 7; ++e) { const float4 t4 = P[e]; Pk[4 * e] = t4.x; Pk[4 * e + 1] = t4.y; Pk[4 * e + 2] = t4.z; Pk[4 * e + 3] = t4.w; }
                }
                const double* Rj = L.Rw + jj * 9;
                const double d[3] = {L.vp[jv][0] - L.Jj[jj * 3], L.vp[jv][1] - L.Jj[jj * 3 + 1], L.vp[jv][2] - L.Jj[jj * 3 + 2]};
                o.x = (float)(wj * (Rj[0] * d[0] + Rj[1] * d[1] + Rj[2] * d[2] + L.tw[jj * 3]));
                o.y = (float)(wj * (Rj[3] * d[0] + Rj[4] * d[1] + Rj[5] * d[2] + L.tw[jj * 3 + 1]));
                o.z = (float)(wj * (Rj[6] * d[0] + Rj[7] * d[1] + Rj[8] * d[2] + L.tw[jj * 3 + 2]));
                o.w = (float)wj;
            }
            *reinterpret_cast<float4*>(L.Ay[jm][jj]) = o;
        }
        // shapedirs entries of this thread's beta item of pass C
        const int um = tid / NU, uu = tid - um * NU, uv = v0 + um;
        const bool ulive = tid < MC * NU && uv < M;
        float sv0 = 0.f, sv1 = 0.f, sv2 = 0.f;
        if (ulive && uu < nb) { const float* Sg = C.mk_S + (size_t)uv * 3 * NB + uu; sv0 = Sg[0]; sv1 = Sg[NB]; sv2 = Sg[2 * NB]; }
        __syncthreads();
#if LM_TIMERS
        if (tid == 0) { const long long t1 = wall_clock64(); s.phase[6] += t1 - tp; tp = t1; }
#endif
        // ---- C: thread (m, k) writes joint k's three rotation columns of marker m ...
        if (tid < MC * NJ) {
            const int m = jm, k = jj, v = jv;
            float* Jr = Jb + (size_t)(m * 3) * LDJS;
            const float mk = jlive ? s.mask[v] : 0.f;
            float d[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};     // [c][a]
            if (jlive) {
                const unsigned long long msk = s.sub[k];
                float u0 = 0.f, u1 = 0.f, u2 = 0.f, uw = 0.f;
                for (int j = k; j < NJ; ++j)                  // descendants have larger indices than their ancestor
                    if ((msk >> j) & 1ull) { const float4 ay = *reinterpret_cast<const float4*>(L.Ay[m][j]); u0 += ay.x; u1 += ay.y; u2 += ay.z; uw += ay.w; }
                u0 -= uw * (float)L.tw[k * 3]; u1 -= uw * (float)L.tw[k * 3 + 1]; u2 -= uw * (float)L.tw[k * 3 + 2];
                const double* Td = L.Tbd[v];
                const float T0[3] = {(float)Td[0], (float)Td[1], (float)Td[2]}, T1[3] = {(float)Td[4], (float)Td[5], (float)Td[6]},
                            T2[3] = {(float)Td[8], (float)Td[9], (float)Td[10]};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* om = L.omega[k][c];
                    d[c][0] = om[1] * u2 - om[2] * u1; d[c][1] = om[2] * u0 - om[0] * u2; d[c][2] = om[0] * u1 - om[1] * u0;
                    if (k >= 1) {
                        const float* dr = L.dR[k][c];
                        float q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
                        for (int e = 0; e < 9; ++e) { q0 += dr[e] * Pk[e * 3]; q1 += dr[e] * Pk[e * 3 + 1]; q2 += dr[e] * Pk[e * 3 + 2]; }
                        d[c][0] += T0[0] * q0 + T0[1] * q1 + T0[2] * q2;
                        d[c][1] += T1[0] * q0 + T1[1] * q1 + T1[2] * q2;
                        d[c][2] += T2[0] * q0 + T2[1] * q1 + T2[2] * q2;
                    }
                }
            }
            const int col0 = k >= 1 ? 3 * (k - 1) : NPOSE + NB;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float val = -mk * d[c][a];
                    Jr[a * LDJS + col0 + c] = val;
                    if (jac_out && jlive) jac_out[(size_t)(v * 3 + a) * DOF + col0 + c] = val;
                }
        }
        // ... and thread (m, u): u < NB: beta column l = u; u == NB: translation columns, the residual column DOF, zero padding
        if (tid < MC * NU) {
            const int m = um, u = uu, v = uv;
            float* Jr = Jb + (size_t)(m * 3) * LDJS;
            const float mk = ulive ? s.mask[v] : 0.f;
            if (u < NB) {
                float d[3] = {0.f, 0.f, 0.f};
                if (ulive && u < nb) {
                    const double* Td = L.Tbd[v];
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                        d[a] = (float)Td[4 * a] * sv0 + (float)Td[4 * a + 1] * sv1 + (float)Td[4 * a + 2] * sv2 + L.Wq[v][u * 3 + a];
                }
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float val = -mk * d[a];
                    Jr[a * LDJS + NPOSE + u] = val;
                    if (jac_out && ulive) jac_out[(size_t)(v * 3 + a) * DOF + NPOSE + u] = val;
                }
            } else {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float val = -mk * (a == c ? 1.f : 0.f);
                        Jr[a * LDJS + NPOSE + NB + 3 + c] = val;
                        if (jac_out && ulive) jac_out[(size_t)(v * 3 + a) * DOF + NPOSE + NB + 3 + c] = val;
                    }
                    Jr[a * LDJS + DOF] = ulive ? s.resid[v * 3 + a] : 0.f;
                    for (int col = DOF + 1; col < LDJ; ++col) Jr[a * LDJS + col] = 0.f;
                }
            }
        }
        __syncthreads();
#if LM_TIMERS
        if (tid == 0) { const long long t1 = wall_clock64(); s.phase[7] += t1 - tp; }
#endif
        // ---- the chunk's contribution to this wave's tiles
        {
            const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
            for (int t = 0; t < BM::TPW; ++t) {
                if (wave + BM::WAVES * t < BM::NTILES && (!grad_only || tmi[t] == BM::NT - 1)) {
                    const float* pa = Jb + fg * LDJS + 16 * tmi[t] + fr;
                    const float* pb = Jb + fg * LDJS + 16 * tnj[t] + fr;
                    constexpr int KSTEPS = BM::CHUNK_ROWS / 4, KB = KSTEPS % 8 == 0 ? 8 : (KSTEPS % 7 == 0 ? 7 : (KSTEPS % 9 == 0 ? 9 : 1));
#pragma unroll 1
                    for (int u0 = 0; u0 < KSTEPS; u0 += KB) {
                        float av[KB], bv[KB];
#pragma unroll
                        for (int u = 0; u < KB; ++u) { av[u] = pa[(u0 + u) * 4 * LDJS]; bv[u] = pb[(u0 + u) * 4 * LDJS]; }
#pragma unroll
                        for (int u = 0; u < KB; ++u) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[u], (double)bv[u], acc[t], 0, 0, 0);
                    }
                }
            }
        }
    }
#if LM_TIMERS
    if (tid == 0) { const long long t1 = wall_clock64(); s.phase[1] += t1 - t0; t0 = t1; }
#endif
    if (tid < 64) {
        float e = 0.f;                        // error metric 0.5 * |r|^2 accumulated in fp32, like the reference
        for (int i = lane; i < M * 3; i += 64) e += s.resid[i] * s.resid[i];
        e = etch_wave_sum_f32(e);
        if (lane == 0) s.err = 0.5 * (double)e;
    }
    __syncthreads();           // every read of the linearisation scratch is done: the packed matrix may overwrite it
    if (sp && grp_G > 1) {
        // exchange of the partial tiles: write mine, arrive, wait for the scan's other workgroups, add all G partials in the order g = 0..G-1
        constexpr int SLOTS = BM::WAVES * BM::TPW;
        double* base = sp->ws + (size_t)(sp->count & 1u) * grp_G * SLOTS * 256;
#pragma unroll
        for (int t = 0; t < BM::TPW; ++t) {
            *reinterpret_cast<f64x4*>(base + ((size_t)grp_g * SLOTS + wave + BM::WAVES * t) * 256 + lane * 4) = acc[t];
        }
        __syncthreads();                                           // every wave's stores are acknowledged by this XCD's L2 (workgroup-scope release)
        if (tid == 0) {
            // ONE wave runs the device-scope fences: they write back / invalidate caches shared by the whole CU / XCD, so repeating them in
            // all 8 - 12 waves only repeats that work (B = 8, G = 3, SMPL: 6.6 -> 6.3 ms per fit)
            __threadfence();                                       // release: the partial tiles are visible device-wide ...
            __hip_atomic_fetch_add(sp->ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)grp_G * (sp->count + 1u);
            unsigned spins = 0;
            bool gave_up = false;
            while (__hip_atomic_load(sp->ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(2);
                // partner workgroups never showed up (not co-resident: a concurrent kernel or a CU partition took their CUs): give up instead of
                // hanging the GPU -- and NEVER go on with tiles that were not written: the scan is flagged, every workgroup of it abandons the fit
                if (++spins > sp->spin_limit) { gave_up = true; break; }
            }
            if (gave_up) __hip_atomic_store(sp->ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (gave_up || __hip_atomic_load(sp->ctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) s.split_failed = 1;
            __threadfence();                                       // ... acquire: the others' tiles are read from memory, not from a stale cache line
        }
        __syncthreads();
        if (s.split_failed) return;                                // uniform over the workgroup (LDS flag behind the barrier)
#pragma unroll
        for (int t = 0; t < BM::TPW; ++t) {
            f64x4 tot = {0.0, 0.0, 0.0, 0.0};
            for (int g2 = 0; g2 < grp_G; ++g2) {
                // plain 32-byte loads: the agent-scope fence above has invalidated this CU's cached copies, and the buffer parity keeps a slot
                // from being rewritten before every workgroup has passed the NEXT exchange
                const f64x4 v = *reinterpret_cast<const volatile f64x4*>(base + ((size_t)g2 * SLOTS + wave + BM::WAVES * t) * 256 + lane * 4);
                tot += v;
            }
            acc[t] = tot;
        }
        if (tid == 0) ++sp->count;         // (LDS: read again only by the next linearisation, many barriers from here)
    }
    {
        // v_mfma_f64_16x16x4_f64 result layout: D[row = (lane >> 4) + 4 q][col = lane & 15]
        const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
        for (int t = 0; t < BM::TPW; ++t) {
            if (wave + BM::WAVES * t < BM::NTILES) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int gi = 16 * tmi[t] + fg + 4 * q, gj = 16 * tnj[t] + fr;
                    if (gi < DOF && gj <= gi) Apk(s.A, gi, gj) = acc[t][q];
                    else if (gi == DOF && gj < DOF) Apk(s.A, DOF, gj) = -acc[t][q];     // rhs g = -J^T r (row DOF of the packed matrix)
                    else if (gi == DOF && gj == DOF) Apk(s.A, DOF, DOF) = 1.0;
                }
            }
        }
    }
    __syncthreads();
#if LM_TIMERS
    if (tid == 0) { const long long t1 = wall_clock64(); s.phase[2] += t1 - t0; }
#endif
}

// 1 / a from v_rcp_f64 and one Newton step (full fp64 accuracy to ~1 ulp) instead of the ~10-instruction correctly rounded division:
// the pivot reciprocal sits in the serial chain of the column-by-column Cholesky
__device__ __forceinline__ double fast_rcp_f64(double a) {
    const double r = __builtin_amdgcn_rcp(a);
    return fma(fma(-a, r, 1.0), r, r);
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// delta = (J^T J + lambda I)^-1 J^T (-r) from the packed normal equations left by lm_linearize.
// Large systems (the 188-DoF model): blocked right-looking Cholesky of the packed lower triangle, 16 pivots per block, the right-hand
// side carried as the extra row DOF (its panel solves ARE the forward substitution y = L^-1 g).  Per block: (1) one wave factors the 16 x 16 diagonal block in
// registers (a lane per row, pivots broadcast with v_readlane); (2) a thread per row below solves its 16 entries against it;
// (3) the trailing matrix takes the rank-16 update on the fp64 matrix cores, one 16 x 16 tile per wave at a time.  3 barriers per
// block (18 for SMPL, 36 for the 188-DoF model) instead of one per column (85 / 188).
template <class BM>
__device__ __attribute__((LM_PHASE_INLINE)) void lm_solve(LmShared<BM>& s, double lambda) {
    constexpr int DOF = BM::DOF, N = BM::DOF, NPACK = BM::NPACK;
    constexpr int NBLK = (N + 15) / 16;                 // pivot blocks
    constexpr int NRT = (N + 1 + 15) / 16;              // row tiles of the (N + 1)-row matrix
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    long long t0 = 0;
#if LM_TIMERS
    if (tid == 0) t0 = wall_clock64();
#endif
    for (int i = tid; i < N; i += BM::THREADS) Apk(s.A, i, i) += lambda;
    __syncthreads();
    if constexpr (N <= 100) {
    // Small systems (SMPL: 85 pivots): right-looking elimination of TWO columns per barrier.  For columns k, k+1 with the pivot block
    // [m00; m10 m11]:  c = m10 / m00,  p1 = m11 - m10 c,  a'_{i,k+1} = a_{i,k+1} - a_{ik} c,  and every trailing entry takes
    //     A_ij -= a_ik a_jk / m00 + a'_{i,k+1} a'_{j,k+1} / p1
    // in one step (each thread recomputes c, 1/m00, 1/p1 from three broadcast reads: no pivot is published through LDS).  The panel
    // entries stay unscaled and un-transformed in place; three short passes at the end turn them into L (odd columns first, they read
    // their even neighbour).  43 barriers instead of 85.  Measured alternatives: one column per barrier 33.5 us, this 29 us, the blocked
    // form below 38 us (its per-block sequential parts do not amortise over 6 blocks), the matrix distributed over the workgroup's
    // REGISTERS with only the two active columns in LDS 79 us (the per-entry case analysis spills).
    constexpr int NPR = (NPACK + BM::THREADS - 1) / BM::THREADS;
    unsigned pr[NPR];
    int tid_o = tid;
    asm volatile("" : "+v"(tid_o));      // opaque copy: keeps this table from being hoisted out of the iteration loop (it would stay
                                         // live through the linearisation and spill its accumulators)
#pragma unroll
    for (int m = 0; m < NPR; ++m) {
        const int e = tid_o + BM::THREADS * m;
        int ii = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
        while ((ii + 1) * (ii + 2) / 2 <= e) ++ii;
        while (ii * (ii + 1) / 2 > e) --ii;
        pr[m] = ((unsigned)ii << 16) | (unsigned)(e - ii * (ii + 1) / 2);
    }
#if LM_ELIM_COLS == 3
    // THREE columns per barrier (29 barriers instead of 43; 8 instead of 9 LDS accesses per entry and three columns).  Pivot block
    // [m00; m10 m11; m20 m21 m22] = L D L^T with l10 = m10/d0, l20 = m20/d0, d1 = m11 - m10 l10, l21 = (m21 - m20 l10)/d1,
    // d2 = m22 - m20 l20 - (m21 - m20 l10) l21;  a row's three panel entries become a0, a1 = a_{k+1} - a0 l10, a2 = a_{k+2} - a0 l20 - a1 l21
    // and every trailing entry takes  A_ij -= a0_i a0_j / d0 + a1_i a1_j / d1 + a2_i a2_j / d2.
    static_assert(N % 3 != 2, "the remainder after the 3-column steps must be empty or a single column");
    for (int k = 0; k + 2 < N; k += 3) {
        const double m00 = Apk(s.A, k, k), m10 = Apk(s.A, k + 1, k), m11 = Apk(s.A, k + 1, k + 1);
        const double m20 = Apk(s.A, k + 2, k), m21 = Apk(s.A, k + 2, k + 1), m22 = Apk(s.A, k + 2, k + 2);
        const double r0 = fast_rcp_f64(m00), l10 = m10 * r0, l20 = m20 * r0;
        const double r1 = fast_rcp_f64(m11 - m10 * l10), t21 = m21 - m20 * l10, l21 = t21 * r1;
        const double r2 = fast_rcp_f64(m22 - m20 * l20 - t21 * l21);
        const int n = N - k - 2;                            // trailing rows k+3 .. N (incl. the rhs row)
        const int npairs = n * (n + 1) / 2;
        const int tk = (k + 3) * (k + 4) / 2;               // packed offset of row k+3
#pragma unroll
        for (int m = 0; m < NPR; ++m) {
            if (tid + BM::THREADS * m < npairs) {
                const int ii = (int)(pr[m] >> 16), jj = (int)(pr[m] & 0xFFFFu);
                const int ri = tk + ii * (ii + 1) / 2 + (k + 3) * ii, rj = tk + jj * (jj + 1) / 2 + (k + 3) * jj;   // row starts of i, j
                const double ai0 = s.A[ri + k], aj0 = s.A[rj + k];
                const double ai1 = s.A[ri + k + 1] - ai0 * l10, aj1 = s.A[rj + k + 1] - aj0 * l10;
                const double ai2 = s.A[ri + k + 2] - ai0 * l20 - ai1 * l21, aj2 = s.A[rj + k + 2] - aj0 * l20 - aj1 * l21;
                s.A[ri + k + 3 + jj] -= ai0 * aj0 * r0 + ai1 * aj1 * r1 + ai2 * aj2 * r2;
            }
        }
        __syncthreads();
    }
    // L from the unscaled panels: (1) per column triple the factors and the reciprocal roots of the pivots, (2) columns = 2, 1, 0 (mod 3)
    // in that order (a column reads its raw left neighbours)
    for (int k = 3 * tid; k < N; k += 3 * BM::THREADS) {
        const double m00 = Apk(s.A, k, k);
        s.rdiag[k] = 1.0 / sqrt(m00);
        if (k + 2 < N) {
            const double m10 = Apk(s.A, k + 1, k), m20 = Apk(s.A, k + 2, k), m21 = Apk(s.A, k + 2, k + 1);
            const double l10 = m10 / m00, l20 = m20 / m00, d1 = Apk(s.A, k + 1, k + 1) - m10 * l10, t21 = m21 - m20 * l10, l21 = t21 / d1;
            s.delta[k] = l10; s.delta[k + 1] = l20; s.delta[k + 2] = l21;      // delta is free until the back-substitution writes it
            s.rdiag[k + 1] = 1.0 / sqrt(d1);
            s.rdiag[k + 2] = 1.0 / sqrt(Apk(s.A, k + 2, k + 2) - m20 * l20 - t21 * l21);
        }
    }
    __syncthreads();
    constexpr int K3 = (N / 3) * 3;                         // columns below K3 belong to a triple
#pragma unroll 1
    for (int role = 2; role >= 0; --role) {
#pragma unroll
        for (int m = 0; m < NPR; ++m) {
            const int e = tid + BM::THREADS * m;
            const int i = (int)(pr[m] >> 16), col = (int)(pr[m] & 0xFFFFu);
            if (e < NPACK && i != col && col < N) {
                const int rr = col < K3 ? col % 3 : 0, k0 = col - rr;
                if (rr == role) {
                    double v = s.A[e];
                    if (rr == 1) v -= s.A[e - 1] * s.delta[k0];
                    if (rr == 2) { const double a0 = s.A[e - 2], a1 = s.A[e - 1] - a0 * s.delta[k0]; v -= a0 * s.delta[k0 + 1] + a1 * s.delta[k0 + 2]; }
                    s.A[e] = v * s.rdiag[col];
                }
            }
        }
        __syncthreads();
    }
#else
    for (int k = 0; k + 1 < N; k += 2) {
        const double m00 = Apk(s.A, k, k), m10 = Apk(s.A, k + 1, k), m11 = Apk(s.A, k + 1, k + 1);
        const double r0 = fast_rcp_f64(m00), c = m10 * r0, r1 = fast_rcp_f64(m11 - m10 * c);
        const int n = N - k - 1;                            // trailing rows k+2 .. N (incl. the rhs row)
        const int npairs = n * (n + 1) / 2;
        const int tk = (k + 2) * (k + 3) / 2;               // packed offset of row k+2
#pragma unroll
        for (int m = 0; m < NPR; ++m) {
            if (tid + BM::THREADS * m < npairs) {
                const int ii = (int)(pr[m] >> 16), jj = (int)(pr[m] & 0xFFFFu);
                const int ri = tk + ii * (ii + 1) / 2 + (k + 2) * ii, rj = tk + jj * (jj + 1) / 2 + (k + 2) * jj;   // row starts of i, j
                const double ai0 = s.A[ri + k], aj0 = s.A[rj + k];
                const double ai1 = s.A[ri + k + 1] - ai0 * c, aj1 = s.A[rj + k + 1] - aj0 * c;
                s.A[ri + k + 2 + jj] -= ai0 * aj0 * r0 + ai1 * aj1 * r1;
            }
        }
        __syncthreads();
    }
    // L from the unscaled panels: (1) per column pair the factor c and the reciprocal roots of both pivots, (2) odd columns, (3) even columns
    for (int k = 2 * tid; k < N; k += 2 * BM::THREADS) {
        const double m00 = Apk(s.A, k, k);
        s.rdiag[k] = 1.0 / sqrt(m00);
        if (k + 1 < N) {
            const double m10 = Apk(s.A, k + 1, k), cc = m10 / m00;
            s.delta[k >> 1] = cc;                           // delta is free until the back-substitution writes it
            s.rdiag[k + 1] = 1.0 / sqrt(Apk(s.A, k + 1, k + 1) - m10 * cc);
        }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NPR; ++m) {                         // odd columns: L_{i,k+1} = (a_{i,k+1} - a_{ik} c) / sqrt(p1)
        const int e = tid + BM::THREADS * m;
        const int i = (int)(pr[m] >> 16), col = (int)(pr[m] & 0xFFFFu);
        if (e < NPACK && i != col && col < N && (col & 1)) s.A[e] = (s.A[e] - s.A[e - 1] * s.delta[col >> 1]) * s.rdiag[col];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NPR; ++m) {                         // even columns: L_ik = a_ik / sqrt(m00)
        const int e = tid + BM::THREADS * m;
        const int i = (int)(pr[m] >> 16), col = (int)(pr[m] & 0xFFFFu);
        if (e < NPACK && i != col && col < N && !(col & 1)) s.A[e] *= s.rdiag[col];
    }
    __syncthreads();
#endif
    } else {
#pragma unroll 1
    for (int kb = 0; kb < NBLK; ++kb) {
        const int c0 = 16 * kb;
        const int w = N - c0 < 16 ? N - c0 : 16;        // pivots of this block
        const int c1 = c0 + w;
        // ---- (1) diagonal block: lane r < w holds row c0 + r; rows / columns >= w are identity padding
        if (wave == 0) {
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = (lane < w && c <= lane) ? Apk(s.A, c0 + lane, c0 + c) : (c == lane ? 1.0 : 0.0);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double dj = sqrt(readlane_f64(a[j], j));
                const double lj = lane == j ? dj : a[j] * (1.0 / dj);       // column j of L (lanes >= j)
                a[j] = lj;
#pragma unroll
                for (int c = j + 1; c < 16; ++c) {
                    const double lc = readlane_f64(lj, c);
                    a[c] -= lane >= c ? lj * lc : 0.0;
                }
            }
            if (lane < w) {
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (c <= lane) Apk(s.A, c0 + lane, c0 + c) = a[c];
                double dl = a[0];
#pragma unroll
                for (int c = 1; c < 16; ++c) dl = lane == c ? a[c] : dl;
                s.rdiag[c0 + lane] = 1.0 / dl;
            }
        }
        __syncthreads();
        // ---- (2) panel: rows c1 .. N (the last one is the right-hand side), a thread per row
        for (int i = c1 + tid; i <= N; i += BM::THREADS) {
            double x[16];
            double* row = s.A + i * (i + 1) / 2 + c0;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (c < w) {
                    double v = row[c];
                    const double* lrow = s.A + (c0 + c) * (c0 + c + 1) / 2 + c0;
#pragma unroll
                    for (int k = 0; k < c; ++k) v -= x[k] * lrow[k];
                    x[c] = v * s.rdiag[c0 + c];
                    row[c] = x[c];
                } else x[c] = 0.0;
            }
        }
        __syncthreads();
        // ---- (3) trailing update A_ij -= sum_k L_ik L_jk over the block's 16 columns, 16 x 16 tiles (ti >= tj > kb) on v_mfma_f64_16x16x4_f64
        if (w == 16) {
            const int T = NRT - (kb + 1);               // trailing row tiles
            const int ntile = T * (T + 1) / 2;
            const int fr = lane & 15, fg = lane >> 4;
            for (int tile = wave; tile < ntile; tile += BM::WAVES) {
                int ti = 0;
                while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
                const int tj = tile - ti * (ti + 1) / 2;
                const int ri = 16 * (kb + 1 + ti) + fr, rj = 16 * (kb + 1 + tj) + fr;          // operand rows of this lane
                const bool vi = ri <= N, vj = rj <= N;
                const double* pi = s.A + (vi ? ri : N) * ((vi ? ri : N) + 1) / 2 + c0 + fg;
                const double* pj = s.A + (vj ? rj : N) * ((vj ? rj : N) + 1) / 2 + c0 + fg;
                f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const double av = vi ? pi[4 * t] : 0.0, bv = vj ? pj[4 * t] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
                // D[row = fg + 4 q][col = fr]
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int gi = 16 * (kb + 1 + ti) + fg + 4 * q, gj = 16 * (kb + 1 + tj) + fr;
                    if (gi <= N && gj < N && gj <= gi) Apk(s.A, gi, gj) -= acc[q];
                }
            }
        }
        __syncthreads();
    }
    }
#if LM_TIMERS
    if (tid == 0) { const long long t1 = wall_clock64(); s.phase[3] += t1 - t0; t0 = t1; }
#endif
    // back substitution L^T delta = y by one wave (lane owns rows lane, lane + 64, ...); pivots broadcast with v_readlane.  The L
    // entries of 4 pivots are loaded BEFORE their dependent chain (delta_i -> y -> delta_{i-1} ...), so the chain itself is
    // register-only: ~40 cycles per pivot instead of an LDS round trip.
    if (tid < 64) {
        constexpr int NR = (DOF + 63) / 64;
        double y[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) y[r] = lane + 64 * r < DOF ? Apk(s.A, DOF, lane + 64 * r) : 0.0;
        for (int i0 = DOF - 1; i0 >= 0; i0 -= 4) {
            double l[4][NR], rd[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 - u;
                rd[u] = i >= 0 ? s.rdiag[i] : 0.0;
#pragma unroll
                for (int r = 0; r < NR; ++r) { const int row = lane + 64 * r; l[u][r] = i >= 0 && row < i ? Apk(s.A, i, row) : 0.0; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 - u;
                if (i >= 0) {                                     // wave-uniform
                    double src = y[0];
#pragma unroll
                    for (int r = 1; r < NR; ++r) src = (i >> 6) == r ? y[r] : src;
                    const double di = readlane_f64(src, i & 63) * rd[u];
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const int row = lane + 64 * r;
                        y[r] = row == i ? di : y[r] - l[u][r] * di;      // rows > i: l = 0, their final values stay
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (lane + 64 * r < DOF) s.delta[lane + 64 * r] = y[r];
    }
    __syncthreads();
#if LM_TIMERS
    if (tid == 0) { const long long t1 = wall_clock64(); s.phase[4] += t1 - t0; }
#endif
}

template <class BM>
__global__ void __launch_bounds__(BM::THREADS) smpl_lm_fit_kernel(SmplConsts C, int M, const float* __restrict__ markers,
                                                                const float* __restrict__ valid, int it0, float step0, float damp0,
                                                                int it1, float step1, float damp1, float* __restrict__ x_out,
                                                                float* __restrict__ x_stage0, float* __restrict__ err_trace, long long* __restrict__ phase_out,
                                                                int G, double* __restrict__ split_ws, unsigned* __restrict__ split_ctr,
                                                                unsigned spin_limit, int drop_group) {
    constexpr int DOF = BM::DOF;
    extern __shared__ __attribute__((aligned(16))) unsigned char lm_smem[];
    LmShared<BM>& s = *reinterpret_cast<LmShared<BM>*>(lm_smem);
    const int b = blockIdx.x / G, tid = threadIdx.x;
    const bool lead = blockIdx.x % G == 0;                      // the scan's workgroup that writes the results (all G hold the same state)
    if (G > 1 && (int)(blockIdx.x % G) == drop_group) return;   // test hook (etch_smpl_lm_debug): a partner that never becomes resident
    if (tid == 0) {
        s.split = LmSplit{(int)(blockIdx.x % G), G, split_ws + (size_t)b * 2 * G * BM::WAVES * BM::TPW * 256, split_ctr + (size_t)b * 64, 0u, spin_limit};
        s.split_failed = 0;
    }
    if (!lead) { x_stage0 = nullptr; err_trace = nullptr; phase_out = nullptr; }
    for (int i = tid; i < DOF; i += BM::THREADS) s.x[i] = 0.0;
    lm_setup(s, C, M, markers + (size_t)b * M * 3, valid + (size_t)b * M);
    int trace_pos = 0;
    for (int stage = 0; stage < 2; ++stage) {
        const int iters = stage == 0 ? it0 : it1;
        const int nb = stage == 0 ? (BM::NB < NB_STAGE0 ? BM::NB : NB_STAGE0) : BM::NB;
        const double step = stage == 0 ? (double)step0 : (double)step1;
        const double lambda = stage == 0 ? (double)damp0 : (double)damp1;
        float last = 0.f;
        bool conv = false;
        for (int it = -1; it < iters; ++it) {          // it = -1: the linearisation at the stage's starting point (error only)
            if (!conv) {
                if (it >= 0) {
                    lm_solve(s, lambda);
                    for (int i = tid; i < DOF; i += BM::THREADS) s.x[i] += step * s.delta[i];
                    __syncthreads();
                }
                lm_linearize(s, M, nb, nullptr, false, G > 1);
                if (G > 1 && s.split_failed) break;            // abandoned (uniform): fall through to the NaN write-out
                const float err = (float)s.err;
                if (it >= 0) {
                    const float a = fabsf(last - err);
                    conv = (a < 1e-10f) || (a / last < 1e-8f);
                }
                last = err;
            }
            if (err_trace && tid == 0) err_trace[(size_t)b * (it0 + it1 + 2) + trace_pos] = last;
            ++trace_pos;
        }
        if (G > 1 && s.split_failed) break;
        if (stage == 0 && x_stage0)
            for (int i = tid; i < DOF; i += BM::THREADS) x_stage0[(size_t)b * DOF + i] = (float)s.x[i];
        __syncthreads();
    }
    if (G > 1 && s.split_failed) {
        // loud failure: the scan's parameters (and its error trace) are NaN, never a fit of partial sums; the flag word stays set in the workspace
        // (etch_smpl_lm_split_failed reads it).  Every workgroup of the scan ends up here: a late one sees the flag at its first exchange.
        const float qnan = __uint_as_float(0x7fc00000u);
        if (lead) {
            for (int i = tid; i < DOF; i += BM::THREADS) { x_out[(size_t)b * DOF + i] = qnan; if (x_stage0) x_stage0[(size_t)b * DOF + i] = qnan; }
            if (err_trace) for (int i = tid; i < it0 + it1 + 2; i += BM::THREADS) err_trace[(size_t)b * (it0 + it1 + 2) + i] = qnan;
        }
        return;
    }
    if (lead)
        for (int i = tid; i < DOF; i += BM::THREADS) x_out[(size_t)b * DOF + i] = (float)s.x[i];
    if (phase_out && tid < 8) phase_out[(size_t)b * 8 + tid] = s.phase[tid];
}


// ---------------------------------------------------------------------------------------------- first-order fitter
// fit_smpl of /root/reference/src/models/fit_SMPL_Adam.py:68-225: torch.optim.Adam (betas 0.9 / 0.999, eps 1e-8, bias-corrected) on
// L = mse_loss(markers(x)[valid], target[valid]) -- the mean over ALL valid marker coordinates of the WHOLE batch (:142, :199), so
// the only coupling between scans is the common factor 1 / n_valid_total.  Stage 0: it0 steps on pose, betas[:2], orient, transl
// (:104-120); stage 1: it1 steps with a FRESH optimizer state on all betas (:166-183).  grad L = (2 / n) J^T r comes out of the same
// marker-restricted linearisation as the LM fit (last tile row of the matrix-core accumulation) instead of autograd through the
// full mesh.  x_last = parameters of the LAST forward pass, i.e. before the final optimizer step: the reference builds its output
// meshes from that forward (:221-225).
// Deliberate divergence: the reference's loss is ONE mean over the batch, so a scan whose valid marker is NaN (conf**20 underflow, flagged by
// etch_marker_status) turns the loss -- and through Adam's moments every scan's parameters -- into NaN.  Here the scans only share the
// factor 1 / n_valid_total; a NaN stays in its own scan (its status word says so) and the others are fitted as if it were absent from the
// gradient but present in n.  Finite inputs give the reference's numbers.
template <class BM>
__global__ void __launch_bounds__(BM::THREADS) smpl_adam_fit_kernel(SmplConsts C, int M, int B, const float* __restrict__ markers,
                                                                  const float* __restrict__ valid, int it0, int it1, float lr, float beta1,
                                                                  float beta2, float eps, float* __restrict__ x_out, float* __restrict__ x_last,
                                                                  float* __restrict__ loss_trace) {
    constexpr int DOF = BM::DOF;
    extern __shared__ __attribute__((aligned(16))) unsigned char lm_smem[];
    LmShared<BM>& s = *reinterpret_cast<LmShared<BM>*>(lm_smem);
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < DOF; i += BM::THREADS) s.x[i] = 0.0;
    lm_setup(s, C, M, markers + (size_t)b * M * 3, valid + (size_t)b * M);
    // n = number of valid marker coordinates of the whole batch (every workgroup counts them: B * M flags)
    if (tid < 64) {
        float cnt = 0.f;
        for (int i = tid; i < B * M; i += 64) cnt += valid[i] != 0.f ? 3.f : 0.f;
        cnt = etch_wave_sum_f32(cnt);
        if (tid == 0) s.rpiv = (double)cnt;
    }
    __syncthreads();
    const double ninv = s.rpiv > 0.0 ? 1.0 / s.rpiv : 0.0;
    double* mom = s.delta;               // first / second moment estimates (the LM solve's vectors are free here)
    double* var = s.rdiag;
    int trace_pos = 0;
    for (int stage = 0; stage < 2; ++stage) {
        const int iters = stage == 0 ? it0 : it1;
        const int nb = stage == 0 ? (BM::NB < NB_STAGE0 ? BM::NB : NB_STAGE0) : BM::NB;
        for (int i = tid; i < DOF; i += BM::THREADS) { mom[i] = 0.0; var[i] = 0.0; }     // a fresh torch.optim.Adam per stage
        __syncthreads();
        double b1t = 1.0, b2t = 1.0;
        for (int it = 0; it < iters; ++it) {
            // the parameters of the LAST forward pass of the whole schedule: the last stage-1 iteration, or -- when stage 1 is empty --
            // the last stage-0 iteration (the reference's `verts` then still hold that forward, fit_SMPL_Adam.py:221-225)
            if (x_last && it == iters - 1 && (stage == 1 || it1 == 0))
                for (int i = tid; i < DOF; i += BM::THREADS) x_last[(size_t)b * DOF + i] = (float)s.x[i];
            lm_linearize(s, M, nb, nullptr, true);
            if (loss_trace && tid == 0) loss_trace[(size_t)b * (it0 + it1) + trace_pos] = (float)(2.0 * s.err * ninv);   // this scan's share of L
            ++trace_pos;
            b1t *= (double)beta1; b2t *= (double)beta2;
            for (int i = tid; i < DOF; i += BM::THREADS) {
                const double g = -2.0 * ninv * Apk(s.A, DOF, i);           // row DOF of the packed matrix holds -J^T r
                const double m1 = (double)beta1 * mom[i] + (1.0 - (double)beta1) * g;
                const double v1 = (double)beta2 * var[i] + (1.0 - (double)beta2) * g * g;
                mom[i] = m1; var[i] = v1;
                s.x[i] -= ((double)lr / (1.0 - b1t)) * m1 / (sqrt(v1) / sqrt(1.0 - b2t) + (double)eps);
            }
            __syncthreads();
        }
    }
    if (it0 == 0 && it1 == 0 && x_last)
        for (int i = tid; i < DOF; i += BM::THREADS) x_last[(size_t)b * DOF + i] = (float)s.x[i];
    for (int i = tid; i < DOF; i += BM::THREADS) x_out[(size_t)b * DOF + i] = (float)s.x[i];
}

// ---------------------------------------------------------------------------------------------- diagnostics (tests)
// One linearisation of the LM kernel at a caller-given x: residual (B,3M), the analytic Jacobian (B,3M,DOF) and the normal
// equations exactly as the fit forms them (same device function, same LDS state) -- lets a test compare the analytic
// marker-restricted Jacobian with autograd through the full-mesh LBS (the reference's AutoDiffCostFunction formulation,
// fit_SMPL.py:176-183) and the matrix-core J^T J / J^T r with their fp64 definition.
template <class BM>
__global__ void __launch_bounds__(BM::THREADS) smpl_lm_linearize_kernel(SmplConsts C, int M, int nb, const float* __restrict__ x_in,
                                                                      const float* __restrict__ markers, const float* __restrict__ valid,
                                                                      float* __restrict__ resid, float* __restrict__ jac, double* __restrict__ normal) {
    constexpr int DOF = BM::DOF;
    extern __shared__ __attribute__((aligned(16))) unsigned char lm_smem[];
    LmShared<BM>& s = *reinterpret_cast<LmShared<BM>*>(lm_smem);
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < DOF; i += BM::THREADS) s.x[i] = (double)x_in[(size_t)b * DOF + i];
    lm_setup(s, C, M, markers + (size_t)b * M * 3, valid + (size_t)b * M);
    lm_linearize(s, M, nb, jac + (size_t)b * M * 3 * DOF);
    for (int i = tid; i < M * 3; i += BM::THREADS) resid[(size_t)b * M * 3 + i] = s.resid[i];
    if (normal)                 // (DOF+1) x (DOF+1) lower triangle, row DOF = -J^T r
        for (int e = tid; e < (DOF + 1) * (DOF + 1); e += BM::THREADS) {
            const int i = e / (DOF + 1), j = e - i * (DOF + 1);
            normal[(size_t)b * (DOF + 1) * (DOF + 1) + e] = j <= i ? Apk(s.A, i, j) : 0.0;
        }
}

// rodrigues_d of the LM / LBS kernels on n rotation vectors: R (n,9) fp64 and dR/dtheta_q (n,3,9) -- pinned by the golden
// emitted from the in-tree batch_rodrigues (src/data_utils/GT_dataloader_mixed.py:29-64).
__global__ void __launch_bounds__(64) rodrigues_kernel(int n, const float* __restrict__ theta, double* __restrict__ R, float* __restrict__ dR) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const double th[3] = {(double)theta[i * 3], (double)theta[i * 3 + 1], (double)theta[i * 3 + 2]};
    double Rl[9];
    float d[3][9];
    rodrigues_d(th, Rl, d, true);
    for (int e = 0; e < 9; ++e) R[(size_t)i * 9 + e] = Rl[e];
    for (int q = 0; q < 3; ++q)
        for (int e = 0; e < 9; ++e) dR[((size_t)i * 3 + q) * 9 + e] = d[q][e];
}

// ---------------------------------------------------------------------------------------------- full-mesh LBS
#define LBS_MAXJ 55
#define LBS_MAXB 20
struct LbsConsts {
    const float* v_template;   // [V][3]
    const float* shapedirs;    // [V][3][NB]
    const float* posedirs;     // [9(NJ-1)][V*3]
    const float* weights;      // [V][NJ]
    const float* J0; const float* Jd; const int* parents;
    const int* extra_vids; int n_extra;
};

__global__ void __launch_bounds__(256) smpl_lbs_kernel(LbsConsts C, int NJ, int NB, int V, const float* __restrict__ x, float* __restrict__ verts,
                                                       float* __restrict__ joints) {
    __shared__ double R[LBS_MAXJ * 9], Rw[LBS_MAXJ * 9], tw[LBS_MAXJ * 3], Jj[LBS_MAXJ * 3], xs[3 * LBS_MAXJ + LBS_MAXB + 3];
    __shared__ float A[LBS_MAXJ][12], pf[9 * (LBS_MAXJ - 1)];
    __shared__ int parents[LBS_MAXJ];
    const int NPOSE = 3 * (NJ - 1), NPF = 9 * (NJ - 1), DOF = NPOSE + NB + 6;
    const int b = blockIdx.y, tid = threadIdx.x;
    if (tid < DOF) xs[tid] = (double)x[(size_t)b * DOF + tid];
    if (tid < NJ) parents[tid] = C.parents[tid];
    __syncthreads();
    if (tid < NJ) {
        double th[3];
        joint_theta(xs, tid, NPOSE, NB, th);
        float dummy[3][9];
        rodrigues_d(th, R + tid * 9, dummy, false);
        for (int c = 0; c < 3; ++c) {
            double v = C.J0[tid * 3 + c];
            for (int l = 0; l < NB; ++l) v += (double)C.Jd[(tid * 3 + c) * NB + l] * xs[NPOSE + l];
            Jj[tid * 3 + c] = v;
        }
    }
    __syncthreads();
    if (tid == 0) fk_chain(NJ, parents, R, Jj, Rw, tw);
    for (int e = tid; e < NPF; e += 256) { const int k = 1 + e / 9, q = e % 9; pf[e] = (float)(R[k * 9 + q] - ((q % 4 == 0) ? 1.0 : 0.0)); }
    __syncthreads();
    if (tid < NJ) {                            // A_j = [Rw_j | tw_j - Rw_j J_j]
        for (int a = 0; a < 3; ++a) {
            double t = tw[tid * 3 + a];
            for (int c = 0; c < 3; ++c) { A[tid][a * 4 + c] = (float)Rw[tid * 9 + a * 3 + c]; t -= Rw[tid * 9 + a * 3 + c] * Jj[tid * 3 + c]; }
            A[tid][a * 4 + 3] = (float)t;
        }
        if (blockIdx.x == 0)
            for (int a = 0; a < 3; ++a) joints[((size_t)b * (NJ + C.n_extra) + tid) * 3 + a] = (float)(tw[tid * 3 + a] + xs[NPOSE + NB + 3 + a]);
    }
    __syncthreads();
    const int v = blockIdx.x * 256 + tid;
    if (v >= V) return;
    float vp[3];
    for (int c = 0; c < 3; ++c) {
        float acc = C.v_template[v * 3 + c];
        for (int l = 0; l < NB; ++l) acc += C.shapedirs[((size_t)v * 3 + c) * NB + l] * (float)xs[NPOSE + l];
        vp[c] = acc;
    }
    for (int e = 0; e < NPF; ++e) {
        const float* P = C.posedirs + (size_t)e * V * 3 + v * 3;
        vp[0] += pf[e] * P[0]; vp[1] += pf[e] * P[1]; vp[2] += pf[e] * P[2];
    }
    float T[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < NJ; ++j) {
        const float w = C.weights[(size_t)v * NJ + j];
        if (w != 0.f)
            for (int i = 0; i < 12; ++i) T[i] += w * A[j][i];
    }
    float o[3];
    for (int a = 0; a < 3; ++a) o[a] = T[a * 4] * vp[0] + T[a * 4 + 1] * vp[1] + T[a * 4 + 2] * vp[2] + T[a * 4 + 3] + (float)xs[NPOSE + NB + 3 + a];
    float* out = verts + ((size_t)b * V + v) * 3;
    out[0] = o[0]; out[1] = o[1]; out[2] = o[2];
    for (int e = 0; e < C.n_extra; ++e)
        if (C.extra_vids[e] == v) {
            float* jo = joints + ((size_t)b * (NJ + C.n_extra) + NJ + e) * 3;
            jo[0] = o[0]; jo[1] = o[1]; jo[2] = o[2];
        }
}

// ---------------------------------------------------------------------------------------------- markers
__device__ __forceinline__ unsigned sortable_f32(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// One wave per (scan, label): top-3 confidences among the points carrying that label (ties -> lower index),
// weights conf^20, weighted centre.  labels int64 (argmax output), conf (B,K,1).
__global__ void __launch_bounds__(64) get_markers_kernel(int K, int M, const float* __restrict__ pts, const long long* __restrict__ labels,
                                                         const float* __restrict__ conf, float* __restrict__ markers,
                                                         float* __restrict__ valid_f, unsigned char* __restrict__ valid_b) {
    const int b = blockIdx.y, m = blockIdx.x, lane = threadIdx.x;
    const long long* lb = labels + (size_t)b * K;
    const float* cf = conf + (size_t)b * K;
    unsigned long long top[3] = {0ull, 0ull, 0ull};
    int cnt = 0;
    for (int i = lane; i < K; i += 64) {
        if (lb[i] == (long long)m) {
            ++cnt;
            unsigned long long key = ((unsigned long long)sortable_f32(cf[i]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
            if (key > top[2]) {
                if (key > top[1]) {
                    top[2] = top[1];
                    if (key > top[0]) { top[1] = top[0]; top[0] = key; } else top[1] = key;
                } else top[2] = key;
            }
        }
    }
    int total = cnt;
    for (int off = 32; off >= 1; off >>= 1) total += __shfl_xor(total, off, 64);
    int sel[3] = {-1, -1, -1};
    for (int r = 0; r < 3; ++r) {
        const unsigned long long best = etch_wave_max_u64(top[0]);
        if (best != 0ull) {
            sel[r] = (int)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull));
            if (top[0] == best) { top[0] = top[1]; top[1] = top[2]; top[2] = 0ull; }
        }
    }
    if (lane == 0) {
        const size_t o = (size_t)b * M + m;
        float c[3] = {0.f, 0.f, 0.f};
        if (total > 0) {
            float wsum = 0.f, acc[3] = {0.f, 0.f, 0.f};
            const int tk = total < 3 ? total : 3;
            for (int r = 0; r < tk; ++r) {
                const float w = powf(cf[sel[r]], 20.0f);
                const float* p = pts + ((size_t)b * K + sel[r]) * 3;
                acc[0] += p[0] * w; acc[1] += p[1] * w; acc[2] += p[2] * w;
                wsum += w;
            }
            c[0] = acc[0] / wsum; c[1] = acc[1] / wsum; c[2] = acc[2] / wsum;
        }
        markers[o * 3] = c[0]; markers[o * 3 + 1] = c[1]; markers[o * 3 + 2] = c[2];
        if (valid_f) valid_f[o] = total > 0 ? 1.f : 0.f;
        if (valid_b) valid_b[o] = total > 0 ? 1 : 0;
    }
}

// scan_status[b]: bit 0 = some valid marker of the scan is non-finite (conf**20 underflowed to 0 for every one of a label's top-3
// points: 0/0, exactly as fit_SMPL.py:52-57 computes it) -> the fit of that scan is NaN; bit 1 = the scan has no valid marker.
__global__ void __launch_bounds__(64) marker_status_kernel(int M, const float* __restrict__ markers, const float* __restrict__ valid_f,
                                                           int* __restrict__ status) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int bad = 0, any = 0;
    for (int m = lane; m < M; m += 64) {
        if (valid_f[(size_t)b * M + m] != 0.f) {
            any = 1;
            const float* p = markers + ((size_t)b * M + m) * 3;
            if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) bad = 1;
        }
    }
    bad = __any(bad); any = __any(any);
    if (lane == 0) status[b] = (bad ? 1 : 0) | (any ? 0 : 2);
}

// labels[r] = argmax_g logits[r, g] (first maximum), int64 like torch.max
__global__ void __launch_bounds__(256) argmax_rows_kernel(long R, int G, const float* __restrict__ logits, long long* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < R; r += (long)gridDim.x * 4) {
        unsigned long long best = 0ull;
        for (int g = lane; g < G; g += 64) {
            const unsigned long long key = ((unsigned long long)sortable_f32(logits[r * G + g]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)g);
            best = key > best ? key : best;
        }
        best = etch_wave_max_u64(best);
        if (lane == 0) out[r] = (long long)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull));
    }
}

// inner = points - (direction * magnitude) / scale      (src/inference_demo.py:58-59)
__global__ void __launch_bounds__(256) inner_points_kernel(long n, const float* __restrict__ pts, const float* __restrict__ dir,
                                                           const float* __restrict__ mag, float scale, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n * 3; i += (long)gridDim.x * 256) {
#pragma clang fp contract(off)
        out[i] = pts[i] - (dir[i] * mag[i / 3]) / scale;
    }
}

// ---------------------------------------------------------------------------------------------- host side
typedef Body<24, 10, 768> BodySMPL;       // the reference's model (fit_SMPL.py:100)
typedef Body<55, 20, 512> BodySMPLX;      // SMPL-X-sized: 55 joints, 10 shape + 10 expression coefficients (BASELINE configs[4])

static inline SmplConsts lm_consts(const void* const* consts) {
    return SmplConsts{(const float*)consts[0], (const float*)consts[1], (const int*)consts[2], (const float*)consts[3], (const float*)consts[4],
                      (const float*)consts[5], (const float*)consts[6]};
}

template <class BM>
static size_t lm_split_bytes(int B, int G) {       // arrival counters (one 256-byte line per scan) + the double-buffered partial tiles
    return (size_t)B * 256 + (size_t)B * 2 * G * BM::WAVES * BM::TPW * 256 * sizeof(double);
}

static unsigned g_lm_spin_limit = 1u << 27;      // polls (s_sleep 2 each) before a split workgroup gives up: ~10 s
static int g_lm_drop_group = -1;                 // test hook: this group index of every scan returns at once (its partners must time out)

template <class BM>
static int launch_lm_fit(int B, int M, const void* const* consts, const float* markers, const float* valid, int it0, float step0, float damp0,
                         int it1, float step1, float damp1, float* x_out, float* x_stage0, float* err_trace, long long* phase_ticks, int G,
                         void* workspace, hipStream_t st) {
    const int lds = (int)sizeof(LmShared<BM>);
    static_assert(sizeof(LmShared<BM>) <= 160 * 1024, "LM state must fit the 160 KB LDS of a gfx950 CU");
    hipError_t e = hipFuncSetAttribute((const void*)smpl_lm_fit_kernel<BM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    unsigned* ctr = nullptr;
    double* ws = nullptr;
    if (G > 1) {
        // every workgroup of a scan must be resident at the same time (they wait for each other): one workgroup per CU (LDS), so B * G <= CUs
        if (!workspace || (long)B * G > etch_cu_count()) return ETCH_EUNSUPPORTED;
        ctr = reinterpret_cast<unsigned*>(workspace);
        ws = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(workspace) + (size_t)B * 256);
        e = hipMemsetAsync(workspace, 0, (size_t)B * 256, st);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(smpl_lm_fit_kernel<BM>, dim3(B * G), dim3(BM::THREADS), lds, st, lm_consts(consts), M, markers, valid, it0, step0, damp0, it1,
                       step1, damp1, x_out, x_stage0, err_trace, phase_ticks, G, ws, reinterpret_cast<unsigned*>(ctr ? ctr : nullptr), g_lm_spin_limit,
                       g_lm_drop_group);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

template <class BM>
static int launch_lm_linearize(int B, int M, int nb, const void* const* consts, const float* x, const float* markers, const float* valid,
                               float* resid, float* jac, double* normal, hipStream_t st) {
    if (nb < 0 || nb > BM::NB) return ETCH_EINVAL;
    const int lds = (int)sizeof(LmShared<BM>);
    hipError_t e = hipFuncSetAttribute((const void*)smpl_lm_linearize_kernel<BM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(smpl_lm_linearize_kernel<BM>, dim3(B), dim3(BM::THREADS), lds, st, lm_consts(consts), M, nb, x, markers, valid, resid, jac, normal);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

template <class BM>
static int launch_adam_fit(int B, int M, const void* const* consts, const float* markers, const float* valid, int it0, int it1, float lr,
                           float beta1, float beta2, float eps, float* x_out, float* x_last, float* loss_trace, hipStream_t st) {
    const int lds = (int)sizeof(LmShared<BM>);
    hipError_t e = hipFuncSetAttribute((const void*)smpl_adam_fit_kernel<BM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(smpl_adam_fit_kernel<BM>, dim3(B), dim3(BM::THREADS), lds, st, lm_consts(consts), M, B, markers, valid, it0, it1, lr, beta1,
                       beta2, eps, x_out, x_last, loss_trace);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

extern "C" {

int etch_inner_points(long n, const float* pts, const float* dir, const float* mag, float scale, float* out, void* stream) {
    if (n <= 0) return ETCH_OK;
    long blocks = (n * 3 + 255) / 256;
    if (blocks > 65535 * 8) blocks = 65535 * 8;
    hipLaunchKernelGGL(inner_points_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, pts, dir, mag, scale, out);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_argmax_rows(long R, int G, const float* logits, long long* out, void* stream) {
    if (R <= 0) return ETCH_OK;
    long blocks = (R + 3) / 4;
    if (blocks > 65535 * 8) blocks = 65535 * 8;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, R, G, logits, out);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_get_markers(int B, int K, int M, const float* pts, const long long* labels, const float* conf, float* markers,
                     float* valid_f, unsigned char* valid_b, void* stream) {
    if (B <= 0 || M <= 0) return ETCH_OK;
    hipLaunchKernelGGL(get_markers_kernel, dim3(M, B), dim3(64), 0, (hipStream_t)stream, K, M, pts, labels, conf, markers, valid_f, valid_b);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_marker_status(int B, int M, const float* markers, const float* valid_f, int* status, void* stream) {
    if (B <= 0) return ETCH_OK;
    if (M <= 0) return ETCH_EINVAL;
    hipLaunchKernelGGL(marker_status_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, M, markers, valid_f, status);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

int etch_smpl_lm_workspace_bytes(int nj, int nb) {
    if (nj == 24 && nb == 10) return (int)sizeof(LmShared<BodySMPL>);
    if (nj == 55 && nb == 20) return (int)sizeof(LmShared<BodySMPLX>);
    return ETCH_EUNSUPPORTED;
}

// consts: 7 device pointers {J0, Jd, parents, mk_vt, mk_S, mk_P, mk_W}
int etch_smpl_lm_fit_split(int B, int M, int nj, int nb, const void* const* consts, const float* markers, const float* valid, int it0, float step0,
                           float damp0, int it1, float step1, float damp1, float* x_out, float* x_stage0, float* err_trace, long long* phase_ticks,
                           int G, void* workspace, void* stream) {
    if (B <= 0) return ETCH_OK;
    if (M <= 0 || M > LM_MAXM || G < 1 || G > 8) return ETCH_EUNSUPPORTED;
    if (nj == 24 && nb == 10)
        return launch_lm_fit<BodySMPL>(B, M, consts, markers, valid, it0, step0, damp0, it1, step1, damp1, x_out, x_stage0, err_trace, phase_ticks, G,
                                       workspace, (hipStream_t)stream);
    if (nj == 55 && nb == 20)
        return launch_lm_fit<BodySMPLX>(B, M, consts, markers, valid, it0, step0, damp0, it1, step1, damp1, x_out, x_stage0, err_trace, phase_ticks, G,
                                        workspace, (hipStream_t)stream);
    return ETCH_EUNSUPPORTED;
}

// Test hook of the split fit's failure path: spin_limit = polls before a workgroup gives up (0: the default, ~10 s); drop_group >= 0: that
// workgroup of every scan returns at launch, so its partners time out, flag the scan and return NaN parameters.  Process-wide; call with
// (0, -1) to restore the defaults.
int etch_smpl_lm_debug(unsigned spin_limit, int drop_group) {
    g_lm_spin_limit = spin_limit ? spin_limit : (1u << 27);
    g_lm_drop_group = drop_group;
    return ETCH_OK;
}

// Number of scans whose split fit was abandoned (their x_out rows are NaN); synchronises the stream.  workspace = the one handed to the fit.
int etch_smpl_lm_split_failed(int B, const void* workspace, int* n_failed, void* stream) {
    if (!n_failed) return ETCH_EINVAL;
    *n_failed = 0;
    if (B <= 0 || !workspace) return ETCH_OK;
    unsigned char* host = (unsigned char*)malloc((size_t)B * 256);
    if (!host) return ETCH_EINVAL;
    hipError_t e = hipMemcpyAsync(host, workspace, (size_t)B * 256, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e == hipSuccess)
        for (int b = 0; b < B; ++b) *n_failed += reinterpret_cast<const unsigned*>(host + (size_t)b * 256)[1] != 0u;
    free(host);
    return e == hipSuccess ? ETCH_OK : (int)e;
}

long etch_smpl_lm_split_workspace_bytes(int B, int nj, int nb, int G) {
    if (B <= 0 || G <= 1) return 0;
    if (nj == 24 && nb == 10) return (long)lm_split_bytes<BodySMPL>(B, G);
    if (nj == 55 && nb == 20) return (long)lm_split_bytes<BodySMPLX>(B, G);
    return ETCH_EUNSUPPORTED;
}

int etch_smpl_lm_fit(int B, int M, int nj, int nb, const void* const* consts, const float* markers, const float* valid, int it0, float step0,
                     float damp0, int it1, float step1, float damp1, float* x_out, float* x_stage0, float* err_trace, long long* phase_ticks, void* stream) {
    return etch_smpl_lm_fit_split(B, M, nj, nb, consts, markers, valid, it0, step0, damp0, it1, step1, damp1, x_out, x_stage0, err_trace, phase_ticks, 1,
                                  nullptr, stream);
}

int etch_smpl_adam_fit(int B, int M, int nj, int nb, const void* const* consts, const float* markers, const float* valid, int it0, int it1,
                       float lr, float beta1, float beta2, float eps, float* x_out, float* x_last, float* loss_trace, void* stream) {
    if (B <= 0) return ETCH_OK;
    if (M <= 0 || M > LM_MAXM) return ETCH_EUNSUPPORTED;
    if (nj == 24 && nb == 10)
        return launch_adam_fit<BodySMPL>(B, M, consts, markers, valid, it0, it1, lr, beta1, beta2, eps, x_out, x_last, loss_trace, (hipStream_t)stream);
    if (nj == 55 && nb == 20)
        return launch_adam_fit<BodySMPLX>(B, M, consts, markers, valid, it0, it1, lr, beta1, beta2, eps, x_out, x_last, loss_trace, (hipStream_t)stream);
    return ETCH_EUNSUPPORTED;
}

int etch_smpl_lm_linearize(int B, int M, int nj, int nb_model, int nb_active, const void* const* consts, const float* x, const float* markers,
                           const float* valid, float* resid, float* jac, double* normal, void* stream) {
    if (B <= 0) return ETCH_OK;
    if (M <= 0 || M > LM_MAXM) return ETCH_EUNSUPPORTED;
    if (nj == 24 && nb_model == 10) return launch_lm_linearize<BodySMPL>(B, M, nb_active, consts, x, markers, valid, resid, jac, normal, (hipStream_t)stream);
    if (nj == 55 && nb_model == 20) return launch_lm_linearize<BodySMPLX>(B, M, nb_active, consts, x, markers, valid, resid, jac, normal, (hipStream_t)stream);
    return ETCH_EUNSUPPORTED;
}

int etch_rodrigues(int n, const float* theta, double* R, float* dR, void* stream) {
    if (n <= 0) return ETCH_OK;
    hipLaunchKernelGGL(rodrigues_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, theta, R, dR);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

// consts: 8 device pointers {v_template, shapedirs, posedirs, weights, J0, Jd, parents, extra_vids}
int etch_smpl_lbs(int B, int V, int nj, int nb, int n_extra, const void* const* consts, const float* x, float* verts, float* joints, void* stream) {
    if (B <= 0) return ETCH_OK;
    if (nj < 1 || nj > LBS_MAXJ || nb < 0 || nb > LBS_MAXB || 3 * nj + nb + 3 > 256) return ETCH_EUNSUPPORTED;
    LbsConsts C{(const float*)consts[0], (const float*)consts[1], (const float*)consts[2], (const float*)consts[3], (const float*)consts[4],
                (const float*)consts[5], (const int*)consts[6], (const int*)consts[7], n_extra};
    hipLaunchKernelGGL(smpl_lbs_kernel, dim3((V + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, C, nj, nb, V, x, verts, joints);
    ETCH_RETURN_IF_LAUNCH_FAILED();
    return ETCH_OK;
}

}  // extern "C"
