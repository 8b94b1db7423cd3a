// HIP kernels of the LM refinement engine (gfx950 / CDNA4, wave64).
//
// One LM round on one shard (see calib_lm.hip for the launch order):
//   view_setup   per view: Euler(deg) -> R, t, rotation-derivative axes      [M threads]
//   jacobian     per point: residual + 2xC Jacobian block -> HBM, sum r^2    [HBM-bound]
//   gram         per view: 16x16 J^T J and J^T r via v_mfma_f64_16x16x4_f64  [HBM-bound read of J]
//   schur        per view: 6x6 Cholesky elimination, block partial sums
//   reduce       fixed-order sum of block partials -> reduce buffer (all-reduced across shards)
//   update       accept/reject, lambda, L x L solve                          [1 thread]
//   backsub      per view: delta_i, writes the next candidate P
#pragma once
#include "point_model.hpp"
#include <stdint.h>

namespace calib {

constexpr int kTile = 256;        // points per jacobian workgroup
constexpr int kGramChunk = 512;   // points per gram work item (one wave)
constexpr int kGStride = 272;     // doubles per item: G 16x16 + g 16
constexpr int kMaxL = 10;
constexpr int kSchurThreads = 128;  // 8 views (16 lanes each) per workgroup
constexpr int kMaxSchurBlocks = 256;

typedef double d4 __attribute__((ext_vector_type(4)));

struct LMState {
    double lam, err_cur, last_err, lam_min, lam_max, err_min;
    int cur;        // index of the buffers holding the current P / view blocks
    int round;      // rounds completed (round 0 = bootstrap)
    int iters;      // LM iterations executed
    int max_iters;
    int done;
    int error;
    int accepted_last;
    int pad;
    double dc[kMaxL];
};

// reduce-buffer layout (doubles): [0] err_cand, then two variants of
//   Bsum[L*L] Ssub[L*L] gc[L] ssub[L] nfail
// variant A = candidate blocks with lambda_accept, B = current blocks with lambda_reject.
__host__ __device__ constexpr int variantSize(int L) { return 2 * L * L + 2 * L + 1; }
__host__ __device__ constexpr int reduceSize(int L) { return 1 + 2 * variantSize(L); }

__device__ __forceinline__ const double* selectP(const double* P0, const double* P1,
                                                 const LMState* st, int sel) {
    // sel 0: explicit P0; sel 1: candidate buffer of the LM state
    if (sel == 0) return P0;
    return (st->cur ^ 1) ? P1 : P0;
}

// ---------------------------------------------------------------- view_setup
// Euler angles in DEGREES -> R = Rz Ry Rx (src/mathutils.py:36-51). The numeric
// Rodrigues of the reference returns I when |theta| <= 1e-8 (np.isclose, :72-79).
template <typename T>
__global__ void view_setup_kernel(const double* __restrict__ P0, const double* __restrict__ P1,
                                  const LMState* __restrict__ st, int sel, int L,
                                  const int* __restrict__ view_ext, int nv, T* __restrict__ VC) {
    if (sel && st->done) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nv) return;
    const double* e = selectP(P0, P1, st, sel) + L + 6 * (int64_t)view_ext[j];
    const double deg = 0.017453292519943295;
    double s[3], c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double th = e[a] * deg;
        sincos(th, &s[a], &c[a]);
        if (fabs(th) <= 1e-8) { s[a] = 0.0; c[a] = 1.0; }
    }
    const double sx = s[0], cx = c[0], sy = s[1], cy = c[1], sz = s[2], cz = c[2];
    T* o = VC + (int64_t)j * kViewStride;
    o[0] = (T)(cz * cy);  o[1] = (T)(cz * sy * sx - sz * cx);  o[2] = (T)(cz * sy * cx + sz * sx);
    o[3] = (T)(sz * cy);  o[4] = (T)(sz * sy * sx + cz * cx);  o[5] = (T)(sz * sy * cx - cz * sx);
    o[6] = (T)(-sy);      o[7] = (T)(cy * sx);                 o[8] = (T)(cy * cx);
    o[9] = (T)e[3];  o[10] = (T)e[4];  o[11] = (T)e[5];
    o[12] = (T)(deg * cz * cy);  o[13] = (T)(deg * sz * cy);  o[14] = (T)(-deg * sy);
    o[15] = (T)(-deg * sz);      o[16] = (T)(deg * cz);       o[17] = (T)0;
}

// ---------------------------------------------------------------- jacobian
template <typename T>
struct JacArgs {
    using T2 = typename Pair<T>::type;
    const double* P0; const double* P1; const LMState* st; int sel;
    const T2* uv; const T2* XY; const T* Z; const int* pt_view; int64_t MN;
    const T* VC;
    T2* J;        // [MN][C] (du, dv)     may be null (projection / error only)
    T2* r;        // [MN]                 may be null
    T2* y;        // [MN] projection      may be null
    double* sse_part;   // [numTiles]
};

template <int MODEL, typename T>
__global__ __launch_bounds__(kTile) void jacobian_kernel(JacArgs<T> a) {
    using T2 = typename Pair<T>::type;
    constexpr int C = ModelTraits<MODEL>::C;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* swave = reinterpret_cast<double*>(smem);          // 4 wave sums (32 B)
    T* svc = reinterpret_cast<T*>(smem + 32);                 // staged view constants
    if (a.sel && a.st->done) return;
    const double* P = selectP(a.P0, a.P1, a.st, a.sel);
    const int tid = threadIdx.x;
    const int64_t tile0 = (int64_t)blockIdx.x * kTile;
    const int64_t p = tile0 + tid;
    const int64_t last = (tile0 + kTile < a.MN ? tile0 + kTile : a.MN) - 1;
    const int v0 = a.pt_view[tile0];
    const int nvt = a.pt_view[last] - v0 + 1;
    const T* src = a.VC + (int64_t)v0 * kViewStride;
    for (int i = tid; i < nvt * kViewStride; i += kTile) svc[i] = src[i];
    __syncthreads();

    double e = 0.0;
    if (p < a.MN) {
        Shared<MODEL, T> sp;
        sp.load(P);
        const T2 m = a.uv[p];
        const T2 xy = a.XY[p];
        const T z = a.Z[p];
        const T* vc = svc + (a.pt_view[p] - v0) * kViewStride;
        T u, v;
        if (a.J) {
            T Ju[C], Jv[C];
            jacobian_point<MODEL, T>(sp, vc, xy.x, xy.y, z, u, v, Ju, Jv);
            T2* dst = a.J + p * C;
#pragma unroll
            for (int c = 0; c < C; ++c) { T2 t; t.x = Ju[c]; t.y = Jv[c]; dst[c] = t; }
        } else {
            project_point<MODEL, T>(sp, vc, xy.x, xy.y, z, u, v);
        }
        const T ru = m.x - u, rv = m.y - v;
        if (a.r) { T2 t; t.x = ru; t.y = rv; a.r[p] = t; }
        if (a.y) { T2 t; t.x = u; t.y = v; a.y[p] = t; }
        e = (double)ru * (double)ru + (double)rv * (double)rv;
    }
    // fixed-order reduction: wave shuffle tree, then the 4 wave sums in order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) e += __shfl_down(e, off, 64);
    if ((tid & 63) == 0) swave[tid >> 6] = e;
    __syncthreads();
    if (tid == 0) a.sse_part[blockIdx.x] = (swave[0] + swave[1]) + (swave[2] + swave[3]);
}

// ---------------------------------------------------------------- gram (J^T J, J^T r)
// One wave per item (<= kGramChunk points of one view). Lane l = (k = l>>4, c = l&15)
// loads the 16-byte (du, dv) chunk of point 4g+k, column c: one wave-load is 1 KiB
// contiguous and is at once the A and the B operand of v_mfma_f64_16x16x4_f64
// (A[i][k] = J[row k][col i], B[k][j] = J[row k][col j]); u rows and v rows go through
// two MFMAs. J^T r rides on the VALU with a 2-step cross-lane sum at the end.
template <typename T, int C>
__global__ __launch_bounds__(256) void gram_kernel(const typename Pair<T>::type* __restrict__ J,
                                                   const typename Pair<T>::type* __restrict__ r,
                                                   const int64_t* __restrict__ item_pt0,
                                                   const int* __restrict__ item_n, int n_items,
                                                   const LMState* __restrict__ st, int sel,
                                                   double* __restrict__ G0, double* __restrict__ G1) {
    using T2 = typename Pair<T>::type;
    if (sel && st->done) return;
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int c = lane & 15, k = lane >> 4;
    const int64_t p0 = item_pt0[item];
    const int n = item_n[item];
    const bool cvalid = c < C;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    double gacc = 0.0;
    const T2* Jp = J + (p0 + k) * C + c;
    const T2* rp = r + p0 + k;
#pragma unroll 4
    for (int g = 0; g < n; g += 4) {
        const bool pv = (g + k) < n;
        double jx = 0.0, jy = 0.0, rx = 0.0, ry = 0.0;
        if (pv) { const T2 t = rp[g]; rx = (double)t.x; ry = (double)t.y; }
        if (pv && cvalid) { const T2 t = Jp[(int64_t)g * C]; jx = (double)t.x; jy = (double)t.y; }
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(jx, jx, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(jy, jy, acc, 0, 0, 0);
        gacc += jx * rx + jy * ry;
    }
    gacc += __shfl_xor(gacc, 16, 64);
    gacc += __shfl_xor(gacc, 32, 64);
    double* G = (sel ? ((st->cur ^ 1) ? G1 : G0) : G0) + (int64_t)item * kGStride;
    // f64 MFMA C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) G[(k + 4 * reg) * 16 + c] = acc[reg];
    if (k == 0) G[256 + c] = gacc;
}

// ---------------------------------------------------------------- per-view elimination
// 16 lanes per view; lane c owns column c of the view's symmetric 16x16 Gram.
__device__ __forceinline__ constexpr int tri(int m, int n) { return m * (m + 1) / 2 + n; }

template <int L>
struct Elim {
    double Lc[21];    // Cholesky factor of V + lam diag(V) (lower, off-diagonals)
    double invd[6];   // 1 / diagonal of the factor
    double z[6];      // Lc^-1 (this lane's column restricted to the view rows)
    double zg[6];     // Lc^-1 g_view
    bool fail;
};

__device__ __forceinline__ void load_view_col(const double* __restrict__ G, int item0, int nitems,
                                              int c, double (&col)[16], double& gc) {
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) col[rr] = 0.0;
    gc = 0.0;
    for (int it = 0; it < nitems; ++it) {
        const double* g = G + (int64_t)(item0 + it) * kGStride;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) col[rr] += g[rr * 16 + c];
        gc += g[256 + c];
    }
}

template <int L>
__device__ __forceinline__ void eliminate(const double (&col)[16], double gc, double lam,
                                          Elim<L>& e) {
    double V[21];
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int n = 0; n <= m; ++n) V[tri(m, n)] = __shfl(col[L + m], L + n, 16);
    double gv[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) gv[m] = __shfl(gc, L + m, 16);
    e.fail = false;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = V[tri(j, j)] + lam * V[tri(j, j)];     // JTJ + lam * diag(JTJ), src/calibrate.py:147,152
#pragma unroll
        for (int q = 0; q < j; ++q) d -= e.Lc[tri(j, q)] * e.Lc[tri(j, q)];
        if (!(d > 0.0)) e.fail = true;
        const double ljj = sqrt(d);
        const double inv = 1.0 / ljj;
        e.invd[j] = inv;
        e.Lc[tri(j, j)] = ljj;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double t = V[tri(i, j)];
#pragma unroll
            for (int q = 0; q < j; ++q) t -= e.Lc[tri(i, q)] * e.Lc[tri(j, q)];
            e.Lc[tri(i, j)] = t * inv;
        }
    }
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        double t = col[L + m], tg = gv[m];
#pragma unroll
        for (int n = 0; n < m; ++n) { t -= e.Lc[tri(m, n)] * e.z[n]; tg -= e.Lc[tri(m, n)] * e.zg[n]; }
        e.z[m] = t * e.invd[m];
        e.zg[m] = tg * e.invd[m];
    }
}

// ---------------------------------------------------------------- schur partials
// grid (nblocks, 2): y = 0 variant A (candidate blocks, lambda_accept),
//                    y = 1 variant B (current blocks, lambda_reject).
template <int L>
__global__ __launch_bounds__(kSchurThreads) void schur_kernel(const double* __restrict__ G0,
                                                              const double* __restrict__ G1,
                                                              const LMState* __restrict__ st,
                                                              const int* __restrict__ view_item0,
                                                              int nv, double* __restrict__ part) {
    constexpr int VA = variantSize(L);
    constexpr int NACC = 2 * L + 3;
    __shared__ double sacc[kSchurThreads / 16][16][NACC];
    if (st->done) return;
    const int variant = blockIdx.y;
    const int tid = threadIdx.x, c = tid & 15, grp = tid >> 4;
    double* out = part + ((int64_t)variant * gridDim.x + blockIdx.x) * VA;
    const bool boot = st->round == 0;
    if (variant == 1 && boot) {       // no "current" blocks yet
        for (int i = tid; i < VA; i += kSchurThreads) out[i] = 0.0;
        return;
    }
    const int cand = st->cur ^ 1;
    const int buf = variant == 0 ? cand : st->cur;
    const double* G = buf ? G1 : G0;
    const double lam = variant == 0 ? (boot ? st->lam : st->lam / 10) : st->lam * 10;

    double Bacc[L], Sacc[L], gacc = 0.0, sac = 0.0, nfail = 0.0;
#pragma unroll
    for (int i = 0; i < L; ++i) { Bacc[i] = 0.0; Sacc[i] = 0.0; }
    const int groupsPerBlock = kSchurThreads / 16;
    for (int v = blockIdx.x * groupsPerBlock + grp; v < nv; v += gridDim.x * groupsPerBlock) {
        double col[16], gc;
        const int i0 = view_item0[v];
        load_view_col(G, i0, view_item0[v + 1] - i0, c, col, gc);
        Elim<L> e;
        eliminate<L>(col, gc, lam, e);
        double dzg = 0.0;
#pragma unroll
        for (int m = 0; m < 6; ++m) dzg += e.z[m] * e.zg[m];
#pragma unroll
        for (int cc = 0; cc < L; ++cc) {
            double t = 0.0;
#pragma unroll
            for (int m = 0; m < 6; ++m) t += e.z[m] * __shfl(e.z[m], cc, 16);
            Sacc[cc] += t;
            Bacc[cc] += col[cc];
        }
        gacc += gc;
        sac += dzg;
        if (e.fail) nfail += 1.0;
    }
#pragma unroll
    for (int i = 0; i < L; ++i) { sacc[grp][c][i] = Bacc[i]; sacc[grp][c][L + i] = Sacc[i]; }
    sacc[grp][c][2 * L] = gacc;
    sacc[grp][c][2 * L + 1] = sac;
    sacc[grp][c][2 * L + 2] = nfail;
    __syncthreads();
    if (grp == 0 && c < L) {
        double t[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) t[i] = 0.0;
        for (int g = 0; g < groupsPerBlock; ++g)
#pragma unroll
            for (int i = 0; i < NACC; ++i) t[i] += sacc[g][c][i];
#pragma unroll
        for (int i = 0; i < L; ++i) { out[c * L + i] = t[i]; out[L * L + c * L + i] = t[L + i]; }
        out[2 * L * L + c] = t[2 * L];
        out[2 * L * L + L + c] = t[2 * L + 1];
        if (c == 0) out[2 * L * L + 2 * L] = t[2 * L + 2];
    }
}

// ---------------------------------------------------------------- reduce
__global__ __launch_bounds__(256) void reduce_kernel(const double* __restrict__ part, int nblocks, int VA,
                                                     const double* __restrict__ sse_part, int64_t n_sse,
                                                     const LMState* __restrict__ st,
                                                     double* __restrict__ red) {
    __shared__ double ssum[256];
    if (st->done) return;
    const int tid = threadIdx.x;
    double e = 0.0;
    for (int64_t i = tid; i < n_sse; i += 256) e += sse_part[i];
    ssum[tid] = e;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) ssum[tid] += ssum[tid + s];
        __syncthreads();
    }
    if (tid == 0) red[0] = ssum[0];
    for (int i = tid; i < 2 * VA; i += 256) {
        const int variant = i / VA, idx = i - variant * VA;
        double t = 0.0;
        for (int b = 0; b < nblocks; ++b) t += part[((int64_t)variant * nblocks + b) * VA + idx];
        red[1 + i] = t;
    }
}

// ---------------------------------------------------------------- update (1 thread)
// The control flow of src/calibrate.py:155-168 on the device.
template <int L>
__global__ void update_kernel(LMState* __restrict__ st, const double* __restrict__ red,
                              double* __restrict__ P0, double* __restrict__ P1,
                              double* __restrict__ trace) {
    constexpr int VA = variantSize(L);
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st->done) return;
    double* Pb[2] = {P0, P1};
    int cur = st->cur;
    const int cand = cur ^ 1;
    const double err_cand = red[0];
    double lam = st->lam;
    const double* sys;
    if (st->round == 0) {
        cur = cand;
        st->cur = cur;
        st->err_cur = err_cand;
        st->last_err = err_cand;
        sys = red + 1;
    } else {
        const int it = st->round - 1;
        const double err_cur = st->err_cur;
        const bool acc = err_cand < err_cur;      // strict; NaN rejects (src/calibrate.py:161)
        if (trace) {
            double* row = trace + (int64_t)it * (5 + L);
            row[0] = it; row[1] = err_cur; row[2] = err_cand; row[3] = lam; row[4] = acc ? 1.0 : 0.0;
            for (int i = 0; i < L; ++i) row[5 + i] = Pb[cur][i];
        }
        st->last_err = err_cur;                   // the reference returns the pre-update error (:155,171)
        if (acc) {
            cur = cand; st->cur = cur; st->err_cur = err_cand;
            lam = lam / 10; sys = red + 1;
        } else {
            lam = lam * 10; sys = red + 1 + VA;
        }
        st->lam = lam;
        st->iters = it + 1;
        st->accepted_last = acc ? 1 : 0;
        if (!(st->lam_min < lam && lam < st->lam_max) || err_cur < st->err_min ||
            it + 1 >= st->max_iters) {
            st->done = 1;
            st->round += 1;
            return;
        }
    }
    st->round += 1;
    if (sys[2 * L * L + 2 * L] > 0.0) { st->error = -3; st->done = 1; return; }
    // S = B + lam diag(B) - sum E Vh^-1 E^T ; s = g_c - sum E Vh^-1 g_v
    double S[L][L + 1];
    for (int i = 0; i < L; ++i) {
        for (int j = 0; j < L; ++j) S[i][j] = sys[i * L + j] - sys[L * L + i * L + j];
        S[i][i] += lam * sys[i * L + i];
        S[i][L] = sys[2 * L * L + i] - sys[2 * L * L + L + i];
    }
    // Gaussian elimination with partial pivoting
    for (int col = 0; col < L; ++col) {
        int piv = col;
        double best = fabs(S[col][col]);
        for (int i = col + 1; i < L; ++i) if (fabs(S[i][col]) > best) { best = fabs(S[i][col]); piv = i; }
        if (!(best > 0.0)) { st->error = -3; st->done = 1; return; }
        if (piv != col) for (int j = col; j <= L; ++j) { const double t = S[col][j]; S[col][j] = S[piv][j]; S[piv][j] = t; }
        const double inv = 1.0 / S[col][col];
        for (int i = col + 1; i < L; ++i) {
            const double f = S[i][col] * inv;
            for (int j = col; j <= L; ++j) S[i][j] -= f * S[col][j];
        }
    }
    double dc[L];
    for (int i = L - 1; i >= 0; --i) {
        double t = S[i][L];
        for (int j = i + 1; j < L; ++j) t -= S[i][j] * dc[j];
        dc[i] = t / S[i][i];
    }
    for (int i = 0; i < L; ++i) {
        st->dc[i] = dc[i];
        Pb[cur ^ 1][i] = Pb[cur][i] + dc[i];
    }
}

// ---------------------------------------------------------------- back-substitution
// delta_i = Vh^-1 (g_i - E_i^T dc); next candidate P[cur^1] = P[cur] + delta.
template <int L>
__global__ __launch_bounds__(kSchurThreads) void backsub_kernel(const double* __restrict__ G0,
                                                                const double* __restrict__ G1,
                                                                const LMState* __restrict__ st,
                                                                const int* __restrict__ view_item0,
                                                                const int* __restrict__ view_ext, int nv,
                                                                double* __restrict__ P0,
                                                                double* __restrict__ P1) {
    if (st->done) return;
    const int tid = threadIdx.x, c = tid & 15;
    const int v = blockIdx.x * (kSchurThreads / 16) + (tid >> 4);
    if (v >= nv) return;       // whole 16-lane group leaves together
    const int cur = st->cur;
    const double* G = cur ? G1 : G0;
    const double* Pc = cur ? P1 : P0;
    double* Pn = cur ? P0 : P1;
    double col[16], gc;
    const int i0 = view_item0[v];
    load_view_col(G, i0, view_item0[v + 1] - i0, c, col, gc);
    Elim<L> e;
    eliminate<L>(col, gc, st->lam, e);
    const double dcc = c < L ? st->dc[c] : 0.0;
    double w[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        double t = dcc * e.z[m];
        t += __shfl_xor(t, 1, 16);
        t += __shfl_xor(t, 2, 16);
        t += __shfl_xor(t, 4, 16);
        t += __shfl_xor(t, 8, 16);
        w[m] = e.zg[m] - t;
    }
    // Lc^T d = w
    double d[6];
#pragma unroll
    for (int m = 5; m >= 0; --m) {
        double t = w[m];
#pragma unroll
        for (int n = m + 1; n < 6; ++n) t -= e.Lc[tri(n, m)] * d[n];
        d[m] = t * e.invd[m];
    }
    if (c < 6) {
        const int64_t o = L + 6 * (int64_t)view_ext[v] + c;
        double dv = d[0];
#pragma unroll
        for (int m = 1; m < 6; ++m) dv = (c == m) ? d[m] : dv;
        Pn[o] = Pc[o] + dv;
    }
}

// ---------------------------------------------------------------- small forward-model kernels
template <int MODEL>
__global__ void distort_points_kernel(const double* __restrict__ xn, const double* __restrict__ k,
                                      int64_t n, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int NK = ModelTraits<MODEL>::NK;
    double kk[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) kk[j] = k[j];
    double xd, yd, a, b, c, dkx[NK], dky[NK];
    distort<MODEL, double>(kk, xn[2 * i], xn[2 * i + 1], xd, yd, a, b, c, dkx, dky);
    out[2 * i] = xd;
    out[2 * i + 1] = yd;
}

template <int MODEL>
__global__ void project_cam_kernel(const double* __restrict__ A, const double* __restrict__ cam,
                                   const double* __restrict__ k, int64_t n,
                                   double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int NK = ModelTraits<MODEL>::NK;
    Shared<MODEL, double> sp;
    sp.al = A[0]; sp.ga = A[1]; sp.uc = A[2]; sp.be = A[4]; sp.vc = A[5];
#pragma unroll
    for (int j = 0; j < NK; ++j) sp.k[j] = k[j];
    const double vc[kViewStride] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    double u, v;
    project_point<MODEL, double>(sp, vc, cam[3 * i], cam[3 * i + 1], cam[3 * i + 2], u, v);
    out[2 * i] = u;
    out[2 * i + 1] = v;
}

}  // namespace calib
