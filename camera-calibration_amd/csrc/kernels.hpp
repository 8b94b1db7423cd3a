// HIP kernels of the LM refinement engine (gfx950 / CDNA4, wave64).
//
// One LM round on one shard (launch order in calib_lm.hip: calib_lm_local / calib_lm_update):
//   fused            per view item: residuals + 2xC Jacobian blocks in registers -> LDS transpose ->
//                    16x16 J^T J, J^T r, sum r^2 via v_mfma_f64_16x16x4_f64 / 4x4x4 blocks   [fp64-ALU-bound]
//     (fused_stream  the same for large uniform fp64 shards: equal shares of 4-point groups per wave, batches across
//                    view boundaries, records straight from the accumulators)
//     (two-kernel mode instead: jacobian -> compact J in HBM -> gram               [HBM-bound])
//   schur            per view: 6x6 Cholesky elimination; sum E Vh^-1 [E^T|g] = W^T W via MFMA
//   reduce           fixed-order sum of block partials -> reduce buffer (all-reduced across shards)
//   update_backsub   every 16-lane group: accept/reject + L x L solve (redundantly, double-buffered
//                    state), its view's delta_i, the next candidate P and its view constants
// view_setup (Euler(deg) -> R, t, derivative axes) runs stand-alone only for round 0 and calib_eval.
#pragma once
#include "point_model.hpp"
#include <stdint.h>
#include <type_traits>

namespace calib {

constexpr int kTile = 256;        // points per jacobian workgroup
constexpr int kGramChunk = 512;   // points per gram work item (one wave)
constexpr int kGramUnroll = 4;    // 4-point groups (1 KiB wave-loads of J) in flight per trip
// Normal equations of one pass over the points. With the parameter order (L shared, 6 view) the
// 16x16 tile G = J^T J of a view item splits into the shared block B = G[0:L, 0:L] and the six view
// rows R = G[L:L+6, 0:16] = [E^T | V | 0] (the E block above the diagonal is R's transpose).
// * Per item, a 768-byte record = the six rows of R (six 128-byte lines) and nothing else: all the per-view
//   elimination reads. V is symmetric and its readers take the lower triangle (n <= m), so the view's six entries
//   of J^T r, g_v, ride in six slots of V's strictly upper triangle (gvSlot). Rounds 1-3 kept a 1 KiB record
//   (R, the whole of g, sum r^2: 113 of 128 doubles, 87 of them ever read); the per-view kernels stream the
//   records of both buffers every LM round, and on large shards that traffic is what they are bound by.
// * Per workgroup of the kernel that forms them (4 waves = up to 4 items), one partial of what is
//   only ever needed summed over all views: B, g_c = g[0:L], sum r^2.
constexpr int kGRows = 0;         // R[m][c] at m * 16 + c
constexpr int kGStride = 96;      // doubles per item record (768 B)
// record slot of g_v[j] = (J^T r)[L + j]: V(0, 1..5) for j = 0..4, V(1, 2) for j = 5
__host__ __device__ __forceinline__ constexpr int gvSlot(int L, int j) { return j < 5 ? L + 1 + j : 16 + L + 2; }
// the j a record slot carries g_v[j] for, or -1
__host__ __device__ __forceinline__ constexpr int gvOfSlot(int L, int slot) {
    const int m = slot >> 4, n = (slot & 15) - L;
    return (m == 0 && n >= 1 && n <= 5) ? n - 1 : ((m == 1 && n == 2) ? 5 : -1);
}
constexpr int kPartStride = 112;  // doubles per workgroup partial: B (L*L), g_c (L), sum r^2
constexpr int kMaxL = 10;

// Fused kernel: how an output slot is assembled from the two accumulator tiles of a wave (TU: the u
// rows of J, TV: the v rows): slot i = TU[tab[i] & 0xffff] + TV[tab[i] >> 16], tile index row * 16 +
// col, kEmitZero = "nothing". Fisheye (C = 15): the residual rides in MFMA column 15, so both tiles
// are plain J^T J with J^T r in row 15 and sum r^2 in the corner. Radial-tangential (C = 16):
// J's constant columns 3 and 4, (1,0) and (0,1), are replaced by one column of (1,1) -- TU[.][3] =
// sum Ju, TV[.][3] = sum Jv -- and column 4 carries the residual (see fused_kernel).
// tab[0 .. kGStride) describes the item record, tab[kGStride .. kGStride + kPartStride) the partial.
constexpr int kEmitZero = 256;
constexpr int kEmitTile = 264;        // doubles per tile in LDS: 256 + the zero slot, padded
constexpr int kEmitTabSize = kGStride + kPartStride;
inline void buildEmitTable(int C, uint32_t* tab /* kEmitTabSize */) {
    const int L = C - 6;
    const auto both = [](int idx) { return (uint32_t)idx | ((uint32_t)idx << 16); };
    const auto pick = [](int iu, int iv) { return (uint32_t)iu | ((uint32_t)iv << 16); };
    const int Z = kEmitZero;
    // entry (row, col) of the true J^T J
    const auto entry = [&](int row, int col) -> uint32_t {
        if (C == 15) return col == 15 ? both(Z) : both(row * 16 + col);
        const bool rs = row == 3 || row == 4, cs = col == 3 || col == 4;
        if (!rs && !cs) return both(row * 16 + col);
        if (!rs) return col == 3 ? pick(row * 16 + 3, Z) : pick(Z, row * 16 + 3);      // sum Ju[row] / sum Jv[row]
        if (!cs) return row == 3 ? pick(3 * 16 + col, Z) : pick(Z, 3 * 16 + col);
        if (row != col) return both(Z);                                                  // (1,0).(0,1) = 0
        return row == 3 ? pick(3 * 16 + 3, Z) : pick(Z, 3 * 16 + 3);                     // point count
    };
    // entry c of J^T r, and sum r^2
    const auto gentry = [&](int c) -> uint32_t {
        if (C == 15) return both(15 * 16 + c);
        return c == 3 ? pick(3 * 16 + 4, Z) : (c == 4 ? pick(Z, 3 * 16 + 4) : both(c * 16 + 4));
    };
    const uint32_t sse = C == 15 ? both(15 * 16 + 15) : both(4 * 16 + 4);
    for (int i = 0; i < kEmitTabSize; ++i) tab[i] = both(Z);
    for (int m = 0; m < 6; ++m)
        for (int c = 0; c < 16; ++c) tab[kGRows + m * 16 + c] = entry(L + m, c);
    for (int j = 0; j < 6; ++j) tab[gvSlot(L, j)] = gentry(L + j);
    uint32_t* part = tab + kGStride;
    for (int r = 0; r < L; ++r)
        for (int c = 0; c < L; ++c) part[r * L + c] = entry(r, c);
    for (int c = 0; c < L; ++c) part[L * L + c] = gentry(c);
    part[L * L + L] = sse;
}

// Compact Jacobian in HBM: groups of 4 points, [group][column][point-in-group] of (du, dv) pairs.
// The jacobian kernel's store of one column then writes 64 B per 4 lanes (instead of 16 B per
// lane at a C*16 B stride), and the 4 points x 16 columns a gram wave-load needs are still one
// contiguous 1 KiB (any lane permutation inside it coalesces the same).
__host__ __device__ __forceinline__ constexpr int64_t jIndex(int64_t p, int c, int C) {
    return ((p >> 2) * C + c) * 4 + (p & 3);
}
constexpr int kSchurThreads = 256;  // 16 views (16 lanes each) per workgroup of the update kernel (128: c4 +1 %)
constexpr int kMaxSchurBlocks = 1024;

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// every lane fetches `v` of lane `srcLane` (ds_bpermute: LDS crossbar, no LDS memory)
__device__ __forceinline__ double2 lane_gather(double2 v, int srcLane) {
    const int a = srcLane << 2;
    int2 lo = __builtin_bit_cast(int2, v.x), hi = __builtin_bit_cast(int2, v.y);
    lo.x = __builtin_amdgcn_ds_bpermute(a, lo.x);  lo.y = __builtin_amdgcn_ds_bpermute(a, lo.y);
    hi.x = __builtin_amdgcn_ds_bpermute(a, hi.x);  hi.y = __builtin_amdgcn_ds_bpermute(a, hi.y);
    double2 o;
    o.x = __builtin_bit_cast(double, lo);  o.y = __builtin_bit_cast(double, hi);
    return o;
}
__device__ __forceinline__ float2 lane_gather(float2 v, int srcLane) {
    const int a = srcLane << 2;
    float2 o;
    o.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a, __builtin_bit_cast(int, v.x)));
    o.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a, __builtin_bit_cast(int, v.y)));
    return o;
}

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
    int* notify;    // host-visible word the writer sets when the loop is over (null: none), so that a host that runs
                    // ahead of the device can stop enqueueing rounds without synchronising
    double dc[kMaxL];
    double B[kMaxL * kMaxL];   // sum_views J_s^T J_s and
    double gc[kMaxL];          // sum_views J_s^T r at the CURRENT parameters (variant B re-uses them)
};

// reduce-buffer layout (doubles): two variants of
//   Bsum[L*L] Ssub[L*L] gc[L] ssub[L] nfail sse
// variant A = candidate blocks with lambda_accept (its sse is err(candidate)),
// variant B = current blocks with lambda_reject.
__host__ __device__ constexpr int variantSize(int L) { return 2 * L * L + 2 * L + 2; }
__host__ __device__ constexpr int reduceSize(int L) { return 2 * variantSize(L); }

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
    const T2* uv; const T2* XY; const T* Z; const int* pt_view;
    int64_t p_begin, p_end;   // this launch covers points [p_begin, p_end) (a chunk of whole views)
    const T* VC;
    T2* J;        // jIndex(p - p_begin, c, C) (du, dv)   may be null (projection / error only)
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
    const int64_t tile0 = a.p_begin + (int64_t)blockIdx.x * kTile;
    const int64_t p = tile0 + tid;
    const int64_t last = (tile0 + kTile < a.p_end ? tile0 + kTile : a.p_end) - 1;
    const int v0 = a.pt_view[tile0];
    const int nvt = a.pt_view[last] - v0 + 1;
    const T* src = a.VC + (int64_t)v0 * kViewStride;
    for (int i = tid; i < nvt * kViewStride; i += kTile) svc[i] = src[i];
    __syncthreads();

    double e = 0.0;
    if (p < a.p_end) {
        Shared<MODEL, T> sp;
        sp.load(P);
        const T2 m = a.uv[p];
        const T2 xy = a.XY[p];
        const T z = a.Z[p];
        const T* vc = svc + (a.pt_view[p] - v0) * kViewStride;
        T u, v;
        if (a.J) {
            T2 Jc[C];
            jacobian_point<MODEL, T>(sp, vc, xy.x, xy.y, z, u, v, Jc);
            T2* dst = a.J + jIndex(p - a.p_begin, 0, C);
#pragma unroll
            for (int c = 0; c < C; ++c) dst[4 * c] = Jc[c];
        } else {
            project_point<MODEL, T>(sp, vc, xy.x, xy.y, z, u, v);
        }
        const T ru = m.x - u, rv = m.y - v;
        if (a.r) { T2 t; t.x = ru; t.y = rv; a.r[p] = t; }
        if (a.y) { T2 t; t.x = u; t.y = v; a.y[p] = t; }
        e = (double)ru * (double)ru + (double)rv * (double)rv;
    }
    if (!a.sse_part) return;      // LM rounds take sum r^2 from the gram kernel
    // fixed-order reduction: wave shuffle tree, then the 4 wave sums in order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) e += __shfl_down(e, off, 64);
    if ((tid & 63) == 0) swave[tid >> 6] = e;
    __syncthreads();
    if (tid == 0) a.sse_part[blockIdx.x] = (swave[0] + swave[1]) + (swave[2] + swave[3]);
}

// ---------------------------------------------------------------- gram (J^T J, J^T r)
// One item = <= kGramChunk points of one view, worked on by WPI waves of a workgroup (1, 2 or 4,
// chosen per problem from the points per item; the item's 4-point groups are split between them
// and their partial tiles summed through LDS in wave order). Lane l = (k = l>>4, c = l&15)
// loads the 16-byte (du, dv) chunk of point 4g+k, column c: one wave-load is 1 KiB contiguous
// and is at once the A and the B operand of v_mfma_f64_16x16x4_f64 (A[i][k] = J[row k][col i],
// B[k][j] = J[row k][col j]); u rows and v rows go through two MFMAs. J^T r and sum r^2 ride on
// the VALU with a 2-step cross-lane sum at the end.
template <typename T, int C>
__global__ __launch_bounds__(256) void gram_kernel(const typename Pair<T>::type* __restrict__ J,
                                                   const typename Pair<T>::type* __restrict__ r,
                                                   const int64_t* __restrict__ item_pt0,
                                                   const int* __restrict__ item_n, int item_begin,
                                                   int item_end, int64_t j_origin, int wpi,
                                                   const LMState* __restrict__ st, int sel,
                                                   double* __restrict__ G0, double* __restrict__ G1,
                                                   double* __restrict__ part, int part_base) {
    using T2 = typename Pair<T>::type;
    constexpr int L = C - 6;
    __shared__ double stile[4][256 + 16 + 8];   // per wave: tile, J^T r, sum r^2
    if (sel && st->done) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lw = wpi == 4 ? 2 : (wpi == 2 ? 1 : 0);           // wpi is 1, 2 or 4
    const int sub = wave & (wpi - 1);
    const int item = item_begin + blockIdx.x * (4 >> lw) + (wave >> lw);
    const bool valid = item < item_end;
    const int c = lane & 15, k = lane >> 4;
    const bool cvalid = c < C;
    const int cc = cvalid ? c : C - 1;          // lane 15 of a 15-column model re-reads column 14, zeroed below
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    double gacc = 0.0, eacc = 0.0;
    if (valid) {
        // The item's points [pbeg, pend) are walked in the storage groups of jIndex (4 points,
        // 4*C chunks of 16 B): lane l loads chunk l of the group -- adjacent lanes, adjacent
        // addresses -- and one ds_bpermute per dword hands lane (k, c) its operand, chunk 4c+k.
        // Points of a boundary group that belong to a neighbouring view are zeroed.
        // q = p - j_origin indexes the chunk's J buffer; r is indexed by the absolute p
        const int64_t pbeg = item_pt0[item] - j_origin, pend = pbeg + item_n[item];
        const T2* rq = r + j_origin;
        const int64_t grp0 = pbeg >> 2, grp1 = (pend + 3) >> 2;
        const int per = (int)((grp1 - grp0 + wpi - 1) >> lw);
        const int64_t gb = grp0 + (int64_t)sub * per;
        const int64_t ge = gb + per < grp1 ? gb + per : grp1;
        const int chunk = lane < 4 * C ? lane : 4 * C - 1;
        const int srcLane = 4 * cc + k;
        d4 accv = {0.0, 0.0, 0.0, 0.0};                      // v rows: a second, independent MFMA chain
        for (int64_t g = gb; g < ge; g += kGramUnroll) {
            T2 raw[kGramUnroll], rv[kGramUnroll];
            // wave-uniform: every point of these groups belongs to the item -- no masks, no 64-bit index selects
            // (the masked form below spends ~50 vector instructions per group, this one ~12)
            if (4 * g >= pbeg && 4 * (g + kGramUnroll) <= pend && g + kGramUnroll <= ge) {
                const T2* Jg = J + g * (4 * C) + chunk;
                const T2* rg = rq + 4 * g + k;
#pragma unroll
                for (int u = 0; u < kGramUnroll; ++u) { raw[u] = Jg[u * 4 * C]; rv[u] = rg[4 * u]; }
#pragma unroll
                for (int u = 0; u < kGramUnroll; ++u) {
                    const T2 jv = lane_gather(raw[u], srcLane);
                    const double rx = (double)rv[u].x, ry = (double)rv[u].y;
                    const double jx = (C == 16 || cvalid) ? (double)jv.x : 0.0;
                    const double jy = (C == 16 || cvalid) ? (double)jv.y : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(jx, jx, acc, 0, 0, 0);
                    accv = __builtin_amdgcn_mfma_f64_16x16x4f64(jy, jy, accv, 0, 0, 0);
                    gacc += jx * rx + jy * ry;
                    eacc += rx * rx + ry * ry;
                }
                continue;
            }
            bool pv[kGramUnroll];
#pragma unroll
            for (int u = 0; u < kGramUnroll; ++u) {
                const int64_t gu = g + u < ge ? g + u : ge - 1;
                const int64_t pt = 4 * (g + u) + k;
                pv[u] = (g + u < ge) && pt >= pbeg && pt < pend;
                raw[u] = J[gu * (4 * C) + chunk];
                rv[u] = rq[pv[u] ? pt : pbeg];
            }
#pragma unroll
            for (int u = 0; u < kGramUnroll; ++u) {
                const T2 jv = lane_gather(raw[u], srcLane);
                const bool ok = pv[u];
                const double rx = ok ? (double)rv[u].x : 0.0, ry = ok ? (double)rv[u].y : 0.0;
                const double jx = (ok && cvalid) ? (double)jv.x : 0.0;
                const double jy = (ok && cvalid) ? (double)jv.y : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(jx, jx, acc, 0, 0, 0);
                accv = __builtin_amdgcn_mfma_f64_16x16x4f64(jy, jy, accv, 0, 0, 0);
                gacc += jx * rx + jy * ry;
                eacc += rx * rx + ry * ry;
            }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) acc[reg] += accv[reg];
        gacc += __shfl_xor(gacc, 16, 64);
        gacc += __shfl_xor(gacc, 32, 64);
        eacc += __shfl_xor(eacc, 16, 64);
        eacc += __shfl_xor(eacc, 32, 64);
    }
    double* Gbase = sel ? ((st->cur ^ 1) ? G1 : G0) : G0;
    // every wave parks its tile (f64 MFMA C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg;
    // zeros for a wave without an item), J^T r and sum r^2 in LDS
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) stile[wave][(k + 4 * reg) * 16 + c] = acc[reg];
    if (k == 0) stile[wave][256 + c] = gacc;
    if (lane == 0) stile[wave][272] = eacc;
    __syncthreads();
    // the workgroup's partial of the shared block (fixed wave order)
    if (threadIdx.x < L * L + L + 1) {
        const int t = threadIdx.x;
        const int idx = t < L * L ? (t / L) * 16 + t % L : (t < L * L + L ? 256 + (t - L * L) : 272);
        part[(int64_t)(part_base + blockIdx.x) * kPartStride + t] =
            (stile[0][idx] + stile[1][idx]) + (stile[2][idx] + stile[3][idx]);
    }
    // the item's record: its first wave sums the item's partial tiles in wave order
    if (sub == 0 && valid) {
        double* G = Gbase + (int64_t)item * kGStride;
        for (int i = lane; i < kGStride; i += 64) {
            const int j = gvOfSlot(L, i);
            const int idx = j >= 0 ? 256 + L + j : (L + i / 16) * 16 + i % 16;
            double t = stile[wave][idx];
            for (int w = 1; w < wpi; ++w) t += stile[wave + w][idx];
            G[i] = t;
        }
    }
}

// ---------------------------------------------------------------- fused jacobian + gram
// The same per-item J^T J / J^T r / sum r^2 as jacobian_kernel + gram_kernel, without the compact
// Jacobian ever leaving the CU: a wave evaluates 64 points (one per lane, closed-form 2 x C block
// in registers), transposes 32 points at a time through its private LDS slab (rows padded to an
// odd number of 16-B chunks: conflict-free ds_write_b128 / ds_read_b128) into the MFMA operand
// map (lane (k, c) <- point 4s+k, column c) and feeds v_mfma_f64_16x16x4_f64. The view constants
// are wave-uniform (scalar loads). HBM traffic per point: the 40 B of inputs (5 values of width w).
// On a gfx950 SIMD, v_mfma_f64 and VALU instructions of DIFFERENT waves do not overlap either (tools/ubench:
// FMA-only waves make no progress while a SIMD mate streams fp64 MFMAs), so the kernel's floor is the sum of
// its matrix and vector issue time; a producer / consumer split of the waves and a persistent-wave form were
// both built and measured slower.
// (second __launch_bounds__ argument = waves per SIMD the register allocator must leave room for. fp32 storage: the
// kernel fits 80 VGPRs without a spill -- 6 waves per SIMD instead of 4 at 115 VGPRs: c4 shard 20.4 -> 18.9 us; 8 waves
// spill and cost 30 %; the several-views-per-wave radtan form needs 94: 5 waves)
constexpr int kFusedRowChunks = 17;                       // 16 columns + 1 pad chunk (odd => conflict-free)

// ROWS = points transposed per LDS pass (32: two passes per 64-point batch, half the lanes
// writing each time; 64: one pass, twice the LDS). WAVES = waves per workgroup.
template <int MODEL, typename T, int ROWS, int WAVES, bool G44, bool MULTI>
__global__ __launch_bounds__(64 * WAVES, (sizeof(T) == 4 ? (MULTI ? 5 : 6)
                                                             : (ModelTraits<MODEL>::C == 16 ? 4 : 3))) void fused_kernel(const double* __restrict__ P0, const double* __restrict__ P1,
                                                    const typename Pair<T>::type* __restrict__ uv,
                                                    const typename Pair<T>::type* __restrict__ XY,
                                                    const T* __restrict__ Z, const T* __restrict__ VC,
                                                    const int64_t* __restrict__ item_pt0,
                                                    const int* __restrict__ item_n,
                                                    const int* __restrict__ item_view, int n_items,
                                                    int uniform_n, int ipw, int wpi,
                                                    const uint32_t* __restrict__ emit_tab,
                                                    const LMState* __restrict__ st, int sel,
                                                    double* __restrict__ G0, double* __restrict__ G1,
                                                    double* __restrict__ part) {
    using T2 = typename Pair<T>::type;
    constexpr int C = ModelTraits<MODEL>::C;
    constexpr int RS = kFusedRowChunks;
    constexpr bool MF32 = sizeof(T) == 4;                  // fp32 storage: fp32 MFMA per pass, fp64 across passes
    // Row r of a wave's slab starts at chunk rowOff(r). fp32 storage: 16 chunks + 1 pad chunk per row. fp64: 16
    // chunks per row and one pad chunk per PAIR of rows, so that rows 2 m and 2 m + 1 share their bank phase: the
    // four 16-lane groups a ds_read_b128 is served in each hold the column sets {0-3, 12-15} of an even and {4-11}
    // of the odd row of a pair (or the other way round) -- 16 distinct chunk phases exactly when the two rows are
    // in phase. (With a pad chunk per row every group had one 2-way conflict: 8 LDS cycles per read instead of 4.)
    // Stores stay conflict-free because a point is not processed by lane = row: see sl below.
    constexpr int SLAB = MF32 ? ROWS * RS : ROWS * 16 + ROWS / 2;
    auto rowOff = [](int r) { return MF32 ? r * RS : r * 16 + (r >> 1); };
    constexpr int HALVES = 64 / ROWS;
    constexpr bool RCOL = C < 16;                          // a free 16th MFMA column: J^T r and sum r^2 for free
    // C == 16 (radial-tangential): columns 3 and 4 of J are the constants (1,0) and (0,1). With the u rows
    // and the v rows accumulated in separate tiles, column 3 can carry (1,1) -- tile_u[.][3] = sum Ju,
    // tile_v[.][3] = sum Jv, i.e. what columns 3 and 4 used to give -- and column 4 the residual
    // (ru, rv): (tile_u + tile_v)[.][4] = J^T r, [4][4] = sum r^2, tile_u[3][4] = sum ru, tile_v[3][4] = sum rv.
    constexpr bool ONES = !RCOL;
    // one slab per wave; after the main loop the same memory holds the wave's two accumulator tiles
    __shared__ __attribute__((aligned(16))) unsigned char smem[WAVES * SLAB * sizeof(T2)];
    if (sel && st->done) return;
    const double* P = selectP(P0, P1, st, sel);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lw = wpi == 4 ? 2 : (wpi == 2 ? 1 : 0);           // wpi is 1, 2 or 4: shifts, not divisions
    const int sub = wave & (wpi - 1);
    // ipw > 1 (host: uniform single-wave items of the tile forms only): the wave works through ipw consecutive
    // items -- one prologue, one workgroup partial and one barrier for all of them, the next item's first points
    // requested during this item's last batch; the shared part of the tiles keeps accumulating across the items
    static_assert(!(G44 && MULTI), "multi-item waves exist for the tile forms only");
    const int item_first = MULTI ? (blockIdx.x * WAVES + wave) * ipw : blockIdx.x * (WAVES >> lw) + (wave >> lw);
    const bool valid = item_first < n_items;
    int item = item_first;
    const int c = lane & 15, k = lane >> 4;
    // Lane l evaluates point sl(l) of its batch and stores row sl(l): inside each 16-lane block lanes 0-7 take the
    // even rows, lanes 8-15 the odd ones. A ds_write_b128 is served 8 consecutive lanes at a time; their rows
    // 0, 2, .. 14 (or 1, 3, .. 15) have 8 different bank phases in the fp64 layout above. Rows stay in point order,
    // so the contraction (which groups rows 4 s .. 4 s + 3) and the handling of the last, partial batch do not care.
    const int sl = MF32 ? lane : ((lane & 48) | ((lane & 7) << 1) | ((lane >> 3) & 1));
    T2* slab = reinterpret_cast<T2*>(smem) + wave * SLAB;
    d4 acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
    // fp64: the Gram is built from 4x4 blocks by v_mfma_f64_4x4x4_4b (see the contraction below): per row kind
    // (u, v) the diagonal blocks (b, b), the blocks (b, b+1) and the blocks (b, b+2) of the 4 x 4 block grid
    double d0u = 0.0, d0v = 0.0, d1u = 0.0, d1v = 0.0, d2u = 0.0;
    double* Gbase = sel ? ((st->cur ^ 1) ? G1 : G0) : G0;
    // One wave per item (tile forms): the item's record goes to HBM straight from the accumulators (lane (k, c) holds
    // rows k + 4 reg -- 4 k + reg behind the fp32 MFMA -- of column c; 16 lanes store 128 contiguous bytes): rows
    // L..L+5 of J^T J but for the six slots that carry g_v, which the residual's row fills. No table, no LDS,
    // nothing to wait for.
    auto emitDirect = [&](int it) {
        constexpr int L = C - 6, RESROW = RCOL ? 15 : 4;
        double* G = Gbase + (int64_t)it * kGStride;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = MF32 ? 4 * k + reg : k + 4 * reg;
            double val = acc[reg] + acc2[reg];
            if (ONES) {
                // radtan: the u-tile's / v-tile's column 3 is the true column 3 / 4 (sum Ju / sum Jv, see above)
                const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(acc2[reg]), 0x111, 0xf, 0xf, false);
                const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(acc2[reg]), 0x111, 0xf, 0xf, false);
                const double fromLeft = __hiloint2double(hi, lo);       // row_shr:1: lane c <- lane c - 1
                val = c == 3 ? acc[reg] : (c == 4 ? fromLeft : val);
            }
            if (row >= L && row < L + 6) {
                const int slot = kGRows + (row - L) * 16 + c;
                if (gvOfSlot(L, slot) < 0) G[slot] = (RCOL && c == 15) ? 0.0 : val;
            } else if (row == RESROW) {
                if (c >= L && c < L + 6) G[gvSlot(L, c - L)] = val;
            }
        }
    };
    if (valid) {
        // uniform_n > 0 (item i = view i = points [i n, (i + 1) n)): the extent comes from the launch arguments and
        // the first points are requested one memory latency earlier than through the item tables
        int64_t pbeg = uniform_n ? (int64_t)item * uniform_n : item_pt0[item];
        const int n = uniform_n ? uniform_n : item_n[item];
        const int item_last = MULTI ? (item_first + ipw < n_items ? item_first + ipw : n_items) - 1 : item_first;
        const int per = ((((n + 3) >> 2) + wpi - 1) >> lw) << 2;    // points per wave, multiple of 4
        const int qbeg = sub * per;
        const int qend = qbeg + per < n ? qbeg + per : n;
        Shared<MODEL, T> sp;
        sp.load(P);
        // the constant columns of the slab rows: d(u,v)/duc = (1,0), d(u,v)/dvc = (0,1); for C == 16
        // column 3 holds (1,1) and column 4 is rewritten with the residual per batch (ONES, above)
        if (lane < ROWS) {
            T2 c3, c4;
            c3.x = T(1); c3.y = ONES ? T(1) : T(0);
            c4.x = T(0); c4.y = T(1);
            slab[rowOff(lane) + 3] = c3;
            if (!ONES) slab[rowOff(lane) + 4] = c4;
            // columns 0, 1, 2 are (xd, 0), (0, yd), (yd, 0): their zero halves are set once, the batches
            // store only the other 8 bytes
            T* zr = reinterpret_cast<T*>(slab + rowOff(lane));
            zr[1] = T(0); zr[2] = T(0); zr[5] = T(0);
        }
        // inputs of the next batch are requested before the current batch is evaluated
        int64_t pn = pbeg + (qbeg + sl < qend ? qbeg + sl : qend - 1);
        T2 m_n = uv[pn], xy_n = XY[pn];
        T z_n = Z[pn];
        for (;; ++item) {
        const T* vc = VC + (int64_t)__builtin_amdgcn_readfirstlane(uniform_n ? item : item_view[item]) * kViewStride;
        for (int q0 = qbeg; q0 < qend; q0 += 64) {
            const int q = q0 + sl;
            const T2 m = m_n, xy = xy_n;
            const T z = z_n;
            if (q0 + 64 < qend) {
                pn = pbeg + (q + 64 < qend ? q + 64 : qend - 1);
                m_n = uv[pn]; xy_n = XY[pn]; z_n = Z[pn];
            } else if (MULTI && item < item_last) {           // the next item's first batch (uniform items: qbeg = 0)
                pn = pbeg + n + (sl < n ? sl : n - 1);
                m_n = uv[pn]; xy_n = XY[pn]; z_n = Z[pn];
            }
            T u, v;
            T2 Jc[C];
            jacobian_point<MODEL, T>(sp, vc, xy.x, xy.y, z, u, v, Jc);
            // lanes past the item's end evaluate a clamped (finite) point; their rows are only ever
            // read as part of the last, partial 4-point group, where they are zeroed at the read
            T2 res;
            res.x = m.x - u;
            res.y = m.y - v;
            // The slab stores are the dearest part of a batch: a ds_write costs its ~13 cycles on a path all four SIMDs
            // of the CU share, with 64 or with 32 active lanes -- and a 32-row pass has only the 32 lanes that own its
            // rows to store. So the chunks travel in PAIRS: v_permlane32_swap exchanges the upper half of chunk A with
            // the lower half of chunk B; afterwards register A holds, for the rows of pass 0, chunk A in lanes 0-31
            // and chunk B in lanes 32-63 (lane l + 32 carries the chunk of lane l's row), register B the same for the
            // rows of pass 1. One full-width store per pair and pass instead of two half-empty ones: 8 instead of
            // 13 / 14 store instructions per pass for 4 vector instructions per pair and batch.
            constexpr int NCH = C - 5 + 1;                      // columns 5..C-1 and the residual
            T2 ch[NCH];
#pragma unroll
            for (int i = 0; i < C - 5; ++i) ch[i] = Jc[5 + i];
            ch[NCH - 1] = res;
            auto colOf = [](int i) { return i < C - 5 ? 5 + i : (RCOL ? 15 : 4); };
            if constexpr (HALVES == 2) {
#pragma unroll
                for (int i = 0; i + 1 < NCH; i += 2) {
                    unsigned a[sizeof(T2) / 4], b[sizeof(T2) / 4];
                    __builtin_memcpy(a, &ch[i], sizeof(T2));
                    __builtin_memcpy(b, &ch[i + 1], sizeof(T2));
#pragma unroll
                    for (int d = 0; d < (int)(sizeof(T2) / 4); ++d) {
                        const auto r = __builtin_amdgcn_permlane32_swap(a[d], b[d], false, false);
                        a[d] = r[0];
                        b[d] = r[1];
                    }
                    __builtin_memcpy(&ch[i], a, sizeof(T2));
                    __builtin_memcpy(&ch[i + 1], b, sizeof(T2));
                }
            }
#pragma unroll
            for (int half = 0; half < HALVES; ++half) {
                if (q0 + ROWS * half >= qend) break;            // wave-uniform
                __builtin_amdgcn_wave_barrier();
                T2* row = slab + rowOff(sl & (ROWS - 1));
                if (HALVES == 1 || (lane >> 5) == half) {
                    T* rh = reinterpret_cast<T*>(row);
                    rh[0] = Jc[0].x; rh[3] = Jc[1].y; rh[4] = Jc[2].x;  // the non-zero halves of columns 0, 1, 2
                }
                if constexpr (HALVES == 2) {
                    // pairs: every lane stores -- lanes 0-31 the pair's first chunk, lanes 32-63 its second, both for
                    // row sl & 31 of this pass; columns 3, 4 are constants, set once above
#pragma unroll
                    for (int i = 0; i + 1 < NCH; i += 2) row[lane < 32 ? colOf(i) : colOf(i + 1)] = ch[i + half];
                    if ((NCH & 1) && (lane >> 5) == half) row[colOf(NCH - 1)] = ch[NCH - 1];
                } else {
#pragma unroll
                    for (int i = 0; i < NCH; ++i) row[colOf(i)] = ch[i];
                }
                __builtin_amdgcn_wave_barrier();
                const int rows = qend - (q0 + ROWS * half);     // valid points in this pass (may exceed ROWS)
                if constexpr (MF32) {
                    // fp32 storage: the pass's 32 points are contracted by v_mfma_f32_16x16x4_f32 (twice the fp64
                    // rate) into fp32 tiles that start at zero, and the tiles are added to the fp64 accumulators:
                    // fp32 rounding stays confined to sums of 32 products
                    f4 fu = {0.f, 0.f, 0.f, 0.f}, fv = {0.f, 0.f, 0.f, 0.f};
                    if (rows >= ROWS) {
                        T2 jv[ROWS / 4];
#pragma unroll
                        for (int s = 0; s < ROWS / 4; ++s) jv[s] = slab[(4 * s + k) * RS + c];
#pragma unroll
                        for (int s = 0; s < ROWS / 4; ++s) {
                            fu = __builtin_amdgcn_mfma_f32_16x16x4f32(jv[s].x, jv[s].x, fu, 0, 0, 0);
                            fv = __builtin_amdgcn_mfma_f32_16x16x4f32(jv[s].y, jv[s].y, fv, 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int s = 0; s < ROWS / 4; ++s) {
                            if (4 * s >= rows) break;                   // wave-uniform
                            T2 ja = slab[(4 * s + k) * RS + c];
                            if (4 * s + k >= rows) { ja.x = T(0); ja.y = T(0); }    // points past the end of the last group
                            fu = __builtin_amdgcn_mfma_f32_16x16x4f32(ja.x, ja.x, fu, 0, 0, 0);
                            fv = __builtin_amdgcn_mfma_f32_16x16x4f32(ja.y, ja.y, fv, 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) { acc[reg] += (double)fu[reg]; acc2[reg] += (double)fv[reg]; }
                } else if constexpr (!G44) {
                    // fp64, 16x16x4 form (items of a single batch): ONE rolled loop over the pass's complete 4-point
                    // groups, full pass or not, keeps the two accumulator tiles in the same registers all the way
                    const int nfull = rows >= ROWS ? ROWS / 4 : rows >> 2;
                    const T2* src = slab + rowOff(k) + c;              // rows 4 s + k: + 66 chunks per group
                    for (int s = 0; s < nfull; ++s) {
                        const T2 ja = *src;
                        src += 66;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)ja.x, (double)ja.x, acc, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)ja.y, (double)ja.y, acc2, 0, 0, 0);
                    }
                    if (rows < ROWS && (rows & 3)) {            // wave-uniform: the last, incomplete group
                        const T2 ja = *src;
                        const bool live = 4 * nfull + k < rows;
                        const double jx = live ? (double)ja.x : 0.0, jy = live ? (double)ja.y : 0.0;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(jx, jx, acc, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(jy, jy, acc2, 0, 0, 0);
                    }
                } else {
                    // fp64, 4x4x4 form (items of several batches). With every CU issuing it, v_mfma_f64_16x16x4_f64 is
                    // held to ~47 TFLOP/s by the chip (tools/ubench/ubench6: 0.54 TFLOP/s per CU on a few CUs, 0.19 on
                    // all of them); v_mfma_f64_4x4x4_4b -- four independent 4x4x4 products per instruction -- sustains
                    // 76 (ubench8), and J^T J is symmetric. So the 16 x 16 tile is built from 4 x 4 blocks: lane
                    // (k, b, x) = (lane >> 4, (lane >> 2) & 3, lane & 3) supplies A = J[point k][4 b + x] and
                    // B = J[point k][4 b' + x] and receives D[4 b + k'][4 b' + x] in lane (k', b, x) (layout:
                    // tools/ubench/mfma4x4_layout). b' = b gives the four diagonal blocks, b' = b + 1 the blocks (0,1)
                    // (1,2) (2,3) (3,0), b' = b + 2 the blocks (0,2) (1,3) and their transposes: every block of the
                    // symmetric tile or its transpose. The A operand is the same (u, v) chunk the 16x16x4 form reads
                    // (column c = lane & 15 of point k); the B operands are columns c + 4 and c + 8 of the same row.
                    // Five instructions per group of 4 points: (b,b) and (b,b+1) for the u rows and for the v rows, and
                    // ONE for the (b,b+2) blocks: slots b = 0, 1 take them for the u rows, slots b = 2, 3 -- whose
                    // (b,b+2) blocks are the transposes (2,0), (3,1) of the same two -- for the v rows. Its operands are
                    // 8-byte reads of the lane's own chunk and of chunk c + 8 at the u or the v half (jh, per lane).
                    // 5 x 256 MACs per 4 points instead of 2 x 1024.
                    const int nfull = rows >= ROWS ? ROWS / 4 : rows >> 2;
                    const T2* src = slab + rowOff(k);                  // rows 4 s + k: + 66 chunks per group
                    const int c1 = (c + 4) & 15, c2 = (c + 8) & 15, jh = (lane >> 3) & 1;      // jh = 1: blocks 2, 3
                    const T* h0 = reinterpret_cast<const T*>(src + c) + jh;
                    const T* h2 = reinterpret_cast<const T*>(src + c2) + jh;
                    auto contract = [&](const T2& ja, const T2& jb, double ha, double hc) {
                        d0u = __builtin_amdgcn_mfma_f64_4x4x4f64((double)ja.x, (double)ja.x, d0u, 0, 0, 0);
                        d0v = __builtin_amdgcn_mfma_f64_4x4x4f64((double)ja.y, (double)ja.y, d0v, 0, 0, 0);
                        d1u = __builtin_amdgcn_mfma_f64_4x4x4f64((double)ja.x, (double)jb.x, d1u, 0, 0, 0);
                        d1v = __builtin_amdgcn_mfma_f64_4x4x4f64((double)ja.y, (double)jb.y, d1v, 0, 0, 0);
                        d2u = __builtin_amdgcn_mfma_f64_4x4x4f64(ha, hc, d2u, 0, 0, 0);
                    };
                    if (rows >= ROWS) {
                        // full pass: every address is the wave's constant base plus an immediate
#pragma unroll
                        for (int s = 0; s < ROWS / 4; ++s)
                            contract(src[66 * s + c], src[66 * s + c1], (double)h0[2 * 66 * s], (double)h2[2 * 66 * s]);
                    } else {
                        for (int s = 0; s < nfull; ++s)
                            contract(src[66 * s + c], src[66 * s + c1], (double)h0[2 * 66 * s], (double)h2[2 * 66 * s]);
                        if (rows & 3) {                                 // wave-uniform: the last, incomplete group
                            const bool live = 4 * nfull + k < rows;
                            T2 ja = src[66 * nfull + c], jb = src[66 * nfull + c1];
                            double ha = (double)h0[2 * 66 * nfull], hc = (double)h2[2 * 66 * nfull];
                            if (!live) { ja.x = T(0); ja.y = T(0); jb.x = T(0); jb.y = T(0); ha = 0.0; hc = 0.0; }
                            contract(ja, jb, ha, hc);
                        }
                    }
                }
            }
        }
        if (!MULTI || item >= item_last) break;
        // more items for this wave: this one's record leaves now, then everything that belongs to its view -- rows
        // and columns L..L+5 of both tiles -- starts again from zero; the shared block, g_c and sum r^2 go on
        emitDirect(item);
        {
            constexpr int L = C - 6;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = MF32 ? 4 * k + reg : k + 4 * reg;
                const bool ofView = (row >= L && row < L + 6) || (c >= L && c < L + 6);
                acc[reg] = ofView ? 0.0 : acc[reg];
                acc2[reg] = ofView ? 0.0 : acc2[reg];
            }
        }
        pbeg += n;
        }
    }
    // Output assembly: every wave parks its two accumulator tiles (f64 MFMA C/D layout: col = lane & 15,
    // row = (lane >> 4) + 4 * reg; zeros for a wave without an item) in its dead slab; then the
    // workgroup writes its partial of the shared block and each item's first wave the item's record,
    // both from the index table -- no case analysis, 16-byte coalesced stores.
    static_assert(SLAB * sizeof(T2) >= 2 * kEmitTile * 8, "tiles must fit the wave's slab");
    double* TU = reinterpret_cast<double*>(slab);
    double* TV = TU + kEmitTile;
    if (!G44 && wpi == 1 && valid) emitDirect(item);            // the (last) item's record
    __builtin_amdgcn_wave_barrier();                            // the slab is this wave's own: no workgroup barrier needed yet
    if constexpr (!G44) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            // C/D register -> tile row: fp64 MFMA k + 4 reg, fp32 MFMA (whose tiles the accumulators mirror) 4 k + reg
            const int row = MF32 ? 4 * k + reg : k + 4 * reg;
            TU[row * 16 + c] = acc[reg];
            TV[row * 16 + c] = acc2[reg];
        }
    } else {
        // block results -> the full symmetric 16 x 16 tiles: lane (i, b, j) holds entry (4 b + i, 4 b' + j) of block (b, b')
        const int bi = (lane >> 2) & 3, j = lane & 3;
        const int row = 4 * bi + k, col0 = 4 * bi + j, col1 = 4 * ((bi + 1) & 3) + j, col2 = 4 * ((bi + 2) & 3) + j;
        TU[row * 16 + col0] = d0u;  TV[row * 16 + col0] = d0v;
        TU[row * 16 + col1] = d1u;  TV[row * 16 + col1] = d1v;
        TU[col1 * 16 + row] = d1u;  TV[col1 * 16 + row] = d1v;  // the transposes of the (b, b+1) blocks
        // (b, b+2): lanes of blocks 0, 1 hold the u rows' (0,2), (1,3), lanes of blocks 2, 3 the v rows' (2,0), (3,1)
        double* T2nd = bi < 2 ? TU : TV;
        T2nd[row * 16 + col2] = d2u;
        T2nd[col2 * 16 + row] = d2u;
    }
    if (lane == 0) { TU[kEmitZero] = 0.0; TV[kEmitZero] = 0.0; }
    __syncthreads();
    if ((int)threadIdx.x < kPartStride) {                       // partial of B, g_c, sum r^2: all waves, wave order
        const uint32_t t = emit_tab[kGStride + threadIdx.x];
        const double* T0 = reinterpret_cast<const double*>(smem);
        double o = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const double* TUw = reinterpret_cast<const double*>(reinterpret_cast<const T2*>(T0) + w * SLAB);
            o += TUw[t & 0xffff] + (TUw + kEmitTile)[t >> 16];
        }
        part[(int64_t)blockIdx.x * kPartStride + threadIdx.x] = o;
    }
    if ((!G44 && wpi == 1) || sub != 0 || !valid) return;       // the item's first wave assembles the record from the parked tiles
    double* G = Gbase + (int64_t)item * kGStride;
    if (lane < kGStride / 2) {
        const int i0 = 2 * lane;                                // kGStride = 96: lanes 0..47, one pass
        const uint2 t = *reinterpret_cast<const uint2*>(emit_tab + i0);
        double2 o;
        o.x = TU[t.x & 0xffff] + TV[t.x >> 16];
        o.y = TU[t.y & 0xffff] + TV[t.y >> 16];
        for (int w = 1; w < wpi; ++w) {                         // partial tiles of the item's other waves, in wave order
            const double* TUw = reinterpret_cast<const double*>(slab + w * SLAB);
            const double* TVw = TUw + kEmitTile;
            o.x += TUw[t.x & 0xffff] + TVw[t.x >> 16];
            o.y += TUw[t.y & 0xffff] + TVw[t.y >> 16];
        }
        *reinterpret_cast<double2*>(G + i0) = o;
    }
}

#ifdef CALIB_STREAM_STAMPS
// diagnostic build only (tools/diag/build_stream_stamps.sh): per-wave s_memtime deltas of fused_stream_kernel's phases
constexpr int kSStampWaves = 8192, kSStampSlots = 16;
__device__ unsigned long long g_sstamps[kSStampWaves * kSStampSlots];
#define SSTAMP(i) do { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); const unsigned long long t__ = __builtin_amdgcn_s_memtime(); tacc[i] += t__ - tlast; tlast = t__; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SSTAMP(i) do {} while (0)
#endif
// ---------------------------------------------------------------- fused jacobian + gram, stream form
// The block-form fused kernel for uniform fp64 shards (every view n points, n a multiple of 4 and >= 64), with the
// per-view fixed cost and the dead lanes of a view's last batch taken out. The shard is ONE stream of 4-point
// groups; wave w of the launch takes groups [w share, (w + 1) share) -- the host sizes the launch to the wave
// slots of the chip, so every SIMD gets the same number of groups whatever the view count (10 000 views of 200 points
// on 4 096 slots used to be 2.44 rounds of one view per wave: three rounds). A wave pays ONE prologue, one first
// wait, one workgroup partial; its batches are always 64 live points (but for the share's last), view boundaries
// fall where they fall:
//  * a batch inside one view evaluates its points with the view's constants in SGPRs -- requested right after the
//    previous batch's Jacobian arithmetic is done with the registers, so they arrive while that batch is contracted;
//    a batch that straddles a boundary uses per-lane constants: lanes 0..35 fetch the 36 doubles of the two views (one
//    load, requested a batch ahead), stage them in 288 B of LDS, and every lane reads the 18 of its own view;
//  * in the contraction, when the next group is the first of a new view, the finished view's record leaves STRAIGHT
//    FROM THE ACCUMULATORS (no tile parking, no barrier): every lane knows from a table (buildStreamOps, 48 B per
//    lane, kept in LDS) where in the 1 KiB record each of its block entries belongs -- once, or twice for an
//    entry whose row AND column are view parameters -- and stores them there; the (b, b+2) blocks, whose u rows' and
//    v rows' sums live in two different lanes, are completed with one ds_bpermute. Then everything that belongs
//    to a view (rows and columns L..L+5) restarts from zero; the shared block, g_c and sum r^2 go on for the share;
//  * records: view v's goes to slot v from the wave that holds the view's first group; what a wave sums of the
//    view its share starts in the middle of goes to its overflow slot nv + w (StreamMap / stream_extra_item:
//    the per-view kernels add the two). Only the entries the per-view kernels read are written: rows L..L+5 of
//    J^T J and the view's six entries of J^T r (the record buffers are zeroed when they are allocated).
// Everything else -- slab layout, chunk pairs, the five 4x4x4 instructions per group -- is fused_kernel's G44 form.
constexpr int kStreamOps = 12;       // byte offsets into the record per lane; kStreamNoOp (past the record) = nothing
constexpr int kStreamNoOp = 0x7ffffff0;
// op -> what is stored: 0, 1: d0u + d0v   2, 3: d1u + d1v   4, 5: d2 + the partner lane's d2
//                       6: d0u  7: d0v  8: d1u  9: d1v  10: d2   (radial-tangential: the (1,1) column, see fused_kernel)
inline bool buildStreamOps(int C, int32_t* ops /* 64 * kStreamOps */) {
    uint32_t tab[kEmitTabSize];
    buildEmitTable(C, tab);
    for (int i = 0; i < 64 * kStreamOps; ++i) ops[i] = kStreamNoOp;
    struct Src { int lane, acc; };          // acc: 0 d0u, 1 d0v, 2 d1u, 3 d1v, 4 d2
    const auto laneOf = [](int k, int b, int x) { return k * 16 + b * 4 + x; };
    // who holds entry idx = row * 16 + col of the u rows' (v rows') tile: see the block layout in fused_kernel
    const auto locate = [&](int idx, bool vtile) -> Src {
        const int r = idx / 16, c = idx % 16, br = r / 4, bc = c / 4;
        if (bc == br) return {laneOf(r % 4, br, c % 4), vtile ? 1 : 0};
        if (bc == ((br + 1) & 3)) return {laneOf(r % 4, br, c % 4), vtile ? 3 : 2};
        if (br == ((bc + 1) & 3)) return {laneOf(c % 4, bc, r % 4), vtile ? 3 : 2};
        const bool own = vtile ? br >= 2 : br < 2;      // (b, b+2): u rows in lanes of blocks 0, 1, v rows in 2, 3
        return own ? Src{laneOf(r % 4, br, c % 4), 4} : Src{laneOf(c % 4, bc, r % 4), 4};
    };
    bool ok = true;
    const auto put = [&](int lane, int first, int count, int slot) {
        for (int j = first; j < first + count; ++j)
            if (ops[lane * kStreamOps + j] == kStreamNoOp) { ops[lane * kStreamOps + j] = slot * 8; return; }
        ok = false;
    };
    for (int slot = 0; slot < kGStride; ++slot) {
        const int iu = (int)(tab[slot] & 0xffff), iv = (int)(tab[slot] >> 16);
        if (iu == kEmitZero && iv == kEmitZero) continue;
        if (iu != kEmitZero && iv != kEmitZero) {
            const Src su = locate(iu, false), sv = locate(iv, true);
            if (iu != iv) { ok = false; continue; }
            if (su.lane == sv.lane) put(su.lane, su.acc == 0 ? 0 : 2, 2, slot);
            else if (su.acc == 4 && sv.acc == 4) put(su.lane, 4, 2, slot);
            else ok = false;
        } else if (iu != kEmitZero) {
            const Src su = locate(iu, false);
            put(su.lane, su.acc == 0 ? 6 : (su.acc == 2 ? 8 : 10), 1, slot);
        } else {
            const Src sv = locate(iv, true);
            put(sv.lane, sv.acc == 1 ? 7 : (sv.acc == 3 ? 9 : 10), 1, slot);
        }
    }
    return ok;
}

#ifndef CALIB_STREAM_MIN_BLOCKS
#define CALIB_STREAM_MIN_BLOCKS 4       // workgroups per CU the register allocation leaves room for (A/B builds: 3, 5)
#endif
template <int MODEL>
__global__ __launch_bounds__(256, CALIB_STREAM_MIN_BLOCKS) void fused_stream_kernel(const double* __restrict__ P0, const double* __restrict__ P1,
                                                              const double2* __restrict__ uv, const double2* __restrict__ XY,
                                                              const double* __restrict__ Z, const double* __restrict__ VC,
                                                              int n, int nv, int share,
                                                              const uint32_t* __restrict__ emit_tab,
                                                              const int32_t* __restrict__ stream_ops,
                                                              const LMState* __restrict__ st, int sel,
                                                              double* __restrict__ G0, double* __restrict__ G1,
                                                              double* __restrict__ part) {
    using T = double;
    using T2 = double2;
    constexpr int C = ModelTraits<MODEL>::C, L = C - 6;
    constexpr int ROWS = 32, WAVES = 4;
    constexpr int SLAB = ROWS * 16 + ROWS / 2;
    constexpr bool RCOL = C < 16, ONES = !RCOL;
    auto rowOff = [](int r) { return r * 16 + (r >> 1); };
    __shared__ __attribute__((aligned(16))) unsigned char smem[WAVES * SLAB * sizeof(T2)];
    __shared__ __attribute__((aligned(16))) double svc[WAVES][2 * kViewStride];
    __shared__ __attribute__((aligned(16))) int32_t sops[64 * kStreamOps];
    static_assert(SLAB * sizeof(T2) >= 2 * kEmitTile * 8, "tiles must fit the wave's slab");
#ifdef CALIB_STREAM_STAMPS
    unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tstart = __builtin_amdgcn_s_memtime();
    unsigned long long tlast = tstart;
#endif
    if (sel && st->done) return;
    const double* P = selectP(P0, P1, st, sel);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, k = lane >> 4;
    const int sl = (lane & 48) | ((lane & 7) << 1) | ((lane >> 3) & 1);
    T2* slab = reinterpret_cast<T2*>(smem) + wave * SLAB;
    double d0u = 0.0, d0v = 0.0, d1u = 0.0, d1v = 0.0, d2u = 0.0;
    double* Gbase = sel ? ((st->cur ^ 1) ? G1 : G0) : G0;
    // block results: lane (i, b, j) = (k, bi, bj) holds entry (4 b + i, 4 b' + j) of block (b, b'), b' = b, b + 1, b + 2
    const int bi = (lane >> 2) & 3, bj = lane & 3;
    const int trow = 4 * bi + k, tcol0 = 4 * bi + bj, tcol1 = 4 * ((bi + 1) & 3) + bj, tcol2 = 4 * ((bi + 2) & 3) + bj;
    auto ofView = [](int i) { return i >= L && i < L + 6; };
    const bool keep0 = !(ofView(trow) || ofView(tcol0)), keep1 = !(ofView(trow) || ofView(tcol1)),
               keep2 = !(ofView(trow) || ofView(tcol2));
    // every wave writes the (same) whole table and every lane only ever reads back its own 24 bytes: no barrier
    {
        const int4* src = reinterpret_cast<const int4*>(stream_ops) + 3 * lane;
        int4* dst = reinterpret_cast<int4*>(sops) + 3 * lane;
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
    }
    const int partner = bj * 16 + ((bi + 2) & 3) * 4 + k;       // holds the other rows' sums of this lane's (b, b+2) entry
    // the finished view's record, straight from the accumulators; then what belongs to a view restarts from zero
    // Every op is ONE buffer store with the lane's byte offset in the voffset: the record is a 1 KiB buffer resource, and
    // an offset past it (kStreamNoOp: "this lane has nothing for this op") is dropped by the hardware's range check -- no
    // compare, no branch, no address arithmetic (the first version unpacked int16 offsets and branched around 11 flat
    // stores: ~70 vector instructions per view, 20 % of c5's).
    auto emitView = [&](int slot) {
        typedef unsigned int u2v __attribute__((ext_vector_type(2)));
        const unsigned long long gp = reinterpret_cast<unsigned long long>(Gbase + (int64_t)slot * kGStride);
        // (readfirstlane returns int: both halves through unsigned, or a set bit 31 of the low half smears into the high one)
        const unsigned glo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)gp);
        const unsigned ghi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(gp >> 32));
        double* G = reinterpret_cast<double*>(((unsigned long long)ghi << 32) | (unsigned long long)glo);
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(G, 0, kGStride * 8, 0x00020000);
        const int4* o4 = reinterpret_cast<const int4*>(sops) + 3 * lane;
        const int4 oa = o4[0], ob = o4[1];
        const double s0 = d0u + d0v, s1 = d1u + d1v;
        const double dp = __hiloint2double(__builtin_amdgcn_ds_bpermute(partner << 2, __double2hiint(d2u)),
                                           __builtin_amdgcn_ds_bpermute(partner << 2, __double2loint(d2u)));
        const double s2 = d2u + dp;
        auto put = [&](int off, double val) {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, val), rsrc, off, 0, 0);
        };
        put(oa.x, s0); put(oa.y, s0);
        put(oa.z, s1); put(oa.w, s1);
        put(ob.x, s2); put(ob.y, s2);
        if (ONES) {
            const int4 oc = o4[2];
            put(ob.z, d0u); put(ob.w, d0v);
            put(oc.x, d1u); put(oc.y, d1v);
            put(oc.z, d2u);
        }
        d0u = keep0 ? d0u : 0.0;  d0v = keep0 ? d0v : 0.0;
        d1u = keep1 ? d1u : 0.0;  d1v = keep1 ? d1v : 0.0;
        d2u = keep2 ? d2u : 0.0;
    };
    const int n4 = n >> 2;
    const int w = blockIdx.x * WAVES + wave;
    const long long gt = (long long)nv * n4, g0 = (long long)w * share;
    const bool valid = g0 < gt;
    if (valid) {
        const int p0 = (int)(4 * g0);
        const int p1 = (int)(4 * (g0 + share < gt ? g0 + share : gt));
        int va = __builtin_amdgcn_readfirstlane(p0 / n);      // the view being accumulated
        int vend = (va + 1) * n;                              // its end = the next boundary, in points
        int slot = p0 != va * n ? nv + w : va;
        Shared<MODEL, T> sp;
        sp.load(P);
        // the constant parts of the slab rows (see fused_kernel)
        if (lane < ROWS) {
            T2 c3, c4;
            c3.x = T(1); c3.y = ONES ? T(1) : T(0);
            c4.x = T(0); c4.y = T(1);
            slab[rowOff(lane) + 3] = c3;
            if (!ONES) slab[rowOff(lane) + 4] = c4;
            T* zr = reinterpret_cast<T*>(slab + rowOff(lane));
            zr[1] = T(0); zr[2] = T(0); zr[5] = T(0);
        }
        int pn = p0 + sl < p1 ? p0 + sl : p1 - 1;
        T2 m_n = uv[pn], xy_n = XY[pn];
        T z_n = Z[pn];
        // does the batch at q0 hold points of two views (boundary ve strictly inside its live points)?
        auto straddles = [&](int q0, int ve) { return ve > q0 && ve < (q0 + 64 < p1 ? q0 + 64 : p1); };
        const int vcLast = nv * kViewStride - 1;
        auto stagedLoad = [&](int view) {                       // lanes 0..35: the 36 doubles of views (view, view + 1)
            const int i = view * kViewStride + (lane < 2 * kViewStride ? lane : 2 * kViewStride - 1);
            return VC[i < vcLast ? i : vcLast];
        };
        double vc_n = 0.0;
        double vcs[kViewStride];                               // one-view batches: the view's constants (SGPRs)
#pragma unroll
        for (int i = 0; i < kViewStride; ++i) vcs[i] = 0.0;
        auto scalarLoad = [&](int view) {
            const T* s = VC + (int64_t)__builtin_amdgcn_readfirstlane(view) * kViewStride;
#pragma unroll
            for (int i = 0; i < kViewStride - 1; ++i) vcs[i] = s[i];
        };
        bool strad = straddles(p0, vend);
        if (strad) vc_n = stagedLoad(va); else scalarLoad(va);
        const T2* src = slab + rowOff(k);                      // rows 4 s + k: + 66 chunks per group
        const int c1 = (c + 4) & 15, c2 = (c + 8) & 15, jh = (lane >> 3) & 1;
        const T* h0 = reinterpret_cast<const T*>(src + c) + jh;
        const T* h2 = reinterpret_cast<const T*>(src + c2) + jh;
        auto contract = [&](const T2& ja, const T2& jb, double ha, double hc) {
            d0u = __builtin_amdgcn_mfma_f64_4x4x4f64(ja.x, ja.x, d0u, 0, 0, 0);
            d0v = __builtin_amdgcn_mfma_f64_4x4x4f64(ja.y, ja.y, d0v, 0, 0, 0);
            d1u = __builtin_amdgcn_mfma_f64_4x4x4f64(ja.x, jb.x, d1u, 0, 0, 0);
            d1v = __builtin_amdgcn_mfma_f64_4x4x4f64(ja.y, jb.y, d1v, 0, 0, 0);
            d2u = __builtin_amdgcn_mfma_f64_4x4x4f64(ha, hc, d2u, 0, 0, 0);
        };
        PointState<MODEL, T> pst;
        SSTAMP(0);
        for (int q0 = p0; q0 < p1; q0 += 64) {
            const int qe = q0 + 64 < p1 ? q0 + 64 : p1;
            const int q = q0 + sl;
            const T2 m = m_n, xy = xy_n;
            const T z = z_n;
            const double vcx = vc_n;
            // a boundary inside this batch (or at its start) is passed during it: what the next batch will see
            const bool more = q0 + 64 < p1;
            const bool passes = vend < qe;
            const int ve2 = passes ? vend + n : vend, va2 = passes ? va + 1 : va;
            const bool strad2 = more && straddles(q0 + 64, ve2);
            const int vfirst2 = ve2 <= q0 + 64 ? va2 + 1 : va2;
            if (more) {
                pn = q + 64 < p1 ? q + 64 : p1 - 1;
                m_n = uv[pn]; xy_n = XY[pn]; z_n = Z[pn];
                if (strad2) vc_n = stagedLoad(va2);
#ifdef CALIB_STREAM_STAMPS
                __builtin_amdgcn_s_waitcnt(0x0F74);     // vmcnt(4): this batch's points have arrived
#endif
            } else {
#ifdef CALIB_STREAM_STAMPS
                __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
            }
            SSTAMP(1);
            T u, v;
            T2 Jc[C];
            if (!strad) {
                jacobian_stage_a<MODEL, T>(sp, vcs, xy.x, xy.y, z, pst);
                jacobian_stage_b<MODEL, T>(sp, vcs, pst, u, v, Jc);
            } else {
                // two views: every lane reads the constants of its own from the staged pair
                double* sv = svc[wave];
                if (lane < 2 * kViewStride) sv[lane] = vcx;
                __builtin_amdgcn_wave_barrier();
                const int qc = q < p1 ? q : p1 - 1;               // clamped lanes follow the share's last point
                const double2* s2 = reinterpret_cast<const double2*>(sv + (qc >= vend ? kViewStride : 0));
                T vcl[kViewStride];
#pragma unroll
                for (int i = 0; i < kViewStride / 2; ++i) {
                    const double2 t = s2[i];
                    vcl[2 * i] = t.x;
                    vcl[2 * i + 1] = t.y;
                }
                __builtin_amdgcn_wave_barrier();
                jacobian_point<MODEL, T>(sp, vcl, xy.x, xy.y, z, u, v, Jc);
            }
            T2 res;
            res.x = m.x - u;
            res.y = m.y - v;
#ifdef CALIB_STREAM_STAMPS
            asm volatile("" :: "v"(res.x), "v"(res.y), "v"(Jc[C - 1].x), "v"(Jc[C - 1].y));
            if (strad) SSTAMP(3); else SSTAMP(2);
#endif
            // the SGPRs of the view constants are free now: the next one-view batch's arrive during the contraction
            if (more && !strad2) scalarLoad(vfirst2);
            strad = strad2;
            // chunk pairs through v_permlane32_swap: one full-width store per pair and pass (see fused_kernel)
            constexpr int NCH = C - 5 + 1;                      // columns 5..C-1 and the residual
            T2 ch[NCH];
#pragma unroll
            for (int i = 0; i < C - 5; ++i) ch[i] = Jc[5 + i];
            ch[NCH - 1] = res;
            auto colOf = [](int i) { return i < C - 5 ? 5 + i : (RCOL ? 15 : 4); };
#pragma unroll
            for (int i = 0; i + 1 < NCH; i += 2) {
                unsigned a[4], b[4];
                __builtin_memcpy(a, &ch[i], sizeof(T2));
                __builtin_memcpy(b, &ch[i + 1], sizeof(T2));
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const auto r = __builtin_amdgcn_permlane32_swap(a[d], b[d], false, false);
                    a[d] = r[0];
                    b[d] = r[1];
                }
                __builtin_memcpy(&ch[i], a, sizeof(T2));
                __builtin_memcpy(&ch[i + 1], b, sizeof(T2));
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int ps = q0 + ROWS * half;                // first point of the pass
                if (ps >= p1) break;                            // wave-uniform
                __builtin_amdgcn_wave_barrier();
                T2* row = slab + rowOff(sl & (ROWS - 1));
                if ((lane >> 5) == half) {
                    T* rh = reinterpret_cast<T*>(row);
                    rh[0] = Jc[0].x; rh[3] = Jc[1].y; rh[4] = Jc[2].x;  // the non-zero halves of columns 0, 1, 2
                }
#pragma unroll
                for (int i = 0; i + 1 < NCH; i += 2) row[lane < 32 ? colOf(i) : colOf(i + 1)] = ch[i + half];
                if ((NCH & 1) && (lane >> 5) == half) row[colOf(NCH - 1)] = ch[NCH - 1];
                __builtin_amdgcn_wave_barrier();
                SSTAMP(4);
                const int ng = (p1 - ps >= ROWS ? ROWS : p1 - ps) >> 2;   // groups of the pass (shares are whole groups)
                const int jb = (vend - ps) >> 2;                // the group a new view starts with (>= ng: none here)
                // groups [FROM, TO) of the pass, FROM and TO known at compile time: every address is the wave's constant base
                // plus an immediate, and the operands of group s + 1 are requested before group s is contracted
                auto runFixed = [&](auto FROMc, auto TOc) {
                    constexpr int FROM = decltype(FROMc)::value, TO = decltype(TOc)::value;
                    if constexpr (FROM < TO) {
                        T2 ja = src[66 * FROM + c], jbb = src[66 * FROM + c1];
                        double ha = h0[2 * 66 * FROM], hc = h2[2 * 66 * FROM];
#pragma unroll
                        for (int s = FROM; s < TO; ++s) {
                            T2 na = ja, nb = jbb;
                            double nha = ha, nhc = hc;
                            if (s + 1 < TO) {
                                na = src[66 * (s + 1) + c]; nb = src[66 * (s + 1) + c1];
                                nha = h0[2 * 66 * (s + 1)]; nhc = h2[2 * 66 * (s + 1)];
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            contract(ja, jbb, ha, hc);
                            __builtin_amdgcn_sched_barrier(0);
                            ja = na; jbb = nb; ha = nha; hc = nhc;
                        }
                    }
                };
                auto nextView = [&]() {                         // the next group is the first of a new view
                    emitView(slot);
                    va += 1;
                    vend += n;
                    slot = va;
                };
                constexpr int NG = ROWS / 4;
                using std::integral_constant;
                if (ng == NG && jb >= NG) {
                    // a whole pass inside one view
                    runFixed(integral_constant<int, 0>{}, integral_constant<int, NG>{});
#ifdef CALIB_STREAM_STAMPS
                    asm volatile("" :: "v"(d0u), "v"(d0v), "v"(d1u), "v"(d1v), "v"(d2u));
#endif
                    SSTAMP(5);
                } else if (ng == NG) {
                    // a whole pass with a view boundary before group jb: one straight-line copy of the pass per position of
                    // the boundary -- groups [0, jb), the finished view's record, groups [jb, 8) -- so that each run keeps
                    // the operand pipeline of the whole-view pass. (Round 3 entered a switch at the run's first group and
                    // every group loaded its operands right before its five MFMAs, one exposed LDS latency per group: on c5
                    // these passes were 36 % of the passes and 19 % of the kernel, twice as dear as the others. Run bounds
                    // known only at run time do not help: behind a conditional request the compiler cannot count what is
                    // in flight and waits for everything, lgkmcnt(0), before every group.)
                    auto boundaryAt = [&](auto JBc) {
                        constexpr int JB = decltype(JBc)::value;
                        runFixed(integral_constant<int, 0>{}, integral_constant<int, JB>{});
#ifdef CALIB_STREAM_STAMPS
                        asm volatile("" :: "v"(d0u), "v"(d0v), "v"(d1u), "v"(d1v), "v"(d2u));
#endif
                        SSTAMP(6);
                        nextView();
                        SSTAMP(7);
                        runFixed(integral_constant<int, JB>{}, integral_constant<int, NG>{});
#ifdef CALIB_STREAM_STAMPS
                        asm volatile("" :: "v"(d0u), "v"(d0v), "v"(d1u), "v"(d1v), "v"(d2u));
#endif
                        SSTAMP(6);
                    };
                    switch (jb) {
                    case 0: boundaryAt(integral_constant<int, 0>{}); break;
                    case 1: boundaryAt(integral_constant<int, 1>{}); break;
                    case 2: boundaryAt(integral_constant<int, 2>{}); break;
                    case 3: boundaryAt(integral_constant<int, 3>{}); break;
                    case 4: boundaryAt(integral_constant<int, 4>{}); break;
                    case 5: boundaryAt(integral_constant<int, 5>{}); break;
                    case 6: boundaryAt(integral_constant<int, 6>{}); break;
                    default: boundaryAt(integral_constant<int, 7>{}); break;
                    }
                } else {
                    // the share's last, partial pass (once per wave): groups [from, to) entered through a switch
                    const int sb = jb < ng ? jb : ng;
                    auto runGroups = [&](int from, int to) {
                        switch (from) {
#define CALIB_GROUP_CASE(S) case S: if (to <= S) break; contract(src[66 * S + c], src[66 * S + c1], h0[2 * 66 * S], h2[2 * 66 * S]); [[fallthrough]];
                        CALIB_GROUP_CASE(0) CALIB_GROUP_CASE(1) CALIB_GROUP_CASE(2) CALIB_GROUP_CASE(3)
                        CALIB_GROUP_CASE(4) CALIB_GROUP_CASE(5) CALIB_GROUP_CASE(6) CALIB_GROUP_CASE(7)
#undef CALIB_GROUP_CASE
                        default: break;
                        }
                    };
                    runGroups(0, sb);
#ifdef CALIB_STREAM_STAMPS
                    asm volatile("" :: "v"(d0u), "v"(d0v), "v"(d1u), "v"(d1v), "v"(d2u));
#endif
                    SSTAMP(6);
                    if (jb < ng) {
                        nextView();
                        SSTAMP(7);
                        runGroups(sb, ng);
#ifdef CALIB_STREAM_STAMPS
                        asm volatile("" :: "v"(d0u), "v"(d0v), "v"(d1u), "v"(d1v), "v"(d2u));
#endif
                        SSTAMP(6);
                    }
                }
            }
        }
        // the share's last view (whole, or cut by the share's end)
        emitView(slot);
    }
    // the workgroup's partial of B, g_c, sum r^2: every wave parks its block accumulators as the two full symmetric
    // 16 x 16 tiles in its (idle) slab; 112 threads add the four waves' entries through the index table
    {
        double* TU = reinterpret_cast<double*>(slab);
        double* TV = TU + kEmitTile;
        __builtin_amdgcn_wave_barrier();
        TU[trow * 16 + tcol0] = d0u;  TV[trow * 16 + tcol0] = d0v;
        TU[trow * 16 + tcol1] = d1u;  TV[trow * 16 + tcol1] = d1v;
        TU[tcol1 * 16 + trow] = d1u;  TV[tcol1 * 16 + trow] = d1v;
        double* T2nd = bi < 2 ? TU : TV;
        T2nd[trow * 16 + tcol2] = d2u;
        T2nd[tcol2 * 16 + trow] = d2u;
        if (lane == 0) { TU[kEmitZero] = 0.0; TV[kEmitZero] = 0.0; }
    }
    SSTAMP(8);
    __syncthreads();
    SSTAMP(9);
    if ((int)threadIdx.x < kPartStride) {
        const uint32_t t = emit_tab[kGStride + threadIdx.x];
        const double* T0 = reinterpret_cast<const double*>(smem);
        double o = 0.0;
#pragma unroll
        for (int wv = 0; wv < WAVES; ++wv) {
            const double* TUw = reinterpret_cast<const double*>(reinterpret_cast<const T2*>(T0) + wv * SLAB);
            o += TUw[t & 0xffff] + (TUw + kEmitTile)[t >> 16];
        }
        part[(int64_t)blockIdx.x * kPartStride + threadIdx.x] = o;
    }
#ifdef CALIB_STREAM_STAMPS
    SSTAMP(10);
    if (lane == 0 && w < kSStampWaves) {
        unsigned long long* o = g_sstamps + (size_t)w * kSStampSlots;
        for (int i = 0; i < 12; ++i) o[i] = tacc[i];
        o[12] = tstart; o[13] = tlast; o[14] = 1;
    }
#endif
}

// ---------------------------------------------------------------- stream form: which records make up a view
// fused_stream_kernel deals the shard's 4-point groups out to its waves in equal shares of `share` groups, whatever
// the view boundaries. The record of view v (n4 groups) is written by the wave that holds the view's first group; a
// wave whose share starts strictly inside a view writes what it sums of that view to its own overflow record nv + w.
// With share >= n4 at most one wave starts inside a view, so a view is one or two records, found by arithmetic.
// The per-view kernels are instantiated with and without the second record (STREAM): the narrow-load forms sit at the
// 128-VGPR limit of four workgroups per CU, and the extra loads spilled them for every shard that has no cut views.
struct StreamMap { int share, n4, nv; };     // share == 0: the shard's records are not in stream form
__host__ __device__ __forceinline__ int stream_extra_item(const StreamMap& sm, int v) {
    if (sm.share == 0) return -1;
    const unsigned first = (unsigned)v * (unsigned)sm.n4, last = first + (unsigned)sm.n4 - 1u;
    const unsigned w = last / (unsigned)sm.share;             // the wave that holds the view's last group
    return w * (unsigned)sm.share > first ? sm.nv + (int)w : -1;
}

// ---------------------------------------------------------------- per-view elimination
// 16 lanes per view, working from the head of the view's record(s): lane c < L owns row c of E, lane L the view gradient.
__device__ __forceinline__ constexpr int tri(int m, int n) { return m * (m + 1) / 2 + n; }

// 1/sqrt(d) for the Cholesky pivots: v_rsq_f64 seed (~2^-26 relative) and ONE third-order step,
// r (1 + e/2 + 3 e^2/8) with e = 1 - d r^2 (error ~ e^3: below 2^-70) -- a chain of 6 dependent operations (two Newton
// steps were 8; sqrt() followed by a division ~40); the six pivots of a 6x6 block are a view's longest chain
__device__ __forceinline__ double rsqrt_nr(double d) {
    const double r = __builtin_amdgcn_rsq(d);
    const double e = __builtin_fma(-(d * r), r, 1.0);
    const double p = __builtin_fma(0.375, e, 0.5);
    return __builtin_fma(r * e, p, r);
}

// lane N of every 16-lane DPP row to all lanes of that row (row_newbcast, gfx90a+)
template <int N>
__device__ __forceinline__ double row_bcast(double v) {
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + N, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + N, 0xf, 0xf, false));
}

// The head of a view's record(s) as the elimination wants it: V (lower triangle of the 6x6 view block,
// the same for all 16 lanes) and this lane's right-hand side b -- lane c < L: row c of E (E[c][m] = R[m][c]);
// lanes >= L: the view gradient g_v. Two forms. WIDE (large shards, throughput-bound): lane c loads column c of
// the six record rows and of the gradient row -- 7 loads of 128 contiguous bytes per view -- and the 21 + 6
// values every lane needs from columns L..L+5 are broadcast inside the 16-lane group (DPP); c5 schur 61 -> 53 us.
// Otherwise (latency-bound shards): one load per value, 27 per lane with 21 of them the same address in all 16
// lanes -- more work for the address coalescer, but no chain of 54 DPP moves behind the loads (c3 / c4 -0.9 us).
// WIDE form in two steps, so that a kernel can have the next view's rows in flight while it factors this one:
// the raw column-c elements of the six rows and of the gradient row (summed over the view's items) ...
__device__ __forceinline__ void request_head_rows(const double* __restrict__ G, int item0, int nitems, int extra, int c,
                                                  double (&r)[6]) {
    const double* g = G + (int64_t)item0 * kGStride;
#pragma unroll
    for (int m = 0; m < 6; ++m) r[m] = g[kGRows + m * 16 + c];
    for (int it = 1; it < nitems; ++it) {                     // > 1 item only for views above kGramChunk points
        g += kGStride;
#pragma unroll
        for (int m = 0; m < 6; ++m) r[m] += g[kGRows + m * 16 + c];
    }
    if (extra >= 0) {                                         // stream form: the part of the view a second wave summed
        g = G + (int64_t)extra * kGStride;
#pragma unroll
        for (int m = 0; m < 6; ++m) r[m] += g[kGRows + m * 16 + c];
    }
}
// ... and their expansion into V and b by broadcasts inside the 16-lane group
template <int L>
__device__ __forceinline__ void expand_head_rows(const double (&r)[6], int c, double (&V)[21], double (&b)[6]) {
    // g_v rides in V's upper triangle (gvSlot): V(0, 1..5) and V(1, 2)
    const double gg[6] = {row_bcast<L + 1>(r[0]), row_bcast<L + 2>(r[0]), row_bcast<L + 3>(r[0]),
                          row_bcast<L + 4>(r[0]), row_bcast<L + 5>(r[0]), row_bcast<L + 2>(r[1])};
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        V[tri(m, 0)] = row_bcast<L>(r[m]);
        if (m >= 1) V[tri(m, 1)] = row_bcast<L + 1>(r[m]);
        if (m >= 2) V[tri(m, 2)] = row_bcast<L + 2>(r[m]);
        if (m >= 3) V[tri(m, 3)] = row_bcast<L + 3>(r[m]);
        if (m >= 4) V[tri(m, 4)] = row_bcast<L + 4>(r[m]);
        if (m >= 5) V[tri(m, 5)] = row_bcast<L + 5>(r[m]);
        b[m] = c < L ? r[m] : gg[m];
    }
}

template <int L, bool WIDE>
__device__ __forceinline__ void load_view_head(const double* __restrict__ G, int item0, int nitems, int extra, int c,
                                               double (&V)[21], double (&b)[6]) {
    const double* g = G + (int64_t)item0 * kGStride;
    if constexpr (WIDE) {
        double r[6];
        request_head_rows(G, item0, nitems, extra, c, r);
        expand_head_rows<L>(r, c, V, b);
    } else {
        // this lane's right-hand side: row c of E at m * 16 + c, or g_v at gvSlot(L, m) = L + 1 + m (m < 5), 16 + L + 2
        const int boff = c < L ? kGRows + c : gvSlot(L, 0);
        const int bstep = c < L ? 16 : 1;
        const int b5 = c < L ? kGRows + 5 * 16 + c : gvSlot(L, 5);
#pragma unroll
        for (int m = 0; m < 6; ++m) {
#pragma unroll
            for (int n = 0; n <= m; ++n) V[tri(m, n)] = g[kGRows + m * 16 + L + n];
            b[m] = g[m < 5 ? boff + m * bstep : b5];
        }
        for (int it = 1; it < nitems; ++it) {                 // > 1 item only for views above kGramChunk points
            g += kGStride;
#pragma unroll
            for (int m = 0; m < 6; ++m) {
#pragma unroll
                for (int n = 0; n <= m; ++n) V[tri(m, n)] += g[kGRows + m * 16 + L + n];
                b[m] += g[m < 5 ? boff + m * bstep : b5];
            }
        }
        if (extra >= 0) {                                     // stream form: the part of the view a second wave summed
            g = G + (int64_t)extra * kGStride;
#pragma unroll
            for (int m = 0; m < 6; ++m) {
#pragma unroll
                for (int n = 0; n <= m; ++n) V[tri(m, n)] += g[kGRows + m * 16 + L + n];
                b[m] += g[m < 5 ? boff + m * bstep : b5];
            }
        }
    }
}

// Cholesky of V + lam diag(V) (JTJ + lam * diag(JTJ), src/calibrate.py:147,152) in place -- on return
// the strictly lower part of V is the factor Lc, invd the reciprocal of its diagonal -- and the
// forward substitution z = Lc^-1 b of this lane's right-hand side. Returns false when a pivot is
// not positive.
__device__ __forceinline__ bool cholesky6(double (&V)[21], double lam, double (&invd)[6]) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = V[tri(j, j)] + lam * V[tri(j, j)];
#pragma unroll
        for (int q = 0; q < j; ++q) d -= V[tri(j, q)] * V[tri(j, q)];
        if (!(d > 0.0)) ok = false;
        const double inv = rsqrt_nr(d);
        invd[j] = inv;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double t = V[tri(i, j)];
#pragma unroll
            for (int q = 0; q < j; ++q) t -= V[tri(i, q)] * V[tri(j, q)];
            V[tri(i, j)] = t * inv;
        }
    }
    return ok;
}
__device__ __forceinline__ void forward6(const double (&V)[21], const double (&invd)[6], const double (&b)[6],
                                         double (&z)[6]) {
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        double t = b[m];
#pragma unroll
        for (int n = 0; n < m; ++n) t -= V[tri(m, n)] * z[n];
        z[m] = t * invd[m];
    }
}
__device__ __forceinline__ bool eliminate(double (&V)[21], const double (&b)[6], double lam,
                                          double (&invd)[6], double (&z)[6]) {
    const bool ok = cholesky6(V, lam, invd);
    forward6(V, invd, b, z);
    return ok;
}

// Lc^T d = z, then the view's part of the next candidate P and the candidate's view constants (one lane per view)
template <int L, typename T>
__device__ __forceinline__ void finish_view_lane(const double (&V)[21], const double (&invd)[6], const double (&z)[6],
                                                 int v, const int* __restrict__ view_ext, const double* __restrict__ Pc,
                                                 double* __restrict__ Pn, T* __restrict__ VC) {
    double d[6];
#pragma unroll
    for (int m = 5; m >= 0; --m) {
        double t = z[m];
#pragma unroll
        for (int n = m + 1; n < 6; ++n) t -= V[tri(n, m)] * d[n];
        d[m] = t * invd[m];
    }
    const int64_t o = L + 6 * (int64_t)view_ext[v];
    double en[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) { en[m] = Pc[o + m] + d[m]; Pn[o + m] = en[m]; }
    // view constants of the candidate (layout of view_setup_kernel)
    const double deg = 0.017453292519943295;
    double sn[3], cs[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double th = en[a] * deg;
        sincos(th, &sn[a], &cs[a]);
        if (fabs(th) <= 1e-8) { sn[a] = 0.0; cs[a] = 1.0; }
    }
    const double sx = sn[0], cx = cs[0], sy = sn[1], cy = cs[1], sz = sn[2], cz = cs[2];
    const double o18[18] = {cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx,
                            sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx,
                            -sy, cy * sx, cy * cx, en[3], en[4], en[5],
                            deg * cz * cy, deg * sz * cy, -deg * sy, -deg * sz, deg * cz, 0.0};
    T* dst = VC + (int64_t)v * kViewStride;
    using T2 = typename Pair<T>::type;
#pragma unroll
    for (int j = 0; j < kViewStride / 2; ++j) {                 // 18 values: 144 / 72 bytes, 16- / 8-byte aligned pairs
        T2 t;
        t.x = (T)o18[2 * j];
        t.y = (T)o18[2 * j + 1];
        reinterpret_cast<T2*>(dst)[j] = t;
    }
}

// ---------------------------------------------------------------- schur partials
// grid (nblocks, 3): y = 0 variant A (candidate blocks, lambda_accept),
//                    y = 1 variant B (current blocks, lambda_reject),
//                    y = 2 plain sums of the workgroup partials of B, g_c, sum r^2 (variant A's fields).
// Elimination (y < 2): 16 lanes per view, 4 views per wave and trip, reading only the head of the
// view's record. Every lane factors the view's damped 6x6 block Vh = Lc Lc^T (loaded by broadcast),
// lane c < L forward-substitutes its row of E (z_c = Lc^-1 E_c), lane L the view gradient
// (z_g = Lc^-1 g_v). Then sum_views E Vh^-1 [E^T | g_v] = W^T W with W = [z_0 .. z_{L-1} z_g] is one
// more tall-skinny Gram: 6 x v_mfma_f64_16x16x4_f64 per trip, K-slot = view, no cross-lane traffic.
constexpr int kSchurBlock = 256;
constexpr int kSchurViewsPerBlock = kSchurBlock / 16;

template <int L, bool WIDE, bool STREAM>
// (narrow head loads on stream records -- 27 + 27 per-value loads -- need 136 registers: three workgroups per CU there,
// instead of four with 20 bytes of scratch per lane; that combination is only ever run on request, CALIB_HEAD_LOADS=narrow)
__global__ __launch_bounds__(kSchurBlock, (!WIDE && STREAM) ? 3 : 4) void schur_kernel(const double* __restrict__ G0,
                                                            const double* __restrict__ G1,
                                                            const LMState* __restrict__ st,
                                                            const int* __restrict__ view_item0,
                                                            int nv, StreamMap sm, const double* __restrict__ bpart,
                                                            int n_bpart, double* __restrict__ part) {
    constexpr int VA = variantSize(L);
    constexpr int kNfail = 2 * L * L + 2 * L, kSse = kNfail + 1;
    __shared__ double sfail[kSchurViewsPerBlock];
    __shared__ double stile[kSchurBlock / 64][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = tid & 15, grp = tid >> 4, k = lane >> 4;

    if (blockIdx.y == 2) {
        if (st->done) return;
        // thread t of each half-block owns one field of the workgroup partials of B, g_c, sum r^2
        constexpr int NF = L * L + L + 1;
        double* out = part + (int64_t)blockIdx.x * VA;
        const int t = tid & 127, half = tid >> 7;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (t < NF) {
            const int64_t step = (int64_t)gridDim.x * 2;
            int64_t it = (int64_t)blockIdx.x * 2 + half;
            const double* src = bpart + t;
            for (; it + 3 * step < n_bpart; it += 4 * step) {
                s0 += src[it * kPartStride];
                s1 += src[(it + step) * kPartStride];
                s2 += src[(it + 2 * step) * kPartStride];
                s3 += src[(it + 3 * step) * kPartStride];
            }
            for (; it < n_bpart; it += step) s0 += src[it * kPartStride];
        }
        double* sh = &stile[0][0];
        sh[tid] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (tid < NF) {
            const double tot = sh[tid] + sh[128 + tid];
            out[tid < L * L ? tid : (tid < L * L + L ? 2 * L * L + (tid - L * L) : kSse)] = tot;
        }
        return;
    }

    // blockIdx.y = 0 / 1 is the BUFFER the block eliminates (G0 / G1), known at launch: the first views' heads
    // are requested before the LM state -- which says whether that buffer holds the candidate (variant A) or the
    // current blocks (variant B) -- has arrived, one memory latency instead of two dependent ones
    const double* G = blockIdx.y ? G1 : G0;
    double V[21], b[6];
    int v0 = blockIdx.x * kSchurViewsPerBlock;
    if (v0 + grp < nv) {
        // view_item0 == nullptr: every view is a single item (item index == view index)
        const int v = v0 + grp, i0 = view_item0 ? view_item0[v] : v;
        load_view_head<L, WIDE>(G, i0, view_item0 ? view_item0[v + 1] - i0 : 1, (STREAM ? stream_extra_item(sm, v) : -1), c, V, b);
    }
    if (st->done) return;
    const bool boot = st->round == 0;
    const int cand = st->cur ^ 1;
    const int variant = (int)blockIdx.y == cand ? 0 : 1;
    double* out = part + ((int64_t)variant * gridDim.x + blockIdx.x) * VA;
    if (variant == 1 && boot) {       // no "current" blocks yet
        for (int i = tid; i < VA; i += kSchurBlock) out[i] = 0.0;
        return;
    }
    const double lam = variant == 0 ? (boot ? st->lam : st->lam / 10) : st->lam * 10;

    d4 acc = {0.0, 0.0, 0.0, 0.0};
    double nfail = 0.0;
    const int stride = gridDim.x * kSchurViewsPerBlock;
    while (v0 < nv) {
        double z[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) z[m] = 0.0;
        const bool live = v0 + grp < nv;                      // whole 16-lane group together
        const bool more = v0 + stride + grp < nv;
        double rn[6];
        if (WIDE && more) {                                   // the next trip's seven rows: in flight during this elimination
            const int v = v0 + stride + grp, i0 = view_item0 ? view_item0[v] : v;
            request_head_rows(G, i0, view_item0 ? view_item0[v + 1] - i0 : 1, (STREAM ? stream_extra_item(sm, v) : -1), c, rn);
        }
        if (live) {
            double invd[6];
            const bool ok = eliminate(V, b, lam, invd, z);
            if (c > L) {                                      // lanes above the gradient column contribute nothing
#pragma unroll
                for (int m = 0; m < 6; ++m) z[m] = 0.0;
            }
            if (!ok) nfail += 1.0;
        }
        v0 += stride;
        if (more) {
            if constexpr (WIDE) {
                expand_head_rows<L>(rn, c, V, b);
            } else {                                          // the next trip's heads, behind this trip's MFMAs
                const int v = v0 + grp, i0 = view_item0 ? view_item0[v] : v;
                load_view_head<L, false>(G, i0, view_item0 ? view_item0[v + 1] - i0 : 1, (STREAM ? stream_extra_item(sm, v) : -1), c, V, b);
            }
        }
        // W^T W: K-slot k = this lane's view, six rows per view
#pragma unroll
        for (int m = 0; m < 6; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(z[m], z[m], acc, 0, 0, 0);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) stile[wave][(k + 4 * reg) * 16 + c] = acc[reg];
    if (c == 0) sfail[grp] = nfail;
    __syncthreads();
    // fixed-order sums over the block's waves / view groups; B, g_c and sum r^2 of variant A come
    // from the y = 2 blocks, variant B re-uses the state's copies (zeros here)
    for (int i = tid; i < L * L; i += kSchurBlock) {
        const int row = i / L, col = i - row * L;
        double t = 0.0;
        for (int w = 0; w < kSchurBlock / 64; ++w) t += stile[w][row * 16 + col];
        out[L * L + i] = t;                                   // sum E Vh^-1 E^T
        if (variant == 1) out[i] = 0.0;
    }
    if (tid < L) {
        double t = 0.0;
        for (int w = 0; w < kSchurBlock / 64; ++w) t += stile[w][tid * 16 + L];
        out[2 * L * L + L + tid] = t;                         // sum E Vh^-1 g_v
        if (variant == 1) out[2 * L * L + tid] = 0.0;
    }
    if (tid == 0) {
        double nf = 0.0;
        for (int g = 0; g < kSchurViewsPerBlock; ++g) nf += sfail[g];
        out[kNfail] = nf;
        if (variant == 1) out[kSse] = 0.0;
    }
}

// ---------------------------------------------------------------- peer exchange over xGMI
// The cross-GPU sum of the reduce buffer, done by reduce_kernel itself: no collective launch, no fence.
// Every rank owns SLOT MEMORY (uncached / fine-grained device memory, mapped into every peer through HIP IPC)
// of world x 2 x kPeerStride 16-byte cells; cell (src, parity, i) of rank d's memory holds element i as
// contributed by rank src in a round of that parity. A cell is two 8-byte words {data half, epoch}: an 8-byte
// word is never torn, so a reader that sees the round's epoch in a word also sees that word's data (the
// "LL" idea of the collective libraries) -- the writer needs no release fence and no separate flag. A wave
// that has summed element i stores it into cell (rank, parity, i) of every OTHER rank (lane d writes to rank d,
// point-to-point over xGMI), then lane s polls cell (s, parity, i) of its own memory until rank s's epoch
// shows up (its own contribution it takes from the register); the values are added in rank order, so every rank gets bitwise the
// same sums. Two parities are enough: a rank can only start writing round r + 2 after it has received
// everybody's round r + 1, which they sent after reading round r.
// Every spin is bounded by the wall clock (100 MHz): on a timeout the fault word is raised, the element
// becomes NaN (the step is rejected) and the host reports the fault; the grid always drains.
constexpr int kPeerStride = 448;       // >= reduceSize(10) = 444 cells per (source, parity)
constexpr int kPeerMaxRanks = 64;      // one lane per rank

constexpr int kPeerInline = 8;         // ranks whose slot pointers travel in the kernel arguments

struct PeerExchange {
    unsigned long long* const* slots;  // [world] rank d's slot memory as mapped on this device (device array)
    unsigned long long* slot8[kPeerInline];  // the first 8 of them by value: no dependent load before the stores
    int* fault;                        // device word: a spin of this handle has timed out
    int* notify;                       // host-visible word (or null): 2 = stop enqueueing rounds, the exchange is broken
    unsigned long long timeout_ticks;
    unsigned epoch;                    // > 0, +1 per exchange; identical on every rank
    int world, rank;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// one 16-byte cell, system scope (sc0 sc1: past this device's caches). The hardware may split the access
// into its two 8-byte words -- each carries its own epoch, so a torn cell is simply not ready yet.
__device__ __forceinline__ void cell_store(unsigned long long* cell, u32x4 v) {
    asm volatile("flat_store_dwordx4 %0, %1 sc0 sc1" : : "v"(cell), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4 cell_load(const unsigned long long* cell) {
    u32x4 v;
    asm volatile("flat_load_dwordx4 %0, %1 sc0 sc1\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" : "=v"(v) : "v"(cell) : "memory");
    return v;
}

__device__ __forceinline__ double peer_sum(const PeerExchange& x, int i, double t, int lane) {
    const unsigned par = x.epoch & 1u;
    // Once a spin of this handle has given up (a peer died, the ranks left lockstep) the exchange is not retried: every
    // later kernel returns NaN at once instead of spinning to its own deadline (64 self-test rounds, or the rounds a
    // host has enqueued ahead, would otherwise wait one full timeout EACH).
    if (__hip_atomic_load(x.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return __builtin_nan("");
    double v = 0.0;
    if (lane < x.world) {
        unsigned long long* base = nullptr, * own = nullptr;
        if (x.world <= kPeerInline) {
#pragma unroll
            for (int r = 0; r < kPeerInline; ++r) {
                base = lane == r ? x.slot8[r] : base;
                own = x.rank == r ? x.slot8[r] : own;
            }
        } else {
            base = x.slots[lane];
            own = x.slots[x.rank];
        }
        unsigned long long* dst = base + 2 * ((int64_t)(x.rank * 2 + par) * kPeerStride + i);
        u32x4 mine;
        mine.x = (unsigned)__double2loint(t); mine.y = x.epoch;
        mine.z = (unsigned)__double2hiint(t); mine.w = x.epoch;
        if (lane != x.rank) cell_store(dst, mine);               // this rank's own contribution stays in its register
        const unsigned long long* src = own + 2 * ((int64_t)(lane * 2 + par) * kPeerStride + i);
        const long long t0 = wall_clock64();
        v = t;
        while (lane != x.rank) {
            const u32x4 c = cell_load(src);
            if (c.y == x.epoch && c.w == x.epoch) {
                v = __hiloint2double((int)c.z, (int)c.x);
                break;
            }
            if ((unsigned long long)(wall_clock64() - t0) > x.timeout_ticks) {
                __hip_atomic_store(x.fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (x.notify) __hip_atomic_store(x.notify, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                v = __builtin_nan("");
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    double acc = 0.0;
    for (int r = 0; r < x.world; ++r) acc += __shfl(v, r, 64);
    return acc;
}

// ---------------------------------------------------------------- reduce
// One wave per reduce-buffer element: lanes stride over the schur block partials, then a
// fixed-order shuffle tree. grid = 2 * VA. With a peer exchange (x.world > 1) the element is then summed
// over the ranks as well.
__global__ __launch_bounds__(64) void reduce_kernel(const double* __restrict__ part, int nblocks, int VA,
                                                    const LMState* __restrict__ st,
                                                    double* __restrict__ red, PeerExchange x) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const int variant = i / VA, idx = i - variant * VA;
    const double* src = part + (int64_t)variant * nblocks * VA + idx;
    // all of a lane's loads are issued before the first add (a rolled load-wait-add loop made this
    // kernel a chain of up to 16 memory latencies); nblocks <= kMaxSchurBlocks = 16 x 64
    double v[kMaxSchurBlocks / 64];
#pragma unroll
    for (int j = 0; j < kMaxSchurBlocks / 64; ++j) {
        const int b = lane + 64 * j;
        v[j] = b < nblocks ? src[(int64_t)b * VA] : 0.0;
    }
#pragma unroll
    for (int w = 1; w < kMaxSchurBlocks / 64; w *= 2)
#pragma unroll
        for (int j = 0; j + w < kMaxSchurBlocks / 64; j += 2 * w) v[j] += v[j + w];
    double t = v[0];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    // the finished flag is looked at only now: its load travels with the partials' instead of ahead of them.
    // A round that comes after the end of the loop still takes part in the exchange (with whatever it summed): the
    // epochs advance per ENQUEUED round on every rank, and two parities are only enough while every epoch is used
    const bool done = st->done != 0;
    if (x.world > 1) t = peer_sum(x, i, __shfl(t, 0, 64), lane);
    if (done) return;
    if (lane == 0) red[i] = t;
}

// calib_peer_selftest: rank r contributes (r + 1) * (i + 1) + round to element i; every rank must read
// back the exact integer sum.
__global__ __launch_bounds__(64) void peer_selftest_kernel(PeerExchange x, int round, int* __restrict__ mismatches) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const double mine = (double)(x.rank + 1) * (i + 1) + round;
    const double got = peer_sum(x, i, mine, lane);
    const double want = 0.5 * x.world * (x.world + 1.0) * (i + 1) + (double)round * x.world;
    if (lane == 0 && !(got == want)) atomicAdd(mismatches, 1);
}

// sum of the jacobian kernel's per-tile partials (calib_eval only), one workgroup
__global__ __launch_bounds__(256) void sse_reduce_kernel(const double* __restrict__ sse_part, int64_t n,
                                                         double* __restrict__ out) {
    __shared__ double ssum[256];
    const int tid = threadIdx.x;
    double e = 0.0;
    for (int64_t i = tid; i < n; i += 256) e += sse_part[i];
    ssum[tid] = e;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) ssum[tid] += ssum[tid + s];
        __syncthreads();
    }
    if (tid == 0) out[0] = ssum[0];
}

// Cross-lane traffic inside a 16-lane view group rides on DPP (a 16-lane group is exactly one DPP
// row): VALU latency, no trip through the LDS crossbar as with ds_bpermute. The four exchange
// patterns below (lane ^ 1, lane ^ 2, mirror inside each half row, mirror of the row) pair every lane
// with the same partner sets as an xor butterfly, so sums come out bit-identical on all 16 lanes.
constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // row_half_mirror: lane i <- lane 7 - i of its half row
constexpr int kDppMirror = 0x140;      // row_mirror: lane i <- lane 15 - i of its row

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    return __hiloint2double(dpp_i<CTRL>(__double2hiint(v)), dpp_i<CTRL>(__double2loint(v)));
}
// lane a (0..3) of every quad to the four lanes of that quad
template <int A>
__device__ __forceinline__ double quad_bcast(double v) { return dpp_d<A * 0x55>(v); }

// sum over the 16 lanes of a view group
__device__ __forceinline__ double group_sum16(double v) {
    v += dpp_d<kDppXor1>(v);
    v += dpp_d<kDppXor2>(v);
    v += dpp_d<kDppHalfMirror>(v);
    v += dpp_d<kDppMirror>(v);
    return v;
}

// a / b with a v_rcp_f64 seed, two Newton steps and one residual correction of the quotient
__device__ __forceinline__ double div_nr(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}

// ---------------------------------------------------------------- N x N solve on 16 lanes
// Gauss-Jordan with partial pivoting; lane i < N owns row i of [S | s] (row[N] = rhs), lanes >= N
// hold zero rows. On return the lane that served as pivot row for column myCol holds x[myCol] in d
// (myCol = -1 on the other lanes). Returns true when a pivot is exactly zero / not finite.
// WAVE0: the caller runs in lanes 0..15 of its wave only (the other lanes inactive), so the pivot
// row is a wave-uniform lane and its broadcast a v_readlane; otherwise one ds_bpermute per dword.
template <int N, bool WAVE0 = false>
__device__ __forceinline__ bool gauss_jordan16(double (&row)[N + 1], int i, int& myCol, double& d) {
    bool used = false, singular = false;
    myCol = -1;
#pragma unroll
    for (int col = 0; col < N; ++col) {
        double best = (!used && i < N) ? fabs(row[col]) : -1.0;
        int bi = i;
        auto take = [&](double ob, int oi) {
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        };
        take(dpp_d<kDppXor1>(best), dpp_i<kDppXor1>(bi));
        take(dpp_d<kDppXor2>(best), dpp_i<kDppXor2>(bi));
        take(dpp_d<kDppHalfMirror>(best), dpp_i<kDppHalfMirror>(bi));
        take(dpp_d<kDppMirror>(best), dpp_i<kDppMirror>(bi));
        if (!(best > 0.0)) singular = true;
        double prow[N + 1];
        if (WAVE0) {
            const int sb = __builtin_amdgcn_readfirstlane(bi);
#pragma unroll
            for (int j = col; j <= N; ++j)
                prow[j] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(row[j]), sb),
                                           __builtin_amdgcn_readlane(__double2loint(row[j]), sb));
        } else {
#pragma unroll
            for (int j = col; j <= N; ++j) prow[j] = __shfl(row[j], bi, 16);
        }
        if (i == bi) {
            used = true;
            myCol = col;
        } else {
            const double f = div_nr(row[col], prow[col]);
#pragma unroll
            for (int j = col + 1; j <= N; ++j) row[j] -= f * prow[j];
            row[col] = 0.0;
        }
    }
    double piv = 1.0;
#pragma unroll
    for (int j = 0; j < N; ++j) if (myCol == j) piv = row[j];
    d = row[N] / piv;
    return singular;
}

// ---------------------------------------------------------------- SPD solve on lanes 0..15 of a wave
// S x = s for a symmetric positive definite S (the damped Schur complement of an SPD system is
// SPD); lane i owns row i of [S | s]. Gaussian elimination without pivoting -- backward stable for
// SPD matrices -- so the pivot row is a compile-time lane and its broadcast a v_readlane: no pivot
// search, no division per row (the 10 x 10 Gauss-Jordan with pivot search took 5 of the update
// kernel's 9-13 us). Every lane returns the whole solution. Returns false when a pivot is not
// positive; the caller then falls back to gauss_jordan16 (same behaviour as before on such input).
// Only valid in lanes 0..15 of a wave with the other lanes inactive.
__device__ __forceinline__ double readlane_d(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}

template <int N>
__device__ __forceinline__ bool spd_solve_wave0(double (&row)[N + 1], int i, double (&x)[N]) {
    bool ok = true;
    double rinv[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double p[N + 1];
#pragma unroll
        for (int k = j; k <= N; ++k) p[k] = readlane_d(row[k], j);
        if (!(p[j] > 0.0)) ok = false;
        rinv[j] = div_nr(1.0, p[j]);
        if (i > j) {
            const double f = row[j] * rinv[j];
#pragma unroll
            for (int k = j + 1; k <= N; ++k) row[k] -= f * p[k];
        }
    }
    double acc = row[N];                                 // back substitution, U x = y
#pragma unroll
    for (int j = N - 1; j >= 0; --j) {
        const double xj = readlane_d(acc, j) * rinv[j];
        x[j] = xj;
        if (i < j) acc -= row[j] * xj;
    }
    return ok;
}

// ---------------------------------------------------------------- update (16 lanes, device function)
// The control flow of src/calibrate.py:155-168 and the L x L solve S dc = s of the shared block.
// The LM state is double-buffered per round (in: this round's, out: next round's), so EVERY
// 16-lane group of the update kernel below can run this redundantly from the same inputs and
// reach the same decision with no grid-wide hand-off; only the `writer` group stores the new
// state, the trace row and the shared part of the next candidate. Lane i owns row i of [S | s]:
// spd_solve_wave0, with Gauss-Jordan and partial pivoting across lanes as the fallback
// (np.linalg.inv in the reference is LU with partial pivoting). Returns false when the loop is over (nothing left to back-substitute).
template <int L>
__device__ __forceinline__ bool lm_update_step(const LMState* __restrict__ in, LMState* __restrict__ out,
                                               const double* __restrict__ red, double* __restrict__ P0,
                                               double* __restrict__ P1, double* __restrict__ trace,
                                               bool writer, int i, int& cur_out, double& lam_out,
                                               double (&dc)[L]) {
    constexpr int VA = variantSize(L);
    constexpr int kNfail = 2 * L * L + 2 * L, kSse = kNfail + 1;
    const bool w0 = writer && i == 0;
    if (in->done) {                                  // keep the finished state visible to later rounds
        if (writer) for (int j = i; j < (int)(sizeof(LMState) / 8); j += 16)
            reinterpret_cast<double*>(out)[j] = reinterpret_cast<const double*>(in)[j];
        return false;
    }
    double* Pb[2] = {P0, P1};
    int cur = in->cur;
    const int cand = cur ^ 1;
    const int round = in->round;
    const double err_cand = red[kSse];
    double lam = in->lam;
    // Both candidate systems (variant A in red[0, VA), variant B behind it) and both copies of B, g_c
    // (variant A's fresh sums, the state's) are requested before the decision is known: one trip to
    // L2 instead of two dependent ones; the unused half is dropped below.
    double sA[L], sB[L], bA[L], bS[L];
    double rhsA = 0.0, rhsB = 0.0, gA = 0.0, gS = 0.0;
#pragma unroll
    for (int j = 0; j < L; ++j) { sA[j] = 0.0; sB[j] = 0.0; bA[j] = 0.0; bS[j] = 0.0; }
    if (i < L) {
#pragma unroll
        for (int j = 0; j < L; ++j) {
            sA[j] = red[L * L + i * L + j];
            sB[j] = red[VA + L * L + i * L + j];
            bA[j] = red[i * L + j];
            bS[j] = in->B[i * L + j];
        }
        rhsA = red[2 * L * L + L + i];
        rhsB = red[VA + 2 * L * L + L + i];
        gA = red[2 * L * L + i];
        gS = in->gc[i];
    }
    const double nfailA = red[kNfail], nfailB = red[VA + kNfail];
    bool useB = false;                               // the reject variant (current blocks, 10 lambda)
    bool done = false;
    double err_cur_new = err_cand, last_err = err_cand;
    int iters = in->iters, accepted = 0;
    if (round == 0) {
        cur = cand;
    } else {
        const int it = round - 1;
        const double err_cur = in->err_cur;
        const bool acc = err_cand < err_cur;         // strict; NaN rejects (src/calibrate.py:161)
        if (trace && writer) {
            double* row = trace + (int64_t)it * (5 + L);
            if (i < L) row[5 + i] = Pb[cur][i];
            if (i == 0) { row[0] = it; row[1] = err_cur; row[2] = err_cand; row[3] = lam; row[4] = acc ? 1.0 : 0.0; }
        }
        if (acc) { cur = cand; lam = lam / 10; } else { lam = lam * 10; useB = true; }
        done = !(in->lam_min < lam && lam < in->lam_max) || err_cur < in->err_min || it + 1 >= in->max_iters;
        last_err = err_cur;                          // the reference returns the pre-update error (:155,171)
        err_cur_new = acc ? err_cand : err_cur;
        iters = it + 1;
        accepted = acc ? 1 : 0;
    }
    int error = 0;
    if (!done && (useB ? nfailB : nfailA) > 0.0) { error = -3; done = true; }

    // B and g_c of the parameters the step starts from: variant A carries them for a freshly
    // accepted (or the bootstrap) point, which also becomes the state's copy; after a rejection
    // the state's copy of the unchanged current point is used.
    double row[L + 1], brow[L];
    const double gci = useB ? gS : gA;
    // row i of S = B + lam diag(B) - sum E Vh^-1 E^T, rhs s = g_c - sum E Vh^-1 g_v (zero rows for i >= L)
#pragma unroll
    for (int j = 0; j < L; ++j) {
        brow[j] = useB ? bS[j] : bA[j];
        row[j] = brow[j] - (useB ? sB[j] : sA[j]);
    }
    row[L] = gci - (useB ? rhsB : rhsA);
#pragma unroll
    for (int j = 0; j < L; ++j) if (j == i) row[j] += lam * brow[j];
    if (!done) {
        double saved[L + 1];
#pragma unroll
        for (int j = 0; j <= L; ++j) saved[j] = row[j];
        if (!spd_solve_wave0<L>(row, i, dc)) {           // not positive definite to working precision
            int myCol = -1;
            double d = 0.0;
            if (gauss_jordan16<L, true>(saved, i, myCol, d)) { error = -3; done = true; }
#pragma unroll
            for (int c = 0; c < L; ++c) dc[c] = group_sum16(myCol == c ? d : 0.0);
        }
    } else {
#pragma unroll
        for (int c = 0; c < L; ++c) dc[c] = 0.0;
    }

    if (writer) {
        if (i < L) {
#pragma unroll
            for (int j = 0; j < L; ++j) out->B[i * L + j] = brow[j];      // (fresh ? new : unchanged) copy
            out->gc[i] = gci;
            double dci = 0.0;
#pragma unroll
            for (int c = 0; c < L; ++c) if (c == i) dci = dc[c];
            out->dc[i] = dci;
            if (!done) Pb[cur ^ 1][i] = Pb[cur][i] + dci;
        }
        if (w0) {
            out->lam = lam; out->err_cur = err_cur_new; out->last_err = last_err;
            out->lam_min = in->lam_min; out->lam_max = in->lam_max; out->err_min = in->err_min;
            out->cur = cur; out->round = round + 1; out->iters = iters; out->max_iters = in->max_iters;
            out->done = done ? 1 : 0; out->error = error; out->accepted_last = accepted; out->pad = 0;
            out->notify = in->notify;
            if (done && in->notify) __hip_atomic_store(in->notify, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    cur_out = cur;
    lam_out = lam;
    return !done;
}

// ---------------------------------------------------------------- update + back-substitution + next view constants
// One launch per LM round after the reduce (and, across GPUs, the all-reduce): every workgroup
// (a) takes the accept/reject decision and solves for dc (lm_update_step, redundantly, one 16-lane
// group per workgroup), then every 16-lane group
// (b) back-substitutes its view, delta_i = Vh^-1 (g_i - E_i^T dc), and writes the view's part of
// the next candidate P[cur^1] = P[cur] + delta, (c) turns the candidate's Euler angles (degrees)
// into the rotation / derivative-axis constants the next round's point kernels read (what
// view_setup_kernel does for round 0).
// The rest of a view's update once its block is factored (V: Cholesky factor, invd, z from eliminate):
// back-substitution, the view's part of the next candidate, and the candidate's view constants.
template <int L, typename T>
__device__ __forceinline__ void finish_view(const double (&V)[21], const double (&invd)[6], const double (&z)[6],
                                            double coef, int c, int v, const int* __restrict__ view_ext,
                                            const double* __restrict__ Pc, double* __restrict__ Pn,
                                            T* __restrict__ VC) {
    // w = Lc^-1 (g_v - E^T dc): lane c < L carries -dc[c] z_c, lane L the gradient's z_g
    double w[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) w[m] = group_sum16(coef * z[m]);
    // Lc^T d = w
    double d[6];
#pragma unroll
    for (int m = 5; m >= 0; --m) {
        double t = w[m];
#pragma unroll
        for (int n = m + 1; n < 6; ++n) t -= V[tri(n, m)] * d[n];
        d[m] = t * invd[m];
    }
    const int64_t o = L + 6 * (int64_t)view_ext[v];
    double en[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) en[m] = Pc[o + m] + d[m];
    if (c < 6) {
        double ev = en[0];
#pragma unroll
        for (int m = 1; m < 6; ++m) ev = (c == m) ? en[m] : ev;
        Pn[o + c] = ev;
    }
    // view constants of the candidate (layout of view_setup_kernel)
    const double deg = 0.017453292519943295;
    // one sincos per wave instruction stream: lane a (< 3) of each quad takes angle a, the results
    // are handed round inside the quad (DPP quad_perm)
    double sn[3], cs[3];
    {
        const int a = c & 3;                             // every quad evaluates the three angles itself
        const double th = (a == 1 ? en[1] : (a == 2 ? en[2] : en[0])) * deg;
        double s1, c1;
        sincos(th, &s1, &c1);
        if (fabs(th) <= 1e-8) { s1 = 0.0; c1 = 1.0; }
        sn[0] = quad_bcast<0>(s1); sn[1] = quad_bcast<1>(s1); sn[2] = quad_bcast<2>(s1);
        cs[0] = quad_bcast<0>(c1); cs[1] = quad_bcast<1>(c1); cs[2] = quad_bcast<2>(c1);
    }
    const double sx = sn[0], cx = cs[0], sy = sn[1], cy = cs[1], sz = sn[2], cz = cs[2];
    const double o18[18] = {cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx,
                            sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx,
                            -sy, cy * sx, cy * cx, en[3], en[4], en[5],
                            deg * cz * cy, deg * sz * cy, -deg * sy, -deg * sz, deg * cz, 0.0};
    double mine = o18[0];
#pragma unroll
    for (int j = 1; j < 16; ++j) mine = (c == j) ? o18[j] : mine;
    T* dst = VC + (int64_t)v * kViewStride;
    dst[c] = (T)mine;
    if (c < 2) dst[16 + c] = (T)(c == 0 ? o18[16] : o18[17]);
}

template <int L, typename T, bool WIDE, bool STREAM>
// (the narrow-load form holds 27 loaded values beside the solve's rows: 130 registers. At four workgroups per CU it
// spilled 12-20 bytes per lane; three per CU fit without scratch but turn c4's 782 workgroups into two rounds (12.3 us
// instead of 10.3). Shards beyond the small kernel's reach therefore load wide by default (calib_lm.hip: wide_heads),
// which needs 122 registers; this form runs on request only, CALIB_HEAD_LOADS=narrow)
__global__ __launch_bounds__(kSchurThreads, WIDE ? 4 : 3) void update_backsub_kernel(
        const double* __restrict__ G0, const double* __restrict__ G1, const LMState* __restrict__ st_in,
        LMState* __restrict__ st_out, const double* __restrict__ red, const int* __restrict__ view_item0,
        const int* __restrict__ view_ext, int nv, StreamMap sm, double* __restrict__ P0, double* __restrict__ P1,
        double* __restrict__ trace, T* __restrict__ VC) {
    const int tid = threadIdx.x, c = tid & 15;
    const bool writer = blockIdx.x == 0 && tid < 16;
    // the first 16-lane group of every workgroup takes the decision and solves; the others get
    // (go, cur, lambda, dc) through LDS
    __shared__ double sdec[L + 3];
    int cur = 0;
    double lam = 0.0, dc[L];
    if (tid < 16) {
        const bool go = lm_update_step<L>(st_in, st_out, red, P0, P1, trace, writer, c, cur, lam, dc);
        if (c == 0) { sdec[0] = go ? 1.0 : 0.0; sdec[1] = (double)cur; sdec[2] = lam; }
        if (c < L) {
            double dci = 0.0;
#pragma unroll
            for (int j = 0; j < L; ++j) if (j == c) dci = dc[j];
            sdec[3 + c] = dci;
        }
    }
    __syncthreads();
    if (sdec[0] == 0.0) return;
    cur = (int)sdec[1];
    lam = sdec[2];
    const double coef = c < L ? -sdec[3 + c] : (c == L ? 1.0 : 0.0);
    const double* G = cur ? G1 : G0;
    const double* Pc = cur ? P1 : P0;
    double* Pn = cur ? P0 : P1;
    // grid-stride over views: the decision / solve above is paid once per workgroup, not per view
    const int stride = gridDim.x * (kSchurThreads / 16);
    int v = blockIdx.x * (kSchurThreads / 16) + (tid >> 4);
    double rn[6];
    if (WIDE && v < nv) {
        const int i0 = view_item0 ? view_item0[v] : v;
        request_head_rows(G, i0, view_item0 ? view_item0[v + 1] - i0 : 1, (STREAM ? stream_extra_item(sm, v) : -1), c, rn);
    }
    for (; v < nv; v += stride) {
        double V[21], b[6], invd[6], z[6];
        if constexpr (WIDE) {
            expand_head_rows<L>(rn, c, V, b);
            if (v + stride < nv) {                            // the next view's rows: in flight during this view's update
                const int vn = v + stride, i0 = view_item0 ? view_item0[vn] : vn;
                request_head_rows(G, i0, view_item0 ? view_item0[vn + 1] - i0 : 1, (STREAM ? stream_extra_item(sm, vn) : -1), c, rn);
            }
        } else {
            const int i0 = view_item0 ? view_item0[v] : v;
            load_view_head<L, false>(G, i0, view_item0 ? view_item0[v + 1] - i0 : 1, (STREAM ? stream_extra_item(sm, v) : -1), c, V, b);
        }
        eliminate(V, b, lam, invd, z);
        finish_view<L, T>(V, invd, z, coef, c, v, view_ext, Pc, Pn, VC);
    }
}

// The same round step for SMALL shards (one view per 16-lane group, a single trip), where the launch is a chain
// of latencies rather than work: wave 0 of a workgroup is the SOLVER (its first 16 lanes: decision + L x L
// solve), the other kUpdViewWaves waves own views. The accept / reject decision itself needs two scalars
// (err(candidate) from the reduce buffer, err(current) from the state), so the view waves take it redundantly
// and request and eliminate their view's block WHILE the solver works; they meet at the barrier, where dc
// arrives through LDS. (For large shards the idle solver lanes and the registers of two code paths cost more
// than the overlap gives: c3 +1.5 us, c5 +12 us; c2 -1.2 us.)
constexpr int kUpdViewWaves = 4;
constexpr int kUpdThreads = 64 * (1 + kUpdViewWaves);
constexpr int kUpdViewsPerBlock = kUpdViewWaves * 4;

template <int L, typename T, bool STREAM>
__global__ __launch_bounds__(kUpdThreads) void update_backsub_small_kernel(
        const double* __restrict__ G0, const double* __restrict__ G1, const LMState* __restrict__ st_in,
        LMState* __restrict__ st_out, const double* __restrict__ red, const int* __restrict__ view_item0,
        const int* __restrict__ view_ext, int nv, StreamMap sm, double* __restrict__ P0, double* __restrict__ P1,
        double* __restrict__ trace, T* __restrict__ VC) {
    constexpr int kSse = 2 * L * L + 2 * L + 1;
    const int tid = threadIdx.x, c = tid & 15, wave = tid >> 6;
    const bool writer = blockIdx.x == 0 && tid < 16;
    __shared__ double sdec[L + 3];
    int cur = 0;
    double lam = 0.0;
    double V[21], b[6], invd[6], z[6];
    const int v = blockIdx.x * kUpdViewsPerBlock + ((tid - 64) >> 4);       // one trip: gridDim.x covers the views
    if (wave == 0) {
        if (tid < 16) {
            double dc[L];
            const bool go = lm_update_step<L>(st_in, st_out, red, P0, P1, trace, writer, c, cur, lam, dc);
            if (c == 0) { sdec[0] = go ? 1.0 : 0.0; sdec[1] = (double)cur; sdec[2] = lam; }
            if (c < L) {
                double dci = 0.0;
#pragma unroll
                for (int j = 0; j < L; ++j) if (j == c) dci = dc[j];
                sdec[3 + c] = dci;
            }
        }
    } else {
        // the decision of lm_update_step (src/calibrate.py:155-168), from the same two numbers
        cur = st_in->cur;
        lam = st_in->lam;
        if (st_in->round == 0) {
            cur ^= 1;
        } else if (red[kSse] < st_in->err_cur) {
            cur ^= 1;
            lam = lam / 10;
        } else {
            lam = lam * 10;
        }
        if (v < nv) {
            const int i0 = view_item0 ? view_item0[v] : v;
            load_view_head<L, false>(cur ? G1 : G0, i0, view_item0 ? view_item0[v + 1] - i0 : 1, (STREAM ? stream_extra_item(sm, v) : -1), c, V, b);
            eliminate(V, b, lam, invd, z);
        }
    }
    __syncthreads();
    if (sdec[0] == 0.0 || wave == 0 || v >= nv) return;
    const double coef = c < L ? -sdec[3 + c] : (c == L ? 1.0 : 0.0);
    finish_view<L, T>(V, invd, z, coef, c, v, view_ext, cur ? P1 : P0, cur ? P0 : P1, VC);
}

// The same round step for LARGE shards, one LANE per view (round 4). With 16 lanes per view every group repeats the
// view's 6x6 Cholesky for its own right-hand side and hands 27 head values round by DPP; per four views that is ~420
// vector instructions, and on c5's 125 000-view shard the kernel was bound by fp64 issue (45 us), not by its 96 MB of
// records. Here a lane owns a view: it reads the view's six record rows itself (eight 16-byte loads per row: the
// lane's own 128-byte line), folds the shared step into the right-hand side on the fly -- rhs = g_v - E^T dc, dc
// wave-uniform in scalar registers -- factors V + lam diag(V) once, substitutes forward and back, writes the view's
// part of the next candidate and the candidate's view constants. ~500 vector instructions per SIXTY-FOUR views.
// Needs many views to fill the chip (a wave per 64 of them): the host takes it above kUpdLaneViews.
template <int L, typename T, bool STREAM>
__global__ __launch_bounds__(kSchurThreads, 3) void update_backsub_lane_kernel(
        const double* __restrict__ G0, const double* __restrict__ G1, const LMState* __restrict__ st_in,
        LMState* __restrict__ st_out, const double* __restrict__ red, const int* __restrict__ view_item0,
        const int* __restrict__ view_ext, int nv, StreamMap sm, double* __restrict__ P0, double* __restrict__ P1,
        double* __restrict__ trace, T* __restrict__ VC) {
    const int tid = threadIdx.x, c = tid & 15;
    const bool writer = blockIdx.x == 0 && tid < 16;
    __shared__ double sdec[L + 3];
    {
        int cur0 = 0;
        double lam0 = 0.0, dc0[L];
        if (tid < 16) {
            const bool go = lm_update_step<L>(st_in, st_out, red, P0, P1, trace, writer, c, cur0, lam0, dc0);
            if (c == 0) { sdec[0] = go ? 1.0 : 0.0; sdec[1] = (double)cur0; sdec[2] = lam0; }
            if (c < L) {
                double dci = 0.0;
#pragma unroll
                for (int j = 0; j < L; ++j) if (j == c) dci = dc0[j];
                sdec[3 + c] = dci;
            }
        }
    }
    __syncthreads();
    if (sdec[0] == 0.0) return;
    const int cur = (int)sdec[1];
    const double lam = sdec[2];
    double dc[L];                                              // wave-uniform: scalar registers
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const double t = sdec[3 + j];
        dc[j] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(t)),
                                 __builtin_amdgcn_readfirstlane(__double2loint(t)));
    }
    const double* G = cur ? G1 : G0;
    const double* Pc = cur ? P1 : P0;
    double* Pn = cur ? P0 : P1;
    for (int v = blockIdx.x * kSchurThreads + tid; v < nv; v += gridDim.x * kSchurThreads) {
        double V[21], gv[6], acc[6];
#pragma unroll
        for (int i = 0; i < 21; ++i) V[i] = 0.0;
#pragma unroll
        for (int m = 0; m < 6; ++m) { gv[m] = 0.0; acc[m] = 0.0; }
        // one record: rows m = 0..5 of [E^T | V], g_v in V's upper triangle (gvSlot)
        auto addRecord = [&](const double* __restrict__ g) {
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                const double2* row = reinterpret_cast<const double2*>(g + kGRows + m * 16);
                double r[16];
#pragma unroll
                for (int q = 0; q < 8; ++q) { const double2 t = row[q]; r[2 * q] = t.x; r[2 * q + 1] = t.y; }
#pragma unroll
                for (int j = 0; j < L; ++j) acc[m] = __builtin_fma(r[j], dc[j], acc[m]);
#pragma unroll
                for (int n = 0; n <= m; ++n) V[tri(m, n)] += r[L + n];
                if (m == 0) {
#pragma unroll
                    for (int j = 0; j < 5; ++j) gv[j] += r[L + 1 + j];
                }
                if (m == 1) gv[5] += r[L + 2];
            }
        };
        const int i0 = view_item0 ? view_item0[v] : v, i1 = view_item0 ? view_item0[v + 1] : v + 1;
        for (int it = i0; it < i1; ++it) addRecord(G + (int64_t)it * kGStride);
        if (STREAM) {
            const int e = stream_extra_item(sm, v);
            if (e >= 0) addRecord(G + (int64_t)e * kGStride);
        }
        double rhs[6], invd[6], z[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) rhs[m] = gv[m] - acc[m];
        eliminate(V, rhs, lam, invd, z);
        finish_view_lane<L, T>(V, invd, z, v, view_ext, Pc, Pn, VC);
    }
}

// ---------------------------------------------------------------- per-view homography LM
// Calibrator._refineHomography (src/calibrate.py:69-111) for every view at once: 16 lanes per
// view run the whole 20-iteration loop (lambda 1e-3, /10 on an accepted step, x10 otherwise,
// stop when lambda leaves (1e-10, 1e10) or the error drops below 1e-12), then H /= H[2,2].
// With p = (X, Y, 1)/w, the 9-column Jacobian rows of HomographyJacobian (src/jacobian.py:88-121)
// are Ju = (p, 0, -u p), Jv = (0, p, -v p), so J^T J = [[A,0,-uA],[0,A,-vA],[.,.,(u^2+v^2)A]] with
// A = p p^T: 24 sums + 9 gradient sums + the error per pass, reduced across the 16 lanes.
__global__ __launch_bounds__(256) void homography_lm_kernel(const int64_t* __restrict__ offs,
                                                            const double2* __restrict__ uv,
                                                            const double2* __restrict__ XY, int64_t M,
                                                            int max_iters, double* __restrict__ H) {
    const int tid = threadIdx.x, i = tid & 15;
    const int64_t view = (int64_t)blockIdx.x * 16 + (tid >> 4);
    if (view >= M) return;                     // whole 16-lane group together
    const int64_t p0 = offs[view];
    const int n = (int)(offs[view + 1] - p0);
    double h[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) h[j] = H[view * 9 + j];
    double lam = 1e-3;
    bool active = n > 0;
    for (int it = 0; it < max_iters && active; ++it) {
        double SA[6] = {0, 0, 0, 0, 0, 0}, SU[6] = {0, 0, 0, 0, 0, 0}, SV[6] = {0, 0, 0, 0, 0, 0},
               SE[6] = {0, 0, 0, 0, 0, 0}, g[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, err0 = 0.0;
        for (int q = i; q < n; q += 16) {
            const double2 m = uv[p0 + q], xy = XY[p0 + q];
            const double iw = 1.0 / (h[6] * xy.x + h[7] * xy.y + h[8]);
            const double u = (h[0] * xy.x + h[1] * xy.y + h[2]) * iw;
            const double v = (h[3] * xy.x + h[4] * xy.y + h[5]) * iw;
            const double px = xy.x * iw, py = xy.y * iw, pz = iw;
            const double a[6] = {px * px, px * py, px * pz, py * py, py * pz, pz * pz};
            const double ru = m.x - u, rv = m.y - v, e2 = u * u + v * v, rw = -(u * ru + v * rv);
#pragma unroll
            for (int j = 0; j < 6; ++j) { SA[j] += a[j]; SU[j] += u * a[j]; SV[j] += v * a[j]; SE[j] += e2 * a[j]; }
            g[0] += ru * px; g[1] += ru * py; g[2] += ru * pz;
            g[3] += rv * px; g[4] += rv * py; g[5] += rv * pz;
            g[6] += rw * px; g[7] += rw * py; g[8] += rw * pz;
            err0 += ru * ru + rv * rv;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) { SA[j] = group_sum16(SA[j]); SU[j] = group_sum16(SU[j]); SV[j] = group_sum16(SV[j]); SE[j] = group_sum16(SE[j]); }
#pragma unroll
        for (int j = 0; j < 9; ++j) g[j] = group_sum16(g[j]);
        err0 = group_sum16(err0);
        // row i of J^T J + lam diag(J^T J) | J^T r; symmetric 3x3 blocks indexed (0,0)(0,1)(0,2)(1,1)(1,2)(2,2)
        auto sym = [](const double (&S)[6], int r, int c) { const int lo = r < c ? r : c, hi = r < c ? c : r;
                                                           return S[lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)]; };
        double row[10];
#pragma unroll
        for (int j = 0; j <= 9; ++j) row[j] = 0.0;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            if (i == r) {
                const int br = r / 3, rr = r % 3;
#pragma unroll
                for (int c = 0; c < 9; ++c) {
                    const int bc = c / 3, cc = c % 3;
                    double t = 0.0;
                    if (br == bc) t = br == 2 ? sym(SE, rr, cc) : sym(SA, rr, cc);
                    else if (br + bc == 2 && br != bc) t = -sym(SU, rr, cc);          // blocks (0,2),(2,0)
                    else if (br + bc == 3) t = -sym(SV, rr, cc);                       // blocks (1,2),(2,1)
                    row[c] = t;
                }
                row[9] = g[r];
            }
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) if (i == c) row[c] += lam * row[c];
        int myCol;
        double d;
        const bool singular = gauss_jordan16<9>(row, i, myCol, d);
        double delta[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) delta[c] = group_sum16(myCol == c ? d : 0.0);
        double h1[9], err1 = 0.0;
#pragma unroll
        for (int j = 0; j < 9; ++j) h1[j] = h[j] + delta[j];
        for (int q = i; q < n; q += 16) {
            const double2 m = uv[p0 + q], xy = XY[p0 + q];
            const double iw = 1.0 / (h1[6] * xy.x + h1[7] * xy.y + h1[8]);
            const double ru = m.x - (h1[0] * xy.x + h1[1] * xy.y + h1[2]) * iw;
            const double rv = m.y - (h1[3] * xy.x + h1[4] * xy.y + h1[5]) * iw;
            err1 += ru * ru + rv * rv;
        }
        err1 = group_sum16(err1);
        const bool accept = !singular && err1 < err0;
        if (accept) {
#pragma unroll
            for (int j = 0; j < 9; ++j) h[j] = h1[j];
            lam = lam / 10;
        } else {
            lam = lam * 10;
        }
        active = (1e-10 < lam && lam < 1e10) && !(err0 < 1e-12) && !singular;
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) if (i == j) H[view * 9 + j] = h[j] / h[8];       // Href /= Href[2,2]
}

// HomographyJacobian.compute (src/jacobian.py:88-121): rows (du, dv)/dh of the projection of (X, Y, 1)
// through H = h.reshape(3,3), interleaved per point; one thread per point, out (2N, 9) row-major.
__global__ void homography_jacobian_kernel(const double* __restrict__ h, const double* __restrict__ xyz, int64_t n,
                                           double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double X = xyz[3 * i], Y = xyz[3 * i + 1];
    const double iw = 1.0 / (h[6] * X + h[7] * Y + h[8]);
    const double u = (h[0] * X + h[1] * Y + h[2]) * iw, v = (h[3] * X + h[4] * Y + h[5]) * iw;
    const double p[3] = {X * iw, Y * iw, iw};
    double* ru = out + 18 * i;
    double* rv = ru + 9;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        ru[j] = p[j];       ru[3 + j] = 0.0;   ru[6 + j] = -u * p[j];
        rv[j] = 0.0;        rv[3 + j] = p[j];  rv[6 + j] = -v * p[j];
    }
}

// ---------------------------------------------------------------- closed-form initialisation (per view)
// Normalised DLT homography (src/linearcalibrate.py:24-90), 16 lanes per view: centroid / mean
// distance normalisation of sensor and model points, then the null vector of M (2N x 9). M^T M has
// the block form [[S,0,-Su],[0,S,-Sv],[.,.,S(u^2+v^2)]] with S = sum p p^T, p = (X, Y, 1) (the same
// structure as the homography LM above), and its smallest eigenvector -- the right singular vector the
// reference takes from an SVD of M -- comes from 4 inverse iterations with the 16-lane solver.
__global__ __launch_bounds__(256) void dlt_kernel(const int64_t* __restrict__ offs, const double2* __restrict__ uv,
                                                  const double2* __restrict__ XY, int64_t M,
                                                  double* __restrict__ H) {
    const int tid = threadIdx.x, i = tid & 15;
    const int64_t view = (int64_t)blockIdx.x * 16 + (tid >> 4);
    if (view >= M) return;
    const int64_t p0 = offs[view];
    const int n = (int)(offs[view + 1] - p0);
    double su = 0, sv = 0, sX = 0, sY = 0;
    for (int q = i; q < n; q += 16) { const double2 m = uv[p0 + q], xy = XY[p0 + q]; su += m.x; sv += m.y; sX += xy.x; sY += xy.y; }
    const double inv_n = 1.0 / (double)n;
    const double mu = group_sum16(su) * inv_n, mv = group_sum16(sv) * inv_n;
    const double mX = group_sum16(sX) * inv_n, mY = group_sum16(sY) * inv_n;
    double da = 0, db = 0;
    for (int q = i; q < n; q += 16) {
        const double2 m = uv[p0 + q], xy = XY[p0 + q];
        da += sqrt((m.x - mu) * (m.x - mu) + (m.y - mv) * (m.y - mv));
        db += sqrt((xy.x - mX) * (xy.x - mX) + (xy.y - mY) * (xy.y - mY));
    }
    const double sa = 1.4142135623730951 / (group_sum16(da) * inv_n);
    const double sb = 1.4142135623730951 / (group_sum16(db) * inv_n);
    double S0[6] = {0, 0, 0, 0, 0, 0}, SU[6] = {0, 0, 0, 0, 0, 0}, SV[6] = {0, 0, 0, 0, 0, 0}, SE[6] = {0, 0, 0, 0, 0, 0};
    for (int q = i; q < n; q += 16) {
        const double2 m = uv[p0 + q], xy = XY[p0 + q];
        const double u = sa * (m.x - mu), v = sa * (m.y - mv), X = sb * (xy.x - mX), Y = sb * (xy.y - mY);
        const double a[6] = {X * X, X * Y, X, Y * Y, Y, 1.0};
        const double e2 = u * u + v * v;
#pragma unroll
        for (int j = 0; j < 6; ++j) { S0[j] += a[j]; SU[j] += u * a[j]; SV[j] += v * a[j]; SE[j] += e2 * a[j]; }
    }
    double tr = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) { S0[j] = group_sum16(S0[j]); SU[j] = group_sum16(SU[j]); SV[j] = group_sum16(SV[j]); SE[j] = group_sum16(SE[j]); }
    tr = 2 * (S0[0] + S0[3] + S0[5]) + SE[0] + SE[3] + SE[5];
    auto sym = [](const double (&S)[6], int r, int c) { const int lo = r < c ? r : c, hi = r < c ? c : r;
                                                       return S[lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)]; };
    double base[9];                        // row i of M^T M
#pragma unroll
    for (int c = 0; c < 9; ++c) base[c] = 0.0;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        if (i == r) {
            const int br = r / 3, rr = r % 3;
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const int bc = c / 3, cc = c % 3;
                double t = 0.0;
                if (br == bc) t = br == 2 ? sym(SE, rr, cc) : sym(S0, rr, cc);
                else if (br + bc == 2) t = -sym(SU, rr, cc);
                else if (br + bc == 3) t = -sym(SV, rr, cc);
                base[c] = t;
            }
        }
    }
    double x[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) x[c] = 1.0;
    const double shift = 1e-15 * tr;       // keeps the solve finite when the data fit a homography exactly
    for (int itn = 0; itn < 4; ++itn) {
        double row[10];
#pragma unroll
        for (int c = 0; c < 9; ++c) row[c] = base[c] + ((i == c) ? shift : 0.0);
        double rhs = 0.0;
#pragma unroll
        for (int c = 0; c < 9; ++c) if (i == c) rhs = x[c];
        row[9] = rhs;
        int myCol;
        double d;
        (void)gauss_jordan16<9>(row, i, myCol, d);
        double nrm = 0.0;
#pragma unroll
        for (int c = 0; c < 9; ++c) { x[c] = group_sum16(myCol == c ? d : 0.0); nrm = fmax(nrm, fabs(x[c])); }
        const double inrm = 1.0 / nrm;
#pragma unroll
        for (int c = 0; c < 9; ++c) x[c] *= inrm;
    }
    // H = Na^-1 Hp Nb with Na = [[sa,0,-sa mu],[0,sa,-sa mv],[0,0,1]], Nb likewise
    double T[9];                            // Hp Nb
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        T[3 * r + 0] = x[3 * r + 0] * sb;
        T[3 * r + 1] = x[3 * r + 1] * sb;
        T[3 * r + 2] = x[3 * r + 2] - sb * (x[3 * r + 0] * mX + x[3 * r + 1] * mY);
    }
    double Hh[9];
    const double isa = 1.0 / sa;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Hh[c] = isa * T[c] + mu * T[6 + c];
        Hh[3 + c] = isa * T[3 + c] + mv * T[6 + c];
        Hh[6 + c] = T[6 + c];
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) if (i == j) H[view * 9 + j] = Hh[j] / Hh[8];
}

// Extrinsics from a homography and A^-1 (src/linearcalibrate.py:306-371): [r0 r1 t] = A^-1 H / |A^-1 h0|,
// r2 = r0 x r1, then the nearest rotation. The reference takes U V^T from an SVD of Q = [r0 r1 r2];
// that is the orthogonal polar factor of Q, reached here by Newton's iteration X <- (X + X^-T) / 2
// (quadratic; Q is within a few percent of orthogonal, 8 steps are far past convergence).
__global__ void extrinsics_kernel(const double* __restrict__ Ainv, const double* __restrict__ H, int64_t M,
                                  double* __restrict__ W) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= M) return;
    double Q[9], Ai[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) Ai[j] = Ainv[j];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            Q[3 * r + c] = Ai[3 * r] * H[v * 9 + c] + Ai[3 * r + 1] * H[v * 9 + 3 + c] + Ai[3 * r + 2] * H[v * 9 + 6 + c];
    const double il = 1.0 / sqrt(Q[0] * Q[0] + Q[3] * Q[3] + Q[6] * Q[6]);
    const double r0[3] = {Q[0] * il, Q[3] * il, Q[6] * il}, r1[3] = {Q[1] * il, Q[4] * il, Q[7] * il};
    const double t[3] = {Q[2] * il, Q[5] * il, Q[8] * il};
    double X[9] = {r0[0], r1[0], r0[1] * r1[2] - r0[2] * r1[1],
                   r0[1], r1[1], r0[2] * r1[0] - r0[0] * r1[2],
                   r0[2], r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
    for (int itn = 0; itn < 8; ++itn) {
        // cofactor matrix C with X^-T = C / det
        const double C[9] = {X[4] * X[8] - X[5] * X[7], X[5] * X[6] - X[3] * X[8], X[3] * X[7] - X[4] * X[6],
                             X[2] * X[7] - X[1] * X[8], X[0] * X[8] - X[2] * X[6], X[1] * X[6] - X[0] * X[7],
                             X[1] * X[5] - X[2] * X[4], X[2] * X[3] - X[0] * X[5], X[0] * X[4] - X[1] * X[3]};
        const double idet = 1.0 / (X[0] * C[0] + X[1] * C[1] + X[2] * C[2]);
#pragma unroll
        for (int j = 0; j < 9; ++j) X[j] = 0.5 * (X[j] + C[j] * idet);
    }
    double* o = W + v * 16;
#pragma unroll
    for (int r = 0; r < 3; ++r) { o[4 * r] = X[3 * r]; o[4 * r + 1] = X[3 * r + 1]; o[4 * r + 2] = X[3 * r + 2]; o[4 * r + 3] = t[r]; }
    o[12] = 0.0; o[13] = 0.0; o[14] = 0.0; o[15] = 1.0;
}

// Normal equations of the linear distortion estimate D k = Ddot (src/distortion.py:110-191 radtan,
// :222-271 fisheye, same row formulas): per workgroup partial of D^T D (upper triangle) and D^T Ddot,
// summed on the host in block order and solved there (|k| <= 5 unknowns).
template <int MODEL>
__global__ __launch_bounds__(256) void distortion_normal_kernel(const double* __restrict__ A, const double* __restrict__ W,
                                                                const int* __restrict__ pt_view,
                                                                const double2* __restrict__ uv, const double2* __restrict__ XY,
                                                                const double* __restrict__ Z, int64_t MN,
                                                                double* __restrict__ part) {
    constexpr int NK = ModelTraits<MODEL>::NK;
    constexpr int NS = NK * (NK + 1) / 2 + NK;
    __shared__ double sred[4][NS];
    double acc[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) acc[j] = 0.0;
    const double fx = A[0], sk = A[1], uc = A[2], fy = A[4], vc = A[5];
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < MN; p += (int64_t)gridDim.x * 256) {
        const double* w = W + (int64_t)pt_view[p] * 16;
        const double2 xy = XY[p], m = uv[p];
        const double z = Z[p];
        const double Xc = w[0] * xy.x + w[1] * xy.y + w[2] * z + w[3];
        const double Yc = w[4] * xy.x + w[5] * xy.y + w[6] * z + w[7];
        const double Zc = w[8] * xy.x + w[9] * xy.y + w[10] * z + w[11];
        const double xn = Xc / Zc, yn = Yc / Zc, r2 = xn * xn + yn * yn;
        const double u = fx * xn + sk * yn + uc, v = fy * yn + vc;
        double du[NK], dv[NK];
        if constexpr (MODEL == kRadtan) {
            du[0] = (u - uc) * r2;            dv[0] = (v - vc) * r2;
            du[1] = (u - uc) * r2 * r2;       dv[1] = (v - vc) * r2 * r2;
            du[2] = fx * (2 * xn * yn);       dv[2] = fy * (r2 + 2 * yn * yn);
            du[3] = fx * (r2 + 2 * xn * xn);  dv[3] = fy * (2 * xn * yn);
            du[4] = (u - uc) * r2 * r2 * r2;  dv[4] = (v - vc) * r2 * r2 * r2;
        } else {
            const double r = sqrt(r2), th = atan(r), tr = th / r, t2 = th * th;
            double pw = t2;
#pragma unroll
            for (int j = 0; j < 4; ++j) { du[j] = fx * (u - uc) * tr * pw; dv[j] = fy * (v - vc) * tr * pw; pw *= t2; }
        }
        const double eu = m.x - u, ev = m.y - v;
        int idx = 0;
#pragma unroll
        for (int a = 0; a < NK; ++a)
#pragma unroll
            for (int b = a; b < NK; ++b) acc[idx++] += du[a] * du[b] + dv[a] * dv[b];
#pragma unroll
        for (int a = 0; a < NK; ++a) acc[idx++] += du[a] * eu + dv[a] * ev;
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        double t = acc[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6][j] = t;
    }
    __syncthreads();
    if (threadIdx.x < NS)
        part[(int64_t)blockIdx.x * NS + threadIdx.x] = (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
}

// ---------------------------------------------------------------- problem packing
// calib_set_problem uploads the caller's (MN,3) model points and (MN,2) sensor points as they are; this kernel
// splits them into the SoA layout of the point kernels (XY pairs, Z, uv pairs, in the storage type T) and
// writes every point's compact view index (binary search in the offsets of the non-empty views).
template <typename T>
__global__ __launch_bounds__(256) void pack_points_kernel(const double* __restrict__ xyz, const double* __restrict__ uv_in,
                                                          int uv_mode /* 0: leave, 1: convert uv_in, 2: zeros */,
                                                          const int64_t* __restrict__ voffs, int nv, int64_t MN,
                                                          typename Pair<T>::type* __restrict__ XY, T* __restrict__ Z,
                                                          typename Pair<T>::type* __restrict__ uv, int* __restrict__ pt_view) {
    using T2 = typename Pair<T>::type;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= MN) return;
    T2 xy;
    xy.x = (T)xyz[3 * p]; xy.y = (T)xyz[3 * p + 1];
    XY[p] = xy;
    Z[p] = (T)xyz[3 * p + 2];
    if (uv_mode == 1) { T2 m; m.x = (T)uv_in[2 * p]; m.y = (T)uv_in[2 * p + 1]; uv[p] = m; }
    else if (uv_mode == 2) { T2 m; m.x = T(0); m.y = T(0); uv[p] = m; }
    int lo = 0, hi = nv;                       // voffs[lo] <= p < voffs[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (voffs[mid] <= p) lo = mid; else hi = mid;
    }
    pt_view[p] = lo;
}

// ---------------------------------------------------------------- parameter vector (de)composition
// Calibrator._composeParameterVector / _decomposeParameterVector (src/calibrate.py:199-267), the per-view part:
// world-to-camera pose W (4x4) <-> (rho_x, rho_y, rho_z [degrees], t). One thread per view.
// Rotation -> Euler angles: Slabaugh's decomposition with the gimbal-lock branches of src/mathutils.py:13-33;
// np.isclose(R31, +-1) there is |R31 -+ 1| <= 1e-8 + 1e-5.
__global__ void compose_views_kernel(const double* __restrict__ W, int64_t M, int L, double* __restrict__ P) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= M) return;
    const double* w = W + v * 16;
    const double R11 = w[0], R12 = w[1], R13 = w[2], R21 = w[4], R31 = w[8], R32 = w[9], R33 = w[10];
    const double tol = 1e-8 + 1e-5;
    const double rad2deg = 57.29577951308232;
    double psi, theta, phi;
    if (fabs(R31 - 1.0) <= tol) {
        phi = 0.0; theta = -1.5707963267948966; psi = atan2(-R12, -R13);
    } else if (fabs(R31 + 1.0) <= tol) {
        phi = 0.0; theta = 1.5707963267948966; psi = atan2(R12, R13);
    } else {
        theta = -asin(R31);
        const double ct = cos(theta);
        psi = atan2(R32 / ct, R33 / ct);
        phi = atan2(R21 / ct, R11 / ct);
    }
    double* o = P + L + 6 * v;
    o[0] = psi * rad2deg; o[1] = theta * rad2deg; o[2] = phi * rad2deg;
    o[3] = w[3]; o[4] = w[7]; o[5] = w[11];
}

// Euler angles (degrees) -> R = Rz Ry Rx with the reference's numeric Rodrigues quirk (an angle with
// |theta| <= 1e-8 rad is the identity, src/mathutils.py:72-79), as view_setup_kernel; W = [R t; 0 0 0 1]
__global__ void decompose_views_kernel(const double* __restrict__ P, int64_t M, int L, double* __restrict__ W) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= M) return;
    const double* e = P + L + 6 * v;
    const double deg = 0.017453292519943295;
    double s[3], c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double th = e[a] * deg;
        sincos(th, &s[a], &c[a]);
        if (fabs(th) <= 1e-8) { s[a] = 0.0; c[a] = 1.0; }
    }
    const double sx = s[0], cx = c[0], sy = s[1], cy = c[1], sz = s[2], cz = c[2];
    double* o = W + v * 16;
    o[0] = cz * cy;  o[1] = cz * sy * sx - sz * cx;  o[2] = cz * sy * cx + sz * sx;  o[3] = e[3];
    o[4] = sz * cy;  o[5] = sz * sy * sx + cz * cx;  o[6] = sz * sy * cx - cz * sx;  o[7] = e[4];
    o[8] = -sy;      o[9] = cy * sx;                 o[10] = cy * cx;                o[11] = e[5];
    o[12] = 0.0; o[13] = 0.0; o[14] = 0.0; o[15] = 1.0;
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
