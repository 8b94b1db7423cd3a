// Per-point projection + analytic Jacobian (device code, gfx950).
//
// Replaces the sympy-differentiated, lambdified expression the reference evaluates
// per view (src/jacobian.py:19-36,147-172) by the closed-form chain rule through
//   Pc = R(rho) Pw + t            src/distortion.py:26-37, src/mathutils.py:36-51
//   (x, y) = Pc.xy / Pc.z          src/mathutils.py:174-192
//   distort                        src/distortion.py:78-108 (radtan), :198-220 (fisheye)
//   u = a xd + g yd + uc, v = b yd + vc    src/distortion.py:57-58
// T is the storage/evaluation type (double or float).
#pragma once
#include <hip/hip_runtime.h>

namespace calib {

constexpr int kRadtan = 0;
constexpr int kFisheye = 1;

template <int MODEL> struct ModelTraits;
template <> struct ModelTraits<kRadtan>  { static constexpr int NK = 5; static constexpr int L = 10; static constexpr int C = 16; };
template <> struct ModelTraits<kFisheye> { static constexpr int NK = 4; static constexpr int L = 9;  static constexpr int C = 15; };

// Per-view constants staged through LDS (written by view_setup_kernel):
//   [0..9)  R row-major   [9..12) t   [12..15) (pi/180) * Rz Ry e_x   [15..17) (pi/180) * (Rz e_y).xy
constexpr int kViewStride = 18;

template <typename T> struct Pair;
template <> struct Pair<double> { using type = double2; };
template <> struct Pair<float>  { using type = float2; };

template <typename T> __device__ __forceinline__ T t_sqrt(T x);
template <> __device__ __forceinline__ double t_sqrt<double>(double x) { return sqrt(x); }
template <> __device__ __forceinline__ float  t_sqrt<float>(float x)   { return sqrtf(x); }
template <typename T> __device__ __forceinline__ T t_atan(T x);
template <> __device__ __forceinline__ double t_atan<double>(double x) { return atan(x); }
template <> __device__ __forceinline__ float  t_atan<float>(float x)   { return atanf(x); }

// ---- reciprocal, reciprocal square root and arctangent for the per-point model -----------------
// The library forms (IEEE division: div_scale / rcp / 2 Newton steps / div_fmas / div_fixup, 10
// instructions; sqrt 15; atan ~55 with 20 coefficients parked in 40 VGPRs) are what the fused kernel's
// producer waves spend a third of their VALU issue slots on. These are accurate to 1-2 ulp, which is
// all the parity bars ask (projection 1e-12 of the pixel scale, Jacobian columns 1e-12 relative). The hardware seeds
// are good to ~2^-26; one third-order correction (error cubed) reaches the last bit as two Newton steps did, with a
// shorter chain (rcp: 4 dependent instructions instead of 5, rsq: 6 instead of 8).
__device__ __forceinline__ double fast_rcp(double x) {          // v_rcp_f64 seed + one third-order step
    const double r = __builtin_amdgcn_rcp(x);                   // r (1 + e + e^2), e = 1 - x r: error e^3
    const double e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(__builtin_fma(e, e, e), r, r);
}
__device__ __forceinline__ float fast_rcp(float x) { return 1.0f / x; }
__device__ __forceinline__ double fast_rsqrt(double x) {        // v_rsq_f64 seed + one third-order step
    const double r = __builtin_amdgcn_rsq(x);                   // r (1 + e/2 + 3 e^2/8), e = 1 - x r^2: error ~ e^3
    const double e = __builtin_fma(-(x * r), r, 1.0);
    return __builtin_fma(r * e, __builtin_fma(0.375, e, 0.5), r);
}
__device__ __forceinline__ float fast_rsqrt(float x) { return 1.0f / sqrtf(x); }

// Coefficients of P below, highest degree first. They are read with scalar loads (uniform address -> SGPR
// pairs) and enter each Horner step as the scalar operand of a v_fma_f64 written in inline asm: one VALU
// instruction per step. Left to itself the compiler keeps all 20 constants in 40 VGPRs for the whole kernel
// and still spends a v_mov_b64 per step to set up a two-address v_fmac_f64.
__device__ const double kAtanPoly[19] = {
    -1.93423475928923e-05,   0.00021423810738603946, -0.0011252544302234645, 0.003751138483965141,
    -0.00899108054265826,    0.01671959606350739,    -0.02556862364437174,   0.03387126702700675,
    -0.040811247503178855,   0.04668745304848529,    -0.052374234719188166,  0.058768281144872724,
    -0.06665764910689723,    0.07692198997458294,    -0.09090899793217341,   0.11111110578002083,
    -0.14285714266926733,    0.1999999999964796,     -0.333333333333307};

__device__ __forceinline__ double fma_sconst(double a, double b, double sc) {      // a * b + sc, sc in SGPRs
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(sc));
    return r;
}

// atan(r) for r >= 0 given ir = 1 / r (any value when r == 0): t = min(r, 1/r) in [0, 1],
// atan(t) = t P(t^2) with P the degree-19 interpolant of atan(sqrt z) / sqrt z at the Chebyshev nodes of
// [0, 1] (coefficients computed with mpmath at 60 digits, max relative error 2.5e-16), pi/2 - atan(1/r) above 1.
__device__ __forceinline__ double atan_pos(double r, double ir) {
    const bool big = r > 1.0;
    const double t = big ? ir : r;
    const double z = t * t;
    double p = kAtanPoly[0];
#pragma unroll
    for (int j = 1; j < 19; ++j) {
        p = fma_sconst(p, z, kAtanPoly[j]);
    }
    p = __builtin_fma(p, z, 1.0);
    const double a = t * p;
    return big ? 1.5707963267948966 - a : a;
}
__device__ __forceinline__ float atan_pos(float r, float) { return atanf(r); }

template <int MODEL, typename T>
struct Shared {               // the L shared parameters, converted once per thread
    T al, be, ga, uc, vc;
    T k[ModelTraits<MODEL>::NK];
    __device__ __forceinline__ void load(const double* __restrict__ P) {
        al = (T)P[0]; be = (T)P[1]; ga = (T)P[2]; uc = (T)P[3]; vc = (T)P[4];
#pragma unroll
        for (int j = 0; j < ModelTraits<MODEL>::NK; ++j) k[j] = (T)P[5 + j];
    }
};

// Distortion in two steps. The CORE is the long dependent chains (radtan: the radial polynomial and its derivative;
// fisheye: 1/r, the arctangent, the two polynomials in theta^2): a handful of values per point. The rest -- the
// distorted point, its derivatives, the coefficient columns -- are a few independent products of them.
template <int MODEL, typename T> struct DistortCore;
template <typename T> struct DistortCore<kRadtan, T> { T r2, rad, drad; };
template <typename T> struct DistortCore<kFisheye, T> { T s, sror, p, t2; };     // p = theta^3 / r (the k1 factor)

template <int MODEL, typename T>
__device__ __forceinline__ void distort_core(const T* __restrict__ k, T x, T y, DistortCore<MODEL, T>& c) {
    const T r2 = x * x + y * y;
    if constexpr (MODEL == kRadtan) {
        const T k1 = k[0], k2 = k[1], k3 = k[4];
        c.r2 = r2;
        c.rad = T(1) + r2 * (k1 + r2 * (k2 + r2 * k3));
        c.drad = k1 + r2 * (T(2) * k2 + T(3) * k3 * r2);
    } else {
        const T k1 = k[0], k2 = k[1], k3 = k[2], k4 = k[3];
        // 1 / r from a clamped r^2: at r = 0 (a point on the optical axis) ir stays finite and r = r2 * ir = 0,
        // so theta = 0 and everything below is finite; the r -> 0 limits are selected by the r2 test
        const T tiny = sizeof(T) == 8 ? T(1e-300) : T(1e-30);
        const T ir = fast_rsqrt(r2 > tiny ? r2 : tiny);
        const T r = r2 * ir;
        const T th = atan_pos(r, ir);
        const T t2 = th * th;
        const T poly = T(1) + t2 * (k1 + t2 * (k2 + t2 * (k3 + t2 * k4)));
        const T gp = (T(1) + t2 * (T(3) * k1 + t2 * (T(5) * k2 + t2 * (T(7) * k3 + T(9) * k4 * t2))))
                     * fast_rcp(T(1) + r2);
        // s = theta poly / r, s_r / r = (g' - s) / r^2; analytic limits at r -> 0
        // (the reference evaluates 0/0 = NaN exactly at r = 0, src/distortion.py:215).
        T thr;
        if (r2 < T(1e-16)) {                            // r < 1e-8
            c.s = T(1); thr = T(1); c.sror = T(2) * k1 - T(2) / T(3);
        } else {
            thr = th * ir;
            c.s = thr * poly;
            c.sror = (gp - c.s) * ir * ir;
        }
        c.p = thr * t2;
        c.t2 = t2;
    }
}

// dk[j] = (dxd/dk_j, dyd/dk_j) for the tangential coefficients (radtan p1, p2); for a RADIAL coefficient, whose
// derivative is (x, y) f_j, both entries hold the factor f_j (jacobian_stage_b multiplies (x, y) in once for all of them).
template <int MODEL, typename T>
__device__ __forceinline__ void distort_finish(const T* __restrict__ k, T x, T y, const DistortCore<MODEL, T>& c,
                                               T& xd, T& yd, T& xd_x, T& xd_y, T& yd_y,
                                               T (&dkx)[ModelTraits<MODEL>::NK], T (&dky)[ModelTraits<MODEL>::NK]) {
    if constexpr (MODEL == kRadtan) {
        const T p1 = k[2], p2 = k[3];
        const T r2 = c.r2, rad = c.rad, drad = c.drad;
        const T r4 = r2 * r2, r6 = r4 * r2;
        const T xy = x * y, xx = x * x, yy = y * y;
        xd = rad * x + T(2) * p1 * xy + p2 * (r2 + T(2) * xx);
        yd = rad * y + p1 * (r2 + T(2) * yy) + T(2) * p2 * xy;
        xd_x = rad + T(2) * xx * drad + T(2) * p1 * y + T(6) * p2 * x;
        xd_y = T(2) * xy * drad + T(2) * p1 * x + T(2) * p2 * y;
        yd_y = rad + T(2) * yy * drad + T(6) * p1 * y + T(2) * p2 * x;
        dkx[0] = r2;  dky[0] = r2;          // radial terms k1, k2, k3: the common factor f_j of (x, y) f_j
        dkx[1] = r4;  dky[1] = r4;
        dkx[4] = r6;  dky[4] = r6;
        dkx[2] = T(2) * xy;           dky[2] = r2 + T(2) * yy;
        dkx[3] = r2 + T(2) * xx;      dky[3] = T(2) * xy;
    } else {
        xd = c.s * x;  yd = c.s * y;
        xd_x = c.s + x * x * c.sror;
        xd_y = x * y * c.sror;
        yd_y = c.s + y * y * c.sror;
        T p = c.p;
#pragma unroll
        for (int j = 0; j < 4; ++j) { dkx[j] = p; dky[j] = p; p *= c.t2; }     // all four radial: the factor f_j
    }
}

// Distortion value + first derivatives at normalised (x, y).
template <int MODEL, typename T>
__device__ __forceinline__ void distort(const T* __restrict__ k, T x, T y,
                                        T& xd, T& yd, T& xd_x, T& xd_y, T& yd_y,
                                        T (&dkx)[ModelTraits<MODEL>::NK],
                                        T (&dky)[ModelTraits<MODEL>::NK]) {
    DistortCore<MODEL, T> c;
    distort_core<MODEL, T>(k, x, y, c);
    distort_finish<MODEL, T>(k, x, y, c, xd, yd, xd_x, xd_y, yd_y, dkx, dky);
}

// Forward projection only (candidate-error style evaluation).
template <int MODEL, typename T>
__device__ __forceinline__ void project_point(const Shared<MODEL, T>& sp, const T* __restrict__ vc,
                                              T X, T Y, T Z, T& u, T& v) {
    const T Xc = vc[0] * X + vc[1] * Y + vc[2] * Z + vc[9];
    const T Yc = vc[3] * X + vc[4] * Y + vc[5] * Z + vc[10];
    const T Zc = vc[6] * X + vc[7] * Y + vc[8] * Z + vc[11];
    const T iz = fast_rcp(Zc);
    T xd, yd, a, b, c;
    T dkx[ModelTraits<MODEL>::NK], dky[ModelTraits<MODEL>::NK];
    distort<MODEL, T>(sp.k, Xc * iz, Yc * iz, xd, yd, a, b, c, dkx, dky);
    u = sp.al * xd + sp.ga * yd + sp.uc;
    v = sp.be * yd + sp.vc;
}

// What the first half of the per-point model hands to the second: the projection and the distortion core (stage A:
// long dependent chains -- reciprocal, reciprocal square root, arctangent, polynomials); stage B forms the distorted
// point, (u, v) and the 2 x C block from it (wide, independent products).
template <int MODEL, typename T>
struct PointState {
    T x, y, iz, q0, q1, q2;
    DistortCore<MODEL, T> core;
};

template <int MODEL, typename T>
__device__ __forceinline__ void jacobian_stage_a(const Shared<MODEL, T>& sp, const T* __restrict__ vc, T X, T Y, T Z,
                                                 PointState<MODEL, T>& st) {
    st.q0 = vc[0] * X + vc[1] * Y + vc[2] * Z;
    st.q1 = vc[3] * X + vc[4] * Y + vc[5] * Z;
    st.q2 = vc[6] * X + vc[7] * Y + vc[8] * Z;
    const T Xc = st.q0 + vc[9], Yc = st.q1 + vc[10], Zc = st.q2 + vc[11];
    st.iz = fast_rcp(Zc);
    st.x = Xc * st.iz;
    st.y = Yc * st.iz;
    distort_core<MODEL, T>(sp.k, st.x, st.y, st.core);
}

template <int MODEL, typename T>
__device__ __forceinline__ void jacobian_stage_b(const Shared<MODEL, T>& sp, const T* __restrict__ vc,
                                                 const PointState<MODEL, T>& st, T& u, T& v,
                                                 typename Pair<T>::type (&J)[ModelTraits<MODEL>::C]) {
    constexpr int NK = ModelTraits<MODEL>::NK;
    constexpr int L = ModelTraits<MODEL>::L;
    const T x = st.x, y = st.y, iz = st.iz, q0 = st.q0, q1 = st.q1, q2 = st.q2;
    T xd, yd, xd_x, xd_y, yd_y;
    T dkx[NK], dky[NK];
    distort_finish<MODEL, T>(sp.k, x, y, st.core, xd, yd, xd_x, xd_y, yd_y, dkx, dky);
    const T yd_x = xd_y;

    u = sp.al * xd + sp.ga * yd + sp.uc;
    v = sp.be * yd + sp.vc;

    J[0].x = xd;   J[0].y = T(0);
    J[1].x = T(0); J[1].y = yd;
    J[2].x = yd;   J[2].y = T(0);
    J[3].x = T(1); J[3].y = T(0);
    J[4].x = T(0); J[4].y = T(1);
    // radial coefficients: d(xd, yd)/dk_j = (x, y) f_j, so (du, dv)/dk_j = (al x + ga y, be y) f_j -- two
    // multiplications per column instead of five (distort<> hands back f_j for them); the tangential p1, p2 in full
    {
        const T ax = sp.al * x + sp.ga * y, by = sp.be * y;
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            const bool radial = MODEL == kFisheye || j < 2 || j == 4;
            if (radial) {
                J[5 + j].x = ax * dkx[j];
                J[5 + j].y = by * dkx[j];
            } else {
                J[5 + j].x = sp.al * dkx[j] + sp.ga * dky[j];
                J[5 + j].y = sp.be * dky[j];
            }
        }
    }
    // d(u,v)/d(x,y), pre-scaled by 1/Zc
    const T ux = (sp.al * xd_x + sp.ga * yd_x) * iz;
    const T uy = (sp.al * xd_y + sp.ga * yd_y) * iz;
    const T vx = sp.be * yd_x * iz;
    const T vy = sp.be * yd_y * iz;
    // dPc/drho. = (pi/180) a. x q,  a_x = Rz Ry e_x, a_y = Rz e_y, a_z = e_z  (q = R Pw)
    const T ax0 = vc[12], ax1 = vc[13], ax2 = vc[14], ay0 = vc[15], ay1 = vc[16];
    const T deg = T(0.017453292519943295);
    T dX, dY, dZ, dx, dy;
    // rho_x
    dX = ax1 * q2 - ax2 * q1;  dY = ax2 * q0 - ax0 * q2;  dZ = ax0 * q1 - ax1 * q0;
    dx = dX - x * dZ;  dy = dY - y * dZ;
    J[L + 0].x = ux * dx + uy * dy;  J[L + 0].y = vx * dx + vy * dy;
    // rho_y  (a_y.z = 0)
    dX = ay1 * q2;  dY = -ay0 * q2;  dZ = ay0 * q1 - ay1 * q0;
    dx = dX - x * dZ;  dy = dY - y * dZ;
    J[L + 1].x = ux * dx + uy * dy;  J[L + 1].y = vx * dx + vy * dy;
    // rho_z  (e_z x q = (-q1, q0, 0))
    dx = -deg * q1;  dy = deg * q0;
    J[L + 2].x = ux * dx + uy * dy;  J[L + 2].y = vx * dx + vy * dy;
    // t
    J[L + 3].x = ux;  J[L + 3].y = vx;
    J[L + 4].x = uy;  J[L + 4].y = vy;
    J[L + 5].x = -(ux * x + uy * y);  J[L + 5].y = -(vx * x + vy * y);
}

// Projection + the point's 2 x C Jacobian block, J[c] = (du/dp_c, dv/dp_c) (one 16-byte pair per
// column: what the HBM layout and the LDS transpose move). Column order is the reference's
// (src/jacobian.py:22-26): [alpha beta gamma uc vc | k.. | rx ry rz tx ty tz].
template <int MODEL, typename T>
__device__ __forceinline__ void jacobian_point(const Shared<MODEL, T>& sp, const T* __restrict__ vc,
                                               T X, T Y, T Z, T& u, T& v,
                                               typename Pair<T>::type (&J)[ModelTraits<MODEL>::C]) {
    PointState<MODEL, T> st;
    jacobian_stage_a<MODEL, T>(sp, vc, X, Y, Z, st);
    jacobian_stage_b<MODEL, T>(sp, vc, st, u, v, J);
}

}  // namespace calib
