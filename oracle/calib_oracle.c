/*
 * CPU oracle in C: the same restatement as oracle/calib_oracle.py (closed-form projection and
 * Jacobian of src/distortion.py + src/jacobian.py, LM loop of src/calibrate.py:117-171 with the
 * step solved through the block-arrow / Schur form), multi-threaded over views with OpenMP.
 *
 * TEST INFRASTRUCTURE ONLY (the checker at sizes the numpy oracle is too slow for, and the
 * `cpu_baseline` leg of bench.py). Parity status: PINNED -- tests/test_oracle_golden.py holds it to
 * the numpy oracle and through it to the golden vectors produced by running the reference.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DEG 0.017453292519943295
enum { RADTAN = 0, FISHEYE = 1 };

static int num_shared(int model) { return model == RADTAN ? 10 : 9; }

/* R = Rz Ry Rx from Euler angles in degrees (src/mathutils.py:36-51); quirk = numeric Rodrigues
 * returns I when |theta| <= 1e-8 (src/mathutils.py:72-79), the symbolic Jacobian path does not */
static void euler_trig(const double* rho, int quirk, double* s, double* c) {
    for (int a = 0; a < 3; ++a) {
        double th = rho[a] * DEG;
        s[a] = sin(th);
        c[a] = cos(th);
        if (quirk && fabs(th) <= 1e-8) { s[a] = 0.0; c[a] = 1.0; }
    }
}

static void rot_from_trig(const double* s, const double* c, double* R) {
    double sx = s[0], cx = c[0], sy = s[1], cy = c[1], sz = s[2], cz = c[2];
    R[0] = cz * cy;  R[1] = cz * sy * sx - sz * cx;  R[2] = cz * sy * cx + sz * sx;
    R[3] = sz * cy;  R[4] = sz * sy * sx + cz * cx;  R[5] = sz * sy * cx - cz * sx;
    R[6] = -sy;      R[7] = cy * sx;                 R[8] = cy * cx;
}

/* distortion value + derivatives (src/distortion.py:78-108, 198-220) */
static void distort(int model, const double* k, double x, double y, double* xd, double* yd,
                    double* xd_x, double* xd_y, double* yd_y, double* dkx, double* dky) {
    double r2 = x * x + y * y;
    if (model == RADTAN) {
        double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3], k3 = k[4];
        double rad = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;
        double drad = k1 + 2 * k2 * r2 + 3 * k3 * r2 * r2;
        *xd = rad * x + 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
        *yd = rad * y + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
        *xd_x = rad + 2 * x * x * drad + 2 * p1 * y + 6 * p2 * x;
        *xd_y = 2 * x * y * drad + 2 * p1 * x + 2 * p2 * y;
        *yd_y = rad + 2 * y * y * drad + 6 * p1 * y + 2 * p2 * x;
        dkx[0] = x * r2;          dky[0] = y * r2;
        dkx[1] = x * r2 * r2;     dky[1] = y * r2 * r2;
        dkx[2] = 2 * x * y;       dky[2] = r2 + 2 * y * y;
        dkx[3] = r2 + 2 * x * x;  dky[3] = 2 * x * y;
        dkx[4] = x * r2 * r2 * r2; dky[4] = y * r2 * r2 * r2;
    } else {
        double k1 = k[0], k2 = k[1], k3 = k[2], k4 = k[3];
        double r = sqrt(r2), th = atan(r), t2 = th * th;
        double poly = 1 + k1 * t2 + k2 * t2 * t2 + k3 * t2 * t2 * t2 + k4 * t2 * t2 * t2 * t2;
        double gp = (1 + 3 * k1 * t2 + 5 * k2 * t2 * t2 + 7 * k3 * t2 * t2 * t2 + 9 * k4 * t2 * t2 * t2 * t2) / (1 + r2);
        double s, sror, thr;
        if (r < 1e-8) { s = 1.0; thr = 1.0; sror = 2 * k1 - 2.0 / 3.0; }   /* reference: 0/0 at r = 0 */
        else { thr = th / r; s = thr * poly; sror = (gp * r - th * poly) / (r2 * r); }
        *xd = s * x;  *yd = s * y;
        *xd_x = s + x * x * sror;  *xd_y = x * y * sror;  *yd_y = s + y * y * sror;
        double p = thr * t2;
        for (int j = 0; j < 4; ++j) { dkx[j] = x * p; dky[j] = y * p; p *= t2; }
    }
}

static void project_pt(int model, const double* P, const double* R, const double* t, const double* Pw,
                       double* u, double* v) {
    int L = num_shared(model);
    double Xc = R[0] * Pw[0] + R[1] * Pw[1] + R[2] * Pw[2] + t[0];
    double Yc = R[3] * Pw[0] + R[4] * Pw[1] + R[5] * Pw[2] + t[1];
    double Zc = R[6] * Pw[0] + R[7] * Pw[1] + R[8] * Pw[2] + t[2];
    double x = Xc / Zc, y = Yc / Zc, xd, yd, a, b, c, dkx[5], dky[5];
    (void)L;
    distort(model, P + 5, x, y, &xd, &yd, &a, &b, &c, dkx, dky);
    *u = P[0] * xd + P[2] * yd + P[3];
    *v = P[1] * yd + P[4];
}

/* 2 x C block of one point, rows (du, dv), columns [shared L | rx ry rz tx ty tz] */
static void jacobian_pt(int model, const double* P, const double* R, const double* t, const double* s,
                        const double* c, const double* Pw, double* Ju, double* Jv) {
    int L = num_shared(model), NK = L - 5;
    double al = P[0], be = P[1], ga = P[2];
    double q0 = R[0] * Pw[0] + R[1] * Pw[1] + R[2] * Pw[2];
    double q1 = R[3] * Pw[0] + R[4] * Pw[1] + R[5] * Pw[2];
    double q2 = R[6] * Pw[0] + R[7] * Pw[1] + R[8] * Pw[2];
    double Xc = q0 + t[0], Yc = q1 + t[1], Zc = q2 + t[2];
    double iz = 1.0 / Zc, x = Xc * iz, y = Yc * iz;
    double xd, yd, xd_x, xd_y, yd_y, dkx[5], dky[5];
    distort(model, P + 5, x, y, &xd, &yd, &xd_x, &xd_y, &yd_y, dkx, dky);
    for (int j = 0; j < L + 6; ++j) { Ju[j] = 0.0; Jv[j] = 0.0; }
    Ju[0] = xd; Jv[1] = yd; Ju[2] = yd; Ju[3] = 1.0; Jv[4] = 1.0;
    for (int j = 0; j < NK; ++j) { Ju[5 + j] = al * dkx[j] + ga * dky[j]; Jv[5 + j] = be * dky[j]; }
    double ux = al * xd_x + ga * xd_y, uy = al * xd_y + ga * yd_y, vx = be * xd_y, vy = be * yd_y;
    /* dPc/drho = (pi/180) a x q with a_x = Rz Ry e_x, a_y = Rz e_y, a_z = e_z */
    double sy = s[1], cy = c[1], sz = s[2], cz = c[2];
    double A[3][3] = {{cz * cy, sz * cy, -sy}, {-sz, cz, 0.0}, {0.0, 0.0, 1.0}};
    double d[6][3];
    for (int a = 0; a < 3; ++a) {
        d[a][0] = DEG * (A[a][1] * q2 - A[a][2] * q1);
        d[a][1] = DEG * (A[a][2] * q0 - A[a][0] * q2);
        d[a][2] = DEG * (A[a][0] * q1 - A[a][1] * q0);
    }
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) d[3 + a][b] = (a == b) ? 1.0 : 0.0;
    for (int j = 0; j < 6; ++j) {
        double dx = (d[j][0] - x * d[j][2]) * iz, dy = (d[j][1] - y * d[j][2]) * iz;
        Ju[L + j] = ux * dx + uy * dy;
        Jv[L + j] = vx * dx + vy * dy;
    }
}

/* projection, residual, compact Jacobian (MN,2,C), sum of squared residual norms */
int oracle_eval(int model, const double* P, int64_t M, const int64_t* offs, const double* sensor,
                const double* pts, double* out_y, double* out_r, double* out_Jc, double* out_sse) {
    int L = num_shared(model), C = L + 6;
    double sse = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : sse)
    for (int64_t i = 0; i < M; ++i) {
        const double* e = P + L + 6 * i;
        double s[3], c[3], Rq[9], sj[3], cj[3], Rj[9];
        euler_trig(e, 1, s, c); rot_from_trig(s, c, Rq);
        euler_trig(e, 0, sj, cj); rot_from_trig(sj, cj, Rj);
        for (int64_t p = offs[i]; p < offs[i + 1]; ++p) {
            double u, v;
            project_pt(model, P, Rq, e + 3, pts + 3 * p, &u, &v);
            if (out_y) { out_y[2 * p] = u; out_y[2 * p + 1] = v; }
            if (sensor) {
                double ru = sensor[2 * p] - u, rv = sensor[2 * p + 1] - v;
                if (out_r) { out_r[2 * p] = ru; out_r[2 * p + 1] = rv; }
                sse += ru * ru + rv * rv;
            }
            if (out_Jc) jacobian_pt(model, P, Rj, e + 3, sj, cj, pts + 3 * p, out_Jc + 2 * p * C, out_Jc + 2 * p * C + C);
        }
    }
    if (out_sse) *out_sse = sse;
    return 0;
}

static int chol6(double* V) {       /* in place, lower; returns 1 on a non-positive pivot */
    for (int j = 0; j < 6; ++j) {
        double d = V[j * 6 + j];
        for (int q = 0; q < j; ++q) d -= V[j * 6 + q] * V[j * 6 + q];
        if (!(d > 0.0)) return 1;
        d = sqrt(d);
        V[j * 6 + j] = d;
        for (int i = j + 1; i < 6; ++i) {
            double t = V[i * 6 + j];
            for (int q = 0; q < j; ++q) t -= V[i * 6 + q] * V[j * 6 + q];
            V[i * 6 + j] = t / d;
        }
    }
    return 0;
}
static void chol6_solve(const double* Lc, double* b) {
    for (int m = 0; m < 6; ++m) { double t = b[m]; for (int n = 0; n < m; ++n) t -= Lc[m * 6 + n] * b[n]; b[m] = t / Lc[m * 6 + m]; }
    for (int m = 5; m >= 0; --m) { double t = b[m]; for (int n = m + 1; n < 6; ++n) t -= Lc[n * 6 + m] * b[n]; b[m] = t / Lc[m * 6 + m]; }
}

static int solve_dense(double* S, double* s, int n) {   /* Gaussian elimination, partial pivoting */
    for (int col = 0; col < n; ++col) {
        int piv = col; double best = fabs(S[col * n + col]);
        for (int i = col + 1; i < n; ++i) if (fabs(S[i * n + col]) > best) { best = fabs(S[i * n + col]); piv = i; }
        if (!(best > 0.0)) return 1;
        if (piv != col) { for (int j = 0; j < n; ++j) { double t = S[col * n + j]; S[col * n + j] = S[piv * n + j]; S[piv * n + j] = t; }
                          double t = s[col]; s[col] = s[piv]; s[piv] = t; }
        for (int i = col + 1; i < n; ++i) {
            double f = S[i * n + col] / S[col * n + col];
            for (int j = col; j < n; ++j) S[i * n + j] -= f * S[col * n + j];
            s[i] -= f * s[col];
        }
    }
    for (int i = n - 1; i >= 0; --i) { double t = s[i]; for (int j = i + 1; j < n; ++j) t -= S[i * n + j] * s[j]; s[i] = t / S[i * n + i]; }
    return 0;
}

/* delta = (J^T J + lam diag(J^T J))^-1 J^T r through the Schur complement of the view blocks */
int oracle_step(int model, const double* P, int64_t M, const int64_t* offs, const double* sensor,
                const double* pts, double lam, double* delta) {
    int L = num_shared(model), C = L + 6;
    double* G = (double*)calloc((size_t)M * (C * C + C), sizeof(double));   /* per view Gram + gradient */
    if (!G) return 2;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M; ++i) {
        const double* e = P + L + 6 * i;
        double s[3], c[3], Rq[9], sj[3], cj[3], Rj[9], Ju[16], Jv[16];
        euler_trig(e, 1, s, c); rot_from_trig(s, c, Rq);
        euler_trig(e, 0, sj, cj); rot_from_trig(sj, cj, Rj);
        double* Gi = G + (size_t)i * (C * C + C);
        double* gi = Gi + C * C;
        for (int64_t p = offs[i]; p < offs[i + 1]; ++p) {
            double u, v;
            project_pt(model, P, Rq, e + 3, pts + 3 * p, &u, &v);
            double ru = sensor[2 * p] - u, rv = sensor[2 * p + 1] - v;
            jacobian_pt(model, P, Rj, e + 3, sj, cj, pts + 3 * p, Ju, Jv);
            for (int a = 0; a < C; ++a) {
                for (int b = 0; b <= a; ++b) Gi[a * C + b] += Ju[a] * Ju[b] + Jv[a] * Jv[b];
                gi[a] += Ju[a] * ru + Jv[a] * rv;
            }
        }
        for (int a = 0; a < C; ++a) for (int b = a + 1; b < C; ++b) Gi[a * C + b] = Gi[b * C + a];
    }
    double S[100], sv[10];
    memset(S, 0, sizeof(S)); memset(sv, 0, sizeof(sv));
    int fail = 0;
    double* Y = (double*)malloc((size_t)M * 6 * (L + 1) * sizeof(double));   /* Vh^-1 [E^T | g_v] */
    double* Lcs = (double*)malloc((size_t)M * 36 * sizeof(double));
    if (!Y || !Lcs) { free(G); free(Y); free(Lcs); return 2; }
    for (int64_t i = 0; i < M; ++i) {       /* serial: fixed summation order */
        const double* Gi = G + (size_t)i * (C * C + C);
        const double* gi = Gi + C * C;
        double* Lc = Lcs + (size_t)i * 36;
        for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) Lc[a * 6 + b] = Gi[(L + a) * C + L + b];
        for (int a = 0; a < 6; ++a) Lc[a * 6 + a] += lam * Lc[a * 6 + a];
        if (chol6(Lc)) { fail = 1; break; }
        double* Yi = Y + (size_t)i * 6 * (L + 1);
        for (int cc = 0; cc <= L; ++cc) {
            double b[6];
            for (int m = 0; m < 6; ++m) b[m] = cc < L ? Gi[cc * C + L + m] : gi[L + m];
            chol6_solve(Lc, b);
            for (int m = 0; m < 6; ++m) Yi[m * (L + 1) + cc] = b[m];
        }
        for (int a = 0; a < L; ++a) {
            for (int b = 0; b < L; ++b) {
                double t = Gi[a * C + b];
                for (int m = 0; m < 6; ++m) t -= Gi[a * C + L + m] * Yi[m * (L + 1) + b];
                S[a * L + b] += t;
            }
            double t = gi[a];
            for (int m = 0; m < 6; ++m) t -= Gi[a * C + L + m] * Yi[m * (L + 1) + L];
            sv[a] += t;
        }
    }
    if (!fail) {
        /* S so far = sum (B_i - E Vh^-1 E^T); add lam * diag(sum B_i) */
        for (int a = 0; a < L; ++a) {
            double bd = 0.0;
            for (int64_t i = 0; i < M; ++i) bd += G[(size_t)i * (C * C + C) + a * C + a];
            S[a * L + a] += lam * bd;
        }
        fail = solve_dense(S, sv, L);
    }
    if (!fail) {
        for (int a = 0; a < L; ++a) delta[a] = sv[a];
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < M; ++i) {
            const double* Yi = Y + (size_t)i * 6 * (L + 1);
            for (int m = 0; m < 6; ++m) {
                double t = Yi[m * (L + 1) + L];
                for (int a = 0; a < L; ++a) t -= Yi[m * (L + 1) + a] * sv[a];
                delta[L + 6 * i + m] = t;
            }
        }
    }
    free(G); free(Y); free(Lcs);
    return fail;
}

/* the loop of src/calibrate.py:143-171; returns 0, or 1 when a step is singular.
 * trace rows: iter, err(P), err(P+delta), lambda, accepted */
int oracle_refine(int model, double* P, int64_t M, const int64_t* offs, const double* sensor,
                  const double* pts, int max_iters, double lam, double lam_min, double lam_max,
                  double err_min, double* out_sse, int* out_iters, double* trace) {
    int L = num_shared(model);
    int64_t K = L + 6 * M;
    double* delta = (double*)malloc((size_t)K * sizeof(double));
    double* P1 = (double*)malloc((size_t)K * sizeof(double));
    if (!delta || !P1) { free(delta); free(P1); return 2; }
    double err = 0.0, err1 = 0.0;
    int it = 0, rc = 0;
    for (; it < max_iters; ++it) {
        rc = oracle_step(model, P, M, offs, sensor, pts, lam, delta);
        if (rc) break;
        oracle_eval(model, P, M, offs, sensor, pts, NULL, NULL, NULL, &err);
        for (int64_t i = 0; i < K; ++i) P1[i] = P[i] + delta[i];
        oracle_eval(model, P1, M, offs, sensor, pts, NULL, NULL, NULL, &err1);
        int acc = err1 < err;
        if (trace) { double* row = trace + 5 * it; row[0] = it; row[1] = err; row[2] = err1; row[3] = lam; row[4] = acc; }
        if (acc) { memcpy(P, P1, (size_t)K * sizeof(double)); lam /= 10; } else lam *= 10;
        if (!(lam_min < lam && lam < lam_max) || err < err_min) { ++it; break; }
    }
    if (out_sse) *out_sse = err;
    if (out_iters) *out_iters = it;
    free(delta); free(P1);
    return rc;
}
