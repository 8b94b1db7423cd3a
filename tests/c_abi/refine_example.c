/* A plain C99 caller of include/calib_lm.h, compiled with gcc and linked against libcalib_lm.so by
 * tests/test_gpu_parity.py::test_c_program_against_the_header: INTEGRATION.md section C's sequence
 * (create, set_problem, refine_awk, decompose, destroy) plus the error paths a C caller sees.
 *
 *   refine_example <problem.bin> <result.bin>
 * problem.bin: int64 model, int64 M, int64 offsets[M+1], double sensor[MN*2], double model_xyz[MN*3],
 *              double A[9], double W[M*16], double k[nk]
 * result.bin:  double sse, double iters, double A[9], double k[nk], double W[M*16], double P[L+6M]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "calib_lm.h"

#define CHECK(call)                                                                       \
    do {                                                                                  \
        int rc__ = (call);                                                                \
        if (rc__ != CALIB_OK) {                                                           \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc__, calib_last_error());           \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)

static int read_exact(FILE* f, void* dst, size_t bytes) { return fread(dst, 1, bytes, f) == bytes ? 0 : -1; }

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s problem.bin result.bin\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int64_t model = 0, M = 0;
    if (read_exact(f, &model, 8) || read_exact(f, &M, 8)) return 1;
    const int nk = model == CALIB_MODEL_RADTAN ? 5 : 4, L = 5 + nk;
    int64_t* offs = (int64_t*)malloc((size_t)(M + 1) * 8);
    if (read_exact(f, offs, (size_t)(M + 1) * 8)) return 1;
    const int64_t MN = offs[M];
    double* sensor = (double*)malloc((size_t)MN * 16);
    double* xyz = (double*)malloc((size_t)MN * 24);
    double A[9], k[5];
    double* W = (double*)malloc((size_t)M * 128);
    if (read_exact(f, sensor, (size_t)MN * 16) || read_exact(f, xyz, (size_t)MN * 24) || read_exact(f, A, 72) ||
        read_exact(f, W, (size_t)M * 128) || read_exact(f, k, (size_t)nk * 8)) return 1;
    fclose(f);

    int ndev = 0;
    CHECK(calib_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no HIP device\n"); return 3; }
    printf("calib_version %d, %d device(s)\n", calib_version(), ndev);

    calib_handle_t h = NULL;
    /* error paths first: bad arguments come back as status codes with a message, nothing aborts */
    if (calib_create(7, CALIB_DTYPE_F64, 0, &h) != CALIB_E_INVALID || h != NULL) return 4;
    if (calib_create((int)model, CALIB_DTYPE_F64, ndev + 5, &h) != CALIB_E_HIP) return 4;
    CHECK(calib_create((int)model, CALIB_DTYPE_F64, 0, &h));
    double sse = 0.0;
    int iters = 0;
    if (calib_refine_awk(h, A, W, k, 10, 1e-3, 1e-10, 1e10, 1e-12, &sse, &iters, NULL) != CALIB_E_STATE) return 4;   /* no problem yet */
    CHECK(calib_set_problem(h, M, offs, sensor, xyz));
    if (calib_refine_awk(h, A, W, k, 0, 1e-3, 1e-10, 1e10, 1e-12, &sse, &iters, NULL) != CALIB_E_INVALID) return 4;  /* maxIters = 0 */
    int Lh = 0;
    int64_t K = 0;
    CHECK(calib_num_shared(h, &Lh));
    CHECK(calib_num_params(h, &K));
    if (Lh != L || K != L + 6 * M) return 4;

    /* the reference's refineCalibrationParameters(Ainitial, Winitial, kInitial, allDetections, maxIters) */
    double* trace = (double*)calloc((size_t)100 * (CALIB_TRACE_HEADER + L), 8);
    CHECK(calib_refine_awk(h, A, W, k, 100, 1e-3, 1e-10, 1e10, 1e-12, &sse, &iters, trace));
    printf("refined in %d LM iterations, sse %.3e, lambda of the last iteration %.1e\n", iters, sse,
           iters > 0 ? trace[(size_t)(iters - 1) * (CALIB_TRACE_HEADER + L) + 3] : 0.0);
    double* P = (double*)malloc((size_t)K * 8);
    CHECK(calib_compose_params((int)model, M, A, W, k, P, 0));
    double err = 0.0;
    CHECK(calib_eval(h, P, NULL, NULL, NULL, &err));          /* _computeReprojectionError at the result */
    printf("reprojection error at the result %.3e\n", err);
    CHECK(calib_destroy(h));

    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 1; }
    const double it = (double)iters;
    fwrite(&sse, 8, 1, f); fwrite(&it, 8, 1, f); fwrite(A, 8, 9, f); fwrite(k, 8, (size_t)nk, f);
    fwrite(W, 8, (size_t)M * 16, f); fwrite(P, 8, (size_t)K, f); fwrite(&err, 8, 1, f);
    fclose(f);
    free(offs); free(sensor); free(xyz); free(W); free(trace); free(P);
    return 0;
}
