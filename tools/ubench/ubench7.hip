// Whole-chip sustained fp64 VALU FMA rate (v_fma_f64, 8 independent chains per lane), and the same with
// fp64 MFMAs of co-resident waves: is the chip-level limit that holds v_mfma_f64 at ~47 TFLOP/s (ubench6)
// a limit of the matrix pipe alone?
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench7 ubench7.hip ; run: ./ubench7
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));

// kind: 0 = FMA only, 1 = MFMA only, 2 = every other wave MFMA, the rest FMA
__global__ __launch_bounds__(256) void stream(int iters, int kind, double* out) {
    const int wave = threadIdx.x >> 6;
    const bool doMfma = kind == 1 || (kind == 2 && ((blockIdx.x + wave) & 1) == 0);
    double s = 0.0;
    if (doMfma) {
        d4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
        const double x = 1.0 + 1e-9 * threadIdx.x;
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a1, 0, 0, 0);
        }
        s = a0[0] + a1[1] + a0[2] + a1[3];
    } else {
        double c[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = 1.0 + j + 1e-9 * threadIdx.x;
        const double m = 1.0 - 1e-12, b = 1e-13;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r)            // 32 FMAs per trip = the nominal issue time of 2 MFMAs (128 cycles)
#pragma unroll
                for (int j = 0; j < 8; ++j) c[j] = __builtin_fma(c[j], m, b);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s += c[j];
    }
    if (s == 12345.678) out[0] = s;
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    double* out;
    hipMalloc(&out, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"fp64 FMA only", "fp64 MFMA only", "half the waves MFMA, half FMA"};
    for (int iters : {300, 20000}) {
        for (int cuUse : {cus, cus / 8}) {
            for (int kind = 0; kind < 3; ++kind) {
                const int blocks = cuUse * 4;                        // 4 waves per SIMD
                hipLaunchKernelGGL(stream, dim3(blocks), dim3(256), 0, 0, iters, kind, out);
                hipDeviceSynchronize();
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(stream, dim3(blocks), dim3(256), 0, 0, iters, kind, out);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                const double t = ms * 1e-3;
                const double waves = (double)blocks * 4;
                double fmaW = kind == 0 ? waves : (kind == 2 ? waves / 2 : 0), mfmaW = kind == 1 ? waves : (kind == 2 ? waves / 2 : 0);
                const double fmaFlop = fmaW * iters * 32 * 64 * 2.0, mfmaFlop = mfmaW * iters * 2 * 2048.0;
                printf("iters %6d CUs %3d %-32s %9.1f us  FMA %6.2f TFLOP/s  MFMA %6.2f TFLOP/s\n", iters, cuUse, names[kind],
                       t * 1e6, fmaFlop / t * 1e-12, mfmaFlop / t * 1e-12);
            }
        }
    }
    return 0;
}
