// Sustained v_mfma_f64_16x16x4_f64 rate of the WHOLE chip (not one SIMD): does a grid that keeps every SIMD
// streaming fp64 MFMAs reach the 78.6 TFLOP/s the data sheet gives (64 cycles per instruction at 2.4 GHz)?
// Varies: waves per SIMD (1, 2, 4), independent accumulators per wave (1, 2, 4), CUs used, run length.
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench6 ubench6.hip ; run: ./ubench6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NA>
__global__ __launch_bounds__(256) void mfma_stream(int iters, double* out) {
    d4 acc[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) acc[a] = d4{0.0, 0.0, 0.0, 0.0};
    const double x = 1.0 + 1e-9 * threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NA; ++a) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc[a], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < NA; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    if (s == 12345.678) out[0] = s;
}

template <int NA>
double run(int blocks, int threads, int iters) {
    double* out;
    hipMalloc(&out, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_stream<NA>, dim3(blocks), dim3(threads), 0, 0, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_stream<NA>, dim3(blocks), dim3(threads), 0, 0, iters, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms * 1e-3;
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("CUs %d\n", cus);
    const double flop = 2.0 * 16 * 16 * 4;
    struct Cfg { int wps, na, cuFrac, iters; };
    for (int iters : {200, 20000}) {
        for (int cuUse : {cus, cus / 8, 1}) {
            for (int wps : {1, 2, 4}) {
                for (int na : {1, 2, 4}) {
                    const int threads = 256;                  // 4 waves = 1 per SIMD
                    const int blocks = cuUse * wps;           // wps workgroups per CU
                    double t = na == 1 ? run<1>(blocks, threads, iters) : (na == 2 ? run<2>(blocks, threads, iters) : run<4>(blocks, threads, iters));
                    const double n = (double)blocks * 4 * iters * na;
                    const double cyc = t * 2.4e9 / ((double)iters * na * wps);
                    printf("iters %6d CUs %3d waves/SIMD %d acc/wave %d: %8.1f us  %6.2f TFLOP/s  %6.1f cycles(2.4GHz)/MFMA/SIMD\n",
                           iters, cuUse, wps, na, t * 1e6, n * flop / t * 1e-12, cyc);
                }
            }
        }
    }
    return 0;
}
