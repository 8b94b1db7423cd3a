// Whole-chip sustained rate of v_mfma_f64_4x4x4_4b (4 independent 4x4x4 blocks per instruction, 256 MACs) next to
// v_mfma_f64_16x16x4 (1024 MACs): under the chip-level limit of ubench6, is the cap on MACs or on instructions?
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench8 ubench8.hip ; run: ./ubench8
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void stream(int iters, double* out) {
    const double x = 1.0 + 1e-9 * threadIdx.x;
    double s = 0.0;
    if (KIND == 0) {
        d4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a1, 0, 0, 0);
        }
        s = a0[0] + a1[1] + a0[2] + a1[3];
    } else {
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, a3, 0, 0, 0);
            a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, a4, 0, 0, 0);
            a5 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, a5, 0, 0, 0);
            a6 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, a6, 0, 0, 0);
            a7 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, a7, 0, 0, 0);
        }
        s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
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
    for (int iters : {300, 20000}) {
        for (int cuUse : {cus, cus / 8}) {
            for (int kind = 0; kind < 2; ++kind) {
                const int blocks = cuUse * 4;
                auto launch = [&]() {
                    if (kind == 0) hipLaunchKernelGGL(stream<0>, dim3(blocks), dim3(256), 0, 0, iters, out);
                    else hipLaunchKernelGGL(stream<1>, dim3(blocks), dim3(256), 0, 0, iters, out);
                };
                launch();
                hipDeviceSynchronize();
                hipEventRecord(e0, 0);
                launch();
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                const double t = ms * 1e-3, waves = (double)blocks * 4;
                const double macs = kind == 0 ? waves * iters * 2 * 1024.0 : waves * iters * 8 * 256.0;
                printf("iters %6d CUs %3d %-24s %9.1f us  %6.2f TFLOP/s\n", iters, cuUse,
                       kind == 0 ? "v_mfma_f64_16x16x4" : "v_mfma_f64_4x4x4_4b", t * 1e6, 2 * macs / t * 1e-12);
            }
        }
    }
    return 0;
}
