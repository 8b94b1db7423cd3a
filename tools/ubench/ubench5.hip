// Issue cost of v_mfma_f64_4x4x4_4b_f64 (4 independent 4x4x4 blocks per instruction) vs v_mfma_f64_16x16x4_f64,
// and of v_mfma_f32_16x16x4_f32 beside fp32 / fp64 FMA waves (does the fp32 MFMA block the VALU of its SIMD mates
// the way the fp64 one does?).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>   // 0: f64 16x16x4, 1: f64 4x4x4_4b, 2: f32 16x16x4
__global__ void issue_kernel(double* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    d4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    double b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    f4 f0 = {0, 0, 0, 0}, f1 = {0, 0, 0, 0};
    const double x = 1.0 + 1e-6 * lane;
    const float xf = (float)x;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a1, 0, 0, 0);
        } else if (KIND == 1) {
            b0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, b0, 0, 0, 0);
            b1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, b1, 0, 0, 0);
            b2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, b2, 0, 0, 0);
            b3 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, b3, 0, 0, 0);
        } else {
            f0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xf, xf, f0, 0, 0, 0);
            f1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xf, xf, f1, 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + b0 + b1 + b2 + b3 + f0[0] + f1[1];
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

// waves 0..3: f32 MFMA stream; waves 4..: FMA only (KIND 0 fp64, 1 fp32)
template <int KIND>
__global__ void spec_kernel(double* out, long long* cyc, int mTrips, int fTrips) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4 f0 = {0, 0, 0, 0}, f1 = {0, 0, 0, 0};
    double a[8]; float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = 1.0 + 1e-3 * (lane + j); g[j] = (float)a[j]; }
    const float xf = 1.0f + 1e-6f * lane;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        for (int i = 0; i < mTrips; ++i) {
            f0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xf, xf, f0, 0, 0, 0);
            f1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xf, xf, f1, 0, 0, 0);
        }
    } else {
        for (int i = 0; i < fTrips; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (KIND == 0) a[j] = __builtin_fma(a[j], 1.0000001, 1e-9);
                    else g[j] = __builtin_fmaf(g[j], 1.0000001f, 1e-9f);
                }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = f0[0] + f1[1];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j] + g[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}
static double median(std::vector<long long> v) { std::sort(v.begin(), v.end()); return (double)v[v.size() / 2]; }

int main() {
    double* out; long long* cyc;
    CHECK(hipMalloc(&out, (size_t)1024 * 256 * 8));
    CHECK(hipMalloc(&cyc, 1024 * 256 / 64 * 8));
    std::vector<long long> h(1024 * 256 / 64);
    const int iters = 2000;
#define RUN_ISSUE(KIND, per, label) do { \
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((issue_kernel<KIND>), dim3(256), dim3(256), 0, 0, out, cyc, iters); CHECK(hipDeviceSynchronize()); } \
        CHECK(hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost)); \
        std::vector<long long> v(h.begin(), h.begin() + 1024); \
        printf("%-34s %.1f cycles per instruction (one wave per SIMD)\n", label, median(v) / (iters * (double)per)); } while (0)
    RUN_ISSUE(0, 2, "v_mfma_f64_16x16x4_f64");
    RUN_ISSUE(1, 4, "v_mfma_f64_4x4x4_4b_f64");
    RUN_ISSUE(2, 2, "v_mfma_f32_16x16x4_f32");
#define RUN_SPEC(KIND, F, mTrips, fTrips, label) do { \
        const int threads = 64 * (4 + 4 * F), wpb = threads / 64; \
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((spec_kernel<KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, mTrips, fTrips); CHECK(hipDeviceSynchronize()); } \
        CHECK(hipMemcpy(h.data(), cyc, 256 * wpb * 8, hipMemcpyDeviceToHost)); \
        std::vector<long long> m, f; \
        for (int b = 0; b < 256; ++b) for (int w = 0; w < wpb; ++w) (w < 4 ? m : f).push_back(h[b * wpb + w]); \
        printf("%-10s F=%d: f32 MFMA stream %7.0f cycles (%.1f per MFMA); FMA waves %7.0f cycles for %d instructions each (alone they need ~%d)\n", \
               label, F, median(m), mTrips ? median(m) / (2.0 * mTrips) : 0.0, median(f), 32 * fTrips, 32 * fTrips * (KIND == 0 ? 9 : 5) / 1); } while (0)
    RUN_SPEC(0, 3, 8000, 500, "fp64 fma");
    RUN_SPEC(1, 3, 8000, 500, "fp32 fma");
    RUN_SPEC(0, 3, 0, 500, "fp64 fma");
    RUN_SPEC(1, 3, 0, 500, "fp32 fma");
    return 0;
}
