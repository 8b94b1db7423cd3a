// VALU throughput of F FMA-only waves per SIMD beside one streaming MFMA wave (or beside nothing), as a function
// of the instruction-level parallelism of the FMA stream (CH independent dependency chains per lane).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int CH>
__global__ void spec_kernel(double* out, long long* cyc, int mTrips, int fTrips) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    double a[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) a[j] = 1.0 + 1e-3 * (lane + j);
    const double x = 1.0 + 1e-6 * lane;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        for (int i = 0; i < mTrips; ++i) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc1, 0, 0, 0);
        }
    } else {
        for (int i = 0; i < fTrips; ++i) {
#pragma unroll
            for (int r = 0; r < 32 / CH; ++r) {
#pragma unroll
                for (int j = 0; j < CH; ++j) a[j] = __builtin_fma(a[j], 1.0000001, 1e-9);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = acc0[0] + acc1[1];
#pragma unroll
    for (int j = 0; j < CH; ++j) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}
static double median(std::vector<long long> v) { std::sort(v.begin(), v.end()); return (double)v[v.size() / 2]; }

int main() {
    double* out; long long* cyc;
    CHECK(hipMalloc(&out, (size_t)1024 * 256 * 8));
    CHECK(hipMalloc(&cyc, 1024 * 256 / 64 * 8));
    std::vector<long long> h(1024 * 256 / 64);
#define RUN(CH, F, mTrips, fTrips) do { \
        const int threads = 64 * (4 + 4 * F), wpb = threads / 64; \
        for (int rep = 0; rep < 2; ++rep) { \
            hipLaunchKernelGGL((spec_kernel<CH>), dim3(256), dim3(threads), 0, 0, out, cyc, mTrips, fTrips); \
            CHECK(hipDeviceSynchronize()); } \
        CHECK(hipMemcpy(h.data(), cyc, 256 * wpb * 8, hipMemcpyDeviceToHost)); \
        std::vector<long long> m, f; \
        for (int b = 0; b < 256; ++b) for (int w = 0; w < wpb; ++w) (w < 4 ? m : f).push_back(h[b * wpb + w]); \
        const double fc = median(f) / (32.0 * fTrips); \
        printf("chains %d  F=%d  mfma stream %7.0f cycles (%s) : FMA waves %7.0f cycles, %.2f cycles per FMA per wave, %.2f per SIMD\n", \
               CH, F, mTrips ? median(m) : 0.0, mTrips ? "beside MFMA" : "no MFMA", median(f), fc, fc / F); } while (0)
    for (int F : {1, 2, 3}) {
        RUN(8, F, 4000, 500);  RUN(4, F, 4000, 500);  RUN(2, F, 4000, 400);  RUN(1, F, 4000, 300);
        RUN(8, F, 0, 500);     RUN(4, F, 0, 500);     RUN(2, F, 0, 400);     RUN(1, F, 0, 300);
    }
    return 0;
}
