// Proxy of the fused kernel's pipe usage (not part of the product): every wave alternates a VALU phase
// (nV independent fp64 FMAs) with an MFMA phase (nM v_mfma_f64_16x16x4_f64 on two accumulators). How well do
// the phases of the waves sharing a SIMD overlap, and which scheduling hint helps?
//   mode 0: plain   1: s_setprio(1) around the MFMA phase   2: s_setprio(1) around the VALU phase
//   mode 3: static priority from the workgroup index   4: odd workgroups start with half a VALU phase (stagger)
//   mode 5: 3 + 1
// Second part: specialised waves -- per SIMD one MFMA-only wave beside 1..3 FMA-only waves: aggregate FMA rate.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void proxy_kernel(double* out, long long* cyc, int trips, int nV8, int nM2) {
    const int lane = threadIdx.x & 63;
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 1.0 + 1e-3 * (lane + j);
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const double x = 1.0 + 1e-6 * lane;
    if (MODE == 3 || MODE == 5) {
        const int p = __builtin_amdgcn_readfirstlane((blockIdx.x >> 8) & 3);
        if (p == 1) __builtin_amdgcn_s_setprio(1);
        else if (p == 2) __builtin_amdgcn_s_setprio(2);
        else if (p == 3) __builtin_amdgcn_s_setprio(3);
    }
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 4 && (__builtin_amdgcn_readfirstlane(blockIdx.x >> 8) & 1)) {
        for (int q = 0; q < nV8 / 2; ++q) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = __builtin_fma(a[j], 1.0000001, 1e-9);
        }
    }
    for (int t = 0; t < trips; ++t) {
        if (MODE == 2) __builtin_amdgcn_s_setprio(1);
        for (int q = 0; q < nV8; ++q) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = __builtin_fma(a[j], 1.0000001, 1e-9);
        }
        if (MODE == 2) __builtin_amdgcn_s_setprio(0);
        if (MODE == 1 || MODE == 5) __builtin_amdgcn_s_setprio(MODE == 5 ? 3 : 1);
        for (int q = 0; q < nM2; ++q) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc1, 0, 0, 0);
        }
        if (MODE == 1) __builtin_amdgcn_s_setprio(0);
        if (MODE == 5) {
            const int p = __builtin_amdgcn_readfirstlane((blockIdx.x >> 8) & 3);
            if (p == 0) __builtin_amdgcn_s_setprio(0);
            else if (p == 1) __builtin_amdgcn_s_setprio(1);
            else if (p == 2) __builtin_amdgcn_s_setprio(2);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = acc0[0] + acc1[1];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

// waves 0..3 of the workgroup: MFMA only; waves 4.. : fp64 FMA only (KIND 0), fp64 mul+add pairs (1), fp32 FMA (2)
template <int KIND>
__global__ void spec_kernel(double* out, long long* cyc, int mTrips, int fTrips) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    double a[8];
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = 1.0 + 1e-3 * (lane + j); f[j] = 1.0f + 1e-3f * (lane + j); }
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
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (KIND == 0) a[j] = __builtin_fma(a[j], 1.0000001, 1e-9);
                    else if (KIND == 1) a[j] = (r & 1) ? a[j] * 1.0000001 : a[j] + 1e-9;
                    else f[j] = __builtin_fmaf(f[j], 1.0000001f, 1e-9f);
                }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = acc0[0] + acc1[1];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j] + f[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

static double median(std::vector<long long> v) { std::sort(v.begin(), v.end()); return (double)v[v.size() / 2]; }
static double vmax(const std::vector<long long>& v) { return (double)*std::max_element(v.begin(), v.end()); }

int main() {
    double* out; long long* cyc;
    const int maxThreads = 1024 * 1024;
    CHECK(hipMalloc(&out, (size_t)maxThreads * 8));
    CHECK(hipMalloc(&cyc, maxThreads / 64 * 8));
    std::vector<long long> h(maxThreads / 64);
    const int trips = 40;
    printf("== proxy: every wave alternates nV fp64 FMAs and nM f64 MFMAs; cycles per trip per SIMD = wave time / trips / (waves per SIMD)\n");
    printf("   ideal serial = nV*4 + nM*64 per wave-trip; ideal overlapped = max(nV*4, nM*64)\n");
#define RUN_PROXY(MODE, wps, nV, nM) do { \
        const int blocks = 256 * wps; \
        for (int rep = 0; rep < 2; ++rep) { \
            hipLaunchKernelGGL((proxy_kernel<MODE>), dim3(blocks), dim3(256), 0, 0, out, cyc, trips, (nV) / 8, (nM) / 2); \
            CHECK(hipDeviceSynchronize()); } \
        CHECK(hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost)); \
        std::vector<long long> v(h.begin(), h.begin() + blocks * 4); \
        printf("mode %d  waves/SIMD %d  nV %3d nM %2d : median wave %.0f cycles/trip -> %.0f per SIMD-trip (max wave %.0f)   serial %d, overlapped %d\n", \
               MODE, wps, nV, nM, median(v) / trips, median(v) / trips / wps, vmax(v) / trips, (nV) * 4 + (nM) * 64, std::max((nV) * 4, (nM) * 64)); } while (0)
    for (int wps : {1, 2, 3, 4}) {
        RUN_PROXY(0, wps, 320, 32);
        RUN_PROXY(1, wps, 320, 32);
        RUN_PROXY(2, wps, 320, 32);
        RUN_PROXY(3, wps, 320, 32);
        RUN_PROXY(4, wps, 320, 32);
        RUN_PROXY(5, wps, 320, 32);
    }
    for (int wps : {3, 4}) {
        RUN_PROXY(0, wps, 160, 16);
        RUN_PROXY(1, wps, 160, 16);
        RUN_PROXY(0, wps, 240, 32);
        RUN_PROXY(1, wps, 240, 32);
        RUN_PROXY(0, wps, 512, 32);
        RUN_PROXY(1, wps, 512, 32);
    }
    printf("== specialised waves: per SIMD one MFMA-only wave beside F FMA-only waves (workgroup of 4 + 4F waves, 1 per CU)\n");
#define RUN_SPEC(KIND, F, label) do { \
        const int threads = 64 * (4 + 4 * F); \
        const int mTrips = 4000, fTrips = 1500; \
        for (int rep = 0; rep < 2; ++rep) { \
            hipLaunchKernelGGL((spec_kernel<KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, mTrips, fTrips); \
            CHECK(hipDeviceSynchronize()); } \
        const int wpb = threads / 64; \
        CHECK(hipMemcpy(h.data(), cyc, 256 * wpb * 8, hipMemcpyDeviceToHost)); \
        std::vector<long long> m, f; \
        for (int b = 0; b < 256; ++b) for (int w = 0; w < wpb; ++w) (w < 4 ? m : f).push_back(h[b * wpb + w]); \
        const double mc = median(m) / (2.0 * mTrips), fc = median(f) / (32.0 * fTrips); \
        printf("%-18s F=%d: MFMA wave %.1f cycles per MFMA (stream lasts %.0f cycles); each FMA wave %.2f cycles per instruction -> %.2f cycles per instruction per SIMD (FMA waves last %.0f cycles)\n", \
               label, F, mc, median(m), fc, fc / F, median(f)); } while (0)
    RUN_SPEC(0, 1, "fp64 fma");
    RUN_SPEC(0, 2, "fp64 fma");
    RUN_SPEC(0, 3, "fp64 fma");
    RUN_SPEC(1, 2, "fp64 mul/add");
    RUN_SPEC(2, 2, "fp32 fma");
    RUN_SPEC(2, 3, "fp32 fma");
    return 0;
}
