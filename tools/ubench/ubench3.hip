// Proxy of the fused kernel (not part of the product): every wave alternates a VALU phase (nV fp64 FMAs) with an
// MFMA phase (nM v_mfma_f64_16x16x4_f64). One workgroup per CU holds all its waves; MODE 1 serialises the MFMA
// phases of the waves that share a SIMD with a token in LDS (one per SIMD, taken from HW_REG_HW_ID), so that
// while one wave streams MFMAs its SIMD mates are in their VALU phases.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void proxy_kernel(double* out, long long* cyc, int* simdOf, int trips, int nV8, int nM2) {
    __shared__ int token[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 4) token[threadIdx.x] = 0;
    __syncthreads();
    // HW_REG_HW_ID (id 4): bits [5:4] = SIMD id on gfx9
    const int hwid = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
    const int simd = (hwid >> 4) & 3;
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 1.0 + 1e-3 * (lane + j);
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const double x = 1.0 + 1e-6 * lane;
    int* tok = token + (MODE == 2 ? (wave & 3) : simd);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < trips; ++t) {
        for (int q = 0; q < nV8; ++q) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = __builtin_fma(a[j], 1.0000001, 1e-9);
        }
        if (MODE != 0) {
            // acquire: one lane tries, the wave follows
            for (;;) {
                int got = 0;
                if (lane == 0) got = atomicCAS(tok, 0, 1) == 0;
                got = __builtin_amdgcn_readfirstlane(got);
                if (got) break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        for (int q = 0; q < nM2; ++q) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc1, 0, 0, 0);
        }
        if (MODE != 0) {
            asm volatile("" :: "v"(acc0[0]), "v"(acc1[0]));
            if (lane == 0) __hip_atomic_store(tok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = acc0[0] + acc1[1];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) { cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0; simdOf[blockIdx.x * (blockDim.x / 64) + wave] = simd; }
}

static double median(std::vector<long long> v) { std::sort(v.begin(), v.end()); return (double)v[v.size() / 2]; }
static double vmax(const std::vector<long long>& v) { return (double)*std::max_element(v.begin(), v.end()); }

int main() {
    double* out; long long* cyc; int* simd;
    const int maxThreads = 1024 * 256;
    CHECK(hipMalloc(&out, (size_t)maxThreads * 8));
    CHECK(hipMalloc(&cyc, maxThreads / 64 * 8));
    CHECK(hipMalloc(&simd, maxThreads / 64 * 4));
    std::vector<long long> h(maxThreads / 64);
    std::vector<int> hs(maxThreads / 64);
    const int trips = 40;
#define RUN(MODE, wps, nV, nM) do { \
        const int threads = 256 * wps; \
        for (int rep = 0; rep < 2; ++rep) { \
            hipLaunchKernelGGL((proxy_kernel<MODE>), dim3(256), dim3(threads), 0, 0, out, cyc, simd, trips, (nV) / 8, (nM) / 2); \
            CHECK(hipDeviceSynchronize()); } \
        CHECK(hipMemcpy(h.data(), cyc, 256 * 4 * wps * 8, hipMemcpyDeviceToHost)); \
        CHECK(hipMemcpy(hs.data(), simd, 256 * 4 * wps * 4, hipMemcpyDeviceToHost)); \
        std::vector<long long> v(h.begin(), h.begin() + 256 * 4 * wps); \
        int cnt[4] = {0, 0, 0, 0}; for (int w = 0; w < 4 * wps; ++w) cnt[hs[w]]++; \
        printf("mode %d  waves/SIMD %d  nV %3d nM %2d : median wave %.0f cycles/trip -> %.0f per SIMD-trip (max wave %.0f)  serial %d, mfma only %d   [block 0 waves per SIMD: %d %d %d %d; wave 0..7 simd: %d %d %d %d %d %d %d %d]\n", \
               MODE, wps, nV, nM, median(v) / trips, median(v) / trips / wps, vmax(v) / trips, (nV) * 4 + (nM) * 64, (nM) * 64, cnt[0], cnt[1], cnt[2], cnt[3], \
               hs[0], hs[1], hs[2], hs[3], hs[4], hs[5], hs[6], hs[7]); } while (0)
    for (int wps : {2, 3, 4}) {
        RUN(0, wps, 320, 32);
        RUN(1, wps, 320, 32);
        RUN(2, wps, 320, 32);
        RUN(0, wps, 240, 32);
        RUN(1, wps, 240, 32);
        RUN(0, wps, 240, 16);
        RUN(1, wps, 240, 16);
        RUN(1, wps, 120, 16);
        RUN(1, wps, 480, 32);
    }
    return 0;
}
