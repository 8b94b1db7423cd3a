// Micro-benchmarks that settle design questions of the fused kernel (not part of the product):
//   1. does a wave64 fp64 VALU instruction cost less when only 32 / 16 / 8 lanes are active?
//   2. issue cost of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, alone and with fp64 / fp32 VALU
//      work of the same wave or of a co-resident wave in between.
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench ubench.hip ; run: ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// mode: active lanes (64, 32, 16, 8); 8 independent fp64 FMA chains per lane, ITER trips
template <int UN>
__global__ void fma64_kernel(double* out, long long* cyc, int active, int iters) {
    const int lane = threadIdx.x & 63;
    double a[UN];
#pragma unroll
    for (int j = 0; j < UN; ++j) a[j] = 1.0 + 1e-3 * (lane + j);
    const double b = 1.0000001, c = 1e-9;
    long long t0 = 0, t1 = 0;
    if (lane < active) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < UN; ++j) a[j] = __builtin_fma(a[j], b, c);
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < UN; ++j) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

// MFMA issue: NM fp64 MFMAs per trip on NA independent accumulators, with NV independent fp64 FMAs
// (VK = 0), fp32 FMAs (VK = 1) or v_mov/int adds (VK = 2) interleaved per trip
template <int NA, int NV, int VK, bool F32>
__global__ void mfma_kernel(double* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    d4 acc[NA];
    f4 accf[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) { acc[j] = (d4){0, 0, 0, 0}; accf[j] = (f4){0, 0, 0, 0}; }
    double v[NV > 0 ? NV : 1];
    float vf[NV > 0 ? NV : 1];
    int vi[NV > 0 ? NV : 1];
#pragma unroll
    for (int j = 0; j < (NV > 0 ? NV : 1); ++j) { v[j] = 1.0 + lane * 1e-3 + j; vf[j] = 1.0f + lane * 1e-3f + j; vi[j] = lane + j; }
    const double x = 1.0 + 1e-6 * lane;
    const float xf = 1.0f + 1e-6f * lane;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            if (F32) accf[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf, xf, accf[j], 0, 0, 0);
            else acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc[j], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV / NA; ++q) {
                const int idx = j * (NV / NA) + q;
                if (VK == 0) v[idx] = __builtin_fma(v[idx], 1.0000001, 1e-9);
                else if (VK == 1) vf[idx] = __builtin_fmaf(vf[idx], 1.0000001f, 1e-9f);
                else vi[idx] = vi[idx] * 3 + 1;
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int j = 0; j < NA; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3] + accf[j][0] + accf[j][1] + accf[j][2] + accf[j][3];
#pragma unroll
    for (int j = 0; j < (NV > 0 ? NV : 1); ++j) s += v[j] + vf[j] + vi[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

// two kinds of waves in one workgroup: even waves MFMA only, odd waves fp64 FMA only (co-execution across waves)
__global__ void mixed_waves_kernel(double* out, long long* cyc, int iters, int fmaPerTrip) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 1.0 + 1e-3 * (lane + j);
    const double x = 1.0 + 1e-6 * lane;
    long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {          // waves 0..3: one per SIMD (typically), MFMA stream
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc1, 0, 0, 0);
        }
    } else {                 // waves 4..7: fp64 FMA stream
        for (int i = 0; i < iters; ++i) {
            for (int q = 0; q < fmaPerTrip; q += 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = __builtin_fma(a[j], 1.0000001, 1e-9);
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = acc0[0] + acc1[1];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

static double median(std::vector<long long>& v) { std::sort(v.begin(), v.end()); return (double)v[v.size() / 2]; }
#include <algorithm>

int main() {
    double* out; long long* cyc;
    const int maxThreads = 256 * 1024;
    CHECK(hipMalloc(&out, maxThreads * 8));
    CHECK(hipMalloc(&cyc, maxThreads / 64 * 8));
    std::vector<long long> h(maxThreads / 64);
    const int iters = 2000;
    printf("== fp64 FMA, 8 independent chains per lane, cycles per wave-instruction\n");
    for (int wavesPerSimd : {1, 2, 4}) {
        for (int active : {64, 32, 16, 8}) {
            const int threads = 256, blocks = 256 * wavesPerSimd;
            hipLaunchKernelGGL(fma64_kernel<8>, dim3(blocks), dim3(threads), 0, 0, out, cyc, active, iters);
            CHECK(hipDeviceSynchronize());
            hipLaunchKernelGGL(fma64_kernel<8>, dim3(blocks), dim3(threads), 0, 0, out, cyc, active, iters);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost));
            std::vector<long long> v(h.begin(), h.begin() + blocks * 4);
            printf("waves/SIMD %d  active lanes %2d : %.2f cycles per fp64 FMA per wave (wave-lifetime / instr)\n", wavesPerSimd, active, median(v) / (iters * 8.0));
        }
    }
#define RUN_MFMA(NA, NV, VK, F32, wps, label) do { \
        const int blocks = 256 * wps; \
        hipLaunchKernelGGL((mfma_kernel<NA, NV, VK, F32>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters); \
        CHECK(hipDeviceSynchronize()); \
        hipLaunchKernelGGL((mfma_kernel<NA, NV, VK, F32>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters); \
        CHECK(hipDeviceSynchronize()); \
        CHECK(hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost)); \
        std::vector<long long> v(h.begin(), h.begin() + blocks * 4); \
        printf("%-64s waves/SIMD %d: %.1f cycles per trip (%d MFMA + %d VALU)\n", label, wps, median(v) / iters, NA, NV); } while (0)
    printf("== MFMA issue (per trip of NA MFMAs on NA accumulators + NV VALU)\n");
    RUN_MFMA(1, 0, 0, false, 1, "f64 mfma, 1 accumulator (dependent)");
    RUN_MFMA(2, 0, 0, false, 1, "f64 mfma, 2 accumulators");
    RUN_MFMA(4, 0, 0, false, 1, "f64 mfma, 4 accumulators");
    RUN_MFMA(2, 0, 0, false, 2, "f64 mfma, 2 accumulators");
    RUN_MFMA(2, 0, 0, false, 4, "f64 mfma, 2 accumulators");
    RUN_MFMA(2, 8, 0, false, 1, "f64 mfma x2 + 8 fp64 fma (same wave)");
    RUN_MFMA(2, 16, 0, false, 1, "f64 mfma x2 + 16 fp64 fma (same wave)");
    RUN_MFMA(2, 32, 0, false, 1, "f64 mfma x2 + 32 fp64 fma (same wave)");
    RUN_MFMA(2, 16, 0, false, 4, "f64 mfma x2 + 16 fp64 fma (same wave)");
    RUN_MFMA(2, 16, 1, false, 1, "f64 mfma x2 + 16 fp32 fma (same wave)");
    RUN_MFMA(2, 32, 1, false, 1, "f64 mfma x2 + 32 fp32 fma (same wave)");
    RUN_MFMA(2, 16, 2, false, 1, "f64 mfma x2 + 16 int mul-add (same wave)");
    RUN_MFMA(2, 32, 2, false, 4, "f64 mfma x2 + 32 int mul-add (same wave)");
    RUN_MFMA(1, 0, 0, true, 1, "f32 mfma 16x16x4, 1 accumulator");
    RUN_MFMA(2, 0, 0, true, 1, "f32 mfma 16x16x4, 2 accumulators");
    RUN_MFMA(4, 0, 0, true, 1, "f32 mfma 16x16x4, 4 accumulators");
    RUN_MFMA(2, 0, 0, true, 4, "f32 mfma 16x16x4, 2 accumulators");
    RUN_MFMA(2, 16, 1, true, 1, "f32 mfma x2 + 16 fp32 fma (same wave)");
    RUN_MFMA(2, 16, 1, true, 4, "f32 mfma x2 + 16 fp32 fma (same wave)");
    RUN_MFMA(2, 16, 0, true, 1, "f32 mfma x2 + 16 fp64 fma (same wave)");
    printf("== MFMA waves (0-3) beside fp64-FMA waves (4-7) in one 512-thread workgroup, 1 WG per CU\n");
    for (int fpt : {0, 8, 16, 32}) {
        hipLaunchKernelGGL(mixed_waves_kernel, dim3(256), dim3(512), 0, 0, out, cyc, iters, fpt);
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(mixed_waves_kernel, dim3(256), dim3(512), 0, 0, out, cyc, iters, fpt);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), cyc, 256 * 8 * 8, hipMemcpyDeviceToHost));
        std::vector<long long> m, f;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : f).push_back(h[b * 8 + w]);
        printf("fma per trip %2d: mfma waves %.1f cycles per trip (2 MFMA), fma waves %.1f cycles per trip\n", fpt, median(m) / iters, median(f) / iters);
    }
    return 0;
}
