// Does the 4x4x4 form of the fp64 MFMA leave issue slots for VALU work on the same SIMD? ubench7 showed that
// v_mfma_f64_16x16x4 does not (half the waves MFMA, half FMA = the sum of their times). Here v_mfma_f64_4x4x4_4b
// (what the fused kernels contract with) beside fp64 FMA, fp32 FMA and integer VALU streams -- in OTHER waves of the
// SIMD, and interleaved in the SAME wave.
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench10 ubench10.hip ; run: ./ubench10
#include <hip/hip_runtime.h>
#include <cstdio>

// kind: 0 = MFMA only (all waves), 1 = VALU only (all waves), 2 = every other wave MFMA, the rest VALU,
//       3 = every wave: 5 MFMA then 20 VALU, repeated (same instruction totals per wave pair as kind 2)
// valu: 0 = v_fma_f64, 1 = v_fma_f32, 2 = v_add_u32 / v_xor
template <int VALU>
__device__ __forceinline__ void valu20(double (&c)[8], float (&f)[8], unsigned (&u)[8]) {
#pragma unroll
    for (int r = 0; r < 20; ++r) {
        const int j = r & 7;
        if (VALU == 0) c[j] = __builtin_fma(c[j], 0.999999999999, 1e-13);
        else if (VALU == 1) f[j] = __builtin_fmaf(f[j], 0.9999999f, 1e-7f);
        else u[j] = (u[j] + 0x9e3779b9u) ^ (u[j] >> 3);
    }
}

template <int VALU>
__global__ __launch_bounds__(256) void stream(int iters, int kind, double* out) {
    const int wave = threadIdx.x >> 6;
    const bool doMfma = kind == 0 || kind == 3 || (kind == 2 && ((blockIdx.x + wave) & 1) == 0);
    const bool doValu = kind == 1 || kind == 3 || (kind == 2 && !doMfma);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
    const double x = 1.0 + 1e-9 * threadIdx.x;
    double c[8]; float f[8]; unsigned u[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { c[j] = 1.0 + j + 1e-9 * threadIdx.x; f[j] = 1.0f + j; u[j] = threadIdx.x * 7 + j; }
    const int n = kind == 3 ? iters : 2 * iters;        // kinds 0-2: a wave does one thing, twice as often
    for (int i = 0; i < n; ++i) {
        if (doMfma) {
            a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, a3, 0, 0, 0);
            a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, a4, 0, 0, 0);
        }
        if (doValu) valu20<VALU>(c, f, u);
    }
    double s = a0 + a1 + a2 + a3 + a4;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += c[j] + f[j] + u[j];
    if (s == 12345.678) out[0] = s;
}

template <int VALU>
void run(const char* vname, int cus, double* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"MFMA only", "VALU only", "half the waves MFMA, half VALU", "every wave 5 MFMA + 20 VALU"};
    const int iters = 20000;
    for (int kind = 0; kind < 4; ++kind) {
        const int blocks = cus * 4;                        // 4 waves per SIMD
        hipLaunchKernelGGL(stream<VALU>, dim3(blocks), dim3(256), 0, 0, iters, kind, out);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(stream<VALU>, dim3(blocks), dim3(256), 0, 0, iters, kind, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // instructions issued per SIMD (4 waves): kinds 0, 1: 4 waves x 2 iters x (5 | 20); kinds 2, 3: half of each
        const double mfmaPerSimd = (kind == 0 ? 4.0 : (kind == 1 ? 0.0 : 2.0)) * 2 * iters * 5;
        const double valuPerSimd = (kind == 1 ? 4.0 : (kind == 0 ? 0.0 : 2.0)) * 2 * iters * 20;
        static double nsMfma = 0, nsValu = 0;              // per instruction and SIMD, from the two pure runs
        const double ns = ms * 1e6;
        if (kind == 0) nsMfma = ns / mfmaPerSimd;
        if (kind == 1) nsValu = ns / valuPerSimd;
        if (kind < 2)
            printf("CUs %3d %-8s %-34s %9.1f us   %.2f ns per %s instruction and SIMD\n", cus, vname, names[kind], ms * 1e3,
                   kind == 0 ? nsMfma : nsValu, kind == 0 ? "MFMA" : "VALU");
        else
            printf("CUs %3d %-8s %-34s %9.1f us   the same instructions one after the other: %9.1f us  (measured / serial = %.2f)\n",
                   cus, vname, names[kind], ms * 1e3, (mfmaPerSimd * nsMfma + valuPerSimd * nsValu) * 1e-3,
                   ns / (mfmaPerSimd * nsMfma + valuPerSimd * nsValu));
    }
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    double* out;
    hipMalloc(&out, 8);
    for (int cuUse : {cus, cus / 8}) {
        run<0>("fma_f64", cuUse, out);
        run<1>("fma_f32", cuUse, out);
        run<2>("int", cuUse, out);
    }
    return 0;
}
