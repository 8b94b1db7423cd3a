// Accuracy probe of point_model.hpp's fast_rcp / fast_rsqrt (hardware seed + one third-order step) over 2^20 arguments
// spread over 40 binades:   hipcc -O3 --offload-arch=gfx950 -o tools/ubench/rcp_check tools/ubench/rcp_check.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../camera-calibration_amd/csrc/point_model.hpp"

__global__ void probe(const double* x, double* o, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { o[2 * i] = calib::fast_rcp(x[i]); o[2 * i + 1] = calib::fast_rsqrt(x[i]); }
}

int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), o(2 * n);
    for (int i = 0; i < n; ++i) x[i] = std::ldexp(1.0 + (double)i / n * 0.999999, (i % 41) - 20) * (1.0 + 1e-9 * i);
    double *dx, *dout;
    if (hipMalloc(&dx, n * 8) != hipSuccess || hipMalloc(&dout, n * 16) != hipSuccess) return 1;
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    probe<<<n / 256, 256>>>(dx, dout, n);
    if (hipMemcpy(o.data(), dout, n * 16, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    double w1 = 0, w2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double r1 = 1.0L / (long double)x[i], r2 = 1.0L / sqrtl((long double)x[i]);
        const double a = (double)fabsl(((long double)o[2 * i] - r1) / r1), b = (double)fabsl(((long double)o[2 * i + 1] - r2) / r2);
        if (a > w1) w1 = a;
        if (b > w2) w2 = b;
    }
    std::printf("max relative error over %d arguments: fast_rcp %.3e, fast_rsqrt %.3e (2^-53 = 1.11e-16)\n", n, w1, w2);
    return (w1 < 3.4e-16 && w2 < 3.4e-16) ? 0 : 2;
}
