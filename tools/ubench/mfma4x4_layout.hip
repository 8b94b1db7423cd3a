// Operand / result lane layout of v_mfma_f64_4x4x4_4b (4 blocks of D(4x4) += A(4x4) B(4x4), one double of A, B and D
// per lane), found by experiment: A = 1 in one lane, B = 1 in one lane, which lane of D becomes 1?
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma4x4_layout mfma4x4_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(int* out) {      // out[la * 64 + lb] = lane of D that is non-zero, or -1
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) out[la * 64 + lb] = m ? __ffsll((long long)m) - 1 : -1;
        }
}

int main() {
    int* d;
    hipMalloc(&d, 64 * 64 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    std::vector<int> h(64 * 64);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // print, for block 0 (lanes 0..15) and a cross-block check
    printf("rows: lane of A, columns: lane of B (0..15), entry: lane of D (-1: no contribution)\n");
    for (int la = 0; la < 16; ++la) {
        printf("A%2d:", la);
        for (int lb = 0; lb < 16; ++lb) printf(" %3d", h[la * 64 + lb]);
        printf("\n");
    }
    int cross = 0;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb)
            if ((la / 16 != lb / 16) && h[la * 64 + lb] >= 0) ++cross;
    printf("contributions across different 16-lane groups: %d\n", cross);
    printf("block 1 sample: A lane 16.. B lane 16..: ");
    for (int lb = 16; lb < 32; ++lb) printf(" %d", h[16 * 64 + lb]);
    printf("\n");
    return 0;
}
