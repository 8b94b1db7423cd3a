// range check of a raw buffer store: lanes with an offset past num_records must not write anywhere
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__global__ void k(double* G, int n) {
    const unsigned long long gp = (unsigned long long)(G + 1024 + 128 * blockIdx.x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)gp), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(gp >> 32));
    double* base = (double*)(((unsigned long long)hi << 32) | lo);
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 1024, 0x00020000);
    const int lane = threadIdx.x;
    // lanes 0..31: in range (offset lane*8); lanes 32..47: just past the end (1024 + ...); 48..63: the sentinel
    const int off = lane < 32 ? lane * 8 : (lane < 48 ? 1024 + (lane - 32) * 8 : 0x7ffffff0);
    u2 d = __builtin_bit_cast(u2, 1.0 + lane);
    __builtin_amdgcn_raw_buffer_store_b64(d, rsrc, off, 0, 0);
    // the last 8 bytes straddle the end: offset 1020
    if (lane == 0) __builtin_amdgcn_raw_buffer_store_b64(d, rsrc, 1020, 0, 0);
}
int main() {
    const int blocks = 4, N = 1024 + 128 * blocks + 1024;
    double* G; hipMalloc(&G, N * 8); hipMemset(G, 0, N * 8);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, G, N);
    std::vector<double> h(N); hipMemcpy(h.data(), G, N * 8, hipMemcpyDeviceToHost);
    int bad = 0, good = 0;
    for (int b = 0; b < blocks; ++b) for (int i = 0; i < 128; ++i) {
        const double v = h[1024 + 128 * b + i];
        if (i < 32) good += v == 1.0 + i; else bad += v != 0.0;
    }
    for (int i = 0; i < 1024; ++i) bad += h[i] != 0.0;
    for (int i = 1024 + 128 * blocks; i < N; ++i) bad += h[i] != 0.0;
    printf("in-range stores landed: %d of %d; writes outside their range: %d (hipGetLastError %d)\n", good, 32 * blocks, bad, (int)hipDeviceSynchronize());
    return 0;
}
