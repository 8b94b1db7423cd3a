// What does it cost to run N workgroups that do (almost) nothing? 256-thread workgroups with 0 / 34 KB of LDS and
// a configurable number of VGPRs, each wave loading one scalar and exiting: the dispatcher's share of a kernel
// made of many short workgroups (c4: 3 125 workgroups, c3: 2 500, c5 shard: 31 250).
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench9 ubench9.hip ; run: ./ubench9
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LDSB, int VG>
__global__ __launch_bounds__(256) void tiny(const int* in, int* out) {
    __shared__ unsigned char lds[LDSB > 0 ? LDSB : 4];
    int v[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) v[i] = threadIdx.x * (i + 1);
    if (LDSB > 0) lds[threadIdx.x] = (unsigned char)threadIdx.x;
    const int x = in[0];                    // one dependent scalar load, like reading the LM state
    int s = x;
#pragma unroll
    for (int i = 0; i < VG; ++i) s ^= v[i];
    if (LDSB > 0) s += lds[(threadIdx.x + 1) & 255];
    if (s == 0x7fffffff) out[0] = s;
}

template <int LDSB, int VG>
double run(int blocks, const int* in, int* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((tiny<LDSB, VG>), dim3(blocks), dim3(256), 0, 0, in, out);
    hipDeviceSynchronize();
    const int reps = 50;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((tiny<LDSB, VG>), dim3(blocks), dim3(256), 0, 0, in, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / reps;
}

int main() {
    int *in, *out;
    hipMalloc(&in, 4); hipMalloc(&out, 4);
    hipMemset(in, 0, 4);
    for (int blocks : {256, 1024, 2500, 3125, 12500, 31250}) {
        printf("%6d workgroups of 256: no LDS, 8 VGPRs %7.2f us | 34 KB LDS, 8 VGPRs %7.2f us | 34 KB LDS, ~100 VGPRs %7.2f us\n", blocks,
               run<0, 4>(blocks, in, out), run<34816, 4>(blocks, in, out), run<34816, 96>(blocks, in, out));
    }
    return 0;
}
