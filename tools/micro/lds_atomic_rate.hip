// Micro-benchmark (developer tool): throughput of ds_add_u32 (no return) per CU for three address patterns:
//   0: lane-distinct consecutive bins (conflict-free), 1: pseudo-random bins in a 4901-bin histogram, 2: all lanes one bin.
// hipcc --offload-arch=gfx950 -O3 tools/micro/lds_atomic_rate.hip -o /tmp/lds_atomic_rate && /tmp/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters) {
    __shared__ unsigned h[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) h[i] = 0;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            unsigned b;
            if (MODE == 0) b = (threadIdx.x + 64 * (it * 8 + u)) & 4095;
            else if (MODE == 1) { s = s * 1664525u + 1013904223u; b = (s >> 8) % 4901u; }
            else b = (it * 8 + u) & 4095;
            atomicAdd(&h[b], 1u);
        }
    }
    __syncthreads();
    unsigned acc = 0;
    for (int i = threadIdx.x; i < 8192; i += 256) acc += h[i];
    if (acc == 0xffffffffu) out[0] = acc;
}
template <int MODE> void run(const char* name) {
    unsigned* d; hipMalloc(&d, 4);
    const int iters = 2000, blocks = 256 * 4;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double n = (double)blocks * 256 * iters * 8;
    printf("%-28s %.3f ms  %.2f G atomics/s  = %.2f per clock per CU (2.4 GHz, 256 CUs)\n", name, ms, n / ms / 1e6, n / (ms * 1e-3) / 2.4e9 / 256);
}
int main() { run<0>("distinct consecutive bins"); run<1>("random bins of 4901"); run<2>("one bin for all lanes"); return 0; }
