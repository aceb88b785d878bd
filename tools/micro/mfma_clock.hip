// Micro-benchmark (developer tool): what does the matrix pipe sustain on THIS device?  A bare v_mfma_f32_32x32x16_bf16 loop on random
// operands held in registers (no memory traffic), 1 or 2 waves per SIMD on every CU, for ~20 ms; reports TFLOP/s by wall clock and the
// in-kernel core clock = delta(s_memtime) / delta(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, "check (6)").  The 2.5 PFLOP/s the
// roofline fractions are quoted against assumes 2.4 GHz; under MFMA load the chip holds a lower clock.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_clock.hip -o tools/micro/mfma_clock && tools/micro/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(const unsigned* seed, float* out, unsigned long long* stamps, int iters) {
    unsigned s = seed[threadIdx.x & 255] ^ (blockIdx.x * 2654435761u);
    union { bf16x8_t v; unsigned u[4]; } a[2], b[4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) { s = s * 1664525u + 1013904223u; a[i].u[j] = (s & 0x807f807fu) | 0x3f003f00u; }   // random signs / mantissas around 0.5 .. 1
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { s = s * 1664525u + 1013904223u; b[i].u[j] = (s & 0x807f807fu) | 0x3f003f00u; }
    f32x16_t acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j].v, b[i].v, acc[i][j], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) t += acc[i][j][e];
    if (t == 123.456f) out[0] = t;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
static void run(int wgs_per_cu) {
    const int cus = 256, blocks = cus * wgs_per_cu, iters = 400000;
    unsigned h[256]; for (int i = 0; i < 256; ++i) h[i] = rand();
    unsigned* ds; float* dout; unsigned long long* dst;
    (void)hipMalloc(&ds, sizeof(h)); (void)hipMalloc(&dout, 4); (void)hipMalloc(&dst, blocks * 16);
    (void)hipMemcpy(ds, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, ds, dout, dst, 1000);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, ds, dout, dst, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> st(blocks * 2);
    (void)hipMemcpy(st.data(), dst, blocks * 16, hipMemcpyDeviceToHost);
    double clk = 0; for (int i = 0; i < blocks; ++i) clk += (double)st[2 * i] / (double)st[2 * i + 1] * 0.1;   // GHz
    const double flop = (double)blocks * 4 * iters * 8 * 2.0 * 32 * 32 * 16;
    printf("%d wave(s) per SIMD: %.1f ms, %.0f TFLOP/s by wall clock, in-kernel clock %.2f GHz (mean over work-groups)\n", wgs_per_cu, ms, flop / ms / 1e9, clk / blocks);
}
int main() { run(1); run(2); run(1); return 0; }
