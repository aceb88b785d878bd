// Micro-benchmark (developer tool): what does a read of 32-byte pieces at a 128-byte stride cost on MI355X, and what do
// rocprofv3's FETCH_SIZE counters report for it?  This is the access pattern of a 16-channel chunk of an NHWC tensor with 64
// channels (conv_ring.hip / conv_mfma.hip halo DMA).  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/micro/piece_bw.hip -o /tmp/piece_bw && /tmp/piece_bw
// Patterns (each lane loads 16 B; every pattern reads the same number of USEFUL bytes):
//   0  contiguous                      1  32-B pieces, 128-B stride (one chunk of 64-channel NHWC)
//   2  64-B pieces, 128-B stride       3  all four 32-B pieces of every line, one pass per piece (4 passes over the buffer)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void read_pieces(const uint4* __restrict__ src, uint4* __restrict__ sink, size_t n_items, int piece_lanes, int stride_lanes, int piece_off) {
    // item i -> piece i / piece_lanes, lane-in-piece i % piece_lanes; address (in 16-B units) = piece * stride_lanes + off + lane
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += (size_t)gridDim.x * blockDim.x) {
        const size_t piece = i / piece_lanes, l = i % piece_lanes;
        const uint4 v = src[piece * stride_lanes + piece_off + l];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;   // never true for the zero-filled buffer: keeps the loads
}

// Store patterns: 0 = contiguous 16 B/lane; 1 = the direct conv epilogue: a wave-instruction writes 32 x 32-byte pieces at a
// 128-byte stride (lane pair -> pixel), four back-to-back instructions complete the 32 lines
__global__ void write_pieces(uint4* __restrict__ dst, size_t n_lines, int mode) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const uint4 v = make_uint4(lane, 1, 2, 3);
    for (size_t base = wave * 32; base < n_lines; base += nwaves * 32) {   // 32 lines (4 KiB) per wave per iteration
        uint4* blk = dst + base * 8;
        if (mode == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) blk[k * 64 + lane] = v;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) blk[(lane >> 1) * 8 + k * 2 + (lane & 1)] = v;
        }
    }
}

int main() {
    const size_t buf_bytes = 4ull << 30;                 // 4 GiB buffer, far past the 256 MiB Infinity Cache
    uint4 *src = nullptr, *sink = nullptr;
    if (hipMalloc(&src, buf_bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 0, buf_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Pat { const char* name; int piece_lanes, stride_lanes, passes; } pats[] = {
        {"contiguous 1 GiB", 1, 1, 1}, {"32-B pieces @128-B stride (1 GiB useful of 4 GiB)", 2, 8, 1},
        {"64-B pieces @128-B stride (1 GiB useful of 2 GiB)", 4, 8, 1}, {"4 passes of 32-B pieces @128 B over 1 GiB (every byte once)", 2, 8, 4}};
    for (int pi = 0; pi < 4; ++pi) {
        const Pat& P = pats[pi];
        const size_t useful = 1ull << 30;
        const size_t n_items = useful / 16 / P.passes;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            for (int ps = 0; ps < P.passes; ++ps)
                hipLaunchKernelGGL(read_pieces, dim3(256 * 16), dim3(256), 0, 0, src, sink, n_items, P.piece_lanes, P.stride_lanes, ps * P.piece_lanes);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("pattern %d  %-62s %.3f ms  %.2f TB/s useful\n", pi, P.name, ms, useful / (ms * 1e-3) / 1e12);
        }
    }
    for (int mode = 0; mode < 2; ++mode) {
        const size_t n_lines = (1ull << 30) / 128;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(write_pieces, dim3(256 * 16), dim3(256), 0, 0, src, n_lines, mode);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("store mode %d  %-58s %.3f ms  %.2f TB/s\n", mode, mode ? "32-B pieces, 4 instructions per line (direct epilogue)" : "contiguous 1 GiB", ms, (1ull << 30) / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
