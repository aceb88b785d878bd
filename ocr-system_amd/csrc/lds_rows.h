// LDS row tiles filled from byte-addressed images with aligned 4-byte loads (shared by resize.hip and jpeg.hip).
#pragma once
#include "common.h"

// ---- LDS row tiles with arbitrary byte alignment -------------------------------------------------------------------------
// Rows of W*C bytes are not 4-byte aligned in general.  fill_rows() copies `nrows` row segments (row r starts at flat byte
// index g0 + r*row_stride) into LDS with ALIGNED 4-byte global loads: LDS row r (pitch words) holds the aligned words that
// cover the segment, so the segment's byte 0 sits at LDS byte offset m_r = (address of the segment start) & 3 of its row.
// All loads of a batch of 8 items per thread are issued before the first LDS write (one memory latency per batch, not one
// per row).  f() transforms a loaded word (identity, or the contrast blend).
template <typename F>
__device__ __forceinline__ void fill_rows(uint32_t* lds, int pitch, int nw, const uint8_t* img, long long g0, long long row_stride, int nrows,
                                          int rlo, int rhi, long long total, F f) {
    const uintptr_t ibase = reinterpret_cast<uintptr_t>(img);
    const int items = nrows * nw;
    for (int i0 = 0; i0 < items; i0 += (int)blockDim.x * 8) {
        uint32_t w[8];
        int dst[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * (int)blockDim.x + (int)threadIdx.x;
            dst[u] = -1;
            w[u] = 0;
            if (i < items) {
                const int r = i / nw, k = i - r * nw;
                if (r >= rlo && r < rhi) {
                    const long long g = g0 + (long long)r * row_stride;
                    const int m = (int)((ibase + (unsigned long long)g) & 3);
                    const long long a = g - m + 4ll * k;
                    dst[u] = r * pitch + k;
                    if (a >= 0 && a + 4 <= total) w[u] = *reinterpret_cast<const uint32_t*>(img + a);
                    else
                        for (int b = 0; b < 4; ++b)
                            if (a + b >= 0 && a + b < total) w[u] |= (uint32_t)img[a + b] << (8 * b);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (dst[u] >= 0) lds[dst[u]] = f(w[u]);
    }
}
struct WordIdentity { __device__ __forceinline__ uint32_t operator()(uint32_t w) const { return w; } };

