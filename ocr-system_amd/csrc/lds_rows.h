// LDS row tiles filled from byte-addressed images with aligned 4-byte loads (shared by resize.hip and jpeg.hip).
#pragma once
#include "common.h"

// ---- LDS row tiles with arbitrary byte alignment -------------------------------------------------------------------------
// Rows of W*C bytes are not 4-byte aligned in general.  fill_rows() copies `nrows` row segments (row r starts at flat byte
// index g0 + r*row_stride) into LDS with ALIGNED 4-byte global loads: LDS row r (pitch words) holds the aligned words that
// cover the segment, so the segment's byte 0 sits at LDS byte offset m_r = (address of the segment start) & 3 of its row.
// All loads of a batch of 8 items per thread are issued before the first LDS write (one memory latency per batch, not one
// per row).  f() transforms a loaded word (identity, or the contrast blend).
// Offsets are 32-bit, relative to a workgroup-uniform 4-byte-aligned base near g0 (64-bit flat indices and a division per word
// made the fill cost more instructions than the filter it feeds); item -> (row, word) by a multiply-high with a per-launch
// reciprocal; words partly outside the tensor (only its first / last word can be) are patched bytewise in a rare second pass.
template <typename F>
__device__ __forceinline__ void fill_rows(uint32_t* lds, int pitch, int nw, const uint8_t* img, long long g0, int row_stride, int nrows,
                                          int rlo, int rhi, long long total, F f) {
    const int m0 = (int)((reinterpret_cast<uintptr_t>(img) + (unsigned long long)g0) & 3);
    const long long gal = g0 - m0;                       // img + gal is 4-byte aligned (it may lie before img: never dereferenced there)
    const uint8_t* pal = img + gal;
    const long long lo64 = -gal, hi64 = total - gal;     // offsets relative to pal that are inside the tensor: [lo64, hi64)
    const int lo_lim = lo64 < -0x7fffffffll ? -0x7fffffff : (lo64 > 0x7fffffffll ? 0x7fffffff : (int)lo64);
    const int hi_lim = hi64 > 0x7fffffffll ? 0x7fffffff : (hi64 < -0x7fffffffll ? -0x7fffffff : (int)hi64);
    const int safe = lo_lim > 0 ? ((lo_lim + 3) & ~3) : 0;   // an aligned offset whose word is readable (tensors here are >= 8 bytes)
    const int items = nrows * nw;
    const unsigned magic = 0xffffffffu / (unsigned)nw + 1u;   // floor(i / nw) == umulhi(i, magic) while i * nw < 2^32
    for (int i0 = 0; i0 < items; i0 += (int)blockDim.x * 8) {
        uint32_t w[8];
        int dst[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * (int)blockDim.x + (int)threadIdx.x;
            const int r = (int)__umulhi((unsigned)i, magic), k = i - r * nw;
            const int a = ((m0 + r * row_stride) & ~3) + 4 * k;
            const bool row_ok = i < items && r >= rlo && r < rhi;
            const bool ld = row_ok && a >= lo_lim && a + 4 <= hi_lim;
            const uint32_t v = *reinterpret_cast<const uint32_t*>(pal + (ld ? a : safe));
            w[u] = ld ? v : 0u;
            dst[u] = row_ok ? r * pitch + k : -1;
        }
        if ((lo64 | hi64) & 3) {                         // the tensor's first / last word is partial: bytewise, once per tensor edge
#pragma unroll 1
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * (int)blockDim.x + (int)threadIdx.x;
                const int r = (int)__umulhi((unsigned)i, magic), k = i - r * nw;
                const int a = ((m0 + r * row_stride) & ~3) + 4 * k;
                if (i < items && r >= rlo && r < rhi && a + 4 > lo_lim && a < hi_lim && (a < lo_lim || a + 4 > hi_lim)) {
                    uint32_t v = 0;
                    for (int b = 0; b < 4; ++b)
                        if (a + b >= lo_lim && a + b < hi_lim) v |= (uint32_t)pal[a + b] << (8 * b);
#pragma unroll
                    for (int uu = 0; uu < 8; ++uu) if (uu == u) w[uu] = v;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (dst[u] >= 0) lds[dst[u]] = f(w[u]);
    }
}
struct WordIdentity { __device__ __forceinline__ uint32_t operator()(uint32_t w) const { return w; } };

