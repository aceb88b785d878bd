// De-skew on the GPU (gfx950): the reference's ImagePreprocessor.deskew
// (/root/reference/backend/utils/image_preprocessing.py:372-460; on by default: backend/config.py:85, ocr_service.py:412-417) =
// Canny(50, 150) -> line segments (Hough, threshold 100, min length 100, max gap 10) -> median folded angle -> skip below 0.5 /
// above 45 degrees -> cubic warpAffine with replicated borders.  The arithmetic is OpenCV's, which is absent offline ("parity
// unpinned"); the definition this file reproduces BIT FOR BIT is oracle/csrc/deskew_oracle.c (its header lists the five steps and
// the one documented deviation: the random-order progressive Hough walk is replaced by an order-independent definition).
// Everything is integer / correctly rounded fp32-fp64 (+ - * / sqrt, no fma), there is no host round trip: the rotation of a
// page is computed on the device from its median segment direction and consumed by the warp kernel.
//
// Stream-ordered kernels per group of pages (all HBM / atomic bound, none MFMA-shaped):
//   canny_map      16x64-pixel tiles: gray + Sobel + |dx|+|dy| + non-maximum suppression from an LDS tile -> map {0 weak, 1 none, 2 strong};
//                  hysteresis starts in the same kernel: the kept pixels of the tile are grouped into 8-connected components by a
//                  union-find in LDS (parents start as the first pixel of a run: one ballot), strong components flag their root
//   cc_border / cc_mark / cc_edges   the page-wide union-find only joins tile components across tile borders (top row, left and
//                  right column of every tile); flags move to the page-wide roots; a kept pixel is an edge when its root is flagged
//                  -> edges 0 / 255 + compacted pixel list
//   hough          one work-group per (page, 2 angles): rho histograms over the reachable bins in LDS (ds_add), every edge pixel votes
//   peak_hist / peak_cut / peak_list   local maxima >= 100; vcut = smallest vote count with <= 512 peaks above it; list (unordered)
//   segments       one wave per peak: 64 pixels of the line per step, ballot -> run detection on the 64-bit mask
//   angle          one work-group per page: fold + gcd-reduce the segment vectors, exact median by rank counting in LDS,
//                  (sin, cos, flag) in fp64
//   warp           fixed-point bicubic gather (32x32 phases, 15-bit weights), or a plain copy when the page is not rotated
#include "deskew.h"

#include <cmath>
#include <cstring>

namespace {

constexpr int DK_LOW = 50, DK_HIGH = 150, DK_TG22 = 13573, DK_NANGLE = 180, DK_THRESH = 100, DK_MINLEN = 100, DK_MAXGAP = 10;
constexpr int DK_MAX_PEAKS = 512, DK_MAX_VOTES = 8192, DK_SEG_PER_PEAK = 8;
constexpr double DK_SIN_HALF_DEG = 0.008726535498373935;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// min-label union-find (parents only ever decrease: the root of a set is its smallest index); used on LDS tiles and on pages
__device__ __forceinline__ int uf_find(const int* L, int i) {
    int p = L[i];
    while (p != i) { i = p; p = L[i]; }
    return i;
}
__device__ __forceinline__ void uf_union(int* L, int a, int b) {
    bool done = false;
    while (!done) {
        a = uf_find(L, a); b = uf_find(L, b);
        if (a < b) { const int old = atomicMin(&L[b], a); done = (old == b); b = old; }
        else if (b < a) { const int old = atomicMin(&L[a], b); done = (old == a); a = old; }
        else done = true;
    }
}

// ---------------------------------------------------------------------------------------------- 1 + 2a: map
constexpr int CT_H = 16, CT_W = 64;
__global__ __launch_bounds__(256) void canny_map_kernel(const uint8_t* rgb, uint8_t* map, int* label, uint8_t* mark, int H, int W, int tiles_x, int tiles_y) {
    __shared__ uint8_t sv[CT_H * CT_W];   // the tile's map values (1 = none, also outside the image)
    __shared__ int sl[CT_H * CT_W];       // union-find parents inside the tile
    __shared__ uint8_t sg[(CT_H + 4) * (CT_W + 4)];
    __shared__ short sdx[(CT_H + 2) * (CT_W + 2)], sdy[(CT_H + 2) * (CT_W + 2)];
    __shared__ int smag[(CT_H + 2) * (CT_W + 2)];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x % (tiles_x * tiles_y), pg = blockIdx.x / (tiles_x * tiles_y);
    const int y0 = (tile / tiles_x) * CT_H, x0 = (tile % tiles_x) * CT_W;
    const uint8_t* img = rgb + (size_t)pg * H * W * 3;
    for (int i = tid; i < (CT_H + 4) * (CT_W + 4); i += 256) {
        const int yy = clampi(y0 - 2 + i / (CT_W + 4), 0, H - 1), xx = clampi(x0 - 2 + i % (CT_W + 4), 0, W - 1);   // replicated border
        const uint8_t* p = img + ((size_t)yy * W + xx) * 3;
        sg[i] = (uint8_t)((4899 * p[0] + 9617 * p[1] + 1868 * p[2] + 8192) >> 14);
    }
    __syncthreads();
    for (int i = tid; i < (CT_H + 2) * (CT_W + 2); i += 256) {
        const int ly = i / (CT_W + 2), lx = i % (CT_W + 2);          // pixel (y0 - 1 + ly, x0 - 1 + lx): gray at sg[(ly + 1), (lx + 1)]
        const int y = y0 - 1 + ly, x = x0 - 1 + lx;
        const uint8_t* g = sg + (ly + 1) * (CT_W + 4) + (lx + 1);
        const int P = CT_W + 4;
        const int gx = ((int)g[-P + 1] + 2 * g[1] + g[P + 1]) - ((int)g[-P - 1] + 2 * g[-1] + g[P - 1]);
        const int gy = ((int)g[P - 1] + 2 * g[P] + g[P + 1]) - ((int)g[-P - 1] + 2 * g[-P] + g[-P + 1]);
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        sdx[i] = (short)gx; sdy[i] = (short)gy;
        smag[i] = in ? abs(gx) + abs(gy) : 0;                        // the magnitude is 0 outside the image
    }
    __syncthreads();
    // Hysteresis starts here: the kept pixels (weak or strong) of the tile are grouped into 8-connected components in LDS, so that
    // the page-wide union-find only has to join components across tile borders (cc_border_kernel).  A pixel's parent starts as the
    // first pixel of its run in the tile row (one ballot); it then unions only where its run first touches a run of the row above:
    //   up kept:                 union unless (left kept and up-left kept) — then the left neighbour made the same join;
    //   up not kept, up-left:    union unless left kept (its "up" is that pixel);
    //   up not kept, up-right:   union unless right kept (its "up" is that pixel).
    // The partition into components is the same as with every neighbour pair joined.
    const size_t per = (size_t)H * W;
    uint8_t vv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // one wave = one 64-pixel tile row (lane == lx)
        const int i = tid + 256 * k, ly = i / CT_W, lx = i % CT_W, y = y0 + ly, x = x0 + lx;
        const bool in = y < H && x < W;
        const int c = (ly + 1) * (CT_W + 2) + (lx + 1), P = CT_W + 2;
        const int m = smag[c];
        uint8_t v = 1;
        if (in && m > DK_LOW) {
            const int xs = sdx[c], ys = sdy[c];
            const int ax = abs(xs), ay = abs(ys) << 15, tg22x = ax * DK_TG22;
            bool keep;
            if (ay < tg22x) keep = m > smag[c - 1] && m >= smag[c + 1];
            else if (ay > tg22x + (ax << 16)) keep = m > smag[c - P] && m >= smag[c + P];
            else { const int s = (xs ^ ys) < 0 ? -1 : 1; keep = m > smag[c - P - s] && m > smag[c + P + s]; }
            if (keep) v = m > DK_HIGH ? 2 : 0;
        }
        const bool kept = v != 1;
        const unsigned long long km = __ballot(kept);
        if (in) map[pg * per + (size_t)y * W + x] = v;
        const unsigned long long z = ~km & ((1ull << lx) - 1ull);
        sv[i] = v; vv[k] = v;
        sl[i] = kept ? ly * CT_W + (z ? 64 - __clzll((long long)z) : 0) : -1;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = tid + 256 * k, ly = i / CT_W, lx = i % CT_W;
        if (vv[k] == 1 || ly == 0) continue;
        const bool up = sv[i - CT_W] != 1, lf = lx > 0 && sv[i - 1] != 1;
        if (up) { if (!(lf && sv[i - CT_W - 1] != 1)) uf_union(sl, i, i - CT_W); }
        else {
            if (lx > 0 && !lf && sv[i - CT_W - 1] != 1) uf_union(sl, i, i - CT_W - 1);
            if (lx + 1 < CT_W && sv[i - CT_W + 1] != 1 && sv[i + 1] == 1) uf_union(sl, i, i - CT_W + 1);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // page-wide parent = the tile component's root (a page-local pixel index); strong components flag their root
        const int i = tid + 256 * k;
        if (vv[k] == 1) continue;
        const int root = uf_find(sl, i), gr = (y0 + root / CT_W) * W + x0 + root % CT_W;
        label[pg * per + (size_t)(y0 + i / CT_W) * W + x0 + i % CT_W] = gr;
        if (vv[k] == 2) mark[pg * per + gr] = 1;
    }
}

// ---------------------------------------------------------------------------------------------- 2b: hysteresis = components
// The hysteresis kernels walk a page in work-groups of 2048 consecutive pixels, 8 per thread: the map bytes of a thread are one
// 8-byte load when the address allows, and 92 % of them say "nothing kept here".
constexpr int CE_PX = 8, CE_BLOCK = 256 * CE_PX;
constexpr unsigned long long MAP_NONE8 = 0x0101010101010101ull;
__device__ __forceinline__ unsigned long long load_map8(const uint8_t* m /* page */, int li0, size_t per_page) {
    if ((size_t)li0 + CE_PX <= per_page && (reinterpret_cast<uintptr_t>(m + li0) & 7) == 0) return *reinterpret_cast<const unsigned long long*>(m + li0);
    unsigned long long mv = MAP_NONE8;   // "none" for the pixels past the page
    for (int k = 0; k < CE_PX; ++k)
        if ((size_t)li0 + k < per_page) mv = (mv & ~(0xffull << (8 * k))) | ((unsigned long long)m[li0 + k] << (8 * k));
    return mv;
}
// Page-wide joins: only pixels on a tile's top row, left column and right column have neighbours in another tile.  One work-group
// per tile; the rules are the ones of the tile pass, restricted to the pairs that cross a tile border.
__global__ __launch_bounds__(128) void cc_border_kernel(const uint8_t* map, int* label, int H, int W, int tiles_x, int tiles_y) {
    const int t = threadIdx.x;
    if (t >= 64 + 2 * (CT_H - 1)) return;
    const int tile = blockIdx.x % (tiles_x * tiles_y), pg = blockIdx.x / (tiles_x * tiles_y);
    const int ly = t < 64 ? 0 : (t - 64) % (CT_H - 1) + 1, lx = t < 64 ? t : (t < 64 + CT_H - 1 ? 0 : CT_W - 1);
    const int y = (tile / tiles_x) * CT_H + ly, x = (tile % tiles_x) * CT_W + lx;
    if (y >= H || x >= W) return;
    const size_t per = (size_t)H * W;
    const uint8_t* m = map + pg * per;
    int* L = label + pg * per;
    const int li = y * W + x;
    if (m[li] == 1) return;
    const bool lf = x > 0 && m[li - 1] != 1;
    if (lx == 0 && lf) uf_union(L, li, li - 1);
    if (y == 0) return;
    const bool up = m[li - W] != 1;
    if (ly == 0) {
        if (up) { if (!(lf && m[li - W - 1] != 1)) uf_union(L, li, li - W); }
        else {
            if (x > 0 && !lf && m[li - W - 1] != 1) uf_union(L, li, li - W - 1);
            if (x + 1 < W && m[li - W + 1] != 1 && m[li + 1] == 1) uf_union(L, li, li - W + 1);
        }
    } else if (!up) {
        if (lx == 0) { if (x > 0 && !lf && m[li - W - 1] != 1) uf_union(L, li, li - W - 1); }
        else if (x + 1 < W && m[li - W + 1] != 1 && m[li + 1] == 1) uf_union(L, li, li - W + 1);
    }
}
// a tile component that holds a strong pixel flagged its own root (canny_map_kernel): pass the flag on to the page-wide root
__global__ __launch_bounds__(256) void cc_mark_kernel(const int* label, uint8_t* mark, int blocks_per_page, size_t per_page) {
    const int pg = blockIdx.x / blocks_per_page, li0 = (blockIdx.x - pg * blocks_per_page) * CE_BLOCK + threadIdx.x * CE_PX;
    const size_t pbase = (size_t)pg * per_page;
    unsigned long long mv = 0;
    if ((size_t)li0 + CE_PX <= per_page && (reinterpret_cast<uintptr_t>(mark + pbase + li0) & 7) == 0) mv = *reinterpret_cast<const unsigned long long*>(mark + pbase + li0);
    else
        for (int k = 0; k < CE_PX; ++k)
            if ((size_t)li0 + k < per_page) mv |= (unsigned long long)mark[pbase + li0 + k] << (8 * k);
    if (mv == 0) return;
#pragma unroll
    for (int k = 0; k < CE_PX; ++k)
        if ((mv >> (8 * k)) & 0xff) {
            const int r = uf_find(label + pbase, li0 + k);
            if (r != li0 + k) mark[pbase + r] = 1;   // every writer stores the same value
        }
}
// edges + compacted list of edge pixels per page (order irrelevant: votes are integer sums).  One work-group = 2048 consecutive
// pixels of ONE page (8 per thread: one 8-byte load / store when the addresses allow), one atomic per work-group: a counter per
// page takes ~1.4 k adds instead of one per wave.
__global__ __launch_bounds__(256) void cc_edges_kernel(const uint8_t* map, const int* label, const uint8_t* mark, uint8_t* edges, int* list, int* count,
                                                        int W, int blocks_per_page, size_t per_page) {
    __shared__ int s_wave[4], s_base;
    const int pg = blockIdx.x / blocks_per_page, blk = blockIdx.x - pg * blocks_per_page;
    const size_t pbase = (size_t)pg * per_page;
    const int li0 = blk * CE_BLOCK + threadIdx.x * CE_PX;
    const bool vout = (size_t)li0 + CE_PX <= per_page && ((reinterpret_cast<uintptr_t>(edges + pbase + li0) & 7) == 0);
    const unsigned long long mv = load_map8(map + pbase, li0, per_page);
    unsigned bits = 0;
    if (mv != MAP_NONE8) {
        // the parents of the 8 pixels are requested together; pixels of one run share their parent (the tile component's root), so
        // the walk to the page-wide root and the flag lookup happen once per run, not once per pixel
        int par[CE_PX];
#pragma unroll
        for (int k = 0; k < CE_PX; ++k) par[k] = ((mv >> (8 * k)) & 0xff) != 1 ? label[pbase + li0 + k] : -1;
        int last = -1;
        unsigned e = 0;
#pragma unroll
        for (int k = 0; k < CE_PX; ++k)
            if (par[k] >= 0) {
                if (par[k] != last) { last = par[k]; e = mark[pbase + uf_find(label + pbase, last)] != 0; }
                bits |= e << k;
            }
    }
    if (vout) {
        unsigned long long ev = 0;
#pragma unroll
        for (int k = 0; k < CE_PX; ++k) ev |= (bits & (1u << k)) ? (0xffull << (8 * k)) : 0ull;
        *reinterpret_cast<unsigned long long*>(edges + pbase + li0) = ev;
    } else
        for (int k = 0; k < CE_PX; ++k)
            if ((size_t)li0 + k < per_page) edges[pbase + li0 + k] = (bits & (1u << k)) ? 255 : 0;
    const int mine = __popc(bits), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = mine;                                     // inclusive prefix over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        s_base = tot ? atomicAdd(&count[pg], tot) : 0;
    }
    __syncthreads();
    int pos = s_base + incl - mine;
    for (int w = 0; w < wave; ++w) pos += s_wave[w];
    int* out = list + pbase;
#pragma unroll
    for (int k = 0; k < CE_PX; ++k)
        if (bits & (1u << k)) { const int li = li0 + k; out[pos++] = ((li / W) << 16) | (li % W); }
}

// ---------------------------------------------------------------------------------------------- 3a: accumulator
// |rho| = |x cos + y sin| <= the page diagonal: the LDS histogram of an angle only spans the reachable bins (2 * rmax + 1 of the
// 2 (W + H) + 1 the accumulator has; 19.6 KB per angle for an A4 page, four work-groups per CU), the others are written as zeros.
// The kernel is bound by vector instructions (tools/micro/lds_atomic_rate.hip: ds_add sustains 7 adds per clock and CU on random
// bins and does NOT serialise lanes that hit the same bin, so runs of a text line's edge pixels need no special care), so a vote
// is kept short: the two angles of the work-group ride in the two halves of packed fp32 instructions (each half an ordinary IEEE
// operation: rho = rint(x cos + y sin) with every operation rounded once, the oracle's expression), rint() is the add of
// 1.5 * 2^23 (round-to-nearest-even leaves the integer in the low mantissa bits, |rho| < 2^22), one shift-add turns the bits into
// the byte offset of the bin.  Whole rounds of 1024 list entries run without bounds checks, four loads per thread in flight.
__global__ __launch_bounds__(256) void hough_kernel(const int* __restrict__ list, const int* __restrict__ count, const float* __restrict__ trig,
                                                     int* __restrict__ accum, int numrho, int rmax, size_t per_page) {
    extern __shared__ unsigned sacc[];   // [2][bins]
    const int pg = blockIdx.y, n0 = blockIdx.x * 2;
    const int bins = 2 * rmax + 1;
    for (int i = threadIdx.x; i < 2 * bins; i += 256) sacc[i] = 0;
    __syncthreads();
    const int half = (numrho - 1) / 2, cnt = count[pg];
    const int* l = list + pg * per_page;
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t cc = {trig[2 * n0], trig[2 * n0 + 2]}, ss = {trig[2 * n0 + 1], trig[2 * n0 + 3]}, magic = {12582912.f, 12582912.f};
    const int unbias = rmax - 0x4B400000;
    unsigned* const h1 = sacc + bins;
    auto vote2 = [&](int v) {   // v = (y << 16) | x
        const float x = (float)(v & 0xffff), y = (float)(v >> 16);
        const f32x2_t xx = {x, x}, yy = {y, y};
        const f32x2_t t = (xx * cc + yy * ss) + magic;
        atomicAdd(&sacc[__float_as_int(t.x) + unbias], 1u);
        atomicAdd(&h1[__float_as_int(t.y) + unbias], 1u);
    };
    const int whole = cnt & ~1023;
    for (int base = 0; base < whole; base += 1024) {
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = l[base + 256 * k + threadIdx.x];
#pragma unroll
        for (int k = 0; k < 4; ++k) vote2(v[k]);
    }
    for (int i = whole + threadIdx.x; i < cnt; i += 256) vote2(l[i]);
    __syncthreads();
    int* out = accum + ((size_t)pg * DK_NANGLE + n0) * numrho;
    const int lo = half - rmax;   // accumulator bin of local bin 0
    for (int i = threadIdx.x; i < 2 * numrho; i += 256) {
        const int a = i >= numrho, r = i - a * numrho - lo;
        out[i] = r >= 0 && r < bins ? (int)sacc[a * bins + r] : 0;
    }
}

// ---------------------------------------------------------------------------------------------- 3b: peaks
__device__ __forceinline__ bool is_peak(const int* acc, int n, int r, int numrho, int* votes) {
    const int v = acc[(size_t)n * numrho + r];
    *votes = v;
    if (v < DK_THRESH) return false;
    const int left = r > 0 ? acc[(size_t)n * numrho + r - 1] : 0, right = r + 1 < numrho ? acc[(size_t)n * numrho + r + 1] : 0;
    const int prev = n > 0 ? acc[(size_t)(n - 1) * numrho + r] : 0, next = n + 1 < DK_NANGLE ? acc[(size_t)(n + 1) * numrho + r] : 0;
    return v > left && v >= right && v > prev && v >= next;
}
__global__ __launch_bounds__(256) void peak_hist_kernel(const int* accum, int* hist, int numrho) {
    const int pg = blockIdx.y, n = blockIdx.x;
    const int* acc = accum + (size_t)pg * DK_NANGLE * numrho;
    for (int r = threadIdx.x; r < numrho; r += 256) {
        int v;
        if (is_peak(acc, n, r, numrho, &v)) atomicAdd(&hist[(size_t)pg * DK_MAX_VOTES + (v < DK_MAX_VOTES ? v : DK_MAX_VOTES - 1)], 1);
    }
}
// vcut = smallest v >= DK_THRESH with count(votes >= v) <= DK_MAX_PEAKS (DK_MAX_VOTES when even the top bin alone is too large)
__global__ __launch_bounds__(256) void peak_cut_kernel(const int* hist, int* vcut) {
    __shared__ int part[256];
    const int pg = blockIdx.x, t = threadIdx.x;
    const int* h = hist + (size_t)pg * DK_MAX_VOTES;
    constexpr int PER = DK_MAX_VOTES / 256;
    int s = 0;
    for (int k = 0; k < PER; ++k) s += h[t * PER + k];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        int cut = DK_MAX_VOTES, above = 0;
        bool stop = false;
        for (int c = 255; c >= 0 && !stop; --c) {
            if (c * PER + PER - 1 < DK_THRESH) break;
            if (c * PER >= DK_THRESH && above + part[c] <= DK_MAX_PEAKS) { above += part[c]; cut = c * PER; continue; }   // the whole chunk fits
            for (int v = c * PER + PER - 1; v >= c * PER && v >= DK_THRESH; --v) {
                if (above + h[v] > DK_MAX_PEAKS) break;
                above += h[v]; cut = v;
            }
            stop = true;
        }
        vcut[pg] = cut;
    }
}
__global__ __launch_bounds__(256) void peak_list_kernel(const int* accum, const int* vcut, int* peaks, int* npeaks, int numrho) {
    const int pg = blockIdx.y, n = blockIdx.x, cut = vcut[pg];
    const int* acc = accum + (size_t)pg * DK_NANGLE * numrho;
    for (int r = threadIdx.x; r < numrho; r += 256) {
        int v;
        if (is_peak(acc, n, r, numrho, &v) && (v < DK_MAX_VOTES ? v : DK_MAX_VOTES - 1) >= cut) {
            const int slot = atomicAdd(&npeaks[pg], 1);
            if (slot < DK_MAX_PEAKS) peaks[(size_t)pg * DK_MAX_PEAKS + slot] = (n << 16) | r;   // (cannot overflow: at most DK_MAX_PEAKS qualify)
        }
    }
}

// ---------------------------------------------------------------------------------------------- 3c: segments, one wave per peak
__global__ __launch_bounds__(64) void segments_kernel(const uint8_t* edges, const int* peaks, const int* npeaks, const float* trig, int* segs, int* nsegs,
                                                       int H, int W, int numrho) {
    const int pg = blockIdx.y, slot = blockIdx.x, lane = threadIdx.x;
    if (slot >= min(npeaks[pg], DK_MAX_PEAKS)) { if (lane == 0) nsegs[(size_t)pg * DK_MAX_PEAKS + slot] = 0; return; }
    const int pk = peaks[(size_t)pg * DK_MAX_PEAKS + slot], n = pk >> 16, r = pk & 0xffff;
    const double cs = (double)trig[2 * n], sn = (double)trig[2 * n + 1], rho = (double)(r - (numrho - 1) / 2);
    const bool xflag = fabs(sn) > fabs(cs);
    const int L = xflag ? W : H, lim = xflag ? H : W;
    const double major = xflag ? sn : cs, minor = xflag ? cs : sn;
    const long long c0 = __double2ll_rn(__dmul_rn(__ddiv_rn(rho, major), 65536.0)) + 32768;
    const long long step = __double2ll_rn(__dmul_rn(__ddiv_rn(-minor, major), 65536.0));
    const uint8_t* e = edges + (size_t)pg * H * W;
    int* out = segs + ((size_t)pg * DK_MAX_PEAKS + slot) * DK_SEG_PER_PEAK * 4;
    int start = -1, last = -1, gap = 0, emitted = 0;
    auto close_seg = [&]() {
        const int cs_ = (int)((c0 + (long long)start * step) >> 16), cl_ = (int)((c0 + (long long)last * step) >> 16);
        const int x_s = xflag ? start : cs_, y_s = xflag ? cs_ : start, x_l = xflag ? last : cl_, y_l = xflag ? cl_ : last;
        if ((abs(x_l - x_s) >= DK_MINLEN || abs(y_l - y_s) >= DK_MINLEN) && emitted < DK_SEG_PER_PEAK) {
            const bool first_is_start = xflag ? true : !(cs > 0);
            if (lane == 0) {
                int* o = out + 4 * emitted;
                if (first_is_start) { o[0] = x_s; o[1] = y_s; o[2] = x_l; o[3] = y_l; }
                else { o[0] = x_l; o[1] = y_l; o[2] = x_s; o[3] = y_s; }
            }
            ++emitted;
        }
        start = -1; gap = 0;
    };
    for (int t0 = 0; t0 < L; t0 += 64) {
        const int t = t0 + lane;
        bool hit = false;
        if (t < L) {
            const long long c = (c0 + (long long)t * step) >> 16;
            if (c >= 0 && c < lim) hit = e[xflag ? (size_t)c * W + t : (size_t)t * W + c] != 0;
        }
        const unsigned long long m = __ballot(hit);
        const int valid = min(64, L - t0);
        int pos = 0;
        while (pos < valid) {                       // wave-uniform: runs of the 64-bit mask
            const unsigned long long rest = m >> pos;
            if (rest & 1ull) {
                const unsigned long long inv = ~rest;
                int k = inv ? __ffsll((long long)inv) - 1 : 64;
                if (k > valid - pos) k = valid - pos;
                if (start < 0) start = t0 + pos;
                last = t0 + pos + k - 1; gap = 0; pos += k;
            } else {
                int k = rest ? __ffsll((long long)rest) - 1 : 64;
                if (k > valid - pos) k = valid - pos;
                if (start >= 0) { if (gap + k > DK_MAXGAP) close_seg(); else gap += k; }
                pos += k;
            }
        }
    }
    if (start >= 0) close_seg();
    if (lane == 0) nsegs[(size_t)pg * DK_MAX_PEAKS + slot] = emitted;
}

// ---------------------------------------------------------------------------------------------- 4: median direction -> rotation
constexpr int AG_MAX = DK_MAX_PEAKS * DK_SEG_PER_PEAK;   // 4096 vectors: 16 B each in LDS
__global__ __launch_bounds__(256) void angle_kernel(const int* segs, const int* nsegs, const int* npeaks, double* rot, int* info) {
    extern __shared__ unsigned char smem_raw[];
    double* key = reinterpret_cast<double*>(smem_raw);              // [AG_MAX]
    int2* vec = reinterpret_cast<int2*>(smem_raw + AG_MAX * 8);     // [AG_MAX]
    __shared__ int s_off[DK_MAX_PEAKS + 1];
    __shared__ int2 s_pick[2];
    const int pg = blockIdx.x, t = threadIdx.x;
    const int np = min(npeaks[pg], DK_MAX_PEAKS);
    if (t == 0) {
        int acc = 0;
        for (int p = 0; p < np; ++p) { s_off[p] = acc; acc += nsegs[(size_t)pg * DK_MAX_PEAKS + p]; }
        s_off[np] = acc;
    }
    __syncthreads();
    const int ns = s_off[np];
    for (int p = t; p < np; p += 256) {
        const int c = s_off[p + 1] - s_off[p];
        const int* sp = segs + ((size_t)pg * DK_MAX_PEAKS + p) * DK_SEG_PER_PEAK * 4;
        for (int k = 0; k < c; ++k) {
            int vx = sp[4 * k + 2] - sp[4 * k], vy = sp[4 * k + 3] - sp[4 * k + 1];
            if (vy < 0 && -vy > vx) { const int q = vx; vx = -vy; vy = q; }
            else if ((vy > 0 && vy > vx) || (vy == 0 && vx < 0)) { const int q = vx; vx = vy; vy = -q; }
            int a = abs(vx), b = abs(vy);
            while (b) { const int q = a % b; a = b; b = q; }
            vx /= a; vy /= a;
            vec[s_off[p] + k] = make_int2(vx, vy);
            key[s_off[p] + k] = vx > 0 ? __ddiv_rn((double)vy, (double)vx) : INFINITY;
        }
    }
    __syncthreads();
    if (ns == 0) {
        if (t == 0) { rot[3 * pg] = 0.0; rot[3 * pg + 1] = 1.0; rot[3 * pg + 2] = 0.0; info[2 * pg] = 0; info[2 * pg + 1] = np; }
        return;
    }
    // exact order statistics by rank counting: an element whose (strictly smaller, smaller-or-equal) counts bracket k is the k-th
    const int k1 = (ns - 1) / 2, k2 = ns / 2;
    for (int i = t; i < ns; i += 256) {
        const double ki = key[i];
        int less = 0, leq = 0;
        for (int j = 0; j < ns; ++j) { const double kj = key[j]; less += kj < ki; leq += kj <= ki; }
        if (less <= k1 && k1 < leq) s_pick[0] = vec[i];     // equal keys hold identical (gcd-reduced) vectors: any writer stores the same value
        if (less <= k2 && k2 < leq) s_pick[1] = vec[i];
    }
    __syncthreads();
    if (t == 0) {
        const int2 a = s_pick[0], b = s_pick[1];
        const double la = __dsqrt_rn(__dadd_rn(__dmul_rn((double)a.x, (double)a.x), __dmul_rn((double)a.y, (double)a.y)));
        const double lb = __dsqrt_rn(__dadd_rn(__dmul_rn((double)b.x, (double)b.x), __dmul_rn((double)b.y, (double)b.y)));
        double c = __ddiv_rn((double)a.x, la), s = __ddiv_rn((double)a.y, la);
        if (ns % 2 == 0) {
            const double cx = __dadd_rn(c, __ddiv_rn((double)b.x, lb)), sx = __dadd_rn(s, __ddiv_rn((double)b.y, lb));
            const double l = __dsqrt_rn(__dadd_rn(__dmul_rn(cx, cx), __dmul_rn(sx, sx)));
            c = __ddiv_rn(cx, l); s = __ddiv_rn(sx, l);
        }
        rot[3 * pg] = s; rot[3 * pg + 1] = c;
        rot[3 * pg + 2] = fabs(s) < DK_SIN_HALF_DEG ? 1.0 : (fabs(s) > c ? 2.0 : 3.0);
        info[2 * pg] = ns; info[2 * pg + 1] = np;
    }
}

// ---------------------------------------------------------------------------------------------- 5: warp (or copy)
// One work-group = a 32x32-pixel output tile (4 pixels per thread).  The source coordinates are sums of a monotone function of x
// and one of y, so the four tile corners bound the source footprint exactly: that rectangle (+ the 4x4 support), with the border
// replicated, is staged in LDS as one dword per pixel by coalesced loads and every tap is an LDS read — no per-tap clamping, no
// unaligned gathers from HBM.  |angle| <= 45 degrees (flag 3) keeps the footprint below 52x52.
constexpr int WP_T = 32, WP_MAX = 52;
struct WarpMat { double M0, M1, M2, M3, M4, M5; };
// M = getRotationMatrix2D((W / 2, H / 2), angle, 1), inverted the way warpAffine inverts it (every operation rounded once)
__device__ __forceinline__ WarpMat warp_matrix(double s, double c, int H, int W) {
    const double cx = (double)(W / 2), cy = (double)(H / 2);
    double M0 = c, M1 = s, M2 = __dsub_rn(__dmul_rn(__dsub_rn(1.0, c), cx), __dmul_rn(s, cy));
    double M3 = -s, M4 = c, M5 = __dadd_rn(__dmul_rn(s, cx), __dmul_rn(__dsub_rn(1.0, c), cy));
    double D = __dsub_rn(__dmul_rn(M0, M4), __dmul_rn(M1, M3));
    D = D != 0 ? __ddiv_rn(1.0, D) : 0;
    const double A11 = __dmul_rn(M4, D), A22 = __dmul_rn(M0, D);
    M0 = A11; M1 = __dmul_rn(M1, -D); M3 = __dmul_rn(M3, -D); M4 = A22;
    const double b1 = __dsub_rn(__dmul_rn(-M0, M2), __dmul_rn(M1, M5)), b2 = __dsub_rn(__dmul_rn(-M3, M2), __dmul_rn(M4, M5));
    return WarpMat{M0, M1, b1, M3, M4, b2};
}
// fixed-point source position of output pixel (x, y): X, Y in 1/32 pixel; integer part clamped like OpenCV's short maps
__device__ __forceinline__ void warp_src(const WarpMat& m, int x, int y, int* sx, int* sy, int* phase) {
    const long long X0 = __double2ll_rn(__dmul_rn(__dadd_rn(__dmul_rn(m.M1, (double)y), m.M2), 1024.0)) + 16;
    const long long Y0 = __double2ll_rn(__dmul_rn(__dadd_rn(__dmul_rn(m.M4, (double)y), m.M5), 1024.0)) + 16;
    const long long X = (X0 + __double2ll_rn(__dmul_rn(__dmul_rn(m.M0, (double)x), 1024.0))) >> 5;
    const long long Y = (Y0 + __double2ll_rn(__dmul_rn(__dmul_rn(m.M3, (double)x), 1024.0))) >> 5;
    long long ix = X >> 5, iy = Y >> 5;
    ix = ix > 32767 ? 32767 : (ix < -32768 ? -32768 : ix);
    iy = iy > 32767 ? 32767 : (iy < -32768 ? -32768 : iy);
    *sx = (int)ix; *sy = (int)iy; *phase = (int)((Y & 31) * 32 + (X & 31));
}
__global__ __launch_bounds__(256) void warp_kernel(const uint8_t* rgb, uint8_t* out, const double* rot, const short* wtab, int H, int W) {
    __shared__ uint32_t tile[WP_MAX * WP_MAX];
    const int pg = blockIdx.z, x0 = blockIdx.x * WP_T, y0 = blockIdx.y * WP_T, tid = threadIdx.x;
    const uint8_t* img = rgb + (size_t)pg * H * W * 3;
    uint8_t* oimg = out + (size_t)pg * H * W * 3;
    typedef uint32_t __attribute__((aligned(1))) u32u_t;
    if ((int)rot[3 * pg + 2] != 3) {   // not rotated: copy the tile, 12 bytes (4 pixels) per thread
        const int row = y0 + (tid >> 3), xq = x0 + (tid & 7) * 4;
        if (row >= H || xq >= W) return;
        const size_t off = ((size_t)row * W + xq) * 3;
        if (xq + 4 <= W) {
            const u32u_t* q = reinterpret_cast<const u32u_t*>(img + off);
            u32u_t* d = reinterpret_cast<u32u_t*>(oimg + off);
            const uint32_t a = q[0], b = q[1], c = q[2];
            d[0] = a; d[1] = b; d[2] = c;
        } else
            for (int k = 0; k < (W - xq) * 3; ++k) oimg[off + k] = img[off + k];
        return;
    }
    const WarpMat m = warp_matrix(rot[3 * pg], rot[3 * pg + 1], H, W);
    const int x1 = min(x0 + WP_T, W) - 1, y1 = min(y0 + WP_T, H) - 1;
    int cx[4], cy[4], ph;
    warp_src(m, x0, y0, &cx[0], &cy[0], &ph); warp_src(m, x1, y0, &cx[1], &cy[1], &ph);
    warp_src(m, x0, y1, &cx[2], &cy[2], &ph); warp_src(m, x1, y1, &cx[3], &cy[3], &ph);
    const int tx0 = min(min(cx[0], cx[1]), min(cx[2], cx[3])) - 1, ty0 = min(min(cy[0], cy[1]), min(cy[2], cy[3])) - 1;
    const int tw = max(max(cx[0], cx[1]), max(cx[2], cx[3])) + 3 - tx0, th = max(max(cy[0], cy[1]), max(cy[2], cy[3])) + 3 - ty0;
    const bool staged = tw <= WP_MAX && th <= WP_MAX;   // (always, for the rotations the estimator produces)
    if (staged) {
        for (int i = tid; i < tw * th; i += 256) {
            const int ty = i / tw, tx = i - ty * tw;
            const uint8_t* p = img + ((size_t)clampi(ty0 + ty, 0, H - 1) * W + clampi(tx0 + tx, 0, W - 1)) * 3;
            tile[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
        }
    }
    __syncthreads();
    const int x = x0 + (tid & 31);
    if (x >= W) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = y0 + (tid >> 5) + 8 * j;
        if (y >= H) break;
        int sx, sy, phase;
        warp_src(m, x, y, &sx, &sy, &phase);
        const uint4* wp = reinterpret_cast<const uint4*>(wtab + (size_t)phase * 16);   // the phase's 4x4 weights: 32 contiguous bytes
        const uint4 w0 = wp[0], w1 = wp[1];
        const uint32_t ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        int sum[3] = {0, 0, 0};
        const int bx = sx - 1, by = sy - 1;
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                const int k = k1 * 4 + k2;
                const int wv = (k & 1) ? (int)(short)(ww[k >> 1] >> 16) : (int)(short)(ww[k >> 1] & 0xffffu);
                uint32_t d;
                if (staged) d = tile[(by + k1 - ty0) * tw + (bx + k2 - tx0)];
                else {
                    const uint8_t* p = img + ((size_t)clampi(by + k1, 0, H - 1) * W + clampi(bx + k2, 0, W - 1)) * 3;
                    d = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
                }
                sum[0] += (int)(d & 255) * wv; sum[1] += (int)((d >> 8) & 255) * wv; sum[2] += (int)((d >> 16) & 255) * wv;
            }
        uint8_t* o = oimg + ((size_t)y * W + x) * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) o[ch] = (uint8_t)clampi((sum[ch] + (1 << 14)) >> 15, 0, 255);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------ host side
void deskew_trig_table(float* tab) {
    const double theta = 3.141592653589793 / 180;
    for (int n = 0; n < DK_NANGLE; ++n) { tab[2 * n] = (float)cos((double)n * theta); tab[2 * n + 1] = (float)sin((double)n * theta); }
}

// OpenCV's fixed-point bicubic table (A = -0.75): float 1-D coefficients, their products as 15-bit shorts, sums forced to 2^15
void deskew_weight_table(short* wtab) {
    float tab[32 * 4];
    const float A = -0.75f, scale = 1.f / 32;
    for (int i = 0; i < 32; ++i) {
        const float x = i * scale;
        float* c = tab + 4 * i;
        c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
        c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
        c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
        c[3] = 1.f - c[0] - c[1] - c[2];
    }
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            short* it = wtab + (size_t)(i * 32 + j) * 16;
            int isum = 0;
            for (int k1 = 0; k1 < 4; ++k1)
                for (int k2 = 0; k2 < 4; ++k2) {
                    const float v = tab[4 * i + k1] * tab[4 * j + k2];
                    long q = lrintf(v * 32768.f);
                    q = q > 32767 ? 32767 : (q < -32768 ? -32768 : q);
                    it[k1 * 4 + k2] = (short)q;
                    isum += (int)q;
                }
            if (isum != 32768) {
                const int diff = isum - 32768;
                int Mk = 5, mk = 5;
                for (int k1 = 1; k1 < 3; ++k1)
                    for (int k2 = 1; k2 < 3; ++k2) {
                        const int k = k1 * 4 + k2;
                        if (it[k] < it[mk]) mk = k;
                        else if (it[k] > it[Mk]) Mk = k;
                    }
                /* the corrected weight saturates like the others: at phase (0, 0) the centre weight is 2^15, stored as 32767 — (src * 32767 +
                   2^14) >> 15 == src for every byte, an integer-aligned pixel is copied exactly (a wrap to -32768 would negate it) */
                int fixed = diff < 0 ? it[Mk] - diff : it[mk] - diff;
                fixed = fixed > 32767 ? 32767 : (fixed < -32768 ? -32768 : fixed);
                if (diff < 0) it[Mk] = (short)fixed; else it[mk] = (short)fixed;
            }
        }
}

static size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

size_t deskew_workspace_bytes(int B, int H, int W) {
    const size_t px = (size_t)B * H * W, numrho = 2 * ((size_t)W + H) + 1;
    return al(px) /*map*/ + al(px) /*edges*/ + al(px) /*mark*/ + 2 * al(px * 4) /*label, list*/ + al((size_t)B * DK_NANGLE * numrho * 4) /*accum*/ +
           al((size_t)B * DK_MAX_VOTES * 4) + al((size_t)B * DK_MAX_PEAKS * 4) + al((size_t)B * DK_MAX_PEAKS * DK_SEG_PER_PEAK * 16) +
           al((size_t)B * DK_MAX_PEAKS * 4) + 4 * al((size_t)B * 16);
}

hipError_t deskew_launch(const DeskewParams& p, void* workspace, hipStream_t st) {
    const int B = p.B, H = p.H, W = p.W;
    if (B <= 0 || H <= 0 || W <= 0 || H >= 32768 || W >= 32768) return hipErrorInvalidValue;
    const size_t per = (size_t)H * W, px = (size_t)B * per;
    if (px >= (1ull << 40) || per >= (1ull << 31)) return hipErrorInvalidValue;
    const int numrho = 2 * (W + H) + 1;
    // |x cos + y sin| <= sqrt(x^2 + y^2) * |(cos, sin)|: the float table's vectors are within 1e-7 of unit length, + 2 covers that and the rounding
    const int rmax = (int)std::sqrt((double)W * W + (double)H * H) + 2;
    const size_t hough_lds = (size_t)2 * (2 * rmax + 1) * 4;
    if (hough_lds > 150 * 1024) return hipErrorInvalidValue;   // the two-angle LDS histograms (page diagonals up to ~9500 px)
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    auto take = [&](size_t bytes) { unsigned char* q = ws; ws += al(bytes); return q; };
    uint8_t* map = take(px);
    uint8_t* edges = p.edges_out ? p.edges_out : take(px);
    if (p.edges_out) take(px);
    uint8_t* mark = take(px);
    int* label = reinterpret_cast<int*>(take(px * 4));
    int* list = reinterpret_cast<int*>(take(px * 4));
    int* accum = reinterpret_cast<int*>(take((size_t)B * DK_NANGLE * numrho * 4));
    int* hist = reinterpret_cast<int*>(take((size_t)B * DK_MAX_VOTES * 4));
    int* peaks = reinterpret_cast<int*>(take((size_t)B * DK_MAX_PEAKS * 4));
    int* segs = reinterpret_cast<int*>(take((size_t)B * DK_MAX_PEAKS * DK_SEG_PER_PEAK * 16));
    int* nsegs = reinterpret_cast<int*>(take((size_t)B * DK_MAX_PEAKS * 4));
    int* count = reinterpret_cast<int*>(take((size_t)B * 16));
    int* npeaks = reinterpret_cast<int*>(take((size_t)B * 16));
    int* vcut = reinterpret_cast<int*>(take((size_t)B * 16));
    hipError_t e;
    if ((e = hipMemsetAsync(count, 0, (size_t)B * 4, st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(npeaks, 0, (size_t)B * 4, st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(hist, 0, (size_t)B * DK_MAX_VOTES * 4, st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(mark, 0, px, st)) != hipSuccess) return e;
    const int tx = ceil_div(W, CT_W), ty = ceil_div(H, CT_H);
    hipLaunchKernelGGL(canny_map_kernel, dim3((unsigned)(B * tx * ty)), dim3(256), 0, st, p.rgb, map, label, mark, H, W, tx, ty);
    const int bpp = (int)((per + CE_BLOCK - 1) / CE_BLOCK);
    hipLaunchKernelGGL(cc_border_kernel, dim3((unsigned)(B * tx * ty)), dim3(128), 0, st, map, label, H, W, tx, ty);
    hipLaunchKernelGGL(cc_mark_kernel, dim3((unsigned)(B * bpp)), dim3(256), 0, st, label, mark, bpp, per);
    hipLaunchKernelGGL(cc_edges_kernel, dim3((unsigned)(B * bpp)), dim3(256), 0, st, map, label, mark, edges, list, count, W, bpp, per);
    { hipError_t e2 = locr_dyn_lds(reinterpret_cast<const void*>(hough_kernel), 150 * 1024); if (e2 != hipSuccess) return e2; }
    hipLaunchKernelGGL(hough_kernel, dim3(DK_NANGLE / 2, B), dim3(256), hough_lds, st, list, count, p.trig, accum, numrho, rmax, per);
    hipLaunchKernelGGL(peak_hist_kernel, dim3(DK_NANGLE, B), dim3(256), 0, st, accum, hist, numrho);
    hipLaunchKernelGGL(peak_cut_kernel, dim3(B), dim3(256), 0, st, hist, vcut);
    hipLaunchKernelGGL(peak_list_kernel, dim3(DK_NANGLE, B), dim3(256), 0, st, accum, vcut, peaks, npeaks, numrho);
    hipLaunchKernelGGL(segments_kernel, dim3(DK_MAX_PEAKS, B), dim3(64), 0, st, edges, peaks, npeaks, p.trig, segs, nsegs, H, W, numrho);
    { hipError_t e2 = locr_dyn_lds(reinterpret_cast<const void*>(angle_kernel), AG_MAX * 16); if (e2 != hipSuccess) return e2; }
    hipLaunchKernelGGL(angle_kernel, dim3(B), dim3(256), AG_MAX * 16, st, segs, nsegs, npeaks, p.rot, p.info);
    if (p.segs_out) {
        if ((e = hipMemcpyAsync(p.segs_out, segs, (size_t)B * DK_MAX_PEAKS * DK_SEG_PER_PEAK * 16, hipMemcpyDeviceToDevice, st)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(p.nsegs_out, nsegs, (size_t)B * DK_MAX_PEAKS * 4, hipMemcpyDeviceToDevice, st)) != hipSuccess) return e;
    }
    if (p.out) hipLaunchKernelGGL(warp_kernel, dim3(ceil_div(W, WP_T), ceil_div(H, WP_T), B), dim3(256), 0, st, p.rgb, p.out, p.rot, p.wtab, H, W);
    return hipGetLastError();
}

hipError_t deskew_warp_launch(const uint8_t* rgb, uint8_t* out, const double* rot, const short* wtab, int B, int H, int W, hipStream_t st) {
    if (B <= 0 || H <= 0 || W <= 0 || ceil_div(H, WP_T) > 65535 || B > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(warp_kernel, dim3(ceil_div(W, WP_T), ceil_div(H, WP_T), B), dim3(256), 0, st, rgb, out, rot, wtab, H, W);
    return hipGetLastError();
}
