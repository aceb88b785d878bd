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
//   canny_map      16x64-pixel tiles: gray + Sobel + |dx|+|dy| + non-maximum suppression from an LDS tile -> map {0 weak, 1 none, 2 strong}
//   cc_init / cc_merge / cc_mark / cc_edges   hysteresis as connected components (min-label union-find over the kept pixels,
//                  8-connected; a component is an edge when it holds a strong pixel) -> edges 0 / 255 + compacted pixel list
//   hough          one work-group per (page, 2 angles): rho histograms in LDS (ds_add), every edge pixel votes
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

// ---------------------------------------------------------------------------------------------- 1 + 2a: map
constexpr int CT_H = 16, CT_W = 64;
__global__ __launch_bounds__(256) void canny_map_kernel(const uint8_t* rgb, uint8_t* map, int H, int W, int tiles_x, int tiles_y) {
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
    for (int i = tid; i < CT_H * CT_W; i += 256) {
        const int ly = i / CT_W, lx = i % CT_W, y = y0 + ly, x = x0 + lx;
        if (y >= H || x >= W) continue;
        const int c = (ly + 1) * (CT_W + 2) + (lx + 1), P = CT_W + 2;
        const int m = smag[c];
        uint8_t v = 1;
        if (m > DK_LOW) {
            const int xs = sdx[c], ys = sdy[c];
            const int ax = abs(xs), ay = abs(ys) << 15, tg22x = ax * DK_TG22;
            bool keep;
            if (ay < tg22x) keep = m > smag[c - 1] && m >= smag[c + 1];
            else if (ay > tg22x + (ax << 16)) keep = m > smag[c - P] && m >= smag[c + P];
            else { const int s = (xs ^ ys) < 0 ? -1 : 1; keep = m > smag[c - P - s] && m > smag[c + P + s]; }
            if (keep) v = m > DK_HIGH ? 2 : 0;
        }
        map[((size_t)pg * H + y) * W + x] = v;
    }
}

// ---------------------------------------------------------------------------------------------- 2b: hysteresis = components
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
__global__ __launch_bounds__(256) void cc_init_kernel(const uint8_t* map, int* label, int* mark, size_t total, size_t per_page) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    label[i] = map[i] != 1 ? (int)(i % per_page) : -1;      // labels are page-local pixel indices
    mark[i] = 0;
}
__global__ __launch_bounds__(256) void cc_merge_kernel(const uint8_t* map, int* label, int H, int W, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total || map[i] == 1) return;
    const size_t per = (size_t)H * W;
    const size_t pg = i / per;
    const int li = (int)(i - pg * per), y = li / W, x = li - y * W;
    const uint8_t* m = map + pg * per;
    int* L = label + pg * per;
    if (x > 0 && m[li - 1] != 1) uf_union(L, li, li - 1);
    if (y > 0) {
        if (m[li - W] != 1) uf_union(L, li, li - W);
        if (x > 0 && m[li - W - 1] != 1) uf_union(L, li, li - W - 1);
        if (x + 1 < W && m[li - W + 1] != 1) uf_union(L, li, li - W + 1);
    }
}
__global__ __launch_bounds__(256) void cc_mark_kernel(const uint8_t* map, const int* label, int* mark, size_t total, size_t per_page) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total || map[i] != 2) return;
    const size_t pg = i / per_page;
    mark[pg * per_page + uf_find(label + pg * per_page, (int)(i - pg * per_page))] = 1;   // every writer stores the same value
}
// edges + compacted list of edge pixels per page (order irrelevant: votes are integer sums).  One work-group = 2048 consecutive
// pixels of ONE page (8 per thread), one atomic per work-group: a counter per page takes ~1.4 k adds instead of one per wave.
constexpr int CE_PX = 8, CE_BLOCK = 256 * CE_PX;
__global__ __launch_bounds__(256) void cc_edges_kernel(const uint8_t* map, const int* label, const int* mark, uint8_t* edges, int* list, int* count,
                                                        int W, int blocks_per_page, size_t per_page) {
    __shared__ int s_wave[4], s_base;
    const int pg = blockIdx.x / blocks_per_page, blk = blockIdx.x - pg * blocks_per_page;
    const size_t pbase = (size_t)pg * per_page;
    const int li0 = blk * CE_BLOCK + threadIdx.x * CE_PX;
    unsigned bits = 0;
#pragma unroll
    for (int k = 0; k < CE_PX; ++k) {
        const int li = li0 + k;
        if ((size_t)li >= per_page) break;
        bool e = false;
        if (map[pbase + li] != 1) e = mark[pbase + uf_find(label + pbase, li)] != 0;
        edges[pbase + li] = e ? 255 : 0;
        bits |= (unsigned)e << k;
    }
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
__global__ __launch_bounds__(256) void hough_kernel(const int* list, const int* count, const float* trig, int* accum, int numrho, size_t per_page) {
    extern __shared__ int sacc[];   // [2][numrho]
    const int pg = blockIdx.y, n0 = blockIdx.x * 2;
    for (int i = threadIdx.x; i < 2 * numrho; i += 256) sacc[i] = 0;
    __syncthreads();
    const float c0 = trig[2 * n0], s0 = trig[2 * n0 + 1], c1 = trig[2 * n0 + 2], s1 = trig[2 * n0 + 3];
    const int half = (numrho - 1) / 2, cnt = count[pg];
    const int* l = list + pg * per_page;
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const int v = l[i];
        const float x = (float)(v & 0xffff), y = (float)(v >> 16);
        const int r0 = (int)rintf(__fadd_rn(__fmul_rn(x, c0), __fmul_rn(y, s0))) + half;
        const int r1 = (int)rintf(__fadd_rn(__fmul_rn(x, c1), __fmul_rn(y, s1))) + half;
        atomicAdd(&sacc[r0], 1);
        atomicAdd(&sacc[numrho + r1], 1);
    }
    __syncthreads();
    int* out = accum + ((size_t)pg * DK_NANGLE + n0) * numrho;
    for (int i = threadIdx.x; i < 2 * numrho; i += 256) out[i] = sacc[i];
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
__global__ __launch_bounds__(256) void warp_kernel(const uint8_t* rgb, uint8_t* out, const double* rot, const short* wtab, int H, int W) {
    const int pg = blockIdx.z, y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const uint8_t* img = rgb + (size_t)pg * H * W * 3;
    uint8_t* o = out + ((size_t)pg * H * W + (size_t)y * W + x) * 3;
    const double s = rot[3 * pg], c = rot[3 * pg + 1];
    if ((int)rot[3 * pg + 2] != 3) {
        const uint8_t* p = img + ((size_t)y * W + x) * 3;
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
        return;
    }
    // M = getRotationMatrix2D((W / 2, H / 2), angle, 1), inverted the way warpAffine inverts it (every operation rounded once)
    const double cx = (double)(W / 2), cy = (double)(H / 2);
    double M0 = c, M1 = s, M2 = __dsub_rn(__dmul_rn(__dsub_rn(1.0, c), cx), __dmul_rn(s, cy));
    double M3 = -s, M4 = c, M5 = __dadd_rn(__dmul_rn(s, cx), __dmul_rn(__dsub_rn(1.0, c), cy));
    double D = __dsub_rn(__dmul_rn(M0, M4), __dmul_rn(M1, M3));
    D = D != 0 ? __ddiv_rn(1.0, D) : 0;
    const double A11 = __dmul_rn(M4, D), A22 = __dmul_rn(M0, D);
    M0 = A11; M1 = __dmul_rn(M1, -D); M3 = __dmul_rn(M3, -D); M4 = A22;
    const double b1 = __dsub_rn(__dmul_rn(-M0, M2), __dmul_rn(M1, M5)), b2 = __dsub_rn(__dmul_rn(-M3, M2), __dmul_rn(M4, M5));
    M2 = b1; M5 = b2;
    const long long X0 = __double2ll_rn(__dmul_rn(__dadd_rn(__dmul_rn(M1, (double)y), M2), 1024.0)) + 16;
    const long long Y0 = __double2ll_rn(__dmul_rn(__dadd_rn(__dmul_rn(M4, (double)y), M5), 1024.0)) + 16;
    const long long X = (X0 + __double2ll_rn(__dmul_rn(__dmul_rn(M0, (double)x), 1024.0))) >> 5;
    const long long Y = (Y0 + __double2ll_rn(__dmul_rn(__dmul_rn(M3, (double)x), 1024.0))) >> 5;
    long long sx = X >> 5, sy = Y >> 5;
    sx = sx > 32767 ? 32767 : (sx < -32768 ? -32768 : sx);
    sy = sy > 32767 ? 32767 : (sy < -32768 ? -32768 : sy);
    short w[16];   // the phase's 4x4 weights: 32 contiguous bytes of the table
    {
        const uint4* wp = reinterpret_cast<const uint4*>(wtab + (size_t)((Y & 31) * 32 + (X & 31)) * 16);
        const uint4 w0 = wp[0], w1 = wp[1];
        const uint32_t ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) { w[2 * k] = (short)(ww[k] & 0xffffu); w[2 * k + 1] = (short)(ww[k] >> 16); }
    }
    int sum[3] = {0, 0, 0};
    const int bx = (int)sx - 1, by = (int)sy - 1;
    if (bx >= 0 && bx + 4 <= W && by >= 0 && by + 4 <= H && ((size_t)(pg + 1) * H * W * 3 - (((size_t)pg * H + by + 3) * W + bx) * 3) >= 16) {
        // interior: the 4 pixels of a row are 12 contiguous bytes -> three (unaligned) dword loads instead of twelve byte loads
        typedef uint32_t __attribute__((aligned(1))) u32u_t;
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            const u32u_t* q = reinterpret_cast<const u32u_t*>(img + ((size_t)(by + k1) * W + bx) * 3);
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
            const int w0 = w[k1 * 4], w1 = w[k1 * 4 + 1], w2 = w[k1 * 4 + 2], w3 = w[k1 * 4 + 3];
            sum[0] += (int)(d0 & 255) * w0 + (int)(d0 >> 24) * w1 + (int)((d1 >> 16) & 255) * w2 + (int)((d2 >> 8) & 255) * w3;
            sum[1] += (int)((d0 >> 8) & 255) * w0 + (int)(d1 & 255) * w1 + (int)(d1 >> 24) * w2 + (int)((d2 >> 16) & 255) * w3;
            sum[2] += (int)((d0 >> 16) & 255) * w0 + (int)((d1 >> 8) & 255) * w1 + (int)(d2 & 255) * w2 + (int)(d2 >> 24) * w3;
        }
    } else {
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            const int yy = clampi(by + k1, 0, H - 1);
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                const int xx = clampi(bx + k2, 0, W - 1);
                const uint8_t* p = img + ((size_t)yy * W + xx) * 3;
                const int wv = w[k1 * 4 + k2];
                sum[0] += p[0] * wv; sum[1] += p[1] * wv; sum[2] += p[2] * wv;
            }
        }
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) o[ch] = (uint8_t)clampi((sum[ch] + (1 << 14)) >> 15, 0, 255);
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
    return al(px) /*map*/ + al(px) /*edges*/ + 3 * al(px * 4) /*label, mark, list*/ + al((size_t)B * DK_NANGLE * numrho * 4) /*accum*/ +
           al((size_t)B * DK_MAX_VOTES * 4) + al((size_t)B * DK_MAX_PEAKS * 4) + al((size_t)B * DK_MAX_PEAKS * DK_SEG_PER_PEAK * 16) +
           al((size_t)B * DK_MAX_PEAKS * 4) + 4 * al((size_t)B * 16);
}

hipError_t deskew_launch(const DeskewParams& p, void* workspace, hipStream_t st) {
    const int B = p.B, H = p.H, W = p.W;
    if (B <= 0 || H < 8 || W < 8 || H >= 32768 || W >= 32768) return hipErrorInvalidValue;   // (H * W >= 64: a wave spans at most two pages)
    const size_t per = (size_t)H * W, px = (size_t)B * per;
    if (px >= (1ull << 40) || per >= (1ull << 31)) return hipErrorInvalidValue;
    const int numrho = 2 * (W + H) + 1;
    if ((size_t)2 * numrho * 4 > 150 * 1024) return hipErrorInvalidValue;   // the two-angle LDS histograms (pages up to ~9500 px W + H)
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    auto take = [&](size_t bytes) { unsigned char* q = ws; ws += al(bytes); return q; };
    uint8_t* map = take(px);
    uint8_t* edges = p.edges_out ? p.edges_out : take(px);
    if (p.edges_out) take(px);
    int* label = reinterpret_cast<int*>(take(px * 4));
    int* mark = reinterpret_cast<int*>(take(px * 4));
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
    const int tx = ceil_div(W, CT_W), ty = ceil_div(H, CT_H);
    const unsigned nb = (unsigned)((px + 255) / 256);
    hipLaunchKernelGGL(canny_map_kernel, dim3((unsigned)(B * tx * ty)), dim3(256), 0, st, p.rgb, map, H, W, tx, ty);
    hipLaunchKernelGGL(cc_init_kernel, dim3(nb), dim3(256), 0, st, map, label, mark, px, per);
    hipLaunchKernelGGL(cc_merge_kernel, dim3(nb), dim3(256), 0, st, map, label, H, W, px);
    hipLaunchKernelGGL(cc_mark_kernel, dim3(nb), dim3(256), 0, st, map, label, mark, px, per);
    const int bpp = (int)((per + CE_BLOCK - 1) / CE_BLOCK);
    hipLaunchKernelGGL(cc_edges_kernel, dim3((unsigned)(B * bpp)), dim3(256), 0, st, map, label, mark, edges, list, count, W, bpp, per);
    { hipError_t e2 = locr_dyn_lds(reinterpret_cast<const void*>(hough_kernel), 150 * 1024); if (e2 != hipSuccess) return e2; }
    hipLaunchKernelGGL(hough_kernel, dim3(DK_NANGLE / 2, B), dim3(256), (size_t)2 * numrho * 4, st, list, count, p.trig, accum, numrho, per);
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
    if (p.out) hipLaunchKernelGGL(warp_kernel, dim3(ceil_div(W, 256), H, B), dim3(256), 0, st, p.rgb, p.out, p.rot, p.wtab, H, W);
    return hipGetLastError();
}

hipError_t deskew_warp_launch(const uint8_t* rgb, uint8_t* out, const double* rot, const short* wtab, int B, int H, int W, hipStream_t st) {
    hipLaunchKernelGGL(warp_kernel, dim3(ceil_div(W, 256), H, B), dim3(256), 0, st, rgb, out, rot, wtab, H, W);
    return hipGetLastError();
}
