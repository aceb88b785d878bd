// DB post-process and recognition crop on the GPU (gfx950), bit-identical by construction to the
// sequential definition in oracle/csrc/dbpost_oracle.c (integer geometry, exact rational area
// comparison, fixed-point score sum, correctly-rounded fp64 corner arithmetic, no fma).
//
// No reference counterpart; output shape = the 4-point quad of
// /root/reference/backend/utils/ocr_postprocessor.py:24 / flat polygon of ocr_service.py:295-301.
//
// Pipeline (all stream-ordered kernels, no host round trip).  Steps 1-7 work on the RUNS of the binarised map, one wave per page row:
//   1 rl_mask / row_scan / rl_fill   threshold -> 64-bit masks per row segment -> run list [xs, xe] of the page in raster order
//   2 rl_merge      union-find (atomicMin) over run ids: a run joins the runs of the row above it touches (8-connectivity)
//   3 rl_roots      parent = root = the component's first run (starts at its smallest linear pixel index: canonical); roots per row
//   4 row_scan / rl_assign : roots in raster order -> component id k < max_boxes, top row
//   5 rl_extent     atomicMax of the component's bottom row
//   6 seg_scan / seg_init   per page: row-extreme segments; 7 rl_extremes: per-row min/max x (atomics, one pair per run)
//   8 comp_box      one work-group per component: hull (monotone chains), rotating calipers over hull
//                   edges (lanes = edges), fixed-point score (lanes = pixels), unclip, corner order
//   9 compact       valid boxes in component order -> boxes / scores / count
#include "dbpost.h"

namespace {

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

// ---- 1-7: connected components over RUNS.  A row of the binarised map is a short list of runs (a text page: ~10 per row, 2 %
// of what a per-pixel label image holds), in raster order; everything between the threshold and the per-component row extremes
// works on that list.  All kernels are one wave per (page, row), four rows per work-group.
#define ROW_WAVE_DECODE                                                        \
    const int wrow = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63; \
    if (wrow >= rows_total) return;                                            \
    const int row = wrow % Hp, pg = wrow / Hp;

// 1a: threshold -> one 64-bit mask per 64-pixel segment of the row + the number of runs in the row; the row's probabilities are
// requested 8 segments at a time
__global__ __launch_bounds__(256) void rl_mask_kernel(const bf16_t* prob, unsigned long long* mask, int* runcnt, int Hp, int Wp, int nseg, int vh, int vw,
                                                      float thresh, int rows_total) {
    ROW_WAVE_DECODE
    const size_t base = ((size_t)pg * Hp + row) * Wp;
    unsigned long long* mrow = mask + ((size_t)pg * Hp + row) * nseg;
    int cnt = 0;
    unsigned long long carry = 0;   // bit 63 of the previous segment
    for (int xb = 0; xb < Wp; xb += 64 * 8) {
        float pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int x = xb + u * 64 + lane;
            pv[u] = bf16_to_f32(prob[base + (x < Wp ? x : Wp - 1)]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int x0 = xb + u * 64, x = x0 + lane;
            if (x0 >= Wp) break;
            const unsigned long long m = __ballot(x < vw && row < vh && pv[u] > thresh);
            if (lane == 0) mrow[x0 >> 6] = m;
            cnt += __popcll(m & ~((m << 1) | carry));
            carry = m >> 63;
        }
    }
    if (lane == 0) runcnt[(size_t)pg * (Hp + 1) + row] = cnt;
}
// exclusive scan of a page's Hp row counts (one wave, chunked): cnt[r] -> offset of row r, cnt[Hp] = total (also -> total_out)
__global__ __launch_bounds__(64) void row_scan_kernel(int* cnt, int* total_out, int Hp) {
    const int pg = blockIdx.x, lane = threadIdx.x;
    int* rc = cnt + (size_t)pg * (Hp + 1);
    int run = 0;
    for (int r0 = 0; r0 < Hp; r0 += 64) {
        const int r = r0 + lane;
        const int v = r < Hp ? rc[r] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (r < Hp) rc[r] = run + inc - v;
        run += __shfl(inc, 63);
    }
    if (lane == 0) { rc[Hp] = run; if (total_out) total_out[pg] = run; }
}
// 1b: masks -> runs [xs, xe] of the row at its offset in the page's run list; a run is its own union-find parent.  Lanes = segments:
// the j-th run start of the row pairs with the j-th run end (a run may span segments), so starts and ends are ranked separately.
__global__ __launch_bounds__(256) void rl_fill_kernel(const unsigned long long* mask, const int* runoff, unsigned short* rxs, unsigned short* rxe, int* parent,
                                                      int Hp, int nseg, size_t runcap, int rows_total) {
    ROW_WAVE_DECODE
    const unsigned long long* mrow = mask + ((size_t)pg * Hp + row) * nseg;
    const size_t rb = (size_t)pg * runcap;
    int sbase = runoff[(size_t)pg * (Hp + 1) + row], ebase = sbase;
    unsigned long long carry = 0;
    for (int s0 = 0; s0 < nseg; s0 += 64) {
        const int sg = s0 + lane;
        const unsigned long long m = sg < nseg ? mrow[sg] : 0ull;
        unsigned long long prev = (unsigned long long)__shfl_up((int)(m >> 63), 1);          // bit 63 of the segment to the left
        if (lane == 0) prev = carry;
        unsigned long long next = (unsigned long long)(__shfl_down((int)(m & 1ull), 1) & 1);  // bit 0 of the segment to the right
        if (lane == 63) next = s0 + 64 < nseg ? (mrow[s0 + 64] & 1ull) : 0ull;
        unsigned long long st = m & ~((m << 1) | prev), en = m & ~((m >> 1) | (next << 63));
        int si = __popcll(st), ei = __popcll(en);
        const int ns = si, ne = ei;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int a = __shfl_up(si, d), b = __shfl_up(ei, d);
            if (lane >= d) { si += a; ei += b; }
        }
        int sp = sbase + si - ns, ep = ebase + ei - ne;   // exclusive ranks
        while (st) { const int bit = __ffsll((long long)st) - 1; st &= st - 1; rxs[rb + sp] = (unsigned short)(sg * 64 + bit); parent[rb + sp] = sp; ++sp; }
        while (en) { const int bit = __ffsll((long long)en) - 1; en &= en - 1; rxe[rb + ep] = (unsigned short)(sg * 64 + bit); ++ep; }
        sbase += __shfl(si, 63); ebase += __shfl(ei, 63);
        carry = (unsigned long long)__shfl((int)(m >> 63), 63);
    }
}
// 2: a run joins every run of the row above that it touches (8-connectivity: [xs - 1, xe + 1] overlaps [xs', xe']).  Run ids grow in
// raster order and the union keeps the smaller root, so a component's root is its first run — the one that starts at the
// component's smallest linear pixel index, the canonical root of the per-pixel definition.
__global__ __launch_bounds__(256) void rl_merge_kernel(const int* runoff, const unsigned short* rxs, const unsigned short* rxe, int* parent, int Hp,
                                                       size_t runcap, int rows_total) {
    ROW_WAVE_DECODE
    if (row == 0) return;
    const int* ro = runoff + (size_t)pg * (Hp + 1);
    const int u0 = ro[row - 1], r0 = ro[row], r1 = ro[row + 1];
    if (u0 == r0) return;
    const size_t rb = (size_t)pg * runcap;
    int* P = parent + rb;
    for (int id = r0 + lane; id < r1; id += 64) {
        const int xs = rxs[rb + id], xe = rxe[rb + id];
        int lo = u0, hi = r0;   // first run of the row above with xe' + 1 >= xs
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)rxe[rb + mid] + 1 < xs) lo = mid + 1; else hi = mid; }
        for (int t = lo; t < r0 && (int)rxs[rb + t] <= xe + 1; ++t) uf_union(P, id, t);
    }
}
// 3 + 4a: every run learns its root; roots per row are counted
__global__ __launch_bounds__(256) void rl_roots_kernel(const int* runoff, int* parent, int* rootcnt, int Hp, size_t runcap, int rows_total) {
    ROW_WAVE_DECODE
    const int* ro = runoff + (size_t)pg * (Hp + 1);
    const int r0 = ro[row], r1 = ro[row + 1];
    int* P = parent + (size_t)pg * runcap;
    int cnt = 0;
    for (int i0 = r0; i0 < r1; i0 += 64) {   // (wave-uniform bounds: the ballot sees every lane)
        const int id = i0 + lane;
        bool root = false;
        if (id < r1) {
            const int p = P[id];
            root = p == id;
            if (!root) P[id] = uf_find(P, p);   // concurrent compressions only ever replace a parent by an ancestor
        }
        cnt += __popcll(__ballot(root));
    }
    if (lane == 0) rootcnt[(size_t)pg * (Hp + 1) + row] = cnt;
}
// 4c: roots in raster order -> component id k (or -1 beyond the cap); the root's row IS the component's top row
__global__ __launch_bounds__(256) void rl_assign_kernel(const int* runoff, const int* rootoff, const int* parent, int* cidr, int* ymin, int* ymax, int Hp,
                                                        size_t runcap, int maxc, int rows_total) {
    ROW_WAVE_DECODE
    const int* ro = runoff + (size_t)pg * (Hp + 1);
    const int r0 = ro[row], r1 = ro[row + 1];
    const size_t rb = (size_t)pg * runcap;
    int k0 = rootoff[(size_t)pg * (Hp + 1) + row];
    for (int i0 = r0; i0 < r1; i0 += 64) {
        const int id = i0 + lane;
        const bool root = id < r1 && parent[rb + id] == id;
        const unsigned long long m = __ballot(root);
        if (root) {
            const int k = k0 + __popcll(m & ((1ull << lane) - 1ull));
            cidr[rb + id] = k < maxc ? k : -1;
            if (k < maxc) { ymin[(size_t)pg * maxc + k] = row; ymax[(size_t)pg * maxc + k] = row; }
        }
        k0 += __popcll(m);
    }
}
// 5: bottom row of every component
__global__ __launch_bounds__(256) void rl_extent_kernel(const int* runoff, const int* parent, const int* cidr, int* ymax, int Hp, size_t runcap, int maxc,
                                                        int rows_total) {
    ROW_WAVE_DECODE
    const int* ro = runoff + (size_t)pg * (Hp + 1);
    const size_t rb = (size_t)pg * runcap;
    for (int id = ro[row] + lane; id < ro[row + 1]; id += 64) {
        const int k = cidr[rb + parent[rb + id]];
        if (k >= 0) atomicMax(&ymax[(size_t)pg * maxc + k], row);
    }
}
// ---- 6: per page segment offsets (one wave) + init of the segments in use ----
__global__ __launch_bounds__(64) void seg_scan_kernel(const int* ncomp, const int* ymin, const int* ymax, int* segoff, int maxc) {
    const int pg = blockIdx.x, lane = threadIdx.x;
    const int n = ncomp[pg] < maxc ? ncomp[pg] : maxc;
    int run = 0;
    for (int k0 = 0; k0 < n; k0 += 64) {
        const int k = k0 + lane;
        const int v = k < n ? (ymax[(size_t)pg * maxc + k] - ymin[(size_t)pg * maxc + k] + 1) : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (k < n) segoff[(size_t)pg * (maxc + 1) + k] = run + inc - v;
        run += __shfl(inc, 63);
    }
    if (lane == 0) segoff[(size_t)pg * (maxc + 1) + n] = run;
}
__global__ __launch_bounds__(256) void seg_init_kernel(const int* ncomp, const int* segoff, int* rowmin, int* rowmax, int maxc, size_t seg_cap) {
    const int pg = blockIdx.y;
    const int n = ncomp[pg] < maxc ? ncomp[pg] : maxc;
    const int total = segoff[(size_t)pg * (maxc + 1) + n];
    for (int g = blockIdx.x * 256 + threadIdx.x; g < total; g += gridDim.x * 256) { rowmin[(size_t)pg * seg_cap + g] = 0x7fffffff; rowmax[(size_t)pg * seg_cap + g] = -1; }
}
// ---- 7: leftmost / rightmost foreground pixel of every component in every row it covers ----
__global__ __launch_bounds__(256) void rl_extremes_kernel(const int* runoff, const unsigned short* rxs, const unsigned short* rxe, const int* parent,
                                                          const int* cidr, const int* ymin, const int* segoff, int* rowmin, int* rowmax, int Hp,
                                                          size_t runcap, int maxc, size_t seg_cap, int rows_total) {
    ROW_WAVE_DECODE
    const int* ro = runoff + (size_t)pg * (Hp + 1);
    const size_t rb = (size_t)pg * runcap;
    for (int id = ro[row] + lane; id < ro[row + 1]; id += 64) {
        const int k = cidr[rb + parent[rb + id]];
        if (k < 0) continue;
        const size_t sg = (size_t)pg * seg_cap + segoff[(size_t)pg * (maxc + 1) + k] + (row - ymin[(size_t)pg * maxc + k]);
        atomicMin(&rowmin[sg], (int)rxs[rb + id]);
        atomicMax(&rowmax[sg], (int)rxe[rb + id]);
    }
}

// ---- 8: one wave per component ----
struct Cand { long long dx, dy, mind, maxd, minn, maxn, L, A; int have; };

__device__ __forceinline__ bool frac_less(unsigned long long a1, unsigned long long l2, unsigned long long a2, unsigned long long l1, bool* eq) {
    // compare a1*l2 vs a2*l1 exactly (128-bit)
    const unsigned long long h1 = __umul64hi(a1, l2), lo1 = a1 * l2, h2 = __umul64hi(a2, l1), lo2 = a2 * l1;
    *eq = (h1 == h2 && lo1 == lo2);
    return h1 < h2 || (h1 == h2 && lo1 < lo2);
}
__device__ __forceinline__ bool cand_better(const Cand& a, const Cand& b) {  // is a strictly better than b
    if (!a.have) return false;
    if (!b.have) return true;
    bool eq;
    const bool less = frac_less((unsigned long long)a.A, (unsigned long long)b.L, (unsigned long long)b.A, (unsigned long long)a.L, &eq);
    if (!eq) return less;
    return a.dx < b.dx || (a.dx == b.dx && a.dy < b.dy);
}
__device__ __forceinline__ long long gcdll(long long a, long long b) {
    a = a < 0 ? -a : a; b = b < 0 ? -b : b;
    while (b) { const long long t = a % b; a = b; b = t; }
    return a;
}
__device__ __forceinline__ long long shfl_ll(long long v, int src) {
    const int lo = __shfl((int)(v & 0xffffffffll), src), hi = __shfl((int)(v >> 32), src);
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ long long shfl_xor_ll(long long v, int m) {
    const int lo = __shfl_xor((int)(v & 0xffffffffll), m), hi = __shfl_xor((int)(v >> 32), m);
    return ((long long)hi << 32) | (unsigned int)lo;
}

constexpr int HULL_LDS = 4096;  // hull points kept in LDS (taller components fall back to the global scratch)

__global__ __launch_bounds__(256) void comp_box_kernel(const bf16_t* prob, const int* ncomp, const int* ymin, const int* ymax, const int* segoff,
                                                       const int* rowmin, const int* rowmax, int2* hullbuf, int* box_tmp, float* score_tmp,
                                                       int* valid_tmp, int Hp, int Wp, int vh, int vw, int maxc, size_t seg_cap,
                                                       float box_thresh, float unclip_ratio, int min_size) {
    __shared__ int2 s_hull[HULL_LDS];
    __shared__ int s_ext[HULL_LDS];  // row minima [0, HULL_LDS/2) and maxima [HULL_LDS/2, HULL_LDS) of the component
    __shared__ Cand s_cand[4];
    __shared__ unsigned long long s_sum[4], s_cnt[4];
    __shared__ int s_nl, s_nr, s_nh;
    const int pg = blockIdx.x / maxc, k = blockIdx.x % maxc, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* vflag = valid_tmp + (size_t)pg * maxc + k;
    const int n = ncomp[pg] < maxc ? ncomp[pg] : maxc;
    if (k >= n) { if (tid == 0) *vflag = 0; return; }
    const int y0 = ymin[(size_t)pg * maxc + k], y1 = ymax[(size_t)pg * maxc + k];
    const size_t seg = (size_t)pg * seg_cap + segoff[(size_t)pg * (maxc + 1) + k];
    const int* rmin = rowmin + seg;
    const int* rmax = rowmax + seg;
    const int rows = y1 - y0 + 1;
    int2* hull = (2 * rows <= HULL_LDS) ? s_hull : hullbuf + 2 * seg;
    // the per-row extremes are fetched by the whole workgroup (coalesced, one latency) so that the two serial chain builders
    // below walk LDS, not global memory (one dependent global load per row otherwise)
    const bool ext_lds = rows <= HULL_LDS / 2;
    if (ext_lds)
        for (int r = tid; r < rows; r += 256) { s_ext[r] = rmin[r]; s_ext[HULL_LDS / 2 + r] = rmax[r]; }
    __syncthreads();
    // ---- hull: thread 0 builds the left chain in hull[0..], thread 64 (another wave) the right chain in hull[rows..] ----
    if (tid == 0 || tid == 64) {
        const bool left = tid == 0;
        int2* ch = hull + (left ? 0 : rows);
        int nc = 0;
        for (int t = 0; t < rows; ++t) {
            const int r = left ? t : rows - 1 - t;
            const int ex = ext_lds ? s_ext[(left ? 0 : HULL_LDS / 2) + r] : (left ? rmin[r] : rmax[r]);
            const int2 p = make_int2(ex, y0 + r);
            while (nc >= 2) {
                const int2 o = ch[nc - 2], a = ch[nc - 1];
                const long long cr = (long long)(a.x - o.x) * (p.y - o.y) - (long long)(a.y - o.y) * (p.x - o.x);
                if (cr >= 0) --nc; else break;
            }
            ch[nc++] = p;
        }
        if (left) s_nl = nc; else s_nr = nc;
    }
    __syncthreads();
    if (tid == 0) {  // splice: left chain then right chain, dropping duplicated joints
        int nh = s_nl;
        const int2* rc = hull + rows;
        for (int t = 0; t < s_nr; ++t) {
            const int2 p = rc[t];
            if (nh && hull[nh - 1].x == p.x && hull[nh - 1].y == p.y) continue;
            hull[nh++] = p;
        }
        if (nh > 1 && hull[0].x == hull[nh - 1].x && hull[0].y == hull[nh - 1].y) --nh;
        s_nh = nh;
    }
    __syncthreads();
    const int nh = s_nh;
    if (nh < 2) { if (tid == 0) *vflag = 0; return; }
    // ---- calipers: threads = edges ----
    Cand best; best.have = 0; best.dx = best.dy = best.mind = best.maxd = best.minn = best.maxn = 0; best.L = 1; best.A = 0;
    for (int e = tid; e < nh; e += 256) {
        const int2 a = hull[e], b = hull[(e + 1) % nh];
        long long dx = b.x - a.x, dy = b.y - a.y;
        if (dx == 0 && dy == 0) continue;
        const long long g = gcdll(dx, dy);
        dx /= g; dy /= g;
        if (dx < 0 || (dx == 0 && dy < 0)) { dx = -dx; dy = -dy; }
        long long mind = 0x7fffffffffffffffll, maxd = -0x7fffffffffffffffll - 1, minn = mind, maxn = maxd;
        for (int t = 0; t < nh; ++t) {
            const int2 q = hull[t];
            const long long pd = (long long)q.x * dx + (long long)q.y * dy, pn = -(long long)q.x * dy + (long long)q.y * dx;
            mind = pd < mind ? pd : mind; maxd = pd > maxd ? pd : maxd;
            minn = pn < minn ? pn : minn; maxn = pn > maxn ? pn : maxn;
        }
        Cand c; c.have = 1; c.dx = dx; c.dy = dy; c.mind = mind; c.maxd = maxd; c.minn = minn; c.maxn = maxn;
        c.L = dx * dx + dy * dy; c.A = (maxd - mind) * (maxn - minn);
        if (cand_better(c, best)) best = c;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        Cand o;
        o.have = __shfl_xor(best.have, m);
        o.dx = shfl_xor_ll(best.dx, m); o.dy = shfl_xor_ll(best.dy, m);
        o.mind = shfl_xor_ll(best.mind, m); o.maxd = shfl_xor_ll(best.maxd, m);
        o.minn = shfl_xor_ll(best.minn, m); o.maxn = shfl_xor_ll(best.maxn, m);
        o.L = shfl_xor_ll(best.L, m); o.A = shfl_xor_ll(best.A, m);
        if (cand_better(o, best)) best = o;
    }
    if (lane == 0) s_cand[wave] = best;
    __syncthreads();
    best = s_cand[0];
#pragma unroll
    for (int wv = 1; wv < 4; ++wv) { const Cand o = s_cand[wv]; if (cand_better(o, best)) best = o; }
    if (!best.have) { if (tid == 0) *vflag = 0; return; }
    const long long wd = best.maxd - best.mind, wn = best.maxn - best.minn;
    const double sqL = __dsqrt_rn((double)best.L);
    const double sside = (double)(wd < wn ? wd : wn) / sqL;
    if (sside < (double)min_size) { if (tid == 0) *vflag = 0; return; }
    // ---- score: integer pixels inside the closed rectangle (exact integer sum: order-free) ----
    double fx0, fx1, fy0, fy1;
    {
        const long long as[4] = {best.mind, best.maxd, best.maxd, best.mind}, bs[4] = {best.minn, best.minn, best.maxn, best.maxn};
        double cx[4], cy[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            cx[t] = (double)(as[t] * best.dx - bs[t] * best.dy) / (double)best.L;
            cy[t] = (double)(as[t] * best.dy + bs[t] * best.dx) / (double)best.L;
        }
        fx0 = fx1 = cx[0]; fy0 = fy1 = cy[0];
#pragma unroll
        for (int t = 1; t < 4; ++t) { fx0 = fmin(fx0, cx[t]); fx1 = fmax(fx1, cx[t]); fy0 = fmin(fy0, cy[t]); fy1 = fmax(fy1, cy[t]); }
    }
    int bx0 = (int)floor(fx0) - 1, bx1 = (int)ceil(fx1) + 1, by0 = (int)floor(fy0) - 1, by1 = (int)ceil(fy1) + 1;
    bx0 = bx0 < 0 ? 0 : bx0; by0 = by0 < 0 ? 0 : by0; bx1 = bx1 > vw - 1 ? vw - 1 : bx1; by1 = by1 > vh - 1 ? vh - 1 : by1;
    unsigned long long sum = 0, cnt = 0;
    const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
    const bf16_t* pp = prob + (size_t)pg * Hp * Wp;
    // all 256 threads sweep the bounding span; 8 probability loads per thread are requested before the first inside-test
    // (a load inside the divergent test costs one memory latency per pixel column)
    const int npx = bw * bh;
    for (int i0 = 0; i0 < npx; i0 += 256 * 8) {
        float pv[8];
        int px[8], py[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256 + tid;
            const int ic = i < npx ? i : 0;
            const int ry = ic / bw;
            py[u] = i < npx ? by0 + ry : -1;
            px[u] = bx0 + (ic - ry * bw);
            pv[u] = bf16_to_f32(pp[(size_t)(by0 + ry) * Wp + px[u]]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long pd = (long long)px[u] * best.dx + (long long)py[u] * best.dy, pn = -(long long)px[u] * best.dy + (long long)py[u] * best.dx;
            const bool in = py[u] >= 0 && pd >= best.mind && pd <= best.maxd && pn >= best.minn && pn <= best.maxn;
            sum += in ? (unsigned long long)(pv[u] * 16777216.0f) : 0ull;
            cnt += in ? 1ull : 0ull;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { sum += (unsigned long long)shfl_xor_ll((long long)sum, m); cnt += (unsigned long long)shfl_xor_ll((long long)cnt, m); }
    if (lane == 0) { s_sum[wave] = sum; s_cnt[wave] = cnt; }
    __syncthreads();
    if (tid != 0) return;
    sum = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    cnt = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (!cnt) { *vflag = 0; return; }
    const double score = ((double)sum / (double)cnt) / 16777216.0;
    if (score < (double)box_thresh) { *vflag = 0; return; }
    // ---- unclip + corners ----
    const double E = ((double)unclip_ratio * (double)(wd * wn)) / (double)(2 * (wd + wn));
    const double sside2 = ((double)(wd < wn ? wd : wn) + 2.0 * E) / sqL;
    if (sside2 < (double)(min_size + 2)) { *vflag = 0; return; }
    const double a0 = (double)best.mind - E, a1 = (double)best.maxd + E, b0 = (double)best.minn - E, b1 = (double)best.maxn + E;
    double qx[4], qy[4];
    {
        const double as[4] = {a0, a1, a1, a0}, bs[4] = {b0, b0, b1, b1};
        const double ddx = (double)best.dx, ddy = (double)best.dy, dL = (double)best.L;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double t1 = __dmul_rn(as[t], ddx), t2 = __dmul_rn(bs[t], ddy), t3 = __dmul_rn(as[t], ddy), t4 = __dmul_rn(bs[t], ddx);
            qx[t] = __ddiv_rn(__dsub_rn(t1, t2), dL);
            qy[t] = __ddiv_rn(__dadd_rn(t3, t4), dL);
        }
    }
    for (int i = 1; i < 4; ++i)
        for (int j = i; j > 0; --j) {
            const bool lt = qx[j] < qx[j - 1] || (qx[j] == qx[j - 1] && qy[j] < qy[j - 1]);
            if (!lt) break;
            const double tx = qx[j], ty = qy[j]; qx[j] = qx[j - 1]; qy[j] = qy[j - 1]; qx[j - 1] = tx; qy[j - 1] = ty;
        }
    double ox[4], oy[4];  // TL, TR, BR, BL
    if (qy[0] <= qy[1]) { ox[0] = qx[0]; oy[0] = qy[0]; ox[3] = qx[1]; oy[3] = qy[1]; } else { ox[0] = qx[1]; oy[0] = qy[1]; ox[3] = qx[0]; oy[3] = qy[0]; }
    if (qy[2] <= qy[3]) { ox[1] = qx[2]; oy[1] = qy[2]; ox[2] = qx[3]; oy[2] = qy[3]; } else { ox[1] = qx[3]; oy[1] = qy[3]; ox[2] = qx[2]; oy[2] = qy[2]; }
    int out[8];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        double rx = rint(ox[t]), ry = rint(oy[t]);
        rx = rx < 0 ? 0 : rx; rx = rx > vw ? vw : rx; ry = ry < 0 ? 0 : ry; ry = ry > vh ? vh : ry;
        out[2 * t] = (int)rx; out[2 * t + 1] = (int)ry;
    }
    {
        const double wx = (double)(out[0] - out[2]), wy = (double)(out[1] - out[3]);
        const double hx = (double)(out[0] - out[6]), hy = (double)(out[1] - out[7]);
        const int rw = (int)sqrt(__dadd_rn(__dmul_rn(wx, wx), __dmul_rn(wy, wy))), rh = (int)sqrt(__dadd_rn(__dmul_rn(hx, hx), __dmul_rn(hy, hy)));
        if (rw <= 3 || rh <= 3) { *vflag = 0; return; }
    }
    int* bo = box_tmp + ((size_t)pg * maxc + k) * 8;
#pragma unroll
    for (int t = 0; t < 8; ++t) bo[t] = out[t];
    score_tmp[(size_t)pg * maxc + k] = (float)score;
    *vflag = 1;
}

// ---- 9 ----
__global__ __launch_bounds__(64) void compact_kernel(const int* box_tmp, const float* score_tmp, const int* valid_tmp, int* boxes, float* scores,
                                                     int* counts, int maxc) {
    const int pg = blockIdx.x, lane = threadIdx.x;
    int run = 0;
    for (int k0 = 0; k0 < maxc; k0 += 64) {
        const int k = k0 + lane;
        const bool v = k < maxc && valid_tmp[(size_t)pg * maxc + k] != 0;
        const unsigned long long m = __ballot(v);
        if (v) {
            const int pos = run + __popcll(m & ((1ull << lane) - 1ull));
#pragma unroll
            for (int t = 0; t < 8; ++t) boxes[((size_t)pg * maxc + pos) * 8 + t] = box_tmp[((size_t)pg * maxc + k) * 8 + t];
            scores[(size_t)pg * maxc + pos] = score_tmp[(size_t)pg * maxc + k];
        }
        run += __popcll(m);
    }
    if (lane == 0) counts[pg] = run;
}

// ---- recognition crop: one block per crop, threads over (row, col) ----
__global__ __launch_bounds__(256) void rec_crop_kernel(const uint8_t* pages, int H, int W, const int* quads, const int* page_idx, uint8_t* crops,
                                                       int* widths) {
    const int ci = blockIdx.x, tid = threadIdx.x;
    long long p[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) { p[t][0] = quads[(size_t)ci * 8 + 2 * t]; p[t][1] = quads[(size_t)ci * 8 + 2 * t + 1]; }
#define D2(a, b) ((p[a][0] - p[b][0]) * (p[a][0] - p[b][0]) + (p[a][1] - p[b][1]) * (p[a][1] - p[b][1]))
    long long cw2 = D2(1, 0) > D2(2, 3) ? D2(1, 0) : D2(2, 3);
    long long ch2 = D2(3, 0) > D2(2, 1) ? D2(3, 0) : D2(2, 1);
#undef D2
    if (4 * ch2 >= 9 * cw2) {
        const long long t0 = p[0][0], t1 = p[0][1];
        p[0][0] = p[1][0]; p[0][1] = p[1][1]; p[1][0] = p[2][0]; p[1][1] = p[2][1];
        p[2][0] = p[3][0]; p[2][1] = p[3][1]; p[3][0] = t0; p[3][1] = t1;
        const long long t = cw2; cw2 = ch2; ch2 = t;
    }
    uint8_t* out = crops + (size_t)ci * 32 * 320 * 3;
    int wc = 0;
    if (ch2 != 0 && cw2 != 0) {
        const double ratio = sqrt(__ddiv_rn((double)cw2, (double)ch2));
        wc = (int)ceil(__dmul_rn(32.0, ratio));
        wc = wc < 1 ? 1 : (wc > 320 ? 320 : wc);
    }
    if (tid == 0) widths[ci] = wc;
    const uint8_t* page = pages + (size_t)page_idx[ci] * H * W * 3;
    const float tlx = (float)p[0][0], tly = (float)p[0][1];
    const float ex = (float)(p[1][0] - p[0][0]), ey = (float)(p[1][1] - p[0][1]);
    const float fx = (float)(p[3][0] - p[0][0]), fy = (float)(p[3][1] - p[0][1]);
    for (int t = tid; t < 32 * 320; t += 256) {
        const int i = t / 320, j = t - i * 320;
        uint8_t r3[3] = {0, 0, 0};
        if (j < wc) {
            const float u = __fdiv_rn((float)j + 0.5f, (float)wc), v = __fdiv_rn((float)i + 0.5f, 32.0f);
            const float t1 = __fmul_rn(u, ex), t2 = __fmul_rn(v, fx), t3 = __fmul_rn(u, ey), t4 = __fmul_rn(v, fy);
            const float sx = __fadd_rn(__fadd_rn(tlx, t1), t2), sy = __fadd_rn(__fadd_rn(tly, t3), t4);
            const float x0f = floorf(sx), y0f = floorf(sy);
            const float ax = __fsub_rn(sx, x0f), ay = __fsub_rn(sy, y0f);
            int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
            x0 = x0 < 0 ? 0 : (x0 > W - 1 ? W - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > W - 1 ? W - 1 : x1);
            y0 = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1);
            const float bx = __fsub_rn(1.0f, ax), by = __fsub_rn(1.0f, ay);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float p00 = page[((size_t)y0 * W + x0) * 3 + c], p01 = page[((size_t)y0 * W + x1) * 3 + c];
                const float p10 = page[((size_t)y1 * W + x0) * 3 + c], p11 = page[((size_t)y1 * W + x1) * 3 + c];
                const float top = __fadd_rn(__fmul_rn(bx, p00), __fmul_rn(ax, p01)), bot = __fadd_rn(__fmul_rn(bx, p10), __fmul_rn(ax, p11));
                float val = rintf(__fadd_rn(__fmul_rn(by, top), __fmul_rn(ay, bot)));
                val = val < 0.f ? 0.f : (val > 255.f ? 255.f : val);
                r3[c] = (uint8_t)val;
            }
        }
        out[(size_t)t * 3 + 0] = r3[0]; out[(size_t)t * 3 + 1] = r3[1]; out[(size_t)t * 3 + 2] = r3[2];
    }
}

}  // namespace

// worst case of a row: every other pixel starts a run
static size_t dbpost_runcap(int Hp, int Wp) { return (size_t)Hp * ((Wp + 1) / 2); }
static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t dbpost_workspace_bytes(int B, int Hp, int Wp, int maxc) {
    const size_t seg_cap = (size_t)maxc * Hp;  // worst case: every candidate spans the page height
    const size_t runcap = dbpost_runcap(Hp, Wp), nseg = (Wp + 63) / 64;
    size_t n = 0;
    n += al256((size_t)B * Hp * nseg * 8);             // masks
    n += 2 * al256((size_t)B * (Hp + 1) * 4);          // run / root counts -> offsets
    n += 2 * al256((size_t)B * 4);                     // nruns, ncomp
    n += 2 * al256((size_t)B * runcap * 2);            // run xs, xe
    n += 2 * al256((size_t)B * runcap * 4);            // parent, component id of root runs
    n += 2 * al256((size_t)B * maxc * 4);              // ymin, ymax
    n += al256((size_t)B * (maxc + 1) * 4);            // segoff
    n += 2 * al256((size_t)B * seg_cap * 4);           // rowmin, rowmax
    n += al256((size_t)B * seg_cap * 2 * 8);           // hull
    n += al256((size_t)B * maxc * 8 * 4) + 2 * al256((size_t)B * maxc * 4);
    return n + 4096;
}

hipError_t dbpost_launch(const DbPostParams& p, void* workspace, hipStream_t st) {
    const int B = p.B, Hp = p.Hp, Wp = p.Wp, maxc = p.max_boxes;
    if (B <= 0 || Hp <= 0 || Wp <= 0 || Wp > 65535 || maxc <= 0 || (size_t)B * Hp >= (1ull << 31) || dbpost_runcap(Hp, Wp) >= (1ull << 31)) return hipErrorInvalidValue;
    const size_t seg_cap = (size_t)maxc * Hp, runcap = dbpost_runcap(Hp, Wp);
    const int nseg = (Wp + 63) / 64;
    uint8_t* w = static_cast<uint8_t*>(workspace);
    auto take = [&](size_t bytes) { void* r = w; w += al256(bytes); return r; };
    unsigned long long* mask = static_cast<unsigned long long*>(take((size_t)B * Hp * nseg * 8));
    int* runoff = static_cast<int*>(take((size_t)B * (Hp + 1) * 4));
    int* rootoff = static_cast<int*>(take((size_t)B * (Hp + 1) * 4));
    int* nruns = static_cast<int*>(take((size_t)B * 4));
    int* ncomp = static_cast<int*>(take((size_t)B * 4));
    unsigned short* rxs = static_cast<unsigned short*>(take((size_t)B * runcap * 2));
    unsigned short* rxe = static_cast<unsigned short*>(take((size_t)B * runcap * 2));
    int* parent = static_cast<int*>(take((size_t)B * runcap * 4));
    int* cidr = static_cast<int*>(take((size_t)B * runcap * 4));
    int* ymin = static_cast<int*>(take((size_t)B * maxc * 4));
    int* ymax = static_cast<int*>(take((size_t)B * maxc * 4));
    int* segoff = static_cast<int*>(take((size_t)B * (maxc + 1) * 4));
    int* rowmin = static_cast<int*>(take((size_t)B * seg_cap * 4));
    int* rowmax = static_cast<int*>(take((size_t)B * seg_cap * 4));
    int2* hull = static_cast<int2*>(take((size_t)B * seg_cap * 2 * 8));
    int* box_tmp = static_cast<int*>(take((size_t)B * maxc * 8 * 4));
    float* score_tmp = static_cast<float*>(take((size_t)B * maxc * 4));
    int* valid_tmp = static_cast<int*>(take((size_t)B * maxc * 4));

    const int rows = B * Hp;
    const dim3 grows((unsigned)((rows + 3) / 4));
    hipLaunchKernelGGL(rl_mask_kernel, grows, dim3(256), 0, st, p.prob, mask, runoff, Hp, Wp, nseg, p.valid_h, p.valid_w, p.thresh, rows);
    hipLaunchKernelGGL(row_scan_kernel, dim3(B), dim3(64), 0, st, runoff, nruns, Hp);
    hipLaunchKernelGGL(rl_fill_kernel, grows, dim3(256), 0, st, mask, runoff, rxs, rxe, parent, Hp, nseg, runcap, rows);
    hipLaunchKernelGGL(rl_merge_kernel, grows, dim3(256), 0, st, runoff, rxs, rxe, parent, Hp, runcap, rows);
    hipLaunchKernelGGL(rl_roots_kernel, grows, dim3(256), 0, st, runoff, parent, rootoff, Hp, runcap, rows);
    hipLaunchKernelGGL(row_scan_kernel, dim3(B), dim3(64), 0, st, rootoff, ncomp, Hp);
    hipLaunchKernelGGL(rl_assign_kernel, grows, dim3(256), 0, st, runoff, rootoff, parent, cidr, ymin, ymax, Hp, runcap, maxc, rows);
    hipLaunchKernelGGL(rl_extent_kernel, grows, dim3(256), 0, st, runoff, parent, cidr, ymax, Hp, runcap, maxc, rows);
    hipLaunchKernelGGL(seg_scan_kernel, dim3(B), dim3(64), 0, st, ncomp, ymin, ymax, segoff, maxc);
    hipLaunchKernelGGL(seg_init_kernel, dim3(64, B), dim3(256), 0, st, ncomp, segoff, rowmin, rowmax, maxc, seg_cap);
    hipLaunchKernelGGL(rl_extremes_kernel, grows, dim3(256), 0, st, runoff, rxs, rxe, parent, cidr, ymin, segoff, rowmin, rowmax, Hp, runcap, maxc, seg_cap, rows);
    hipLaunchKernelGGL(comp_box_kernel, dim3(B * maxc), dim3(256), 0, st, p.prob, ncomp, ymin, ymax, segoff, rowmin, rowmax, hull, box_tmp,
                       score_tmp, valid_tmp, Hp, Wp, p.valid_h, p.valid_w, maxc, seg_cap, p.box_thresh, p.unclip_ratio, p.min_size);
    hipLaunchKernelGGL(compact_kernel, dim3(B), dim3(64), 0, st, box_tmp, score_tmp, valid_tmp, p.boxes, p.scores, p.counts, maxc);
    return hipGetLastError();
}

hipError_t rec_crop_launch(const uint8_t* pages, int H, int W, const int* quads, const int* page_idx, int n, uint8_t* crops, int* widths,
                           hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(rec_crop_kernel, dim3(n), dim3(256), 0, st, pages, H, W, quads, page_idx, crops, widths);
    return hipGetLastError();
}
