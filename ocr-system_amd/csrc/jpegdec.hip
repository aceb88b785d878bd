// Baseline JPEG decoder on the device (jpegdec.h): what the reference's Image.open(...) does for .jpg inputs
// (/root/reference/backend/utils/image_preprocessing.py:57-75), byte-identical to Pillow / libjpeg-turbo.
//
// A JPEG scan is ONE sequential bit stream per file (no restart markers in what Pillow, scanners' firmware or this engine's own
// encoder write by default), so the entropy decode is made parallel the way self-synchronising Huffman streams allow:
//   1. un-stuff: the 0xFF00 pairs and RSTn markers leave the stream (three byte-parallel kernels: mark / scan / compact); the RSTn
//      positions become a segment table (a segment start is an exact synchronisation point: byte aligned, state reset);
//   2. the clean stream is cut into 256-byte chunks, one thread each.  A chunk's decoder state is (bit position, block-in-MCU,
//      coefficient index).  Pass 0 decodes every chunk from a GUESSED state (its first bit, block 0, DC next); wrong guesses
//      re-synchronise with the true decode inside the chunk most of the time (Huffman codes self-synchronise within a few symbols,
//      the coefficient index at the next end-of-block, the block phase after a few table switches).  Pass j >= 1 starts chunk i from
//      chunk i-1's end state of pass j-1 and re-decodes only if that differs from what it started from before: a Jacobi iteration
//      whose fixed point IS the sequential decode (chunk 0 starts exact; by induction every chunk does).  Typically 2-4 passes;
//   3. a last pass writes the quantised coefficients (block indices from a prefix sum of the per-chunk DC counts), DC differences
//      are integrated per component (segmented by restart interval), then inverse DCT ("islow" integer), fancy chroma up-sampling
//      and YCbCr -> RGB are embarrassingly parallel kernels.
// Header parsing (a few hundred bytes of markers) is host code.
#include "jpegdec.h"

#include <cstring>
#include <vector>

#include "engine.h"

namespace {

constexpr int JD_CH = 1024;     // clean-stream bytes per chunk (256: a busy A4 page holds ~3 MCUs per chunk and only 36 % of the guessed starts re-synchronise inside it: 28 passes; 1024: 8)
constexpr int JD_UB = 1024;     // raw bytes per un-stuff block (256 threads x 4)
constexpr int JD_PASSES = 4;    // synchronisation passes between two looks at the "changed" flags

struct JdHuff { int maxcode[17]; int valptr[17]; unsigned short fast[256]; unsigned char vals[256]; };   // fast[8 bits] = len << 8 | symbol for codes of <= 8 bits, else 0
struct JdFile {
    unsigned raw_off, raw_len, clean_off, ublk_off, nublk, chunk_off, nchunk_cap, seg_off, seg_cap, coef_off, plane_off[3];
    int width, height, ncomp, hs, vs, mcux, mcuy, bpm, restart, nblk, valid;
    int pw[3], ph[3], comp_of_blk[6], first_blk_of_comp[3], td[3], ta[3];
    unsigned short q[3][64];
    JdHuff dc[3], ac[3];
};
struct JdDyn { unsigned clean_len; int nseg; int err; int total_dc; };   // per file, written on the device

__constant__ unsigned char JD_ZZ[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                        41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                        15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
const unsigned char H_ZZ[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ------------------------------------------------------------------------------------------------ 1. un-stuff
__device__ __forceinline__ bool jd_is_rst(unsigned char c) { return (c & 0xF8) == 0xD0; }
// raw byte j leaves the stream: a stuffed zero, or either byte of an RSTn marker
__device__ __forceinline__ bool jd_drop(const unsigned char* raw, unsigned j, unsigned len) {
    const unsigned char c = raw[j];
    if (c == 0x00) return j > 0 && raw[j - 1] == 0xFF;
    if (c == 0xFF) return j + 1 < len && jd_is_rst(raw[j + 1]);
    return jd_is_rst(c) && j > 0 && raw[j - 1] == 0xFF;
}
__device__ __forceinline__ bool jd_marker_start(const unsigned char* raw, unsigned j, unsigned len) {
    return raw[j] == 0xFF && j + 1 < len && jd_is_rst(raw[j + 1]);
}

__global__ __launch_bounds__(256) void jd_mark_kernel(const JdFile* files, const unsigned char* rawbuf, int* blk_keep, int* blk_rst) {
    const JdFile& F = files[blockIdx.y];
    if (!F.valid || blockIdx.x >= F.nublk) return;
    __shared__ int sk, sr;
    if (threadIdx.x == 0) { sk = 0; sr = 0; }
    __syncthreads();
    const unsigned char* raw = rawbuf + F.raw_off;
    int k = 0, r = 0;
    for (int e = 0; e < 4; ++e) {
        const unsigned j = blockIdx.x * JD_UB + threadIdx.x * 4 + e;
        if (j >= F.raw_len) break;
        k += jd_drop(raw, j, F.raw_len) ? 0 : 1;
        r += jd_marker_start(raw, j, F.raw_len) ? 1 : 0;
    }
    if (k) atomicAdd(&sk, k);
    if (r) atomicAdd(&sr, r);
    __syncthreads();
    if (threadIdx.x == 0) { blk_keep[F.ublk_off + blockIdx.x] = sk; blk_rst[F.ublk_off + blockIdx.x] = sr; }
}

// exclusive scan of n ints by one 256-thread workgroup (in place)
__device__ void jd_block_scan(int* a, int n) {
    __shared__ int part[256];
    const int t = threadIdx.x, per = (n + 255) / 256, lo = t * per, hi = min(lo + per, n);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += a[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) { int c = 0; for (int i = 0; i < 256; ++i) { const int v = part[i]; part[i] = c; c += v; } }
    __syncthreads();
    int c = part[t];
    for (int i = lo; i < hi; ++i) { const int v = a[i]; a[i] = c; c += v; }
    __syncthreads();
}

__global__ __launch_bounds__(256) void jd_scan_kernel(const JdFile* files, int* blk_keep, int* blk_rst, JdDyn* dyn) {
    const JdFile& F = files[blockIdx.x];
    if (!F.valid) return;
    __shared__ int tk, tr;
    int* k = blk_keep + F.ublk_off;
    int* r = blk_rst + F.ublk_off;
    const int n = (int)F.nublk;
    if (threadIdx.x == 0) {   // totals first (the scans below overwrite the counts)
        tk = 0; tr = 0;
    }
    __syncthreads();
    int sk = 0, sr = 0;
    for (int i = threadIdx.x; i < n; i += 256) { sk += k[i]; sr += r[i]; }
    if (sk) atomicAdd(&tk, sk);
    if (sr) atomicAdd(&tr, sr);
    __syncthreads();
    const int total_k = tk, total_r = tr;
    jd_block_scan(k, n);
    jd_block_scan(r, n);
    if (threadIdx.x == 0) {
        dyn[blockIdx.x].clean_len = (unsigned)total_k;
        dyn[blockIdx.x].nseg = total_r > (int)F.seg_cap ? -1 : total_r;   // more markers than the header's restart interval allows: corrupt
        dyn[blockIdx.x].err = 0;
        dyn[blockIdx.x].total_dc = 0;
    }
}

__global__ __launch_bounds__(256) void jd_compact_kernel(const JdFile* files, const unsigned char* rawbuf, const int* blk_keep, const int* blk_rst,
                                                         const JdDyn* dyn, unsigned char* cleanbuf, unsigned* seg_start) {
    const JdFile& F = files[blockIdx.y];
    if (!F.valid || blockIdx.x >= F.nublk) return;
    __shared__ int wk[4], wr[4];
    const unsigned char* raw = rawbuf + F.raw_off;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool keep[4], mark[4];
    int k = 0, r = 0;
    for (int e = 0; e < 4; ++e) {
        const unsigned j = blockIdx.x * JD_UB + threadIdx.x * 4 + e;
        keep[e] = j < F.raw_len && !jd_drop(raw, j, F.raw_len);
        mark[e] = j < F.raw_len && jd_marker_start(raw, j, F.raw_len);
        k += keep[e]; r += mark[e];
    }
    // exclusive scan over the 256 threads: wave scan by shuffles, wave totals through LDS
    int ik = k, ir = r;
    for (int d = 1; d < 64; d <<= 1) {
        const int a = __shfl_up(ik, d), b = __shfl_up(ir, d);
        if (lane >= d) { ik += a; ir += b; }
    }
    if (lane == 63) { wk[wave] = ik; wr[wave] = ir; }
    __syncthreads();
    int bk = 0, br = 0;
    for (int w = 0; w < wave; ++w) { bk += wk[w]; br += wr[w]; }
    int pk = blk_keep[F.ublk_off + blockIdx.x] + bk + ik - k;     // clean index of this thread's first kept byte
    int pr = blk_rst[F.ublk_off + blockIdx.x] + br + ir - r;
    unsigned char* clean = cleanbuf + F.clean_off;
    const int nseg = dyn[blockIdx.y].nseg;
    for (int e = 0; e < 4; ++e) {
        const unsigned j = blockIdx.x * JD_UB + threadIdx.x * 4 + e;
        if (mark[e] && nseg >= 0 && pr < (int)F.seg_cap) seg_start[F.seg_off + pr] = (unsigned)pk;   // the segment after this marker starts at the next kept byte
        pr += mark[e];
        if (keep[e]) clean[pk] = raw[j];
        pk += keep[e];
    }
}

// ------------------------------------------------------------------------------------------------ 2. parallel Huffman decode
struct JdRd {   // bit reader over the clean stream: acc holds `have` valid bits, MSB first, starting at stream bit `pos`
    const unsigned* w; unsigned pos; unsigned long long acc; int have;
};
__device__ __forceinline__ void jd_seek(JdRd& r, unsigned pos) {
    const unsigned byte = pos >> 3, wi = byte >> 2, sh = (byte & 3) * 8 + (pos & 7);
    const unsigned a = __builtin_bswap32(r.w[wi]), b = __builtin_bswap32(r.w[wi + 1]), c = __builtin_bswap32(r.w[wi + 2]);
    const unsigned long long hi = ((unsigned long long)a << 32) | b;
    r.acc = sh ? (hi << sh) | ((unsigned long long)c >> (32 - sh)) : hi;
    r.have = 64; r.pos = pos;
}
__device__ __forceinline__ unsigned jd_peek(JdRd& r, int n) {   // 1 <= n <= 32
    if (r.have < n) jd_seek(r, r.pos);
    return (unsigned)(r.acc >> (64 - n));
}
__device__ __forceinline__ void jd_skip(JdRd& r, int n) { r.acc <<= n; r.have -= n; r.pos += (unsigned)n; }
// -> symbol, or -1 (no code matches: corrupt data or a decoder that is not synchronised); always consumes at least one bit
__device__ __forceinline__ int jd_sym(JdRd& r, const JdHuff& t) {
    const unsigned p16 = jd_peek(r, 16);
    const unsigned f = t.fast[p16 >> 8];
    if (f) { jd_skip(r, (int)(f >> 8)); return (int)(f & 255u); }
    for (int l = 9; l <= 16; ++l) {
        const int code = (int)(p16 >> (16 - l));
        if (code <= t.maxcode[l]) { jd_skip(r, l); return t.vals[(t.valptr[l] + code) & 255]; }
    }
    jd_skip(r, 16);
    return -1;
}
__device__ __forceinline__ int jd_extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// Decodes from state (pos, b, k) until the first symbol boundary at or beyond end_bit (or the end of the stream).
// WRITE: stores the coefficients (DC as a difference) — blk = index of the block in progress (k > 0) or of the next block (k == 0).
template <bool WRITE>
__device__ void jd_span(const JdFile& F, const unsigned* clean_w, unsigned total_bits, const unsigned* seg, int nseg, unsigned pos, int b, int k,
                        unsigned end_bit, unsigned* out_pos, int* out_bk, int* out_ndc, short* coef, int blk, int* err) {
    JdRd r; r.w = clean_w; r.have = 0; r.pos = pos; r.acc = 0;
    // the segment (restart interval) the position lies in: seg[s] = byte offset where segment s + 1 starts
    int s = 0;
    { int lo = 0, hi = nseg; while (lo < hi) { const int mid = (lo + hi) >> 1; if (seg[mid] * 8u <= pos) lo = mid + 1; else hi = mid; } s = lo; }
    unsigned seg_end = s < nseg ? seg[s] * 8u : total_bits;
    int ndc = 0;
    int cur = k > 0 ? blk - 1 : blk;   // block whose AC coefficients are being written
    if (!WRITE) (void)cur;
    while (r.pos < end_bit) {
        const unsigned rem = seg_end - r.pos;
        if (rem < 8u) {   // inside the last byte of a segment: 1-bits up to its end are padding (no Huffman code is all ones, and no symbol crosses a marker)
            bool pad = rem == 0u;
            if (!pad) pad = jd_peek(r, (int)rem) == ((1u << rem) - 1u);
            if (pad) {
                if (seg_end >= total_bits) { r.pos = total_bits; b = 0; k = 0; break; }
                r.pos = seg_end; r.have = 0; b = 0; k = 0;
                ++s; seg_end = s < nseg ? seg[s] * 8u : total_bits;
                continue;
            }
        }
        const int comp = F.comp_of_blk[b];
        if (k == 0) {
            int sym = jd_sym(r, F.dc[F.td[comp]]);
            if (sym < 0 || sym > 15) { if (WRITE) *err = 1; sym = 0; }
            int diff = 0;
            if (sym) { diff = jd_extend((int)jd_peek(r, sym), sym); jd_skip(r, sym); }
            if (WRITE) { cur = blk + ndc; if (cur < F.nblk) coef[(size_t)cur * 64] = (short)diff; else *err = 1; }
            ++ndc; k = 1;
        } else {
            const int rs = jd_sym(r, F.ac[F.ta[comp]]);
            bool end_block = false;
            if (rs < 0) { if (WRITE) *err = 1; end_block = true; }
            else {
                const int run = rs >> 4, sz = rs & 15;
                if (sz == 0) {
                    if (run == 15) { k += 16; if (k > 63) { end_block = true; if (WRITE && k > 64) *err = 1; } }
                    else end_block = true;
                } else {
                    k += run;
                    const int v = jd_extend((int)jd_peek(r, sz), sz);
                    jd_skip(r, sz);
                    if (k > 63) { if (WRITE) *err = 1; end_block = true; }
                    else {
                        if (WRITE && cur >= 0 && cur < F.nblk) coef[(size_t)cur * 64 + JD_ZZ[k]] = (short)v;
                        if (++k > 63) end_block = true;
                    }
                }
            }
            if (end_block) { k = 0; b = b + 1 == F.bpm ? 0 : b + 1; }
        }
        if (r.pos > seg_end) {   // a symbol ran over a segment boundary: only a decoder that is not synchronised (or a corrupt file) gets here
            if (WRITE) *err = 1;
            if (seg_end >= total_bits) { r.pos = total_bits; b = 0; k = 0; break; }
            r.pos = seg_end; r.have = 0; b = 0; k = 0;
            ++s; seg_end = s < nseg ? seg[s] * 8u : total_bits;
        }
    }
    *out_pos = r.pos; *out_bk = (b << 8) | k; *out_ndc = ndc;
}

struct JdChunks { unsigned* start_pos; int* start_bk; unsigned* end_pos[2]; int* end_bk[2]; int* ndc; int* first_blk; };

__global__ __launch_bounds__(64) void jd_sync_kernel(const JdFile* files, const JdDyn* dyn, const unsigned char* cleanbuf, const unsigned* seg_start, JdChunks C,
                                                     int pass, int* changed /* [file] of this pass */) {
    const JdFile& F = files[blockIdx.y];
    const JdDyn& D = dyn[blockIdx.y];
    if (!F.valid || D.nseg < 0) return;
    const unsigned i = blockIdx.x * 64 + threadIdx.x;
    const unsigned nchunks = (D.clean_len + JD_CH - 1) / JD_CH;
    if (i >= nchunks) return;
    const unsigned ci = F.chunk_off + i;
    const int src = pass & 1, dst = src ^ 1;   // pass j reads end[j & 1] (written by pass j - 1), writes end[(j & 1) ^ 1]
    unsigned pos; int bk;
    if (i == 0) { pos = 0; bk = 0; }
    else if (pass == 0) { pos = i * (JD_CH * 8u); bk = 0; }
    else { pos = C.end_pos[src][ci - 1]; bk = C.end_bk[src][ci - 1]; }
    if (pass > 0 && pos == C.start_pos[ci] && bk == C.start_bk[ci]) {   // same start as before: same end
        C.end_pos[dst][ci] = C.end_pos[src][ci]; C.end_bk[dst][ci] = C.end_bk[src][ci];
        return;
    }
    C.start_pos[ci] = pos; C.start_bk[ci] = bk;
    const unsigned total_bits = D.clean_len * 8u;
    unsigned end_bit = (i + 1) * (JD_CH * 8u);
    if (end_bit > total_bits) end_bit = total_bits;
    unsigned op = pos; int obk = bk, ondc = 0, err = 0;
    if (pos < end_bit)
        jd_span<false>(F, reinterpret_cast<const unsigned*>(cleanbuf + F.clean_off), total_bits, seg_start + F.seg_off, D.nseg, pos, bk >> 8, bk & 255, end_bit,
                       &op, &obk, &ondc, nullptr, 0, &err);
    C.end_pos[dst][ci] = op; C.end_bk[dst][ci] = obk; C.ndc[ci] = ondc;
    if (pass > 0) changed[blockIdx.y] = 1;
}

__global__ __launch_bounds__(256) void jd_blkscan_kernel(const JdFile* files, JdDyn* dyn, JdChunks C) {
    const JdFile& F = files[blockIdx.x];
    JdDyn& D = dyn[blockIdx.x];
    if (!F.valid || D.nseg < 0) return;
    const int n = (int)((D.clean_len + JD_CH - 1) / JD_CH);
    __shared__ int tot;
    if (threadIdx.x == 0) tot = 0;
    __syncthreads();
    int s = 0;
    for (int i = threadIdx.x; i < n; i += 256) { const int v = C.ndc[F.chunk_off + i]; C.first_blk[F.chunk_off + i] = v; s += v; }
    if (s) atomicAdd(&tot, s);
    __syncthreads();
    jd_block_scan(C.first_blk + F.chunk_off, n);
    if (threadIdx.x == 0) { D.total_dc = tot; if (tot != F.nblk) D.err = 1; }
}

__global__ __launch_bounds__(64) void jd_write_kernel(const JdFile* files, JdDyn* dyn, const unsigned char* cleanbuf, const unsigned* seg_start, JdChunks C,
                                                      short* coefbuf) {
    const JdFile& F = files[blockIdx.y];
    JdDyn& D = dyn[blockIdx.y];
    if (!F.valid || D.nseg < 0 || D.err) return;
    const unsigned i = blockIdx.x * 64 + threadIdx.x;
    const unsigned nchunks = (D.clean_len + JD_CH - 1) / JD_CH;
    if (i >= nchunks) return;
    const unsigned ci = F.chunk_off + i;
    const unsigned pos = C.start_pos[ci];
    const int bk = C.start_bk[ci];
    const unsigned total_bits = D.clean_len * 8u;
    unsigned end_bit = (i + 1) * (JD_CH * 8u);
    if (end_bit > total_bits) end_bit = total_bits;
    if (pos >= end_bit) return;
    unsigned op; int obk, ondc, err = 0;
    jd_span<true>(F, reinterpret_cast<const unsigned*>(cleanbuf + F.clean_off), total_bits, seg_start + F.seg_off, D.nseg, pos, bk >> 8, bk & 255, end_bit, &op, &obk,
                  &ondc, coefbuf + (size_t)F.coef_off * 64, C.first_blk[ci], &err);
    if (err) D.err = 1;
}

// ------------------------------------------------------------------------------------------------ 3. DC integration, IDCT, colour
// one workgroup per (file, component): DC differences -> absolute values, the prediction reset at every restart interval
__global__ __launch_bounds__(256) void jd_dc_kernel(const JdFile* files, const JdDyn* dyn, short* coefbuf) {
    const JdFile& F = files[blockIdx.y];
    const int c = blockIdx.x;
    if (!F.valid || dyn[blockIdx.y].nseg < 0 || dyn[blockIdx.y].err || c >= F.ncomp) return;
    short* coef = coefbuf + (size_t)F.coef_off * 64;
    const int nb = c == 0 ? F.hs * F.vs : 1, mcus = F.mcux * F.mcuy, n = mcus * nb, first = F.first_blk_of_comp[c];
    __shared__ int ssum[256], sreset[256], carry[256];
    const int t = threadIdx.x, per = (n + 255) / 256, lo = t * per, hi = min(lo + per, n);
    auto blk_of = [&](int j) { return (j / nb) * F.bpm + first + (j % nb); };
    auto resets = [&](int j) { return F.restart > 0 && j % nb == 0 && (j / nb) % F.restart == 0; };
    int s = 0, rs = 0;
    for (int j = lo; j < hi; ++j) {
        if (resets(j)) { s = 0; rs = 1; }
        s += coef[(size_t)blk_of(j) * 64];
    }
    ssum[t] = s; sreset[t] = rs;
    __syncthreads();
    if (t == 0) { int cc = 0; for (int i = 0; i < 256; ++i) { carry[i] = cc; cc = sreset[i] ? ssum[i] : cc + ssum[i]; } }
    __syncthreads();
    int run = carry[t];
    for (int j = lo; j < hi; ++j) {
        if (resets(j)) run = 0;
        run += coef[(size_t)blk_of(j) * 64];
        coef[(size_t)blk_of(j) * 64] = (short)run;
    }
}

#define JD_DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))
__device__ __forceinline__ unsigned char jd_limit(int x) {   // the IDCT's range_limit[x & RANGE_MASK]: clamp(x + 128) for -512 <= x < 512
    const int i = x & 1023;
    return (unsigned char)(i < 128 ? i + 128 : (i < 512 ? 255 : (i < 896 ? 0 : i - 896)));
}
// 1-D "islow" butterfly on 8 de-quantised inputs (jidctint.c); outputs before the final descale
__device__ __forceinline__ void jd_idct8(const int in[8], int out[8]) {
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * 4433;
    int tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;
    int tmp0 = (in[0] + in[4]) * 8192, tmp1 = (in[0] - in[4]) * 8192;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * 9633;
    tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    out[0] = tmp10 + tmp3; out[7] = tmp10 - tmp3; out[1] = tmp11 + tmp2; out[6] = tmp11 - tmp2;
    out[2] = tmp12 + tmp1; out[5] = tmp12 - tmp1; out[3] = tmp13 + tmp0; out[4] = tmp13 - tmp0;
}

// 8 lanes per block: lane l does column l (pass 1, into LDS), then row l (pass 2, one 8-byte store into the component plane)
__global__ __launch_bounds__(256) void jd_idct_kernel(const JdFile* files, const JdDyn* dyn, const short* coefbuf, unsigned char* planebuf) {
    const JdFile& F = files[blockIdx.y];
    __shared__ int ws[32][64];
    const int lb = threadIdx.x >> 3, l = threadIdx.x & 7;
    const int B = blockIdx.x * 32 + lb;
    const bool on = F.valid && dyn[blockIdx.y].nseg >= 0 && !dyn[blockIdx.y].err && B < F.nblk;
    int comp = 0, sub = 0, m = 0;
    if (on) {
        m = B / F.bpm;
        const int i = B - m * F.bpm;
        comp = F.comp_of_blk[i]; sub = i - F.first_blk_of_comp[comp];
        const short* c = coefbuf + ((size_t)F.coef_off + B) * 64;
        const unsigned short* q = F.q[comp];
        int in[8], o[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = (int)c[8 * r + l] * (int)q[8 * r + l];
        jd_idct8(in, o);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[lb][8 * r + l] = JD_DESCALE(o[r], 11);
    }
    __syncthreads();
    if (on) {
        int in[8], o[8];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) in[cc] = ws[lb][8 * l + cc];
        jd_idct8(in, o);
        const int hsc = comp == 0 ? F.hs : 1, vsc = comp == 0 ? F.vs : 1;
        const int my = m / F.mcux, mx = m - my * F.mcux, by = sub / hsc, bx = sub - by * hsc;
        unsigned char* dst = planebuf + F.plane_off[comp] + (size_t)((my * vsc + by) * 8 + l) * F.pw[comp] + (mx * hsc + bx) * 8;
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) { lo |= (unsigned)jd_limit(JD_DESCALE(o[cc], 18)) << (8 * cc); hi |= (unsigned)jd_limit(JD_DESCALE(o[4 + cc], 18)) << (8 * cc); }
        *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
    }
}

// chroma sample at full-resolution position (x, y): libjpeg-turbo's fancy up-sampling (jdsample.c), or replication for very narrow planes
__device__ __forceinline__ int jd_chroma(const unsigned char* pl, int pw, int dw, int dh, int x, int y, int h2, int v2) {
    if (!h2) return pl[(size_t)y * pw + x];
    const int cx = x >> 1, cy = v2 ? y >> 1 : y;
    const unsigned char* r0 = pl + (size_t)cy * pw;
    if (dw <= 2) return r0[cx];
    if (!v2) {
        if (x == 0) return r0[0];
        if (x == 2 * dw - 1) return r0[dw - 1];
        return (x & 1) ? (3 * r0[cx] + r0[cx + 1] + 2) >> 2 : (3 * r0[cx] + r0[cx - 1] + 1) >> 2;
    }
    int far = (y & 1) ? cy + 1 : cy - 1;
    far = far < 0 ? 0 : (far > dh - 1 ? dh - 1 : far);
    const unsigned char* r1 = pl + (size_t)far * pw;
    const int cur = 3 * r0[cx] + r1[cx];
    if (x & 1) {
        if (cx == dw - 1) return (cur * 4 + 7) >> 4;
        return (cur * 3 + 3 * r0[cx + 1] + r1[cx + 1] + 7) >> 4;
    }
    if (cx == 0) return (cur * 4 + 8) >> 4;
    return (cur * 3 + 3 * r0[cx - 1] + r1[cx - 1] + 8) >> 4;
}
__device__ __forceinline__ unsigned char jd_clamp(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

__global__ __launch_bounds__(256) void jd_color_kernel(const JdFile* files, const JdDyn* dyn, const unsigned char* planebuf, unsigned char* out, int height, int width) {
    const JdFile& F = files[blockIdx.z];
    if (!F.valid || dyn[blockIdx.z].nseg < 0 || dyn[blockIdx.z].err) return;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= width || y >= height) return;
    const int yy = planebuf[F.plane_off[0] + (size_t)y * F.pw[0] + x];
    unsigned char* o = out + (((size_t)blockIdx.z * height + y) * width + x) * 3;
    if (F.ncomp == 1) { o[0] = o[1] = o[2] = (unsigned char)yy; return; }
    const int h2 = F.hs == 2, v2 = F.vs == 2;
    const int dw = (width + F.hs - 1) / F.hs, dh = (height + F.vs - 1) / F.vs;
    const int cb = jd_chroma(planebuf + F.plane_off[1], F.pw[1], dw, dh, x, y, h2, v2) - 128;
    const int cr = jd_chroma(planebuf + F.plane_off[2], F.pw[2], dw, dh, x, y, h2, v2) - 128;
    o[0] = jd_clamp(yy + ((91881 * cr + 32768) >> 16));
    o[1] = jd_clamp(yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
    o[2] = jd_clamp(yy + ((116130 * cb + 32768) >> 16));
}

// per-file outcome of an asynchronous decode: host parse code, else -1 (corrupt), -5 (the fixed number of passes did not reach the fixed point), 0
__global__ void jd_status_kernel(const JdFile* files, const JdDyn* dyn, const int* host_rc, const int* changed_last, int* status, int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    int rc = host_rc[i];
    if (rc == 0 && files[i].valid) rc = dyn[i].nseg < 0 ? -1 : (changed_last[i] ? -5 : (dyn[i].err ? -1 : 0));   // (before the fixed point the error flag means nothing)
    status[i] = rc;
}

// ------------------------------------------------------------------------------------------------ host: headers
struct HostHeader {
    int width = 0, height = 0, ncomp = 0, hs[3] = {1, 1, 1}, vs[3] = {1, 1, 1}, tq[3] = {0, 0, 0}, td[3] = {0, 0, 0}, ta[3] = {0, 0, 0}, restart = 0;
    unsigned short q[4][64];
    unsigned char bits[2][4][17], vals[2][4][256];
    bool have_q[4] = {false, false, false, false}, have_h[2][4] = {{false, false, false, false}, {false, false, false, false}};
    size_t scan_off = 0, scan_end = 0;
};

int parse_header(const uint8_t* f, size_t n, HostHeader* h) {
    if (n < 4 || f[0] != 0xFF || f[1] != 0xD8) return -1;
    size_t p = 2;
    bool sof = false;
    for (;;) {
        if (p + 4 > n || f[p] != 0xFF) return -1;
        while (p < n && f[p] == 0xFF) ++p;
        if (p >= n) return -1;
        const int m = f[p++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9 || p + 2 > n) return -1;
        const size_t len = ((size_t)f[p] << 8) | f[p + 1];
        if (len < 2 || p + len > n) return -1;
        const uint8_t* s = f + p + 2;
        const size_t sl = len - 2;
        if (m == 0xDB) {
            size_t i = 0;
            while (i < sl) {
                const int pq = s[i] >> 4, t = s[i] & 15;
                ++i;
                if (t > 3 || pq > 1 || i + (pq ? 128 : 64) > sl) return -1;
                for (int k = 0; k < 64; ++k) { h->q[t][H_ZZ[k]] = pq ? (unsigned short)((s[i] << 8) | s[i + 1]) : s[i]; i += pq ? 2 : 1; }
                h->have_q[t] = true;
            }
        } else if (m == 0xC4) {
            size_t i = 0;
            while (i < sl) {
                const int tc = s[i] >> 4, t = s[i] & 15;
                ++i;
                if (tc > 1 || t > 3 || i + 16 > sl) return -1;
                int cnt = 0;
                h->bits[tc][t][0] = 0;
                for (int k = 1; k <= 16; ++k) { h->bits[tc][t][k] = s[i + k - 1]; cnt += s[i + k - 1]; }
                i += 16;
                if (cnt > 256 || i + cnt > sl) return -1;
                memset(h->vals[tc][t], 0, 256);
                memcpy(h->vals[tc][t], s + i, (size_t)cnt);
                i += cnt;
                h->have_h[tc][t] = true;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (sl < 6 || sof) return -1;
            if (s[0] != 8) return -2;
            h->height = (s[1] << 8) | s[2]; h->width = (s[3] << 8) | s[4]; h->ncomp = s[5];
            if (h->height == 0 || h->width == 0 || (h->ncomp != 1 && h->ncomp != 3)) return -2;
            if (sl < (size_t)(6 + 3 * h->ncomp)) return -1;
            for (int c = 0; c < h->ncomp; ++c) {
                if (s[6 + 3 * c] != c + 1) return -2;
                h->hs[c] = s[7 + 3 * c] >> 4; h->vs[c] = s[7 + 3 * c] & 15; h->tq[c] = s[8 + 3 * c];
                if (h->tq[c] > 3) return -1;
            }
            sof = true;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return -2;
        } else if (m == 0xDD) {
            if (sl < 2) return -1;
            h->restart = (s[0] << 8) | s[1];
        } else if (m == 0xEE) {
            if (sl >= 12 && memcmp(s, "Adobe", 5) == 0 && s[11] != 1 && h->ncomp != 1) return -2;
        } else if (m == 0xDA) {
            if (!sof) return -1;
            if (sl < 1 || s[0] != h->ncomp || sl < (size_t)(1 + 2 * h->ncomp + 3)) return -2;
            for (int c = 0; c < h->ncomp; ++c) {
                if (s[1 + 2 * c] != c + 1) return -2;
                h->td[c] = s[2 + 2 * c] >> 4; h->ta[c] = s[2 + 2 * c] & 15;
                if (h->td[c] > 3 || h->ta[c] > 3 || !h->have_h[0][h->td[c]] || !h->have_h[1][h->ta[c]] || !h->have_q[h->tq[c]]) return -1;
            }
            h->scan_off = p + len;
            break;
        }
        p += len;
    }
    if (h->ncomp == 1) { h->hs[0] = h->vs[0] = 1; }
    else {
        if (h->hs[1] != 1 || h->vs[1] != 1 || h->hs[2] != 1 || h->vs[2] != 1) return -2;
        if (!((h->hs[0] == 1 && h->vs[0] == 1) || (h->hs[0] == 2 && h->vs[0] == 1) || (h->hs[0] == 2 && h->vs[0] == 2))) return -2;
    }
    // the entropy-coded segment ends at the last EOI (trailing bytes after it are ignored, as libjpeg does)
    size_t e = n;
    while (e >= h->scan_off + 2 && !(f[e - 2] == 0xFF && f[e - 1] == 0xD9)) --e;
    if (e < h->scan_off + 2) return -1;
    h->scan_end = e - 2;
    // any marker other than RSTn inside the scan (DNL, a second SOS ...) is outside the subset
    for (const uint8_t* q = f + h->scan_off; q + 1 < f + h->scan_end;) {   // (memchr: the scan is megabytes, 0xFF bytes are rare)
        q = static_cast<const uint8_t*>(memchr(q, 0xFF, (size_t)(f + h->scan_end - 1 - q)));
        if (q == nullptr) break;
        if (q[1] != 0x00 && (q[1] & 0xF8) != 0xD0 && q[1] != 0xFF) return -2;
        ++q;
    }
    return 0;
}

bool build_huff(const unsigned char bits[17], const unsigned char* vals, JdHuff* t) {
    int code = 0, k = 0;
    memset(t, 0, sizeof(*t));
    memcpy(t->vals, vals, 256);
    for (int l = 1; l <= 16; ++l) {
        t->valptr[l] = k - code;
        if (bits[l]) {
            if (l <= 8)
                for (int i = 0; i < bits[l]; ++i) {
                    const int c = code + i;
                    for (int f = 0; f < (1 << (8 - l)); ++f) t->fast[(c << (8 - l)) | f] = (unsigned short)((l << 8) | vals[k + i]);
                }
            k += bits[l]; code += bits[l];
            if (code > (1 << l) || k > 256) return false;
            t->maxcode[l] = code - 1;
        } else t->maxcode[l] = -1;
        code <<= 1;
    }
    return true;
}

template <class T> T* carve(uint8_t*& p, size_t count) {
    uintptr_t a = (reinterpret_cast<uintptr_t>(p) + 255) & ~(uintptr_t)255;
    T* r = reinterpret_cast<T*>(a);
    p = reinterpret_cast<uint8_t*>(a + count * sizeof(T));
    return r;
}

}  // namespace

#define JDCHK(expr)                                                                   \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) return locr_fail(eng, #expr, hipGetErrorString(_e));    \
    } while (0)

int jpegdec_probe(const uint8_t* file, size_t n, JdInfo* info) {
    HostHeader h;
    const int rc = parse_header(file, n, &h);
    if (info) { info->width = h.width; info->height = h.height; info->ncomp = h.ncomp; info->hs = h.hs[0]; info->vs = h.vs[0]; info->restart = h.restart; }
    return rc;
}

int jpegdec_run(lumina_ocr* eng, const uint8_t* const* files, const size_t* sizes, int n, int height, int width, uint8_t* out_dev, int* status, hipStream_t st,
                int async_passes) {
    // async_passes > 0: that many synchronisation passes are enqueued without looking at their result, nothing is synchronised, and
    // `status` (PINNED host memory of the caller) is filled by a copy at the end of the stream's work: -5 = the passes did not suffice
    std::vector<JdFile> F((size_t)n);
    std::vector<const uint8_t*> scan_src((size_t)n, nullptr);
    // the scan bytes of all files go through ONE pinned staging buffer (kept by the engine) and ONE asynchronous copy
    size_t raw_cap = 0;
    for (int i = 0; i < n; ++i) raw_cap += ((sizes[i] + 16 + 255) & ~(size_t)255);
    // two staging buffers in turn: an asynchronous call's upload may still be queued when the next call fills its buffer
    const int sb = eng->jd_stage_next; eng->jd_stage_next ^= 1;
    if (eng->jd_stage_ev[sb]) JDCHK(hipEventSynchronize(eng->jd_stage_ev[sb]));
    else JDCHK(hipEventCreateWithFlags(&eng->jd_stage_ev[sb], hipEventDisableTiming));
    if (raw_cap > eng->jd_stage_cap[sb]) {
        if (eng->jd_stage[sb]) (void)hipHostFree(eng->jd_stage[sb]);
        eng->jd_stage[sb] = nullptr; eng->jd_stage_cap[sb] = 0;
        JDCHK(hipHostMalloc(reinterpret_cast<void**>(&eng->jd_stage[sb]), raw_cap + 4096, hipHostMallocDefault));
        eng->jd_stage_cap[sb] = raw_cap;
    }
    uint8_t* raw = eng->jd_stage[sb];
    size_t raw_total = 0, ublk_total = 0, chunk_total = 0, seg_total = 0, blk_total = 0, plane_total = 0;
    unsigned max_ublk = 0, max_chunks = 0;
    int max_blk = 0, any = 0;
    for (int i = 0; i < n; ++i) {
        JdFile& f = F[(size_t)i];
        memset(&f, 0, sizeof(f));
        HostHeader h;
        int rc = parse_header(files[i], sizes[i], &h);
        if (rc == 0 && (h.width != width || h.height != height)) rc = -4;
        status[i] = rc;
        if (rc) continue;
        f.valid = 1; ++any;
        f.width = h.width; f.height = h.height; f.ncomp = h.ncomp; f.hs = h.hs[0]; f.vs = h.vs[0]; f.restart = h.restart;
        f.mcux = (h.width + 8 * f.hs - 1) / (8 * f.hs); f.mcuy = (h.height + 8 * f.vs - 1) / (8 * f.vs);
        f.bpm = f.hs * f.vs + (h.ncomp == 3 ? 2 : 0);
        f.nblk = f.mcux * f.mcuy * f.bpm;
        int b = 0;
        for (int c = 0; c < h.ncomp; ++c) {
            f.first_blk_of_comp[c] = b;
            for (int k = 0; k < (c == 0 ? f.hs * f.vs : 1); ++k) f.comp_of_blk[b++] = c;
            f.pw[c] = f.mcux * (c == 0 ? f.hs : 1) * 8; f.ph[c] = f.mcuy * (c == 0 ? f.vs : 1) * 8;
            f.plane_off[c] = (unsigned)plane_total; plane_total += ((size_t)f.pw[c] * f.ph[c] + 255) & ~(size_t)255;
            memcpy(f.q[c], h.q[h.tq[c]], sizeof(f.q[c]));
            f.td[c] = c; f.ta[c] = c;   // every component gets its own copy of its tables: no index indirection on the device
            if (!build_huff(h.bits[0][h.td[c]], h.vals[0][h.td[c]], &f.dc[c]) || !build_huff(h.bits[1][h.ta[c]], h.vals[1][h.ta[c]], &f.ac[c])) { status[i] = -1; f.valid = 0; }
        }
        if (!f.valid) { --any; continue; }
        f.raw_len = (unsigned)(h.scan_end - h.scan_off);
        f.raw_off = (unsigned)raw_total; f.clean_off = f.raw_off;
        const size_t padded = ((size_t)f.raw_len + 16 + 255) & ~(size_t)255;   // (the bit reader loads three aligned words past its position)
        scan_src[(size_t)i] = files[i] + h.scan_off;
        raw_total += padded;
        f.nublk = (f.raw_len + JD_UB - 1) / JD_UB; f.ublk_off = (unsigned)ublk_total; ublk_total += f.nublk;
        f.nchunk_cap = (f.raw_len + JD_CH - 1) / JD_CH + 1; f.chunk_off = (unsigned)chunk_total; chunk_total += f.nchunk_cap;
        f.seg_cap = h.restart ? (unsigned)((f.mcux * f.mcuy + h.restart - 1) / h.restart) : 0; f.seg_off = (unsigned)seg_total; seg_total += f.seg_cap + 1;
        f.coef_off = (unsigned)blk_total; blk_total += (size_t)f.nblk;
        if (f.nublk > max_ublk) max_ublk = f.nublk;
        if (f.nchunk_cap > max_chunks) max_chunks = f.nchunk_cap;
        if (f.nblk > max_blk) max_blk = f.nblk;
        if (raw_total >= (1ull << 31) || blk_total >= (1ull << 31)) return locr_fail(eng, "jpeg_decode", "batch too large (2 GiB of scan data / 2^31 blocks)");
    }
    if (!any) { if (async_passes > 0) { /* status is already final */ } return 0; }
    // ---- workspace ----
    size_t need = 4096;
    auto add = [&](size_t bytes) { need += (bytes + 255) & ~(size_t)255; };
    add(sizeof(JdFile) * n); add(sizeof(JdDyn) * n); add(raw_total); add(raw_total + 64); add(4 * ublk_total * 2 + 64); add(4 * seg_total);
    for (int k = 0; k < 8; ++k) add(4 * chunk_total + 64);
    add(blk_total * 128); add(plane_total); add(4 * (size_t)n * (JD_PASSES + 1)); add(4096); add(8 * (size_t)n + 512);
    if (eng_ws_reserve(eng, need)) return 1;
    uint8_t* p = eng->ws;
    JdFile* dF = carve<JdFile>(p, (size_t)n);
    JdDyn* dD = carve<JdDyn>(p, (size_t)n);
    uint8_t* dRaw = carve<uint8_t>(p, raw_total);
    uint8_t* dClean = carve<uint8_t>(p, raw_total + 64);
    int* dKeep = carve<int>(p, ublk_total + 8);
    int* dRst = carve<int>(p, ublk_total + 8);
    unsigned* dSeg = carve<unsigned>(p, seg_total);
    JdChunks C;
    C.start_pos = carve<unsigned>(p, chunk_total + 8); C.start_bk = carve<int>(p, chunk_total + 8);
    C.end_pos[0] = carve<unsigned>(p, chunk_total + 8); C.end_pos[1] = carve<unsigned>(p, chunk_total + 8);
    C.end_bk[0] = carve<int>(p, chunk_total + 8); C.end_bk[1] = carve<int>(p, chunk_total + 8);
    C.ndc = carve<int>(p, chunk_total + 8); C.first_blk = carve<int>(p, chunk_total + 8);
    short* dCoef = carve<short>(p, blk_total * 64);
    uint8_t* dPlane = carve<uint8_t>(p, plane_total);
    int* dChanged = carve<int>(p, (size_t)n * (JD_PASSES + 1));
    int* dHostRc = carve<int>(p, (size_t)n);
    int* dStatus = carve<int>(p, (size_t)n);
    if ((size_t)(p - eng->ws) > eng->ws_cap) return locr_fail(eng, "jpeg_decode", "workspace layout exceeds the reservation");
    JDCHK(hipMemcpyAsync(dF, F.data(), sizeof(JdFile) * n, hipMemcpyHostToDevice, st));
    // scan bytes: host copy into the pinned buffer and the asynchronous upload of the previous files overlap (groups of ~8 MB)
    {
        size_t sent = 0;
        for (int i = 0; i < n; ++i) {
            const JdFile& f = F[(size_t)i];
            if (!f.valid) continue;
            const size_t padded = ((size_t)f.raw_len + 16 + 255) & ~(size_t)255;
            memcpy(raw + f.raw_off, scan_src[(size_t)i], f.raw_len);
            memset(raw + f.raw_off + f.raw_len, 0, padded - f.raw_len);
            const size_t end = (size_t)f.raw_off + padded;
            if (end - sent >= (8u << 20) || end == raw_total) {
                JDCHK(hipMemcpyAsync(dRaw + sent, raw + sent, end - sent, hipMemcpyHostToDevice, st));
                sent = end;
            }
        }
        if (sent < raw_total) JDCHK(hipMemcpyAsync(dRaw + sent, raw + sent, raw_total - sent, hipMemcpyHostToDevice, st));
    }
    JDCHK(hipEventRecord(eng->jd_stage_ev[sb], st));   // the staging buffer is free again once the uploads have run
    JDCHK(hipMemsetAsync(dClean, 0, raw_total + 64, st));
    JDCHK(hipMemsetAsync(dCoef, 0, blk_total * 128, st));
    // ---- 1. un-stuff ----
    hipLaunchKernelGGL(jd_mark_kernel, dim3(max_ublk, n), dim3(256), 0, st, dF, dRaw, dKeep, dRst);
    hipLaunchKernelGGL(jd_scan_kernel, dim3(n), dim3(256), 0, st, dF, dKeep, dRst, dD);
    hipLaunchKernelGGL(jd_compact_kernel, dim3(max_ublk, n), dim3(256), 0, st, dF, dRaw, dKeep, dRst, dD, dClean, dSeg);
    // ---- 2. synchronise the chunk decoders (Jacobi passes until nothing changes) ----
    std::vector<int> changed((size_t)n * (JD_PASSES + 1));
    const dim3 cgrid((max_chunks + 63) / 64, n);
    int pass = 0;
    if (async_passes > 0) {
        // host parse codes travel with the call (the descriptor upload above was from pageable memory: already consumed)
        JDCHK(hipMemcpyAsync(dHostRc, status, sizeof(int) * n, hipMemcpyHostToDevice, st));
        JDCHK(hipMemsetAsync(dChanged, 0, sizeof(int) * (size_t)n * (JD_PASSES + 1), st));
        for (; pass < async_passes; ++pass) {
            if (pass == async_passes - 1) JDCHK(hipMemsetAsync(dChanged, 0, sizeof(int) * n, st));
            hipLaunchKernelGGL(jd_sync_kernel, cgrid, dim3(64), 0, st, dF, dD, dClean, dSeg, C, pass, dChanged);
        }
        eng->jd_last_passes = pass;
        hipLaunchKernelGGL(jd_blkscan_kernel, dim3(n), dim3(256), 0, st, dF, dD, C);
        hipLaunchKernelGGL(jd_write_kernel, cgrid, dim3(64), 0, st, dF, dD, dClean, dSeg, C, dCoef);
        hipLaunchKernelGGL(jd_dc_kernel, dim3(3, n), dim3(256), 0, st, dF, dD, dCoef);
        hipLaunchKernelGGL(jd_idct_kernel, dim3((max_blk + 31) / 32, n), dim3(256), 0, st, dF, dD, dCoef, dPlane);
        hipLaunchKernelGGL(jd_color_kernel, dim3((width + 63) / 64, (height + 3) / 4, n), dim3(256), 0, st, dF, dD, dPlane, out_dev, height, width);
        hipLaunchKernelGGL(jd_status_kernel, dim3((n + 63) / 64), dim3(64), 0, st, dF, dD, dHostRc, dChanged, dStatus, n);
        JDCHK(hipMemcpyAsync(status, dStatus, sizeof(int) * n, hipMemcpyDeviceToHost, st));
        JDCHK(hipGetLastError());
        return 0;
    }
    for (;;) {
        JDCHK(hipMemsetAsync(dChanged, 0, sizeof(int) * changed.size(), st));
        for (int k = 0; k < JD_PASSES; ++k, ++pass) hipLaunchKernelGGL(jd_sync_kernel, cgrid, dim3(64), 0, st, dF, dD, dClean, dSeg, C, pass, dChanged + (size_t)k * n);
        JDCHK(hipMemcpyAsync(changed.data(), dChanged, sizeof(int) * changed.size(), hipMemcpyDeviceToHost, st));
        JDCHK(hipStreamSynchronize(st));
        bool again = false;
        for (int i = 0; i < n; ++i) again = again || changed[(size_t)(JD_PASSES - 1) * n + i] != 0;
        if (!again) break;
        if ((unsigned)pass > max_chunks + JD_PASSES) return locr_fail(eng, "jpeg_decode", "the chunk decoders did not reach a fixed point");
    }
    eng->jd_last_passes = pass;
    // ---- 3. coefficients, DC, IDCT, colour ----
    hipLaunchKernelGGL(jd_blkscan_kernel, dim3(n), dim3(256), 0, st, dF, dD, C);
    hipLaunchKernelGGL(jd_write_kernel, cgrid, dim3(64), 0, st, dF, dD, dClean, dSeg, C, dCoef);
    hipLaunchKernelGGL(jd_dc_kernel, dim3(3, n), dim3(256), 0, st, dF, dD, dCoef);
    hipLaunchKernelGGL(jd_idct_kernel, dim3((max_blk + 31) / 32, n), dim3(256), 0, st, dF, dD, dCoef, dPlane);
    hipLaunchKernelGGL(jd_color_kernel, dim3((width + 63) / 64, (height + 3) / 4, n), dim3(256), 0, st, dF, dD, dPlane, out_dev, height, width);
    std::vector<JdDyn> dyn((size_t)n);
    JDCHK(hipMemcpyAsync(dyn.data(), dD, sizeof(JdDyn) * n, hipMemcpyDeviceToHost, st));
    JDCHK(hipStreamSynchronize(st));
    JDCHK(hipGetLastError());
    for (int i = 0; i < n; ++i)
        if (F[(size_t)i].valid && (dyn[(size_t)i].err || dyn[(size_t)i].nseg < 0)) status[i] = -1;
    return 0;
}
