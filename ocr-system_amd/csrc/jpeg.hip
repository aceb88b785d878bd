// Baseline JPEG encoder on gfx950, byte-identical to the reference's Pillow call
//   image.save(buffer, format='JPEG', quality=q, optimize=True)        (/root/reference/backend/utils/image_preprocessing.py:526-538)
// i.e. libjpeg's integer pipeline: JFIF 1.01, YCbCr 4:2:0, 16-bit fixed-point colour conversion, 2x2 box down-sampling with
// the alternating 1,2 bias, edge replication inside partial blocks, zero-AC dummy blocks beyond them, "islow" forward DCT,
// round-half-away quantisation, OPTIMISED Huffman tables (T.81 K.2 with the IJG tie-breaking and 16-bit length limiting),
// byte stuffing.  Everything runs on the device, stream-ordered, with no host round trip:
//   1 jpeg_coef      one thread per 8x8 block: RGB -> component samples -> DCT -> quantise -> zig-zag int16        (HBM: 3 B/px in)
//   2 jpeg_dummy     right / bottom edge MCUs: DC of the dummy luma blocks
//   3 jpeg_stats     one thread per block: run-length symbols -> per-page histograms (LDS, then global atomics)
//   4 jpeg_tables    one wave per (page, table): optimal code lengths, length limiting, code assignment
//   5 jpeg_blockbits one thread per block: coded size in bits;  6 jpeg_scan: exclusive prefix sum per page
//   7 jpeg_emit      one thread per block: codes OR-ed into the zero-initialised bit stream at the block's bit offset
//   8 jpeg_finish    one workgroup per page: JFIF / DQT / SOF0 / DHT / SOS headers, 0xFF byte stuffing, EOI, file size
// Integer arithmetic only: the parallel result is exact by construction (tests/test_gpu_jpeg.py compares files byte for byte).
#include "jpeg.h"
#include "lds_rows.h"

namespace {

struct Geo { int w, h, mx, my, mcus, ybw, ybh, crows, nblk; };
struct QTab { uint16_t l[64], c[64]; };  // natural order
struct DevHT { uint32_t nval; uint8_t bits[17]; uint8_t vals[256]; uint16_t code[256]; uint8_t size[256]; };

__device__ const uint8_t d_ZZ[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                     41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                     15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
constexpr uint8_t STD_LUMA[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57,
                                  69, 56, 14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55,
                                  64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
constexpr uint8_t STD_CHROMA[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                    99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

// ---- colour conversion (16-bit fixed point; constants = (int)(x * 65536 + 0.5)) ----
__device__ __forceinline__ int ycc_y(int r, int g, int b) { return (19595 * r + 38470 * g + 7471 * b + 32768) >> 16; }
__device__ __forceinline__ int ycc_cb(int r, int g, int b) { return (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16; }
__device__ __forceinline__ int ycc_cr(int r, int g, int b) { return (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16; }

// ---- one 1-D pass of the "islow" DCT over 8 values (PASS 0: rows, result scaled by 4; PASS 1: columns, descaled) ----
#define JDESC(x, n) (((x) + (1 << ((n)-1))) >> (n))
template <int PASS>
__device__ __forceinline__ void dct8(int& d0, int& d1, int& d2, int& d3, int& d4, int& d5, int& d6, int& d7) {
    constexpr int SH = PASS == 0 ? 11 : 15;
    const int t0 = d0 + d7, t7 = d0 - d7, t1 = d1 + d6, t6 = d1 - d6, t2 = d2 + d5, t5 = d2 - d5, t3 = d3 + d4, t4 = d3 - d4;
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    if (PASS == 0) { d0 = (t10 + t11) << 2; d4 = (t10 - t11) << 2; }
    else { d0 = JDESC(t10 + t11, 2); d4 = JDESC(t10 - t11, 2); }
    int z1 = (t12 + t13) * 4433;
    d2 = JDESC(z1 + t13 * 6270, SH);
    d6 = JDESC(z1 + t12 * (-15137), SH);
    z1 = t4 + t7;
    int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
    const int z5 = (z3 + z4) * 9633;
    const int a4 = t4 * 2446, a5 = t5 * 16819, a6 = t6 * 25172, a7 = t7 * 12299;
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5; z4 += z5;
    d7 = JDESC(a4 + z1 + z3, SH);
    d5 = JDESC(a5 + z2 + z4, SH);
    d3 = JDESC(a6 + z2 + z3, SH);
    d1 = JDESC(a7 + z1 + z4, SH);
}

// ---- 1: samples -> DCT -> quantise -> zig-zag.  One workgroup = 16 MCUs of one MCU row (256 x 16 pixels) ----
//   a) the RGB rows go to LDS with aligned word loads (lds_rows.h);  b) all 256 threads convert: Y for every pixel, Cb / Cr for
//   every 2x2 quad (edge replication = clamped tile coordinates), into planar u8 tiles;  c) 96 threads (16 MCUs x 6 blocks)
//   read their 8x8 samples as 8-byte rows, run the DCT in registers, quantise and store the zig-zag block.
constexpr int JT_PX = 256, JT_PITCH = (JT_PX * 3 + 3) / 4 + 2;
__global__ __launch_bounds__(256) void jpeg_coef_kernel(const uint8_t* rgb, int n, Geo g, QTab q, int16_t* coefs) {
    __shared__ uint32_t tile[16 * JT_PITCH];
    __shared__ __attribute__((aligned(8))) uint8_t Ys[16][JT_PX];
    __shared__ __attribute__((aligned(8))) uint8_t Cs[2][8][JT_PX / 2];
    const int tid = threadIdx.x, page = blockIdx.z, mcy = blockIdx.y;
    const int x0 = blockIdx.x * JT_PX, y0 = mcy * 16;
    const int rows_valid = min(16, g.h - y0), cols_valid = min(JT_PX, g.w - x0);
    const long long g0 = (((long long)page * g.h + y0) * g.w + x0) * 3;
    fill_rows(tile, JT_PITCH, (cols_valid * 3 + 3 + 3) / 4, rgb, g0, (long long)g.w * 3, rows_valid, 0, rows_valid, (long long)n * g.h * g.w * 3,
              WordIdentity());
    __syncthreads();
    const uintptr_t ibase = reinterpret_cast<uintptr_t>(rgb);
    const uint8_t* tb = reinterpret_cast<const uint8_t*>(tile);
    const int m0 = (int)((ibase + (unsigned long long)g0) & 3), dm = (g.w * 3) & 3;
#define JPX(r_, c_) (tb + (r_) * (JT_PITCH * 4) + ((m0 + (r_) * dm) & 3) + (c_) * 3)
#pragma unroll
    for (int k = 0; k < 16; ++k) {   // luma: rows / columns past the image repeat its last row / column
        const int r = k, c = tid;
        const uint8_t* px = JPX(min(r, rows_valid - 1), min(c, cols_valid - 1));
        Ys[r][c] = (uint8_t)ycc_y(px[0], px[1], px[2]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {    // chroma: rows past the last down-sampled row repeat IT; columns are replicated at full resolution
        const int qd = tid + 256 * k, yc = qd >> 7, xc = qd & 127;
        const int ycl = min((y0 >> 1) + yc, g.crows - 1) - (y0 >> 1);
        const int r0 = 2 * ycl, r1 = min(2 * ycl + 1, rows_valid - 1);
        const int c0 = min(2 * xc, cols_valid - 1), c1 = min(2 * xc + 1, cols_valid - 1);
        const uint8_t *p00 = JPX(r0, c0), *p01 = JPX(r0, c1), *p10 = JPX(r1, c0), *p11 = JPX(r1, c1);
        const int bias = (xc & 1) ? 2 : 1;
        Cs[0][yc][xc] = (uint8_t)((ycc_cb(p00[0], p00[1], p00[2]) + ycc_cb(p01[0], p01[1], p01[2]) + ycc_cb(p10[0], p10[1], p10[2]) + ycc_cb(p11[0], p11[1], p11[2]) + bias) >> 2);
        Cs[1][yc][xc] = (uint8_t)((ycc_cr(p00[0], p00[1], p00[2]) + ycc_cr(p01[0], p01[1], p01[2]) + ycc_cr(p10[0], p10[1], p10[2]) + ycc_cr(p11[0], p11[1], p11[2]) + bias) >> 2);
    }
#undef JPX
    __syncthreads();
    if (tid >= 96) return;
    const int ml = tid / 6, b = tid - ml * 6, mcx = blockIdx.x * 16 + ml;
    if (mcx >= g.mx) return;
    const size_t blk = ((size_t)page * g.mcus + (size_t)mcy * g.mx + mcx) * 6 + b;
    uint4* dst = reinterpret_cast<uint4*>(coefs + blk * 64);
    int d[64];
    if (b < 4) {
        const int bx = 2 * mcx + (b & 1), by = 2 * mcy + (b >> 1);
        if (bx >= g.ybw || by >= g.ybh) {  // dummy block: zero AC, DC resolved by jpeg_dummy_kernel
#pragma unroll
            for (int i = 0; i < 8; ++i) dst[i] = make_uint4(0, 0, 0, 0);
            return;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint2 v = *reinterpret_cast<const uint2*>(&Ys[(b >> 1) * 8 + r][(2 * ml + (b & 1)) * 8]);
#pragma unroll
            for (int c = 0; c < 8; ++c) d[r * 8 + c] = (int)(((c < 4 ? v.x : v.y) >> (8 * (c & 3))) & 0xffu) - 128;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint2 v = *reinterpret_cast<const uint2*>(&Cs[b - 4][r][ml * 8]);
#pragma unroll
            for (int c = 0; c < 8; ++c) d[r * 8 + c] = (int)(((c < 4 ? v.x : v.y) >> (8 * (c & 3))) & 0xffu) - 128;
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) dct8<0>(d[r * 8], d[r * 8 + 1], d[r * 8 + 2], d[r * 8 + 3], d[r * 8 + 4], d[r * 8 + 5], d[r * 8 + 6], d[r * 8 + 7]);
#pragma unroll
    for (int c = 0; c < 8; ++c) dct8<1>(d[c], d[8 + c], d[16 + c], d[24 + c], d[32 + c], d[40 + c], d[48 + c], d[56 + c]);
    const uint16_t* qt = b < 4 ? q.l : q.c;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const int qv = (int)qt[i] << 3;
        int v = d[i];
        if (v < 0) { v = -v + (qv >> 1); v = v >= qv ? v / qv : 0; v = -v; }
        else { v += qv >> 1; v = v >= qv ? v / qv : 0; }
        d[i] = v;
    }
    constexpr int ZZ[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                            41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                            15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t wv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[j] = ((uint32_t)d[ZZ[8 * i + 2 * j]] & 0xffffu) | ((uint32_t)d[ZZ[8 * i + 2 * j + 1]] << 16);
        dst[i] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    }
}

// ---- 2: DC of dummy luma blocks (right edge: the block to the left; bottom dummy row: the last block of the row above) ----
__global__ void jpeg_dummy_kernel(int n, Geo g, int16_t* coefs) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * g.mcus) return;
    const int page = t / g.mcus, m = t - page * g.mcus, mcx = m % g.mx, mcy = m / g.mx;
    if (mcx != g.mx - 1 && mcy != g.my - 1) return;
    int16_t* base = coefs + ((size_t)page * g.nblk + (size_t)m * 6) * 64;
    for (int b = 1; b < 4; ++b) {
        const int v = b >> 1, bx = 2 * mcx + (b & 1), by = 2 * mcy + v;
        if (bx < g.ybw && by < g.ybh) continue;
        base[b * 64] = base[((by < g.ybh) ? b - 1 : v * 2 - 1) * 64];
    }
}

__device__ __forceinline__ int nbits_of(int v) { v = v < 0 ? -v : v; return v ? 32 - __clz(v) : 0; }
__device__ __forceinline__ int prev_dc(const int16_t* pc, int m, int b) {  // pc = the page's coefficients
    if (b > 0 && b < 4) return pc[((size_t)m * 6 + b - 1) * 64];
    if (m == 0) return 0;
    return pc[((size_t)(m - 1) * 6 + (b == 0 ? 3 : b)) * 64];
}

// ---- 3: symbol statistics: hist[page][4][256]  (0 DC luma, 1 AC luma, 2 DC chroma, 3 AC chroma) ----
__global__ __launch_bounds__(256) void jpeg_stats_kernel(const int16_t* coefs, Geo g, uint32_t* hist) {
    __shared__ uint32_t lh[4 * 256];
    const int page = blockIdx.y, blk = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < 1024; i += 256) lh[i] = 0;
    __syncthreads();
    if (blk < g.nblk) {
        const int16_t* pc = coefs + (size_t)page * g.nblk * 64;
        const int m = blk / 6, b = blk - m * 6, tb = b < 4 ? 0 : 2;
        __attribute__((aligned(16))) int16_t c[64];
        const uint4* src = reinterpret_cast<const uint4*>(pc + (size_t)blk * 64);
#pragma unroll
        for (int i = 0; i < 8; ++i) reinterpret_cast<uint4*>(c)[i] = src[i];
        atomicAdd(&lh[tb * 256 + nbits_of((int)c[0] - prev_dc(pc, m, b))], 1u);
        int r = 0;
#pragma unroll
        for (int k = 1; k < 64; ++k) {
            const int v = c[k];
            if (v == 0) { ++r; continue; }
            while (r > 15) { atomicAdd(&lh[(tb + 1) * 256 + 0xF0], 1u); r -= 16; }
            atomicAdd(&lh[(tb + 1) * 256 + (r << 4) + nbits_of(v)], 1u);
            r = 0;
        }
        if (r > 0) atomicAdd(&lh[(tb + 1) * 256], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256)
        if (lh[i]) atomicAdd(&hist[(size_t)page * 1024 + i], lh[i]);
}

// ---- 4: optimal Huffman table, one wave per (page, table) ----
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
    const unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffull), m), hi = __shfl_xor((unsigned)(v >> 32), m);
    return ((unsigned long long)hi << 32) | lo;
}
__global__ __launch_bounds__(64) void jpeg_tables_kernel(const uint32_t* hist, DevHT* tabs) {
    // libjpeg's jpeg_gen_optimal_table: up to 256 rounds of "merge the two least frequent trees" (ties -> the LARGER symbol), each
    // bumping the code size of every member of both trees by walking their `others` chains.  Here the 257 frequencies, code sizes and
    // tree labels (a symbol remembers the first symbol of its tree) live in registers, 5 per lane; a round is one butterfly that
    // carries the two smallest keys at once, and the bump / re-label of the members is a compare per register: the same
    // increments, no LDS round trip in the loop (the serial version took 0.65 ms per table, this one ~0.15).
    __shared__ int codesize[257], bits[33];
    const int t = blockIdx.x, lane = threadIdx.x;
    int f[5], cs[5], tr[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int i = lane + 64 * j;
        f[j] = i < 256 ? (int)hist[(size_t)t * 256 + i] : (i == 256 ? 1 : 0);
        cs[j] = 0; tr[j] = i;
    }
    if (lane < 33) bits[lane] = 0;
    const unsigned long long NONE = ~0ull;
    for (;;) {
        // key = freq << 16 | (0xFFFF - symbol): the minimum is the smallest non-zero frequency, ties -> the larger symbol
        unsigned long long a1 = NONE, a2 = NONE;
#pragma unroll
        for (int j = 0; j < 5; ++j)
            if (f[j]) {
                const unsigned long long k = ((unsigned long long)f[j] << 16) | (unsigned)(0xFFFF - (lane + 64 * j));
                if (k < a1) { a2 = a1; a1 = k; } else if (k < a2) a2 = k;
            }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {   // the keys are distinct and the two halves of a butterfly step hold disjoint sets
            const unsigned long long b1 = shfl_xor_u64(a1, m), b2 = shfl_xor_u64(a2, m);
            const unsigned long long lo = a1 < b1 ? a1 : b1, hi = a1 < b1 ? b1 : a1, s2 = a2 < b2 ? a2 : b2;
            a1 = lo; a2 = hi < s2 ? hi : s2;
        }
        if (a2 == NONE) break;
        const int c1 = 0xFFFF - (int)(a1 & 0xFFFF), c2 = 0xFFFF - (int)(a2 & 0xFFFF), f2 = (int)(a2 >> 16);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int i = lane + 64 * j;
            if (i == c1) f[j] += f2;
            if (i == c2) f[j] = 0;
            if (tr[j] == c1 || tr[j] == c2) { cs[j]++; tr[j] = c1; }
        }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) if (lane + 64 * j < 257) codesize[lane + 64 * j] = cs[j];
    __syncthreads();
    // The rest of jpeg_gen_optimal_table + the canonical code assignment, with only the 16-step length bookkeeping left to one lane
    // (the serial "for every length, for every symbol" sweeps were 0.4 ms per table).
    __shared__ int first_k[18], code_start[18];
#pragma unroll
    for (int j = 0; j < 5; ++j) if (lane + 64 * j < 257 && cs[j]) atomicAdd(&bits[cs[j] > 32 ? 32 : cs[j]], 1);
    __syncthreads();
    DevHT* T = tabs + t;
    if (lane == 0) {
        for (int i = 32; i > 16; --i)
            while (bits[i] > 0) {
                int j = i - 2;
                while (bits[j] == 0) --j;
                bits[i] -= 2; bits[i - 1]++; bits[j + 1] += 2; bits[j]--;
            }
        int i = 16;
        while (i > 0 && bits[i] == 0) --i;
        if (i > 0) bits[i]--;  // the pseudo-symbol 256 reserved the all-ones code
        T->bits[0] = 0;
        int k = 0, code = 0;
        for (int len = 1; len <= 16; ++len) {
            T->bits[len] = (uint8_t)bits[len];
            first_k[len] = k; code_start[len] = code;
            k += bits[len]; code = (code + bits[len]) << 1;
        }
        first_k[17] = k;
    }
    __syncthreads();
    // huffval = the symbols 0..255 in order of (code size, symbol): a symbol's position is the number of symbols before it
    int nval = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int sym = lane + 64 * j, mine = cs[j];
        int rank = 0;
        for (int o = 0; o < 256; ++o) {
            const int oc = codesize[o];
            rank += (oc != 0 && (oc < mine || (oc == mine && o < sym))) ? 1 : 0;
            if (j == 0 && lane == 0) nval += oc != 0;
        }
        uint16_t code = 0; uint8_t size = 0;
        if (mine) {
            T->vals[rank] = (uint8_t)sym;
            for (int len = 1; len <= 16; ++len)
                if (rank >= first_k[len] && rank < first_k[len + 1]) { size = (uint8_t)len; code = (uint16_t)(code_start[len] + rank - first_k[len]); }
        }
        T->size[sym] = size; T->code[sym] = code;
    }
    if (lane == 0) T->nval = (uint32_t)nval;
}

// ---- 4': the ITU-T T.81 Annex K.3 typical tables (optimize == 0: what libjpeg writes without the statistics pass) ----
__device__ const uint8_t d_STD_BITS_DC_L[17] = {
    0x00, 0x00, 0x01, 0x05, 0x01, 0x01, 0x01, 0x01, 0x01, 0x01, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00,
};
__device__ const uint8_t d_STD_VALS_DC_L[12] = {
    0x00, 0x01, 0x02, 0x03, 0x04, 0x05, 0x06, 0x07, 0x08, 0x09, 0x0a, 0x0b,
};
__device__ const uint8_t d_STD_BITS_AC_L[17] = {
    0x00, 0x00, 0x02, 0x01, 0x03, 0x03, 0x02, 0x04, 0x03, 0x05, 0x05, 0x04, 0x04, 0x00, 0x00, 0x01, 0x7d,
};
__device__ const uint8_t d_STD_VALS_AC_L[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08,
    0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
    0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
    0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
    0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa,
};
__device__ const uint8_t d_STD_BITS_DC_C[17] = {
    0x00, 0x00, 0x03, 0x01, 0x01, 0x01, 0x01, 0x01, 0x01, 0x01, 0x01, 0x01, 0x00, 0x00, 0x00, 0x00, 0x00,
};
__device__ const uint8_t d_STD_VALS_DC_C[12] = {
    0x00, 0x01, 0x02, 0x03, 0x04, 0x05, 0x06, 0x07, 0x08, 0x09, 0x0a, 0x0b,
};
__device__ const uint8_t d_STD_BITS_AC_C[17] = {
    0x00, 0x00, 0x02, 0x01, 0x02, 0x04, 0x04, 0x03, 0x04, 0x07, 0x05, 0x04, 0x04, 0x00, 0x01, 0x02, 0x77,
};
__device__ const uint8_t d_STD_VALS_AC_C[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91,
    0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
    0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
    0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa,
};
__global__ __launch_bounds__(64) void jpeg_std_tables_kernel(DevHT* tabs) {
    if (threadIdx.x != 0) return;
    const int t = blockIdx.x & 3;   // 0 DC luma, 1 AC luma, 2 DC chroma, 3 AC chroma
    const uint8_t* bits = t == 0 ? d_STD_BITS_DC_L : (t == 1 ? d_STD_BITS_AC_L : (t == 2 ? d_STD_BITS_DC_C : d_STD_BITS_AC_C));
    const uint8_t* vals = t == 0 ? d_STD_VALS_DC_L : (t == 1 ? d_STD_VALS_AC_L : (t == 2 ? d_STD_VALS_DC_C : d_STD_VALS_AC_C));
    const int nv = (t & 1) ? 162 : 12;
    DevHT* T = tabs + blockIdx.x;
    for (int k = 0; k <= 16; ++k) T->bits[k] = bits[k];
    for (int k = 0; k < nv; ++k) T->vals[k] = vals[k];
    T->nval = (uint32_t)nv;
    for (int s2 = 0; s2 < 256; ++s2) { T->size[s2] = 0; T->code[s2] = 0; }
    int k = 0, code = 0;
    for (int len = 1; len <= 16; ++len) {
        for (int j = 0; j < bits[len]; ++j, ++k) { T->code[vals[k]] = (uint16_t)code++; T->size[vals[k]] = (uint8_t)len; }
        code <<= 1;
    }
}

// ---- 5: coded size of every block in bits ----
__global__ __launch_bounds__(256) void jpeg_blockbits_kernel(const int16_t* coefs, Geo g, const DevHT* tabs, uint32_t* blkbits) {
    __shared__ uint8_t sz[4 * 256];
    const int page = blockIdx.y, blk = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < 1024; i += 256) sz[i] = tabs[(size_t)page * 4 + (i >> 8)].size[i & 255];
    __syncthreads();
    if (blk >= g.nblk) return;
    const int16_t* pc = coefs + (size_t)page * g.nblk * 64;
    const int m = blk / 6, b = blk - m * 6, tb = b < 4 ? 0 : 2;
    __attribute__((aligned(16))) int16_t c[64];
    const uint4* src = reinterpret_cast<const uint4*>(pc + (size_t)blk * 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) reinterpret_cast<uint4*>(c)[i] = src[i];
    int nb = nbits_of((int)c[0] - prev_dc(pc, m, b));
    uint32_t total = sz[tb * 256 + nb] + nb;
    int r = 0;
#pragma unroll
    for (int k = 1; k < 64; ++k) {
        const int v = c[k];
        if (v == 0) { ++r; continue; }
        while (r > 15) { total += sz[(tb + 1) * 256 + 0xF0]; r -= 16; }
        nb = nbits_of(v);
        total += sz[(tb + 1) * 256 + (r << 4) + nb] + nb;
        r = 0;
    }
    if (r > 0) total += sz[(tb + 1) * 256];
    blkbits[(size_t)page * g.nblk + blk] = total;
}

// ---- 6: exclusive prefix sum of the block sizes of a page (one workgroup per page) -> bit offsets, total bits ----
__global__ __launch_bounds__(256) void jpeg_scan_kernel(uint32_t* blkbits, Geo g, unsigned long long* totalbits) {
    __shared__ uint32_t wsum[4];
    __shared__ unsigned long long run;
    const int page = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* a = blkbits + (size_t)page * g.nblk;
    if (tid == 0) run = 0;
    __syncthreads();
    constexpr int PER = 8;   // blocks per thread and round: three barriers per 2048 blocks instead of per 256
    for (int i0 = 0; i0 < g.nblk; i0 += 256 * PER) {
        const int ib = i0 + tid * PER;
        uint32_t v[PER], tot = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { v[k] = ib + k < g.nblk ? a[ib + k] : 0; tot += v[k]; }
        uint32_t inc = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int w2 = 0; w2 < wave; ++w2) woff += wsum[w2];
        const unsigned long long base = run;
        uint32_t off = (uint32_t)(base + woff + inc - tot);   // offsets fit 32 bits (checked against the raw capacity in finish)
#pragma unroll
        for (int k = 0; k < PER; ++k) { if (ib + k < g.nblk) a[ib + k] = off; off += v[k]; }
        __syncthreads();
        if (tid == 0) run = base + wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (tid == 0) totalbits[page] = run;
}

// ---- 7: entropy-coded bits, OR-ed into the zeroed stream (big-endian bit order inside big-endian 32-bit words) ----
// A block's bits are contiguous, so only its first and last 32-bit words are shared with the neighbouring blocks: those two are
// OR-ed atomically into the zeroed stream, every word in between is owned by this thread and stored plainly.
struct BitOut {
    uint32_t* words; unsigned long long pos; unsigned long long cap_bits;
    unsigned long long acc; int nacc; bool first;   // acc: pending bits, left-aligned at bit 63; nacc counts them (incl. the lead-in)
    __device__ __forceinline__ void begin() { nacc = (int)(pos & 31); acc = 0; first = true; pos &= ~31ull; }
    __device__ __forceinline__ void flush_word() {  // emit the top 32 bits of acc at word pos/32
        if (pos + 32 <= cap_bits) {
            uint32_t* w = words + (pos >> 5);
            const uint32_t v = (uint32_t)(acc >> 32);
            if (first) { if (v) atomicOr(w, v); first = false; }
            else *w = v;
        }
        acc <<= 32; nacc -= 32; pos += 32;
    }
    __device__ __forceinline__ void put(uint32_t code, int size) {
        if (size == 0) return;
        acc |= ((unsigned long long)(code & ((1u << size) - 1u))) << (64 - size - nacc);
        nacc += size;
        if (nacc >= 32) flush_word();
    }
    __device__ __forceinline__ void end() {
        if (nacc > 0 && pos + 32 <= cap_bits) {
            const uint32_t v = (uint32_t)(acc >> 32);
            if (v) atomicOr(words + (pos >> 5), v);
        }
    }
};
__global__ __launch_bounds__(256) void jpeg_emit_kernel(const int16_t* coefs, Geo g, const DevHT* tabs, const uint32_t* bitoff, uint32_t* raw,
                                                        size_t raw_words_per_page) {
    __shared__ uint8_t sz[4 * 256];
    __shared__ uint16_t cd[4 * 256];
    const int page = blockIdx.y, blk = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < 1024; i += 256) { const DevHT& T = tabs[(size_t)page * 4 + (i >> 8)]; sz[i] = T.size[i & 255]; cd[i] = T.code[i & 255]; }
    __syncthreads();
    if (blk >= g.nblk) return;
    const int16_t* pc = coefs + (size_t)page * g.nblk * 64;
    const int m = blk / 6, b = blk - m * 6, tb = b < 4 ? 0 : 2;
    __attribute__((aligned(16))) int16_t c[64];
    const uint4* src = reinterpret_cast<const uint4*>(pc + (size_t)blk * 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) reinterpret_cast<uint4*>(c)[i] = src[i];
    BitOut bo{raw + (size_t)page * raw_words_per_page, bitoff[(size_t)page * g.nblk + blk], (unsigned long long)raw_words_per_page * 32 - 32, 0, 0, true};
    bo.begin();
    const int dcd = (int)c[0] - prev_dc(pc, m, b);
    int nb = nbits_of(dcd);
    bo.put(((uint32_t)cd[tb * 256 + nb] << nb) | ((uint32_t)(dcd < 0 ? dcd - 1 : dcd) & ((1u << nb) - 1u)), sz[tb * 256 + nb] + nb);
    int r = 0;
#pragma unroll
    for (int k = 1; k < 64; ++k) {
        const int v = c[k];
        if (v == 0) { ++r; continue; }
        while (r > 15) { bo.put(cd[(tb + 1) * 256 + 0xF0], sz[(tb + 1) * 256 + 0xF0]); r -= 16; }
        nb = nbits_of(v);
        const int s = (r << 4) + nb;
        bo.put(((uint32_t)cd[(tb + 1) * 256 + s] << nb) | ((uint32_t)(v < 0 ? v - 1 : v) & ((1u << nb) - 1u)), sz[(tb + 1) * 256 + s] + nb);
        r = 0;
    }
    if (r > 0) bo.put(cd[(tb + 1) * 256], sz[(tb + 1) * 256]);
    bo.end();
}

// ---- 8: headers + byte stuffing + EOI -> the finished file ----
// Every 0xFF byte of the entropy-coded stream is followed by a stuffed 0x00, so a byte's place in the file depends on the 0xFF bytes
// before it.  Three steps (one work-group per page walking the stream with three barriers per 4 KB took 0.6 ms per page, whatever
// the batch): 8a counts the 0xFF bytes of every 4 KB piece, 8b writes the headers, scans the piece counts and closes the file,
// 8c moves the pieces to their places.
constexpr int FIN_PIECE = 256 * 16;
// the 16 stream bytes of thread-slot kb (the last byte of the stream is padded with ones) -> by[], returns how many are 0xFF
__device__ __forceinline__ uint32_t stream_bytes16(const uint32_t* rw, unsigned long long kb, unsigned long long nbytes, unsigned long long tb, uint8_t* by) {
    uint32_t nff = 0;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) {
        const unsigned long long wi = (kb >> 2) + wv;
        const uint32_t w = (wi * 4 < nbytes) ? rw[wi] : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned long long k = kb + wv * 4 + j;
            uint32_t bv = (w >> (24 - 8 * j)) & 0xffu;
            if (k == nbytes - 1 && (tb & 7)) bv |= 0xffu >> (tb & 7);  // pad the last byte with ones
            by[wv * 4 + j] = (uint8_t)bv;
            if (k < nbytes && bv == 0xffu) ++nff;
        }
    }
    return nff;
}
__global__ __launch_bounds__(256) void jpeg_ffcount_kernel(const uint32_t* raw, size_t raw_words_per_page, const unsigned long long* totalbits, uint32_t* ffcnt,
                                                           int pieces) {
    __shared__ uint32_t wsum[4];
    const int page = blockIdx.y, piece = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long tb = totalbits[page], nbytes = (tb + 7) >> 3;
    const unsigned long long kb = (unsigned long long)piece * FIN_PIECE + (unsigned long long)tid * 16;
    uint32_t nff = 0;
    if (kb < nbytes && nbytes + 4 <= raw_words_per_page * 4) {
        uint8_t by[16];
        nff = stream_bytes16(raw + (size_t)page * raw_words_per_page, kb, nbytes, tb, by);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) nff += __shfl_xor(nff, m);
    if (lane == 0) wsum[wave] = nff;
    __syncthreads();
    if (tid == 0) ffcnt[(size_t)page * pieces + piece] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(64) void jpeg_head_kernel(const unsigned long long* totalbits, size_t raw_words_per_page, const DevHT* tabs, Geo g, QTab q, uint32_t* ffcnt,
                                                       int pieces, int* hdr_len, uint8_t* out, size_t out_stride, int32_t* sizes) {
    const int page = blockIdx.x, lane = threadIdx.x;
    uint8_t* o = out + (size_t)page * out_stride;
    const DevHT* T = tabs + (size_t)page * 4;
    const unsigned long long tb = totalbits[page], nbytes = (tb + 7) >> 3;
    const bool overflow = nbytes + 4 > raw_words_per_page * 4;
    // piece counts -> exclusive offsets (in place), 64 pieces per step
    uint32_t* fc = ffcnt + (size_t)page * pieces;
    uint32_t run = 0;
    for (int p0 = 0; p0 < pieces; p0 += 64) {
        const int pi = p0 + lane;
        const uint32_t v = pi < pieces ? fc[pi] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t ov = __shfl_up(inc, d); if (lane >= d) inc += ov; }
        if (pi < pieces) fc[pi] = run + inc - v;
        run += __shfl(inc, 63);
    }
    if (lane != 0) return;
    size_t p = 0;
    auto put = [&](int v) { if (p < out_stride) o[p] = (uint8_t)v; ++p; };
    auto put16 = [&](int v) { put(v >> 8); put(v & 0xff); };
    put(0xFF); put(0xD8);
    put(0xFF); put(0xE0); put16(16);
    const uint8_t jf[14] = {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
    for (int i = 0; i < 14; ++i) put(jf[i]);
    for (int t = 0; t < 2; ++t) {
        put(0xFF); put(0xDB); put16(67); put(t);
        for (int i = 0; i < 64; ++i) put((t ? q.c : q.l)[d_ZZ[i]]);
    }
    put(0xFF); put(0xC0); put16(17); put(8); put16(g.h); put16(g.w); put(3);
    put(1); put(0x22); put(0); put(2); put(0x11); put(1); put(3); put(0x11); put(1);
    for (int t = 0; t < 2; ++t)
        for (int a = 0; a < 2; ++a) {
            const DevHT& H = T[t * 2 + a];
            put(0xFF); put(0xC4); put16(2 + 1 + 16 + (int)H.nval); put((a << 4) | t);
            for (int i = 1; i <= 16; ++i) put(H.bits[i]);
            for (uint32_t i = 0; i < H.nval; ++i) put(H.vals[i]);
        }
    put(0xFF); put(0xDA); put16(12); put(3); put(1); put(0x00); put(2); put(0x11); put(3); put(0x11); put(0); put(63); put(0);
    const size_t hl = p;
    hdr_len[page] = (int)hl;
    const unsigned long long stuffed = overflow ? 0ull : (unsigned long long)run;
    const unsigned long long total = hl + nbytes + stuffed + 2;
    p = hl + (size_t)nbytes + (size_t)stuffed;
    if (p < out_stride) o[p] = 0xFF;
    if (p + 1 < out_stride) o[p + 1] = 0xD9;
    sizes[page] = (!overflow && total <= out_stride && total < 0x7fffffffull) ? (int32_t)total : -(int32_t)(total < 0x7fffffffull ? total : 0x7fffffff);
}
__global__ __launch_bounds__(256) void jpeg_stuff_kernel(const uint32_t* raw, size_t raw_words_per_page, const unsigned long long* totalbits, const uint32_t* ffoff,
                                                         int pieces, const int* hdr_len, uint8_t* out, size_t out_stride) {
    __shared__ uint32_t wsum[4];
    const int page = blockIdx.y, piece = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long tb = totalbits[page], nbytes = (tb + 7) >> 3;
    const unsigned long long k0 = (unsigned long long)piece * FIN_PIECE, kb = k0 + (unsigned long long)tid * 16;
    if (k0 >= nbytes || nbytes + 4 > raw_words_per_page * 4) return;   // (work-group-uniform)
    uint8_t by[16];
    const uint32_t nff = stream_bytes16(raw + (size_t)page * raw_words_per_page, kb, nbytes, tb, by);
    uint32_t inc = nff;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t ov = __shfl_up(inc, d); if (lane >= d) inc += ov; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w2 = 0; w2 < wave; ++w2) woff += wsum[w2];
    uint8_t* o = out + (size_t)page * out_stride;
    size_t p = (size_t)hdr_len[page] + (size_t)kb + (size_t)ffoff[(size_t)page * pieces + piece] + (size_t)(woff + inc - nff);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (kb + j >= nbytes) break;
        if (p < out_stride) o[p] = by[j];
        ++p;
        if (by[j] == 0xff) { if (p < out_stride) o[p] = 0; ++p; }
    }
}

Geo make_geo(int height, int width) {
    Geo g;
    g.w = width; g.h = height; g.mx = (width + 15) / 16; g.my = (height + 15) / 16; g.mcus = g.mx * g.my;
    g.ybw = (width + 7) / 8; g.ybh = (height + 7) / 8; g.crows = (height + 1) / 2; g.nblk = g.mcus * 6;
    return g;
}
QTab make_qtab(int quality) {
    QTab q;
    if (quality <= 0) quality = 1;
    if (quality > 100) quality = 100;
    const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    for (int i = 0; i < 64; ++i) {
        const long a = ((long)STD_LUMA[i] * scale + 50L) / 100L, b = ((long)STD_CHROMA[i] * scale + 50L) / 100L;
        q.l[i] = (uint16_t)(a <= 0 ? 1 : (a > 255 ? 255 : a));
        q.c[i] = (uint16_t)(b <= 0 ? 1 : (b > 255 ? 255 : b));
    }
    return q;
}
inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

// workspace: coefs | hist | tables | blkbits | totalbits | raw
size_t jpeg_workspace_bytes(int n, int height, int width) {
    const Geo g = make_geo(height, width);
    const size_t coef = al256((size_t)n * g.nblk * 64 * 2), hist = al256((size_t)n * 1024 * 4), tabs = al256((size_t)n * 4 * sizeof(DevHT));
    const size_t bb = al256((size_t)n * g.nblk * 4), tbits = al256((size_t)n * 8), raw = al256((size_t)n * ((size_t)g.nblk * 128 + 64));
    const size_t pieces = ((size_t)g.nblk * 128 + 64 + FIN_PIECE - 1) / FIN_PIECE;
    return coef + hist + tabs + bb + tbits + raw + al256((size_t)n * pieces * 4) + al256((size_t)n * 4);
}

hipError_t jpeg_coefficients_launch(const uint8_t* rgb, int n, int height, int width, int quality, int16_t* coefs, hipStream_t st) {
    const Geo g = make_geo(height, width);
    const QTab q = make_qtab(quality);
    hipLaunchKernelGGL(jpeg_coef_kernel, dim3((g.mx + 15) / 16, g.my, n), dim3(256), 0, st, rgb, n, g, q, coefs);
    hipLaunchKernelGGL(jpeg_dummy_kernel, dim3((n * g.mcus + 255) / 256), dim3(256), 0, st, n, g, coefs);
    return hipGetLastError();
}

hipError_t jpeg_encode_launch(const JpegParams& p, void* workspace, hipStream_t st) {
    if (p.n <= 0 || p.height <= 0 || p.width <= 0 || p.height > 65535 || p.width > 65535) return hipErrorInvalidValue;
    const Geo g = make_geo(p.height, p.width);
    const QTab q = make_qtab(p.quality);
    uint8_t* w = static_cast<uint8_t*>(workspace);
    auto take = [&](size_t bytes) { void* r = w; w += al256(bytes); return r; };
    int16_t* coefs = static_cast<int16_t*>(take((size_t)p.n * g.nblk * 64 * 2));
    uint32_t* hist = static_cast<uint32_t*>(take((size_t)p.n * 1024 * 4));
    DevHT* tabs = static_cast<DevHT*>(take((size_t)p.n * 4 * sizeof(DevHT)));
    uint32_t* blkbits = static_cast<uint32_t*>(take((size_t)p.n * g.nblk * 4));
    unsigned long long* totalbits = static_cast<unsigned long long*>(take((size_t)p.n * 8));
    const size_t raw_words = ((size_t)g.nblk * 128 + 64) / 4;
    uint32_t* raw = static_cast<uint32_t*>(take((size_t)p.n * raw_words * 4));
    const int pieces = (int)((raw_words * 4 + FIN_PIECE - 1) / FIN_PIECE);
    uint32_t* ffcnt = static_cast<uint32_t*>(take((size_t)p.n * pieces * 4));
    int* hdrlen = static_cast<int*>(take((size_t)p.n * 4));
    hipError_t e = hipMemsetAsync(hist, 0, (size_t)p.n * 1024 * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(raw, 0, (size_t)p.n * raw_words * 4, st);
    if (e != hipSuccess) return e;
    e = jpeg_coefficients_launch(p.rgb, p.n, p.height, p.width, p.quality, coefs, st);
    if (e != hipSuccess) return e;
    const dim3 gb((g.nblk + 255) / 256, p.n);
    if (p.optimize) {
        hipLaunchKernelGGL(jpeg_stats_kernel, gb, dim3(256), 0, st, coefs, g, hist);
        hipLaunchKernelGGL(jpeg_tables_kernel, dim3(p.n * 4), dim3(64), 0, st, hist, tabs);
    } else {
        hipLaunchKernelGGL(jpeg_std_tables_kernel, dim3(p.n * 4), dim3(64), 0, st, tabs);
    }
    hipLaunchKernelGGL(jpeg_blockbits_kernel, gb, dim3(256), 0, st, coefs, g, tabs, blkbits);
    hipLaunchKernelGGL(jpeg_scan_kernel, dim3(p.n), dim3(256), 0, st, blkbits, g, totalbits);
    hipLaunchKernelGGL(jpeg_emit_kernel, gb, dim3(256), 0, st, coefs, g, tabs, blkbits, raw, raw_words);
    hipLaunchKernelGGL(jpeg_ffcount_kernel, dim3(pieces, p.n), dim3(256), 0, st, raw, raw_words, totalbits, ffcnt, pieces);
    hipLaunchKernelGGL(jpeg_head_kernel, dim3(p.n), dim3(64), 0, st, totalbits, raw_words, tabs, g, q, ffcnt, pieces, hdrlen, p.out, p.out_stride, p.sizes);
    hipLaunchKernelGGL(jpeg_stuff_kernel, dim3(pieces, p.n), dim3(256), 0, st, raw, raw_words, totalbits, ffcnt, pieces, hdrlen, p.out, p.out_stride);
    return hipGetLastError();
}
