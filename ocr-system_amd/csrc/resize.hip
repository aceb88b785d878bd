// Reference pre-processing on the GPU, byte-exact with PIL (the reference's CPU path):
//   resize_if_needed  /root/reference/backend/utils/image_preprocessing.py:81-110  (LANCZOS, 8-bit, 22-bit fixed point,
//                     horizontal pass then vertical pass — Pillow's ImagingResample)
//   enhance_contrast  :132-144 (blend with the mean-gray image, factor 1.2)
//   enhance_sharpness :146-158 (blend with the 3x3 SMOOTH-filtered image, factor 1.1)
// Coefficient tables are computed on the host in double precision with libm sin(), exactly like Pillow's
// precompute_coeffs / normalize_coeffs_8bpc, and cached on the device per (in, out) size.
// HBM-bound: 3 B/px read + 3 B/px written per pass.
#include "resize.h"
#include "lds_rows.h"

#include <cmath>

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

// Horizontal pass: one workgroup per input row.  The row is staged in LDS with aligned 4-byte loads, every thread then
// produces output bytes e = xo*C + c (coalesced byte stores); coefficient rows come from L1/L2 (shared by all rows).
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* in, uint8_t* out, const int* bounds, const int* kk, int ksize, int H, int W,
                                                         int C, int Wo, size_t total_bytes) {
    extern __shared__ uint32_t row_lds[];
    const int row = blockIdx.x;  // n*H + y
    const size_t start = (size_t)row * W * C;
    const int off = (int)(start & 3);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(in + (start - off));
    const int nwords = (off + W * C + 3) >> 2;
    const size_t wbase = start - off;
    for (int i = threadIdx.x; i < nwords; i += 256) {
        uint32_t w;
        if (wbase + 4 * (size_t)(i + 1) <= total_bytes) w = src[i];
        else {  // last word of the buffer: never read past the allocation
            w = 0;
            for (int b = 0; b < 4; ++b)
                if (wbase + 4 * (size_t)i + b < total_bytes) w |= (uint32_t)in[wbase + 4 * (size_t)i + b] << (8 * b);
        }
        row_lds[i] = w;
    }
    __syncthreads();
    const uint8_t* rb = reinterpret_cast<const uint8_t*>(row_lds) + off;
    uint8_t* orow = out + (size_t)row * Wo * C;
    for (int xo = threadIdx.x; xo < Wo; xo += 256) {   // one output pixel (all channels) per thread: bounds / coefficients read once
        const int lo = bounds[2 * xo], cnt = bounds[2 * xo + 1];
        const int* k = kk + (size_t)xo * ksize;
        const uint8_t* p = rb + lo * C;
        if (C == 3) {
            int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
            for (int j = 0; j < cnt; ++j) {
                const int kj = k[j];
                s0 += (int)p[3 * j] * kj; s1 += (int)p[3 * j + 1] * kj; s2 += (int)p[3 * j + 2] * kj;
            }
            s0 >>= PRECISION_BITS; s1 >>= PRECISION_BITS; s2 >>= PRECISION_BITS;
            orow[3 * xo] = (uint8_t)(s0 < 0 ? 0 : (s0 > 255 ? 255 : s0));
            orow[3 * xo + 1] = (uint8_t)(s1 < 0 ? 0 : (s1 > 255 ? 255 : s1));
            orow[3 * xo + 2] = (uint8_t)(s2 < 0 ? 0 : (s2 > 255 ? 255 : s2));
        } else {
            for (int c = 0; c < C; ++c) {
                int ss = 1 << (PRECISION_BITS - 1);
                for (int j = 0; j < cnt; ++j) ss += (int)p[j * C + c] * k[j];
                const int v = ss >> PRECISION_BITS;
                orow[xo * C + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    }
}

// Vertical pass: one workgroup per output row; threads sweep the row VEC bytes at a time (coefficients are row-uniform).
template <int VEC>
__global__ __launch_bounds__(256) void resample_v_kernel(const uint8_t* in, uint8_t* out, const int* bounds, const int* kk, int ksize, int H,
                                                         int rowbytes, int Ho) {
    const int n = blockIdx.x / Ho, yo = blockIdx.x - n * Ho;
    const int lo = bounds[2 * yo], cnt = bounds[2 * yo + 1];
    const int* k = kk + (size_t)yo * ksize;
    const uint8_t* base = in + ((size_t)n * H + lo) * rowbytes;
    uint8_t* orow = out + ((size_t)n * Ho + yo) * rowbytes;
    for (int xb = threadIdx.x * VEC; xb < rowbytes; xb += 256 * VEC) {
        int ss[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) ss[v] = 1 << (PRECISION_BITS - 1);
        for (int j = 0; j < cnt; ++j) {
            const uint8_t* p = base + (size_t)j * rowbytes + xb;
            const int kj = k[j];
            if (VEC == 4) {
                const uint32_t w = *reinterpret_cast<const uint32_t*>(p);
                ss[0] += (int)(w & 0xff) * kj; ss[1 % VEC] += (int)((w >> 8) & 0xff) * kj;
                ss[2 % VEC] += (int)((w >> 16) & 0xff) * kj; ss[3 % VEC] += (int)(w >> 24) * kj;
            } else if (VEC == 2) {
                const uint32_t w = *reinterpret_cast<const uint16_t*>(p);
                ss[0] += (int)(w & 0xff) * kj; ss[1 % VEC] += (int)(w >> 8) * kj;
            } else {
                ss[0] += (int)p[0] * kj;
            }
        }
        uint32_t packed = 0;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            int q = ss[v] >> PRECISION_BITS;
            q = q < 0 ? 0 : (q > 255 ? 255 : q);
            packed |= (uint32_t)q << (8 * v);
        }
        if (VEC == 4) *reinterpret_cast<uint32_t*>(orow + xb) = packed;
        else if (VEC == 2) *reinterpret_cast<uint16_t*>(orow + xb) = (uint16_t)packed;
        else orow[xb] = (uint8_t)packed;
    }
}

// Horizontal pass, RGB fast path (<= HK <= HKMAX taps, segment <= HPXMAX input pixels): one workgroup = HR rows x HX output pixels, one
// output pixel (of all HR rows) per thread.  The thread's tap coefficients are fetched up front into registers, the HR input
// row segments go to LDS by fill_rows(), results are staged in LDS and leave as aligned 4-byte stores.
constexpr int HR = 8, HX = 256, HKMAX = 12, HPXMAX = 400;
constexpr int HP_IN = (HPXMAX * 3 + 3) / 4 + 2 + HKMAX;   // LDS pitch (words) of an input row segment (+ slack for zero taps)
constexpr int HP_OUT = (HX * 3 + 3) / 4 + 2;           // LDS pitch (words) of an output row segment
template <int HK>
__global__ __launch_bounds__(256) void resample_h_rgb_kernel(const uint8_t* in, uint8_t* out, const int* bounds, const int* kk, int ksize, int rows,
                                                             int W, int Wo, long long total_in) {
    __shared__ uint32_t lin[HR * HP_IN];
    __shared__ uint32_t lout[HR * HP_OUT];
    const int tid = threadIdx.x;
    const int xo0 = blockIdx.x * HX, row0 = blockIdx.y * HR;
    const int nx = min(HX, Wo - xo0), nrows = min(HR, rows - row0);
    const int xo = xo0 + min(tid, nx - 1);
    const int lo = bounds[2 * xo];
    int kreg[HK];
#pragma unroll
    for (int j = 0; j < HK; ++j) kreg[j] = j < ksize ? kk[(size_t)xo * ksize + j] : 0;  // entries >= cnt are zero in the table
    const int in_lo = bounds[2 * xo0];
    const int in_hi = bounds[2 * (xo0 + nx - 1)] + bounds[2 * (xo0 + nx - 1) + 1];
    const int rowb = W * 3;
    const long long g0 = (long long)row0 * rowb + (long long)in_lo * 3;
    const int nw = ((in_hi - in_lo) * 3 + 3 + 3) / 4;
    fill_rows(lin, HP_IN, nw, in, g0, rowb, nrows, 0, nrows, total_in, WordIdentity());
    __syncthreads();
    const uintptr_t ibase = reinterpret_cast<uintptr_t>(in), obase = reinterpret_cast<uintptr_t>(out);
    const uint8_t* lb = reinterpret_cast<const uint8_t*>(lin);
    uint8_t* ob = reinterpret_cast<uint8_t*>(lout);
    if (tid < nx) {
#pragma unroll
        for (int r = 0; r < HR; ++r) {
            if (r >= nrows) break;
            const int m = (int)((ibase + (unsigned long long)(g0 + (long long)r * rowb)) & 3);
            const uint8_t* p = lb + r * HP_IN * 4 + m + (lo - in_lo) * 3;
            int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
#pragma unroll
            for (int j = 0; j < HK; ++j) {  // straight-line: taps beyond the pixel's count have coefficient 0 (reads stay inside the LDS slack)
                s0 += __mul24((int)p[3 * j], kreg[j]); s1 += __mul24((int)p[3 * j + 1], kreg[j]); s2 += __mul24((int)p[3 * j + 2], kreg[j]);   // |coefficient| < 2^23: v_mad_i32_i24 (v_mul_lo_u32 is quarter rate)
            }
            s0 >>= PRECISION_BITS; s1 >>= PRECISION_BITS; s2 >>= PRECISION_BITS;
            const long long go = ((long long)(row0 + r) * Wo + xo0) * 3;
            const int mo = (int)((obase + (unsigned long long)go) & 3);
            uint8_t* o = ob + r * HP_OUT * 4 + mo + tid * 3;
            o[0] = (uint8_t)(s0 < 0 ? 0 : (s0 > 255 ? 255 : s0));
            o[1] = (uint8_t)(s1 < 0 ? 0 : (s1 > 255 ? 255 : s1));
            o[2] = (uint8_t)(s2 < 0 ? 0 : (s2 > 255 ? 255 : s2));
        }
    }
    __syncthreads();
    const int segb = nx * 3;
    for (int i = tid; i < nrows * HP_OUT; i += 256) {
        const int r = i / HP_OUT, k = i - r * HP_OUT;
        const long long go = ((long long)(row0 + r) * Wo + xo0) * 3;
        const int mo = (int)((obase + (unsigned long long)go) & 3);
        const int b0 = 4 * k - mo;  // segment byte index of this word's byte 0
        if (b0 >= segb || b0 + 3 < 0) continue;
        const uint32_t w = lout[r * HP_OUT + k];
        uint8_t* dst = out + go + b0;
        if (b0 >= 0 && b0 + 4 <= segb) *reinterpret_cast<uint32_t*>(dst) = w;
        else
            for (int b = 0; b < 4; ++b)
                if (b0 + b >= 0 && b0 + b < segb) dst[b] = (uint8_t)(w >> (8 * b));
    }
}

// Vertical pass, tiled (<= VK <= VKMAX taps): one workgroup = VT_R output rows x VT_XB row bytes.  The input rows those outputs
// touch are staged in LDS by fill_rows() (lds_rows rows are allocated: the largest span of any row group + VK of slack, so the
// straight-line tap loop may read rows past the span with zero coefficients), the VT_R coefficient rows are staged too; every
// thread owns one ALIGNED 4-byte word of each output row and, per tap, reads the 4 input bytes above it as two LDS words +
// v_alignbyte.
constexpr int VT_XB = 1024, VT_R = 8, VKMAX = 16, VT_PAD = 4;
constexpr int VT_NW = (VT_XB + 2 * VT_PAD) / 4 + 1, VT_PITCH = VT_NW + 2;
template <int VK>
__global__ __launch_bounds__(256) void resample_v_tile_kernel(const uint8_t* in, uint8_t* out, const int* bounds, const int* kk, int ksize, int H,
                                                              int rowbytes, int Ho, long long total_in) {
    extern __shared__ uint32_t lds[];   // [lds_rows][VT_PITCH]
    __shared__ int kl[VT_R * VK];
    const int tid = threadIdx.x;
    const int n = blockIdx.z, yo0 = blockIdx.y * VT_R, x0 = blockIdx.x * VT_XB;
    const int nyo = min(VT_R, Ho - yo0);
    for (int i = tid; i < VT_R * VK; i += 256) {
        const int rr = i / VK, j = i - rr * VK;
        kl[i] = (rr < nyo && j < ksize) ? kk[(size_t)(yo0 + rr) * ksize + j] : 0;
    }
    const int in_lo = bounds[2 * yo0];
    const int in_hi = bounds[2 * (yo0 + nyo - 1)] + bounds[2 * (yo0 + nyo - 1) + 1];  // exclusive
    const uintptr_t ibase = reinterpret_cast<uintptr_t>(in), obase = reinterpret_cast<uintptr_t>(out);
    const long long g0 = ((long long)n * H + in_lo) * rowbytes + x0 - VT_PAD;
    fill_rows(lds, VT_PITCH, VT_NW, in, g0, rowbytes, in_hi - in_lo, 0, in_hi - in_lo, total_in, WordIdentity());
    __syncthreads();
    const int m0 = (int)((ibase + (unsigned long long)g0) & 3), dm = rowbytes & 3;
    for (int rr = 0; rr < nyo; ++rr) {
        const int yo = yo0 + rr;
        const int lo = bounds[2 * yo] - in_lo;
        const long long go = ((long long)n * Ho + yo) * rowbytes + x0;
        const int mo = (int)((obase + (unsigned long long)go) & 3);
        const int ew = x0 - mo + 4 * tid;
        if (ew >= rowbytes || ew + 3 < 0) continue;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0, s3 = s0;
#pragma unroll
        for (int j = 0; j < VK; ++j) {  // straight-line: taps beyond the row's count have coefficient 0
            const int r = lo + j;
            const int off = ((m0 + r * dm) & 3) - mo + 4 * tid + VT_PAD;
            const uint32_t* lr = lds + r * VT_PITCH + (off >> 2);
            const uint32_t w = __builtin_amdgcn_alignbyte(lr[1], lr[0], (uint32_t)(off & 3));
            const int kj = kl[rr * VK + j];
            s0 += __mul24((int)(w & 0xff), kj); s1 += __mul24((int)((w >> 8) & 0xff), kj); s2 += __mul24((int)((w >> 16) & 0xff), kj); s3 += __mul24((int)(w >> 24), kj);
        }
        s0 >>= PRECISION_BITS; s1 >>= PRECISION_BITS; s2 >>= PRECISION_BITS; s3 >>= PRECISION_BITS;
        const uint32_t res = (uint32_t)(s0 < 0 ? 0 : (s0 > 255 ? 255 : s0)) | ((uint32_t)(s1 < 0 ? 0 : (s1 > 255 ? 255 : s1)) << 8) |
                             ((uint32_t)(s2 < 0 ? 0 : (s2 > 255 ? 255 : s2)) << 16) | ((uint32_t)(s3 < 0 ? 0 : (s3 > 255 ? 255 : s3)) << 24);
        uint8_t* dst = out + go - mo + 4 * tid;
        if (ew >= 0 && ew + 4 <= rowbytes) *reinterpret_cast<uint32_t*>(dst) = res;
        else
            for (int j = 0; j < 4; ++j)
                if (ew + j >= 0 && ew + j < rowbytes) dst[j] = (uint8_t)(res >> (8 * j));
    }
}

__global__ __launch_bounds__(256) void gray_sum_kernel(const uint8_t* img, unsigned long long* sums, int HW) {
    __shared__ unsigned long long part[4];
    const int n = blockIdx.y;
    const uint8_t* p = img + (size_t)n * HW * 3;
    unsigned long long s = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        const unsigned r = p[(size_t)i * 3], g = p[(size_t)i * 3 + 1], b = p[(size_t)i * 3 + 2];
        s += (r * 19595u + g * 38470u + b * 7471u + 0x8000u) >> 16;
    }
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)(s & 0xffffffffull), m), hi = __shfl_xor((unsigned)(s >> 32), m);
        s += ((unsigned long long)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sums[n], part[0] + part[1] + part[2] + part[3]);  // one atomic per workgroup
}

__device__ __forceinline__ uint8_t blend_u8(int deg, int v, float alpha) {
    const float t = __fadd_rn((float)deg, __fmul_rn(alpha, (float)(v - deg)));
    // PIL clips only when alpha is outside [0, 1]; inside, t already lies between the two byte values, so one unconditional
    // clamp (v_med3_f32) is the same function without a per-byte branch
    return (uint8_t)fminf(fmaxf(t, 0.f), 255.f);
}

// Fused contrast + sharpness.  One workgroup = EN_R rows x EN_XB row bytes.  The contrast-blended bytes of the EN_R + 2 rows
// (8 bytes of apron left and right) are built ONCE in LDS from aligned 4-byte global loads — every LDS row keeps its global
// word alignment, so the fill is load -> 4 blends -> ds_write_b32 — and the 3x3 SMOOTH blend then reads 12-byte windows at
// arbitrary byte offsets (4 aligned LDS words + v_alignbyte).  Each thread owns one ALIGNED 4-byte word of the output row
// (rows of W*3 bytes are not word aligned in general: the strip of a row is shifted left by the row's misalignment), so
// the result is stored with one dword store; the first / last word of a row is stored bytewise.
// Arithmetic order is Pillow's (ImageFilter.SMOOTH: row y+1, y, y-1, left to right, + 0.5 offset, float32), see
// /root/reference/backend/utils/image_preprocessing.py:132-158.
constexpr int EN_XB = 1024, EN_R = 8, EN_PAD = 8;
constexpr int EN_NW = (EN_XB + 2 * EN_PAD) / 4 + 1;  // words per LDS row (misalignment <= 3 bytes)
constexpr int EN_PITCH = EN_NW + 3;

__device__ __forceinline__ uint32_t blend_u8x4(int deg, uint32_t w, float alpha) {
    return (uint32_t)blend_u8(deg, (int)(w & 0xff), alpha) | ((uint32_t)blend_u8(deg, (int)((w >> 8) & 0xff), alpha) << 8) |
           ((uint32_t)blend_u8(deg, (int)((w >> 16) & 0xff), alpha) << 16) | ((uint32_t)blend_u8(deg, (int)(w >> 24), alpha) << 24);
}

struct ContrastWord {
    int mean; float alpha;
    __device__ __forceinline__ uint32_t operator()(uint32_t w) const { return blend_u8x4(mean, w, alpha); }
};

__global__ __launch_bounds__(256) void enhance_kernel(const uint8_t* img, uint8_t* out, const unsigned long long* sums, int H, int W,
                                                      float alpha_c, float alpha_s, long long total_bytes) {
    __shared__ uint32_t lds[(EN_R + 2) * EN_PITCH];
    const int tid = threadIdx.x;
    const int n = blockIdx.z, yb = blockIdx.y * EN_R, e0 = blockIdx.x * EN_XB;
    const int rowb = W * 3;
    const int mean = (int)((double)sums[n] / (double)((long long)H * W) + 0.5);
    const uintptr_t ibase = reinterpret_cast<uintptr_t>(img), obase = reinterpret_cast<uintptr_t>(out);
    const long long g0 = ((long long)n * H + yb - 1) * rowb + e0 - EN_PAD;  // flat byte index of LDS row 0's byte 0 (image row yb - 1)
    const int rlo = yb == 0 ? 1 : 0, rhi = min(EN_R + 2, H - yb + 1);       // LDS rows whose image row exists
    fill_rows(lds, EN_PITCH, EN_NW, img, g0, rowb, EN_R + 2, rlo, rhi, total_bytes, ContrastWord{mean, alpha_c});
    __syncthreads();

    for (int rr = 0; rr < EN_R; ++rr) {
        const int y = yb + rr;
        if (y >= H) break;
        const long long go = ((long long)n * H + y) * rowb + e0;
        const int mo = (int)((obase + (unsigned long long)go) & 3);
        const int ew = e0 - mo + 4 * tid;  // row-relative byte index of this thread's aligned output word
        if (ew >= rowb || ew + 3 < 0) continue;
        uint32_t win[3][3];  // [row y-1, y, y+1][12 bytes starting at row byte ew - 3]
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int yy = y - 1 + d;
            const long long g = ((long long)n * H + yy) * rowb + e0 - EN_PAD;
            const int m = (int)((ibase + (unsigned long long)g) & 3);
            const int off = m - mo + 4 * tid + (EN_PAD - 3);
            const int q = off >> 2;
            const uint32_t sh = (uint32_t)(off & 3);
            const uint32_t* lr = lds + (rr + d) * EN_PITCH + q;
            const uint32_t w0 = lr[0], w1 = lr[1], w2 = lr[2], w3 = lr[3];
            win[d][0] = __builtin_amdgcn_alignbyte(w1, w0, sh);
            win[d][1] = __builtin_amdgcn_alignbyte(w2, w1, sh);
            win[d][2] = __builtin_amdgcn_alignbyte(w3, w2, sh);
        }
        const bool yin = y > 0 && y < H - 1;
        uint32_t res = 0;
#define WB(d_, p_) ((float)((win[d_][(p_) >> 2] >> (8 * ((p_) & 3))) & 0xffu))
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = ew + j;
            const int v = (int)((win[1][(j + 3) >> 2] >> (8 * ((j + 3) & 3))) & 0xffu);
            // straight-line: the smoothed value is computed for every byte and selected for interior pixels only
            const float k1 = 1.0f / 13.0f, k5 = 5.0f / 13.0f;
            float ss = 0.5f;
            float a = __fmul_rn(WB(2, j), k1);
            a = __fadd_rn(a, __fmul_rn(WB(2, j + 3), k1));
            a = __fadd_rn(a, __fmul_rn(WB(2, j + 6), k1));
            ss = __fadd_rn(ss, a);
            a = __fmul_rn(WB(1, j), k1);
            a = __fadd_rn(a, __fmul_rn(WB(1, j + 3), k5));
            a = __fadd_rn(a, __fmul_rn(WB(1, j + 6), k1));
            ss = __fadd_rn(ss, a);
            a = __fmul_rn(WB(0, j), k1);
            a = __fadd_rn(a, __fmul_rn(WB(0, j + 3), k1));
            a = __fadd_rn(a, __fmul_rn(WB(0, j + 6), k1));
            ss = __fadd_rn(ss, a);
            const int sm = (int)fminf(fmaxf(ss, 0.f), 255.f);
            const int deg = (yin && e >= 3 && e < rowb - 3) ? sm : v;
            res |= (uint32_t)blend_u8(deg, v, alpha_s) << (8 * j);
        }
#undef WB
        uint8_t* dst = out + go - mo + 4 * tid;
        if (ew >= 0 && ew + 4 <= rowb) *reinterpret_cast<uint32_t*>(dst) = res;
        else
            for (int j = 0; j < 4; ++j)
                if (ew + j >= 0 && ew + j < rowb) dst[j] = (uint8_t)(res >> (8 * j));
    }
}

inline int grid_for(size_t total) {
    size_t g = (total + 255) / 256;
    return (int)(g > 256 * 32 ? 256 * 32 : (g ? g : 1));
}

double sinc_filter(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return sin(x) / x;
}
double lanczos_filter(double x) {
    if (-3.0 <= x && x < 3.0) return sinc_filter(x) * sinc_filter(x / 3);
    return 0.0;
}

}  // namespace

void lanczos_coeffs(int in_size, int out_size, int* ksize_out, std::vector<int>* bounds, std::vector<int>* kk) {
    const float in0 = 0.0f, in1 = (float)in_size;
    double filterscale, scale;
    filterscale = scale = (double)(in1 - in0) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 3.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    bounds->assign((size_t)out_size * 2, 0);
    kk->assign((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = in0 + (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = lanczos_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) k[x] /= ww;
            (*kk)[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << PRECISION_BITS)) : (int)(0.5 + k[x] * (1 << PRECISION_BITS));
        }
        (*bounds)[2 * xx] = xmin;
        (*bounds)[2 * xx + 1] = xmax;
    }
    *ksize_out = ksize;
}

hipError_t resample_launch(const uint8_t* in, uint8_t* out, const int* bounds_dev, const int* kk_dev, int ksize, int N, int H, int W, int C,
                           int out_len, int axis, const int* bounds_host, hipStream_t st) {
    { hipError_t e = locr_dyn_lds(reinterpret_cast<const void*>(resample_h_kernel), 150 * 1024); if (e != hipSuccess) return e; }
    if (axis == 0) {
        const size_t lds = ((size_t)W * C + 8 + 3) / 4 * 4;
        if (lds > 150 * 1024) return hipErrorInvalidValue;
        bool fast = C == 3 && ksize <= HKMAX && bounds_host != nullptr;
        for (int x0 = 0; fast && x0 < out_len; x0 += HX) {
            const int x1 = (x0 + HX < out_len ? x0 + HX : out_len) - 1;
            if (bounds_host[2 * x1] + bounds_host[2 * x1 + 1] - bounds_host[2 * x0] > HPXMAX) fast = false;
        }
        if (fast) {
            const int rows = N * H;
            const dim3 grid((out_len + HX - 1) / HX, (rows + HR - 1) / HR);
            if (ksize <= 9) hipLaunchKernelGGL(resample_h_rgb_kernel<9>, grid, dim3(256), 0, st, in, out, bounds_dev, kk_dev, ksize, rows, W, out_len, (long long)N * H * W * C);
            else hipLaunchKernelGGL(resample_h_rgb_kernel<HKMAX>, grid, dim3(256), 0, st, in, out, bounds_dev, kk_dev, ksize, rows, W, out_len, (long long)N * H * W * C);
        } else {
            hipLaunchKernelGGL(resample_h_kernel, dim3(N * H), dim3(256), lds, st, in, out, bounds_dev, kk_dev, ksize, H, W, C, out_len, (size_t)N * H * W * C);
        }
    } else {
        const int rowbytes = W * C;
        // tiled kernel when the taps fit its unrolled loop and the largest input-row span of a group of VT_R output rows fits LDS
        int span = 0;
        for (int y0 = 0; bounds_host != nullptr && y0 < out_len; y0 += VT_R) {
            const int y1 = (y0 + VT_R < out_len ? y0 + VT_R : out_len) - 1;
            const int sp = bounds_host[2 * y1] + bounds_host[2 * y1 + 1] - bounds_host[2 * y0];
            if (sp > span) span = sp;
        }
        const int vk = ksize <= 9 ? 9 : VKMAX;
        const size_t lds_bytes = (size_t)(span + vk) * VT_PITCH * 4;
        if (bounds_host != nullptr && ksize <= VKMAX && lds_bytes <= 60 * 1024) {
            const dim3 grid((rowbytes + 3 + VT_XB - 1) / VT_XB, (out_len + VT_R - 1) / VT_R, N);
            if (vk == 9) hipLaunchKernelGGL(resample_v_tile_kernel<9>, grid, dim3(256), lds_bytes, st, in, out, bounds_dev, kk_dev, ksize, H, rowbytes, out_len, (long long)N * H * rowbytes);
            else hipLaunchKernelGGL(resample_v_tile_kernel<VKMAX>, grid, dim3(256), lds_bytes, st, in, out, bounds_dev, kk_dev, ksize, H, rowbytes, out_len, (long long)N * H * rowbytes);
            return hipGetLastError();
        }
        const bool a4 = rowbytes % 4 == 0 && (reinterpret_cast<uintptr_t>(in) % 4 == 0) && (reinterpret_cast<uintptr_t>(out) % 4 == 0);
        const bool a2 = rowbytes % 2 == 0 && (reinterpret_cast<uintptr_t>(in) % 2 == 0) && (reinterpret_cast<uintptr_t>(out) % 2 == 0);
        if (a4) hipLaunchKernelGGL(resample_v_kernel<4>, dim3(N * out_len), dim3(256), 0, st, in, out, bounds_dev, kk_dev, ksize, H, rowbytes, out_len);
        else if (a2) hipLaunchKernelGGL(resample_v_kernel<2>, dim3(N * out_len), dim3(256), 0, st, in, out, bounds_dev, kk_dev, ksize, H, rowbytes, out_len);
        else hipLaunchKernelGGL(resample_v_kernel<1>, dim3(N * out_len), dim3(256), 0, st, in, out, bounds_dev, kk_dev, ksize, H, rowbytes, out_len);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- binarisation
// image_preprocessing.py:175-185 (`binarize`: L > 128, what the reference's adaptive_binarize degrades to without OpenCV, :473-475)
// and :462-494 (`adaptive_binarize`: cv2.adaptiveThreshold(gray, 255, GAUSSIAN_C, BINARY, 11, 2), restated in oracle/preprocess.py).
// L = PIL's convert('L'): (19595 R + 38470 G + 7471 B + 0x8000) >> 16.  Output: the 0 / 255 value on all three channels.
namespace {
constexpr int BZ_TH = 16, BZ_TW = 64, BZ_R = 5;
__constant__ float c_gauss11[11] = {0x1.20c256p-7f, 0x1.bcb86ap-6f, 0x1.0ab50ap-4f, 0x1.f2464cp-4f, 0x1.6a7e1ep-3f, 0x1.9ac20ap-3f,
                                    0x1.6a7e1ep-3f, 0x1.f2464cp-4f, 0x1.0ab50ap-4f, 0x1.bcb86ap-6f, 0x1.20c256p-7f};
__device__ __forceinline__ int lum_L(const uint8_t* p) { return (19595 * p[0] + 38470 * p[1] + 7471 * p[2] + 0x8000) >> 16; }

__global__ __launch_bounds__(256) void binarize_kernel(const uint8_t* img, uint8_t* out, int H, int W, int adaptive, int threshold) {
    __shared__ uint8_t sL[(BZ_TH + 2 * BZ_R) * (BZ_TW + 2 * BZ_R)];
    __shared__ float sRow[(BZ_TH + 2 * BZ_R) * BZ_TW];
    const int pg = blockIdx.z, y0 = blockIdx.y * BZ_TH, x0 = blockIdx.x * BZ_TW, tid = threadIdx.x;
    const uint8_t* src = img + (size_t)pg * H * W * 3;
    uint8_t* dst = out + (size_t)pg * H * W * 3;
    constexpr int LW = BZ_TW + 2 * BZ_R, LH = BZ_TH + 2 * BZ_R;
    if (!adaptive) {
        for (int i = tid; i < BZ_TH * BZ_TW; i += 256) {
            const int y = y0 + i / BZ_TW, x = x0 + i % BZ_TW;
            if (y >= H || x >= W) continue;
            const uint8_t v = lum_L(src + ((size_t)y * W + x) * 3) > threshold ? 255 : 0;
            uint8_t* o = dst + ((size_t)y * W + x) * 3;
            o[0] = v; o[1] = v; o[2] = v;
        }
        return;
    }
    for (int i = tid; i < LH * LW; i += 256) {   // L tile with replicated borders
        const int y = min(max(y0 - BZ_R + i / LW, 0), H - 1), x = min(max(x0 - BZ_R + i % LW, 0), W - 1);
        sL[i] = (uint8_t)lum_L(src + ((size_t)y * W + x) * 3);
    }
    __syncthreads();
    for (int i = tid; i < LH * BZ_TW; i += 256) {   // horizontal pass: taps in ascending order, one multiply and one add each
        const int r = i / BZ_TW, c = i % BZ_TW;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) acc = __fadd_rn(acc, __fmul_rn((float)sL[r * LW + c + k], c_gauss11[k]));
        sRow[i] = acc;
    }
    __syncthreads();
    for (int i = tid; i < BZ_TH * BZ_TW; i += 256) {
        const int r = i / BZ_TW, c = i % BZ_TW, y = y0 + r, x = x0 + c;
        if (y >= H || x >= W) continue;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) acc = __fadd_rn(acc, __fmul_rn(sRow[(r + k) * BZ_TW + c], c_gauss11[k]));
        const int mean = min(max((int)rintf(acc), 0), 255);
        const uint8_t v = ((int)sL[(r + BZ_R) * LW + c + BZ_R] - mean > -2) ? 255 : 0;
        uint8_t* o = dst + ((size_t)y * W + x) * 3;
        o[0] = v; o[1] = v; o[2] = v;
    }
}
}  // namespace

namespace {
// ImagePreprocessor.convert_to_grayscale (image_preprocessing.py:167-169) = PIL convert('L'): L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16,
// written to all three channels (the engine's page format; ImageEnhance on an L image and on its R = G = B replica give the same bytes)
__global__ __launch_bounds__(256) void grayscale_kernel(const uint8_t* img, uint8_t* out, long long npix) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const uint8_t* s = img + i * 3;
        const unsigned l = (19595u * s[0] + 38470u * s[1] + 7471u * s[2] + 0x8000u) >> 16;
        out[i * 3 + 0] = (uint8_t)l; out[i * 3 + 1] = (uint8_t)l; out[i * 3 + 2] = (uint8_t)l;
    }
}

// ImagePreprocessor.denoise (:160-165) = PIL ImageFilter.MedianFilter(3): per channel the 5th smallest of the 3x3 neighbourhood, the
// image extended by replicating its edge pixels (RankFilter expands the image by size // 2 before filtering).  One thread per pixel,
// all three channels; the nine values go through the 19-exchange median network (min / max only: branch-free).
__device__ __forceinline__ void mm(int& a, int& b) { const int lo = min(a, b), hi = max(a, b); a = lo; b = hi; }
__device__ __forceinline__ int median9(int v0, int v1, int v2, int v3, int v4, int v5, int v6, int v7, int v8) {
    mm(v1, v2); mm(v4, v5); mm(v7, v8); mm(v0, v1); mm(v3, v4); mm(v6, v7); mm(v1, v2); mm(v4, v5); mm(v7, v8);
    mm(v0, v3); mm(v5, v8); mm(v4, v7); mm(v3, v6); mm(v1, v4); mm(v2, v5); mm(v4, v7); mm(v4, v2); mm(v6, v4); mm(v4, v2);
    return v4;
}
__global__ __launch_bounds__(256) void median3_kernel(const uint8_t* img, uint8_t* out, int H, int W) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const uint8_t* base = img + (size_t)blockIdx.z * H * W * 3;
    int v[3][9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = min(max(y + dy - 1, 0), H - 1);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = min(max(x + dx - 1, 0), W - 1);
            const uint8_t* s = base + ((size_t)yy * W + xx) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c][dy * 3 + dx] = s[c];
        }
    }
    uint8_t* d = out + (((size_t)blockIdx.z * H + y) * W + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = (uint8_t)median9(v[c][0], v[c][1], v[c][2], v[c][3], v[c][4], v[c][5], v[c][6], v[c][7], v[c][8]);
}
}  // namespace

namespace {
// ImageOps.exif_transpose (auto_orient, image_preprocessing.py:171-173 / :213): EXIF orientation 2..8 as a pixel permutation.  out (y, x) <- in (sy, sx);
// orientations 5..8 swap width and height.  One thread per output pixel (3 bytes); the reads of the transposing modes stride through the input.
__global__ __launch_bounds__(256) void exif_transpose_kernel(const uint8_t* img, uint8_t* out, int H, int W, int orientation) {
    const int Ho = orientation >= 5 ? W : H, Wo = orientation >= 5 ? H : W;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= Wo || y >= Ho) return;
    int sy, sx;
    switch (orientation) {
        case 2: sy = y; sx = W - 1 - x; break;
        case 3: sy = H - 1 - y; sx = W - 1 - x; break;
        case 4: sy = H - 1 - y; sx = x; break;
        case 5: sy = x; sx = y; break;
        case 6: sy = H - 1 - x; sx = y; break;
        case 7: sy = H - 1 - x; sx = W - 1 - y; break;
        case 8: sy = x; sx = W - 1 - y; break;
        default: sy = y; sx = x; break;
    }
    const uint8_t* s = img + (((size_t)blockIdx.z * H + sy) * W + sx) * 3;
    uint8_t* d = out + (((size_t)blockIdx.z * Ho + y) * Wo + x) * 3;
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
}
}  // namespace

hipError_t exif_transpose_launch(const uint8_t* img, uint8_t* out, int N, int H, int W, int orientation, hipStream_t st) {
    const int Ho = orientation >= 5 ? W : H, Wo = orientation >= 5 ? H : W;
    hipLaunchKernelGGL(exif_transpose_kernel, dim3((Wo + 63) / 64, (Ho + 3) / 4, N), dim3(256), 0, st, img, out, H, W, orientation);
    return hipGetLastError();
}

hipError_t grayscale_launch(const uint8_t* img, uint8_t* out, int N, int H, int W, hipStream_t st) {
    const long long npix = (long long)N * H * W;
    long long g = (npix + 255) / 256;
    hipLaunchKernelGGL(grayscale_kernel, dim3((unsigned)(g > 65536 ? 65536 : g)), dim3(256), 0, st, img, out, npix);
    return hipGetLastError();
}

hipError_t median3_launch(const uint8_t* img, uint8_t* out, int N, int H, int W, hipStream_t st) {
    hipLaunchKernelGGL(median3_kernel, dim3((W + 63) / 64, (H + 3) / 4, N), dim3(256), 0, st, img, out, H, W);
    return hipGetLastError();
}

hipError_t binarize_launch(const uint8_t* img, uint8_t* out, int N, int H, int W, int adaptive, int threshold, hipStream_t st) {
    hipLaunchKernelGGL(binarize_kernel, dim3((W + BZ_TW - 1) / BZ_TW, (H + BZ_TH - 1) / BZ_TH, N), dim3(256), 0, st, img, out, H, W, adaptive, threshold);
    return hipGetLastError();
}

hipError_t enhance_launch(const uint8_t* img, uint8_t* tmp, uint8_t* out, unsigned long long* sums_dev, int N, int H, int W, float contrast,
                          float sharpness, hipStream_t st) {
    hipError_t e = hipMemsetAsync(sums_dev, 0, sizeof(unsigned long long) * N, st);
    if (e != hipSuccess) return e;
    const int HW = H * W;
    int gx = (HW + 255) / 256;
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(gray_sum_kernel, dim3(gx > 64 ? 64 : gx, N), dim3(256), 0, st, img, sums_dev, HW);
    (void)tmp;
    const int rowb = W * 3;
    hipLaunchKernelGGL(enhance_kernel, dim3((rowb + 3 + EN_XB - 1) / EN_XB, (H + EN_R - 1) / EN_R, N), dim3(256), 0, st, img, out, sums_dev, H, W,
                       contrast, sharpness, (long long)N * H * rowb);
    return hipGetLastError();
}
