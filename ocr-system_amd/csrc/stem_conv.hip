// Stem convolution fused with input normalisation: u8 HWC page/crop -> 3x3 / stride 2 conv (Cin = 3)
// -> bias + activation -> bf16 NHWC.  Also a standalone normalise kernel (u8 HWC -> bf16 NHWC/NCHW).
//
// The reference has no normalise-to-tensor step (its pre-processing stops at PIL images:
// /root/reference/backend/utils/image_preprocessing.py:191-242); the definition
//   xn = bf16( float(u8) * scale_c + shift_c ),  0 outside the valid page
// is the build's (oracle/nets.py det_normalize / rec_normalize).
//
// One workgroup = 8 x 32 output pixels.  The (17 x 65) x 3 input halo is normalised once into
// LDS as bf16; K = 27 (padded to 32) is two 32x32x16 MFMA steps whose B fragments are gathered
// from LDS (for a fixed kh the 9 (kw, c) values of a pixel are contiguous), weights (A operand,
// 2 KB, pre-packed [kstep][half][cout][8]) come straight from L2.
#include "stem_conv.h"

namespace {

constexpr int TH = 8, TW = 32;
constexpr int HH = 2 * (TH - 1) + 3, HW = 2 * (TW - 1) + 3;  // 17 x 65
constexpr int ROW = HW * 3 + 1;                              // bf16 elements per halo row (odd -> fewer conflicts)
constexpr int STAGE_PITCH = 32 * 2 + 16;

// Normalised bf16 halo of one workgroup: rows iy0 .. iy0 + NHH - 1, pixels ix0 .. ix0 + NHW - 1 of image n_img, 3 channels
// interleaved, NROW = NHW * 3 + 1 elements per LDS row (element NHW * 3 of every row is a dump slot nobody reads).
// Aligned 4-byte words of every row segment, all requested before the first is used; addresses are 32-bit offsets from a
// 4-byte-aligned, workgroup-uniform base inside this image; e0 = position of a word's first byte in its row segment.  The
// arithmetic per byte is straight-line (masks and selected indices, no branches: a conditional on the value makes hipcc
// branch around the conversion, ~35 instructions per byte with the old 64-bit offsets against ~9 now).
template <int NHH, int NHW, int NROW>
__device__ __forceinline__ void fill_halo_u8(bf16_t* halo, const StemParams& p, int n_img, int iy0, int ix0, int vh, int vw, int tid) {
    constexpr int SEG = NHW * 3, WPR = (SEG + 3) / 4 + 1, NIT = (NHH * WPR + 255) / 256;
    static_assert(NROW > SEG, "one dump element per row");
    const long long img_off = (long long)n_img * p.H * p.W * 3, all_bytes = (long long)p.N * p.H * p.W * 3;
    const unsigned char* xal = p.x + (img_off & ~3ll);       // p.x is an allocation base (>= 4-byte aligned)
    const int mis = (int)(img_off & 3);
    const long long lo64 = -(img_off & ~3ll), hi64 = all_bytes - (img_off & ~3ll);      // valid byte range relative to xal ...
    const int lo_lim = lo64 < -0x7fffffffll ? -0x7fffffff : (int)lo64, hi_lim = hi64 > 0x7fffffffll ? 0x7fffffff : (int)hi64;   // ... clamped: offsets are 32-bit
    const float sc0 = p.scale[0], sc1 = p.scale[1], sc2 = p.scale[2], sh0 = p.shift[0], sh1 = p.shift[1], sh2 = p.shift[2];
    uint32_t word[NIT];
    int e0s[NIT];
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int i = tid + 256 * u;
        const int hy = i / WPR, wi = i - hy * WPR, iy = iy0 + hy;
        const int q = mis + (iy * p.W + ix0) * 3;            // segment start relative to xal (negative at x = -1, y = 0)
        const int w0 = (q & ~3) + 4 * wi;                    // this thread's aligned word
        e0s[u] = w0 - q;
        // words that start before the tensor hold only pixels left of the page (zeroed below); the one word that may
        // straddle the tensor's end is patched after the loop
        const bool ld = i < NHH * WPR && iy >= 0 && iy < vh && w0 >= lo_lim && w0 + 4 <= hi_lim;
        const uint32_t wv = *reinterpret_cast<const uint32_t*>(xal + (ld ? w0 : 0));   // xal itself is always readable: lo_lim <= 0 < hi_lim
        word[u] = ld ? wv : 0u;
    }
    if (hi_lim & 3) {                                        // tensor size not a multiple of 4: its last, partial word byte by byte
#pragma unroll 1
        for (int u = 0; u < NIT; ++u) {
            const int i = tid + 256 * u;
            const int hy = i / WPR, wi = i - hy * WPR, iy = iy0 + hy;
            const int q = mis + (iy * p.W + ix0) * 3, w0 = (q & ~3) + 4 * wi;
            if (i < NHH * WPR && iy >= 0 && iy < vh && w0 >= lo_lim && w0 < hi_lim && w0 + 4 > hi_lim) {
                uint32_t wv = 0;
                for (int b = 0; b < 4; ++b) if (w0 + b < hi_lim) wv |= (uint32_t)xal[w0 + b] << (8 * b);
#pragma unroll
                for (int uu = 0; uu < NIT; ++uu) if (uu == u) word[uu] = wv;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int i = tid + 256 * u;
        const int hy = min(i / WPR, NHH - 1), iy = iy0 + hy;
        const bool rowin = iy >= 0 && iy < vh, item = i < NHH * WPR;
        const int e0 = e0s[u];                                // -3 .. SEG + 3
        const int hx0 = (e0 + 3) / 3 - 1, c0 = e0 - hx0 * 3;  // pixel and channel of the word's first byte (e0 >= -3)
        bf16_t* hrow = halo + hy * NROW;
        // the channels of the four bytes are c0, c0+1, c0+2, c0 (mod 3): rotate the constants once per word
        float S[4], T[4];
        S[0] = c0 == 0 ? sc0 : (c0 == 1 ? sc1 : sc2); S[1] = c0 == 0 ? sc1 : (c0 == 1 ? sc2 : sc0); S[2] = c0 == 0 ? sc2 : (c0 == 1 ? sc0 : sc1); S[3] = S[0];
        T[0] = c0 == 0 ? sh0 : (c0 == 1 ? sh1 : sh2); T[1] = c0 == 0 ? sh1 : (c0 == 1 ? sh2 : sh0); T[2] = c0 == 0 ? sh2 : (c0 == 1 ? sh0 : sh1); T[3] = T[0];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int e = e0 + b;
            const int ix = ix0 + hx0 + ((c0 + b >= 3) ? 1 : 0);
            float v = (float)((word[u] >> (8 * b)) & 0xffu) * S[b];
            v = v + T[b];
            const uint32_t keep = (rowin && (unsigned)ix < (unsigned)vw) ? 0xffffu : 0u;   // 0 outside the page
            const int idx = (item && (unsigned)e < (unsigned)SEG) ? e : SEG;
            hrow[idx] = (bf16_t)(f32_to_bf16(v) & keep);
        }
    }
}

// halo offsets of a lane's 16 K-indices (k = ks * 16 + h * 8 + j -> row kh = k / 9, element kr = k % 9), the same for every
// pixel; the five padding indices (k >= 27) re-read element 0: their weights are zero and the data is finite
template <int NROW>
__device__ __forceinline__ void stem_koff(int h, int (&koff)[2][8]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ks * 16 + h * 8 + j, kh = k / 9;
            koff[ks][j] = k < 27 ? kh * NROW + (k - kh * 9) : 0;
        }
}

__global__ __launch_bounds__(256) void stem_conv_kernel(const StemParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[256 * STAGE_PITCH > HH * ROW * 2 ? 256 * STAGE_PITCH : HH * ROW * 2];
    bf16_t* halo = reinterpret_cast<bf16_t*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int bid = blockIdx.x;
    const int n_img = bid / (tiles_x * tiles_y);
    const int trem = bid - n_img * tiles_x * tiles_y;
    const int tile_y = trem / tiles_x, tile_x = trem - tile_y * tiles_x;
    const int vw = p.valid_w_per_img ? p.valid_w_per_img[n_img] : p.valid_w;
    const int vh = p.valid_h;

    // ---- stage + normalise the halo ----
    const int iy0 = tile_y * TH * 2 - 1, ix0 = tile_x * TW * 2 - 1;
    fill_halo_u8<HH, HW, ROW>(halo, p, n_img, iy0, ix0, vh, vw, tid);
    __syncthreads();

    // ---- MFMA: D[cout][pixel] ----
    f32x16_t acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[mt][j] = 0.f;
    int koff[2][8];
    stem_koff<ROW>(h, koff);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const bf16x8_t afr = *reinterpret_cast<const bf16x8_t*>(p.wpk + ((ks * 2 + h) * 32 + r) * 8);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const bf16_t* hb = halo + (2 * (wave * 2 + mt)) * ROW + 6 * r;
            union { bf16x8_t v; bf16_t s[8]; } b;
#pragma unroll
            for (int j = 0; j < 8; ++j) b.s[j] = hb[koff[ks][j]];
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, b.v, acc[mt], 0, 0, 0);
        }
    }
    __syncthreads();

    // ---- epilogue ----
    unsigned char* stage = smem;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b4 = *reinterpret_cast<const float4*>(p.bias + 8 * g + 4 * h);
            acc[mt][4 * g + 0] += b4.x; acc[mt][4 * g + 1] += b4.y; acc[mt][4 * g + 2] += b4.z; acc[mt][4 * g + 3] += b4.w;
        }
    // the activation switch once per kernel, not once per element
#define STEM_ALL(expr) _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int j = 0; j < 16; ++j) { const float v = acc[mt][j]; acc[mt][j] = (expr); }
    if (p.act == ACT_RELU) { STEM_ALL(fmaxf(v, 0.f)) }
    else if (p.act == ACT_HSWISH) { STEM_ALL(v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f)) }
    else if (p.act != ACT_NONE) { STEM_ALL(apply_act(v, p.act)) }
#undef STEM_ALL
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int tp = (wave * 2 + mt) * TW + r;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 o;
            o.x = pack_bf16x2(acc[mt][4 * g + 0], acc[mt][4 * g + 1]);
            o.y = pack_bf16x2(acc[mt][4 * g + 2], acc[mt][4 * g + 3]);
            *reinterpret_cast<uint2*>(stage + tp * STAGE_PITCH + (8 * g + 4 * h) * 2) = o;
        }
    }
    __syncthreads();
    const int cpp = p.Cout_store / 8;  // 16-byte chunks stored per pixel (1..4)
    for (int i = tid; i < 256 * cpp; i += 256) {
        const int tp = i / cpp, ch = i - tp * cpp;
        const int ty = tp / TW, tx = tp - ty * TW;
        const int oy = tile_y * TH + ty, ox = tile_x * TW + tx;
        if (oy >= p.Ho || ox >= p.Wo) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(stage + tp * STAGE_PITCH + ch * 16);
        *reinterpret_cast<uint4*>(p.y + (((size_t)n_img * p.Ho + oy) * p.Wo + ox) * p.Cout_store + ch * 8) = v;
    }
}

// ---- stem.conv1 + stem.conv2 fused (DBNet): u8 page -> conv 3x3/s2 (3 -> 32) + ReLU -> conv 3x3/s1 (32 -> 32) + ReLU ----
// The 32-channel half-resolution tensor between the two (46 MB per A4 page, written and read back) only exists as an LDS tile.
// One workgroup = 8 x 32 output pixels of conv2:
//   A  the (21 x 69) x 3 u8 input region is normalised into LDS as bf16 (aligned word loads, requested in one batch);
//   B  conv1 on the matrix cores for the 10 x 34 pixels conv2 needs (11 column tiles of 32 over the four waves; B operand gathered
//      from the halo exactly as in stem_conv_kernel), + bias, ReLU, bf16 — pixels outside the half-resolution image become ZERO
//      (conv2 pads its input, not conv1's) — written to a [pixel][32 ch] LDS tile whose 16-byte slots are XOR-swizzled by
//      (pixel >> 2) & 3 so that conv2's stride-64-byte fragment reads are conflict free;
//   C  conv2: 2 chunks of 16 channels x 9 taps (the order of the stand-alone kernel: identical sums), A fragments pre-loaded from
//      the packed weights in L1/L2 into registers, B fragments from the LDS tile; bias, ReLU, staged 64-byte-per-pixel stores.
constexpr int F_T1H = TH + 2, F_T1W = TW + 2, F_T1PX = F_T1H * F_T1W;           // 10 x 34 = 340 conv1 pixels
constexpr int F_HH = 2 * (F_T1H - 1) + 3, F_HW = 2 * (F_T1W - 1) + 3;           // 21 x 69 input pixels
constexpr int F_ROW = F_HW * 3 + 1;                                             // bf16 elements per halo row
constexpr int F_HALO_BYTES = F_HH * F_ROW * 2, F_T1_BYTES = ((F_T1PX + 31) / 32) * 32 * 64;
__global__ __launch_bounds__(256) void stem12_kernel(const StemParams p, const bf16_t* w2pk, const float* bias2) {
    static_assert(256 * STAGE_PITCH <= F_T1_BYTES, "the output stage re-uses the t1 tile");
    __shared__ __attribute__((aligned(16))) unsigned char smem[F_HALO_BYTES + F_T1_BYTES];
    bf16_t* halo = reinterpret_cast<bf16_t*>(smem);
    unsigned char* t1 = smem + F_HALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int bid = blockIdx.x;
    const int n_img = bid / (tiles_x * tiles_y);
    const int trem = bid - n_img * tiles_x * tiles_y;
    const int tile_y = trem / tiles_x, tile_x = trem - tile_y * tiles_x;
    const int vw = p.valid_w, vh = p.valid_h;

    // ---- A: input region -> normalised bf16 halo ----
    const int y1_0 = tile_y * TH - 1, x1_0 = tile_x * TW - 1;   // conv1-output coordinates of the t1 tile's first pixel
    const int iy0 = 2 * y1_0 - 1, ix0 = 2 * x1_0 - 1;
    fill_halo_u8<F_HH, F_HW, F_ROW>(halo, p, n_img, iy0, ix0, vh, vw, tid);
    __syncthreads();

    // ---- B: conv1 (3 -> 32, stride 2) for the 340 pixels of the t1 tile ----
    {
        bf16x8_t a1[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a1[ks] = *reinterpret_cast<const bf16x8_t*>(p.wpk + ((ks * 2 + h) * 32 + r) * 8);
        float4 b1[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) b1[g] = *reinterpret_cast<const float4*>(p.bias + 8 * g + 4 * h);
        int koff[2][8];
        stem_koff<F_ROW>(h, koff);
        for (int tile = wave; tile * 32 < F_T1PX; tile += 4) {
            const int pidx = tile * 32 + r;
            const int pc = min(pidx, F_T1PX - 1);
            const int py = pc / F_T1W, px = pc - py * F_T1W;
            const bf16_t* hb = halo + (2 * py) * F_ROW + 6 * px;
            f32x16_t acc;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                union { bf16x8_t v; bf16_t s[8]; } b;
#pragma unroll
                for (int j = 0; j < 8; ++j) b.s[j] = hb[koff[ks][j]];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[ks], b.v, acc, 0, 0, 0);
            }
            const int y1 = y1_0 + py, x1 = x1_0 + px;
            const bool inimg = y1 >= 0 && y1 < p.Ho && x1 >= 0 && x1 < p.Wo;   // outside: conv2's zero padding
            if (pidx < F_T1PX) {
                const int swz = (pidx >> 2) & 3;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 o = make_uint2(0, 0);
                    if (inimg) {
                        o.x = pack_bf16x2(fmaxf(acc[4 * g + 0] + b1[g].x, 0.f), fmaxf(acc[4 * g + 1] + b1[g].y, 0.f));   // ReLU (checked at launch)
                        o.y = pack_bf16x2(fmaxf(acc[4 * g + 2] + b1[g].z, 0.f), fmaxf(acc[4 * g + 3] + b1[g].w, 0.f));
                    }
                    *reinterpret_cast<uint2*>(t1 + pidx * 64 + ((g ^ swz) * 16) + 8 * h) = o;
                }
            }
        }
    }
    // conv2 weight fragments (MFMA A operand) of one 16-channel chunk at a time, from L2 (18 KB shared by every work-group):
    // holding both chunks through phases A and B costs the registers of two more resident work-groups
    bf16x8_t w2[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) w2[tap] = *reinterpret_cast<const bf16x8_t*>(w2pk + ((size_t)h * (9 * 32) + tap * 32 + r) * 8);
    __syncthreads();

    // ---- C: conv2 (32 -> 32), chunk (16 channels) outer, tap inner: the stand-alone kernel's summation order ----
    f32x16_t acc2[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc2[mt][j] = 0.f;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int pp = (wave * 2 + mt + tap / 3) * F_T1W + r + tap % 3;
                const bf16x8_t bf = *reinterpret_cast<const bf16x8_t*>(t1 + pp * 64 + (((2 * kc + h) ^ ((pp >> 2) & 3)) * 16));
                acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2[tap], bf, acc2[mt], 0, 0, 0);
            }
        }
        if (kc == 0) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) w2[tap] = *reinterpret_cast<const bf16x8_t*>(w2pk + ((size_t)(2 + h) * (9 * 32) + tap * 32 + r) * 8);
        }
    }
    __syncthreads();   // the t1 tile is re-used as the output stage

    // ---- epilogue: bias + ReLU -> bf16 -> staged 64-byte-per-pixel stores ----
    unsigned char* stage = t1;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int tp = (wave * 2 + mt) * TW + r;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias2 + 8 * g + 4 * h);
            uint2 o;
            o.x = pack_bf16x2(apply_act(acc2[mt][4 * g + 0] + b4.x, ACT_RELU), apply_act(acc2[mt][4 * g + 1] + b4.y, ACT_RELU));
            o.y = pack_bf16x2(apply_act(acc2[mt][4 * g + 2] + b4.z, ACT_RELU), apply_act(acc2[mt][4 * g + 3] + b4.w, ACT_RELU));
            *reinterpret_cast<uint2*>(stage + tp * STAGE_PITCH + (8 * g + 4 * h) * 2) = o;
        }
    }
    __syncthreads();
    for (int i = tid; i < 256 * 4; i += 256) {
        const int tp = i >> 2, ch = i & 3;
        const int ty = tp / TW, tx = tp - ty * TW;
        const int oy = tile_y * TH + ty, ox = tile_x * TW + tx;
        if (oy >= p.Ho || ox >= p.Wo) continue;
        *reinterpret_cast<uint4*>(p.y + (((size_t)n_img * p.Ho + oy) * p.Wo + ox) * 32 + ch * 8) = *reinterpret_cast<const uint4*>(stage + tp * STAGE_PITCH + ch * 16);
    }
}

// standalone normalise: u8 [N,H,W,3] -> bf16 [N,Hp,Wp,4]-free NHWC(3) or NCHW, zero outside (vh, vw)
__global__ void normalize_kernel(const uint8_t* x, bf16_t* y, int N, int H, int W, int Hp, int Wp, int vh, int vw,
                                 float s0, float s1, float s2, float b0, float b1, float b2, int nchw) {
    const size_t total = (size_t)N * Hp * Wp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % Wp);
        const size_t t = i / Wp;
        const int yy = (int)(t % Hp), n = (int)(t / Hp);
        float v[3] = {0.f, 0.f, 0.f};
        if (yy < vh && xx < vw) {
            const uint8_t* s = x + (((size_t)n * H + yy) * W + xx) * 3;
            v[0] = (float)s[0] * s0; v[0] = v[0] + b0;
            v[1] = (float)s[1] * s1; v[1] = v[1] + b1;
            v[2] = (float)s[2] * s2; v[2] = v[2] + b2;
        }
        if (nchw) {
            const size_t plane = (size_t)Hp * Wp;
            bf16_t* d = y + (size_t)n * 3 * plane + (size_t)yy * Wp + xx;
            d[0] = f32_to_bf16(v[0]); d[plane] = f32_to_bf16(v[1]); d[2 * plane] = f32_to_bf16(v[2]);
        } else {
            bf16_t* d = y + i * 3;
            d[0] = f32_to_bf16(v[0]); d[1] = f32_to_bf16(v[1]); d[2] = f32_to_bf16(v[2]);
        }
    }
}

}  // namespace

void pack_stem_weights(const bf16_t* ohwi, int cout, bf16_t* out /* [2][2][32][8] */) {
    for (int ks = 0; ks < 2; ++ks)
        for (int h = 0; h < 2; ++h)
            for (int co = 0; co < 32; ++co)
                for (int j = 0; j < 8; ++j) {
                    const int k = ks * 16 + h * 8 + j;  // k = (kh*3 + kw)*3 + c == OHWI inner order
                    out[((ks * 2 + h) * 32 + co) * 8 + j] = (co < cout && k < 27) ? ohwi[co * 27 + k] : (bf16_t)0;
                }
}

hipError_t stem_conv_launch(const StemParams& p, hipStream_t stream) {
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    hipLaunchKernelGGL(stem_conv_kernel, dim3(p.N * tiles_x * tiles_y), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t stem12_launch(const StemParams& p, const bf16_t* w2pk, const float* bias2, hipStream_t stream) {
    if (p.valid_w_per_img != nullptr || p.Cout_store != 32 || p.act != ACT_RELU) return hipErrorInvalidValue;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    hipLaunchKernelGGL(stem12_kernel, dim3(p.N * tiles_x * tiles_y), dim3(256), 0, stream, p, w2pk, bias2);
    return hipGetLastError();
}

hipError_t normalize_launch(const uint8_t* x, bf16_t* y, int N, int H, int W, int Hp, int Wp, int vh, int vw,
                            const float* scale, const float* shift, int nchw, hipStream_t stream) {
    const size_t total = (size_t)N * Hp * Wp;
    int grid = (int)((total + 255) / 256);
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(normalize_kernel, dim3(grid), dim3(256), 0, stream, x, y, N, H, W, Hp, Wp, vh, vw, scale[0], scale[1],
                       scale[2], shift[0], shift[1], shift[2], nchw);
    return hipGetLastError();
}
