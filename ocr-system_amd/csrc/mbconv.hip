// MobileNetV3 block front half, fused: 1x1 expand conv (+ bias, activation) -> K x K depthwise conv (+ bias, activation),
// for the recogniser's inverted-residual blocks (crop feature maps of <= 16 rows x 160 columns).
//
// The expanded tensor is the widest one of a block (up to 288 channels at the block's input resolution) and has exactly one
// consumer; unfused it is written to and read back from HBM (~10 GB per 64-page step).  Here it only ever exists as an LDS
// tile:
//   * a workgroup walks (crop, 32-column strip) items.  Phase 1 computes the expanded, activated, bf16-rounded tensor of the
//     strip plus PAD columns either side, for all H rows, on the matrix cores (D[exp channel][pixel] = W x X, K = Cin <= 48:
//     B fragments straight from the NHWC input, A fragments from the packed weights in L1/L2) and writes it to LDS as
//     [row][column][channel] with a channel pitch of expC * 2 + 16 bytes; columns outside the image are ZERO (the depthwise
//     conv pads its input, not the expand conv's);
//   * phase 2 is dwconv_kernel's loop (ops.hip) reading its K + 3 input vectors per kernel row from that tile instead of HBM:
//     one thread = 4 output pixels x 8 channels, fp32 FMA in tap order, fp32 weights in LDS, bias + activation, 16-byte store.
// Arithmetic is identical to the unfused pair (same MFMA k order, same rounding points), so both paths produce the same
// bits; tests/test_gpu_rec.py checks that.
#include "mbconv.h"

namespace {

__device__ __forceinline__ void unpack8(const uint4 v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xFFFF0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xFFFF0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xFFFF0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xFFFF0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}

constexpr int MB_TW = 32, MB_KSMAX = 3;

template <int K, int SH, int ACT>
__global__ __launch_bounds__(256, 2) void mbconv_kernel(const MbParams p) {
    constexpr int PAD = K / 2, EW = MB_TW + 2 * PAD, XG = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int expc = p.expc, cg = expc >> 3;
    const int pitch = expc * 2 + 16;                     // bytes per tile pixel
    const int tile_bytes = (p.H * EW * pitch + 15) & ~15;
    float* wl = reinterpret_cast<float*>(smem + tile_bytes);  // depthwise weights [K*K][2][cg][4] fp32, then bias [expc]
    for (int i = tid; i < K * K * expc; i += 256) {
        const int tap = i / expc, c = i - tap * expc;
        wl[((tap * 2 + ((c >> 2) & 1)) * cg + (c >> 3)) * 4 + (c & 3)] = bf16_to_f32(p.wd[i]);
    }
    float* bl = wl + K * K * expc;
    for (int i = tid; i < expc; i += 256) bl[i] = p.bd[i];

    const int ksteps = p.cin >> 4;
    const int mtiles = (expc + 31) >> 5;
    const int npix = p.H * EW, ptiles = (npix + 31) >> 5;
    const int strips_x = (p.W + MB_TW - 1) / MB_TW;
    const int nitems = p.N * strips_x;

    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int n = item / strips_x, x0 = (item - n * strips_x) * MB_TW;
        __syncthreads();  // the previous strip's phase 2 (and the weight fill) is done with LDS
        // ---------------- phase 1: expand on the matrix cores -> LDS tile ----------------
        for (int pt = wave; pt < ptiles; pt += 4) {
            const int pi = pt * 32 + r;
            const bool valid = pi < npix;
            const int row = pi / EW, col = pi - row * EW;
            const int gx = x0 - PAD + col;
            const bool inimg = valid && gx >= 0 && gx < p.W;
            bf16x8_t bfr[MB_KSMAX];
#pragma unroll
            for (int ks = 0; ks < MB_KSMAX; ++ks) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (ks < ksteps && inimg) v = *reinterpret_cast<const uint4*>(p.x + (((size_t)n * p.H + row) * p.W + gx) * p.cin + ks * 16 + h * 8);
                bfr[ks] = *reinterpret_cast<const bf16x8_t*>(&v);
            }
            unsigned char* trow = smem + (size_t)pi * pitch;
            for (int mt = 0; mt < mtiles; ++mt) {
                f32x16_t acc;
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
                for (int ks = 0; ks < MB_KSMAX; ++ks)
                    if (ks < ksteps) {
                        const bf16x8_t afr = *reinterpret_cast<const bf16x8_t*>(p.we + ((((size_t)mt * ksteps + ks) * 2 + h) * 32 + r) * 8);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr[ks], acc, 0, 0, 0);
                    }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = mt * 32 + 8 * g + 4 * h;
                    if (!valid || c >= expc) continue;
                    uint2 o = make_uint2(0, 0);
                    if (inimg) {
                        const float4 b4 = *reinterpret_cast<const float4*>(p.be + c);
                        o.x = pack_bf16x2(apply_act(acc[4 * g + 0] + b4.x, ACT), apply_act(acc[4 * g + 1] + b4.y, ACT));
                        o.y = pack_bf16x2(apply_act(acc[4 * g + 2] + b4.z, ACT), apply_act(acc[4 * g + 3] + b4.w, ACT));
                    }
                    *reinterpret_cast<uint2*>(trow + c * 2) = o;
                }
            }
        }
        __syncthreads();
        // ---------------- phase 2: depthwise K x K, stride (SH, 1), from the LDS tile ----------------
        // thread (pg, c8) owns channel group c8 and walks the pixel groups t = pg, pg + groups, ... (the summation order of the
        // squeeze-excite pooling, mbconv.h)
        const int groups = 256 / cg, pg = tid / cg, c8_fixed = tid - pg * cg;
        float psum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int t = pg; t < p.Ho * (MB_TW / XG) && pg < groups; t += groups) {
            // the channel group is the same in every iteration, but the compiler must not know: it would hoist the group's bias and
            // depthwise weights out of the loop (+58 registers: the kernel drops from three resident workgroups per CU to two, +24 % time)
            int c8 = c8_fixed;
            asm volatile("" : "+v"(c8));
            const int xg = t % (MB_TW / XG), oy = t / (MB_TW / XG);
            const int ox0 = xg * XG;  // strip-local output column; tile column ox0 + j is image column x0 + ox0 - PAD + j
            if (x0 + ox0 >= p.W) continue;
            float a[XG][8];
#pragma unroll
            for (int o = 0; o < XG; ++o)
#pragma unroll
                for (int j = 0; j < 8; ++j) a[o][j] = 0.f;
#pragma unroll
            for (int kh = 0; kh < K; ++kh) {
                const int iy = oy * SH - PAD + kh;
                if (iy < 0 || iy >= p.H) continue;
                float in[K + XG - 1][8];
                const unsigned char* src = smem + (size_t)(iy * EW + ox0) * pitch + c8 * 16;
#pragma unroll
                for (int j = 0; j < K + XG - 1; ++j) unpack8(*reinterpret_cast<const uint4*>(src + j * pitch), in[j]);
#pragma unroll
                for (int kw = 0; kw < K; ++kw) {
                    const float4 g0 = *reinterpret_cast<const float4*>(wl + (((kh * K + kw) * 2 + 0) * cg + c8) * 4);
                    const float4 g1 = *reinterpret_cast<const float4*>(wl + (((kh * K + kw) * 2 + 1) * cg + c8) * 4);
                    const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                    for (int o = 0; o < XG; ++o)
#pragma unroll
                        for (int j = 0; j < 8; ++j) a[o][j] = __builtin_fmaf(in[o + kw][j], g[j], a[o][j]);
                }
            }
            float bb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) bb[j] = bl[c8 * 8 + j];
            bf16_t* yrow = p.d + (((size_t)n * p.Ho + oy) * p.W + x0 + ox0) * expc + c8 * 8;
#pragma unroll
            for (int o = 0; o < XG; ++o) {
                if (x0 + ox0 + o >= p.W) break;
#pragma unroll
                for (int j = 0; j < 8; ++j) a[o][j] = apply_act(a[o][j] + bb[j], ACT);
                const uint4 v = pack8(a[o]);
                *reinterpret_cast<uint4*>(yrow + (size_t)o * expc) = v;
                if (p.pool != nullptr) {   // pooled sum of the values as stored
                    float f[8];
                    unpack8(v, f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) psum[j] += f[j];
                }
            }
        }
        if (p.pool != nullptr) {   // (uniform branch) groups' sums -> LDS (the tile is dead) -> one fixed-order sum per channel
            __syncthreads();
            float* red = reinterpret_cast<float*>(smem);
            if (pg < groups) {
#pragma unroll
                for (int j = 0; j < 8; ++j) red[pg * expc + c8_fixed * 8 + j] = psum[j];
            }
            __syncthreads();
            for (int c = tid; c < expc; c += 256) {
                float s = 0.f;
                for (int g = 0; g < groups; ++g) s += red[g * expc + c];
                p.pool[((size_t)n * strips_x + (x0 / MB_TW)) * expc + c] = s;
            }
        }
    }
}

// the same sums from a stored tensor (the unfused expand + depthwise path): one workgroup per (crop, strip), identical order
__global__ __launch_bounds__(256) void se_pool_kernel(const bf16_t* d, float* pool, int N, int Ho, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XG = 4;
    const int tid = threadIdx.x, cg = C >> 3, groups = 256 / cg, pg = tid / cg, c8 = tid - pg * cg;
    const int strips_x = (W + MB_TW - 1) / MB_TW;
    const int n = blockIdx.x / strips_x, x0 = (blockIdx.x - n * strips_x) * MB_TW;
    float psum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t = pg; t < Ho * (MB_TW / XG) && pg < groups; t += groups) {
        const int xg = t % (MB_TW / XG), oy = t / (MB_TW / XG);
        const int ox0 = xg * XG;
        if (x0 + ox0 >= W) continue;
        const bf16_t* yrow = d + (((size_t)n * Ho + oy) * W + x0 + ox0) * C + c8 * 8;
#pragma unroll
        for (int o = 0; o < XG; ++o) {
            if (x0 + ox0 + o >= W) break;
            float f[8];
            unpack8(*reinterpret_cast<const uint4*>(yrow + (size_t)o * C), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) psum[j] += f[j];
        }
    }
    float* red = reinterpret_cast<float*>(smem);
    if (pg < groups) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[pg * C + c8 * 8 + j] = psum[j];
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
        for (int g = 0; g < groups; ++g) s += red[g * C + c];
        pool[((size_t)n * strips_x + (x0 / MB_TW)) * C + c] = s;
    }
}

}  // namespace

size_t mbconv_expand_packed_elems(int expc, int cin) { return (size_t)((expc + 31) / 32) * (cin / 16) * 2 * 32 * 8; }

void mbconv_pack_expand(const bf16_t* w /*[expc][cin]*/, int expc, int cin, bf16_t* out) {
    const int mtiles = (expc + 31) / 32, ksteps = cin / 16;
    for (int mt = 0; mt < mtiles; ++mt)
        for (int ks = 0; ks < ksteps; ++ks)
            for (int hh = 0; hh < 2; ++hh)
                for (int rr = 0; rr < 32; ++rr)
                    for (int j = 0; j < 8; ++j) {
                        const int row = mt * 32 + rr, k = ks * 16 + hh * 8 + j;
                        out[((((size_t)mt * ksteps + ks) * 2 + hh) * 32 + rr) * 8 + j] = row < expc ? w[(size_t)row * cin + k] : (bf16_t)0;
                    }
}

bool mbconv_supported(const MbParams& p, int k, int sh) {
    if (!(k == 3 || k == 5) || !(sh == 1 || sh == 2)) return false;
    if (p.cin % 16 != 0 || p.cin > 16 * MB_KSMAX || p.expc % 16 != 0 || p.expc % 8 != 0) return false;
    if (p.act != ACT_RELU && p.act != ACT_HSWISH) return false;
    return mbconv_lds_bytes(p, k) <= 78 * 1024;  // two workgroups per CU
}

size_t mbconv_lds_bytes(const MbParams& p, int k) {
    const int ew = MB_TW + 2 * (k / 2), pitch = p.expc * 2 + 16;
    return (((size_t)p.H * ew * pitch + 15) & ~(size_t)15) + ((size_t)k * k * p.expc + p.expc) * sizeof(float);
}

hipError_t se_pool_launch(const bf16_t* d, float* pool, int N, int Ho, int W, int C, hipStream_t st) {
    if (C % 8 != 0 || C / 8 > 256 || C < 8) return hipErrorInvalidValue;
    const size_t lds = (size_t)(256 / (C / 8)) * C * sizeof(float);
    hipLaunchKernelGGL(se_pool_kernel, dim3(N * mb_strips(W)), dim3(256), lds, st, d, pool, N, Ho, W, C);
    return hipGetLastError();
}

hipError_t mbconv_launch(const MbParams& p, int k, int sh, hipStream_t st) {
    if (!mbconv_supported(p, k, sh)) return hipErrorInvalidValue;
    const size_t lds = mbconv_lds_bytes(p, k);
    const int items = p.N * ((p.W + MB_TW - 1) / MB_TW);
    const int grid = items < 4096 ? items : 4096;
#define MB_LAUNCH(K_, SH_, A_)                                                                                                    \
    {                                                                                                                            \
        auto kern = mbconv_kernel<K_, SH_, A_>;                                                                                  \
        { hipError_t e = locr_dyn_lds(reinterpret_cast<const void*>(kern), 80 * 1024); if (e != hipSuccess) return e; }                                                                                                                        \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, p);                                                             \
    }
#define MB_ACT(K_, SH_) { if (p.act == ACT_RELU) MB_LAUNCH(K_, SH_, ACT_RELU) else MB_LAUNCH(K_, SH_, ACT_HSWISH) }
    if (k == 3 && sh == 1) MB_ACT(3, 1)
    else if (k == 3 && sh == 2) MB_ACT(3, 2)
    else if (k == 5 && sh == 1) MB_ACT(5, 1)
    else MB_ACT(5, 2)
#undef MB_ACT
#undef MB_LAUNCH
    return hipGetLastError();
}
