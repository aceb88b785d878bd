// 1x1 convolution / plain GEMM, A-stationary: the bandwidth-bound pointwise layers of DBNet (FPN laterals, stage-0
// shortcut, DBHead transposed convs) and of the recogniser (MobileNetV3 expand / project, LSTM x-projection).
//
// One 256-thread workgroup owns 128 consecutive pixels (flat index over N*H*W) and ALL output channels: the pixel tile
// [128 x K] is staged in LDS once, then the workgroup loops over 64-channel weight tiles (streamed through a small LDS
// buffer, prefetched in registers), so the activation is read from HBM/L2 exactly once instead of once per channel tile.
// Same MFMA orientation, plane LDS layout, packed weight image and fused epilogue (bias, residual / nearest-upsampled
// top-down add, activation, channel-slice write, 2x2 pixel-shuffle, fused DBHead tail) as conv_mfma.hip.
#include "conv_mfma.h"

namespace {

constexpr int PX = 128, BN = 64, PLANE_A = PX + 4, STAGE_PITCH = BN * 2 + 16, MAX_WIT = 9;  // K <= 288

__device__ __forceinline__ bf16x8_t ldsf(const unsigned char* p) { return *reinterpret_cast<const bf16x8_t*>(p); }

__global__ __launch_bounds__(256, 2) void conv1x1_kernel(const ConvParams p, const long long total_px) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int npl = p.Cin >> 3;
    unsigned char* sA = smem;
    unsigned char* sW = sA + (size_t)npl * PLANE_A * 16;
    unsigned char* stage = sW + (size_t)npl * BN * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const long long pix0 = (long long)blockIdx.x * PX;

    for (int i = tid; i < PX * npl; i += 256) {
        const int row = i / npl, c = i - row * npl;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (pix0 + row < total_px) v = *reinterpret_cast<const uint4*>(p.x + (size_t)(pix0 + row) * p.Cin + c * 8);
        *reinterpret_cast<uint4*>(sA + ((size_t)c * PLANE_A + row) * 16) = v;
    }
    const int w_items = BN * npl;
    uint4 wreg[MAX_WIT];
#define LOAD_W(nt_)                                                                                  \
    {                                                                                                \
        const uint4* src = reinterpret_cast<const uint4*>(p.wpk + (size_t)(nt_) * w_items * 8);      \
        _Pragma("unroll") for (int it = 0; it < MAX_WIT; ++it) {                                     \
            const int i = tid + 256 * it;                                                            \
            uint4 t_ = make_uint4(0, 0, 0, 0);                                                       \
            if (i < w_items) t_ = src[i];                                                            \
            wreg[it] = t_;                                                                           \
        }                                                                                            \
    }
    LOAD_W(0);

    // this lane's pixel (MFMA column) and its coordinates for the residual / pixel-shuffle addressing
    const long long mypix = pix0 + wave * 32 + r;
    const bool pvalid = mypix < total_px;
    const int hw = p.H * p.W;
    const int n_img = (int)(mypix / hw);
    const int prem = (int)(mypix - (long long)n_img * hw);
    const int oy = prem / p.W, ox = prem - oy * p.W;
    const int cout_r8 = (p.Cout + 7) & ~7;
    const int ksteps = p.Cin >> 4;
    const bool has_res = p.res != nullptr;
    const bf16_t* rrow = nullptr;
    if (has_res && pvalid)
        rrow = p.res + (((size_t)n_img * p.res_h + (oy >> p.res_shift)) * p.res_w + (ox >> p.res_shift)) * p.res_cstride;

    for (int nt = 0; nt < p.n_tiles; ++nt) {
        __syncthreads();  // previous tile's sW / stage readers are done (also orders the initial sA fill)
#pragma unroll
        for (int it = 0; it < MAX_WIT; ++it) {
            const int i = tid + 256 * it;
            if (i < w_items) *reinterpret_cast<uint4*>(sW + (size_t)i * 16) = wreg[it];
        }
        __syncthreads();
        if (nt + 1 < p.n_tiles) LOAD_W(nt + 1);
        // epilogue operands of this tile, requested before the MFMAs
        float4 bias_r[2][4];
        uint2 res_r[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cn = nt * BN + t * 32 + 8 * g + 4 * h;
                bias_r[t][g] = *reinterpret_cast<const float4*>(p.bias + cn);
                uint2 rv = make_uint2(0, 0);
                if (rrow != nullptr && cn < cout_r8) rv = *reinterpret_cast<const uint2*>(rrow + cn);
                res_r[t][g] = rv;
            }
        f32x16_t acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
        for (int kc = 0; kc < ksteps; ++kc) {
            const bf16x8_t bfr = ldsf(sA + ((size_t)(2 * kc + h) * PLANE_A + wave * 32 + r) * 16);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8_t afr = ldsf(sW + ((size_t)(2 * kc + h) * BN + t * 32 + r) * 16);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = bias_r[t][g];
                const uint2 rv = res_r[t][g];
                acc[t][4 * g + 0] += b4.x + __uint_as_float(rv.x << 16); acc[t][4 * g + 1] += b4.y + __uint_as_float(rv.x & 0xFFFF0000u);
                acc[t][4 * g + 2] += b4.z + __uint_as_float(rv.y << 16); acc[t][4 * g + 3] += b4.w + __uint_as_float(rv.y & 0xFFFF0000u);
            }
#define FOR_ACC(expr) _Pragma("unroll") for (int t = 0; t < 2; ++t) _Pragma("unroll") for (int j = 0; j < 16; ++j) { const float v = acc[t][j]; acc[t][j] = (expr); }
        if (p.act == ACT_RELU) { FOR_ACC(fmaxf(v, 0.f)) }
        else if (p.act == ACT_HSWISH) { FOR_ACC(v * fminf(fmaxf(v + 3.f, 0.f), 6.f) / 6.f) }
        else if (p.act == ACT_SIGMOID) { FOR_ACC(1.f / (1.f + expf(-v))) }
        else if (p.act == ACT_HSIGMOID) { FOR_ACC(fminf(fmaxf(0.2f * v + 0.5f, 0.f), 1.f)) }
#undef FOR_ACC
        const int tp = wave * 32 + r;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 o;
                o.x = pack_bf16x2(acc[t][4 * g + 0], acc[t][4 * g + 1]);
                o.y = pack_bf16x2(acc[t][4 * g + 2], acc[t][4 * g + 3]);
                *reinterpret_cast<uint2*>(stage + tp * STAGE_PITCH + (t * 32 + 8 * g + 4 * h) * 2) = o;
            }
        if (p.out_mode == OUT_CONVT && p.fuse_w != nullptr) {
            // fused DBHead tail: each wave re-reads only the 32 rows it staged itself (wave-local ordering, no barrier)
            f32x16_t d2;
#pragma unroll
            for (int j = 0; j < 16; ++j) d2[j] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8_t bfr = ldsf(stage + tp * STAGE_PITCH + (ks * 16 + h * 8) * 2);
                const bf16x8_t afr = *reinterpret_cast<const bf16x8_t*>(p.fuse_w + ((ks * 2 + h) * 32 + r) * 8);
                d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, d2, 0, 0, 0);
            }
            if (h == 0 && pvalid) {
                const int q = nt;  // convt_c == 64: one sub-pixel per channel tile
                const int yy = 4 * oy + 2 * (q >> 1), xx = 4 * ox + 2 * (q & 1);
                bf16_t* dst = p.y + ((size_t)n_img * (4 * p.H) + yy) * (size_t)(4 * p.W) + xx;
                *reinterpret_cast<uint32_t*>(dst) = pack_bf16x2(apply_act(d2[0] + p.fuse_b, ACT_SIGMOID), apply_act(d2[1] + p.fuse_b, ACT_SIGMOID));
                *reinterpret_cast<uint32_t*>(dst + 4 * p.W) = pack_bf16x2(apply_act(d2[2] + p.fuse_b, ACT_SIGMOID), apply_act(d2[3] + p.fuse_b, ACT_SIGMOID));
            }
            continue;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // 128 pixels x 8 chunks of 16 bytes
            const int i = tid + 256 * k;
            const int sp = i >> 3, ch = i & 7;
            const long long px = pix0 + sp;
            const int co = nt * BN + ch * 8;
            if (px >= total_px || co >= cout_r8) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(stage + sp * STAGE_PITCH + ch * 16);
            if (p.out_mode == OUT_NORMAL) {
                *reinterpret_cast<uint4*>(p.y + (size_t)px * p.y_cstride + p.y_coff + co) = v;
            } else {  // OUT_CONVT: 2x2 pixel shuffle
                const int ni = (int)(px / hw);
                const int pr = (int)(px - (long long)ni * hw);
                const int y0 = pr / p.W, x0 = pr - y0 * p.W;
                const int q = co / p.convt_c, cc = co - q * p.convt_c;
                bf16_t* dst = p.y + (((size_t)ni * (2 * p.H) + 2 * y0 + (q >> 1)) * (2 * p.W) + 2 * x0 + (q & 1)) * p.y_cstride + p.y_coff + cc;
                *reinterpret_cast<uint4*>(dst) = v;
            }
        }
    }
#undef LOAD_W
}

}  // namespace

bool conv1x1_supported(const ConvKernelCfg& cfg, const ConvParams& p) {
    return cfg.ks == 1 && cfg.stride == 1 && cfg.bn == 64 && p.Cin % 16 == 0 && p.Cin <= 288 && p.Cin >= 16 &&
           (p.out_mode == OUT_NORMAL || p.out_mode == OUT_CONVT) && p.Cout > 32;
}

hipError_t conv1x1_launch(ConvParams p, long long total_px, hipStream_t stream) {
    p.n_tiles = ceil_div(p.Cout, BN);
    const int npl = p.Cin / 8;
    const size_t lds = (size_t)npl * (PLANE_A + BN) * 16 + (size_t)PX * STAGE_PITCH;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    const long long blocks = (total_px + PX - 1) / PX;
    hipLaunchKernelGGL(conv1x1_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, p, total_px);
    return hipGetLastError();
}
