// Pointwise (1x1) convolution for the narrow-K, wide-N layers of DBNet (Cin 64 / 128 -> Cout up to 256: FPN laterals on c2 /
// c3, the stage-0 shortcut, DBHead's transposed convs), pixel-stationary:
//   * one 256-thread workgroup owns 256 consecutive pixels (flat index over N*H*W) and ALL output channels.  The [256 x Cin]
//     pixel tile is copied global -> LDS ONCE by global_load_lds_dwordx4 (lane-linear image [pixel][Cin/8 slices]; the slice
//     order of a pixel is XOR-swizzled through the per-lane SOURCE address so that the stride-Cin fragment reads are
//     bank-conflict free), so the activation is read from HBM once instead of once per 64-channel tile;
//   * the workgroup then walks the output channels in groups of 64.  Weight fragments (MFMA A operand) come straight from
//     the packed weight image in L1/L2 into registers — the whole layer's weights are 8-64 KB and shared by every workgroup —
//     and the next group's fragments are requested before the current group's epilogue: no barrier inside the loop, the four
//     waves run independently (wave w owns pixels 64w .. 64w+63 = two 32-pixel MFMA column tiles);
//   * epilogue per group, in registers: bias, nearest-upsampled top-down add (FPN), activation; then NHWC stores through a
//     wave-local LDS stage (full 128-byte lines per pixel; Cin = 128: v_permlane32_swap + 16-byte stores), or (DBHead) the
//     fused transposed-conv tail: the activated 64-channel tile is
//     staged wave-locally in LDS, contracted with the 64->4 weights on the matrix cores, sigmoid, 2x2 probability block.
// Same arithmetic (fp32 MFMA accumulation over k in the same order, one bf16 rounding per stored tensor), same packed
// weights (bn 64, ck 32) and the same ConvParams as conv_mfma.hip, which remains the general 1x1 path.
#include "conv_mfma.h"

namespace {

__device__ __forceinline__ bf16x8_t ldsf(const unsigned char* p) { return *reinterpret_cast<const bf16x8_t*>(p); }

#define GLDS16(src_, dst_) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_), (__attribute__((address_space(3))) void*)(dst_), 16, 0, 0)

constexpr int PW_PX = 256, PW_STAGE_PITCH = 64 * 2 + 16;

// MODE: 0 = NHWC output, 1 = NHWC output + top-down add, 2 = fused DBHead tail (compile-time so that each variant only keeps
// the registers it needs)
template <int CIN, int MODE>
__global__ __launch_bounds__(256, 2) void conv_pw_kernel(const ConvParams p, const long long total_px) {
    constexpr int NS = CIN / 8, KSTEPS = CIN / 16, A_BYTES = PW_PX * CIN * 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;
    unsigned char* stage = smem + A_BYTES;  // wave-local rows: output staging / the fused DBHead tail's operand
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const long long pix0 = (long long)blockIdx.x * PW_PX;

    // ---- pixel tile -> LDS (DMA); slot s of pixel px holds channel slice s ^ swz(px) ----
#pragma unroll
    for (int it = 0; it < NS; ++it) {
        const int i = tid + 256 * it;
        const int px = i / NS, s = i - px * NS;
        const int slice = s ^ (NS == 8 ? ((px >> 1) & 7) : (px & (NS - 1)));
        const long long gp = pix0 + px;
        const bf16_t* src = gp < total_px ? p.x + (size_t)gp * CIN + slice * 8 : p.zeros;
        GLDS16(src, sA + (i - lane) * 16);
    }

    // ---- per-lane pixel coordinates (two MFMA column tiles) ----
    const int hw = p.H * p.W;
    long long gpix[2];
    bool pvalid[2];
    int n_img[2], oy[2], ox[2], aoff[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int tp = wave * 64 + mt * 32 + r;
        gpix[mt] = pix0 + tp;
        pvalid[mt] = gpix[mt] < total_px;
        n_img[mt] = (int)(gpix[mt] / hw);
        const int prem = (int)(gpix[mt] - (long long)n_img[mt] * hw);
        oy[mt] = prem / p.W; ox[mt] = prem - oy[mt] * p.W;
        aoff[mt] = tp * CIN * 2;
    }
    const int swz[2] = {NS == 8 ? (((wave * 64 + r) >> 1) & 7) : ((wave * 64 + r) & (NS - 1)),
                        NS == 8 ? (((wave * 64 + 32 + r) >> 1) & 7) : ((wave * 64 + 32 + r) & (NS - 1))};

    const int nchunks = CIN / 32;
    const int ngroups = (p.Cout + 63) / 64;
    const int cout_r8 = (p.Cout + 7) & ~7;
    constexpr bool has_res = MODE == 1, fused = MODE == 2;

    // weight fragments of one 64-channel group: [kstep][nt]
    bf16x8_t afr[KSTEPS][2];
#define LOAD_W_RANGE(q_, ks0_, ks1_)                                                                                        \
    _Pragma("unroll") for (int ks = (ks0_); ks < (ks1_); ++ks)                                                             \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                                   \
            afr[ks][nt] = *reinterpret_cast<const bf16x8_t*>(p.wpk + ((((size_t)(q_) * nchunks + (ks >> 1)) * 4 + 2 * (ks & 1) + h) * 64 + nt * 32 + r) * 8);
#define LOAD_W(q_) LOAD_W_RANGE(q_, 0, KSTEPS)
    // Cin = 128: a whole group of fragments (64 registers) next to the accumulators, the bias and the residual of the epilogue does
    // not fit 256 registers (hipcc spilled 34 of them to scratch inside the MFMA loop): the second half is requested once the
    // bias / residual registers are dead; the next group's first MFMAs only need the first half
    constexpr int KS_EARLY = CIN > 64 ? KSTEPS / 2 : KSTEPS;
    LOAD_W(0)
    // fused tail: the 64 -> 4 weights of the last transposed conv are the same for every group and pixel: once per workgroup
    // (inside the loop hipcc cannot hoist them past the stores and re-reads them 32 times per tile)
    bf16x8_t w3fr[4];
    if constexpr (MODE == 2) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) w3fr[ks] = *reinterpret_cast<const bf16x8_t*>(p.fuse_w + ((ks * 2 + h) * 32 + r) * 8);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int q = 0; q < ngroups; ++q) {
        // epilogue operands of this group first (one latency, hidden behind the MFMAs)
        float4 bias_r[2][4];
        uint2 res_r[has_res ? 2 : 1][2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) bias_r[nt][g] = *reinterpret_cast<const float4*>(p.bias + q * 64 + nt * 32 + 8 * g + 4 * h);
        if constexpr (has_res) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const bf16_t* rrow = p.res + (((size_t)n_img[mt] * p.res_h + (oy[mt] >> p.res_shift)) * p.res_w + (ox[mt] >> p.res_shift)) * p.res_cstride + q * 64;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int cn = nt * 32 + 8 * g + 4 * h;
                        uint2 rv = make_uint2(0, 0);
                        if (pvalid[mt] && q * 64 + cn < cout_r8) rv = *reinterpret_cast<const uint2*>(rrow + cn);
                        res_r[mt][nt][g] = rv;
                    }
            }
        }
        f32x16_t acc[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[mt][nt][j] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            bf16x8_t bfr[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) bfr[mt] = ldsf(sA + aoff[mt] + (((2 * ks + h) ^ swz[mt]) * 16));
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][nt], bfr[mt], acc[mt][nt], 0, 0, 0);
        }
        if (q + 1 < ngroups) { LOAD_W_RANGE(q + 1, 0, KS_EARLY) }  // next group's weights fly under this group's epilogue

#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b4 = bias_r[nt][g];
                    acc[mt][nt][4 * g + 0] += b4.x; acc[mt][nt][4 * g + 1] += b4.y;
                    acc[mt][nt][4 * g + 2] += b4.z; acc[mt][nt][4 * g + 3] += b4.w;
                    if constexpr (has_res) {
                        const uint2 rv = res_r[mt][nt][g];
                        acc[mt][nt][4 * g + 0] += __uint_as_float(rv.x << 16); acc[mt][nt][4 * g + 1] += __uint_as_float(rv.x & 0xFFFF0000u);
                        acc[mt][nt][4 * g + 2] += __uint_as_float(rv.y << 16); acc[mt][nt][4 * g + 3] += __uint_as_float(rv.y & 0xFFFF0000u);
                    }
                }
        if (p.act == ACT_RELU) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[mt][nt][j] = fmaxf(acc[mt][nt][j], 0.f);
        } else if (p.act != ACT_NONE) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[mt][nt][j] = apply_act(acc[mt][nt][j], p.act);
        }

        if constexpr (KS_EARLY < KSTEPS) {
            __builtin_amdgcn_sched_barrier(0);   // (the scheduler would hoist these loads above the adds and bring the pressure back)
            if (q + 1 < ngroups) { LOAD_W_RANGE(q + 1, KS_EARLY, KSTEPS) }
            __builtin_amdgcn_sched_barrier(0);
        }

        if constexpr (!fused) {
            if constexpr (CIN != 64) {
                // direct stores: v_permlane32_swap gives every lane 8 consecutive channels, 32 bytes per pixel per instruction
                // (Cin = 128: the 64 KB pixel tile leaves no LDS for a stage at two workgroups per CU)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int gp = 0; gp < 2; ++gp) {
                            const int g0 = 2 * gp, g1 = 2 * gp + 1;
                            const uint32_t q0x = pack_bf16x2(acc[mt][nt][4 * g0 + 0], acc[mt][nt][4 * g0 + 1]), q0y = pack_bf16x2(acc[mt][nt][4 * g0 + 2], acc[mt][nt][4 * g0 + 3]);
                            const uint32_t q1x = pack_bf16x2(acc[mt][nt][4 * g1 + 0], acc[mt][nt][4 * g1 + 1]), q1y = pack_bf16x2(acc[mt][nt][4 * g1 + 2], acc[mt][nt][4 * g1 + 3]);
                            const auto sx = __builtin_amdgcn_permlane32_swap(q0x, q1x, false, false);
                            const auto sy = __builtin_amdgcn_permlane32_swap(q0y, q1y, false, false);
                            const int co = q * 64 + nt * 32 + 16 * gp + 8 * h;
                            if (pvalid[mt] && co < cout_r8)
                                *reinterpret_cast<uint4*>(p.y + (size_t)gpix[mt] * p.y_cstride + p.y_coff + co) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                        }
            } else {
                // wave-local LDS stage -> every store instruction writes 8 pixels x 128 contiguous bytes: full lines instead of four
                // 32-byte pieces per line (measured on fpn.in2, 1.5 GB written per 16 pages: 0.68 -> 0.55 ms)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int tp = wave * 64 + mt * 32 + r;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            uint2 o;
                            o.x = pack_bf16x2(acc[mt][nt][4 * g + 0], acc[mt][nt][4 * g + 1]);
                            o.y = pack_bf16x2(acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]);
                            *reinterpret_cast<uint2*>(stage + tp * PW_STAGE_PITCH + (nt * 32 + 8 * g + 4 * h) * 2) = o;
                        }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = lane + 64 * k;           // 64 pixels x 8 chunks of this wave
                    const int sp = wave * 64 + (i >> 3), ch = i & 7;
                    const long long gp = pix0 + sp;
                    const int co = q * 64 + ch * 8;
                    const uint4 v = *reinterpret_cast<const uint4*>(stage + sp * PW_STAGE_PITCH + ch * 16);
                    if (gp < total_px && co < cout_r8) *reinterpret_cast<uint4*>(p.y + (size_t)gp * p.y_cstride + p.y_coff + co) = v;
                }
            }
        } else {
            // fused DBHead tail: group q is sub-pixel q of the first transposed conv (convt_c == 64).  The wave stages its own 64
            // activated pixels x 64 channels and contracts them with the 64 -> 4 weights of the second transposed conv.
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int tp = wave * 64 + mt * 32 + r;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        uint2 o;
                        o.x = pack_bf16x2(acc[mt][nt][4 * g + 0], acc[mt][nt][4 * g + 1]);
                        o.y = pack_bf16x2(acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]);
                        *reinterpret_cast<uint2*>(stage + tp * PW_STAGE_PITCH + (nt * 32 + 8 * g + 4 * h) * 2) = o;
                    }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int tp = wave * 64 + mt * 32 + r;
                f32x16_t d2;
#pragma unroll
                for (int j = 0; j < 16; ++j) d2[j] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8_t bfr = ldsf(stage + tp * PW_STAGE_PITCH + (ks * 16 + h * 8) * 2);
                    d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3fr[ks], bfr, d2, 0, 0, 0);
                }
                if (h == 0 && pvalid[mt]) {
                    const int yy = 4 * oy[mt] + 2 * (q >> 1), xx = 4 * ox[mt] + 2 * (q & 1);
                    bf16_t* dst = p.y + ((size_t)n_img[mt] * (4 * p.H) + yy) * (size_t)(4 * p.W) + xx;
                    *reinterpret_cast<uint32_t*>(dst) = pack_bf16x2(apply_act(d2[0] + p.fuse_b, ACT_SIGMOID), apply_act(d2[1] + p.fuse_b, ACT_SIGMOID));
                    *reinterpret_cast<uint32_t*>(dst + 4 * p.W) = pack_bf16x2(apply_act(d2[2] + p.fuse_b, ACT_SIGMOID), apply_act(d2[3] + p.fuse_b, ACT_SIGMOID));
                }
            }
        }
    }
#undef LOAD_W
#undef LOAD_W_RANGE
}

template <int CIN, int MODE>
hipError_t launch_pw(const ConvParams& p, long long total_px, hipStream_t stream) {
    auto kern = conv_pw_kernel<CIN, MODE>;
    const size_t lds = (size_t)PW_PX * CIN * 2 + (CIN == 64 ? (size_t)PW_PX * PW_STAGE_PITCH : 0);
    { hipError_t e = locr_dyn_lds(reinterpret_cast<const void*>(kern), PW_PX * CIN * 2 + PW_PX * PW_STAGE_PITCH); if (e != hipSuccess) return e; }
    const long long blocks = (total_px + PW_PX - 1) / PW_PX;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, p, total_px);
    return hipGetLastError();
}

}  // namespace

bool conv_pw_supported(const ConvKernelCfg& cfg, const ConvParams& p) {
    if (cfg.ks != 1 || cfg.stride != 1 || cfg.bn != 64 || cfg.ck != 32 || cfg.nw != 4) return false;
    if (p.Cin != 64 && p.Cin != 128) return false;
    if (p.gate != nullptr || p.zeros == nullptr || p.Cout < 64) return false;
    if (p.out_mode == OUT_NORMAL) return p.res == nullptr || p.pix_limit == 0;  // the top-down add needs real (y, x) coordinates
    if (p.out_mode == OUT_CONVT) return p.fuse_w != nullptr && p.convt_c == 64 && p.Cin == 64 && p.res == nullptr && p.pix_limit == 0 && p.Cout % 64 == 0;
    return false;
}

hipError_t conv_pw_launch(const ConvParams& p, hipStream_t stream) {
    const long long total = p.pix_limit ? (long long)p.pix_limit : (long long)p.N * p.H * p.W;
    if (total <= 0 || (total + PW_PX - 1) / PW_PX > 0x7fffffffLL) return hipErrorInvalidValue;
    if (p.out_mode == OUT_CONVT) return launch_pw<64, 2>(p, total, stream);
    if (p.Cin == 64) return p.res ? launch_pw<64, 1>(p, total, stream) : launch_pw<64, 0>(p, total, stream);
    return p.res ? launch_pw<128, 1>(p, total, stream) : launch_pw<128, 0>(p, total, stream);
}
