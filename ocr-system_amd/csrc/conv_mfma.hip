// Implicit-GEMM convolution for gfx950 (CDNA4): NHWC bf16 activations, OHWI bf16 weights,
// fp32 MFMA accumulation, fused bias + residual / top-down add + activation epilogue.
//
// Replaces the engine slot of the reference provider (there is no local convolution code in
// the reference: /root/reference/backend/services/ocr_service.py:213-246 is a remote call).
//
// Structure (one 256-thread workgroup = 4 waves):
//   * output tile  = TH x TW (=32) pixels  x  BN output channels; wave w owns MT pixel rows (MT 32-pixel MFMA column
//     tiles) x all BN/32 row tiles.  Variants: MT = 2 (8x32 tile, any kernel size / stride) and MT = 4 (16x32 tile, 3x3 / s1
//     with >= 64 input channels, picked per launch when the grid still fills the chip: less LDS and L2 traffic per MFMA).
//   * the K loop walks input-channel chunks of CK; per chunk the (TH-1)*S+KS x (TW-1)*S+KS input halo tile is staged ONCE
//     in LDS and re-used by all KS*KS taps (im2col never touches HBM), together with the chunk's [tap][BN][CK] weight slab
//     (pre-packed on the host in the exact LDS order, so its loads are fully coalesced).
//   * staging, DB == 0: global -> registers (issued before the current chunk's MFMAs) -> ds_write after the barrier, LDS
//     images in "plane" layout: plane c holds the c-th 16-byte (8-channel) slice of every halo pixel / weight row.  A 32x32x16
//     MFMA fragment read (lane = pixel or cout, 8 consecutive k) is a unit-stride ds_read_b128: conflict-free without
//     swizzle; the A plane stride is == 4 (mod 16) entries so the 4-lanes-per-pixel staging writes are conflict-free too.
//   * staging, DB == 3 (16x32 tile, CK = 16): global_load_lds_dwordx4 straight into a 2-deep LDS ring (no VGPR round trip, no
//     ds_write pass, ONE barrier per chunk).  The LDS image is the lane-linear one a wave's DMA writes, [pixel][2 slices];
//     with 16-channel chunks a fragment read is still one contiguous 1 KB.  Out-of-image halo pixels read a block of zeros.
//   * fragment reads are software-pipelined one (tap, k-step) ahead of the MFMAs (sched_group_barrier interleave).
//   * orientation: D[cout][pixel] = W[cout][k] * X[k][pixel]  (weights are the MFMA A operand), so each lane ends up with
//     4 consecutive output channels of one pixel per accumulator quad.  Epilogue: bias + residual / top-down add + activation
//     in registers, then either packed 8-byte LDS staging -> 16-byte coalesced stores (all output modes, fused DBHead tail),
//     or, for plain NHWC output of the register-staged 3x3 kernels, v_permlane32_swap + 16-byte stores with no LDS/barrier.
//   * two workgroups per CU (BN <= 64) overlap each other's barriers; block ids are remapped per XCD.
// Measured dead ends (removed from the tree): 8-wave double-buffered LDS, 8-wave ping-pong groups,
// A-stationary 1x1 kernel, persistent multi-tile LDS-DMA kernel, 64-channel chunks for 1x1.
#include "conv_mfma.h"

#include <cstdlib>

namespace {

template <int KS, int S, int BN, int CK, int TW, int NW = 4, int DB = 0, int MT = 2>
struct Cfg {
    static constexpr int NTHR = NW * 64;
    static constexpr bool DMA = (DB == 3 || DB == 4);  // operands arrive by LDS-DMA (global_load_lds) into a 2-deep LDS ring
    static constexpr bool POOL = (DB == 4);            // ... and the epilogue is the fused 3x3/s2 max pool (its own instantiation)
    static constexpr bool SC = (DB == 5);              // 3x3 / stride 2 with the block's 2x2 / stride-2 shortcut conv computed from the same halo (ConvParams::wpk2)
    static_assert(!SC || (KS == 3 && S == 2 && CK == 16), "fused shortcut: the 3x3 / stride-2 block-entry kernel");
    static constexpr int TH = NW * MT;              // TW == 32: every wave owns MT pixel rows (MT 32-pixel MFMA column tiles)
    static constexpr int PAD = (KS == 3) ? 1 : 0;
    static constexpr int HH = (TH - 1) * S + KS;
    static constexpr int HW = (TW - 1) * S + KS;
    static constexpr int HWP = HW;
    static constexpr int NPL = CK / 8;
    static constexpr int HPX = HH * HWP;
    static constexpr int PMOD = (NPL >= 8) ? 2 : 4;
    static constexpr int PLANE_A = HPX + ((PMOD - HPX % 16) + 16) % 16;
    static constexpr int TAPS = KS * KS;
    static constexpr int WROWS = TAPS * BN;
    static constexpr int PLANE_W = WROWS;
    static constexpr int NT = BN / 32;
    static constexpr int A_BYTES = DMA ? HPX * NPL * 16 : PLANE_A * NPL * 16;
    static constexpr int W_BYTES = PLANE_W * NPL * 16;
    static constexpr int W2_ROWS = SC ? 4 * BN : 0, W2_ITEMS = W2_ROWS * NPL, W2IT = (W2_ITEMS + NTHR - 1) / NTHR, W2_BYTES = W2_ITEMS * 16;
    // LDS addressing of one 16-byte entry (pixel or weight row, 8-channel slice): the plane layout keeps a slice's entries
    // together (register staging scatters into it); the DMA layout is the lane-linear image a wave's global_load_lds writes,
    // [pixel][slice] — with CK == 16 (two slices) a fragment read is still one contiguous, conflict-free 1 KB
    static constexpr int A_PIX = DMA ? NPL * 16 : 16, A_SL = DMA ? 16 : PLANE_A * 16;
    static constexpr int W_ROW = DMA ? NPL * 16 : 16, W_SL = DMA ? 16 : PLANE_W * 16;
    static constexpr int STAGE_PITCH = BN * 2 + 16;
    static constexpr int STAGE_BYTES = TH * TW * STAGE_PITCH;
    static constexpr int BUF_BYTES = A_BYTES + W_BYTES + W2_BYTES;
    static constexpr int LDS_BYTES = (BUF_BYTES * (DMA ? 2 : 1)) > STAGE_BYTES ? (BUF_BYTES * (DMA ? 2 : 1)) : STAGE_BYTES;
    static_assert(!DMA || (CK == 16 && KS == 3 && S == 1), "DMA variant: 3x3 / stride 1 / 16-channel chunks");
    static constexpr int A_ITEMS = HH * HW * NPL;
    static constexpr int AIT = (A_ITEMS + NTHR - 1) / NTHR;
    static constexpr int W_ITEMS = WROWS * NPL;
    static constexpr int WIT = (W_ITEMS + NTHR - 1) / NTHR;
    static constexpr int CPP = BN / 8;  // 16-byte chunks per pixel in the store pass
};

__device__ __forceinline__ bf16x8_t lds_frag(const unsigned char* p) { return *reinterpret_cast<const bf16x8_t*>(p); }

// 8 x (bf16 * bf16 -> fp32 -> bf16, round to nearest even): the fused squeeze-excite scaling
__device__ __forceinline__ uint32_t mul_bf16x2(uint32_t a, uint32_t b) {
    const float lo = __uint_as_float(a << 16) * __uint_as_float(b << 16);
    const float hi = __uint_as_float(a & 0xFFFF0000u) * __uint_as_float(b & 0xFFFF0000u);
    return pack_bf16x2(lo, hi);
}
__device__ __forceinline__ uint4 mul_bf16x8(uint4 a, uint4 b) {
    return make_uint4(mul_bf16x2(a.x, b.x), mul_bf16x2(a.y, b.y), mul_bf16x2(a.z, b.z), mul_bf16x2(a.w, b.w));
}

template <int KS, int S, int BN, int CK, int TW, int NW, int DB, int MT>
__global__ __launch_bounds__(NW * 64, (BN <= 64 ? 2 : 1)) void conv_mfma_kernel(const ConvParams p) {
    using C = Cfg<KS, S, BN, CK, TW, NW, DB, MT>;
    constexpr int NTHR = C::NTHR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;  // wave w owns pixel rows MT*w .. MT*w + MT - 1
    unsigned char* sA = smem;
    unsigned char* sW = sA + C::A_BYTES;
    const int r = lane & 31, h = lane >> 5;

    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = lid % p.n_tiles;
    const int mtile = lid / p.n_tiles;
    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int n_img = mtile / tiles_per_img;
    const int trem = mtile - n_img * tiles_per_img;
    const int tile_y = trem / p.tiles_x, tile_x = trem - tile_y * p.tiles_x;

    const bf16_t* ximg = p.x + (size_t)n_img * p.H * p.W * p.Cin;
    const int nchunks = p.Cin / CK;
    const bf16_t* wbase = p.wpk + (size_t)ntile * nchunks * C::W_ITEMS * 8;

    // ---- per-thread staging descriptors (independent of the chunk) ----
    // timing experiment (LUMINA_CONV_DBG & 8, CK == 16 only): address the input as if it were stored channel-blocked
    // [C/16][H][W][16] — same bytes per image, wrong values — to price the NHWC line over-fetch of 16-channel chunks
    const bool blk_dbg = CK == 16 && (p.dbg_skip & 8);
    const int pstride = blk_dbg ? 16 : p.Cin;
    const int cadv = blk_dbg ? p.H * p.W * 16 : CK;
    int a_goff[C::AIT], a_loff[C::AIT];
    int a_gate[KS == 1 ? C::AIT : 1];   // fused SE gate (1x1 layers): offset of this item's 8 gate values, -1 = none
    // output-tile origin.  OUT_POOL: a tile yields (TH-2)/2 x (TW-2)/2 pooled pixels and needs the conv rows / columns
    // 2*py - 1 .. 2*py + 1 of each: tiles step by TH-2 / TW-2 conv pixels and start one row / column early (overlap = the
    // price of never writing the un-pooled tensor: 14/16 x 30/32 = 82 % of the MFMA work is net)
    constexpr bool pool = C::POOL;
    const int oyb = pool ? tile_y * (C::TH - 2) - 1 : tile_y * C::TH, oxb = pool ? tile_x * (TW - 2) - 1 : tile_x * TW;
    const int iy0 = oyb * S - C::PAD, ix0 = oxb * S - C::PAD;
#pragma unroll
    for (int it = 0; it < C::AIT; ++it) {
        const int i = tid + NTHR * it;
        const int pi = i / C::NPL, c = i - pi * C::NPL;
        const int hy = pi / C::HW, hx = pi - hy * C::HW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool item = i < C::A_ITEMS;
        const bool inb = item && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && (p.pix_limit == 0 || iy * p.W + ix < p.pix_limit);
        a_goff[it] = inb ? (iy * p.W + ix) * pstride + c * 8 : -1;
        a_loff[it] = item ? (c * C::PLANE_A + hy * C::HWP + hx) * 16 : -1;
        if constexpr (KS == 1) a_gate[it] = (inb && p.gate != nullptr) ? ((iy * p.W + ix) / p.gate_hw) * p.Cin + c * 8 : -1;
    }

    uint4 a_reg[C::AIT], w_reg[C::WIT];
    uint4 g_reg[KS == 1 ? C::AIT : 1];
    uint4 w2_reg[C::SC ? C::W2IT : 1];
    unsigned char* sW2 = sW + C::W_BYTES;
    const bf16_t* wbase2 = C::SC ? p.wpk2 + (size_t)ntile * nchunks * C::W2_ITEMS * 8 : nullptr;
#define ISSUE_LOADS(chunk_)                                                                          \
    {                                                                                                \
        const bf16_t* xa = ximg + (chunk_) * cadv;                                                   \
        const bool skipa_ = (p.dbg_skip & 2) && (chunk_) > 0, skipw_ = (p.dbg_skip & 1) && (chunk_) > 0; \
        _Pragma("unroll") for (int it = 0; it < C::AIT; ++it) {                                      \
            uint4 t_ = make_uint4(0, 0, 0, 0);                                                       \
            if (a_goff[it] >= 0 && !skipa_) t_ = *reinterpret_cast<const uint4*>(xa + a_goff[it]);   \
            a_reg[it] = t_;                                                                          \
            if constexpr (KS == 1) {                                                                 \
                uint4 g_ = make_uint4(0, 0, 0, 0);                                                   \
                if (a_gate[it] >= 0) g_ = *reinterpret_cast<const uint4*>(p.gate + a_gate[it] + (chunk_) * CK); \
                g_reg[it] = g_;                                                                      \
            }                                                                                        \
        }                                                                                            \
        const uint4* wsrc = reinterpret_cast<const uint4*>(wbase + (size_t)(chunk_) * C::W_ITEMS * 8); \
        _Pragma("unroll") for (int it = 0; it < C::WIT; ++it) {                                      \
            const int i = tid + NTHR * it;                                                          \
            uint4 t_ = make_uint4(0, 0, 0, 0);                                                       \
            if ((C::W_ITEMS % NTHR == 0 || i < C::W_ITEMS) && !skipw_) t_ = wsrc[i];                 \
            w_reg[it] = t_;                                                                          \
        }                                                                                            \
        if constexpr (C::SC) {                                                                       \
            const uint4* w2src = reinterpret_cast<const uint4*>(wbase2 + (size_t)(chunk_) * C::W2_ITEMS * 8); \
            _Pragma("unroll") for (int it = 0; it < C::W2IT; ++it) {                                 \
                const int i = tid + NTHR * it;                                                      \
                uint4 t_ = make_uint4(0, 0, 0, 0);                                                   \
                if (C::W2_ITEMS % NTHR == 0 || i < C::W2_ITEMS) t_ = w2src[i];                       \
                w2_reg[it] = t_;                                                                     \
            }                                                                                        \
        }                                                                                            \
    }
#define WRITE_LDS(boff_)                                                                             \
    {                                                                                                \
        _Pragma("unroll") for (int it = 0; it < C::AIT; ++it) {                                      \
            if constexpr (KS == 1) { if (a_gate[it] >= 0) a_reg[it] = mul_bf16x8(a_reg[it], g_reg[it]); } \
            if (a_loff[it] >= 0) *reinterpret_cast<uint4*>(sA + (boff_) + a_loff[it]) = a_reg[it];   \
        }                                                                                            \
        _Pragma("unroll") for (int it = 0; it < C::WIT; ++it) {                                      \
            const int i = tid + NTHR * it;                                                          \
            if (C::W_ITEMS % NTHR == 0 || i < C::W_ITEMS) *reinterpret_cast<uint4*>(sW + (boff_) + i * 16) = w_reg[it]; \
        }                                                                                            \
        if constexpr (C::SC) {                                                                       \
            _Pragma("unroll") for (int it = 0; it < C::W2IT; ++it) {                                 \
                const int i = tid + NTHR * it;                                                      \
                if (C::W2_ITEMS % NTHR == 0 || i < C::W2_ITEMS) *reinterpret_cast<uint4*>(sW2 + (boff_) + i * 16) = w2_reg[it]; \
            }                                                                                        \
        }                                                                                            \
    }

    // LDS-DMA staging: every wave-instruction moves 64 x 16 B straight into the lane-linear LDS image (no VGPR round trip,
    // no ds_write pass); halo pixels outside the image read a 16-byte block of zeros instead.
#define DMA_ISSUE(chunk_, boff_)                                                                     \
    {                                                                                                \
        const bf16_t* xa = ximg + (chunk_) * cadv;                                                   \
        const bool skipa_ = (p.dbg_skip & 2) && (chunk_) > 0, skipw_ = (p.dbg_skip & 1) && (chunk_) > 0; \
        _Pragma("unroll") for (int it = 0; it < C::AIT; ++it) {                                      \
            const int i = tid + NTHR * it;                                                           \
            if ((C::A_ITEMS % NTHR == 0 || i < C::A_ITEMS) && !skipa_) {                             \
                const bf16_t* src_ = a_goff[it] >= 0 ? xa + a_goff[it] : p.zeros;                    \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_), \
                    (__attribute__((address_space(3))) void*)(sA + (boff_) + (i - lane) * 16), 16, 0, 0); \
            }                                                                                        \
        }                                                                                            \
        const bf16_t* wsrc = wbase + (size_t)(chunk_) * C::W_ITEMS * 8;                              \
        _Pragma("unroll") for (int it = 0; it < C::WIT; ++it) {                                      \
            const int i = tid + NTHR * it;                                                           \
            if ((C::W_ITEMS % NTHR == 0 || i < C::W_ITEMS) && !skipw_)                               \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + i * 8), \
                    (__attribute__((address_space(3))) void*)(sW + (boff_) + (i - lane) * 16), 16, 0, 0); \
        }                                                                                            \
    }

    f32x16_t acc[MT][C::NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mt][nt][j] = 0.f;
    f32x16_t acc2[C::SC ? MT : 1][C::SC ? C::NT : 1];   // fused shortcut: its own accumulators
    if constexpr (C::SC) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc2[mt][nt][j] = 0.f;
    }

    // fragment read bases (bytes)
    int aoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int ty = wave * MT + mt, tx = r;  // TW == 32
        aoff[mt] = h * C::A_SL + ((ty * S) * C::HWP + tx * S) * C::A_PIX;
    }
    const int woff = h * C::W_SL + r * C::W_ROW;
    const int woff2 = h * (C::W2_ROWS * 16) + r * 16;   // the shortcut's slab: plane layout, plane stride = its own row count

    // Fragment reads are software-pipelined one (tap, k-step) ahead of the MFMAs that consume them: the reads of step s+1 are
    // in flight while the 2*NT MFMAs of step s issue (hipcc otherwise places every ds_read right in front of its MFMA and the
    // wave idles for the LDS latency once per step).
    constexpr int NSTEPS = C::TAPS * (CK / 16);
    bf16x8_t bfr[2][MT], afr[2][C::NT];
#define LOAD_FRAGS(boff_, st_, set_)                                                                               \
    {                                                                                                              \
        const int tap_ = (st_) / (CK / 16), kc_ = (st_) % (CK / 16), kh_ = tap_ / KS, kw_ = tap_ % KS;              \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                           \
            bfr[set_][mt] = lds_frag(sA + (boff_) + aoff[mt] + (2 * kc_) * C::A_SL + (kh_ * C::HWP + kw_) * C::A_PIX); \
        _Pragma("unroll") for (int nt = 0; nt < C::NT; ++nt)                                                       \
            afr[set_][nt] = lds_frag(sW + (boff_) + woff + (2 * kc_) * C::W_SL + (tap_ * BN + nt * 32) * C::W_ROW);   \
    }
#define COMPUTE_CHUNK(boff_)                                                                                       \
    LOAD_FRAGS(boff_, 0, 0)                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    _Pragma("unroll") for (int st = 0; st < NSTEPS; ++st) {                                                        \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                           \
            _Pragma("unroll") for (int nt = 0; nt < C::NT; ++nt)                                                   \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[st & 1][nt], bfr[st & 1][mt], acc[mt][nt], 0, 0, 0); \
        if constexpr (C::SC) {   /* taps (1,1) (1,2) (2,1) (2,2) of the 3x3 window are the shortcut's taps 0..3: same pixel fragments */ \
            if (st == 4 || st == 5 || st == 7 || st == 8) {                                                        \
                const int t2_ = (st == 4) ? 0 : (st == 5) ? 1 : (st == 7) ? 2 : 3;                                 \
                bf16x8_t a2_[C::NT];                                                                               \
                _Pragma("unroll") for (int nt = 0; nt < C::NT; ++nt) a2_[nt] = lds_frag(sW2 + (boff_) + woff2 + (t2_ * BN + nt * 32) * 16); \
                _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                   \
                    _Pragma("unroll") for (int nt = 0; nt < C::NT; ++nt)                                           \
                        acc2[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2_[nt], bfr[st & 1][mt], acc2[mt][nt], 0, 0, 0); \
            }                                                                                                      \
        }                                                                                                          \
        if (st + 1 < NSTEPS) LOAD_FRAGS(boff_, st + 1, (st + 1) & 1)                                               \
        _Pragma("unroll") for (int q_ = 0; q_ < MT * C::NT; ++q_) {                                                \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* one MFMA of this step ... */                     \
            if (q_ < MT + C::NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); /* ... one read of the next */ \
        }                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    }
    if constexpr (C::DMA) {
        // 2-deep LDS ring, ONE barrier per chunk: the DMA of chunk c+1 is issued right after the barrier that retires the reads
        // of chunk c-1 (its target buffer) and flies under all MFMAs of chunk c; vmcnt(0) + barrier publish it.
        DMA_ISSUE(0, 0);
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const int boff = (chunk & 1) * C::BUF_BYTES;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (chunk + 1 < nchunks) DMA_ISSUE(chunk + 1, C::BUF_BYTES - boff);
            COMPUTE_CHUNK(boff)
        }
    } else {
        ISSUE_LOADS(0);
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            __syncthreads();
            WRITE_LDS(0);
            __syncthreads();
            if (chunk + 1 < nchunks) ISSUE_LOADS(chunk + 1);
            COMPUTE_CHUNK(0)
        }
    }
#undef COMPUTE_CHUNK
#undef LOAD_FRAGS
#undef DMA_ISSUE

    // ---------------- epilogue: bias + residual + act -> bf16 -> LDS stage -> coalesced store ----------------
    // All bias / residual loads are issued back to back BEFORE the barrier (one latency, not one per quad), the
    // activation switch is hoisted out of the per-element code.
    const int cout_r8 = (p.Cout + 7) & ~7;
    const float* bptr = p.bias + ntile * BN;
    float4 bias_r[C::NT][4];
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) bias_r[nt][g] = *reinterpret_cast<const float4*>(bptr + nt * 32 + 8 * g + 4 * h);
    uint2 res_r[MT][C::NT][4];
    const bool has_res = p.res != nullptr;
    if (has_res) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int ty = wave * MT + mt, tx = r;
            const int oy = oyb + ty, ox = oxb + tx;
            const bool pvalid = oy >= 0 && ox >= 0 && oy < p.Ho && ox < p.Wo && (p.pix_limit == 0 || oy * p.Wo + ox < p.pix_limit);
            const bf16_t* rrow = p.res + (((size_t)n_img * p.res_h + (oy >> p.res_shift)) * p.res_w + (ox >> p.res_shift)) * p.res_cstride + ntile * BN;
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cn = nt * 32 + 8 * g + 4 * h;
                    uint2 rv = make_uint2(0, 0);
                    if (pvalid && ntile * BN + cn < cout_r8) rv = *reinterpret_cast<const uint2*>(rrow + cn);
                    res_r[mt][nt][g] = rv;
                }
        }
    }
    // Register-staged 3x3 layers with plain NHWC output store straight from the accumulators (below): no LDS stage, no barrier,
    // every wave retires on its own (measured +3..10 % on the short-K stem / stride-2 layers; neutral on the LDS-DMA kernel and
    // 3x slower for the replicated OUT_UPSAMPLE writes, which keep the staged path).  The other modes stage the bf16 tile in
    // LDS (it aliases the operand buffers: barrier).
    const bool direct = KS == 3 && !C::DMA && p.out_mode == OUT_NORMAL && !(p.dbg_skip & 32);
    if (!direct) __syncthreads();
    unsigned char* stage = smem;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = bias_r[nt][g];
                acc[mt][nt][4 * g + 0] += b4.x; acc[mt][nt][4 * g + 1] += b4.y;
                acc[mt][nt][4 * g + 2] += b4.z; acc[mt][nt][4 * g + 3] += b4.w;
                if (has_res) {
                    const uint2 rv = res_r[mt][nt][g];
                    acc[mt][nt][4 * g + 0] += __uint_as_float(rv.x << 16); acc[mt][nt][4 * g + 1] += __uint_as_float(rv.x & 0xFFFF0000u);
                    acc[mt][nt][4 * g + 2] += __uint_as_float(rv.y << 16); acc[mt][nt][4 * g + 3] += __uint_as_float(rv.y & 0xFFFF0000u);
                }
            }
#define FOR_ALL_ACC(expr)                                                          \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                               \
        _Pragma("unroll") for (int nt = 0; nt < C::NT; ++nt)                       \
            _Pragma("unroll") for (int j = 0; j < 16; ++j) { const float v = acc[mt][nt][j]; acc[mt][nt][j] = (expr); }
    if (p.act == ACT_RELU) { FOR_ALL_ACC(fmaxf(v, 0.f)) }
    else if (p.act == ACT_HSWISH) { FOR_ALL_ACC(v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f)) }
    else if (p.act == ACT_SIGMOID) { FOR_ALL_ACC(fast_sigmoidf(v)) }
    else if (p.act == ACT_HSIGMOID) { FOR_ALL_ACC(fminf(fmaxf(0.2f * v + 0.5f, 0.f), 1.f)) }
    else if (p.act == ACT_GELU) { FOR_ALL_ACC(gelu_erf(v)) }
#undef FOR_ALL_ACC
    if constexpr (KS == 3) {
        if (direct) {
            // A lane holds 4 consecutive channels per accumulator quad (8 B as bf16); v_permlane32_swap exchanges quad g of the
            // upper half-wave with quad g+1 of the lower one, after which every lane owns 8 consecutive channels (16 B) of its
            // pixel: half-wave h stores channels 16*gp + 8*h .. +7, a pixel's 32 bytes per instruction are contiguous.
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int oy = oyb + wave * MT + mt, ox = oxb + r;
                const bool pvalid = oy < p.Ho && ox < p.Wo;
#pragma unroll
                for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        const int g0 = 2 * gp, g1 = 2 * gp + 1;
                        const uint32_t q0x = pack_bf16x2(acc[mt][nt][4 * g0 + 0], acc[mt][nt][4 * g0 + 1]), q0y = pack_bf16x2(acc[mt][nt][4 * g0 + 2], acc[mt][nt][4 * g0 + 3]);
                        const uint32_t q1x = pack_bf16x2(acc[mt][nt][4 * g1 + 0], acc[mt][nt][4 * g1 + 1]), q1y = pack_bf16x2(acc[mt][nt][4 * g1 + 2], acc[mt][nt][4 * g1 + 3]);
                        const auto sx = __builtin_amdgcn_permlane32_swap(q0x, q1x, false, false);
                        const auto sy = __builtin_amdgcn_permlane32_swap(q0y, q1y, false, false);
                        const uint4 v = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                        const int co = ntile * BN + nt * 32 + 16 * gp + 8 * h;
                        if (!pvalid || co >= cout_r8) continue;
                        *reinterpret_cast<uint4*>(p.y + (((size_t)n_img * p.Ho + oy) * p.Wo + ox) * p.y_cstride + p.y_coff + co) = v;
                    }
            }
            if constexpr (C::SC) {   // the shortcut's tile: + its bias, no activation, same store pattern into y2
                const float* b2 = p.bias2 + ntile * BN;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int oy = oyb + wave * MT + mt, ox = oxb + r;
                    const bool pvalid = oy < p.Ho && ox < p.Wo;
#pragma unroll
                    for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
                        for (int gp = 0; gp < 2; ++gp) {
                            const int g0 = 2 * gp, g1 = 2 * gp + 1;
                            const float4 ba = *reinterpret_cast<const float4*>(b2 + nt * 32 + 8 * g0 + 4 * h), bb = *reinterpret_cast<const float4*>(b2 + nt * 32 + 8 * g1 + 4 * h);
                            const uint32_t q0x = pack_bf16x2(acc2[mt][nt][4 * g0 + 0] + ba.x, acc2[mt][nt][4 * g0 + 1] + ba.y), q0y = pack_bf16x2(acc2[mt][nt][4 * g0 + 2] + ba.z, acc2[mt][nt][4 * g0 + 3] + ba.w);
                            const uint32_t q1x = pack_bf16x2(acc2[mt][nt][4 * g1 + 0] + bb.x, acc2[mt][nt][4 * g1 + 1] + bb.y), q1y = pack_bf16x2(acc2[mt][nt][4 * g1 + 2] + bb.z, acc2[mt][nt][4 * g1 + 3] + bb.w);
                            const auto sx = __builtin_amdgcn_permlane32_swap(q0x, q1x, false, false);
                            const auto sy = __builtin_amdgcn_permlane32_swap(q0y, q1y, false, false);
                            const uint4 v = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                            const int co = ntile * BN + nt * 32 + 16 * gp + 8 * h;
                            if (!pvalid || co >= cout_r8) continue;
                            *reinterpret_cast<uint4*>(p.y2 + (((size_t)n_img * p.Ho + oy) * p.Wo + ox) * p.y2_cstride + co) = v;
                        }
                }
            }
            return;
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int tp = (wave * MT + mt) * TW + r;
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cn = nt * 32 + 8 * g + 4 * h;
                uint2 o;
                o.x = pack_bf16x2(acc[mt][nt][4 * g + 0], acc[mt][nt][4 * g + 1]);
                o.y = pack_bf16x2(acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]);
                *reinterpret_cast<uint2*>(stage + tp * C::STAGE_PITCH + cn * 2) = o;
            }
    }
    __syncthreads();
    if (p.out_mode == OUT_CONVT && p.fuse_w != nullptr) {
        // Fused DBHead tail on the matrix cores: D2[q2][pixel] = W3[q2][c] * act[c][pixel] (K = 64, rows >= 4 of W3 are zero).
        // B fragments come straight from the staged bf16 tile (row pitch 144 B: conflict-free b128 reads), A fragments from the
        // 2 KB pre-packed weight image [kstep][half][32][8]; lanes with h == 0 end up with the 4 outputs of their pixel.
        if constexpr (BN == 64) {
            const int q = ntile;  // convt_c == 64: one sub-pixel per cout tile
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x16_t d2;
#pragma unroll
                for (int j = 0; j < 16; ++j) d2[j] = 0.f;
                const int tp = (wave * MT + mt) * TW + r;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8_t bfr = lds_frag(stage + tp * C::STAGE_PITCH + (ks * 16 + h * 8) * 2);
                    const bf16x8_t afr = *reinterpret_cast<const bf16x8_t*>(p.fuse_w + ((ks * 2 + h) * 32 + r) * 8);
                    d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, d2, 0, 0, 0);
                }
                const int ty = wave * MT + mt, tx = r;
                const int oy = oyb + ty, ox = oxb + tx;
                if (h == 0 && oy < p.Ho && ox < p.Wo) {
                    const int yy = 4 * oy + 2 * (q >> 1), xx = 4 * ox + 2 * (q & 1);
                    bf16_t* dst = p.y + ((size_t)n_img * (4 * p.Ho) + yy) * (size_t)(4 * p.Wo) + xx;
                    *reinterpret_cast<uint32_t*>(dst) = pack_bf16x2(apply_act(d2[0] + p.fuse_b, ACT_SIGMOID), apply_act(d2[1] + p.fuse_b, ACT_SIGMOID));
                    *reinterpret_cast<uint32_t*>(dst + 4 * p.Wo) = pack_bf16x2(apply_act(d2[2] + p.fuse_b, ACT_SIGMOID), apply_act(d2[3] + p.fuse_b, ACT_SIGMOID));
                }
            }
        }
        return;
    }
    if constexpr (C::POOL) {
        {
            // fused 3x3 / s2 / p1 max pool over the staged bf16 tile: pooled pixel (pr, pc) of this tile = max over tile rows
            // 2pr .. 2pr+2, columns 2pc .. 2pc+2 that lie inside the conv output (pool padding is -inf: skipped)
            constexpr int PH = (C::TH - 2) / 2, PW = (TW - 2) / 2;
            const int Hq = (p.Ho - 1) / 2 + 1, Wq = (p.Wo - 1) / 2 + 1;
            for (int i = tid; i < PH * PW * C::CPP; i += NTHR) {
                const int pp = i / C::CPP, ch = i - pp * C::CPP;
                const int pr = pp / PW, pc = pp - pr * PW;
                const int py = tile_y * PH + pr, px = tile_x * PW + pc;
                const int co = ntile * BN + ch * 8;
                if (py >= Hq || px >= Wq || co >= cout_r8) continue;
                float m[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = -3.0e38f;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int lr = 2 * pr + dy, lc = 2 * pc + dx;
                        const int cy = oyb + lr, cx = oxb + lc;
                        if (cy < 0 || cy >= p.Ho || cx < 0 || cx >= p.Wo) continue;
                        const uint4 v = *reinterpret_cast<const uint4*>(stage + (lr * TW + lc) * C::STAGE_PITCH + ch * 16);
                        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            m[2 * j] = fmaxf(m[2 * j], __uint_as_float(w4[j] << 16));
                            m[2 * j + 1] = fmaxf(m[2 * j + 1], __uint_as_float(w4[j] & 0xFFFF0000u));
                        }
                    }
                const uint4 o = make_uint4(pack_bf16x2(m[0], m[1]), pack_bf16x2(m[2], m[3]), pack_bf16x2(m[4], m[5]), pack_bf16x2(m[6], m[7]));
                *reinterpret_cast<uint4*>(p.y + (((size_t)n_img * Hq + py) * Wq + px) * p.y_cstride + p.y_coff + co) = o;
            }
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < (C::TH * TW * C::CPP + NTHR - 1) / NTHR; ++k) {
        const int i = tid + NTHR * k;
        const int tp = i / C::CPP, ch = i - tp * C::CPP;
        if (tp >= C::TH * TW) continue;
        const int ty = tp / TW, tx = tp - ty * TW;
        const int oy = oyb + ty, ox = oxb + tx;
        const int co = ntile * BN + ch * 8;
        if (oy >= p.Ho || ox >= p.Wo || co >= cout_r8 || (p.pix_limit != 0 && oy * p.Wo + ox >= p.pix_limit)) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(stage + tp * C::STAGE_PITCH + ch * 16);
        if (p.out_mode == OUT_NORMAL) {
            bf16_t* dst = p.y + (((size_t)n_img * p.Ho + oy) * p.Wo + ox) * p.y_cstride + p.y_coff + co;
            *reinterpret_cast<uint4*>(dst) = v;
        } else if (p.out_mode == OUT_UPSAMPLE) {
            const int f = 1 << p.up_shift;
            const int HoU = p.Ho << p.up_shift, WoU = p.Wo << p.up_shift;
            for (int dy = 0; dy < f; ++dy)
                for (int dx = 0; dx < f; ++dx) {
                    bf16_t* dst = p.y + (((size_t)n_img * HoU + (oy * f + dy)) * WoU + (ox * f + dx)) * p.y_cstride + p.y_coff + co;
                    *reinterpret_cast<uint4*>(dst) = v;
                }
        } else if (p.out_mode == OUT_CONVT) {
            const int q = co / p.convt_c, cc = co - q * p.convt_c;
            const int yy = 2 * oy + (q >> 1), xx = 2 * ox + (q & 1);
            bf16_t* dst = p.y + (((size_t)n_img * (2 * p.Ho) + yy) * (2 * p.Wo) + xx) * p.y_cstride + p.y_coff + cc;
            *reinterpret_cast<uint4*>(dst) = v;
        } else {  // OUT_CONVT1: columns 0..3 = 2x2 block of a single-channel map
            if (ch == 0) {
                bf16_t* dst = p.y + ((size_t)n_img * (2 * p.Ho) + 2 * oy) * (size_t)(2 * p.Wo) + 2 * ox;
                *reinterpret_cast<uint32_t*>(dst) = v.x;
                *reinterpret_cast<uint32_t*>(dst + 2 * p.Wo) = v.y;
            }
        }
    }
}

template <int KS, int S, int BN, int CK, int TW, int NW = 4, int DB = 0, int MT = 2>
hipError_t launch_t(const ConvParams& p, hipStream_t stream) {
    using C = Cfg<KS, S, BN, CK, TW, NW, DB, MT>;
    auto kern = conv_mfma_kernel<KS, S, BN, CK, TW, NW, DB, MT>;
    { hipError_t e = locr_dyn_lds(reinterpret_cast<const void*>(kern), C::LDS_BYTES); if (e != hipSuccess) return e; }
    const int grid = p.N * p.tiles_x * p.tiles_y * p.n_tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTHR), C::LDS_BYTES, stream, p);
    return hipGetLastError();
}

}  // namespace

size_t conv_packed_weight_elems(int cout_gemm, int ks, int cin, int bn) {
    const int ntiles = (cout_gemm + bn - 1) / bn;
    return (size_t)ntiles * bn * ks * ks * cin;
}

void pack_conv_weights(const bf16_t* ohwi, int cout_gemm, int ks, int cin, int bn, int ck, bf16_t* out, int row_major) {
    const int ntiles = (cout_gemm + bn - 1) / bn, taps = ks * ks, npl = ck / 8, nchunks = cin / ck;
    const size_t wrows = (size_t)taps * bn;
    if (row_major) {  // LDS-DMA variant: [ntile][chunk][tap*BN + n][ck] — a chunk's slab is copied to LDS verbatim
        for (int nt = 0; nt < ntiles; ++nt)
            for (int chunk = 0; chunk < nchunks; ++chunk)
                for (int tap = 0; tap < taps; ++tap)
                    for (int n = 0; n < bn; ++n) {
                        const int co = nt * bn + n;
                        bf16_t* dst = out + (((size_t)nt * nchunks + chunk) * wrows + (size_t)tap * bn + n) * ck;
                        for (int j = 0; j < ck; ++j)
                            dst[j] = co < cout_gemm ? ohwi[((size_t)co * taps + tap) * cin + chunk * ck + j] : (bf16_t)0;
                    }
        return;
    }
    for (int nt = 0; nt < ntiles; ++nt)
        for (int chunk = 0; chunk < nchunks; ++chunk)
            for (int c = 0; c < npl; ++c)
                for (int tap = 0; tap < taps; ++tap)
                    for (int n = 0; n < bn; ++n) {
                        const int co = nt * bn + n;
                        bf16_t* dst = out + ((((size_t)nt * nchunks + chunk) * npl + c) * wrows + (size_t)tap * bn + n) * 8;
                        for (int j = 0; j < 8; ++j)
                            dst[j] = co < cout_gemm ? ohwi[((size_t)co * taps + tap) * cin + chunk * ck + c * 8 + j] : (bf16_t)0;
                    }
}

bool conv_pick_cfg(int ks, int stride, int cin, int cout_gemm, ConvKernelCfg* cfg) {
    if (!((ks == 1 && stride == 1) || (ks == 2 && stride == 2) || (ks == 3 && (stride == 1 || stride == 2)))) return false;
    if (cin % 16 != 0) return false;
    cfg->ks = ks; cfg->stride = stride; cfg->tw = 32; cfg->nw = 4;
    cfg->ck = (cin % 32 == 0) ? 32 : 16;
    cfg->bn = cout_gemm <= 32 ? 32 : (cout_gemm <= 64 ? 64 : 128);
    // measured: 64-channel tiles with >= 2 workgroups per CU beat 128-channel tiles with one (3x3: 460 -> 700-850 TFLOP/s
    // on the 128..512-channel stages; 1x1 layers are bandwidth-bound and want 4 workgroups per CU)
    if (cfg->bn == 128) cfg->bn = 64;
    if (ks == 3 && stride == 2 && cfg->bn == 128 && cfg->ck == 32) cfg->bn = 64;  // halo tile is 4x larger: keep LDS < 160 KB
    // stride-2 layers: the halo tile is 4x larger per output pixel; 16-channel chunks keep LDS <= ~55 KB so that two to three
    // workgroups share a CU (measured 3x3/s2: 0.305 -> 0.234 ms, 0.242 -> 0.186 ms per 16 pages)
    if (stride == 2) cfg->ck = 16;
    if (const char* e = getenv("LUMINA_CONV_CKS2")) { if (stride == 2) cfg->ck = atoi(e) == 32 && cin % 32 == 0 ? 32 : 16; }
    if (ks == 3 && stride == 1 && cin == 32 && cfg->bn == 32) cfg->ck = 16;  // stem.conv2: two pipelined chunks, +4 %
    if (const char* e = getenv("LUMINA_CONV_CK3")) { if (ks == 3 && stride == 1 && cin >= atoi(e)) cfg->ck = 16; }
    if (const char* e = getenv("LUMINA_CONV_NW")) {
        if ((atoi(e) == 5 || atoi(e) == 6) && ks == 3 && stride == 1 && cfg->bn == 64 && cin >= 64) { cfg->nw = atoi(e); cfg->ck = 16; }
    }
    return true;
}

const char* conv_kernel_name(const ConvKernelCfg& c) {
    static thread_local char buf[64];
    const int db = c.nw == 6 ? 3 : 0, mt = (c.nw == 5 || c.nw == 6) ? 4 : 2;
    snprintf(buf, sizeof(buf), "conv_mfma_kernel<%d,%d,%d,%d,32,4,%d,%d>", c.ks, c.stride, c.bn, c.ck, db, mt);  // as rocprofv3 prints it (modulo spaces)
    return buf;
}

#define DISPATCH(KS_, S_, BN_, CK_)                                                             \
    if (cfg.ks == KS_ && cfg.stride == S_ && cfg.bn == BN_ && cfg.ck == CK_) return launch_t<KS_, S_, BN_, CK_, 32>(p, stream);

hipError_t conv_launch(const ConvKernelCfg& cfg, ConvParams p, hipStream_t stream) {
    static const int dbg = getenv("LUMINA_CONV_DBG") ? atoi(getenv("LUMINA_CONV_DBG")) : 0;
    p.dbg_skip = dbg;
    p.tiles_x = ceil_div(p.Wo, 32);
    p.tiles_y = ceil_div(p.Ho, 8);
    p.n_tiles = ceil_div(p.Cout, cfg.bn);
    if (cfg.nw == 5) {  // 4 waves, 4 pixel rows per wave: 16x32-pixel tile, 4x2 MFMA register tile, CK = 16
        p.tiles_y = ceil_div(p.Ho, 16);
        if (cfg.ks == 3 && cfg.stride == 1 && cfg.bn == 64 && cfg.ck == 16) return launch_t<3, 1, 64, 16, 32, 4, 0, 4>(p, stream);
        return hipErrorInvalidValue;
    }
    if (p.wpk2 != nullptr) {   // 3x3 / stride 2 + the block's 2x2 / stride-2 shortcut from the same halo
        if (!(cfg.ks == 3 && cfg.stride == 2 && cfg.bn == 64 && cfg.ck == 16 && cfg.nw == 4) || p.out_mode != OUT_NORMAL || (dbg & 32) || p.bias2 == nullptr ||
            p.y2 == nullptr || p.H % 2 != 0 || p.W % 2 != 0 || p.res != nullptr || p.y2_cstride % 8 != 0) return hipErrorInvalidValue;
        return launch_t<3, 2, 64, 16, 32, 4, 5, 2>(p, stream);
    }
    if (p.out_mode == OUT_POOL && cfg.nw != 6) return hipErrorInvalidValue;
    if (cfg.nw == 6) {  // same tile as nw == 5, operands by LDS-DMA into a 2-deep ring (row-major weight packing)
        p.tiles_y = ceil_div(p.Ho, 16);
        if (p.out_mode == OUT_POOL) {  // tiles step over the POOLED grid: 7 x 15 pooled pixels per tile
            p.tiles_y = ceil_div((p.Ho - 1) / 2 + 1, 7);
            p.tiles_x = ceil_div((p.Wo - 1) / 2 + 1, 15);
        }
        if (p.zeros == nullptr) return hipErrorInvalidValue;
        if (cfg.ks == 3 && cfg.stride == 1 && cfg.bn == 64 && cfg.ck == 16)
            return p.out_mode == OUT_POOL ? launch_t<3, 1, 64, 16, 32, 4, 4, 4>(p, stream) : launch_t<3, 1, 64, 16, 32, 4, 3, 4>(p, stream);
        return hipErrorInvalidValue;
    }
    DISPATCH(3, 1, 32, 32) DISPATCH(3, 1, 64, 32) DISPATCH(3, 1, 128, 32) DISPATCH(3, 1, 64, 16)
    DISPATCH(3, 2, 32, 32) DISPATCH(3, 2, 64, 32) DISPATCH(3, 2, 64, 16)
    DISPATCH(1, 1, 32, 32) DISPATCH(1, 1, 64, 32) DISPATCH(1, 1, 128, 32)
    DISPATCH(1, 1, 32, 16) DISPATCH(1, 1, 64, 16) DISPATCH(1, 1, 128, 16)
    DISPATCH(2, 2, 32, 32) DISPATCH(2, 2, 64, 32) DISPATCH(2, 2, 128, 32) DISPATCH(2, 2, 32, 16) DISPATCH(2, 2, 64, 16)
    DISPATCH(3, 2, 32, 16) DISPATCH(3, 1, 32, 16)
    return hipErrorInvalidValue;
}
