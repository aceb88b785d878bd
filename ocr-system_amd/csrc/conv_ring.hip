// Persistent 3x3 / stride-1 implicit-GEMM convolution for gfx950: the LDS-DMA kernel of conv_mfma.hip (16x32-pixel tile x 64
// output channels, 16-channel chunks, 4 waves, 4x2 MFMA register tile per wave) re-shaped so that its per-tile fixed costs
// disappear from the critical path.  Serves the backbone / head layers of the detector (the reference has no local
// convolution code: /root/reference/backend/services/ocr_service.py:213-246 is a remote call; this is the engine slot).
//
// What the one-tile-per-workgroup kernel pays per tile (time = 2.7 + 0.63 * chunks chunk-times) is the DMA
// latency of its first chunk, the bias / address prologue, the staged epilogue with its two barriers, and the drain of its
// stores — more than half the time of a 64-channel layer.  Here:
//   * 2 workgroups per CU stay resident and walk a strided list of tiles (each XCD owns a contiguous range of tiles, the
//     workgroups of an XCD take neighbouring tiles at the same time: halos and weight slabs are shared in that XCD's L2);
//   * (tile, chunk) pairs form ONE continuous 2-deep LDS ring: the first chunk of tile t+1 is requested at the start of the
//     last chunk of tile t and lands under its 72 MFMAs;
//   * the epilogue never touches LDS: bias + residual + ReLU in registers, v_permlane32_swap makes 16-byte pieces, and the
//     stores of tile t are issued AFTER the DMA of (t+1, chunk 1): they drain under the MFMAs of (t+1, chunk 0) and the
//     vmcnt(0) at the next chunk boundary finds them done (vector-memory operations retire in issue order on gfx9, so a
//     store issued before a DMA would hold that DMA's wait: this is what sank the first persistent attempt);
//   * orientation: the 16-row x 32-lane tile can lie either way on the image (TR: lanes run down the image).  NHWC makes
//     both equally coalesced (a pixel's 16-channel chunk is one 32-byte piece either way); the launch picks the orientation
//     that wastes fewer padded pixels (a 63x45 map: 1.44x -> 1.08x).  The summation order (chunk, tap, k) is the same in
//     both orientations and in conv_mfma.hip's kernels: results are bit-identical.
#include "conv_mfma.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int R_TH = 16, R_TW = 32, R_HH = 18, R_HW = 34, R_BN = 64, R_MT = 4, R_NT = 2;
constexpr int R_A_ITEMS = R_HH * R_HW * 2;   // 16-byte pieces of one chunk's halo tile, [pixel][2 slices]
constexpr int R_W_ITEMS = 9 * R_BN * 2;      // ... of one chunk's weight slab, [tap * 64 + n][2 slices]
constexpr int R_AIT = (R_A_ITEMS + 255) / 256;
constexpr int R_A_BYTES = R_AIT * 256 * 16;  // padded to whole rounds of 256 pieces: every halo DMA is a full, branch-free instruction
constexpr int R_W_BYTES = R_W_ITEMS * 16, R_BUF = R_A_BYTES + R_W_BYTES;
constexpr int R_BIAS_BYTES = 4096;           // the layer's bias vector (<= 1024 output channels)
constexpr int R_LDS = 2 * R_BUF + R_BIAS_BYTES;   // 81,920 B: two workgroups per CU fill the 160 KB exactly
static_assert(2 * R_LDS <= 160 * 1024, "two resident workgroups per CU");

struct RingTile { int n_img, ntile, oyb, oxb; };

__device__ __forceinline__ bf16x8_t ring_frag(const unsigned char* p) { return *reinterpret_cast<const bf16x8_t*>(p); }

// PROF (developer tool, LUMINA_RING_PROF=1): per-phase core-clock sums of every wave -> prof[0..7] (see conv_ring_launch)
// NWV = 8: 8 waves = a 32 x 32-pixel tile, ONE work-group per CU — the weight slab of a chunk serves twice the MFMAs (8 LDS-DMA requests
// per wave and chunk instead of 10), the halo overlaps less with its neighbours.  Same fragments, same summation order: bit-identical.
template <int NWV> struct RingCfg {
    static constexpr int NTHR = 64 * NWV, TH = R_MT * NWV, HH = TH + 2, A_ITEMS = HH * R_HW * 2;
    static constexpr int AIT = (A_ITEMS + NTHR - 1) / NTHR, WIT = (R_W_ITEMS + NTHR - 1) / NTHR;
    static constexpr int A_BYTES = AIT * NTHR * 16, BUF = A_BYTES + R_W_BYTES, LDS = 2 * BUF + R_BIAS_BYTES;
};
static_assert(RingCfg<4>::LDS == R_LDS && RingCfg<8>::LDS <= 160 * 1024, "LDS budget");

template <int TR, bool PROF = false, bool POOL = false, int NWV = 4>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1) void conv_ring_kernel(const ConvParams p, const int total_tiles, const int per_xcd, const int wg_per_xcd,
                                                           unsigned long long* prof) {
    // (the tile constants of this instantiation shadow the 4-wave ones declared at namespace scope)
    constexpr int NTHR = RingCfg<NWV>::NTHR, R_TH = RingCfg<NWV>::TH, R_A_ITEMS = RingCfg<NWV>::A_ITEMS, R_AIT = RingCfg<NWV>::AIT;
    constexpr int R_WIT = RingCfg<NWV>::WIT, R_A_BYTES = RingCfg<NWV>::A_BYTES, R_BUF = RingCfg<NWV>::BUF;
    long long t_wait = 0, t_bar = 0, t_issue = 0, t_comp = 0, t_epi = 0, t0 = 0, t1 = 0;
    const long long t_begin = PROF ? clock64() : 0;
    const unsigned long long r_begin = PROF ? __builtin_amdgcn_s_memrealtime() : 0ull;   // 100 MHz: with t_begin gives the core clock the kernel ran at
#define RING_T(acc_) if constexpr (PROF) { t1 = clock64(); acc_ += t1 - t0; t0 = t1; }

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;   // (wave: a scalar, so that everything derived from it stays in SGPRs)

    // logical tile axes: y = the 16-row axis, x = the 32-lane axis; image pixel (ly, lx) sits at pixel offset ly * sy + lx * sx
    const int LH = TR ? p.W : p.H, LW = TR ? p.H : p.W;
    const int sy = TR ? 1 : p.W, sx = TR ? p.W : 1;
    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int nchunks = p.Cin >> 4;
    const int hw = p.H * p.W;
    const int chunk_adv = p.x_blk ? hw * 16 : 16;   // elements from one 16-channel chunk of the input to the next

    const int xcd = blockIdx.x & 7;
    int lid = xcd * per_xcd + (blockIdx.x >> 3);
    const int lid_end = min((xcd + 1) * per_xcd, total_tiles);
    if (lid >= lid_end) return;

    auto decode = [&](int id) {
        RingTile t;
        t.ntile = id % p.n_tiles;
        const int mtile = id / p.n_tiles;
        t.n_img = mtile / tiles_per_img;
        const int trem = mtile - t.n_img * tiles_per_img;
        const int ty = trem / p.tiles_x;
        t.oyb = ty * R_TH; t.oxb = (trem - ty * p.tiles_x) * R_TW;
        // fused 3x3/s2 max pool: a tile yields 7 x 15 pooled pixels from conv rows / columns 2q - 1 .. 2q + 1: tiles step by
        // 14 x 30 conv pixels and start one row / column early (82 % of the MFMA work is net, the un-pooled tensor is never written)
        if constexpr (POOL) { t.oyb = ty * (R_TH - 2) - 1; t.oxb = (trem - ty * p.tiles_x) * (R_TW - 2) - 1; }
        return t;
    };

    // multi-source input: source k supplies xs_nchunk[k] consecutive chunks, stored at 1 / 2^shift of the resolution, pixels xs_cstride[k] apart
    const int nsrc = p.n_src > 1 ? p.n_src : 1;
    // (selected with constant indices: indexing the kernel argument with a run-time `src` makes hipcc copy the struct to scratch,
    // and scratch loads queue with the LDS-DMA on the vector-memory counter)
    auto src_shift = [&](int src) { return src == 0 ? p.xs_shift[0] : (src == 1 ? p.xs_shift[1] : (src == 2 ? p.xs_shift[2] : p.xs_shift[3])); };
    auto src_ptr = [&](int src) { return src == 0 ? p.xs[0] : (src == 1 ? p.xs[1] : (src == 2 ? p.xs[2] : p.xs[3])); };
    auto src_nchunk = [&](int src) { return src == 0 ? p.xs_nchunk[0] : (src == 1 ? p.xs_nchunk[1] : (src == 2 ? p.xs_nchunk[2] : p.xs_nchunk[3])); };
    auto src_cstride = [&](int src) { return src == 0 ? p.xs_cstride[0] : (src == 1 ? p.xs_cstride[1] : (src == 2 ? p.xs_cstride[2] : p.xs_cstride[3])); };
    // halo pieces of this thread: piece i = tid + 256 * it is slice c of halo pixel (hy, hx) — the same for every tile, so the
    // divisions are done once and kept packed, two pieces per register (left to itself hipcc hoists the unpacked (hy, hx) of all
    // ten pieces out of the tile loop, spills them, and reloads them from scratch in the middle of a chunk: a scratch load sits
    // on the vector-memory counter BEHIND the LDS-DMA requests in flight, i.e. it waits for them to land) ...
    unsigned a_item[(R_AIT + 1) / 2];
#pragma unroll
    for (int it = 0; it < R_AIT; ++it) {
        const int i = tid + NTHR * it, pi = i >> 1, hy = pi / R_HW, hx = pi - hy * R_HW;
        const unsigned e = (unsigned)hy | ((unsigned)hx << 6) | ((unsigned)(i & 1) << 12) | ((unsigned)(i < R_A_ITEMS) << 13);
        if (it & 1) a_item[it >> 1] |= e << 16; else a_item[it >> 1] = e;
    }
    // ... and their global offsets for the tile / source being requested
    int a_goff[R_AIT];
    auto describe = [&](const RingTile& t, int src) {
        const int sh = nsrc > 1 ? src_shift(src) : 0, cin_s = nsrc > 1 ? src_cstride(src) : p.Cin, ws = p.W >> sh;
#pragma unroll
        for (int j = 0; j < (R_AIT + 1) / 2; ++j) asm volatile("" : "+v"(a_item[j]));   // opaque: the unpacking below stays inside the tile loop
#pragma unroll
        for (int it = 0; it < R_AIT; ++it) {
            const unsigned e = (a_item[it >> 1] >> (16 * (it & 1))) & 0xffffu;
            const int hy = e & 63, hx = (e >> 6) & 63, c = (e >> 12) & 1;
            const int ly = t.oyb - 1 + hy, lx = t.oxb - 1 + hx;
            const bool inb = (e >> 13) != 0 && ly >= 0 && ly < LH && lx >= 0 && lx < LW;
            const int cs = c ^ ((hx >> 3) & 1);   // LDS slot c of this pixel holds slice cs (bank swizzle, see the fragment reads)
            const int iy = TR ? lx : ly, ix = TR ? ly : lx;   // image row / column of the halo pixel
            const int pix = nsrc > 1 ? (iy >> sh) * ws + (ix >> sh) : ly * sy + lx * sx;   // nearest-upsampled read of a low-resolution source
            a_goff[it] = inb ? pix * (p.x_blk ? 16 : cin_s) + cs * 8 : -1;
            if (PROF && (p.dbg_skip & 8)) a_goff[it] = (tid + NTHR * it) * 8;   // timing experiment: a perfectly coalesced halo (wrong values)
        }
    };
    const bf16_t* ximg;
    const bf16_t* wbase;
    auto rebase = [&](const RingTile& t, int src) {
        if (nsrc > 1) { const int sh = src_shift(src); ximg = src_ptr(src) + (size_t)t.n_img * (p.H >> sh) * (p.W >> sh) * src_cstride(src); }
        else ximg = p.x + (size_t)t.n_img * p.H * p.W * p.Cin;
        wbase = p.wpk + (size_t)t.ntile * nchunks * (R_W_ITEMS * 8);
    };

    // One LDS-DMA wave-instruction (64 x 16 B): piece `it` of the halo / of the weight slab of (tile, chunk) -> ring buffer.
    // Halo pieces outside the image (and the padding pieces of the last round) read a block of zeros; the source is selected
    // arithmetically so that the instruction sits in straight-line code and can be scheduled between the MFMAs.
    auto dma_a = [&](int it, const bf16_t* xa, int tgt) {
        const unsigned long long in_addr = (unsigned long long)xa + 2ull * (unsigned)a_goff[it];
        const unsigned long long addr = a_goff[it] >= 0 ? in_addr : (unsigned long long)p.zeros;
        if (!(PROF && (p.dbg_skip & 2)))
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)addr,
                (__attribute__((address_space(3))) void*)(smem + tgt + (wave * 64 + NTHR * it) * 16), 16, 0, 0);
    };
    auto dma_w = [&](int it, const bf16_t* wsrc, int tgt) {
        const int i = tid + NTHR * it;
        if ((R_W_ITEMS % NTHR == 0 || (it + 1) * NTHR <= R_W_ITEMS || wave < (R_W_ITEMS % NTHR) / 64) && !(PROF && (p.dbg_skip & 1)))
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (i ^ ((i >> 4) & 1)) * 8),   // the two slices of a row swapped for rows 8..15 (mod 16)
                (__attribute__((address_space(3))) void*)(smem + R_A_BYTES + tgt + (wave * 64 + NTHR * it) * 16), 16, 0, 0);
    };
    static_assert(R_W_ITEMS % 64 == 0, "weight pieces split on wave boundaries");
    // the k-th of the R_AIT + R_WIT requests of a chunk: halo first (it may come from HBM), weights (L2) after
    auto dma_k = [&](int k, const bf16_t* xa, const bf16_t* wsrc, int tgt) {
        if (k < R_AIT) dma_a(k, xa, tgt); else dma_w(k - R_AIT, wsrc, tgt);
    };
    constexpr int R_NDMA = R_AIT + R_WIT;

    f32x16_t acc[R_MT][R_NT];
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // Fragment read bases.  A 32x32x16 operand wants lane (r, h) = 16 bytes: slice h of pixel / weight row r.  In the plain
    // [pixel][2 slices] image 16 consecutive lanes would read 16 bytes at a 32-byte stride: only half of the 64 banks, a 2-way
    // conflict on every ds_read_b128 (PMC: conflict cycles 48 % of LDS-active cycles, with LDS reads at ~75 % of peak before
    // conflicts).  The two slices of a pixel are therefore stored swapped when bit 3 of its halo column (of its row index,
    // for weights) is set: 16 consecutive columns then cover every 16-byte slot of the 256-byte bank row exactly once.  The
    // swizzle costs nothing at run time: it is folded into the DMA source address and into three per-lane base offsets
    // (one per column offset of a tap).
    int aoff[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) aoff[d] = ((wave * R_MT) * R_HW + r + d) * 32 + 16 * (h ^ (((r + d) >> 3) & 1));   // + (mt + row offset) * R_HW * 32
    const int woff = R_A_BYTES + r * 32 + 16 * (h ^ ((r >> 3) & 1));                                       // + (tap * 64 + nt * 32) * 32

    // fragment reads run one tap ahead of the MFMAs that use them (see conv_mfma.hip)
    bf16x8_t bfr[2][R_MT], afr[2][R_NT];
#define RING_FRAGS(boff_, tap_, set_)                                                                              \
    {                                                                                                              \
        const int kh_ = (tap_) / 3, kw_ = (tap_) % 3;                                                               \
        const int drow_ = TR ? kw_ : kh_, dcol_ = TR ? kh_ : kw_;                                                  \
        _Pragma("unroll") for (int mt = 0; mt < R_MT; ++mt) bfr[set_][mt] = ring_frag(smem + (boff_) + aoff[dcol_] + (mt + drow_) * (R_HW * 32)); \
        _Pragma("unroll") for (int nt = 0; nt < R_NT; ++nt) afr[set_][nt] = ring_frag(smem + (boff_) + woff + ((tap_) * R_BN + nt * 32) * 32); \
    }
    // One chunk: 9 taps x 8 MFMAs.  The R_NDMA requests for the NEXT ring slot (xa_n / ws_n -> tgt_) are spread over the
    // first 7 taps, one or two per tap, each placed behind an MFMA: a wave pays 60-190 cycles of issue time per LDS-DMA
    // instruction, which the matrix pipe covers when they are apart (issued in one burst at the chunk boundary they cost
    // 35-45 % of all wave cycles, measured with the PROF build).
#define RING_COMPUTE(boff_, tgt_, first_)                                                                          \
    RING_FRAGS(boff_, 0, 0)                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    _Pragma("unroll") for (int st = 0; st < 9; ++st) {                                                             \
        /* Pinned order (sched_barrier(0) after every instruction group): behind MFMA q comes memory instruction q — the two   \
           weight fragments and four pixel fragments of the NEXT tap first (their latency hides under the remaining MFMAs, the  \
           compiler's lgkmcnt before the next tap's first MFMA only covers what that MFMA reads), DMA requests last.  Left to \
           itself (sched_group_barrier hints included) hipcc issues five MFMAs, then the reads and requests in a clump, then    \
           waits lgkmcnt(0). */                                                                                    \
        const int nx_ = (st + 1) & 1, tapn_ = st + 1, khn_ = tapn_ / 3, kwn_ = tapn_ % 3;                           \
        const int drown_ = TR ? kwn_ : khn_, dcoln_ = TR ? khn_ : kwn_;                                            \
        int kd_[2] = {-1, -1}, nd_ = 0;                                                                            \
        _Pragma("unroll") for (int k = 0; k < R_NDMA; ++k) if (k * 7 / R_NDMA == st) kd_[nd_++] = k;               \
        _Pragma("unroll") for (int q_ = 0; q_ < R_MT * R_NT; ++q_) {                                               \
            const int mt = q_ / R_NT, nt = q_ % R_NT;                                                              \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[st & 1][nt], bfr[st & 1][mt], ((first_) && st == 0) ? zero16 : acc[mt][nt], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            if (st + 1 < 9) {                                                                                      \
                if (q_ < R_NT) afr[nx_][q_] = ring_frag(smem + (boff_) + woff + (tapn_ * R_BN + q_ * 32) * 32);    \
                else if (q_ < R_NT + R_MT) bfr[nx_][q_ - R_NT] = ring_frag(smem + (boff_) + aoff[dcoln_] + (q_ - R_NT + drown_) * (R_HW * 32)); \
            }                                                                                                      \
            if (q_ >= R_NT + R_MT && kd_[q_ - R_NT - R_MT] >= 0) dma_k(kd_[q_ - R_NT - R_MT], xa_n, ws_n, tgt_);   \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
        }                                                                                                          \
    }

    // finished tile waiting for its stores: 8 channels (16 B) of one pixel per lane and register
    uint4 pk[R_MT][R_NT][2];
    RingTile done{};
    bool pending = false;
    auto store_done = [&]() {
        if constexpr (POOL) {   // pk[0] / pk[1] = pooled rows 2 * wave, 2 * wave + 1 of the tile; even lanes hold pooled column r / 2
            const int Hq = (p.H - 1) / 2 + 1, Wq = (p.W - 1) / 2 + 1;
            const int px = (done.oxb + 1) / 2 + (r >> 1);
            if ((r & 1) || (r >> 1) >= (R_TW - 2) / 2 || px >= Wq) return;
            bf16_t* yimg = p.y + (size_t)done.n_img * Hq * Wq * p.y_cstride + p.y_coff + done.ntile * R_BN + 8 * h;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int prow = 2 * wave + pr, py = (done.oyb + 1) / 2 + prow;
                if (prow >= (R_TH - 2) / 2 || py >= Hq) continue;
                bf16_t* ypix = yimg + (py * Wq + px) * p.y_cstride;
#pragma unroll
                for (int nt = 0; nt < R_NT; ++nt)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) *reinterpret_cast<uint4*>(ypix + nt * 32 + 16 * gp) = pk[pr][nt][gp];
            }
            return;
        }
        // NHWC: pixel stride y_cstride, the 16-byte pieces of a pixel 16 channels apart; blocked: pixel stride 16, pieces hw * 16 apart
        const int ypix_stride = p.y_blk ? 16 : p.y_cstride, yblk_stride = p.y_blk ? hw * 16 : 16;
        bf16_t* yimg = p.y + (size_t)done.n_img * hw * p.y_cstride + (p.y_blk ? (p.y_coff + done.ntile * R_BN) / 16 * (hw * 16) : p.y_coff + done.ntile * R_BN) + 8 * h;
        const int ox = done.oxb + r;
#pragma unroll
        for (int mt = 0; mt < R_MT; ++mt) {
            const int oy = done.oyb + wave * R_MT + mt;
            if (oy >= LH || ox >= LW) continue;
            bf16_t* ypix = yimg + (oy * sy + ox * sx) * ypix_stride;
#pragma unroll
            for (int nt = 0; nt < R_NT; ++nt)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) *reinterpret_cast<uint4*>(ypix + (nt * 2 + gp) * yblk_stride) = pk[mt][nt][gp];
        }
    };

    // Epilogue of `cur`, one pixel row (mt) at a time; the residual of all four rows is requested before the first is used (the
    // fragment registers are dead here, so the 64 registers fit).  Bias comes from LDS (copied once per workgroup): an ordinary global load here
    // would make hipcc drain the in-flight DMA at its first use.
    const float* sBias = reinterpret_cast<const float*>(smem + 2 * R_BUF);
    RingTile cur{};
    auto epilogue = [&](auto has_res_t, int free_slot) {
        constexpr bool HAS_RES = decltype(has_res_t)::value;
        int hb = h;
        asm volatile("" : "+v"(hb));   // opaque copy: the bias address is formed here, not kept in a register across the chunk loop (it was the one spill,
                                       // and its scratch reload waited for every LDS-DMA request in flight)
        const float* bsrc = sBias + cur.ntile * R_BN + 4 * hb;
        const int rpix_stride = p.res_blk ? 16 : p.res_cstride, rblk_stride = p.res_blk ? hw * 16 : 16;   // as for the output
        const bf16_t* rimg = HAS_RES ? p.res + (size_t)cur.n_img * hw * p.res_cstride + (p.res_blk ? cur.ntile * (R_BN / 16) * (hw * 16) : cur.ntile * R_BN) + 8 * h : nullptr;
        const int ox = cur.oxb + r;
        // The residual is read the way the output is written: 16 bytes = 8 consecutive channels per lane (16 requests of 1 KB per
        // tile; as 8-byte pieces in accumulator layout it took 32, every 128-byte line was touched by 8 of them and the 64 KB
        // tile thrashed the 32 KB L1).  v_permlane32_swap — its own inverse — turns a piece into the two accumulator quads.
        uint4 rr[R_MT][R_NT][2];
        auto load_res = [&](int mt) {
            const int oy = cur.oyb + wave * R_MT + mt;
            const bf16_t* rpix = rimg + ((oy < LH && ox < LW) ? oy * sy + ox * sx : 0) * rpix_stride;
#pragma unroll
            for (int nt = 0; nt < R_NT; ++nt)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) rr[mt][nt][gp] = *reinterpret_cast<const uint4*>(rpix + (nt * 2 + gp) * rblk_stride);
        };
        if constexpr (HAS_RES) {   // every row requested before the first is used: one exposed memory latency per tile, not one per row
#pragma unroll
            for (int mt = 0; mt < R_MT; ++mt) load_res(mt);
        }
#pragma unroll
        for (int mt = 0; mt < R_MT; ++mt) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < R_NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b4 = *reinterpret_cast<const float4*>(bsrc + nt * 32 + 8 * g);
                    acc[mt][nt][4 * g + 0] += b4.x; acc[mt][nt][4 * g + 1] += b4.y; acc[mt][nt][4 * g + 2] += b4.z; acc[mt][nt][4 * g + 3] += b4.w;
                }
#pragma unroll
            for (int nt = 0; nt < R_NT; ++nt)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    uint32_t rq[2][2] = {{0u, 0u}, {0u, 0u}};   // [quad g0 / g1][channel pair]
                    if constexpr (HAS_RES) {
                        const uint4 rv = rr[mt][nt][gp];
                        const auto sx2 = __builtin_amdgcn_permlane32_swap(rv.x, rv.z, false, false);
                        const auto sy2 = __builtin_amdgcn_permlane32_swap(rv.y, rv.w, false, false);
                        rq[0][0] = sx2[0]; rq[0][1] = sy2[0]; rq[1][0] = sx2[1]; rq[1][1] = sy2[1];
                    }
#pragma unroll
                    for (int gq = 0; gq < 2; ++gq) {
                        const int g = 2 * gp + gq;
                        float v0 = acc[mt][nt][4 * g + 0], v1 = acc[mt][nt][4 * g + 1], v2 = acc[mt][nt][4 * g + 2], v3 = acc[mt][nt][4 * g + 3];
                        if constexpr (HAS_RES) {
                            v0 += __uint_as_float(rq[gq][0] << 16); v1 += __uint_as_float(rq[gq][0] & 0xFFFF0000u);
                            v2 += __uint_as_float(rq[gq][1] << 16); v3 += __uint_as_float(rq[gq][1] & 0xFFFF0000u);
                        }
                        if (p.act == ACT_RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                        acc[mt][nt][4 * g + 0] = v0; acc[mt][nt][4 * g + 1] = v1; acc[mt][nt][4 * g + 2] = v2; acc[mt][nt][4 * g + 3] = v3;
                    }
                }
            // a lane holds 4 consecutive channels per accumulator quad; v_permlane32_swap trades quad g of the upper half-wave
            // for quad g+1 of the lower one: every lane then owns 8 consecutive channels of its pixel
#pragma unroll
            for (int nt = 0; nt < R_NT; ++nt)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const int g0 = 2 * gp, g1 = 2 * gp + 1;
                    const uint32_t q0x = pack_bf16x2(acc[mt][nt][4 * g0 + 0], acc[mt][nt][4 * g0 + 1]), q0y = pack_bf16x2(acc[mt][nt][4 * g0 + 2], acc[mt][nt][4 * g0 + 3]);
                    const uint32_t q1x = pack_bf16x2(acc[mt][nt][4 * g1 + 0], acc[mt][nt][4 * g1 + 1]), q1y = pack_bf16x2(acc[mt][nt][4 * g1 + 2], acc[mt][nt][4 * g1 + 3]);
                    const auto sx2 = __builtin_amdgcn_permlane32_swap(q0x, q1x, false, false);
                    const auto sy2 = __builtin_amdgcn_permlane32_swap(q0y, q1y, false, false);
                    pk[mt][nt][gp] = make_uint4(sx2[0], sy2[0], sx2[1], sy2[1]);
                    if constexpr (POOL) {   // conv pixels outside the map are pool padding: ReLU output is >= 0, so 0 is neutral
                        const int oy = cur.oyb + wave * R_MT + mt;
                        const uint32_t keep = (oy >= 0 && oy < LH && ox >= 0 && ox < LW) ? 0xffffffffu : 0u;
                        pk[mt][nt][gp].x &= keep; pk[mt][nt][gp].y &= keep; pk[mt][nt][gp].z &= keep; pk[mt][nt][gp].w &= keep;
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (POOL) {
            // 3x3/s2 max pool on the packed bf16 values.  After ReLU they are >= +0, where bf16 bit patterns order like signed
            // 16-bit integers: v_pk_max_i16 is the float max.  Vertical: wave w owns conv rows 4w..4w+3 -> pooled row 2w is local
            // (rows 0,1,2), pooled row 2w+1 needs row 0 of wave w+1, handed over through the ring slot that was just consumed
            // (two barriers: its readers are done / the rows are there).  Horizontal: even lane r takes lanes r, r+1, r+2.
            typedef short s16x2_t __attribute__((ext_vector_type(2)));
            auto mx = [](uint32_t a, uint32_t b) {
                const s16x2_t m = __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, a), __builtin_bit_cast(s16x2_t, b));
                return __builtin_bit_cast(uint32_t, m);
            };
            auto mx4 = [&](const uint4& a, const uint4& b) { return make_uint4(mx(a.x, b.x), mx(a.y, b.y), mx(a.z, b.z), mx(a.w, b.w)); };
            auto hmax = [&](uint32_t v) {
                const uint32_t v1 = (uint32_t)__builtin_amdgcn_ds_bpermute((lane + 1) * 4, (int)v);
                const uint32_t v2 = (uint32_t)__builtin_amdgcn_ds_bpermute((lane + 2) * 4, (int)v);
                return mx(mx(v, v1), v2);
            };
            unsigned char* exch = smem + free_slot;
            __syncthreads();
            if (wave > 0) {
#pragma unroll
                for (int nt = 0; nt < R_NT; ++nt)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) *reinterpret_cast<uint4*>(exch + (wave - 1) * 4096 + ((nt * 2 + gp) * 64 + lane) * 16) = pk[0][nt][gp];
            }
            __syncthreads();
#pragma unroll
            for (int nt = 0; nt < R_NT; ++nt)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    uint4 nb = make_uint4(0, 0, 0, 0);
                    if (wave < NWV - 1) nb = *reinterpret_cast<const uint4*>(exch + wave * 4096 + ((nt * 2 + gp) * 64 + lane) * 16);
                    const uint4 top = mx4(mx4(pk[0][nt][gp], pk[1][nt][gp]), pk[2][nt][gp]);
                    const uint4 bot = mx4(mx4(pk[2][nt][gp], pk[3][nt][gp]), nb);
                    pk[0][nt][gp] = make_uint4(hmax(top.x), hmax(top.y), hmax(top.z), hmax(top.w));
                    pk[1][nt][gp] = make_uint4(hmax(bot.x), hmax(bot.y), hmax(bot.z), hmax(bot.w));
                }
        }
    };

    for (int i = tid; i < p.Cout; i += NTHR) reinterpret_cast<float*>(smem + 2 * R_BUF)[i] = p.bias[i];
    cur = decode(lid);
    RingTile nxt = cur;
    describe(cur, 0); rebase(cur, 0);
    int boff = 0;
    int src_n = 0, src_end = nsrc > 1 ? src_nchunk(0) : 0;   // source of the chunk being requested / first chunk of the next source
    const bf16_t* xa_n = ximg;    // source of the ring slot being requested: (tile, chunk) after the one being computed
    const bf16_t* ws_n = wbase;
#pragma unroll
    for (int k = 0; k < R_NDMA; ++k) dma_k(k, xa_n, ws_n, 0);
    for (;;) {
        const int next_lid = lid + wg_per_xcd;
        // ---- chunk 0 of `cur` (peeled: the previous tile's stores go out here, a whole chunk of MFMAs before the next vmcnt(0))
        if constexpr (PROF) t0 = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RING_T(t_wait)
        __syncthreads();
        RING_T(t_bar)
        xa_n += chunk_adv; ws_n += R_W_ITEMS * 8;     // nchunks >= 2: request chunk 1
        if (pending) store_done();
        RING_T(t_issue)
        RING_COMPUTE(boff, R_BUF - boff, true)    // first tap: C = 0 (the accumulators are not cleared separately)
        RING_T(t_comp)
        boff = R_BUF - boff;
        for (int chunk = 1; chunk < nchunks; ++chunk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            RING_T(t_wait)
            __syncthreads();
            RING_T(t_bar)
            if (chunk + 1 < nchunks) {
                if (nsrc > 1 && chunk + 1 == src_end) {   // the next chunk is the first of another source tensor
                    ++src_n; src_end += src_nchunk(src_n);
                    describe(cur, src_n); rebase(cur, src_n);
                    xa_n = ximg;
                } else xa_n += chunk_adv;
                ws_n += R_W_ITEMS * 8;
            } else if (next_lid < lid_end) {   // the ring runs on into the next tile
                nxt = decode(next_lid);
                describe(nxt, 0); rebase(nxt, 0);
                xa_n = ximg; ws_n = wbase;
                if (nsrc > 1) { src_n = 0; src_end = src_nchunk(0); }
            } else if (nsrc > 1) {             // no next tile: the spare request must still read valid memory -> this tile's first source again
                describe(cur, 0); rebase(cur, 0);
                xa_n = ximg;
            }                                  // (single source: the last chunk is requested once more, into the free slot)
            RING_T(t_issue)
            RING_COMPUTE(boff, R_BUF - boff, false)
            RING_T(t_comp)
            boff = R_BUF - boff;
        }

        // ---- epilogue in registers: + bias (+ residual), activation, bf16, 16-byte pieces ----
        if (!POOL && p.res != nullptr) epilogue(std::true_type{}, R_BUF - boff); else epilogue(std::false_type{}, R_BUF - boff);   // boff was toggled: R_BUF - boff is the slot just consumed
        RING_T(t_epi)
        done = cur; pending = true;
        lid = next_lid;
        if (lid >= lid_end) break;
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the spare request of the last chunk must not outlive the workgroup's LDS
    store_done();
    if constexpr (PROF) {
        if (lane == 0) {
            const long long all = clock64() - t_begin;
            atomicAdd(prof + 0, (unsigned long long)t_wait); atomicAdd(prof + 1, (unsigned long long)t_bar);
            atomicAdd(prof + 2, (unsigned long long)t_issue); atomicAdd(prof + 3, (unsigned long long)t_comp);
            atomicAdd(prof + 4, (unsigned long long)t_epi); atomicAdd(prof + 5, (unsigned long long)all);
            atomicAdd(prof + 6, 1ull);
            atomicAdd(prof + 7, __builtin_amdgcn_s_memrealtime() - r_begin);
        }
    }
#undef RING_T
#undef RING_COMPUTE
#undef RING_FRAGS
}

}  // namespace

bool conv_ring_supported(const ConvKernelCfg& cfg, const ConvParams& p) {
    static const bool off = getenv("LUMINA_CONV_NO_RING") != nullptr;
    if (off) return false;
    if (!(cfg.nw == 6 && cfg.ks == 3 && cfg.stride == 1 && cfg.bn == 64 && cfg.ck == 16)) return false;
    if ((p.out_mode != OUT_NORMAL && p.out_mode != OUT_POOL) || p.pix_limit != 0 || p.gate != nullptr || p.zeros == nullptr) return false;
    if (p.out_mode == OUT_POOL && (p.act != ACT_RELU || p.res != nullptr)) return false;   // the pool compares bf16 bit patterns: values must be >= 0
    if (p.Cin < 32 || p.Cin % 16 != 0 || p.Cout % 64 != 0 || p.Cout * 4 > R_BIAS_BYTES || p.Ho != p.H || p.Wo != p.W) return false;
    if (p.n_src > 1) {   // chunks per source >= 2 (the peeled first chunk requests chunk 1 of the same source), whole low-resolution pixels
        if (p.n_src > 4 || p.x_blk || p.out_mode != OUT_NORMAL) return false;
        int chunks = 0;
        for (int k = 0; k < p.n_src; ++k) {
            if (p.xs[k] == nullptr || p.xs_shift[k] < 0 || p.xs_shift[k] > 3 || p.H % (1 << p.xs_shift[k]) != 0 || p.W % (1 << p.xs_shift[k]) != 0) return false;
            if (p.xs_nchunk[k] < 2 || p.xs_cstride[k] < 16 * p.xs_nchunk[k] || p.xs_cstride[k] % 8 != 0) return false;
            if ((long long)(p.H >> p.xs_shift[k]) * (p.W >> p.xs_shift[k]) * p.xs_cstride[k] >= (1ll << 31)) return false;
            chunks += p.xs_nchunk[k];
        }
        if (chunks * 16 != p.Cin) return false;
    }
    if (p.act != ACT_NONE && p.act != ACT_RELU) return false;
    if (p.res != nullptr && (p.res_shift != 0 || p.res_h != p.H || p.res_w != p.W || p.res_cstride % 4 != 0 || (long long)p.H * p.W * p.res_cstride >= (1ll << 31))) return false;
    if ((long long)p.H * p.W * (p.Cin > p.y_cstride ? p.Cin : p.y_cstride) >= (1ll << 31)) return false;   // 32-bit per-image offsets
    if (p.y_cstride % 8 != 0 || p.y_coff % 8 != 0) return false;
    if ((p.y_blk && (p.y_cstride % 16 != 0 || p.y_coff % 16 != 0 || p.out_mode != OUT_NORMAL)) || (p.res_blk && p.res_cstride % 16 != 0)) return false;
    return true;
}

// orientation: -1 = whichever pads less (ties: lanes along the image rows), 0 / 1 forced (tests)
static bool ring_transposed_th(const ConvParams& p, int orientation, int th) {
    if (p.out_mode == OUT_POOL) return false;   // the pooled variant is built for one orientation
    auto padded = [&](int lh, int lw) { return (long long)ceil_div(lh, th) * th * ceil_div(lw, R_TW) * R_TW; };
    return orientation < 0 ? padded(p.W, p.H) < padded(p.H, p.W) : orientation != 0;
}
// Does this launch take the 8-wave / 32-row tile?  The short-K layers do (Cin <= 64: four chunks per tile — stage 0 gains 11-14 %,
// same-device A/B; with >= 128 input channels the 4-wave kernel's two independent work-groups per CU win by 0-6 %), where the
// taller tile pads at most 5 % more pixels.  LUMINA_RING_NW8=0 / 2: never / whenever the padding allows (developer A/B).
static bool ring_use_nw8(const ConvParams& p, int orientation) {
    static const int mode = getenv("LUMINA_RING_NW8") ? atoi(getenv("LUMINA_RING_NW8")) : 1;
    if (mode == 0 || p.out_mode != OUT_NORMAL || (mode == 1 && p.Cin > 64)) return false;   // (the pooled stem.conv3 measured the same on either tile: it stays on the 4-wave one)
    auto padded = [&](bool tr, int th) { const int lh = tr ? p.W : p.H, lw = tr ? p.H : p.W; return (double)ceil_div(lh, th) * th * ceil_div(lw, R_TW) * R_TW; };
    const double p4 = padded(ring_transposed_th(p, orientation, R_TH), R_TH), p8 = padded(ring_transposed_th(p, orientation, 2 * R_TH), 2 * R_TH);
    return p8 <= 1.05 * p4;
}
bool conv_ring_transposed(const ConvParams& p, int orientation) {
    return ring_transposed_th(p, orientation, ring_use_nw8(p, orientation) ? 2 * R_TH : R_TH);
}
// the instantiation a launch resolves to, spelled like the profiler spells it (minus blanks)
const char* conv_ring_kernel_name(const ConvParams& p, int orientation) {
    if (p.out_mode == OUT_POOL) return "conv_ring_kernel<0,false,true,4>";
    const bool tr = conv_ring_transposed(p, orientation);
    if (ring_use_nw8(p, orientation)) return tr ? "conv_ring_kernel<1,false,false,8>" : "conv_ring_kernel<0,false,false,8>";
    return tr ? "conv_ring_kernel<1,false,false,4>" : "conv_ring_kernel<0,false,false,4>";
}

hipError_t conv_ring_launch(ConvParams p, int orientation, hipStream_t stream) {
    const bool tr = conv_ring_transposed(p, orientation);
    const int lh = tr ? p.W : p.H, lw = tr ? p.H : p.W;
    const bool pool = p.out_mode == OUT_POOL;
    const bool nw8 = ring_use_nw8(p, orientation);
    p.tiles_y = ceil_div(lh, nw8 ? 2 * R_TH : R_TH); p.tiles_x = ceil_div(lw, R_TW); p.n_tiles = p.Cout / R_BN;
    if (pool) { p.tiles_y = ceil_div((p.H - 1) / 2 + 1, (R_TH - 2) / 2); p.tiles_x = ceil_div((p.W - 1) / 2 + 1, (R_TW - 2) / 2); }
    const long long total = (long long)p.N * p.tiles_x * p.tiles_y * p.n_tiles;
    if (total <= 0 || total >= (1ll << 31)) return hipErrorInvalidValue;
    const int per_xcd = (int)((total + 7) / 8);
    // LUMINA_RING_WGS (developer experiment): fewer resident workgroups per XCD.  32 halves the kernel's HBM reads on the
    // 1/4-resolution layers (64 concurrent 16x32 tiles x 612 halo lines x 128 B = 5 MB do not fit the XCD's 4 MB L2 and the
    // four 32-byte chunk passes over a 64-channel NHWC line re-fetch it) but costs more in latency hiding than it saves.
    static const int wg_cap = getenv("LUMINA_RING_WGS") ? atoi(getenv("LUMINA_RING_WGS")) : 64;
    const int wg_per_xcd = per_xcd < wg_cap ? per_xcd : wg_cap;   // 32 CUs per XCD, two resident workgroups each
    static const bool prof = getenv("LUMINA_RING_PROF") != nullptr;
    static const int dbg = getenv("LUMINA_CONV_DBG") ? atoi(getenv("LUMINA_CONV_DBG")) : 0;
    p.dbg_skip = dbg;
    if (nw8) {
        const int wgs = per_xcd < 32 ? per_xcd : 32;   // one resident work-group per CU
        constexpr int lds8 = RingCfg<8>::LDS;
        const void* fn8 = tr ? reinterpret_cast<const void*>(conv_ring_kernel<1, false, false, 8>) : reinterpret_cast<const void*>(conv_ring_kernel<0, false, false, 8>);
        { hipError_t e = locr_dyn_lds(fn8, lds8); if (e != hipSuccess) return e; }
        unsigned long long* none8 = nullptr;
        if (tr) hipLaunchKernelGGL((conv_ring_kernel<1, false, false, 8>), dim3(8 * wgs), dim3(512), lds8, stream, p, (int)total, per_xcd, wgs, none8);
        else hipLaunchKernelGGL((conv_ring_kernel<0, false, false, 8>), dim3(8 * wgs), dim3(512), lds8, stream, p, (int)total, per_xcd, wgs, none8);
        return hipGetLastError();
    }
    if (prof && !pool) {   // developer tool: where do the waves' cycles go (synchronises, prints one line per launch)
        static unsigned long long* dprof = nullptr;
        if (!dprof && hipMalloc(&dprof, 64) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(dprof, 0, 64, stream);
        const void* fnp = tr ? reinterpret_cast<const void*>(conv_ring_kernel<1, true>) : reinterpret_cast<const void*>(conv_ring_kernel<0, true>);
        hipError_t e = hipFuncSetAttribute(fnp, hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS);
        if (e != hipSuccess) return e;
        if (tr) hipLaunchKernelGGL((conv_ring_kernel<1, true>), dim3(8 * wg_per_xcd), dim3(256), R_LDS, stream, p, (int)total, per_xcd, wg_per_xcd, dprof);
        else hipLaunchKernelGGL((conv_ring_kernel<0, true>), dim3(8 * wg_per_xcd), dim3(256), R_LDS, stream, p, (int)total, per_xcd, wg_per_xcd, dprof);
        unsigned long long hp[8] = {0};
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(hp, dprof, 64, hipMemcpyDeviceToHost);
        const double w = hp[6] ? (double)hp[6] : 1.0, all = hp[5] ? (double)hp[5] : 1.0;
        fprintf(stderr, "[ring prof] %dx%d cin %d cout %d tr %d tiles %lld waves %.0f cycles/wave %.0f clock %.2f GHz: dma-wait %.1f%% barrier %.1f%% issue+stores %.1f%% compute %.1f%% epilogue %.1f%%\n",
                p.H, p.W, p.Cin, p.Cout, (int)tr, total, w, all / w, hp[7] ? 0.1 * all / (double)hp[7] : 0.0, 100.0 * hp[0] / all, 100.0 * hp[1] / all, 100.0 * hp[2] / all, 100.0 * hp[3] / all, 100.0 * hp[4] / all);
        return hipGetLastError();
    }
    const void* fn = pool ? reinterpret_cast<const void*>(conv_ring_kernel<0, false, true>)
                          : (tr ? reinterpret_cast<const void*>(conv_ring_kernel<1>) : reinterpret_cast<const void*>(conv_ring_kernel<0>));
    { hipError_t e = locr_dyn_lds(fn, R_LDS); if (e != hipSuccess) return e; }
    unsigned long long* none = nullptr;
    if (pool) hipLaunchKernelGGL((conv_ring_kernel<0, false, true>), dim3(8 * wg_per_xcd), dim3(256), R_LDS, stream, p, (int)total, per_xcd, wg_per_xcd, none);
    else if (tr) hipLaunchKernelGGL((conv_ring_kernel<1>), dim3(8 * wg_per_xcd), dim3(256), R_LDS, stream, p, (int)total, per_xcd, wg_per_xcd, none);
    else hipLaunchKernelGGL((conv_ring_kernel<0>), dim3(8 * wg_per_xcd), dim3(256), R_LDS, stream, p, (int)total, per_xcd, wg_per_xcd, none);
    return hipGetLastError();
}
