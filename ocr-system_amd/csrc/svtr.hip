// SVTR recogniser (Tiny / Base: dimensions come from the weight blob) on gfx950, in bf16 or fp16 (BASELINE configs[4]: "SVTR-base
// multilingual, fp16 MFMA").  No reference counterpart exists (the model is named by BASELINE; SURVEY.md §0.5: no code, weights
// or dictionary ship): parity unpinned, arithmetic definition = oracle/nets.py svtr_backbone — stored tensors 16-bit (bf16 or
// fp16), fp32 inside (accumulation, soft-max, mean / variance, GELU), one rounding per stored tensor.
//
// Four kernels, each templated on the storage type:
//   svtr_im2col     u8 crop -> normalised 3x3/s2 patches [N*16*160][32] (27 taps + 5 zeros): the first conv becomes a GEMM
//   svtr_gemm       every linear layer AND every convolution of the model: Y = epi(gather(X) W^T + b).  128 token rows x
//                   32*NT output channels per work-group, K in 32-deep steps through a 2-deep LDS ring filled by LDS-DMA
//                   (global_load_lds 16 B, slices XOR-swizzled on the source side: conflict-free ds_read_b128), 32x32x16 MFMA
//                   with the weights as the A operand (a lane ends up with 4 consecutive channels of ONE token per quad).
//                   gather = 3x3 taps with stride (sh, sw) and zero padding over a token grid (taps = 1: plain GEMM).
//                   epilogue in registers: + bias, + residual (a row of the same tensor shape, or a row of a [res_mod][N] table:
//                   the positional embedding), activation, rounding, and optionally LayerNorm over the N channels of the token
//                   (a wave owns complete token rows: the row statistics are a lane-local sum + one cross-half shuffle) — the
//                   un-normalised tensor never goes to memory, there is no separate LayerNorm pass.
//   svtr_attn       flash-attention form on the matrix cores, head dimension 32; K and V^T of a head in LDS; LOCAL mixing blocks
//                   use 4x8-token query tiles whose 7x11 windows cover <= 10 key rows x 21 key columns: one 32-key tile per key
//                   row instead of every key of the grid.
//   svtr_rowmean    mean over the remaining token rows.
// The CTC head (ops.hip ctc_fc_argmax, templated on the same storage type) follows.
#include "svtr.h"

#include <cstdio>

#include <cstdlib>

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

template <int DT> struct Num;
template <> struct Num<0> {   // bf16
    typedef bf16x8_t frag_t;
    static __device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
    static __device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
    static __device__ __forceinline__ float one(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
    static __device__ __forceinline__ uint32_t pack(float a, float b) { return pack_bf16x2(a, b); }
    static __device__ __forceinline__ uint16_t cvt(float a) { return f32_to_bf16(a); }
    static __device__ __forceinline__ f32x16_t mfma(frag_t a, frag_t b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Num<1> {   // fp16
    typedef f16x8_t frag_t;
    static __device__ __forceinline__ float lo(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[0]; }
    static __device__ __forceinline__ float hi(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[1]; }
    static __device__ __forceinline__ float one(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
    static __device__ __forceinline__ uint32_t pack(float a, float b) { f16x2_t v; v[0] = (_Float16)a; v[1] = (_Float16)b; return __builtin_bit_cast(uint32_t, v); }   // round to nearest even
    static __device__ __forceinline__ uint16_t cvt(float a) { return __builtin_bit_cast(uint16_t, (_Float16)a); }
    static __device__ __forceinline__ f32x16_t mfma(frag_t a, frag_t b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

inline int grid_for(size_t total) {
    size_t g = (total + 255) / 256;
    return (int)(g > 256 * 32 ? 256 * 32 : (g ? g : 1));
}

// ------------------------------------------------------------------------------------------------ im2col of the first conv
// crops u8 [N][32][320][3] -> patches [N][16][160][32]: element (kh * 3 + kw) * 3 + c = xn(2 oy + kh - 1, 2 ox + kw - 1, c), xn = T(u8 * (2 / 255) - 1)
// inside the crop's valid width, 0 outside the image / beyond the valid width (zero padding in normalised space); 27..31 = 0.
template <int DT>
__global__ __launch_bounds__(256) void svtr_im2col_kernel(const uint8_t* crops, const int* widths, uint16_t* out, int N) {
    const size_t total = (size_t)N * 16 * 160;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ox = (int)(i % 160), oy = (int)((i / 160) % 16), n = (int)(i / 2560);
        const int vw = widths ? widths[n] : 320;
        const uint8_t* img = crops + (size_t)n * 32 * 320 * 3;
        uint16_t v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = 0;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iy = 2 * oy + kh - 1, ix = 2 * ox + kw - 1;
                if (iy < 0 || iy >= 32 || ix < 0 || ix >= vw) continue;
#pragma unroll
                for (int c = 0; c < 3; ++c) v[(kh * 3 + kw) * 3 + c] = Num<DT>::cvt(__fadd_rn(__fmul_rn((float)img[(iy * 320 + ix) * 3 + c], 2.0f / 255.0f), -1.0f));
            }
        uint4* dst = reinterpret_cast<uint4*>(out + i * 32);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            dst[q] = make_uint4(v[8 * q] | ((uint32_t)v[8 * q + 1] << 16), v[8 * q + 2] | ((uint32_t)v[8 * q + 3] << 16), v[8 * q + 4] | ((uint32_t)v[8 * q + 5] << 16),
                                v[8 * q + 6] | ((uint32_t)v[8 * q + 7] << 16));
    }
}

// ------------------------------------------------------------------------------------------------ GEMM with gather + fused epilogue
constexpr int G_BM = 128;
template <int NT, int BK> struct GCfg {
    static constexpr int BN = 32 * NT, SL = BK / 8;                           // 16-byte k-slices per row and stage
    static constexpr int X_BYTES = G_BM * BK * 2, W_BYTES = BN * BK * 2, BUF = X_BYTES + W_BYTES;
    static constexpr int LDS = 2 * BUF;
    static constexpr int XIT = (G_BM * SL) / 256, WIT = (BN * SL + 255) / 256;   // 16-byte pieces per thread and stage
    // slot of k-slice s in row q: s ^ swz(q); 16 consecutive rows must cover the 16 slots of the 256-byte bank row exactly once
    static __host__ __device__ constexpr int swz(int q) { return BK == 32 ? ((q >> 2) & 3) : ((q >> 1) & 7); }
};

// BK = 64 halves the barriers per MFMA (layers whose input channels come in multiples of 64); BK = 32 serves the rest
template <int DT, int NT, bool LN, int BK>
__global__ __launch_bounds__(256, (GCfg<NT, BK>::LDS <= 80 * 1024 && NT <= 6 ? 2 : 1)) void svtr_gemm_kernel(const SvtrGemmParams p) {
    using C = GCfg<NT, BK>;
    constexpr int G_BK = BK, SL = C::SL;
    using NM = Num<DT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * G_BM, n0 = blockIdx.y * C::BN;
    const int nk = p.K / G_BK, cin_blocks = p.Cin / G_BK;   // K = taps * Cin; a 32-deep K block lies inside one tap

    // ---- staging descriptors.  LDS image of a stage: X rows [128][4 slices of 16 B] then W rows [BN][4 slices]; slot s of row q
    // holds k-slice s ^ ((q >> 2) & 3): 16 consecutive rows cover the 16 slots of the 256-byte bank row exactly once.
    int x_row[C::XIT], x_sl[C::XIT];       // token row of this thread's X piece (relative to m0) and its k-slice
    int x_n[C::XIT], x_oy[C::XIT], x_ox[C::XIT];
#pragma unroll
    for (int it = 0; it < C::XIT; ++it) {
        const int i = tid + 256 * it, q = i / SL, s = (i % SL) ^ C::swz(q);
        x_row[it] = q; x_sl[it] = s;
        int m = m0 + q;
        if (m >= p.M) m = p.M - 1;
        x_n[it] = m / p.Tout;
        const int t = m - x_n[it] * p.Tout;
        x_oy[it] = t / p.Wout; x_ox[it] = t - x_oy[it] * p.Wout;
    }
    const uint16_t* wt = p.w + (size_t)n0 * p.K;
    auto stage = [&](int kb, int buf) {
        const int tap = kb / cin_blocks, c0 = (kb - tap * cin_blocks) * G_BK;
        const int kh = tap / 3, kw = tap - kh * 3;
        unsigned char* dst = smem + buf * C::BUF;
#pragma unroll
        for (int it = 0; it < C::XIT; ++it) {
            const int i = tid + 256 * it;
            const uint16_t* src = p.zeros;
            if (p.taps == 1) src = p.x + (size_t)min(m0 + x_row[it], p.M - 1) * p.Cin + c0 + x_sl[it] * 8;
            else {
                const int iy = x_oy[it] * p.sh + kh - 1, ix = x_ox[it] * p.sw + kw - 1;
                if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) src = p.x + ((size_t)x_n[it] * p.Hin * p.Win + (size_t)iy * p.Win + ix) * p.Cin + c0 + x_sl[it] * 8;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(dst + (i - lane) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < C::WIT; ++it) {
            const int i = tid + 256 * it;
            if (C::BN * SL % 256 == 0 || i < C::BN * SL) {
                const int q = i / SL, s = (i % SL) ^ C::swz(q);
                const uint16_t* src = wt + (size_t)q * p.K + kb * G_BK + s * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(dst + C::X_BYTES + (i - lane) * 16), 16, 0, 0);
            }
        }
    };
    static_assert((C::BN * SL) % 64 == 0, "weight pieces split on wave boundaries");

    f32x16_t acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[nt][j] = 0.f;
    // fragment read offsets: row q, k-slice (2 ks + h) -> slot (2 ks + h) ^ swz(q)
    const int xq = wave * 32 + r;
    const int xoff = xq * (BK * 2), xsw = C::swz(xq);
    const int wsw = C::swz(r);   // weight row q = nt * 32 + r: swz(q) == swz(r)

    stage(0, 0);
    for (int kb = 0; kb < nk; ++kb) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // stage kb has landed for every wave; buffer (kb + 1) & 1 is free again
        if (kb + 1 < nk) stage(kb + 1, (kb + 1) & 1);
        const unsigned char* buf = smem + (kb & 1) * C::BUF;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const typename NM::frag_t xb = *reinterpret_cast<const typename NM::frag_t*>(buf + xoff + (((2 * ks + h) ^ xsw) << 4));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const typename NM::frag_t wa = *reinterpret_cast<const typename NM::frag_t*>(buf + C::X_BYTES + (nt * 32 + r) * (BK * 2) + (((2 * ks + h) ^ wsw) << 4));
                acc[nt] = NM::mfma(wa, xb, acc[nt]);
            }
        }
    }

    // ---- epilogue: lane (token r, half h) holds channels n0 + nt * 32 + 8 g + 4 h + j (g, j in 0..3) of its token
    const int m = m0 + wave * 32 + r;
    const bool valid = m < p.M;
    const int mc = valid ? m : p.M - 1;
    const uint16_t* rrow = p.res ? p.res + (size_t)(p.res_mod ? mc % p.res_mod : mc) * p.N + n0 : nullptr;
    float rsum = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = nt * 32 + 8 * g + 4 * h;
            const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n0 + c);
            float v[4] = {acc[nt][4 * g] + b4.x, acc[nt][4 * g + 1] + b4.y, acc[nt][4 * g + 2] + b4.z, acc[nt][4 * g + 3] + b4.w};
            float rv4[4] = {0.f, 0.f, 0.f, 0.f};
            if (rrow) {
                const uint2 rv = *reinterpret_cast<const uint2*>(rrow + c);
                rv4[0] = NM::lo(rv.x); rv4[1] = NM::hi(rv.x); rv4[2] = NM::lo(rv.y); rv4[3] = NM::hi(rv.y);
                if (!p.res_post) { v[0] += rv4[0]; v[1] += rv4[1]; v[2] += rv4[2]; v[3] += rv4[3]; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float y = v[j];
                if (p.act == ACT_GELU) y = gelu_erf(y);
                else if (p.act == ACT_HSWISH) y = y * fminf(fmaxf(y + 3.f, 0.f), 6.f) * (1.f / 6.f);
                if (rrow && p.res_post) y = NM::one(NM::cvt(y)) + rv4[j];   // the activation's output is a stored (rounded) tensor of the definition
                if constexpr (LN) { y = NM::one(NM::cvt(y)); rsum += y; }   // the (un-stored) linear output is a 16-bit tensor in the definition
                acc[nt][4 * g + j] = y;
            }
        }
    if constexpr (LN) {   // LayerNorm over the token's N = 32 NT channels (this work-group holds all of them: gridDim.y == 1)
        rsum += __shfl_xor(rsum, 32);
        const float mu = rsum / (float)C::BN;
        float q = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 16; ++j) { const float d = acc[nt][j] - mu; q += d * d; }
        q += __shfl_xor(q, 32);
        const float rstd = 1.0f / sqrtf(q / (float)C::BN + p.eps);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = nt * 32 + 8 * g + 4 * h;
                const float4 ga = *reinterpret_cast<const float4*>(p.gamma + c), be = *reinterpret_cast<const float4*>(p.beta + c);
                acc[nt][4 * g] = (acc[nt][4 * g] - mu) * rstd * ga.x + be.x; acc[nt][4 * g + 1] = (acc[nt][4 * g + 1] - mu) * rstd * ga.y + be.y;
                acc[nt][4 * g + 2] = (acc[nt][4 * g + 2] - mu) * rstd * ga.z + be.z; acc[nt][4 * g + 3] = (acc[nt][4 * g + 3] - mu) * rstd * ga.w + be.w;
            }
    }
    // v_permlane32_swap trades quad g of the upper half-wave for quad g + 1 of the lower one: 16-byte pieces of 8 consecutive channels
    uint16_t* yrow = p.y + (size_t)mc * p.N + n0 + 8 * h;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            const int g0 = 2 * gp, g1 = 2 * gp + 1;
            const uint32_t q0x = NM::pack(acc[nt][4 * g0], acc[nt][4 * g0 + 1]), q0y = NM::pack(acc[nt][4 * g0 + 2], acc[nt][4 * g0 + 3]);
            const uint32_t q1x = NM::pack(acc[nt][4 * g1], acc[nt][4 * g1 + 1]), q1y = NM::pack(acc[nt][4 * g1 + 2], acc[nt][4 * g1 + 3]);
            const auto sx = __builtin_amdgcn_permlane32_swap(q0x, q1x, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(q0y, q1y, false, false);
            if (valid) *reinterpret_cast<uint4*>(yrow + nt * 32 + gp * 16) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
}

// ------------------------------------------------------------------------------------------------ attention
// One work-group = MA_W waves x 32 queries of one (crop, head); K ([Tpad + 32][32], rows padded to 80 B) and V TRANSPOSED ([32][T + 40])
// of the head are staged in LDS once.  Per 32-key tile a wave computes S^T[key][query] = K Q^T (two 32x32x16 MFMAs, A = K rows from
// LDS, B = the wave's Q rows in registers), masks, updates the running maximum / sum of its queries (a query's 32 scores live in 16
// registers of lane q and 16 of lane q + 32), converts P to the storage type (v_permlane32_swap turns the accumulator layout into
// the 8-consecutive-keys operand layout) and accumulates O^T[d][query] += V^T P^T with two more MFMAs.
// Global blocks: a wave's queries are 32 consecutive tokens, every key tile is visited.  Local blocks (7 x 11 window): a wave's
// queries are a 4-row x 8-column patch of the token grid; their windows lie in key rows y0-3 .. y0+6 and key columns x0-5 .. x0+12,
// i.e. inside ONE 32-key tile per key row starting at the 8-aligned column x0-8: <= 10 tiles instead of T / 32.
constexpr int MA_W = 8, MA_KP = 40, AT_HD = 32;
template <int DT>
__global__ __launch_bounds__(64 * MA_W) void svtr_attn_kernel(const uint16_t* qkv, uint16_t* out, int T, int heads, int gh, int gw, int local) {
    using NM = Num<DT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Tpad = ((T + 31) & ~31) + 32;                 // a local key tile may start up to 24 keys before a row end: readable padding
    const int vp = Tpad + 8;                                // V^T row pitch in elements
    uint16_t* sK = reinterpret_cast<uint16_t*>(smem);      // [Tpad][MA_KP]
    uint16_t* sVt = sK + (size_t)Tpad * MA_KP;             // [32][vp]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int n = blockIdx.z, hd = blockIdx.y, tile = blockIdx.x * MA_W + wave;
    const int C = heads * AT_HD;
    const uint16_t* base = qkv + (size_t)n * T * 3 * C;
    for (int i = tid; i < Tpad * 4; i += 64 * MA_W) {
        const int t = i >> 2, s4 = i & 3;
        uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (t < T) {
            kv = *reinterpret_cast<const uint4*>(base + ((size_t)t * 3 + 1) * C + hd * AT_HD + s4 * 8);
            vv = *reinterpret_cast<const uint4*>(base + ((size_t)t * 3 + 2) * C + hd * AT_HD + s4 * 8);
        }
        *reinterpret_cast<uint4*>(sK + (size_t)t * MA_KP + s4 * 8) = kv;
        const uint32_t w4[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sVt[(size_t)(s4 * 8 + 2 * j) * vp + t] = (uint16_t)(w4[j] & 0xffffu);
            sVt[(size_t)(s4 * 8 + 2 * j + 1) * vp + t] = (uint16_t)(w4[j] >> 16);
        }
    }
    // this lane's query token
    const int tiles_x = gw / 8;
    const int y0 = local ? (tile / tiles_x) * 4 : 0, x0 = local ? (tile % tiles_x) * 8 : 0;
    const int qy = local ? y0 + (r >> 3) : 0, qx = local ? x0 + (r & 7) : 0;
    const int qraw = local ? qy * gw + qx : tile * 32 + r;
    const bool qvalid = local ? (qy < gh) : (qraw < T);
    const int qt = qvalid ? qraw : T - 1;
    typename NM::frag_t qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const typename NM::frag_t*>(base + (size_t)qt * 3 * C + hd * AT_HD + ks * 16 + h * 8);
    __syncthreads();
    if (local ? (y0 >= gh) : (tile * 32 >= T)) return;
    // scores are kept in the log2 domain (32^-0.5 * log2 e folded into one factor): exp(s - m) = exp2(s' - m') is ONE v_exp_f32 instead of
    // libm's expf (range reduction + polynomial, ~20 instructions) — 17 of them per key tile and lane made the kernel transcendental-bound
    const float scale = 0.17677669529663687f * 1.4426950408889634f;
    float m = -3.0e38f, l = 0.f;
    f32x16_t o;
#pragma unroll
    for (int j = 0; j < 16; ++j) o[j] = 0.f;
    // key tiles: (first key, number of keys that count) — global: every 32-key tile; local: one tile per key row of the patch's window
    const int kx_lo = local ? max(x0 - 8, 0) : 0, kx_n = local ? min(gw, x0 + 13) - kx_lo : 0;   // columns [kx_lo, kx_lo + kx_n), kx_n <= 21
    const int ky_lo = local ? max(y0 - 3, 0) : 0, ky_hi = local ? min(gh - 1, y0 + 6) : 0;
    const int ntiles = local ? ky_hi - ky_lo + 1 : (T + 31) / 32;
    for (int it = 0; it < ntiles; ++it) {
        const int k0 = local ? (ky_lo + it) * gw + kx_lo : it * 32;
        const int kcount = local ? kx_n : min(32, T - k0);
        f32x16_t sc;
#pragma unroll
        for (int j = 0; j < 16; ++j) sc[j] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const typename NM::frag_t kf = *reinterpret_cast<const typename NM::frag_t*>(sK + (size_t)(k0 + r) * MA_KP + ks * 16 + h * 8);
            sc = NM::mfma(kf, qf[ks], sc);
        }
        // lane (query = r): register j is key k0 + (j & 3) + 8 * (j >> 2) + 4 * h
        float tmax = -3.0e38f;
        const int ky = ky_lo + it;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int off = (j & 3) + 8 * (j >> 2) + 4 * h;
            bool ok = off < kcount;
            if (local) { const int kx = kx_lo + off; ok = ok && ky >= qy - 3 && ky <= qy + 3 && kx >= qx - 5 && kx <= qx + 5; }
            sc[j] = ok ? sc[j] * scale : -3.0e38f;
            tmax = fmaxf(tmax, sc[j]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mn = fmaxf(m, tmax);
        const float corr = __builtin_amdgcn_exp2f(m - mn);     // (both -3e38 before the first valid key: exp2(0) = 1 scales zeros)
        float psum = 0.f;
        uint32_t pk[8];
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            const float p0 = sc[j] > -1.0e38f ? __builtin_amdgcn_exp2f(sc[j] - mn) : 0.f, p1 = sc[j + 1] > -1.0e38f ? __builtin_amdgcn_exp2f(sc[j + 1] - mn) : 0.f;
            psum += p0 + p1;
            pk[j >> 1] = NM::pack(p0, p1);
        }
        psum += __shfl_xor(psum, 32);
        l = l * corr + psum;
        m = mn;
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j] *= corr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[4 * ks], pk[4 * ks + 2], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[4 * ks + 1], pk[4 * ks + 3], false, false);
            union { uint32_t u[4]; typename NM::frag_t v; } pf;
            pf.u[0] = s0[0]; pf.u[1] = s1[0]; pf.u[2] = s0[1]; pf.u[3] = s1[1];
            const typename NM::frag_t vf = *reinterpret_cast<const typename NM::frag_t*>(sVt + (size_t)r * vp + k0 + ks * 16 + h * 8);
            o = NM::mfma(vf, pf.v, o);
        }
    }
    if (qvalid) {   // lane (query = r): register j is d = (j & 3) + 8 * (j >> 2) + 4 * h
        const float inv = 1.0f / l;
        uint16_t* dst = out + ((size_t)n * T + qraw) * C + hd * AT_HD;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w;
            w.x = NM::pack(o[4 * g] * inv, o[4 * g + 1] * inv);
            w.y = NM::pack(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
            *reinterpret_cast<uint2*>(dst + 8 * g + 4 * h) = w;
        }
    }
}

// y[n, x, c] = T(mean over the H rows of x[n, :, x, c])
template <int DT>
__global__ void svtr_rowmean_kernel(const uint16_t* x, uint16_t* y, int N, int H, int W, int C) {
    const size_t total = (size_t)N * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t t = i / C;
        const int xw = (int)(t % W), n = (int)(t / W);
        float s = 0.f;
        for (int rr = 0; rr < H; ++rr) s += Num<DT>::one(x[(((size_t)n * H + rr) * W + xw) * C + c]);
        y[i] = Num<DT>::cvt(s / (float)H);
    }
}

template <int DT, int NT, bool LN, int BK>
hipError_t gemm_launch_bk(const SvtrGemmParams& p, hipStream_t st) {
    auto kern = svtr_gemm_kernel<DT, NT, LN, BK>;
    constexpr int lds = GCfg<NT, BK>::LDS;
    { hipError_t e = locr_dyn_lds(reinterpret_cast<const void*>(kern), lds); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(kern, dim3((p.M + G_BM - 1) / G_BM, p.N / (32 * NT)), dim3(256), lds, st, p);
    return hipGetLastError();
}
template <int DT, int NT, bool LN>
hipError_t gemm_launch_t(const SvtrGemmParams& p, hipStream_t st) {
    // (measured: a 64-deep K step or four work-groups per CU change nothing for the 128-channel tiles: these short-K products are
    // bound by the bytes of their operands and results, ~100 FLOP per byte, not by barriers)
    if constexpr (NT >= 6) { if (p.Cin % 64 == 0) return gemm_launch_bk<DT, NT, LN, 64>(p, st); }
    return gemm_launch_bk<DT, NT, LN, 32>(p, st);
}

}  // namespace

hipError_t svtr_im2col_launch(const uint8_t* crops, const int* widths, uint16_t* out, int N, int dtype, hipStream_t st) {
    const int g = grid_for((size_t)N * 2560);
    if (dtype) hipLaunchKernelGGL(svtr_im2col_kernel<1>, dim3(g), dim3(256), 0, st, crops, widths, out, N);
    else hipLaunchKernelGGL(svtr_im2col_kernel<0>, dim3(g), dim3(256), 0, st, crops, widths, out, N);
    return hipGetLastError();
}

hipError_t svtr_gemm_launch(const SvtrGemmParams& p, int dtype, hipStream_t st) {
    if (p.M <= 0 || p.K <= 0 || p.K % 32 != 0 || p.N % 32 != 0 || (p.taps != 1 && p.taps != 9) || p.K != p.taps * p.Cin || p.Cin % 32 != 0 ||
        p.zeros == nullptr || p.Tout <= 0 || p.Wout <= 0)
        return hipErrorInvalidValue;
    if ((long long)p.M * p.N >= (1ll << 40)) return hipErrorInvalidValue;
    const bool ln = p.gamma != nullptr;
    // tile width: a LayerNorm epilogue needs the whole token (N = 64 .. 384 channels) in one work-group; other layers use 128-channel tiles
    int nt = ln ? p.N / 32 : ((p.N % 128 == 0) ? 4 : (p.N % 192 == 0 ? 6 : (p.N % 64 == 0 ? 2 : (p.N == 32 ? 1 : 0))));
#define GL(DT_, NT_, LN_) return gemm_launch_t<DT_, NT_, LN_>(p, st)
#define GD(NT_, LN_) { if (dtype) GL(1, NT_, LN_); else GL(0, NT_, LN_); }
    if (ln) {
        switch (nt) { case 2: GD(2, true) case 4: GD(4, true) case 8: GD(8, true) case 12: GD(12, true) default: return hipErrorInvalidValue; }
    }
    switch (nt) { case 1: GD(1, false) case 2: GD(2, false) case 4: GD(4, false) case 6: GD(6, false) default: return hipErrorInvalidValue; }
#undef GD
#undef GL
}

// the instantiation svtr_gemm_launch resolves to, spelled like the profiler spells it (minus blanks)
const char* svtr_gemm_kernel_name(const SvtrGemmParams& p, int dtype) {
    static thread_local char buf[64];
    const bool ln = p.gamma != nullptr;
    const int nt = ln ? p.N / 32 : ((p.N % 128 == 0) ? 4 : (p.N % 192 == 0 ? 6 : (p.N % 64 == 0 ? 2 : (p.N == 32 ? 1 : 0))));
    const int bk = (nt >= 6 && p.Cin % 64 == 0) ? 64 : 32;
    snprintf(buf, sizeof(buf), "svtr_gemm_kernel<%d,%d,%s,%d>", dtype ? 1 : 0, nt, ln ? "true" : "false", bk);
    return buf;
}

hipError_t svtr_rowmean_launch(const uint16_t* x, uint16_t* y, int N, int H, int W, int C, int dtype, hipStream_t st) {
    if (dtype) hipLaunchKernelGGL(svtr_rowmean_kernel<1>, dim3(grid_for((size_t)N * W * C)), dim3(256), 0, st, x, y, N, H, W, C);
    else hipLaunchKernelGGL(svtr_rowmean_kernel<0>, dim3(grid_for((size_t)N * W * C)), dim3(256), 0, st, x, y, N, H, W, C);
    return hipGetLastError();
}

hipError_t svtr_attention_launch(const uint16_t* qkv, uint16_t* out, int N, int T, int heads, int gh, int gw, int local, int dtype, hipStream_t st) {
    if (gh * gw != T || N <= 0 || gw < 32 || gw % 8 != 0 || (local && gh % 4 != 0)) return hipErrorInvalidValue;
    const int Tpad = ((T + 31) & ~31) + 32;
    const size_t lds = (size_t)Tpad * MA_KP * 2 + (size_t)32 * (Tpad + 8) * 2;
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    const int tiles = local ? (gh / 4) * (gw / 8) : (T + 31) / 32;
    const dim3 grid((tiles + MA_W - 1) / MA_W, heads, N);
    if (dtype) {
        { hipError_t e = locr_dyn_lds(reinterpret_cast<const void*>(svtr_attn_kernel<1>), 150 * 1024); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(svtr_attn_kernel<1>, grid, dim3(64 * MA_W), lds, st, qkv, out, T, heads, gh, gw, local);
    } else {
        { hipError_t e = locr_dyn_lds(reinterpret_cast<const void*>(svtr_attn_kernel<0>), 150 * 1024); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(svtr_attn_kernel<0>, grid, dim3(64 * MA_W), lds, st, qkv, out, T, heads, gh, gw, local);
    }
    return hipGetLastError();
}
