// Bandwidth-bound helper kernels (NHWC bf16), the LSTM recurrence and the CTC tail for gfx950.
// No reference counterpart exists; semantics are defined by oracle/nets.py.
#include "ops.h"

#include <type_traits>

#include <cstdlib>

namespace {

__device__ __forceinline__ void unpack8(const uint4 v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xFFFF0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xFFFF0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xFFFF0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xFFFF0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
    v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
    return v;
}

// ------------------------------------------------------------------ max pool
__global__ void maxpool_kernel(const bf16_t* x, bf16_t* y, int N, int H, int W, int C, int k, int s, int pad, int Ho, int Wo) {
    const int cg = C >> 3;
    const size_t total = (size_t)N * Ho * Wo * cg;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cg);
        size_t t = i / cg;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -3.0e38f;
        for (int dy = 0; dy < k; ++dy) {
            const int iy = oy * s - pad + dy;
            if (iy < 0 || iy >= H) continue;
            for (int dx = 0; dx < k; ++dx) {
                const int ix = ox * s - pad + dx;
                if (ix < 0 || ix >= W) continue;
                float f[8];
                unpack8(*reinterpret_cast<const uint4*>(x + (((size_t)n * H + iy) * W + ix) * C + c8 * 8), f);
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], f[j]);
            }
        }
        *reinterpret_cast<uint4*>(y + i * 8) = pack8(m);
    }
}

// ------------------------------------------------------------------ depthwise conv
// One thread = 4 consecutive output pixels x 8 channels: per kernel row the K+3 input vectors are loaded once (branch-free:
// clamped address + select; a guarded load per tap serialises one memory latency per tap) and re-used by the 4 sliding
// windows; fp32 FMA accumulation in tap order (kh, kw).  The layer's weights and bias are converted to fp32 once per
// workgroup and live in LDS ([tap][half][C/8][4] floats: a tap is two conflict-free ds_read_b128 across lanes), the
// activation is a template parameter, index arithmetic is 32-bit.  VALU-bound (PMC: VALU busy ~75 % of the kernel time,
// ~2.3 k vector instructions per wave for 4 x 8 outputs x K*K taps).
template <int K, int ACT>
__global__ __launch_bounds__(256) void dwconv_kernel(const bf16_t* x, const bf16_t* w, const float* bias, bf16_t* y, int N, int H, int W, int C,
                                                     int sh, int Ho) {
    constexpr int PAD = K / 2, XG = 4;
    extern __shared__ float wl[];  // [K*K][2][cg][4] weights, then bias [C]
    const int cg = C >> 3, wg = (W + XG - 1) / XG;
    for (int i = threadIdx.x; i < K * K * C; i += 256) {
        const int tap = i / C, c = i - tap * C;
        wl[((tap * 2 + ((c >> 2) & 1)) * cg + (c >> 3)) * 4 + (c & 3)] = bf16_to_f32(w[i]);
    }
    float* bl = wl + K * K * C;
    for (int i = threadIdx.x; i < C; i += 256) bl[i] = bias[i];
    __syncthreads();
    const unsigned total = (unsigned)N * Ho * wg * cg;  // < 2^31 (checked by the launcher)
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int c8 = (int)(i % (unsigned)cg);
        unsigned t = i / (unsigned)cg;
        const int xg = (int)(t % (unsigned)wg); t /= (unsigned)wg;
        const int oy = (int)(t % (unsigned)Ho);
        const int n = (int)(t / (unsigned)Ho);
        const int ox0 = xg * XG;
        float a[XG][8];
#pragma unroll
        for (int o = 0; o < XG; ++o)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[o][j] = 0.f;
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            const int iy = oy * sh - PAD + kh;
            if (iy < 0 || iy >= H) continue;
            float in[K + XG - 1][8];
            const bf16_t* row = x + ((size_t)(n * H + iy) * W) * C + c8 * 8;
#pragma unroll
            for (int j = 0; j < K + XG - 1; ++j) {
                const int ix = ox0 - PAD + j;
                const int ixc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
                uint4 v = *reinterpret_cast<const uint4*>(row + ixc * C);
                if (ix != ixc) v = make_uint4(0, 0, 0, 0);
                unpack8(v, in[j]);
            }
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
        bf16_t* yrow = y + ((size_t)(n * Ho + oy) * W + ox0) * C + c8 * 8;
#pragma unroll
        for (int o = 0; o < XG; ++o) {
            if (ox0 + o >= W) break;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[o][j] = apply_act(a[o][j] + bb[j], ACT);
            *reinterpret_cast<uint4*>(yrow + o * C) = pack8(a[o]);
        }
    }
}

// ------------------------------------------------------------------ squeeze-excite gate from the pooled sums (one block per crop)
// pool [N][strips][C]: per-strip sums of the depthwise output, written by the fused expand + depthwise kernel's epilogue (mbconv.hip) or by
// se_pool_kernel — the tensor itself is not read again (the nine full passes over it per recogniser forward were 0.9 ms of a 64-page step).
__global__ __launch_bounds__(256) void se_fc_kernel(const float* pool, int strips, const bf16_t* w1, const float* b1, const bf16_t* w2,
                                                    const float* b2, bf16_t* gate, int HW, int C, int mid) {
    extern __shared__ float sm[];  // mean[C], hid[mid]
    const int n = blockIdx.x, tid = threadIdx.x;
    float* mean = sm;
    float* hid = mean + C;
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
        for (int g = 0; g < strips; ++g) s += pool[((size_t)n * strips + g) * C + c];
        mean[c] = bf16_to_f32(f32_to_bf16(s / (float)HW));
    }
    __syncthreads();
    for (int m = tid; m < mid; m += 256) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = s + bf16_to_f32(w1[(size_t)m * C + c]) * mean[c];
        hid[m] = bf16_to_f32(f32_to_bf16(fmaxf(s + b1[m], 0.f)));
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
        for (int m = 0; m < mid; ++m) s = s + bf16_to_f32(w2[(size_t)c * mid + m]) * hid[m];
        gate[(size_t)n * C + c] = f32_to_bf16(apply_act(s + b2[c], ACT_HSIGMOID));
    }
}

__global__ void se_scale_kernel(const bf16_t* x, const bf16_t* gate, bf16_t* y, int N, int HW, int C) {
    const int cg = C >> 3;
    const size_t total = (size_t)N * HW * cg;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cg);
        const int n = (int)(i / ((size_t)HW * cg));
        float f[8], g[8];
        unpack8(*reinterpret_cast<const uint4*>(x + i * 8), f);
        unpack8(*reinterpret_cast<const uint4*>(gate + (size_t)n * C + c8 * 8), g);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = f[j] * g[j];
        *reinterpret_cast<uint4*>(y + i * 8) = pack8(f);
    }
}

// ------------------------------------------------------------------ LSTM recurrence on the matrix cores
// One workgroup = 32 sequences x one direction, 3 waves; wave u owns hidden units [32u, 32u+32).  Per time step
//   gates[4][32 units][32 seqs] = W_hh (A operand, resident in registers for all 80 steps) * h_{t-1} (B operand, LDS)
// with the accumulators initialised from the stored x-projection.  The four gate tiles of a (unit, sequence) pair
// land in the same lane / register slot, so the cell update is register-local; h_t goes to LDS (next step's operand)
// and to HBM (layer output) as packed 8-byte quads.  c stays fp32 in registers, h is bf16 (oracle/nets.py lstm_dir).
constexpr int LH = 96, LSEQ = 32, LPITCH = 208;  // bytes per sequence row of the h buffer (192 + 16: conflict-free b128 reads)

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 2.f * fast_sigmoid(2.f * x) - 1.f; }

__global__ __launch_bounds__(192) void lstm_kernel(const bf16_t* xproj, const bf16_t* whh, bf16_t* out, int N, int T) {
    __shared__ __attribute__((aligned(16))) unsigned char hbuf[2][LSEQ * LPITCH];
    const int tid = threadIdx.x, lane = tid & 63, u = tid >> 6, r = lane & 31, h = lane >> 5;
    const int dir = blockIdx.x & 1, n0 = (blockIdx.x >> 1) * LSEQ;
    const int seq = n0 + r;
    const bool valid = seq < N;
    const int sq = valid ? seq : N - 1;  // clamp loads of the ragged last group
    bf16x8_t afr[4][6];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int ks = 0; ks < 6; ++ks)
            afr[g][ks] = *reinterpret_cast<const bf16x8_t*>(whh + ((size_t)dir * 4 * LH + g * LH + 32 * u + r) * LH + ks * 16 + h * 8);
    for (int i = tid; i < 2 * LSEQ * LPITCH / 16; i += 192) reinterpret_cast<uint4*>(&hbuf[0][0])[i] = make_uint4(0, 0, 0, 0);
    float c[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) c[j] = 0.f;
    const size_t xrow = (size_t)sq * T;
    const int xcol = dir * 4 * LH + 32 * u + 4 * h;
    uint2 xq[4][4];
    {
        const int t = dir ? T - 1 : 0;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) xq[g][q] = *reinterpret_cast<const uint2*>(xproj + (xrow + t) * (8 * LH) + xcol + g * LH + 8 * q);
    }
    __syncthreads();
    int cur = 0;
    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
        f32x16_t acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[g][4 * q + 0] = __uint_as_float(xq[g][q].x << 16); acc[g][4 * q + 1] = __uint_as_float(xq[g][q].x & 0xFFFF0000u);
                acc[g][4 * q + 2] = __uint_as_float(xq[g][q].y << 16); acc[g][4 * q + 3] = __uint_as_float(xq[g][q].y & 0xFFFF0000u);
            }
        if (step + 1 < T) {  // prefetch the next step's x-projection under this step's MFMAs
            const int tn = dir ? (t - 1) : (t + 1);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) xq[g][q] = *reinterpret_cast<const uint2*>(xproj + (xrow + tn) * (8 * LH) + xcol + g * LH + 8 * q);
        }
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const bf16x8_t bfr = *reinterpret_cast<const bf16x8_t*>(&hbuf[cur][r * LPITCH + (ks * 16 + h * 8) * 2]);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[g][ks], bfr, acc[g], 0, 0, 0);
        }
        bf16_t* orow = out + ((size_t)sq * T + t) * (2 * LH) + dir * LH + 32 * u + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float hn[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 4 * q + e;
                const float ig = fast_sigmoid(acc[0][j]), fg = fast_sigmoid(acc[1][j]), gg = fast_tanh(acc[2][j]), og = fast_sigmoid(acc[3][j]);
                c[j] = fg * c[j] + ig * gg;
                hn[e] = og * fast_tanh(c[j]);
            }
            uint2 o;
            o.x = pack_bf16x2(hn[0], hn[1]);
            o.y = pack_bf16x2(hn[2], hn[3]);
            *reinterpret_cast<uint2*>(&hbuf[cur ^ 1][r * LPITCH + (32 * u + 8 * q + 4 * h) * 2]) = o;
            if (valid) *reinterpret_cast<uint2*>(orow + 8 * q) = o;
        }
        __syncthreads();
        cur ^= 1;
    }
}

// ------------------------------------------------------------------ CTC head: fused FC + argmax + softmax-max
// A-stationary MFMA GEMM: the block keeps its 128 sequence rows (K <= 192) in LDS and streams all
// 64-class weight tiles through a second LDS region; running (max, argmax, sum-exp) stay in registers,
// the [M, C] logits never exist in memory.
constexpr int CT_MT = 1, CT_NT = 2;  // per wave: CT_MT x 32 rows, CT_NT x 32 classes per weight tile (2 workgroups per CU)
constexpr int CT_ROWS = 4 * 32 * CT_MT, CT_BN = 32 * CT_NT, CT_KMAX = 192, CT_WIT = CT_BN * (CT_KMAX / 8) / 256;
constexpr int CT_PLANE_A = CT_ROWS + 4;  // entries, == 4 (mod 16)

typedef _Float16 ctc_f16x8_t __attribute__((ext_vector_type(8)));
template <int DT>   // storage type of the sequence and the weights: 0 bf16, 1 fp16 (SVTR fp16 mode)
__global__ __launch_bounds__(256, 2) void ctc_fc_argmax_kernel(const CtcFcParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int npl = p.K >> 3;
    unsigned char* sA = smem;
    unsigned char* sW = smem + (size_t)npl * CT_PLANE_A * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * CT_ROWS;

    for (int i = tid; i < CT_ROWS * npl; i += 256) {
        const int row = i / npl, c = i - row * npl;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row0 + row < p.M) v = *reinterpret_cast<const uint4*>(p.seq + (size_t)(row0 + row) * p.K + c * 8);
        *reinterpret_cast<uint4*>(sA + ((size_t)c * CT_PLANE_A + row) * 16) = v;
    }
    const int w_items = CT_BN * npl;           // 16-byte items per weight tile (<= 3072)
    const int wit = (w_items + 255) / 256;     // <= CT_WIT
    uint4 wreg[CT_WIT];
#define CTC_LOAD_W(tile_)                                                                              \
    {                                                                                                  \
        const uint4* src = reinterpret_cast<const uint4*>(p.wpk + (size_t)(tile_) * w_items * 8);      \
        _Pragma("unroll") for (int it = 0; it < CT_WIT; ++it) {                                            \
            const int i = tid + 256 * it;                                                              \
            uint4 t_ = make_uint4(0, 0, 0, 0);                                                         \
            if (it < wit && i < w_items) t_ = src[i];                                                  \
            wreg[it] = t_;                                                                             \
        }                                                                                              \
    }
    float run_m[CT_MT], run_s[CT_MT];
    int run_i[CT_MT];
#pragma unroll
    for (int mt = 0; mt < CT_MT; ++mt) { run_m[mt] = -3.0e38f; run_s[mt] = 0.f; run_i[mt] = 0; }

    CTC_LOAD_W(0);
    const int ksteps = p.K >> 4;
    for (int tile = 0; tile < p.ntiles; ++tile) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < CT_WIT; ++it) {
            const int i = tid + 256 * it;
            if (it < wit && i < w_items) *reinterpret_cast<uint4*>(sW + (size_t)i * 16) = wreg[it];
        }
        __syncthreads();
        if (tile + 1 < p.ntiles) CTC_LOAD_W(tile + 1);
        f32x16_t acc[CT_MT][CT_NT];
#pragma unroll
        for (int mt = 0; mt < CT_MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < CT_NT; ++nt)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[mt][nt][j] = 0.f;
        for (int kc = 0; kc < ksteps; ++kc) {
            typedef typename std::conditional<DT == 1, ctc_f16x8_t, bf16x8_t>::type frag_t;
            frag_t bfr[CT_MT], afr[CT_NT];
#pragma unroll
            for (int mt = 0; mt < CT_MT; ++mt)
                bfr[mt] = *reinterpret_cast<const frag_t*>(sA + ((size_t)(2 * kc + h) * CT_PLANE_A + wave * (32 * CT_MT) + mt * 32 + r) * 16);
#pragma unroll
            for (int nt = 0; nt < CT_NT; ++nt)
                afr[nt] = *reinterpret_cast<const frag_t*>(sW + ((size_t)(2 * kc + h) * CT_BN + nt * 32 + r) * 16);
#pragma unroll
            for (int mt = 0; mt < CT_MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < CT_NT; ++nt) {
                    if constexpr (DT == 1) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[nt], bfr[mt], acc[mt][nt], 0, 0, 0);
                    else acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[nt], bfr[mt], acc[mt][nt], 0, 0, 0);
                }
        }
        // per-tile reduction over this lane's 64 classes, then the partner half-wave, then the running state
        const float* bt = p.bias + tile * CT_BN;
#pragma unroll
        for (int mt = 0; mt < CT_MT; ++mt) {
            float tm = -3.0e38f;
            int ti = 0;
#pragma unroll
            for (int nt = 0; nt < CT_NT; ++nt)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int cls = nt * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
                    const float v = acc[mt][nt][j] + bt[cls];
                    acc[mt][nt][j] = v;
                    if (v > tm) { tm = v; ti = cls; }
                }
            const float om = __shfl_xor(tm, 32);
            const int oi = __shfl_xor(ti, 32);
            if (om > tm || (om == tm && oi < ti)) { tm = om; ti = oi; }
            float ts = 0.f;
#pragma unroll
            for (int nt = 0; nt < CT_NT; ++nt)
#pragma unroll
                for (int j = 0; j < 16; ++j) ts += __builtin_amdgcn_exp2f((acc[mt][nt][j] - tm) * 1.4426950408889634f);
            ts += __shfl_xor(ts, 32);
            if (tm > run_m[mt]) {
                run_s[mt] = run_s[mt] * __builtin_amdgcn_exp2f((run_m[mt] - tm) * 1.4426950408889634f) + ts;
                run_m[mt] = tm;
                run_i[mt] = tile * CT_BN + ti;
            } else {
                run_s[mt] += ts * __builtin_amdgcn_exp2f((tm - run_m[mt]) * 1.4426950408889634f);
            }
        }
    }
    if (h == 0) {
#pragma unroll
        for (int mt = 0; mt < CT_MT; ++mt) {
            const int row = row0 + wave * (32 * CT_MT) + mt * 32 + r;
            if (row < p.M) { p.out_idx[row] = run_i[mt]; p.out_prob[row] = 1.f / run_s[mt]; }
        }
    }
}

// ------------------------------------------------------------------ CTC greedy collapse (one wave per sequence)
__global__ __launch_bounds__(64) void ctc_collapse_kernel(const int* idx, const float* prob, int* text, int* len, float* score, int T) {
    __shared__ float kept_p[128];
    const int n = blockIdx.x, lane = threadIdx.x;
    int base = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        const int cur = t < T ? idx[(size_t)n * T + t] : 0;
        int prev = __shfl_up(cur, 1);
        if (lane == 0) prev = t0 > 0 ? idx[(size_t)n * T + t0 - 1] : -1;
        const bool keep = t < T && cur != 0 && cur != prev;
        const unsigned long long mask = __ballot(keep);
        const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
        if (keep) { text[(size_t)n * T + pos] = cur; kept_p[pos] = prob[(size_t)n * T + t]; }
        base += __popcll(mask);
    }
    for (int t = base + lane; t < T; t += 64) text[(size_t)n * T + t] = -1;
    __syncthreads();
    if (lane == 0) {
        float s = 0.f;
        for (int k = 0; k < base; ++k) s = s + kept_p[k];  // time order, fp32 (oracle/nets.py ctc_greedy)
        len[n] = base;
        score[n] = base ? s / (float)base : 0.f;
    }
}

int grid_for(size_t total) {
    size_t g = (total + 255) / 256;
    return (int)(g > 256 * 64 ? 256 * 64 : (g ? g : 1));
}

}  // namespace

hipError_t maxpool_launch(const bf16_t* x, bf16_t* y, int N, int H, int W, int C, int k, int s, int pad, int Ho, int Wo, hipStream_t st) {
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for((size_t)N * Ho * Wo * (C / 8))), dim3(256), 0, st, x, y, N, H, W, C, k, s, pad, Ho, Wo);
    return hipGetLastError();
}

hipError_t dwconv_launch(const bf16_t* x, const bf16_t* w, const float* bias, bf16_t* y, int N, int H, int W, int C, int k,
                         int sh, int act, hipStream_t st) {
    const int Ho = (H + 2 * (k / 2) - k) / sh + 1;
    const size_t total = (size_t)N * Ho * ((W + 3) / 4) * (C / 8);
    if (total > 0x7fffffffull || (k != 3 && k != 5)) return hipErrorInvalidValue;
    const size_t lds = ((size_t)k * k * C + C) * sizeof(float);
    if (lds > 60 * 1024) return hipErrorInvalidValue;
    const dim3 g(grid_for(total));
#define DW(K_, A_) hipLaunchKernelGGL((dwconv_kernel<K_, A_>), g, dim3(256), lds, st, x, w, bias, y, N, H, W, C, sh, Ho)
#define DW_ACT(K_)                                                     \
    switch (act) {                                                     \
        case ACT_NONE: DW(K_, ACT_NONE); break;                        \
        case ACT_RELU: DW(K_, ACT_RELU); break;                        \
        case ACT_HSWISH: DW(K_, ACT_HSWISH); break;                    \
        default: return hipErrorInvalidValue;                          \
    }
    if (k == 3) { DW_ACT(3) } else { DW_ACT(5) }
#undef DW_ACT
#undef DW
    return hipGetLastError();
}

hipError_t se_fc_launch(const float* pool, int strips, const bf16_t* w1, const float* b1, const bf16_t* w2, const float* b2, bf16_t* gate,
                        int N, int HW, int C, int mid, hipStream_t st) {
    hipLaunchKernelGGL(se_fc_kernel, dim3(N), dim3(256), ((size_t)C + mid) * sizeof(float), st, pool, strips, w1, b1, w2, b2, gate, HW, C, mid);
    return hipGetLastError();
}

hipError_t se_scale_launch(const bf16_t* x, const bf16_t* gate, bf16_t* y, int N, int HW, int C, hipStream_t st) {
    hipLaunchKernelGGL(se_scale_kernel, dim3(grid_for((size_t)N * HW * (C / 8))), dim3(256), 0, st, x, gate, y, N, HW, C);
    return hipGetLastError();
}

hipError_t lstm_recurrent_launch(const bf16_t* xproj, const bf16_t* whh, bf16_t* out, int N, int T, hipStream_t st) {
    hipLaunchKernelGGL(lstm_kernel, dim3(((N + LSEQ - 1) / LSEQ) * 2), dim3(192), 0, st, xproj, whh, out, N, T);
    return hipGetLastError();
}

size_t ctc_packed_weight_elems(int C, int K) { return (size_t)((C + CT_BN - 1) / CT_BN) * CT_BN * K; }

void pack_ctc_weights(const bf16_t* w, int C, int K, bf16_t* out) {
    const int ntiles = (C + CT_BN - 1) / CT_BN, npl = K / 8;
    for (int t = 0; t < ntiles; ++t)
        for (int c = 0; c < npl; ++c)
            for (int n = 0; n < CT_BN; ++n)
                for (int j = 0; j < 8; ++j) {
                    const int cls = t * CT_BN + n;
                    out[(((size_t)t * npl + c) * CT_BN + n) * 8 + j] = cls < C ? w[(size_t)cls * K + c * 8 + j] : (bf16_t)0;
                }
}

hipError_t ctc_fc_argmax_launch(const CtcFcParams& p, hipStream_t st) {
    if (p.K % 16 != 0 || p.K > CT_KMAX) return hipErrorInvalidValue;
    const size_t lds = (size_t)(p.K / 8) * (CT_PLANE_A + CT_BN) * 16;
    if (p.f16) {
        { hipError_t e = locr_dyn_lds(reinterpret_cast<const void*>(ctc_fc_argmax_kernel<1>), 160 * 1024); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(ctc_fc_argmax_kernel<1>, dim3((p.M + CT_ROWS - 1) / CT_ROWS), dim3(256), lds, st, p);
    } else {
        { hipError_t e = locr_dyn_lds(reinterpret_cast<const void*>(ctc_fc_argmax_kernel<0>), 160 * 1024); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(ctc_fc_argmax_kernel<0>, dim3((p.M + CT_ROWS - 1) / CT_ROWS), dim3(256), lds, st, p);
    }
    return hipGetLastError();
}

hipError_t ctc_collapse_launch(const int* idx, const float* prob, int* text, int* len, float* score, int N, int T, hipStream_t st) {
    if (T > 128) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3(N), dim3(64), 0, st, idx, prob, text, len, score, T);
    return hipGetLastError();
}
