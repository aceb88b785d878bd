// SVTR-Tiny recogniser glue kernels (the linear layers and convolutions run on conv_mfma.hip / stem_conv.hip):
// positional-embedding add, LayerNorm (with the row selection of the stride-(2,1) merging conv and the final row pooling), and
// the mixing blocks' attention core (local 7 x 11 window or global).  Arithmetic definition: oracle/nets.py svtr_backbone —
// bf16 stored tensors, fp32 inside (scores, soft-max, probability-weighted sum, mean / variance), one rounding per stored tensor.
// No reference counterpart exists (BASELINE configs[4] names the model; SURVEY.md §0.5: nothing of it ships): parity unpinned.
#include "svtr.h"

#include <cstdlib>

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// y[n, t, c] = bf16(x[n, t, c] + pos[t, c])
__global__ void svtr_add_pos_kernel(const bf16_t* x, const bf16_t* pos, bf16_t* y, size_t total8, int tc8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 a = reinterpret_cast<const uint4*>(x)[i], b = reinterpret_cast<const uint4*>(pos)[i % tc8];
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = pack_bf16x2(__uint_as_float(aw[j] << 16) + __uint_as_float(bw[j] << 16), __uint_as_float(aw[j] & 0xFFFF0000u) + __uint_as_float(bw[j] & 0xFFFF0000u));
        reinterpret_cast<uint4*>(y)[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// LayerNorm over C, one wave per output token.  Input token of output (n, oy, x) is (n, oy * row_step, x) of an [N, Hin, W, C]
// tensor (row_step = 2: the stride-(2,1) merging conv was computed at stride 1 and only its even rows are kept).
template <int CPL>  // channels per lane: C = 64 * CPL
__global__ __launch_bounds__(256) void svtr_layernorm_kernel(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y, int N, int Hin, int Hout, int W,
                                                             int row_step, float eps) {
    constexpr int C = 64 * CPL;
    const int lane = threadIdx.x & 63;
    const long long tok = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= (long long)N * Hout * W) return;
    const int xw = (int)(tok % W);
    const long long t2 = tok / W;
    const int oy = (int)(t2 % Hout), n = (int)(t2 / Hout);
    const bf16_t* src = x + (((size_t)n * Hin + (size_t)oy * row_step) * W + xw) * C + lane * CPL;
    float v[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) v[j] = bf16_to_f32(src[j]);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) s += v[j];
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { const float d = v[j] - mu; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    bf16_t* dst = y + (size_t)tok * C + lane * CPL;
#pragma unroll
    for (int j = 0; j < CPL; ++j) dst[j] = f32_to_bf16((v[j] - mu) * rstd * gamma[lane * CPL + j] + beta[lane * CPL + j]);
}

// y[n, x, c] = bf16(mean over the H rows of x[n, :, x, c])   (H = 2 at the end of the backbone)
__global__ void svtr_rowmean_kernel(const bf16_t* x, bf16_t* y, int N, int H, int W, int C) {
    const size_t total = (size_t)N * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t t = i / C;
        const int xw = (int)(t % W), n = (int)(t / W);
        float s = 0.f;
        for (int r = 0; r < H; ++r) s += bf16_to_f32(x[(((size_t)n * H + r) * W + xw) * C + c]);
        y[i] = f32_to_bf16(s / (float)H);
    }
}

// Attention core, head dimension 32.  One workgroup = 64 queries of one (crop, head); the head's K ([T][32], expanded to fp32 so
// that the score loop has no unpacking) and V ([T][32] bf16) are staged in LDS; thread (q, part) walks every AT_PARTS-th key of the
// query's key set (the 7 x 11 window for local blocks, all T keys otherwise) ONCE with an online soft-max (running maximum,
// rescaled sum and weighted V accumulator); the AT_PARTS partial soft-maxes of a query are merged through LDS in a fixed order.  fp32 throughout,
// one bf16 rounding of the output.
constexpr int AT_HD = 32, AT_Q = 64, AT_RED = 36;   // floats per (part, query) in the merge buffer: m, l, o[32] (+pad)
constexpr int AT_PARTS = 8;                          // threads per query (512-thread workgroups: two waves per SIMD at one workgroup per CU)
__global__ __launch_bounds__(64 * AT_PARTS) void svtr_attn_kernel(const bf16_t* qkv, bf16_t* out, int T, int heads, int gh, int gw, int local) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sK = reinterpret_cast<float*>(smem);                                  // [T][32] fp32
    bf16_t* sV = reinterpret_cast<bf16_t*>(smem + (size_t)T * AT_HD * 4);         // [T][32] bf16
    float* red = reinterpret_cast<float*>(smem);                                 // aliases sK after the key loop
    const int tid = threadIdx.x, ql = tid & 63, part = tid >> 6;
    const int n = blockIdx.z, hd = blockIdx.y, q0 = blockIdx.x * AT_Q;
    const int C = heads * AT_HD;
    const bf16_t* base = qkv + (size_t)n * T * 3 * C;
    for (int i = tid; i < T * 4; i += 64 * AT_PARTS) {   // token t, 8-channel slice s4
        const int t = i >> 2, s4 = i & 3;
        const uint4 kv = *reinterpret_cast<const uint4*>(base + ((size_t)t * 3 + 1) * C + hd * AT_HD + s4 * 8);
        reinterpret_cast<uint4*>(sV)[i] = *reinterpret_cast<const uint4*>(base + ((size_t)t * 3 + 2) * C + hd * AT_HD + s4 * 8);
        float4* kd = reinterpret_cast<float4*>(sK + (size_t)t * AT_HD + s4 * 8);
        kd[0] = make_float4(__uint_as_float(kv.x << 16), __uint_as_float(kv.x & 0xFFFF0000u), __uint_as_float(kv.y << 16), __uint_as_float(kv.y & 0xFFFF0000u));
        kd[1] = make_float4(__uint_as_float(kv.z << 16), __uint_as_float(kv.z & 0xFFFF0000u), __uint_as_float(kv.w << 16), __uint_as_float(kv.w & 0xFFFF0000u));
    }
    const int qt = min(q0 + ql, T - 1);
    float q[AT_HD];
    {
        const uint4* qp = reinterpret_cast<const uint4*>(base + (size_t)qt * 3 * C + hd * AT_HD);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const uint4 v = qp[s4];
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { q[s4 * 8 + 2 * j] = __uint_as_float(w4[j] << 16); q[s4 * 8 + 2 * j + 1] = __uint_as_float(w4[j] & 0xFFFF0000u); }
        }
    }
    __syncthreads();
    const float scale = 0.17677669529663687f;  // 32^-0.5
    const int qy = qt / gw, qx = qt - qy * gw;
    const int nkeys = local ? 77 : T;
    float m = -3.0e38f, l = 0.f, o[AT_HD];
#pragma unroll
    for (int d = 0; d < AT_HD; ++d) o[d] = 0.f;
    for (int i = part; i < nkeys; i += AT_PARTS) {
        int key = i;
        if (local) {
            const int wy = i / 11, ky = qy - 3 + wy, kx = qx - 5 + (i - wy * 11);
            key = (ky >= 0 && ky < gh && kx >= 0 && kx < gw) ? ky * gw + kx : -1;
        }
        if (key < 0) continue;
        const float4* kp = reinterpret_cast<const float4*>(sK + (size_t)key * AT_HD);
        float sc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 kk = kp[j];
            sc = __builtin_fmaf(q[4 * j], kk.x, sc); sc = __builtin_fmaf(q[4 * j + 1], kk.y, sc);
            sc = __builtin_fmaf(q[4 * j + 2], kk.z, sc); sc = __builtin_fmaf(q[4 * j + 3], kk.w, sc);
        }
        sc *= scale;
        if (sc > m) {   // new running maximum: rescale what has been accumulated (exp(-huge) = 0 on the first key)
            const float corr = expf(m - sc);
            l *= corr;
#pragma unroll
            for (int d = 0; d < AT_HD; ++d) o[d] *= corr;
            m = sc;
        }
        const float pr = expf(sc - m);
        l += pr;
        const uint4* vp = reinterpret_cast<const uint4*>(sV + (size_t)key * AT_HD);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const uint4 v = vp[s4];
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[s4 * 8 + 2 * j] = __builtin_fmaf(pr, __uint_as_float(w4[j] << 16), o[s4 * 8 + 2 * j]);
                o[s4 * 8 + 2 * j + 1] = __builtin_fmaf(pr, __uint_as_float(w4[j] & 0xFFFF0000u), o[s4 * 8 + 2 * j + 1]);
            }
        }
    }
    __syncthreads();   // every thread is done with sK: the merge buffer may overwrite it
    float* mine = red + (part * 64 + ql) * AT_RED;
    mine[0] = m; mine[1] = l;
#pragma unroll
    for (int d = 0; d < AT_HD; ++d) mine[2 + d] = o[d];
    __syncthreads();
    if (part == 0 && q0 + ql < T) {   // merge the partial soft-maxes in a fixed order
        float mt = red[ql * AT_RED];
#pragma unroll
        for (int pp = 1; pp < AT_PARTS; ++pp) mt = fmaxf(mt, red[(pp * 64 + ql) * AT_RED]);
        float f[AT_PARTS], lt = 0.f;
#pragma unroll
        for (int pp = 0; pp < AT_PARTS; ++pp) { f[pp] = expf(red[(pp * 64 + ql) * AT_RED] - mt); lt += red[(pp * 64 + ql) * AT_RED + 1] * f[pp]; }
        const float inv = 1.0f / lt;
        uint32_t w[16];
#pragma unroll
        for (int d = 0; d < AT_HD; d += 2) {
            float a = 0.f, b2 = 0.f;
#pragma unroll
            for (int pp = 0; pp < AT_PARTS; ++pp) { a += red[(pp * 64 + ql) * AT_RED + 2 + d] * f[pp]; b2 += red[(pp * 64 + ql) * AT_RED + 3 + d] * f[pp]; }
            w[d >> 1] = pack_bf16x2(a * inv, b2 * inv);
        }
        uint4* dst = reinterpret_cast<uint4*>(out + ((size_t)n * T + q0 + ql) * C + hd * AT_HD);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) dst[s4] = make_uint4(w[4 * s4], w[4 * s4 + 1], w[4 * s4 + 2], w[4 * s4 + 3]);
    }
}

// Attention core on the matrix cores (flash-attention form, head dimension 32).  One workgroup = MA_W waves x 32 queries of one
// (crop, head).  K ([T][32] bf16, rows padded to 80 B) and V TRANSPOSED ([32][T] bf16, rows padded) are staged in LDS once.
// Per 32-key tile a wave computes S^T[key][query] = K Q^T with two 32x32x16 MFMAs (A = K rows from LDS, B = the wave's Q
// rows, held in registers), masks the keys outside the 7 x 11 window (local blocks), updates the running maximum / sum of its
// queries (a query's 32 scores live in 16 registers of lane q and 16 of lane q + 32: one cross-half shuffle per reduction),
// converts P to bf16 (v_permlane32_swap turns the accumulator layout into the 8-consecutive-keys operand layout) and accumulates
// O^T[d][query] += V^T P^T with two more MFMAs.  Soft-max statistics are fp32; P is rounded to bf16 for the second product.
constexpr int MA_W = 8, MA_KP = 40;   // waves per workgroup; K row pitch in elements (80 B: conflict-free 16-byte fragment reads)
__global__ __launch_bounds__(64 * MA_W) void svtr_attn_mfma_kernel(const bf16_t* qkv, bf16_t* out, int T, int heads, int gh, int gw, int local) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int vp = T + 8;                                   // V^T row pitch in elements ((2T + 16) B: odd multiple of 16 B mod 128)
    bf16_t* sK = reinterpret_cast<bf16_t*>(smem);          // [Tpad][MA_KP]
    const int Tpad = (T + 31) & ~31;
    bf16_t* sVt = sK + (size_t)Tpad * MA_KP;               // [32][vp]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int n = blockIdx.z, hd = blockIdx.y, q0 = blockIdx.x * (32 * MA_W) + wave * 32;
    const int C = heads * AT_HD;
    const bf16_t* base = qkv + (size_t)n * T * 3 * C;
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
            sVt[(size_t)(s4 * 8 + 2 * j) * vp + t] = (bf16_t)(w4[j] & 0xffffu);
            sVt[(size_t)(s4 * 8 + 2 * j + 1) * vp + t] = (bf16_t)(w4[j] >> 16);
        }
    }
    // the wave's Q rows as the MFMA B operand: lane (r = query, h) holds d = 16*ks + 8*h .. +7
    const int qt = min(q0 + r, T - 1);
    bf16x8_t qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(base + (size_t)qt * 3 * C + hd * AT_HD + ks * 16 + h * 8);
    __syncthreads();
    if (q0 >= T) return;
    const float scale = 0.17677669529663687f;
    const int qy = qt / gw, qx = qt - qy * gw;
    const int qy_lo = q0 / gw, qy_hi = min(q0 + 31, T - 1) / gw;
    float m = -3.0e38f, l = 0.f;
    f32x16_t o;
#pragma unroll
    for (int j = 0; j < 16; ++j) o[j] = 0.f;
    for (int k0 = 0; k0 < Tpad; k0 += 32) {
        if (local) {   // whole key tile outside the row band of this query tile: nothing to add
            const int ky_lo = k0 / gw, ky_hi = min(k0 + 31, T - 1) / gw;
            if (ky_hi < qy_lo - 3 || ky_lo > qy_hi + 3) continue;
        }
        f32x16_t sc;
#pragma unroll
        for (int j = 0; j < 16; ++j) sc[j] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sK + (size_t)(k0 + r) * MA_KP + ks * 16 + h * 8);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sc, 0, 0, 0);
        }
        // lane (query = r): register j is key k0 + (j & 3) + 8 * (j >> 2) + 4 * h
        float tmax = -3.0e38f;
        const int ky0 = k0 / gw, kx0 = k0 - ky0 * gw;   // grid position of the tile's first key (a tile of 32 wraps at most once: gw >= 32)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int off = (j & 3) + 8 * (j >> 2) + 4 * h;
            bool ok = k0 + off < T;
            if (local) {
                int kx = kx0 + off, ky = ky0;
                if (kx >= gw) { kx -= gw; ++ky; }
                ok = ok && ky >= qy - 3 && ky <= qy + 3 && kx >= qx - 5 && kx <= qx + 5;
            }
            sc[j] = ok ? sc[j] * scale : -3.0e38f;
            tmax = fmaxf(tmax, sc[j]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mn = fmaxf(m, tmax);
        const float corr = expf(m - mn);     // (both -3e38 before the first valid key: exp(0) = 1 scales zeros)
        float psum = 0.f;
        uint32_t pk[8];
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            const float p0 = sc[j] > -1.0e38f ? expf(sc[j] - mn) : 0.f, p1 = sc[j + 1] > -1.0e38f ? expf(sc[j + 1] - mn) : 0.f;
            psum += p0 + p1;
            pk[j >> 1] = pack_bf16x2(p0, p1);
        }
        psum += __shfl_xor(psum, 32);
        l = l * corr + psum;
        m = mn;
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j] *= corr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // keys 16*ks .. +15 of the tile: X = registers 8*ks .. +3 (packed pk[4ks], pk[4ks+1]), Y = registers 8*ks+4 .. +7
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[4 * ks], pk[4 * ks + 2], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[4 * ks + 1], pk[4 * ks + 3], false, false);
            union { uint32_t u[4]; bf16x8_t v; } pf;
            pf.u[0] = s0[0]; pf.u[1] = s1[0]; pf.u[2] = s0[1]; pf.u[3] = s1[1];
            const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(sVt + (size_t)r * vp + k0 + ks * 16 + h * 8);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf.v, o, 0, 0, 0);
        }
    }
    if (q0 + r < T) {   // lane (query = r): register j is d = (j & 3) + 8 * (j >> 2) + 4 * h
        const float inv = 1.0f / l;
        bf16_t* dst = out + ((size_t)n * T + q0 + r) * C + hd * AT_HD;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w;
            w.x = pack_bf16x2(o[4 * g] * inv, o[4 * g + 1] * inv);
            w.y = pack_bf16x2(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
            *reinterpret_cast<uint2*>(dst + 8 * g + 4 * h) = w;
        }
    }
}

inline int grid_for(size_t total) {
    size_t g = (total + 255) / 256;
    return (int)(g > 256 * 32 ? 256 * 32 : (g ? g : 1));
}

}  // namespace

hipError_t svtr_add_pos_launch(const bf16_t* x, const bf16_t* pos, bf16_t* y, int N, int T, int C, hipStream_t st) {
    const size_t total8 = (size_t)N * T * C / 8;
    hipLaunchKernelGGL(svtr_add_pos_kernel, dim3(grid_for(total8)), dim3(256), 0, st, x, pos, y, total8, T * C / 8);
    return hipGetLastError();
}

hipError_t svtr_layernorm_launch(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y, int N, int Hin, int Hout, int W, int C, int row_step,
                                 float eps, hipStream_t st) {
    const long long tokens = (long long)N * Hout * W;
    const dim3 grid((unsigned)((tokens + 3) / 4));
    if (C == 64) hipLaunchKernelGGL(svtr_layernorm_kernel<1>, grid, dim3(256), 0, st, x, gamma, beta, y, N, Hin, Hout, W, row_step, eps);
    else if (C == 128) hipLaunchKernelGGL(svtr_layernorm_kernel<2>, grid, dim3(256), 0, st, x, gamma, beta, y, N, Hin, Hout, W, row_step, eps);
    else if (C == 256) hipLaunchKernelGGL(svtr_layernorm_kernel<4>, grid, dim3(256), 0, st, x, gamma, beta, y, N, Hin, Hout, W, row_step, eps);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t svtr_rowmean_launch(const bf16_t* x, bf16_t* y, int N, int H, int W, int C, hipStream_t st) {
    hipLaunchKernelGGL(svtr_rowmean_kernel, dim3(grid_for((size_t)N * W * C)), dim3(256), 0, st, x, y, N, H, W, C);
    return hipGetLastError();
}

hipError_t svtr_attention_launch(const bf16_t* qkv, bf16_t* out, int N, int T, int heads, int gh, int gw, int local, hipStream_t st) {
    if (gh * gw != T || N <= 0 || gw < 32) return hipErrorInvalidValue;
    size_t lds = (size_t)T * AT_HD * (4 + 2);   // K fp32 + V bf16; the merge buffer aliases K
    if (lds < (size_t)AT_PARTS * 64 * AT_RED * sizeof(float)) lds = (size_t)AT_PARTS * 64 * AT_RED * sizeof(float);
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    { hipError_t e = locr_dyn_lds(reinterpret_cast<const void*>(svtr_attn_kernel), 150 * 1024); if (e != hipSuccess) return e; }
    static const bool valu = getenv("LUMINA_SVTR_ATTN_VALU") != nullptr;   // A/B switch: the fp32 VALU kernel
    if (!valu && T % 32 == 0) {
        const int Tpad = (T + 31) & ~31;
        const size_t lds2 = (size_t)Tpad * MA_KP * 2 + (size_t)32 * (T + 8) * 2;
        { hipError_t e = locr_dyn_lds(reinterpret_cast<const void*>(svtr_attn_mfma_kernel), 150 * 1024); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(svtr_attn_mfma_kernel, dim3((T + 32 * MA_W - 1) / (32 * MA_W), heads, N), dim3(64 * MA_W), lds2, st, qkv, out, T, heads, gh, gw, local);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(svtr_attn_kernel, dim3((T + AT_Q - 1) / AT_Q, heads, N), dim3(64 * AT_PARTS), lds, st, qkv, out, T, heads, gh, gw, local);
    return hipGetLastError();
}
