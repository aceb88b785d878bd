// Reference pre-processing on the GPU, byte-exact with PIL (the reference's CPU path):
//   resize_if_needed  /root/reference/backend/utils/image_preprocessing.py:81-110  (LANCZOS, 8-bit, 22-bit fixed point,
//                     horizontal pass then vertical pass — Pillow's ImagingResample)
//   enhance_contrast  :132-144 (blend with the mean-gray image, factor 1.2)
//   enhance_sharpness :146-158 (blend with the 3x3 SMOOTH-filtered image, factor 1.1)
// Coefficient tables are computed on the host in double precision with libm sin(), exactly like Pillow's
// precompute_coeffs / normalize_coeffs_8bpc, and cached on the device per (in, out) size.
// HBM-bound: 3 B/px read + 3 B/px written per pass.
#include "resize.h"

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
    if (alpha >= 0.f && alpha <= 1.f) return (uint8_t)t;
    if (t <= 0.f) return 0;
    if (t >= 255.f) return 255;
    return (uint8_t)t;
}

// contrast: 4 bytes per thread (the image size H*W*3 need not be a multiple of 4: the tail is done bytewise)
__global__ void contrast_kernel(const uint8_t* img, uint8_t* out, const unsigned long long* sums, int HW, float alpha) {
    const int n = blockIdx.y;
    const int mean = (int)((double)sums[n] / (double)HW + 0.5);
    const unsigned total = (unsigned)HW * 3u;
    const uint8_t* p = img + (size_t)n * total;
    uint8_t* o = out + (size_t)n * total;
    const unsigned mis = (unsigned)(reinterpret_cast<uintptr_t>(p) & 3u);          // leading bytes until 4-byte alignment
    const unsigned head = mis ? 4u - mis : 0u;
    const bool same = (reinterpret_cast<uintptr_t>(o) & 3u) == mis;                // in/out equally aligned (they are: same layout)
    const unsigned words = same && total > head ? (total - head) / 4u : 0u;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(p + head + 4u * i);
        const uint32_t r0 = blend_u8(mean, (int)(w & 0xff), alpha), r1 = blend_u8(mean, (int)((w >> 8) & 0xff), alpha);
        const uint32_t r2 = blend_u8(mean, (int)((w >> 16) & 0xff), alpha), r3 = blend_u8(mean, (int)(w >> 24), alpha);
        *reinterpret_cast<uint32_t*>(o + head + 4u * i) = r0 | (r1 << 8) | (r2 << 16) | (r3 << 24);
    }
    if (blockIdx.x == 0) {
        for (unsigned i = threadIdx.x; i < head && i < total; i += blockDim.x) o[i] = blend_u8(mean, p[i], alpha);
        for (unsigned i = head + 4u * words + threadIdx.x; i < total; i += blockDim.x) o[i] = blend_u8(mean, p[i], alpha);
    }
}

// sharpen: grid (x-chunks, rows, images): no 64-bit index arithmetic, one output byte per thread, 9 neighbour bytes
__global__ void sharpen_kernel(const uint8_t* img, uint8_t* out, int H, int W, float alpha) {
    const int n = blockIdx.z, y = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;   // byte inside the row: x*3 + c
    const int rowb = W * 3;
    if (e >= rowb) return;
    const size_t base = ((size_t)n * H + y) * rowb;
    const uint8_t* r0 = img + base + e;
    const int v = r0[0];
    int deg = v;
    if (e >= 3 && e < rowb - 3 && y > 0 && y < H - 1) {
        const float k1 = 1.0f / 13.0f, k5 = 5.0f / 13.0f;
        const uint8_t* r1 = r0 + rowb;
        const uint8_t* r_1 = r0 - rowb;
        float ss = 0.5f;
        // Pillow order: row y+1 (kernel[0..2]), row y (kernel[3..5]), row y-1 (kernel[6..8]); left-to-right adds
        float a = __fmul_rn((float)r1[-3], k1);
        a = __fadd_rn(a, __fmul_rn((float)r1[0], k1));
        a = __fadd_rn(a, __fmul_rn((float)r1[3], k1));
        ss = __fadd_rn(ss, a);
        a = __fmul_rn((float)r0[-3], k1);
        a = __fadd_rn(a, __fmul_rn((float)r0[0], k5));
        a = __fadd_rn(a, __fmul_rn((float)r0[3], k1));
        ss = __fadd_rn(ss, a);
        a = __fmul_rn((float)r_1[-3], k1);
        a = __fadd_rn(a, __fmul_rn((float)r_1[0], k1));
        a = __fadd_rn(a, __fmul_rn((float)r_1[3], k1));
        ss = __fadd_rn(ss, a);
        deg = ss <= 0.f ? 0 : (ss >= 255.f ? 255 : (int)ss);
    }
    out[base + e] = blend_u8(deg, v, alpha);
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
                           int out_len, int axis, hipStream_t st) {
    if (axis == 0) {
        const size_t lds = ((size_t)W * C + 8 + 3) / 4 * 4;
        if (lds > 150 * 1024) return hipErrorInvalidValue;
        static bool attr = false;
        if (!attr) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(resample_h_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            if (e != hipSuccess) return e;
            attr = true;
        }
        hipLaunchKernelGGL(resample_h_kernel, dim3(N * H), dim3(256), lds, st, in, out, bounds_dev, kk_dev, ksize, H, W, C, out_len, (size_t)N * H * W * C);
    } else {
        const int rowbytes = W * C;
        const bool a4 = rowbytes % 4 == 0 && (reinterpret_cast<uintptr_t>(in) % 4 == 0) && (reinterpret_cast<uintptr_t>(out) % 4 == 0);
        const bool a2 = rowbytes % 2 == 0 && (reinterpret_cast<uintptr_t>(in) % 2 == 0) && (reinterpret_cast<uintptr_t>(out) % 2 == 0);
        if (a4) hipLaunchKernelGGL(resample_v_kernel<4>, dim3(N * out_len), dim3(256), 0, st, in, out, bounds_dev, kk_dev, ksize, H, rowbytes, out_len);
        else if (a2) hipLaunchKernelGGL(resample_v_kernel<2>, dim3(N * out_len), dim3(256), 0, st, in, out, bounds_dev, kk_dev, ksize, H, rowbytes, out_len);
        else hipLaunchKernelGGL(resample_v_kernel<1>, dim3(N * out_len), dim3(256), 0, st, in, out, bounds_dev, kk_dev, ksize, H, rowbytes, out_len);
    }
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
    hipLaunchKernelGGL(contrast_kernel, dim3(gx, N), dim3(256), 0, st, img, tmp, sums_dev, HW, contrast);
    hipLaunchKernelGGL(sharpen_kernel, dim3((W * 3 + 255) / 256, H, N), dim3(256), 0, st, tmp, out, H, W, sharpness);
    return hipGetLastError();
}
