// SVTR recogniser kernels (svtr.hip): token-major 16-bit tensors, storage type bf16 (dtype 0) or fp16 (dtype 1).
#pragma once
#include "common.h"

// Y[m, :] = epi( sum_taps X[src(m, tap), :] W_tap^T + bias ), see svtr.hip.  All pointers device.
struct SvtrGemmParams {
    const uint16_t* x;       // [rows][Cin]
    const uint16_t* w;       // [N][K] row-major, K = taps * Cin (tap-major: OHWI flattened)
    const float* bias;       // [N]
    const uint16_t* res;     // optional residual [M][N] (res_mod == 0) or table [res_mod][N] indexed by m % res_mod
    const float *gamma, *beta;  // optional LayerNorm over the N channels (then N = 64 / 128 / 256 / 384 and one tile spans it)
    uint16_t* y;             // [M][N]
    const uint16_t* zeros;   // >= 16 bytes of zeros (source of the padding taps)
    int M, K, N, act, res_mod;
    int res_post;            // 0: the residual is added before the activation / rounding; 1: y = T(T(act(..)) + res) (positional embedding)
    float eps;
    // gather geometry: output token m = (n, oy, ox) on a Tout = Hout x Wout grid reads input token (oy * sh + kh - 1, ox * sw + kw - 1)
    // of an Hin x Win grid; taps == 1: plain GEMM over rows (Tout = Wout = 1 is fine)
    int taps, Cin, Hin, Win, Tout, Wout, sh, sw;
};
hipError_t svtr_gemm_launch(const SvtrGemmParams& p, int dtype, hipStream_t st);
const char* svtr_gemm_kernel_name(const SvtrGemmParams& p, int dtype);
// crops u8 [N][32][320][3] (+ optional valid widths) -> normalised 3x3 / stride-2 patches [N * 16 * 160][32]
hipError_t svtr_im2col_launch(const uint8_t* crops, const int* widths, uint16_t* out, int N, int dtype, hipStream_t st);
// y[n,x,c] = mean_r x[n,r,x,c]
hipError_t svtr_rowmean_launch(const uint16_t* x, uint16_t* y, int N, int H, int W, int C, int dtype, hipStream_t st);
// qkv [N,T,3,heads,32] -> out [N,T,heads*32]; soft-max(QK^T / sqrt(32)) V per head, keys restricted to the 7x11 window of the
// gh x gw token grid when local != 0 (gw % 8 == 0, local: gh % 4 == 0)
hipError_t svtr_attention_launch(const uint16_t* qkv, uint16_t* out, int N, int T, int heads, int gh, int gw, int local, int dtype, hipStream_t st);
