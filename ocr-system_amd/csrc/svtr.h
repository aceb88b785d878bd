// SVTR-Tiny recogniser: glue kernels between the linear layers (svtr.hip).  NHWC / token-major bf16 tensors.
#pragma once
#include "common.h"

// y[n,t,c] = bf16(x[n,t,c] + pos[t,c]);  T*C % 8 == 0
hipError_t svtr_add_pos_launch(const bf16_t* x, const bf16_t* pos, bf16_t* y, int N, int T, int C, hipStream_t st);
// LayerNorm over C (64 / 128 / 256) of the tokens (n, oy*row_step, x) of x [N,Hin,W,C] -> y [N,Hout,W,C]
hipError_t svtr_layernorm_launch(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y, int N, int Hin, int Hout, int W, int C, int row_step,
                                 float eps, hipStream_t st);
// y[n,x,c] = bf16(mean_r x[n,r,x,c])
hipError_t svtr_rowmean_launch(const bf16_t* x, bf16_t* y, int N, int H, int W, int C, hipStream_t st);
// qkv [N,T,3,heads,32] -> out [N,T,heads*32]; soft-max(QK^T / sqrt(32)) V per head, keys restricted to the 7x11 window of the
// gh x gw token grid when local != 0
hipError_t svtr_attention_launch(const bf16_t* qkv, bf16_t* out, int N, int T, int heads, int gh, int gw, int local, hipStream_t st);
