// Bandwidth-bound helper kernels of the det/rec networks (NHWC bf16) and the CTC tail.
#pragma once
#include "common.h"

// max pool k x k / stride s / pad p over NHWC (C % 8 == 0)
hipError_t maxpool_launch(const bf16_t* x, bf16_t* y, int N, int H, int W, int C, int k, int s, int pad, int Ho, int Wo, hipStream_t st);

// depthwise k x k conv, stride (sh, 1), pad k/2; w: [k*k][C] bf16, bias fp32 [C]; fused act
hipError_t dwconv_launch(const bf16_t* x, const bf16_t* w, const float* bias, bf16_t* y, int N, int H, int W, int C, int k,
                         int sh, int act, hipStream_t st);

// squeeze-excite gate: s = hsigmoid(W2 * relu(W1 * mean_hw(x) + b1) + b2), all stored values bf16, from the per-strip pooled sums
// pool [N][strips][C] (mbconv.h: written by the fused expand + depthwise kernel or by se_pool_launch).
// w1: [mid][C] bf16, w2: [C][mid] bf16 (row-major, unpadded mid), gate: [N][C] bf16
hipError_t se_fc_launch(const float* pool, int strips, const bf16_t* w1, const float* b1, const bf16_t* w2, const float* b2, bf16_t* gate,
                        int N, int HW, int C, int mid, hipStream_t st);
// y = bf16(x * gate[n, c])
hipError_t se_scale_launch(const bf16_t* x, const bf16_t* gate, bf16_t* y, int N, int HW, int C, hipStream_t st);

// LSTM recurrence for one layer, both directions. xproj: [N][T][2*4H] bf16 (fw gates then bw gates, bias included),
// whh: [2][4H][H] bf16, out: [N][T][2H] bf16 (fw | bw). H == 96.
hipError_t lstm_recurrent_launch(const bf16_t* xproj, const bf16_t* whh, bf16_t* out, int N, int T, hipStream_t st);

// CTC head: logits = seq[M][K] * W^T[K][C] + b, never materialised; per row -> argmax index + softmax max-prob.
// wpk: weights packed [ntile of 64 classes][plane][64 rows][8] (pack_ctc_weights).
struct CtcFcParams {
    const bf16_t* seq;  // [M][K], K % 32 == 0
    const bf16_t* wpk;
    const float* bias;  // [ntiles*64], padded entries = -1e30 (never win)
    float* part_max; int* part_idx; float* part_sum;  // [M][ntiles]
    int* out_idx; float* out_prob;                    // [M]
    int M, K, C, ntiles;
    int f16;   // sequence and packed weights are fp16 (SVTR fp16 mode) instead of bf16
};
size_t ctc_packed_weight_elems(int C, int K);
void pack_ctc_weights(const bf16_t* w /*[C][K]*/, int C, int K, bf16_t* out);
hipError_t ctc_fc_argmax_launch(const CtcFcParams& p, hipStream_t st);

// CTC greedy collapse: idx/prob [N][T] -> text [N][T] (class indices, -1 padded), len [N], score [N]
hipError_t ctc_collapse_launch(const int* idx, const float* prob, int* text, int* len, float* score, int N, int T, hipStream_t st);
