#pragma once
#include "common.h"

struct StemParams {
    const uint8_t* x;  // [N, H, W, 3] u8
    const bf16_t* wpk; // packed [2][2][32][8]
    const float* bias; // [32] zero padded
    bf16_t* y;         // [N, Ho, Wo, Cout_store]
    const int* valid_w_per_img;  // optional per-image valid width (rec crops); else valid_w
    int N, H, W;       // buffer dims of x
    int valid_h, valid_w;
    int Ho, Wo, Cout_store;  // Cout_store in {8,16,24,32}
    int act;
    float scale[3], shift[3];
};

void pack_stem_weights(const bf16_t* ohwi, int cout, bf16_t* out);
hipError_t stem_conv_launch(const StemParams& p, hipStream_t stream);
// DBNet stem.conv1 + stem.conv2 fused: p describes conv1 (3 -> 32, stride 2, Cout_store 32) and the output grid; y receives conv2's
// output [N, Ho, Wo, 32].  w2pk / bias2: stem.conv2 in conv_mfma's packing for bn = 32, ck = 16 (plane layout).
hipError_t stem12_launch(const StemParams& p, const bf16_t* w2pk, const float* bias2, hipStream_t stream);
hipError_t normalize_launch(const uint8_t* x, bf16_t* y, int N, int H, int W, int Hp, int Wp, int vh, int vw,
                            const float* scale, const float* shift, int nchw, hipStream_t stream);
