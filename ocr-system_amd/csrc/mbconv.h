// Fused MobileNetV3 expand (1x1) + depthwise (K x K) kernel of the recogniser (mbconv.hip).
#pragma once
#include "common.h"

struct MbParams {
    const bf16_t* x;    // [N, H, W, cin] bf16 (cin % 16 == 0, <= 48)
    const bf16_t* we;   // expand weights, mbconv_pack_expand()
    const float* be;    // expand bias, fp32 [round_up(expc, 32)]
    const bf16_t* wd;   // depthwise weights [K*K][expc] bf16
    const float* bd;    // depthwise bias fp32 [expc]
    bf16_t* d;          // [N, Ho, W, expc]
    int N, H, W, cin, expc, Ho, act;
};

size_t mbconv_expand_packed_elems(int expc, int cin);
// w: [expc][cin] bf16 (already zero padded) -> [m tile][k step][half][32 rows][8]: one MFMA A fragment = two 512-byte runs.
void mbconv_pack_expand(const bf16_t* w, int expc, int cin, bf16_t* out);
size_t mbconv_lds_bytes(const MbParams& p, int k);
bool mbconv_supported(const MbParams& p, int k, int sh);
hipError_t mbconv_launch(const MbParams& p, int k, int sh, hipStream_t st);
