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
    // squeeze-excite blocks: per (crop, 32-column strip, channel) sums of the STORED (bf16) outputs, fp32, in the fixed order of
    // se_pool_order (below) -> pool [N][strips][expc]; se_fc_launch (ops.h) finishes the gate.  nullptr: not wanted.
    float* pool;
};

// The order in which a strip's outputs are summed (shared by the fused kernel and the stand-alone se_pool kernel, so that both give
// the same bits): thread (pg, c8), pg < 256 / (C / 8), adds the pixels of its pixel groups t = pg, pg + groups, ... (t = oy * 8 + xg,
// 4 pixels each, left to right) for its 8 channels; the groups' sums are then added in pg order.
constexpr int MB_STRIP = 32;
inline int mb_strips(int W) { return (W + MB_STRIP - 1) / MB_STRIP; }

size_t mbconv_expand_packed_elems(int expc, int cin);
// w: [expc][cin] bf16 (already zero padded) -> [m tile][k step][half][32 rows][8]: one MFMA A fragment = two 512-byte runs.
void mbconv_pack_expand(const bf16_t* w, int expc, int cin, bf16_t* out);
size_t mbconv_lds_bytes(const MbParams& p, int k);
bool mbconv_supported(const MbParams& p, int k, int sh);
hipError_t mbconv_launch(const MbParams& p, int k, int sh, hipStream_t st);
// stand-alone twin of the fused kernel's pooling (for the unfused path): pool [N][strips][C] from d [N, Ho, W, C]
hipError_t se_pool_launch(const bf16_t* d, float* pool, int N, int Ho, int W, int C, hipStream_t st);
