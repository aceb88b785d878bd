// Host-visible description of one implicit-GEMM convolution launch (NHWC bf16, MFMA).
#pragma once
#include "common.h"

enum ConvOutMode : int {
    OUT_NORMAL = 0,    // y[n, oy, ox, coff + co]
    OUT_UPSAMPLE = 1,  // nearest-neighbour replicate each output pixel f x f times (f = 1 << up_shift)
    OUT_CONVT = 2,     // 2x2/s2 transposed conv: gemm column q*C + co -> y[n, 2oy+dy, 2ox+dx, co], q = dy*2+dx
    OUT_CONVT1 = 3,    // same with C == 1 (probability map): gemm columns 0..3 -> 2x2 block of a 1-channel map
    OUT_POOL = 4,      // 3x3 / stride-2 / pad-1 max pool of the conv output, fused: y[n, (Ho-1)/2+1, (Wo-1)/2+1, co] (LDS-DMA 16x32-tile kernel only)
};

struct ConvParams {
    const bf16_t* x;    // [N, H, W, Cin]   (Cin % 16 == 0)
    const bf16_t* wpk;  // packed weights, see pack_conv_weights()
    const float* bias;  // [n_tiles * BN] (zero padded)
    const bf16_t* res;  // optional residual / top-down tensor, read at (oy >> res_shift, ox >> res_shift)
    bf16_t* y;
    int N, H, W, Cin;
    int Ho, Wo;       // conv output grid
    int Cout;         // real gemm columns (<= n_tiles * BN); chunks of 8 beyond round_up(Cout,8) are not stored
    int y_cstride, y_coff;
    int res_cstride, res_h, res_w, res_shift;
    int act;
    int out_mode, up_shift, convt_c;
    int tiles_x, tiles_y, n_tiles;  // spatial tiles per image, cout tiles
    // OUT_CONVT only: fuse a following 2x2/s2 transposed conv to ONE channel + sigmoid (DBHead's last layer) into the
    // epilogue: fuse_w [4][convt_c] bf16 (row q2 = dy2*2+dx2), fuse_b scalar; y is then the bf16 probability map
    // [N, 4*Ho, 4*Wo] and the intermediate 64-channel tensor is never written.
    const bf16_t* fuse_w;
    float fuse_b;
    // 1x1 only: squeeze-excite gate fused into the operand staging: x[n, p, c] is multiplied by gate[n, c] (bf16, one rounding)
    // before the contraction; gate_hw = pixels per image (the flat GEMM view hides the image boundaries)
    const bf16_t* gate;
    int gate_hw;
    int dbg_skip;   // timing experiments only (LUMINA_CONV_DBG): 1 skip weight reloads, 2 skip halo reloads, 8 channel-blocked addressing (wrong results); 32 = staged epilogue everywhere
    int pix_limit;  // flat-GEMM mode (1x1): pixels >= pix_limit of an image are neither read nor written (0 = off)
    const bf16_t* zeros;  // >= 16 bytes of zeros in device memory (LDS-DMA variant: source of the out-of-image halo)
    // channel-blocked tensors [n][C/16][H][W][16] (conv_ring.hip only): a 16-channel chunk of a pixel row is contiguous, every
    // chunk pass of the K loop touches its own cache lines exactly once (NHWC: 32 bytes of every 128-byte line per pass)
    int x_blk, y_blk, res_blk;
    // multi-source input (conv_ring.hip only): the Cin input channels are the concatenation of n_src NHWC tensors, source k
    // contributing xs_nchunk[k] chunks of 16 channels out of pixels xs_cstride[k] elements apart (a channel slice of a wider tensor:
    // xs[k] points at the slice's first channel), stored at 1 / 2^xs_shift[k] of the resolution and read nearest-upsampled —
    // DBHead's conv over the FPN concat [up8(p5), up4(p4), up2(p3), p2] and the composed FPN p2 over [c2, up2(out3)] without a
    // concat or an upsampled tensor ever being written.  n_src <= 1: the single tensor x.
    int n_src;
    const bf16_t* xs[4];
    int xs_shift[4];
    int xs_nchunk[4], xs_cstride[4];
    // fused vd shortcut (3x3 / stride-2 block-entry kernel only): the block's 2x2 / stride-2 shortcut conv reads pixels (2oy, 2ox) ..
    // (2oy+1, 2ox+1) — taps (1,1), (1,2), (2,1), (2,2) of the 3x3 window whose halo is already in LDS.  wpk2 = that layer's packed
    // weights (ks 2, same bn / ck), bias2 its bias; its output (no activation) goes to y2 [N, Ho, Wo, y2_cstride].  Same products in
    // the same (chunk, tap) order as the separate kernel: bit-identical, and the input is read once instead of twice.
    const bf16_t* wpk2;
    const float* bias2;
    bf16_t* y2;
    int y2_cstride;
};

struct ConvKernelCfg {
    int ks, stride, bn, ck, tw, nw;  // nw selects the variant: 4 = 8x32-pixel tile; 5 = 16x32 tile (4 rows per wave); 6 = 16x32 tile fed by LDS-DMA
};

// Size in bytes of the packed weight image for (cout, ks, cin) under cfg.
size_t conv_packed_weight_elems(int cout_gemm, int ks, int cin, int bn);
// Pack OHWI bf16 weights [cout_gemm][ks][ks][cin] into [ntile][chunk][plane][tap*BN + n][8].
// row_major != 0 (cfg.nw == 6, the LDS-DMA kernel): [ntile][chunk][tap*BN + n][ck].
void pack_conv_weights(const bf16_t* ohwi, int cout_gemm, int ks, int cin, int bn, int ck, bf16_t* out, int row_major = 0);

// Choose a kernel configuration for a layer; launch it. Returns hipSuccess or an error.
bool conv_pick_cfg(int ks, int stride, int cin, int cout_gemm, ConvKernelCfg* cfg);
hipError_t conv_launch(const ConvKernelCfg& cfg, ConvParams p, hipStream_t stream);
const char* conv_kernel_name(const ConvKernelCfg& cfg);

// Persistent LDS-DMA ring kernel (conv_ring.hip) for 3x3 / stride-1 / OUT_NORMAL layers packed for cfg.nw == 6: same tile,
// same summation order, bit-identical results; orientation -1 = the tile orientation that pads less, 0 / 1 forced.
bool conv_ring_supported(const ConvKernelCfg& cfg, const ConvParams& p);
bool conv_ring_transposed(const ConvParams& p, int orientation);
hipError_t conv_ring_launch(ConvParams p, int orientation, hipStream_t stream);
const char* conv_ring_kernel_name(const ConvParams& p, int orientation);

// Pixel-stationary pointwise kernel (conv_pw.hip) for Cin 64 / 128, all output channels per workgroup; same packed weights
// (bn 64, ck 32).  Output modes: OUT_NORMAL (optional top-down add) and OUT_CONVT with the fused DBHead tail.
bool conv_pw_supported(const ConvKernelCfg& cfg, const ConvParams& p);
hipError_t conv_pw_launch(const ConvParams& p, hipStream_t stream);

