// Host-side engine: weight loading/packing, workspace, and the det / rec layer schedules.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.h"
#include "conv_mfma.h"
#include "mbconv.h"
#include "svtr.h"

struct Tensor4 {
    bf16_t* p = nullptr;
    int n = 0, h = 0, w = 0, c = 0;
    bool blk = false;  // channel-blocked layout [n][c/16][h][w][16] instead of NHWC (conv_mfma.h: ConvParams::x_blk)
    // virtual concat (ConvParams::n_src): the c channels are n_src tensors, source k at 1 / 2^shift[k] resolution contributing xs_c[k]
    // channels (0: c / n_src) out of pixels xs_cs[k] channels wide (0: xs_c[k])
    int n_src = 0; const bf16_t* xs[4] = {nullptr, nullptr, nullptr, nullptr}; int xs_shift[4] = {0, 0, 0, 0};
    int xs_c[4] = {0, 0, 0, 0}, xs_cs[4] = {0, 0, 0, 0};
    int src_c(int k) const { return xs_c[k] ? xs_c[k] : c / n_src; }
    int src_cs(int k) const { return xs_cs[k] ? xs_cs[k] : src_c(k); }
    size_t elems() const { return (size_t)n * h * w * c; }
};

struct ConvLayer {
    std::string name;
    int ks = 1, stride = 1, cin = 0, cout = 0;  // padded gemm dims
    int act = ACT_NONE;
    bool convt = false;
    int convt_c = 0;
    ConvKernelCfg cfg{};
    bf16_t* wpk = nullptr;  // device
    // second packing for the 16x32-tile / 4x2-register-tile kernel (3x3, stride 1, >= 64 input channels): picked per launch
    // when the grid is large enough to fill the chip with the bigger tiles
    ConvKernelCfg cfg_big{}; bf16_t* wpk_big = nullptr;
    bool force_big = false;   // conv2d test hook: take the 16x32-tile kernel whatever the grid size / channel count
    int launch_group = 0;     // > 0: images per launch (measured: the 256-channel 1/4-resolution layers run 7 % faster in launches of 16 pages than of 64)
    bool small_only = false;  // never switch to the 16x32-tile kernel (layers whose maps are only 4-8 rows high)
    bf16_t* fuse_w = nullptr; float fuse_b = 0.f;  // optional fused DBHead tail (see ConvParams)
    double alg_flop_per_px = 0;   // > 0: algorithmic FLOP per output pixel of the architecture's layers this launch replaces (a composed layer executes more)
    float* bias = nullptr;  // device, n_tiles*BN
};

struct DwLayer { int k = 3, c = 0; bf16_t* w = nullptr; float* bias = nullptr; };
struct SeLayer { int c = 0, mid = 0; bf16_t *w1 = nullptr, *w2 = nullptr; float *b1 = nullptr, *b2 = nullptr; };

struct RecBlock {
    int k, cin, exp, cout, se_mid, stride_h, act;
    bool se, res;
    ConvLayer expand, project;
    DwLayer dw;
    bf16_t* we_pk = nullptr; float* be_pk = nullptr;  // expand weights / bias in the fused expand+depthwise kernel's layout (mbconv.h)
    SeLayer sel;
};

// SVTR recogniser (arch.svtr_block_table; Tiny or Base: the dimensions travel in the blob's svtr.config tensor).  Every linear layer
// and every convolution is one svtr_gemm launch (svtr.h); proj / fc2 / the merging convs carry the LayerNorm that follows them.
struct SvtrLinear {
    int K = 0, N = 0, taps = 1, cin = 0, act = ACT_NONE;
    uint16_t* w = nullptr;   // device, [N][K] in the model's storage type
    float* bias = nullptr;
    float *gamma = nullptr, *beta = nullptr;   // LayerNorm fused into the epilogue (or null)
};
struct SvtrBlock { int dim = 0, heads = 0, gh = 0, gw = 0; bool local = false; SvtrLinear qkv, proj, fc1, fc2; };
struct SvtrModel {
    bool loaded = false;
    int dtype = 0;            // storage / MFMA type: 0 bf16, 1 fp16
    int dims[3] = {64, 128, 256}, depths[3] = {3, 6, 3}, heads[3] = {2, 4, 8}, local_blocks = 6, out_ch = 192;
    int num_classes = 0, ctc_ntiles = 0;
    SvtrLinear pe1, pe2, sub[2], last;
    uint16_t* pos = nullptr;  // [640][dims[0]]
    std::vector<SvtrBlock> blocks;
    bf16_t* ctc_wpk = nullptr; float* ctc_bias = nullptr;
};

struct HostBlobTensor { int dtype; std::vector<int> dims; const uint8_t* data; size_t nbytes; };

struct lumina_ocr {
    int device = 0;
    std::string err;
    // ---- det ----
    bool det_loaded = false;
    std::map<std::string, ConvLayer> det;
    bf16_t* stem_wpk = nullptr; float* stem_bias = nullptr;
    // ---- rec ----
    bool rec_loaded = false;
    int num_classes = 0, ctc_ntiles = 0;
    bf16_t* rstem_wpk = nullptr; float* rstem_bias = nullptr;
    std::vector<RecBlock> rblocks;
    ConvLayer rconv2, xproj[2];
    bf16_t* whh[2] = {nullptr, nullptr};
    bf16_t* ctc_wpk = nullptr; float* ctc_bias = nullptr;
    SvtrModel svtr;
    // ---- workspace ----
    uint8_t* ws = nullptr; size_t ws_cap = 0, ws_off = 0;
    std::vector<void*> owned;  // device allocations freed at destroy
    bf16_t* zero_block = nullptr;  // 256 B of zeros (out-of-image halo source of the LDS-DMA conv)
    int det_sub_batch = 16, rec_sub_batch = 4096, post_group = 64;
    int tail_group = 16;   // pages per pass of the detector's 1/4-resolution tail (lateral in2 -> p2 -> head) inside one det forward
    std::map<std::string, Tensor4> taps;  // last forward's intermediates (debug / parity tests)
    int keep_taps = 0;   // 1: every intermediate (switches the fusions that would skip one off); 2: fusions stay on, tap what still exists
    bool fuse_head = true;  // head.convt3 fused into head.convt2's epilogue
    bool fuse_pool = true;  // stem.conv3's epilogue does the 3x3/s2 max pool
    int conv_big_min = 1024;  // 16x32-tile kernels once a full sub-batch gives at least this many work-groups
    bool blocked_layout = true;   // stage-0 activations between ring-kernel layers in channel-blocked layout (bit-identical)
    bool conv_ring = true;  // persistent ring kernel for the 3x3 / stride-1 layers (conv_ring.hip)
    int svtr_f16 = -1;      // storage type of the next SVTR load: -1 = what the blob's svtr.config says, 0 bf16, 1 fp16
    int conv2d_variant = 0; // lumina_ocr_conv2d: 0 = the layer's default kernel, 1 = LDS-DMA 16x32 tile, 2 = ring kernel (tests)
    int ring_orient = -1;   // its tile orientation: -1 auto, 0 / 1 forced (tests)
    bool fuse_short = true;   // stages 1-3: the block entry's 2x2 / stride-2 shortcut conv computed by its 3x3 / stride-2 conv0 kernel (one read of the input)
    bool fpn_compose = true;  // the lateral fpn.in2 composed into fpn.p2 (the 256-channel 1/4-resolution lateral is never computed); needs fpn_multi
    bool fpn_multi = true;  // head.conv1 reads p5 / p4 / p3 / p2 at their own resolution (ring kernel): the FPN concat is never written
    bool fuse_stem = true;  // stem.conv1 + stem.conv2 in one kernel (the first 32-channel tensor stays in LDS)
    bool fuse_mb = true;    // recogniser blocks: expand + depthwise in one kernel (the expanded tensor stays in LDS)
    // per-kernel event timing (bench roofline): accumulated conv-kernel time of the last det forward
    bool time_convs = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> conv_events;
    std::vector<double> conv_flops, conv_bytes;
    std::vector<std::string> conv_names, conv_kernels;
    // ---- pre-processing (resize / enhance) ----
    struct Coeffs { int ksize = 0; int* bounds = nullptr; int* kk = nullptr; std::shared_ptr<std::vector<int>> bounds_host; };
    std::map<std::pair<int, int>, Coeffs> coeff_cache;  // (in, out) -> device tables
    uint8_t* aux = nullptr; size_t aux_cap = 0;         // resize intermediate
    unsigned long long* sums = nullptr; int sums_cap = 0;
    int jd_last_passes = 0;   // synchronisation passes the last JPEG decode needed (incl. the one that found nothing to change)
    uint8_t* jd_stage[2] = {nullptr, nullptr}; size_t jd_stage_cap[2] = {0, 0}; hipEvent_t jd_stage_ev[2] = {nullptr, nullptr}; int jd_stage_next = 0;   // pinned staging buffers of the JPEG decoder (jpegdec.hip)
    float* dk_trig = nullptr; short* dk_wtab = nullptr;   // de-skew tables (deskew.h), uploaded at first use
};

int locr_fail(lumina_ocr* eng, const char* what, const char* detail);
bool parse_blob(lumina_ocr* eng, const void* blob, size_t n, std::map<std::string, HostBlobTensor>* out);

int eng_load_det(lumina_ocr* eng, const void* blob, size_t n);
int eng_load_rec(lumina_ocr* eng, const void* blob, size_t n);
int eng_det_forward(lumina_ocr* eng, const uint8_t* pages, int B, int H, int W, int Hp, int Wp, bf16_t* prob, hipStream_t st);
int eng_rec_forward(lumina_ocr* eng, const uint8_t* crops, const int* widths, int N, int* idx, float* prob, hipStream_t st);
int eng_load_svtr(lumina_ocr* eng, const void* blob, size_t n);
int eng_svtr_forward(lumina_ocr* eng, const uint8_t* crops, const int* widths, int N, int* idx, float* prob, hipStream_t st);
int eng_ws_reserve(lumina_ocr* eng, size_t bytes);
void* eng_ws_alloc(lumina_ocr* eng, size_t bytes);
// short_l / short_y: the block's shortcut layer and its output, computed by the same launch (ConvParams::wpk2)
int eng_run_conv(lumina_ocr* eng, const ConvLayer& L, const Tensor4& x, Tensor4* y, const Tensor4* res, int res_shift, int out_mode,
                 int up_shift, int y_cstride, int y_coff, bool flat, hipStream_t st, const bf16_t* gate = nullptr,
                 const ConvLayer* short_l = nullptr, Tensor4* short_y = nullptr);
