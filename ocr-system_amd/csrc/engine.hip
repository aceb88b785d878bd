// Engine: LOCW weight-blob parsing + packing, device workspace, det (DBNet-R18vd) and rec
// (CRNN-MobileNetV3) layer schedules.  Layer names/shapes mirror lumina_ocr/arch.py.
#include "engine.h"

#include <cstdlib>
#include <cstring>

#include "ops.h"
#include "stem_conv.h"

int locr_fail(lumina_ocr* eng, const char* what, const char* detail) {
    if (eng) eng->err = std::string(what) + ": " + (detail ? detail : "");
    return 1;
}

#define HIPCHK(expr)                                                                  \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) return locr_fail(eng, #expr, hipGetErrorString(_e));    \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is per device: remember (device, kernel) pairs, not kernels
#include <mutex>
#include <set>
hipError_t locr_dyn_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({dev, kernel})) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert({dev, kernel});
    return e;
}

// ------------------------------------------------------------------------------ blob
bool parse_blob(lumina_ocr* eng, const void* blob, size_t n, std::map<std::string, HostBlobTensor>* out) {
    const uint8_t* b = static_cast<const uint8_t*>(blob);
    if (n < 12 || memcmp(b, "LOCW", 4) != 0) { locr_fail(eng, "parse_blob", "bad magic"); return false; }
    uint32_t ver, cnt;
    memcpy(&ver, b + 4, 4); memcpy(&cnt, b + 8, 4);
    if (ver != 1) { locr_fail(eng, "parse_blob", "unsupported version"); return false; }
    size_t off = 12;
    for (uint32_t i = 0; i < cnt; ++i) {
        if (off + 2 > n) { locr_fail(eng, "parse_blob", "truncated"); return false; }
        uint16_t ln; memcpy(&ln, b + off, 2); off += 2;
        if (off + ln + 2 > n) { locr_fail(eng, "parse_blob", "truncated"); return false; }
        std::string name(reinterpret_cast<const char*>(b + off), ln); off += ln;
        HostBlobTensor t;
        t.dtype = b[off]; const int nd = b[off + 1]; off += 2;
        if (t.dtype > 1 || nd < 1 || nd > 4) { locr_fail(eng, "parse_blob: unknown dtype / rank", name.c_str()); return false; }
        if (off + 4 * (size_t)nd + 8 > n) { locr_fail(eng, "parse_blob", "truncated"); return false; }
        uint64_t count = 1;
        for (int d = 0; d < nd; ++d) {
            uint32_t v; memcpy(&v, b + off, 4); off += 4;
            if (v == 0 || v > (1u << 28) || count * v > (1ull << 34)) { locr_fail(eng, "parse_blob: implausible dimension", name.c_str()); return false; }
            count *= v; t.dims.push_back((int)v);
        }
        uint64_t nb; memcpy(&nb, b + off, 8); off += 8;
        off += (16 - off % 16) % 16;
        if (off > n || nb > n - off) { locr_fail(eng, "parse_blob", "truncated data"); return false; }   // (off + nb could wrap)
        if (nb != count * (t.dtype ? 2u : 4u)) { locr_fail(eng, "parse_blob: byte count does not match the dimensions", name.c_str()); return false; }
        t.data = b + off; t.nbytes = (size_t)nb; off += nb;
        (*out)[name] = t;
    }
    return true;
}

// ------------------------------------------------------------------------------ memory
static void* dev_upload(lumina_ocr* eng, const void* host, size_t bytes) {
    void* d = nullptr;
    if (hipMalloc(&d, bytes ? bytes : 16) != hipSuccess) return nullptr;
    if (bytes && hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    eng->owned.push_back(d);
    return d;
}

int eng_ws_reserve(lumina_ocr* eng, size_t bytes) {
    if (bytes <= eng->ws_cap) return 0;
    HIPCHK(hipDeviceSynchronize());
    if (eng->ws) HIPCHK(hipFree(eng->ws));
    eng->ws = nullptr; eng->ws_cap = 0;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&eng->ws), bytes));
    eng->ws_cap = bytes;
    return 0;
}

void* eng_ws_alloc(lumina_ocr* eng, size_t bytes) {
    const size_t a = (eng->ws_off + 255) & ~(size_t)255;
    eng->ws_off = a + bytes;
    if (eng->ws == nullptr || eng->ws_off > eng->ws_cap) return nullptr;  // dry run or overflow
    return eng->ws + a;
}

static Tensor4 ws_tensor(lumina_ocr* eng, int n, int h, int w, int c) {
    Tensor4 t; t.n = n; t.h = h; t.w = w; t.c = c;
    t.p = static_cast<bf16_t*>(eng_ws_alloc(eng, t.elems() * sizeof(bf16_t)));
    return t;
}

static inline int rup(int v, int m) { return (v + m - 1) / m * m; }

// ------------------------------------------------------------------------------ layer construction
static bool get_wb(lumina_ocr* eng, const std::map<std::string, HostBlobTensor>& m, const std::string& name,
                   const HostBlobTensor** w, const HostBlobTensor** b) {
    auto iw = m.find(name + ".w"), ib = m.find(name + ".b");
    if (iw == m.end() || ib == m.end() || iw->second.dtype != 1 || ib->second.dtype != 0) {
        locr_fail(eng, "missing/ill-typed tensor", name.c_str());
        return false;
    }
    *w = &iw->second; *b = &ib->second;
    return true;
}

// Build a conv layer from blob tensor name.w [cout_r][ks][ks][cin_r] (bf16) / name.b [cout_r or bias_r] (f32).
static bool make_conv(lumina_ocr* eng, const std::map<std::string, HostBlobTensor>& m, const std::string& name, int ks, int stride,
                      int cin_r, int cout_r, int cin_p, int cout_p, int act, ConvLayer* L, int bias_repeat = 1) {
    const HostBlobTensor *w, *b;
    if (!get_wb(eng, m, name, &w, &b)) return false;
    if (w->dims.size() != 4 || w->dims[0] != cout_r || w->dims[1] != ks || w->dims[3] != cin_r ||
        (int)b->dims[0] * bias_repeat != cout_r) {
        locr_fail(eng, "unexpected tensor shape", name.c_str());
        return false;
    }
    L->name = name; L->ks = ks; L->stride = stride; L->cin = cin_p; L->cout = cout_p; L->act = act;
    if (!conv_pick_cfg(ks, stride, cin_p, cout_p, &L->cfg)) { locr_fail(eng, "no conv kernel config", name.c_str()); return false; }
    const int taps = ks * ks;
    std::vector<bf16_t> padded((size_t)cout_p * taps * cin_p, 0);
    const bf16_t* src = reinterpret_cast<const bf16_t*>(w->data);
    for (int co = 0; co < cout_r; ++co)
        for (int t = 0; t < taps; ++t)
            memcpy(&padded[((size_t)co * taps + t) * cin_p], &src[((size_t)co * taps + t) * cin_r], sizeof(bf16_t) * cin_r);
    std::vector<bf16_t> packed(conv_packed_weight_elems(cout_p, ks, cin_p, L->cfg.bn));
    pack_conv_weights(padded.data(), cout_p, ks, cin_p, L->cfg.bn, L->cfg.ck, packed.data());
    L->wpk = static_cast<bf16_t*>(dev_upload(eng, packed.data(), packed.size() * sizeof(bf16_t)));
    // 16x32-tile kernel: operands by LDS-DMA into a 2-deep ring (nw == 6, default: +3..5 % on the 128..256-channel layers) or
    // through registers (nw == 5, LUMINA_CONV_DMA=0)
    static const int dma = getenv("LUMINA_CONV_DMA") != nullptr ? atoi(getenv("LUMINA_CONV_DMA")) : 1;
    if (ks == 3 && stride == 1 && L->cfg.bn == 64 && (cin_p >= 64 || name == "stem.conv3") && L->cfg.nw == 4) {  // stem.conv3: for the fused max pool
        L->cfg_big = L->cfg; L->cfg_big.nw = dma ? 6 : 5; L->cfg_big.ck = 16;
        std::vector<bf16_t> packed2(conv_packed_weight_elems(cout_p, ks, cin_p, 64));
        pack_conv_weights(padded.data(), cout_p, ks, cin_p, 64, 16, packed2.data(), dma ? 1 : 0);
        L->wpk_big = static_cast<bf16_t*>(dev_upload(eng, packed2.data(), packed2.size() * sizeof(bf16_t)));
    }
    const int ntiles = (cout_p + L->cfg.bn - 1) / L->cfg.bn;
    std::vector<float> bias((size_t)ntiles * L->cfg.bn, 0.f);
    const float* bs = reinterpret_cast<const float*>(b->data);
    const int bn_r = b->dims[0];
    for (int i = 0; i < cout_r; ++i) bias[i] = bs[i % bn_r];
    L->bias = static_cast<float*>(dev_upload(eng, bias.data(), bias.size() * sizeof(float)));
    if (!L->wpk || !L->bias) { locr_fail(eng, "device upload failed", name.c_str()); return false; }
    return true;
}

static inline float host_bf16_to_f32(bf16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static inline bf16_t host_f32_to_bf16(float f) {   // round to nearest even (finite inputs)
    uint32_t u; memcpy(&u, &f, 4);
    return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// The lateral fpn.in2 (1x1, 64 -> 256, no bias in DBFPN) composed into the smoothing conv fpn.p2 (3x3, 256 -> 64):
//     p2 = conv3x3(W_p2, W_in2 c2 + up2(out3)) = conv3x3(W_c, c2) + conv3x3(W_p2, up2(out3)),   W_c[co][tap][ci] = sum_m W_p2[co][tap][m] W_in2[m][ci]
// -> layer "fpn.p2c": ONE 3x3 conv over the 320 channels [c2 (64, full resolution) | out3 (256, half resolution, read nearest-upsampled)];
// the 256-channel 1/4-resolution lateral (1.5 GB per 16 A4 pages, written by one kernel and read back by the next) never exists.
// W_c is formed in fp64 in m order, rounded to fp32 and then to bf16 (oracle/nets.py compose_fpn_p2 does the same, bit for bit); the
// lateral sum is no longer rounded to bf16 on its way.  Exact with zero padding only for a lateral without bias — which is what
// DBFPN has; a blob with a non-zero fpn.in2 bias keeps the two-kernel path.
static int compose_fpn_p2(lumina_ocr* eng, const std::map<std::string, HostBlobTensor>& m) {
    const HostBlobTensor *wi, *bi, *wp, *bp;
    if (!get_wb(eng, m, "fpn.in2", &wi, &bi) || !get_wb(eng, m, "fpn.p2", &wp, &bp)) return 1;
    const float* b_in2 = reinterpret_cast<const float*>(bi->data);
    for (int i = 0; i < 256; ++i)
        if (b_in2[i] != 0.f) return 0;   // (no "fpn.p2c" entry: det_forward_sub keeps in2 -> p2)
    const bf16_t* Wi = reinterpret_cast<const bf16_t*>(wi->data);   // [256][1][1][64]
    const bf16_t* Wp = reinterpret_cast<const bf16_t*>(wp->data);   // [64][3][3][256]
    std::vector<bf16_t> w((size_t)64 * 9 * 320);
    std::vector<double> win((size_t)256 * 64);
    for (size_t i = 0; i < win.size(); ++i) win[i] = (double)host_bf16_to_f32(Wi[i]);
    for (int co = 0; co < 64; ++co)
        for (int t = 0; t < 9; ++t) {
            double acc[64] = {0};
            const bf16_t* row = Wp + ((size_t)co * 9 + t) * 256;
            for (int mm = 0; mm < 256; ++mm) {
                const double a = (double)host_bf16_to_f32(row[mm]);
                for (int ci = 0; ci < 64; ++ci) acc[ci] += a * win[(size_t)mm * 64 + ci];
            }
            bf16_t* dst = &w[((size_t)co * 9 + t) * 320];
            for (int ci = 0; ci < 64; ++ci) dst[ci] = host_f32_to_bf16((float)acc[ci]);
            memcpy(dst + 64, row, 256 * sizeof(bf16_t));
        }
    std::map<std::string, HostBlobTensor> mm;
    mm["fpn.p2c.w"] = HostBlobTensor{1, {64, 3, 3, 320}, reinterpret_cast<const uint8_t*>(w.data()), w.size() * sizeof(bf16_t)};
    mm["fpn.p2c.b"] = *bp;
    ConvLayer& L = eng->det["fpn.p2c"];
    if (!make_conv(eng, mm, "fpn.p2c", 3, 1, 320, 64, 320, 64, ACT_NONE, &L)) return 1;
    L.alg_flop_per_px = 2.0 * 64 * 256 + 2.0 * 9 * 256 * 64;   // the architecture's fpn.in2 + fpn.p2 (the composed conv executes 2 * 9 * 320 * 64)
    return 0;
}

int eng_load_det(lumina_ocr* eng, const void* blob, size_t n) {
    std::map<std::string, HostBlobTensor> m;
    if (!parse_blob(eng, blob, n, &m)) return 1;
    HIPCHK(hipSetDevice(eng->device));
    eng->det.clear();
    // stem.conv1 (Cin = 3): dedicated kernel
    {
        const HostBlobTensor *w, *b;
        if (!get_wb(eng, m, "stem.conv1", &w, &b)) return 1;
        if (w->dims.size() != 4 || w->dims[0] != 32 || w->dims[1] != 3 || w->dims[3] != 3) return locr_fail(eng, "stem.conv1", "shape");
        bf16_t packed[2 * 2 * 32 * 8];
        pack_stem_weights(reinterpret_cast<const bf16_t*>(w->data), 32, packed);
        eng->stem_wpk = static_cast<bf16_t*>(dev_upload(eng, packed, sizeof(packed)));
        eng->stem_bias = static_cast<float*>(dev_upload(eng, b->data, 32 * sizeof(float)));
    }
    auto add = [&](const std::string& name, int ks, int stride, int cin, int cout, int act) -> bool {
        return make_conv(eng, m, name, ks, stride, cin, cout, cin, cout, act, &eng->det[name]);
    };
    if (!add("stem.conv2", 3, 1, 32, 32, ACT_RELU) || !add("stem.conv3", 3, 1, 32, 64, ACT_RELU)) return 1;
    const int chs[4] = {64, 128, 256, 512};
    int cin = 64;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 2; ++j) {
            const std::string p = "s" + std::to_string(i) + ".b" + std::to_string(j);
            const int stride = (i > 0 && j == 0) ? 2 : 1;
            if (!add(p + ".conv0", 3, stride, cin, chs[i], ACT_RELU)) return 1;
            if (!add(p + ".conv1", 3, 1, chs[i], chs[i], ACT_RELU)) return 1;
            if (j == 0) {
                if (i == 0) { if (!add(p + ".short", 1, 1, cin, chs[i], ACT_NONE)) return 1; }
                else { if (!add(p + ".short", 2, 2, cin, chs[i], ACT_NONE)) return 1; }
            }
            cin = chs[i];
        }
    }
    for (int l = 5; l >= 2; --l) {
        if (!add("fpn.in" + std::to_string(l), 1, 1, chs[l - 2], 256, ACT_NONE)) return 1;
        if (!add("fpn.p" + std::to_string(l), 3, 1, 256, 64, ACT_NONE)) return 1;
    }
    if (!add("head.conv1", 3, 1, 256, 64, ACT_RELU)) return 1;
    if (compose_fpn_p2(eng, m)) return 1;
    {
        ConvLayer& L = eng->det["head.convt2"];
        if (!make_conv(eng, m, "head.convt2", 1, 1, 64, 256, 64, 256, ACT_RELU, &L, 4)) return 1;
        L.convt = true; L.convt_c = 64;
        ConvLayer& L3 = eng->det["head.convt3"];
        if (!make_conv(eng, m, "head.convt3", 1, 1, 64, 4, 64, 4, ACT_SIGMOID, &L3, 4)) return 1;
        L3.convt = true; L3.convt_c = 1;
        // fused tail: raw convt3 weights [4][64] + scalar bias live next to convt2
        const HostBlobTensor *w3, *b3;
        if (!get_wb(eng, m, "head.convt3", &w3, &b3)) return 1;
        ConvLayer& Lf = eng->det["head.convt2.fused"];
        Lf = L;
        Lf.name = "head.convt2+3";
        {   // MFMA A-operand image [kstep 4][half 2][row 32][8]: rows 0..3 = convt3 weights, the rest zero
            const bf16_t* w3s = reinterpret_cast<const bf16_t*>(w3->data);
            std::vector<bf16_t> img(4 * 2 * 32 * 8, 0);
            for (int ks = 0; ks < 4; ++ks)
                for (int hh = 0; hh < 2; ++hh)
                    for (int row = 0; row < 4; ++row)
                        for (int j = 0; j < 8; ++j) img[((ks * 2 + hh) * 32 + row) * 8 + j] = w3s[row * 64 + ks * 16 + hh * 8 + j];
            Lf.fuse_w = static_cast<bf16_t*>(dev_upload(eng, img.data(), img.size() * sizeof(bf16_t)));
        }
        Lf.fuse_b = reinterpret_cast<const float*>(b3->data)[0];
        if (!Lf.fuse_w) return locr_fail(eng, "upload", "head.convt3 fused weights");
    }
    eng->det_loaded = true;
    return 0;
}

// ------------------------------------------------------------------------------ conv dispatch
// does a 3x3 / stride-1 layer on an ho x wo map take the 16x32-tile (LDS-DMA / ring) kernels?  Depends on the layer geometry and the
// configured sub-batch only, never on the number of images in the call (the variants sum in different orders)
static bool conv_takes_big(const lumina_ocr* eng, const ConvLayer& L, int n, int ho, int wo, bool flat) {
    static const bool no_big = getenv("LUMINA_CONV_NO_BIG") != nullptr;
    static const long long big_min_env = getenv("LUMINA_CONV_BIG_MIN") ? atoll(getenv("LUMINA_CONV_BIG_MIN")) : -1;
    const long long nb_eff = n > eng->det_sub_batch ? n : eng->det_sub_batch;
    const long long big_blocks = nb_eff * ((ho + 15) / 16) * ((wo + 31) / 32) * ((L.cout + L.cfg.bn - 1) / L.cfg.bn);
    const long long big_min = big_min_env >= 0 ? big_min_env : eng->conv_big_min;
    return (L.force_big && L.wpk_big != nullptr) || (!no_big && !flat && !L.small_only && L.wpk_big != nullptr && big_blocks >= big_min);
}

int eng_run_conv(lumina_ocr* eng, const ConvLayer& L, const Tensor4& x, Tensor4* y, const Tensor4* res, int res_shift, int out_mode,
                 int up_shift, int y_cstride, int y_coff, bool flat, hipStream_t st, const bf16_t* gate, const ConvLayer* short_l, Tensor4* short_y) {
    if (short_l != nullptr && L.launch_group > 0) return locr_fail(eng, "fused shortcut", "not available with launch groups");
    if (L.launch_group > 0 && x.n > L.launch_group && !flat && gate == nullptr && x.p != nullptr && y->p != nullptr &&
        (out_mode == OUT_NORMAL) && !x.blk && !y->blk && (!res || !res->blk)) {
        // the same layer in launches of launch_group images (slices of the NHWC tensors; results are per image, hence identical)
        ConvLayer one = L;
        one.launch_group = 0;
        const size_t ystride = (size_t)y->h * y->w * (y_cstride ? y_cstride : y->c);
        for (int b0 = 0; b0 < x.n; b0 += L.launch_group) {
            const int nb = x.n - b0 < L.launch_group ? x.n - b0 : L.launch_group;
            Tensor4 xs = x, ys = *y, rs;
            xs.n = nb; xs.p = x.p + (size_t)b0 * x.h * x.w * (x.n_src > 1 ? x.src_cs(x.n_src - 1) : x.c);   // (multi-source: p is the last source)
            for (int k = 0; k < x.n_src; ++k) xs.xs[k] = x.xs[k] + (size_t)b0 * (x.h >> x.xs_shift[k]) * (x.w >> x.xs_shift[k]) * x.src_cs(k);
            ys.n = nb; ys.p = y->p + (size_t)b0 * ystride;
            if (res) { rs = *res; rs.n = nb; rs.p = res->p + (size_t)b0 * res->h * res->w * res->c; }
            if (eng_run_conv(eng, one, xs, &ys, res ? &rs : nullptr, res_shift, out_mode, up_shift, y_cstride, y_coff, flat, st, gate)) return 1;
        }
        return 0;
    }
    ConvParams p{};
    if (eng->zero_block == nullptr) {
        const uint32_t z[64] = {0};
        eng->zero_block = static_cast<bf16_t*>(dev_upload(eng, z, sizeof(z)));
        if (!eng->zero_block) return locr_fail(eng, "conv", "zero block upload failed");
    }
    p.zeros = eng->zero_block;
    p.gate = gate; p.gate_hw = x.h * x.w;
    p.x = x.p; p.wpk = L.wpk; p.bias = L.bias; p.res = res ? res->p : nullptr; p.y = y->p;
    p.Cin = L.cin; p.Cout = L.cout; p.act = L.act;
    p.out_mode = out_mode; p.up_shift = up_shift; p.convt_c = L.convt_c;
    p.fuse_w = (out_mode == OUT_CONVT) ? L.fuse_w : nullptr; p.fuse_b = L.fuse_b;
    if (flat) {  // 1x1 conv == GEMM over all pixels: re-tile as rows of 32 so every 8x32 tile is full
        const long long m = (long long)x.n * x.h * x.w;
        p.N = 1; p.W = 32; p.H = (int)((m + 31) / 32); p.pix_limit = (int)m;
        p.Ho = p.H; p.Wo = 32;
        p.res_h = p.H; p.res_w = 32;
    } else {
        p.N = x.n; p.H = x.h; p.W = x.w; p.pix_limit = 0;
        p.Ho = (L.ks == 3) ? (x.h - 1) / L.stride + 1 : x.h / L.stride;
        p.Wo = (L.ks == 3) ? (x.w - 1) / L.stride + 1 : x.w / L.stride;
        if (res) { p.res_h = res->h; p.res_w = res->w; }
    }
    p.res_shift = res_shift;
    p.x_blk = x.blk; p.y_blk = y->blk; p.res_blk = res ? res->blk : 0;
    p.n_src = x.n_src;
    for (int k = 0; k < 4; ++k) {
        p.xs[k] = x.xs[k]; p.xs_shift[k] = x.xs_shift[k];
        p.xs_nchunk[k] = k < x.n_src ? x.src_c(k) / 16 : 0; p.xs_cstride[k] = k < x.n_src ? x.src_cs(k) : 0;
    }
    p.res_cstride = res ? res->c : 0;
    p.y_cstride = y_cstride ? y_cstride : y->c;
    p.y_coff = y_coff;
    if (short_l != nullptr) {
        if (short_y == nullptr || short_l->cin != L.cin || short_l->cout != L.cout || short_l->cfg.bn != L.cfg.bn || short_l->cfg.ck != L.cfg.ck ||
            short_l->ks != 2 || short_l->stride != 2 || short_y->c != L.cout || short_y->h != y->h || short_y->w != y->w)
            return locr_fail(eng, "fused shortcut: layer mismatch", L.name.c_str());
        p.wpk2 = short_l->wpk; p.bias2 = short_l->bias; p.y2 = short_y->p; p.y2_cstride = short_y->c;
    }
    if (x.c != L.cin) return locr_fail(eng, "conv input channels mismatch", L.name.c_str());
    if (x.p == nullptr || y->p == nullptr) return 0;  // dry run (workspace sizing)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (eng->time_convs) {
        HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
        HIPCHK(hipEventRecord(e0, st));
    }
    if (out_mode == OUT_POOL && (L.wpk_big == nullptr || L.cfg_big.nw != 6)) return locr_fail(eng, "fused max pool needs the LDS-DMA conv kernel", L.name.c_str());
    // 16x32 tiles (less LDS and L2 traffic per MFMA) once they still give >= 2 workgroups per CU on all 256 CUs twice over
    const bool use_big = out_mode == OUT_POOL || p.n_src > 1 || conv_takes_big(eng, L, p.N, p.Ho, p.Wo, flat);   // (a multi-source input is the ring kernel's)
    if (L.force_big && eng->conv2d_variant == 2 && !conv_ring_supported(L.cfg_big, p)) return locr_fail(eng, "conv2d_variant 2: the ring kernel does not take this layer", L.name.c_str());
    ConvKernelCfg cfg = L.cfg;
    if (use_big) { cfg = L.cfg_big; p.wpk = L.wpk_big; }
    static const bool no_pw = getenv("LUMINA_CONV_NO_PW") != nullptr;
    const bool use_pw = !no_pw && !use_big && conv_pw_supported(cfg, p);
    const bool use_ring = eng->conv_ring && use_big && !use_pw && conv_ring_supported(cfg, p);
    if (p.n_src > 1 && !use_ring) return locr_fail(eng, "a multi-source input reached a kernel other than the ring kernel", L.name.c_str());
    if (!use_ring && (p.x_blk || p.y_blk || p.res_blk)) return locr_fail(eng, "a channel-blocked tensor reached a kernel that cannot address it", L.name.c_str());
    hipError_t e = use_ring ? conv_ring_launch(p, eng->ring_orient, st) : (use_pw ? conv_pw_launch(p, st) : conv_launch(cfg, p, st));
    if (e != hipSuccess) return locr_fail(eng, L.name.c_str(), hipGetErrorString(e));
    if (eng->time_convs) {
        HIPCHK(hipEventRecord(e1, st));
        eng->conv_events.push_back({e0, e1});
        const double px = flat ? (double)p.pix_limit : (double)p.N * p.Ho * p.Wo;
        eng->conv_flops.push_back((L.alg_flop_per_px > 0 ? px * L.alg_flop_per_px : 2.0 * px * L.ks * L.ks * L.cin * L.cout) +
                                  (short_l ? 2.0 * px * 4 * short_l->cin * short_l->cout : 0.0));
        // algorithmic HBM bytes: input once + output once (+ residual) + weights once
        double in_elems = (double)x.elems();
        if (x.n_src > 1) { in_elems = 0; for (int k = 0; k < x.n_src; ++k) in_elems += (double)x.n * (x.h >> x.xs_shift[k]) * (x.w >> x.xs_shift[k]) * x.src_c(k); }
        eng->conv_bytes.push_back(2.0 * (in_elems + px * (out_mode == OUT_CONVT && p.fuse_w ? 4.0 : (double)L.cout) * (out_mode == OUT_UPSAMPLE ? (double)(1 << (2 * up_shift)) : (out_mode == OUT_POOL ? 0.25 : 1.0)) + (res ? px * (double)L.cout / (double)(1 << (2 * res_shift)) : 0.0) + (double)L.ks * L.ks * L.cin * L.cout +
                                         (short_l ? px * (double)short_l->cout + 4.0 * short_l->cin * short_l->cout : 0.0)));
        eng->conv_names.push_back(short_l ? L.name + "+short" : L.name);
        std::string kname = use_pw ? (L.cin == 64 ? "conv_pw_kernel<64>" : "conv_pw_kernel<128>") : conv_kernel_name(cfg);
        if (use_ring) kname = conv_ring_kernel_name(p, eng->ring_orient);
        if (short_l) { const size_t pos = kname.rfind(",0,2>"); if (pos != std::string::npos) kname.replace(pos, 5, ",5,2>"); }   // the fused-shortcut instantiation
        if (out_mode == OUT_POOL && !use_ring) { const size_t pos = kname.rfind(",3,4>"); if (pos != std::string::npos) kname.replace(pos, 5, ",4,4>"); }  // the fused-pool instantiation
        eng->conv_kernels.push_back(kname);
    }
    return 0;
}

#define RUN(expr) do { if ((expr) != 0) return 1; } while (0)

// option time_convs: HIP events on the launch stream around ONE launch of the recogniser paths too (bench.py --config 3 / 5);
// flop / bytes are the launch's algorithmic work (operands and results once)
struct LaunchTimer {
    lumina_ocr* eng; hipStream_t st; hipEvent_t e0 = nullptr, e1 = nullptr; bool on;
    LaunchTimer(lumina_ocr* e, hipStream_t s, bool enabled) : eng(e), st(s), on(enabled && e->time_convs) {
        if (on) { on = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess && hipEventRecord(e0, st) == hipSuccess; }
    }
    void done(const std::string& layer, const std::string& kernel, double flop, double bytes) {
        if (!on || hipEventRecord(e1, st) != hipSuccess) return;
        eng->conv_events.push_back({e0, e1});
        eng->conv_flops.push_back(flop); eng->conv_bytes.push_back(bytes);
        eng->conv_names.push_back(layer); eng->conv_kernels.push_back(kernel);
    }
};

static void tap(lumina_ocr* eng, const char* name, const Tensor4& t) { if (eng->keep_taps && t.p) eng->taps[name] = t; }

// ------------------------------------------------------------------------------ det forward
static int det_forward_sub(lumina_ocr* eng, const uint8_t* pages, int B, int H, int W, int Hp, int Wp, bf16_t* prob, hipStream_t st) {
    eng->ws_off = 0;
    const bool dry = (eng->ws == nullptr) || pages == nullptr;
    auto& D = eng->det;
    // stem.conv1 + stem.conv2 fused (the first half-resolution tensor stays in LDS) unless it is wanted as a tap
    ConvLayer& c2 = D["stem.conv2"];
    const bool fuse_stem = eng->fuse_stem && eng->keep_taps != 1 && c2.cfg.bn == 32 && c2.cfg.ck == 16 && c2.act == ACT_RELU;
    Tensor4 t1{};
    if (!fuse_stem || dry) t1 = ws_tensor(eng, B, Hp / 2, Wp / 2, 32);
    Tensor4 t2 = ws_tensor(eng, B, Hp / 2, Wp / 2, 32);
    if (!dry && t2.p && (fuse_stem || t1.p)) {
        StemParams sp{};
        sp.x = pages; sp.wpk = eng->stem_wpk; sp.bias = eng->stem_bias; sp.y = fuse_stem ? t2.p : t1.p; sp.valid_w_per_img = nullptr;
        sp.N = B; sp.H = H; sp.W = W; sp.valid_h = H; sp.valid_w = W; sp.Ho = Hp / 2; sp.Wo = Wp / 2; sp.Cout_store = 32;
        sp.act = ACT_RELU;
        const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
        for (int c = 0; c < 3; ++c) { sp.scale[c] = 1.0f / (255.0f * stdv[c]); sp.shift[c] = -mean[c] / stdv[c]; }
        hipError_t e = fuse_stem ? stem12_launch(sp, c2.wpk, c2.bias, st) : stem_conv_launch(sp, st);
        if (e != hipSuccess) return locr_fail(eng, "stem.conv1", hipGetErrorString(e));
    }
    if (!fuse_stem) {
        tap(eng, "stem.conv1", t1);
        RUN(eng_run_conv(eng, c2, t1, &t2, nullptr, 0, OUT_NORMAL, 0, 0, 0, false, st)); tap(eng, "stem.conv2", t2);
    }
    // stem.conv3 + 3x3/s2 max pool: fused (the 64-channel half-resolution tensor, 93 MB per page, is never written) unless the
    // intermediate is wanted as a tap
    const bool fuse_pool = eng->fuse_pool && eng->keep_taps != 1;
    Tensor4 t3{};
    if (!fuse_pool || dry) t3 = ws_tensor(eng, B, Hp / 2, Wp / 2, 64);
    Tensor4 x = ws_tensor(eng, B, Hp / 4, Wp / 4, 64);
    if (fuse_pool) {
        RUN(eng_run_conv(eng, D["stem.conv3"], t2, &x, nullptr, 0, OUT_POOL, 0, 0, 0, false, st));
    } else {
        RUN(eng_run_conv(eng, D["stem.conv3"], t2, &t3, nullptr, 0, OUT_NORMAL, 0, 0, 0, false, st)); tap(eng, "stem.conv3", t3);
        if (!dry && x.p) {
            hipError_t e = maxpool_launch(t3.p, x.p, B, t3.h, t3.w, 64, 3, 2, 1, x.h, x.w, st);
            if (e != hipSuccess) return locr_fail(eng, "stem.pool", hipGetErrorString(e));
        }
    }
    tap(eng, "stem.pool", x);
    const int chs[4] = {64, 128, 256, 512};
    Tensor4 feats[4];
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 2; ++j) {
            const std::string p = "s" + std::to_string(i) + ".b" + std::to_string(j);
            const int stride = (i > 0 && j == 0) ? 2 : 1;
            Tensor4 y = ws_tensor(eng, B, x.h / stride, x.w / stride, chs[i]);
            // stage-0 tensors that only the ring kernel touches are stored channel-blocked [n][C/16][H][W][16]: a 16-channel chunk
            // pass then reads / writes whole cache lines (NHWC: 32 of every 128 bytes per pass); same values, DESIGN.md 3.2
            const bool blk = i == 0 && eng->blocked_layout && eng->keep_taps == 0 && eng->conv_ring && !dry &&
                             conv_takes_big(eng, D[p + ".conv0"], B, x.h, x.w, false) && conv_takes_big(eng, D[p + ".conv1"], B, x.h, x.w, false);
            y.blk = blk;
            // stages 1-3, first block: conv0 (3x3 / s2) and the vd shortcut (2x2 / s2) read the same input — one launch computes both
            const ConvLayer& c0 = D[p + ".conv0"];
            const bool fuse_sc = eng->fuse_short && i > 0 && j == 0 && x.h % 2 == 0 && x.w % 2 == 0 && c0.cfg.bn == 64 && c0.cfg.ck == 16 && c0.cfg.nw == 4 &&
                                 D[p + ".short"].cfg.bn == 64 && D[p + ".short"].cfg.ck == 16;
            Tensor4 sc = x;
            if (j == 0) sc = ws_tensor(eng, B, y.h, y.w, chs[i]);
            if (fuse_sc) {
                RUN(eng_run_conv(eng, c0, x, &y, nullptr, 0, OUT_NORMAL, 0, 0, 0, false, st, nullptr, &D[p + ".short"], &sc));
            } else {
                RUN(eng_run_conv(eng, c0, x, &y, nullptr, 0, OUT_NORMAL, 0, 0, 0, false, st));
                if (j == 0) RUN(eng_run_conv(eng, D[p + ".short"], x, &sc, nullptr, 0, OUT_NORMAL, 0, 0, 0, i == 0, st));
            }
            Tensor4 o = ws_tensor(eng, B, y.h, y.w, chs[i]);
            o.blk = blk && j == 0;
            RUN(eng_run_conv(eng, D[p + ".conv1"], y, &o, &sc, 0, OUT_NORMAL, 0, 0, 0, false, st));
            tap(eng, p.c_str(), o);
            x = o;
        }
        feats[i] = x;
    }
    // FPN: laterals with fused top-down nearest-upsample add
    Tensor4 in5 = ws_tensor(eng, B, feats[3].h, feats[3].w, 256);
    RUN(eng_run_conv(eng, D["fpn.in5"], feats[3], &in5, nullptr, 0, OUT_NORMAL, 0, 0, 0, false, st));
    Tensor4 out4 = ws_tensor(eng, B, feats[2].h, feats[2].w, 256);
    RUN(eng_run_conv(eng, D["fpn.in4"], feats[2], &out4, &in5, 1, OUT_NORMAL, 0, 0, 0, false, st));
    Tensor4 out3 = ws_tensor(eng, B, feats[1].h, feats[1].w, 256);
    RUN(eng_run_conv(eng, D["fpn.in3"], feats[1], &out3, &out4, 1, OUT_NORMAL, 0, 0, 0, false, st));
    // DBHead's first conv runs over the concat [up8(p5), up4(p4), up2(p3), p2].  Default: the smoothing convs write p5 .. p2 at their
    // own resolution and head.conv1 (ring kernel) reads them nearest-upsampled through its halo addressing — the 1/4-resolution
    // 256-channel concat (1.5 GB per 16 A4 pages, written 4 / 16 / 64-fold replicated) never exists.  Same products, same order.
    ConvLayer& hc1 = D["head.conv1"];
    const bool multi = eng->fpn_multi && eng->keep_taps != 1 && eng->conv_ring && conv_takes_big(eng, hc1, B, Hp / 4, Wp / 4, false) &&
                       (Hp / 4) % 8 == 0 && (Wp / 4) % 8 == 0;
    auto slice = [](const Tensor4& t, int b0, int nb) {
        Tensor4 r = t;
        r.n = nb;
        if (t.p) r.p = t.p + (size_t)b0 * t.h * t.w * t.c;
        return r;
    };
    // fpn.p2.  Default: ONE conv over [c2 | up2(out3)] with the lateral fpn.in2 composed into its weights (compose_fpn_p2; ring kernel,
    // multi-source input, whatever the page size: the arithmetic definition must not depend on which kernel a geometry selects) —
    // the 256-channel 1/4-resolution lateral is never computed.  Otherwise (option fpn_compose = 0, ring kernel off, a lateral with
    // a bias): in2 -> lateral tensor -> p2.
    const bool compose = eng->fpn_compose && eng->conv_ring && D.count("fpn.p2c") != 0;
    auto run_p2 = [&](const Tensor4& c2, const Tensor4& o3, Tensor4* lateral, Tensor4* dst, int y_cstride, int y_coff) -> int {
        if (compose) {
            Tensor4 in = c2;
            in.c = 320; in.n_src = 2;
            in.xs[0] = c2.p; in.xs[1] = o3.p; in.p = o3.p;
            in.xs_shift[0] = 0; in.xs_shift[1] = 1; in.xs_c[0] = 64; in.xs_c[1] = 256;
            return eng_run_conv(eng, D["fpn.p2c"], in, dst, nullptr, 0, OUT_NORMAL, 0, y_cstride, y_coff, false, st);
        }
        RUN(eng_run_conv(eng, D["fpn.in2"], c2, lateral, &o3, 1, OUT_NORMAL, 0, 0, 0, false, st));
        return eng_run_conv(eng, D["fpn.p2"], *lateral, dst, nullptr, 0, OUT_NORMAL, 0, y_cstride, y_coff, false, st);
    };
    auto head_tail = [&](const Tensor4& h1, bf16_t* prob_g, int nb) -> int {
        Tensor4 pm; pm.p = dry ? nullptr : prob_g; pm.n = nb; pm.h = Hp; pm.w = Wp; pm.c = 1;
        if (eng->fuse_head && eng->keep_taps != 1)   // DBHead tail in one launch: the 64-channel 1/2-resolution tensor never exists
            return eng_run_conv(eng, D["head.convt2.fused"], h1, &pm, nullptr, 0, OUT_CONVT, 0, 1, 0, false, st);
        Tensor4 h2 = ws_tensor(eng, nb, Hp / 2, Wp / 2, 64);
        RUN(eng_run_conv(eng, D["head.convt2"], h1, &h2, nullptr, 0, OUT_CONVT, 0, 0, 0, false, st)); tap(eng, "head.convt2", h2);
        return eng_run_conv(eng, D["head.convt3"], h2, &pm, nullptr, 0, OUT_CONVT1, 0, 1, 0, false, st);
    };
    if (multi) {
        Tensor4 p5 = ws_tensor(eng, B, in5.h, in5.w, 64), p4 = ws_tensor(eng, B, out4.h, out4.w, 64), p3 = ws_tensor(eng, B, out3.h, out3.w, 64);
        RUN(eng_run_conv(eng, D["fpn.p5"], in5, &p5, nullptr, 0, OUT_NORMAL, 0, 0, 0, false, st)); tap(eng, "fpn.p5", p5);
        RUN(eng_run_conv(eng, D["fpn.p4"], out4, &p4, nullptr, 0, OUT_NORMAL, 0, 0, 0, false, st)); tap(eng, "fpn.p4", p4);
        RUN(eng_run_conv(eng, D["fpn.p3"], out3, &p3, nullptr, 0, OUT_NORMAL, 0, 0, 0, false, st)); tap(eng, "fpn.p3", p3);
        // The 1/4-resolution tail — lateral in2 (1.5 GB per 16 pages), p2, head.conv1, DBHead tail — runs in groups of tail_group
        // pages: each consumer then finds part of its producer's output still in the 256 MB Infinity Cache (A/B on one device,
        // 64 A4 pages in one forward: p2 3.68 -> 3.47 ms, head.conv1 3.32 -> 3.12 ms), and the 256-channel lateral only ever exists for
        // one group.  Launch order only: results are per page.
        const int G = (eng->keep_taps == 0 && eng->tail_group > 0 && B > eng->tail_group) ? eng->tail_group : B;
        Tensor4 out2{};
        if (!compose) out2 = ws_tensor(eng, G, feats[0].h, feats[0].w, 256);
        Tensor4 p2 = ws_tensor(eng, G, feats[0].h, feats[0].w, 64);
        Tensor4 h1 = ws_tensor(eng, G, Hp / 4, Wp / 4, 64);
        for (int g0 = 0; g0 < B; g0 += G) {
            const int nb = B - g0 < G ? B - g0 : G;
            Tensor4 o2 = slice(out2, 0, nb), p2g = slice(p2, 0, nb), h1g = slice(h1, 0, nb), o3 = slice(out3, g0, nb);
            RUN(run_p2(slice(feats[0], g0, nb), o3, &o2, &p2g, 0, 0));
            tap(eng, "fpn.p2", p2g);
            Tensor4 cat = p2g;
            cat.c = 256; cat.n_src = 4;
            cat.xs[0] = slice(p5, g0, nb).p; cat.xs[1] = slice(p4, g0, nb).p; cat.xs[2] = slice(p3, g0, nb).p; cat.xs[3] = p2g.p;
            cat.xs_shift[0] = 3; cat.xs_shift[1] = 2; cat.xs_shift[2] = 1; cat.xs_shift[3] = 0;
            RUN(eng_run_conv(eng, hc1, cat, &h1g, nullptr, 0, OUT_NORMAL, 0, 0, 0, false, st)); tap(eng, "head.conv1", h1g);
            RUN(head_tail(h1g, prob ? prob + (size_t)g0 * Hp * Wp : nullptr, nb));
        }
        return 0;
    }
    Tensor4 out2{};
    if (!compose) out2 = ws_tensor(eng, B, feats[0].h, feats[0].w, 256);
    // smooth convs write straight into the channel slices of the 1/4-resolution concat (replicated)
    Tensor4 fuse = ws_tensor(eng, B, Hp / 4, Wp / 4, 256);
    RUN(eng_run_conv(eng, D["fpn.p5"], in5, &fuse, nullptr, 0, OUT_UPSAMPLE, 3, 256, 0, false, st));
    RUN(eng_run_conv(eng, D["fpn.p4"], out4, &fuse, nullptr, 0, OUT_UPSAMPLE, 2, 256, 64, false, st));
    RUN(eng_run_conv(eng, D["fpn.p3"], out3, &fuse, nullptr, 0, OUT_UPSAMPLE, 1, 256, 128, false, st));
    RUN(run_p2(feats[0], out3, &out2, &fuse, 256, 192));
    tap(eng, "fpn.fuse", fuse);
    Tensor4 h1 = ws_tensor(eng, B, Hp / 4, Wp / 4, 64);
    RUN(eng_run_conv(eng, hc1, fuse, &h1, nullptr, 0, OUT_NORMAL, 0, 0, 0, false, st)); tap(eng, "head.conv1", h1);
    return head_tail(h1, prob, B);
}

int eng_det_forward(lumina_ocr* eng, const uint8_t* pages, int B, int H, int W, int Hp, int Wp, bf16_t* prob, hipStream_t st) {
    if (!eng->det_loaded) return locr_fail(eng, "det_forward", "det weights not loaded");
    if (Hp % 32 || Wp % 32 || Hp < H || Wp < W || B <= 0) return locr_fail(eng, "det_forward", "Hp/Wp must be multiples of 32 and >= H/W");
    HIPCHK(hipSetDevice(eng->device));
    const int sb = eng->det_sub_batch < B ? eng->det_sub_batch : B;
    // size the workspace with a dry run
    uint8_t* keep = eng->ws; eng->ws = nullptr;
    int rc = det_forward_sub(eng, nullptr, sb, H, W, Hp, Wp, nullptr, st);
    const size_t need = eng->ws_off + 4096;
    eng->ws = keep;
    if (rc) return rc;
    RUN(eng_ws_reserve(eng, need));
    eng->taps.clear();
    for (int b0 = 0; b0 < B; b0 += sb) {
        const int nb = (B - b0) < sb ? (B - b0) : sb;
        RUN(det_forward_sub(eng, pages + (size_t)b0 * H * W * 3, nb, H, W, Hp, Wp, prob + (size_t)b0 * Hp * Wp, st));
    }
    return 0;
}

// ------------------------------------------------------------------------------ rec
static inline int cp16(int c) { return rup(c, 16); }
static int make_div(double v, int d = 8) {
    int nv = (int)(v + d / 2.0) / d * d;
    if (nv < d) nv = d;
    if (nv < 0.9 * v) nv += d;
    return nv;
}

static bool make_dw(lumina_ocr* eng, const std::map<std::string, HostBlobTensor>& m, const std::string& name, int k, int c_r, int c_p, DwLayer* L) {
    const HostBlobTensor *w, *b;
    if (!get_wb(eng, m, name, &w, &b)) return false;
    if (w->dims.size() != 4 || w->dims[0] != c_r || w->dims[1] != k || w->dims[3] != 1) { locr_fail(eng, "dw shape", name.c_str()); return false; }
    std::vector<bf16_t> wt((size_t)k * k * c_p, 0);
    const bf16_t* src = reinterpret_cast<const bf16_t*>(w->data);
    for (int c = 0; c < c_r; ++c)
        for (int t = 0; t < k * k; ++t) wt[(size_t)t * c_p + c] = src[(size_t)c * k * k + t];
    std::vector<float> bias(c_p, 0.f);
    memcpy(bias.data(), b->data, sizeof(float) * c_r);
    L->k = k; L->c = c_p;
    L->w = static_cast<bf16_t*>(dev_upload(eng, wt.data(), wt.size() * sizeof(bf16_t)));
    L->bias = static_cast<float*>(dev_upload(eng, bias.data(), bias.size() * sizeof(float)));
    return L->w && L->bias;
}

static bool make_se(lumina_ocr* eng, const std::map<std::string, HostBlobTensor>& m, const std::string& pfx, int c_r, int c_p, int mid, SeLayer* L) {
    const HostBlobTensor *w1, *b1, *w2, *b2;
    if (!get_wb(eng, m, pfx + ".se1", &w1, &b1) || !get_wb(eng, m, pfx + ".se2", &w2, &b2)) return false;
    if (w1->dims[0] != mid || w1->dims[3] != c_r || w2->dims[0] != c_r || w2->dims[3] != mid) { locr_fail(eng, "se shape", pfx.c_str()); return false; }
    std::vector<bf16_t> a((size_t)mid * c_p, 0), bmat((size_t)c_p * mid, 0);
    const bf16_t* s1 = reinterpret_cast<const bf16_t*>(w1->data);
    const bf16_t* s2 = reinterpret_cast<const bf16_t*>(w2->data);
    for (int i = 0; i < mid; ++i) memcpy(&a[(size_t)i * c_p], &s1[(size_t)i * c_r], sizeof(bf16_t) * c_r);
    memcpy(bmat.data(), s2, sizeof(bf16_t) * (size_t)c_r * mid);
    std::vector<float> bb2(c_p, 0.f);
    memcpy(bb2.data(), b2->data, sizeof(float) * c_r);
    L->c = c_p; L->mid = mid;
    L->w1 = static_cast<bf16_t*>(dev_upload(eng, a.data(), a.size() * sizeof(bf16_t)));
    L->w2 = static_cast<bf16_t*>(dev_upload(eng, bmat.data(), bmat.size() * sizeof(bf16_t)));
    L->b1 = static_cast<float*>(dev_upload(eng, b1->data, sizeof(float) * mid));
    L->b2 = static_cast<float*>(dev_upload(eng, bb2.data(), bb2.size() * sizeof(float)));
    return L->w1 && L->w2 && L->b1 && L->b2;
}

int eng_load_rec(lumina_ocr* eng, const void* blob, size_t n) {
    std::map<std::string, HostBlobTensor> m;
    if (!parse_blob(eng, blob, n, &m)) return 1;
    HIPCHK(hipSetDevice(eng->device));
    const double scale = 0.5;
    const int c0 = make_div(16 * scale);
    {
        const HostBlobTensor *w, *b;
        if (!get_wb(eng, m, "rec.conv1", &w, &b)) return 1;
        if (w->dims[0] != c0 || w->dims[1] != 3 || w->dims[3] != 3) return locr_fail(eng, "rec.conv1", "shape");
        bf16_t packed[2 * 2 * 32 * 8];
        pack_stem_weights(reinterpret_cast<const bf16_t*>(w->data), c0, packed);
        float bias[32] = {0};
        memcpy(bias, b->data, sizeof(float) * c0);
        eng->rstem_wpk = static_cast<bf16_t*>(dev_upload(eng, packed, sizeof(packed)));
        eng->rstem_bias = static_cast<float*>(dev_upload(eng, bias, sizeof(bias)));
    }
    struct Row { int k, exp, c; bool se; int act, sh; };
    const Row rows[11] = {{3, 16, 16, true, ACT_RELU, 1},    {3, 72, 24, false, ACT_RELU, 2},   {3, 88, 24, false, ACT_RELU, 1},
                          {5, 96, 40, true, ACT_HSWISH, 2},  {5, 240, 40, true, ACT_HSWISH, 1}, {5, 240, 40, true, ACT_HSWISH, 1},
                          {5, 120, 48, true, ACT_HSWISH, 1}, {5, 144, 48, true, ACT_HSWISH, 1}, {5, 288, 96, true, ACT_HSWISH, 2},
                          {5, 576, 96, true, ACT_HSWISH, 1}, {5, 576, 96, true, ACT_HSWISH, 1}};
    eng->rblocks.clear();
    eng->rblocks.resize(11);
    int cin = c0;
    for (int i = 0; i < 11; ++i) {
        RecBlock& B = eng->rblocks[i];
        B.k = rows[i].k; B.cin = cin; B.exp = make_div(rows[i].exp * scale); B.cout = make_div(rows[i].c * scale);
        B.se = rows[i].se; B.se_mid = B.exp / 4; B.stride_h = rows[i].sh; B.act = rows[i].act;
        B.res = (B.stride_h == 1 && B.cin == B.cout);
        const std::string p = "rec.b" + std::to_string(i);
        if (!make_conv(eng, m, p + ".expand", 1, 1, B.cin, B.exp, cp16(B.cin), cp16(B.exp), B.act, &B.expand)) return 1;
        if (!make_dw(eng, m, p + ".dw", B.k, B.exp, cp16(B.exp), &B.dw)) return locr_fail(eng, "dw", p.c_str());
        {   // the same expand weights once more, in the fused kernel's fragment order
            const HostBlobTensor *w, *b;
            if (!get_wb(eng, m, p + ".expand", &w, &b)) return 1;
            const int ec = cp16(B.exp), cc = cp16(B.cin);
            std::vector<bf16_t> padded((size_t)ec * cc, 0), packed(mbconv_expand_packed_elems(ec, cc));
            const bf16_t* src = reinterpret_cast<const bf16_t*>(w->data);
            for (int r = 0; r < B.exp; ++r) memcpy(&padded[(size_t)r * cc], &src[(size_t)r * B.cin], sizeof(bf16_t) * B.cin);
            mbconv_pack_expand(padded.data(), ec, cc, packed.data());
            std::vector<float> bias((size_t)((ec + 31) / 32) * 32, 0.f);
            memcpy(bias.data(), b->data, sizeof(float) * B.exp);
            B.we_pk = static_cast<bf16_t*>(dev_upload(eng, packed.data(), packed.size() * sizeof(bf16_t)));
            B.be_pk = static_cast<float*>(dev_upload(eng, bias.data(), bias.size() * sizeof(float)));
            if (!B.we_pk || !B.be_pk) return locr_fail(eng, "upload", (p + ".expand (fused)").c_str());
        }
        if (B.se && !make_se(eng, m, p, B.exp, cp16(B.exp), B.se_mid, &B.sel)) return 1;
        if (!make_conv(eng, m, p + ".project", 1, 1, B.exp, B.cout, cp16(B.exp), cp16(B.cout), ACT_NONE, &B.project)) return 1;
        cin = B.cout;
    }
    if (!make_conv(eng, m, "rec.conv2", 1, 1, cin, 288, cp16(cin), 288, ACT_HSWISH, &eng->rconv2)) return 1;
    // LSTM: x-projection of both directions as one GEMM; recurrent weights [2][4H][H]
    const int Hh = 96;
    for (int l = 0; l < 2; ++l) {
        const int din = l == 0 ? 288 : 2 * Hh;
        const std::string pf = "lstm.l" + std::to_string(l) + ".fw", pb = "lstm.l" + std::to_string(l) + ".bw";
        auto wf = m.find(pf + ".w_ih"), wb = m.find(pb + ".w_ih"), bf = m.find(pf + ".b"), bb = m.find(pb + ".b");
        auto hf = m.find(pf + ".w_hh"), hb = m.find(pb + ".w_hh");
        if (wf == m.end() || wb == m.end() || bf == m.end() || bb == m.end() || hf == m.end() || hb == m.end())
            return locr_fail(eng, "lstm tensors missing", pf.c_str());
        if (wf->second.dims[0] != 4 * Hh || wf->second.dims[1] != din || hf->second.dims[1] != Hh) return locr_fail(eng, "lstm shape", pf.c_str());
        // synthesise a 1x1 conv blob entry [8H][1][1][din]
        std::vector<uint8_t> wcat(wf->second.nbytes + wb->second.nbytes), bcat(bf->second.nbytes + bb->second.nbytes);
        memcpy(wcat.data(), wf->second.data, wf->second.nbytes); memcpy(wcat.data() + wf->second.nbytes, wb->second.data, wb->second.nbytes);
        memcpy(bcat.data(), bf->second.data, bf->second.nbytes); memcpy(bcat.data() + bf->second.nbytes, bb->second.data, bb->second.nbytes);
        std::map<std::string, HostBlobTensor> mm;
        mm["x.w"] = HostBlobTensor{1, {8 * Hh, 1, 1, din}, wcat.data(), wcat.size()};
        mm["x.b"] = HostBlobTensor{0, {8 * Hh}, bcat.data(), bcat.size()};
        if (!make_conv(eng, mm, "x", 1, 1, din, 8 * Hh, din, 8 * Hh, ACT_NONE, &eng->xproj[l])) return 1;
        eng->xproj[l].name = "lstm.l" + std::to_string(l) + ".xproj";
        std::vector<uint8_t> hcat(hf->second.nbytes + hb->second.nbytes);
        memcpy(hcat.data(), hf->second.data, hf->second.nbytes); memcpy(hcat.data() + hf->second.nbytes, hb->second.data, hb->second.nbytes);
        eng->whh[l] = static_cast<bf16_t*>(dev_upload(eng, hcat.data(), hcat.size()));
        if (!eng->whh[l]) return locr_fail(eng, "upload", "whh");
    }
    {
        auto w = m.find("ctc.fc.w"), b = m.find("ctc.fc.b");
        if (w == m.end() || b == m.end() || w->second.dims[1] != 2 * Hh) return locr_fail(eng, "ctc.fc", "missing/shape");
        const int C = w->second.dims[0];
        eng->num_classes = C; eng->ctc_ntiles = (C + 63) / 64;
        std::vector<bf16_t> packed(ctc_packed_weight_elems(C, 2 * Hh));
        pack_ctc_weights(reinterpret_cast<const bf16_t*>(w->second.data), C, 2 * Hh, packed.data());
        std::vector<float> bias((size_t)eng->ctc_ntiles * 64, -1.0e30f);
        memcpy(bias.data(), b->second.data, sizeof(float) * C);
        eng->ctc_wpk = static_cast<bf16_t*>(dev_upload(eng, packed.data(), packed.size() * sizeof(bf16_t)));
        eng->ctc_bias = static_cast<float*>(dev_upload(eng, bias.data(), bias.size() * sizeof(float)));
        if (!eng->ctc_wpk || !eng->ctc_bias) return locr_fail(eng, "upload", "ctc");
    }
    eng->rec_loaded = true;
    return 0;
}

#define LAUNCH(name, expr)                                                                  \
    do {                                                                                    \
        if (!dry) { hipError_t _e = (expr); if (_e != hipSuccess) return locr_fail(eng, name, hipGetErrorString(_e)); } \
    } while (0)

static int rec_forward_sub(lumina_ocr* eng, const uint8_t* crops, const int* widths, int N, int* idx, float* prob, hipStream_t st) {
    eng->ws_off = 0;
    const bool dry = (eng->ws == nullptr) || crops == nullptr;
    const int T = 80;
    Tensor4 x = ws_tensor(eng, N, 16, 160, 16);
    if (!dry && x.p) {
        StemParams sp{};
        sp.x = crops; sp.wpk = eng->rstem_wpk; sp.bias = eng->rstem_bias; sp.y = x.p; sp.valid_w_per_img = widths;
        sp.N = N; sp.H = 32; sp.W = 320; sp.valid_h = 32; sp.valid_w = 320; sp.Ho = 16; sp.Wo = 160; sp.Cout_store = 16;
        sp.act = ACT_HSWISH;
        for (int c = 0; c < 3; ++c) { sp.scale[c] = 2.0f / 255.0f; sp.shift[c] = -1.0f; }
        hipError_t e = stem_conv_launch(sp, st);
        if (e != hipSuccess) return locr_fail(eng, "rec.conv1", hipGetErrorString(e));
    }
    tap(eng, "rec.conv1", x);
    for (size_t bi = 0; bi < eng->rblocks.size(); ++bi) {
        RecBlock& B = eng->rblocks[bi];
        const int ho = (x.h + 2 * (B.k / 2) - B.k) / B.stride_h + 1;
        MbParams mp{};
        mp.we = B.we_pk; mp.be = B.be_pk; mp.wd = B.dw.w; mp.bd = B.dw.bias;
        mp.N = N; mp.H = x.h; mp.W = x.w; mp.cin = x.c; mp.expc = cp16(B.exp); mp.Ho = ho; mp.act = B.act;
        const bool fused_mb = eng->fuse_mb && mbconv_supported(mp, B.k, B.stride_h);
        Tensor4 e1{};
        if (!fused_mb || dry) e1 = ws_tensor(eng, N, x.h, x.w, cp16(B.exp));   // (dry run sizes the arena for the unfused path too)
        Tensor4 d = ws_tensor(eng, N, ho, x.w, cp16(B.exp));
        float* pool = B.se ? static_cast<float*>(eng_ws_alloc(eng, (size_t)N * mb_strips(x.w) * d.c * sizeof(float))) : nullptr;
        if (fused_mb) {
            mp.x = x.p; mp.d = d.p; mp.pool = pool;   // squeeze-excite blocks: the pooled sums leave with the tile (the tensor is not read again for them)
            if (!dry && x.p && d.p) {
                LaunchTimer tm(eng, st, true);
                LAUNCH("mbconv", mbconv_launch(mp, B.k, B.stride_h, st));
                const double opx = (double)N * ho * x.w, ipx = (double)N * x.h * x.w;
                tm.done("rec.b" + std::to_string(bi) + ".expand+dw", "mbconv_kernel<" + std::to_string(B.k) + "," + std::to_string(B.stride_h) + "," + std::to_string(B.act) + ">",
                        2.0 * ipx * x.c * mp.expc + 2.0 * opx * B.k * B.k * mp.expc, 2.0 * (ipx * x.c + opx * mp.expc));
            }
        } else {
            RUN(eng_run_conv(eng, B.expand, x, &e1, nullptr, 0, OUT_NORMAL, 0, 0, 0, true, st));
            LAUNCH("dwconv", dwconv_launch(e1.p, B.dw.w, B.dw.bias, d.p, N, e1.h, e1.w, e1.c, B.k, B.stride_h, B.act, st));
            if (B.se) LAUNCH("se_pool", se_pool_launch(d.p, pool, N, d.h, d.w, d.c, st));   // the same sums in the same order
        }
        const bf16_t* se_gate_ptr = nullptr;
        if (B.se) {
            bf16_t* gate = static_cast<bf16_t*>(eng_ws_alloc(eng, (size_t)N * d.c * sizeof(bf16_t)));
            LAUNCH("se_fc", se_fc_launch(pool, mb_strips(d.w), B.sel.w1, B.sel.b1, B.sel.w2, B.sel.b2, gate, N, d.h * d.w, d.c, B.sel.mid, st));
            se_gate_ptr = gate;   // the scaling itself is fused into the project conv's operand staging
        }
        Tensor4 o = ws_tensor(eng, N, d.h, d.w, cp16(B.cout));
        RUN(eng_run_conv(eng, B.project, d, &o, B.res ? &x : nullptr, 0, OUT_NORMAL, 0, 0, 0, true, st, se_gate_ptr));
        tap(eng, ("rec.b" + std::to_string(bi)).c_str(), o);
        x = o;
    }
    Tensor4 f = ws_tensor(eng, N, x.h, x.w, 288);
    RUN(eng_run_conv(eng, eng->rconv2, x, &f, nullptr, 0, OUT_NORMAL, 0, 0, 0, true, st)); tap(eng, "rec.conv2", f);
    Tensor4 seq = ws_tensor(eng, N, 1, T, 288);
    LAUNCH("rec.pool", maxpool_launch(f.p, seq.p, N, f.h, f.w, 288, 2, 2, 0, 1, T, st));
    tap(eng, "rec.feat", seq);
    for (int l = 0; l < 2; ++l) {
        Tensor4 xp = ws_tensor(eng, N, 1, T, 8 * 96);
        RUN(eng_run_conv(eng, eng->xproj[l], seq, &xp, nullptr, 0, OUT_NORMAL, 0, 0, 0, true, st));
        Tensor4 hs = ws_tensor(eng, N, 1, T, 2 * 96);
        {
            LaunchTimer tm(eng, st, !dry);
            LAUNCH("lstm", lstm_recurrent_launch(xp.p, eng->whh[l], hs.p, N, T, st));
            tm.done("lstm.l" + std::to_string(l), "lstm_kernel", 2.0 * N * T * 2 * 96 * 384, 2.0 * N * T * (768 + 192));
        }
        tap(eng, l == 0 ? "lstm.l0" : "lstm.l1", hs);
        seq = hs;
    }
    if (!dry) {
        CtcFcParams cp{};
        cp.seq = seq.p; cp.wpk = eng->ctc_wpk; cp.bias = eng->ctc_bias; cp.out_idx = idx; cp.out_prob = prob;
        cp.M = N * T; cp.K = 192; cp.C = eng->num_classes; cp.ntiles = eng->ctc_ntiles;
        LaunchTimer tm(eng, st, true);
        hipError_t e = ctc_fc_argmax_launch(cp, st);
        if (e != hipSuccess) return locr_fail(eng, "ctc_fc_argmax", hipGetErrorString(e));
        tm.done("ctc.fc+argmax", "ctc_fc_argmax_kernel<0>", 2.0 * cp.M * cp.K * cp.C, 2.0 * cp.M * cp.K + 8.0 * cp.M + 2.0 * cp.K * cp.C);
    }
    return 0;
}

int eng_rec_forward(lumina_ocr* eng, const uint8_t* crops, const int* widths, int N, int* idx, float* prob, hipStream_t st) {
    if (!eng->rec_loaded) return locr_fail(eng, "rec_forward", "rec weights not loaded");
    if (N <= 0) return 0;
    HIPCHK(hipSetDevice(eng->device));
    const int sb = eng->rec_sub_batch < N ? eng->rec_sub_batch : N;
    uint8_t* keep = eng->ws; eng->ws = nullptr;
    int rc = rec_forward_sub(eng, nullptr, nullptr, sb, nullptr, nullptr, st);
    const size_t need = eng->ws_off + 4096;
    eng->ws = keep;
    if (rc) return rc;
    RUN(eng_ws_reserve(eng, need));
    eng->taps.clear();
    for (int b0 = 0; b0 < N; b0 += sb) {
        const int nb = (N - b0) < sb ? (N - b0) : sb;
        RUN(rec_forward_sub(eng, crops + (size_t)b0 * 32 * 320 * 3, widths ? widths + b0 : nullptr, nb, idx + (size_t)b0 * 80, prob + (size_t)b0 * 80, st));
    }
    return 0;
}

// ------------------------------------------------------------------------------ SVTR recogniser (Tiny / Base, bf16 / fp16)
static float* upload_f32(lumina_ocr* eng, const std::map<std::string, HostBlobTensor>& m, const std::string& name, int n) {
    auto it = m.find(name);
    if (it == m.end() || it->second.dtype != 0 || (int)it->second.dims[0] != n) { locr_fail(eng, "missing/ill-shaped f32 tensor", name.c_str()); return nullptr; }
    return static_cast<float*>(dev_upload(eng, it->second.data, sizeof(float) * n));
}

static inline uint16_t bf16_bits_to(uint16_t b, int dtype) {   // bf16 bits -> the model's storage type (fp16: exact for |w| >= 2^-14)
    if (!dtype) return b;
    uint32_t u = (uint32_t)b << 16;
    float f; memcpy(&f, &u, 4);
    const _Float16 hv = (_Float16)f;
    uint16_t o; memcpy(&o, &hv, 2);
    return o;
}

// name.w [N][ks][ks][cin] bf16 (+ name.b) -> SvtrLinear with w [N][taps * cin_pad] in the storage type; ln != "" fuses that LayerNorm
static bool make_linear(lumina_ocr* eng, const std::map<std::string, HostBlobTensor>& m, const std::string& name, int ks, int cin, int cin_pad, int N, int act,
                        const std::string& ln, int dtype, bool flat_k, SvtrLinear* L) {
    const HostBlobTensor *w, *b;
    if (!get_wb(eng, m, name, &w, &b)) return false;
    if (w->dims.size() != 4 || w->dims[0] != N || w->dims[1] != ks || w->dims[2] != ks || w->dims[3] != cin || (int)b->dims[0] != N) {
        locr_fail(eng, "unexpected tensor shape", name.c_str());
        return false;
    }
    const int taps = ks * ks;
    const bf16_t* src = reinterpret_cast<const bf16_t*>(w->data);
    // flat_k: the taps * cin real values of a row are packed densely and the ROW is padded to cin_pad (patch embedding 1: 27 -> 32)
    const int K = flat_k ? cin_pad : taps * cin_pad;
    std::vector<uint16_t> packed((size_t)N * K, 0);
    for (int n = 0; n < N; ++n)
        for (int t = 0; t < taps; ++t)
            for (int c = 0; c < cin; ++c)
                packed[(size_t)n * K + (flat_k ? t * cin + c : t * cin_pad + c)] = bf16_bits_to(src[((size_t)n * taps + t) * cin + c], dtype);
    L->K = K; L->N = N; L->taps = flat_k ? 1 : taps; L->cin = flat_k ? cin_pad : cin_pad; L->act = act;
    L->w = static_cast<uint16_t*>(dev_upload(eng, packed.data(), packed.size() * 2));
    L->bias = static_cast<float*>(dev_upload(eng, b->data, sizeof(float) * N));
    if (!ln.empty()) {
        L->gamma = upload_f32(eng, m, ln + ".g", N); L->beta = upload_f32(eng, m, ln + ".b", N);
        if (!L->gamma || !L->beta) return false;
    }
    if (!L->w || !L->bias) { locr_fail(eng, "device upload failed", name.c_str()); return false; }
    return true;
}

int eng_load_svtr(lumina_ocr* eng, const void* blob, size_t n) {
    std::map<std::string, HostBlobTensor> m;
    if (!parse_blob(eng, blob, n, &m)) return 1;
    HIPCHK(hipSetDevice(eng->device));
    SvtrModel& M = eng->svtr;
    M = SvtrModel();
    {   // variant + storage type
        auto it = m.find("svtr.config");
        if (it != m.end()) {
            if (it->second.dtype != 0 || it->second.dims.size() != 1 || it->second.dims[0] != 12) return locr_fail(eng, "svtr.config", "expected 12 f32 values");
            const float* c = reinterpret_cast<const float*>(it->second.data);
            for (int i = 0; i < 3; ++i) { M.dims[i] = (int)c[i]; M.depths[i] = (int)c[3 + i]; M.heads[i] = (int)c[6 + i]; }
            M.local_blocks = (int)c[9]; M.out_ch = (int)c[10]; M.dtype = (int)c[11] != 0;
        }
        if (eng->svtr_f16 >= 0) M.dtype = eng->svtr_f16;
        for (int i = 0; i < 3; ++i) {
            const int d = M.dims[i];
            if (!(d == 64 || d == 128 || d == 256 || d == 384) || M.heads[i] * 32 != d || M.depths[i] < 1 || M.depths[i] > 32)
                return locr_fail(eng, "svtr.config", "dims must be 64 / 128 / 256 / 384 with 32-wide heads");
        }
        if (M.out_ch != 192 || M.dims[0] > 128) return locr_fail(eng, "svtr.config", "out_channels must be 192 (CTC head K), dims[0] <= 128");
    }
    const int dt = M.dtype, d0 = M.dims[0];
    if (!make_linear(eng, m, "svtr.pe1", 3, 3, 32, d0 / 2, ACT_GELU, "", dt, true, &M.pe1)) return 1;
    if (!make_linear(eng, m, "svtr.pe2", 3, d0 / 2, d0 / 2, d0, ACT_GELU, "", dt, false, &M.pe2)) return 1;
    {
        auto it = m.find("svtr.pos.w");
        if (it == m.end() || it->second.dtype != 1 || it->second.dims.size() != 2 || it->second.dims[0] != 640 || it->second.dims[1] != d0)
            return locr_fail(eng, "svtr.pos", "missing/shape");
        const bf16_t* src = reinterpret_cast<const bf16_t*>(it->second.data);
        std::vector<uint16_t> pos((size_t)640 * d0);
        for (size_t i = 0; i < pos.size(); ++i) pos[i] = bf16_bits_to(src[i], dt);
        M.pos = static_cast<uint16_t*>(dev_upload(eng, pos.data(), pos.size() * 2));
        if (!M.pos) return locr_fail(eng, "upload", "svtr.pos");
    }
    int idx = 0, gh = 8;
    for (int s = 0; s < 3; ++s) {
        const int c = M.dims[s];
        for (int d = 0; d < M.depths[s]; ++d, ++idx) {
            M.blocks.emplace_back();
            SvtrBlock& B = M.blocks.back();
            B.dim = c; B.heads = M.heads[s]; B.gh = gh; B.gw = 80; B.local = idx < M.local_blocks;
            if (B.local && gh % 4 != 0) return locr_fail(eng, "svtr.config", "local mixing blocks need a token grid of at least 4 rows");
            const std::string p = "svtr.b" + std::to_string(idx);
            if (!make_linear(eng, m, p + ".qkv", 1, c, c, 3 * c, ACT_NONE, "", dt, false, &B.qkv)) return 1;
            if (!make_linear(eng, m, p + ".proj", 1, c, c, c, ACT_NONE, p + ".ln1", dt, false, &B.proj)) return 1;
            if (!make_linear(eng, m, p + ".fc1", 1, c, c, 4 * c, ACT_GELU, "", dt, false, &B.fc1)) return 1;
            if (!make_linear(eng, m, p + ".fc2", 1, 4 * c, 4 * c, c, ACT_NONE, p + ".ln2", dt, false, &B.fc2)) return 1;
        }
        if (s < 2) {
            const std::string p = "svtr.sub" + std::to_string(s);
            if (!make_linear(eng, m, p, 3, c, c, M.dims[s + 1], ACT_NONE, p + ".ln", dt, false, &M.sub[s])) return 1;
            gh /= 2;
        }
    }
    if (!make_linear(eng, m, "svtr.last", 1, M.dims[2], M.dims[2], M.out_ch, ACT_HSWISH, "", dt, false, &M.last)) return 1;
    {
        auto w = m.find("svtr.ctc.fc.w"), b = m.find("svtr.ctc.fc.b");
        if (w == m.end() || b == m.end() || w->second.dims.size() != 2 || w->second.dims[1] != M.out_ch) return locr_fail(eng, "svtr.ctc.fc", "missing/shape");
        const int C = w->second.dims[0];
        M.num_classes = C; M.ctc_ntiles = (C + 63) / 64;
        std::vector<bf16_t> conv((size_t)C * M.out_ch);
        const bf16_t* src = reinterpret_cast<const bf16_t*>(w->second.data);
        for (size_t i = 0; i < conv.size(); ++i) conv[i] = bf16_bits_to(src[i], dt);
        std::vector<bf16_t> packed(ctc_packed_weight_elems(C, M.out_ch));
        pack_ctc_weights(conv.data(), C, M.out_ch, packed.data());
        std::vector<float> bias((size_t)M.ctc_ntiles * 64, -1.0e30f);
        memcpy(bias.data(), b->second.data, sizeof(float) * C);
        M.ctc_wpk = static_cast<bf16_t*>(dev_upload(eng, packed.data(), packed.size() * sizeof(bf16_t)));
        M.ctc_bias = static_cast<float*>(dev_upload(eng, bias.data(), bias.size() * sizeof(float)));
        if (!M.ctc_wpk || !M.ctc_bias) return locr_fail(eng, "upload", "svtr ctc");
    }
    M.loaded = true;
    return 0;
}

// one svtr_gemm launch: y = epi(gather(x) w^T + b)
static int run_linear(lumina_ocr* eng, const SvtrLinear& L, const Tensor4& x, Tensor4* y, const bf16_t* res, int res_mod, int res_post, int hin, int win,
                      int hout, int wout, int sh, int sw, bool dry, hipStream_t st) {
    if (dry) return 0;
    if (eng->zero_block == nullptr) {
        const uint32_t z[64] = {0};
        eng->zero_block = static_cast<bf16_t*>(dev_upload(eng, z, sizeof(z)));
        if (!eng->zero_block) return locr_fail(eng, "svtr", "zero block upload failed");
    }
    SvtrGemmParams p{};
    p.x = x.p; p.w = L.w; p.bias = L.bias; p.res = res; p.res_mod = res_mod; p.res_post = res_post; p.gamma = L.gamma; p.beta = L.beta; p.y = y->p;
    p.zeros = eng->zero_block; p.M = (int)(y->elems() / L.N); p.K = L.K; p.N = L.N; p.act = L.act; p.eps = 1e-6f;
    p.taps = L.taps; p.Cin = L.cin; p.Hin = hin; p.Win = win; p.Tout = hout * wout; p.Wout = wout; p.sh = sh; p.sw = sw;
    if (x.c != L.cin || y->c != L.N) return locr_fail(eng, "svtr linear: channel mismatch", "");
    LaunchTimer tm(eng, st, true);
    hipError_t e = svtr_gemm_launch(p, eng->svtr.dtype, st);
    if (e != hipSuccess) return locr_fail(eng, "svtr_gemm", hipGetErrorString(e));
    // algorithmic bytes: the input tensor, the weights, the result and the residual once (a 9-tap gather re-reads its input from cache)
    tm.done("svtr.linear", svtr_gemm_kernel_name(p, eng->svtr.dtype), 2.0 * p.M * p.K * p.N,
            2.0 * ((double)x.elems() + (double)p.N * p.K + (double)p.M * p.N + (res && res_mod == 0 ? (double)p.M * p.N : 0.0)));
    return 0;
}

static int svtr_forward_sub(lumina_ocr* eng, const uint8_t* crops, const int* widths, int N, int* idx, float* prob, hipStream_t st) {
    eng->ws_off = 0;
    const bool dry = (eng->ws == nullptr) || crops == nullptr;
    SvtrModel& M = eng->svtr;
    const int T = 80, dt = M.dtype, d0 = M.dims[0];
    Tensor4 patches = ws_tensor(eng, N, 16, 160, 32);
    LAUNCH("svtr.im2col", svtr_im2col_launch(crops, widths, patches.p, N, dt, st));
    Tensor4 e1 = ws_tensor(eng, N, 16, 160, d0 / 2);
    RUN(run_linear(eng, M.pe1, patches, &e1, nullptr, 0, 0, 1, 1, 1, 1, 1, 1, dry, st));
    Tensor4 x = ws_tensor(eng, N, 8, 80, d0);   // patch embedding 2 (3x3 / s2, GELU, rounded) + positional embedding (rounded again)
    RUN(run_linear(eng, M.pe2, e1, &x, M.pos, 640, 1, 16, 160, 8, 80, 2, 2, dry, st));
    tap(eng, "svtr.embed", x);
    int stage = 0;
    for (size_t bi = 0; bi < M.blocks.size(); ++bi) {
        SvtrBlock& B = M.blocks[bi];
        const int want_stage = (int)bi < M.depths[0] ? 0 : ((int)bi < M.depths[0] + M.depths[1] ? 1 : 2);
        if (want_stage != stage) {
            // height merging: 3x3 conv, stride (2, 1), + LayerNorm (fused epilogue)
            Tensor4 y = ws_tensor(eng, N, x.h / 2, x.w, M.dims[stage + 1]);
            RUN(run_linear(eng, M.sub[stage], x, &y, nullptr, 0, 0, x.h, x.w, x.h / 2, x.w, 2, 1, dry, st));
            tap(eng, stage == 0 ? "svtr.sub0" : "svtr.sub1", y);
            x = y; ++stage;
        }
        const int Tk = x.h * x.w, c = B.dim;
        Tensor4 qkv = ws_tensor(eng, N, x.h, x.w, 3 * c);
        RUN(run_linear(eng, B.qkv, x, &qkv, nullptr, 0, 0, 1, 1, 1, 1, 1, 1, dry, st));
        Tensor4 att = ws_tensor(eng, N, x.h, x.w, c);
        {
            LaunchTimer tm(eng, st, !dry);
            LAUNCH("svtr.attn", svtr_attention_launch(qkv.p, att.p, N, Tk, B.heads, B.gh, B.gw, B.local ? 1 : 0, dt, st));
            const double keys = B.local ? 77.0 : (double)Tk;   // keys a query attends (7 x 11 window / all)
            tm.done("svtr.attn", std::string("svtr_attn_kernel<") + (dt ? "1>" : "0>"), 4.0 * N * Tk * keys * c, 2.0 * N * Tk * 4.0 * c);
        }
        Tensor4 x1 = ws_tensor(eng, N, x.h, x.w, c);
        RUN(run_linear(eng, B.proj, att, &x1, x.p, 0, 0, 1, 1, 1, 1, 1, 1, dry, st));        // + residual, LayerNorm 1
        Tensor4 f1 = ws_tensor(eng, N, x.h, x.w, 4 * c);
        RUN(run_linear(eng, B.fc1, x1, &f1, nullptr, 0, 0, 1, 1, 1, 1, 1, 1, dry, st));
        Tensor4 x2 = ws_tensor(eng, N, x.h, x.w, c);
        RUN(run_linear(eng, B.fc2, f1, &x2, x1.p, 0, 0, 1, 1, 1, 1, 1, 1, dry, st));         // + residual, LayerNorm 2
        tap(eng, ("svtr.b" + std::to_string(bi)).c_str(), x2);
        x = x2;
    }
    Tensor4 pooled = ws_tensor(eng, N, 1, T, M.dims[2]);
    LAUNCH("svtr.pool", svtr_rowmean_launch(x.p, pooled.p, N, x.h, x.w, M.dims[2], dt, st));
    Tensor4 seq = ws_tensor(eng, N, 1, T, M.out_ch);
    RUN(run_linear(eng, M.last, pooled, &seq, nullptr, 0, 0, 1, 1, 1, 1, 1, 1, dry, st));
    tap(eng, "svtr.seq", seq);
    if (!dry) {
        CtcFcParams cp{};
        cp.seq = seq.p; cp.wpk = M.ctc_wpk; cp.bias = M.ctc_bias; cp.out_idx = idx; cp.out_prob = prob;
        cp.M = N * T; cp.K = M.out_ch; cp.C = M.num_classes; cp.ntiles = M.ctc_ntiles; cp.f16 = dt;
        LaunchTimer tm(eng, st, true);
        hipError_t e = ctc_fc_argmax_launch(cp, st);
        if (e != hipSuccess) return locr_fail(eng, "svtr ctc_fc_argmax", hipGetErrorString(e));
        tm.done("svtr.ctc.fc+argmax", std::string("ctc_fc_argmax_kernel<") + (dt ? "1>" : "0>"), 2.0 * cp.M * cp.K * cp.C, 2.0 * cp.M * cp.K + 8.0 * cp.M + 2.0 * cp.K * cp.C);
    }
    return 0;
}

int eng_svtr_forward(lumina_ocr* eng, const uint8_t* crops, const int* widths, int N, int* idx, float* prob, hipStream_t st) {
    if (!eng->svtr.loaded) return locr_fail(eng, "svtr_forward", "SVTR weights not loaded");
    if (N <= 0) return 0;
    HIPCHK(hipSetDevice(eng->device));
    int sb = eng->rec_sub_batch / (eng->svtr.dims[2] > 256 ? 4 : 2);   // ~3 MB (Tiny) / ~7 MB (Base) of activations per crop
    if (sb < 1) sb = 1;
    if (sb > N) sb = N;
    uint8_t* keep = eng->ws; eng->ws = nullptr;
    int rc = svtr_forward_sub(eng, nullptr, nullptr, sb, nullptr, nullptr, st);
    const size_t need = eng->ws_off + 4096;
    eng->ws = keep;
    if (rc) return rc;
    RUN(eng_ws_reserve(eng, need));
    eng->taps.clear();
    for (int b0 = 0; b0 < N; b0 += sb) {
        const int nb = (N - b0) < sb ? (N - b0) : sb;
        RUN(svtr_forward_sub(eng, crops + (size_t)b0 * 32 * 320 * 3, widths ? widths + b0 : nullptr, nb, idx + (size_t)b0 * 80, prob + (size_t)b0 * 80, st));
    }
    return 0;
}
