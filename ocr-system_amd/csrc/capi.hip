// extern "C" surface of liblumina_ocr.so (declared in include/lumina_ocr.h). Nothing throws.
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/lumina_ocr.h"
#include "dbpost.h"
#include "deskew.h"
#include "engine.h"
#include "ops.h"
#include "resize.h"
#include "jpeg.h"
#include "jpegdec.h"
#include "stem_conv.h"

#define API_TRY try {
#define API_CATCH(h)                                                             \
    }                                                                            \
    catch (const std::exception& e) { return locr_fail(h, "exception", e.what()); } \
    catch (...) { return locr_fail(h, "exception", "unknown"); }

// every entry that allocates, copies or launches binds the calling thread to the handle's device first: provider threads
// (asyncio.to_thread workers) start on device 0 whatever device the handle was created on.  The thread's previous device is put
// back when the entry returns (a host that keeps its own current device — torch without an explicit device, a second engine on
// another GPU — is not silently rebound).
namespace {
struct DeviceGuard {
    int prev = -1;
    hipError_t err;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        err = hipSetDevice(dev);
        if (prev == dev) prev = -1;   // nothing to restore
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
}  // namespace
#define BIND(h)                                                                                              \
    if (!(h)) return 1;                                                                                      \
    DeviceGuard _device_guard((h)->device);                                                                  \
    if (_device_guard.err != hipSuccess) return locr_fail((h), "hipSetDevice", hipGetErrorString(_device_guard.err))

extern "C" {

const char* lumina_ocr_version(void) { return "lumina-ocr-mi355x 0.1 (gfx950)"; }

int lumina_ocr_create(int device, lumina_ocr_t** out) {
    if (!out) return 1;
    *out = nullptr;
    lumina_ocr* eng = new (std::nothrow) lumina_ocr();
    if (!eng) return 1;
    eng->device = device;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        eng->err = "no HIP device " + std::to_string(device) + " (device count " + std::to_string(ndev) + ")";
        *out = eng;  // handle is returned so the caller can read the error
        return 2;
    }
    if (hipSetDevice(device) != hipSuccess) { eng->err = "hipSetDevice failed"; *out = eng; return 2; }
    if (const char* e = getenv("LUMINA_RING_ORIENT")) eng->ring_orient = atoi(e) < 0 ? -1 : (atoi(e) != 0);   // developer A/B (profiler runs)
    if (getenv("LUMINA_CONV_NO_RING")) eng->conv_ring = false;
    if (getenv("LUMINA_BLOCKED")) eng->blocked_layout = atoi(getenv("LUMINA_BLOCKED")) != 0;
    *out = eng;
    return 0;
}

void lumina_ocr_destroy(lumina_ocr_t* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    for (void* p : h->owned) (void)hipFree(p);
    if (h->ws) (void)hipFree(h->ws);
    if (h->aux) (void)hipFree(h->aux);
    for (int k = 0; k < 2; ++k) { if (h->jd_stage[k]) (void)hipHostFree(h->jd_stage[k]); if (h->jd_stage_ev[k]) (void)hipEventDestroy(h->jd_stage_ev[k]); }
    for (auto& ev : h->conv_events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    delete h;
}

const char* lumina_ocr_last_error(const lumina_ocr_t* h) { return h ? h->err.c_str() : "null handle"; }

int lumina_ocr_set_option(lumina_ocr_t* h, const char* key, int value) {
    if (!h || !key) return 1;
    if (!strcmp(key, "det_sub_batch")) h->det_sub_batch = value > 0 ? value : 1;
    else if (!strcmp(key, "rec_sub_batch")) h->rec_sub_batch = value > 0 ? value : 1;
    else if (!strcmp(key, "keep_taps")) h->keep_taps = value < 0 || value > 2 ? 0 : value;
    else if (!strcmp(key, "time_convs")) h->time_convs = value != 0;
    else if (!strcmp(key, "fuse_head")) h->fuse_head = value != 0;
    else if (!strcmp(key, "fuse_mb")) h->fuse_mb = value != 0;
    else if (!strcmp(key, "fuse_pool")) h->fuse_pool = value != 0;
    else if (!strcmp(key, "fuse_stem")) h->fuse_stem = value != 0;
    else if (!strcmp(key, "fpn_multi")) h->fpn_multi = value != 0;
    else if (!strcmp(key, "fpn_compose")) h->fpn_compose = value != 0;
    else if (!strcmp(key, "fuse_short")) h->fuse_short = value != 0;
    else if (!strcmp(key, "conv_ring")) h->conv_ring = value != 0;
    else if (!strcmp(key, "blocked_layout")) h->blocked_layout = value != 0;
    else if (!strcmp(key, "conv_big_min")) h->conv_big_min = value >= 0 ? value : 1024;
    else if (!strcmp(key, "ring_orient")) h->ring_orient = value < 0 ? -1 : (value != 0);
    else if (!strcmp(key, "post_group")) h->post_group = value > 0 ? value : 1;
    else if (!strcmp(key, "tail_group")) h->tail_group = value > 0 ? value : 0;
    else if (!strcmp(key, "svtr_f16")) h->svtr_f16 = value < 0 ? -1 : (value != 0);
    else if (!strcmp(key, "conv2d_variant")) h->conv2d_variant = value < 0 || value > 2 ? 0 : value;
    else return locr_fail(h, "set_option: unknown key", key);
    return 0;
}

int lumina_ocr_load_det_weights(lumina_ocr_t* h, const void* blob, size_t nbytes) {
    if (!h || !blob) return 1;
    BIND(h);
    API_TRY return eng_load_det(h, blob, nbytes); API_CATCH(h)
}
int lumina_ocr_load_rec_weights(lumina_ocr_t* h, const void* blob, size_t nbytes) {
    if (!h || !blob) return 1;
    BIND(h);
    API_TRY return eng_load_rec(h, blob, nbytes); API_CATCH(h)
}
int lumina_ocr_num_classes(const lumina_ocr_t* h) { return h ? h->num_classes : 0; }

int lumina_ocr_normalize(lumina_ocr_t* h, const uint8_t* img_dev, int n, int height, int width, int hp, int wp, const float scale[3],
                         const float shift[3], int layout_nchw, uint16_t* out_dev, void* stream) {
    if (!h || !img_dev || !out_dev || hp < height || wp < width) return locr_fail(h, "normalize", "bad arguments");
    BIND(h);
    hipError_t e = normalize_launch(img_dev, out_dev, n, height, width, hp, wp, height, width, scale, shift, layout_nchw, (hipStream_t)stream);
    return e == hipSuccess ? 0 : locr_fail(h, "normalize", hipGetErrorString(e));
}

int lumina_ocr_det_forward(lumina_ocr_t* h, const uint8_t* pages_dev, int batch, int height, int width, int hp, int wp, uint16_t* prob_dev,
                           void* stream) {
    if (!h || !pages_dev || !prob_dev) return locr_fail(h, "det_forward", "null argument");
    BIND(h);
    API_TRY return eng_det_forward(h, pages_dev, batch, height, width, hp, wp, prob_dev, (hipStream_t)stream); API_CATCH(h)
}

int lumina_ocr_det_postprocess(lumina_ocr_t* h, const uint16_t* prob_dev, int batch, int hp, int wp, int valid_h, int valid_w, float thresh,
                               float box_thresh, float unclip_ratio, int min_size, int max_boxes, int32_t* boxes_dev, float* scores_dev,
                               int32_t* counts_dev, void* stream) {
    if (!h || !prob_dev || !boxes_dev || !scores_dev || !counts_dev) return locr_fail(h, "det_postprocess", "null argument");
    if (batch <= 0 || max_boxes <= 0 || valid_h > hp || valid_w > wp) return locr_fail(h, "det_postprocess", "bad dimensions");
    BIND(h);
    API_TRY
    // pages are processed in groups that bound the workspace (~66 MB per A4 page: worst-case run list + row-extreme segments and hull scratch
    // of max_boxes page-high candidates)
    const int group = h->post_group;
    for (int b0 = 0; b0 < batch; b0 += group) {
        const int nb = batch - b0 < group ? batch - b0 : group;
        const size_t need = dbpost_workspace_bytes(nb, hp, wp, max_boxes);
        if (eng_ws_reserve(h, need)) return 1;
        DbPostParams p{};
        p.prob = prob_dev + (size_t)b0 * hp * wp; p.B = nb; p.Hp = hp; p.Wp = wp; p.valid_h = valid_h; p.valid_w = valid_w;
        p.thresh = thresh; p.box_thresh = box_thresh; p.unclip_ratio = unclip_ratio; p.min_size = min_size; p.max_boxes = max_boxes;
        p.boxes = boxes_dev + (size_t)b0 * max_boxes * 8; p.scores = scores_dev + (size_t)b0 * max_boxes; p.counts = counts_dev + b0;
        hipError_t e = dbpost_launch(p, h->ws, (hipStream_t)stream);
        if (e != hipSuccess) return locr_fail(h, "det_postprocess", hipGetErrorString(e));
    }
    return 0;
    API_CATCH(h)
}

int lumina_ocr_rec_crop(lumina_ocr_t* h, const uint8_t* pages_dev, int batch, int height, int width, const int32_t* quads_dev,
                        const int32_t* page_idx_dev, int n_crops, uint8_t* crops_dev, int32_t* widths_dev, void* stream) {
    if (!h || !pages_dev || !quads_dev || !page_idx_dev || !crops_dev || !widths_dev) return locr_fail(h, "rec_crop", "null argument");
    BIND(h);
    (void)batch;
    hipError_t e = rec_crop_launch(pages_dev, height, width, quads_dev, page_idx_dev, n_crops, crops_dev, widths_dev, (hipStream_t)stream);
    return e == hipSuccess ? 0 : locr_fail(h, "rec_crop", hipGetErrorString(e));
}

int lumina_ocr_rec_forward(lumina_ocr_t* h, const uint8_t* crops_dev, const int32_t* widths_dev, int n_crops, int32_t* idx_dev, float* prob_dev,
                           void* stream) {
    if (!h || !crops_dev || !idx_dev || !prob_dev) return locr_fail(h, "rec_forward", "null argument");
    BIND(h);
    API_TRY return eng_rec_forward(h, crops_dev, widths_dev, n_crops, idx_dev, prob_dev, (hipStream_t)stream); API_CATCH(h)
}

int lumina_ocr_load_svtr_weights(lumina_ocr_t* h, const void* blob, size_t nbytes) {
    if (!h || !blob) return locr_fail(h, "load_svtr_weights", "null argument");
    BIND(h);
    API_TRY return eng_load_svtr(h, blob, nbytes); API_CATCH(h)
}

int lumina_ocr_svtr_forward(lumina_ocr_t* h, const uint8_t* crops_dev, const int32_t* widths_dev, int n_crops, int32_t* idx_dev, float* prob_dev,
                            void* stream) {
    if (!h || !crops_dev || !idx_dev || !prob_dev) return locr_fail(h, "svtr_forward", "null argument");
    BIND(h);
    API_TRY return eng_svtr_forward(h, crops_dev, widths_dev, n_crops, idx_dev, prob_dev, (hipStream_t)stream); API_CATCH(h)
}

int lumina_ocr_ctc_decode(lumina_ocr_t* h, const int32_t* idx_dev, const float* prob_dev, int n, int32_t* text_dev, int32_t* len_dev,
                          float* score_dev, void* stream) {
    if (!h || !idx_dev || !prob_dev || !text_dev || !len_dev || !score_dev) return locr_fail(h, "ctc_decode", "null argument");
    BIND(h);
    if (n <= 0) return 0;
    hipError_t e = ctc_collapse_launch(idx_dev, prob_dev, text_dev, len_dev, score_dev, n, LUMINA_REC_T, (hipStream_t)stream);
    return e == hipSuccess ? 0 : locr_fail(h, "ctc_decode", hipGetErrorString(e));
}

int lumina_ocr_conv2d(lumina_ocr_t* h, const uint16_t* x_dev, int n, int height, int width, int cin, const uint16_t* w_host,
                      const float* bias_host, int cout, int ks, int stride, int act, const uint16_t* res_dev, uint16_t* y_dev, void* stream) {
    if (!h || !x_dev || !w_host || !bias_host || !y_dev) return locr_fail(h, "conv2d", "null argument");
    if (cout % 8 != 0) return locr_fail(h, "conv2d", "cout must be a multiple of 8");
    BIND(h);
    API_TRY
    lumina_ocr* eng = h;
    ConvLayer L;
    L.name = "conv2d"; L.ks = ks; L.stride = stride; L.cin = cin; L.cout = cout; L.act = act;
    if (!conv_pick_cfg(ks, stride, cin, cout, &L.cfg)) return locr_fail(h, "conv2d", "unsupported ks/stride/cin");
    std::vector<bf16_t> packed(conv_packed_weight_elems(cout, ks, cin, L.cfg.bn));
    pack_conv_weights(w_host, cout, ks, cin, L.cfg.bn, L.cfg.ck, packed.data(), L.cfg.nw == 6 ? 1 : 0);
    const int ntiles = (cout + L.cfg.bn - 1) / L.cfg.bn;
    std::vector<float> bias((size_t)ntiles * L.cfg.bn, 0.f);
    memcpy(bias.data(), bias_host, sizeof(float) * cout);
    void *dw = nullptr, *db = nullptr;
    if (hipMalloc(&dw, packed.size() * 2) != hipSuccess || hipMalloc(&db, bias.size() * 4) != hipSuccess) return locr_fail(h, "conv2d", "hipMalloc");
    (void)hipMemcpy(dw, packed.data(), packed.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, bias.data(), bias.size() * 4, hipMemcpyHostToDevice);
    L.wpk = static_cast<bf16_t*>(dw); L.bias = static_cast<float*>(db);
    // option conv2d_variant (parity tests): 1 = the LDS-DMA 16x32-tile kernel, 2 = the persistent ring kernel, for the layers
    // those kernels serve in the detector (3x3 / stride 1, >= 32 input channels in chunks of 16, output channels in tiles of 64)
    void* dw2 = nullptr;
    const bool big_ok = ks == 3 && stride == 1 && cin >= 32 && cin % 16 == 0 && cout % 64 == 0 && L.cfg.bn == 64;
    if (h->conv2d_variant > 0 && !big_ok) { (void)hipFree(dw); (void)hipFree(db); return locr_fail(h, "conv2d", "conv2d_variant 1/2 needs ks 3, stride 1, cin % 16 == 0, cin >= 32, cout % 64 == 0"); }
    if (h->conv2d_variant > 0) {
        L.cfg_big = L.cfg; L.cfg_big.nw = 6; L.cfg_big.ck = 16;
        std::vector<bf16_t> packed2(conv_packed_weight_elems(cout, ks, cin, 64));
        pack_conv_weights(w_host, cout, ks, cin, 64, 16, packed2.data(), 1);
        if (hipMalloc(&dw2, packed2.size() * 2) != hipSuccess) { (void)hipFree(dw); (void)hipFree(db); return locr_fail(h, "conv2d", "hipMalloc"); }
        (void)hipMemcpy(dw2, packed2.data(), packed2.size() * 2, hipMemcpyHostToDevice);
        L.wpk_big = static_cast<bf16_t*>(dw2);
        L.force_big = true;
    }
    const bool keep_ring = h->conv_ring;
    if (h->conv2d_variant == 1) h->conv_ring = false;
    if (h->conv2d_variant == 2) h->conv_ring = true;
    Tensor4 x; x.p = const_cast<bf16_t*>(x_dev); x.n = n; x.h = height; x.w = width; x.c = cin;
    Tensor4 y; y.p = y_dev; y.n = n; y.c = cout;
    y.h = (ks == 3) ? (height - 1) / stride + 1 : height / stride;
    y.w = (ks == 3) ? (width - 1) / stride + 1 : width / stride;
    Tensor4 r; r.p = const_cast<bf16_t*>(res_dev); r.n = n; r.h = y.h; r.w = y.w; r.c = cout;
    int rc = eng_run_conv(eng, L, x, &y, res_dev ? &r : nullptr, 0, OUT_NORMAL, 0, 0, 0, false, (hipStream_t)stream);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    h->conv_ring = keep_ring;
    (void)hipFree(dw); (void)hipFree(db);
    if (dw2) (void)hipFree(dw2);
    if (rc) return rc;
    return e == hipSuccess ? 0 : locr_fail(h, "conv2d sync", hipGetErrorString(e));
    API_CATCH(h)
}

int lumina_ocr_read_tap(lumina_ocr_t* h, const char* name, uint16_t* out_host, size_t capacity_elems, int dims[4]) {
    if (!h || !name || !dims) return 1;
    auto it = h->taps.find(name);
    if (it == h->taps.end()) return locr_fail(h, "read_tap: unknown tap", name);
    BIND(h);
    const Tensor4& t = it->second;
    dims[0] = t.n; dims[1] = t.h; dims[2] = t.w; dims[3] = t.c;
    if (!out_host) return 0;
    if (capacity_elems < t.elems()) return locr_fail(h, "read_tap", "buffer too small");
    if (hipDeviceSynchronize() != hipSuccess) return locr_fail(h, "read_tap", "sync failed");
    hipError_t e = hipMemcpy(out_host, t.p, t.elems() * sizeof(bf16_t), hipMemcpyDeviceToHost);
    return e == hipSuccess ? 0 : locr_fail(h, "read_tap", hipGetErrorString(e));
}

int lumina_ocr_conv_timing(lumina_ocr_t* h, double* total_ms, double* total_flops, int* launches) {
    if (!h || !total_ms || !total_flops || !launches) return 1;
    BIND(h);
    if (hipDeviceSynchronize() != hipSuccess) return locr_fail(h, "conv_timing", "sync failed");
    double ms = 0, fl = 0;
    for (size_t i = 0; i < h->conv_events.size(); ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, h->conv_events[i].first, h->conv_events[i].second) == hipSuccess) ms += t;
        fl += h->conv_flops[i];
        (void)hipEventDestroy(h->conv_events[i].first); (void)hipEventDestroy(h->conv_events[i].second);
    }
    *total_ms = ms; *total_flops = fl; *launches = (int)h->conv_events.size();
    h->conv_events.clear(); h->conv_flops.clear(); h->conv_bytes.clear(); h->conv_names.clear(); h->conv_kernels.clear();
    return 0;
}

static int get_coeffs(lumina_ocr* h, int in_size, int out_size, lumina_ocr::Coeffs* out) {
    auto key = std::make_pair(in_size, out_size);
    auto it = h->coeff_cache.find(key);
    if (it == h->coeff_cache.end()) {
        std::vector<int> bounds, kk;
        lumina_ocr::Coeffs c;
        lanczos_coeffs(in_size, out_size, &c.ksize, &bounds, &kk);
        c.bounds_host = std::make_shared<std::vector<int>>(bounds);
        if (hipMalloc(reinterpret_cast<void**>(&c.bounds), bounds.size() * 4) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&c.kk), kk.size() * 4) != hipSuccess)
            return locr_fail(h, "resize", "hipMalloc coefficient tables");
        (void)hipMemcpy(c.bounds, bounds.data(), bounds.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(c.kk, kk.data(), kk.size() * 4, hipMemcpyHostToDevice);
        h->owned.push_back(c.bounds); h->owned.push_back(c.kk);
        it = h->coeff_cache.emplace(key, c).first;
    }
    *out = it->second;
    return 0;
}

/* image_preprocessing.py:81-110 on the device: two-pass 8-bit LANCZOS, byte-exact with PIL. */
int lumina_ocr_resize_lanczos(lumina_ocr_t* h, const uint8_t* in_dev, int n, int height, int width, int channels, uint8_t* out_dev,
                              int out_h, int out_w, void* stream) {
    if (!h || !in_dev || !out_dev || n <= 0 || height <= 0 || width <= 0 || out_h <= 0 || out_w <= 0 || channels <= 0)
        return locr_fail(h, "resize_lanczos", "bad arguments");
    BIND(h);
    API_TRY
    hipStream_t st = (hipStream_t)stream;
    const uint8_t* src = in_dev;
    int cur_w = width;
    const bool need_h = out_w != width, need_v = out_h != height;
    if (!need_h && !need_v) {
        hipError_t e = hipMemcpyAsync(out_dev, in_dev, (size_t)n * height * width * channels, hipMemcpyDeviceToDevice, st);
        return e == hipSuccess ? 0 : locr_fail(h, "resize_lanczos", hipGetErrorString(e));
    }
    if (need_h) {
        lumina_ocr::Coeffs c;
        if (get_coeffs(h, width, out_w, &c)) return 1;
        uint8_t* dst = out_dev;
        if (need_v) {
            const size_t need = (size_t)n * height * out_w * channels;
            if (need > h->aux_cap) {
                if (hipDeviceSynchronize() != hipSuccess) return locr_fail(h, "resize_lanczos", "sync");
                if (h->aux) (void)hipFree(h->aux);
    for (int k = 0; k < 2; ++k) { if (h->jd_stage[k]) (void)hipHostFree(h->jd_stage[k]); if (h->jd_stage_ev[k]) (void)hipEventDestroy(h->jd_stage_ev[k]); }
                h->aux = nullptr; h->aux_cap = 0;
                if (hipMalloc(reinterpret_cast<void**>(&h->aux), need) != hipSuccess) return locr_fail(h, "resize_lanczos", "hipMalloc");
                h->aux_cap = need;
            }
            dst = h->aux;
        }
        hipError_t e = resample_launch(src, dst, c.bounds, c.kk, c.ksize, n, height, width, channels, out_w, 0, c.bounds_host->data(), st);
        if (e != hipSuccess) return locr_fail(h, "resize_lanczos", hipGetErrorString(e));
        src = dst; cur_w = out_w;
    }
    if (need_v) {
        lumina_ocr::Coeffs c;
        if (get_coeffs(h, height, out_h, &c)) return 1;
        hipError_t e = resample_launch(src, out_dev, c.bounds, c.kk, c.ksize, n, height, cur_w, channels, out_h, 1, c.bounds_host->data(), st);
        if (e != hipSuccess) return locr_fail(h, "resize_lanczos", hipGetErrorString(e));
    }
    return 0;
    API_CATCH(h)
}

/* image_preprocessing.py:132-158 (ImageEnhance.Contrast then .Sharpness) on RGB u8 [n,H,W,3]; tmp_dev: scratch, same size. */
int lumina_ocr_enhance(lumina_ocr_t* h, const uint8_t* img_dev, int n, int height, int width, float contrast, float sharpness,
                       uint8_t* tmp_dev, uint8_t* out_dev, void* stream) {
    if (!h || !img_dev || !tmp_dev || !out_dev || n <= 0) return locr_fail(h, "enhance", "bad arguments");
    BIND(h);
    if (n > h->sums_cap) {
        unsigned long long* s = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&s), sizeof(unsigned long long) * (size_t)n) != hipSuccess) return locr_fail(h, "enhance", "hipMalloc");
        h->owned.push_back(s);
        h->sums = s; h->sums_cap = n;
    }
    hipError_t e = enhance_launch(img_dev, tmp_dev, out_dev, h->sums, n, height, width, contrast, sharpness, (hipStream_t)stream);
    return e == hipSuccess ? 0 : locr_fail(h, "enhance", hipGetErrorString(e));
}

int lumina_ocr_binarize(lumina_ocr_t* h, const uint8_t* img_dev, int n, int height, int width, int adaptive, int threshold, uint8_t* out_dev, void* stream) {
    if (!h || !img_dev || !out_dev || n <= 0 || height <= 0 || width <= 0) return locr_fail(h, "binarize", "bad arguments");
    BIND(h);
    hipError_t e = binarize_launch(img_dev, out_dev, n, height, width, adaptive != 0, threshold, (hipStream_t)stream);
    return e == hipSuccess ? 0 : locr_fail(h, "binarize", hipGetErrorString(e));
}

int lumina_ocr_exif_transpose(lumina_ocr_t* h, const uint8_t* img_dev, int n, int height, int width, int orientation, uint8_t* out_dev, void* stream) {
    if (!h || !img_dev || !out_dev || n <= 0 || height <= 0 || width <= 0 || orientation < 1 || orientation > 8 || img_dev == out_dev)
        return locr_fail(h, "exif_transpose", "bad arguments (orientation 1..8, not in place)");
    BIND(h);
    hipError_t e = exif_transpose_launch(img_dev, out_dev, n, height, width, orientation, (hipStream_t)stream);
    return e == hipSuccess ? 0 : locr_fail(h, "exif_transpose", hipGetErrorString(e));
}

int lumina_ocr_grayscale(lumina_ocr_t* h, const uint8_t* img_dev, int n, int height, int width, uint8_t* out_dev, void* stream) {
    if (!h || !img_dev || !out_dev || n <= 0 || height <= 0 || width <= 0) return locr_fail(h, "grayscale", "bad arguments");
    BIND(h);
    hipError_t e = grayscale_launch(img_dev, out_dev, n, height, width, (hipStream_t)stream);
    return e == hipSuccess ? 0 : locr_fail(h, "grayscale", hipGetErrorString(e));
}

int lumina_ocr_denoise(lumina_ocr_t* h, const uint8_t* img_dev, int n, int height, int width, uint8_t* out_dev, void* stream) {
    if (!h || !img_dev || !out_dev || n <= 0 || height <= 0 || width <= 0 || img_dev == out_dev) return locr_fail(h, "denoise", "bad arguments (not in place)");
    BIND(h);
    hipError_t e = median3_launch(img_dev, out_dev, n, height, width, (hipStream_t)stream);
    return e == hipSuccess ? 0 : locr_fail(h, "denoise", hipGetErrorString(e));
}

int lumina_ocr_jpeg_encode(lumina_ocr_t* h, const uint8_t* pages_dev, int n, int height, int width, int quality, int optimize,
                           uint8_t* out_dev, size_t out_stride, int32_t* sizes_dev, void* stream) {
    if (!h || !pages_dev || !out_dev || !sizes_dev || n <= 0 || height <= 0 || width <= 0) return locr_fail(h, "jpeg_encode", "bad arguments");
    BIND(h);
    API_TRY
    if (eng_ws_reserve(h, jpeg_workspace_bytes(n, height, width))) return 1;
    JpegParams p{};
    p.rgb = pages_dev; p.n = n; p.height = height; p.width = width; p.quality = quality; p.optimize = optimize != 0; p.out = out_dev; p.out_stride = out_stride; p.sizes = sizes_dev;
    hipError_t e = jpeg_encode_launch(p, h->ws, (hipStream_t)stream);
    return e == hipSuccess ? 0 : locr_fail(h, "jpeg_encode", hipGetErrorString(e));
    API_CATCH(h)
}

int lumina_ocr_jpeg_probe(const uint8_t* file, size_t size, int info[6]) {
    if (!file || !info) return -1;
    JdInfo i{};
    const int rc = jpegdec_probe(file, size, &i);
    info[0] = i.width; info[1] = i.height; info[2] = i.ncomp; info[3] = i.hs; info[4] = i.vs; info[5] = i.restart;
    return rc;
}

int lumina_ocr_jpeg_decode(lumina_ocr_t* h, const uint8_t* const* files, const size_t* sizes, int n, int height, int width, uint8_t* out_dev, int* status,
                           void* stream) {
    if (!h || !files || !sizes || !out_dev || !status || n <= 0 || height <= 0 || width <= 0) return locr_fail(h, "jpeg_decode", "bad arguments");
    BIND(h);
    API_TRY
    return jpegdec_run(h, files, sizes, n, height, width, out_dev, status, (hipStream_t)stream);
    API_CATCH(h)
}

int lumina_ocr_jpeg_decode_async(lumina_ocr_t* h, const uint8_t* const* files, const size_t* sizes, int n, int height, int width, uint8_t* out_dev,
                                 int* status_pinned, int passes, void* stream) {
    if (!h || !files || !sizes || !out_dev || !status_pinned || n <= 0 || height <= 0 || width <= 0 || passes < 2) return locr_fail(h, "jpeg_decode_async", "bad arguments");
    BIND(h);
    API_TRY
    return jpegdec_run(h, files, sizes, n, height, width, out_dev, status_pinned, (hipStream_t)stream, passes);
    API_CATCH(h)
}

int lumina_ocr_jpeg_last_passes(const lumina_ocr_t* h) { return h ? h->jd_last_passes : 0; }

int lumina_ocr_jpeg_coefficients(lumina_ocr_t* h, const uint8_t* pages_dev, int n, int height, int width, int quality, int16_t* coefs_dev,
                                 void* stream) {
    if (!h || !pages_dev || !coefs_dev || n <= 0 || height <= 0 || width <= 0) return locr_fail(h, "jpeg_coefficients", "bad arguments");
    BIND(h);
    hipError_t e = jpeg_coefficients_launch(pages_dev, n, height, width, quality, coefs_dev, (hipStream_t)stream);
    return e == hipSuccess ? 0 : locr_fail(h, "jpeg_coefficients", hipGetErrorString(e));
}

static int deskew_tables(lumina_ocr* h) {
    if (h->dk_trig && h->dk_wtab) return 0;
    float trig[360];
    std::vector<short> wtab(32 * 32 * 16);
    deskew_trig_table(trig);
    deskew_weight_table(wtab.data());
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, sizeof(trig)) != hipSuccess || hipMalloc(&b, wtab.size() * 2) != hipSuccess) return locr_fail(h, "deskew", "hipMalloc tables");
    (void)hipMemcpy(a, trig, sizeof(trig), hipMemcpyHostToDevice);
    (void)hipMemcpy(b, wtab.data(), wtab.size() * 2, hipMemcpyHostToDevice);
    h->owned.push_back(a); h->owned.push_back(b);
    h->dk_trig = static_cast<float*>(a); h->dk_wtab = static_cast<short*>(b);
    return 0;
}

int lumina_ocr_deskew(lumina_ocr_t* h, const uint8_t* pages_dev, int n, int height, int width, uint8_t* out_dev, double* rot_dev, int32_t* info_dev,
                      uint8_t* edges_dev, int32_t* segs_dev, int32_t* nsegs_dev, void* stream) {
    if (!h || !pages_dev || !rot_dev || !info_dev || n <= 0 || height <= 0 || width <= 0) return locr_fail(h, "deskew", "bad arguments");
    if ((segs_dev == nullptr) != (nsegs_dev == nullptr)) return locr_fail(h, "deskew", "segs_dev and nsegs_dev go together");
    BIND(h);
    API_TRY
    if (deskew_tables(h)) return 1;
    // pages are processed in groups that bound the workspace (~16 bytes per pixel + the accumulators)
    const int group = h->post_group;
    for (int b0 = 0; b0 < n; b0 += group) {
        const int nb = n - b0 < group ? n - b0 : group;
        if (eng_ws_reserve(h, deskew_workspace_bytes(nb, height, width))) return 1;
        const size_t px = (size_t)height * width;
        DeskewParams p{};
        p.rgb = pages_dev + (size_t)b0 * px * 3; p.out = out_dev ? out_dev + (size_t)b0 * px * 3 : nullptr;
        p.B = nb; p.H = height; p.W = width; p.trig = h->dk_trig; p.wtab = h->dk_wtab;
        p.rot = rot_dev + (size_t)b0 * 3; p.info = info_dev + (size_t)b0 * 2;
        p.edges_out = edges_dev ? edges_dev + (size_t)b0 * px : nullptr;
        p.segs_out = segs_dev ? segs_dev + (size_t)b0 * DESKEW_MAX_PEAKS * DESKEW_SEG_PER_PEAK * 4 : nullptr;
        p.nsegs_out = nsegs_dev ? nsegs_dev + (size_t)b0 * DESKEW_MAX_PEAKS : nullptr;
        hipError_t e = deskew_launch(p, h->ws, (hipStream_t)stream);
        if (e != hipSuccess) return locr_fail(h, "deskew", hipGetErrorString(e));
    }
    return 0;
    API_CATCH(h)
}

int lumina_ocr_deskew_warp(lumina_ocr_t* h, const uint8_t* pages_dev, int n, int height, int width, const double* rot_dev, uint8_t* out_dev, void* stream) {
    if (!h || !pages_dev || !rot_dev || !out_dev || n <= 0 || height <= 0 || width <= 0) return locr_fail(h, "deskew_warp", "bad arguments");
    BIND(h);
    API_TRY
    if (deskew_tables(h)) return 1;
    hipError_t e = deskew_warp_launch(pages_dev, out_dev, rot_dev, h->dk_wtab, n, height, width, (hipStream_t)stream);
    return e == hipSuccess ? 0 : locr_fail(h, "deskew_warp", hipGetErrorString(e));
    API_CATCH(h)
}

int lumina_ocr_svtr_num_classes(const lumina_ocr_t* h) { return h ? h->svtr.num_classes : 0; }
int lumina_ocr_svtr_dtype(const lumina_ocr_t* h) { return h ? h->svtr.dtype : 0; }

int lumina_ocr_conv_timing_detail(lumina_ocr_t* h, char* buf, size_t cap) {
    if (!h || !buf || cap == 0) return 1;
    BIND(h);
    if (hipDeviceSynchronize() != hipSuccess) return locr_fail(h, "conv_timing_detail", "sync failed");
    std::string out;
    for (size_t i = 0; i < h->conv_events.size(); ++i) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, h->conv_events[i].first, h->conv_events[i].second);
        char line[256];
        snprintf(line, sizeof(line), "%s %s %.4f %.4f %.4f\n", h->conv_names[i].c_str(), h->conv_kernels[i].c_str(), t, h->conv_flops[i] * 1e-9, h->conv_bytes[i] * 1e-6);
        out += line;
        (void)hipEventDestroy(h->conv_events[i].first); (void)hipEventDestroy(h->conv_events[i].second);
    }
    h->conv_events.clear(); h->conv_flops.clear(); h->conv_bytes.clear(); h->conv_names.clear(); h->conv_kernels.clear();
    snprintf(buf, cap, "%s", out.c_str());
    return 0;
}

}  // extern "C"
