/* lumina_ocr.h — C ABI of the MI355X-native det+rec OCR engine (liblumina_ocr.so).
 *
 * This is the drop-in boundary underneath the reference's OCR-provider interface.  The reference
 * has no FFI for this path: its provider is a Python module whose engine slot is one call,
 *   Azure : /root/reference/backend/services/ocr_service.py:213-246  (_analyze_with_azure)
 *   Paddle: /root/reference/backend/services/ocr_service_paddleocr_backup.py:285 (pipeline.predict)
 * and whose result is reshaped by ocr_service.py:248-376 (_extract_layout_boxes).  The entry
 * points below are what a ctypes binding inside that slot calls (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; it never throws and never
 *     aborts.  lumina_ocr_last_error() returns a description of the last failure on the handle
 *     (the reference's error convention is "errors are data": ocr_service.py:464-475).
 *   - all *_dev pointers are device pointers owned by the caller (e.g. torch-ROCm tensors,
 *     tensor.data_ptr()); `stream` is a hipStream_t passed as void* (NULL = default stream).
 *     Work is enqueued asynchronously on that stream unless stated otherwise.
 *   - a handle is bound to one device and must not be used from two threads at once (the
 *     reference serialises pages with a Semaphore(1): ocr_service.py:157, :404).  Every entry
 *     runs with the handle's device current and puts the calling thread's previous device back
 *     before it returns.
 *   - images are uint8 HWC RGB; activations are bf16 NHWC; the probability map is bf16.
 */
#ifndef LUMINA_OCR_H
#define LUMINA_OCR_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lumina_ocr lumina_ocr_t;

#define LUMINA_REC_H 32
#define LUMINA_REC_W 320
#define LUMINA_REC_T 80
#define LUMINA_MAX_BOXES 1000 /* DB max_candidates; box buffers are [B][LUMINA_MAX_BOXES][8] */

/* lifecycle — replaces OCRService._ensure_client_initialized (ocr_service.py:166-207) */
int lumina_ocr_create(int device, lumina_ocr_t** out);
void lumina_ocr_destroy(lumina_ocr_t* h);
const char* lumina_ocr_last_error(const lumina_ocr_t* h);
const char* lumina_ocr_version(void);
/* options: "det_sub_batch", "rec_sub_batch", "post_group", "tail_group", "keep_taps", "time_convs"; developer A/B switches (results are
 * bit-identical either way): "fuse_head", "fuse_pool", "fuse_stem", "fuse_mb", "fpn_multi", "conv_ring", "ring_orient", "conv_big_min",
 * "blocked_layout" (experiment: fails loudly when a blocked tensor would reach a kernel other than the ring kernel);
 * "svtr_f16" (storage type of the next SVTR load), "conv2d_variant" (kernel choice of lumina_ocr_conv2d, parity tests) */
int lumina_ocr_set_option(lumina_ocr_t* h, const char* key, int value);

/* weights: "LOCW" container (ocr-system_amd/lumina_ocr/arch.py write_blob), host memory.
 * Replaces PaddleOCRVL(...) model construction (ocr_service_paddleocr_backup.py:204-253). */
int lumina_ocr_load_det_weights(lumina_ocr_t* h, const void* blob, size_t nbytes);
int lumina_ocr_load_rec_weights(lumina_ocr_t* h, const void* blob, size_t nbytes);
int lumina_ocr_num_classes(const lumina_ocr_t* h);

/* image -> tensor: xn = bf16(u8 * scale[c] + shift[c]), zero outside (valid_h, valid_w).
 * layout_nchw = 1 writes [N,3,Hp,Wp], 0 writes [N,Hp,Wp,3]. */
int lumina_ocr_normalize(lumina_ocr_t* h, const uint8_t* img_dev, int n, int height, int width, int hp, int wp,
                         const float scale[3], const float shift[3], int layout_nchw, uint16_t* out_dev, void* stream);

/* DBNet (ResNet18_vd + DBFPN + DBHead): pages_dev uint8 [B,H,W,3] -> prob_dev bf16 [B,Hp,Wp],
 * Hp, Wp multiples of 32, >= H, W (the page is zero-padded in normalised space).
 * This is the arithmetic of the engine slot (ocr_service.py:231-238). */
int lumina_ocr_det_forward(lumina_ocr_t* h, const uint8_t* pages_dev, int batch, int height, int width, int hp, int wp,
                           uint16_t* prob_dev, void* stream);

/* DB post-process on device: prob -> integer quads (TL,TR,BR,BL) in page pixels.
 * boxes_dev int32 [B][max_boxes][8], scores_dev float [B][max_boxes], counts_dev int32 [B].
 * Output shape follows ocr_postprocessor.py:24 / ocr_service.py:295-301 (4 points / flat 8). */
int lumina_ocr_det_postprocess(lumina_ocr_t* h, const uint16_t* prob_dev, int batch, int hp, int wp, int valid_h, int valid_w,
                               float thresh, float box_thresh, float unclip_ratio, int min_size, int max_boxes,
                               int32_t* boxes_dev, float* scores_dev, int32_t* counts_dev, void* stream);

/* Recognition crops: for crop i, sample quad quads_dev[i] (8 int32) of page page_idx_dev[i] into
 * crops_dev uint8 [n][32][320][3]; widths_dev int32 [n] receives the valid width. */
int lumina_ocr_rec_crop(lumina_ocr_t* h, const uint8_t* pages_dev, int batch, int height, int width, const int32_t* quads_dev,
                        const int32_t* page_idx_dev, int n_crops, uint8_t* crops_dev, int32_t* widths_dev, void* stream);

/* CRNN (MobileNetV3-small x0.5 + 2xBiLSTM(96) + FC) with the CTC FC, arg-max and soft-max fused:
 * crops uint8 [n][32][320][3] (+ optional valid widths) -> idx int32 [n][80], prob float [n][80]. */
int lumina_ocr_rec_forward(lumina_ocr_t* h, const uint8_t* crops_dev, const int32_t* widths_dev, int n_crops, int32_t* idx_dev,
                           float* prob_dev, void* stream);

/* CTC greedy decode: collapse repeats, drop blank(0). text int32 [n][80] (class ids, -1 padded),
 * len int32 [n], score float [n] (mean max-prob of kept steps) — the (text, confidence) pair of
 * ocr_postprocessor.py:73-93 and the "confidence" key of ocr_service.py:298. */
int lumina_ocr_ctc_decode(lumina_ocr_t* h, const int32_t* idx_dev, const float* prob_dev, int n, int32_t* text_dev, int32_t* len_dev,
                          float* score_dev, void* stream);

/* Reference pre-processing on the device, byte-exact with the reference's PIL path:
 * resize_if_needed (image_preprocessing.py:81-110): 8-bit two-pass LANCZOS to (out_h, out_w); any channel count. */
int lumina_ocr_resize_lanczos(lumina_ocr_t* h, const uint8_t* in_dev, int n, int height, int width, int channels, uint8_t* out_dev,
                              int out_h, int out_w, void* stream);
/* enhance_contrast(contrast) then enhance_sharpness(sharpness) (image_preprocessing.py:132-158, :234-240) on RGB
 * uint8 [n,H,W,3]; tmp_dev is a scratch buffer of the same size. */
int lumina_ocr_enhance(lumina_ocr_t* h, const uint8_t* img_dev, int n, int height, int width, float contrast, float sharpness,
                       uint8_t* tmp_dev, uint8_t* out_dev, void* stream);

/* binarize (image_preprocessing.py:175-185) / adaptive_binarize (:462-494; settings.PREPROCESSING_APPLY_BINARIZE, off by default,
 * replaces contrast + sharpness when on: :613-622).  adaptive = 0: PIL L > threshold (what the reference's adaptive_binarize degrades
 * to without OpenCV, :473-475; byte-exact with it); adaptive = 1: cv2.adaptiveThreshold(GAUSSIAN_C, BINARY, blockSize 11, C 2)
 * restated ("parity unpinned").  RGB uint8 [n,H,W,3] in; the 0 / 255 value on all three channels out. */
int lumina_ocr_binarize(lumina_ocr_t* h, const uint8_t* img_dev, int n, int height, int width, int adaptive, int threshold, uint8_t* out_dev,
                        void* stream);

/* optimize_for_ocr's optional steps (image_preprocessing.py:225-231; both off by default in the reference's callers):
 * convert_to_grayscale (:167-169) = PIL convert('L'), written to all three channels of the page; denoise (:160-165) = PIL
 * MedianFilter(3), per channel, image edge-replicated.  uint8 [n,H,W,3] in and out (denoise: not in place); byte-exact with Pillow. */
/* auto_orient (image_preprocessing.py:171-173, first step of optimize_for_ocr :213) = PIL ImageOps.exif_transpose: EXIF orientation 1..8
 * applied on the device (the companion of lumina_ocr_jpeg_decode for camera / phone files); out_dev is [n][width][height][3] for 5..8. */
int lumina_ocr_exif_transpose(lumina_ocr_t* h, const uint8_t* img_dev, int n, int height, int width, int orientation, uint8_t* out_dev, void* stream);
int lumina_ocr_grayscale(lumina_ocr_t* h, const uint8_t* img_dev, int n, int height, int width, uint8_t* out_dev, void* stream);
int lumina_ocr_denoise(lumina_ocr_t* h, const uint8_t* img_dev, int n, int height, int width, uint8_t* out_dev, void* stream);

/* deskew (image_preprocessing.py:372-460; on by default in the provider: backend/config.py:85, ocr_service.py:412-417): Canny(50, 150)
 * -> Hough line segments (threshold 100, min length 100, max gap 10) -> median of the segment angles folded into [-45, 45] ->
 * unchanged below 0.5 / above 45 degrees -> cubic affine warp about (W / 2, H / 2) with replicated borders.  The reference does
 * this with OpenCV (absent offline: "parity unpinned"); the arithmetic is defined by oracle/csrc/deskew_oracle.c and reproduced
 * bit for bit.  pages_dev uint8 [n,H,W,3]; out_dev (same shape, may be NULL: estimate only) receives the rotated page or a copy;
 * rot_dev double [n][3] = sin, cos of the angle and a flag (0 no line found, 1 below 0.5 degrees, 2 above 45 degrees, 3 rotated);
 * info_dev int32 [n][2] = segments found, Hough peaks walked.  Optional parity hooks (NULL to skip): edges_dev uint8 [n,H,W]
 * (0 / 255), segs_dev int32 [n][512][8][4] (x1, y1, x2, y2) with nsegs_dev int32 [n][512] segments per peak slot.  Asynchronous. */
int lumina_ocr_deskew(lumina_ocr_t* h, const uint8_t* pages_dev, int n, int height, int width, uint8_t* out_dev, double* rot_dev,
                      int32_t* info_dev, uint8_t* edges_dev, int32_t* segs_dev, int32_t* nsegs_dev, void* stream);
/* The warp alone, for given (sin, cos, flag) triples: cv2.getRotationMatrix2D((w // 2, h // 2), angle, 1.0) + cv2.warpAffine(...,
 * flags=INTER_CUBIC, borderMode=BORDER_REPLICATE) (image_preprocessing.py:446-454) in OpenCV's fixed-point arithmetic. */
int lumina_ocr_deskew_warp(lumina_ocr_t* h, const uint8_t* pages_dev, int n, int height, int width, const double* rot_dev, uint8_t* out_dev,
                           void* stream);

/* Second recogniser family (BASELINE configs[4]: "SVTR-base multilingual (Hindi dict), fp16 MFMA"): same slot and the same outputs as
 * lumina_ocr_load_rec_weights / lumina_ocr_rec_forward (the `rec` model of the engine call, ocr_service_paddleocr_backup.py:232-238,
 * :285), with an SVTR backbone (patch embedding, local / global mixing blocks, CTC head) instead of CRNN.  Blob: LOCW with the
 * `svtr.*` tensors of lumina_ocr/arch.py make_svtr_weights; the optional f32 tensor `svtr.config` = [dim0, dim1, dim2, depth0,
 * depth1, depth2, heads0, heads1, heads2, local_blocks, out_channels, dtype] selects the variant (absent: SVTR-Tiny, bf16;
 * Base = 128/256/384, 3/6/9, 4/8/12, 8 local blocks) and the storage / MFMA type (0 bf16, 1 fp16: v_mfma_f32_32x32x16_f16,
 * activations and weights stored as IEEE half); option "svtr_f16" (0 / 1, -1 = as the blob says) overrides the type at load time. */
int lumina_ocr_load_svtr_weights(lumina_ocr_t* h, const void* blob, size_t nbytes);
int lumina_ocr_svtr_forward(lumina_ocr_t* h, const uint8_t* crops_dev, const int32_t* widths_dev, int n_crops, int32_t* idx_dev, float* prob_dev,
                            void* stream);
int lumina_ocr_svtr_num_classes(const lumina_ocr_t* h);   /* class count of the loaded SVTR head (lumina_ocr_num_classes: the CRNN's) */
int lumina_ocr_svtr_dtype(const lumina_ocr_t* h);         /* storage / MFMA type of the loaded SVTR model: 0 bf16, 1 fp16 */

/* JPEG hand-off of the processed page: replaces image.save(buffer, format='JPEG', quality=q, optimize=True) inside
 * ImagePreprocessor.compress_for_azure (backend/utils/image_preprocessing.py:526-538; the bytes become OCROutput.processed_image_bytes,
 * ocr_service.py:459, saved by backend/utils/file_manager.py:283-287).  Byte-identical to Pillow's output: JFIF 1.01, YCbCr 4:2:0,
 * integer DCT, optimised Huffman tables.  pages_dev: uint8 [n, height, width, 3]; out_dev: uint8 [n, out_stride] receives one complete
 * file per page; sizes_dev[i] = its length, or a negative number (-(length needed), saturated) when it does not fit out_stride — the
 * caller's cue to retry at a lower quality, as the reference's loop does when a file exceeds its 2 MB target.  optimize = 0 writes
 * the typical Huffman tables of T.81 Annex K.3 instead (the reference's size probe `image.save(buffer, 'JPEG', quality=min_quality)`,
 * :548).  Asynchronous on stream. */
int lumina_ocr_jpeg_encode(lumina_ocr_t* h, const uint8_t* pages_dev, int n, int height, int width, int quality, int optimize,
                           uint8_t* out_dev, size_t out_stride, int32_t* sizes_dev, void* stream);
/* Parity hook for the encoder's first half: quantised DCT coefficients, int16 [n][ceil(w/16)*ceil(h/16)][6][64] in zig-zag order
 * (4 luma, Cb, Cr blocks per MCU; dummy edge blocks resolved). */
/* JPEG decode on the device — the pixel work of the reference's Image.open(path) / Image.open(BytesIO(bytes)) for .jpg inputs
 * (ImagePreprocessor.load_image / load_image_bytes, image_preprocessing.py:57-75), byte-identical to Pillow's decode.
 * lumina_ocr_jpeg_probe (host only, no handle): info = {width, height, components, luma h, luma v, restart interval}; returns 0 for a
 * file the device decodes (sequential Huffman baseline, 8 bit, grey or YCbCr in one interleaved scan, 4:4:4 / 4:2:2 / 4:2:0),
 * -1 corrupt / not a JPEG, -2 valid but outside that subset (progressive, CMYK, ...): decode those with Pillow, as the reference does.
 * lumina_ocr_jpeg_decode: files / sizes are HOST arrays of n file images, all height x width; out_dev uint8 [n][height][width][3]
 * (a grey file: its value on all three channels); status HOST int [n] (0 ok, -1, -2, -4 = other size: that page is not written).
 * Synchronises `stream` (the parallel Huffman decode iterates to a fixed point). */
int lumina_ocr_jpeg_probe(const uint8_t* file, size_t size, int info[6]);
int lumina_ocr_jpeg_decode(lumina_ocr_t* h, const uint8_t* const* files, const size_t* sizes, int n, int height, int width, uint8_t* out_dev,
                           int* status, void* stream);
/* The same without any host synchronisation (a pipeline decodes the next batch while the device still works on the previous one): `passes`
 * synchronisation passes are enqueued blindly (12 suffice for 1 KB chunks on busy A4 pages: 8 needed), `status_pinned` must be pinned host
 * memory and is valid once `stream` has run; -5 = the passes did not reach the fixed point (decode that batch again with the form above). */
int lumina_ocr_jpeg_decode_async(lumina_ocr_t* h, const uint8_t* const* files, const size_t* sizes, int n, int height, int width, uint8_t* out_dev,
                                 int* status_pinned, int passes, void* stream);
/* synchronisation passes over the chunk decoders the last lumina_ocr_jpeg_decode call needed (diagnostic) */
int lumina_ocr_jpeg_last_passes(const lumina_ocr_t* h);

int lumina_ocr_jpeg_coefficients(lumina_ocr_t* h, const uint8_t* pages_dev, int n, int height, int width, int quality, int16_t* coefs_dev,
                                 void* stream);

/* ---- kernel-level entry points (parity tests, benchmarks) ---- */
/* Generic NHWC bf16 convolution through the MFMA implicit-GEMM kernel.  w_host: OHWI bf16 bits
 * [cout][ks][ks][cin], bias_host float [cout]; ks/stride in {1/1, 2/2, 3/1, 3/2}; cin % 16 == 0,
 * cout % 8 == 0; act: 0 none, 1 relu, 2 hswish, 3 hsigmoid, 4 sigmoid; res_dev optional [N,Ho,Wo,cout].
 * Synchronous (packs and uploads the weights, runs, waits). */
int lumina_ocr_conv2d(lumina_ocr_t* h, const uint16_t* x_dev, int n, int height, int width, int cin, const uint16_t* w_host,
                      const float* bias_host, int cout, int ks, int stride, int act, const uint16_t* res_dev, uint16_t* y_dev,
                      void* stream);
/* Copy an intermediate activation of the last det/rec forward (option keep_taps=1) to host memory.
 * dims receives n,h,w,c; returns non-zero when the tap is unknown or the buffer too small. */
int lumina_ocr_read_tap(lumina_ocr_t* h, const char* name, uint16_t* out_host, size_t capacity_elems, int dims[4]);
/* Sum of conv-kernel device time (ms) and algorithmic FLOPs since the last call (option time_convs=1). */
int lumina_ocr_conv_timing(lumina_ocr_t* h, double* total_ms, double* total_flops, int* launches);
/* Same, per launch, as text lines "name kernel ms gflop\n" written into buf (truncated to cap). Clears the records. */
int lumina_ocr_conv_timing_detail(lumina_ocr_t* h, char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
