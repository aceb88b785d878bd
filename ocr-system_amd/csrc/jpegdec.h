// Baseline JPEG decoder on the device: the pixel work behind the reference's
//   Image.open(path) / Image.open(io.BytesIO(bytes))        (ImagePreprocessor.load_image / load_image_bytes,
//   /root/reference/backend/utils/image_preprocessing.py:57-75; .jpg / .jpeg inputs of ocr_service.py:695-731)
// byte-identical to Pillow / libjpeg-turbo (oracle/csrc/jpegdec_oracle.c is the pinned CPU restatement).
#pragma once
#include <cstddef>

#include "common.h"

struct JdInfo { int width, height, ncomp, hs, vs, restart; };
// Host only: 0 = a file the device decoder handles (sequential Huffman baseline, 8 bit, 1 or 3 components in one interleaved scan,
// luma sampling 1x1 / 2x1 / 2x2), -1 = not a JPEG / truncated, -2 = valid but outside that subset (the caller decodes it with Pillow,
// as the reference does).
int jpegdec_probe(const uint8_t* file, size_t n, JdInfo* info);

struct lumina_ocr;
// files: HOST pointers (the bytes arrive from disk or the network); all n files must be height x width.  out: device RGB u8
// [n][height][width][3] (grey-scale files: the grey value on all three channels).  status: HOST int [n], 0 ok / -1 corrupt /
// -2 unsupported / -4 size mismatch (such a page's pixels are not written).  Synchronises the stream once (the parallel
// Huffman decode iterates to a fixed point and reads back one flag per pass group).
// async_passes > 0: nothing is synchronised — that many synchronisation passes are enqueued, `status` must be PINNED host memory and is
// valid once the stream has run (-5 = the passes did not reach the fixed point: decode that batch again with the synchronous form).
int jpegdec_run(lumina_ocr* eng, const uint8_t* const* files, const size_t* sizes, int n, int height, int width, uint8_t* out_dev, int* status,
                hipStream_t st, int async_passes = 0);
