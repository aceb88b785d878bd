// Baseline JPEG encoder on the device: the reference's hand-off of the processed page,
//   image.save(buffer, format='JPEG', quality=q, optimize=True)
// (/root/reference/backend/utils/image_preprocessing.py:526-538; consumer backend/utils/file_manager.py:283-287),
// byte-identical to Pillow / libjpeg-turbo (oracle/csrc/jpeg_oracle.c is the pinned CPU restatement).
#pragma once
#include "common.h"

struct JpegParams {
    const uint8_t* rgb;   // [n, height, width, 3] u8, device
    int n, height, width, quality;
    int optimize;         // 1: per-page optimal Huffman tables (PIL optimize=True); 0: the Annex K.3 typical tables
    uint8_t* out;         // [n, out_stride] device: complete JFIF files
    size_t out_stride;
    int32_t* sizes;       // [n] device: file length in bytes, or -(needed length) when it exceeds out_stride
};

// Device workspace bytes for encoding n pages of height x width.
size_t jpeg_workspace_bytes(int n, int height, int width);
// Enqueue the whole encoder (colour transform + DCT + quantisation, symbol statistics, optimal Huffman tables, entropy coding,
// byte stuffing, header assembly) on `st`; nothing is synchronised.
hipError_t jpeg_encode_launch(const JpegParams& p, void* workspace, hipStream_t st);
// Test hook: quantised coefficients only -> coefs [n][mcus][6][64] int16 in zig-zag order (dummy blocks resolved).
hipError_t jpeg_coefficients_launch(const uint8_t* rgb, int n, int height, int width, int quality, int16_t* coefs, hipStream_t st);
