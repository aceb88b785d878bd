#pragma once
#include <vector>

#include "common.h"

// Pillow-exact LANCZOS coefficient tables for one axis (host).
void lanczos_coeffs(int in_size, int out_size, int* ksize_out, std::vector<int>* bounds, std::vector<int>* kk);
// One resampling pass over [N,H,W,C] u8: axis 0 -> [N,H,out_len,C], axis 1 -> [N,out_len,W,C].
// bounds_host: host copy of the bounds table (vertical pass: picks the tiled kernel when the row spans fit its LDS tile) or null.
hipError_t resample_launch(const uint8_t* in, uint8_t* out, const int* bounds_dev, const int* kk_dev, int ksize, int N, int H, int W, int C,
                           int out_len, int axis, const int* bounds_host, hipStream_t st);
// contrast (mean-gray blend) then sharpness (3x3 smooth blend) on RGB u8 [N,H,W,3]; tmp = scratch of the same size.
hipError_t enhance_launch(const uint8_t* img, uint8_t* tmp, uint8_t* out, unsigned long long* sums_dev, int N, int H, int W, float contrast,
                          float sharpness, hipStream_t st);
// binarisation (image_preprocessing.py:175-185 / :462-494): adaptive = 0: L > threshold; 1: Gaussian 11x11 adaptive threshold, C = 2.
// RGB u8 [N,H,W,3] in, the 0 / 255 value on all three channels out.
hipError_t binarize_launch(const uint8_t* img, uint8_t* out, int N, int H, int W, int adaptive, int threshold, hipStream_t st);
// optimize_for_ocr's optional steps (image_preprocessing.py:160-169, :225-231): PIL convert('L') replicated to three channels, and
// PIL MedianFilter(3) (edge-replicated); RGB u8 [N,H,W,3] in and out, byte-exact with Pillow (tests/golden/preprocess_vectors.npz)
hipError_t grayscale_launch(const uint8_t* img, uint8_t* out, int N, int H, int W, hipStream_t st);
hipError_t median3_launch(const uint8_t* img, uint8_t* out, int N, int H, int W, hipStream_t st);
// ImageOps.exif_transpose (auto_orient): EXIF orientation 1..8 applied to RGB u8 [N,H,W,3]; out is [N,W,H,3] for orientations 5..8
hipError_t exif_transpose_launch(const uint8_t* img, uint8_t* out, int N, int H, int W, int orientation, hipStream_t st);
