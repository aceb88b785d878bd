#pragma once
#include "common.h"

struct DbPostParams {
    const bf16_t* prob;  // [B][Hp][Wp]
    int B, Hp, Wp, valid_h, valid_w;
    float thresh, box_thresh, unclip_ratio;
    int min_size, max_boxes;
    int* boxes;     // [B][max_boxes][8]
    float* scores;  // [B][max_boxes]
    int* counts;    // [B]
};

size_t dbpost_workspace_bytes(int B, int Hp, int Wp, int max_boxes);
hipError_t dbpost_launch(const DbPostParams& p, void* workspace, hipStream_t st);
hipError_t rec_crop_launch(const uint8_t* pages, int H, int W, const int* quads, const int* page_idx, int n, uint8_t* crops, int* widths,
                           hipStream_t st);
