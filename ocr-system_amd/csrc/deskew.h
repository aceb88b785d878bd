#pragma once
#include "common.h"

// De-skew of a batch of pages on the device (deskew.hip; definition: oracle/csrc/deskew_oracle.c).
struct DeskewParams {
    const uint8_t* rgb;   // [B][H][W][3]
    uint8_t* out;         // [B][H][W][3] rotated page (flag 3) or a copy; nullptr: estimate only
    int B, H, W;
    const float* trig;    // device, [180][2] = (float)cos / sin(n pi / 180)     (deskew_trig_table)
    const short* wtab;    // device, [32 * 32][16] fixed-point bicubic weights    (deskew_weight_table)
    double* rot;          // device, [B][3]: sin, cos of the median folded angle; flag 0 no segment, 1 below 0.5 deg, 2 above 45 deg, 3 rotated
    int* info;            // device, [B][2]: segments, peaks visited
    uint8_t* edges_out;   // optional parity hooks: Canny edge map [B][H][W] (0 / 255),
    int* segs_out;        //   segments [B][512][8][4] (x1, y1, x2, y2) and
    int* nsegs_out;       //   their count per peak slot [B][512] (slot order is arbitrary, the set is not)
};
constexpr int DESKEW_MAX_PEAKS = 512, DESKEW_SEG_PER_PEAK = 8;

void deskew_trig_table(float* tab /* [360] */);
void deskew_weight_table(short* wtab /* [32 * 32 * 16] */);
size_t deskew_workspace_bytes(int B, int H, int W);
hipError_t deskew_launch(const DeskewParams& p, void* workspace, hipStream_t st);
hipError_t deskew_warp_launch(const uint8_t* rgb, uint8_t* out, const double* rot, const short* wtab, int B, int H, int W, hipStream_t st);
