/* TEST INFRASTRUCTURE (oracle): sequential CPU restatement of the baseline JPEG encoder behind the reference's
 *   image.save(buffer, format='JPEG', quality=q, optimize=True)
 * (/root/reference/backend/utils/image_preprocessing.py:526-538, :343; consumer: backend/utils/file_manager.py:283-287).
 * The algorithm lives in a third-party dependency of the reference (Pillow 12.2.0 -> libjpeg-turbo, libjpeg API 6.2), not in
 * /root/reference; this file restates its published algorithm (ITU-T T.81 + the IJG integer pipeline):
 *   JFIF 1.01 header, two quantisation tables (Annex K scaled by the IJG quality rule), YCbCr 4:2:0, 16-bit fixed-point
 *   colour conversion, 2x2 box chroma down-sampling with the alternating 1,2 rounding bias, edge replication inside partial
 *   blocks and zero-AC dummy blocks beyond them, the "islow" integer forward DCT, round-half-away quantisation, optimised
 *   Huffman tables (two-pass, Annex K.2 with the IJG tie-breaking and 16-bit length limiting), byte stuffing, EOI.
 * Parity is PINNED: tests/test_golden_jpeg.py compares its output byte for byte with Pillow's on the committed fixtures and on
 * seeded images generated in the build container (tools/make_golden.py writes the digests).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const uint8_t ZZ[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                               41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                               15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
static const uint8_t STD_LUMA[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57,
                                     69, 56, 14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55,
                                     64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t STD_CHROMA[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                       99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                       99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

void oracle_jpeg_quant_tables(int quality, uint16_t luma[64], uint16_t chroma[64]) {
    if (quality <= 0) quality = 1;
    if (quality > 100) quality = 100;
    const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    for (int i = 0; i < 64; ++i) {
        long a = ((long)STD_LUMA[i] * scale + 50L) / 100L, b = ((long)STD_CHROMA[i] * scale + 50L) / 100L;
        luma[i] = (uint16_t)(a <= 0 ? 1 : (a > 255 ? 255 : a));
        chroma[i] = (uint16_t)(b <= 0 ? 1 : (b > 255 ? 255 : b));
    }
}

/* ---- colour conversion: 16-bit fixed point ---- */
#define FIXC(x) ((int32_t)((x) * 65536.0 + 0.5))
static void rgb_to_ycc(const uint8_t* rgb, int n, uint8_t* y, uint8_t* cb, uint8_t* cr) {
    const int32_t half = 1 << 15, off = 128 << 16;
    for (int i = 0; i < n; ++i) {
        const int32_t r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
        y[i] = (uint8_t)((FIXC(0.29900) * r + FIXC(0.58700) * g + FIXC(0.11400) * b + half) >> 16);
        cb[i] = (uint8_t)((-FIXC(0.16874) * r - FIXC(0.33126) * g + FIXC(0.50000) * b + off + half - 1) >> 16);
        cr[i] = (uint8_t)((FIXC(0.50000) * r - FIXC(0.41869) * g - FIXC(0.08131) * b + off + half - 1) >> 16);
    }
}

/* ---- forward DCT, "islow" (Loeffler-Ligtenberg-Moschytz, 13-bit constants, 2 extra bits after pass 1) ---- */
#define CB 13
#define P1 2
#define F_0_298631336 2446
#define F_0_390180644 3196
#define F_0_541196100 4433
#define F_0_765366865 6270
#define F_0_899976223 7373
#define F_1_175875602 9633
#define F_1_501321110 12299
#define F_1_847759065 15137
#define F_1_961570560 16069
#define F_2_053119869 16819
#define F_2_562915447 20995
#define F_3_072711026 25172
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))
static void fdct_islow(int16_t* d) {
    for (int pass = 0; pass < 2; ++pass) {
        const int st = pass == 0 ? 1 : 8, adv = pass == 0 ? 8 : 1;
        for (int k = 0; k < 8; ++k) {
            int16_t* p = d + k * adv;
            const int32_t t0 = p[0] + p[7 * st], t7 = p[0] - p[7 * st], t1 = p[st] + p[6 * st], t6 = p[st] - p[6 * st];
            const int32_t t2 = p[2 * st] + p[5 * st], t5 = p[2 * st] - p[5 * st], t3 = p[3 * st] + p[4 * st], t4 = p[3 * st] - p[4 * st];
            const int32_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
            if (pass == 0) {
                p[0] = (int16_t)((t10 + t11) << P1);
                p[4 * st] = (int16_t)((t10 - t11) << P1);
            } else {
                p[0] = (int16_t)DESCALE(t10 + t11, P1);
                p[4 * st] = (int16_t)DESCALE(t10 - t11, P1);
            }
            const int sh = pass == 0 ? CB - P1 : CB + P1;
            int32_t z1 = (t12 + t13) * F_0_541196100;
            p[2 * st] = (int16_t)DESCALE(z1 + t13 * F_0_765366865, sh);
            p[6 * st] = (int16_t)DESCALE(z1 + t12 * (-F_1_847759065), sh);
            z1 = t4 + t7;
            int32_t z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
            const int32_t z5 = (z3 + z4) * F_1_175875602;
            int32_t a4 = t4 * F_0_298631336, a5 = t5 * F_2_053119869, a6 = t6 * F_3_072711026, a7 = t7 * F_1_501321110;
            z1 = z1 * (-F_0_899976223); z2 = z2 * (-F_2_562915447);
            z3 = z3 * (-F_1_961570560); z4 = z4 * (-F_0_390180644);
            z3 += z5; z4 += z5;
            p[7 * st] = (int16_t)DESCALE(a4 + z1 + z3, sh);
            p[5 * st] = (int16_t)DESCALE(a5 + z2 + z4, sh);
            p[3 * st] = (int16_t)DESCALE(a6 + z2 + z3, sh);
            p[st] = (int16_t)DESCALE(a7 + z1 + z4, sh);
        }
    }
}

/* one 8x8 block of a padded plane -> quantised coefficients in natural order */
static void block_coefs(const uint8_t* plane, int pitch, int bx, int by, const uint16_t* q, int16_t* out) {
    int16_t w[64];
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) w[r * 8 + c] = (int16_t)((int)plane[(size_t)(by * 8 + r) * pitch + bx * 8 + c] - 128);
    fdct_islow(w);
    for (int i = 0; i < 64; ++i) {
        const int32_t qv = (int32_t)q[i] << 3;
        int32_t t = w[i];
        if (t < 0) { t = -t; t += qv >> 1; t = t >= qv ? t / qv : 0; t = -t; }
        else { t += qv >> 1; t = t >= qv ? t / qv : 0; }
        out[i] = (int16_t)t;
    }
}

/* Quantised coefficient blocks of one image in MCU scan order: per MCU 4 Y + Cb + Cr blocks of 64 (natural order).
 * Returns the number of MCUs; coefs must hold mcus * 6 * 64 int16. */
int oracle_jpeg_coefficients(const uint8_t* rgb, int w, int h, int quality, int16_t* coefs) {
    uint16_t ql[64], qc[64];
    oracle_jpeg_quant_tables(quality, ql, qc);
    const int mx = (w + 15) / 16, my = (h + 15) / 16, pw = mx * 16, ph = my * 16, cw = pw / 2, chh = ph / 2;
    uint8_t* Y = (uint8_t*)malloc((size_t)pw * ph);
    uint8_t* CbF = (uint8_t*)malloc((size_t)pw * ph);
    uint8_t* CrF = (uint8_t*)malloc((size_t)pw * ph);
    uint8_t* Cb = (uint8_t*)malloc((size_t)cw * chh);
    uint8_t* Cr = (uint8_t*)malloc((size_t)cw * chh);
    for (int y = 0; y < h; ++y) {
        rgb_to_ycc(rgb + (size_t)y * w * 3, w, Y + (size_t)y * pw, CbF + (size_t)y * pw, CrF + (size_t)y * pw);
        for (int x = w; x < pw; ++x) {  /* right edge: replicate the last converted sample */
            Y[(size_t)y * pw + x] = Y[(size_t)y * pw + w - 1];
            CbF[(size_t)y * pw + x] = CbF[(size_t)y * pw + w - 1];
            CrF[(size_t)y * pw + x] = CrF[(size_t)y * pw + w - 1];
        }
    }
    /* bottom edge.  Luma: replicate the last row.  Chroma: an odd last row is paired with a copy of itself, the image is
     * down-sampled, and it is the last DOWN-SAMPLED row that is replicated to the MCU boundary. */
    for (int y = h; y < ph; ++y) memcpy(Y + (size_t)y * pw, Y + (size_t)(h - 1) * pw, pw);
    if (h & 1) {
        memcpy(CbF + (size_t)h * pw, CbF + (size_t)(h - 1) * pw, pw);
        memcpy(CrF + (size_t)h * pw, CrF + (size_t)(h - 1) * pw, pw);
    }
    const int crows = (h + 1) / 2;
    for (int y = 0; y < crows; ++y) {  /* 2x2 box filter, bias 1,2,1,2 along the row */
        int bias = 1;
        for (int x = 0; x < cw; ++x) {
            const uint8_t *a = CbF + (size_t)(2 * y) * pw + 2 * x, *b = CrF + (size_t)(2 * y) * pw + 2 * x;
            Cb[(size_t)y * cw + x] = (uint8_t)((a[0] + a[1] + a[pw] + a[pw + 1] + bias) >> 2);
            Cr[(size_t)y * cw + x] = (uint8_t)((b[0] + b[1] + b[pw] + b[pw + 1] + bias) >> 2);
            bias ^= 3;
        }
    }
    for (int y = crows; y < chh; ++y) {
        memcpy(Cb + (size_t)y * cw, Cb + (size_t)(crows - 1) * cw, cw);
        memcpy(Cr + (size_t)y * cw, Cr + (size_t)(crows - 1) * cw, cw);
    }
    /* real blocks per component (beyond them: zero-AC dummy blocks whose DC repeats the previous block of the MCU) */
    const int ybw = (w + 7) / 8, ybh = (h + 7) / 8;
    for (int m = 0; m < mx * my; ++m) {
        const int mcx = m % mx, mcy = m / mx;
        int16_t* mc = coefs + (size_t)m * 6 * 64;
        for (int v = 0; v < 2; ++v)
            for (int hh = 0; hh < 2; ++hh) {
                int16_t* blk = mc + (v * 2 + hh) * 64;
                const int bx = mcx * 2 + hh, by = mcy * 2 + v;
                if (by < ybh && bx < ybw) block_coefs(Y, pw, bx, by, ql, blk);
                else {
                    memset(blk, 0, 128);
                    /* right-edge dummy: DC of the block to its left; bottom dummy row: DC of the last block of the row above */
                    blk[0] = (by < ybh) ? (blk - 64)[0] : (mc + (v * 2 - 1) * 64)[0];
                }
            }
        block_coefs(Cb, cw, mcx, mcy, qc, mc + 4 * 64);
        block_coefs(Cr, cw, mcx, mcy, qc, mc + 5 * 64);
    }
    free(Y); free(CbF); free(CrF); free(Cb); free(Cr);
    return mx * my;
}

/* ---- Huffman ---- */
/* ITU-T T.81 Annex K.3 typical tables: what libjpeg writes when optimize is off (the reference's size probe, :548) */
static const uint8_t STD_BITS_DC_L[17] = {
    0x00, 0x00, 0x01, 0x05, 0x01, 0x01, 0x01, 0x01, 0x01, 0x01, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00,
};
static const uint8_t STD_VALS_DC_L[12] = {
    0x00, 0x01, 0x02, 0x03, 0x04, 0x05, 0x06, 0x07, 0x08, 0x09, 0x0a, 0x0b,
};
static const uint8_t STD_BITS_AC_L[17] = {
    0x00, 0x00, 0x02, 0x01, 0x03, 0x03, 0x02, 0x04, 0x03, 0x05, 0x05, 0x04, 0x04, 0x00, 0x00, 0x01, 0x7d,
};
static const uint8_t STD_VALS_AC_L[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08,
    0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
    0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
    0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
    0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa,
};
static const uint8_t STD_BITS_DC_C[17] = {
    0x00, 0x00, 0x03, 0x01, 0x01, 0x01, 0x01, 0x01, 0x01, 0x01, 0x01, 0x01, 0x00, 0x00, 0x00, 0x00, 0x00,
};
static const uint8_t STD_VALS_DC_C[12] = {
    0x00, 0x01, 0x02, 0x03, 0x04, 0x05, 0x06, 0x07, 0x08, 0x09, 0x0a, 0x0b,
};
static const uint8_t STD_BITS_AC_C[17] = {
    0x00, 0x00, 0x02, 0x01, 0x02, 0x04, 0x04, 0x03, 0x04, 0x07, 0x05, 0x04, 0x04, 0x00, 0x01, 0x02, 0x77,
};
static const uint8_t STD_VALS_AC_C[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91,
    0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
    0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
    0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa,
};

typedef struct { uint8_t bits[17]; uint8_t val[256]; uint16_t code[256]; uint8_t size[256]; } HT;

static int nbits_of(int v) { int n = 0; if (v < 0) v = -v; while (v) { ++n; v >>= 1; } return n; }

/* optimal code lengths from symbol frequencies (T.81 K.2 with the IJG conventions: pseudo-symbol 256 reserves the all-ones
 * code; ties go to the larger symbol value; lengths are limited to 16) */
void oracle_jpeg_gen_table(const long* freq_in, uint8_t bits_out[17], uint8_t* val_out, int* nval) {
    long freq[257];
    int codesize[257], others[257], bits[33];
    memcpy(freq, freq_in, sizeof(long) * 256);
    freq[256] = 1;
    memset(codesize, 0, sizeof(codesize));
    memset(bits, 0, sizeof(bits));
    for (int i = 0; i < 257; ++i) others[i] = -1;
    for (;;) {
        int c1 = -1, c2 = -1;
        long v = 1000000000L;
        for (int i = 0; i <= 256; ++i) if (freq[i] && freq[i] <= v) { v = freq[i]; c1 = i; }
        v = 1000000000L;
        for (int i = 0; i <= 256; ++i) if (freq[i] && freq[i] <= v && i != c1) { v = freq[i]; c2 = i; }
        if (c2 < 0) break;
        freq[c1] += freq[c2]; freq[c2] = 0;
        codesize[c1]++;
        while (others[c1] >= 0) { c1 = others[c1]; codesize[c1]++; }
        others[c1] = c2;
        codesize[c2]++;
        while (others[c2] >= 0) { c2 = others[c2]; codesize[c2]++; }
    }
    for (int i = 0; i <= 256; ++i) if (codesize[i]) bits[codesize[i] > 32 ? 32 : codesize[i]]++;
    for (int i = 32; i > 16; --i)
        while (bits[i] > 0) {
            int j = i - 2;
            while (bits[j] == 0) --j;
            bits[i] -= 2; bits[i - 1]++; bits[j + 1] += 2; bits[j]--;
        }
    int i = 16;
    while (bits[i] == 0) --i;
    bits[i]--;  /* the pseudo-symbol's code is never emitted */
    bits_out[0] = 0;
    for (int k = 1; k <= 16; ++k) bits_out[k] = (uint8_t)bits[k];
    int p = 0;
    for (int len = 1; len <= 32; ++len)
        for (int s = 0; s < 256; ++s) if (codesize[s] == len) val_out[p++] = (uint8_t)s;
    *nval = p;
}

static void derive(HT* t, int nval) {
    int k = 0, code = 0;
    memset(t->size, 0, sizeof(t->size));
    for (int len = 1; len <= 16; ++len) {
        for (int i = 0; i < t->bits[len]; ++i, ++k) { t->code[t->val[k]] = (uint16_t)code++; t->size[t->val[k]] = (uint8_t)len; }
        code <<= 1;
    }
    (void)nval;
}

typedef struct { uint8_t* p; size_t n, cap; uint64_t acc; int nacc; } BW;
static void put_byte(BW* b, int v) { if (b->n < b->cap) b->p[b->n] = (uint8_t)v; b->n++; }
static void put_bits(BW* b, unsigned code, int size) {
    if (!size) return;
    b->acc = (b->acc << size) | (code & ((1u << size) - 1));
    b->nacc += size;
    while (b->nacc >= 8) {
        const int c = (int)((b->acc >> (b->nacc - 8)) & 0xff);
        put_byte(b, c);
        if (c == 0xff) put_byte(b, 0);
        b->nacc -= 8;
    }
}
static void put_marker(BW* b, int m) { put_byte(b, 0xff); put_byte(b, m); }
static void put_u16(BW* b, int v) { put_byte(b, (v >> 8) & 0xff); put_byte(b, v & 0xff); }

/* Full encoder: RGB (interleaved, w x h) -> JFIF byte stream identical to Pillow's save(format='JPEG', quality, optimize=True).
 * Returns the stream length (may exceed cap: then only cap bytes were written). */
size_t oracle_jpeg_encode_ex(const uint8_t* rgb, int w, int h, int quality, int optimize, uint8_t* out, size_t cap);
size_t oracle_jpeg_encode(const uint8_t* rgb, int w, int h, int quality, uint8_t* out, size_t cap) {
    return oracle_jpeg_encode_ex(rgb, w, h, quality, 1, out, cap);
}
/* optimize != 0: two-pass optimal tables; optimize == 0: the Annex K.3 tables */
size_t oracle_jpeg_encode_ex(const uint8_t* rgb, int w, int h, int quality, int optimize, uint8_t* out, size_t cap) {
    const int mx = (w + 15) / 16, my = (h + 15) / 16, mcus = mx * my;
    int16_t* coefs = (int16_t*)malloc((size_t)mcus * 6 * 64 * sizeof(int16_t));
    oracle_jpeg_coefficients(rgb, w, h, quality, coefs);
    /* pass 1: symbol statistics.  tables: 0 = luma, 1 = chroma */
    long dcf[2][256], acf[2][256];
    memset(dcf, 0, sizeof(dcf)); memset(acf, 0, sizeof(acf));
    int last[3] = {0, 0, 0};
    for (int m = 0; m < mcus; ++m)
        for (int b = 0; b < 6; ++b) {
            const int16_t* blk = coefs + ((size_t)m * 6 + b) * 64;
            const int comp = b < 4 ? 0 : b - 3, tb = comp ? 1 : 0;
            dcf[tb][nbits_of(blk[0] - last[comp])]++;
            last[comp] = blk[0];
            int r = 0;
            for (int k = 1; k < 64; ++k) {
                const int v = blk[ZZ[k]];
                if (v == 0) { ++r; continue; }
                while (r > 15) { acf[tb][0xF0]++; r -= 16; }
                acf[tb][(r << 4) + nbits_of(v)]++;
                r = 0;
            }
            if (r > 0) acf[tb][0]++;
        }
    HT dc[2], ac[2];
    int ndc[2], nac[2];
    for (int t = 0; t < 2; ++t) {
        if (optimize) {
            oracle_jpeg_gen_table(dcf[t], dc[t].bits, dc[t].val, &ndc[t]);
            oracle_jpeg_gen_table(acf[t], ac[t].bits, ac[t].val, &nac[t]);
        } else {
            memcpy(dc[t].bits, t ? STD_BITS_DC_C : STD_BITS_DC_L, 17); memcpy(dc[t].val, t ? STD_VALS_DC_C : STD_VALS_DC_L, 12); ndc[t] = 12;
            memcpy(ac[t].bits, t ? STD_BITS_AC_C : STD_BITS_AC_L, 17); memcpy(ac[t].val, t ? STD_VALS_AC_C : STD_VALS_AC_L, 162); nac[t] = 162;
        }
        derive(&dc[t], ndc[t]); derive(&ac[t], nac[t]);
    }
    /* headers */
    BW bw = {out, 0, cap, 0, 0};
    uint16_t ql[64], qc[64];
    oracle_jpeg_quant_tables(quality, ql, qc);
    put_marker(&bw, 0xD8);
    put_marker(&bw, 0xE0); put_u16(&bw, 16);
    { const uint8_t j[14] = {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0}; for (int i = 0; i < 14; ++i) put_byte(&bw, j[i]); }
    for (int t = 0; t < 2; ++t) {
        put_marker(&bw, 0xDB); put_u16(&bw, 67); put_byte(&bw, t);
        for (int i = 0; i < 64; ++i) put_byte(&bw, (t ? qc : ql)[ZZ[i]]);
    }
    put_marker(&bw, 0xC0); put_u16(&bw, 17); put_byte(&bw, 8); put_u16(&bw, h); put_u16(&bw, w); put_byte(&bw, 3);
    put_byte(&bw, 1); put_byte(&bw, 0x22); put_byte(&bw, 0);
    put_byte(&bw, 2); put_byte(&bw, 0x11); put_byte(&bw, 1);
    put_byte(&bw, 3); put_byte(&bw, 0x11); put_byte(&bw, 1);
    for (int t = 0; t < 2; ++t)
        for (int a = 0; a < 2; ++a) {
            const HT* T = a ? &ac[t] : &dc[t];
            const int n = a ? nac[t] : ndc[t];
            put_marker(&bw, 0xC4); put_u16(&bw, 2 + 1 + 16 + n); put_byte(&bw, (a << 4) | t);
            for (int i = 1; i <= 16; ++i) put_byte(&bw, T->bits[i]);
            for (int i = 0; i < n; ++i) put_byte(&bw, T->val[i]);
        }
    put_marker(&bw, 0xDA); put_u16(&bw, 12); put_byte(&bw, 3);
    put_byte(&bw, 1); put_byte(&bw, 0x00); put_byte(&bw, 2); put_byte(&bw, 0x11); put_byte(&bw, 3); put_byte(&bw, 0x11);
    put_byte(&bw, 0); put_byte(&bw, 63); put_byte(&bw, 0);
    /* pass 2: entropy-coded segment */
    last[0] = last[1] = last[2] = 0;
    for (int m = 0; m < mcus; ++m)
        for (int b = 0; b < 6; ++b) {
            const int16_t* blk = coefs + ((size_t)m * 6 + b) * 64;
            const int comp = b < 4 ? 0 : b - 3, tb = comp ? 1 : 0;
            int d = blk[0] - last[comp];
            last[comp] = blk[0];
            int nb = nbits_of(d);
            put_bits(&bw, dc[tb].code[nb], dc[tb].size[nb]);
            if (nb) put_bits(&bw, (unsigned)(d < 0 ? d - 1 : d), nb);
            int r = 0;
            for (int k = 1; k < 64; ++k) {
                const int v = blk[ZZ[k]];
                if (v == 0) { ++r; continue; }
                while (r > 15) { put_bits(&bw, ac[tb].code[0xF0], ac[tb].size[0xF0]); r -= 16; }
                nb = nbits_of(v);
                put_bits(&bw, ac[tb].code[(r << 4) + nb], ac[tb].size[(r << 4) + nb]);
                put_bits(&bw, (unsigned)(v < 0 ? v - 1 : v), nb);
                r = 0;
            }
            if (r > 0) put_bits(&bw, ac[tb].code[0], ac[tb].size[0]);
        }
    put_bits(&bw, 0x7F, 7);  /* pad the last byte with ones */
    bw.nacc = 0;
    put_marker(&bw, 0xD9);
    free(coefs);
    return bw.n;
}
