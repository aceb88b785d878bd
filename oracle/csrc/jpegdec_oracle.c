/* TEST INFRASTRUCTURE (oracle): sequential CPU restatement of the baseline JPEG DECODER behind the reference's
 *   Image.open(path) / Image.open(io.BytesIO(bytes))      (ImagePreprocessor.load_image / load_image_bytes,
 *   /root/reference/backend/utils/image_preprocessing.py:57-75)
 * for .jpg / .jpeg inputs (ocr_service.py:695-731 dispatches them to process_image).  The algorithm lives in a third-party
 * dependency of the reference (Pillow 12.2.0 -> libjpeg-turbo, libjpeg API 6.2), not in /root/reference; this file restates
 * its published algorithm for what Pillow's decoder does by default:
 *   sequential baseline DCT (SOF0; SOF1 with 8-bit samples reads the same), Huffman entropy coding (ITU-T T.81 Annex F.2.2)
 *   with restart intervals, 1 or 3 components in ONE interleaved scan, sampling factors 1x1 / 2x1 / 2x2 on the first component,
 *   de-quantisation + the "islow" integer inverse DCT (13-bit constants, 2 extra bits after pass 1, 10-bit range-limit mask),
 *   "fancy" (triangle-filter) chroma up-sampling h2v1 / h2v2 with edge replication and libjpeg-turbo's alternating rounding
 *   biases (plain replication when the down-sampled width is <= 2), 16-bit fixed-point YCbCr -> RGB.
 * Anything else (progressive, arithmetic coding, CMYK, 12-bit, several scans, 4:4:0 / 4:1:1) is reported as unsupported (-2):
 * the provider then decodes on the host with Pillow, exactly as the reference does.
 * Parity is PINNED: tests/test_golden_jpegdec.py compares the output byte for byte with Pillow's decode of the same files
 * (generated with Pillow at test time from seeded images) and with the digests tools/make_golden.py wrote.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const uint8_t DZZ[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

typedef struct {
    int width, height, ncomp;
    int hs[3], vs[3], tq[3], td[3], ta[3];   /* sampling factors, quant table, DC / AC Huffman table per component */
    int hmax, vmax, mcux, mcuy, restart;
    uint16_t q[4][64];                       /* natural order */
    uint8_t bits[2][4][17], vals[2][4][256]; /* [dc/ac][table] */
    int have_q[4], have_h[2][4];
    size_t scan_off;                         /* first byte of the entropy-coded segment */
} JpegHeader;

/* Parses the markers up to SOS.  0 = a file this decoder handles, -1 = not a JPEG / truncated, -2 = valid but unsupported. */
static int parse_header(const uint8_t* f, size_t n, JpegHeader* h) {
    memset(h, 0, sizeof(*h));
    if (n < 4 || f[0] != 0xFF || f[1] != 0xD8) return -1;
    size_t p = 2;
    int seen_sof = 0;
    for (;;) {
        if (p + 4 > n) return -1;
        if (f[p] != 0xFF) return -1;
        while (p < n && f[p] == 0xFF) ++p;             /* fill bytes */
        if (p >= n) return -1;
        const int m = f[p++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return -1;
        if (p + 2 > n) return -1;
        const size_t len = ((size_t)f[p] << 8) | f[p + 1];
        if (len < 2 || p + len > n) return -1;
        const uint8_t* s = f + p + 2;
        const size_t sl = len - 2;
        if (m == 0xDB) {                               /* DQT */
            size_t i = 0;
            while (i < sl) {
                const int pq = s[i] >> 4, t = s[i] & 15;
                ++i;
                if (t > 3 || pq > 1 || i + (pq ? 128 : 64) > sl) return -1;
                for (int k = 0; k < 64; ++k) {
                    h->q[t][DZZ[k]] = pq ? (uint16_t)((s[i] << 8) | s[i + 1]) : s[i];
                    i += pq ? 2 : 1;
                }
                h->have_q[t] = 1;
            }
        } else if (m == 0xC4) {                        /* DHT */
            size_t i = 0;
            while (i < sl) {
                const int tc = s[i] >> 4, t = s[i] & 15;
                ++i;
                if (tc > 1 || t > 3 || i + 16 > sl) return -1;
                int cnt = 0;
                h->bits[tc][t][0] = 0;
                for (int k = 1; k <= 16; ++k) { h->bits[tc][t][k] = s[i + k - 1]; cnt += s[i + k - 1]; }
                i += 16;
                if (cnt > 256 || i + cnt > sl) return -1;
                memcpy(h->vals[tc][t], s + i, (size_t)cnt);
                i += cnt;
                h->have_h[tc][t] = 1;
            }
        } else if (m == 0xC0 || m == 0xC1) {           /* SOF0 / SOF1 (extended sequential, Huffman) */
            if (sl < 6 || seen_sof) return -1;
            if (s[0] != 8) return -2;
            h->height = (s[1] << 8) | s[2]; h->width = (s[3] << 8) | s[4]; h->ncomp = s[5];
            if (h->height == 0 || h->width == 0) return -2;
            if (h->ncomp != 1 && h->ncomp != 3) return -2;
            if (sl < (size_t)(6 + 3 * h->ncomp)) return -1;
            for (int c = 0; c < h->ncomp; ++c) {
                if (s[6 + 3 * c] != c + 1) return -2;  /* JFIF component ids 1, 2, 3 = Y, Cb, Cr (anything else: let Pillow decide) */
                h->hs[c] = s[7 + 3 * c] >> 4; h->vs[c] = s[7 + 3 * c] & 15; h->tq[c] = s[8 + 3 * c];
                if (h->tq[c] > 3) return -1;
            }
            seen_sof = 1;
        } else if ((m >= 0xC2 && m <= 0xCF) && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return -2;                                 /* progressive, lossless, arithmetic ... */
        } else if (m == 0xDD) {                        /* DRI */
            if (sl < 2) return -1;
            h->restart = (s[0] << 8) | s[1];
        } else if (m == 0xEE) {                        /* Adobe: a transform flag other than YCbCr changes the colour model */
            if (sl >= 12 && memcmp(s, "Adobe", 5) == 0 && s[11] != 1 && h->ncomp != 1) return -2;
        } else if (m == 0xDA) {                        /* SOS */
            if (!seen_sof || sl < 1 || s[0] != h->ncomp || sl < (size_t)(1 + 2 * h->ncomp + 3)) return seen_sof ? -2 : -1;
            for (int c = 0; c < h->ncomp; ++c) {
                if (s[1 + 2 * c] != c + 1) return -2;
                h->td[c] = s[2 + 2 * c] >> 4; h->ta[c] = s[2 + 2 * c] & 15;
                if (h->td[c] > 3 || h->ta[c] > 3 || !h->have_h[0][h->td[c]] || !h->have_h[1][h->ta[c]] || !h->have_q[h->tq[c]]) return -1;
            }
            h->scan_off = p + len;
            break;
        }
        p += len;
    }
    if (h->ncomp == 1) { h->hs[0] = h->vs[0] = 1; }    /* a single-component scan is never interleaved: one block per MCU */
    else {
        if (h->hs[1] != 1 || h->vs[1] != 1 || h->hs[2] != 1 || h->vs[2] != 1) return -2;
        if (!((h->hs[0] == 1 && h->vs[0] == 1) || (h->hs[0] == 2 && h->vs[0] == 1) || (h->hs[0] == 2 && h->vs[0] == 2))) return -2;
    }
    h->hmax = h->hs[0]; h->vmax = h->vs[0];
    h->mcux = (h->width + 8 * h->hmax - 1) / (8 * h->hmax);
    h->mcuy = (h->height + 8 * h->vmax - 1) / (8 * h->vmax);
    return 0;
}

/* ---- Huffman tables in the canonical form of T.81 F.2.2.3 ---- */
typedef struct { int32_t maxcode[18]; int32_t valptr[17]; uint8_t vals[256]; } Huff;
static int build_huff(const uint8_t bits[17], const uint8_t* vals, Huff* t) {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        t->valptr[l] = k - code;
        if (bits[l]) {
            k += bits[l]; code += bits[l];
            if (code > (1 << l)) return -1;
            t->maxcode[l] = code - 1;
        } else t->maxcode[l] = -1;
        code <<= 1;
    }
    t->maxcode[17] = 0x7fffffff;
    memcpy(t->vals, vals, 256);
    return 0;
}

typedef struct { const uint8_t* p; const uint8_t* end; uint32_t acc; int cnt; int marker; } Bits;
static void fill(Bits* b) {
    while (b->cnt <= 24) {
        int v = 0;
        if (!b->marker && b->p < b->end) {
            v = *b->p;
            if (v == 0xFF) {
                if (b->p + 1 < b->end && b->p[1] == 0) b->p += 2;      /* stuffed zero */
                else { b->marker = 1; v = 0; }                          /* a marker: feed zeros until the caller deals with it */
            } else b->p += 1;
        }
        b->acc |= (uint32_t)v << (24 - b->cnt);
        b->cnt += 8;
    }
}
static int getbits(Bits* b, int n) {
    if (n == 0) return 0;
    fill(b);
    const int v = (int)(b->acc >> (32 - n));
    b->acc <<= n; b->cnt -= n;
    return v;
}
static int decode_sym(Bits* b, const Huff* t) {
    fill(b);
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | (int)(b->acc >> 31);
        b->acc <<= 1; b->cnt -= 1;
        if (code <= t->maxcode[l] && t->maxcode[l] >= 0) return t->vals[(t->valptr[l] + code) & 255];
        if (b->cnt == 0) fill(b);
    }
    return -1;
}
static int extend(int v, int s) { return s == 0 ? 0 : (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v); }

/* ---- inverse DCT "islow" with de-quantisation (jidctint.c), one block: coef natural order -> 8x8 samples ---- */
#define CB 13
#define P1 2
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))
static uint8_t idct_limit(int32_t x) {   /* range_limit[(x) & RANGE_MASK] of the IDCT: == clamp(x + 128) for -512 <= x < 512 */
    const int i = (int)(x & 1023);
    return (uint8_t)(i < 128 ? i + 128 : (i < 512 ? 255 : (i < 896 ? 0 : i - 896)));
}
static void idct_islow(const int16_t* c, const uint16_t* q, uint8_t* out, int stride) {
    int32_t ws[64];
    for (int col = 0; col < 8; ++col) {
#define D(r) ((int32_t)c[8 * (r) + col] * (int32_t)q[8 * (r) + col])
        int32_t z2 = D(2), z3 = D(6);
        int32_t z1 = (z2 + z3) * 4433;
        int32_t tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;
        z2 = D(0); z3 = D(4);
        int32_t tmp0 = (z2 + z3) * (1 << CB), tmp1 = (z2 - z3) * (1 << CB);
        const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = D(7); tmp1 = D(5); tmp2 = D(3); tmp3 = D(1);
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int32_t z4 = tmp1 + tmp3;
        const int32_t z5 = (z3 + z4) * 9633;
        tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
        z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        ws[col] = DESCALE(tmp10 + tmp3, CB - P1); ws[56 + col] = DESCALE(tmp10 - tmp3, CB - P1);
        ws[8 + col] = DESCALE(tmp11 + tmp2, CB - P1); ws[48 + col] = DESCALE(tmp11 - tmp2, CB - P1);
        ws[16 + col] = DESCALE(tmp12 + tmp1, CB - P1); ws[40 + col] = DESCALE(tmp12 - tmp1, CB - P1);
        ws[24 + col] = DESCALE(tmp13 + tmp0, CB - P1); ws[32 + col] = DESCALE(tmp13 - tmp0, CB - P1);
#undef D
    }
    for (int row = 0; row < 8; ++row) {
        const int32_t* w = ws + 8 * row;
        int32_t z2 = w[2], z3 = w[6];
        int32_t z1 = (z2 + z3) * 4433;
        int32_t tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;
        int32_t tmp0 = (w[0] + w[4]) * (1 << CB), tmp1 = (w[0] - w[4]) * (1 << CB);
        const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int32_t z4 = tmp1 + tmp3;
        const int32_t z5 = (z3 + z4) * 9633;
        tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
        z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        uint8_t* o = out + row * stride;
        o[0] = idct_limit(DESCALE(tmp10 + tmp3, CB + P1 + 3)); o[7] = idct_limit(DESCALE(tmp10 - tmp3, CB + P1 + 3));
        o[1] = idct_limit(DESCALE(tmp11 + tmp2, CB + P1 + 3)); o[6] = idct_limit(DESCALE(tmp11 - tmp2, CB + P1 + 3));
        o[2] = idct_limit(DESCALE(tmp12 + tmp1, CB + P1 + 3)); o[5] = idct_limit(DESCALE(tmp12 - tmp1, CB + P1 + 3));
        o[3] = idct_limit(DESCALE(tmp13 + tmp0, CB + P1 + 3)); o[4] = idct_limit(DESCALE(tmp13 - tmp0, CB + P1 + 3));
    }
}

static uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* One output row of a chroma plane at full width.  in0 = the nearer input row, in1 = the farther one (h2v2), dw = down-sampled width. */
static void upsample_row(const uint8_t* in0, const uint8_t* in1, int dw, int h2, int v2, int v, uint8_t* out /* 2 * dw or dw values */) {
    if (!h2) { memcpy(out, in0, (size_t)dw); return; }
    if (dw <= 2) {                                          /* libjpeg-turbo: no fancy up-sampling for such narrow components */
        for (int x = 0; x < dw; ++x) out[2 * x] = out[2 * x + 1] = in0[x];
        return;
    }
    if (!v2) {                                              /* h2v1 fancy */
        out[0] = in0[0];
        out[1] = (uint8_t)((in0[0] * 3 + in0[1] + 2) >> 2);
        for (int x = 1; x < dw - 1; ++x) {
            const int iv = in0[x] * 3;
            out[2 * x] = (uint8_t)((iv + in0[x - 1] + 1) >> 2);
            out[2 * x + 1] = (uint8_t)((iv + in0[x + 1] + 2) >> 2);
        }
        out[2 * dw - 2] = (uint8_t)((in0[dw - 1] * 3 + in0[dw - 2] + 1) >> 2);
        out[2 * dw - 1] = in0[dw - 1];
        return;
    }
    (void)v;
    /* h2v2 fancy: column sums 3 * near + far, then 3:1 horizontally; biases 8 (left output) / 7 (right output) */
    int last, cur = in0[0] * 3 + in1[0], next = in0[1] * 3 + in1[1];
    out[0] = (uint8_t)((cur * 4 + 8) >> 4);
    out[1] = (uint8_t)((cur * 3 + next + 7) >> 4);
    last = cur; cur = next;
    for (int x = 1; x < dw - 1; ++x) {
        next = in0[x + 1] * 3 + in1[x + 1];
        out[2 * x] = (uint8_t)((cur * 3 + last + 8) >> 4);
        out[2 * x + 1] = (uint8_t)((cur * 3 + next + 7) >> 4);
        last = cur; cur = next;
    }
    out[2 * dw - 2] = (uint8_t)((cur * 3 + last + 8) >> 4);
    out[2 * dw - 1] = (uint8_t)((cur * 4 + 7) >> 4);
}

/* -> 0 ok, -1 corrupt / truncated, -2 unsupported (valid, but not this decoder's subset).  info: width, height, ncomp, h, v, restart */
int oracle_jpeg_info(const uint8_t* file, size_t n, int info[6]) {
    JpegHeader h;
    const int rc = parse_header(file, n, &h);
    if (rc == 0 || (rc == -2 && h.width)) { info[0] = h.width; info[1] = h.height; info[2] = h.ncomp; info[3] = h.hs[0]; info[4] = h.vs[0]; info[5] = h.restart; }
    return rc;
}

/* Quantised coefficients of every block in scan order (natural order inside a block), absolute DC: int16 [mcus * blocks_per_mcu][64].
 * Returns the number of blocks, < 0 on error. */
static long decode_coefficients(const uint8_t* file, size_t n, const JpegHeader* h, int16_t* coef) {
    Huff dc[4], ac[4];
    for (int t = 0; t < 4; ++t) {
        if (h->have_h[0][t] && build_huff(h->bits[0][t], h->vals[0][t], &dc[t])) return -1;
        if (h->have_h[1][t] && build_huff(h->bits[1][t], h->vals[1][t], &ac[t])) return -1;
    }
    Bits b = {file + h->scan_off, file + n, 0, 0, 0};
    int pred[3] = {0, 0, 0};
    const long mcus = (long)h->mcux * h->mcuy;
    int bpm = 0;
    for (int c = 0; c < h->ncomp; ++c) bpm += h->hs[c] * h->vs[c];
    long blk = 0;
    int rst_left = h->restart, next_rst = 0;
    for (long m = 0; m < mcus; ++m) {
        if (h->restart && rst_left == 0) {                 /* byte-align, expect RSTn, reset the predictions */
            b.acc = 0; b.cnt = 0;
            if (!b.marker) {                                /* the bit reader has not run into the marker yet: it must be next */
                while (b.p < b.end && !(b.p[0] == 0xFF && b.p + 1 < b.end && b.p[1] != 0 && b.p[1] != 0xFF)) ++b.p;
            }
            if (b.p + 1 >= b.end || b.p[0] != 0xFF || b.p[1] != 0xD0 + next_rst) return -1;
            b.p += 2; b.marker = 0;
            next_rst = (next_rst + 1) & 7;
            pred[0] = pred[1] = pred[2] = 0;
            rst_left = h->restart;
        }
        for (int c = 0; c < h->ncomp; ++c)
            for (int i = 0; i < h->hs[c] * h->vs[c]; ++i) {
                int16_t* blkp = coef + 64 * blk++;
                memset(blkp, 0, 64 * sizeof(int16_t));
                int s = decode_sym(&b, &dc[h->td[c]]);
                if (s < 0 || s > 11) return -1;
                pred[c] += extend(getbits(&b, s), s);
                blkp[0] = (int16_t)pred[c];
                for (int k = 1; k < 64;) {
                    const int rs = decode_sym(&b, &ac[h->ta[c]]);
                    if (rs < 0) return -1;
                    const int r = rs >> 4;
                    s = rs & 15;
                    if (s == 0) {
                        if (r == 15) { k += 16; continue; }
                        break;                               /* EOB */
                    }
                    k += r;
                    if (k > 63) return -1;
                    blkp[DZZ[k]] = (int16_t)extend(getbits(&b, s), s);
                    ++k;
                }
            }
        if (h->restart) --rst_left;
    }
    return blk;
}

long oracle_jpeg_decode_coefficients(const uint8_t* file, size_t n, int16_t* coef, long cap_blocks) {
    JpegHeader h;
    const int rc = parse_header(file, n, &h);
    if (rc) return rc;
    int bpm = 0;
    for (int c = 0; c < h.ncomp; ++c) bpm += h.hs[c] * h.vs[c];
    if ((long)h.mcux * h.mcuy * bpm > cap_blocks) return -3;
    return decode_coefficients(file, n, &h, coef);
}

/* file -> RGB u8 [height][width][3] (a grey-scale file: the grey value on all three channels; *ncomp_out says which it was) */
int oracle_jpeg_decode_rgb(const uint8_t* file, size_t n, uint8_t* out, int cap_w, int cap_h, int* ncomp_out) {
    JpegHeader h;
    const int rc = parse_header(file, n, &h);
    if (rc) return rc;
    if (h.width != cap_w || h.height != cap_h) return -3;
    int bpm = 0;
    for (int c = 0; c < h.ncomp; ++c) bpm += h.hs[c] * h.vs[c];
    const long nblk = (long)h.mcux * h.mcuy * bpm;
    int16_t* coef = (int16_t*)malloc((size_t)nblk * 64 * sizeof(int16_t));
    uint8_t* plane[3] = {0, 0, 0};
    int pw[3], ph[3];
    if (!coef) return -1;
    if (decode_coefficients(file, n, &h, coef) != nblk) { free(coef); return -1; }
    for (int c = 0; c < h.ncomp; ++c) {
        pw[c] = h.mcux * h.hs[c] * 8; ph[c] = h.mcuy * h.vs[c] * 8;
        plane[c] = (uint8_t*)malloc((size_t)pw[c] * ph[c]);
    }
    long blk = 0;
    for (int my = 0; my < h.mcuy; ++my)
        for (int mx = 0; mx < h.mcux; ++mx)
            for (int c = 0; c < h.ncomp; ++c)
                for (int by = 0; by < h.vs[c]; ++by)
                    for (int bx = 0; bx < h.hs[c]; ++bx)
                        idct_islow(coef + 64 * blk++, h.q[h.tq[c]], plane[c] + (size_t)((my * h.vs[c] + by) * 8) * pw[c] + (mx * h.hs[c] + bx) * 8, pw[c]);
    free(coef);
    if (ncomp_out) *ncomp_out = h.ncomp;
    if (h.ncomp == 1) {
        for (int y = 0; y < h.height; ++y)
            for (int x = 0; x < h.width; ++x) {
                const uint8_t v = plane[0][(size_t)y * pw[0] + x];
                uint8_t* o = out + ((size_t)y * h.width + x) * 3;
                o[0] = o[1] = o[2] = v;
            }
        free(plane[0]);
        return 0;
    }
    const int h2 = h.hmax == 2, v2 = h.vmax == 2;
    const int dw = (h.width + h.hmax - 1) / h.hmax, dh = (h.height + h.vmax - 1) / h.vmax;   /* real down-sampled size */
    uint8_t* rowb = (uint8_t*)malloc((size_t)2 * (2 * dw + 8));
    uint8_t* rowr = rowb + 2 * dw + 8;
    for (int y = 0; y < h.height; ++y) {
        const int cy = v2 ? y >> 1 : y;
        int far = cy;
        if (v2) { far = (y & 1) ? cy + 1 : cy - 1; if (far < 0) far = 0; if (far > dh - 1) far = dh - 1; }   /* context rows: edge replication */
        upsample_row(plane[1] + (size_t)cy * pw[1], plane[1] + (size_t)far * pw[1], dw, h2, v2, y & 1, rowb);
        upsample_row(plane[2] + (size_t)cy * pw[2], plane[2] + (size_t)far * pw[2], dw, h2, v2, y & 1, rowr);
        for (int x = 0; x < h.width; ++x) {
            const int yy = plane[0][(size_t)y * pw[0] + x], cb = rowb[x] - 128, cr = rowr[x] - 128;
            uint8_t* o = out + ((size_t)y * h.width + x) * 3;
            o[0] = clamp8(yy + ((91881 * cr + 32768) >> 16));
            o[1] = clamp8(yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
            o[2] = clamp8(yy + ((116130 * cb + 32768) >> 16));
        }
    }
    free(rowb);
    for (int c = 0; c < 3; ++c) free(plane[c]);
    return 0;
}
