/* ORACLE (test infrastructure only; never linked into the product).
 *
 * Plain-C, sequential restatement of the DB post-process, the recognition crop and the CTC
 * collapse that sit between the det and rec networks.  The reference has no local
 * implementation of any of this (the engine slot is
 * /root/reference/backend/services/ocr_service.py:213-246); the output shape it must produce
 * is the 4-point quad of /root/reference/backend/utils/ocr_postprocessor.py:24 and the flat
 * 8-number polygon of /root/reference/backend/services/ocr_service.py:295-311.
 * "parity unpinned": the steps follow the public PaddleOCR DBPostProcess convention
 * (threshold -> components -> min-area rectangle -> box score -> unclip -> integer quad) but
 * are DEFINED here, in integer / correctly-rounded fp64 arithmetic, so that a parallel
 * implementation can be bit-identical:
 *
 *  1. bitmap  = prob > thresh inside the real page (valid_h x valid_w).
 *  2. components: 8-connected; label = smallest linear index y*W+x of the component;
 *     candidates = the first max_candidates components in label order.
 *  3. hull: convex hull of the per-row leftmost/rightmost pixel centres (exact integers).
 *  4. min-area rectangle over hull edges, areas compared as exact rationals
 *     (wd*wn/L, 128-bit cross products); ties -> lexicographically smallest canonical
 *     direction.  A single-point component is skipped (zero-size).
 *  5. short side  min(wd,wn)/sqrt(L) < min_size -> skip.
 *  6. score = mean of prob (fixed point, trunc(prob * 2^24), exact integer sum) over the
 *     integer pixels inside the closed rectangle and the page; score < box_thresh -> skip.
 *  7. unclip: grow the rectangle by E = ratio*wd*wn / (2*(wd+wn)) frame units on each side
 *     (the exact min-area rectangle of the round-join offset polygon);
 *     short side (min(wd,wn)+2E)/sqrt(L) < min_size+2 -> skip.
 *  8. corners in fp64, ordered TL,TR,BR,BL (sort by (x,y); left pair by y; right pair by y),
 *     rint() to integers, clipped to [0,valid_w] x [0,valid_h];
 *     drop when int(|TL-TR|) <= 3 or int(|TL-BL|) <= 3.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

static float bf16_to_f32(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

typedef struct { int64_t x, y; } pt_t;

static int64_t cross(pt_t o, pt_t a, pt_t b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); }

/* canonical direction: primitive, dx > 0 or (dx == 0 and dy > 0) */
static int64_t gcd64(int64_t a, int64_t b) { a = llabs(a); b = llabs(b); while (b) { int64_t t = a % b; a = b; b = t; } return a; }
static void canon_dir(int64_t* dx, int64_t* dy) {
    int64_t g = gcd64(*dx, *dy);
    *dx /= g; *dy /= g;
    if (*dx < 0 || (*dx == 0 && *dy < 0)) { *dx = -*dx; *dy = -*dy; }
}

typedef struct { double x, y; } dpt_t;

static int cmp_xy(const void* a, const void* b) {
    const dpt_t* p = (const dpt_t*)a; const dpt_t* q = (const dpt_t*)b;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    return 0;
}

/* returns number of boxes written (<= max_out). boxes: [n][8] = TLx,TLy,TRx,TRy,BRx,BRy,BLx,BLy */
int oracle_db_postprocess(const uint16_t* prob, int H, int W, int valid_h, int valid_w, float thresh,
                          float box_thresh, float unclip_ratio, int min_size, int max_candidates,
                          int32_t* boxes, float* scores, int max_out, int32_t* n_components) {
    int32_t* label = (int32_t*)malloc(sizeof(int32_t) * (size_t)H * W);
    int32_t* stack = (int32_t*)malloc(sizeof(int32_t) * (size_t)H * W);
    for (size_t i = 0; i < (size_t)H * W; ++i) label[i] = -1;
    int nout = 0, ncomp = 0;
    int32_t* rmin = (int32_t*)malloc(sizeof(int32_t) * H);
    int32_t* rmax = (int32_t*)malloc(sizeof(int32_t) * H);
    pt_t* hull = (pt_t*)malloc(sizeof(pt_t) * (2 * (size_t)H + 4));
    pt_t* chain = (pt_t*)malloc(sizeof(pt_t) * ((size_t)H + 2));

    for (int y0 = 0; y0 < valid_h; ++y0)
        for (int x0 = 0; x0 < valid_w; ++x0) {
            size_t i0 = (size_t)y0 * W + x0;
            if (label[i0] >= 0 || !(bf16_to_f32(prob[i0]) > thresh)) continue;
            /* raster scan => i0 is the smallest linear index of a new component */
            int cand = ncomp < max_candidates;
            ++ncomp;
            int sp = 0, ymin = y0, ymax = y0;
            stack[sp++] = (int32_t)i0; label[i0] = (int32_t)i0;
            if (cand) { for (int r = 0; r < H; ++r) { rmin[r] = INT32_MAX; rmax[r] = -1; } }
            while (sp) {
                int32_t i = stack[--sp];
                int y = i / W, x = i % W;
                if (cand) { if (x < rmin[y]) rmin[y] = x; if (x > rmax[y]) rmax[y] = x; }
                if (y < ymin) ymin = y; if (y > ymax) ymax = y;
                for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
                    int yy = y + dy, xx = x + dx;
                    if ((dy | dx) == 0 || yy < 0 || yy >= valid_h || xx < 0 || xx >= valid_w) continue;
                    size_t j = (size_t)yy * W + xx;
                    if (label[j] < 0 && bf16_to_f32(prob[j]) > thresh) { label[j] = (int32_t)i0; stack[sp++] = (int32_t)j; }
                }
            }
            if (!cand) continue;
            /* ---- convex hull of row extremes: left chain top->bottom, right chain bottom->top ---- */
            int nh = 0, nc = 0;
            for (int r = ymin; r <= ymax; ++r) {          /* left chain: keep strict left turns (y down) */
                pt_t p = { rmin[r], r };
                while (nc >= 2 && cross(chain[nc - 2], chain[nc - 1], p) >= 0) --nc;
                chain[nc++] = p;
            }
            for (int k = 0; k < nc; ++k) hull[nh++] = chain[k];
            nc = 0;
            for (int r = ymax; r >= ymin; --r) {          /* right chain */
                pt_t p = { rmax[r], r };
                while (nc >= 2 && cross(chain[nc - 2], chain[nc - 1], p) >= 0) --nc;
                chain[nc++] = p;
            }
            for (int k = 0; k < nc; ++k) {
                if (nh && hull[nh - 1].x == chain[k].x && hull[nh - 1].y == chain[k].y) continue;
                hull[nh++] = chain[k];
            }
            if (nh > 1 && hull[0].x == hull[nh - 1].x && hull[0].y == hull[nh - 1].y) --nh;
            if (nh < 2) continue; /* single pixel */
            /* ---- min-area rectangle over hull edges ---- */
            int have = 0;
            int64_t bdx = 0, bdy = 0, bmind = 0, bmaxd = 0, bminn = 0, bmaxn = 0, bL = 1, bA = 0;
            for (int e = 0; e < nh; ++e) {
                pt_t a = hull[e], b = hull[(e + 1) % nh];
                int64_t dx = b.x - a.x, dy = b.y - a.y;
                if (dx == 0 && dy == 0) continue;
                canon_dir(&dx, &dy);
                int64_t mind = INT64_MAX, maxd = INT64_MIN, minn = INT64_MAX, maxn = INT64_MIN;
                for (int k = 0; k < nh; ++k) {
                    int64_t pd = hull[k].x * dx + hull[k].y * dy, pn = -hull[k].x * dy + hull[k].y * dx;
                    if (pd < mind) mind = pd; if (pd > maxd) maxd = pd;
                    if (pn < minn) minn = pn; if (pn > maxn) maxn = pn;
                }
                int64_t L = dx * dx + dy * dy, A = (maxd - mind) * (maxn - minn);
                int better;
                if (!have) better = 1;
                else {
                    u128 lhs = (u128)(uint64_t)A * (uint64_t)bL, rhs = (u128)(uint64_t)bA * (uint64_t)L;
                    if (lhs != rhs) better = lhs < rhs;
                    else better = (dx < bdx) || (dx == bdx && dy < bdy);
                }
                if (better) { have = 1; bdx = dx; bdy = dy; bmind = mind; bmaxd = maxd; bminn = minn; bmaxn = maxn; bL = L; bA = A; }
            }
            if (!have) continue;
            int64_t wd = bmaxd - bmind, wn = bmaxn - bminn;
            double sqL = sqrt((double)bL);
            double sside = (double)(wd < wn ? wd : wn) / sqL;
            if (sside < (double)min_size) continue;
            /* ---- score over integer pixels inside the closed rectangle ---- */
            double cx[4], cy[4];
            {
                int64_t as[4] = { bmind, bmaxd, bmaxd, bmind }, bs[4] = { bminn, bminn, bmaxn, bmaxn };
                for (int k = 0; k < 4; ++k) {
                    cx[k] = (double)(as[k] * bdx - bs[k] * bdy) / (double)bL;
                    cy[k] = (double)(as[k] * bdy + bs[k] * bdx) / (double)bL;
                }
            }
            double fx0 = cx[0], fx1 = cx[0], fy0 = cy[0], fy1 = cy[0];
            for (int k = 1; k < 4; ++k) { if (cx[k] < fx0) fx0 = cx[k]; if (cx[k] > fx1) fx1 = cx[k]; if (cy[k] < fy0) fy0 = cy[k]; if (cy[k] > fy1) fy1 = cy[k]; }
            int bx0 = (int)floor(fx0) - 1, bx1 = (int)ceil(fx1) + 1, by0 = (int)floor(fy0) - 1, by1 = (int)ceil(fy1) + 1;
            if (bx0 < 0) bx0 = 0; if (by0 < 0) by0 = 0; if (bx1 > valid_w - 1) bx1 = valid_w - 1; if (by1 > valid_h - 1) by1 = valid_h - 1;
            uint64_t sum = 0, cnt = 0;
            for (int y = by0; y <= by1; ++y) for (int x = bx0; x <= bx1; ++x) {
                int64_t pd = (int64_t)x * bdx + (int64_t)y * bdy, pn = -(int64_t)x * bdy + (int64_t)y * bdx;
                if (pd < bmind || pd > bmaxd || pn < bminn || pn > bmaxn) continue;
                sum += (uint64_t)(bf16_to_f32(prob[(size_t)y * W + x]) * 16777216.0f);
                ++cnt;
            }
            if (!cnt) continue;
            double score = ((double)sum / (double)cnt) / 16777216.0;
            if (score < (double)box_thresh) continue;
            /* ---- unclip ---- */
            double E = ((double)unclip_ratio * (double)(wd * wn)) / (double)(2 * (wd + wn));
            double sside2 = ((double)(wd < wn ? wd : wn) + 2.0 * E) / sqL;
            if (sside2 < (double)(min_size + 2)) continue;
            double a0 = (double)bmind - E, a1 = (double)bmaxd + E, b0 = (double)bminn - E, b1 = (double)bmaxn + E;
            dpt_t c[4];
            {
                double as[4] = { a0, a1, a1, a0 }, bs[4] = { b0, b0, b1, b1 };
                for (int k = 0; k < 4; ++k) {
                    double t1 = as[k] * (double)bdx, t2 = bs[k] * (double)bdy, t3 = as[k] * (double)bdy, t4 = bs[k] * (double)bdx;
                    c[k].x = (t1 - t2) / (double)bL;
                    c[k].y = (t3 + t4) / (double)bL;
                }
            }
            qsort(c, 4, sizeof(dpt_t), cmp_xy);
            dpt_t tl, bl, tr, br;
            if (c[0].y <= c[1].y) { tl = c[0]; bl = c[1]; } else { tl = c[1]; bl = c[0]; }
            if (c[2].y <= c[3].y) { tr = c[2]; br = c[3]; } else { tr = c[3]; br = c[2]; }
            dpt_t q[4] = { tl, tr, br, bl };
            int32_t out[8];
            for (int k = 0; k < 4; ++k) {
                double rx = rint(q[k].x), ry = rint(q[k].y);
                if (rx < 0) rx = 0; if (rx > valid_w) rx = valid_w; if (ry < 0) ry = 0; if (ry > valid_h) ry = valid_h;
                out[2 * k] = (int32_t)rx; out[2 * k + 1] = (int32_t)ry;
            }
            {
                double wx = (double)(out[0] - out[2]), wy = (double)(out[1] - out[3]);
                double hx = (double)(out[0] - out[6]), hy = (double)(out[1] - out[7]);
                int rw = (int)sqrt(wx * wx + wy * wy), rh = (int)sqrt(hx * hx + hy * hy);
                if (rw <= 3 || rh <= 3) continue;
            }
            if (nout < max_out) { memcpy(boxes + 8 * nout, out, sizeof(out)); scores[nout] = (float)score; ++nout; }
        }
    if (n_components) *n_components = ncomp;
    free(label); free(stack); free(rmin); free(rmax); free(hull); free(chain);
    return nout;
}

/* Recognition crop: sample the (rotated) rectangle TL,TR,BR,BL of an HxWx3 u8 page into a
 * 32 x wc strip (wc <= 320, stored in a 32x320x3 buffer, unused columns 0).  Affine map through
 * three corners, bilinear, fp32 with one rounding per operation (no fma), replicate border.
 *   cw2 = max(|TR-TL|^2, |BR-BL|^2), ch2 = max(|BL-TL|^2, |BR-TR|^2)
 *   4*ch2 >= 9*cw2 (h/w >= 1.5): rotate 90 deg counter-clockwise (TL<-TR, TR<-BR, BR<-BL, BL<-TL)
 *   wc = clamp(ceil(32 * sqrt(cw2/ch2)), 1, 320)
 */
int oracle_rec_crop(const uint8_t* page, int H, int W, const int32_t* box, uint8_t* out) {
    int64_t p[4][2];
    for (int k = 0; k < 4; ++k) { p[k][0] = box[2 * k]; p[k][1] = box[2 * k + 1]; }
#define D2(a, b) ((p[a][0] - p[b][0]) * (p[a][0] - p[b][0]) + (p[a][1] - p[b][1]) * (p[a][1] - p[b][1]))
    int64_t cw2 = D2(1, 0) > D2(2, 3) ? D2(1, 0) : D2(2, 3);
    int64_t ch2 = D2(3, 0) > D2(2, 1) ? D2(3, 0) : D2(2, 1);
    if (4 * ch2 >= 9 * cw2) {
        int64_t t0 = p[0][0], t1 = p[0][1];
        p[0][0] = p[1][0]; p[0][1] = p[1][1]; p[1][0] = p[2][0]; p[1][1] = p[2][1];
        p[2][0] = p[3][0]; p[2][1] = p[3][1]; p[3][0] = t0; p[3][1] = t1;
        int64_t t = cw2; cw2 = ch2; ch2 = t;
    }
    memset(out, 0, 32 * 320 * 3);
    if (ch2 == 0 || cw2 == 0) return 0;
    double ratio = sqrt((double)cw2 / (double)ch2);
    int wc = (int)ceil(32.0 * ratio);
    if (wc < 1) wc = 1; if (wc > 320) wc = 320;
    float tlx = (float)p[0][0], tly = (float)p[0][1];
    float ex = (float)(p[1][0] - p[0][0]), ey = (float)(p[1][1] - p[0][1]);
    float fx = (float)(p[3][0] - p[0][0]), fy = (float)(p[3][1] - p[0][1]);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < wc; ++j) {
        volatile float u = ((float)j + 0.5f) / (float)wc, v = ((float)i + 0.5f) / 32.0f;
        volatile float t1 = u * ex, t2 = v * fx, t3 = u * ey, t4 = v * fy;
        volatile float sx0 = tlx + t1, sy0 = tly + t3;
        volatile float sx = sx0 + t2, sy = sy0 + t4;
        float x0f = floorf(sx), y0f = floorf(sy);
        volatile float ax = sx - x0f, ay = sy - y0f;
        int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
        if (x0 < 0) x0 = 0; if (x0 > W - 1) x0 = W - 1; if (x1 < 0) x1 = 0; if (x1 > W - 1) x1 = W - 1;
        if (y0 < 0) y0 = 0; if (y0 > H - 1) y0 = H - 1; if (y1 < 0) y1 = 0; if (y1 > H - 1) y1 = H - 1;
        volatile float bx = 1.0f - ax, by = 1.0f - ay;
        for (int c = 0; c < 3; ++c) {
            float p00 = page[((size_t)y0 * W + x0) * 3 + c], p01 = page[((size_t)y0 * W + x1) * 3 + c];
            float p10 = page[((size_t)y1 * W + x0) * 3 + c], p11 = page[((size_t)y1 * W + x1) * 3 + c];
            volatile float a = bx * p00, b = ax * p01, cc = bx * p10, d = ax * p11;
            volatile float top = a + b, bot = cc + d;
            volatile float e = by * top, f = ay * bot;
            volatile float val = e + f;
            float r = rintf(val);
            if (r < 0.f) r = 0.f; if (r > 255.f) r = 255.f;
            out[((size_t)i * 320 + j) * 3 + c] = (uint8_t)r;
        }
    }
    return wc;
}
