/*
 * v3d_oracle.c -- CPU restatement of the hot path.  TEST INFRASTRUCTURE ONLY (see v3d_oracle.h).
 * PARITY UNPINNED: restates OpenCV 4.x algorithms from their published behaviour; OpenCV is
 * not available in the build container and the reference ships no golden vectors.
 *
 * Layout conventions: images row-major; cost volumes [y][xr][d] with d fastest, xr = x - D.
 */
#include "v3d_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define MAX_COST 32767
#define DISP_SHIFT 4
#define DISP_SCALE 16

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int16_t sat16(int v) { return (int16_t)iclamp(v, -32768, 32767); }

/* ------------------------------------------------------------------------------------------
 * depth.py:315-325 -- cv2.StereoSGBM_create(minDisparity=0, numDisparities=64, blockSize=5,
 * P1=8*3*5**2, P2=32*3*5**2, disp12MaxDiff=1, uniquenessRatio=10, speckleWindowSize=100,
 * speckleRange=32); preFilterCap and mode keep their defaults (0, MODE_SGBM).
 * ------------------------------------------------------------------------------------------ */
void orc_sgbm_default_params(orc_sgbm_params* p)
{
    p->minDisparity = 0; p->numDisparities = 64; p->blockSize = 5;
    p->P1 = 600; p->P2 = 2400; p->disp12MaxDiff = 1; p->preFilterCap = 0;
    p->uniquenessRatio = 10; p->speckleWindowSize = 100; p->speckleRange = 32; p->mode = 0;
}

/* derived constants exactly as StereoSGBM derives them */
typedef struct { int D, SW2, SH2, P1, P2, ftzero, uniq, d12, W1; } derived_t;

static int derive(const orc_sgbm_params* p, int W, derived_t* q)
{
    if (p->minDisparity != 0 || p->numDisparities <= 0 || (p->numDisparities % 16)) return -1;
    q->D = p->numDisparities;
    int bs = p->blockSize > 0 ? p->blockSize : 1;          /* calcSADWindowSize */
    q->SW2 = q->SH2 = bs / 2;
    q->P1 = p->P1 > 0 ? p->P1 : 2;
    q->P2 = imax(p->P2 > 0 ? p->P2 : 5, q->P1 + 1);
    q->ftzero = imax(p->preFilterCap, 15) | 1;
    q->uniq = p->uniquenessRatio >= 0 ? p->uniquenessRatio : 10;
    q->d12 = p->disp12MaxDiff > 0 ? p->disp12MaxDiff : 1;
    q->W1 = W - q->D;                                      /* maxX1 - minX1, minD = 0 */
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Pre-filter planes of one image row (StereoSGBM calcPixelCostBT, first half):
 *   grad[x] = clip(x-Sobel, +-ftzero) + ftzero for 1 <= x <= W-2; raw[x] = I[y][x];
 *   columns 0 and W-1 of BOTH planes are tab[0] = ftzero.
 * ------------------------------------------------------------------------------------------ */
static void prefilter_row(const uint8_t* I, int W, int H, int y, int ftzero, uint8_t* grad, uint8_t* raw)
{
    const uint8_t* r0 = I + (size_t)y * W;
    const uint8_t* ru = I + (size_t)(y > 0 ? y - 1 : y) * W;
    const uint8_t* rd = I + (size_t)(y < H - 1 ? y + 1 : y) * W;
    grad[0] = grad[W - 1] = raw[0] = raw[W - 1] = (uint8_t)ftzero;
    for (int x = 1; x < W - 1; x++) {
        int g = (r0[x + 1] - r0[x - 1]) * 2 + (ru[x + 1] - ru[x - 1]) + (rd[x + 1] - rd[x - 1]);
        grad[x] = (uint8_t)(iclamp(g, -ftzero, ftzero) + ftzero);
        raw[x] = r0[x];
    }
}

/* half-sample interval [lo, hi] of plane p at x (Birchfield-Tomasi) */
static inline void half_interval(const uint8_t* p, int W, int x, int* lo, int* hi)
{
    int v = p[x];
    int l = x > 0 ? (v + p[x - 1]) / 2 : v;
    int r = x < W - 1 ? (v + p[x + 1]) / 2 : v;
    *lo = imin(imin(l, r), v);
    *hi = imax(imax(l, r), v);
}

/* BT pixel cost of one row: pix[xr*D + d], xr in [0,W1), = BT(grad) + (BT(raw) >> 2) */
static void pixel_cost_row(const uint8_t* I1, const uint8_t* I2, int W, int H, int y,
                           const derived_t* q, uint16_t* pix, uint8_t* tmp /* 4*W */)
{
    const int D = q->D, W1 = q->W1;
    uint8_t *g1 = tmp, *r1 = tmp + W, *g2 = tmp + 2 * W, *r2 = tmp + 3 * W;
    prefilter_row(I1, W, H, y, q->ftzero, g1, r1);
    prefilter_row(I2, W, H, y, q->ftzero, g2, r2);
    memset(pix, 0, sizeof(uint16_t) * (size_t)W1 * D);
    for (int plane = 0; plane < 2; plane++) {
        const uint8_t* p1 = plane ? r1 : g1;
        const uint8_t* p2 = plane ? r2 : g2;
        const int shift = plane ? 2 : 0;
        for (int x = D; x < W; x++) {
            int u = p1[x], u0, u1;
            half_interval(p1, W, x, &u0, &u1);
            uint16_t* out = pix + (size_t)(x - D) * D;
            for (int d = 0; d < D; d++) {
                int v = p2[x - d], v0, v1;
                half_interval(p2, W, x - d, &v0, &v1);
                int c0 = imax(imax(0, u - v1), v0 - u);
                int c1 = imax(imax(0, v - u1), u0 - v);
                out[d] = (uint16_t)(out[d] + (imin(c0, c1) >> shift));
            }
        }
    }
}

/* C[y][xr][d] = P2 + sum over the (2SH2+1)x(2SW2+1) window with replicated borders of the COST region */
static int cost_volume(const derived_t* q, const uint8_t* I1, const uint8_t* I2, int W, int H, int16_t* C)
{
    const int D = q->D, W1 = q->W1, SW2 = q->SW2, SH2 = q->SH2;
    const size_t row = (size_t)W1 * D;
    uint16_t* pix = (uint16_t*)malloc(sizeof(uint16_t) * row);
    int32_t* hsum = (int32_t*)malloc(sizeof(int32_t) * row * H);
    uint8_t* tmp = (uint8_t*)malloc((size_t)4 * W);
    if (!pix || !hsum || !tmp) { free(pix); free(hsum); free(tmp); return -2; }
    for (int y = 0; y < H; y++) {
        pixel_cost_row(I1, I2, W, H, y, q, pix, tmp);
        int32_t* hs = hsum + row * y;
        for (int xr = 0; xr < W1; xr++)
            for (int d = 0; d < D; d++) {
                int s = 0;
                for (int j = -SW2; j <= SW2; j++) s += pix[(size_t)iclamp(xr + j, 0, W1 - 1) * D + d];
                hs[(size_t)xr * D + d] = s;
            }
    }
    for (int y = 0; y < H; y++) {
        int16_t* Cy = C + row * y;
        for (size_t i = 0; i < row; i++) {
            int s = q->P2;
            for (int k = -SH2; k <= SH2; k++) s += hsum[row * iclamp(y + k, 0, H - 1) + i];
            Cy[i] = (int16_t)s;
        }
    }
    free(pix); free(hsum); free(tmp);
    return 0;
}

int orc_sgbm_cost_volume(const orc_sgbm_params* p, const uint8_t* left, const uint8_t* right,
                         int W, int H, int16_t* C)
{
    derived_t q;
    if (derive(p, W, &q) || q.W1 <= 0) return -1;
    return cost_volume(&q, left, right, W, H, C);
}

/* one SGM path step: L[d] = C[d] + min(Lp[d], Lp[d-1]+P1, Lp[d+1]+P1, delta) - delta, delta = minLp + P2.
   Lp has MAX_COST sentinels at [-1] and [D].  returns min_d L[d] */
static inline int path_step(const int16_t* Cp, const int16_t* Lp, int minLp, int D, int P1, int P2, int16_t* L)
{
    const int delta = minLp + P2;
    int mn = MAX_COST;
    for (int d = 0; d < D; d++) {
        int m = imin(imin(Lp[d], Lp[d - 1] + P1), imin(Lp[d + 1] + P1, delta));
        int v = Cp[d] + m - delta;
        L[d] = (int16_t)v;
        mn = imin(mn, v);
    }
    return mn;
}

/* WTA + uniqueness + disp2 + sub-pixel for one pixel of the final S (stereosgbm.cpp's per-row tail) */
static inline void wta_pixel(const int16_t* Sp, int x_img, const derived_t* q,
                             int16_t* disp_row, int16_t* disp2, int16_t* disp2cost)
{
    const int D = q->D;
    int minS = MAX_COST, best = -1;
    for (int d = 0; d < D; d++)
        if (Sp[d] < minS) { minS = Sp[d]; best = d; }
    int d;
    for (d = 0; d < D; d++)
        if (Sp[d] * (100 - q->uniq) < minS * 100 && abs(best - d) > 1) break;
    if (d < D) return;
    d = best;
    if (d < 0) return;                      /* cannot happen for S < MAX_COST; guards x2 */
    int x2 = x_img - d;
    if (disp2cost[x2] > minS) { disp2cost[x2] = (int16_t)minS; disp2[x2] = (int16_t)d; }
    if (0 < d && d < D - 1) {
        int denom2 = imax(Sp[d - 1] + Sp[d + 1] - 2 * Sp[d], 1);
        d = d * DISP_SCALE + ((Sp[d - 1] - Sp[d + 1]) * DISP_SCALE + denom2) / (denom2 * 2);
    } else
        d *= DISP_SCALE;
    disp_row[x_img] = (int16_t)d;
}

static void lr_check_row(int16_t* disp_row, const int16_t* disp2, int W, const derived_t* q)
{
    for (int x = q->D; x < W; x++) {
        int d1 = disp_row[x];
        if (d1 == -DISP_SCALE) continue;
        int _d = d1 >> DISP_SHIFT, d_ = (d1 + DISP_SCALE - 1) >> DISP_SHIFT;
        int _x = x - _d, x_ = x - d_;
        if (0 <= _x && _x < W && disp2[_x] >= 0 && abs(disp2[_x] - _d) > q->d12 &&
            0 <= x_ && x_ < W && disp2[x_] >= 0 && abs(disp2[x_] - d_) > q->d12)
            disp_row[x] = (int16_t)(-DISP_SCALE);
    }
}

/* L buffers: [slot 0..W1+1][D+2], slot = xr+1; slots 0 and W1+1 are the always-zero borders */
#define LSLOT(buf, xr) ((buf) + (size_t)((xr) + 1) * (D + 2) + 1)

static int sgbm_raw(const derived_t* q, int mode, const uint8_t* I1, const uint8_t* I2, int W, int H,
                    int16_t* disp, int16_t* S_out)
{
    const int D = q->D, W1 = q->W1, P1 = q->P1, P2 = q->P2;
    const size_t row = (size_t)W1 * D;
    for (size_t i = 0; i < (size_t)W * H; i++) disp[i] = -DISP_SCALE;
    if (W1 <= 0) return 0;

    int16_t* C = (int16_t*)malloc(sizeof(int16_t) * row * H);
    int16_t* S = (mode == 1 || S_out) ? (S_out ? S_out : (int16_t*)malloc(sizeof(int16_t) * row * H)) : NULL;
    int16_t* Srow = (int16_t*)malloc(sizeof(int16_t) * row);
    const size_t lsz = (size_t)(W1 + 2) * (D + 2);
    int16_t* Lbuf = (int16_t*)calloc(lsz * 8, sizeof(int16_t));   /* [2 rows][4 dirs] */
    int16_t* mbuf = (int16_t*)calloc((size_t)(W1 + 2) * 8, sizeof(int16_t));
    int16_t* disp2 = (int16_t*)malloc(sizeof(int16_t) * W * 2);
    if (!C || !Srow || !Lbuf || !mbuf || !disp2 || ((mode == 1 || S_out) && !S)) return -2;
    int16_t* disp2cost = disp2 + W;
    int rc = cost_volume(q, I1, I2, W, H, C);
    if (rc) return rc;

    const int npasses = mode == 1 ? 2 : 1;
    for (int pass = 1; pass <= npasses; pass++) {
        const int y1 = pass == 1 ? 0 : H - 1, y2 = pass == 1 ? H : -1, dy = pass == 1 ? 1 : -1;
        const int x1 = pass == 1 ? 0 : W1 - 1, x2 = pass == 1 ? W1 : -1, dx = pass == 1 ? 1 : -1;
        memset(Lbuf, 0, sizeof(int16_t) * lsz * 8);
        memset(mbuf, 0, sizeof(int16_t) * (size_t)(W1 + 2) * 8);
        int cur = 0;
        for (int y = y1; y != y2; y += dy) {
            const int16_t* Cy = C + row * y;
            int16_t* Sy = S ? S + row * y : Srow;
            if (pass == 1) memset(Sy, 0, sizeof(int16_t) * row);
            int16_t* Lc[4]; int16_t* Lp[4]; int16_t* mc[4]; int16_t* mp[4];
            for (int r = 0; r < 4; r++) {
                Lc[r] = Lbuf + lsz * (cur * 4 + r);       Lp[r] = Lbuf + lsz * ((1 - cur) * 4 + r);
                mc[r] = mbuf + (size_t)(W1 + 2) * (cur * 4 + r) + 1;
                mp[r] = mbuf + (size_t)(W1 + 2) * ((1 - cur) * 4 + r) + 1;
            }
            /* directions 0: (x-dx, y)   1: (x-1, y-dy)   2: (x, y-dy)   3: (x+1, y-dy) */
            for (int x = x1; x != x2; x += dx) {
                int16_t* pr[4] = { LSLOT(Lc[0], x - dx), LSLOT(Lp[1], x - 1), LSLOT(Lp[2], x), LSLOT(Lp[3], x + 1) };
                int pm[4] = { mc[0][x - dx], mp[1][x - 1], mp[2][x], mp[3][x + 1] };
                const int16_t* Cp = Cy + (size_t)x * D;
                int16_t* Sp = Sy + (size_t)x * D;
                int32_t acc[512];
                for (int d = 0; d < D; d++) acc[d] = Sp[d];
                for (int r = 0; r < 4; r++) {
                    pr[r][-1] = pr[r][D] = MAX_COST;
                    int16_t* L = LSLOT(Lc[r], x);
                    mc[r][x] = (int16_t)path_step(Cp, pr[r], pm[r], D, P1, P2, L);
                    for (int d = 0; d < D; d++) acc[d] += L[d];
                }
                for (int d = 0; d < D; d++) Sp[d] = sat16(acc[d]);
            }
            if (pass == npasses) {
                int16_t* drow = disp + (size_t)y * W;
                for (int x = 0; x < W; x++) { disp2[x] = -DISP_SCALE; disp2cost[x] = MAX_COST; }
                for (int x = W1 - 1; x >= 0; x--) {
                    int16_t* Sp = Sy + (size_t)x * D;
                    if (npasses == 1) {
                        /* direction 4: (x+1, y), right-to-left on the same row; reuses the dir-0 slots */
                        int16_t* prv = LSLOT(Lc[0], x + 1);
                        prv[-1] = prv[D] = MAX_COST;
                        int16_t* L = LSLOT(Lc[0], x);
                        mc[0][x] = (int16_t)path_step(Cy + (size_t)x * D, prv, mc[0][x + 1], D, P1, P2, L);
                        for (int d = 0; d < D; d++) Sp[d] = sat16(Sp[d] + L[d]);
                    }
                    wta_pixel(Sp, x + D, q, drow, disp2, disp2cost);
                }
                lr_check_row(drow, disp2, W, q);
            }
            cur = 1 - cur;
            /* the border slots of the row that becomes "current" must read as zero: they are never
               written except for the sentinels, which path_step never reads as [0..D) */
        }
    }
    free(C); if (S && S != S_out) free(S); free(Srow); free(Lbuf); free(mbuf); free(disp2);
    return 0;
}

int orc_sgbm_raw(const orc_sgbm_params* p, const uint8_t* left, const uint8_t* right,
                 int W, int H, int16_t* disp16, int16_t* S_out)
{
    derived_t q;
    if (derive(p, W, &q) || q.D > 512) return -1;
    return sgbm_raw(&q, p->mode, left, right, W, H, disp16, S_out);
}

/* cv::medianBlur(disp, disp, 3) on CV_16S: replicated borders, INVALID participates as a value */
void orc_median3x3_i16(const int16_t* src, int W, int H, int16_t* dst)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int16_t v[9]; int n = 0;
            for (int j = -1; j <= 1; j++)
                for (int i = -1; i <= 1; i++)
                    v[n++] = src[(size_t)iclamp(y + j, 0, H - 1) * W + iclamp(x + i, 0, W - 1)];
            for (int a = 1; a < 9; a++) {           /* insertion sort */
                int16_t t = v[a]; int b = a - 1;
                while (b >= 0 && v[b] > t) { v[b + 1] = v[b]; b--; }
                v[b + 1] = t;
            }
            dst[(size_t)y * W + x] = v[4];
        }
}

/* cv::filterSpeckles: 4-connected flood fill over pixels != newVal, edge iff |a-b| <= maxDiff;
   regions with <= maxSpeckleSize pixels are set to newVal */
void orc_filter_speckles(int16_t* img, int W, int H, int newVal, int maxSpeckleSize, int maxDiff)
{
    const size_t n = (size_t)W * H;
    int32_t* labels = (int32_t*)calloc(n, sizeof(int32_t));
    int32_t* stack = (int32_t*)malloc(n * sizeof(int32_t));
    uint8_t* small = (uint8_t*)calloc(n + 1, 1);
    int cur = 0;
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            size_t at = (size_t)i * W + j;
            if (img[at] == newVal) continue;
            if (labels[at]) { if (small[labels[at]]) img[at] = (int16_t)newVal; continue; }
            int32_t* ws = stack; int32_t p = (int32_t)at; int count = 0;
            labels[at] = ++cur;
            for (;;) {
                count++;
                int py = p / W, px = p % W, dp = img[p];
                if (py < H - 1 && !labels[p + W] && img[p + W] != newVal && abs(dp - img[p + W]) <= maxDiff) { labels[p + W] = cur; *ws++ = p + W; }
                if (py > 0 && !labels[p - W] && img[p - W] != newVal && abs(dp - img[p - W]) <= maxDiff) { labels[p - W] = cur; *ws++ = p - W; }
                if (px < W - 1 && !labels[p + 1] && img[p + 1] != newVal && abs(dp - img[p + 1]) <= maxDiff) { labels[p + 1] = cur; *ws++ = p + 1; }
                if (px > 0 && !labels[p - 1] && img[p - 1] != newVal && abs(dp - img[p - 1]) <= maxDiff) { labels[p - 1] = cur; *ws++ = p - 1; }
                if (ws == stack) break;
                p = *--ws;
            }
            if (count <= maxSpeckleSize) { small[cur] = 1; img[at] = (int16_t)newVal; }
        }
    free(labels); free(stack); free(small);
}

/* StereoSGBMImpl::compute: SGM, then medianBlur(3), then filterSpeckles if speckleWindowSize > 0 */
int orc_sgbm_compute(const orc_sgbm_params* p, const uint8_t* left, const uint8_t* right,
                     int W, int H, int16_t* disp16)
{
    int16_t* raw = (int16_t*)malloc(sizeof(int16_t) * (size_t)W * H);
    if (!raw) return -2;
    int rc = orc_sgbm_raw(p, left, right, W, H, raw, NULL);
    if (rc) { free(raw); return rc; }
    orc_median3x3_i16(raw, W, H, disp16);
    free(raw);
    if (p->speckleWindowSize > 0)
        orc_filter_speckles(disp16, W, H, (p->minDisparity - 1) * DISP_SCALE, p->speckleWindowSize,
                            DISP_SCALE * p->speckleRange);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * depth.py:250-268: cv2.resize(half, (2*halfW, H), INTER_LANCZOS4) on 8-bit BGR.
 * 8-tap Lanczos (a=4), taps quantised to int16 at 2^11, separable, int32 intermediate, result
 * (v + 2^21) >> 22 saturated; vertical scale 1 => vertical taps are the identity (x2048).
 * ------------------------------------------------------------------------------------------ */
void orc_lanczos4_taps(float x, int16_t taps[8])
{
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[8][2] = { {1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45} };
    const double PI = 3.1415926535897932384626433832795;
    float c[8], sum = 0.f;
    double y0 = -(x + 3) * PI * 0.25, s0 = sin(y0), c0 = cos(y0);
    for (int i = 0; i < 8; i++) {
        float yi = (x + 3 - i);
        if (fabsf(yi) >= 1e-6f) {
            double y = -yi * PI * 0.25;
            c[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        } else
            c[i] = 1e30f;
        sum += c[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; i++) {
        float v = c[i] * sum * 2048.f;
        long r = lrintf(v);                       /* cvRound: round-half-even */
        taps[i] = (int16_t)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
    }
}

/* horizontal Lanczos resize of one interleaved row, cn channels, sw -> dw */
static void lanczos_row(const uint8_t* src, int sw, int cn, uint8_t* dst, int dw)
{
    const double scale = (double)sw / dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        int16_t t[8];
        orc_lanczos4_taps(fx, t);
        for (int c = 0; c < cn; c++) {
            int acc = 0;
            for (int k = 0; k < 8; k++) acc += src[(size_t)iclamp(sx + k - 3, 0, sw - 1) * cn + c] * t[k];
            long long v = ((long long)acc * 2048 + (1 << 21)) >> 22;
            dst[(size_t)dx * cn + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

int orc_split_sbs(const uint8_t* sbs, int W, int H, int unsqueeze, uint8_t* L, uint8_t* R)
{
    if (W % 2) return -1;
    const int hw = W / 2, ow = unsqueeze ? W : hw;
    for (int y = 0; y < H; y++) {
        const uint8_t* row = sbs + (size_t)y * W * 3;
        if (unsqueeze) {
            lanczos_row(row, hw, 3, L + (size_t)y * ow * 3, ow);
            lanczos_row(row + (size_t)hw * 3, hw, 3, R + (size_t)y * ow * 3, ow);
        } else {
            memcpy(L + (size_t)y * ow * 3, row, (size_t)hw * 3);
            memcpy(R + (size_t)y * ow * 3, row + (size_t)hw * 3, (size_t)hw * 3);
        }
    }
    return 0;
}

/* depth.py:274-275 (BGR2RGB) then :337-338 (RGB2GRAY): Y = (R*9798 + G*19235 + B*3735 + 2^14) >> 15 */
void orc_bgr_to_gray(const uint8_t* bgr, int n, uint8_t* gray)
{
    for (int i = 0; i < n; i++)
        gray[i] = (uint8_t)((bgr[3 * i + 2] * 9798 + bgr[3 * i + 1] * 19235 + bgr[3 * i] * 3735 + (1 << 14)) >> 15);
}

int orc_sbs_to_gray(const uint8_t* sbs, int W, int H, int unsqueeze, uint8_t* L, uint8_t* R)
{
    if (W % 2) return -1;
    const int ow = unsqueeze ? W : W / 2;
    uint8_t* lb = (uint8_t*)malloc((size_t)ow * H * 3);
    uint8_t* rb = (uint8_t*)malloc((size_t)ow * H * 3);
    if (!lb || !rb) { free(lb); free(rb); return -2; }
    orc_split_sbs(sbs, W, H, unsqueeze, lb, rb);
    orc_bgr_to_gray(lb, ow * H, L);
    orc_bgr_to_gray(rb, ow * H, R);
    free(lb); free(rb);
    return 0;
}

/* depth.py:341 `.astype(np.float32) / 16.0` and :374 `d[d <= 0] = 0` */
void orc_disp_to_depth(const int16_t* disp16, int n, float* out)
{
    for (int i = 0; i < n; i++) { float f = (float)disp16[i] / 16.0f; out[i] = f <= 0.f ? 0.f : f; }
}

/* depth.py:397-406: ((d - min) / (max - min) * 65535).astype(uint16), float32 arithmetic, zeros if flat */
void orc_depth_to_u16(const float* d, int n, uint16_t* out)
{
    float mn = d[0], mx = d[0];
    for (int i = 1; i < n; i++) { if (d[i] < mn) mn = d[i]; if (d[i] > mx) mx = d[i]; }
    if (!(mx > mn)) { memset(out, 0, sizeof(uint16_t) * (size_t)n); return; }
    const float range = mx - mn;
    for (int i = 0; i < n; i++) {
        volatile float a = d[i] - mn;         /* volatile: forbid fused/extended evaluation */
        volatile float b = a / range;
        volatile float c = b * 65535.0f;
        out[i] = (uint16_t)c;
    }
}

/* ------------------------------------------------------------------------------------------
 * depth.py:344-374 -- the "hybrid" blend of the stereo disparity with a monocular depth map.
 *   :353-354  mono = cv2.resize(mono, (W, H))            (default INTER_LINEAR, float32 image)
 *   :359-360  mono_n = (mono - min) / (max - min) * 64   (float32 NumPy arithmetic; skipped if max == min)
 *   :363      combined = 0.7 * disparity + 0.3 * mono_n  (disparity = compute()/16: invalid pixels are -1.0)
 *   :374      combined[combined <= 0] = 0
 * cv2.resize INTER_LINEAR on CV_32F [RECALLED, imgproc/resize.cpp]: scale = 1 / ((double)dst / src);
 * f = (float)((d + 0.5) * scale - 0.5); s = floor(f); f -= s; in x a tap outside the row snaps to the border
 * pixel with weight 0 (s < 0 -> s = 0, f = 0; s >= w-1 -> s = w-1, f = 0); in y the two row indices are
 * clamped and the weights kept.  Horizontal pass first (S[s]*(1-f) + S[s+1]*f in float), then vertical
 * (r0*(1-fy) + r1*fy).  Same-size input is passed through untouched (depth.py:352).
 * ------------------------------------------------------------------------------------------ */
void orc_resize_linear_f32(const float* src, int Ws, int Hs, int Wd, int Hd, float* dst)
{
    if (Ws == Wd && Hs == Hd) { memcpy(dst, src, sizeof(float) * (size_t)Ws * Hs); return; }
    const double scx = 1.0 / ((double)Wd / Ws), scy = 1.0 / ((double)Hd / Hs);
    for (int y = 0; y < Hd; y++) {
        float fy = (float)((y + 0.5) * scy - 0.5);
        int sy = (int)floorf(fy); fy -= (float)sy;
        const int ya = iclamp(sy, 0, Hs - 1), yb = iclamp(sy + 1, 0, Hs - 1);
        const float b0 = 1.f - fy, b1 = fy;
        for (int x = 0; x < Wd; x++) {
            float fx = (float)((x + 0.5) * scx - 0.5);
            int sx = (int)floorf(fx); fx -= (float)sx;
            if (sx < 0) { fx = 0.f; sx = 0; }
            if (sx >= Ws - 1) { fx = 0.f; sx = Ws - 1; }
            const int sx1 = sx + 1 < Ws ? sx + 1 : sx;
            const float a0 = 1.f - fx, a1 = fx;
            volatile float t0 = src[(size_t)ya * Ws + sx] * a0, t1 = src[(size_t)ya * Ws + sx1] * a1;
            volatile float r0 = t0 + t1;
            volatile float u0 = src[(size_t)yb * Ws + sx] * a0, u1 = src[(size_t)yb * Ws + sx1] * a1;
            volatile float r1 = u0 + u1;
            volatile float v0 = r0 * b0, v1 = r1 * b1;
            dst[(size_t)y * Wd + x] = v0 + v1;
        }
    }
}

void orc_mono_blend(const int16_t* disp16, int W, int H, const float* mono, int mw, int mh,
                    float w_stereo, float w_mono, float* out)
{
    const size_t n = (size_t)W * H;
    float* m = (float*)malloc(sizeof(float) * n);
    orc_resize_linear_f32(mono, mw, mh, W, H, m);
    float mn = m[0], mx = m[0];
    for (size_t i = 1; i < n; i++) { if (m[i] < mn) mn = m[i]; if (m[i] > mx) mx = m[i]; }
    const int flat = !(mx > mn);
    const float range = mx - mn;
    for (size_t i = 0; i < n; i++) {
        volatile float d = (float)disp16[i] / 16.0f;
        float c = d;
        if (!flat) {
            volatile float a = m[i] - mn;
            volatile float b = a / range;
            volatile float e = b * 64.0f;
            volatile float s0 = w_stereo * d, s1 = w_mono * e;
            c = s0 + s1;
        }
        out[i] = c <= 0.f ? 0.f : c;
    }
    free(m);
}

/* ------------------------------------------------------------------------------------------
 * Guided-filter joint upsampling (He, Sun, Tang) -- SURVEY Appendix B.1, float64.
 * ------------------------------------------------------------------------------------------ */
void orc_bilinear_resize(const float* src, int Ws, int Hs, int Wd, int Hd, double* dst)
{
    const double sx = (double)Ws / Wd, sy = (double)Hs / Hd;
    for (int y = 0; y < Hd; y++) {
        double fy = (y + 0.5) * sy - 0.5;
        int y0 = (int)floor(fy); double wy = fy - y0;
        int ya = iclamp(y0, 0, Hs - 1), yb = iclamp(y0 + 1, 0, Hs - 1);
        for (int x = 0; x < Wd; x++) {
            double fx = (x + 0.5) * sx - 0.5;
            int x0 = (int)floor(fx); double wx = fx - x0;
            int xa = iclamp(x0, 0, Ws - 1), xb = iclamp(x0 + 1, 0, Ws - 1);
            double top = src[(size_t)ya * Ws + xa] * (1 - wx) + src[(size_t)ya * Ws + xb] * wx;
            double bot = src[(size_t)yb * Ws + xa] * (1 - wx) + src[(size_t)yb * Ws + xb] * wx;
            dst[(size_t)y * Wd + x] = top * (1 - wy) + bot * wy;
        }
    }
}

/* box mean with window clipped at the border and divided by the true count (two separable passes) */
static void box_mean(const double* src, int W, int H, int r, double* dst, double* tmp)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int a = imax(x - r, 0), b = imin(x + r, W - 1);
            double s = 0;
            for (int i = a; i <= b; i++) s += src[(size_t)y * W + i];
            tmp[(size_t)y * W + x] = s;
        }
    for (int y = 0; y < H; y++) {
        int a = imax(y - r, 0), b = imin(y + r, H - 1);
        for (int x = 0; x < W; x++) {
            double s = 0;
            for (int j = a; j <= b; j++) s += tmp[(size_t)j * W + x];
            int cnt = (b - a + 1) * (imin(x + r, W - 1) - imax(x - r, 0) + 1);
            dst[(size_t)y * W + x] = s / cnt;
        }
    }
}

int orc_guided_upscale(const float* depth_lo, int Wlo, int Hlo, const uint8_t* guide, int W, int H,
                       int r, double eps, double* out)
{
    const size_t n = (size_t)W * H;
    double* buf = (double*)malloc(sizeof(double) * n * 8);
    if (!buf) return -2;
    double *I = buf, *p = buf + n, *t0 = buf + 2 * n, *mI = buf + 3 * n, *mp = buf + 4 * n,
           *a = buf + 5 * n, *b = buf + 6 * n, *tmp = buf + 7 * n;
    for (size_t i = 0; i < n; i++) I[i] = guide[i] / 255.0;
    orc_bilinear_resize(depth_lo, Wlo, Hlo, W, H, p);
    box_mean(I, W, H, r, mI, tmp);
    box_mean(p, W, H, r, mp, tmp);
    for (size_t i = 0; i < n; i++) t0[i] = I[i] * p[i];
    box_mean(t0, W, H, r, a, tmp);                       /* a <- mean(I*p) */
    for (size_t i = 0; i < n; i++) t0[i] = I[i] * I[i];
    box_mean(t0, W, H, r, b, tmp);                       /* b <- mean(I*I) */
    for (size_t i = 0; i < n; i++) {
        double cov = a[i] - mI[i] * mp[i], var = b[i] - mI[i] * mI[i];
        a[i] = cov / (var + eps);
        b[i] = mp[i] - a[i] * mI[i];
    }
    box_mean(a, W, H, r, mI, tmp);                       /* mI <- mean(a) */
    box_mean(b, W, H, r, mp, tmp);                       /* mp <- mean(b) */
    for (size_t i = 0; i < n; i++) out[i] = mI[i] * I[i] + mp[i];
    free(buf);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * CREStereo-style local group correlation (SURVEY Appendix B.2 form A), fp32 accumulate.
 * out[g*9+k][y][x] = (1/Cg) * sum_{c in group g} fl[c][y][x] * fr'[c][y+dy][x+dx],
 * fr' = bilinear sample of fr at (x + flow_x, y + flow_y) (zeros outside, like grid_sample
 * padding_mode='zeros', align_corners=True pixel coordinates), window offsets replicate-padded.
 * ------------------------------------------------------------------------------------------ */
static float sample_zero(const float* img, int h, int w, float sx, float sy)
{
    int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
    float wx = sx - x0, wy = sy - y0, acc = 0.f;
    for (int j = 0; j < 2; j++)
        for (int i = 0; i < 2; i++) {
            int xx = x0 + i, yy = y0 + j;
            if (xx < 0 || xx >= w || yy < 0 || yy >= h) continue;
            acc += img[(size_t)yy * w + xx] * (i ? wx : 1.f - wx) * (j ? wy : 1.f - wy);
        }
    return acc;
}

int orc_corr_lookup(const float* fl, const float* fr, const float* flow, int C, int h, int w,
                    int G, int pattern, float* out)
{
    if (C % G) return -1;
    const int Cg = C / G;
    const size_t hw = (size_t)h * w;
    float* warped = (float*)malloc(sizeof(float) * hw * C);
    if (!warped) return -2;
    for (int c = 0; c < C; c++)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
                warped[c * hw + (size_t)y * w + x] =
                    sample_zero(fr + c * hw, h, w, x + flow[(size_t)y * w + x], y + flow[hw + (size_t)y * w + x]);
    for (int g = 0; g < G; g++)
        for (int k = 0; k < 9; k++) {
            int dx = pattern == 0 ? k - 4 : (k % 3) - 1, dy = pattern == 0 ? 0 : (k / 3) - 1;
            float* o = out + (size_t)(g * 9 + k) * hw;
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    int yy = iclamp(y + dy, 0, h - 1), xx = iclamp(x + dx, 0, w - 1);
                    float acc = 0.f;
                    for (int c = 0; c < Cg; c++) {
                        size_t ch = (size_t)(g * Cg + c) * hw;
                        acc += fl[ch + (size_t)y * w + x] * warped[ch + (size_t)yy * w + xx];
                    }
                    o[(size_t)y * w + x] = acc / (float)Cg;
                }
        }
    free(warped);
    return 0;
}
