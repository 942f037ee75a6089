/*
 * orb.c -- oracle (test infrastructure, see oracle.h): ORB detect + describe.
 *
 * Follows FeatureExtractor::process, reference src/FeatureExtractor.cpp:13-31:
 *   detector.detect(frameBuffer, keypoints);            (:17)
 *   extractor.compute(frameBuffer, keypoints, desc);    (:19)
 * with detector/extractor the default-constructed OpenCV 2.4.13 ORB objects
 * (src/FeatureExtractor.h:23-24).  OpenCV is not vendored in the reference and is not
 * present here, so the steps below restate the published 2.4.13 algorithm
 * (modules/features2d/src/orb.cpp, fast.cpp, fast_score.cpp, keypoint.cpp;
 * modules/imgproc/src/imgwarp.cpp, smooth.cpp, filter.cpp, color.cpp) -- PARITY UNPINNED.
 *
 * One definition is this build's own: OpenCV's retainBest leaves the ORDER of the kept
 * keypoints to std::nth_element (library-defined); the kept SET (every keypoint whose
 * response is >= the n-th best) is well defined and is reproduced exactly; inside a
 * level the keypoints are emitted row-major (y, then x), which is FAST's emission order.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

static const int8_t k_pattern[256 * 4] = {
#include "orb_pattern_31.inc"
};

/* cvRound: lrint under the default rounding mode = round half to even */
static inline int cv_round(double v) { return (int)lrint(v); }
static inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
static inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }
static inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

void orc_orb_default_params(orc_orb_params* p)
{
    p->nfeatures = 500; p->scale_factor = 1.2f; p->nlevels = 8; p->edge_threshold = 31;
    p->first_level = 0; p->wta_k = 2; p->score_type = 0; p->patch_size = 31; p->fast_threshold = 20;
}

/* ORB::operator() level loop + computeKeyPoints' per-level feature budget */
int orc_orb_level_sizes(int w, int h, const orc_orb_params* p, int* lw, int* lh, float* sf, float* inv_sf, int* nfeat)
{
    double scaleFactor = (double)p->scale_factor; /* ORB stores the float ctor arg in a double */
    int L = p->nlevels;
    for (int l = 0; l < L; l++) {
        float s = (float)pow(scaleFactor, (double)(l - p->first_level)); /* getScale */
        float inv = 1 / s;
        sf[l] = s; inv_sf[l] = inv;
        lw[l] = cv_round(w * inv); /* Size(cvRound(image.cols*scale), cvRound(image.rows*scale)) */
        lh[l] = cv_round(h * inv);
    }
    float factor = (float)(1.0 / scaleFactor);
    float nd = p->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)L));
    int sum = 0;
    for (int l = 0; l < L - 1; l++) {
        nfeat[l] = cv_round(nd);
        sum += nfeat[l];
        nd *= factor;
    }
    nfeat[L - 1] = p->nfeatures - sum > 0 ? p->nfeatures - sum : 0;
    return L;
}

/* cvtColor(BGR2GRAY) on 8U: fixed point, yuv_shift = 14 */
void orc_bgr2gray(const uint8_t* src, int w, int h, int stride, int channels, uint8_t* dst, int dstride)
{
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * stride;
        uint8_t* d = dst + (size_t)y * dstride;
        if (channels == 1) { memcpy(d, s, (size_t)w); continue; }
        for (int x = 0; x < w; x++, s += channels)
            d[x] = (uint8_t)((s[0] * 1868 + s[1] * 9617 + s[2] * 4899 + (1 << 13)) >> 14);
    }
}

/* resize(..., INTER_LINEAR) for CV_8UC1: 11-bit fixed-point coefficients */
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int* xofs = (int*)malloc(sizeof(int) * (size_t)dw);
    short* alpha = (short*)malloc(sizeof(short) * 2 * (size_t)dw);
    int* rows[2];
    rows[0] = (int*)malloc(sizeof(int) * (size_t)dw);
    rows[1] = (int*)malloc(sizeof(int) * (size_t)dw);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        alpha[2 * dx] = sat_short(cv_round((1.f - fx) * 2048));
        alpha[2 * dx + 1] = sat_short(cv_round(fx * 2048));
    }
    int prev_sy[2] = { -1, -1 };
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy0 = cv_floor(fy);
        fy -= sy0;
        short b0 = sat_short(cv_round((1.f - fy) * 2048));
        short b1 = sat_short(cv_round(fy * 2048));
        int sy[2];
        for (int k = 0; k < 2; k++) {
            int v = sy0 + k;
            sy[k] = v < 0 ? 0 : (v < sh ? v : sh - 1); /* clip(sy0 - ksize2 + 1 + k, 0, ssize.height) */
        }
        for (int k = 0; k < 2; k++) {
            if (sy[k] == prev_sy[k]) continue;
            const uint8_t* S = src + (size_t)sy[k] * sstride;
            int* D = rows[k];
            for (int dx = 0; dx < dw; dx++) {
                int sx = xofs[dx];
                int sx1 = sx + 1 < sw ? sx + 1 : sx; /* dx >= xmax: D = S[sx]*ONE; a1 is 0 there */
                D[dx] = S[sx] * alpha[2 * dx] + S[sx1] * alpha[2 * dx + 1];
            }
            prev_sy[k] = sy[k];
        }
        uint8_t* d = dst + (size_t)dy * dstride;
        const int *S0 = rows[0], *S1 = rows[1];
        for (int x = 0; x < dw; x++)
            d[x] = (uint8_t)((((b0 * (S0[x] >> 4)) >> 16) + ((b1 * (S1[x] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(alpha); free(rows[0]); free(rows[1]);
}

/* getGaussianKernel(7, 2, CV_32F) converted to CV_32S with scale 256 (createSeparableLinearFilter) */
void orc_gaussian_kernel7_q8(int k[7])
{
    const int n = 7;
    const double sigma = 2.0;
    double scale2X = -0.5 / (sigma * sigma);
    float cf[7];
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        cf[i] = (float)exp(scale2X * x * x);
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) {
        cf[i] = (float)(cf[i] * sum);
        k[i] = cv_round(cf[i] * 256.f); /* cvtScale in float, saturate_cast<int> = cvRound */
    }
}

static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

/* GaussianBlur(img, img, Size(7,7), 2, 2, BORDER_REFLECT_101) on 8U: integer row pass,
 * column pass with FixedPtCastEx<int,uchar>(16): (sum + (1 << 15)) >> 16 */
void orc_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride)
{
    int k[7];
    orc_gaussian_kernel7_q8(k);
    int* tmp = (int*)malloc(sizeof(int) * (size_t)w * (size_t)h);
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        int* t = tmp + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            if (x >= 3 && x < w - 3) {
                for (int j = 0; j < 7; j++) acc += k[j] * s[x + j - 3];
            } else {
                for (int j = 0; j < 7; j++) acc += k[j] * s[reflect101(x + j - 3, w)];
            }
            t[x] = acc;
        }
    }
    for (int y = 0; y < h; y++) {
        const int* r[7];
        for (int j = 0; j < 7; j++) r[j] = tmp + (size_t)reflect101(y + j - 3, h) * w;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int j = 0; j < 7; j++) acc += k[j] * r[j][x];
            d[x] = sat_u8((acc + (1 << 15)) >> 16);
        }
    }
    free(tmp);
}

/* Bresenham circle of radius 3, FAST's pixel order (fast_score.cpp makeOffsets) */
static const int ring_dx[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
static const int ring_dy[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };

/* FAST-9/16 segment test + cornerScore<16>: returns 0 for a non-corner, else the score
 * = (largest threshold for which the pixel is still a corner), always >= thr. */
static inline int fast_score(const uint8_t* p, const int* ofs, int thr)
{
    int v = p[0];
    /* high-speed rejection: any arc of 9 contains pixel 0 or 8, and pixel 4 or 12 */
    int d0 = v - p[ofs[0]], d8 = v - p[ofs[8]];
    if (d0 <= thr && d0 >= -thr && d8 <= thr && d8 >= -thr) return 0;
    int d4 = v - p[ofs[4]], d12 = v - p[ofs[12]];
    if (d4 <= thr && d4 >= -thr && d12 <= thr && d12 >= -thr) return 0;
    int d[25];
    for (int k = 0; k < 16; k++) d[k] = v - p[ofs[k]];
    for (int k = 16; k < 25; k++) d[k] = d[k - 16];
    int amax = -256, bmax = -256; /* max over the 16 arcs of min(d) and of min(-d) */
    for (int k = 0; k < 16; k++) {
        int mn = d[k], mx = d[k];
        for (int j = 1; j < 9; j++) {
            if (d[k + j] < mn) mn = d[k + j];
            if (d[k + j] > mx) mx = d[k + j];
        }
        if (mn > amax) amax = mn;
        if (-mx > bmax) bmax = -mx;
    }
    int best = amax > bmax ? amax : bmax;
    if (best <= thr) return 0; /* no arc of 9 all brighter / all darker by more than thr */
    return best - 1;           /* cornerScore: threshold = -b0 - 1 */
}

/* FAST-9/16 score map BEFORE non-max suppression: out[y*w+x] = cornerScore (>= thr) for every pixel that passes the
 * segment test, 0 elsewhere (and in the 3-pixel border FAST_t skips). */
void orc_fast_score_map(const uint8_t* img, int w, int h, int stride, int thr, uint8_t* out)
{
    int ofs[16];
    for (int k = 0; k < 16; k++) ofs[k] = ring_dy[k] * stride + ring_dx[k];
    memset(out, 0, (size_t)w * h);
    /* FAST_t: corners for 3 <= x < w-3, 3 <= y < h-3; everything else scores 0 */
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++)
            out[(size_t)y * w + x] = (uint8_t)fast_score(img + (size_t)y * stride + x, ofs, thr);
}

void orc_fast_nms_map(const uint8_t* img, int w, int h, int stride, int thr, uint8_t* out)
{
    uint8_t* sc = (uint8_t*)malloc((size_t)w * h);
    orc_fast_score_map(img, w, h, stride, thr, sc);
    memset(out, 0, (size_t)w * h);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int s = sc[(size_t)y * w + x];
            if (!s) continue;
            const uint8_t* c = sc + (size_t)y * w + x;
            if (s > c[-1] && s > c[1] && s > c[-w - 1] && s > c[-w] && s > c[-w + 1] &&
                s > c[w - 1] && s > c[w] && s > c[w + 1])
                out[(size_t)y * w + x] = (uint8_t)s;
        }
    free(sc);
}

/* HarrisResponses(img, pts, 7, 0.04f) for one point */
float orc_harris_response(const uint8_t* img, int stride, int x, int y)
{
    const int blockSize = 7, r = blockSize / 2;
    float scale = (1 << 2) * blockSize * 255.0f;
    scale = 1.0f / scale;
    float scale_sq_sq = scale * scale * scale * scale;
    const float harris_k = 0.04f;
    int x0 = x - r, y0 = y - r; /* cvRound(pt - r) of integer-valued floats */
    int a = 0, b = 0, c = 0;
    for (int i = 0; i < blockSize; i++)
        for (int j = 0; j < blockSize; j++) {
            const uint8_t* ptr = img + (size_t)(y0 + i) * stride + (x0 + j);
            int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-stride + 1] - ptr[-stride - 1]) + (ptr[stride + 1] - ptr[stride - 1]);
            int Iy = (ptr[stride] - ptr[-stride]) * 2 + (ptr[stride - 1] - ptr[-stride - 1]) + (ptr[stride + 1] - ptr[-stride + 1]);
            a += Ix * Ix;
            b += Iy * Iy;
            c += Ix * Iy;
        }
    return ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

/* cv::fastAtan2 (2.4.x: 7th-order odd polynomial, degrees) */
float orc_fast_atan2(float y, float x)
{
    static const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

static void make_umax(int umax[17])
{
    const int half = 15;
    int v, v0, vmax = cv_floor(half * sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(half * sqrt(2.f) / 2);
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round(sqrt((double)half * half - v * v));
    for (v = half, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

/* IC_Angle: intensity centroid over the circular patch of radius 15 */
float orc_ic_angle(const uint8_t* img, int stride, int x, int y)
{
    static int umax[17];
    static int init = 0;
    if (!init) { make_umax(umax); init = 1; }
    const int half_k = 15;
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = img + (size_t)y * stride + x;
    for (int u = -half_k; u <= half_k; ++u) m_10 += u * center[u];
    for (int v = 1; v <= half_k; ++v) {
        int v_sum = 0;
        int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* computeOrbDescriptor, WTA_K = 2.  The trigonometry is evaluated in double and rounded
 * to float ((float)cos(angle) with the global-namespace double cos); the rotated
 * coordinates are float products/sums without contraction, rounded half-to-even. */
void orc_orb_descriptor(const uint8_t* img, int stride, int cx, int cy, float angle_deg, uint8_t desc[32])
{
    float angle = angle_deg;
    angle *= (float)(3.1415926535897932384626433832795 / 180.f);
    float a = (float)cos((double)angle), b = (float)sin((double)angle);
    const uint8_t* center = img + (size_t)cy * stride + cx;
    const int8_t* pat = k_pattern;
    for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int j = 0; j < 8; j++, pat += 4) {
            float x0 = pat[0] * a - pat[1] * b, y0 = pat[0] * b + pat[1] * a;
            float x1 = pat[2] * a - pat[3] * b, y1 = pat[2] * b + pat[3] * a;
            int t0 = center[cv_round(y0) * stride + cv_round(x0)];
            int t1 = center[cv_round(y1) * stride + cv_round(x1)];
            val |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

typedef struct { int x, y; float resp; } cand_t;

/* KeyPointsFilter::retainBest as a set: keep k iff fewer than n candidates have a strictly
 * greater response (== response >= the n-th best); size <= n keeps everything, n == 0 clears.
 * Order (row-major) is preserved. */
static int retain_best(cand_t* c, int m, int n)
{
    if (n < 0 || m <= n) return m; /* if( n_points >= 0 && size > n_points ) */
    if (n == 0) return 0;           /*     if( n_points == 0 ) { keypoints.clear(); return; } */
    float* r = (float*)malloc(sizeof(float) * (size_t)m);
    for (int i = 0; i < m; i++) r[i] = c[i].resp;
    /* n-th largest by simple selection on a copy (m is at most a few 10^5) */
    int lo = 0, hi = m - 1, kth = n - 1;
    while (lo < hi) { /* quickselect, descending */
        float pv = r[(lo + hi) / 2];
        int i = lo, j = hi;
        while (i <= j) {
            while (r[i] > pv) i++;
            while (r[j] < pv) j--;
            if (i <= j) { float t = r[i]; r[i] = r[j]; r[j] = t; i++; j--; }
        }
        if (kth <= j) hi = j; else if (kth >= i) lo = i; else break;
    }
    float thr = r[kth];
    free(r);
    int k = 0;
    for (int i = 0; i < m; i++) if (c[i].resp >= thr) c[k++] = c[i];
    return k;
}

typedef struct {
    int L;
    int lw[16], lh[16], nfeat[16];
    float sf[16], inv_sf[16];
    uint8_t* lvl[16];
} pyr_t;

static int build_pyramid(const uint8_t* img, int w, int h, int stride, int channels, const orc_orb_params* p, pyr_t* py)
{
    if (p->nlevels < 1 || p->nlevels > 16 || p->first_level != 0 || p->wta_k != 2 || p->patch_size != 31 ||
        !(p->scale_factor > 1.0f) || (channels != 1 && channels != 3 && channels != 4))
        return -1;
    py->L = orc_orb_level_sizes(w, h, p, py->lw, py->lh, py->sf, py->inv_sf, py->nfeat);
    for (int l = 0; l < py->L; l++) {
        if (py->lw[l] < 1 || py->lh[l] < 1) { for (int j = 0; j < l; j++) free(py->lvl[j]); return -1; }
        py->lvl[l] = (uint8_t*)malloc((size_t)py->lw[l] * py->lh[l]);
        if (l == 0) orc_bgr2gray(img, w, h, stride, channels, py->lvl[0], w);
        else orc_resize_linear_u8(py->lvl[l - 1], py->lw[l - 1], py->lh[l - 1], py->lw[l - 1], py->lvl[l], py->lw[l], py->lh[l], py->lw[l]);
    }
    return 0;
}

int orc_orb_level(const uint8_t* img, int w, int h, int stride, int channels, const orc_orb_params* p,
                  int level, int blurred, uint8_t* out, int* lw, int* lh)
{
    pyr_t py;
    if (build_pyramid(img, w, h, stride, channels, p, &py)) return -1;
    int rc = 0;
    if (level < 0 || level >= py.L) rc = -1;
    else {
        *lw = py.lw[level]; *lh = py.lh[level];
        if (blurred) orc_gaussian_blur7(py.lvl[level], *lw, *lh, *lw, out, *lw);
        else memcpy(out, py.lvl[level], (size_t)*lw * *lh);
    }
    for (int l = 0; l < py.L; l++) free(py.lvl[l]);
    return rc;
}

int orc_orb_extract(const uint8_t* img, int w, int h, int stride, int channels, const orc_orb_params* p,
                    orc_keypoint* kps, uint8_t* desc, int cap, int* n_out)
{
    pyr_t py;
    if (build_pyramid(img, w, h, stride, channels, p, &py)) return -1;
    int n = 0, rc = 0;
    const int edge = p->edge_threshold;
    for (int l = 0; l < py.L && rc == 0; l++) {
        int lw = py.lw[l], lh = py.lh[l];
        const uint8_t* im = py.lvl[l];
        /* --- detect pass (computeKeyPoints) --- */
        cand_t* c = NULL;
        int m = 0;
        if (lw > 2 * edge && lh > 2 * edge && lw > 6 && lh > 6) {
            uint8_t* map = (uint8_t*)malloc((size_t)lw * lh);
            orc_fast_nms_map(im, lw, lh, lw, p->fast_threshold, map);
            c = (cand_t*)malloc(sizeof(cand_t) * ((size_t)(lw / 2 + 1) * (lh / 2 + 1)));
            /* runByImageBorder(keypoints, levelSize, edgeThreshold) */
            for (int y = edge; y < lh - edge; y++)
                for (int x = edge; x < lw - edge; x++)
                    if (map[(size_t)y * lw + x]) { c[m].x = x; c[m].y = y; c[m].resp = (float)map[(size_t)y * lw + x]; m++; }
            free(map);
        }
        int nf = py.nfeat[l];
        if (p->score_type == 0) {
            m = retain_best(c, m, 2 * nf);
            for (int i = 0; i < m; i++) c[i].resp = orc_harris_response(im, lw, c[i].x, c[i].y);
        }
        m = retain_best(c, m, nf);
        /* --- describe pass (second ORB::operator() call with the provided keypoints) --- */
        uint8_t* blur = NULL;
        if (m > 0) {
            blur = (uint8_t*)malloc((size_t)lw * lh);
            orc_gaussian_blur7(im, lw, lh, lw, blur, lw);
        }
        float sf = py.sf[l], inv = py.inv_sf[l];
        for (int i = 0; i < m; i++) {
            float px = (float)c[i].x, py_ = (float)c[i].y;
            if (l != 0) { px *= sf; py_ *= sf; } /* detect output, level-0 coordinates */
            /* compute(): runByImageBorder(_keypoints, image.size(), edgeThreshold) in level-0 coordinates */
            if (!(px >= (float)edge && px < (float)(w - edge) && py_ >= (float)edge && py_ < (float)(h - edge)))
                continue;
            float lx = px, ly = py_;
            if (l != 0) { lx *= inv; ly *= inv; } /* keypoint->pt *= 1/getScale */
            if (n >= cap) { rc = -3; break; }
            orc_keypoint* k = &kps[n];
            k->angle = orc_ic_angle(im, lw, c[i].x, c[i].y); /* computed in the detect pass */
            k->response = c[i].resp;
            k->octave = l;
            k->size = p->patch_size * sf;
            orc_orb_descriptor(blur, lw, cv_round(lx), cv_round(ly), k->angle, desc + (size_t)n * 32);
            if (l != 0) { lx *= sf; ly *= sf; }
            k->x = lx; k->y = ly;
            n++;
        }
        free(blur);
        free(c);
    }
    for (int l = 0; l < py.L; l++) free(py.lvl[l]);
    *n_out = n;
    return rc;
}
