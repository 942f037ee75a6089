/*
 * match.c -- oracle (test infrastructure, see oracle.h): brute-force Hamming kNN-2.
 *
 * Follows matchFeatures, reference src/CameraPoseEstimator.cpp:200-213:
 *   BFMatcher matcher(NORM_HAMMING, false); matcher.knnMatch(d1, d2, raw, 2);
 *   keep raw[i][0] iff raw[i][0].distance < raw[i][1].distance * ratio.
 * OpenCV 2.4.13's BFMatcher is not vendored; its kNN semantics are fully determined by
 * the metric and the scan order: distance = popcount(q XOR t) over the 32 descriptor
 * bytes (the in-tree statement of the same metric is ThirdParty/DBoW2/DBoW2/FORB.cpp:81-101),
 * train rows scanned in increasing index with strict `<` insertion, so equal distances
 * rank by lower train index.  Parity unpinned (no golden vectors in the reference).
 */
#include "oracle.h"
#include <string.h>

static inline int hamming256(const uint8_t* a, const uint8_t* b)
{
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

int orc_match_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist)
{
    if (!q || !t || !idx || !dist || nq < 0 || nt < 0) return -1;
    if (nt < 2) return -4; /* the reference reads raw_matches[i][1] unguarded (:209) */
    for (int i = 0; i < nq; i++) {
        int d0 = 1 << 30, d1 = 1 << 30, i0 = -1, i1 = -1;
        const uint8_t* qi = q + (size_t)i * 32;
        for (int j = 0; j < nt; j++) {
            int d = hamming256(qi, t + (size_t)j * 32);
            if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
            else if (d < d1) { d1 = d; i1 = j; }
        }
        idx[2 * i] = i0; idx[2 * i + 1] = i1;
        dist[2 * i] = d0; dist[2 * i + 1] = d1;
    }
    return 0;
}

int orc_ratio_filter(const int32_t* idx, const int32_t* dist, int nq, float ratio,
                     int32_t* query_idx, int32_t* train_idx, float* d, int* n_out)
{
    int n = 0;
    for (int i = 0; i < nq; i++) {
        /* DMatch.distance is float; the product is evaluated in float (:209) */
        float d0 = (float)dist[2 * i], d1 = (float)dist[2 * i + 1];
        if (d0 < d1 * ratio) {
            query_idx[n] = i; train_idx[n] = idx[2 * i]; d[n] = d0; n++;
        }
    }
    *n_out = n;
    return 0;
}
