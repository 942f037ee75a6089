#include "Synth.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include "Geometry.h"

namespace synth {

const double K_MINITAUR[9] = { 1060.0, 0.0, 913.5, 0.0, 1060.0, 550.2, 0.0, 0.0, 1.0 };

namespace {
inline double uni(geom::Rng& r) { return (double)(r.next() >> 11) * (1.0 / 9007199254740992.0); }
// approximately normal (sum of 4 uniforms, variance-normalised): sensor noise does not need better
inline float noise(geom::Rng& r, float sigma)
{
    const uint64_t v = r.next();
    const float s = (float)((v & 0xFFFF) + ((v >> 16) & 0xFFFF) + ((v >> 32) & 0xFFFF) + (v >> 48)) * (1.0f / 65536.0f) - 2.0f; // var 1/3
    return s * 1.7320508f * sigma;
}
inline uint8_t clip8(float v) { v = std::nearbyint(v); return (uint8_t)(v < 0.f ? 0.f : (v > 255.f ? 255.f : v)); }
} // namespace

void texturedFrame(int w, int h, uint64_t seed, uint8_t* out)
{
    geom::Rng rng(seed * 0x9E3779B97F4A7C15ull + 1);
    const double area = (double)w * h / (640.0 * 480.0);
    const int n_blobs = std::max(40, (int)(400 * area)), n_rects = std::max(20, (int)(200 * area));
    std::vector<float> img((size_t)w * h, 128.0f);
    for (int i = 0; i < n_rects; i++) {   // rectangles first (sharp corners), then blobs on top (smooth texture)
        const int rw = 6 + (int)rng.below(42), rh = 6 + (int)rng.below(42);
        const int x0 = (int)rng.below((uint32_t)std::max(1, w - rw)), y0 = (int)rng.below((uint32_t)std::max(1, h - rh));
        const float v = (float)rng.below(256);
        for (int y = y0; y < std::min(h, y0 + rh); y++) std::fill(img.begin() + (size_t)y * w + x0, img.begin() + (size_t)y * w + std::min(w, x0 + rw), v);
    }
    for (int i = 0; i < n_blobs; i++) {
        const double s = 1.5 + 4.5 * uni(rng);
        const int r = (int)(3 * s) + 1, cx = (int)rng.below((uint32_t)w), cy = (int)rng.below((uint32_t)h);
        const double amp = -110.0 + 220.0 * uni(rng);
        for (int y = std::max(0, cy - r); y < std::min(h, cy + r + 1); y++)
            for (int x = std::max(0, cx - r); x < std::min(w, cx + r + 1); x++)
                img[(size_t)y * w + x] += (float)(amp * std::exp(-((x - cx) * (x - cx) + (y - cy) * (y - cy)) / (2 * s * s)));
    }
    for (size_t i = 0; i < img.size(); i++) out[i] = clip8(img[i] + noise(rng, 2.0f));
}

void shiftedFrame(const uint8_t* img, int w, int h, int dx, int dy, uint64_t seed, uint8_t* out)
{
    geom::Rng rng(seed * 0xD1B54A32D192ED03ull + 7);
    for (int y = 0; y < h; y++) {
        const uint8_t* src = img + (size_t)(((y - dy) % h + h) % h) * w;
        for (int x = 0; x < w; x++) out[(size_t)y * w + x] = clip8((float)src[((x - dx) % w + w) % w] + noise(rng, 1.5f));
    }
}

void pnpProblem(int P, uint64_t seed, const double K[9], int w, int h, double* X, double* x)
{
    geom::Rng rng(seed * 0xA0761D6478BD642Full + 3);
    // small rotation (Rodrigues of a random 0.1 rad vector) and translation
    double rv[3], t[3];
    for (int i = 0; i < 3; i++) { rv[i] = 0.1 * (2 * uni(rng) - 1); t[i] = 0.3 * (2 * uni(rng) - 1); }
    const double th = std::sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]) + 1e-12;
    const double k[3] = { rv[0] / th, rv[1] / th, rv[2] / th }, c = std::cos(th), s = std::sin(th);
    const double R[9] = { c + k[0] * k[0] * (1 - c), k[0] * k[1] * (1 - c) - k[2] * s, k[0] * k[2] * (1 - c) + k[1] * s,
                          k[1] * k[0] * (1 - c) + k[2] * s, c + k[1] * k[1] * (1 - c), k[1] * k[2] * (1 - c) - k[0] * s,
                          k[2] * k[0] * (1 - c) - k[1] * s, k[2] * k[1] * (1 - c) + k[0] * s, c + k[2] * k[2] * (1 - c) };
    for (int i = 0; i < P; i++) {
        // a pixel and a depth in the CAMERA frame, mapped back to the world: every inlier projects inside the image
        const double u = 40 + (w - 80) * uni(rng), v = 40 + (h - 80) * uni(rng), z = 2.0 + 8.0 * uni(rng);
        const double pc[3] = { (u - K[2]) / K[0] * z, (v - K[5]) / K[4] * z, z };
        const double d[3] = { pc[0] - t[0], pc[1] - t[1], pc[2] - t[2] };
        for (int a = 0; a < 3; a++) X[3 * i + a] = R[a] * d[0] + R[3 + a] * d[1] + R[6 + a] * d[2];   // R^T (pc - t)
        if (uni(rng) < 0.3) { x[2 * i] = w * uni(rng); x[2 * i + 1] = h * uni(rng); }               // gross outlier
        else { x[2 * i] = u + noise(rng, 0.5f); x[2 * i + 1] = v + noise(rng, 0.5f); }
    }
}

} // namespace synth
