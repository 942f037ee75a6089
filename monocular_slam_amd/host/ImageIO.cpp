#include "ImageIO.h"
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <new>
#include <vector>

namespace imageio {
namespace {

bool fail(std::string* err, const std::string& m) { if (err) *err = m; return false; }

// Header dimensions come from the file: refuse anything a camera frame cannot be before allocating for it
// (the reference's FrameLoader only ever sees 640x480 .. 1920x1080 frames, src/FrameLoader.cpp:171-214).
constexpr int kMaxDim = 16384;
constexpr size_t kMaxSamples = (size_t)1 << 28;   // w * h * channels
bool saneSize(int w, int h, int ch) { return w >= 1 && h >= 1 && w <= kMaxDim && h <= kMaxDim && ch >= 1 && (size_t)w * h * ch <= kMaxSamples; }

bool readFile(const std::string& path, std::vector<uint8_t>& buf)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    in.seekg(0, std::ios::end);
    const std::streamoff n = in.tellg();
    in.seekg(0);
    buf.resize((size_t)n);
    in.read((char*)buf.data(), n);
    return (bool)in;
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// samples[h][w][ch] as uint16 (8-bit files widened), ch = 1..4 in file order (RGB / RGBA / gray / gray+alpha)
bool decodePng(const std::vector<uint8_t>& f, int& w, int& h, int& ch, int& depth, std::vector<uint16_t>& samples, std::string* err)
{
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    if (f.size() < 33 || std::memcmp(f.data(), sig, 8)) return fail(err, "not a PNG file");
    size_t pos = 8;
    int ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    w = h = 0;
    while (pos + 12 <= f.size()) {
        const uint32_t len = be32(&f[pos]);
        const uint8_t* type = &f[pos + 4];
        if (pos + 12 + (size_t)len > f.size()) return fail(err, "truncated PNG chunk");
        const uint8_t* d = &f[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) return fail(err, "bad IHDR");
            w = (int)be32(d); h = (int)be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12];
        } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(d, d + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (w <= 0 || h <= 0 || ctype < 0) return fail(err, "PNG without IHDR");
    if (!saneSize(w, h, 4)) return fail(err, "PNG dimensions out of range");
    if (interlace) return fail(err, "interlaced PNG is not supported");
    int fch;
    switch (ctype) { case 0: fch = 1; break; case 2: fch = 3; break; case 3: fch = 1; break; case 4: fch = 2; break; case 6: fch = 4; break;
                     default: return fail(err, "bad PNG colour type"); }
    if (!(depth == 8 || depth == 16 || (depth < 8 && (ctype == 0 || ctype == 3)))) return fail(err, "unsupported PNG bit depth");
    const size_t bpp_bits = (size_t)fch * depth, row_bytes = ((size_t)w * bpp_bits + 7) / 8, bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;
    std::vector<uint8_t> raw((row_bytes + 1) * (size_t)h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) return fail(err, "PNG inflate failed");
    // undo the scanline filters in place
    std::vector<uint8_t> prev(row_bytes, 0);
    for (int y = 0; y < h; y++) {
        uint8_t* row = &raw[(row_bytes + 1) * (size_t)y];
        const int ft = row[0];
        uint8_t* cur = row + 1;
        for (size_t i = 0; i < row_bytes; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int v = cur[i];
            switch (ft) { case 0: break; case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) >> 1; break; case 4: v += paeth(a, b, c); break;
                          default: return fail(err, "bad PNG filter"); }
            cur[i] = (uint8_t)v;
        }
        std::memcpy(prev.data(), cur, row_bytes);
    }
    ch = ctype == 3 ? 3 : fch;
    samples.assign((size_t)w * h * ch, 0);
    for (int y = 0; y < h; y++) {
        const uint8_t* cur = &raw[(row_bytes + 1) * (size_t)y + 1];
        for (int x = 0; x < w; x++) {
            for (int k = 0; k < fch; k++) {
                uint32_t v;
                if (depth == 16) v = ((uint32_t)cur[((size_t)x * fch + k) * 2] << 8) | cur[((size_t)x * fch + k) * 2 + 1];
                else if (depth == 8) v = cur[(size_t)x * fch + k];
                else { const size_t bit = (size_t)x * depth; v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
                       if (ctype == 0) v = v * 255u / ((1u << depth) - 1u); }
                if (ctype == 3) {
                    if ((size_t)v * 3 + 2 >= plte.size()) return fail(err, "PNG palette index out of range");
                    for (int q = 0; q < 3; q++) samples[((size_t)y * w + x) * 3 + q] = plte[(size_t)v * 3 + q];
                } else samples[((size_t)y * w + x) * ch + k] = (uint16_t)v;
            }
        }
    }
    if (depth < 8 || ctype == 3) depth = 8;
    return true;
}

bool decodePnm(const std::vector<uint8_t>& f, int& w, int& h, int& ch, int& depth, std::vector<uint16_t>& samples, std::string* err)
{
    if (f.size() < 7 || f[0] != 'P' || (f[1] != '5' && f[1] != '6')) return fail(err, "not a binary PGM/PPM file");
    ch = f[1] == '5' ? 1 : 3;
    size_t pos = 2;
    int vals[3], nv = 0;
    while (nv < 3 && pos < f.size()) {
        if (f[pos] == '#') { while (pos < f.size() && f[pos] != '\n') pos++; continue; }
        if (isspace(f[pos])) { pos++; continue; }
        if (!isdigit(f[pos])) return fail(err, "bad PNM header");
        long v = 0;
        while (pos < f.size() && isdigit(f[pos])) { v = v * 10 + (f[pos++] - '0'); if (v > 65535) return fail(err, "PNM header value out of range"); }
        vals[nv++] = (int)v;
    }
    pos++; // the single whitespace after maxval
    if (nv < 3) return fail(err, "bad PNM header");
    w = vals[0]; h = vals[1]; depth = vals[2] > 255 ? 16 : 8;
    if (!saneSize(w, h, ch) || vals[2] < 1) return fail(err, "PNM dimensions / maxval out of range");
    const size_t n = (size_t)w * h * ch, need = n * (depth / 8);
    if (pos + need > f.size()) return fail(err, "truncated PNM data");
    samples.resize(n);
    for (size_t i = 0; i < n; i++) samples[i] = depth == 16 ? (uint16_t)((f[pos + 2 * i] << 8) | f[pos + 2 * i + 1]) : f[pos + i];
    return true;
}

bool hasSuffix(const std::string& s, const char* suf)
{
    const size_t n = std::strlen(suf);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; i++) if (tolower(s[s.size() - n + i]) != suf[i]) return false;
    return true;
}

} // namespace

bool imread(const std::string& path, cv::Mat& out, int flags, std::string* err)
{
    std::vector<uint8_t> f;
    if (!readFile(path, f)) return fail(err, "cannot read " + path);
    int w = 0, h = 0, ch = 0, depth = 8;
    std::vector<uint16_t> s;
    bool ok;
    try {
        if (hasSuffix(path, ".png")) ok = decodePng(f, w, h, ch, depth, s, err);
        else if (hasSuffix(path, ".pgm") || hasSuffix(path, ".ppm") || hasSuffix(path, ".pnm")) ok = decodePnm(f, w, h, ch, depth, s, err);
        else return fail(err, "unsupported image type (PNG and binary PGM/PPM only): " + path);
    } catch (const std::bad_alloc&) { return fail(err, "out of memory decoding " + path); }
    if (!ok) return false;
    // 16-bit files are narrowed to 8 bits here (>> 8); the reference's imread(CV_LOAD_IMAGE_UNCHANGED) would keep CV_16U
    // (src/FrameLoader.cpp:62) -- its ORB stage cannot take such frames either, so nothing on the path depends on it
    if (flags == LOAD_ANYDEPTH_F32) { // depth maps: first channel, native depth, as float
        out.create(h, w, cv::CV_32F);
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) out.at<float>(y, x) = (float)s[((size_t)y * w + x) * ch];
        return true;
    }
    // 8-bit, OpenCV channel order: gray | BGR | BGRA (gray+alpha files drop the alpha, as imread does)
    const int och = ch == 2 ? 1 : ch;
    out.create(h, w, cv::CV_MAKETYPE(cv::CV_8U, och));
    const int shift = depth == 16 ? 8 : 0;
    for (int y = 0; y < h; y++) {
        uint8_t* o = out.ptr<uint8_t>(y);
        for (int x = 0; x < w; x++) {
            const uint16_t* p = &s[((size_t)y * w + x) * ch];
            if (och == 1) o[x] = (uint8_t)(p[0] >> shift);
            else { o[x * och + 0] = (uint8_t)(p[2] >> shift); o[x * och + 1] = (uint8_t)(p[1] >> shift); o[x * och + 2] = (uint8_t)(p[0] >> shift);
                   if (och == 4) o[x * 4 + 3] = (uint8_t)(p[3] >> shift); }
        }
    }
    return true;
}

bool imwritePng(const std::string& path, const cv::Mat& img, std::string* err)
{
    const int ch = img.channels();
    if (img.empty() || cv::cv_depth(img.type()) != cv::CV_8U || (ch != 1 && ch != 3)) return fail(err, "imwritePng: 8-bit gray or BGR only");
    const int w = img.cols, h = img.rows;
    std::vector<uint8_t> raw(((size_t)w * ch + 1) * h);
    for (int y = 0; y < h; y++) {
        uint8_t* r = &raw[((size_t)w * ch + 1) * y];
        r[0] = 0;
        const uint8_t* s = img.ptr<uint8_t>(y);
        for (int x = 0; x < w; x++)
            if (ch == 1) r[1 + x] = s[x];
            else { r[1 + 3 * x] = s[3 * x + 2]; r[2 + 3 * x] = s[3 * x + 1]; r[3 + 3 * x] = s[3 * x]; }
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(clen);
    if (compress2(z.data(), &clen, raw.data(), (uLong)raw.size(), 1) != Z_OK) return fail(err, "deflate failed");
    std::ofstream o(path, std::ios::binary);
    if (!o) return fail(err, "cannot write " + path);
    auto chunk = [&](const char* type, const uint8_t* d, uint32_t n) {
        uint8_t hdr[8] = { (uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n, (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3] };
        o.write((char*)hdr, 8);
        if (n) o.write((const char*)d, n);
        uLong c = crc32(0L, hdr + 4, 4);
        if (n) c = crc32(c, d, n);
        uint8_t cb[4] = { (uint8_t)(c >> 24), (uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c };
        o.write((char*)cb, 4);
    };
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    o.write((const char*)sig, 8);
    uint8_t ihdr[13] = { (uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w, (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h,
                         8, (uint8_t)(ch == 1 ? 0 : 2), 0, 0, 0 };
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), (uint32_t)clen);
    chunk("IEND", nullptr, 0);
    return (bool)o;
}

} // namespace imageio
