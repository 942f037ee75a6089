// imageio_check -- decodes an image with host/ImageIO and dumps it as raw bytes (CPU-only test hook):
//   imageio_check in.(png|pgm|ppm) out.raw [--depth]   -> stdout "rows cols channels type"
#include <cstdio>
#include <cstring>
#include "ImageIO.h"

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    cv::Mat m;
    std::string err;
    const bool depth = argc > 3 && !std::strcmp(argv[3], "--depth");
    if (!imageio::imread(argv[1], m, depth ? imageio::LOAD_ANYDEPTH_F32 : imageio::LOAD_UNCHANGED, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    FILE* f = std::fopen(argv[2], "wb");
    if (!f) return 2;
    for (int y = 0; y < m.rows; y++) std::fwrite(m.ptr<uint8_t>(y), 1, (size_t)m.cols * m.elemSize(), f);
    std::fclose(f);
    if (argc > 4) { std::string e2; if (!imageio::imwritePng(argv[4], m, &e2)) { std::fprintf(stderr, "%s\n", e2.c_str()); return 1; } }
    std::printf("%d %d %d %d\n", m.rows, m.cols, m.channels(), m.type());
    return 0;
}
