// png_io.cpp — minimal PNG reader (zlib inflate + PNG unfiltering) standing in for the two cv::imread
// calls on the hot path's input side (pose_functions.cpp:526 colour, :548 IMREAD_GRAYSCALE).
// 8-bit, non-interlaced, colour types 0/2/3/4/6 — what the reference's bundled data uses.
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "o3dr_host.h"

namespace o3dr_host {

static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

Image8 read_png(const std::string& path, bool grayscale)
{
    Image8 out;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return out;
    std::vector<uint8_t> file;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
    fclose(f);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 33 || memcmp(file.data(), sig, 8) != 0) return out;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    for (size_t pos = 8; pos + 12 <= file.size();) {
        const uint32_t len = be32(&file[pos]);
        const uint8_t* type = &file[pos + 4];
        const uint8_t* data = &file[pos + 8];
        if (pos + 12 + len > file.size()) return out;
        if (!memcmp(type, "IHDR", 4)) {
            w = be32(data);
            h = be32(data + 4);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(data, data + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + len;
    }
    if (!w || !h || depth != 8 || interlace != 0) return out;
    int spp;  // samples per pixel in the file
    switch (ctype) {
        case 0: spp = 1; break;
        case 2: spp = 3; break;
        case 3: spp = 1; break;
        case 4: spp = 2; break;
        case 6: spp = 4; break;
        default: return out;
    }
    const size_t stride = (size_t)w * spp;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) return out;
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t ft = raw[(stride + 1) * y];
        const uint8_t* src = &raw[(stride + 1) * y + 1];
        uint8_t* cur = &img[stride * y];
        const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)spp ? cur[i - spp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)spp) ? up[i - spp] : 0;
            int v = src[i];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: return out;
            }
            cur[i] = (uint8_t)v;
        }
    }
    out.rows = (int)h;
    out.cols = (int)w;
    out.channels = grayscale ? 1 : 3;
    out.data.resize((size_t)w * h * out.channels);
    for (size_t p = 0; p < (size_t)w * h; ++p) {
        uint8_t r, g, b;
        const uint8_t* s = &img[p * spp];
        if (ctype == 0 || ctype == 4) {
            r = g = b = s[0];
        } else if (ctype == 3) {
            if ((size_t)s[0] * 3 + 2 >= plte.size()) { out = Image8(); return out; }
            r = plte[s[0] * 3]; g = plte[s[0] * 3 + 1]; b = plte[s[0] * 3 + 2];
        } else {
            r = s[0]; g = s[1]; b = s[2];
        }
        if (grayscale) {
            // grey files pass through unchanged (the disparity PNGs are 8-bit grey); colour input is
            // reduced with OpenCV's fixed-point BT.601 weights — not bit-pinned against libpng's own path
            out.data[p] = (r == g && g == b) ? r : (uint8_t)((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14);
        } else {
            out.data[p * 3] = b;  // cv::imread delivers B,G,R
            out.data[p * 3 + 1] = g;
            out.data[p * 3 + 2] = r;
        }
    }
    return out;
}

}  // namespace o3dr_host
