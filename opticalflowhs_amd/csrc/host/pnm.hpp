// pnm.hpp -- minimal binary PGM (P5) / PPM (P6) reader-writer and the drawing primitives the
// reference used from OpenCV highgui / cxcore (cvLoadImage, cvSaveImage, cvCircle, cvLine),
// which do not exist on this platform (baseline JPEG input: jpeg_baseline.hpp).  Host-side plumbing
// only; nothing here is on the hot path.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace pnm {

struct Image {
    int width = 0, height = 0, channels = 0; // channels: 1 (gray) or 3 (RGB order, as in PPM)
    std::vector<uint8_t> data;
    uint8_t *row(int y) { return data.data() + (size_t)y * width * channels; }
    const uint8_t *row(int y) const { return data.data() + (size_t)y * width * channels; }
};

inline bool skip_ws_and_comments(FILE *f)
{
    int c;
    while ((c = fgetc(f)) != EOF) {
        if (c == '#') { while ((c = fgetc(f)) != EOF && c != '\n') {} }
        else if (c != ' ' && c != '\t' && c != '\n' && c != '\r') { ungetc(c, f); return true; }
    }
    return false;
}

inline bool read_int(FILE *f, int &v)
{
    if (!skip_ws_and_comments(f)) return false;
    return fscanf(f, "%d", &v) == 1;
}

inline bool load(const std::string &path, Image &img)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0, 0, 0};
    if (fread(magic, 1, 2, f) != 2 || magic[0] != 'P' || (magic[1] != '5' && magic[1] != '6')) { fclose(f); return false; }
    int w, h, maxv;
    if (!read_int(f, w) || !read_int(f, h) || !read_int(f, maxv) || w <= 0 || h <= 0 || maxv != 255) { fclose(f); return false; }
    fgetc(f); // single whitespace after maxval
    img.width = w; img.height = h; img.channels = magic[1] == '6' ? 3 : 1;
    img.data.resize((size_t)w * h * img.channels);
    const bool ok = fread(img.data.data(), 1, img.data.size(), f) == img.data.size();
    fclose(f);
    return ok;
}

inline bool save(const std::string &path, const Image &img)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "P%c\n%d %d\n255\n", img.channels == 3 ? '6' : '5', img.width, img.height);
    const bool ok = fwrite(img.data.data(), 1, img.data.size(), f) == img.data.size();
    fclose(f);
    return ok;
}

// cvCvtColor(..., CV_BGR2GRAY) arithmetic on an RGB-ordered pixel (see oracle/hs_preproc_oracle.c).
inline void to_gray(const Image &src, Image &gray)
{
    gray.width = src.width; gray.height = src.height; gray.channels = 1;
    gray.data.resize((size_t)src.width * src.height);
    if (src.channels == 1) { gray.data = src.data; return; }
    for (size_t i = 0; i < gray.data.size(); i++) {
        const uint8_t *p = &src.data[3 * i]; // R, G, B
        gray.data[i] = (uint8_t)((1868 * p[2] + 9617 * p[1] + 4899 * p[0] + 8192) >> 14);
    }
}

inline void put(Image &img, int x, int y, uint8_t r, uint8_t g, uint8_t b)
{
    if (x < 0 || y < 0 || x >= img.width || y >= img.height) return;
    uint8_t *p = img.row(y) + 3 * x;
    p[0] = r; p[1] = g; p[2] = b;
}

inline void filled_circle(Image &img, int cx, int cy, int radius, uint8_t r, uint8_t g, uint8_t b)
{
    for (int dy = -radius; dy <= radius; dy++)
        for (int dx = -radius; dx <= radius; dx++)
            if (dx * dx + dy * dy <= radius * radius) put(img, cx + dx, cy + dy, r, g, b);
}

// 8-connected line, rasterised the way OpenCV 2.1's cvLine(thickness 1) does (its LineIterator):
// start at the LEFT end point, error term major - 2*minor, take the diagonal step while the error
// is negative.  With this the drawings are pixel-identical to the pictures the reference wrote.
inline void line(Image &img, int x0, int y0, int x1, int y1, uint8_t r, uint8_t g, uint8_t b)
{
    int dx = x1 - x0, dy = y1 - y0;
    if (dx < 0) { x0 = x1; y0 = y1; dx = -dx; dy = -dy; }
    const int sy = dy < 0 ? -1 : 1;
    dy = abs(dy);
    const bool steep = dy > dx;
    const int major = steep ? dy : dx, minor = steep ? dx : dy;
    int err = major - 2 * minor, x = x0, y = y0;
    for (int i = 0; i <= major; i++) {
        put(img, x, y, r, g, b);
        if (err < 0) { err += 2 * major - 2 * minor; x++; y += sy; }
        else { err -= 2 * minor; if (steep) y += sy; else x++; }
    }
}

} // namespace pnm
