// ppm2jpeg in.ppm|in.pgm out.jpg [quality] -- encodes with host/jpeg_encode.hpp (the CLI's writer); used by tests/test_jpeg.py
#include <cstdio>
#include <cstdlib>

#include "jpeg_encode.hpp"

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: ppm2jpeg in.ppm out.jpg [quality]\n"); return 2; }
    pnm::Image img;
    if (!pnm::load(argv[1], img)) { fprintf(stderr, "ppm2jpeg: cannot read %s\n", argv[1]); return 1; }
    return jpegw::save(argv[2], img, argc > 3 ? atoi(argv[3]) : 95) ? 0 : 1;
}
