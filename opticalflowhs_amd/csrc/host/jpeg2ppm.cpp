// jpeg2ppm in.jpg out.ppm -- decodes with host/jpeg_baseline.hpp (the CLI's reader); used by tests/test_jpeg.py
#include <cstdio>
#include <string>

#include "jpeg_baseline.hpp"

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: jpeg2ppm in.jpg out.ppm\n"); return 2; }
    pnm::Image img;
    std::string why;
    if (!jpegb::load(argv[1], img, &why)) { fprintf(stderr, "jpeg2ppm: %s: %s\n", argv[1], why.c_str()); return 1; }
    return pnm::save(argv[2], img) ? 0 : 1;
}
