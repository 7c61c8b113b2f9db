// jpeg_encode.hpp -- baseline JPEG writer for the drop-in CLI: what cvSaveImage("x.jpg") does.
//
// The reference saves its flow pictures with cvSaveImage (OpticalFlowOpenCV.cpp:47,
// HSOpticalFlowOpenCL.cpp:771) -> libjpeg with its defaults: JFIF 1.01, quality 95 scaling of the
// Annex K quantisation tables, 4:2:0 chroma (h2v2 box average with alternating rounding bias), the
// "islow" forward DCT, the standard Huffman tables, no restart markers.  This header restates exactly
// that pipeline so that the file the CLI writes is the file the reference would have written; the tests
// compare it byte for byte with PIL's encoder (libjpeg-turbo) and with the reference's own pictures.
// Host-side plumbing only; nothing here is on the hot path.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "pnm.hpp"

namespace jpegw {

static const uint8_t kDcLumBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t kDcLumVals[12] = {0x00, 0x01, 0x02, 0x03, 0x04, 0x05, 0x06, 0x07, 0x08, 0x09, 0x0a, 0x0b};
static const uint8_t kAcLumBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125};
static const uint8_t kAcLumVals[162] = {0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t kDcChrBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t kDcChrVals[12] = {0x00, 0x01, 0x02, 0x03, 0x04, 0x05, 0x06, 0x07, 0x08, 0x09, 0x0a, 0x0b};
static const uint8_t kAcChrBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119};
static const uint8_t kAcChrVals[162] = {0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

static const uint8_t kQuantLum[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57,
                                      69, 56, 14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64,
                                      81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t kQuantChr[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                      99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                      99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
static const int kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffEnc {
    uint16_t code[256];
    uint8_t size[256];
    void build(const uint8_t *bits, const uint8_t *vals)
    {
        std::memset(size, 0, sizeof size);
        int c = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            for (int i = 0; i < bits[len - 1]; i++) { code[vals[k]] = (uint16_t)c; size[vals[k]] = (uint8_t)len; c++; k++; }
            c <<= 1;
        }
    }
};

struct BitWriter {
    std::vector<uint8_t> &out;
    uint32_t acc = 0;
    int n = 0;
    explicit BitWriter(std::vector<uint8_t> &o) : out(o) {}
    void put(unsigned code, int size)
    {
        acc = (acc << size) | (code & ((1u << size) - 1));
        n += size;
        while (n >= 8) {
            const uint8_t b = (uint8_t)(acc >> (n - 8));
            out.push_back(b);
            if (b == 0xFF) out.push_back(0);
            n -= 8;
        }
    }
    void flush() { if (n) put(0x7F, 8 - n); } // pad the last byte with 1-bits
};

// libjpeg's "islow" forward DCT on 8x8 samples (already level-shifted), results scaled up by 8.
inline void fdct_islow(int *d)
{
    constexpr int CB = 13, P1 = 2;
    constexpr long F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633,
                   F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
    auto descale = [](long x, int n) { return (x + (1L << (n - 1))) >> n; };
    for (int pass = 0; pass < 2; pass++) {
        const int step = pass ? 8 : 1, next = pass ? 1 : 8; // pass 0: rows, pass 1: columns
        for (int i = 0; i < 8; i++) {
            int *p = d + i * next;
            long tmp0 = p[0] + p[7 * step], tmp7 = p[0] - p[7 * step], tmp1 = p[step] + p[6 * step], tmp6 = p[step] - p[6 * step];
            long tmp2 = p[2 * step] + p[5 * step], tmp5 = p[2 * step] - p[5 * step], tmp3 = p[3 * step] + p[4 * step], tmp4 = p[3 * step] - p[4 * step];
            const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            if (!pass) {
                p[0] = (int)((tmp10 + tmp11) * (1L << P1));
                p[4 * step] = (int)((tmp10 - tmp11) * (1L << P1));
            } else {
                p[0] = (int)descale(tmp10 + tmp11, P1);
                p[4 * step] = (int)descale(tmp10 - tmp11, P1);
            }
            const int sh = pass ? CB + P1 : CB - P1;
            long z1 = (tmp12 + tmp13) * F_0_541;
            p[2 * step] = (int)descale(z1 + tmp13 * F_0_765, sh);
            p[6 * step] = (int)descale(z1 + tmp12 * (-F_1_847), sh);
            z1 = tmp4 + tmp7;
            long z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
            const long z5 = (z3 + z4) * F_1_175;
            tmp4 *= F_0_298; tmp5 *= F_2_053; tmp6 *= F_3_072; tmp7 *= F_1_501;
            z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
            z3 += z5; z4 += z5;
            p[7 * step] = (int)descale(tmp4 + z1 + z3, sh);
            p[5 * step] = (int)descale(tmp5 + z2 + z4, sh);
            p[3 * step] = (int)descale(tmp6 + z2 + z3, sh);
            p[step] = (int)descale(tmp7 + z1 + z4, sh);
        }
    }
}

inline void put16(std::vector<uint8_t> &o, int v) { o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }

// RGB (3 channels) or gray (1 channel) image -> baseline JFIF bytes.  Colour images are written 4:2:0.
inline std::vector<uint8_t> encode(const pnm::Image &img, int quality = 95)
{
    const int W = img.width, H = img.height, nc = img.channels == 3 ? 3 : 1;
    quality = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    int q[2][64];
    for (int t = 0; t < 2; t++)
        for (int i = 0; i < 64; i++) {
            long v = ((long)(t ? kQuantChr[i] : kQuantLum[i]) * scale + 50) / 100;
            q[t][i] = (int)(v < 1 ? 1 : (v > 255 ? 255 : v));
        }
    std::vector<uint8_t> o;
    o.reserve((size_t)W * H / 2 + 1024);
    // SOI, JFIF APP0 (version 1.01, no units, 1:1, no thumbnail)
    const uint8_t head[] = {0xFF, 0xD8, 0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
    o.insert(o.end(), head, head + sizeof head);
    for (int t = 0; t < (nc == 3 ? 2 : 1); t++) { // one DQT segment per table, zigzag order
        o.push_back(0xFF); o.push_back(0xDB); put16(o, 67); o.push_back((uint8_t)t);
        for (int i = 0; i < 64; i++) o.push_back((uint8_t)q[t][kZigzag[i]]);
    }
    o.push_back(0xFF); o.push_back(0xC0); put16(o, 8 + 3 * nc); o.push_back(8); put16(o, H); put16(o, W); o.push_back((uint8_t)nc);
    if (nc == 3) { const uint8_t c[] = {1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1}; o.insert(o.end(), c, c + 9); }
    else { const uint8_t c[] = {1, 0x11, 0}; o.insert(o.end(), c, c + 3); }
    auto dht = [&](int cls_id, const uint8_t *bits, const uint8_t *vals, int nvals) {
        o.push_back(0xFF); o.push_back(0xC4); put16(o, 2 + 1 + 16 + nvals); o.push_back((uint8_t)cls_id);
        o.insert(o.end(), bits, bits + 16); o.insert(o.end(), vals, vals + nvals);
    };
    dht(0x00, kDcLumBits, kDcLumVals, (int)sizeof kDcLumVals);
    dht(0x10, kAcLumBits, kAcLumVals, (int)sizeof kAcLumVals);
    if (nc == 3) {
        dht(0x01, kDcChrBits, kDcChrVals, (int)sizeof kDcChrVals);
        dht(0x11, kAcChrBits, kAcChrVals, (int)sizeof kAcChrVals);
    }
    o.push_back(0xFF); o.push_back(0xDA); put16(o, 6 + 2 * nc); o.push_back((uint8_t)nc);
    if (nc == 3) { const uint8_t c[] = {1, 0x00, 2, 0x11, 3, 0x11}; o.insert(o.end(), c, c + 6); }
    else { o.push_back(1); o.push_back(0x00); }
    o.push_back(0); o.push_back(63); o.push_back(0);

    // colour conversion (libjpeg's 16-bit fixed-point tables) into planes padded to whole MCUs by edge replication
    const int mcu = nc == 3 ? 16 : 8;
    const int PW = (W + mcu - 1) / mcu * mcu, PH = (H + mcu - 1) / mcu * mcu;
    std::vector<uint8_t> Y((size_t)PW * PH), Cb, Cr;
    if (nc == 3) { Cb.resize((size_t)PW * PH); Cr.resize((size_t)PW * PH); }
    auto fix = [](double v) { return (long)(v * 65536.0 + 0.5); };
    const long half = 32768, off = 128L << 16;
    for (int y = 0; y < PH; y++) {
        const uint8_t *row = img.row(y < H ? y : H - 1);
        for (int x = 0; x < PW; x++) {
            const uint8_t *p = row + (size_t)(x < W ? x : W - 1) * nc;
            if (nc == 1) { Y[(size_t)y * PW + x] = p[0]; continue; }
            const long r = p[0], g = p[1], b = p[2];
            Y[(size_t)y * PW + x] = (uint8_t)((fix(0.29900) * r + fix(0.58700) * g + fix(0.11400) * b + half) >> 16);
            Cb[(size_t)y * PW + x] = (uint8_t)((-fix(0.16874) * r - fix(0.33126) * g + fix(0.50000) * b + off + half - 1) >> 16);
            Cr[(size_t)y * PW + x] = (uint8_t)((fix(0.50000) * r - fix(0.41869) * g - fix(0.08131) * b + off + half - 1) >> 16);
        }
    }
    // 4:2:0: box average of 2x2 with the rounding bias alternating 1, 2, 1, 2 ... along a row
    std::vector<uint8_t> cb2, cr2;
    const int CW = PW / 2, CHh = PH / 2;
    if (nc == 3) {
        cb2.resize((size_t)CW * CHh); cr2.resize((size_t)CW * CHh);
        // libjpeg pads the FULL-resolution rows only up to a whole row pair, downsamples, and then pads
        // the DOWNSAMPLED plane to the MCU height by repeating its last row
        const int real_rows = (H + 1) / 2;
        for (int y = 0; y < CHh; y++)
            for (int x = 0; x < CW; x++) {
                if (y >= real_rows) {
                    cb2[(size_t)y * CW + x] = cb2[(size_t)(real_rows - 1) * CW + x];
                    cr2[(size_t)y * CW + x] = cr2[(size_t)(real_rows - 1) * CW + x];
                    continue;
                }
                const int bias = 1 + (x & 1);
                const size_t a = (size_t)(2 * y) * PW + 2 * x, b = a + PW;
                cb2[(size_t)y * CW + x] = (uint8_t)((Cb[a] + Cb[a + 1] + Cb[b] + Cb[b + 1] + bias) >> 2);
                cr2[(size_t)y * CW + x] = (uint8_t)((Cr[a] + Cr[a + 1] + Cr[b] + Cr[b + 1] + bias) >> 2);
            }
    }
    HuffEnc hdc[2], hac[2];
    hdc[0].build(kDcLumBits, kDcLumVals); hac[0].build(kAcLumBits, kAcLumVals);
    hdc[1].build(kDcChrBits, kDcChrVals); hac[1].build(kAcChrBits, kAcChrVals);
    BitWriter bw(o);
    int pred[3] = {0, 0, 0};
    // One block: forward DCT, quantisation, entropy coding.  dummy_dc >= -0x7FFFFFF: a dummy block (all AC
    // zero, DC = that value) -- libjpeg pads an MCU beyond the component's own block grid with those, not
    // with transformed padding samples.  Returns the quantised DC.
    constexpr int kReal = -0x7FFFFFFF - 1;
    auto block = [&](const uint8_t *src, int stride, int tbl, int &dcpred, int dummy_dc) -> int {
        int zz[64];
        if (dummy_dc != kReal) {
            std::memset(zz, 0, sizeof zz);
            zz[0] = dummy_dc;
        } else {
            int d[64];
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) d[8 * y + x] = (int)src[(size_t)y * stride + x] - 128;
            fdct_islow(d);
            for (int i = 0; i < 64; i++) { // quantise: rounded division of the 8x-scaled coefficient
                const int k = kZigzag[i];
                const long qv = (long)q[tbl][k] << 3;
                long t = d[k];
                if (t < 0) { t = -t; t += qv >> 1; t = t >= qv ? t / qv : 0; t = -t; }
                else { t += qv >> 1; t = t >= qv ? t / qv : 0; }
                zz[i] = (int)t;
            }
        }
        auto nbits = [](int v) { int a = v < 0 ? -v : v, n = 0; while (a) { n++; a >>= 1; } return n; };
        int diff = zz[0] - dcpred;
        dcpred = zz[0];
        int n = nbits(diff);
        bw.put(hdc[tbl].code[n], hdc[tbl].size[n]);
        if (n) bw.put((unsigned)(diff < 0 ? diff - 1 : diff), n);
        int run = 0;
        for (int i = 1; i < 64; i++) {
            const int v = zz[i];
            if (!v) { run++; continue; }
            while (run > 15) { bw.put(hac[tbl].code[0xF0], hac[tbl].size[0xF0]); run -= 16; }
            n = nbits(v);
            bw.put(hac[tbl].code[(run << 4) | n], hac[tbl].size[(run << 4) | n]);
            bw.put((unsigned)(v < 0 ? v - 1 : v), n);
            run = 0;
        }
        if (run) bw.put(hac[tbl].code[0], hac[tbl].size[0]);
        return zz[0];
    };
    const int ybw = (W + 7) / 8, ybh = (H + 7) / 8; // the luma component's own block grid
    for (int my = 0; my < PH / mcu; my++)
        for (int mx = 0; mx < PW / mcu; mx++) {
            if (nc == 1) { block(Y.data() + (size_t)my * 8 * PW + mx * 8, PW, 0, pred[0], kReal); continue; }
            int last_dc = 0; // quantised DC of the previous luma block of this MCU
            for (int by = 0; by < 2; by++)
                for (int bx = 0; bx < 2; bx++) {
                    const bool dummy = mx * 2 + bx >= ybw || my * 2 + by >= ybh;
                    last_dc = block(Y.data() + (size_t)(my * 16 + by * 8) * PW + mx * 16 + bx * 8, PW, 0, pred[0], dummy ? last_dc : kReal);
                }
            block(cb2.data() + (size_t)my * 8 * CW + mx * 8, CW, 1, pred[1], kReal);
            block(cr2.data() + (size_t)my * 8 * CW + mx * 8, CW, 1, pred[2], kReal);
        }
    bw.flush();
    o.push_back(0xFF); o.push_back(0xD9);
    return o;
}

inline bool save(const std::string &path, const pnm::Image &img, int quality = 95)
{
    const std::vector<uint8_t> bytes = encode(img, quality);
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    fclose(f);
    return ok;
}

} // namespace jpegw

namespace pnm {
// By extension, like cvSaveImage: ".jpg" / ".jpeg" -> baseline JPEG (quality 95, 4:2:0), else PGM / PPM.
inline bool save_image(const std::string &path, const Image &img)
{
    auto ends = [&](const char *e) {
        const size_t n = std::strlen(e);
        if (path.size() < n) return false;
        for (size_t i = 0; i < n; i++) {
            const char c = path[path.size() - n + i];
            if ((c >= 'A' && c <= 'Z' ? c + 32 : c) != e[i]) return false;
        }
        return true;
    };
    if (ends(".jpg") || ends(".jpeg")) return jpegw::save(path, img);
    return save(path, img);
}
} // namespace pnm
