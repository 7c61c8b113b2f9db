// jpeg_baseline.hpp -- baseline (sequential, Huffman, 8-bit) JPEG reader for the drop-in CLI.
//
// The reference reads its inputs with cvLoadImage (highgui -> libjpeg; OpticalFlowOpenCV.cpp:15,18,
// HSOpticalFlowOpenCL.cpp:716,727) and ships them as baseline 4:2:0 JFIF files (city_1.jpg, bunny_1.jpg,
// ...).  highgui does not exist here, so this header restates the published decoding process (ITU-T
// T.81 baseline) with the choices libjpeg makes by default, because those decide the pixel values the
// solver sees:  the "islow" integer inverse DCT (13-bit constants, two passes), "fancy" triangle
// upsampling of the chroma planes (h2v2 and h2v1), and the 16-bit fixed-point YCbCr -> RGB tables.
// tests/test_jpeg.py holds it against PIL's decoder (libjpeg-turbo) bit for bit.
// Supported: 1 or 3 components, sampling 1x1 / 2x1 / 2x2 for luma with 1x1 chroma, restart intervals.
// Not supported (load fails): progressive, arithmetic coding, 12-bit, CMYK, other sampling factors.
// Host-side plumbing only; nothing here is on the hot path.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "pnm.hpp"

namespace jpegb {

struct Huff {
    // canonical code tables: maxcode[k] = largest code of length k (or -1), valptr / mincode as in T.81 F.2.2.3
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    bool present = false;
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int wblocks = 0, hblocks = 0; // blocks per row / column including MCU padding
    std::vector<uint8_t> plane;   // decoded samples, wblocks*8 x hblocks*8
    int dc_pred = 0;
};

struct Reader {
    const uint8_t *p, *end;
    uint32_t bits = 0;
    int nbits = 0;
    bool bad = false;
    int marker = 0; // a marker met inside the entropy-coded segment (0 if none)

    void fill()
    {
        while (nbits <= 24) {
            int b = 0;
            if (marker) b = 0; // feed zeros after a marker, like libjpeg
            else if (p >= end) { b = 0; marker = 0xD9; }
            else {
                b = *p++;
                if (b == 0xFF) {
                    int b2 = p < end ? *p : 0xD9;
                    while (b2 == 0xFF && p + 1 < end) { p++; b2 = *p; } // fill bytes
                    if (b2 == 0) p++;
                    else { marker = b2; p++; b = 0; }
                }
            }
            bits |= (uint32_t)b << (24 - nbits);
            nbits += 8;
        }
    }
    int get(int n)
    {
        if (n == 0) return 0;
        if (nbits < n) fill();
        const int v = (int)(bits >> (32 - n));
        bits <<= n;
        nbits -= n;
        return v;
    }
    int decode(const Huff &h)
    {
        int code = get(1), k = 1;
        while (k <= 16 && (h.maxcode[k] < 0 || code > h.maxcode[k])) { code = (code << 1) | get(1); k++; }
        if (k > 16) { bad = true; return 0; }
        return h.vals[h.valptr[k] + code - h.mincode[k]];
    }
    static int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }
    void restart() { bits = 0; nbits = 0; marker = 0; }
};

static const int kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline uint8_t clamp255(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// libjpeg's "islow" inverse DCT on de-quantised coefficients (natural order), output 8x8 samples.
inline void idct_islow(const int *coef, uint8_t *out, int stride)
{
    constexpr int CB = 13, P1 = 2;
    constexpr long F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633,
                   F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
    auto descale = [](long x, int n) { return (x + (1L << (n - 1))) >> n; };
    long ws[64];
    for (int c = 0; c < 8; c++) { // pass 1: columns
        const int *in = coef + c;
        long *w = ws + c;
        if (!(in[8] | in[16] | in[24] | in[32] | in[40] | in[48] | in[56])) {
            const long dc = (long)in[0] * (1L << P1);
            for (int r = 0; r < 8; r++) w[8 * r] = dc;
            continue;
        }
        long z2 = in[16], z3 = in[48];
        long z1 = (z2 + z3) * F_0_541;
        long tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
        z2 = in[0]; z3 = in[32];
        long tmp0 = (z2 + z3) * (1L << CB), tmp1 = (z2 - z3) * (1L << CB);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[56]; tmp1 = in[40]; tmp2 = in[24]; tmp3 = in[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F_1_175;
        tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
        z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        w[0] = descale(tmp10 + tmp3, CB - P1); w[56] = descale(tmp10 - tmp3, CB - P1);
        w[8] = descale(tmp11 + tmp2, CB - P1); w[48] = descale(tmp11 - tmp2, CB - P1);
        w[16] = descale(tmp12 + tmp1, CB - P1); w[40] = descale(tmp12 - tmp1, CB - P1);
        w[24] = descale(tmp13 + tmp0, CB - P1); w[32] = descale(tmp13 - tmp0, CB - P1);
    }
    for (int r = 0; r < 8; r++) { // pass 2: rows; +128 level shift and clamp
        const long *w = ws + 8 * r;
        uint8_t *o = out + (size_t)r * stride;
        // (libjpeg's shortcut for rows without AC terms gives the same values as the general path)
        long z2 = w[2], z3 = w[6];
        long z1 = (z2 + z3) * F_0_541;
        long tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
        long tmp0 = (w[0] + w[4]) * (1L << CB), tmp1 = (w[0] - w[4]) * (1L << CB);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F_1_175;
        tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
        z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        constexpr int S = CB + P1 + 3;
        o[0] = clamp255((int)descale(tmp10 + tmp3, S) + 128); o[7] = clamp255((int)descale(tmp10 - tmp3, S) + 128);
        o[1] = clamp255((int)descale(tmp11 + tmp2, S) + 128); o[6] = clamp255((int)descale(tmp11 - tmp2, S) + 128);
        o[2] = clamp255((int)descale(tmp12 + tmp1, S) + 128); o[5] = clamp255((int)descale(tmp12 - tmp1, S) + 128);
        o[3] = clamp255((int)descale(tmp13 + tmp0, S) + 128); o[4] = clamp255((int)descale(tmp13 - tmp0, S) + 128);
    }
}

// libjpeg's fancy (triangle) upsampling.  `src` has sw x sh real samples (stride ss); `dst` gets
// 2*sw x (V2 ? 2*sh : sh) samples.  Rows / columns beyond the edge replicate the edge sample.
inline void upsample_fancy(const uint8_t *src, int ss, int sw, int sh, bool v2, std::vector<uint8_t> &dst)
{
    const int dw = 2 * sw, dh = v2 ? 2 * sh : sh;
    dst.assign((size_t)dw * dh, 0);
    for (int y = 0; y < dh; y++) {
        uint8_t *o = dst.data() + (size_t)y * dw;
        if (!v2) { // h2v1: 3/4 nearer + 1/4 farther, rounding alternates (+1, +2)
            const uint8_t *in = src + (size_t)y * ss;
            if (sw == 1) { o[0] = o[1] = in[0]; continue; }
            o[0] = in[0];
            o[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
            for (int x = 1; x < sw - 1; x++) {
                o[2 * x] = (uint8_t)((in[x] * 3 + in[x - 1] + 1) >> 2);
                o[2 * x + 1] = (uint8_t)((in[x] * 3 + in[x + 1] + 2) >> 2);
            }
            o[2 * sw - 2] = (uint8_t)((in[sw - 1] * 3 + in[sw - 2] + 1) >> 2);
            o[2 * sw - 1] = in[sw - 1];
            continue;
        }
        // h2v2: vertical 3/4 + 1/4 first (column sums), then the same horizontally; rounding (+8, +7) >> 4
        const int sy = y >> 1;
        int ny = (y & 1) ? sy + 1 : sy - 1; // the farther source row
        if (ny < 0) ny = 0;
        if (ny >= sh) ny = sh - 1;
        const uint8_t *in0 = src + (size_t)sy * ss, *in1 = src + (size_t)ny * ss;
        if (sw == 1) {
            const int t = in0[0] * 3 + in1[0];
            o[0] = (uint8_t)((t * 4 + 8) >> 4);
            o[1] = (uint8_t)((t * 4 + 7) >> 4);
            continue;
        }
        int thiss = in0[0] * 3 + in1[0], nexts = in0[1] * 3 + in1[1], lasts;
        o[0] = (uint8_t)((thiss * 4 + 8) >> 4);
        o[1] = (uint8_t)((thiss * 3 + nexts + 7) >> 4);
        lasts = thiss; thiss = nexts;
        for (int x = 1; x < sw - 1; x++) {
            nexts = in0[x + 1] * 3 + in1[x + 1];
            o[2 * x] = (uint8_t)((thiss * 3 + lasts + 8) >> 4);
            o[2 * x + 1] = (uint8_t)((thiss * 3 + nexts + 7) >> 4);
            lasts = thiss; thiss = nexts;
        }
        o[2 * sw - 2] = (uint8_t)((thiss * 3 + lasts + 8) >> 4);
        o[2 * sw - 1] = (uint8_t)((thiss * 4 + 7) >> 4);
    }
}

// Plain replication (libjpeg's h2v1_upsample / h2v2_upsample), same interface.
inline void upsample_box(const uint8_t *src, int ss, int sw, int sh, bool v2, std::vector<uint8_t> &dst)
{
    const int dw = 2 * sw, dh = v2 ? 2 * sh : sh;
    dst.assign((size_t)dw * dh, 0);
    for (int y = 0; y < dh; y++) {
        const uint8_t *in = src + (size_t)(v2 ? y >> 1 : y) * ss;
        uint8_t *o = dst.data() + (size_t)y * dw;
        for (int x = 0; x < sw; x++) o[2 * x] = o[2 * x + 1] = in[x];
    }
}

inline bool decode(const std::vector<uint8_t> &file, pnm::Image &img, std::string *why = nullptr)
{
    auto fail = [&](const char *m) { if (why) *why = m; return false; };
    const uint8_t *p = file.data(), *end = p + file.size();
    if (file.size() < 4 || p[0] != 0xFF || p[1] != 0xD8) return fail("not a JPEG file");
    p += 2;
    uint16_t qt[4][64];
    bool qt_ok[4] = {false, false, false, false};
    Huff hdc[4], hac[4];
    std::vector<Component> comp;
    int W = 0, H = 0, hmax = 1, vmax = 1, restart_interval = 0;
    bool have_sof = false;
    while (p + 4 <= end) {
        if (p[0] != 0xFF) { p++; continue; }
        const int m = p[1];
        if (m == 0xFF) { p++; continue; }
        p += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;
        if (p + 2 > end) return fail("truncated marker");
        const int len = (p[0] << 8) | p[1];
        if (len < 2 || p + len > end) return fail("bad segment length");
        const uint8_t *s = p + 2, *se = p + len;
        if (m == 0xDB) { // DQT
            while (s < se) {
                const int pq = s[0] >> 4, tq = s[0] & 15;
                s++;
                if (tq > 3 || s + (pq ? 128 : 64) > se) return fail("bad DQT");
                for (int i = 0; i < 64; i++) {
                    qt[tq][kZigzag[i]] = pq ? (uint16_t)((s[0] << 8) | s[1]) : s[0];
                    s += pq ? 2 : 1;
                }
                qt_ok[tq] = true;
            }
        } else if (m == 0xC4) { // DHT
            while (s < se) {
                const int tc = s[0] >> 4, th = s[0] & 15;
                if (tc > 1 || th > 3 || s + 17 > se) return fail("bad DHT");
                Huff &h = tc ? hac[th] : hdc[th];
                int counts[17], total = 0;
                for (int k = 1; k <= 16; k++) { counts[k] = s[k]; total += s[k]; }
                s += 17;
                if (total > 256 || s + total > se) return fail("bad DHT");
                std::memcpy(h.vals, s, (size_t)total);
                s += total;
                int code = 0, idx = 0;
                for (int k = 1; k <= 16; k++) {
                    h.valptr[k] = idx;
                    h.mincode[k] = code;
                    h.maxcode[k] = counts[k] ? code + counts[k] - 1 : -1;
                    code = (code + counts[k]) << 1;
                    idx += counts[k];
                }
                h.maxcode[17] = 0x7FFFFFFF;
                h.present = true;
            }
        } else if (m == 0xC0 || m == 0xC1) { // SOF0 / SOF1 (extended sequential, Huffman, 8-bit only)
            if (have_sof) return fail("more than one frame header");
            if (se - s < 6 || s[0] != 8) return fail("only 8-bit samples are supported");
            H = (s[1] << 8) | s[2]; W = (s[3] << 8) | s[4];
            hmax = vmax = 1;
            const int n = s[5];
            if (W <= 0 || H <= 0 || (long long)W * H > (1LL << 28) || (n != 1 && n != 3) || se - s < 6 + 3 * n)
                return fail("unsupported frame header");
            comp.resize((size_t)n);
            for (int i = 0; i < n; i++) {
                comp[i].id = s[6 + 3 * i];
                comp[i].h = s[7 + 3 * i] >> 4; comp[i].v = s[7 + 3 * i] & 15;
                comp[i].tq = s[8 + 3 * i];
                if (comp[i].tq > 3) return fail("bad quantisation table index");
                if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4) return fail("bad sampling factor");
                hmax = comp[i].h > hmax ? comp[i].h : hmax;
                vmax = comp[i].v > vmax ? comp[i].v : vmax;
            }
            have_sof = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            return fail("progressive / lossless / arithmetic JPEG is not supported");
        } else if (m == 0xDD) {
            if (se - s < 2) return fail("bad DRI");
            restart_interval = (s[0] << 8) | s[1];
        } else if (m == 0xDA) { // SOS: baseline has exactly one scan with all components interleaved
            if (!have_sof) return fail("scan before frame header");
            if (se - s < 1) return fail("truncated scan header");
            const int n = s[0];
            if (n != (int)comp.size() || se - s < 1 + 2 * n + 3) return fail("unsupported scan layout");
            for (int i = 0; i < n; i++) {
                Component *c = nullptr;
                for (auto &cc : comp) if (cc.id == s[1 + 2 * i]) c = &cc;
                if (!c) return fail("scan names an unknown component");
                c->td = s[2 + 2 * i] >> 4; c->ta = s[2 + 2 * i] & 15;
                if (c->td > 3 || c->ta > 3 || !hdc[c->td].present || !hac[c->ta].present || !qt_ok[c->tq]) return fail("missing table");
            }
            if (comp.size() == 3) {
                if (comp[1].h != 1 || comp[1].v != 1 || comp[2].h != 1 || comp[2].v != 1 || comp[0].h > 2 || comp[0].v > 2 ||
                    (comp[0].h == 1 && comp[0].v == 2))
                    return fail("unsupported chroma sampling");
            } else { hmax = vmax = 1; comp[0].h = comp[0].v = 1; }
            const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
            for (auto &c : comp) {
                c.wblocks = mcux * c.h; c.hblocks = mcuy * c.v;
                c.plane.assign((size_t)c.wblocks * 8 * c.hblocks * 8, 0);
                c.dc_pred = 0;
            }
            Reader r;
            r.p = se; r.end = end;
            int coef[64];
            int until_restart = restart_interval;
            for (int my = 0; my < mcuy; my++)
                for (int mx = 0; mx < mcux; mx++) {
                    if (restart_interval && until_restart == 0) {
                        // byte-align, expect RSTn
                        if (!r.marker) { r.nbits = 0; r.bits = 0; while (r.p + 1 < r.end && !(r.p[0] == 0xFF && r.p[1] >= 0xD0 && r.p[1] <= 0xD7)) r.p++; if (r.p + 1 < r.end) r.p += 2; }
                        r.restart();
                        for (auto &c : comp) c.dc_pred = 0;
                        until_restart = restart_interval;
                    }
                    for (auto &c : comp)
                        for (int by = 0; by < c.v; by++)
                            for (int bx = 0; bx < c.h; bx++) {
                                std::memset(coef, 0, sizeof coef);
                                const int t = r.decode(hdc[c.td]);
                                if (t > 11) return fail("bad DC code");
                                const int diff = t ? Reader::extend(r.get(t), t) : 0;
                                c.dc_pred += diff;
                                coef[0] = c.dc_pred * qt[c.tq][0];
                                for (int k = 1; k < 64;) {
                                    const int rs = r.decode(hac[c.ta]);
                                    const int run = rs >> 4, sz = rs & 15;
                                    if (sz == 0) {
                                        if (run == 15) { k += 16; continue; }
                                        break; // EOB
                                    }
                                    k += run;
                                    if (k > 63) return fail("bad AC run");
                                    coef[kZigzag[k]] = Reader::extend(r.get(sz), sz) * qt[c.tq][kZigzag[k]];
                                    k++;
                                }
                                if (r.bad) return fail("corrupt entropy-coded data");
                                const int px = (mx * c.h + bx) * 8, py = (my * c.v + by) * 8;
                                idct_islow(coef, c.plane.data() + (size_t)py * c.wblocks * 8 + px, c.wblocks * 8);
                            }
                    if (restart_interval) until_restart--;
                }
            // colour conversion
            img.width = W; img.height = H; img.channels = (int)comp.size() == 3 ? 3 : 1;
            img.data.assign((size_t)W * H * img.channels, 0);
            if (comp.size() == 1) {
                for (int y = 0; y < H; y++) std::memcpy(img.row(y), comp[0].plane.data() + (size_t)y * comp[0].wblocks * 8, (size_t)W);
                return true;
            }
            const int cw = (W + comp[0].h - 1) / comp[0].h, ch = (H + comp[0].v - 1) / comp[0].v; // real chroma samples
            // every plane must cover what the conversion below reads (a header that lies about sampling must not over-read)
            if (comp[0].wblocks * 8 < W || comp[0].hblocks * 8 < H || comp[1].wblocks * 8 < cw || comp[1].hblocks * 8 < ch ||
                comp[2].wblocks * 8 < cw || comp[2].hblocks * 8 < ch)
                return fail("component planes smaller than the frame");
            std::vector<uint8_t> cb, cr;
            const uint8_t *pcb, *pcr;
            int cstride;
            if (comp[0].h == 2) {
                // libjpeg uses the triangle filter only for planes wider than 2 samples, plain replication below
                auto up = cw > 2 ? upsample_fancy : upsample_box;
                up(comp[1].plane.data(), comp[1].wblocks * 8, cw, ch, comp[0].v == 2, cb);
                up(comp[2].plane.data(), comp[2].wblocks * 8, cw, ch, comp[0].v == 2, cr);
                pcb = cb.data(); pcr = cr.data(); cstride = 2 * cw;
            } else {
                pcb = comp[1].plane.data(); pcr = comp[2].plane.data(); cstride = comp[1].wblocks * 8;
            }
            auto fix = [](double v) { return (long)(v * 65536.0 + 0.5); };
            const long f_cr_r = fix(1.40200), f_cb_b = fix(1.77200), f_cr_g = fix(0.71414), f_cb_g = fix(0.34414);
            for (int y = 0; y < H; y++) {
                const uint8_t *Y = comp[0].plane.data() + (size_t)y * comp[0].wblocks * 8;
                const uint8_t *CB = pcb + (size_t)y * cstride, *CR = pcr + (size_t)y * cstride;
                uint8_t *o = img.row(y);
                for (int x = 0; x < W; x++) {
                    const int yy = Y[x], b = CB[x] - 128, rr = CR[x] - 128;
                    const int r_ = yy + (int)((f_cr_r * rr + 32768) >> 16);
                    const int b_ = yy + (int)((f_cb_b * b + 32768) >> 16);
                    const int g_ = yy + (int)((-f_cb_g * b + 32768 - f_cr_g * rr) >> 16);
                    o[3 * x] = clamp255(r_); o[3 * x + 1] = clamp255(g_); o[3 * x + 2] = clamp255(b_);
                }
            }
            return true;
        }
        p += len;
    }
    return fail("no scan found");
}

inline bool load(const std::string &path, pnm::Image &img, std::string *why = nullptr)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { if (why) *why = "cannot open file"; return false; }
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    return decode(buf, img, why);
}

} // namespace jpegb

namespace pnm {
// PGM / PPM or baseline JPEG, by magic number: what the reference gets from cvLoadImage for its inputs.
inline bool load_image(const std::string &path, Image &img)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    unsigned char m[2] = {0, 0};
    const bool got = fread(m, 1, 2, f) == 2;
    fclose(f);
    if (!got) return false;
    if (m[0] == 0xFF && m[1] == 0xD8) return jpegb::load(path, img);
    return load(path, img);
}
} // namespace pnm
