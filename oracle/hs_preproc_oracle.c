/*
 * hs_preproc_oracle.c -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * Restatement of the two pre-processing steps the reference's CPU route applies before the
 * solver:
 *   - cvCvtColor(img, gray, CV_BGR2GRAY) ......... OpticalFlowHS/OpticalFlowOpenCV.cpp:17,20,78,85
 *   - cvSmooth(img, img, CV_BLUR, 3, 3, 0, 0) .... OpticalFlowHS/OpticalFlowOpenCV.cpp:27-28,92-93
 *     (in place, inside the reference's timed region :26-30)
 * Both are third-party OpenCV 2.1.0 arithmetic (module cv, binary only in Release/cv210.dll; its
 * source is not under /root/reference), restated from the published algorithm:
 *   gray = (1868*B + 9617*G + 4899*R + 8192) >> 14      (fixed point, 14 fractional bits:
 *          0.114, 0.587, 0.299 rounded to 1/16384)
 *   blur = round_half_even( sum_3x3 / 9 ), replicate border (cvSmooth's border mode)
 * Pinned only through the pictures of the CPU route (tests/refpics.py: without the blur they are not
 * reproduced); the reference holds no vector for either; parity is defined on identical u8
 * inputs to the solver, so these only matter for end-to-end runs from colour images.
 */
#include <stdint.h>
#include <stdlib.h>
#include <math.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

int hs_oracle_bgr2gray_u8(const uint8_t *bgr, int bgrStep, int W, int H, uint8_t *gray, int grayStep)
{
    if (!bgr || !gray) return -2;
    if (W <= 0 || H <= 0 || 3 * W > bgrStep || W > grayStep) return -1;
    for (int y = 0; y < H; y++) {
        const uint8_t *s = bgr + (size_t)y * bgrStep;
        uint8_t *d = gray + (size_t)y * grayStep;
        for (int x = 0; x < W; x++)
            d[x] = (uint8_t)((1868 * s[3 * x] + 9617 * s[3 * x + 1] + 4899 * s[3 * x + 2] + 8192) >> 14);
    }
    return 0;
}

/* src and dst may alias (the reference blurs in place); a private copy of src is taken. */
int hs_oracle_box_blur3_u8(const uint8_t *src, int srcStep, int W, int H, uint8_t *dst, int dstStep)
{
    if (!src || !dst) return -2;
    if (W <= 0 || H <= 0 || W > srcStep || W > dstStep) return -1;
    uint8_t *tmp = (uint8_t *)malloc((size_t)W * H);
    if (!tmp) return -3;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) tmp[(size_t)y * W + x] = src[(size_t)y * srcStep + x];
    const double scale = 1.0 / 9.0;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int s = 0;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++)
                    s += tmp[(size_t)clampi(y + dy, 0, H - 1) * W + clampi(x + dx, 0, W - 1)];
            dst[(size_t)y * dstStep + x] = (uint8_t)lrint((double)s * scale);
        }
    free(tmp);
    return 0;
}
