/*
 * hs_classic_oracle.c -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * Restatement of the reference's OpenCL ("-cl") discretisation of Horn-Schunck, i.e. what
 * OpticalFlowHS/Kernels.cl computes per pixel, with the missing v update restored
 * (Kernels.cl:84-86 writes only u; SURVEY.md section 9 item 1 -- a bug we do not reproduce):
 *   - Tex2D clamp-to-edge fetch ............... Kernels.cl:2-9
 *   - ComputeDerivativesKernel (2x2x2 cube) ... Kernels.cl:13-39
 *   - u_v_avgKernel (1/6, 1/12 weights) ....... Kernels.cl:43-68
 *   - u_v_updateKernel ........................ Kernels.cl:71-90
 *   - host loop: zero u,v; derivatives once; `iterations` x (avg, update)
 *                                               HSOpticalFlowOpenCL.cpp:321-332, :748-752
 * Only lane 0 of the reference's float4 pixels carries data (HSOpticalFlowOpenCL.cpp:13-20), so
 * planes here are single-channel fp32.  Frames enter as u8 gray, converted with (float) exactly as
 * readInputImage does (HSOpticalFlowOpenCL.cpp:15-20).
 *
 * PINNED BY THE REFERENCE'S RECORDED OUTPUTS, at drawing resolution: the two pictures its OpenCL
 * route wrote (OpticalFlowHS/city_cl_out.jpg, bunny_cl_out.jpg; dot + line where |u| > 0.5) are
 * reproduced pixel for pixel by the as-shipped form below (updateV = 0, alpha 15, 10 sweeps; see
 * tests/refpics.py).  No numeric output is recorded, and an OpenCL compiler may contract a*b+c
 * (FP_CONTRACT defaults ON), so below the drawing's resolution this oracle states the un-contracted
 * fp32 evaluation in source order: "parity unpinned" at the level of the last bits.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline float tex(const float *p, int w, int h, int i, int j)
{
    return p[(size_t)clampi(j, 0, h - 1) * w + clampi(i, 0, w - 1)];
}

/* Ex, Ey, Et: Kernels.cl:25-38, sums in source order, (1.0/4) folded to 0.25f. */
int hs_oracle_classic_derivatives(const uint8_t *imgA, const uint8_t *imgB, int imgStep, int W,
                                  int H, float *Ex, float *Ey, float *Et)
{
    if (!imgA || !imgB || !Ex || !Ey || !Et) return -2;
    if (W <= 0 || H <= 0 || W > imgStep) return -1;
    float *I1 = (float *)malloc((size_t)W * H * 2 * sizeof(float));
    if (!I1) return -3;
    float *I2 = I1 + (size_t)W * H;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            I1[(size_t)y * W + x] = (float)imgA[(size_t)y * imgStep + x];
            I2[(size_t)y * W + x] = (float)imgB[(size_t)y * imgStep + x];
        }
#pragma omp parallel for schedule(static)
    for (int j = 0; j < H; j++)
        for (int i = 0; i < W; i++) {
            const float a00 = tex(I1, W, H, i, j), a10 = tex(I1, W, H, i + 1, j);
            const float a01 = tex(I1, W, H, i, j + 1), a11 = tex(I1, W, H, i + 1, j + 1);
            const float b00 = tex(I2, W, H, i, j), b10 = tex(I2, W, H, i + 1, j);
            const float b01 = tex(I2, W, H, i, j + 1), b11 = tex(I2, W, H, i + 1, j + 1);
            const size_t p = (size_t)j * W + i;
            Ex[p] = 0.25f * (a10 - a00 + a11 - a01 + b10 - b00 + b11 - b01);
            Ey[p] = 0.25f * (a01 - a00 + a11 - a10 + b01 - b00 + b11 - b10);
            Et[p] = 0.25f * (b00 - a00 + b10 - a10 + b01 - a01 + b11 - a11);
        }
    free(I1);
    return 0;
}

/* Full solve, fixed iteration count (the reference has no other stop rule, :750-751).
 * updateV = 1: the intended scheme.  updateV = 0: Kernels.cl exactly as shipped -- line 86 writes
 * u only, so v keeps its initial value (zero, HSOpticalFlowOpenCL.cpp:332) for ever; this form
 * exists to hold the restatement against the pictures the reference's OpenCL route wrote
 * (OpticalFlowHS/city_cl_out.jpg, bunny_cl_out.jpg; tests/test_reference_pictures.py). */
int hs_oracle_classic_ex(const uint8_t *imgA, const uint8_t *imgB, int imgStep, int W, int H,
                         int usePrevious, float *u, float *v, int velStep, float alpha, int iterations,
                         int updateV)
{
    if (!imgA || !imgB || !u || !v) return -2;
    if (W <= 0 || H <= 0 || W > imgStep || (velStep & 3) || W * 4 > velStep || iterations < 0)
        return -1;
    const size_t N = (size_t)W * H;
    const int vs = velStep / 4;
    float *mem = (float *)malloc(N * 7 * sizeof(float));
    if (!mem) return -3;
    float *Ex = mem, *Ey = mem + N, *Et = mem + 2 * N, *ua = mem + 3 * N, *va = mem + 4 * N,
          *uu = mem + 5 * N, *vv = mem + 6 * N;
    int st = hs_oracle_classic_derivatives(imgA, imgB, imgStep, W, H, Ex, Ey, Et);
    if (st) {
        free(mem);
        return st;
    }
    for (int y = 0; y < H; y++) {
        if (usePrevious) {
            memcpy(uu + (size_t)y * W, u + (size_t)y * vs, (size_t)W * sizeof(float));
            memcpy(vv + (size_t)y * W, v + (size_t)y * vs, (size_t)W * sizeof(float));
        } else { /* HSOpticalFlowOpenCL.cpp:331-332 */
            memset(uu + (size_t)y * W, 0, (size_t)W * sizeof(float));
            memset(vv + (size_t)y * W, 0, (size_t)W * sizeof(float));
        }
    }
    const float c6 = (float)(1.0 / 6), c12 = (float)(1.0 / 12);
    const float a2 = alpha * alpha;
    for (int it = 0; it < iterations; it++) {
#pragma omp parallel for schedule(static)
        for (int j = 0; j < H; j++) /* Kernels.cl:55-63 */
            for (int i = 0; i < W; i++) {
                const size_t p = (size_t)j * W + i;
                ua[p] = c6 * (tex(uu, W, H, i - 1, j) + tex(uu, W, H, i + 1, j) +
                              tex(uu, W, H, i, j - 1) + tex(uu, W, H, i, j + 1)) +
                        c12 * (tex(uu, W, H, i - 1, j - 1) + tex(uu, W, H, i + 1, j - 1) +
                               tex(uu, W, H, i - 1, j + 1) + tex(uu, W, H, i + 1, j + 1));
                va[p] = c6 * (tex(vv, W, H, i - 1, j) + tex(vv, W, H, i + 1, j) +
                              tex(vv, W, H, i, j - 1) + tex(vv, W, H, i, j + 1)) +
                        c12 * (tex(vv, W, H, i - 1, j - 1) + tex(vv, W, H, i + 1, j - 1) +
                               tex(vv, W, H, i - 1, j + 1) + tex(vv, W, H, i + 1, j + 1));
            }
#pragma omp parallel for schedule(static)
        for (size_t p = 0; p < N; p++) { /* Kernels.cl:84-86 (+ restored v) */
            float t = Ex[p] * ua[p] + Ey[p] * va[p] + Et[p];
            t /= a2 + Ex[p] * Ex[p] + Ey[p] * Ey[p];
            uu[p] = ua[p] - Ex[p] * t;
            if (updateV) vv[p] = va[p] - Ey[p] * t;
        }
    }
    for (int y = 0; y < H; y++) {
        memcpy(u + (size_t)y * vs, uu + (size_t)y * W, (size_t)W * sizeof(float));
        memcpy(v + (size_t)y * vs, vv + (size_t)y * W, (size_t)W * sizeof(float));
    }
    free(mem);
    return 0;
}

int hs_oracle_classic(const uint8_t *imgA, const uint8_t *imgB, int imgStep, int W, int H,
                      int usePrevious, float *u, float *v, int velStep, float alpha, int iterations)
{
    return hs_oracle_classic_ex(imgA, imgB, imgStep, W, H, usePrevious, u, v, velStep, alpha, iterations, 1);
}
