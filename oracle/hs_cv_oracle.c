/*
 * hs_cv_oracle.c -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * Plain-C restatement of the Horn-Schunck solver that the reference calls as its CPU
 * back-end:  cvCalcOpticalFlowHS(prev, curr, use_previous, velx, vely, lambda, criteria)
 *   - call sites:   OpticalFlowHS/OpticalFlowOpenCV.cpp:29 (disk), :94 (camera)
 *   - declaration:  OpenCV2.1/include/cv.h:481-483, CvTermCriteria OpenCV2.1/include/cxtypes.h:892-915
 *   - arithmetic:   third-party OpenCV 2.1.0 (OpenCV2.1/include/cvver.h:50-52), module cv,
 *                   routine icvCalcOpticalFlowHS_8u32fR.  Its C source is NOT under
 *                   /root/reference; only the Win32 binary Release/cv210.dll is
 *                   (VA 0x1012e040-0x1012f1c1, wrapper at 0x1012f1d0).  The restatement below
 *                   follows the published algorithm as pinned by SURVEY.md section 8c and by a
 *                   static read of that routine's disassembly (never executed or loaded):
 *                     prologue  VA 0x1012e046-0x1012e085  Ilambda = fl32(1/fl32(lambda))
 *                     records   VA 0x1012e7e4-0x1012e839  {xx,xy,yy,xt,yt,alpha} as fp32 stores,
 *                                                          alpha = 1/(yy + (xx + Ilambda))
 *                     update    VA 0x1012ecdf-0x1012ed2d  order of operations of u', v'
 *                     eps       VA 0x1012ed2f-0x1012eda5  Eps = max |old-new| (fp32)
 *                     stop      VA 0x1012f10b-0x1012f14a
 *
 * PINNED BY THE REFERENCE'S RECORDED OUTPUTS, at drawing resolution: the reference holds no test
 * or numeric vector for this path and the DLL cannot be run here, but it does hold the two
 * pictures its CPU route wrote with the real cvCalcOpticalFlowHS (OpticalFlowHS/city_cv_out.jpg,
 * bunny_cv_out.jpg: a dot + line per 4x4 grid point where |u| or |v| > 1).  Re-drawing THIS
 * oracle's flow (lambda 0.1, 10 sweeps, after gray + 3x3 blur) and saving it as JPEG decodes to
 * those pictures with no pixel different: 24 360 drawn / not-drawn decisions and every line end
 * point trunc(x+u/2), trunc(y+v/2) agree, the closest sampled values lying 1e-4..1e-3 from a
 * boundary (tests/refpics.py, tests/test_reference_pictures.py).  Nothing pins the last fp32 bits:
 * below ~1e-3 the oracle rests on the disassembly read, analytic known-answer tests and an
 * independent NumPy restatement (oracle/hs_numpy.py); "parity unpinned" at that level.
 *
 * x87 note: the DLL evaluates expressions on the x87 stack under the Win32 default 53-bit
 * precision control and rounds to fp32 at every store to a float.  `double` temporaries with
 * `float` stores at exactly the same places reproduce that (build with -ffp-contract=off).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define HS_TERMCRIT_ITER 1 /* CV_TERMCRIT_ITER, cxtypes.h:894 */
#define HS_TERMCRIT_EPS 2  /* CV_TERMCRIT_EPS,  cxtypes.h:896 */

#define HS_OK 0
#define HS_BADSIZE (-1) /* CV_BADSIZE_ERR: `or eax,-1` at VA 0x1012e0b0 */
#define HS_NULLPTR (-2) /* CV_NULLPTR_ERR: `mov eax,0xfffffffe` at VA 0x1012e090 */
#define HS_OUTOFMEM (-3)

typedef struct {
    float xx, xy, yy, xt, yt, alpha; /* 24-byte record, VA 0x1012e7f4-0x1012e839 */
} hs_rec;

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Derivatives of frame A (3x3 Sobel / 8, replicate border) and It = B - A.
 * SURVEY.md 8c item 3; all three values are exact in fp32 (multiples of 1/8 below 2^8). */
static inline void hs_gradients(const uint8_t *A, const uint8_t *B, int step, int W, int H, int x,
                                int y, float *gx, float *gy, float *gt)
{
    const int xm = clampi(x - 1, 0, W - 1), xp = clampi(x + 1, 0, W - 1);
    const int ym = clampi(y - 1, 0, H - 1), yp = clampi(y + 1, 0, H - 1);
    const uint8_t *r0 = A + (size_t)ym * step, *r1 = A + (size_t)y * step, *r2 = A + (size_t)yp * step;
    /* vertical [1 2 1] smoothing at columns x-1 and x+1, horizontal difference */
    const int sv_m = r0[xm] + 2 * r1[xm] + r2[xm];
    const int sv_p = r0[xp] + 2 * r1[xp] + r2[xp];
    /* horizontal [1 2 1] smoothing at rows y-1 and y+1, vertical difference */
    const int sh_m = r0[xm] + 2 * r0[x] + r0[xp];
    const int sh_p = r2[xm] + 2 * r2[x] + r2[xp];
    *gx = (float)(sv_p - sv_m) * 0.125f;
    *gy = (float)(sh_p - sh_m) * 0.125f;
    *gt = (float)((int)B[(size_t)y * step + x] - (int)r1[x]);
}

static inline void hs_make_record(hs_rec *r, float gx, float gy, float gt, float Ilambda)
{
    r->xx = gx * gx; /* exact in fp32 */
    r->xy = gx * gy;
    r->yy = gy * gy;
    r->xt = gx * gt;
    r->yt = gy * gt;
    /* VA 0x1012e833-0x1012e839: fadd (xx + Ilambda); faddp (+ yy); fdivr 1/sum; fstp float */
    double s = (double)r->xx + (double)Ilambda;
    s = (double)r->yy + s;
    r->alpha = (float)(1.0 / s);
}

/* One pixel of the Jacobi sweep.  l,r,u,d are the four (clamped) neighbours from iteration k-1. */
static inline void hs_update_pixel(const hs_rec *q, float ul, float ur, float uu, float ud, float vl,
                                   float vr, float vu, float vd, float *un, float *vn)
{
    /* mean in x87 order L,R,U,D (VA 0x1012ee1a-0x1012ee36), * 0.25, stored as float */
    const float ax = (float)(((((double)ul + (double)ur) + (double)uu) + (double)ud) * 0.25);
    const float ay = (float)(((((double)vl + (double)vr) + (double)vu) + (double)vd) * 0.25);
    /* VA 0x1012ecdf-0x1012ed04: ((xy*ay + xx*ax) + xt) * alpha, then ax - that, fstp float */
    const double tx = (((double)q->xy * (double)ay + (double)q->xx * (double)ax) + (double)q->xt) *
                      (double)q->alpha;
    *un = (float)((double)ax - tx);
    /* VA 0x1012ed0d-0x1012ed25: ((xy*ax + yy*ay) + yt) * alpha, then ay - that, float */
    const double ty = (((double)q->xy * (double)ax + (double)q->yy * (double)ay) + (double)q->yt) *
                      (double)q->alpha;
    *vn = (float)((double)ay - ty);
}

static int hs_check_args(const uint8_t *imgA, const uint8_t *imgB, int imgStep, int W, int H,
                         const float *vx, const float *vy, int velStep)
{
    if (!imgA || !imgB || !vx || !vy)
        return HS_NULLPTR; /* VA 0x1012e089-0x1012e0a7 */
    if (W <= 0 || H <= 0 || W > imgStep || (velStep & 3) || W * 4 > velStep)
        return HS_BADSIZE; /* VA 0x1012e0a9-0x1012e0c7 */
    return HS_OK;
}

/* Record table for the whole frame (derivative pass, first half of the routine). */
static hs_rec *hs_build_records(const uint8_t *A, const uint8_t *B, int step, int W, int H,
                                float Ilambda)
{
    hs_rec *II = (hs_rec *)malloc((size_t)W * H * sizeof(hs_rec));
    if (!II)
        return NULL;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float gx, gy, gt;
            hs_gradients(A, B, step, W, H, x, y, &gx, &gy, &gt);
            hs_make_record(&II[(size_t)y * W + x], gx, gy, gt, Ilambda);
        }
    return II;
}

/*
 * Faithful single-threaded form: Jacobi sweep with a two-row line buffer, row y-1 copied back
 * only after row y has been computed (VA 0x1012f085-0x1012f0b8; last row 0x1012f0dd-0x1012f106).
 * velStep and imgStep are in BYTES, as in the OpenCV routine.  Returns 0 or a negative status;
 * *itersDone receives the number of sweeps executed, *lastEps the last Eps (0 if EPS unused).
 */
int hs_oracle_cv_8u32f(const uint8_t *imgA, const uint8_t *imgB, int imgStep, int W, int H,
                       int usePrevious, float *velX, float *velY, int velStep, float lambda,
                       int termType, int maxIter, double epsilon, int *itersDone, float *lastEps)
{
    int st = hs_check_args(imgA, imgB, imgStep, W, H, velX, velY, velStep);
    if (st != HS_OK)
        return st;
    const int vs = velStep / 4;
    const float Ilambda = 1.0f / lambda; /* fld1/fdivrp + fstp float, VA 0x1012e054-0x1012e085 */

    hs_rec *II = hs_build_records(imgA, imgB, imgStep, W, H, Ilambda);
    float *buf = (float *)malloc((size_t)W * 4 * sizeof(float));
    if (!II || !buf) {
        free(II);
        free(buf);
        return HS_OUTOFMEM;
    }
    float *bx[2] = {buf, buf + W}, *by[2] = {buf + 2 * W, buf + 3 * W};

    if (!usePrevious) /* per-row memset, VA 0x1012eb02-0x1012eb5d */
        for (int y = 0; y < H; y++) {
            memset(velX + (size_t)y * vs, 0, (size_t)W * sizeof(float));
            memset(velY + (size_t)y * vs, 0, (size_t)W * sizeof(float));
        }

    int iter = 0, stop = 0;
    float Eps = 0.f;
    while (!stop) {
        Eps = 0.f;
        iter++;
        for (int y = 0; y < H; y++) {
            const float *u1 = velX + (size_t)clampi(y - 1, 0, H - 1) * vs;
            const float *u2 = velX + (size_t)y * vs;
            const float *u3 = velX + (size_t)clampi(y + 1, 0, H - 1) * vs;
            const float *v1 = velY + (size_t)clampi(y - 1, 0, H - 1) * vs;
            const float *v2 = velY + (size_t)y * vs;
            const float *v3 = velY + (size_t)clampi(y + 1, 0, H - 1) * vs;
            float *nx = bx[y & 1], *ny = by[y & 1];
            const hs_rec *q = II + (size_t)y * W;
            for (int x = 0; x < W; x++) {
                const int xm = x > 0 ? x - 1 : 0, xp = x < W - 1 ? x + 1 : W - 1;
                hs_update_pixel(&q[x], u2[xm], u2[xp], u1[x], u3[x], v2[xm], v2[xp], v1[x], v3[x],
                                &nx[x], &ny[x]);
                if (termType & HS_TERMCRIT_EPS) {
                    /* VA 0x1012ed39-0x1012eda5: difference and |.| each stored as float */
                    float t = (float)fabs((double)(float)((double)u2[x] - (double)nx[x]));
                    if (t > Eps) Eps = t;
                    t = (float)fabs((double)(float)((double)v2[x] - (double)ny[x]));
                    if (t > Eps) Eps = t;
                }
            }
            if (y > 0) { /* row y-1 is no longer needed by anyone: write it back */
                memcpy(velX + (size_t)(y - 1) * vs, bx[(y - 1) & 1], (size_t)W * sizeof(float));
                memcpy(velY + (size_t)(y - 1) * vs, by[(y - 1) & 1], (size_t)W * sizeof(float));
            }
        }
        memcpy(velX + (size_t)(H - 1) * vs, bx[(H - 1) & 1], (size_t)W * sizeof(float));
        memcpy(velY + (size_t)(H - 1) * vs, by[(H - 1) & 1], (size_t)W * sizeof(float));

        /* VA 0x1012f10b-0x1012f14a */
        if ((termType & HS_TERMCRIT_ITER) && iter == maxIter) stop = 1;
        if ((termType & HS_TERMCRIT_EPS) && (double)Eps < epsilon) stop = 1;
    }
    if (itersDone) *itersDone = iter;
    if (lastEps) *lastEps = Eps;
    free(II);
    free(buf);
    return HS_OK;
}

/*
 * Same arithmetic, second implementation: whole-plane ping-pong buffers, rows in parallel
 * (OpenMP).  Bit-identical to hs_oracle_cv_8u32f by construction (pure Jacobi); used to time the
 * "all host cores" CPU baseline and as an in-oracle cross-check.  nthreads <= 0: OpenMP default.
 */
int hs_oracle_cv_8u32f_mt(const uint8_t *imgA, const uint8_t *imgB, int imgStep, int W, int H,
                          int usePrevious, float *velX, float *velY, int velStep, float lambda,
                          int termType, int maxIter, double epsilon, int nthreads, int *itersDone,
                          float *lastEps)
{
    int st = hs_check_args(imgA, imgB, imgStep, W, H, velX, velY, velStep);
    if (st != HS_OK)
        return st;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    const int vs = velStep / 4;
    const float Ilambda = 1.0f / lambda;
    hs_rec *II = hs_build_records(imgA, imgB, imgStep, W, H, Ilambda);
    float *pu = (float *)malloc((size_t)W * H * 4 * sizeof(float));
    if (!II || !pu) {
        free(II);
        free(pu);
        return HS_OUTOFMEM;
    }
    const size_t N = (size_t)W * H;
    float *U[2] = {pu, pu + N}, *V[2] = {pu + 2 * N, pu + 3 * N};
    for (int y = 0; y < H; y++) {
        if (usePrevious) {
            memcpy(U[0] + (size_t)y * W, velX + (size_t)y * vs, (size_t)W * sizeof(float));
            memcpy(V[0] + (size_t)y * W, velY + (size_t)y * vs, (size_t)W * sizeof(float));
        } else {
            memset(U[0] + (size_t)y * W, 0, (size_t)W * sizeof(float));
            memset(V[0] + (size_t)y * W, 0, (size_t)W * sizeof(float));
        }
    }
    int iter = 0, stop = 0, cur = 0;
    float Eps = 0.f;
    while (!stop) {
        Eps = 0.f;
        iter++;
        const float *u = U[cur], *v = V[cur];
        float *un = U[cur ^ 1], *vn = V[cur ^ 1];
        float eps_all = 0.f;
#pragma omp parallel for schedule(static) reduction(max : eps_all)
        for (int y = 0; y < H; y++) {
            const size_t o1 = (size_t)clampi(y - 1, 0, H - 1) * W, o2 = (size_t)y * W,
                         o3 = (size_t)clampi(y + 1, 0, H - 1) * W;
            float e = 0.f;
            for (int x = 0; x < W; x++) {
                const int xm = x > 0 ? x - 1 : 0, xp = x < W - 1 ? x + 1 : W - 1;
                hs_update_pixel(&II[o2 + x], u[o2 + xm], u[o2 + xp], u[o1 + x], u[o3 + x],
                                v[o2 + xm], v[o2 + xp], v[o1 + x], v[o3 + x], &un[o2 + x],
                                &vn[o2 + x]);
                if (termType & HS_TERMCRIT_EPS) {
                    float t = (float)fabs((double)(float)((double)u[o2 + x] - (double)un[o2 + x]));
                    if (t > e) e = t;
                    t = (float)fabs((double)(float)((double)v[o2 + x] - (double)vn[o2 + x]));
                    if (t > e) e = t;
                }
            }
            if (e > eps_all) eps_all = e;
        }
        Eps = eps_all;
        cur ^= 1;
        if ((termType & HS_TERMCRIT_ITER) && iter == maxIter) stop = 1;
        if ((termType & HS_TERMCRIT_EPS) && (double)Eps < epsilon) stop = 1;
    }
    for (int y = 0; y < H; y++) {
        memcpy(velX + (size_t)y * vs, U[cur] + (size_t)y * W, (size_t)W * sizeof(float));
        memcpy(velY + (size_t)y * vs, V[cur] + (size_t)y * W, (size_t)W * sizeof(float));
    }
    if (itersDone) *itersDone = iter;
    if (lastEps) *lastEps = Eps;
    free(II);
    free(pu);
    return HS_OK;
}

/* Derivative planes only (tests of the derivative kernel): Ix, Iy, It as W*H fp32, pitch W. */
int hs_oracle_cv_derivatives(const uint8_t *imgA, const uint8_t *imgB, int imgStep, int W, int H,
                             float *Ix, float *Iy, float *It)
{
    if (!imgA || !imgB || !Ix || !Iy || !It) return HS_NULLPTR;
    if (W <= 0 || H <= 0 || W > imgStep) return HS_BADSIZE;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            hs_gradients(imgA, imgB, imgStep, W, H, x, y, &Ix[(size_t)y * W + x],
                         &Iy[(size_t)y * W + x], &It[(size_t)y * W + x]);
    return HS_OK;
}

int hs_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
