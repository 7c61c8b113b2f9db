// hs_kernels.hip.h -- hand-written CDNA4 (gfx950) kernels of the Horn-Schunck hot path.
//
// What the path computes is fixed by the reference:
//   a1  derivative pass      cvCalcOpticalFlowHS, first half  (cv210.dll VA 0x1012e25e-0x1012eb02)
//                            replaces ComputeDerivativesKernel (OpticalFlowHS/Kernels.cl:13-39)
//   a2  Jacobi u/v update    cvCalcOpticalFlowHS, second half (cv210.dll VA 0x1012ebd3-0x1012f14a)
//                            replaces u_v_avgKernel + u_v_updateKernel (Kernels.cl:43-90)
// How it is computed is MI355X-first:
//   * planar layout, one 32-bit word of packed derivatives per pixel (lossless: 8*Ix and 8*Iy are
//     integers in [-1020,1020] -> 11 bits each, It in [-255,255] -> 10 bits), so one Jacobi
//     sweep moves 4 + 8 + 8 bytes per pixel instead of the 28 "algorithmic" bytes;
//   * 16-byte-per-lane row-coalesced global accesses (4 pixels per lane);
//   * the fused kernel keeps a tile (core + halo) of u,v in LDS and the pixel coefficients in
//     VGPRs and runs T Jacobi sweeps per launch (temporal blocking): HBM/L2 traffic per sweep
//     drops by ~T, the sweep itself runs out of LDS + registers;
//   * wavefront (64-lane) DPP shifts fetch the left/right neighbours held by adjacent lanes.
// All arithmetic is fp32 with explicit fmaf; the file is compiled with -ffp-contract=off so that
// every kernel variant produces bit-identical flow (tests rely on that).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hsk {

// ------------------------------------------------------------------------------------------
// shared per-pixel arithmetic
// ------------------------------------------------------------------------------------------

// Packed derivative word: bits [10:0] = 8*Ix, [21:11] = 8*Iy, [31:22] = It (two's complement).
__device__ __forceinline__ uint32_t pack_deriv(int ix8, int iy8, int it)
{
    return ((uint32_t)ix8 & 0x7FFu) | (((uint32_t)iy8 & 0x7FFu) << 11) | ((uint32_t)it << 22);
}

__device__ __forceinline__ void unpack_deriv(uint32_t c, float &Ix, float &Iy, float &It)
{
    Ix = (float)(((int)(c << 21)) >> 21) * 0.125f;
    Iy = (float)(((int)(c << 10)) >> 21) * 0.125f;
    It = (float)(((int)c) >> 22);
}

// alpha of the oracle's record (cv210.dll VA 0x1012e833-0x1012e839): 1/(1/lambda + Ix^2 + Iy^2).
// Ix^2 + Iy^2 is exact in fp32 (multiple of 1/64 below 2^15); one rounding in the sum, one in
// the (correctly rounded) division.
__device__ __forceinline__ float alpha_of(float Ix, float Iy, float ilambda)
{
    const float q = Ix * Ix + Iy * Iy;
    return 1.0f / (ilambda + q);
}

// Per-pixel coefficients of the sweep, derived once per launch from the packed derivatives:
//   al = Ix*s, be = Iy*s, ga = It*s   with   s = sqrt(alpha)      (sqrt correctly rounded)
// so that the oracle's update  u' = ub - Ix*(Ix*ub + Iy*vb + It)*alpha  becomes
//   q = al*ub + be*vb + ga,   u' = ub - al*q,   v' = vb - be*q
// -- the same linear map with the factor alpha split evenly over its two uses: three registers
// and four fused multiply-adds per pixel instead of four registers and five operations.
__device__ __forceinline__ void sweep_coefs(uint32_t c, float ilambda, float &al, float &be, float &ga)
{
    float Ix, Iy, It;
    unpack_deriv(c, Ix, Iy, It);
    const float s = sqrtf(alpha_of(Ix, Iy, ilambda));
    al = Ix * s;
    be = Iy * s;
    ga = It * s;
}

// One Jacobi update in the form above.
//
// Canonical order of the 4-neighbour sum (every kernel uses it, so all variants agree bit for bit):
//   even image column:  ((R + (U + D)) + L) * 0.25      odd column:  ((L + (U + D)) + R) * 0.25
// This is what the packed-fp32 strip kernel computes without any register shuffles, and it is
// symmetric under the even reflection x -> -1-x (parity and the roles of L and R flip together),
// which the reflection halo of the strip kernel needs.
template <int ODD>
__device__ __forceinline__ void update_cv(float uL, float uR, float uU, float uD, float vL,
                                          float vR, float vU, float vD, float al, float be,
                                          float ga, float &un, float &vn)
{
    const float su = ODD ? ((uL + (uU + uD)) + uR) : ((uR + (uU + uD)) + uL);
    const float sv = ODD ? ((vL + (vU + vD)) + vR) : ((vR + (vU + vD)) + vL);
    const float ub = su * 0.25f;
    const float vb = sv * 0.25f;
    const float q = __fmaf_rn(al, ub, __fmaf_rn(be, vb, ga));
    un = __fmaf_rn(-al, q, ub);
    vn = __fmaf_rn(-be, q, vb);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8, each with its own
// 4 MiB L2).  This bijection hands every XCD one CONTIGUOUS range of tiles, so neighbouring tiles
// -- which re-read each other's halo rows and columns -- share an L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_contiguous_tile(int b, int nb)
{
    const int q = nb >> 3, r = nb & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// max over the 64 lanes of a wavefront
__device__ __forceinline__ float wave_max(float x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}

// Maximum of a NON-NEGATIVE value over the wavefront, returned wave-uniform.  DPP only (no LDS
// permutes): prefix-doubling row_shr 1/2/4/8 leaves each 16-lane row's maximum in its lane 15,
// row_bcast:15 / row_bcast:31 carry it on to lane 63.  Lanes without a source read 0 (bound_ctrl,
// old = 0), the identity for non-negative inputs.
__device__ __forceinline__ float wave_max_nonneg(float x)
{
#define HS_DPP_MAX(ctrl, rowmask)                                                                  \
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rowmask, 0xF, true)))
    HS_DPP_MAX(0x111, 0xF); // row_shr:1
    HS_DPP_MAX(0x112, 0xF); // row_shr:2
    HS_DPP_MAX(0x114, 0xF); // row_shr:4
    HS_DPP_MAX(0x118, 0xF); // row_shr:8
    HS_DPP_MAX(0x142, 0xA); // row_bcast:15 into rows 1 and 3
    HS_DPP_MAX(0x143, 0xC); // row_bcast:31 into rows 2 and 3
#undef HS_DPP_MAX
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

// ------------------------------------------------------------------------------------------
// a1: derivative pass, CV mode.  One lane = 4 consecutive pixels of one row.
//     reads 2 B/pixel (u8 A with a 3x3 neighbourhood from L1/L2, u8 B), writes 4 B/pixel.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_deriv_cv(const uint8_t *__restrict__ A,
                                                  const uint8_t *__restrict__ B,
                                                  uint32_t *__restrict__ coef, int W, int H, int P,
                                                  long long plane)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= W || y >= H) return;
    const long long base = (long long)blockIdx.z * plane;
    const uint8_t *a0 = A + base + (long long)clampi(y - 1, 0, H - 1) * P;
    const uint8_t *a1 = A + base + (long long)y * P;
    const uint8_t *a2 = A + base + (long long)clampi(y + 1, 0, H - 1) * P;
    const uint8_t *b1 = B + base + (long long)y * P;

    int r0[6], r1[6], r2[6], bb[4];
    if (x0 > 0 && x0 + 4 < W) { // interior: aligned 32-bit word + one byte either side
        const uint32_t w0 = *(const uint32_t *)(a0 + x0), w1 = *(const uint32_t *)(a1 + x0),
                       w2 = *(const uint32_t *)(a2 + x0), wb = *(const uint32_t *)(b1 + x0);
        r0[0] = a0[x0 - 1]; r1[0] = a1[x0 - 1]; r2[0] = a2[x0 - 1];
        r0[5] = a0[x0 + 4]; r1[5] = a1[x0 + 4]; r2[5] = a2[x0 + 4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            r0[k + 1] = (w0 >> (8 * k)) & 0xFF;
            r1[k + 1] = (w1 >> (8 * k)) & 0xFF;
            r2[k + 1] = (w2 >> (8 * k)) & 0xFF;
            bb[k] = (wb >> (8 * k)) & 0xFF;
        }
    } else { // image border: replicate by clamping every column index
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int xc = clampi(x0 + k - 1, 0, W - 1);
            r0[k] = a0[xc]; r1[k] = a1[xc]; r2[k] = a2[xc];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) bb[k] = b1[clampi(x0 + k, 0, W - 1)];
    }
    uint32_t out[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // Sobel (not yet divided by 8): vertical [1 2 1] at columns x-1, x+1; horizontal at rows y-1, y+1
        const int ix8 = (r0[k + 2] + 2 * r1[k + 2] + r2[k + 2]) - (r0[k] + 2 * r1[k] + r2[k]);
        const int iy8 = (r2[k] + 2 * r2[k + 1] + r2[k + 2]) - (r0[k] + 2 * r0[k + 1] + r0[k + 2]);
        const int it = bb[k] - r1[k + 1];
        out[k] = pack_deriv(ix8, iy8, it);
    }
    *(uint4 *)(coef + base + (long long)y * P + x0) = make_uint4(out[0], out[1], out[2], out[3]);
}

// Decode the packed plane to three fp32 planes (hsflow_get_derivatives; not on the hot path).
__global__ __launch_bounds__(256) void k_unpack_deriv(const uint32_t *__restrict__ coef,
                                                      float *__restrict__ dx,
                                                      float *__restrict__ dy,
                                                      float *__restrict__ dt, int W, int H, int P)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    float Ix, Iy, It;
    unpack_deriv(coef[(long long)y * P + x], Ix, Iy, It);
    dx[(long long)y * W + x] = Ix;
    dy[(long long)y * W + x] = Iy;
    dt[(long long)y * W + x] = It;
}

// ------------------------------------------------------------------------------------------
// a2, form 1: one Jacobi sweep per launch straight from global memory ("simple").
//     One lane = 4 consecutive pixels; block = 64 lanes x 4 rows.  Up/down rows and the two
//     side pixels come from L1/L2.  20 B/pixel of compulsory traffic (4 coef + 8 read + 8 write).
// ------------------------------------------------------------------------------------------
template <bool EPS, bool ZERO>
__global__ __launch_bounds__(256) void k_jacobi_simple(const uint32_t *__restrict__ coef,
                                                       const float *__restrict__ u_in,
                                                       const float *__restrict__ v_in,
                                                       float *__restrict__ u_out,
                                                       float *__restrict__ v_out, int W, int H,
                                                       int P, long long plane, float ilambda,
                                                       unsigned *__restrict__ eps_out)
{
    // ZERO: the incoming flow is identically zero (first sweep of a solve): nothing is read
    // EPS: eps_out[0] receives max |new - old| of this sweep (atomicMax on the float's bit pattern)
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const bool active = (x0 < W) && (y < H);
    float e = 0.f;
    if (active) {
        const long long base = (long long)blockIdx.z * plane;
        const long long rc = base + (long long)y * P + x0;
        const long long ru = base + (long long)clampi(y - 1, 0, H - 1) * P + x0;
        const long long rd = base + (long long)clampi(y + 1, 0, H - 1) * P + x0;
        const uint4 cw = *(const uint4 *)(coef + rc);
        float4 uc = make_float4(0.f, 0.f, 0.f, 0.f), vc = uc, uu = uc, vu = uc, ud = uc, vd = uc;
        if (!ZERO) {
            uc = *(const float4 *)(u_in + rc); vc = *(const float4 *)(v_in + rc);
            uu = *(const float4 *)(u_in + ru); vu = *(const float4 *)(v_in + ru);
            ud = *(const float4 *)(u_in + rd); vd = *(const float4 *)(v_in + rd);
        }
        // six-wide windows: [0] = pixel x0-1, [1..4] = own pixels, [5] = pixel x0+4
        float wu[6] = {uc.x, uc.x, uc.y, uc.z, uc.w, uc.w};
        float wv[6] = {vc.x, vc.x, vc.y, vc.z, vc.w, vc.w};
        if (!ZERO && x0 > 0) { wu[0] = u_in[rc - 1]; wv[0] = v_in[rc - 1]; }
        if (!ZERO && x0 + 4 < W) { wu[5] = u_in[rc + 4]; wv[5] = v_in[rc + 4]; }
        const float au[4] = {uu.x, uu.y, uu.z, uu.w}, av[4] = {vu.x, vu.y, vu.z, vu.w};
        const float bu[4] = {ud.x, ud.y, ud.z, ud.w}, bv[4] = {vd.x, vd.y, vd.z, vd.w};
        const uint32_t cc[4] = {cw.x, cw.y, cw.z, cw.w};
        float nu[4], nv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float al, be, ga;
            sweep_coefs(cc[k], ilambda, al, be, ga);
            // replicate border: the last image column is its own right neighbour
            const bool last = (x0 + k >= W - 1);
            const float uR = last ? wu[k + 1] : wu[k + 2], vR = last ? wv[k + 1] : wv[k + 2];
            if (k & 1) update_cv<1>(wu[k], uR, au[k], bu[k], wv[k], vR, av[k], bv[k], al, be, ga, nu[k], nv[k]);
            else update_cv<0>(wu[k], uR, au[k], bu[k], wv[k], vR, av[k], bv[k], al, be, ga, nu[k], nv[k]);
            if (EPS && x0 + k < W)
                e = fmaxf(e, fmaxf(fabsf(wu[k + 1] - nu[k]), fabsf(wv[k + 1] - nv[k])));
        }
        *(float4 *)(u_out + rc) = make_float4(nu[0], nu[1], nu[2], nu[3]);
        *(float4 *)(v_out + rc) = make_float4(nv[0], nv[1], nv[2], nv[3]);
    }
    if (EPS) {
        e = wave_max(e);
        if (threadIdx.x == 0) atomicMax(eps_out, __float_as_uint(e));
    }
}

// ------------------------------------------------------------------------------------------
// a2, form 2: T Jacobi sweeps per launch on an LDS tile with a T-pixel halo ("fused").
// ------------------------------------------------------------------------------------------
struct FusedGeom {
    int W, H, P;
    long long plane;        // elements between consecutive pairs
    int CW, CH;             // core (output) tile, CW % 4 == 0
    int T;                  // sweeps per launch
    int HX;                 // horizontal halo in pixels, multiple of 4, >= T
    int RW4, RH;            // region = core + halo: RW4 groups of 4 pixels wide, RH rows high
    int RS;                 // LDS row stride in floats = 4*RW4 + 8 (one guard group each side)
    int G;                  // RW4 * RH
    int tiles_x, tiles_y;
    int zero_in;            // incoming flow is identically zero: do not read u_in / v_in
};

enum : unsigned { F_ACTIVE = 1u, F_CORE = 2u, F_GU = 4u, F_GD = 8u, F_GL = 16u, F_GR = 32u };
// bits [9:8] of the flag word: position pr of image column W-1 inside the group (valid with F_GR)

// GFX9 DPP whole-wavefront shifts by one lane (no LDS traffic): lane l receives lane l-1 / l+1.
// bound_ctrl with a zero `old` lets the compiler fold the shift into the consuming VALU instruction.
__device__ __forceinline__ float wave_from_prev_lane(float x) // lane 0 receives 0
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_from_next_lane(float x) // lane 63 receives 0
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130, 0xF, 0xF, true));
}

// Work decomposition: the region (core tile + halo) is cut into groups of 4 consecutive pixels;
// lane `tid` owns groups tid, tid+NT, ... (K of them), so consecutive lanes own horizontally
// adjacent groups.  A group's own u,v (4+4 floats) and its 16 coefficients stay in VGPRs for the
// whole launch; LDS carries the rows for the up/down neighbours; the left/right neighbours come
// from the adjacent lanes by DPP (LRMODE 1) or from LDS (LRMODE 0).
// Validity: a pixel at distance d from the region edge is exact through sweep d; the core lies
// >= T pixels inside, so it is exact after T sweeps.  Image borders are exact (replicate) because
// edge pixels are their own neighbours (selects for left/right, ghost rows in LDS for up/down).
template <int NT, int K, bool EPS, int LRMODE>
__global__ __launch_bounds__(NT) void k_jacobi_fused(const uint32_t *__restrict__ coef,
                                                     const float *__restrict__ u_in,
                                                     const float *__restrict__ v_in,
                                                     float *__restrict__ u_out,
                                                     float *__restrict__ v_out, const FusedGeom g,
                                                     const float ilambda,
                                                     unsigned *__restrict__ eps_out, const int eps_stride)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *su = smem;
    float *sv = smem + g.RS * (g.RH + 2);

    const int tid = threadIdx.x;
    const int tpp = g.tiles_x * g.tiles_y;
    const int tile = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    const int pair = tile / tpp;
    const int t2 = tile - pair * tpp;
    const int by = t2 / g.tiles_x, bx = t2 - by * g.tiles_x;
    const int rx0 = bx * g.CW - g.HX, ry0 = by * g.CH - g.T;
    const long long base = (long long)pair * g.plane;

    int o[K];            // LDS float offset of the group's first pixel
    int go[K];           // global element offset (within the pair) of the group's first pixel
    unsigned fl[K];      // F_* flags
    float4 cu[K], cv[K]; // the group's own flow, current sweep
    float cAl[K][4], cBe[K][4], cGa[K][4];

#pragma unroll
    for (int k = 0; k < K; k++) {
        const int gi = tid + k * NT;
        const bool valid = gi < g.G;
        const int j = gi / g.RW4, i4 = gi - j * g.RW4;
        const int x0 = rx0 + 4 * i4, y = ry0 + j;
        o[k] = valid ? (j + 1) * g.RS + 4 + 4 * i4 : g.RS + 4;
        go[k] = y * g.P + x0;
        fl[k] = 0;
        cu[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        cv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) { // fill registers + LDS, replicate-clamped at the image border
            const int yc = clampi(y, 0, g.H - 1);
            const long long row = base + (long long)yc * g.P;
            if (g.zero_in) {
                // cu, cv stay zero
            } else if (x0 >= 0 && x0 + 3 < g.W) {
                cu[k] = *(const float4 *)(u_in + row + x0);
                cv[k] = *(const float4 *)(v_in + row + x0);
            } else {
                const int xa = clampi(x0, 0, g.W - 1), xb = clampi(x0 + 1, 0, g.W - 1),
                          xc = clampi(x0 + 2, 0, g.W - 1), xd = clampi(x0 + 3, 0, g.W - 1);
                cu[k] = make_float4(u_in[row + xa], u_in[row + xb], u_in[row + xc], u_in[row + xd]);
                cv[k] = make_float4(v_in[row + xa], v_in[row + xb], v_in[row + xc], v_in[row + xd]);
            }
            *(float4 *)(su + o[k]) = cu[k];
            *(float4 *)(sv + o[k]) = cv[k];
            if (y >= 0 && y < g.H && x0 >= 0 && x0 < g.W) {
                unsigned f = F_ACTIVE;
                const int ic = 4 * i4 - g.HX, jc = j - g.T;
                if (ic >= 0 && ic < g.CW && jc >= 0 && jc < g.CH) f |= F_CORE;
                if (y == 0) f |= F_GU;
                if (y == g.H - 1) f |= F_GD;
                if (x0 == 0) f |= F_GL;
                const int pr = g.W - 1 - x0;
                if (pr <= 3) f |= F_GR | ((unsigned)pr << 8);
                fl[k] = f;
                const uint4 cw = *(const uint4 *)(coef + base + go[k]);
                const uint32_t cc[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    sweep_coefs(cc[p], ilambda, cAl[k][p], cBe[k][p], cGa[k][p]);
                }
            }
        }
    }
    __syncthreads();

    const int lane = tid & 63;
    for (int s = 0; s < g.T; s++) {
        float e = 0.f;
#pragma unroll
        for (int k = 0; k < K; k++) {
            float uL, uR, vL, vR;
            if (LRMODE == 1) { // all 64 lanes take part (convergent), results used by active lanes
                uL = wave_from_prev_lane(cu[k].w); vL = wave_from_prev_lane(cv[k].w);
                uR = wave_from_next_lane(cu[k].x); vR = wave_from_next_lane(cv[k].x);
            }
            if (fl[k] & F_ACTIVE) {
                const float4 uu = *(const float4 *)(su + o[k] - g.RS), ud = *(const float4 *)(su + o[k] + g.RS);
                const float4 vu = *(const float4 *)(sv + o[k] - g.RS), vd = *(const float4 *)(sv + o[k] + g.RS);
                if (LRMODE == 0) {
                    uL = su[o[k] - 1]; uR = su[o[k] + 4];
                    vL = sv[o[k] - 1]; vR = sv[o[k] + 4];
                } else { // the neighbour group lives in another wavefront only at lanes 0 / 63
                    if (lane == 0) { uL = su[o[k] - 1]; vL = sv[o[k] - 1]; }
                    if (lane == 63) { uR = su[o[k] + 4]; vR = sv[o[k] + 4]; }
                }
                const unsigned f = fl[k];
                const int pr = (f & F_GR) ? (int)((f >> 8) & 3u) : 7; // image column W-1 inside the group
                if (f & F_GL) { uL = cu[k].x; vL = cv[k].x; }        // replicate: column 0 is its own left
                if (pr == 3) { uR = cu[k].w; vR = cv[k].w; }          // replicate: column W-1 is its own right
                float nu[4], nv[4];
                update_cv<0>(uL, cu[k].y, uu.x, ud.x, vL, cv[k].y, vu.x, vd.x, cAl[k][0], cBe[k][0], cGa[k][0], nu[0], nv[0]);
                update_cv<1>(cu[k].x, cu[k].z, uu.y, ud.y, cv[k].x, cv[k].z, vu.y, vd.y, cAl[k][1], cBe[k][1], cGa[k][1], nu[1], nv[1]);
                update_cv<0>(cu[k].y, cu[k].w, uu.z, ud.z, cv[k].y, cv[k].w, vu.z, vd.z, cAl[k][2], cBe[k][2], cGa[k][2], nu[2], nv[2]);
                update_cv<1>(cu[k].z, uR, uu.w, ud.w, cv[k].z, vR, vu.w, vd.w, cAl[k][3], cBe[k][3], cGa[k][3], nu[3], nv[3]);
                if (EPS && (f & F_CORE)) { // columns > pr lie outside the image
                    e = fmaxf(e, fmaxf(fabsf(cu[k].x - nu[0]), fabsf(cv[k].x - nv[0])));
                    if (pr >= 1) e = fmaxf(e, fmaxf(fabsf(cu[k].y - nu[1]), fabsf(cv[k].y - nv[1])));
                    if (pr >= 2) e = fmaxf(e, fmaxf(fabsf(cu[k].z - nu[2]), fabsf(cv[k].z - nv[2])));
                    if (pr >= 3) e = fmaxf(e, fmaxf(fabsf(cu[k].w - nu[3]), fabsf(cv[k].w - nv[3])));
                }
                // columns right of W-1 inside the group mirror column W-1 (its right neighbour)
                if (pr == 0) { nu[1] = nu[0]; nv[1] = nv[0]; }
                if (pr <= 1) { nu[2] = nu[1]; nv[2] = nv[1]; }
                if (pr <= 2) { nu[3] = nu[2]; nv[3] = nv[2]; }
                cu[k] = make_float4(nu[0], nu[1], nu[2], nu[3]);
                cv[k] = make_float4(nv[0], nv[1], nv[2], nv[3]);
            }
        }
        if (EPS) {
            e = wave_max(e);
            if (lane == 0) atomicMax(eps_out + (size_t)s * eps_stride, __float_as_uint(e));
        }
        if (s == g.T - 1) break;
        __syncthreads(); // every LDS read of sweep s is done
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (fl[k] & F_ACTIVE) {
                *(float4 *)(su + o[k]) = cu[k];
                *(float4 *)(sv + o[k]) = cv[k];
                // ghost rows: image rows 0 and H-1 are their own up / down neighbours
                if (fl[k] & F_GU) { *(float4 *)(su + o[k] - g.RS) = cu[k]; *(float4 *)(sv + o[k] - g.RS) = cv[k]; }
                if (fl[k] & F_GD) { *(float4 *)(su + o[k] + g.RS) = cu[k]; *(float4 *)(sv + o[k] + g.RS) = cv[k]; }
            }
        }
        __syncthreads(); // sweep s+1 may read
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        if ((fl[k] & (F_ACTIVE | F_CORE)) == (F_ACTIVE | F_CORE)) {
            *(float4 *)(u_out + base + go[k]) = cu[k];
            *(float4 *)(v_out + base + go[k]) = cv[k];
        }
    }
}

// ------------------------------------------------------------------------------------------
// a2, form 3: T Jacobi sweeps per launch on register-resident strips ("strip").
//
// A workgroup of NW wavefronts owns a region of 256 columns x (NW*R) rows.  Wavefront w holds rows
// [w*R, (w+1)*R) entirely in VGPRs: lane l owns the 4 pixels of columns 4l..4l+3 in each of its R
// rows (u, v and the four coefficients).  Per sweep:
//   * left/right neighbours come from lanes l-1 / l+1 by DPP wave shifts (the wavefront spans the
//     whole region width, so there is no seam: lanes 0 and 63 sit on the region edge);
//   * up/down neighbours inside the strip are the lane's own registers;
//   * only the strip's first and last row go through LDS (double-buffered, ONE barrier per sweep)
//     to reach the wavefronts above and below.
// LDS traffic per sweep is 2 of R rows instead of all of them and nothing is re-read, so the sweep
// is bound by VALU issue (~13 flops per pixel) rather than by LDS or barriers.
//
// Image borders cost nothing inside the sweep loop: Jacobi with a replicate border is exactly
// Jacobi on the EVEN REFLECTION of the image (u(-1-k) = u(k), same for the coefficients): the
// mirrored pixel sees the mirrored neighbour set, so the extension stays a reflection sweep after
// sweep, bit for bit, and pixel 0's left neighbour u(-1) equals u(0) -- the replicate rule.  So the
// halo outside the image is simply LOADED from mirrored coordinates and then swept like any other
// pixel; no select, no ghost copy.
// ------------------------------------------------------------------------------------------
struct StripGeom {
    int W, H, P;
    long long plane;
    int T, HX;          // sweeps per launch; horizontal halo (multiple of 4, >= T)
    int CW, CH;         // core = (256 - 2*HX) x (NW*R - 2*T)
    int NW;             // wavefronts per workgroup
    int tiles_x, tiles_y;
    int zero_in;        // incoming flow is identically zero: do not read u_in / v_in
};

// index of the even reflection: ..., 1, 0 | 0, 1, ..., n-1 | n-1, n-2, ...
__device__ __forceinline__ int mirror_index(int i, int n)
{
    if (i < 0) i = -1 - i;            // one bounce covers every image at least as large as the halo
    if (i >= n) i = 2 * n - 1 - i;
    if ((unsigned)i >= (unsigned)n) { // tiny image: general even-periodic extension
        const int p = 2 * n;
        int m = i % p;
        if (m < 0) m += p;
        i = m < n ? m : p - 1 - m;
    }
    return i;
}

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 f2_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 f2_swap(f2 a) { return __builtin_shufflevector(a, a, 1, 0); }

// One row of one lane: pixels (p0,p1) = P and (p2,p3) = Q as two register pairs, so that every
// arithmetic step except the four side-neighbour additions is a packed (2 pixels per
// instruction) v_pk_add/mul/fma_f32.  Summation order = the canonical one of update_cv<>:
//   p0: ((p1 + (U+D)) + left)   p1: ((p0 + (U+D)) + p2)   p2: ((p3 + (U+D)) + p1)   p3: ((p2 + (U+D)) + right)
struct RowCoef { f2 alP, alQ, beP, beQ, gaP, gaQ; };

__device__ __forceinline__ void strip_row_update(f2 &uP, f2 &uQ, f2 &vP, f2 &vQ, const f2 upuP, const f2 upuQ,
                                                 const f2 upvP, const f2 upvQ, const f2 dnuP, const f2 dnuQ,
                                                 const f2 dnvP, const f2 dnvQ, const RowCoef &c)
{
    // u plane
    f2 tP = f2_swap(uP) + (upuP + dnuP);
    f2 tQ = f2_swap(uQ) + (upuQ + dnuQ);
    tP.x += wave_from_prev_lane(uQ.y);
    tP.y += uQ.x;
    tQ.x += uP.y;
    tQ.y += wave_from_next_lane(uP.x);
    const f2 ubP = tP * 0.25f, ubQ = tQ * 0.25f;
    // v plane
    f2 sP = f2_swap(vP) + (upvP + dnvP);
    f2 sQ = f2_swap(vQ) + (upvQ + dnvQ);
    sP.x += wave_from_prev_lane(vQ.y);
    sP.y += vQ.x;
    sQ.x += vP.y;
    sQ.y += wave_from_next_lane(vP.x);
    const f2 vbP = sP * 0.25f, vbQ = sQ * 0.25f;
    // update
    const f2 qP = f2_fma(c.alP, ubP, f2_fma(c.beP, vbP, c.gaP));
    const f2 qQ = f2_fma(c.alQ, ubQ, f2_fma(c.beQ, vbQ, c.gaQ));
    uP = f2_fma(-c.alP, qP, ubP);
    vP = f2_fma(-c.beP, qP, vbP);
    uQ = f2_fma(-c.alQ, qQ, ubQ);
    vQ = f2_fma(-c.beQ, qQ, vbQ);
}

template <int R, int NTMAX, int EPS>
__global__ __launch_bounds__(NTMAX) void k_jacobi_strip(const uint32_t *__restrict__ coef,
                                                        const float *__restrict__ u_in,
                                                        const float *__restrict__ v_in,
                                                        float *__restrict__ u_out,
                                                        float *__restrict__ v_out, const StripGeom g,
                                                        const float ilambda,
                                                        unsigned *__restrict__ eps_out, const int eps_stride,
                                                        unsigned long long *__restrict__ stamps,
                                                        const float eps_thr)
{
    // EPS == 1: eps_out[sweep * eps_stride + workgroup] receives that workgroup's max |new - old| over
    // its core pixels (plain stores, no atomics; the host reduces over the workgroups afterwards).
    // EPS == 2 ("witness"): the cheap way to PROVE that no sweep of this launch had Eps < epsilon.
    // At the end of a sweep a wavefront whose first row is a core row asks "did u change by >= eps_thr
    // at column x0 of any lane of that row?" -- old and new value come back from the two exchange
    // buffers, so nothing is kept in registers for it -- and counts the
    // per-sweep answers in an SGPR.  Each |change| is a lower bound of that sweep's Eps, so a
    // wavefront that answered yes in EVERY sweep proves Eps_k >= eps_thr for all k of the launch;
    // eps_out[workgroup] = +inf if any wavefront of the workgroup did, else 0.
    // `stamps` is a diagnostic buffer (NULL in production: no stamp executes).  When set, lane 0 of
    // wavefront 0 records shader-clock / 100 MHz wall-clock stamps at the phase boundaries into
    // memory nothing else reads (HSFLOW_DEBUG_STAMPS, see hs_runtime.hip.h).
    extern __shared__ __attribute__((aligned(16))) float4 ex[]; // [2][NW][4][64], then 32 floats for Eps
    unsigned long long st0 = 0, sr0 = 0, st1 = 0, st2 = 0;
    if (stamps) { st0 = __builtin_amdgcn_s_memtime(); sr0 = __builtin_amdgcn_s_memrealtime(); }
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NW = g.NW;
    float *eps_lds = (float *)(ex + (size_t)2 * NW * 4 * 64);
    // Synchronisation between sweeps: ONE workgroup barrier per sweep.  Experiment kept behind
    // HS_STRIP_P2P (default 0, measured SLOWER: 0.220 vs 0.200 ms at 1080p / 100 sweeps): a wavefront only
    // depends on the strips directly above and below, so each wavefront raises a counter in LDS once its
    // edge rows of a sweep are published and waits only for its two neighbours' counters.  LDS executes a
    // wavefront's requests in order, so "counter >= s" implies that neighbour's rows for sweep s are in
    // place AND that its reads of the buffer about to be overwritten are done (they precede its publish).
#ifndef HS_STRIP_P2P
#define HS_STRIP_P2P 0
#endif
    constexpr bool P2P = HS_STRIP_P2P && EPS != 1; // the per-sweep Eps fold of EPS == 1 relies on the barrier
    volatile unsigned *flags = (volatile unsigned *)(eps_lds + 32); // [NW] sweeps published so far
    const int tpp = g.tiles_x * g.tiles_y;
    const int tile = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    const int pair = tile / tpp;
    const int t2 = tile - pair * tpp;
    const int by = t2 / g.tiles_x, bx = t2 - by * g.tiles_x;
    const int x0 = bx * g.CW - g.HX + 4 * lane;
    const int y0 = by * g.CH - g.T + w * R;
    const long long base = (long long)pair * g.plane;
    const bool xin = (x0 >= 0) && (x0 + 3 < g.W); // the whole group lies inside the image

    f2 uP[R], uQ[R], vP[R], vQ[R];
    RowCoef cf[R];
    float4 lu[R], lv[R];
    uint4 lc[R];
    // Workgroup-uniform: does the region (core + halo) stick out of the image on the left or right?
    // Tiles that do not (the vast majority) load with plain aligned 16-byte accesses only.
    const int rx0 = bx * g.CW - g.HX;
    const bool xedge = !(rx0 >= 0 && rx0 + 256 <= g.W);
    if (!xedge) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const long long off = base + (long long)mirror_index(y0 + r, g.H) * g.P + x0;
            lu[r] = lv[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!g.zero_in) {
                lu[r] = *(const float4 *)(u_in + off);
                lv[r] = *(const float4 *)(v_in + off);
            }
            lc[r] = *(const uint4 *)(coef + off);
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const long long row = base + (long long)mirror_index(y0 + r, g.H) * g.P;
            // A group that lies completely outside the image on the left mirrors onto an aligned
            // group read backwards (columns -1-k <-> k); the same holds on the right when W % 4 == 0.
            // Those lanes keep the 16-byte loads (from the mirrored address, components reversed).
            // Only groups that straddle column W-1 or sit right of it when W % 4 != 0 (and images
            // narrower than the halo) fall back to four reflected scalar loads per plane.
            int xg = x0;
            bool rev = false, slow = false;
            if (!xin) {
                if (x0 < 0 && -x0 <= g.W) { xg = -x0 - 4; rev = true; }
                else if (x0 >= g.W && (g.W & 3) == 0 && 2 * g.W - x0 - 4 >= 0) { xg = 2 * g.W - x0 - 4; rev = true; }
                else { xg = 0; slow = true; }
            }
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (!g.zero_in) {
                a = *(const float4 *)(u_in + row + xg);
                b = *(const float4 *)(v_in + row + xg);
            }
            uint4 c = *(const uint4 *)(coef + row + xg);
            if (rev) {
                a = make_float4(a.w, a.z, a.y, a.x);
                b = make_float4(b.w, b.z, b.y, b.x);
                c = make_uint4(c.w, c.z, c.y, c.x);
            }
            if (slow) { // volatile keeps this a separate, rarely taken path
                const volatile float *uv = u_in + row, *vv = v_in + row;
                const volatile uint32_t *cv = coef + row;
                const int xa = mirror_index(x0, g.W), xb = mirror_index(x0 + 1, g.W),
                          xc = mirror_index(x0 + 2, g.W), xd = mirror_index(x0 + 3, g.W);
                if (!g.zero_in) {
                    a = make_float4(uv[xa], uv[xb], uv[xc], uv[xd]);
                    b = make_float4(vv[xa], vv[xb], vv[xc], vv[xd]);
                }
                c = make_uint4(cv[xa], cv[xb], cv[xc], cv[xd]);
            }
            lu[r] = a; lv[r] = b; lc[r] = c;
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const float4 lu_ = lu[r], lv_ = lv[r];
        const uint4 cw = lc[r];
        uP[r] = f2{lu_.x, lu_.y}; uQ[r] = f2{lu_.z, lu_.w};
        vP[r] = f2{lv_.x, lv_.y}; vQ[r] = f2{lv_.z, lv_.w};
        float al[4], be[4], ga[4];
        const uint32_t cc[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
        for (int p = 0; p < 4; p++) sweep_coefs(cc[p], ilambda, al[p], be[p], ga[p]);
        cf[r].alP = f2{al[0], al[1]}; cf[r].alQ = f2{al[2], al[3]};
        cf[r].beP = f2{be[0], be[1]}; cf[r].beQ = f2{be[2], be[3]};
        cf[r].gaP = f2{ga[0], ga[1]}; cf[r].gaQ = f2{ga[2], ga[3]};
    }
    // core membership (for the store and for Eps): rows as a bit mask, lanes as a flag
    unsigned rowcore = 0;
    int rdist[R]; // distance of each row from the core rows (0 inside): wave-uniform
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int j = w * R + r, y = y0 + r;
        if (j >= g.T && j < g.T + g.CH && y >= 0 && y < g.H) rowcore |= 1u << r;
        rdist[r] = j < g.T ? g.T - j : (j >= g.T + g.CH ? j - (g.T + g.CH - 1) : 0);
    }
    const bool lanecore = (x0 >= 0) && (x0 < g.W) && (4 * lane >= g.HX) && (4 * lane < g.HX + g.CW);
    const int pr = g.W - 1 - x0; // image columns of this group: 0..min(pr,3)

    // One row update.  up*/dn* are OLD neighbour rows; the new row replaces uP[r].. in place.
    // A row at distance d from the core is only needed through sweep T-1-d (trapezoid): later
    // sweeps skip it (wave-uniform branch), which trims the redundant halo work by about half.
#ifdef HS_DIAG_NO_COMPUTE /* diagnostic build only: wrong results, times the exchange alone */
#define HS_ROW(r, UUP, UUQ, UVP, UVQ, DUP, DUQ, DVP, DVQ, CF) do { uP[r] += UUP + DUP; } while (0)
#else
#define HS_ROW(r, UUP, UUQ, UVP, UVQ, DUP, DUQ, DVP, DVQ, CF)                                      \
    do {                                                                                           \
        if (rdist[r] <= g.T - 1 - s) {                                                             \
            const f2 ouP = uP[r], ouQ = uQ[r], ovP = vP[r], ovQ = vQ[r];                           \
            strip_row_update(uP[r], uQ[r], vP[r], vQ[r], UUP, UUQ, UVP, UVQ, DUP, DUQ, DVP, DVQ, CF);    \
            if (EPS) {                                                                             \
                if ((rowcore >> (r)) & 1u) { /* wave-uniform; lanes outside the core are masked once per sweep */ \
                    if (EPS == 1) {                                                                \
                        const f2 dUP = ouP - uP[r], dUQ = ouQ - uQ[r], dVP = ovP - vP[r], dVQ = ovQ - vQ[r]; \
                        if (!xedge) { /* workgroup-uniform: every column of the region is an image column */ \
                            e = fmaxf(fmaxf(e, fabsf(dUP.x)), fabsf(dUP.y));                       \
                            e = fmaxf(fmaxf(e, fabsf(dUQ.x)), fabsf(dUQ.y));                       \
                            e = fmaxf(fmaxf(e, fabsf(dVP.x)), fabsf(dVP.y));                       \
                            e = fmaxf(fmaxf(e, fabsf(dVQ.x)), fabsf(dVQ.y));                       \
                        } else {                                                                   \
                            e = fmaxf(e, fmaxf(fabsf(dUP.x), fabsf(dVP.x)));                       \
                            if (pr >= 1) e = fmaxf(e, fmaxf(fabsf(dUP.y), fabsf(dVP.y)));          \
                            if (pr >= 2) e = fmaxf(e, fmaxf(fabsf(dUQ.x), fabsf(dVQ.x)));          \
                            if (pr >= 3) e = fmaxf(e, fmaxf(fabsf(dUQ.y), fabsf(dVQ.y)));          \
                        }                                                                          \
                    }                                                                              \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
        /* (no scheduling barrier between rows: letting the scheduler overlap them is 1.7 % faster    \
           and, with this compiler, also spills less in the Eps variants) */                      \
    } while (0)
#endif
#if defined(HS_DIAG_NO_EXCHANGE) || defined(HS_DIAG_NO_LDS) /* diagnostic builds only: wrong results */
#define HS_PUBLISH(buf) do { } while (0)
#else
#define HS_PUBLISH(buf)                                                                            \
    do {                                                                                           \
        float4 *exw = ex + ((size_t)((buf) * NW + w) * 4) * 64 + lane;                             \
        exw[0] = make_float4(uP[0].x, uP[0].y, uQ[0].x, uQ[0].y);                                  \
        exw[64] = make_float4(vP[0].x, vP[0].y, vQ[0].x, vQ[0].y);                                 \
        exw[128] = make_float4(uP[R - 1].x, uP[R - 1].y, uQ[R - 1].x, uQ[R - 1].y);                \
        exw[192] = make_float4(vP[R - 1].x, vP[R - 1].y, vQ[R - 1].x, vQ[R - 1].y);                \
    } while (0)
#endif

#define HS_RAISE(n)                                                                                \
    do {                                                                                           \
        if (P2P) {                                                                                 \
            asm volatile("" ::: "memory"); /* the rows first (LDS keeps a wavefront's order) */    \
            if (lane == 0) flags[w] = (unsigned)(n);                                               \
        }                                                                                          \
    } while (0)

    // Exchange slots: ex[buf][wave][0..3][lane] = {first row u, first row v, last row u, last row v}.
    // Sweep s reads buffer s&1 and publishes its new edge rows into buffer (s+1)&1, then meets the
    // other wavefronts at ONE barrier.  The edge rows are updated and published FIRST so that the
    // LDS writes drain while the interior rows are being computed.
    HS_PUBLISH(0);
    if (P2P && threadIdx.x < NW) flags[threadIdx.x] = 0;
    __syncthreads();
    if (stamps) st1 = __builtin_amdgcn_s_memtime();
    const int wu = w > 0 ? w - 1 : 0, su = w > 0 ? 2 : 0;          // strip above: its last row
    const int wd = w < NW - 1 ? w + 1 : w, sd = w < NW - 1 ? 0 : 2; // strip below: its first row
    // (at the region edge the strip's own edge row stands in: junk the validity argument tolerates)
    int seen_n = 0; // EPS == 2, wave-uniform: sweeps so far that had a change >= eps_thr (a counter: a
                    // loop-carried flag makes the register allocator spill inside the loop)
#pragma unroll 1
    for (int s = 0; s < g.T; s++) {
#if defined(HS_DIAG_NO_EXCHANGE) || defined(HS_DIAG_NO_LDS)
        const float4 hu4 = make_float4(uP[0].x, uP[0].y, uQ[0].x, uQ[0].y), hv4 = hu4, du4 = hu4, dv4 = hu4;
#else
        const float4 *eu = ex + ((size_t)((s & 1) * NW + wu) * 4 + su) * 64 + lane;
        const float4 *ed = ex + ((size_t)((s & 1) * NW + wd) * 4 + sd) * 64 + lane;
        float4 hu4, hv4, du4, dv4;
        if (P2P) {
            // counters first, rows right behind them in the same batch of LDS reads: if the counters
            // (read earlier) say "published", the rows (read later) are the published ones; otherwise
            // the batch is simply repeated
            for (;;) {
                asm volatile("" ::: "memory");
                const unsigned fa = flags[wu], fb = flags[wd];
                hu4 = eu[0]; hv4 = eu[64];   // old row above the strip
                du4 = ed[0]; dv4 = ed[64];   // old row below the strip
                if (fa >= (unsigned)s && fb >= (unsigned)s) break;
                __builtin_amdgcn_s_sleep(1);
            }
        } else {
            hu4 = eu[0]; hv4 = eu[64];
            du4 = ed[0]; dv4 = ed[64];
        }
#endif
        const f2 huP = f2{hu4.x, hu4.y}, huQ = f2{hu4.z, hu4.w}, hvP = f2{hv4.x, hv4.y}, hvQ = f2{hv4.z, hv4.w};
        const f2 duP = f2{du4.x, du4.y}, duQ = f2{du4.z, du4.w}, dvP = f2{dv4.x, dv4.y}, dvQ = f2{dv4.z, dv4.w};
        float e = 0.f;
        if (R == 1) {
            HS_ROW(0, huP, huQ, hvP, hvQ, duP, duQ, dvP, dvQ, cf[0]);
        } else {
            constexpr int R1 = R > 1 ? 1 : 0, RM = R > 2 ? R - 2 : 0;
            const f2 o0uP = uP[0], o0uQ = uQ[0], o0vP = vP[0], o0vQ = vQ[0];                 // old first row
            const f2 oNuP = uP[R - 1], oNuQ = uQ[R - 1], oNvP = vP[R - 1], oNvQ = vQ[R - 1]; // old last row
            HS_ROW(0, huP, huQ, hvP, hvQ, uP[R1], uQ[R1], vP[R1], vQ[R1], cf[0]);
            if (R == 2) HS_ROW(R - 1, o0uP, o0uQ, o0vP, o0vQ, duP, duQ, dvP, dvQ, cf[R - 1]);
            else HS_ROW(R - 1, uP[RM], uQ[RM], vP[RM], vQ[RM], duP, duQ, dvP, dvQ, cf[R - 1]);
            if (EPS == 2 || s + 1 < g.T) { HS_PUBLISH((s + 1) & 1); HS_RAISE(s + 1); }
            f2 puP = o0uP, puQ = o0uQ, pvP = o0vP, pvQ = o0vQ; // old row r-1 while walking the interior rows
#pragma unroll
            for (int r = 1; r < R - 1; r++) {
                const f2 kuP = uP[r], kuQ = uQ[r], kvP = vP[r], kvQ = vQ[r];
                const int rn = r + 1 < R ? r + 1 : r;
                if (r + 1 == R - 1) HS_ROW(r, puP, puQ, pvP, pvQ, oNuP, oNuQ, oNvP, oNvQ, cf[r]);
                else HS_ROW(r, puP, puQ, pvP, pvQ, uP[rn], uQ[rn], vP[rn], vQ[rn], cf[r]);
                puP = kuP; puQ = kuQ; pvP = kvP; pvQ = kvQ;
            }
        }
        if (R == 1 && (EPS == 2 || s + 1 < g.T)) { HS_PUBLISH((s + 1) & 1); HS_RAISE(s + 1); }
        if (EPS == 1) { // per-wavefront maximum -> LDS; wavefront 0 folds the previous sweep's 16 values
            e = wave_max_nonneg(lanecore ? e : 0.f);
            if (lane == 0) eps_lds[(s & 1) * 16 + w] = e;
            if (s > 0 && w == 0) {
                float x = lane < NW ? eps_lds[((s - 1) & 1) * 16 + lane] : 0.f;
                x = wave_max_nonneg(x);
                if (lane == 0) eps_out[(size_t)(s - 1) * eps_stride + blockIdx.x] = __float_as_uint(x);
            }
        }
        if (EPS == 2 && (rowcore & 1u)) {
            // witness, read back from the exchange buffers at the end of the sweep (no register is kept
            // for it): slot 0 of this wavefront holds its first row of u, new in buffer (s+1)&1 and old
            // in buffer s&1; component x is column x0, an image column wherever lanecore holds
            const float nu = *(const float *)(ex + ((size_t)(((s + 1) & 1) * NW + w) * 4) * 64 + lane);
            const float ou = *(const float *)(ex + ((size_t)((s & 1) * NW + w) * 4) * 64 + lane);
            seen_n += __builtin_amdgcn_ballot_w64(lanecore && fabsf(ou - nu) >= eps_thr) != 0 ? 1 : 0;
        }
#if !defined(HS_DIAG_NO_EXCHANGE) && !defined(HS_DIAG_NO_BARRIER)
        if (!P2P && s + 1 < g.T) __syncthreads();
#endif
    }
    if (EPS == 1) {
        __syncthreads();
        if (w == 0) {
            float x = lane < NW ? eps_lds[((g.T - 1) & 1) * 16 + lane] : 0.f;
            x = wave_max_nonneg(x);
            if (lane == 0) eps_out[(size_t)(g.T - 1) * eps_stride + blockIdx.x] = __float_as_uint(x);
        }
    }
    if (EPS == 2) {
        if (lane == 0) eps_lds[w] = (seen_n == g.T && (rowcore & 1u)) ? __builtin_inff() : 0.f;
        __syncthreads();
        if (w == 0) {
            const float y = wave_max_nonneg(lane < NW ? eps_lds[lane] : 0.f);
            if (lane == 0) eps_out[blockIdx.x] = __float_as_uint(y);
        }
    }
#undef HS_ROW
#undef HS_PUBLISH
#undef HS_RAISE
    if (stamps) st2 = __builtin_amdgcn_s_memtime();

    if (lanecore) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            if ((rowcore >> r) & 1u) {
                const long long off = base + (long long)(y0 + r) * g.P + x0;
#if defined(HS_EXP_NT_STORE) /* experiment (slower): non-temporal stores */
                typedef float v4f __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(v4f{uP[r].x, uP[r].y, uQ[r].x, uQ[r].y}, (v4f *)(u_out + off));
                __builtin_nontemporal_store(v4f{vP[r].x, vP[r].y, vQ[r].x, vQ[r].y}, (v4f *)(v_out + off));
#elif defined(HS_EXP_SC1_STORE) /* experiment: agent-scope write-through stores (nothing dirty at kernel end) */
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f su_ = v4f{uP[r].x, uP[r].y, uQ[r].x, uQ[r].y}, sv_ = v4f{vP[r].x, vP[r].y, vQ[r].x, vQ[r].y};
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(u_out + off), "v"(su_) : "memory");
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(v_out + off), "v"(sv_) : "memory");
#else
                *(float4 *)(u_out + off) = make_float4(uP[r].x, uP[r].y, uQ[r].x, uQ[r].y);
                *(float4 *)(v_out + off) = make_float4(vP[r].x, vP[r].y, vQ[r].x, vQ[r].y);
#endif
            }
        }
    }
    if (stamps && threadIdx.x == 0) {
        __builtin_amdgcn_s_waitcnt(0); // stores issued and acknowledged
        unsigned long long *o = stamps + (size_t)blockIdx.x * 8;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memtime();
        o[4] = sr0; o[5] = __builtin_amdgcn_s_memrealtime();
        o[6] = (unsigned long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20); // XCC_ID
        o[7] = (unsigned long long)tile;
    }
}

// ------------------------------------------------------------------------------------------
// a2, form 4: "folded" strips.  Same register-resident scheme as k_jacobi_strip, but one wavefront
// holds TWO vertically adjacent strips of 128 columns: lanes 0-31 the upper one (rows top->bottom
// in registers 0..R-1), lanes 32-63 the lower one in MIRRORED order (register r = block row
// 2R-1-r).  Both halves then have their wave-internal boundary at register row R-1 and their
// outer edge at register row 0, so
//   * the inner boundary is exchanged inside the wavefront (v_permlane32_swap, no LDS),
//   * each wavefront publishes ONE row per half through LDS (2 ds_write_b128 + 2 ds_read_b128 per
//     sweep instead of 4 + 4) -- the LDS edge-row exchange is what bounds the strip kernel,
//   * no per-half selects are needed: the update is symmetric in up/down, so the lower half simply
//     walks its rows in the opposite direction.
// The DPP wave shifts cross the lane 31/32 seam, which is harmless: lanes 31 and 32 sit on the
// region's right / left edge (junk the validity argument tolerates).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float lane_xor32(float x, bool lower)
{
    // value of the same register in lane (l ^ 32)
    const unsigned b = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false); // r[0] = {lo, lo}, r[1] = {hi, hi}
    return __uint_as_float(lower ? r[0] : r[1]);
}

template <int R, int NTMAX, int EPS> // EPS: 0 none, 1 Eps of every sweep, 2 witness (see k_jacobi_strip)
__global__ __launch_bounds__(NTMAX) void k_jacobi_fold(const uint32_t *__restrict__ coef,
                                                       const float *__restrict__ u_in,
                                                       const float *__restrict__ v_in,
                                                       float *__restrict__ u_out,
                                                       float *__restrict__ v_out, const StripGeom g,
                                                       const float ilambda,
                                                       unsigned *__restrict__ eps_out, const int eps_stride,
                                                       unsigned long long *__restrict__ stamps,
                                                       const float eps_thr)
{
    extern __shared__ __attribute__((aligned(16))) float4 ex[]; // [2 buf][NW][2 half][2 plane][32], then Eps
    unsigned long long st0 = 0, sr0 = 0, st1 = 0, st2 = 0;
    if (stamps) { st0 = __builtin_amdgcn_s_memtime(); sr0 = __builtin_amdgcn_s_memrealtime(); }
    const int lane = threadIdx.x & 63, hl = lane & 31;
    const bool lower = lane >= 32;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NW = g.NW;
    float *eps_lds = (float *)(ex + (size_t)2 * NW * 2 * 2 * 32);
    const int tpp = g.tiles_x * g.tiles_y;
    const int tile = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    const int pair = tile / tpp;
    const int t2 = tile - pair * tpp;
    const int by = t2 / g.tiles_x, bx = t2 - by * g.tiles_x;
    const int rx0 = bx * g.CW - g.HX;
    const int x0 = rx0 + 4 * hl;
    const int yb = by * g.CH - g.T + w * 2 * R; // first block row of this wavefront
    const long long base = (long long)pair * g.plane;
    const bool xin = (x0 >= 0) && (x0 + 3 < g.W);

    f2 uP[R], uQ[R], vP[R], vQ[R];
    RowCoef cf[R];
    float4 lu[R], lv[R];
    uint4 lc[R];
    const bool side = !(rx0 >= 0 && rx0 + 128 <= g.W); // workgroup-uniform
    int xg = x0;
    bool rev = false, slow = false;
    if (side && !xin) { // see k_jacobi_strip: mirrored aligned group, or the general scalar path
        if (x0 < 0 && -x0 <= g.W) { xg = -x0 - 4; rev = true; }
        else if (x0 >= g.W && (g.W & 3) == 0 && 2 * g.W - x0 - 4 >= 0) { xg = 2 * g.W - x0 - 4; rev = true; }
        else { xg = 0; slow = true; }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int y = yb + (lower ? 2 * R - 1 - r : r);
        const long long row = base + (long long)mirror_index(y, g.H) * g.P;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (!g.zero_in) {
            a = *(const float4 *)(u_in + row + xg);
            b = *(const float4 *)(v_in + row + xg);
        }
        uint4 c = *(const uint4 *)(coef + row + xg);
        if (side) {
            if (rev) {
                a = make_float4(a.w, a.z, a.y, a.x);
                b = make_float4(b.w, b.z, b.y, b.x);
                c = make_uint4(c.w, c.z, c.y, c.x);
            }
            if (slow) { // volatile keeps this a separate, rarely taken path
                const volatile float *uv = u_in + row, *vv = v_in + row;
                const volatile uint32_t *cv = coef + row;
                const int xa = mirror_index(x0, g.W), xb = mirror_index(x0 + 1, g.W),
                          xc = mirror_index(x0 + 2, g.W), xd = mirror_index(x0 + 3, g.W);
                if (!g.zero_in) {
                    a = make_float4(uv[xa], uv[xb], uv[xc], uv[xd]);
                    b = make_float4(vv[xa], vv[xb], vv[xc], vv[xd]);
                }
                c = make_uint4(cv[xa], cv[xb], cv[xc], cv[xd]);
            }
        }
        lu[r] = a; lv[r] = b; lc[r] = c;
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        uP[r] = f2{lu[r].x, lu[r].y}; uQ[r] = f2{lu[r].z, lu[r].w};
        vP[r] = f2{lv[r].x, lv[r].y}; vQ[r] = f2{lv[r].z, lv[r].w};
        float al[4], be[4], ga[4];
        const uint32_t cc[4] = {lc[r].x, lc[r].y, lc[r].z, lc[r].w};
#pragma unroll
        for (int p = 0; p < 4; p++) sweep_coefs(cc[p], ilambda, al[p], be[p], ga[p]);
        cf[r].alP = f2{al[0], al[1]}; cf[r].alQ = f2{al[2], al[3]};
        cf[r].beP = f2{be[0], be[1]}; cf[r].beQ = f2{be[2], be[3]};
        cf[r].gaP = f2{ga[0], ga[1]}; cf[r].gaQ = f2{ga[2], ga[3]};
    }
    // core membership: per lane (the two halves hold different rows); skip distances: per wavefront
    unsigned rowcore = 0;
    int rdist[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int jl = w * 2 * R + (lower ? 2 * R - 1 - r : r), y = yb + (lower ? 2 * R - 1 - r : r);
        if (jl >= g.T && jl < g.T + g.CH && y >= 0 && y < g.H) rowcore |= 1u << r;
        const int ju = w * 2 * R + r, jd = w * 2 * R + 2 * R - 1 - r;
        const int du = ju < g.T ? g.T - ju : (ju >= g.T + g.CH ? ju - (g.T + g.CH - 1) : 0);
        const int dd = jd < g.T ? g.T - jd : (jd >= g.T + g.CH ? jd - (g.T + g.CH - 1) : 0);
        rdist[r] = du < dd ? du : dd; // the row is computed while either half still needs it
    }
    const bool lanecore = (x0 >= 0) && (x0 < g.W) && (4 * hl >= g.HX) && (4 * hl < g.HX + g.CW);
    const int pr = g.W - 1 - x0;

#define HF_ROW(r, UUP, UUQ, UVP, UVQ, DUP, DUQ, DVP, DVQ)                                          \
    do {                                                                                           \
        if (rdist[r] <= g.T - 1 - s) {                                                             \
            const f2 ouP = uP[r], ouQ = uQ[r], ovP = vP[r], ovQ = vQ[r];                           \
            strip_row_update(uP[r], uQ[r], vP[r], vQ[r], UUP, UUQ, UVP, UVQ, DUP, DUQ, DVP, DVQ, cf[r]); \
            if (EPS == 1) {                                                                        \
                if (((rowcore >> (r)) & 1u) && lanecore) {                                         \
                    e = fmaxf(e, fmaxf(fabsf(ouP.x - uP[r].x), fabsf(ovP.x - vP[r].x)));           \
                    if (pr >= 1) e = fmaxf(e, fmaxf(fabsf(ouP.y - uP[r].y), fabsf(ovP.y - vP[r].y))); \
                    if (pr >= 2) e = fmaxf(e, fmaxf(fabsf(ouQ.x - uQ[r].x), fabsf(ovQ.x - vQ[r].x))); \
                    if (pr >= 3) e = fmaxf(e, fmaxf(fabsf(ouQ.y - uQ[r].y), fabsf(ovQ.y - vQ[r].y))); \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
    // slot of (buffer, wavefront, half): two planes of 32 float4
#define HF_SLOT(buf, ww, hh) (ex + ((size_t)(((buf) * NW + (ww)) * 2 + (hh)) * 2) * 32)
#define HF_PUBLISH(buf)                                                                            \
    do {                                                                                           \
        float4 *exw = HF_SLOT(buf, w, lower ? 1 : 0) + hl;                                         \
        exw[0] = make_float4(uP[0].x, uP[0].y, uQ[0].x, uQ[0].y);                                  \
        exw[32] = make_float4(vP[0].x, vP[0].y, vQ[0].x, vQ[0].y);                                 \
    } while (0)

    HF_PUBLISH(0);
    __syncthreads();
    if (stamps) st1 = __builtin_amdgcn_s_memtime();
    // outer neighbour: upper half <- bottom row of the wavefront above (its half 1),
    //                  lower half <- top row of the wavefront below (its half 0);
    // at the region edge the wavefront's own slot stands in (junk the validity argument tolerates)
    const int wo = lower ? (w < NW - 1 ? w + 1 : w) : (w > 0 ? w - 1 : w);
    const int ho = lower ? (w < NW - 1 ? 0 : 1) : (w > 0 ? 1 : 0);
    int seen_n = 0; // EPS == 2: sweeps in which some lane of this wavefront saw a change >= eps_thr
#pragma unroll 1
    for (int s = 0; s < g.T; s++) {
        const float4 *eo = HF_SLOT(s & 1, wo, ho) + hl;
        const float4 h4u = eo[0], h4v = eo[32];
        const f2 ouP_ = f2{h4u.x, h4u.y}, ouQ_ = f2{h4u.z, h4u.w}, ovP_ = f2{h4v.x, h4v.y}, ovQ_ = f2{h4v.z, h4v.w};
        // inner neighbour: the other half's register row R-1 (old values), in-register exchange
        const f2 iuP = f2{lane_xor32(uP[R - 1].x, lower), lane_xor32(uP[R - 1].y, lower)};
        const f2 iuQ = f2{lane_xor32(uQ[R - 1].x, lower), lane_xor32(uQ[R - 1].y, lower)};
        const f2 ivP = f2{lane_xor32(vP[R - 1].x, lower), lane_xor32(vP[R - 1].y, lower)};
        const f2 ivQ = f2{lane_xor32(vQ[R - 1].x, lower), lane_xor32(vQ[R - 1].y, lower)};
        float e = 0.f;
        // register row 0 (the published outer edge) first, so that its LDS write drains under the
        // other rows; then rows 1..R-1 walking towards the inner boundary, keeping one old row
        f2 puP = uP[0], puQ = uQ[0], pvP = vP[0], pvQ = vQ[0];
        if (R == 1) {
            HF_ROW(0, ouP_, ouQ_, ovP_, ovQ_, iuP, iuQ, ivP, ivQ);
        } else {
            constexpr int R1 = R > 1 ? 1 : 0;
            HF_ROW(0, ouP_, ouQ_, ovP_, ovQ_, uP[R1], uQ[R1], vP[R1], vQ[R1]);
        }
        if (EPS == 2 || s + 1 < g.T) HF_PUBLISH((s + 1) & 1);
#pragma unroll
        for (int r = 1; r < R; r++) {
            const f2 kuP = uP[r], kuQ = uQ[r], kvP = vP[r], kvQ = vQ[r];
            const int rn = r + 1 < R ? r + 1 : r;
            if (r == R - 1) HF_ROW(r, puP, puQ, pvP, pvQ, iuP, iuQ, ivP, ivQ);
            else HF_ROW(r, puP, puQ, pvP, pvQ, uP[rn], uQ[rn], vP[rn], vQ[rn]);
            puP = kuP; puQ = kuQ; pvP = kvP; pvQ = kvQ;
        }
        if (EPS == 2) {
            // witness (k_jacobi_strip explains it): old and new value of the published row -- register
            // row 0 of each half -- at column x0 come back from the two exchange buffers
            const float nu = *(const float *)(HF_SLOT((s + 1) & 1, w, lower ? 1 : 0) + hl);
            const float ou = *(const float *)(HF_SLOT(s & 1, w, lower ? 1 : 0) + hl);
            seen_n += __builtin_amdgcn_ballot_w64((rowcore & 1u) && lanecore && fabsf(ou - nu) >= eps_thr) != 0 ? 1 : 0;
        }
        if (EPS == 1) {
            e = wave_max(e);
            if (lane == 0) eps_lds[(s & 1) * 16 + w] = e;
            if (s > 0 && w == 0) {
                float x = lane < NW ? eps_lds[((s - 1) & 1) * 16 + lane] : 0.f;
                x = wave_max(x);
                if (lane == 0) eps_out[(size_t)(s - 1) * eps_stride + blockIdx.x] = __float_as_uint(x);
            }
        }
        if (s + 1 < g.T) __syncthreads();
    }
    if (EPS == 1) {
        __syncthreads();
        if (w == 0) {
            float x = lane < NW ? eps_lds[((g.T - 1) & 1) * 16 + lane] : 0.f;
            x = wave_max(x);
            if (lane == 0) eps_out[(size_t)(g.T - 1) * eps_stride + blockIdx.x] = __float_as_uint(x);
        }
    }
    if (EPS == 2) {
        if (lane == 0) eps_lds[w] = seen_n == g.T ? __builtin_inff() : 0.f;
        __syncthreads();
        if (w == 0) {
            const float y = wave_max_nonneg(lane < NW ? eps_lds[lane] : 0.f);
            if (lane == 0) eps_out[blockIdx.x] = __float_as_uint(y);
        }
    }
#undef HF_ROW
#undef HF_PUBLISH
#undef HF_SLOT
    if (stamps) st2 = __builtin_amdgcn_s_memtime();

    if (lanecore) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            if ((rowcore >> r) & 1u) {
                const int y = yb + (lower ? 2 * R - 1 - r : r);
                const long long off = base + (long long)y * g.P + x0;
                *(float4 *)(u_out + off) = make_float4(uP[r].x, uP[r].y, uQ[r].x, uQ[r].y);
                *(float4 *)(v_out + off) = make_float4(vP[r].x, vP[r].y, vQ[r].x, vQ[r].y);
            }
        }
    }
    if (stamps && threadIdx.x == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        unsigned long long *o = stamps + (size_t)blockIdx.x * 8;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memtime();
        o[4] = sr0; o[5] = __builtin_amdgcn_s_memrealtime();
        o[6] = (unsigned long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);
        o[7] = (unsigned long long)tile;
    }
}

// Eps of every sweep = maximum over that sweep's row of per-workgroup values (bit patterns of
// non-negative floats order like unsigned integers).  One workgroup per sweep.
__global__ __launch_bounds__(256) void k_eps_reduce(const unsigned *__restrict__ tiles, int stride,
                                                    unsigned *__restrict__ out)
{
    __shared__ unsigned part[4];
    const unsigned *row = tiles + (size_t)blockIdx.x * stride;
    unsigned m = 0;
    for (int i = threadIdx.x; i < stride; i += 256) m = max(m, row[i]);
    float f = wave_max(__uint_as_float(m));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = __float_as_uint(f);
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = max(max(part[0], part[1]), max(part[2], part[3]));
}

} // namespace hsk
