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
// every kernel variant produces bit-identical flow (tests rely on that; the strip kernels' scaled state,
// hs_kernels_strip.hip.h, keeps it down to the edge of the denormal range).
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

#ifndef HS_EXACT_SQRT_ALPHA
#define HS_EXACT_SQRT_ALPHA 0
#endif

// Per-pixel coefficients of the sweep, derived once per launch from the packed derivatives:
//   al = Ix*s, be = Iy*s, ga = It*s   with   s = sqrt(alpha) = 1/sqrt(1/lambda + Ix^2 + Iy^2)
// so that the oracle's update  u' = ub - Ix*(Ix*ub + Iy*vb + It)*alpha  becomes
//   q = al*ub + be*vb + ga,   u' = ub - al*q,   v' = vb - be*q
// -- the same linear map with the factor alpha split evenly over its two uses: three registers
// and four fused multiply-adds per pixel instead of four registers and five operations.
__device__ __forceinline__ void sweep_coefs(uint32_t c, float ilambda, float &al, float &be, float &ga)
{
    float Ix, Iy, It;
    unpack_deriv(c, Ix, Iy, It);
#if HS_EXACT_SQRT_ALPHA
    const float s = sqrtf(alpha_of(Ix, Iy, ilambda));
#else
    // one v_rsq_f32 (1 ulp) instead of an IEEE division and an IEEE square root (~19 instructions per pixel,
    // a tenth of a 20-sweep launch); the argument is never denormal (the host keeps ilambda >= FLT_MIN)
    const float s = __builtin_amdgcn_rsqf(ilambda + (Ix * Ix + Iy * Iy));
#endif
    al = Ix * s;
    be = Iy * s;
    ga = It * s;
}

// One Jacobi update in the form above.
//
// Canonical order of the 4-neighbour sum (every kernel uses it, so all variants agree bit for bit): the four
// neighbours are added as two DIAGONAL pairs, chosen by the pixel's checkerboard parity (x + y) & 1:
//   even pixel:  ((D + R) + (U + L)) * 0.25        odd pixel:  ((D + L) + (U + R)) * 0.25
// Why pairs: the pair D(y,x) + R(y,x) = u(y+1,x) + u(y,x+1) of an even pixel is also the pair U + L of the pixel
// (y+1,x+1) -- even as well -- and likewise along the other diagonal for odd pixels, so a kernel that holds whole
// rows computes ONE such "cross sum" per pixel and row boundary and adds two of them per pixel: 2 additions per
// pixel and plane instead of 3 (hs_kernels_strip.hip.h).  Why the checkerboard: under the even reflections
// x -> -1-x and y -> -1-y the parity flips together with the roles of L/R and U/D, so a mirrored pixel forms the
// same two pair sums as its source -- the reflection halo of the strip kernels stays a reflection bit for bit.
// par = (x + y) & 1 of the pixel.
__device__ __forceinline__ void update_cv(int par, float uL, float uR, float uU, float uD, float vL,
                                          float vR, float vU, float vD, float al, float be,
                                          float ga, float &un, float &vn)
{
    const float su = par ? ((uD + uL) + (uU + uR)) : ((uD + uR) + (uU + uL));
    const float sv = par ? ((vD + vL) + (vU + vR)) : ((vD + vR) + (vU + vL));
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

// Sum of lanes 0..15 (the other lanes must hold 0), returned wave-uniform: row_shr prefix sums leave it in lane 15.
__device__ __forceinline__ float wave_sum16(float x)
{
#define HS_DPP_ADD(ctrl) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xF, 0xF, true))
    HS_DPP_ADD(0x111); // row_shr:1
    HS_DPP_ADD(0x112); // row_shr:2
    HS_DPP_ADD(0x114); // row_shr:4
    HS_DPP_ADD(0x118); // row_shr:8
#undef HS_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 15));
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
__device__ __forceinline__ void unpack_classic_deriv(uint32_t w, float &ex, float &ey, float &et); // hs_kernels_classic.hip.h
template <bool CLASSIC>
__global__ __launch_bounds__(256) void k_unpack_deriv(const uint32_t *__restrict__ coef,
                                                      float *__restrict__ dx,
                                                      float *__restrict__ dy,
                                                      float *__restrict__ dt, int W, int H, int P)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    float Ix, Iy, It;
    if (CLASSIC) unpack_classic_deriv(coef[(long long)y * P + x], Ix, Iy, It);
    else unpack_deriv(coef[(long long)y * P + x], Ix, Iy, It);
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
                                                       unsigned *__restrict__ eps_out, int org, int ey0, int ey1)
{
    // ey0, ey1: rows whose changes count for Eps (hsflow_set_eps_rows; the whole frame by default)
    // org: frame row of this context's row 0, modulo 2 (row slabs: the checkerboard of update_cv is the whole frame's)
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
            update_cv((k + y + org) & 1, wu[k], uR, au[k], bu[k], wv[k], vR, av[k], bv[k], al, be, ga, nu[k], nv[k]); // x0 % 4 == 0
            if (EPS && x0 + k < W && y >= ey0 && y < ey1)
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
    int org;                // frame row of this context's row 0, modulo 2 (checkerboard phase of update_cv)
};

enum : unsigned { F_ACTIVE = 1u, F_CORE = 2u, F_GU = 4u, F_GD = 8u, F_GL = 16u, F_GR = 32u };
// bits [9:8] of the flag word: position pr of image column W-1 inside the group (valid with F_GR); bit 10: y & 1

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
                f |= (unsigned)((y + g.org) & 1) << 10;
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
                const int py = (int)((f >> 10) & 1u); // parity of the group's image row (x0 % 4 == 0)
                update_cv(py, uL, cu[k].y, uu.x, ud.x, vL, cv[k].y, vu.x, vd.x, cAl[k][0], cBe[k][0], cGa[k][0], nu[0], nv[0]);
                update_cv(py ^ 1, cu[k].x, cu[k].z, uu.y, ud.y, cv[k].x, cv[k].z, vu.y, vd.y, cAl[k][1], cBe[k][1], cGa[k][1], nu[1], nv[1]);
                update_cv(py, cu[k].y, cu[k].w, uu.z, ud.z, cv[k].y, cv[k].w, vu.z, vd.z, cAl[k][2], cBe[k][2], cGa[k][2], nu[2], nv[2]);
                update_cv(py ^ 1, cu[k].z, uR, uu.w, ud.w, cv[k].z, vR, vu.w, vd.w, cAl[k][3], cBe[k][3], cGa[k][3], nu[3], nv[3]);
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

} // namespace hsk

#include "hs_kernels_strip.hip.h"

namespace hsk {

// Eps of every sweep = maximum over that sweep's row of per-workgroup values (bit patterns of
// non-negative floats order like unsigned integers).  One workgroup per sweep.
// Row r holds cnt_first valid words for r < n_first, else cnt_last (launches of different plans have
// different workgroup counts; words beyond are never read, so nothing needs clearing).  `out` is
// page-locked host memory seen through its device address: the words are on the host when the stream drains.
// seq / host_mark (may be null): the last of the kernel's workgroups to finish also plays k_mark_done (below) -- an asynchronous
// ITER|EPS solve of a marked context (hsflow_set_async_reduce) then needs no marker kernel of its own.
__global__ __launch_bounds__(256) void k_eps_reduce(const unsigned *__restrict__ tiles, int stride,
                                                    unsigned *__restrict__ out, int n_first, int cnt_first, int cnt_last,
                                                    unsigned *__restrict__ seq, unsigned *__restrict__ host_mark)
{
    __shared__ unsigned part[4];
    const unsigned *row = tiles + (size_t)blockIdx.x * stride;
    const int n = (int)blockIdx.x < n_first ? cnt_first : cnt_last;
    unsigned m = 0;
    for (int i = threadIdx.x; i < n; i += 256) m = max(m, row[i]);
    float f = wave_max(__uint_as_float(m));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = __float_as_uint(f);
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x] = max(max(part[0], part[1]), max(part[2], part[3]));
        __threadfence_system();
        if (seq) { // (every workgroup's word is on the host before its count: the last one's marker comes after all of them)
            if (atomicAdd(&seq[1], 1u) == gridDim.x - 1u) {
                seq[1] = 0u;
                const unsigned s = seq[0] + 1u;
                seq[0] = s;
                __hip_atomic_store(host_mark, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// "Everything this context enqueued before me is done": counts up a device word and writes the count to page-locked host
// memory, where the host polls it (hsflow_wait_solve of a context whose stream is shared: a stream-wide wait would also
// wait for what other contexts queued later, and an event record between the solves of a stream costs the stream 6 %,
// tools/event_cost.py).
__global__ void k_mark_done(unsigned *__restrict__ seq, unsigned *__restrict__ host_mark)
{
    if (threadIdx.x == 0) {
        const unsigned s = seq[0] + 1u;
        seq[0] = s;
        __hip_atomic_store(host_mark, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

} // namespace hsk
