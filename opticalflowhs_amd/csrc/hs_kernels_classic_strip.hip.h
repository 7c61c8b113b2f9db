// hs_kernels_classic_strip.hip.h -- the classic mode (Kernels.cl semantics, hs_kernels_classic.hip.h) as a REGISTER STRIP
// kernel: T sweeps per launch on a 256-column x NW*R-row region whose rows live in VGPRs (a lane owns 4 consecutive
// columns of R rows: u, v, Ex, Ey, Et -- loaded as one packed word per pixel, pack_classic_deriv -- and the denominator), left / right neighbours by DPP, the rows of the
// wavefronts above and below through one LDS exchange and one barrier per sweep -- the skeleton of k_jacobi_strip.
//   u_v_avgKernel      OpticalFlowHS/Kernels.cl:43-68   ((L+R)+U)+D and ((UL+UR)+DL)+DR, 1/6 and 1/12
//   u_v_updateKernel   OpticalFlowHS/Kernels.cl:71-90   t = (Ex*ua + Ey*va + Et) / (alpha^2 + Ex^2 + Ey^2)
// Evaluation order = Kernels.cl source order without contraction and an IEEE division = oracle/hs_classic_oracle.c
// (bit-exact parity; tests/test_gpu_parity.py).
//
// What differs from the CV-mode strip kernel is the border.  There the 4-neighbour sum is symmetric under reflection,
// so a mirrored halo IS the replicate border and costs nothing.  Kernels.cl's sums are ordered -- ((UL+UR)+DL)+DR is
// not what its mirror image computes -- so the clamps of Tex2D (Kernels.cl:2-9) are applied where a value is READ:
//   columns: the lane that holds column 0 takes its own first pixel as its left neighbour, the lane that ends on column
//            W-1 its own last pixel as its right neighbour (one select per row and plane on the DPP result); when
//            W % 4 != 0 the pixels right of column W-1 inside a lane are kept equal to column W-1 after every update;
//   rows:    the planner aligns the regions so that image row 0 is always register row 0 of a wavefront and image row
//            H-1 register row R-1 of one (row halo = a multiple of R, last tile row aligned to the bottom); such a
//            wavefront reads its OWN published edge row as the row above / below -- an LDS address, no instruction.
//            A region that meets the bottom border unaligned (the tile row above the last one) is never trusted near
//            it: the planner gives those rows to the last tile row (ClassicStripGeom::CH rows, re-based at H - CH).
// In-place update, top to bottom, without copies: everything row r+1 needs from the OLD row r -- its L+R sums (the
// UL+UR of row r+1) and (L+R of row r+1) + (centre of row r) -- is formed before row r is overwritten; both are sums
// the stencil needs anyway.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hsk {

struct ClassicStripGeom {
    int W, H, P;
    long long plane;
    int T, TH, HX;      // sweeps per launch; row halo (multiple of R, >= T); column halo (multiple of 4, >= T)
    int CW, CH;         // core = (256 - 2*HX) x (NW*R - 2*TH)
    int NW;             // wavefronts per workgroup
    int tiles_x, tiles_y;
    int ylast;          // first region row of the LAST tile row (bottom-aligned: H - CH - TH; -TH if there is one tile row)
    int zero_in;        // incoming flow is identically zero: u_in = v_in = ONE row of zeros (at least P floats), read with pitch 0
};

// The IEEE division t / den of Kernels.cl:85 with the denominator's share of the work done once per pixel.  This is the
// compiler's own correctly rounded sequence (v_div_scale, v_rcp, two refinement steps of the reciprocal, quotient,
// two residual steps, v_div_fmas, v_div_fixup) with the three instructions that depend on the denominator alone hoisted
// out of the sweep loop -- legal because the denominator alpha^2 + Ex^2 + Ey^2 is a normal number far from the ends of
// the exponent range (the host admits 2^-40 <= alpha^2 <= 2^40 to these kernels), so v_div_scale never rescales IT; the
// numerator keeps its v_div_scale (tiny numerators -- flow that has only just reached a static area -- are scaled by
// 2^64 for the residuals and scaled back by v_div_fmas, as in the compiler's sequence).  Same instructions on the same
// operands: the same bits (tests: bit-exact against oracle/hs_classic_oracle.c, which divides with `/`).
// It costs one more register per pixel: used by the shapes that have them (R = 2, 3 at 1024 threads, 4 at 768, 6 at 512).
__device__ __forceinline__ float refined_rcp(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}
__device__ __forceinline__ float div_by(float n, float d, float r)
{
    bool scaled;
    const float ns = __builtin_amdgcn_div_scalef(n, d, true, &scaled); // the numerator, times 2^64 if the residuals would underflow
    const float q0 = ns * r;
    const float e0 = __builtin_fmaf(-d, q0, ns);
    const float q1 = __builtin_fmaf(e0, r, q0);
    const float e1 = __builtin_fmaf(-d, q1, ns);
    return __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e1, r, q1, scaled), d, n);
}

// Two quotients at once: the four multiply / fused multiply-add steps of div_by as packed instructions (v_pk_mul_f32,
// v_pk_fma_f32 round each half exactly like their scalar forms).
typedef float f2c __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2c div_by2(f2c n, f2c d, f2c r)
{
    bool sx, sy;
    const f2c ns = f2c{__builtin_amdgcn_div_scalef(n.x, d.x, true, &sx), __builtin_amdgcn_div_scalef(n.y, d.y, true, &sy)};
    const f2c q0 = ns * r;
    const f2c e0 = __builtin_elementwise_fma(-d, q0, ns);
    const f2c q1 = __builtin_elementwise_fma(e0, r, q0);
    const f2c e1 = __builtin_elementwise_fma(-d, q1, ns);
    return f2c{__builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e1.x, r.x, q1.x, sx), d.x, n.x),
               __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e1.y, r.y, q1.y, sy), d.y, n.y)};
}

// The shapes without a register for the reciprocal divide with the complete sequence, two quotients at a time: the
// compiler's own expansion of `/` (v_div_scale of both operands, v_rcp, two refinement steps, quotient, two residual
// steps, v_div_fmas, v_div_fixup) with its six multiply / fused multiply-add steps packed.
__device__ __forceinline__ f2c div_full2(f2c n, f2c d)
{
    bool sx, sy, tx, ty;
    const f2c ds = f2c{__builtin_amdgcn_div_scalef(n.x, d.x, false, &tx), __builtin_amdgcn_div_scalef(n.y, d.y, false, &ty)};
    const f2c ns = f2c{__builtin_amdgcn_div_scalef(n.x, d.x, true, &sx), __builtin_amdgcn_div_scalef(n.y, d.y, true, &sy)};
    const f2c r0 = f2c{__builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y)};
    const f2c e = __builtin_elementwise_fma(-ds, r0, f2c{1.0f, 1.0f});
    const f2c r = __builtin_elementwise_fma(e, r0, r0);
    const f2c q0 = ns * r;
    const f2c e0 = __builtin_elementwise_fma(-ds, q0, ns);
    const f2c q1 = __builtin_elementwise_fma(e0, r, q0);
    const f2c e1 = __builtin_elementwise_fma(-ds, q1, ns);
    return f2c{__builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e1.x, r.x, q1.x, sx), d.x, n.x),
               __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e1.y, r.y, q1.y, sy), d.y, n.y)};
}

struct CWin { float l, c0, c1, c2, c3, r; }; // columns x0-1 .. x0+4 of one row of one plane
struct CH4 { float h0, h1, h2, h3; };         // L + R of the four pixels

template <int R, int NTMAX, bool WRITE_V, bool GHOST> // GHOST: W % 4 != 0 (pixels right of column W-1 inside a lane)
__global__ __launch_bounds__(NTMAX) void k_classic_strip(const uint32_t *__restrict__ coef, const float *__restrict__ u_in,
                                                         const float *__restrict__ v_in, float *__restrict__ u_out,
                                                         float *__restrict__ v_out, const ClassicStripGeom g,
                                                         const float alpha2)
{
    extern __shared__ __attribute__((aligned(16))) float4 ex[]; // [2][NW][4][64]: {row 0: u, v; row R-1: u, v}
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NW = g.NW;
    const int tpp = g.tiles_x * g.tiles_y;
    const int tile = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    const int pair = tile / tpp;
    const int t2 = tile - pair * tpp;
    const int by = t2 / g.tiles_x, bx = t2 - by * g.tiles_x;
    const bool lastrow = by == g.tiles_y - 1;
    const int ry0 = lastrow ? g.ylast : by * g.CH - g.TH;
    const int x0 = bx * g.CW - g.HX + 4 * lane;
    const int y0 = ry0 + w * R;
    const long long base = (long long)pair * g.plane;
    // rows this tile row owns (stores): the last tile row takes [H - CH, H), the others stop there
    const int own_lo = (lastrow && g.tiles_y > 1) ? g.H - g.CH : by * g.CH;
    const int own_hi = lastrow ? g.H : min((by + 1) * g.CH, g.H - g.CH);

    // Loads: every lane reads aligned 16-byte groups from inside the image -- its own columns, or for a lane that lies
    // outside the image the nearest group (its values are never read by a pixel that counts: the clamps above apply where
    // the neighbour is READ).  Only the lane that straddles column W-1 (W % 4 != 0) needs the clamped columns themselves:
    // four scalar loads per plane, after all the rows' vector loads are in flight.
    const bool xin = x0 >= 0 && x0 + 3 < g.W;
    const bool slow = GHOST && !xin && ((x0 >= 0 && x0 < g.W) || g.W < 4);
    const int xg = g.W < 4 ? 0 : clampi(x0, 0, (g.W - 4) & ~3);
    float4 lu[R], lv[R];
    uint4 lc[R]; // packed derivatives (pack_classic_deriv)
#pragma unroll
    for (int r = 0; r < R; r++) {
        const long long row = base + (long long)clampi(y0 + r, 0, g.H - 1) * g.P;
        // (zero_in: u_in = v_in = ONE row of zeros, read with pitch 0 -- the loads stay unconditional: under a branch per
        // row the compiler waits for each row's two loads before it issues the next row's, see k_jacobi_strip)
        const long long row_uv = g.zero_in ? 0ll : row;
        lu[r] = *(const float4 *)(u_in + row_uv + xg);
        lv[r] = *(const float4 *)(v_in + row_uv + xg);
        lc[r] = *(const uint4 *)(coef + row + xg);
    }
    if (GHOST && __builtin_amdgcn_ballot_w64(slow) != 0) { // Tex2D clamp (Kernels.cl:2-9): columns right of W-1 read column W-1
        if (slow) {
            const int xa = clampi(x0, 0, g.W - 1), xb = clampi(x0 + 1, 0, g.W - 1), xc = clampi(x0 + 2, 0, g.W - 1),
                      xd = clampi(x0 + 3, 0, g.W - 1);
#pragma unroll
            for (int r = 0; r < R; r++) {
                const long long row = base + (long long)clampi(y0 + r, 0, g.H - 1) * g.P;
                {
                    const long long ru = g.zero_in ? 0ll : row;
                    lu[r] = make_float4(u_in[ru + xa], u_in[ru + xb], u_in[ru + xc], u_in[ru + xd]);
                    lv[r] = make_float4(v_in[ru + xa], v_in[ru + xb], v_in[ru + xc], v_in[ru + xd]);
                }
                lc[r] = make_uint4(coef[row + xa], coef[row + xb], coef[row + xc], coef[row + xd]);
            }
        }
    }
    float u[R][4], v[R][4], cEx[R][4], cEy[R][4], cEt[R][4], cDn[R][4];
    constexpr bool RCP = R <= 4 || R == 6; // the division with the reciprocal precomputed (div_by2)
    float cRc[RCP ? R : 1][4];
#pragma unroll
    for (int r = 0; r < R; r++) {
        u[r][0] = lu[r].x; u[r][1] = lu[r].y; u[r][2] = lu[r].z; u[r][3] = lu[r].w;
        v[r][0] = lv[r].x; v[r][1] = lv[r].y; v[r][2] = lv[r].z; v[r][3] = lv[r].w;
        unpack_classic_deriv(lc[r].x, cEx[r][0], cEy[r][0], cEt[r][0]);
        unpack_classic_deriv(lc[r].y, cEx[r][1], cEy[r][1], cEt[r][1]);
        unpack_classic_deriv(lc[r].z, cEx[r][2], cEy[r][2], cEt[r][2]);
        unpack_classic_deriv(lc[r].w, cEx[r][3], cEy[r][3], cEt[r][3]);
#pragma unroll
        for (int p = 0; p < 4; p++) {
            cDn[r][p] = alpha2 + cEx[r][p] * cEx[r][p] + cEy[r][p] * cEy[r][p]; // Kernels.cl:85
            if constexpr (RCP) cRc[r][p] = refined_rcp(cDn[r][p]);
        }
    }
    // Rows this wavefront stores, and how far the wavefront is from them: the row halo is a whole number of wavefronts
    // (TH % R == 0), so a wavefront is all core or all halo, and a halo wavefront d rows away from the rows its tile owns
    // is needed through sweep T-1-d only -- one wave-uniform branch around the straight-line sweep of its R rows.
    unsigned rowown = 0;
#pragma unroll
    for (int r = 0; r < R; r++)
        if (y0 + r >= own_lo && y0 + r < own_hi) rowown |= 1u << r;
    const int wdist = y0 + R - 1 < own_lo ? own_lo - (y0 + R - 1) : (y0 >= own_hi ? y0 - (own_hi - 1) : 0);
    const bool laneown = x0 < g.W && 4 * lane >= g.HX && 4 * lane < g.HX + g.CW;
    // column clamps, applied where the neighbour is read
    const bool isL = x0 == 0, isR = x0 + 3 == g.W - 1;
    const int pr = g.W - 1 - x0; // W % 4 != 0: image columns of the lane that straddles column W-1 are 0..pr
    auto window = [&](float c0, float c1, float c2, float c3) __attribute__((always_inline)) {
        CWin o;
        const float pl = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c3), 0x138, 0xF, 0xF, true)); // lane-1
        const float nr = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c0), 0x130, 0xF, 0xF, true)); // lane+1
        o.l = isL ? c0 : pl;
        o.r = isR ? c3 : nr;
        o.c0 = c0; o.c1 = c1; o.c2 = c2; o.c3 = c3;
        return o;
    };
    auto hsum = [&](const CWin &q) __attribute__((always_inline)) { // Kernels.cl:55: L + R
        return CH4{q.l + q.c1, q.c0 + q.c2, q.c1 + q.c3, q.c2 + q.r};
    };

#define HC_PUBLISH(buf)                                                                            \
    do {                                                                                           \
        float4 *exw = ex + ((size_t)((buf) * NW + w) * 4) * 64 + lane;                             \
        exw[0] = make_float4(u[0][0], u[0][1], u[0][2], u[0][3]);                                  \
        exw[64] = make_float4(v[0][0], v[0][1], v[0][2], v[0][3]);                                 \
        exw[128] = make_float4(u[R - 1][0], u[R - 1][1], u[R - 1][2], u[R - 1][3]);                \
        exw[192] = make_float4(v[R - 1][0], v[R - 1][1], v[R - 1][2], v[R - 1][3]);                \
    } while (0)
    HC_PUBLISH(0);
    __syncthreads();
    // Whose slot is the row above register row 0 / below register row R-1?  The neighbouring wavefront's edge row -- or
    // this wavefront's own one: at the image border that IS the clamp, at the region's edge it is junk the validity
    // argument tolerates.
    const bool top0 = y0 == 0 || w == 0, botL = y0 + R - 1 == g.H - 1 || w == NW - 1;
    const int wu = top0 ? w : w - 1, su = top0 ? 0 : 2;
    const int wd = botL ? w : w + 1, sd = botL ? 2 : 0;
    const float c6 = (float)(1.0 / 6), c12 = (float)(1.0 / 12); // Kernels.cl:55,57 (double literals, converted)

#pragma unroll 1
    for (int s = 0; s < g.T; s++) {
        if (wdist <= g.T - 1 - s) { // wave-uniform
            const float4 *eu = ex + ((size_t)((s & 1) * NW + wu) * 4 + su) * 64 + lane;
            const float4 *ed = ex + ((size_t)((s & 1) * NW + wd) * 4 + sd) * 64 + lane;
            const float4 hu4 = eu[0], hv4 = eu[64]; // old row above the strip
            const float4 du4 = ed[0], dv4 = ed[64]; // old row below the strip
            const CWin Uu = window(hu4.x, hu4.y, hu4.z, hu4.w), Uv = window(hv4.x, hv4.y, hv4.z, hv4.w);
            CH4 hpu = hsum(Uu), hpv = hsum(Uv); // L+R sums of the old row r-1
            const CWin Cu = window(u[0][0], u[0][1], u[0][2], u[0][3]), Cv = window(v[0][0], v[0][1], v[0][2], v[0][3]);
            CH4 hcu = hsum(Cu), hcv = hsum(Cv); // ... of the old row r
            // (L+R of row r) + (centre of old row r-1)
            f2c peuA = f2c{hcu.h0, hcu.h1} + f2c{Uu.c0, Uu.c1}, peuB = f2c{hcu.h2, hcu.h3} + f2c{Uu.c2, Uu.c3};
            f2c pevA = f2c{hcv.h0, hcv.h1} + f2c{Uv.c0, Uv.c1}, pevB = f2c{hcv.h2, hcv.h3} + f2c{Uv.c2, Uv.c3};
#pragma unroll
            for (int r = 0; r < R; r++) {
                float nu0, nu1, nu2, nu3, nv0, nv1, nv2, nv3; // centres of the old row below
                if (r + 1 < R) {
                    nu0 = u[r + 1][0]; nu1 = u[r + 1][1]; nu2 = u[r + 1][2]; nu3 = u[r + 1][3];
                    nv0 = v[r + 1][0]; nv1 = v[r + 1][1]; nv2 = v[r + 1][2]; nv3 = v[r + 1][3];
                } else {
                    nu0 = du4.x; nu1 = du4.y; nu2 = du4.z; nu3 = du4.w;
                    nv0 = dv4.x; nv1 = dv4.y; nv2 = dv4.z; nv3 = dv4.w;
                }
                const CWin Nu = window(nu0, nu1, nu2, nu3), Nv = window(nv0, nv1, nv2, nv3);
                CH4 hnu, hnv;
                f2c pnuA, pnuB, pnvA, pnvB;
                if (r + 1 < R) { // for the row below, before this row is overwritten
                    hnu = hsum(Nu); hnv = hsum(Nv);
                    pnuA = f2c{hnu.h0, hnu.h1} + f2c{u[r][0], u[r][1]}; pnuB = f2c{hnu.h2, hnu.h3} + f2c{u[r][2], u[r][3]};
                    pnvA = f2c{hnv.h0, hnv.h1} + f2c{v[r][0], v[r][1]}; pnvB = f2c{hnv.h2, hnv.h3} + f2c{v[r][2], v[r][3]};
                }
                // Kernels.cl:55-63: c6 * (((L + R) + U) + D) + c12 * (((UL + UR) + DL) + DR); pairs of pixels as packed
                // operations where both operands are register pairs (the edge term, the weights, the update): -15 % per
                // solve.  (Holding a lane's pixels as the pairs (p0, p3), (p1, p2) makes the L+R sums and the carried sums
                // packed as well -- 8 instructions fewer per row -- but measured no faster at 512 threads and slower at
                // 1024, where it spills: not kept.  Nor was keeping every window in the shifted pairs (l, p0), (p1, p2),
                // (p3, r) too (v_pk_mov_b32), which packs the L+R sums and the corner term: the moves eat the saving.)
                const f2c k6 = f2c{c6, c6}, k12 = f2c{c12, c12};
                const f2c euA = peuA + f2c{Nu.c0, Nu.c1}, euB = peuB + f2c{Nu.c2, Nu.c3};
                const f2c evA = pevA + f2c{Nv.c0, Nv.c1}, evB = pevB + f2c{Nv.c2, Nv.c3};
                const f2c cuA = f2c{(hpu.h0 + Nu.l) + Nu.c1, (hpu.h1 + Nu.c0) + Nu.c2}, cuB = f2c{(hpu.h2 + Nu.c1) + Nu.c3, (hpu.h3 + Nu.c2) + Nu.r};
                const f2c cvA = f2c{(hpv.h0 + Nv.l) + Nv.c1, (hpv.h1 + Nv.c0) + Nv.c2}, cvB = f2c{(hpv.h2 + Nv.c1) + Nv.c3, (hpv.h3 + Nv.c2) + Nv.r};
                const f2c uaA = k6 * euA + k12 * cuA, uaB = k6 * euB + k12 * cuB;
                const f2c vaA = k6 * evA + k12 * cvA, vaB = k6 * evB + k12 * cvB;
#pragma unroll
                for (int h = 0; h < 2; h++) { // Kernels.cl:84-86, two pixels at a time
                    const f2c ua2 = h ? uaB : uaA, va2 = h ? vaB : vaA;
                    const f2c ex2 = f2c{cEx[r][2 * h], cEx[r][2 * h + 1]}, ey2 = f2c{cEy[r][2 * h], cEy[r][2 * h + 1]},
                              et2 = f2c{cEt[r][2 * h], cEt[r][2 * h + 1]};
                    f2c t2 = ex2 * ua2 + ey2 * va2 + et2;
                    if constexpr (RCP) {
                        t2 = div_by2(t2, f2c{cDn[r][2 * h], cDn[r][2 * h + 1]}, f2c{cRc[r][2 * h], cRc[r][2 * h + 1]});
                    } else {
                        t2 = div_full2(t2, f2c{cDn[r][2 * h], cDn[r][2 * h + 1]});
                    }
                    const f2c un = ua2 - ex2 * t2;
                    u[r][2 * h] = un.x; u[r][2 * h + 1] = un.y;
                    if (WRITE_V) { // restored: the reference kernel forgot this line
                        const f2c vn = va2 - ey2 * t2;
                        v[r][2 * h] = vn.x; v[r][2 * h + 1] = vn.y;
                    }
                }
                if (GHOST) { // pixels right of column W-1 inside a lane replicate column W-1
                    u[r][1] = pr == 0 ? u[r][0] : u[r][1]; v[r][1] = pr == 0 ? v[r][0] : v[r][1];
                    u[r][2] = (pr >= 0 && pr <= 1) ? u[r][1] : u[r][2]; v[r][2] = (pr >= 0 && pr <= 1) ? v[r][1] : v[r][2];
                    u[r][3] = (pr >= 0 && pr <= 2) ? u[r][2] : u[r][3]; v[r][3] = (pr >= 0 && pr <= 2) ? v[r][2] : v[r][3];
                }
                if (r + 1 < R) {
                    hpu = hcu; hpv = hcv;
                    hcu = hnu; hcv = hnv;
                    peuA = pnuA; peuB = pnuB; pevA = pnvA; pevB = pnvB;
                }
            }
        }
        if (s + 1 < g.T) {
            HC_PUBLISH((s + 1) & 1);
            __syncthreads();
        }
    }
#undef HC_PUBLISH

    if (laneown) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            if ((rowown >> r) & 1u) {
                const long long off = base + (long long)(y0 + r) * g.P + x0; // (planes are padded to the pitch)
                *(float4 *)(u_out + off) = make_float4(u[r][0], u[r][1], u[r][2], u[r][3]);
                *(float4 *)(v_out + off) = make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
            }
        }
    }
}

} // namespace hsk
