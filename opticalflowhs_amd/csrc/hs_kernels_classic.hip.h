// hs_kernels_classic.hip.h -- the reference's own OpenCL discretisation ("-cl" route), re-written
// for gfx950 with the v update restored:
//   ComputeDerivativesKernel   OpticalFlowHS/Kernels.cl:13-39   2x2x2 cube differences over both frames (k_deriv_classic_packed)
//   u_v_avgKernel              OpticalFlowHS/Kernels.cl:43-68   1/6 (W,E,N,S) + 1/12 (corners)
//   u_v_updateKernel           OpticalFlowHS/Kernels.cl:71-90   alpha^2 regulariser; writes u AND v here
// The two per-iteration kernels of the reference are fused into one pass (u_avg / v_avg never touch
// memory) and the planes are planar fp32 instead of float4.  Evaluation order = source order of
// Kernels.cl without contraction, i.e. exactly oracle/hs_classic_oracle.c (bit-exact parity).
// k_jacobi_classic: one sweep per launch, 9-point stencil straight from L1/L2 (HBM-bound at 12 (Ex,Ey,Et)
// + 8 + 8 bytes per pixel per sweep).  k_jacobi_classic_fused (below, the default): several sweeps per
// launch on an LDS tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hsk {

// The three derivatives of a pixel in ONE 32-bit word (what the strip kernel loads: 4 instead of 12 bytes per pixel and
// launch).  4*Ex, 4*Ey, 4*Et are integers in [-1020, 1020] (sums of four differences of 8-bit values), and each pixel
// of the 2x2x2 cube enters each of them exactly once, so all three have the SAME parity: 11 bits of 4*Ex, then 4*Ey and
// 4*Et without their lowest bit, 10 bits each.  0.25f * (float)n reproduces Kernels.cl:25-38's value bit for bit (the
// sums there are exact in fp32, a zero sum is +0).
__device__ __forceinline__ uint32_t pack_classic_deriv(float sx, float sy, float st) // the sums, before the 1/4
{
    const int ex4 = (int)sx, ey4 = (int)sy, et4 = (int)st;
    return ((uint32_t)ex4 & 0x7FFu) | (((uint32_t)(ey4 >> 1) & 0x3FFu) << 11) | (((uint32_t)(et4 >> 1) & 0x3FFu) << 21);
}
__device__ __forceinline__ void unpack_classic_deriv(uint32_t w, float &ex, float &ey, float &et)
{
    const int ex4 = (int)(w << 21) >> 21, par = ex4 & 1;
    const int ey4 = (((int)(w << 11) >> 22) << 1) | par, et4 = (((int)(w << 1) >> 22) << 1) | par;
    ex = 0.25f * (float)ex4; ey = 0.25f * (float)ey4; et = 0.25f * (float)et4;
}

__global__ __launch_bounds__(256) void k_deriv_classic_packed(const uint8_t *__restrict__ A, const uint8_t *__restrict__ B,
                                                              uint32_t *__restrict__ coef, int W, int H, int P, long long plane)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= W || y >= H) return;
    const long long base = (long long)blockIdx.z * plane;
    const int y1 = y < H - 1 ? y + 1 : H - 1; // Tex2D clamp (Kernels.cl:2-9)
    const uint8_t *a0 = A + base + (long long)y * P, *a1 = A + base + (long long)y1 * P;
    const uint8_t *b0 = B + base + (long long)y * P, *b1 = B + base + (long long)y1 * P;
    float ra0[5], ra1[5], rb0[5], rb1[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int xc = x0 + k < W - 1 ? x0 + k : W - 1;
        ra0[k] = (float)a0[xc]; ra1[k] = (float)a1[xc];
        rb0[k] = (float)b0[xc]; rb1[k] = (float)b1[xc];
    }
    uint32_t wd[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float a00 = ra0[k], a10 = ra0[k + 1], a01 = ra1[k], a11 = ra1[k + 1];
        const float b00 = rb0[k], b10 = rb0[k + 1], b01 = rb1[k], b11 = rb1[k + 1];
        wd[k] = pack_classic_deriv(a10 - a00 + a11 - a01 + b10 - b00 + b11 - b01, a01 - a00 + a11 - a10 + b01 - b00 + b11 - b10,
                                   b00 - a00 + b10 - a10 + b01 - a01 + b11 - a11);
    }
    *(uint4 *)(coef + base + (long long)y * P + x0) = make_uint4(wd[0], wd[1], wd[2], wd[3]);
}

// The packed plane as three fp32 planes of the same pitch: for the LDS-tile and the one-sweep kernels below (which
// read planes) and for hsflow_get_derivatives.
__global__ __launch_bounds__(256) void k_unpack_classic_deriv(const uint32_t *__restrict__ coef, float *__restrict__ Ex,
                                                              float *__restrict__ Ey, float *__restrict__ Et, int W, int H,
                                                              int P, long long plane)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= W || y >= H) return;
    const long long o = (long long)blockIdx.z * plane + (long long)y * P + x0;
    const uint4 w = *(const uint4 *)(coef + o);
    float ex[4], ey[4], et[4];
    unpack_classic_deriv(w.x, ex[0], ey[0], et[0]);
    unpack_classic_deriv(w.y, ex[1], ey[1], et[1]);
    unpack_classic_deriv(w.z, ex[2], ey[2], et[2]);
    unpack_classic_deriv(w.w, ex[3], ey[3], et[3]);
    *(float4 *)(Ex + o) = make_float4(ex[0], ex[1], ex[2], ex[3]);
    *(float4 *)(Ey + o) = make_float4(ey[0], ey[1], ey[2], ey[3]);
    *(float4 *)(Et + o) = make_float4(et[0], et[1], et[2], et[3]);
}

// WRITE_V = false is Kernels.cl exactly as shipped: u_v_updateKernel writes u only (Kernels.cl:86), so
// v keeps its starting value for ever (the ping-pong buffers just carry it along).  The product default
// restores the v update; the as-shipped form exists so that the HIP path can be held against the
// pictures the reference's OpenCL route wrote (tests/refpics.py).
template <bool ZERO, bool WRITE_V>
__global__ __launch_bounds__(256) void k_jacobi_classic(const float *__restrict__ Ex, const float *__restrict__ Ey,
                                                        const float *__restrict__ Et, const float *__restrict__ u_in,
                                                        const float *__restrict__ v_in, float *__restrict__ u_out,
                                                        float *__restrict__ v_out, int W, int H, int P,
                                                        long long plane, float alpha2)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= W || y >= H) return;
    const long long base = (long long)blockIdx.z * plane;
    const long long rows[3] = {base + (long long)(y > 0 ? y - 1 : 0) * P, base + (long long)y * P,
                               base + (long long)(y < H - 1 ? y + 1 : H - 1) * P};
    float wu[3][6], wv[3][6]; // columns x0-1 .. x0+4 (clamped) of rows y-1, y, y+1
#pragma unroll
    for (int j = 0; j < 3; j++) {
        float4 cu = make_float4(0.f, 0.f, 0.f, 0.f), cv = cu;
        float lu = 0.f, lv = 0.f, ru = 0.f, rv = 0.f;
        if (!ZERO) {
            cu = *(const float4 *)(u_in + rows[j] + x0);
            cv = *(const float4 *)(v_in + rows[j] + x0);
            lu = x0 > 0 ? u_in[rows[j] + x0 - 1] : cu.x;
            lv = x0 > 0 ? v_in[rows[j] + x0 - 1] : cv.x;
            ru = x0 + 4 < W ? u_in[rows[j] + x0 + 4] : 0.f;
            rv = x0 + 4 < W ? v_in[rows[j] + x0 + 4] : 0.f;
        }
        wu[j][0] = lu; wu[j][1] = cu.x; wu[j][2] = cu.y; wu[j][3] = cu.z; wu[j][4] = cu.w; wu[j][5] = ru;
        wv[j][0] = lv; wv[j][1] = cv.x; wv[j][2] = cv.y; wv[j][3] = cv.z; wv[j][4] = cv.w; wv[j][5] = rv;
    }
    const long long o = base + (long long)y * P + x0;
    const float4 e4x = *(const float4 *)(Ex + o), e4y = *(const float4 *)(Ey + o), e4t = *(const float4 *)(Et + o);
    const float ex[4] = {e4x.x, e4x.y, e4x.z, e4x.w}, ey[4] = {e4y.x, e4y.y, e4y.z, e4y.w}, et[4] = {e4t.x, e4t.y, e4t.z, e4t.w};
    const float c6 = (float)(1.0 / 6), c12 = (float)(1.0 / 12); // Kernels.cl:55,57 (double literals, converted)
    float nu[4], nv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // clamp to edge in x: the last image column is its own right neighbour
        const int r = (x0 + k >= W - 1) ? k + 1 : k + 2, c = k + 1, l = k;
        const float ua = c6 * (wu[1][l] + wu[1][r] + wu[0][c] + wu[2][c]) + c12 * (wu[0][l] + wu[0][r] + wu[2][l] + wu[2][r]);
        const float va = c6 * (wv[1][l] + wv[1][r] + wv[0][c] + wv[2][c]) + c12 * (wv[0][l] + wv[0][r] + wv[2][l] + wv[2][r]);
        float t = ex[k] * ua + ey[k] * va + et[k];
        t /= alpha2 + ex[k] * ex[k] + ey[k] * ey[k];
        nu[k] = ua - ex[k] * t;
        nv[k] = WRITE_V ? va - ey[k] * t // restored: the reference kernel forgot this line (Kernels.cl:84-86)
                        : wv[1][c];
    }
    *(float4 *)(u_out + o) = make_float4(nu[0], nu[1], nu[2], nu[3]);
    *(float4 *)(v_out + o) = make_float4(nv[0], nv[1], nv[2], nv[3]);
}

// ------------------------------------------------------------------------------------------
// T sweeps per launch on an LDS tile with a T-pixel halo: the classic-mode twin of k_jacobi_fused
// (same FusedGeom, same work decomposition: a lane owns K groups of 4 consecutive pixels whose own
// u, v and coefficients stay in VGPRs; LDS carries the two planes for the neighbours).  What
// differs is the stencil -- the 8-neighbour 1/6, 1/12 mean needs the rows above and below with
// their left / right neighbours (18 LDS values per plane and group instead of 10) -- the
// coefficients (Ex, Ey, Et as fp32 planes plus the precomputed denominator) and the IEEE division.
// Borders are true clamps (Tex2D, Kernels.cl:2-9): ghost rows in LDS for rows 0 and H-1, selects
// for columns 0 and W-1, columns right of W-1 inside a group replicate column W-1.
// Evaluation order = Kernels.cl source order = oracle/hs_classic_oracle.c (bit-exact).
// ------------------------------------------------------------------------------------------
template <int NT, int K, bool WRITE_V>
__global__ __launch_bounds__(NT) void k_jacobi_classic_fused(const float *__restrict__ Ex, const float *__restrict__ Ey,
                                                             const float *__restrict__ Et,
                                                             const float *__restrict__ u_in,
                                                             const float *__restrict__ v_in,
                                                             float *__restrict__ u_out, float *__restrict__ v_out,
                                                             const FusedGeom g, const float alpha2)
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

    int o[K], go[K];
    unsigned fl[K];
    float4 cu[K], cv[K];
    float cEx[K][4], cEy[K][4], cEt[K][4], cDen[K][4];

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
        if (valid) {
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
                // the derivative planes are padded to the pitch (a multiple of 64): a 16-byte load
                // of a group that straddles column W-1 stays inside the row
                const float4 e4x = *(const float4 *)(Ex + base + go[k]), e4y = *(const float4 *)(Ey + base + go[k]),
                             e4t = *(const float4 *)(Et + base + go[k]);
                const float ex[4] = {e4x.x, e4x.y, e4x.z, e4x.w}, ey[4] = {e4y.x, e4y.y, e4y.z, e4y.w},
                            et[4] = {e4t.x, e4t.y, e4t.z, e4t.w};
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    cEx[k][p] = ex[p]; cEy[k][p] = ey[p]; cEt[k][p] = et[p];
                    cDen[k][p] = alpha2 + ex[p] * ex[p] + ey[p] * ey[p]; // Kernels.cl:85
                }
            }
        }
    }
    __syncthreads(); // (region rows outside the image were filled with the clamped row: the first ghost rows)

    const float c6 = (float)(1.0 / 6), c12 = (float)(1.0 / 12); // Kernels.cl:55,57
    for (int s = 0; s < g.T; s++) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (fl[k] & F_ACTIVE) {
                const unsigned f = fl[k];
                const int pr = (f & F_GR) ? (int)((f >> 8) & 3u) : 7; // image column W-1 inside the group
                const bool gl = (f & F_GL) != 0, gr = pr == 3;
                float nu[4], nv[4];
                // one plane at a time: rows above / own / below as 6 values each (columns x0-1 .. x0+4)
                float ua[4], va[4];
#pragma unroll
                for (int pl = 0; pl < 2; pl++) {
                    const float *sp = pl ? sv : su;
                    const float4 own = pl ? cv[k] : cu[k];
                    const float4 up = *(const float4 *)(sp + o[k] - g.RS), dn = *(const float4 *)(sp + o[k] + g.RS);
                    float upL = sp[o[k] - g.RS - 1], upR = sp[o[k] - g.RS + 4];
                    float owL = sp[o[k] - 1], owR = sp[o[k] + 4];
                    float dnL = sp[o[k] + g.RS - 1], dnR = sp[o[k] + g.RS + 4];
                    if (gl) { upL = up.x; owL = own.x; dnL = dn.x; }   // clamp: column 0 is its own left
                    if (gr) { upR = up.w; owR = own.w; dnR = dn.w; }   // clamp: column W-1 is its own right
                    const float ru[6] = {upL, up.x, up.y, up.z, up.w, upR};
                    const float ro[6] = {owL, own.x, own.y, own.z, own.w, owR};
                    const float rd[6] = {dnL, dn.x, dn.y, dn.z, dn.w, dnR};
                    float *dst = pl ? va : ua;
#pragma unroll
                    for (int p = 0; p < 4; p++) // Kernels.cl:55-63, source order
                        dst[p] = c6 * (ro[p] + ro[p + 2] + ru[p + 1] + rd[p + 1]) +
                                 c12 * (ru[p] + ru[p + 2] + rd[p] + rd[p + 2]);
                }
#pragma unroll
                for (int p = 0; p < 4; p++) { // Kernels.cl:84-86
                    float t = cEx[k][p] * ua[p] + cEy[k][p] * va[p] + cEt[k][p];
                    t /= cDen[k][p];
                    nu[p] = ua[p] - cEx[k][p] * t;
                    nv[p] = WRITE_V ? va[p] - cEy[k][p] * t : (p == 0 ? cv[k].x : p == 1 ? cv[k].y : p == 2 ? cv[k].z : cv[k].w);
                }
                // columns right of W-1 inside the group replicate column W-1 (its clamped right neighbour)
                if (pr == 0) { nu[1] = nu[0]; nv[1] = nv[0]; }
                if (pr <= 1) { nu[2] = nu[1]; nv[2] = nv[1]; }
                if (pr <= 2) { nu[3] = nu[2]; nv[3] = nv[2]; }
                cu[k] = make_float4(nu[0], nu[1], nu[2], nu[3]);
                cv[k] = make_float4(nv[0], nv[1], nv[2], nv[3]);
            }
        }
        if (s == g.T - 1) break;
        __syncthreads(); // every LDS read of sweep s is done
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (fl[k] & F_ACTIVE) {
                *(float4 *)(su + o[k]) = cu[k];
                *(float4 *)(sv + o[k]) = cv[k];
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
