// hs_kernels_strip.hip.h -- the register-resident multi-sweep Jacobi kernels (k_jacobi_strip, the default, and
// its folded form k_jacobi_fold).  Included by hs_kernels.hip.h, which holds the shared arithmetic.
#pragma once
#include <type_traits>

// Last sweep of a launch as a second copy of the sweep code after the loop (the loop then publishes and
// meets the barrier unconditionally).  Folded kernel: far fewer spills (R = 4: 33 -> 7, R = 5: 33 -> 1) and
// 15 VGPRs less at R = 3; strip kernel: slightly worse (R = 5 / 16 wavefronts: 2 -> 11 spills), so not there.
// Scaled state: inside a launch the flow is carried as 4^k * u after k sweeps, so that the neighbour SUM is
// the average at the next scale and the multiplication by 0.25 disappears (the constant term ga is rescaled
// instead: one packed multiply per pair of pixels where there were two).  Powers of 4 commute with every
// rounding, so the result is bit-identical to the canonical arithmetic of update_cv<> -- except for values that
// would be denormal there (below 1.2e-38), which keep their bits here.  T <= 24: 4^24 = 2.8e14, no overflow for
// any flow a valid lambda allows.
#ifndef HS_SCALED
#define HS_SCALED 1
#endif
#ifndef HS_UNROLL2_FOLD
#define HS_UNROLL2_FOLD 1
#endif
#ifndef HS_PEEL_LAST_STRIP
#define HS_PEEL_LAST_STRIP 1
#endif
#ifndef HS_PEEL_LAST_FOLD
#define HS_PEEL_LAST_FOLD 1
#endif
#ifndef HS_CORE_PRIO /* s_setprio for the wavefronts that are busy in every sweep (0: off); 1: -0.3 % one context, -0.7 % in the stream */
#define HS_CORE_PRIO 1
#endif
#ifndef HS_SWEEP_STAMPS
#define HS_SWEEP_STAMPS 0
#endif
#ifndef HS_DIAG /* bit mask of diagnostic knobs in the strip sweep (timing experiments only, results are wrong): 1 no LDS
                   exchange, 2 no barrier, 4 no trapezoid gating, 8 no arithmetic */
#define HS_DIAG 0
#endif


namespace hsk {

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
    int zero_in;        // incoming flow is identically zero: u_in = v_in = ONE row of zeros (at least P floats), read with pitch 0
    int org;            // frame row of this context's row 0, modulo 2 (row slabs; checkerboard phase of update_cv)
    int ey0, ey1;       // rows [ey0, ey1) of the context whose changes count for Eps and the witness (hsflow_set_eps_rows: a row slab's
                        // owned rows -- its halo rows repeat a neighbour's, stale towards the slab's edge); the strip kernel only
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
typedef float f4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// The persistent form of the strip kernel (k_jacobi_strip_persist): ONE launch for a whole solve.  Every workgroup
// keeps its region's coefficients and flow in registers across PHASES of T sweeps; between two phases it publishes its
// core tile (write-through stores), raises its tile's phase counter, waits for the counters of its 8 neighbours and
// reloads only the halo pixels of its region -- no kernel boundary, no second read of the coefficients, no second
// v_rsq per pixel.  Replaces the host loop of HSOpticalFlowOpenCL.cpp:748-752 at the level of one kernel.
// Hand-off protocol (MI355X_MICROARCH.md, inter-workgroup visibility, first row of the table): every byte handed over
// is stored with `sc1` (write-through) and loaded with `sc1` (L1 bypass); every storing wave drains its stores
// (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, ONE lane then stores the counter `sc1`; the consumer polls
// with `sc1` loads from one wave and the other waves load after a workgroup barrier that wave joins.
// All workgroups must be resident at once (the host checks occupancy x CUs >= grid); every wait is bounded in time:
// a workgroup that gives up writes the error word and an abort word the others watch, and returns.
// ------------------------------------------------------------------------------------------
struct PersistArgs {
    unsigned *flags;        // [tiles] phases published by each tile, counted up across solves; [tiles] = abort word
    unsigned *err;          // error word in page-locked host memory: 1 = a wait timed out (results invalid)
    float *ub[2], *vb[2];   // phase p publishes into buffer p & 1; the last phase's buffer is the solve's output
    int n_phase, T_last;    // phases of the solve; sweeps of the last one (all others run g.T)
    unsigned wait_ticks;    // a wait gives up after this many 100 MHz ticks
};

// The hand-off's 16-byte accesses as raw buffer instructions with the sc1 bit (aux 16: write-through store, L1-bypassing
// load): the address is a scalar row offset + a per-lane byte offset, so no 64-bit vector arithmetic, and -- unlike
// inline assembly -- the compiler keeps track of the loads in flight.  One descriptor per plane (no bounds: the
// offsets are those of the ordinary loads and stores).
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(float *p)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, 0xFFFFFFFFu, 0x00020000);
}
__device__ __forceinline__ void store_f4_sc1(__amdgpu_buffer_rsrc_t rs, unsigned row_bytes, unsigned lane_bytes, f4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), rs, (int)lane_bytes, (int)row_bytes, 16);
}
__device__ __forceinline__ f4 load_f4_sc1(__amdgpu_buffer_rsrc_t rs, unsigned row_bytes, unsigned lane_bytes)
{
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane_bytes, (int)row_bytes, 16));
}
__device__ __forceinline__ unsigned load_u32_sc1(const unsigned *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The lane number from the hardware, at the point of use (two instructions): a register holding it across the sweep loop
// is one the R = 5 kernels do not have (the mask goes through an empty asm so that the optimiser cannot hoist the pair).
__device__ __forceinline__ int hs_lane_now()
{
    unsigned m = ~0u;
    asm volatile("" : "+s"(m));
    return (int)__builtin_amdgcn_mbcnt_hi(m, __builtin_amdgcn_mbcnt_lo(m, 0u));
}

__device__ __forceinline__ f2 f2_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 f2_swap(f2 a) { return __builtin_shufflevector(a, a, 1, 0); }
// old - new of a sweep, in the NEW value's scale (HS_SCALED: the old one is a factor 4 behind)
#if HS_SCALED
#define HS_DIFF(o, n) f2_fma((o), f2{4.0f, 4.0f}, -(n))
#define HS_DIFF1(o, n) fmaf(4.0f, (o), -(n))
#else
#define HS_DIFF(o, n) ((o) - (n))
#define HS_DIFF1(o, n) ((o) - (n))
#endif

// One row of one lane: pixels (p0,p3) = P and (p1,p2) = Q as two register pairs (p0 is an even image column: the
// region starts at a multiple of 4), so that all arithmetic is packed (2 pixels per v_pk_add/mul/fma_f32; cross_rows).
struct RowCoef { f2 alP, alQ, beP, beQ, gaP, gaQ; };

// The canonical 4-neighbour sum (update_cv, hs_kernels.hip.h) adds two diagonal pairs chosen by the pixel's
// checkerboard parity.  Every such pair straddles one row boundary and belongs to two pixels, so it is formed ONCE:
//   cross sum of the boundary between rows r and r+1, at pixel i of row r:
//       s_r[i] = x[r+1][i] + x[r][i+1]   (pixel (r,i) even)        s_r[i] = x[r+1][i] + x[r][i-1]   (odd)
//   neighbour sum of pixel (r,i):  s_r[i] + s_(r-1)[i-1]  (even)   s_r[i] + s_(r-1)[i+1]  (odd)
// A lane's four pixels p0..p3 of a row live in the two register pairs P = (p0, p3) and Q = (p1, p2).  With E = parity
// of pixel p0 of the upper row: for E = 0 every pixel's partner lies in the lane's OTHER pair (two plain packed adds);
// for E = 1 the inner pixels p1, p2 are each other's partners (one packed add with swapped halves) and the outer ones
// reach into the adjacent lanes (two DPP adds); the neighbour sum is the other way round.  Rows alternate, so a row
// costs 3 packed + 2 DPP adds per plane and no unpacked one (pairing (p0, p1), (p2, p3): 2 packed + 2 DPP + 2 plain;
// the straightforward sum: 4 + 2 + 2).
struct Cross { f2 uP, uQ, vP, vQ; };

// a + swap(b) in one packed add (written out: left to the compiler, the swap of a row that has just come from LDS is
// done with two moves first)
__device__ __forceinline__ f2 pk_add_swapped(f2 a, f2 b)
{
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// s of the boundary below row `c` (whose side neighbours it uses; `d` is the row underneath), E = parity of c's p0
template <int E>
__device__ __forceinline__ void cross_rows(Cross &s, const f2 cuP, const f2 cuQ, const f2 cvP, const f2 cvQ,
                                           const f2 duP, const f2 duQ, const f2 dvP, const f2 dvQ)
{
    if (E == 0) { // p0 + c1, p3 + c2 | p1 + c0, p2 + c3
        s.uP = duP + cuQ; s.uQ = duQ + cuP;
        s.vP = dvP + cvQ; s.vQ = dvQ + cvP;
    } else {      // p1 + c2, p2 + c1 | p0 + previous lane's c3, p3 + next lane's c0
        s.uQ = pk_add_swapped(duQ, cuQ);
        s.uP.x = duP.x + wave_from_prev_lane(cuP.y); s.uP.y = duP.y + wave_from_next_lane(cuP.x);
        s.vQ = pk_add_swapped(dvQ, cvQ);
        s.vP.x = dvP.x + wave_from_prev_lane(cvP.y); s.vP.y = dvP.y + wave_from_next_lane(cvP.x);
    }
}

// One row: neighbour sums from the cross sums below (sc) and above (sp) it, then the update in place.
// E = parity of the row's p0.  HS_SCALED: the sums ARE the averages at the next scale (file header).
template <int E>
__device__ __forceinline__ void strip_row_update(f2 &uP, f2 &uQ, f2 &vP, f2 &vQ, const Cross &sc, const Cross &sp, RowCoef &c)
{
    f2 tP, tQ, sP, sQ;
    if (E == 0) {
        tQ = pk_add_swapped(sc.uQ, sp.uQ);
        tP.x = sc.uP.x + wave_from_prev_lane(sp.uP.y); tP.y = sc.uP.y + wave_from_next_lane(sp.uP.x);
        sQ = pk_add_swapped(sc.vQ, sp.vQ);
        sP.x = sc.vP.x + wave_from_prev_lane(sp.vP.y); sP.y = sc.vP.y + wave_from_next_lane(sp.vP.x);
    } else {
        tP = sc.uP + sp.uQ; tQ = sc.uQ + sp.uP;
        sP = sc.vP + sp.vQ; sQ = sc.vQ + sp.vP;
    }
#if HS_SCALED
    const f2 ubP = tP, ubQ = tQ, vbP = sP, vbQ = sQ; // 4^(k+1) * average
#else
    const f2 ubP = tP * 0.25f, ubQ = tQ * 0.25f, vbP = sP * 0.25f, vbQ = sQ * 0.25f;
#endif
    const f2 qP = f2_fma(c.alP, ubP, f2_fma(c.beP, vbP, c.gaP));
    const f2 qQ = f2_fma(c.alQ, ubQ, f2_fma(c.beQ, vbQ, c.gaQ));
    uP = f2_fma(-c.alP, qP, ubP);
    vP = f2_fma(-c.beP, qP, vbP);
    uQ = f2_fma(-c.alQ, qQ, ubQ);
    vQ = f2_fma(-c.beQ, qQ, vbQ);
#if HS_SCALED
    c.gaP *= 4.0f; // the constant term at the next sweep's scale
    c.gaQ *= 4.0f;
#endif
}

// The packed derivative words of a lane's R rows straight from the two frames (DERIV launches: the first
// launch of a solve then needs no separate derivative pass).  Same arithmetic as k_deriv_cv, on the EVEN
// REFLECTION of frame A -- which is the replicate border at the image edge (A(-1) = A(0)) and, further out,
// yields the derivative of the mirrored pixel with the sign of the component across the mirror flipped; the
// flip is undone here because the halo wants the mirrored pixel's coefficients unchanged.  The column
// neighbours x0-1 and x0+4 come from the adjacent lanes (the region is contiguous in the reflected image);
// lanes 0 and 63 have none, so the region's outermost columns get a wrong Ix -- they are invalid after the
// first sweep anyway (HX >= T >= 1).  Host-side condition (strip_deriv_fusable): the image is at least as
// large as the region, so that one bounce suffices.
// Register row r of the lane is image row y0 + dir * r (dir = -1: the folded kernel's lower half, whose
// registers run bottom-up).
template <int R>
__device__ __forceinline__ void strip_derive(const uint8_t *__restrict__ fA, const uint8_t *__restrict__ fB,
                                             const StripGeom &g, long long base, int x0, int y0, int dir, bool xin,
                                             uint4 (&lc)[R])
{
    // How this lane reads its four columns of a frame row: 0 an aligned word (group inside the image), 1 an
    // aligned word read backwards (group wholly mirrored: left of the image, or right of it when W % 4 == 0),
    // 2 four reflected bytes (W % 4 != 0: the group that straddles column W-1 and those right of it).
    const int mode = xin ? 0 : ((x0 < 0 || (g.W & 3) == 0) ? 1 : 2);
    int xg = x0, xb[4] = {0, 0, 0, 0};
    unsigned flipx = 0; // pixels of the group that are mirrored columns: their Ix changes sign
    if (mode == 1) {
        xg = x0 < 0 ? -x0 - 4 : 2 * g.W - x0 - 4;
        if (xg < 0 || xg + 4 > g.W) xg = 0; // excluded by the host; keeps the load inside the row regardless
        flipx = 0xFu;
    } else if (mode == 2) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            xb[k] = mirror_index(x0 + k, g.W);
            if (x0 + k >= g.W) flipx |= 1u << k;
        }
    }
    // All word loads first, branch-free (mode 2 lanes read a harmless aligned word), so that they overlap;
    // then ONE wave-uniform branch in which the mode 2 lanes -- if the wavefront has any -- fetch their bytes.
    uint32_t wa[R + 2], wbv[R];
#pragma unroll
    for (int j = 0; j < R + 2; j++)
        wa[j] = *(const uint32_t *)(fA + base + (long long)mirror_index(y0 + dir * (j - 1), g.H) * g.P + (mode == 2 ? 0 : xg));
#pragma unroll
    for (int r = 0; r < R; r++)
        wbv[r] = *(const uint32_t *)(fB + base + (long long)mirror_index(y0 + dir * r, g.H) * g.P + (mode == 2 ? 0 : xg));
    if (mode == 1) {
#pragma unroll
        for (int j = 0; j < R + 2; j++) wa[j] = __builtin_bswap32(wa[j]);
#pragma unroll
        for (int r = 0; r < R; r++) wbv[r] = __builtin_bswap32(wbv[r]);
    }
    if (__builtin_amdgcn_ballot_w64(mode == 2) != 0) {
        if (mode == 2) {
            auto bytes = [&](const uint8_t *row) -> uint32_t {
                return (uint32_t)row[xb[0]] | ((uint32_t)row[xb[1]] << 8) | ((uint32_t)row[xb[2]] << 16) | ((uint32_t)row[xb[3]] << 24);
            };
#pragma unroll
            for (int j = 0; j < R + 2; j++) wa[j] = bytes(fA + base + (long long)mirror_index(y0 + dir * (j - 1), g.H) * g.P);
#pragma unroll
            for (int r = 0; r < R; r++) wbv[r] = bytes(fB + base + (long long)mirror_index(y0 + dir * r, g.H) * g.P);
        }
    }
    int a[R + 2][6]; // columns x0-1 .. x0+4 of the reflected rows y0 - dir, y0, ..., y0 + dir * R
#pragma unroll
    for (int j = 0; j < R + 2; j++) {
        const uint32_t wd = wa[j];
        a[j][0] = (int)((uint32_t)__builtin_amdgcn_update_dpp(0, (int)wd, 0x138, 0xF, 0xF, true) >> 24);
        a[j][5] = (int)((uint32_t)__builtin_amdgcn_update_dpp(0, (int)wd, 0x130, 0xF, 0xF, true) & 0xFFu);
#pragma unroll
        for (int k = 0; k < 4; k++) a[j][k + 1] = (int)((wd >> (8 * k)) & 0xFFu);
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int y = y0 + dir * r;
        const uint32_t wb = wbv[r];
        const bool yflip = (y < 0 || y >= g.H) != (dir < 0); // a[r + 2] is the row BELOW only when dir > 0
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int ix8 = (a[r][k + 2] + 2 * a[r + 1][k + 2] + a[r + 2][k + 2]) - (a[r][k] + 2 * a[r + 1][k] + a[r + 2][k]);
            int iy8 = (a[r + 2][k] + 2 * a[r + 2][k + 1] + a[r + 2][k + 2]) - (a[r][k] + 2 * a[r][k + 1] + a[r][k + 2]);
            if ((flipx >> k) & 1u) ix8 = -ix8;
            if (yflip) iy8 = -iy8;
            o[k] = pack_deriv(ix8, iy8, (int)((wb >> (8 * k)) & 0xFFu) - a[r + 1][k + 1]);
        }
        lc[r] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

template <int R, int NTMAX, int EPS, int E0, bool DERIV, bool PERSIST = false>
__device__ __forceinline__ void strip_body(const uint32_t *__restrict__ coef,
                                                        const float *__restrict__ u_in,
                                                        const float *__restrict__ v_in,
                                                        float *__restrict__ u_out,
                                                        float *__restrict__ v_out, const StripGeom g,
                                                        const float ilambda,
                                                        unsigned *__restrict__ eps_out, const int eps_stride,
                                                        unsigned long long *__restrict__ stamps,
                                                        const float eps_thr, const uint8_t *__restrict__ fA,
                                                        const uint8_t *__restrict__ fB, uint32_t *__restrict__ coef_w,
                                                        const PersistArgs pa = PersistArgs())
{
    static_assert(!PERSIST || EPS == 0 || EPS == 2, "the persistent form runs plain or witness phases");
    // EPS == 1: eps_out[sweep * eps_stride + workgroup] receives that workgroup's max |new - old| over
    // its core pixels (plain stores, no atomics; the host reduces over the workgroups afterwards).
    // EPS == 2 ("witness"): the cheap way to PROVE that no sweep of this launch had Eps < epsilon.
    // At the end of a sweep a wavefront whose first row is a core row asks "did u change by >= eps_thr
    // at column x0 of any lane of that row?" -- old and new value come back from the two exchange
    // buffers, so nothing is kept in registers for it -- and counts the
    // per-sweep answers in an SGPR.  Each |change| is a lower bound of that sweep's Eps, so a
    // wavefront that answered yes in EVERY sweep proves Eps_k >= eps_thr for all k of the launch;
    // eps_out[workgroup] = +inf if any wavefront of the workgroup did, else 0.
    // EPS == 3: witness for sweeps 0 .. T-2 (word eps_out[workgroup]) and the exact Eps of the LAST sweep
    // (word eps_out[eps_stride + workgroup]): what the final launch of an ITER|EPS solve needs -- "no early
    // stop before the budget" plus last_eps -- at the price of one measured sweep instead of T.
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
    // Synchronisation between sweeps: ONE workgroup barrier per sweep.  (Tried and measured slower, 0.220 vs 0.200 ms
    // at 1080p / 100 sweeps in round 1: per-wavefront counters in LDS, each wavefront waiting only for the strips
    // directly above and below.)
    const int tpp = g.tiles_x * g.tiles_y;
    const int tile = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    const int pair = tile / tpp;
    const int t2 = tile - pair * tpp;
    const int by = t2 / g.tiles_x, bx = t2 - by * g.tiles_x;
    const int x0 = bx * g.CW - g.HX + 4 * lane;
    const int y0 = by * g.CH - g.T + w * R;
    const long long base = (long long)pair * g.plane;
    const bool xin = (x0 >= 0) && (x0 + 3 < g.W); // the whole group lies inside the image
    // The sweep code is written for ONE checkerboard phase: pixel p0 of register row 0 must be an "E0" pixel
    // (x0 % 4 == 0, so that is the parity of the row, counted from the frame's row 0: g.org).  R even: every strip
    // of the launch starts on a row of the parity of T + org (CH is even), which is the template parameter E0 -- the
    // host launches the matching kernel.
    // R odd: strips alternate, E0 = 0 is the only instantiation, and a strip that starts on an odd row keeps its
    // rows in REVERSE order (register row r = image row y0 + R-1-r): mirrored in y, an odd pixel forms the pair
    // sums of an even one (update_cv: U and D swap roles), so the same code computes the same bits.  Only the
    // addressing knows: which image row a register row is, and which exchange slot holds a neighbour's edge row.
    const bool rev = (R & 1) != 0 && ((y0 + g.org) & 1) != 0; // wave-uniform
    auto img_row = [&](int r) { return rev ? R - 1 - r : r; }; // row of the strip that register row r holds

    f2 uP[R], uQ[R], vP[R], vQ[R];
    RowCoef cf[R];
    float4 lu[R], lv[R];
    uint4 lc[R];
    // Workgroup-uniform: does the region (core + halo) stick out of the image on the left or right?
    // Tiles that do not (the vast majority) load with plain aligned 16-byte accesses only.
    const int rx0 = bx * g.CW - g.HX;
    const bool xedge = !(rx0 >= 0 && rx0 + 256 <= g.W);
    // Where does this lane read its four columns?  Inside the image: at x0.  A group that lies completely outside the
    // image on the left mirrors onto an aligned group read backwards (columns -1-k <-> k); the same holds on the right
    // when W % 4 == 0: those lanes keep the 16-byte loads (from the mirrored address, components reversed afterwards).
    // Only groups that straddle column W-1 or sit right of it when W % 4 != 0 (and images narrower than the halo) need
    // four reflected scalar loads per plane ("slow").  ALL rows' loads are issued first, branch-free -- edge tiles then
    // load as fast as interior ones (with a branch per row their loads did not overlap: +5 000 cycles per launch on
    // the two edge tile columns, which every launch then waited for) -- and the fix-ups follow.
    int xg = x0;
    bool xrev = false, slow = false;
    if (xedge && !xin) {
        if (x0 < 0 && -x0 <= g.W) { xg = -x0 - 4; xrev = true; }
        else if (x0 >= g.W && (g.W & 3) == 0 && 2 * g.W - x0 - 4 >= 0) { xg = 2 * g.W - x0 - 4; xrev = true; }
        else { xg = 0; slow = true; }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const long long off = base + (long long)mirror_index(y0 + img_row(r), g.H) * g.P + xg;
        // (zero_in: u_in = v_in = one row of zeros, see StripGeom -- the loads stay unconditional: under a branch per
        // row the compiler waits for each row's two loads before it issues the next row's)
        const long long off_uv = g.zero_in ? (long long)xg : off;
        lc[r] = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (!DERIV) {
            lu[r] = *(const float4 *)(u_in + off_uv);
            lv[r] = *(const float4 *)(v_in + off_uv);
            lc[r] = *(const uint4 *)(coef + off);
        } else {
            lu[r] = lv[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if constexpr (DERIV) {
        // The launch with the derivative pass is nearly always the one that starts from zero flow: then it reads no flow at
        // all and goes straight to the frames (strip_derive); otherwise ONE branch around all the rows' loads.
        if (!g.zero_in) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const long long off = base + (long long)mirror_index(y0 + img_row(r), g.H) * g.P + xg;
                lu[r] = *(const float4 *)(u_in + off);
                lv[r] = *(const float4 *)(v_in + off);
            }
        }
    }
    // The lane's pixels p0..p3 go into the register pairs P = (p0, p3), Q = (p1, p2) (cross_rows); the packed derivative
    // word becomes the three coefficients of the update (sweep_coefs: one v_rsq per pixel).
    auto unpack_row = [&](const int r) __attribute__((always_inline)) {
        const float4 lu_ = lu[r], lv_ = lv[r];
        const uint4 cw = lc[r];
        constexpr int iPx = 0, iPy = 3, iQx = 1, iQy = 2;
        const float lu4[4] = {lu_.x, lu_.y, lu_.z, lu_.w}, lv4[4] = {lv_.x, lv_.y, lv_.z, lv_.w};
        uP[r] = f2{lu4[iPx], lu4[iPy]}; uQ[r] = f2{lu4[iQx], lu4[iQy]};
        vP[r] = f2{lv4[iPx], lv4[iPy]}; vQ[r] = f2{lv4[iQx], lv4[iQy]};
        float al[4], be[4], ga[4];
        const uint32_t cc[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
        for (int p = 0; p < 4; p++) sweep_coefs(cc[p], ilambda, al[p], be[p], ga[p]);
        cf[r].alP = f2{al[iPx], al[iPy]}; cf[r].alQ = f2{al[iQx], al[iQy]};
        cf[r].beP = f2{be[iPx], be[iPy]}; cf[r].beQ = f2{be[iQx], be[iQy]};
        cf[r].gaP = f2{ga[iPx], ga[iPy]} * (HS_SCALED ? 4.0f : 1.0f); cf[r].gaQ = f2{ga[iQx], ga[iQy]} * (HS_SCALED ? 4.0f : 1.0f);
    };
    // The rows are unpacked as they arrive, straight behind the loads with no branch in between (a third of the launch's
    // set-up is this arithmetic: behind the side tiles' fix-ups the compiler waits for ALL loads first).  A mirrored
    // group arrives reversed: (p0, p1, p2, p3) -> (p3, p2, p1, p0) is a swap of the halves of every register pair.
    if constexpr (!DERIV) {
#pragma unroll
        for (int r = 0; r < R; r++) unpack_row(r);
    }
    if (xedge) { // workgroup-uniform
#pragma unroll
        for (int r = 0; r < R; r++) {
            if constexpr (DERIV) { // (not unpacked yet; the derivative words come in lane order from strip_derive)
                if (xrev) {
                    lu[r] = make_float4(lu[r].w, lu[r].z, lu[r].y, lu[r].x);
                    lv[r] = make_float4(lv[r].w, lv[r].z, lv[r].y, lv[r].x);
                }
            } else {
                uP[r] = xrev ? f2_swap(uP[r]) : uP[r]; uQ[r] = xrev ? f2_swap(uQ[r]) : uQ[r];
                vP[r] = xrev ? f2_swap(vP[r]) : vP[r]; vQ[r] = xrev ? f2_swap(vQ[r]) : vQ[r];
                cf[r].alP = xrev ? f2_swap(cf[r].alP) : cf[r].alP; cf[r].alQ = xrev ? f2_swap(cf[r].alQ) : cf[r].alQ;
                cf[r].beP = xrev ? f2_swap(cf[r].beP) : cf[r].beP; cf[r].beQ = xrev ? f2_swap(cf[r].beQ) : cf[r].beQ;
                cf[r].gaP = xrev ? f2_swap(cf[r].gaP) : cf[r].gaP; cf[r].gaQ = xrev ? f2_swap(cf[r].gaQ) : cf[r].gaQ;
            }
        }
        if (__builtin_amdgcn_ballot_w64(slow) != 0) { // wave-uniform: rare (W % 4 != 0, or an image narrower than the halo)
            if (slow) {
                const int xa = mirror_index(x0, g.W), xb = mirror_index(x0 + 1, g.W),
                          xc = mirror_index(x0 + 2, g.W), xd = mirror_index(x0 + 3, g.W);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const long long row = base + (long long)mirror_index(y0 + img_row(r), g.H) * g.P;
                    {
                        const float *uv = u_in + (g.zero_in ? 0 : row), *vv = v_in + (g.zero_in ? 0 : row);
                        lu[r] = make_float4(uv[xa], uv[xb], uv[xc], uv[xd]);
                        lv[r] = make_float4(vv[xa], vv[xb], vv[xc], vv[xd]);
                    }
                    if constexpr (!DERIV) {
                        const uint32_t *cv = coef + row;
                        lc[r] = make_uint4(cv[xa], cv[xb], cv[xc], cv[xd]);
                        unpack_row(r);
                    }
                }
            }
        }
    }
    if constexpr (DERIV) {
        strip_derive<R>(fA, fB, g, base, x0, rev ? y0 + R - 1 : y0, rev ? -1 : 1, xin, lc);
#pragma unroll
        for (int r = 0; r < R; r++) unpack_row(r);
    }
    // core membership (for the store and for Eps): rows as a bit mask, lanes as a flag
    unsigned rowcore = 0;
    int rdist[R]; // distance of each row from the core rows (0 inside): wave-uniform
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int j = w * R + img_row(r), y = y0 + img_row(r);
        if (j >= g.T && j < g.T + g.CH && y >= 0 && y < g.H) rowcore |= 1u << r;
        rdist[r] = j < g.T ? g.T - j : (j >= g.T + g.CH ? j - (g.T + g.CH - 1) : 0);
    }
    const bool lanecore = (x0 >= 0) && (x0 < g.W) && (4 * lane >= g.HX) && (4 * lane < g.HX + g.CW);
    const int pr = g.W - 1 - x0; // image columns of this group: 0..min(pr,3)
    // From here on the Eps / witness kernels read `lane` from the hardware where it is used (hs_lane_now): nothing below
    // keeps it in a register, and with the three LDS addresses of a sweep formed from that one value the witness kernel
    // no longer spills (it reloaded a coefficient pair in every sweep: ITER|EPS stream 0.1325 -> 0.1254 ms per pair).  The
    // plain kernel has the registers and is 0.8 % faster with the lane number kept.
    const int lane_kept = lane;
    constexpr bool LANE_HW = EPS != 0 || PERSIST;
#define lane (LANE_HW ? hs_lane_now() : lane_kept)
    unsigned epscore = 0; // the core rows that lie in the Eps window (wave-uniform; all of them unless hsflow_set_eps_rows narrowed it)
    if (EPS != 0) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int y = y0 + img_row(r);
            if (((rowcore >> r) & 1u) && y >= g.ey0 && y < g.ey1) epscore |= 1u << r;
        }
    }
    if (DERIV && lanecore) { // the cores tile the image: this launch leaves the complete derivative plane behind
#pragma unroll
        for (int r = 0; r < R; r++)
            if ((rowcore >> r) & 1u) *(uint4 *)(coef_w + base + (long long)(y0 + img_row(r)) * g.P + x0) = lc[r];
    }

    // One row: neighbour sums from the cross sums of its two boundaries (sc below, sp above), update in place.
    // A row at distance d from the core is only needed through sweep T-1-d (trapezoid): later sweeps skip it
    // (wave-uniform branch), which trims the redundant halo work by about half.  PE = parity of the row's pixel p0.
#define HS_ACT(r) (rdist[r] <= last)
#define HS_ROW(r, PE, SC, SP)                                                                      \
    do {                                                                                           \
        if (HS_ACT(r)) {                                                                           \
            f2 ouP, ouQ, ovP, ovQ;                                                                 \
            if (EM == 1) { ouP = uP[r]; ouQ = uQ[r]; ovP = vP[r]; ovQ = vQ[r]; }                   \
            if (HS_DIAG & 8) uP[r] += SC.uP + SP.uP;                                               \
            else strip_row_update<PE>(uP[r], uQ[r], vP[r], vQ[r], SC, SP, cf[r]);                  \
            if (EM == 1) {                                                                         \
                if ((epscore >> (r)) & 1u) { /* wave-uniform; lanes outside the core are masked once per sweep */ \
                    const f2 dUP = HS_DIFF(ouP, uP[r]), dUQ = HS_DIFF(ouQ, uQ[r]), dVP = HS_DIFF(ovP, vP[r]), dVQ = HS_DIFF(ovQ, vQ[r]); \
                    if (!xedge) { /* workgroup-uniform: every column of the region is an image column */ \
                        e = fmaxf(fmaxf(e, fabsf(dUP.x)), fabsf(dUP.y));                           \
                        e = fmaxf(fmaxf(e, fabsf(dUQ.x)), fabsf(dUQ.y));                           \
                        e = fmaxf(fmaxf(e, fabsf(dVP.x)), fabsf(dVP.y));                           \
                        e = fmaxf(fmaxf(e, fabsf(dVQ.x)), fabsf(dVQ.y));                           \
                    } else {                                                                       \
                        /* image columns of the group are p0 .. p(pr); P = (p0, p3), Q = (p1, p2) */ \
                        e = fmaxf(e, fmaxf(fabsf(dUP.x), fabsf(dVP.x)));                           \
                        if (pr >= 1) e = fmaxf(e, fmaxf(fabsf(dUQ.x), fabsf(dVQ.x)));              \
                        if (pr >= 2) e = fmaxf(e, fmaxf(fabsf(dUQ.y), fabsf(dVQ.y)));              \
                        if (pr >= 3) e = fmaxf(e, fmaxf(fabsf(dUP.y), fabsf(dVP.y)));              \
                    }                                                                              \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
    } while (0)
    // cross sums of the boundary below register row A (row B underneath), needed while either row is still swept
#define HS_CROSS(S, PE, A, B)                                                                      \
    do {                                                                                           \
        if (HS_ACT(A) || HS_ACT(B)) {                                                              \
            if (HS_DIAG & 8) S.uP = uP[A] + uP[B];                                                 \
            else cross_rows<PE>(S, uP[A], uQ[A], vP[A], vQ[A], uP[B], uQ[B], vP[B], vQ[B]);        \
        }                                                                                          \
    } while (0)
#define HS_PUBLISH(buf)                                                                            \
    do {                                                                                           \
        float4 *exw = ex + ((size_t)((buf) * NW + w) * 4) * 64 + HS_LANE;                          \
        exw[0] = make_float4(uP[0].x, uP[0].y, uQ[0].x, uQ[0].y);                                  \
        exw[64] = make_float4(vP[0].x, vP[0].y, vQ[0].x, vQ[0].y);                                 \
        exw[128] = make_float4(uP[R - 1].x, uP[R - 1].y, uQ[R - 1].x, uQ[R - 1].y);                \
        exw[192] = make_float4(vP[R - 1].x, vP[R - 1].y, vQ[R - 1].x, vQ[R - 1].y);                \
    } while (0)

    // Exchange slots: ex[buf][wave][0..3][lane] = {register row 0: u, v; register row R-1: u, v}.
    // Sweep s reads buffer s&1 and publishes its new edge rows into buffer (s+1)&1, then meets the
    // other wavefronts at ONE barrier.  The edge rows are updated and published FIRST so that the
    // LDS writes drain while the interior rows are being computed.
    if (stamps) st1 = __builtin_amdgcn_s_memtime();
    // Whose slot is the row above register row 0 (wu, su) and the row below register row R-1 (wd, sd)?  A strip's
    // bottom image row is its register row R-1 (slot pair 2) unless it is reversed (then row 0: slot pair 0), its
    // top image row the other one; for odd R the neighbouring strips are reversed exactly when this one is not.
    // A reversed strip has the strip BELOW it above its register row 0.
    // (at the region edge the strip's own edge row stands in: junk the validity argument tolerates)
    const bool nrev = (R & 1) != 0 && !rev; // the strips above and below are reversed
    const int wa = w > 0 ? w - 1 : w, sa = w > 0 ? (nrev ? 0 : 2) : 0;            // image row above this strip
    const int wb = w < NW - 1 ? w + 1 : w, sb = w < NW - 1 ? (nrev ? 2 : 0) : 2;  // image row below this strip
    const int wu = rev ? wb : wa, su = rev ? sb : sa;
    const int wd = rev ? wa : wb, sd = rev ? sa : sb;
    int seen_n = 0; // EPS == 2, wave-uniform: sweeps so far that had a change >= eps_thr (a counter: a
                    // loop-carried flag makes the register allocator spill inside the loop)
    const unsigned long long wit_mask = (epscore & 1u) ? __builtin_amdgcn_ballot_w64(lanecore) : 0ull; // lanes whose answer counts
    // Who looks, and when: the wavefronts whose rows are all core rows take turns, one of them per sweep (a sweep
    // needs ONE witness in the workgroup; sixteen wavefronts looking in every sweep cost 4 % of the launch).  A
    // workgroup without such a wavefront (thin cores, clipped bottom tiles) keeps the old rule: every wavefront whose
    // register row 0 is a core row looks in every sweep.  wit_cnt counts down to this wavefront's next turn.
    const int core_here = min(g.CH, g.H - by * g.CH);
    const int wit_lo = (g.T + R - 1) / R, wit_n = (g.T + core_here) / R - wit_lo; // wavefronts wit_lo .. wit_lo + wit_n - 1
    const bool wit_rot = wit_n > 0;
    const int wit_per = wit_rot ? wit_n : 1;
    const int wit_cnt0 = !wit_rot ? 0 : (w >= wit_lo && w < wit_lo + wit_n) ? w - wit_lo : (1 << 30);
    int wit_cnt = wit_cnt0;
    // Sweeps of the current phase: the launch's T, except in the last phase of a persistent launch (PERSIST).  A shorter
    // phase simply starts further down the trapezoid: rows further than Tp - 1 from the core are never swept.
    int Tp = g.T;
    // One sweep.  EM is the Eps mode of THIS sweep: the launch's own (EPS 0, 1, 2), or for EPS == 3 witness
    // (2) in all sweeps but the last and measured (1) in the last -- a second copy of the sweep code after the
    // loop, so that the loop keeps the registers of the witness kernel.  E0 = parity of pixel p0 of register
    // row 0 (image row y0 + image column x0, x0 % 4 == 0): wave-uniform, rows alternate from there.
    // (A second, straight-line copy of the sweep for wavefronts whose rows are all core rows -- no compare and branch per
    // row -- was tried and rejected: two copies of the body make the register allocator spill 66 registers.  So was a
    // trapezoid per WAVEFRONT instead of per row, one branch per sweep: 16 -> 3 branches and 40 -> 12 scalar instructions
    // in the loop, but the straight-line body keeps more values alive and the ITER kernel then spills inside the loop:
    // ITER|EPS 0.1573 -> 0.1556 ms, ITER 0.1522 -> 0.1555 ms at 1080p / 100.)
    auto sweep = [&](const int s, auto em_tag) __attribute__((always_inline)) {
        constexpr int EM = decltype(em_tag)::value;
#if HS_DIAG & 4 /* diagnostic build: every row swept in every sweep (no trapezoid) */
        const int last = 1 << 20;
#else
        const int last = Tp - 1 - s; // rows with rdist <= last are still swept
#endif
        // HS_SCALED: this sweep takes the flow from scale 4^s to 4^(s+1)
        const float unscale = HS_SCALED ? __builtin_ldexpf(1.0f, -2 * (s + 1)) : 1.0f;
        // eps_thr * 4^(s+1) by integer arithmetic on the exponent: scalar instructions only (v_ldexp + v_readfirstlane put
        // a vector-to-scalar round trip into every sweep).  Exact for a normal eps_thr; eps_thr = 0 or a product beyond
        // the float range merely makes the witness fail, and the exact pass decides (the host never sends a denormal).
        const float thr_s = HS_SCALED ? __int_as_float(__float_as_int(eps_thr) + ((s + 1) << 24)) : eps_thr;
#if HS_DIAG & 1 /* diagnostic build (wrong results): no LDS traffic, the strip's own edge rows stand in */
        const float4 hu4 = make_float4(uP[0].x, uP[0].y, uQ[0].x, uQ[0].y), hv4 = make_float4(vP[0].x, vP[0].y, vQ[0].x, vQ[0].y);
        const float4 du4 = make_float4(uP[R - 1].x, uP[R - 1].y, uQ[R - 1].x, uQ[R - 1].y), dv4 = make_float4(vP[R - 1].x, vP[R - 1].y, vQ[R - 1].x, vQ[R - 1].y);
#else
        // (Eps / witness kernels: the three LDS addresses of a sweep are formed from ONE value, the lane number as the
        // hardware gives it at this point (hs_lane_now): left alone the compiler keeps three loop-invariant address
        // registers alive across the sweep loop, which has none to spare -- the witness kernel then reloaded a spilled
        // coefficient pair in every sweep.)
        const int lane_l = lane;
#define HS_LANE lane_l
        const float4 *eu = ex + ((size_t)((s & 1) * NW + wu) * 4 + su) * 64 + lane_l;
        const float4 *ed = ex + ((size_t)((s & 1) * NW + wd) * 4 + sd) * 64 + lane_l;
        // (Idle strips -- late sweeps, halo wavefronts -- still read and publish their edge rows.  Gating the reads makes the
        // register allocator spill inside the loop; gating only the publish costs two more branches per sweep than the
        // LDS traffic it saves: 0.1363 against 0.1353 ms per pair, round 3.)
        const float4 hu4 = eu[0], hv4 = eu[64]; // old row above the strip
        const float4 du4 = ed[0], dv4 = ed[64]; // old row below the strip
#endif
        const f2 huP = f2{hu4.x, hu4.y}, huQ = f2{hu4.z, hu4.w}, hvP = f2{hv4.x, hv4.y}, hvQ = f2{hv4.z, hv4.w};
        const f2 duP = f2{du4.x, du4.y}, duQ = f2{du4.z, du4.w}, dvP = f2{dv4.x, dv4.y}, dvQ = f2{dv4.z, dv4.w};

        float e = 0.f;
        constexpr int EL = E0 ^ ((R - 1) & 1); // parity of the last register row
        Cross sA, s0, sK, sL; // above row 0, below row 0, above row R-1, below row R-1
        const bool wturn = EM == 2 && wit_cnt == 0; // wave-uniform: this wavefront is the sweep's witness
        float w0 = 0.f;
        if (wturn) w0 = uP[0].x; // witness: u at column x0 of register row 0 before the sweep
        // --- first row (the strip's upper edge)
        if (HS_ACT(0)) cross_rows<E0 ^ 1>(sA, huP, huQ, hvP, hvQ, uP[0], uQ[0], vP[0], vQ[0]);
        if (R == 1) {
            if (HS_ACT(0)) cross_rows<E0>(s0, uP[0], uQ[0], vP[0], vQ[0], duP, duQ, dvP, dvQ);
        } else {
            constexpr int R1 = R > 1 ? 1 : 0;
            HS_CROSS(s0, E0, 0, R1);
        }
        HS_ROW(0, E0, s0, sA);
        // witness: did u change by >= eps_thr at column x0 of register row 0 (where that is a core row) in any lane?
        // (The host runs witness launches only with plans in which some wavefront has such a row: strip_has_witness.)
        if (EM == 2) { // (wit_mask: the core lanes if register row 0 is a core row, else none)
            if (wturn) seen_n += (__builtin_amdgcn_ballot_w64(fabsf(HS_DIFF1(w0, uP[0].x)) >= thr_s) & wit_mask) != 0 ? 1 : 0;
            wit_cnt = wturn ? wit_per - 1 : wit_cnt - 1;
        }
        if (R >= 2) {
            // --- last row (the lower edge), then both edges go to the other wavefronts
            constexpr int RM = R > 2 ? R - 2 : 0;
            if (HS_ACT(R - 1)) cross_rows<EL>(sL, uP[R - 1], uQ[R - 1], vP[R - 1], vQ[R - 1], duP, duQ, dvP, dvQ);
                if (R >= 3) HS_CROSS(sK, EL ^ 1, RM, R - 1);
            if (R == 2) HS_ROW(R - 1, EL, sL, s0);
            else HS_ROW(R - 1, EL, sL, sK);
        }
#if !(HS_DIAG & 1)
        if (s + 1 < Tp) HS_PUBLISH((s + 1) & 1);
#endif
        // --- interior rows, top to bottom: each needs the cross sum above it (kept) and the one below (new)
        if (R >= 3) {
            Cross sp = s0;
#pragma unroll
            for (int r = 1; r < R - 1; r++) {
                Cross sc;
                if (r == R - 2) sc = sK;
                else {
                    const int rn = r + 1 < R ? r + 1 : r;
                    if (r & 1) HS_CROSS(sc, E0 ^ 1, r, rn);
                    else HS_CROSS(sc, E0, r, rn);
                }
                if (r & 1) HS_ROW(r, E0 ^ 1, sc, sp);
                else HS_ROW(r, E0, sc, sp);
                sp = sc;
            }
        }
        if (EM == 1 && EPS == 3) { // the one measured sweep of a witness launch: folded after the loop
            e = wave_max_nonneg(lanecore ? e : 0.f) * unscale;
            if (lane == 0) eps_lds[16 + w] = e;
        }
        if (EM == 1 && EPS == 1) { // per-wavefront maximum -> LDS; wavefront 0 folds the previous sweep's 16 values
            e = wave_max_nonneg(lanecore ? e : 0.f) * unscale;
            if (lane == 0) eps_lds[(s & 1) * 16 + w] = e;
            if (s > 0 && w == 0) {
                float x = lane < NW ? eps_lds[((s - 1) & 1) * 16 + lane] : 0.f;
                x = wave_max_nonneg(x);
                if (lane == 0) eps_out[(size_t)(s - 1) * eps_stride + blockIdx.x] = __float_as_uint(x);
            }
        }
#if !(HS_DIAG & 2) /* diagnostic build: no barrier */
        if (s + 1 < Tp) __syncthreads();
#endif
        // diagnostic (stamps != NULL only): when each sweep ended -- all wavefronts leave the barrier together, so
        // wavefront 0 sees the workgroup's sweep times; slots behind the 8 phase stamps of every workgroup
#if HS_SWEEP_STAMPS /* (diagnostic builds: tools/diag_build.sh; tools/stamps_sweeps.py reads them) */
        if (stamps && gridDim.x <= 8192 && threadIdx.x == 0)
            stamps[(size_t)gridDim.x * 8 + (size_t)blockIdx.x * 32 + (s & 31)] = __builtin_amdgcn_s_memtime();
#endif
    };
#if HS_CORE_PRIO
    // Wavefronts whose rows are all core rows never drop out of a sweep (trapezoid): they are the ones every barrier waits
    // for, so they go first at the issue port.
    if (rowcore == (1u << R) - 1u) __builtin_amdgcn_s_setprio(HS_CORE_PRIO);
#endif
    // PERSIST: the phases of the solve; otherwise one pass (the `break` after the store is unconditional).
    int ph = 0;
    unsigned flag_base = 0; // this tile's phase counter at the start of the launch (all tiles agree; wavefront 0 only)
    unsigned long long pt_sweep = 0, pt_pub = 0, pt_wait = 0, pt_load = 0, pt0 = 0, pt_first = 0; // PERSIST diagnostics (stamps)
    if constexpr (PERSIST) {
        if (w == 0) flag_base = (unsigned)__builtin_amdgcn_readfirstlane((int)load_u32_sc1(pa.flags + tile));
    }
#pragma unroll 1
    for (;;) {
    if constexpr (PERSIST) {
        Tp = ph == pa.n_phase - 1 ? pa.T_last : g.T;
        seen_n = 0;
        wit_cnt = wit_cnt0;
        if (stamps && ph == 0) pt_first = pt0 = __builtin_amdgcn_s_memtime();
    }
#undef HS_LANE
#define HS_LANE lane
    HS_PUBLISH(0);
    __syncthreads();
    if (PERSIST && stamps && ph > 0) { // (the reloaded rows are first used by the publish above: their latency belongs to the reload)
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        pt_load += t - pt0;
        pt0 = t;
    }
    if constexpr (EPS == 3) {
#pragma unroll 1
        for (int s = 0; s + 1 < Tp; s++) sweep(s, std::integral_constant<int, 2>{});
        sweep(Tp - 1, std::integral_constant<int, 1>{});
    } else if constexpr (HS_PEEL_LAST_STRIP && EPS != 1) {
#pragma unroll 1
        for (int s = 0; s + 1 < Tp; s++) sweep(s, std::integral_constant<int, EPS>{});
        sweep(Tp - 1, std::integral_constant<int, EPS>{});
    } else {
#pragma unroll 1
        for (int s = 0; s < Tp; s++) sweep(s, std::integral_constant<int, EPS>{});
    }
    unsigned *const eps_ph = PERSIST ? eps_out + (size_t)ph * eps_stride : eps_out; // PERSIST: one row of words per phase
    if (EPS == 1) {
        __syncthreads();
        if (w == 0) {
            float x = lane < NW ? eps_lds[((Tp - 1) & 1) * 16 + lane] : 0.f;
            x = wave_max_nonneg(x);
            if (lane == 0) eps_out[(size_t)(Tp - 1) * eps_stride + blockIdx.x] = __float_as_uint(x);
        }
    }
    if (EPS == 2 || EPS == 3) {
        const int witnessed = EPS == 3 ? Tp - 1 : Tp; // sweeps that ran in witness mode
        // sweeps this wavefront vouches for; taking turns they add up to the launch, otherwise one wavefront must have all
        if (lane == 0) eps_lds[w] = (float)seen_n;
        __syncthreads();
        if (w == 0) {
            const float x = lane < NW ? eps_lds[lane] : 0.f;
            const float n = wit_rot ? wave_sum16(x) : wave_max_nonneg(x);
            if (lane == 0) eps_ph[blockIdx.x] = __float_as_uint(n == (float)witnessed ? __builtin_inff() : 0.f);
            if (EPS == 3) { // second word: Eps of the last sweep, exact
                const float x = wave_max_nonneg(lane < NW ? eps_lds[16 + lane] : 0.f);
                if (lane == 0) eps_out[(size_t)eps_stride + blockIdx.x] = __float_as_uint(x);
            }
        }
    }
    if (stamps) st2 = __builtin_amdgcn_s_memtime();

    if constexpr (!PERSIST) {
        // (the lane's column is worked out afresh from a copy of the lane number the optimiser cannot see through: kept
        // alive across the sweep loop it is one of the three registers the loop has no room for -- every spilled register
        // is 250 KB of scratch written and read per launch, which showed up as HBM writes: profiles/r03_traffic_by_kernel.json)
        int lane_s = lane;
        if constexpr (!LANE_HW) asm volatile("" : "+v"(lane_s)); // (the kept lane number: a copy the optimiser cannot see through)
        const int x0s = bx * g.CW - g.HX + 4 * lane_s;
        if (lanecore) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                if ((rowcore >> r) & 1u) {
                    const long long off = base + (long long)(y0 + img_row(r)) * g.P + x0s;
                    // (non-temporal and agent-scope write-through stores were tried here: both slower)
                    const float fin = HS_SCALED ? __builtin_ldexpf(1.0f, -2 * g.T) : 1.0f; // back to scale 1 (exact)
                    // P = (p0, p3), Q = (p1, p2)
                    *(float4 *)(u_out + off) = make_float4(uP[r].x * fin, uQ[r].x * fin, uQ[r].y * fin, uP[r].y * fin);
                    *(float4 *)(v_out + off) = make_float4(vP[r].x * fin, vQ[r].x * fin, vQ[r].y * fin, vP[r].y * fin);
                }
            }
        }
        break;
    } else {
        if (stamps) { pt_sweep += st2 - pt0; pt0 = st2; }
        // --- publish: the core rows go back to scale 1 in their registers (exact) and out with write-through stores
        const __amdgpu_buffer_rsrc_t pu = plane_rsrc(pa.ub[ph & 1]), pv = plane_rsrc(pa.vb[ph & 1]);
        const float fin = HS_SCALED ? __builtin_ldexpf(1.0f, -2 * Tp) : 1.0f;
        // The lane's column and where it reads its halo from are worked out afresh in every phase, from a copy of the
        // lane number the optimiser cannot see through: hoisted out of the phase loop they would be two more registers
        // alive across the sweep loop, which has none to spare (the sweep then reloads spilled coefficients).
        const int lane_x = lane;
        const int x0x = bx * g.CW - g.HX + 4 * lane_x;
        int xgx = x0x;
        bool xrevx = false;
        if (xedge && !((x0x >= 0) && (x0x + 3 < g.W))) { // (host: W % 4 == 0 and W >= 256, so no group straddles the border)
            xgx = x0x < 0 ? -x0x - 4 : 2 * g.W - x0x - 4;
            xrevx = true;
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            if ((rowcore >> r) & 1u) { // wave-uniform
                uP[r] *= fin; uQ[r] *= fin; vP[r] *= fin; vQ[r] *= fin;
                if (lanecore) {
                    const unsigned row = 4u * (unsigned)(base + (long long)(y0 + img_row(r)) * g.P); // wave-uniform
                    store_f4_sc1(pu, row, 4u * (unsigned)x0x, f4{uP[r].x, uQ[r].x, uQ[r].y, uP[r].y});
                    store_f4_sc1(pv, row, 4u * (unsigned)x0x, f4{vP[r].x, vQ[r].x, vQ[r].y, vP[r].y});
                }
            }
        }
        if (ph == pa.n_phase - 1) break; // the launch's end publishes the last phase
        // the constant term back to the scale a phase starts with: a row was swept max(0, Tp - rdist) times
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int nact = Tp - rdist[r] > 0 ? Tp - rdist[r] : 0;
            const float back = HS_SCALED ? __builtin_ldexpf(1.0f, -2 * nact) : 1.0f;
            cf[r].gaP *= back; cf[r].gaQ *= back;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wavefront's stores have been written through
        __syncthreads();                                  // ... and so have every other wavefront's
        if (stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); pt_pub += t - pt0; pt0 = t; }
        int *const dead_lds = (int *)(eps_lds + 31);
        if (w == 0) {
            const unsigned target = flag_base + (unsigned)ph + 1u; // "phase ph of this tile is published"
            if (lane == 0) __hip_atomic_store(pa.flags + tile, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // lanes 0..7 watch the 8 neighbouring tiles (same pair), lane 8 the abort word
            int nb = -1;
            if (lane < 8) {
                const int k = lane < 4 ? lane : lane + 1; // 0..8 without the centre
                const int nx = bx + k % 3 - 1, ny = by + k / 3 - 1;
                if (nx >= 0 && nx < g.tiles_x && ny >= 0 && ny < g.tiles_y) nb = pair * tpp + ny * g.tiles_x + nx;
            } else if (lane == 8) nb = (int)gridDim.x;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            int dead = 0;
            for (;;) {
                unsigned v = target;
                if (nb >= 0) v = load_u32_sc1(pa.flags + nb);
                const bool aborted = lane == 8 && v != 0u;
                const bool behind = lane < 8 && (int)(v - target) < 0;
                if (__builtin_amdgcn_ballot_w64(aborted) != 0) { dead = 1; break; }
                if (__builtin_amdgcn_ballot_w64(behind) == 0) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)pa.wait_ticks) { // a neighbour never came
                    dead = 1;
                    if (lane == 0) {
                        __hip_atomic_store(pa.flags + gridDim.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(pa.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            if (lane == 0) *dead_lds = dead;
        }
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(*dead_lds)) return; // workgroup-uniform (and said so: a divergent exit would put the loop's counters into VGPRs)
        if (stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); pt_wait += t - pt0; pt0 = t; }
        // --- reload the halo: whole rows outside the core rows, the side lanes of core rows (host: W % 4 == 0 and
        // the image at least as large as the region, so every group is an aligned group, read backwards where mirrored)
        {
            f4 tu[R], tv[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const unsigned row = 4u * (unsigned)(base + (long long)mirror_index(y0 + img_row(r), g.H) * g.P); // wave-uniform
                tu[r] = f4{uP[r].x, uQ[r].x, uQ[r].y, uP[r].y};
                tv[r] = f4{vP[r].x, vQ[r].x, vQ[r].y, vP[r].y};
                if (!((rowcore >> r) & 1u) || !lanecore) {
                    tu[r] = load_f4_sc1(pu, row, 4u * (unsigned)xgx);
                    tv[r] = load_f4_sc1(pv, row, 4u * (unsigned)xgx);
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (xrevx) { // (mirrored groups lie outside the image: never core lanes)
                    tu[r] = f4{tu[r].w, tu[r].z, tu[r].y, tu[r].x};
                    tv[r] = f4{tv[r].w, tv[r].z, tv[r].y, tv[r].x};
                }
                uP[r] = f2{tu[r].x, tu[r].w}; uQ[r] = f2{tu[r].y, tu[r].z};
                vP[r] = f2{tv[r].x, tv[r].w}; vQ[r] = f2{tv[r].y, tv[r].z};
            }
        }
        ph++;
    }
    } // phases
#undef HS_ROW
#undef HS_ACT
#undef HS_CROSS
#undef HS_PUBLISH
#undef HS_LANE
#undef lane
    if (stamps && threadIdx.x == 0) {
        __builtin_amdgcn_s_waitcnt(0); // stores issued and acknowledged
        unsigned long long *o = stamps + (size_t)blockIdx.x * 8;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memtime();
        o[4] = sr0; o[5] = __builtin_amdgcn_s_memrealtime();
        o[6] = (unsigned long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20); // XCC_ID
        o[7] = (unsigned long long)tile;
        if (PERSIST && gridDim.x <= 8192) { // totals over the phases: sweeps, publish, wait, reload (cycles)
            unsigned long long *q = stamps + (size_t)gridDim.x * 8 + (size_t)blockIdx.x * 32;
            q[0] = pt_sweep; q[1] = pt_pub; q[2] = pt_wait; q[3] = pt_load; q[4] = pt_first - st1; q[5] = o[3] - st2;
        }
    }
}

template <int R, int NTMAX, int EPS, int E0>
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
    strip_body<R, NTMAX, EPS, E0, false>(coef, u_in, v_in, u_out, v_out, g, ilambda, eps_out, eps_stride, stamps, eps_thr,
                                     nullptr, nullptr, nullptr);
}

// First launch of a solve with the derivative pass folded in: reads the two frames instead of the packed
// derivative plane and writes that plane for the launches that follow (and for hsflow_get_derivatives).
template <int R, int NTMAX, int EPS, int E0>
__global__ __launch_bounds__(NTMAX) void k_jacobi_strip_deriv(const uint8_t *__restrict__ fA,
                                                              const uint8_t *__restrict__ fB,
                                                              uint32_t *__restrict__ coef_w,
                                                              const float *__restrict__ u_in,
                                                              const float *__restrict__ v_in,
                                                              float *__restrict__ u_out,
                                                              float *__restrict__ v_out, const StripGeom g,
                                                              const float ilambda,
                                                              unsigned *__restrict__ eps_out, const int eps_stride,
                                                              unsigned long long *__restrict__ stamps,
                                                              const float eps_thr)
{
    strip_body<R, NTMAX, EPS, E0, true>(nullptr, u_in, v_in, u_out, v_out, g, ilambda, eps_out, eps_stride, stamps, eps_thr,
                                    fA, fB, coef_w);
}

// One launch for the whole solve (PersistArgs above).  DERIV: phase 0 computes the derivative words from the frames and
// leaves the packed plane behind (coef_rw is written); otherwise coef_rw is only read.
template <int R, int NTMAX, int EPS, int E0, bool DERIV>
__global__ __launch_bounds__(NTMAX) void k_jacobi_strip_persist(const uint8_t *__restrict__ fA,
                                                                const uint8_t *__restrict__ fB,
                                                                uint32_t *coef_rw,
                                                                const float *__restrict__ u_in,
                                                                const float *__restrict__ v_in, const StripGeom g,
                                                                const float ilambda,
                                                                unsigned *__restrict__ eps_out, const int eps_stride,
                                                                unsigned long long *__restrict__ stamps,
                                                                const float eps_thr, const PersistArgs pa)
{
    strip_body<R, NTMAX, EPS, E0, DERIV, true>(coef_rw, u_in, v_in, nullptr, nullptr, g, ilambda, eps_out, eps_stride, stamps, eps_thr,
                                               fA, fB, coef_rw, pa);
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

// E0 = parity of pixel p0 of register row 0 of BOTH halves: yb has the parity of T (CH is even) and the lower half's
// reversed order turns its rows' parities around (k_jacobi_strip explains the reversal), so one code path serves both.
template <int R, int NTMAX, int EPS, int E0, bool DERIV> // EPS: 0 none, 1 Eps of every sweep, 2 witness (see k_jacobi_strip)
__device__ __forceinline__ void fold_body(const uint32_t *__restrict__ coef,
                                                       const float *__restrict__ u_in,
                                                       const float *__restrict__ v_in,
                                                       float *__restrict__ u_out,
                                                       float *__restrict__ v_out, const StripGeom g,
                                                       const float ilambda,
                                                       unsigned *__restrict__ eps_out, const int eps_stride,
                                                       unsigned long long *__restrict__ stamps,
                                                       const float eps_thr, const uint8_t *__restrict__ fA,
                                                       const uint8_t *__restrict__ fB, uint32_t *__restrict__ coef_w)
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
    // all rows' loads first, branch-free; then the fix-ups of the side tiles (k_jacobi_strip explains why)
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int y = yb + (lower ? 2 * R - 1 - r : r);
        const long long off = base + (long long)mirror_index(y, g.H) * g.P + xg;
        const long long off_uv = g.zero_in ? (long long)xg : off; // (zero_in: one row of zeros, k_jacobi_strip)
        lc[r] = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (!DERIV) {
            lu[r] = *(const float4 *)(u_in + off_uv);
            lv[r] = *(const float4 *)(v_in + off_uv);
            lc[r] = *(const uint4 *)(coef + off);
        } else {
            lu[r] = lv[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if constexpr (DERIV) { // (k_jacobi_strip: no flow loads for a solve from zero flow, one branch around them otherwise)
        if (!g.zero_in) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int y = yb + (lower ? 2 * R - 1 - r : r);
                const long long off = base + (long long)mirror_index(y, g.H) * g.P + xg;
                lu[r] = *(const float4 *)(u_in + off);
                lv[r] = *(const float4 *)(v_in + off);
            }
        }
    }
    // the lane's pixels p0..p3 go into the register pairs P = (p0, p3), Q = (p1, p2) (cross_rows); rows are unpacked as they
    // arrive, before the side tiles' fix-ups (k_jacobi_strip: a reversed group = the halves of every pair swapped)
    auto unpack_row = [&](const int r) __attribute__((always_inline)) {
        constexpr int iPx = 0, iPy = 3, iQx = 1, iQy = 2;
        const float lu4[4] = {lu[r].x, lu[r].y, lu[r].z, lu[r].w}, lv4[4] = {lv[r].x, lv[r].y, lv[r].z, lv[r].w};
        uP[r] = f2{lu4[iPx], lu4[iPy]}; uQ[r] = f2{lu4[iQx], lu4[iQy]};
        vP[r] = f2{lv4[iPx], lv4[iPy]}; vQ[r] = f2{lv4[iQx], lv4[iQy]};
        float al[4], be[4], ga[4];
        const uint32_t cc[4] = {lc[r].x, lc[r].y, lc[r].z, lc[r].w};
#pragma unroll
        for (int p = 0; p < 4; p++) sweep_coefs(cc[p], ilambda, al[p], be[p], ga[p]);
        cf[r].alP = f2{al[iPx], al[iPy]}; cf[r].alQ = f2{al[iQx], al[iQy]};
        cf[r].beP = f2{be[iPx], be[iPy]}; cf[r].beQ = f2{be[iQx], be[iQy]};
        cf[r].gaP = f2{ga[iPx], ga[iPy]} * (HS_SCALED ? 4.0f : 1.0f); cf[r].gaQ = f2{ga[iQx], ga[iQy]} * (HS_SCALED ? 4.0f : 1.0f);
    };
    if constexpr (!DERIV) {
#pragma unroll
        for (int r = 0; r < R; r++) unpack_row(r);
    }
    if (side) { // workgroup-uniform
#pragma unroll
        for (int r = 0; r < R; r++) {
            if constexpr (DERIV) {
                if (rev) {
                    lu[r] = make_float4(lu[r].w, lu[r].z, lu[r].y, lu[r].x);
                    lv[r] = make_float4(lv[r].w, lv[r].z, lv[r].y, lv[r].x);
                }
            } else {
                uP[r] = rev ? f2_swap(uP[r]) : uP[r]; uQ[r] = rev ? f2_swap(uQ[r]) : uQ[r];
                vP[r] = rev ? f2_swap(vP[r]) : vP[r]; vQ[r] = rev ? f2_swap(vQ[r]) : vQ[r];
                cf[r].alP = rev ? f2_swap(cf[r].alP) : cf[r].alP; cf[r].alQ = rev ? f2_swap(cf[r].alQ) : cf[r].alQ;
                cf[r].beP = rev ? f2_swap(cf[r].beP) : cf[r].beP; cf[r].beQ = rev ? f2_swap(cf[r].beQ) : cf[r].beQ;
                cf[r].gaP = rev ? f2_swap(cf[r].gaP) : cf[r].gaP; cf[r].gaQ = rev ? f2_swap(cf[r].gaQ) : cf[r].gaQ;
            }
        }
        if (__builtin_amdgcn_ballot_w64(slow) != 0) { // wave-uniform, rare
            if (slow) {
                const int xa = mirror_index(x0, g.W), xb = mirror_index(x0 + 1, g.W),
                          xc = mirror_index(x0 + 2, g.W), xd = mirror_index(x0 + 3, g.W);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int y = yb + (lower ? 2 * R - 1 - r : r);
                    const long long row = base + (long long)mirror_index(y, g.H) * g.P;
                    {
                        const float *uv = u_in + (g.zero_in ? 0 : row), *vv = v_in + (g.zero_in ? 0 : row);
                        lu[r] = make_float4(uv[xa], uv[xb], uv[xc], uv[xd]);
                        lv[r] = make_float4(vv[xa], vv[xb], vv[xc], vv[xd]);
                    }
                    if constexpr (!DERIV) {
                        const uint32_t *cv = coef + row;
                        lc[r] = make_uint4(cv[xa], cv[xb], cv[xc], cv[xd]);
                        unpack_row(r);
                    }
                }
            }
        }
    }
    // (the wave shifts in strip_derive cross the lane 31/32 seam like those of the sweep: region-edge columns)
    if constexpr (DERIV) {
        strip_derive<R>(fA, fB, g, base, x0, lower ? yb + 2 * R - 1 : yb, lower ? -1 : 1, xin, lc);
#pragma unroll
        for (int r = 0; r < R; r++) unpack_row(r);
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
    if (DERIV && lanecore) { // the cores tile the image: this launch leaves the complete derivative plane behind
#pragma unroll
        for (int r = 0; r < R; r++)
            if ((rowcore >> r) & 1u)
                *(uint4 *)(coef_w + base + (long long)(yb + (lower ? 2 * R - 1 - r : r)) * g.P + x0) = lc[r];
    }

#define HF_ROW(r, PE, SC, SP)                                                                      \
    do {                                                                                           \
        if (rdist[r] <= last) {                                                                    \
            f2 ouP, ouQ, ovP, ovQ;                                                                 \
            if (EM == 1) { ouP = uP[r]; ouQ = uQ[r]; ovP = vP[r]; ovQ = vQ[r]; }                   \
            strip_row_update<PE>(uP[r], uQ[r], vP[r], vQ[r], SC, SP, cf[r]);                       \
            if (EM == 1) {                                                                         \
                if (((rowcore >> (r)) & 1u) && lanecore) {                                         \
                    /* image columns of the group are p0 .. p(pr); P = (p0, p3), Q = (p1, p2) */   \
                    e = fmaxf(e, fmaxf(fabsf(HS_DIFF1(ouP.x, uP[r].x)), fabsf(HS_DIFF1(ovP.x, vP[r].x))));           \
                    if (pr >= 1) e = fmaxf(e, fmaxf(fabsf(HS_DIFF1(ouQ.x, uQ[r].x)), fabsf(HS_DIFF1(ovQ.x, vQ[r].x)))); \
                    if (pr >= 2) e = fmaxf(e, fmaxf(fabsf(HS_DIFF1(ouQ.y, uQ[r].y)), fabsf(HS_DIFF1(ovQ.y, vQ[r].y)))); \
                    if (pr >= 3) e = fmaxf(e, fmaxf(fabsf(HS_DIFF1(ouP.y, uP[r].y)), fabsf(HS_DIFF1(ovP.y, vP[r].y)))); \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#define HF_CROSS(S, PE, A, B)                                                                      \
    do {                                                                                           \
        if (rdist[A] <= last || rdist[B] <= last)                                                  \
            cross_rows<PE>(S, uP[A], uQ[A], vP[A], vQ[A], uP[B], uQ[B], vP[B], vQ[B]);             \
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
    const unsigned long long wit_mask = __builtin_amdgcn_ballot_w64((rowcore & 1u) && lanecore); // lanes whose answer counts
    // one sweep in Eps mode EM (k_jacobi_strip explains EPS == 3: witness sweeps, then one measured sweep)
    auto sweep = [&](const int s, auto em_tag) __attribute__((always_inline)) {
        constexpr int EM = decltype(em_tag)::value;
        const int last = g.T - 1 - s; // rows with rdist <= last are still swept
        const float unscale = HS_SCALED ? __builtin_ldexpf(1.0f, -2 * (s + 1)) : 1.0f;
        const float thr_s = HS_SCALED ? __int_as_float(__float_as_int(eps_thr) + ((s + 1) << 24)) : eps_thr; // (see k_jacobi_strip)
        const float4 *eo = HF_SLOT(s & 1, wo, ho) + hl;
        const float4 h4u = eo[0], h4v = eo[32];
        const f2 ouP_ = f2{h4u.x, h4u.y}, ouQ_ = f2{h4u.z, h4u.w}, ovP_ = f2{h4v.x, h4v.y}, ovQ_ = f2{h4v.z, h4v.w};
        // inner neighbour: the other half's register row R-1 (old values), in-register exchange
        const f2 iuP = f2{lane_xor32(uP[R - 1].x, lower), lane_xor32(uP[R - 1].y, lower)};
        const f2 iuQ = f2{lane_xor32(uQ[R - 1].x, lower), lane_xor32(uQ[R - 1].y, lower)};
        const f2 ivP = f2{lane_xor32(vP[R - 1].x, lower), lane_xor32(vP[R - 1].y, lower)};
        const f2 ivQ = f2{lane_xor32(vQ[R - 1].x, lower), lane_xor32(vQ[R - 1].y, lower)};
        float e = 0.f;
        // register rows 0 .. R-1, from the published outer edge towards the inner boundary: each row needs the cross
        // sums of the boundary above it (kept from the previous row) and below it (new); row 0 goes to the LDS as soon
        // as it is done, so that the write drains under the other rows
        const float w0 = uP[0].x; // witness: the published row's u at column x0 before the sweep
        Cross sp, sc;
        if (rdist[0] <= last) cross_rows<E0 ^ 1>(sp, ouP_, ouQ_, ovP_, ovQ_, uP[0], uQ[0], vP[0], vQ[0]);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int rn = r + 1 < R ? r + 1 : r;
            if (r == R - 1) {
                if (rdist[r] <= last) {
                    if (r & 1) cross_rows<E0 ^ 1>(sc, uP[r], uQ[r], vP[r], vQ[r], iuP, iuQ, ivP, ivQ);
                    else cross_rows<E0>(sc, uP[r], uQ[r], vP[r], vQ[r], iuP, iuQ, ivP, ivQ);
                }
            } else if (r & 1) HF_CROSS(sc, E0 ^ 1, r, rn);
            else HF_CROSS(sc, E0, r, rn);
            if (r & 1) HF_ROW(r, E0 ^ 1, sc, sp);
            else HF_ROW(r, E0, sc, sp);
            if (r == 0) {
                // witness (k_jacobi_strip explains it): old and new value of the published row at column x0
                if (EM == 2)
                    seen_n += (__builtin_amdgcn_ballot_w64(fabsf(HS_DIFF1(w0, uP[0].x)) >= thr_s) & wit_mask) != 0 ? 1 : 0;
                if (s + 1 < g.T) HF_PUBLISH((s + 1) & 1);
            }
            sp = sc;
        }
        if (EM == 1 && EPS == 3) {
            e = wave_max(e) * unscale;
            if (lane == 0) eps_lds[16 + w] = e;
        }
        if (EM == 1 && EPS == 1) {
            e = wave_max(e) * unscale;
            if (lane == 0) eps_lds[(s & 1) * 16 + w] = e;
            if (s > 0 && w == 0) {
                float x = lane < NW ? eps_lds[((s - 1) & 1) * 16 + lane] : 0.f;
                x = wave_max(x);
                if (lane == 0) eps_out[(size_t)(s - 1) * eps_stride + blockIdx.x] = __float_as_uint(x);
            }
        }
        if (s + 1 < g.T) __syncthreads();
    };
    if constexpr (EPS == 3) {
#pragma unroll 1
        for (int s = 0; s + 1 < g.T; s++) sweep(s, std::integral_constant<int, 2>{});
        sweep(g.T - 1, std::integral_constant<int, 1>{});
    } else if constexpr (HS_PEEL_LAST_FOLD && EPS != 1) {
        int s = 0;
        if constexpr (HS_UNROLL2_FOLD) { // two sweeps per trip: the register copies at the loop's back edge halve
#pragma unroll 1
            for (; s + 2 < g.T; s += 2) {
                sweep(s, std::integral_constant<int, EPS>{});
                sweep(s + 1, std::integral_constant<int, EPS>{});
            }
        }
#pragma unroll 1
        for (; s + 1 < g.T; s++) sweep(s, std::integral_constant<int, EPS>{});
        sweep(g.T - 1, std::integral_constant<int, EPS>{});
    } else {
#pragma unroll 1
        for (int s = 0; s < g.T; s++) sweep(s, std::integral_constant<int, EPS>{});
    }
    if (EPS == 1) {
        __syncthreads();
        if (w == 0) {
            float x = lane < NW ? eps_lds[((g.T - 1) & 1) * 16 + lane] : 0.f;
            x = wave_max(x);
            if (lane == 0) eps_out[(size_t)(g.T - 1) * eps_stride + blockIdx.x] = __float_as_uint(x);
        }
    }
    if (EPS == 2 || EPS == 3) {
        if (lane == 0) eps_lds[w] = seen_n == (EPS == 3 ? g.T - 1 : g.T) ? __builtin_inff() : 0.f;
        __syncthreads();
        if (w == 0) {
            const float y = wave_max_nonneg(lane < NW ? eps_lds[lane] : 0.f);
            if (lane == 0) eps_out[blockIdx.x] = __float_as_uint(y);
            if (EPS == 3) { // second word: Eps of the last sweep, exact
                const float x = wave_max_nonneg(lane < NW ? eps_lds[16 + lane] : 0.f);
                if (lane == 0) eps_out[(size_t)eps_stride + blockIdx.x] = __float_as_uint(x);
            }
        }
    }
#undef HF_ROW
#undef HF_CROSS
#undef HF_PUBLISH
#undef HF_SLOT
    if (stamps) st2 = __builtin_amdgcn_s_memtime();

    if (lanecore) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            if ((rowcore >> r) & 1u) {
                const int y = yb + (lower ? 2 * R - 1 - r : r);
                const long long off = base + (long long)y * g.P + x0;
                const float fin = HS_SCALED ? __builtin_ldexpf(1.0f, -2 * g.T) : 1.0f; // back to scale 1 (exact)
                // P = (p0, p3), Q = (p1, p2)
                *(float4 *)(u_out + off) = make_float4(uP[r].x * fin, uQ[r].x * fin, uQ[r].y * fin, uP[r].y * fin);
                *(float4 *)(v_out + off) = make_float4(vP[r].x * fin, vQ[r].x * fin, vQ[r].y * fin, vP[r].y * fin);
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

template <int R, int NTMAX, int EPS, int E0>
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
    fold_body<R, NTMAX, EPS, E0, false>(coef, u_in, v_in, u_out, v_out, g, ilambda, eps_out, eps_stride, stamps, eps_thr,
                                    nullptr, nullptr, nullptr);
}

// The folded kernel as the first launch of a solve, derivative pass included (see k_jacobi_strip_deriv).
template <int R, int NTMAX, int EPS, int E0>
__global__ __launch_bounds__(NTMAX) void k_jacobi_fold_deriv(const uint8_t *__restrict__ fA,
                                                             const uint8_t *__restrict__ fB,
                                                             uint32_t *__restrict__ coef_w,
                                                             const float *__restrict__ u_in,
                                                             const float *__restrict__ v_in,
                                                             float *__restrict__ u_out,
                                                             float *__restrict__ v_out, const StripGeom g,
                                                             const float ilambda,
                                                             unsigned *__restrict__ eps_out, const int eps_stride,
                                                             unsigned long long *__restrict__ stamps,
                                                             const float eps_thr)
{
    fold_body<R, NTMAX, EPS, E0, true>(nullptr, u_in, v_in, u_out, v_out, g, ilambda, eps_out, eps_stride, stamps, eps_thr,
                                   fA, fB, coef_w);
}

} // namespace hsk
