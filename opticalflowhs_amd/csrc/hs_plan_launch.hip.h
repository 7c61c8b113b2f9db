// hs_plan_launch.hip.h -- part of libhsflow.so: launch planners (LDS-tile, strip / fold cost model) and the
// kernel launch wrappers.
#pragma once

namespace {
// How many compute units the planners count on.  A context that shares the device with other contexts' solves in flight
// (the slots of a pair pipeline: hsflow_set_cu_share) plans for its share of the chip: the shapes that minimise the
// CU-time of a solve -- few large tiles, little halo redundancy -- instead of those that spread one small frame thinly
// over all 256 CUs to shorten ITS latency while the other solves wait.  With a share the launch has no round structure
// of its own (its workgroups start wherever a CU falls free), so the cost models then count fractional rounds.
// rows whose changes count for Eps / the witness (hsflow_set_eps_rows; the whole frame unless narrowed)
int eps_row0(const hsflow_ctx *c) { return c->eps_rows > 0 ? c->eps_row0 : 0; }
int eps_row1(const hsflow_ctx *c) { return c->eps_rows > 0 ? c->eps_row0 + c->eps_rows : c->H; }
bool eps_windowed(const hsflow_ctx *c) { return c->eps_rows > 0 && (c->eps_row0 > 0 || c->eps_rows < c->H); }

int plan_cus(const hsflow_ctx *c) { return c->cu_share > 0 ? c->cu_share : kNumCU; }
bool plan_shared(const hsflow_ctx *c) { return c->cu_share > 0 && c->cu_share < kNumCU; }

// ------------------------------------------------------------------------------------------
// Tile planner for the fused kernel.  Cost model: the launch takes ceil(tiles / CUs) rounds of
// one workgroup per CU; a round costs the region area (LDS sweeps dominate) plus a fixed part.
// ------------------------------------------------------------------------------------------
bool make_plan(const hsflow_ctx *c, int T, int tw, int th, int nt, FusedPlan &best)
{
    const int W = c->W, H = c->H;
    const int HX = round_up(T, 4);
    double best_cost = 1e300;
    bool found = false;
    const int nts[3] = {1024, 512, 256};
    for (int nti = 0; nti < 3; nti++) {
        const int NT = nts[nti];
        if (nt && nt != NT) continue;
        const int Kmax = NT == 1024 ? 3 : 4;
        const int wg_per_cu = 1; // LDS-heavy tiles: plan for one resident workgroup per CU
        const int cw_lo = tw ? tw : 4, cw_hi = tw ? tw : std::min(round_up(W, 4), 1024);
        for (int CW = cw_lo; CW <= cw_hi; CW += 4) {
            const int RW4 = (CW + 2 * HX) / 4;
            const int ch_lo = th ? th : 1, ch_hi = th ? th : std::min(H, 1024);
            for (int CH = ch_lo; CH <= ch_hi; CH++) {
                const int RH = CH + 2 * T;
                const long long G = (long long)RW4 * RH;
                if (G > (long long)NT * Kmax) break; // CH only grows
                const int RS = 4 * RW4 + 8;
                const long long lds = 2LL * RS * (RH + 2) * 4;
                if (lds > kLdsLimit) break;
                const int tx = (W + CW - 1) / CW, ty = (H + CH - 1) / CH;
                const long long tiles = (long long)tx * ty * c->N;
                const int K = (int)((G + NT - 1) / NT);
                const long long ncu = plan_cus(c);
                const long long rounds = (tiles + ncu * wg_per_cu - 1) / (ncu * wg_per_cu);
                // per-round cost ~ K sweeps-worth of work per lane * T, plus load/store of the tile
                const double per_round = (double)K * NT * 4 * (T + 3.0) + 2000.0;
                const double cost = (double)rounds * per_round;
                if (cost < best_cost - 1e-9) {
                    best_cost = cost;
                    found = true;
                    best.NT = NT;
                    best.K = K;
                    best.lds_bytes = (int)lds;
                    best.tiles = (int)tiles;
                    hsk::FusedGeom &g = best.g;
                    g.W = W; g.H = H; g.P = c->P; g.plane = c->plane;
                    g.CW = CW; g.CH = CH; g.T = T; g.HX = HX;
                    g.RW4 = RW4; g.RH = RH; g.RS = RS; g.G = (int)G;
                    g.tiles_x = tx; g.tiles_y = ty; g.zero_in = 0; g.org = c->org;
                }
            }
        }
    }
    return found;
}

template <int NT, int K, bool EPS, int LR>
hipError_t launch_fused_t(const hsflow_ctx *c, const FusedPlan &p, const float *ui, const float *vi,
                          float *uo, float *vo, float coeff, bool configure_only)
{
    auto kern = hsk::k_jacobi_fused<NT, K, EPS, LR>;
    static std::atomic<bool> configured[64]; // per instantiation and device: raise the dynamic-LDS cap once (several host threads may get here)
    if (p.lds_bytes > 32 * 1024 && !configured[c->device & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
        if (e != hipSuccess) return e;
        configured[c->device & 63] = true;
    }
    if (configure_only) return hipSuccess; // done ahead of a stream capture
    hipLaunchKernelGGL(kern, dim3(p.tiles), dim3(NT), p.lds_bytes, c->stream, c->dCoef, ui, vi, uo,
                       vo, p.g, coeff, c->epsPtr, c->epsStride);
    return hipGetLastError();
}

template <bool EPS, int LR>
hipError_t launch_fused_e(const hsflow_ctx *c, const FusedPlan &p, const float *ui, const float *vi,
                          float *uo, float *vo, float coeff, bool cfg)
{
#define HS_CASE(NT_, K_)                                                                          \
    if (p.NT == NT_ && p.K == K_) return launch_fused_t<NT_, K_, EPS, LR>(c, p, ui, vi, uo, vo, coeff, cfg);
    HS_CASE(1024, 1) HS_CASE(1024, 2) HS_CASE(1024, 3)
    HS_CASE(512, 1) HS_CASE(512, 2) HS_CASE(512, 3) HS_CASE(512, 4)
    HS_CASE(256, 1) HS_CASE(256, 2) HS_CASE(256, 3) HS_CASE(256, 4)
#undef HS_CASE
    return hipErrorInvalidConfiguration;
}

hipError_t launch_fused(const hsflow_ctx *c, const FusedPlan &p, bool eps, int lr, const float *ui,
                        const float *vi, float *uo, float *vo, float coeff, bool cfg = false)
{
    if (eps) return lr ? launch_fused_e<true, 1>(c, p, ui, vi, uo, vo, coeff, cfg)
                       : launch_fused_e<true, 0>(c, p, ui, vi, uo, vo, coeff, cfg);
    return lr ? launch_fused_e<false, 1>(c, p, ui, vi, uo, vo, coeff, cfg)
              : launch_fused_e<false, 0>(c, p, ui, vi, uo, vo, coeff, cfg);
}


template <int NT, int K>
hipError_t launch_classic_fused_t(const hsflow_ctx *c, const FusedPlan &p, bool write_v, const float *ui, const float *vi,
                                  float *uo, float *vo, float alpha2, bool configure_only)
{
    auto kern = write_v ? hsk::k_jacobi_classic_fused<NT, K, true> : hsk::k_jacobi_classic_fused<NT, K, false>;
    static std::atomic<bool> configured[2][64];
    if (p.lds_bytes > 32 * 1024 && !configured[write_v][c->device & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
        if (e != hipSuccess) return e;
        configured[write_v][c->device & 63] = true;
    }
    if (configure_only) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(p.tiles), dim3(NT), p.lds_bytes, c->stream, c->dE[0], c->dE[1], c->dE[2], ui, vi, uo, vo, p.g, alpha2);
    return hipGetLastError();
}

hipError_t launch_classic_fused(const hsflow_ctx *c, const FusedPlan &p, bool write_v, const float *ui, const float *vi,
                                float *uo, float *vo, float alpha2, bool cfg = false)
{
#define HS_CASE(NT_, K_)                                                                          \
    if (p.NT == NT_ && p.K == K_) return launch_classic_fused_t<NT_, K_>(c, p, write_v, ui, vi, uo, vo, alpha2, cfg);
    HS_CASE(1024, 1) HS_CASE(1024, 2) HS_CASE(1024, 3)
    HS_CASE(512, 1) HS_CASE(512, 2) HS_CASE(512, 3) HS_CASE(512, 4)
    HS_CASE(256, 1) HS_CASE(256, 2) HS_CASE(256, 3) HS_CASE(256, 4)
#undef HS_CASE
    return hipErrorInvalidConfiguration;
}

// ------------------------------------------------------------------------------------------
// Classic mode, register strip kernel (hs_kernels_classic_strip.hip.h): plan and launch.
// The shapes that are compiled: R rows per lane with at most classic_strip_max_waves(R) wavefronts (register budget:
// 24 registers of state per row and about 45 of working set).
// ------------------------------------------------------------------------------------------
struct ClassicStripPlan {
    hsk::ClassicStripGeom g;
    int R, tiles, lds_bytes;
};

int classic_strip_max_waves(int R) { return R <= 3 ? 16 : (R <= 5 ? 12 : 8); }

// Is (R, NW) a legal shape for T sweeps on this image?  The row clamps are free only where image row 0 is register row 0
// of a wavefront and image row H-1 register row R-1 of one (kernel header); a region that meets a border unaligned must
// own no row within T rows of it.
bool classic_strip_geom(const hsflow_ctx *c, int T, int R, int NW, hsk::ClassicStripGeom &g)
{
    const int W = c->W, H = c->H;
    const int TH = round_up(T, R), HX = round_up(T, 4);
    const int CW = 256 - 2 * HX, CH = NW * R - 2 * TH;
    if (CW < 4 || CH < 1 || CH < TH) return false;
    const int ty = (H + CH - 1) / CH;
    if (ty == 1) {
        if (H % R != 0) return false; // one tile row: top-aligned, so the bottom is aligned only if R divides H
    } else {
        const int ylast = H - CH - TH; // last tile row, aligned to the bottom
        if (ylast < 0 && (-ylast) % R != 0 && H - CH < T) return false; // row 0 inside it, unaligned and too near
    }
    g.W = W; g.H = H; g.P = c->P; g.plane = c->plane;
    g.T = T; g.TH = TH; g.HX = HX; g.CW = CW; g.CH = CH; g.NW = NW;
    g.tiles_x = (W + CW - 1) / CW; g.tiles_y = ty;
    g.ylast = ty == 1 ? -TH : H - CH - TH;
    g.zero_in = 0;
    return true;
}

// Modelled time of one launch in microseconds, fitted to tools/sweep_classic_strip.py on MI355X at 480p, 720p, 1080p and
// 4K (profiles/r02_sweep_classic_strip.txt; within 10 % there): a launch gap, the planes through the cache hierarchy
// (12 bytes per pixel in, 8 out), then per round of workgroups the rows the busiest SIMD holds, each loaded and
// unpacked once and swept T times (halo wavefronts drop out one by one, fewer resident wavefronts hide less latency).
double classic_strip_launch_us(const hsflow_ctx *c, const hsk::ClassicStripGeom &g, int R, long long tiles, int *wg_per_cu_out = nullptr)
{
    const int NW = g.NW, lds = NW * 8192;
    const int wg_per_cu = std::max(1, std::min(kLdsLimit / lds, classic_strip_max_waves(R) / NW));
    if (wg_per_cu_out) *wg_per_cu_out = wg_per_cu;
    const int ncu = plan_cus(c);
    const double rr = (double)tiles / ((double)ncu * wg_per_cu);
    const double rounds = plan_shared(c) ? std::max(1.0, rr) : std::ceil(rr); // (a share of the chip: no round structure, see plan_cus)
    const long long conc = std::min<long long>(wg_per_cu, (tiles + ncu - 1) / ncu); // workgroups sharing a CU
    const double wps = (double)((NW + 3) / 4) * (double)conc;                              // wavefronts on the busiest SIMD
    const double rps = wps * R;                                                              // ... and their rows
    // per row and sweep: 0.176 us for the shapes that divide with the precomputed reciprocal (R = 2, 3, 4, 6), 0.2 for the
    // others; a single wavefront per SIMD hides less latency (two are as good as four since the arithmetic is packed)
    const double row_us = ((R <= 4 || R == 6) ? 0.176 : 0.2) * (wps >= 1.5 ? 1.0 : 1.3);
    const double halo_frac = std::min(1.0, 2.0 * g.TH / (double)(NW * R));
    const double bytes_us = (double)g.W * g.H * c->N * 12.0 / 7e6;
    return 2.0 + bytes_us + rounds * rps * (0.32 + row_us * g.T * (1.0 - 0.25 * halo_frac));
}

bool make_classic_strip_plan(const hsflow_ctx *c, int T, int rows, int threads, ClassicStripPlan &best, double *cost_out = nullptr)
{
    double best_cost = 1e300;
    bool found = false;
    for (int R = 2; R <= 8; R++) {
        if (rows && rows != R) continue;
        for (int NW = 1; NW <= classic_strip_max_waves(R); NW++) {
            if (threads && threads != NW * 64) continue;
            hsk::ClassicStripGeom g;
            if (!classic_strip_geom(c, T, R, NW, g)) continue;
            const long long tiles = (long long)g.tiles_x * g.tiles_y * c->N;
            // (equal in the model: the same rows per SIMD in fewer, longer wavefronts measure 1 - 3 % faster)
            const double cost = classic_strip_launch_us(c, g, R, tiles) * (1.0 - 0.004 * R);
            if (cost < best_cost) {
                best_cost = cost;
                found = true;
                best.g = g; best.R = R; best.tiles = (int)tiles; best.lds_bytes = NW * 8192;
            }
        }
    }
    if (cost_out) *cost_out = best_cost;
    return found;
}

// Sweeps per launch when the caller leaves them open: minimise the modelled time of the whole solve (full launches of T
// plus one tail launch).  The regions are at most 64 rows high (register budget), so the answer is small: 6 at 1080p.
int pick_classic_strip_T(const hsflow_ctx *c, int iters, int rows, int threads)
{
    double best = 1e300;
    int bestT = 0;
    for (int T = 1; T <= std::min(iters, 16); T++) {
        ClassicStripPlan sp;
        double cfull = 0, ctail = 0;
        if (!make_classic_strip_plan(c, T, rows, threads, sp, &cfull)) continue;
        const int rem = iters % T;
        if (rem && !make_classic_strip_plan(c, rem, rows, threads, sp, &ctail)) continue;
        const double total = (iters / T) * cfull + (rem ? ctail : 0.0);
        if (total < best) { best = total; bestT = T; }
    }
    return bestT;
}

template <int R, int NTMAX>
hipError_t launch_classic_strip_t(const hsflow_ctx *c, const ClassicStripPlan &p, bool write_v, const float *ui, const float *vi,
                                  float *uo, float *vo, float alpha2, bool configure_only)
{
    const bool ghost = (p.g.W & 3) != 0;
    auto kern = write_v ? (ghost ? hsk::k_classic_strip<R, NTMAX, true, true> : hsk::k_classic_strip<R, NTMAX, true, false>)
                        : (ghost ? hsk::k_classic_strip<R, NTMAX, false, true> : hsk::k_classic_strip<R, NTMAX, false, false>);
    static std::atomic<bool> configured[4][64];
    const int ki = (write_v ? 2 : 0) + (ghost ? 1 : 0);
    if (p.lds_bytes > 32 * 1024 && !configured[ki][c->device & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
        if (e != hipSuccess) return e;
        configured[ki][c->device & 63] = true;
    }
    if (configure_only) return hipSuccess;
    if (p.g.zero_in) ui = vi = c->dZero; // (flow from zero: one row of zeros stands in for both planes)
    hipLaunchKernelGGL(kern, dim3(p.tiles), dim3(p.g.NW * 64), p.lds_bytes, c->stream, c->dCoef, ui, vi, uo, vo, p.g, alpha2);
    return hipGetLastError();
}

hipError_t launch_classic_strip(const hsflow_ctx *c, const ClassicStripPlan &p, bool write_v, const float *ui, const float *vi,
                                float *uo, float *vo, float alpha2, bool cfg = false)
{
    switch (p.R) {
    case 2: return launch_classic_strip_t<2, 1024>(c, p, write_v, ui, vi, uo, vo, alpha2, cfg);
    case 3: return launch_classic_strip_t<3, 1024>(c, p, write_v, ui, vi, uo, vo, alpha2, cfg);
    case 4: return launch_classic_strip_t<4, 768>(c, p, write_v, ui, vi, uo, vo, alpha2, cfg);
    case 5: return launch_classic_strip_t<5, 768>(c, p, write_v, ui, vi, uo, vo, alpha2, cfg);
    case 6: return launch_classic_strip_t<6, 512>(c, p, write_v, ui, vi, uo, vo, alpha2, cfg);
    case 7: return launch_classic_strip_t<7, 512>(c, p, write_v, ui, vi, uo, vo, alpha2, cfg);
    case 8: return launch_classic_strip_t<8, 512>(c, p, write_v, ui, vi, uo, vo, alpha2, cfg);
    }
    return hipErrorInvalidConfiguration;
}

// ------------------------------------------------------------------------------------------
// Planner for the strip kernel: rows per lane R and wavefronts per workgroup NW.
// Register budget fixes the wavefronts a SIMD can hold: R <= 5 -> 4, R = 6 -> 3, R <= 8 -> 2.
// ------------------------------------------------------------------------------------------

// Cost model (shader cycles at ~2.2 GHz), fitted to in-kernel phase stamps on MI355X at 1080p
// (tools/stamps.py; profiles/): one launch = fixed launch/drain gap + per round [tile load +
// T sweeps], where a sweep costs ~1.5 x (VALU time of the busiest SIMD + LDS edge-row exchange).
int strip_max_waves(int R, int fold) { const int r = fold ? R + 1 : R; return r <= 5 ? 16 : (r <= 6 ? 12 : 8); }


double strip_launch_cost(const hsflow_ctx *c, int T, int R, int NW, long long tiles, int fold, double image_pixels, int *wg_per_cu_out = nullptr)
{
    const int ncu = plan_cus(c);
    // Parameters fitted (least squares on log time, rms 8 %) to profiles/r01_sweep_1080p_strip5.csv,
    // r01_sweep_4k_b.csv and r01_sweep_batch16.csv.  The sweep is VALU-issue bound (~34 instructions
    // per row per wavefront, ~4.2 cycles each per SIMD with 4 resident wavefronts, more with fewer);
    // a launch boundary costs ~2.7 us plus the L2 write-back of the 8 bytes per pixel just stored.
    const int per_simd = strip_max_waves(R, fold) / 4;
    const int lds = NW * (fold ? 4096 : 8192);
    const int wg_per_cu = std::max(1, std::min(std::min(kLdsLimit / lds, (per_simd * 4) / NW), 8));
    if (wg_per_cu_out) *wg_per_cu_out = wg_per_cu;
    const long long slots = (long long)ncu * wg_per_cu;
    const long long conc = std::min<long long>(wg_per_cu, (tiles + ncu - 1) / ncu); // WGs sharing a CU
    const double wps = (double)(conc * NW) / 4.0;                                          // wavefronts per SIMD
    const double cpi = wps >= 3.5 ? 4.2 : (wps >= 2.5 ? 5.6 : (wps >= 1.5 ? 6.5 : 8.0));
    const int rows_per_lane = fold ? 2 * R : R;
    const double halo_frac = std::min(1.0, 2.0 * T / (double)(NW * rows_per_lane));
    const double instr_per_row = fold ? 42.0 : 34.0;
    const double valu = std::max(wps, 1.0) * R * instr_per_row * cpi * (1.0 - 0.45 * halo_frac);
    const double exchange = 300.0 + (fold ? 12.0 : 20.0) * NW * conc;
    const double sweep = valu + exchange;
    double load = 2050.0 + 0.4 * 256.0 * R * NW * conc;
    if (conc > 1) load *= 0.4; // another workgroup's sweeps hide part of it
    const double r = (double)tiles / (double)slots;
    const double rounds = plan_shared(c) ? std::max(1.0, r) : conc == 1 ? std::ceil(r) : std::max(1.0, r + 0.7);
    return 6000.0 + 6e-4 * 8.0 * image_pixels + rounds * (load + T * sweep);
}

bool make_strip_plan(const hsflow_ctx *c, int T, int rows, int threads, int fold, StripPlan &best, double *cost_out = nullptr)
{
    const int W = c->W, H = c->H;
    const int HX = round_up(T, 4);
    const int CW = (fold ? 128 : 256) - 2 * HX;
    if (CW < 4) return false;
    double best_cost = 1e300;
    bool found = false;
    for (int R = 1; R <= 8; R++) {
        if (rows && rows != R) continue;
        for (int NW = 1; NW <= strip_max_waves(R, fold); NW++) {
            if (threads && threads != NW * 64) continue;
            const int CH = NW * R * (fold ? 2 : 1) - 2 * T;
            if (CH < 1) continue;
            const int lds = NW * (fold ? 4096 : 8192) + (fold ? 128 : 256); // edge-row exchange + 32 floats for Eps (+ sweep counters)
            if (lds > kLdsLimit) continue;
            const int tx = (W + CW - 1) / CW, ty = (H + CH - 1) / CH;
            const long long tiles = (long long)tx * ty * c->N;
            const double cost = strip_launch_cost(c, T, R, NW, tiles, fold, (double)W * H * c->N);
            if (cost < best_cost - 1e-9) {
                best_cost = cost;
                found = true;
                best.R = R;
                best.fold = fold;
                best.lds_bytes = lds;
                best.tiles = (int)tiles;
                hsk::StripGeom &g = best.g;
                g.W = W; g.H = H; g.P = c->P; g.plane = c->plane;
                g.T = T; g.HX = HX; g.CW = CW; g.CH = CH; g.NW = NW;
                g.tiles_x = tx; g.tiles_y = ty; g.zero_in = 0; g.org = c->org;
                g.ey0 = eps_row0(c); g.ey1 = eps_row1(c);
            }
        }
    }
    if (cost_out) *cost_out = best_cost;
    return found;
}

// Sweeps per launch for a budget of `iters` sweeps: minimise the modelled time of the whole solve
// (full launches of T plus one tail launch of iters % T).
int pick_strip_T(const hsflow_ctx *c, int iters, const hsflow_params &p, int fold, double *total_out = nullptr)
{
    double best = 1e300;
    int bestT = 1;
    for (int T = 1; T <= std::min(iters, 24); T++) {
        StripPlan sp;
        double cfull = 0, ctail = 0;
        if (!make_strip_plan(c, T, p.strip_rows, p.threads, fold, sp, &cfull)) continue;
        const int rem = iters % T;
        if (rem && !make_strip_plan(c, rem, p.strip_rows, p.threads, fold, sp, &ctail)) continue;
        const double total = (iters / T) * cfull + (rem ? ctail : 0.0);
        if (total < best) { best = total; bestT = T; }
    }
    if (total_out) *total_out = best;
    return bestT;
}

// The checkerboard phase the kernel is compiled for (hs_kernels_strip.hip.h): the folded kernel and the strip kernel
// with an even row count start every strip on a row of the parity of T; odd row counts have one instantiation
// (strips that start on an odd row keep their rows in reverse order instead).
int strip_phase(const StripPlan &p) { return (p.fold || (p.R & 1) == 0) ? ((p.g.T + p.g.org) & 1) : 0; }

template <int R, int NTMAX, int EPS, bool FOLD, int E0> // EPS: 0 none, 1 every sweep, 2 witness, 3 witness + last sweep measured
hipError_t launch_strip_te(const hsflow_ctx *c, const StripPlan &p, const float *ui, const float *vi,
                           float *uo, float *vo, float coeff, bool configure_only)
{
    auto kern = [] {
        if constexpr (FOLD) return hsk::k_jacobi_fold<R, NTMAX, EPS, E0>;
        else return hsk::k_jacobi_strip<R, NTMAX, EPS, E0>;
    }();
    static std::atomic<bool> configured[64];
    if (p.lds_bytes > 32 * 1024 && !configured[c->device & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
        if (e != hipSuccess) return e;
        configured[c->device & 63] = true;
    }
    if (configure_only) return hipSuccess;
    if (p.g.zero_in) ui = vi = c->dZero; // flow from zero: one row of zeros stands in for both planes (StripGeom::zero_in)
    hipLaunchKernelGGL(kern, dim3(p.tiles), dim3(p.g.NW * 64), p.lds_bytes, c->stream, c->dCoef, ui, vi,
                       uo, vo, p.g, coeff, c->epsPtr, c->epsStride, p.tiles <= 65536 ? c->dStamps : nullptr, c->epsThr);
    return hipGetLastError();
}

template <int R, int NTMAX, int EPS, bool FOLD>
hipError_t launch_strip_t(const hsflow_ctx *c, const StripPlan &p, const float *ui, const float *vi,
                          float *uo, float *vo, float coeff, bool cfg)
{
    if constexpr (FOLD || (R & 1) == 0) {
        if (strip_phase(p)) return launch_strip_te<R, NTMAX, EPS, FOLD, 1>(c, p, ui, vi, uo, vo, coeff, cfg);
    }
    return launch_strip_te<R, NTMAX, EPS, FOLD, 0>(c, p, ui, vi, uo, vo, coeff, cfg);
}

template <int EPS, bool FOLD>
hipError_t launch_strip_e(const hsflow_ctx *c, const StripPlan &p, const float *ui, const float *vi,
                          float *uo, float *vo, float coeff, bool cfg)
{
#ifdef HS_DIAG_MIN /* diagnostic builds (tools/diag_build.sh): the R = 4 / 5 / 6 strip kernels without Eps only */
    if constexpr (!FOLD && EPS == 0) {
        if (p.R == 4) return launch_strip_t<4, 1024, EPS, FOLD>(c, p, ui, vi, uo, vo, coeff, cfg);
        if (p.R == 5) return launch_strip_t<5, 1024, EPS, FOLD>(c, p, ui, vi, uo, vo, coeff, cfg);
        if (p.R == 6) return launch_strip_t<6, 768, EPS, FOLD>(c, p, ui, vi, uo, vo, coeff, cfg);
    }
    return hipErrorInvalidConfiguration;
#else
    switch (p.R) {
    case 1: return launch_strip_t<1, 1024, EPS, FOLD>(c, p, ui, vi, uo, vo, coeff, cfg);
    case 2: return launch_strip_t<2, 1024, EPS, FOLD>(c, p, ui, vi, uo, vo, coeff, cfg);
    case 3: return launch_strip_t<3, 1024, EPS, FOLD>(c, p, ui, vi, uo, vo, coeff, cfg);
    case 4: return launch_strip_t<4, 1024, EPS, FOLD>(c, p, ui, vi, uo, vo, coeff, cfg);
    case 5:
        if (!FOLD && EPS && p.g.NW <= 12) return launch_strip_t<5, 768, EPS, FOLD>(c, p, ui, vi, uo, vo, coeff, cfg);
        return launch_strip_t<5, FOLD ? 768 : 1024, EPS, FOLD>(c, p, ui, vi, uo, vo, coeff, cfg);
    case 6: return launch_strip_t<6, FOLD ? 512 : 768, EPS, FOLD>(c, p, ui, vi, uo, vo, coeff, cfg);
    case 7: return launch_strip_t<7, 512, EPS, FOLD>(c, p, ui, vi, uo, vo, coeff, cfg);
    case 8: return launch_strip_t<8, 512, EPS, FOLD>(c, p, ui, vi, uo, vo, coeff, cfg);
    }
    return hipErrorInvalidConfiguration;
#endif
}

// The strip / folded kernel with the derivative pass in its load phase (first launch of a solve).
template <int R, int NTMAX, int EPS, bool FOLD, int E0>
hipError_t launch_strip_deriv_te(const hsflow_ctx *c, const StripPlan &p, const float *ui, const float *vi,
                                 float *uo, float *vo, float coeff, bool configure_only)
{
    auto kern = [] {
        if constexpr (FOLD) return hsk::k_jacobi_fold_deriv<R, NTMAX, EPS, E0>;
        else return hsk::k_jacobi_strip_deriv<R, NTMAX, EPS, E0>;
    }();
    static std::atomic<bool> configured[64];
    if (p.lds_bytes > 32 * 1024 && !configured[c->device & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
        if (e != hipSuccess) return e;
        configured[c->device & 63] = true;
    }
    if (configure_only) return hipSuccess;
    if (p.g.zero_in) ui = vi = c->dZero;
    hipLaunchKernelGGL(kern, dim3(p.tiles), dim3(p.g.NW * 64), p.lds_bytes, c->stream, c->dA, c->dB, c->dCoef, ui, vi,
                       uo, vo, p.g, coeff, c->epsPtr, c->epsStride, p.tiles <= 65536 ? c->dStamps : nullptr, c->epsThr);
    return hipGetLastError();
}

template <int R, int NTMAX, int EPS, bool FOLD>
hipError_t launch_strip_deriv_t(const hsflow_ctx *c, const StripPlan &p, const float *ui, const float *vi,
                                float *uo, float *vo, float coeff, bool cfg)
{
    if constexpr (FOLD || (R & 1) == 0) {
        if (strip_phase(p)) return launch_strip_deriv_te<R, NTMAX, EPS, FOLD, 1>(c, p, ui, vi, uo, vo, coeff, cfg);
    }
    return launch_strip_deriv_te<R, NTMAX, EPS, FOLD, 0>(c, p, ui, vi, uo, vo, coeff, cfg);
}

template <int EPS, bool FOLD>
hipError_t launch_strip_deriv_e(const hsflow_ctx *c, const StripPlan &p, const float *ui, const float *vi,
                                float *uo, float *vo, float coeff, bool cfg)
{
#ifdef HS_DIAG_MIN
    return hipErrorInvalidConfiguration;
#else
    switch (p.R) { // the same thread limits as launch_strip_e
    case 1: return launch_strip_deriv_t<1, 1024, EPS, FOLD>(c, p, ui, vi, uo, vo, coeff, cfg);
    case 2: return launch_strip_deriv_t<2, 1024, EPS, FOLD>(c, p, ui, vi, uo, vo, coeff, cfg);
    case 3: return launch_strip_deriv_t<3, 1024, EPS, FOLD>(c, p, ui, vi, uo, vo, coeff, cfg);
    case 4: return launch_strip_deriv_t<4, 1024, EPS, FOLD>(c, p, ui, vi, uo, vo, coeff, cfg);
    case 5:
        if (!FOLD && EPS && p.g.NW <= 12) return launch_strip_deriv_t<5, 768, EPS, FOLD>(c, p, ui, vi, uo, vo, coeff, cfg);
        return launch_strip_deriv_t<5, FOLD ? 768 : 1024, EPS, FOLD>(c, p, ui, vi, uo, vo, coeff, cfg);
    case 6: return launch_strip_deriv_t<6, FOLD ? 512 : 768, EPS, FOLD>(c, p, ui, vi, uo, vo, coeff, cfg);
    }
    return hipErrorInvalidConfiguration;
#endif
}

// ------------------------------------------------------------------------------------------
// The strip kernel as ONE persistent launch per solve (k_jacobi_strip_persist, hs_kernels_strip.hip.h).
// ------------------------------------------------------------------------------------------
constexpr unsigned kPersistWaitTicks = 5000000u; // 50 ms of the 100 MHz clock: a healthy wait is a few microseconds

// Why this solve cannot run as one persistent launch (nullptr: it can).  sp: the strip plan for T sweeps per phase.
const char *persist_obstacle(const hsflow_ctx *c, const StripPlan &sp, int iters, const hsflow_params &p, bool async, bool use_eps)
{
    static const bool off = getenv("HSFLOW_NO_PERSIST") != nullptr;
    const hsk::StripGeom &g = sp.g;
    if (off) return "HSFLOW_NO_PERSIST is set";
    if (c->persist_off) return "a persistent launch timed out on this context earlier";
    if (sp.fold) return "the folded kernel has no persistent form";
    if (sp.R != 5 || g.NW * 64 > 1024) return "the persistent form is compiled for 5 rows per lane";
    if (iters <= g.T) return "a single phase";
    if ((g.W & 3) != 0 || g.W < 256 || g.H < g.NW * sp.R) return "the frame must be at least one region large and its width a multiple of 4";
    if (g.CH < g.T || g.CW < g.HX) return "the halo reaches past the neighbouring tiles";
    if (sp.tiles > (c->num_cu > 0 ? c->num_cu : kNumCU) || sp.tiles > kMaxPersistTiles) return "more tiles than compute units: not all workgroups would be resident";
    if (use_eps && !async) return "synchronous ITER|EPS measures its last sweep: launch per fuse_steps";
    if (c->device >= 0 && g_live_ctx[c->device & 63].load() > 1) return "another context is alive on this device";
    return nullptr;
}

template <int R, int NTMAX, int EPS, bool DERIV, int E0>
hipError_t launch_persist_te(hsflow_ctx *c, const StripPlan &p, const hsk::PersistArgs &pa, const float *ui, const float *vi,
                             float coeff, bool configure_only)
{
    auto kern = hsk::k_jacobi_strip_persist<R, NTMAX, EPS, E0, DERIV>;
    static std::atomic<int> resident[64]; // workgroups per CU the runtime promises for this shape (0: not asked yet)
    const int dv = c->device & 63;
    if (resident[dv].load() == 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
        if (e != hipSuccess) return e;
        int nb = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kern), p.g.NW * 64, (size_t)p.lds_bytes);
        if (e != hipSuccess) return e;
        resident[dv] = nb >= 1 ? nb : -1;
    }
    if (resident[dv].load() < 1) return hipErrorCooperativeLaunchTooLarge;
    if (configure_only) return hipSuccess;
    if (p.g.zero_in) ui = vi = c->dZero;
    hipLaunchKernelGGL(kern, dim3(p.tiles), dim3(p.g.NW * 64), p.lds_bytes, c->stream, c->dA, c->dB, c->dCoef, ui, vi, p.g, coeff,
                       c->epsPtr, c->epsStride, p.tiles <= 8192 ? c->dStamps : nullptr, c->epsThr, pa);
    return hipGetLastError();
}

// eps: 0 plain phases, 2 witness phases (one row of c->epsStride words per phase at c->epsPtr)
hipError_t launch_persist(hsflow_ctx *c, const StripPlan &sp, const hsk::PersistArgs &pa, int eps, bool deriv, const float *ui,
                          const float *vi, float coeff, bool cfg = false)
{
    if (sp.R != 5) return hipErrorInvalidConfiguration;
    if (eps == 2) return deriv ? launch_persist_te<5, 1024, 2, true, 0>(c, sp, pa, ui, vi, coeff, cfg)
                               : launch_persist_te<5, 1024, 2, false, 0>(c, sp, pa, ui, vi, coeff, cfg);
    return deriv ? launch_persist_te<5, 1024, 0, true, 0>(c, sp, pa, ui, vi, coeff, cfg)
                 : launch_persist_te<5, 1024, 0, false, 0>(c, sp, pa, ui, vi, coeff, cfg);
}

// Can the first launch of a solve compute the derivatives itself (k_jacobi_strip_deriv / k_jacobi_fold_deriv)?
// The kernels' reflection argument wants a single bounce: an image at least as large as one workgroup's
// region (256 or 128 columns x all its rows).
bool strip_deriv_fusable(const hsflow_ctx *c, const JPlan &pl)
{
#ifdef HS_DIAG_MIN
    return false;
#endif
    static const bool off = getenv("HSFLOW_NO_DERIV_FUSION") != nullptr;
    if (off || (pl.kind != HSFLOW_KERNEL_STRIP && pl.kind != HSFLOW_KERNEL_FOLD)) return false;
    if (pl.s.R < 1 || pl.s.R > 6) return false; // 7 and 8 rows per lane: not instantiated
    const int region_w = pl.s.fold ? 128 : 256, region_h = pl.s.g.NW * pl.s.R * (pl.s.fold ? 2 : 1);
    return c->W >= region_w && c->H >= region_h;
}

// Can a witness launch of this plan prove anything?  The kernels watch register row 0 of a wavefront (of either half
// in the folded kernel), which must be a core row of the tile.  With an odd row count the strips alternate between
// normal and reversed row order (hs_kernels_strip.hip.h), starting with either, depending on the tile row.
bool strip_has_witness(const JPlan &pl)
{
    if (pl.kind != HSFLOW_KERNEL_STRIP && pl.kind != HSFLOW_KERNEL_FOLD) return false;
    const hsk::StripGeom &g = pl.s.g;
    const int R = pl.s.R;
    auto core = [&](int j) { return j >= g.T && j < g.T + g.CH; };
    if (pl.s.fold) {
        for (int w = 0; w < g.NW; w++)
            if (core(w * 2 * R) || core(w * 2 * R + 2 * R - 1)) return true;
        return false;
    }
    for (int first_rev = 0; first_rev < ((R & 1) ? 2 : 1); first_rev++) {
        bool any = false;
        for (int w = 0; w < g.NW && !any; w++) {
            const bool rev = (R & 1) && (((w & 1) != 0) != (first_rev != 0));
            any = core(w * R + (rev ? R - 1 : 0));
        }
        if (!any) return false;
    }
    return true;
}

bool make_jplan(const hsflow_ctx *c, int kind, int T, const hsflow_params &p, JPlan &out)
{
    out.kind = kind;
    out.T = T;
    if (kind == HSFLOW_KERNEL_STRIP) return make_strip_plan(c, T, p.strip_rows, p.threads, 0, out.s);
    if (kind == HSFLOW_KERNEL_FOLD) return make_strip_plan(c, T, p.strip_rows, p.threads, 1, out.s);
    return make_plan(c, T, p.tile_w, p.tile_h, p.threads, out.f);
}

// A plan for T sweeps that witness launches can use: the planner's own choice if it qualifies, else the first shape
// (rows per lane, wavefronts) that does, as far as the caller left those open.  Only thin frames and very short
// launches need the search (a core tile thinner than a strip).
bool make_witness_jplan(const hsflow_ctx *c, int kind, int T, const hsflow_params &p, JPlan &out)
{
    if (make_jplan(c, kind, T, p, out) && strip_has_witness(out)) return true;
    if (kind != HSFLOW_KERNEL_STRIP && kind != HSFLOW_KERNEL_FOLD) return false;
    static const int order[8] = {5, 4, 6, 3, 2, 8, 7, 1};
    for (int R : order) {
        if (p.strip_rows && p.strip_rows != R) continue;
        for (int NW = 16; NW >= 1; NW--) {
            if (p.threads && p.threads != NW * 64) continue;
            hsflow_params q = p;
            q.strip_rows = R;
            q.threads = NW * 64;
            JPlan alt;
            if (make_jplan(c, kind, T, q, alt) && strip_has_witness(alt)) { out = alt; return true; }
        }
    }
    return false;
}

// eps: 0 none, 1 Eps of every sweep, 2 witness (strip / fold: one lower bound per launch), 3 witness + the exact Eps
// of the last sweep (two words per workgroup)
// deriv: this launch also does the derivative pass (only where strip_deriv_fusable() said so)
hipError_t launch_j(const hsflow_ctx *c, const JPlan &pl, int eps, const float *ui, const float *vi,
                    float *uo, float *vo, float coeff, bool cfg = false, int zero_in = 0, bool deriv = false)
{
    if (pl.kind == HSFLOW_KERNEL_STRIP || pl.kind == HSFLOW_KERNEL_FOLD) {
        StripPlan sp = pl.s;
        sp.g.zero_in = zero_in;
        if (eps == 3) { // witness + exact Eps of the last sweep
            if (sp.fold) return deriv ? launch_strip_deriv_e<3, true>(c, sp, ui, vi, uo, vo, coeff, cfg)
                                      : launch_strip_e<3, true>(c, sp, ui, vi, uo, vo, coeff, cfg);
            return deriv ? launch_strip_deriv_e<3, false>(c, sp, ui, vi, uo, vo, coeff, cfg)
                         : launch_strip_e<3, false>(c, sp, ui, vi, uo, vo, coeff, cfg);
        }
        if (deriv && sp.fold) return eps == 2 ? launch_strip_deriv_e<2, true>(c, sp, ui, vi, uo, vo, coeff, cfg)
                                     : eps  ? launch_strip_deriv_e<1, true>(c, sp, ui, vi, uo, vo, coeff, cfg)
                                            : launch_strip_deriv_e<0, true>(c, sp, ui, vi, uo, vo, coeff, cfg);
        if (deriv) return eps == 2 ? launch_strip_deriv_e<2, false>(c, sp, ui, vi, uo, vo, coeff, cfg)
                          : eps  ? launch_strip_deriv_e<1, false>(c, sp, ui, vi, uo, vo, coeff, cfg)
                                 : launch_strip_deriv_e<0, false>(c, sp, ui, vi, uo, vo, coeff, cfg);
        if (sp.fold) return eps == 2 ? launch_strip_e<2, true>(c, sp, ui, vi, uo, vo, coeff, cfg)
                            : eps  ? launch_strip_e<1, true>(c, sp, ui, vi, uo, vo, coeff, cfg)
                                   : launch_strip_e<0, true>(c, sp, ui, vi, uo, vo, coeff, cfg);
        return eps == 2 ? launch_strip_e<2, false>(c, sp, ui, vi, uo, vo, coeff, cfg)
               : eps  ? launch_strip_e<1, false>(c, sp, ui, vi, uo, vo, coeff, cfg)
                      : launch_strip_e<0, false>(c, sp, ui, vi, uo, vo, coeff, cfg);
    }
    FusedPlan fp = pl.f;
    fp.g.zero_in = zero_in;
    return launch_fused(c, fp, eps != 0, 1, ui, vi, uo, vo, coeff, cfg);
}

void plan_to_info(hsflow_ctx *c, const JPlan &pl)
{
    hsflow_info &i = c->info;
    i.fuse_steps = pl.T;
    if (pl.kind == HSFLOW_KERNEL_STRIP || pl.kind == HSFLOW_KERNEL_FOLD) {
        i.tile_w = pl.s.g.CW; i.tile_h = pl.s.g.CH; i.threads = pl.s.g.NW * 64;
        i.groups_per_thread = pl.s.R; i.tiles = pl.s.tiles; i.lds_bytes = pl.s.lds_bytes;
    } else {
        i.tile_w = pl.f.g.CW; i.tile_h = pl.f.g.CH; i.threads = pl.f.NT;
        i.groups_per_thread = pl.f.K; i.tiles = pl.f.tiles; i.lds_bytes = pl.f.lds_bytes;
    }
}

hipError_t launch_simple(const hsflow_ctx *c, bool eps, const float *ui, const float *vi, float *uo,
                         float *vo, float coeff, int zero_in = 0)
{
    const dim3 grid((c->W + 255) / 256, (c->H + 3) / 4, c->N), block(64, 4);
#define HS_SIMPLE(E, Z)                                                                            \
    hipLaunchKernelGGL((hsk::k_jacobi_simple<E, Z>), grid, block, 0, c->stream, c->dCoef, ui, vi, uo, vo, \
                       c->W, c->H, c->P, c->plane, coeff, c->epsPtr, c->org, eps_row0(c), eps_row1(c))
    if (eps) { if (zero_in) HS_SIMPLE(true, true); else HS_SIMPLE(true, false); }
    else { if (zero_in) HS_SIMPLE(false, true); else HS_SIMPLE(false, false); }
#undef HS_SIMPLE
    return hipGetLastError();
}

hipError_t launch_deriv(const hsflow_ctx *c)
{
    const dim3 grid((c->W + 255) / 256, (c->H + 3) / 4, c->N), block(64, 4);
    hipLaunchKernelGGL(hsk::k_deriv_cv, grid, block, 0, c->stream, c->dA, c->dB, c->dCoef, c->W, c->H,
                       c->P, c->plane);
    return hipGetLastError();
}

} // namespace
