// hsflow.hip -- C ABI (include/hsflow.h) over the HIP kernels in hs_kernels.hip.h.
//
// Host side of the hot path, i.e. what HSOpticalFlowOpenCL::setupCL / runDerivatives /
// runCLKernels / cleanup did with OpenCL (OpticalFlowHS/HSOpticalFlowOpenCL.cpp:67-679,
// :849-892), re-designed for MI355X: device-resident planar buffers, no per-iteration
// host<->device copies (the reference moved u,v over PCIe twice per iteration, :483-501 and
// :655-675), one stream, the whole launch sequence optionally captured as a hipGraph.
#include "../../include/hsflow.h"
#include "hs_kernels.hip.h"
#include "hs_kernels_pre.hip.h"
#include "hs_kernels_classic.hip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

namespace {

constexpr int kMaxFuse = 32;        // upper bound on sweeps per fused launch
constexpr int kLdsLimit = 160 * 1024; // bytes of LDS per CU on gfx950
constexpr int kNumCU = 256;

thread_local std::string g_create_error; // last error of a call without a context, per host thread

struct FusedPlan {
    hsk::FusedGeom g;
    int NT, K, lds_bytes, tiles;
};

struct StripPlan {
    hsk::StripGeom g;
    int R, lds_bytes, tiles;
    int fold; // 0: k_jacobi_strip (256 columns, one strip per wavefront); 1: k_jacobi_fold (128 columns, two)
};

// A launch plan for T sweeps with either multi-sweep kernel.
struct JPlan {
    int kind = 0; // HSFLOW_KERNEL_FUSED, HSFLOW_KERNEL_STRIP or HSFLOW_KERNEL_FOLD
    int T = 0;
    FusedPlan f{};
    StripPlan s{};
};

struct GraphKey {
    int mode, kernel, max_iter, T, tw, th, nt, lr, cur, use_prev; // lr: K (fused) or R (strip)
    float coeff;
    float eps_thr = -1.f; // >= 0: the graph of an ITER|EPS witness pass with that threshold
    bool operator<(const GraphKey &o) const
    {
        return std::tie(mode, kernel, max_iter, T, tw, th, nt, lr, cur, use_prev, coeff, eps_thr) <
               std::tie(o.mode, o.kernel, o.max_iter, o.T, o.tw, o.th, o.nt, o.lr, o.cur, o.use_prev, o.coeff, o.eps_thr);
    }
};

struct GraphEntry {
    hipGraph_t graph;
    hipGraphExec_t exec;
    int cur_after, launches;
};

} // namespace

struct hsflow_ctx {
    int device = 0;
    int W = 0, H = 0, N = 0, P = 0;
    long long plane = 0; // elements per pair plane
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint8_t *dA = nullptr, *dB = nullptr;
    uint32_t *dCoef = nullptr;
    float *dE[3] = {nullptr, nullptr, nullptr}; // CLASSIC mode: Ex, Ey, Et planes (allocated on first use)
    int coef_mode = -1;          // discretisation the current derivatives belong to
    float *dU[2] = {nullptr, nullptr}, *dV[2] = {nullptr, nullptr};
    unsigned long long *dStamps = nullptr; // diagnostic phase stamps (HSFLOW_DEBUG_STAMPS), else NULL
    unsigned *dEps = nullptr;   // kMaxFuse words: Eps sink of launches that do not collect it
    unsigned *epsPtr = nullptr; // where the running launch records Eps: [sweep][epsStride] words
    float epsThr = 0.f;         // witness launches: smallest float >= epsilon
    unsigned *hEps = nullptr;   // page-locked read-back buffer for the per-sweep Eps words
    size_t hEpsCap = 0;
    int epsStride = 1;          // words per sweep: one per workgroup (strip / fold), else 1
    unsigned *dEpsTiles = nullptr; // per-sweep, per-workgroup Eps of the launches of one solve
    size_t epsTilesCap = 0;
    unsigned *dEpsAll = nullptr; // one word per sweep of a whole ITER|EPS solve (speculative run)
    int epsAllCap = 0;
    float *dUb = nullptr, *dVb = nullptr; // backup of the starting flow (ITER|EPS with use_previous)
    void *dScratch = nullptr;   // staging for colour frames / derivative read-back
    size_t scratch_bytes = 0;
    int cur = 0;                // which of dU/dV holds the current flow
    bool frames_set = false;
    bool coef_valid = false;
    hsflow_info info;
    std::string err;
    // an ITER|EPS solve enqueued by hsflow_solve_async whose early-stop check is still owed
    struct Pending {
        bool active = false;
        hsflow_params params;
        int iters = 0, slots = 0, launches = 0, cur0 = 0;
    } pend;
    bool force_exact = false; // the exact per-sweep pass is wanted (set while a pending solve is settled)
    std::map<GraphKey, GraphEntry> graphs;
    std::vector<hipEvent_t> events;
};

namespace {

hsflow_ctx *g_oneshot = nullptr; // context kept by hsflow_calc_optical_flow_hs_8u32f between calls
std::mutex g_oneshot_mutex;

int fail(hsflow_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define HS_HIP(c, call)                                                                           \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail((c), e_ == hipErrorOutOfMemory ? HSFLOW_E_OOM : HSFLOW_E_DEVICE,          \
                        std::string(#call) + ": " + hipGetErrorString(e_));                       \
    } while (0)

int round_up(int v, int m) { return (v + m - 1) / m * m; }

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
                const long long rounds = (tiles + (long long)kNumCU * wg_per_cu - 1) / ((long long)kNumCU * wg_per_cu);
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
                    g.tiles_x = tx; g.tiles_y = ty; g.zero_in = 0;
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
    static bool configured[64] = {}; // per instantiation and device: raise the dynamic-LDS cap once
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
                                  float *uo, float *vo, float alpha2)
{
    auto kern = write_v ? hsk::k_jacobi_classic_fused<NT, K, true> : hsk::k_jacobi_classic_fused<NT, K, false>;
    static bool configured[2][64] = {};
    if (p.lds_bytes > 32 * 1024 && !configured[write_v][c->device & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
        if (e != hipSuccess) return e;
        configured[write_v][c->device & 63] = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles), dim3(NT), p.lds_bytes, c->stream, c->dE[0], c->dE[1], c->dE[2], ui, vi, uo, vo, p.g, alpha2);
    return hipGetLastError();
}

hipError_t launch_classic_fused(const hsflow_ctx *c, const FusedPlan &p, bool write_v, const float *ui, const float *vi,
                                float *uo, float *vo, float alpha2)
{
#define HS_CASE(NT_, K_)                                                                          \
    if (p.NT == NT_ && p.K == K_) return launch_classic_fused_t<NT_, K_>(c, p, write_v, ui, vi, uo, vo, alpha2);
    HS_CASE(1024, 1) HS_CASE(1024, 2) HS_CASE(1024, 3)
    HS_CASE(512, 1) HS_CASE(512, 2) HS_CASE(512, 3) HS_CASE(512, 4)
    HS_CASE(256, 1) HS_CASE(256, 2) HS_CASE(256, 3) HS_CASE(256, 4)
#undef HS_CASE
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


double strip_launch_cost(int T, int R, int NW, long long tiles, int fold, double image_pixels, int *wg_per_cu_out = nullptr)
{
    // Parameters fitted (least squares on log time, rms 8 %) to profiles/r01_sweep_1080p_strip5.csv,
    // r01_sweep_4k_b.csv and r01_sweep_batch16.csv.  The sweep is VALU-issue bound (~34 instructions
    // per row per wavefront, ~4.2 cycles each per SIMD with 4 resident wavefronts, more with fewer);
    // a launch boundary costs ~2.7 us plus the L2 write-back of the 8 bytes per pixel just stored.
    const int per_simd = strip_max_waves(R, fold) / 4;
    const int lds = NW * (fold ? 4096 : 8192);
    const int wg_per_cu = std::max(1, std::min(std::min(kLdsLimit / lds, (per_simd * 4) / NW), 8));
    if (wg_per_cu_out) *wg_per_cu_out = wg_per_cu;
    const long long slots = (long long)kNumCU * wg_per_cu;
    const long long conc = std::min<long long>(wg_per_cu, (tiles + kNumCU - 1) / kNumCU); // WGs sharing a CU
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
    const double rounds = conc == 1 ? std::ceil(r) : std::max(1.0, r + 0.7);
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
            const double cost = strip_launch_cost(T, R, NW, tiles, fold, (double)W * H * c->N);
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
                g.tiles_x = tx; g.tiles_y = ty; g.zero_in = 0;
            }
        }
    }
    if (cost_out) *cost_out = best_cost;
    return found;
}

// Sweeps per launch for a budget of `iters` sweeps: minimise the modelled time of the whole solve
// (full launches of T plus one tail launch of iters % T).
int pick_strip_T(const hsflow_ctx *c, int iters, const hsflow_params &p, int fold)
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
    return bestT;
}

template <int R, int NTMAX, int EPS, bool FOLD> // EPS: 0 none, 1 every sweep, 2 witness (strip kernel only)
hipError_t launch_strip_t(const hsflow_ctx *c, const StripPlan &p, const float *ui, const float *vi,
                          float *uo, float *vo, float coeff, bool configure_only)
{
    auto kern = [] {
        if constexpr (FOLD) return hsk::k_jacobi_fold<R, NTMAX, EPS>;
        else return hsk::k_jacobi_strip<R, NTMAX, EPS>;
    }();
    static bool configured[64] = {};
    if (p.lds_bytes > 32 * 1024 && !configured[c->device & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
        if (e != hipSuccess) return e;
        configured[c->device & 63] = true;
    }
    if (configure_only) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(p.tiles), dim3(p.g.NW * 64), p.lds_bytes, c->stream, c->dCoef, ui, vi,
                       uo, vo, p.g, coeff, c->epsPtr, c->epsStride, p.tiles <= 65536 ? c->dStamps : nullptr, c->epsThr);
    return hipGetLastError();
}

template <int EPS, bool FOLD>
hipError_t launch_strip_e(const hsflow_ctx *c, const StripPlan &p, const float *ui, const float *vi,
                          float *uo, float *vo, float coeff, bool cfg)
{
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
}

bool make_jplan(const hsflow_ctx *c, int kind, int T, const hsflow_params &p, JPlan &out)
{
    out.kind = kind;
    out.T = T;
    if (kind == HSFLOW_KERNEL_STRIP) return make_strip_plan(c, T, p.strip_rows, p.threads, 0, out.s);
    if (kind == HSFLOW_KERNEL_FOLD) return make_strip_plan(c, T, p.strip_rows, p.threads, 1, out.s);
    return make_plan(c, T, p.tile_w, p.tile_h, p.threads, out.f);
}

// eps: 0 none, 1 Eps of every sweep, 2 witness (strip kernel only: one lower bound per launch)
hipError_t launch_j(const hsflow_ctx *c, const JPlan &pl, int eps, const float *ui, const float *vi,
                    float *uo, float *vo, float coeff, bool cfg = false, int zero_in = 0)
{
    if (pl.kind == HSFLOW_KERNEL_STRIP || pl.kind == HSFLOW_KERNEL_FOLD) {
        StripPlan sp = pl.s;
        sp.g.zero_in = zero_in;
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
                       c->W, c->H, c->P, c->plane, coeff, c->epsPtr)
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

// The graph cache is keyed by everything a captured launch sequence depends on (sizes, kernel shape,
// lambda, epsilon ...); a caller that varies those from call to call must not grow it without bound.
constexpr size_t kMaxGraphs = 32;
void trim_graph_cache(hsflow_ctx *c)
{
    if (c->graphs.size() < kMaxGraphs) return;
    hipStreamSynchronize(c->stream); // no replay of an old graph may still be running
    for (auto &kv : c->graphs) {
        if (kv.second.exec) hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) hipGraphDestroy(kv.second.graph);
    }
    c->graphs.clear();
}

int check_ctx(hsflow_ctx *c, int pair)
{
    if (!c) return fail(nullptr, HSFLOW_E_ARG, "null context");
    if (pair < 0 || pair >= c->N) return fail(c, HSFLOW_E_ARG, "pair index out of range");
    if (hipSetDevice(c->device) != hipSuccess) return fail(c, HSFLOW_E_DEVICE, "hipSetDevice failed");
    return HSFLOW_OK;
}

struct Profiler { // brackets kernels with events when params.profile is set
    hsflow_ctx *c;
    bool on;
    std::vector<std::pair<int, size_t>> marks; // (kind, index of start event); kind 0 deriv, 1 jacobi
    size_t used = 0;
    hipEvent_t ev(size_t i)
    {
        while (c->events.size() <= i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            c->events.push_back(e);
        }
        return c->events[i];
    }
    void begin(int kind)
    {
        if (!on) return;
        marks.push_back({kind, used});
        hipEventRecord(ev(used), c->stream);
        used++;
    }
    void end()
    {
        if (!on) return;
        hipEventRecord(ev(used), c->stream);
        used++;
    }
    void collect()
    {
        if (!on || marks.empty()) return;
        hipStreamSynchronize(c->stream);
        float d = 0, j = 0, t = 0;
        for (auto &m : marks) {
            float ms = 0;
            hipEventElapsedTime(&ms, c->events[m.second], c->events[m.second + 1]);
            (m.first == 0 ? d : j) += ms;
        }
        hipEventElapsedTime(&t, c->events[marks.front().second], c->events[used - 1]);
        c->info.deriv_ms = d;
        c->info.jacobi_ms = j;
        c->info.solve_ms = t;
    }
};

// Enqueue derivative pass + `iters` Jacobi sweeps (no host synchronisation inside).
int enqueue_fixed(hsflow_ctx *c, const hsflow_params &p, float coeff, int iters, int kernel, int T,
                  const JPlan *plan, const JPlan *tail_plan, Profiler &prof, bool do_deriv, bool zero_flow)
{
    // u = v = 0 at the start (reference behaviour, use_previous = 0): instead of clearing two
    // planes and reading them back, the first launch is told that its input is zero.
    int zero_in = zero_flow ? 1 : 0;
    if (zero_flow) c->cur = 0;
    if (do_deriv) {
        prof.begin(0);
        HS_HIP(c, launch_deriv(c));
        prof.end();
    }
    int left = iters, launches = 0;
    while (left > 0) {
        const int a = c->cur, b = a ^ 1;
        if (kernel == HSFLOW_KERNEL_SIMPLE) {
            prof.begin(1);
            HS_HIP(c, launch_simple(c, false, c->dU[a], c->dV[a], c->dU[b], c->dV[b], coeff, zero_in));
            prof.end();
            left -= 1;
        } else {
            const JPlan *pl = (left >= T) ? plan : tail_plan;
            prof.begin(1);
            HS_HIP(c, launch_j(c, *pl, false, c->dU[a], c->dV[a], c->dU[b], c->dV[b], coeff, false, zero_in));
            prof.end();
            left -= pl->T;
        }
        c->cur = b;
        zero_in = 0;
        launches++;
    }
    c->info.jacobi_launches = launches;
    (void)p;
    return HSFLOW_OK;
}

// Eps bookkeeping of an EPS-terminated solve: `sweeps` rows of `stride` words, cleared, plus the
// reduction of the rows into dEpsAll[0..sweeps).
// Buffers for `sweeps` Eps words of `stride` workgroups each (device) and their host copy; allocation
// only, so that what follows can be captured in a graph.
int eps_reserve(hsflow_ctx *c, int sweeps, int stride)
{
    const size_t need = (size_t)sweeps * stride;
    if (c->epsTilesCap < need) {
        hipFree(c->dEpsTiles);
        c->dEpsTiles = nullptr; c->epsTilesCap = 0;
        HS_HIP(c, hipMalloc((void **)&c->dEpsTiles, need * sizeof(unsigned)));
        c->epsTilesCap = need;
    }
    if (c->epsAllCap < sweeps) {
        hipFree(c->dEpsAll);
        c->dEpsAll = nullptr; c->epsAllCap = 0;
        HS_HIP(c, hipMalloc((void **)&c->dEpsAll, (size_t)sweeps * sizeof(unsigned)));
        c->epsAllCap = sweeps;
    }
    if (c->hEpsCap < (size_t)sweeps) {
        HS_HIP(c, hipStreamSynchronize(c->stream)); // nothing in flight may still write the old buffer
        if (c->hEps) hipHostFree(c->hEps);
        c->hEps = nullptr; c->hEpsCap = 0;
        const size_t cap = std::max<size_t>(256, (size_t)sweeps * 2);
        HS_HIP(c, hipHostMalloc((void **)&c->hEps, cap * sizeof(unsigned), hipHostMallocDefault));
        c->hEpsCap = cap;
    }
    c->epsStride = stride;
    return HSFLOW_OK;
}

int eps_clear(hsflow_ctx *c, int sweeps, int stride)
{
    HS_HIP(c, hipMemsetAsync(c->dEpsTiles, 0, (size_t)sweeps * stride * sizeof(unsigned), c->stream));
    return HSFLOW_OK;
}

int eps_prepare(hsflow_ctx *c, int sweeps, int stride)
{
    const int st = eps_reserve(c, sweeps, stride);
    return st ? st : eps_clear(c, sweeps, stride);
}

int eps_collect_enqueue(hsflow_ctx *c, int sweeps) // buffers from eps_reserve; nothing allocated here
{
    hipLaunchKernelGGL(hsk::k_eps_reduce, dim3(sweeps), dim3(256), 0, c->stream, c->dEpsTiles, c->epsStride, c->dEpsAll);
    HS_HIP(c, hipGetLastError());
    HS_HIP(c, hipMemcpyAsync(c->hEps, c->dEpsAll, (size_t)sweeps * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    c->epsPtr = c->dEps;
    c->epsStride = 1;
    return HSFLOW_OK;
}

int eps_collect(hsflow_ctx *c, int sweeps, std::vector<unsigned> &host)
{
    int st = eps_collect_enqueue(c, sweeps);
    if (st) return st;
    HS_HIP(c, hipStreamSynchronize(c->stream));
    host.assign(c->hEps, c->hEps + sweeps);
    return HSFLOW_OK;
}

// Witness slots of a speculative ITER|EPS pass (host copy): true if they prove that the early stop
// cannot have fired before the budget ran out; *last = Eps of the final sweep.
bool witness_proven(const unsigned *w, int slots, double epsilon, float *last)
{
    float e = 0.f;
    for (int i = 0; i < slots; i++) {
        std::memcpy(&e, &w[i], sizeof(float));
        if (!((double)e >= epsilon) && i != slots - 1) return false; // a stop at the very last sweep = the budget
    }
    *last = e;
    return true;
}

int plan_eps_stride(int kernel, const JPlan &pl) { return (kernel == HSFLOW_KERNEL_STRIP || kernel == HSFLOW_KERNEL_FOLD) ? pl.s.tiles : 1; }

// Diagnostic only: with HSFLOW_DEBUG_STAMPS=<file> every strip launch records per-workgroup phase
// stamps (8 x u64) and hsflow_solve appends those of the LAST launch to <file> as text.
constexpr int kStampTiles = 65536;
void dump_stamps(hsflow_ctx *c, int tiles)
{
    const char *path = getenv("HSFLOW_DEBUG_STAMPS");
    if (!path || !c->dStamps) return;
    tiles = std::min(tiles, kStampTiles);
    std::vector<unsigned long long> h((size_t)tiles * 8);
    if (hipMemcpy(h.data(), c->dStamps, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
    FILE *f = fopen(path, "a");
    if (!f) return;
    fprintf(f, "# solve tiles=%d T=%d R=%d threads=%d\n", tiles, c->info.fuse_steps, c->info.groups_per_thread, c->info.threads);
    for (int i = 0; i < tiles; i++) {
        const unsigned long long *o = &h[(size_t)i * 8];
        fprintf(f, "%d %llu %llu %llu %llu %llu %llu %llu\n", i, o[1] - o[0], o[2] - o[1], o[3] - o[2], o[3] - o[0],
                o[5] - o[4], o[6], o[7]);
    }
    fclose(f);
}

int pick_T(int max_iter, int requested)
{
    if (requested > 0) return std::min(requested, kMaxFuse);
    // default sweeps per launch; prefer a divisor of max_iter near 8 so that launches are uniform
    const int pref[] = {8, 10, 7, 9, 6, 12, 5, 4};
    for (int t : pref)
        if (max_iter % t == 0) return t;
    return std::min(8, std::max(1, max_iter));
}

// CLASSIC mode (Kernels.cl semantics, v restored): derivatives once, then max_iter fused
// average+update sweeps, one launch each.  The reference loop has no other stop rule
// (HSOpticalFlowOpenCL.cpp:750-751), so only ITER termination is accepted.
int solve_classic(hsflow_ctx *c, const hsflow_params &p, bool async)
{
    if (p.term_type != HSFLOW_TERM_ITER) return fail(c, HSFLOW_E_ARG, "CLASSIC mode supports ITER termination only");
    if (p.max_iter <= 0) return fail(c, HSFLOW_E_NOTERM, "ITER termination with max_iter <= 0 would never stop");
    if (!(p.alpha > 0.f) || !std::isfinite(p.alpha)) return fail(c, HSFLOW_E_ARG, "alpha must be positive");
    if (p.use_graph || (async && p.profile)) return fail(c, HSFLOW_E_ARG, "CLASSIC mode: use_graph / async profiling not supported");
    const size_t px = (size_t)c->plane * c->N;
    for (int i = 0; i < 3; i++)
        if (!c->dE[i]) HS_HIP(c, hipMalloc((void **)&c->dE[i], px * sizeof(float)));
    Profiler prof{c, p.profile != 0};
    const dim3 grid((c->W + 255) / 256, (c->H + 3) / 4, c->N), block(64, 4);
    if (!(p.reuse_derivatives && c->coef_valid && c->coef_mode == HSFLOW_MODE_CLASSIC)) {
        prof.begin(0);
        hipLaunchKernelGGL(hsk::k_deriv_classic, grid, block, 0, c->stream, c->dA, c->dB, c->dE[0], c->dE[1], c->dE[2],
                           c->W, c->H, c->P, c->plane);
        HS_HIP(c, hipGetLastError());
        prof.end();
    }
    c->coef_valid = true;
    c->coef_mode = HSFLOW_MODE_CLASSIC;
    const float a2 = p.alpha * p.alpha; // Kernels.cl:85
    const bool write_v = p.mode != HSFLOW_MODE_CLASSIC_AS_SHIPPED;
    int zero = p.use_previous ? 0 : 1;
    if (zero) c->cur = 0;
    if (p.kernel != HSFLOW_KERNEL_SIMPLE) {
        // several sweeps per launch on an LDS tile (k_jacobi_classic_fused)
        if (p.kernel != HSFLOW_KERNEL_AUTO && p.kernel != HSFLOW_KERNEL_FUSED)
            return fail(c, HSFLOW_E_ARG, "CLASSIC mode has the simple and the fused (LDS tile) kernels only");
        // 18 LDS values per plane and group and an IEEE division make a sweep dearer than in CV mode:
        // the halo pays off up to about 6 sweeps per launch (tools/sweep_classic.py on MI355X)
        const int T = p.fuse_steps > 0 ? std::min(p.fuse_steps, kMaxFuse) : std::min(6, p.max_iter);
        FusedPlan plan;
        if (!make_plan(c, T, p.tile_w, p.tile_h, p.threads, plan))
            return fail(c, HSFLOW_E_SIZE, "no feasible tile for the requested fuse_steps / tile / threads");
        int done = 0, launches = 0;
        while (done < p.max_iter) {
            const int chunk = std::min(T, p.max_iter - done);
            FusedPlan cp = plan;
            if (chunk != T && !make_plan(c, chunk, p.tile_w, p.tile_h, p.threads, cp))
                return fail(c, HSFLOW_E_SIZE, "no feasible tile for the tail launch");
            cp.g.zero_in = zero;
            const int a = c->cur, b = a ^ 1;
            prof.begin(1);
            hipError_t e = launch_classic_fused(c, cp, write_v, c->dU[a], c->dV[a], c->dU[b], c->dV[b], a2);
            prof.end();
            HS_HIP(c, e);
            c->cur = b;
            zero = 0;
            done += chunk;
            launches++;
        }
        hsflow_info &i = c->info;
        i.kernel = HSFLOW_KERNEL_FUSED; i.fuse_steps = T; i.tile_w = plan.g.CW; i.tile_h = plan.g.CH; i.threads = plan.NT;
        i.groups_per_thread = plan.K; i.tiles = plan.tiles; i.lds_bytes = plan.lds_bytes; i.jacobi_launches = launches;
        i.iterations_done = p.max_iter; i.last_eps = 0.f; i.deriv_ms = i.jacobi_ms = i.solve_ms = 0.f;
        if (!async) {
            HS_HIP(c, hipStreamSynchronize(c->stream));
            prof.collect();
        }
        return HSFLOW_OK;
    }
    for (int it = 0; it < p.max_iter; it++) {
        const int a = c->cur, b = a ^ 1;
        prof.begin(1);
        auto kern = zero ? (write_v ? hsk::k_jacobi_classic<true, true> : hsk::k_jacobi_classic<true, false>)
                         : (write_v ? hsk::k_jacobi_classic<false, true> : hsk::k_jacobi_classic<false, false>);
        hipLaunchKernelGGL(kern, grid, block, 0, c->stream, c->dE[0], c->dE[1], c->dE[2], c->dU[a], c->dV[a], c->dU[b], c->dV[b],
                           c->W, c->H, c->P, c->plane, a2);
        HS_HIP(c, hipGetLastError());
        prof.end();
        c->cur = b;
        zero = 0;
    }
    hsflow_info &i = c->info;
    i.kernel = HSFLOW_KERNEL_SIMPLE; i.fuse_steps = 1; i.tile_w = i.tile_h = 0; i.threads = 256;
    i.groups_per_thread = 1; i.tiles = 0; i.lds_bytes = 0; i.jacobi_launches = p.max_iter;
    i.iterations_done = p.max_iter; i.last_eps = 0.f; i.deriv_ms = i.jacobi_ms = i.solve_ms = 0.f;
    if (!async) {
        HS_HIP(c, hipStreamSynchronize(c->stream));
        prof.collect();
    }
    return HSFLOW_OK;
}

int solve_impl(hsflow_ctx *c, const hsflow_params *pp, bool async);

// Settles an ITER|EPS solve that hsflow_solve_async left unverified: waits for the stream, looks at the
// witness words and, if they do not prove "no early stop", runs the exact pass from the saved start.
int settle_pending(hsflow_ctx *c)
{
    if (!c->pend.active) return HSFLOW_OK;
    c->pend.active = false;
    HS_HIP(c, hipSetDevice(c->device));
    HS_HIP(c, hipStreamSynchronize(c->stream));
    float last = 0.f;
    if (witness_proven(c->hEps, c->pend.slots, c->pend.params.epsilon, &last)) {
        c->info.iterations_done = c->pend.iters;
        c->info.last_eps = last;
        return HSFLOW_OK;
    }
    hsflow_params q = c->pend.params;
    q.reuse_derivatives = 1; // the coefficient plane of that solve is still in place
    if (q.use_previous) {
        const size_t px = (size_t)c->plane * c->N;
        c->cur = c->pend.cur0;
        HS_HIP(c, hipMemcpyAsync(c->dU[c->cur], c->dUb, px * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        HS_HIP(c, hipMemcpyAsync(c->dV[c->cur], c->dVb, px * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    }
    c->force_exact = true;
    const int st = solve_impl(c, &q, false);
    c->force_exact = false;
    c->info.eps_rerun = 1;
    c->info.jacobi_launches += c->pend.launches;
    return st;
}

// Replays the hipGraph cached under `key`, capturing it first if needed.  `configure` sets kernel
// attributes (not allowed inside a capture), `enqueue` issues the launch sequence on c->stream and
// reports how many Jacobi launches it made.  On return c->cur is where the sequence leaves the flow.
template <class Configure, class Enqueue>
int run_captured(hsflow_ctx *c, const GraphKey &key, Configure configure, Enqueue enqueue, int *launches)
{
    if (!c->stream)
        return fail(c, HSFLOW_E_ARG, "use_graph: the default (NULL) stream cannot be captured; create the "
                                     "context on a non-default stream or with own_stream");
    auto it = c->graphs.find(key);
    if (it == c->graphs.end()) {
        int st = configure();
        if (st) return st;
        HS_HIP(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        const int cur0 = c->cur;
        int n = 0;
        st = enqueue(&n);
        hipGraph_t graph = nullptr;
        const hipError_t e = hipStreamEndCapture(c->stream, &graph);
        if (st) { if (graph) hipGraphDestroy(graph); c->cur = cur0; return st; }
        if (e != hipSuccess) { c->cur = cur0; return fail(c, HSFLOW_E_DEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(e)); }
        GraphEntry ge{};
        ge.graph = graph;
        ge.cur_after = c->cur;
        ge.launches = n;
        HS_HIP(c, hipGraphInstantiate(&ge.exec, graph, nullptr, nullptr, 0));
        trim_graph_cache(c);
        it = c->graphs.emplace(key, ge).first;
    }
    HS_HIP(c, hipGraphLaunch(it->second.exec, c->stream));
    c->cur = it->second.cur_after;
    *launches = it->second.launches;
    return HSFLOW_OK;
}

// What solve_impl works out once and the three termination paths share.
struct SolveSetup {
    float coeff;        // Ilambda = fl32(1 / fl32(lambda))
    int kernel;         // kernel actually used (AUTO resolved)
    bool multi;         // several sweeps per launch (every kernel but the simple one)
    bool use_iter, use_eps;
    long long budget;   // sweep budget (huge when ITER does not apply)
    int T;              // sweeps per full launch
    JPlan plan;         // launch plan for T sweeps
};

// ITER termination: a fixed sweep count, nothing on the host between launches (optionally one hipGraph).
int solve_fixed(hsflow_ctx *c, const hsflow_params &p, const SolveSetup &S, Profiler &prof, bool async)
{
    const float coeff = S.coeff;
    const int kernel = S.kernel, T = S.T;
    const bool multi = S.multi;
    const JPlan &plan = S.plan;
    int st = HSFLOW_OK;
    const long long budget = S.budget;
    JPlan tail;
    const int iters = (int)budget;
    const int rem = multi ? iters % T : 0;
    if (rem && !make_jplan(c, kernel, rem, p, tail))
        return fail(c, HSFLOW_E_SIZE, "no feasible launch plan for the tail launch");
    const bool zero = !p.use_previous;
    const bool do_deriv = !(p.reuse_derivatives && c->coef_valid && c->coef_mode == HSFLOW_MODE_CV);
    if (p.use_graph && !p.profile) {
        GraphKey key{p.mode, kernel, iters, T, c->info.tile_w, c->info.tile_h, c->info.threads,
                     c->info.groups_per_thread, zero ? 0 : c->cur, p.use_previous * 2 + (do_deriv ? 1 : 0), coeff};
        auto configure = [&]() -> int {
            if (multi) {
                HS_HIP(c, launch_j(c, plan, false, nullptr, nullptr, nullptr, nullptr, coeff, true));
                if (rem) HS_HIP(c, launch_j(c, tail, false, nullptr, nullptr, nullptr, nullptr, coeff, true));
            }
            return HSFLOW_OK;
        };
        auto enqueue = [&](int *n) -> int {
            const int e = enqueue_fixed(c, p, coeff, iters, kernel, T, &plan, &tail, prof, do_deriv, zero);
            *n = c->info.jacobi_launches;
            return e;
        };
        int n = 0;
        if ((st = run_captured(c, key, configure, enqueue, &n))) return st;
        c->info.jacobi_launches = n;
    } else {
        st = enqueue_fixed(c, p, coeff, iters, kernel, T, &plan, &tail, prof, do_deriv, zero);
        if (st) return st;
    }
    c->coef_valid = true;
    c->coef_mode = HSFLOW_MODE_CV;
    c->info.iterations_done = iters;
    if (!async) {
        HS_HIP(c, hipStreamSynchronize(c->stream));
        prof.collect();
        if (c->dStamps && (kernel == HSFLOW_KERNEL_STRIP || kernel == HSFLOW_KERNEL_FOLD)) dump_stamps(c, plan.s.tiles);
    }
    return HSFLOW_OK;
}

// ITER|EPS -- the way the reference calls the solver (OpticalFlowOpenCV.cpp:29).  On real image
// pairs Eps never drops below 1e-6 within the sweep budget, so the budget is run SPECULATIVELY at
// full speed (no host round trip between launches) while every sweep records its Eps on the device;
// one read-back at the end finds the first sweep k with Eps_k < epsilon.  If there is none the
// result stands; otherwise exactly k sweeps are re-run from the saved starting flow, which
// reproduces the oracle's stopping sweep.
int solve_iter_eps(hsflow_ctx *c, const hsflow_params &p, const SolveSetup &S, Profiler &prof, bool async)
{
    const float coeff = S.coeff;
    const int kernel = S.kernel, T = S.T;
    const bool multi = S.multi;
    const JPlan &plan = S.plan;
    int st = HSFLOW_OK;
    const long long budget = S.budget;
    const int iters = (int)budget;
    const size_t px = (size_t)c->plane * c->N;
    const bool witness = (kernel == HSFLOW_KERNEL_STRIP || kernel == HSFLOW_KERNEL_FOLD) && !c->force_exact;
    const bool do_deriv = !(p.reuse_derivatives && c->coef_valid && c->coef_mode == HSFLOW_MODE_CV);
    if (p.use_previous) { // the starting flow is kept: the ping-pong buffers get overwritten
        if (!c->dUb) HS_HIP(c, hipMalloc((void **)&c->dUb, px * sizeof(float)));
        if (!c->dVb) HS_HIP(c, hipMalloc((void **)&c->dVb, px * sizeof(float)));
    }
    auto save_start = [&]() -> int {
        if (!p.use_previous) return HSFLOW_OK;
        HS_HIP(c, hipMemcpyAsync(c->dUb, c->dU[c->cur], px * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        HS_HIP(c, hipMemcpyAsync(c->dVb, c->dV[c->cur], px * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        return HSFLOW_OK;
    };
    // every launch of this solve uses the same number of workgroups or fewer (tail): stride = max
    int stride = multi ? plan_eps_stride(kernel, plan) : 1;
    JPlan tailp;
    const bool has_tail = multi && iters % T;
    if (has_tail) {
        if (!make_jplan(c, kernel, iters % T, p, tailp)) return fail(c, HSFLOW_E_SIZE, "no feasible launch plan for a chunk");
        stride = std::max(stride, plan_eps_stride(kernel, tailp));
    }
    int launches = 0;
    if (witness) {
        // Witness pass: all launches but the last run k_jacobi_strip<.., 2>, which costs almost
        // nothing over the ITER-only kernel and yields one number per launch that proves "Eps >=
        // epsilon in every one of my sweeps" when it is >= epsilon.  The last launch measures every
        // sweep (it also provides last_eps).  If every bound holds and no sweep of the last launch
        // but possibly its final one fell below epsilon, the early stop cannot have fired before the
        // budget ran out and the result stands.  Otherwise (a flat or converged input) the exact
        // per-sweep path below starts over from the saved flow.
        // threshold as the smallest float >= epsilon: "change >= epsThr" then implies "Eps >= epsilon"
        c->epsThr = p.epsilon > 0 ? (float)p.epsilon : 0.f;
        if ((double)c->epsThr < p.epsilon) c->epsThr = std::nextafterf(c->epsThr, INFINITY);
        const int n_launch = (iters + T - 1) / T;
        const int last_chunk = iters - (n_launch - 1) * T;
        const int slots = (n_launch - 1) + last_chunk;
        if ((st = eps_reserve(c, slots, stride))) return st;
        const int cur0 = c->cur;
        // the whole pass as one enqueue sequence (nothing allocated, nothing synchronised: capturable)
        auto enqueue = [&]() -> int {
            int e0 = save_start();
            if (e0) return e0;
            c->epsStride = stride;
            if ((e0 = eps_clear(c, slots, stride))) return e0;
            if (do_deriv) {
                prof.begin(0);
                HS_HIP(c, launch_deriv(c));
                prof.end();
            }
            int zero_w = p.use_previous ? 0 : 1;
            if (zero_w) c->cur = 0;
            for (int L = 0; L < n_launch; L++) {
                const bool is_last = L == n_launch - 1;
                const JPlan &cp = (is_last && last_chunk != T) ? tailp : plan;
                const int a = c->cur, b = a ^ 1;
                c->epsPtr = c->dEpsTiles + (size_t)L * stride;
                prof.begin(1);
                hipError_t e = launch_j(c, cp, is_last ? 1 : 2, c->dU[a], c->dV[a], c->dU[b], c->dV[b], coeff, false, zero_w);
                prof.end();
                HS_HIP(c, e);
                c->cur = b;
                zero_w = 0;
                launches++;
            }
            return eps_collect_enqueue(c, slots);
        };
        if (p.use_graph && !p.profile) {
            GraphKey key{p.mode, kernel, iters, T, c->info.tile_w, c->info.tile_h, c->info.threads,
                         c->info.groups_per_thread, p.use_previous ? c->cur : 0, p.use_previous * 2 + (do_deriv ? 1 : 0), coeff, c->epsThr};
            auto configure = [&]() -> int {
                HS_HIP(c, launch_j(c, plan, 2, nullptr, nullptr, nullptr, nullptr, coeff, true));
                HS_HIP(c, launch_j(c, plan, 1, nullptr, nullptr, nullptr, nullptr, coeff, true));
                if (has_tail) HS_HIP(c, launch_j(c, tailp, 1, nullptr, nullptr, nullptr, nullptr, coeff, true));
                return HSFLOW_OK;
            };
            auto enqueue_n = [&](int *n) -> int {
                launches = 0;
                const int e = enqueue();
                *n = launches;
                return e;
            };
            if ((st = run_captured(c, key, configure, enqueue_n, &launches))) return st;
            c->epsPtr = c->dEps;
            c->epsStride = 1;
        } else if ((st = enqueue())) {
            return st;
        }
        c->coef_valid = true;
        c->coef_mode = HSFLOW_MODE_CV;
        if (async) { // the check is owed: hsflow_synchronize (or the next call that needs results) settles it
            c->pend.active = true;
            c->pend.params = p;
            c->pend.iters = iters; c->pend.slots = slots; c->pend.launches = launches; c->pend.cur0 = cur0;
            c->info.iterations_done = iters;
            c->info.jacobi_launches = launches;
            return HSFLOW_OK;
        }
        HS_HIP(c, hipStreamSynchronize(c->stream));
        std::vector<unsigned> hw(c->hEps, c->hEps + slots);

        float last = 0.f;
        if (witness_proven(hw.data(), slots, p.epsilon, &last)) {
            c->info.iterations_done = iters;
            c->info.last_eps = last;
            c->info.jacobi_launches = launches;
            prof.collect();
            return HSFLOW_OK;
        }
        c->info.eps_rerun = 1;
        // not proven: restore the starting flow and measure every sweep
        if (p.use_previous) {
            c->cur = cur0;
            HS_HIP(c, hipMemcpyAsync(c->dU[c->cur], c->dUb, px * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            HS_HIP(c, hipMemcpyAsync(c->dV[c->cur], c->dVb, px * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        }
        if ((st = eps_prepare(c, iters, stride))) return st;
    } else {
        if ((st = save_start())) return st;
        if ((st = eps_prepare(c, iters, stride))) return st;
        if (do_deriv) {
            prof.begin(0);
            HS_HIP(c, launch_deriv(c));
            prof.end();
        }
        c->coef_valid = true;
        c->coef_mode = HSFLOW_MODE_CV;
    }
    int zero_in = p.use_previous ? 0 : 1, done = 0;
    if (zero_in) c->cur = 0;
    while (done < iters) {
        const int chunk = multi ? std::min(T, iters - done) : 1;
        JPlan cp = plan;
        if (multi && chunk != T && !make_jplan(c, kernel, chunk, p, cp))
            return fail(c, HSFLOW_E_SIZE, "no feasible launch plan for a chunk");
        const int a = c->cur, b = a ^ 1;
        c->epsPtr = c->dEpsTiles + (size_t)done * stride;
        prof.begin(1);
        hipError_t e = multi ? launch_j(c, cp, true, c->dU[a], c->dV[a], c->dU[b], c->dV[b], coeff, false, zero_in)
                             : launch_simple(c, true, c->dU[a], c->dV[a], c->dU[b], c->dV[b], coeff, zero_in);
        prof.end();
        HS_HIP(c, e);
        c->cur = b;
        zero_in = 0;
        done += chunk;
        launches++;
    }
    std::vector<unsigned> heps;
    if ((st = eps_collect(c, iters, heps))) return st;
    int hit = -1;
    float last = 0.f;
    for (int s2 = 0; s2 < iters; s2++) {
        std::memcpy(&last, &heps[(size_t)s2], sizeof(float));
        if ((double)last < p.epsilon) { hit = s2; break; }
    }
    if (hit >= 0 && hit + 1 < iters) { // converged early: redo exactly hit+1 sweeps from the start
        const int k = hit + 1;
        if (p.use_previous) {
            HS_HIP(c, hipMemcpyAsync(c->dU[c->cur], c->dUb, px * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            HS_HIP(c, hipMemcpyAsync(c->dV[c->cur], c->dVb, px * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        }
        JPlan kp, kt;
        int Tk = 1;
        if (multi) {
            Tk = std::min(T, k);
            if (!make_jplan(c, kernel, Tk, p, kp)) return fail(c, HSFLOW_E_SIZE, "no feasible launch plan for the re-run");
            if (k % Tk && !make_jplan(c, kernel, k % Tk, p, kt)) return fail(c, HSFLOW_E_SIZE, "no feasible launch plan for the re-run tail");
        }
        st = enqueue_fixed(c, p, coeff, k, kernel, Tk, &kp, &kt, prof, false, !p.use_previous);
        if (st) return st;
        launches += c->info.jacobi_launches;
        HS_HIP(c, hipStreamSynchronize(c->stream));
        c->info.iterations_done = k;
    } else {
        c->info.iterations_done = hit >= 0 ? hit + 1 : iters;
    }
    c->info.last_eps = last;
    c->info.jacobi_launches = launches;
    prof.collect();
    return HSFLOW_OK;
}

// EPS without a usable sweep budget (CV_TERMCRIT_EPS alone): Eps_k = max |u_k - u_{k-1}|, |v_k - v_{k-1}|
// is produced per sweep by the kernel; the host looks at it after every chunk and, if the
// threshold was crossed inside the chunk, replays the chunk up to that sweep (its input buffer
// is still intact), which reproduces the oracle's stopping sweep exactly.
int solve_eps_chunks(hsflow_ctx *c, const hsflow_params &p, const SolveSetup &S, Profiler &prof)
{
    const float coeff = S.coeff;
    const int kernel = S.kernel, T = S.T;
    const bool multi = S.multi;
    const JPlan &plan = S.plan;
    int st = HSFLOW_OK;
    const long long budget = S.budget;
    const bool use_iter = S.use_iter;
    if (!p.use_previous) {
        c->cur = 0;
        HS_HIP(c, hipMemsetAsync(c->dU[0], 0, (size_t)c->plane * c->N * sizeof(float), c->stream));
        HS_HIP(c, hipMemsetAsync(c->dV[0], 0, (size_t)c->plane * c->N * sizeof(float), c->stream));
    }
    if (!(p.reuse_derivatives && c->coef_valid && c->coef_mode == HSFLOW_MODE_CV)) {
        prof.begin(0);
        HS_HIP(c, launch_deriv(c));
        prof.end();
    }
    c->coef_valid = true;
    c->coef_mode = HSFLOW_MODE_CV;
    long long done = 0;
    int launches = 0;
    float last = 0.f;
    bool stop = false;
    while (!stop) {
        const int chunk = (int)std::min<long long>(T, budget - done);
        JPlan cp = plan;
        if (multi && chunk != T && !make_jplan(c, kernel, chunk, p, cp))
            return fail(c, HSFLOW_E_SIZE, "no feasible launch plan for a chunk");
        const int a = c->cur, b = a ^ 1;
        const int n = multi ? chunk : 1;
        if ((st = eps_prepare(c, n, multi ? plan_eps_stride(kernel, cp) : 1))) return st;
        c->epsPtr = c->dEpsTiles;
        prof.begin(1);
        if (!multi)
            HS_HIP(c, launch_simple(c, true, c->dU[a], c->dV[a], c->dU[b], c->dV[b], coeff));
        else
            HS_HIP(c, launch_j(c, cp, true, c->dU[a], c->dV[a], c->dU[b], c->dV[b], coeff));
        prof.end();
        launches++;
        std::vector<unsigned> heps;
        if ((st = eps_collect(c, n, heps))) return st;
        int hit = -1;
        for (int s = 0; s < n; s++) {
            float e;
            std::memcpy(&e, &heps[(size_t)s], sizeof(float));
            last = e;
            if ((double)e < p.epsilon) { hit = s; break; }
        }
        if (hit >= 0 && hit < n - 1) { // crossed inside the chunk: redo exactly hit+1 sweeps
            JPlan rp;
            if (!make_jplan(c, kernel, hit + 1, p, rp))
                return fail(c, HSFLOW_E_SIZE, "no feasible launch plan for the replay");
            prof.begin(1);
            HS_HIP(c, launch_j(c, rp, false, c->dU[a], c->dV[a], c->dU[b], c->dV[b], coeff));
            prof.end();
            launches++;
            done += hit + 1;
            stop = true;
        } else {
            done += n;
            if (hit >= 0) stop = true;
        }
        c->cur = b;
        if (use_iter && p.max_iter > 0 && done >= budget) stop = true;
    }
    HS_HIP(c, hipStreamSynchronize(c->stream));
    c->info.iterations_done = (int)done;
    c->info.last_eps = last;
    c->info.jacobi_launches = launches;
    prof.collect();
    return HSFLOW_OK;
}

int solve_impl(hsflow_ctx *c, const hsflow_params *pp, bool async)
{
    int st = check_ctx(c, 0);
    if (st) return st;
    if ((st = settle_pending(c))) return st; // an unverified asynchronous solve comes first
    if (!pp || pp->struct_size != sizeof(hsflow_params))
        return fail(c, HSFLOW_E_ARG, "params null or struct_size mismatch");
    const hsflow_params &p = *pp;
    if (!c->frames_set) return fail(c, HSFLOW_E_STATE, "frames were not set");
    if (p.mode == HSFLOW_MODE_CLASSIC || p.mode == HSFLOW_MODE_CLASSIC_AS_SHIPPED) return solve_classic(c, p, async);
    if (p.mode != HSFLOW_MODE_CV) return fail(c, HSFLOW_E_ARG, "unknown mode");
    const bool use_iter = (p.term_type & HSFLOW_TERM_ITER) != 0, use_eps = (p.term_type & HSFLOW_TERM_EPS) != 0;
    if (!use_iter && !use_eps) return fail(c, HSFLOW_E_ARG, "term_type must include ITER and/or EPS");
    if (use_iter && p.max_iter <= 0 && !use_eps)
        return fail(c, HSFLOW_E_NOTERM, "ITER termination with max_iter <= 0 would never stop");
    if (!(p.lambda > 0.f) || !std::isfinite(p.lambda)) return fail(c, HSFLOW_E_ARG, "lambda must be positive");
    if (async && p.profile) return fail(c, HSFLOW_E_ARG, "solve_async does not support profiling");
    c->info.eps_rerun = 0;

    const float coeff = 1.0f / p.lambda; // Ilambda = fl32(1/fl32(lambda)), cv210.dll VA 0x1012e054-0x1012e085
    // AUTO: the register-strip kernel; below ~1.5 Mpixel per context its folded form (128-column strips:
    // twice the tiles across, so small frames reach more CUs -- measured 5-25 % faster from 160x120 to
    // 1600x900 at 100 sweeps, tools/crossover.py).
    const bool small_frame = (long long)c->W * c->H * c->N <= 1500000LL;
    const int kernel = p.kernel != HSFLOW_KERNEL_AUTO ? p.kernel : (small_frame ? HSFLOW_KERNEL_FOLD : HSFLOW_KERNEL_STRIP);
    if (kernel != HSFLOW_KERNEL_SIMPLE && kernel != HSFLOW_KERNEL_FUSED && kernel != HSFLOW_KERNEL_STRIP &&
        kernel != HSFLOW_KERNEL_FOLD)
        return fail(c, HSFLOW_E_ARG, "unknown kernel selector");
    const bool multi = kernel != HSFLOW_KERNEL_SIMPLE;
    if (async && use_eps && !(use_iter && p.max_iter > 0 && p.max_iter <= (1 << 16) &&
                              (kernel == HSFLOW_KERNEL_STRIP || kernel == HSFLOW_KERNEL_FOLD) && !c->force_exact))
        return fail(c, HSFLOW_E_ARG, "solve_async with EPS termination needs ITER|EPS with a sweep budget and the strip / fold kernel "
                                     "(ITER-only termination works with every kernel)");
    // With ITER the sweep budget is max_iter (a budget <= 0 with EPS never triggers ITER);
    // EPS-only runs use chunks until Eps < epsilon.
    const long long budget = (use_iter && p.max_iter > 0) ? p.max_iter : (1LL << 40);

    int T = 1;
    JPlan plan;
    if (multi) {
        const int horizon = budget > (1 << 30) ? 64 : (int)budget; // EPS-only runs: plan for chunks
        if (p.fuse_steps > 0) T = std::min(p.fuse_steps, kMaxFuse);
        else if (kernel == HSFLOW_KERNEL_STRIP || kernel == HSFLOW_KERNEL_FOLD)
            T = (use_eps && !(use_iter && p.max_iter > 0)) ? std::min(8, horizon)
                                                          : pick_strip_T(c, horizon, p, kernel == HSFLOW_KERNEL_FOLD);
        else T = pick_T(horizon, 0);
        if (budget < T) T = (int)budget;
        if (!make_jplan(c, kernel, T, p, plan))
            return fail(c, HSFLOW_E_SIZE, "no feasible launch plan for the requested tile/threads/rows/fuse_steps");
        plan_to_info(c, plan);
    } else {
        c->info.fuse_steps = 1; c->info.tile_w = c->info.tile_h = 0; c->info.threads = 256;
        c->info.groups_per_thread = 1; c->info.tiles = 0; c->info.lds_bytes = 0;
    }
    c->info.kernel = kernel;
    c->info.deriv_ms = c->info.jacobi_ms = c->info.solve_ms = 0.f;
    c->info.last_eps = 0.f;
    Profiler prof{c, p.profile != 0};

    SolveSetup S{coeff, kernel, multi, use_iter, use_eps, budget, T, plan};
    if (!use_eps) return solve_fixed(c, p, S, prof, async);
    constexpr long long kSpecMax = 1 << 16; // speculative ITER|EPS: the whole budget in one go
    if (use_iter && p.max_iter > 0 && budget <= kSpecMax) return solve_iter_eps(c, p, S, prof, async);
    return solve_eps_chunks(c, p, S, prof);
}

int copy_frame_in(hsflow_ctx *c, uint8_t *dst, const void *src, size_t stride, hipMemcpyKind kind, bool sync)
{
    if (sync) HS_HIP(c, hipMemcpy2D(dst, c->P, src, stride, c->W, c->H, kind));
    else HS_HIP(c, hipMemcpy2DAsync(dst, c->P, src, stride, c->W, c->H, kind, c->stream));
    return HSFLOW_OK;
}

} // namespace

extern "C" {

void hsflow_default_params(hsflow_params *p)
{
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->struct_size = sizeof(*p);
    p->mode = HSFLOW_MODE_CV;
    p->lambda = 1.0f;
    p->alpha = 1.0f;
    p->term_type = HSFLOW_TERM_ITER | HSFLOW_TERM_EPS; // as the reference calls it, OpticalFlowOpenCV.cpp:29
    p->max_iter = 100;                                  // main.cpp:4
    p->epsilon = (double)1e-6f;                         // cvTermCriteria rounds through float, cxtypes.h:912
    p->kernel = HSFLOW_KERNEL_AUTO;
}

int hsflow_device_count(int *count)
{
    if (!count) return HSFLOW_E_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return HSFLOW_E_DEVICE; }
    *count = n;
    return HSFLOW_OK;
}

int hsflow_host_alloc(void **out, size_t bytes)
{
    if (!out || !bytes) return fail(nullptr, HSFLOW_E_ARG, "hsflow_host_alloc: null out or zero size");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { *out = nullptr; return fail(nullptr, HSFLOW_E_OOM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
    return HSFLOW_OK;
}

int hsflow_host_free(void *p)
{
    if (!p) return HSFLOW_OK;
    hipError_t e = hipHostFree(p);
    return e == hipSuccess ? HSFLOW_OK : fail(nullptr, HSFLOW_E_ARG, std::string("hipHostFree: ") + hipGetErrorString(e));
}

int hsflow_host_register(void *p, size_t bytes)
{
    if (!p || !bytes) return fail(nullptr, HSFLOW_E_ARG, "hsflow_host_register: null pointer or zero size");
    hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    return e == hipSuccess ? HSFLOW_OK : fail(nullptr, HSFLOW_E_DEVICE, std::string("hipHostRegister: ") + hipGetErrorString(e));
}

int hsflow_host_unregister(void *p)
{
    if (!p) return HSFLOW_OK;
    hipError_t e = hipHostUnregister(p);
    return e == hipSuccess ? HSFLOW_OK : fail(nullptr, HSFLOW_E_ARG, std::string("hipHostUnregister: ") + hipGetErrorString(e));
}

int hsflow_version(void) { return HSFLOW_VERSION_MAJOR * 1000 + HSFLOW_VERSION_MINOR; }

const char *hsflow_status_string(int s)
{
    switch (s) {
    case HSFLOW_OK: return "ok";
    case HSFLOW_E_ARG: return "invalid argument";
    case HSFLOW_E_SIZE: return "invalid size or stride";
    case HSFLOW_E_DEVICE: return "device error";
    case HSFLOW_E_OOM: return "out of memory";
    case HSFLOW_E_STATE: return "invalid call order";
    case HSFLOW_E_NOTERM: return "termination criteria never met";
    default: return "unknown status";
    }
}

const char *hsflow_last_error(hsflow_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int hsflow_create(hsflow_ctx **out, int device, int width, int height, int n_pairs, void *stream, int own_stream)
{
    if (!out) return fail(nullptr, HSFLOW_E_ARG, "out is null");
    *out = nullptr;
    if (width <= 0 || height <= 0 || n_pairs <= 0) return fail(nullptr, HSFLOW_E_SIZE, "width, height, n_pairs must be positive");
    if ((long long)round_up(width, 64) * height > (1LL << 30)) return fail(nullptr, HSFLOW_E_SIZE, "plane too large (pitch*height > 2^30)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, HSFLOW_E_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(nullptr, HSFLOW_E_ARG, "device ordinal out of range");
    hsflow_ctx *c = new (std::nothrow) hsflow_ctx();
    if (!c) return fail(nullptr, HSFLOW_E_OOM, "host allocation failed");
    c->device = device; c->W = width; c->H = height; c->N = n_pairs;
    c->P = round_up(width, 64);
    c->plane = (long long)c->P * height;
    std::memset(&c->info, 0, sizeof(c->info));
    c->info.struct_size = sizeof(hsflow_info);
    c->info.width = width; c->info.height = height; c->info.n_pairs = n_pairs; c->info.pitch = c->P;
    auto bail = [&](int code, const std::string &m) { g_create_error = m; hsflow_destroy(c); return code; };
#define HS_TRY(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return bail(e_ == hipErrorOutOfMemory ? HSFLOW_E_OOM : HSFLOW_E_DEVICE,               \
                        std::string(#call) + ": " + hipGetErrorString(e_));                       \
    } while (0)
    HS_TRY(hipSetDevice(device));
    if (own_stream) {
        HS_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    } else {
        c->stream = (hipStream_t)stream;
    }
    const size_t px = (size_t)c->plane * n_pairs;
    HS_TRY(hipMalloc((void **)&c->dA, px));
    HS_TRY(hipMalloc((void **)&c->dB, px));
    HS_TRY(hipMalloc((void **)&c->dCoef, px * sizeof(uint32_t)));
    for (int i = 0; i < 2; i++) {
        HS_TRY(hipMalloc((void **)&c->dU[i], px * sizeof(float)));
        HS_TRY(hipMalloc((void **)&c->dV[i], px * sizeof(float)));
    }
    HS_TRY(hipMalloc((void **)&c->dEps, kMaxFuse * sizeof(unsigned)));
    c->epsPtr = c->dEps;
    if (getenv("HSFLOW_DEBUG_STAMPS")) HS_TRY(hipMalloc((void **)&c->dStamps, (size_t)kStampTiles * 8 * sizeof(unsigned long long)));
    // deterministic contents for padding columns and the initial flow
    HS_TRY(hipMemsetAsync(c->dA, 0, px, c->stream));
    HS_TRY(hipMemsetAsync(c->dB, 0, px, c->stream));
    HS_TRY(hipMemsetAsync(c->dCoef, 0, px * sizeof(uint32_t), c->stream));
    for (int i = 0; i < 2; i++) {
        HS_TRY(hipMemsetAsync(c->dU[i], 0, px * sizeof(float), c->stream));
        HS_TRY(hipMemsetAsync(c->dV[i], 0, px * sizeof(float), c->stream));
    }
    HS_TRY(hipStreamSynchronize(c->stream));
#undef HS_TRY
    *out = c;
    return HSFLOW_OK;
}

int hsflow_destroy(hsflow_ctx *c)
{
    if (!c) return HSFLOW_OK;
    hipSetDevice(c->device);
    if (c->stream || !c->own_stream) hipStreamSynchronize(c->stream);
    for (auto &kv : c->graphs) {
        if (kv.second.exec) hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) hipGraphDestroy(kv.second.graph);
    }
    for (hipEvent_t e : c->events) hipEventDestroy(e);
    hipFree(c->dA); hipFree(c->dB); hipFree(c->dCoef);
    for (int i = 0; i < 3; i++) hipFree(c->dE[i]);
    for (int i = 0; i < 2; i++) { hipFree(c->dU[i]); hipFree(c->dV[i]); }
    hipFree(c->dEps);
    hipFree(c->dEpsAll); hipFree(c->dEpsTiles); hipFree(c->dUb); hipFree(c->dVb);
    hipFree(c->dStamps);
    hipFree(c->dScratch);
    if (c->hEps) hipHostFree(c->hEps);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
    return HSFLOW_OK;
}

int hsflow_set_frames_u8(hsflow_ctx *c, int pair, const uint8_t *prev, size_t ps, const uint8_t *curr, size_t cs)
{
    int st = check_ctx(c, pair);
    if (st) return st;
    if ((st = settle_pending(c))) return st; // an unverified asynchronous solve still needs the old inputs
    if (!prev || !curr) return fail(c, HSFLOW_E_ARG, "null frame pointer");
    if (ps < (size_t)c->W || cs < (size_t)c->W) return fail(c, HSFLOW_E_SIZE, "frame stride smaller than width");
    HS_HIP(c, hipStreamSynchronize(c->stream));
    if ((st = copy_frame_in(c, c->dA + pair * c->plane, prev, ps, hipMemcpyHostToDevice, true))) return st;
    if ((st = copy_frame_in(c, c->dB + pair * c->plane, curr, cs, hipMemcpyHostToDevice, true))) return st;
    c->frames_set = true;
    c->coef_valid = false;
    return HSFLOW_OK;
}

// One transfer of `rows` rows of `rowb` bytes between a pitched device plane and a host buffer on
// ctx's stream.  Dense layouts on both sides take the 1-D path (a single SDMA copy).
static hipError_t copy_rows_async(hsflow_ctx *c, void *dst, size_t dpitch, const void *src, size_t spitch, size_t rowb,
                                  size_t rows, hipMemcpyKind kind)
{
    if (dpitch == rowb && spitch == rowb) return hipMemcpyAsync(dst, src, rowb * rows, kind, c->stream);
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, rowb, rows, kind, c->stream);
}

int hsflow_set_frames_u8_async(hsflow_ctx *c, int pair, const uint8_t *prev, size_t ps, const uint8_t *curr, size_t cs)
{
    int st = check_ctx(c, pair);
    if (st) return st;
    if ((st = settle_pending(c))) return st; // an unverified asynchronous solve still needs the old inputs
    if (!prev || !curr) return fail(c, HSFLOW_E_ARG, "null frame pointer");
    if (ps < (size_t)c->W || cs < (size_t)c->W) return fail(c, HSFLOW_E_SIZE, "frame stride smaller than width");
    HS_HIP(c, copy_rows_async(c, c->dA + pair * c->plane, c->P, prev, ps, c->W, c->H, hipMemcpyHostToDevice));
    HS_HIP(c, copy_rows_async(c, c->dB + pair * c->plane, c->P, curr, cs, c->W, c->H, hipMemcpyHostToDevice));
    c->frames_set = true;
    c->coef_valid = false;
    return HSFLOW_OK;
}

int hsflow_set_frames_u8_device(hsflow_ctx *c, int pair, const void *dprev, size_t ps, const void *dcurr, size_t cs)
{
    int st = check_ctx(c, pair);
    if (st) return st;
    if ((st = settle_pending(c))) return st; // an unverified asynchronous solve still needs the old inputs
    if (!dprev || !dcurr) return fail(c, HSFLOW_E_ARG, "null frame pointer");
    if (ps < (size_t)c->W || cs < (size_t)c->W) return fail(c, HSFLOW_E_SIZE, "frame stride smaller than width");
    if ((st = copy_frame_in(c, c->dA + pair * c->plane, dprev, ps, hipMemcpyDeviceToDevice, false))) return st;
    if ((st = copy_frame_in(c, c->dB + pair * c->plane, dcurr, cs, hipMemcpyDeviceToDevice, false))) return st;
    c->frames_set = true;
    c->coef_valid = false;
    return HSFLOW_OK;
}

int hsflow_push_frame_u8(hsflow_ctx *c, int pair, const uint8_t *next, size_t ns)
{
    int st = check_ctx(c, pair);
    if (st) return st;
    if ((st = settle_pending(c))) return st; // an unverified asynchronous solve still needs the old inputs
    if (!next) return fail(c, HSFLOW_E_ARG, "null frame pointer");
    if (ns < (size_t)c->W) return fail(c, HSFLOW_E_SIZE, "frame stride smaller than width");
    if (!c->frames_set) return fail(c, HSFLOW_E_STATE, "push_frame needs a previous pair");
    HS_HIP(c, hipMemcpyAsync(c->dA + pair * c->plane, c->dB + pair * c->plane, (size_t)c->plane, hipMemcpyDeviceToDevice, c->stream));
    HS_HIP(c, hipStreamSynchronize(c->stream));
    if ((st = copy_frame_in(c, c->dB + pair * c->plane, next, ns, hipMemcpyHostToDevice, true))) return st;
    c->coef_valid = false;
    return HSFLOW_OK;
}

// Upload one host frame (colour or gray) into scratch, convert / blur on the device into dst.
static int preprocess_frame(hsflow_ctx *c, uint8_t *dst, const uint8_t *host, size_t stride, bool colour, bool blur)
{
    const size_t bgr_bytes = (size_t)c->W * 3 * c->H, gray_bytes = (size_t)c->plane;
    const size_t need = bgr_bytes + 2 * gray_bytes;
    if (c->scratch_bytes < need) {
        hipFree(c->dScratch);
        c->dScratch = nullptr; c->scratch_bytes = 0;
        HS_HIP(c, hipMalloc(&c->dScratch, need));
        c->scratch_bytes = need;
    }
    uint8_t *dBgr = (uint8_t *)c->dScratch, *dGray = dBgr + bgr_bytes;
    const dim3 grid((c->W + 255) / 256, (c->H + 3) / 4), block(64, 4);
    HS_HIP(c, hipStreamSynchronize(c->stream));
    if (colour) {
        HS_HIP(c, hipMemcpy2D(dBgr, (size_t)c->W * 3, host, stride, (size_t)c->W * 3, c->H, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(hsk::k_bgr2gray, grid, block, 0, c->stream, dBgr, (long long)c->W * 3, blur ? dGray : dst, c->W, c->H, c->P);
    } else {
        HS_HIP(c, hipMemcpy2D(blur ? dGray : dst, c->P, host, stride, c->W, c->H, hipMemcpyHostToDevice));
    }
    if (blur) hipLaunchKernelGGL(hsk::k_box_blur3, grid, block, 0, c->stream, dGray, dst, c->W, c->H, c->P);
    HS_HIP(c, hipGetLastError());
    HS_HIP(c, hipStreamSynchronize(c->stream));
    return HSFLOW_OK;
}

int hsflow_set_frames_bgr8(hsflow_ctx *c, int pair, const uint8_t *prev, size_t ps, const uint8_t *curr, size_t cs, int blur3x3)
{
    int st = check_ctx(c, pair);
    if (st) return st;
    if ((st = settle_pending(c))) return st; // an unverified asynchronous solve still needs the old inputs
    if (!prev || !curr) return fail(c, HSFLOW_E_ARG, "null frame pointer");
    if (ps < (size_t)c->W * 3 || cs < (size_t)c->W * 3) return fail(c, HSFLOW_E_SIZE, "colour frame stride smaller than 3*width");
    if ((st = preprocess_frame(c, c->dA + pair * c->plane, prev, ps, true, blur3x3 != 0))) return st;
    if ((st = preprocess_frame(c, c->dB + pair * c->plane, curr, cs, true, blur3x3 != 0))) return st;
    c->frames_set = true;
    c->coef_valid = false;
    return HSFLOW_OK;
}

int hsflow_set_frames_gray8_blur(hsflow_ctx *c, int pair, const uint8_t *prev, size_t ps, const uint8_t *curr, size_t cs)
{
    int st = check_ctx(c, pair);
    if (st) return st;
    if ((st = settle_pending(c))) return st; // an unverified asynchronous solve still needs the old inputs
    if (!prev || !curr) return fail(c, HSFLOW_E_ARG, "null frame pointer");
    if (ps < (size_t)c->W || cs < (size_t)c->W) return fail(c, HSFLOW_E_SIZE, "frame stride smaller than width");
    if ((st = preprocess_frame(c, c->dA + pair * c->plane, prev, ps, false, true))) return st;
    if ((st = preprocess_frame(c, c->dB + pair * c->plane, curr, cs, false, true))) return st;
    c->frames_set = true;
    c->coef_valid = false;
    return HSFLOW_OK;
}

int hsflow_solve(hsflow_ctx *c, const hsflow_params *p) { return solve_impl(c, p, false); }
int hsflow_solve_async(hsflow_ctx *c, const hsflow_params *p) { return solve_impl(c, p, true); }

int hsflow_synchronize(hsflow_ctx *c)
{
    int st = check_ctx(c, 0);
    if (st) return st;
    if ((st = settle_pending(c))) return st;
    HS_HIP(c, hipStreamSynchronize(c->stream));
    return HSFLOW_OK;
}

int hsflow_get_flow(hsflow_ctx *c, int pair, float *u, size_t us, float *v, size_t vs)
{
    int st = check_ctx(c, pair);
    if (st) return st;
    if (!u || !v) return fail(c, HSFLOW_E_ARG, "null flow pointer");
    const size_t rowb = (size_t)c->W * 4;
    if ((us & 3) || (vs & 3) || us < rowb || vs < rowb) return fail(c, HSFLOW_E_SIZE, "flow stride must be a multiple of 4 and >= 4*width");
    if ((st = settle_pending(c))) return st;
    HS_HIP(c, hipStreamSynchronize(c->stream));
    HS_HIP(c, hipMemcpy2D(u, us, c->dU[c->cur] + pair * c->plane, (size_t)c->P * 4, rowb, c->H, hipMemcpyDeviceToHost));
    HS_HIP(c, hipMemcpy2D(v, vs, c->dV[c->cur] + pair * c->plane, (size_t)c->P * 4, rowb, c->H, hipMemcpyDeviceToHost));
    return HSFLOW_OK;
}

int hsflow_get_flow_async(hsflow_ctx *c, int pair, float *u, size_t us, float *v, size_t vs)
{
    int st = check_ctx(c, pair);
    if (st) return st;
    if (!u || !v) return fail(c, HSFLOW_E_ARG, "null flow pointer");
    const size_t rowb = (size_t)c->W * 4;
    if ((us & 3) || (vs & 3) || us < rowb || vs < rowb) return fail(c, HSFLOW_E_SIZE, "flow stride must be a multiple of 4 and >= 4*width");
    HS_HIP(c, copy_rows_async(c, u, us, c->dU[c->cur] + pair * c->plane, (size_t)c->P * 4, rowb, c->H, hipMemcpyDeviceToHost));
    HS_HIP(c, copy_rows_async(c, v, vs, c->dV[c->cur] + pair * c->plane, (size_t)c->P * 4, rowb, c->H, hipMemcpyDeviceToHost));
    return HSFLOW_OK;
}

static int flow_rows_args(hsflow_ctx *c, int pair, int row0, int nrows, const void *du, size_t us, const void *dv, size_t vs)
{
    int st = check_ctx(c, pair);
    if (st) return st;
    if (!du || !dv) return fail(c, HSFLOW_E_ARG, "null flow pointer");
    if (row0 < 0 || nrows <= 0 || row0 + nrows > c->H) return fail(c, HSFLOW_E_SIZE, "row range outside the frame");
    const size_t rowb = (size_t)c->W * 4;
    if ((us & 3) || (vs & 3) || us < rowb || vs < rowb) return fail(c, HSFLOW_E_SIZE, "flow stride must be a multiple of 4 and >= 4*width");
    return HSFLOW_OK;
}

int hsflow_get_flow_device(hsflow_ctx *c, int pair, int row0, int nrows, void *du, size_t us, void *dv, size_t vs)
{
    int st = flow_rows_args(c, pair, row0, nrows, du, us, dv, vs);
    if (st) return st;
    const size_t rowb = (size_t)c->W * 4;
    const long long off = pair * c->plane + (long long)row0 * c->P;
    HS_HIP(c, hipMemcpy2DAsync(du, us, c->dU[c->cur] + off, (size_t)c->P * 4, rowb, nrows, hipMemcpyDeviceToDevice, c->stream));
    HS_HIP(c, hipMemcpy2DAsync(dv, vs, c->dV[c->cur] + off, (size_t)c->P * 4, rowb, nrows, hipMemcpyDeviceToDevice, c->stream));
    return HSFLOW_OK;
}

int hsflow_set_flow_device(hsflow_ctx *c, int pair, int row0, int nrows, const void *du, size_t us, const void *dv, size_t vs)
{
    int st = flow_rows_args(c, pair, row0, nrows, du, us, dv, vs);
    if (st) return st;
    if ((st = settle_pending(c))) return st; // an unverified asynchronous solve still needs the old inputs
    const size_t rowb = (size_t)c->W * 4;
    const long long off = pair * c->plane + (long long)row0 * c->P;
    HS_HIP(c, hipMemcpy2DAsync(c->dU[c->cur] + off, (size_t)c->P * 4, du, us, rowb, nrows, hipMemcpyDeviceToDevice, c->stream));
    HS_HIP(c, hipMemcpy2DAsync(c->dV[c->cur] + off, (size_t)c->P * 4, dv, vs, rowb, nrows, hipMemcpyDeviceToDevice, c->stream));
    return HSFLOW_OK;
}

int hsflow_get_derivatives(hsflow_ctx *c, int pair, float *dx, float *dy, float *dt, size_t stride)
{
    int st = check_ctx(c, pair);
    if (st) return st;
    if (!dx || !dy || !dt) return fail(c, HSFLOW_E_ARG, "null derivative pointer");
    const size_t rowb = (size_t)c->W * 4;
    if ((stride & 3) || stride < rowb) return fail(c, HSFLOW_E_SIZE, "stride must be a multiple of 4 and >= 4*width");
    if (!c->coef_valid) return fail(c, HSFLOW_E_STATE, "no derivatives yet: call hsflow_solve first");
    const size_t need = (size_t)c->W * c->H * 3 * sizeof(float);
    if (c->scratch_bytes < need) {
        hipFree(c->dScratch);
        c->dScratch = nullptr; c->scratch_bytes = 0;
        HS_HIP(c, hipMalloc(&c->dScratch, need));
        c->scratch_bytes = need;
    }
    if (c->coef_mode == HSFLOW_MODE_CLASSIC) { // planar fp32 already: strided copies
        HS_HIP(c, hipStreamSynchronize(c->stream));
        float *dst[3] = {dx, dy, dt};
        for (int i = 0; i < 3; i++)
            HS_HIP(c, hipMemcpy2D(dst[i], stride, c->dE[i] + pair * c->plane, (size_t)c->P * 4, rowb, c->H, hipMemcpyDeviceToHost));
        return HSFLOW_OK;
    }
    float *sx = (float *)c->dScratch, *sy = sx + (size_t)c->W * c->H, *stt = sy + (size_t)c->W * c->H;
    hipLaunchKernelGGL(hsk::k_unpack_deriv, dim3((c->W + 255) / 256, c->H), dim3(256), 0, c->stream,
                       c->dCoef + pair * c->plane, sx, sy, stt, c->W, c->H, c->P);
    HS_HIP(c, hipGetLastError());
    HS_HIP(c, hipStreamSynchronize(c->stream));
    HS_HIP(c, hipMemcpy2D(dx, stride, sx, rowb, rowb, c->H, hipMemcpyDeviceToHost));
    HS_HIP(c, hipMemcpy2D(dy, stride, sy, rowb, rowb, c->H, hipMemcpyDeviceToHost));
    HS_HIP(c, hipMemcpy2D(dt, stride, stt, rowb, rowb, c->H, hipMemcpyDeviceToHost));
    return HSFLOW_OK;
}

int hsflow_get_frames_u8(hsflow_ctx *c, int pair, uint8_t *prev, size_t ps, uint8_t *curr, size_t cs)
{
    int st = check_ctx(c, pair);
    if (st) return st;
    if (!prev || !curr) return fail(c, HSFLOW_E_ARG, "null frame pointer");
    if (ps < (size_t)c->W || cs < (size_t)c->W) return fail(c, HSFLOW_E_SIZE, "frame stride smaller than width");
    HS_HIP(c, hipStreamSynchronize(c->stream));
    HS_HIP(c, hipMemcpy2D(prev, ps, c->dA + pair * c->plane, c->P, c->W, c->H, hipMemcpyDeviceToHost));
    HS_HIP(c, hipMemcpy2D(curr, cs, c->dB + pair * c->plane, c->P, c->W, c->H, hipMemcpyDeviceToHost));
    return HSFLOW_OK;
}

int hsflow_get_info(hsflow_ctx *c, hsflow_info *info)
{
    if (!c) return fail(nullptr, HSFLOW_E_ARG, "null context");
    if (!info || info->struct_size != sizeof(hsflow_info)) return fail(c, HSFLOW_E_ARG, "info null or struct_size mismatch");
    const int st = settle_pending(c); // iterations_done / last_eps of an asynchronous ITER|EPS solve
    if (st) return st;
    *info = c->info;
    return HSFLOW_OK;
}

int hsflow_calc_optical_flow_hs_8u32f(const uint8_t *prev, const uint8_t *curr, int img_step, int width, int height,
                                      int use_previous, float *velx, float *vely, int vel_step, float lambda,
                                      int term_type, int max_iter, double epsilon)
{
    // argument checks of the original, cv210.dll VA 0x1012e089-0x1012e0c7
    if (!prev || !curr || !velx || !vely) return fail(nullptr, HSFLOW_E_ARG, "null pointer");
    if (width <= 0 || height <= 0 || width > img_step || (vel_step & 3) || width * 4 > vel_step)
        return fail(nullptr, HSFLOW_E_SIZE, "bad size or step");
    // The reference calls cvCalcOpticalFlowHS once per frame of a stream (OpticalFlowOpenCV.cpp:94):
    // building a context per call (7 device allocations, a stream) would cost several solves, so the
    // one-shot form keeps the last context and reuses it while the frame size stays the same.
    std::lock_guard<std::mutex> lock(g_oneshot_mutex);
    hsflow_ctx *c = g_oneshot;
    int st = HSFLOW_OK;
    if (!c || c->W != width || c->H != height) {
        if (c) hsflow_destroy(c);
        g_oneshot = c = nullptr;
        if ((st = hsflow_create(&c, 0, width, height, 1, nullptr, 1))) return st;
        g_oneshot = c;
    }
    hsflow_params p;
    hsflow_default_params(&p);
    p.lambda = lambda; p.term_type = term_type; p.max_iter = max_iter; p.epsilon = epsilon;
    p.use_previous = use_previous ? 1 : 0;
    st = hsflow_set_frames_u8(c, 0, prev, (size_t)img_step, curr, (size_t)img_step);
    if (!st && use_previous) { // the caller's velx/vely are the starting flow
        const size_t rowb = (size_t)width * 4;
        hipError_t e = hipMemcpy2D(c->dU[c->cur], (size_t)c->P * 4, velx, (size_t)vel_step, rowb, height, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy2D(c->dV[c->cur], (size_t)c->P * 4, vely, (size_t)vel_step, rowb, height, hipMemcpyHostToDevice);
        if (e != hipSuccess) st = HSFLOW_E_DEVICE;
    }
    if (!st) st = hsflow_solve(c, &p);
    if (!st) st = hsflow_get_flow(c, 0, velx, (size_t)vel_step, vely, (size_t)vel_step);
    if (st) { // do not keep a context in an unknown state
        g_create_error = c->err;
        hsflow_destroy(c);
        g_oneshot = nullptr;
    }
    return st;
}

void hsflow_release_cached(void)
{
    std::lock_guard<std::mutex> lock(g_oneshot_mutex);
    if (g_oneshot) hsflow_destroy(g_oneshot);
    g_oneshot = nullptr;
}

} // extern "C"
