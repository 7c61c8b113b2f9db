// hs_context.hip.h -- part of libhsflow.so (one translation unit, see hsflow.hip): plan / graph types, the
// context object behind hsflow_ctx, error plumbing.
#pragma once

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

// What solve_impl works out once and the three termination paths share.
struct SolveSetup {
    float coeff;        // Ilambda = fl32(1 / fl32(lambda))
    int kernel;         // kernel actually used (AUTO resolved)
    bool multi;         // several sweeps per launch (every kernel but the simple one)
    bool use_iter, use_eps;
    long long budget;   // sweep budget (huge when ITER does not apply)
    int T;              // sweeps per full launch
    JPlan plan;         // launch plan for T sweeps
    bool persist;       // the whole budget as ONE persistent launch in phases of T sweeps (HSFLOW_KERNEL_PERSIST)
    hsflow_params eff;  // the caller's parameters as the solve paths use them (strip_rows may have been fixed, see prepare_solve)
};

// prepare_solve's result for one set of parameters (hsflow_ctx::plan_cache)
struct PlanKey {
    hsflow_params p;
    int async, exact;
};
struct PlanEntry {
    PlanKey key;
    SolveSetup S;
    hsflow_info info;
};

} // namespace

struct hsflow_ctx {
    int device = 0;
    int W = 0, H = 0, N = 0, P = 0;
    int org = 0;         // frame row of this context's row 0, modulo 2 (hsflow_set_row_origin)
    long long plane = 0; // elements per pair plane
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint8_t *dA = nullptr, *dB = nullptr;
    uint32_t *dCoef = nullptr;
    float *dE[3] = {nullptr, nullptr, nullptr}; // CLASSIC mode: Ex, Ey, Et planes for the kernels that read planes (allocated on first use)
    bool dE_valid = false;       // ... and whether they hold the current derivatives (dCoef always does, packed)
    int coef_mode = -1;          // discretisation the current derivatives belong to
    float *dU[2] = {nullptr, nullptr}, *dV[2] = {nullptr, nullptr};
    unsigned long long *dStamps = nullptr; // diagnostic phase stamps (HSFLOW_DEBUG_STAMPS), else NULL
    unsigned *dEps = nullptr;   // kMaxFuse words: Eps sink of launches that do not collect it
    unsigned *epsPtr = nullptr; // where the running launch records Eps: [sweep][epsStride] words
    float epsThr = 0.f;         // witness launches: smallest float >= epsilon
    unsigned *hEps = nullptr;   // page-locked, device-visible: k_eps_reduce writes the per-sweep Eps words here
    unsigned *hEpsDev = nullptr; // the device's address of hEps
    size_t hEpsCap = 0;
    int epsStride = 1;          // words per sweep: one per workgroup (strip / fold), else 1
    unsigned *dEpsTiles = nullptr; // per-sweep, per-workgroup Eps of the launches of one solve
    size_t epsTilesCap = 0;
    float *dUb = nullptr, *dVb = nullptr; // backup of the starting flow (ITER|EPS with use_previous)
    // the persistent launch (HSFLOW_KERNEL_PERSIST, hs_kernels_strip.hip.h): a third flow buffer (its phases alternate
    // between this one and the ping-pong buffer the result lands in; the starting flow stays intact), the tiles' phase
    // counters, the error word (page-locked, device-visible)
    float *dUp = nullptr, *dVp = nullptr;
    unsigned *dFlags = nullptr;
    unsigned *hErr = nullptr, *hErrDev = nullptr;
    int persist_tiles = 0;       // grid the phase counters are consistent for (0: to be cleared before the next launch)
    bool persist_off = false;    // a persistent launch timed out on this context: not used again
    bool persist_unchecked = false; // an asynchronous persistent solve whose error word has not been looked at yet
    bool counted = false;        // this context is in g_live_ctx
    int eps_row0 = 0, eps_rows = 0; // hsflow_set_eps_rows: rows whose changes count for Eps (0 rows: the whole frame)
    std::vector<float> sweep_eps;   // Eps of every sweep of the last exact (per-sweep) pass: hsflow_solve_probe hands it out
    // hsflow_set_async_reduce: every asynchronous solve is followed by k_mark_done; the host waits for a solve by polling the
    // page-locked word (hsflow_wait_solve) instead of waiting for the stream
    unsigned *dSeq = nullptr, *hMark = nullptr, *hMarkDev = nullptr;
    float *dZero = nullptr; // one row of zeros (P floats): stands in for u and v in strip / fold launches that start from zero flow
    unsigned mark_issued = 0;       // markers enqueued so far = the value the last one will write
    bool last_marked = false;       // the last solve was followed by a marker
    bool async_reduce = false;      // hsflow_set_async_reduce
    int cu_share = 0;            // > 0: the planners count on this many CUs only (hsflow_set_cu_share); 0: the whole chip
    int num_cu = 0;              // compute units of the device (one workgroup of the persistent launch per CU)
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
        int stride = 1, n_first = 0, cnt_first = 0, cnt_last = 0; // layout of its witness words (k_eps_reduce's arguments)
        bool reduced = false;   // the reduction into hEps was enqueued with the solve (async_reduce)
        unsigned mark = 0;      // ... and the marker behind it
        bool marked_by_reduce = false; // ... written by the reduction kernel's last workgroup
    } pend;
    // what it takes to measure last_eps of an asynchronous ITER|EPS solve on demand (hsflow_get_info): its last
    // launch again, from the input buffer that launch left intact, with the final sweep's Eps measured
    struct LastLaunch {
        bool valid = false;    // last_eps not measured yet
        JPlan plan;
        int in = 0, zero_in = 0;
        float coeff = 0.f, eps_thr = 0.f;
        bool from_third = false; // its input lies in dUp / dVp (last phase of a persistent launch), not the other ping-pong buffer
    } lastl;
    bool force_exact = false; // the exact per-sweep pass is wanted (set while a pending solve is settled)
    std::map<GraphKey, GraphEntry> graphs;
    std::vector<PlanEntry> plan_cache; // prepare_solve's results by parameters; cleared when what they rest on changes
    std::vector<hipEvent_t> events;
};

namespace {

// live contexts per device (this process): the persistent launch wants the device to itself -- two persistent grids
// from two contexts could each hold part of the CUs and wait for workgroups that cannot start
std::atomic<int> g_live_ctx[64];
constexpr int kMaxPersistTiles = 4096;

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

} // namespace
