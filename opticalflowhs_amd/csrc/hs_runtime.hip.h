// hs_runtime.hip.h -- part of libhsflow.so: graph cache, HIP-event profiler, Eps buffers and read-back,
// diagnostic stamps.
#pragma once

namespace {
// The graph cache is keyed by everything a captured launch sequence depends on (sizes, kernel shape,
// lambda, epsilon ...); a caller that varies those from call to call must not grow it without bound.
constexpr size_t kMaxGraphs = 32;
void drop_graphs(hsflow_ctx *c)
{
    if (c->graphs.empty()) return;
    hipStreamSynchronize(c->stream); // no replay of an old graph may still be running
    for (auto &kv : c->graphs) {
        if (kv.second.exec) hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) hipGraphDestroy(kv.second.graph);
    }
    c->graphs.clear();
}

void trim_graph_cache(hsflow_ctx *c)
{
    if (c->graphs.size() >= kMaxGraphs) drop_graphs(c);
}

int check_ctx(hsflow_ctx *c, int pair)
{
    if (!c) return fail(nullptr, HSFLOW_E_ARG, "null context");
    if (pair < 0 || pair >= c->N) return fail(c, HSFLOW_E_ARG, "pair index out of range");
    if (hipSetDevice(c->device) != hipSuccess) return fail(c, HSFLOW_E_DEVICE, "hipSetDevice failed");
    // The launch wrappers report hipGetLastError() after a launch; that is the LAST error of this host thread, so a HIP call
    // that failed earlier -- anywhere in the process, already reported to its own caller -- must not surface here.
    (void)hipGetLastError();
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

// --- the persistent launch (HSFLOW_KERNEL_PERSIST) -------------------------------------------------------------
// Buffers it needs (allocation only: called outside any capture).
int persist_reserve(hsflow_ctx *c)
{
    const size_t px = (size_t)c->plane * c->N;
    if (!c->dUp) HS_HIP(c, hipMalloc((void **)&c->dUp, px * sizeof(float)));
    if (!c->dVp) HS_HIP(c, hipMalloc((void **)&c->dVp, px * sizeof(float)));
    if (!c->dFlags) {
        HS_HIP(c, hipMalloc((void **)&c->dFlags, (size_t)(kMaxPersistTiles + 1) * sizeof(unsigned)));
        c->persist_tiles = 0;
    }
    if (!c->hErr) {
        HS_HIP(c, hipHostMalloc((void **)&c->hErr, 64, hipHostMallocMapped | hipHostMallocCoherent));
        HS_HIP(c, hipHostGetDevicePointer((void **)&c->hErrDev, c->hErr, 0));
        *c->hErr = 0u;
    }
    return HSFLOW_OK;
}

// The tiles' phase counters count up across solves and agree at every launch boundary as long as the grid stays the
// same; a different grid (or an aborted launch) starts from cleared counters.  Enqueued ahead of the launch, outside
// any capture.
int persist_prepare_flags(hsflow_ctx *c, int tiles)
{
    if (c->persist_tiles == tiles) return HSFLOW_OK;
    HS_HIP(c, hipMemsetAsync(c->dFlags, 0, (size_t)(kMaxPersistTiles + 1) * sizeof(unsigned), c->stream));
    c->persist_tiles = tiles;
    return HSFLOW_OK;
}

// A persistent launch gave up waiting (another grid held part of the CUs, a neighbour never came): the flow it left is
// invalid.  The context goes back to a launch per fuse_steps iterations for good.
void persist_failed(hsflow_ctx *c)
{
    if (c->hErr) *c->hErr = 0u;
    c->persist_tiles = 0;
    c->persist_off = true;
    c->persist_unchecked = false;
    c->plan_cache.clear();
}

// Did the last persistent launch(es) of this context give up?  Only meaningful once the stream has drained.
bool persist_error(const hsflow_ctx *c) { return c->hErr && *(volatile unsigned *)c->hErr != 0u; }

// One persistent launch = `iters` sweeps in phases of sp.g.T: input dU[cur] (or zero), output dU[cur ^ 1]; the phases
// alternate between that buffer and the third one so that the last phase lands in it.
int enqueue_persist(hsflow_ctx *c, const StripPlan &sp0, int iters, int eps, bool deriv, int zero_in, float coeff)
{
    StripPlan sp = sp0;
    sp.g.zero_in = zero_in;
    hsk::PersistArgs pa;
    const int T = sp.g.T;
    pa.n_phase = (iters + T - 1) / T;
    pa.T_last = iters - (pa.n_phase - 1) * T;
    pa.flags = c->dFlags;
    pa.err = c->hErrDev;
    // (HSFLOW_PERSIST_WAIT_TICKS: test hook -- with 0 every wait that is not already satisfied gives up, which drives the
    // abort and fall-back path)
    static const char *ticks_env = getenv("HSFLOW_PERSIST_WAIT_TICKS");
    pa.wait_ticks = ticks_env ? (unsigned)strtoul(ticks_env, nullptr, 10) : kPersistWaitTicks;
    const int a = c->cur, b = a ^ 1;
    const int lastb = (pa.n_phase - 1) & 1;
    pa.ub[lastb] = c->dU[b]; pa.vb[lastb] = c->dV[b];
    pa.ub[lastb ^ 1] = c->dUp; pa.vb[lastb ^ 1] = c->dVp;
    HS_HIP(c, launch_persist(c, sp, pa, eps, deriv, c->dU[a], c->dV[a], coeff));
    c->cur = b;
    return HSFLOW_OK;
}

// Enqueue derivative pass + `iters` Jacobi sweeps (no host synchronisation inside).
int enqueue_fixed(hsflow_ctx *c, const hsflow_params &p, float coeff, int iters, int kernel, int T,
                  const JPlan *plan, const JPlan *tail_plan, Profiler &prof, bool do_deriv, bool zero_flow, bool persist = false)
{
    // u = v = 0 at the start (reference behaviour, use_previous = 0): instead of clearing two
    // planes and reading them back, the first launch is told that its input is zero.
    int zero_in = zero_flow ? 1 : 0;
    if (zero_flow) c->cur = 0;
    // the derivative pass rides in the first Jacobi launch where the kernel can do it (not when profiling:
    // deriv_ms / jacobi_ms then keep their meaning)
    bool fuse = do_deriv && !p.profile && kernel != HSFLOW_KERNEL_SIMPLE &&
                strip_deriv_fusable(c, iters >= T ? *plan : *tail_plan);
    if (do_deriv && !fuse) {
        prof.begin(0);
        HS_HIP(c, launch_deriv(c));
        prof.end();
    }
    int left = iters, launches = 0;
    if (persist) { // the whole budget as one launch
        prof.begin(1);
        const int st = enqueue_persist(c, plan->s, iters, 0, fuse, zero_in, coeff);
        prof.end();
        if (st) return st;
        left = 0;
        launches = 1;
    }
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
            HS_HIP(c, launch_j(c, *pl, false, c->dU[a], c->dV[a], c->dU[b], c->dV[b], coeff, false, zero_in, fuse));
            prof.end();
            fuse = false;
            left -= pl->T;
        }
        c->cur = b;
        zero_in = 0;
        launches++;
    }
    c->info.jacobi_launches = launches;
    return HSFLOW_OK;
}

// Eps bookkeeping of an EPS-terminated solve: `sweeps` rows of `stride` words, cleared, plus the
// reduction of the rows into hEps[0..sweeps).
// Buffers for `sweeps` Eps words of `stride` workgroups each (device) and their host copy; allocation
// only, so that what follows can be captured in a graph.
int eps_reserve(hsflow_ctx *c, int sweeps, int stride)
{
    const size_t need = (size_t)sweeps * stride;
    if (c->epsTilesCap < need) {
        drop_graphs(c); // captured launches hold the old address (this also waits for the stream)
        hipFree(c->dEpsTiles);
        c->dEpsTiles = nullptr; c->epsTilesCap = 0;
        HS_HIP(c, hipMalloc((void **)&c->dEpsTiles, need * sizeof(unsigned)));
        c->epsTilesCap = need;
    }
    if (c->hEpsCap < (size_t)sweeps) {
        drop_graphs(c);
        HS_HIP(c, hipStreamSynchronize(c->stream)); // nothing in flight may still write the old buffer
        if (c->hEps) hipHostFree(c->hEps);
        c->hEps = c->hEpsDev = nullptr; c->hEpsCap = 0;
        const size_t cap = std::max<size_t>(256, (size_t)sweeps * 2);
        HS_HIP(c, hipHostMalloc((void **)&c->hEps, cap * sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent));
        HS_HIP(c, hipHostGetDevicePointer((void **)&c->hEpsDev, c->hEps, 0));
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

// Reduces the rows of per-workgroup words to one word per row, straight into the host's buffer (no copy
// node).  Rows [0, n_first) hold cnt_first valid words, the others cnt_last (defaults: every row is
// epsStride words, which then must have been cleared where a launch had fewer workgroups).
// mark: the kernel's last workgroup also writes the context's marker (hsflow_set_async_reduce; the caller counts it).
int eps_collect_enqueue(hsflow_ctx *c, int sweeps, int n_first = 0, int cnt_first = 0, int cnt_last = -1, bool mark = false)
{
    hipLaunchKernelGGL(hsk::k_eps_reduce, dim3(sweeps), dim3(256), 0, c->stream, c->dEpsTiles, c->epsStride, c->hEpsDev,
                       n_first, cnt_first, cnt_last < 0 ? c->epsStride : cnt_last, mark ? c->dSeq : nullptr, mark ? c->hMarkDev : nullptr);
    HS_HIP(c, hipGetLastError());
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
// last_is_exact: the last slot is the measured Eps of the final sweep (a stop there IS the budget, so it proves
// nothing and fails nothing); otherwise every slot is a witness word and all of them must clear epsilon.
bool witness_proven(const unsigned *w, int slots, double epsilon, float *last, bool last_is_exact)
{
    float e = 0.f;
    for (int i = 0; i < slots; i++) {
        std::memcpy(&e, &w[i], sizeof(float));
        if (!((double)e >= epsilon) && !(last_is_exact && i == slots - 1)) return false;
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
    // the persistent launch: cycles spent in sweeps / publish (stores drained, barrier) / wait (counters) / halo reload,
    // summed over the phases, per workgroup
    if (c->info.persistent && tiles <= 8192) {
        std::vector<unsigned long long> pq((size_t)tiles * 32);
        if (hipMemcpy(pq.data(), c->dStamps + (size_t)tiles * 8, pq.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
            for (int i = 0; i < tiles; i++)
                fprintf(f, "P %d %llu %llu %llu %llu %d %llu %llu\n", i, pq[(size_t)i * 32], pq[(size_t)i * 32 + 1], pq[(size_t)i * 32 + 2],
                        pq[(size_t)i * 32 + 3], c->info.persistent, pq[(size_t)i * 32 + 4], pq[(size_t)i * 32 + 5]);
    }
    // per-sweep end stamps of the strip kernel (cycles since the end of the load phase), HSFLOW_DEBUG_STAMPS_SWEEPS=1
    if (getenv("HSFLOW_DEBUG_STAMPS_SWEEPS") && tiles <= 8192 && c->info.kernel == HSFLOW_KERNEL_STRIP) {
        const int T = std::min(c->info.fuse_steps, 32);
        std::vector<unsigned long long> sw((size_t)tiles * 32);
        if (hipMemcpy(sw.data(), c->dStamps + (size_t)tiles * 8, sw.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            for (int i = 0; i < tiles; i++) {
                fprintf(f, "S %d", i);
                for (int k = 0; k < T; k++) fprintf(f, " %llu", sw[(size_t)i * 32 + k] - h[(size_t)i * 8 + 1]);
                fprintf(f, "\n");
            }
        }
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

} // namespace
