// hs_solve.hip.h -- part of libhsflow.so: the solve paths (classic mode; CV mode with a fixed sweep count,
// speculative ITER|EPS with witness launches, EPS-only chunks) and the deferred check of asynchronous solves.
#pragma once

namespace {
int solve_impl(hsflow_ctx *c, const hsflow_params *pp, bool async);

// Settles an ITER|EPS solve that hsflow_solve_async left unverified: waits for the stream, looks at the
// witness words and, if they do not prove "no early stop", runs the exact pass from the saved start.
// After the stream has drained: did an asynchronous persistent launch give up?  Its flow is invalid then; the context
// goes back to a launch per fuse_steps iterations and the caller is told.
int check_persist(hsflow_ctx *c)
{
    if (!c->persist_unchecked) return HSFLOW_OK;
    c->persist_unchecked = false;
    if (!persist_error(c)) return HSFLOW_OK;
    persist_failed(c);
    c->coef_valid = false;
    return fail(c, HSFLOW_E_DEVICE, "a persistent launch of an asynchronous solve timed out (another grid on the device?): its flow is invalid; "
                                    "this context now launches per fuse_steps iterations, solve again");
}

// verdict_only: report whether the witness words prove "no early stop" and leave it at that (no exact pass; the flow of
// the whole budget stands) -- for a driver that decides over several contexts (row slabs: hsflow_take_verdict).
// Waits until the marker kernel that wrote `target` has run (k_mark_done: everything enqueued before it is done, and what the
// reduction kernel wrote to host memory is visible).  Polls page-locked memory; gives the stream a proper wait after 2 s.
int wait_marker(hsflow_ctx *c, unsigned target)
{
    if (!c->hMark) { HS_HIP(c, hipStreamSynchronize(c->stream)); return HSFLOW_OK; }
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while ((int)(__atomic_load_n(c->hMark, __ATOMIC_ACQUIRE) - target) < 0) {
        if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            HS_HIP(c, hipStreamSynchronize(c->stream));
            break;
        }
    }
    return HSFLOW_OK;
}

int settle_pending(hsflow_ctx *c, int *verdict_only = nullptr)
{
    if (!c->pend.active) return HSFLOW_OK;
    c->pend.active = false;
    HS_HIP(c, hipSetDevice(c->device));
    // the witness words of that solve's launches are still per workgroup: an asynchronous solve enqueues no
    // reduction (a stream of solves would pay a kernel and a boundary per solve for words nobody reads); now
    // that somebody wants the verdict, reduce them into the host's buffer and wait
    if (!c->pend.reduced) { // (hsflow_set_async_reduce: the solve enqueued the reduction itself, right behind its last launch)
        c->epsStride = c->pend.stride;
        int st0 = eps_collect_enqueue(c, c->pend.slots, c->pend.n_first, c->pend.cnt_first, c->pend.cnt_last);
        if (st0) return st0;
        HS_HIP(c, hipStreamSynchronize(c->stream));
    } else {
        // ... and a marker behind it: only THAT is waited for, not what other contexts may have enqueued on the same stream
        // since (the slots of a pair pipeline share streams: pair_pipeline.cpp)
        int stw = wait_marker(c, c->pend.mark);
        if (stw) return stw;
    }
    float last = 0.f;
    const bool gave_up = c->persist_unchecked && persist_error(c); // a persistent launch that timed out proves nothing
    c->persist_unchecked = false;
    if (gave_up) {
        persist_failed(c);
        c->coef_valid = false;
    }
    if (!gave_up && witness_proven(c->hEps, c->pend.slots, c->pend.params.epsilon, &last, false)) {
        c->info.iterations_done = c->pend.iters;
        c->info.last_eps = NAN; // not measured by an asynchronous solve; hsflow_get_info measures it on demand (c->lastl)
        if (verdict_only) *verdict_only = 1;
        return HSFLOW_OK;
    }
    if (verdict_only && !gave_up) { // not proven, and the caller decides what follows
        *verdict_only = 0;
        c->info.iterations_done = c->pend.iters;
        c->info.last_eps = NAN;
        return HSFLOW_OK;
    }
    c->lastl.valid = false;
    hsflow_params q = c->pend.params;
    q.reuse_derivatives = gave_up ? 0 : 1; // the coefficient plane of that solve is still in place
    if (q.kernel == HSFLOW_KERNEL_PERSIST) q.kernel = HSFLOW_KERNEL_STRIP;
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

// last_eps of an asynchronous ITER|EPS solve, on demand: the solve ran witness launches only (they prove "no early
// stop" but measure nothing).  Its last launch left its input buffer intact, so that launch is simply run again with
// the final sweep's Eps measured (mode 3); it rewrites the flow with the same values.
int measure_last_eps(hsflow_ctx *c)
{
    if (!c->lastl.valid) return HSFLOW_OK;
    c->lastl.valid = false;
    const hsflow_ctx::LastLaunch &L = c->lastl;
    const int stride = L.plan.s.tiles;
    int st = eps_reserve(c, 2, stride);
    if (st) return st;
    const int b = c->cur, a = b ^ 1;
    c->epsPtr = c->dEpsTiles;
    c->epsStride = stride;
    c->epsThr = L.eps_thr;
    const float *ui = L.from_third ? c->dUp : c->dU[a], *vi = L.from_third ? c->dVp : c->dV[a];
    const hipError_t e = launch_j(c, L.plan, 3, ui, vi, c->dU[b], c->dV[b], L.coeff, false, L.zero_in, false);
    HS_HIP(c, e);
    if ((st = eps_collect_enqueue(c, 2, 0, 0, stride))) return st; // (resets epsPtr / epsStride)
    HS_HIP(c, hipStreamSynchronize(c->stream));
    float last = 0.f;
    std::memcpy(&last, &c->hEps[1], sizeof(float));
    c->info.last_eps = last;
    return HSFLOW_OK;
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

// CLASSIC mode (Kernels.cl semantics, v restored): derivatives once, then max_iter fused average+update sweeps --
// several per launch on a register strip (k_classic_strip) or an LDS tile (k_jacobi_classic_fused), or one per launch
// (k_jacobi_classic).  The reference loop has no other stop rule (HSOpticalFlowOpenCL.cpp:750-751), so only ITER
// termination is accepted.  The launch sequence can be one hipGraph (use_graph), like the CV-mode solve.
struct ClassicSetup {
    int kernel, T; // kernel actually used (AUTO resolved), sweeps per full launch
    ClassicStripPlan splan, stail;
    FusedPlan fplan, ftail;
};

// Checks the parameters, picks the kernel and the launch plans and writes the plan into c->info (hsflow_plan_query
// stops here; no device is touched).
int prepare_classic(hsflow_ctx *c, const hsflow_params &p, ClassicSetup &S)
{
    if (p.term_type != HSFLOW_TERM_ITER) return fail(c, HSFLOW_E_ARG, "CLASSIC mode supports ITER termination only");
    if (p.max_iter <= 0) return fail(c, HSFLOW_E_NOTERM, "ITER termination with max_iter <= 0 would never stop");
    if (!(p.alpha > 0.f) || !std::isfinite(p.alpha)) return fail(c, HSFLOW_E_ARG, "alpha must be positive");
    const float a2 = p.alpha * p.alpha; // Kernels.cl:85
    // which kernel: the register strip wherever the image has an aligned shape for it (classic_strip_geom), else the LDS tile
    int kernel = p.kernel;
    if (kernel != HSFLOW_KERNEL_AUTO && kernel != HSFLOW_KERNEL_SIMPLE && kernel != HSFLOW_KERNEL_FUSED && kernel != HSFLOW_KERNEL_STRIP)
        return fail(c, HSFLOW_E_ARG, "CLASSIC mode has the simple, the fused (LDS tile) and the strip kernels only");
    int T = 1;
    if (kernel == HSFLOW_KERNEL_AUTO || kernel == HSFLOW_KERNEL_STRIP) {
        T = p.fuse_steps > 0 ? std::min(p.fuse_steps, kMaxFuse) : pick_classic_strip_T(c, p.max_iter, p.strip_rows, p.threads);
        const bool tiles_apply = T > 0 && (kernel == HSFLOW_KERNEL_STRIP || (!p.tile_w && !p.tile_h));
        if (T <= 0) T = 1;
        // (the strip kernels divide with a precomputed reciprocal: legal while alpha^2 is nowhere near the ends of the
        // exponent range, hs_kernels_classic_strip.hip.h)
        const bool alpha_ok = a2 >= 0x1p-40f && a2 <= 0x1p40f;
        bool ok = alpha_ok && tiles_apply && make_classic_strip_plan(c, T, p.strip_rows, p.threads, S.splan);
        const int rem = p.max_iter % T;
        if (ok && rem) ok = make_classic_strip_plan(c, rem, p.strip_rows, p.threads, S.stail);
        if (ok) kernel = HSFLOW_KERNEL_STRIP;
        else if (kernel == HSFLOW_KERNEL_STRIP)
            return fail(c, alpha_ok ? HSFLOW_E_SIZE : HSFLOW_E_ARG,
                        alpha_ok ? "CLASSIC mode: no strip shape for this image / fuse_steps / strip_rows / threads"
                                 : "CLASSIC mode: the strip kernel takes 2^-20 <= alpha <= 2^20");
        else kernel = HSFLOW_KERNEL_FUSED;
    }
    if (kernel == HSFLOW_KERNEL_FUSED) {
        // 18 LDS values per plane and group and an IEEE division make a sweep dearer than in CV mode:
        // the halo pays off up to about 6 sweeps per launch (tools/sweep_classic.py on MI355X)
        T = p.fuse_steps > 0 ? std::min(p.fuse_steps, kMaxFuse) : std::min(6, p.max_iter);
        if (!make_plan(c, T, p.tile_w, p.tile_h, p.threads, S.fplan))
            return fail(c, HSFLOW_E_SIZE, "no feasible tile for the requested fuse_steps / tile / threads");
        const int rem = p.max_iter % T;
        if (rem && !make_plan(c, rem, p.tile_w, p.tile_h, p.threads, S.ftail))
            return fail(c, HSFLOW_E_SIZE, "no feasible tile for the tail launch");
    }
    if (kernel == HSFLOW_KERNEL_SIMPLE) T = 1;
    S.kernel = kernel;
    S.T = T;
    hsflow_info &i = c->info;
    i.kernel = kernel; i.fuse_steps = T;
    if (kernel == HSFLOW_KERNEL_STRIP) {
        i.tile_w = S.splan.g.CW; i.tile_h = S.splan.g.CH; i.threads = S.splan.g.NW * 64;
        i.groups_per_thread = S.splan.R; i.tiles = S.splan.tiles; i.lds_bytes = S.splan.lds_bytes;
    } else if (kernel == HSFLOW_KERNEL_FUSED) {
        i.tile_w = S.fplan.g.CW; i.tile_h = S.fplan.g.CH; i.threads = S.fplan.NT;
        i.groups_per_thread = S.fplan.K; i.tiles = S.fplan.tiles; i.lds_bytes = S.fplan.lds_bytes;
    } else {
        i.tile_w = i.tile_h = 0; i.threads = 256; i.groups_per_thread = 1; i.tiles = 0; i.lds_bytes = 0;
    }
    i.jacobi_launches = (p.max_iter + T - 1) / T;
    return HSFLOW_OK;
}

int solve_classic(hsflow_ctx *c, const hsflow_params &p, bool async)
{
    ClassicSetup S;
    int st = prepare_classic(c, p, S);
    if (st) return st;
    if (p.profile && (p.use_graph || async)) return fail(c, HSFLOW_E_ARG, "CLASSIC mode: profiling needs a synchronous solve without use_graph");
    Profiler prof{c, p.profile != 0};
    const dim3 grid((c->W + 255) / 256, (c->H + 3) / 4, c->N), block(64, 4);
    const bool do_deriv = !(p.reuse_derivatives && c->coef_valid && c->coef_mode == HSFLOW_MODE_CLASSIC);
    const float a2 = p.alpha * p.alpha; // Kernels.cl:85
    const bool write_v = p.mode != HSFLOW_MODE_CLASSIC_AS_SHIPPED;
    const bool zero0 = !p.use_previous;
    const int kernel = S.kernel, T = S.T;
    const ClassicStripPlan &splan = S.splan, &stail = S.stail;
    const FusedPlan &fplan = S.fplan, &ftail = S.ftail;
    // The derivatives live as one packed word per pixel (k_deriv_classic_packed; what the strip kernel loads); the
    // LDS-tile and the one-sweep kernels read three fp32 planes, unpacked once per derivative pass when they run.
    const bool need_planes = kernel != HSFLOW_KERNEL_STRIP;
    if (need_planes) {
        const size_t px = (size_t)c->plane * c->N;
        for (int i = 0; i < 3; i++)
            if (!c->dE[i]) HS_HIP(c, hipMalloc((void **)&c->dE[i], px * sizeof(float)));
    }
    const bool do_unpack = need_planes && (do_deriv || !c->dE_valid);
    auto enqueue = [&](int *n) -> int {
        if (do_deriv) {
            prof.begin(0);
            hipLaunchKernelGGL(hsk::k_deriv_classic_packed, grid, block, 0, c->stream, c->dA, c->dB, c->dCoef, c->W, c->H, c->P, c->plane);
            HS_HIP(c, hipGetLastError());
            prof.end();
        }
        if (do_unpack) {
            hipLaunchKernelGGL(hsk::k_unpack_classic_deriv, grid, block, 0, c->stream, c->dCoef, c->dE[0], c->dE[1], c->dE[2],
                               c->W, c->H, c->P, c->plane);
            HS_HIP(c, hipGetLastError());
        }
        int zero = zero0 ? 1 : 0;
        if (zero) c->cur = 0;
        int done = 0, launches = 0;
        while (done < p.max_iter) {
            const int chunk = std::min(T, p.max_iter - done);
            const int a = c->cur, b = a ^ 1;
            prof.begin(1);
            hipError_t e;
            if (kernel == HSFLOW_KERNEL_STRIP) {
                ClassicStripPlan cp = chunk == T ? splan : stail;
                cp.g.zero_in = zero;
                e = launch_classic_strip(c, cp, write_v, c->dU[a], c->dV[a], c->dU[b], c->dV[b], a2);
            } else if (kernel == HSFLOW_KERNEL_FUSED) {
                FusedPlan cp = chunk == T ? fplan : ftail;
                cp.g.zero_in = zero;
                e = launch_classic_fused(c, cp, write_v, c->dU[a], c->dV[a], c->dU[b], c->dV[b], a2);
            } else {
                auto kern = zero ? (write_v ? hsk::k_jacobi_classic<true, true> : hsk::k_jacobi_classic<true, false>)
                                 : (write_v ? hsk::k_jacobi_classic<false, true> : hsk::k_jacobi_classic<false, false>);
                hipLaunchKernelGGL(kern, grid, block, 0, c->stream, c->dE[0], c->dE[1], c->dE[2], c->dU[a], c->dV[a], c->dU[b], c->dV[b],
                                   c->W, c->H, c->P, c->plane, a2);
                e = hipGetLastError();
            }
            prof.end();
            HS_HIP(c, e);
            c->cur = b;
            zero = 0;
            done += chunk;
            launches++;
        }
        *n = launches;
        return HSFLOW_OK;
    };
    auto configure = [&]() -> int { // kernel attributes cannot be set inside a capture
        if (kernel == HSFLOW_KERNEL_STRIP) {
            HS_HIP(c, launch_classic_strip(c, splan, write_v, nullptr, nullptr, nullptr, nullptr, a2, true));
            if (p.max_iter % T) HS_HIP(c, launch_classic_strip(c, stail, write_v, nullptr, nullptr, nullptr, nullptr, a2, true));
        } else if (kernel == HSFLOW_KERNEL_FUSED) {
            HS_HIP(c, launch_classic_fused(c, fplan, write_v, nullptr, nullptr, nullptr, nullptr, a2, true));
            if (p.max_iter % T) HS_HIP(c, launch_classic_fused(c, ftail, write_v, nullptr, nullptr, nullptr, nullptr, a2, true));
        }
        return HSFLOW_OK;
    };
    hsflow_info &i = c->info;
    int launches = 0;
    if (p.use_graph) {
        GraphKey key{p.mode, kernel, p.max_iter, T, i.tile_w, i.tile_h, i.threads, i.groups_per_thread, zero0 ? 0 : c->cur,
                     p.use_previous * 2 + (do_deriv ? 1 : 0) + (do_unpack ? 4 : 0), p.alpha};
        if ((st = run_captured(c, key, configure, enqueue, &launches))) return st;
    } else if ((st = enqueue(&launches))) return st;
    c->coef_valid = true;
    c->coef_mode = HSFLOW_MODE_CLASSIC;
    c->dE_valid = do_unpack || (c->dE_valid && !do_deriv);
    i.jacobi_launches = launches;
    i.iterations_done = p.max_iter; i.last_eps = 0.f; i.deriv_ms = i.jacobi_ms = i.solve_ms = 0.f;
    if (!async) {
        HS_HIP(c, hipStreamSynchronize(c->stream));
        prof.collect();
    }
    return HSFLOW_OK;
}


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
    const bool persist = S.persist;
    const int rem = (multi && !persist) ? iters % T : 0;
    if (rem && !make_jplan(c, kernel, rem, p, tail))
        return fail(c, HSFLOW_E_SIZE, "no feasible launch plan for the tail launch");
    const bool zero = !p.use_previous;
    const bool do_deriv = !(p.reuse_derivatives && c->coef_valid && c->coef_mode == HSFLOW_MODE_CV);
    c->info.deriv_fused = do_deriv && !p.profile && multi && strip_deriv_fusable(c, iters >= T ? plan : tail); // enqueue_fixed's rule
    if (persist) { // buffers and phase counters: outside any capture
        if ((st = persist_reserve(c)) || (st = persist_prepare_flags(c, plan.s.tiles))) return st;
    }
    if (p.use_graph && !p.profile) {
        GraphKey key{p.mode, persist ? HSFLOW_KERNEL_PERSIST : kernel, iters, T, c->info.tile_w, c->info.tile_h, c->info.threads,
                     c->info.groups_per_thread, zero ? 0 : c->cur, p.use_previous * 2 + (do_deriv ? 1 : 0), coeff};
        auto configure = [&]() -> int {
            if (persist) {
                hsk::PersistArgs none{};
                HS_HIP(c, launch_persist(c, plan.s, none, 0, c->info.deriv_fused != 0, nullptr, nullptr, coeff, true));
                return HSFLOW_OK;
            }
            if (multi) {
                HS_HIP(c, launch_j(c, plan, false, nullptr, nullptr, nullptr, nullptr, coeff, true));
                if (rem) HS_HIP(c, launch_j(c, tail, false, nullptr, nullptr, nullptr, nullptr, coeff, true));
                const JPlan &first = iters >= T ? plan : tail;
                if (do_deriv && strip_deriv_fusable(c, first))
                    HS_HIP(c, launch_j(c, first, false, nullptr, nullptr, nullptr, nullptr, coeff, true, 0, true));
            }
            return HSFLOW_OK;
        };
        auto enqueue = [&](int *n) -> int {
            const int e = enqueue_fixed(c, p, coeff, iters, kernel, T, &plan, &tail, prof, do_deriv, zero, persist);
            *n = c->info.jacobi_launches;
            return e;
        };
        int n = 0;
        if ((st = run_captured(c, key, configure, enqueue, &n))) return st;
        c->info.jacobi_launches = n;
    } else {
        if (persist) {
            hsk::PersistArgs none{};
            HS_HIP(c, launch_persist(c, plan.s, none, 0, c->info.deriv_fused != 0, nullptr, nullptr, coeff, true));
        }
        st = enqueue_fixed(c, p, coeff, iters, kernel, T, &plan, &tail, prof, do_deriv, zero, persist);
        if (st) return st;
    }
    c->coef_valid = true;
    c->coef_mode = HSFLOW_MODE_CV;
    c->info.iterations_done = iters;
    if (persist && async) c->persist_unchecked = true; // looked at when the stream is next drained (check_persist)
    if (!async) {
        HS_HIP(c, hipStreamSynchronize(c->stream));
        if (persist && persist_error(c)) { // a wait timed out: back to a launch per fuse_steps iterations, for good
            persist_failed(c);
            c->cur ^= 1;            // the starting flow is intact (the phases wrote the other two buffers)
            c->coef_valid = false;  // a workgroup that never started left its part of the derivative plane unwritten
            hsflow_params q = p;
            if (q.kernel == HSFLOW_KERNEL_PERSIST) q.kernel = HSFLOW_KERNEL_STRIP;
            q.reuse_derivatives = 0;
            return solve_impl(c, &q, false);
        }
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
    bool witness = (kernel == HSFLOW_KERNEL_STRIP || kernel == HSFLOW_KERNEL_FOLD) && !c->force_exact && strip_has_witness(plan);
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
        if (witness && !strip_has_witness(tailp)) { // as in prepare_solve
            JPlan alt;
            if (make_witness_jplan(c, kernel, iters % T, p, alt)) tailp = alt;
        }
        stride = std::max(stride, plan_eps_stride(kernel, tailp));
        // (a tail of one sweep is measured, not witnessed: the synchronous pass's mode 3; an asynchronous pass needs it able)
        if (witness && !S.persist && (async || iters % T > 1) && !strip_has_witness(tailp)) witness = false; // (persist: the tail phase keeps the plan's geometry)
    }
    if (async && !witness)
        return fail(c, HSFLOW_E_ARG, "solve_async with ITER|EPS: this launch plan (core tile thinner than a strip) cannot run witness launches; "
                                     "use hsflow_solve or other tuning parameters");
    int launches = 0;
    if (witness) {
        // Witness pass: all launches but the last run k_jacobi_strip<.., 2>, which costs almost
        // nothing over the ITER-only kernel and yields one number per launch that proves "Eps >=
        // epsilon in every one of my sweeps" when it is >= epsilon.  The last launch also provides
        // last_eps: it witnesses its sweeps but measures the final one (mode 3).  If every bound
        // holds, the early stop cannot have fired before the budget ran out (a stop AT the final
        // sweep is the budget) and the result stands.  Otherwise (a flat or converged input) the
        // exact per-sweep path below starts over from the saved flow.
        // threshold as the smallest float >= epsilon: "change >= epsThr" then implies "Eps >= epsilon"
        c->epsThr = p.epsilon > 0 ? (float)p.epsilon : 0.f;
        if ((double)c->epsThr < p.epsilon) c->epsThr = std::nextafterf(c->epsThr, INFINITY);
        if (c->epsThr < FLT_MIN) c->epsThr = c->epsThr > 0.f ? FLT_MIN : 0.f; // the kernels scale it through its exponent bits
        const bool persist = S.persist && async; // (prepare_solve grants it to asynchronous solves only)
        const int n_launch = (iters + T - 1) / T; // persist: phases of the one launch; the witness words are laid out alike
        const int last_chunk = iters - (n_launch - 1) * T;
        if (persist && ((st = persist_reserve(c)) || (st = persist_prepare_flags(c, plan.s.tiles)))) return st;
        // Synchronous solves report last_eps at once: their last launch measures its final sweep (mode 3: two words
        // per workgroup, the witness and that sweep's Eps).  Asynchronous solves run witness launches only; their
        // last_eps is measured if and when hsflow_get_info asks for it (measure_last_eps).
        const int last_mode = async ? 2 : 3;
        const int slots = async ? n_launch : (n_launch - 1) + 2;
        if ((st = eps_reserve(c, slots, stride))) return st;
        const int cur0 = c->cur;
        // the first launch also does the derivative pass where the kernel can (hs_plan_launch.hip.h)
        const JPlan &firstp = (n_launch == 1 && last_chunk != T) ? tailp : plan;
        const bool fuse_deriv = do_deriv && !p.profile && strip_deriv_fusable(c, firstp);
        c->info.deriv_fused = fuse_deriv;
        // the whole pass as one enqueue sequence (nothing allocated, nothing synchronised: capturable)
        auto enqueue = [&]() -> int {
            int e0 = save_start();
            if (e0) return e0;
            c->epsStride = stride; // (no clearing: every launch writes all its words, the reduction reads only those)
            bool fuse = fuse_deriv;
            if (do_deriv && !fuse) {
                prof.begin(0);
                HS_HIP(c, launch_deriv(c));
                prof.end();
            }
            int zero_w = p.use_previous ? 0 : 1;
            if (zero_w) c->cur = 0;
            if (persist) { // one launch, a row of witness words per phase
                c->epsPtr = c->dEpsTiles;
                prof.begin(1);
                const int e = enqueue_persist(c, plan.s, iters, 2, fuse, zero_w, coeff);
                prof.end();
                if (e) return e;
                // what measure_last_eps needs: the last phase again as an ordinary launch, from the third buffer
                c->lastl.plan = last_chunk != T ? tailp : plan; c->lastl.zero_in = 0; c->lastl.coeff = coeff; c->lastl.eps_thr = c->epsThr;
                c->lastl.from_third = true;
                launches = 1;
            }
            for (int L = persist ? n_launch : 0; L < n_launch; L++) {
                const bool is_last = L == n_launch - 1;
                const JPlan &cp = (is_last && last_chunk != T) ? tailp : plan;
                const int a = c->cur, b = a ^ 1;
                c->epsPtr = c->dEpsTiles + (size_t)L * stride;
                prof.begin(1);
                hipError_t e = launch_j(c, cp, is_last ? last_mode : 2, c->dU[a], c->dV[a], c->dU[b], c->dV[b], coeff, false, zero_w, fuse);
                prof.end();
                HS_HIP(c, e);
                if (is_last && async) { // what measure_last_eps needs
                    c->lastl.plan = cp; c->lastl.zero_in = zero_w; c->lastl.coeff = coeff; c->lastl.eps_thr = c->epsThr;
                    c->lastl.from_third = false;
                }
                c->cur = b;
                zero_w = 0;
                fuse = false;
                launches++;
            }
            if (async && !c->async_reduce) { // the reduction of the witness words waits until somebody settles the check (settle_pending)
                c->epsPtr = c->dEps;
                c->epsStride = 1;
                return HSFLOW_OK;
            }
            // (an asynchronous solve gets here only with the in-stream reduction on: its last workgroup writes the marker too)
            return eps_collect_enqueue(c, slots, n_launch - 1, plan_eps_stride(kernel, plan),
                                       plan_eps_stride(kernel, (last_chunk != T && !persist) ? tailp : plan), async && c->hMark != nullptr);
        };
        if (p.use_graph && !p.profile) {
            GraphKey key{p.mode, persist ? HSFLOW_KERNEL_PERSIST : kernel, iters, T, c->info.tile_w, c->info.tile_h, c->info.threads,
                         c->info.groups_per_thread, p.use_previous ? c->cur : 0, p.use_previous * 2 + (do_deriv ? 1 : 0) + (async ? 4 : 0) + (async && c->async_reduce ? 8 : 0), coeff, c->epsThr};
            auto configure = [&]() -> int {
                if (persist) {
                    hsk::PersistArgs none{};
                    HS_HIP(c, launch_persist(c, plan.s, none, 2, fuse_deriv, nullptr, nullptr, coeff, true));
                    HS_HIP(c, launch_j(c, has_tail ? tailp : plan, 3, nullptr, nullptr, nullptr, nullptr, coeff, true)); // measure_last_eps
                    return HSFLOW_OK;
                }
                HS_HIP(c, launch_j(c, plan, 2, nullptr, nullptr, nullptr, nullptr, coeff, true));
                HS_HIP(c, launch_j(c, plan, last_mode, nullptr, nullptr, nullptr, nullptr, coeff, true));
                if (has_tail) HS_HIP(c, launch_j(c, tailp, last_mode, nullptr, nullptr, nullptr, nullptr, coeff, true));
                if (fuse_deriv) HS_HIP(c, launch_j(c, firstp, n_launch == 1 ? last_mode : 2, nullptr, nullptr, nullptr, nullptr, coeff, true, 0, true));
                return HSFLOW_OK;
            };
            const hsflow_ctx::LastLaunch keep = c->lastl;
            bool captured = false;
            auto enqueue_n = [&](int *n) -> int {
                launches = 0;
                captured = true;
                const int e = enqueue();
                *n = launches;
                return e;
            };
            if ((st = run_captured(c, key, configure, enqueue_n, &launches))) return st;
            c->epsPtr = c->dEps;
            c->epsStride = 1;
            if (async && !captured) { // a replay: the description of the last launch is what the capture recorded
                const bool is_tail = last_chunk != T;
                c->lastl = keep;
                c->lastl.plan = is_tail ? tailp : plan;
                c->lastl.zero_in = (n_launch == 1 && !p.use_previous) ? 1 : 0;
                c->lastl.coeff = coeff;
                c->lastl.eps_thr = c->epsThr;
                c->lastl.from_third = persist;
            }
        } else {
            if (persist) {
                hsk::PersistArgs none{};
                HS_HIP(c, launch_persist(c, plan.s, none, 2, fuse_deriv, nullptr, nullptr, coeff, true));
            }
            if ((st = enqueue())) return st;
        }
        c->coef_valid = true;
        c->coef_mode = HSFLOW_MODE_CV;
        if (persist) c->persist_unchecked = true;
        if (async) { // the check is owed: hsflow_synchronize (or the next call that needs results) settles it
            c->lastl.valid = true;
            c->pend.active = true;
            c->pend.params = p;
            c->pend.iters = iters; c->pend.slots = slots; c->pend.launches = launches; c->pend.cur0 = cur0;
            c->pend.stride = stride; c->pend.n_first = n_launch - 1; c->pend.cnt_first = plan_eps_stride(kernel, plan);
            c->pend.cnt_last = plan_eps_stride(kernel, (last_chunk != T && !persist) ? tailp : plan);
            c->pend.reduced = c->async_reduce;
            c->pend.marked_by_reduce = c->async_reduce && c->hMark != nullptr;
            c->info.iterations_done = iters;
            c->info.jacobi_launches = launches;
            return HSFLOW_OK;
        }
        HS_HIP(c, hipStreamSynchronize(c->stream));
        std::vector<unsigned> hw(c->hEps, c->hEps + slots);

        float last = 0.f;
        if (witness_proven(hw.data(), slots, p.epsilon, &last, true)) {
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
    c->sweep_eps.resize((size_t)iters); // (hsflow_solve_probe hands these out)
    std::memcpy(c->sweep_eps.data(), heps.data(), (size_t)iters * sizeof(float));
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
    // Without a sweep budget the only exit is Eps < epsilon.  A positive epsilon below the fp32 limit cycle of
    // the iteration (Eps stalls around 1e-7 * |flow|) would keep the host launching for ever -- the original
    // does exactly that; here the solve gives up with HSFLOW_E_NOTERM (flow, iterations_done and last_eps
    // stay valid) once Eps has not reached a new minimum for kStallSweeps sweeps or after kMaxSweeps.
    constexpr long long kStallSweeps = 4096, kMaxSweeps = 1LL << 24;
    float best_eps = INFINITY;
    long long best_at = 0;
    bool stalled = false;
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
            if (e < best_eps) { best_eps = e; best_at = done + s; }
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
        if (!stop && budget > kMaxSweeps && (done - best_at >= kStallSweeps || done >= kMaxSweeps)) stop = stalled = true;
    }
    HS_HIP(c, hipStreamSynchronize(c->stream));
    c->info.iterations_done = (int)std::min<long long>(done, INT32_MAX);
    c->info.last_eps = last;
    c->info.jacobi_launches = launches;
    prof.collect();
    if (stalled)
        return fail(c, HSFLOW_E_NOTERM, "EPS termination: Eps stopped decreasing above epsilon (fp32 limit cycle) -- "
                                        "the flow of the sweeps done so far is kept");
    return HSFLOW_OK;
}

// CV mode: argument checks, kernel choice (AUTO rule) and launch plan.  Touches no device state, so the
// planner can also be queried without a GPU (hsflow_plan_query).  Fills c->info's plan fields.
int prepare_solve(hsflow_ctx *c, const hsflow_params &p, bool async, SolveSetup &S)
{
    const bool use_iter = (p.term_type & HSFLOW_TERM_ITER) != 0, use_eps = (p.term_type & HSFLOW_TERM_EPS) != 0;
    if (!use_iter && !use_eps) return fail(c, HSFLOW_E_ARG, "term_type must include ITER and/or EPS");
    if (use_iter && p.max_iter <= 0 && !use_eps)
        return fail(c, HSFLOW_E_NOTERM, "ITER termination with max_iter <= 0 would never stop");
    if (!(p.lambda > 0.f) || !std::isfinite(p.lambda)) return fail(c, HSFLOW_E_ARG, "lambda must be positive");
    // EPS with no sweep budget (EPS alone, or ITER|EPS with max_iter <= 0, which the original treats the same way,
    // cv210.dll VA 0x1012f10b-0x1012f14a) stops only on Eps < epsilon: with epsilon <= 0 or NaN that never happens
    // and the original spins forever.  Refused here; a positive epsilon below the fp32 limit cycle of the
    // iteration is caught at run time (solve_eps_chunks).
    if (use_eps && !(use_iter && p.max_iter > 0) && !(p.epsilon > 0.0 && std::isfinite(p.epsilon)))
        return fail(c, HSFLOW_E_NOTERM, "EPS termination without a sweep budget needs a finite epsilon > 0");
    if (async && p.profile) return fail(c, HSFLOW_E_ARG, "solve_async does not support profiling");

    // Ilambda = fl32(1/fl32(lambda)), cv210.dll VA 0x1012e054-0x1012e085.  Kept out of the denormal range
    // (lambda > 8.5e37): v_rsq_f32 in sweep_coefs flushes denormals, and where it matters -- a pixel with
    // Ix = Iy = 0 -- any finite value gives the same update (al = be = 0).
    const float coeff = std::max(1.0f / p.lambda, FLT_MIN);
    // AUTO: the register-strip kernel; below ~1.5 Mpixel per context its folded form (128-column strips:
    // twice the tiles across, so small frames reach more CUs -- measured 5-25 % faster from 160x120 to
    // 1600x900 at 100 sweeps, tools/crossover.py).
    bool small_frame = (long long)c->W * c->H * c->N <= 1500000LL;
    // A context that plans for a share of the chip (hsflow_set_cu_share: pair pipeline slots) wants the cheaper of the two
    // in CU-time, whatever the frame size: the folded kernel's 128-column strips pay twice the column halo.
    if (plan_shared(c) && p.kernel == HSFLOW_KERNEL_AUTO && (p.term_type & HSFLOW_TERM_ITER) && p.max_iter > 0 && p.max_iter <= (1 << 16) && p.fuse_steps <= 0) {
        double cs = 1e300, cf = 1e300;
        pick_strip_T(c, p.max_iter, p, 0, &cs);
        pick_strip_T(c, p.max_iter, p, 1, &cf);
        small_frame = cf < cs;
    }
    // PERSIST is the strip kernel as one launch per solve; AUTO takes it where it can run (persist_obstacle)
    const bool persist_asked = p.kernel == HSFLOW_KERNEL_PERSIST;
    if (use_eps && eps_windowed(c)) small_frame = false; // (the Eps row window is the strip kernel's)
    const int kernel = persist_asked ? HSFLOW_KERNEL_STRIP
                                     : p.kernel != HSFLOW_KERNEL_AUTO ? p.kernel : (small_frame ? HSFLOW_KERNEL_FOLD : HSFLOW_KERNEL_STRIP);
    if (kernel != HSFLOW_KERNEL_SIMPLE && kernel != HSFLOW_KERNEL_FUSED && kernel != HSFLOW_KERNEL_STRIP &&
        kernel != HSFLOW_KERNEL_FOLD)
        return fail(c, HSFLOW_E_ARG, "unknown kernel selector");
    if (use_eps && eps_windowed(c) && (kernel == HSFLOW_KERNEL_FOLD || kernel == HSFLOW_KERNEL_FUSED))
        return fail(c, HSFLOW_E_ARG, "EPS termination over a row window (hsflow_set_eps_rows) runs on the strip or the simple kernel");
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
    hsflow_params eff = p;
    if (multi) {
        const int horizon = budget > (1 << 30) ? 64 : (int)budget; // EPS-only runs: plan for chunks
        if (p.fuse_steps > 0) T = std::min(p.fuse_steps, kMaxFuse);
        else if (kernel == HSFLOW_KERNEL_STRIP || kernel == HSFLOW_KERNEL_FOLD)
            T = (use_eps && !(use_iter && p.max_iter > 0)) ? std::min(8, horizon)
                                                          : pick_strip_T(c, horizon, p, kernel == HSFLOW_KERNEL_FOLD);
        else T = pick_T(horizon, 0);
        if (budget < T) T = (int)budget;
        // The strip kernels carry 4^k * u inside a launch (hs_kernels_strip.hip.h).  The largest flow and
        // constant term a pixel can have grow like sqrt(lambda); beyond lambda = 1e20 keep the launches
        // short so that 4^(T+1) times those stays far inside the float range.
        if ((kernel == HSFLOW_KERNEL_STRIP || kernel == HSFLOW_KERNEL_FOLD) && coeff < 1e-20f) T = std::min(T, 8);
        if (!make_jplan(c, kernel, T, p, plan))
            return fail(c, HSFLOW_E_SIZE, "no feasible launch plan for the requested tile/threads/rows/fuse_steps");
        // ITER|EPS runs witness launches, which watch an edge row of each strip: a plan whose core tile is thinner
        // than a strip may have no wavefront with a core row there (thin frames, very short launches); another shape
        // then takes its place as far as the caller left rows / wavefronts open (make_witness_jplan).
        if (use_eps && use_iter && p.max_iter > 0 && (kernel == HSFLOW_KERNEL_STRIP || kernel == HSFLOW_KERNEL_FOLD) &&
            !strip_has_witness(plan)) {
            JPlan alt;
            if (make_witness_jplan(c, kernel, T, p, alt)) plan = alt;
        }
        plan_to_info(c, plan);
    } else {
        c->info.fuse_steps = 1; c->info.tile_w = c->info.tile_h = 0; c->info.threads = 256;
        c->info.groups_per_thread = 1; c->info.tiles = 0; c->info.lds_bytes = 0;
    }
    c->info.kernel = kernel;
    bool persist = false;
    // AUTO keeps the launch per fuse_steps: measured on MI355X at 1080p / 100 the persistent launch spends as long at a
    // phase boundary (write-through publish 2.6 us + counters 3.2 us + halo reload 4 us) as the stream does at a kernel
    // boundary -- ITER 0.148 against 0.149 ms, ITER|EPS slower -- so it runs on request only
    // (HSFLOW_PERSIST_AUTO=1 lets AUTO take it, for experiments).
    static const bool persist_auto = getenv("HSFLOW_PERSIST_AUTO") != nullptr;
    if (persist_asked || (persist_auto && p.kernel == HSFLOW_KERNEL_AUTO && kernel == HSFLOW_KERNEL_STRIP)) {
        const char *why = !(use_iter && p.max_iter > 0 && budget <= (1 << 16)) ? "needs a sweep budget (ITER)"
                                                                               : persist_obstacle(c, plan.s, (int)budget, p, async, use_eps);
        if (!why && use_eps && !strip_has_witness(plan)) why = "this launch plan cannot run witness phases";
        if (!why) persist = true;
        else if (persist_asked) return fail(c, HSFLOW_E_SIZE, std::string("HSFLOW_KERNEL_PERSIST: ") + why);
    }
    c->info.persistent = persist ? (int)((budget + T - 1) / T) : 0;
    S = SolveSetup{coeff, kernel, multi, use_iter, use_eps, budget, T, plan, persist, eff};
    return HSFLOW_OK;
}

int solve_impl_inner(hsflow_ctx *c, const hsflow_params *pp, bool async, bool *took_over);

int solve_impl(hsflow_ctx *c, const hsflow_params *pp, bool async)
{
    // A solve that takes an owed early-stop check over (below) drops it first; should it then fail before it has
    // registered its own, the owed check comes back -- the flow of the earlier solve is still unverified.
    const hsflow_ctx::Pending owed = c ? c->pend : hsflow_ctx::Pending();
    bool took_over = false;
    const int st = solve_impl_inner(c, pp, async, &took_over);
    if (st && took_over && !c->pend.active) c->pend = owed;
    if (c) c->last_marked = false;
    if (!st && async && c->async_reduce && c->hMark) { // a marker behind everything this solve enqueued (hsflow_wait_solve, settle_pending)
        const bool by_reduce = c->pend.active && c->pend.marked_by_reduce; // (the reduction kernel of an ITER|EPS solve wrote it)
        if (!by_reduce) hipLaunchKernelGGL(hsk::k_mark_done, dim3(1), dim3(64), 0, c->stream, c->dSeq, c->hMarkDev);
        if (by_reduce || hipGetLastError() == hipSuccess) {
            c->mark_issued++;
            c->last_marked = true;
            if (c->pend.active) c->pend.mark = c->mark_issued;
        } else if (c->pend.active) c->pend.reduced = false; // (no marker: the owed check is settled the slow way)
    }
    return st;
}

int solve_impl_inner(hsflow_ctx *c, const hsflow_params *pp, bool async, bool *took_over)
{
    int st = check_ctx(c, 0);
    if (st) return st;
    // An unverified asynchronous solve comes first -- unless this call repeats it exactly: while a check is owed
    // the inputs cannot have changed (every entry point that changes frames or flow settles first), so an
    // asynchronous solve with bit-identical parameters that starts from zero flow recomputes the very same
    // result and leaves the very same witness words; the owed check simply passes on to it and the host does
    // not wait for the stream (a caller that streams solves never pays a round trip per solve).
    const bool repeat = c->pend.active && async && !c->force_exact && pp && pp->struct_size == sizeof(hsflow_params) &&
                        !pp->use_previous && std::memcmp(pp, &c->pend.params, sizeof(hsflow_params)) == 0;
    if (repeat) { c->pend.active = false; *took_over = true; }
    else if ((st = settle_pending(c))) return st;
    c->lastl.valid = false;
    if (!pp || pp->struct_size != sizeof(hsflow_params))
        return fail(c, HSFLOW_E_ARG, "params null or struct_size mismatch");
    const hsflow_params &p = *pp;
    if (!c->frames_set) return fail(c, HSFLOW_E_STATE, "frames were not set");
    if (p.mode == HSFLOW_MODE_CLASSIC || p.mode == HSFLOW_MODE_CLASSIC_AS_SHIPPED) return solve_classic(c, p, async);
    if (p.mode != HSFLOW_MODE_CV) return fail(c, HSFLOW_E_ARG, "unknown mode");
    c->info.eps_rerun = 0;
    c->info.deriv_fused = 0;
    // The plan of a solve depends on the parameters, not on the frames: a stream of solves with the same parameters plans
    // once (the planner tries every sweep count x rows x wavefronts: tens of microseconds of host time per solve, which is
    // what bounded a stream of small frames).  Not cached: a persistent launch (it depends on who else is alive).
    SolveSetup S;
    PlanKey key;
    std::memset(&key, 0, sizeof(key)); // (compared bytewise: padding too)
    key.p = p;
    key.p.use_previous = key.p.reuse_derivatives = key.p.use_graph = key.p.profile = 0; // (do not enter the plan)
    key.async = async ? 1 : 0;
    key.exact = c->force_exact ? 1 : 0;
    const bool cacheable = p.kernel != HSFLOW_KERNEL_PERSIST && !getenv("HSFLOW_PERSIST_AUTO");
    const PlanEntry *hit = nullptr;
    if (cacheable)
        for (const PlanEntry &e : c->plan_cache)
            if (std::memcmp(&e.key, &key, sizeof(key)) == 0) { hit = &e; break; }
    if (hit) {
        S = hit->S;
        S.eff = p; // (the cached copy carries the first caller's use_previous / reuse_derivatives / use_graph)
        hsflow_info &i = c->info;
        const hsflow_info &j = hit->info;
        i.kernel = j.kernel; i.fuse_steps = j.fuse_steps; i.tile_w = j.tile_w; i.tile_h = j.tile_h; i.threads = j.threads;
        i.groups_per_thread = j.groups_per_thread; i.tiles = j.tiles; i.lds_bytes = j.lds_bytes; i.persistent = j.persistent;
    } else {
        if ((st = prepare_solve(c, p, async, S))) return st;
        if (cacheable) {
            if (c->plan_cache.size() >= 16) c->plan_cache.erase(c->plan_cache.begin());
            c->plan_cache.push_back(PlanEntry{key, S, c->info});
        }
    }
    c->info.deriv_ms = c->info.jacobi_ms = c->info.solve_ms = 0.f;
    c->info.last_eps = 0.f;
    Profiler prof{c, p.profile != 0};
    if (!S.use_eps) return solve_fixed(c, S.eff, S, prof, async);
    constexpr long long kSpecMax = 1 << 16; // speculative ITER|EPS: the whole budget in one go
    if (S.use_iter && p.max_iter > 0 && S.budget <= kSpecMax) return solve_iter_eps(c, S.eff, S, prof, async);
    return solve_eps_chunks(c, S.eff, S, prof);
}

int copy_frame_in(hsflow_ctx *c, uint8_t *dst, const void *src, size_t stride, hipMemcpyKind kind, bool sync)
{
    if (sync) HS_HIP(c, hipMemcpy2D(dst, c->P, src, stride, c->W, c->H, kind));
    else HS_HIP(c, hipMemcpy2DAsync(dst, c->P, src, stride, c->W, c->H, kind, c->stream));
    return HSFLOW_OK;
}

} // namespace
