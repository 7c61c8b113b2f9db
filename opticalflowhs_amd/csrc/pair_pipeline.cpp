// Pair pipeline: independent image pairs streamed through one GPU with the host<->device copies
// of neighbouring pairs overlapped with the solve of the current one (include/hsflow.h).
//
// Reference behaviour replaced: HSOpticalFlowOpenCL::run() handles a pair as blocking write ->
// derivatives -> iterations -> blocking read (HSOpticalFlowOpenCL.cpp:744-767); nothing overlaps.
// Here `depth` single-pair contexts, each on a private stream, are used round-robin; the copy
// engines and the compute queue work on three different pairs at once.  Built purely on the public
// C ABI, so it is also the example of how a host drives several contexts from one thread.
#include "../../include/hsflow.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <new>
#include <string>
#include <vector>

struct hsflow_pipeline {
    struct Slot {
        hsflow_ctx *ctx = nullptr;
        bool busy = false;
        uint64_t ticket = 0;
        float *u = nullptr, *v = nullptr; // where the running job's flow goes
        size_t us = 0, vs = 0;
        hsflow_info done;                 // of the last job that finished on this slot
        bool has_done = false;
        uint64_t done_ticket = 0;
        hipStream_t stream = nullptr;     // the lane this slot works on (shared with the slots k +- lanes)
        hipEvent_t ev = nullptr;          // behind everything the running job enqueued: what finish_slot waits for
    };
    std::vector<Slot> slots;
    std::vector<hipStream_t> lanes;
    int device = 0;
    uint64_t next = 0;
    int width = 0, height = 0;
    std::string err;
};

namespace {

thread_local std::string g_pipeline_create_error;

int pfail(hsflow_pipeline *pl, int code, const std::string &msg)
{
    if (pl) pl->err = msg; else g_pipeline_create_error = msg;
    return code;
}

// Forwards a context error into the pipeline's error text.
int ctx_fail(hsflow_pipeline *pl, hsflow_ctx *ctx, int code, const char *what)
{
    return pfail(pl, code, std::string(what) + ": " + hsflow_last_error(ctx));
}

// The launch shape for a pair whose caller left everything to the planner, when three or more lanes (streams) share the chip and the
// frame is small: the planner shapes a solve for its own latency -- a 600x480 frame is cut into 210 tiles of 88x16 so
// that every CU gets one, at five times the halo arithmetic -- but with the other slots' solves running beside it the
// chip is full anyway and what counts is the CU-time a solve costs: few large tiles (strip kernel, 20 sweeps per launch,
// 5 rows per lane; 16 wavefronts, or 12 where that would leave fewer than ~50 tiles).  Measured at depth 8 on MI355X,
// ms per pair, planner's shape -> this one: 424x240 0.066 -> 0.051, 640x480 0.085 -> 0.053, 800x600 0.093 -> 0.061,
// 1280x720 0.114 -> 0.075 (depth 3: 0.107 -> 0.066); from 1080p on the planner's shape is already this one
// (profiles/r03_pipeline_shapes.txt).  Results are bit-identical whatever the shape.  HSFLOW_PIPELINE_AUTO_SHAPE=0: off.
hsflow_params stream_shape(const hsflow_pipeline *pl, const hsflow_params &in)
{
    static const bool off = getenv("HSFLOW_PIPELINE_AUTO_SHAPE") && atoi(getenv("HSFLOW_PIPELINE_AUTO_SHAPE")) == 0;
    hsflow_params p = in;
    const long long px = (long long)pl->width * pl->height;
    if (off || pl->lanes.size() < 3 || p.struct_size != sizeof(hsflow_params) || p.mode != HSFLOW_MODE_CV || p.kernel != HSFLOW_KERNEL_AUTO ||
        p.fuse_steps || p.strip_rows || p.threads || p.tile_w || p.tile_h || !(p.term_type & HSFLOW_TERM_ITER) || p.max_iter <= 0 ||
        px > 1500000LL || pl->width < 256 || pl->height < 80)
        return p;
    const int T = p.max_iter < 20 ? p.max_iter : 20, HX = (T + 3) / 4 * 4, CW = 256 - 2 * HX;
    if (CW < 64) return p;
    const int CH16 = 80 - 2 * T; // 16 wavefronts x 5 rows
    const long long tiles16 = CH16 > 0 ? (long long)((pl->width + CW - 1) / CW) * ((pl->height + CH16 - 1) / CH16) : 0;
    p.kernel = HSFLOW_KERNEL_STRIP;
    p.fuse_steps = T;
    p.strip_rows = 5;
    p.threads = tiles16 >= 50 ? 1024 : 768;
    return p;
}

int finish_slot(hsflow_pipeline *pl, hsflow_pipeline::Slot &s)
{
    if (!s.busy) return HSFLOW_OK;
    s.busy = false; // also on failure: the job is over either way
    // Only THIS job is waited for -- the event behind what it enqueued -- not the jobs other slots have queued on the same
    // lane since; hsflow_wait_solve then settles the early-stop check of an ITER|EPS solve (its witness words were reduced
    // in-stream, before the event).
    // A device-resident job carries no event at all (an event record between the solves of a stream costs the stream 6 %,
    // tools/event_cost.py): the marker kernel behind its solve is polled (hsflow_wait_solve).  A host-memory job's
    // downloads come after the solve, so the event behind them is what says "done".
    // (a slot that has its lane to itself -- hsflow_pipeline_create -- simply waits for its stream, no event needed)
    if (s.u) {
        const bool own_lane = pl->lanes.size() == pl->slots.size();
        if (hipSetDevice(pl->device) != hipSuccess || (own_lane ? hipStreamSynchronize(s.stream) : hipEventSynchronize(s.ev)) != hipSuccess)
            return pfail(pl, HSFLOW_E_DEVICE, "waiting for the slot's downloads failed");
    }
    int st = hsflow_wait_solve(s.ctx);
    if (st) return ctx_fail(pl, s.ctx, st, "hsflow_wait_solve");
    hsflow_info info;
    info.struct_size = sizeof(info);
    // (without last_eps: an asynchronous ITER|EPS solve measures it only on demand -- hsflow_pipeline_info)
    if ((st = hsflow_get_info_ex(s.ctx, &info, 0))) return ctx_fail(pl, s.ctx, st, "hsflow_get_info_ex");
    if (info.eps_rerun && s.u) { // the solve was repeated exactly: what the queued download copied is stale
        if ((st = hsflow_get_flow(s.ctx, 0, s.u, s.us, s.v, s.vs))) return ctx_fail(pl, s.ctx, st, "hsflow_get_flow");
    }                            // (a device-resident job keeps its flow in the slot: nothing to copy again)
    s.done = info; s.has_done = true; s.done_ticket = s.ticket;
    return HSFLOW_OK;
}

} // namespace

extern "C" {

int hsflow_pipeline_create(hsflow_pipeline **out, int device, int width, int height, int depth)
{
    return hsflow_pipeline_create_lanes(out, device, width, height, depth, depth);
}

int hsflow_pipeline_create_lanes(hsflow_pipeline **out, int device, int width, int height, int depth, int lanes)
{
    if (!out) return pfail(nullptr, HSFLOW_E_ARG, "out is null");
    *out = nullptr;
    if (depth < 1 || depth > 16) return pfail(nullptr, HSFLOW_E_ARG, "depth must be 1..16");
    if (lanes < 1 || lanes > depth) return pfail(nullptr, HSFLOW_E_ARG, "lanes must be 1..depth");
    hsflow_pipeline *pl = new (std::nothrow) hsflow_pipeline();
    if (!pl) return pfail(nullptr, HSFLOW_E_OOM, "host allocation failed");
    pl->slots.resize((size_t)depth);
    pl->width = width;
    pl->height = height;
    pl->device = device;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        (void)hipGetLastError();
        hsflow_pipeline_destroy(pl);
        return pfail(nullptr, HSFLOW_E_ARG, "hsflow_create: device ordinal out of range");
    }
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); hsflow_pipeline_destroy(pl); return pfail(nullptr, HSFLOW_E_DEVICE, "hipSetDevice failed"); }
    for (int l = 0; l < lanes; l++) {
        hipStream_t st_ = nullptr;
        if (hipStreamCreateWithFlags(&st_, hipStreamNonBlocking) != hipSuccess) { hsflow_pipeline_destroy(pl); return pfail(nullptr, HSFLOW_E_DEVICE, "hipStreamCreateWithFlags failed"); }
        pl->lanes.push_back(st_);
    }
    int k_ = 0;
    for (auto &s : pl->slots) {
        s.stream = pl->lanes[(size_t)(k_++ % lanes)];
        if (hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) != hipSuccess) { hsflow_pipeline_destroy(pl); return pfail(nullptr, HSFLOW_E_DEVICE, "hipEventCreateWithFlags failed"); }
        const int st = hsflow_create(&s.ctx, device, width, height, 1, (void *)s.stream, /*own_stream*/ 0);
        if (st) {
            g_pipeline_create_error = std::string("hsflow_create: ") + hsflow_last_error(nullptr);
            hsflow_pipeline_destroy(pl);
            return st;
        }
        hsflow_set_async_reduce(s.ctx, 1); // the slots' streams overlap: the witness words are reduced in-stream, settling is a wait and a look
        // HSFLOW_PIPELINE_CU_SHARE=<n>: every slot plans for n CUs (hsflow_set_cu_share) -- an experiment knob.  Off by
        // default: the planners' cost models were fitted to solves that have the chip to themselves, and with a share
        // they picked worse shapes than without at 1080p (0.161 against 0.139 ms per pair at depth 2) and at 600x480
        // (0.097 against 0.082), while a hand-picked large-tile shape does win at 600x480 / depth 8 (0.053 against 0.077:
        // profiles/r03_pipeline_shapes.txt).  A caller that knows its stream passes that shape in the params.
        if (const char *ov = getenv("HSFLOW_PIPELINE_CU_SHARE")) {
            const int share = atoi(ov);
            if (share > 0) hsflow_set_cu_share(s.ctx, share);
        }
    }
    *out = pl;
    return HSFLOW_OK;
}

int hsflow_pipeline_destroy(hsflow_pipeline *pl)
{
    if (!pl) return HSFLOW_OK;
    if (!pl->lanes.empty()) hipSetDevice(pl->device);
    for (auto &s : pl->slots) {
        hsflow_destroy(s.ctx); // destroy synchronises the slot's stream
        if (s.ev) hipEventDestroy(s.ev);
    }
    for (hipStream_t l : pl->lanes) hipStreamDestroy(l);
    delete pl;
    return HSFLOW_OK;
}

int hsflow_pipeline_submit(hsflow_pipeline *pl, const uint8_t *prev, size_t ps, const uint8_t *curr, size_t cs,
                           float *u, size_t us, float *v, size_t vs, const hsflow_params *params, uint64_t *ticket)
{
    return hsflow_pipeline_submit_ex(pl, HSFLOW_FRAMES_GRAY8, prev, ps, curr, cs, u, us, v, vs, params, ticket);
}

int hsflow_pipeline_submit_ex(hsflow_pipeline *pl, int format, const uint8_t *prev, size_t ps, const uint8_t *curr, size_t cs,
                              float *u, size_t us, float *v, size_t vs, const hsflow_params *params, uint64_t *ticket)
{
    if (!pl) return HSFLOW_E_ARG;
    if (format < HSFLOW_FRAMES_GRAY8 || format > HSFLOW_FRAMES_BGR8_BLUR) return pfail(pl, HSFLOW_E_ARG, "unknown frame format");
    if (!params) return pfail(pl, HSFLOW_E_ARG, "params is null");
    if (!u || !v) return pfail(pl, HSFLOW_E_ARG, "null flow pointer");
    hsflow_pipeline::Slot &s = pl->slots[pl->next % pl->slots.size()];
    int st = finish_slot(pl, s); // the job that used this slot `depth` submissions ago
    if (st) return st;
    switch (format) { // the reference CPU route's pre-processing (gray, 3x3 blur) can ride along on the device
    case HSFLOW_FRAMES_GRAY8: st = hsflow_set_frames_u8_async(s.ctx, 0, prev, ps, curr, cs); break;
    case HSFLOW_FRAMES_GRAY8_BLUR: st = hsflow_set_frames_gray8_blur_async(s.ctx, 0, prev, ps, curr, cs); break;
    default: st = hsflow_set_frames_bgr8_async(s.ctx, 0, prev, ps, curr, cs, format == HSFLOW_FRAMES_BGR8_BLUR); break;
    }
    if (st) {
        hsflow_synchronize(s.ctx); // one of the two uploads may have been queued already
        return ctx_fail(pl, s.ctx, st, "upload of the frames");
    }
    hsflow_params shaped;
    const hsflow_params *use = params; // (a struct of another size is not copied: hsflow_solve_async refuses it)
    if (params->struct_size == sizeof(hsflow_params)) { shaped = stream_shape(pl, *params); use = &shaped; }
    if ((st = hsflow_solve_async(s.ctx, use))) {
        hsflow_synchronize(s.ctx); // the uploads were queued: do not leave them reading caller memory
        return ctx_fail(pl, s.ctx, st, "hsflow_solve_async");
    }
    if ((st = hsflow_get_flow_async(s.ctx, 0, u, us, v, vs))) {
        hsflow_synchronize(s.ctx);
        return ctx_fail(pl, s.ctx, st, "hsflow_get_flow_async");
    }
    if (pl->lanes.size() != pl->slots.size() && hipEventRecord(s.ev, s.stream) != hipSuccess) { // (behind the downloads; see finish_slot)
        hsflow_synchronize(s.ctx);
        return pfail(pl, HSFLOW_E_DEVICE, "hipEventRecord failed");
    }
    s.busy = true;
    s.ticket = pl->next;
    s.u = u; s.v = v; s.us = us; s.vs = vs;
    if (ticket) *ticket = pl->next;
    pl->next++;
    return HSFLOW_OK;
}

int hsflow_pipeline_submit_device(hsflow_pipeline *pl, const void *d_prev, size_t ps, const void *d_curr, size_t cs,
                                  const hsflow_params *params, uint64_t *ticket)
{
    if (!pl) return HSFLOW_E_ARG;
    if (!params) return pfail(pl, HSFLOW_E_ARG, "params is null");
    hsflow_pipeline::Slot &s = pl->slots[pl->next % pl->slots.size()];
    int st = finish_slot(pl, s); // the job that used this slot `depth` submissions ago
    if (st) return st;
    if ((st = hsflow_set_frames_u8_device(s.ctx, 0, d_prev, ps, d_curr, cs))) {
        hsflow_synchronize(s.ctx);
        return ctx_fail(pl, s.ctx, st, "hsflow_set_frames_u8_device");
    }
    hsflow_params shaped;
    const hsflow_params *use = params; // (a struct of another size is not copied: hsflow_solve_async refuses it)
    if (params->struct_size == sizeof(hsflow_params)) { shaped = stream_shape(pl, *params); use = &shaped; }
    if ((st = hsflow_solve_async(s.ctx, use))) {
        hsflow_synchronize(s.ctx); // the frame copies were queued: do not leave them reading caller memory
        return ctx_fail(pl, s.ctx, st, "hsflow_solve_async");
    }
    s.busy = true;
    s.ticket = pl->next;
    s.u = s.v = nullptr; s.us = s.vs = 0; // the flow stays in the slot (hsflow_pipeline_flow_device)
    if (ticket) *ticket = pl->next;
    pl->next++;
    return HSFLOW_OK;
}

int hsflow_pipeline_flow_device(hsflow_pipeline *pl, uint64_t ticket, const float **du, const float **dv, size_t *stride_bytes)
{
    if (!pl) return HSFLOW_E_ARG;
    if (!du || !dv || !stride_bytes) return pfail(pl, HSFLOW_E_ARG, "null out pointer");
    int st = hsflow_pipeline_wait(pl, ticket);
    if (st) return st;
    hsflow_pipeline::Slot &s = pl->slots[ticket % pl->slots.size()];
    if (s.busy || !s.has_done || s.done_ticket != ticket) return pfail(pl, HSFLOW_E_STATE, "the slot of that ticket has been reused by a later pair");
    if ((st = hsflow_flow_view_device(s.ctx, 0, du, dv, stride_bytes))) return ctx_fail(pl, s.ctx, st, "hsflow_flow_view_device");
    return HSFLOW_OK;
}

int hsflow_pipeline_wait(hsflow_pipeline *pl, uint64_t ticket)
{
    if (!pl) return HSFLOW_E_ARG;
    if (ticket >= pl->next) return pfail(pl, HSFLOW_E_ARG, "ticket was never issued");
    hsflow_pipeline::Slot &s = pl->slots[ticket % pl->slots.size()];
    // a later job on the same slot means this one was already waited for inside submit()
    if (!s.busy || s.ticket != ticket) return HSFLOW_OK;
    return finish_slot(pl, s);
}

int hsflow_pipeline_info(hsflow_pipeline *pl, uint64_t ticket, hsflow_info *out)
{
    if (!pl) return HSFLOW_E_ARG;
    if (!out || out->struct_size != sizeof(hsflow_info)) return pfail(pl, HSFLOW_E_ARG, "info null or struct_size mismatch");
    int st = hsflow_pipeline_wait(pl, ticket);
    if (st) return st;
    hsflow_pipeline::Slot &s = pl->slots[ticket % pl->slots.size()];
    if (!s.has_done || s.done_ticket != ticket) return pfail(pl, HSFLOW_E_STATE, "the slot of that ticket has been reused by a later pair");
    if (!s.busy) { // the slot's context still holds that solve: last_eps can be measured now (NaN otherwise)
        hsflow_info info;
        info.struct_size = sizeof(info);
        if ((st = hsflow_get_info_ex(s.ctx, &info, 1))) return ctx_fail(pl, s.ctx, st, "hsflow_get_info_ex");
        s.done.last_eps = info.last_eps;
    }
    *out = s.done;
    return HSFLOW_OK;
}

int hsflow_pipeline_drain(hsflow_pipeline *pl)
{
    if (!pl) return HSFLOW_E_ARG;
    int first = HSFLOW_OK;
    for (auto &s : pl->slots) {
        const int st = finish_slot(pl, s);
        if (st && !first) first = st;
    }
    return first;
}

int hsflow_pipeline_depth(hsflow_pipeline *pl) { return pl ? (int)pl->slots.size() : 0; }

const char *hsflow_pipeline_last_error(hsflow_pipeline *pl) { return pl ? pl->err.c_str() : g_pipeline_create_error.c_str(); }

} // extern "C"
