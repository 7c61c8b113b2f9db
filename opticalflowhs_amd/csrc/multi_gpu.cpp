// multi_gpu.cpp -- the two multi-GPU forms of the hot path behind the C ABI (include/hsflow.h, hsflow_multi_* and
// hsflow_slab_*), for a host program that drives all the GPUs of a node from ONE process, as the reference's own
// caller would (OpticalFlowHS/main.cpp:91-109 constructs one solver object and calls run()).  Built on the public
// single-device entry points only, plus HIP streams / events / peer copies.
//
//   hsflow_multi_*  independent pairs (BASELINE config C4): one pair pipeline per device, each fed by a
//                   host thread of its own; pair i goes to device i mod ndev; no collective of any kind.
//   hsflow_slab_*   one large frame in row slabs (config C5): slab k owns a contiguous range of rows plus `halo` rows
//                   either side; sweeps run in chunks of <= halo, then neighbouring slabs swap `halo` rows of u and v
//                   device to device (hipMemcpyPeerAsync over xGMI between GPUs; an ordinary copy when two slabs share
//                   a device), ordered by events -- the host only enqueues.  Replaces the per-iteration host round
//                   trip of the reference (HSOpticalFlowOpenCL.cpp:483-501, 655-675) at the multi-GPU level.
//                   (The one-process-per-GPU form with RCCL send/recv is opticalflowhs_amd/slab.py.)
#include "../../include/hsflow.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_multi_create_error;

// ------------------------------------------------------------------------------------------------------------
// independent pairs
// ------------------------------------------------------------------------------------------------------------
struct Job {
    uint64_t ticket;
    int format;
    const uint8_t *prev, *curr;
    size_t ps, cs;
    float *u, *v;
    size_t us, vs;
    hsflow_params params;
};

struct Worker {
    int device = 0;
    hsflow_pipeline *pl = nullptr;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<Job> queue;
    struct Flight { uint64_t ticket, pipeline_ticket; bool submitted; };
    std::deque<Flight> inflight; // oldest first; submitted = false: the submit failed, nothing to wait for
    uint64_t done_upto = 0;   // every ticket of this worker below this has finished (tickets of a worker grow)
    uint64_t submitted = 0;   // jobs handed to this worker so far
    uint64_t finished = 0;    // jobs this worker has completed
    bool stop = false;
    int status = HSFLOW_OK;   // first error
    std::string err;
};

} // namespace

struct hsflow_multi {
    std::vector<Worker *> workers;
    uint64_t next = 0;
    std::string err;
};

namespace {

int mfail(hsflow_multi *m, int code, const std::string &msg)
{
    if (m) m->err = msg; else g_multi_create_error = msg;
    return code;
}

void worker_fail(Worker *w, int st, const char *what)
{
    if (w->status == HSFLOW_OK) {
        w->status = st;
        w->err = std::string(what) + ": " + hsflow_pipeline_last_error(w->pl);
    }
}

// One host thread per device: takes jobs in order, keeps up to `depth` of them in flight in the device's pipeline
// (upload, solve and download of neighbouring pairs overlap there) and retires the oldest when it runs dry or full.
void worker_main(Worker *w)
{
    if (hipSetDevice(w->device) != hipSuccess) {
        std::lock_guard<std::mutex> lk(w->mu);
        w->status = HSFLOW_E_DEVICE;
        w->err = "hipSetDevice failed in the worker thread of device " + std::to_string(w->device);
    }
    const int depth = hsflow_pipeline_depth(w->pl);
    for (;;) {
        Job job;
        bool have = false;
        {
            std::unique_lock<std::mutex> lk(w->mu);
            if (w->queue.empty() && w->inflight.empty()) w->cv_work.wait(lk, [&] { return w->stop || !w->queue.empty(); });
            if (w->stop && w->queue.empty() && w->inflight.empty()) return;
            if (!w->queue.empty() && (int)w->inflight.size() < depth) {
                job = w->queue.front();
                w->queue.pop_front();
                have = true;
            }
        }
        if (have) {
            uint64_t pt = 0;
            bool dead;
            { std::lock_guard<std::mutex> lk(w->mu); dead = w->status == HSFLOW_E_DEVICE && w->err.rfind("hipSetDevice", 0) == 0; }
            const int st = dead ? HSFLOW_E_DEVICE : hsflow_pipeline_submit_ex(w->pl, job.format, job.prev, job.ps, job.curr, job.cs, job.u, job.us, job.v, job.vs,
                                                     &job.params, &pt);
            std::lock_guard<std::mutex> lk(w->mu);
            if (st) worker_fail(w, st, "hsflow_pipeline_submit_ex");
            w->inflight.push_back({job.ticket, pt, st == HSFLOW_OK}); // (retired in order either way)
            continue;
        }
        // nothing to submit right now (pipeline full, or queue empty with pairs still in flight): retire the oldest
        Worker::Flight old;
        {
            std::lock_guard<std::mutex> lk(w->mu);
            if (w->inflight.empty()) continue;
            old = w->inflight.front();
        }
        const int st = old.submitted ? hsflow_pipeline_wait(w->pl, old.pipeline_ticket) : HSFLOW_OK;
        std::lock_guard<std::mutex> lk(w->mu);
        if (st) worker_fail(w, st, "hsflow_pipeline_wait");
        w->inflight.pop_front();
        w->finished++;
        w->done_upto = old.ticket + 1;
        w->cv_done.notify_all();
    }
}

} // namespace

extern "C" {

int hsflow_multi_create(hsflow_multi **out, const int *devices, int ndev, int width, int height, int depth)
{
    if (!out) return mfail(nullptr, HSFLOW_E_ARG, "out is null");
    *out = nullptr;
    if (!devices || ndev < 1 || ndev > 64) return mfail(nullptr, HSFLOW_E_ARG, "devices null or ndev out of range (1..64)");
    hsflow_multi *m = new (std::nothrow) hsflow_multi();
    if (!m) return mfail(nullptr, HSFLOW_E_OOM, "host allocation failed");
    for (int i = 0; i < ndev; i++) {
        Worker *w = new (std::nothrow) Worker();
        if (!w) { hsflow_multi_destroy(m); return mfail(nullptr, HSFLOW_E_OOM, "host allocation failed"); }
        w->device = devices[i];
        m->workers.push_back(w);
        const int st = hsflow_pipeline_create(&w->pl, devices[i], width, height, depth);
        if (st) {
            g_multi_create_error = std::string("hsflow_pipeline_create (device ") + std::to_string(devices[i]) + "): " + hsflow_pipeline_last_error(nullptr);
            hsflow_multi_destroy(m);
            return st;
        }
    }
    for (Worker *w : m->workers) w->th = std::thread(worker_main, w);
    *out = m;
    return HSFLOW_OK;
}

int hsflow_multi_destroy(hsflow_multi *m)
{
    if (!m) return HSFLOW_OK;
    for (Worker *w : m->workers) {
        if (w->th.joinable()) {
            { std::lock_guard<std::mutex> lk(w->mu); w->stop = true; }
            w->cv_work.notify_all();
            w->th.join(); // drains what was submitted first
        }
        hsflow_pipeline_destroy(w->pl);
        delete w;
    }
    delete m;
    return HSFLOW_OK;
}

int hsflow_multi_devices(hsflow_multi *m) { return m ? (int)m->workers.size() : 0; }

int hsflow_multi_submit(hsflow_multi *m, int format, const uint8_t *prev, size_t ps, const uint8_t *curr, size_t cs,
                        float *u, size_t us, float *v, size_t vs, const hsflow_params *params, uint64_t *ticket)
{
    if (!m) return HSFLOW_E_ARG;
    if (!params || params->struct_size != sizeof(hsflow_params)) return mfail(m, HSFLOW_E_ARG, "params null or struct_size mismatch");
    if (!prev || !curr || !u || !v) return mfail(m, HSFLOW_E_ARG, "null frame or flow pointer");
    Worker *w = m->workers[m->next % m->workers.size()]; // pair i -> device i mod ndev
    Job j{m->next, format, prev, curr, ps, cs, u, v, us, vs, *params};
    {
        std::lock_guard<std::mutex> lk(w->mu);
        if (w->status) return mfail(m, w->status, w->err);
        w->queue.push_back(j);
        w->submitted++;
    }
    w->cv_work.notify_one();
    if (ticket) *ticket = m->next;
    m->next++;
    return HSFLOW_OK;
}

int hsflow_multi_wait(hsflow_multi *m, uint64_t ticket)
{
    if (!m) return HSFLOW_E_ARG;
    if (ticket >= m->next) return mfail(m, HSFLOW_E_ARG, "ticket was never issued");
    Worker *w = m->workers[ticket % m->workers.size()];
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv_done.wait(lk, [&] { return w->done_upto > ticket; });
    if (w->status) return mfail(m, w->status, w->err);
    return HSFLOW_OK;
}

int hsflow_multi_drain(hsflow_multi *m)
{
    if (!m) return HSFLOW_E_ARG;
    int first = HSFLOW_OK;
    for (Worker *w : m->workers) {
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv_done.wait(lk, [&] { return w->finished == w->submitted; });
        if (w->status && !first) first = mfail(m, w->status, w->err);
    }
    return first;
}

const char *hsflow_multi_last_error(hsflow_multi *m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }

} // extern "C"

// ------------------------------------------------------------------------------------------------------------
// one frame in row slabs
// ------------------------------------------------------------------------------------------------------------
namespace {

struct Slab {
    int device = 0;
    int lo = 0, hi = 0, top = 0, bot = 0; // owned rows [lo, hi), halo rows actually held above / below
    int row0 = 0, local_h = 0;            // first frame row held, rows held
    hsflow_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    // staging rows for the exchange, on this slab's device: what it sends up / down and what it receives from above / below
    float *send_up = nullptr, *send_dn = nullptr, *recv_up = nullptr, *recv_dn = nullptr; // each: 2 planes x halo rows x W
    hipEvent_t sent_up = nullptr, sent_dn = nullptr; // this slab's rows for the neighbour above / below are in its send buffer
    // the neighbour above / below has copied them out (buffer free again).  Recorded on the NEIGHBOUR's stream, so they
    // are created on the neighbour's device (an event can only be recorded on a stream of the device it was created on)
    hipEvent_t took_up = nullptr, took_dn = nullptr;
    bool took_up_valid = false, took_dn_valid = false;
    // ITER|EPS: copies of the slab's flow (all local rows, 2 planes): at the start of the solve (use_previous) and at the
    // start of the chunk being measured -- allocated when first needed
    float *start_uv = nullptr, *chunk_uv = nullptr;
};

} // namespace

struct hsflow_slab {
    int W = 0, H = 0, halo = 0;
    std::vector<Slab> slabs;
    bool frames_set = false;
    bool solved = false;      // a solve has run on these frames: the halos hold what its last chunk left there (stale)
    int exchanges = 0;
    int iterations_done = 0;  // sweeps of the last solve (fewer than the budget when the early stop fired)
    int eps_measured = 0;     // 1 if the last ITER|EPS solve had to measure Eps sweep by sweep (no slab's witness held)
    std::string err;
};

namespace {

int sfail(hsflow_slab *s, int code, const std::string &msg)
{
    if (s) s->err = msg; else g_multi_create_error = msg;
    return code;
}

#define SL_HIP(s, call)                                                                           \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return sfail((s), e_ == hipErrorOutOfMemory ? HSFLOW_E_OOM : HSFLOW_E_DEVICE,         \
                         std::string(#call) + ": " + hipGetErrorString(e_));                      \
    } while (0)
#define SL_CTX(s, sl, call)                                                                       \
    do {                                                                                          \
        const int st_ = (call);                                                                   \
        if (st_) return sfail((s), st_, std::string(#call) + ": " + hsflow_last_error((sl).ctx)); \
    } while (0)

// `halo` rows of u, v starting at local row src_row of slab S go into local rows [dst_row, dst_row + halo) of slab D.
// send / recv: S's and D's staging buffers for this direction; sent / took: the two events that order the buffers.
int move_rows(hsflow_slab *s, Slab &S, int src_row, float *send, hipEvent_t sent, hipEvent_t took, bool &took_valid,
              Slab &D, int dst_row, float *recv)
{
    const size_t rowb = (size_t)s->W * sizeof(float), plane = (size_t)s->halo * s->W;
    SL_HIP(s, hipSetDevice(S.device));
    if (took_valid) SL_HIP(s, hipStreamWaitEvent(S.stream, took, 0)); // the previous round's copy out of `send` is done
    SL_CTX(s, S, hsflow_get_flow_device(S.ctx, 0, src_row, s->halo, send, rowb, send + plane, rowb));
    SL_HIP(s, hipEventRecord(sent, S.stream));
    SL_HIP(s, hipSetDevice(D.device));
    SL_HIP(s, hipStreamWaitEvent(D.stream, sent, 0));
    if (S.device != D.device) SL_HIP(s, hipMemcpyPeerAsync(recv, D.device, send, S.device, 2 * plane * sizeof(float), D.stream));
    else SL_HIP(s, hipMemcpyAsync(recv, send, 2 * plane * sizeof(float), hipMemcpyDeviceToDevice, D.stream));
    SL_HIP(s, hipEventRecord(took, D.stream));
    took_valid = true;
    SL_CTX(s, D, hsflow_set_flow_device(D.ctx, 0, dst_row, s->halo, recv, rowb, recv + plane, rowb));
    return HSFLOW_OK;
}

} // namespace

extern "C" {

int hsflow_slab_destroy(hsflow_slab *s)
{
    if (!s) return HSFLOW_OK;
    for (Slab &sl : s->slabs) {
        hipSetDevice(sl.device);
        if (sl.stream) hipStreamSynchronize(sl.stream);
    }
    for (Slab &sl : s->slabs) {
        hipSetDevice(sl.device);
        hsflow_destroy(sl.ctx);
        hipFree(sl.send_up); hipFree(sl.send_dn); hipFree(sl.recv_up); hipFree(sl.recv_dn);
        hipFree(sl.start_uv); hipFree(sl.chunk_uv);
        for (hipEvent_t e : {sl.sent_up, sl.sent_dn, sl.took_up, sl.took_dn})
            if (e) hipEventDestroy(e);
        if (sl.stream) hipStreamDestroy(sl.stream);
    }
    delete s;
    return HSFLOW_OK;
}

int hsflow_slab_create(hsflow_slab **out, const int *devices, int nslab, int width, int height, int halo)
{
    if (!out) return sfail(nullptr, HSFLOW_E_ARG, "out is null");
    *out = nullptr;
    if (!devices || nslab < 1 || nslab > 64) return sfail(nullptr, HSFLOW_E_ARG, "devices null or slab count out of range (1..64)");
    if (width <= 0 || height <= 0) return sfail(nullptr, HSFLOW_E_SIZE, "width and height must be positive");
    if (halo < 1) return sfail(nullptr, HSFLOW_E_ARG, "halo must be >= 1");
    if (nslab > 1 && height / nslab < halo) return sfail(nullptr, HSFLOW_E_SIZE, "slabs would be thinner than the halo");
    hsflow_slab *s = new (std::nothrow) hsflow_slab();
    if (!s) return sfail(nullptr, HSFLOW_E_OOM, "host allocation failed");
    s->W = width; s->H = height; s->halo = halo;
    s->slabs.resize((size_t)nslab);
    const int base = height / nslab, extra = height % nslab;
    auto bail = [&](int code) { const std::string m = s->err; hsflow_slab_destroy(s); g_multi_create_error = m; return code; };
    for (int k = 0; k < nslab; k++) {
        Slab &sl = s->slabs[(size_t)k];
        sl.device = devices[k];
        sl.lo = k * base + std::min(k, extra);                 // contiguous rows, sizes differ by at most one
        sl.hi = sl.lo + base + (k < extra ? 1 : 0);
        sl.top = std::min(halo, sl.lo);
        sl.bot = std::min(halo, height - sl.hi);
        sl.row0 = sl.lo - sl.top;
        sl.local_h = (sl.hi + sl.bot) - sl.row0;
#define SL_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { s->err = std::string(#call) + ": " + hipGetErrorString(e_); return bail(e_ == hipErrorOutOfMemory ? HSFLOW_E_OOM : HSFLOW_E_DEVICE); } } while (0)
        SL_TRY(hipSetDevice(sl.device));
        SL_TRY(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        int st = hsflow_create(&sl.ctx, sl.device, width, sl.local_h, 1, sl.stream, 0);
        if (st) { s->err = std::string("hsflow_create: ") + hsflow_last_error(nullptr); return bail(st); }
        if ((st = hsflow_set_row_origin(sl.ctx, sl.row0))) { s->err = std::string("hsflow_set_row_origin: ") + hsflow_last_error(sl.ctx); return bail(st); }
        // Eps of the frame = the maximum over the slabs of the Eps of their OWNED rows (the halo rows repeat a neighbour's)
        if ((st = hsflow_set_eps_rows(sl.ctx, sl.top, sl.hi - sl.lo))) { s->err = std::string("hsflow_set_eps_rows: ") + hsflow_last_error(sl.ctx); return bail(st); }
        const size_t bytes = (size_t)2 * halo * width * sizeof(float);
        if (k > 0) { SL_TRY(hipMalloc((void **)&sl.send_up, bytes)); SL_TRY(hipMalloc((void **)&sl.recv_up, bytes)); }
        if (k + 1 < nslab) { SL_TRY(hipMalloc((void **)&sl.send_dn, bytes)); SL_TRY(hipMalloc((void **)&sl.recv_dn, bytes)); }
        for (hipEvent_t *e : {&sl.sent_up, &sl.sent_dn}) SL_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    for (int k = 0; k < nslab; k++) { // "taken" events live on the device of the slab that does the taking
        Slab &sl = s->slabs[(size_t)k];
        if (k > 0) {
            SL_TRY(hipSetDevice(s->slabs[(size_t)k - 1].device));
            SL_TRY(hipEventCreateWithFlags(&sl.took_up, hipEventDisableTiming));
        }
        if (k + 1 < nslab) {
            SL_TRY(hipSetDevice(s->slabs[(size_t)k + 1].device));
            SL_TRY(hipEventCreateWithFlags(&sl.took_dn, hipEventDisableTiming));
        }
    }
    // peer access between neighbouring slabs on different GPUs (xGMI); without it the runtime stages the copy
    for (int k = 0; k + 1 < nslab; k++) {
        const int a = s->slabs[(size_t)k].device, b = s->slabs[(size_t)k + 1].device;
        if (a == b) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) { hipSetDevice(a); (void)hipDeviceEnablePeerAccess(b, 0); }
        if (hipDeviceCanAccessPeer(&can, b, a) == hipSuccess && can) { hipSetDevice(b); (void)hipDeviceEnablePeerAccess(a, 0); }
        (void)hipGetLastError(); // "already enabled" is fine
    }
#undef SL_TRY
    *out = s;
    return HSFLOW_OK;
}

int hsflow_slab_create_overlapped(hsflow_slab **out, const int *devices, int ndev, int width, int height, int halo)
{
    if (!out) return sfail(nullptr, HSFLOW_E_ARG, "out is null");
    *out = nullptr;
    if (!devices || ndev < 1 || ndev > 32) return sfail(nullptr, HSFLOW_E_ARG, "devices null or device count out of range (1..32)");
    std::vector<int> twice;
    for (int k = 0; k < ndev; k++) { twice.push_back(devices[k]); twice.push_back(devices[k]); }
    return hsflow_slab_create(out, twice.data(), 2 * ndev, width, height, halo);
}

int hsflow_slab_count(hsflow_slab *s) { return s ? (int)s->slabs.size() : 0; }

int hsflow_slab_rows(hsflow_slab *s, int k, int *lo, int *hi)
{
    if (!s || k < 0 || k >= (int)s->slabs.size()) return HSFLOW_E_ARG;
    if (lo) *lo = s->slabs[(size_t)k].lo;
    if (hi) *hi = s->slabs[(size_t)k].hi;
    return HSFLOW_OK;
}

int hsflow_slab_set_frames_u8(hsflow_slab *s, const uint8_t *prev, size_t ps, const uint8_t *curr, size_t cs)
{
    if (!s) return HSFLOW_E_ARG;
    if (!prev || !curr) return sfail(s, HSFLOW_E_ARG, "null frame pointer");
    if (ps < (size_t)s->W || cs < (size_t)s->W) return sfail(s, HSFLOW_E_SIZE, "frame stride smaller than the width");
    for (Slab &sl : s->slabs) { // every slab uploads its own rows plus the halo rows (the derivative stencil needs rows y-1 .. y+1)
        SL_HIP(s, hipSetDevice(sl.device));
        SL_CTX(s, sl, hsflow_set_frames_u8(sl.ctx, 0, prev + (size_t)sl.row0 * ps, ps, curr + (size_t)sl.row0 * cs, cs));
    }
    s->frames_set = true;
    s->solved = false;
    return HSFLOW_OK;
}

} // extern "C"

namespace {

// Neighbouring slabs swap `halo` rows of u, v (enqueue only; ordered by events).
int exchange_all(hsflow_slab *s)
{
    const int n = (int)s->slabs.size();
    for (int k = 0; k + 1 < n; k++) {
        Slab &A = s->slabs[(size_t)k], &B = s->slabs[(size_t)k + 1];
        // A's last owned rows -> B's top halo [0, halo);  B's first owned rows -> A's bottom halo
        int st = move_rows(s, A, A.top + (A.hi - A.lo) - s->halo, A.send_dn, A.sent_dn, A.took_dn, A.took_dn_valid, B, 0, B.recv_up);
        if (st) return st;
        st = move_rows(s, B, B.top, B.send_up, B.sent_up, B.took_up, B.took_up_valid, A, A.top + (A.hi - A.lo), A.recv_dn);
        if (st) return st;
    }
    s->exchanges++;
    return HSFLOW_OK;
}

// `sweeps` sweeps on every slab, enqueued (ITER) -- term_type ITER|EPS: witness launches, the checks stay owed.
int chunk_async(hsflow_slab *s, const hsflow_params &pp, int sweeps, bool from_zero, bool with_deriv, int term_type)
{
    hsflow_params q = pp;
    q.max_iter = sweeps;
    q.term_type = term_type;
    q.use_previous = from_zero ? 0 : 1;
    q.reuse_derivatives = with_deriv ? 0 : 1;
    q.profile = 0;
    for (Slab &sl : s->slabs) {
        SL_HIP(s, hipSetDevice(sl.device));
        SL_CTX(s, sl, hsflow_solve_async(sl.ctx, &q));
    }
    return HSFLOW_OK;
}

// The slabs' flow (all local rows, halos included) to / from a backup of its own.
int copy_flows(hsflow_slab *s, float *Slab::*buf, bool save)
{
    const size_t rowb = (size_t)s->W * sizeof(float);
    for (Slab &sl : s->slabs) {
        SL_HIP(s, hipSetDevice(sl.device));
        const size_t plane = (size_t)sl.local_h * s->W;
        if (!(sl.*buf)) SL_HIP(s, hipMalloc((void **)&(sl.*buf), 2 * plane * sizeof(float)));
        float *b = sl.*buf;
        if (save) SL_CTX(s, sl, hsflow_get_flow_device(sl.ctx, 0, 0, sl.local_h, b, rowb, b + plane, rowb));
        else SL_CTX(s, sl, hsflow_set_flow_device(sl.ctx, 0, 0, sl.local_h, b, rowb, b + plane, rowb));
    }
    return HSFLOW_OK;
}

// `done` sweeps from the start of the solve, in the chunks the solve itself used (the replay of a deterministic run).
int replay(hsflow_slab *s, const hsflow_params &pp, int done)
{
    if (pp.use_previous) { const int st = copy_flows(s, &Slab::start_uv, false); if (st) return st; }
    int at = 0;
    while (at < done) {
        const int chunk = std::min(s->halo, done - at);
        int st = chunk_async(s, pp, chunk, at == 0 && !pp.use_previous, false, HSFLOW_TERM_ITER);
        if (st) return st;
        at += chunk;
        if ((st = exchange_all(s))) return st; // (every replayed chunk was followed by one: the solve went on after it)
    }
    return HSFLOW_OK;
}

} // namespace

extern "C" {

// Termination as the reference calls the solver (OpticalFlowOpenCV.cpp:29,94: ITER|EPS): the frame's Eps of a sweep is
// the maximum over the slabs of the Eps of their owned rows, so
//   * a chunk in which ANY slab's witness proves "my Eps stayed >= epsilon" cannot contain the stop (fast path: witness
//     launches on every slab, one look at the verdicts per chunk);
//   * when no slab can vouch for a chunk, the solve is replayed to that chunk's start (it is deterministic) and from
//     there on every chunk is measured: each slab reports the Eps of every sweep (hsflow_solve_probe), the host takes
//     the maximum over the slabs and finds the first sweep below epsilon; the chunk is then repeated up to that sweep
//     from a copy of its starting flow -- the stopping sweep of the one-context solve, bit for bit
//     (cv210.dll@0x1012f10b-0x1012f14a).
int hsflow_slab_solve(hsflow_slab *s, const hsflow_params *pp)
{
    if (!s) return HSFLOW_E_ARG;
    if (!pp || pp->struct_size != sizeof(hsflow_params)) return sfail(s, HSFLOW_E_ARG, "params null or struct_size mismatch");
    if (!s->frames_set) return sfail(s, HSFLOW_E_STATE, "frames were not set");
    if (pp->mode != HSFLOW_MODE_CV) return sfail(s, HSFLOW_E_ARG, "row slabs run the CV discretisation");
    const bool use_eps = (pp->term_type & HSFLOW_TERM_EPS) != 0;
    if (!(pp->term_type & HSFLOW_TERM_ITER) || pp->max_iter <= 0)
        return sfail(s, HSFLOW_E_ARG, "row slabs need a sweep budget: ITER or ITER|EPS with max_iter > 0");
    const int n = (int)s->slabs.size(), budget = pp->max_iter;
    s->exchanges = 0;
    s->eps_measured = 0;
    s->iterations_done = 0;
    int st = HSFLOW_OK;
    if (n == 1) { // one slab is the whole frame: the context's own solve, its own stop rule
        Slab &sl = s->slabs[0];
        SL_HIP(s, hipSetDevice(sl.device));
        SL_CTX(s, sl, hsflow_solve(sl.ctx, pp));
        hsflow_info info;
        info.struct_size = sizeof(info);
        SL_CTX(s, sl, hsflow_get_info_ex(sl.ctx, &info, 0));
        s->iterations_done = info.iterations_done;
        s->eps_measured = info.eps_rerun;
        s->solved = true;
        return HSFLOW_OK;
    }
    if (pp->use_previous && s->solved && (st = exchange_all(s))) return st; // the halos the last solve left are stale
    if (pp->use_previous && use_eps && (st = copy_flows(s, &Slab::start_uv, true))) return st;
    int done = 0;
    bool measure = false; // every chunk from here on is measured sweep by sweep
    std::vector<float> eps_s, eps_max;
    while (done < budget) {
        const int chunk = std::min(s->halo, budget - done);
        const bool first = done == 0;
        const bool from_zero = first && !pp->use_previous;
        if (!use_eps) {
            if ((st = chunk_async(s, *pp, chunk, from_zero, first, HSFLOW_TERM_ITER))) return st;
        } else if (!measure) {
            bool vouched = false;
            st = chunk_async(s, *pp, chunk, from_zero, first, HSFLOW_TERM_ITER | HSFLOW_TERM_EPS);
            if (st == HSFLOW_E_ARG) {
                // a slab whose launch plan cannot run witness launches (a core tile thinner than a strip: very thin slabs):
                // nobody vouches, the solve is measured from here on (the replay below also undoes what the other slabs
                // may already have enqueued for this chunk)
                for (Slab &sl : s->slabs) { // (their owed checks, if any, are dropped with the flow they belong to)
                    SL_HIP(s, hipSetDevice(sl.device));
                    int proven = 0;
                    (void)hsflow_take_verdict(sl.ctx, &proven);
                }
            } else if (st) {
                return st;
            } else {
                for (Slab &sl : s->slabs) {
                    SL_HIP(s, hipSetDevice(sl.device));
                    int proven = 0;
                    SL_CTX(s, sl, hsflow_take_verdict(sl.ctx, &proven));
                    vouched = vouched || proven != 0;
                }
            }
            if (!vouched) { // back to this chunk's start, then measure
                measure = true;
                s->eps_measured = 1;
                if ((st = replay(s, *pp, done))) return st;
                continue;
            }
        } else {
            if ((st = copy_flows(s, &Slab::chunk_uv, true))) return st;
            hsflow_params q = *pp;
            q.max_iter = chunk;
            q.use_previous = from_zero ? 0 : 1;
            q.reuse_derivatives = first ? 0 : 1;
            q.use_graph = 0;
            eps_s.assign((size_t)chunk, 0.f);
            eps_max.assign((size_t)chunk, 0.f);
            for (Slab &sl : s->slabs) {
                SL_HIP(s, hipSetDevice(sl.device));
                SL_CTX(s, sl, hsflow_solve_probe(sl.ctx, &q, eps_s.data()));
                for (int k = 0; k < chunk; k++) eps_max[(size_t)k] = std::max(eps_max[(size_t)k], eps_s[(size_t)k]);
            }
            int hit = -1;
            for (int k = 0; k < chunk && hit < 0; k++)
                if ((double)eps_max[(size_t)k] < pp->epsilon) hit = k;
            if (hit >= 0) { // the solve ends after sweep hit + 1 of this chunk
                if (hit + 1 < chunk) {
                    if ((st = copy_flows(s, &Slab::chunk_uv, false))) return st;
                    if ((st = chunk_async(s, *pp, hit + 1, from_zero, false, HSFLOW_TERM_ITER))) return st;
                }
                done += hit + 1;
                break;
            }
        }
        done += chunk;
        if (done < budget && (st = exchange_all(s))) return st; // stale rows: closer than `chunk` to a slab's artificial edge
    }
    for (Slab &sl : s->slabs) {
        SL_HIP(s, hipSetDevice(sl.device));
        SL_CTX(s, sl, hsflow_synchronize(sl.ctx));
    }
    s->iterations_done = done;
    s->solved = true;
    return HSFLOW_OK;
}

int hsflow_slab_iterations_done(hsflow_slab *s) { return s ? s->iterations_done : 0; }
int hsflow_slab_eps_measured(hsflow_slab *s) { return s ? s->eps_measured : 0; }

int hsflow_slab_exchanges(hsflow_slab *s) { return s ? s->exchanges : 0; }

int hsflow_slab_get_flow(hsflow_slab *s, float *u, size_t us, float *v, size_t vs)
{
    if (!s) return HSFLOW_E_ARG;
    if (!u || !v) return sfail(s, HSFLOW_E_ARG, "null flow pointer");
    const size_t rowb = (size_t)s->W * sizeof(float);
    if ((us & 3) || (vs & 3) || us < rowb || vs < rowb) return sfail(s, HSFLOW_E_SIZE, "flow stride must be a multiple of 4 and >= 4*width");
    std::vector<float> tu, tv;
    for (Slab &sl : s->slabs) { // each slab's context holds its halo rows too: fetch the slab, keep the owned rows
        SL_HIP(s, hipSetDevice(sl.device));
        tu.resize((size_t)sl.local_h * s->W);
        tv.resize((size_t)sl.local_h * s->W);
        SL_CTX(s, sl, hsflow_get_flow(sl.ctx, 0, tu.data(), rowb, tv.data(), rowb));
        for (int y = sl.lo; y < sl.hi; y++) {
            std::copy_n(tu.data() + (size_t)(y - sl.row0) * s->W, s->W, (float *)((char *)u + (size_t)y * us));
            std::copy_n(tv.data() + (size_t)(y - sl.row0) * s->W, s->W, (float *)((char *)v + (size_t)y * vs));
        }
    }
    return HSFLOW_OK;
}

const char *hsflow_slab_last_error(hsflow_slab *s) { return s ? s->err.c_str() : g_multi_create_error.c_str(); }

} // extern "C"
