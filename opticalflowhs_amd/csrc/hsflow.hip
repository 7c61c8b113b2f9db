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
#include "hs_kernels_classic_strip.hip.h"

#include <atomic>
#include <chrono>
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "hs_context.hip.h"
#include "hs_plan_launch.hip.h"
#include "hs_runtime.hip.h"
#include "hs_solve.hip.h"


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
    HS_TRY(hipMalloc((void **)&c->dZero, ((size_t)c->P + 64) * sizeof(float)));
    {
        int ncu = 0;
        HS_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
        c->num_cu = ncu;
    }
    if (getenv("HSFLOW_DEBUG_STAMPS")) HS_TRY(hipMalloc((void **)&c->dStamps, (size_t)kStampTiles * 8 * sizeof(unsigned long long)));
    // deterministic contents for padding columns and the initial flow
    HS_TRY(hipMemsetAsync(c->dZero, 0, ((size_t)c->P + 64) * sizeof(float), c->stream));
    HS_TRY(hipMemsetAsync(c->dA, 0, px, c->stream));
    HS_TRY(hipMemsetAsync(c->dB, 0, px, c->stream));
    HS_TRY(hipMemsetAsync(c->dCoef, 0, px * sizeof(uint32_t), c->stream));
    for (int i = 0; i < 2; i++) {
        HS_TRY(hipMemsetAsync(c->dU[i], 0, px * sizeof(float), c->stream));
        HS_TRY(hipMemsetAsync(c->dV[i], 0, px * sizeof(float), c->stream));
    }
    HS_TRY(hipStreamSynchronize(c->stream));
#undef HS_TRY
    g_live_ctx[device & 63]++;
    c->counted = true;
    *out = c;
    return HSFLOW_OK;
}

int hsflow_set_row_origin(hsflow_ctx *c, int first_row)
{
    int st = check_ctx(c, 0);
    if (st) return st;
    if ((st = settle_pending(c))) return st;
    const int org = first_row & 1;
    if (org != c->org) {
        drop_graphs(c); // captured launches carry the phase (kernel choice and geometry)
        c->plan_cache.clear();
        c->lastl.valid = false;
        c->org = org;
    }
    return HSFLOW_OK;
}

int hsflow_set_cu_share(hsflow_ctx *c, int compute_units)
{
    int st = check_ctx(c, 0);
    if (st) return st;
    if (compute_units < 0) return fail(c, HSFLOW_E_ARG, "compute_units must be >= 0 (0: the whole chip)");
    const int share = compute_units >= kNumCU ? 0 : compute_units;
    if (share != c->cu_share) {
        if ((st = settle_pending(c))) return st;
        c->cu_share = share; // (graphs are keyed by the launch shape, so cached ones stay valid)
        c->plan_cache.clear();
    }
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
    hipFree(c->dEpsTiles); hipFree(c->dUb); hipFree(c->dVb);
    hipFree(c->dUp); hipFree(c->dVp); hipFree(c->dFlags);
    hipFree(c->dSeq);
    hipFree(c->dZero);
    if (c->hMark) hipHostFree(c->hMark);
    if (c->hErr) hipHostFree(c->hErr);
    if (c->counted) g_live_ctx[c->device & 63]--;
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
    { // both frames in one launch (two 2-D copies cost two launches and their gaps: 5 % of a 1080p / 100 solve)
        const dim3 grid((c->W + 1023) / 1024, (c->H + 3) / 4, 2), block(64, 4);
        const bool aligned = (((uintptr_t)dprev | (uintptr_t)dcurr | ps | cs) & 15u) == 0;
        uint8_t *dA = c->dA + pair * c->plane, *dB = c->dB + pair * c->plane;
        if (aligned)
            hipLaunchKernelGGL(hsk::k_copy_pair_u8<true>, grid, block, 0, c->stream, (const uint8_t *)dprev, (long long)ps, (const uint8_t *)dcurr,
                               (long long)cs, dA, dB, c->W, c->H, c->P);
        else
            hipLaunchKernelGGL(hsk::k_copy_pair_u8<false>, grid, block, 0, c->stream, (const uint8_t *)dprev, (long long)ps, (const uint8_t *)dcurr,
                               (long long)cs, dA, dB, c->W, c->H, c->P);
        HS_HIP(c, hipGetLastError());
    }
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

// Upload one host frame (colour or gray) into scratch, convert / blur on the device into dst.  `slot`
// (0 / 1) picks one of two scratch areas, so that the two frames of a pair can be in flight together;
// sync = false only enqueues (host buffer must stay valid until the stream has passed the copy).
static int preprocess_frame(hsflow_ctx *c, uint8_t *dst, const uint8_t *host, size_t stride, bool colour, bool blur, int slot = 0,
                            bool sync = true)
{
    const size_t bgr_bytes = (size_t)c->W * 3 * c->H, gray_bytes = (size_t)c->plane;
    const size_t one = bgr_bytes + 2 * gray_bytes, need = 2 * one;
    if (c->scratch_bytes < need) {
        HS_HIP(c, hipStreamSynchronize(c->stream)); // nothing in flight may still use the old area
        hipFree(c->dScratch);
        c->dScratch = nullptr; c->scratch_bytes = 0;
        HS_HIP(c, hipMalloc(&c->dScratch, need));
        c->scratch_bytes = need;
    }
    uint8_t *dBgr = (uint8_t *)c->dScratch + (size_t)slot * one, *dGray = dBgr + bgr_bytes;
    const dim3 grid((c->W + 255) / 256, (c->H + 3) / 4), block(64, 4);
    if (sync) HS_HIP(c, hipStreamSynchronize(c->stream));
    if (colour) {
        if (sync) HS_HIP(c, hipMemcpy2D(dBgr, (size_t)c->W * 3, host, stride, (size_t)c->W * 3, c->H, hipMemcpyHostToDevice));
        else HS_HIP(c, copy_rows_async(c, dBgr, (size_t)c->W * 3, host, stride, (size_t)c->W * 3, c->H, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(hsk::k_bgr2gray, grid, block, 0, c->stream, dBgr, (long long)c->W * 3, blur ? dGray : dst, c->W, c->H, c->P);
    } else {
        if (sync) HS_HIP(c, hipMemcpy2D(blur ? dGray : dst, c->P, host, stride, c->W, c->H, hipMemcpyHostToDevice));
        else HS_HIP(c, copy_rows_async(c, blur ? dGray : dst, c->P, host, stride, c->W, c->H, hipMemcpyHostToDevice));
    }
    if (blur) hipLaunchKernelGGL(hsk::k_box_blur3, grid, block, 0, c->stream, dGray, dst, c->W, c->H, c->P);
    HS_HIP(c, hipGetLastError());
    if (sync) HS_HIP(c, hipStreamSynchronize(c->stream));
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

// Asynchronous forms: everything is only enqueued on ctx's stream (uploads into two scratch areas, then
// the conversion / blur kernels); the host buffers belong to the context until hsflow_synchronize.
int hsflow_set_frames_bgr8_async(hsflow_ctx *c, int pair, const uint8_t *prev, size_t ps, const uint8_t *curr, size_t cs, int blur3x3)
{
    int st = check_ctx(c, pair);
    if (st) return st;
    if ((st = settle_pending(c))) return st; // an unverified asynchronous solve still needs the old inputs
    if (!prev || !curr) return fail(c, HSFLOW_E_ARG, "null frame pointer");
    if (ps < (size_t)c->W * 3 || cs < (size_t)c->W * 3) return fail(c, HSFLOW_E_SIZE, "colour frame stride smaller than 3*width");
    if ((st = preprocess_frame(c, c->dA + pair * c->plane, prev, ps, true, blur3x3 != 0, 0, false))) return st;
    if ((st = preprocess_frame(c, c->dB + pair * c->plane, curr, cs, true, blur3x3 != 0, 1, false))) return st;
    c->frames_set = true;
    c->coef_valid = false;
    return HSFLOW_OK;
}

int hsflow_set_frames_gray8_blur_async(hsflow_ctx *c, int pair, const uint8_t *prev, size_t ps, const uint8_t *curr, size_t cs)
{
    int st = check_ctx(c, pair);
    if (st) return st;
    if ((st = settle_pending(c))) return st; // an unverified asynchronous solve still needs the old inputs
    if (!prev || !curr) return fail(c, HSFLOW_E_ARG, "null frame pointer");
    if (ps < (size_t)c->W || cs < (size_t)c->W) return fail(c, HSFLOW_E_SIZE, "frame stride smaller than width");
    if ((st = preprocess_frame(c, c->dA + pair * c->plane, prev, ps, false, true, 0, false))) return st;
    if ((st = preprocess_frame(c, c->dB + pair * c->plane, curr, cs, false, true, 1, false))) return st;
    c->frames_set = true;
    c->coef_valid = false;
    return HSFLOW_OK;
}

int hsflow_set_async_reduce(hsflow_ctx *c, int on)
{
    int st = check_ctx(c, 0);
    if (st) return st;
    if ((st = settle_pending(c))) return st;
    if (on && !c->hMark) { // the marker word (page-locked, device-visible) and its device-side counter
        HS_HIP(c, hipMalloc((void **)&c->dSeq, 2 * sizeof(unsigned)));
        HS_HIP(c, hipMemsetAsync(c->dSeq, 0, 2 * sizeof(unsigned), c->stream));
        HS_HIP(c, hipHostMalloc((void **)&c->hMark, 64, hipHostMallocMapped | hipHostMallocCoherent));
        HS_HIP(c, hipHostGetDevicePointer((void **)&c->hMarkDev, c->hMark, 0));
        *c->hMark = 0u;
        c->mark_issued = 0;
    }
    c->last_marked = false;
    c->async_reduce = on != 0;
    return HSFLOW_OK;
}

int hsflow_set_eps_rows(hsflow_ctx *c, int first_row, int rows)
{
    int st = check_ctx(c, 0);
    if (st) return st;
    if (rows > 0 && (first_row < 0 || first_row + rows > c->H)) return fail(c, HSFLOW_E_SIZE, "Eps row window outside the frame");
    if ((st = settle_pending(c))) return st;
    const int r0 = rows > 0 ? first_row : 0, n = rows > 0 ? rows : 0;
    if (r0 != c->eps_row0 || n != c->eps_rows) {
        drop_graphs(c); // captured launches carry the window in their geometry
        c->plan_cache.clear();
        c->lastl.valid = false;
        c->eps_row0 = r0;
        c->eps_rows = n;
    }
    return HSFLOW_OK;
}

int hsflow_solve_probe(hsflow_ctx *c, const hsflow_params *pp, float *sweep_eps)
{
    int st = check_ctx(c, 0);
    if (st) return st;
    if (!pp || pp->struct_size != sizeof(hsflow_params)) return fail(c, HSFLOW_E_ARG, "params null or struct_size mismatch");
    if (!sweep_eps) return fail(c, HSFLOW_E_ARG, "sweep_eps is null");
    if (pp->mode != HSFLOW_MODE_CV) return fail(c, HSFLOW_E_ARG, "hsflow_solve_probe: CV mode only");
    if (pp->max_iter <= 0 || pp->max_iter > (1 << 16)) return fail(c, HSFLOW_E_ARG, "hsflow_solve_probe: max_iter must be 1 .. 65536");
    hsflow_params q = *pp;
    q.term_type = HSFLOW_TERM_ITER | HSFLOW_TERM_EPS;
    q.epsilon = 0.0;  // Eps < 0 never holds: every sweep of the budget runs
    q.profile = 0;
    if (q.kernel == HSFLOW_KERNEL_PERSIST) q.kernel = HSFLOW_KERNEL_STRIP;
    c->force_exact = true; // the per-sweep pass, not the witness pass
    st = solve_impl(c, &q, false);
    c->force_exact = false;
    if (st) return st;
    if ((int)c->sweep_eps.size() != pp->max_iter) return fail(c, HSFLOW_E_STATE, "hsflow_solve_probe: the per-sweep pass did not run");
    std::memcpy(sweep_eps, c->sweep_eps.data(), (size_t)pp->max_iter * sizeof(float));
    return HSFLOW_OK;
}

int hsflow_take_verdict(hsflow_ctx *c, int *proven)
{
    int st = check_ctx(c, 0);
    if (st) return st;
    if (!proven) return fail(c, HSFLOW_E_ARG, "proven is null");
    if (!c->pend.active) return fail(c, HSFLOW_E_STATE, "no asynchronous ITER|EPS solve is waiting for its early-stop check");
    *proven = 0;
    return settle_pending(c, proven);
}

int hsflow_solve(hsflow_ctx *c, const hsflow_params *p) { return solve_impl(c, p, false); }
int hsflow_solve_async(hsflow_ctx *c, const hsflow_params *p) { return solve_impl(c, p, true); }

int hsflow_wait_solve(hsflow_ctx *c)
{
    int st = check_ctx(c, 0);
    if (st) return st;
    // (with the witness words reduced in-stream the owed check is settled by waiting for the solve's event; otherwise
    // settle_pending enqueues the reduction and waits for the stream)
    if ((st = settle_pending(c))) return st;
    if (c->last_marked) { if ((st = wait_marker(c, c->mark_issued))) return st; }
    else HS_HIP(c, hipStreamSynchronize(c->stream));
    return check_persist(c);
}

int hsflow_synchronize(hsflow_ctx *c)
{
    int st = check_ctx(c, 0);
    if (st) return st;
    if ((st = settle_pending(c))) return st;
    HS_HIP(c, hipStreamSynchronize(c->stream));
    return check_persist(c);
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
    if ((st = check_persist(c))) return st;
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

int hsflow_flow_view_device(hsflow_ctx *c, int pair, const float **du, const float **dv, size_t *stride_bytes)
{
    int st = check_ctx(c, pair);
    if (st) return st;
    if (!du || !dv || !stride_bytes) return fail(c, HSFLOW_E_ARG, "null out pointer");
    if ((st = settle_pending(c))) return st;
    if (c->last_marked) { if ((st = wait_marker(c, c->mark_issued))) return st; } // (the flow is final behind the last solve)
    else HS_HIP(c, hipStreamSynchronize(c->stream));
    if ((st = check_persist(c))) return st;
    *du = c->dU[c->cur] + pair * c->plane;
    *dv = c->dV[c->cur] + pair * c->plane;
    *stride_bytes = (size_t)c->P * sizeof(float);
    return HSFLOW_OK;
}

int hsflow_get_flow_device(hsflow_ctx *c, int pair, int row0, int nrows, void *du, size_t us, void *dv, size_t vs)
{
    int st = flow_rows_args(c, pair, row0, nrows, du, us, dv, vs);
    if (st) return st;
    // rows handed to another consumer on the device (halo exchange) must be final: an ITER|EPS check that
    // hsflow_solve_async still owes is settled first (a re-run would change them)
    if ((st = settle_pending(c))) return st;
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
    c->lastl.valid = false;                  // (its last_eps can no longer be measured: the flow is about to change)
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
    float *sx = (float *)c->dScratch, *sy = sx + (size_t)c->W * c->H, *stt = sy + (size_t)c->W * c->H;
    if (c->coef_mode == HSFLOW_MODE_CLASSIC) // (each mode packs its derivatives its own way)
        hipLaunchKernelGGL(hsk::k_unpack_deriv<true>, dim3((c->W + 255) / 256, c->H), dim3(256), 0, c->stream,
                           c->dCoef + pair * c->plane, sx, sy, stt, c->W, c->H, c->P);
    else
        hipLaunchKernelGGL(hsk::k_unpack_deriv<false>, dim3((c->W + 255) / 256, c->H), dim3(256), 0, c->stream,
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

int hsflow_get_info(hsflow_ctx *c, hsflow_info *info) { return hsflow_get_info_ex(c, info, 1); }

int hsflow_get_info_ex(hsflow_ctx *c, hsflow_info *info, int measure_eps)
{
    if (!c) return fail(nullptr, HSFLOW_E_ARG, "null context");
    if (!info || info->struct_size != sizeof(hsflow_info)) return fail(c, HSFLOW_E_ARG, "info null or struct_size mismatch");
    int st = settle_pending(c); // iterations_done of an asynchronous ITER|EPS solve
    if (st) return st;
    if (hipSetDevice(c->device) != hipSuccess) return fail(c, HSFLOW_E_DEVICE, "hipSetDevice failed");
    if (measure_eps && (st = measure_last_eps(c))) return st; // ... and its last_eps, measured now that somebody asks
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

int hsflow_plan_query(int width, int height, int n_pairs, const hsflow_params *pp, hsflow_info *out)
{
    if (!pp || pp->struct_size != sizeof(hsflow_params)) return fail(nullptr, HSFLOW_E_ARG, "params null or struct_size mismatch");
    if (!out || out->struct_size != sizeof(hsflow_info)) return fail(nullptr, HSFLOW_E_ARG, "info null or struct_size mismatch");
    if (width <= 0 || height <= 0 || n_pairs <= 0) return fail(nullptr, HSFLOW_E_SIZE, "width, height, n_pairs must be positive");
    if ((long long)round_up(width, 64) * height > (1LL << 30)) return fail(nullptr, HSFLOW_E_SIZE, "plane too large (pitch*height > 2^30)");
    hsflow_ctx c; // host-side shell only: the planners read sizes, nothing touches a device
    c.W = width; c.H = height; c.N = n_pairs;
    c.P = round_up(width, 64);
    c.plane = (long long)c.P * height;
    std::memset(&c.info, 0, sizeof(c.info));
    c.info.struct_size = sizeof(hsflow_info);
    c.info.width = width; c.info.height = height; c.info.n_pairs = n_pairs; c.info.pitch = c.P;
    const hsflow_params &p = *pp;
    int st = HSFLOW_OK;
    if (p.mode == HSFLOW_MODE_CV) {
        SolveSetup S;
        st = prepare_solve(&c, p, false, S);
        if (!st) {
            const long long b = S.budget > (1LL << 30) ? 0 : S.budget;
            c.info.jacobi_launches = S.persist ? 1 : S.multi ? (int)((b + S.T - 1) / S.T) : (int)b;
        }
    } else if (p.mode == HSFLOW_MODE_CLASSIC || p.mode == HSFLOW_MODE_CLASSIC_AS_SHIPPED) {
        ClassicSetup S;
        st = prepare_classic(&c, p, S);
    } else st = fail(&c, HSFLOW_E_ARG, "unknown mode");
    if (st) { g_create_error = c.err; return st; }
    *out = c.info;
    return HSFLOW_OK;
}

void hsflow_release_cached(void)
{
    std::lock_guard<std::mutex> lock(g_oneshot_mutex);
    if (g_oneshot) hsflow_destroy(g_oneshot);
    g_oneshot = nullptr;
}

} // extern "C"
