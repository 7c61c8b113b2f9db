/*
 * hsflow.h -- C ABI of the MI355X-native Horn-Schunck optical-flow solver (libhsflow.so).
 *
 * This is the drop-in boundary for the hot path of miczi/OpticalFlowHS:
 * everything the reference's HSOpticalFlowOpenCL::setupCL / runDerivatives / runCLKernels /
 * cleanup did through OpenCL goes through these entry points instead.  Plain pointers and
 * sizes only; no C++ or torch types.  All planes are single-channel and PLANAR (u8 frames,
 * fp32 flow) -- not the reference's float4-per-pixel layout (HSOpticalFlowOpenCL.hpp:29-41).
 *
 * Reference interfaces replaced, one by one:
 *   hsflow_create / hsflow_destroy ...... HSOpticalFlowOpenCL::setupCL  (HSOpticalFlowOpenCL.cpp:67-319)
 *                                          HSOpticalFlowOpenCL::cleanup  (HSOpticalFlowOpenCL.cpp:849-892)
 *   hsflow_set_frames_u8[_device] ....... clEnqueueWriteBuffer of inputImageBuffer1/2
 *                                          (HSOpticalFlowOpenCL.cpp:339-357); frames are what
 *                                          readInputImage produced (:4-44) but kept as u8
 *   hsflow_solve / hsflow_solve_async ... runDerivatives() + iterations x runCLKernels()
 *                                          (HSOpticalFlowOpenCL.cpp:321-474, :476-679, loop :749-751)
 *                                          and, argument for argument, cvCalcOpticalFlowHS
 *                                          (OpenCV2.1/include/cv.h:481-483) as called at
 *                                          OpticalFlowOpenCV.cpp:29,94
 *   hsflow_get_flow[_device] ............ clEnqueueReadBuffer of uBuffer/vBuffer
 *                                          (HSOpticalFlowOpenCL.cpp:655-675); read at :765-767
 *   hsflow_get_derivatives .............. clEnqueueReadBuffer of Ex/Ey/Et (:437-468)
 *   hsflow_calc_optical_flow_hs_8u32f ... one-shot form with the argument list of OpenCV's
 *                                          icvCalcOpticalFlowHS_8u32fR (cv210.dll VA 0x1012e040)
 *   status codes ........................ SDK_SUCCESS 0 / SDK_FAILURE 1 (SDKUtil/include/SDKCommon.hpp:23-24)
 *                                          become 0 / enumerated non-zero
 *
 * Threading: a context is single-owner (not thread-safe); one context per (thread, device).
 * All device work of a context is issued on ONE HIP stream (given at creation, or its own).
 */
#ifndef HSFLOW_H_
#define HSFLOW_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HSFLOW_VERSION_MAJOR 0
#define HSFLOW_VERSION_MINOR 5 /* 0.5: hsflow_multi_*, hsflow_slab_*, hsflow_set_row_origin, hsflow_get_info_ex */

/* status codes (0 = success, like SDK_SUCCESS) */
#define HSFLOW_OK 0
#define HSFLOW_E_ARG 1     /* null pointer / bad enum / bad struct size            */
#define HSFLOW_E_SIZE 2    /* non-positive size, stride too small or misaligned    */
#define HSFLOW_E_DEVICE 3  /* a HIP call failed; text in hsflow_last_error         */
#define HSFLOW_E_OOM 4     /* host or device allocation failed                     */
#define HSFLOW_E_STATE 5   /* call order (solve before frames were set, ...)       */
#define HSFLOW_E_NOTERM 6  /* termination rule that would never stop               */

/* termination flags: values of CV_TERMCRIT_ITER / CV_TERMCRIT_EPS (cxtypes.h:894-896) */
#define HSFLOW_TERM_ITER 1
#define HSFLOW_TERM_EPS 2

/* discretisation */
#define HSFLOW_MODE_CV 0      /* cvCalcOpticalFlowHS semantics: Sobel/8 on frame A, 4-neighbour
                                 mean, lambda (graded parity target)            */
#define HSFLOW_MODE_CLASSIC 1 /* Kernels.cl semantics: 2x2x2 cube derivatives, 1/6-1/12 mean,
                                 alpha^2, with the v update restored.
                                 ITER termination only (the reference's loop has no other rule).
                                 Kernels: SIMPLE, FUSED, STRIP (rows per lane 2..8); AUTO takes
                                 STRIP wherever the image has an aligned shape for it and
                                 2^-20 <= alpha <= 2^20, else FUSED.  All bit-identical.          */

#define HSFLOW_MODE_CLASSIC_AS_SHIPPED 2 /* Kernels.cl exactly as shipped: u_v_updateKernel writes u only
                                 (Kernels.cl:86), v stays at its starting value.  Reproduces the
                                 pictures the reference's OpenCL route wrote; for verification      */

/* Jacobi kernel selection */
#define HSFLOW_KERNEL_AUTO 0
#define HSFLOW_KERNEL_SIMPLE 1 /* one iteration per launch, straight from HBM/L2             */
#define HSFLOW_KERNEL_FUSED 2  /* `fuse_steps` iterations per launch on an LDS tile with halo */
#define HSFLOW_KERNEL_STRIP 3  /* `fuse_steps` iterations per launch on register-resident strips:
                                  a wavefront holds 256 columns x strip_rows rows in VGPRs, DPP
                                  for left/right, LDS only for strip-edge rows (AUTO picks this, or
                                  its folded form below ~1.5 Mpixel per context)              */
#define HSFLOW_KERNEL_FOLD 4   /* as STRIP, two 128-column strips per wavefront (half the LDS
                                  exchange; inner boundary swapped in registers)              */
#define HSFLOW_KERNEL_PERSIST 5 /* STRIP as ONE launch per solve: workgroups keep their tile in registers across
                                  phases of `fuse_steps` iterations and swap halos through HBM, ordered by per-tile
                                  phase counters (replaces the host loop HSOpticalFlowOpenCL.cpp:748-752 inside one
                                  kernel).  Needs every workgroup resident at once: one tile per CU at most, width
                                  a multiple of 4, ITER or asynchronous ITER|EPS, and the context must be the only one alive on its
                                  device (two persistent grids could starve each other; every wait inside is bounded
                                  and a timed-out solve is repeated launch by launch).  On request only: a phase
                                  boundary measures as dear as a kernel boundary, so AUTO keeps STRIP.
                                  HSFLOW_E_SIZE when it cannot run.  hsflow_info.kernel reports STRIP,
                                  hsflow_info.persistent the number of phases.                                  */

typedef struct hsflow_ctx hsflow_ctx;

typedef struct hsflow_params {
    uint32_t struct_size; /* = sizeof(hsflow_params); guards ABI growth                  */
    int32_t mode;         /* HSFLOW_MODE_*                                               */
    float lambda;         /* CV mode: Lagrange multiplier of cvCalcOpticalFlowHS         */
    float alpha;          /* CLASSIC mode: smoothness weight (Kernels.cl:85)             */
    int32_t term_type;    /* HSFLOW_TERM_ITER | HSFLOW_TERM_EPS                          */
    int32_t max_iter;     /* CvTermCriteria.max_iter                                     */
    double epsilon;       /* CvTermCriteria.epsilon (caller rounds through float if it
                             wants cvTermCriteria()'s behaviour, cxtypes.h:912)          */
    int32_t use_previous; /* 0: u=v=0 first (reference behaviour); 1: continue from the
                             flow currently held by the context                          */
    int32_t kernel;       /* HSFLOW_KERNEL_*                                             */
    int32_t fuse_steps;   /* FUSED: iterations per launch, 0 = auto                      */
    int32_t tile_w;       /* FUSED: core tile width  (multiple of 4), 0 = auto           */
    int32_t tile_h;       /* FUSED: core tile height, 0 = auto                           */
    int32_t threads;      /* FUSED: workgroup size 256/512/1024; STRIP: 64 x wavefronts
                             per workgroup (64..1024); 0 = auto                          */
    int32_t strip_rows;   /* STRIP / FOLD: rows held per lane (1..8), 0 = auto           */
    int32_t reuse_derivatives; /* 1: skip the derivative pass if the frames did not change since
                             the last solve of this context (row-slab chunks, warm starts)   */
    int32_t use_graph;    /* 1: capture the launch sequence in a hipGraph and replay it  */
    int32_t profile;      /* 1: bracket every kernel with HIP events (see hsflow_info)   */
} hsflow_params;

typedef struct hsflow_info {
    uint32_t struct_size;
    int32_t width, height, n_pairs, pitch; /* pitch in elements, same for every plane    */
    int32_t iterations_done;  /* sweeps executed by the last solve                        */
    float last_eps;           /* Eps of the last sweep (EPS termination only).  An asynchronous
                                 ITER|EPS solve does not measure it; hsflow_get_info then runs that
                                 solve's last launch once more to obtain it (NaN if the flow was
                                 changed through hsflow_set_flow_device in between)               */
    int32_t kernel;           /* kernel actually used                                     */
    int32_t fuse_steps, tile_w, tile_h, threads, groups_per_thread;
    int32_t tiles;            /* workgroups per fused launch                              */
    int32_t lds_bytes;        /* dynamic LDS per workgroup                                */
    int32_t jacobi_launches;  /* launches of the Jacobi kernel in the last solve          */
    float deriv_ms;           /* profile=1: derivative kernel time                        */
    float jacobi_ms;          /* profile=1: sum of Jacobi kernel times                    */
    float solve_ms;           /* profile=1: first event to last event of the solve        */
    int32_t eps_rerun;        /* ITER|EPS: 1 if the fast pass could not prove "no early stop"
                                 and the solve was repeated with Eps measured in every sweep */
    int32_t deriv_fused;      /* 1 if the derivative pass ran inside the first Jacobi launch of the
                                 last solve instead of as a kernel of its own                    */
    int32_t persistent;       /* phases of the one persistent launch the last solve ran as
                                 (HSFLOW_KERNEL_PERSIST), 0 for a launch per `fuse_steps` iterations */
} hsflow_info;

/* --- lifecycle ---------------------------------------------------------------------------- */

/* Fills p with the defaults (CV mode, lambda 1, ITER|EPS, 100 iterations, eps 1e-6f, auto). */
void hsflow_default_params(hsflow_params *p);

/* n_pairs independent image pairs of width x height live in one context (n_pairs >= 1).
 * device: HIP ordinal.  stream: the hipStream_t all work is issued on (e.g. torch's current
 * stream; NULL is the device's default stream).  own_stream != 0: ignore `stream` and create a
 * private non-blocking stream instead. */
int hsflow_create(hsflow_ctx **out, int device, int width, int height, int n_pairs, void *stream,
                  int own_stream);
int hsflow_destroy(hsflow_ctx *ctx); /* NULL is accepted; idempotent per handle */
/* Row-slab decomposition: the context holds rows [first_row, first_row + height) of a larger frame.
 * The Jacobi update adds its four neighbours in an order that depends on the pixel's checkerboard parity
 * (x + y) & 1 in the FRAME; telling the context where its row 0 sits (only the parity matters) makes a slab
 * compute bit for bit what the whole-frame solve computes for the same pixels.  Default 0. */
int hsflow_set_row_origin(hsflow_ctx *ctx, int first_row);

/* The launch planners count on `compute_units` CUs instead of the whole chip (0: the whole chip again).  For a context
 * whose solves run BESIDE other contexts' solves -- the slots of a pair pipeline do this by themselves, with
 * CUs / depth -- so that each solve takes the shape that costs the least CU-time (few large tiles, little halo
 * redundancy) rather than the one that spreads a small frame over every CU to shorten its own latency: at the reference's
 * 600x480 default (main.cpp:4-8) that is the difference between 210 tiles of 88x16 and 36 of 216x40 per launch.
 * Results are bit-identical whatever the shape. */
int hsflow_set_cu_share(hsflow_ctx *ctx, int compute_units);

/* on != 0: an asynchronous ITER|EPS solve enqueues the reduction of its witness words right behind its last launch (a
 * small kernel per solve on the context's stream) instead of leaving it to whoever settles the check, and EVERY asynchronous
 * solve is followed by a one-thread kernel that writes a running count to page-locked memory: settling and hsflow_wait_solve
 * then poll that word -- no launch, no stream-wide wait, no event record (which costs a stream of solves 6 %).  Pays where the stream is not the bottleneck -- the
 * slots of a pair pipeline set it: their streams overlap, and for small frames the host's time per pair is what bounds
 * the stream.  Off by default: back-to-back solves on ONE stream would pay the kernel and its boundary every time. */
int hsflow_set_async_reduce(hsflow_ctx *ctx, int on);

/* Waits until the last solve of THIS context has finished and settles the early-stop check it may owe.  With
 * hsflow_set_async_reduce on, that is a poll of the marker behind the solve: unlike hsflow_synchronize it does not wait for
 * what other contexts have enqueued on the same stream since (the slots of a pair pipeline share streams); without it,
 * the same as hsflow_synchronize. */
int hsflow_wait_solve(hsflow_ctx *ctx);

/* --- building blocks for drivers that run ONE solve over several contexts (row slabs, hsflow_slab_*) ----------- */

/* Only the changes of rows [first_row, first_row + rows) count for Eps and for the witness of ITER|EPS solves (rows <= 0:
 * the whole frame again).  A row slab sets its OWNED rows: its halo rows repeat the neighbour's and go stale towards
 * the slab's edge inside a chunk, so their changes say nothing about the frame's Eps
 * (cv210.dll@0x1012ed2f-0x1012eda5 takes the maximum over the frame).  Strip and simple kernels. */
int hsflow_set_eps_rows(hsflow_ctx *ctx, int first_row, int rows);
/* Exactly params->max_iter sweeps (whatever params->term_type says: nothing stops them) with the Eps of every sweep
 * -- over the rows of hsflow_set_eps_rows -- written to sweep_eps[0 .. max_iter).  Synchronous.  What a driver needs to
 * find the stopping sweep of a solve that is spread over several contexts: Eps_k of the frame = the maximum of the
 * contexts' Eps_k. */
int hsflow_solve_probe(hsflow_ctx *ctx, const hsflow_params *params, float *sweep_eps);
/* The early-stop check an asynchronous ITER|EPS solve still owes, looked at WITHOUT acting on it: waits for the stream;
 * *proven = 1 if the witness words prove that Eps stayed >= epsilon in every sweep (over the rows of
 * hsflow_set_eps_rows), 0 if they do not -- the flow of the whole budget stands either way and nothing is re-run.  A
 * proof from ANY context of a spread solve covers the frame (its Eps is the maximum).  HSFLOW_E_STATE if nothing is owed. */
int hsflow_take_verdict(hsflow_ctx *ctx, int *proven);

/* --- frames in ---------------------------------------------------------------------------- */

/* Host u8 single-channel frames, row strides in bytes (>= width).  Synchronous. */
int hsflow_set_frames_u8(hsflow_ctx *ctx, int pair, const uint8_t *prev, size_t prev_stride,
                         const uint8_t *curr, size_t curr_stride);
/* Same, but only enqueued on ctx's stream: the host buffers must stay valid and unchanged until
 * the stream has passed the copy (hsflow_synchronize).  Truly asynchronous only from page-locked
 * memory (hsflow_host_alloc / hsflow_host_register); from pageable memory HIP stages the copy. */
int hsflow_set_frames_u8_async(hsflow_ctx *ctx, int pair, const uint8_t *prev, size_t prev_stride,
                               const uint8_t *curr, size_t curr_stride);
/* Same, source already in device memory on ctx's device; enqueued on ctx's stream. */
int hsflow_set_frames_u8_device(hsflow_ctx *ctx, int pair, const void *d_prev, size_t prev_stride,
                                const void *d_curr, size_t curr_stride);
/* Host 8-bit BGR frames (3 bytes/pixel): BGR->gray then optional 3x3 box blur on the GPU, i.e.
 * the reference CPU route's pre-processing (OpticalFlowOpenCV.cpp:17,20,27-28). Synchronous. */
int hsflow_set_frames_bgr8(hsflow_ctx *ctx, int pair, const uint8_t *prev_bgr, size_t prev_stride,
                           const uint8_t *curr_bgr, size_t curr_stride, int blur3x3);
/* Host u8 gray frames, 3x3 box blur (cvSmooth CV_BLUR, replicate border) on the GPU. Synchronous. */
int hsflow_set_frames_gray8_blur(hsflow_ctx *ctx, int pair, const uint8_t *prev, size_t prev_stride,
                                 const uint8_t *curr, size_t curr_stride);
/* Asynchronous forms of the two above: only enqueued on ctx's stream, host buffers owned by the context
 * until hsflow_synchronize. */
int hsflow_set_frames_bgr8_async(hsflow_ctx *ctx, int pair, const uint8_t *prev_bgr, size_t prev_stride,
                                 const uint8_t *curr_bgr, size_t curr_stride, int blur3x3);
int hsflow_set_frames_gray8_blur_async(hsflow_ctx *ctx, int pair, const uint8_t *prev, size_t prev_stride,
                                       const uint8_t *curr, size_t curr_stride);
/* Streaming (camera loop, HSOpticalFlowOpenCL.cpp:810-834): the current frame becomes the
 * previous one on the device and only the new frame is uploaded. */
int hsflow_push_frame_u8(hsflow_ctx *ctx, int pair, const uint8_t *next, size_t next_stride);

/* --- solve -------------------------------------------------------------------------------- */

/* Derivative pass + Jacobi iterations for every pair of the context.  hsflow_solve returns
 * after the device finished; hsflow_solve_async only enqueues (no profile) and the caller
 * synchronises the stream or calls hsflow_synchronize.  Asynchronous solves take ITER termination
 * with any kernel, or ITER|EPS (the reference's call, OpticalFlowOpenCV.cpp:29) with the strip /
 * fold kernels (what AUTO picks): the early-stop check is then owed until hsflow_synchronize / hsflow_get_flow /
 * hsflow_get_info / the next solve settles it -- if the fast pass cannot prove that the stop
 * never fired, the solve is repeated exactly (hsflow_info.eps_rerun = 1), so flow copied out by
 * an earlier hsflow_get_flow_async has to be fetched again in that case.  One exception keeps a
 * stream of solves free of host round trips: an hsflow_solve_async that repeats the owed solve bit
 * for bit (same parameters, use_previous = 0; the frames cannot have changed, setting them settles)
 * recomputes the same result and takes the owed check over instead of waiting for it. */
int hsflow_solve(hsflow_ctx *ctx, const hsflow_params *params);
int hsflow_solve_async(hsflow_ctx *ctx, const hsflow_params *params);
int hsflow_synchronize(hsflow_ctx *ctx);

/* --- results out -------------------------------------------------------------------------- */

/* fp32 flow to host, row strides in bytes (multiple of 4, >= 4*width).  Synchronous. */
int hsflow_get_flow(hsflow_ctx *ctx, int pair, float *u, size_t u_stride, float *v, size_t v_stride);
/* Same, only enqueued on ctx's stream (after the solve enqueued before it); u, v are complete
 * after hsflow_synchronize.  Page-locked destination for a real overlap with other streams. */
int hsflow_get_flow_async(hsflow_ctx *ctx, int pair, float *u, size_t u_stride, float *v, size_t v_stride);
/* Row range [row0, row0+nrows) of the flow to / from device memory, on ctx's stream (used for
 * the row-slab halo exchange).  set_ writes into the flow the next
 * use_previous=1 solve continues from.  Both settle an ITER|EPS check that hsflow_solve_async still
 * owes (they wait for the stream in that case); after an ITER-only solve they only enqueue. */
/* Where the context holds the current flow of `pair`, without a copy: device pointers to row 0 and the row
 * stride in bytes (rows are `width` floats; the pitch is that of hsflow_info).  Settles an ITER|EPS check that
 * hsflow_solve_async still owes and waits for the stream, so the planes are final; they stay valid and unchanged
 * until the next call that changes this context's flow (solve, set_flow_device).  What a consumer on the device
 * (rendering, the next stage of a pipeline) reads instead of HSOpticalFlowOpenCL.cpp:655-675's blocking read-back. */
int hsflow_flow_view_device(hsflow_ctx *ctx, int pair, const float **d_u, const float **d_v, size_t *stride_bytes);
int hsflow_get_flow_device(hsflow_ctx *ctx, int pair, int row0, int nrows, void *d_u,
                           size_t u_stride, void *d_v, size_t v_stride);
int hsflow_set_flow_device(hsflow_ctx *ctx, int pair, int row0, int nrows, const void *d_u,
                           size_t u_stride, const void *d_v, size_t v_stride);
/* Derivative planes of the last solve as fp32 (CV: Ix, Iy, It; CLASSIC: Ex, Ey, Et). */
int hsflow_get_derivatives(hsflow_ctx *ctx, int pair, float *dx, float *dy, float *dt,
                           size_t stride);
/* The pre-processed u8 frames the solver actually sees (after gray/blur), to host. */
int hsflow_get_frames_u8(hsflow_ctx *ctx, int pair, uint8_t *prev, size_t prev_stride,
                         uint8_t *curr, size_t curr_stride);

/* --- introspection ------------------------------------------------------------------------ */

int hsflow_get_info(hsflow_ctx *ctx, hsflow_info *info);
/* measure_last_eps = 0: as hsflow_get_info, but last_eps of an asynchronous ITER|EPS solve is left as it is
 * (NaN until measured) instead of running that solve's last launch again; 1: hsflow_get_info. */
int hsflow_get_info_ex(hsflow_ctx *ctx, hsflow_info *info, int measure_last_eps);
const char *hsflow_last_error(hsflow_ctx *ctx); /* ctx may be NULL: last create() error */
const char *hsflow_status_string(int status);
int hsflow_version(void); /* major*1000 + minor */
int hsflow_device_count(int *count);

/* --- page-locked host memory (staging for the async copies) ------------------------------- */

/* The reference aliased host planes into the device with CL_MEM_USE_HOST_PTR
 * (HSOpticalFlowOpenCL.cpp:184-228); here the caller keeps ownership of its host buffers and may
 * page-lock them so that uploads / downloads overlap the solver. */
int hsflow_host_alloc(void **out, size_t bytes);   /* hipHostMalloc */
int hsflow_host_free(void *p);                     /* NULL accepted */
int hsflow_host_register(void *p, size_t bytes);   /* page-lock memory the caller allocated */
int hsflow_host_unregister(void *p);

/* --- pair pipeline: host frames in, host flow out, copies overlapped with solves ----------- */

/* Independent pairs streamed through ONE device (BASELINE config C4, SURVEY.md 8e: "one host
 * thread + >= 2 streams per GPU, double-buffered staging").  The pipeline owns `depth` single-pair
 * contexts, each on its own stream; submit() enqueues upload -> solve -> download of one pair on
 * the next slot and returns at once, so that the upload of pair i+1 and the download of pair i-1
 * run beside the solve of pair i.  It replaces the per-pair body of the reference's run()
 * (HSOpticalFlowOpenCL.cpp:744-767: write frames, derivatives, iterations, read u, v).
 * The host buffers of a submitted pair belong to the pipeline until wait(ticket) returned; use
 * page-locked memory for them.  Termination: ITER, or ITER|EPS (strip / fold kernel) -- whatever
 * hsflow_solve_async accepts; a pair whose early stop fired is re-solved inside wait().
 * Single-owner like a context; one pipeline per (thread, device). */
typedef struct hsflow_pipeline hsflow_pipeline;
int hsflow_pipeline_create(hsflow_pipeline **out, int device, int width, int height, int depth);
/* The same with the `depth` slots spread over `lanes` streams (slot k on stream k mod lanes; 1 <= lanes <= depth;
 * hsflow_pipeline_create: lanes = depth, which is what host-memory pairs want -- upload, solve and download of three
 * pairs on three queues).  For pairs that are already in device memory TWO lanes with 4 - 8 slots are the shape to use:
 * two solves side by side is what fills the chip's gaps, and the further slots keep both streams' queues
 * full while the host settles and refills the oldest slot -- 0.125 ms per 1080p / 100 pair against 0.139 with two slots
 * on two streams. */
int hsflow_pipeline_create_lanes(hsflow_pipeline **out, int device, int width, int height, int depth, int lanes);
int hsflow_pipeline_destroy(hsflow_pipeline *pl); /* drains first; NULL accepted */
/* ticket (optional out): 0, 1, 2, ... in submission order.  Blocks only while the slot it is
 * about to reuse (ticket - depth) is still running. */
int hsflow_pipeline_submit(hsflow_pipeline *pl, const uint8_t *prev, size_t prev_stride,
                           const uint8_t *curr, size_t curr_stride, float *u, size_t u_stride,
                           float *v, size_t v_stride, const hsflow_params *params, uint64_t *ticket);
/* Same with the frames in another layout; the CPU route's pre-processing (OpticalFlowOpenCV.cpp:17-28)
 * then runs on the device as part of the pair's queue. */
#define HSFLOW_FRAMES_GRAY8 0      /* u8 gray, as hsflow_pipeline_submit                      */
#define HSFLOW_FRAMES_GRAY8_BLUR 1 /* u8 gray, 3x3 box blur on the device (cvSmooth CV_BLUR)  */
#define HSFLOW_FRAMES_BGR8 2       /* 8-bit BGR (3 bytes / pixel): BGR->gray on the device    */
#define HSFLOW_FRAMES_BGR8_BLUR 3  /* BGR->gray and blur: what runFromImg does before solving */
int hsflow_pipeline_submit_ex(hsflow_pipeline *pl, int format, const uint8_t *prev, size_t prev_stride,
                              const uint8_t *curr, size_t curr_stride, float *u, size_t u_stride,
                              float *v, size_t v_stride, const hsflow_params *params, uint64_t *ticket);
/* The same for a stream of pairs that are ALREADY IN DEVICE MEMORY (a decoder's or a camera's output, the frames of a
 * resident sequence): the frames are copied device to device into the slot, the flow stays in the slot and is handed
 * out by hsflow_pipeline_flow_device -- no host buffer anywhere.  This is the reference's camera loop
 * (OpticalFlowOpenCV.cpp:91-95: fresh frames, ITER|EPS, every pair) at the speed of the solver: while pair k's
 * early-stop check is still owed, pair k+1 is already running on the next slot's stream; the check is looked at when
 * somebody asks for pair k (wait / flow_device / info / the slot's reuse `depth` submissions later), and a pair whose
 * early stop fired is re-solved from its slot's frames, which nothing has touched.  The caller's frame buffers must
 * be complete when submit is called (they are read by a copy enqueued on the slot's stream) and may be reused once
 * any later call on the pipeline has returned that waited for this ticket. */
int hsflow_pipeline_submit_device(hsflow_pipeline *pl, const void *d_prev, size_t prev_stride, const void *d_curr,
                                  size_t curr_stride, const hsflow_params *params, uint64_t *ticket);
/* wait(ticket) + where that pair's flow lies (hsflow_flow_view_device of its slot): valid until `depth` further
 * pairs have been submitted.  HSFLOW_E_STATE if the slot has been reused already. */
int hsflow_pipeline_flow_device(hsflow_pipeline *pl, uint64_t ticket, const float **d_u, const float **d_v, size_t *stride_bytes);
int hsflow_pipeline_wait(hsflow_pipeline *pl, uint64_t ticket); /* u, v of that pair are complete */
/* wait(ticket) + iterations_done, last_eps, eps_rerun ... of that pair; HSFLOW_E_STATE once a later
 * pair has finished on the same slot (ask before submitting `depth` more pairs). */
int hsflow_pipeline_info(hsflow_pipeline *pl, uint64_t ticket, hsflow_info *info);
int hsflow_pipeline_drain(hsflow_pipeline *pl);                 /* wait for everything submitted */
int hsflow_pipeline_depth(hsflow_pipeline *pl);
const char *hsflow_pipeline_last_error(hsflow_pipeline *pl);    /* pl may be NULL: create() error */

/* --- several GPUs from one host process ----------------------------------------------------- */

/* Independent pairs over the GPUs of a node (BASELINE config C4): one pair pipeline of `depth` slots per
 * device, each driven by a host thread of its own; pair i goes to devices[i mod ndev]; no collective.  What the
 * reference's run() did per pair on one device (HSOpticalFlowOpenCL.cpp:744-767), spread over the node.  A device may
 * be listed more than once.  submit() never blocks on the device; the host buffers of a pair (page-locked for
 * overlapped copies) belong to the library until wait(ticket) / drain() returned.  format: HSFLOW_FRAMES_*.
 * Single-owner: submit / wait / drain / destroy from one thread. */
typedef struct hsflow_multi hsflow_multi;
int hsflow_multi_create(hsflow_multi **out, const int *devices, int ndev, int width, int height, int depth);
int hsflow_multi_destroy(hsflow_multi *m); /* drains first; NULL accepted */
int hsflow_multi_devices(hsflow_multi *m);
int hsflow_multi_submit(hsflow_multi *m, int format, const uint8_t *prev, size_t prev_stride, const uint8_t *curr, size_t curr_stride,
                        float *u, size_t u_stride, float *v, size_t v_stride, const hsflow_params *params, uint64_t *ticket);
int hsflow_multi_wait(hsflow_multi *m, uint64_t ticket);
int hsflow_multi_drain(hsflow_multi *m);
const char *hsflow_multi_last_error(hsflow_multi *m); /* m may be NULL: create() error */

/* ONE large frame in row slabs over several GPUs (BASELINE config C5): slab k holds a contiguous range
 * of rows plus `halo` rows either side on devices[k]; the sweeps run in chunks of <= halo and after every chunk
 * neighbouring slabs swap `halo` rows of u and v device to device (peer copies over xGMI, ordered by events; the host
 * only enqueues) -- k-row halos every k sweeps instead of one row per sweep: the same bytes in k times fewer
 * messages.  The result is bit-identical to the whole-frame solve on one GPU (hsflow_set_row_origin keeps each slab
 * on the frame's checkerboard).  Termination: ITER, or ITER|EPS as the reference calls the solver
 * (OpticalFlowOpenCV.cpp:29,94 -- hsflow_default_params as they are): the frame's Eps is the maximum over the slabs'
 * owned rows, so a chunk that ANY slab's witness vouches for holds no stop; when none does, the solve is replayed to
 * that chunk and measured sweep by sweep, the maximum taken on the host between chunks, and it ends on the sweep the
 * one-context solve ends on.  use_previous continues from the flow of the last solve (halos refreshed first).  A
 * device may be listed more than once -- listing each device TWICE gives every GPU two sub-slabs on streams of their
 * own, so that one's peer copies run under the other's sweeps (hsflow_slab_create_overlapped does that).  The
 * one-process-per-GPU form of the same scheme, with RCCL send / recv, is opticalflowhs_amd/slab.py.
 * The cross-device copies (hipMemcpyPeerAsync) have only ever run between slabs that share a card (one-GPU boxes). */
typedef struct hsflow_slab hsflow_slab;
int hsflow_slab_create(hsflow_slab **out, const int *devices, int nslab, int width, int height, int halo);
int hsflow_slab_destroy(hsflow_slab *s); /* NULL accepted */
int hsflow_slab_count(hsflow_slab *s);
int hsflow_slab_rows(hsflow_slab *s, int k, int *lo, int *hi); /* rows [lo, hi) slab k owns */
int hsflow_slab_set_frames_u8(hsflow_slab *s, const uint8_t *prev, size_t prev_stride, const uint8_t *curr, size_t curr_stride);
int hsflow_slab_solve(hsflow_slab *s, const hsflow_params *params); /* returns after every device finished */
int hsflow_slab_exchanges(hsflow_slab *s);                          /* halo exchanges of the last solve */
int hsflow_slab_iterations_done(hsflow_slab *s);                    /* sweeps of the last solve (< max_iter: the early stop fired) */
int hsflow_slab_eps_measured(hsflow_slab *s);                       /* 1: no slab's witness held, Eps was measured sweep by sweep */
/* 2 * ndev slabs, devices[k] holding slabs 2k and 2k+1 (see above). */
int hsflow_slab_create_overlapped(hsflow_slab **out, const int *devices, int ndev, int width, int height, int halo);
int hsflow_slab_get_flow(hsflow_slab *s, float *u, size_t u_stride, float *v, size_t v_stride);
const char *hsflow_slab_last_error(hsflow_slab *s); /* s may be NULL: create() error */

/* Planner introspection, no device needed: the kernel, sweeps per launch, tile shape, workgroup size,
 * tiles per launch, LDS bytes and launch count hsflow_solve would use for a context of this size with
 * these parameters (same code path as the solver's own planning; text of a refusal in
 * hsflow_last_error(NULL)).  info->struct_size must be set. */
int hsflow_plan_query(int width, int height, int n_pairs, const hsflow_params *params, hsflow_info *info);

/* --- one-shot ------------------------------------------------------------------------------ */

/* Same argument list as OpenCV's inner routine behind cvCalcOpticalFlowHS: strides in bytes,
 * term_type/max_iter/epsilon = CvTermCriteria.  Uploads, solves and downloads on device 0, on a
 * context the library keeps between calls while width and height stay the same (the reference
 * calls cvCalcOpticalFlowHS once per camera frame, OpticalFlowOpenCV.cpp:94); serialised by a
 * mutex.  use_previous != 0 reads velx/vely as the starting flow. */
int hsflow_calc_optical_flow_hs_8u32f(const uint8_t *prev, const uint8_t *curr, int img_step,
                                      int width, int height, int use_previous, float *velx,
                                      float *vely, int vel_step, float lambda, int term_type,
                                      int max_iter, double epsilon);

/* Frees the context kept by hsflow_calc_optical_flow_hs_8u32f (optional; e.g. before unloading). */
void hsflow_release_cached(void);

#ifdef __cplusplus
}
#endif
#endif /* HSFLOW_H_ */
