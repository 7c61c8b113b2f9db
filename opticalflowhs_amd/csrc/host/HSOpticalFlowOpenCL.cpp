// HSOpticalFlowOpenCL.cpp -- see HSOpticalFlowOpenCL.hpp.  Host orchestration only; every pixel is
// computed by libhsflow.so.
#include "HSOpticalFlowOpenCL.hpp"

#include <chrono>
#include <cmath>
#include <cstring>
#include <iostream>

namespace {

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Frames of the "-cam" route come from numbered files instead of a capture device:
// $HSFLOW_CAMERA_DIR/frame_0000.pgm (or .ppm / .jpg), frame_0001.pgm, ... until one is missing.
std::string camera_frame(int i)
{
    const char *dir = getenv("HSFLOW_CAMERA_DIR");
    const std::string base = std::string(dir ? dir : ".");
    char name[64];
    for (const char *ext : {"pgm", "ppm", "jpg"}) {
        snprintf(name, sizeof(name), "/frame_%04d.%s", i, ext);
        if (FILE *f = fopen((base + name).c_str(), "rb")) { fclose(f); return base + name; }
    }
    snprintf(name, sizeof(name), "/frame_%04d.pgm", i);
    return base + name;
}

} // namespace

HSOpticalFlowOpenCL::HSOpticalFlowOpenCL(const char *name, char *src_, char *in1, char *in2, char *out, float alp,
                                         int it, int gs, char *dType)
    : SDKSample(name), alpha(alp), iterations(it), blockSizeX(gs), src(src_ ? src_ : ""), input1(in1 ? in1 : ""),
      input2(in2 ? in2 : ""), output(out ? out : "")
{
    gpu = !(dType && strcmp(dType, "CPU") == 0); // HSOpticalFlowOpenCL.hpp:157-160
}

HSOpticalFlowOpenCL::HSOpticalFlowOpenCL(const char *name, char *src_, float alp, int it, int gs, char *dType)
    : SDKSample(name), alpha(alp), iterations(it), blockSizeX(gs), src(src_ ? src_ : "")
{
    gpu = !(dType && strcmp(dType, "CPU") == 0);
}

HSOpticalFlowOpenCL::~HSOpticalFlowOpenCL() { cleanup(); }

int HSOpticalFlowOpenCL::initialize() { return SDKSample::initialize(); }
int HSOpticalFlowOpenCL::setup() { return SDK_SUCCESS; }
int HSOpticalFlowOpenCL::verifyResults() { return SDK_SUCCESS; }

int HSOpticalFlowOpenCL::cleanup()
{
    if (ctx) { hsflow_destroy(ctx); ctx = nullptr; }
    return SDK_SUCCESS;
}

int HSOpticalFlowOpenCL::ensureContext(int w, int h)
{
    if (ctx && (unsigned)w == width && (unsigned)h == height) return SDK_SUCCESS;
    cleanup();
    if (hsflow_create(&ctx, 0, w, h, 1, nullptr, 1) != HSFLOW_OK) {
        std::cout << "hsflow_create: " << hsflow_last_error(nullptr) << std::endl;
        ctx = nullptr;
        return SDK_FAILURE;
    }
    width = w; height = h;
    u.assign((size_t)w * h, 0.f);
    v.assign((size_t)w * h, 0.f);
    return SDK_SUCCESS;
}

int HSOpticalFlowOpenCL::solvePair(const pnm::Image &a, const pnm::Image &b, bool streaming)
{
    int st = streaming ? hsflow_push_frame_u8(ctx, 0, b.data.data(), b.width)
                       : hsflow_set_frames_u8(ctx, 0, a.data.data(), a.width, b.data.data(), b.width);
    if (st != HSFLOW_OK) { std::cout << hsflow_last_error(ctx) << std::endl; return SDK_FAILURE; }
    hsflow_params p;
    hsflow_default_params(&p);
    // the "-cl" route keeps the reference kernels' own discretisation (Kernels.cl: cube derivatives,
    // 1/6-1/12 mean, alpha^2) -- with the v update the reference forgot; HSFLOW_CL_AS_CV=1 switches
    // to the OpenCV discretisation with the equivalent regulariser lambda = 1/alpha^2,
    // HSFLOW_CL_AS_SHIPPED=1 to Kernels.cl as shipped (v never written: the reference's pictures)
    if (getenv("HSFLOW_CL_AS_CV")) p.lambda = 1.0f / (alpha * alpha);
    else { p.mode = getenv("HSFLOW_CL_AS_SHIPPED") ? HSFLOW_MODE_CLASSIC_AS_SHIPPED : HSFLOW_MODE_CLASSIC; p.alpha = alpha; }
    p.term_type = HSFLOW_TERM_ITER;        // the reference loop runs a fixed count (:750-751)
    p.max_iter = iterations;
    const double t0 = now_ms();
    st = hsflow_solve(ctx, &p);
    if (st == HSFLOW_OK) st = hsflow_get_flow(ctx, 0, u.data(), (size_t)width * 4, v.data(), (size_t)width * 4);
    lastMs = now_ms() - t0;
    if (st != HSFLOW_OK) { std::cout << hsflow_last_error(ctx) << std::endl; return SDK_FAILURE; }
    return SDK_SUCCESS;
}

// Arrow rendering of the reference (HSOpticalFlowOpenCL.cpp:762-770): 4-pixel grid, |u| or |v| > 0.5,
// blue dot + red full-length line.
void HSOpticalFlowOpenCL::drawFlow(pnm::Image &imgFlow) const
{
    imgFlow.width = width; imgFlow.height = height; imgFlow.channels = 3;
    imgFlow.data.assign((size_t)width * height * 3, 0);
    const int step = 4;
    for (unsigned i = 0; i < height; i += step)
        for (unsigned j = 0; j < width; j += step) {
            const float fu = u[j + (size_t)i * width], fv = v[j + (size_t)i * width];
            if (fu > 0.5f || fv > 0.5f || fu < -0.5f || fv < -0.5f) {
                pnm::filled_circle(imgFlow, j, i, 2, 0, 0, 255);
                pnm::line(imgFlow, j, i, (int)(j + fu), (int)(i + fv), 255, 0, 0);
            }
        }
}

int HSOpticalFlowOpenCL::run()
{
    if (!gpu) {
        std::cout << "dType CPU is not available in this build (GPU only)." << std::endl;
        return SDK_FAILURE;
    }
    if (!(alpha > 0.f) || iterations <= 0) {
        std::cout << "alpha and the iteration count must be positive." << std::endl;
        return SDK_FAILURE;
    }
    if (src == "-hd") {
        pnm::Image c1, c2, g1, g2;
        if (!pnm::load_image(input1, c1) || !pnm::load_image(input2, c2)) {
            std::cout << "Input image error.\n";
            return -1; // HSOpticalFlowOpenCL.cpp:724,735
        }
        pnm::to_gray(c1, g1);
        pnm::to_gray(c2, g2);
        if (g1.width != g2.width || g1.height != g2.height) { std::cout << "Input image error.\n"; return -1; }
        if (ensureContext(g1.width, g1.height) != SDK_SUCCESS) return SDK_FAILURE;
        if (solvePair(g1, g2, false) != SDK_SUCCESS) return SDK_FAILURE;
        std::cout << "Avg time: " << lastMs << " [ms]" << std::endl; // :755
        pnm::Image imgFlow;
        drawFlow(imgFlow);
        if (!output.empty() && !pnm::save_image(output, imgFlow)) return SDK_FAILURE;
        return 0;
    }
    // "-cam": previous frame stays on the device, only the new frame is uploaded (:810-834)
    pnm::Image prev, cur, gprev, gcur;
    if (!pnm::load_image(camera_frame(0), prev)) { std::cerr << "ERROR: capture is NULL \n"; return -1; }
    pnm::to_gray(prev, gprev);
    if (ensureContext(gprev.width, gprev.height) != SDK_SUCCESS) return SDK_FAILURE;
    double total = 0.0;
    int count = 0;
    for (int i = 1; pnm::load_image(camera_frame(i), cur); i++) {
        pnm::to_gray(cur, gcur);
        if (gcur.width != gprev.width || gcur.height != gprev.height) break;
        if (solvePair(gprev, gcur, count > 0) != SDK_SUCCESS) return SDK_FAILURE;
        total += lastMs;
        count++;
        if (getenv("HSFLOW_CAMERA_OUT")) {
            pnm::Image imgFlow;
            drawFlow(imgFlow);
            char name[64];
            snprintf(name, sizeof(name), "/flow_%04d.ppm", i);
            pnm::save_image(std::string(getenv("HSFLOW_CAMERA_OUT")) + name, imgFlow);
        }
        gprev = gcur;
    }
    if (count) std::cout << "Avg time: " << total / count << " [ms]" << std::endl; // :838
    return SDK_SUCCESS;
}

// ---- GPU counterpart of OpticalFlowOpenCV (OpticalFlowHS/OpticalFlowOpenCV.cpp:7-52) --------------

// arrows of the CPU route: 4-pixel grid, |.| > 1, half length (OpticalFlowOpenCV.cpp:33-46, :98-111)
static void draw_cv_flow(pnm::Image &imgFlow, const std::vector<float> &u, const std::vector<float> &v, int W, int H)
{
    imgFlow.width = W; imgFlow.height = H; imgFlow.channels = 3;
    imgFlow.data.assign((size_t)W * H * 3, 0);
    for (int y = 0; y < H; y += 4)
        for (int x = 0; x < W; x += 4) {
            const float px = u[(size_t)y * W + x], py = v[(size_t)y * W + x];
            if (px > 1 || py > 1 || px < -1 || py < -1) {
                pnm::filled_circle(imgFlow, x, y, 2, 0, 0, 255);
                pnm::line(imgFlow, x, y, (int)(x + px / 2), (int)(y + py / 2), 255, 0, 0);
            }
        }
}

int OpticalFlowOpenCV::runFromImg(char *input1, char *input2, char *output, float lambda, int it)
{
    pnm::Image c1, c2;
    if (!pnm::load_image(input1, c1) || !pnm::load_image(input2, c2) || c1.width != c2.width || c1.height != c2.height) {
        std::cout << "Input image error.\n";
        return -1;
    }
    const int W = c1.width, H = c1.height;
    hsflow_ctx *ctx = nullptr;
    if (hsflow_create(&ctx, 0, W, H, 1, nullptr, 1) != HSFLOW_OK) { std::cout << hsflow_last_error(nullptr) << std::endl; return 1; }
    int st;
    const double t0 = now_ms();
    if (c1.channels == 3) { // PPM is RGB; the C ABI takes BGR like cvLoadImage delivers
        std::vector<uint8_t> b1(c1.data), b2(c2.data);
        for (size_t i = 0; i < b1.size(); i += 3) { std::swap(b1[i], b1[i + 2]); std::swap(b2[i], b2[i + 2]); }
        st = hsflow_set_frames_bgr8(ctx, 0, b1.data(), (size_t)W * 3, b2.data(), (size_t)W * 3, 1); // gray + cvSmooth(CV_BLUR,3,3)
    } else {
        st = hsflow_set_frames_gray8_blur(ctx, 0, c1.data.data(), (size_t)W, c2.data.data(), (size_t)W);
    }
    hsflow_params p;
    hsflow_default_params(&p);          // ITER|EPS, eps = (float)1e-6 as at OpticalFlowOpenCV.cpp:29
    p.lambda = lambda;
    p.max_iter = it;
    std::vector<float> u((size_t)W * H), v((size_t)W * H);
    if (st == HSFLOW_OK) st = hsflow_solve(ctx, &p);
    if (st == HSFLOW_OK) st = hsflow_get_flow(ctx, 0, u.data(), (size_t)W * 4, v.data(), (size_t)W * 4);
    const double ms = now_ms() - t0;
    if (st != HSFLOW_OK) { std::cout << hsflow_last_error(ctx) << std::endl; hsflow_destroy(ctx); return 1; }
    hsflow_destroy(ctx);
    pnm::Image imgFlow;
    draw_cv_flow(imgFlow, u, v, W, H);
    pnm::save_image(output, imgFlow);
    std::cout << "Avg time: " << ms << " [ms]" << std::endl;
    return 0;
}

// The camera loop of the CPU route (OpticalFlowOpenCV.cpp:56-131) on numbered frame files instead of a
// capture device ($HSFLOW_CAMERA_DIR, see camera_frame).  Faithful to the reference's loop, including
// its quirk: cvSmooth works in place and the blurred new frame becomes the next old frame (:93,:117),
// so from the second pair on the old frame enters the solver blurred TWICE.  The blurred frames never
// leave the device: the new one is read back only to be handed in again as the next old one.
int OpticalFlowOpenCV::runFromCamera(float lambda, int it)
{
    pnm::Image frame, gold, gnew;
    if (!pnm::load_image(camera_frame(0), frame)) { std::cout << "ERROR: capture is NULL \n"; return -1; }
    pnm::to_gray(frame, gold);                                   // cvCvtColor(imgTmp, imgOld, CV_BGR2GRAY) :79
    const int W = gold.width, H = gold.height;
    hsflow_ctx *ctx = nullptr;
    if (hsflow_create(&ctx, 0, W, H, 1, nullptr, 1) != HSFLOW_OK) { std::cout << hsflow_last_error(nullptr) << std::endl; return 1; }
    hsflow_params p;
    hsflow_default_params(&p);                                   // ITER|EPS, eps (float)1e-6 :94
    p.lambda = lambda;
    p.max_iter = it;
    std::vector<float> u((size_t)W * H), v((size_t)W * H);
    std::vector<uint8_t> scratch((size_t)W * H);
    double total = 0.0;
    int count = 0;
    for (int i = 1; pnm::load_image(camera_frame(i), frame); i++) {
        pnm::to_gray(frame, gnew);
        if (gnew.width != W || gnew.height != H) break;
        const double t0 = now_ms();
        int st = hsflow_set_frames_gray8_blur(ctx, 0, gold.data.data(), (size_t)W, gnew.data.data(), (size_t)W); // :92-93
        if (st == HSFLOW_OK) st = hsflow_solve(ctx, &p);
        if (st == HSFLOW_OK) st = hsflow_get_flow(ctx, 0, u.data(), (size_t)W * 4, v.data(), (size_t)W * 4);
        total += now_ms() - t0;
        // imgOld = imgNew (:117): the blurred new frame
        if (st == HSFLOW_OK) st = hsflow_get_frames_u8(ctx, 0, scratch.data(), (size_t)W, gold.data.data(), (size_t)W);
        if (st != HSFLOW_OK) { std::cout << hsflow_last_error(ctx) << std::endl; hsflow_destroy(ctx); return 1; }
        count++;
        if (getenv("HSFLOW_CAMERA_OUT")) {
            pnm::Image imgFlow;
            draw_cv_flow(imgFlow, u, v, W, H);
            char name[64];
            snprintf(name, sizeof(name), "/flow_%04d.ppm", i);
            pnm::save_image(std::string(getenv("HSFLOW_CAMERA_OUT")) + name, imgFlow);
        }
    }
    hsflow_destroy(ctx);
    if (count) std::cout << "Avg time: " << total / count << " [ms]" << std::endl; // :122 (per frame)
    return 0;
}
