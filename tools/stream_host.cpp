// A C++ host of the device-resident pair pipeline, the way the reference's camera loop would drive it
// (OpticalFlowOpenCV.cpp:91-95: a fresh pair per step, ITER|EPS): time per pair without an interpreter between the calls.
// usage: stream_host W H ITERS DEPTH STEPS [lambda]        build: make -C tools bin/stream_host
#include "../include/hsflow.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char **argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 600, H = argc > 2 ? atoi(argv[2]) : 480, IT = argc > 3 ? atoi(argv[3]) : 100;
    const int depth = argc > 4 ? atoi(argv[4]) : 8, steps = argc > 5 ? atoi(argv[5]) : 2000;
    const float lam = argc > 6 ? (float)atof(argv[6]) : 0.1f;
    std::vector<unsigned char> a((size_t)W * H), b((size_t)W * H);
    for (int k = 0; k < 2; k++) { // two different textured pairs (a smooth pattern and its copy shifted by one pixel)
        (void)k;
    }
    unsigned char *d[4];
    for (int k = 0; k < 2; k++) {
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const double t = 128.0 + 50.0 * std::sin(0.07 * x + 0.3 * k) * std::cos(0.05 * y) + 30.0 * std::sin(0.023 * (x + 2 * y) + k);
                const double t1 = 128.0 + 50.0 * std::sin(0.07 * (x - 1) + 0.3 * k) * std::cos(0.05 * y) + 30.0 * std::sin(0.023 * ((x - 1) + 2 * y) + k);
                a[(size_t)y * W + x] = (unsigned char)std::lround(t);
                b[(size_t)y * W + x] = (unsigned char)std::lround(t1);
            }
        if (hipMalloc((void **)&d[2 * k], a.size()) != hipSuccess || hipMalloc((void **)&d[2 * k + 1], a.size()) != hipSuccess) return 2;
        hipMemcpy(d[2 * k], a.data(), a.size(), hipMemcpyHostToDevice);
        hipMemcpy(d[2 * k + 1], b.data(), b.size(), hipMemcpyHostToDevice);
    }
    hsflow_pipeline *pl = nullptr;
    if (hsflow_pipeline_create(&pl, 0, W, H, depth)) { fprintf(stderr, "create: %s\n", hsflow_pipeline_last_error(nullptr)); return 3; }
    hsflow_params p;
    hsflow_default_params(&p); // ITER|EPS, eps (float)1e-6
    p.lambda = lam;
    p.max_iter = IT;
    p.use_graph = 1;
    auto run = [&](int n) -> int {
        for (int k = 0; k < n; k++) {
            uint64_t t;
            if (hsflow_pipeline_submit_device(pl, d[2 * (k & 1)], (size_t)W, d[2 * (k & 1) + 1], (size_t)W, &p, &t)) return 1;
        }
        return hsflow_pipeline_drain(pl);
    };
    if (run(8 * depth)) { fprintf(stderr, "run: %s\n", hsflow_pipeline_last_error(pl)); return 4; }
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        const auto t0 = std::chrono::steady_clock::now();
        if (run(steps)) { fprintf(stderr, "run: %s\n", hsflow_pipeline_last_error(pl)); return 4; }
        hipDeviceSynchronize();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / steps;
        if (ms < best) best = ms;
    }
    hsflow_info info;
    info.struct_size = sizeof(info);
    uint64_t t;
    hsflow_pipeline_submit_device(pl, d[0], (size_t)W, d[1], (size_t)W, &p, &t);
    hsflow_pipeline_info(pl, t, &info);
    printf("{\"stream_host\": \"C++\", \"width\": %d, \"height\": %d, \"iters\": %d, \"depth\": %d, \"ms_per_pair\": %.5f, \"mpix_iter_per_s\": %.0f, "
           "\"tiles\": %d, \"fuse_steps\": %d, \"rows\": %d, \"threads\": %d, \"iterations_done\": %d, \"eps_rerun\": %d}\n",
           W, H, IT, depth, best, (double)W * H * IT / best / 1e3, info.tiles, info.fuse_steps, info.groups_per_thread, info.threads,
           info.iterations_done, info.eps_rerun);
    hsflow_pipeline_destroy(pl);
    for (auto *q : d) hipFree(q);
    return 0;
}
