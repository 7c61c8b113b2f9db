/* C host of the multi-GPU entry points (include/hsflow.h: hsflow_slab_*, hsflow_multi_*) the way a C caller such as the
 * reference's main.cpp would use them: one frame in row slabs, and a batch of independent pairs, each against the plain
 * single-context solve -- bit for bit.  argv[1] = comma list of devices, e.g. "0" or "0,0" (slabs / workers may share
 * a device).  Built and run by tests/test_gpu_multi.py. */
#include "hsflow.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static unsigned long long rng = 88172645463325252ULL;
static unsigned next8(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (unsigned)(rng >> 32) & 0xFFu; }

static void make_pair(unsigned char *a, unsigned char *b, int w, int h, int seed)
{
    int x, y;
    rng = 88172645463325252ULL + (unsigned long long)seed * 7919ULL;
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++) {
            /* a smooth-ish texture and its copy shifted by one pixel, plus a little noise */
            const int base = (int)(128.0 + 60.0 * ((x * 7 + y * 3) % 97) / 97.0 - 30.0 * ((x * 2 + y * 11) % 53) / 53.0);
            a[(size_t)y * w + x] = (unsigned char)((base + (int)(next8() & 7)) & 0xFF);
        }
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++) b[(size_t)y * w + x] = a[(size_t)y * w + (x > 0 ? x - 1 : 0)];
}

/* Equal bit for bit, except where both values lie below 1e-30: the strip kernels carry 4^k u inside a launch, so flow that
 * would be denormal (< 1.2e-38) keeps bits there that depend on where the launches' boundaries fall (chunks of `halo`
 * sweeps against launches of fuse_steps), and such an addend can still decide the rounding of sums up to about 1e-32.
 * Only synthetic flat frames like the early-stop pair get there (flow of interest is > 1e-6). */
static int same_flow(const float *x, const float *y, size_t n)
{
    size_t i;
    for (i = 0; i < n; i++)
        if (memcmp(x + i, y + i, 4) && !(x[i] > -1e-30f && x[i] < 1e-30f && y[i] > -1e-30f && y[i] < 1e-30f)) return 0;
    return 1;
}

static int last_iterations = 0; /* iterations_done of the last single() */

/* p2 != NULL: a second solve on the same context after the first (use_previous continuations) */
static int single2(int dev, int w, int h, const unsigned char *a, const unsigned char *b, const hsflow_params *p, const hsflow_params *p2,
                   float *u, float *v)
{
    hsflow_ctx *c = 0;
    hsflow_info info;
    int st = hsflow_create(&c, dev, w, h, 1, 0, 1);
    if (st) return st;
    info.struct_size = sizeof(info);
    if (!(st = hsflow_set_frames_u8(c, 0, a, (size_t)w, b, (size_t)w)) && !(st = hsflow_solve(c, p)) && !(p2 && (st = hsflow_solve(c, p2))) &&
        !(st = hsflow_get_info(c, &info)))
        st = hsflow_get_flow(c, 0, u, (size_t)w * 4, v, (size_t)w * 4);
    if (st) fprintf(stderr, "single: %s\n", hsflow_last_error(c));
    else last_iterations = info.iterations_done;
    hsflow_destroy(c);
    return st;
}

static int single(int dev, int w, int h, const unsigned char *a, const unsigned char *b, const hsflow_params *p, float *u, float *v)
{
    return single2(dev, w, h, a, b, p, 0, u, v);
}

int main(int argc, char **argv)
{
    int devs[16], nd = 0, st, k, i;
    const int W = 700, H = 263, IT = 57, HALO = 10, NP = 7;
    const size_t px = (size_t)W * H;
    unsigned char *a = (unsigned char *)malloc(px * NP), *b = (unsigned char *)malloc(px * NP);
    float *u0 = (float *)malloc(px * 4), *v0 = (float *)malloc(px * 4), *u = (float *)malloc(px * 4 * NP), *v = (float *)malloc(px * 4 * NP);
    hsflow_params p;
    hsflow_slab *s = 0;
    hsflow_multi *m = 0;
    char *tok = strtok(argc > 1 ? argv[1] : (char *)"0", ",");
    for (; tok && nd < 16; tok = strtok(0, ",")) devs[nd++] = atoi(tok);
    if (!a || !b || !u0 || !v0 || !u || !v) return 90;
    hsflow_default_params(&p);
    p.term_type = HSFLOW_TERM_ITER;
    p.max_iter = IT;
    p.lambda = 0.7f;

    /* one frame in row slabs */
    make_pair(a, b, W, H, 1);
    if ((st = single(devs[0], W, H, a, b, &p, u0, v0))) return 10 + st;
    if ((st = hsflow_slab_create(&s, devs, nd, W, H, HALO))) { fprintf(stderr, "slab_create: %s\n", hsflow_slab_last_error(0)); return 20 + st; }
    if (hsflow_slab_count(s) != nd) return 29;
    for (k = 0; k < nd; k++) {
        int lo, hi;
        if (hsflow_slab_rows(s, k, &lo, &hi) || lo >= hi || (k == 0 && lo != 0) || (k == nd - 1 && hi != H)) return 28;
    }
    if ((st = hsflow_slab_set_frames_u8(s, a, (size_t)W, b, (size_t)W)) || (st = hsflow_slab_solve(s, &p)) ||
        (st = hsflow_slab_get_flow(s, u, (size_t)W * 4, v, (size_t)W * 4))) {
        fprintf(stderr, "slab: %s\n", hsflow_slab_last_error(s));
        return 30 + st;
    }
    if (memcmp(u, u0, px * 4) || memcmp(v, v0, px * 4)) { fprintf(stderr, "slab result differs from the whole-frame solve\n"); return 40; }
    if (hsflow_slab_exchanges(s) != (nd > 1 ? (IT + HALO - 1) / HALO - 1 : 0)) return 41;
    if (hsflow_slab_iterations_done(s) != IT) return 42;
    {   /* the library's defaults as they are -- ITER|EPS, 100 sweeps, eps 1e-6f: how the reference calls the solver
           (OpticalFlowOpenCV.cpp:29) -- must be accepted and end where the one-context solve ends */
        hsflow_params d, w1, w2;
        hsflow_default_params(&d);
        if ((st = single(devs[0], W, H, a, b, &d, u0, v0))) return 10 + st;
        if ((st = hsflow_slab_solve(s, &d)) || (st = hsflow_slab_get_flow(s, u, (size_t)W * 4, v, (size_t)W * 4))) {
            fprintf(stderr, "slab (default params): %s\n", hsflow_slab_last_error(s));
            return 30 + st;
        }
        if (hsflow_slab_iterations_done(s) != last_iterations || memcmp(u, u0, px * 4) || memcmp(v, v0, px * 4)) {
            fprintf(stderr, "slab with the default parameters: %d sweeps against %d, or the flow differs\n", hsflow_slab_iterations_done(s), last_iterations);
            return 43;
        }
        /* a warm start: 30 sweeps, then 27 more from that flow (the halos the first solve left are refreshed first) */
        w1 = d; w1.term_type = HSFLOW_TERM_ITER; w1.max_iter = 30; w1.lambda = 0.7f;
        w2 = w1; w2.max_iter = 27; w2.use_previous = 1; w2.term_type = HSFLOW_TERM_ITER | HSFLOW_TERM_EPS;
        if ((st = single2(devs[0], W, H, a, b, &w1, &w2, u0, v0))) return 10 + st;
        if ((st = hsflow_slab_solve(s, &w1)) || (st = hsflow_slab_solve(s, &w2)) || (st = hsflow_slab_get_flow(s, u, (size_t)W * 4, v, (size_t)W * 4))) {
            fprintf(stderr, "slab (warm start): %s\n", hsflow_slab_last_error(s));
            return 30 + st;
        }
        if (hsflow_slab_iterations_done(s) != last_iterations || memcmp(u, u0, px * 4) || memcmp(v, v0, px * 4)) { fprintf(stderr, "slab warm start differs\n"); return 44; }
    }
    {   /* a pair whose iteration converges inside the budget: a flat frame with one patch one grey level apart.  No slab's
           witness can vouch for the chunk in which Eps drops below epsilon: the solve is replayed, measured sweep by sweep
           (maximum over the slabs on the host) and must stop on the very sweep the one-context solve stops on */
        hsflow_params e;
        int x, y;
        for (y = 0; y < H; y++)
            for (x = 0; x < W; x++) {
                const int in = y >= 60 && y < 200 && x >= 150 && x < 520;
                a[(size_t)y * W + x] = (unsigned char)(in ? 120 : 90);
                b[(size_t)y * W + x] = (unsigned char)(in ? 121 : 90);
            }
        hsflow_default_params(&e);
        e.lambda = 1e-3f; e.max_iter = 400; e.epsilon = 1e-4;
        if ((st = single(devs[0], W, H, a, b, &e, u0, v0))) return 10 + st;
        if (last_iterations <= 1 || last_iterations >= 400) { fprintf(stderr, "the early-stop pair ran %d sweeps\n", last_iterations); return 45; }
        if ((st = hsflow_slab_set_frames_u8(s, a, (size_t)W, b, (size_t)W)) || (st = hsflow_slab_solve(s, &e)) ||
            (st = hsflow_slab_get_flow(s, u, (size_t)W * 4, v, (size_t)W * 4))) {
            fprintf(stderr, "slab (early stop): %s\n", hsflow_slab_last_error(s));
            return 30 + st;
        }
        if (hsflow_slab_iterations_done(s) != last_iterations || !same_flow(u, u0, px) || !same_flow(v, v0, px)) {
            fprintf(stderr, "slab early stop: %d sweeps against %d, or the flow differs\n", hsflow_slab_iterations_done(s), last_iterations);
            return 46;
        }
        if (nd > 1 && !hsflow_slab_eps_measured(s)) return 47;
        printf("slab early stop: sweep %d of %d on %d slab(s)\n", last_iterations, e.max_iter, nd);
    }
    p.term_type = HSFLOW_TERM_EPS;   /* refused: row slabs need a sweep budget */
    if (hsflow_slab_solve(s, &p) != HSFLOW_E_ARG) return 48;
    p.term_type = HSFLOW_TERM_ITER;
    hsflow_slab_destroy(s);
    hsflow_slab_destroy(0);
    if (nd == 1) { /* two sub-slabs per device: one's copies run under the other's sweeps */
        make_pair(a, b, W, H, 1);
        if ((st = single(devs[0], W, H, a, b, &p, u0, v0))) return 10 + st;
        if ((st = hsflow_slab_create_overlapped(&s, devs, nd, W, H, HALO)) || hsflow_slab_count(s) != 2 * nd) return 49;
        if ((st = hsflow_slab_set_frames_u8(s, a, (size_t)W, b, (size_t)W)) || (st = hsflow_slab_solve(s, &p)) ||
            (st = hsflow_slab_get_flow(s, u, (size_t)W * 4, v, (size_t)W * 4)) || memcmp(u, u0, px * 4) || memcmp(v, v0, px * 4)) return 49;
        hsflow_slab_destroy(s);
    }
    printf("slab ok: %d slab(s), %d sweeps, halo %d\n", nd, IT, HALO);

    /* independent pairs, pair i on device i mod nd */
    for (i = 0; i < NP; i++) make_pair(a + px * i, b + px * i, W, H, 10 + i);
    if ((st = hsflow_multi_create(&m, devs, nd, W, H, 2))) { fprintf(stderr, "multi_create: %s\n", hsflow_multi_last_error(0)); return 50 + st; }
    if (hsflow_multi_devices(m) != nd) return 59;
    p.term_type = HSFLOW_TERM_ITER | HSFLOW_TERM_EPS;   /* the reference's own criteria */
    p.epsilon = (double)1e-6f;
    for (i = 0; i < NP; i++) {
        unsigned long long t = 0;
        uint64_t tk = 0;
        st = hsflow_multi_submit(m, HSFLOW_FRAMES_GRAY8, a + px * i, (size_t)W, b + px * i, (size_t)W, u + px * i, (size_t)W * 4, v + px * i,
                                 (size_t)W * 4, &p, &tk);
        t = (unsigned long long)tk;
        if (st || t != (unsigned long long)i) { fprintf(stderr, "multi_submit: %s\n", hsflow_multi_last_error(m)); return 60 + st; }
    }
    if ((st = hsflow_multi_wait(m, 2)) || (st = hsflow_multi_drain(m))) { fprintf(stderr, "multi: %s\n", hsflow_multi_last_error(m)); return 70 + st; }
    if (hsflow_multi_wait(m, 99) != HSFLOW_E_ARG) return 79;
    for (i = 0; i < NP; i++) {
        if ((st = single(devs[i % nd], W, H, a + px * i, b + px * i, &p, u0, v0))) return 80 + st;
        if (memcmp(u + px * i, u0, px * 4) || memcmp(v + px * i, v0, px * 4)) { fprintf(stderr, "pair %d differs\n", i); return 89; }
    }
    hsflow_multi_destroy(m);
    hsflow_multi_destroy(0);
    printf("multi ok: %d pairs over %d worker(s)\n", NP, nd);
    free(a); free(b); free(u0); free(v0); free(u); free(v);
    return 0;
}
