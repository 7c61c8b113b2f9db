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

static int single(int dev, int w, int h, const unsigned char *a, const unsigned char *b, const hsflow_params *p, float *u, float *v)
{
    hsflow_ctx *c = 0;
    int st = hsflow_create(&c, dev, w, h, 1, 0, 1);
    if (st) return st;
    if (!(st = hsflow_set_frames_u8(c, 0, a, (size_t)w, b, (size_t)w)) && !(st = hsflow_solve(c, p)))
        st = hsflow_get_flow(c, 0, u, (size_t)w * 4, v, (size_t)w * 4);
    if (st) fprintf(stderr, "single: %s\n", hsflow_last_error(c));
    hsflow_destroy(c);
    return st;
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
    p.term_type = HSFLOW_TERM_ITER | HSFLOW_TERM_EPS;   /* refused: EPS needs a reduction over the slabs */
    if (hsflow_slab_solve(s, &p) != HSFLOW_E_ARG) return 42;
    p.term_type = HSFLOW_TERM_ITER;
    hsflow_slab_destroy(s);
    hsflow_slab_destroy(0);
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
