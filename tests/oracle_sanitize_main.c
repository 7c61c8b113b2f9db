/* Driver for tests/test_oracle.py::test_oracle_under_sanitizers: every oracle entry point on small and
 * degenerate sizes, exact-size heap buffers, built with -fsanitize=address,undefined. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int hs_oracle_cv_8u32f(const uint8_t *, const uint8_t *, int, int, int, int, float *, float *, int, float, int, int, double, int *, float *);
int hs_oracle_cv_8u32f_mt(const uint8_t *, const uint8_t *, int, int, int, int, float *, float *, int, float, int, int, double, int, int *, float *);
int hs_oracle_cv_derivatives(const uint8_t *, const uint8_t *, int, int, int, float *, float *, float *);
int hs_oracle_classic_ex(const uint8_t *, const uint8_t *, int, int, int, int, float *, float *, int, float, int, int);
int hs_oracle_classic_derivatives(const uint8_t *, const uint8_t *, int, int, int, float *, float *, float *);
int hs_oracle_bgr2gray_u8(const uint8_t *, int, int, int, uint8_t *, int);
int hs_oracle_box_blur3_u8(const uint8_t *, int, int, int, uint8_t *, int);

int main(void)
{
    static const int sizes[][2] = {{1, 1}, {1, 7}, {7, 1}, {2, 2}, {3, 3}, {37, 29}, {64, 5}};
    unsigned seed = 12345u;
    for (unsigned k = 0; k < sizeof sizes / sizeof sizes[0]; k++) {
        const int W = sizes[k][0], H = sizes[k][1], N = W * H;
        uint8_t *A = malloc(N), *B = malloc(N), *bgr = malloc(3 * N), *g = malloc(N), *bl = malloc(N);
        float *u = malloc(N * 4), *v = malloc(N * 4), *x = malloc(N * 4), *y = malloc(N * 4), *t = malloc(N * 4);
        for (int i = 0; i < N; i++) { seed = seed * 1664525u + 1013904223u; A[i] = seed >> 24; seed = seed * 1664525u + 1013904223u; B[i] = seed >> 24; }
        for (int i = 0; i < 3 * N; i++) { seed = seed * 1664525u + 1013904223u; bgr[i] = seed >> 24; }
        int it = 0; float eps = 0;
        if (hs_oracle_cv_8u32f(A, B, W, W, H, 0, u, v, W * 4, 0.5f, 3, 9, 1e-6, &it, &eps)) return 1;
        if (hs_oracle_cv_8u32f(A, B, W, W, H, 1, u, v, W * 4, 0.5f, 1, 3, 0.0, &it, &eps)) return 2;
        if (hs_oracle_cv_8u32f_mt(A, B, W, W, H, 0, u, v, W * 4, 2.0f, 3, 9, 1e-6, 3, &it, &eps)) return 3;
        if (hs_oracle_cv_derivatives(A, B, W, W, H, x, y, t)) return 4;
        if (hs_oracle_classic_ex(A, B, W, W, H, 0, u, v, W * 4, 3.0f, 7, 1)) return 5;
        if (hs_oracle_classic_ex(A, B, W, W, H, 1, u, v, W * 4, 3.0f, 2, 0)) return 6;
        if (hs_oracle_classic_derivatives(A, B, W, W, H, x, y, t)) return 7;
        if (hs_oracle_bgr2gray_u8(bgr, 3 * W, W, H, g, W)) return 8;
        if (hs_oracle_box_blur3_u8(g, W, W, H, bl, W)) return 9;
        free(A); free(B); free(bgr); free(g); free(bl); free(u); free(v); free(x); free(y); free(t);
    }
    puts("SANITIZED-OK");
    return 0;
}
