// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the VALU instructions the
// strip kernel is made of, at 1 / 2 / 4 wavefronts per SIMD.  Diagnostic tool, not product code.
// build: hipcc -O3 --offload-arch=gfx950 -o ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, int iters)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 1.0001f, b1 = 0.9999f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b0, b1};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { // v_fma_f32, 8 independent chains x 2
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));)
        } else if (OP == 1) { // v_pk_fma_f32, 4 independent chains x 2
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                               "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));)
        } else if (OP == 2) { // v_pk_add_f32
            REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                               "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));)
        } else if (OP == 3) { // v_pk_mul_f32
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                               "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));)
        } else if (OP == 4) { // v_add_f32 with DPP wave_shr
            REP16(asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %3, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %4, %5, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %6, %7, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %3, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %4, %5, %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %6, %7, %6 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (OP == 5) { // v_add_f32 plain
            REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                               "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
        } else if (OP == 6) { // v_add_f32 with row_shr DPP (within 16 lanes)
            REP16(asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %6, %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %0, %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %3, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %4, %5, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %6, %7, %6 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (OP == 7) { // v_mov_b32
            REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %4, %5\n v_mov_b32 %6, %7\n"
                               "v_mov_b32 %1, %0\n v_mov_b32 %3, %2\n v_mov_b32 %5, %4\n v_mov_b32 %7, %6\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (OP == 8) { // v_pk_fma with op_sel swizzle + neg
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %4 op_sel:[1,0,0] op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %1, %4, %4 neg_lo:[1,0,0] neg_hi:[1,0,0]\n v_pk_fma_f32 %2, %2, %4, %4 op_sel:[1,0,0] op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %3, %4, %4\n"
                               "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    // every wavefront reports; the host takes the slowest one of the block (the SIMD arbiter favours
    // the oldest wavefront, so wavefront 0 alone would under-state the cost)
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (s == 12345.678f) out[blockIdx.x * 16] = 0; // keep the values alive
}

template <int OP>
void run(const char *name, unsigned long long *d)
{
    const int iters = 200;
    const double n = 16.0 * 8.0 * iters; // instructions per wave
    printf("%-28s", name);
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 64 * 4 * waves_per_simd;
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, d, iters);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, d, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 16);
        hipMemcpy(h.data(), d, 256 * 16 * 8, hipMemcpyDeviceToHost);
        double s = 0;
        const int nw = threads / 64;
        for (int b = 0; b < 256; b++) {
            unsigned long long m = 0;
            for (int w = 0; w < nw; w++) m = std::max(m, h[b * 16 + w]);
            s += (double)m;
        }
        s /= 256;
        // cycles per instruction per SIMD = elapsed / (instructions issued on that SIMD)
        printf("  %dw/SIMD: %.2f cyc/instr", waves_per_simd, s / (n * waves_per_simd));
    }
    printf("\n");
}

int main()
{
    unsigned long long *d;
    hipMalloc(&d, 256 * 16 * 8);
    run<5>("v_add_f32", d);
    run<0>("v_fma_f32", d);
    run<7>("v_mov_b32", d);
    run<2>("v_pk_add_f32", d);
    run<3>("v_pk_mul_f32", d);
    run<1>("v_pk_fma_f32", d);
    run<8>("v_pk_fma_f32 op_sel/neg", d);
    run<4>("v_add_f32_dpp wave_shr/shl", d);
    run<6>("v_add_f32_dpp row_shr/shl", d);
    hipFree(d);
    return 0;
}
