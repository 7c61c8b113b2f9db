// Cost of one s_barrier per loop trip for a workgroup of N wavefronts (one workgroup per CU), with a little VALU work
// and optionally an LDS write + read per trip -- the skeleton of the strip kernel's sweep loop.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/ubench_barrier tools/ubench_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE> // 0: barrier only; 1: + 16 dependent-free v_pk_add; 2: + LDS write (4 x b128) + read (4 x b128)
__global__ __launch_bounds__(1024) void k(unsigned long long *out, float *sink, int iters)
{
    extern __shared__ float4 lds[];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8];
    for (int i = 0; i < 8; i++) a[i] = f2{(float)threadIdx.x + i, 1.0f};
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < iters; s++) {
        if (MODE >= 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) { a[i] = a[i] + a[(i + 1) & 7]; a[i] = a[i] + f2{1.0f, 2.0f}; }
        }
        if (MODE >= 2) {
            float4 *wr = lds + ((size_t)((s & 1) * nw + w) * 4) * 64 + lane;
            wr[0] = make_float4(a[0].x, a[0].y, a[1].x, a[1].y); wr[64] = make_float4(a[2].x, a[2].y, a[3].x, a[3].y);
            wr[128] = make_float4(a[4].x, a[4].y, a[5].x, a[5].y); wr[192] = make_float4(a[6].x, a[6].y, a[7].x, a[7].y);
        }
        __syncthreads();
        if (MODE >= 2) {
            const float4 *rd = lds + ((size_t)((s & 1) * nw + ((w + 1) % nw)) * 4) * 64 + lane;
            const float4 x = rd[0], y = rd[64], z = rd[128], q = rd[192];
            a[0].x += x.x; a[1].x += y.y; a[2].x += z.z; a[3].x += q.w;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
    for (int i = 0; i < 8; i++) acc += a[i].x + a[i].y;
    if (acc == 12345.678f) sink[0] = acc;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(int nthreads, const char *name)
{
    unsigned long long *d; float *sink;
    hipMalloc(&d, 256 * 8); hipMalloc(&sink, 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 2000;
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(nthreads), 131072 + 1024, 0, d, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto x : h) sum += (double)x;
    printf("%-40s %4d threads: %7.1f cycles per trip\n", name, nthreads, sum / 256 / iters);
    hipFree(d); hipFree(sink);
}
int main()
{
    for (int nt : {256, 512, 1024}) {
        run<0>(nt, "s_barrier only");
        run<1>(nt, "16 v_pk_add + s_barrier");
        run<2>(nt, "16 v_pk_add + LDS 4w/4r b128 + s_barrier");
    }
    return 0;
}
