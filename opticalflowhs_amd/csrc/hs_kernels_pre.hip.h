// hs_kernels_pre.hip.h -- pre-processing of the reference's CPU route, on the GPU:
//   cvCvtColor(img, gray, CV_BGR2GRAY)        OpticalFlowHS/OpticalFlowOpenCV.cpp:17,20
//   cvSmooth(img, img, CV_BLUR, 3, 3, 0, 0)   OpticalFlowHS/OpticalFlowOpenCV.cpp:27-28
// Byte work, HBM-bound: each lane handles 4 consecutive pixels (one 32-bit store); the 3x3
// neighbourhood of the blur comes from L1/L2.  Arithmetic = oracle/hs_preproc_oracle.c, bit exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hsk {

// gray = (1868 B + 9617 G + 4899 R + 8192) >> 14 on interleaved BGR bytes (row stride in bytes)
__global__ __launch_bounds__(256) void k_bgr2gray(const uint8_t *__restrict__ bgr, long long bgr_stride,
                                                  uint8_t *__restrict__ gray, int W, int H, int P)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= W || y >= H) return;
    const uint8_t *s = bgr + (long long)y * bgr_stride + 3 * x0;
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (x0 + k < W) {
            const uint32_t g = (1868u * s[3 * k] + 9617u * s[3 * k + 1] + 4899u * s[3 * k + 2] + 8192u) >> 14;
            out |= g << (8 * k);
        }
    }
    *(uint32_t *)(gray + (long long)y * P + x0) = out; // row pitch P is a multiple of 64: in bounds
}

// 3x3 box blur, replicate border, round(sum / 9): (2 s + 9) / 18 is exact because s / 9 is never
// half-way between two integers.  src and dst are distinct planes of pitch P.
__global__ __launch_bounds__(256) void k_box_blur3(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                   int W, int H, int P)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= W || y >= H) return;
    const uint8_t *r0 = src + (long long)(y > 0 ? y - 1 : 0) * P;
    const uint8_t *r1 = src + (long long)y * P;
    const uint8_t *r2 = src + (long long)(y < H - 1 ? y + 1 : H - 1) * P;
    int col[6]; // column sums of columns x0-1 .. x0+4 (clamped)
#pragma unroll
    for (int k = 0; k < 6; k++) {
        int xc = x0 + k - 1;
        xc = xc < 0 ? 0 : (xc > W - 1 ? W - 1 : xc);
        col[k] = r0[xc] + r1[xc] + r2[xc];
    }
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int s = col[k] + col[k + 1] + col[k + 2];
        out |= (uint32_t)((2 * s + 9) / 18) << (8 * k);
    }
    *(uint32_t *)(dst + (long long)y * P + x0) = out;
}

// Both frames of a pair from device memory into the context's planes in ONE launch (hsflow_set_frames_u8_device: a
// stream of resident pairs pays one copy launch per pair instead of two 2-D copies).  One lane = 16 bytes of one row
// of either frame (blockIdx.z: which frame); ALIGNED: both sources are 16-byte aligned with strides that are multiples
// of 16, so whole groups move as one 128-bit access; the ragged end of a row (W % 16) goes byte by byte.
template <bool ALIGNED>
__global__ __launch_bounds__(256) void k_copy_pair_u8(const uint8_t *__restrict__ srcA, long long strideA,
                                                      const uint8_t *__restrict__ srcB, long long strideB,
                                                      uint8_t *__restrict__ dstA, uint8_t *__restrict__ dstB, int W, int H, int P)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 16;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= W || y >= H) return;
    const uint8_t *s = (blockIdx.z ? srcB + (long long)y * strideB : srcA + (long long)y * strideA) + x0;
    uint8_t *d = (blockIdx.z ? dstB : dstA) + (long long)y * P + x0;
    if (ALIGNED && x0 + 16 <= W) {
        *(uint4 *)d = *(const uint4 *)s;
    } else {
        const int n = W - x0 < 16 ? W - x0 : 16;
        for (int k = 0; k < n; k++) d[k] = s[k];
    }
}

} // namespace hsk
