// Read-ceiling probe for the scan kernel's access pattern: a wave covers a 64-row x 256-B group (16 KiB) as two
// passes of 8 loads; in each load 8 adjacent lanes read one contiguous 128-B half of a row (rows 256 B apart).
// MODE 0: that pattern.  MODE 1: same bytes, fully contiguous 1 KiB per load (16 loads).  MODE 2: like 0 but the
// second half is read only after the first halves of the NEXT group (mimics the kernel's chunk order + prefetch).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(1024) void k(const uint8_t *__restrict__ p, uint64_t n_groups, uint32_t *out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x >> 6);
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t g = wave; g < n_groups; g += nw) {
        const uint8_t *base = p + g * 16384;
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc |= *(const u32x4 *)(base + j * 1024 + lane * 16);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = *(const u32x4 *)(base + (j * 8 + lane / 8) * 256 + h * 128 + (lane % 8) * 16);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc |= v[j];
            }
        }
    }
    uint32_t r = acc[0] | acc[1] | acc[2] | acc[3];
    if (r == 0x12345678u) out[0] = r;
}
extern "C" int launch(const void *p, uint64_t bytes, void *out, int blocks, int threads, int mode, void *stream) {
    const uint64_t ng = bytes / 16384;
    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, (const uint8_t *)p, ng, (uint32_t *)out);
    else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, (const uint8_t *)p, ng, (uint32_t *)out);
    return (int)hipGetLastError();
}
