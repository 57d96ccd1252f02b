// Empirical HBM streaming-read ceiling: every lane loads 16 B (coalesced, plain or nontemporal), ORs it into a register.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void stream_read(const u32x4 *__restrict__ p, uint64_t n16, uint32_t *out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) v[k] = NT ? __builtin_nontemporal_load(p + i + k * stride) : p[i + k * stride];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) acc |= v[k];
    }
    for (; i < n16; i += stride) acc |= p[i];
    uint32_t r = acc[0] | acc[1] | acc[2] | acc[3];
    if (r == 0x12345678u) out[0] = r; // practically never: keeps the loads alive
}
extern "C" int stream_read_launch(const void *p, uint64_t bytes, void *out, int blocks, int unroll, void *stream) {
    const uint64_t n16 = bytes / 16;
    // unroll: 4 | 8 plain loads, 104 | 108 the same with nontemporal (`nt`) loads
    if (unroll == 8) hipLaunchKernelGGL((stream_read<8, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4 *)p, n16, (uint32_t *)out);
    else if (unroll == 4) hipLaunchKernelGGL((stream_read<4, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4 *)p, n16, (uint32_t *)out);
    else if (unroll == 108) hipLaunchKernelGGL((stream_read<8, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4 *)p, n16, (uint32_t *)out);
    else if (unroll == 104) hipLaunchKernelGGL((stream_read<4, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4 *)p, n16, (uint32_t *)out);
    else hipLaunchKernelGGL((stream_read<1, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4 *)p, n16, (uint32_t *)out);
    return (int)hipGetLastError();
}
