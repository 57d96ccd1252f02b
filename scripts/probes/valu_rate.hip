// Integer VALU issue rates on gfx950, measured per SIMD: W waves per SIMD each run `iters` x 32 independent copies of ONE instruction;
// cycles (s_memtime) per wave-instruction per SIMD = elapsed / (W * iters * 32).  What the filter / walk kernels' "VALU busy" is priced with.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate scripts/probes/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define OP32(STR)                                                                                                                       \
    _Pragma("unroll") for (int j = 0; j < 32; ++j) asm volatile(STR : "+v"(r[j]) : "v"(x), "v"(y));

template <int KIND>
__global__ __launch_bounds__(1024) void rate_kernel(uint32_t iters, uint32_t seed, uint64_t *cycles, uint32_t *sink) {
    uint32_t r[32];
    const uint32_t x = seed * 2654435761u + threadIdx.x, y = (seed ^ 0x9E3779B9u) + threadIdx.x * 40503u;
#pragma unroll
    for (int j = 0; j < 32; ++j) r[j] = x + j;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < iters; ++i) {
        if (KIND == 0) { OP32("v_add_u32 %0, %0, %1") }
        if (KIND == 1) { OP32("v_and_b32 %0, %0, %1") }
        if (KIND == 2) { OP32("v_dot2_u32_u16 %0, %1, %2, %0") }
        if (KIND == 3) { OP32("v_and_or_b32 %0, %0, %1, %2") }
        if (KIND == 4) { OP32("v_alignbit_b32 %0, %1, %0, 1") }
        if (KIND == 5) { OP32("v_lshrrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD") }
        if (KIND == 6) { OP32("v_mad_u32_u24 %0, %0, %1, %2") }
        if (KIND == 7) { OP32("v_perm_b32 %0, %0, %1, %2") }
        if (KIND == 8) { OP32("v_lshl_add_u32 %0, %0, 2, %1") }
        if (KIND == 9) { OP32("v_med3_u32 %0, %0, %1, %2") }
        if (KIND == 10) { OP32("v_mul_lo_u32 %0, %0, %1") }
        if (KIND == 11) { OP32("v_cndmask_b32 %0, %0, %1, vcc") }
        if (KIND == 12) { OP32("v_bfe_u32 %0, %0, %1, 5") }
        if (KIND == 13) { OP32("v_lshrrev_b32 %0, 4, %0") }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) acc ^= r[j];
    if (acc == 0x12345u) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) cycles[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(const char *name, int waves_per_simd) {
    const int blocks = 256, threads = 64 * 4 * waves_per_simd; // one workgroup per CU
    const uint32_t iters = 2000;
    uint64_t *d_c;
    uint32_t *d_s;
    hipMalloc(&d_c, sizeof(uint64_t) * blocks * 16);
    hipMalloc(&d_s, 64);
    rate_kernel<KIND><<<blocks, threads>>>(iters, 1, d_c, d_s);
    rate_kernel<KIND><<<blocks, threads>>>(iters, 2, d_c, d_s);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(blocks * (threads / 64));
    hipMemcpy(h.data(), d_c, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (uint64_t v : h) mean += (double)v;
    mean /= h.size();
    printf("%-22s %d waves/SIMD: %.2f cycles per wave-instruction per SIMD (wave lifetime %.0f cycles)\n", name, waves_per_simd,
           mean / ((double)waves_per_simd * iters * 32), mean);
    hipFree(d_c);
    hipFree(d_s);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_add_u32", w);
        run<1>("v_and_b32", w);
        run<2>("v_dot2_u32_u16", w);
        run<3>("v_and_or_b32", w);
        run<4>("v_alignbit_b32", w);
        run<5>("v_lshrrev_b32_sdwa", w);
        run<6>("v_mad_u32_u24", w);
        run<7>("v_perm_b32", w);
        run<8>("v_lshl_add_u32", w);
        run<9>("v_med3_u32", w);
        run<10>("v_mul_lo_u32", w);
        run<11>("v_cndmask_b32", w);
        run<12>("v_bfe_u32", w);
        run<13>("v_lshrrev_b32", w);
    }
    return 0;
}
