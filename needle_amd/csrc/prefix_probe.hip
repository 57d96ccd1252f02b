// Measurement aid (libneedle_probe.so, not the product ABI): the GPU analogue of the reference's literal-prefix prefilter
// (`indexOf(prefix)` ahead of the DFA walk: DFAClassBuilder.java:365-376, gated by CompilationPolicy.java:44-57) in the one
// shape where its unit of skipping is contiguous text of ONE row -- few long rows, a wave on a 4 KiB stripe, a lane on 64
// contiguous bytes.  Per stripe: does a window of 2..4 bytes equal to the literal START in it (windows may reach into the
// next stripe)?  Per row: the first such position.  SURVEY.md s8 f-4; the A/B that uses it: scripts/prefix_prefilter_ab.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// rows: n_rows x stride bytes (8-bit code units, stride a multiple of 4096).  lit: the literal's first n_lit (2..4) bytes in
// the low bytes of a dword, mask: 0xFFFF / 0xFFFFFF / 0xFFFFFFFF.  stripe_hit[row * spr + s] = 1 if a window starts in
// stripe s; first[row] = atomicMin of the first position (initialised by the caller to INT_MAX).
__global__ __launch_bounds__(1024) void prefix_scan(const uint8_t *rows, uint64_t n_rows, uint64_t stride, uint32_t lit, uint32_t mask,
                                                    uint8_t *stripe_hit, int32_t *first) {
    const int lane = threadIdx.x & 63;
    const uint64_t spr = stride / 4096;
    const uint64_t total = n_rows * spr;
    const uint64_t waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t v = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; v < total; v += waves) {
        const uint64_t row = v / spr, s = v - row * spr;
        const uint8_t *p = rows + row * stride + s * 4096 + (uint64_t)lane * 64;
        u32x4 d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = __builtin_nontemporal_load((const u32x4 *)(p + 16 * j));
        uint32_t w[17];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w[4 * j] = d[j][0], w[4 * j + 1] = d[j][1], w[4 * j + 2] = d[j][2], w[4 * j + 3] = d[j][3];
        }
        // the dword after this lane's 64 bytes: the next lane's first one; the last lane reads it from memory (0 at the row's end)
        uint32_t nxt = (uint32_t)__shfl_down((int)w[0], 1);
        if (lane == 63) nxt = (s + 1 < spr) ? *(const uint32_t *)(p + 64) : 0u;
        w[16] = nxt;
        bool any = false;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t a = w[j], b = w[j + 1];
            any = any || ((a & mask) == lit) || ((__builtin_amdgcn_alignbyte(b, a, 1) & mask) == lit) ||
                  ((__builtin_amdgcn_alignbyte(b, a, 2) & mask) == lit) || ((__builtin_amdgcn_alignbyte(b, a, 3) & mask) == lit);
        }
        const uint64_t hits = __ballot(any);
        if (hits != 0ull) { // rare for a rare literal: the exact first position inside the first lane that has one
            if (lane == 0) stripe_hit[v] = 1;
            const int fl = __builtin_ctzll(hits);
            if (lane == fl) {
                int pos = -1;
                for (int j = 0; j < 16 && pos < 0; ++j)
                    for (int k = 0; k < 4 && pos < 0; ++k) {
                        const uint32_t win = k ? __builtin_amdgcn_alignbyte(w[j + 1], w[j], k) : w[j];
                        if ((win & mask) == lit) pos = 4 * j + k;
                    }
                atomicMin(&first[row], (int32_t)(s * 4096 + (uint64_t)lane * 64 + (uint64_t)pos));
            }
        } else if (lane == 0) {
            stripe_hit[v] = 0;
        }
    }
}

extern "C" int prefix_scan_launch(const void *rows, uint64_t n_rows, uint64_t stride, uint32_t lit, uint32_t mask, void *stripe_hit, void *first,
                                  int blocks, void *stream) {
    hipLaunchKernelGGL(prefix_scan, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, (const uint8_t *)rows, n_rows, stride, lit, mask,
                       (uint8_t *)stripe_hit, (int32_t *)first);
    return (int)hipGetLastError();
}
