// Measurement aid (libneedle_probe.so, not the product ABI): the n-gram candidate filter of needle_ngram.h ALONE over a
// contiguous byte stream -- how fast can the chip test one hashed 4-byte window every S chars against an LDS bitmap?  (The
// product kernel, needle_ngram.hip, adds the candidate queue and the automaton runs.)  Counts the windows that pass.
#include "needle_ngram.h"
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) unsigned char ng_smem[];

template <int S>
__global__ __launch_bounds__(1024) void ngram_filter_probe(const uint8_t *text, uint64_t n_units /* 1 KiB each */, const uint32_t *bitmap,
                                                           needle::NgramParams np, unsigned long long *n_pass) {
    const int lane = threadIdx.x & 63;
    for (uint32_t i = threadIdx.x * 16u; i < np.bm_bytes; i += blockDim.x * 16u) *(u32x4 *)(ng_smem + i) = *(const u32x4 *)((const uint8_t *)bitmap + i);
    __syncthreads();
    const uint64_t waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t wv = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (uint64_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // a wave takes 16 consecutive units (one 64-row group of 256-byte rows) at a time; n_units is a multiple of 16
    constexpr int PF = 4;
    uint32_t count = 0, carry = 0;
    u32x4 r[PF];
    const uint64_t n_groups = n_units >> 4;
    // (unconditional loads at clamped addresses: a load under a branch makes the compiler drain vmcnt at the join)
    auto addr = [&](uint64_t grp, int j) {
        const uint64_t unit = (grp < n_groups ? grp : n_groups - 1) * 16 + (uint64_t)j;
        return (const u32x4 *)(text + (unit << 10) + (uint64_t)lane * 16u);
    };
    uint64_t g = wv;
    if (g >= n_groups) return;
#pragma unroll
    for (int k = 0; k < PF; ++k) r[k] = __builtin_nontemporal_load(addr(g, k));
    for (; g < n_groups; g += waves) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const u32x4 v = r[j % PF];
            r[j % PF] = __builtin_nontemporal_load(j + PF < 16 ? addr(g, j + PF) : addr(g + waves, j + PF - 16));
            const uint32_t pw = needle::ngram_prev_dword(v[3], carry);
            carry = (uint32_t)__builtin_amdgcn_readlane((int)v[3], 63);
            const uint32_t log = needle::ngram_piece<S>(0u, pw, v[0], v[1], v[2], v[3], np.m1 | np.m2 << 16, np.addr_mask, 0u);
            count += (uint32_t)__builtin_popcount(log);
        }
    }
    for (int o = 32; o > 0; o >>= 1) count += (uint32_t)__shfl_xor((int)count, o);
    if (lane == 0 && count) atomicAdd(n_pass, (unsigned long long)count);
}

extern "C" int ngram_filter_probe_launch(const void *text, uint64_t n_units, const void *bitmap, const needle::NgramParams *np, void *n_pass,
                                         int blocks, void *stream) {
    auto launch = [&](auto kern) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), np->bm_bytes, (hipStream_t)stream, (const uint8_t *)text, n_units, (const uint32_t *)bitmap, *np,
                           (unsigned long long *)n_pass);
    };
    if (np->stride == 4) launch(ngram_filter_probe<4>);
    else if (np->stride == 2) launch(ngram_filter_probe<2>);
    else launch(ngram_filter_probe<1>);
    return (int)hipGetLastError();
}
