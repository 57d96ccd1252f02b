// Hand-written gfx950 (CDNA4) kernels for needle's DFA table-walk hot path.
//
// One haystack ("row") per lane, 64 rows per wavefront step, 16 wavefronts per workgroup sharing one
// LDS copy of the lowered automaton.  Per step a wave stages a 64-row x 128-byte tile with eight
// `global_load_lds_dwordx4` (HBM -> LDS DMA, 16 B/lane, each 8 lanes covering one full 128-B line of one
// row => fully coalesced), with the SOURCE address XOR-swizzled so that the following per-lane row reads
// (ds_read_b128, row stride 128 B) are bank-conflict free.  Latency is hidden by occupancy: 16 waves x
// 8 KiB tiles per CU in flight, no intra-wave software pipeline (hipcc drains an LDS-DMA with vmcnt(0)
// before any may-alias ds_read anyway).
//
// The loops restated here (reference: needle-compiler/src/main/java/com/justinblank/strings/
// DFAClassBuilder.java): matches() :892-910, containedIn() :1004-1022, indexForwards() :438-468,
// indexBackwards() :565-583, find() :629-657.  Dead state (-1), the `c > maxChar` exits and "index
// past the row length" are folded into the lowered tables on the host (needle_lower.cpp): sink state 0,
// OVER and PAD columns -- so the inner loops here are branch-free lookups.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include "needle_device.h"

namespace needle {

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

__device__ __forceinline__ void stage_tile(const uint8_t *rows, uint64_t row0, uint64_t n_rows, uint64_t stride_bytes,
                                           uint64_t total_bytes, uint32_t byte_off, unsigned char *buf, int lane) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = j * 8 + (lane >> 3);          // tile row this lane's 16 B land in
        const int kk = (lane & 7) ^ ((r >> 1) & 7); // which 16-B piece of the row's 128-B chunk goes there
        uint64_t row = row0 + (uint64_t)r;
        if (row >= n_rows) row = n_rows - 1;
        uint64_t off = row * stride_bytes + byte_off + (uint32_t)(kk * 16);
        if (off > total_bytes - 16) off = total_bytes - 16;
        __builtin_amdgcn_global_load_lds((gvoid_t *)(rows + off), (lvoid_t *)(buf + j * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ uint4 tile_piece(const unsigned char *buf, int lane, int kk) {
    return *(const uint4 *)(buf + lane * kChunkBytes + ((kk ^ ((lane >> 1) & 7)) << 4));
}

__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t t = (uint32_t)__shfl_xor((int)v, o);
        v = v > t ? v : t;
    }
    return v;
}

// Pointers into the LDS (or global) copy of a lowered automaton.
struct ProgView {
    const uint32_t *f;
    const uint8_t *cmap;
    const uint8_t *ptab;
    const uint8_t *pages;
    const uint8_t *t8;
    const uint16_t *t16;
    uint32_t n_cols, pad_col, accept_lo;
};

template <int CW>
__device__ __forceinline__ uint32_t column_of(const ProgView &pv, uint32_t c) {
    if (CW == 1) return pv.cmap[c];
    return pv.pages[((uint32_t)pv.ptab[c >> 8] << 8) | (c & 255u)];
}

// One transition.  `st` is 4*state in MODE_NIBBLE, the state id otherwise.
template <int MODE, int CW, bool GUARD>
__device__ __forceinline__ uint32_t step(const ProgView &pv, uint32_t st, uint32_t c, bool in_row) {
    if (MODE == MODE_NIBBLE) {
        uint32_t F;
        if (CW == 1) {
            uint32_t i = GUARD ? (in_row ? c : 256u) : c;
            F = pv.f[i];
        } else {
            uint32_t col = column_of<2>(pv, c);
            if (GUARD) col = in_row ? col : pv.pad_col;
            F = pv.f[col];
        }
        return __builtin_amdgcn_ubfe(F, st, 4) << 2;
    } else {
        uint32_t col = column_of<CW>(pv, c);
        if (GUARD) col = in_row ? col : pv.pad_col;
        const uint32_t i = st * pv.n_cols + col;
        if (MODE == MODE_TABLE8) return pv.t8[i];
        return pv.t16[i];
    }
}

template <int OP, int CW, int MODE, bool GUARD>
__global__ __launch_bounds__(kWavesPerBlock * 64) void scan_kernel(const ScanArgs a) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // ---- stage the automaton in LDS (once per workgroup)
    const int n_waves = blockDim.x >> 6; // 16, 8 or 4: chosen by the launcher from the automaton's LDS footprint
    for (uint32_t i = tid * 16u; i < a.hdr.lds_bytes; i += blockDim.x * 16u)
        *(uint4 *)(smem + i) = *(const uint4 *)(a.prog + i);
    __syncthreads();

    ProgView pv;
    pv.f = (const uint32_t *)(smem + a.hdr.off_f);
    pv.cmap = smem + a.hdr.off_cmap;
    pv.ptab = smem + a.hdr.off_ptab;
    pv.pages = smem + a.hdr.off_pages;
    pv.t8 = smem + a.hdr.off_table;
    pv.t16 = (MODE == MODE_GLOBAL) ? (const uint16_t *)(a.prog + a.hdr.off_table) : (const uint16_t *)(smem + a.hdr.off_table);
    pv.n_cols = a.hdr.n_cols;
    pv.pad_col = a.hdr.pad_col;
    pv.accept_lo = a.hdr.accept_lo;
    constexpr uint32_t SCALE = (MODE == MODE_NIBBLE) ? 4u : 1u; // state representation scale
    const uint32_t accept_lo = a.hdr.accept_lo * SCALE;
    const uint32_t start_state = a.hdr.start * SCALE;

    unsigned char *buf = smem + ((a.hdr.lds_bytes + 15u) & ~15u) + wave * kTileBytes;

    const uint64_t n_groups = (a.n_rows + 63) >> 6;
    const uint64_t wave_gid = (uint64_t)blockIdx.x * n_waves + wave;
    const uint64_t wave_cnt = (uint64_t)gridDim.x * n_waves;
    constexpr int CPP = 16 / CW; // chars per 16-B piece

    for (uint64_t g = wave_gid; g < n_groups; g += wave_cnt) {
        const uint64_t row0 = g << 6;
        const uint64_t my_row = row0 + lane;
        const bool row_ok = my_row < a.n_rows;
        uint32_t len = 0;
        if (row_ok) len = a.lengths ? a.lengths[my_row] : a.row_len;
        const uint32_t max_len = GUARD ? wave_max(len) : a.row_len;
        const uint32_t n_chunks = (max_len * CW + kChunkBytes - 1) / kChunkBytes;

        uint32_t st = start_state;
        int32_t last = -1; // OP_FIND: lastMatch of indexForwards
        if (OP == OP_FIND && a.hdr.root_accepting) last = 0; // DFAClassBuilder.java:356 (+ first-iteration check :440 with index == 0)

        for (uint32_t ck = 0; ck < n_chunks; ++ck) {
            stage_tile(a.rows, row0, a.n_rows, a.stride_bytes, a.total_bytes, ck * kChunkBytes, buf, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            uint32_t idx = ck * (kChunkBytes / CW);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const uint4 v = tile_piece(buf, lane, kk);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int d = 0; d < 4; ++d) {
#pragma unroll
                    for (int b = 0; b < 4 / CW; ++b) {
                        const uint32_t c = (CW == 1) ? ((w[d] >> (8 * b)) & 0xFFu) : ((w[d] >> (16 * b)) & 0xFFFFu);
                        st = step<MODE, CW, GUARD>(pv, st, c, idx < len);
                        ++idx;
                        if (OP == OP_FIND) last = (st >= accept_lo) ? (int32_t)idx : last;
                    }
                }
            }
            // wave-uniform early exit: every lane has an absorbing verdict (sink, or accepted for containedIn)
            bool live;
            if (OP == OP_CONTAINED_IN) live = st < accept_lo;
            else live = st != 0;
            if (GUARD) live = live && (idx < len);
            if (__ballot(live) == 0ull) break;
        }

        bool res;
        if (OP == OP_FIND) res = row_ok && (last >= 0);
        else res = row_ok && (st >= accept_lo);
        const uint64_t word = __ballot(res);
        if (lane == 0) a.bitmap[g] = word;

        if (OP == OP_FIND) {
            int32_t s = -1, e = res ? last : -1;
            if (a.fixed_len >= 0) {
                s = res ? last - a.fixed_len : -1; // DFAClassBuilder.java:640-646
            } else {
                // indexBackwards(end - 1, 0): DFAClassBuilder.java:536-583, automaton walked out of HBM/L2
                const uint8_t *bp = a.bprog;
                const uint8_t *bcmap = bp + a.bhdr.off_cmap, *bptab = bp + a.bhdr.off_ptab, *bpages = bp + a.bhdr.off_pages;
                const uint16_t *bt = (const uint16_t *)(bp + a.bhdr.off_table);
                const uint32_t bcols = a.bhdr.n_cols, bacc = a.bhdr.accept_lo;
                const uint8_t *rowp = a.rows + my_row * a.stride_bytes;
                int32_t idx = last - 1;
                uint32_t bs = a.bhdr.start;
                int32_t lastb = a.bhdr.root_accepting ? 0 : INT_MAX;
                bool active = res;
                while (__ballot(active) != 0ull) {
                    if (active) {
                        if (idx < 0) {
                            active = false;
                        } else {
                            const uint32_t c = (CW == 1) ? rowp[idx] : ((const uint16_t *)rowp)[idx];
                            const uint32_t col = (CW == 1) ? bcmap[c] : bpages[((uint32_t)bptab[c >> 8] << 8) | (c & 255u)];
                            bs = bt[bs * bcols + col];
                            if (bs == 0) {
                                active = false;
                            } else {
                                if (bs >= bacc) lastb = idx;
                                --idx;
                            }
                        }
                    }
                }
                s = res ? lastb : -1;
            }
            if (row_ok) {
                a.start[my_row] = s;
                a.end[my_row] = e;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------------
struct LaunchShape {
    int grid, waves;
    size_t lds;
};

template <int OP, int CW, int MODE, bool GUARD>
static hipError_t launch_one(const ScanArgs &a, LaunchShape sh, hipStream_t stream) {
    const int grid = sh.grid;
    const size_t lds = sh.lds;
    auto k = scan_kernel<OP, CW, MODE, GUARD>;
    static thread_local size_t configured = 0;
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) return e;
        configured = 160 * 1024;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(sh.waves * 64), lds, stream, a);
    return hipGetLastError();
}

template <int OP, int CW, int MODE>
static hipError_t launch_g(const ScanArgs &a, bool guard, LaunchShape sh, hipStream_t s) {
    return guard ? launch_one<OP, CW, MODE, true>(a, sh, s) : launch_one<OP, CW, MODE, false>(a, sh, s);
}

template <int OP, int CW>
static hipError_t launch_m(const ScanArgs &a, bool guard, LaunchShape sh, hipStream_t s) {
    switch (a.hdr.mode) {
    case MODE_NIBBLE: return launch_g<OP, CW, MODE_NIBBLE>(a, guard, sh, s);
    case MODE_TABLE8: return launch_g<OP, CW, MODE_TABLE8>(a, guard, sh, s);
    case MODE_TABLE16: return launch_g<OP, CW, MODE_TABLE16>(a, guard, sh, s);
    default: return launch_g<OP, CW, MODE_GLOBAL>(a, guard, sh, s);
    }
}

template <int OP>
static hipError_t launch_c(const ScanArgs &a, int cw, bool guard, LaunchShape sh, hipStream_t s) {
    return cw == 1 ? launch_m<OP, 1>(a, guard, sh, s) : launch_m<OP, 2>(a, guard, sh, s);
}

// Workgroup shape: as many waves as the 160 KiB LDS admits next to the automaton (16 -> 8 -> 4), one
// workgroup per CU, persistent over 64-row groups.
int waves_for_lds_bytes(uint32_t prog_lds_bytes) {
    const size_t p = (prog_lds_bytes + 15u) & ~15u;
    for (int w = kWavesPerBlock; w >= 4; w >>= 1)
        if (p + (size_t)w * kTileBytes <= 160u * 1024u) return w;
    return 0;
}

hipError_t launch_scan(int op, int char_width, const ScanArgs &a, int n_cus, hipStream_t stream) {
    if (a.n_rows == 0) return hipSuccess;
    LaunchShape sh;
    sh.waves = waves_for_lds_bytes(a.hdr.lds_bytes);
    if (sh.waves == 0) return hipErrorInvalidValue;
    const uint64_t n_groups = (a.n_rows + 63) >> 6;
    uint64_t blocks = (n_groups + sh.waves - 1) / sh.waves;
    if (blocks > (uint64_t)n_cus) blocks = (uint64_t)n_cus;
    sh.grid = (int)blocks;
    sh.lds = ((a.hdr.lds_bytes + 15u) & ~15u) + (size_t)sh.waves * kTileBytes;
    const bool guard = a.lengths != nullptr || ((uint64_t)a.row_len * char_width) % kChunkBytes != 0;
    switch (op) {
    case OP_MATCHES: return launch_c<OP_MATCHES>(a, char_width, guard, sh, stream);
    case OP_CONTAINED_IN: return launch_c<OP_CONTAINED_IN>(a, char_width, guard, sh, stream);
    default: return launch_c<OP_FIND>(a, char_width, guard, sh, stream);
    }
}

} // namespace needle
