// needle_dict.hip -- the big-automaton walk with TWO 64-row sets per wavefront (gfx950 / CDNA4).
//
// What bounds the walk of an automaton that fills the LDS (a 1000-keyword dictionary: DFAClassBuilder.java:438-468 with a
// 275 KB STATES_FORWARDS table) is the per-char dependent chain -- LDS round trip, compare, select, address, LDS -- and the
// number of such chains a CU keeps in flight: one row per lane, 16 waves, 1024 (the workgroup limit: the rows' walks have to
// share the ONE LDS copy of the automaton).  profiles/r03_pmc.md: LDS array 53 % busy, VALU 62 %, 57 % of the wave-cycles
// parked.  Here a wave walks two 64-row sets side by side -- the two chains' lookups are issued together and their selects
// interleave -- with 12 waves per workgroup: 1536 chains per CU, and every wave hides its own LDS latency behind the other
// set's instructions.  The price is LDS for the rows in flight, paid with 32-byte tiles (a quarter of a line per row and set);
// a line still crosses HBM once: its four quarters are requested back to back into registers, one line ahead of the walk.
//
// Scope: 8-bit rows that fill their stride (no lengths, no cursors), stride a multiple of 128 bytes, automata in the
// compressed (MODE_SPARSE) or plain uint16 (MODE_TABLE16) LDS form.  find() leaves lastMatch in end[]; a start by
// indexBackwards is found afterwards, one lane per matched row (backward_row_kernel, needle_stripe.hip).  Everything else
// takes the ordinary tiled kernel (needle_scan.h).
#include "needle_walk.h"

namespace needle {

constexpr int kDictWaves = 12;    // 3 per SIMD: 168 VGPRs each
constexpr int kDictTileB = 32;    // bytes per row and tile
constexpr int kDictSetTile = 64 * kDictTileB;

// both sets' transitions for one char: lookups issued together, ONE rare branch for the record chains of either
template <int MODE>
__device__ __forceinline__ void apply2(const Walk &wk, uint32_t &sa, uint32_t &sb, uint32_t ca, uint32_t cb) {
    if (MODE == MODE_SPARSE) {
        const uint32_t tb = (uint32_t)kLdsTable1;
        uint32_t aa, ab;
        asm("v_mad_u32_u16 %0, %1, 4, %2" : "=v"(aa) : "v"(sa), "v"(ca));
        asm("v_mad_u32_u16 %0, %1, 4, %2" : "=v"(ab) : "v"(sb), "v"(cb));
        u32x2 ra = lds_u32x2((sa >> 16) + tb);
        u32x2 rb = lds_u32x2((sb >> 16) + tb);
        const uint32_t da = lds_u32(aa + tb);
        const uint32_t db = lds_u32(ab + tb);
        // (lane masks straight from the compares, and the selects written against those masks: the compiler otherwise compares
        // twice -- once for the mask, once for the select)
        const uint64_t eqa = __builtin_amdgcn_uicmp(ra[0] & 0xFFFFu, ca, 32), eqb = __builtin_amdgcn_uicmp(rb[0] & 0xFFFFu, cb, 32);
        uint32_t na, nb;
        // (s_nop 1: an SDWA compare's SGPR result needs wait states before a VALU reads it -- the hazard recogniser does not see
        // into inline asm, so they are spelled out)
        asm("s_nop 1\n\tv_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(na) : "v"(da), "v"(ra[1]), "s"(eqa));
        asm("s_nop 1\n\tv_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(nb) : "v"(db), "v"(rb[1]), "s"(eqb));
        const uint64_t chained = (__builtin_amdgcn_uicmp(ra[0], 0xFFFFu, 34) & ~eqa) | (__builtin_amdgcn_uicmp(rb[0], 0xFFFFu, 34) & ~eqb);
        if (__builtin_expect(chained != 0ull, 0)) { // states with two or three exceptions (rare lanes): follow the records
            bool ha = (ra[0] & 0xFFFFu) == ca, hb = (rb[0] & 0xFFFFu) == cb;
            bool ma = !ha && ra[0] > 0xFFFFu, mb = !hb && rb[0] > 0xFFFFu;
            do {
                if (ma) {
                    ra = lds_u32x2((ra[0] >> 16) + tb);
                    ha = (ra[0] & 0xFFFFu) == ca;
                    na = ha ? ra[1] : na;
                    ma = !ha && ra[0] > 0xFFFFu;
                }
                if (mb) {
                    rb = lds_u32x2((rb[0] >> 16) + tb);
                    hb = (rb[0] & 0xFFFFu) == cb;
                    nb = hb ? rb[1] : nb;
                    mb = !hb && rb[0] > 0xFFFFu;
                }
            } while (__ballot(ma || mb) != 0ull);
        }
        sa = na;
        sb = nb;
    } else {
        sa = apply<MODE, 1>(wk, sa, ca);
        sb = apply<MODE, 1>(wk, sb, cb);
    }
}

template <int OP, int MODE>
__global__ __launch_bounds__(kDictWaves * 64) void dict_kernel(const ScanArgs a) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();
    for (uint32_t i = tid * 16u; i < a.hdr.lds_bytes; i += blockDim.x * 16u) *(u32x4 *)(smem + i) = *(const u32x4 *)(a.prog + i);
    __syncthreads();

    Walk wk;
    constexpr uint32_t ELEM = MODE == MODE_TABLE16 ? 2u : 1u;
    wk.ncols_e = a.hdr.n_cols * ELEM;
    wk.pad_e = wk.pre_e = wk.pad_b = wk.pre_b = 0;
    wk.win_on = a.hdr.win_on, wk.win_lo = a.hdr.win_lo_e, wk.win_hi = a.hdr.win_hi_e;
    wk.sp_chains = a.hdr.sp_chains, wk.sp_pad_ident = 0, wk.dead_hi = 0;
    wk.table_off = a.hdr.off_table, wk.lane4 = 0, wk.gtable = nullptr, wk.hot_last = 0;
    const uint32_t accept_lo = a.hdr.accept_lo, start_state = a.hdr.start;

    // this wave's two tiles: 64 rows x 32 bytes each; piece p of row r at r * 32 + ((p ^ swz(r)) << 4), swz(r) = (r >> 3) & 1: the
    // ds_read_b128 of 16 lanes then touch all 64 banks once (rows 8 apart would share banks)
    const uint32_t tile0 = ((a.hdr.lds_bytes + 15u) & ~15u) + (uint32_t)wave * 2u * kDictSetTile;
    const uint32_t my_swz = ((uint32_t)lane >> 3) & 1u;
    const uint32_t rd_addr = (uint32_t)lane * kDictTileB; // + set * kDictSetTile + ((p ^ my_swz) << 4)
    // loads: one instruction = 32 rows x 32 bytes, lane -> row (lane >> 1), half (lane & 1); stored lane-linear, so the lane
    // fetches SOURCE piece half ^ swz(row) -- and swz(row) = (lane >> 4) & 1 for both instructions of a tile
    const uint32_t st_addr = (uint32_t)lane * 16u; // + set * kDictSetTile + j * 1024
    const uint32_t stride = (uint32_t)a.stride_bytes;
    const uint32_t src_half = (((uint32_t)lane & 1u) ^ (((uint32_t)lane >> 4) & 1u)) << 4;
    const uint32_t o0 = ((uint32_t)lane >> 1) * stride + src_half, o1 = (32u + ((uint32_t)lane >> 1)) * stride + src_half;
    const uint32_t n_lines = stride >> 7;

    const uint64_t n_pg = a.n_rows >> 7; // whole 128-row pairs only (the launcher gives the rest to the ordinary kernel)
    const uint64_t wave_cnt = (uint64_t)gridDim.x * kDictWaves;
    uint64_t pg = (uint64_t)blockIdx.x * kDictWaves + wave;
    if (pg >= n_pg) return;

    u32x4 R[2][4][2]; // [set][quarter of the line][32-row half of the set]: one line per row, one line ahead of the walk
    auto fetch_quarter = [&](uint64_t g, uint32_t line, int s, int t) __attribute__((always_inline)) {
        const uint8_t *base = a.rows + ((g << 7) + (uint64_t)s * 64u) * a.stride_bytes + (uint64_t)line * 128u + (uint32_t)t * kDictTileB;
        R[s][t][0] = load_row16<false>(base + o0);
        R[s][t][1] = load_row16<false>(base + o1);
    };
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        fetch_quarter(pg, 0, 0, t);
        fetch_quarter(pg, 0, 1, t);
    }
    for (;;) {
        uint32_t sa = start_state, sb = start_state;
        int32_t la = -1, lb = -1;
        const uint64_t npg = pg + wave_cnt;
        for (uint32_t line = 0; line < n_lines; ++line) {
            const bool last_line = line + 1 == n_lines;
            const bool have_next = !last_line || npg < n_pg;
            const uint64_t ng = last_line ? npg : pg;
            const uint32_t nl = last_line ? 0u : line + 1u;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                // stage quarter t of both sets, then ask for the same quarter of the next line into the freed registers
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    *(lds_u32x4 *)(uintptr_t)(tile0 + (uint32_t)s * kDictSetTile + st_addr) = R[s][t][0];
                    *(lds_u32x4 *)(uintptr_t)(tile0 + (uint32_t)s * kDictSetTile + 1024u + st_addr) = R[s][t][1];
                }
                asm volatile("" ::: "memory");
                // the next line is requested WHOLE, its four quarters back to back, once the last quarter of this one is staged
                // (asking for each quarter as its registers come free would spread a line's four requests over a tile walk each --
                // long enough for the line to leave the vector L1 and the L2 in between: up to four fetches of every line)
                if (t == 3 && have_next) {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        fetch_quarter(ng, nl, 0, tt);
                        fetch_quarter(ng, nl, 1, tt);
                    }
                }
                asm volatile("" ::: "memory");
                const uint32_t idx0 = line * 128u + (uint32_t)t * kDictTileB;
#pragma unroll 1
                for (int p = 0; p < 2; ++p) {
                    const u32x4 va = *(const lds_u32x4 *)(uintptr_t)(tile0 + rd_addr + (((uint32_t)p ^ my_swz) << 4));
                    const u32x4 vb = *(const lds_u32x4 *)(uintptr_t)(tile0 + kDictSetTile + rd_addr + (((uint32_t)p ^ my_swz) << 4));
                    const uint32_t wa[4] = {va[0], va[1], va[2], va[3]}, wb[4] = {vb[0], vb[1], vb[2], vb[3]};
                    uint32_t ca[16], cb[16];
                    piece_lookups<MODE, 1, false>(wk, wa, 0, 0, 0, ca);
                    piece_lookups<MODE, 1, false>(wk, wb, 0, 0, 0, cb);
                    uint32_t lra = 0, lrb = 0;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        apply2<MODE>(wk, sa, sb, ca[i], cb[i]);
                        if (OP == OP_FIND) {
                            lra = sa >= accept_lo ? (uint32_t)(i + 1) : lra;
                            lrb = sb >= accept_lo ? (uint32_t)(i + 1) : lrb;
                        }
                    }
                    if (OP == OP_FIND) {
                        la = lra ? (int32_t)(idx0 + (uint32_t)p * 16u + lra) : la;
                        lb = lrb ? (int32_t)(idx0 + (uint32_t)p * 16u + lrb) : lb;
                    }
                }
            }
            // wave-uniform early exit: every row of both sets has an absorbing verdict (the sink, or accepted for containedIn)
            bool live;
            if (OP == OP_CONTAINED_IN) live = sa < accept_lo || sb < accept_lo;
            else live = sa != 0u || sb != 0u;
            if (!last_line && __ballot(live) == 0ull) {
                if (npg < n_pg) { // (the line in flight belongs to this pair: replace it with the next pair's first line)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        fetch_quarter(npg, 0, 0, t);
                        fetch_quarter(npg, 0, 1, t);
                    }
                }
                break;
            }
        }
        // ---- verdicts
        const uint64_t row_a = (pg << 7) + (uint64_t)lane, row_b = row_a + 64u;
        bool res_a, res_b;
        if (OP == OP_FIND) res_a = la >= 0, res_b = lb >= 0;
        else res_a = sa >= accept_lo, res_b = sb >= accept_lo;
        const uint64_t word_a = __ballot(res_a), word_b = __ballot(res_b);
        if (lane == 0) {
            a.bitmap[pg * 2] = word_a;
            a.bitmap[pg * 2 + 1] = word_b;
        }
        if (OP == OP_FIND) {
            a.end[row_a] = res_a ? la : -1;
            a.end[row_b] = res_b ? lb : -1;
            if (a.fixed_len >= 0) { // :640-646; else indexBackwards afterwards (backward_row_kernel)
                a.start[row_a] = res_a ? la - a.fixed_len : -1;
                a.start[row_b] = res_b ? lb - a.fixed_len : -1;
            }
        }
        if (npg >= n_pg) break;
        pg = npg;
    }
}

template <int OP, int MODE>
static hipError_t launch_dict_one(const ScanArgs &a, int grid, size_t lds, hipStream_t stream) {
    auto k = dict_kernel<OP, MODE>;
    static thread_local uint64_t configured = 0;
    if (hipError_t e = allow_full_lds((const void *)k, configured); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(kDictWaves * 64), lds, stream, a);
    return hipGetLastError();
}

// true: the batch / program pair is one this kernel takes (see the header comment)
bool dict_kernel_applies(int char_width, const ScanArgs &a) {
    if (char_width != 1 || a.lengths || a.from || a.end_state) return false;
    if (a.hdr.mode != MODE_SPARSE && a.hdr.mode != MODE_TABLE16) return false;
    if (a.row_len == 0 || a.row_len != a.stride_bytes || (a.stride_bytes & 127u) != 0 || a.stride_bytes > (1u << 16)) return false;
    if (a.n_rows < 128 * 256) return false; // (small batches: the ordinary kernel's finer groups fill the chip better)
    return ((a.hdr.lds_bytes + 15u) & ~15u) + (size_t)kDictWaves * 2 * kDictSetTile <= 160u * 1024u;
}

// the first (n_rows / 128) * 128 rows of the batch; the caller runs the ordinary kernel on the rest
hipError_t launch_dict(int op, const ScanArgs &a, int n_cus, hipStream_t stream) {
    const uint64_t n_pg = a.n_rows >> 7;
    if (n_pg == 0) return hipSuccess;
    uint64_t blocks = (n_pg + kDictWaves - 1) / kDictWaves;
    if (blocks > (uint64_t)n_cus) blocks = (uint64_t)n_cus;
    const size_t lds = ((a.hdr.lds_bytes + 15u) & ~15u) + (size_t)kDictWaves * 2 * kDictSetTile;
    const bool sparse = a.hdr.mode == MODE_SPARSE;
    switch (op) {
    case OP_MATCHES: return sparse ? launch_dict_one<OP_MATCHES, MODE_SPARSE>(a, (int)blocks, lds, stream) : launch_dict_one<OP_MATCHES, MODE_TABLE16>(a, (int)blocks, lds, stream);
    case OP_CONTAINED_IN: return sparse ? launch_dict_one<OP_CONTAINED_IN, MODE_SPARSE>(a, (int)blocks, lds, stream) : launch_dict_one<OP_CONTAINED_IN, MODE_TABLE16>(a, (int)blocks, lds, stream);
    default: return sparse ? launch_dict_one<OP_FIND, MODE_SPARSE>(a, (int)blocks, lds, stream) : launch_dict_one<OP_FIND, MODE_TABLE16>(a, (int)blocks, lds, stream);
    }
}

} // namespace needle
