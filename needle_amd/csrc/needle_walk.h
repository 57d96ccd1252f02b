// Device-side helpers shared by the kernels of libneedle_hip.so (needle_kernels.hip: the one-row-per-lane scan;
// needle_stripe.hip: the stripe path for few long rows and the packed-rows conversion): LDS access at absolute
// addresses, SDWA byte selects, the tile geometry and the per-char lookup / transition of a lowered automaton.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "needle_device.h"

#ifndef NEEDLE_MASK_DONE_LANES
#define NEEDLE_MASK_DONE_LANES 1
#endif
#ifndef NEEDLE_NT_LOADS
#define NEEDLE_NT_LOADS 1
#endif
#ifndef NEEDLE_SPLIT_BOUNDARY
#define NEEDLE_SPLIT_BOUNDARY 1
#endif
#ifndef NEEDLE_BIG_ROLLED // 1: the big-table modes' piece loops stay rolled (code size; needle_scan.h walk_tile).  Measured
#define NEEDLE_BIG_ROLLED 0 // (profiles/r04_code_diet.md): 70 -> 23 KB per kernel, 2-3 % SLOWER, I-cache misses ~0 either way
#endif
#ifndef NEEDLE_PIECE_FENCE
#define NEEDLE_PIECE_FENCE 1
#endif

namespace needle {

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

// native 16-byte vector (a first-class SSA value: tiles held across loop iterations stay in VGPRs; HIP's uint4
// wrapper struct gets demoted to scratch when it is conditionally re-assigned)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int CHB>
struct Geom {
    static constexpr int kPieces = CHB / 16;         // 16-B pieces per row chunk: 8 | 4
    static constexpr int kRowsPerInstr = 64 / kPieces; // rows covered by one wave-wide 16 B/lane load: 8 | 16
    static constexpr int kInstrs = 64 / kRowsPerInstr; // loads per lane per tile: 8 | 4
    static constexpr int kTileBytes = 64 * CHB;
    // bank-slot swizzle of tile row r (see header comment): distinct for the rows one ds_read_b128 lane group touches
    __device__ static __forceinline__ int swz(int r) { return CHB == 128 ? ((r >> 1) & 7) : ((r >> 2) & 3); }
};

typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;

// 16 bytes of haystack.  STREAM: the wave consumes whole 128-byte lines exactly once -> `global_load ... nt`
// (measured on the 10M x 256 batch: 5.5 -> 6.2 TB/s).  Not for the 64-byte-piece shape: there the second half of a
// line must still be in L2 when its request arrives right behind the first one's (nt there: 0.70 -> 0.91 ms).
template <bool STREAM>
__device__ __forceinline__ u32x4 load_row16(const uint8_t *p) {
    if (STREAM && NEEDLE_NT_LOADS) return __builtin_nontemporal_load((const u32x4 *)p);
    return *(const u32x4 *)p;
}

// A wave's LDS tile: 64 rows of CHB bytes at `row_stride` bytes apart (row_stride == CHB for the plain layout; 256
// when the rows live in the unused upper halves of the packed-mode F rows, see shape_for_program).
struct Tile {
    uint32_t store_addr;  // this lane's first store slot: base + (lane / pieces) * row_stride + (lane % pieces) * 16
    uint32_t store_step;  // rows-per-instruction * row_stride
    uint32_t row_addr;    // base + lane * row_stride: this lane's own row
};

__device__ __forceinline__ void store_piece(const Tile &t, int j, u32x4 v) {
    *(lds_u32x4 *)(uintptr_t)(t.store_addr + j * t.store_step) = v;
}

template <int CHB>
__device__ __forceinline__ u32x4 tile_piece(const Tile &t, int lane, int kk) {
    return *(const lds_u32x4 *)(uintptr_t)(t.row_addr + ((kk ^ Geom<CHB>::swz(lane)) << 4));
}

__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t t = (uint32_t)__shfl_xor((int)v, o);
        v = v > t ? v : t;
    }
    return v;
}

// ---- one-instruction byte/word extraction (SDWA operand selects): the walk is VALU-issue bound (one wave
// instruction per ~4 cycles per SIMD), so every per-char VALU instruction saved is throughput.
template <int K>
__device__ __forceinline__ uint32_t shl_byte(uint32_t w, uint32_t sh) { // (byte K of w) << sh
    uint32_t r;
    if (K == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(sh), "v"(w));
    if (K == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(sh), "v"(w));
    if (K == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(sh), "v"(w));
    if (K == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(sh), "v"(w));
    return r;
}
template <int K>
__device__ __forceinline__ uint32_t or_byte(uint32_t a, uint32_t w) { // a | (byte K of w)
    uint32_t r;
    if (K == 0) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(a), "v"(w));
    if (K == 1) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(a), "v"(w));
    if (K == 2) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(a), "v"(w));
    if (K == 3) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(a), "v"(w));
    return r;
}

template <int K>
__device__ __forceinline__ uint32_t shl_word(uint32_t w, uint32_t sh) { // (16-bit half K of w) << sh
    uint32_t r;
    if (K == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(sh), "v"(w));
    if (K == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(sh), "v"(w));
    return r;
}

// element size of a mode's table cells as a shift (window addressing: char << shift, clamped, IS the column offset)
template <int MODE>
__device__ __forceinline__ constexpr uint32_t elem_shift() {
    return MODE == MODE_SPARSE ? 2u : (MODE == MODE_TABLE16 || MODE == MODE_HYBRID) ? 1u : 0u;
}

// LDS reads at an ABSOLUTE LDS byte address through address-space-3 pointers.  The dynamic segment starts at LDS
// address 0 (this file declares no static __shared__; scan_kernel traps if that ever changes), so table offsets
// are plain immediates: going through `smem` costs a `v_add 0` (late-resolved symbol) per access, going through
// generic pointers a null-check v_cndmask on top.
#define NEEDLE_LDS(T) __attribute__((address_space(3))) const T *
__device__ __forceinline__ uint32_t lds_u8(uint32_t a) { return *(NEEDLE_LDS(uint8_t))(uintptr_t)(a); }
__device__ __forceinline__ uint32_t lds_u16(uint32_t a) { return *(NEEDLE_LDS(uint16_t))(uintptr_t)(a); }
__device__ __forceinline__ uint32_t lds_u32(uint32_t a) { return *(NEEDLE_LDS(uint32_t))(uintptr_t)(a); }
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 lds_u32x2(uint32_t a) { return *(NEEDLE_LDS(u32x2))(uintptr_t)(a); }

// One wait for every LDS read issued so far, and nothing scheduled across it.
__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
}

// Walk constants a lane keeps in registers (see the fixed LDS layout in needle_device.h).
struct Walk {
    uint32_t ncols_e;   // table modes: row stride in BYTES of the next-state table (n_cols * element size)
    uint32_t pad_e;     // table modes: PAD column * element size;  packed mode: F of the PAD column
    uint32_t pre_e;     // same for the PRE column (identity: chars before the row's find() cursor)
    uint32_t pad_b, pre_b; // pair mode: PAD / PRE as the SECOND char of a pair (pad_e / pre_e: as the first)
    uint32_t table_off; // char_width 2 table modes: LDS byte offset of the table
    uint32_t lane4;     // lane * 4 (byte 0 of the packed-mode F address)
    const uint16_t *gtable; // MODE_GLOBAL / MODE_HYBRID: the whole table in HBM
    uint32_t hot_last;      // MODE_HYBRID: byte offset of the last table entry held in LDS (hot_bytes - 2)
    uint32_t win_on, win_lo, win_hi; // window addressing (needle_device.h): clamp bounds of char * element size
    uint32_t dead_hi;       // find(): states 0 .. dead_hi end the search (0: the sink alone; the "lengths" programs add their
                            // dead-with-a-match-pending states D_L, needle_lower.h)
    uint32_t sp_chains;     // MODE_SPARSE: some state has more than one exception record (wave-uniform)
    uint32_t sp_pad_ident;  // MODE_SPARSE: PAD is the identity (matches / containedIn) rather than the way to the sink
    uint32_t flat = 0;      // UTF-16 table modes: the page map is flat (ProgHeader::flat_pages): column = pages[char], no ptab lookup
};

template <int CW>
__device__ __forceinline__ uint32_t column_of(const uint8_t *cmap, const uint8_t *ptab, const uint8_t *pages, uint32_t c) {
    if (CW == 1) return cmap[c];
    return pages[((uint32_t)ptab[c >> 8] << 8) | (c & 255u)];
}

// UTF-16 table programs: the table's LDS offset (a kernel argument, not a compile-time constant as for 8-bit rows) is added to the COLUMN
// offset ahead of the state chain, so that a transition is one v_mad_u32_u24 + the LDS read instead of v_mul_u32_u24 + v_add3_u32 + the read.
template <int MODE, int CW>
__device__ __forceinline__ constexpr bool col_has_table_off() { return CW == 2 && (MODE == MODE_TABLE8 || MODE == MODE_TABLE16); }

// A transition in two halves so that a whole 16-byte piece can be batched: `lookup` is everything that does not
// depend on the automaton state (char -> F, or char -> column * element size); `apply` is the dependent part.
// K: char number inside dword w (0..3 for bytes, 0..1 for UTF-16 units).
template <int MODE, int CW, bool GUARD, int K>
__device__ __forceinline__ uint32_t lookup(const Walk &wk, uint32_t w, bool in_row, bool before_cursor) {
    uint32_t col; // packed mode: F;  table modes: column * element size
    if (CW == 1) {
        // packed mode: F[byte][32 lane copies]: address = byte << 8 | (lane & 31) * 4, formed by ONE v_perm_b32;
        // every lane reads its own LDS bank, so the lookup is conflict-free whatever the text looks like
        if (MODE == MODE_PACK) col = lds_u32(__builtin_amdgcn_perm(w, wk.lane4, 0x0C0C0400u + ((uint32_t)K << 8)) + kLdsF1);
        else if (MODE == MODE_PAIR) col = lds_u16(shl_byte<K>(w, 1) + ((K & 1) ? kLdsCmapB1 : kLdsCmap1)); // first | second char of a pair
        else col = lds_u16(shl_byte<K>(w, 1) + kLdsCmap1);
    } else {
        if (MODE == MODE_PACK) { // ptab64[high byte] = {base, mask}; F of the char at base | (low byte * 4 & mask)
            const u32x2 pg = lds_u32x2(shl_byte<(2 * K + 1) & 3>(w, 3) + kLdsPtab2);
            col = lds_u32(((shl_byte<(2 * K) & 3>(w, 2) & pg[1]) | pg[0]) + kLdsPagesF2);
        } else {
            const uint32_t pg = lds_u16(shl_byte<(2 * K + 1) & 3>(w, 1) + kLdsPtab2);  // page base = page * 256
            col = lds_u8(or_byte<(2 * K) & 3>(pg, w) + kLdsPages2Table);                // pages hold column * element size
        }
    }
    // (opaque to the optimiser: else the record-key compare of apply() is narrowed to 16 bits and col re-extended with a v_and per char)
    if (MODE == MODE_SPARSE) asm("" : "+v"(col));
    if (GUARD) {
        col = in_row ? col : ((MODE == MODE_PAIR && (K & 1)) ? wk.pad_b : wk.pad_e);
        col = before_cursor ? ((MODE == MODE_PAIR && (K & 1)) ? wk.pre_b : wk.pre_e) : col;
    }
    return col;
}
// st: 5 * state in MODE_PACK (the bit offset of the state's field in F), the state id otherwise.
template <int MODE, int CW>
__device__ __forceinline__ uint32_t apply(const Walk &wk, uint32_t st, uint32_t col) {
    if (MODE == MODE_PACK) return __builtin_amdgcn_ubfe(col, st, 5);
    if (MODE == MODE_HYBRID) {
        // st carries the "accepting" flag in bit 15 (so that accepted(s) stays s >= accept_lo = 0x8000 whatever the
        // numbering); rows of the hot states come from LDS, colder ones through the scalar cache: a per-lane HBM load
        // would have to be waited for with vmcnt, behind the tile prefetch that is in flight.
        const uint32_t i = __umul24(st & 0x7FFFu, wk.ncols_e) + col; // byte offset into the table
        // every lane reads LDS -- a cold lane whatever sits at its index (tile bytes, or 0 beyond the workgroup's LDS: out-
        // of-range DS reads return 0 and do not fault); clamping the address cost a v_min per char: C3-sparse 1.287 -> 1.236 ms ...
        uint32_t v = lds_u16(i + (CW == 1 ? (uint32_t)kLdsTable1 : wk.table_off));
        uint64_t cold = __ballot(i > wk.hot_last);                  // ... and the cold ones are patched below
        while (cold != 0ull) { // rare: one scalar load per cold lane
            const int l = __builtin_ctzll(cold);
            cold &= cold - 1ull;
            const uint32_t ci = (uint32_t)__builtin_amdgcn_readlane((int)i, l);
            typedef __attribute__((address_space(4))) const uint32_t *cptr_t; // constant address space: s_load_dword
            const uint32_t word = *(cptr_t)((uintptr_t)wk.gtable + (ci & ~3u));
            const uint32_t e = (word >> ((ci & 2u) * 8u)) & 0xFFFFu;
            v = (__lane_id() == (unsigned)l) ? e : v;
        }
        return v;
    }
    if (MODE == MODE_SPARSE) {
        // st = recB << 16 | rowA4 (needle_device.h): the cell of the dense row the state reads (its own, or its default
        // row's) and the state's first exception record are fetched side by side -- ONE LDS round trip; the record wins when
        // its key is this char's column * 4.  Dense states point at a dummy record whose key matches nothing: all of them
        // read one address, which the LDS broadcasts.  col = column * 4.
        const uint32_t tb = CW == 1 ? (uint32_t)kLdsTable1 : wk.table_off;
        uint32_t a_addr;
        asm("v_mad_u32_u16 %0, %1, 4, %2" : "=v"(a_addr) : "v"(st), "v"(col)); // (st & 0xFFFF) * 4 + col
        const uint32_t a = lds_u32(a_addr + tb);
        u32x2 b = lds_u32x2((st >> 16) + tb);
        bool hit = (b[0] & 0xFFFFu) == col;
        uint32_t nxt = hit ? b[1] : a;
        // States with two or three exceptions chain their records (the key's dword carries the next record's address in
        // its high half): rare lanes, but the test sits on every char's dependent chain, so it is kept to one VALU compare
        // and two scalar ops -- lane masks straight from the compares (a ballot of the combined predicate costs two more
        // VALU ops), no flag test in front of it (programs without chains simply never branch).
        const uint64_t chained = __builtin_amdgcn_uicmp(b[0], 0xFFFFu, 34 /* ugt */) & ~__builtin_amdgcn_uicmp(b[0] & 0xFFFFu, col, 32 /* eq */);
        if (__builtin_expect(chained != 0ull, 0)) {
            bool more = (chained >> __lane_id()) & 1ull;
            do {
                if (more) {
                    b = lds_u32x2((b[0] >> 16) + tb);
                    hit = (b[0] & 0xFFFFu) == col;
                    nxt = hit ? b[1] : nxt;
                    more = !hit && b[0] > 0xFFFFu;
                }
            } while (__ballot(more) != 0ull);
        }
        return nxt;
    }
    const uint32_t i = __umul24(st, wk.ncols_e) + col;
    if (MODE == MODE_GLOBAL) return wk.gtable[i];
    const uint32_t addr = i + (CW == 1 ? (uint32_t)kLdsTable1 : (col_has_table_off<MODE, CW>() ? 0u : wk.table_off));
    return MODE == MODE_TABLE8 ? lds_u8(addr) : lds_u16(addr);
}

// Compressed lengths programs (needle_device.h): where the END of the row leads from state st -- the target of the END record in
// the state's chain, the sink if it has none.  Lanes not in `live` keep their state.
template <int CW>
__device__ __forceinline__ uint32_t sparse_end(const Walk &wk, uint32_t st, bool live, uint32_t end_key) {
    const uint32_t tb = CW == 1 ? (uint32_t)kLdsTable1 : wk.table_off;
    uint32_t r = live ? 0u : st, rec = st >> 16;
    bool more = live;
    while (__ballot(more) != 0ull) {
        if (more) {
            const u32x2 b = lds_u32x2(rec + tb);
            const bool hit = (b[0] & 0xFFFFu) == end_key;
            r = hit ? b[1] : r;
            more = !hit && b[0] > 0xFFFFu;
            rec = b[0] >> 16;
        }
    }
    return r;
}

// Everything of a 16-byte piece's walk that does not depend on the automaton state: per char its F (packed mode) or
// its column * element size (table modes).  GUARD: chars at p0 + i >= rem take the PAD column, chars at p0 + i < skip PRE.
template <int MODE, int CW, bool GUARD>
__device__ __forceinline__ void piece_lookups(const Walk &wk, const uint32_t (&w)[4], uint32_t p0, uint32_t rem, uint32_t skip,
                                              uint32_t (&col)[16 / CW]) {
    constexpr int CPP = 16 / CW;
    if (MODE != MODE_PACK && MODE != MODE_PAIR && wk.win_on) { // wave-uniform: window addressing -- no column-map lookup at all
        const uint32_t sh = elem_shift<MODE>();
        const uint32_t lo_v = wk.win_lo;
#define NEEDLE_WIN(I, EXPR)                                                        \
    {                                                                              \
        uint32_t c = EXPR;                                                         \
        asm("v_med3_u32 %0, %1, %2, %3" : "=v"(c) : "v"(c), "v"(lo_v), "s"(wk.win_hi)); /* one SGPR per VOP3 on gfx9 */ \
        if (GUARD) {                                                               \
            c = (p0 + (I) < rem) ? c : wk.pad_e;                                   \
            c = (p0 + (I) < skip) ? wk.pre_e : c;                                  \
        }                                                                          \
        col[I] = c;                                                                \
    }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if constexpr (CW == 1) {
                NEEDLE_WIN(d * 4 + 0, shl_byte<0>(w[d], sh))
                NEEDLE_WIN(d * 4 + 1, shl_byte<1>(w[d], sh))
                NEEDLE_WIN(d * 4 + 2, shl_byte<2>(w[d], sh))
                NEEDLE_WIN(d * 4 + 3, shl_byte<3>(w[d], sh))
            } else {
                NEEDLE_WIN(d * 2 + 0, shl_word<0>(w[d], sh))
                NEEDLE_WIN(d * 2 + 1, shl_word<1>(w[d], sh))
            }
        }
#undef NEEDLE_WIN
        if (col_has_table_off<MODE, CW>()) {
#pragma unroll
            for (int i = 0; i < CPP; ++i) {
                col[i] += wk.table_off;
                asm("" : "+v"(col[i])); // (kept off the state chain: the compiler would fold it back into the transition's add)
            }
        }
        return;
    }
    // all state-independent lookups of the piece first (they pipeline in the LDS) ...
    if (CW == 2) {
        // UTF-16: two dependent lookups per char before the state chain -- packed mode: page table -> F; table modes:
        // page table -> page -> column (the table lookup follows on the state chain).  Issued as batches of 8 with ONE
        // wait between batches: left to the scheduler they come out as ~14 short waits per piece, each exposing a full
        // LDS round trip.
        uint32_t pg[CPP];
        if (MODE == MODE_PACK) {
            uint32_t lo4[CPP];
            u32x2 bm[CPP]; // {base, mask} of the char's page
#define NEEDLE_PG(D, K)                                                                              \
    bm[(D) * 2 + (K)] = lds_u32x2(shl_byte<(2 * (K) + 1) & 3>(w[D], 3) + kLdsPtab2);                 \
    lo4[(D) * 2 + (K)] = shl_byte<(2 * (K)) & 3>(w[D], 2);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                NEEDLE_PG(d, 0)
                NEEDLE_PG(d, 1)
            }
#undef NEEDLE_PG
            lds_fence();
#pragma unroll
            for (int i = 0; i < CPP; ++i) {
                uint32_t c = lds_u32(((lo4[i] & bm[i][1]) | bm[i][0]) + kLdsPagesF2); // v_and_or_b32
                if (GUARD) {
                    c = (p0 + i < rem) ? c : wk.pad_e;
                    c = (p0 + i < skip) ? wk.pre_e : c;
                }
                col[i] = c;
            }
        } else {
#define NEEDLE_PG(D, K) pg[(D) * 2 + (K)] = lds_u16(shl_byte<(2 * (K) + 1) & 3>(w[D], 1) + kLdsPtab2);
#define NEEDLE_CE(D, K) col[(D) * 2 + (K)] = lds_u8(or_byte<(2 * (K)) & 3>(pg[(D) * 2 + (K)], w[D]) + kLdsPages2Table);
            if (wk.flat) { // wave-uniform: a 64 KB map indexed by the char itself -- one lookup per char, no fence between two levels
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    col[d * 2 + 0] = lds_u8(shl_word<0>(w[d], 0) + kLdsPages2Table);
                    col[d * 2 + 1] = lds_u8(shl_word<1>(w[d], 0) + kLdsPages2Table);
                }
            } else {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                NEEDLE_PG(d, 0)
                NEEDLE_PG(d, 1)
            }
            lds_fence();
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                NEEDLE_CE(d, 0)
                NEEDLE_CE(d, 1)
            }
            }
#undef NEEDLE_PG
#undef NEEDLE_CE
            if (MODE == MODE_SPARSE) {
#pragma unroll
                for (int i = 0; i < CPP; ++i) asm("" : "+v"(col[i]));
            }
            if (GUARD) {
#pragma unroll
                for (int i = 0; i < CPP; ++i) {
                    uint32_t c = col[i];
                    c = (p0 + i < rem) ? c : wk.pad_e;
                    c = (p0 + i < skip) ? wk.pre_e : c;
                    col[i] = c;
                }
            }
            if (col_has_table_off<MODE, CW>()) {
#pragma unroll
                for (int i = 0; i < CPP; ++i) {
                    col[i] += wk.table_off;
                    asm("" : "+v"(col[i]));
                }
            }
        }
    } else {
#define NEEDLE_LOOKUP(D, K)                                                                         \
    col[(D) * 4 + (K)] = lookup<MODE, 1, GUARD, K>(wk, w[D], p0 + (D) * 4 + (K) < rem, p0 + (D) * 4 + (K) < skip);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            NEEDLE_LOOKUP(d, 0)
            NEEDLE_LOOKUP(d, 1)
            NEEDLE_LOOKUP(d, 2)
            NEEDLE_LOOKUP(d, 3)
        }
#undef NEEDLE_LOOKUP
    }
}

// One 16-byte piece of one row: w = its four dwords, p0 = index of its first char inside the tile (or row), rem /
// skip = GUARD: chars of the row from the tile start on / chars before the find() cursor, st = the automaton state
// (5 * id in packed mode), last_rel = OP_FIND: index + 1 of the last accepting char seen (same origin as p0).
// HIST (packed mode, find() on full rows): instead of "last_rel = accepted ? position : last_rel" -- two VALU ops per
// char in a walk that is VALU-issue bound -- every char's accept flag, which is bit 0 of the packed state (accepting
// states sit at odd field offsets), is shifted into acc_hist with ONE v_alignbit_b32; the caller turns the log into a
// position every 32 chars (char i of the 32 at bit i).
template <int OP, int CW, int MODE, bool GUARD, bool HIST = false>
__device__ __forceinline__ void walk_piece(const Walk &wk, const uint32_t (&w)[4], uint32_t p0, uint32_t rem, uint32_t skip,
                                           uint32_t accept_lo, uint32_t &st, int32_t &last_rel, uint32_t *acc_hist = nullptr) {
    constexpr int CPP = 16 / CW; // chars per 16-byte piece
    // Table modes are bound by LDS cycles, not by issue: a lane whose verdict is already final (sink, or
    // accepted for containedIn) is masked out of the piece's lookups, so its LDS passes and the bank
    // conflicts it would cause disappear.  (Packed mode is conflict-free by construction: no masking.)
    bool lane_live = true;
    if (MODE != MODE_PACK && NEEDLE_MASK_DONE_LANES)
        lane_live = (OP == OP_CONTAINED_IN) ? (st - 1u < accept_lo - 1u) : (st > wk.dead_hi);
    if (lane_live) {
    // all state-independent lookups of the piece first (they pipeline in the LDS) ...
    uint32_t col[CPP];
    piece_lookups<MODE, CW, GUARD>(wk, w, p0, rem, skip, col);
    // ... then ONE wait for all of them instead of one s_waitcnt per char (the walk is issue-bound), ...
    // (packed mode only: in the table and pair modes the same fence costs 5-8 %, their lookups are better left
    // interleaved with the dependent chain)
    if (MODE == MODE_PACK && NEEDLE_PIECE_FENCE) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
    }
    // ... then the dependent chain
    uint32_t lr = 0;                                          // OP_FIND: last accepting position inside this piece, + 1
    const uint32_t skip_rel = skip > p0 ? skip - p0 : 0u;     // GUARD: chars of this piece before the cursor
    if (MODE == MODE_PAIR) {
        uint32_t pair_e[CPP / 2]; // off the chain: column pair offsets
#pragma unroll
        for (int i = 0; i < CPP; i += 2) pair_e[i / 2] = col[i] + col[i + 1];
#pragma unroll
        for (int i = 0; i < CPP; i += 2) {
            const uint32_t e = lds_u16(__umul24(st, wk.ncols_e) + pair_e[i / 2] + kLdsPairTable1);
            st = e & 0xFFu;
            if (OP == OP_FIND) {
                const uint32_t code = e >> 8;            // 0 | 1: accepted after char i only | 2: after char i + 1
                const uint32_t pos = i + code;           // = index of the accepting char inside the piece + 1
                bool acc = code != 0u;
                if (GUARD) acc = acc && (pos > skip_rel); // an accepting start state must not count before the cursor
                lr = acc ? pos : lr;
            }
        }
    } else if (HIST) {
        uint32_t h = *acc_hist;
#pragma unroll
        for (int i = 0; i < CPP; ++i) {
            st = apply<MODE, CW>(wk, st, col[i]);
            h = __builtin_amdgcn_alignbit(st, h, 1); // ({st, h} >> 1): the accept flag enters at bit 31
        }
        *acc_hist = h;
        return;
    } else
#pragma unroll
    for (int i = 0; i < CPP; ++i) {
        if (MODE == MODE_SPARSE && GUARD) { // PAD / PRE are not columns of this mode: selected after the lookup
            uint32_t ns = apply<MODE, CW>(wk, st, col[i]);
            ns = (p0 + i < rem) ? ns : (wk.sp_pad_ident ? st : 0u);
            st = (p0 + i < skip) ? st : ns;
        } else
        st = apply<MODE, CW>(wk, st, col[i]);
        if (OP == OP_FIND) {
            bool acc = st >= accept_lo;
            if (GUARD) acc = acc && ((uint32_t)i >= skip_rel); // an accepting start state must not count before the cursor
            if (GUARD && MODE == MODE_SPARSE) acc = acc && (p0 + i < rem); // (lengths programs: the state FREEZES at the row's end)
            lr = acc ? (uint32_t)(i + 1) : lr;
        }
    }
    // (positions are tracked inside the piece, 1 .. 16: inline constants.  Tracking p0 + i + 1 directly made the
    // compiler keep 64 position literals in VGPRs across the unrolled tile loop -- 128 VGPRs, spills, and a scratch
    // reload whose s_waitcnt vmcnt(0) drained the prefetched tile loads.)
    if (OP == OP_FIND) last_rel = lr ? (int32_t)(p0 + lr) : last_rel;
    } // lane_live
}

// indexBackwards(en - 1, bound), DFAClassBuilder.java:536-583, for the lanes of `act`.  The row bytes [win_b0, win_b0 +
// win_bytes) are in LDS at win_addr (byte b at win_addr + ((b - win_b0) ^ swz16): the find-all tile is bank-swizzled -- SWZ --
// a plain window is not); anything else is read from memory.  At least 8 chars' worth of LDS precede win_addr (it lies behind
// the program image).  The backward automaton rides in the forward program's LDS part (packed functions, popcount-compressed
// rows, a small dense table) or is walked out of HBM / L2.
//
// The walk is LOCK-STEP: every lane still walking has taken the same number of steps, so step k of a round reads char
// idx0 - k for all of them and positions need no per-lane bookkeeping; a lane that is done carries state 0 (the sink: row 0 of
// every table, field 0 of every packed function, leads to 0), chars below `bound` (the loop bound `index >= FROM`, :549) lead
// there by a select on the lane's room.  Per step of the packed form: v_bfe, a compare + select for the bound, a compare +
// select for "accepting" -- 5 VALU ops where the per-lane (active, index, state, last) bookkeeping of rounds 1-4 took 12 and a
// branch around a memory fallback per char.
// A kernel-argument word read where it is used.  The compiler hoists plain `a.hdr.x` reads out of every loop and keeps them in SGPRs for the
// whole kernel; the backward walk's header words, needed once per group of rows, pushed the find() kernels past their 102 SGPRs (up to 52
// spilled into VGPR lanes, reloaded by v_readlane_b32 inside the forward walk's loop).  Read through the kernarg segment pointer and an
// address the optimiser cannot see through, the scalar loads stay where the walk is.  (Not through &a: an escaping address of the by-value
// argument makes the compiler keep a private copy of all 832 bytes in scratch.)  ScanArgs is the FIRST member of every kernel argument whose
// kernel walks backwards (scan_kernel, short_kernel: ScanArgs itself; find_all_kernel: FindAllArgs::s).
typedef __attribute__((address_space(4))) const char *KernargPtr;
__device__ __forceinline__ KernargPtr kernarg_here() {
    KernargPtr p = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
__device__ __forceinline__ uint32_t kernarg_u32(KernargPtr base, uint32_t byte_offset) {
    return *(__attribute__((address_space(4))) const uint32_t *)(base + byte_offset);
}
template <typename T>
__device__ __forceinline__ T *kernarg_ptr(KernargPtr base, uint32_t byte_offset) {
    typedef T *Ptr; // (a generic pointer stored in the kernarg segment)
    return *(__attribute__((address_space(4))) const Ptr *)(base + byte_offset);
}
#define NEEDLE_KA_HDR(f) kernarg_u32(ka, (uint32_t)(offsetof(ScanArgs, hdr) + offsetof(ProgHeader, f)))
#define NEEDLE_KA_BHDR(f) kernarg_u32(ka, (uint32_t)(offsetof(ScanArgs, bhdr) + offsetof(ProgHeader, f)))
struct BackwardHdr { // the backward automaton as the walk sees it
    uint32_t off_bpack, bacc, start, root_accepting, bcols, off_bcmap, off_bptab, off_bpages, off_bsp_bm, off_bsp_base, off_bsp_edges, off_btable, b_off_table;
};
__device__ __forceinline__ BackwardHdr backward_hdr() {
    const KernargPtr ka = kernarg_here();
    BackwardHdr h;
    h.off_bpack = NEEDLE_KA_HDR(off_bpack);
    h.bacc = h.off_bpack ? NEEDLE_KA_HDR(bpack_accept_off) : NEEDLE_KA_BHDR(accept_lo);
    h.start = h.off_bpack ? NEEDLE_KA_HDR(bpack_start_off) : NEEDLE_KA_BHDR(start);
    h.root_accepting = NEEDLE_KA_BHDR(root_accepting);
    h.bcols = NEEDLE_KA_BHDR(n_cols);
    h.off_bcmap = NEEDLE_KA_HDR(off_bcmap), h.off_bptab = NEEDLE_KA_HDR(off_bptab), h.off_bpages = NEEDLE_KA_HDR(off_bpages);
    h.off_bsp_bm = NEEDLE_KA_HDR(off_bsp_bm), h.off_bsp_base = NEEDLE_KA_HDR(off_bsp_base), h.off_bsp_edges = NEEDLE_KA_HDR(off_bsp_edges);
    h.off_btable = NEEDLE_KA_HDR(off_btable);
    h.b_off_table = NEEDLE_KA_BHDR(off_table);
    return h;
}

template <int CW, bool SWZ = false, int STEPS = 8> // STEPS chars per round (4: rows of one 16-byte piece -- their matches are short)
__device__ __forceinline__ int32_t backward_walk(const ScanArgs &a, bool act, int32_t en, int32_t bound, uint32_t win_addr, uint32_t win_b0,
                                                 uint32_t win_bytes, uint32_t swz16, const uint8_t *rowp) {
    const BackwardHdr h = backward_hdr();
    const uint16_t *gbt = (const uint16_t *)(a.bprog + h.b_off_table);
    const uint32_t bcols = h.bcols, bacc = h.bacc;
    int32_t idx0 = en - 1;                                     // the char step 0 of the round reads
    uint32_t bs = act ? h.start : 0u;
    int32_t lastb = h.root_accepting ? bound : INT_MAX;        // :543-547
    for (;;) {
        const int32_t room = idx0 - bound;                     // chars idx0 .. idx0 - room may be read
        const bool live = bs != 0u && room >= 0;
        if (__ballot(live) == 0ull) break;
        // ---- the round's text: chars idx0 - (STEPS - 1) .. idx0
        const int32_t rel0 = idx0 * CW - (int32_t)win_b0;      // window offset of char idx0
        const int32_t lo = idx0 - (STEPS - 1) > bound ? idx0 - (STEPS - 1) : bound; // the lowest char this lane can need
        const bool from_mem = live && ((uint32_t)rel0 >= win_bytes || lo * CW < (int32_t)win_b0);
        uint32_t cs[STEPS];
        if (!SWZ) {
            // one address per lane, the chars at immediate offsets (a lane outside its window reads the window's start: unused)
            // (a lane that is not walking reads LDS offset 0: all of them the same chars, so that their map / table lookups below are
            // one broadcast address instead of 64 scattered ones -- C5: 70 % of the lanes)
            uint32_t rd = win_addr + ((uint32_t)rel0 < win_bytes ? (uint32_t)rel0 : (uint32_t)(STEPS - 1) * CW) - (uint32_t)(STEPS - 1) * CW;
            rd = live ? rd : 0u;
#pragma unroll
            for (int k = 0; k < STEPS; ++k) cs[k] = (CW == 1) ? lds_u8(rd + (uint32_t)(STEPS - 1 - k)) : lds_u16(rd + (uint32_t)(STEPS - 1 - k) * 2u);
        } else {
#pragma unroll
            for (int k = 0; k < STEPS; ++k) {
                const uint32_t rel = (uint32_t)(rel0 - k * CW);
                const uint32_t ad = live ? win_addr + ((rel < win_bytes ? rel : 0u) ^ swz16) : 0u;
                cs[k] = (CW == 1) ? lds_u8(ad) : lds_u16(ad);
            }
        }
        if (__ballot(from_mem) != 0ull) { // text outside the window (rare): waited for inside the branch, as in walk_tile
            uint32_t m[8];
#pragma unroll
            for (int k = STEPS; k < 8; ++k) m[k] = 0;
#pragma unroll
            for (int k = 0; k < STEPS; ++k) {
                const int32_t p = idx0 - k;
                m[k] = cs[k];
                if (live && k <= room && (uint32_t)(p * CW - (int32_t)win_b0) >= win_bytes)
                    m[k] = (CW == 1) ? (uint32_t)rowp[p] : (uint32_t)((const uint16_t *)rowp)[p];
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]));
#pragma unroll
            for (int k = 0; k < STEPS; ++k) cs[k] = m[k];
        }
        uint32_t last_k = (uint32_t)STEPS; // the round's last accepting step (STEPS: none)
        if (h.off_bpack) { // wave-uniform: packed backward automaton -- 8 independent char -> F lookups, then the chain
            uint32_t fb[STEPS];
#pragma unroll
            for (int k = 0; k < STEPS; ++k) {
                if (CW == 1) {
                    fb[k] = lds_u32(h.off_bpack + (cs[k] << 2));
                } else {
                    const u32x2 pg = lds_u32x2(h.off_bpack + ((cs[k] >> 8) << 3));
                    fb[k] = lds_u32((((cs[k] & 255u) << 2) & pg[1]) | pg[0]); // absolute address (needle_lower.cpp)
                }
            }
#pragma unroll
            for (int k = 0; k < STEPS; ++k) {
                uint32_t nb = __builtin_amdgcn_ubfe(fb[k], bs, 5);
                nb = room >= k ? nb : 0u;
                last_k = nb >= bacc ? (uint32_t)k : last_k;
                bs = nb;
            }
        } else {
            // the backward automaton's char -> column maps, at absolute LDS addresses: all eight ahead of the dependent chain
            uint32_t col[STEPS];
#pragma unroll
            for (int k = 0; k < STEPS; ++k) {
                if (CW == 1) col[k] = lds_u8(h.off_bcmap + cs[k]);
                else col[k] = lds_u8(h.off_bpages + ((lds_u8(h.off_bptab + (cs[k] >> 8)) << 8) | (cs[k] & 255u)));
            }
#pragma unroll
            for (int k = 0; k < STEPS; ++k) {
                if (bs != 0u) { // (the lanes that are done stay out of the LDS)
                    uint32_t nb;
                    if (h.off_bsp_bm) { // wave-uniform: popcount-compressed rows in LDS (needle_device.h)
                        const uint32_t bm = lds_u32(h.off_bsp_bm + bs * 4u);
                        const uint32_t at = lds_u16(h.off_bsp_base + bs * 2u) + (uint32_t)__builtin_popcount(bm & ((1u << col[k]) - 1u));
                        const uint32_t tgt = lds_u16(h.off_bsp_edges + at * 2u);
                        nb = ((bm >> col[k]) & 1u) ? tgt : 0u;
                    } else if (h.off_btable) { // wave-uniform: small dense table in LDS
                        nb = lds_u16(h.off_btable + (bs * bcols + col[k]) * 2u);
                    } else { // dense table in HBM / L2 (waited for here: no vmcnt wait on the other paths)
                        nb = gbt[bs * bcols + col[k]];
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(nb));
                    }
                    nb = room >= k ? nb : 0u;
                    last_k = nb >= bacc ? (uint32_t)k : last_k;
                    bs = nb;
                }
            }
        }
        lastb = last_k < (uint32_t)STEPS ? idx0 - (int32_t)last_k : lastb;
        idx0 = idx0 - STEPS > -1 ? idx0 - STEPS : -1; // (lanes that are done do not run away below their rows)
    }
    return lastb;
}

// Host side: let kernel `fn` use the whole 160 KiB of LDS.  The attribute belongs to (function, device): it is set
// once per device a host thread launches `fn` on (a bit mask of devices already done, per instantiation).
inline hipError_t allow_full_lds(const void *fn, uint64_t &done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done_mask & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    if (e == hipSuccess) done_mask |= bit;
    return e;
}

} // namespace needle
