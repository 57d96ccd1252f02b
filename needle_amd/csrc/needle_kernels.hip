// Hand-written gfx950 (CDNA4) kernels for needle's DFA table-walk hot path.
//
// One haystack ("row") per lane, 64 rows per wavefront step, up to 16 wavefronts per workgroup sharing ONE LDS
// copy of the lowered automaton, one workgroup per CU, persistent over 64-row groups.
//
// Data movement per wave and step ("tile" = 64 rows x CHB bytes, CHB = 128 or 64):
//   HBM --global_load_dwordx4 (16 B/lane; CHB/16 adjacent lanes cover one contiguous CHB-byte piece of one row,
//        i.e. whole 128-B lines for CHB = 128: fully coalesced)--> VGPRs (the NEXT tile, prefetched while the
//        current one is walked) --ds_write_b128 (lane-linear, conflict-free)--> LDS tile
//        --ds_read_b128 (each lane its own row; the global SOURCE piece index is XOR-swizzled so that these
//        row-strided reads hit 16 distinct 16-B bank slots per 16-lane service group)--> per-char walk.
// Register staging (instead of global_load_lds DMA) is what lets a wave keep a full tile of HBM reads in flight
// while it walks the previous one: in-flight bytes are not capped by the LDS tile buffers.
//
// The loops restated here (reference: needle-compiler/src/main/java/com/justinblank/strings/
// DFAClassBuilder.java): matches() :892-910, containedIn() :1004-1022, indexForwards() :438-468,
// indexBackwards() :565-583, find() :629-657.  Dead state (-1), the `c > maxChar` exits and "index past the row
// length" are folded into the lowered tables on the host (needle_lower.cpp): sink state 0, OVER and PAD columns
// -- so the inner loops here are branch-free lookups.
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

// LDS reads at an ABSOLUTE LDS byte address through address-space-3 pointers.  The dynamic segment starts at LDS
// address 0 (this file declares no static __shared__; scan_kernel traps if that ever changes), so table offsets
// are plain immediates: going through `smem` costs a `v_add 0` (late-resolved symbol) per access, going through
// generic pointers a null-check v_cndmask on top.
#define NEEDLE_LDS(T) __attribute__((address_space(3))) const T *
__device__ __forceinline__ uint32_t lds_u8(uint32_t a) { return *(NEEDLE_LDS(uint8_t))(uintptr_t)(a); }
__device__ __forceinline__ uint32_t lds_u16(uint32_t a) { return *(NEEDLE_LDS(uint16_t))(uintptr_t)(a); }
__device__ __forceinline__ uint32_t lds_u32(uint32_t a) { return *(NEEDLE_LDS(uint32_t))(uintptr_t)(a); }

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
    uint32_t table_off; // char_width 2 table modes: LDS byte offset of the table
    uint32_t lane4;     // lane * 4 (byte 0 of the packed-mode F address)
    const uint16_t *gtable; // MODE_GLOBAL
};

template <int CW>
__device__ __forceinline__ uint32_t column_of(const uint8_t *cmap, const uint8_t *ptab, const uint8_t *pages, uint32_t c) {
    if (CW == 1) return cmap[c];
    return pages[((uint32_t)ptab[c >> 8] << 8) | (c & 255u)];
}

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
        else col = lds_u16(shl_byte<K>(w, 1) + kLdsCmap1);
    } else {
        const uint32_t pg = lds_u16(shl_byte<(2 * K + 1) & 3>(w, 1) + kLdsPtab2);  // page base = page * 256
        const uint32_t ce = lds_u8(or_byte<(2 * K) & 3>(pg, w) + (MODE == MODE_PACK ? kLdsPages2Pack : kLdsPages2Table));
        col = (MODE == MODE_PACK) ? lds_u32(ce + kLdsF2) : ce; // pages hold column * 4 (packed) | * element size
    }
    if (GUARD) {
        col = in_row ? col : wk.pad_e;
        col = before_cursor ? wk.pre_e : col;
    }
    return col;
}
// st: 5 * state in MODE_PACK (the bit offset of the state's field in F), the state id otherwise.
template <int MODE, int CW>
__device__ __forceinline__ uint32_t apply(const Walk &wk, uint32_t st, uint32_t col) {
    if (MODE == MODE_PACK) return __builtin_amdgcn_ubfe(col, st, 5);
    const uint32_t i = __umul24(st, wk.ncols_e) + col;
    if (MODE == MODE_GLOBAL) return wk.gtable[i];
    const uint32_t addr = i + (CW == 1 ? (uint32_t)kLdsTable1 : wk.table_off);
    return MODE == MODE_TABLE8 ? lds_u8(addr) : lds_u16(addr);
}

template <int OP, int CW, int MODE, bool GUARD, int CHB>
__global__ __launch_bounds__(kWavesPerBlock * 64) void scan_kernel(const ScanArgs a) {
    using G = Geom<CHB>;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_waves = blockDim.x >> 6; // 16, 12, 8 or 4: chosen by the launcher from the automaton's LDS footprint

    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();
    // ---- stage the automaton in LDS (once per workgroup)
    for (uint32_t i = tid * 16u; i < a.hdr.lds_bytes; i += blockDim.x * 16u)
        *(u32x4 *)(smem + i) = *(const u32x4 *)(a.prog + i);
    __syncthreads();

    Walk wk;
    constexpr uint32_t ELEM = (MODE == MODE_TABLE16) ? 2u : 1u;
    wk.ncols_e = a.hdr.n_cols * ELEM;
    wk.pad_e = (MODE == MODE_PACK) ? a.hdr.pad_f : a.hdr.pad_col * ELEM;
    wk.pre_e = (MODE == MODE_PACK) ? a.hdr.pre_f : (a.hdr.pad_col + 1u) * ELEM;
    wk.table_off = a.hdr.off_table;
    wk.lane4 = (uint32_t)(lane & 31) * 4u; // lanes l and l+32 are served in different LDS passes: 32 copies suffice
    wk.gtable = (const uint16_t *)(a.prog + a.hdr.off_table);
    constexpr uint32_t SCALE = (MODE == MODE_PACK) ? 5u : 1u; // state representation scale
    const uint32_t accept_lo = a.hdr.accept_lo * SCALE;
    const uint32_t start_state = a.hdr.start * SCALE;

    // ---- this wave's LDS tile
    Tile tile;
    {
        uint32_t base, row_stride;
        if (a.tiles_in_f_rows && wave < 4) { // rows in the upper 128 B of F rows wave*64 .. wave*64+63
            base = kLdsF1 + (uint32_t)wave * 64u * 256u + 128u;
            row_stride = 256u;
        } else {
            const uint32_t first = a.tiles_in_f_rows ? 4u : 0u;
            base = ((a.hdr.lds_bytes + 15u) & ~15u) + ((uint32_t)wave - first) * G::kTileBytes;
            row_stride = CHB;
        }
        tile.store_addr = base + (uint32_t)(lane / G::kPieces) * row_stride + (uint32_t)(lane % G::kPieces) * 16u;
        tile.store_step = G::kRowsPerInstr * row_stride;
        tile.row_addr = base + (uint32_t)lane * row_stride;
    }

    const uint64_t n_groups = (a.n_rows + 63) >> 6;
    const uint64_t wave_cnt = (uint64_t)gridDim.x * n_waves;
    uint64_t g = (uint64_t)blockIdx.x * n_waves + wave;
    if (g >= n_groups) return;

    // Global address of load j of a tile = uniform base (SGPRs: group start + chunk offset + j * rows-per-load *
    // stride) + one of TWO per-lane 32-bit offsets: with the XOR swizzle the piece index only depends on the parity
    // of j (CHB 128) or not on j at all (CHB 64).  No per-load 64-bit VALU math, 2 VGPRs of addressing state.
    const uint32_t q = (uint32_t)lane >> 4;
    const uint32_t p_in_row = (uint32_t)(lane % G::kPieces);
    const uint32_t row_in_instr = (uint32_t)(lane / G::kPieces);
    const uint32_t o_even = row_in_instr * (uint32_t)a.stride_bytes + 16u * (p_in_row ^ q);
    const uint32_t o_odd = CHB == 128 ? (row_in_instr * (uint32_t)a.stride_bytes + 16u * (p_in_row ^ q ^ 4u)) : o_even;
    const uint64_t load_step = (uint64_t)G::kRowsPerInstr * a.stride_bytes;

    // A "fetch unit" is NT consecutive tiles of one group: one tile of 128-byte pieces, or TWO tiles of 64-byte
    // pieces = the two halves of the same 128-byte lines, requested back to back.  L2 lines are 128 B and every miss
    // fetches the whole line, so asking for the second half one tile-walk later (by when the XCD has streamed its
    // whole 4 MiB L2 once) would fetch every line twice (measured: TCC_EA0_RDREQ_128B x 128 B = 2.18x the batch).
    constexpr int NT = (CHB == 64) ? 2 : 1;
    u32x4 R[NT][G::kInstrs];
    // Store tile T of the unit held in R to LDS; with do_fetch, re-issue the loads of ALL the unit's registers for
    // unit `unit` of group grp piece by piece (every register of the unit is free once its last tile is staged): the
    // wave keeps loads in flight at all times instead of draining to zero at every tile boundary.
    auto stage_and_fetch = [&](auto tc, bool do_fetch, uint64_t grp, uint32_t unit) __attribute__((always_inline)) {
        constexpr int T = decltype(tc)::value;
        if (!do_fetch) {
#pragma unroll
            for (int j = 0; j < G::kInstrs; ++j) store_piece(tile, j, R[T][j]);
            return;
        }
        const uint8_t *base = a.rows + (grp << 6) * a.stride_bytes + unit * (NT * CHB);
#pragma unroll
        for (int j = 0; j < G::kInstrs; ++j) {
            store_piece(tile, j, R[T][j]);
            asm volatile("" ::: "memory"); // keep store j ahead of load j (else all loads hoist: two tiles live)
#pragma unroll
            for (int t = 0; t < NT; ++t) R[t][j] = load_row16<NT == 1>(base + t * CHB + j * load_step + ((j & 1) ? o_odd : o_even));
            asm volatile("" ::: "memory");
        }
    };
    auto fetch = [&](uint64_t grp, uint32_t unit) __attribute__((always_inline)) { // plain (re)load of a unit, no staging
        const uint8_t *base = a.rows + (grp << 6) * a.stride_bytes + unit * (NT * CHB);
#pragma unroll
        for (int j = 0; j < G::kInstrs; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) R[t][j] = load_row16<NT == 1>(base + t * CHB + j * load_step + ((j & 1) ? o_odd : o_even));
    };
    // The last 64-row group may hold fewer than 64 rows and its last chunk may reach past the end of the buffer:
    // it is fetched with every clamp applied, by the one wave that owns it, outside the pipelined loop.
    auto fetch_clamped = [&](uint64_t grp, uint32_t chunk) __attribute__((always_inline)) {
        const uint32_t last_r = (uint32_t)(a.n_rows - 1 - (grp << 6));
        const uint32_t stride = (uint32_t)a.stride_bytes;
        const uint8_t *gbase = a.rows + (grp << 6) * a.stride_bytes;
#pragma unroll
        for (int j = 0; j < G::kInstrs; ++j) {
            uint32_t r = (uint32_t)(j * G::kRowsPerInstr) + row_in_instr;
            const uint32_t kk = p_in_row ^ (uint32_t)G::swz((int)r);
            r = r < last_r ? r : last_r;
            uint32_t pb = chunk * CHB + kk * 16u;     // byte offset of the piece inside its row
            if (pb + 16u > stride) pb = stride - 16u; // keep the 16-byte read inside the row (stride >= 16)
            R[0][j] = load_row16<false>(gbase + (r * stride + pb));
        }
    };

    // per-group state
    uint64_t my_row;
    bool row_ok;
    uint32_t len, n_chunks, st;
    int32_t last;
    int32_t cursor = 0;  // OP_FIND with per-row cursors: Matcher.nextStart (FROM of find(FROM, TO)); < 0 = exhausted
    bool dead = false;
    // OP_FIND with a backward automaton: the three dwords of row text ending with the one that holds the char at
    // lastMatch - 1 (copied out of the LDS tile while it is there), and the last two dwords of the previous tile.
    // indexBackwards then starts on registers: going back to the row in memory costs a second fetch of its 128-byte
    // line from HBM per matched row (C3: +1.28 GB on a 2.56 GB batch).
    constexpr int CPD = 4 / CW;            // chars per dword
    constexpr int HN = (CW == 1) ? 3 : 5; // snapshot dwords: >= 9 chars back from lastMatch - 1 at any alignment
    // (native vectors, constant indices after unrolling: plain arrays captured by the lambdas end up in scratch)
    u32x8 hist = {0, 0, 0, 0, 0, 0, 0, 0}; // [0 .. HN)
    u32x4 carry = {0, 0, 0, 0};            // carry[i]: dword (last - i) of the previous tile
    auto tile_dword = [&](uint32_t j) __attribute__((always_inline)) -> uint32_t {
        return lds_u32(tile.row_addr + ((((j >> 2) ^ (uint32_t)G::swz(lane))) << 4) + ((j & 3u) << 2));
    };
    auto begin_group = [&](uint64_t grp) __attribute__((always_inline)) {
        my_row = (grp << 6) + lane;
        row_ok = my_row < a.n_rows;
        len = 0;
        if (row_ok) len = a.lengths ? a.lengths[my_row] : a.row_len;
        const uint32_t max_len = GUARD ? wave_max(len) : a.row_len;
        n_chunks = (max_len * CW + CHB - 1) / CHB;
        if (n_chunks == 0) n_chunks = 1; // empty rows still take one (fully PAD-guarded) step
        st = start_state;
        if (GUARD && OP == OP_FIND && a.from) {
            cursor = row_ok ? a.from[my_row] : -1;
            dead = cursor < 0; // find(): `if nextStart == -1 return false`, DFAClassBuilder.java:629-630
            if (dead) cursor = 0;
        }
        last = -1; // OP_FIND: lastMatch of indexForwards
        // :356 literal 0, then the first loop iteration's wasAccepted check (:440) moves it to FROM if FROM < length
        if (OP == OP_FIND && a.hdr.root_accepting) last = ((uint32_t)cursor < len) ? cursor : 0;
    };

    // Walk the tile in LDS (chunk ck of the current group).  Returns true when no lane needs a further chunk.
    auto walk_tile = [&](uint32_t ck) __attribute__((always_inline)) -> bool {
        const uint32_t idx0 = ck * (CHB / CW);           // index of the tile's first char
        const uint32_t rem = len > idx0 ? len - idx0 : 0; // GUARD: chars of this row inside the tile and beyond
        const uint32_t skip = (uint32_t)cursor > idx0 ? (uint32_t)cursor - idx0 : 0; // GUARD: chars before the cursor
        int32_t last_rel = -1;                            // OP_FIND: last accepting position inside this tile
        // ragged rows keep more values live per char: unroll less there or it spills
        constexpr int kUnroll = GUARD ? (OP == OP_FIND ? 1 : 2) : G::kPieces;
        constexpr int CPP = 16 / CW; // chars per 16-byte piece
        u32x4 v = tile_piece<CHB>(tile, lane, 0);
#pragma unroll kUnroll
        for (int kk = 0; kk < G::kPieces; ++kk) {
            const uint32_t w[4] = {v[0], v[1], v[2], v[3]};
            if (kk + 1 < G::kPieces) v = tile_piece<CHB>(tile, lane, kk + 1); // next piece: its latency hides below
            const uint32_t p0 = kk * CPP;
            // Table modes are bound by LDS cycles, not by issue: a lane whose verdict is already final (sink, or
            // accepted for containedIn) is masked out of the piece's lookups, so its LDS passes and the bank
            // conflicts it would cause disappear.  (Packed mode is conflict-free by construction: no masking.)
            bool lane_live = true;
            if (MODE != MODE_PACK && NEEDLE_MASK_DONE_LANES)
                lane_live = (OP == OP_CONTAINED_IN) ? (st - 1u < accept_lo - 1u) : (st != 0u);
            if (lane_live) {
            // all state-independent lookups of the piece first (they pipeline in the LDS) ...
            uint32_t col[CPP];
            if (CW == 2) {
                // UTF-16: three dependent lookups per char (page table -> page -> F).  Issued as three batches of 8
                // with ONE wait between batches: left to the scheduler they come out as ~14 short waits per piece,
                // each exposing a full LDS round trip.
                constexpr uint32_t kPages = (MODE == MODE_PACK) ? kLdsPages2Pack : kLdsPages2Table;
                uint32_t pg[CPP], ce[CPP];
#define NEEDLE_PG(D, K) pg[(D) * 2 + (K)] = lds_u16(shl_byte<(2 * (K) + 1) & 3>(w[D], 1) + kLdsPtab2);
#define NEEDLE_CE(D, K) ce[(D) * 2 + (K)] = lds_u8(or_byte<(2 * (K)) & 3>(pg[(D) * 2 + (K)], w[D]) + kPages);
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
                lds_fence();
#undef NEEDLE_PG
#undef NEEDLE_CE
#pragma unroll
                for (int i = 0; i < CPP; ++i) {
                    uint32_t c = (MODE == MODE_PACK) ? lds_u32(ce[i] + kLdsF2) : ce[i]; // pages hold column * 4 | * element size
                    if (GUARD) {
                        c = (p0 + i < rem) ? c : wk.pad_e;
                        c = (p0 + i < skip) ? wk.pre_e : c;
                    }
                    col[i] = c;
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
            // ... then ONE wait for all of them instead of one s_waitcnt per char (the walk is issue-bound), ...
            if (MODE == MODE_PACK && NEEDLE_PIECE_FENCE) {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0)
                __builtin_amdgcn_sched_barrier(0);
            }
            // ... then the dependent chain
#pragma unroll
            for (int i = 0; i < CPP; ++i) {
                st = apply<MODE, CW>(wk, st, col[i]);
                if (OP == OP_FIND) {
                    bool acc = st >= accept_lo;
                    if (GUARD) acc = acc && (p0 + i >= skip); // an accepting start state must not count before the cursor
                    last_rel = acc ? (int32_t)(p0 + i + 1) : last_rel;
                }
            }
            } // lane_live
        }
        if (OP == OP_FIND) {
            if (a.fixed_len < 0) { // wave-uniform
                if (last_rel >= 0) {
                    const uint32_t de = (uint32_t)(last_rel - 1) / CPD; // tile dword holding the accepting char
#pragma unroll
                    for (int i = 0; i < HN; ++i) {
                        const uint32_t t = tile_dword(de >= (uint32_t)i ? de - i : 0);
                        const uint32_t back = (uint32_t)i - de - 1u; // when de < i: how far into the previous tile
                        uint32_t c = carry[0];
#pragma unroll
                        for (int k = 1; k < HN - 1; ++k) c = back == (uint32_t)k ? carry[k] : c; // back <= i - 1 <= HN - 2
                        hist[i] = de >= (uint32_t)i ? t : c;
                    }
                }
                const u32x4 tail = tile_piece<CHB>(tile, lane, G::kPieces - 1);
                carry = tail.wzyx;
            }
            last = last_rel >= 0 ? (int32_t)idx0 + last_rel : last;
        }
        // wave-uniform early exit: every lane has an absorbing verdict (sink, or accepted for containedIn)
        bool live;
        if (OP == OP_CONTAINED_IN) live = st < accept_lo;
        else live = st != 0;
        if (GUARD) live = live && (idx0 + CHB / CW < len);
        return __ballot(live) == 0ull;
    };

    // Verdicts of the finished group: bitmap word, and for find() the start index (DFAClassBuilder.java:640-656).
    auto finish_group = [&](uint64_t grp) __attribute__((always_inline)) {
        bool res;
        if (OP == OP_FIND) res = row_ok && !dead && (last >= 0);
        else res = row_ok && (st >= accept_lo);
        const uint64_t word = __ballot(res);
        if (lane == 0) a.bitmap[grp] = word;
        if (OP != OP_FIND) return;
        int32_t s = -1;
        const int32_t e = res ? last : -1;
        if (a.fixed_len >= 0) {
            s = res ? last - a.fixed_len : -1; // :640-646
        } else {
            // indexBackwards(end - 1, 0), :536-583.  Column map in LDS, row bytes (L2-hot) fetched 8 at a time,
            // backward table walked out of HBM/L2.
            const uint8_t *bcmap = smem + a.hdr.off_bcmap, *bptab = smem + a.hdr.off_bptab, *bpages = smem + a.hdr.off_bpages;
            const uint16_t *bt = a.hdr.off_btable ? (const uint16_t *)(smem + a.hdr.off_btable)
                                                   : (const uint16_t *)(a.bprog + a.bhdr.off_table);
            const uint32_t bcols = a.bhdr.n_cols, bacc = a.bhdr.accept_lo;
            const uint8_t *rowp = a.rows + (row_ok ? my_row : 0) * a.stride_bytes;
            int32_t idx_b = last - 1;
            const int32_t hist_dword = (last > 0 ? last - 1 : 0) / CPD; // row dword index of hist0
            uint32_t bs = a.bhdr.start;
            int32_t lastb = a.bhdr.root_accepting ? cursor : INT_MAX; // :543-547 (LENGTH var = FROM)
            bool active = res;
            while (__ballot(active) != 0ull) {
                uint32_t cs[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int32_t p = idx_b - k;
                    cs[k] = 0;
                    if (active && p >= cursor) {
                        const int32_t rel = hist_dword - p / CPD; // 0 .. HN-1: still inside the snapshot
                        if (rel < HN) {
                            uint32_t word = hist[0];
#pragma unroll
                            for (int i = 1; i < HN; ++i) word = rel == i ? hist[i] : word;
                            cs[k] = (word >> ((uint32_t)(p % CPD) * (8u * CW))) & (CW == 1 ? 0xFFu : 0xFFFFu);
                        } else {
                            cs[k] = (CW == 1) ? rowp[p] : ((const uint16_t *)rowp)[p]; // a match longer than the snapshot
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (active) {
                        if (idx_b < cursor) { // loop bound `index >= FROM`, :549
                            active = false;
                        } else {
                            const uint32_t col = column_of<CW>(bcmap, bptab, bpages, cs[k]);
                            bs = bt[bs * bcols + col];
                            if (bs == 0) {
                                active = false;
                            } else {
                                if (bs >= bacc) lastb = idx_b;
                                --idx_b;
                            }
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
    };

    // ---- pipelined main loop over every group whose unclamped unit reads provably stay inside the buffer: a unit
    // read of group g ends before (g + 1) * 64 * stride + NT * CHB, so all groups but the last are safe when rows are
    // at least that far apart, and a few more trailing groups are excluded for narrower rows
    uint64_t last_group = n_groups - 1; // first group handled by the clamped tail below
    {
        const uint64_t group_bytes = 64 * a.stride_bytes;
        const uint64_t safe = a.total_bytes >= (uint64_t)(NT * CHB) ? (a.total_bytes - NT * CHB) / group_bytes : 0;
        if (safe < last_group) last_group = safe;
    }
    if (g < last_group) {
        uint32_t ck = 0;
        uint32_t pred_exit = 0xFFFFFFFFu; // chunk after which the previous group left early (prefetch predictor)
        begin_group(g);
        fetch(g, 0);
        // One tile: stage it, prefetch, walk it.  Returns 0 = same group continues with the next tile, 1 = a new
        // group was begun (its unit 0 is in R or in flight), 2 = no safe group left for this wave.
        auto step = [&](auto tc) __attribute__((always_inline)) -> int {
            constexpr int T = decltype(tc)::value;
            // Prefetch while this tile is walked whenever the unit's registers are all free after staging it: at
            // the unit's last tile, at the group's last chunk, or where the previous group left early.
            const bool do_pf = (T == NT - 1) || (ck + 1 >= n_chunks) || (ck >= pred_exit);
            const bool pf_same = (T == NT - 1) && (ck + 1 < n_chunks) && (ck < pred_exit);
            const uint64_t pf_g = pf_same ? g : g + wave_cnt;
            stage_and_fetch(tc, do_pf && pf_g < last_group, pf_g, pf_same ? (ck + 1) / NT : 0u);
            asm volatile("" ::: "memory"); // keep the prefetch issued ahead of the walk
            const bool group_done = walk_tile(ck) || (ck + 1 >= n_chunks);
            if (!group_done) {
                if (do_pf && !pf_same) fetch(g, (ck + 1) / NT); // predicted an exit that did not happen
                ++ck;
                return 0;
            }
            finish_group(g);
            pred_exit = (ck + 1 < n_chunks) ? ck : 0xFFFFFFFFu;
            const uint64_t ng = g + wave_cnt;
            g = ng;
            if (ng >= last_group) return 2;
            if (!do_pf || pf_same) fetch(ng, 0); // nothing (or this group's next unit) was prefetched: (re)direct
            ck = 0;
            begin_group(g);
            return 1;
        };
        for (;;) {
            int r = step(std::integral_constant<int, 0>{});
            if (NT == 2 && r == 0) r = step(std::integral_constant<int, NT - 1>{});
            if (r == 2) break;
        }
    }
    // ---- the batch's last group(s): clamped loads, no pipelining (at most a couple of waves in the whole grid)
    for (; g < n_groups; g += wave_cnt) {
        begin_group(g);
        for (uint32_t ck = 0; ck < n_chunks; ++ck) {
            fetch_clamped(g, ck);
            stage_and_fetch(std::integral_constant<int, 0>{}, false, 0, 0);
            if (walk_tile(ck)) break;
        }
        finish_group(g);
    }
}

// ------------------------------------------------------------------------------------------------
// packed ("CSR") rows -> fixed-stride rows: one thread per 16-byte piece of the output.  Output stores are
// lane-linear 16-byte pieces (fully coalesced); the source of a piece starts at an arbitrary byte of the packed
// buffer, so it is read as five ALIGNED dwords and funnel-shifted (v_alignbyte) into place.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unpack_kernel(const uint8_t *__restrict__ data, const uint64_t *__restrict__ offsets,
                                                     uint64_t n_rows, uint32_t cw, uint8_t *__restrict__ out,
                                                     uint64_t stride_bytes, uint32_t *__restrict__ lengths,
                                                     int32_t *overflow) {
    const uint64_t ppr = stride_bytes >> 4; // pieces per row
    const uint64_t total = n_rows * ppr;
    const uint64_t end_bytes = offsets[n_rows] * cw;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t row = q / ppr;
        const uint32_t k = (uint32_t)(q - row * ppr);
        const uint64_t b0 = offsets[row] * cw;
        uint64_t len_b = offsets[row + 1] * cw - b0;
        if (len_b > stride_bytes) {
            len_b = stride_bytes;
            if (k == 0 && overflow) *overflow = 1;
        }
        if (k == 0) lengths[row] = (uint32_t)(len_b / cw);
        const uint64_t pos = (uint64_t)k * 16u; // byte position of this piece inside the row
        u32x4 v = {0, 0, 0, 0};
        if (pos < len_b) {
            const uint64_t src = b0 + pos;
            const uint64_t base = src & ~(uint64_t)3;
            const uint32_t sh = (uint32_t)(src & 3u);
            uint32_t d[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) d[i] = (base + 4u * i < end_bytes) ? *(const uint32_t *)(data + base + 4u * i) : 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
            const uint64_t valid = len_b - pos; // bytes of the row in this piece (>= 16: all of it)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint64_t lo = 4u * i;
                if (valid <= lo) v[i] = 0;
                else if (valid < lo + 4u) v[i] &= (1u << (8u * (uint32_t)(valid - lo))) - 1u;
            }
        }
        *(u32x4 *)(out + row * stride_bytes + pos) = v;
    }
}

hipError_t launch_unpack(const void *data, const uint64_t *offsets, uint64_t n_rows, uint32_t cw, void *out,
                         uint64_t stride_bytes, uint32_t *lengths, int32_t *overflow, int n_cus, hipStream_t stream) {
    const uint64_t total = n_rows * (stride_bytes >> 4);
    uint64_t blocks = (total + 255) / 256;
    const uint64_t cap = (uint64_t)(n_cus > 0 ? n_cus : 256) * 16;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint8_t *)data, offsets, n_rows, cw,
                       (uint8_t *)out, stride_bytes, lengths, overflow);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Long rows (packed mode): intra-row parallelism by function composition.
//
// One row per lane cannot use the chip when there are few, long rows.  With <= 6 device states the per-char
// transition function F[c] is one dword (5-bit fields); so is the function of ANY substring: compose(A then B)
// field i = B[A[i]].  A wave takes one 4 KiB stripe: each lane loads its own 64 contiguous bytes straight into
// registers (no LDS transposition needed: a lane's bytes are contiguous), walks them for ALL entry states at once
// (one v_bfe_u32 per state and char), and an in-wave ordered scan composes the 64 lane functions.
//   pass 1  stripe_kernel<CW, false>: every stripe's function                        -> fn[row][stripe]
//   prefix  stripe_prefix_kernel: per row, sequentially over its stripes: entry state of each stripe (in place),
//           final state -> matches()/containedIn() verdict
//   pass 2  stripe_kernel<CW, true>  (find): lane entry state = stripe entry state through the exclusive lane scan,
//           then the lane walks its bytes again tracking the last accepting position; lanes after the automaton died
//           enter in the sink and accept nothing, so lastMatch (DFAClassBuilder.java:438-468) is simply the MAX of the
//           accepting positions: atomicMax per row
//   start   backward_row_kernel: indexBackwards from lastMatch - 1, one lane per row
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kIdentFn = (0u << 0) | (5u << 5) | (10u << 10) | (15u << 15) | (20u << 20) | (25u << 25);

// B after A: field i of the result = B[A[i]]
__device__ __forceinline__ uint32_t compose_fn(uint32_t a, uint32_t b) {
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) r |= __builtin_amdgcn_ubfe(b, __builtin_amdgcn_ubfe(a, 5u * i, 5), 5) << (5u * i);
    return r;
}

// NS = device states incl. the sink (2..6).  The sink maps to itself under every char, so only states 1 .. NS-1 are
// tracked: NS - 1 v_bfe_u32 per char.
template <int CW, bool FIND, int NS>
__global__ __launch_bounds__(kWavesPerBlock * 64) void stripe_kernel(const StripeArgs a) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();
    for (uint32_t i = tid * 16u; i < a.hdr.lds_bytes; i += blockDim.x * 16u) *(u32x4 *)(smem + i) = *(const u32x4 *)(a.prog + i);
    __syncthreads();
    Walk wk;
    wk.ncols_e = 0, wk.pad_e = a.hdr.pad_f, wk.pre_e = a.hdr.pre_f, wk.table_off = 0, wk.gtable = nullptr;
    wk.lane4 = (uint32_t)(lane & 31) * 4u;
    constexpr int CPL = 64 / CW; // chars per lane and stripe
    const uint32_t accept_lo = a.hdr.accept_lo * 5u;
    const uint64_t total = a.n_rows * a.spr;
    for (uint64_t v = (uint64_t)blockIdx.x * kWavesPerBlock + wave; v < total; v += (uint64_t)gridDim.x * kWavesPerBlock) {
        const uint64_t row = v / a.spr;
        const uint32_t s = (uint32_t)(v - row * a.spr);
        const uint32_t len = a.lengths ? a.lengths[row] : a.row_len;
        const uint64_t first = (uint64_t)s * (kStripeBytes / CW) + (uint64_t)lane * CPL; // this lane's first char
        const uint32_t n_valid = first >= len ? 0u : (uint32_t)(len - first < (uint64_t)CPL ? len - first : CPL);
        uint32_t entry = 0; // FIND: 5 * state in which the row's automaton reaches this stripe
        if (FIND) entry = a.fn[v];
        if ((uint64_t)s * (kStripeBytes / CW) >= len || (FIND && entry == 0)) { // stripe past the row's end / automaton dead
            if (!FIND && lane == 0) a.fn[v] = kIdentFn;
            continue;
        }
        u32x4 d[4] = {};
        const uint64_t off = (uint64_t)s * kStripeBytes + (uint64_t)lane * 64u;
        if (n_valid) { // 64 B inside the row's stride (len <= stride, stride a multiple of 16 B)
            const uint8_t *p = a.rows + row * a.stride_bytes + off;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (off + 16u * j < a.stride_bytes) d[j] = *(const u32x4 *)(p + 16 * j);
        }
        // the per-char functions of this lane's chars (kept for the second walk of FIND)
        uint32_t g[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) g[i] = 5u * i; // fields NS .. 5 stay identity (never a real state)
        const bool full = __ballot(n_valid != (uint32_t)CPL) == 0ull;
        auto walk_all = [&](auto per_char) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t w[4] = {d[j][0], d[j][1], d[j][2], d[j][3]};
                uint32_t f[16 / CW];
#define NEEDLE_F(D, K) f[(D) * (4 / CW) + (K)] = lookup<MODE_PACK, CW, false, K>(wk, w[D], true, false);
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    NEEDLE_F(dd, 0)
                    NEEDLE_F(dd, 1)
                    if (CW == 1) {
                        NEEDLE_F(dd, 2)
                        NEEDLE_F(dd, 3)
                    }
                }
#undef NEEDLE_F
                lds_fence();
#pragma unroll
                for (int i = 0; i < 16 / CW; ++i) per_char(j * (16 / CW) + i, f[i]);
            }
        };
        if (full) {
            walk_all([&](int, uint32_t f) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 1; i < NS; ++i) g[i] = __builtin_amdgcn_ubfe(f, g[i], 5);
            });
        } else {
            walk_all([&](int c, uint32_t f) __attribute__((always_inline)) {
                f = (uint32_t)c < n_valid ? f : kIdentFn;
#pragma unroll
                for (int i = 1; i < NS; ++i) g[i] = __builtin_amdgcn_ubfe(f, g[i], 5);
            });
        }
        uint32_t fn = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) fn |= g[i] << (5u * i);
        // ordered inclusive scan over the lanes: after step d lane l holds the function of lanes l-2d+1 .. l
        uint32_t incl = fn;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const uint32_t left = (uint32_t)__shfl_up((int)incl, dlt);
            if (lane >= dlt) incl = compose_fn(left, incl);
        }
        if (!FIND) {
            if (lane == 63) a.fn[v] = incl;
            continue;
        }
        uint32_t excl = (uint32_t)__shfl_up((int)incl, 1);
        if (lane == 0) excl = kIdentFn;
        uint32_t st = __builtin_amdgcn_ubfe(excl, entry, 5);
        int32_t last_rel = -1;
        if (full) {
            walk_all([&](int c, uint32_t f) __attribute__((always_inline)) {
                st = __builtin_amdgcn_ubfe(f, st, 5);
                last_rel = st >= accept_lo ? c + 1 : last_rel;
            });
        } else {
            walk_all([&](int c, uint32_t f) __attribute__((always_inline)) {
                st = __builtin_amdgcn_ubfe((uint32_t)c < n_valid ? f : kIdentFn, st, 5);
                last_rel = (st >= accept_lo && (uint32_t)c < n_valid) ? c + 1 : last_rel;
            });
        }
        int32_t best = last_rel >= 0 ? (int32_t)first + last_rel : -1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int32_t t = __shfl_xor(best, o);
            best = t > best ? t : best;
        }
        if (lane == 0 && best >= 0) atomicMax(a.end + row, best);
    }
}

// fn[] -> entry states, and the verdicts.  One workgroup per row: thread t composes its contiguous chunk of stripe
// functions, an ordered workgroup scan (wave shuffles + 16 wave totals through LDS) gives every chunk its entry
// function, and a second pass over the chunk replaces each function by the stripe's entry state.  (One thread per
// row walking 262144 stripes of a 1 GiB row one dependent load at a time took 25 ms.)
__global__ __launch_bounds__(1024) void stripe_prefix_kernel(const StripeArgs a) {
    __shared__ uint32_t wave_total[16];
    const uint32_t accept_lo = a.hdr.accept_lo * 5u;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = (blockDim.x + 63) >> 6;
    const uint32_t per = (a.spr + blockDim.x - 1) / blockDim.x; // stripes per thread
    for (uint64_t row = blockIdx.x; row < a.n_rows; row += gridDim.x) {
        uint32_t *fn = a.fn + row * a.spr;
        const uint32_t s0 = (uint32_t)tid * per < a.spr ? (uint32_t)tid * per : a.spr;
        const uint32_t s1 = s0 + per < a.spr ? s0 + per : a.spr;
        uint32_t mine = kIdentFn;
        for (uint32_t s = s0; s < s1; ++s) mine = compose_fn(mine, fn[s]);
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t left = (uint32_t)__shfl_up((int)incl, d);
            if (lane >= d) incl = compose_fn(left, incl);
        }
        if (lane == 63) wave_total[wave] = incl;
        __syncthreads();
        uint32_t before = kIdentFn; // everything in the waves before this one
        for (int w = 0; w < wave; ++w) before = compose_fn(before, wave_total[w]);
        uint32_t excl = (uint32_t)__shfl_up((int)incl, 1);
        if (lane == 0) excl = kIdentFn;
        excl = compose_fn(before, excl);
        uint32_t q = __builtin_amdgcn_ubfe(excl, a.hdr.start * 5u, 5);
        for (uint32_t s = s0; s < s1; ++s) {
            const uint32_t f = fn[s];
            fn[s] = q;
            q = __builtin_amdgcn_ubfe(f, q, 5);
        }
        if (tid == (int)blockDim.x - 1) { // its chunk is the last one (possibly empty): q is the row's final state
            if (a.op == OP_FIND) a.end[row] = a.hdr.root_accepting ? 0 : -1; // :356 lastMatch before the first char
            else if (q >= accept_lo) atomicOr((unsigned long long *)(a.bitmap + (row >> 6)), 1ull << (row & 63));
        }
        __syncthreads(); // wave_total is reused by the next row
    }
    (void)n_waves;
}

// find(): matched bit + start (DFAClassBuilder.java:640-656) for the long-row path; one lane per row, everything
// read from global memory (the rows here are few).
template <int CW>
__global__ __launch_bounds__(256) void backward_row_kernel(const StripeArgs a) {
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < a.n_rows; base += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t row = base + threadIdx.x;
        bool res = false;
        if (row < a.n_rows) {
            const int32_t last = a.end[row];
            res = last >= 0;
            int32_t s = -1;
            if (res) {
                if (a.fixed_len >= 0) {
                    s = last - a.fixed_len;
                } else {
                    const uint8_t *bcmap = a.prog + a.hdr.off_bcmap, *bptab = a.prog + a.hdr.off_bptab, *bpages = a.prog + a.hdr.off_bpages;
                    const uint16_t *bt = (const uint16_t *)(a.bprog + a.bhdr.off_table);
                    const uint8_t *rowp = a.rows + row * a.stride_bytes;
                    uint32_t bs = a.bhdr.start;
                    int32_t lastb = a.bhdr.root_accepting ? 0 : INT_MAX;
                    for (int32_t p = last - 1; p >= 0; --p) {
                        const uint32_t c = (CW == 1) ? rowp[p] : ((const uint16_t *)rowp)[p];
                        bs = bt[bs * a.bhdr.n_cols + column_of<CW>(bcmap, bptab, bpages, c)];
                        if (bs == 0) break;
                        if (bs >= a.bhdr.accept_lo) lastb = p;
                    }
                    s = lastb;
                }
            }
            a.start[row] = s;
            if (!res) a.end[row] = -1;
        }
        const uint64_t word = __ballot(res);
        if ((threadIdx.x & 63) == 0 && row < a.n_rows) a.bitmap[row >> 6] = word;
    }
}

template <int CW, bool FIND, int NS>
static hipError_t launch_stripe(const StripeArgs &a, dim3 grid, size_t lds, hipStream_t stream) {
    auto k = stripe_kernel<CW, FIND, NS>;
    static thread_local bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) return e;
        configured = true;
    }
    hipLaunchKernelGGL(k, grid, dim3(kWavesPerBlock * 64), lds, stream, a);
    return hipGetLastError();
}
template <int CW, bool FIND>
static hipError_t launch_stripe_n(const StripeArgs &a, dim3 grid, size_t lds, hipStream_t stream) {
    switch (a.hdr.n_states) {
    case 0: case 1: case 2: return launch_stripe<CW, FIND, 2>(a, grid, lds, stream);
    case 3: return launch_stripe<CW, FIND, 3>(a, grid, lds, stream);
    case 4: return launch_stripe<CW, FIND, 4>(a, grid, lds, stream);
    case 5: return launch_stripe<CW, FIND, 5>(a, grid, lds, stream);
    default: return launch_stripe<CW, FIND, 6>(a, grid, lds, stream);
    }
}

hipError_t launch_long_rows(int char_width, const StripeArgs &a, int n_cus, hipStream_t stream) {
    const uint64_t total = a.n_rows * a.spr;
    uint64_t blocks = (total + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks > (uint64_t)n_cus) blocks = (uint64_t)n_cus;
    const size_t lds = (a.hdr.lds_bytes + 15u) & ~15u;
    const dim3 grid((unsigned)blocks);
    const unsigned pblocks = (unsigned)((a.n_rows + 255) / 256);
    hipError_t e = char_width == 1 ? launch_stripe_n<1, false>(a, grid, lds, stream) : launch_stripe_n<2, false>(a, grid, lds, stream);
    if (e != hipSuccess) return e;
    {
        if (a.op != OP_FIND) { // verdict bits are OR-ed in
            e = hipMemsetAsync(a.bitmap, 0, ((a.n_rows + 63) / 64) * 8, stream);
            if (e != hipSuccess) return e;
        }
        unsigned threads = 64;
        while (threads < 1024 && threads < a.spr) threads <<= 1;
        const uint64_t max_blocks = (uint64_t)n_cus * (2048 / threads);
        hipLaunchKernelGGL(stripe_prefix_kernel, dim3((unsigned)(a.n_rows < max_blocks ? a.n_rows : max_blocks)), dim3(threads), 0, stream, a);
    }
    if (a.op == OP_FIND) {
        e = char_width == 1 ? launch_stripe_n<1, true>(a, grid, lds, stream) : launch_stripe_n<2, true>(a, grid, lds, stream);
        if (e != hipSuccess) return e;
        if (char_width == 1) hipLaunchKernelGGL(backward_row_kernel<1>, dim3(pblocks), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(backward_row_kernel<2>, dim3(pblocks), dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------------
struct LaunchShape {
    int grid, waves, chb;
    size_t lds;
};

template <int OP, int CW, int MODE, bool GUARD, int CHB>
static hipError_t launch_one(const ScanArgs &a, LaunchShape sh, hipStream_t stream) {
    auto k = scan_kernel<OP, CW, MODE, GUARD, CHB>;
    static thread_local bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) return e;
        configured = true;
    }
    hipLaunchKernelGGL(k, dim3(sh.grid), dim3(sh.waves * 64), sh.lds, stream, a);
    return hipGetLastError();
}

template <int OP, int CW, int MODE, bool GUARD>
static hipError_t launch_h(const ScanArgs &a, LaunchShape sh, hipStream_t s) {
    return sh.chb == 128 ? launch_one<OP, CW, MODE, GUARD, 128>(a, sh, s) : launch_one<OP, CW, MODE, GUARD, 64>(a, sh, s);
}

template <int OP, int CW, int MODE>
static hipError_t launch_g(const ScanArgs &a, bool guard, LaunchShape sh, hipStream_t s) {
    return guard ? launch_h<OP, CW, MODE, true>(a, sh, s) : launch_h<OP, CW, MODE, false>(a, sh, s);
}

template <int OP, int CW>
static hipError_t launch_m(const ScanArgs &a, bool guard, LaunchShape sh, hipStream_t s) {
    switch (a.hdr.mode) {
    case MODE_PACK: return launch_g<OP, CW, MODE_PACK>(a, guard, sh, s);
    case MODE_TABLE8: return launch_g<OP, CW, MODE_TABLE8>(a, guard, sh, s);
    case MODE_TABLE16: return launch_g<OP, CW, MODE_TABLE16>(a, guard, sh, s);
    default: return launch_g<OP, CW, MODE_GLOBAL>(a, guard, sh, s);
    }
}

template <int OP>
static hipError_t launch_c(const ScanArgs &a, int cw, bool guard, LaunchShape sh, hipStream_t s) {
    return cw == 1 ? launch_m<OP, 1>(a, guard, sh, s) : launch_m<OP, 2>(a, guard, sh, s);
}

// Workgroup shape from the automaton's LDS footprint: keep 16 waves per CU (the latency-hiding budget) as long as
// possible.  Packed mode on 8-bit rows is special: its F table is 256 rows x 256 B of which only the lower 128 B
// (32 lane-bank copies) are used, so the first 4 waves keep their tiles in the upper halves of those rows and the
// whole 160 KiB holds F + 16 tiles of 8 KiB.
bool shape_for_program(const ProgHeader &h, int char_width, int *waves, int *chb, int *tiles_in_f_rows) {
    const size_t p = (h.lds_bytes + 15u) & ~15u;
    const size_t cap = 160u * 1024u;
    *tiles_in_f_rows = 0;
    static const char *force = getenv("NEEDLE_SHAPE"); // e.g. "16x64" (tuning experiments only)
    if (force) {
        int w = 0, c = 0;
        if (sscanf(force, "%dx%d", &w, &c) == 2 && (c == 64 || c == 128) && w >= 1 && w <= 16 && p + (size_t)w * 64 * c <= cap) {
            *waves = w;
            *chb = c;
            return true;
        }
    }
    if (h.mode == MODE_PACK && char_width == 1 && p + 12u * 8192u <= cap) {
        *waves = 16;
        *chb = 128;
        *tiles_in_f_rows = 1;
        return true;
    }
    static const int cand[6][2] = {{16, 128}, {12, 128}, {16, 64}, {12, 64}, {8, 64}, {4, 64}};
    for (const auto &c : cand) {
        if (p + (size_t)c[0] * 64 * c[1] <= cap) {
            *waves = c[0];
            *chb = c[1];
            return true;
        }
    }
    return false;
}

hipError_t launch_scan(int op, int char_width, const ScanArgs &a_in, int n_cus, hipStream_t stream) {
    if (a_in.n_rows == 0) return hipSuccess;
    ScanArgs a = a_in;
    LaunchShape sh;
    int in_f = 0;
    if (!shape_for_program(a.hdr, char_width, &sh.waves, &sh.chb, &in_f)) return hipErrorInvalidValue;
    a.tiles_in_f_rows = (uint32_t)in_f;
    const uint64_t n_groups = (a.n_rows + 63) >> 6;
    uint64_t blocks = (n_groups + sh.waves - 1) / sh.waves;
    // One persistent workgroup per CU owns the whole CU (LDS); NEEDLE_RESERVE_CUS=k leaves k CUs to kernels that
    // must run CONCURRENTLY (the RCCL gather of the previous step's bitmap): otherwise that kernel steals a CU
    // from a statically partitioned launch and the whole step finishes late.
    static const int reserve = getenv("NEEDLE_RESERVE_CUS") ? atoi(getenv("NEEDLE_RESERVE_CUS")) : 0;
    if (reserve > 0 && n_cus > reserve + 8) n_cus -= reserve;
    if (blocks > (uint64_t)n_cus) blocks = (uint64_t)n_cus;
    sh.grid = (int)blocks;
    sh.lds = ((a.hdr.lds_bytes + 15u) & ~15u) + (size_t)(sh.waves - (in_f ? 4 : 0)) * 64 * sh.chb;
    // unguarded kernels assume every row fills a whole number of tiles
    const bool guard = a.lengths != nullptr || a.from != nullptr || a.row_len == 0 ||
                       ((uint64_t)a.row_len * char_width) % sh.chb != 0;
    switch (op) {
    case OP_MATCHES: return launch_c<OP_MATCHES>(a, char_width, guard, sh, stream);
    case OP_CONTAINED_IN: return launch_c<OP_CONTAINED_IN>(a, char_width, guard, sh, stream);
    default: return launch_c<OP_FIND>(a, char_width, guard, sh, stream);
    }
}

} // namespace needle
