// needle_scan.h -- the tiled scan kernel of libneedle_hip.so (hand-written for gfx950 / CDNA4) and its launch
// templates.  Included by one translation unit per reference loop (needle_scan_*.hip), which compile in parallel.
//
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
// while it walks the previous one: in-flight bytes are not capped by the LDS tile buffers.  Every 128-byte line is
// requested exactly once: by one `nt` load instruction (CHB = 128) or by the two halves of a fetch unit issued back to
// back (CHB = 64).  The per-char code itself (all automaton modes) is walk_piece() in needle_walk.h.
//
// Survivor pool.  Rows are independent, but a wave walks 64 of them in lockstep: once most of its rows have their
// verdict (find() after the match, matches() after the first mismatch, containedIn() after the first hit) every
// further tile costs the full instruction stream for a handful of live lanes.  So after each 128-byte line a group
// whose unresolved rows are few hands them over -- row, next chunk, automaton state, lastMatch: four registers --
// to free lanes of the wave's POOL (a cross-lane compaction with ds_permute / ds_bpermute, no memory involved) and
// ends.  When the pool is nearly full the wave runs a pool step: one more line of up to 64 pooled rows, gathered by
// per-lane row addresses into the same LDS tile and walked by the same code, each lane from its own row offset;
// rows that resolve are written out (bitmap bit by atomicOr into the word their group already stored), the others
// stay pooled.  Resolved rows' remaining lines are never fetched.
//
// The loops restated here (reference: needle-compiler/src/main/java/com/justinblank/strings/
// DFAClassBuilder.java): matches() :892-910, containedIn() :1004-1022, indexForwards() :438-468,
// indexBackwards() :565-583, find() :629-657.  Dead state (-1), the `c > maxChar` exits and "index past the row
// length" are folded into the lowered tables on the host (needle_lower.cpp): sink state 0, OVER and PAD columns
// -- so the inner loops here are branch-free lookups.
#pragma once
#include "needle_walk.h"

namespace needle {

// LEN (OP_FIND on a "lengths" program, needle_lower.h): start = end - pend[stop state] -- the instantiation carries no text
// snapshots and no backward walk at all (in the ragged-row kernels those cost registers the walk then spills: C3 find on
// ragged rows 0.66 -> 0.53 ms)
template <int OP, int CW, int MODE, bool GUARD, int CHB, bool LEN = false>
__global__ __launch_bounds__(kWavesPerBlock * 64) void scan_kernel(const ScanArgs a) {
    using G = Geom<CHB>;
    // The survivor pool is compiled into the kernels of the big-table automata (64-byte tiles: the shape the launcher
    // picks when the table fills the LDS).  Packed mode has no use for it (its lanes cost the same dead or alive and its
    // scans run at the HBM rate), and the 128-byte-tile kernels have no registers to spare (128 VGPRs at 16 waves).
    constexpr bool POOL = MODE != MODE_PACK && CHB == 64;
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
    constexpr uint32_t ELEM = (MODE == MODE_TABLE16 || MODE == MODE_HYBRID) ? 2u : 1u;
    wk.ncols_e = a.hdr.n_cols * ELEM;
    wk.pad_e = (MODE == MODE_PACK) ? a.hdr.pad_f : a.hdr.pad_col * ELEM;
    wk.pre_e = (MODE == MODE_PACK) ? a.hdr.pre_f : (a.hdr.pad_col + 1u) * ELEM;
    wk.pad_b = wk.pre_b = 0;
    if (MODE == MODE_PAIR) { // row stride of [state][col1][col2] uint16; columns premultiplied for either position
        wk.ncols_e = a.hdr.n_cols * a.hdr.n_cols * 2u;
        wk.pad_e = a.hdr.pad_col * a.hdr.n_cols * 2u;
        wk.pre_e = (a.hdr.pad_col + 1u) * a.hdr.n_cols * 2u;
        wk.pad_b = a.hdr.pad_col * 2u;
        wk.pre_b = (a.hdr.pad_col + 1u) * 2u;
    }
    if (MODE == MODE_SPARSE) wk.pad_e = wk.pre_e = a.hdr.win_lo_e; // (any valid column: the guarded walk overrides the result, needle_walk.h)
    // window addressing (needle_device.h): the column offset is the clamped char itself; the table sits win_lo bytes further up
    wk.win_on = a.hdr.win_on;
    wk.win_lo = a.hdr.win_lo_e;
    wk.win_hi = a.hdr.win_hi_e;
    wk.dead_hi = OP == OP_FIND ? a.hdr.fa_dead_hi : 0u; // (the D_L states follow the sink: needle_device.h)
    wk.sp_chains = a.hdr.sp_chains;
    wk.sp_pad_ident = a.hdr.sp_pad_ident;
    wk.flat = (CW == 2 && (MODE == MODE_TABLE8 || MODE == MODE_TABLE16)) ? a.hdr.flat_pages : 0u;
    // (window addressing: column offsets are not rebased -- wraps are fine in 32-bit address math; the compressed form carries
    // the bias inside its image)
    wk.table_off = a.hdr.off_table - (MODE == MODE_SPARSE ? 0u : a.hdr.win_lo_e);
    wk.lane4 = (uint32_t)(lane & 31) * 4u; // lanes l and l+32 are served in different LDS passes: 32 copies suffice
    wk.gtable = (const uint16_t *)(a.prog + (MODE == MODE_HYBRID ? a.hdr.off_gtable : a.hdr.off_table));
    if (MODE == MODE_HYBRID) wk.gtable = (const uint16_t *)((const uint8_t *)wk.gtable - a.hdr.win_lo_e);
    wk.hot_last = a.hdr.hot_bytes - 2u + a.hdr.win_lo_e;
    // packed mode: a state is the bit offset of its field in F (needle_device.h)
    const uint32_t accept_lo = MODE == MODE_PACK ? a.hdr.accept_off : a.hdr.accept_lo;
    const uint32_t start_state = MODE == MODE_PACK ? a.hdr.start_off : a.hdr.start;

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
    // OP_FIND with a backward automaton: a snapshot of the row text around lastMatch, copied out of the LDS tile while
    // it is there: snapA = the 16-byte piece holding the char at lastMatch - 1, snapB = the piece before it (the
    // previous tile's last piece -- `carry` -- when snapA is a tile's first).  indexBackwards then starts on registers
    // (16 .. 31 bytes of text): going back to the row in memory costs a second fetch of its 128-byte line from HBM per
    // matched row (C3: +1.28 GB on a 2.56 GB batch), and a per-lane load in the walk has to be waited for behind the
    // tile prefetch (without `carry`, i.e. 5 % of the matched rows reading memory: C3 0.53 -> 0.58 ms).  Longer matches
    // read on from memory.
    // (native vectors: plain arrays captured by the lambdas end up in scratch)
    u32x4 snapA = {0, 0, 0, 0}, snapB = {0, 0, 0, 0}, carry = {0, 0, 0, 0};
    int32_t snap_pi = 0; // index of snapA's piece inside its row (16-byte units)
    bool snapB_ok = true, carry_ok = true; // (pool steps: the previous tile in LDS belonged to other rows)
    // survivor pool: lane l's slot holds one unresolved row when bit l of pool_mask is set
    uint32_t p_row = 0, p_ck = 0, p_st = 0;
    int32_t p_last = -1;
    uint64_t pool_mask = 0; // wave-uniform
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
            if (dead) cursor = 0, st = 0; // parked in the sink: no lookups, and an all-exhausted wave leaves after one tile
        }
        snapB_ok = carry_ok = true;
        last = -1; // OP_FIND: lastMatch of indexForwards
        // :356 literal 0, then the first loop iteration's wasAccepted check (:440) moves it to FROM if FROM < length
        if (OP == OP_FIND && a.hdr.root_accepting) last = ((uint32_t)cursor < len) ? cursor : 0;
    };

    // Walk the tile in LDS: per lane the chars [idx0, idx0 + CHB / CW) of its row (piece0 = idx0 in 16-byte pieces;
    // both wave-uniform for a group walked in place, per lane in a pool step).  Returns the mask of lanes that need a
    // further chunk.
    auto walk_tile = [&](uint32_t idx0, uint32_t piece0) __attribute__((always_inline)) -> uint64_t {
        const uint32_t rem = len > idx0 ? len - idx0 : 0; // GUARD: chars of this row inside the tile and beyond
        const uint32_t skip = (uint32_t)cursor > idx0 ? (uint32_t)cursor - idx0 : 0; // GUARD: chars before the cursor
        int32_t last_rel = -1;                            // OP_FIND: last accepting position inside this tile
        // ragged rows keep more values live per char: unroll less there or it spills.  The big-table modes (compressed automaton,
        // hot rows) are 70 .. 136 KB of ISA per instantiation this way (scripts/kernel_code_size.py) -- more than the 64 KB
        // instruction cache a CU pair shares.  NEEDLE_BIG_ROLLED=1 keeps their piece loops rolled (23 .. 84 KB); measured on the
        // C3-sparse walk (profiles/r04_code_diet.md): SQC_ICACHE_MISSES is ~2 .. 5 thousand of 1.8e8 requests per launch in
        // BOTH forms -- the pipelined loop's hot part fits, the rest is never fetched -- and the rolled form is 2-3 % slower
        // (loop overhead on the chain).  So the unrolled form stays.
        constexpr bool BIG = NEEDLE_BIG_ROLLED && (MODE == MODE_SPARSE || MODE == MODE_HYBRID);
        constexpr int kUnroll = BIG ? 1 : GUARD ? (OP == OP_FIND ? 1 : 2) : G::kPieces;
        constexpr int kUnrollSplit = BIG ? 1 : G::kPieces;
        constexpr int CPP = 16 / CW; // chars per 16-byte piece
        u32x4 v = tile_piece<CHB>(tile, lane, 0);
        if (GUARD && NEEDLE_SPLIT_BOUNDARY) {
            // Ragged rows / per-row cursors: a row has at most ONE piece that its length cuts and ONE that its find()
            // cursor cuts.  Pieces wholly between the two run the unguarded code under an exec mask; pieces wholly
            // before the cursor or past the length are skipped (chars there can only park the automaton: PRE is
            // identity, PAD identity or the sink); the cut pieces -- different ones per lane -- are walked with the
            // per-char guards, the cursor's before the loop and the length's after it.
            const uint32_t n_in = rem / CPP;                  // pieces of this tile wholly inside the row (may exceed kPieces)
            const uint32_t first_in = (skip + CPP - 1) / CPP; // first piece wholly at or after the cursor
            const bool cursor_cut = (skip % CPP != 0) && (skip / CPP < (uint32_t)G::kPieces);
            if (cursor_cut) {
                const uint32_t cp = skip / CPP;
                const u32x4 c = tile_piece<CHB>(tile, lane, (int)cp);
                const uint32_t w[4] = {c[0], c[1], c[2], c[3]};
                walk_piece<OP, CW, MODE, true>(wk, w, cp * CPP, rem, skip, accept_lo, st, last_rel);
            }
#pragma unroll kUnrollSplit
            for (int kk = 0; kk < G::kPieces; ++kk) {
                const uint32_t w[4] = {v[0], v[1], v[2], v[3]};
                if (kk + 1 < G::kPieces) v = tile_piece<CHB>(tile, lane, kk + 1);
                if ((uint32_t)kk >= first_in && (uint32_t)kk < n_in)
                    walk_piece<OP, CW, MODE, false>(wk, w, kk * CPP, 0, 0, accept_lo, st, last_rel);
            }
            if (n_in < (uint32_t)G::kPieces && rem % CPP != 0 && !(cursor_cut && skip / CPP == n_in)) {
                const u32x4 b = tile_piece<CHB>(tile, lane, (int)n_in);
                const uint32_t w[4] = {b[0], b[1], b[2], b[3]};
                walk_piece<OP, CW, MODE, true>(wk, w, n_in * CPP, rem, skip, accept_lo, st, last_rel);
            }
        } else {
        constexpr bool HIST = OP == OP_FIND && MODE == MODE_PACK && !GUARD; // accept flags logged, not selected (walk_piece)
        uint32_t acc_hist = 0;
#pragma unroll kUnroll
        for (int kk = 0; kk < G::kPieces; ++kk) {
            const uint32_t w[4] = {v[0], v[1], v[2], v[3]};
            if (kk + 1 < G::kPieces) v = tile_piece<CHB>(tile, lane, kk + 1); // next piece: its latency hides below
            const uint32_t p0 = kk * CPP;
            walk_piece<OP, CW, MODE, GUARD, HIST>(wk, w, p0, rem, skip, accept_lo, st, last_rel, &acc_hist);
            if (HIST && ((kk + 1) * CPP) % 32 == 0) {
                // 32 chars logged: char i of them at bit i.  The last accepting one, if any, is the highest set bit.
                const int32_t top = 31 - (int32_t)__builtin_clz(acc_hist | 1u);
                last_rel = acc_hist ? (int32_t)((kk + 1) * CPP - 32) + top + 1 : last_rel;
                acc_hist = 0;
            }
        }
        }
        if (OP == OP_FIND) {
            if (a.fixed_len < 0 && !LEN && !a.hdr.fa_len_off) { // wave-uniform (a "lengths" program needs no text for its starts)
                if (last_rel >= 0) {
                    const uint32_t pi = ((uint32_t)(last_rel - 1) * CW) >> 4; // tile piece holding the accepting char
                    snapA = tile_piece<CHB>(tile, lane, (int)pi);
                    const u32x4 before = tile_piece<CHB>(tile, lane, (int)(pi ? pi - 1 : 0));
                    snapB = pi ? before : carry;
                    snapB_ok = pi != 0u || carry_ok;
                    snap_pi = (int32_t)(piece0 + pi);
                }
                carry = tile_piece<CHB>(tile, lane, G::kPieces - 1);
            }
            last = last_rel >= 0 ? (int32_t)idx0 + last_rel : last;
        }
        // wave-uniform early exit: every lane has an absorbing verdict (sink, or accepted for containedIn)
        bool live;
        if (OP == OP_CONTAINED_IN) live = st < accept_lo;
        else live = st > wk.dead_hi;
        if (GUARD) live = live && (idx0 + CHB / CW < len);
        return __ballot(live);
    };

    // Verdicts of the rows with row_ok: bitmap, and for find() the start index (DFAClassBuilder.java:640-656).  A group
    // finished in place stores its whole bitmap word (rows it deferred to the pool count as 0 for now); pooled rows OR
    // their bit in later.
    auto finish_rows = [&](uint64_t grp, auto pooled_c) __attribute__((always_inline)) {
        constexpr bool POOLED = decltype(pooled_c)::value;
        // (the result pointers: read from the kernel arguments here, once per group, instead of living in SGPRs through the walk -- see
        // kernarg_here, needle_walk.h)
        const KernargPtr ka = kernarg_here();
        uint64_t *const o_bitmap = kernarg_ptr<uint64_t>(ka, (uint32_t)offsetof(ScanArgs, bitmap));
        uint32_t *const o_end_state = kernarg_ptr<uint32_t>(ka, (uint32_t)offsetof(ScanArgs, end_state));
        bool res;
        if (OP == OP_FIND) res = row_ok && !dead && (last >= 0);
        else res = row_ok && (st >= accept_lo);
        if (!POOLED) {
            const uint64_t word = __ballot(res);
            // In a kernel with a pool the word is written by an agent-scope ATOMIC store: rows the group deferred OR their
            // bits into this same word later (below), from another lane of this same wave.  Both are then atomic accesses
            // of one address issued in program order by one wavefront, which the memory pipeline keeps in issue order per
            // address up to the L2 (the point of coherence); a plain store would leave that to how the vector L1 happens to
            // drain.  (Relaxed: no fence, no vmcnt wait -- the tile prefetch in flight is not drained.)
            if (lane == 0) {
                if (POOL) __hip_atomic_store(&o_bitmap[grp], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else o_bitmap[grp] = word;
            }
        } else if (res) {
            __hip_atomic_fetch_or(&o_bitmap[my_row >> 6], 1ull << (my_row & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (o_end_state && row_ok) o_end_state[my_row] = st; // (speculative stripes, table modes only: the state at the stripe's end)
        if (OP != OP_FIND) return;
        int32_t s = -1;
        const int32_t e = res ? last : -1;
        if (a.fixed_len >= 0) {
            s = res ? last - a.fixed_len : -1; // :640-646
        } else if (LEN || a.hdr.fa_len_off) {
            // the "lengths" automaton (needle_lower.h): the state the walk stopped in -- a dead-with-a-match-pending state D_L,
            // or at the row's end any state -- remembers how long its last match was: :640-646 generalised per state, no
            // indexBackwards
            uint32_t pidx = st;
            if (MODE == MODE_SPARSE) { // a live stop state (the row ended) asks its END record for the D_L of its pending length
                const uint32_t st_end = sparse_end<CW>(wk, st, st > wk.dead_hi, a.hdr.sp_end_col4);
                pidx = (st_end & 0xFFFFu) - a.hdr.sp_dead_row0;
            }
            s = res ? last - (int32_t)lds_u8(a.hdr.fa_len_off + pidx) : -1;
        } else {
            // indexBackwards(end - 1, FROM), :536-583: backward_walk (needle_walk.h).  The text it reads: the 32-byte window
            // [snapB | snapA] goes to the lane's own row of the LDS tile (walked and free by now), so that char p is ONE ds_read
            // at a per-lane address instead of a select chain over eight registers per char; text before the window comes from
            // memory (the row's line was fetched a moment ago: L2).
            const uint8_t *rowp = a.rows + (row_ok ? my_row : 0) * a.stride_bytes;
            if (res) {
                *(lds_u32x4 *)(uintptr_t)(tile.row_addr) = snapB;
                *(lds_u32x4 *)(uintptr_t)(tile.row_addr + 16u) = snapA;
            }
            const uint32_t win_lo = snapB_ok ? 0u : 16u;                        // first valid byte of the window
            const uint32_t win0 = (uint32_t)(snap_pi - 1) * 16u + win_lo;       // byte offset (in the row) of its start
            const int32_t lastb = backward_walk<CW>(a, res, last, cursor, tile.row_addr + win_lo, win0, 32u - win_lo, 0u, rowp);
            s = res ? lastb : -1;
        }
        if (row_ok) {
            uint32_t *const o_packed = kernarg_ptr<uint32_t>(ka, (uint32_t)offsetof(ScanArgs, packed));
            if (o_packed && kernarg_u32(ka, (uint32_t)offsetof(ScanArgs, packed8))) { // wave-uniform: one uint16 per row (rows <= 256 chars)
                ((uint16_t *)o_packed)[my_row] = pack8(s, e);
            } else if (o_packed) { // wave-uniform: one dword per row (no match: s = e = -1 -> 0xFFFFFFFF)
                o_packed[my_row] = ((uint32_t)s & 0xFFFFu) | ((uint32_t)e << 16);
            } else {
                kernarg_ptr<int32_t>(ka, (uint32_t)offsetof(ScanArgs, start))[my_row] = s;
                kernarg_ptr<int32_t>(ka, (uint32_t)offsetof(ScanArgs, end))[my_row] = e;
            }
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
    // ---- survivor pool: hand the unresolved rows of the current group (lanes of `live`, all continuing at chunk
    // next_ck) to free pool lanes.  The caller has checked that there are enough free lanes.
    auto defer_rows = [&](uint64_t live, uint32_t next_ck) __attribute__((always_inline)) {
        const bool is_live = (live >> lane) & 1ull;
        const uint32_t n_live = (uint32_t)__builtin_popcountll(live);
        const uint64_t free = ~pool_mask;
        // rank of this lane among the live lanes / among the free lanes below it
        const uint32_t q = __builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0u));
        const uint32_t rf = __builtin_amdgcn_mbcnt_hi((uint32_t)(free >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)free, 0u));
        // push every lane's index to a distinct lane -- the r-th live lane to lane r -- so that lane r holds "the
        // r-th live lane"; the r-th free pool lane then pulls that lane's row context
        const uint32_t dest = is_live ? q : n_live + ((uint32_t)lane - q);
        const uint32_t nth_live = (uint32_t)__builtin_amdgcn_ds_permute((int)(dest << 2), lane);
        const bool take = ((free >> lane) & 1ull) && rf < n_live;
        const uint32_t src = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((rf & 63u) << 2), (int)nth_live) << 2;
        const uint32_t t_row = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)(uint32_t)my_row);
        const uint32_t t_st = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)st);
        p_row = take ? t_row : p_row;
        p_st = take ? t_st : p_st;
        p_ck = take ? next_ck : p_ck;
        if (OP == OP_FIND) {
            const int32_t t_last = __builtin_amdgcn_ds_bpermute((int)src, last);
            p_last = take ? t_last : p_last;
        }
        pool_mask |= __ballot(take);
        row_ok = row_ok && !is_live; // the deferred rows' verdicts come later
    };

    // One pool step: the next 128-byte line of every pooled row.  Uses (and clobbers) the tile registers R and the
    // per-group state.  Rows that resolve are written out and leave the pool.
    auto pool_step = [&]() __attribute__((always_inline)) {
        const bool in_pool = (pool_mask >> lane) & 1ull;
        my_row = p_row;
        row_ok = in_pool;
        len = 0;
        if (in_pool) len = a.lengths ? a.lengths[p_row] : a.row_len;
        st = in_pool ? p_st : 0u;
        last = p_last;
        cursor = 0;
        dead = false;
        snap_pi = 1 << 24; // (beyond any real piece) nothing of the row's earlier text is held: indexBackwards reads it from memory
        snapB_ok = false;
        carry_ok = false;
        // (free slots gather row 0, unit 0: a valid address whatever the batch -- their stale row / chunk could point anywhere)
        const uint32_t g_row = in_pool ? p_row : 0u;
        const uint32_t unit = in_pool ? p_ck / NT : 0u;
        // gather: slot s of the LDS tile = the row of pool lane s
#pragma unroll
        for (int j = 0; j < G::kInstrs; ++j) {
            const uint32_t slot = (uint32_t)(j * G::kRowsPerInstr) + row_in_instr;
            const uint32_t srow = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(slot << 2), (int)g_row);
            const uint32_t sunit = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(slot << 2), (int)unit);
            const uint32_t kk = p_in_row ^ (uint32_t)G::swz((int)slot);
            const uint8_t *ptr = a.rows + (uint64_t)srow * a.stride_bytes + (uint64_t)sunit * (NT * CHB) + kk * 16u;
#pragma unroll
            for (int t = 0; t < NT; ++t) R[t][j] = load_row16<NT == 1>(ptr + t * CHB);
        }
        uint32_t n_ch = (len * CW + CHB - 1) / CHB; // chunks of this lane's row
        if (n_ch == 0) n_ch = 1;
        bool mine = in_pool;
        for (int t = 0; t < NT; ++t) { // (a run-time loop: one copy of the tile walk; R[1] moves down for the second tile)
#pragma unroll
            for (int j = 0; j < G::kInstrs; ++j) store_piece(tile, j, R[0][j]);
            const uint64_t live = walk_tile((p_ck + (uint32_t)t) * (CHB / CW), (p_ck + (uint32_t)t) * G::kPieces);
            carry_ok = true;
            // rows that resolved in this tile, or whose last chunk it was, are written out and leave the pool; they sit
            // out the line's second tile parked in the sink (it may lie beyond their row)
            const bool more = mine && ((live >> lane) & 1ull) && (p_ck + (uint32_t)t + 1u < n_ch);
            row_ok = mine && !more;
            pool_mask &= ~__ballot(row_ok);
            finish_rows(0, std::true_type{});
            mine = more;
            if (!mine) st = 0u;
            if (__ballot(mine) == 0ull) break;
            if (NT == 2) {
#pragma unroll
                for (int j = 0; j < G::kInstrs; ++j) R[0][j] = R[NT - 1][j];
            }
        }
        p_st = st;
        p_last = last;
        p_ck = in_pool ? p_ck + NT : 0u; // (only rows in the pool advance)
    };

    if (g < last_group) {
        uint32_t ck = 0;
        uint32_t pred_exit = 0xFFFFFFFFu; // chunk after which the previous group left early (prefetch predictor)
        begin_group(g);
        fetch(g, 0);
        // One tile: stage it, prefetch, walk it.  Returns 0 = same group continues with the next tile; the group ended
        // and g moved on: 1 = the unit 0 of the new g is in R or in flight, 3 = it still has to be fetched.
        auto step = [&](auto tc) __attribute__((always_inline)) -> int {
            constexpr int T = decltype(tc)::value;
            // Prefetch while this tile is walked whenever the unit's registers are all free after staging it: at
            // the unit's last tile, at the group's last chunk, or where the previous group left early.
            const bool do_pf = (T == NT - 1) || (ck + 1 >= n_chunks) || (ck >= pred_exit);
            const bool pf_same = (T == NT - 1) && (ck + 1 < n_chunks) && (ck < pred_exit);
            const uint64_t pf_g = pf_same ? g : g + wave_cnt;
            stage_and_fetch(tc, do_pf && pf_g < last_group, pf_g, pf_same ? (ck + 1) / NT : 0u);
            asm volatile("" ::: "memory"); // keep the prefetch issued ahead of the walk
            const uint64_t live = walk_tile(ck * (CHB / CW), ck * G::kPieces);
            bool group_done = live == 0ull || (ck + 1 >= n_chunks);
            if (POOL && !group_done && T == NT - 1 && a.defer_max_live) {
                // few unresolved rows after a whole line: they go to the pool, the group ends here
                const uint32_t n_live = (uint32_t)__builtin_popcountll(live);
                if (n_live <= a.defer_max_live && n_live <= 64u - (uint32_t)__builtin_popcountll(pool_mask)) {
                    defer_rows(live, ck + 1);
                    group_done = true;
                }
            }
            if (!group_done) {
                if (do_pf && !pf_same) fetch(g, (ck + 1) / NT); // predicted an exit that did not happen
                ++ck;
                return 0;
            }
            finish_rows(g, std::false_type{});
            pred_exit = (ck + 1 < n_chunks) ? ck : 0xFFFFFFFFu;
            g += wave_cnt;
            return (!do_pf || pf_same) ? 3 : 1; // 3: nothing (or this group's next unit) was prefetched
        };
        for (;;) {
            int r = step(std::integral_constant<int, 0>{});
            if (NT == 2 && r == 0) r = step(std::integral_constant<int, NT - 1>{});
            if (r == 0) continue;
            // a group ended.  Run pool steps while the pool could not take another group's survivors -- or, after this
            // wave's last pipelined group, until it is empty.
            const bool more_groups = g < last_group;
            bool have_unit = r == 1;
            if (POOL) {
                const uint32_t keep = more_groups ? 64u - a.defer_max_live : 0u;
                while ((uint32_t)__builtin_popcountll(pool_mask) > keep) {
                    pool_step();
                    have_unit = false; // R was used
                }
            }
            if (!more_groups) break;
            if (!have_unit) fetch(g, 0);
            ck = 0;
            begin_group(g);
        }
    }
    // ---- the batch's last group(s): clamped loads, no pipelining (at most a couple of waves in the whole grid)
    for (; g < n_groups; g += wave_cnt) {
        begin_group(g);
        for (uint32_t ck = 0; ck < n_chunks; ++ck) {
            fetch_clamped(g, ck);
            stage_and_fetch(std::integral_constant<int, 0>{}, false, 0, 0);
            if (walk_tile(ck * (CHB / CW), ck * G::kPieces) == 0ull) break;
        }
        finish_rows(g, std::false_type{});
    }
}

// ------------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------------
struct LaunchShape {
    int grid, waves, chb;
    size_t lds;
};

template <int OP, int CW, int MODE, bool GUARD, int CHB, bool LEN>
static hipError_t launch_one(const ScanArgs &a, LaunchShape sh, hipStream_t stream) {
    auto k = scan_kernel<OP, CW, MODE, GUARD, CHB, LEN>;
    static thread_local uint64_t configured = 0;
    if (hipError_t e = allow_full_lds((const void *)k, configured); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(sh.grid), dim3(sh.waves * 64), sh.lds, stream, a);
    return hipGetLastError();
}

template <int OP, int CW, int MODE, bool GUARD, bool LEN>
static hipError_t launch_h(const ScanArgs &a, LaunchShape sh, hipStream_t s) {
    return sh.chb == 128 ? launch_one<OP, CW, MODE, GUARD, 128, LEN>(a, sh, s) : launch_one<OP, CW, MODE, GUARD, 64, LEN>(a, sh, s);
}

template <int OP, int CW, int MODE, bool LEN = false>
static hipError_t launch_g(const ScanArgs &a, bool guard, LaunchShape sh, hipStream_t s) {
    if constexpr (OP == OP_FIND && !LEN && (MODE == MODE_TABLE8 || MODE == MODE_TABLE16 || MODE == MODE_SPARSE || MODE == MODE_PAIR)) {
        if (a.hdr.fa_len_off) return launch_g<OP, CW, MODE, true>(a, guard, sh, s); // (the only modes such programs have)
    }
    return guard ? launch_h<OP, CW, MODE, true, LEN>(a, sh, s) : launch_h<OP, CW, MODE, false, LEN>(a, sh, s);
}

template <int OP, int CW>
static hipError_t launch_m(const ScanArgs &a, bool guard, LaunchShape sh, hipStream_t s) {
    switch (a.hdr.mode) {
    case MODE_PACK: return launch_g<OP, CW, MODE_PACK>(a, guard, sh, s);
    case MODE_TABLE8: return launch_g<OP, CW, MODE_TABLE8>(a, guard, sh, s);
    case MODE_TABLE16: return launch_g<OP, CW, MODE_TABLE16>(a, guard, sh, s);
    case MODE_PAIR: return CW == 1 ? launch_g<OP, 1, MODE_PAIR>(a, guard, sh, s) : hipErrorInvalidValue; // 8-bit rows only
    case MODE_HYBRID: return launch_g<OP, CW, MODE_HYBRID>(a, guard, sh, s);
    case MODE_SPARSE: return launch_g<OP, CW, MODE_SPARSE>(a, guard, sh, s);
    default: return launch_g<OP, CW, MODE_GLOBAL>(a, guard, sh, s);
    }
}

template <int OP>
static hipError_t launch_c(const ScanArgs &a, int cw, bool guard, LaunchShape sh, hipStream_t s) {
    return cw == 1 ? launch_m<OP, 1>(a, guard, sh, s) : launch_m<OP, 2>(a, guard, sh, s);
}

} // namespace needle
