// needle_find_all.hip -- every non-overlapping match of every row in ONE pass over the batch (the
// reference's repeated Matcher.find(), DFAClassBuilder.java:616-659; DFACompilerTest.java:66-78,671-699).
//
// The round-per-match form (needle_find_next_dev fed its own `end` as the next cursor) reads the whole batch once per
// round: a keyword dictionary over text finds a handful of matches per row and up to a few dozen in the worst row of
// ten million, so the batch crosses the HBM bus dozens of times.  Here a row is fetched once.
//
// Same data movement as the scan kernel (needle_scan.h): one row per lane, 64-row groups, whole lines HBM -> VGPRs
// -> XOR-swizzled LDS tile, the lowered automaton staged once per workgroup.  What differs is the walk.  After a match
// [start, end) the reference restarts the search automaton AT `end` -- chars the walk already consumed while it
// waited for the automaton to die -- so the rows of a wave stop being at the same char.  Every lane therefore keeps its
// own PIECE index (16-byte piece of its row) and the wave iterates "each live lane walks its current piece": a
// ds_read_b128 at a per-lane tile address, then walk_piece_fa() below, whose one per-char guard hides the chars before
// the lane's cursor (the restart point, anywhere inside a piece).  A lane whose automaton died files the match, moves
// its cursor to `end`, steps back to the piece holding `end` (from memory, in the rare case it is in the previous tile)
// and starts again; lanes that reached the tile's end wait there for the others.
//
// Results: dense per-row slots (needle_find_all_dev), or compact filing at caller-computed offsets after a counting
// pass of the same kernel that files nothing (needle_count_matches_dev / needle_find_all_csr_dev).
//
// Start indices (indexBackwards, :529-586).  A fixed-length pattern has start = end - L.  Otherwise the backward
// automaton walks right to left from end - 1, bounded by the cursor the match was searched from.  Doing that at the
// moment a lane resolves would run the backward walk's code for the one or two lanes resolving in any given
// iteration.  The walk therefore only files the ENDS; at the end of a 64-row group every start is an independent
// indexBackwards (a match's bound is the end of the one before it), so the group's matches are numbered through (a
// prefix sum of the lanes' counts) and handed out 64 at a time, one per lane, whichever row they belong to: no lane
// waits for another row's longer list of matches.  Their text comes back from memory / L2 (the 32 bytes ending with
// the match's last char, into the lane's by then free tile row).  Patterns that match the empty string need every
// start at once -- an empty match ends its row (see needle_find_all_dev in needle_hip.h) -- and take the immediate form.
// (Tried before: a register stack of pending ends flushed tile by tile with the text still in LDS -- a round per pending
// match of the busiest lane and tile: dictionary 3.3 ms against 2.6; the same rounds as a kernel of its own: 3.0 ms, its
// re-reads of ends and text all miss the L2.)
#include "needle_walk.h"
#include "needle_find_all.h"

namespace needle {

// One 16-byte piece of one row in the find-all walk.  Chars before the lane's cursor (the first skip_rel of the piece)
// go through the PRE column (identity); chars past the row's end are walked like any others -- the search ends with
// that piece whatever its state is, and the caller masks their accept flags.  Accept flags are LOGGED, one shift per
// char (walk_piece selects a position per char: three VALU ops in a walk that is issue-bound under its guards):
// returns the flags of the piece's chars, char i at bit i.
// CUT (the "lengths" form on the piece a ragged row ends in): chars from in_row on take the PAD column, which there leads
// to the dead state that remembers the pending match -- the state the piece ends in is then the one the ROW ends in.
// SKIPST (programs with skip states, needle_device.h fa_skip_lo): no cursor guard at all -- a search restarted inside the piece
// enters it in the skip state that swallows the chars before its cursor.
template <int CW, int MODE, bool CUT = false, bool SKIPST = false>
__device__ __forceinline__ uint32_t walk_piece_fa(const Walk &wk, const uint32_t (&w)[4], uint32_t skip_rel, uint32_t accept_lo,
                                                  uint32_t &st, uint32_t in_row = 0) {
    constexpr int CPP = 16 / CW;
    uint32_t col[CPP];
    piece_lookups<MODE, CW, false>(wk, w, 0, 0, 0, col);
    const uint32_t tb_in_col = col_has_table_off<MODE, CW>() ? wk.table_off : 0u; // (UTF-16 table programs: needle_walk.h)
    if (MODE == MODE_PACK) lds_fence();
    if (MODE == MODE_SPARSE) {
        // The compressed automaton (needle_device.h) has no PRE / PAD columns: chars before the lane's cursor and chars past the
        // row's end (in_row: chars of the piece inside the row) leave the state as it is -- a lengths program's state FREEZES
        // at the row's end and its END record names the pending length (needle_scan.h finish_rows).
        uint32_t h = 0;
#pragma unroll
        for (int i = 0; i < CPP; ++i) {
            const uint32_t ns = apply<MODE, CW>(wk, st, col[i]);
            st = ((uint32_t)i >= skip_rel && (uint32_t)i < in_row) ? ns : st;
            h |= (st >= accept_lo ? 1u : 0u) << i; // (flags of chars outside [skip_rel, in_row) are masked by the caller)
        }
        return h;
    }
    if (CUT) {
#pragma unroll
        for (int i = 0; i < CPP; ++i) col[i] = ((uint32_t)i < in_row) ? col[i] : wk.pad_e + tb_in_col;
    }
    if (!SKIPST) {
#pragma unroll
        for (int i = 0; i < CPP; ++i) col[i] = ((uint32_t)i < skip_rel) ? wk.pre_e + tb_in_col : col[i];
    }
    uint32_t h = 0;
    const uint32_t acc_m1 = accept_lo - 1u;
#pragma unroll
    for (int i = 0; i < CPP; ++i) {
        st = apply<MODE, CW>(wk, st, col[i]);
        if (MODE == MODE_PACK) h = __builtin_amdgcn_alignbit(st, h, 1);                 // accepting states: odd field offsets
        else if (MODE == MODE_HYBRID) h = __builtin_amdgcn_alignbit(h, st << 16, 31);   // accepting: bit 15 of the entry
        else h = __builtin_amdgcn_alignbit(h, acc_m1 - st, 31);                          // accepting: st >= accept_lo
    }
    // packed: char i at bit 32 - CPP + i; the others: char i at bit CPP - 1 - i
    if (MODE != MODE_PACK) h = __builtin_bitreverse32(h);
    return h >> (32 - CPP);
}

// LM: the "lengths" form (fa.lmode) of a program with skip states
template <int CW, int MODE, int CHB, bool LM>
__global__ __launch_bounds__(kWavesPerBlock * 64) void find_all_kernel(const FindAllArgs fa) {
    using G = Geom<CHB>;
    const ScanArgs &a = fa.s;
    constexpr int CPP = 16 / CW; // chars per 16-byte piece
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_waves = blockDim.x >> 6;

    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();
    for (uint32_t i = tid * 16u; i < a.hdr.lds_bytes; i += blockDim.x * 16u)
        *(u32x4 *)(smem + i) = *(const u32x4 *)(a.prog + i);
    __syncthreads();

    Walk wk;
    constexpr uint32_t ELEM = (MODE == MODE_TABLE16 || MODE == MODE_HYBRID) ? 2u : 1u;
    wk.ncols_e = a.hdr.n_cols * ELEM;
    wk.pad_e = (MODE == MODE_PACK) ? a.hdr.pad_f : a.hdr.pad_col * ELEM;
    wk.pre_e = (MODE == MODE_PACK) ? a.hdr.pre_f : (a.hdr.pad_col + 1u) * ELEM;
    wk.pad_b = wk.pre_b = 0;
    wk.table_off = a.hdr.off_table;
    wk.win_on = 0, wk.win_lo = 0, wk.win_hi = 0; // (the find-all programs are lowered without window addressing)
    wk.sp_chains = 0, wk.sp_pad_ident = 0, wk.dead_hi = 0;
    wk.flat = (CW == 2 && (MODE == MODE_TABLE8 || MODE == MODE_TABLE16)) ? a.hdr.flat_pages : 0u;
    if ((MODE == MODE_TABLE8 || MODE == MODE_TABLE16) && a.hdr.win_on) { // a lengths program in window layout (needle_scan.h sets these up the same way)
        wk.win_on = 1, wk.win_lo = a.hdr.win_lo_e, wk.win_hi = a.hdr.win_hi_e;
        wk.table_off = a.hdr.off_table - a.hdr.win_lo_e;
    }
    if (MODE == MODE_SPARSE) { // the scan kernels' compressed lengths program (needle_scan.h sets these up the same way)
        wk.pad_e = wk.pre_e = a.hdr.win_lo_e;
        wk.win_on = a.hdr.win_on, wk.win_lo = a.hdr.win_lo_e, wk.win_hi = a.hdr.win_hi_e;
        wk.dead_hi = a.hdr.fa_dead_hi;
        wk.sp_chains = a.hdr.sp_chains, wk.sp_pad_ident = a.hdr.sp_pad_ident;
    }
    wk.lane4 = (uint32_t)lane * 4u; // packed mode on 8-bit rows: all 64 lane copies of F are there (no tiles in the F rows)
    wk.gtable = (const uint16_t *)(a.prog + (MODE == MODE_HYBRID ? a.hdr.off_gtable : a.hdr.off_table));
    wk.hot_last = a.hdr.hot_bytes - 2u;
    const uint32_t accept_lo = MODE == MODE_PACK ? a.hdr.accept_off : a.hdr.accept_lo;
    const uint32_t start_state = MODE == MODE_PACK ? a.hdr.start_off : a.hdr.start;

    Tile tile;
    {
        const uint32_t base = ((a.hdr.lds_bytes + 15u) & ~15u) + (uint32_t)wave * G::kTileBytes;
        tile.store_addr = base + (uint32_t)(lane / G::kPieces) * CHB + (uint32_t)(lane % G::kPieces) * 16u;
        tile.store_step = G::kRowsPerInstr * CHB;
        tile.row_addr = base + (uint32_t)lane * CHB;
    }
    const uint32_t swz16 = (uint32_t)G::swz(lane) << 4; // byte b of this lane's tile row is at row_addr + (b ^ swz16)

    const uint64_t n_groups = (a.n_rows + 63) >> 6;
    const uint64_t wave_cnt = (uint64_t)gridDim.x * n_waves;
    uint64_t g = (uint64_t)blockIdx.x * n_waves + wave;
    if (g >= n_groups) return;

    // tile fetch: as in scan_kernel (needle_scan.h) -- a fetch unit is one 128-byte line per row: one tile of 128-byte
    // pieces or the two 64-byte tiles of the same lines, requested back to back
    const uint32_t q = (uint32_t)lane >> 4;
    const uint32_t p_in_row = (uint32_t)(lane % G::kPieces);
    const uint32_t row_in_instr = (uint32_t)(lane / G::kPieces);
    const uint32_t o_even = row_in_instr * (uint32_t)a.stride_bytes + 16u * (p_in_row ^ q);
    const uint32_t o_odd = CHB == 128 ? (row_in_instr * (uint32_t)a.stride_bytes + 16u * (p_in_row ^ q ^ 4u)) : o_even;
    const uint64_t load_step = (uint64_t)G::kRowsPerInstr * a.stride_bytes;
    constexpr int NT = (CHB == 64) ? 2 : 1;
    u32x4 R[NT][G::kInstrs];
    auto fetch = [&](uint64_t grp, uint32_t unit) __attribute__((always_inline)) {
        const uint8_t *base = a.rows + (grp << 6) * a.stride_bytes + unit * (NT * CHB);
#pragma unroll
        for (int j = 0; j < G::kInstrs; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) R[t][j] = load_row16<NT == 1>(base + t * CHB + j * load_step + ((j & 1) ? o_odd : o_even));
    };
    auto fetch_clamped = [&](uint64_t grp, uint32_t chunk) __attribute__((always_inline)) {
        const uint32_t last_r = (uint32_t)(a.n_rows - 1 - (grp << 6));
        const uint32_t stride = (uint32_t)a.stride_bytes;
        const uint8_t *gbase = a.rows + (grp << 6) * a.stride_bytes;
#pragma unroll
        for (int j = 0; j < G::kInstrs; ++j) {
            uint32_t r = (uint32_t)(j * G::kRowsPerInstr) + row_in_instr;
            const uint32_t kk = p_in_row ^ (uint32_t)G::swz((int)r);
            r = r < last_r ? r : last_r;
            uint32_t pb = chunk * CHB + kk * 16u;
            if (pb + 16u > stride) pb = stride - 16u;
            R[0][j] = load_row16<false>(gbase + (r * stride + pb));
        }
    };

    // ---- per-group (per-row) state
    uint64_t my_row = 0;
    bool row_ok = false, done = true;
    uint32_t len = 0, n_chunks = 1, st = 0, pi = 0, count = 0;
    int32_t last = -1, cursor = 0;
    const uint8_t *rowp = a.rows;
    uint64_t out0 = 0;      // index of this row's first result slot (dense: row * slots; compact: offsets[row])
    uint32_t cap = 0;       // matches this row may file

    auto backward = [&](bool act, int32_t en, int32_t bound, uint32_t tile_b0) __attribute__((always_inline)) -> int32_t {
        return backward_walk<CW, true>(a, act, en, bound, tile.row_addr, tile_b0, (uint32_t)CHB, swz16, rowp);
    };

    auto begin_group = [&](uint64_t grp) __attribute__((always_inline)) {
        my_row = (grp << 6) + lane;
        row_ok = my_row < a.n_rows;
        len = 0;
        if (row_ok) len = a.lengths ? a.lengths[my_row] : a.row_len;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(len)); // (the tile it is about to stage was requested earlier still)
        const uint32_t max_len = a.lengths ? wave_max(len) : a.row_len;
        n_chunks = (max_len * CW + CHB - 1) / CHB;
        if (n_chunks == 0) n_chunks = 1;
        rowp = a.rows + (row_ok ? my_row : 0) * a.stride_bytes;
        done = !row_ok;
        cursor = 0;
        st = start_state;
        last = a.hdr.root_accepting ? 0 : -1; // :356 literal 0 (cursor 0: the same whether 0 < length or not)
        pi = 0;
        count = 0;
        out0 = fa.kshift ? (my_row >> 6) * fa.slots * 64u + (my_row & 63u) : my_row * fa.slots;
        cap = fa.count_only ? 0xFFFFFFFFu : fa.slots;
        if (fa.offsets) {
            out0 = row_ok ? fa.offsets[my_row] : 0;
            cap = row_ok ? (uint32_t)(fa.offsets[my_row + 1] - out0) : 0u;
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(cap)); // (as for len above: no vmcnt wait inside the walk)
        }
    };

    // Walk the tile in LDS (chunk ck of the group's rows) until every live lane is past it.
    auto walk_tile = [&](uint32_t ck) __attribute__((always_inline)) {
        const uint32_t tile_p0 = ck * G::kPieces, tile_p1 = tile_p0 + G::kPieces, tile_b0 = ck * CHB;
        for (;;) {
            const uint32_t p0 = pi * CPP;
            const bool beyond = p0 >= len; // nothing of the row there (a walk over PAD: the search ends)
            const bool active = !done && (pi < tile_p1 || beyond);
            if (__ballot(active) == 0ull) break;
            // The iteration is straight-line code for all 64 lanes -- idle lanes walk a piece too and their results are
            // dropped by selects: every divergent region here costs the compiler a copy of the loop-carried lane state per
            // path, and with it more VALU ops than the walk itself.
            const bool in_lds = active && !beyond && pi >= tile_p0;
            u32x4 v = *(const lds_u32x4 *)(uintptr_t)(tile.row_addr + (((in_lds ? pi - tile_p0 : 0u) << 4) ^ swz16));
            const bool in_mem = active && !beyond && pi < tile_p0;
            if (__ballot(in_mem) != 0ull) { // a restart in the previous tile (rare): waited for HERE, so that the common path
                                            // carries no vmcnt wait (tile prefetch and match stores are in flight)
                if (in_mem) {
                    v = *(const u32x4 *)(rowp + (uint64_t)pi * 16u);
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v));
                }
            }
            const uint32_t w[4] = {v[0], v[1], v[2], v[3]}; // (a piece beyond the row: whatever is there -- its flags are masked)
            const uint32_t skip_rel = (uint32_t)cursor > p0 ? (uint32_t)cursor - p0 : 0u; // < CPP: the cursor's piece, or none
            const uint32_t st_old = st;
            uint32_t st_new = st;
            const uint32_t in_row = len > p0 ? len - p0 : 0u; // chars of the piece inside the row (all, if >= CPP)
            uint32_t acc = walk_piece_fa<CW, MODE, false, LM>(wk, w, skip_rel, accept_lo, st_new, in_row);
            if (!LM) acc &= ~((1u << skip_rel) - 1u);          // an accepting start state does not count before the cursor
            acc &= in_row < (uint32_t)CPP ? (1u << in_row) - 1u : 0xFFFFFFFFu;
            acc = active ? acc : 0u;
            last = acc ? (int32_t)(p0 + 32u - (uint32_t)__builtin_clz(acc)) : last;
            st = active ? st_new : st;
            // (fa_dead_n: the "lengths" automaton's dead-with-a-match-pending states; 0 for every other program)
            const bool died = MODE == MODE_SPARSE ? st_new <= wk.dead_hi : (st_new == 0u || st_new - a.hdr.fa_dead_lo < a.hdr.fa_dead_n);
            const bool ended = active && (died || p0 + CPP >= len);
            pi += (active && !ended) ? 1u : 0u;
            if (__ballot(ended) == 0ull) continue;
            // ---- find() returns for the lanes of `ended` (:629-657)
            const bool hit = ended && last >= 0;
            const int32_t en = last;
            if (ended && !hit) done = true; // no further match in this row
            if (LM || fa.lmode) {
                // The "lengths" automaton (needle_lower.h): the state the search ended in remembers how long its last match
                // was -- start = end - pend[state], no indexBackwards (DFAClassBuilder.java:640-646 generalised per state).
                // A ragged row that ends INSIDE this piece was walked past its end above (harmless for the flags, which are
                // masked, but not for the state): that piece is walked again from its entry state with the PAD column.
                uint32_t st_end = st_new;
                if (MODE == MODE_TABLE8 || MODE == MODE_TABLE16) { // (the only modes such a program has)
                    const bool cut = hit && in_row < (uint32_t)CPP;
                    if (__ballot(cut) != 0ull) {
                        uint32_t st_fix = st_old;
                        (void)walk_piece_fa<CW, MODE, true, LM>(wk, w, skip_rel, accept_lo, st_fix, in_row);
                        st_end = cut ? st_fix : st_end;
                    }
                }
                if (MODE == MODE_SPARSE) { // a live end state (the row ended) asks its END record for the D_L of its pending length
                    const uint32_t e_st = sparse_end<CW>(wk, st_new, hit && st_new > wk.dead_hi, a.hdr.sp_end_col4);
                    st_end = (e_st & 0xFFFFu) - a.hdr.sp_dead_row0;
                }
                const int32_t mlen = (int32_t)lds_u8(a.hdr.fa_len_off + (hit ? st_end : 0u));
                const bool file = hit && count < cap;
                if (hit && !file) *fa.more = 1;
                done = done || (hit && !file);
                if (file && !fa.count_only) {
                    if (fa.packed) {
                        fa.packed[out0 + ((uint64_t)count << fa.kshift)] = (uint32_t)(en - mlen) | ((uint32_t)en << 16);
                    } else {
                        fa.starts[out0 + ((uint64_t)count << fa.kshift)] = en - mlen;
                        fa.ends[out0 + ((uint64_t)count << fa.kshift)] = en;
                    }
                }
                count += file ? 1u : 0u;
                cursor = file ? en : cursor;
                const uint32_t pi_en = ((uint32_t)en * CW) >> 4;
                uint32_t st_again = start_state;
                if (LM) { // en - pi_en * CPP chars of the piece lie before the new cursor: S_k swallows them
                    const uint32_t rel = (uint32_t)en - pi_en * (uint32_t)CPP;
                    st_again = rel ? a.hdr.fa_skip_lo + rel - 1u : start_state;
                }
                st = file ? st_again : st;
                last = file ? -1 : last;
                pi = file ? pi_en : pi;
            } else if (fa.defer) {
                // not nullable, start by indexBackwards: the match is not empty and ends beyond its cursor -- the row goes on.
                // Written as selects, not branches: this block runs in most iterations (some lane of 64 has just resolved)
                // and every divergent branch costs a copy of the loop-carried lane state per path.
                const bool file = hit && count < cap;
                if (hit && !file) *fa.more = 1;
                done = done || (hit && !file);
                if (file && !fa.count_only) { // (counting: nothing is filed)
                    if (fa.packed) fa.packed[out0 + ((uint64_t)count << fa.kshift)] = (uint32_t)en << 16; // (the start joins it in starts_phase)
                    else fa.ends[out0 + ((uint64_t)count << fa.kshift)] = en;
                }
                count += file ? 1u : 0u;
                cursor = file ? en : cursor;
                st = file ? start_state : st;
                last = file ? -1 : last;
                pi = file ? (((uint32_t)en * CW) >> 4) : pi;
            } else {
                int32_t s = en - a.fixed_len;
                if (a.fixed_len < 0) s = backward(hit, en, cursor, tile_b0);
                // en < s: the wrapped pseudo-match of a nullable pattern searched from cursor == length; dropped, ends the row
                const bool valid = hit && en >= s;
                if (hit && !valid) done = true;
                if (valid) {
                    if (count < cap) {
                        if (!fa.count_only) {
                            if (fa.packed) {
                                fa.packed[out0 + ((uint64_t)count << fa.kshift)] = (uint32_t)s | ((uint32_t)en << 16);
                            } else {
                                fa.starts[out0 + ((uint64_t)count << fa.kshift)] = s;
                                fa.ends[out0 + ((uint64_t)count << fa.kshift)] = en;
                            }
                        }
                        ++count;
                        // the row goes on only while the cursor advances (needle_hip.h)
                        if (en == s || en <= cursor) {
                            done = true;
                        } else {
                            cursor = en;
                            st = start_state;
                            last = a.hdr.root_accepting ? (((uint32_t)cursor < len) ? cursor : 0) : -1;
                            pi = ((uint32_t)en * CW) >> 4;
                        }
                    } else {
                        *fa.more = 1;
                        done = true;
                    }
                }
            }
        }
    };
    // defer != 0: the starts of the group's matches, found at the end of the group (the walk has filed every match's
    // end; match k of a row was searched from the end of match k - 1, so every start is an independent indexBackwards).  The
    // matches of the 64 rows are numbered through (prefix sum of the counts) and handed out 64 at a time, one per lane: no
    // lane waits for another row's longer list, which is what the tile-by-tile form pays for.  The text comes back from
    // memory / L2 into the lane's (by now free) tile row; the ends are read back with agent-scope loads (this wave wrote them
    // a moment ago: the plain stores are in L2 once vmcnt says so, a plain load might still hit a stale L1 line).
    auto starts_phase = [&](uint64_t grp) __attribute__((always_inline)) {
        uint32_t incl = count;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)incl, o);
            incl += lane >= o ? t : 0u;
        }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (total == 0u) return;
        const uint32_t excl = incl - count;
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): this wave's stores of the ends have reached L2
        for (uint32_t j0 = 0; j0 < total; j0 += 64u) {
            const uint32_t j = j0 + (uint32_t)lane;
            const bool act = j < total;
            uint32_t lo = 0, hi = 63;
#pragma unroll
            for (int it = 0; it < 6; ++it) { // the first lane whose inclusive count exceeds j
                const uint32_t mid = (lo + hi) >> 1;
                const uint32_t pm = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(mid << 2), (int)incl);
                const bool right = pm <= j;
                lo = right ? mid + 1u : lo;
                hi = right ? hi : mid;
            }
            const uint32_t owner = act ? lo : (uint32_t)lane;
            const uint32_t k = j - (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)excl);
            const uint64_t o_out0 = (uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)(uint32_t)out0) |
                                    ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)(uint32_t)(out0 >> 32)) << 32);
            const uint8_t *o_rowp = a.rows + ((grp << 6) + owner) * a.stride_bytes;
            int32_t en = 1, bound = 0;
            if (act) {
                if (fa.packed) {
                    en = (int32_t)(__hip_atomic_load(&fa.packed[o_out0 + ((uint64_t)k << fa.kshift)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 16);
                    if (k) bound = (int32_t)(__hip_atomic_load(&fa.packed[o_out0 + ((uint64_t)(k - 1) << fa.kshift)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 16);
                } else {
                    en = __hip_atomic_load(&fa.ends[o_out0 + ((uint64_t)k << fa.kshift)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (k) bound = __hip_atomic_load(&fa.ends[o_out0 + ((uint64_t)(k - 1) << fa.kshift)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            const uint32_t pa = ((uint32_t)(en - 1) * CW) >> 4; // window: the piece holding char en - 1 and the one before it
            const uint32_t pb = pa ? pa - 1u : 0u;
            u32x4 va = {0, 0, 0, 0}, vb = {0, 0, 0, 0};
            if (act) {
                va = *(const u32x4 *)(o_rowp + (uint64_t)pa * 16u);
                vb = *(const u32x4 *)(o_rowp + (uint64_t)pb * 16u);
            }
            *(lds_u32x4 *)(uintptr_t)(tile.row_addr) = vb;
            *(lds_u32x4 *)(uintptr_t)(tile.row_addr + 16u) = va;
            const uint32_t win_b0 = pa ? pb * 16u : 0u;
            const uint32_t w_addr = pa ? tile.row_addr : tile.row_addr + 16u;
            const int32_t st_k = fa.defer == 2u ? bound : backward_walk<CW>(a, act, en, bound, w_addr, win_b0, pa ? 32u : 16u, 0u, o_rowp); // (2: measurement aid)
            if (act) {
                if (fa.packed) fa.packed[o_out0 + ((uint64_t)k << fa.kshift)] = (uint32_t)st_k | ((uint32_t)en << 16);
                else fa.starts[o_out0 + ((uint64_t)k << fa.kshift)] = st_k;
            }
        }
    };
    auto end_group = [&](uint64_t grp) __attribute__((always_inline)) {
        if (row_ok && fa.counts) fa.counts[my_row] = count;
        if (fa.defer && !fa.count_only) starts_phase(grp);
    };
    auto stage = [&](auto tc) __attribute__((always_inline)) {
        constexpr int T = decltype(tc)::value;
#pragma unroll
        for (int j = 0; j < G::kInstrs; ++j) store_piece(tile, j, R[T][j]);
    };

    uint64_t last_group = n_groups - 1; // first group handled by the clamped tail below (as in scan_kernel)
    {
        const uint64_t group_bytes = 64 * a.stride_bytes;
        const uint64_t safe = a.total_bytes >= (uint64_t)(NT * CHB) ? (a.total_bytes - NT * CHB) / group_bytes : 0;
        if (safe < last_group) last_group = safe;
    }
    if (g < last_group) {
        fetch(g, 0);
        for (;;) {
            begin_group(g);
            uint32_t ck = 0;
            bool have_next = false; // R holds (or will hold) unit 0 of this wave's next group
            // the registers of a unit are free once its last tile is staged: the next unit -- of this group, or the
            // first one of the wave's next group -- is requested then and arrives while the tile is walked
            auto prefetch = [&]() __attribute__((always_inline)) {
                if (ck + 1 < n_chunks) fetch(g, (ck + 1) / NT);
                else if (g + wave_cnt < last_group) fetch(g + wave_cnt, 0), have_next = true;
            };
            for (;;) {
                stage(std::integral_constant<int, 0>{});
                asm volatile("" ::: "memory");
                if (NT == 1) prefetch();
                asm volatile("" ::: "memory");
                walk_tile(ck);
                ++ck;
                if (ck >= n_chunks || __ballot(!done) == 0ull) break;
                if (NT == 2) {
                    stage(std::integral_constant<int, NT - 1>{});
                    asm volatile("" ::: "memory");
                    prefetch();
                    asm volatile("" ::: "memory");
                    walk_tile(ck);
                    ++ck;
                    if (ck >= n_chunks || __ballot(!done) == 0ull) break;
                }
            }
            end_group(g);
            g += wave_cnt;
            if (g >= last_group) break;
            if (!have_next) fetch(g, 0);
        }
    }
    for (; g < n_groups; g += wave_cnt) {
        begin_group(g);
        for (uint32_t ck = 0; ck < n_chunks; ++ck) {
            fetch_clamped(g, ck);
            stage(std::integral_constant<int, 0>{});
            walk_tile(ck);
            if (__ballot(!done) == 0ull) break;
        }
        end_group(g);
    }
}

// ------------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------------
template <int CW, int MODE, int CHB, bool LM = false>
static hipError_t launch_fa(const FindAllArgs &fa, int grid, int waves, size_t lds, hipStream_t stream) {
    auto k = find_all_kernel<CW, MODE, CHB, LM>;
    static thread_local uint64_t configured = 0;
    if (hipError_t e = allow_full_lds((const void *)k, configured); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(waves * 64), lds, stream, fa);
    return hipGetLastError();
}
template <int CW, int MODE>
static hipError_t launch_fa_h(const FindAllArgs &fa, int chb, int grid, int waves, size_t lds, hipStream_t s) {
    if constexpr (MODE == MODE_TABLE8 || MODE == MODE_TABLE16) {
        if (fa.lmode && fa.s.hdr.fa_skip_lo)
            return chb == 128 ? launch_fa<CW, MODE, 128, true>(fa, grid, waves, lds, s) : launch_fa<CW, MODE, 64, true>(fa, grid, waves, lds, s);
    }
    return chb == 128 ? launch_fa<CW, MODE, 128>(fa, grid, waves, lds, s) : launch_fa<CW, MODE, 64>(fa, grid, waves, lds, s);
}
template <int CW>
static hipError_t launch_fa_m(const FindAllArgs &fa, int chb, int grid, int waves, size_t lds, hipStream_t s) {
    switch (fa.s.hdr.mode) {
    case MODE_PACK: return launch_fa_h<CW, MODE_PACK>(fa, chb, grid, waves, lds, s);
    case MODE_TABLE8: return launch_fa_h<CW, MODE_TABLE8>(fa, chb, grid, waves, lds, s);
    case MODE_TABLE16: return launch_fa_h<CW, MODE_TABLE16>(fa, chb, grid, waves, lds, s);
    case MODE_HYBRID: return launch_fa_h<CW, MODE_HYBRID>(fa, chb, grid, waves, lds, s);
    case MODE_GLOBAL: return launch_fa_h<CW, MODE_GLOBAL>(fa, chb, grid, waves, lds, s);
    case MODE_SPARSE: return fa.lmode ? launch_fa_h<CW, MODE_SPARSE>(fa, chb, grid, waves, lds, s) : hipErrorInvalidValue; // (lengths programs only)
    default: return hipErrorInvalidValue; // (pair mode: the caller lowers the automaton without it)
    }
}

// One persistent workgroup per CU; the shape (waves x tile bytes) follows the automaton's LDS footprint.
hipError_t launch_find_all(int char_width, const FindAllArgs &fa, int n_cus, hipStream_t stream) {
    if (fa.s.n_rows == 0) return hipSuccess;
    const size_t p = (fa.s.hdr.lds_bytes + 15u) & ~15u, cap = 160u * 1024u;
    static const int cand[6][2] = {{16, 128}, {12, 128}, {16, 64}, {12, 64}, {8, 64}, {4, 64}};
    int waves = 0, chb = 0;
    static const char *force = getenv("NEEDLE_FIND_ALL_SHAPE"); // e.g. "8x128" (tuning experiments only)
    if (force) {
        int w = 0, c = 0;
        if (sscanf(force, "%dx%d", &w, &c) == 2 && (c == 64 || c == 128) && w >= 1 && w <= 16 && p + (size_t)w * 64 * c <= cap) waves = w, chb = c;
    }
    if (!waves)
    for (const auto &c : cand)
        if (p + (size_t)c[0] * 64 * c[1] <= cap) {
            waves = c[0];
            chb = c[1];
            break;
        }
    if (!waves) return hipErrorInvalidValue;
    const uint64_t n_groups = (fa.s.n_rows + 63) >> 6;
    uint64_t blocks = (n_groups + waves - 1) / waves;
    if (blocks > (uint64_t)n_cus) blocks = (uint64_t)n_cus;
    const size_t lds = p + (size_t)waves * 64 * chb;
    return char_width == 1 ? launch_fa_m<1>(fa, chb, (int)blocks, waves, lds, stream) : launch_fa_m<2>(fa, chb, (int)blocks, waves, lds, stream);
}

} // namespace needle
