// needle_ngram.hip -- containedIn() / find() behind the n-gram candidate filter: hand-written for gfx950.
//
// The ordinary kernel (needle_scan.h) walks the automaton over every char of every row; with an automaton that fills the LDS
// (a 1000-keyword dictionary: 4487 states, 96 KB) that walk is a chain of dependent LDS lookups at 1024 chains per CU and runs
// at 0.30 of the HBM rate.  Here the text is never transposed and no lane owns a row: the batch is one byte stream, a lane
// tests the windows that end in the 16 bytes IT loaded (needle_ngram.h: no dependence between chars), and the automaton only
// runs where a window passes -- one candidate per lane, 64 candidates at a time, K + S - 1 chars each, on text re-read from
// L2 -- from the start state, K chars ahead of the window's end.  What makes that the same answer as the reference's walk from
// the row's start (DFAClassBuilder.java:438-468 indexForwards, :1004-1022 containedIn) is established on the table by
// needle_ngram_host.cpp: a restarted walk has caught up after K chars, and no first accept happens without a window in the bitmap.
//
// Per wave: 64-row groups, as in the scan kernel (one bitmap word per group).  A group is a contiguous run of 64 * stride
// bytes = a whole number of 1 KiB units (stride % 64 == 0: four units per batch); unit u is loaded as 64 lanes x 16 bytes,
// four units in flight.  Candidates -- byte offset of the window's END inside the group -- go to the wave's LDS queue; full
// sets of 64 are run as soon as they exist, the rest at the group's end.  A run that accepts reports (first accepting index,
// end, start) to its row's LDS slot by a 64-bit minimum: several windows of a row may find matches, the reference's is the one
// that accepts first.  find() takes "lengths" programs (start = end - pend[stop state], needle_lower.h) or patterns of one
// length (DFAClassBuilder.java:640-646).
#include <string.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#include "needle_walk.h"
#include "needle_ngram.h"

namespace needle {

struct NgramArgs {
    ScanArgs a;
    NgramParams ng;
    const uint32_t *ng_bitmap;
    NgramLayout lay;        // where the bitmap and the waves' queues sit in LDS (ngram_layout)
    uint32_t *stats;        // optional: [0] += candidates, [1] += KiB units of text seen by this launch
    // OP_NG_FIND_ALL (every non-overlapping match of every row, dense per-row slots: needle_find_all.h FindAllArgs)
    uint32_t fa_slots;
    uint32_t fa_kshift; // 6: group-blocked slots (needle_find_all.h FindAllArgs::kshift)
    uint32_t *fa_counts;
    int32_t *fa_starts, *fa_ends;
    uint32_t *fa_packed;
    int32_t *fa_more;
    const uint64_t *fa_offsets; // != nullptr: compact filing -- match k of row r at offsets[r] + k, room for offsets[r + 1] - offsets[r]
    uint32_t fa_count_only;     // 1: nothing is filed, every match is counted
    uint32_t dbg;           // measurement builds (-DNEEDLE_TUNING) only: NEEDLE_NG_DBG -- 1: candidates are dropped, 2: text gathered but
                            // no walk, 3: walk on zeros (no gather), +16: runs start as soon as 32 candidates wait; 0 in the product
    uint64_t *stamps;       // measurement builds only (NEEDLE_NG_STAMPS): per wave 8 x uint64 -- shader cycles (s_memtime) by section of the
                            // kernel: 0 waiting for the unit's text, 1 hashing / probing (+ issuing the next load), 2 queue pushes and loop
                            // control, 3 second-level windows, 4 verify walks, 5 a group's begin / end (slots, results), 6 staging, 7 total
    uint32_t stride_log2;   // stride_bytes is a power of two (else 0xFFFFFFFF)
    uint32_t stride_recip;  // floor(2^32 / stride_bytes)
    uint32_t char_width;    // 2: UTF-16 rows narrowed on the fly (needle_ngram.h narrow16); a.stride_bytes / a.total_bytes then count CHARS
    uint32_t page4, sub4;   // ... the pattern's page of the BMP and the byte that stands for every char outside it, in all four bytes of a dword
};

static_assert(kNgWaves == (uint32_t)kWavesPerBlock, "ngram_layout assumes the scan kernels' workgroup");
static_assert(offsetof(NgramArgs, a) == 0, "the kernel reads ScanArgs words from the kernarg segment at their own offsets (kernarg_here, needle_walk.h)");
// kernel-argument words that are needed once per group of rows (result pointers, the lengths table's place): read from the kernarg segment
// where they are used instead of living in SGPRs through the filter loop (needle_walk.h kernarg_here: the kernel spilled 24 of them)
#define NEEDLE_NG_PTR(T, member) kernarg_ptr<T>(ka, (uint32_t)offsetof(NgramArgs, member))
#define NEEDLE_NG_U32(member) kernarg_u32(ka, (uint32_t)offsetof(NgramArgs, member))
constexpr int OP_NG_FIND_ALL = 3; // (beside OP_CONTAINED_IN / OP_FIND of needle_device.h)
constexpr int kNgPF = 4;                                // units in flight per wave = units per batch

typedef u32x4 u32x4_u __attribute__((aligned(1)));
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef __attribute__((address_space(3))) uint64_t lds_u64_t;

// CW = 2: UTF-16 rows of a pattern whose chars all lie below 0xFF -- every offset, stride and length below is in CHARS, the text is
// narrowed to bytes where it is loaded (the probe stream, the candidates' pieces, the second-level windows), nothing else differs.
// WIDE (CW = 2 only): the pattern lives on several pages of the BMP -- nothing is narrowed: the windows are four 16-bit code units hashed as
// they stand (needle_ngram.h ngram_piece16), the candidates walk the UTF-16 program (two-level page map in LDS, table out of HBM / L2).
// BWD (find(), 8-bit rows, LDS-resident automata): patterns WITHOUT bounded match lengths (`(kw1|..|kw1000)[0-9]+`: no lengths automaton) --
// a verified candidate's start is indexBackwards(end - 1, 0) (DFAClassBuilder.java:529-586) by the lock-step backward_walk of needle_walk.h
// on the row's text out of L2, for the lanes whose run found a match; the program is the ordinary forward program with its backward column
// maps, a.bprog the backward table.
template <int OP, int MODE, int S, int CW = 1, bool WIDE = false, bool BWD = false>
__global__ __launch_bounds__(kWavesPerBlock * 64) void ngram_kernel(const NgramArgs A) {
    static_assert(!WIDE || (CW == 2 && MODE == MODE_GLOBAL), "the wide filter verifies on the UTF-16 HBM-table program");
    static_assert(!BWD || (OP == OP_FIND && CW == 1 && !WIDE), "backward walks: find() on 8-bit rows");
    constexpr int TW = WIDE ? 2 : 1; // width of the code units the probes and the walks see
    const ScanArgs &a = A.a;
    constexpr int NW = 16 / S; // windows per 16-byte piece
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_waves = blockDim.x >> 6;
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();
    const uint32_t bm_base = A.lay.bm_base;
    for (uint32_t i = tid * 16u; i < a.hdr.lds_bytes; i += blockDim.x * 16u) *(u32x4 *)(smem + i) = *(const u32x4 *)(a.prog + i);
    for (uint32_t i = tid * 16u; i < A.ng.bm_bytes; i += blockDim.x * 16u) *(u32x4 *)(smem + bm_base + i) = *(const u32x4 *)((const uint8_t *)A.ng_bitmap + i);
    // the second-level bitmap (5-byte windows, needle_ngram.h) rides behind the first in HBM
    const bool L2ON = A.ng.on2 != 0u; // wave-uniform (the launcher clears it where the LDS has no room for the second queue and bitmap)
    if (L2ON)
        for (uint32_t i = tid * 16u; i < A.ng.bm2_bytes; i += blockDim.x * 16u)
            *(u32x4 *)(smem + A.lay.bm2_base + i) = *(const u32x4 *)((const uint8_t *)A.ng_bitmap + A.ng.bm_bytes + i);
    __syncthreads();

    Walk wk;
    constexpr uint32_t ELEM = MODE == MODE_TABLE16 ? 2u : 1u;
    wk.ncols_e = a.hdr.n_cols * ELEM;
    wk.pad_e = wk.pre_e = wk.pad_b = wk.pre_b = 0;
    wk.win_on = a.hdr.win_on;
    wk.win_lo = a.hdr.win_lo_e;
    wk.win_hi = a.hdr.win_hi_e;
    constexpr bool FINDLIKE = OP != OP_CONTAINED_IN; // find() and find-all: first accept, then on until the automaton dies
    constexpr bool FA = OP == OP_NG_FIND_ALL;
    wk.dead_hi = FINDLIKE ? a.hdr.fa_dead_hi : 0u;
    wk.sp_chains = a.hdr.sp_chains;
    wk.sp_pad_ident = a.hdr.sp_pad_ident;
    wk.table_off = a.hdr.off_table - (MODE == MODE_SPARSE ? 0u : a.hdr.win_lo_e);
    wk.lane4 = 0;
    // MODE_GLOBAL (lower_filter_hbm: an automaton that fits the LDS in no form): the candidates' walks read the plain uint16 table out
    // of HBM / L2 -- the LDS holds the bitmap, the column map and the queues only
    wk.gtable = MODE == MODE_GLOBAL ? (const uint16_t *)(a.prog + a.hdr.off_table) : nullptr;
    wk.hot_last = 0;
    const uint32_t accept_lo = a.hdr.accept_lo, start_state = a.hdr.start;
    const uint32_t qbase = A.lay.q_base + (uint32_t)wave * (FA ? kNgWaveLdsFA + (A.ng.on2 ? kNgQueue * 4u : 0u) : kNgWaveLds);
    const uint32_t q2base = qbase + kNgQueue * 4u; // find / containedIn: the second queue (candidates that passed the second-level window)
    const uint32_t sbase = qbase + ((FA && !A.ng.on2) ? 1u : 2u) * kNgQueue * 4u; // find / containedIn: the rows' slots; find-all: two candidate slots per row ...
    const uint32_t cbase = sbase + 64u * kNgRowSlots * 8u; // ... and a counter per row
    const uint32_t mm = A.ng.m1 | A.ng.m2 << 16, amask = A.ng.addr_mask;
    const uint32_t mmB = A.ng.m1b | A.ng.m2b << 16; // (wide: the multipliers of a window's second dword)
    const uint32_t K = A.ng.warm;
#ifdef NEEDLE_TUNING
    const uint32_t dbg = A.dbg & 15u, full_set = (A.dbg & 16u) ? 32u : 64u;
    // NEEDLE_NG_STAMPS: where a wave's cycles go -- every NG_STAMP(k) books the shader cycles since the previous stamp on section k
    const bool stamps_on = A.stamps != nullptr;
    uint64_t T[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t t_last = __builtin_amdgcn_s_memtime(), t_first = t_last;
#define NG_STAMP(k)                                              \
    if (stamps_on) {                                             \
        const uint64_t n_ = __builtin_amdgcn_s_memtime();        \
        T[k] += n_ - t_last;                                     \
        t_last = n_;                                             \
    }
#define NG_WAIT_UNIT()                                                                                   \
    if (stamps_on) {                                                                                     \
        if (CW == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                    \
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                            \
    }
#else
    constexpr uint32_t dbg = 0, full_set = 64u;
#define NG_STAMP(k)
#define NG_WAIT_UNIT()
#endif
    NG_STAMP(6)
    const uint32_t stride = (uint32_t)a.stride_bytes;

    const uint64_t n_groups = (a.n_rows + 63) >> 6;
    const uint64_t wave_cnt = (uint64_t)gridDim.x * n_waves;
    uint64_t g = (uint64_t)blockIdx.x * n_waves + wave;
    if (g >= n_groups) return;
    // a group is stride / 16 units; batches are kNgPF units: where that does not divide, a group's last batch reaches into the next
    // group's text (read, masked, not used -- ngram_shape_ok bounds the waste)
    const uint32_t units_full = (((64u * stride) >> 10) + (uint32_t)(kNgPF - 1)) & ~(uint32_t)(kNgPF - 1);
    auto units_of = [&](uint64_t grp) -> uint32_t {
        if (grp + 1 < n_groups) return units_full;
        const uint32_t rows_in = (uint32_t)(a.n_rows - (grp << 6));
        return (((rows_in * stride + 1023u) >> 10) + (kNgPF - 1)) & ~(uint32_t)(kNgPF - 1);
    };
    // The prefetch cursor, one batch (kNgPF units) ahead of the one being filtered: the byte offset of its first unit is carried along
    // (+ 4 KiB per batch) and only recomputed when the cursor moves to another group, together with a flag that says whether the whole
    // group lies inside the rows -- a unit of such a group needs no clamping.
    const uint32_t lane16 = (uint32_t)lane * 16u;
    uint64_t pf_g = g, pf_base = 0;
    uint32_t pf_u = 0, pf_units = 0;
    bool pf_interior = false;
    auto pf_enter_group = [&]() __attribute__((always_inline)) {
        pf_u = 0;
        pf_base = (pf_g << 6) * a.stride_bytes;
        pf_units = pf_g < n_groups ? units_of(pf_g) : (uint32_t)kNgPF;
        pf_interior = pf_g < n_groups && pf_base + ((uint64_t)pf_units << 10) <= a.total_bytes;
    };
    // The cursor stands on a BATCH (kNgPF units of one group); unit k of it is loaded while unit k of the batch before is filtered, and the
    // cursor moves on after the batch's last unit.  A unit's load is ONE instruction -- the batch's base (SGPRs) + a per-unit lane offset
    // nb_off[k] that is set when the cursor moves: lane * 16 + k * 1024, clamped to the rows' last 16 bytes in the batch's last group(s)
    // (units past the rows read their last KiB, lanes past them their last 16 bytes -- never used).  No branch around a load: with one the
    // compiler's vmcnt bookkeeping gives up and waits for EVERY load in flight (measured: + 4.5 %).  (Round 5's per-UNIT cursor cost ~18
    // instructions and two branches per KiB -- a quarter of the filter phase: profiles/r06_filter_trace.md.)
    struct Raw { u32x4 lo, hi; }; // 16 chars as loaded (CW = 1: lo only)
    const uint8_t *nb_ptr = a.rows;
    uint32_t nb_off[kNgPF]; // in bytes
    auto set_batch = [&]() __attribute__((always_inline)) {
        uint64_t base = pf_base;
        uint32_t room = 0xFFFFFFFFu; // chars between the batch's base and the last place a 16-char load may start
        if (!pf_interior) { // wave-uniform: the rows' last group(s), or a prefetch past their end
            const uint64_t last = a.total_bytes - 16u;
            base = base < last ? base : last;
            const uint64_t r = last - base;
            room = r < 0x7FFFFFFFull ? (uint32_t)r : 0x7FFFFFFFu;
        }
        nb_ptr = a.rows + base * CW;
#pragma unroll
        for (int k = 0; k < kNgPF; ++k) {
            const uint32_t o = lane16 + (uint32_t)k * 1024u;
            nb_off[k] = (o < room ? o : room) * CW;
        }
    };
    auto load_unit = [&](int k) __attribute__((always_inline)) -> Raw {
        // (tried for MODE_GLOBAL, whose walks read the table out of L2: nontemporal text loads -- c3x 1.09 -> 1.19 ms: the candidates' own
        // text then never hits the L2 either)
        Raw v;
        const uint8_t *src = nb_ptr + nb_off[k];
        v.lo = *(const u32x4 *)src;
        if (CW == 2) v.hi = *(const u32x4 *)(src + 16);
        return v;
    };
    auto advance_batch = [&]() __attribute__((always_inline)) {
        pf_u += (uint32_t)kNgPF;
        pf_base += 1024u * kNgPF;
        if (pf_u >= pf_units) {
            pf_g += wave_cnt;
            pf_enter_group();
        }
        set_batch();
    };
    pf_enter_group();
    set_batch();
    Raw R[kNgPF];
#pragma unroll
    for (int k = 0; k < kNgPF; ++k) R[k] = load_unit(k);
    advance_batch();
    // 16 chars of text at p (unaligned) as 16 bytes
    auto text16 = [&](const uint8_t *p) __attribute__((always_inline)) -> u32x4 {
        if (CW == 1) return *(const u32x4_u *)p;
        return narrow16(*(const u32x4_u *)p, *(const u32x4_u *)(p + 16), A.page4, A.sub4);
    };

    // Run the automaton for one row per lane from the start state: chars [r, ..) of row `row` of group grp, looking for a FIRST accept
    // at indexes qn .. lim0 - 1 (after it the walk runs on until the automaton dies: the reference's lastMatch), and report to the
    // row's slot.  A candidate: r = K chars ahead of the window's end qn, lim0 = qn + S - 1.  A whole row (the flood fallback below):
    // r = qn = 0, lim0 = the row's length.  Text comes from memory (L2, mostly) 16 bytes at a time: the first piece wherever r is,
    // the following ones aligned (stride % 16 == 0: inside the row), chars already walked skipped.
    struct Hit {
        bool found, died; // died: the automaton died without a first accept at or after qn (it had passed an earlier match)
        bool crossed;     // find-all: the run passed through an accepting state BEFORE qn -- it crossed an earlier match, where the reference
                          // restarts (and its search automaton prunes the restart threads: DFA_SEARCH keeps the higher-priority longer
                          // alternative only) -- so what it says about its window is not the reference's walk
        uint32_t first, last;
        int32_t start;
    };
    auto walk_row = [&](uint64_t grp, uint32_t row, bool valid, uint32_t qn, uint32_t r, uint32_t lim0) __attribute__((always_inline)) -> Hit {
        const uint64_t grow = (grp << 6) + row;
        uint32_t len = a.row_len;
        const KernargPtr ka = kernarg_here();
        const uint32_t *const lens = NEEDLE_NG_PTR(const uint32_t, a.lengths);
        if (lens) len = valid ? lens[grow] : 0u;
        valid = valid && qn <= len;
        const uint64_t rowabs = grow * a.stride_bytes;
        const uint8_t *rowp = a.rows + (valid ? rowabs : 0ull) * CW;
        uint32_t lim = lim0 < len ? lim0 : len;
        uint32_t st = start_state, last = 0, first = 0;
        bool found = false, over = !valid, died = false, crossed = false;
        // the piece being walked starts at `base`; chars before `cur` are not walked: the walk starts AT r, in the start state
        // (EVERY 16-byte read stays inside the batch: in its last 16 bytes a piece starts earlier and the chars before `cur` are
        // skipped -- also for the lanes that are over and only ride along while others walk on)
        uint32_t cur = valid ? r : 0u, base = cur;
        const uint64_t room = a.total_bytes - 16u - (valid ? rowabs : 0ull);
        base = (uint64_t)base < room ? base : (uint32_t)room;
        auto step = [&](uint32_t colv, uint32_t pos) __attribute__((always_inline)) {
            const bool go = !over && pos >= cur && pos < lim;
            const uint32_t ns = apply<MODE, TW>(wk, st, colv);
            st = go ? ns : st;
            const bool acc_any = go && st >= accept_lo;
            const bool acc = acc_any && pos + 1u >= qn;
            if (FA) crossed = crossed || (acc_any && !acc);
            if (FINDLIKE) {
                last = acc ? pos + 1u : last;
                first = (acc && !found) ? pos + 1u : first;
                lim = acc ? len : lim; // after the first accept the walk runs on until the automaton dies
                found = found || acc;
                died = died || (go && st <= wk.dead_hi);
                over = over || died;
            } else {
                found = found || acc;
                over = over || acc;
            }
        };
        for (;;) {
            uint32_t col[16];
            if (WIDE) { // 16 code units = two 16-byte pieces, their columns through the page map
                const u32x4 t0 = *(const u32x4_u *)(rowp + base * 2u), t1 = *(const u32x4_u *)(rowp + base * 2u + 16u);
                const uint32_t w0[4] = {t0[0], t0[1], t0[2], t0[3]}, w1[4] = {t1[0], t1[1], t1[2], t1[3]};
                uint32_t c0[8], c1[8];
                piece_lookups<MODE, 2, false>(wk, w0, 0u, 0u, 0u, c0);
                piece_lookups<MODE, 2, false>(wk, w1, 0u, 0u, 0u, c1);
#pragma unroll
                for (int k = 0; k < 8; ++k) col[k] = c0[k], col[8 + k] = c1[k];
            } else {
            u32x4 tx = {0, 0, 0, 0};
            if (dbg != 3u) tx = text16(rowp + base * CW);
            if (dbg == 2u) over = over || tx[0] != 0x12345678u; // (the text is waited for, the walk is not taken)
            const uint32_t w[4] = {tx[0], tx[1], tx[2], tx[3]};
            piece_lookups<MODE, 1, false>(wk, w, 0u, 0u, 0u, col);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                step(col[k], base + (uint32_t)k);
                if ((k >= 7 || (k & 3) == 3) && k != 15 && __ballot(!over && base + (uint32_t)k + 1u < lim) == 0ull) break; // (K + S - 1 = 9 or 10 steps is the usual run)
            }
            const uint32_t done_to = base + 16u;
            cur = cur > done_to ? cur : done_to;
            if (__ballot(!over && cur < lim) == 0ull) break; // (rare for a candidate: a match that runs past its 16 bytes)
            base = done_to & ~15u;
            base = (uint64_t)base < room ? base : (uint32_t)room;
        }
        Hit h;
        h.found = found, h.died = died && !found, h.crossed = crossed, h.first = first, h.last = last, h.start = 0;
        if (FINDLIKE) {
            const int32_t fixed_len = (int32_t)NEEDLE_NG_U32(a.fixed_len);
            if (BWD) {
                h.start = backward_walk<1>(a, found, (int32_t)last, 0, 16u, 0u, 0u, 0u, rowp); // (no window in LDS: every char from memory)
            } else if (fixed_len >= 0) {
                h.start = (int32_t)last - fixed_len; // :640-646
            } else {
                uint32_t pidx = st;
                if (MODE == MODE_SPARSE) { // (needle_scan.h finish_rows: a live stop state asks its END record)
                    const uint32_t st_end = sparse_end<1>(wk, st, found && st > wk.dead_hi, NEEDLE_NG_U32(a.hdr.sp_end_col4));
                    pidx = (st_end & 0xFFFFu) - NEEDLE_NG_U32(a.hdr.sp_dead_row0);
                }
                const uint32_t len_off = NEEDLE_NG_U32(a.hdr.fa_len_off);
                if (MODE == MODE_GLOBAL) h.start = (int32_t)last - (int32_t)a.prog[len_off + (found ? pidx : 0u)]; // (pend[] behind the table)
                else h.start = (int32_t)last - (int32_t)lds_u8(len_off + (found ? pidx : 0u));
            }
        }
        return h;
    };
    // A candidate (find / containedIn): its run reports to the row's slot.  find-all: runs that found a match are filed with their row
    // {window end, first - end | last, length} -- the first two of a row in its slots, all of them counted; the rows sort them out at
    // the group's end.
    auto run_rows = [&](uint64_t grp, uint32_t row, bool valid, uint32_t qn, uint32_t r, uint32_t lim0) __attribute__((always_inline)) {
        const Hit h = walk_row(grp, row, valid, qn, r, lim0);
        if (FA) {
            // (a run whose automaton died on the way to its window -- it crossed an earlier match, after which the reference restarts
            // and this run did not -- knows nothing about the window: filed as such, the row runs it again from its cursor)
            // (so does a run that CROSSED an accept before its window and lived on: in a post-accept state the search automaton has
            // dropped the restart threads, e.g. `international|inter|nation` on "internationa..": the run for "tion" passes "inter",
            // stays alive for "international" and never sees "nation")
            if (h.found || h.died || h.crossed) {
                const uint32_t ord = __hip_atomic_fetch_add((lds_u32_t *)(uintptr_t)(cbase + row * 4u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                const uint64_t ent = (uint64_t)qn << 48 | (uint64_t)((h.found && !h.crossed) ? h.first - qn : 0xFFu) << 32 | (uint64_t)(h.last & 0xFFFFu) << 16 |
                                     (uint64_t)((uint32_t)((int32_t)h.last - h.start) & 0xFFFFu);
                if (ord < kNgRowSlots) *(lds_u64_t *)(uintptr_t)(sbase + (row * kNgRowSlots + ord) * 8u) = ent;
            }
        } else if (OP == OP_FIND) {
            if (h.found) {
                const uint64_t key = (uint64_t)h.first << 32 | (uint64_t)h.last << 16 | (uint64_t)(uint32_t)h.start;
                __hip_atomic_fetch_min((lds_u64_t *)(uintptr_t)(sbase + row * 8u), key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        } else if (h.found) {
            __hip_atomic_fetch_or((lds_u64_t *)(uintptr_t)sbase, 1ull << row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    };

    // window end `e` (byte offset inside the group) -> its row and the offset qn inside the row
    auto locate = [&](uint32_t e, uint32_t &row, uint32_t &qn) __attribute__((always_inline)) {
        const uint32_t em1 = e - 1u;
        if (A.stride_log2 != 0xFFFFFFFFu) {
            row = em1 >> A.stride_log2;
        } else {
            row = __umulhi(em1, A.stride_recip);
            if (em1 - row * stride >= stride) ++row;
        }
        qn = e - row * stride; // window [qn - 4, qn) of the row
    };
    // the second-level window of a candidate (valid: row < rows_in, qn >= 4): text bytes [qn - 5, qn) hashed into the second bitmap
    // (needle_ngram.h); candidates within 5 chars of the row's start pass as they are
    const uint32_t bm2_base = A.lay.bm2_base, amask2 = A.ng.addr_mask2, m3 = A.ng.m3;
    auto level2 = [&](uint64_t grp, uint32_t row, uint32_t qn) __attribute__((always_inline)) -> bool {
        typedef uint32_t u32_u __attribute__((aligned(1)));
        const uint8_t *rowp = a.rows + ((grp << 6) + row) * a.stride_bytes * CW;
        const bool deep = qn >= 5u;
        uint32_t w, c5;
        if (WIDE) { // the window's two dwords as they stand (qn is even: they are aligned) + the unit in front of them
            typedef uint16_t u16_u __attribute__((aligned(1)));
            const uint32_t x0 = *(const u32_u *)(rowp + (qn - 4u) * 2u), x1 = *(const u32_u *)(rowp + (qn - 2u) * 2u);
            c5 = deep ? (uint32_t)*(const u16_u *)(rowp + (qn - 5u) * 2u) : 0u;
            return !deep || ngram_probe2_16(x0, x1, c5, mm, mmB, m3, amask2, bm2_base) != 0u;
        }
        // two-sided (NgramParams::on2 == 2): also the 5 chars that end one char BEHIND the window -- chars [qn - 4, qn + 1): the window's
        // first char in front of its last three and the char at qn.  (That char may lie behind the row's end -- then this window cannot be
        // the one S - 1 ahead of an accept, and whatever the probe says only lets a candidate through to the automaton; the read itself is
        // kept inside the batch.)
        const bool two = A.ng.on2 == 2u; // wave-uniform
        const uint32_t fo = (two && ((grp << 6) + row) * a.stride_bytes + qn < a.total_bytes) ? 1u : 0u;
        uint32_t wf = 0;
        if (CW == 1) {
            w = *(const u32_u *)(rowp + qn - 4u);
            c5 = deep ? (uint32_t)rowp[qn - 5u] : 0u;
            if (two) wf = *(const u32_u *)(rowp + qn - 4u + fo);
        } else {
            typedef uint16_t u16_u __attribute__((aligned(1)));
            w = narrow_pair_patched(*(const u32_u *)(rowp + (qn - 4u) * 2u), *(const u32_u *)(rowp + (qn - 2u) * 2u), A.page4, A.sub4);
            c5 = deep ? (uint32_t)*(const u16_u *)(rowp + (qn - 5u) * 2u) : 0u;
            c5 = (c5 >> 8) == (A.page4 & 0xFFu) ? (c5 & 0xFFu) : (A.sub4 & 0xFFu);
            if (two) wf = narrow_pair_patched(*(const u32_u *)(rowp + (qn - 4u + fo) * 2u), *(const u32_u *)(rowp + (qn - 2u + fo) * 2u), A.page4, A.sub4);
        }
        if (!deep) return true;
        bool pass = ngram_probe2(w, c5, mm, m3, amask2, bm2_base) != 0u;
        if (two) pass = pass || (fo != 0u && ngram_probe2(wf, w & 0xFFu, mm, m3, amask2, bm2_base) != 0u);
        return pass;
    };
    uint32_t q2head = 0, q2tail = 0; // wave-uniform: the second queue
    uint32_t qhead = 0, qtail = 0; // wave-uniform
    uint32_t n_cand = 0, n_units = 0; // what this wave saw: candidates, KiB units of text (-> A.stats: the host's flood watch)
    for (; g < n_groups; g += wave_cnt) {
        const uint32_t rows_in = (g + 1 < n_groups) ? 64u : (uint32_t)(a.n_rows - (g << 6));
        const uint32_t gbytes = rows_in * stride;
        const uint32_t units = units_of(g);
        n_units += units;
        NG_STAMP(5)
        // ---- the group's result slots
        if (FA) *(lds_u32_t *)(uintptr_t)(cbase + (uint32_t)lane * 4u) = 0u;
        else if (OP == OP_FIND) *(lds_u64_t *)(uintptr_t)(sbase + (uint32_t)lane * 8u) = ~0ull;
        else if (lane == 0) *(lds_u64_t *)(uintptr_t)sbase = 0ull;
        uint32_t carry = 0; // (the window reaching back from a row's first bytes is dropped below: what it holds does not matter)
        // run the automaton on the second queue's candidates, 64 at a time, while at least `at_least` wait
        auto drain2 = [&](uint32_t at_least) __attribute__((always_inline)) {
            while (q2tail - q2head >= at_least && q2tail != q2head) {
                const uint32_t n_take = q2tail - q2head < 64u ? q2tail - q2head : 64u;
                const bool act = (uint32_t)lane < n_take;
                uint32_t e = *(const lds_u32_t *)(uintptr_t)(q2base + (((q2head + (uint32_t)lane) & (kNgQueue - 1u)) << 2));
                q2head += n_take;
                e = act ? e : 4u;
                uint32_t row, qn;
                locate(e, row, qn);
                run_rows(g, row, act && row < rows_in && qn >= 4u, qn, qn > K ? qn - K : 0u, qn + (uint32_t)S - 1u);
            }
        };
        for (uint32_t u0 = 0; u0 < units; u0 += kNgPF) {
            // ---- filter: four units, each slot re-loaded for the batch after next as soon as it is read
            uint32_t log = 0;
#pragma unroll
            for (int k = 0; k < kNgPF; ++k) {
                NG_STAMP(2)
                NG_WAIT_UNIT()
                NG_STAMP(0)
                const Raw raw = R[k];
                asm volatile("" ::: "memory");
                R[k] = load_unit(k);
                if (k == kNgPF - 1) advance_batch();
                asm volatile("" ::: "memory");
                if (dbg == 4u) { // (measurement builds: the text is loaded and waited for, nothing is probed)
                    log |= raw.lo[0] == 0x12345678u ? 1u : 0u;
                    if (CW == 2) log |= raw.hi[0] == 0x12345678u ? 1u : 0u;
                    continue;
                }
                if (WIDE) {
                    const uint32_t pw = ngram_prev_dword(raw.hi[3], carry);
                    carry = (uint32_t)__builtin_amdgcn_readlane((int)raw.hi[3], 63);
                    log = ngram_piece16<S>(log, pw, raw.lo, raw.hi, mm, mmB, amask, bm_base);
                } else {
                const u32x4 v = CW == 1 ? raw.lo : narrow16(raw.lo, raw.hi, A.page4, A.sub4);
                const uint32_t pw = ngram_prev_dword(v[3], carry);
                carry = (uint32_t)__builtin_amdgcn_readlane((int)v[3], 63);
                log = ngram_piece<S>(log, pw, v[0], v[1], v[2], v[3], mm, amask, bm_base);
                }
                NG_STAMP(1)
            }
            NG_STAMP(1)
            const uint32_t po0 = (u0 << 10) + lane16; // byte offset of this lane's piece of the batch's first unit
            if (NW * kNgPF < 32) log >>= 32 - NW * kNgPF; // window wi of unit j at bit j * NW + wi
            if (gbytes < ((u0 + kNgPF) << 10)) { // wave-uniform: the batch's last group: pieces past its rows hold nothing
#pragma unroll
                for (int k = 0; k < kNgPF; ++k)
                    if (po0 + ((uint32_t)k << 10) >= gbytes) log &= ~(((1u << NW) - 1u) << (k * NW));
            }
            // ---- candidates to the queue; full sets of 64 run at once, the rest with the group's last batch
            const bool last_batch = u0 + kNgPF >= units;
            for (;;) {
                const uint64_t any = __ballot(log != 0u);
                if (any != 0ull) {
                    const bool has = log != 0u;
                    const uint32_t b = (uint32_t)__builtin_ctz(log | 0x80000000u);
                    const uint32_t e = po0 + ((b / NW) << 10) + ((b % NW) + 1u) * S; // end of the window inside the group
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(any >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)any, 0u));
                    if (has) *(lds_u32_t *)(uintptr_t)(qbase + (((qtail + rank) & (kNgQueue - 1u)) << 2)) = e;
                    qtail += (uint32_t)__builtin_popcountll(any);
                    n_cand += (uint32_t)__builtin_popcountll(any);
                    log &= log - 1u;
                }
                const bool more = __ballot(log != 0u) != 0ull;
                const uint32_t thr = (more || !last_batch) ? full_set : 1u;
                if (dbg == 1u) qhead = qtail;
                while (qtail - qhead >= thr) {
                    // ---- up to 64 candidates, one per lane
                    const uint32_t n_take = qtail - qhead < 64u ? qtail - qhead : 64u;
                    const bool act = (uint32_t)lane < n_take;
                    uint32_t e = *(const lds_u32_t *)(uintptr_t)(qbase + (((qhead + (uint32_t)lane) & (kNgQueue - 1u)) << 2));
                    qhead += n_take;
                    e = act ? e : 4u;
                    uint32_t row, qn;
                    locate(e, row, qn);
                    const bool valid = act && row < rows_in && qn >= 4u;
                    if (!L2ON) { // ---- run the automaton on them
                        NG_STAMP(2)
                        run_rows(g, row, valid, qn, qn > K ? qn - K : 0u, qn + (uint32_t)S - 1u);
                        NG_STAMP(4)
                        continue;
                    }
                    NG_STAMP(2)
                    // ---- second level: the 5-byte window [qn - 5, qn) -- 8 bytes of the candidate's text from memory, one more probe;
                    // what passes (on random text 1 in 27 of the first level's chance hits, and every real keyword tail) waits in the
                    // second queue until 64 of them make a run worth its ~10 dependent lookups
                    const bool pass = valid && level2(g, row, qn);
                    const uint64_t pm = __ballot(pass);
                    const uint32_t prank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                    if (pass) *(lds_u32_t *)(uintptr_t)(q2base + (((q2tail + prank) & (kNgQueue - 1u)) << 2)) = e;
                    q2tail += (uint32_t)__builtin_popcountll(pm);
                    NG_STAMP(3)
                    drain2((thr == 1u && qtail == qhead) ? 1u : 64u);
                    NG_STAMP(4)
                }
                NG_STAMP(2)
                if (L2ON && thr == 1u) drain2(1u); // the group's end: whatever still waits
                NG_STAMP(4)
                if (!more) break;
            }
        }
        // ---- the group's verdicts: lane = row
        NG_STAMP(2)
        asm volatile("" ::: "memory");
        if (FA) {
            // Every non-overlapping match of the row, as the reference's repeated find() reports them (DFAClassBuilder.java:616-659): after
            // a match the search restarts AT its end.  A filed candidate's run started K chars ahead of its window: where that is at or
            // after the row's cursor it IS the reference's walk (the run-up argument of needle_ngram_host.cpp) and its match is the next
            // one; where it is not -- two matches within K chars of each other, a match overlapping the one before -- the window is run
            // again from the cursor, and so is a window whose run died on the way (it crossed an earlier match).  (A window whose run
            // stayed alive and found nothing has nothing from the cursor either: both walks are in the same state by then.)  Rows with more
            // than two filed candidates are searched from the cursor match by match -- exact, slow, rare on the text the filter is for.
            const bool row_ok = (uint32_t)lane < rows_in;
            const uint32_t n_mine = *(const lds_u32_t *)(uintptr_t)(cbase + (uint32_t)lane * 4u);
            uint64_t e0 = ~0ull, e1 = ~0ull; // this row's filed candidates, by window end
            if (n_mine >= 1u) e0 = *(const lds_u64_t *)(uintptr_t)(sbase + ((uint32_t)lane * kNgRowSlots) * 8u);
            if (n_mine >= 2u) e1 = *(const lds_u64_t *)(uintptr_t)(sbase + ((uint32_t)lane * kNgRowSlots + 1u) * 8u);
            if (e1 < e0) {
                const uint64_t t = e0;
                e0 = e1, e1 = t;
            }
            bool slow = row_ok && n_mine > kNgRowSlots;
            uint32_t cursor = 0, cnt = 0;
            bool more_f = false;
            uint64_t out0 = A.fa_kshift ? g * (uint64_t)A.fa_slots * 64u + (uint64_t)lane : ((g << 6) + lane) * (uint64_t)A.fa_slots;
            uint32_t cap = A.fa_count_only ? 0xFFFFFFFFu : A.fa_slots;
            if (A.fa_offsets) {
                out0 = row_ok ? A.fa_offsets[(g << 6) + lane] : 0ull;
                cap = row_ok ? (uint32_t)(A.fa_offsets[(g << 6) + lane + 1] - out0) : 0u;
            }
            auto emit = [&](bool hit, uint32_t last, int32_t start) __attribute__((always_inline)) {
                const bool file = hit && cnt < cap;
                more_f = more_f || (hit && !file);
                if (file && !A.fa_count_only) {
                    const uint64_t o = out0 + ((uint64_t)cnt << (A.fa_offsets ? 0u : A.fa_kshift));
                    if (A.fa_packed) {
                        A.fa_packed[o] = (uint32_t)start | (last << 16);
                    } else {
                        A.fa_starts[o] = start;
                        A.fa_ends[o] = (int32_t)last;
                    }
                }
                cnt += file ? 1u : 0u;
                cursor = file ? last : cursor;
            };
#pragma unroll 1
            for (int j = 0; j < (int)kNgRowSlots; ++j) { // (a loop, not unrolled: one copy of the re-run)
                const uint64_t x = e0;
                e0 = e1, e1 = ~0ull;
                const bool have = row_ok && !slow && !more_f && x != ~0ull;
                const uint32_t e = (uint32_t)(x >> 48), last = (uint32_t)((x >> 16) & 0xFFFFu), mlen = (uint32_t)(x & 0xFFFFu);
                const bool unknown = ((x >> 32) & 0xFFu) == 0xFFu;
                const uint32_t r0 = e > K ? e - K : 0u;
                const bool exact = have && !unknown && r0 >= cursor;
                const bool again = have && !exact && e + (uint32_t)S - 1u > cursor;
                Hit h2;
                h2.found = false, h2.died = false, h2.crossed = false, h2.first = 0, h2.last = 0, h2.start = 0;
                if (__ballot(again) != 0ull) h2 = walk_row(g, (uint32_t)lane, again, e > cursor ? e : cursor + 1u, cursor, e + (uint32_t)S - 1u);
                // (a re-run from the cursor IS the reference's search; should it accept before this window -- a match whose own window
                // was never filed -- the row is searched match by match below: exact whatever the filter missed)
                const bool lost = again && h2.crossed;
                slow = slow || lost;
                emit(exact || (again && h2.found && !lost), exact ? last : h2.last, exact ? (int32_t)(last - mlen) : h2.start);
            }
            while (__ballot(slow && !more_f) != 0ull) { // the reference's loop, one find() at a time, for the rows that need it
                const bool todo = slow && !more_f;
                const Hit h = walk_row(g, (uint32_t)lane, todo, cursor, cursor, 0xFFFFFFFFu);
                emit(todo && h.found, h.last, h.start);
                slow = todo && h.found && !more_f;
            }
            if (row_ok && A.fa_counts) A.fa_counts[(g << 6) + lane] = cnt;
            if (__ballot(more_f) != 0ull && lane == 0) *A.fa_more = 1;
        } else if (OP == OP_FIND) {
            const uint64_t key = *(const lds_u64_t *)(uintptr_t)(sbase + (uint32_t)lane * 8u);
            const bool row_ok = (uint32_t)lane < rows_in;
            const bool res = row_ok && key != ~0ull;
            const uint64_t word = __ballot(res);
            const KernargPtr ka = kernarg_here();
            if (lane == 0) NEEDLE_NG_PTR(uint64_t, a.bitmap)[g] = word;
            if (row_ok) {
                uint32_t *const o_packed = NEEDLE_NG_PTR(uint32_t, a.packed);
                if (o_packed && NEEDLE_NG_U32(a.packed8)) { // one uint16 per row (rows <= 256 chars)
                    ((uint16_t *)o_packed)[(g << 6) + lane] = pack8(res ? (int32_t)(key & 0xFFFFu) : -1, res ? (int32_t)((key >> 16) & 0xFFFFu) : -1);
                } else if (o_packed) { // the key's low dword is end << 16 | start already; ~0 = no match
                    o_packed[(g << 6) + lane] = (uint32_t)key;
                } else {
                    NEEDLE_NG_PTR(int32_t, a.start)[(g << 6) + lane] = res ? (int32_t)(key & 0xFFFFu) : -1;
                    NEEDLE_NG_PTR(int32_t, a.end)[(g << 6) + lane] = res ? (int32_t)((key >> 16) & 0xFFFFu) : -1;
                }
            }
        } else if (lane == 0) {
            const KernargPtr ka = kernarg_here();
            NEEDLE_NG_PTR(uint64_t, a.bitmap)[g] = *(const lds_u64_t *)(uintptr_t)sbase;
        }
        asm volatile("" ::: "memory");
    }
    NG_STAMP(5)
#ifdef NEEDLE_TUNING
    if (stamps_on && lane == 0) {
        uint64_t *o = A.stamps + ((uint64_t)blockIdx.x * kWavesPerBlock + (uint32_t)wave) * 8u;
        T[7] = t_last - t_first;
        for (int k = 0; k < 8; ++k) o[k] = T[k];
    }
#endif
    // what the host's flood watch reads (needle_api.cpp): candidates and KiB of text of this launch
    {
        const KernargPtr ka = kernarg_here();
        uint32_t *const o_stats = NEEDLE_NG_PTR(uint32_t, stats);
        if (o_stats && lane == 0) {
            atomicAdd(&o_stats[0], n_cand);
            atomicAdd(&o_stats[1], n_units);
        }
    }
}

template <int OP, int MODE, int S, int CW, bool WIDE = false, bool BWD = false>
static hipError_t launch_ng(const NgramArgs &A, int n_cus, size_t lds, hipStream_t stream) {
    auto k = ngram_kernel<OP, MODE, S, CW, WIDE, BWD>;
    static thread_local uint64_t configured = 0;
    if (hipError_t e = allow_full_lds((const void *)k, configured); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(n_cus), dim3(kWavesPerBlock * 64), lds, stream, A);
    return hipGetLastError();
}

template <int OP, int MODE>
static hipError_t launch_ng_s(const NgramArgs &A, int n_cus, size_t lds, hipStream_t stream) {
    if (A.char_width == 2) return A.ng.stride == 4 ? launch_ng<OP, MODE, 4, 2>(A, n_cus, lds, stream) : launch_ng<OP, MODE, 2, 2>(A, n_cus, lds, stream);
    if constexpr (OP == OP_FIND) {
        if (A.a.bprog) // find() of a pattern without bounded match lengths: starts by backward walks
            return A.ng.stride == 4 ? launch_ng<OP, MODE, 4, 1, false, true>(A, n_cus, lds, stream) : launch_ng<OP, MODE, 2, 1, false, true>(A, n_cus, lds, stream);
    }
    return A.ng.stride == 4 ? launch_ng<OP, MODE, 4, 1>(A, n_cus, lds, stream) : launch_ng<OP, MODE, 2, 1>(A, n_cus, lds, stream);
}

template <int OP>
static hipError_t launch_ng_m(const NgramArgs &A, int n_cus, size_t lds, hipStream_t stream) {
    if (A.ng.wide) { // UTF-16 rows, windows of four code units, walks on the UTF-16 HBM-table program (lower_filter_wide)
        if (A.char_width != 2 || A.a.hdr.mode != MODE_GLOBAL) return hipErrorInvalidValue;
        return A.ng.stride == 4 ? launch_ng<OP, MODE_GLOBAL, 4, 2, true>(A, n_cus, lds, stream) : launch_ng<OP, MODE_GLOBAL, 2, 2, true>(A, n_cus, lds, stream);
    }
    switch (A.a.hdr.mode) {
    case MODE_TABLE8: return launch_ng_s<OP, MODE_TABLE8>(A, n_cus, lds, stream);
    case MODE_TABLE16: return launch_ng_s<OP, MODE_TABLE16>(A, n_cus, lds, stream);
    case MODE_SPARSE: return launch_ng_s<OP, MODE_SPARSE>(A, n_cus, lds, stream);
    case MODE_GLOBAL: return launch_ng_s<OP, MODE_GLOBAL>(A, n_cus, lds, stream);
    default: return hipErrorInvalidValue;
    }
}

// LDS a launch takes; 0 = does not fit (the caller keeps the ordinary kernel)
size_t ngram_lds_bytes(const ProgHeader &h, const NgramParams &ng) {
    NgramLayout l;
    if (ng.on2 && ngram_layout(h.lds_bytes, ng.bm_bytes, &l, kNgWaveLds, ng.bm2_bytes)) return l.total;
    return ngram_layout(h.lds_bytes, ng.bm_bytes, &l) ? l.total : 0;
}

// Whether this batch shape can take the filter kernel at all: 8-bit rows 64 .. 4096 bytes apart in steps of 16 (a 64-row group is then
// stride / 16 KiB units), where rounding a group up to whole batches of kNgPF units wastes at most a quarter of the reads.
bool ngram_shape_ok(const ScanArgs &a) {
    const uint64_t units = a.stride_bytes / 16, rounded = (units + (kNgPF - 1)) & ~(uint64_t)(kNgPF - 1);
    return a.stride_bytes % 16 == 0 && a.stride_bytes >= 64 && a.stride_bytes <= 4096 && rounded * 4 <= units * 5 && a.total_bytes >= 16384 &&
           a.from == nullptr && a.end_state == nullptr && a.row_len <= 65535u;
}

static hipError_t launch_ngram_any(int op, const ScanArgs &a, const NgramParams &ng, const uint32_t *d_bitmap, uint32_t *d_stats, int n_cus, hipStream_t stream,
                                   const NgramArgs *fa, int char_width = 1, int page = 0, int sub = 0xFF);

// char_width 2: UTF-16 rows behind the BYTE program's filter (patterns below 0xFF only: the caller checks) -- a.stride_bytes and
// a.total_bytes count chars then
hipError_t launch_ngram(int op, const ScanArgs &a, const NgramParams &ng, const uint32_t *d_bitmap, uint32_t *d_stats, int n_cus, hipStream_t stream,
                        int char_width, int page, int sub) {
    return launch_ngram_any(op, a, ng, d_bitmap, d_stats, n_cus, stream, nullptr, char_width, page, sub);
}

// LDS of the find-all form; 0 = does not fit
size_t ngram_find_all_lds_bytes(const ProgHeader &h, const NgramParams &ng) {
    NgramLayout l;
    if (ng.on2 && ngram_layout(h.lds_bytes, ng.bm_bytes, &l, kNgWaveLdsFA + kNgQueue * 4u, ng.bm2_bytes)) return l.total;
    return ngram_layout(h.lds_bytes, ng.bm_bytes, &l, kNgWaveLdsFA) ? l.total : 0;
}

// Every non-overlapping match of every row behind the filter (dense per-row slots, compact filing or counting only; needle_find_all.h FindAllArgs): `a` carries the
// rows and the lengths program, the outputs are the find-all ones.
hipError_t launch_ngram_find_all(const ScanArgs &a, const NgramParams &ng, const uint32_t *d_bitmap, uint32_t *d_stats, uint32_t slots, uint32_t *counts,
                                 int32_t *starts, int32_t *ends, uint32_t *packed, int32_t *more, const uint64_t *offsets, bool count_only, int n_cus,
                                 hipStream_t stream, int char_width, int page, int sub, uint32_t kshift) {
    NgramArgs F;
    memset(&F, 0, sizeof(F));
    F.fa_kshift = offsets ? 0u : kshift;
    F.fa_slots = slots, F.fa_counts = counts, F.fa_starts = starts, F.fa_ends = ends, F.fa_packed = packed, F.fa_more = more;
    F.fa_offsets = offsets, F.fa_count_only = count_only ? 1u : 0u;
    return launch_ngram_any(OP_NG_FIND_ALL, a, ng, d_bitmap, d_stats, n_cus, stream, &F, char_width, page, sub);
}

static hipError_t launch_ngram_any(int op, const ScanArgs &a, const NgramParams &ng, const uint32_t *d_bitmap, uint32_t *d_stats, int n_cus, hipStream_t stream,
                                   const NgramArgs *fa, int char_width, int page, int sub) {
    NgramArgs A;
    memset(&A, 0, sizeof(A));
    if (fa) A = *fa;
    A.char_width = (uint32_t)char_width;
    A.page4 = (uint32_t)(page & 255) * 0x01010101u, A.sub4 = (uint32_t)(sub & 255) * 0x01010101u;
    A.a = a;
    A.ng = ng;
    A.ng_bitmap = d_bitmap;
    A.stats = d_stats;
    const uint32_t stride = (uint32_t)a.stride_bytes;
    A.stride_log2 = 0xFFFFFFFFu;
    if ((stride & (stride - 1u)) == 0u) A.stride_log2 = (uint32_t)__builtin_ctz(stride);
    A.stride_recip = (uint32_t)((1ull << 32) / stride);
    // the second level needs its bitmap and a second queue per wave in LDS: find / containedIn were sized for it by the host; the find-all
    // form (two slots + a counter per row beside the queues) takes it where it still fits (walks out of HBM: yes; a 96 KB automaton: no)
    const uint32_t wb1 = op == OP_NG_FIND_ALL ? kNgWaveLdsFA : kNgWaveLds;
    const uint32_t wb2 = op == OP_NG_FIND_ALL ? kNgWaveLdsFA + kNgQueue * 4u : kNgWaveLds;
    if (ng.on2 && !ngram_layout(a.hdr.lds_bytes, ng.bm_bytes, &A.lay, wb2, ng.bm2_bytes)) A.ng.on2 = 0;
    if (!A.ng.on2 && !ngram_layout(a.hdr.lds_bytes, ng.bm_bytes, &A.lay, wb1)) return hipErrorInvalidValue;
    if (ng.addr_shift != 24u) return hipErrorInvalidValue;
    A.dbg = 0;
#ifdef NEEDLE_TUNING
    static const uint32_t dbg_env = getenv("NEEDLE_NG_DBG") ? (uint32_t)atoi(getenv("NEEDLE_NG_DBG")) : 0u;
    A.dbg = dbg_env;
#endif
    const size_t lds = A.lay.total;
#ifdef NEEDLE_TUNING
    // NEEDLE_NG_STAMPS=1: every launch is followed by a device synchronisation and one line on stderr -- the waves' shader cycles by
    // section (the kernel's NG_STAMP points), summed over all waves, as shares of their total lifetime
    static const bool stamps_env = getenv("NEEDLE_NG_STAMPS") && atoi(getenv("NEEDLE_NG_STAMPS")) != 0;
    if (stamps_env) {
        static uint64_t *d_stamps = nullptr;
        const size_t n_waves = (size_t)n_cus * kWavesPerBlock, bytes = n_waves * 8 * sizeof(uint64_t);
        if (!d_stamps && hipMalloc((void **)&d_stamps, 4096 * 8 * sizeof(uint64_t)) != hipSuccess) return hipErrorOutOfMemory;
        if (n_waves > 4096) return hipErrorInvalidValue;
        (void)hipMemsetAsync(d_stamps, 0, bytes, stream);
        A.stamps = d_stamps;
        hipError_t e = op == OP_NG_FIND_ALL ? launch_ng_m<OP_NG_FIND_ALL>(A, n_cus, lds, stream)
                       : op == OP_FIND      ? launch_ng_m<OP_FIND>(A, n_cus, lds, stream)
                                            : launch_ng_m<OP_CONTAINED_IN>(A, n_cus, lds, stream);
        if (e != hipSuccess) return e;
        std::vector<uint64_t> h(n_waves * 8);
        e = hipStreamSynchronize(stream);
        if (e == hipSuccess) e = hipMemcpy(h.data(), d_stamps, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return e;
        double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mx = 0;
        for (size_t w = 0; w < n_waves; ++w) {
            for (int k = 0; k < 8; ++k) sum[k] += (double)h[w * 8 + k];
            if ((double)h[w * 8 + 7] > mx) mx = (double)h[w * 8 + 7];
        }
        fprintf(stderr, "NG-STAMPS op %d mode %u S %u cw %d wide %u: waves %zu, mean lifetime %.0f cycles (max %.0f); shares: text-wait %.3f probe %.3f queue %.3f level2 %.3f walk %.3f group %.3f staging %.3f\n",
                op, a.hdr.mode, ng.stride, char_width, ng.wide, n_waves, sum[7] / n_waves, mx, sum[0] / sum[7], sum[1] / sum[7], sum[2] / sum[7],
                sum[3] / sum[7], sum[4] / sum[7], sum[5] / sum[7], sum[6] / sum[7]);
        { // how unevenly the waves finish: percentiles of their lifetimes, and the mean by XCD (block % 8) and by wave slot
            std::vector<double> life(n_waves);
            double xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0}, slot[kWavesPerBlock];
            for (int k = 0; k < kWavesPerBlock; ++k) slot[k] = 0;
            for (size_t w = 0; w < n_waves; ++w) life[w] = (double)h[w * 8 + 7], xcd[(w / kWavesPerBlock) % 8] += life[w], slot[w % kWavesPerBlock] += life[w];
            std::vector<double> srt = life;
            std::sort(srt.begin(), srt.end());
            fprintf(stderr, "NG-LIFE p01 %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f; by xcd:", srt[n_waves / 100], srt[n_waves / 10], srt[n_waves / 2], srt[n_waves * 9 / 10],
                    srt[n_waves * 99 / 100], srt[n_waves - 1]);
            for (int k = 0; k < 8; ++k) fprintf(stderr, " %.0f", xcd[k] / (n_waves / 8));
            fprintf(stderr, "; by wave slot:");
            for (int k = 0; k < kWavesPerBlock; ++k) fprintf(stderr, " %.0f", slot[k] / n_cus);
            fprintf(stderr, "\n");
            for (int q = 0; q < 4; ++q) { // the four waves of a SIMD, oldest first: mean cycles by section
                double sec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (size_t w = 0; w < n_waves; ++w)
                    if ((int)((w % kWavesPerBlock) / 4) == q)
                        for (int k = 0; k < 8; ++k) sec[k] += (double)h[w * 8 + k];
                fprintf(stderr, "NG-SLOT %d: text-wait %.0f probe %.0f queue %.0f level2 %.0f walk %.0f group %.0f total %.0f\n", q, sec[0] / (n_waves / 4), sec[1] / (n_waves / 4),
                        sec[2] / (n_waves / 4), sec[3] / (n_waves / 4), sec[4] / (n_waves / 4), sec[5] / (n_waves / 4), sec[7] / (n_waves / 4));
            }
        }
        return hipSuccess;
    }
#endif
    if (op == OP_NG_FIND_ALL) return launch_ng_m<OP_NG_FIND_ALL>(A, n_cus, lds, stream);
    return op == OP_FIND ? launch_ng_m<OP_FIND>(A, n_cus, lds, stream) : launch_ng_m<OP_CONTAINED_IN>(A, n_cus, lds, stream);
}

} // namespace needle
