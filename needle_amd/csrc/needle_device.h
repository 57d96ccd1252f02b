// Shared between the host lowering (needle_lower.cpp), the launcher and the kernels (needle_kernels.hip).
#pragma once
#include <stdint.h>

namespace needle {

// Which generated loop of the reference a launch stands for.
enum Op : int { OP_MATCHES = 0, OP_CONTAINED_IN = 1, OP_FIND = 2 };

// How a lowered automaton is walked on the device.
enum Mode : int {
    MODE_PACK = 0,    // <= 6 states: per-char transition FUNCTION packed 5 bits/state in one dword; the state IS the
                      // bit offset of its field, so a transition is one v_bfe_u32 and no dependent LDS lookup
    MODE_TABLE8 = 1,  // [state][column] uint8 next-state table in LDS
    MODE_TABLE16 = 2, // [state][column] uint16 next-state table in LDS
    MODE_GLOBAL = 3,  // uint16 table too large for LDS: walked out of HBM/L2
    MODE_HYBRID = 5,  // uint16 table too large for LDS, but automata live in few states: the rows of the first `hot`
                      // states in breadth-first order from the start state are in LDS, the whole table in HBM; a lane
                      // in a colder state fetches its entry through the scalar cache.  Entries carry the "accepting" bit
                      // (bit 15) so that the numbering is free to follow the breadth-first order.
    MODE_SPARSE = 6,  // table too large for LDS as a dense [state][column] array, but compressible: the states near the start
                      // state (breadth-first) keep DENSE rows of uint32 cells, every other state is a "default row + exception"
                      // chain of 8-byte records -- and the WHOLE automaton is LDS-resident.  A state is the pair
                      //   (rowA4 = LDS address / 4 of the dense row to read, recB = LDS address of its first record)
                      // in one dword (recB << 16 | rowA4): a transition is ONE LDS round trip -- the dense cell and the
                      // record are read in parallel and the record wins when its key is the char's column.
    MODE_PAIR = 4,    // 8-bit rows, <= 256 states, n_states * n_cols^2 entries fit the LDS: TWO chars per dependent
                      // lookup -- uint16 [state][col1][col2] = next state after both | code << 8 (find: 0 no accept,
                      // 1 accepted after the first char only, 2 accepted after the second)
};

// Device-side numbering of a lowered automaton (independent of the reference's state numbers):
//   0            sink (dead, absorbing, non-accepting)
//   1 .. A0-1    non-accepting states
//   A0 .. n-1    accepting states          => accepted(s) == (s >= A0)
// Column layout of a row: reference classes 0..N-1, then OVER (char > maxChar), PAD (index >= row length) and
// PRE (index < the row's find() cursor: identity, the walk has not started yet).
struct ProgHeader {
    uint32_t mode;       // Mode
    uint32_t n_states;   // incl. sink
    uint32_t n_cols;     // N + 3
    uint32_t start;      // device id of the reference's state 0
    uint32_t accept_lo;  // A0
    uint32_t root_accepting;
    uint32_t lds_bytes;  // bytes of the blob that the kernel stages in LDS (0 for MODE_GLOBAL table part)
    // byte offsets inside the blob (all 16-byte aligned); 0xFFFFFFFF = absent
    uint32_t off_f;      // MODE_NIBBLE: uint32 F[] -- char_width 1: F[byte][32], one copy per LDS bank so that lane l
                         //              always reads bank l & 31 (no bank conflicts whatever the text);
                         //              char_width 2: n_cols entries indexed by column
    uint32_t pad_f;      // packed mode: F of the PAD column (selected in registers for chars past the row length)
    uint32_t pre_f;      // packed mode: F of the PRE column (identity)
    uint32_t off_cmap;   // char_width 1 table modes: uint8 column[256]
    uint32_t off_ptab;   // char_width 2: uint8 page_of[256] (high byte -> page)
    uint32_t off_pages;  // char_width 2: uint8 column[n_pages][256]
    uint32_t off_table;  // table modes: next-state table [n_states][n_cols]
    uint32_t n_pages;
    uint32_t pad_col;    // column index of PAD (= n_cols - 2); PRE = n_cols - 1
    // OP_FIND forward programs also carry the BACKWARD automaton's char -> column maps (staged in LDS with the
    // rest; the backward table itself is walked out of HBM/L2)
    uint32_t off_bcmap, off_bptab, off_bpages;
    uint32_t off_btable; // != 0: the backward uint16 table is small and staged in LDS too (else read bprog from HBM/L2)
    // MODE_PACK: bit offset of every device state's field (pack_off[0] = 0: the sink), the offsets of the start state
    // and of the first accepting state (32 = none), the identity function; the same for a packed backward automaton
    uint8_t pack_off[8];
    uint32_t start_off, accept_off, ident_fn;
    uint32_t bpack_start_off, bpack_accept_off;
    uint32_t hot_bytes;  // MODE_HYBRID: bytes of the table prefix (whole rows) that is in LDS at off_table
    uint32_t off_gtable; // MODE_HYBRID: the whole table inside the blob in HBM (after lds_bytes)
    uint32_t off_bsp_bm; // != 0: a backward automaton too big for a dense table in LDS but SPARSE (a reversed keyword trie:
                         // 1704 live transitions in 720 x 31 cells) rides along popcount-compressed: uint32 bitmap[state]
                         // of the columns with a live target here, uint16 base[state] at off_bsp_base, and the live targets
                         // back to back at off_bsp_edges: next = bit(col) ? edges[base + popcount(bitmap below col)] : sink
    uint32_t off_bsp_base, off_bsp_edges;
    // MODE_SPARSE (all addresses relative to the table base: kLdsTable1 for 8-bit rows, off_table for UTF-16 ones):
    //   [0, (N+1)*4)        the sink's row (all cells 0; the sink is state value 0)
    //   [sp_rec_base, ..)   records {uint16 key = column * 4, uint16 next_record (0 = none), uint32 target state}: the
    //                       never-matching dummy of the non-accepting dense states first, then the chains of the non-accepting
    //                       sparse states, then -- from sp_accept_rec on -- the accepting dummy and the accepting states'
    //                       chains, so that accepted(s) == (s >= accept_lo) with accept_lo = sp_accept_rec << 16
    //   [sp_rows_base, ..)  dense rows: uint32 cells [N + 1 columns] = the target's state value
    // PAD / PRE are not columns in this mode (identity differs for every state: each would cost every sparse state an
    // exception): the guarded kernels select them after the lookup.
    uint32_t sp_rec_base, sp_accept_rec, sp_rows_base;
    uint32_t sp_chains;    // some state needs more than one record: the walk loops while a lane's record has a successor
    uint32_t sp_pad_ident; // PAD (chars past the row's length) is the identity (matches / containedIn) or leads to the sink
    uint32_t sp_dense, sp_records; // statistics: dense rows (without the sink), records (without the two dummies)
    // Window addressing (table modes, needle_lower.cpp): when the char -> column map is constant below some char and constant
    // above another one, the table's columns are the CHARS of the window in between (the chars cl - 1 and ch + 1 standing
    // for everything below / above) and the per-char column-map lookup -- one LDS read per char -- becomes one clamp:
    // column offset = min(max(char * element size, win_lo_e), win_hi_e).  The offset is NOT rebased to 0: the table sits
    // win_lo_e bytes further up instead.  n_cols / pad_col describe the window layout then.
    uint32_t win_on, win_lo_e, win_hi_e;
    // find-all "lengths" form (needle_lower.h, MatchLengths): pend[] by device state in LDS; the dead-with-a-pending-match
    // states are the device ids fa_dead_lo .. fa_dead_lo + fa_dead_n - 1
    uint32_t fa_len_off, fa_dead_lo, fa_dead_n;
    // the scan kernels' test for "this lane's search is over": state value <= fa_dead_hi (plain tables: = fa_dead_n; the compressed
    // form: the value of D_K, whose rows are the first dense rows).  Compressed lengths programs: sp_end_col4 = key of the END
    // records (row end -> the D_L of the pending length; the states with a match pending carry one, last in their chain),
    // sp_dead_row0 = address field of D_1's row, sp_len_cols = cells per row
    uint32_t fa_dead_hi, sp_end_col4, sp_dead_row0, sp_len_cols;
    // find-all programs of that form also carry "skip" states: S_k (device id fa_skip_lo + k - 1) goes to S_(k-1) on EVERY
    // column and S_1 to the start state -- a search restarted k chars into a 16-byte piece enters the piece in S_k and needs
    // no per-char cursor guard (0: none)
    uint32_t fa_skip_lo;
    uint32_t flat_pages; // char_width 2 table modes: 1 = every high byte has a page of its own and ptab[hi] = hi * 256 (the map is
                         // then 64 KB and a char's column is ONE lookup at the char itself; the two-level lookup still gives
                         // the same answer, so kernels that do not know the flag stay correct)
    // find-all transducer programs (needle_lower.h lower_find_all_transducer; walked by needle_find_all_ls.hip): table entries are
    // state << 4 | code, code != 0 = "this transition ends a match"; codes[code] = (k + length) | k << 16 (uint32, LDS at ft_codes_off):
    // end = index of the char that took the transition - k, start = end - length.  State 0 = dead (row finished).  ft_odd: at most 8
    // codes, numbered 1, 3, .. 15 (bit 0 of an entry = "a match ends here").
    // ft_direct: every code has k = 0 and names its own length -- no table lookup when a match is filed: 1 = the code is the length (1 .. 15),
    // 2 = the code is length << 1 | 1 (lengths 1 .. 7; ft_odd holds too).
    uint32_t ft_on, ft_codes_off, ft_odd, ft_direct;
    uint32_t off_bpack;  // != 0: the backward automaton has <= 5 states and rides along as packed functions: 8-bit rows
                         // u32 F[256] there; UTF-16 rows ptab64[256] there ({absolute F address, mask} per high byte)
                         // followed by its F area.  The backward walk then needs no state-dependent lookup.
};

struct ScanArgs {
    const uint8_t *rows;
    uint64_t n_rows;
    uint64_t stride_bytes;
    uint64_t total_bytes;   // n_rows * stride_bytes (clamp for tail reads)
    uint32_t row_len;       // chars, when lengths == nullptr
    const uint32_t *lengths;
    const int32_t *from;    // OP_FIND: optional per-row cursor (Matcher.nextStart): walk starts there; < 0 = row exhausted
    const uint8_t *prog;    // forward program blob (device)
    ProgHeader hdr;
    const uint8_t *bprog;   // OP_FIND: backward program blob (device, walked out of global memory), or nullptr
    ProgHeader bhdr;
    int32_t fixed_len;      // OP_FIND: >= 0 => start = end - fixed_len
    uint32_t tiles_in_f_rows; // packed mode, 8-bit rows: waves 0..3 keep their tiles inside the F rows' upper halves
    uint64_t *bitmap;
    int32_t *start;
    int32_t *end;
    uint32_t *packed;       // OP_FIND, rows of at most 65 534 chars: when set, a row's result is stored as ONE dword here --
                            // start | end << 16, 0xFFFFFFFF = no match (needle_find_packed16_dev) -- and start / end are not written
    uint32_t packed8;       // != 0: `packed` points at uint16 results (rows of at most 256 chars, needle_find_packed8_dev): pack8() below
    uint32_t *end_state;    // optional: the automaton state (device id) in which every row's walk stopped
    uint32_t short_window;   // short_kernel (rows <= 64 B), find(): LDS holds an 80-byte slot per lane behind the program -- the
                             // matched rows' text goes there for the backward walk instead of being re-read from memory
    uint32_t defer_max_live; // survivor pool (needle_kernels.hip): a group with at most this many unresolved rows after
                             // a 128-byte line hands them to its wave's pool and ends; 0 = off
};

// Long rows of table-mode automata (needle_stripe.hip, "speculative stripes"): every stripe is first scanned as a row of
// its own from the START state (the tiled kernel, all stripes in parallel); then each stripe whose true entry state
// differs is re-walked next to the speculative run until the two states meet.
struct SpecArgs {
    const uint8_t *rows;
    uint64_t n_rows;
    uint64_t stride_bytes;   // of a row
    uint32_t stripe_bytes;   // divides stride_bytes
    uint32_t spr;            // stripes per row
    uint32_t char_width;
    uint32_t op;             // OP_CONTAINED_IN | OP_FIND
    uint32_t row_len;
    const uint32_t *lengths; // per row, or nullptr
    const uint8_t *gprog;    // forward automaton in the HBM-table layout: column maps at the front, uint16 table at hdr.off_table
    ProgHeader hdr;
    // per stripe (n_rows * spr)
    uint32_t *slen;          // chars of the stripe inside its row
    const uint32_t *spec_end_state;
    const int32_t *spec_last;    // find: lastMatch inside the stripe from the speculative run (-1 none)
    const uint64_t *spec_bitmap; // containedIn: the speculative run accepted
    uint32_t *entry;         // current guess of the true entry state
    uint32_t *entry_done;    // entry state the stripe's results were last computed for (0xFFFFFFFF: never)
    uint32_t *true_end_state;
    int32_t *true_last;      // find: lastMatch inside the stripe (-1 none); containedIn: 1 accepted / -1 not
    int32_t *changed;        // set when some stripe's successor got a new entry state
    // per row
    uint64_t *bitmap;
    int32_t *end;
};

// Long rows: a row is cut into 4 KiB stripes (one wave-step each, 64 contiguous bytes per lane) and
// the automaton's transition FUNCTION of every stripe is computed for all entry states at once (packed mode only).
constexpr uint32_t kStripeBytes = 4096;
struct StripeArgs {
    const uint8_t *rows;
    uint64_t n_rows;
    uint64_t stride_bytes;
    uint32_t row_len;        // chars, when lengths == nullptr
    const uint32_t *lengths;
    const uint8_t *prog;     // packed-mode program blob
    ProgHeader hdr;
    uint32_t spr;            // stripes per row = ceil(stride_bytes / kStripeBytes)
    uint32_t *fn;            // [n_rows][spr]: pass 1 writes each stripe's function, the prefix pass replaces it by the
                             // stripe's ENTRY state (5 * device state id)
    uint64_t *bitmap;        // prefix pass: verdicts (matches / containedIn), or "matched" for find (set by the last pass)
    uint32_t *cand;          // find (nullptr: off): [n_rows][spr], pass 1: bit k set <=> entered in its k-th tracked state the stripe
                             // passes through an accepting state
    int32_t *cand_stripe;    // find: [n_rows], prefix pass: the LAST stripe whose true entry state has that bit set (-1: none) --
                             // the only stripe of the row pass 2 walks again
    int32_t *end;            // find: lastMatch per row (pass 2, atomicMax); initialised by the prefix pass
    int32_t *start;          // find: written by the backward pass
    const uint8_t *bprog;    // find: backward program (global-walk layout) and header
    ProgHeader bhdr;
    int32_t fixed_len;
    uint32_t op;
};

// Fixed LDS byte offsets of the forward automaton (compile-time so that they fold into ds_read immediates).
//   char_width 1:  packed: F[256][64] u32 at 0 (one copy per lane) table modes: cmap16[256] at 0, table at 512
//   char_width 2:  packed: ptab64[256] at 0 = {base, mask} per high byte, F area at 2048: the F of char (hi, lo) is the
//                          u32 at  2048 + (base | (lo * 4 & mask)).  A page whose 256 entries differ is stored whole
//                          (base = k * 1024, mask = 0x3FC); a page on which F is CONSTANT is one shared dword (mask = 0):
//                          every lane reading such a page hits the same address and the LDS broadcasts it -- no bank
//                          conflicts however mixed the text (plain ASCII under a non-Latin class regex, CJK, ...).
//                          Two dependent lookups per char, not three.  (Tried instead: a flat 64 KiB code unit -> column
//                          map, one VALU op less per char -- but it only leaves room for 64-byte tiles: 0.96 -> 1.00 ms.)
//                          The packed BACKWARD automaton of find() rides along in the same form with absolute addresses.
//                  table modes: ptab16[256] at 0 (page * 256), pages8 (col * elem) at 512, table at hdr.off_table
// find() results of rows of at most 256 chars as ONE uint16 (needle_find_packed8_dev): start | (end - start) << 8; the pairs with
// start + length > 256 cannot occur and serve as escapes -- 0xFFFF = no match, 0xFFFE = the match (0, 256), the only one of length 256.
__host__ __device__ inline uint16_t pack8(int32_t s, int32_t e) {
    if (e < 0) return 0xFFFFu;
    const uint32_t len = (uint32_t)(e - s);
    return len > 255u ? (uint16_t)0xFFFEu : (uint16_t)(((uint32_t)s & 0xFFu) | len << 8);
}
constexpr uint32_t kLdsF1 = 0, kLdsCmap1 = 0, kLdsTable1 = 512;
//                  pair mode: cmapA16[256] at 0 (col * n_cols * 2: first char of a pair), cmapB16[256] at 512 (col * 2)
constexpr uint32_t kLdsCmapB1 = 512, kLdsPairTable1 = 1024;
constexpr uint32_t kLdsPtab2 = 0;
constexpr uint32_t kLdsPagesF2 = 2048, kLdsPages2Table = 512;
// packed mode on UTF-16 rows needs ptab64 + one KiB per distinct non-constant page in LDS next to the tiles
constexpr uint32_t kMaxPackPagesBytes = 96u * 1024u;

constexpr int kWavesPerBlock = 16;     // 1024 threads: one workgroup per CU shares one LDS copy of the tables
// Largest automaton footprint that still leaves room for the smallest workgroup shape (4 waves x 64 rows x 64 B).
constexpr uint32_t kMaxProgLdsBytes = 160u * 1024u - 4u * 64u * 64u;

} // namespace needle
