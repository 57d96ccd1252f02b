// Host-side lowering: the reference's per-regex tables -> device "programs" for the HIP kernels.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "needle_device.h"
#include "needle_ngram_host.h"

namespace needle {

enum Which : int { W_MATCHES = 0, W_CONTAINED_IN = 1, W_FORWARDS = 2, W_BACKWARDS = 3 };

// One automaton exactly as the reference's generated class holds it
// (STATES_X flat [state * N + class], -1 = no transition; DFAClassBuilder.java:317-333).
struct RefDfa {
    int32_t n_states = 0;
    int32_t max_char = 0xFFFF;      // spec.dfa.maxChar(), DFA.java:384-398
    std::vector<int16_t> table;     // n_states * stride
    std::vector<uint8_t> accepting; // n_states
};

struct RefTables {
    std::vector<uint8_t> class_map; // 65536 = BYTE_CLASSES[0..65535], DFAClassBuilder.java:269-305
    int32_t stride = 0;             // N
    RefDfa dfa[4];
    int32_t fixed_len = -1;         // Factorization.canOnlyHaveOneLength() ? minLength : -1
    int32_t min_len = -1, max_len = -1;
};

struct Program {
    ProgHeader hdr;
    std::vector<uint8_t> blob;
    NgramFilter ng; // ng.p.on: the program may run behind the n-gram candidate filter (needle_ngram_host.h)
};

// ByteClassUtil.fillMultipleByteClassesFromString*_singleArray (needle-types/.../ByteClassUtil.java:50-120)
// over an array pre-filled with -1.  Returns false + message on malformed input.
bool decode_table_string(const char *s, int32_t n_states, int32_t stride, std::vector<int16_t> &out, std::string &err);

// Validates shapes / value ranges of tables coming across the C ABI.
bool validate_tables(const RefTables &t, std::string &err);

// Lower one automaton for `char_width`-byte haystacks.  `lds_table_budget`: bytes of LDS the automaton may
// take (tables above it are walked out of HBM: MODE_GLOBAL).  `global_walk`: build the plain uint16 layout
// read from global memory (used for the backward automaton of find()).
// `with_backward_maps` (W_FORWARDS only): append the backward automaton's char -> column maps to the blob.
// `no_pair`: never the two-chars-per-lookup table (the find-all kernel walks from per-lane char positions).
struct MatchLengths;
Program lower(const RefTables &t, Which which, int char_width, size_t lds_table_budget, bool global_walk,
              bool with_backward_maps = false, bool no_pair = false, const MatchLengths *ml = nullptr);

// ---- find-all without backward walks (needle_find_all.hip, "lengths" form) -----------------------------------------
// The reference finds a match's start by walking the reversed automaton back from its end (indexBackwards,
// DFAClassBuilder.java:529-586) unless the WHOLE pattern has one length (Factorization.canOnlyHaveOneLength, :640-646).
// Generalised here per STATE: the forward search automaton is refined (a product with "what the reversed automaton would
// report from here", kept for every one of its states; Moore-minimised with the match length as output) until every
// accepting state stands for ONE match length, and then made to REMEMBER the length of its last match until it dies -- dead states "D_L" instead of the
// sink.  A find() that ends in state s then reports start = end - pend[s]: no backward walk, no second look at the text.
// Possible for keyword unions and other patterns whose accepting runs have bounded, state-determined lengths; anything
// else (an accepting state reachable with two match lengths and no finite refinement, e.g. `[0-9]+`) keeps indexBackwards.
struct MatchLengths {
    bool ok = false;
    RefDfa dfa;                // refined forward search automaton; ref state 0 = start, states 1 .. n_dead = D_L (absorbing)
    std::vector<uint8_t> pend; // per ref state: the match length a find() ending there reports (0: no match pending)
    std::vector<int16_t> over; // per ref state: the target on a char beyond dfa.max_char (-1 = dead, nothing pending)
    int n_dead = 0;
};
MatchLengths match_length_automaton(const RefTables &t);
// The device program of that automaton (plain table modes only; mode == MODE_GLOBAL with an empty blob when it does not fit
// the LDS as one): hdr.fa_len_off = LDS offset of pend[] by DEVICE state, hdr.fa_dead_lo (= 1) / fa_dead_n = device ids of the
// D_L states -- "the search is over" is state <= fa_dead_n.  plain: for the find-all kernel (no window addressing).
Program lower_match_lengths(const RefTables &t, const MatchLengths &ml, int char_width, size_t lds_table_budget, bool plain = true);


// The filter program of an automaton too big for the LDS in any form (needle_lower.cpp): HBM-table layout + the n-gram filter.
Program lower_filter_hbm(const RefTables &t, Which which, const MatchLengths *ml, bool with_backward_maps = false);
Program lower_filter_wide(const RefTables &t, Which which, const MatchLengths *ml); // UTF-16 rows, windows of four code units (needle_ngram.h)

// ---- find-all in LOCK-STEP (needle_find_all_ls.hip): the find-all transducer -------------------------------------------
// The reference's repeated find() (DFAClassBuilder.java:616-659) restarts the search automaton AT the end of every match -- on chars
// the walk has already consumed while it waited for the automaton to die.  The one-pass kernel (needle_find_all.hip) does that
// restart per lane, so its lanes leave lock-step.  Here the restart is folded into the AUTOMATON: a state of the transducer is
// either a state m of the lengths automaton above with no match pending, or a triple (m, s, k) -- m has a match pending, whose last
// accepting char lies k chars back, and s is the "shadow": where a search restarted at that match's end stands by now.  When m dies
// the transition EMITS the match (code = which (length, k): end = index of the killing char - k, start = end - length) and leads
// to the shadow's successor, which IS the reference's restarted search, already past the chars in between.  One table lookup per
// char, every lane at the same char.  Exact when a shadow never accepts while its main walk still lives with a match pending
// (checked transition by transition while the product is built: `international|inter|nation` is refused -- "nation" accepts
// inside a live "international") and the codes fit 4 bits, the states 12: table entry = state << 4 | code.
// Column layout of the rows: the reference classes, OVER, PAD (the row's end: emits what is pending; leads to the dead state 0).
// empty blob: not available (the one-pass kernel stays).  hdr: mode MODE_TABLE16, ft_on = 1, ft_codes_off = LDS offset of
// uint32 codes[16] = (k + length) | k << 16 (ft_odd: at most 8 codes, numbered 1, 3, .. 15), start = the start state, pad_col, window addressing as for the scan kernels' tables.
Program lower_find_all_transducer(const RefTables &t, const MatchLengths &ml, int char_width, size_t lds_table_budget);
// The RUN transducer: lock-step find-all for patterns WITHOUT bounded match lengths whose matches are "runs" -- `[0-9]+`,
// `[a-z]{3}[a-z]*` (BASELINE's C2 / C5 patterns).  Established on the tables: (1) after an accept every live successor state accepts,
// so the search dies on the char right behind its match and restarts ON that char; (2) from the char that takes the search automaton
// out of its start state until it is back there or dead, the anchored automaton started on that char lives exactly as long and accepts
// exactly where the search does -- a match's start (indexBackwards, DFAClassBuilder.java:529-586) is then the run's first char.  The table
// is the search automaton with its dying transitions redirected through the start state; entry = state << 4 | code, code bit 0: a match
// ends in front of this char, bit 1: this char may begin a run.  The kernel logs both bits per char and keeps the last run start per lane
// (needle_find_all_ls.hip).  hdr.ft_on = 2.  empty blob: not such a pattern.
Program lower_find_all_runs(const RefTables &t, int char_width, size_t lds_table_budget);

} // namespace needle
