// Launch arguments of the one-pass find-all kernel (needle_find_all.hip); shared with the launcher's caller in
// needle_api.cpp.
#pragma once
#include <stddef.h>
#include "needle_device.h"

namespace needle {

// Every non-overlapping match of every row in one pass (needle_find_all.hip).
struct FindAllArgs {
    ScanArgs s;        // rows, lengths, forward program (+ backward maps), backward program, fixed_len; no outputs
    uint32_t slots;    // matches filed per row at most
    uint32_t defer;    // starts by indexBackwards: 1 = at the end of each 64-row group, the group's matches handed out one per
                       // lane (patterns that do not match the empty string); 0 = every start at the moment its match is
                       // found; 2 = 1 without the walks (measurement aid)
    uint32_t *counts;  // [n_rows]
    int32_t *starts;   // [n_rows][slots]
    int32_t *ends;     // [n_rows][slots]
    uint32_t *packed;  // != nullptr: [n_rows][slots] (or the compact filing) of start | end << 16 instead of starts / ends
                       // (rows of at most 65 535 chars): one store per match, half the result lines
    int32_t *more;     // set to 1 when some row has a match beyond its last slot
    const uint64_t *offsets; // != nullptr: compact (CSR) filing -- match k of row r at offsets[r] + k, room for
                             // offsets[r + 1] - offsets[r] matches; slots is not used
    uint32_t count_only;     // 1: nothing is filed (starts / ends may be null), every match is counted
    uint32_t kshift;         // 0: a row's slots are consecutive ([row][slot]); 6: GROUP-BLOCKED slots -- match k of row r at
                             // ((r >> 6) * slots + k) * 64 + (r & 63): slot k of a group's 64 rows is one 256-byte run, so the lanes'
                             // stores fill whole lines in L2 instead of 4 bytes of one line per row (needle_find_all_blocked16_dev)
    uint32_t lmode;          // 1: the program is the "lengths" automaton (needle_lower.h): start = end - pend[end state], read
                             // from LDS at hdr.fa_len_off; the search also ends in the states fa_dead_lo .. + fa_dead_n - 1
};
// backward_walk (needle_walk.h) reads the ScanArgs header words from the kernarg segment at their offsets INSIDE ScanArgs: it must be the first member
static_assert(offsetof(FindAllArgs, s) == 0, "ScanArgs must be the first member of FindAllArgs (kernarg_here, needle_walk.h)");


} // namespace needle
