// Host side of the n-gram candidate filter (device side: needle_ngram.h / needle_ngram.hip).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "needle_ngram.h"

namespace needle {

struct NgramFilter {
    NgramParams p;                // p.on == 0: no filter (why says why)
    std::vector<uint32_t> bitmap; // p.bm_bytes / 4 words; bit (i & 31) of word (i >> 5)
    std::vector<uint32_t> bitmap2; // p.on2: the second level's, p.bm2_bytes / 4 words
    std::string why;
};

// The automaton as the kernels walk it, BEFORE any mode-specific encoding: next[state * n_cols + column] in device numbering
// (0 = sink; states <= dead_hi end a search; states >= accept_lo accept), cmap8[byte] = column of an 8-bit code unit.
// `absorbing`: containedIn (the first accepting state ends the walk).  prog_lds_bytes: the LDS the program itself takes (the
// bitmap is sized so that ngram_layout() places it and the waves' queues behind it).  cmap16 != nullptr (65 536 entries: column of every
// UTF-16 code unit): the WIDE filter -- the same analysis with windows of four code units, for UTF-16 rows of patterns that live on more
// than one page of the BMP (DFA.java:438-463: the reference's class map covers all 65 536 units).
//
// What is established ON THE TABLE, not argued from the regex (any failure => no filter for this program):
//  * the start state does not accept, and the shortest accepted string has min_len >= 4 chars;
//  * K ("warm"): whatever non-accepting state s a walk is in, a second walk started in the start state at the same char is in
//    the SAME state after K chars, and never accepts before the first one does -- so a walk restarted K chars before a
//    position reports what the walk from the row's start reports there (as long as that one has not accepted yet);
//  * the windows: every 4-column sequence that labels the 4 transitions ending o chars ahead of a FIRST accepting transition
//    (o = 0 .. S - 1), expanded to bytes and hashed into the bitmap.  A first accept at char index i therefore has, for the
//    one o with (i - o) = 0 (mod S), a window [i - o - 4, i - o) in the bitmap: no accept without a candidate.
//  * the second level (when min_len >= 5 + S - 1): the same with 5-column sequences -- a first accept at least 5 + o chars into
//    its row also has [i - o - 5, i - o) in the second bitmap (candidates nearer the row's start are not asked).
//    NEEDLE_PREFILTER_LEVEL2=0: no second level (A/B, tests).
NgramFilter build_ngram_filter(const uint16_t *next, int n_dev, int n_cols, const uint8_t *cmap8, int start, int accept_lo, int dead_hi,
                               bool absorbing, size_t prog_lds_bytes, const uint8_t *cmap16 = nullptr);

} // namespace needle
