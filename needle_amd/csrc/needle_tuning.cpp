// needle_tuning_info(): the one list of the environment switches libneedle_hip.so reads (each once per process, at first use).
// None of them changes an answer: they choose between kernels / layouts that are parity-tested against the same oracle, size
// host-side staging, or print diagnostics.  tests/test_abi.py checks this table against every getenv("NEEDLE_...") in the sources.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include "../../include/needle_hip.h"

namespace {
struct Switch {
    const char *name, *dflt, *scope, *effect;
};
// scope: "layout" (which lowering / kernel runs), "size" (staging, budgets), "debug" (prints), "measurement build" (compiled in only
// with -DNEEDLE_TUNING, scripts/build_tuning.sh: not in the shipping library)
const Switch kSwitches[] = {
    {"NEEDLE_PREFILTER", "1", "layout", "n-gram candidate filter in front of the automaton (needle_ngram.hip): 0 never, 1 for automata in the compressed form, 2 for every LDS-table automaton that allows one"},
    {"NEEDLE_PREFILTER_LEVEL2", "1", "layout", "0: no second-level window (a candidate's 5-byte window looked up in a second bitmap before the automaton runs on it)"},
    {"NEEDLE_PREFILTER_UTF16", "1", "layout", "0: UTF-16 rows never take a filter kernel (patterns on one page of the BMP: the byte program's, text narrowed as it is loaded; several pages: the wide filter)"},
    {"NEEDLE_PREFILTER_WIDE", "1", "layout", "0: UTF-16 rows of patterns on several pages of the BMP never take the wide filter (windows of four 16-bit code units, needle_ngram.h ngram_piece16); NEEDLE_PREFILTER_UTF16=0 switches it off too"},
    {"NEEDLE_PREFILTER_UNBOUNDED", "1", "layout", "0: find() of patterns without bounded match lengths never runs behind the n-gram filter (whose verified candidates find their starts by backward walks)"},
    {"NEEDLE_PREFILTER_STRIDE", "4", "layout", "largest window stride the n-gram filter may choose (2: never 4 -- keeps the second-level window for patterns whose shortest match is 7 chars)"},
    {"NEEDLE_PREFILTER_WATCH", "1", "layout", "0: the filter kernel is never suspended (the flood watch: after a launch that saw more than 16 candidates per KiB of text the program's next 32 .. 1024 calls take the ordinary scan kernel)"},
    {"NEEDLE_FIND_LENGTHS", "1", "layout", "find() by the lengths automaton (start = end - length, no backward walk): 0 never (forward + backward walks), 1 where the ordinary program is an LDS table, 2 also instead of a pair table"},
    {"NEEDLE_FIND_LENGTHS_SPARSE", "1", "layout", "0: compressed-form automata keep the two walks"},
    {"NEEDLE_FIND_LENGTHS_PAIR", "1", "layout", "0: pair-table automata keep the two walks"},
    {"NEEDLE_FIND_ALL_LENGTHS", "1", "layout", "find-all's starts: 0 by backward walks, 1 by the lengths automaton where it fits the LDS as a plain table, 2 also in its compressed form (big dictionaries; measured: no faster)"},
    {"NEEDLE_FIND_ALL_LOCKSTEP", "1", "layout", "0: find-all never takes the lock-step kernel (the find-all transducer, needle_find_all_ls.hip); patterns that have one keep the per-lane one-pass kernel"},
    {"NEEDLE_FIND_ALL_RUNS", "1", "layout", "0: find-all of run patterns (`[0-9]+`: no bounded match length) never takes the lock-step kernel with the run transducer; they keep the per-lane one-pass kernel and its backward walks"},
    {"NEEDLE_FIND_ALL_FILTER", "1", "layout", "0: find-all never runs behind the n-gram candidate filter (dictionaries whose find() does keep the one-pass find-all kernel)"},
    {"NEEDLE_FIND_ALL_WINDOW", "1", "layout", "0: the find-all kernel's lengths program keeps column-map lookups instead of window addressing"},
    {"NEEDLE_FIND_ALL_DEFER", "1", "layout", "0: find-all (two-walk form) finds each start as the match is found instead of deferring them to the row's end"},
    {"NEEDLE_FIND_ALL_ROUNDS", "0", "layout", "1: find-all as rounds of needle_find_next_dev (one pass over the batch per match rank) instead of the one-pass kernel"},
    {"NEEDLE_FIND_ALL_SHAPE", "(by LDS footprint)", "layout", "\"<waves>x<tile bytes>\" workgroup shape of the find-all kernel"},
    {"NEEDLE_SPARSE", "1", "layout", "0: automata larger than the LDS skip the compressed form (dense rows + exception records) and fall to hot rows + HBM table"},
    {"NEEDLE_SPARSE_ROOM", "98304", "size", "LDS bytes the compressed form may take"},
    {"NEEDLE_HYBRID", "1", "layout", "0: no hot-rows form either: plain HBM table"},
    {"NEEDLE_WINDOW", "1", "layout", "0: column-map lookups instead of window addressing (clamped char = column offset)"},
    {"NEEDLE_FLAT_MAP", "1", "layout", "0: UTF-16 rows of LDS-table automata always use the compact two-level page map instead of the flat 64 KB one (one column lookup per char)"},
    {"NEEDLE_BTABLE_LDS_MAX", "8192", "size", "largest dense backward table of find() that rides in the forward program's LDS image when 16 waves of tiles still fit beside it (2048: round 4's rule)"},
    {"NEEDLE_PAIR_MAX_BYTES", "98304", "size", "largest pair table ([state][col][col] uint16, two chars per lookup); 0: never"},
    {"NEEDLE_MAX_PROG_LDS", "(device limit)", "size", "LDS bytes an automaton may take (tests lower it to force the HBM-table mode)"},
    {"NEEDLE_SHAPE", "(by LDS footprint)", "layout", "\"<waves>x<tile bytes>\" workgroup shape of the tiled scan kernel"},
    {"NEEDLE_PACK_WAVES", "14", "layout", "fewest waves a packed-mode workgroup may have before 128-byte tiles are given up"},
    {"NEEDLE_SHORT_ROWS", "1", "layout", "0: rows of at most 64 bytes take the tiled kernel instead of the register-resident one"},
    {"NEEDLE_SHORT_WGS", "2", "layout", "workgroups per CU of the short-row kernel"},
    {"NEEDLE_DEFER", "16", "layout", "survivor pool: a 64-row group with at most this many unresolved rows hands them to the wave's pool (0: off; at most 32)"},
    {"NEEDLE_RESERVE_CUS", "0", "layout", "CUs left free by the persistent scan launch (for an RCCL gather running beside it)"},
    {"NEEDLE_LONG_ROWS", "-1", "layout", "few long rows: -1 by shape, 0 never take the stripe paths, 1 always"},
    {"NEEDLE_STRIPE_CAND", "1", "layout", "0: stripe-path find() re-walks every stripe instead of only the last accepting one"},
    {"NEEDLE_MULTI_PACK16", "1", "layout", "needle_multi_scan find(): 0 peers send int32 start / end, 1 one dword per row, 2 also for shards on the root's device (tests)"},
    {"NEEDLE_MULTI_NO_RCCL", "0", "layout", "1: needle_multi gathers by peer copies instead of RCCL"},
    {"NEEDLE_HOST_CHUNK_BYTES", "2147483648", "size", "device bytes one chunk of a host batch may take"},
    {"NEEDLE_HOST_RESULT_BYTES", "536870912", "size", "device bytes the results of one find-all host chunk may take"},
    {"NEEDLE_SCRATCH_KEEP_MB", "512", "size", "freed scratch memory the library's pool keeps for the next call (needle_trim_scratch hands back the rest)"},
    {"NEEDLE_SPARSE_DEBUG", "(unset)", "debug", "prints the compressed form's sizing to stderr"},
    {"NEEDLE_ML_DEBUG", "(unset)", "debug", "prints why a pattern has no lengths automaton"},
    {"NEEDLE_COMPILE_TIMING", "(unset)", "debug", "prints where needle_compile's time goes (subset construction, pruning, minimisation per automaton)"},
    {"NEEDLE_DEBUG_NFA", "(unset)", "debug", "dumps the forward Thompson program of needle_compile"},
    {"NEEDLE_DICT", "0", "measurement build", "two 64-row sets per wave for big automata (needle_dict.hip): 1 compressed form, 2 also uint16 tables"},
    {"NEEDLE_NG_DBG", "0", "measurement build", "n-gram filter kernel time breakdown (drops candidates / skips walks: timing only)"},
    {"NEEDLE_NG_STAMPS", "0", "measurement build", "1: every n-gram filter launch is synchronised and prints where its waves' shader cycles went (text wait / probes / queue / second level / verify walks / group ends)"},
    {"NEEDLE_DEBUG_NO_BACKWARD", "(unset)", "measurement build", "find() with start := end (the bound the lengths automaton was built to reach)"},
};
} // namespace

extern "C" int needle_tuning_info(char *buf, size_t cap, size_t *needed) {
    std::string out = "name\tdefault\tcurrent\tscope\teffect\n";
    for (const Switch &s : kSwitches) {
        const char *cur = getenv(s.name);
        out += s.name;
        out += '\t';
        out += s.dflt;
        out += '\t';
        out += cur ? cur : "";
        out += '\t';
        out += s.scope;
        out += '\t';
        out += s.effect;
        out += '\n';
    }
    if (needed) *needed = out.size() + 1;
    if (buf && cap) {
        const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return NEEDLE_OK;
}
