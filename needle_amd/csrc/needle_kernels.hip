// Launcher of the tiled scan kernel (needle_scan.h): workgroup shape from the automaton's LDS footprint, the choice
// between the tiled and the register-resident short-row kernel, and the dispatch to the per-loop translation units
// (needle_scan_matches.hip / _contained.hip / _find1.hip / _find2.hip: the kernel template's instantiations compile
// in parallel there).
#include "needle_walk.h"

namespace needle {

struct LaunchShape {
    int grid, waves, chb;
    size_t lds;
};
hipError_t launch_scan_matches(const ScanArgs &a, int cw, bool guard, LaunchShape sh, hipStream_t s);
hipError_t launch_scan_contained_in(const ScanArgs &a, int cw, bool guard, LaunchShape sh, hipStream_t s);
hipError_t launch_scan_find(const ScanArgs &a, int cw, bool guard, LaunchShape sh, hipStream_t s);

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
    if (h.mode == MODE_PACK && char_width == 1) {
        // (find() programs carry the packed backward automaton behind F, ~1.3 KB: one wave fewer keeps the 128-byte tiles --
        // whole lines, `nt` loads -- where the generic ladder below would fall to 64-byte tiles.  NEEDLE_PACK_WAVES=16: old rule)
        static const int min_waves = getenv("NEEDLE_PACK_WAVES") ? atoi(getenv("NEEDLE_PACK_WAVES")) : 14;
        for (int w = 16; w >= min_waves && w >= 5; --w)
            if (p + (size_t)(w - 4) * 8192u <= cap) {
                *waves = w;
                *chb = 128;
                *tiles_in_f_rows = 1;
                return true;
            }
    }
    static const int cand[8][2] = {{16, 128}, {12, 128}, {16, 64}, {14, 64}, {12, 64}, {10, 64}, {8, 64}, {4, 64}};
    for (const auto &c : cand) {
        if (p + (size_t)c[0] * 64 * c[1] <= cap) {
            *waves = c[0];
            *chb = c[1];
            return true;
        }
    }
    return false;
}

hipError_t launch_short_rows(int op, int char_width, const ScanArgs &a, int n_cus, hipStream_t stream); // needle_stripe.hip

hipError_t launch_scan(int op, int char_width, const ScanArgs &a_in, int n_cus, hipStream_t stream) {
    if (a_in.n_rows == 0) return hipSuccess;
    {
        static const int short_rows = getenv("NEEDLE_SHORT_ROWS") ? atoi(getenv("NEEDLE_SHORT_ROWS")) : 1; // 0: tuning / tests
        // (hot-rows and compressed automata on short rows take the tiled kernel: the register-resident one has no such modes)
        if (short_rows && a_in.stride_bytes <= 64 && a_in.hdr.mode != MODE_HYBRID && a_in.hdr.mode != MODE_SPARSE)
            return launch_short_rows(op, char_width, a_in, n_cus, stream);
    }
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
    // survivor pool (see the file header): on unless per-row cursors / end states are wanted (find_next, the speculative
    // stripe pass) or row indices do not fit the pool's 32-bit slots.  NEEDLE_DEFER=<n>: threshold (0 = off; tuning)
    static const int defer_env = getenv("NEEDLE_DEFER") ? atoi(getenv("NEEDLE_DEFER")) : 16;
    a.defer_max_live = 0;
    if (defer_env > 0 && defer_env <= 32 && !a.from && !a.end_state && a.n_rows < (1ull << 32)) a.defer_max_live = (uint32_t)defer_env;
    // unguarded kernels assume every row fills a whole number of tiles
    const bool guard = a.lengths != nullptr || a.from != nullptr || a.row_len == 0 ||
                       ((uint64_t)a.row_len * char_width) % sh.chb != 0;
    switch (op) {
    case OP_MATCHES: return launch_scan_matches(a, char_width, guard, sh, stream);
    case OP_CONTAINED_IN: return launch_scan_contained_in(a, char_width, guard, sh, stream);
    default: return launch_scan_find(a, char_width, guard, sh, stream);
    }
}

} // namespace needle
