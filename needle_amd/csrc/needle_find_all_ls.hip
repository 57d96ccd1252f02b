// needle_find_all_ls.hip -- every non-overlapping match of every row in ONE pass, in LOCK-STEP (the reference's
// repeated Matcher.find(), DFAClassBuilder.java:616-659; DFACompilerTest.java:66-78,671-699).  Hand-written for gfx950.
//
// The one-pass kernel of needle_find_all.hip restarts the search per lane where the reference does -- AT the end of each match, on
// chars the walk has already consumed -- so its lanes stop being at the same char: per-lane piece addresses (two thirds of its LDS
// cycles are bank conflicts), a tile costs its busiest lane's iterations, every match a re-walked piece (18.9 VALU + 2.3 LDS
// instructions per char-wave on the 1000-keyword dictionary).  Here the restart lives in the AUTOMATON (needle_lower.h: the find-all
// transducer -- a state is a state of the lengths automaton, or one with a match pending plus the "shadow" state a search restarted
// at that match's end has reached meanwhile; when the pending walk dies, the transition emits the match and leads to the shadow's
// successor).  The walk is the scan kernel's: one row per lane, every lane at the same char, whole lines HBM -> VGPRs -> XOR-swizzled
// LDS tile, ONE dependent table lookup per char:
//     entry = T[(entry >> 4) * row_bytes + column(char)]        (v_lshrrev, v_mad_u32_u24, ds_read_u16)
//     log   = {entry, log} >> 4                                 (v_alignbit: the entry's low 4 bits = the match code, 0 = none)
// After 8 chars the log holds their 8 codes; a non-zero nibble at char j with codes[code] = (length, k) is the match
// [end - length, end), end = index(j) - k (k chars lie between a match's last char and the char that killed its walk).  Matches are
// filed from the log, a lane at a time where there are any -- the walk itself never branches on them.  The row's end is one more
// transition (the PAD column: emits what is pending).  Results as in needle_find_all.hip: dense per-row slots (two arrays or one
// dword per match), compact filing at caller-computed offsets, or counting only.
#include "needle_walk.h"
#include "needle_find_all.h"

namespace needle {

// RUNS: the run transducer (needle_lower.h lower_find_all_runs: patterns without bounded match lengths whose matches are runs -- `[0-9]+`,
// BASELINE's C2 / C5): a code's bit 0 says "a match ends in front of this char", bit 1 "this char may begin a run"; the lane keeps the
// index of the last char that could (run_start) -- brought up to date from the log's bit-1 nibbles every 8 chars (four VALU ops) -- and a
// match ending at char i starts at the last such char below i.  No lengths, no backward walk (DFAClassBuilder.java:529-586 finds the
// same start: the lowering proved it on the tables).
template <int CW, bool GUARD, int CHB, bool RUNS = false>
__global__ __launch_bounds__(kWavesPerBlock * 64) void find_all_lockstep_kernel(const FindAllArgs fa) {
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
    wk.ncols_e = a.hdr.n_cols * 2u;
    wk.pad_e = a.hdr.pad_col * 2u;
    wk.pre_e = wk.pad_e; // (no cursors here)
    wk.pad_b = wk.pre_b = 0;
    wk.win_on = a.hdr.win_on, wk.win_lo = a.hdr.win_lo_e, wk.win_hi = a.hdr.win_hi_e;
    wk.table_off = a.hdr.off_table - a.hdr.win_lo_e; // (window addressing: column offsets are not rebased, needle_device.h)
    wk.sp_chains = 0, wk.sp_pad_ident = 0, wk.dead_hi = 0, wk.lane4 = 0, wk.gtable = nullptr, wk.hot_last = 0;
    wk.flat = 0;
    const uint32_t tbase = CW == 1 ? (uint32_t)kLdsTable1 : 0u;           // (UTF-16: piece_lookups has added the table's offset to the columns)
    const uint32_t pad_addr = wk.pad_e + (CW == 1 ? (uint32_t)kLdsTable1 : wk.table_off);
    const uint32_t codes_off = a.hdr.ft_codes_off;
    const uint32_t e_start = a.hdr.start << 4;
    const uint32_t kshift4 = fa.offsets ? 2u : fa.kshift + 2u; // bytes between a row's consecutive matches: 4, or 256 in the group-blocked layout

    Tile tile;
    {
        const uint32_t base = ((a.hdr.lds_bytes + 15u) & ~15u) + (uint32_t)wave * G::kTileBytes;
        tile.store_addr = base + (uint32_t)(lane / G::kPieces) * CHB + (uint32_t)(lane % G::kPieces) * 16u;
        tile.store_step = G::kRowsPerInstr * CHB;
        tile.row_addr = base + (uint32_t)lane * CHB;
    }

    const uint64_t n_groups = (a.n_rows + 63) >> 6;
    const uint64_t wave_cnt = (uint64_t)gridDim.x * n_waves;
    uint64_t g = (uint64_t)blockIdx.x * n_waves + wave;
    if (g >= n_groups) return;

    // tile fetch: as in scan_kernel (needle_scan.h) -- a fetch unit is one 128-byte line per row: one tile of 128-byte pieces or the
    // two 64-byte tiles of the same lines, requested back to back
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
    auto stage = [&](auto tc) __attribute__((always_inline)) {
        constexpr int T = decltype(tc)::value;
#pragma unroll
        for (int j = 0; j < G::kInstrs; ++j) store_piece(tile, j, R[T][j]);
    };

    // ---- per-group (per-row) state
    uint64_t my_row = 0;
    bool row_ok = false;
    uint32_t len = 0, n_chunks = 1, e = 0;
    uint32_t run_start = 0; // RUNS: index of the last char that may have begun a run
    uint32_t count = 0; // matches of this row so far -- ALL of them; the first `cap` are filed
    uint32_t cap = 0;   // matches this row may file
    // Result addressing: a wave-uniform 64-bit base per group (SGPRs: the group's first slot) + a 32-bit byte offset per lane --
    // the stores take the saddr form, one VALU op per match for the address (launch_find_all_lockstep keeps the shapes whose
    // per-group offsets could outgrow 32 bits on the one-pass kernel)
    uint64_t gbase = 0; // index of the group's first result slot (dense: first row * slots; compact: offsets[first row])
    uint32_t voff = 0;  // (this row's first slot - gbase) * 4

    auto begin_group = [&](uint64_t grp) __attribute__((always_inline)) {
        my_row = (grp << 6) + lane;
        row_ok = my_row < a.n_rows;
        len = 0;
        if (row_ok) len = a.lengths ? a.lengths[my_row] : a.row_len;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(len)); // (the tile it is about to stage was requested earlier still)
        const uint32_t max_len = a.lengths ? wave_max(len) : a.row_len;
        n_chunks = (max_len * CW + CHB - 1) / CHB;
        if (n_chunks == 0) n_chunks = 1;
        e = row_ok ? e_start : 0u;
        count = 0;
        run_start = 0;
        gbase = (grp << 6) * fa.slots; // (group-blocked slots: the same first slot -- group * slots * 64)
        voff = fa.kshift ? (uint32_t)lane * 4u : (uint32_t)lane * fa.slots * 4u;
        cap = fa.count_only ? 0xFFFFFFFFu : fa.slots;
        if (fa.offsets) {
            const uint64_t o0 = row_ok ? fa.offsets[my_row] : 0ull;
            const uint64_t o1 = row_ok ? fa.offsets[my_row + 1] : 0ull;
            gbase = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(o0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)o0);
            voff = (uint32_t)(o0 - gbase) * 4u; // (lane 0 is the group's first row: always there)
            cap = (uint32_t)(o1 - o0);
        }
    };
    // One match (the lanes of `hit`): the reference's find() returned true with start() = pos - (d & 0xFFFF), end() = pos - (d >> 16)
    // (:640-657).  d = codes[code] = (k + length) | k << 16 (needle_device.h).  Stores in the saddr form: uniform base + 32-bit lane offset.
    auto store32 = [&](const void *base, uint32_t off, uint32_t val) __attribute__((always_inline)) {
        asm volatile("global_store_dword %0, %1, %2" : : "v"(off), "v"(val), "s"(base) : "memory");
    };
    auto file = [&](bool hit, uint32_t pos, uint32_t d) __attribute__((always_inline)) {
        if (hit && count < cap && !fa.count_only) {
            const uint32_t off = voff + (count << kshift4); // (v_lshl_add_u32 with the shift in an SGPR)
            if (fa.packed) {
                store32(fa.packed + gbase, off, __umul24(pos, 0x10001u) - d); // start | end << 16
            } else {
                store32(fa.starts + gbase, off, pos - (d & 0xFFFFu));
                store32(fa.ends + gbase, off, pos - (d >> 16));
            }
        }
        count += hit ? 1u : 0u;
    };
    // RUNS: a match [st, en) as it stands -- a run's length is not bounded by 16 bits the way (length, k) codes are (rows of up to 8 MB)
    auto file_se = [&](bool hit, uint32_t st, uint32_t en) __attribute__((always_inline)) {
        if (hit && count < cap && !fa.count_only) {
            const uint32_t off = voff + (count << kshift4);
            if (fa.packed) {
                store32(fa.packed + gbase, off, st | en << 16); // (one-dword forms: rows of at most 65 535 chars)
            } else {
                store32(fa.starts + gbase, off, st);
                store32(fa.ends + gbase, off, en);
            }
        }
        count += hit ? 1u : 0u;
    };
    // The match codes of 8 consecutive chars (char j of them in nibble j of h; pos0 = row index of char 0)
    const bool direct_codes = a.hdr.ft_direct != 0u; // wave-uniform: codes are lengths (k = 0)
    const bool odd_codes = a.hdr.ft_odd != 0u; // wave-uniform: at most 8 codes, all odd -- bit 0 of a nibble = "a match ends here"
    auto decode = [&](uint32_t h, uint32_t pos0) __attribute__((always_inline)) {
        if (RUNS) {
            uint32_t t = h & 0x11111111u;              // bit 4j: a match ends in front of char j
            const uint32_t tf = (h >> 1) & 0x11111111u; // bit 4j: char j may begin a run
            if (fa.count_only) {
                count += (uint32_t)__builtin_popcount(t);
                return;
            }
            if (__ballot(t != 0u) != 0ull) {
                do {
                    const bool has = t != 0u;
                    uint32_t b;
                    asm("v_ffbl_b32 %0, %1" : "=v"(b) : "v"(t));
                    t &= t - 1u;
                    const uint32_t below = tf & ((1u << (b & 31u)) - 1u); // the chars of this log in front of the match's end
                    const uint32_t st = below ? pos0 + ((31u - (uint32_t)__builtin_clz(below)) >> 2) : run_start;
                    const uint32_t en = pos0 + ((b & 31u) >> 2);
                    file_se(has, st, en);
                } while (__ballot(t != 0u) != 0ull);
            }
            run_start = tf ? pos0 + ((31u - (uint32_t)__builtin_clz(tf)) >> 2) : run_start;
            return;
        }
        if (__ballot(h != 0u) == 0ull) return;
        uint32_t t = h; // bit 4j: char j ends a match
        if (!odd_codes) {
            t |= t >> 1;
            t |= t >> 2;
        }
        t &= 0x11111111u;
        if (fa.count_only) { // wave-uniform: nothing is filed, no limit
            count += (uint32_t)__builtin_popcount(t);
            return;
        }
        if (direct_codes) { // wave-uniform: the code names the match's length, k = 0 -- no lookup on the filing chain
            const uint32_t hl = a.hdr.ft_direct == 2u ? h >> 1 : h; // (2: length << 1 | 1)
            do {
                const bool has = t != 0u;
                uint32_t b; // bit index of the next match's nibble (lanes that have none: 0xFFFFFFFF -- nothing filed)
                asm("v_ffbl_b32 %0, %1" : "=v"(b) : "v"(t));
                t &= t - 1u;
                file(has, pos0 + (b >> 2), __builtin_amdgcn_ubfe(hl, b, a.hdr.ft_direct == 2u ? 3u : 4u));
            } while (__ballot(t != 0u) != 0ull);
            return;
        }
        do {
            const bool has = t != 0u;
            uint32_t b;
            asm("v_ffbl_b32 %0, %1" : "=v"(b) : "v"(t));
            t &= t - 1u;
            const uint32_t d = lds_u32(codes_off + (__builtin_amdgcn_ubfe(h, b, 4) << 2));
            file(has, pos0 + (b >> 2), d);
        } while (__ballot(t != 0u) != 0ull);
    };

    // Walk the tile in LDS: chars [ck * CHB / CW, ..) of every row of the group.  Returns the lanes still alive.
    auto walk_tile = [&](uint32_t ck) __attribute__((always_inline)) -> uint64_t {
        const uint32_t idx0 = ck * (uint32_t)(CHB / CW);
        const uint32_t rem = len > idx0 ? len - idx0 : 0u; // GUARD: chars of this row from the tile's start on
        u32x4 v = tile_piece<CHB>(tile, lane, 0);
#pragma unroll
        for (int kk = 0; kk < G::kPieces; ++kk) {
            const uint32_t w[4] = {v[0], v[1], v[2], v[3]};
            if (kk + 1 < G::kPieces) v = tile_piece<CHB>(tile, lane, kk + 1); // next piece: its latency hides below
            uint32_t col[CPP];
            piece_lookups<MODE_TABLE16, CW, GUARD>(wk, w, (uint32_t)(kk * CPP), rem, 0u, col);
            uint32_t h0 = 0, h1 = 0;
#pragma unroll
            for (int i = 0; i < CPP; ++i) {
                e = lds_u16(__umul24(e >> 4, wk.ncols_e) + col[i] + tbase);
                if (i < 8) h0 = __builtin_amdgcn_alignbit(e, h0, 4);
                else h1 = __builtin_amdgcn_alignbit(e, h1, 4);
            }
            decode(h0, idx0 + (uint32_t)(kk * CPP));
            if (CPP > 8) decode(h1, idx0 + (uint32_t)(kk * CPP) + 8u);
        }
        return __ballot((e >> 4) != 0u);
    };
    auto end_group = [&](uint64_t grp) __attribute__((always_inline)) {
        // the row's end: one more transition, on the PAD column -- a pending match is emitted (rows that ended inside a tile took it
        // there and sit in the dead state, whose PAD entry is 0)
        const uint32_t ee = lds_u16(__umul24(e >> 4, wk.ncols_e) + pad_addr);
        const uint32_t code = ee & 15u;
        if (RUNS) {
            if (__ballot(code != 0u) != 0ull) {
                if (fa.count_only) count += code & 1u;
                else file_se((code & 1u) != 0u, run_start, len);
            }
        } else if (__ballot(code != 0u) != 0ull) file(code != 0u, len, lds_u32(codes_off + (code << 2)));
        if (row_ok && fa.counts) fa.counts[my_row] = count < cap ? count : cap;
        if (__ballot(count > cap) != 0ull && lane == 0) *fa.more = 1;
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
            // the registers of a unit are free once its last tile is staged: the next unit -- of this group, or the first one of
            // the wave's next group -- is requested then and arrives while the tile is walked
            auto prefetch = [&]() __attribute__((always_inline)) {
                if (ck + 1 < n_chunks) fetch(g, (ck + 1) / NT);
                else if (g + wave_cnt < last_group) fetch(g + wave_cnt, 0), have_next = true;
            };
            for (;;) {
                stage(std::integral_constant<int, 0>{});
                asm volatile("" ::: "memory");
                if (NT == 1) prefetch();
                asm volatile("" ::: "memory");
                uint64_t live = walk_tile(ck);
                ++ck;
                if (ck >= n_chunks) break;
                if (NT == 1 && live == 0ull) break; // (NT == 2: the line's second half is in registers anyway, and nothing else was asked for)
                if (NT == 2) {
                    stage(std::integral_constant<int, NT - 1>{});
                    asm volatile("" ::: "memory");
                    prefetch();
                    asm volatile("" ::: "memory");
                    live = walk_tile(ck);
                    ++ck;
                    if (ck >= n_chunks || live == 0ull) break;
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
            if (walk_tile(ck) == 0ull) break;
        }
        end_group(g);
    }
}

// ------------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------------
template <int CW, bool GUARD, int CHB, bool RUNS>
static hipError_t launch_ls(const FindAllArgs &fa, int grid, int waves, size_t lds, hipStream_t stream) {
    auto k = find_all_lockstep_kernel<CW, GUARD, CHB, RUNS>;
    static thread_local uint64_t configured = 0;
    if (hipError_t e = allow_full_lds((const void *)k, configured); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(waves * 64), lds, stream, fa);
    return hipGetLastError();
}
template <int CW, bool GUARD>
static hipError_t launch_ls_h(const FindAllArgs &fa, int chb, int grid, int waves, size_t lds, hipStream_t s) {
    if (fa.s.hdr.ft_on == 2u) return chb == 128 ? launch_ls<CW, GUARD, 128, true>(fa, grid, waves, lds, s) : launch_ls<CW, GUARD, 64, true>(fa, grid, waves, lds, s);
    return chb == 128 ? launch_ls<CW, GUARD, 128, false>(fa, grid, waves, lds, s) : launch_ls<CW, GUARD, 64, false>(fa, grid, waves, lds, s);
}

// The kernel forms a match's result address as (uniform 64-bit group base) + (32-bit byte offset of the lane's row inside the group):
// dense slots: 64 rows x slots x 4 B; compact filing: the group's matches x 4 B -- at most one match per char.
bool find_all_lockstep_shape_ok(const FindAllArgs &fa) {
    return (uint64_t)fa.slots < (1ull << 23) && fa.s.stride_bytes < (1ull << 23);  // (blocked: slots * 256 B per group -- the same bound)
}

// One persistent workgroup per CU; the shape (waves x tile bytes) follows the transducer's LDS footprint.
hipError_t launch_find_all_lockstep(int char_width, const FindAllArgs &fa, int n_cus, hipStream_t stream) {
    if (fa.s.n_rows == 0) return hipSuccess;
    if (!fa.s.hdr.ft_on || !find_all_lockstep_shape_ok(fa)) return hipErrorInvalidValue;
    const size_t p = (fa.s.hdr.lds_bytes + 15u) & ~15u, cap = 160u * 1024u;
    static const int cand[7][2] = {{16, 128}, {16, 64}, {14, 64}, {12, 64}, {10, 64}, {8, 64}, {4, 64}};
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
    // unguarded kernels assume every row fills a whole number of tiles
    const bool guard = fa.s.lengths != nullptr || fa.s.row_len == 0 || ((uint64_t)fa.s.row_len * char_width) % chb != 0;
    if (char_width == 1)
        return guard ? launch_ls_h<1, true>(fa, chb, (int)blocks, waves, lds, stream) : launch_ls_h<1, false>(fa, chb, (int)blocks, waves, lds, stream);
    return guard ? launch_ls_h<2, true>(fa, chb, (int)blocks, waves, lds, stream) : launch_ls_h<2, false>(fa, chb, (int)blocks, waves, lds, stream);
}

} // namespace needle
