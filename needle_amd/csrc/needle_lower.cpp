// Lowering of the reference's tables into device programs (see needle_lower.h, needle_device.h).
//
// What gets folded in here so that the kernels' inner loops are pure lookups:
//   * dead state -1            -> sink state 0 (matches / indexForwards / indexBackwards stop there:
//                                 DFAClassBuilder.java:892,461,578) or -> the start state for containedIn
//                                 (its outer loop restarts with state = 0 at the NEXT char, :975-1001,
//                                 and wasAccepted(-1) == wasAccepted(0) == false whenever that matters)
//   * `c > maxChar` exits      -> an OVER column with the same targets (:899-901, :1012-1016, :451-457, :573-575)
//   * containedIn's per-char `if (wasAccepted(state)) return true` (:1008) -> accepting states made absorbing
//   * ragged rows              -> a PAD column (identity for matches/containedIn, sink for the index walks)
//   * wasAccepted<X>(state)    -> states renumbered so that accepted(s) == (s >= A0)
#include "needle_lower.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>

namespace needle {

bool decode_table_string(const char *s, int32_t n_states, int32_t stride, std::vector<int16_t> &out, std::string &err) {
    out.assign((size_t)n_states * stride, (int16_t)-1);
    const char *p = s;
    auto hex = [&](long &v) -> bool {
        char *e = nullptr;
        v = strtol(p, &e, 16);
        if (e == p) return false;
        p = e;
        return true;
    };
    while (*p) {
        long st;
        if (!hex(st) || *p != ':') { err = "malformed table string (state)"; return false; }
        ++p;
        for (;;) {
            long bc, tgt;
            if (!hex(bc) || *p != '-') { err = "malformed table string (class)"; return false; }
            ++p;
            if (!hex(tgt)) { err = "malformed table string (target)"; return false; }
            if (st < 0 || st >= n_states || bc < 0 || bc >= stride || tgt < 0 || tgt > 32767) {
                err = "table string entry out of range";
                return false;
            }
            out[(size_t)st * stride + bc] = (int16_t)tgt;
            if (*p == ',') { ++p; continue; }
            break;
        }
        if (*p == ';') { ++p; continue; }
        if (*p != 0) { err = "malformed table string (separator)"; return false; }
    }
    return true;
}

bool validate_tables(const RefTables &t, std::string &err) {
    if (t.class_map.size() != 65536) { err = "class_map must have 65536 entries"; return false; }
    if (t.stride < 1 || t.stride > 255) { err = "stride (N) out of range"; return false; }
    for (int c = 0; c < 65536; ++c)
        if (t.class_map[c] >= t.stride) { err = "class_map entry >= stride"; return false; }
    for (int w = 0; w < 4; ++w) {
        const RefDfa &d = t.dfa[w];
        if (d.n_states < 1 || d.n_states > 16383) { // DFACompiler.checkForOverLongDFAs, DFACompiler.java:76-83
            err = "n_states out of range (1..16383)";
            return false;
        }
        if (d.table.size() != (size_t)d.n_states * t.stride || d.accepting.size() != (size_t)d.n_states) {
            err = "table/accepting size mismatch";
            return false;
        }
        if (d.max_char < 0 || d.max_char > 0xFFFF) { err = "max_char out of range"; return false; }
        for (int16_t v : d.table)
            if (v < -1 || v >= d.n_states) { err = "table target out of range"; return false; }
    }
    return true;
}

static uint32_t append(std::vector<uint8_t> &blob, const void *src, size_t n) {
    while (blob.size() % 16) blob.push_back(0);
    const uint32_t off = (uint32_t)blob.size();
    const uint8_t *b = (const uint8_t *)src;
    blob.insert(blob.end(), b, b + n);
    return off;
}

// char -> column maps of one automaton: flat uint8[256] for 8-bit rows; page_of[256] + deduplicated pages for UTF-16
struct ColumnMaps {
    std::vector<uint8_t> cmap8, ptab, pages;
};
static ColumnMaps column_maps(const RefTables &t, const RefDfa &d, int char_width) {
    const int OVER = t.stride;
    auto col_of = [&](int c) -> uint8_t { return (uint8_t)(c > d.max_char ? OVER : t.class_map[c]); };
    ColumnMaps m;
    m.cmap8.resize(256);
    m.ptab.resize(256);
    for (int c = 0; c < 256; ++c) m.cmap8[c] = col_of(c);
    if (char_width == 2) {
        std::map<std::vector<uint8_t>, int> seen;
        for (int hi = 0; hi < 256; ++hi) {
            std::vector<uint8_t> pg(256);
            for (int lo = 0; lo < 256; ++lo) pg[lo] = col_of((hi << 8) | lo);
            auto it = seen.find(pg);
            if (it == seen.end()) {
                it = seen.emplace(pg, (int)seen.size()).first;
                m.pages.insert(m.pages.end(), pg.begin(), pg.end());
            }
            m.ptab[hi] = (uint8_t)it->second;
        }
    }
    return m;
}

Program lower(const RefTables &t, Which which, int char_width, size_t lds_table_budget, bool global_walk,
              bool with_backward_maps, bool no_pair) {
    const RefDfa &d = t.dfa[which];
    const int N = t.stride;
    const int n_ref = d.n_states;
    const int n_dev = n_ref + 1;
    const int n_cols = N + 3, OVER = N, PAD = N + 1, PRE = N + 2;

    // device numbering: 0 sink | non-accepting | accepting
    std::vector<int> dev(n_ref);
    int next_id = 1;
    for (int s = 0; s < n_ref; ++s)
        if (!d.accepting[s]) dev[s] = next_id++;
    const int accept_lo = next_id;
    for (int s = 0; s < n_ref; ++s)
        if (d.accepting[s]) dev[s] = next_id++;

    const bool contained = which == W_CONTAINED_IN;
    const int dead = contained ? dev[0] : 0;
    std::vector<uint16_t> next((size_t)n_dev * n_cols, 0); // sink row: all 0
    for (int s = 0; s < n_ref; ++s) {
        uint16_t *row = &next[(size_t)dev[s] * n_cols];
        if (contained && d.accepting[s]) {
            for (int k = 0; k < n_cols; ++k) row[k] = (uint16_t)dev[s];
            continue;
        }
        (void)PRE;
        for (int k = 0; k < N; ++k) {
            const int16_t tgt = d.table[(size_t)s * N + k];
            row[k] = (uint16_t)(tgt < 0 ? dead : dev[tgt]);
        }
        row[OVER] = (uint16_t)dead;
        row[PAD] = (which == W_MATCHES || contained) ? (uint16_t)dev[s] : (uint16_t)0;
        row[PRE] = (uint16_t)dev[s];
    }

    Program p;
    memset(&p.hdr, 0, sizeof(p.hdr));
    p.hdr.n_states = n_dev;
    p.hdr.n_cols = n_cols;
    p.hdr.start = dev[0];
    p.hdr.accept_lo = accept_lo;
    p.hdr.root_accepting = d.accepting[0] ? 1 : 0;
    p.hdr.pad_col = PAD;

    const ColumnMaps cm = column_maps(t, d, char_width);
    p.hdr.n_pages = (uint32_t)(cm.pages.size() / 256);

    if (global_walk) {
        // backward automaton of find(): only the uint16 table, read from HBM/L2 (its column maps travel inside the
        // forward program, see below)
        p.hdr.mode = MODE_GLOBAL;
        p.hdr.off_table = append(p.blob, next.data(), next.size() * 2);
        p.hdr.lds_bytes = 0;
        while (p.blob.size() % 16) p.blob.push_back(0);
        return p;
    }

    Mode mode;
    // packed functions on UTF-16 rows: the pages hold F itself (u32 per code unit of every DISTINCT non-constant page;
    // one shared dword per constant page value), which has to fit the LDS next to the tiles; automata over very many
    // distinct pages take the table modes
    std::vector<int> page_const(p.hdr.n_pages, -1); // column of a constant page, else -1
    size_t n_full = 0;
    for (uint32_t g = 0; g < p.hdr.n_pages; ++g) {
        const uint8_t *pg = &cm.pages[(size_t)g * 256];
        bool same = true;
        for (int i = 1; i < 256 && same; ++i) same = pg[i] == pg[0];
        if (same) page_const[g] = pg[0];
        else ++n_full;
    }
    const size_t pack2_bytes = kLdsPagesF2 + n_full * 1024 + (size_t)n_cols * 4;
    if (n_dev <= 5 && (char_width == 1 || (pack2_bytes <= kMaxPackPagesBytes && pack2_bytes <= lds_table_budget))) mode = MODE_PACK;
    else if (n_dev <= 256) mode = MODE_TABLE8;
    else mode = MODE_TABLE16;

    // Pair mode: the dependent LDS chain of the table modes is what bounds them (4 waves per SIMD cannot hide it), so
    // when [state][col][col] fits, one lookup advances TWO chars.  NEEDLE_PAIR_MAX_BYTES=0 turns it off (A/B, tests).
    static const size_t pair_budget = getenv("NEEDLE_PAIR_MAX_BYTES") ? (size_t)atol(getenv("NEEDLE_PAIR_MAX_BYTES")) : (size_t)(96u << 10);
    const size_t pair_bytes = (size_t)n_dev * n_cols * n_cols * 2;
    if (!no_pair && mode == MODE_TABLE8 && char_width == 1 && pair_bytes <= pair_budget && pair_bytes + 4096 <= lds_table_budget) mode = MODE_PAIR;

    ColumnMaps bm;
    if (with_backward_maps) bm = column_maps(t, t.dfa[W_BACKWARDS], char_width);
    auto emit_backward_maps = [&]() {
        if (!with_backward_maps) return;
        if (char_width == 1) {
            p.hdr.off_bcmap = append(p.blob, bm.cmap8.data(), 256);
        } else {
            p.hdr.off_bptab = append(p.blob, bm.ptab.data(), 256);
            p.hdr.off_bpages = append(p.blob, bm.pages.data(), bm.pages.size());
        }
        // a small backward table rides along in LDS (same layout as the global-walk program: uint16 [n_dev][n_cols])
        const Program bp = lower(t, W_BACKWARDS, char_width, lds_table_budget, true, false, false);
        const size_t tbytes = bp.blob.size() - bp.hdr.off_table;
        if (tbytes <= 2048) p.hdr.off_btable = append(p.blob, bp.blob.data() + bp.hdr.off_table, tbytes);
        else if (bp.hdr.n_cols <= 32 && mode != MODE_GLOBAL && mode != MODE_HYBRID) {
            // (when the FORWARD table itself overflows the LDS, every byte goes to its hot rows instead)
            // a big but sparse backward table (most cells lead to the sink): popcount-compressed rows, if they still fit
            // beside the forward program and 16 waves of 64-byte tiles (else the dense table is walked out of HBM / L2:
            // a chain of L2 round trips per matched row)
            const uint16_t *bt = (const uint16_t *)(bp.blob.data() + bp.hdr.off_table);
            const uint32_t bn = bp.hdr.n_states, bc = bp.hdr.n_cols;
            std::vector<uint32_t> bm(bn, 0);
            std::vector<uint16_t> base(bn, 0), edges;
            for (uint32_t st = 0; st < bn; ++st) {
                base[st] = (uint16_t)edges.size();
                for (uint32_t c = 0; c < bc; ++c)
                    if (bt[(size_t)st * bc + c] != 0) {
                        bm[st] |= 1u << c;
                        edges.push_back(bt[(size_t)st * bc + c]);
                    }
            }
            const size_t sparse_bytes = bm.size() * 4 + base.size() * 2 + edges.size() * 2 + 64;
            if (edges.size() < 65536 && p.blob.size() + sparse_bytes <= (96u << 10) && p.blob.size() + sparse_bytes <= lds_table_budget) {
                p.hdr.off_bsp_bm = append(p.blob, bm.data(), bm.size() * 4);
                p.hdr.off_bsp_base = append(p.blob, base.data(), base.size() * 2);
                p.hdr.off_bsp_edges = append(p.blob, edges.data(), edges.size() * 2);
            }
        }
        // ... and a backward automaton of <= 6 states as packed functions (same device numbering as `bp`): its walk is
        // then 8 independent char -> F lookups and a chain of v_bfe_u32, not 8 x (2-3 dependent lookups)
        const Program pk = lower(t, W_BACKWARDS, char_width, 64u << 10, false, false, false);
        if (pk.hdr.mode == MODE_PACK && (char_width == 1 || pk.hdr.lds_bytes <= (24u << 10))) {
            p.hdr.bpack_start_off = pk.hdr.start_off;
            p.hdr.bpack_accept_off = pk.hdr.accept_off;
            if (char_width == 1) { // the forward layout replicates F per lane (64 KiB): one copy is plenty here
                std::vector<uint32_t> f(256);
                for (int c = 0; c < 256; ++c) memcpy(&f[c], &pk.blob[kLdsF1 + 256 * (size_t)c], 4);
                p.hdr.off_bpack = append(p.blob, f.data(), 1024);
            } else {
                while (p.blob.size() % 1024) p.blob.push_back(0); // F pages stay 1 KiB aligned: base | (lo * 4 & mask)
                const uint32_t o = (uint32_t)p.blob.size();
                p.blob.insert(p.blob.end(), pk.blob.begin(), pk.blob.begin() + pk.hdr.lds_bytes);
                for (int hi = 0; hi < 256; ++hi) { // relative F offsets -> absolute LDS addresses
                    uint32_t base;
                    memcpy(&base, &p.blob[o + kLdsPtab2 + 8 * (size_t)hi], 4);
                    base += o + kLdsPagesF2;
                    memcpy(&p.blob[o + kLdsPtab2 + 8 * (size_t)hi], &base, 4);
                }
                p.hdr.off_bpack = o;
            }
        }
    };
    auto put16 = [&](size_t off, uint32_t v) { p.blob[off] = (uint8_t)(v & 255); p.blob[off + 1] = (uint8_t)(v >> 8); };
    auto put32 = [&](size_t off, uint32_t v) { put16(off, v & 0xFFFF); put16(off + 2, v >> 16); };

    if (mode == MODE_PACK) {
        // F: 5-bit field per state at bit 5*s holding 5*next(s): a transition is v_bfe_u32(F, state_field_offset, 5)
        // field offsets (needle_device.h): non-accepting states (device ids 0 .. accept_lo - 1) at 0, 6, 12, ...;
        // accepting ones at the odd offsets behind them, 6 apart: n_dev <= 5 keeps the last field inside 32 bits
        uint32_t off[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int s = 0; s < n_dev; ++s) off[s] = s < accept_lo ? 6u * s : 6u * accept_lo - 1u + 6u * (s - accept_lo);
        for (int s = 0; s < 8; ++s) p.hdr.pack_off[s] = (uint8_t)off[s];
        p.hdr.start_off = off[dev[0]];
        p.hdr.accept_off = accept_lo < n_dev ? off[accept_lo] : 32u;
        for (int s = 0; s < n_dev; ++s) p.hdr.ident_fn |= off[s] << off[s];
        auto pack = [&](int col) {
            uint32_t F = 0;
            for (int s = 0; s < n_dev; ++s) F |= off[next[(size_t)s * n_cols + col]] << off[s];
            return F;
        };
        p.hdr.pad_f = pack(PAD);
        p.hdr.pre_f = pack(PRE);
        if (char_width == 1) {
            p.blob.assign(256 * 256, 0); // F[byte][64 lane copies] at kLdsF1 = 0: one private LDS bank per lane
            for (int c = 0; c < 256; ++c)
                for (int l = 0; l < 64; ++l) put32(kLdsF1 + 256 * c + 4 * l, pack(cm.cmap8[c]));
        } else {
            p.blob.assign(pack2_bytes, 0);
            // F area: the non-constant pages first (1 KiB each, so that base | (lo * 4) needs no add), then one dword
            // per column for the constant pages
            const uint32_t const_base = (uint32_t)(n_full * 1024);
            for (int k = 0; k < n_cols; ++k) put32(kLdsPagesF2 + const_base + 4 * k, pack(k));
            std::vector<uint32_t> page_base(p.hdr.n_pages), page_mask(p.hdr.n_pages);
            uint32_t next_full = 0;
            for (uint32_t g = 0; g < p.hdr.n_pages; ++g) {
                if (page_const[g] >= 0) {
                    page_base[g] = const_base + 4u * (uint32_t)page_const[g];
                    page_mask[g] = 0;
                } else {
                    page_base[g] = next_full * 1024u;
                    page_mask[g] = 0x3FCu;
                    for (int lo = 0; lo < 256; ++lo) put32(kLdsPagesF2 + page_base[g] + 4 * lo, pack(cm.pages[(size_t)g * 256 + lo]));
                    ++next_full;
                }
            }
            for (int hi = 0; hi < 256; ++hi) {
                put32(kLdsPtab2 + 8 * hi, page_base[cm.ptab[hi]]);
                put32(kLdsPtab2 + 8 * hi + 4, page_mask[cm.ptab[hi]]);
            }
        }
        emit_backward_maps();
        p.hdr.lds_bytes = (uint32_t)p.blob.size();
    } else if (mode == MODE_PAIR) {
        p.blob.assign(kLdsPairTable1 + pair_bytes, 0);
        for (int c = 0; c < 256; ++c) {
            put16(kLdsCmap1 + 2 * c, (uint32_t)cm.cmap8[c] * n_cols * 2u);
            put16(kLdsCmapB1 + 2 * c, (uint32_t)cm.cmap8[c] * 2u);
        }
        for (int s = 0; s < n_dev; ++s)
            for (int c1 = 0; c1 < n_cols; ++c1) {
                const uint32_t s1 = next[(size_t)s * n_cols + c1];
                for (int c2 = 0; c2 < n_cols; ++c2) {
                    const uint32_t s2 = next[(size_t)s1 * n_cols + c2];
                    const uint32_t code = (int)s2 >= accept_lo ? 2u : ((int)s1 >= accept_lo ? 1u : 0u);
                    put16(kLdsPairTable1 + 2 * (((size_t)s * n_cols + c1) * n_cols + c2), s2 | (code << 8));
                }
            }
        p.hdr.off_table = kLdsPairTable1;
        emit_backward_maps();
        p.hdr.lds_bytes = (uint32_t)p.blob.size();
    } else {
        // table modes: element size 1 (uint8 table, or the HBM-resident uint16 table) or 2 (uint16 table in LDS)
        auto build = [&](Mode m) {
            const uint32_t elem = (m == MODE_TABLE16) ? 2u : 1u;
            p.blob.clear();
            if (char_width == 1) {
                p.blob.assign(512, 0); // cmap16 at kLdsCmap1 = 0
                for (int c = 0; c < 256; ++c) put16(kLdsCmap1 + 2 * c, cm.cmap8[c] * elem);
            } else {
                p.blob.assign(kLdsPages2Table + cm.pages.size(), 0);
                for (int hi = 0; hi < 256; ++hi) put16(kLdsPtab2 + 2 * hi, (uint32_t)cm.ptab[hi] * 256u);
                for (size_t i = 0; i < cm.pages.size(); ++i) p.blob[kLdsPages2Table + i] = (uint8_t)(cm.pages[i] * elem);
            }
            if (m == MODE_GLOBAL) {
                emit_backward_maps();
                while (p.blob.size() % 16) p.blob.push_back(0);
                p.hdr.lds_bytes = (uint32_t)p.blob.size();
                p.hdr.off_table = append(p.blob, next.data(), next.size() * 2);
            } else {
                if (m == MODE_TABLE8) {
                    std::vector<uint8_t> t8(next.size());
                    for (size_t i = 0; i < next.size(); ++i) t8[i] = (uint8_t)next[i];
                    p.hdr.off_table = append(p.blob, t8.data(), t8.size());
                } else {
                    p.hdr.off_table = append(p.blob, next.data(), next.size() * 2);
                }
                emit_backward_maps();
                p.hdr.lds_bytes = (uint32_t)p.blob.size();
            }
        };
        const uint32_t elem = (mode == MODE_TABLE16) ? 2u : 1u;
        if (char_width == 2 && (uint32_t)n_cols * elem > 255u) mode = MODE_GLOBAL; // pages hold column * elem in a byte
        build(mode);
        if (mode != MODE_GLOBAL && p.blob.size() > lds_table_budget) {
            mode = MODE_GLOBAL;
            build(mode);
            // Hot rows in LDS + the whole table in HBM.  Search automata on real text sit in the few states near their
            // start state (a 1000-keyword union over random text: 99.8 % of the steps are in states of depth <= 3), so
            // the rows of the first states in breadth-first order are the ones worth the LDS.  The hot prefix is sized
            // to leave room for 16 waves x 64-byte tiles.  NEEDLE_HYBRID=0: plain HBM table (tests, A/B).
            static const bool hybrid_on = !(getenv("NEEDLE_HYBRID") && atoi(getenv("NEEDLE_HYBRID")) == 0);
            const size_t row_bytes = (size_t)n_cols * 2;
            const bool cols_ok = char_width == 1 || row_bytes <= 255;
            const size_t room = std::min<size_t>(lds_table_budget, 96u << 10);
            const size_t fixed = p.hdr.lds_bytes + 64; // column maps (+ backward maps) as just built for the HBM-table layout
            const size_t hot_rows = room > fixed ? std::min<size_t>((room - fixed) / row_bytes, (size_t)n_dev) : 0;
            if (hybrid_on && cols_ok && hot_rows >= 32 && n_dev <= 0x8000) {
                // breadth-first numbering from the start state (0 stays the sink)
                std::vector<int> bfs(n_ref, -1), order;
                bfs[0] = 1;
                order.push_back(0);
                for (size_t h = 0; h < order.size(); ++h)
                    for (int k = 0; k < N; ++k) {
                        const int16_t tgt = d.table[(size_t)order[h] * N + k];
                        if (tgt >= 0 && bfs[tgt] < 0) {
                            bfs[tgt] = (int)order.size() + 1;
                            order.push_back(tgt);
                        }
                    }
                for (int st = 0; st < n_ref; ++st) // (unreachable states, if any, go last)
                    if (bfs[st] < 0) { bfs[st] = (int)order.size() + 1; order.push_back(st); }
                auto flagged = [&](int ref_state) { return (uint16_t)(bfs[ref_state] | (d.accepting[ref_state] ? 0x8000 : 0)); };
                const uint16_t dead_h = contained ? flagged(0) : (uint16_t)0;
                std::vector<uint16_t> nh((size_t)n_dev * n_cols, 0);
                for (int st = 0; st < n_ref; ++st) {
                    uint16_t *row = &nh[(size_t)bfs[st] * n_cols];
                    if (contained && d.accepting[st]) {
                        for (int k = 0; k < n_cols; ++k) row[k] = flagged(st);
                        continue;
                    }
                    for (int k = 0; k < N; ++k) {
                        const int16_t tgt = d.table[(size_t)st * N + k];
                        row[k] = tgt < 0 ? dead_h : flagged(tgt);
                    }
                    row[OVER] = dead_h;
                    row[PAD] = (which == W_MATCHES || contained) ? flagged(st) : (uint16_t)0;
                    row[PRE] = flagged(st);
                }
                // layout: column maps (element size 2) | hot rows | backward maps || whole table (HBM only)
                p.blob.clear();
                p.hdr.off_bcmap = p.hdr.off_bptab = p.hdr.off_bpages = p.hdr.off_btable = p.hdr.off_bpack = 0;
                if (char_width == 1) {
                    p.blob.assign(512, 0);
                    for (int c = 0; c < 256; ++c) put16(kLdsCmap1 + 2 * c, cm.cmap8[c] * 2u);
                } else {
                    p.blob.assign(kLdsPages2Table + cm.pages.size(), 0);
                    for (int hi = 0; hi < 256; ++hi) put16(kLdsPtab2 + 2 * hi, (uint32_t)cm.ptab[hi] * 256u);
                    for (size_t i = 0; i < cm.pages.size(); ++i) p.blob[kLdsPages2Table + i] = (uint8_t)(cm.pages[i] * 2u);
                }
                p.hdr.hot_bytes = (uint32_t)(hot_rows * row_bytes);
                p.hdr.off_table = append(p.blob, nh.data(), p.hdr.hot_bytes);
                emit_backward_maps();
                while (p.blob.size() % 16) p.blob.push_back(0);
                p.hdr.lds_bytes = (uint32_t)p.blob.size();
                p.hdr.off_gtable = append(p.blob, nh.data(), nh.size() * 2);
                p.hdr.start = flagged(0);
                p.hdr.accept_lo = 0x8000;
                mode = MODE_HYBRID;
            }
        }
    }
    while (p.blob.size() % 16) p.blob.push_back(0);
    p.hdr.mode = mode;
    return p;
}

} // namespace needle
