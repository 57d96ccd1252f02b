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
#include "needle_ngram_host.h"
#include <algorithm>
#include <array>
#include <cstdio>
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

// ---- window addressing (needle_device.h) ------------------------------------------------------------------------------
// The reference walks `state = T[cls + state * N]` with `cls = BYTE_CLASSES[c]` (DFAClassBuilder.java:438-468): two dependent
// loads per char.  On the GPU the class lookup is an LDS read per char that competes with the transition lookups for the
// same LDS cycles (a third of them on a keyword dictionary).  When the char -> column map is constant below some char and
// constant above another (an ASCII-letter dictionary: everything below 'a' and everything above 'z' is "other"), the
// table can be indexed by the CHAR instead -- columns = the chars of the window, each a copy of its class's column -- and
// the lookup becomes a clamp in registers.
namespace {
struct Window {
    bool ok = false;
    int cl = 0, ch = 0, W = 0; // chars cl .. ch are the table's columns; cl stands for every char below it, ch for every one above
    std::vector<int> cols;     // class column (OVER included) of char cl + j
};
// `same(a, b)`: columns a and b have the same content in this automaton (such chars may share a run)
template <typename Same>
Window find_window(const ColumnMaps &cm, int char_width, Same same) {
    const int maxc = char_width == 1 ? 255 : 65535;
    auto col = [&](int c) -> int { return char_width == 1 ? cm.cmap8[c] : cm.pages[(size_t)cm.ptab[c >> 8] * 256 + (c & 255)]; };
    int p = 1;
    while (p <= maxc && same(col(p), col(0))) ++p; // chars [0, p) behave alike
    int q = maxc;
    while (q > 0 && same(col(q - 1), col(maxc))) --q; // chars [q, maxc] behave alike
    Window w;
    w.cl = p - 1;
    if (w.cl & 1) --w.cl; // (even: the hot-rows mode reads its HBM table in dwords at win_lo_e-biased offsets)
    w.ch = std::max(q, w.cl);
    w.W = w.ch - w.cl + 1;
    if (w.W > 1024) return w;
    w.cols.resize(w.W);
    for (int j = 0; j < w.W; ++j) w.cols[j] = col(w.cl + j);
    w.ok = true;
    return w;
}
} // namespace

// ---- MODE_SPARSE: "dense rows near the start state + default-row / exception records for every other state" ----------
// The reference's table (DFAClassBuilder.java:317-333; walked at :438-468) is a dense [state][class] array.  Search
// automata of big alternations (a keyword dictionary) have thousands of states whose rows differ from the row of a
// shallower state in one or two cells (Aho-Corasick's failure links, seen from the table).  Here the first states in
// breadth-first order keep dense rows; every other state is stored as the dense row it mostly equals plus a chain of
// {column, target} exception records.  The result is verified cell by cell against the dense table before it is used.
namespace {
struct SparseImage {
    std::vector<uint8_t> img; // relative to the table base
    uint32_t rec_base = 0, accept_rec = 0, rows_base = 0, start = 0, accept_lo = 0, chains = 0, dense = 0, records = 0;
    // lengths programs (n_dead > 0): the dead-with-a-match-pending states D_1 .. D_K own the first dense rows, so that "the search
    // is over" is value <= dead_hi; dead_row0 = the address field (row address / 4) of D_1's row, D_k's = dead_row0 + (k - 1) * NC
    uint32_t dead_hi = 0, dead_row0 = 0, end_key = 0;
};

constexpr int kSparseMaxChain = 3; // states that need more exceptions than this against every candidate row become dense

// next_full: device-numbered table [n_dev][n_cols_full] (0 = sink | non-accepting | accepting from accept_lo_dev on); only
// the columns listed in `cols` take part, renumbered 0 .. cols.size() - 1 in that order.  room: bytes the image may take.
bool build_sparse(const std::vector<uint16_t> &next_full, int n_dev, int n_cols_full, const std::vector<int> &cols, int col_bias, int start_dev,
                  int accept_lo_dev, size_t room, SparseImage &out, int n_dead = 0, const std::vector<uint16_t> *end_tgt = nullptr) {
    // end_tgt (lengths programs): per state, where the END of the row leads -- the D_L of its pending length, or 0.  END is no
    // column of the dense rows (a cell per row for the few states that have a match pending would not fit: C3-sparse has 76 bytes
    // to spare): the states with a target keep a record of their own for it, keyed NC * 4 (one past the last column), which
    // only finish_rows' record-chain walk (needle_scan.h, sparse_end) ever asks for; they are never dense.
    auto pending = [&](int s) { return end_tgt && s > n_dead && (*end_tgt)[s] != 0; };
    // col_bias (window addressing): the walk's column offsets are (col_bias + j) * 4, not rebased -- every row sits col_bias
    // cells further up than its address says, and the record keys carry the bias
    const int NC = (int)cols.size();
    const size_t row_bytes = (size_t)NC * 4;
    const uint32_t bias4 = (uint32_t)col_bias * 4u;
    if (n_dev < 2 || NC < 1 || (NC + col_bias) * 4 > 0xFFFC) return false;
    if (room <= bias4) return false;
    room -= bias4;
    // only the columns some char of the haystack's width maps to take part (8-bit rows never see the class of U+FFFF, nor
    // OVER when maxChar >= 255): compacted copy [n_dev][NC]
    std::vector<uint16_t> next((size_t)n_dev * NC);
    for (int s = 0; s < n_dev; ++s)
        for (int j = 0; j < NC; ++j) next[(size_t)s * NC + j] = next_full[(size_t)s * n_cols_full + cols[j]];
    n_cols_full = NC;
    auto cell = [&](int s, int c) -> int { return next[(size_t)s * n_cols_full + c]; };
    // breadth-first order from the start state; def[s] = the state the automaton would be in had it not seen the first
    // char of the shortest string leading to s (Aho-Corasick's failure state, derived from the table alone): the row s most
    // likely shares
    std::vector<int> order, pos(n_dev, -1), def(n_dev, 0);
    order.reserve(n_dev);
    pos[start_dev] = 0;
    def[start_dev] = start_dev;
    order.push_back(start_dev);
    for (size_t h = 0; h < order.size(); ++h) {
        const int p = order[h];
        for (int c = 0; c < NC; ++c) {
            const int s = cell(p, c);
            if (s == 0 || pos[s] >= 0) continue;
            pos[s] = (int)order.size();
            order.push_back(s);
            int d = p == start_dev ? start_dev : cell(def[p], c);
            if (d == s) d = start_dev;
            def[s] = d;
        }
    }
    const int S = (int)order.size(); // reachable states without the sink (unreachable ones -- containedIn's accepting states
                                     // but the first, see lower() -- get no storage: no cell leads to them)

    std::vector<uint8_t> is_dense(n_dev, 0);
    std::vector<int> dflt(n_dev, 0);
    std::vector<std::vector<int>> exc(n_dev); // columns in which s differs from its default row
    uint64_t work = 0;
    const uint64_t work_cap = 600ull * 1000 * 1000;
    auto diff = [&](int s, int d, int stop) { // number of differing cells, counting stops beyond `stop`
        int n = 0;
        const uint16_t *a = &next[(size_t)s * n_cols_full], *b = &next[(size_t)d * n_cols_full];
        for (int c = 0; c < NC && n <= stop; ++c) n += a[c] != b[c];
        work += (uint64_t)NC;
        return n;
    };
    // with the BFS prefix [0, D) dense: defaults and exceptions of every other state; returns the image size
    auto evaluate = [&](int D) -> size_t {
        std::fill(is_dense.begin(), is_dense.end(), 0);
        is_dense[0] = 1;
        size_t n_rec = 0, n_dense = 0;
        for (int i = 0; i < D; ++i)
            if (!pending(order[i])) is_dense[order[i]] = 1, ++n_dense;
        for (int d = 1; d <= n_dead; ++d) // (lengths programs: the D_L states keep rows of their own)
            if (!is_dense[d]) is_dense[d] = 1, ++n_dense;
        for (int i = 0; i < S; ++i) {
            const int s = order[i];
            if (is_dense[s]) continue;
            int best = 0, best_n = diff(s, 0, NC); // the sink's row (all cells 0) is always a candidate
            auto consider = [&](int d) {
                if (!is_dense[d] || d == best) return;
                const int n = diff(s, d, best_n);
                if (n < best_n) best = d, best_n = n;
            };
            int x = def[s];
            for (int hop = 0; hop < 64 && x != 0; ++hop) { // the nearest dense state(s) on the failure chain
                if (is_dense[x]) { consider(x); break; }
                const int nx = def[x];
                if (nx == x) break;
                x = nx;
            }
            consider(start_dev);
            for (int d = 1; d <= n_dead; ++d) consider(d); // (a state every char of which ends the match: D_L's own row)
            if (best_n > 1 && work < work_cap) // the heuristic candidates are poor: look at every dense row
                for (int j = 0; j < D && best_n > 1; ++j) consider(order[j]);
            if (best_n > kSparseMaxChain && !pending(s)) { // no row is close: the state keeps a dense row of its own
                is_dense[s] = 2;
                ++n_dense;
                continue;
            }
            dflt[s] = best;
            exc[s].clear();
            for (int c = 0; c < NC; ++c)
                if (cell(s, c) != cell(best, c)) exc[s].push_back(c);
            if (pending(s)) exc[s].push_back(NC); // the END record, last in the chain
            n_rec += exc[s].size();
        }
        return ((row_bytes + 7) & ~(size_t)7) + (2 + n_rec) * 8 + n_dense * row_bytes + 8;
    };
    // the largest dense prefix whose image fits the room
    int D = (int)std::min<size_t>((size_t)S, room / row_bytes);
    size_t bytes = 0;
    for (int iter = 0; iter < 40; ++iter) {
        bytes = evaluate(D);
        if (getenv("NEEDLE_SPARSE_DEBUG")) fprintf(stderr, "[sparse] iter %d: S=%d D=%d bytes=%zu room=%zu\n", iter, S, D, bytes, room);
        if (bytes <= room) break;
        const size_t over = bytes - room, per_row = row_bytes > 16 ? row_bytes - 8 : 8;
        const int drop = (int)((over + per_row - 1) / per_row);
        if (D == 0) return false;
        D = std::max(0, D - std::max(1, drop));
    }
    static const bool dbg = getenv("NEEDLE_SPARSE_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[sparse] S=%d NC=%d room=%zu D=%d bytes=%zu work=%llu\n", S, NC, room, D, bytes, (unsigned long long)work);
    if (bytes > room) return false;
    // spare room: promote the states with the longest chains (they cost extra LDS round trips), breadth-first among equals
    for (int want = kSparseMaxChain; want >= 2; --want)
        for (int i = D; i < S; ++i) {
            const int s = order[i];
            if (is_dense[s] || (int)exc[s].size() != want || pending(s)) continue;
            const size_t nb = bytes + row_bytes - 8 * exc[s].size();
            if (nb > room) continue;
            is_dense[s] = 2;
            exc[s].clear();
            bytes = nb;
        }

    // ---- addresses (physical, relative to the table base; a row's ADDRESS field is its physical start minus bias4)
    const uint32_t rec_base = (uint32_t)((bias4 + row_bytes + 7) & ~(size_t)7);
    std::vector<uint32_t> rec_at(n_dev, 0); // address of a sparse state's first record (its acceptance class's dummy when it has none)
    uint32_t at = rec_base;
    const uint32_t dummy_nonacc = at;
    at += 8;
    uint32_t n_records = 0, chains = 0;
    auto place = [&](bool accepting) {
        for (int i = 0; i < S; ++i) {
            const int s = order[i];
            if (is_dense[s] || (s >= accept_lo_dev) != accepting) continue;
            if (exc[s].empty()) continue;
            rec_at[s] = at;
            at += 8u * (uint32_t)exc[s].size();
            n_records += (uint32_t)exc[s].size();
            if (exc[s].size() > 1) chains = 1;
        }
    };
    place(false);
    const uint32_t dummy_acc = at;
    at += 8;
    place(true);
    if (dbg) fprintf(stderr, "[sparse] records end at %u (%u records, chains %u)\n", at, n_records, chains);
    if (at > 0x10000u) return false; // record addresses are 16-bit fields
    const uint32_t rows_base = (at + 3u) & ~3u;
    std::vector<uint32_t> row_at(n_dev, 0);
    uint32_t n_dense = 0;
    {
        uint32_t r = rows_base;
        for (int d = 1; d <= n_dead; ++d) { // D_1 .. D_K first, one row apart whether reachable or not
            row_at[d] = r;
            r += (uint32_t)row_bytes;
            if (is_dense[d]) ++n_dense;
        }
        for (int i = 0; i < S; ++i)
            if (is_dense[order[i]] && !(order[i] >= 1 && order[i] <= n_dead)) { row_at[order[i]] = r; r += (uint32_t)row_bytes; ++n_dense; }
        if (r / 4 > 0xFFFFu || r > room + bias4 + 8) return false;
        out.img.assign(r, 0);
    }
    row_at[0] = bias4; // the sink's row
    auto value = [&](int s) -> uint32_t { // state value: recB << 16 | rowA4
        if (s == 0) return 0u;
        const bool acc = s >= accept_lo_dev;
        if (is_dense[s]) return ((acc ? dummy_acc : dummy_nonacc) << 16) | ((row_at[s] - bias4) >> 2);
        const uint32_t rec = exc[s].empty() ? (acc ? dummy_acc : dummy_nonacc) : rec_at[s];
        return (rec << 16) | ((row_at[dflt[s]] - bias4) >> 2); // (the sink's row: address 0)
    };
    auto put32 = [&](uint32_t off, uint32_t v) { memcpy(&out.img[off], &v, 4); };
    put32(dummy_nonacc, 0xFFFFu); // key 0xFFFF: no column * 4 equals it; no successor; target unused
    put32(dummy_acc, 0xFFFFu);
    for (int i = 0; i < S; ++i) {
        const int s = order[i];
        if (is_dense[s]) {
            for (int c = 0; c < NC; ++c) put32(row_at[s] + 4u * (uint32_t)c, value(cell(s, c)));
        } else {
            for (size_t k = 0; k < exc[s].size(); ++k) {
                const uint32_t a = rec_at[s] + 8u * (uint32_t)k;
                const uint32_t nxt = k + 1 < exc[s].size() ? a + 8u : 0u;
                put32(a, ((uint32_t)exc[s][k] * 4u + bias4) | (nxt << 16));
                put32(a + 4, value(exc[s][k] == NC ? (int)(*end_tgt)[s] : cell(s, exc[s][k])));
            }
        }
    }
    out.rec_base = rec_base;
    out.accept_rec = dummy_acc;
    out.rows_base = rows_base;
    out.start = value(start_dev);
    out.end_key = (uint32_t)NC * 4u + bias4;
    if (n_dead) {
        for (int d = 1; d <= n_dead; ++d)
            if (d >= accept_lo_dev || (pos[d] >= 0 && !is_dense[d])) return false;
        out.dead_row0 = (row_at[1] - bias4) >> 2;
        out.dead_hi = (dummy_nonacc << 16) | ((row_at[n_dead] - bias4) >> 2);
    }
    out.accept_lo = dummy_acc << 16;
    out.chains = chains;
    out.dense = n_dense;
    out.records = n_records;

    // ---- verification: walk the image the way the kernel does (needle_walk.h, apply<MODE_SPARSE>) for every state and
    // column and compare with the dense table
    auto rd32 = [&](uint32_t off) -> uint32_t {
        uint32_t v = 0;
        if ((size_t)off + 4 <= out.img.size()) memcpy(&v, &out.img[off], 4); // (out-of-range DS reads return 0)
        return v;
    };
    for (int s = 0; s < n_dev; ++s) {
        if (s != 0 && pos[s] < 0) continue;
        const uint32_t sv = value(s);
        if ((sv >= out.accept_lo) != (s >= accept_lo_dev && s != 0)) return false;
        for (int c = 0; c < NC; ++c) {
            const uint32_t col4 = (uint32_t)c * 4u + bias4;
            const uint32_t a = rd32((sv & 0xFFFFu) * 4u + col4);
            uint32_t b0 = rd32(sv >> 16), b1 = rd32((sv >> 16) + 4);
            bool hit = (b0 & 0xFFFFu) == col4;
            uint32_t nx = hit ? b1 : a;
            int guard = 0;
            while (!hit && b0 > 0xFFFFu && guard++ < 16) {
                const uint32_t r = b0 >> 16;
                b0 = rd32(r);
                b1 = rd32(r + 4);
                hit = (b0 & 0xFFFFu) == col4;
                nx = hit ? b1 : nx;
            }
            if (nx != value(cell(s, c))) {
                if (dbg) fprintf(stderr, "[sparse] verification failed: state %d column %d: got %08x want %08x (target %d, dense %d/%d, exc %zu, dflt %d, sv %08x)\n", s, c, nx,
                                 value(cell(s, c)), cell(s, c), (int)is_dense[s], (int)is_dense[cell(s, c)], exc[s].size(), dflt[s], sv);
                return false;
            }
        }
        if (end_tgt && s > n_dead) { // END: the record chain alone (needle_scan.h, sparse_end); no record = the sink
            uint32_t b0 = rd32(sv >> 16), b1 = rd32((sv >> 16) + 4), r = 0;
            int guard = 0;
            for (;;) {
                if ((b0 & 0xFFFFu) == out.end_key) { r = b1; break; }
                if (b0 <= 0xFFFFu || guard++ > 16) break;
                const uint32_t nr = b0 >> 16;
                b0 = rd32(nr);
                b1 = rd32(nr + 4);
            }
            if (r != value((int)(*end_tgt)[s])) {
                if (dbg) fprintf(stderr, "[sparse] verification failed: state %d END\n", s);
                return false;
            }
        }
    }
    return true;
}
} // namespace

namespace {
// what the n-gram filter analysis (needle_ngram_host.cpp) needs of a lowering: the device-numbered table before any encoding
struct LowerAux {
    std::vector<uint16_t> next;
    std::vector<uint8_t> cmap8;
    std::vector<uint8_t> cmap16; // UTF-16 lowerings: the column of every code unit (65 536 entries)
    int n_dev = 0, n_cols = 0, start = 0, accept_lo = 0, dead_hi = 0;
    bool ml_in_hbm = false; // lower_filter_hbm: a lengths program may keep the HBM-table layout (pend[] then rides behind the table, in HBM)
};
} // namespace
static Program lower_core(const RefTables &t, Which which, int char_width, size_t lds_table_budget, bool global_walk,
                          bool with_backward_maps, bool no_pair, const MatchLengths *ml, LowerAux *aux);

// NEEDLE_PREFILTER: 0 = never build / use the n-gram candidate filter, 1 (default) = for automata in the compressed form
// (MODE_SPARSE: the walk is latency-bound there), 2 = for every LDS table automaton that allows one (tests, A/B)
int ngram_level() {
    static const int level = getenv("NEEDLE_PREFILTER") ? atoi(getenv("NEEDLE_PREFILTER")) : 1;
    return level;
}

Program lower(const RefTables &t, Which which, int char_width, size_t lds_table_budget, bool global_walk,
              bool with_backward_maps, bool no_pair, const MatchLengths *ml) {
    // The scan kernels' containedIn / forward programs on 8-bit rows may carry an n-gram candidate filter (needle_ngram_host.h).
    const bool want = ngram_level() > 0 && char_width == 1 && !global_walk && !no_pair && lds_table_budget > 0 &&
                      (which == W_CONTAINED_IN || which == W_FORWARDS);
    LowerAux aux;
    Program p = lower_core(t, which, char_width, lds_table_budget, global_walk, with_backward_maps, no_pair, ml, want ? &aux : nullptr);
    memset(&p.ng.p, 0, sizeof(p.ng.p));
    if (!want || p.blob.empty()) return p;
    const bool mode_ok = p.hdr.mode == MODE_SPARSE || (ngram_level() > 1 && (p.hdr.mode == MODE_TABLE8 || p.hdr.mode == MODE_TABLE16));
    if (!mode_ok) return p;
    p.ng = build_ngram_filter(aux.next.data(), aux.n_dev, aux.n_cols, aux.cmap8.data(), aux.start, aux.accept_lo, aux.dead_hi,
                              which == W_CONTAINED_IN, p.hdr.lds_bytes);
    return p;
}

// The n-gram candidate filter in front of an automaton that fits the LDS in NO form (a dictionary at the reference's limit of 16 383
// states, DFACompiler.java:76-83: 3000 keywords of 6..8 chars = 12 270 states, 690 KB as a table).  The filter kernel then keeps only
// the Bloom bitmap (and the 512-byte column map) in LDS and verifies its candidates -- ~3 per KiB of text, ~10 steps each -- by
// walking the plain uint16 table out of HBM / L2.  which: W_CONTAINED_IN, or W_FORWARDS with ml (start = end - pend[stop state]) or
// for one-length patterns.  8-bit rows.  ng.p.on = 0: no filter (the blob is still a valid HBM-table program).
Program lower_filter_hbm(const RefTables &t, Which which, const MatchLengths *ml, bool with_backward_maps) {
    LowerAux aux;
    aux.ml_in_hbm = true;
    // (with_backward_maps: find() of a pattern without bounded match lengths -- the backward automaton's column maps ride in the LDS part,
    // the filter kernel's verified candidates find their starts by indexBackwards)
    Program p = lower_core(t, which, 1, 0, false, with_backward_maps, false, ml, &aux);
    memset(&p.ng.p, 0, sizeof(p.ng.p));
    if (p.blob.empty() || p.hdr.mode != MODE_GLOBAL || ngram_level() <= 0) return p;
    p.ng = build_ngram_filter(aux.next.data(), aux.n_dev, aux.n_cols, aux.cmap8.data(), aux.start, aux.accept_lo, aux.dead_hi, which == W_CONTAINED_IN,
                              p.hdr.lds_bytes);
    return p;
}

// The WIDE filter program (needle_ngram.h): UTF-16 rows of a pattern that lives on MORE than one page of the BMP (Latin + Cyrillic + CJK
// keyword dictionaries: DFA.java:438-463 -- the reference's class map covers all 65 536 code units of any pattern, its prefilters run on any
// String, DFAClassBuilder.java:365-376).  The filter hashes windows of four 16-bit code units as they stand (no narrowing to a byte
// program); its candidates are verified on the UTF-16 HBM-table program -- the two-level page map (512 bytes + 256 per distinct page) is
// all the LDS holds of the automaton.  ng.p.on = 0: no filter.
Program lower_filter_wide(const RefTables &t, Which which, const MatchLengths *ml) {
    LowerAux aux;
    aux.ml_in_hbm = true;
    Program p = lower_core(t, which, 2, 0, false, false, false, ml, &aux);
    memset(&p.ng.p, 0, sizeof(p.ng.p));
    if (p.blob.empty() || p.hdr.mode != MODE_GLOBAL || ngram_level() <= 0 || aux.cmap16.size() != 65536) return p;
    p.ng = build_ngram_filter(aux.next.data(), aux.n_dev, aux.n_cols, aux.cmap8.data(), aux.start, aux.accept_lo, aux.dead_hi, which == W_CONTAINED_IN,
                              p.hdr.lds_bytes, aux.cmap16.data());
    return p;
}

static Program lower_core(const RefTables &t, Which which, int char_width, size_t lds_table_budget, bool global_walk,
                          bool with_backward_maps, bool no_pair, const MatchLengths *ml, LowerAux *aux) {
    // ml (W_FORWARDS only): the refined "lengths" automaton stands in for the reference's search automaton (needle_lower.h)
    const RefDfa &d = ml ? ml->dfa : t.dfa[which];
    const int N = t.stride;
    const int n_ref = d.n_states;
    // (the find-all form of the lengths automaton -- ml && no_pair -- carries skip states: needle_device.h fa_skip_lo)
    const int n_skip = (ml && no_pair) ? 16 / char_width - 1 : 0;
    const int n_dev = n_ref + 1 + n_skip;
    const int n_cols = N + 3, OVER = N, PAD = N + 1, PRE = N + 2;

    // device numbering: 0 sink | [lengths form: the dead-with-a-match-pending states D_L, so that "the search is over" is
    // state <= fa_dead_n] | non-accepting | accepting
    std::vector<int> dev(n_ref);
    int next_id = 1;
    if (ml)
        for (int s = 1; s <= ml->n_dead; ++s) dev[s] = next_id++;
    const int skip_lo = next_id;
    next_id += n_skip;
    for (int s = 0; s < n_ref; ++s)
        if (!d.accepting[s] && !(ml && s >= 1 && s <= ml->n_dead)) dev[s] = next_id++;
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
        if (ml) {
            // a char beyond the refined automaton's maxChar has a target of its own per state; the end of the row (PAD) ends the
            // search like any dying transition: with a match pending it leads to D_L, not to the sink
            row[OVER] = (uint16_t)(ml->over[s] < 0 ? 0 : dev[ml->over[s]]);
            int dl = 0;
            for (int k = 1; k <= ml->n_dead; ++k)
                if (ml->pend[s] && ml->pend[k] == ml->pend[s]) dl = dev[k];
            row[PAD] = (uint16_t)dl;
        }
    }

    for (int k = 1; k <= n_skip; ++k) { // S_k -> S_(k-1) on every column, S_1 -> the start state
        uint16_t *row = &next[(size_t)(skip_lo + k - 1) * n_cols];
        for (int c = 0; c < n_cols; ++c) row[c] = (uint16_t)(k == 1 ? dev[0] : skip_lo + k - 2);
    }

    if (aux) {
        aux->next = next;
        aux->n_dev = n_dev;
        aux->n_cols = n_cols;
        aux->start = dev[0];
        aux->accept_lo = accept_lo;
        aux->dead_hi = ml ? ml->n_dead : 0;
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
    if (aux) aux->cmap8 = cm.cmap8;
    if (aux && char_width == 2) {
        aux->cmap16.resize(65536);
        for (int c = 0; c < 65536; ++c) aux->cmap16[(size_t)c] = cm.pages[(size_t)cm.ptab[c >> 8] * 256 + (c & 255)];
    }

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
    // (ml: the scan kernels' lengths programs take the pair table too -- D_L states are ordinary absorbing rows of it)
    if (!no_pair && mode == MODE_TABLE8 && char_width == 1 && pair_bytes <= pair_budget && pair_bytes + 4096 <= lds_table_budget) mode = MODE_PAIR;

    // Window addressing for the table modes (needle_device.h; not for the plain layouts other kernels walk: global_walk, the
    // HBM-table variant with no LDS budget, the find-all programs).  NEEDLE_WINDOW=0 turns it off (A/B, tests).
    Window win;
    {
        static const bool window_on = !(getenv("NEEDLE_WINDOW") && atoi(getenv("NEEDLE_WINDOW")) == 0);
        // (the find-all kernel's lengths program -- ml && no_pair -- takes window addressing too since round 4: NEEDLE_FIND_ALL_WINDOW=0 off)
        static const bool fa_window = !(getenv("NEEDLE_FIND_ALL_WINDOW") && atoi(getenv("NEEDLE_FIND_ALL_WINDOW")) == 0);
        if (window_on && (!no_pair || (ml && fa_window)) && lds_table_budget > 0 && mode != MODE_PACK && mode != MODE_PAIR) {
            auto same = [&](int a, int b) {
                if (a == b) return true;
                for (int st = 0; st < n_dev; ++st)
                    if (next[(size_t)st * n_cols + a] != next[(size_t)st * n_cols + b]) return false;
                return true;
            };
            win = find_window(cm, char_width, same);
            // worth it when the window is not much wider than the classes it replaces (a dictionary over [a-z]: 28 chars for 29
            // classes), or the table is small anyway
            const size_t e = mode == MODE_TABLE16 ? 2 : 1;
            const size_t bytes_w = (size_t)n_dev * (win.W + 2) * e, bytes_c = (size_t)n_dev * n_cols * e;
            if (win.ok && !(bytes_w <= bytes_c * 5 / 4 || bytes_w <= (16u << 10))) win.ok = false;
        }
    }
    // a table in window layout: columns = the window's chars (each a copy of its class's column), then PAD and PRE
    auto window_table = [&](const std::vector<uint16_t> &tab, int n_rows) {
        const int nw = win.W + 2;
        std::vector<uint16_t> o((size_t)n_rows * nw);
        for (int st = 0; st < n_rows; ++st) {
            for (int j = 0; j < win.W; ++j) o[(size_t)st * nw + j] = tab[(size_t)st * n_cols + win.cols[j]];
            o[(size_t)st * nw + win.W] = tab[(size_t)st * n_cols + PAD];
            o[(size_t)st * nw + win.W + 1] = tab[(size_t)st * n_cols + PRE];
        }
        return o;
    };
    auto set_window_header = [&](uint32_t elem) {
        p.hdr.win_on = 1;
        p.hdr.win_lo_e = (uint32_t)win.cl * elem;
        p.hdr.win_hi_e = (uint32_t)win.ch * elem;
        p.hdr.n_cols = (uint32_t)win.W + 2;
        p.hdr.pad_col = (uint32_t)(win.cl + win.W); // (column offsets are not rebased: PAD is "char ch + 1", PRE "char ch + 2")
    };
    auto clear_window_header = [&]() {
        p.hdr.win_on = p.hdr.win_lo_e = p.hdr.win_hi_e = 0;
        p.hdr.n_cols = n_cols;
        p.hdr.pad_col = PAD;
    };
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
        // (up to 2 KB always; up to 8 KB when the program then still leaves room for 16 waves x 64-byte tiles -- C5w's reversed automaton is
        // 34 states x 33 columns = 2244 bytes, its 33 columns rule out the popcount form below, and out of HBM / L2 every step of every
        // matched row's backward walk waited for a load: NEEDLE_BTABLE_LDS_MAX=2048 brings that back, A/B)
        static const size_t btable_max = getenv("NEEDLE_BTABLE_LDS_MAX") ? (size_t)atol(getenv("NEEDLE_BTABLE_LDS_MAX")) : (size_t)8192;
        const bool roomy = tbytes <= btable_max && mode != MODE_GLOBAL && mode != MODE_HYBRID && mode != MODE_PACK &&
                           p.blob.size() + tbytes + 64 + 16u * 64u * 64u <= 160u * 1024u && p.blob.size() + tbytes + 64 <= lds_table_budget;
        if (tbytes <= 2048 || roomy) p.hdr.off_btable = append(p.blob, bp.blob.data() + bp.hdr.off_table, tbytes);
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
        Program pk;
        pk.hdr.mode = MODE_GLOBAL;
        if (t.dfa[W_BACKWARDS].n_states + 1 <= 5) pk = lower(t, W_BACKWARDS, char_width, 64u << 10, false, false, false);
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
        bool use_flat = false; // UTF-16 rows: the flat page map (below)
        auto build = [&](Mode m) {
            const uint32_t elem = (m == MODE_TABLE16) ? 2u : 1u;
            p.blob.clear();
            p.hdr.flat_pages = 0;
            clear_window_header();
            if (win.ok && m != MODE_GLOBAL) {
                // window addressing: no column maps; the table sits win_lo_e bytes above where the kernels' fixed offsets point
                set_window_header(elem);
                const std::vector<uint16_t> tw = window_table(next, n_dev);
                if (char_width == 1) p.blob.assign(kLdsTable1 + p.hdr.win_lo_e, 0);
                else p.blob.assign(16, 0);
                p.hdr.off_table = (uint32_t)p.blob.size() - (char_width == 1 ? p.hdr.win_lo_e : 0u);
                if (m == MODE_TABLE8) {
                    for (uint16_t v : tw) p.blob.push_back((uint8_t)v);
                } else {
                    const uint8_t *b = (const uint8_t *)tw.data();
                    p.blob.insert(p.blob.end(), b, b + tw.size() * 2);
                }
                if (char_width == 2) p.hdr.off_table = 16; // (the kernels subtract win_lo_e themselves: 32-bit address math)
                emit_backward_maps();
                p.hdr.lds_bytes = (uint32_t)p.blob.size();
                return;
            }
            if (char_width == 1) {
                p.blob.assign(512, 0); // cmap16 at kLdsCmap1 = 0
                for (int c = 0; c < 256; ++c) put16(kLdsCmap1 + 2 * c, cm.cmap8[c] * elem);
            } else if (use_flat && m != MODE_GLOBAL) {
                // flat page map: a page per high byte, ptab[hi] = hi * 256 -- the scan / find-all kernels then read a char's column
                // at pages[char] in ONE lookup (needle_walk.h piece_lookups); every other reader still goes through ptab
                p.blob.assign(kLdsPages2Table + 65536, 0);
                for (int hi = 0; hi < 256; ++hi) {
                    put16(kLdsPtab2 + 2 * hi, (uint32_t)hi * 256u);
                    for (int lo = 0; lo < 256; ++lo)
                        p.blob[kLdsPages2Table + (size_t)hi * 256 + lo] = (uint8_t)(cm.pages[(size_t)cm.ptab[hi] * 256 + lo] * elem);
                }
                p.hdr.flat_pages = 1;
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
        if (char_width == 2 && (uint32_t)n_cols * elem > 255u && !win.ok) mode = MODE_GLOBAL; // pages hold column * elem in a byte
        // UTF-16 rows, plain LDS tables: the FLAT page map (64 KB) when the program then still leaves room for 16 waves x 64-byte tiles
        // -- one column lookup per char instead of two dependent ones (C5w: the kernel is LDS-bound, 3.4 LDS instructions per char).
        // NEEDLE_FLAT_MAP=0: the compact two-level map always (A/B, tests).
        static const bool flat_on = !(getenv("NEEDLE_FLAT_MAP") && atoi(getenv("NEEDLE_FLAT_MAP")) == 0);
        if (flat_on && char_width == 2 && !win.ok && (mode == MODE_TABLE8 || mode == MODE_TABLE16) && cm.pages.size() < 65536) {
            use_flat = true;
            build(mode);
            if (p.blob.size() + 16u * 64u * 64u > 160u * 1024u || p.blob.size() > lds_table_budget) {
                use_flat = false;
                p.hdr.flat_pages = 0;
            }
        }
        if (!use_flat) build(mode);
        bool sparse_done = false;
        if (mode != MODE_GLOBAL && p.blob.size() > lds_table_budget && !no_pair) { // (ml: the scan kernels' lengths form too)
            // Too big for a dense table in LDS.  First choice: the compressed whole-automaton form (MODE_SPARSE, above).
            // NEEDLE_SPARSE=0 turns it off (A/B, tests of the hot-rows mode); NEEDLE_SPARSE_ROOM: LDS bytes the program may take
            // (default: what leaves room for 16 waves x 64-byte tiles).
            static const bool sparse_on = !(getenv("NEEDLE_SPARSE") && atoi(getenv("NEEDLE_SPARSE")) == 0);
            static const size_t sparse_room = getenv("NEEDLE_SPARSE_ROOM") ? (size_t)atol(getenv("NEEDLE_SPARSE_ROOM")) : (size_t)(96u << 10);
            // columns in use: the reference classes (and OVER) that some code unit of this width maps to, renumbered densely
            std::vector<int> cols, col_id(n_cols, -1);
            {
                std::vector<uint8_t> used(n_cols, 0);
                if (char_width == 1) for (int c = 0; c < 256; ++c) used[cm.cmap8[c]] = 1;
                else for (uint8_t c : cm.pages) used[c] = 1;
                for (int k = 0; k < n_cols; ++k)
                    if (used[k]) { col_id[k] = (int)cols.size(); cols.push_back(k); }
            }
            // lengths programs: the row's end is a column of its own (END = the PAD column, which lower() pointed at the D_L of every
            // state's pending length): the walk takes it once, after the row's last char, and lands in the state that names the length
            std::vector<uint16_t> end_tgt; // lengths programs: where the row's END leads from each state (the PAD column lower() patched)
            if (ml) {
                end_tgt.resize(n_dev);
                for (int st = 0; st < n_dev; ++st) end_tgt[st] = next[(size_t)st * n_cols + PAD];
            }
            const int NC = (int)cols.size();
            const bool cols_ok = char_width == 1 || NC * 4 <= 256 || win.ok; // UTF-16: the pages hold column * 4 in a byte
            if (sparse_on && cols_ok) {
                // containedIn: every accepting state is absorbing -- one state as far as the walk is concerned
                std::vector<uint16_t> canon;
                const std::vector<uint16_t> *tab = &next;
                if (contained && accept_lo < n_dev) {
                    canon = next;
                    for (auto &v : canon)
                        if ((int)v >= accept_lo) v = (uint16_t)accept_lo;
                    for (int k = 0; k < n_cols; ++k) canon[(size_t)accept_lo * n_cols + k] = (uint16_t)accept_lo;
                    tab = &canon;
                }
                const size_t room = std::min(lds_table_budget, sparse_room);
                SparseImage im;
                bool built = false;
                for (int attempt = win.ok ? 0 : 1; attempt < 2 && !built; ++attempt) { // window addressing first, column maps else
                    const bool w = attempt == 0;
                    p.blob.clear();
                    clear_window_header();
                    p.hdr.off_bcmap = p.hdr.off_bptab = p.hdr.off_bpages = p.hdr.off_btable = p.hdr.off_bpack = 0;
                    if (w) {
                        p.blob.assign(char_width == 1 ? (size_t)kLdsTable1 : 16, 0);
                    } else if (char_width == 1) {
                        p.blob.assign(kLdsTable1, 0);
                        for (int c = 0; c < 256; ++c) put16(kLdsCmap1 + 2 * c, (uint32_t)col_id[cm.cmap8[c]] * 4u);
                    } else {
                        if (NC * 4 > 256) break;
                        p.blob.assign(kLdsPages2Table + cm.pages.size(), 0);
                        for (int hi = 0; hi < 256; ++hi) put16(kLdsPtab2 + 2 * hi, (uint32_t)cm.ptab[hi] * 256u);
                        for (size_t i = 0; i < cm.pages.size(); ++i) p.blob[kLdsPages2Table + i] = (uint8_t)(col_id[cm.pages[i]] * 4);
                        while (p.blob.size() % 16) p.blob.push_back(0);
                    }
                    const size_t fixed = p.blob.size() + (with_backward_maps ? (char_width == 1 ? 256 : 256 + bm.pages.size()) + 64 : 0) + 64 +
                                         (ml ? (size_t)ml->n_dead * (size_t)NC + 48 : 0); // (+ pend[] of the lengths form)
                    if (room <= fixed) continue;
                    const int nd = ml ? ml->n_dead : 0;
                    built = w ? build_sparse(*tab, n_dev, n_cols, win.cols, win.cl, dev[0], accept_lo, room - fixed, im, nd, ml ? &end_tgt : nullptr)
                              : build_sparse(*tab, n_dev, n_cols, cols, 0, dev[0], accept_lo, room - fixed, im, nd, ml ? &end_tgt : nullptr);
                    if (built && ml) {
                        p.hdr.sp_end_col4 = im.end_key;
                        p.hdr.sp_len_cols = (uint32_t)(w ? win.cols.size() : cols.size());
                    }
                    if (built && w) {
                        set_window_header(4);
                        p.hdr.n_cols = n_cols; // (PAD / PRE are no columns in this mode; n_cols stays the reference's)
                        p.hdr.pad_col = PAD;
                    }
                }
                if (built) {
                    p.hdr.off_table = (uint32_t)p.blob.size();
                    p.blob.insert(p.blob.end(), im.img.begin(), im.img.end());
                    mode = MODE_SPARSE;
                    emit_backward_maps();
                    while (p.blob.size() % 16) p.blob.push_back(0);
                    p.hdr.lds_bytes = (uint32_t)p.blob.size();
                    p.hdr.start = im.start;
                    p.hdr.accept_lo = im.accept_lo;
                    p.hdr.sp_rec_base = im.rec_base;
                    p.hdr.sp_accept_rec = im.accept_rec;
                    p.hdr.sp_rows_base = im.rows_base;
                    p.hdr.sp_chains = im.chains;
                    p.hdr.sp_pad_ident = (which == W_MATCHES || contained || ml) ? 1u : 0u; // (lengths programs: the state freezes at the
                                                                                                // row's end; finish_rows asks its END record)
                    p.hdr.sp_dead_row0 = im.dead_row0;
                    p.hdr.fa_dead_hi = im.dead_hi;
                    p.hdr.sp_dense = im.dense;
                    p.hdr.sp_records = im.records;
                    sparse_done = true;
                } else {
                    build(mode); // (the dense layout again, for the fallbacks below)
                }
            }
        }
        if (!sparse_done && mode != MODE_GLOBAL && p.blob.size() > lds_table_budget) {
            mode = MODE_GLOBAL;
            build(mode);
            // Hot rows in LDS + the whole table in HBM.  Search automata on real text sit in the few states near their
            // start state (a 1000-keyword union over random text: 99.8 % of the steps are in states of depth <= 3), so
            // the rows of the first states in breadth-first order are the ones worth the LDS.  The hot prefix is sized
            // to leave room for 16 waves x 64-byte tiles.  NEEDLE_HYBRID=0: plain HBM table (tests, A/B).
            static const bool hybrid_on = !(getenv("NEEDLE_HYBRID") && atoi(getenv("NEEDLE_HYBRID")) == 0);
            const bool hw = win.ok; // window addressing: rows of W + 2 cells, no column maps
            const size_t row_bytes = (size_t)(hw ? win.W + 2 : n_cols) * 2;
            const bool cols_ok = char_width == 1 || row_bytes <= 255 || hw;
            const size_t room = std::min<size_t>(lds_table_budget, 96u << 10);
            // column maps (+ backward maps) as just built for the HBM-table layout
            const size_t fixed = p.hdr.lds_bytes + 64 + (hw ? (size_t)win.cl * 2 : 0);
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
                clear_window_header();
                p.hdr.off_bcmap = p.hdr.off_bptab = p.hdr.off_bpages = p.hdr.off_btable = p.hdr.off_bpack = 0;
                if (hw) { // the table in window layout, physically win_lo_e bytes above the offsets the kernels use
                    set_window_header(2);
                    nh = window_table(nh, n_dev);
                    p.blob.assign(char_width == 1 ? (size_t)kLdsTable1 + p.hdr.win_lo_e : 16, 0);
                    p.hdr.hot_bytes = (uint32_t)(hot_rows * row_bytes);
                    p.hdr.off_table = char_width == 1 ? (uint32_t)kLdsTable1 : 16u;
                    const uint8_t *b = (const uint8_t *)nh.data();
                    p.blob.insert(p.blob.end(), b, b + p.hdr.hot_bytes);
                } else {
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
                }
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
    if (ml) {
        // pend[] by DEVICE state rides in the LDS part; only the plain table modes number states the way it is indexed
        if (mode == MODE_SPARSE) {
            // the compressed form: only the D_L states are ever asked (a live stop state takes the END column first) -- pend[] is
            // indexed by their rows' address fields, sp_len_cols apart from sp_dead_row0 on
            std::vector<uint8_t> pend_rows((size_t)ml->n_dead * p.hdr.sp_len_cols + 16, 0);
            for (int k = 1; k <= ml->n_dead; ++k) pend_rows[(size_t)(k - 1) * p.hdr.sp_len_cols] = ml->pend[k];
            p.hdr.fa_len_off = (uint32_t)p.blob.size();
            p.blob.insert(p.blob.end(), pend_rows.begin(), pend_rows.end());
            while (p.blob.size() % 16) p.blob.push_back(0);
            p.hdr.lds_bytes = (uint32_t)p.blob.size();
            p.hdr.fa_dead_lo = 1;
            p.hdr.fa_dead_n = (uint32_t)ml->n_dead;
            return p;
        }
        if (mode == MODE_GLOBAL && aux && aux->ml_in_hbm) {
            // the filter kernel's verify walk out of HBM / L2 (lower_filter_hbm): plain device numbering, pend[] by device state BEHIND the
            // table -- read from memory once per verified candidate, not staged in LDS
            std::vector<uint8_t> pend_dev(n_dev, 0);
            for (int s = 0; s < n_ref; ++s) pend_dev[dev[s]] = ml->pend[s];
            p.hdr.fa_len_off = append(p.blob, pend_dev.data(), pend_dev.size());
            while (p.blob.size() % 16) p.blob.push_back(0);
            p.hdr.fa_dead_hi = (uint32_t)ml->n_dead;
            p.hdr.fa_dead_lo = 1;
            p.hdr.fa_dead_n = (uint32_t)ml->n_dead;
            return p;
        }
        if (mode != MODE_TABLE8 && mode != MODE_TABLE16 && mode != MODE_PAIR) {
            p.blob.clear();
            p.hdr.mode = MODE_GLOBAL;
            return p;
        }
        p.hdr.fa_dead_hi = (uint32_t)ml->n_dead;
        std::vector<uint8_t> pend_dev(n_dev, 0);
        for (int s = 0; s < n_ref; ++s) pend_dev[dev[s]] = ml->pend[s];
        p.hdr.fa_len_off = (uint32_t)p.blob.size();
        p.blob.insert(p.blob.end(), pend_dev.begin(), pend_dev.end());
        while (p.blob.size() % 16) p.blob.push_back(0);
        p.hdr.lds_bytes = (uint32_t)p.blob.size();
        p.hdr.fa_dead_lo = 1;
        p.hdr.fa_dead_n = (uint32_t)ml->n_dead;
        p.hdr.fa_skip_lo = n_skip ? (uint32_t)skip_lo : 0u;
    }
    return p;
}

} // namespace needle

// ---- per-state match lengths for find-all (needle_lower.h) -----------------------------------------------------------
namespace needle {

MatchLengths match_length_automaton(const RefTables &t) {
    MatchLengths out;
    const RefDfa &F = t.dfa[W_FORWARDS], &B = t.dfa[W_BACKWARDS];
    const int N = t.stride;
    if (F.n_states < 1 || B.n_states < 1 || F.accepting[0] || B.accepting[0] || t.dfa[W_MATCHES].accepting[0]) return out; // (empty matches: the immediate form)
    static const bool dbg = getenv("NEEDLE_ML_DEBUG") != nullptr;
    // the alphabet: what the two automata can tell apart -- (class, beyond F's maxChar, beyond B's maxChar) of some code unit
    std::vector<std::array<int, 3>> syms;
    {
        std::map<std::array<int, 3>, int> seen;
        for (int c = 0; c < 65536; ++c) {
            const std::array<int, 3> k = {(int)t.class_map[c], c > F.max_char ? 1 : 0, c > B.max_char ? 1 : 0};
            if (seen.emplace(k, 0).second) syms.push_back(k);
        }
    }
    const int S = (int)syms.size();
    auto stepF = [&](int f, const std::array<int, 3> &y) { return y[1] ? -1 : (int)F.table[(size_t)f * N + y[0]]; };
    auto stepB = [&](int b, const std::array<int, 3> &y) { return y[2] ? -1 : (int)B.table[(size_t)b * N + y[0]]; };
    // 1. What would indexBackwards report if the search ended HERE?  The reversed automaton B reads the text right to left
    // from the end, remembering the leftmost index at which it was accepting, until it dies (DFAClassBuilder.java:549-583; NOT
    // simply the longest match: B is built with the reference's priority pruning, so `bc|abc` reports "bc" inside "abc").  Scanning
    // left to right, keep for EVERY state q of B the answer "B started in q at the current end: the largest number of chars
    // after which it was accepting" -- V[q] -- because one more char c turns it into V'[q] = V[B(q, c)] + 1 (or 1 when
    // B(q, c) accepts and nothing longer does).  The product of the search automaton with those vectors (sparse: few states of
    // B lead anywhere on a given text) is finite when match lengths are bounded; its accepting states report V[B's start].
    std::vector<std::vector<std::vector<int>>> pred(S, std::vector<std::vector<int>>(B.n_states)); // pred[y][q1]: q with B(q, y) = q1
    for (int y = 0; y < S; ++y)
        for (int q = 0; q < B.n_states; ++q) {
            const int q1 = stepB(q, syms[y]);
            if (q1 >= 0) pred[y][q1].push_back(q);
        }
    std::vector<int> b_acc;
    for (int q = 0; q < B.n_states; ++q)
        if (B.accepting[q]) b_acc.push_back(q);
    typedef std::vector<uint32_t> Vec; // sorted (q << 8 | length)
    std::map<std::pair<int, Vec>, int> index;
    std::vector<std::pair<int, Vec>> prod;
    std::vector<int> trans; // prod x S
    std::vector<int> outL;  // match length of an accepting product state, 0 otherwise
    const size_t kMaxProd = 60000;
    uint64_t work = 0;
    auto intern = [&](int f, Vec &&r) -> int {
        auto key = std::make_pair(f, std::move(r));
        auto it = index.find(key);
        if (it != index.end()) return it->second;
        const int id = (int)prod.size();
        index.emplace(key, id);
        int L = 0;
        if (F.accepting[f]) {
            // V[B's start state 0]: entries are sorted by q, so it is the first one if present
            if (!key.second.empty() && (key.second[0] >> 8) == 0u) L = (int)(key.second[0] & 255u);
            if (L == 0) L = -1; // the search automaton accepts but the reversed one finds no start: the two disagree -- give up
        }
        outL.push_back(L);
        prod.push_back(std::move(key));
        return id;
    };
    intern(0, Vec());
    std::vector<int> scratch(B.n_states, 0);
    for (size_t i = 0; i < prod.size(); ++i) {
        if (prod.size() > kMaxProd || work > 400ull * 1000 * 1000) { if (dbg) fprintf(stderr, "[ml] product too big\n"); return out; }
        if (outL[i] < 0) { if (dbg) fprintf(stderr, "[ml] F accepts (state %d) but the reversed automaton finds no start\n", prod[i].first); return out; }
        trans.resize((i + 1) * S, -1);
        for (int y = 0; y < S; ++y) {
            const int f = prod[i].first;
            const int f2 = stepF(f, syms[y]);
            if (f2 < 0) continue;
            Vec r;
            bool too_long = false;
            // scratch[q] = V'[q] (0 = undefined)
            std::vector<int> touched;
            for (int q1 : b_acc)
                for (int q : pred[y][q1]) {
                    if (!scratch[q]) touched.push_back(q);
                    scratch[q] = 1;
                }
            for (uint32_t x : prod[i].second) {
                const int q1 = (int)(x >> 8), len = (int)(x & 255u) + 1;
                if (len > 250) too_long = true;
                for (int q : pred[y][q1]) {
                    if (!scratch[q]) touched.push_back(q);
                    scratch[q] = len; // (longer than the 1 an accepting B(q, c) alone gives)
                }
                work += pred[y][q1].size() + 1;
            }
            std::sort(touched.begin(), touched.end());
            r.reserve(touched.size());
            for (int q : touched) {
                r.push_back(((uint32_t)q << 8) | (uint32_t)scratch[q]);
                scratch[q] = 0;
            }
            work += touched.size();
            if (too_long || r.size() > 8192) { if (dbg) fprintf(stderr, "[ml] unbounded match lengths\n"); return out; } // `[0-9]+`, `a.*b`: keep indexBackwards
            const int id = intern(f2, std::move(r)); // (may reallocate prod: indices only from here on)
            trans[i * S + y] = id;
        }
    }
    const int P = (int)prod.size();
    // 2. Moore minimisation with the match length as output: the smallest automaton whose accepting states each stand for
    // ONE length (1000 keywords of 3..5 chars: 1401 states -> 1463)
    std::vector<int> part(P), nxt_part(P);
    {
        std::map<int, int> cls;
        for (int i = 0; i < P; ++i) part[i] = cls.emplace(outL[i], (int)cls.size()).first->second;
    }
    for (size_t n_cls = 0;;) {
        std::map<std::vector<int>, int> sig;
        std::vector<int> key(S + 1);
        for (int i = 0; i < P; ++i) {
            key[0] = part[i];
            for (int y = 0; y < S; ++y) key[y + 1] = trans[(size_t)i * S + y] < 0 ? -1 : part[trans[(size_t)i * S + y]];
            nxt_part[i] = sig.emplace(key, (int)sig.size()).first->second;
        }
        part.swap(nxt_part);
        if (sig.size() == n_cls) break;
        n_cls = sig.size();
    }
    int Q = 0;
    for (int i = 0; i < P; ++i) Q = std::max(Q, part[i] + 1);
    std::vector<int> rep(Q, -1);
    for (int i = 0; i < P; ++i)
        if (rep[part[i]] < 0) rep[part[i]] = i;
    // 3. remember the last match until the automaton dies: states (q, pending length); deaths with a match pending lead
    // to D_L.  (Keyword unions: no growth at all -- their accepting states die on every char.)
    std::vector<int> lens; // distinct match lengths, ascending
    for (int q = 0; q < Q; ++q)
        if (outL[rep[q]] > 0) lens.push_back(outL[rep[q]]);
    std::sort(lens.begin(), lens.end());
    lens.erase(std::unique(lens.begin(), lens.end()), lens.end());
    if (lens.empty() || lens.size() > 64) { if (getenv("NEEDLE_ML_DEBUG")) fprintf(stderr, "[ml] lens %zu\n", lens.size()); return out; }
    const int K = (int)lens.size();
    auto dead_of = [&](int L) { return 1 + (int)(std::lower_bound(lens.begin(), lens.end(), L) - lens.begin()); };
    std::map<std::pair<int, int>, int> pidx;
    std::vector<std::pair<int, int>> pst; // ref states 1 + K ..: (q, pending)
    auto pin = [&](int q, int pend) -> int {
        auto it = pidx.find({q, pend});
        if (it != pidx.end()) return it->second;
        const int id = (int)pst.size();
        pidx.emplace(std::make_pair(q, pend), id);
        pst.push_back({q, pend});
        return id;
    };
    // ref numbering: 0 = start, 1 .. K = D_L, then the other (q, pending) states in discovery order
    auto ref_of = [&](int pid) { return pid == 0 ? 0 : pid + K; };
    pin(part[0], 0);
    std::vector<int16_t> table;
    for (size_t i = 0; i < pst.size(); ++i) {
        if (pst.size() + K > 16383) return out; // DFACompiler.checkForOverLongDFAs, DFACompiler.java:76-83
        const int q = pst[i].first, pend = pst[i].second;
        // (a class with chars on both sides of a maxChar is two symbols here but ONE table column: representable only if
        // both symbols lead to the same place -- checked below)
        std::vector<int> row(N + 1, -2); // the reference classes, then OVER: every char beyond EITHER automaton's maxChar
        for (int y = 0; y < S; ++y) {
            const int tq = trans[(size_t)rep[q] * S + y];
            int tgt;
            if (tq < 0) tgt = pend ? dead_of(pend) : -1;
            else {
                const int q2 = part[tq];
                const int L2 = outL[rep[q2]];
                tgt = ref_of(pin(q2, L2 > 0 ? L2 : pend));
            }
            int &cell = row[(syms[y][1] || syms[y][2]) ? N : syms[y][0]];
            if (cell != -2 && cell != tgt) { if (getenv("NEEDLE_ML_DEBUG")) fprintf(stderr, "[ml] column conflict: state (%d,%d) class %d over %d/%d: %d vs %d\n", q, pend, syms[y][0], syms[y][1], syms[y][2], cell, tgt); return out; }
            cell = tgt;
        }
        for (int c = 0; c <= N; ++c) table.push_back((int16_t)(row[c] == -2 ? (pend ? dead_of(pend) : -1) : row[c]));
    }
    const int n_ref = 1 + K + (int)pst.size() - 1;
    out.dfa.n_states = n_ref;
    out.dfa.max_char = std::min(F.max_char, B.max_char); // chars beyond it take the OVER column: out.over[]
    out.over.assign(n_ref, (int16_t)-1);
    out.dfa.table.assign((size_t)n_ref * N, (int16_t)-1);
    out.dfa.accepting.assign(n_ref, 0);
    out.pend.assign(n_ref, 0);
    for (int k = 0; k < K; ++k) { // D_L: absorbing, not accepting
        for (int c = 0; c < N; ++c) out.dfa.table[(size_t)(1 + k) * N + c] = (int16_t)(1 + k);
        out.over[1 + k] = (int16_t)(1 + k);
        out.pend[1 + k] = (uint8_t)lens[k];
    }
    for (size_t i = 0; i < pst.size(); ++i) {
        const int r = ref_of((int)i);
        for (int c = 0; c < N; ++c) out.dfa.table[(size_t)r * N + c] = table[i * (N + 1) + c];
        out.over[r] = table[i * (N + 1) + N];
        const int L = outL[rep[pst[i].first]];
        out.dfa.accepting[r] = L > 0 ? 1 : 0;
        out.pend[r] = (uint8_t)(L > 0 ? L : pst[i].second);
    }
    out.n_dead = K;
    out.ok = true;
    return out;
}

Program lower_match_lengths(const RefTables &t, const MatchLengths &ml, int char_width, size_t lds_table_budget, bool plain) {
    // plain: the find-all kernel's form (no pair table, no window addressing, no compressed automaton); else the scan kernel's
    return lower(t, W_FORWARDS, char_width, lds_table_budget, false, false, plain, &ml);
}


// ---- the find-all transducer (needle_lower.h) -------------------------------------------------------------------------------------
// The device image of a find-all transducer (both kinds below): tab[state][NC] = target << 4 | code in reference columns (classes, OVER,
// PAD) -> window layout or column maps + uint16 table + codes[16], MODE_TABLE16; empty blob when it does not fit.
static Program emit_transducer(Program p, const RefTables &t, const RefDfa &d, std::vector<uint16_t> &tab, const int n_t, const int NC, const int PAD,
                               const std::map<std::pair<int, int>, int> &codes, int char_width, size_t lds_table_budget, uint32_t ft_on) {
    // entries were written with ids that may exceed what existed when a row was made -- all ids are final now; nothing to patch
    const ColumnMaps cm = column_maps(t, d, char_width);
    Window win;
    {
        static const bool window_on = !(getenv("NEEDLE_WINDOW") && atoi(getenv("NEEDLE_WINDOW")) == 0);
        if (window_on) {
            auto same = [&](int a, int b) {
                if (a == b) return true;
                for (int st = 0; st < n_t; ++st)
                    if (tab[(size_t)st * NC + a] != tab[(size_t)st * NC + b]) return false;
                return true;
            };
            win = find_window(cm, char_width, same);
            const size_t bytes_w = (size_t)n_t * (win.W + 1) * 2, bytes_c = (size_t)n_t * NC * 2;
            if (win.ok && !(bytes_w <= bytes_c * 5 / 4 || bytes_w <= (16u << 10))) win.ok = false;
        }
    }
    auto put16 = [&](size_t off, uint32_t v) { p.blob[off] = (uint8_t)(v & 255); p.blob[off + 1] = (uint8_t)(v >> 8); };
    std::vector<uint16_t> out;
    if (win.ok) {
        const int nw = win.W + 1; // the window's chars, then PAD ("char ch + 1": column offsets are not rebased)
        out.resize((size_t)n_t * nw);
        for (int st = 0; st < n_t; ++st) {
            for (int j = 0; j < win.W; ++j) out[(size_t)st * nw + j] = tab[(size_t)st * NC + win.cols[j]];
            out[(size_t)st * nw + win.W] = tab[(size_t)st * NC + PAD];
        }
        p.hdr.win_on = 1;
        p.hdr.win_lo_e = (uint32_t)win.cl * 2u;
        p.hdr.win_hi_e = (uint32_t)win.ch * 2u;
        p.hdr.n_cols = (uint32_t)nw;
        p.hdr.pad_col = (uint32_t)(win.cl + win.W);
        if (char_width == 1) {
            p.blob.assign(kLdsTable1 + p.hdr.win_lo_e, 0);
            p.hdr.off_table = kLdsTable1;
        } else {
            p.blob.assign(16, 0);
            p.hdr.off_table = 16;
        }
    } else {
        out = tab;
        p.hdr.n_cols = (uint32_t)NC;
        p.hdr.pad_col = (uint32_t)PAD;
        if (char_width == 1) {
            p.blob.assign(512, 0); // cmap16 at kLdsCmap1 = 0
            for (int c = 0; c < 256; ++c) put16(kLdsCmap1 + 2 * c, cm.cmap8[c] * 2u);
            p.hdr.off_table = kLdsTable1;
        } else {
            if ((uint32_t)NC * 2u > 255u) return p; // (the pages hold column * 2 in a byte)
            p.blob.assign(kLdsPages2Table + cm.pages.size(), 0);
            for (int hi = 0; hi < 256; ++hi) put16(kLdsPtab2 + 2 * hi, (uint32_t)cm.ptab[hi] * 256u);
            for (size_t i = 0; i < cm.pages.size(); ++i) p.blob[kLdsPages2Table + i] = (uint8_t)(cm.pages[i] * 2u);
            while (p.blob.size() % 16) p.blob.push_back(0);
            p.hdr.off_table = (uint32_t)p.blob.size();
        }
    }
    p.hdr.n_pages = (uint32_t)(cm.pages.size() / 256);
    {
        const uint8_t *b = (const uint8_t *)out.data();
        p.blob.insert(p.blob.end(), b, b + out.size() * 2);
    }
    // codes[code] = (k + length) | k << 16: a match filed at char index pos is [pos - (k + length), pos - k) -- in the one-dword form
    // pos * 0x10001 - codes[code].  With at most 8 codes they are numbered 1, 3, .. 15: bit 0 of a log nibble then says "a match
    // ends here" and the kernel finds its next one with one and + one find-first-bit (ft_odd).
    // ft_direct: every code has k = 0 and a length below 16 (keyword unions whose accepting states die on every char): the code IS
    // the length -- the kernel files a match without a table lookup.
    bool direct = true, direct_odd = true;
    for (const auto &kv : codes) {
        direct = direct && kv.first.second == 0 && kv.first.first >= 1 && kv.first.first <= 15;
        direct_odd = direct_odd && kv.first.second == 0 && kv.first.first >= 1 && kv.first.first <= 7;
    }
    // (lengths up to 7: the code is length << 1 | 1 -- bit 0 of a log nibble says "a match ends here", as for ft_odd, AND the length needs
    // no lookup: ft_direct = 2)
    const bool odd = !direct && codes.size() <= 8;
    uint32_t ct[16] = {0};
    auto renum = [&](int c, int L) { return direct_odd ? 2 * L + 1 : direct ? L : odd ? 2 * c - 1 : c; };
    std::vector<int> code_to(16, 0);
    for (const auto &kv : codes) {
        code_to[kv.second] = renum(kv.second, kv.first.first);
        ct[code_to[kv.second]] = (uint32_t)(kv.first.first + kv.first.second) | (uint32_t)kv.first.second << 16;
    }
    if (ft_on == 1u) {
        uint16_t *cells = (uint16_t *)(p.blob.data() + (p.blob.size() - out.size() * 2));
        for (size_t i = 0; i < out.size(); ++i)
            if (cells[i] & 15u) cells[i] = (uint16_t)((cells[i] & ~15u) | (uint32_t)code_to[cells[i] & 15u]);
    }
    // (the run transducer's codes are event bits -- 1: a match ends, 2: a run may start -- and stay as written)
    p.hdr.ft_odd = ft_on == 1u && (odd || direct_odd) ? 1u : 0u;
    p.hdr.ft_direct = ft_on != 1u ? 0u : direct_odd ? 2u : direct ? 1u : 0u;
    p.hdr.ft_codes_off = append(p.blob, ct, sizeof(ct));
    while (p.blob.size() % 16) p.blob.push_back(0);
    if (p.blob.size() > lds_table_budget || p.blob.size() + 4u * 64u * 64u > 160u * 1024u) { // (no room beside even 4 waves of tiles)
        p.blob.clear();
        return p;
    }
    p.hdr.mode = MODE_TABLE16;
    p.hdr.ft_on = ft_on;
    p.hdr.n_states = (uint32_t)n_t;
    p.hdr.start = 1;
    p.hdr.accept_lo = (uint32_t)n_t; // (no accepting states as far as any other reader is concerned)
    p.hdr.lds_bytes = (uint32_t)p.blob.size();
    return p;
}
Program lower_find_all_transducer(const RefTables &t, const MatchLengths &ml, int char_width, size_t lds_table_budget) {
    Program p;
    memset(&p.hdr, 0, sizeof(p.hdr));
    memset(&p.ng.p, 0, sizeof(p.ng.p));
    p.hdr.mode = MODE_GLOBAL; // (= not available, with the empty blob)
    if (!ml.ok) return p;
    static const bool dbg = getenv("NEEDLE_ML_DEBUG") != nullptr;
    const RefDfa &d = ml.dfa;
    const int N = t.stride, OVER = N, PAD = N + 1, NC = N + 2, K = ml.n_dead;
    auto step = [&](int m, int c) -> int { return c == OVER ? (int)ml.over[m] : (int)d.table[(size_t)m * N + c]; };
    auto is_dead = [&](int m) { return m >= 1 && m <= K; };
    // states: {m, s, k}; s = -2: no match pending (a plain state of the lengths automaton); s = -1: the shadow has died
    std::map<std::array<int, 3>, int> ids;
    std::vector<std::array<int, 3>> keys(1, std::array<int, 3>{-1, -1, -1}); // 0 = dead
    auto tid = [&](int m, int s, int k) -> int {
        const std::array<int, 3> key = {m, s, k};
        auto it = ids.find(key);
        if (it != ids.end()) return it->second;
        const int id = (int)keys.size();
        ids.emplace(key, id);
        keys.push_back(key);
        return id;
    };
    std::map<std::pair<int, int>, int> codes; // (length, k) -> 1 .. 15
    bool ok = true;
    auto code_of = [&](int L, int k) -> int {
        auto it = codes.find({L, k});
        if (it != codes.end()) return it->second;
        const int c = (int)codes.size() + 1;
        if (c > 15 || L > 255 || k > 255) ok = false;
        codes.emplace(std::make_pair(L, k), c);
        return c;
    };
    // after a transition that leaves plain state m2 of the lengths automaton current: dead, accepting (a match pending from here on:
    // its shadow starts in the start state) or plain
    auto enter = [&](int m2) -> int { return m2 < 0 ? 0 : d.accepting[m2] ? tid(m2, 0, 0) : tid(m2, -2, 0); };
    std::vector<uint16_t> tab; // [state][NC]: target << 4 | code
    tab.assign(NC, 0);         // the dead state's row
    if (d.accepting[0] || ml.pend[0]) return p;
    tid(0, -2, 0); // the start state: id 1
    for (size_t i = 1; i < keys.size() && ok; ++i) {
        if (keys.size() > 4095) { ok = false; break; }
        const int m = keys[i][0], s = keys[i][1], k = keys[i][2];
        tab.resize((i + 1) * NC, 0);
        for (int c = 0; c < NC; ++c) {
            uint32_t tgt = 0, code = 0;
            if (s == -2) { // nothing pending
                if (c != PAD) {
                    const int m2 = step(m, c);
                    if (m2 >= 0 && is_dead(m2)) { ok = false; break; } // (cannot be: nothing is pending)
                    tgt = (uint32_t)enter(m2);
                }
            } else if (c == PAD) { // the row ends with a match pending
                code = (uint32_t)code_of(ml.pend[m], k);
            } else {
                const int m2 = step(m, c), s2 = s >= 0 ? step(s, c) : -1;
                if (m2 < 0) { ok = false; break; } // (cannot be: a state with a match pending dies into its D_L)
                if (d.accepting[m2]) {
                    tgt = (uint32_t)tid(m2, 0, 0); // a later accept of the same search: the shadow starts over
                } else if (is_dead(m2)) {
                    code = (uint32_t)code_of(ml.pend[m], k);
                    if (s2 >= 0 && is_dead(s2)) { ok = false; break; } // (cannot be: the shadow has not accepted)
                    tgt = (uint32_t)enter(s2); // the restarted search, already past the chars since the match's end
                } else {
                    if (s2 >= 0 && d.accepting[s2]) { // the restarted search would accept while this one still lives: a shadow of a
                        if (dbg) fprintf(stderr, "[ft] shadow accepts under a live match: state (%d,%d,%d) column %d\n", m, s, k, c); // shadow
                        ok = false;
                        break;
                    }
                    if (k + 1 > 250) { ok = false; break; }
                    tgt = (uint32_t)tid(m2, s2 < 0 ? -1 : s2, k + 1);
                }
            }
            tab[i * NC + c] = (uint16_t)(tgt << 4 | code);
        }
    }
    if (!ok || keys.size() > 4095) {
        if (dbg) fprintf(stderr, "[ft] no transducer (%zu states, %zu codes)\n", keys.size(), codes.size());
        return p;
    }
    return emit_transducer(p, t, d, tab, (int)keys.size(), NC, PAD, codes, char_width, lds_table_budget, 1u);
}

// The RUN transducer (needle_lower.h): lock-step find-all for patterns whose matches have no bounded length but are "runs" -- `[0-9]+`,
// `[a-z]{3}[a-z]*`: BASELINE's C2 and C5.  Read off the forward search automaton F (indexForwards', DFAClassBuilder.java:335-471) and the
// anchored automaton M (matches(), :854-912); empty blob = the pattern is not of that kind (the one-pass kernel stays).
Program lower_find_all_runs(const RefTables &t, int char_width, size_t lds_table_budget) {
    Program p;
    memset(&p.hdr, 0, sizeof(p.hdr));
    memset(&p.ng.p, 0, sizeof(p.ng.p));
    p.hdr.mode = MODE_GLOBAL; // (= not available, with the empty blob)
    static const bool dbg = getenv("NEEDLE_ML_DEBUG") != nullptr;
    auto no = [&](const char *why) {
        if (dbg) fprintf(stderr, "[runs] no run transducer: %s\n", why);
        return p;
    };
    const RefDfa &F = t.dfa[W_FORWARDS], &M = t.dfa[W_MATCHES];
    const int N = t.stride, PAD = N + 1, NC = N + 2, nF = F.n_states;
    if (nF <= 0 || nF > 4000 || M.n_states <= 0) return no("no automaton");
    if (F.accepting[0] || M.accepting[0]) return no("the pattern matches the empty string");
    if (F.max_char != 0xFFFF) return no("the search automaton has a maxChar");
    auto stepF = [&](int q, int c) -> int { return (int)F.table[(size_t)q * N + c]; };
    auto stepM = [&](int m, int c) -> int { return (int)M.table[(size_t)m * N + c]; };
    // (the classes some char has: the table's width is rounded up -- DFAClassBuilder.java:240-253 -- and the padding columns are dead)
    std::vector<int> cls;
    {
        std::vector<uint8_t> used(N, 0);
        for (uint8_t c : t.class_map)
            if (c < N) used[c] = 1;
        for (int c = 0; c < N; ++c)
            if (used[c]) cls.push_back(c);
    }
    // ---- states before a first accept (reachable from the start state through non-accepting states) never die; POST: the accepting
    // states and what follows them -- every live successor accepts again, so a search dies on the char right behind its match (k = 0)
    std::vector<uint8_t> pre(nF, 0), post(nF, 0);
    {
        std::vector<int> q{0};
        pre[0] = 1;
        for (size_t h = 0; h < q.size(); ++h)
            for (int c : cls) {
                const int q2 = stepF(q[h], c);
                if (q2 < 0) return no("the search can die before a first match");
                if (F.accepting[q2]) { post[q2] = 1; continue; }
                if (!pre[q2]) pre[q2] = 1, q.push_back(q2);
            }
        std::vector<int> w;
        for (int s = 0; s < nF; ++s)
            if (post[s]) w.push_back(s);
        for (size_t h = 0; h < w.size(); ++h)
            for (int c : cls) {
                const int q2 = stepF(w[h], c);
                if (q2 < 0) continue;
                if (!F.accepting[q2]) return no("a match can stay pending over chars that do not extend it");
                if (!post[q2]) post[q2] = 1, w.push_back(q2);
            }
        for (int s = 0; s < nF; ++s)
            if (pre[s] && post[s]) return no("a state before and after a first accept");
    }
    // ---- a run IS one attempt.  Pairs (state of F after u, state of the anchored automaton M after u) over every non-empty string u on
    // which the attempt that began with u's first char is still alive (M alive): F must not stand in its start state there -- an attempt
    // would be alive that the start state hides (`9*y`: the start state loops on '9', and indexBackwards, :529-586, walks back over the
    // '9's) -- F must not accept where that attempt does not (the match would belong to a later start), and where the attempt dies F must
    // be dead or back in its start state (else a later start lives on inside the run).  Then, whenever F stands in its start state no
    // attempt is alive, a run is carried by the attempt of its first char, and a match's start -- the leftmost s with text[s, end) in the
    // language -- is that char.
    {
        std::vector<uint32_t> seen((size_t)nF * (size_t)M.n_states / 32 + 1, 0u);
        std::vector<std::pair<int, int>> w;
        auto push = [&](int q, int m) {
            const size_t id = (size_t)q * (size_t)M.n_states + (size_t)m;
            if (seen[id >> 5] >> (id & 31) & 1u) return;
            seen[id >> 5] |= 1u << (id & 31);
            w.emplace_back(q, m);
        };
        w.emplace_back(0, 0); // (the empty string: not marked seen -- a return to (start, start) is the failure below)
        for (size_t h = 0; h < w.size(); ++h)
            for (int c : cls) {
                const int q2 = stepF(w[h].first, c), m2 = stepM(w[h].second, c);
                if (m2 < 0) {
                    if (q2 > 0) return no("the search lives on where the attempt of the run's first char has died (a later start)");
                    continue;
                }
                if (q2 < 0) continue; // (the search has pruned what the anchored automaton still follows: nothing is reported there)
                if (q2 == 0) return no("an attempt is alive while the search stands in its start state (the pattern begins with a loop)");
                if (F.accepting[q2] && !M.accepting[m2]) return no("the search accepts where the attempt of the run's first char does not");
                push(q2, m2);
                if (w.size() > (1u << 22)) return no("too many state pairs");
            }
    }
    // ---- the table: device state = F state + 1 (0: the row is over); code bit 0: a match ends in front of this char (the search died on
    // it and restarts ON it: the target is where the start state goes), bit 1: this char may begin a run (the source is the start state,
    // or the restart); PAD: a pending match ends with the row
    std::vector<uint16_t> tab((size_t)(nF + 1) * NC, 0);
    for (int q = 0; q < nF; ++q)
        for (int c = 0; c < N; ++c) {
            const int q2 = stepF(q, c);
            uint32_t tgt, code = q == 0 ? 2u : 0u;
            if (q2 >= 0) tgt = (uint32_t)q2 + 1u;
            else if (!post[q]) tgt = 0; // (unreachable: states before a first accept do not die)
            else tgt = (uint32_t)stepF(0, c) + 1u, code = 3u;
            tab[(size_t)(q + 1) * NC + c] = (uint16_t)(tgt << 4 | code);
        }
    for (int q = 0; q < nF; ++q) tab[(size_t)(q + 1) * NC + PAD] = post[q] ? 1u : 0u; // (OVER: no char is beyond maxChar 0xFFFF)
    const std::map<std::pair<int, int>, int> none;
    p = emit_transducer(p, t, F, tab, nF + 1, NC, PAD, none, char_width, lds_table_budget, 2u);
    return p;
}


} // namespace needle
