// Host analysis behind the n-gram candidate filter (needle_ngram_host.h): everything the kernel relies on is read off the
// lowered table itself -- the depth after which a restarted walk has caught up, the shortest match, the byte windows that
// can precede a first accepting transition.  Cold path (once per program).
//
// Reference: the CPU-side narrowing this generalises -- prefix `indexOf` (DFAClassBuilder.java:365-376), first-byte mask
// (:420-426, :508-511; DFA.initialAsciiBytes, DFA.java:706-726), their gate (CompilationPolicy.java:44-57 over
// Factorization.getPrefixes(), Factorization.java:116).  Those read literals off the regex AST; here the table is the source,
// so whatever DFACompiler produced (leftmost-first pruning, case folding, classes) is covered without a second semantics.
#include "needle_ngram_host.h"
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include <unordered_set>

namespace needle {

namespace {
constexpr int kN = 4;            // window length in chars (= one dword of 8-bit text)
constexpr int kMaxWarm = 14;     // K + S - 1 <= 16: the run ahead of a window fits one 16-byte load
constexpr size_t kMaxWindows = 1u << 20;
constexpr size_t kMaxFrontier = 4u << 20;

uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
} // namespace

NgramFilter build_ngram_filter(const uint16_t *next, int n_dev, int n_cols, const uint8_t *cmap8, int start, int accept_lo, int dead_hi,
                               bool absorbing, size_t prog_lds_bytes, const uint8_t *cmap16) {
    // cmap16 != nullptr: the WIDE filter -- windows of four UTF-16 code units (needle_ngram.h), columns expanded to the units of the BMP
    const bool wide = cmap16 != nullptr;
    const int n_sym = wide ? 65536 : 256;
    const uint8_t *cmap = wide ? cmap16 : cmap8;
    NgramFilter f;
    memset(&f.p, 0, sizeof(f.p));
    auto no = [&](const char *why) {
        f.why = why;
        f.p.on = 0;
        f.bitmap.clear();
        return f;
    };
    (void)absorbing; // (accepting states are only ever ENTERED here: what they do afterwards does not matter)
    if (n_dev <= 1 || start <= 0 || start >= n_dev) return no("no automaton");
    if (start >= accept_lo) return no("the start state accepts");
    if (!ngram_layout((uint32_t)prog_lds_bytes, 4096u, nullptr)) return no("no LDS left for a bitmap");
    // columns some code unit maps to, and their units
    std::vector<std::vector<uint16_t>> bytes_of(n_cols);
    for (int c = 0; c < n_sym; ++c) {
        if (cmap[c] >= n_cols) return no("column map out of range");
        bytes_of[cmap[c]].push_back((uint16_t)c);
    }
    std::vector<int> cols;
    for (int k = 0; k < n_cols; ++k)
        if (!bytes_of[k].empty()) cols.push_back(k);
    auto nx = [&](int s, int k) { return (int)next[(size_t)s * n_cols + k]; };
    auto accepting = [&](int s) { return s >= accept_lo; };

    // ---- states a walk can be in before its first accepting transition
    std::vector<uint8_t> pre(n_dev, 0);
    std::vector<int> order{start};
    pre[start] = 1;
    for (size_t h = 0; h < order.size(); ++h)
        for (int k : cols) {
            const int t = nx(order[h], k);
            if (!accepting(t) && !pre[t]) pre[t] = 1, order.push_back(t);
        }
    // a walk that can die before it ever accepted (sink, or a dead state) would be revived by a restart: no filter
    for (int s : order)
        if (s <= dead_hi) return no("the walk can end before a first match");

    // ---- shortest accepted string
    int min_len = -1;
    {
        std::vector<int> dist(n_dev, -1), q{start};
        dist[start] = 0;
        for (size_t h = 0; h < q.size() && min_len < 0; ++h)
            for (int k : cols) {
                const int t = nx(q[h], k);
                if (accepting(t)) { min_len = dist[q[h]] + 1; break; }
                if (dist[t] < 0) dist[t] = dist[q[h]] + 1, q.push_back(t);
            }
    }
    if (min_len < 0) return no("no accepting state is reachable");
    if (min_len < kN) return no("matches shorter than a window");

    // ---- K: pairs (state of the walk from the row's start, state of a walk restarted in `start`) run in lockstep until they meet
    int warm = -1;
    {
        std::vector<uint32_t> cur;
        for (int s : order)
            if (s != start) cur.push_back((uint32_t)s << 16 | (uint32_t)start);
        for (int k = 1; k <= kMaxWarm && warm < 0; ++k) {
            std::vector<uint32_t> nxt;
            bool apart = false; // some pair is still in two states after k chars
            for (uint32_t pr : cur) {
                const int s = (int)(pr >> 16), t = (int)(pr & 0xFFFFu);
                for (int c : cols) {
                    const int s2 = nx(s, c), t2 = nx(t, c);
                    if (!accepting(s2) && accepting(t2)) return no("a restarted walk can accept where the walk from the row's start does not");
                    if (s2 == t2) continue;
                    apart = true; // (an accept of the real walk that the restarted one misses counts: the k-th char is where reports begin)
                    // the real walk accepted inside the run-up: this restart is not the one that reports it -- nothing to follow
                    if (!accepting(s2)) nxt.push_back((uint32_t)s2 << 16 | (uint32_t)t2);
                }
                if (nxt.size() > kMaxFrontier) {
                    std::sort(nxt.begin(), nxt.end());
                    nxt.erase(std::unique(nxt.begin(), nxt.end()), nxt.end());
                    if (nxt.size() > kMaxFrontier / 2) return no("too many state pairs");
                }
            }
            std::sort(nxt.begin(), nxt.end());
            nxt.erase(std::unique(nxt.begin(), nxt.end()), nxt.end());
            cur.swap(nxt);
            if (!apart) warm = k;
        }
        if (order.size() == 1) warm = 0;
    }
    if (warm < 0) return no("a restarted walk does not catch up within 14 chars");

    // ---- stride: one window every S chars needs min_len >= 4 + S - 1 and K + S - 1 <= 16
    // (NEEDLE_PREFILTER_STRIDE=2: never 4 -- stride 4 halves the probes but a pattern whose shortest match is 7 chars then has no second
    // level, which wants 5 + S - 1 <= min_len: A/B)
    static const int max_stride = getenv("NEEDLE_PREFILTER_STRIDE") ? atoi(getenv("NEEDLE_PREFILTER_STRIDE")) : 4;
    // Stride 4 only where the second level survives it (5 + 4 - 1 <= min_len): with shortest matches of 7 chars, stride 2 + the second
    // level beats stride 4 without one (c3u, a dictionary followed by [0-9]+: 0.89 against 1.00 ms) -- what the automaton runs on costs
    // more than the probes saved.  NEEDLE_PREFILTER_LEVEL2=0 (no second level at all): stride 4 from 7 chars on, as before.
    static const bool level2_wanted = !(getenv("NEEDLE_PREFILTER_LEVEL2") && atoi(getenv("NEEDLE_PREFILTER_LEVEL2")) == 0);
    // Round 6: the TWO-SIDED second level (NgramParams::on2 == 2) needs min_len >= 5 + S - 2 only: shortest matches of 7 chars take stride 4
    // with it (was: stride 2), of 5 chars stride 2 with it (was: no second level).  The wide filter keeps the one-sided second level.
    // (Tried and dropped: stride 3 -- 24-byte pieces, 1.5 K units, a third fewer probes per byte than stride 2 -- for shortest matches of 6
    // chars: c3s 0.600 against 0.604 ms, c3x 0.79 against 0.73: the extra candidates of its two-sided second level cost what the probes save;
    // profiles/r06_filter_trace.md.)
    int S = 1;
    for (int cand : {4, 2}) {
        const int need2 = kN + 1 + cand - (wide ? 1 : 2); // the shortest match for which this stride still has a second level
        if (cand > 2 && level2_wanted && min_len < need2) continue;
        if (cand <= max_stride && kN + cand - 1 <= min_len && warm + cand - 1 <= 16) { S = cand; break; }
    }
    if (S == 1) return no("matches shorter than 5 chars: every char would need a window (the kernel samples every 2nd or 4th)");
    if (warm + S - 1 > 16) return no("run-up longer than one load");

    // ---- T[j]: states from which a FIRST accepting transition is exactly j chars away (j = 0: the accepting states entered
    // from a pre-accept state); T[j >= 1] within the pre-accept states
    const int depth = kN + 1 + S - 1; // (one deeper: the second level's 5-column windows)
    std::vector<std::vector<uint8_t>> T(depth + 1, std::vector<uint8_t>(n_dev, 0));
    for (int s : order)
        for (int k : cols)
            if (accepting(nx(s, k))) T[0][nx(s, k)] = 1;
    for (int j = 1; j <= depth; ++j)
        for (int s : order)
            for (int k : cols)
                if (T[j - 1][nx(s, k)]) { T[j][s] = 1; break; }

    // ---- windows: label sequences x1..xN of paths p0 -> .. -> pN with p_m in T[o + N - m]; frontier = (state, labels so far).
    // N = 4: the filter's windows; N = 5: the second level's.  false: too many paths.
    auto enum_labels = [&](int N, std::unordered_set<uint64_t> &labels, int n_off) -> bool {
        for (int o = 0; o < n_off; ++o) {
            std::vector<std::pair<uint32_t, uint64_t>> fr; // (state, labels: 8 bits per column)
            for (int s : order)
                if (T[o + N][s]) fr.emplace_back((uint32_t)s, 0ull);
            for (int m = 1; m <= N; ++m) {
                std::vector<std::pair<uint32_t, uint64_t>> nf;
                for (const auto &e : fr) {
                    const int s = (int)e.first;
                    for (int k : cols) {
                        const int t = nx(s, k);
                        if (!T[o + N - m][t]) continue;
                        if (m < N && accepting(t)) continue; // (only the last step of a path may enter an accepting state, and only for o = 0)
                        nf.emplace_back((uint32_t)t, e.second | (uint64_t)k << (8 * (m - 1)));
                    }
                    if (nf.size() > kMaxFrontier) {
                        std::sort(nf.begin(), nf.end());
                        nf.erase(std::unique(nf.begin(), nf.end()), nf.end());
                        if (nf.size() > kMaxFrontier / 2) return false;
                    }
                }
                std::sort(nf.begin(), nf.end());
                nf.erase(std::unique(nf.begin(), nf.end()), nf.end());
                fr.swap(nf);
            }
            for (const auto &e : fr) labels.insert(e.second);
        }
        return true;
    };
    std::unordered_set<uint64_t> labels;
    if (!enum_labels(kN, labels, S)) return no("too many window paths");
    if (labels.empty()) return no("no windows");

    // ---- expand columns to code units (16 bits per position; 8-bit programs: values below 256)
    std::vector<uint64_t> grams;
    for (uint64_t lab : labels) {
        const std::vector<uint16_t> *b[kN];
        size_t n = 1;
        for (int i = 0; i < kN; ++i) {
            b[i] = &bytes_of[(lab >> (8 * i)) & 255u];
            n *= b[i]->size();
            if (n > kMaxWindows) return no("too many byte windows");
        }
        if (grams.size() + n > kMaxWindows) return no("too many byte windows");
        for (uint16_t c0 : *b[0])
            for (uint16_t c1 : *b[1])
                for (uint16_t c2 : *b[2])
                    for (uint16_t c3 : *b[3]) grams.push_back((uint64_t)c0 | (uint64_t)c1 << 16 | (uint64_t)c2 << 32 | (uint64_t)c3 << 48);
    }
    std::sort(grams.begin(), grams.end());
    grams.erase(std::unique(grams.begin(), grams.end()), grams.end());
    // the hash of a window (needle_ngram.h): 8-bit text: the four bytes as one dword, two 16-bit multipliers; wide: two dwords, four
    auto sym = [](uint64_t g, int i) -> uint32_t { return (uint32_t)(g >> (16 * i)) & 0xFFFFu; };
    auto hash_of = [&](uint64_t g, uint32_t m1, uint32_t m2, uint32_t m1b, uint32_t m2b) -> uint32_t {
        if (wide) return ngram_hash16_host(sym(g, 0) | sym(g, 1) << 16, sym(g, 2) | sym(g, 3) << 16, m1, m2, m1b, m2b);
        return ngram_hash_host(sym(g, 0) | sym(g, 1) << 8 | sym(g, 2) << 16 | sym(g, 3) << 24, m1, m2);
    };

    // ---- bitmap size: the largest power of two that fits behind the program (ngram_layout), at most 64 KiB; useless when it fills up
    size_t bm_bytes = 65536;
    while (bm_bytes > 4096 && !ngram_layout((uint32_t)prog_lds_bytes, (uint32_t)bm_bytes, nullptr)) bm_bytes >>= 1;
    while (bm_bytes > 4096 && grams.size() * 256 < bm_bytes * 8) bm_bytes >>= 1; // (fill below 1/256: a smaller one is as good)
    const double fill = (double)grams.size() / (double)(bm_bytes * 8);
    if (fill > 0.05) return no("the windows would fill the bitmap");
    f.p.addr_shift = 24;
    f.p.addr_mask = (uint32_t)(bm_bytes - 1) & ~3u;
    f.p.bm_bytes = (uint32_t)bm_bytes;

    // ---- multipliers: the text is not ours to know; judge a pair by the windows over the SAME bytes (per position) that are
    // not in the set -- near misses are what real text is made of
    std::vector<uint16_t> alpha[kN];
    {
        std::vector<uint8_t> seen((size_t)kN * 65536, 0);
        for (uint64_t g : grams)
            for (int i = 0; i < kN; ++i) seen[(size_t)i * 65536 + sym(g, i)] = 1;
        for (int i = 0; i < kN; ++i)
            for (int c = 0; c < n_sym; ++c)
                if (seen[(size_t)i * 65536 + c]) alpha[i].push_back((uint16_t)c);
    }
    // (16-bit odd multipliers: the word's address is bits 2 .. of u, its bit u's bits 24 .. 28 -- both halves of the window reach both)
    static const uint32_t kMul[][2] = {{0x9E37u, 0x85EBu}, {0xB529u, 0x68E3u}, {0x7FEBu, 0xC2B3u}, {0xD35Bu, 0x1B87u}, {0xA24Bu, 0xE655u}, {0x2C1Bu, 0x5BD1u},
                                       {0xC6A5u, 0x935Du}, {0x6A09u, 0xBB67u}, {0x3C6Fu, 0xA54Fu}, {0x510Fu, 0x9B05u}, {0x1F83u, 0x5BE1u}, {0xCBBBu, 0x9D5Du},
                                       {0x629Bu, 0x367Du}, {0x9159u, 0x152Fu}, {0xF70Fu, 0x4FA5u}, {0x8EB5u, 0x7B3Du}};
    size_t best_fp = (size_t)-1;
    std::vector<uint32_t> bm(bm_bytes / 4);
    const size_t n_mul = sizeof(kMul) / sizeof(kMul[0]);
    for (size_t mi = 0; mi < n_mul; ++mi) {
        const uint32_t *mm = kMul[mi], *mb = kMul[(mi + 5) % n_mul]; // (wide: the second dword of a window has multipliers of its own)
        std::fill(bm.begin(), bm.end(), 0u);
        for (uint64_t g : grams) {
            const uint32_t u = hash_of(g, mm[0], mm[1], mb[0], mb[1]);
            bm[ngram_word_index(u, f.p.addr_mask)] |= ngram_word_bits(u, f.p.addr_shift);
        }
        uint64_t seed = 0x5EED1234u;
        size_t fp = 0;
        for (int t = 0; t < 65536; ++t) {
            const uint64_t r = splitmix(seed);
            uint64_t x = 0;
            for (int i = 0; i < kN; ++i) x |= (uint64_t)alpha[i][(r >> (16 * i)) % alpha[i].size()] << (16 * i);
            const uint32_t u = hash_of(x, mm[0], mm[1], mb[0], mb[1]), bits = ngram_word_bits(u, f.p.addr_shift);
            fp += (bm[ngram_word_index(u, f.p.addr_mask)] & bits) == bits;
        }
        if (fp < best_fp) {
            best_fp = fp;
            f.p.m1 = mm[0], f.p.m2 = mm[1];
            f.p.m1b = wide ? mb[0] : 0u, f.p.m2b = wide ? mb[1] : 0u;
            f.bitmap = bm;
        }
    }
    f.p.wide = wide ? 1u : 0u;
    // ---- second level (needle_ngram.h): the 5-byte windows, same construction one column deeper -- when every match is long enough
    // for them to lie inside it (else their first column is "any char" and they select nothing)
    f.p.on2 = 0;
    static const bool level2_on = !(getenv("NEEDLE_PREFILTER_LEVEL2") && atoi(getenv("NEEDLE_PREFILTER_LEVEL2")) == 0);
    // TWO-SIDED (min_len == 5 + S - 2; not for the wide filter): the 5-column windows ending o = 0 .. S - 2 chars ahead of a first accept.  A
    // candidate's window ends o chars ahead of the accept it announces, o = 0 .. S - 1 (unknown to the kernel).  o <= S - 2: the 5 chars
    // ending where the window ends lie inside the match (it is >= 5 + o chars long) and are in the set.  o = S - 1: the 5 chars ending
    // one char BEHIND the window's end -- the window's four and the match's next char -- end S - 2 ahead of the accept: in the set.  The
    // kernel asks for both and lets the candidate pass when either is there (needle_ngram.hip level2).
    const bool two_sided = !wide && S >= 2 && min_len == kN + 1 + S - 2;
    if (level2_on && (min_len >= kN + 1 + S - 1 || two_sided)) {
        std::unordered_set<uint64_t> labels5;
        std::vector<std::pair<uint64_t, uint32_t>> grams5; // (the 4-unit window, the unit in front of it)
        bool ok5 = enum_labels(kN + 1, labels5, two_sided ? S - 1 : S) && !labels5.empty();
        for (uint64_t lab : labels5) {
            if (!ok5) break;
            const std::vector<uint16_t> *b[kN + 1];
            size_t n = 1;
            for (int i = 0; i <= kN; ++i) {
                b[i] = &bytes_of[(lab >> (8 * i)) & 255u];
                n *= b[i]->size();
                if (n > kMaxWindows) break;
            }
            if (n > kMaxWindows || grams5.size() + n > kMaxWindows) { ok5 = false; break; }
            for (uint16_t c0 : *b[0])
                for (uint16_t c1 : *b[1])
                    for (uint16_t c2 : *b[2])
                        for (uint16_t c3 : *b[3])
                            for (uint16_t c4 : *b[4])
                                grams5.emplace_back((uint64_t)c1 | (uint64_t)c2 << 16 | (uint64_t)c3 << 32 | (uint64_t)c4 << 48, (uint32_t)c0);
        }
        if (ok5) {
            std::sort(grams5.begin(), grams5.end());
            grams5.erase(std::unique(grams5.begin(), grams5.end()), grams5.end());
            // the largest second bitmap (<= 16 KiB) that still fits behind the queues; useless when it fills up
            size_t bm2 = 16384;
            while (bm2 >= 1024 && !ngram_layout((uint32_t)prog_lds_bytes, (uint32_t)bm_bytes, nullptr, kNgWaveLds, (uint32_t)bm2)) bm2 >>= 1;
            while (bm2 > 1024 && grams5.size() * 128 < bm2 * 8) bm2 >>= 1;
            if (bm2 >= 1024 && (double)grams5.size() / (double)(bm2 * 8) <= 0.10) {
                f.p.on2 = two_sided ? 2u : 1u;
                f.p.m3 = 0x9E3779u;
                f.p.bm2_bytes = (uint32_t)bm2;
                f.p.addr_mask2 = (uint32_t)(bm2 - 1) & ~3u;
                f.p.n_grams2 = (uint32_t)grams5.size();
                f.bitmap2.assign(bm2 / 4, 0u);
                for (const auto &g5 : grams5) {
                    const uint32_t u = hash_of(g5.first, f.p.m1, f.p.m2, f.p.m1b, f.p.m2b) + g5.second * f.p.m3; // (ngram_hash2_host)
                    f.bitmap2[ngram_word_index(u, f.p.addr_mask2)] |= ngram_word_bits(u, 24u);
                }
            }
        }
    }
    f.p.on = 1;
    f.p.stride = (uint32_t)S;
    f.p.warm = (uint32_t)warm;
    f.p.min_len = (uint32_t)min_len;
    f.p.n_grams = (uint32_t)grams.size();
    return f;
}

} // namespace needle
