// Regex -> the four automata + char-class map + tables, in the layout the reference's generated class holds.
//
// A from-scratch C++ table generator that follows the reference's compile pipeline stage by stage so that the
// resulting TABLES (not just the language) agree with it -- state numbering included wherever the reference is
// deterministic.  Stages and the reference code each one follows (needle-compiler/src/main/java/com/
// justinblank/strings/...):
//
//   Parser            RegexParser.java:100-275 (stack machine), :297-370 (collapse rules), :372-529 (escapes),
//                     :586-760 (char sets);  AST rules RegexAST/Union.java:62-80,82-165, Concatenation.java:22-41,
//                     LiteralNode.java, Node.reversed()/minLength()/maxLength() in each RegexAST class
//   build_program     RegexInstrBuilder.java:28-209 (Thompson program with per-instruction priorities)
//   SubsetBuilder     NFAToDFACompiler.java:34-178 (modes BASIC / CONTAINED_IN / DFA_SEARCH), StateSet.java:16-53,
//                     NFA.java:225-266 (epsilon closure), CharRange.java:81-157 (coverings)
//   prune_dead        DFA.java:745-792
//   minimize          MinimizeDFA.java:18-217 (result = coarsest partition stable under "same range list, targets in
//                     the same block", numbered in first-encounter order :23-30)
//   byte_classes      DFA.java:438-566, RangeGroup.java
//   fill_table        DFAStateTransitions.java:30-62, DFAClassBuilder.java:240-253 (row stride), :79-85
//   compile_regex     DFACompiler.java:45-83, Factorization.java:101-106,351-378 (min/max length only)
//
// Java's HashSet<Integer> iteration order is observable in NFAToDFACompiler.getEpsilonClosure (ties between equal
// distances keep the FIRST priority, StateSet.java:16-27), so sets of NFA states are kept in a small emulation of
// java.util.HashMap's bucket order (JOrder below; no treeified bins -- not reachable with these key patterns).
//
// UNICODE_CHARACTER_CLASS forms of \d \s \w and UNICODE_CASE folding come, in the reference, from whatever Unicode
// version the running JDK's java.lang.Character carries (RegexParser.java:40-63,277-291).  Here they come from
// needle_unicode_tables.h, generated (scripts/gen_unicode_tables.py) from the Unicode Character Database 13.0.0 -- what
// JDK 15..18 answer; a reference running on an older or newer JDK differs exactly where Unicode itself changed.
#include "needle_regex.h"
#include "needle_unicode_tables.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cstdint>
#include <map>
#include <memory>
#include <system_error>
#include <thread>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/needle_hip.h"

namespace needle {
namespace {

struct SyntaxError : std::runtime_error { using std::runtime_error::runtime_error; };
struct CompileError : std::runtime_error { using std::runtime_error::runtime_error; };
struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };

struct CR { int start, end; }; // inclusive char range, 0..0xFFFF
inline bool operator==(const CR &a, const CR &b) { return a.start == b.start && a.end == b.end; }
inline bool cr_less(const CR &a, const CR &b) { return a.start != b.start ? a.start < b.start : a.end < b.end; }

// ------------------------------------------------------------------------------------------------ Unicode data
// java.lang.Character's answers for the BMP (needle_unicode_tables.h): the char sets behind \d \s \w under
// UNICODE_CHARACTER_CLASS and Character.toUpperCase / toLowerCase(char) behind UNICODE_CASE.
struct UniData {
    std::vector<uint16_t> upper, lower;             // simple case mappings, identity where none
    std::vector<std::vector<uint16_t>> by_key;      // key = toLowerCase(toUpperCase(x)) -> every x < 0xFFFF with that key, ascending
    std::vector<int> digits, spaces, words;         // members, ascending (U+FFFF never is one)
    UniData() : upper(65536), lower(65536), by_key(65536) {
        for (int c = 0; c < 65536; ++c) upper[c] = lower[c] = (uint16_t)c;
        for (int i = 0; i < needle_unicode::kUpper_n; ++i) upper[needle_unicode::kUpper[i][0]] = needle_unicode::kUpper[i][1];
        for (int i = 0; i < needle_unicode::kLower_n; ++i) lower[needle_unicode::kLower[i][0]] = needle_unicode::kLower[i][1];
        for (int x = 0; x < 0xFFFF; ++x) by_key[lower[upper[x]]].push_back((uint16_t)x); // `candidate < Character.MAX_VALUE`
        auto expand = [](const uint16_t (*r)[2], int n, std::vector<int> &out) {
            for (int i = 0; i < n; ++i)
                for (int c = r[i][0]; c <= r[i][1]; ++c) out.push_back(c);
        };
        expand(needle_unicode::kDigit, needle_unicode::kDigit_n, digits);
        expand(needle_unicode::kSpace, needle_unicode::kSpace_n, spaces);
        expand(needle_unicode::kWord, needle_unicode::kWord_n, words);
    }
};
const UniData &uni() {
    static const UniData d;
    return d;
}
// capacity of a java.util.HashMap after n insertions (16 buckets, doubling above a load of 3/4)
struct JOrderLite {
    int cap = 16, size = 0;
    void on_insert() {
        ++size;
        if (size > cap / 4 * 3) cap *= 2;
    }
};

// ------------------------------------------------------------------------------------------------ AST
enum Kind { K_LITERAL, K_RANGE, K_CONCAT, K_UNION, K_REP, K_COUNTED, K_LPAREN };
struct Node;
using NodeP = std::shared_ptr<Node>;
struct Node {
    Kind kind;
    std::u16string lit; // K_LITERAL (mutable: the reference appends in place)
    CR range{0, 0};     // K_RANGE
    NodeP a, b;         // concat: head/tail; union: left/right; rep/counted: a
    bool with_priority = false;
    int min = 0, max = 0;
};
NodeP mk(Kind k) { auto n = std::make_shared<Node>(); n->kind = k; return n; }
NodeP lit(const std::u16string &s) { auto n = mk(K_LITERAL); n->lit = s; return n; }
NodeP range(int s, int e) {
    if (s > e) throw SyntaxError("Tried to create a character range with start larger than end");
    auto n = mk(K_RANGE); n->range = CR{s, e}; return n;
}
bool node_equals(const NodeP &x, const NodeP &y) { // LiteralNode.equals by content, identity otherwise
    if (!x) throw SyntaxError("Unknown error while parsing regex (null union branch)");
    if (!y) return false;
    if (x->kind == K_LITERAL) return y->kind == K_LITERAL && x->lit == y->lit;
    return x.get() == y.get();
}
NodeP make_union(const NodeP &l, const NodeP &r, bool prio) { // Union.of, Union.java:62-80
    if (!l) throw SyntaxError("Cannot union nothing");
    if (node_equals(l, r)) return l;
    if (l->kind == K_UNION) {
        if (node_equals(l->a, r) || node_equals(l->b, r)) return l;
    }
    if (r && r->kind == K_UNION) {
        if (node_equals(r->a, l) || node_equals(r->b, l)) return r;
    }
    auto n = mk(K_UNION); n->a = l; n->b = r; n->with_priority = prio; return n;
}
NodeP concat_nodes(const NodeP &h, const NodeP &t) { // Concatenation.concatenate, Concatenation.java:22-41
    if (!h || !t) throw SyntaxError("Cannot concatenate nothing");
    if (h->kind == K_LITERAL && t->kind == K_LITERAL) { h->lit += t->lit; return h; }
    if (h->kind == K_LITERAL && t->kind == K_CONCAT && t->a->kind == K_LITERAL) {
        h->lit += t->a->lit;
        auto n = mk(K_CONCAT); n->a = h; n->b = t->b; return n;
    }
    auto n = mk(K_CONCAT); n->a = h; n->b = t; return n;
}
NodeP counted(const NodeP &x, int mn, int mx) {
    if (!x) throw SyntaxError("Cannot repeat nothing");
    if (mn > mx) throw SyntaxError("Repetition with invalid range");
    auto n = mk(K_COUNTED); n->a = x; n->min = mn; n->max = mx; return n;
}
NodeP rep(const NodeP &x) {
    if (!x) throw SyntaxError("Cannot repeat nothing");
    auto n = mk(K_REP); n->a = x; return n;
}
NodeP of_chars(std::vector<int> cs) { // Union.ofChars, Union.java:82-117
    if (cs.empty()) throw SyntaxError("Cannot create a union of zero characters");
    if (cs.size() == 1) return range(cs[0], cs[0]);
    std::sort(cs.begin(), cs.end());
    NodeP u;
    size_t start = 0;
    for (size_t i = 0; i < cs.size(); ++i) {
        const bool last = i + 1 == cs.size();
        if (last || cs[i] + 1 != cs[i + 1]) {
            NodeP r = range(cs[start], cs[i]);
            u = u ? make_union(u, r, false) : r;
            start = i + 1;
        }
    }
    return u;
}
NodeP complement_ranges(std::vector<CR> rs) { // Union.complement(List), Union.java:119-152
    if (rs.empty()) throw SyntaxError("Can't complement empty set of ranges");
    std::sort(rs.begin(), rs.end(), cr_less);
    std::vector<NodeP> out;
    bool have_last = false;
    CR lastr{0, 0};
    for (const CR &cur : rs) {
        if (have_last) {
            const int low = (lastr.end + 1) & 0xFFFF, high = (cur.start - 1) & 0xFFFF;
            if (low <= high) out.push_back(range(low, high));
        } else {
            out.push_back(range(0, (cur.start - 1) & 0xFFFF));
        }
        lastr = cur;
        have_last = true;
    }
    out.push_back(range((lastr.end + 1) & 0xFFFF, 0xFFFF));
    if (out.size() < 2) throw SyntaxError("Unknown error while parsing regex (complement)");
    NodeP u = make_union(out[0], out[1], false);
    for (size_t i = 2; i < out.size(); ++i) u = make_union(u, out[i], false);
    return u;
}
NodeP complement_chars(const std::vector<int> &cs) { // Union.complement(String)
    if (cs.size() < 2) throw SyntaxError("Silly short complement");
    std::vector<CR> rs;
    for (int c : cs) rs.push_back(CR{c, c});
    return complement_ranges(rs);
}

int min_length(const NodeP &n) {
    switch (n->kind) {
    case K_LITERAL: return (int)n->lit.size();
    case K_RANGE: return 1;
    case K_CONCAT: return min_length(n->a) + min_length(n->b);
    case K_UNION: if (!n->b) throw CompileError("union with an empty branch"); return std::min(min_length(n->a), min_length(n->b));
    case K_REP: return 0;
    case K_COUNTED: return n->min * min_length(n->a);
    default: throw CompileError("unexpected node");
    }
}
long max_length(const NodeP &n) { // -1 = unbounded
    switch (n->kind) {
    case K_LITERAL: return (long)n->lit.size();
    case K_RANGE: return 1;
    case K_CONCAT: { long x = max_length(n->a), y = max_length(n->b); return (x < 0 || y < 0) ? -1 : x + y; }
    case K_UNION: { if (!n->b) throw CompileError("union with an empty branch"); long x = max_length(n->a), y = max_length(n->b); return (x < 0 || y < 0) ? -1 : std::max(x, y); }
    case K_REP: return -1;
    case K_COUNTED: { long x = max_length(n->a); return x < 0 ? -1 : x * n->max; }
    default: throw CompileError("unexpected node");
    }
}
NodeP reversed(const NodeP &n) {
    switch (n->kind) {
    case K_LITERAL: { std::u16string s(n->lit.rbegin(), n->lit.rend()); return lit(s); }
    case K_RANGE: return n;
    case K_CONCAT: { auto c = mk(K_CONCAT); c->a = reversed(n->b); c->b = reversed(n->a); return c; }
    case K_UNION: if (!n->b) throw CompileError("union with an empty branch"); return make_union(reversed(n->a), reversed(n->b), false);
    case K_REP: return rep(reversed(n->a));
    case K_COUNTED: return counted(reversed(n->a), n->min, n->max);
    default: throw CompileError("unexpected node");
    }
}

// ------------------------------------------------------------------------------------------------ parser
class Parser {
  public:
    Parser(const std::u16string &r, int flags) : re(r) {
        dot_all = flags & NEEDLE_DOTALL;
        ci = flags & NEEDLE_CASE_INSENSITIVE;
        ucc = flags & NEEDLE_UNICODE_CHARACTER_CLASS;
        uci = ucc ? true : (flags & NEEDLE_UNICODE_CASE) != 0;
    }
    NodeP parse() {
        while (idx < re.size()) {
            const char16_t c = take();
            switch (c) {
            case u'.':
                if (dot_all) push(range(0, 0xFFFF));
                else push(make_union(range(0, 9), make_union(range(0xB, 0xC), range(0xE, 0xFFFF), false), false));
                break;
            case u'^': throw err("'^' not supported yet");
            case u'$': throw err("'$' not supported yet");
            case u'(':
                push(mk(K_LPAREN));
                if (peek_str(u"?:")) { take(); take(); }
                else if (peek_str(u"?<")) consume_named_group();
                break;
            case u'{': {
                if (nodes.empty()) throw err("Found '{' with no preceding regex");
                const int left = consume_int();
                char16_t next = take();
                if (next == u'}') {
                    push(counted(pop(), left, left));
                    no_lazy_or_possessive();
                    break;
                } else if (next != u',') throw err("Expected ','");
                const int right = consume_int();
                push(counted(pop(), left, right));
                next = take();
                if (next != u'}') throw err("Found unclosed brackets");
                no_lazy_or_possessive();
                break;
            }
            case u'?':
                if (nodes.empty()) throw err("'?' with no preceding regex");
                no_lazy_or_possessive();
                push(counted(pop(), 0, 1));
                break;
            case u'[': {
                NodeP n = build_char_set();
                if (n) push(n);
                break;
            }
            case u'+': {
                if (nodes.empty()) throw err("Found '+' with no preceding regex");
                no_lazy_or_possessive();
                NodeP last = pop();
                push(concatenate(last, rep(last)));
                break;
            }
            case u'*':
                if (nodes.empty()) throw err("Found '*' with no preceding regex");
                no_lazy_or_possessive();
                push(rep(pop()));
                break;
            case u'|': {
                if (nodes.empty()) throw err("'|' cannot be the final character in a regex");
                collapse_literals();
                NodeP last = pop();
                push(make_union(last, nullptr, true));
                break;
            }
            case u'\\': push(parse_escape()); break;
            case u')': collapse_paren(); break;
            default: push_literal_char(c);
            }
        }
        if (nodes.empty()) return lit(u"");
        NodeP node = pop();
        if (node->kind == K_LPAREN) throw err("Unbalanced '(' found");
        while (!nodes.empty()) {
            NodeP next = pop();
            if (next->kind == K_UNION && !next->b) node = make_union(next->a, node, true);
            else if (next->kind == K_LITERAL && node->kind == K_LITERAL) node = lit(next->lit + node->lit);
            else if (next->kind == K_LPAREN) throw err("Unbalanced '(' found");
            else node = concatenate(next, node);
        }
        return node;
    }

  private:
    const std::u16string &re;
    size_t idx = 0;
    bool dot_all, ci, uci, ucc;
    std::vector<NodeP> nodes;

    SyntaxError err(const std::string &m) { return SyntaxError(m + ". Regex index=" + std::to_string(idx)); }
    char16_t take() {
        if (idx >= re.size()) throw err("Unknown error while parsing regex (unexpected end)");
        return re[idx++];
    }
    bool peek_char(char16_t c) const { return idx < re.size() && re[idx] == c; }
    bool peek_str(const char16_t *s) const {
        size_t i = idx;
        for (; *s; ++s, ++i)
            if (i >= re.size() || re[i] != *s) return false;
        return true;
    }
    void push(const NodeP &n) { nodes.push_back(n); }
    NodeP pop() {
        if (nodes.empty()) throw err("Unknown error while parsing regex (empty stack)");
        NodeP n = nodes.back();
        nodes.pop_back();
        return n;
    }
    void no_lazy_or_possessive() {
        if (peek_char(u'?')) throw err("Reluctant quantifiers are not supported");
        if (peek_char(u'+')) throw err("Possessive quantifiers are not supported");
    }
    NodeP concatenate(const NodeP &next, const NodeP &node) { // RegexParser.concatenate :358-364
        if (next->kind == K_LITERAL && node->kind == K_LITERAL) { next->lit += node->lit; return next; }
        if (next->kind == K_LPAREN || node->kind == K_LPAREN) throw err("Unknown error while parsing regex (paren)");
        return concat_nodes(next, node);
    }
    void push_literal_char(char16_t c) { // RegexParser.java:211-247
        if (!ci) { push(lit(std::u16string(1, c))); return; }
        if (uci) {
            std::vector<int> cs = unicode_case_variants(c);
            if (cs.size() > 1) {
                cs = hash_set_order(cs); // `for (var caseChar : characters)`, RegexParser.java:218
                NodeP u;
                for (int v : cs) { NodeP l = lit(std::u16string(1, (char16_t)v)); u = u ? make_union(u, l, false) : l; }
                push(u);
            } else push(lit(std::u16string(1, c)));
            return;
        }
        NodeP n = lit(std::u16string(1, c));
        if (c >= u'A' && c <= u'Z') push(make_union(n, lit(std::u16string(1, (char16_t)(c + 32))), false));
        else if (c >= u'a' && c <= u'z') push(make_union(n, lit(std::u16string(1, (char16_t)(c - 32))), false));
        else push(n);
    }
    // RegexParser.addCaseInsensitiveMatches :277-291: c itself, and -- when c is cased, i.e. toLowerCase(toUpperCase(c))
    // differs from toUpperCase(c) -- every x < U+FFFF with the same toLowerCase(toUpperCase(x)).  (Not an equivalence
    // relation: U+1E9E finds U+00DF, U+00DF alone finds nothing.)  In insertion order: c, then the others ascending.
    static std::vector<int> unicode_case_variants(int c) {
        const UniData &u = uni();
        std::vector<int> out = {c};
        const int up = u.upper[c], lo = u.lower[up];
        if (lo != up)
            for (uint16_t x : u.by_key[lo])
                if (x != c) out.push_back(x);
        return out;
    }
    // the iteration order of the reference's HashSet<Character> holding those chars (hash = the char value)
    static std::vector<int> hash_set_order(const std::vector<int> &inserted) {
        JOrderLite jo;
        for (size_t i = 0; i < inserted.size(); ++i) jo.on_insert();
        std::vector<std::vector<int>> buckets((size_t)jo.cap);
        for (int v : inserted) buckets[(size_t)(v & (jo.cap - 1))].push_back(v);
        std::vector<int> out;
        for (const auto &b : buckets) out.insert(out.end(), b.begin(), b.end());
        return out;
    }
    void collapse_literals() { // :297-321
        NodeP last = pop();
        while (!nodes.empty()) {
            NodeP prev = nodes.back();
            if (prev->kind != K_UNION && prev->kind != K_LPAREN) {
                nodes.pop_back();
                last = concatenate(prev, last);
            } else if (prev->kind == K_UNION) {
                if (!prev->b) { nodes.pop_back(); last = make_union(prev->a, last, true); }
                else { nodes.pop_back(); last = concat_nodes(prev, last); }
            } else break;
        }
        push(last);
    }
    void collapse_paren() { // :323-356
        if (nodes.empty()) throw err("found unbalanced ')'");
        NodeP node;
        for (;;) {
            if (nodes.empty()) throw err("Unknown error while parsing regex (unbalanced)");
            if (nodes.back()->kind == K_LPAREN) break;
            NodeP prev = pop();
            if (!node) node = prev;
            else if (prev->kind == K_UNION) {
                if (prev->a && prev->b) { node = concat_nodes(prev, node); continue; }
                if (nodes.empty()) throw err("found '|' with no preceding content");
                if (nodes.back()->kind == K_LPAREN) {
                    nodes.pop_back();
                    push(make_union(prev->a, node, true));
                    return;
                }
                node = make_union(prev->a, node, true);
            } else node = concatenate(prev, node);
            if (nodes.empty()) throw err("found unbalanced ')'");
        }
        nodes.pop_back();
        if (!node) node = lit(u"");
        push(node);
    }
    int consume_int() { // :535-551
        const size_t start = idx;
        while (idx < re.size()) {
            const char16_t n = re[idx];
            if (n < u'0' || n > u'9') {
                if (idx == start || idx - start > 9) throw err("Expected number");
                int v = 0;
                for (size_t i = start; i < idx; ++i) v = v * 10 + (re[i] - u'0');
                return v;
            }
            ++idx;
        }
        throw err("Expected number");
    }
    void consume_named_group() { // :762-776
        size_t g = idx + 2;
        while (g < re.size()) {
            const char16_t c = re[g];
            if ((c >= u'A' && c <= u'Z') || (c >= u'a' && c <= u'z') || (c >= u'0' && c <= u'9')) ++g;
            else if (c == u'>') { idx = g + 1; return; }
            else throw err("Unknown error while parsing regex (group name)"); // the reference spins here
        }
    }
    bool peek_octal() const { return idx < re.size() && re[idx] >= u'0' && re[idx] <= u'7'; }
    bool peek_hex() const {
        if (idx >= re.size()) return false;
        const char16_t c = re[idx];
        return (c >= u'0' && c <= u'9') || (c >= u'A' && c <= u'F');
    }
    static std::vector<int> chars_of(const char16_t *s) { std::vector<int> v; for (; *s; ++s) v.push_back(*s); return v; }
    NodeP parse_escape() { // :372-529
        if (idx >= re.size()) throw err("'\\' character with nothing following it");
        const char16_t c = take();
        // RegexParser.java:419-422 (\h), :480-483 (\v), :438 (\s)
        static const char16_t HSPACE[] = {0x20, 0x09, 0xA0, 0x1680, 0x180E, 0x2000, 0x2001, 0x2002, 0x2003, 0x2004, 0x2005,
                                          0x2006, 0x2007, 0x2008, 0x2009, 0x200A, 0x202F, 0x205F, 0x3000, 0};
        static const char16_t VSPACE[] = {0x0A, 0x0B, 0x0C, 0x0D, 0x85, 0x2028, 0x2029, 0};
        static const char16_t SPACE[] = {0x20, 0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0};
        switch (c) {
        case u'a': return range(7, 7);
        case u'A': case u'B': case u'b': case u'c': case u'G': case u'p': case u'Z': case u'z':
            throw err("escape not supported yet");
        case u'd': if (ucc) return of_chars(uni().digits); return range('0', '9');       // :393-399
        case u'D': if (ucc) return complement_chars(uni().digits); return complement_ranges({CR{'0', '9'}});
        case u'e': return range(0x1B, 0x1B);
        case u'f': return range(0xC, 0xC);
        case u'H': return complement_chars(chars_of(HSPACE));
        case u'h': return of_chars(chars_of(HSPACE));
        case u'n': return range('\n', '\n');
        case u'r': return range('\r', '\r');
        case u's': if (ucc) return of_chars(uni().spaces); return of_chars(chars_of(SPACE));           // :433-439
        case u'S': if (ucc) return complement_chars(uni().spaces); return complement_chars(chars_of(SPACE));
        case u't': return range('\t', '\t');
        case u'w':
            if (ucc) return of_chars(uni().words); // :452-455
            return make_union(range('0', '9'), make_union(range('_', '_'), make_union(range('a', 'z'), range('A', 'Z'), false), false), false);
        case u'W':
            if (ucc) return complement_chars(uni().words);
            return complement_ranges({CR{'0', '9'}, CR{'_', '_'}, CR{'a', 'z'}, CR{'A', 'Z'}});
        case u'x': {
            int count = 0, v = 0;
            while (count < 2 && peek_hex()) {
                const char16_t h = take();
                v = v * 16 + (h <= u'9' ? h - u'0' : h - u'A' + 10);
                ++count;
            }
            if (count != 2) throw err("Wrong number of hex chars");
            return range(v, v);
        }
        case u'V': return complement_chars(chars_of(VSPACE));
        case u'v': return of_chars(chars_of(VSPACE));
        case u'0': {
            int count = 0, v = 0;
            char16_t first = 0;
            while (count < 3 && peek_octal()) {
                if (count == 2 && first > u'3') break;
                const char16_t o = take();
                if (count == 0) first = o;
                v = v * 8 + (o - u'0');
                ++count;
            }
            if (count == 0) throw err("Illegal octal escape");
            return range(v, v);
        }
        case u'\\': case u'[': case u'|': case u'(': case u')': case u'$': case u'*': case u'?': case u'+': case u'{':
        case u':': case u'^': case u'.':
            return range(c, c);
        default: break;
        }
        if (c >= u'1' && c <= u'9') throw err("Backreferences are not supported");
        if (c < u'A' || (c > u'Z' && c < u'a') || c > u'z') return range(c, c);
        throw err("Escape with unrecognized escaped character");
    }

    static NodeP with_alternate(const NodeP &node, const NodeP &alt) { // :714-723
        if (node) return alt ? make_union(alt, node, false) : node;
        return alt;
    }
    static std::vector<CR> compact(std::vector<CR> rs) { // CharRange.compact, CharRange.java:190-207
        std::sort(rs.begin(), rs.end(), cr_less);
        std::vector<CR> out;
        CR cur = rs[0];
        for (size_t i = 1; i < rs.size(); ++i) {
            if (((cur.end + 1) & 0xFFFF) == rs[i].start) cur = CR{cur.start, rs[i].end};
            else { out.push_back(cur); cur = rs[i]; }
        }
        out.push_back(cur);
        return out;
    }
    static void add_range(std::vector<CR> &set, CR r) { // HashSet<CharRange>.add
        for (const CR &x : set) if (x == r) return;
        set.push_back(r);
    }
    NodeP build_node(const std::vector<CR> &ranges, bool complemented) { // :725-760
        if (ranges.empty()) return nullptr;
        if (ranges.size() == 1) {
            if (complemented) return complement_ranges({ranges[0]});
            return range(ranges[0].start, ranges[0].end);
        }
        std::vector<CR> s = compact(ranges);
        if (s.size() == 1) {
            if (complemented) return complement_ranges({s[0]});
            return range(s[0].start, s[0].end);
        }
        if (complemented) return complement_ranges(s);
        NodeP n = make_union(range(s[0].start, s[0].end), range(s[1].start, s[1].end), false);
        for (size_t i = 2; i < s.size(); ++i) n = make_union(n, range(s[i].start, s[i].end), false);
        return n;
    }
    NodeP build_char_set() { // :586-712
        std::vector<CR> ranges;
        int last = -1;
        const size_t starting = idx;
        NodeP alternate;
        bool complemented = false;
        while (idx < re.size()) {
            char16_t c = take();
            if (c == u'^' && idx == starting + 1) {
                complemented = true;
            } else if (c == u']') {
                if (last >= 0) add_range(ranges, CR{last, last});
                return with_alternate(build_node(ranges, complemented), alternate);
            } else if (c == u'-') {
                if (idx == re.size()) throw err("Unterminated character range");
                if (peek_char(u']')) {
                    if (last >= 0) add_range(ranges, CR{last, last});
                    last = u'-';
                    continue;
                }
                if (last < 0) { last = c; continue; }
                char16_t next = take();
                if (next == u'\\') {
                    if (peek_char(u'[') || peek_char(u']') || peek_char(u'\\')) next = take();
                }
                if (next < last) throw err("Start of range must be less than the end");
                const CR r{last, next};
                if (ci) {
                    if (uci) {
                        std::set<int> cs;
                        for (int rc = last; rc <= next; ++rc)
                            for (int v : unicode_case_variants(rc)) cs.insert(v);
                        for (int v : cs) add_range(ranges, CR{v, v});
                    } else if (next < u'A' || u'z' < last) {
                        add_range(ranges, r);
                    } else {
                        const int us = std::max<int>(r.start, 'A'), ue = std::min<int>(r.end, 'Z');
                        const int ls = std::max<int>(r.start, 'a'), le = std::min<int>(r.end, 'z');
                        if (us <= ue) { add_range(ranges, CR{us, ue}); add_range(ranges, CR{us + 32, ue + 32}); }
                        if (ls <= le) { add_range(ranges, CR{ls, le}); add_range(ranges, CR{ls - 32, le - 32}); }
                        add_range(ranges, r);
                    }
                } else add_range(ranges, r);
                last = -1;
            } else if (c == u'[') {
                if (!ranges.empty()) {
                    std::vector<CR> s = ranges;
                    std::sort(s.begin(), s.end(), cr_less);
                    NodeP rn;
                    for (const CR &x : s) { NodeP q = range(x.start, x.end); rn = rn ? make_union(rn, q, false) : q; }
                    ranges.clear();
                    alternate = with_alternate(rn, alternate);
                }
                NodeP inner = build_char_set();
                if (!inner) throw err("Unbalanced [ token");
                if (peek_char(u']')) {
                    take();
                    return with_alternate(inner, alternate);
                }
                alternate = with_alternate(inner, alternate);
            } else if (c == u'\\') {
                if (peek_char(u'[') || peek_char(u']') || peek_char(u'\\')) {
                    const char16_t n = take();
                    add_range(ranges, CR{n, n});
                    last = n;
                } else alternate = parse_escape();
            } else {
                if (last >= 0) add_range(ranges, CR{last, last});
                last = c;
            }
        }
        throw err("Parsing failed, unmatched [");
    }
};

// ------------------------------------------------------------------------------------------------ Thompson program
enum Opcode { OP_CHAR, OP_JUMP, OP_SPLIT, OP_MATCH };
struct Instr {
    Opcode op;
    int start = 0, end = 0, target = -1, priority = 0;
    std::vector<int> targets;
};
class ProgramBuilder { // RegexInstrBuilder.java
  public:
    explicit ProgramBuilder(bool lml) : leftmost_longest(lml) {}
    std::vector<Instr> build(const NodeP &ast) {
        std::vector<Instr> v;
        partial(ast, v);
        const int match_index = (int)v.size();
        int prio = INT_MAX; // priorityForMatch :38-57
        for (const Instr &i : v) {
            if (i.op == OP_JUMP && i.target == match_index) prio = std::min(prio, i.priority);
            else if (i.op == OP_SPLIT)
                for (int t : i.targets) if (t == match_index) { prio = std::min(prio, i.priority); break; }
        }
        Instr m; m.op = OP_MATCH; m.priority = std::max(prio, 0);
        v.push_back(m);
        auto resolve = [&](int j) { // getResolvedJump :83-91
            int r = -1;
            int t = j;
            while (v[t].op == OP_JUMP) { r = v[t].target; t = r; }
            return r;
        };
        for (Instr &i : v) { // resolveJumps :59-81
            if (i.op == OP_JUMP) { const int r = resolve(i.target); if (r != -1) i.target = r; }
            else if (i.op == OP_SPLIT) for (int &t : i.targets) { const int r = resolve(t); if (r != -1) t = r; }
        }
        return v;
    }

  private:
    bool leftmost_longest;
    int max_priority = 1; // STARTING_PRIORITY
    static Instr placeholder() { Instr i; i.op = OP_JUMP; i.target = -1; i.priority = -1; return i; }
    static Instr jump(int t, int p) { Instr i; i.op = OP_JUMP; i.target = t; i.priority = std::max(p, 0); return i; }
    static Instr split(std::vector<int> t, int p) { Instr i; i.op = OP_SPLIT; i.targets = std::move(t); i.priority = std::max(p, 0); return i; }
    static Instr chr(int s, int e, int p) { Instr i; i.op = OP_CHAR; i.start = s; i.end = e; i.priority = std::max(p, 0); return i; }
    void partial(const NodeP &n, std::vector<Instr> &v) { // createPartial :107-209
        switch (n->kind) {
        case K_CONCAT: partial(n->a, v); partial(n->b, v); break;
        case K_REP: {
            const int split_index = (int)v.size();
            v.push_back(placeholder());
            partial(n->a, v);
            v.push_back(jump(split_index, max_priority));
            if (!leftmost_longest) ++max_priority;
            const int post = (int)v.size();
            v[split_index] = split({split_index + 1, post}, max_priority);
            break;
        }
        case K_COUNTED: {
            int r = 0;
            for (; r < n->min; ++r) partial(n->a, v);
            std::vector<int> sw;
            for (; r < n->max; ++r) { sw.push_back((int)v.size()); v.push_back(placeholder()); partial(n->a, v); }
            const int fin = (int)v.size();
            for (int s : sw) v[s] = split({s + 1, fin}, max_priority);
            break;
        }
        case K_UNION: {
            if (!n->b) throw CompileError("union with an empty branch");
            const int split_index = (int)v.size();
            if (n->a->kind != K_UNION) v.push_back(placeholder());
            const int first_target = (int)v.size();
            const int first_prio = max_priority;
            partial(n->a, v);
            if (!leftmost_longest && n->with_priority) ++max_priority;
            const int first_jump = (int)v.size();
            v.push_back(placeholder());
            const int second_target = (int)v.size();
            partial(n->b, v);
            if (!leftmost_longest && n->with_priority) ++max_priority;
            const int fin = (int)v.size();
            v[first_jump] = jump(fin, first_prio);
            std::vector<int> ts;
            if (first_target >= (int)v.size()) throw CompileError("empty union branch");
            if (v[first_target].op == OP_SPLIT) for (int t : v[first_target].targets) ts.push_back(t);
            else ts.push_back(first_target);
            if (second_target < (int)v.size() && v[second_target].op == OP_SPLIT) for (int t : v[second_target].targets) ts.push_back(t);
            else ts.push_back(second_target);
            v[split_index] = split(ts, first_prio);
            break;
        }
        case K_RANGE: v.push_back(chr(n->range.start, n->range.end, max_priority)); break;
        case K_LITERAL: for (char16_t c : n->lit) v.push_back(chr(c, c, max_priority)); break;
        default: throw CompileError("Unhandled ast node type");
        }
    }
};

// ------------------------------------------------------------------------------------------------ Java HashSet<Integer> order
// Keys in insertion order + the table capacity java.util.HashMap would have after the same history.  Iteration
// order = bucket (key & (cap-1)) ascending, insertion order inside a bucket (splits on resize keep it).
struct JOrder {
    int cap = 0, size = 0;
    void on_insert() {
        if (cap == 0) cap = 16;
        ++size;
        if (size > cap / 4 * 3) cap *= 2;
    }
    void on_remove() { --size; }
};
template <class Item, class KeyFn>
void java_order(const std::vector<Item> &items, int cap, KeyFn key, std::vector<int> &out, std::vector<int> &scratch) {
    const int n = (int)items.size();
    out.resize(n);
    if (n == 0) return;
    const int mask = cap - 1;
    bool sorted = true; // common case: all keys < cap and inserted ascending
    for (int i = 0; i < n; ++i) {
        const int k = key(items[i]);
        if (k > mask || (i && key(items[i - 1]) >= k)) { sorted = false; break; }
    }
    if (sorted) { for (int i = 0; i < n; ++i) out[i] = i; return; }
    scratch.assign(cap + 1, 0);
    for (int i = 0; i < n; ++i) ++scratch[(key(items[i]) & mask) + 1];
    for (int b = 0; b < cap; ++b) scratch[b + 1] += scratch[b];
    for (int i = 0; i < n; ++i) out[scratch[key(items[i]) & mask]++] = i;
}

// ------------------------------------------------------------------------------------------------ DFA
struct Trans { int start, end, target; };
struct DState { bool accepting = false; std::vector<Trans> tr; };
struct Dfa { std::vector<DState> st; };

void add_transition(DState &s, CR r, int target) { // DFA.addTransition :63-83
    for (Trans &t : s.tr) {
        if (t.start == r.start && t.end == r.end) return;
        if (t.end + 1 == r.start && t.target == target) { t.end = r.end; return; }
    }
    s.tr.push_back(Trans{r.start, r.end, target});
    std::stable_sort(s.tr.begin(), s.tr.end(), [](const Trans &a, const Trans &b) { return a.start < b.start; });
}

std::vector<CR> minimal_covering(std::vector<CR> ranges) { // CharRange.minimalCovering :118-157
    if (ranges.size() < 2) return ranges;
    std::stable_sort(ranges.begin(), ranges.end(), [](const CR &a, const CR &b) { return a.start < b.start; });
    std::vector<CR> out;
    int last_start = -1, last_end = -1;
    for (size_t i = 0; i < ranges.size(); ++i) {
        const CR cur = ranges[i];
        while (last_end < cur.end) {
            int start = cur.start, end = cur.end;
            if (last_start >= start) start = last_start + 1;
            if (last_end >= start) start = last_end + 1;
            for (size_t j = i + 1; j < ranges.size(); ++j) {
                const CR nx = ranges[j];
                if (nx.start > start && nx.start <= end) end = nx.start - 1;
                if (nx.end >= start && nx.end <= end) end = nx.end;
            }
            last_start = start;
            last_end = end;
            out.push_back(CR{start, end});
        }
    }
    std::stable_sort(out.begin(), out.end(), [](const CR &a, const CR &b) { return a.start < b.start; });
    return out;
}
std::vector<CR> cover_all_chars(const std::vector<CR> &ranges) { // CharRange.coverAllChars :81-109
    std::vector<CR> all;
    if (ranges.empty()) { all.push_back(CR{0, 0xFFFF}); return all; }
    bool have = false;
    CR cur{0, 0};
    for (const CR &r : ranges) {
        if (!have) {
            cur = r; have = true;
            if (cur.start > 0) all.push_back(CR{0, cur.start - 1});
            all.push_back(cur);
        } else {
            if (r.start > cur.end + 1) all.push_back(CR{cur.end + 1, r.start - 1});
            all.push_back(r);
            cur = r;
        }
    }
    if (cur.end < 0xFFFF) all.push_back(CR{cur.end + 1, 0xFFFF});
    return all;
}

enum ConvMode { BASIC, CONTAINED_IN, DFA_SEARCH };

struct SItem { int key, dist, prio; };
struct StateSet { // StateSet.java
    std::vector<SItem> items; // insertion order of the `states` HashSet
    JOrder jo;
    bool seen_accepting = false;
    // key -> index in items: a small open-addressing table sized by the SET, not by the program (a 3000-keyword dictionary's search
    // automaton has 12 000 states x 28 ranges, each a fresh set over a 25 000-instruction program: an array per set indexed by
    // instruction was 100 GB of memset per compile)
    std::vector<int32_t> ht; // index + 1; 0 = empty; size a power of two
    uint64_t hsum = 0, hxor = 0; // order-independent fingerprint of the key set (SubsetBuilder::lookup)
    explicit StateSet(int) {}
    static uint64_t mix(uint64_t k) {
        k += 0x9E3779B97F4A7C15ull;
        k = (k ^ (k >> 30)) * 0xBF58476D1CE4E5B9ull;
        k = (k ^ (k >> 27)) * 0x94D049BB133111EBull;
        return k ^ (k >> 31);
    }
    int find(int k) const {
        if (ht.empty()) return -1;
        const size_t mask = ht.size() - 1;
        for (size_t h = (size_t)mix((uint64_t)k) & mask;; h = (h + 1) & mask) {
            const int32_t v = ht[h];
            if (v == 0) return -1;
            if (items[v - 1].key == k) return v - 1;
        }
    }
    void rehash(size_t cap) {
        ht.assign(cap, 0);
        const size_t mask = cap - 1;
        for (size_t i = 0; i < items.size(); ++i) {
            size_t h = (size_t)mix((uint64_t)items[i].key) & mask;
            while (ht[h]) h = (h + 1) & mask;
            ht[h] = (int32_t)i + 1;
        }
    }
    void add(int k, int dist, int prio) { // :16-27
        const int i = find(k);
        if (i >= 0) {
            if (items[i].dist < dist) { items[i].dist = dist; items[i].prio = prio; }
            return;
        }
        items.push_back(SItem{k, dist, prio});
        const uint64_t m = mix((uint64_t)k);
        hsum += m;
        hxor ^= mix(m);
        if (items.size() * 2 > ht.size()) rehash(ht.empty() ? 16 : ht.size() * 4);
        else {
            const size_t mask = ht.size() - 1;
            size_t h = (size_t)m & mask;
            while (ht[h]) h = (h + 1) & mask;
            ht[h] = (int32_t)items.size();
        }
        jo.on_insert();
    }
    bool prune(int accepting_state, int boundary, int priority) { // :37-53
        bool removed = false;
        std::vector<SItem> keep;
        for (const SItem &it : items) {
            if (it.key != accepting_state && (it.dist < boundary || priority < it.prio)) {
                removed = true;
                const uint64_t m = mix((uint64_t)it.key);
                hsum -= m;
                hxor ^= mix(m);
                jo.on_remove();
            } else keep.push_back(it);
        }
        if (removed) {
            items.swap(keep);
            size_t cap = 16;
            while (cap < items.size() * 2) cap *= 2;
            rehash(cap);
        }
        return removed;
    }
    std::vector<int> sorted_keys() const {
        std::vector<int> k(items.size());
        for (size_t i = 0; i < items.size(); ++i) k[i] = items[i].key;
        std::sort(k.begin(), k.end());
        return k;
    }
};

class SubsetBuilder { // NFAToDFACompiler.java
  public:
    explicit SubsetBuilder(const std::vector<Instr> &p) : prog(p), n((int)p.size()), closure_cache(p.size()), closure_done(p.size(), 0) {}
    Dfa compile(ConvMode mode) {
        Dfa dfa;
        if (mode != BASIC) make_root_template(); // (before the first store(): has_root of the initial set)
        StateSet init(n);
        init.add(0, 0, 1);
        StateSet states = epsilon_closure(init);
        dfa.st.emplace_back();
        dfa.st[0].accepting = has_accepting(states);
        states.seen_accepting = dfa.st[0].accepting;
        store(states, 0);
        std::vector<StateSet> pending;
        pending.push_back(std::move(states));
        std::vector<int> order, scratch;
        double T[8] = {0};
        long dbg_n[3] = {0, 0, 0};
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        static const bool timing = getenv("NEEDLE_COMPILE_TIMING") != nullptr;
        while (!pending.empty()) {
            double ta = timing ? now() : 0.0; // (the clock is only read when NEEDLE_COMPILE_TIMING asks for it)
            auto lap = [&](int k) { if (timing) { const double tb = now(); T[k] += tb - ta; ta = tb; } };
            StateSet cur = std::move(pending.back());
            pending.pop_back();
            const int dfa_id = lookup(cur);
            if (dfa_id < 0) throw CompileError("internal: popped state set without a DFA state");
            StateSet ec = epsilon_closure(cur);
            lap(0);
            const bool accepting = ec.seen_accepting;
            if (accepting && mode == CONTAINED_IN) continue;
            if (mode == CONTAINED_IN || (!accepting && mode == DFA_SEARCH)) ec.add(0, 0, 1);
            // findCharRanges over the set in HashSet order, then the coverings
            java_order(ec.items, ec.jo.cap, [](const SItem &s) { return s.key; }, order, scratch);
            // (exact duplicates dropped, first occurrences kept in order: minimalCovering's output does not depend on them -- a
            // later copy of a range is covered when its turn comes and clips nothing its original did not -- and a dictionary's
            // root closure holds thousands of copies of 26 letters: its inner loop is quadratic in the list)
            std::vector<CR> crs;
            {
                std::map<std::pair<int, int>, char> seen_cr;
                for (int i : order) {
                    const Instr &in = prog[ec.items[i].key];
                    if (in.op == OP_CHAR && seen_cr.emplace(std::make_pair(in.start, in.end), 1).second) crs.push_back(CR{in.start, in.end});
                }
            }
            const std::vector<CR> ranges = cover_all_chars(minimal_covering(crs));
            lap(1);
            // which items move on which range: the ranges partition the chars (sorted, contiguous), a CHAR instruction covers a run of
            // them -- filed per range in `order` (= the order the reference's loop over the set visits them), instead of testing
            // every item against every range
            std::vector<std::vector<int>> movers(ranges.size());
            for (int i : order) {
                const Instr &in = prog[ec.items[i].key];
                if (in.op != OP_CHAR) continue;
                size_t lo = 0, hi = ranges.size(); // first range whose start >= in.start
                while (lo < hi) {
                    const size_t mid = (lo + hi) / 2;
                    if (ranges[mid].start < in.start) lo = mid + 1;
                    else hi = mid;
                }
                for (size_t k = lo; k < ranges.size() && ranges[k].start <= in.end; ++k) movers[k].push_back(i);
            }
            for (size_t ri = 0; ri < ranges.size(); ++ri) {
                const CR &r = ranges[ri];
                StateSet tr(n); // transition(epsilonClosure, range.getStart()) :168-178
                for (int i : movers[ri]) {
                    const SItem &it = ec.items[i];
                    tr.add(it.key + 1, it.dist + 1, prog[it.key].priority);
                }
                lap(2);
                StateSet post = epsilon_closure(tr);
                lap(3);
                if (!post.seen_accepting) post.seen_accepting = ec.seen_accepting || has_accepting(post);
                if (post.seen_accepting && (mode == CONTAINED_IN || mode == DFA_SEARCH)) {
                    // prune around the (single) MATCH instruction until nothing is removed :92-106
                    const int acc = n - 1;
                    for (;;) {
                        const int ai = post.find(acc);
                        if (ai < 0) break;
                        if (!post.prune(acc, post.items[ai].dist, post.items[ai].prio)) break;
                    }
                }
                lap(4);
                int target = -2;
                if (!post.seen_accepting && (mode == CONTAINED_IN || mode == DFA_SEARCH)) {
                    post.add(0, 0, 1);
                    // (most of these sets exist already: looked up as "the root's closure + these few" without being built)
                    target = lookup_root_plus(post);
                    if (timing) { ++dbg_n[0]; dbg_n[1] += target != -2; dbg_n[2] += target >= 0; }
                    if (target < 0) post = epsilon_closure(post);
                }
                lap(5);
                if (target < 0) target = lookup(post);
                lap(6);
                if (target < 0) {
                    target = (int)dfa.st.size();
                    dfa.st.emplace_back();
                    dfa.st[target].accepting = has_accepting(post);
                    store(post, target);
                    pending.push_back(std::move(post));
                }
                add_transition(dfa.st[dfa_id], r, target);
                lap(7);
            }
            if (dfa.st.size() > 40000) throw CompileError("Can't compile DFAs with more than 16383 states");
        }
        if (timing) fprintf(stderr, "[compile]     root sets %ld fast path %ld found %ld\n", dbg_n[0], dbg_n[1], dbg_n[2]);
        if (timing) fprintf(stderr, "[compile]     closure(cur) %.2f ranges %.2f transition %.2f closure(tr) %.2f prune %.2f root+closure %.2f lookup %.2f store %.2f\n", T[0], T[1], T[2], T[3], T[4], T[5], T[6], T[7]);
        return dfa;
    }

  private:
    const std::vector<Instr> &prog;
    int n;
    struct Stored { bool seen_accepting; int dfa; };
    // HashMap<StateSet, List<Pair<StateSet, DFA>>>: looked up by an order-independent fingerprint of the key set (no sort per
    // lookup); a fingerprint hit is VERIFIED key by key against the stored set before it counts
    struct SetEntry { std::vector<int> keys; std::vector<Stored> list; bool has_root; };
    struct Fp {
        uint64_t a, b;
        size_t n;
        bool operator<(const Fp &o) const { return a != o.a ? a < o.a : b != o.b ? b < o.b : n < o.n; }
    };
    std::map<Fp, std::vector<SetEntry>> sets;
    static Fp fp_of(const StateSet &s) { return Fp{s.hsum, s.hxor, s.items.size()}; }
    static bool same_keys(const StateSet &s, const std::vector<int> &keys) {
        if (keys.size() != s.items.size()) return false;
        for (int k : keys)
            if (s.find(k) < 0) return false;
        return true;
    }
    std::vector<std::vector<int>> closure_cache;          // nfa.epsilonClosure(state) in HashSet iteration order
    std::vector<char> closure_done;

    bool has_accepting(const StateSet &s) const { return s.find(n - 1) >= 0; }
    void store(const StateSet &s, int dfa) {
        std::vector<SetEntry> &bucket = sets[fp_of(s)];
        for (SetEntry &e : bucket)
            if (same_keys(s, e.keys)) {
                e.list.push_back(Stored{s.seen_accepting, dfa});
                return;
            }
        SetEntry ne{s.sorted_keys(), {Stored{s.seen_accepting, dfa}}, false};
        if (root_template_ok) { // (lookup_root_plus: does this set hold the whole closure of the root?)
            ne.has_root = ne.keys.size() >= root_template.items.size();
            for (size_t i = 0; ne.has_root && i < root_template.items.size(); ++i)
                ne.has_root = std::binary_search(ne.keys.begin(), ne.keys.end(), root_template.items[i].key);
        }
        bucket.push_back(std::move(ne));
    }
    int lookup(const StateSet &s) { // getDFA :124-135
        auto it = sets.find(fp_of(s));
        if (it == sets.end()) return -1;
        for (const SetEntry &e : it->second) {
            if (!same_keys(s, e.keys)) continue;
            for (const Stored &p : e.list)
                if (s.items.size() == 1 || s.seen_accepting == p.seen_accepting) return p.dfa;
            return -1;
        }
        return -1;
    }
    const std::vector<int> &nfa_closure(int state) { // NFA.epsilonClosure :239-266
        if (closure_done[state]) return closure_cache[state];
        std::vector<char> seen(n, 0), in_closure(n, 0);
        std::vector<int> items; // closure HashSet, insertion order
        JOrder jo;
        std::vector<int> queue{state};
        for (size_t qi = 0; qi < queue.size(); ++qi) {
            const int next = queue[qi];
            seen[next] = 1;
            const Instr &in = prog[next];
            if (in.op == OP_SPLIT) {
                for (int t : in.targets) if (!seen[t]) queue.push_back(t);
            } else if (in.op == OP_JUMP) {
                if (!seen[in.target]) queue.push_back(in.target);
            } else if (!in_closure[next]) {
                in_closure[next] = 1;
                items.push_back(next);
                jo.on_insert();
            }
        }
        std::vector<int> order, scratch;
        java_order(items, jo.cap, [](int k) { return k; }, order, scratch);
        std::vector<int> out(items.size());
        for (size_t i = 0; i < order.size(); ++i) out[i] = items[order[i]];
        closure_cache[state] = std::move(out);
        closure_done[state] = 1;
        return closure_cache[state];
    }
    // Every non-accepting state of the search automata is epsilon_closure(a closed set + the root, instruction 0), and the root's
    // closure is the same few thousand instructions every time (a dictionary: the first char of every keyword).  Most of those sets
    // exist already; lookup_root_plus finds them by fingerprint without building them.  root_template: the root's closure as a set.
    StateSet root_template{0};
    std::vector<char> in_root_closure;
    bool root_template_ok = false, root_template_tried = false;
    void make_root_template() {
        if (!root_template_tried) {
            root_template_tried = true;
            const std::vector<int> &rc = nfa_closure(0);
            in_root_closure.assign(n, 0);
            root_template_ok = true;
            for (int e : rc) {
                in_root_closure[e] = 1;
                if (e == n - 1) root_template_ok = false; // (a nullable pattern: the general path keeps seen_accepting right)
                root_template.add(e, 0, prog[0].priority);
            }
            if (!rc.empty() && rc.size() == 1 && rc[0] == 0) root_template_ok = false; // (instruction 0 is a CHAR: nothing to gain)
        }
    }
    // getDFA for epsilon_closure(states), states = a closed set + the root just added, WITHOUT building that closure: as a SET it
    // is the root's closure plus the items of `states` that are not in it (iteration order plays no part in set equality), so its
    // fingerprint is the template's plus those few items'; a stored set of that size which holds the root's whole closure and every
    // one of the items IS that set.  -1: not there yet (the caller builds it, the exact way); -2: the shortcut does not apply.
    int lookup_root_plus(const StateSet &states) {
        make_root_template();
        if (!root_template_ok) return -2;
        Fp fp{root_template.hsum, root_template.hxor, root_template.items.size()};
        for (const SItem &it : states.items) {
            if (it.key == 0 || in_root_closure[it.key]) continue;
            if (prog[it.key].op == OP_SPLIT || prog[it.key].op == OP_JUMP || it.key == n - 1) return -2;
            const uint64_t m = StateSet::mix((uint64_t)it.key);
            fp.a += m;
            fp.b ^= StateSet::mix(m);
            ++fp.n;
        }
        auto it = sets.find(fp);
        if (it == sets.end()) return -1;
        for (const SetEntry &e : it->second) {
            if (!e.has_root || e.keys.size() != fp.n) continue;
            bool all = true;
            for (const SItem &x : states.items)
                if (x.key != 0 && !std::binary_search(e.keys.begin(), e.keys.end(), x.key)) { all = false; break; }
            if (!all) continue;
            for (const Stored &p : e.list)
                if (fp.n == 1 || !p.seen_accepting) return p.dfa; // (the set's seen_accepting is false here: no MATCH beside the root)
            return -1;
        }
        return -1;
    }
    StateSet epsilon_closure(const StateSet &states) { // getEpsilonClosure :137-155
        StateSet c(n);
        std::vector<int> order, scratch;
        java_order(states.items, states.jo.cap, [](const SItem &s) { return s.key; }, order, scratch);
        for (int i : order) {
            const SItem &it = states.items[i];
            const int prio = prog[it.key].priority;
            for (int e : nfa_closure(it.key)) {
                if (e == n - 1) c.seen_accepting = true;
                const int j = states.find(e);
                c.add(e, j >= 0 ? states.items[j].dist : it.dist, prio);
            }
        }
        c.seen_accepting = c.seen_accepting || states.seen_accepting;
        return c;
    }
};

void prune_dead(Dfa &d) { // DFA.pruneDeadStates :745-792
    const int n = (int)d.st.size();
    std::vector<char> live(n, 0);
    live[0] = 1;
    for (int i = 0; i < n; ++i) if (d.st[i].accepting) live[i] = 1;
    // reverse reachability from the live seeds
    std::vector<std::vector<int>> pred(n);
    for (int i = 0; i < n; ++i) for (const Trans &t : d.st[i].tr) pred[t.target].push_back(i);
    std::vector<int> work;
    for (int i = 0; i < n; ++i) if (live[i]) work.push_back(i);
    while (!work.empty()) {
        const int s = work.back();
        work.pop_back();
        for (int p : pred[s]) if (!live[p]) { live[p] = 1; work.push_back(p); }
    }
    std::vector<int> renum(n, -1);
    int k = 0;
    for (int i = 0; i < n; ++i) if (live[i]) renum[i] = k++;
    Dfa out;
    out.st.resize(k);
    for (int i = 0; i < n; ++i) {
        if (!live[i]) continue;
        DState &o = out.st[renum[i]];
        o.accepting = d.st[i].accepting;
        for (const Trans &t : d.st[i].tr) if (live[t.target]) o.tr.push_back(Trans{t.start, t.end, renum[t.target]});
    }
    d = std::move(out);
}

Dfa minimize(const Dfa &d) { // MinimizeDFA.java
    const int n = (int)d.st.size();
    std::vector<int> block(n);
    { // initial partition: accepting x transition count x sum of range starts :109-145 (any finer split is implied)
        std::map<std::tuple<bool, size_t, long>, int> ids;
        for (int i = 0; i < n; ++i) {
            long total = 0;
            for (const Trans &t : d.st[i].tr) total += t.start;
            auto key = std::make_tuple(d.st[i].accepting, d.st[i].tr.size(), total);
            auto it = ids.find(key);
            if (it == ids.end()) it = ids.emplace(key, (int)ids.size()).first;
            block[i] = it->second;
        }
    }
    for (;;) { // refine until stable: equivalent() :201-217 = same range list, targets in the same block
        std::map<std::vector<int>, int> ids;
        std::vector<int> nb(n);
        for (int i = 0; i < n; ++i) {
            std::vector<int> sig;
            sig.reserve(2 + 3 * d.st[i].tr.size());
            sig.push_back(block[i]);
            for (const Trans &t : d.st[i].tr) { sig.push_back(t.start); sig.push_back(t.end); sig.push_back(block[t.target]); }
            auto it = ids.find(sig);
            if (it == ids.end()) it = ids.emplace(std::move(sig), (int)ids.size()).first;
            nb[i] = it->second;
        }
        int before = 0, after = (int)ids.size();
        { std::set<int> s(block.begin(), block.end()); before = (int)s.size(); }
        block.swap(nb);
        if (after == before) break;
    }
    // number blocks in first-encounter order: each original state, then its transition targets :23-30
    std::vector<int> id_of_block(n, -1);
    std::vector<int> rep_state;
    auto get = [&](int s) {
        int &id = id_of_block[block[s]];
        if (id < 0) { id = (int)rep_state.size(); rep_state.push_back(s); }
        return id;
    };
    for (int i = 0; i < n; ++i) {
        get(i);
        for (const Trans &t : d.st[i].tr) get(t.target);
    }
    Dfa out;
    out.st.resize(rep_state.size());
    for (size_t b = 0; b < rep_state.size(); ++b) {
        const DState &src = d.st[rep_state[b]];
        out.st[b].accepting = src.accepting;
        for (const Trans &t : src.tr) add_transition(out.st[b], CR{t.start, t.end}, id_of_block[block[t.target]]);
    }
    return out;
}

// ------------------------------------------------------------------------------------------------ char classes + tables
struct ByteClasses { std::vector<uint8_t> map; int count = 0; bool ok = false; };

ByteClasses byte_classes(const Dfa &d) { // DFA.byteClasses :438-463 over the search DFA
    // getSortedTransitions :561-566
    std::vector<CR> all;
    for (const DState &s : d.st) for (const Trans &t : s.tr) all.push_back(CR{t.start, t.end});
    std::sort(all.begin(), all.end(), cr_less);
    all.erase(std::unique(all.begin(), all.end()), all.end());
    // getDistinctCharRanges :509-546
    std::vector<CR> distinct;
    {
        int next_start = 0;
        bool done = false;
        for (size_t i = 0; i < all.size() && !done; ++i) {
            const CR left = all[i];
            next_start = std::max(next_start, left.start);
            while (next_start <= left.end) {
                int next_end = left.end;
                for (size_t j = i + 1; j < all.size(); ++j) {
                    const CR right = all[j];
                    if (right.end < next_start) continue;
                    if (right.start > next_end) break;
                    if (next_start >= right.start) next_end = std::min(next_end, right.end);
                    else next_end = right.start - 1;
                }
                distinct.push_back(CR{next_start, next_end});
                if (next_end == 0xFFFF) { done = true; break; }
                next_start = next_end + 1;
            }
        }
    }
    // charRanges :484-500 + generateRangeGroups :465-482
    std::map<std::vector<std::pair<int, int>>, std::vector<CR>> groups;
    for (const CR &r : distinct) {
        std::vector<std::pair<int, int>> sig;
        for (size_t s = 0; s < d.st.size(); ++s)
            for (const Trans &t : d.st[s].tr)
                if (t.start <= r.end && t.end >= r.start) sig.emplace_back((int)s, t.target);
        std::sort(sig.begin(), sig.end());
        sig.erase(std::unique(sig.begin(), sig.end()), sig.end());
        groups[sig].push_back(r);
    }
    std::vector<std::vector<CR>> gl;
    for (auto &kv : groups) { std::sort(kv.second.begin(), kv.second.end(), cr_less); gl.push_back(kv.second); }
    std::sort(gl.begin(), gl.end(), [](const std::vector<CR> &a, const std::vector<CR> &b) { // RangeGroup.compareTo
        for (size_t i = 0; i < a.size(); ++i) {
            if (i >= b.size()) return false;
            if (!(a[i] == b[i])) return cr_less(a[i], b[i]);
        }
        return a.size() < b.size();
    });
    ByteClasses bc;
    bc.map.assign(65536, 0);
    int cls = 1;
    for (const auto &g : gl) {
        for (const CR &r : g) {
            const int hi = std::min(r.end + 1, 65535); // Arrays.fill(.., min(end+1, 65535), ..): U+FFFF stays class 0
            for (int c = r.start; c < hi; ++c) bc.map[c] = (uint8_t)cls;
        }
        ++cls;
        if (cls > 255) return bc; // Optional.empty()
    }
    bc.count = cls & 0xFF;
    bc.ok = true;
    return bc;
}

int effective_class_count(int c) { // DFAClassBuilder.getEffectiveByteClassCount :240-253
    if (c > 16) return c;
    if (c < 3) return c;
    if (c < 4) return 4;
    if (c < 8) return 8;
    if (c < 16) return 16;
    return c;
}

void fill_ref_dfa(const Dfa &d, const ByteClasses &bc, int stride, RefDfa &out) {
    const int n = (int)d.st.size();
    out.n_states = n;
    out.table.assign((size_t)n * stride, (int16_t)-1);
    out.accepting.assign(n, 0);
    int max_char = 0;
    // runs of equal class over 0..0xFFFF
    std::vector<int> run_end(65536);
    for (int c = 65535, e = 65535; c >= 0; --c) {
        if (c < 65535 && bc.map[c] != bc.map[c + 1]) e = c;
        run_end[c] = e;
    }
    std::vector<char> seen(256);
    for (int s = 0; s < n; ++s) {
        out.accepting[s] = d.st[s].accepting ? 1 : 0;
        std::fill(seen.begin(), seen.end(), 0);
        for (const Trans &t : d.st[s].tr) { // DFAStateTransitions.buildByteClassString :30-62 (first class hit wins)
            max_char = std::max(max_char, std::max(t.start, t.end));
            for (int c = t.start; c <= t.end;) {
                const int k = bc.map[c];
                if (!seen[k]) { seen[k] = 1; out.table[(size_t)s * stride + k] = (int16_t)t.target; }
                c = std::min(run_end[c], t.end) + 1;
            }
        }
    }
    out.max_char = max_char;
}

Dfa build(const std::vector<Instr> &prog, ConvMode mode) { // NFAToDFACompiler.compile :24-32
    static const bool timing = getenv("NEEDLE_COMPILE_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    SubsetBuilder sb(prog);
    Dfa d = sb.compile(mode);
    const double t1 = now();
    prune_dead(d);
    const double t2 = now();
    Dfa m = minimize(d);
    if (timing) fprintf(stderr, "[compile]   subset %.3f s (%zu states)  prune %.3f s  minimise %.3f s (%zu states)\n", t1 - t0, d.st.size(), t2 - t1, now() - t2, m.st.size());
    return m;
}

} // namespace

int compile_regex(const std::u16string &regex, int flags, RefTables &out, std::string &err) {
    try {
        NodeP ast = Parser(regex, flags).parse();
        const int min_len = min_length(ast);
        const long max_len = max_length(ast);
        const bool lml = (flags & NEEDLE_LEFTMOST_LONGEST) != 0;
        const std::vector<Instr> fwd = ProgramBuilder(lml).build(ast);
        const std::vector<Instr> rev = ProgramBuilder(lml).build(reversed(ast));
        if (getenv("NEEDLE_DEBUG_NFA")) { // developer aid: dump the forward Thompson program
            for (size_t i = 0; i < fwd.size(); ++i) {
                const Instr &in = fwd[i];
                fprintf(stderr, "%3zu: %s", i, in.op == OP_CHAR ? "CHAR" : in.op == OP_JUMP ? "JUMP" : in.op == OP_SPLIT ? "SPLIT" : "MATCH");
                if (in.op == OP_CHAR) fprintf(stderr, " %04x-%04x", in.start, in.end);
                if (in.op == OP_JUMP) fprintf(stderr, " ->%d", in.target);
                for (int t : in.targets) fprintf(stderr, " %d", t);
                fprintf(stderr, "  prio=%d\n", in.priority);
            }
        }
        Dfa dfas[4];
        static const bool timing = getenv("NEEDLE_COMPILE_TIMING") != nullptr; // developer aid: where a big dictionary's compile time goes
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double t0 = now();
        auto lap = [&](const char *what) {
            if (timing) fprintf(stderr, "[compile] %-14s %.3f s\n", what, now() - t0);
            t0 = now();
        };
        // the four automata are independent: the two search-mode ones (each state set carries the root's closure: the heavy ones
        // for a dictionary) are built side by side
        // (only for big programs -- dictionaries: a small regex compiles in microseconds and gains nothing from a thread; and a thread that
        // cannot be created -- EAGAIN in a restricted container -- must not turn a compile that works sequentially into an error)
        std::exception_ptr err_ci;
        std::thread t_ci;
        bool threaded = false;
        if (fwd.size() >= 512) {
            try {
                t_ci = std::thread([&] {
                    try {
                        dfas[W_CONTAINED_IN] = build(fwd, CONTAINED_IN);
                    } catch (...) {
                        err_ci = std::current_exception();
                    }
                });
                threaded = true;
            } catch (const std::system_error &) {
                threaded = false;
            }
        }
        try {
            dfas[W_MATCHES] = build(fwd, BASIC);
            dfas[W_BACKWARDS] = build(rev, BASIC);
            dfas[W_FORWARDS] = build(fwd, DFA_SEARCH);
            if (!threaded) dfas[W_CONTAINED_IN] = build(fwd, CONTAINED_IN);
        } catch (...) {
            if (threaded) t_ci.join();
            throw;
        }
        if (threaded) t_ci.join();
        if (err_ci) std::rethrow_exception(err_ci);
        lap("four automata");
        for (const Dfa &d : dfas) // DFACompiler.checkForOverLongDFAs :76-83
            if ((int)d.st.size() > 16383) throw CompileError("Can't compile DFAs with more than 16383 states");
        const ByteClasses bc = byte_classes(dfas[W_FORWARDS]); // DFAClassBuilder.java:66-76: classes of dfaSearch for all four
        if (!bc.ok) throw Unsupported("search DFA needs more than 255 char classes: no table form (DFA.java:454-456)");
        out.class_map = bc.map;
        out.stride = effective_class_count(bc.count);
        for (int w = 0; w < 4; ++w) fill_ref_dfa(dfas[w], bc, out.stride, out.dfa[w]);
        out.min_len = min_len;
        out.max_len = max_len < 0 ? -1 : (int)std::min<long>(max_len, INT_MAX);
        out.fixed_len = (max_len >= 0 && (long)min_len == max_len) ? min_len : -1; // Factorization.canOnlyHaveOneLength :376-378
        return NEEDLE_OK;
    } catch (const SyntaxError &e) {
        err = std::string("PatternSyntaxException: ") + e.what();
        return NEEDLE_ERR_SYNTAX;
    } catch (const Unsupported &e) {
        err = e.what();
        return NEEDLE_ERR_UNSUPPORTED;
    } catch (const CompileError &e) {
        err = std::string("PatternClassCompilationException: ") + e.what();
        return NEEDLE_ERR_COMPILE;
    } catch (const std::exception &e) {
        err = std::string("PatternClassCompilationException: ") + e.what();
        return NEEDLE_ERR_COMPILE;
    }
}

} // namespace needle
