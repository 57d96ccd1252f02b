// needle_ngram.h -- the chain-free n-gram candidate filter, device side.
//
// The reference narrows where its DFA has to run at all: `indexOf(prefix)` ahead of the walk (DFAClassBuilder.java:365-376),
// the first-byte mask pre-scan (:420-426, :508-511; mask from DFA.initialAsciiBytes, DFA.java:706-726), gated by
// CompilationPolicy.java:44-57 over Factorization.getPrefixes() (Factorization.java:116).  Those are skip loops for a scalar
// CPU.  The table-level generalisation built here (host side: needle_ngram_host.cpp) reads, off the automaton's own table, every
// 4-byte window that can stand `o` chars ahead of an accepting transition (o = 0 .. S-1) and hashes them into a bitmap that
// is staged in LDS next to the automaton.  The kernel tests one window every S chars -- S = 2: half of them are aligned
// dwords of the text, the others one v_alignbit away -- with no dependence between chars: a dot product of the window's two
// halves with the multipliers (v_dot2_u32_u16), an and-or that makes the LDS address, the read, a shift by the hash's top byte.  Only where a window passes does the automaton run, from K chars before the window's end (K: the depth after
// which the automaton has forgotten where it was started, verified on the table by the host), for K + S - 1 chars.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace needle {

// What the device needs to know about a filter (filled by the host analysis, needle_ngram_host.cpp).
struct NgramParams {
    uint32_t on;          // 0: no filter for this program
    uint32_t stride;      // S: one window every S chars (1, 2 or 4), row-relative positions q = 0 (mod S)
    uint32_t warm;        // K: chars the automaton is run ahead of a window's end
    uint32_t m1, m2;      // hash multipliers (16 bits each): u = (x & 0xFFFF) * m1 + (x >> 16) * m2 (mod 2^32) -- one v_dot2_u32_u16
    uint32_t addr_shift;  // the window's two bits of the word: (u >> addr_shift) & 31 and (u >> (addr_shift - 8)) & 31; addr_shift = 24: the shift
                          // amounts are u's bytes 3 and 2 as they stand (SDWA)
    uint32_t addr_mask;   // byte offset of the word inside the bitmap = u & addr_mask (the bitmap's LDS base is a multiple of its size:
                          // base | offset is the address -- one v_and_or_b32)
    uint32_t bm_bytes;    // bitmap size (power of two)
    uint32_t min_len;     // shortest accepted string (informational)
    uint32_t n_grams;     // distinct byte windows in the bitmap (informational)
    // Second level (find / containedIn): a candidate whose 4-byte window passed is checked once more before the automaton runs --
    // the 5-byte window [qn - 5, qn) that ends where its window ends, hashed u2 = u + byte(qn - 5) * m3 into a second, smaller bitmap
    // (same two-bits-of-one-word scheme, addr_shift 24).  On random text that drops 26 of 27 chance hits of the first level; what is
    // left is mostly real keyword tails, and only they cost a run of the automaton.
    uint32_t on2;         // 0: no second level; 1: as above; 2: TWO-SIDED -- for patterns whose shortest match is one char too short for 1 (min_len =
                          // 5 + S - 2): the second bitmap holds the 5-char windows ending o = 0 .. S - 2 chars ahead of a first accept, a candidate
                          // passes when the 5 chars ending AT its window's end or the 5 chars ending one char BEHIND it are in it (a window S - 1
                          // ahead of the accept has its match's next char behind it, the others the 5th char in front: needle_ngram_host.cpp)
    uint32_t m3;          // 24-bit multiplier of the fifth byte
    uint32_t addr_mask2;  // byte offset of the word inside the second bitmap = u2 & addr_mask2
    uint32_t bm2_bytes;   // its size (power of two; it follows the first bitmap in the device buffer)
    uint32_t n_grams2;    // distinct 5-byte windows (informational)
    // WIDE filter (UTF-16 rows of a pattern on several pages of the BMP): a window is four 16-bit code units = two dwords x0 (units 0, 1)
    // and x1 (units 2, 3); u = dot2(x0, m1 | m2 << 16) + dot2(x1, m1b | m2b << 16) -- two v_dot2_u32_u16, the second one accumulating;
    // the second level adds unit(q - 5) * m3.  Everything else (bitmaps, bits, queues) as for bytes.
    uint32_t wide, m1b, m2b;
};

// LDS of the filter kernel (needle_ngram.hip), per wave: the candidate queue + one u64 result slot per row of the group
constexpr uint32_t kNgQueue = 128;                      // candidates a wave can hold (at most 63 wait while 64 more arrive)
constexpr uint32_t kNgWaveLds = 2 * kNgQueue * 4 + 64 * 8; // (two queues: first-level candidates, second-level survivors)
// ... of its find-all form: the queue + per row of the group two slots for verified candidates (8 bytes each) and a counter
constexpr uint32_t kNgRowSlots = 2;
constexpr uint32_t kNgWaveLdsFA = kNgQueue * 4 + 64 * kNgRowSlots * 8 + 64 * 4;
constexpr uint32_t kNgWaves = 16;
constexpr uint32_t kNgLdsCap = 160u * 1024u;

// Where the bitmap and the waves' queues sit behind a program of prog_bytes: the bitmap at the next multiple of its own size
// (its address bits and the hash's do not overlap), the queues in the gap in front of it when they fit there, else behind it.
struct NgramLayout {
    uint32_t bm_base, q_base, total, bm2_base;
};
// bm2_bytes != 0: the second-level bitmap behind everything else, at a multiple of its own size
inline bool ngram_layout(uint32_t prog_bytes, uint32_t bm_bytes, NgramLayout *out, uint32_t wave_bytes = kNgWaveLds, uint32_t bm2_bytes = 0) {
    if (bm_bytes < 4096u || (bm_bytes & (bm_bytes - 1u))) return false;
    const uint32_t p = (prog_bytes + 15u) & ~15u, qb = kNgWaves * wave_bytes;
    NgramLayout l;
    l.bm_base = (p + bm_bytes - 1u) & ~(bm_bytes - 1u);
    if (l.bm_base - p >= qb) l.q_base = p, l.total = l.bm_base + bm_bytes;
    else l.q_base = l.bm_base + bm_bytes, l.total = l.q_base + qb;
    l.bm2_base = 0;
    if (bm2_bytes) {
        if (bm2_bytes < 1024u || (bm2_bytes & (bm2_bytes - 1u))) return false;
        l.bm2_base = (l.total + bm2_bytes - 1u) & ~(bm2_bytes - 1u);
        l.total = l.bm2_base + bm2_bytes;
    }
    if (l.total > kNgLdsCap) return false;
    if (out) *out = l;
    return true;
}

// host mirror of the hash.  A window owns TWO bits of ONE bitmap word (a blocked Bloom filter: one LDS read tests both):
// word (u & addr_mask) >> 2, bits (u >> addr_shift) & 31 and (u >> (addr_shift - 8)) & 31 -- the shift amounts are u's bytes 3
// and 2 as they stand (SDWA).  With ~2000 windows in 8192 words a text window that is not in the set passes with probability
// ~0.12 % (one bit: 0.76 %) -- every pass costs a 16-byte re-read of the row and a share of an automaton run.
inline uint32_t ngram_hash_host(uint32_t x, uint32_t m1, uint32_t m2) { return (x & 0xFFFFu) * m1 + (x >> 16) * m2; }
inline uint32_t ngram_hash16_host(uint32_t x0, uint32_t x1, uint32_t m1, uint32_t m2, uint32_t m1b, uint32_t m2b) {
    return (x0 & 0xFFFFu) * m1 + (x0 >> 16) * m2 + (x1 & 0xFFFFu) * m1b + (x1 >> 16) * m2b;
}
inline uint32_t ngram_word_index(uint32_t u, uint32_t addr_mask) { return (u & addr_mask) >> 2; }
inline uint32_t ngram_word_bits(uint32_t u, uint32_t addr_shift) { return 1u << ((u >> addr_shift) & 31u) | 1u << ((u >> (addr_shift - 8u)) & 31u); }

// Bit 0 of the result: window x has both its bits in the bitmap.  Hash, address, LDS read, two shifts, an and: five VALU ops.
// m = m1 | m2 << 16; bm_base: a multiple of the bitmap's size (ngram_layout).
__device__ __forceinline__ uint32_t ngram_probe(uint32_t x, uint32_t m, uint32_t addr_mask, uint32_t bm_base) {
    uint32_t u, r1, r2;
    asm("v_dot2_u32_u16 %0, %1, %2, 0" : "=v"(u) : "v"(x), "v"(m));
    // (u & addr_mask) | bm_base as ONE instruction: left to the compiler the two wave-uniform operands sit in SGPRs, of which a VOP3 on
    // gfx9 may read one -- it emits v_and + v_or; the base in a VGPR (loop-invariant: one v_mov per kernel) makes v_and_or_b32 legal
    uint32_t a;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(a) : "v"(u), "s"(addr_mask), "v"(bm_base));
    const uint32_t w = *(__attribute__((address_space(3))) const uint32_t *)(uintptr_t)a;
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r1) : "v"(u), "v"(w)); // w >> (u >> 24 & 31)
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r2) : "v"(u), "v"(w)); // w >> (u >> 16 & 31)
    return r1 & r2;
}

// Second level: window x (4 bytes) + the byte c5 in front of it.  u2 = hash(x) + c5 * m3; same word / two-bit test in the second bitmap.
inline uint32_t ngram_hash2_host(uint32_t x, uint32_t c5, uint32_t m1, uint32_t m2, uint32_t m3) { return ngram_hash_host(x, m1, m2) + c5 * m3; }
__device__ __forceinline__ uint32_t ngram_probe2(uint32_t x, uint32_t c5, uint32_t m, uint32_t m3, uint32_t addr_mask2, uint32_t bm2_base) {
    uint32_t u;
    asm("v_dot2_u32_u16 %0, %1, %2, 0" : "=v"(u) : "v"(x), "v"(m));
    u += c5 * m3;
    const uint32_t a = (u & addr_mask2) | bm2_base;
    const uint32_t w = *(__attribute__((address_space(3))) const uint32_t *)(uintptr_t)a;
    return (w >> ((u >> 24) & 31u)) & (w >> ((u >> 16) & 31u)) & 1u;
}

// WIDE second level: the window's two dwords x0, x1 + the code unit c5 in front of them.
__device__ __forceinline__ uint32_t ngram_probe2_16(uint32_t x0, uint32_t x1, uint32_t c5, uint32_t mA, uint32_t mB, uint32_t m3, uint32_t addr_mask2,
                                                    uint32_t bm2_base) {
    uint32_t u;
    asm("v_dot2_u32_u16 %0, %1, %2, 0" : "=v"(u) : "v"(x0), "v"(mA));
    asm("v_dot2_u32_u16 %0, %1, %2, %3" : "=v"(u) : "v"(x1), "v"(mB), "v"(u));
    u += c5 * m3;
    const uint32_t a = (u & addr_mask2) | bm2_base;
    const uint32_t w = *(__attribute__((address_space(3))) const uint32_t *)(uintptr_t)a;
    return (w >> ((u >> 24) & 31u)) & (w >> ((u >> 16) & 31u)) & 1u;
}

// One 16-byte piece of text held by one lane (w0 .. w3; pw = the dword before it: the previous lane's w3).  Tests the windows
// that END in this piece at the sampled positions -- S = 2: the windows starting at byte -2, 0, 2, .. 12 of the piece -- and
// shifts their verdicts into `log` from the top (v_alignbit): afterwards bit 31 = the last window of this piece, bit 32 - n = its
// first one (n = 16 / S windows), and whatever the log held before sits n bits further down.
// Three phases, so that ONE LDS round trip is exposed per piece instead of one per window or two: all hashes and addresses, all
// bitmap reads back to back, one wait, then the shifts (left to the scheduler the first reads of a piece are each waited for at once).
template <int S>
__device__ __forceinline__ uint32_t ngram_piece(uint32_t log, uint32_t pw, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t m,
                                                uint32_t addr_mask, uint32_t bm_base) {
    constexpr int NWIN = 16 / S;
    uint32_t x[NWIN];
    if (S == 4) { // windows ending at byte 4, 8, 12, 16: the piece's own dwords
        x[0] = w0, x[1] = w1, x[2] = w2, x[3] = w3;
    } else if (S == 2) { // ending at byte 2, 4, .. 16
        const uint32_t q[5] = {pw, w0, w1, w2, w3};
#pragma unroll
        for (int i = 0; i < 4; ++i) x[2 * i] = __builtin_amdgcn_alignbit(q[i + 1], q[i], 16), x[2 * i + 1] = q[i + 1];
    } else { // S == 1: ending at byte 1 .. 16
        const uint32_t q[5] = {pw, w0, w1, w2, w3};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[4 * i] = __builtin_amdgcn_alignbit(q[i + 1], q[i], 8), x[4 * i + 1] = __builtin_amdgcn_alignbit(q[i + 1], q[i], 16);
            x[4 * i + 2] = __builtin_amdgcn_alignbit(q[i + 1], q[i], 24), x[4 * i + 3] = q[i + 1];
        }
    }
    uint32_t u[NWIN], wv[NWIN];
#pragma unroll
    for (int i = 0; i < NWIN; ++i) {
        asm("v_dot2_u32_u16 %0, %1, %2, 0" : "=v"(u[i]) : "v"(x[i]), "v"(m));
        uint32_t a;
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(a) : "v"(u[i]), "s"(addr_mask), "v"(bm_base)); // (one SGPR per VOP3: the base rides in a VGPR)
        wv[i] = *(__attribute__((address_space(3))) const uint32_t *)(uintptr_t)a;
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): all of the piece's bitmap words
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NWIN; ++i) {
        uint32_t r1, r2;
        asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r1) : "v"(u[i]), "v"(wv[i])); // w >> (u >> 24 & 31)
        asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r2) : "v"(u[i]), "v"(wv[i])); // w >> (u >> 16 & 31)
        log = __builtin_amdgcn_alignbit(r1 & r2, log, 1);
    }
    return log;
}

// The WIDE form: 16 UTF-16 code units held by one lane as eight dwords (lo, hi; pw = the dword before them: the previous lane's hi[3]).  A
// window is four code units = two dwords (x0, x1), u = dot2(x0, mA) + dot2(x1, mB): with S = 2 every window is a pair of adjacent dwords and
// each dword's first product serves the next window -- two v_dot2_u32_u16 per dword, no v_alignbit; S = 4: the pairs (0, 1), (2, 3), ...
// Same log layout as ngram_piece: 16 / S verdicts shifted in from the top, the piece's last window at bit 31.
typedef uint32_t ng_u32x4_t __attribute__((ext_vector_type(4)));
template <int S>
__device__ __forceinline__ uint32_t ngram_piece16(uint32_t log, uint32_t pw, const ng_u32x4_t &lo, const ng_u32x4_t &hi, uint32_t mA, uint32_t mB,
                                                  uint32_t addr_mask, uint32_t bm_base) {
    static_assert(S == 2 || S == 4, "one window every 2 or 4 code units");
    constexpr int NWIN = 16 / S;
    const uint32_t d[9] = {pw, lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; // d[i + 1] = dword i of the 16 units
    uint32_t u[NWIN], wv[NWIN];
#pragma unroll
    for (int i = 0; i < NWIN; ++i) {
        const uint32_t x0 = S == 2 ? d[i] : d[2 * i + 1], x1 = S == 2 ? d[i + 1] : d[2 * i + 2];
        uint32_t t;
        asm("v_dot2_u32_u16 %0, %1, %2, 0" : "=v"(t) : "v"(x0), "v"(mA));
        asm("v_dot2_u32_u16 %0, %1, %2, %3" : "=v"(u[i]) : "v"(x1), "v"(mB), "v"(t));
        uint32_t a;
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(a) : "v"(u[i]), "s"(addr_mask), "v"(bm_base));
        wv[i] = *(__attribute__((address_space(3))) const uint32_t *)(uintptr_t)a;
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): all of the piece's bitmap words
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NWIN; ++i) {
        uint32_t r1, r2;
        asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r1) : "v"(u[i]), "v"(wv[i]));
        asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r2) : "v"(u[i]), "v"(wv[i]));
        log = __builtin_amdgcn_alignbit(r1 & r2, log, 1);
    }
    return log;
}

// ---- UTF-16 rows (Java's strings) behind the byte filter.  Where the pattern lives on ONE page P of the BMP -- every char outside the
// page is of the pattern's "other" class, and so is the page's char P << 8 | sub (needle_api.cpp utf16_route) -- a char outside the page
// behaves like byte `sub` of the program lowered from the tables rebased to the page.  So the filter kernel NARROWS the text as it loads
// it -- 16 chars = 32 bytes -> 16 bytes, the low bytes picked by four v_perm_b32 -- and everything behind the load (windows, bitmaps,
// queues, the byte program's walks, positions in chars) is the 8-bit kernel.  Chars outside the page become `sub`.  Page 0 (ASCII /
// Latin-1 patterns: sub = 0xFF, "beyond maxChar" -- DFAClassBuilder.java:440, :565 -- for the pattern as for its byte program): the high
// bytes are or-ed together (v_or3_b32) and tested once per 16 chars, the per-byte patch runs only for the loads where some lane holds a
// char above 0xFF.  Other pages (Cyrillic, Greek, Hebrew, ...: their text is full of spaces and digits of page 0): always the patch.
// page4 / sub4: the page's / the substitute's byte in all four bytes of a dword.
__device__ __forceinline__ uint32_t narrow_pair(uint32_t x, uint32_t y) { // chars {x.lo16, x.hi16, y.lo16, y.hi16} -> their low bytes
    return __builtin_amdgcn_perm(y, x, 0x06040200u);
}
__device__ __forceinline__ uint32_t narrow_pair_patched(uint32_t x, uint32_t y, uint32_t page4 = 0u, uint32_t sub4 = 0xFFFFFFFFu) { // ... chars outside the page -> sub
    const uint32_t lo = __builtin_amdgcn_perm(y, x, 0x06040200u), hi = __builtin_amdgcn_perm(y, x, 0x07050301u) ^ page4;
    uint32_t m = (((hi & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | hi) & 0x80808080u; // bit 7 of every byte whose high byte is not the page's
    m |= m - (m >> 7);                                                      // -> 0xFF there
    return (lo & ~m) | (sub4 & m);                                          // v_bfi_b32
}
typedef uint32_t ng_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ng_u32x4 narrow16(const ng_u32x4 &A, const ng_u32x4 &B, uint32_t page4 = 0u, uint32_t sub4 = 0xFFFFFFFFu) {
    ng_u32x4 o;
    if (page4 != 0u) { // wave-uniform
        o[0] = narrow_pair_patched(A[0], A[1], page4, sub4), o[1] = narrow_pair_patched(A[2], A[3], page4, sub4);
        o[2] = narrow_pair_patched(B[0], B[1], page4, sub4), o[3] = narrow_pair_patched(B[2], B[3], page4, sub4);
        return o;
    }
    o[0] = narrow_pair(A[0], A[1]), o[1] = narrow_pair(A[2], A[3]), o[2] = narrow_pair(B[0], B[1]), o[3] = narrow_pair(B[2], B[3]);
    const uint32_t any_hi = ((A[0] | A[1] | A[2]) | (A[3] | B[0] | B[1]) | (B[2] | B[3])) & 0xFF00FF00u;
    if (__builtin_expect(__ballot(any_hi != 0u) != 0ull, 0)) {
        o[0] = narrow_pair_patched(A[0], A[1], 0u, sub4), o[1] = narrow_pair_patched(A[2], A[3], 0u, sub4);
        o[2] = narrow_pair_patched(B[0], B[1], 0u, sub4), o[3] = narrow_pair_patched(B[2], B[3], 0u, sub4);
    }
    return o;
}

// The dword in front of this lane's piece when lanes hold consecutive pieces: lane l - 1's w3 (DPP wave_shr:1); lane 0 takes
// `carry` (the previous load's lane 63, wave-uniform).
__device__ __forceinline__ uint32_t ngram_prev_dword(uint32_t w3, uint32_t carry) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)w3, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}

} // namespace needle
