// needle_ngram.h -- the chain-free n-gram candidate filter, device side.
//
// The reference narrows where its DFA has to run at all: `indexOf(prefix)` ahead of the walk (DFAClassBuilder.java:365-376),
// the first-byte mask pre-scan (:420-426, :508-511; mask from DFA.initialAsciiBytes, DFA.java:706-726), gated by
// CompilationPolicy.java:44-57 over Factorization.getPrefixes() (Factorization.java:116).  Those are skip loops for a scalar
// CPU.  The table-level generalisation built here (host side: needle_ngram_host.cpp) reads, off the automaton's own table, every
// 4-byte window that can stand `o` chars ahead of an accepting transition (o = 0 .. S-1) and hashes them into a bitmap that
// is staged in LDS next to the automaton.  The kernel tests one window every S chars -- S = 2: half of them are aligned
// dwords of the text, the others one v_alignbit away -- with no dependence between chars: two multiply-adds, one LDS read,
// a shift.  Only where a window passes does the automaton run, from K chars before the window's end (K: the depth after
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
    uint32_t m1, m2;      // hash multipliers (24 bits each): u = (x & 0xFFFFFF) * m1 + (x >> 16) * m2
    uint32_t addr_shift;  // word address = ((u >> addr_shift) & addr_mask) + LDS base of the bitmap; bit = u & 31
    uint32_t addr_mask;
    uint32_t bm_bytes;    // bitmap size (power of two)
    uint32_t min_len;     // shortest accepted string (informational)
    uint32_t n_grams;     // distinct byte windows in the bitmap (informational)
};

// u = (x & 0xFFFFFF) * m1 + (x >> 16) * m2 (mod 2^32): two full-rate VALU ops.
__device__ __forceinline__ uint32_t ngram_hash(uint32_t x, uint32_t m1, uint32_t m2) {
    uint32_t t, u;
    asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(t) : "v"(x), "v"(m2));
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(u) : "v"(x), "v"(m1), "v"(t));
    return u;
}

// The bitmap word of window x (LDS read at an absolute address, needle_walk.h) shifted so that bit 0 is the window's bit.
__device__ __forceinline__ uint32_t ngram_probe(uint32_t x, uint32_t m1, uint32_t m2, uint32_t addr_shift, uint32_t addr_mask, uint32_t bm_base) {
    const uint32_t u = ngram_hash(x, m1, m2);
    const uint32_t a = ((u >> addr_shift) & addr_mask) + bm_base;
    const uint32_t w = *(__attribute__((address_space(3))) const uint32_t *)(uintptr_t)a;
    return w >> (u & 31u); // (v_lshrrev_b32 takes the low five bits of u by itself)
}

// One 16-byte piece of text held by one lane (w0 .. w3; pw = the dword before it: the previous lane's w3).  Tests the windows
// that END in this piece at the sampled positions -- S = 2: the windows starting at byte -2, 0, 2, .. 12 of the piece -- and
// shifts their verdicts into `log` from the top (v_alignbit): afterwards bit 31 = the last window of this piece, bit 32 - n = its
// first one (n = 16 / S windows), and whatever the log held before sits n bits further down.
template <int S>
__device__ __forceinline__ uint32_t ngram_piece(uint32_t log, uint32_t pw, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t m1,
                                                uint32_t m2, uint32_t addr_shift, uint32_t addr_mask, uint32_t bm_base) {
#define NEEDLE_NG(X) log = __builtin_amdgcn_alignbit(ngram_probe((X), m1, m2, addr_shift, addr_mask, bm_base), log, 1);
    if (S == 4) { // windows ending at byte 4, 8, 12, 16: the piece's own dwords
        NEEDLE_NG(w0) NEEDLE_NG(w1) NEEDLE_NG(w2) NEEDLE_NG(w3)
    } else if (S == 2) { // ending at byte 2, 4, .. 16
        NEEDLE_NG(__builtin_amdgcn_alignbit(w0, pw, 16)) NEEDLE_NG(w0)
        NEEDLE_NG(__builtin_amdgcn_alignbit(w1, w0, 16)) NEEDLE_NG(w1)
        NEEDLE_NG(__builtin_amdgcn_alignbit(w2, w1, 16)) NEEDLE_NG(w2)
        NEEDLE_NG(__builtin_amdgcn_alignbit(w3, w2, 16)) NEEDLE_NG(w3)
    } else { // S == 1: ending at byte 1 .. 16
        NEEDLE_NG(__builtin_amdgcn_alignbit(w0, pw, 8)) NEEDLE_NG(__builtin_amdgcn_alignbit(w0, pw, 16)) NEEDLE_NG(__builtin_amdgcn_alignbit(w0, pw, 24)) NEEDLE_NG(w0)
        NEEDLE_NG(__builtin_amdgcn_alignbit(w1, w0, 8)) NEEDLE_NG(__builtin_amdgcn_alignbit(w1, w0, 16)) NEEDLE_NG(__builtin_amdgcn_alignbit(w1, w0, 24)) NEEDLE_NG(w1)
        NEEDLE_NG(__builtin_amdgcn_alignbit(w2, w1, 8)) NEEDLE_NG(__builtin_amdgcn_alignbit(w2, w1, 16)) NEEDLE_NG(__builtin_amdgcn_alignbit(w2, w1, 24)) NEEDLE_NG(w2)
        NEEDLE_NG(__builtin_amdgcn_alignbit(w3, w2, 8)) NEEDLE_NG(__builtin_amdgcn_alignbit(w3, w2, 16)) NEEDLE_NG(__builtin_amdgcn_alignbit(w3, w2, 24)) NEEDLE_NG(w3)
    }
#undef NEEDLE_NG
    return log;
}

// The dword in front of this lane's piece when lanes hold consecutive pieces: lane l - 1's w3 (DPP wave_shr:1); lane 0 takes
// `carry` (the previous load's lane 63, wave-uniform).
__device__ __forceinline__ uint32_t ngram_prev_dword(uint32_t w3, uint32_t carry) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)w3, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}

} // namespace needle
