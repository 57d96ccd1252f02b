/*
 * TEST INFRASTRUCTURE -- the parity oracle, not product code.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product (needle_amd/) never does.
 *
 * A plain-C restatement of the four loops the reference GENERATES per regex (the JVM bytecode emitted
 * by needle-compiler/src/main/java/com/justinblank/strings/DFAClassBuilder.java), in the
 * "all-byteclass" configuration that DFAClassBuilder.java:66-76 always selects when
 * DFA.byteClasses() is present, and WITHOUT the CPU-only prefilters (prefix/suffix/infix indexOf,
 * first-byte mask, predicate seek, offset look-ahead, maxStart), which do not change results.
 *
 *   ndl_matches          <- createMatchesMethod            DFAClassBuilder.java:854-912
 *   ndl_contained_in     <- createContainedInMethod        DFAClassBuilder.java:956-1025
 *   ndl_index_forwards   <- createIndexMethod              DFAClassBuilder.java:335-471
 *   ndl_index_backwards  <- createIndexMethodReversed      DFAClassBuilder.java:529-586
 *   ndl_find             <- createFindMethodInternal       DFAClassBuilder.java:625-659
 *   accepted()           <- addWasAcceptedMethod           DFAClassBuilder.java:701-720
 *
 * Pinned against the reference's own compiled output: tests/test_oracle_snapshots.py replays the
 * vectors recorded by interpreting needle-compiler/src/test/resources/snapshots/<Name>.class
 * (tests/golden/snapshots/<Name>.json) through these functions and requires identical results.
 *
 * Tables are exactly what the generated class holds: BYTE_CLASSES (char -> class, byte[65537], entry
 * 65536 unused here), STATES_X flat [state * N + class] with -1 = no transition
 * (populateByteClassArrays :317-333 + ByteClassUtil.fillMultipleByteClassesFromString*_singleArray,
 * needle-types/.../ByteClassUtil.java:50-120), widened to int16 for both element types.
 */
#include <stdint.h>
#include <stddef.h>
#include <limits.h>

typedef struct {
    const uint8_t *class_map; /* 65536 entries */
    int32_t stride;           /* N = getEffectiveByteClassCount, DFAClassBuilder.java:240-253 */
    const int16_t *table;     /* n_states * stride, -1 = dead */
    const uint8_t *accepting; /* n_states, 1 = accepting */
    int32_t n_states;
    int32_t max_char;         /* spec.dfa.maxChar(), DFA.java:384-398 */
} ndl_dfa;

#define NDL_WOULD_THROW (-2)

static inline int32_t ch(const void *s, int w, int64_t i) {
    return w == 1 ? (int32_t)((const uint8_t *)s)[i] : (int32_t)((const uint16_t *)s)[i];
}

/* wasAccepted<X>(state): one accepting state -> `state == k`; several -> `state != -1 && ARRAY[state]`. */
static inline int accepted(const ndl_dfa *d, int32_t state) {
    return state >= 0 && d->accepting[state];
}

static inline int32_t step(const ndl_dfa *d, int32_t state, int32_t c) {
    return d->table[d->class_map[c] + state * d->stride];
}

/* DFAClassBuilder.java:892-910 */
int ndl_matches(const ndl_dfa *d, const void *s, int w, int32_t length) {
    int32_t index = 0, state = 0;
    while (state != -1) {
        if (index == length) return accepted(d, state);
        int32_t c = ch(s, w, index);
        if (c > d->max_char) return 0; /* :899-901, emitted unconditionally */
        state = step(d, state, c);
        index++;
    }
    return accepted(d, state);
}

/* DFAClassBuilder.java:971-1022 (no prefix, no maxStart) */
int ndl_contained_in(const ndl_dfa *d, const void *s, int w, int32_t length) {
    int32_t index = 0, state = 0;
    while (index < length) {
        state = 0;
        while (index < length && state != -1) {
            if (accepted(d, state)) return 1;
            int32_t c = ch(s, w, index);
            if (d->max_char < 0xFFFF && c > d->max_char) { /* :1012-1016 */
                state = 0;
                index++;
                break;
            }
            state = step(d, state, c);
            index++;
        }
    }
    return accepted(d, state);
}

/* DFAClassBuilder.java:355-468 with the plain `state = 0` outer body (:428).  The generated method
 * overwrites its second parameter with this.length (DFAMethodComponents.setLengthLocalVariable :19-21
 * writes local slot 2), so only `from` is an argument here. */
int32_t ndl_index_forwards(const ndl_dfa *d, const void *s, int w, int32_t length, int32_t from) {
    int32_t index = from, state = 0;
    const int root_accepting = d->accepting[0];
    int32_t last_match = root_accepting ? 0 : -1; /* :356 literal 0, not `from` */
    while (index < length) {
        state = 0;
        while (index < length) {
            if (root_accepting && accepted(d, state)) last_match = index; /* :433,440 */
            int32_t c = ch(s, w, index);
            index++;
            if (d->max_char < 0xFFFF && c > d->max_char) { /* :451-457 */
                state = -1;
                if (last_match > -1) return last_match;
                /* skip(): the generated loop would continue with state == -1 and index the table
                 * out of bounds.  Unreachable for DFA_SEARCH automata. */
                return NDL_WOULD_THROW;
            }
            state = step(d, state, c);
            if (state == -1) return last_match; /* :461 */
            if (accepted(d, state)) last_match = index; /* :464 */
        }
    }
    return last_match;
}

/* DFAClassBuilder.java:536-583; `from` is the generated method's second parameter (its LENGTH var). */
int32_t ndl_index_backwards(const ndl_dfa *d, const void *s, int w, int32_t index, int32_t from) {
    int32_t state = 0;
    int32_t last_match = d->accepting[0] ? from : INT_MAX; /* :543-547 */
    while (index >= from) {
        int32_t c = ch(s, w, index);
        if (d->max_char < 0xFFFF && c > d->max_char) return last_match; /* :573-575 */
        state = step(d, state, c);
        if (state == -1) return last_match;
        if (accepted(d, state)) last_match = index;
        index--;
    }
    return last_match;
}

/* generateSingleCharacterReverseScan, DFAClassBuilder.java:588-614 */
int32_t ndl_index_backwards_single_char(int32_t c0, const void *s, int w, int32_t index, int32_t from) {
    while (index >= from) {
        if (ch(s, w, index) == c0) return index;
        index--;
    }
    return INT_MAX;
}

/* find(FROM, TO) on a fresh Matcher (nextStart == 0): DFAClassBuilder.java:629-657.
 * fixed_len >= 0  <=> Factorization.canOnlyHaveOneLength(): start = end - minLength (:640-646);
 * otherwise start = indexBackwards(end - 1, FROM) (:648-656), or the single-char scan when
 * single_char >= 0.  Returns matched; *start / *end are only written as the reference writes them
 * (end always, start only on a match). */
int ndl_find(const ndl_dfa *fwd, const ndl_dfa *bwd, int32_t fixed_len, int32_t single_char,
             const void *s, int w, int32_t length, int32_t from, int32_t *start, int32_t *end) {
    int32_t index = ndl_index_forwards(fwd, s, w, length, from);
    *end = index;
    if (index == -1) return 0;
    if (index == NDL_WOULD_THROW) return NDL_WOULD_THROW;
    if (fixed_len >= 0) *start = index - fixed_len;
    else if (single_char >= 0) *start = ndl_index_backwards_single_char(single_char, s, w, index - 1, from);
    else *start = ndl_index_backwards(bwd, s, w, index - 1, from);
    return 1;
}

/* ---- batch drivers (rows at a fixed stride, optional per-row lengths); OpenMP over rows -------- */

static inline int32_t row_len(const uint32_t *lengths, uint32_t fixed, int64_t r) {
    return lengths ? (int32_t)lengths[r] : (int32_t)fixed;
}

void ndl_batch_matches(const ndl_dfa *d, const void *rows, int w, int64_t n_rows, int64_t stride_chars,
                       const uint32_t *lengths, uint32_t fixed_len_chars, uint8_t *out, int threads) {
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t r = 0; r < n_rows; r++)
        out[r] = (uint8_t)ndl_matches(d, (const char *)rows + r * stride_chars * w, w, row_len(lengths, fixed_len_chars, r));
}

void ndl_batch_contained_in(const ndl_dfa *d, const void *rows, int w, int64_t n_rows, int64_t stride_chars,
                            const uint32_t *lengths, uint32_t fixed_len_chars, uint8_t *out, int threads) {
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t r = 0; r < n_rows; r++)
        out[r] = (uint8_t)ndl_contained_in(d, (const char *)rows + r * stride_chars * w, w, row_len(lengths, fixed_len_chars, r));
}

/* unmatched rows: matched = 0, start = end = -1 (the batch convention of include/needle_hip.h). */
void ndl_batch_find(const ndl_dfa *fwd, const ndl_dfa *bwd, int32_t fixed_len, int32_t single_char,
                    const void *rows, int w, int64_t n_rows, int64_t stride_chars, const uint32_t *lengths,
                    uint32_t fixed_len_chars, uint8_t *matched, int32_t *start, int32_t *end, int threads) {
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t r = 0; r < n_rows; r++) {
        int32_t st = -1, en = -1;
        int m = ndl_find(fwd, bwd, fixed_len, single_char, (const char *)rows + r * stride_chars * w, w,
                         row_len(lengths, fixed_len_chars, r), 0, &st, &en);
        if (m != 1) { st = -1; en = -1; }
        matched[r] = (uint8_t)(m == 1);
        start[r] = st;
        end[r] = en;
    }
}
