// Host-side lowering: the reference's per-regex tables -> device "programs" for the HIP kernels.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "needle_device.h"

namespace needle {

enum Which : int { W_MATCHES = 0, W_CONTAINED_IN = 1, W_FORWARDS = 2, W_BACKWARDS = 3 };

// One automaton exactly as the reference's generated class holds it
// (STATES_X flat [state * N + class], -1 = no transition; DFAClassBuilder.java:317-333).
struct RefDfa {
    int32_t n_states = 0;
    int32_t max_char = 0xFFFF;      // spec.dfa.maxChar(), DFA.java:384-398
    std::vector<int16_t> table;     // n_states * stride
    std::vector<uint8_t> accepting; // n_states
};

struct RefTables {
    std::vector<uint8_t> class_map; // 65536 = BYTE_CLASSES[0..65535], DFAClassBuilder.java:269-305
    int32_t stride = 0;             // N
    RefDfa dfa[4];
    int32_t fixed_len = -1;         // Factorization.canOnlyHaveOneLength() ? minLength : -1
    int32_t min_len = -1, max_len = -1;
};

struct Program {
    ProgHeader hdr;
    std::vector<uint8_t> blob;
};

// ByteClassUtil.fillMultipleByteClassesFromString*_singleArray (needle-types/.../ByteClassUtil.java:50-120)
// over an array pre-filled with -1.  Returns false + message on malformed input.
bool decode_table_string(const char *s, int32_t n_states, int32_t stride, std::vector<int16_t> &out, std::string &err);

// Validates shapes / value ranges of tables coming across the C ABI.
bool validate_tables(const RefTables &t, std::string &err);

// Lower one automaton for `char_width`-byte haystacks.  `lds_table_budget`: bytes of LDS the automaton may
// take (tables above it are walked out of HBM: MODE_GLOBAL).  `global_walk`: build the plain uint16 layout
// read from global memory (used for the backward automaton of find()).
// `with_backward_maps` (W_FORWARDS only): append the backward automaton's char -> column maps to the blob.
// `no_pair`: never the two-chars-per-lookup table (the find-all kernel walks from per-lane char positions).
Program lower(const RefTables &t, Which which, int char_width, size_t lds_table_budget, bool global_walk,
              bool with_backward_maps = false, bool no_pair = false);

} // namespace needle
