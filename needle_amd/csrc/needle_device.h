// Shared between the host lowering (needle_lower.cpp), the launcher and the kernels (needle_kernels.hip).
#pragma once
#include <stdint.h>

namespace needle {

// Which generated loop of the reference a launch stands for.
enum Op : int { OP_MATCHES = 0, OP_CONTAINED_IN = 1, OP_FIND = 2 };

// How a lowered automaton is walked on the device.
enum Mode : int {
    MODE_NIBBLE = 0,  // <= 8 states: per-char transition FUNCTION packed 4 bits/state in one dword (no dependent LDS lookup)
    MODE_TABLE8 = 1,  // [state][column] uint8 next-state table in LDS
    MODE_TABLE16 = 2, // [state][column] uint16 next-state table in LDS
    MODE_GLOBAL = 3,  // uint16 table too large for LDS: walked out of HBM/L2
};

// Device-side numbering of a lowered automaton (independent of the reference's state numbers):
//   0            sink (dead, absorbing, non-accepting)
//   1 .. A0-1    non-accepting states
//   A0 .. n-1    accepting states          => accepted(s) == (s >= A0)
// Column layout of a row: reference classes 0..N-1, then OVER (char > maxChar), then PAD (index >= row length).
struct ProgHeader {
    uint32_t mode;       // Mode
    uint32_t n_states;   // incl. sink
    uint32_t n_cols;     // N + 2
    uint32_t start;      // device id of the reference's state 0
    uint32_t accept_lo;  // A0
    uint32_t root_accepting;
    uint32_t lds_bytes;  // bytes of the blob that the kernel stages in LDS (0 for MODE_GLOBAL table part)
    // byte offsets inside the blob (all 16-byte aligned); 0xFFFFFFFF = absent
    uint32_t off_f;      // MODE_NIBBLE: uint32 F[] -- char_width 1: 257 entries indexed by byte (256 = PAD);
                         //              char_width 2: n_cols entries indexed by column
    uint32_t off_cmap;   // char_width 1 table modes: uint8 column[256]
    uint32_t off_ptab;   // char_width 2: uint8 page_of[256] (high byte -> page)
    uint32_t off_pages;  // char_width 2: uint8 column[n_pages][256]
    uint32_t off_table;  // table modes: next-state table [n_states][n_cols]
    uint32_t n_pages;
    uint32_t pad_col;    // column index of PAD (= n_cols - 1)
};

struct ScanArgs {
    const uint8_t *rows;
    uint64_t n_rows;
    uint64_t stride_bytes;
    uint64_t total_bytes;   // n_rows * stride_bytes (clamp for tail reads)
    uint32_t row_len;       // chars, when lengths == nullptr
    const uint32_t *lengths;
    const uint8_t *prog;    // forward program blob (device)
    ProgHeader hdr;
    const uint8_t *bprog;   // OP_FIND: backward program blob (device, walked out of global memory), or nullptr
    ProgHeader bhdr;
    int32_t fixed_len;      // OP_FIND: >= 0 => start = end - fixed_len
    uint64_t *bitmap;
    int32_t *start;
    int32_t *end;
};

constexpr int kWavesPerBlock = 16;     // 1024 threads: one workgroup per CU shares one LDS copy of the tables
constexpr int kChunkBytes = 128;       // bytes of each row staged per step (one full 128-B line per row)
constexpr int kTileBytes = 64 * kChunkBytes;  // 64 rows (one per lane) x 128 B = 8 KiB per wave

} // namespace needle
