// Compact result form of find() (SURVEY.md s8d "results landed in host-visible memory", s5 "compact encoding").
//
// The reference reports a row's find() through three values on its Matcher -- matched, start(), end()
// (DFAClassBuilder.java:625-659, :661-667) -- and the batch ABI mirrors that as a bitmap plus two int32 per ROW: 80 MB
// per 10M rows, all of which a host has to pull over PCIe although unmatched rows carry nothing but -1 / -1.  Here the
// matched rows only are written, in row order, as {uint32 row, uint16 start, uint16 end}: 8 bytes per MATCHED row.
//
// Three small kernels behind the ordinary find() scan (which leaves the bitmap and start / end in HBM):
//   word_scan_kernel   popcount of every bitmap word, exclusive prefix inside blocks of 2048 words, block totals
//   block_scan_kernel  exclusive prefix of the block totals (one workgroup), total number of matched rows
//   fill_kernel        one wave per bitmap word, one row per lane: coalesced reads of start / end, the record index is the
//                      word's prefix + the number of set bits below the lane
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstdlib>
#include <string>

#include "../../include/needle_hip.h"

namespace needle {
int set_error(int code, const std::string &msg);
hipError_t scratch_malloc(void **out, size_t bytes, hipStream_t stream); // needle_api.cpp: the library's own memory pool
hipError_t scratch_free(void *p, hipStream_t stream);
}

namespace {

constexpr int kWordsPerBlock = 2048; // 256 threads x 8 words

__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t *lds4, uint32_t &block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) lds4[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t s = lds4[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    block_total = tot;
    return base + incl - v;
}

// Bits of the last bitmap word at or beyond n_rows do not count, whatever the scan that wrote the word left there.
__device__ __forceinline__ uint64_t live_bits(uint64_t word, uint64_t w, uint64_t n_words, uint64_t n_rows) {
    return (w + 1 == n_words && (n_rows & 63ull)) ? word & ((1ull << (n_rows & 63ull)) - 1ull) : word;
}

__global__ __launch_bounds__(256) void word_scan_kernel(const uint64_t *bitmap, uint64_t n_words, uint64_t n_rows, uint32_t *word_off, uint32_t *block_sum) {
    __shared__ uint32_t lds4[4];
    const uint64_t w0 = (uint64_t)blockIdx.x * kWordsPerBlock + (uint64_t)threadIdx.x * 8;
    uint32_t c[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        c[k] = (w0 + k < n_words) ? (uint32_t)__popcll(live_bits(bitmap[w0 + k], w0 + k, n_words, n_rows)) : 0u;
        sum += c[k];
    }
    uint32_t total;
    uint32_t at = block_exclusive_scan_256(sum, lds4, total);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (w0 + k < n_words) word_off[w0 + k] = at;
        at += c[k];
    }
    if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void block_scan_kernel(const uint32_t *block_sum, uint32_t n_blocks, uint64_t *block_off, uint64_t *total_out) {
    __shared__ uint32_t lds4[4];
    uint64_t carry = 0;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 256) {
        const uint32_t b = b0 + threadIdx.x;
        const uint32_t v = b < n_blocks ? block_sum[b] : 0u;
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan_256(v, lds4, tot);
        if (b < n_blocks) block_off[b] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(256) void fill_kernel(const uint64_t *bitmap, uint64_t n_words, uint64_t n_rows, const int32_t *start, const int32_t *end,
                                                    const uint32_t *word_off, const uint64_t *block_off, needle_match_rec *out, uint64_t cap,
                                                    uint64_t row_base) {
    const int lane = threadIdx.x & 63;
    const uint64_t waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n_words; w += waves) {
        const uint64_t m = live_bits(bitmap[w], w, n_words, n_rows);
        if (m == 0ull) continue;
        const uint64_t row = (w << 6) + (uint64_t)lane;
        if (((m >> lane) & 1ull) && row < n_rows) {
            const uint64_t idx = block_off[w / kWordsPerBlock] + word_off[w] + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
            if (idx < cap) {
                needle_match_rec r;
                r.row = (uint32_t)(row_base + row);
                r.start = (uint16_t)start[row];
                r.end = (uint16_t)end[row];
                out[idx] = r;
            }
        }
    }
}

int fail(int code, const std::string &msg) { return needle::set_error(code, msg); }

// row_base: added to the row numbers written (chunks of a host batch)
int find_compact(const needle_pattern *p, const needle_batch_view *v, uint64_t *d_bitmap, needle_match_rec *d_recs, uint64_t cap,
                 uint64_t *d_n_matched, uint64_t row_base, void *stream_) {
    if (!p || !v) return fail(NEEDLE_ERR_INVALID, "NULL argument");
    if (!d_bitmap || !d_n_matched || (cap && !d_recs)) return fail(NEEDLE_ERR_INVALID, "output buffer is NULL");
    // (per-row lengths are device memory: the stride bounds them, and it may be the limit rounded up to the 16-byte row alignment)
    if (v->lengths ? v->row_stride > 65536u : v->row_len > 65534u)
        return fail(NEEDLE_ERR_UNSUPPORTED, "the compact records hold 16-bit offsets: rows of at most 65 534 chars (use needle_find_dev)");
    if (row_base + v->n_rows >= (1ull << 32)) return fail(NEEDLE_ERR_UNSUPPORTED, "the compact records hold 32-bit row numbers");
    hipStream_t stream = (hipStream_t)stream_;
    if (v->n_rows == 0) {
        if (hipMemsetAsync(d_n_matched, 0, 8, stream) != hipSuccess) return fail(NEEDLE_ERR_DEVICE, "hipMemsetAsync");
        return NEEDLE_OK;
    }
    const uint64_t n = v->n_rows, n_words = (n + 63) / 64;
    const uint32_t n_blocks = (uint32_t)((n_words + kWordsPerBlock - 1) / kWordsPerBlock);
    auto up = [](uint64_t x) { return (x + 255) & ~(uint64_t)255; };
    const uint64_t o_end = up(n * 4), o_woff = o_end + up(n * 4), o_bsum = o_woff + up(n_words * 4), o_boff = o_bsum + up((uint64_t)n_blocks * 4),
                   total = o_boff + up((uint64_t)n_blocks * 8);
    uint8_t *tmp = nullptr;
    if (needle::scratch_malloc((void **)&tmp, total, stream) != hipSuccess) return fail(NEEDLE_ERR_DEVICE, "hipMallocAsync (compact find scratch)");
    auto done = [&](int code) {
        (void)needle::scratch_free(tmp, stream);
        return code;
    };
    int32_t *d_start = (int32_t *)tmp, *d_end = (int32_t *)(tmp + o_end);
    uint32_t *word_off = (uint32_t *)(tmp + o_woff), *block_sum = (uint32_t *)(tmp + o_bsum);
    uint64_t *block_off = (uint64_t *)(tmp + o_boff);
    const int rc = needle_find_dev(p, v, d_bitmap, d_start, d_end, stream_);
    if (rc) return done(rc);
    hipLaunchKernelGGL(word_scan_kernel, dim3(n_blocks), dim3(256), 0, stream, d_bitmap, n_words, n, word_off, block_sum);
    hipLaunchKernelGGL(block_scan_kernel, dim3(1), dim3(256), 0, stream, block_sum, n_blocks, block_off, d_n_matched);
    const unsigned grid = (unsigned)std::min<uint64_t>((n_words + 3) / 4, 4096);
    hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(256), 0, stream, d_bitmap, n_words, n, d_start, d_end, word_off, block_off, d_recs, cap, row_base);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return done(fail(NEEDLE_ERR_DEVICE, std::string("compact find kernels: ") + hipGetErrorString(e)));
    return done(NEEDLE_OK);
}

// ---- every match of every row in COMPACT (CSR) form in one call (needle_find_all_compact16_dev).  The two-pass form
// (needle_count_matches_dev, the caller's prefix sum, needle_find_all_csr_dev) walks the text TWICE; here the text is walked once --
// needle_find_all_blocked16_dev into scratch: group-blocked slots, slot k of 64 rows one 256-byte run -- and three small kernels turn
// counts + blocks into offsets + a dense match array:
//   group_sum_kernel     one wave per 64-row group: the group's matches
//   u32_scan_kernel      exclusive prefix of the group sums inside blocks of 2048 groups, block totals (then block_scan_kernel)
//   compact_kernel       one wave per group, one row per lane: offsets[row] = group base + prefix of the counts below it; slot k of the
//                        group is read as one coalesced run and its live entries go to offsets[row] + k
__global__ __launch_bounds__(256) void group_sum_kernel(const uint32_t *counts, uint64_t n_rows, uint64_t n_groups, uint32_t *gsum) {
    const int lane = threadIdx.x & 63;
    const uint64_t waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; g < n_groups; g += waves) {
        const uint64_t row = (g << 6) + (uint64_t)lane;
        uint32_t c = row < n_rows ? counts[row] : 0u;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
        if (lane == 0) gsum[g] = c;
    }
}

__global__ __launch_bounds__(256) void u32_scan_kernel(const uint32_t *v, uint64_t n, uint32_t *off, uint32_t *block_sum) {
    __shared__ uint32_t lds4[4];
    const uint64_t i0 = (uint64_t)blockIdx.x * kWordsPerBlock + (uint64_t)threadIdx.x * 8;
    uint32_t c[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        c[k] = (i0 + k < n) ? v[i0 + k] : 0u;
        sum += c[k];
    }
    uint32_t total;
    uint32_t at = block_exclusive_scan_256(sum, lds4, total);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (i0 + k < n) off[i0 + k] = at;
        at += c[k];
    }
    if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void compact_kernel(const uint32_t *counts, const uint32_t *blocks, uint32_t slots, uint64_t n_rows, uint64_t n_groups,
                                                       const uint32_t *goff, const uint64_t *block_off, const uint64_t *total, uint64_t *offsets,
                                                       uint32_t *out, uint64_t cap) {
    const int lane = threadIdx.x & 63;
    const uint64_t waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; g < n_groups; g += waves) {
        const uint64_t row = (g << 6) + (uint64_t)lane;
        const uint32_t c = row < n_rows ? counts[row] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)incl, o);
            if (lane >= o) incl += t;
        }
        const uint64_t base = block_off[g / kWordsPerBlock] + goff[g] + (incl - c);
        if (row < n_rows) offsets[row] = base;
        uint32_t most = c;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const uint32_t t = (uint32_t)__shfl_xor((int)most, o);
            most = t > most ? t : most;
        }
        const uint32_t *blk = blocks + g * (uint64_t)slots * 64u + (uint32_t)lane;
        for (uint32_t k = 0; k < most; ++k) { // (wave-uniform trip count: the group's busiest row)
            if (k < c) {
                const uint32_t m = blk[(uint64_t)k * 64u];
                if (base + k < cap) out[base + k] = m;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) offsets[n_rows] = *total;
}

int find_all_compact16(const needle_pattern *p, const needle_batch_view *v, uint32_t max_per_row, uint64_t *d_offsets, uint32_t *d_start_end16,
                       uint64_t cap, uint64_t *d_total, int *more, void *stream_) {
    if (!p || !v) return fail(NEEDLE_ERR_INVALID, "NULL argument");
    if (!d_offsets || !d_total || (cap && !d_start_end16)) return fail(NEEDLE_ERR_INVALID, "output buffer is NULL");
    if (max_per_row == 0 || max_per_row > 4096) return fail(NEEDLE_ERR_INVALID, "max_per_row must be 1 .. 4096");
    hipStream_t stream = (hipStream_t)stream_;
    if (more) *more = 0;
    if (v->n_rows == 0) {
        if (hipMemsetAsync(d_total, 0, 8, stream) != hipSuccess || hipMemsetAsync(d_offsets, 0, 8, stream) != hipSuccess) return fail(NEEDLE_ERR_DEVICE, "hipMemsetAsync");
        return NEEDLE_OK;
    }
    const uint64_t n = v->n_rows, n_groups = (n + 63) / 64;
    const uint32_t n_blocks = (uint32_t)((n_groups + kWordsPerBlock - 1) / kWordsPerBlock);
    auto up = [](uint64_t x) { return (x + 255) & ~(uint64_t)255; };
    const uint64_t o_blocks = up(n * 4), o_gsum = o_blocks + up(n_groups * (uint64_t)max_per_row * 256), o_goff = o_gsum + up(n_groups * 4),
                   o_bsum = o_goff + up(n_groups * 4), o_boff = o_bsum + up((uint64_t)n_blocks * 4), total = o_boff + up((uint64_t)n_blocks * 8);
    uint8_t *tmp = nullptr;
    if (needle::scratch_malloc((void **)&tmp, total, stream) != hipSuccess) return fail(NEEDLE_ERR_DEVICE, "hipMallocAsync (compact find-all scratch)");
    auto done = [&](int code) {
        (void)needle::scratch_free(tmp, stream);
        return code;
    };
    uint32_t *counts = (uint32_t *)tmp, *blocks = (uint32_t *)(tmp + o_blocks), *gsum = (uint32_t *)(tmp + o_gsum), *goff = (uint32_t *)(tmp + o_goff),
             *bsum = (uint32_t *)(tmp + o_bsum);
    uint64_t *boff = (uint64_t *)(tmp + o_boff);
    // (more == NULL: no synchronisation -- a row with more than max_per_row matches is then silently cut, as in the dense forms)
    const int rc = needle_find_all_blocked16_dev(p, v, max_per_row, counts, blocks, more, stream_);
    if (rc) return done(rc);
    const unsigned grid = (unsigned)std::min<uint64_t>((n_groups + 3) / 4, 4096);
    hipLaunchKernelGGL(group_sum_kernel, dim3(grid), dim3(256), 0, stream, counts, n, n_groups, gsum);
    hipLaunchKernelGGL(u32_scan_kernel, dim3(n_blocks), dim3(256), 0, stream, gsum, n_groups, goff, bsum);
    hipLaunchKernelGGL(block_scan_kernel, dim3(1), dim3(256), 0, stream, bsum, n_blocks, boff, d_total);
    hipLaunchKernelGGL(compact_kernel, dim3(grid), dim3(256), 0, stream, counts, blocks, max_per_row, n, n_groups, goff, boff, d_total, d_offsets,
                       d_start_end16, cap);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return done(fail(NEEDLE_ERR_DEVICE, std::string("compact find-all kernels: ") + hipGetErrorString(e)));
    return done(NEEDLE_OK);
}

// ---- host batches: chunks of at most ~2 GiB of rows resident at a time (64-row boundaries: whole bitmap words)
struct HostChunk {
    uint8_t *d_rows = nullptr;
    uint32_t *d_len = nullptr;
    needle_batch_view view;
    ~HostChunk() {
        if (d_rows) (void)hipFree(d_rows);
        if (d_len) (void)hipFree(d_len);
    }
    int upload(const needle_batch_view *v, uint64_t r0, uint64_t cnt) {
        const uint64_t cw = v->char_width, src_stride = v->row_stride * cw;
        uint64_t dst_stride = (src_stride + 15) & ~(uint64_t)15;
        if (dst_stride == 0) dst_stride = 16;
        hipError_t e = hipMalloc((void **)&d_rows, cnt * dst_stride);
        const uint8_t *src = (const uint8_t *)v->rows + r0 * src_stride;
        if (e == hipSuccess) {
            if (dst_stride == src_stride) e = hipMemcpy(d_rows, src, cnt * src_stride, hipMemcpyHostToDevice);
            else {
                e = hipMemset(d_rows, 0, cnt * dst_stride);
                if (e == hipSuccess && src_stride) e = hipMemcpy2D(d_rows, dst_stride, src, src_stride, src_stride, cnt, hipMemcpyHostToDevice);
            }
        }
        if (e == hipSuccess && v->lengths) {
            e = hipMalloc((void **)&d_len, cnt * 4);
            if (e == hipSuccess) e = hipMemcpy(d_len, v->lengths + r0, cnt * 4, hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) return fail(NEEDLE_ERR_DEVICE, std::string("host chunk upload: ") + hipGetErrorString(e));
        view = *v;
        view.rows = d_rows;
        view.lengths = d_len;
        view.n_rows = cnt;
        view.row_stride = dst_stride / cw;
        return NEEDLE_OK;
    }
};

int check_host_view(const needle_batch_view *v) {
    if (!v) return fail(NEEDLE_ERR_INVALID, "batch view is NULL");
    if (v->char_width != 1 && v->char_width != 2) return fail(NEEDLE_ERR_INVALID, "char_width must be 1 or 2");
    if (v->n_rows && !v->rows) return fail(NEEDLE_ERR_INVALID, "rows is NULL");
    if (v->row_len > v->row_stride) return fail(NEEDLE_ERR_INVALID, "row_len > row_stride");
    if (v->lengths)
        for (uint64_t r = 0; r < v->n_rows; ++r)
            if (v->lengths[r] > v->row_stride) return fail(NEEDLE_ERR_INVALID, "lengths[r] > row_stride");
    return NEEDLE_OK;
}

uint64_t rows_per_chunk(const needle_batch_view *v) {
    static const uint64_t kHostChunkBytes = getenv("NEEDLE_HOST_CHUNK_BYTES") ? (uint64_t)atoll(getenv("NEEDLE_HOST_CHUNK_BYTES")) : (2ull << 30);
    const uint64_t row_bytes = std::max<uint64_t>(16, (v->row_stride * v->char_width + 15) & ~(uint64_t)15);
    return std::max<uint64_t>(64, (kHostChunkBytes / row_bytes) & ~(uint64_t)63);
}

} // namespace

extern "C" {

int needle_find_all_compact16_dev(const needle_pattern *p, const needle_batch_view *v, uint32_t max_per_row, uint64_t *d_offsets,
                                  uint32_t *d_start_end16, uint64_t cap, uint64_t *d_total, int *more, void *stream) {
    return find_all_compact16(p, v, max_per_row, d_offsets, d_start_end16, cap, d_total, more, stream);
}

int needle_find_compact_dev(const needle_pattern *p, const needle_batch_view *v, uint64_t *d_bitmap, needle_match_rec *d_recs, uint64_t cap,
                            uint64_t *d_n_matched, void *stream_) {
    return find_compact(p, v, d_bitmap, d_recs, cap, d_n_matched, 0, stream_);
}

// Host batch -> bitmap + the matched rows' records: what crosses PCIe on the way back is 1 bit per row + 8 bytes per
// MATCHED row (needle_find_host: 8 bytes per row).  *n_matched is the total; at most cap records are written.
int needle_find_compact_host(const needle_pattern *p, const needle_batch_view *v, uint64_t *bitmap, needle_match_rec *recs, uint64_t cap,
                             uint64_t *n_matched) {
    if (!p || !n_matched) return fail(NEEDLE_ERR_INVALID, "NULL argument");
    int rc = check_host_view(v);
    if (rc) return rc;
    *n_matched = 0;
    if (v->n_rows == 0) return NEEDLE_OK;
    if (!bitmap || (cap && !recs)) return fail(NEEDLE_ERR_INVALID, "output buffer is NULL");
    if ((v->lengths ? v->row_stride : v->row_len) > 65534u) // the caller's own stride: the upload pads it to 16 bytes
        return fail(NEEDLE_ERR_UNSUPPORTED, "the compact records hold 16-bit offsets: rows of at most 65 534 chars (use needle_find_host)");
    const uint64_t per = rows_per_chunk(v);
    for (uint64_t r0 = 0; r0 < v->n_rows; r0 += per) {
        const uint64_t cnt = std::min<uint64_t>(per, v->n_rows - r0), words = (cnt + 63) / 64;
        HostChunk ch;
        if ((rc = ch.upload(v, r0, cnt))) return rc;
        uint8_t *d_out = nullptr; // bitmap | count | records
        const uint64_t o_n = (words * 8 + 15) & ~(uint64_t)15, o_rec = o_n + 16;
        const uint64_t room = cap > *n_matched ? std::min<uint64_t>(cap - *n_matched, cnt) : 0;
        if (hipMalloc((void **)&d_out, o_rec + room * sizeof(needle_match_rec)) != hipSuccess) return fail(NEEDLE_ERR_DEVICE, "hipMalloc (compact find results)");
        rc = find_compact(p, &ch.view, (uint64_t *)d_out, (needle_match_rec *)(d_out + o_rec), room, (uint64_t *)(d_out + o_n), r0, nullptr);
        uint64_t m = 0;
        hipError_t e = hipSuccess;
        if (rc == NEEDLE_OK) {
            e = hipMemcpy(&m, d_out + o_n, 8, hipMemcpyDeviceToHost); // (synchronises with the kernels on the null stream)
            if (e == hipSuccess) e = hipMemcpy(bitmap + r0 / 64, d_out, words * 8, hipMemcpyDeviceToHost);
            if (e == hipSuccess && std::min(m, room)) e = hipMemcpy(recs + *n_matched, d_out + o_rec, std::min(m, room) * sizeof(needle_match_rec), hipMemcpyDeviceToHost);
        }
        (void)hipFree(d_out);
        if (rc) return rc;
        if (e != hipSuccess) return fail(NEEDLE_ERR_DEVICE, std::string("compact find download: ") + hipGetErrorString(e));
        *n_matched += m;
    }
    return NEEDLE_OK;
}

// needle_find_host with start / end as ONE dword per row (low half start, high half end, 0xFFFF = no match: the form
// needle_pack_start_end16_dev writes): 4 bytes per row over PCIe instead of 8.  Rows of at most 65 534 chars.
int needle_find_packed16_host(const needle_pattern *p, const needle_batch_view *v, uint64_t *bitmap, uint32_t *start_end16) {
    if (!p) return fail(NEEDLE_ERR_INVALID, "pattern is NULL");
    int rc = check_host_view(v);
    if (rc) return rc;
    if (v->n_rows == 0) return NEEDLE_OK;
    if (!bitmap || !start_end16) return fail(NEEDLE_ERR_INVALID, "output buffer is NULL");
    if ((v->lengths ? v->row_stride : v->row_len) > 65534u) // the caller's own stride: the upload pads it to 16 bytes
        return fail(NEEDLE_ERR_UNSUPPORTED, "16-bit offsets: rows of at most 65 534 chars (use needle_find_host)");
    const uint64_t per = rows_per_chunk(v);
    for (uint64_t r0 = 0; r0 < v->n_rows; r0 += per) {
        const uint64_t cnt = std::min<uint64_t>(per, v->n_rows - r0), words = (cnt + 63) / 64;
        HostChunk ch;
        if ((rc = ch.upload(v, r0, cnt))) return rc;
        uint8_t *d_out = nullptr; // bitmap | packed
        const uint64_t o_p = (words * 8 + 15) & ~(uint64_t)15;
        if (hipMalloc((void **)&d_out, o_p + cnt * 4) != hipSuccess) return fail(NEEDLE_ERR_DEVICE, "hipMalloc (find results)");
        rc = needle_find_packed16_dev(p, &ch.view, (uint64_t *)d_out, (uint32_t *)(d_out + o_p), nullptr);
        hipError_t e = hipSuccess;
        if (rc == NEEDLE_OK) {
            e = hipMemcpy(bitmap + r0 / 64, d_out, words * 8, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(start_end16 + r0, d_out + o_p, cnt * 4, hipMemcpyDeviceToHost);
        }
        (void)hipFree(d_out);
        if (rc) return rc;
        if (e != hipSuccess) return fail(NEEDLE_ERR_DEVICE, std::string("find download: ") + hipGetErrorString(e));
    }
    return NEEDLE_OK;
}

// needle_find_host with start / end as ONE dword per row (low half start, high half end, 0xFFFF = no match: the form
// needle_pack_start_len8_dev writes): 4 bytes per row over PCIe instead of 8.  Rows of at most 65 534 chars.
int needle_find_packed8_host(const needle_pattern *p, const needle_batch_view *v, uint64_t *bitmap, uint16_t *start_len8) {
    if (!p) return fail(NEEDLE_ERR_INVALID, "pattern is NULL");
    int rc = check_host_view(v);
    if (rc) return rc;
    if (v->n_rows == 0) return NEEDLE_OK;
    if (!bitmap || !start_len8) return fail(NEEDLE_ERR_INVALID, "output buffer is NULL");
    if ((v->lengths ? v->row_stride : v->row_len) > 256u) // the caller's own stride: the upload pads it to 16 bytes
        return fail(NEEDLE_ERR_UNSUPPORTED, "8-bit start / length: rows of at most 256 chars (use needle_find_packed16_host)");
    const uint64_t per = rows_per_chunk(v);
    for (uint64_t r0 = 0; r0 < v->n_rows; r0 += per) {
        const uint64_t cnt = std::min<uint64_t>(per, v->n_rows - r0), words = (cnt + 63) / 64;
        HostChunk ch;
        if ((rc = ch.upload(v, r0, cnt))) return rc;
        uint8_t *d_out = nullptr; // bitmap | packed
        const uint64_t o_p = (words * 8 + 15) & ~(uint64_t)15;
        if (hipMalloc((void **)&d_out, o_p + cnt * 2) != hipSuccess) return fail(NEEDLE_ERR_DEVICE, "hipMalloc (find results)");
        rc = needle_find_packed8_dev(p, &ch.view, (uint64_t *)d_out, (uint16_t *)(d_out + o_p), nullptr);
        hipError_t e = hipSuccess;
        if (rc == NEEDLE_OK) {
            e = hipMemcpy(bitmap + r0 / 64, d_out, words * 8, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(start_len8 + r0, d_out + o_p, cnt * 2, hipMemcpyDeviceToHost);
        }
        (void)hipFree(d_out);
        if (rc) return rc;
        if (e != hipSuccess) return fail(NEEDLE_ERR_DEVICE, std::string("find download: ") + hipGetErrorString(e));
    }
    return NEEDLE_OK;
}

} // extern "C"
