// C ABI of libneedle_hip.so (include/needle_hip.h): pattern objects, per-device program cache, batch entry
// points, and the single-haystack Matcher mirror.  No CPU matching path exists in this library.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/needle_hip.h"
#include "needle_device.h"
#include "needle_find_all.h"
#include "needle_lower.h"
#include "needle_regex.h"

namespace needle {
hipError_t launch_scan(int op, int char_width, const ScanArgs &a, int n_cus, hipStream_t stream);
bool shape_for_program(const ProgHeader &h, int char_width, int *waves, int *chb, int *tiles_in_f_rows);
hipError_t launch_find_all(int char_width, const FindAllArgs &fa, int n_cus, hipStream_t stream); // needle_find_all.hip
hipError_t launch_find_all_lockstep(int char_width, const FindAllArgs &fa, int n_cus, hipStream_t stream); // needle_find_all_ls.hip
bool find_all_lockstep_shape_ok(const FindAllArgs &fa);
hipError_t launch_find_all_collect(uint64_t n_rows, uint32_t slots, uint32_t k, const int32_t *s, const int32_t *e, int32_t *cursor,
                                   uint32_t *counts, int32_t *starts, int32_t *ends, int32_t *any_hit, int n_cus, hipStream_t stream);
hipError_t launch_long_rows(int char_width, const StripeArgs &a, int n_cus, hipStream_t stream);
hipError_t launch_spec_len(const SpecArgs &a, hipStream_t stream);
hipError_t launch_spec_init(const SpecArgs &a, hipStream_t stream);
hipError_t launch_spec_fix(const SpecArgs &a, hipStream_t stream);
hipError_t launch_spec_reduce(const SpecArgs &a, hipStream_t stream);
hipError_t launch_backward_rows(int char_width, const StripeArgs &a, hipStream_t stream);
#ifdef NEEDLE_TUNING // needle_dict.hip (two row sets per wave; measured, no faster -- DESIGN.md s4) is part of measurement builds only
bool dict_kernel_applies(int char_width, const ScanArgs &a);
hipError_t launch_dict(int op, const ScanArgs &a, int n_cus, hipStream_t stream);
#endif
// needle_ngram.hip: containedIn / find behind the n-gram candidate filter
bool ngram_shape_ok(const ScanArgs &a);
size_t ngram_lds_bytes(const ProgHeader &h, const NgramParams &ng);
hipError_t launch_ngram(int op, const ScanArgs &a, const NgramParams &ng, const uint32_t *d_bitmap, uint32_t *d_stats, int n_cus, hipStream_t stream,
                        int char_width = 1, int page = 0, int sub = 0xFF);
size_t ngram_find_all_lds_bytes(const ProgHeader &h, const NgramParams &ng);
hipError_t launch_ngram_find_all(const ScanArgs &a, const NgramParams &ng, const uint32_t *d_bitmap, uint32_t *d_stats, uint32_t slots, uint32_t *counts,
                                 int32_t *starts, int32_t *ends, uint32_t *packed, int32_t *more, const uint64_t *offsets, bool count_only, int n_cus,
                                 hipStream_t stream, int char_width, int page, int sub, uint32_t kshift);
int ngram_level(); // needle_lower.cpp (NEEDLE_PREFILTER)
hipError_t launch_unpack(const void *data, const uint64_t *offsets, uint64_t n_rows, uint32_t cw, void *out,
                         uint64_t stride_bytes, uint32_t *lengths, int32_t *overflow, int n_cus, hipStream_t stream);
} // namespace needle

using namespace needle;

static thread_local std::string g_err;

static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

namespace needle {
int set_error(int code, const std::string &msg) { return fail(code, msg); } // (needle_multi.cpp reports through the same channel)
} // namespace needle

// Stream-ordered scratch memory comes from a pool of the library's own, one per device, that KEEPS what it is given back:
// HIP's default pool returns freed memory to the driver at the next synchronisation point (release threshold 0), so a
// caller that synchronises after every call would pay a fresh driver allocation of tens of megabytes per call
// (needle_find_compact_dev: 2.5 ms per step on a 10M-row batch before this).  What the pool keeps is bounded:
// NEEDLE_SCRATCH_KEEP_MB (default 512) is its release threshold -- freed memory above it goes back to the driver at the next
// synchronisation point, so one large batch does not pin its peak scratch for the life of the process -- and
// needle_trim_scratch() hands back everything that is free.
namespace needle {
static std::mutex g_scratch_mu;
static std::map<int, hipMemPool_t> g_scratch_pools;
hipError_t scratch_trim(size_t keep_bytes) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    hipError_t first = hipSuccess;
    for (auto &kv : g_scratch_pools)
        if (kv.second) {
            const hipError_t e = hipMemPoolTrimTo(kv.second, keep_bytes);
            if (e != hipSuccess && first == hipSuccess) first = e;
        }
    return first;
}
hipError_t scratch_malloc(void **out, size_t bytes, hipStream_t stream) {
    std::mutex &mu = g_scratch_mu;
    std::map<int, hipMemPool_t> &pools = g_scratch_pools;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemPool_t pool = nullptr;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = pools.find(dev);
        if (it == pools.end()) {
            hipMemPoolProps props;
            memset(&props, 0, sizeof(props));
            props.allocType = hipMemAllocationTypePinned;
            props.handleTypes = hipMemHandleTypeNone;
            props.location.type = hipMemLocationTypeDevice;
            props.location.id = dev;
            if (hipMemPoolCreate(&pool, &props) == hipSuccess) {
                static const uint64_t keep_mb = getenv("NEEDLE_SCRATCH_KEEP_MB") ? (uint64_t)atoll(getenv("NEEDLE_SCRATCH_KEEP_MB")) : 512;
                uint64_t keep = keep_mb << 20;
                (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
            } else {
                (void)hipGetLastError();
                pool = nullptr; // (no pool of our own: the device's default one)
            }
            it = pools.emplace(dev, pool).first;
        }
        pool = it->second;
    }
    return pool ? hipMallocFromPoolAsync(out, bytes, pool, stream) : hipMallocAsync(out, bytes, stream);
}
hipError_t scratch_free(void *p, hipStream_t stream) { return hipFreeAsync(p, stream); }
} // namespace needle

static int hip_fail(hipError_t e, const char *what) {
    return fail(NEEDLE_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return hip_fail(_e, #expr); \
    } while (0)

struct DevProgram {
    Program prog;
    uint8_t *d_blob = nullptr;
    uint32_t *d_ng = nullptr; // the n-gram filter's bitmap (prog.ng.p.on)
    // Flood watch of the n-gram filter kernel.  Text that passes the filter almost everywhere (built from the dictionary's own
    // keyword tails: one automaton run per window) makes that kernel several times SLOWER than the ordinary scan (measured:
    // 5.3 against 1.13 ms on the C3-sparse dictionary, scripts/ngram_worstcase.py).  Every filter launch adds its candidates
    // and KiB of text to d_ng_stats; the pair is copied to the pinned h_ng_stats behind the kernel (on ng_stream, below).  The NEXT call
    // reads it without waiting: above 16 candidates per KiB (the break-even; the bench text has 3.3) the filter is suspended for
    // the program's next 32 calls, doubling up to 1024 while the text stays like that.  Answers are the same either way.
    uint32_t *d_ng_stats = nullptr;            // {candidates, KiB of text}: MONOTONIC device counters (never reset by the host)
    volatile uint64_t *h_ng_stats = nullptr;    // pinned: the pair as it stood behind the last completed filter launch (one 8-byte copy)
    // The copy runs on a stream of its own behind an event of the launch: on the caller's stream it could queue behind another stream's
    // bulk D2H on the copy engine and hold the NEXT scan back until that finished (measured: pipelined host landing of a 1.1 ms scan
    // at 1.86 ms per step = scan + copy, serialised).
    hipStream_t ng_stream = nullptr;
    hipEvent_t ng_ev = nullptr;
    // the watch's own state, per program (= per pattern x device x op), shared by every stream and thread that uses it
    mutable std::mutex ng_mu;
    mutable uint32_t ng_seen_cand = 0, ng_seen_kib = 0; // what the last evaluation had seen: the watch works on deltas
    mutable int ng_suspend = 0, ng_backoff = 32;
    mutable float ng_last_rate = 0.0f;
    mutable uint64_t ng_launches = 0, ng_suspended_calls = 0;
    DevProgram() = default;
    DevProgram(DevProgram &&o) noexcept
        : prog(std::move(o.prog)), d_blob(o.d_blob), d_ng(o.d_ng), d_ng_stats(o.d_ng_stats), h_ng_stats(o.h_ng_stats), ng_stream(o.ng_stream), ng_ev(o.ng_ev) { // (moved before first use: the watch's state starts fresh)
        o.d_blob = nullptr, o.d_ng = nullptr, o.d_ng_stats = nullptr, o.h_ng_stats = nullptr, o.ng_stream = nullptr, o.ng_ev = nullptr;
    }
};

struct needle_pattern {
    RefTables t;
    std::mutex mu;
    // (device, which, char_width, variant) -> program resident in that device's HBM
    // variant: 0 plain, 1 global-walk layout (backward automaton of find), 2 forward + backward column maps,
    //          3 HBM-table layout forced (column maps + uint16 table in one blob: the speculative-stripe fix-up walks it)
    //          4 / 5 as 0 / 2 without the pair table (the one-pass find-all kernel)
    //          6 the find-all "lengths" automaton (W_FORWARDS only; absent when the pattern does not allow it)
    //          7 the same for find() in the scan kernels
    //          8 the find-all transducer (lock-step find-all, needle_find_all_ls.hip; absent when the pattern does not allow it)
    //          9 the filter program of an automaton that fits the LDS in no form (lower_filter_hbm: HBM-table layout + n-gram filter;
    //            W_CONTAINED_IN, or W_FORWARDS in the lengths form / for one-length patterns; absent when no filter can be built)
    //          12 variant 9 for find() of a pattern WITHOUT bounded match lengths: the forward search automaton itself (no lengths form) with
    //            the backward automaton's column maps in its LDS part -- verified candidates find their starts by backward walks
    //          11 the RUN transducer (lock-step find-all of patterns without bounded match lengths whose matches are runs: `[0-9]+`;
    //            needle_lower.h lower_find_all_runs); absent when the pattern is not of that kind
    //          10 the WIDE filter program (lower_filter_wide: char_width 2 only -- UTF-16 rows of a pattern on several pages of the BMP:
    //            windows of four code units, UTF-16 HBM-table program); absent when no filter can be built
    std::map<std::tuple<int, int, int, int>, DevProgram> cache;
    std::map<int, int> cus; // device -> CU count
    // needle_pattern_prefilter_info answers (lowering a big dictionary takes seconds): per `which`, filled once
    struct PrefilterCache {
        bool have = false;
        needle_prefilter_info info;
        uint32_t m1b = 0, m2b = 0;
        std::vector<uint32_t> bitmap;
    } pf_cache[8]; // [which]: the byte programs' filter; [4 + which]: the WIDE filter's (needle_pattern_prefilter_info2)
    std::mutex pf_mu;
    std::atomic<int> pf_mode{0}; // needle_pattern_set_prefilter: 0 auto (the flood watch decides), 1 on (never suspended), 2 off (never used)
    // UTF-16 rows behind a BYTE program (utf16_route): the pattern's one page and the byte that stands for every char outside it;
    // the tables rebased to that page (page_tables: what the byte programs of a page other than 0 are lowered from)
    std::mutex u16_mu;
    int u16_state = 0, u16_page = -1, u16_sub = 0; // state 0: not looked at yet
    struct PageTables {
        RefTables t;
        bool have_ml = false;
        MatchLengths ml;
    };
    std::map<int, PageTables> page_tables; // (under `mu`)
    std::mutex ml_mu;       // guards the one-time match-length analysis only: scans of programs that exist already do not wait for it
    int ml_state = 0;       // 0: not analysed yet, 1: find-all can report starts as end - length (ml), -1: it cannot
    MatchLengths ml;
    ~needle_pattern() {
        for (auto &kv : cache) {
            if (kv.second.ng_stream) (void)hipStreamSynchronize(kv.second.ng_stream); // (a stats copy may still be on its way into h_ng_stats)
            if (kv.second.d_blob) (void)hipFree(kv.second.d_blob);
            if (kv.second.d_ng) (void)hipFree(kv.second.d_ng);
            if (kv.second.d_ng_stats) (void)hipFree(kv.second.d_ng_stats);
            if (kv.second.h_ng_stats) (void)hipHostFree((void *)kv.second.h_ng_stats);
            if (kv.second.ng_stream) (void)hipStreamDestroy(kv.second.ng_stream);
            if (kv.second.ng_ev) (void)hipEventDestroy(kv.second.ng_ev);
        }
    }
};

// Automaton LDS budget.  NEEDLE_MAX_PROG_LDS (bytes) lowers it: tests use that to force the HBM-table mode.
static size_t max_prog_lds() {
    static const size_t v = getenv("NEEDLE_MAX_PROG_LDS") ? (size_t)atol(getenv("NEEDLE_MAX_PROG_LDS")) : (size_t)kMaxProgLdsBytes;
    return v < kMaxProgLdsBytes ? v : (size_t)kMaxProgLdsBytes;
}

// The pattern's match-length analysis (needle_lower.h), run once per pattern -- it can take seconds of host time on a big
// dictionary -- and shared by every caller: the scans, needle_pattern_program_info / _prefilter_info / _match_lengths.
// nullptr: the pattern does not allow the "lengths" form.
static const MatchLengths *pattern_ml(const needle_pattern *cp) {
    needle_pattern *p = const_cast<needle_pattern *>(cp);
    std::lock_guard<std::mutex> lk(p->ml_mu);
    if (p->ml_state == 0) {
        p->ml = match_length_automaton(p->t);
        p->ml_state = p->ml.ok ? 1 : -1;
    }
    return p->ml_state > 0 ? &p->ml : nullptr;
}

static int get_program(needle_pattern *p, int which, int cw, int variant, const DevProgram **out, int *n_cus) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    // cw = 1 | page << 8: the BYTE program of the pattern rebased to one page of the BMP (UTF-16 rows narrowed on the fly: utf16_route)
    const int page = cw >> 8, cw_key = cw;
    cw &= 0xFF;
    const bool wants_ml = variant == 6 || variant == 7 || variant == 8 || ((variant == 9 || variant == 10) && which == W_FORWARDS && p->t.fixed_len < 0);
    const MatchLengths *ml67 = wants_ml ? pattern_ml(p) : nullptr; // (before p->mu: see ml_mu)
    std::lock_guard<std::mutex> lk(p->mu);
    const RefTables *tt = &p->t;
    if (page) { // chars page << 8 | b become bytes b: the class map's page in front, every automaton's maxChar moved along
        auto pit = p->page_tables.find(page);
        if (pit == p->page_tables.end()) {
            needle_pattern::PageTables pt;
            pt.t = p->t;
            for (int b = 0; b < 256; ++b) pt.t.class_map[b] = p->t.class_map[(size_t)(page << 8) | b];
            auto rebase = [&](int32_t mc) { const int32_t r = mc - (page << 8); return r > 255 ? 255 : (r < -1 ? -1 : r); };
            for (int w = 0; w < 4; ++w) pt.t.dfa[w].max_char = rebase(p->t.dfa[w].max_char);
            pit = p->page_tables.emplace(page, std::move(pt)).first;
        }
        if (ml67 && !pit->second.have_ml) {
            pit->second.ml = *ml67;
            const int32_t r = ml67->dfa.max_char - (page << 8);
            pit->second.ml.dfa.max_char = r > 255 ? 255 : (r < -1 ? -1 : r);
            pit->second.have_ml = true;
        }
        tt = &pit->second.t;
        if (ml67) ml67 = &pit->second.ml;
    }
    if (!p->cus.count(dev)) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        p->cus[dev] = prop.multiProcessorCount;
    }
    if (n_cus) *n_cus = p->cus[dev];
    auto key = std::make_tuple(dev, which, cw_key, variant);
    auto it = p->cache.find(key);
    if (it == p->cache.end()) {
        DevProgram dp;
        if (variant == 9 || variant == 10 || variant == 12) {
            if ((wants_ml && !ml67) || (variant == 10 && cw != 2) || (variant == 12 && (which != W_FORWARDS || cw != 1))) {
                *out = nullptr;
                return NEEDLE_OK;
            }
            dp.prog = variant == 10 ? lower_filter_wide(*tt, (Which)which, ml67) : lower_filter_hbm(*tt, (Which)which, variant == 12 ? nullptr : ml67, variant == 12);
            if (dp.prog.blob.empty() || !dp.prog.ng.p.on) { // (no filter: the ordinary program is what runs)
                p->cache.emplace(key, DevProgram());
                *out = nullptr;
                return NEEDLE_OK;
            }
        } else if (variant == 11) { // the RUN transducer (lock-step find-all of `[0-9]+`-like patterns: lower_find_all_runs); absent when the pattern is not one
            dp.prog = lower_find_all_runs(*tt, cw, max_prog_lds());
            if (dp.prog.blob.empty()) {
                p->cache.emplace(key, DevProgram());
                *out = nullptr;
                return NEEDLE_OK;
            }
        } else if (variant == 6 || variant == 7 || variant == 8) { // "lengths" form: the refined forward automaton + pend[] (needle_lower.h);
                                            // 6: the find-all kernel's plain layout, 7: the scan kernels' (window addressing);
                                            // 8: the find-all transducer built on it
            if (!ml67) {
                *out = nullptr;
                return NEEDLE_OK;
            }
            dp.prog = variant == 8 ? lower_find_all_transducer(*tt, *ml67, cw, max_prog_lds())
                                   : lower_match_lengths(*tt, *ml67, cw, max_prog_lds(), variant == 6);
            if (dp.prog.blob.empty()) { // (does not fit the LDS as a plain table: the ordinary program with backward walks)
                p->cache.emplace(key, DevProgram());
                *out = nullptr;
                return NEEDLE_OK;
            }
        } else
        dp.prog = lower(*tt, (Which)which, cw, variant == 3 ? 0 : max_prog_lds(), variant == 1, variant == 2 || variant == 5, variant >= 4);
        HIP_TRY(hipMalloc((void **)&dp.d_blob, dp.prog.blob.size()));
        if (hipError_t ce = hipMemcpy(dp.d_blob, dp.prog.blob.data(), dp.prog.blob.size(), hipMemcpyHostToDevice); ce != hipSuccess) {
            (void)hipFree(dp.d_blob);
            return hip_fail(ce, "hipMemcpy(program blob)");
        }
        if (dp.prog.ng.p.on) {
            const size_t nb = dp.prog.ng.bitmap.size() * 4, nb2 = dp.prog.ng.p.on2 ? dp.prog.ng.bitmap2.size() * 4 : 0; // (the second level's follows)
            hipError_t ce = hipMalloc((void **)&dp.d_ng, nb + nb2 + 16);
            if (ce == hipSuccess) ce = hipMemcpy(dp.d_ng, dp.prog.ng.bitmap.data(), nb, hipMemcpyHostToDevice);
            if (ce == hipSuccess && nb2) ce = hipMemcpy((uint8_t *)dp.d_ng + nb, dp.prog.ng.bitmap2.data(), nb2, hipMemcpyHostToDevice);
            if (ce == hipSuccess) ce = hipMalloc((void **)&dp.d_ng_stats, 8);
            if (ce == hipSuccess) ce = hipMemset(dp.d_ng_stats, 0, 8);
            if (ce == hipSuccess) ce = hipHostMalloc((void **)&dp.h_ng_stats, 8, hipHostMallocDefault);
            if (ce == hipSuccess) ce = hipStreamCreateWithFlags(&dp.ng_stream, hipStreamNonBlocking);
            if (ce == hipSuccess) ce = hipEventCreateWithFlags(&dp.ng_ev, hipEventDisableTiming);
            if (ce != hipSuccess) {
                (void)hipFree(dp.d_blob);
                if (dp.d_ng) (void)hipFree(dp.d_ng);
                if (dp.d_ng_stats) (void)hipFree(dp.d_ng_stats);
                if (dp.h_ng_stats) (void)hipHostFree((void *)dp.h_ng_stats);
                if (dp.ng_stream) (void)hipStreamDestroy(dp.ng_stream);
                if (dp.ng_ev) (void)hipEventDestroy(dp.ng_ev);
                return hip_fail(ce, "hipMalloc/hipMemcpy(n-gram bitmap)");
            }
            dp.h_ng_stats[0] = 0;
        }
        it = p->cache.emplace(key, std::move(dp)).first;
    }
    *out = it->second.d_blob ? &it->second : nullptr; // (variant 6: an empty entry = "not available for this pattern / width")
    return NEEDLE_OK;
}

static int check_view(const needle_batch_view *v, bool device) {
    if (!v) return fail(NEEDLE_ERR_INVALID, "batch view is NULL");
    if (v->char_width != 1 && v->char_width != 2) return fail(NEEDLE_ERR_INVALID, "char_width must be 1 or 2");
    if (v->n_rows && !v->rows) return fail(NEEDLE_ERR_INVALID, "rows is NULL");
    if (v->row_len > v->row_stride) return fail(NEEDLE_ERR_INVALID, "row_len > row_stride");
    if (device) {
        if ((v->row_stride * v->char_width) % 16 != 0)
            return fail(NEEDLE_ERR_INVALID, "row_stride * char_width must be a multiple of 16 bytes for device batches");
        if (((uintptr_t)v->rows) % 16 != 0) return fail(NEEDLE_ERR_INVALID, "rows must be 16-byte aligned");
        if (v->n_rows && v->row_stride == 0) return fail(NEEDLE_ERR_INVALID, "row_stride is 0");
    }
    return NEEDLE_OK;
}

// Host batches: per-row lengths are readable here, so an oversized one is an argument error, not an out-of-bounds
// read on the device (the kernels derive their chunk counts from the lengths and trust len <= row_stride).
static int check_host_lengths(const needle_batch_view *v) {
    if (!v->lengths) return NEEDLE_OK;
    for (uint64_t r = 0; r < v->n_rows; ++r)
        if (v->lengths[r] > v->row_stride) return fail(NEEDLE_ERR_INVALID, "lengths[r] > row_stride");
    return NEEDLE_OK;
}

// 16-bit result offsets: what the API can tell about the longest row -- row_len, or with per-row lengths (device memory, not
// readable here) the stride, which may be the caller's limit rounded up to the 16-byte alignment of device rows (65 536); the
// lengths themselves must stay within `limit` (the host entry points check them).
static bool offsets16_ok(const needle_batch_view *v, uint32_t limit) {
    return v->lengths ? v->row_stride <= 65536u : v->row_len <= limit;
}

// Few, long rows: one row per lane would leave the chip idle.  Packed-mode automata take the stripe path (function
// composition across 4 KiB stripes, needle_stripe.hip); NEEDLE_LONG_ROWS=0 turns it off, =1 forces it (tests).
static bool wants_stripe_path(const needle_batch_view *v, const ProgHeader &hdr, bool has_cursors) {
    static const int force = getenv("NEEDLE_LONG_ROWS") ? atoi(getenv("NEEDLE_LONG_ROWS")) : -1;
    const uint64_t stride_bytes = v->row_stride * v->char_width;
    const bool wanted = force >= 0 ? force == 1 : (v->n_rows < 65536 && stride_bytes >= 8 * (uint64_t)kStripeBytes);
    return wanted && hdr.mode == MODE_PACK && !has_cursors;
}

static int get_program(needle_pattern *p, int which, int cw, int variant, const DevProgram **out, int *n_cus);

static int run_stripe_path(needle_pattern *p, int op, const needle_batch_view *v, const DevProgram *fp, int n_cus,
                           uint64_t *d_bitmap, int32_t *d_start, int32_t *d_end, void *stream) {
    const uint64_t stride_bytes = v->row_stride * v->char_width;
    StripeArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.rows = (const uint8_t *)v->rows;
    sa.n_rows = v->n_rows;
    sa.stride_bytes = stride_bytes;
    sa.row_len = v->row_len;
    sa.lengths = v->lengths;
    sa.prog = fp->d_blob;
    sa.hdr = fp->prog.hdr;
    sa.spr = (uint32_t)((stride_bytes + kStripeBytes - 1) / kStripeBytes);
    sa.bitmap = d_bitmap;
    sa.start = d_start;
    sa.end = d_end;
    sa.fixed_len = -1;
    sa.op = (uint32_t)op;
    if (op == OP_FIND) {
        sa.fixed_len = p->t.fixed_len;
        if (sa.fixed_len < 0) {
            const DevProgram *bp = nullptr;
            int rc = get_program(p, W_BACKWARDS, (int)v->char_width, 1, &bp, nullptr);
            if (rc) return rc;
            sa.bprog = bp->d_blob;
            sa.bhdr = bp->prog.hdr;
        }
    }
    // find(): pass 1 also marks the stripes that pass through an accepting state, and only the last such stripe of a row is walked
    // again for lastMatch (needle_stripe.hip).  NEEDLE_STRIPE_CAND=0: every stripe is (A/B, tests).
    static const bool cand_on = !(getenv("NEEDLE_STRIPE_CAND") && atoi(getenv("NEEDLE_STRIPE_CAND")) == 0);
    const size_t fn_bytes = ((size_t)sa.n_rows * sa.spr * 4 + 15) & ~(size_t)15;
    const bool cand = op == OP_FIND && cand_on;
    HIP_TRY(scratch_malloc((void **)&sa.fn, fn_bytes * (cand ? 2 : 1) + (cand ? (size_t)sa.n_rows * 4 : 0), (hipStream_t)stream));
    if (cand) {
        sa.cand = (uint32_t *)((uint8_t *)sa.fn + fn_bytes);
        sa.cand_stripe = (int32_t *)((uint8_t *)sa.fn + 2 * fn_bytes);
    }
    hipError_t e = launch_long_rows((int)v->char_width, sa, n_cus, (hipStream_t)stream);
    (void)scratch_free(sa.fn, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "launch_long_rows");
    return NEEDLE_OK;
}

// Flood watch of the filter kernel (DevProgram).  Text that passes the filter almost everywhere makes that kernel several times
// slower than the ordinary scan; every filter launch ADDS its candidates and KiB of text to two monotonic device counters, copied
// (one 8-byte copy) to pinned memory behind the kernel.  The next call of that program looks at what has arrived -- without waiting --
// and works on the DELTA against what the last evaluation saw: above 16 candidates per KiB the filter is suspended for the program's
// next 32 calls, doubling to 1024 while the text stays like that.  One mutex per program: the state is per pattern x device x op, NOT per
// stream -- concurrent streams share one verdict (and one backoff).  Answers are the same either way.
//   needle_pattern_set_prefilter(p, NEEDLE_PREFILTER_AUTO | _ON | _OFF) pins the decision (ON: never suspended; OFF: the ordinary kernels);
//   needle_pattern_prefilter_state() reports it.  Under HIP-graph capture the decision is the one taken at CAPTURE time and is replayed
//   as captured -- the stats copy is not captured (a captured graph neither feeds nor obeys the watch): pin the mode for captured work.
// false = this call takes the ordinary kernel.  NEEDLE_PREFILTER_WATCH=0: the watch never suspends.
static bool ngram_watch_allows(const needle_pattern *p, const DevProgram *fp) {
    const int mode = p->pf_mode.load();
    if (mode == 2) return false;
    static const bool watch_on = !(getenv("NEEDLE_PREFILTER_WATCH") && atoi(getenv("NEEDLE_PREFILTER_WATCH")) == 0);
    std::lock_guard<std::mutex> lk(fp->ng_mu);
    const uint64_t both = fp->h_ng_stats[0]; // (one aligned 8-byte read of what one 8-byte copy wrote)
    const uint32_t cand = (uint32_t)both, kib = (uint32_t)(both >> 32);
    const uint32_t d_cand = cand - fp->ng_seen_cand, d_kib = kib - fp->ng_seen_kib; // (unsigned: the counters may wrap)
    if (d_kib >= 1024u) {
        fp->ng_seen_cand = cand, fp->ng_seen_kib = kib;
        fp->ng_last_rate = (float)d_cand / (float)d_kib;
        if (mode == 1) {
            // (pinned ON: the rate is still reported; no suspension is scheduled for a later return to AUTO)
        } else if ((uint64_t)d_cand > 16ull * d_kib) {
            fp->ng_suspend = fp->ng_backoff;
            fp->ng_backoff = fp->ng_backoff < 1024 ? 2 * fp->ng_backoff : 1024;
        } else {
            fp->ng_backoff = 32;
        }
    }
    if (mode == 1 || !watch_on || fp->ng_suspend <= 0) {
        ++fp->ng_launches;
        return true;
    }
    --fp->ng_suspend;
    ++fp->ng_suspended_calls;
    return false;
}
// (behind the kernel, on the program's own copy stream; neither the host nor the caller's stream ever waits for it)
static hipError_t ngram_watch_after_launch(const DevProgram *fp, hipStream_t stream) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return hipSuccess; // (not captured: see above)
    std::lock_guard<std::mutex> lk(fp->ng_mu); // (one event per program: record / wait pairs of concurrent callers must not interleave)
    hipError_t e = hipEventRecord(fp->ng_ev, stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(fp->ng_stream, fp->ng_ev, 0);
    if (e == hipSuccess) e = hipMemcpyAsync((void *)fp->h_ng_stats, fp->d_ng_stats, 8, hipMemcpyDeviceToHost, fp->ng_stream);
    return e;
}

// UTF-16 rows behind a BYTE program's n-gram filter (needle_ngram.h narrow16).  The pattern must live on ONE page of the BMP: every char
// outside page P shares one class ("other": in no range of the pattern -- the class with the most chars), and some char P << 8 | sub of the
// page is of that class too.  Then a char outside the page behaves exactly as byte `sub` does in the program lowered from the tables
// REBASED to the page (get_program, cw = 1 | P << 8; page 0: the ordinary 8-bit program) -- the searching automata the filter runs have
// maxChar 0xFFFF, so "beyond maxChar" never comes into it.  ASCII / Latin-1 dictionaries: page 0, sub 0xFF; Cyrillic: page 4; ...
// NEEDLE_PREFILTER_UTF16=0: never.
struct Utf16Route {
    int page = -1, sub = 0;
};
static Utf16Route utf16_route(const needle_pattern *cp) {
    static const bool off = getenv("NEEDLE_PREFILTER_UTF16") && atoi(getenv("NEEDLE_PREFILTER_UTF16")) == 0;
    needle_pattern *p = const_cast<needle_pattern *>(cp);
    Utf16Route r;
    if (off) return r;
    std::lock_guard<std::mutex> lk(p->u16_mu);
    if (p->u16_state == 0) {
        p->u16_state = 1;
        const std::vector<uint8_t> &cm = p->t.class_map;
        if (cm.size() == 65536 && p->t.dfa[W_CONTAINED_IN].max_char == 0xFFFF && p->t.dfa[W_FORWARDS].max_char == 0xFFFF) {
            // classes with the same column in all four automata are one class here (the reference numbers every gap between two of the
            // pattern's ranges separately: "below 'a'" and "above 'z'" are two classes with identical columns)
            const int N = p->t.stride;
            std::vector<int> canon(256, 0);
            {
                std::vector<uint64_t> sig((size_t)N, 1469598103934665603ull);
                for (int w = 0; w < 4; ++w) {
                    const RefDfa &d = p->t.dfa[w];
                    for (int k = 0; k < N; ++k) {
                        uint64_t h = sig[(size_t)k];
                        for (int32_t st = 0; st < d.n_states; ++st) h = (h ^ (uint64_t)(uint16_t)d.table[(size_t)st * N + k]) * 1099511628211ull;
                        sig[(size_t)k] = h * 31u + (uint64_t)w;
                    }
                }
                for (int k = 0; k < N; ++k) {
                    canon[k] = k;
                    for (int j = 0; j < k; ++j) {
                        if (sig[(size_t)j] != sig[(size_t)k]) continue;
                        bool same = true; // (the hash only nominates: compared entry by entry)
                        for (int w = 0; w < 4 && same; ++w) {
                            const RefDfa &d = p->t.dfa[w];
                            for (int32_t st = 0; st < d.n_states && same; ++st) same = d.table[(size_t)st * N + j] == d.table[(size_t)st * N + k];
                        }
                        if (same) {
                            canon[k] = canon[j];
                            break;
                        }
                    }
                }
            }
            // (a char beyond an automaton's maxChar takes the `c > maxChar` exit there: only classes whose column is dead in that
            // automaton reach beyond it, so class equality covers it -- checked below for the chars the rule relies on)
            uint32_t n_of[256] = {0};
            for (uint8_t c : cm) ++n_of[canon[c]];
            int other = 0;
            for (int k = 1; k < 256; ++k)
                if (n_of[k] > n_of[other]) other = k;
            int page = -1;
            bool one = true;
            for (int c = 0; c < 65536 && one; ++c)
                if (canon[cm[c]] != other) {
                    if (page < 0) page = c >> 8;
                    else one = page == (c >> 8);
                }
            // "dead beyond maxChar" must be what the other class does anyway in the automata that have a maxChar below 0xFFFF
            for (int w = 0; w < 4 && one; ++w) {
                const RefDfa &d = p->t.dfa[w];
                if (d.max_char >= 0xFFFF) continue;
                for (int k = 0; k < N && one; ++k)
                    if (canon[k] == other)
                        for (int32_t st = 0; st < d.n_states && one; ++st) one = d.table[(size_t)st * N + k] < 0;
            }
            if (one && page >= 0) {
                for (int b = 255; b >= 0; --b)
                    if (canon[cm[(size_t)(page << 8) | b]] == other) {
                        p->u16_page = page, p->u16_sub = b;
                        break;
                    }
            }
        }
    }
    r.page = p->u16_page, r.sub = p->u16_sub;
    return r;
}

// NEEDLE_FIND_LENGTHS: 0 = find() always by forward + backward walks, 1 (default) = the "lengths" automaton where the ordinary
// program is a plain LDS table, 2 = also instead of the pair table (measured slower: DESIGN.md s4)
static bool find_lengths_for(uint32_t mode) {
    static const int level = getenv("NEEDLE_FIND_LENGTHS") ? atoi(getenv("NEEDLE_FIND_LENGTHS")) : 1;
    static const bool sparse_too = !(getenv("NEEDLE_FIND_LENGTHS_SPARSE") && atoi(getenv("NEEDLE_FIND_LENGTHS_SPARSE")) == 0);
    static const bool pair_too = !(getenv("NEEDLE_FIND_LENGTHS_PAIR") && atoi(getenv("NEEDLE_FIND_LENGTHS_PAIR")) == 0);
    return level > 0 && (mode == MODE_TABLE8 || mode == MODE_TABLE16 || (sparse_too && mode == MODE_SPARSE) || ((pair_too || level > 1) && mode == MODE_PAIR));
}

static int run_dev(const needle_pattern *cp, int op, const needle_batch_view *v, uint64_t *d_bitmap, int32_t *d_start,
                   int32_t *d_end, void *stream, const int32_t *d_from = nullptr, uint32_t *d_end_state = nullptr,
                   bool no_backward = false, uint32_t *d_packed = nullptr, bool packed8 = false);

// The WIDE filter (lower_filter_wide) stands in for the UTF-16 scan kernels where the ordinary UTF-16 program is NOT a plain LDS table
// (compressed automaton, hot rows + HBM table, HBM table): a latency-bound or collapsing walk.  NEEDLE_PREFILTER=2: for every
// automaton that allows a filter (tests, A/B), as for 8-bit rows.  NEEDLE_PREFILTER_WIDE=0: never.
static bool wide_filter_wanted(uint32_t ordinary_mode) {
    // (NEEDLE_PREFILTER_UTF16=0: no filter in front of UTF-16 rows at all -- the one-page route and this one)
    static const bool off = (getenv("NEEDLE_PREFILTER_WIDE") && atoi(getenv("NEEDLE_PREFILTER_WIDE")) == 0) ||
                            (getenv("NEEDLE_PREFILTER_UTF16") && atoi(getenv("NEEDLE_PREFILTER_UTF16")) == 0);
    if (off || ngram_level() <= 0) return false;
    return ordinary_mode == MODE_SPARSE || ordinary_mode == MODE_HYBRID || ordinary_mode == MODE_GLOBAL || ngram_level() > 1;
}

// Few, long rows of an automaton too big for function composition: speculative stripes (needle_stripe.hip).  Returns
// NEEDLE_OK with *done = false when the path does not apply or did not reach its fixpoint (the caller then walks the
// rows one lane each).
static int run_speculative_stripes(needle_pattern *p, int op, const needle_batch_view *v, uint64_t *d_bitmap, int32_t *d_start,
                                   int32_t *d_end, void *stream_, bool *done) {
    *done = false;
    static const int force = getenv("NEEDLE_LONG_ROWS") ? atoi(getenv("NEEDLE_LONG_ROWS")) : -1;
    const uint64_t stride_bytes = v->row_stride * v->char_width;
    // measured (scripts/mid_rows_table_rate.py): containedIn gains up to 60 000 rows; find breaks even around 20 000;
    // matches() usually dies in the first chars of a row, which only the lane path turns into an early exit
    const uint64_t max_rows = op == OP_CONTAINED_IN ? 65536 : op == OP_FIND ? 8192 : 256;
    const bool wanted = force >= 0 ? force == 1 : (v->n_rows < max_rows && stride_bytes >= 8 * (uint64_t)kStripeBytes);
    if (!wanted) return NEEDLE_OK;
    uint32_t stripe = kStripeBytes; // largest power of two <= 4 KiB that divides the row stride
    while (stripe > 256 && stride_bytes % stripe) stripe >>= 1;
    if (stride_bytes % stripe || stride_bytes / stripe < 2) return NEEDLE_OK;
    const int which = op == OP_MATCHES ? W_MATCHES : op == OP_CONTAINED_IN ? W_CONTAINED_IN : W_FORWARDS;
    if (op != OP_MATCHES && p->t.dfa[which].accepting[0]) return NEEDLE_OK; // an accepting start state makes every stripe start look like a match
    const DevProgram *gp = nullptr, *fp = nullptr, *bp = nullptr;
    int n_cus = 0;
    int rc = get_program(p, which, (int)v->char_width, 3, &gp, &n_cus); // HBM-table layout: column maps + uint16 table
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    SpecArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.rows = (const uint8_t *)v->rows;
    sa.n_rows = v->n_rows;
    sa.stride_bytes = stride_bytes;
    sa.stripe_bytes = stripe;
    sa.spr = (uint32_t)(stride_bytes / stripe);
    sa.char_width = v->char_width;
    sa.op = (uint32_t)op;
    sa.row_len = v->row_len;
    sa.lengths = v->lengths;
    sa.gprog = gp->d_blob;
    sa.hdr = gp->prog.hdr;
    const size_t ns = (size_t)(sa.n_rows * sa.spr), words = (ns + 63) / 64;
    // slen | spec_end_state | spec_last | spec_start(unused) | entry | entry_done | true_end_state | true_last  (4 B each), bitmap, flag
    uint8_t *tmp = nullptr;
    const size_t o_bm = 8 * ns * 4, o_flag = o_bm + words * 8, total = o_flag + 16;
    HIP_TRY(scratch_malloc((void **)&tmp, total, stream));
    auto finish = [&](int code) {
        (void)scratch_free(tmp, stream);
        return code;
    };
    uint32_t *u = (uint32_t *)tmp;
    sa.slen = u;
    uint32_t *spec_end_state = u + ns;
    int32_t *spec_last = (int32_t *)(u + 2 * ns), *spec_start = (int32_t *)(u + 3 * ns);
    sa.spec_end_state = spec_end_state;
    sa.spec_last = spec_last;
    sa.entry = u + 4 * ns;
    sa.entry_done = u + 5 * ns;
    sa.true_end_state = u + 6 * ns;
    sa.true_last = (int32_t *)(u + 7 * ns);
    sa.spec_bitmap = (const uint64_t *)(tmp + o_bm);
    sa.changed = (int32_t *)(tmp + o_flag);
    sa.bitmap = d_bitmap;
    sa.end = d_end;
    hipError_t e = launch_spec_len(sa, stream);
    if (e != hipSuccess) return finish(hip_fail(e, "spec_len"));
    // pass 1: every stripe as a row of its own, from the start state, through the tiled kernel
    needle_batch_view sv;
    memset(&sv, 0, sizeof(sv));
    sv.rows = v->rows;
    sv.char_width = v->char_width;
    sv.n_rows = ns;
    sv.row_stride = stripe / v->char_width;
    sv.lengths = sa.slen;
    rc = run_dev(p, op, &sv, (uint64_t *)(tmp + o_bm), spec_start, spec_last, stream_, nullptr, spec_end_state, true);
    if (rc) return finish(rc);
    e = launch_spec_init(sa, stream);
    if (e != hipSuccess) return finish(hip_fail(e, "spec_init"));
    bool fixed = false;
    for (int round = 0; round < 48 && !fixed; ++round) {
        if (hipMemsetAsync(sa.changed, 0, 4, stream) != hipSuccess) return finish(fail(NEEDLE_ERR_DEVICE, "hipMemsetAsync"));
        e = launch_spec_fix(sa, stream);
        int32_t changed = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&changed, sa.changed, 4, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return finish(hip_fail(e, "spec_fix"));
        fixed = changed == 0;
    }
    if (!fixed) return finish(NEEDLE_OK); // e.g. a DOTALL `.*` tail: every round settles one more stripe only
    if (op != OP_FIND && hipMemsetAsync(d_bitmap, 0, ((sa.n_rows + 63) / 64) * 8, stream) != hipSuccess)
        return finish(fail(NEEDLE_ERR_DEVICE, "hipMemsetAsync"));
    e = launch_spec_reduce(sa, stream);
    if (e != hipSuccess) return finish(hip_fail(e, "spec_reduce"));
    if (op == OP_FIND) { // matched bits + start: indexBackwards from lastMatch, one lane per row
        rc = get_program(p, W_FORWARDS, (int)v->char_width, p->t.fixed_len < 0 ? 2 : 0, &fp, nullptr);
        if (rc) return finish(rc);
        StripeArgs ba;
        memset(&ba, 0, sizeof(ba));
        ba.rows = (const uint8_t *)v->rows;
        ba.n_rows = v->n_rows;
        ba.stride_bytes = stride_bytes;
        ba.prog = fp->d_blob;
        ba.hdr = fp->prog.hdr;
        ba.bitmap = d_bitmap;
        ba.start = d_start;
        ba.end = d_end;
        ba.fixed_len = p->t.fixed_len;
        ba.op = OP_FIND;
        if (ba.fixed_len < 0) {
            rc = get_program(p, W_BACKWARDS, (int)v->char_width, 1, &bp, nullptr);
            if (rc) return finish(rc);
            ba.bprog = bp->d_blob;
            ba.bhdr = bp->prog.hdr;
        }
        e = launch_backward_rows((int)v->char_width, ba, stream);
        if (e != hipSuccess) return finish(hip_fail(e, "backward_rows"));
    }
    *done = true;
    return finish(NEEDLE_OK);
}

// Launch arguments of the filter kernel (needle_ngram.hip) for program `tp` on batch `v`.  stride: bytes between rows -- or CHARS, for UTF-16
// rows behind the byte program (launch_ngram with char_width 2 scales the addresses).
static ScanArgs filter_scan_args(const needle_batch_view *v, uint64_t stride, const DevProgram *tp, int32_t fixed_len, uint64_t *d_bitmap, int32_t *d_start,
                                 int32_t *d_end, uint32_t *d_packed, bool packed8 = false) {
    ScanArgs a;
    memset(&a, 0, sizeof(a));
    a.rows = (const uint8_t *)v->rows;
    a.n_rows = v->n_rows;
    a.stride_bytes = stride;
    a.total_bytes = a.n_rows * a.stride_bytes;
    a.row_len = v->row_len;
    a.lengths = v->lengths;
    a.prog = tp->d_blob;
    a.hdr = tp->prog.hdr;
    a.fixed_len = fixed_len;
    a.bitmap = d_bitmap;
    a.start = d_start;
    a.end = d_end;
    a.packed = d_packed;
    a.packed8 = packed8 ? 1u : 0u;
    return a;
}

// d_packed (OP_FIND, needle_find_packed16_dev): a row's start / end go there as one dword, stored by the scan kernel itself;
// d_start / d_end are not used.  The paths for few long rows (stripes) and the opt-in two-row-set kernel keep their int32
// arrays: they run into scratch and one pack pass follows.
static int run_dev(const needle_pattern *cp, int op, const needle_batch_view *v, uint64_t *d_bitmap, int32_t *d_start,
                   int32_t *d_end, void *stream, const int32_t *d_from, uint32_t *d_end_state, bool no_backward, uint32_t *d_packed, bool packed8) {
    needle_pattern *p = const_cast<needle_pattern *>(cp);
    if (!p) return fail(NEEDLE_ERR_INVALID, "pattern is NULL");
    int rc = check_view(v, true);
    if (rc) return rc;
    if (v->n_rows == 0) return NEEDLE_OK;
    if (!d_bitmap) return fail(NEEDLE_ERR_INVALID, "bitmap is NULL");
    if (op == OP_FIND && !d_packed && (!d_start || !d_end)) return fail(NEEDLE_ERR_INVALID, "start/end is NULL");
#ifdef NEEDLE_TUNING
    static const int dict_env = getenv("NEEDLE_DICT") ? atoi(getenv("NEEDLE_DICT")) : 0;
#else
    constexpr int dict_env = 0;
#endif
    // (NEEDLE_LONG_ROWS=1 forces the stripe paths for any stride: they only know the int32 arrays)
    static const bool long_rows_forced = getenv("NEEDLE_LONG_ROWS") && atoi(getenv("NEEDLE_LONG_ROWS")) == 1;
    if (d_packed && (dict_env > 0 || long_rows_forced || v->row_stride * v->char_width >= 8 * (uint64_t)kStripeBytes)) {
        if (packed8) return fail(NEEDLE_ERR_UNSUPPORTED, "needle_find_packed8_dev: not on the stripe / two-row-set paths (tuning switches)");
        int32_t *tmp = nullptr;
        HIP_TRY(scratch_malloc((void **)&tmp, (size_t)v->n_rows * 8, (hipStream_t)stream));
        rc = run_dev(cp, op, v, d_bitmap, tmp, tmp + v->n_rows, stream, d_from, d_end_state, no_backward, nullptr);
        if (rc == NEEDLE_OK) rc = needle_pack_start_end16_dev(tmp, tmp + v->n_rows, v->n_rows, d_packed, stream);
        (void)scratch_free(tmp, (hipStream_t)stream);
        return rc;
    }

    const int which = op == OP_MATCHES ? W_MATCHES : op == OP_CONTAINED_IN ? W_CONTAINED_IN : W_FORWARDS;
    const DevProgram *fp = nullptr, *bp = nullptr;
    int n_cus = 0;
    const bool need_backward = op == OP_FIND && p->t.fixed_len < 0;
    // UTF-16 rows (Java's strings) of a pattern that lives on one page of the BMP -- ASCII / Latin-1, Cyrillic, Greek ... dictionaries: behind
    // the n-gram filter of the BYTE program of that page, the text narrowed as it is loaded (utf16_route above; needle_ngram.h narrow16).
    // The program is chosen as for 8-bit rows below; whatever rules the filter out there (no filter for this automaton, the shape, the
    // flood watch) leaves these rows to the UTF-16 kernels.
    const Utf16Route u16 = v->char_width == 2 ? utf16_route(p) : Utf16Route();
    if (v->char_width == 2 && op != OP_MATCHES && !d_from && !d_end_state && !no_backward && ngram_level() > 0 && dict_env == 0 && u16.page >= 0 &&
        v->row_stride * 2 < 8 * (uint64_t)kStripeBytes) do {
        // (every step of this route is speculative -- lowering and uploading the page's byte programs for a pattern that may have no
        // filter at all: a failure here means "route unavailable", the UTF-16 kernels below serve the call)
        const DevProgram *tp = nullptr;
        const int cw8 = 1 | (u16.page << 8); // the byte program of the pattern's page
        rc = get_program(p, which, cw8, need_backward ? 2 : 0, &tp, &n_cus);
        if (rc) break;
        bool ok = false;
        if (tp->prog.hdr.mode == MODE_HYBRID || tp->prog.hdr.mode == MODE_GLOBAL) {
            rc = get_program(p, which, cw8, 9, &tp, nullptr);
            if (rc) break;
            ok = tp && tp->d_ng && tp->prog.ng.p.on && (op == OP_CONTAINED_IN || tp->prog.hdr.fa_len_off || p->t.fixed_len >= 0);
        } else {
            bool lengths8 = false;
            if (need_backward && find_lengths_for(tp->prog.hdr.mode)) {
                const DevProgram *lp = nullptr;
                rc = get_program(p, W_FORWARDS, cw8, 7, &lp, nullptr);
                if (rc) break;
                if (lp && !(tp->prog.hdr.mode == MODE_PAIR && lp->prog.hdr.mode != MODE_PAIR)) tp = lp, lengths8 = true;
            }
            ok = tp->d_ng && tp->prog.ng.p.on && (op == OP_CONTAINED_IN || lengths8 || p->t.fixed_len >= 0);
        }
        if (ok) {
            const ScanArgs a = filter_scan_args(v, v->row_stride /* chars */, tp, op == OP_FIND ? p->t.fixed_len : -1, d_bitmap, d_start, d_end, d_packed, packed8);
            if (ngram_shape_ok(a) && ngram_lds_bytes(a.hdr, tp->prog.ng.p) && ngram_watch_allows(p, tp)) {
                HIP_TRY(launch_ngram(op, a, tp->prog.ng.p, tp->d_ng, tp->d_ng_stats, n_cus, (hipStream_t)stream, 2, u16.page, u16.sub));
                HIP_TRY(ngram_watch_after_launch(tp, (hipStream_t)stream));
                return NEEDLE_OK;
            }
        }
    } while (0);
    rc = get_program(p, which, (int)v->char_width, need_backward ? 2 : 0, &fp, &n_cus);
    if (rc) return rc;
    // UTF-16 rows of a pattern that lives on SEVERAL pages of the BMP (Latin + Cyrillic + CJK dictionaries: DFA.java:438-463, the reference's
    // class map covers every code unit of any pattern) and whose automaton is too big for a plain LDS table: the WIDE filter -- windows of
    // four 16-bit code units hashed as they stand, candidates verified on the UTF-16 HBM-table program (lower_filter_wide).  Without it these
    // rows take hot rows + HBM table: 9.3 ms on the 10M-row batch where the one-page route runs at 1.1.  NEEDLE_PREFILTER_WIDE=0: never.
    if (v->char_width == 2 && u16.page < 0 && wide_filter_wanted(fp->prog.hdr.mode) && op != OP_MATCHES && !d_from && !d_end_state && !no_backward &&
        dict_env == 0 && v->row_stride * 2 < 8 * (uint64_t)kStripeBytes) {
        const DevProgram *tp = nullptr;
        rc = get_program(p, which, 2, 10, &tp, nullptr);
        if (rc) return rc;
        if (tp && tp->d_ng && tp->prog.ng.p.on && (op == OP_CONTAINED_IN || tp->prog.hdr.fa_len_off || p->t.fixed_len >= 0)) {
            const ScanArgs a = filter_scan_args(v, v->row_stride /* chars */, tp, op == OP_FIND ? p->t.fixed_len : -1, d_bitmap, d_start, d_end, d_packed, packed8);
            if (ngram_shape_ok(a) && ngram_lds_bytes(a.hdr, tp->prog.ng.p) && ngram_watch_allows(p, tp)) {
                HIP_TRY(launch_ngram(op, a, tp->prog.ng.p, tp->d_ng, tp->d_ng_stats, n_cus, (hipStream_t)stream, 2, 0, 0));
                HIP_TRY(ngram_watch_after_launch(tp, (hipStream_t)stream));
                return NEEDLE_OK;
            }
        }
    }
    if (d_end_state && (fp->prog.hdr.mode == MODE_HYBRID || fp->prog.hdr.mode == MODE_SPARSE)) {
        // the speculative-stripe pass wants every stripe's end state in the numbering of the HBM-table layout its fix-up
        // kernel walks (variant 3); the hot-rows and compressed forms number / encode states their own way
        rc = get_program(p, which, (int)v->char_width, 3, &fp, &n_cus);
        if (rc) return rc;
    }
    if (!d_end_state && wants_stripe_path(v, fp->prog.hdr, d_from != nullptr)) return run_stripe_path(p, op, v, fp, n_cus, d_bitmap, d_start, d_end, stream);
    if (!d_end_state && !d_from && fp->prog.hdr.mode != MODE_PACK) {
        bool done = false;
        rc = run_speculative_stripes(p, op, v, d_bitmap, d_start, d_end, stream, &done);
        if (rc || done) return rc;
    }
    // the tiled kernel forms per-lane row offsets in 32 bits (up to 63 x stride); only the stripe path above takes
    // rows of tens of megabytes and more
    if (v->row_stride * v->char_width >= (1ull << 26))
        return fail(NEEDLE_ERR_UNSUPPORTED, "rows of 64 MiB or more are only supported on the stripe paths (automata of at most 5 states, or ones that re-synchronise; not with NEEDLE_LONG_ROWS=0, per-row cursors or empty-matching patterns)");
    // An automaton that fits the LDS in no form (hot rows + HBM table, or the HBM table alone): containedIn() / find() behind the
    // n-gram candidate filter with the verify walks out of HBM / L2 (lower_filter_hbm) -- the per-char walk of such an automaton
    // collapses on text that leaves the hot states (near-miss rows: 20 ms on the 10M-row batch), the filter's does not.
    if ((fp->prog.hdr.mode == MODE_HYBRID || fp->prog.hdr.mode == MODE_GLOBAL) && v->char_width == 1 && op != OP_MATCHES && !d_from && !d_end_state &&
        !no_backward && ngram_level() > 0 && dict_env == 0) {
        const DevProgram *tp = nullptr;
        rc = get_program(p, which, 1, 9, &tp, nullptr);
        if (rc) return rc;
        // find() without bounded match lengths (no lengths form): the forward search automaton + backward walks for the starts (variant 12)
        static const bool unbounded_on = !(getenv("NEEDLE_PREFILTER_UNBOUNDED") && atoi(getenv("NEEDLE_PREFILTER_UNBOUNDED")) == 0);
        const DevProgram *bwp = nullptr;
        if (!tp && op == OP_FIND && need_backward && unbounded_on) {
            rc = get_program(p, which, 1, 12, &tp, nullptr);
            if (rc) return rc;
            if (tp) {
                rc = get_program(p, W_BACKWARDS, 1, 1, &bwp, nullptr);
                if (rc) return rc;
                if (!bwp) tp = nullptr;
            }
        }
        if (tp && tp->d_ng && tp->prog.ng.p.on && (op == OP_CONTAINED_IN || tp->prog.hdr.fa_len_off || p->t.fixed_len >= 0 || bwp)) {
            ScanArgs a = filter_scan_args(v, v->row_stride, tp, op == OP_FIND ? p->t.fixed_len : -1, d_bitmap, d_start, d_end, d_packed, packed8);
            if (bwp) a.bprog = bwp->d_blob, a.bhdr = bwp->prog.hdr;
            if (ngram_shape_ok(a) && ngram_lds_bytes(a.hdr, tp->prog.ng.p) && ngram_watch_allows(p, tp)) {
                HIP_TRY(launch_ngram(op, a, tp->prog.ng.p, tp->d_ng, tp->d_ng_stats, n_cus, (hipStream_t)stream));
                HIP_TRY(ngram_watch_after_launch(tp, (hipStream_t)stream));
                return NEEDLE_OK;
            }
        }
    }
    // find() whose pattern allows it: the "lengths" automaton -- the state the walk stops in remembers how long the match was,
    // start = end - pend[state], no indexBackwards, no text snapshots (needle_lower.h).  Taken where the ordinary program is a
    // plain LDS table (the modes pend[] can be indexed in).  NEEDLE_FIND_LENGTHS=0: off (A/B, tests).
    bool lengths_form = false;
    if (need_backward && !d_end_state && !no_backward && find_lengths_for(fp->prog.hdr.mode)) {
        const DevProgram *lp = nullptr;
        rc = get_program(p, W_FORWARDS, (int)v->char_width, 7, &lp, nullptr);
        if (rc) return rc;
        // (a pair-table automaton whose lengths program no longer fits the pair table keeps its two walks: two chars per lookup
        // beat the saved backward walk -- NEEDLE_FIND_LENGTHS=2 takes the plain table all the same)
        static const bool force_tables = getenv("NEEDLE_FIND_LENGTHS") && atoi(getenv("NEEDLE_FIND_LENGTHS")) > 1;
        if (lp && fp->prog.hdr.mode == MODE_PAIR && lp->prog.hdr.mode != MODE_PAIR && !force_tables) lp = nullptr;
        if (lp) fp = lp, lengths_form = true;
    }
    ScanArgs a;
    memset(&a, 0, sizeof(a));
    a.rows = (const uint8_t *)v->rows;
    a.n_rows = v->n_rows;
    a.stride_bytes = v->row_stride * v->char_width;
    a.total_bytes = a.n_rows * a.stride_bytes;
    a.row_len = v->row_len;
    a.lengths = v->lengths;
    a.from = d_from;
    a.prog = fp->d_blob;
    a.hdr = fp->prog.hdr;
    a.fixed_len = -1;
    if (op == OP_FIND) {
        a.fixed_len = p->t.fixed_len;
        if (a.fixed_len < 0 && !lengths_form) {
            rc = get_program(p, W_BACKWARDS, (int)v->char_width, 1, &bp, nullptr);
            if (rc) return rc;
            a.bprog = bp->d_blob;
            a.bhdr = bp->prog.hdr;
        }
    }
    a.bitmap = d_bitmap;
    a.start = d_start;
    a.end = d_end;
    a.packed = d_packed;
    a.packed8 = packed8 ? 1u : 0u;
    a.end_state = d_end_state;
    bool skip_backward = no_backward; // (speculative pass: only lastMatch is wanted)
#ifdef NEEDLE_TUNING // measurement builds only (scripts/build_tuning.sh): a switch that changes ANSWERS (start = end) never ships
    static const bool dbg_no_backward = getenv("NEEDLE_DEBUG_NO_BACKWARD") != nullptr;
    skip_backward = skip_backward || dbg_no_backward;
#endif
    if (skip_backward && op == OP_FIND) a.fixed_len = 0, a.bprog = nullptr;
#ifdef NEEDLE_TUNING
    // Big automata on full 8-bit rows: two 64-row sets per wave (needle_dict.hip) over the whole 128-row pairs of the batch, the
    // ordinary kernel on what is left; find()'s starts by indexBackwards afterwards, one lane per matched row.
    // Measurement builds only (scripts/build_tuning.sh).  NEEDLE_DICT: 0 off (default: measured, it is no faster -- DESIGN.md s4),
    // 1 on for the compressed automaton, 2 also for plain uint16 LDS tables.
    if (dict_env > 0 && !lengths_form && (a.hdr.mode == MODE_SPARSE || dict_env > 1) && dict_kernel_applies((int)v->char_width, a)) {
        HIP_TRY(launch_dict(op, a, n_cus, (hipStream_t)stream));
        const uint64_t done_rows = (a.n_rows >> 7) << 7;
        if (op == OP_FIND && a.fixed_len < 0 && done_rows) {
            StripeArgs ba;
            memset(&ba, 0, sizeof(ba));
            ba.rows = a.rows;
            ba.n_rows = done_rows;
            ba.stride_bytes = a.stride_bytes;
            ba.prog = fp->d_blob;
            ba.hdr = fp->prog.hdr;
            ba.bitmap = d_bitmap;
            ba.start = d_start;
            ba.end = d_end;
            ba.fixed_len = -1;
            ba.op = OP_FIND;
            ba.bprog = bp->d_blob;
            ba.bhdr = bp->prog.hdr;
            HIP_TRY(launch_backward_rows((int)v->char_width, ba, (hipStream_t)stream));
        }
        if (done_rows == a.n_rows) return NEEDLE_OK;
        a.rows += done_rows * a.stride_bytes; // the last n_rows % 128 rows
        a.n_rows -= done_rows;
        a.total_bytes = a.n_rows * a.stride_bytes;
        a.bitmap += done_rows >> 6;
        if (a.start) a.start += done_rows, a.end += done_rows;
    }
#endif
    // The n-gram candidate filter (needle_ngram.hip): the automaton only runs where a hashed 4-byte window of the
    // text can stand ahead of a match.  For programs whose lowering established that this gives the reference's answers
    // (needle_ngram_host.cpp), on containedIn() and on find() whose start is end - length (lengths programs, one-length patterns).
    // (find() of a pattern WITHOUT bounded match lengths -- `(kw1|..|kw1000)[0-9]+` -- takes it too: its verified candidates find their
    // starts by indexBackwards, the lock-step backward walk on text out of L2.  NEEDLE_PREFILTER_UNBOUNDED=0: the scan kernel)
    static const bool unbounded_on = !(getenv("NEEDLE_PREFILTER_UNBOUNDED") && atoi(getenv("NEEDLE_PREFILTER_UNBOUNDED")) == 0);
    const bool by_backward_walk = op == OP_FIND && a.fixed_len < 0 && !lengths_form && a.bprog != nullptr && unbounded_on && !d_from;
    if (fp->d_ng && fp->prog.ng.p.on && ngram_level() > 0 && v->char_width == 1 && op != OP_MATCHES && !skip_backward &&
        (op == OP_CONTAINED_IN || lengths_form || a.fixed_len >= 0 || by_backward_walk) && ngram_shape_ok(a) && ngram_lds_bytes(a.hdr, fp->prog.ng.p)) {
        if (ngram_watch_allows(p, fp)) {
            HIP_TRY(launch_ngram(op, a, fp->prog.ng.p, fp->d_ng, fp->d_ng_stats, n_cus, (hipStream_t)stream));
            HIP_TRY(ngram_watch_after_launch(fp, (hipStream_t)stream));
            return NEEDLE_OK;
        }
    }
    HIP_TRY(launch_scan(op, (int)v->char_width, a, n_cus, (hipStream_t)stream));
    return NEEDLE_OK;
}

// Small host batches (above all the one-row batches of the Matcher mirror): one grow-only device arena + pinned
// staging buffer + stream per host thread, ONE upload and ONE download per call -- instead of five hipMalloc/hipFree
// pairs and as many synchronous copies.
namespace {
struct HostArena {
    int dev = -1;
    uint8_t *d = nullptr, *h = nullptr;
    size_t cap = 0;
    hipStream_t stream = nullptr;
    ~HostArena() { release(); }
    void release() {
        if (d) (void)hipFree(d);
        if (h) (void)hipHostFree(h);
        if (stream) (void)hipStreamDestroy(stream);
        d = h = nullptr;
        stream = nullptr;
        cap = 0;
        dev = -1;
    }
    hipError_t reserve(size_t bytes) {
        int cur = 0;
        hipError_t e = hipGetDevice(&cur);
        if (e != hipSuccess) return e;
        if (cur == dev && bytes <= cap) return hipSuccess;
        release();
        size_t want = 1 << 16;
        while (want < bytes) want <<= 1;
        if ((e = hipMalloc((void **)&d, want)) != hipSuccess) return e;
        if ((e = hipHostMalloc((void **)&h, want, hipHostMallocMapped)) != hipSuccess) return e;
        if ((e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)) != hipSuccess) return e;
        cap = want;
        dev = cur;
        return hipSuccess;
    }
};
constexpr size_t kSmallHostBatchBytes = 4u << 20;
constexpr size_t kZeroCopyBytes = 16u << 10;
} // namespace

static int run_host_small(const needle_pattern *p, int op, const needle_batch_view *v, uint64_t dst_stride, uint64_t *bitmap,
                          int32_t *start, int32_t *end) {
    static thread_local HostArena arena;
    const size_t cw = v->char_width, n = (size_t)v->n_rows;
    const size_t src_stride = (size_t)v->row_stride * cw;
    const size_t words = (n + 63) / 64;
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    // in: rows | lengths      out: bitmap | start | end
    const size_t o_rows = 0, o_len = up16(n * dst_stride), in_bytes = o_len + (v->lengths ? up16(n * 4) : 0);
    const size_t o_bm = in_bytes, o_s = o_bm + up16(words * 8), o_e = o_s + up16(n * 4), total = o_e + up16(n * 4);
    HIP_TRY(arena.reserve(total));
    if (dst_stride == src_stride) {
        memcpy(arena.h + o_rows, v->rows, n * src_stride);
    } else {
        for (size_t r = 0; r < n; ++r) {
            memcpy(arena.h + o_rows + r * dst_stride, (const uint8_t *)v->rows + r * src_stride, src_stride);
            memset(arena.h + o_rows + r * dst_stride + src_stride, 0, dst_stride - src_stride);
        }
    }
    if (v->lengths) memcpy(arena.h + o_len, v->lengths, n * 4);
    // Tiny batches (one Matcher call): the kernel reads the pinned staging buffer and writes its results there
    // directly over PCIe -- one launch and one wait, no copy commands at all.
    const bool zero_copy = total <= kZeroCopyBytes;
    uint8_t *base = arena.d;
    if (zero_copy) {
        void *mapped = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&mapped, arena.h, 0));
        base = (uint8_t *)mapped;
    } else {
        HIP_TRY(hipMemcpyAsync(arena.d, arena.h, in_bytes, hipMemcpyHostToDevice, arena.stream));
    }
    needle_batch_view dv = *v;
    dv.rows = base + o_rows;
    dv.lengths = v->lengths ? (const uint32_t *)(base + o_len) : nullptr;
    dv.row_stride = dst_stride / cw;
    int rc = run_dev(p, op, &dv, (uint64_t *)(base + o_bm), (int32_t *)(base + o_s), (int32_t *)(base + o_e), arena.stream);
    if (rc) return rc;
    if (!zero_copy) {
        const size_t out_bytes = op == OP_FIND ? total - o_bm : up16(words * 8);
        HIP_TRY(hipMemcpyAsync(arena.h + o_bm, arena.d + o_bm, out_bytes, hipMemcpyDeviceToHost, arena.stream));
    }
    HIP_TRY(hipStreamSynchronize(arena.stream));
    memcpy(bitmap, arena.h + o_bm, words * 8);
    if (op == OP_FIND) {
        memcpy(start, arena.h + o_s, n * 4);
        memcpy(end, arena.h + o_e, n * 4);
    }
    return NEEDLE_OK;
}

// Host-buffer convenience: pad rows to a 16-byte stride, upload, run, download.
static int run_host_one(const needle_pattern *p, int op, const needle_batch_view *v, uint64_t *bitmap, int32_t *start,
                        int32_t *end);

// Host batches of any size: at most kHostChunkBytes of rows are resident on the device at a time (chunks start on
// 64-row boundaries, so every chunk owns whole bitmap words).
static int run_host(const needle_pattern *p, int op, const needle_batch_view *v, uint64_t *bitmap, int32_t *start,
                    int32_t *end) {
    int rc = check_view(v, false);
    if (rc) return rc;
    if ((rc = check_host_lengths(v))) return rc;
    static const uint64_t kHostChunkBytes = getenv("NEEDLE_HOST_CHUNK_BYTES") ? (uint64_t)atoll(getenv("NEEDLE_HOST_CHUNK_BYTES")) : (2ull << 30);
    const uint64_t row_bytes = std::max<uint64_t>(16, (v->row_stride * v->char_width + 15) & ~(uint64_t)15);
    uint64_t per = std::max<uint64_t>(64, (kHostChunkBytes / row_bytes) & ~(uint64_t)63);
    if (v->n_rows <= per) return run_host_one(p, op, v, bitmap, start, end);
    for (uint64_t r0 = 0; r0 < v->n_rows; r0 += per) {
        needle_batch_view c = *v;
        c.n_rows = std::min<uint64_t>(per, v->n_rows - r0);
        c.rows = (const uint8_t *)v->rows + r0 * v->row_stride * v->char_width;
        c.lengths = v->lengths ? v->lengths + r0 : nullptr;
        rc = run_host_one(p, op, &c, bitmap ? bitmap + r0 / 64 : nullptr, start ? start + r0 : nullptr, end ? end + r0 : nullptr);
        if (rc) return rc;
    }
    return NEEDLE_OK;
}

static int run_host_one(const needle_pattern *p, int op, const needle_batch_view *v, uint64_t *bitmap, int32_t *start,
                        int32_t *end) {
    int rc = check_view(v, false);
    if (rc) return rc;
    if (v->n_rows == 0) return NEEDLE_OK;
    if (!bitmap) return fail(NEEDLE_ERR_INVALID, "bitmap is NULL");
    const size_t cw = v->char_width;
    const uint64_t src_stride = v->row_stride * cw;
    uint64_t dst_stride = (src_stride + 15) & ~(uint64_t)15;
    if (dst_stride == 0) dst_stride = 16;
    const size_t words = (v->n_rows + 63) / 64;
    if (op == OP_FIND && (!start || !end)) return fail(NEEDLE_ERR_INVALID, "start/end is NULL");
    if (v->n_rows * dst_stride + v->n_rows * 16 <= kSmallHostBatchBytes) return run_host_small(p, op, v, dst_stride, bitmap, start, end);
    uint8_t *d_rows = nullptr;
    uint32_t *d_len = nullptr;
    uint64_t *d_bm = nullptr;
    int32_t *d_s = nullptr, *d_e = nullptr;
    auto cleanup = [&]() {
        if (d_rows) (void)hipFree(d_rows);
        if (d_len) (void)hipFree(d_len);
        if (d_bm) (void)hipFree(d_bm);
        if (d_s) (void)hipFree(d_s);
        if (d_e) (void)hipFree(d_e);
    };
#define HIP_TRY_C(expr)                                    \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) {                            \
            cleanup();                                     \
            return hip_fail(_e, #expr);                    \
        }                                                  \
    } while (0)
    HIP_TRY_C(hipMalloc((void **)&d_rows, v->n_rows * dst_stride));
    if (dst_stride == src_stride) {
        HIP_TRY_C(hipMemcpy(d_rows, v->rows, v->n_rows * src_stride, hipMemcpyHostToDevice));
    } else {
        HIP_TRY_C(hipMemset(d_rows, 0, v->n_rows * dst_stride));
        if (src_stride)
            HIP_TRY_C(hipMemcpy2D(d_rows, dst_stride, v->rows, src_stride, src_stride, v->n_rows, hipMemcpyHostToDevice));
    }
    if (v->lengths) {
        HIP_TRY_C(hipMalloc((void **)&d_len, v->n_rows * 4));
        HIP_TRY_C(hipMemcpy(d_len, v->lengths, v->n_rows * 4, hipMemcpyHostToDevice));
    }
    HIP_TRY_C(hipMalloc((void **)&d_bm, words * 8));
    if (op == OP_FIND) {
        if (!start || !end) {
            cleanup();
            return fail(NEEDLE_ERR_INVALID, "start/end is NULL");
        }
        HIP_TRY_C(hipMalloc((void **)&d_s, v->n_rows * 4));
        HIP_TRY_C(hipMalloc((void **)&d_e, v->n_rows * 4));
    }
    needle_batch_view dv = *v;
    dv.rows = d_rows;
    dv.lengths = d_len;
    dv.row_stride = dst_stride / cw;
    rc = run_dev(p, op, &dv, d_bm, d_s, d_e, nullptr);
    if (rc) {
        cleanup();
        return rc;
    }
    HIP_TRY_C(hipDeviceSynchronize());
    HIP_TRY_C(hipMemcpy(bitmap, d_bm, words * 8, hipMemcpyDeviceToHost));
    if (op == OP_FIND) {
        HIP_TRY_C(hipMemcpy(start, d_s, v->n_rows * 4, hipMemcpyDeviceToHost));
        HIP_TRY_C(hipMemcpy(end, d_e, v->n_rows * 4, hipMemcpyDeviceToHost));
    }
    cleanup();
    return NEEDLE_OK;
#undef HIP_TRY_C
}

static int check_packed(const needle_packed_view *v) {
    if (!v) return fail(NEEDLE_ERR_INVALID, "packed view is NULL");
    if (v->char_width != 1 && v->char_width != 2) return fail(NEEDLE_ERR_INVALID, "char_width must be 1 or 2");
    if (!v->offsets) return fail(NEEDLE_ERR_INVALID, "offsets is NULL");
    return NEEDLE_OK;
}

static int run_packed_host_one(const needle_pattern *p, int op, const needle_packed_view *v, uint64_t *bitmap,
                               int32_t *start, int32_t *end);

// Packed host batch.  The fixed-stride layout the kernels read pads every row to the longest one: harmless when the
// lengths are alike, ruinous when one 1 MB document sits among a million 40-char strings.  Rows are therefore grouped
// into length classes (stride 64 B, 256 B, 1 KiB, ... x4) and every class runs as its own batch, so the padded
// bytes stay below 4x the text.
static int run_packed_host(const needle_pattern *p, int op, const needle_packed_view *v, uint64_t *bitmap, int32_t *start,
                           int32_t *end) {
    if (!p) return fail(NEEDLE_ERR_INVALID, "pattern is NULL");
    int rc = check_packed(v);
    if (rc) return rc;
    if (v->n_rows == 0) return NEEDLE_OK;
    if (!bitmap) return fail(NEEDLE_ERR_INVALID, "bitmap is NULL");
    if (op == OP_FIND && (!start || !end)) return fail(NEEDLE_ERR_INVALID, "start/end is NULL");
    const uint64_t cw = v->char_width, n = v->n_rows;
    uint64_t max_len = 0;
    for (uint64_t r = 0; r < n; ++r) {
        if (v->offsets[r + 1] < v->offsets[r]) return fail(NEEDLE_ERR_INVALID, "offsets must be non-decreasing");
        max_len = std::max<uint64_t>(max_len, v->offsets[r + 1] - v->offsets[r]);
    }
    if (v->offsets[n] && !v->data) return fail(NEEDLE_ERR_INVALID, "data is NULL");
    const uint64_t total_bytes = v->offsets[n] * cw;
    const uint64_t padded = n * std::max<uint64_t>(16, (max_len * cw + 15) & ~(uint64_t)15);
    if (padded <= 4 * total_bytes + (64u << 10)) return run_packed_host_one(p, op, v, bitmap, start, end);
    auto klass = [&](uint64_t len_chars) { // smallest k with len * cw <= 64 << 2k
        int k = 0;
        while ((len_chars * cw) > (64ull << (2 * k))) ++k;
        return k;
    };
    const int n_classes = klass(max_len) + 1;
    std::vector<std::vector<uint64_t>> rows_of((size_t)n_classes);
    for (uint64_t r = 0; r < n; ++r) rows_of[(size_t)klass(v->offsets[r + 1] - v->offsets[r])].push_back(r);
    memset(bitmap, 0, ((n + 63) / 64) * 8);
    std::vector<uint8_t> data;
    std::vector<uint64_t> off, bm;
    std::vector<int32_t> st, en;
    for (const auto &ids : rows_of) {
        if (ids.empty()) continue;
        off.assign(ids.size() + 1, 0);
        for (size_t i = 0; i < ids.size(); ++i) off[i + 1] = off[i] + (v->offsets[ids[i] + 1] - v->offsets[ids[i]]);
        data.resize((size_t)(off.back() * cw));
        for (size_t i = 0; i < ids.size(); ++i)
            memcpy(data.data() + off[i] * cw, (const uint8_t *)v->data + v->offsets[ids[i]] * cw, (size_t)((off[i + 1] - off[i]) * cw));
        needle_packed_view sub;
        sub.data = data.data();
        sub.char_width = v->char_width;
        sub.n_rows = ids.size();
        sub.offsets = off.data();
        bm.assign((ids.size() + 63) / 64, 0);
        if (op == OP_FIND) {
            st.assign(ids.size(), -1);
            en.assign(ids.size(), -1);
        }
        rc = run_packed_host_one(p, op, &sub, bm.data(), st.data(), en.data());
        if (rc) return rc;
        for (size_t i = 0; i < ids.size(); ++i) {
            if ((bm[i >> 6] >> (i & 63)) & 1) bitmap[ids[i] >> 6] |= 1ull << (ids[i] & 63);
            if (op == OP_FIND) {
                start[ids[i]] = st[i];
                end[ids[i]] = en[i];
            }
        }
    }
    return NEEDLE_OK;
}

static int run_packed_host_one(const needle_pattern *p, int op, const needle_packed_view *v, uint64_t *bitmap,
                               int32_t *start, int32_t *end) {
    if (!p) return fail(NEEDLE_ERR_INVALID, "pattern is NULL");
    int rc = check_packed(v);
    if (rc) return rc;
    if (v->n_rows == 0) return NEEDLE_OK;
    if (!bitmap) return fail(NEEDLE_ERR_INVALID, "bitmap is NULL");
    if (op == OP_FIND && (!start || !end)) return fail(NEEDLE_ERR_INVALID, "start/end is NULL");
    const uint64_t cw = v->char_width;
    uint64_t max_len = 0;
    for (uint64_t r = 0; r < v->n_rows; ++r) {
        if (v->offsets[r + 1] < v->offsets[r]) return fail(NEEDLE_ERR_INVALID, "offsets must be non-decreasing");
        max_len = std::max<uint64_t>(max_len, v->offsets[r + 1] - v->offsets[r]);
    }
    if (max_len > 0xFFFFFFFFull) return fail(NEEDLE_ERR_INVALID, "row longer than 2^32 - 1 chars");
    const uint64_t total_chars = v->offsets[v->n_rows];
    if (total_chars && !v->data) return fail(NEEDLE_ERR_INVALID, "data is NULL");
    uint64_t stride_bytes = (max_len * cw + 15) & ~(uint64_t)15;
    if (stride_bytes == 0) stride_bytes = 16;
    const size_t words = (v->n_rows + 63) / 64;
    void *d_data = nullptr, *d_rows = nullptr;
    uint64_t *d_off = nullptr, *d_bm = nullptr;
    uint32_t *d_len = nullptr;
    int32_t *d_s = nullptr, *d_e = nullptr;
    auto cleanup = [&]() {
        for (void *q : {d_data, d_rows, (void *)d_off, (void *)d_bm, (void *)d_len, (void *)d_s, (void *)d_e})
            if (q) (void)hipFree(q);
    };
#define HIP_TRY_C(expr)                                    \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) {                            \
            cleanup();                                     \
            return hip_fail(_e, #expr);                    \
        }                                                  \
    } while (0)
    const size_t data_bytes = (size_t)((total_chars * cw + 3) & ~(uint64_t)3);
    HIP_TRY_C(hipMalloc(&d_data, data_bytes ? data_bytes : 4));
    if (total_chars) HIP_TRY_C(hipMemcpy(d_data, v->data, (size_t)(total_chars * cw), hipMemcpyHostToDevice));
    HIP_TRY_C(hipMalloc((void **)&d_off, (v->n_rows + 1) * 8));
    HIP_TRY_C(hipMemcpy(d_off, v->offsets, (v->n_rows + 1) * 8, hipMemcpyHostToDevice));
    HIP_TRY_C(hipMalloc(&d_rows, v->n_rows * stride_bytes));
    HIP_TRY_C(hipMalloc((void **)&d_len, v->n_rows * 4));
    HIP_TRY_C(hipMalloc((void **)&d_bm, words * 8));
    if (op == OP_FIND) {
        HIP_TRY_C(hipMalloc((void **)&d_s, v->n_rows * 4));
        HIP_TRY_C(hipMalloc((void **)&d_e, v->n_rows * 4));
    }
    needle_packed_view dpv = *v;
    dpv.data = d_data;
    dpv.offsets = d_off;
    rc = needle_rows_from_packed_dev(&dpv, d_rows, stride_bytes / cw, d_len, nullptr, nullptr);
    if (rc) {
        cleanup();
        return rc;
    }
    needle_batch_view bv;
    memset(&bv, 0, sizeof(bv));
    bv.rows = d_rows;
    bv.char_width = v->char_width;
    bv.n_rows = v->n_rows;
    bv.row_stride = stride_bytes / cw;
    bv.lengths = d_len;
    rc = run_dev(p, op, &bv, d_bm, d_s, d_e, nullptr);
    if (rc) {
        cleanup();
        return rc;
    }
    HIP_TRY_C(hipDeviceSynchronize());
    HIP_TRY_C(hipMemcpy(bitmap, d_bm, words * 8, hipMemcpyDeviceToHost));
    if (op == OP_FIND) {
        HIP_TRY_C(hipMemcpy(start, d_s, v->n_rows * 4, hipMemcpyDeviceToHost));
        HIP_TRY_C(hipMemcpy(end, d_e, v->n_rows * 4, hipMemcpyDeviceToHost));
    }
    cleanup();
    return NEEDLE_OK;
#undef HIP_TRY_C
}

// ------------------------------------------------------------------------------------------------
extern "C" {

int needle_rows_from_packed_dev(const needle_packed_view *v, void *d_rows, uint64_t row_stride, uint32_t *d_lengths,
                                int32_t *d_overflow, void *stream) {
    int rc = check_packed(v);
    if (rc) return rc;
    if (v->n_rows == 0) return NEEDLE_OK;
    if (!d_rows || !d_lengths) return fail(NEEDLE_ERR_INVALID, "output buffer is NULL");
    const uint64_t stride_bytes = row_stride * v->char_width;
    if (stride_bytes == 0 || stride_bytes % 16 != 0)
        return fail(NEEDLE_ERR_INVALID, "row_stride * char_width must be a non-zero multiple of 16 bytes");
    if (((uintptr_t)d_rows) % 16 != 0) return fail(NEEDLE_ERR_INVALID, "d_rows must be 16-byte aligned");
    if (((uintptr_t)v->data) % 4 != 0) return fail(NEEDLE_ERR_INVALID, "packed data must be 4-byte aligned");
    int dev = 0, cus = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    HIP_TRY(launch_unpack(v->data, v->offsets, v->n_rows, v->char_width, d_rows, stride_bytes, d_lengths, d_overflow,
                          cus, (hipStream_t)stream));
    return NEEDLE_OK;
}

int needle_matches_packed_host(const needle_pattern *p, const needle_packed_view *v, uint64_t *bm) {
    return run_packed_host(p, OP_MATCHES, v, bm, nullptr, nullptr);
}
int needle_contained_in_packed_host(const needle_pattern *p, const needle_packed_view *v, uint64_t *bm) {
    return run_packed_host(p, OP_CONTAINED_IN, v, bm, nullptr, nullptr);
}
int needle_find_packed_host(const needle_pattern *p, const needle_packed_view *v, uint64_t *bm, int32_t *st, int32_t *en) {
    return run_packed_host(p, OP_FIND, v, bm, st, en);
}

const char *needle_version(void) { return "needle_hip 0.1 (gfx950)"; }
const char *needle_last_error(void) { return g_err.c_str(); }

int needle_trim_scratch(size_t keep_bytes) {
    const hipError_t e = needle::scratch_trim(keep_bytes);
    return e == hipSuccess ? NEEDLE_OK : hip_fail(e, "hipMemPoolTrimTo");
}
int needle_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int needle_compile(const uint16_t *regex, size_t n, int flags, needle_pattern **out) {
    if (!out) return fail(NEEDLE_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!regex && n) return fail(NEEDLE_ERR_INVALID, "regex is NULL"); // Objects.requireNonNull, RegexParser.java:87
    if (flags & ~NEEDLE_ALL_FLAGS) return fail(NEEDLE_ERR_INVALID, "unknown flag bits"); // CompilerOptions.java:9-16
    needle_pattern *p = new needle_pattern();
    std::string err;
    int rc = compile_regex(std::u16string((const char16_t *)regex, n), flags, p->t, err);
    if (rc != NEEDLE_OK) {
        delete p;
        return fail(rc, err);
    }
    if (!validate_tables(p->t, err)) {
        delete p;
        return fail(NEEDLE_ERR_COMPILE, err);
    }
    *out = p;
    return NEEDLE_OK;
}

static int take_dfa(const needle_dfa_desc &d, int stride, RefDfa &o, std::string &err) {
    if (d.n_states < 1 || d.n_states > 16383) { err = "n_states out of range (1..16383)"; return NEEDLE_ERR_COMPILE; }
    if (!d.accepting) { err = "accepting is NULL"; return NEEDLE_ERR_INVALID; }
    o.n_states = d.n_states;
    o.max_char = d.max_char;
    o.accepting.resize(d.n_states);
    for (int i = 0; i < d.n_states; ++i) o.accepting[i] = d.accepting[i] ? 1 : 0;
    if (d.table) {
        o.table.assign(d.table, d.table + (size_t)d.n_states * stride);
    } else if (d.table_string) {
        if (!decode_table_string(d.table_string, d.n_states, stride, o.table, err)) return NEEDLE_ERR_INVALID;
    } else {
        err = "neither table nor table_string given";
        return NEEDLE_ERR_INVALID;
    }
    return NEEDLE_OK;
}

int needle_pattern_from_tables(const needle_table_desc *desc, needle_pattern **out) {
    if (!out) return fail(NEEDLE_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!desc || !desc->class_map) return fail(NEEDLE_ERR_INVALID, "desc / class_map is NULL");
    if (desc->stride < 1 || desc->stride > 255) return fail(NEEDLE_ERR_INVALID, "stride out of range");
    needle_pattern *p = new needle_pattern();
    p->t.class_map.assign(desc->class_map, desc->class_map + 65536);
    p->t.stride = desc->stride;
    p->t.fixed_len = desc->fixed_len < 0 ? -1 : desc->fixed_len;
    const needle_dfa_desc *ds[4] = {&desc->matches, &desc->contained_in, &desc->forwards, &desc->backwards};
    std::string err;
    for (int w = 0; w < 4; ++w) {
        int rc = take_dfa(*ds[w], desc->stride, p->t.dfa[w], err);
        if (rc) {
            delete p;
            return fail(rc, err);
        }
    }
    if (!validate_tables(p->t, err)) {
        delete p;
        return fail(NEEDLE_ERR_INVALID, err);
    }
    *out = p;
    return NEEDLE_OK;
}

void needle_pattern_destroy(needle_pattern *p) { delete p; }

int needle_pattern_get_info(const needle_pattern *p, needle_pattern_info *o) {
    if (!p || !o) return fail(NEEDLE_ERR_INVALID, "NULL argument");
    memset(o, 0, sizeof(*o));
    o->stride = p->t.stride;
    o->fixed_len = p->t.fixed_len;
    o->min_len = p->t.min_len;
    o->max_len = p->t.max_len;
    for (int w = 0; w < 4; ++w) {
        o->n_states[w] = p->t.dfa[w].n_states;
        o->max_char[w] = p->t.dfa[w].max_char;
        o->kernel_mode[w] = (int)lower(p->t, (Which)w, 1, max_prog_lds(), false).hdr.mode;
    }
    return NEEDLE_OK;
}

int needle_pattern_program_info(const needle_pattern *p, int which, int char_width, int with_backward, needle_program_info *o) {
    if (!p || !o) return fail(NEEDLE_ERR_INVALID, "NULL argument");
    if (which < 0 || which > 2 || (char_width != 1 && char_width != 2)) return fail(NEEDLE_ERR_INVALID, "which / char_width out of range");
    memset(o, 0, sizeof(*o));
    const bool backward = with_backward != 0 && which == W_FORWARDS && p->t.fixed_len < 0;
    Program pr = lower(p->t, (Which)which, char_width, max_prog_lds(), false, backward);
    if (backward && find_lengths_for(pr.hdr.mode)) { // (as run_dev chooses)
        if (const MatchLengths *ml = pattern_ml(p)) {
            Program lp = lower_match_lengths(p->t, *ml, char_width, max_prog_lds(), false);
            static const bool force_tables = getenv("NEEDLE_FIND_LENGTHS") && atoi(getenv("NEEDLE_FIND_LENGTHS")) > 1;
            const bool pair_lost = pr.hdr.mode == MODE_PAIR && lp.hdr.mode != MODE_PAIR && !force_tables;
            if (!lp.blob.empty() && !pair_lost) pr = std::move(lp), o->lengths_form = 1;
        }
    }
    o->mode = (int32_t)pr.hdr.mode;
    o->n_states = (int32_t)pr.hdr.n_states;
    o->lds_bytes = (int32_t)pr.hdr.lds_bytes;
    o->blob_bytes = (int32_t)pr.blob.size();
    int waves = 0, chb = 0, in_f = 0;
    if (shape_for_program(pr.hdr, char_width, &waves, &chb, &in_f)) o->waves = waves, o->tile_bytes = chb;
    o->dense_rows = (int32_t)pr.hdr.sp_dense;
    o->records = (int32_t)pr.hdr.sp_records;
    o->chains = (int32_t)pr.hdr.sp_chains;
    if (pr.hdr.mode == MODE_HYBRID) o->hot_rows = (int32_t)(pr.hdr.hot_bytes / (pr.hdr.n_cols * 2u));
    o->window = (int32_t)pr.hdr.win_on;
    if (pr.hdr.win_on) {
        const uint32_t e = pr.hdr.mode == MODE_SPARSE ? 4u : (pr.hdr.mode == MODE_TABLE16 || pr.hdr.mode == MODE_HYBRID) ? 2u : 1u;
        o->window_lo = (int32_t)(pr.hdr.win_lo_e / e);
        o->window_hi = (int32_t)(pr.hdr.win_hi_e / e);
    }
    return NEEDLE_OK;
}

// The n-gram candidate filter (needle_ngram_host.h) of the program containedIn() (which = 1) / find() (which = 2) runs on 8-bit
// rows, as run_dev chooses it: whether there is one, its parameters, why not, and (bitmap != NULL) the bitmap itself.
static int prefilter_info_uncached(const needle_pattern *p, int which, needle_prefilter_info *o, std::vector<uint32_t> *bitmap_out, bool wide,
                                   uint32_t *m1b, uint32_t *m2b);

// wide: the filter of UTF-16 rows of a pattern on several pages of the BMP (lower_filter_wide).  At most cap_words bitmap words are written.
int needle_pattern_prefilter_info2(const needle_pattern *cp, int which, int wide, needle_prefilter_info2 *o, uint32_t *bitmap, size_t cap_words) {
    needle_pattern *p = const_cast<needle_pattern *>(cp);
    if (!p || !o) return fail(NEEDLE_ERR_INVALID, "NULL argument");
    if (which != W_CONTAINED_IN && which != W_FORWARDS) return fail(NEEDLE_ERR_INVALID, "which must be 1 (contained_in) or 2 (forwards)");
    std::lock_guard<std::mutex> lk(p->pf_mu);
    needle_pattern::PrefilterCache &c = p->pf_cache[(wide ? 4 : 0) + which];
    if (!c.have) {
        const int rc = prefilter_info_uncached(p, which, &c.info, &c.bitmap, wide != 0, &c.m1b, &c.m2b);
        if (rc) return rc;
        c.have = true;
    }
    memset(o, 0, sizeof(*o));
    o->base = c.info;
    o->wide = wide ? 1 : 0;
    o->m1b = c.m1b, o->m2b = c.m2b;
    if (bitmap && c.info.on) memcpy(bitmap, c.bitmap.data(), std::min(cap_words, c.bitmap.size()) * 4);
    return NEEDLE_OK;
}

// (the original contract: the FIRST level's bitmap_bytes / 4 words only -- the second level's come through needle_pattern_prefilter_info2,
// which takes the caller's capacity)
int needle_pattern_prefilter_info(const needle_pattern *cp, int which, needle_prefilter_info *o, uint32_t *bitmap) {
    if (!o) return fail(NEEDLE_ERR_INVALID, "NULL argument");
    needle_prefilter_info2 i2;
    int rc = needle_pattern_prefilter_info2(cp, which, 0, &i2, nullptr, 0);
    if (rc) return rc;
    *o = i2.base;
    if (bitmap && o->on) rc = needle_pattern_prefilter_info2(cp, which, 0, &i2, bitmap, (size_t)o->bitmap_bytes / 4);
    return rc;
}

int needle_pattern_set_prefilter(needle_pattern *p, int mode) {
    if (!p) return fail(NEEDLE_ERR_INVALID, "pattern is NULL");
    if (mode < 0 || mode > 2) return fail(NEEDLE_ERR_INVALID, "mode must be NEEDLE_PREFILTER_AUTO (0), _ON (1) or _OFF (2)");
    p->pf_mode.store(mode);
    return NEEDLE_OK;
}

int needle_pattern_utf16_route(const needle_pattern *p, int32_t *page, int32_t *sub) {
    if (!p || !page || !sub) return fail(NEEDLE_ERR_INVALID, "NULL argument");
    const Utf16Route r = utf16_route(p);
    *page = r.page, *sub = r.page >= 0 ? r.sub : 0;
    return NEEDLE_OK;
}

// What the flood watch of this pattern's filter program(s) for `which` on the CURRENT device knows (no device needed to ask; all zero
// before the first scan).  Several programs may carry a filter (plain / lengths / HBM-table forms): the one that ran last is reported.
int needle_pattern_prefilter_state(const needle_pattern *cp, int which, needle_prefilter_state *o) {
    needle_pattern *p = const_cast<needle_pattern *>(cp);
    if (!p || !o) return fail(NEEDLE_ERR_INVALID, "NULL argument");
    if (which != W_CONTAINED_IN && which != W_FORWARDS) return fail(NEEDLE_ERR_INVALID, "which must be 1 (contained_in) or 2 (forwards)");
    memset(o, 0, sizeof(*o));
    o->mode = p->pf_mode.load();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return NEEDLE_OK; // (no device: nothing has run)
    std::lock_guard<std::mutex> lk(p->mu);
    uint64_t best = 0;
    for (auto &kv : p->cache) {
        if (std::get<0>(kv.first) != dev || std::get<1>(kv.first) != which || !kv.second.d_ng) continue;
        const DevProgram &dp = kv.second;
        std::lock_guard<std::mutex> lk2(dp.ng_mu);
        o->has_filter = 1;
        const uint64_t used = dp.ng_launches + dp.ng_suspended_calls;
        if (used < best) continue;
        best = used;
        o->suspended_calls_left = dp.ng_suspend;
        o->backoff = dp.ng_backoff;
        o->last_candidates_per_kib = dp.ng_last_rate;
        o->filter_launches = dp.ng_launches;
        o->suspended_calls = dp.ng_suspended_calls;
    }
    return NEEDLE_OK;
}

static int prefilter_info_uncached(const needle_pattern *p, int which, needle_prefilter_info *o, std::vector<uint32_t> *bitmap_out, bool wide,
                                   uint32_t *m1b, uint32_t *m2b) {
    memset(o, 0, sizeof(*o));
    *m1b = *m2b = 0;
    const bool backward = which == W_FORWARDS && p->t.fixed_len < 0;
    Program pr;
    bool usable = which == W_CONTAINED_IN || p->t.fixed_len >= 0;
    if (wide) { // (as run_dev / find_all_one_pass build it; whether a batch takes it also depends on the ordinary UTF-16 program's mode)
        const MatchLengths *ml = backward ? pattern_ml(p) : nullptr;
        if (p->t.class_map.size() == 65536 && (!backward || ml)) pr = lower_filter_wide(p->t, (Which)which, ml), usable = true;
        else memset(&pr.hdr, 0, sizeof(pr.hdr)), memset(&pr.ng.p, 0, sizeof(pr.ng.p)), pr.hdr.mode = MODE_GLOBAL;
    } else
    pr = lower(p->t, (Which)which, 1, max_prog_lds(), false, backward);
    if (!wide && backward && find_lengths_for(pr.hdr.mode)) {
        if (const MatchLengths *ml = pattern_ml(p)) {
            Program lp = lower_match_lengths(p->t, *ml, 1, max_prog_lds(), false);
            static const bool force_tables = getenv("NEEDLE_FIND_LENGTHS") && atoi(getenv("NEEDLE_FIND_LENGTHS")) > 1;
            const bool pair_lost = pr.hdr.mode == MODE_PAIR && lp.hdr.mode != MODE_PAIR && !force_tables;
            if (!lp.blob.empty() && !pair_lost) pr = std::move(lp), usable = true;
        }
    }
    if (!wide && (pr.hdr.mode == MODE_HYBRID || pr.hdr.mode == MODE_GLOBAL) && ngram_level() > 0) {
        // an automaton that fits the LDS in no form: the filter program walks its table out of HBM / L2 (lower_filter_hbm)
        const MatchLengths *ml = backward ? pattern_ml(p) : nullptr;
        static const bool unbounded_hbm = !(getenv("NEEDLE_PREFILTER_UNBOUNDED") && atoi(getenv("NEEDLE_PREFILTER_UNBOUNDED")) == 0);
        if (!backward || ml) {
            Program hp = lower_filter_hbm(p->t, (Which)which, ml);
            if (!hp.blob.empty() && hp.hdr.mode == MODE_GLOBAL) pr = std::move(hp), usable = true;
        } else if (unbounded_hbm) { // no bounded match lengths: the forward search automaton + backward walks (get_program variant 12)
            Program hp = lower_filter_hbm(p->t, (Which)which, nullptr, true);
            if (!hp.blob.empty() && hp.hdr.mode == MODE_GLOBAL) pr = std::move(hp), usable = true;
        }
    }
    // (find() of a pattern without bounded match lengths: behind the filter of its ordinary LDS-resident program, starts by backward walks)
    static const bool unbounded_on = !(getenv("NEEDLE_PREFILTER_UNBOUNDED") && atoi(getenv("NEEDLE_PREFILTER_UNBOUNDED")) == 0);
    if (!wide && !usable && backward && unbounded_on && pr.hdr.mode != MODE_GLOBAL && pr.hdr.mode != MODE_HYBRID) usable = true;
    const NgramFilter &f = pr.ng;
    o->mode = (int32_t)pr.hdr.mode;
    if (!usable) {
        snprintf(o->why, sizeof(o->why), "find() needs its backward walk for this pattern");
        return NEEDLE_OK;
    }
    o->on = (int32_t)(f.p.on && ngram_lds_bytes(pr.hdr, f.p) ? 1 : 0);
    o->stride = (int32_t)f.p.stride;
    o->warm = (int32_t)f.p.warm;
    o->min_len = (int32_t)f.p.min_len;
    o->n_windows = (int32_t)f.p.n_grams;
    o->bitmap_bytes = (int32_t)f.p.bm_bytes;
    o->m1 = f.p.m1, o->m2 = f.p.m2, o->addr_shift = f.p.addr_shift, o->addr_mask = f.p.addr_mask;
    *m1b = f.p.m1b, *m2b = f.p.m2b;
    o->on2 = (int32_t)(o->on ? f.p.on2 : 0); // (2: two-sided, NgramParams::on2)
    if (o->on2) o->n_windows2 = (int32_t)f.p.n_grams2, o->bitmap2_bytes = (int32_t)f.p.bm2_bytes, o->m3 = f.p.m3, o->addr_mask2 = f.p.addr_mask2;
    snprintf(o->why, sizeof(o->why), "%s", f.p.on ? "" : (f.why.empty() ? (ngram_level() > 0 ? "not a mode the filter is built for" : "NEEDLE_PREFILTER=0") : f.why.c_str()));
    if (f.p.on) {
        *bitmap_out = f.bitmap;
        if (o->on2) bitmap_out->insert(bitmap_out->end(), f.bitmap2.begin(), f.bitmap2.end()); // (the second level's right behind)
    }
    return NEEDLE_OK;
}

// The find-all "lengths" automaton (needle_lower.h), for inspection and CPU-side tests: 1 in *available when the pattern
// allows it.  table: n_states * (stride + 1) int16 (reference layout + one column for chars beyond *max_char, -1 = dead);
// accepting, pend: n_states bytes each.
int needle_pattern_match_lengths(const needle_pattern *cp, int32_t *available, int32_t *n_states, int32_t *n_dead, int32_t *max_char,
                                 int16_t *table, uint8_t *accepting, uint8_t *pend) {
    needle_pattern *p = const_cast<needle_pattern *>(cp);
    if (!p || !available) return fail(NEEDLE_ERR_INVALID, "NULL argument");
    *available = pattern_ml(p) ? 1 : 0;
    if (!*available) return NEEDLE_OK;
    if (n_states) *n_states = p->ml.dfa.n_states;
    if (n_dead) *n_dead = p->ml.n_dead;
    if (max_char) *max_char = p->ml.dfa.max_char;
    if (table) {
        const int N = p->t.stride;
        for (int s = 0; s < p->ml.dfa.n_states; ++s) {
            memcpy(table + (size_t)s * (N + 1), &p->ml.dfa.table[(size_t)s * N], (size_t)N * 2);
            table[(size_t)s * (N + 1) + N] = p->ml.over[s];
        }
    }
    if (accepting) memcpy(accepting, p->ml.dfa.accepting.data(), p->ml.dfa.accepting.size());
    if (pend) memcpy(pend, p->ml.pend.data(), p->ml.pend.size());
    return NEEDLE_OK;
}

// The find-all transducer's device program (needle_lower.h), for inspection and CPU-side tests that walk the blob the way the kernel
// does.  info[0..11] = n_states, n_cols, pad_col, start, win_on, win_lo_e, win_hi_e, off_table, ft_codes_off, lds_bytes, n_pages, 0.
int needle_pattern_find_all_transducer(const needle_pattern *cp, int char_width, int32_t *available, int32_t *info, void *blob, size_t cap,
                                       size_t *needed) {
    needle_pattern *p = const_cast<needle_pattern *>(cp);
    if (!p || !available || (char_width != 1 && char_width != 2)) return fail(NEEDLE_ERR_INVALID, "bad argument");
    *available = 0;
    const MatchLengths *ml = pattern_ml(p);
    Program pr;
    memset(&pr.hdr, 0, sizeof(pr.hdr));
    if (ml) pr = lower_find_all_transducer(p->t, *ml, char_width, max_prog_lds());
    // (no transducer on the lengths automaton: the RUN transducer, if the pattern is one of runs -- info[11] = 2)
    if ((pr.blob.empty() || !pr.hdr.ft_on) && p->t.fixed_len < 0) pr = lower_find_all_runs(p->t, char_width, max_prog_lds());
    if (pr.blob.empty() || !pr.hdr.ft_on) return NEEDLE_OK;
    *available = 1;
    if (info) {
        const ProgHeader &h = pr.hdr;
        const uint32_t v[12] = {h.n_states, h.n_cols, h.pad_col, h.start, h.win_on, h.win_lo_e, h.win_hi_e, h.off_table, h.ft_codes_off, h.lds_bytes, h.n_pages, h.ft_on};
        for (int i = 0; i < 12; ++i) info[i] = (int32_t)v[i];
    }
    if (needed) *needed = pr.blob.size();
    if (blob && cap >= pr.blob.size()) memcpy(blob, pr.blob.data(), pr.blob.size());
    return NEEDLE_OK;
}

int needle_pattern_get_class_map(const needle_pattern *p, uint8_t *cm) {
    if (!p || !cm) return fail(NEEDLE_ERR_INVALID, "NULL argument");
    memcpy(cm, p->t.class_map.data(), 65536);
    return NEEDLE_OK;
}

int needle_pattern_get_table(const needle_pattern *p, int which, int16_t *table, uint8_t *accepting) {
    if (!p || which < 0 || which > 3) return fail(NEEDLE_ERR_INVALID, "bad argument");
    const RefDfa &d = p->t.dfa[which];
    if (table) memcpy(table, d.table.data(), d.table.size() * 2);
    if (accepting) memcpy(accepting, d.accepting.data(), d.accepting.size());
    return NEEDLE_OK;
}

// ---- precompiled-pattern blob ("NDLT" v1)
static void put_i32(std::vector<uint8_t> &b, int32_t v) {
    for (int i = 0; i < 4; ++i) b.push_back((uint8_t)((uint32_t)v >> (8 * i)));
}
static bool get_i32(const uint8_t *&p, const uint8_t *end, int32_t &v) {
    if (end - p < 4) return false;
    v = (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
    p += 4;
    return true;
}

int needle_pattern_serialize(const needle_pattern *p, void *buf, size_t cap, size_t *needed) {
    if (!p || !needed) return fail(NEEDLE_ERR_INVALID, "NULL argument");
    std::vector<uint8_t> b;
    const char magic[4] = {'N', 'D', 'L', 'T'};
    b.insert(b.end(), magic, magic + 4);
    put_i32(b, 1);
    put_i32(b, p->t.stride);
    put_i32(b, p->t.fixed_len);
    put_i32(b, p->t.min_len);
    put_i32(b, p->t.max_len);
    for (int w = 0; w < 4; ++w) {
        put_i32(b, p->t.dfa[w].n_states);
        put_i32(b, p->t.dfa[w].max_char);
    }
    b.insert(b.end(), p->t.class_map.begin(), p->t.class_map.end());
    for (int w = 0; w < 4; ++w) {
        for (int16_t v : p->t.dfa[w].table) {
            b.push_back((uint8_t)((uint16_t)v & 255));
            b.push_back((uint8_t)((uint16_t)v >> 8));
        }
        b.insert(b.end(), p->t.dfa[w].accepting.begin(), p->t.dfa[w].accepting.end());
    }
    *needed = b.size();
    if (cap == 0) return NEEDLE_OK;
    if (!buf || cap < b.size()) return fail(NEEDLE_ERR_INVALID, "buffer too small");
    memcpy(buf, b.data(), b.size());
    return NEEDLE_OK;
}

int needle_pattern_deserialize(const void *buf, size_t n, needle_pattern **out) {
    if (!out) return fail(NEEDLE_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!buf) return fail(NEEDLE_ERR_INVALID, "buf is NULL");
    const uint8_t *q = (const uint8_t *)buf, *end = q + n;
    if (n < 8 || memcmp(q, "NDLT", 4) != 0) return fail(NEEDLE_ERR_INVALID, "not a needle table blob");
    q += 4;
    int32_t ver = 0;
    needle_pattern *p = new needle_pattern();
    bool ok = get_i32(q, end, ver) && ver == 1 && get_i32(q, end, p->t.stride) && get_i32(q, end, p->t.fixed_len) &&
              get_i32(q, end, p->t.min_len) && get_i32(q, end, p->t.max_len);
    for (int w = 0; ok && w < 4; ++w) ok = get_i32(q, end, p->t.dfa[w].n_states) && get_i32(q, end, p->t.dfa[w].max_char);
    ok = ok && p->t.stride >= 1 && p->t.stride <= 255 && (size_t)(end - q) >= 65536;
    if (ok) {
        p->t.class_map.assign(q, q + 65536);
        q += 65536;
    }
    for (int w = 0; ok && w < 4; ++w) {
        RefDfa &d = p->t.dfa[w];
        ok = d.n_states >= 1 && d.n_states <= 16383;
        const size_t cells = ok ? (size_t)d.n_states * p->t.stride : 0;
        ok = ok && (size_t)(end - q) >= cells * 2 + (size_t)d.n_states;
        if (!ok) break;
        d.table.resize(cells);
        for (size_t i = 0; i < cells; ++i) d.table[i] = (int16_t)((uint16_t)q[2 * i] | ((uint16_t)q[2 * i + 1] << 8));
        q += cells * 2;
        d.accepting.assign(q, q + d.n_states);
        q += d.n_states;
    }
    std::string err = "truncated or malformed table blob";
    if (!ok || q != end || !validate_tables(p->t, err)) {
        delete p;
        return fail(NEEDLE_ERR_INVALID, err);
    }
    *out = p;
    return NEEDLE_OK;
}

int needle_matches_dev(const needle_pattern *p, const needle_batch_view *v, uint64_t *bm, void *s) {
    return run_dev(p, OP_MATCHES, v, bm, nullptr, nullptr, s);
}
int needle_contained_in_dev(const needle_pattern *p, const needle_batch_view *v, uint64_t *bm, void *s) {
    return run_dev(p, OP_CONTAINED_IN, v, bm, nullptr, nullptr, s);
}
int needle_find_dev(const needle_pattern *p, const needle_batch_view *v, uint64_t *bm, int32_t *st, int32_t *en, void *s) {
    return run_dev(p, OP_FIND, v, bm, st, en, s);
}
// find() with a row's start / end as one dword (start | end << 16, 0xFFFFFFFF = no match), stored by the scan kernel itself:
// 4 result bytes per row instead of 8, no separate pack pass.  Rows of at most 65 534 chars.
int needle_find_packed16_dev(const needle_pattern *p, const needle_batch_view *v, uint64_t *bm, uint32_t *start_end16, void *s) {
    if (v && v->n_rows && !start_end16) return fail(NEEDLE_ERR_INVALID, "start_end16 is NULL");
    if (v && !offsets16_ok(v, 65534u)) return fail(NEEDLE_ERR_UNSUPPORTED, "16-bit offsets: rows of at most 65 534 chars (use needle_find_dev)");
    return run_dev(p, OP_FIND, v, bm, nullptr, nullptr, s, nullptr, nullptr, false, start_end16);
}
// find() with a row's result as ONE uint16 -- start | (end - start) << 8, 0xFFFF = no match, 0xFFFE = the match (0, 256) -- stored by the
// kernels themselves: 2 result bytes per row.  Rows of at most 256 chars.
int needle_find_packed8_dev(const needle_pattern *p, const needle_batch_view *v, uint64_t *bm, uint16_t *start_len8, void *s) {
    if (v && v->n_rows && !start_len8) return fail(NEEDLE_ERR_INVALID, "start_len8 is NULL");
    if (v && (v->lengths ? v->row_stride : v->row_len) > 256u) return fail(NEEDLE_ERR_UNSUPPORTED, "8-bit start / length: rows of at most 256 chars (use needle_find_packed16_dev)");
    return run_dev(p, OP_FIND, v, bm, nullptr, nullptr, s, nullptr, nullptr, false, (uint32_t *)start_len8, true);
}
int needle_find_next_dev(const needle_pattern *p, const needle_batch_view *v, const int32_t *cur, uint64_t *bm, int32_t *st,
                         int32_t *en, void *s) {
    if (!cur) return fail(NEEDLE_ERR_INVALID, "cursor is NULL");
    return run_dev(p, OP_FIND, v, bm, st, en, s, cur);
}
// The round-per-match form: one needle_find_next pass over the batch per round, one stream synchronisation per round.
// Rows of 64 MiB and more (stripe paths only) take it; NEEDLE_FIND_ALL_ROUNDS=1 forces it (tests cross-check the two).
static int find_all_rounds(const needle_pattern *p, const needle_batch_view *v, uint32_t slots, uint32_t *d_counts, int32_t *d_start,
                           int32_t *d_end, int *more, hipStream_t stream) {
    const size_t n = (size_t)v->n_rows, words = (n + 63) / 64;
    int dev = 0, cus = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    uint8_t *tmp = nullptr; // cursor | start | end (int32 each) | bitmap | any-hit flag
    const size_t o_cur = 0, o_s = n * 4, o_e = 2 * n * 4, o_bm = (3 * n * 4 + 15) & ~(size_t)15, o_flag = o_bm + words * 8;
    HIP_TRY(scratch_malloc((void **)&tmp, o_flag + 16, stream));
    auto done = [&](int code) {
        (void)scratch_free(tmp, stream);
        return code;
    };
    if (hipMemsetAsync(tmp + o_cur, 0, n * 4, stream) != hipSuccess) return done(fail(NEEDLE_ERR_DEVICE, "hipMemsetAsync"));
    for (uint32_t k = 0;; ++k) {
        if (hipMemsetAsync(tmp + o_flag, 0, 4, stream) != hipSuccess) return done(fail(NEEDLE_ERR_DEVICE, "hipMemsetAsync"));
        int rc = run_dev(p, OP_FIND, v, (uint64_t *)(tmp + o_bm), (int32_t *)(tmp + o_s), (int32_t *)(tmp + o_e), stream,
                         (const int32_t *)(tmp + o_cur));
        if (rc) return done(rc);
        hipError_t e = launch_find_all_collect(n, slots, k, (const int32_t *)(tmp + o_s), (const int32_t *)(tmp + o_e),
                                               (int32_t *)(tmp + o_cur), d_counts, d_start, d_end, (int32_t *)(tmp + o_flag), cus, stream);
        if (e != hipSuccess) return done(hip_fail(e, "find_all_collect"));
        int32_t any = 0;
        e = hipMemcpyAsync(&any, tmp + o_flag, 4, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return done(hip_fail(e, "find_all round"));
        if (!any) break;          // round k found nothing anywhere: every row is exhausted
        if (k >= slots) {         // a match beyond the last slot exists
            if (more) *more = 1;
            break;
        }
    }
    return done(NEEDLE_OK);
}

// One pass over the batch: every row is fetched once, each lane restarts its search where its last match ended
// (needle_find_all.hip).  Dense slots (offsets == nullptr), compact filing at caller-computed offsets, or counting only.
static int find_all_one_pass(needle_pattern *p, const needle_batch_view *v, uint32_t slots, uint32_t *d_counts, int32_t *d_start,
                             int32_t *d_end, const uint64_t *d_offsets, bool count_only, int *more, hipStream_t stream,
                             uint32_t *d_packed = nullptr, uint32_t kshift = 0) {
    const uint64_t stride_bytes = v->row_stride * v->char_width;
    if (stride_bytes >= (1ull << 26)) return fail(NEEDLE_ERR_UNSUPPORTED, "rows of 64 MiB or more: only needle_find_all_dev (round per match) takes them");
    const DevProgram *fp = nullptr, *bp = nullptr;
    int n_cus = 0;
    bool need_backward = p->t.fixed_len < 0;
    int rc = NEEDLE_OK;
    // start = end - (the match length the automaton's end state remembers): no backward walks at all, when the pattern
    // allows it (needle_lower.h: keyword unions and the like).  NEEDLE_FIND_ALL_LENGTHS=0: off (A/B, tests).
    static const bool lengths_on = !(getenv("NEEDLE_FIND_ALL_LENGTHS") && atoi(getenv("NEEDLE_FIND_ALL_LENGTHS")) == 0);
    bool lmode = false;
    if (need_backward && lengths_on && !count_only) {
        rc = get_program(p, W_FORWARDS, (int)v->char_width, 6, &fp, &n_cus);
        if (rc) return rc;
        lmode = fp != nullptr;
        static const bool sparse_lengths = getenv("NEEDLE_FIND_ALL_LENGTHS") && atoi(getenv("NEEDLE_FIND_ALL_LENGTHS")) > 1;
        if (!lmode && sparse_lengths && find_lengths_for(MODE_SPARSE)) {
            // no plain LDS table holds the lengths automaton (a big dictionary): the scan kernels' compressed form of it, if there is
            // one.  Opt-in (NEEDLE_FIND_ALL_LENGTHS=2): measured on C3-sparse (profiles/r04_find_all.md) it is no faster than hot rows +
            // backward walks, 2.05 against 1.98 ms -- the per-lane piece walk is what costs there, not the 0.25 starts per row
            const DevProgram *sp = nullptr;
            rc = get_program(p, W_FORWARDS, (int)v->char_width, 7, &sp, &n_cus);
            if (rc) return rc;
            if (sp && sp->prog.hdr.mode == MODE_SPARSE) fp = sp, lmode = true;
        }
        if (lmode) need_backward = false;
    }
    // Dictionaries whose find() runs behind the n-gram candidate filter (needle_ngram.hip): their find-all does too -- the filter
    // kernel's find-all form files every verified candidate and each row sorts its own out against its moving cursor (dense slots,
    // the counting pass and the compact filing alike).  NEEDLE_FIND_ALL_FILTER=0: off (A/B, tests).
    static const bool fa_filter = !(getenv("NEEDLE_FIND_ALL_FILTER") && atoi(getenv("NEEDLE_FIND_ALL_FILTER")) == 0);
    // (UTF-16 rows of a pattern on one page of the BMP: that page's byte programs, the text narrowed as it is loaded -- utf16_route)
    const Utf16Route u16 = v->char_width == 2 ? utf16_route(p) : Utf16Route();
    if (fa_filter && (count_only || d_offsets || slots) && ngram_level() > 0 && (p->t.fixed_len >= 0 || find_lengths_for(MODE_SPARSE))) {
        const DevProgram *sp = nullptr;
        int cus = 0;
        if (v->char_width == 2 && u16.page < 0) { // several pages of the BMP: the WIDE filter, where find() would take it (run_dev)
            const DevProgram *op16 = nullptr;
            rc = get_program(p, W_FORWARDS, 2, p->t.fixed_len < 0 ? 2 : 0, &op16, &cus);
            if (rc) return rc;
            if (op16 && wide_filter_wanted(op16->prog.hdr.mode)) {
                rc = get_program(p, W_FORWARDS, 2, 10, &sp, &cus);
                if (rc) return rc;
            }
        } else {
        const int cw8 = 1 | ((v->char_width == 2 ? u16.page : 0) << 8);
        rc = get_program(p, W_FORWARDS, cw8, p->t.fixed_len >= 0 ? 0 : 7, &sp, &cus);
        if (rc) return rc;
        if (!(sp && sp->d_ng && sp->prog.ng.p.on)) { // an automaton that fits the LDS in no form: the filter with its walks out of HBM / L2
            rc = get_program(p, W_FORWARDS, cw8, 9, &sp, &cus);
            if (rc) return rc;
        }
        }
        if (sp && sp->d_ng && sp->prog.ng.p.on && ngram_find_all_lds_bytes(sp->prog.hdr, sp->prog.ng.p)) {
            // (UTF-16 rows: the stride in CHARS -- launch_ngram_find_all with char_width 2)
            const ScanArgs a = filter_scan_args(v, v->row_stride, sp, p->t.fixed_len, nullptr, nullptr, nullptr, nullptr);
            if (ngram_shape_ok(a) && ngram_watch_allows(p, sp)) {
                int32_t *d_more = nullptr;
                HIP_TRY(scratch_malloc((void **)&d_more, 16, stream));
                hipError_t e = hipMemsetAsync(d_more, 0, 4, stream);
                if (e == hipSuccess) e = launch_ngram_find_all(a, sp->prog.ng.p, sp->d_ng, sp->d_ng_stats, slots, d_counts, d_start, d_end, d_packed, d_more, d_offsets, count_only, cus, stream, (int)v->char_width, u16.page > 0 ? u16.page : 0, v->char_width == 2 ? u16.sub : 0xFF, kshift);
                if (e == hipSuccess) e = ngram_watch_after_launch(sp, stream);
                int32_t m = 0;
                if (e == hipSuccess && more) {
                    e = hipMemcpyAsync(&m, d_more, 4, hipMemcpyDeviceToHost, stream);
                    if (e == hipSuccess) e = hipStreamSynchronize(stream);
                }
                (void)scratch_free(d_more, stream);
                if (e != hipSuccess) return hip_fail(e, "find_all (filter kernel)");
                if (more) *more = m != 0;
                return NEEDLE_OK;
            }
        }
    }
    // Patterns with a find-all transducer (needle_lower.h: bounded match lengths, no match inside a longer live one) are walked in
    // LOCK-STEP: one table lookup per char, every lane at the same char, the restarts folded into the automaton (needle_find_all_ls.hip).
    // NEEDLE_FIND_ALL_LOCKSTEP=0: off (A/B, tests: the per-lane one-pass kernel below).
    static const bool lockstep_on = !(getenv("NEEDLE_FIND_ALL_LOCKSTEP") && atoi(getenv("NEEDLE_FIND_ALL_LOCKSTEP")) == 0);
    FindAllArgs shape; // (what the launcher's own check looks at: one definition of "the lock-step kernel takes this shape")
    memset(&shape, 0, sizeof(shape));
    shape.slots = slots, shape.s.stride_bytes = stride_bytes;
    if (lockstep_on && find_all_lockstep_shape_ok(shape)) {
        const DevProgram *tp = nullptr;
        int cus = 0;
        if (lengths_on) {
            rc = get_program(p, W_FORWARDS, (int)v->char_width, 8, &tp, &cus);
            if (rc) return rc;
        }
        // ... or, without bounded match lengths, the RUN transducer (`[0-9]+`, `[a-z]{3}[a-z]*`: starts from a per-lane run-start register)
        static const bool runs_on = !(getenv("NEEDLE_FIND_ALL_RUNS") && atoi(getenv("NEEDLE_FIND_ALL_RUNS")) == 0);
        if (!tp && runs_on && p->t.fixed_len < 0) {
            rc = get_program(p, W_FORWARDS, (int)v->char_width, 11, &tp, &cus);
            if (rc) return rc;
        }
        if (tp) {
            FindAllArgs fl;
            memset(&fl, 0, sizeof(fl));
            fl.s.rows = (const uint8_t *)v->rows;
            fl.s.n_rows = v->n_rows;
            fl.s.stride_bytes = stride_bytes;
            fl.s.total_bytes = fl.s.n_rows * fl.s.stride_bytes;
            fl.s.row_len = v->row_len;
            fl.s.lengths = v->lengths;
            fl.s.prog = tp->d_blob;
            fl.s.hdr = tp->prog.hdr;
            fl.s.fixed_len = -1;
            fl.slots = slots;
            fl.kshift = kshift;
            fl.offsets = d_offsets;
            fl.count_only = count_only ? 1u : 0u;
            fl.counts = d_counts;
            fl.starts = d_start;
            fl.ends = d_end;
            fl.packed = d_packed;
            int32_t *d_more = nullptr;
            HIP_TRY(scratch_malloc((void **)&d_more, 16, stream));
            hipError_t e = hipMemsetAsync(d_more, 0, 4, stream);
            fl.more = d_more;
            if (e == hipSuccess) e = launch_find_all_lockstep((int)v->char_width, fl, cus, stream);
            int32_t m = 0;
            if (e == hipSuccess && more) { // the only synchronisation: the caller asked whether its slots sufficed
                e = hipMemcpyAsync(&m, d_more, 4, hipMemcpyDeviceToHost, stream);
                if (e == hipSuccess) e = hipStreamSynchronize(stream);
            }
            (void)scratch_free(d_more, stream);
            if (e != hipSuccess) return hip_fail(e, "find_all (lock-step kernel)");
            if (more) *more = m != 0;
            return NEEDLE_OK;
        }
    }
    if (!lmode) rc = get_program(p, W_FORWARDS, (int)v->char_width, need_backward ? 5 : 4, &fp, &n_cus);
    if (rc) return rc;
    FindAllArgs fa;
    memset(&fa, 0, sizeof(fa));
    ScanArgs &a = fa.s;
    a.rows = (const uint8_t *)v->rows;
    a.n_rows = v->n_rows;
    a.stride_bytes = stride_bytes;
    a.total_bytes = a.n_rows * a.stride_bytes;
    a.row_len = v->row_len;
    a.lengths = v->lengths;
    a.prog = fp->d_blob;
    a.hdr = fp->prog.hdr;
    a.fixed_len = p->t.fixed_len;
    fa.lmode = lmode ? 1u : 0u;
    if (need_backward) {
        rc = get_program(p, W_BACKWARDS, (int)v->char_width, 1, &bp, nullptr);
        if (rc) return rc;
        a.bprog = bp->d_blob;
        a.bhdr = bp->prog.hdr;
    }
    fa.slots = slots;
    fa.kshift = kshift;
    fa.offsets = d_offsets;
    fa.count_only = count_only ? 1u : 0u;
    static const bool no_defer = getenv("NEEDLE_FIND_ALL_DEFER") && atoi(getenv("NEEDLE_FIND_ALL_DEFER")) == 0; // A/B, tests
    fa.defer = (a.fixed_len < 0 && !a.hdr.root_accepting && !no_defer && !lmode) ? 1u : 0u;
#ifdef NEEDLE_TUNING // measurement builds only: start = the search cursor (wrong answers; never in the shipping library)
    static const bool dbg_no_backward = getenv("NEEDLE_DEBUG_NO_BACKWARD") != nullptr;
    if (fa.defer && dbg_no_backward) fa.defer = 2;
#endif
    fa.counts = d_counts;
    fa.starts = d_start;
    fa.ends = d_end;
    fa.packed = d_packed;
    int32_t *d_more = nullptr;
    HIP_TRY(scratch_malloc((void **)&d_more, 16, stream));
    auto done = [&](int code) {
        (void)scratch_free(d_more, stream);
        return code;
    };
    if (hipMemsetAsync(d_more, 0, 4, stream) != hipSuccess) return done(fail(NEEDLE_ERR_DEVICE, "hipMemsetAsync"));
    fa.more = d_more;
    hipError_t e = launch_find_all((int)v->char_width, fa, n_cus, stream);
    if (e != hipSuccess) return done(hip_fail(e, "find_all"));
    if (more) { // the only synchronisation: the caller asked whether its slots sufficed
        int32_t m = 0;
        e = hipMemcpyAsync(&m, d_more, 4, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return done(hip_fail(e, "find_all"));
        *more = m != 0;
    }
    return done(NEEDLE_OK);
}

int needle_find_all_dev(const needle_pattern *cp, const needle_batch_view *v, uint32_t slots, uint32_t *d_counts, int32_t *d_start,
                        int32_t *d_end, int *more, void *stream_) {
    needle_pattern *p = const_cast<needle_pattern *>(cp);
    if (!p) return fail(NEEDLE_ERR_INVALID, "pattern is NULL");
    int rc = check_view(v, true);
    if (rc) return rc;
    if (more) *more = 0;
    if (v->n_rows == 0) return NEEDLE_OK;
    if (!d_counts || (slots && (!d_start || !d_end))) return fail(NEEDLE_ERR_INVALID, "output buffer is NULL");
    hipStream_t stream = (hipStream_t)stream_;
    static const bool rounds = getenv("NEEDLE_FIND_ALL_ROUNDS") && atoi(getenv("NEEDLE_FIND_ALL_ROUNDS")) != 0;
    if (rounds || v->row_stride * v->char_width >= (1ull << 26)) return find_all_rounds(p, v, slots, d_counts, d_start, d_end, more, stream);
    return find_all_one_pass(p, v, slots, d_counts, d_start, d_end, nullptr, false, more, stream);
}

// needle_find_all_dev with each match as ONE dword (start | end << 16): half the result bytes, one store per match.
int needle_find_all_packed16_dev(const needle_pattern *cp, const needle_batch_view *v, uint32_t slots, uint32_t *d_counts,
                                 uint32_t *d_start_end16, int *more, void *stream_) {
    needle_pattern *p = const_cast<needle_pattern *>(cp);
    if (!p) return fail(NEEDLE_ERR_INVALID, "pattern is NULL");
    int rc = check_view(v, true);
    if (rc) return rc;
    if (more) *more = 0;
    if (v->n_rows == 0) return NEEDLE_OK;
    if (!d_counts || (slots && !d_start_end16)) return fail(NEEDLE_ERR_INVALID, "output buffer is NULL");
    if (!offsets16_ok(v, 65535u)) return fail(NEEDLE_ERR_UNSUPPORTED, "16-bit start / end: rows of at most 65535 chars");
    return find_all_one_pass(p, v, slots, d_counts, nullptr, nullptr, nullptr, false, more, (hipStream_t)stream_, d_start_end16);
}

// needle_find_all_packed16_dev with GROUP-BLOCKED slots: match k of row r at d_blocks[((r >> 6) * slots + k) * 64 + (r & 63)].
int needle_find_all_blocked16_dev(const needle_pattern *cp, const needle_batch_view *v, uint32_t slots, uint32_t *d_counts,
                                  uint32_t *d_blocks, int *more, void *stream_) {
    needle_pattern *p = const_cast<needle_pattern *>(cp);
    if (!p) return fail(NEEDLE_ERR_INVALID, "pattern is NULL");
    int rc = check_view(v, true);
    if (rc) return rc;
    if (more) *more = 0;
    if (v->n_rows == 0) return NEEDLE_OK;
    if (!d_counts || (slots && !d_blocks)) return fail(NEEDLE_ERR_INVALID, "output buffer is NULL");
    if (!offsets16_ok(v, 65535u)) return fail(NEEDLE_ERR_UNSUPPORTED, "16-bit start / end: rows of at most 65535 chars");
    return find_all_one_pass(p, v, slots, d_counts, nullptr, nullptr, nullptr, false, more, (hipStream_t)stream_, d_blocks, 6u);
}

int needle_count_matches_dev(const needle_pattern *cp, const needle_batch_view *v, uint32_t *d_counts, void *stream_) {
    needle_pattern *p = const_cast<needle_pattern *>(cp);
    if (!p) return fail(NEEDLE_ERR_INVALID, "pattern is NULL");
    int rc = check_view(v, true);
    if (rc) return rc;
    if (v->n_rows == 0) return NEEDLE_OK;
    if (!d_counts) return fail(NEEDLE_ERR_INVALID, "output buffer is NULL");
    return find_all_one_pass(p, v, 0, d_counts, nullptr, nullptr, nullptr, true, nullptr, (hipStream_t)stream_);
}

int needle_find_all_csr_dev(const needle_pattern *cp, const needle_batch_view *v, const uint64_t *d_offsets, int32_t *d_start, int32_t *d_end,
                            int *more, void *stream_) {
    needle_pattern *p = const_cast<needle_pattern *>(cp);
    if (!p) return fail(NEEDLE_ERR_INVALID, "pattern is NULL");
    int rc = check_view(v, true);
    if (rc) return rc;
    if (more) *more = 0;
    if (v->n_rows == 0) return NEEDLE_OK;
    if (!d_offsets || !d_start || !d_end) return fail(NEEDLE_ERR_INVALID, "offsets / output buffer is NULL");
    return find_all_one_pass(p, v, 0, nullptr, d_start, d_end, d_offsets, false, more, (hipStream_t)stream_);
}

// start_end16 != nullptr: the one-dword-per-match form (needle_find_all_packed16_dev) -- start / end are not used
static int find_all_host_one(const needle_pattern *p, const needle_batch_view *v, uint32_t slots, uint32_t *counts, int32_t *start,
                             int32_t *end, int *more, uint32_t *start_end16 = nullptr) {
    const size_t cw = v->char_width, n = (size_t)v->n_rows;
    const size_t src_stride = (size_t)v->row_stride * cw;
    size_t dst_stride = (src_stride + 15) & ~(size_t)15;
    if (dst_stride == 0) dst_stride = 16;
    uint8_t *d = nullptr; // rows | lengths | counts | start | end
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t o_len = up16(n * dst_stride), o_cnt = o_len + up16(n * 4), o_s = o_cnt + up16(n * 4);
    const size_t o_e = o_s + up16(n * slots * 4), total = o_e + (start_end16 ? 0 : up16(n * slots * 4));
    HIP_TRY(hipMalloc((void **)&d, total));
    auto done = [&](int code) {
        (void)hipFree(d);
        return code;
    };
    hipError_t e = hipSuccess;
    if (dst_stride == src_stride) {
        e = hipMemcpy(d, v->rows, n * src_stride, hipMemcpyHostToDevice);
    } else {
        e = hipMemset(d, 0, n * dst_stride);
        if (e == hipSuccess && src_stride) e = hipMemcpy2D(d, dst_stride, v->rows, src_stride, src_stride, n, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && v->lengths) e = hipMemcpy(d + o_len, v->lengths, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && slots) e = hipMemset(d + o_s, 0xFF, total - o_s); // -1 in every slot
    if (e != hipSuccess) return done(hip_fail(e, "find_all_host upload"));
    needle_batch_view dv = *v;
    dv.rows = d;
    dv.lengths = v->lengths ? (const uint32_t *)(d + o_len) : nullptr;
    dv.row_stride = dst_stride / cw;
    int rc = start_end16 ? needle_find_all_packed16_dev(p, &dv, slots, (uint32_t *)(d + o_cnt), (uint32_t *)(d + o_s), more, nullptr)
                         : needle_find_all_dev(p, &dv, slots, (uint32_t *)(d + o_cnt), (int32_t *)(d + o_s), (int32_t *)(d + o_e), more, nullptr);
    if (rc) return done(rc);
    e = hipMemcpy(counts, d + o_cnt, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && slots) e = hipMemcpy(start_end16 ? (void *)start_end16 : (void *)start, d + o_s, n * slots * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && slots && !start_end16) e = hipMemcpy(end, d + o_e, n * slots * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return done(hip_fail(e, "find_all_host download"));
    return done(NEEDLE_OK);
}

// (like the other host entry points: at most ~2 GiB of rows + results resident on the device at a time)
// One chunk of needle_find_all_csr_host: upload, count pass, prefix sum on the host (the counts come back anyway), fill
// pass while the rows are still resident, download.  offsets: n + 1 entries, offsets[0] given by the caller.
static int find_all_csr_host_one(const needle_pattern *p, const needle_batch_view *v, uint64_t *offsets, int32_t *start, int32_t *end,
                                 uint64_t capacity) {
    const size_t cw = v->char_width, n = (size_t)v->n_rows;
    const size_t src_stride = (size_t)v->row_stride * cw;
    size_t dst_stride = (src_stride + 15) & ~(size_t)15;
    if (dst_stride == 0) dst_stride = 16;
    uint8_t *d = nullptr, *d_out = nullptr; // rows | lengths | counts | offsets;  start | end
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t o_len = up16(n * dst_stride), o_cnt = o_len + up16(n * 4), o_off = o_cnt + up16(n * 4), total = o_off + up16((n + 1) * 8);
    HIP_TRY(hipMalloc((void **)&d, total));
    auto done = [&](int code) {
        (void)hipFree(d);
        if (d_out) (void)hipFree(d_out);
        return code;
    };
    hipError_t e = hipSuccess;
    if (dst_stride == src_stride) {
        e = hipMemcpy(d, v->rows, n * src_stride, hipMemcpyHostToDevice);
    } else {
        e = hipMemset(d, 0, n * dst_stride);
        if (e == hipSuccess && src_stride) e = hipMemcpy2D(d, dst_stride, v->rows, src_stride, src_stride, n, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && v->lengths) e = hipMemcpy(d + o_len, v->lengths, n * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) return done(hip_fail(e, "find_all_csr_host upload"));
    needle_batch_view dv = *v;
    dv.rows = d;
    dv.lengths = v->lengths ? (const uint32_t *)(d + o_len) : nullptr;
    dv.row_stride = dst_stride / cw;
    int rc = needle_count_matches_dev(p, &dv, (uint32_t *)(d + o_cnt), nullptr);
    if (rc) return done(rc);
    std::vector<uint32_t> counts(n);
    e = hipMemcpy(counts.data(), d + o_cnt, n * 4, hipMemcpyDeviceToHost); // (synchronises with the count pass)
    if (e != hipSuccess) return done(hip_fail(e, "find_all_csr_host counts"));
    for (size_t r = 0; r < n; ++r) offsets[r + 1] = offsets[r] + counts[r];
    const uint64_t m = offsets[n] - offsets[0];
    if (m == 0 || offsets[n] > capacity) return done(NEEDLE_OK); // nothing to file, or the caller's buffers are too small
    // The fill pass runs over sub-ranges of the chunk's rows so that the results resident on the device stay bounded too:
    // a dense-match batch (a one-char pattern over 256-char rows files ~2 KiB per row) would otherwise ask for several
    // times the chunk's row bytes in one allocation.  NEEDLE_HOST_RESULT_BYTES: that bound (tests shrink it).
    static const uint64_t kResultBytes = getenv("NEEDLE_HOST_RESULT_BYTES") ? (uint64_t)atoll(getenv("NEEDLE_HOST_RESULT_BYTES")) : (512ull << 20);
    const uint64_t max_m = std::max<uint64_t>(kResultBytes / 8, 1);
    std::vector<std::pair<size_t, size_t>> ranges; // [r0, r1): at least one row, at most max_m matches (one row may exceed it)
    uint64_t biggest = 0;
    for (size_t r0 = 0; r0 < n;) {
        size_t r1 = r0 + 1;
        while (r1 < n && offsets[r1 + 1] - offsets[r0] <= max_m) ++r1;
        ranges.emplace_back(r0, r1);
        biggest = std::max<uint64_t>(biggest, offsets[r1] - offsets[r0]);
        r0 = r1;
    }
    e = hipMalloc((void **)&d_out, 2 * up16(biggest * 4) + 16);
    if (e != hipSuccess) return done(hip_fail(e, "find_all_csr_host results"));
    std::vector<uint64_t> local;
    for (const auto &rg : ranges) {
        const size_t r0 = rg.first, nr = rg.second - rg.first;
        const uint64_t mr = offsets[rg.second] - offsets[r0];
        if (mr == 0) continue;
        local.resize(nr + 1);
        for (size_t r = 0; r <= nr; ++r) local[r] = offsets[r0 + r] - offsets[r0];
        e = hipMemcpy(d + o_off, local.data(), (nr + 1) * 8, hipMemcpyHostToDevice);
        if (e != hipSuccess) return done(hip_fail(e, "find_all_csr_host offsets"));
        needle_batch_view sv = dv;
        sv.rows = d + r0 * dst_stride;
        sv.lengths = dv.lengths ? dv.lengths + r0 : nullptr;
        sv.n_rows = nr;
        int more = 0;
        rc = needle_find_all_csr_dev(p, &sv, (const uint64_t *)(d + o_off), (int32_t *)d_out, (int32_t *)(d_out + up16(biggest * 4)), &more, nullptr);
        if (rc) return done(rc);
        if (more) return done(fail(NEEDLE_ERR_DEVICE, "find_all_csr_host: count pass and fill pass disagree"));
        e = hipMemcpy(start + offsets[r0], d_out, mr * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(end + offsets[r0], d_out + up16(biggest * 4), mr * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return done(hip_fail(e, "find_all_csr_host download"));
    }
    return done(NEEDLE_OK);
}
int needle_find_all_csr_host(const needle_pattern *p, const needle_batch_view *v, uint64_t *offsets, int32_t *start, int32_t *end,
                             uint64_t capacity, uint64_t *total) {
    if (!p) return fail(NEEDLE_ERR_INVALID, "pattern is NULL");
    int rc = check_view(v, false);
    if (rc) return rc;
    if ((rc = check_host_lengths(v))) return rc;
    if (!offsets || !total || (capacity && (!start || !end))) return fail(NEEDLE_ERR_INVALID, "output buffer is NULL");
    offsets[0] = 0;
    *total = 0;
    if (v->n_rows == 0) return NEEDLE_OK;
    static const uint64_t kHostChunkBytes = getenv("NEEDLE_HOST_CHUNK_BYTES") ? (uint64_t)atoll(getenv("NEEDLE_HOST_CHUNK_BYTES")) : (2ull << 30);
    const uint64_t row_bytes = std::max<uint64_t>(16, (v->row_stride * v->char_width + 15) & ~(uint64_t)15) + 16;
    const uint64_t per = std::max<uint64_t>(64, (kHostChunkBytes / row_bytes) & ~(uint64_t)63);
    for (uint64_t r0 = 0; r0 < v->n_rows; r0 += per) {
        needle_batch_view c = *v;
        c.n_rows = std::min<uint64_t>(per, v->n_rows - r0);
        c.rows = (const uint8_t *)v->rows + r0 * v->row_stride * v->char_width;
        c.lengths = v->lengths ? v->lengths + r0 : nullptr;
        rc = find_all_csr_host_one(p, &c, offsets + r0, start, end, capacity);
        if (rc) return rc;
    }
    *total = offsets[v->n_rows];
    return NEEDLE_OK;
}
static int find_all_host(const needle_pattern *p, const needle_batch_view *v, uint32_t slots, uint32_t *counts, int32_t *start,
                         int32_t *end, int *more, uint32_t *start_end16) {
    if (!p) return fail(NEEDLE_ERR_INVALID, "pattern is NULL");
    int rc = check_view(v, false);
    if (rc) return rc;
    if ((rc = check_host_lengths(v))) return rc;
    if (more) *more = 0;
    if (v->n_rows == 0) return NEEDLE_OK;
    if (!counts || (slots && !start_end16 && (!start || !end))) return fail(NEEDLE_ERR_INVALID, "output buffer is NULL");
    static const uint64_t kHostChunkBytes = getenv("NEEDLE_HOST_CHUNK_BYTES") ? (uint64_t)atoll(getenv("NEEDLE_HOST_CHUNK_BYTES")) : (2ull << 30);
    const uint64_t row_bytes = std::max<uint64_t>(16, (v->row_stride * v->char_width + 15) & ~(uint64_t)15) + 8 + 8ull * slots;
    const uint64_t per = std::max<uint64_t>(64, (kHostChunkBytes / row_bytes) & ~(uint64_t)63);
    // (rows of at most 65 534 chars: an empty match at index 65 535 would read as an unfiled slot; NEEDLE_FIND_ALL_ROUNDS: the tests'
    // cross-check of the round-per-match form goes through needle_find_all_dev)
    static const bool rounds_forced = getenv("NEEDLE_FIND_ALL_ROUNDS") && atoi(getenv("NEEDLE_FIND_ALL_ROUNDS")) != 0;
    const bool packed_inside = !rounds_forced && (v->lengths ? v->row_stride : v->row_len) <= 65534u; // (lengths[r] <= row_stride: checked above)
    std::vector<uint32_t> stage;
    for (uint64_t r0 = 0; r0 < v->n_rows; r0 += per) {
        needle_batch_view c = *v;
        c.n_rows = std::min<uint64_t>(per, v->n_rows - r0);
        c.rows = (const uint8_t *)v->rows + r0 * v->row_stride * v->char_width;
        c.lengths = v->lengths ? v->lengths + r0 : nullptr;
        int m = 0;
        if (!start_end16 && slots && packed_inside) {
            // int32 results wanted, rows of at most 65 535 chars: the one-dword form on the device and over PCIe (half the result
            // bytes both ways), opened into the caller's two arrays here on the host
            stage.resize((size_t)c.n_rows * slots);
            rc = find_all_host_one(p, &c, slots, counts + r0, nullptr, nullptr, &m, stage.data());
            if (rc) return rc;
            int32_t *so = start + r0 * slots, *eo = end + r0 * slots;
            for (size_t i = 0; i < stage.size(); ++i) {
                const uint32_t w = stage[i];
                const bool none = w == 0xFFFFFFFFu; // an unfiled slot (a real match has start <= end, never 0xFFFF | 0xFFFF << 16)
                so[i] = none ? -1 : (int32_t)(w & 0xFFFFu);
                eo[i] = none ? -1 : (int32_t)(w >> 16);
            }
        } else {
            rc = find_all_host_one(p, &c, slots, counts + r0, start ? start + r0 * slots : nullptr, end ? end + r0 * slots : nullptr, &m,
                                   start_end16 ? start_end16 + r0 * slots : nullptr);
            if (rc) return rc;
        }
        if (m && more) *more = 1;
    }
    return NEEDLE_OK;
}
int needle_find_all_host(const needle_pattern *p, const needle_batch_view *v, uint32_t slots, uint32_t *counts, int32_t *start,
                         int32_t *end, int *more) {
    return find_all_host(p, v, slots, counts, start, end, more, nullptr);
}
int needle_find_all_packed16_host(const needle_pattern *p, const needle_batch_view *v, uint32_t slots, uint32_t *counts,
                                  uint32_t *start_end16, int *more) {
    if (slots && !start_end16) return fail(NEEDLE_ERR_INVALID, "output buffer is NULL");
    if (v && (v->lengths ? v->row_stride : v->row_len) > 65535u) // (the caller's stride, before any padding; lengths[r] <= row_stride is checked below)
        return fail(NEEDLE_ERR_UNSUPPORTED, "16-bit start / end: rows of at most 65535 chars");
    static uint32_t none = 0; // (slots == 0: counting only; a non-null marker keeps the packed form)
    return find_all_host(p, v, slots, counts, nullptr, nullptr, more, start_end16 ? start_end16 : &none);
}
int needle_matches_host(const needle_pattern *p, const needle_batch_view *v, uint64_t *bm) {
    return run_host(p, OP_MATCHES, v, bm, nullptr, nullptr);
}
int needle_contained_in_host(const needle_pattern *p, const needle_batch_view *v, uint64_t *bm) {
    return run_host(p, OP_CONTAINED_IN, v, bm, nullptr, nullptr);
}
int needle_find_host(const needle_pattern *p, const needle_batch_view *v, uint64_t *bm, int32_t *st, int32_t *en) {
    return run_host(p, OP_FIND, v, bm, st, en);
}

} // extern "C"

// ------------------------------------------------------------------------------------------------
// Matcher mirror: fields as the generated class declares them (DFAClassBuilder.addFields :688-699); the
// constructor leaves them at the JVM default 0 (the generated <init> only stores string and length).
struct needle_matcher {
    const needle_pattern *p;
    std::vector<uint16_t> s;
    int next_start = 0, start = 0, end = 0;
};

static int one_row(const needle_matcher *m, int op, int from, int *matched, int *st, int *en) {
    const size_t n = m->s.size() - (size_t)from;
    needle_batch_view v;
    memset(&v, 0, sizeof(v));
    v.rows = n ? (const void *)(m->s.data() + from) : (const void *)&v; // never read when n == 0
    v.char_width = 2;
    v.n_rows = 1;
    v.row_stride = n;
    v.row_len = (uint32_t)n;
    uint64_t bm = 0;
    int32_t s32 = -1, e32 = -1;
    int rc = run_host(m->p, op, &v, &bm, &s32, &e32);
    if (rc) return rc;
    *matched = (int)(bm & 1);
    if (st) *st = s32;
    if (en) *en = e32;
    return NEEDLE_OK;
}

extern "C" {

int needle_matcher_create(const needle_pattern *p, const uint16_t *s, size_t n, needle_matcher **out) {
    if (!p || !out || (!s && n)) return fail(NEEDLE_ERR_INVALID, "NULL argument");
    needle_matcher *m = new needle_matcher();
    m->p = p;
    m->s.assign(s, s + n);
    *out = m;
    return NEEDLE_OK;
}

void needle_matcher_destroy(needle_matcher *m) { delete m; }

int needle_matcher_matches(needle_matcher *m, int *r) {
    if (!m || !r) return fail(NEEDLE_ERR_INVALID, "NULL argument");
    return one_row(m, OP_MATCHES, 0, r, nullptr, nullptr);
}

int needle_matcher_contained_in(needle_matcher *m, int *r) {
    if (!m || !r) return fail(NEEDLE_ERR_INVALID, "NULL argument");
    return one_row(m, OP_CONTAINED_IN, 0, r, nullptr, nullptr);
}

// find(FROM, TO): DFAClassBuilder.createFindMethodInternal :625-659.  TO is ignored by the generated
// indexForwards (its slot is overwritten with this.length, DFAMethodComponents.java:19-21).
int needle_matcher_find_range(needle_matcher *m, int from, int to, int *r) {
    (void)to;
    if (!m || !r) return fail(NEEDLE_ERR_INVALID, "NULL argument");
    *r = 0;
    if (m->next_start == -1) return NEEDLE_OK; // :629-630
    const int length = (int)m->s.size();
    const RefDfa &fw = m->p->t.dfa[W_FORWARDS];
    int index;
    int st = 0;
    bool have_start = false;
    if (from < 0) return fail(NEEDLE_ERR_INVALID, "from < 0 (StringIndexOutOfBoundsException in the reference)");
    if (from >= length) {
        // both generated loops are skipped: indexForwards returns its initial lastMatch (:355-356,468)
        index = fw.accepting[0] ? 0 : -1;
    } else {
        int matched = 0, s32 = -1, e32 = -1;
        int rc = one_row(m, OP_FIND, from, &matched, &s32, &e32);
        if (rc) return rc;
        if (matched) {
            index = e32 + from;
            st = s32 + from; // the backward walk is bounded by FROM (:651-652) == index 0 of the sub-row
            have_start = true;
        } else {
            index = -1;
        }
    }
    m->end = index;
    m->next_start = index;
    if (index == -1) return NEEDLE_OK;
    if (!have_start) {
        // index came from the literal 0 above; start as the reference computes it with an empty walk
        if (m->p->t.fixed_len >= 0) st = index - m->p->t.fixed_len;
        else st = m->p->t.dfa[W_BACKWARDS].accepting[0] ? from : 0x7FFFFFFF; // :543-547 with index-1 < FROM
    }
    m->start = st;
    *r = 1;
    return NEEDLE_OK;
}

int needle_matcher_find(needle_matcher *m, int *r) {
    if (!m) return fail(NEEDLE_ERR_INVALID, "NULL argument");
    return needle_matcher_find_range(m, m->next_start, (int)m->s.size(), r); // :616-623
}

int needle_matcher_start(const needle_matcher *m) { return m ? m->start : -1; }
int needle_matcher_end(const needle_matcher *m) { return m ? m->end : -1; }

} // extern "C"
