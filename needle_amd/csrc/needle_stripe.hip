// Kernels beside the tiled scan (needle_kernels.hip), gfx950 only:
//   unpack_kernel                      haystacks packed back to back -> the fixed-stride layout
//   stripe_kernel / stripe_prefix_kernel / backward_row_kernel
//                                      few, long rows: per-stripe transition functions composed across a row
//                                      (intra-row parallelism for packed-mode automata)
//   short_kernel                       rows of at most 64 bytes: register-resident, no LDS transposition
#include "needle_walk.h"

namespace needle {

// ------------------------------------------------------------------------------------------------
// packed ("CSR") rows -> fixed-stride rows: one thread per 16-byte piece of the output.  Output stores are
// lane-linear 16-byte pieces (fully coalesced); the source of a piece starts at an arbitrary byte of the packed
// buffer, so it is read as five ALIGNED dwords and funnel-shifted (v_alignbyte) into place.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unpack_kernel(const uint8_t *__restrict__ data, const uint64_t *__restrict__ offsets,
                                                     uint64_t n_rows, uint32_t cw, uint8_t *__restrict__ out,
                                                     uint64_t stride_bytes, uint32_t *__restrict__ lengths,
                                                     int32_t *overflow) {
    const uint64_t ppr = stride_bytes >> 4; // pieces per row
    const uint64_t total = n_rows * ppr;
    const uint64_t end_bytes = offsets[n_rows] * cw;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t row = q / ppr;
        const uint32_t k = (uint32_t)(q - row * ppr);
        const uint64_t b0 = offsets[row] * cw;
        uint64_t len_b = offsets[row + 1] * cw - b0;
        if (len_b > stride_bytes) {
            len_b = stride_bytes;
            if (k == 0 && overflow) *overflow = 1;
        }
        if (k == 0) lengths[row] = (uint32_t)(len_b / cw);
        const uint64_t pos = (uint64_t)k * 16u; // byte position of this piece inside the row
        u32x4 v = {0, 0, 0, 0};
        if (pos < len_b) {
            const uint64_t src = b0 + pos;
            const uint64_t base = src & ~(uint64_t)3;
            const uint32_t sh = (uint32_t)(src & 3u);
            uint32_t d[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) d[i] = (base + 4u * i < end_bytes) ? *(const uint32_t *)(data + base + 4u * i) : 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
            const uint64_t valid = len_b - pos; // bytes of the row in this piece (>= 16: all of it)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint64_t lo = 4u * i;
                if (valid <= lo) v[i] = 0;
                else if (valid < lo + 4u) v[i] &= (1u << (8u * (uint32_t)(valid - lo))) - 1u;
            }
        }
        *(u32x4 *)(out + row * stride_bytes + pos) = v;
    }
}

hipError_t launch_unpack(const void *data, const uint64_t *offsets, uint64_t n_rows, uint32_t cw, void *out,
                         uint64_t stride_bytes, uint32_t *lengths, int32_t *overflow, int n_cus, hipStream_t stream) {
    const uint64_t total = n_rows * (stride_bytes >> 4);
    uint64_t blocks = (total + 255) / 256;
    const uint64_t cap = (uint64_t)(n_cus > 0 ? n_cus : 256) * 16;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint8_t *)data, offsets, n_rows, cw,
                       (uint8_t *)out, stride_bytes, lengths, overflow);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Long rows (packed mode): intra-row parallelism by function composition.
//
// One row per lane cannot use the chip when there are few, long rows.  With <= 6 device states the per-char
// transition function F[c] is one dword (5-bit fields); so is the function of ANY substring: compose(A then B)
// field i = B[A[i]].  A wave takes one 4 KiB stripe: each lane loads its own 64 contiguous bytes straight into
// registers (no LDS transposition needed: a lane's bytes are contiguous), walks them for ALL entry states at once
// (one v_bfe_u32 per state and char), and an in-wave ordered scan composes the 64 lane functions.
//   pass 1  stripe_kernel<CW, false>: every stripe's function                        -> fn[row][stripe]
//   prefix  stripe_prefix_kernel: per row, sequentially over its stripes: entry state of each stripe (in place),
//           final state -> matches()/containedIn() verdict
//   pass 2  stripe_kernel<CW, true>  (find): lane entry state = stripe entry state through the exclusive lane scan,
//           then the lane walks its bytes again tracking the last accepting position; lanes after the automaton died
//           enter in the sink and accept nothing, so lastMatch (DFAClassBuilder.java:438-468) is simply the MAX of the
//           accepting positions: atomicMax per row.  Round 3: only ONE stripe per row is walked again.  Pass 1 also notes, per
//           stripe and tracked entry state, whether the stripe passes through an accepting state at all (one v_or per char
//           and state: accepting states are the odd field offsets) -- the prefix pass, which knows every stripe's true entry
//           state, picks the row's last such stripe, and lastMatch lies in it.  (Before: every stripe of every row was read
//           and walked twice -- find() on rows without an early match cost 2.3 x containedIn().)
//   start   backward_row_kernel: indexBackwards from lastMatch - 1, one lane per row
// ------------------------------------------------------------------------------------------------
// Field offsets of the packed functions (ProgHeader::pack_off): the sink at 0, the other states' fields wherever the
// lowering put them.  Only the first n_states fields exist.
struct Fields {
    uint32_t off[5];
    uint32_t ident; // the identity function: field i holds off[i]
    int n;
};
__device__ __forceinline__ Fields fields_of(const ProgHeader &h) {
    Fields f;
#pragma unroll
    for (int i = 0; i < 5; ++i) f.off[i] = h.pack_off[i];
    f.ident = h.ident_fn;
    f.n = (int)h.n_states;
    return f;
}

// B after A: field i of the result = B[A[i]]
__device__ __forceinline__ uint32_t compose_fn(const Fields &fl, uint32_t a, uint32_t b) {
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
        if (i < fl.n) r |= __builtin_amdgcn_ubfe(b, __builtin_amdgcn_ubfe(a, fl.off[i], 5), 5) << fl.off[i];
    return r;
}

// NS = device states incl. the sink (2..5).  The sink maps to itself under every char, so only states 1 .. NS-1 are
// tracked: NS - 1 v_bfe_u32 per char.
template <int CW, bool FIND, int NS, bool CAND = false>
__global__ __launch_bounds__(kWavesPerBlock * 64) void stripe_kernel(const StripeArgs a) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();
    for (uint32_t i = tid * 16u; i < a.hdr.lds_bytes; i += blockDim.x * 16u) *(u32x4 *)(smem + i) = *(const u32x4 *)(a.prog + i);
    __syncthreads();
    Walk wk;
    wk.ncols_e = 0, wk.pad_e = a.hdr.pad_f, wk.pre_e = a.hdr.pre_f, wk.table_off = 0, wk.gtable = nullptr;
    wk.win_on = 0, wk.win_lo = 0, wk.win_hi = 0, wk.sp_chains = 0, wk.sp_pad_ident = 0, wk.dead_hi = 0;
    wk.lane4 = (uint32_t)(lane & 31) * 4u;
    constexpr int CPL = 64 / CW; // chars per lane and stripe
    const uint32_t accept_lo = a.hdr.accept_off;
    const Fields fl = fields_of(a.hdr);
    const uint32_t kIdentFn = fl.ident;
    const bool one_per_row = FIND && a.cand_stripe != nullptr; // (pass 2: the row's candidate stripe only)
    const uint64_t total = one_per_row ? a.n_rows : a.n_rows * a.spr;
    if constexpr (!FIND) {
        // ---- pass 1, software-pipelined (round 5).  The loop below it loads a stripe and waits for it at once: every iteration a full
        // memory latency, hidden by the CU's other 15 waves only (1000 x 1 MiB rows: 3.8 TB/s).  Here the NEXT stripe of this wave is
        // in flight while the current one is walked and scanned.  As in short_kernel (below): nothing is pending at the loop head, a
        // stripe's loads are ALWAYS four (addresses clamped into the row's stride -- a conditional load makes the compiler's vmcnt
        // conservative at the join), its row length travels with it, and it is collected before this iteration's stores enter the queue.
        const uint64_t step = (uint64_t)gridDim.x * kWavesPerBlock;
        uint64_t it = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
        if (it >= total) return;
        const bool has_len = a.lengths != nullptr;
        struct Stripe { uint64_t row; uint32_t s; };
        auto place = [&](uint64_t v) __attribute__((always_inline)) -> Stripe {
            Stripe st;
            st.row = v / a.spr;
            st.s = (uint32_t)(v - st.row * a.spr);
            return st;
        };
        auto fetch = [&](const Stripe &st, u32x4 (&d)[4], uint32_t &d_len) __attribute__((always_inline)) {
            const uint64_t off = (uint64_t)st.s * kStripeBytes + (uint64_t)lane * 64u;
            const uint8_t *base = a.rows + st.row * a.stride_bytes;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint64_t o = off + 16u * j;
                o = o + 16u <= a.stride_bytes ? o : a.stride_bytes - 16u; // (past the stride: the row's last 16 bytes, masked by n_valid)
                d[j] = *(const u32x4 *)(base + o);
            }
            if (has_len) d_len = a.lengths[st.row];
        };
        auto landed = [&](u32x4 (&d)[4], uint32_t &d_len) __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d_len));
        };
        Stripe cur_st = place(it);
        u32x4 nxt[4];
        uint32_t nxt_len = a.row_len;
        fetch(cur_st, nxt, nxt_len);
        landed(nxt, nxt_len);
        for (;;) {
            u32x4 d[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = nxt[j];
            const uint32_t len = nxt_len;
            const uint64_t v = it;
            const Stripe me = cur_st;
            const uint64_t itn = it + step;
            const bool more = itn < total;
            cur_st = place(more ? itn : it);
            fetch(cur_st, nxt, nxt_len); // (the last iteration re-reads its own stripe: unused)
            const uint64_t first = (uint64_t)me.s * (kStripeBytes / CW) + (uint64_t)lane * CPL; // this lane's first char
            const uint32_t n_valid = first >= len ? 0u : (uint32_t)(len - first < (uint64_t)CPL ? len - first : CPL);
            const bool beyond = (uint64_t)me.s * (kStripeBytes / CW) >= len; // stripe past the row's end (wave-uniform)
            uint32_t incl = kIdentFn, bits = 0;
            if (!beyond) {
                uint32_t g[NS], seen[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) g[i] = fl.off[i], seen[i] = 0;
                const bool full = __ballot(n_valid != (uint32_t)CPL) == 0ull;
                auto walk_all = [&](auto per_char) __attribute__((always_inline)) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t w[4] = {d[j][0], d[j][1], d[j][2], d[j][3]};
                        uint32_t f[16 / CW];
#define NEEDLE_F(D, K) f[(D) * (4 / CW) + (K)] = lookup<MODE_PACK, CW, false, K>(wk, w[D], true, false);
#pragma unroll
                        for (int dd = 0; dd < 4; ++dd) {
                            NEEDLE_F(dd, 0)
                            NEEDLE_F(dd, 1)
                            if (CW == 1) {
                                NEEDLE_F(dd, 2)
                                NEEDLE_F(dd, 3)
                            }
                        }
#undef NEEDLE_F
                        lds_fence();
#pragma unroll
                        for (int i = 0; i < 16 / CW; ++i) per_char(j * (16 / CW) + i, f[i]);
                    }
                };
                if (full) {
                    walk_all([&](int, uint32_t f) __attribute__((always_inline)) {
#pragma unroll
                        for (int i = 1; i < NS; ++i) {
                            g[i] = __builtin_amdgcn_ubfe(f, g[i], 5);
                            if (CAND) seen[i] |= g[i];
                        }
                    });
                } else {
                    walk_all([&](int c, uint32_t f) __attribute__((always_inline)) {
                        const bool in_row = (uint32_t)c < n_valid;
                        f = in_row ? f : kIdentFn;
#pragma unroll
                        for (int i = 1; i < NS; ++i) {
                            g[i] = __builtin_amdgcn_ubfe(f, g[i], 5);
                            if (CAND) seen[i] |= in_row ? g[i] : 0u;
                        }
                    });
                }
                uint32_t fn = 0;
#pragma unroll
                for (int i = 0; i < NS; ++i) fn |= g[i] << fl.off[i];
                incl = fn; // ordered inclusive scan over the lanes: after step d lane l holds the function of lanes l-2d+1 .. l
#pragma unroll
                for (int dlt = 1; dlt < 64; dlt <<= 1) {
                    const uint32_t left = (uint32_t)__shfl_up((int)incl, dlt);
                    if (lane >= dlt) incl = compose_fn(fl, left, incl);
                }
                if (CAND) {
                    uint32_t excl = (uint32_t)__shfl_up((int)incl, 1);
                    if (lane == 0) excl = kIdentFn;
#pragma unroll
                    for (int i = 1; i < NS; ++i) {
                        const uint32_t e = __builtin_amdgcn_ubfe(excl, fl.off[i], 5); // this lane's entry state when the stripe is entered in state i
                        uint32_t sv = 0;
#pragma unroll
                        for (int k = 1; k < NS; ++k) sv = e == fl.off[k] ? seen[k] : sv; // (the sink: nothing seen)
                        if (__ballot((sv & 1u) != 0u) != 0ull) bits |= 1u << i;
                    }
                }
            }
            landed(nxt, nxt_len); // the next stripe, before this one's stores are queued
            if (lane == 63) a.fn[v] = incl; // (a stripe past its row's end: the identity -- every lane holds it then)
            if (CAND && lane == 0) a.cand[v] = bits;
            if (!more) break;
            it = itn;
        }
        return;
    }
    for (uint64_t it = (uint64_t)blockIdx.x * kWavesPerBlock + wave; it < total; it += (uint64_t)gridDim.x * kWavesPerBlock) {
        uint64_t v = it;
        if (one_per_row) {
            const int32_t cs = a.cand_stripe[it];
            if (cs < 0) continue;
            v = it * a.spr + (uint32_t)cs;
        }
        const uint64_t row = v / a.spr;
        const uint32_t s = (uint32_t)(v - row * a.spr);
        const uint32_t len = a.lengths ? a.lengths[row] : a.row_len;
        const uint64_t first = (uint64_t)s * (kStripeBytes / CW) + (uint64_t)lane * CPL; // this lane's first char
        const uint32_t n_valid = first >= len ? 0u : (uint32_t)(len - first < (uint64_t)CPL ? len - first : CPL);
        uint32_t entry = 0; // FIND: 5 * state in which the row's automaton reaches this stripe
        if (FIND) entry = a.fn[v];
        if ((uint64_t)s * (kStripeBytes / CW) >= len || (FIND && entry == 0)) { // stripe past the row's end / automaton dead
            if (!FIND && lane == 0) {
                a.fn[v] = kIdentFn;
                if (CAND) a.cand[v] = 0u;
            }
            continue;
        }
        u32x4 d[4] = {};
        const uint64_t off = (uint64_t)s * kStripeBytes + (uint64_t)lane * 64u;
        if (n_valid) { // 64 B inside the row's stride (len <= stride, stride a multiple of 16 B)
            const uint8_t *p = a.rows + row * a.stride_bytes + off;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (off + 16u * j < a.stride_bytes) d[j] = *(const u32x4 *)(p + 16 * j);
        }
        // the per-char functions of this lane's chars (kept for the second walk of FIND)
        uint32_t g[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) g[i] = fl.off[i];
        const bool full = __ballot(n_valid != (uint32_t)CPL) == 0ull;
        auto walk_all = [&](auto per_char) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t w[4] = {d[j][0], d[j][1], d[j][2], d[j][3]};
                uint32_t f[16 / CW];
#define NEEDLE_F(D, K) f[(D) * (4 / CW) + (K)] = lookup<MODE_PACK, CW, false, K>(wk, w[D], true, false);
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    NEEDLE_F(dd, 0)
                    NEEDLE_F(dd, 1)
                    if (CW == 1) {
                        NEEDLE_F(dd, 2)
                        NEEDLE_F(dd, 3)
                    }
                }
#undef NEEDLE_F
                lds_fence();
#pragma unroll
                for (int i = 0; i < 16 / CW; ++i) per_char(j * (16 / CW) + i, f[i]);
            }
        };
        uint32_t seen[NS]; // CAND: OR of the states the walk from entry state i went through (bit 0: some of them accept)
#pragma unroll
        for (int i = 0; i < NS; ++i) seen[i] = 0;
        if (full) {
            walk_all([&](int, uint32_t f) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 1; i < NS; ++i) {
                    g[i] = __builtin_amdgcn_ubfe(f, g[i], 5);
                    if (CAND) seen[i] |= g[i];
                }
            });
        } else {
            walk_all([&](int c, uint32_t f) __attribute__((always_inline)) {
                const bool in_row = (uint32_t)c < n_valid;
                f = in_row ? f : kIdentFn;
#pragma unroll
                for (int i = 1; i < NS; ++i) {
                    g[i] = __builtin_amdgcn_ubfe(f, g[i], 5);
                    if (CAND) seen[i] |= in_row ? g[i] : 0u;
                }
            });
        }
        uint32_t fn = 0;
#pragma unroll
        for (int i = 0; i < NS; ++i) fn |= g[i] << fl.off[i];
        // ordered inclusive scan over the lanes: after step d lane l holds the function of lanes l-2d+1 .. l
        uint32_t incl = fn;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const uint32_t left = (uint32_t)__shfl_up((int)incl, dlt);
            if (lane >= dlt) incl = compose_fn(fl, left, incl);
        }
        uint32_t excl = (uint32_t)__shfl_up((int)incl, 1);
        if (lane == 0) excl = kIdentFn;
        if (!FIND) {
            if (lane == 63) a.fn[v] = incl;
            if (CAND) {
                uint32_t bits = 0;
#pragma unroll
                for (int i = 1; i < NS; ++i) {
                    const uint32_t e = __builtin_amdgcn_ubfe(excl, fl.off[i], 5); // this lane's entry state when the stripe is entered in state i
                    uint32_t sv = 0;
#pragma unroll
                    for (int k = 1; k < NS; ++k) sv = e == fl.off[k] ? seen[k] : sv; // (the sink: nothing seen)
                    if (__ballot((sv & 1u) != 0u) != 0ull) bits |= 1u << i;
                }
                if (lane == 0) a.cand[v] = bits;
            }
            continue;
        }
        uint32_t st = __builtin_amdgcn_ubfe(excl, entry, 5);
        int32_t last_rel = -1;
        if (full) {
            walk_all([&](int c, uint32_t f) __attribute__((always_inline)) {
                st = __builtin_amdgcn_ubfe(f, st, 5);
                last_rel = st >= accept_lo ? c + 1 : last_rel;
            });
        } else {
            walk_all([&](int c, uint32_t f) __attribute__((always_inline)) {
                st = __builtin_amdgcn_ubfe((uint32_t)c < n_valid ? f : kIdentFn, st, 5);
                last_rel = (st >= accept_lo && (uint32_t)c < n_valid) ? c + 1 : last_rel;
            });
        }
        int32_t best = last_rel >= 0 ? (int32_t)first + last_rel : -1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int32_t t = __shfl_xor(best, o);
            best = t > best ? t : best;
        }
        if (lane == 0 && best >= 0) atomicMax(a.end + row, best);
    }
}

// fn[] -> entry states, and the verdicts.  One workgroup per row: thread t composes its contiguous chunk of stripe
// functions, an ordered workgroup scan (wave shuffles + 16 wave totals through LDS) gives every chunk its entry
// function, and a second pass over the chunk replaces each function by the stripe's entry state.  (One thread per
// row walking 262144 stripes of a 1 GiB row one dependent load at a time took 25 ms.)
__global__ __launch_bounds__(1024) void stripe_prefix_kernel(const StripeArgs a) {
    __shared__ uint32_t wave_total[16];
    __shared__ int32_t wave_cand[16];
    const uint32_t accept_lo = a.hdr.accept_off;
    const Fields fl = fields_of(a.hdr);
    const uint32_t kIdentFn = fl.ident;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = (blockDim.x + 63) >> 6;
    const uint32_t per = (a.spr + blockDim.x - 1) / blockDim.x; // stripes per thread
    for (uint64_t row = blockIdx.x; row < a.n_rows; row += gridDim.x) {
        uint32_t *fn = a.fn + row * a.spr;
        const uint32_t s0 = (uint32_t)tid * per < a.spr ? (uint32_t)tid * per : a.spr;
        const uint32_t s1 = s0 + per < a.spr ? s0 + per : a.spr;
        uint32_t mine = kIdentFn;
        for (uint32_t s = s0; s < s1; ++s) mine = compose_fn(fl, mine, fn[s]);
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t left = (uint32_t)__shfl_up((int)incl, d);
            if (lane >= d) incl = compose_fn(fl, left, incl);
        }
        if (lane == 63) wave_total[wave] = incl;
        __syncthreads();
        uint32_t before = kIdentFn; // everything in the waves before this one
        for (int w = 0; w < wave; ++w) before = compose_fn(fl, before, wave_total[w]);
        uint32_t excl = (uint32_t)__shfl_up((int)incl, 1);
        if (lane == 0) excl = kIdentFn;
        excl = compose_fn(fl, before, excl);
        uint32_t q = __builtin_amdgcn_ubfe(excl, a.hdr.start_off, 5);
        int32_t last_cand = -1;
        for (uint32_t s = s0; s < s1; ++s) {
            const uint32_t f = fn[s];
            fn[s] = q;
            if (a.cand) { // (find) entered in q, does stripe s pass through an accepting state?
                const uint32_t c = a.cand[row * a.spr + s];
                uint32_t k = 0;
#pragma unroll
                for (int i = 1; i < 5; ++i) k = (i < fl.n && q == fl.off[i]) ? (uint32_t)i : k;
                if (k && ((c >> k) & 1u)) last_cand = (int32_t)s;
            }
            q = __builtin_amdgcn_ubfe(f, q, 5);
        }
        if (a.cand) { // the row's last candidate stripe: maximum over the workgroup
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const int32_t t = __shfl_xor(last_cand, o);
                last_cand = t > last_cand ? t : last_cand;
            }
            if (lane == 0) wave_cand[wave] = last_cand;
            __syncthreads();
            if (tid == 0) {
                int32_t best = -1;
                for (int w = 0; w < n_waves; ++w) best = wave_cand[w] > best ? wave_cand[w] : best;
                a.cand_stripe[row] = best;
            }
        }
        if (tid == (int)blockDim.x - 1) { // its chunk is the last one (possibly empty): q is the row's final state
            if (a.op == OP_FIND) a.end[row] = a.hdr.root_accepting ? 0 : -1; // :356 lastMatch before the first char
            else if (q >= accept_lo) atomicOr((unsigned long long *)(a.bitmap + (row >> 6)), 1ull << (row & 63));
        }
        __syncthreads(); // wave_total is reused by the next row
    }
    (void)n_waves;
}

// find(): matched bit + start (DFAClassBuilder.java:640-656) for the long-row path; one lane per row, everything
// read from global memory (the rows here are few).
template <int CW>
__global__ __launch_bounds__(256) void backward_row_kernel(const StripeArgs a) {
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < a.n_rows; base += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t row = base + threadIdx.x;
        bool res = false;
        if (row < a.n_rows) {
            const int32_t last = a.end[row];
            res = last >= 0;
            int32_t s = -1;
            if (res) {
                if (a.fixed_len >= 0) {
                    s = last - a.fixed_len;
                } else {
                    const uint8_t *bcmap = a.prog + a.hdr.off_bcmap, *bptab = a.prog + a.hdr.off_bptab, *bpages = a.prog + a.hdr.off_bpages;
                    const uint16_t *bt = (const uint16_t *)(a.bprog + a.bhdr.off_table);
                    const uint8_t *rowp = a.rows + row * a.stride_bytes;
                    uint32_t bs = a.bhdr.start;
                    int32_t lastb = a.bhdr.root_accepting ? 0 : INT_MAX;
                    for (int32_t p = last - 1; p >= 0; --p) {
                        const uint32_t c = (CW == 1) ? rowp[p] : ((const uint16_t *)rowp)[p];
                        bs = bt[bs * a.bhdr.n_cols + column_of<CW>(bcmap, bptab, bpages, c)];
                        if (bs == 0) break;
                        if (bs >= a.bhdr.accept_lo) lastb = p;
                    }
                    s = lastb;
                }
            }
            a.start[row] = s;
            if (!res) a.end[row] = -1;
        }
        const uint64_t word = __ballot(res);
        if ((threadIdx.x & 63) == 0 && row < a.n_rows) a.bitmap[row >> 6] = word;
    }
}

hipError_t launch_backward_rows(int char_width, const StripeArgs &a, hipStream_t stream) {
    const unsigned pblocks = (unsigned)((a.n_rows + 255) / 256);
    if (char_width == 1) hipLaunchKernelGGL(backward_row_kernel<1>, dim3(pblocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(backward_row_kernel<2>, dim3(pblocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

template <int CW, bool FIND, int NS, bool CAND>
static hipError_t launch_stripe(const StripeArgs &a, dim3 grid, size_t lds, hipStream_t stream) {
    auto k = stripe_kernel<CW, FIND, NS, CAND>;
    static thread_local uint64_t configured = 0;
    if (hipError_t e = allow_full_lds((const void *)k, configured); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, dim3(kWavesPerBlock * 64), lds, stream, a);
    return hipGetLastError();
}
template <int CW, bool FIND, bool CAND = false>
static hipError_t launch_stripe_n(const StripeArgs &a, dim3 grid, size_t lds, hipStream_t stream) {
    switch (a.hdr.n_states) {
    case 0: case 1: case 2: return launch_stripe<CW, FIND, 2, CAND>(a, grid, lds, stream);
    case 3: return launch_stripe<CW, FIND, 3, CAND>(a, grid, lds, stream);
    case 4: return launch_stripe<CW, FIND, 4, CAND>(a, grid, lds, stream);
    default: return launch_stripe<CW, FIND, 5, CAND>(a, grid, lds, stream); // packed mode has at most 5 states
    }
}

hipError_t launch_long_rows(int char_width, const StripeArgs &a, int n_cus, hipStream_t stream) {
    const uint64_t total = a.n_rows * a.spr;
    uint64_t blocks = (total + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks > (uint64_t)n_cus) blocks = (uint64_t)n_cus;
    const size_t lds = (a.hdr.lds_bytes + 15u) & ~15u;
    const dim3 grid((unsigned)blocks);
    const unsigned pblocks = (unsigned)((a.n_rows + 255) / 256);
    const bool cand = a.op == OP_FIND && a.cand != nullptr;
    hipError_t e = cand ? (char_width == 1 ? launch_stripe_n<1, false, true>(a, grid, lds, stream) : launch_stripe_n<2, false, true>(a, grid, lds, stream))
                        : (char_width == 1 ? launch_stripe_n<1, false>(a, grid, lds, stream) : launch_stripe_n<2, false>(a, grid, lds, stream));
    if (e != hipSuccess) return e;
    {
        if (a.op != OP_FIND) { // verdict bits are OR-ed in
            e = hipMemsetAsync(a.bitmap, 0, ((a.n_rows + 63) / 64) * 8, stream);
            if (e != hipSuccess) return e;
        }
        unsigned threads = 64;
        while (threads < 1024 && threads < a.spr) threads <<= 1;
        const uint64_t max_blocks = (uint64_t)n_cus * (2048 / threads);
        hipLaunchKernelGGL(stripe_prefix_kernel, dim3((unsigned)(a.n_rows < max_blocks ? a.n_rows : max_blocks)), dim3(threads), 0, stream, a);
    }
    if (a.op == OP_FIND) {
        dim3 grid2 = grid;
        if (cand) { // one stripe per row
            const uint64_t b2 = (a.n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
            grid2 = dim3((unsigned)(b2 < (uint64_t)n_cus ? b2 : (uint64_t)n_cus));
        }
        e = char_width == 1 ? launch_stripe_n<1, true>(a, grid2, lds, stream) : launch_stripe_n<2, true>(a, grid2, lds, stream);
        if (e != hipSuccess) return e;
        if (char_width == 1) hipLaunchKernelGGL(backward_row_kernel<1>, dim3(pblocks), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(backward_row_kernel<2>, dim3(pblocks), dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Speculative stripes: long rows of table-mode automata (SpecArgs, needle_device.h).
//
// Function composition (the stripe path above) needs the whole transition function in a register; an automaton of
// hundreds of states has none.  Search automata forget, though: whatever state a stripe is entered in, after a few
// chars the walk is in the same state as one that started from the start state.  So every stripe is first scanned
// from the start state as a row of its own (the tiled kernel: all stripes in parallel, full speed), and then
//   spec_init_kernel     stripe lengths; entry guess of stripe k + 1 = speculative end state of stripe k
//   spec_fix_kernel      one lane per stripe whose entry guess is not the start state: walks the true run (from the
//                        guess) and the speculative run side by side until their states meet -- from there on the
//                        speculative results hold; before, the true run's do.  Hands its true end state to the next
//                        stripe; a changed hand-over raises `changed` and the host runs another round (a fixpoint:
//                        normally reached in two rounds; bounded, with the one-lane walk as the fallback).
//   spec_reduce_kernel   per row: lastMatch = max over its stripes (stripes behind the automaton's death are entered in
//                        the sink and accept nothing), verdict bits
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spec_len_kernel(SpecArgs a) {
    const uint64_t total = a.n_rows * a.spr;
    const uint32_t per = a.stripe_bytes / a.char_width; // chars per stripe
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t row = v / a.spr;
        const uint32_t k = (uint32_t)(v - row * a.spr);
        const uint32_t len = a.lengths ? a.lengths[row] : a.row_len;
        const uint64_t first = (uint64_t)k * per;
        a.slen[v] = first >= len ? 0u : (uint32_t)(len - first < per ? len - first : per);
    }
}

__global__ __launch_bounds__(256) void spec_init_kernel(SpecArgs a) {
    const uint64_t total = a.n_rows * a.spr;
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t k = (uint32_t)(v % a.spr);
        a.entry[v] = k == 0 ? a.hdr.start : a.spec_end_state[v - 1];
        a.entry_done[v] = 0xFFFFFFFFu;
    }
}

template <int CW>
__global__ __launch_bounds__(256) void spec_fix_kernel(SpecArgs a) {
    const uint64_t total = a.n_rows * a.spr;
    const uint8_t *cmap = a.gprog + (CW == 1 ? (uint32_t)kLdsCmap1 : 0u);
    const uint8_t *ptab = a.gprog + kLdsPtab2, *pages = a.gprog + kLdsPages2Table;
    const uint16_t *table = (const uint16_t *)(a.gprog + a.hdr.off_table);
    const uint32_t n_cols = a.hdr.n_cols, start = a.hdr.start, accept_lo = a.hdr.accept_lo;
    const bool find = a.op != OP_CONTAINED_IN; // find and matches(): the sink is the absorbing state; containedIn: accepts are
    const bool positions = a.op == OP_FIND;
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t t = a.entry[v];
        if (a.entry_done[v] == t) continue; // results are already those of this entry state
        a.entry_done[v] = t;
        const uint32_t k = (uint32_t)(v % a.spr);
        const uint32_t n = a.slen[v];
        uint32_t end_state;
        int32_t last;
        const int32_t spec_last = positions ? a.spec_last[v] : (a.op == OP_MATCHES ? -1 : (((a.spec_bitmap[v >> 6] >> (v & 63)) & 1) ? 1 : -1));
        if (t == start || n == 0) { // the speculative run IS the true run (an empty stripe hands its entry state on)
            end_state = n == 0 ? t : a.spec_end_state[v];
            last = n == 0 ? -1 : spec_last;
        } else if (find && t == 0) { // entered in the sink: nothing can happen any more
            end_state = 0;
            last = -1;
        } else if (!find && t >= accept_lo) { // containedIn: already accepted (absorbing)
            end_state = t;
            last = 1;
        } else {
            const uint64_t row = v / a.spr;
            const uint8_t *p = a.rows + row * a.stride_bytes + (uint64_t)k * a.stripe_bytes;
            uint32_t qt = t, qs = start;
            int32_t last_t = -1;
            uint32_t i = 0;
            bool met = false;
            for (; i < n; ++i) {
                const uint32_t c = CW == 1 ? p[i] : ((const uint16_t *)p)[i];
                uint32_t col;
                if (CW == 1) col = ((const uint16_t *)cmap)[c]; // HBM-table layout: cmap16 holds the plain column
                else col = pages[(((uint32_t)((const uint16_t *)ptab)[c >> 8])) + (c & 255u)];
                qt = table[qt * n_cols + col];
                qs = table[qs * n_cols + col];
                if (qt >= accept_lo) last_t = positions ? (int32_t)(i + 1) : 1;
                if (qt == qs) {
                    met = true;
                    ++i;
                    break;
                }
                if (!find && qt >= accept_lo) break; // containedIn: decided
                // find / matches: the true run died (e.g. the stripe was entered in an accepting state and the match is over) --
                // the sink is absorbing and the speculative run never is in it at a live char, so the two would not meet before
                // the stripe's end: 4096 one-lane chars for nothing (Sherlock over 1 GiB of long rows: 1.04 of find()'s 1.35 ms)
                if (find && qt == 0u) break;
            }
            if (met) { // from char i on the two runs are one
                end_state = a.spec_end_state[v];
                if (positions) last = spec_last > (int32_t)i ? spec_last : last_t;
                else if (a.op == OP_MATCHES) last = -1;
                // containedIn: accepting states are absorbing, so a speculative run that had accepted BEFORE the meeting
                // point would have made the common state accepting (and last_t set): an accept it reports lies behind it
                else last = (last_t > 0 || spec_last > 0) ? 1 : -1;
            } else {
                end_state = qt;
                last = last_t;
            }
        }
        a.true_end_state[v] = end_state;
        a.true_last[v] = last;
        // a stripe that ends in the sink (find) or in an accepting, absorbing state (containedIn) ends the row's story:
        // nothing is handed on (the reduction stops at the first such stripe), so a death does not crawl through the
        // rest of the row one stripe per round
        const bool terminal = find ? end_state == 0u : end_state >= accept_lo;
        if (!terminal && k + 1 < a.spr && a.entry[v + 1] != end_state) {
            a.entry[v + 1] = end_state;
            *a.changed = 1;
        }
    }
}

__global__ __launch_bounds__(256) void spec_reduce_kernel(SpecArgs a) {
    __shared__ uint32_t first_terminal;
    __shared__ int32_t best;
    const uint32_t per = a.stripe_bytes / a.char_width;
    const bool find = a.op != OP_CONTAINED_IN; // (the sink ends the story; containedIn: an accept does)
    const uint32_t accept_lo = a.hdr.accept_lo;
    for (uint64_t row = blockIdx.x; row < a.n_rows; row += gridDim.x) {
        if (threadIdx.x == 0) first_terminal = a.spr - 1, best = -1;
        __syncthreads();
        // the first stripe in which the row's story ends (see spec_fix_kernel); stripes behind it were never settled
        uint32_t d = a.spr - 1;
        for (uint32_t k = threadIdx.x; k < a.spr; k += blockDim.x) {
            const uint32_t e = a.true_end_state[row * a.spr + k];
            if (find ? e == 0u : e >= accept_lo) {
                d = k;
                break;
            }
        }
        atomicMin(&first_terminal, d);
        __syncthreads();
        d = first_terminal;
        int32_t m = -1;
        for (uint32_t k = threadIdx.x; k <= d; k += blockDim.x) {
            const int32_t l = a.true_last[row * a.spr + k];
            if (l >= 0) {
                const int32_t pos = a.op == OP_FIND ? (int32_t)(k * per) + l : 1;
                m = pos > m ? pos : m;
            }
        }
        atomicMax(&best, m);
        __syncthreads();
        if (threadIdx.x == 0) {
            if (a.op == OP_FIND) a.end[row] = best;
            else if (a.op == OP_CONTAINED_IN) {
                if (best >= 0) atomicOr((unsigned long long *)(a.bitmap + (row >> 6)), 1ull << (row & 63));
            } else { // matches(): no stripe ended in the sink, and the state behind the last char is accepting
                const uint32_t e_last = a.true_end_state[row * a.spr + a.spr - 1];
                const bool died = a.true_end_state[row * a.spr + first_terminal] == 0u;
                if (!died && e_last >= accept_lo) atomicOr((unsigned long long *)(a.bitmap + (row >> 6)), 1ull << (row & 63));
            }
        }
        __syncthreads();
    }
}

hipError_t launch_spec_len(const SpecArgs &a, hipStream_t stream) {
    const uint64_t total = a.n_rows * a.spr;
    hipLaunchKernelGGL(spec_len_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_spec_init(const SpecArgs &a, hipStream_t stream) {
    const uint64_t total = a.n_rows * a.spr;
    hipLaunchKernelGGL(spec_init_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_spec_fix(const SpecArgs &a, hipStream_t stream) {
    const uint64_t total = a.n_rows * a.spr;
    const dim3 grid((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096));
    if (a.char_width == 1) hipLaunchKernelGGL(spec_fix_kernel<1>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(spec_fix_kernel<2>, grid, dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_spec_reduce(const SpecArgs &a, hipStream_t stream) {
    hipLaunchKernelGGL(spec_reduce_kernel, dim3((unsigned)(a.n_rows < 4096 ? a.n_rows : 4096)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// find-all bookkeeping (needle_find_all_dev): after round k of needle_find_next, file every row's match in its slot
// k, count it, and move the row's cursor (Matcher.nextStart) -- an empty match ends its row, as does no match.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void find_all_collect_kernel(uint64_t n_rows, uint32_t slots, uint32_t k, const int32_t *s,
                                                               const int32_t *e, int32_t *cursor, uint32_t *counts,
                                                               int32_t *starts, int32_t *ends, int32_t *any_hit) {
    bool hit_any = false;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * blockDim.x) {
        const int32_t en = e[r], st = s[r];
        const int32_t from = cursor[r]; // the cursor this round searched from (< 0: the row was exhausted already)
        if (k == 0) counts[r] = 0;
        // en < st: a nullable pattern searched from cursor == length reports end = the literal 0 of
        // DFAClassBuilder.java:356 with start = length; the reference's repeated find() would cycle for ever there.
        // That wrapped pseudo-match is dropped and ends the row.
        if (from >= 0 && en >= 0 && en >= st) {
            if (k < slots) {
                starts[r * slots + k] = st;
                ends[r * slots + k] = en;
                counts[r] = k + 1;
            }
            // the row goes on only while the cursor advances (an empty match, or any match that does not end beyond
            // the cursor it was searched from, is filed once and ends it)
            cursor[r] = (en == st || en <= from) ? -1 : en;
            hit_any = true;
        } else {
            cursor[r] = -1;
        }
    }
    if (__ballot(hit_any) != 0ull && (threadIdx.x & 63) == 0) *any_hit = 1;
}

hipError_t launch_find_all_collect(uint64_t n_rows, uint32_t slots, uint32_t k, const int32_t *s, const int32_t *e, int32_t *cursor,
                                   uint32_t *counts, int32_t *starts, int32_t *ends, int32_t *any_hit, int n_cus, hipStream_t stream) {
    uint64_t blocks = (n_rows + 255) / 256;
    const uint64_t cap = (uint64_t)(n_cus > 0 ? n_cus : 256) * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(find_all_collect_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, n_rows, slots, k, s, e, cursor, counts,
                       starts, ends, any_hit);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Short rows (stride <= 64 bytes): no LDS transposition at all.  A row is at most four 16-byte pieces, so every lane
// loads ITS OWN row straight into registers (64 lanes x 16 B at stride 16..64: whole lines per wave instruction) and
// walks it there; the next group's loads are in flight meanwhile.  The tiled kernel spends a whole 128-byte tile
// step per 64 rows whatever their length (45 G rows/s: 0.76 TB/s on 16-byte rows).  Always "guarded": per-row lengths
// and find() cursors cost a few selects on at most 64 chars.
// ------------------------------------------------------------------------------------------------
// GUARD = false: every row fills its stride and there are no cursors -- the per-char length / cursor selects go, and packed
// automata log find()'s accept flags (one v_alignbit per char) instead of selecting a position per char (needle_walk.h).
// LDS slot per lane for the backward walk's text: the row's stride + one dword, so that the lanes' slots are an ODD number of dwords
// apart (5 / 9 / 13 / 17): the dword writes and the walk's byte reads at equal offsets are conflict-free (round 4: 80 bytes, 4-way)
__host__ __device__ constexpr uint32_t short_slot_bytes(uint32_t stride_bytes) { return stride_bytes + 4u; }
template <int OP, int CW, int MODE, bool GUARD>
__global__ __launch_bounds__(kWavesPerBlock * 64) void short_kernel(const ScanArgs a) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();
    for (uint32_t i = tid * 16u; i < a.hdr.lds_bytes; i += blockDim.x * 16u) *(u32x4 *)(smem + i) = *(const u32x4 *)(a.prog + i);
    __syncthreads();
    Walk wk;
    constexpr uint32_t ELEM = (MODE == MODE_TABLE16) ? 2u : 1u;
    wk.ncols_e = a.hdr.n_cols * ELEM;
    wk.pad_e = (MODE == MODE_PACK) ? a.hdr.pad_f : a.hdr.pad_col * ELEM;
    wk.pre_e = (MODE == MODE_PACK) ? a.hdr.pre_f : (a.hdr.pad_col + 1u) * ELEM;
    wk.pad_b = wk.pre_b = 0;
    if (MODE == MODE_PAIR) {
        wk.ncols_e = a.hdr.n_cols * a.hdr.n_cols * 2u;
        wk.pad_e = a.hdr.pad_col * a.hdr.n_cols * 2u;
        wk.pre_e = (a.hdr.pad_col + 1u) * a.hdr.n_cols * 2u;
        wk.pad_b = a.hdr.pad_col * 2u;
        wk.pre_b = (a.hdr.pad_col + 1u) * 2u;
    }
    wk.table_off = a.hdr.off_table - a.hdr.win_lo_e; // (window addressing, needle_device.h)
    wk.win_on = a.hdr.win_on, wk.win_lo = a.hdr.win_lo_e, wk.win_hi = a.hdr.win_hi_e;
    wk.sp_chains = 0, wk.sp_pad_ident = 0;
    wk.dead_hi = OP == OP_FIND ? a.hdr.fa_dead_hi : 0u;
    wk.lane4 = (uint32_t)(lane & 31) * 4u;
    wk.gtable = (const uint16_t *)(a.prog + a.hdr.off_table);
    const uint32_t accept_lo = MODE == MODE_PACK ? a.hdr.accept_off : a.hdr.accept_lo;
    const uint32_t start_state = MODE == MODE_PACK ? a.hdr.start_off : a.hdr.start;
    constexpr int CPP = 16 / CW;
    const uint32_t n_pieces = (uint32_t)(a.stride_bytes >> 4); // 1..4, wave-uniform
    const uint64_t n_groups = (a.n_rows + 63) >> 6;
    const uint64_t wave_cnt = (uint64_t)gridDim.x * kWavesPerBlock;
    uint64_t g = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
    if (g >= n_groups) return;
    // (GUARD: the rows' lengths and find() cursors travel with the text, one group ahead; the unguarded kernel has neither)
    const bool has_len = GUARD && a.lengths != nullptr, has_from = GUARD && OP == OP_FIND && a.from != nullptr;
    auto fetch = [&](uint64_t grp, u32x4 (&d)[4], uint32_t &d_len, int32_t &d_from) __attribute__((always_inline)) {
        uint64_t row = (grp << 6) + lane;
        if (row >= a.n_rows) row = a.n_rows - 1; // lanes past the batch re-read its last row (verdict masked below)
        const uint8_t *p = a.rows + row * a.stride_bytes;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((uint32_t)j < n_pieces) d[j] = *(const u32x4 *)(p + 16 * j);
        if (has_len) d_len = a.lengths[row];
        if (has_from) d_from = a.from[row];
    };
    u32x4 cur[4] = {}, nxt[4] = {};
    uint32_t cur_len = a.row_len, nxt_len = a.row_len;
    int32_t cur_from = 0, nxt_from = 0;
    // The compiler counts vmcnt conservatively wherever paths with different numbers of loads / stores meet: with the first group's
    // loads still pending at the loop head, or this group's result stores issued ahead of the copy nxt -> cur, every wait became a
    // vmcnt(0) RIGHT BEHIND the prefetch it had just issued (each iteration a full memory latency, hidden only by the other waves).
    // So: nothing is pending at the loop head, and the prefetch is collected before the result stores are issued.
    auto landed = [&](u32x4 (&d)[4], uint32_t &d_len, int32_t &d_from) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d_len), "+v"(d_from));
    };
    fetch(g, cur, cur_len, cur_from);
    landed(cur, cur_len, cur_from);
    for (;;) {
        const uint64_t ng = g + wave_cnt;
        if (ng < n_groups) fetch(ng, nxt, nxt_len, nxt_from);
        const uint64_t my_row = (g << 6) + lane;
        const bool row_ok = my_row < a.n_rows;
        const uint32_t len = row_ok ? cur_len : 0u;
        int32_t cursor = 0;
        bool dead = false;
        if (has_from) {
            cursor = row_ok ? cur_from : -1;
            dead = cursor < 0; // find(): `if nextStart == -1 return false`, DFAClassBuilder.java:629-630
            if (dead) cursor = 0;
        }
        uint32_t st = dead ? 0u : start_state; // exhausted rows park in the sink
        int32_t last = -1;
        if (OP == OP_FIND && a.hdr.root_accepting) last = ((uint32_t)cursor < len) ? cursor : 0; // :356, :440
        int32_t last_rel = -1;
        constexpr bool HIST = !GUARD && OP == OP_FIND && MODE == MODE_PACK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if ((uint32_t)j < n_pieces) {
                const uint32_t w[4] = {cur[j][0], cur[j][1], cur[j][2], cur[j][3]};
                uint32_t acc_hist = 0;
                walk_piece<OP, CW, MODE, GUARD, HIST>(wk, w, (uint32_t)(j * CPP), len, (uint32_t)cursor, accept_lo, st, last_rel, &acc_hist);
                if (HIST && acc_hist) // the piece's accept flags: char i at bit 32 - CPP + i; the last accepting one is the highest set bit
                    last_rel = (int32_t)(j * CPP) + (31 - (int32_t)__builtin_clz(acc_hist)) - (32 - CPP) + 1;
            }
        }
        if (OP == OP_FIND) last = last_rel >= 0 ? last_rel : last;
        bool res;
        if (OP == OP_FIND) res = row_ok && !dead && (last >= 0);
        else res = row_ok && (st >= accept_lo);
        const uint64_t word = __ballot(res);
        int32_t s = -1, e = -1;
        if (OP == OP_FIND) {
            e = res ? last : -1;
            if (a.fixed_len >= 0) {
                s = res ? last - a.fixed_len : -1; // :640-646
            } else if (a.hdr.fa_len_off) { // the "lengths" automaton: the end state remembers the match length (needle_scan.h)
                s = res ? last - (int32_t)lds_u8(a.hdr.fa_len_off + st) : -1;
            } else if (a.short_window) {
                // indexBackwards(end - 1, FROM), :536-583, on the row's text parked in this lane's LDS slot (it is in
                // registers, which cannot be indexed per lane): one ds_read per char instead of a load from L2, and the packed /
                // popcount-compressed / small dense forms of the backward automaton that ride in the program (needle_walk.h)
                const uint32_t slot = ((a.hdr.lds_bytes + 15u) & ~15u) + ((uint32_t)wave * 64u + (uint32_t)lane) * short_slot_bytes((uint32_t)a.stride_bytes);
                if (res) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if ((uint32_t)j < n_pieces) {
#pragma unroll
                            for (int d = 0; d < 4; ++d) *(__attribute__((address_space(3))) uint32_t *)(uintptr_t)(slot + 16u * j + 4u * d) = cur[j][d];
                        }
                }
                const uint8_t *rowp = a.rows + (row_ok ? my_row : 0) * a.stride_bytes;
                // (rows of one piece: rounds of 4 chars -- what ends in 16 bytes is short, and half a round's lookups and chain go)
                const int32_t sb = n_pieces == 1u ? backward_walk<CW, false, 4>(a, res, last, cursor, slot, 0u, (uint32_t)a.stride_bytes, 0u, rowp)
                                                  : backward_walk<CW>(a, res, last, cursor, slot, 0u, (uint32_t)a.stride_bytes, 0u, rowp);
                s = res ? sb : -1;
            } else {
                // no room for the text slots behind the program: the same walk with an empty window, the text out of L2
                const uint8_t *rowp = a.rows + (row_ok ? my_row : 0) * a.stride_bytes;
                const int32_t sb = backward_walk<CW>(a, res, last, cursor, 16u, 0u, 0u, 0u, rowp);
                s = res ? sb : -1;
            }
        }
        // The next group's text is collected BEFORE this group's stores enter the queue (see `landed`), and on EVERY path to the loop
        // head, the last iteration's included (nothing is pending there): a path around the wait, even one no wave ever takes, makes
        // the compiler wait at the loop head again.
        landed(nxt, nxt_len, nxt_from);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((uint32_t)j < n_pieces) cur[j] = nxt[j];
        cur_len = nxt_len, cur_from = nxt_from;
        if (lane == 0) a.bitmap[g] = word;
        if (OP == OP_FIND && row_ok) {
            if (a.packed && a.packed8) { // wave-uniform: one uint16 per row (needle_find_packed8_dev)
                ((uint16_t *)a.packed)[my_row] = pack8(s, e);
            } else if (a.packed) { // wave-uniform: one dword per row (ScanArgs::packed)
                a.packed[my_row] = ((uint32_t)s & 0xFFFFu) | ((uint32_t)e << 16);
            } else {
                a.start[my_row] = s;
                a.end[my_row] = e;
            }
        }
        if (ng >= n_groups) break;
        g = ng;
    }
}

template <int OP, int CW, int MODE, bool GUARD>
static hipError_t launch_short_g(const ScanArgs &a, int grid, size_t lds, hipStream_t stream) {
    auto k = short_kernel<OP, CW, MODE, GUARD>;
    static thread_local uint64_t configured = 0;
    if (hipError_t e = allow_full_lds((const void *)k, configured); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(kWavesPerBlock * 64), lds, stream, a);
    return hipGetLastError();
}
template <int OP, int CW, int MODE>
static hipError_t launch_short_one(const ScanArgs &a_in, int grid, size_t lds, hipStream_t stream) {
    ScanArgs a = a_in;
    // every row fills its stride, no cursors: the unguarded walk
    const bool full = !a.lengths && !a.from && a.row_len != 0 && (uint64_t)a.row_len * CW == a.stride_bytes;
    // find() by indexBackwards: room for the lanes' text slots behind the program?
    a.short_window = 0;
    const size_t slots = (size_t)kWavesPerBlock * 64 * short_slot_bytes((uint32_t)a.stride_bytes);
    if (OP == OP_FIND && a.fixed_len < 0 && !a.hdr.fa_len_off && lds + slots <= 160u * 1024u) {
        a.short_window = 1;
        lds += slots;
    }
    return full ? launch_short_g<OP, CW, MODE, false>(a, grid, lds, stream) : launch_short_g<OP, CW, MODE, true>(a, grid, lds, stream);
}
template <int OP, int CW>
static hipError_t launch_short_m(const ScanArgs &a, int grid, size_t lds, hipStream_t s) {
    switch (a.hdr.mode) {
    case MODE_PACK: return launch_short_one<OP, CW, MODE_PACK>(a, grid, lds, s);
    case MODE_TABLE8: return launch_short_one<OP, CW, MODE_TABLE8>(a, grid, lds, s);
    case MODE_TABLE16: return launch_short_one<OP, CW, MODE_TABLE16>(a, grid, lds, s);
    case MODE_PAIR: return CW == 1 ? launch_short_one<OP, 1, MODE_PAIR>(a, grid, lds, s) : hipErrorInvalidValue;
    default: return launch_short_one<OP, CW, MODE_GLOBAL>(a, grid, lds, s);
    }
}
template <int OP>
static hipError_t launch_short_c(const ScanArgs &a, int cw, int grid, size_t lds, hipStream_t s) {
    return cw == 1 ? launch_short_m<OP, 1>(a, grid, lds, s) : launch_short_m<OP, 2>(a, grid, lds, s);
}

// rows of at most 64 bytes (stride a multiple of 16)
hipError_t launch_short_rows(int op, int char_width, const ScanArgs &a, int n_cus, hipStream_t stream) {
    const uint64_t n_groups = (a.n_rows + 63) >> 6;
    uint64_t blocks = (n_groups + kWavesPerBlock - 1) / kWavesPerBlock;
    const size_t lds = (a.hdr.lds_bytes + 15u) & ~15u;
    // matches() / containedIn() on rows of 16 and 32 bytes: the kernels need at most 64 VGPRs and no LDS beyond the program, so
    // TWO workgroups share a CU when two copies of the program fit its LDS -- 32 waves instead of 16 hide the memory latency of
    // rows this short (a wave has only 64 x stride bytes in flight): 16-byte rows 3.75 -> 4.9 TB/s, 32-byte rows 5.0 -> 5.3; rows
    // of 48 and 64 bytes lose 5 % that way and keep one.  find() has no room for a second copy beside its text slots.
    // NEEDLE_SHORT_WGS=1: one workgroup per CU everywhere (A/B).
    static const int wgs_env = getenv("NEEDLE_SHORT_WGS") ? atoi(getenv("NEEDLE_SHORT_WGS")) : 2;
    // find() on full rows (the unguarded kernels: at most 64 VGPRs) shares a CU the same way when two programs AND two sets of text slots
    // fit (table-mode programs; the packed program of 8-bit rows is 64 KB by itself)
    const bool full = !a.lengths && !a.from && a.row_len != 0 && (uint64_t)a.row_len * char_width == a.stride_bytes;
    const size_t find_lds = lds + ((a.fixed_len < 0 && !a.hdr.fa_len_off) ? (size_t)kWavesPerBlock * 64 * short_slot_bytes((uint32_t)a.stride_bytes) : 0);
    const bool two = wgs_env >= 2 && a.stride_bytes <= 32 && (op != OP_FIND ? 2 * lds <= 160u * 1024u : (full && 2 * find_lds <= 160u * 1024u));
    const uint64_t per_cu = two ? 2 : 1;
    if (blocks > (uint64_t)n_cus * per_cu) blocks = (uint64_t)n_cus * per_cu;
    switch (op) {
    case OP_MATCHES: return launch_short_c<OP_MATCHES>(a, char_width, (int)blocks, lds, stream);
    case OP_CONTAINED_IN: return launch_short_c<OP_CONTAINED_IN>(a, char_width, (int)blocks, lds, stream);
    default: return launch_short_c<OP_FIND>(a, char_width, (int)blocks, lds, stream);
    }
}

} // namespace needle
