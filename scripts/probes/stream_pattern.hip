// What HBM read rate do the kernels' FETCH PATTERNS reach by themselves on gfx950?  Persistent grid (one 1024-thread workgroup per CU, as the
// scan / filter kernels), every wave keeps PF 1 KiB loads (64 lanes x 16 bytes) in flight and xors what arrives; nothing else.  Patterns:
//   0  filter-like: a wave owns a CHUNK (16 KiB = a 64-row group of 256-byte rows) and reads it front to back, 1 KiB per load; the 16 waves
//      of a CU own 16 adjacent chunks; next chunk = + all the grid's waves (needle_ngram.hip)
//   1  CU-interleaved: the 16 waves of a CU read ONE 16-chunk region together, wave w the units w, w + 16, ...
//   2  scan-like: a chunk = 64 rows, a load = 128-byte lines of 8 rows (lane -> row l / 8 + 8 j, bytes (l & 7) * 16 of the line), lines left
//      to right (needle_scan.h)
//   3  wave-major: a wave owns one long contiguous range of the buffer (4096 streams spread over all of it)
//   4  like 0, the waves of a CU rotated: wave w starts its chunk at unit (5 w) mod units and wraps
//   5  like 0, but a wave takes its NEXT chunk from a global counter (atomicAdd) when it starts one: XCDs / CUs that get more bandwidth take more chunks
// hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_pattern scripts/probes/stream_pattern.hip && /tmp/stream_pattern
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int PF, int MODE>
__global__ __launch_bounds__(1024) void stream_kernel(const uint8_t *base, uint64_t total, uint32_t chunk, uint32_t stride, uint32_t *sink) {
    extern __shared__ uint8_t smem[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t n_waves = (uint64_t)gridDim.x * 16, gw = (uint64_t)blockIdx.x * 16 + wave;
    const uint32_t units = chunk >> 10, ushift = 31u - (uint32_t)__builtin_clz(units); // 1 KiB loads per chunk (a power of two >= PF)
    const uint64_t n_chunks = total / chunk, per = n_chunks / n_waves;
    uint32_t acc = 0;
    if (smem[threadIdx.x & 15] == 77) acc = 1; // (keeps the LDS allocation alive)
    // unit t of this wave's sequence -> address
    auto addr = [&](uint64_t t) -> const uint8_t * {
        uint64_t c, u = t & (units - 1u);
        const uint64_t i = t >> ushift;
        if (MODE == 3) {
            c = gw * per + i;
            if (i >= per) return base + (uint64_t)lane * 16u;
        } else if (MODE == 1) {
            // region r = 16 chunks of this CU; the wave's units inside it: w, w + 16, ...
            const uint64_t r = (uint64_t)blockIdx.x + i * gridDim.x;
            if ((r + 1) * 16 > n_chunks) return base + (uint64_t)lane * 16u;
            return base + r * 16u * chunk + (u * 16u + wave) * 1024u + lane * 16u;
        } else {
            c = gw + i * n_waves;
        }
        if (c >= n_chunks) return base + (uint64_t)lane * 16u;
        if (MODE == 2) { // units: line-major, 8 loads of 8 rows each per 128-byte line
            const uint32_t j = (uint32_t)u & 7u, line = (uint32_t)u >> 3;
            return base + c * chunk + (uint64_t)((lane >> 3) + 8u * j) * stride + line * 128u + (lane & 7u) * 16u;
        }
        if (MODE == 4) u = (u + 5u * wave) & (units - 1u);
        return base + c * chunk + u * 1024u + lane * 16u;
    };
    uint64_t my_units;
    if (MODE == 3) my_units = per * units;
    else if (MODE == 1) my_units = ((n_chunks / 16 + gridDim.x - 1) / gridDim.x) * units;
    else my_units = ((n_chunks + n_waves - 1) / n_waves) * units;
    u32x4 R[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) R[k] = *(const u32x4 *)addr((uint64_t)k);
    for (uint64_t t = 0; t < my_units; t += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const u32x4 v = R[k];
            asm volatile("" ::: "memory");
            R[k] = *(const u32x4 *)addr(t + PF + k);
            acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
        }
    }
#pragma unroll
    for (int k = 0; k < PF; ++k) acc ^= R[k][0];
    if (acc == 0x12345678u) sink[0] = acc;
}

// when do the XCDs finish?  (block b runs on XCD b % 8: the dispatcher deals workgroups round-robin)  The filter-like pattern, every wave's
// finish time (s_memtime since its start) written out: mean by XCD
template <int PF>
__global__ __launch_bounds__(1024) void stream_timed(const uint8_t *base, uint64_t total, uint32_t chunk, uint32_t *sink, uint64_t *finish) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t n_waves = (uint64_t)gridDim.x * 16, gw = (uint64_t)blockIdx.x * 16 + wave;
    const uint32_t units = chunk >> 10;
    const uint64_t n_chunks = total / chunk;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    for (uint64_t c = gw; c < n_chunks; c += n_waves) {
        const uint8_t *p = base + c * chunk + lane * 16u;
        u32x4 R[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) R[k] = *(const u32x4 *)(p + (uint64_t)k * 1024u);
        for (uint32_t u = 0; u < units; u += PF) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const u32x4 v = R[k];
                asm volatile("" ::: "memory");
                const uint32_t nu = u + PF + k < units ? u + PF + k : k;
                R[k] = *(const u32x4 *)(p + (uint64_t)nu * 1024u);
                acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
            }
        }
#pragma unroll
        for (int k = 0; k < PF; ++k) acc ^= R[k][0];
    }
    if (acc == 0x12345678u) sink[0] = acc;
    if (lane == 0) finish[gw] = __builtin_amdgcn_s_memtime() - t0;
}

template <int PF>
__global__ __launch_bounds__(1024) void stream_dynamic(const uint8_t *base, uint64_t total, uint32_t chunk, uint32_t *sink, uint32_t *counter) {
    extern __shared__ uint8_t smem[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t units = chunk >> 10;
    const uint32_t n_chunks = (uint32_t)(total / chunk);
    uint32_t acc = 0;
    if (smem[threadIdx.x & 15] == 77) acc = 1;
    auto grab = [&]() -> uint32_t {
        uint32_t n = 0;
        if (lane == 0) n = atomicAdd(counter, 1u);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
    };
    uint32_t c = grab(), c_next = grab();
    // the cursor (what is loaded next) runs PF units ahead of what is consumed
    uint32_t lc = c, lu = 0; // cursor: chunk, unit
    uint32_t ln = c_next;    // the chunk the cursor enters after lc
    auto next_addr = [&]() -> const uint8_t * {
        const uint8_t *p = lc < n_chunks ? base + (uint64_t)lc * chunk + (uint64_t)lu * 1024u + lane * 16u : base + (uint64_t)lane * 16u;
        if (++lu == units) {
            lu = 0;
            lc = ln;
            ln = lc < n_chunks ? grab() : lc;
        }
        return p;
    };
    u32x4 R[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) R[k] = *(const u32x4 *)next_addr();
    uint32_t cu = 0; // consumed units of chunk c
    while (c < n_chunks) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const u32x4 v = R[k];
            asm volatile("" ::: "memory");
            R[k] = *(const u32x4 *)next_addr();
            acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
        }
        cu += PF;
        if (cu == units) {
            cu = 0;
            c = c_next;       // (the consumer follows the cursor's sequence: c_next was grabbed when c began)
            c_next = lc == c ? ln : lc; // the chunk after: what the cursor holds
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int PF>
static void run_dynamic(const uint8_t *d, uint64_t total, uint32_t chunk, uint32_t *d_sink, size_t lds) {
    auto k = stream_dynamic<PF>;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    uint32_t *d_ctr;
    hipMalloc(&d_ctr, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipMemsetAsync(d_ctr, 0, 64, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(256), dim3(1024), lds, 0, d, total, chunk, d_sink, d_ctr);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    hipError_t e = hipGetLastError();
    printf("%-16s PF %d chunk %6u LDS %6zu: %.3f ms = %.2f TB/s%s\n", "dynamic", PF, chunk, lds, best, (double)total / best * 1e-9, e == hipSuccess ? "" : " (ERROR)");
    hipFree(d_ctr);
}

template <int PF, int MODE>
static void run(const char *name, const uint8_t *d, uint64_t total, uint32_t chunk, uint32_t stride, uint32_t *d_sink, size_t lds) {
    auto k = stream_kernel<PF, MODE>;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(256), dim3(1024), lds, 0, d, total, chunk, stride, d_sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    hipError_t e = hipGetLastError();
    printf("%-16s PF %d chunk %6u LDS %6zu: %.3f ms = %.2f TB/s%s\n", name, PF, chunk, lds, best, (double)total / best * 1e-9, e == hipSuccess ? "" : " (ERROR)");
}

// the same stream with an UNEVEN static split: block b streams the contiguous chunk range [range[b], range[b + 1]), its 16 waves interleaved
// inside it -- ranges proportional to per-XCD weights (block b on XCD b % 8)
template <int PF>
__global__ __launch_bounds__(1024) void stream_ranges(const uint8_t *base, uint32_t chunk, const uint32_t *range, uint32_t *sink, uint64_t *finish) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t units = chunk >> 10;
    const uint32_t c0 = range[blockIdx.x], c1 = range[blockIdx.x + 1];
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    for (uint32_t c = c0 + wave; c < c1; c += 16) {
        const uint8_t *p = base + (uint64_t)c * chunk + lane * 16u;
        u32x4 R[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) R[k] = *(const u32x4 *)(p + (uint64_t)k * 1024u);
        for (uint32_t u = 0; u < units; u += PF) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const u32x4 v = R[k];
                asm volatile("" ::: "memory");
                const uint32_t nu = u + PF + k < units ? u + PF + k : k;
                R[k] = *(const u32x4 *)(p + (uint64_t)nu * 1024u);
                acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
            }
        }
#pragma unroll
        for (int k = 0; k < PF; ++k) acc ^= R[k][0];
    }
    if (acc == 0x12345678u) sink[0] = acc;
    if (lane == 0) finish[(uint64_t)blockIdx.x * 16 + wave] = __builtin_amdgcn_s_memtime() - t0;
}

static void run_ranges(const uint8_t *d, uint64_t total, uint32_t *d_sink, const double (&w)[8], const char *tag) {
    const uint32_t chunk = 16384, n_chunks = (uint32_t)(total / chunk);
    std::vector<uint32_t> range(257, 0);
    double sum = 0;
    for (int b = 0; b < 256; ++b) sum += w[b % 8];
    double accw = 0;
    for (int b = 0; b < 256; ++b) {
        accw += w[b % 8];
        range[b + 1] = (uint32_t)((double)n_chunks * accw / sum);
    }
    range[256] = n_chunks;
    uint32_t *d_range;
    uint64_t *d_fin;
    hipMalloc(&d_range, 257 * 4);
    hipMalloc(&d_fin, 4096 * 8);
    hipMemcpy(d_range, range.data(), 257 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(stream_ranges<4>, dim3(256), dim3(1024), 0, 0, d, chunk, d_range, d_sink, d_fin);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    std::vector<uint64_t> h(4096);
    hipMemcpy(h.data(), d_fin, 4096 * 8, hipMemcpyDeviceToHost);
    double xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mx = 0;
    for (int i = 0; i < 4096; ++i) {
        xcd[(i / 16) % 8] += (double)h[i];
        if ((double)h[i] > mx) mx = (double)h[i];
    }
    printf("ranges %-22s: %.3f ms = %.2f TB/s; finish max %.0f, by XCD:", tag, best, (double)total / best * 1e-9, mx);
    for (int k = 0; k < 8; ++k) printf(" %.0f", xcd[k] / 512);
    printf("\n");
    hipFree(d_range);
    hipFree(d_fin);
}

static void run_timed(const uint8_t *d, uint64_t total, uint32_t *d_sink) {
    uint64_t *d_fin;
    hipMalloc(&d_fin, 4096 * 8);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(stream_timed<4>, dim3(256), dim3(1024), 0, 0, d, total, 16384u, d_sink, d_fin);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(4096);
    hipMemcpy(h.data(), d_fin, 4096 * 8, hipMemcpyDeviceToHost);
    double xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mx = 0, mean = 0;
    for (int w = 0; w < 4096; ++w) {
        xcd[(w / 16) % 8] += (double)h[w];
        mean += (double)h[w];
        if ((double)h[w] > mx) mx = (double)h[w];
    }
    printf("filter-like, wave finish times (cycles): mean %.0f max %.0f; by XCD:", mean / 4096, mx);
    for (int k = 0; k < 8; ++k) printf(" %.0f", xcd[k] / 512);
    printf("\n");
    hipFree(d_fin);
}

int main() {
    const uint64_t total = 10000000ull * 256ull / 16384ull * 16384ull; // whole 16 KiB chunks of the bench batch (10^7 rows of 256 bytes)
    uint8_t *d;
    uint32_t *d_sink;
    if (hipMalloc(&d, total + 65536) != hipSuccess) return 1;
    hipMalloc(&d_sink, 64);
    hipMemset(d, 1, total + 65536);
    run_timed(d, total, d_sink);
    {
        const double even[8] = {1, 1, 1, 1, 1, 1, 1, 1};
        run_ranges(d, total, d_sink, even, "equal shares");
        const double w90[8] = {1, 0.93, 1, 0.93, 1, 0.93, 1, 0.93};
        run_ranges(d, total, d_sink, w90, "odd XCDs 0.93");
        const double w87[8] = {1, 0.87, 1, 0.87, 1, 0.87, 1, 0.87};
        run_ranges(d, total, d_sink, w87, "odd XCDs 0.87");
        const double w80[8] = {1, 0.80, 1, 0.80, 1, 0.80, 1, 0.80};
        run_ranges(d, total, d_sink, w80, "odd XCDs 0.80");
        const double inv[8] = {0.87, 1, 0.87, 1, 0.87, 1, 0.87, 1};
        run_ranges(d, total, d_sink, inv, "EVEN XCDs 0.87");
    }
    for (size_t lds : {(size_t)0, (size_t)160 * 1024}) {
        run<4, 0>("filter-like", d, total, 16384, 256, d_sink, lds);
        run<8, 0>("filter-like", d, total, 16384, 256, d_sink, lds);
        run<4, 0>("filter-like", d, total, 65536, 256, d_sink, lds);
        run<4, 1>("cu-interleaved", d, total, 16384, 256, d_sink, lds);
        run<8, 1>("cu-interleaved", d, total, 16384, 256, d_sink, lds);
        run<4, 2>("scan-like", d, total, 16384, 256, d_sink, lds);
        run<8, 2>("scan-like", d, total, 16384, 256, d_sink, lds);
        run<4, 3>("wave-major", d, total, 16384, 256, d_sink, lds);
        run<8, 3>("wave-major", d, total, 16384, 256, d_sink, lds);
        run<4, 4>("filter-rotated", d, total, 16384, 256, d_sink, lds);
        run<8, 4>("filter-rotated", d, total, 16384, 256, d_sink, lds);
        run_dynamic<4>(d, total, 16384, d_sink, lds);
        run_dynamic<8>(d, total, 16384, d_sink, lds);
        run_dynamic<4>(d, total, 65536, d_sink, lds);
    }
    return 0;
}
