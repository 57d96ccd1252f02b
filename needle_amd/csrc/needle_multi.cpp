// Row sharding over several devices of one node from ONE host process (config C4): what a JVM host
// uses.  Rows are independent units (a Matcher is per haystack, DFAClassBuilder.java:669-699), so every device scans
// its contiguous block of rows with the ordinary kernels on its own stream and there is no data-path collective; the
// only communication is the gather of the RESULTS to the root device (devices[0]): bitmap words, and for find() the
// per-row start / end.  Over RCCL that gather is one group of ncclSend / ncclRecv pairs -- every peer sends over its
// own direct xGMI link to the root (7 links in parallel) -- which is how NCCL / RCCL express a gather.  RCCL is loaded
// lazily (dlopen) the first time a handle over more than one distinct device is created: single-device use of the
// library never touches it.  When all shards sit on one device (tests, or oversubscribing one GPU) the gather is a
// device-to-device copy.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/needle_hip.h"

namespace needle {
int set_error(int code, const std::string &msg); // needle_api.cpp: the calling thread's needle_last_error()
}

namespace {

// ---- the few RCCL entry points used, resolved at run time (types as in rccl.h: ncclResult_t / ncclDataType_t are ints,
// ncclComm_t an opaque pointer; ncclUint64 = 5, ncclInt32 = 2)
typedef void *comm_t;
struct UniqueId { // ncclUniqueId: passed BY VALUE to ncclCommInitRank
    char internal[NEEDLE_UNIQUE_ID_BYTES];
};
struct Rccl {
    void *h = nullptr;
    int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(comm_t *, int, UniqueId, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, comm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load(std::string &err) {
        if (h) return true;
        // a process that already has an RCCL mapped (e.g. PyTorch's bundled one) must keep using that one
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            if ((h = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break;
        }
        if (!h)
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
            }
        if (!h) {
            err = std::string("RCCL not found: ") + dlerror();
            return false;
        }
#define NEEDLE_SYM(field, sym)                                                 \
    *(void **)(&field) = dlsym(h, sym);                                        \
    if (!field) {                                                              \
        err = std::string("RCCL symbol missing: ") + sym;                      \
        return false;                                                          \
    }
        NEEDLE_SYM(CommInitAll, "ncclCommInitAll")
        NEEDLE_SYM(GetUniqueId, "ncclGetUniqueId")
        NEEDLE_SYM(CommInitRank, "ncclCommInitRank")
        NEEDLE_SYM(AllGather, "ncclAllGather")
        NEEDLE_SYM(CommDestroy, "ncclCommDestroy")
        NEEDLE_SYM(GroupStart, "ncclGroupStart")
        NEEDLE_SYM(GroupEnd, "ncclGroupEnd")
        NEEDLE_SYM(Send, "ncclSend")
        NEEDLE_SYM(Recv, "ncclRecv")
        NEEDLE_SYM(GetErrorString, "ncclGetErrorString")
#undef NEEDLE_SYM
        return true;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mu;
constexpr int kNcclInt32 = 2, kNcclUint64 = 5;

int fail(int code, const std::string &msg) { return needle::set_error(code, msg); } // -> needle_last_error()

struct DeviceGuard { // restores the caller's current device
    int prev = 0;
    DeviceGuard() { (void)hipGetDevice(&prev); }
    ~DeviceGuard() { (void)hipSetDevice(prev); }
};

} // namespace

struct needle_multi {
    std::vector<int> dev;
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> done;
    std::vector<comm_t> comm; // per rank, when RCCL carries the gather
    bool rccl = false, loopback = false;
    int rank = -1, world = 0; // >= 0: one process per device (needle_multi_create_rank): comm[0] is this rank's communicator
    struct Buf {
        uint8_t *p = nullptr;
        size_t cap = 0;
    };
    std::vector<Buf> local; // per shard: results of a shard that is not written in place (bitmap | start | end | packed)
    Buf root_stage;         // on devices[0]: the peers' packed start / end (one dword per row) before they are unpacked
    hipEvent_t stage_free = nullptr; // recorded on the root's stream behind the unpack: the next call's pushes wait for it
    std::string transport_note; // why RCCL is not carrying the gather, if it was wanted
};

extern "C" {

int needle_multi_create(const int *devices, int n_devices, unsigned flags, needle_multi **out) {
    if (!out) return fail(NEEDLE_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) return fail(NEEDLE_ERR_INVALID, "devices: 1..64 entries");
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have < 1) return fail(NEEDLE_ERR_DEVICE, "no HIP device");
    bool distinct = true;
    for (int i = 0; i < n_devices; ++i) {
        if (devices[i] < 0 || devices[i] >= have) return fail(NEEDLE_ERR_INVALID, "device index out of range");
        for (int j = 0; j < i; ++j) distinct = distinct && devices[j] != devices[i];
    }
    DeviceGuard guard;
    needle_multi *m = new needle_multi();
    m->dev.assign(devices, devices + n_devices);
    m->stream.assign(n_devices, nullptr);
    m->done.assign(n_devices, nullptr);
    m->local.resize(n_devices);
    m->loopback = (flags & NEEDLE_MULTI_LOOPBACK) != 0;
    for (int i = 0; i < n_devices; ++i) {
        hipError_t e = hipSetDevice(m->dev[i]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->stream[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->done[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            needle_multi_destroy(m);
            return fail(NEEDLE_ERR_DEVICE, std::string("stream/event creation: ") + hipGetErrorString(e));
        }
    }
    // RCCL carries the gather between distinct devices (and, for tests of the plumbing on a single GPU, a one-rank
    // communicator sending to itself: NEEDLE_MULTI_LOOPBACK).  Shards that share a device are gathered by copies.
    if ((n_devices > 1 && distinct) || (m->loopback && n_devices == 1)) {
        // No RCCL (library not loadable, ncclCommInitAll refused, NEEDLE_MULTI_NO_RCCL=1): the gather falls back to
        // peer-to-peer copies over the same xGMI links (hipMemcpyPeerAsync on each shard's stream) -- a handle is still
        // made; needle_multi_transport() says which path it took and why.
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        std::string err;
        static const bool no_rccl = getenv("NEEDLE_MULTI_NO_RCCL") && atoi(getenv("NEEDLE_MULTI_NO_RCCL")) != 0;
        if (no_rccl) {
            m->transport_note = "NEEDLE_MULTI_NO_RCCL=1";
        } else if (!g_rccl.load(err)) {
            m->transport_note = err;
        } else {
            m->comm.assign(n_devices, nullptr);
            const int rc = g_rccl.CommInitAll(m->comm.data(), n_devices, m->dev.data());
            if (rc != 0) {
                m->comm.clear();
                m->transport_note = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(rc);
            } else {
                m->rccl = true;
            }
        }
        if (!m->rccl && m->loopback && n_devices == 1) { // (the loopback handle exists to exercise RCCL: without it, fail)
            const std::string why = m->transport_note;
            needle_multi_destroy(m);
            return fail(NEEDLE_ERR_DEVICE, why);
        }
        if (!m->rccl) // peers write into the root's memory directly when the runtime allows it (else it stages the copies)
            for (int i = 1; i < n_devices; ++i) {
                int can = 0;
                if (m->dev[i] != m->dev[0] && hipDeviceCanAccessPeer(&can, m->dev[i], m->dev[0]) == hipSuccess && can) {
                    (void)hipSetDevice(m->dev[i]);
                    (void)hipDeviceEnablePeerAccess(m->dev[0], 0); // (already enabled: an error we do not care about)
                    (void)hipGetLastError();
                }
            }
    }
    *out = m;
    return NEEDLE_OK;
}

void needle_multi_destroy(needle_multi *m) {
    if (!m) return;
    DeviceGuard guard;
    for (size_t i = 0; i < m->dev.size(); ++i) {
        (void)hipSetDevice(m->dev[i]);
        if (i < m->stream.size() && m->stream[i]) (void)hipStreamSynchronize(m->stream[i]);
        if (i < m->comm.size() && m->comm[i]) (void)g_rccl.CommDestroy(m->comm[i]);
        if (i < m->local.size() && m->local[i].p) (void)hipFree(m->local[i].p);
        if (i == 0 && m->root_stage.p) (void)hipFree(m->root_stage.p);
        if (i == 0 && m->stage_free) (void)hipEventDestroy(m->stage_free);
        if (i < m->done.size() && m->done[i]) (void)hipEventDestroy(m->done[i]);
        if (i < m->stream.size() && m->stream[i]) (void)hipStreamDestroy(m->stream[i]);
    }
    delete m;
}

// ---- one process per device: the communicator is built from an id that rank 0 creates and hands to every rank over any
// side channel (bench.py: a torch.distributed broadcast); the gathers run on a stream of the caller's choice.
int needle_multi_unique_id(void *id_out) {
    if (!id_out) return fail(NEEDLE_ERR_INVALID, "id_out is NULL");
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    std::string err;
    if (!g_rccl.load(err)) return fail(NEEDLE_ERR_DEVICE, err);
    UniqueId id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc != 0) return fail(NEEDLE_ERR_DEVICE, std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(rc));
    memcpy(id_out, &id, sizeof(id));
    return NEEDLE_OK;
}

int needle_multi_create_rank(const void *unique_id, int rank, int world, int device, needle_multi **out) {
    if (!out) return fail(NEEDLE_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!unique_id || world < 1 || rank < 0 || rank >= world) return fail(NEEDLE_ERR_INVALID, "unique_id / rank / world");
    {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        std::string err;
        if (!g_rccl.load(err)) return fail(NEEDLE_ERR_DEVICE, err);
    }
    DeviceGuard guard;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail(NEEDLE_ERR_INVALID, std::string("hipSetDevice: ") + hipGetErrorString(e));
    needle_multi *m = new needle_multi();
    m->dev.assign(1, device);
    m->stream.assign(1, nullptr);
    m->done.assign(1, nullptr);
    m->local.resize(1);
    m->rank = rank;
    m->world = world;
    e = hipStreamCreateWithFlags(&m->stream[0], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&m->done[0], hipEventDisableTiming);
    if (e != hipSuccess) {
        needle_multi_destroy(m);
        return fail(NEEDLE_ERR_DEVICE, std::string("stream/event creation: ") + hipGetErrorString(e));
    }
    UniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    m->comm.assign(1, nullptr);
    const int rc = g_rccl.CommInitRank(&m->comm[0], world, id, rank);
    if (rc != 0) {
        m->comm.clear();
        needle_multi_destroy(m);
        return fail(NEEDLE_ERR_DEVICE, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(rc));
    }
    m->rccl = true;
    *out = m;
    return NEEDLE_OK;
}

// The bitmap words of every rank to every rank (d_recv: world * count words, in rank order), on `stream`.
int needle_multi_all_gather_u64(needle_multi *m, const uint64_t *d_send, uint64_t count, uint64_t *d_recv, void *stream) {
    if (!m || m->rank < 0 || !d_send || !d_recv) return fail(NEEDLE_ERR_INVALID, "needs a handle from needle_multi_create_rank and buffers");
    const int rc = g_rccl.AllGather(d_send, d_recv, (size_t)count, kNcclUint64, m->comm[0], (hipStream_t)stream);
    if (rc != 0) return fail(NEEDLE_ERR_DEVICE, std::string("ncclAllGather: ") + g_rccl.GetErrorString(rc));
    return NEEDLE_OK;
}

// `count` int32 of every rank to rank 0 (d_recv_root: world * count, in rank order; ignored on the other ranks), on
// `stream`: one group of send / receive pairs, every peer over its own link to the root.
int needle_multi_gather_i32(needle_multi *m, const int32_t *d_send, uint64_t count, int32_t *d_recv_root, void *stream) {
    if (!m || m->rank < 0 || !d_send || (m->rank == 0 && !d_recv_root)) return fail(NEEDLE_ERR_INVALID, "needs a handle from needle_multi_create_rank and buffers");
    int rc = g_rccl.GroupStart();
    if (rc == 0) rc = g_rccl.Send(d_send, (size_t)count, kNcclInt32, 0, m->comm[0], (hipStream_t)stream);
    if (m->rank == 0)
        for (int r = 0; r < m->world && rc == 0; ++r)
            rc = g_rccl.Recv(d_recv_root + (size_t)r * count, (size_t)count, kNcclInt32, r, m->comm[0], (hipStream_t)stream);
    const int rc2 = g_rccl.GroupEnd();
    if (rc == 0) rc = rc2;
    if (rc != 0) return fail(NEEDLE_ERR_DEVICE, std::string("RCCL gather: ") + g_rccl.GetErrorString(rc));
    return NEEDLE_OK;
}

// find()'s start / end as one dword per row for the fan-in to rank 0: two 16-bit halves, 0xFFFF = -1 (no match).  Rows
// of up to 65 534 chars: halves the bytes every peer pushes through its link to the root (10 -> 5 MB per rank at 10M rows
// over 8 GPUs).
namespace {
__global__ void pack_start_end16_kernel(const int32_t *start, const int32_t *end, uint64_t n, uint32_t *out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = ((uint32_t)start[i] & 0xFFFFu) | ((uint32_t)end[i] << 16);
}
__global__ void unpack_start_end16_kernel(const uint32_t *in, uint64_t n, int32_t *start, int32_t *end) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = in[i], s = v & 0xFFFFu, e = v >> 16;
        start[i] = s == 0xFFFFu ? -1 : (int32_t)s;
        end[i] = e == 0xFFFFu ? -1 : (int32_t)e;
    }
}
unsigned pack_grid(uint64_t n) { return (unsigned)std::min<uint64_t>((n + 255) / 256, 2048); }
} // namespace

int needle_pack_start_end16_dev(const int32_t *d_start, const int32_t *d_end, uint64_t n, uint32_t *d_out, void *stream) {
    if (n == 0) return NEEDLE_OK;
    if (!d_start || !d_end || !d_out) return fail(NEEDLE_ERR_INVALID, "NULL buffer");
    hipLaunchKernelGGL(pack_start_end16_kernel, dim3(pack_grid(n)), dim3(256), 0, (hipStream_t)stream, d_start, d_end, n, d_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NEEDLE_ERR_DEVICE, std::string("pack kernel: ") + hipGetErrorString(e));
    return NEEDLE_OK;
}
int needle_unpack_start_end16_dev(const uint32_t *d_in, uint64_t n, int32_t *d_start, int32_t *d_end, void *stream) {
    if (n == 0) return NEEDLE_OK;
    if (!d_start || !d_end || !d_in) return fail(NEEDLE_ERR_INVALID, "NULL buffer");
    hipLaunchKernelGGL(unpack_start_end16_kernel, dim3(pack_grid(n)), dim3(256), 0, (hipStream_t)stream, d_in, n, d_start, d_end);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NEEDLE_ERR_DEVICE, std::string("unpack kernel: ") + hipGetErrorString(e));
    return NEEDLE_OK;
}

int needle_multi_device_count(const needle_multi *m) { return m ? (int)m->dev.size() : 0; }
void *needle_multi_stream(const needle_multi *m, int i) { return (m && i >= 0 && i < (int)m->stream.size()) ? (void *)m->stream[i] : nullptr; }

// "rccl" | "peer-copy: <why RCCL is not used>" | "local" (every shard on one device: plain device copies)
const char *needle_multi_transport(const needle_multi *m) {
    static thread_local std::string s;
    if (!m) return "";
    if (m->rccl) return "rccl";
    bool distinct = false;
    for (size_t i = 1; i < m->dev.size(); ++i) distinct = distinct || m->dev[i] != m->dev[0];
    if (!distinct) return "local";
    s = "peer-copy: " + m->transport_note;
    return s.c_str();
}

int needle_multi_sync(needle_multi *m) {
    if (!m) return fail(NEEDLE_ERR_INVALID, "handle is NULL");
    DeviceGuard guard;
    for (size_t i = 0; i < m->dev.size(); ++i) {
        hipError_t e = hipSetDevice(m->dev[i]);
        if (e == hipSuccess) e = hipStreamSynchronize(m->stream[i]);
        if (e != hipSuccess) return fail(NEEDLE_ERR_DEVICE, std::string("stream synchronize: ") + hipGetErrorString(e));
    }
    return NEEDLE_OK;
}

// op: 0 matches, 1 containedIn, 2 find.  shards[g] is resident on devices[g]; every shard but the last holds a multiple
// of 64 rows (whole bitmap words).  Results land in the root device's buffers, rows in shard order.
int needle_multi_scan(needle_multi *m, const needle_pattern *p, int op, const needle_batch_view *shards, uint64_t *d_bitmap,
                      int32_t *d_start, int32_t *d_end) {
    if (!m || !p || !shards) return fail(NEEDLE_ERR_INVALID, "NULL argument");
    if (m->rank >= 0) return fail(NEEDLE_ERR_INVALID, "a per-rank handle gathers with needle_multi_all_gather_u64 / needle_multi_gather_i32");
    if (op < 0 || op > 2) return fail(NEEDLE_ERR_INVALID, "op: 0 matches, 1 containedIn, 2 find");
    const int n = (int)m->dev.size();
    const bool find = op == 2;
    if (!d_bitmap || (find && (!d_start || !d_end))) return fail(NEEDLE_ERR_INVALID, "output buffer is NULL");
    std::vector<uint64_t> row0(n + 1, 0);
    for (int g = 0; g < n; ++g) {
        if (shards[g].n_rows && row0[g] % 64 != 0)
            return fail(NEEDLE_ERR_INVALID, "every shard but the last non-empty one must hold a multiple of 64 rows (whole bitmap words)");
        row0[g + 1] = row0[g] + shards[g].n_rows;
    }
    DeviceGuard guard;
    auto scan = [&](int g, uint64_t *bm, int32_t *st, int32_t *en) -> int {
        if (shards[g].n_rows == 0) return NEEDLE_OK;
        switch (op) {
        case 0: return needle_matches_dev(p, &shards[g], bm, m->stream[g]);
        case 1: return needle_contained_in_dev(p, &shards[g], bm, m->stream[g]);
        default: return needle_find_dev(p, &shards[g], bm, st, en, m->stream[g]);
        }
    };
    struct Part { // where shard g's results sit before the gather (g > 0, or g == 0 in loopback mode)
        uint64_t *bm;
        int32_t *st, *en;
        uint32_t *packed; // find, shard on another device than the root: start / end as one dword per row for the link
        uint64_t stage_off; // ... and where they land in the root's staging buffer (dwords)
    };
    std::vector<Part> part(n, Part{nullptr, nullptr, nullptr, nullptr, 0});
    // find(): start / end of a shard on ANOTHER device cross the link as one dword per row (two 16-bit halves, 0xFFFF = no
    // match: needle_pack_start_end16_dev) and are unpacked on the root -- half the bytes every peer pushes to the root.
    // Exact for rows of at most 65 534 chars; longer rows keep the two int32 arrays.  NEEDLE_MULTI_PACK16=0 turns it off,
    // =2 also packs shards that share the root's device (tests on a one-GPU box).
    static const int pack_env = getenv("NEEDLE_MULTI_PACK16") ? atoi(getenv("NEEDLE_MULTI_PACK16")) : 1;
    bool can_pack = find && pack_env != 0;
    for (int g = 0; g < n; ++g) can_pack = can_pack && (shards[g].n_rows == 0 || shards[g].row_stride <= 65534);
    // (such a shard's scan stores the dword form itself -- needle_find_packed16_dev -- into the send buffer: no int32 arrays, no pack pass)
    uint64_t stage_rows = 0;
    for (int g = 0; g < n; ++g) {
        hipError_t e = hipSetDevice(m->dev[g]);
        if (e != hipSuccess) return fail(NEEDLE_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
        const uint64_t rows = shards[g].n_rows, words = (rows + 63) / 64;
        const bool in_place = g == 0 && !m->loopback; // the root's own shard is the first block of the result
        if (in_place) {
            int rc = scan(g, d_bitmap, d_start, d_end);
            if (rc) return fail(rc, needle_last_error());
            continue;
        }
        const bool pack = can_pack && rows && (m->dev[g] != m->dev[0] || pack_env == 2);
        const size_t o_st = (size_t)words * 8, o_en = o_st + (find && !pack ? (size_t)rows * 4 : 0), o_pk = o_en + (find && !pack ? (size_t)rows * 4 : 0),
                     total = o_pk + (pack ? (size_t)rows * 4 : 0);
        needle_multi::Buf &b = m->local[g];
        if (b.cap < total) {
            if (b.p) {
                (void)hipStreamSynchronize(m->stream[g]);
                (void)hipFree(b.p);
            }
            b.p = nullptr;
            b.cap = 0;
            size_t want = total + total / 4 + 256;
            if ((e = hipMalloc((void **)&b.p, want)) != hipSuccess) return fail(NEEDLE_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
            b.cap = want;
        }
        part[g] = Part{(uint64_t *)b.p, (int32_t *)(b.p + o_st), (int32_t *)(b.p + o_en), pack ? (uint32_t *)(b.p + o_pk) : nullptr, stage_rows};
        int rc = pack ? needle_find_packed16_dev(p, &shards[g], part[g].bm, part[g].packed, m->stream[g]) : scan(g, part[g].bm, part[g].st, part[g].en);
        if (rc) return fail(rc, needle_last_error());
        if (pack) stage_rows += rows;
    }
    if (stage_rows) { // the root's staging buffer for the packed halves
        hipError_t e = hipSetDevice(m->dev[0]);
        needle_multi::Buf &b = m->root_stage;
        if (e == hipSuccess && b.cap < stage_rows * 4) {
            if (b.p) {
                (void)hipStreamSynchronize(m->stream[0]);
                (void)hipFree(b.p);
            }
            b.p = nullptr;
            b.cap = 0;
            const size_t want = (size_t)stage_rows * 4 + (size_t)stage_rows + 256;
            if ((e = hipMalloc((void **)&b.p, want)) == hipSuccess) b.cap = want;
        }
        if (e != hipSuccess) return fail(NEEDLE_ERR_DEVICE, std::string("root staging buffer: ") + hipGetErrorString(e));
    }
    uint32_t *const stage = (uint32_t *)m->root_stage.p;
    auto unpack_on_root = [&]() -> int { // after the root's stream has the peers' halves: back to the ABI's two int32 arrays
        (void)hipSetDevice(m->dev[0]);
        for (int g = 0; g < n; ++g)
            if (part[g].packed) {
                const int rc = needle_unpack_start_end16_dev(stage + part[g].stage_off, shards[g].n_rows, d_start + row0[g], d_end + row0[g], m->stream[0]);
                if (rc) return rc;
            }
        if (stage_rows) {
            if (!m->stage_free && hipEventCreateWithFlags(&m->stage_free, hipEventDisableTiming) != hipSuccess) return fail(NEEDLE_ERR_DEVICE, "hipEventCreate");
            if (hipEventRecord(m->stage_free, m->stream[0]) != hipSuccess) return fail(NEEDLE_ERR_DEVICE, "hipEventRecord");
        }
        return NEEDLE_OK;
    };
    // ---- gather to the root
    if (m->rccl) {
        int rc = g_rccl.GroupStart();
        for (int g = 0; g < n && rc == 0; ++g) {
            if (!part[g].bm || shards[g].n_rows == 0) continue;
            const uint64_t rows = shards[g].n_rows, words = (rows + 63) / 64;
            (void)hipSetDevice(m->dev[g]);
            const bool pk = part[g].packed != nullptr;
            rc = g_rccl.Send(part[g].bm, words, kNcclUint64, 0, m->comm[g], m->stream[g]);
            if (rc == 0 && find && pk) rc = g_rccl.Send(part[g].packed, rows, kNcclInt32, 0, m->comm[g], m->stream[g]);
            if (rc == 0 && find && !pk) rc = g_rccl.Send(part[g].st, rows, kNcclInt32, 0, m->comm[g], m->stream[g]);
            if (rc == 0 && find && !pk) rc = g_rccl.Send(part[g].en, rows, kNcclInt32, 0, m->comm[g], m->stream[g]);
            (void)hipSetDevice(m->dev[0]);
            if (rc == 0) rc = g_rccl.Recv(d_bitmap + row0[g] / 64, words, kNcclUint64, g, m->comm[0], m->stream[0]);
            if (rc == 0 && find && pk) rc = g_rccl.Recv(stage + part[g].stage_off, rows, kNcclInt32, g, m->comm[0], m->stream[0]);
            if (rc == 0 && find && !pk) rc = g_rccl.Recv(d_start + row0[g], rows, kNcclInt32, g, m->comm[0], m->stream[0]);
            if (rc == 0 && find && !pk) rc = g_rccl.Recv(d_end + row0[g], rows, kNcclInt32, g, m->comm[0], m->stream[0]);
        }
        const int rc2 = g_rccl.GroupEnd();
        if (rc == 0) rc = rc2;
        if (rc != 0) return fail(NEEDLE_ERR_DEVICE, std::string("RCCL gather: ") + g_rccl.GetErrorString(rc));
        if (const int urc = unpack_on_root()) return fail(urc, needle_last_error()); // (queued behind the receives on the root's stream)
    } else {
        for (int g = 0; g < n; ++g) {
            if (!part[g].bm || shards[g].n_rows == 0) continue;
            const uint64_t rows = shards[g].n_rows, words = (rows + 63) / 64;
            hipError_t e = hipSetDevice(m->dev[g]);
            // every peer pushes its results to the root on its own stream: device-to-device copies when the shard shares
            // the root's device, peer-to-peer copies (its own xGMI link to the root) otherwise
            const int sd = m->dev[g], rd = m->dev[0];
            auto push = [&](void *dst, const void *src, size_t bytes) {
                return sd == rd ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, m->stream[g])
                                : hipMemcpyPeerAsync(dst, rd, src, sd, bytes, m->stream[g]);
            };
            const bool pk = part[g].packed != nullptr;
            if (e == hipSuccess && pk && m->stage_free) e = hipStreamWaitEvent(m->stream[g], m->stage_free, 0); // the last call's unpack has read the staging buffer
            if (e == hipSuccess) e = push(d_bitmap + row0[g] / 64, part[g].bm, words * 8);
            if (e == hipSuccess && find && pk) e = push(stage + part[g].stage_off, part[g].packed, rows * 4);
            if (e == hipSuccess && find && !pk) e = push(d_start + row0[g], part[g].st, rows * 4);
            if (e == hipSuccess && find && !pk) e = push(d_end + row0[g], part[g].en, rows * 4);
            if (e == hipSuccess) e = hipEventRecord(m->done[g], m->stream[g]);
            if (e == hipSuccess && g != 0) {
                (void)hipSetDevice(m->dev[0]);
                e = hipStreamWaitEvent(m->stream[0], m->done[g], 0); // the root's stream is the one to wait on
            }
            if (e != hipSuccess) return fail(NEEDLE_ERR_DEVICE, std::string("gather copy: ") + hipGetErrorString(e));
        }
        if (const int urc = unpack_on_root()) return fail(urc, needle_last_error()); // (the root's stream has waited for every push)
    }
    return NEEDLE_OK;
}

// A host batch sharded over the devices: contiguous row blocks on 64-row boundaries, upload, scan, gather to the root,
// download (PCIe-inclusive; the Java host's one-call form, GpuPattern.*Batch with several devices).
int needle_scan_host_multi(needle_multi *m, const needle_pattern *p, int op, const needle_batch_view *rows, uint64_t *bitmap,
                           int32_t *start, int32_t *end) {
    if (!m || !p || !rows) return fail(NEEDLE_ERR_INVALID, "NULL argument");
    if (rows->char_width != 1 && rows->char_width != 2) return fail(NEEDLE_ERR_INVALID, "char_width must be 1 or 2");
    if (rows->n_rows == 0) return NEEDLE_OK;
    if (!rows->rows || !bitmap || (op == 2 && (!start || !end))) return fail(NEEDLE_ERR_INVALID, "buffer is NULL");
    if (rows->row_len > rows->row_stride) return fail(NEEDLE_ERR_INVALID, "row_len > row_stride");
    if (rows->lengths)
        for (uint64_t r = 0; r < rows->n_rows; ++r)
            if (rows->lengths[r] > rows->row_stride) return fail(NEEDLE_ERR_INVALID, "lengths[r] > row_stride");
    const int n = (int)m->dev.size();
    const uint64_t total = rows->n_rows, cw = rows->char_width;
    uint64_t per = (total + n - 1) / n;
    per = (per + 63) / 64 * 64;
    const uint64_t src_stride = rows->row_stride * cw;
    uint64_t dst_stride = (src_stride + 15) & ~(uint64_t)15;
    if (dst_stride == 0) dst_stride = 16;
    DeviceGuard guard;
    std::vector<needle_batch_view> shards(n);
    std::vector<void *> d_rows(n, nullptr), d_len(n, nullptr);
    void *d_bm = nullptr, *d_st = nullptr, *d_en = nullptr;
    auto cleanup = [&]() {
        (void)needle_multi_sync(m);
        for (int g = 0; g < n; ++g) {
            (void)hipSetDevice(m->dev[g]);
            if (d_rows[g]) (void)hipFree(d_rows[g]);
            if (d_len[g]) (void)hipFree(d_len[g]);
        }
        (void)hipSetDevice(m->dev[0]);
        for (void *q : {d_bm, d_st, d_en})
            if (q) (void)hipFree(q);
    };
    auto bail = [&](int code, const std::string &msg) {
        cleanup();
        return fail(code, msg);
    };
    for (int g = 0; g < n; ++g) {
        const uint64_t r0 = std::min<uint64_t>((uint64_t)g * per, total), cnt = std::min<uint64_t>(per, total - r0);
        needle_batch_view &v = shards[g];
        v = *rows;
        v.n_rows = cnt;
        v.row_stride = dst_stride / cw;
        v.rows = nullptr;
        v.lengths = nullptr;
        if (cnt == 0) continue;
        hipError_t e = hipSetDevice(m->dev[g]);
        if (e == hipSuccess) e = hipMalloc(&d_rows[g], cnt * dst_stride);
        const uint8_t *src = (const uint8_t *)rows->rows + r0 * src_stride;
        if (e == hipSuccess) {
            if (dst_stride == src_stride) e = hipMemcpyAsync(d_rows[g], src, cnt * src_stride, hipMemcpyHostToDevice, m->stream[g]);
            else {
                e = hipMemsetAsync(d_rows[g], 0, cnt * dst_stride, m->stream[g]);
                if (e == hipSuccess && src_stride)
                    e = hipMemcpy2DAsync(d_rows[g], dst_stride, src, src_stride, src_stride, cnt, hipMemcpyHostToDevice, m->stream[g]);
            }
        }
        if (e == hipSuccess && rows->lengths) {
            e = hipMalloc(&d_len[g], cnt * 4);
            if (e == hipSuccess) e = hipMemcpyAsync(d_len[g], rows->lengths + r0, cnt * 4, hipMemcpyHostToDevice, m->stream[g]);
        }
        if (e != hipSuccess) return bail(NEEDLE_ERR_DEVICE, std::string("shard upload: ") + hipGetErrorString(e));
        v.rows = d_rows[g];
        v.lengths = (const uint32_t *)d_len[g];
    }
    const uint64_t words = (total + 63) / 64;
    hipError_t e = hipSetDevice(m->dev[0]);
    if (e == hipSuccess) e = hipMalloc(&d_bm, words * 8);
    if (e == hipSuccess && op == 2) e = hipMalloc(&d_st, total * 4);
    if (e == hipSuccess && op == 2) e = hipMalloc(&d_en, total * 4);
    if (e != hipSuccess) return bail(NEEDLE_ERR_DEVICE, std::string("result buffers: ") + hipGetErrorString(e));
    int rc = needle_multi_scan(m, p, op, shards.data(), (uint64_t *)d_bm, (int32_t *)d_st, (int32_t *)d_en);
    if (rc == NEEDLE_OK) rc = needle_multi_sync(m);
    if (rc) {
        const std::string msg = needle_last_error();
        return bail(rc, msg);
    }
    (void)hipSetDevice(m->dev[0]);
    e = hipMemcpy(bitmap, d_bm, words * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && op == 2) e = hipMemcpy(start, d_st, total * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && op == 2) e = hipMemcpy(end, d_en, total * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return bail(NEEDLE_ERR_DEVICE, std::string("download: ") + hipGetErrorString(e));
    cleanup();
    return NEEDLE_OK;
}

} // extern "C"
