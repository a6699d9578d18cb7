// Multi-GPU behind the C ABI: one host process drives the GPUs of a node (the reference is single-GPU,
// EventDrivenMap.cu:80-94, Driver.cu:20).  A group holds one context (device + stream of its own) per shard.
//   * table interpolation: contiguous query shards, the table replicated on every device, no exchange unless the
//     caller wants every device to hold the whole result vector -- then an RCCL all-gather over xGMI
//     (ncclAllGather on the per-device communicators of ncclCommInitAll, one ncclGroupStart/End section);
//   * EventDrivenMap::ComputeF: realisation shards (the realisation is already the unit of work, EventDrivenMap.cu:196),
//     exchange = the partial block of MI_EDM_PARTIAL_LEN(S) doubles per shard.  In one process the blocks are already
//     on the host (pinned D2H at the end of each shard's pipeline), so the default adds them there in shard order;
//     MI_GROUP_REDUCE_RCCL does the same sum with ncclAllReduce on the devices (what one-process-per-GPU callers do
//     with torch.distributed, sharding.py).
// RCCL is bound at run time (dlopen): the library must not drag a second RCCL into a process that already carries
// PyTorch's.  A group may name the same device several times (rehearsal of the shard arithmetic on a box with fewer
// GPUs than shards): RCCL cannot form a communicator over duplicates, so such a group gathers with device-to-device
// copies and reduces on the host.
#include <dlfcn.h>

#include <new>
#include <set>
#include <vector>

#include "mi_common.hpp"

// the few RCCL declarations used (rccl.h:236,260,339,611,678; kept local so that the header is not a build dependency)
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
enum { mi_ncclFloat64 = 8, mi_ncclSum = 0 };

struct mi_group {
    std::vector<int> dev;
    std::vector<mi_ctx*> ctx;
    std::vector<hipEvent_t> done;      // per shard: its kernels of the current call have been enqueued up to here
    bool distinct = true;
    int reduce_mode = 0;
    // RCCL, bound lazily
    void* lib = nullptr;
    bool rccl_tried = false;
    std::vector<ncclComm_t> comms;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    std::vector<double*> red_dev;      // MI_EDM partial blocks on the devices (RCCL reduce mode)
    // chunked gather (mi_group_set_gather_chunks): a second stream per member for the exchange, one event per chunk
    int gather_chunks = 1;
    std::vector<hipStream_t> gstream;
    std::vector<std::vector<hipEvent_t>> cdone;   // [member][chunk]: the chunk's kernel has been enqueued up to here
    std::vector<hipEvent_t> gdone;                // [member]: its exchange stream has been enqueued up to here
};

struct mi_group_grid1 {
    mi_group* g;
    std::vector<mi_grid1*> grid;
};

struct mi_group_grid2 {
    mi_group* g;
    std::vector<mi_grid2*> grid;
};

struct mi_group_edm {
    mi_group* g;
    mi_edm_params total;
    std::vector<mi_edm*> shard;
    std::vector<size_t> lo, hi;
};

namespace {

constexpr int kMaxSpikesG = 8;

mi_status bind_rccl(mi_group* g)
{
    if (g->comms.size() == g->dev.size()) return MI_OK;
    if (!g->distinct)
        return mi::fail(nullptr, MI_ERR_INVALID_ARG, "mi_group: RCCL needs distinct devices (this group repeats one)");
    if (!g->rccl_tried) {
        g->rccl_tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            g->lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (g->lib) break;
        }
        if (g->lib) {
            g->CommInitAll = (decltype(g->CommInitAll))dlsym(g->lib, "ncclCommInitAll");
            g->CommDestroy = (decltype(g->CommDestroy))dlsym(g->lib, "ncclCommDestroy");
            g->GetErrorString = (decltype(g->GetErrorString))dlsym(g->lib, "ncclGetErrorString");
            g->AllReduce = (decltype(g->AllReduce))dlsym(g->lib, "ncclAllReduce");
            g->AllGather = (decltype(g->AllGather))dlsym(g->lib, "ncclAllGather");
            g->GroupStart = (decltype(g->GroupStart))dlsym(g->lib, "ncclGroupStart");
            g->GroupEnd = (decltype(g->GroupEnd))dlsym(g->lib, "ncclGroupEnd");
            g->CommCount = (decltype(g->CommCount))dlsym(g->lib, "ncclCommCount");
            g->Broadcast = (decltype(g->Broadcast))dlsym(g->lib, "ncclBroadcast");
        }
    }
    if (!g->lib || !g->CommInitAll || !g->CommDestroy || !g->AllReduce || !g->AllGather || !g->GroupStart || !g->GroupEnd)
        return mi::fail(nullptr, MI_ERR_HIP, "mi_group: librccl.so could not be loaded (%s)", g->lib ? "symbol missing" : dlerror());
    g->comms.assign(g->dev.size(), nullptr);
    const ncclResult_t r = g->CommInitAll(g->comms.data(), (int)g->dev.size(), g->dev.data());
    if (r != 0) {
        g->comms.clear();
        return mi::fail(nullptr, MI_ERR_HIP, "ncclCommInitAll over %zu devices failed: %s", g->dev.size(),
                        g->GetErrorString ? g->GetErrorString(r) : "?");
    }
    return MI_OK;
}

#define MI_NCCL(g, call)                                                                              \
    do {                                                                                              \
        const ncclResult_t r_ = (call);                                                               \
        if (r_ != 0)                                                                                  \
            return mi::fail(nullptr, MI_ERR_HIP, "%s failed: %s", #call,                              \
                            (g)->GetErrorString ? (g)->GetErrorString(r_) : "?");                     \
    } while (0)

}  // namespace

static mi_status gather_shards(mi_group* g, double* const* part_dev, double* const* gathered_dev, size_t n_per_shard);
static mi_status interp1_chunked_gather(mi_group* g, const mi_group_grid1* t, const double* const* xq_dev, double* const* yq_dev,
                                        size_t nq_per_shard, double extrap, double* const* gathered_dev);

extern "C" {

void mi_shard_bounds(size_t n, int rank, int world, size_t* lo, size_t* hi)
{
    // contiguous, balanced: the first n % world shards get one extra unit (= sharding.shard_bounds)
    const size_t w = world > 0 ? (size_t)world : 1, r = rank > 0 ? (size_t)rank : 0;
    const size_t base = n / w, rem = n % w;
    const size_t a = r * base + (r < rem ? r : rem);
    if (lo) *lo = a;
    if (hi) *hi = a + base + (r < rem ? 1 : 0);
}

mi_status mi_group_create(int ndev, const int* devices, mi_group** out)
{
    MI_REQUIRE(nullptr, out != nullptr, "mi_group_create: out is NULL");
    *out = nullptr;
    MI_REQUIRE(nullptr, ndev >= 1 && ndev <= 64, "mi_group_create: ndev=%d not in [1,64]", ndev);
    mi_group* g = new (std::nothrow) mi_group();
    if (!g) return mi::fail(nullptr, MI_ERR_NOMEM, "mi_group_create: out of host memory");
    std::set<int> seen;
    for (int r = 0; r < ndev; ++r) {
        const int d = devices ? devices[r] : r;
        g->dev.push_back(d);
        if (!seen.insert(d).second) g->distinct = false;
    }
    for (int r = 0; r < ndev; ++r) {
        mi_ctx* c = nullptr;
        mi_status st = mi_ctx_create(g->dev[r], &c);
        if (st == MI_OK) st = mi_ctx_own_stream(c);
        hipEvent_t ev = nullptr;
        if (st == MI_OK && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
            st = mi::fail(nullptr, MI_ERR_HIP, "mi_group_create: hipEventCreate failed on device %d", g->dev[r]);
        if (st != MI_OK) {
            if (c) mi_ctx_destroy(c);
            mi_group_destroy(g);
            return st;
        }
        // a group member works on the queries it is handed, in whatever order: no probe traffic between devices' calls
        g->ctx.push_back(c);
        g->done.push_back(ev);
    }
    *out = g;
    return MI_OK;
}

mi_status mi_group_destroy(mi_group* g)
{
    if (!g) return MI_OK;
    // Every stream of every member drained BEFORE any communicator goes: the chunked gather leaves its broadcasts on the
    // members' second streams, and a communicator must not be torn down under work that still refers to it.
    for (size_t r = 0; r < g->ctx.size(); ++r) {
        (void)hipSetDevice(g->dev[r]);
        (void)hipStreamSynchronize(g->ctx[r]->stream);
        if (r < g->gstream.size() && g->gstream[r]) (void)hipStreamSynchronize(g->gstream[r]);
    }
    for (size_t r = 0; r < g->ctx.size(); ++r) {
        (void)hipSetDevice(g->dev[r]);
        if (r < g->comms.size() && g->comms[r] && g->CommDestroy) (void)g->CommDestroy(g->comms[r]);
    }
    for (size_t r = 0; r < g->ctx.size(); ++r) {
        (void)hipSetDevice(g->dev[r]);
        if (r < g->red_dev.size() && g->red_dev[r]) (void)hipFree(g->red_dev[r]);
        if (g->done[r]) (void)hipEventDestroy(g->done[r]);
        if (r < g->gstream.size() && g->gstream[r]) (void)hipStreamDestroy(g->gstream[r]);
        if (r < g->gdone.size() && g->gdone[r]) (void)hipEventDestroy(g->gdone[r]);
        if (r < g->cdone.size())
            for (hipEvent_t e : g->cdone[r])
                if (e) (void)hipEventDestroy(e);
        mi_ctx_destroy(g->ctx[r]);
    }
    // the RCCL handle stays open for the life of the process (communicator teardown may still use it)
    delete g;
    return MI_OK;
}

int mi_group_size(const mi_group* g) { return g ? (int)g->ctx.size() : 0; }

mi_ctx* mi_group_ctx(mi_group* g, int rank)
{
    if (!g || rank < 0 || rank >= (int)g->ctx.size()) return nullptr;
    return g->ctx[rank];
}

int mi_group_rccl_ranks(mi_group* g)
{
    // the size RCCL itself reports for the group's communicator (ncclCommCount of shard 0's handle), forming the
    // communicator first if no collective has needed it yet; 0 when the group cannot use RCCL (repeated devices,
    // library missing) -- the error text is then in mi_last_error(NULL)
    if (!g || bind_rccl(g) != MI_OK || g->comms.empty() || !g->CommCount) return 0;
    int n = 0;
    if (g->CommCount(g->comms[0], &n) != 0) return 0;
    return n;
}

mi_status mi_group_set_gather_chunks(mi_group* g, int chunks)
{
    MI_REQUIRE(nullptr, g != nullptr, "mi_group_set_gather_chunks: group is NULL");
    MI_REQUIRE(nullptr, chunks >= 1 && chunks <= 64, "mi_group_set_gather_chunks: chunks=%d not in [1,64]", chunks);
    g->gather_chunks = chunks;
    return MI_OK;
}

mi_status mi_group_set_reduce(mi_group* g, int mode)
{
    MI_REQUIRE(nullptr, g != nullptr, "mi_group_set_reduce: group is NULL");
    MI_REQUIRE(nullptr, mode == MI_GROUP_REDUCE_HOST || mode == MI_GROUP_REDUCE_RCCL, "mi_group_set_reduce: unknown mode %d", mode);
    if (mode == MI_GROUP_REDUCE_RCCL) {
        mi_status st = bind_rccl(g);
        if (st != MI_OK) return st;
    }
    g->reduce_mode = mode;
    return MI_OK;
}

mi_status mi_group_synchronize(mi_group* g)
{
    MI_REQUIRE(nullptr, g != nullptr, "mi_group_synchronize: group is NULL");
    for (size_t r = 0; r < g->ctx.size(); ++r) {
        mi_status st = mi_ctx_synchronize(g->ctx[r]);
        if (st != MI_OK) return st;
    }
    return MI_OK;
}

// ---- table interpolation ------------------------------------------------------------------------------------------

mi_status mi_group_grid1_create(mi_group* g, const double* x, const double* y, size_t n, unsigned flags, mi_group_grid1** out)
{
    MI_REQUIRE(nullptr, g && out, "mi_group_grid1_create: NULL argument");
    *out = nullptr;
    MI_REQUIRE(nullptr, (flags & MI_GRID_DEVICE_PTRS) == 0, "mi_group_grid1_create: the table is given in host memory (it is replicated on every device)");
    mi_group_grid1* t = new (std::nothrow) mi_group_grid1();
    if (!t) return mi::fail(nullptr, MI_ERR_NOMEM, "mi_group_grid1_create: out of host memory");
    t->g = g;
    for (size_t r = 0; r < g->ctx.size(); ++r) {
        mi_grid1* gr = nullptr;
        const mi_status st = mi_grid1_create(g->ctx[r], x, y, n, flags, &gr);
        if (st != MI_OK) {
            mi_group_grid1_destroy(t);
            return st;
        }
        t->grid.push_back(gr);
    }
    *out = t;
    return MI_OK;
}

mi_status mi_group_grid1_destroy(mi_group_grid1* t)
{
    if (!t) return MI_OK;
    for (mi_grid1* gr : t->grid) mi_grid1_destroy(gr);
    delete t;
    return MI_OK;
}

mi_status mi_group_interp1_f64_dev(mi_group* g, const mi_group_grid1* t, const double* const* xq_dev, double* const* yq_dev,
                                   size_t nq_per_shard, double extrap, double* const* gathered_dev)
{
    MI_REQUIRE(nullptr, g && t && xq_dev && yq_dev, "mi_group_interp1_f64_dev: NULL argument");
    MI_REQUIRE(nullptr, t->g == g && t->grid.size() == g->ctx.size(), "mi_group_interp1_f64_dev: the table belongs to another group");
    const size_t P = g->ctx.size();
    if (gathered_dev && g->distinct) {
        const mi_status st = bind_rccl(g);
        if (st != MI_OK) return st;
    }
    if (gathered_dev && g->gather_chunks > 1 && nq_per_shard >= 2 * (size_t)g->gather_chunks)
        return interp1_chunked_gather(g, t, xq_dev, yq_dev, nq_per_shard, extrap, gathered_dev);
    for (size_t r = 0; r < P; ++r) {
        const mi_status st = mi_interp1_f64_dev(g->ctx[r], t->grid[r], xq_dev[r], yq_dev[r], nq_per_shard, extrap);
        if (st != MI_OK) return st;
        if (gathered_dev && !g->distinct) MI_HIP(g->ctx[r], hipEventRecord(g->done[r], g->ctx[r]->stream));
    }
    if (!gathered_dev || nq_per_shard == 0) return MI_OK;
    // every device receives the P shards, shard s at offset s * nq_per_shard (in place when yq_dev[r] already points
    // there): one all-gather over xGMI, enqueued behind each device's kernel on its own stream -- or, in a rehearsal
    // group (a device named more than once), the same data movement with device-to-device copies
    return gather_shards(g, yq_dev, gathered_dev, nq_per_shard);
}

mi_status mi_group_interp1_f64_host(mi_group* g, const mi_group_grid1* t, const double* xq, double* yq, size_t nq, double extrap)
{
    MI_REQUIRE(nullptr, g && t, "mi_group_interp1_f64_host: NULL argument");
    MI_REQUIRE(nullptr, t->g == g && t->grid.size() == g->ctx.size(), "mi_group_interp1_f64_host: the table belongs to another group");
    if (nq == 0) return MI_OK;
    MI_REQUIRE(nullptr, xq && yq, "mi_group_interp1_f64_host: NULL query/result pointer");
    const int P = (int)g->ctx.size();
    // every shard: upload, kernel, download on its device's own stream; nothing waits until all are enqueued
    const bool pin_in = mi::pin_host(xq, nq * sizeof(double)), pin_out = mi::pin_host(yq, nq * sizeof(double));
    mi_status st = MI_OK;
    hipError_t herr = hipSuccess;
    for (int r = 0; r < P && st == MI_OK && herr == hipSuccess; ++r) {
        size_t lo, hi;
        mi_shard_bounds(nq, r, P, &lo, &hi);
        if (hi == lo) continue;
        mi_ctx* c = g->ctx[r];
        herr = hipSetDevice(g->dev[r]);
        if (herr != hipSuccess) break;
        const size_t bytes = (hi - lo) * sizeof(double);
        st = mi::ensure_scratch(c, 0, bytes);
        if (st == MI_OK) st = mi::ensure_scratch(c, 1, bytes);
        if (st != MI_OK) break;
        herr = hipMemcpyAsync(c->scratch[0], xq + lo, bytes, hipMemcpyHostToDevice, c->stream);
        if (herr != hipSuccess) break;
        st = mi_interp1_f64_dev(c, t->grid[r], (const double*)c->scratch[0], (double*)c->scratch[1], hi - lo, extrap);
        if (st != MI_OK) break;
        herr = hipMemcpyAsync(yq + lo, c->scratch[1], bytes, hipMemcpyDeviceToHost, c->stream);
    }
    hipError_t esync = hipSuccess;
    for (int r = 0; r < P; ++r) {   // drain every stream before the ranges are released, on success or error
        (void)hipSetDevice(g->dev[r]);
        const hipError_t e = hipStreamSynchronize(g->ctx[r]->stream);
        if (e != hipSuccess && esync == hipSuccess) esync = e;
    }
    if (pin_in) mi::unpin_host(xq);
    if (pin_out) mi::unpin_host(yq);
    if (st != MI_OK) return st;
    if (herr != hipSuccess) return mi::fail(nullptr, MI_ERR_HIP, "mi_group_interp1_f64_host: copy failed: %s", hipGetErrorString(herr));
    if (esync != hipSuccess) return mi::fail(nullptr, MI_ERR_HIP, "mi_group_interp1_f64_host: %s", hipGetErrorString(esync));
    return MI_OK;
}

// ---- 2-D table interpolation (BASELINE config 3 over several GPUs: scattered query shards, the table replicated) ------

mi_status mi_group_grid2_create(mi_group* g, const double* x, size_t nx, const double* y, size_t ny, const double* z,
                                unsigned flags, mi_group_grid2** out)
{
    MI_REQUIRE(nullptr, g && out, "mi_group_grid2_create: NULL argument");
    *out = nullptr;
    MI_REQUIRE(nullptr, (flags & MI_GRID_DEVICE_PTRS) == 0, "mi_group_grid2_create: the table is given in host memory (it is replicated on every device)");
    mi_group_grid2* t = new (std::nothrow) mi_group_grid2();
    if (!t) return mi::fail(nullptr, MI_ERR_NOMEM, "mi_group_grid2_create: out of host memory");
    t->g = g;
    for (size_t r = 0; r < g->ctx.size(); ++r) {
        mi_grid2* gr = nullptr;
        const mi_status st = mi_grid2_create(g->ctx[r], x, nx, y, ny, z, flags, &gr);
        if (st != MI_OK) {
            mi_group_grid2_destroy(t);
            return st;
        }
        t->grid.push_back(gr);
    }
    *out = t;
    return MI_OK;
}

mi_status mi_group_grid2_destroy(mi_group_grid2* t)
{
    if (!t) return MI_OK;
    for (mi_grid2* gr : t->grid) mi_grid2_destroy(gr);
    delete t;
    return MI_OK;
}

// results of every shard to every device: RCCL all-gather (distinct devices) or device-to-device copies (rehearsal group)
static mi_status gather_shards(mi_group* g, double* const* part_dev, double* const* gathered_dev, size_t n_per_shard)
{
    const size_t P = g->ctx.size();
    if (g->distinct) {
        MI_NCCL(g, g->GroupStart());
        for (size_t r = 0; r < P; ++r) {
            const ncclResult_t rc = g->AllGather(part_dev[r], gathered_dev[r], n_per_shard, mi_ncclFloat64, g->comms[r], g->ctx[r]->stream);
            if (rc != 0) {
                (void)g->GroupEnd();
                return mi::fail(nullptr, MI_ERR_HIP, "ncclAllGather failed: %s", g->GetErrorString ? g->GetErrorString(rc) : "?");
            }
        }
        MI_NCCL(g, g->GroupEnd());
        return MI_OK;
    }
    for (size_t r = 0; r < P; ++r) {
        MI_HIP(g->ctx[r], hipSetDevice(g->dev[r]));
        for (size_t s = 0; s < P; ++s) {
            double* dst = gathered_dev[r] + s * n_per_shard;
            if (dst == part_dev[s]) continue;
            MI_HIP(g->ctx[r], hipStreamWaitEvent(g->ctx[r]->stream, g->done[s], 0));
            MI_HIP(g->ctx[r], hipMemcpyAsync(dst, part_dev[s], n_per_shard * sizeof(double), hipMemcpyDeviceToDevice, g->ctx[r]->stream));
        }
    }
    return MI_OK;
}

// The all-gather hidden behind the kernels (north_star: "an RCCL all-gather over xGMI to reassemble the residual vector";
// at 8 GPUs the exchange of a 1e8-query call is 0.65 ms per link against 0.1 ms of kernel): every
// shard is cut into `gather_chunks` contiguous chunks; the kernel of chunk k+1 runs on the member's own stream while chunk k
// travels on the member's exchange stream.  A chunk of every shard has to land at its place INSIDE that shard's slot of the
// gathered vector (shard s at s * nq_per_shard), which ncclAllGather cannot do (it packs the pieces back to back), so a chunk
// is exchanged as P grouped ncclBroadcast calls, root s sending its piece of the chunk (in place where yq_dev already points
// into gathered_dev).  Repeated devices (rehearsal): the same schedule with device-to-device copies.  Results are those of the
// unchunked call bit for bit (tests/test_group_gpu.py); the timing on a real 8-GPU node is unmeasured.
static mi_status interp1_chunked_gather(mi_group* g, const mi_group_grid1* t, const double* const* xq_dev, double* const* yq_dev,
                                        size_t nq_per_shard, double extrap, double* const* gathered_dev)
{
    const size_t P = g->ctx.size();
    const size_t K = (size_t)g->gather_chunks;
    if (g->distinct && !g->Broadcast) return mi::fail(nullptr, MI_ERR_HIP, "mi_group: ncclBroadcast is not exported by the loaded librccl");
    // exchange streams and events, created on first use
    if (g->gstream.size() != P) {
        g->gstream.assign(P, nullptr);
        g->gdone.assign(P, nullptr);
        g->cdone.assign(P, std::vector<hipEvent_t>());
    }
    for (size_t r = 0; r < P; ++r) {
        MI_HIP(g->ctx[r], hipSetDevice(g->dev[r]));
        if (!g->gstream[r]) MI_HIP(g->ctx[r], hipStreamCreateWithFlags(&g->gstream[r], hipStreamNonBlocking));
        if (!g->gdone[r]) MI_HIP(g->ctx[r], hipEventCreateWithFlags(&g->gdone[r], hipEventDisableTiming));
        while (g->cdone[r].size() < K) {
            hipEvent_t e = nullptr;
            MI_HIP(g->ctx[r], hipEventCreateWithFlags(&e, hipEventDisableTiming));
            g->cdone[r].push_back(e);
        }
        // the exchange stream starts behind whatever the member's stream already holds (an earlier call may still read
        // or write the buffers)
        MI_HIP(g->ctx[r], hipEventRecord(g->gdone[r], g->ctx[r]->stream));
        MI_HIP(g->ctx[r], hipStreamWaitEvent(g->gstream[r], g->gdone[r], 0));
    }
    size_t c = (nq_per_shard + K - 1) / K;
    c += c & 1;                                        // even: 16-byte aligned chunk starts for the vector kernels
    for (size_t k = 0; k * c < nq_per_shard; ++k) {
        const size_t off = k * c, len = std::min(c, nq_per_shard - off);
        for (size_t r = 0; r < P; ++r) {
            const mi_status st = mi_interp1_f64_dev(g->ctx[r], t->grid[r], xq_dev[r] + off, yq_dev[r] + off, len, extrap);
            if (st != MI_OK) return st;
            MI_HIP(g->ctx[r], hipEventRecord(g->cdone[r][k], g->ctx[r]->stream));
        }
        if (g->distinct) {
            for (size_t r = 0; r < P; ++r) {
                MI_HIP(g->ctx[r], hipSetDevice(g->dev[r]));
                MI_HIP(g->ctx[r], hipStreamWaitEvent(g->gstream[r], g->cdone[r][k], 0));
            }
            MI_NCCL(g, g->GroupStart());
            for (size_t s = 0; s < P; ++s) {
                for (size_t r = 0; r < P; ++r) {
                    const ncclResult_t rc = g->Broadcast(yq_dev[r] + off, gathered_dev[r] + s * nq_per_shard + off, len, mi_ncclFloat64,
                                                         (int)s, g->comms[r], g->gstream[r]);
                    if (rc != 0) {
                        (void)g->GroupEnd();
                        return mi::fail(nullptr, MI_ERR_HIP, "ncclBroadcast failed: %s", g->GetErrorString ? g->GetErrorString(rc) : "?");
                    }
                }
            }
            MI_NCCL(g, g->GroupEnd());
        } else {
            for (size_t r = 0; r < P; ++r) {
                MI_HIP(g->ctx[r], hipSetDevice(g->dev[r]));
                for (size_t s = 0; s < P; ++s) {
                    double* dst = gathered_dev[r] + s * nq_per_shard + off;
                    if (dst == yq_dev[s] + off) continue;
                    MI_HIP(g->ctx[r], hipStreamWaitEvent(g->gstream[r], g->cdone[s][k], 0));
                    MI_HIP(g->ctx[r], hipMemcpyAsync(dst, yq_dev[s] + off, len * sizeof(double), hipMemcpyDeviceToDevice, g->gstream[r]));
                }
            }
        }
    }
    // the member's own stream ends behind its exchange stream: mi_group_synchronize (and any later call) covers the gather
    for (size_t r = 0; r < P; ++r) {
        MI_HIP(g->ctx[r], hipSetDevice(g->dev[r]));
        MI_HIP(g->ctx[r], hipEventRecord(g->gdone[r], g->gstream[r]));
        MI_HIP(g->ctx[r], hipStreamWaitEvent(g->ctx[r]->stream, g->gdone[r], 0));
    }
    return MI_OK;
}

mi_status mi_group_interp2_f64_dev(mi_group* g, const mi_group_grid2* t, const double* const* xq_dev, const double* const* yq_dev,
                                   double* const* zq_dev, size_t nq_per_shard, double extrap, double* const* gathered_dev)
{
    MI_REQUIRE(nullptr, g && t && xq_dev && yq_dev && zq_dev, "mi_group_interp2_f64_dev: NULL argument");
    MI_REQUIRE(nullptr, t->g == g && t->grid.size() == g->ctx.size(), "mi_group_interp2_f64_dev: the table belongs to another group");
    const size_t P = g->ctx.size();
    if (gathered_dev && g->distinct) {
        const mi_status st = bind_rccl(g);
        if (st != MI_OK) return st;
    }
    for (size_t r = 0; r < P; ++r) {
        const mi_status st = mi_interp2_f64_dev(g->ctx[r], t->grid[r], xq_dev[r], yq_dev[r], zq_dev[r], nq_per_shard, extrap);
        if (st != MI_OK) return st;
        if (gathered_dev && !g->distinct) MI_HIP(g->ctx[r], hipEventRecord(g->done[r], g->ctx[r]->stream));
    }
    if (!gathered_dev || nq_per_shard == 0) return MI_OK;
    return gather_shards(g, zq_dev, gathered_dev, nq_per_shard);
}

mi_status mi_group_interp2_f64_host(mi_group* g, const mi_group_grid2* t, const double* xq, const double* yq, double* zq, size_t nq,
                                    double extrap)
{
    MI_REQUIRE(nullptr, g && t, "mi_group_interp2_f64_host: NULL argument");
    MI_REQUIRE(nullptr, t->g == g && t->grid.size() == g->ctx.size(), "mi_group_interp2_f64_host: the table belongs to another group");
    if (nq == 0) return MI_OK;
    MI_REQUIRE(nullptr, xq && yq && zq, "mi_group_interp2_f64_host: NULL query/result pointer");
    const int P = (int)g->ctx.size();
    const bool pin_x = mi::pin_host(xq, nq * sizeof(double)), pin_y = mi::pin_host(yq, nq * sizeof(double)),
               pin_z = mi::pin_host(zq, nq * sizeof(double));
    mi_status st = MI_OK;
    hipError_t herr = hipSuccess;
    for (int r = 0; r < P && st == MI_OK && herr == hipSuccess; ++r) {
        size_t lo, hi;
        mi_shard_bounds(nq, r, P, &lo, &hi);
        if (hi == lo) continue;
        mi_ctx* c = g->ctx[r];
        herr = hipSetDevice(g->dev[r]);
        if (herr != hipSuccess) break;
        const size_t bytes = (hi - lo) * sizeof(double);
        for (int k = 0; k < 3 && st == MI_OK; ++k) st = mi::ensure_scratch(c, k, bytes);
        if (st != MI_OK) break;
        herr = hipMemcpyAsync(c->scratch[0], xq + lo, bytes, hipMemcpyHostToDevice, c->stream);
        if (herr != hipSuccess) break;
        herr = hipMemcpyAsync(c->scratch[1], yq + lo, bytes, hipMemcpyHostToDevice, c->stream);
        if (herr != hipSuccess) break;
        st = mi_interp2_f64_dev(c, t->grid[r], (const double*)c->scratch[0], (const double*)c->scratch[1], (double*)c->scratch[2],
                                hi - lo, extrap);
        if (st != MI_OK) break;
        herr = hipMemcpyAsync(zq + lo, c->scratch[2], bytes, hipMemcpyDeviceToHost, c->stream);
    }
    hipError_t esync = hipSuccess;
    for (int r = 0; r < P; ++r) {   // drain every stream before the ranges are released, on success or error
        (void)hipSetDevice(g->dev[r]);
        const hipError_t e = hipStreamSynchronize(g->ctx[r]->stream);
        if (e != hipSuccess && esync == hipSuccess) esync = e;
    }
    if (pin_x) mi::unpin_host(xq);
    if (pin_y) mi::unpin_host(yq);
    if (pin_z) mi::unpin_host(zq);
    if (st != MI_OK) return st;
    if (herr != hipSuccess) return mi::fail(nullptr, MI_ERR_HIP, "mi_group_interp2_f64_host: copy failed: %s", hipGetErrorString(herr));
    if (esync != hipSuccess) return mi::fail(nullptr, MI_ERR_HIP, "mi_group_interp2_f64_host: %s", hipGetErrorString(esync));
    return MI_OK;
}

// ---- EventDrivenMap -----------------------------------------------------------------------------------------------

static mi_status edm_shard_params(const mi_group_edm* e, int r, mi_edm_params* p)
{
    *p = e->total;
    p->n_real = (uint32_t)(e->hi[r] - e->lo[r]);
    p->real_offset = e->total.real_offset + (uint32_t)e->lo[r];
    return MI_OK;
}

static mi_status edm_layout(mi_group_edm* e, const mi_edm_params* p)
{
    const int P = (int)e->g->ctx.size();
    MI_REQUIRE(nullptr, p->n_real >= (uint32_t)P, "mi_group_edm: %u realisations cannot be split over %d shards", p->n_real, P);
    e->total = *p;
    e->lo.resize(P);
    e->hi.resize(P);
    for (int r = 0; r < P; ++r) mi_shard_bounds(p->n_real, r, P, &e->lo[r], &e->hi[r]);
    return MI_OK;
}

mi_status mi_group_edm_create(mi_group* g, const mi_edm_params* p, mi_group_edm** out)
{
    MI_REQUIRE(nullptr, g && p && out, "mi_group_edm_create: NULL argument");
    *out = nullptr;
    mi_group_edm* e = new (std::nothrow) mi_group_edm();
    if (!e) return mi::fail(nullptr, MI_ERR_NOMEM, "mi_group_edm_create: out of host memory");
    e->g = g;
    mi_status st = edm_layout(e, p);
    for (size_t r = 0; st == MI_OK && r < g->ctx.size(); ++r) {
        mi_edm_params ps;
        edm_shard_params(e, (int)r, &ps);
        mi_edm* s = nullptr;
        st = mi_edm_create(g->ctx[r], &ps, &s);
        if (st == MI_OK) e->shard.push_back(s);
    }
    if (st != MI_OK) {
        mi_group_edm_destroy(e);
        return st;
    }
    *out = e;
    return MI_OK;
}

mi_status mi_group_edm_destroy(mi_group_edm* e)
{
    if (!e) return MI_OK;
    for (mi_edm* s : e->shard) mi_edm_destroy(s);
    delete e;
    return MI_OK;
}

mi_status mi_group_edm_set_params(mi_group_edm* e, const mi_edm_params* p)
{
    MI_REQUIRE(nullptr, e && p, "mi_group_edm_set_params: NULL argument");
    mi_status st = edm_layout(e, p);
    for (size_t r = 0; st == MI_OK && r < e->shard.size(); ++r) {
        mi_edm_params ps;
        edm_shard_params(e, (int)r, &ps);
        st = mi_edm_set_params(e->shard[r], &ps);
    }
    return st;
}

mi_edm* mi_group_edm_shard(mi_group_edm* e, int rank)
{
    if (!e || rank < 0 || rank >= (int)e->shard.size()) return nullptr;
    return e->shard[rank];
}

mi_status mi_group_edm_shard_bounds(const mi_group_edm* e, int rank, size_t* lo, size_t* hi)
{
    MI_REQUIRE(nullptr, e && rank >= 0 && rank < (int)e->shard.size(), "mi_group_edm_shard_bounds: bad argument");
    if (lo) *lo = e->lo[rank];
    if (hi) *hi = e->hi[rank];
    return MI_OK;
}

mi_status mi_group_edm_compute_f(mi_group_edm* e, const double* z, double* f, double* partial_total)
{
    MI_REQUIRE(nullptr, e && z && f, "mi_group_edm_compute_f: NULL argument");
    mi_group* g = e->g;
    const size_t P = e->shard.size();
    const uint32_t S = e->total.n_spikes;
    const uint32_t L = 2 * S + 1;
    // every shard's whole pipeline is enqueued before any is waited for: the devices run concurrently
    for (size_t r = 0; r < P; ++r) {
        const mi_status st = mi_edm_compute_f_begin(e->shard[r], z);
        if (st != MI_OK) {
            double fr[kMaxSpikesG];
            for (size_t q = 0; q < r; ++q) (void)mi_edm_compute_f_end(e->shard[q], fr, nullptr);   // nothing stays pending
            return st;
        }
    }
    double tot[2 * kMaxSpikesG + 1] = {0};
    std::vector<double> parts(P * L);
    mi_status first = MI_OK;
    for (size_t r = 0; r < P; ++r) {
        double fr[kMaxSpikesG];
        const mi_status st = mi_edm_compute_f_end(e->shard[r], fr, &parts[r * L]);
        if (st != MI_OK && first == MI_OK) first = st;
    }
    if (first != MI_OK) return first;
    if (g->reduce_mode == MI_GROUP_REDUCE_RCCL && P > 1) {
        // the same sum on the devices: all-reduce of 2S+1 doubles over xGMI (what one-process-per-GPU callers do)
        mi_status st = bind_rccl(g);
        if (st != MI_OK) return st;
        if (g->red_dev.size() != P) {
            g->red_dev.assign(P, nullptr);
            for (size_t r = 0; r < P; ++r) {
                MI_HIP(g->ctx[r], hipSetDevice(g->dev[r]));
                MI_HIP(g->ctx[r], hipMalloc(&g->red_dev[r], (2 * kMaxSpikesG + 1) * sizeof(double)));
            }
        }
        for (size_t r = 0; r < P; ++r) {
            MI_HIP(g->ctx[r], hipSetDevice(g->dev[r]));
            MI_HIP(g->ctx[r], hipMemcpyAsync(g->red_dev[r], &parts[r * L], L * sizeof(double), hipMemcpyHostToDevice, g->ctx[r]->stream));
        }
        MI_NCCL(g, g->GroupStart());
        for (size_t r = 0; r < P; ++r) {
            const ncclResult_t rc = g->AllReduce(g->red_dev[r], g->red_dev[r], L, mi_ncclFloat64, mi_ncclSum, g->comms[r], g->ctx[r]->stream);
            if (rc != 0) {
                (void)g->GroupEnd();
                return mi::fail(nullptr, MI_ERR_HIP, "ncclAllReduce failed: %s", g->GetErrorString ? g->GetErrorString(rc) : "?");
            }
        }
        MI_NCCL(g, g->GroupEnd());
        MI_HIP(g->ctx[0], hipSetDevice(g->dev[0]));
        MI_HIP(g->ctx[0], hipMemcpyAsync(tot, g->red_dev[0], L * sizeof(double), hipMemcpyDeviceToHost, g->ctx[0]->stream));
        for (size_t r = 0; r < P; ++r) {
            MI_HIP(g->ctx[r], hipSetDevice(g->dev[r]));
            MI_HIP(g->ctx[r], hipStreamSynchronize(g->ctx[r]->stream));
        }
    } else {
        for (size_t r = 0; r < P; ++r)           // fixed order: the sum is reproducible run to run
            for (uint32_t k = 0; k < L; ++k) tot[k] += parts[r * L + k];
    }
    if (partial_total)
        for (uint32_t k = 0; k < L; ++k) partial_total[k] = tot[k];
    return mi_edm_residual_from_sums(&e->total, z, tot, f);
}

}  // extern "C"
