// Context, error reporting and HIP-event timers of libmi355interp.so.
#include "mi_common.hpp"

#include <map>

namespace mi {

char* tls_error()
{
    static thread_local char buf[512] = {0};
    return buf;
}

mi_status fail(const mi_ctx* ctx, mi_status code, const char* fmt, ...)
{
    char* dst = ctx ? ctx->err : tls_error();
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    if (ctx) {  // keep the thread-local copy in sync so mi_last_error(NULL) also works
        strncpy(tls_error(), dst, 511);
    }
    return code;
}

mi_status ensure_scratch(mi_ctx* ctx, int slot, size_t bytes)
{
    if (ctx->scratch_bytes[slot] >= bytes) return MI_OK;
    if (ctx->scratch[slot]) {
        MI_HIP(ctx, hipFree(ctx->scratch[slot]));
        ctx->scratch[slot] = nullptr;
        ctx->scratch_bytes[slot] = 0;
    }
    size_t want = bytes + (bytes >> 2) + 4096;
    hipError_t e = hipMalloc(&ctx->scratch[slot], want);
    if (e != hipSuccess)
        return fail(ctx, MI_ERR_NOMEM, "hipMalloc(%zu) for scratch failed: %s", want, hipGetErrorString(e));
    ctx->scratch_bytes[slot] = want;
    return MI_OK;
}

mi_status ensure_aux_stream(mi_ctx* ctx)
{
    if (ctx->aux_stream) return MI_OK;
    MI_HIP(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
    MI_HIP(ctx, hipEventCreateWithFlags(&ctx->aux_event, hipEventDisableTiming));
    return MI_OK;
}

namespace {
struct PinEntry { size_t bytes; unsigned refs; };
std::mutex g_pin_mu;
std::map<const void*, PinEntry>& pin_table()
{
    static std::map<const void*, PinEntry> t;
    return t;
}
}  // namespace

bool pin_host(const void* p, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_pin_mu);
    auto& t = pin_table();
    auto it = t.find(p);
    if (it != t.end()) {
        if (it->second.bytes >= bytes) { ++it->second.refs; return true; }
        return false;   // a longer range from the same base while a shorter one is pinned: blocking copies for this call
    }
    // portable: a group call (mi_group_interp1_f64_host) copies from / to the same range on several devices
    const hipError_t e = hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }   // pinned by the caller already, or not pinnable
    t.emplace(p, PinEntry{bytes, 1u});
    return true;
}

void unpin_host(const void* p)
{
    std::lock_guard<std::mutex> lk(g_pin_mu);
    auto& t = pin_table();
    auto it = t.find(p);
    if (it == t.end()) return;
    if (--it->second.refs == 0) {
        (void)hipHostUnregister(const_cast<void*>(p));
        (void)hipGetLastError();   // an unregister that fails must not poison later launch checks
        t.erase(it);
    }
}

size_t pinned_ranges()
{
    std::lock_guard<std::mutex> lk(g_pin_mu);
    return pin_table().size();
}

}  // namespace mi

extern "C" {

int mi_abi_version(void) { return MI355_INTERP_ABI_VERSION; }

size_t mi_debug_pinned_ranges(void) { return mi::pinned_ranges(); }

const char* mi_last_error(const mi_ctx* ctx) { return ctx ? ctx->err : mi::tls_error(); }

mi_status mi_ctx_create(int device, mi_ctx** out)
{
    if (!out) return mi::fail(nullptr, MI_ERR_INVALID_ARG, "mi_ctx_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return mi::fail(nullptr, MI_ERR_NO_DEVICE,
                        "mi_ctx_create: no HIP device available (%s); this library has no CPU fallback",
                        e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= count)
        return mi::fail(nullptr, MI_ERR_INVALID_ARG, "mi_ctx_create: device %d out of range [0,%d)", device, count);
    MI_HIP(nullptr, hipSetDevice(device));
    hipDeviceProp_t prop;
    MI_HIP(nullptr, hipGetDeviceProperties(&prop, device));
    mi_ctx* c = new (std::nothrow) mi_ctx();
    if (!c) return mi::fail(nullptr, MI_ERR_NOMEM, "mi_ctx_create: out of host memory");
    c->device = device;
    c->compute_units = prop.multiProcessorCount;
    c->hbm_bytes = prop.totalGlobalMem;
    snprintf(c->name, sizeof(c->name), "%s (%s)", prop.name, prop.gcnArchName);
    e = hipMalloc(&c->reduce_ws, mi_ctx::kReduceWsBytes);
    if (e != hipSuccess) {
        delete c;
        return mi::fail(nullptr, MI_ERR_NOMEM, "mi_ctx_create: hipMalloc of the reduction workspace failed: %s",
                        hipGetErrorString(e));
    }
    const int consts[3] = {0, 1, 0};
    e = hipMemcpy(static_cast<char*>(c->reduce_ws) + mi_ctx::kFlagOffset, consts, sizeof(consts), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(c->reduce_ws);
        delete c;
        return mi::fail(nullptr, MI_ERR_HIP, "mi_ctx_create: workspace initialisation failed: %s", hipGetErrorString(e));
    }
    e = hipHostMalloc(reinterpret_cast<void**>(&c->probe_host), 64, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->probe_host_dev), c->probe_host, 0);
    if (e != hipSuccess) {
        if (c->probe_host) (void)hipHostFree(c->probe_host);
        (void)hipFree(c->reduce_ws);
        delete c;
        return mi::fail(nullptr, MI_ERR_HIP, "mi_ctx_create: probe mailbox allocation failed: %s", hipGetErrorString(e));
    }
    *c->probe_host = -1;
    *out = c;
    return MI_OK;
}

mi_status mi_ctx_set_query_order(mi_ctx* ctx, int order)
{
    MI_REQUIRE(ctx, ctx != nullptr, "mi_ctx_set_query_order: ctx is NULL");
    MI_REQUIRE(ctx, order == MI_QUERIES_AUTO || order == MI_QUERIES_RANDOM || order == MI_QUERIES_ORDERED,
               "mi_ctx_set_query_order: unknown value %d", order);
    ctx->query_order = order;
    *reinterpret_cast<volatile int*>(ctx->probe_host) = -1;   // forget what earlier query sets looked like
    return MI_OK;
}

mi_status mi_ctx_destroy(mi_ctx* ctx)
{
    if (!ctx) return MI_OK;
    (void)hipSetDevice(ctx->device);
    // drain first: kernels in flight still use the reduction workspace (partials, order flags), the scratch slots
    // and the probe mailbox
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->owned_stream && ctx->owned_stream != ctx->stream) (void)hipStreamSynchronize(ctx->owned_stream);
    if (ctx->aux_stream) (void)hipStreamSynchronize(ctx->aux_stream);
    for (int i = 0; i < 3; ++i)
        if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
    if (ctx->reduce_ws) (void)hipFree(ctx->reduce_ws);
    if (ctx->owned_stream) (void)hipStreamDestroy(ctx->owned_stream);
    if (ctx->aux_event) (void)hipEventDestroy(ctx->aux_event);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->probe_host) (void)hipHostFree(ctx->probe_host);
    delete ctx;
    return MI_OK;
}

mi_status mi_ctx_set_stream(mi_ctx* ctx, void* stream)
{
    MI_REQUIRE(ctx, ctx != nullptr, "mi_ctx_set_stream: ctx is NULL");
    ctx->stream = (hipStream_t)stream;
    return MI_OK;
}

mi_status mi_ctx_own_stream(mi_ctx* ctx)
{
    MI_REQUIRE(ctx, ctx != nullptr, "mi_ctx_own_stream: ctx is NULL");
    MI_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->owned_stream) MI_HIP(ctx, hipStreamCreateWithFlags(&ctx->owned_stream, hipStreamNonBlocking));
    ctx->stream = ctx->owned_stream;
    return MI_OK;
}

mi_status mi_ctx_synchronize(mi_ctx* ctx)
{
    MI_REQUIRE(ctx, ctx != nullptr, "mi_ctx_synchronize: ctx is NULL");
    MI_HIP(ctx, hipSetDevice(ctx->device));
    MI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MI_OK;
}

mi_status mi_ctx_device_info(mi_ctx* ctx, char* name, size_t name_len, int* compute_units, size_t* hbm_bytes)
{
    MI_REQUIRE(ctx, ctx != nullptr, "mi_ctx_device_info: ctx is NULL");
    if (name && name_len) {
        strncpy(name, ctx->name, name_len - 1);
        name[name_len - 1] = 0;
    }
    if (compute_units) *compute_units = ctx->compute_units;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    return MI_OK;
}

mi_status mi_timer_create(mi_ctx* ctx, mi_timer** out)
{
    MI_REQUIRE(ctx, ctx && out, "mi_timer_create: NULL argument");
    MI_HIP(ctx, hipSetDevice(ctx->device));
    mi_timer* t = new (std::nothrow) mi_timer();
    if (!t) return mi::fail(ctx, MI_ERR_NOMEM, "mi_timer_create: out of host memory");
    t->ctx = ctx;
    MI_HIP(ctx, hipEventCreate(&t->start));
    MI_HIP(ctx, hipEventCreate(&t->stop));
    *out = t;
    return MI_OK;
}

mi_status mi_timer_destroy(mi_timer* t)
{
    if (!t) return MI_OK;
    (void)hipEventDestroy(t->start);
    (void)hipEventDestroy(t->stop);
    delete t;
    return MI_OK;
}

mi_status mi_timer_start(mi_timer* t)
{
    MI_REQUIRE(nullptr, t != nullptr, "mi_timer_start: timer is NULL");
    MI_HIP(t->ctx, hipSetDevice(t->ctx->device));
    MI_HIP(t->ctx, hipEventRecord(t->start, t->ctx->stream));
    return MI_OK;
}

mi_status mi_timer_stop(mi_timer* t)
{
    MI_REQUIRE(nullptr, t != nullptr, "mi_timer_stop: timer is NULL");
    MI_HIP(t->ctx, hipSetDevice(t->ctx->device));
    MI_HIP(t->ctx, hipEventRecord(t->stop, t->ctx->stream));
    return MI_OK;
}

mi_status mi_timer_elapsed_ms(mi_timer* t, float* ms)
{
    MI_REQUIRE(nullptr, t && ms, "mi_timer_elapsed_ms: NULL argument");
    MI_HIP(t->ctx, hipEventSynchronize(t->stop));
    MI_HIP(t->ctx, hipEventElapsedTime(ms, t->start, t->stop));
    return MI_OK;
}

}  // extern "C"
