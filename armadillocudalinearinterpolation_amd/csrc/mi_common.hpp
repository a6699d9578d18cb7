// Internal declarations shared by the translation units of libmi355interp.so.
// gfx950 only; nothing here is part of the public ABI (include/mi355_interp.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>

#include "mi355_interp.h"

struct mi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int compute_units = 0;
    size_t hbm_bytes = 0;
    char name[128] = {0};
    mutable char err[512] = {0};
    // scratch for host-convenience entry points (grown on demand, never shrunk)
    void* scratch[3] = {nullptr, nullptr, nullptr};
    size_t scratch_bytes[3] = {0, 0, 0};
    // fixed workspace for cross-workgroup reductions, allocated at creation so
    // that device entry points never allocate
    void* reduce_ws = nullptr;
    static constexpr size_t kReduceWsBytes = 192 * 1024;
    // three ints inside the workspace: constant 0, constant 1, and the query-order probe's verdict
    static constexpr size_t kFlagOffset = 176 * 1024;
    int query_order = 0;         // MI_QUERIES_AUTO / _RANDOM / _ORDERED (mi_ctx_set_query_order)
    // MI_QUERIES_AUTO: the probe kernel also drops its verdict into a pinned host int (never waited for); the host
    // reads whatever is there at the next call and uses it only to PREDICT which kernel to launch -- either kernel
    // is correct on any input.  -1 = no verdict yet (both kernels are launched, gated on the device-side flag).
    // host-convenience entry points overlap H2D, kernel and D2H of consecutive chunks: the copy back runs here
    hipStream_t owned_stream = nullptr;  // mi_ctx_own_stream
    hipStream_t aux_stream = nullptr;   // created on first use
    hipEvent_t aux_event = nullptr;
    int* probe_host = nullptr;       // host view
    int* probe_host_dev = nullptr;   // device view of the same int
};

struct mi_timer {
    mi_ctx* ctx;
    hipEvent_t start, stop;
};

namespace mi {

// last error of the calling thread for calls that have no context yet
char* tls_error();

mi_status fail(const mi_ctx* ctx, mi_status code, const char* fmt, ...);

// streaming launch shape: enough 256-thread workgroups to fill 256 CUs at
// 8 workgroups/CU, grid-stride over the rest (hipguide Guideline 11)
inline unsigned stream_grid(const mi_ctx* ctx, size_t work_items, unsigned block, unsigned per_cu = 8)
{
    size_t need = (work_items + block - 1) / block;
    size_t cap = (size_t)(ctx->compute_units > 0 ? ctx->compute_units : 256) * per_cu;
    if (need < 1) need = 1;
    return (unsigned)(need < cap ? need : cap);
}

mi_status ensure_scratch(mi_ctx* ctx, int slot, size_t bytes);
mi_status ensure_aux_stream(mi_ctx* ctx);

// Page-locking of a caller's host range for the duration of one host-convenience call (asynchronous copies need it).
// Process-wide and reference counted: two calls (two contexts, two threads) that hand over the SAME array share one
// registration, and the range is unregistered only when the last of them has drained its streams -- an early
// hipHostUnregister by one call would pull the pinning out from under the other's copies in flight.  A range the
// library did not register itself (already pinned by the caller, or not pinnable) is left alone: hipMemcpyAsync then
// takes its blocking path.  pin_host returns true when this call holds a reference it must give back with unpin_host.
bool pin_host(const void* p, size_t bytes);
void unpin_host(const void* p);
size_t pinned_ranges();   // test hook (mi_debug_pinned_ranges)

}  // namespace mi

#define MI_HIP(ctx, call)                                                                   \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return mi::fail((ctx), MI_ERR_HIP, "%s failed at %s:%d: %s", #call, __FILE__,   \
                            __LINE__, hipGetErrorString(e_));                               \
    } while (0)

#define MI_LAUNCH_CHECK(ctx, what)                                                          \
    do {                                                                                    \
        hipError_t e_ = hipGetLastError();                                                  \
        if (e_ != hipSuccess)                                                               \
            return mi::fail((ctx), MI_ERR_HIP, "launch of %s failed at %s:%d: %s", (what),  \
                            __FILE__, __LINE__, hipGetErrorString(e_));                     \
    } while (0)

#define MI_REQUIRE(ctx, cond, ...)                                                          \
    do {                                                                                    \
        if (!(cond)) return mi::fail((ctx), MI_ERR_INVALID_ARG, __VA_ARGS__);               \
    } while (0)
