// 1-D table interpolation on MI355X (gfx950): kernel choice and the mi_interp1_* entry points.  Hand-written HIP;
// memory-bound (8 B in + 8 B out per query), no MFMA.
//
// Kernels (all share eval_batch_from, mi_interp1_eval.hpp, so their results are bit-identical; launch_mode picks one):
//   interp1_vec_kernel        streaming: full-size grid, two 16-B vectors per lane.  Ordered / clustered queries,
//                             small tables, tails.                                         (mi_interp1_stream.hpp)
//   interp1_lds_kernel        the whole table in LDS (<= 128 KiB): unordered queries over small tables.
//   interp1_scalar_kernel     unaligned pointers.
//   interp1_order_probe       1024-sample test "are the queries already ordered?" (also inlined into the kernels).
//   interp1_sweep_kernel      region sweep: persistent workgroups order a 16 K-query tile by table region in LDS so
//   interp1_sweep_pipe_kernel that the whole chip gathers from the same part of the table at the same time.
//                             Unordered queries over tables that outgrow L2 (and mid-size {x,y} tables).
//                                                                                           (mi_interp1_sweep.hpp)
// Tables: mi_interp1_tables.hip.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "mi_interp1_stream.hpp"
#include "mi_interp1_sweep.hpp"

using namespace mi_interp1;

namespace {

// environment hooks of the tuning scripts (scripts/gpu_sweep_midsize.py, gpu_strong_scaling_shards.sh,
// tests/test_interp_gpu.py::test_both_sweep_kernel_forms_are_bit_identical), each read ONCE per process
struct SweepEnv {
    long min_bytes;          // MI_SWEEP_MIN_BYTES: table size from which the sweep may be picked (-1: the measured windows)
    size_t min_tiles_per_cu; // MI_SWEEP_MIN_TILES_PER_CU
    int variant;             // MI_SWEEP_VARIANT: 2 pipelined two-group form (default), 1 one phase after the other
    bool variant_forced;
};
const SweepEnv& sweep_env()
{
    static const SweepEnv e = [] {
        SweepEnv v{-1, kSweepMinTilesPerCu, 2, false};
        if (const char* s = getenv("MI_SWEEP_MIN_BYTES")) v.min_bytes = (long)strtoull(s, nullptr, 10);
        if (const char* s = getenv("MI_SWEEP_MIN_TILES_PER_CU")) v.min_tiles_per_cu = (size_t)strtoull(s, nullptr, 10);
        if (const char* s = getenv("MI_SWEEP_VARIANT")) {
            const int x = atoi(s);
            v.variant = (x == 1 || x == 2) ? x : 2;
            v.variant_forced = true;
        }
        return v;
    }();
    return e;
}

// Gated launches (order_flag != nullptr) exit at once when the probe saw unordered queries; what that costs is the
// dispatch of the grid's waves (26-30 us at 1e8 queries; 1024-lane workgroups cost the same, 4x fewer waves save
// 15-19 us but run 15 % slower on ordered input), which is why launch_mode only gates while it has no prediction.
constexpr int kGatedBlock = 256, kGatedVpl = 2;
template <int MODE, int FORMULA, int BLOCK, int VPL>
mi_status launch_vec_shape(mi_ctx* ctx, const G1Dev& d, const double* xq, double* yq, size_t nq, double extrap,
                           const int* order_flag, const ProbeArgs& probe)
{
    const size_t lanes = (nq >> 1) + (nq & 1);                 // one lane per vector (+ one for an odd tail)
    const size_t per_block = (size_t)BLOCK * VPL;
    const size_t grid = (lanes + per_block - 1) / per_block;
    if (grid > 0x7fffffffull) return mi::fail(ctx, MI_ERR_INVALID_ARG, "mi_interp1_f64_dev: nq=%zu too large for one launch", nq);
    hipLaunchKernelGGL((interp1_vec_kernel<MODE, FORMULA, BLOCK, VPL>), dim3((unsigned)grid), dim3(BLOCK), 0, ctx->stream,
                       d, xq, yq, nq, extrap, order_flag, probe);
    MI_LAUNCH_CHECK(ctx, "interp1 streaming kernel");
    return MI_OK;
}

template <int MODE, int FORMULA>
mi_status launch_vec(mi_ctx* ctx, const G1Dev& d, const double* xq, double* yq, size_t nq, double extrap,
                     const int* order_flag = nullptr, const ProbeArgs& probe = ProbeArgs{})
{
    if (order_flag)
        return launch_vec_shape<MODE, FORMULA, kGatedBlock, kGatedVpl>(ctx, d, xq, yq, nq, extrap, order_flag, probe);
    return launch_vec_shape<MODE, FORMULA, kBlock, kVecVpl>(ctx, d, xq, yq, nq, extrap, nullptr, probe);
}

template <int MODE, int FORMULA = 0>
mi_status launch_mode(mi_ctx* ctx, const G1Dev& d, size_t table_bytes, const double* xq, double* yq, size_t nq,
                      double extrap)
{
    const bool aligned = ((reinterpret_cast<uintptr_t>(xq) | reinterpret_cast<uintptr_t>(yq)) & 15u) == 0;
    if (!aligned) {
        const size_t grid = (nq + kBlock - 1) / kBlock;
        if (grid > 0x7fffffffull) return mi::fail(ctx, MI_ERR_INVALID_ARG, "mi_interp1_f64_dev: nq=%zu too large for one launch", nq);
        hipLaunchKernelGGL((interp1_scalar_kernel<MODE, FORMULA>), dim3((unsigned)grid), dim3(kBlock), 0, ctx->stream, d, xq, yq,
                           nq, extrap);
        MI_LAUNCH_CHECK(ctx, "interp1 scalar kernel");
        return MI_OK;
    }
    // Region sweep only pays when the table cannot live in one XCD's 4 MiB L2 (measured window, random queries,
    // profiles/r01_sweep_vs_stream_by_table_size.log: 0.91-0.99x below 5 MB, 1.06x at 5.6 MB, 1.39x at 8 MB, 1.7x at
    // 16-32 MB, still 1.05x at 512 MB) and there are enough tiles to keep every CU busy for several sweeps; ordered query sets are detected on the device (or declared by the caller).
    const unsigned cus = (unsigned)(ctx->compute_units > 0 ? ctx->compute_units : 256);
    const size_t ntiles = nq / kSweepTile;
    // {x,y} tables with the centred guess (mode 3) gain from the sweep much earlier: ordering a tile by region makes
    // the lanes of a wave share lines, and three gathers per query are expensive in the streaming kernel
    // (profiles/r01_sweep_vs_stream_midsize_tables.log: 1.3x at 0.16-0.5 MB, 1.05-1.17x at 1-2.4 MB, 0.92-0.99x between
    // 2.8 and 3.6 MB, 1.12-1.25x from 4 MB); closed-form tables gain 5-7 % between 0.26 and 1.6 MB, not taken.
    bool size_ok = table_bytes >= ((size_t)5 << 20);
    if (MODE == 3) size_ok = table_bytes > kLdsMaxTableBytes && !(table_bytes > 2600000 && table_bytes < 3900000);
    if (sweep_env().min_bytes >= 0) size_ok = table_bytes >= (size_t)sweep_env().min_bytes;   // tuning hook
    // enough tiles for every CU to run a few sweeps (profiles/r02_strong_scaling_shards.log)
    const bool sweep_ok = ctx->query_order != MI_QUERIES_ORDERED && size_ok &&
                          ntiles >= (size_t)cus * sweep_env().min_tiles_per_cu && std::isfinite(d.xmax - d.xmin) && (d.xmax - d.xmin) > 0.0;
    if constexpr (MODE == 0 || MODE == 3) {
        // Whole table in LDS: unordered queries over a table that outgrows L1 (32 KiB) but fits LDS (128 KiB).
        // scripts/gpu_small_table_timing.py, 1e8 queries: random 0.516 -> 0.287 ms at 10-16 K nodes; sorted queries are
        // better off in the streaming kernel (0.25 vs 0.29 ms), so AUTO follows the previous call's probe verdict here
        // too (the first call takes the LDS kernel, which is never far off).
        const size_t ybytes = ((size_t)d.n + 1) * (MODE == 0 ? sizeof(double) : sizeof(d2));
        if (ctx->query_order != MI_QUERIES_ORDERED && ybytes > (MODE == 0 ? kLdsMinTableBytes0 : kLdsMinTableBytes3) && ybytes <= kLdsMaxTableBytes &&
            nq >= 8 * kLdsQueriesPerBlock && std::isfinite(d.xmax - d.xmin) && (d.xmax - d.xmin) > 0.0) {
            ProbeArgs probe{};
            if (ctx->query_order == MI_QUERIES_AUTO) {
                const int predicted = *reinterpret_cast<volatile int*>(ctx->probe_host);
                probe = ProbeArgs{xq, nq, d.xmin, (double)kSweepBins / (d.xmax - d.xmin), nullptr, ctx->probe_host_dev};
                if (predicted == 1) return launch_vec<MODE, FORMULA>(ctx, d, xq, yq, nq, extrap, nullptr, probe);
            }
            MI_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&interp1_lds_kernel<MODE, FORMULA>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMaxTableBytes));
            const size_t grid = (nq + kLdsQueriesPerBlock - 1) / kLdsQueriesPerBlock;
            hipLaunchKernelGGL((interp1_lds_kernel<MODE, FORMULA>), dim3((unsigned)grid), dim3(kLdsBlock), ybytes, ctx->stream, d, xq,
                               yq, nq, extrap, probe);
            MI_LAUNCH_CHECK(ctx, "interp1 LDS-table kernel");
            return MI_OK;
        }
    }
    if (!sweep_ok) return launch_vec<MODE, FORMULA>(ctx, d, xq, yq, nq, extrap);

    const double bscale = (double)kSweepBins / (d.xmax - d.xmin);
    int* flags = reinterpret_cast<int*>(static_cast<char*>(ctx->reduce_ws) + mi_ctx::kFlagOffset);   // {0, 1, probe}
    const size_t head = ntiles * kSweepTile;
    const unsigned grid = (unsigned)std::min<size_t>(ntiles, (size_t)cus);   // persistent, one workgroup per CU
    // plan 0: region sweep, 1: streaming kernel, 2: separate probe, then both kernels gated on its device-side flag
    int plan = 0;
    ProbeArgs probe{};   // host_mailbox == nullptr: no probe
    if (ctx->query_order == MI_QUERIES_AUTO) {
        // What did the probe say about an earlier query set?  (Never waited for; stale or missing is fine: the
        // verdict only selects the faster of two kernels that are both correct on any input.)
        const int predicted = *reinterpret_cast<volatile int*>(ctx->probe_host);
        plan = (predicted == 0 || predicted == 1) ? predicted : 2;
        probe = ProbeArgs{xq, nq, d.xmin, bscale, plan == 2 ? flags + 2 : nullptr, ctx->probe_host_dev};
    }
    if (plan == 1) return launch_vec<MODE, FORMULA>(ctx, d, xq, yq, nq, extrap, nullptr, probe);   // probes inline
    if (plan == 2) {
        hipLaunchKernelGGL(interp1_order_probe, dim3(1), dim3(64), 0, ctx->stream, probe);
        MI_LAUNCH_CHECK(ctx, "interp1 order probe");
        hipLaunchKernelGGL((interp1_sweep_kernel<MODE, FORMULA>), dim3(grid), dim3(kSweepThreads), 0, ctx->stream, d, xq,
                           yq, ntiles, extrap, bscale, flags + 2, (size_t)0, ProbeArgs{});
        MI_LAUNCH_CHECK(ctx, "interp1 region-sweep kernel");
        mi_status st = launch_vec<MODE, FORMULA>(ctx, d, xq, yq, head, extrap, flags + 2);   // ordered after all
        if (st != MI_OK) return st;
        if (nq > head) return launch_vec<MODE, FORMULA>(ctx, d, xq + head, yq + head, nq - head, extrap);
        return MI_OK;
    }
    // one launch: sort-and-gather tiles, the ragged tail, and the probe for the next call (flags[0] is a constant 0)
    // the pipelined form pays one extra (fill) step per workgroup: it wins from about 16 tiles per CU (1e8 queries: 0.66-0.69
    // against 0.71 ms) and is 1 % behind at 3-12 tiles per CU (profiles/r02_strong_scaling_shards.log)
    if (sweep_env().variant == 2 && (sweep_env().variant_forced || ntiles >= (size_t)cus * 16)) {
        const unsigned pgrid = (unsigned)std::min<size_t>(ntiles, (size_t)cus);   // one 1024-lane workgroup per CU
        hipLaunchKernelGGL((interp1_sweep_pipe_kernel<MODE, FORMULA>), dim3(pgrid), dim3(kPipeThreads), 0, ctx->stream, d, xq, yq,
                           ntiles, extrap, bscale, flags, nq - head, probe);
        MI_LAUNCH_CHECK(ctx, "interp1 pipelined region-sweep kernel");
        return MI_OK;
    }
    hipLaunchKernelGGL((interp1_sweep_kernel<MODE, FORMULA>), dim3(grid), dim3(kSweepThreads), 0, ctx->stream, d, xq, yq,
                       ntiles, extrap, bscale, flags, nq - head, probe);
    MI_LAUNCH_CHECK(ctx, "interp1 region-sweep kernel");
    return MI_OK;
}

}  // namespace

extern "C" {

mi_status mi_interp1_f64_dev(mi_ctx* ctx, const mi_grid1* g, const double* xq, double* yq, size_t nq, double extrap)
{
    MI_REQUIRE(ctx, ctx && g, "mi_interp1_f64_dev: NULL context or grid");
    if (nq == 0) return MI_OK;
    MI_REQUIRE(ctx, xq && yq, "mi_interp1_f64_dev: NULL query/result pointer");
    MI_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(xq) | reinterpret_cast<uintptr_t>(yq)) & 7u) == 0,
               "mi_interp1_f64_dev: pointers must be 8-byte aligned");
    MI_HIP(ctx, hipSetDevice(ctx->device));   // a process may hold contexts on several devices (mi_group)
    switch (g->mode) {
        case 0:
            if (g->d.formula == 1) return launch_mode<0, 1>(ctx, g->d, g->table_bytes, xq, yq, nq, extrap);
            if (g->d.formula == 2) return launch_mode<0, 2>(ctx, g->d, g->table_bytes, xq, yq, nq, extrap);
            if (g->d.formula == 3) return launch_mode<0, 3>(ctx, g->d, g->table_bytes, xq, yq, nq, extrap);
            return launch_mode<0, 0>(ctx, g->d, g->table_bytes, xq, yq, nq, extrap);
        case 1:
            if (g->d.centred) return launch_mode<3>(ctx, g->d, g->table_bytes, xq, yq, nq, extrap);
            return launch_mode<1>(ctx, g->d, g->table_bytes, xq, yq, nq, extrap);
        default: return launch_mode<2>(ctx, g->d, g->table_bytes, xq, yq, nq, extrap);
    }
}

mi_status mi_interp1_f64_host(mi_ctx* ctx, const mi_grid1* g, const double* xq, double* yq, size_t nq, double extrap)
{
    MI_REQUIRE(ctx, ctx && g, "mi_interp1_f64_host: NULL context or grid");
    if (nq == 0) return MI_OK;
    MI_REQUIRE(ctx, xq && yq, "mi_interp1_f64_host: NULL query/result pointer");
    MI_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = nq * sizeof(double);
    mi_status st = mi::ensure_scratch(ctx, 0, bytes);
    if (st != MI_OK) return st;
    st = mi::ensure_scratch(ctx, 1, bytes);
    if (st != MI_OK) return st;
    // Chunks of 8 M queries: the copy back of chunk k (aux stream) overlaps the upload of chunk k+1 (main stream) --
    // PCIe carries both directions at once (harness exp_pcie.hip, scripts/ARCHIVE.md: 0.8 GB each way 14 + 14 ms one after the other,
    // 16.5 ms together); the kernel is 4 % of either copy.  Small calls keep the single-shot form.
    const size_t chunk = (size_t)8 << 20;
    if (nq <= 2 * chunk) {
        MI_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], xq, bytes, hipMemcpyHostToDevice, ctx->stream));
        st = mi_interp1_f64_dev(ctx, g, (const double*)ctx->scratch[0], (double*)ctx->scratch[1], nq, extrap);
        if (st != MI_OK) return st;
        MI_HIP(ctx, hipMemcpyAsync(yq, ctx->scratch[1], bytes, hipMemcpyDeviceToHost, ctx->stream));
        MI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return MI_OK;
    }
    st = mi::ensure_aux_stream(ctx);
    if (st != MI_OK) return st;
    // Asynchronous copies need page-locked host memory: pin the caller's arrays for the duration of the call (the
    // runtime keeps the pinning of a range it has seen before, so this costs nothing from the second call on; a
    // range that cannot be pinned, or is pinned already, simply takes the blocking path inside hipMemcpyAsync).
    const bool pin_in = mi::pin_host(xq, bytes), pin_out = mi::pin_host(yq, bytes);
    const double* din = (const double*)ctx->scratch[0];
    double* dout = (double*)ctx->scratch[1];
    // No early return between here and the unpinning below: a failing call leaves the loop, both streams are
    // drained (copies already queued still write into the caller's arrays), and only then are the ranges released.
    hipError_t herr = hipSuccess;
    const char* what = "";
    if (getenv("MI_TEST_FAIL_HOST_CHUNK")) { herr = hipErrorUnknown; what = "MI_TEST_FAIL_HOST_CHUNK (error-path test hook)"; }
    for (size_t off = 0; off < nq && herr == hipSuccess && st == MI_OK; off += chunk) {
        const size_t m = std::min(chunk, nq - off);
        herr = hipMemcpyAsync((void*)(din + off), xq + off, m * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (herr != hipSuccess) { what = "upload of a query chunk"; break; }
        st = mi_interp1_f64_dev(ctx, g, din + off, dout + off, m, extrap);
        if (st != MI_OK) break;
        herr = hipEventRecord(ctx->aux_event, ctx->stream);
        if (herr == hipSuccess) herr = hipStreamWaitEvent(ctx->aux_stream, ctx->aux_event, 0);
        if (herr == hipSuccess) herr = hipMemcpyAsync(yq + off, dout + off, m * sizeof(double), hipMemcpyDeviceToHost, ctx->aux_stream);
        if (herr != hipSuccess) what = "download of a result chunk";
    }
    const hipError_t e1 = hipStreamSynchronize(ctx->stream), e2 = hipStreamSynchronize(ctx->aux_stream);
    if (pin_in) mi::unpin_host(xq);
    if (pin_out) mi::unpin_host(yq);
    if (st != MI_OK) return st;
    if (herr != hipSuccess) return mi::fail(ctx, MI_ERR_HIP, "mi_interp1_f64_host: %s failed: %s", what, hipGetErrorString(herr));
    MI_HIP(ctx, e1);
    MI_HIP(ctx, e2);
    return MI_OK;
}

mi_status mi_interp1_f64(mi_ctx* ctx, const double* x, const double* y, size_t n, const double* xq, double* yq,
                         size_t nq, double extrap)
{
    mi_grid1* g = nullptr;
    mi_status st = mi_grid1_create(ctx, x, y, n, MI_GRID_SANITISE, &g);
    if (st != MI_OK) return st;
    st = mi_interp1_f64_host(ctx, g, xq, yq, nq, extrap);
    mi_grid1_destroy(g);
    return st;
}

}  // extern "C"
