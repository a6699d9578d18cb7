// interp1 kernels that stream the queries straight through (one pass, no reordering): the vector kernel (ordered /
// clustered queries, small tables, tails), the whole-table-in-LDS kernel (unordered queries over tables up to 128 KiB),
// the scalar kernel (unaligned pointers) and the stand-alone query-order probe.  Arithmetic: mi_interp1_eval.hpp.
#pragma once
#include "mi_interp1_eval.hpp"

namespace mi_interp1 {

constexpr int kBlock = 256;

__global__ __launch_bounds__(64) void interp1_order_probe(ProbeArgs p) { order_probe_wave(p); }

// Vector kernel: a fixed VPL = 2 16-B vectors (four queries) per lane and one workgroup per 8 KiB of
// queries, no grid-stride loop.  Measured on MI355X (profiles/r01_exp_stream_shapes.log): this shape streams 8 B in +
// 8 B out per element at 6.5 TB/s, a grid capped at 2048 workgroups with a grid-stride loop at 5.0 TB/s.
// Non-temporal loads/stores: the streams must not evict the table from L2.  Requires xq, yq 16-B aligned.
constexpr int kVecVpl = 2;   // 16-B vectors per lane: 1 / 2 / 4 measured 0.276 / 0.250 / 0.253 ms (sorted), random unchanged
template <int MODE, int FORMULA, int BLOCK, int VPL>
__global__ __launch_bounds__(BLOCK) void interp1_vec_kernel(G1Dev g, const double* __restrict__ xq,
                                                             double* __restrict__ yq, size_t nq,
                                                             double extrap, const int* __restrict__ order_flag,
                                                             ProbeArgs probe)
{
    if (order_flag && *order_flag == 0) return;   // unordered queries: the region-sweep kernel does the work
    if (probe.host_mailbox && blockIdx.x == 0 && threadIdx.x < 64) order_probe_wave(probe);   // for the next call
    const size_t nvec = nq >> 1;
    const size_t base = (size_t)blockIdx.x * (BLOCK * VPL) + threadIdx.x;
    double q[2 * VPL], r[2 * VPL];
    bool full = base + (size_t)(VPL - 1) * BLOCK < nvec;
    if (full) {
#pragma unroll
        for (int u = 0; u < VPL; ++u) {
            const d2 v = __builtin_nontemporal_load(reinterpret_cast<const d2*>(xq) + base + (size_t)u * BLOCK);
            q[2 * u] = v.x;
            q[2 * u + 1] = v.y;
        }
        eval_batch<MODE, 2 * VPL, FORMULA, true>(g, q, r, extrap);
#pragma unroll
        for (int u = 0; u < VPL; ++u) {
            d2 o;
            o.x = r[2 * u];
            o.y = r[2 * u + 1];
            __builtin_nontemporal_store(o, reinterpret_cast<d2*>(yq) + base + (size_t)u * BLOCK);
        }
    } else {
        for (int u = 0; u < VPL; ++u) {
            const size_t i = base + (size_t)u * BLOCK;
            if (i < nvec) {
                const d2 v = __builtin_nontemporal_load(reinterpret_cast<const d2*>(xq) + i);
                double q1[2] = {v.x, v.y}, r1[2];
                eval_batch<MODE, 2, FORMULA>(g, q1, r1, extrap);
                d2 o;
                o.x = r1[0];
                o.y = r1[1];
                __builtin_nontemporal_store(o, reinterpret_cast<d2*>(yq) + i);
            } else if ((nq & 1) && i == nvec) {   // odd tail element, handled by the first lane past the vectors
                double q1[1] = {xq[nq - 1]}, r1[1];
                eval_batch<MODE, 1, FORMULA>(g, q1, r1, extrap);
                yq[nq - 1] = r1[0];
            }
        }
    }
}

// ---- small tables: the whole table in LDS ----------------------------------------------------------------
// A table of up to 128 KiB (16 K nodes of a closed-form grid, 8 K {x,y} nodes of a centred-guess grid) fits one CU's LDS.  Unordered queries over such a table are bound by the
// L2 request rate in the streaming kernel (one L2 hit per query: 0.6 ms per 1e8 queries); from LDS the two-node read
// is a ds_read2_b64 and the kernel runs at the streaming rate.  One 1024-lane workgroup copies the table (L2 hits)
// and then evaluates kLdsQueriesPerBlock queries, so the copy is a few per cent of the block's traffic; the grid is
// full-size (one workgroup per chunk), as for the streaming kernel.  Arithmetic = eval_batch: bit-identical.
constexpr int kLdsBlock = 1024;
constexpr size_t kLdsMaxTableBytes = 128 * 1024;
constexpr size_t kLdsMinTableBytes0 = 32 * 1024;  // mode 0: smaller tables live in L1 and one gather per query streams as fast
constexpr size_t kLdsMinTableBytes3 = 2 * 1024;   // mode 3: three gathers per query are TA-bound even from L1 (0.44 vs 0.28 ms)
constexpr size_t kLdsQueriesPerBlock = 1u << 17;
template <int MODE, int FORMULA>
__global__ __launch_bounds__(kLdsBlock) void interp1_lds_kernel(G1Dev g, const double* __restrict__ xq,
                                                                double* __restrict__ yq, size_t nq, double extrap,
                                                                ProbeArgs probe)
{
    extern __shared__ __attribute__((aligned(16))) double ys[];
    if (probe.host_mailbox && blockIdx.x == 0 && threadIdx.x < 64) order_probe_wave(probe);   // for the next call
    {   // n + 1 entries (padding node) of 8 B (mode 0: Y) or 16 B (mode 3: {x,y})
        const double* src = MODE == 0 ? g.y : reinterpret_cast<const double*>(g.nodes);
        const int words = (MODE == 0 ? 1 : 2) * (g.n + 1);
        for (int i = threadIdx.x; i < words; i += kLdsBlock) ys[i] = src[i];
    }
    __syncthreads();
    const size_t q0 = (size_t)blockIdx.x * kLdsQueriesPerBlock;
    const size_t q1 = min(nq, q0 + kLdsQueriesPerBlock);
    const size_t nvec = (q1 - q0) >> 1;                                     // q0 is even: 16-B aligned vectors
    const d2* in = reinterpret_cast<const d2*>(xq + q0);
    d2* out = reinterpret_cast<d2*>(yq + q0);
    size_t v = threadIdx.x;
    for (; v + kLdsBlock < nvec; v += 2 * kLdsBlock) {                      // two vectors (four queries) per lane per trip
        const d2 a = __builtin_nontemporal_load(in + v), b = __builtin_nontemporal_load(in + v + kLdsBlock);
        const double q[4] = {a.x, a.y, b.x, b.y};
        double r[4];
        eval_batch_from<MODE, 4, FORMULA, true, true>(g, q, r, extrap, ys);
        d2 o0, o1;
        o0.x = r[0]; o0.y = r[1]; o1.x = r[2]; o1.y = r[3];
        __builtin_nontemporal_store(o0, out + v);
        __builtin_nontemporal_store(o1, out + v + kLdsBlock);
    }
    for (; v < nvec; v += kLdsBlock) {
        const d2 a = __builtin_nontemporal_load(in + v);
        const double q[2] = {a.x, a.y};
        double r[2];
        eval_batch_from<MODE, 2, FORMULA, true, true>(g, q, r, extrap, ys);
        d2 o;
        o.x = r[0]; o.y = r[1];
        __builtin_nontemporal_store(o, out + v);
    }
    if (((q1 - q0) & 1) && threadIdx.x == 0) {                              // odd tail element of the last chunk
        const double q[1] = {xq[q1 - 1]};
        double r[1];
        eval_batch_from<MODE, 1, FORMULA, true, true>(g, q, r, extrap, ys);
        yq[q1 - 1] = r[0];
    }
}

// Scalar kernel for unaligned query/result pointers.
template <int MODE, int FORMULA>
__global__ __launch_bounds__(kBlock) void interp1_scalar_kernel(G1Dev g, const double* __restrict__ xq,
                                                                double* __restrict__ yq, size_t nq,
                                                                double extrap)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < nq) {
        double q[1] = {xq[i]}, r[1];
        eval_batch<MODE, 1, FORMULA>(g, q, r, extrap);
        yq[i] = r[0];
    }
}

}  // namespace mi_interp1
