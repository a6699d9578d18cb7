// 1-D table interpolation on MI355X (gfx950): HBM-resident tables and the gather-and-blend kernels behind
// mi_interp1_f64_dev.  Hand-written HIP; memory-bound (8 B in + 8 B out per query), no MFMA.
//
// Semantics: include/mi355_interp.h ("fp64 blend") == oracle/interp_oracle.c.
// The file is compiled with -ffp-contract=off so that every product and sum of
// the blend rounds separately, exactly like the oracle.
//
// Kernels (all share eval_batch_from, so their results are bit-identical; launch_mode picks one):
//   interp1_vec_kernel    streaming: full-size grid, two 16-B vectors per lane.  Ordered / clustered queries,
//                         small tables, tails.
//   interp1_sweep_kernel  region sweep: persistent workgroups order a 16 K-query tile by table region in LDS so
//                         that the whole chip gathers from the same part of the table at the same time.
//                         Unordered queries over tables that outgrow L2 (and mid-size {x,y} tables).
//   interp1_lds_kernel    the whole table in LDS (<= 128 KiB): unordered queries over small tables.
//   interp1_scalar_kernel unaligned pointers.
//   interp1_order_probe   1024-sample test "are the queries already ordered?" (also inlined into the kernels).
//
// Table layouts (all with one padding node so that node l+1 is always
// readable):
//   mode 0  implicit uniform   y[n+1]                 8 B/node, X_i from a closed form:
//             formula 0: fma(i, dx, x0)   (mi_grid1_create_uniform)
//             formula 1: x0 + i*dx        (two roundings; numpy.linspace-style)
//             formula 2: x0 + span*(i/(n-1))          (IEEE division)
//             formula 3: as 2, the quotient by a 3-op Markstein step
//                        q0 = i*y, q = fma(fma(-den,q0,i), y, q0), y = RN(1/den);
//                        tried before formula 2 and accepted only if it reproduces
//                        every node bit for bit, so it needs no rounding proof
//           An EXPLICIT grid whose every node is reproduced bit for bit by one of
//           these forms (checked node by node at build time; the last node may be
//           an exception, as linspace pins it) is stored this way too: half the
//           table bytes and one 16-B gather per query instead of 32 B.
//   mode 1  explicit + guess   {x,y}[n+1]            16 B/node, analytic guess + bounded walk
//   mode 3  explicit + centred guess, same storage: the grid stays within one cell of a straight line, origin and
//           slope are chosen so that the guess equals i at every node (verified), the bracket is G-1 or G: no walk
//   mode 2  explicit + buckets {x,y}[n+1] + u32[nb+1] bucket index, then a
//           binary search confined to the bucket's node range
// Mode 1/3 is chosen when the analytic guess g(q) = (q-xmin)*(n-1)/(xmax-xmin)
// provably lands within a few nodes of the bracket for every possible query
// (verified against every node at build time); otherwise mode 2.
#include <algorithm>
#include <cmath>
#include <new>
#include <cstdlib>
#include <vector>

#include "mi_common.hpp"

typedef double d2 __attribute__((ext_vector_type(2)));

struct G1Dev {
    const d2* nodes;     // modes 1, 2
    const double* y;     // mode 0
    const uint32_t* s;   // mode 2
    int n;               // nodes
    int nb;              // buckets (mode 2)
    double xmin, xmax;
    double scale;        // mode 0: 1/dx; mode 1: (n-1)/(xmax-xmin); mode 2: nb/(xmax-xmin)
    double x0, dx;       // mode 0
    double span, den;    // mode 0, formulas 2, 3
    double rden;         // RN(1/den), formula 3
    int formula;         // mode 0: closed form of the abscissae
    int pin_last;        // mode 0: node n-1 is xmax exactly (not the closed form)
    double gorg;         // mode 1, centred guess: G(x) = (int)((x - gorg) * scale) equals i at EVERY node i (build_explicit)
    int centred;         // mode 1: the centred guess holds -> the bracket is G(q) - 1 or G(q), no walk
};

struct mi_grid1 {
    mi_ctx* ctx;
    int device;          // copied at creation: mi_grid1_destroy must not dereference a context that may be gone
    int mode;
    size_t n;
    size_t table_bytes;
    void* dev_nodes;     // d2[n+1] or double[n+1]
    void* dev_s;         // u32[nb+1] or null
    G1Dev d;
};

#ifndef MI_M3_TWO_LOOKUPS
#define MI_M3_TWO_LOOKUPS 0   // mode 3, unordered queries: see eval_batch_from
#endif
#ifndef MI_INTERP1_COOP
#define MI_INTERP1_COOP 0   // wavefront reuse of shared nodes through __shfl (mode 3, ordered queries): implemented,
                            // bit-exact, and slower than the three gathers it replaces (0.304 -> 0.366 ms per 1e8 sorted
                            // queries on the jittered 1e6-node grid): off.  See eval_batch_from.
#endif

namespace {

constexpr int kBlock = 256;
constexpr int kMaxWalk = 4;   // widest guess window for which mode 1 is used

struct __attribute__((packed, aligned(8))) ypair {
    double a, b;
};

__device__ __forceinline__ double blend(double xa, double ya, double xb, double yb, double q)
{
    const double a = q - xa;
    const double b = xb - q;
    const double w = (a > 0.0) ? a / (a + b) : 0.0;
    return (1.0 - w) * ya + w * yb;
}

// abscissa of node i of an implicit grid; must stay in sync with host_unode() below
template <int FORMULA>
__device__ __forceinline__ double unode(const G1Dev& g, int i)
{
    double x;
    if constexpr (FORMULA == 0) x = fma((double)i, g.dx, g.x0);
    else if constexpr (FORMULA == 1) x = g.x0 + (double)i * g.dx;
    else if constexpr (FORMULA == 2) x = g.x0 + g.span * ((double)i / g.den);
    else {
        const double q0 = (double)i * g.rden;
        x = g.x0 + g.span * fma(fma(-g.den, q0, (double)i), g.rden, q0);
    }
    if (g.pin_last && i == g.n - 1) x = g.xmax;
    return x;
}

// NQ independent queries per lane: all guesses first, then all gathers (so the
// loads of the NQ queries are in flight together), then the rare fix-up walks
// and the blend.
// WIN (mode 3 only): fetch node G-1 together with G and G+1 (ordered queries: the lanes share lines and the extra
// gather is nearly free, while a dependent one would stall the stream); without it node G-1 is fetched only by the
// lanes that need it (unordered queries: every gather is an L2 request).
typedef __attribute__((address_space(3))) const double lds_cdouble;

// node i of an {x,y} table that lives in global memory, or (LDSY) in a workgroup's LDS copy (explicit address space)
template <bool LDSY>
__device__ __forceinline__ d2 load_node(const double* tab, int i)
{
    d2 v;
    if constexpr (LDSY) {
        const lds_cdouble* p = (const lds_cdouble*)(tab + 2 * (size_t)i);
        v.x = p[0];
        v.y = p[1];
    } else {
        v = reinterpret_cast<const d2*>(tab)[i];
    }
    return v;
}

template <int MODE, int NQ, int FORMULA = 0, bool WIN = false, bool LDSY = false>
__device__ __forceinline__ void eval_batch_from(const G1Dev& g, const double (&q)[NQ], double (&out)[NQ],
                                                double extrap, const double* ytab)
{
    // ytab: the table -- g.y (mode 0) or g.nodes (mode 3) in HBM/L2, or (LDSY) a workgroup's LDS copy of it;
    // modes 1 and 2 read g.nodes directly
    double qs[NQ];
    int l[NQ];
    bool oor[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        oor[k] = !(q[k] >= g.xmin && q[k] <= g.xmax);   // true for NaN as well
        qs[k] = oor[k] ? g.xmin : q[k];
    }
    if constexpr (MODE == 0) {
        ypair yp[NQ];
        double xl[NQ], xr[NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            int i = (int)((qs[k] - g.x0) * g.scale);
            i = min(max(i, 0), g.n - 1);
            // invariant: xl = node(i), xr = node(min(i+1, n-1)); the walks are rare (rounding of the guess)
            double a = unode<FORMULA>(g, i), b = unode<FORMULA>(g, min(i + 1, g.n - 1));
            while (i > 0 && a > qs[k]) { --i; b = a; a = unode<FORMULA>(g, i); }
            while (i < g.n - 1 && b <= qs[k]) { ++i; a = b; b = unode<FORMULA>(g, min(i + 1, g.n - 1)); }
            l[k] = i;
            xl[k] = a;
            xr[k] = b;
        }
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            if constexpr (LDSY) {   // explicit LDS address space: ds_read2_b64 instead of a flat load
                const lds_cdouble* p = (const lds_cdouble*)(ytab + l[k]);
                yp[k].a = p[0];
                yp[k].b = p[1];
            } else {
                yp[k] = *reinterpret_cast<const ypair*>(ytab + l[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < NQ; ++k) out[k] = blend(xl[k], yp[k].a, xr[k], yp[k].b, qs[k]);
    } else if constexpr (MODE == 3) {
        // Mode 1 with a centred guess: G is monotone and G(X_i) == i at every node (verified at build time with
        // this very expression), so X_l <= q < X_{l+1} gives G(q) in {l, l+1}: the bracket is G-1 or G.  Three
        // independent gathers, a select, no dependent load and no loop.
        d2 nm[NQ], n0[NQ], n1[NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int i = (int)((qs[k] - g.gorg) * g.scale);
            l[k] = min(max(i, 0), g.n - 1);
        }
#if MI_INTERP1_COOP
        // Wavefront reuse of shared abscissae (ordered queries, streaming kernel): when every lane's guess lies within
        // +-30 nodes of lane 0's, the wave loads one 64-node window with a single coalesced gather (lane j: node
        // base + j) and each lane picks its nodes out of the other lanes' registers with __shfl (ds_bpermute) --
        // 1 + 8 cross-lane reads instead of three 16-B gathers per query.  Otherwise: the three gathers.
        // Measured (scripts/gpu_interp_timing.py, -DMI_INTERP1_COOP=1): 21 % SLOWER -- lanes that share a line are
        // already merged by the texture path, and eight ds_bpermute plus their waits cost more than what is left.
        bool coop[NQ];
        if constexpr (WIN && !LDSY) {
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const int ref = __builtin_amdgcn_readfirstlane(l[k]);
                coop[k] = __ballot(1) == ~0ull && __all(abs(l[k] - ref) <= 30) != 0;   // the window needs all 64 lanes
                if (coop[k]) {
                    const int base = ref - 31;                                   // window [ref-31, ref+32]
                    const int lane = (int)(threadIdx.x & 63u);
                    const d2 mine = load_node<false>(ytab, min(max(base + lane, 0), g.n));
                    const int r = l[k] - base;                                   // 1 .. 61
                    n0[k].x = __shfl(mine.x, r, 64);
                    n0[k].y = __shfl(mine.y, r, 64);
                    const bool down = qs[k] < n0[k].x;
                    const int ro = down ? r - 1 : r + 1;
                    const double ox = __shfl(mine.x, ro, 64), oy = __shfl(mine.y, ro, 64);
                    nm[k].x = ox; nm[k].y = oy;
                    n1[k].x = ox; n1[k].y = oy;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NQ; ++k) coop[k] = false;
        }
#else
        bool coop[NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) coop[k] = false;
#endif
        if constexpr (WIN) {
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                if (coop[k]) continue;
                nm[k] = load_node<LDSY>(ytab, max(l[k] - 1, 0));
                n0[k] = load_node<LDSY>(ytab, l[k]);
                n1[k] = load_node<LDSY>(ytab, l[k] + 1);        // index n is the padding node
            }
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const bool down = qs[k] < n0[k].x;   // then l = G-1 >= 0 (G = 0 has X_0 <= q)
                const d2 a = down ? nm[k] : n0[k], b = down ? n0[k] : n1[k];
                out[k] = blend(a.x, a.y, b.x, b.y, qs[k]);
            }
        } else {
            // Unordered queries (region sweep, 1e8 queries on the jittered 1e6-node grid): three eager gathers 1.25 ms;
            // G and G+1 eager plus G-1 where the comparison with X_G asks for it 1.03 ms (shipped); G first and then one
            // dependent gather of G-1 or G+1 (-DMI_M3_TWO_LOOKUPS=1: 2 lookups per query instead of 2.5) 1.08 ms.
#if MI_M3_TWO_LOOKUPS
#pragma unroll
            for (int k = 0; k < NQ; ++k) n0[k] = load_node<LDSY>(ytab, l[k]);
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const bool down = qs[k] < n0[k].x;
                n1[k] = load_node<LDSY>(ytab, down ? max(l[k] - 1, 0) : l[k] + 1);
            }
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const bool down = qs[k] < n0[k].x;
                const d2 a = down ? n1[k] : n0[k], b = down ? n0[k] : n1[k];
                out[k] = blend(a.x, a.y, b.x, b.y, qs[k]);
            }
#else
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                n0[k] = load_node<LDSY>(ytab, l[k]);
                n1[k] = load_node<LDSY>(ytab, l[k] + 1);        // index n is the padding node
            }
#pragma unroll   // fetch node G-1 only where it is needed (one dependent gather, no loop)
            for (int k = 0; k < NQ; ++k) nm[k] = (qs[k] < n0[k].x) ? load_node<LDSY>(ytab, max(l[k] - 1, 0)) : n0[k];
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const bool down = qs[k] < n0[k].x;   // then l = G-1 >= 0 (G = 0 has X_0 <= q)
                const d2 a = down ? nm[k] : n0[k], b = down ? n0[k] : n1[k];
                out[k] = blend(a.x, a.y, b.x, b.y, qs[k]);
            }
#endif
        }
    } else {
        d2 n0[NQ], n1[NQ];
        if constexpr (MODE == 1) {
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                int i = (int)((qs[k] - g.xmin) * g.scale);
                l[k] = min(max(i, 0), g.n - 2);
            }
        } else {
            uint32_t lo[NQ], hi[NQ];
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                int b = (int)((qs[k] - g.xmin) * g.scale);
                b = min(max(b, 0), g.nb - 1);
                lo[k] = g.s[b];
                hi[k] = g.s[b + 1];
            }
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                uint32_t a = lo[k], c = hi[k];       // bracket index is in [a, c]
                while (c > a) {
                    const uint32_t mid = (a + c + 1u) >> 1;
                    if (g.nodes[mid].x <= qs[k]) a = mid; else c = mid - 1u;
                }
                l[k] = (int)a;
            }
        }
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            n0[k] = g.nodes[l[k]];
            n1[k] = g.nodes[l[k] + 1];
        }
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            if constexpr (MODE == 1) {
                int i = l[k];
                while (qs[k] < n0[k].x && i > 0) { --i; n1[k] = n0[k]; n0[k] = g.nodes[i]; }
                while (qs[k] >= n1[k].x && i < g.n - 1) { ++i; n0[k] = n1[k]; n1[k] = g.nodes[i + 1]; }
            }
            out[k] = blend(n0[k].x, n0[k].y, n1[k].x, n1[k].y, qs[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        if (oor[k]) out[k] = (q[k] != q[k]) ? __builtin_nan("") : extrap;
    }
}

template <int MODE, int NQ, int FORMULA = 0, bool WIN = false>
__device__ __forceinline__ void eval_batch(const G1Dev& g, const double (&q)[NQ], double (&out)[NQ], double extrap)
{
    eval_batch_from<MODE, NQ, FORMULA, WIN>(g, q, out, extrap, MODE == 0 ? g.y : reinterpret_cast<const double*>(g.nodes));
}

#ifndef MI_SWEEP_BINS
#define MI_SWEEP_BINS 256
#endif
constexpr int kSweepBins = MI_SWEEP_BINS;

__device__ __forceinline__ int sweep_bin(double q, double xmin, double bscale)
{
    const int b = (int)((q - xmin) * bscale);             // NaN -> 0, out of range clamps: any bin is correct
    return min(max(b, 0), kSweepBins - 1);
}

// Are the queries already ordered locally (sorted / clustered sets)?  1024 samples: a query and the one 4096
// positions later fall into the same or adjacent region; >= 75 % => ordered (streaming kernel), else unordered
// (region sweep).  One wave does it (16 samples per lane, no LDS): either as its own kernel, ahead of kernels gated
// on the verdict, or inside the interpolation kernel when the verdict is only wanted for the next call.
struct ProbeArgs {
    const double* xq;    // the whole query vector of the call
    size_t nq;           // >= 4098
    double xmin, bscale;
    int* flag;           // device-side verdict (may be null)
    int* host_mailbox;   // pinned host int, device view (null: no probe)
};

__device__ __forceinline__ void order_probe_wave(const ProbeArgs& p)
{
    const int lane = threadIdx.x & 63;
    const double step = (double)(p.nq - 4097) / 1024.0;
    unsigned near = 0;
#pragma unroll 8
    for (int k = 0; k < 16; ++k) {
        const size_t j = (size_t)(((double)(k * 64 + lane) + 0.5) * step);
        const int a = sweep_bin(p.xq[j], p.xmin, p.bscale), b = sweep_bin(p.xq[j + 4096], p.xmin, p.bscale);
        near += (abs(a - b) <= 1) ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) near += __shfl_xor(near, off, 64);
    if (lane == 0) {
        const int verdict = (near >= 768u) ? 1 : 0;
        if (p.flag) *p.flag = verdict;                                                            // gates this call's kernels
        __hip_atomic_store(p.host_mailbox, verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // predicts the next call's
    }
}

__global__ __launch_bounds__(64) void interp1_order_probe(ProbeArgs p) { order_probe_wave(p); }

// Vector kernel: a fixed VPL = 2 16-B vectors (four queries) per lane and one workgroup per 8 KiB of
// queries, no grid-stride loop.  Measured on MI355X (profiles/r01_exp_stream_shapes.log): this shape streams 8 B in +
// 8 B out per element at 6.5 TB/s, a grid capped at 2048 workgroups with a grid-stride loop at 5.0 TB/s.
// Non-temporal loads/stores: the streams must not evict the table from L2.  Requires xq, yq 16-B aligned.
#ifndef MI_INTERP1_VPL
#define MI_INTERP1_VPL 2     // 16-B vectors per lane: 1 / 2 / 4 measured 0.276 / 0.250 / 0.253 ms (sorted), random unchanged
#endif
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

// ---- region sweep: random queries over a table that does not fit L2 ------------------------------------
// A uniformly random gather is one L2 request per query, and with an 8 MB table ~40 % of them miss the 4 MiB L2
// of the XCD (DESIGN.md "Random queries").  The misses go away if, at any moment, the whole chip works on the same
// table region.  Persistent workgroups (all start together, all do the same work per tile, so they stay in step
// without synchronising) each take a tile of 16384 queries (128 KiB of LDS, one workgroup of 512 lanes per CU),
// order it by table region with an in-LDS counting sort (256 regions), and gather + blend in that order: lane j of
// round u holds sorted position j + 512u, so every wave of every CU is in about the same region at about the same
// time and L2 only has to hold that region.  Successive tiles sweep the regions up, down, up, ...: the half of the
// table touched last is still in L2 when the next tile starts there (gather misses 12.0 M -> 7.8 M per launch).
// Results overwrite the sorted LDS slot; each lane reads its own results back through the sorted positions it
// remembered and stores them coalesced, so the output order is untouched.  Arithmetic = eval_batch, identical to
// the streaming kernel.  Shapes measured in this kernel (1e8 queries, 1e6 nodes, before the up/down order):
// 256x32x2/CU 0.872 ms, 1024x16 0.823, 512x32 0.787 (64 regions), 0.770 (256), 0.797 (1024); with the up/down
// order 0.69 ms.  Phase times, the L2 request-rate ceiling (2.7e11 gathers/s chip-wide) and the overlap schemes
// that did not pay: DESIGN.md section 4, profiles/r01_exp_region_sweep_phases.log, r01_exp_gather_rate.log.
// Experiment hook (off unless MI_STREAM_LD / MI_STREAM_ST are given at build time, e.g. -DMI_STREAM_LD=2 -DMI_STREAM_ST=2;
// simple form and SCHED 0 of the pipelined form only): the query/result streams of the sweep kernels with explicit
// cache-policy bits -- 0 none, 1 nt, 2 sc1 nt, 3 sc0 sc1 nt, 4 sc0 sc1, 5 sc0 nt, 6 sc1 -- to see whether any policy keeps
// the streams from displacing the table in L2.  Result: profiles/r02_exp_stream_cache_policy.log
#if defined(MI_STREAM_LD) || defined(MI_STREAM_ST)
#define MI_STREAM_ASM 1
#ifndef MI_STREAM_LD
#define MI_STREAM_LD 1
#endif
#ifndef MI_STREAM_ST
#define MI_STREAM_ST 1
#endif
#define MI_POLICY_BITS_0 ""
#define MI_POLICY_BITS_1 "nt"
#define MI_POLICY_BITS_2 "sc1 nt"
#define MI_POLICY_BITS_3 "sc0 sc1 nt"
#define MI_POLICY_BITS_4 "sc0 sc1"
#define MI_POLICY_BITS_5 "sc0 nt"
#define MI_POLICY_BITS_6 "sc1"
#define MI_POLICY_CAT(a, b) a##b
#define MI_POLICY_BITS(n) MI_POLICY_CAT(MI_POLICY_BITS_, n)
#define MI_STREAM_LD_BITS MI_POLICY_BITS(MI_STREAM_LD)
#define MI_STREAM_ST_BITS MI_POLICY_BITS(MI_STREAM_ST)
__device__ __forceinline__ d2 stream_load(const d2* p)      // the caller waits (s_waitcnt vmcnt(0)) before the first use
{
    d2 v;
    asm volatile("global_load_dwordx4 %0, %1, off " MI_STREAM_LD_BITS : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void stream_store(d2 v, d2* p)
{
    asm volatile("global_store_dwordx4 %0, %1, off " MI_STREAM_ST_BITS "\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void stream_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#else
#define MI_STREAM_ASM 0
__device__ __forceinline__ d2 stream_load(const d2* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stream_store(d2 v, d2* p) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void stream_wait() {}
#endif

#ifndef MI_SWEEP_THREADS
#define MI_SWEEP_THREADS 512
#endif
#ifndef MI_SWEEP_K
#define MI_SWEEP_K 32
#endif
#ifndef MI_SWEEP_BLOCKS_PER_CU
#define MI_SWEEP_BLOCKS_PER_CU 1
#endif
#ifndef MI_SWEEP_DYNAMIC
#define MI_SWEEP_DYNAMIC 0   // hand the gather chunks out dynamically: measured 0.711-0.716 vs 0.707-0.716 ms static -- no gain, off
#endif
constexpr size_t kSweepMinTilesPerCu = 2;   // profiles/r02_strong_scaling_shards.log: at 3 tiles per CU (1.25e7 queries) the sweep still wins 0.097 vs 0.135 ms
#ifndef MI_SWEEP_M3_BATCH
#define MI_SWEEP_M3_BATCH 4  // mode 3, pipelined form: queries per lane whose gathers are in flight together; 8 does not
                             // fit the 128 registers of a 1024-lane workgroup (139-161 spilled: 1.59 ms against 1.03)
#endif
#ifndef MI_SWEEP_WIN
#define MI_SWEEP_WIN 0       // mode 3: fetch node G-1 with G and G+1 (three independent gathers) instead of on demand:
                             // 1.25 ms against 1.03 (profiles/r02_mode3_gather_variants.log)
#endif
constexpr bool kSweepWin = MI_SWEEP_WIN != 0;
constexpr int kSweepThreads = MI_SWEEP_THREADS;
constexpr int kSweepK = MI_SWEEP_K;                       // queries per lane per tile
constexpr int kSweepTile = kSweepThreads * kSweepK;       // queries per tile (8 B of LDS each)
template <int MODE, int FORMULA>
__global__ __launch_bounds__(kSweepThreads) void interp1_sweep_kernel(G1Dev g, const double* __restrict__ xq,
                                                                      double* __restrict__ yq, size_t ntiles,
                                                                      double extrap, double bscale,
                                                                      const int* __restrict__ order_flag,
                                                                      size_t tail, ProbeArgs probe)
{
    __shared__ double sq[kSweepTile];
    __shared__ unsigned hist[kSweepBins];
#if MI_SWEEP_DYNAMIC
    __shared__ unsigned next_chunk;          // gather phase: chunks of 256 sorted positions handed out to the waves
#endif
    const int tid = threadIdx.x;
    if (*order_flag != 0) return;            // queries already ordered locally: the streaming kernel does the work
    // The last workgroup has the fewest tiles: it also probes the query order for the next call (one wave, while
    // the others wait for their first tile) and evaluates the ragged tail after its tiles.
    const bool last_wg = blockIdx.x == gridDim.x - 1;
    if (probe.host_mailbox && last_wg && tid < 64) order_probe_wave(probe);
    bool rev = false;                        // regions are swept up, down, up, ...: L2 still holds the turn-around half
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x, rev = !rev) {
        const d2* q2 = reinterpret_cast<const d2*>(xq + t * kSweepTile);
        d2* o2 = reinterpret_cast<d2*>(yq + t * kSweepTile);
        double q[kSweepK];
#pragma unroll
        for (int u = 0; u < kSweepK / 2; ++u) {
            const d2 v = stream_load(q2 + tid + u * kSweepThreads);
            q[2 * u] = v.x;
            q[2 * u + 1] = v.y;
        }
        stream_wait();
        for (int b = tid; b < kSweepBins; b += kSweepThreads) hist[b] = 0;
#if MI_SWEEP_DYNAMIC
        if (tid == 0) next_chunk = 0;
#endif
        __syncthreads();
        unsigned short bin[kSweepK], rank[kSweepK];
#pragma unroll
        for (int u = 0; u < kSweepK; ++u) {
            const int b = sweep_bin(q[u], g.xmin, bscale);
            bin[u] = (unsigned short)b;
            rank[u] = (unsigned short)atomicAdd(&hist[b], 1u);
        }
        __syncthreads();
        if (tid < 64) {                                   // exclusive prefix over the regions (one wave, 64 at a time)
            unsigned run = 0;
#pragma unroll
            for (int base = 0; base < kSweepBins; base += 64) {
                const unsigned v = hist[base + tid];
                unsigned incl = v;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned o = __shfl_up(incl, off, 64);
                    if (tid >= off) incl += o;
                }
                hist[base + tid] = run + incl - v;
                run += __shfl(incl, 63, 64);
            }
        }
        __syncthreads();
        unsigned short sp[kSweepK];
#pragma unroll
        for (int u = 0; u < kSweepK; ++u) {
            sp[u] = (unsigned short)(hist[bin[u]] + rank[u]);
            sq[sp[u]] = q[u];
        }
        __syncthreads();
#if MI_SWEEP_DYNAMIC
        // The waves of a CU do not reach the end of the gather rounds together (2.4 us between the first and the last of
        // eight, profiles/r02_sweep_pipelined_phases.log).  Handing the sorted positions out in chunks of 256 (four per
        // lane), in order, to whichever wave is free does not shorten the phase (the memory path returns in order: the
        // wave that issued last finishes last whatever it was given): experiment kept for reference.
        for (;;) {
            unsigned c = 0;
            if ((tid & 63) == 0) c = atomicAdd(&next_chunk, 1u);
            c = __builtin_amdgcn_readfirstlane(c);
            if (c >= (unsigned)(kSweepTile / 256)) break;
            double qq[4], rr[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int p = (int)c * 256 + w * 64 + (tid & 63);
                qq[w] = sq[rev ? kSweepTile - 1 - p : p];
            }
            eval_batch<MODE, 4, FORMULA, kSweepWin>(g, qq, rr, extrap);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int p = (int)c * 256 + w * 64 + (tid & 63);
                sq[rev ? kSweepTile - 1 - p : p] = rr[w];
            }
        }
#else
#pragma unroll
        for (int u = 0; u < kSweepK; u += 4) {
            double qq[4], rr[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int p = tid + (u + w) * kSweepThreads;
                qq[w] = sq[rev ? kSweepTile - 1 - p : p];
            }
            eval_batch<MODE, 4, FORMULA, kSweepWin>(g, qq, rr, extrap);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int p = tid + (u + w) * kSweepThreads;
                sq[rev ? kSweepTile - 1 - p : p] = rr[w];
            }
        }
#endif
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kSweepK / 2; ++u) {
            d2 v;
            v.x = sq[sp[2 * u]];
            v.y = sq[sp[2 * u + 1]];
            stream_store(v, o2 + tid + u * kSweepThreads);
        }
        __syncthreads();   // the next tile's scatter reuses sq
    }
    if (tail && last_wg) {
        const double* tq = xq + ntiles * kSweepTile;
        double* to = yq + ntiles * kSweepTile;
        double q[kSweepK];                   // tail < one tile: all loads in flight at once, one latency
#pragma unroll
        for (int u = 0; u < kSweepK; ++u) {
            const size_t i = (size_t)tid + (size_t)u * kSweepThreads;
            q[u] = i < tail ? tq[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kSweepK; u += 4) {
            const double qq[4] = {q[u], q[u + 1], q[u + 2], q[u + 3]};
            double rr[4];
            eval_batch<MODE, 4, FORMULA, kSweepWin>(g, qq, rr, extrap);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const size_t i = (size_t)tid + (size_t)(u + w) * kSweepThreads;
                if (i < tail) to[i] = rr[w];
            }
        }
    }
}

// ---- region sweep, pipelined form (round 2) ----------------------------------------------------------------
// Same tiles, same in-LDS counting sort, same arithmetic as interp1_sweep_kernel, but the HBM streams of one tile run
// WHILE another tile gathers.  Measured (scripts/exp_mix.hip, profiles/r02_exp_mix.log): stream loads or stores issued
// by OTHER waves of the CU, a few wave-instructions at a time, cost the L2-hit gathers of the gathering waves about
// 12 % -- what hurt every overlap scheme of round 1 was the burst (all CUs loading 128 KiB at the same moment) and
// loads issued by the gathering waves themselves (vmcnt is in order within a wave).  So: one 1024-lane workgroup per
// CU, two groups of 8 waves that swap roles tile by tile.
//   gatherer of tile t : gather + blend rounds over the sorted LDS tile (8 rounds of 4 queries per lane), reads its
//                        results back, stores them (no wait) and issues the loads of tile t+2 into registers (no wait)
//   preparer of tile t+1: its queries arrived in registers during the previous step; region histogram (LDS atomics,
//                        its own histogram), prefix, sorted positions -- all while the other group gathers -- and the
//                        scatter into the LDS tile once the gatherer has read its results out.
// The one LDS tile (16 384 queries, 128 KiB) is the only hand-over point; six workgroup barriers per tile, none of
// which waits for vector memory.  Every workgroup runs the same schedule, so the chip still sweeps the table regions
// in step (that is what keeps the gathers in L2).
constexpr int kPipeGroup = 512;                          // lanes per group = kSweepTile / kSweepK
constexpr int kPipeThreads = 2 * kPipeGroup;
static_assert(kSweepTile == kPipeGroup * kSweepK, "one group covers a tile with kSweepK queries per lane");

// test hook (mi_debug_sweep_timing; compiled in with -DMI_PIPE_TIMING=1 only: the accumulators cost the pipelined kernel
// registers it does not have): when set, lane 0 of each group accumulates wall_clock64 ticks (100 MHz) per role and
// schedule interval into [workgroup][group][role: 0 gather, 1 prepare][interval 0..5]
#ifndef MI_PIPE_TIMING
#define MI_PIPE_TIMING 0
#endif
__device__ unsigned long long* g_pipe_timing = nullptr;

__device__ __forceinline__ void pipe_barrier()
{
    // LDS traffic of this wave done, then the workgroup barrier; vector memory stays in flight across it
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// two gather rounds (2 x 4 queries per lane) starting at sorted position first + lane, walking by `stride` (+512 sweeping
// the regions upwards, -512 downwards); kept as a rolled loop: the unrolled form of all 8 rounds x 2 roles x 2 groups
// overwhelms the register allocator at the 128 registers a 1024-lane workgroup leaves per lane
template <int MODE, int FORMULA>
__device__ __forceinline__ void pipe_gather_rounds(const G1Dev& g, double* sq, int first, int stride, double extrap)
{
    constexpr int B = (MODE == 3) ? MI_SWEEP_M3_BATCH : 4;
#pragma unroll 1
    for (int r = 0; r < 8 / B; ++r) {
        double qq[B], rr[B];
#pragma unroll
        for (int w = 0; w < B; ++w) qq[w] = sq[first + (B * r + w) * stride];
        eval_batch<MODE, B, FORMULA, kSweepWin>(g, qq, rr, extrap);
#pragma unroll
        for (int w = 0; w < B; ++w) sq[first + (B * r + w) * stride] = rr[w];
    }
}

// SCHED 0: the gatherer issues all sixteen loads of its next tile right behind its result stores (one burst per tile);
// SCHED 1: the preparer issues them itself at the start of its step, in two halves; SCHED 2: one vector at a time,
// spread over the first ~10 us of the other group's gather rounds.
// Three workgroup barriers per tile: after the gather rounds, after the read-back, after the scatter.  The preparing
// group orders its own histogram -> prefix -> positions passes with a counter in LDS that only its 8 waves touch, so
// the gathering waves run their 8 rounds without stopping (barriers inside the rounds cost 1.2 us each: every
// interval then ends with its slowest wave).
template <int MODE, int FORMULA, int SCHED>
__global__ __launch_bounds__(kPipeThreads) void interp1_sweep_pipe_kernel(G1Dev g, const double* __restrict__ xq,
                                                                          double* __restrict__ yq, size_t ntiles,
                                                                          double extrap, double bscale,
                                                                          const int* __restrict__ order_flag,
                                                                          size_t tail, ProbeArgs probe)
{
    __shared__ double sq[kSweepTile];
    __shared__ unsigned hist[2][kSweepBins];
    __shared__ unsigned gbar[2];
    if (*order_flag != 0) return;            // queries already ordered locally: the streaming kernel does the work
    if (threadIdx.x < 2) gbar[threadIdx.x] = 0;
    const int tid = threadIdx.x & (kPipeGroup - 1);
    const int grp = threadIdx.x >> 9;        // wave-uniform: waves 0-7 / 8-15
    const bool last_wg = blockIdx.x == gridDim.x - 1;
    if (probe.host_mailbox && last_wg && threadIdx.x >= kPipeThreads - 64) order_probe_wave(probe);   // for the next call
    const long nloc = ntiles > blockIdx.x ? (long)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
    double q[kSweepK];                       // preparer: the tile's queries; gatherer: its results on their way out
    unsigned sp2[kSweepK / 2];               // sorted positions of this group's tile, two per register
    auto load_part = [&](long it, int u0, int u1) {          // vectors u0..u1-1 of the 16 per lane
        const d2* q2 = reinterpret_cast<const d2*>(xq + ((size_t)blockIdx.x + (size_t)it * gridDim.x) * kSweepTile);
#pragma unroll
        for (int u = 0; u < kSweepK / 2; ++u) {
            if (u >= u0 && u < u1) {
                const d2 v = stream_load(q2 + tid + u * kPipeGroup);
                q[2 * u] = v.x;
                q[2 * u + 1] = v.y;
            }
        }
    };
    [[maybe_unused]] auto load_tile = [&](long it) { load_part(it, 0, kSweepK / 2); };
    auto store_tile = [&](long it) {
        d2* o2 = reinterpret_cast<d2*>(yq + ((size_t)blockIdx.x + (size_t)it * gridDim.x) * kSweepTile);
#pragma unroll
        for (int u = 0; u < kSweepK / 2; ++u) {
            d2 v;
            v.x = q[2 * u];
            v.y = q[2 * u + 1];
            stream_store(v, o2 + tid + u * kPipeGroup);
        }
    };
    for (int b = threadIdx.x; b < 2 * kSweepBins; b += kPipeThreads) (&hist[0][0])[b] = 0;
    if (SCHED == 0 && grp == 0 && nloc > 0) load_tile(0);
    pipe_barrier();
    unsigned* const myhist = hist[grp];
#if MI_PIPE_TIMING
    unsigned long long* const tdbg = g_pipe_timing;
    unsigned long long tacc[2][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};
    unsigned long long tlast = tdbg ? wall_clock64() : 0;
#define MI_PIPE_STAMP(role, slot)                                       \
    if (tdbg) {                                                         \
        const unsigned long long now_ = wall_clock64();                 \
        tacc[role][slot] += now_ - tlast;                               \
        tlast = now_;                                                   \
    }
#else
#define MI_PIPE_STAMP(role, slot)
#endif
    // barrier among the 8 waves of this group only: a monotonic arrival counter in LDS
    unsigned gb_target = 0;
    auto group_barrier = [&]() {
        gb_target += kPipeGroup / 64;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if ((threadIdx.x & 63) == 0) atomicAdd(&gbar[grp], 1u);
        while (*reinterpret_cast<volatile unsigned*>(&gbar[grp]) < gb_target) __builtin_amdgcn_s_sleep(2);
        asm volatile("" ::: "memory");
    };
    // One step of the schedule: in step `it` the owner of tile `it` (group it & 1) gathers it and the owner of tile
    // it+1 prepares it.  The two roles are separate code paths run in strict alternation by each group (group 0:
    // prepare, gather, prepare, ...; group 1: gather, prepare, gather, ...), so that the register allocator sees that
    // the 64 registers of a tile's queries are dead while its owner gathers.
    auto gather_step = [&](long it) {        // this group owns tile `it` (it = -1: nothing yet, barriers only)
        const bool act = it >= 0;
        // regions are swept up, down, up, ...: L2 still holds the turn-around half.  Lane j of round u handles sorted
        // position j + 512 u (up) or 16383 - j - 512 u (down).
        const bool rev = (it & 1) != 0;
        const int stride = rev ? -kPipeGroup : kPipeGroup;
        int first = rev ? kSweepTile - 1 - tid : tid;
        if (act) {
#pragma unroll 1
            for (int iv = 0; iv < 4; ++iv) { // eight rounds; the other group prepares its tile meanwhile
                pipe_gather_rounds<MODE, FORMULA>(g, sq, first, stride, extrap);
                first += 8 * stride;
            }
        }
        MI_PIPE_STAMP(0, 3)                   // own work: the eight rounds
        pipe_barrier();
        MI_PIPE_STAMP(0, 0)                   // waiting for the preparer
        if (act) {                           // results out of the tile (own queries: positions remembered in sp2)
#pragma unroll
            for (int u = 0; u < kSweepK; u += 2) {
                q[u] = sq[sp2[u / 2] & 0xffffu];
                q[u + 1] = sq[sp2[u / 2] >> 16];
                if ((u & 6) == 6) __builtin_amdgcn_sched_barrier(0);   // eight at a time: bounded register pressure
            }
        } else {
#pragma unroll
            for (int u = 0; u < kSweepK; ++u) q[u] = 0.0;   // explicit definition on every path: q is dead during the rounds
        }
        pipe_barrier();
        MI_PIPE_STAMP(0, 1)                   // ... the other group's tile goes in meanwhile
        if (act) store_tile(it);             // results to HBM; nothing waited for
        // Stores and loads are issued HERE, while nobody gathers (the other group scatters into LDS): measured against
        // leaving the last quarter / eighth of the loads (0.682 / 0.671 ms vs 0.672 ms) or all stores and loads (0.712 vs
        // 0.665 ms) to the start of this group's prepare step, where they would run beside the other group's gather rounds
        if (SCHED == 0 && it + 2 < nloc) {   // (it = -1: group 1's first tile)
            load_tile(it + 2);
        } else {
#pragma unroll
            for (int u = 0; u < kSweepK; ++u) q[u] = 0.0;
        }
        pipe_barrier();
        MI_PIPE_STAMP(0, 2)
    };
    auto prep_step = [&](long it) {          // this group owns tile it+1 (past the last tile: barriers only)
        const bool act = it + 1 < nloc;
        unsigned rank2[kSweepK / 2];         // rank inside the region (histogram ticket), two per register
#pragma unroll
        for (int u = 0; u < kSweepK / 2; ++u) rank2[u] = 0;
        stream_wait();
        if (act) {
            if (SCHED == 1) {
                load_part(it + 1, 0, kSweepK / 4);
                __builtin_amdgcn_s_sleep(100);               // ~2.7 us
                load_part(it + 1, kSweepK / 4, kSweepK / 2);
            } else if (SCHED == 2) {
#pragma unroll
                for (int u = 0; u < kSweepK / 2; ++u) {
                    load_part(it + 1, u, u + 1);
                    __builtin_amdgcn_s_sleep(24);            // ~0.65 us apart: the tile arrives over ~10 us
                }
            }
            // (also measured: one wave's sixteen loads at a time, the eight waves 1.1 / 1.5 / 2.1 us apart -- the loads then
            // land 10+ us after they were issued, behind the gather requests queued on the same path: 0.78-0.82 ms)
            MI_PIPE_STAMP(1, 3)               // loads issued
            // region histogram (own histogram, cleared in the previous step)
#pragma unroll
            for (int u = 0; u < kSweepK; u += 2) {
                const unsigned r0 = atomicAdd(&myhist[sweep_bin(q[u], g.xmin, bscale)], 1u);
                const unsigned r1 = atomicAdd(&myhist[sweep_bin(q[u + 1], g.xmin, bscale)], 1u);
                rank2[u / 2] = r0 | (r1 << 16);
                if ((u & 6) == 6) __builtin_amdgcn_sched_barrier(0);
            }
            group_barrier();
            if (tid < 64) {                  // exclusive prefix over the regions (one wave, 64 at a time)
                unsigned run = 0;
#pragma unroll
                for (int base = 0; base < kSweepBins; base += 64) {
                    const unsigned v = myhist[base + tid];
                    unsigned incl = v;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const unsigned o = __shfl_up(incl, off, 64);
                        if (tid >= off) incl += o;
                    }
                    myhist[base + tid] = run + incl - v;
                    run += __shfl(incl, 63, 64);
                }
            }
            group_barrier();
#pragma unroll
            for (int u = 0; u < kSweepK; u += 2) {   // sorted positions
                // the region is recomputed from the query (three instructions) rather than kept: handed through an
                // empty asm so that the compiler does not keep the 32 fp64 products of the histogram pass alive
                double qa = q[u], qb = q[u + 1];
                asm volatile("" : "+v"(qa), "+v"(qb));
                const unsigned p0 = myhist[sweep_bin(qa, g.xmin, bscale)] + (rank2[u / 2] & 0xffffu);
                const unsigned p1 = myhist[sweep_bin(qb, g.xmin, bscale)] + (rank2[u / 2] >> 16);
                sp2[u / 2] = p0 | (p1 << 16);
                if ((u & 6) == 6) __builtin_amdgcn_sched_barrier(0);
            }
            MI_PIPE_STAMP(1, 5)               // sorted
        } else {
#pragma unroll
            for (int u = 0; u < kSweepK / 2; ++u) sp2[u] = 0;
        }
        pipe_barrier();                      // the gather rounds of the other group are over
        MI_PIPE_STAMP(1, 0)                   // waiting for the gatherer
        if (act) {
            for (int b = tid; b < kSweepBins; b += kPipeGroup) myhist[b] = 0;   // every lane read its region bases before the barrier
        }
        pipe_barrier();                      // (the gatherer has taken its results out of the tile)
        MI_PIPE_STAMP(1, 1)
        if (act) {                           // this group's tile goes in
#pragma unroll
            for (int u = 0; u < kSweepK; u += 2) {
                sq[sp2[u / 2] & 0xffffu] = q[u];
                sq[sp2[u / 2] >> 16] = q[u + 1];
                if ((u & 6) == 6) __builtin_amdgcn_sched_barrier(0);
            }
        }
        pipe_barrier();
        MI_PIPE_STAMP(1, 2)
    };
    if (grp == 0) {
        for (long it = -1;;) {
            prep_step(it);
            if (++it >= nloc) break;
            gather_step(it);
            if (++it >= nloc) break;
        }
    } else {
        for (long it = -1;;) {
            gather_step(it);
            if (++it >= nloc) break;
            prep_step(it);
            if (++it >= nloc) break;
        }
    }
#undef MI_PIPE_STAMP
#if MI_PIPE_TIMING
    if (tdbg && tid == 0) {
        for (int r = 0; r < 2; ++r)
            for (int k = 0; k < 6; ++k) tdbg[((size_t)blockIdx.x * 2 + grp) * 12 + r * 6 + k] = tacc[r][k];
    }
#endif
    if (tail && last_wg && grp == 0) {       // ragged tail (< one tile), four queries per lane at a time
        const double* tq = xq + ntiles * kSweepTile;
        double* to = yq + ntiles * kSweepTile;
#pragma unroll 1
        for (int u = 0; u < kSweepK; u += 4) {
            double qq[4], rr[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const size_t i = (size_t)tid + (size_t)(u + w) * kPipeGroup;
                qq[w] = i < tail ? tq[i] : 0.0;
            }
            eval_batch<MODE, 4, FORMULA, kSweepWin>(g, qq, rr, extrap);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const size_t i = (size_t)tid + (size_t)(u + w) * kPipeGroup;
                if (i < tail) to[i] = rr[w];
            }
        }
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

// Gated launches (order_flag != nullptr) exit at once when the probe saw unordered queries; what that costs is the
// dispatch of the grid's waves (26-30 us at 1e8 queries; 1024-lane workgroups cost the same, 4x fewer waves save
// 15-19 us but run 15 % slower on ordered input), which is why launch_mode only gates while it has no prediction.
#ifndef MI_INTERP1_GATED_BLOCK
#define MI_INTERP1_GATED_BLOCK 256
#endif
#ifndef MI_INTERP1_GATED_VPL
#define MI_INTERP1_GATED_VPL 2
#endif
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
        return launch_vec_shape<MODE, FORMULA, MI_INTERP1_GATED_BLOCK, MI_INTERP1_GATED_VPL>(ctx, d, xq, yq, nq, extrap, order_flag, probe);
    return launch_vec_shape<MODE, FORMULA, kBlock, MI_INTERP1_VPL>(ctx, d, xq, yq, nq, extrap, nullptr, probe);
}

// MI_SWEEP_VARIANT = 2 (default): the pipelined form (two wave groups per CU swapping roles: 0.657-0.672 ms per 1e8
// queries, 7 % less than form 1 on the same box in every run, profiles/r02_sweep_pipelined_phases.log);
// 1: one 512-lane workgroup per CU, the phases of a tile one after the other (0.710-0.716 ms).  Read once.
inline int sweep_variant()
{
    static const int v = [] {
        const char* e = getenv("MI_SWEEP_VARIANT");
        const int x = e ? atoi(e) : 2;
        return (x == 1 || x == 2) ? x : 2;
    }();
    return v;
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
    if (const char* env = getenv("MI_SWEEP_MIN_BYTES")) size_ok = table_bytes >= (size_t)strtoull(env, nullptr, 10);   // tuning hook
    // enough tiles for every CU to run a few sweeps (MI_SWEEP_MIN_TILES_PER_CU: tuning hook behind kSweepMinTilesPerCu,
    // profiles/r02_strong_scaling_shards.log)
    static const size_t min_tiles_per_cu = [] { const char* e = getenv("MI_SWEEP_MIN_TILES_PER_CU"); return e ? (size_t)strtoull(e, nullptr, 10) : kSweepMinTilesPerCu; }();
    const bool sweep_ok = ctx->query_order != MI_QUERIES_ORDERED && size_ok &&
                          ntiles >= (size_t)cus * min_tiles_per_cu && std::isfinite(d.xmax - d.xmin) && (d.xmax - d.xmin) > 0.0;
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
    const unsigned grid = (unsigned)std::min<size_t>(ntiles, (size_t)cus * MI_SWEEP_BLOCKS_PER_CU);   // persistent
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
    static const bool force_pipe = getenv("MI_SWEEP_VARIANT") != nullptr;
    if (sweep_variant() == 2 && (force_pipe || ntiles >= (size_t)cus * 16)) {
        const unsigned pgrid = (unsigned)std::min<size_t>(ntiles, (size_t)cus);   // one 1024-lane workgroup per CU
        static const int sched = [] { const char* e = getenv("MI_SWEEP_SCHED"); return e ? atoi(e) : 0; }();   // A-B hook
        if (sched == 1)
            hipLaunchKernelGGL((interp1_sweep_pipe_kernel<MODE, FORMULA, 1>), dim3(pgrid), dim3(kPipeThreads), 0, ctx->stream, d, xq, yq,
                               ntiles, extrap, bscale, flags, nq - head, probe);
        else if (sched == 2)
            hipLaunchKernelGGL((interp1_sweep_pipe_kernel<MODE, FORMULA, 2>), dim3(pgrid), dim3(kPipeThreads), 0, ctx->stream, d, xq, yq,
                               ntiles, extrap, bscale, flags, nq - head, probe);
        else
        hipLaunchKernelGGL((interp1_sweep_pipe_kernel<MODE, FORMULA, 0>), dim3(pgrid), dim3(kPipeThreads), 0, ctx->stream, d, xq, yq,
                           ntiles, extrap, bscale, flags, nq - head, probe);
        MI_LAUNCH_CHECK(ctx, "interp1 pipelined region-sweep kernel");
        return MI_OK;
    }
    hipLaunchKernelGGL((interp1_sweep_kernel<MODE, FORMULA>), dim3(grid), dim3(kSweepThreads), 0, ctx->stream, d, xq, yq,
                       ntiles, extrap, bscale, flags, nq - head, probe);
    MI_LAUNCH_CHECK(ctx, "interp1 region-sweep kernel");
    return MI_OK;
}

// ---- host-side table construction -----------------------------------------


mi_status upload(mi_ctx* ctx, void** dev, const void* host, size_t bytes)
{
    hipError_t e = hipMalloc(dev, bytes);
    if (e != hipSuccess) return mi::fail(ctx, MI_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    MI_HIP(ctx, hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
    return MI_OK;
}

inline double host_unode(int formula, double x0, double dx, double span, double den, size_t i)
{
    if (formula == 0) return std::fma((double)i, dx, x0);
    if (formula == 1) return x0 + (double)i * dx;
    if (formula == 2) return x0 + span * ((double)i / den);
    const double rden = 1.0 / den, q0 = (double)i * rden;
    return x0 + span * std::fma(std::fma(-den, q0, (double)i), rden, q0);
}

// Does a closed form reproduce EVERY node of xs bit for bit (the last one may be pinned)?  Fills d on success.
bool detect_closed_form(const std::vector<double>& xs, G1Dev* d)
{
    const size_t n = xs.size();
    if (n < 3) return false;
    const double x0 = xs[0], xl = xs[n - 1], den = (double)(n - 1), span = xl - x0;
    const double dx_cands[2] = {span / den, xs[1] - xs[0]};
    const int order[4] = {0, 1, 3, 2};   // cheapest evaluation first
    for (int fi = 0; fi < 4; ++fi) {
        const int formula = order[fi];
        for (int c = 0; c < (formula >= 2 ? 1 : 2); ++c) {
            const double dx = dx_cands[c];
            if (!(dx > 0.0) || !std::isfinite(dx)) continue;
            bool ok = true;
            for (size_t i = 0; i + 1 < n && ok; ++i) ok = host_unode(formula, x0, dx, span, den, i) == xs[i];
            if (!ok) continue;
            const bool last_ok = host_unode(formula, x0, dx, span, den, n - 1) == xl;
            // the closed-form abscissae must be strictly increasing up to the (possibly pinned) last node
            if (!last_ok && !(host_unode(formula, x0, dx, span, den, n - 2) < xl)) continue;
            d->x0 = x0;
            d->dx = dx;
            d->span = span;
            d->den = den;
            d->rden = 1.0 / den;
            d->formula = formula;
            d->pin_last = last_ok ? 0 : 1;
            d->xmin = x0;
            d->xmax = xl;
            d->scale = 1.0 / dx;
            return true;
        }
    }
    return false;
}

// xs strictly increasing, finite, n >= 2
mi_status build_explicit(mi_ctx* ctx, const std::vector<double>& xs, const std::vector<double>& ys, mi_grid1** out)
{
    const size_t n = xs.size();
    if (n > 0x7ffffff0u) return mi::fail(ctx, MI_ERR_INVALID_ARG, "mi_grid1_create: n=%zu exceeds 2^31", n);
    mi_grid1* g = new (std::nothrow) mi_grid1();
    if (!g) return mi::fail(ctx, MI_ERR_NOMEM, "mi_grid1_create: out of host memory");
    g->ctx = ctx;
    g->device = ctx->device;
    g->n = n;
    g->dev_nodes = g->dev_s = nullptr;
    {   // linspace-like explicit grid: keep only Y, recompute X in registers (bit-exact by construction)
        G1Dev cf;
        memset(&cf, 0, sizeof(cf));
        if (detect_closed_form(xs, &cf)) {
            std::vector<double> yp(ys);
            yp.push_back(ys[n - 1]);
            mi_status st = upload(ctx, &g->dev_nodes, yp.data(), (n + 1) * sizeof(double));
            if (st != MI_OK) { delete g; return st; }
            cf.y = (const double*)g->dev_nodes;
            cf.n = (int)n;
            g->d = cf;
            g->mode = 0;
            g->table_bytes = (n + 1) * sizeof(double);
            *out = g;
            return MI_OK;
        }
    }
    std::vector<d2> nodes(n + 1);
    for (size_t i = 0; i < n; ++i) {
        nodes[i].x = xs[i];
        nodes[i].y = ys[i];
    }
    nodes[n] = nodes[n - 1];   // padding node: r = min(l+1, n-1)
    G1Dev& d = g->d;
    memset(&d, 0, sizeof(d));
    d.n = (int)n;
    d.xmin = xs[0];
    d.xmax = xs[n - 1];
    const double span = d.xmax - d.xmin;

    // Mode-1 test.  g(.) is monotone, so for a query q with bracket i
    // (X[i] <= q < X[i+1]):  g(X[i]) <= g(q) <= g(X[i+1]).  With
    // e_i = g(X[i]) - i in [e_lo, e_hi] for every node, the bracket lies in
    // [g(q) - 1 - e_hi, g(q) - e_lo]: at most e_hi - e_lo + 1 walk steps.
    const double scale1 = (double)(n - 1) / span;
    long e_lo = 0, e_hi = 0;
    bool finite_scale = std::isfinite(scale1) && scale1 > 0.0;
    if (finite_scale) {
        for (size_t i = 0; i < n; ++i) {
            const double t = (xs[i] - d.xmin) * scale1;
            long gi = (t >= 2147483000.0) ? 2147483000L : (long)(int)t;
            gi = std::min<long>(std::max<long>(gi, 0), (long)n - 2);
            const long e = gi - (long)i;
            e_lo = std::min(e_lo, e);
            e_hi = std::max(e_hi, e);
        }
    }
    // the clamp to n-2 makes e = -1 at the last node; harmless (walk-up handles it)
    if (finite_scale && (e_hi - e_lo + 1) <= kMaxWalk) {
        g->mode = 1;
        d.scale = scale1;
        // Centred guess: shift the origin so that every node's scaled abscissa sits inside its own unit cell,
        // t_i = (X_i - gorg) * scale in [i + m, i + 1 - m].  Holds whenever the grid deviates from a straight line by
        // less than one cell (jittered / mildly stretched grids).  Accepted only if the device expression gives
        // exactly i at every node; then eval_batch<3> needs no walk (see there).
        // Two candidate slopes: through the end nodes, and the least-squares line (end nodes of a jittered grid are
        // themselves jittered, which tilts the first one by up to a cell over the length of the table).
        double mi_ = 0.0, mx = 0.0;
        for (size_t i = 0; i < n; ++i) { mi_ += (double)i; mx += xs[i] - d.xmin; }
        mi_ /= (double)n;
        mx /= (double)n;
        double sxy = 0.0, sxx = 0.0;
        for (size_t i = 0; i < n; ++i) {
            const double di = (double)i - mi_;
            sxy += di * ((xs[i] - d.xmin) - mx);
            sxx += di * di;
        }
        const double cand[2] = {scale1, (sxy > 0.0) ? sxx / sxy : 0.0};
        for (int c = 0; c < 2 && !d.centred; ++c) {
            const double sc = cand[c];
            if (!(sc > 0.0) || !std::isfinite(sc) || n >= 0x7fffff00u) continue;
            double dlo = INFINITY, dhi = -INFINITY;
            for (size_t i = 0; i < n; ++i) {
                const double di = (xs[i] - d.xmin) * sc - (double)i;
                dlo = std::min(dlo, di);
                dhi = std::max(dhi, di);
            }
            const double m = 0.5 * (1.0 - (dhi - dlo));
            if (!(m > 1e-6)) continue;
            const double gorg = d.xmin + (dlo - m) / sc;
            bool ok = std::isfinite(gorg);
            for (size_t i = 0; i < n && ok; ++i) {
                const double t = (xs[i] - gorg) * sc;
                ok = t >= 0.0 && t < 2147483000.0 && (size_t)(int)t == i;
            }
            if (ok) {
                d.gorg = gorg;
                d.scale = sc;
                d.centred = 1;
            }
        }
    } else {
        g->mode = 2;
        size_t nb = n;
        double bscale = (double)nb / span;
        if (!std::isfinite(bscale) || !(bscale > 0.0)) {   // span underflow/overflow: one bucket
            nb = 1;
            bscale = 0.0;
        }
        d.nb = (int)nb;
        d.scale = bscale;
        // s[b] = largest node whose bucket is < b (0 if none); s[nb] = n-1
        std::vector<uint32_t> s(nb + 1, 0);
        auto bucket = [&](double x) {
            const double t = (x - d.xmin) * bscale;
            long b = (t >= 2147483000.0) ? 2147483000L : (long)(int)t;
            return (size_t)std::min<long>(std::max<long>(b, 0), (long)nb - 1);
        };
        size_t bprev = bucket(xs[0]);   // == 0
        for (size_t i = 1; i < n; ++i) {
            const size_t bi = bucket(xs[i]);
            for (size_t b = bprev + 1; b <= bi; ++b) s[b] = (uint32_t)(i - 1);
            bprev = bi;
        }
        for (size_t b = bprev + 1; b <= nb; ++b) s[b] = (uint32_t)(n - 1);
        mi_status st = upload(ctx, &g->dev_s, s.data(), (nb + 1) * sizeof(uint32_t));
        if (st != MI_OK) { delete g; return st; }
        d.s = (const uint32_t*)g->dev_s;
    }
    mi_status st = upload(ctx, &g->dev_nodes, nodes.data(), (n + 1) * sizeof(d2));
    if (st != MI_OK) {
        if (g->dev_s) (void)hipFree(g->dev_s);
        delete g;
        return st;
    }
    d.nodes = (const d2*)g->dev_nodes;
    g->table_bytes = (n + 1) * sizeof(d2) + (g->mode == 2 ? ((size_t)d.nb + 1) * 4 : 0);
    *out = g;
    return MI_OK;
}

mi_status fetch(mi_ctx* ctx, const double* p, size_t n, bool dev, std::vector<double>& v)
{
    v.resize(n);
    if (dev) {
        MI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        MI_HIP(ctx, hipMemcpy(v.data(), p, n * sizeof(double), hipMemcpyDeviceToHost));
    } else {
        memcpy(v.data(), p, n * sizeof(double));
    }
    return MI_OK;
}

}  // namespace

extern "C" {

mi_status mi_grid1_create(mi_ctx* ctx, const double* x, const double* y, size_t n, unsigned flags, mi_grid1** out)
{
    MI_REQUIRE(ctx, ctx && x && y && out, "mi_grid1_create: NULL argument");
    MI_REQUIRE(ctx, (flags & ~(MI_GRID_SANITISE | MI_GRID_DEVICE_PTRS)) == 0, "mi_grid1_create: unknown flags 0x%x", flags);
    *out = nullptr;
    if (n < 2) return mi::fail(ctx, MI_ERR_GRID, "mi_grid1_create: X must have at least two elements (n=%zu)", n);
    MI_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<double> xs, ys;
    mi_status st = fetch(ctx, x, n, flags & MI_GRID_DEVICE_PTRS, xs);
    if (st != MI_OK) return st;
    st = fetch(ctx, y, n, flags & MI_GRID_DEVICE_PTRS, ys);
    if (st != MI_OK) return st;
    for (size_t i = 0; i < n; ++i)
        if (!std::isfinite(xs[i]))
            return mi::fail(ctx, MI_ERR_GRID, "mi_grid1_create: X[%zu] is not finite", i);
    bool increasing = true;
    for (size_t i = 1; i < n && increasing; ++i) increasing = xs[i - 1] < xs[i];
    if ((flags & MI_GRID_SANITISE) && !increasing) {
        // arma::interp1 front end: unique + ascending sort of X, Y permuted
        // alike (first occurrence of a duplicate abscissa is kept).  A grid that is
        // already strictly increasing (the usual case) skips the 40 ms sort of 1e6 nodes.
        std::vector<size_t> idx(n);
        for (size_t i = 0; i < n; ++i) idx[i] = i;
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return xs[a] < xs[b]; });
        std::vector<double> x2, y2;
        x2.reserve(n);
        y2.reserve(n);
        for (size_t k = 0; k < n; ++k) {
            if (!x2.empty() && xs[idx[k]] == x2.back()) continue;
            x2.push_back(xs[idx[k]]);
            y2.push_back(ys[idx[k]]);
        }
        xs.swap(x2);
        ys.swap(y2);
        if (xs.size() < 2)
            return mi::fail(ctx, MI_ERR_GRID, "mi_grid1_create: X must have at least two unique elements");
    } else if (!increasing) {
        for (size_t i = 1; i < n; ++i)
            if (!(xs[i - 1] < xs[i]))
                return mi::fail(ctx, MI_ERR_GRID,
                                "mi_grid1_create: X not strictly increasing at %zu (pass MI_GRID_SANITISE)", i);
    }
    return build_explicit(ctx, xs, ys, out);
}

mi_status mi_grid1_create_uniform(mi_ctx* ctx, double x0, double dx, const double* y, size_t n, unsigned flags,
                                  mi_grid1** out)
{
    MI_REQUIRE(ctx, ctx && y && out, "mi_grid1_create_uniform: NULL argument");
    MI_REQUIRE(ctx, (flags & ~MI_GRID_DEVICE_PTRS) == 0, "mi_grid1_create_uniform: unknown flags 0x%x", flags);
    *out = nullptr;
    if (n < 2) return mi::fail(ctx, MI_ERR_GRID, "mi_grid1_create_uniform: need at least two nodes (n=%zu)", n);
    if (n > 0x7ffffff0u) return mi::fail(ctx, MI_ERR_INVALID_ARG, "mi_grid1_create_uniform: n=%zu exceeds 2^31", n);
    if (!(dx > 0.0) || !std::isfinite(dx) || !std::isfinite(x0) || !std::isfinite(std::fma((double)(n - 1), dx, x0)))
        return mi::fail(ctx, MI_ERR_GRID, "mi_grid1_create_uniform: need finite x0 and dx > 0");
    if (!(std::fma(1.0, dx, x0) > x0))
        return mi::fail(ctx, MI_ERR_GRID, "mi_grid1_create_uniform: dx too small relative to x0");
    MI_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<double> ys;
    mi_status st = fetch(ctx, y, n, flags & MI_GRID_DEVICE_PTRS, ys);
    if (st != MI_OK) return st;
    ys.push_back(ys[n - 1]);
    mi_grid1* g = new (std::nothrow) mi_grid1();
    if (!g) return mi::fail(ctx, MI_ERR_NOMEM, "mi_grid1_create_uniform: out of host memory");
    g->ctx = ctx;
    g->device = ctx->device;
    g->mode = 0;
    g->n = n;
    g->dev_s = nullptr;
    st = upload(ctx, &g->dev_nodes, ys.data(), (n + 1) * sizeof(double));
    if (st != MI_OK) { delete g; return st; }
    G1Dev& d = g->d;
    memset(&d, 0, sizeof(d));
    d.y = (const double*)g->dev_nodes;
    d.n = (int)n;
    d.x0 = x0;
    d.dx = dx;
    d.xmin = x0;
    d.xmax = std::fma((double)(n - 1), dx, x0);
    d.scale = 1.0 / dx;
    g->table_bytes = (n + 1) * sizeof(double);
    *out = g;
    return MI_OK;
}

mi_status mi_grid1_destroy(mi_grid1* g)
{
    if (!g) return MI_OK;
    (void)hipSetDevice(g->device);
    if (g->dev_nodes) (void)hipFree(g->dev_nodes);
    if (g->dev_s) (void)hipFree(g->dev_s);
    delete g;
    return MI_OK;
}

mi_status mi_debug_sweep_timing(mi_ctx* ctx, unsigned long long* ticks_dev)
{
    MI_REQUIRE(ctx, ctx != nullptr, "mi_debug_sweep_timing: ctx is NULL");
    MI_HIP(ctx, hipSetDevice(ctx->device));
    MI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MI_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_pipe_timing), &ticks_dev, sizeof(ticks_dev)));
    return MI_OK;
}

mi_status mi_grid1_info(const mi_grid1* g, size_t* n_nodes, int* mode, size_t* table_bytes)
{
    MI_REQUIRE(nullptr, g != nullptr, "mi_grid1_info: grid is NULL");
    if (n_nodes) *n_nodes = g->n;
    if (mode) *mode = (g->mode == 1 && g->d.centred) ? 3 : g->mode;
    if (table_bytes) *table_bytes = g->table_bytes;
    return MI_OK;
}

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
    // PCIe carries both directions at once (scripts/exp_pcie.hip: 0.8 GB each way 14 + 14 ms one after the other,
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
