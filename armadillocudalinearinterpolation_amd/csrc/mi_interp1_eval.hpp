// 1-D table interpolation on MI355X (gfx950): table layouts and the gather-and-blend arithmetic every interp1 kernel
// shares (eval_batch_from), so that the kernels' results are bit-identical.  Included by mi_interp1.hip (kernels and
// dispatch) and mi_interp1_tables.hip (host-side table construction).
//
// Semantics: include/mi355_interp.h ("fp64 blend") == oracle/interp_oracle.c.  Every TU that includes this is compiled
// with -ffp-contract=off so that every product and sum of the blend rounds separately, exactly like the oracle.
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
#pragma once
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

namespace mi_interp1 {

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
        if constexpr (WIN) {
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
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
            // dependent gather of G-1 or G+1 (2 lookups per query instead of 2.5) 1.08 ms (profiles/r02_mode3_gather_variants.log).
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

constexpr int kSweepBins = 256;   // table regions of the sweep's counting sort (64 / 256 / 1024 measured: 0.787 / 0.770 / 0.797 ms)

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

}  // namespace mi_interp1
