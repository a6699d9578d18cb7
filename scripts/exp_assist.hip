// Exploration harness (NOT product code), round 3: SCALAR-path assist for the region sweep's gather rounds.
// profiles/r03_exp_scalar_beside_vector_gathers.log: scalar gathers (s_load_dwordx4 through the scalar data cache) by some
// waves of a CU ADD to what its vector memory path gathers (1.03 + 0.21 per ns per CU).  In the pipelined sweep the
// preparing group idles for ~10 us per tile after its sort; here it evaluates the last NA rounds (512 sorted positions
// each) of the tile the other group is gathering: index and abscissae on the vector ALU (64 queries at a time), the
// 16-B node pair of every query by a scalar load (NF in flight per wave, address out of the lane with v_readlane, data
// back into the lane with four v_writelane), blend on the vector ALU.  Same arithmetic as eval_batch_from<0>; the driver
// compares the outputs with the product kernel's bit for bit.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mi_interp1_sweep.hpp"

using namespace mi_interp1;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned u4v __attribute__((ext_vector_type(4)));

// one query per lane, mode 0 (closed-form abscissae): eval_batch_from<0, 1, FORMULA> with the node pair fetched by scalar loads
template <int FORMULA, int NF>
__device__ __forceinline__ double scalar_eval(const G1Dev& g, double q, double extrap)
{
    const bool oor = !(q >= g.xmin && q <= g.xmax);
    const double qs = oor ? g.xmin : q;
    int i = (int)((qs - g.x0) * g.scale);
    i = min(max(i, 0), g.n - 1);
    double a = unode<FORMULA>(g, i), b = unode<FORMULA>(g, min(i + 1, g.n - 1));
    while (i > 0 && a > qs) { --i; b = a; a = unode<FORMULA>(g, i); }
    while (i < g.n - 1 && b <= qs) { ++i; a = b; b = unode<FORMULA>(g, min(i + 1, g.n - 1)); }
    const unsigned off = (unsigned)i * 8u;                     // byte offset of Y[i] (tables below 4 GiB)
    int y0 = 0, y1 = 0, y2 = 0, y3 = 0;                        // {Y[i], Y[i+1]} as four dwords of this lane
#pragma unroll
    for (int l0 = 0; l0 < 64; l0 += NF) {
        u4v v[NF];
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            const unsigned o = __builtin_amdgcn_readlane(off, l0 + k);
            asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(v[k]) : "s"(g.y), "s"(o) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            const int ln = l0 + k;
            asm volatile("s_mov_b32 m0, %8\n\tv_writelane_b32 %0, %4, m0\n\tv_writelane_b32 %1, %5, m0\n\tv_writelane_b32 %2, %6, m0\n\tv_writelane_b32 %3, %7, m0"
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "s"(v[k].x), "s"(v[k].y), "s"(v[k].z), "s"(v[k].w), "s"(ln) : "m0");
        }
    }
    const double ya = __hiloint2double(y1, y0), yb = __hiloint2double(y3, y2);
    double r = blend(a, ya, b, yb, qs);
    if (oor) r = (q != q) ? __builtin_nan("") : extrap;
    return r;
}

template <int MODE, int FORMULA, int NA, int NF>
__global__ __launch_bounds__(kPipeThreads) void assist_kernel(G1Dev g, const double* __restrict__ xq,
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
    auto load_tile = [&](long it) {                          // the 16 vectors per lane of this group's next tile
        const d2* q2 = reinterpret_cast<const d2*>(xq + ((size_t)blockIdx.x + (size_t)it * gridDim.x) * kSweepTile);
#pragma unroll
        for (int u = 0; u < kSweepK / 2; ++u) {
            const d2 v = stream_load(q2 + tid + u * kPipeGroup);
            q[2 * u] = v.x;
            q[2 * u + 1] = v.y;
        }
    };
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
    if (grp == 0 && nloc > 0) load_tile(0);
    pipe_barrier();
    unsigned* const myhist = hist[grp];
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
            for (int bq = 0; bq < (kSweepK - NA) / 4; ++bq) {   // rounds 0 .. 31-NA in batches of four; the last NA rounds are the other group's (scalar gathers)
                double qq[4], rr[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) qq[w] = sq[first + w * stride];
                eval_batch<MODE, 4, FORMULA, kSweepWin>(g, qq, rr, extrap);
#pragma unroll
                for (int w = 0; w < 4; ++w) sq[first + w * stride] = rr[w];
                first += 4 * stride;
            }
        }
        pipe_barrier();                      // (the preparer is done with its sort)
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
        if (act) store_tile(it);             // results to HBM; nothing waited for
        // Stores and loads are issued HERE, while nobody gathers (the other group scatters into LDS): measured against
        // leaving the last quarter / eighth of the loads (0.682 / 0.671 ms vs 0.672 ms) or all stores and loads (0.712 vs
        // 0.665 ms) to the start of this group's prepare step, where they would run beside the other group's gather rounds.
        // Also measured and dropped (profiles/r02_sweep_pipelined_phases.log): the preparer issuing its own loads in two
        // halves, one vector every 0.65 us, or one wave at a time -- they queue behind the gather requests of the other
        // group on the CU's one vector-memory path and land 9-19 us later (0.67-0.83 ms).
        if (it + 2 < nloc) {                 // (it = -1: group 1's first tile)
            load_tile(it + 2);
        } else {
#pragma unroll
            for (int u = 0; u < kSweepK; ++u) q[u] = 0.0;
        }
        pipe_barrier();
    };
    auto prep_step = [&](long it) {          // this group owns tile it+1 (past the last tile: barriers only)
        const bool act = it + 1 < nloc;
        unsigned rank2[kSweepK / 2];         // rank inside the region (histogram ticket), two per register
#pragma unroll
        for (int u = 0; u < kSweepK / 2; ++u) rank2[u] = 0;
        if (act) {
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
        } else {
#pragma unroll
            for (int u = 0; u < kSweepK / 2; ++u) sp2[u] = 0;
        }
        if (NA > 0 && it >= 0) {             // the other group is gathering tile `it`: take its last NA rounds through the scalar path
            const bool rev_o = (it & 1) != 0;
#pragma unroll 1
            for (int u = kSweepK - NA; u < kSweepK; ++u) {
                const int p = tid + u * kPipeGroup;
                const int pp = rev_o ? kSweepTile - 1 - p : p;
                sq[pp] = scalar_eval<FORMULA, NF>(g, sq[pp], extrap);
            }
        }
        pipe_barrier();                      // the gather rounds of the other group are over
        if (act) {
            for (int b = tid; b < kSweepBins; b += kPipeGroup) myhist[b] = 0;   // every lane read its region bases before the barrier
        }
        pipe_barrier();                      // (the gatherer has taken its results out of the tile)
        if (act) {                           // this group's tile goes in
#pragma unroll
            for (int u = 0; u < kSweepK; u += 2) {
                sq[sp2[u / 2] & 0xffffu] = q[u];
                sq[sp2[u / 2] >> 16] = q[u + 1];
                if ((u & 6) == 6) __builtin_amdgcn_sched_barrier(0);
            }
        }
        pipe_barrier();
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


template <int MODE, int FORMULA, int STEP, int U0, int NF>
__global__ __launch_bounds__(kPipeThreads) void assist4_kernel(G1Dev g, const double* __restrict__ xq,
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
    auto load_tile = [&](long it) {                          // the 16 vectors per lane of this group's next tile
        const d2* q2 = reinterpret_cast<const d2*>(xq + ((size_t)blockIdx.x + (size_t)it * gridDim.x) * kSweepTile);
#pragma unroll
        for (int u = 0; u < kSweepK / 2; ++u) {
            const d2 v = stream_load(q2 + tid + u * kPipeGroup);
            q[2 * u] = v.x;
            q[2 * u + 1] = v.y;
        }
    };
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
    if (grp == 0 && nloc > 0) load_tile(0);
    pipe_barrier();
    unsigned* const myhist = hist[grp];
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
            // rounds 0 .. 31 in order, four in flight; the rounds u >= U0 with u % STEP == STEP - 1 belong to the other group
            // (scalar gathers): interleaved with this group's rounds, so their table lines are warm in L2
            int u = -1;
            auto next_u = [&]() {
                do { ++u; } while (STEP > 0 && u >= U0 && (u % (STEP > 0 ? STEP : 1)) == STEP - 1);
                return u;
            };
#pragma unroll 1
            for (;;) {
                int pp[4];
                double qq[4], rr[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int uw = next_u();
                    const int p = tid + uw * kPipeGroup;
                    pp[w] = uw < kSweepK ? (rev ? kSweepTile - 1 - p : p) : -1;
                    qq[w] = pp[w] >= 0 ? sq[pp[w]] : g.xmin;
                }
                if (pp[0] < 0) break;
                eval_batch<MODE, 4, FORMULA, kSweepWin>(g, qq, rr, extrap);
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    if (pp[w] >= 0) sq[pp[w]] = rr[w];
            }
        }
        pipe_barrier();                      // (the preparer is done with its sort)
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
        if (act) store_tile(it);             // results to HBM; nothing waited for
        // Stores and loads are issued HERE, while nobody gathers (the other group scatters into LDS): measured against
        // leaving the last quarter / eighth of the loads (0.682 / 0.671 ms vs 0.672 ms) or all stores and loads (0.712 vs
        // 0.665 ms) to the start of this group's prepare step, where they would run beside the other group's gather rounds.
        // Also measured and dropped (profiles/r02_sweep_pipelined_phases.log): the preparer issuing its own loads in two
        // halves, one vector every 0.65 us, or one wave at a time -- they queue behind the gather requests of the other
        // group on the CU's one vector-memory path and land 9-19 us later (0.67-0.83 ms).
        if (it + 2 < nloc) {                 // (it = -1: group 1's first tile)
            load_tile(it + 2);
        } else {
#pragma unroll
            for (int u = 0; u < kSweepK; ++u) q[u] = 0.0;
        }
        pipe_barrier();
    };
    auto prep_step = [&](long it) {          // this group owns tile it+1 (past the last tile: barriers only)
        const bool act = it + 1 < nloc;
        unsigned rank2[kSweepK / 2];         // rank inside the region (histogram ticket), two per register
#pragma unroll
        for (int u = 0; u < kSweepK / 2; ++u) rank2[u] = 0;
        if (act) {
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
        } else {
#pragma unroll
            for (int u = 0; u < kSweepK / 2; ++u) sp2[u] = 0;
        }
        if (STEP > 0 && it >= 0) {           // the other group is gathering tile `it`: its rounds u >= U0, u % STEP == STEP - 1 through the scalar path
            const bool rev_o = (it & 1) != 0;
#pragma unroll 1
            for (int u = U0; u < kSweepK; ++u) {
                if ((u % (STEP > 0 ? STEP : 1)) != STEP - 1) continue;
                const int p = tid + u * kPipeGroup;
                const int pp = rev_o ? kSweepTile - 1 - p : p;
                sq[pp] = scalar_eval<FORMULA, NF>(g, sq[pp], extrap);
            }
        }
        pipe_barrier();                      // the gather rounds of the other group are over
        if (act) {
            for (int b = tid; b < kSweepBins; b += kPipeGroup) myhist[b] = 0;   // every lane read its region bases before the barrier
        }
        pipe_barrier();                      // (the gatherer has taken its results out of the tile)
        if (act) {                           // this group's tile goes in
#pragma unroll
            for (int u = 0; u < kSweepK; u += 2) {
                sq[sp2[u / 2] & 0xffffu] = q[u];
                sq[sp2[u / 2] >> 16] = q[u + 1];
                if ((u & 6) == 6) __builtin_amdgcn_sched_barrier(0);
            }
        }
        pipe_barrier();
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



template <int MODE, int FORMULA, int NA, int NF>
__global__ __launch_bounds__(kPipeThreads) void assist2_kernel(G1Dev g, const double* __restrict__ xq,
                                                                          double* __restrict__ yq, size_t ntiles,
                                                                          double extrap, double bscale,
                                                                          const int* __restrict__ order_flag,
                                                                          size_t tail, ProbeArgs probe)
{
    __shared__ double sq[kSweepTile];
    __shared__ unsigned hist[2][kSweepBins];
    __shared__ unsigned gbar[2];
    __shared__ unsigned claim;               // next unclaimed chunk of 64 sorted positions of the tile being gathered
    if (*order_flag != 0) return;            // queries already ordered locally: the streaming kernel does the work
    if (threadIdx.x < 2) gbar[threadIdx.x] = 0;
    if (threadIdx.x == 2) claim = 0;
    const int tid = threadIdx.x & (kPipeGroup - 1);
    const int grp = threadIdx.x >> 9;        // wave-uniform: waves 0-7 / 8-15
    const bool last_wg = blockIdx.x == gridDim.x - 1;
    if (probe.host_mailbox && last_wg && threadIdx.x >= kPipeThreads - 64) order_probe_wave(probe);   // for the next call
    const long nloc = ntiles > blockIdx.x ? (long)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
    double q[kSweepK];                       // preparer: the tile's queries; gatherer: its results on their way out
    unsigned sp2[kSweepK / 2];               // sorted positions of this group's tile, two per register
    auto load_tile = [&](long it) {                          // the 16 vectors per lane of this group's next tile
        const d2* q2 = reinterpret_cast<const d2*>(xq + ((size_t)blockIdx.x + (size_t)it * gridDim.x) * kSweepTile);
#pragma unroll
        for (int u = 0; u < kSweepK / 2; ++u) {
            const d2 v = stream_load(q2 + tid + u * kPipeGroup);
            q[2 * u] = v.x;
            q[2 * u + 1] = v.y;
        }
    };
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
    if (grp == 0 && nloc > 0) load_tile(0);
    pipe_barrier();
    unsigned* const myhist = hist[grp];
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
            for (;;) {                         // chunks of 4 x 64 sorted positions, claimed in order by whichever wave is free
                unsigned c = 0;
                if ((threadIdx.x & 63) == 0) c = atomicAdd(&claim, 4u);
                c = (unsigned)__builtin_amdgcn_readfirstlane((int)c);
                if (c >= (unsigned)(kSweepTile / 64)) break;
                double qq[4], rr[4];
                int pp[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int p = (int)(c + w) * 64 + (int)(threadIdx.x & 63);
                    pp[w] = p < kSweepTile ? (rev ? kSweepTile - 1 - p : p) : -1;
                    qq[w] = pp[w] >= 0 ? sq[pp[w]] : g.xmin;
                }
                eval_batch<MODE, 4, FORMULA, kSweepWin>(g, qq, rr, extrap);
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    if (pp[w] >= 0) sq[pp[w]] = rr[w];
            }
        }
        pipe_barrier();                      // (the preparer is done with its sort)
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
        if (act) store_tile(it);             // results to HBM; nothing waited for
        // Stores and loads are issued HERE, while nobody gathers (the other group scatters into LDS): measured against
        // leaving the last quarter / eighth of the loads (0.682 / 0.671 ms vs 0.672 ms) or all stores and loads (0.712 vs
        // 0.665 ms) to the start of this group's prepare step, where they would run beside the other group's gather rounds.
        // Also measured and dropped (profiles/r02_sweep_pipelined_phases.log): the preparer issuing its own loads in two
        // halves, one vector every 0.65 us, or one wave at a time -- they queue behind the gather requests of the other
        // group on the CU's one vector-memory path and land 9-19 us later (0.67-0.83 ms).
        if (it + 2 < nloc) {                 // (it = -1: group 1's first tile)
            load_tile(it + 2);
        } else {
#pragma unroll
            for (int u = 0; u < kSweepK; ++u) q[u] = 0.0;
        }
        pipe_barrier();
    };
    auto prep_step = [&](long it) {          // this group owns tile it+1 (past the last tile: barriers only)
        const bool act = it + 1 < nloc;
        unsigned rank2[kSweepK / 2];         // rank inside the region (histogram ticket), two per register
#pragma unroll
        for (int u = 0; u < kSweepK / 2; ++u) rank2[u] = 0;
        if (act) {
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
        } else {
#pragma unroll
            for (int u = 0; u < kSweepK / 2; ++u) sp2[u] = 0;
        }
        if (NA > 0 && it >= 0) {             // the other group is gathering tile `it`: help through the scalar path, one chunk of 64 at a time
            const bool rev_o = (it & 1) != 0;
            for (;;) {
                unsigned c = 0;
                if ((threadIdx.x & 63) == 0) c = atomicAdd(&claim, 1u);
                c = (unsigned)__builtin_amdgcn_readfirstlane((int)c);
                if (c >= (unsigned)(kSweepTile / 64)) break;
                const int p = (int)c * 64 + (int)(threadIdx.x & 63);
                const int pp = rev_o ? kSweepTile - 1 - p : p;
                sq[pp] = scalar_eval<FORMULA, NF>(g, sq[pp], extrap);
            }
        }
        pipe_barrier();                      // the gather rounds of the other group are over
        if (act) {
            for (int b = tid; b < kSweepBins; b += kPipeGroup) myhist[b] = 0;   // every lane read its region bases before the barrier
        }
        pipe_barrier();                      // (the gatherer has taken its results out of the tile)
        if (act) {                           // this group's tile goes in
#pragma unroll
            for (int u = 0; u < kSweepK; u += 2) {
                sq[sp2[u / 2] & 0xffffu] = q[u];
                sq[sp2[u / 2] >> 16] = q[u + 1];
                if ((u & 6) == 6) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (tid == 0) claim = 0;             // nobody claims between barriers A and C
        pipe_barrier();
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



template <int MODE, int FORMULA, int NA, int NF, int GC>
__global__ __launch_bounds__(kPipeThreads) void assist3_kernel(G1Dev g, const double* __restrict__ xq,
                                                                          double* __restrict__ yq, size_t ntiles,
                                                                          double extrap, double bscale,
                                                                          const int* __restrict__ order_flag,
                                                                          size_t tail, ProbeArgs probe)
{
    __shared__ double sq[kSweepTile];
    __shared__ unsigned hist[2][kSweepBins];
    __shared__ unsigned gbar[2];
    __shared__ unsigned claim;               // next unclaimed chunk of 64 sorted positions of the tile being gathered
    if (*order_flag != 0) return;            // queries already ordered locally: the streaming kernel does the work
    if (threadIdx.x < 2) gbar[threadIdx.x] = 0;
    if (threadIdx.x == 2) claim = 0;
    const int tid = threadIdx.x & (kPipeGroup - 1);
    const int grp = threadIdx.x >> 9;        // wave-uniform: waves 0-7 / 8-15
    const bool last_wg = blockIdx.x == gridDim.x - 1;
    if (probe.host_mailbox && last_wg && threadIdx.x >= kPipeThreads - 64) order_probe_wave(probe);   // for the next call
    const long nloc = ntiles > blockIdx.x ? (long)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
    double q[kSweepK];                       // preparer: the tile's queries; gatherer: its results on their way out
    unsigned sp2[kSweepK / 2];               // sorted positions of this group's tile, two per register
    auto load_tile = [&](long it) {                          // the 16 vectors per lane of this group's next tile
        const d2* q2 = reinterpret_cast<const d2*>(xq + ((size_t)blockIdx.x + (size_t)it * gridDim.x) * kSweepTile);
#pragma unroll
        for (int u = 0; u < kSweepK / 2; ++u) {
            const d2 v = stream_load(q2 + tid + u * kPipeGroup);
            q[2 * u] = v.x;
            q[2 * u + 1] = v.y;
        }
    };
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
    if (grp == 0 && nloc > 0) load_tile(0);
    pipe_barrier();
    unsigned* const myhist = hist[grp];
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
            // chunks of GC x 64 sorted positions, claimed in order by whichever wave is free; the NEXT claim is issued before
            // the current chunks are evaluated, so the LDS atomic's latency is hidden
            unsigned cn = 0;
            if ((threadIdx.x & 63) == 0) cn = atomicAdd(&claim, (unsigned)GC);
            for (;;) {
                const unsigned c = (unsigned)__builtin_amdgcn_readfirstlane((int)cn);
                if (c >= (unsigned)(kSweepTile / 64)) break;
                if ((threadIdx.x & 63) == 0) cn = atomicAdd(&claim, (unsigned)GC);
#pragma unroll
                for (int h = 0; h < GC / 4; ++h) {
                    double qq[4], rr[4];
                    int pp[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const int p = (int)(c + 4 * h + w) * 64 + (int)(threadIdx.x & 63);
                        pp[w] = p < kSweepTile ? (rev ? kSweepTile - 1 - p : p) : -1;
                        qq[w] = pp[w] >= 0 ? sq[pp[w]] : g.xmin;
                    }
                    eval_batch<MODE, 4, FORMULA, kSweepWin>(g, qq, rr, extrap);
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                        if (pp[w] >= 0) sq[pp[w]] = rr[w];
                }
            }
        }
        pipe_barrier();                      // (the preparer is done with its sort)
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
        if (act) store_tile(it);             // results to HBM; nothing waited for
        // Stores and loads are issued HERE, while nobody gathers (the other group scatters into LDS): measured against
        // leaving the last quarter / eighth of the loads (0.682 / 0.671 ms vs 0.672 ms) or all stores and loads (0.712 vs
        // 0.665 ms) to the start of this group's prepare step, where they would run beside the other group's gather rounds.
        // Also measured and dropped (profiles/r02_sweep_pipelined_phases.log): the preparer issuing its own loads in two
        // halves, one vector every 0.65 us, or one wave at a time -- they queue behind the gather requests of the other
        // group on the CU's one vector-memory path and land 9-19 us later (0.67-0.83 ms).
        if (it + 2 < nloc) {                 // (it = -1: group 1's first tile)
            load_tile(it + 2);
        } else {
#pragma unroll
            for (int u = 0; u < kSweepK; ++u) q[u] = 0.0;
        }
        pipe_barrier();
    };
    auto prep_step = [&](long it) {          // this group owns tile it+1 (past the last tile: barriers only)
        const bool act = it + 1 < nloc;
        unsigned rank2[kSweepK / 2];         // rank inside the region (histogram ticket), two per register
#pragma unroll
        for (int u = 0; u < kSweepK / 2; ++u) rank2[u] = 0;
        if (act) {
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
        } else {
#pragma unroll
            for (int u = 0; u < kSweepK / 2; ++u) sp2[u] = 0;
        }
        if (NA > 0 && it >= 0) {             // the other group is gathering tile `it`: help through the scalar path, one chunk of 64 at a time
            const bool rev_o = (it & 1) != 0;
            for (;;) {
                unsigned c = 0;
                if ((threadIdx.x & 63) == 0) c = atomicAdd(&claim, 1u);
                c = (unsigned)__builtin_amdgcn_readfirstlane((int)c);
                if (c >= (unsigned)(kSweepTile / 64)) break;
                const int p = (int)c * 64 + (int)(threadIdx.x & 63);
                const int pp = rev_o ? kSweepTile - 1 - p : p;
                sq[pp] = scalar_eval<FORMULA, NF>(g, sq[pp], extrap);
            }
        }
        pipe_barrier();                      // the gather rounds of the other group are over
        if (act) {
            for (int b = tid; b < kSweepBins; b += kPipeGroup) myhist[b] = 0;   // every lane read its region bases before the barrier
        }
        pipe_barrier();                      // (the gatherer has taken its results out of the tile)
        if (act) {                           // this group's tile goes in
#pragma unroll
            for (int u = 0; u < kSweepK; u += 2) {
                sq[sp2[u / 2] & 0xffffu] = q[u];
                sq[sp2[u / 2] >> 16] = q[u + 1];
                if ((u & 6) == 6) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (tid == 0) claim = 0;             // nobody claims between barriers A and C
        pipe_barrier();
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




__global__ void fill_random(double* x, size_t n, unsigned long long seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        x[i] = (double)(z >> 11) * 0x1.0p-53;
    }
}

int main(int argc, char** argv)
{
    const size_t nq = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000000ull;
    const int ng = 1000000;
    std::vector<double> y(ng + 1);
    for (int i = 0; i < ng; ++i) { const double x = (double)i / (ng - 1); y[i] = sin(6.283185307179586 * x) + 0.5 * x; }
    y[ng] = y[ng - 1];
    double *dy, *xq, *ya, *yb;
    CK(hipMalloc(&dy, (ng + 1) * 8)); CK(hipMemcpy(dy, y.data(), (ng + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&xq, nq * 8)); CK(hipMalloc(&ya, nq * 8)); CK(hipMalloc(&yb, nq * 8));
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, xq, nq, 0x5EED0003ull);
    CK(hipMemset(ya, 0, nq * 8)); CK(hipMemset(yb, 0xff, nq * 8));
    G1Dev g; memset(&g, 0, sizeof g);
    g.y = dy; g.n = ng; g.xmin = 0.0; g.xmax = 1.0; g.x0 = 0.0; g.span = 1.0; g.den = ng - 1; g.rden = 1.0 / g.den;
    g.dx = 1.0 / (ng - 1); g.scale = 1.0 / g.dx; g.formula = 3; g.pin_last = 1;
    const double bscale = (double)kSweepBins;
    const size_t ntiles = nq / kSweepTile;
    int* flag; CK(hipMalloc(&flag, 16)); CK(hipMemset(flag, 0, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        std::vector<float> ts;
        for (int r = 0; r < 7; ++r) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 5; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / 5);
        }
        std::sort(ts.begin(), ts.end());
        printf("%-58s %.4f ms (min %.4f, max %.4f)  %.1f %% of 8 TB/s\n", name, ts[3], ts[0], ts[6], (16.0 * nq + 8e6) / (ts[3] * 1e-3) / 8e12 * 100);
        return ts[3];
    };
    auto prod = [&] { hipLaunchKernelGGL((interp1_sweep_pipe_kernel<0, 3>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, ya, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{}); };
    time("product pipelined kernel", prod);
    time("assist NA=0 (control: same code, no assist)", [&] { hipLaunchKernelGGL((assist_kernel<0, 3, 0, 8>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{}); });
    time("assist NA=4 rounds (12.5 %), 8 scalar loads in flight", [&] { hipLaunchKernelGGL((assist_kernel<0, 3, 4, 8>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{}); });
    time("assist NA=4, 16 in flight", [&] { hipLaunchKernelGGL((assist_kernel<0, 3, 4, 16>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{}); });
    time("assist NA=8 rounds (25 %), 16 in flight", [&] { hipLaunchKernelGGL((assist_kernel<0, 3, 8, 16>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{}); });
    time("assist2 (dynamic chunks) NA=0: gathering group only", [&] { hipLaunchKernelGGL((assist2_kernel<0, 3, 0, 8>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{}); });
    time("assist2 (dynamic chunks) + scalar helpers, 8 in flight", [&] { hipLaunchKernelGGL((assist2_kernel<0, 3, 1, 8>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{}); });
    time("assist2 (dynamic chunks) + scalar helpers, 16 in flight", [&] { hipLaunchKernelGGL((assist2_kernel<0, 3, 1, 16>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{}); });
    time("assist3 GC=4 pipelined claims, no helpers", [&] { hipLaunchKernelGGL((assist3_kernel<0, 3, 0, 8, 4>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{}); });
    time("assist3 GC=4 pipelined claims + scalar helpers (8)", [&] { hipLaunchKernelGGL((assist3_kernel<0, 3, 1, 8, 4>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{}); });
    time("assist3 GC=8 pipelined claims + scalar helpers (8)", [&] { hipLaunchKernelGGL((assist3_kernel<0, 3, 1, 8, 8>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{}); });
    time("assist3 GC=8 pipelined claims + scalar helpers (4 in flight)", [&] { hipLaunchKernelGGL((assist3_kernel<0, 3, 1, 4, 8>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{}); });
    time("assist4 static interleave: rounds 15,23,31 (STEP 8, U0 12)", [&] { hipLaunchKernelGGL((assist4_kernel<0, 3, 8, 12, 8>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{}); });
    time("assist4 static interleave: rounds 23,31 (STEP 8, U0 20)", [&] { hipLaunchKernelGGL((assist4_kernel<0, 3, 8, 20, 8>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{}); });
    time("assist4 static interleave: rounds 15,19,..,31 (STEP 4, U0 12)", [&] { hipLaunchKernelGGL((assist4_kernel<0, 3, 4, 12, 8>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{}); });
    time("assist4 control (no helper rounds, same loop)", [&] { hipLaunchKernelGGL((assist4_kernel<0, 3, 0, 0, 8>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{}); });
    time("product pipelined kernel (again)", prod);
    hipLaunchKernelGGL((assist4_kernel<0, 3, 8, 12, 8>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{});
    CK(hipDeviceSynchronize());
    std::vector<double> a(1 << 22), b(1 << 22);
    size_t bad = 0;
    for (size_t off = 0; off < ntiles * kSweepTile; off += a.size()) {
        const size_t m = std::min(a.size(), ntiles * kSweepTile - off);
        CK(hipMemcpy(a.data(), ya + off, m * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), yb + off, m * 8, hipMemcpyDeviceToHost));
        if (memcmp(a.data(), b.data(), m * 8) != 0) for (size_t i = 0; i < m; ++i) bad += memcmp(&a[i], &b[i], 8) != 0;
    }
    printf("outputs: %zu of %zu differ\n", bad, ntiles * kSweepTile);
    return 0;
}
