// interp1 region-sweep kernels: unordered queries over a table that does not fit one XCD's L2.
#pragma once
#include "mi_interp1_eval.hpp"

namespace mi_interp1 {

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
// The query / result streams are non-temporal ("nt"): of the seven sc0/sc1/nt combinations timed on both streams, nt on
// both is the fastest (profiles/r02_exp_stream_cache_policy.log) -- the streams must not displace the table in L2.
__device__ __forceinline__ d2 stream_load(const d2* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stream_store(d2 v, d2* p) { __builtin_nontemporal_store(v, p); }

constexpr size_t kSweepMinTilesPerCu = 2;   // profiles/r02_strong_scaling_shards.log: at 3 tiles per CU (1.25e7 queries) the sweep still wins 0.097 vs 0.135 ms
constexpr int kSweepM3Batch = 4;     // mode 3, pipelined form: queries per lane whose gathers are in flight together; 8 does not
                                     // fit the 128 registers of a 1024-lane workgroup (139-161 spilled: 1.59 ms against 1.03)
constexpr bool kSweepWin = false;    // mode 3: node G-1 on demand; fetching it eagerly with G and G+1 costs 1.25 ms against 1.03
                                     // (profiles/r02_mode3_gather_variants.log)
constexpr int kSweepThreads = 512;
constexpr int kSweepK = 32;                               // queries per lane per tile
constexpr int kSweepTile = kSweepThreads * kSweepK;       // queries per tile (8 B of LDS each)
// (handing the gather chunks of a tile out dynamically to whichever wave is free was measured too: 0.711-0.716 ms against
// 0.707-0.716 ms static -- the memory path returns in order, the wave that issued last finishes last whatever it was given)
template <int MODE, int FORMULA>
__global__ __launch_bounds__(kSweepThreads) void interp1_sweep_kernel(G1Dev g, const double* __restrict__ xq,
                                                                      double* __restrict__ yq, size_t ntiles,
                                                                      double extrap, double bscale,
                                                                      const int* __restrict__ order_flag,
                                                                      size_t tail, ProbeArgs probe)
{
    __shared__ double sq[kSweepTile];
    __shared__ unsigned hist[kSweepBins];
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
        for (int b = tid; b < kSweepBins; b += kSweepThreads) hist[b] = 0;
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
// WHILE another tile gathers.  Measured (profiles/r02_exp_mix_stream_beside_gathers.log; harness: scripts/ARCHIVE.md): stream loads or stores issued
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
    constexpr int B = (MODE == 3) ? kSweepM3Batch : 4;
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

// The gatherer issues all sixteen loads of its next tile right behind its result stores (one burst per tile).
// Three workgroup barriers per tile: after the gather rounds, after the read-back, after the scatter.  The preparing
// group orders its own histogram -> prefix -> positions passes with a counter in LDS that only its 8 waves touch, so
// the gathering waves run their 8 rounds without stopping (barriers inside the rounds cost 1.2 us each: every
// interval then ends with its slowest wave).
template <int MODE, int FORMULA>
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
            for (int iv = 0; iv < 4; ++iv) { // eight rounds; the other group prepares its tile meanwhile
                pipe_gather_rounds<MODE, FORMULA>(g, sq, first, stride, extrap);
                first += 8 * stride;
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

}  // namespace mi_interp1
