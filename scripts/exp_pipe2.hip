// Exploration harness (NOT product code), round 3: the pipelined region sweep WITHOUT the workgroup barrier after the scatter.
// In interp1_sweep_pipe_kernel the group that has just gathered issues its 16 result stores and the 16 loads of its next
// tile per lane (a 256-KiB burst per CU that every CU issues at the same moment: the issue blocks for ~6.7 us), and the
// group that has just scattered its tile into LDS waits for that at barrier C before it starts its gather rounds.  Only the
// scattering group's own eight waves need to agree that the tile is in: here C is a barrier among those waves only, and the
// rounds start while the other group is still issuing.  VARIANT 0 = the product schedule (control), 1 = no barrier C.
//
// VARIANT 2 (time slots): the CUs of an XCD in two halves (parity of blockIdx / 8) that take turns on the L2: the gather
// rounds of a tile are cut into four quarters, and a quarter only starts inside a time slot of the workgroup's parity
// (slot = global 100 MHz clock / TAU ticks, the same on every CU: no communication).  Question: if only half of an XCD's
// CUs gather at any moment, do they gather twice as fast (the L2 request rate being what bounds the rounds), so that the
// other half of the time is free for the streams?  VARIANT 3: same slots, every workgroup parity 0 (control: all gather
// in the even slots and nobody in the odd ones).
//
// VARIANT 4 (time slots + deferred burst): as 2, and the group that has gathered does NOT issue its stores and next loads
// between the tiles: it keeps the results in registers through barrier C and issues the stores in the first slot of the other
// parity of its next (prepare) step and the loads of its next tile in the following such slot -- slots in which the other
// half of the XCD's CUs gather and this CU's own vector memory path is idle.  Then a tile costs eight slots plus the
// read-back and the scatter, and the bursts of one half of the XCD fall into the rounds of the other.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I armadillocudalinearinterpolation_amd/csrc \
//         -o scripts/exp_pipe2 scripts/exp_pipe2.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mi_interp1_sweep.hpp"

using namespace mi_interp1;

#ifndef TAU
#define TAU 256          // slot length in 100 MHz ticks (256 = 2.56 us)
#endif
__device__ unsigned long long g_active_ticks;   // time inside the gather rounds proper (wave 0 of each gathering group), summed over the grid
__device__ __forceinline__ void wait_slot(unsigned parity)
{
    for (;;) {
        const unsigned long long t = wall_clock64();
        if ((unsigned)((t / TAU) & 1ull) == parity) break;
        __builtin_amdgcn_s_sleep(4);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE, int FORMULA, int VARIANT>
__global__ __launch_bounds__(kPipeThreads) void pipe2_kernel(G1Dev g, const double* __restrict__ xq,
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
    if (VARIANT != 4 && grp == 0 && nloc > 0) load_tile(0);
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
    unsigned active_ticks = 0;
    long pending = -1;                       // VARIANT 4: tile whose results sit in q, not yet stored
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
                if (VARIANT == 2 || VARIANT == 4) wait_slot((blockIdx.x >> 3) & 1u);
                if (VARIANT == 3) wait_slot(0u);
                const unsigned ta_ = (unsigned)wall_clock64();
                pipe_gather_rounds<MODE, FORMULA>(g, sq, first, stride, extrap);
                if (VARIANT >= 2 && __builtin_amdgcn_readfirstlane((int)(threadIdx.x & 511)) == 0) active_ticks += (unsigned)wall_clock64() - ta_;
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
        if (VARIANT == 4) {                  // results stay in q; stored at the start of this group's next prepare step
            pending = act ? it : -1;
            pipe_barrier();
            return;
        }
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
        if (VARIANT != 1) pipe_barrier();   // C (product form): wait until the other group's tile is in
    };
    auto prep_step = [&](long it) {          // this group owns tile it+1 (past the last tile: barriers only)
        const bool act = it + 1 < nloc;
        unsigned rank2[kSweepK / 2];         // rank inside the region (histogram ticket), two per register
#pragma unroll
        for (int u = 0; u < kSweepK / 2; ++u) rank2[u] = 0;
        if (VARIANT == 4) {
            const unsigned par = (blockIdx.x >> 3) & 1u;
            if (pending >= 0) { wait_slot(1u - par); store_tile(pending); pending = -1; wait_slot(par); }
            if (act) { wait_slot(1u - par); load_tile(it + 1); }
        }
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
        if (VARIANT != 1) pipe_barrier();   // C (product form)
        else group_barrier();               // only the scattering group waits for its own scatter; the other group is still issuing stores / loads
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
    if (VARIANT == 4 && pending >= 0) store_tile(pending);
    if (VARIANT >= 2 && (threadIdx.x & 511) == 0) atomicAdd(&g_active_ticks, (unsigned long long)active_ticks);
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
    auto active = [&](const char* name, auto launch) {     // time inside the gather rounds, per tile
        unsigned long long z = 0, v = 0;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_active_ticks), &z, 8));
        launch();
        CK(hipDeviceSynchronize());
        CK(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_active_ticks), 8));
        printf("    %s: %.2f us per tile inside the gather rounds (the four quarters, waits excluded)\n", name, (double)v * 0.01 / (double)ntiles);
    };
    auto v2 = [&] { hipLaunchKernelGGL((pipe2_kernel<0, 3, 2>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{}); };
    auto v3 = [&] { hipLaunchKernelGGL((pipe2_kernel<0, 3, 3>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{}); };
    time("VARIANT 2: quarter-rounds in alternating time slots (two halves of each XCD)", v2);
    active("VARIANT 2", v2);
    auto v4 = [&] { hipLaunchKernelGGL((pipe2_kernel<0, 3, 4>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{}); };
    time("VARIANT 4: time slots + stores / loads deferred into the other half's slots", v4);
    active("VARIANT 4", v4);
    time("VARIANT 3: quarter-rounds in the even slots only, every CU (control)", v3);
    active("VARIANT 3", v3);
    printf("TAU = %d ticks = %.2f us\n", TAU, TAU * 0.01);
    float t[2][2];
    for (int rep = 0; rep < 2; ++rep) {     // alternate the two forms: box-to-box and run-to-run drift is a few per cent
        t[rep][0] = time("VARIANT 0: product schedule (barrier C for all 16 waves)", [&] {
            hipLaunchKernelGGL((pipe2_kernel<0, 3, 0>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, ya, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{});
        });
        t[rep][1] = time("VARIANT 1: rounds start when the scatter group is done", [&] {
            hipLaunchKernelGGL((pipe2_kernel<0, 3, 1>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{});
        });
    }
    printf("variant 1 / variant 0 = %.3f, %.3f\n", t[0][1] / t[0][0], t[1][1] / t[1][0]);
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
