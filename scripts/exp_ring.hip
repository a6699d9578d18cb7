// Exploration harness (NOT product code), round 3: DESIGN.md section 9 item 1 / VERDICT r2 item 3 -- the pipelined region
// sweep with the next tile's queries TRICKLED in by a dedicated loader wave through an LDS ring (LDS-DMA,
// global_load_lds_dwordx4) instead of one 128-KiB burst per CU issued by the waves that have just gathered.
//
// Built on the product headers (same eval_batch arithmetic, same counting sort); the driver runs the product's pipelined
// kernel and the ring kernel on the same queries, compares the outputs bit for bit and times both.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I armadillocudalinearinterpolation_amd/csrc \
//         -o scripts/exp_ring scripts/exp_ring.hip
//
// Roles.  One 1024-lane workgroup per CU, two groups of 8 waves that swap roles tile by tile as in
// interp1_sweep_pipe_kernel.  In the GATHERING group only waves 0-6 gather (the rounds are bound by the L2 request rate,
// seven waves keep it as busy as eight); its wave 7 is the loader of the OTHER group's next tile: 16 chunks of 8 KiB
// (one 16-B vector per preparing lane), each chunk eight 1-KiB LDS-DMA instructions into a ring of NSLOT slots.  The
// PREPARING group's eight waves take each chunk out of the ring into registers as it lands (ds_read_b128) and histogram
// it on the spot, so the sort is finished one chunk after the last byte arrived.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mi_interp1_sweep.hpp"

using namespace mi_interp1;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#ifndef NSLOT
#define NSLOT 3
#endif
#ifndef PACE
#define PACE 0          // s_sleep argument between chunks of the loader (0: as fast as the ring allows)
#endif
#ifndef GB
#define GB 4            // gathers in flight per lane in the rounds
#endif
constexpr int kChunkBytes = 8192;                       // 512 lanes x 16 B
constexpr int kChunks = kSweepTile * 8 / kChunkBytes;   // 16 per tile
constexpr int kGatherLanes = 7 * 64;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// stamps: [wg][slot] accumulated wall_clock64 ticks (100 MHz); slots:
// 0 gather rounds (wave 0 of the gathering group), 1 wait at barrier A, 2 read-back, 3 stores + barrier C,
// 4 loader: step start -> last chunk landed, 5 preparer: step start -> sorted positions ready, 6 preparer: wait at barrier A,
// 7 scatter (barrier B -> barrier C)
constexpr int kStampSlots = 8;

template <int MODE, int FORMULA, int TIMED>
__global__ __launch_bounds__(kPipeThreads) void ring_kernel(G1Dev g, const double* __restrict__ xq, double* __restrict__ yq,
                                                            size_t ntiles, double extrap, double bscale,
                                                            unsigned long long* __restrict__ stamps)
{
    __shared__ double sq[kSweepTile];
    __shared__ __attribute__((aligned(16))) double ring[NSLOT][kChunkBytes / 8];
    __shared__ unsigned hist[2][kSweepBins];
    __shared__ unsigned gbar[2];
    __shared__ unsigned ring_arrived, ring_consumed;   // monotonic: chunks landed, wave-consumptions
    if (threadIdx.x < 2) gbar[threadIdx.x] = 0;
    if (threadIdx.x == 2) ring_arrived = 0;
    if (threadIdx.x == 3) ring_consumed = 0;
    const int tid = threadIdx.x & (kPipeGroup - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // provably wave-uniform: scalar branches on the roles
    const int grp = wave >> 3;
    const int wl = wave & 7;                             // wave inside its group
    const int lane = threadIdx.x & 63;
    const long nloc = ntiles > blockIdx.x ? (long)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
    double q[kSweepK];
    unsigned sp2[kSweepK / 2];
    for (int b = threadIdx.x; b < 2 * kSweepBins; b += kPipeThreads) (&hist[0][0])[b] = 0;
    pipe_barrier();
    unsigned* const myhist = hist[grp];
    unsigned gb_target = 0;
    auto group_barrier = [&]() {
        gb_target += kPipeGroup / 64;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) atomicAdd(&gbar[grp], 1u);
        while (*reinterpret_cast<volatile unsigned*>(&gbar[grp]) < gb_target) __builtin_amdgcn_s_sleep(2);
        asm volatile("" ::: "memory");
    };
    unsigned tacc[kStampSlots] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned tlast = 0;
#define STAMP0() if (TIMED) tlast = (unsigned)wall_clock64();
#define STAMP(slot) if (TIMED) { const unsigned now_ = (unsigned)wall_clock64(); tacc[slot] += now_ - tlast; tlast = now_; }
    const unsigned ring_base = (unsigned)(size_t)(&ring[0][0]);   // LDS byte address (generic -> low 32 bits = LDS offset)

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
    // loader (wave 7 of the gathering group): tile `it` of this workgroup for the other group
    auto load_tile_ring = [&](long it) {
        const d2* q2 = reinterpret_cast<const d2*>(xq + ((size_t)blockIdx.x + (size_t)it * gridDim.x) * kSweepTile);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's result stores of a step ago: long done
        const unsigned c0 = (unsigned)it * kChunks;
        for (int u = 0; u < kChunks; ++u) {
            const unsigned c = c0 + u;
            // slot free: every preparing wave has taken chunk c - NSLOT out of the ring
            if (c >= NSLOT) {
                const unsigned need = 8u * (c - NSLOT + 1);
                while (*reinterpret_cast<volatile unsigned*>(&ring_consumed) < need) __builtin_amdgcn_s_sleep(1);
            }
            const unsigned dst = __builtin_amdgcn_readfirstlane(ring_base + (c % NSLOT) * kChunkBytes);
#pragma unroll
            for (int i = 0; i < 8; ++i) glds16(q2 + u * kPipeGroup + i * 64 + lane, dst + i * 1024);
            if (u > 0) {                                             // chunk u-1 has landed when at most 8 loads are outstanding
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                if (lane == 0) *reinterpret_cast<volatile unsigned*>(&ring_arrived) = c;   // chunks 0..c-1 landed
            }
            if (PACE) __builtin_amdgcn_s_sleep(PACE);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) *reinterpret_cast<volatile unsigned*>(&ring_arrived) = c0 + kChunks;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    auto gather_step = [&](long it) {        // this group owns tile `it` (it = -1: nothing yet)
        const bool act = it >= 0;
        const bool rev = (it & 1) != 0;
        STAMP0()
        if (wl == 7) {
            if (it + 1 < nloc) load_tile_ring(it + 1);
            STAMP(4)
        } else if (act) {
            // 37 rounds of 448 sorted positions; GB rounds in flight together
            const int p0 = wl * 64 + lane;
#pragma unroll 1
            for (int r0 = 0; r0 * kGatherLanes < kSweepTile; r0 += GB) {
                double qq[GB], rr[GB];
                int pp[GB];
#pragma unroll
                for (int w = 0; w < GB; ++w) {
                    const int p = p0 + (r0 + w) * kGatherLanes;
                    pp[w] = p < kSweepTile ? (rev ? kSweepTile - 1 - p : p) : -1;
                    qq[w] = pp[w] >= 0 ? sq[pp[w]] : g.xmin;
                }
                eval_batch<MODE, GB, FORMULA, kSweepWin>(g, qq, rr, extrap);
#pragma unroll
                for (int w = 0; w < GB; ++w)
                    if (pp[w] >= 0) sq[pp[w]] = rr[w];
            }
            STAMP(0)
        }
        pipe_barrier();                      // A: rounds over, the other group's sort done
        if (wl == 0) STAMP(1)
        if (act) {
#pragma unroll
            for (int u = 0; u < kSweepK; u += 2) {
                q[u] = sq[sp2[u / 2] & 0xffffu];
                q[u + 1] = sq[sp2[u / 2] >> 16];
                if ((u & 6) == 6) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < kSweepK; ++u) q[u] = 0.0;
        }
        pipe_barrier();                      // B: results are out of the tile
        if (wl == 0) STAMP(2)
        if (act) store_tile(it);
#pragma unroll
        for (int u = 0; u < kSweepK; ++u) q[u] = 0.0;   // dead until this group's prepare step fills it from the ring
        pipe_barrier();                      // C: the other group's tile is in
        if (wl == 0) STAMP(3)
    };
    auto prep_step = [&](long it) {          // this group owns tile it+1
        const bool act = it + 1 < nloc;
        unsigned rank2[kSweepK / 2];
#pragma unroll
        for (int u = 0; u < kSweepK / 2; ++u) rank2[u] = 0;
        STAMP0()
        if (act) {
            const unsigned c0 = (unsigned)(it + 1) * kChunks;
#pragma unroll
            for (int u = 0; u < kChunks; ++u) {
                const unsigned c = c0 + u;
                while (*reinterpret_cast<volatile unsigned*>(&ring_arrived) <= c) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                const d2 v = *reinterpret_cast<const d2*>(&ring[c % NSLOT][2 * tid]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) atomicAdd(&ring_consumed, 1u);
                q[2 * u] = v.x;
                q[2 * u + 1] = v.y;
                const unsigned r0 = atomicAdd(&myhist[sweep_bin(v.x, g.xmin, bscale)], 1u);
                const unsigned r1 = atomicAdd(&myhist[sweep_bin(v.y, g.xmin, bscale)], 1u);
                rank2[u] = r0 | (r1 << 16);
            }
            group_barrier();
            if (tid < 64) {
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
            for (int u = 0; u < kSweepK; u += 2) {
                double qa = q[u], qb = q[u + 1];
                asm volatile("" : "+v"(qa), "+v"(qb));
                const unsigned p0 = myhist[sweep_bin(qa, g.xmin, bscale)] + (rank2[u / 2] & 0xffffu);
                const unsigned p1 = myhist[sweep_bin(qb, g.xmin, bscale)] + (rank2[u / 2] >> 16);
                sp2[u / 2] = p0 | (p1 << 16);
                if ((u & 6) == 6) __builtin_amdgcn_sched_barrier(0);
            }
            if (wl == 0) STAMP(5)
        } else {
#pragma unroll
            for (int u = 0; u < kSweepK / 2; ++u) sp2[u] = 0;
        }
        pipe_barrier();                      // A
        if (wl == 0) STAMP(6)
        if (act) {
            for (int b = tid; b < kSweepBins; b += kPipeGroup) myhist[b] = 0;
        }
        pipe_barrier();                      // B
        if (wl == 0) STAMP0()
        if (act) {
#pragma unroll
            for (int u = 0; u < kSweepK; u += 2) {
                sq[sp2[u / 2] & 0xffffu] = q[u];
                sq[sp2[u / 2] >> 16] = q[u + 1];
                if ((u & 6) == 6) __builtin_amdgcn_sched_barrier(0);
            }
        }
        pipe_barrier();                      // C
        if (wl == 0) STAMP(7)
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
    if (TIMED && lane == 0 && (wl == 0 || wl == 7)) {
        for (int k = 0; k < kStampSlots; ++k)
            if (tacc[k]) atomicAdd(&stamps[(size_t)blockIdx.x * kStampSlots + k], (unsigned long long)tacc[k]);
    }
#undef STAMP
#undef STAMP0
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
    unsigned long long* stamps; CK(hipMalloc(&stamps, 256 * kStampSlots * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        std::vector<float> ts;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 5; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / 5);
        }
        std::sort(ts.begin(), ts.end());
        printf("%-52s %.4f ms (min %.4f, max %.4f)  %.1f %% of 8 TB/s\n", name, ts[2], ts[0], ts[4], (16.0 * nq + 8e6) / (ts[2] * 1e-3) / 8e12 * 100);
        return ts[2];
    };
    const float tp = time("product interp1_sweep_pipe_kernel<0,3>", [&] {
        hipLaunchKernelGGL((interp1_sweep_pipe_kernel<0, 3>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, ya, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{});
    });
    const float tr = time("ring kernel (NSLOT " "x8KiB, loader wave)", [&] {
        hipLaunchKernelGGL((ring_kernel<0, 3, 0>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, stamps);
    });
    printf("NSLOT=%d PACE=%d GB=%d: ring / product = %.3f\n", NSLOT, PACE, GB, tr / tp);
    // bit-for-bit comparison of the two outputs over the tiled part
    {
        std::vector<double> a(1 << 22), b(1 << 22);
        size_t bad = 0;
        for (size_t off = 0; off < ntiles * kSweepTile; off += a.size()) {
            const size_t m = std::min(a.size(), ntiles * kSweepTile - off);
            CK(hipMemcpy(a.data(), ya + off, m * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), yb + off, m * 8, hipMemcpyDeviceToHost));
            if (memcmp(a.data(), b.data(), m * 8) != 0) for (size_t i = 0; i < m; ++i) bad += memcmp(&a[i], &b[i], 8) != 0;
        }
        printf("outputs: %zu of %zu differ\n", bad, ntiles * kSweepTile);
    }
    // phase stamps
    CK(hipMemset(stamps, 0, 256 * kStampSlots * 8));
    hipLaunchKernelGGL((ring_kernel<0, 3, 1>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, stamps);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(256 * kStampSlots);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    const char* names[kStampSlots] = {"gather rounds (wave 0)", "gatherer waits at A", "read-back", "stores + wait C", "loader: start -> tile landed", "preparer: start -> sorted", "preparer waits at A", "scatter + wait C"};
    const double tiles_per_wg = (double)ntiles / 256;
    for (int k = 0; k < kStampSlots; ++k) {
        double s = 0;
        for (int w = 0; w < 256; ++w) s += (double)h[(size_t)w * kStampSlots + k];
        printf("  %-30s %7.2f us per tile\n", names[k], s / 256 * 0.01 / tiles_per_wg);
    }
    return 0;
}
