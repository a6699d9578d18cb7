// Exploration harness (NOT product code), round 2: does a CU's vector L1 (TCP) return data in order ACROSS waves?
// If it does, a few HBM-latency stream loads mixed into a flow of L2-hit gathers hold every gather behind them
// (head-of-line blocking) and "load the next tile while this one gathers" cannot work inside one CU -- which is what
// every overlap experiment of round 1 (profiles/r01_exp_region_sweep_phases.log) looked like.
// One 512-lane workgroup per CU.  GW waves gather (hash-random 8-B loads from a 1 MB, L2-resident table, 4 in flight);
// SW waves stream (16 B per lane, SU wave-instructions in flight) from / to a buffer far larger than any cache, or
// from a small L2-resident buffer.  Each role stamps wall_clock64 when it finishes, so the gather time is known
// separately from the kernel time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// SMODE: 0 none, 1 nt loads from the big buffer, 2 nt stores to it, 3 loads from a small (L2-resident) buffer
template <int GW, int SW, int SU, int SMODE>
__global__ __launch_bounds__(512) void mix(const double* __restrict__ t, unsigned mask, int giters, const d2* __restrict__ big, d2* __restrict__ bigw, size_t vec_per_cu, int siters, unsigned small_mask, unsigned long long* __restrict__ stamps, double* __restrict__ sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long t0 = wall_clock64();
    double acc = 0;
    if (wave < GW) {
        unsigned s = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u;
        for (int it = 0; it < giters; ++it) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { s = hash(s + u + it); v[u] = t[s & mask]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u];
        }
        if (lane == 0) stamps[(size_t)blockIdx.x * 16 + wave] = wall_clock64() - t0;
    } else if (wave >= 8 - SW && SMODE != 0) {
        const int sw = wave - (8 - SW);
        const size_t base = (size_t)blockIdx.x * vec_per_cu;
        size_t pos = (size_t)sw * 64 + lane;                 // vectors; SW waves interleave 1 KB chunks
        for (int it = 0; it < siters; ++it) {
            d2 v[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                size_t p = pos + (size_t)u * SW * 64;
                if (SMODE == 3) p &= small_mask;
                else if (p >= vec_per_cu) p -= vec_per_cu;
                if (SMODE == 2) { d2 w; w.x = (double)it; w.y = (double)u; __builtin_nontemporal_store(w, bigw + base + p); }
                else v[u] = __builtin_nontemporal_load(big + base + p);
            }
            if (SMODE != 2) {
#pragma unroll
                for (int u = 0; u < SU; ++u) acc += v[u].x + v[u].y;
            }
            pos += (size_t)SU * SW * 64;
            if (pos >= vec_per_cu) pos -= vec_per_cu;
        }
        if (SMODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) stamps[(size_t)blockIdx.x * 16 + 8 + sw] = wall_clock64() - t0;
    }
    if (acc == 1.2345) sink[0] = acc;
}
template <int GW, int SW, int SU, int SMODE>
void run(const char* name, const double* t, unsigned mask, int giters, const d2* big, d2* bigw, size_t vec_per_cu, int siters, unsigned long long* stamps, double* sink, int blocks = 256) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f; std::vector<unsigned long long> h((size_t)blocks * 16);
    double g_us = 0, s_us = 0;
    for (int r = 0; r < 3; ++r) {
        CK(hipMemset(stamps, 0, (size_t)blocks * 16 * 8));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((mix<GW, SW, SU, SMODE>), dim3(blocks), dim3(512), 0, 0, t, mask, giters, big, bigw, vec_per_cu, siters, (unsigned)(4096 - 1), stamps, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) {
            best = ms; CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
            double gm = 0, sm = 0;
            for (int bl = 0; bl < blocks; ++bl) { unsigned long long g = 0, s2 = 0; for (int w = 0; w < 8; ++w) { g = std::max(g, h[(size_t)bl * 16 + w]); s2 = std::max(s2, h[(size_t)bl * 16 + 8 + w]); } gm += g * 0.01; sm += s2 * 0.01; }
            g_us = gm / blocks; s_us = sm / blocks;
        }
    }
    const double gathers_per_cu = GW ? (double)GW * 64 * 4 * giters : 0;
    const double sbytes_per_cu = SMODE ? (double)SW * 64 * 16 * SU * siters : 0;
    printf("%-58s kernel %7.3f ms | gather done %8.1f us (%.3f ns/gather/CU) | stream done %8.1f us (%6.1f GB/s/CU, %5.2f TB/s chip) | %.1f B streamed per gather\n", name, best, g_us, gathers_per_cu ? g_us * 1e3 / gathers_per_cu : 0.0, s_us, s_us ? sbytes_per_cu / s_us * 1e-3 : 0.0, s_us ? sbytes_per_cu * blocks / s_us * 1e-6 : 0.0, gathers_per_cu ? sbytes_per_cu / gathers_per_cu : 0.0);
}
int main() {
    const size_t ntab = 1 << 17;                                  // 1 MB table: L2-resident
    std::vector<double> h(ntab); for (size_t i = 0; i < ntab; ++i) h[i] = (double)(i & 1023);
    double *t, *sink; CK(hipMalloc(&t, ntab * 8)); CK(hipMalloc(&sink, 64)); CK(hipMemcpy(t, h.data(), ntab * 8, hipMemcpyHostToDevice));
    const size_t vec_per_cu = (size_t)1 << 20;                    // 16 MB per CU, 4 GB in all
    d2* big; CK(hipMalloc(&big, vec_per_cu * 256 * 16)); CK(hipMemset(big, 0, vec_per_cu * 256 * 16));
    unsigned long long* stamps; CK(hipMalloc(&stamps, 256 * 16 * 8));
    const unsigned mask = (unsigned)ntab - 1;
    const int gi = 1024;
    printf("# gather: 8-B loads, hash-random over a 1 MB table, 4 in flight per lane; stream: 16 B/lane nt, SU wave-instructions in flight per wave\n");
    run<8, 0, 1, 0>("8 waves gather, no stream", t, mask, gi, big, big, vec_per_cu, 0, stamps, sink);
    run<7, 0, 1, 0>("7 waves gather, no stream", t, mask, gi, big, big, vec_per_cu, 0, stamps, sink);
    run<4, 0, 1, 0>("4 waves gather, no stream", t, mask, gi, big, big, vec_per_cu, 0, stamps, sink);
    run<0, 1, 8, 1>("1 wave streams loads (8 in flight), no gather", t, mask, gi, big, big, vec_per_cu, 2048, stamps, sink);
    run<0, 1, 16, 1>("1 wave streams loads (16 in flight), no gather", t, mask, gi, big, big, vec_per_cu, 1024, stamps, sink);
    run<0, 4, 8, 1>("4 waves stream loads (8 in flight), no gather", t, mask, gi, big, big, vec_per_cu, 2048, stamps, sink);
    run<0, 1, 16, 2>("1 wave streams stores (16 in flight), no gather", t, mask, gi, big, big, vec_per_cu, 1024, stamps, sink);
    // mixes: the stream runs for the whole gather time (siters large enough), so 'gather done' shows the gather rate under load
    run<7, 1, 2, 1>("7 gather + 1 stream loads, 2 in flight", t, mask, gi, big, big, vec_per_cu, 8192, stamps, sink);
    run<7, 1, 8, 1>("7 gather + 1 stream loads, 8 in flight", t, mask, gi, big, big, vec_per_cu, 4096, stamps, sink);
    run<7, 1, 16, 1>("7 gather + 1 stream loads, 16 in flight", t, mask, gi, big, big, vec_per_cu, 2048, stamps, sink);
    run<4, 4, 8, 1>("4 gather + 4 stream loads, 8 in flight", t, mask, gi, big, big, vec_per_cu, 4096, stamps, sink);
    run<7, 1, 16, 2>("7 gather + 1 stream stores, 16 in flight", t, mask, gi, big, big, vec_per_cu, 2048, stamps, sink);
    run<7, 1, 16, 3>("7 gather + 1 stream loads from a 64 KB L2-resident buffer", t, mask, gi, big, big, vec_per_cu, 2048, stamps, sink);
    // the same with only a quarter of the CUs (no chip-level L2 / HBM contention): isolates the per-CU effect
    run<7, 0, 1, 0>("64 wg: 7 waves gather, no stream", t, mask, gi, big, big, vec_per_cu, 0, stamps, sink, 64);
    run<7, 1, 16, 1>("64 wg: 7 gather + 1 stream loads, 16 in flight", t, mask, gi, big, big, vec_per_cu, 2048, stamps, sink, 64);
    run<7, 1, 2, 1>("64 wg: 7 gather + 1 stream loads, 2 in flight", t, mask, gi, big, big, vec_per_cu, 8192, stamps, sink, 64);
    run<0, 1, 16, 1>("64 wg: 1 wave streams loads (16 in flight), no gather", t, mask, gi, big, big, vec_per_cu, 1024, stamps, sink, 64);
    return 0;
}
