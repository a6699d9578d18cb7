// Exploration harness (NOT product code), round 3: do SCALAR gathers (s_load_dwordx4 through the scalar data cache, a path
// of its own from the CU to L2) add to what the vector path of a CU can gather from an L2-resident table, or do both hit
// the same ceiling (the XCD L2's request rate)?  One 1024-lane workgroup per CU; VW waves gather with vector loads (16 B
// per lane, 4 in flight), SW waves with scalar loads (16 B each, NF in flight per wave, addresses walked with v_readlane);
// hash-random 16-B-aligned addresses in a TABLE_MB table.  Each role reports gathers / ns / CU from wall_clock64 stamps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int VW, int SW, int NF>
__global__ __launch_bounds__(1024) void k(const d2* __restrict__ t, unsigned mask, int viters, int siters, unsigned long long* __restrict__ stamps, double* __restrict__ sink)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    unsigned s = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 12345u;
    const unsigned long long t0 = wall_clock64();
    if (wave < VW) {
        double acc = 0;
        for (int it = 0; it < viters; ++it) {
            d2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { s = hash(s + u + it); v[u] = t[s & mask]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y;
        }
        if (acc == 1.2345) sink[0] = acc;
        if (lane == 0) stamps[(size_t)blockIdx.x * 32 + wave] = wall_clock64() - t0;
    } else if (wave >= 16 - SW) {
        unsigned acc = 0;
        for (int it = 0; it < siters; ++it) {
            s = hash(s + it);
            const unsigned off = (s & mask) * 16u;
#pragma unroll 1
            for (int l0 = 0; l0 < 64; l0 += NF) {
                u4v v[NF];
#pragma unroll
                for (int q = 0; q < NF; ++q) {
                    const unsigned o = __builtin_amdgcn_readlane(off, l0 + q);
                    asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(v[q]) : "s"(t), "s"(o) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int q = 0; q < NF; ++q) acc += v[q].x + v[q].w;
            }
        }
        if (acc == 0x12345u) sink[1] = acc;
        if (lane == 0) stamps[(size_t)blockIdx.x * 32 + 16 + (wave - (16 - SW))] = wall_clock64() - t0;
    }
}

template <int VW, int SW, int NF>
void run(const char* name, const d2* t, unsigned mask, unsigned long long* stamps, double* sink)
{
    const int viters = VW ? 512 : 0, siters = SW ? 64 : 0;      // vector: 512 x 4 x 64 lanes per wave; scalar: 64 x 64 per wave
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<unsigned long long> h(256 * 32);
    float best = 1e9f; double vns = 0, sns = 0;
    for (int r = 0; r < 3; ++r) {
        CK(hipMemset(stamps, 0, 256 * 32 * 8));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<VW, SW, NF>), dim3(256), dim3(1024), 0, 0, t, mask, viters, siters, stamps, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) {
            best = ms; CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
            double vm = 0, sm = 0;
            for (int bl = 0; bl < 256; ++bl) { unsigned long long v = 0, s2 = 0; for (int w = 0; w < 16; ++w) { v = std::max(v, h[bl * 32 + w]); s2 = std::max(s2, h[bl * 32 + 16 + w]); } vm += v * 10.0; sm += s2 * 10.0; }
            vns = vm / 256; sns = sm / 256;
        }
    }
    const double vg = (double)VW * 64 * 4 * viters, sg = (double)SW * 64 * siters;
    printf("%-46s kernel %7.3f ms | vector: %9.0f gathers/CU in %8.0f ns = %.3f /ns/CU | scalar: %8.0f in %8.0f ns = %.3f /ns/CU\n", name, best, vg, vns, vns ? vg / vns : 0.0, sg, sns, sns ? sg / sns : 0.0);
}

int main(int argc, char** argv)
{
    const size_t mb = argc > 1 ? atoi(argv[1]) : 2;
    const size_t n = mb * (1 << 20) / 16;
    d2* t; CK(hipMalloc(&t, n * 16)); CK(hipMemset(t, 0, n * 16));
    unsigned long long* stamps; CK(hipMalloc(&stamps, 256 * 32 * 8));
    double* sink; CK(hipMalloc(&sink, 64));
    const unsigned mask = (unsigned)n - 1;
    printf("# table %zu MiB, 256 workgroups of 16 waves\n", mb);
    run<8, 0, 4>("8 vector waves", t, mask, stamps, sink);
    run<12, 0, 4>("12 vector waves", t, mask, stamps, sink);
    run<16, 0, 4>("16 vector waves", t, mask, stamps, sink);
    run<0, 8, 8>("8 scalar waves, 8 in flight", t, mask, stamps, sink);
    run<0, 16, 8>("16 scalar waves, 8 in flight", t, mask, stamps, sink);
    run<8, 8, 8>("8 vector + 8 scalar waves", t, mask, stamps, sink);
    run<12, 4, 8>("12 vector + 4 scalar waves", t, mask, stamps, sink);
    return 0;
}
