// Exploration harness (NOT product code), round 3: 1e8 random 32-B cell reads issued as SCALAR loads (s_load_dwordx8).
// Round 2 found that a scalar-load miss fills only a 64-B half line in L2 (profiles/r02_exp_table_prefetch.log) while a
// vector-load miss always fetches the whole 128-B line (profiles/r02_exp_large_table_cell_reads.log) -- config 3 pays
// 128 B of fabric traffic for every 32-B cell.  Question: does the scalar path move 1e8 cells
// faster than the vector path's 1.7-1.8 ms?  Every wave walks its 64 lane-indices with v_readlane and keeps NF scalar
// loads in flight (lgkmcnt allows 15 per wave); the data is only summed (SALU), so this is the memory path alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int CELLS>
__global__ __launch_bounds__(256) void vec(const d2* __restrict__ t, unsigned mask, double* __restrict__ out) {
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    d2 a[CELLS], b[CELLS];
#pragma unroll
    for (int u = 0; u < CELLS; ++u) { s = hash(s + u); const size_t c = (size_t)(s & mask) * 2; a[u] = t[c]; b[u] = t[c + 1]; }
    double acc = 0;
#pragma unroll
    for (int u = 0; u < CELLS; ++u) acc += a[u].x + a[u].y + b[u].x + b[u].y;
    if (acc == 1.2345) out[0] = acc;
}

// one wave: 64 lane indices, NF scalar loads in flight
template <int NF, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void sca(const unsigned* __restrict__ t, unsigned mask, int rounds, unsigned* __restrict__ out) {
    unsigned s = (blockIdx.x * (64 * WAVES) + threadIdx.x) * 2654435761u + 12345u;
    unsigned acc = 0;
    for (int r = 0; r < rounds; ++r) {
        s = hash(s + r);
        const unsigned off = (s & mask) * 32u;                     // byte offset of the lane's cell (table < 4 GiB)
#pragma unroll 1
        for (int l0 = 0; l0 < 64; l0 += NF) {
            u8v v[NF];
#pragma unroll
            for (int k = 0; k < NF; ++k) {
                const unsigned o = __builtin_amdgcn_readlane(off, l0 + k);
                asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(v[k]) : "s"(t), "s"(o) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < NF; ++k) acc += v[k].s0 + v[k].s3 + v[k].s4 + v[k].s7;
        }
    }
    if (acc == 0x12345u) out[0] = acc;
}

int main() {
    double* out; CK(hipMalloc(&out, 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (size_t mib : {(size_t)512, (size_t)128}) {
        const size_t cells = mib * (1 << 20) / 32;
        d2* t; CK(hipMalloc(&t, cells * 32)); CK(hipMemset(t, 0, cells * 32));
        const size_t nreads = 100000000;
        auto time = [&](const char* name, auto launch) {
            launch(); CK(hipDeviceSynchronize());
            std::vector<float> ts;
            for (int r = 0; r < 3; ++r) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
            std::sort(ts.begin(), ts.end());
            printf("%-44s table %4zu MiB : %7.3f ms per 1e8 cells = %5.1f G cells/s (128 B each: %.2f TB/s, 64 B each: %.2f TB/s)\n", name, mib, ts[1], 1e8 / ts[1] * 1e-6, 1e8 * 128 / ts[1] * 1e-9, 1e8 * 64 / ts[1] * 1e-9);
        };
        time("vector loads, 2 cells in flight per lane", [&] { hipLaunchKernelGGL((vec<2>), dim3((unsigned)(nreads / 512)), dim3(256), 0, 0, t, (unsigned)(cells - 1), out); });
        time("vector loads, 4 cells in flight per lane", [&] { hipLaunchKernelGGL((vec<4>), dim3((unsigned)(nreads / 1024)), dim3(256), 0, 0, t, (unsigned)(cells - 1), out); });
        // scalar: persistent-ish grids; each wave does `rounds` x 64 cells
        for (int wgs_per_cu : {2, 4}) {
            const unsigned blocks = 256u * wgs_per_cu;
            const int rounds = (int)(nreads / ((size_t)blocks * 256));   // 256 lanes (4 waves) per block
            char nm[96];
            snprintf(nm, sizeof nm, "scalar loads, 4 in flight/wave, %d waves/CU", 4 * wgs_per_cu);
            time(nm, [&] { hipLaunchKernelGGL((sca<4, 4>), dim3(blocks), dim3(256), 0, 0, (const unsigned*)t, (unsigned)(cells - 1), rounds, (unsigned*)out); });
            snprintf(nm, sizeof nm, "scalar loads, 8 in flight/wave, %d waves/CU", 4 * wgs_per_cu);
            time(nm, [&] { hipLaunchKernelGGL((sca<8, 4>), dim3(blocks), dim3(256), 0, 0, (const unsigned*)t, (unsigned)(cells - 1), rounds, (unsigned*)out); });
        }
        {
            const unsigned blocks = 256u * 2;                             // 16 waves per block, 2 blocks per CU = 32 waves/CU
            const int rounds = (int)(nreads / ((size_t)blocks * 1024));
            time("scalar loads, 8 in flight/wave, 32 waves/CU", [&] { hipLaunchKernelGGL((sca<8, 16>), dim3(blocks), dim3(1024), 0, 0, (const unsigned*)t, (unsigned)(cells - 1), rounds, (unsigned*)out); });
        }
        CK(hipFree(t));
    }
    return 0;
}
