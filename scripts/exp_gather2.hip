// Exploration harness (NOT product code): raw divergent-gather rate of one CU's vector memory path.
// Every lane of every wave loads from its own random line of a table of a given size; no stream, no stores.
// Reports ns and clocks per lane-address per CU for tables that live in L1 (16 KB), L2 (256 KB - 2 MB) or beyond (8 MB+).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int WIDTH, int UNROLL>
__global__ __launch_bounds__(512) void g(const double* __restrict__ t, unsigned mask, int iters, double* __restrict__ out) {
    unsigned s = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u;
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
        double v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            s = hash(s + u + it);
            const unsigned i = s & mask;
            if (WIDTH == 8) v[u] = t[i];
            else if (WIDTH == 16) { const d2 w = *(const d2*)(t + (i & ~1u)); v[u] = w.x + w.y; }
            else { v[u] = (double)((const float*)t)[i]; }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    if (acc == 1.2345) out[0] = acc;
}
template <int WIDTH, int UNROLL>
void run(const char* name, const double* t, size_t n_elems, int blocks, double* out) {
    const int iters = 4096 / UNROLL;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((g<WIDTH, UNROLL>), dim3(blocks), dim3(512), 0, 0, t, (unsigned)(n_elems - 1), iters, out); CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < 3; ++r) { CK(hipEventRecord(a)); hipLaunchKernelGGL((g<WIDTH, UNROLL>), dim3(blocks), dim3(512), 0, 0, t, (unsigned)(n_elems - 1), iters, out); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    const double lane_addr_per_cu = 512.0 * 4096 * (blocks / 256.0);
    const double ns = ts[1] * 1e6 / lane_addr_per_cu;
    printf("%-28s table %8.0f KB blocks %4d : %.3f ms  %.3f ns/lane-address/CU (%.2f clk at 2.4 GHz)  %.1f G gathers/s chip\n", name, n_elems * 8 / 1024.0, blocks, ts[1], ns, ns * 2.4, 512.0 * 4096 * blocks / (ts[1] * 1e6));
}
int main() {
    const size_t nmax = 1 << 24;   // 128 MB
    std::vector<double> h(nmax); for (size_t i = 0; i < nmax; ++i) h[i] = (double)(i & 1023);
    double *t, *out; CK(hipMalloc(&t, nmax * 8)); CK(hipMalloc(&out, 64)); CK(hipMemcpy(t, h.data(), nmax * 8, hipMemcpyHostToDevice));
    for (size_t n : {(size_t)1 << 11, (size_t)1 << 15, (size_t)1 << 17, (size_t)1 << 18, (size_t)1 << 20, (size_t)1 << 24}) {
        run<8, 4>("8-B loads, 4 in flight", t, n, 256, out);
        run<8, 8>("8-B loads, 8 in flight", t, n, 256, out);
        run<16, 4>("16-B aligned, 4 in flight", t, n, 256, out);
        run<4, 4>("4-B loads, 4 in flight", t, n, 256, out);
    }
    run<8, 4>("8-B, 2 wg/CU", t, (size_t)1 << 17, 512, out);
    run<8, 4>("8-B, 4 wg/CU", t, (size_t)1 << 17, 1024, out);
    run<8, 4>("8-B, half the CUs", t, (size_t)1 << 17, 128, out);
    run<8, 4>("8-B, 32 wg", t, (size_t)1 << 17, 32, out);
    run<8, 4>("8-B, 8 wg (one per XCD)", t, (size_t)1 << 17, 8, out);
    return 0;
}
