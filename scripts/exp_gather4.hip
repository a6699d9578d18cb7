// Exploration harness (NOT product code), round 2: what does one random 32-B cell read from a table far larger than L2
// cost, and does a cache-policy bit (nt / sc0 / sc1) or an uncached allocation make the L2 fetch less than a 128-B line?
// BASELINE configs[2] (4096^2 bilinear, quad-cell table of 512 MiB) reads exactly that: 1e8 random cells.  The sweep
// kernel's counters (profiles/r02_sweep_kernel_counters.json) show every fabric read of a miss is a 128-B request.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int POLICY>
__device__ __forceinline__ d2 ld16(const d2* p) {
    d2 v;
    if (POLICY == 0) v = *p;
    else if (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    else if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    else if (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    else if (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// every lane reads CELLS random 32-B cells (two 16-B loads each), all in flight together
template <int POLICY, int CELLS>
__global__ __launch_bounds__(256) void g(const d2* __restrict__ t, unsigned mask, double* __restrict__ out) {
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    d2 a[CELLS], b[CELLS];
#pragma unroll
    for (int u = 0; u < CELLS; ++u) { s = hash(s + u); const size_t c = (size_t)(s & mask) * 2; a[u] = ld16<POLICY>(t + c); b[u] = ld16<POLICY>(t + c + 1); }
    if (POLICY != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    double acc = 0;
#pragma unroll
    for (int u = 0; u < CELLS; ++u) acc += a[u].x + a[u].y + b[u].x + b[u].y;
    if (acc == 1.2345) out[0] = acc;
}
template <int POLICY>
void run(const char* name, const d2* t, size_t cells, double* out) {
    const size_t nreads = 100000000; const int CELLS = 2;
    const unsigned blocks = (unsigned)(nreads / (256 * CELLS));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((g<POLICY, CELLS>), dim3(blocks), dim3(256), 0, 0, t, (unsigned)(cells - 1), out); CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < 3; ++r) { CK(hipEventRecord(a)); hipLaunchKernelGGL((g<POLICY, CELLS>), dim3(blocks), dim3(256), 0, 0, t, (unsigned)(cells - 1), out); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    printf("%-34s table %6.0f MiB : %.3f ms per 1e8 cells = %5.1f G cells/s; if 128 B per cell left HBM: %.2f TB/s, if 64 B: %.2f TB/s\n", name, cells * 32.0 / (1 << 20), ts[1], 1e8 / ts[1] * 1e-6, 1e8 * 128 / ts[1] * 1e-9, 1e8 * 64 / ts[1] * 1e-9);
}
int main() {
    double* out; CK(hipMalloc(&out, 64));
    for (size_t mib : {(size_t)512, (size_t)128}) {
        const size_t cells = mib * (1 << 20) / 32;
        d2* t; CK(hipMalloc(&t, cells * 32)); CK(hipMemset(t, 0, cells * 32));
        run<0>("plain", t, cells, out);
        run<1>("nt", t, cells, out);
        run<2>("sc1", t, cells, out);
        run<5>("sc0", t, cells, out);
        run<3>("sc0 sc1", t, cells, out);
        run<4>("sc0 sc1 nt", t, cells, out);
        CK(hipFree(t));
        d2* u = nullptr;
        if (hipExtMallocWithFlags((void**)&u, cells * 32, hipDeviceMallocUncached) == hipSuccess) {
            CK(hipMemset(u, 0, cells * 32));
            run<0>("uncached allocation, plain", u, cells, out);
            run<1>("uncached allocation, nt", u, cells, out);
            CK(hipFree(u));
        } else { printf("hipDeviceMallocUncached not available\n"); (void)hipGetLastError(); }
        d2* f = nullptr;
        if (hipExtMallocWithFlags((void**)&f, cells * 32, hipDeviceMallocFinegrained) == hipSuccess) {
            CK(hipMemset(f, 0, cells * 32));
            run<0>("fine-grained allocation, plain", f, cells, out);
            CK(hipFree(f));
        } else { printf("hipDeviceMallocFinegrained not available\n"); (void)hipGetLastError(); }
    }
    return 0;
}
