// Exploration harness (NOT product code): what streaming shape gets closest to the HBM roofline for
// "8 B in + 8 B out per element" on MI355X?  hipcc --offload-arch=gfx950 -O3 -o exp_stream exp_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
// NT: 0 plain, 1 nt loads+stores, 2 nt stores only, 3 nt loads only.  CONTIG: each block owns a contiguous chunk.
template <int NT, int UNROLL, bool CONTIG>
__global__ __launch_bounds__(256) void k(const d2* __restrict__ x, d2* __restrict__ y, size_t nvec) {
    size_t stride, i, end;
    if (CONTIG) { const size_t per = (nvec + gridDim.x - 1) / gridDim.x; i = blockIdx.x * per + threadIdx.x; end = min(nvec, (blockIdx.x + 1) * per); stride = 256; }
    else { stride = (size_t)gridDim.x * 256; i = (size_t)blockIdx.x * 256 + threadIdx.x; end = nvec; }
    for (; i + (size_t)(UNROLL - 1) * stride < end; i += (size_t)UNROLL * stride) {
        d2 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = (NT == 1 || NT == 3) ? __builtin_nontemporal_load(x + i + (size_t)u * stride) : x[i + (size_t)u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { v[u].x = v[u].x * 1.5 + 0.25; v[u].y = v[u].y * 1.5 + 0.25; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { if (NT == 1 || NT == 2) __builtin_nontemporal_store(v[u], y + i + (size_t)u * stride); else y[i + (size_t)u * stride] = v[u]; }
    }
    for (; i < end; i += stride) { d2 v = x[i]; v.x = v.x * 1.5 + 0.25; v.y = v.y * 1.5 + 0.25; y[i] = v; }
}
template <int NT, int UNROLL, bool CONTIG>
float run(const d2* x, d2* y, size_t nvec, int blocks) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<NT, UNROLL, CONTIG>), dim3(blocks), dim3(256), 0, 0, x, y, nvec); CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < 9; ++r) { CK(hipEventRecord(a)); hipLaunchKernelGGL((k<NT, UNROLL, CONTIG>), dim3(blocks), dim3(256), 0, 0, x, y, nvec); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end()); return ts[ts.size() / 2];
}
int main() {
    const size_t n = 100000000, nvec = n / 2; d2 *x, *y; CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMemset(x, 0x11, n * 8));
#define R(NT, U, C, B) { float ms = run<NT, U, C>(x, y, nvec, B); printf("nt %d unroll %d contig %d blocks %6d : %.4f ms  %.0f GB/s\n", NT, U, (int)C, B, ms, 1.6e9 / ms / 1e6); }
    int full = (int)((nvec + 255) / 256);
    R(1,1,false,2048) R(1,2,false,2048) R(1,4,false,2048) R(1,8,false,2048)
    R(1,2,false,512) R(1,2,false,1024) R(1,2,false,4096) R(1,2,false,8192) R(1,2,false,16384) R(1,1,false,full) R(1,2,false,full/2)
    R(0,2,false,2048) R(0,4,false,2048) R(0,1,false,full) R(2,2,false,2048) R(3,2,false,2048) R(2,4,false,4096) R(3,4,false,4096)
    R(1,2,true,2048) R(1,4,true,2048) R(1,4,true,4096) R(0,4,true,2048) R(1,4,false,4096) R(1,4,false,8192) R(1,8,false,4096)
    return 0;
}
