// Exploration harness (NOT product code): per-lane register sort of K queries by table position, then ordered
// processing (step s of every lane touches about the same table region chip-wide).  Results are written in
// sorted order (wrong positions) -- timing only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
typedef double d2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(8))) ypair { double a, b; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ double blend(double xa, double ya, double xb, double yb, double q) {
    const double a = q - xa, b = xb - q; const double w = (a > 0.0) ? a / (a + b) : 0.0; return (1.0 - w) * ya + w * yb; }
__device__ __forceinline__ void cswap(double& a, double& b) { const double lo = fmin(a, b), hi = fmax(a, b); a = lo; b = hi; }
// Batcher odd-even merge sort network for K = 16 (63 compare-exchanges) on the query values themselves
template <int K> __device__ __forceinline__ void sortK(double (&q)[K]) {
#pragma unroll
    for (int p = 1; p < K; p <<= 1)
#pragma unroll
        for (int k = p; k >= 1; k >>= 1)
#pragma unroll
            for (int j = k % p; j + k < K; j += 2 * k)
#pragma unroll
                for (int i = 0; i < k; ++i)
                    if (i + j + k < K && (i + j) / (2 * p) == (i + j + k) / (2 * p)) cswap(q[i + j], q[i + j + k]);
}
template <int THREADS, int K, bool SORT>
__global__ __launch_bounds__(THREADS) void k(const double* __restrict__ y, int n, double dx, double inv_dx, const double* __restrict__ xq, double* __restrict__ yq, size_t nq) {
    const size_t T = (size_t)THREADS * K, ntiles = nq / T;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const d2* q2 = (const d2*)(xq + t * T); d2* o2 = (d2*)(yq + t * T);
        double q[K], r[K];
#pragma unroll
        for (int u = 0; u < K / 2; ++u) { d2 v = __builtin_nontemporal_load(q2 + threadIdx.x + u * THREADS); q[2 * u] = v.x; q[2 * u + 1] = v.y; }
        if (SORT) sortK<K>(q);
#pragma unroll
        for (int g = 0; g < K; g += 4) {
            int l[4]; ypair yp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { int i = (int)(q[g + u] * inv_dx); l[u] = min(max(i, 0), n - 2); }
#pragma unroll
            for (int u = 0; u < 4; ++u) yp[u] = *(const ypair*)(y + l[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) r[g + u] = blend(l[u] * dx, yp[u].a, (l[u] + 1) * dx, yp[u].b, q[g + u]);
        }
#pragma unroll
        for (int u = 0; u < K / 2; ++u) { d2 v; v.x = r[2 * u]; v.y = r[2 * u + 1]; __builtin_nontemporal_store(v, o2 + threadIdx.x + u * THREADS); }
    }
}
template <int THREADS, int K, bool SORT>
float run(const double* y, int n, const double* xq, double* yq, size_t nq, int blocks) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); double dx = 1.0 / (n - 1);
    hipLaunchKernelGGL((k<THREADS, K, SORT>), dim3(blocks), dim3(THREADS), 0, 0, y, n, dx, 1.0 / dx, xq, yq, nq); CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < 7; ++r) { CK(hipEventRecord(a)); hipLaunchKernelGGL((k<THREADS, K, SORT>), dim3(blocks), dim3(THREADS), 0, 0, y, n, dx, 1.0 / dx, xq, yq, nq); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end()); return ts[ts.size() / 2];
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1000000; const size_t nq = 100000000 / 65536 * 65536;
    std::vector<double> hy(n + 1); for (int i = 0; i <= n; ++i) hy[i] = sin(6.28 * i / n);
    std::vector<double> hq(nq); unsigned long long s = 12345; for (size_t i = 0; i < nq; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; hq[i] = (double)(s >> 11) * 0x1.0p-53; }
    double *y, *xq, *yq; CK(hipMalloc(&y, (n + 1) * 8)); CK(hipMalloc(&xq, nq * 8)); CK(hipMalloc(&yq, nq * 8));
    CK(hipMemcpy(y, hy.data(), (n + 1) * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(xq, hq.data(), nq * 8, hipMemcpyHostToDevice));
    printf("n=%d\n", n);
#define R(T, K, S, B) printf("threads %4d K %2d sort %d blocks %5d : %.4f ms\n", T, K, (int)S, B, run<T, K, S>(y, n, xq, yq, nq, B));
    R(256, 16, false, 1024) R(256, 16, true, 1024) R(256, 16, true, 2048) R(256, 16, true, 512) R(256, 16, true, 256)
    R(256, 32, true, 1024) R(256, 32, true, 512) R(256, 8, true, 1024) R(256, 8, true, 2048) R(512, 16, true, 512) R(1024, 16, true, 256) R(256, 4, true, 2048)
    // sorted input through the sorting kernel (cost of sorting what is already sorted)
    std::sort(hq.begin(), hq.end()); CK(hipMemcpy(xq, hq.data(), nq * 8, hipMemcpyHostToDevice));
    printf("globally sorted input:\n"); R(256, 16, false, 1024) R(256, 16, true, 1024)
    return 0;
}
