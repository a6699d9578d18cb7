// Exploration harness (NOT product code): premise test of a "region sweep".  Persistent workgroups (one per CU)
// walk tiles of T queries; the host pre-sorts every tile by table position, emulating an in-LDS binning pass.
// If CUs stay roughly aligned, L2 only has to hold the table region currently being swept.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
struct __attribute__((packed, aligned(8))) ypair { double a, b; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ double blend(double xa, double ya, double xb, double yb, double q) {
    const double a = q - xa, b = xb - q; const double w = (a > 0.0) ? a / (a + b) : 0.0; return (1.0 - w) * ya + w * yb; }
template <int THREADS, int UNR>
__global__ __launch_bounds__(THREADS) void k(const double* __restrict__ y, int n, double dx, double inv_dx, const double* __restrict__ xq, double* __restrict__ yq, size_t nq, int T) {
    const size_t ntiles = nq / T;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const double* q = xq + t * T; double* o = yq + t * T;
        for (int j = threadIdx.x; j + (UNR - 1) * THREADS < T; j += UNR * THREADS) {
            double qq[UNR]; int l[UNR]; ypair yp[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) qq[u] = __builtin_nontemporal_load(q + j + u * THREADS);
#pragma unroll
            for (int u = 0; u < UNR; ++u) { int i = (int)(qq[u] * inv_dx); l[u] = min(max(i, 0), n - 2); }
#pragma unroll
            for (int u = 0; u < UNR; ++u) yp[u] = *(const ypair*)(y + l[u]);
#pragma unroll
            for (int u = 0; u < UNR; ++u) __builtin_nontemporal_store(blend(l[u] * dx, yp[u].a, (l[u] + 1) * dx, yp[u].b, qq[u]), o + j + u * THREADS);
        }
    }
}
template <int THREADS, int UNR>
float run(const double* y, int n, const double* xq, double* yq, size_t nq, int T, int blocks) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); double dx = 1.0 / (n - 1);
    hipLaunchKernelGGL((k<THREADS, UNR>), dim3(blocks), dim3(THREADS), 0, 0, y, n, dx, 1.0 / dx, xq, yq, nq, T); CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < 7; ++r) { CK(hipEventRecord(a)); hipLaunchKernelGGL((k<THREADS, UNR>), dim3(blocks), dim3(THREADS), 0, 0, y, n, dx, 1.0 / dx, xq, yq, nq, T); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end()); return ts[ts.size() / 2];
}
int main() {
    const int n = 1000000; const size_t nq = 100000000 / 16384 * 16384;
    std::vector<double> hy(n + 1); for (int i = 0; i <= n; ++i) hy[i] = sin(6.28 * i / n);
    std::vector<double> hq(nq); unsigned long long s = 12345; for (size_t i = 0; i < nq; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; hq[i] = (double)(s >> 11) * 0x1.0p-53; }
    double *y, *xq, *yq; CK(hipMalloc(&y, (n + 1) * 8)); CK(hipMalloc(&xq, nq * 8)); CK(hipMalloc(&yq, nq * 8));
    CK(hipMemcpy(y, hy.data(), (n + 1) * 8, hipMemcpyHostToDevice));
    for (int pass = 0; pass < 3; ++pass) {
        const int T = 16384;
        if (pass == 1) for (size_t t = 0; t < nq / T; ++t) std::sort(hq.begin() + t * T, hq.begin() + (t + 1) * T);            // every tile fully sorted
        if (pass == 2) { // coarse binning only: 64 bins, order inside a bin random (what an LDS counting sort would give)
            unsigned long long r = 999;
            for (size_t t = 0; t < nq / T; ++t) { auto b = hq.begin() + t * T; std::sort(b, b + T, [&](double u, double v) { return (int)(u * 64) < (int)(v * 64); }); }
            (void)r;
        }
        CK(hipMemcpy(xq, hq.data(), nq * 8, hipMemcpyHostToDevice));
        const char* name = pass == 0 ? "random tiles     " : pass == 1 ? "tiles sorted     " : "tiles 64-binned  ";
        printf("%s T=16384: 256x1024 unr4 %.4f ms | 256x1024 unr8 %.4f | 512x512 unr4 %.4f | 512x1024 unr4 %.4f | 1024x256 unr4 %.4f\n", name,
               run<1024, 4>(y, n, xq, yq, nq, T, 256), run<1024, 8>(y, n, xq, yq, nq, T, 256), run<512, 4>(y, n, xq, yq, nq, T, 512), run<1024, 4>(y, n, xq, yq, nq, T, 512), run<256, 4>(y, n, xq, yq, nq, T, 1024));
    }
    return 0;
}
