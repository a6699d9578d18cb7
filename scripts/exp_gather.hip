// Exploration harness (NOT product code): variants of the random-gather inner loop, to find what bounds
// the 1e8-random-query case on MI355X.  hipcc --offload-arch=gfx950 -O3 -o exp_gather exp_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef int i4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(8))) ypair { double a, b; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ double blend(double xa, double ya, double xb, double yb, double q) {
    const double a = q - xa, b = xb - q;
    const double w = (a > 0.0) ? a / (a + b) : 0.0;
    return (1.0 - w) * ya + w * yb;
}
// VAR: 0 plain gather, 1 nt gather, 2 sc1 buffer gather, 3 two 8-B gathers, 4 XCD-split G=2, 5 no gather (ALU+stream only),
//      6 gather only 8 B (one load), 7 XCD-split G=4
template <int VAR, int UNROLL>
__global__ __launch_bounds__(256) void k(const double* __restrict__ y, int n, double dx, double inv_dx, const double* __restrict__ xq, double* __restrict__ yq, size_t nq) {
    const size_t nvec = nq >> 1;
    const d2* xv = (const d2*)xq; d2* yv = (d2*)yq;
    size_t nblk = gridDim.x, bid = blockIdx.x;
    int grp = 0, ngrp = 1;
    if (VAR == 4 || VAR == 7) {          // blocks b, b+8 share an XCD (round-robin dispatch); group = XCD / (8/ngrp)
        ngrp = (VAR == 4) ? 2 : 4;
        const int xcd = blockIdx.x & 7;
        grp = xcd / (8 / ngrp);
        // blocks of one group: renumber so that each group covers ALL queries
        const int per = 8 / ngrp;
        bid = (blockIdx.x >> 3) * per + (xcd % per);
        nblk = gridDim.x / ngrp;
    }
    const size_t stride = nblk * 256;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (n + 1) * 8, 0x00020000);
    for (size_t i = bid * 256 + threadIdx.x; i + (size_t)(UNROLL - 1) * stride < nvec; i += (size_t)UNROLL * stride) {
        double q[2 * UNROLL], r[2 * UNROLL]; int l[2 * UNROLL]; ypair yp[2 * UNROLL]; bool mine[2 * UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { d2 v = __builtin_nontemporal_load(xv + i + (size_t)u * stride); q[2*u] = v.x; q[2*u+1] = v.y; }
#pragma unroll
        for (int j = 0; j < 2 * UNROLL; ++j) {
            int t = (int)(q[j] * inv_dx); t = min(max(t, 0), n - 2);
            l[j] = t;
            mine[j] = (ngrp == 1) || ((int)(((long long)t * ngrp) / n) == grp);
        }
#pragma unroll
        for (int j = 0; j < 2 * UNROLL; ++j) {
            if (VAR == 0 || VAR == 4 || VAR == 7) { if (mine[j]) yp[j] = *(const ypair*)(y + l[j]); else { yp[j].a = 0; yp[j].b = 0; } }
            else if (VAR == 1) { yp[j].a = __builtin_nontemporal_load(y + l[j]); yp[j].b = __builtin_nontemporal_load(y + l[j] + 1); }
            else if (VAR == 2) { i4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, l[j] * 8, 0, 16); yp[j].a = __builtin_bit_cast(double, ((long long)(unsigned)w.y << 32) | (unsigned)w.x); yp[j].b = __builtin_bit_cast(double, ((long long)(unsigned)w.w << 32) | (unsigned)w.z); }
            else if (VAR == 3) { yp[j].a = y[l[j]]; yp[j].b = y[l[j] + 1]; }
            else if (VAR == 5) { yp[j].a = q[j]; yp[j].b = q[j] + 1; }
            else if (VAR == 6) { yp[j].a = y[l[j]]; yp[j].b = yp[j].a + 1; }
        }
#pragma unroll
        for (int j = 0; j < 2 * UNROLL; ++j) r[j] = blend(l[j] * dx, yp[j].a, (l[j] + 1) * dx, yp[j].b, q[j]);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (ngrp == 1) { d2 v; v.x = r[2*u]; v.y = r[2*u+1]; __builtin_nontemporal_store(v, yv + i + (size_t)u * stride); }
            else {
                double* o = (double*)(yv + i + (size_t)u * stride);
                if (mine[2*u] && mine[2*u+1]) { d2 v; v.x = r[2*u]; v.y = r[2*u+1]; __builtin_nontemporal_store(v, (d2*)o); }
                else if (mine[2*u]) __builtin_nontemporal_store(r[2*u], o);
                else if (mine[2*u+1]) __builtin_nontemporal_store(r[2*u+1], o + 1);
            }
        }
    }
}
template <int VAR, int UNROLL>
float run(const double* y, int n, const double* xq, double* yq, size_t nq, int blocks, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    double dx = 1.0 / (n - 1);
    hipLaunchKernelGGL((k<VAR, UNROLL>), dim3(blocks), dim3(256), 0, 0, y, n, dx, 1.0 / dx, xq, yq, nq);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a)); hipLaunchKernelGGL((k<VAR, UNROLL>), dim3(blocks), dim3(256), 0, 0, y, n, dx, 1.0 / dx, xq, yq, nq); CK(hipEventRecord(b));
        CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end()); return ts[ts.size() / 2];
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1000000; const size_t nq = 100000000;
    std::vector<double> hy(n + 1); for (int i = 0; i <= n; ++i) hy[i] = sin(6.28 * i / n);
    std::vector<double> hq(nq); unsigned long long s = 12345; for (size_t i = 0; i < nq; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; hq[i] = (double)(s >> 11) * 0x1.0p-53; }
    double *y, *xq, *yq, *yref; CK(hipMalloc(&y, (n + 1) * 8)); CK(hipMalloc(&xq, nq * 8)); CK(hipMalloc(&yq, nq * 8)); CK(hipMalloc(&yref, nq * 8));
    CK(hipMemcpy(y, hy.data(), (n + 1) * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(xq, hq.data(), nq * 8, hipMemcpyHostToDevice));
    printf("n=%d table %.1f MB\n", n, n * 8e-6);
#define R(V, U, B) printf("var %d unroll %d blocks %5d : %.4f ms\n", V, U, B, run<V, U>(y, n, xq, yq, nq, B, 7));
    R(0, 2, 2048) R(0, 1, 2048) R(0, 4, 2048) R(0, 2, 1024) R(0, 2, 4096) R(0, 2, 8192)
    R(1, 2, 2048) R(2, 2, 2048) R(3, 2, 2048) R(5, 2, 2048) R(6, 2, 2048) R(4, 2, 2048) R(4, 2, 4096) R(7, 2, 4096)
    // correctness of the split variants against var 0
    run<0, 2>(y, n, xq, yref, nq, 2048, 1);
    for (int v = 0; v < 2; ++v) {
        CK(hipMemset(yq, 0xff, nq * 8));
        if (v == 0) run<4, 2>(y, n, xq, yq, nq, 2048, 1); else run<7, 2>(y, n, xq, yq, nq, 4096, 1);
        std::vector<double> a(1 << 20), b(1 << 20); CK(hipMemcpy(a.data(), yq, a.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), yref, b.size() * 8, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < a.size(); ++i) bad += (a[i] != b[i]); printf("split variant %d mismatches in first 1M: %zu\n", v, bad);
    }
    return 0;
}
