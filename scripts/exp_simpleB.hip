// Exploration harness (NOT product code), round 3: gathers in flight per lane in the region sweep's rounds.
// With 8 % of the gathers missing L2, nearly every 64-lane gather instruction contains a miss (0.92^64 = 0.5 % do not), so
// a batch of B rounds completes after one miss latency whatever B is: is the gather phase bound by (rounds / B) x miss
// latency?  The simple (one phase after the other) kernel has 256 registers per lane: B = 4 (product), 8, 16, 32.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mi_interp1_sweep.hpp"

using namespace mi_interp1;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE, int FORMULA, int B>
__global__ __launch_bounds__(kSweepThreads) void simple_kernel(G1Dev g, const double* __restrict__ xq,
                                                                      double* __restrict__ yq, size_t ntiles,
                                                                      double extrap, double bscale,
                                                                      const int* __restrict__ order_flag,
                                                                      size_t tail, ProbeArgs probe)
{
    __shared__ double sq[kSweepTile];
    __shared__ unsigned hist[kSweepBins];
    const int tid = threadIdx.x;
    if (*order_flag != 0) return;            // queries already ordered locally: the streaming kernel does the work
    // The last workgroup has the fewest tiles: it also probes the query order for the next call (one wave, while
    // the others wait for their first tile) and evaluates the ragged tail after its tiles.
    const bool last_wg = blockIdx.x == gridDim.x - 1;
    if (probe.host_mailbox && last_wg && tid < 64) order_probe_wave(probe);
    bool rev = false;                        // regions are swept up, down, up, ...: L2 still holds the turn-around half
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x, rev = !rev) {
        const d2* q2 = reinterpret_cast<const d2*>(xq + t * kSweepTile);
        d2* o2 = reinterpret_cast<d2*>(yq + t * kSweepTile);
        double q[kSweepK];
#pragma unroll
        for (int u = 0; u < kSweepK / 2; ++u) {
            const d2 v = stream_load(q2 + tid + u * kSweepThreads);
            q[2 * u] = v.x;
            q[2 * u + 1] = v.y;
        }
        for (int b = tid; b < kSweepBins; b += kSweepThreads) hist[b] = 0;
        __syncthreads();
        unsigned short bin[kSweepK], rank[kSweepK];
#pragma unroll
        for (int u = 0; u < kSweepK; ++u) {
            const int b = sweep_bin(q[u], g.xmin, bscale);
            bin[u] = (unsigned short)b;
            rank[u] = (unsigned short)atomicAdd(&hist[b], 1u);
        }
        __syncthreads();
        if (tid < 64) {                                   // exclusive prefix over the regions (one wave, 64 at a time)
            unsigned run = 0;
#pragma unroll
            for (int base = 0; base < kSweepBins; base += 64) {
                const unsigned v = hist[base + tid];
                unsigned incl = v;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned o = __shfl_up(incl, off, 64);
                    if (tid >= off) incl += o;
                }
                hist[base + tid] = run + incl - v;
                run += __shfl(incl, 63, 64);
            }
        }
        __syncthreads();
        unsigned short sp[kSweepK];
#pragma unroll
        for (int u = 0; u < kSweepK; ++u) {
            sp[u] = (unsigned short)(hist[bin[u]] + rank[u]);
            sq[sp[u]] = q[u];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kSweepK; u += B) {
            double qq[B], rr[B];
#pragma unroll
            for (int w = 0; w < B; ++w) {
                const int p = tid + (u + w) * kSweepThreads;
                qq[w] = sq[rev ? kSweepTile - 1 - p : p];
            }
            eval_batch<MODE, B, FORMULA, kSweepWin>(g, qq, rr, extrap);
#pragma unroll
            for (int w = 0; w < B; ++w) {
                const int p = tid + (u + w) * kSweepThreads;
                sq[rev ? kSweepTile - 1 - p : p] = rr[w];
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kSweepK / 2; ++u) {
            d2 v;
            v.x = sq[sp[2 * u]];
            v.y = sq[sp[2 * u + 1]];
            stream_store(v, o2 + tid + u * kSweepThreads);
        }
        __syncthreads();   // the next tile's scatter reuses sq
    }
    if (tail && last_wg) {
        const double* tq = xq + ntiles * kSweepTile;
        double* to = yq + ntiles * kSweepTile;
        double q[kSweepK];                   // tail < one tile: all loads in flight at once, one latency
#pragma unroll
        for (int u = 0; u < kSweepK; ++u) {
            const size_t i = (size_t)tid + (size_t)u * kSweepThreads;
            q[u] = i < tail ? tq[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kSweepK; u += 4) {
            const double qq[4] = {q[u], q[u + 1], q[u + 2], q[u + 3]};
            double rr[4];
            eval_batch<MODE, 4, FORMULA, kSweepWin>(g, qq, rr, extrap);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const size_t i = (size_t)tid + (size_t)(u + w) * kSweepThreads;
                if (i < tail) to[i] = rr[w];
            }
        }
    }
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
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        std::vector<float> ts;
        for (int r = 0; r < 7; ++r) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 5; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / 5);
        }
        std::sort(ts.begin(), ts.end());
        printf("%-58s %.4f ms (min %.4f, max %.4f)  %.1f %% of 8 TB/s\n", name, ts[3], ts[0], ts[6], (16.0 * nq + 8e6) / (ts[3] * 1e-3) / 8e12 * 100);
        return ts[3];
    };
    time("product pipelined kernel", [&] {
        hipLaunchKernelGGL((interp1_sweep_pipe_kernel<0, 3>), dim3(256), dim3(kPipeThreads), 0, 0, g, xq, ya, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{});
    });
    time("simple kernel, B = 4 gathers in flight per lane", [&] {
        hipLaunchKernelGGL((simple_kernel<0, 3, 4>), dim3(256), dim3(kSweepThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{});
    });
    time("simple kernel, B = 8", [&] {
        hipLaunchKernelGGL((simple_kernel<0, 3, 8>), dim3(256), dim3(kSweepThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{});
    });
    time("simple kernel, B = 16", [&] {
        hipLaunchKernelGGL((simple_kernel<0, 3, 16>), dim3(256), dim3(kSweepThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{});
    });
    time("simple kernel, B = 32", [&] {
        hipLaunchKernelGGL((simple_kernel<0, 3, 32>), dim3(256), dim3(kSweepThreads), 0, 0, g, xq, yb, ntiles, __builtin_nan(""), bscale, flag, (size_t)0, ProbeArgs{});
    });
    std::vector<double> a(1 << 22), b(1 << 22);
    size_t bad = 0;
    for (size_t off = 0; off < ntiles * kSweepTile; off += a.size()) {
        const size_t m = std::min(a.size(), ntiles * kSweepTile - off);
        CK(hipMemcpy(a.data(), ya + off, m * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), yb + off, m * 8, hipMemcpyDeviceToHost));
        if (memcmp(a.data(), b.data(), m * 8) != 0) for (size_t i = 0; i < m; ++i) bad += memcmp(&a[i], &b[i], 8) != 0;
    }
    printf("outputs: %zu of %zu differ\n", bad, ntiles * kSweepTile);
    return 0;
}
