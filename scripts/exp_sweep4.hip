// Exploration harness (NOT product code): phase timing of the region-sweep tile loop (wall_clock64, 100 MHz).
// Same loop as interp1_sweep_kernel (512 threads x 32 queries, 256 regions, one workgroup per CU); thread 0 of
// every workgroup accumulates the time between phase boundaries.  WAITST=1 also drains the stores inside the tile.
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
#ifdef EXP_FOLD
__constant__ int g_fold = 0x7fffffff;   // exp_sweep6: the gather address is folded into a smaller footprint (index & g_fold)
#endif
template <int GATHER>
__device__ __forceinline__ double evalq(const double* __restrict__ y, int n, double dx, double inv_dx, double q) {
    int i = (int)(q * inv_dx); i = min(max(i, 0), n - 2);
    ypair yp;
#ifdef EXP_FOLD
    if (GATHER == 1) yp = *(const ypair*)(y + (i & g_fold));
    else
#endif
    if (GATHER == 1) yp = *(const ypair*)(y + i);                      // one unaligned 16-B load
    else if (GATHER == 2) { yp.a = y[i]; yp.b = y[i + 1]; }            // two 8-B loads
    else if (GATHER == 3) { const d2 v = *(const d2*)(y + 2 * (size_t)i); yp.a = v.x; yp.b = v.y; }   // aligned pair table
    else if (GATHER == 4) { yp.a = y[i]; yp.b = yp.a + 1.0; }          // one 8-B load (rate probe, wrong result)
    else { yp.a = (double)i; yp.b = (double)(i + 1); }
    return blend(i * dx, yp.a, (i + 1) * dx, yp.b, q); }
constexpr int NPH = 8;
static unsigned* roles = nullptr;
#ifndef DELAY_TICKS
#define DELAY_TICKS 1500
#endif
template <int THREADS, int K, int NB, int GATHER, int WAITST, int TIMED, int G = 4, int ORDER = 0>
__global__ __launch_bounds__(THREADS) void k(const double* __restrict__ y, int n, double dx, double inv_dx, const double* __restrict__ xq, double* __restrict__ yq, size_t nq, unsigned long long* __restrict__ ph, unsigned* __restrict__ roles) {
    constexpr int T = THREADS * K;
    __shared__ double sq[T];
    __shared__ unsigned hist[NB];
    const size_t ntiles = nq / T;
    const double bscale = (double)NB;
    unsigned long long acc[NPH] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i) if (TIMED) { const unsigned long long now = wall_clock64(); acc[i] += now - last; last = now; }
    __shared__ int role_s;
    if (ORDER >= 7) {
        if (threadIdx.x == 0) {
            const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
            const unsigned key = (xcc & 15u) * 256u + ((hwid >> 8) & 255u);
            role_s = (int)(atomicAdd(&roles[key], 1u) & 1u);
        }
        __syncthreads();
    }
    const bool delayed = (ORDER >= 7) ? (role_s != 0) : ((ORDER >= 4) && (blockIdx.x >= gridDim.x / 2));
    const bool groupB = (ORDER == 6) ? false : (ORDER >= 7) ? delayed : (ORDER >= 4) ? (blockIdx.x >= gridDim.x / 2) : ((ORDER >= 2) && (((blockIdx.x >> 3) & 1) != 0));
    if ((ORDER >= 4) ? delayed : groupB) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < (unsigned long long)DELAY_TICKS) __builtin_amdgcn_s_sleep(32); }
    unsigned long long last = TIMED ? wall_clock64() : 0;
    int it = 0;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x, ++it) {
        const bool rev = (ORDER == 1 || ORDER == 6 || ORDER == 8) ? (it & 1) : (ORDER == 2 ? groupB : ((ORDER == 3 || ORDER == 5 || ORDER == 7) ? (groupB != (bool)(it & 1)) : (ORDER == 4 ? groupB : false)));
        const d2* q2 = (const d2*)(xq + t * T); d2* o2 = (d2*)(yq + t * T);
        double q[K];
#pragma unroll
        for (int u = 0; u < K / 2; ++u) { d2 v = __builtin_nontemporal_load(q2 + threadIdx.x + u * THREADS); q[2 * u] = v.x; q[2 * u + 1] = v.y; }
        for (int b = threadIdx.x; b < NB; b += THREADS) hist[b] = 0;
        if (TIMED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        STAMP(0)   // load (+ store drain of the previous tile: vmcnt is in order)
        unsigned short bin[K], rank[K];
#pragma unroll
        for (int u = 0; u < K; ++u) { int b = (int)(q[u] * bscale); b = min(max(b, 0), NB - 1); bin[u] = (unsigned short)b; rank[u] = (unsigned short)atomicAdd(&hist[b], 1u); }
        __syncthreads();
        STAMP(1)   // histogram
        if (threadIdx.x < 64) {
            unsigned run = 0;
            for (int base = 0; base < NB; base += 64) {
                const int b = base + threadIdx.x; unsigned v = (b < NB) ? hist[b] : 0, incl = v;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { unsigned o = __shfl_up(incl, off, 64); if ((int)threadIdx.x >= off) incl += o; }
                if (b < NB) hist[b] = run + incl - v;
                run += __shfl(incl, 63, 64);
            }
        }
        __syncthreads();
        STAMP(2)   // prefix
        unsigned short sp[K];
#pragma unroll
        for (int u = 0; u < K; ++u) { sp[u] = (unsigned short)(hist[bin[u]] + rank[u]); sq[sp[u]] = q[u]; }
        __syncthreads();
        STAMP(3)   // scatter
#pragma unroll
        for (int u = 0; u < K; u += G) {
            double qq[G], rr[G];
#pragma unroll
            for (int w = 0; w < G; ++w) qq[w] = sq[rev ? T - 1 - (threadIdx.x + (u + w) * THREADS) : threadIdx.x + (u + w) * THREADS];
#pragma unroll
            for (int w = 0; w < G; ++w) rr[w] = evalq<GATHER>(y, n, dx, inv_dx, qq[w]);
#pragma unroll
            for (int w = 0; w < G; ++w) sq[rev ? T - 1 - (threadIdx.x + (u + w) * THREADS) : threadIdx.x + (u + w) * THREADS] = rr[w];
        }
        __syncthreads();
        STAMP(4)   // gather + blend
#pragma unroll
        for (int u = 0; u < K / 2; ++u) { d2 v; v.x = sq[sp[2 * u]]; v.y = sq[sp[2 * u + 1]]; __builtin_nontemporal_store(v, o2 + threadIdx.x + u * THREADS); }
        if (WAITST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        STAMP(5)   // read back + store issue (+ drain when WAITST)
    }
    if (TIMED && threadIdx.x == 0) for (int i = 0; i < NPH; ++i) ph[blockIdx.x * NPH + i] = acc[i];
}
template <int THREADS, int K, int NB, int GATHER, int WAITST, int TIMED, int G = 4, int ORDER = 0>
float run(const char* name, const double* y, int n, const double* xq, double* yq, size_t nq, int blocks, unsigned long long* ph) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); double dx = 1.0 / (n - 1);
    CK(hipMemset(roles, 0, 4096 * 4)); hipLaunchKernelGGL((k<THREADS, K, NB, GATHER, WAITST, TIMED, G, ORDER>), dim3(blocks), dim3(THREADS), 0, 0, y, n, dx, 1.0 / dx, xq, yq, nq, ph, roles); CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < 5; ++r) { CK(hipMemsetAsync(roles, 0, 4096 * 4)); CK(hipEventRecord(a)); hipLaunchKernelGGL((k<THREADS, K, NB, GATHER, WAITST, TIMED, G, ORDER>), dim3(blocks), dim3(THREADS), 0, 0, y, n, dx, 1.0 / dx, xq, yq, nq, ph, roles); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    printf("%-34s threads %4d K %2d bins %3d blocks %4d : %.4f ms", name, THREADS, K, NB, blocks, ts[ts.size() / 2]);
    if (TIMED) {
        std::vector<unsigned long long> h((size_t)blocks * NPH); CK(hipMemcpy(h.data(), ph, h.size() * 8, hipMemcpyDeviceToHost));
        const double tiles_per_block = (double)(nq / (THREADS * K)) / blocks; double tot = 0;
        printf("  | us/tile:");
        const char* nm[6] = {"load", "hist", "prefix", "scatter", "gather", "store"};
        for (int i = 0; i < 6; ++i) { double s = 0; for (int bl = 0; bl < blocks; ++bl) s += (double)h[(size_t)bl * NPH + i]; s = s / blocks / tiles_per_block * 0.01; tot += s; printf(" %s %.2f", nm[i], s); }
        printf(" sum %.2f", tot);
    }
    printf("\n");
    return ts[ts.size() / 2];
}
// Prefetching variant: the next tile's queries are loaded into registers during the gather rounds (two vectors per
// round, issued after the round's gathers), boustrophedon region order.
template <int THREADS, int K, int NB, int TIMED, int G, int PF, int XSTAG = 0>
__global__ __launch_bounds__(THREADS) void kp(const double* __restrict__ y, int n, double dx, double inv_dx, const double* __restrict__ xq, double* __restrict__ yq, size_t nq, unsigned long long* __restrict__ ph) {
    constexpr int T = THREADS * K;
    constexpr int ROUNDS = K / G, VPR = (K / 2 + ROUNDS - 1) / ROUNDS;   // prefetch vectors per round
    __shared__ double sq[T];
    __shared__ unsigned hist[NB];
    const size_t ntiles = nq / T;
    const double bscale = (double)NB;
    unsigned long long acc[NPH] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (XSTAG) {   // XCDs 4-7 start half a tile period late: their HBM phases fall into the other half's gather phases
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;
        if (xcc >= 4) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < (unsigned long long)XSTAG) __builtin_amdgcn_s_sleep(32); }
    }
    unsigned long long last = TIMED ? wall_clock64() : 0;
    int it = 0;
    double pfacc = 0.0;
    d2 qv[K / 2], qn[K / 2];
    if (blockIdx.x < ntiles) {
        const d2* q2 = (const d2*)(xq + (size_t)blockIdx.x * T);
#pragma unroll
        for (int u = 0; u < K / 2; ++u) qv[u] = __builtin_nontemporal_load(q2 + threadIdx.x + u * THREADS);
    }
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x, ++it) {
        const bool rev = it & 1;
        d2* o2 = (d2*)(yq + t * T);
        const size_t tn = t + gridDim.x;
        const bool has_next = tn < ntiles;
        const d2* qn2 = (const d2*)(xq + (has_next ? tn : t) * T);
        for (int b = threadIdx.x; b < NB; b += THREADS) hist[b] = 0;
        if (TIMED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        STAMP(0)
        unsigned short bin[K], rank[K];
#pragma unroll
        for (int u = 0; u < K; ++u) { const double qu = (u & 1) ? qv[u / 2].y : qv[u / 2].x; int b = (int)(qu * bscale); b = min(max(b, 0), NB - 1); bin[u] = (unsigned short)b; rank[u] = (unsigned short)atomicAdd(&hist[b], 1u); }
        __syncthreads();
        STAMP(1)
        if (threadIdx.x < 64) {
            unsigned run = 0;
            for (int base = 0; base < NB; base += 64) {
                const int b = base + threadIdx.x; unsigned v = (b < NB) ? hist[b] : 0, incl = v;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { unsigned o = __shfl_up(incl, off, 64); if ((int)threadIdx.x >= off) incl += o; }
                if (b < NB) hist[b] = run + incl - v;
                run += __shfl(incl, 63, 64);
            }
        }
        __syncthreads();
        STAMP(2)
        unsigned short sp[K];
#pragma unroll
        for (int u = 0; u < K; ++u) { sp[u] = (unsigned short)(hist[bin[u]] + rank[u]); sq[sp[u]] = (u & 1) ? qv[u / 2].y : qv[u / 2].x; }
        __syncthreads();
        STAMP(3)
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            double qq[G], rr[G]; int ii[G]; ypair yp[G];
#pragma unroll
            for (int w = 0; w < G; ++w) { const int p = threadIdx.x + (r * G + w) * THREADS; qq[w] = sq[rev ? T - 1 - p : p]; }
#pragma unroll
            for (int w = 0; w < G; ++w) { int i = (int)(qq[w] * inv_dx); i = min(max(i, 0), n - 2); ii[w] = i; yp[w] = *(const ypair*)(y + i); }
            if (PF == 1) {
#pragma unroll
                for (int v = 0; v < VPR; ++v) { const int u = r * VPR + v; if (u < K / 2) qn[u] = __builtin_nontemporal_load(qn2 + threadIdx.x + u * THREADS); }
            }
            if (PF >= 2 && r + 1 < ROUNDS && (PF == 2 || (rev ? (r + 1 >= ROUNDS / 2) : (r + 1 >= ROUNDS / 2)))) {
                // table prefetch: the XCD's 32 workgroups touch every 128-B line of the NEXT round's table span once
                const int npr = n / ROUNDS;
                const int start = (rev ? (ROUNDS - 2 - r) : (r + 1)) * npr;
                const int gidx = (int)(((blockIdx.x >> 3) & 31u) * THREADS + threadIdx.x) * 16;
                pfacc += (gidx < npr) ? y[start + gidx] : 0.0;
            }
#pragma unroll
            for (int w = 0; w < G; ++w) rr[w] = blend(ii[w] * dx, yp[w].a, (ii[w] + 1) * dx, yp[w].b, qq[w]);
#pragma unroll
            for (int w = 0; w < G; ++w) { const int p = threadIdx.x + (r * G + w) * THREADS; sq[rev ? T - 1 - p : p] = rr[w]; }
        }
        __syncthreads();
        STAMP(4)
#pragma unroll
        for (int u = 0; u < K / 2; ++u) { d2 v; v.x = sq[sp[2 * u]]; v.y = sq[sp[2 * u + 1]]; __builtin_nontemporal_store(v, o2 + threadIdx.x + u * THREADS); }
        if (PF != 1 && has_next) {
#pragma unroll
            for (int u = 0; u < K / 2; ++u) qn[u] = __builtin_nontemporal_load(qn2 + threadIdx.x + u * THREADS);
        }
#pragma unroll
        for (int u = 0; u < K / 2; ++u) qv[u] = qn[u];
        __syncthreads();
        STAMP(5)
    }
    if (pfacc == 1.2345e300) yq[0] = pfacc;
    if (TIMED && threadIdx.x == 0) for (int i = 0; i < NPH; ++i) ph[blockIdx.x * NPH + i] = acc[i];
}
template <int THREADS, int K, int NB, int TIMED, int G, int PF, int XSTAG = 0>
float runp(const char* name, const double* y, int n, const double* xq, double* yq, size_t nq, int blocks, unsigned long long* ph) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); double dx = 1.0 / (n - 1);
    hipLaunchKernelGGL((kp<THREADS, K, NB, TIMED, G, PF, XSTAG>), dim3(blocks), dim3(THREADS), 0, 0, y, n, dx, 1.0 / dx, xq, yq, nq, ph); CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < 5; ++r) { CK(hipEventRecord(a)); hipLaunchKernelGGL((kp<THREADS, K, NB, TIMED, G, PF, XSTAG>), dim3(blocks), dim3(THREADS), 0, 0, y, n, dx, 1.0 / dx, xq, yq, nq, ph); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    printf("%-34s threads %4d K %2d bins %3d blocks %4d : %.4f ms", name, THREADS, K, NB, blocks, ts[ts.size() / 2]);
    if (TIMED) {
        std::vector<unsigned long long> h((size_t)blocks * NPH); CK(hipMemcpy(h.data(), ph, h.size() * 8, hipMemcpyDeviceToHost));
        const double tiles_per_block = (double)(nq / (THREADS * K)) / blocks; double tot = 0;
        printf("  | us/tile:");
        const char* nm[6] = {"wait", "hist", "prefix", "scatter", "gather", "store"};
        for (int i = 0; i < 6; ++i) { double s = 0; for (int bl = 0; bl < blocks; ++bl) s += (double)h[(size_t)bl * NPH + i]; s = s / blocks / tiles_per_block * 0.01; tot += s; printf(" %s %.2f", nm[i], s); }
        printf(" sum %.2f", tot);
    }
    printf("\n");
    return ts[ts.size() / 2];
}
// Two wave groups of 512 threads share one LDS tile and alternate roles: while one group sorts/gathers/stores its
// tile, the other group's loads of the next tile are in flight (its waves only sit in the barriers).
template <int K, int NB, int G, int TIMED, int GT = 512>
__global__ __launch_bounds__(2 * GT) void k2(const double* __restrict__ y, int n, double dx, double inv_dx, const double* __restrict__ xq, double* __restrict__ yq, size_t nq, unsigned long long* __restrict__ ph) {
    constexpr int T = GT * K, ROUNDS = K / G;
    __shared__ double sq[T];
    __shared__ unsigned hist[NB];
    const size_t ntiles = nq / T;
    const double bscale = (double)NB;
    const int grp = threadIdx.x / GT, tid = threadIdx.x & (GT - 1);
    const int nsteps = (blockIdx.x < ntiles) ? (int)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
    unsigned long long acc[NPH] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last = TIMED ? wall_clock64() : 0;
    d2 qv[K / 2];
    if (grp == 0 && nsteps > 0) {
        const d2* q2 = (const d2*)(xq + (size_t)blockIdx.x * T);
#pragma unroll
        for (int u = 0; u < K / 2; ++u) qv[u] = __builtin_nontemporal_load(q2 + tid + u * GT);
    }
    for (int s = 0; s < nsteps; ++s) {
        const bool active = (s & 1) == grp;
        const bool rev = s & 1;
        const size_t t = blockIdx.x + (size_t)s * gridDim.x;
        if (active) for (int b = tid; b < NB; b += GT) hist[b] = 0;
        __syncthreads();
        STAMP(0)
        unsigned pk[K / 2];   // two 16-bit ranks, later two sorted positions, per register
        if (active) {
#pragma unroll
            for (int u = 0; u < K / 2; ++u) {
                int b0 = (int)(qv[u].x * bscale); b0 = min(max(b0, 0), NB - 1);
                int b1 = (int)(qv[u].y * bscale); b1 = min(max(b1, 0), NB - 1);
                const unsigned r0 = atomicAdd(&hist[b0], 1u), r1 = atomicAdd(&hist[b1], 1u);
                pk[u] = r0 | (r1 << 16);
            }
        }
        __syncthreads();
        STAMP(1)
        if (active && tid < 64) {
            unsigned run = 0;
            for (int base = 0; base < NB; base += 64) {
                const int b = base + tid; unsigned v = (b < NB) ? hist[b] : 0, incl = v;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { unsigned o = __shfl_up(incl, off, 64); if (tid >= off) incl += o; }
                if (b < NB) hist[b] = run + incl - v;
                run += __shfl(incl, 63, 64);
            }
        }
        __syncthreads();
        STAMP(2)
        if (active) {
#pragma unroll
            for (int u = 0; u < K / 2; ++u) {
                int b0 = (int)(qv[u].x * bscale); b0 = min(max(b0, 0), NB - 1);
                int b1 = (int)(qv[u].y * bscale); b1 = min(max(b1, 0), NB - 1);
                const unsigned p0 = hist[b0] + (pk[u] & 0xffffu), p1 = hist[b1] + (pk[u] >> 16);
                sq[p0] = qv[u].x; sq[p1] = qv[u].y;
                pk[u] = p0 | (p1 << 16);
            }
        }
        __syncthreads();
        STAMP(3)
        if (!active && s + 1 < nsteps) {   // the other group's next tile streams in while this tile is gathered
            const d2* q2 = (const d2*)(xq + (t + gridDim.x) * T);
#pragma unroll
            for (int u = 0; u < K / 2; ++u) qv[u] = __builtin_nontemporal_load(q2 + tid + u * GT);
        }
        if (active) {
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                double qq[G], rr[G]; int ii[G]; ypair yp[G];
#pragma unroll
                for (int w = 0; w < G; ++w) { const int p = tid + (r * G + w) * GT; qq[w] = sq[rev ? T - 1 - p : p]; }
#pragma unroll
                for (int w = 0; w < G; ++w) { int i = (int)(qq[w] * inv_dx); i = min(max(i, 0), n - 2); ii[w] = i; yp[w] = *(const ypair*)(y + i); }
#pragma unroll
                for (int w = 0; w < G; ++w) rr[w] = blend(ii[w] * dx, yp[w].a, (ii[w] + 1) * dx, yp[w].b, qq[w]);
#pragma unroll
                for (int w = 0; w < G; ++w) { const int p = tid + (r * G + w) * GT; sq[rev ? T - 1 - p : p] = rr[w]; }
            }
        }
        __syncthreads();
        STAMP(4)
        if (active) {
            d2* o2 = (d2*)(yq + t * T);
#pragma unroll
            for (int u = 0; u < K / 2; ++u) { d2 v; v.x = sq[pk[u] & 0xffffu]; v.y = sq[pk[u] >> 16]; __builtin_nontemporal_store(v, o2 + tid + u * GT); }
        }
        STAMP(5)
    }
    if (TIMED && threadIdx.x == 0) for (int i = 0; i < NPH; ++i) ph[blockIdx.x * NPH + i] = acc[i];
}
template <int K, int NB, int G, int TIMED, int GT = 512>
float run2(const char* name, const double* y, int n, const double* xq, double* yq, size_t nq, int blocks, unsigned long long* ph) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); double dx = 1.0 / (n - 1);
    hipLaunchKernelGGL((k2<K, NB, G, TIMED, GT>), dim3(blocks), dim3(2 * GT), 0, 0, y, n, dx, 1.0 / dx, xq, yq, nq, ph); CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < 5; ++r) { CK(hipEventRecord(a)); hipLaunchKernelGGL((k2<K, NB, G, TIMED, GT>), dim3(blocks), dim3(2 * GT), 0, 0, y, n, dx, 1.0 / dx, xq, yq, nq, ph); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    printf("%-34s K %2d bins %3d G %d blocks %4d : %.4f ms", name, K, NB, G, blocks, ts[ts.size() / 2]);
    if (TIMED) {
        std::vector<unsigned long long> h((size_t)blocks * NPH); CK(hipMemcpy(h.data(), ph, h.size() * 8, hipMemcpyDeviceToHost));
        const double tiles_per_block = (double)(nq / ((size_t)GT * K)) / blocks; double tot = 0;
        printf("  | us/tile:");
        const char* nm[6] = {"wait", "hist", "prefix", "scatter", "gather", "store"};
        for (int i = 0; i < 6; ++i) { double s = 0; for (int bl = 0; bl < blocks; ++bl) s += (double)h[(size_t)bl * NPH + i]; s = s / blocks / tiles_per_block * 0.01; tot += s; printf(" %s %.2f", nm[i], s); }
        printf(" sum %.2f", tot);
    }
    printf("\n");
    return ts[ts.size() / 2];
}
// k3: two wave groups of 512 threads, EACH with its own LDS tile (2 x 64 KiB), locked half a tile period apart by the
// workgroup barriers: while one group gathers (L2-bound), the other stores its previous tile, loads, histograms,
// prefixes and scatters its next one (HBM- and LDS-bound).  Group 0 sweeps the regions upwards, group 1 downwards.
template <int K, int NB, int TIMED>
__global__ __launch_bounds__(1024) void k3(const double* __restrict__ y, int n, double dx, double inv_dx, const double* __restrict__ xq, double* __restrict__ yq, size_t nq, unsigned long long* __restrict__ ph) {
    constexpr int GT = 512, T = GT * K, G = K / 4;      // 4 gather rounds of G queries
    __shared__ double sq2[2][T];
    __shared__ unsigned hist2[2][NB];
    const int grp = threadIdx.x >> 9, tid = threadIdx.x & (GT - 1);
    double* sq = sq2[grp];
    unsigned* hist = hist2[grp];
    const size_t ntiles = nq / T;
    const double bscale = (double)NB;
    // tiles of this workgroup: pairs (2p, 2p+1) for p = blockIdx.x, blockIdx.x + gridDim.x, ...
    const size_t npairs = (ntiles + 1) / 2;
    const int J = (blockIdx.x < npairs) ? (int)((npairs - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;   // tiles per group
    unsigned long long acc[NPH] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last = TIMED ? wall_clock64() : 0;
    d2 qv[K / 2];
    unsigned pk[K / 2];
    const bool rev = grp != 0;
    auto tile_of = [&](int j) -> size_t { return ((size_t)blockIdx.x + (size_t)j * gridDim.x) * 2 + grp; };
    for (int h = 0; h <= 2 * J + 1; ++h) {
        const int hh = h - grp;
        const bool prep = hh >= 0 && (hh & 1) == 0;
        const int j = prep ? hh / 2 : (hh - 1) / 2;
        const bool have = hh >= (prep ? 0 : 1) && j < J && tile_of(j) < ntiles;          // tile j exists for this group
        const bool have_prev = prep && j >= 1 && j - 1 < J && tile_of(j - 1) < ntiles;
        if (prep) {
            // interval 0: store the previous tile's results, issue the loads of the next tile
            if (have_prev) {
                d2* o2 = (d2*)(yq + tile_of(j - 1) * T);
#pragma unroll
                for (int u = 0; u < K / 2; ++u) { d2 v; v.x = sq[pk[u] & 0xffffu]; v.y = sq[pk[u] >> 16]; __builtin_nontemporal_store(v, o2 + tid + u * GT); }
            }
            if (have) {
                const d2* q2 = (const d2*)(xq + tile_of(j) * T);
#pragma unroll
                for (int u = 0; u < K / 2; ++u) qv[u] = __builtin_nontemporal_load(q2 + tid + u * GT);
                for (int b = tid; b < NB; b += GT) hist[b] = 0;
            }
        } else if (have) {
#pragma unroll
            for (int w = 0; w < G; ++w) { const int p = tid + (0 * G + w) * GT; const int pp = rev ? T - 1 - p : p; const double qq = sq[pp]; int i = (int)(qq * inv_dx); i = min(max(i, 0), n - 2); const ypair yp = *(const ypair*)(y + i); sq[pp] = blend(i * dx, yp.a, (i + 1) * dx, yp.b, qq); }
        }
        __syncthreads();
        STAMP(0)
        if (prep) {
            if (have) {
#pragma unroll
                for (int u = 0; u < K / 2; ++u) {
                    int b0 = (int)(qv[u].x * bscale); b0 = min(max(b0, 0), NB - 1);
                    int b1 = (int)(qv[u].y * bscale); b1 = min(max(b1, 0), NB - 1);
                    const unsigned r0 = atomicAdd(&hist[b0], 1u), r1 = atomicAdd(&hist[b1], 1u);
                    pk[u] = r0 | (r1 << 16);
                }
            }
        } else if (have) {
#pragma unroll
            for (int w = 0; w < G; ++w) { const int p = tid + (1 * G + w) * GT; const int pp = rev ? T - 1 - p : p; const double qq = sq[pp]; int i = (int)(qq * inv_dx); i = min(max(i, 0), n - 2); const ypair yp = *(const ypair*)(y + i); sq[pp] = blend(i * dx, yp.a, (i + 1) * dx, yp.b, qq); }
        }
        __syncthreads();
        STAMP(1)
        if (prep) {
            if (have && tid < 64) {
                unsigned run = 0;
                for (int base = 0; base < NB; base += 64) {
                    const int b = base + tid; unsigned v = (b < NB) ? hist[b] : 0, incl = v;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) { unsigned o = __shfl_up(incl, off, 64); if (tid >= off) incl += o; }
                    if (b < NB) hist[b] = run + incl - v;
                    run += __shfl(incl, 63, 64);
                }
            }
        } else if (have) {
#pragma unroll
            for (int w = 0; w < G; ++w) { const int p = tid + (2 * G + w) * GT; const int pp = rev ? T - 1 - p : p; const double qq = sq[pp]; int i = (int)(qq * inv_dx); i = min(max(i, 0), n - 2); const ypair yp = *(const ypair*)(y + i); sq[pp] = blend(i * dx, yp.a, (i + 1) * dx, yp.b, qq); }
        }
        __syncthreads();
        STAMP(2)
        if (prep) {
            if (have) {
#pragma unroll
                for (int u = 0; u < K / 2; ++u) {
                    int b0 = (int)(qv[u].x * bscale); b0 = min(max(b0, 0), NB - 1);
                    int b1 = (int)(qv[u].y * bscale); b1 = min(max(b1, 0), NB - 1);
                    const unsigned p0 = hist[b0] + (pk[u] & 0xffffu), p1 = hist[b1] + (pk[u] >> 16);
                    sq[p0] = qv[u].x; sq[p1] = qv[u].y;
                    pk[u] = p0 | (p1 << 16);
                }
            }
        } else if (have) {
#pragma unroll
            for (int w = 0; w < G; ++w) { const int p = tid + (3 * G + w) * GT; const int pp = rev ? T - 1 - p : p; const double qq = sq[pp]; int i = (int)(qq * inv_dx); i = min(max(i, 0), n - 2); const ypair yp = *(const ypair*)(y + i); sq[pp] = blend(i * dx, yp.a, (i + 1) * dx, yp.b, qq); }
        }
        __syncthreads();
        STAMP(3)
    }
    if (TIMED && threadIdx.x == 0) for (int i = 0; i < NPH; ++i) ph[blockIdx.x * NPH + i] = acc[i];
}
template <int K, int NB, int TIMED>
float run3(const char* name, const double* y, int n, const double* xq, double* yq, size_t nq, int blocks, unsigned long long* ph) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); double dx = 1.0 / (n - 1);
    hipLaunchKernelGGL((k3<K, NB, TIMED>), dim3(blocks), dim3(1024), 0, 0, y, n, dx, 1.0 / dx, xq, yq, nq, ph); CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < 5; ++r) { CK(hipEventRecord(a)); hipLaunchKernelGGL((k3<K, NB, TIMED>), dim3(blocks), dim3(1024), 0, 0, y, n, dx, 1.0 / dx, xq, yq, nq, ph); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    printf("%-34s K %2d bins %3d blocks %4d : %.4f ms", name, K, NB, blocks, ts[ts.size() / 2]);
    if (TIMED) {
        std::vector<unsigned long long> h((size_t)blocks * NPH); CK(hipMemcpy(h.data(), ph, h.size() * 8, hipMemcpyDeviceToHost));
        const double halfsteps = (double)(nq / ((size_t)512 * K)) / blocks; double tot = 0;     // = tiles per workgroup = half-steps (2 groups)
        printf("  | us per half-step interval:");
        const char* nm[4] = {"store+load|G1", "hist|G2", "prefix|G3", "scatter|G4"};
        for (int i = 0; i < 4; ++i) { double s = 0; for (int bl = 0; bl < blocks; ++bl) s += (double)h[(size_t)bl * NPH + i]; s = s / blocks / halfsteps * 0.01; tot += s; printf(" %s %.2f", nm[i], s); }
        printf(" sum %.2f", tot);
    }
    printf("\n");
    return ts[ts.size() / 2];
}
#ifndef EXP_NO_MAIN
int main(int argc, char** argv) {
    const int n = 1000000; const size_t nq = 100000000 / 65536 * 65536;
    std::vector<double> hy(n + 1); for (int i = 0; i <= n; ++i) hy[i] = sin(6.28 * i / n);
    std::vector<double> hq(nq); unsigned long long s = 12345; for (size_t i = 0; i < nq; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; hq[i] = (double)(s >> 11) * 0x1.0p-53; }
    double *y, *xq, *yq; unsigned long long* ph; CK(hipMalloc(&y, (n + 1) * 8)); CK(hipMalloc(&xq, nq * 8)); CK(hipMalloc(&yq, nq * 8)); CK(hipMalloc(&ph, 4096 * NPH * 8)); CK(hipMalloc(&roles, 4096 * 4)); CK(hipMemset(roles, 0, 4096 * 4));
    CK(hipMemcpy(y, hy.data(), (n + 1) * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(xq, hq.data(), nq * 8, hipMemcpyHostToDevice));
    // aligned pair table: y2[2i] = y[i], y2[2i+1] = y[i+1]
    std::vector<double> hy2(2 * (size_t)n); for (int i = 0; i < n; ++i) { hy2[2 * (size_t)i] = hy[i]; hy2[2 * (size_t)i + 1] = hy[i + 1]; }
    double* y2; CK(hipMalloc(&y2, hy2.size() * 8)); CK(hipMemcpy(y2, hy2.data(), hy2.size() * 8, hipMemcpyHostToDevice));
    run<512, 32, 256, 1, 0, 0, 4, 0>("1 wg/CU 512x32 ascending (reference output)", y, n, xq, yq, nq, 256, ph);
    double* yr; CK(hipMalloc(&yr, nq * 8)); CK(hipMemcpy(yr, yq, nq * 8, hipMemcpyDeviceToDevice));
    run<512, 32, 256, 1, 0, 0, 4, 1>("1 wg/CU 512x32 boustrophedon", y, n, xq, yq, nq, 256, ph);
    runp<512, 32, 256, 0, 4, 0>("kp baseline (1 tile of 16K per CU)", y, n, xq, yq, nq, 256, ph);
    runp<512, 32, 256, 1, 4, 0>("kp baseline timed", y, n, xq, yq, nq, 256, ph);
    CK(hipMemset(yq, 0, nq * 8));
    runp<512, 32, 256, 0, 4, 2>("kp + table prefetch, every round", y, n, xq, yq, nq, 256, ph);
    { std::vector<double> a(nq), b(nq); CK(hipMemcpy(a.data(), yq, nq * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), yr, nq * 8, hipMemcpyDeviceToHost)); size_t bad = 0; for (size_t i2 = 0; i2 < nq; ++i2) bad += a[i2] != b[i2]; printf("mismatches of kp+prefetch vs reference: %zu of %zu\n", bad, nq); }
    runp<512, 32, 256, 1, 4, 2>("kp + table prefetch, every round, timed", y, n, xq, yq, nq, 256, ph);
    runp<512, 32, 256, 0, 4, 3>("kp + table prefetch, second half of the sweep", y, n, xq, yq, nq, 256, ph);
    runp<512, 32, 256, 1, 4, 3>("kp + table prefetch, second half, timed", y, n, xq, yq, nq, 256, ph);
    runp<512, 32, 256, 0, 8, 2>("kp G8 + table prefetch", y, n, xq, yq, nq, 256, ph);
    return 0;
}
#endif
