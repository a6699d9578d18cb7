// Exploration harness (NOT product code), round 2: DOUBLE-TILE sweep.  What the region sweep pays for is the table lines
// that pass through an XCD's L2 per sweep (5.3 MB per 32 tiles of 16 384 queries).  A CU's LDS holds one sorted tile, but
// its registers hold two tiles' worth of queries (1024 lanes x 32): so sort 32 768 queries per workgroup at once, put the
// half that falls into the lower half of the table into LDS, sweep eighths 0..3, take the results back into registers, then
// the same for the upper half -- one pass over the table per 32 768 queries per CU instead of per 16 384.
#define EXP_NO_MAIN
#include "exp_sweep4.hip"
#ifndef EXP_K7
#define EXP_K7 32
#endif
constexpr int T7 = 1024, K7 = EXP_K7, NB7 = 256, CAP7 = K7 * 512 + (K7 >= 32 ? 3072 : 2048);   // 152 KB of LDS for a half (16 384 expected, sigma 90)
template <int GATHER, int G>
__global__ __launch_bounds__(T7) void k7(const double* __restrict__ y, int n, double dx, double inv_dx, const double* __restrict__ xq, double* __restrict__ yq, size_t nq) {
    __shared__ double sq[CAP7];
    __shared__ unsigned hist[NB7];
    __shared__ unsigned cnt[2];
    constexpr int TT = T7 * K7;
    const size_t ntiles = nq / TT;
    const double bscale = (double)NB7;
    int it = 0;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x, ++it) {
        const d2* q2 = (const d2*)(xq + t * TT); d2* o2 = (d2*)(yq + t * TT);
        double q[K7];
#pragma unroll
        for (int u = 0; u < K7 / 2; ++u) { d2 v = __builtin_nontemporal_load(q2 + threadIdx.x + u * T7); q[2 * u] = v.x; q[2 * u + 1] = v.y; }
        for (int b = threadIdx.x; b < NB7; b += T7) hist[b] = 0;
        __syncthreads();
        unsigned sp2[K7 / 2]; unsigned hi = 0;   // sorted positions, two per register; hi: bit u = query u belongs to the upper half
#pragma unroll
        for (int u = 0; u < K7; u += 2) {
            int b0 = (int)(q[u] * bscale); b0 = min(max(b0, 0), NB7 - 1);
            int b1 = (int)(q[u + 1] * bscale); b1 = min(max(b1, 0), NB7 - 1);
            const unsigned r0 = atomicAdd(&hist[b0], 1u), r1 = atomicAdd(&hist[b1], 1u);
            sp2[u / 2] = r0 | (r1 << 16);
            hi |= ((unsigned)(b0 >> 7) << u) | ((unsigned)(b1 >> 7) << (u + 1));
            if ((u & 6) == 6) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            for (int half = 0; half < 2; ++half) {
                unsigned run = 0;
                for (int base = half * 128; base < half * 128 + 128; base += 64) {
                    const int b = base + threadIdx.x; unsigned v = hist[b], incl = v;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) { unsigned o = __shfl_up(incl, off, 64); if ((int)threadIdx.x >= off) incl += o; }
                    hist[b] = run + incl - v;
                    run += __shfl(incl, 63, 64);
                }
                if (threadIdx.x == 0) cnt[half] = min(run, (unsigned)CAP7);
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < K7; u += 2) {
            double qa = q[u], qb = q[u + 1];
            asm volatile("" : "+v"(qa), "+v"(qb));
            int b0 = (int)(qa * bscale); b0 = min(max(b0, 0), NB7 - 1);
            int b1 = (int)(qb * bscale); b1 = min(max(b1, 0), NB7 - 1);
            const unsigned p0 = min(hist[b0] + (sp2[u / 2] & 0xffffu), (unsigned)CAP7 - 1u), p1 = min(hist[b1] + (sp2[u / 2] >> 16), (unsigned)CAP7 - 1u);
            sp2[u / 2] = p0 | (p1 << 16);
            if ((u & 6) == 6) __builtin_amdgcn_sched_barrier(0);
        }
        const bool down = (it & 1) != 0;
        for (int pp = 0; pp < 2; ++pp) {
            const unsigned ph = down ? 1u - pp : pp;
#pragma unroll
            for (int u = 0; u < K7; ++u) { if (((hi >> u) & 1u) == ph) sq[(sp2[u / 2] >> (16 * (u & 1))) & 0xffffu] = q[u]; if ((u & 7) == 7) __builtin_amdgcn_sched_barrier(0); }
            __syncthreads();
            const int np = (int)cnt[ph];
#pragma unroll 1
            for (int p0 = 0; p0 < np; p0 += G * T7) {
                double qq[G], rr[G]; int pos[G];
#pragma unroll
                for (int w = 0; w < G; ++w) { const int p = p0 + w * T7 + threadIdx.x; pos[w] = p < np ? (down ? np - 1 - p : p) : -1; qq[w] = pos[w] >= 0 ? sq[pos[w]] : 0.5; }
#pragma unroll
                for (int w = 0; w < G; ++w) rr[w] = evalq<GATHER>(y, n, dx, inv_dx, qq[w]);
#pragma unroll
                for (int w = 0; w < G; ++w) if (pos[w] >= 0) sq[pos[w]] = rr[w];
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < K7; ++u) { if (((hi >> u) & 1u) == ph) q[u] = sq[(sp2[u / 2] >> (16 * (u & 1))) & 0xffffu]; if ((u & 7) == 7) __builtin_amdgcn_sched_barrier(0); }
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < K7 / 2; ++u) { d2 v; v.x = q[2 * u]; v.y = q[2 * u + 1]; __builtin_nontemporal_store(v, o2 + threadIdx.x + u * T7); }
    }
}
template <int GATHER, int G>
void run7(const char* name, const double* y, int n, const double* xq, double* yq, size_t nq, int blocks) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); double dx = 1.0 / (n - 1);
    hipLaunchKernelGGL((k7<GATHER, G>), dim3(blocks), dim3(T7), 0, 0, y, n, dx, 1.0 / dx, xq, yq, nq); CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < 5; ++r) { CK(hipEventRecord(a)); hipLaunchKernelGGL((k7<GATHER, G>), dim3(blocks), dim3(T7), 0, 0, y, n, dx, 1.0 / dx, xq, yq, nq); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    const double done = (double)(nq / (T7 * K7)) * (T7 * K7);
    printf("%-40s K %d blocks %4d : %.4f ms = %.4f ms per 1e8 queries\n", name, K7, blocks, ts[2], ts[2] * 1e8 / done);
}
int main() {
    const size_t nq = 100000000 / 65536 * 65536;
    std::vector<double> hq(nq); unsigned long long s = 12345; for (size_t i = 0; i < nq; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; hq[i] = (double)(s >> 11) * 0x1.0p-53; }
    double *xq, *yq, *yref; unsigned long long* ph; CK(hipMalloc(&xq, nq * 8)); CK(hipMalloc(&yq, nq * 8)); CK(hipMalloc(&yref, nq * 8)); CK(hipMalloc(&ph, 4096 * NPH * 8)); CK(hipMalloc(&roles, 4096 * 4)); CK(hipMemset(roles, 0, 4096 * 4));
    CK(hipMemcpy(xq, hq.data(), nq * 8, hipMemcpyHostToDevice));
    const int n = 1000000;
    std::vector<double> hy(n + 1); for (int i = 0; i <= n; ++i) hy[i] = sin(6.28 * i / n);
    double* y; CK(hipMalloc(&y, (n + 1) * 8)); CK(hipMemcpy(y, hy.data(), (n + 1) * 8, hipMemcpyHostToDevice));
    run<512, 32, 256, 1, 0, 0, 4, 1>("single tile, 512 lanes (reference)", y, n, xq, yref, nq, 256, ph);
    for (int rep = 0; rep < 2; ++rep) {
        run7<1, 4>("double tile, 4 gathers in flight", y, n, xq, yq, nq, 256);
        run7<1, 2>("double tile, 2 gathers in flight", y, n, xq, yq, nq, 256);
        run7<1, 8>("double tile, 8 gathers in flight", y, n, xq, yq, nq, 256);
    }
    run7<1, 4>("double tile, 4 in flight (for the check)", y, n, xq, yq, nq, 256);
    std::vector<double> a(1 << 20), b(1 << 20); size_t bad = 0;
    for (size_t off : {(size_t)0, nq / 4, nq / 2}) {
        CK(hipMemcpy(a.data(), yq + off, a.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), yref + off, b.size() * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
    }
    printf("mismatches against the single-tile kernel on 3 x 2^20 samples: %zu\n", bad);
    return 0;
}
