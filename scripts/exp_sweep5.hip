// Exploration harness (NOT product code), round 2: is the region sweep's gather phase pinned by a per-CU limit or by
// the XCD L2's request rate?  Round 1 found the phase at ~18.5 us per 16K-query tile with 256, 128 and 64 workgroups
// over the 8 MB table -- but fewer workgroups per sweep also mean MORE table misses per query (each sweep re-reads
// half the table per XCD whatever the number of tiles in it).  Here the same phase-timed loop (exp_sweep4.hip's k)
// runs over a 1 MB table (125 000 nodes, fully L2-resident: no misses at any workgroup count) and over the 8 MB table.
#define EXP_NO_MAIN
#include "exp_sweep4.hip"
int main() {
    const size_t nq = 100000000 / 65536 * 65536;
    std::vector<double> hq(nq); unsigned long long s = 12345; for (size_t i = 0; i < nq; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; hq[i] = (double)(s >> 11) * 0x1.0p-53; }
    double *xq, *yq; unsigned long long* ph; CK(hipMalloc(&xq, nq * 8)); CK(hipMalloc(&yq, nq * 8)); CK(hipMalloc(&ph, 4096 * NPH * 8)); CK(hipMalloc(&roles, 4096 * 4)); CK(hipMemset(roles, 0, 4096 * 4));
    CK(hipMemcpy(xq, hq.data(), nq * 8, hipMemcpyHostToDevice));
    for (int n : {125000, 500000, 1000000}) {
        std::vector<double> hy(n + 1); for (int i = 0; i <= n; ++i) hy[i] = sin(6.28 * i / n);
        double* y; CK(hipMalloc(&y, (n + 1) * 8)); CK(hipMemcpy(y, hy.data(), (n + 1) * 8, hipMemcpyHostToDevice));
        printf("## table %d nodes (%.1f MB)\n", n, n * 8e-6);
        for (int blocks : {256, 128, 64, 32}) {
            run<512, 32, 256, 1, 0, 1, 4, 1>("boustrophedon timed", y, n, xq, yq, nq, blocks, ph);
            run<512, 32, 256, 0, 0, 1, 4, 1>("  no gather, timed", y, n, xq, yq, nq, blocks, ph);
        }
        CK(hipFree(y));
    }
    return 0;
}
