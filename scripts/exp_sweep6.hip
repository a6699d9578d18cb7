// Exploration harness (NOT product code), round 2: what would the region sweep's gather phase cost over the 8 MB table if
// every gather hit L2?  Same phase-timed loop as exp_sweep4.hip's k, same 1e6-node index stream (so the same density of
// queries per table line and per region), but the gather ADDRESS is folded into the first 2 MB / 1 MB of the table
// (index & mask): the access pattern inside a region is unchanged, the footprint fits the 4 MiB L2 of an XCD beside the
// query and result streams.  (Results are wrong by construction; only the phase times are of interest.)
#define EXP_NO_MAIN
#define EXP_FOLD
#include "exp_sweep4.hip"
int main() {
    const size_t nq = 100000000 / 65536 * 65536;
    std::vector<double> hq(nq); unsigned long long s = 12345; for (size_t i = 0; i < nq; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; hq[i] = (double)(s >> 11) * 0x1.0p-53; }
    double *xq, *yq; unsigned long long* ph; CK(hipMalloc(&xq, nq * 8)); CK(hipMalloc(&yq, nq * 8)); CK(hipMalloc(&ph, 4096 * NPH * 8)); CK(hipMalloc(&roles, 4096 * 4)); CK(hipMemset(roles, 0, 4096 * 4));
    CK(hipMemcpy(xq, hq.data(), nq * 8, hipMemcpyHostToDevice));
    const int n = 1000000;
    std::vector<double> hy(n + 1); for (int i = 0; i <= n; ++i) hy[i] = sin(6.28 * i / n);
    double* y; CK(hipMalloc(&y, (n + 1) * 8)); CK(hipMemcpy(y, hy.data(), (n + 1) * 8, hipMemcpyHostToDevice));
    for (int fold : {0x7fffffff, (1 << 19) - 1, (1 << 18) - 1, (1 << 17) - 1, (1 << 15) - 1}) {
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_fold), &fold, sizeof(fold)));
        printf("## 1e6-node index stream, gather footprint %s\n", fold == 0x7fffffff ? "8 MB (unfolded)" : fold == (1 << 19) - 1 ? "4 MB" : fold == (1 << 18) - 1 ? "2 MB" : fold == (1 << 17) - 1 ? "1 MB" : "256 KB");
        for (int blocks : {256, 128}) run<512, 32, 256, 1, 0, 1, 4, 1>("boustrophedon timed", y, n, xq, yq, nq, blocks, ph);
    }
    return 0;
}
