// Exploration harness (NOT product code), round 2: do the query/result streams pollute the XCD L2 less when their buffers are
// allocated uncached / fine-grained (hipExtMallocWithFlags)?  Same phase-timed sweep loop as exp_sweep4.hip's k, table in
// ordinary memory; only the allocation of xq / yq changes.
#define EXP_NO_MAIN
#include "exp_sweep4.hip"
int main() {
    const size_t nq = 100000000 / 65536 * 65536;
    std::vector<double> hq(nq); unsigned long long s = 12345; for (size_t i = 0; i < nq; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; hq[i] = (double)(s >> 11) * 0x1.0p-53; }
    unsigned long long* ph; CK(hipMalloc(&ph, 4096 * NPH * 8)); CK(hipMalloc(&roles, 4096 * 4)); CK(hipMemset(roles, 0, 4096 * 4));
    const int n = 1000000;
    std::vector<double> hy(n + 1); for (int i = 0; i <= n; ++i) hy[i] = sin(6.28 * i / n);
    double* y; CK(hipMalloc(&y, (n + 1) * 8)); CK(hipMemcpy(y, hy.data(), (n + 1) * 8, hipMemcpyHostToDevice));
    const unsigned flags[3] = {hipDeviceMallocDefault, hipDeviceMallocUncached, hipDeviceMallocFinegrained};
    const char* names[3] = {"streams in ordinary memory", "streams in uncached memory", "streams in fine-grained memory"};
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 3; ++m) {
            double *xq, *yq;
            if (hipExtMallocWithFlags((void**)&xq, nq * 8, flags[m]) != hipSuccess || hipExtMallocWithFlags((void**)&yq, nq * 8, flags[m]) != hipSuccess) { printf("%s: allocation failed\n", names[m]); continue; }
            CK(hipMemcpy(xq, hq.data(), nq * 8, hipMemcpyHostToDevice));
            printf("## %s\n", names[m]);
            run<512, 32, 256, 1, 0, 1, 4, 1>("boustrophedon timed", y, n, xq, yq, nq, 256, ph);
            run<512, 32, 256, 1, 0, 0, 4, 1>("boustrophedon", y, n, xq, yq, nq, 256, ph);
            CK(hipFree(xq)); CK(hipFree(yq));
        }
    return 0;
}
