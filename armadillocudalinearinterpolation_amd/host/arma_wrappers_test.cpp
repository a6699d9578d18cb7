// GPU test of the Armadillo-facing wrappers (include/mi355_arma.hpp): arma::vec / arma::mat in, arma::vec out.
// Writes results as raw doubles so that the Python test can compare them with the oracle bit for bit.
//   arma_wrappers_test OUT_DIR
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "mi355_arma.hpp"

static void dump(const std::string& path, const arma::vec& v)
{
    FILE* fp = std::fopen(path.c_str(), "wb");
    std::fwrite(v.memptr(), sizeof(double), v.n_elem, fp);
    std::fclose(fp);
}

int main(int argc, char** argv)
{
    const std::string out = argc > 1 ? argv[1] : ".";
    // config-1 shape: NG = 1e4 uniform grid, NQ = 1e5 queries (deterministic LCG, re-created in Python)
    const arma::uword ng = 10000, nq = 100000;
    arma::vec X(ng), Y(ng), XI(nq), YI;
    for (arma::uword i = 0; i < ng; ++i) { X(i) = (double)i / (double)(ng - 1); Y(i) = std::sin(6.283185307179586 * X(i)) + 0.5 * X(i); }
    unsigned long long s = 42;
    for (arma::uword i = 0; i < nq; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; XI(i) = (double)(s >> 11) * 0x1.0p-53 * 1.1 - 0.05; }
    mi355::interp1(X, Y, XI, YI);                       // one-shot arma::interp1 shape
    dump(out + "/w_interp1.bin", YI);
    mi355::Interp1Table table(X, Y);                    // resident table, two query batches
    arma::vec YA, YB;
    table(XI, YA);
    table(XI, YB, -1.0);
    dump(out + "/w_table_nan.bin", YA);
    dump(out + "/w_table_extrap.bin", YB);
    {   // the same call with the queries sharded over a device group: three shards rehearsed on GPU 0, and GPUs 0..0
        mi355::DeviceGroup grp(std::vector<int>{0, 0, 0});
        mi355::GroupInterp1Table gtab(grp, X, Y);
        arma::vec YG;
        gtab(XI, YG);
        dump(out + "/w_group_table.bin", YG);
        mi355::DeviceGroup one(1);
        mi355::GroupInterp1Table gone(one, X, Y);
        gone(XI, YG, -1.0);
        dump(out + "/w_group1_table_extrap.bin", YG);
    }
    // bilinear: Z is Y.n_elem x X.n_elem (arma::mat, column-major)
    const arma::uword nx = 40, ny = 25, n2 = 20000;
    arma::vec xg(nx), yg(ny), xq(n2), yq(n2), ZI;
    arma::mat Z(ny, nx);
    for (arma::uword j = 0; j < nx; ++j) xg(j) = 0.1 * j * (1.0 + 0.01 * j);
    for (arma::uword i = 0; i < ny; ++i) yg(i) = -1.0 + 0.2 * i;
    for (arma::uword j = 0; j < nx; ++j)
        for (arma::uword i = 0; i < ny; ++i) Z(i, j) = std::sin(xg(j)) * std::cos(yg(i)) + 0.1 * xg(j) * yg(i);
    for (arma::uword k = 0; k < n2; ++k) {
        s = s * 6364136223846793005ull + 1442695040888963407ull; xq(k) = (double)(s >> 11) * 0x1.0p-53 * (xg(nx - 1) + 0.2) - 0.1;
        s = s * 6364136223846793005ull + 1442695040888963407ull; yq(k) = (double)(s >> 11) * 0x1.0p-53 * 5.2 - 1.1;
    }
    mi355::interp2(xg, yg, Z, xq, yq, ZI);
    dump(out + "/w_interp2.bin", ZI);
    {   // the same over a device group (GPU 0 named twice: two query shards)
        mi355::DeviceGroup grp2(std::vector<int>{0, 0});
        mi355::GroupInterp2Table g2(grp2, xg, yg, Z);
        arma::vec ZG;
        g2(xq, yq, ZG);
        dump(out + "/w_group_interp2.bin", ZG);
    }
    // the reference's own interpolation on host vectors (known answer of SURVEY 8c + a small batch)
    arma::fvec t0(4), t1(4), xr;
    std::vector<uint16_t> i0 = {512, 100, 1023, 0}, i1 = {514, 101, 1023, 1};
    t0(0) = 4.f; t1(0) = 6.f; t0(1) = 4.5f; t1(1) = 5.25f; t0(2) = 1.f; t1(2) = 9.f; t0(3) = 4.99f; t1(3) = 5.01f;
    mi355::restrict_to_horizon(t0, i0, t1, i1, 5.0f, 3.0f, 1024, xr);
    {
        FILE* fp = std::fopen((out + "/w_restrict.bin").c_str(), "wb");
        std::fwrite(xr.memptr(), sizeof(float), xr.n_elem, fp);
        std::fclose(fp);
    }
    if (xr(0) != 0.005859375f) { std::printf("restrict KAT failed: %.9g\n", xr(0)); return 1; }
    // error convention: mismatched sizes throw
    int threw = 0;
    try { arma::vec bad(3); mi355::interp1(X, bad, XI, YI); } catch (const std::exception&) { threw = 1; }
    std::printf("wrappers ok, size-mismatch threw=%d, armadillo=%d\n", threw, MI355_HAVE_ARMADILLO);
    return threw ? 0 : 1;
}
