// End-to-end timing of the Armadillo-facing calls (include/mi355_arma.hpp) at BASELINE configs[1] size:
// arma::vec X, Y (1e6 nodes), XI (1e8 random queries) -> arma::vec YI, host memory in, host memory out.
// This is the PCIe-inclusive drop-in call; bench.py's headline is the device-resident kernel.
//   arma_interp_bench [NQ] [NG]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "mi355_arma.hpp"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const arma::uword nq = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 100000000ull;
    const arma::uword ng = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1000000ull;
    arma::vec X(ng), Y(ng), XI(nq), YI;
    for (arma::uword i = 0; i < ng; ++i) { X(i) = (double)i / (double)(ng - 1); Y(i) = std::sin(6.283185307179586 * X(i)) + 0.5 * X(i); }
    unsigned long long s = 0x5EED0003ull;
    for (arma::uword i = 0; i < nq; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; XI(i) = (double)(s >> 11) * 0x1.0p-53; }
    for (int rep = 0; rep < 4; ++rep) {
        const double t = now();
        mi355::interp1(X, Y, XI, YI);                      // builds the table, uploads, interpolates, downloads
        const double dt = now() - t;
        std::printf("mi355::interp1(X, Y, XI, YI)  call %d: %.1f ms  %.3g points/s\n", rep, dt * 1e3, (double)nq / dt);
    }
    mi355::Interp1Table table(X, Y);                       // resident table: what a Newton loop would keep
    for (int rep = 0; rep < 4; ++rep) {
        const double t = now();
        table(XI, YI);
        const double dt = now() - t;
        std::printf("Interp1Table::operator()      call %d: %.1f ms  %.3g points/s\n", rep, dt * 1e3, (double)nq / dt);
    }
    double chk = 0.0;
    for (arma::uword i = 0; i < nq; i += 9973) chk += YI(i);
    std::printf("checksum %.17g\n", chk);
    return 0;
}
