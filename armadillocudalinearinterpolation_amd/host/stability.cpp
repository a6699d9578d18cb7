#include "stability.hpp"

#include <cmath>
#include <stdexcept>

namespace mi355 {

namespace {
typedef std::complex<double> cd;

// Givens rotation G = [c s; -conj(s) c] with G*[a; b] = [r; 0], c real
inline void givens(cd a, cd b, double& c, cd& s)
{
    const double na = std::abs(a), nb = std::abs(b);
    if (nb == 0.0) { c = 1.0; s = 0.0; return; }
    if (na == 0.0) { c = 0.0; s = std::conj(b) / nb; return; }
    const double r = std::hypot(na, nb);
    c = na / r;
    s = (a / na) * std::conj(b) / r;
}
}  // namespace

std::vector<std::complex<double>> eig_general(const double* a, int n)
{
    std::vector<cd> H((size_t)n * n);
    auto at = [&](int i, int j) -> cd& { return H[(size_t)i + (size_t)j * n]; };
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) at(i, j) = a[(size_t)i + (size_t)j * n];
    // unitary similarity to upper Hessenberg form (Givens rotations)
    for (int k = 0; k + 2 < n; ++k)
        for (int i = k + 2; i < n; ++i) {
            if (at(i, k) == cd(0)) continue;
            double c; cd s;
            givens(at(k + 1, k), at(i, k), c, s);
            for (int j = 0; j < n; ++j) {           // rows k+1, i
                const cd x = at(k + 1, j), y = at(i, j);
                at(k + 1, j) = c * x + s * y;
                at(i, j) = -std::conj(s) * x + c * y;
            }
            for (int r = 0; r < n; ++r) {           // columns k+1, i (G^H from the right)
                const cd x = at(r, k + 1), y = at(r, i);
                at(r, k + 1) = c * x + std::conj(s) * y;
                at(r, i) = -s * x + c * y;
            }
        }
    std::vector<cd> ev(n);
    int hi = n - 1, iter = 0;
    const double eps = 2.220446049250313e-16;
    while (hi >= 0) {
        if (hi == 0) { ev[0] = at(0, 0); break; }
        int l = hi;                                   // start of the active unreduced block
        while (l > 0) {
            const double sub = std::abs(at(l, l - 1));
            const double diag = std::abs(at(l - 1, l - 1)) + std::abs(at(l, l));
            if (sub <= eps * (diag > 0.0 ? diag : 1.0)) { at(l, l - 1) = 0.0; break; }
            --l;
        }
        if (l == hi) { ev[hi] = at(hi, hi); --hi; iter = 0; continue; }
        if (++iter > 300) throw std::runtime_error("eig_general: QR iteration did not converge");
        // Wilkinson shift: eigenvalue of the trailing 2x2 closer to the last diagonal entry
        const cd a11 = at(hi - 1, hi - 1), a12 = at(hi - 1, hi), a21 = at(hi, hi - 1), a22 = at(hi, hi);
        const cd tr = a11 + a22, det = a11 * a22 - a12 * a21;
        const cd disc = std::sqrt(tr * tr - 4.0 * det);
        const cd m1 = 0.5 * (tr + disc), m2 = 0.5 * (tr - disc);
        cd mu = (std::abs(m1 - a22) < std::abs(m2 - a22)) ? m1 : m2;
        if (iter % 11 == 10) mu += cd(std::abs(a21), std::abs(a21));   // exceptional shift against cycling
        for (int i = l; i <= hi; ++i) at(i, i) -= mu;
        std::vector<double> cs(hi - l);
        std::vector<cd> sn(hi - l);
        for (int k = l; k < hi; ++k) {               // QR of the active block by rotations on rows k, k+1
            givens(at(k, k), at(k + 1, k), cs[k - l], sn[k - l]);
            const double c = cs[k - l]; const cd s = sn[k - l];
            for (int j = k; j < n; ++j) {
                const cd x = at(k, j), y = at(k + 1, j);
                at(k, j) = c * x + s * y;
                at(k + 1, j) = -std::conj(s) * x + c * y;
            }
        }
        for (int k = l; k < hi; ++k) {               // RQ: rotations' conjugate transposes on columns k, k+1
            const double c = cs[k - l]; const cd s = sn[k - l];
            const int rmax = (k + 2 <= hi) ? k + 2 : hi;
            for (int r = 0; r <= rmax; ++r) {
                const cd x = at(r, k), y = at(r, k + 1);
                at(r, k) = c * x + std::conj(s) * y;
                at(r, k + 1) = -s * x + c * y;
            }
        }
        for (int i = l; i <= hi; ++i) at(i, i) += mu;
    }
    return ev;
}

}  // namespace mi355

Stability::Stability(ProblemType type, AbstractNonlinearProblem* pProblem)
    : problem_(pProblem), jacobian_(nullptr), type_(type)
{
}

Stability::Stability(ProblemType type, AbstractNonlinearProblem* pProblem, AbstractNonlinearProblemJacobian* pProblemJacobian)
    : problem_(pProblem), jacobian_(pProblemJacobian), type_(type)
{
}

int Stability::Count(const std::vector<std::complex<double>>& ev) const
{
    int n = 0;
    for (const auto& l : ev) n += (type_ == ProblemType::flow) ? (l.real() > 0.0) : (std::abs(l) > 1.0);
    return n;
}

// Stability.cpp:75-111 (same perturbation/restore pattern as NewtonSolver's)
void Stability::ForwardDifferenceJacobian(const arma::vec& u, arma::mat& J)
{
    const arma::uword n = u.n_rows;
    arma::vec du(u), f(n), df(n);
    const double inv_eps = std::pow(eps_, -1);
    problem_->ComputeF(u, f);
    for (arma::uword i = 0; i < n; ++i) {
        if (i > 0) du(i - 1) = u(i - 1);
        du(i) += eps_;
        problem_->ComputeF(du, df);
        for (arma::uword r = 0; r < n; ++r) J(r, i) = (df(r) - f(r)) * inv_eps;
    }
}

std::vector<std::complex<double>> Stability::ComputeEigenvalues(const arma::vec& u)
{
    const arma::uword n = u.n_rows;
    arma::mat J(n, n);
    J.zeros();
    if (jacobian_) jacobian_->ComputeDFDU(u, J);
    else ForwardDifferenceJacobian(u, J);
    if (type_ == ProblemType::equationFree)
        for (arma::uword i = 0; i < n; ++i) J(i, i) += 1.0;          // Stability.cpp:67-70
    return mi355::eig_general(J.memptr(), (int)n);
}

int Stability::ComputeNumUnstableEigenvalues(const arma::vec& u) { return Count(ComputeEigenvalues(u)); }

int Stability::ComputeNumUnstableEigenvalues(const arma::mat& jacobian)
{
    return Count(mi355::eig_general(jacobian.memptr(), (int)jacobian.n_rows));   // no +I here, Stability.cpp:37-50
}
