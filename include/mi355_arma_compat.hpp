// mi355_arma_compat.hpp -- pulls in Armadillo when it is installed, otherwise a
// minimal stand-in for the handful of Armadillo facilities the host layer uses.
//
// The reference is written against Armadillo (arma::vec / arma::mat /
// arma::fvec, conv_to, norm, solve: EventDrivenMap.cu:61-66,172,237-239,
// NewtonSolver.cpp:66-101).  Armadillo and LAPACK are NOT installed in the
// build image and cannot be fetched, so the stand-in below exists ONLY to let
// the host layer, its tests and the Driver compile and run here.  It is not a
// replacement for Armadillo and is never used when <armadillo> is present:
// the host layer restricts itself to the common subset (element access,
// n_elem/n_rows/n_cols, memptr, set_size, zeros/fill, col-major mat, norm(v,2),
// solve(A,b), conv_to<>::from) so the same sources build either way.
#pragma once

#if defined(MI355_FORCE_ARMA_SHIM)
#define MI355_HAVE_ARMADILLO 0
#elif defined(__has_include)
#if __has_include(<armadillo>)
#define MI355_HAVE_ARMADILLO 1
#else
#define MI355_HAVE_ARMADILLO 0
#endif
#else
#define MI355_HAVE_ARMADILLO 0
#endif

#if MI355_HAVE_ARMADILLO
#include <armadillo>
#else

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <initializer_list>
#include <iomanip>
#include <ostream>
#include <stdexcept>
#include <vector>

namespace arma {

typedef unsigned long long uword;

template <typename T>
class Col {
  public:
    uword n_elem = 0, n_rows = 0;
    const uword n_cols = 1;

    Col() {}
    explicit Col(uword n) { set_size(n); }
    Col(std::initializer_list<T> l) : d_(l) { sync(); }
    Col(const Col& o) : d_(o.d_) { sync(); }
    Col& operator=(const Col& o) { d_ = o.d_; sync(); return *this; }

    void set_size(uword n) { d_.resize(n); sync(); }
    void resize(uword n) { d_.resize(n, T(0)); sync(); }
    Col& zeros() { std::fill(d_.begin(), d_.end(), T(0)); return *this; }
    Col& zeros(uword n) { d_.assign(n, T(0)); sync(); return *this; }
    Col& fill(T v) { std::fill(d_.begin(), d_.end(), v); return *this; }
    T* memptr() { return d_.data(); }
    const T* memptr() const { return d_.data(); }
    T* begin() { return d_.data(); }
    const T* begin() const { return d_.data(); }
    T* end() { return d_.data() + d_.size(); }
    const T* end() const { return d_.data() + d_.size(); }
    T& operator()(uword i) { return d_.at(i); }
    const T& operator()(uword i) const { return d_.at(i); }
    T& operator[](uword i) { return d_[i]; }
    const T& operator[](uword i) const { return d_[i]; }
    bool is_empty() const { return d_.empty(); }
    // element-wise arithmetic (eagerly evaluated): what NewtonSolver.cpp:101,104,134,194 of the reference uses
    Col operator-() const { Col r(*this); for (auto& x : r.d_) x = -x; return r; }
    Col& operator+=(const Col& o) { same(o); for (uword i = 0; i < n_elem; ++i) d_[i] += o.d_[i]; return *this; }
    Col& operator-=(const Col& o) { same(o); for (uword i = 0; i < n_elem; ++i) d_[i] -= o.d_[i]; return *this; }
    Col& operator*=(T k) { for (auto& x : d_) x *= k; return *this; }
    friend Col operator+(Col a, const Col& b) { a += b; return a; }
    friend Col operator-(Col a, const Col& b) { a -= b; return a; }
    friend Col operator*(Col a, T k) { a *= k; return a; }
    friend Col operator*(T k, Col a) { a *= k; return a; }
    Col head(uword n) const { if (n > n_elem) throw std::out_of_range("arma shim head()"); Col r(n); std::copy(d_.begin(), d_.begin() + n, r.d_.begin()); return r; }

  private:
    void same(const Col& o) const { if (o.n_elem != n_elem) throw std::invalid_argument("arma shim: element-wise operation on vectors of different length"); }
    void sync() { n_elem = n_rows = d_.size(); }
    std::vector<T> d_;
};

typedef Col<double> vec;
typedef Col<float> fvec;

template <typename T>
class Mat {   // column-major, like arma::Mat
  public:
    uword n_rows = 0, n_cols = 0, n_elem = 0;
    Mat() {}
    Mat(uword r, uword c) { set_size(r, c); }
    void set_size(uword r, uword c) { n_rows = r; n_cols = c; n_elem = r * c; d_.resize(n_elem); }
    Mat& zeros() { std::fill(d_.begin(), d_.end(), T(0)); return *this; }
    T* memptr() { return d_.data(); }
    const T* memptr() const { return d_.data(); }
    T* colptr(uword j) { return d_.data() + j * n_rows; }
    const T* colptr(uword j) const { return d_.data() + j * n_rows; }
    T& operator()(uword i, uword j) { return d_.at(i + j * n_rows); }
    const T& operator()(uword i, uword j) const { return d_.at(i + j * n_rows); }
    // J.col(i) = v   (NewtonSolver.cpp:194)
    struct ColRef {
        Mat& m;
        uword j;
        ColRef& operator=(const Col<T>& v)
        {
            if (v.n_elem != m.n_rows) throw std::invalid_argument("arma shim col(): length mismatch");
            std::copy(v.begin(), v.end(), m.colptr(j));
            return *this;
        }
    };
    ColRef col(uword j) { if (j >= n_cols) throw std::out_of_range("arma shim col()"); return ColRef{*this, j}; }
    Col<T> col(uword j) const { if (j >= n_cols) throw std::out_of_range("arma shim col()"); Col<T> r(n_rows); std::copy(colptr(j), colptr(j) + n_rows, r.begin()); return r; }

  private:
    std::vector<T> d_;
};

typedef Mat<double> mat;

template <typename Out>
struct conv_to {
    template <typename In>
    static Out from(const Col<In>& v)
    {
        Out o(v.n_elem);
        for (uword i = 0; i < v.n_elem; ++i) o[i] = static_cast<decltype(+o[0])>(v[i]);
        return o;
    }
};

template <typename T>
inline double norm(const Col<T>& v, int p = 2)
{
    if (p != 2) throw std::invalid_argument("arma shim: only the 2-norm is provided");
    // scaled sum of squares (robust against overflow), as LAPACK dnrm2 does
    double scale = 0.0, ssq = 1.0;
    for (uword i = 0; i < v.n_elem; ++i) {
        const double a = std::fabs((double)v[i]);
        if (a != a) return a;
        if (a > 0.0) {
            if (scale < a) { ssq = 1.0 + ssq * (scale / a) * (scale / a); scale = a; }
            else ssq += (a / scale) * (a / scale);
        }
    }
    return scale * std::sqrt(ssq);
}

// solve(A, b): Gaussian elimination with partial pivoting (what dgesv does for the 3x3 systems of
// NewtonSolver.cpp:101); throws on a singular matrix like arma::solve.
inline vec solve(const mat& A, const vec& b)
{
    const uword n = A.n_rows;
    if (A.n_cols != n || b.n_elem != n) throw std::invalid_argument("arma shim solve(): size mismatch");
    std::vector<double> a(A.memptr(), A.memptr() + n * n);
    vec x(b);
    for (uword k = 0; k < n; ++k) {
        uword piv = k;
        for (uword i = k + 1; i < n; ++i)
            if (std::fabs(a[i + k * n]) > std::fabs(a[piv + k * n])) piv = i;
        if (a[piv + k * n] == 0.0 || a[piv + k * n] != a[piv + k * n]) throw std::runtime_error("solve(): singular matrix");
        if (piv != k) {
            for (uword j = 0; j < n; ++j) std::swap(a[k + j * n], a[piv + j * n]);
            std::swap(x[k], x[piv]);
        }
        for (uword i = k + 1; i < n; ++i) {
            const double m = a[i + k * n] / a[k + k * n];
            a[i + k * n] = 0.0;
            for (uword j = k + 1; j < n; ++j) a[i + j * n] -= m * a[k + j * n];
            x[i] -= m * x[k];
        }
    }
    for (uword kk = n; kk-- > 0;) {
        double s = x[kk];
        for (uword j = kk + 1; j < n; ++j) s -= a[kk + j * n] * x[j];
        x[kk] = s / a[kk + kk * n];
    }
    return x;
}

template <typename T>
inline std::ostream& operator<<(std::ostream& os, const Col<T>& v)
{
    for (uword i = 0; i < v.n_elem; ++i) os << "   " << std::setprecision(4) << std::fixed << v[i] << "\n";
    return os;
}

}  // namespace arma

#endif  // MI355_HAVE_ARMADILLO
