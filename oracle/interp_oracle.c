/*
 * oracle/interp_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the linear-interpolation semantics the MI355X path
 * must reproduce.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (libmi355interp.so)
 * never links, includes or calls anything in oracle/.
 *
 * PARITY STATUS
 *   - orc_restrict_f32 / orc_masked_mean_f32 restate device code that IS in the
 *     reference (EventDrivenMap.cu:769-785, :787-824).  The reference holds no
 *     golden vectors or tests and cannot be compiled here
 *     (needs <armadillo>, curand.h, nvcc) -> pinned only by analytic
 *     known-answer tests that are exact in fp32 (tests/test_oracle_cpu.py).
 *   - orc_interp1_* restate arma::interp1(X,Y,XI,YI,"linear",extrap) from the
 *     third-party dependency Armadillo (unversioned: Makefile:5 `-larmadillo`;
 *     header set visible in Driver.o.dep dates it to ~5.x/6.x).  Armadillo is
 *     NOT vendored under /root/reference and is not installed in this image;
 *     the reference never calls interp1 itself.  => "parity unpinned" at that
 *     boundary.  The restatement follows the published algorithm of
 *     armadillo_bits/fn_interp1.hpp (interp1_helper_linear / interp1_helper)
 *     and is cross-checked against numpy.interp and 50-digit mpmath
 *     (scripts/make_golden.py), which are independent implementations, not the
 *     reference.
 *   - orc_interp2_bilinear has no counterpart anywhere (BASELINE.json config 3
 *     asks for it); semantics are defined in DESIGN.md and restated here.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 * -ffp-contract=off matters: the blend must round every product and sum
 * separately, exactly as the HIP kernels do (they are compiled with the same
 * contraction setting), so GPU and oracle outputs are bit-identical.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_OK 0
#define ORC_ERR_ARG 1
#define ORC_ERR_GRID 2
#define ORC_ERR_NOMEM 3

/* ------------------------------------------------------------------------
 * The two-point blend of Armadillo's interp1_helper_linear:
 *     weight = (a_err > 0) ? a_err / (a_err + b_err) : 0
 *     YI     = (1 - weight) * Y[a] + weight * Y[b]
 * with a = left bracket node, b = right bracket node (a == b only at the very
 * last grid node), a_err = |X[a]-XI|, b_err = |X[b]-XI|.
 * ---------------------------------------------------------------------- */
static inline double orc_blend(double xa, double ya, double xb, double yb, double q)
{
    const double a_err = q - xa;            /* == |xa - q| exactly, q >= xa   */
    const double b_err = xb - q;            /* == |xb - q| exactly, q <= xb   */
    const double w = (a_err > 0.0) ? a_err / (a_err + b_err) : 0.0;
    return (1.0 - w) * ya + w * yb;
}

/*
 * Literal restatement of interp1_helper_linear (forward "nearest node" scan
 * that resumes from the previous optimum; XG sorted ascending and unique, XI
 * sorted ascending).  O(NG + NI).  Kept literal on purpose -- including the
 * nearest-node search and the a/b swap -- so that the bracket formulation used
 * everywhere else can be tested against it.
 */
void orc_interp1_scan_sorted(const double* xg, const double* yg, size_t ng,
                             const double* xi, size_t ni, double extrap, double* yi)
{
    const double xg_min = xg[0];
    const double xg_max = xg[ng - 1];
    size_t a_best_j = 0, b_best_j = 0;
    for (size_t i = 0; i < ni; ++i) {
        const double v = xi[i];
        if (v != v) { yi[i] = NAN; continue; }          /* deviation: see header of orc_interp1_arma */
        if (v < xg_min || v > xg_max) { yi[i] = extrap; continue; }
        double a_best_err = INFINITY, b_best_err = INFINITY;
        for (size_t j = a_best_j; j < ng; ++j) {
            const double tmp = xg[j] - v;
            const double err = (tmp >= 0.0) ? tmp : -tmp;
            if (err >= a_best_err) break;
            a_best_err = err;
            a_best_j = j;
        }
        if (xg[a_best_j] - v <= 0.0)
            b_best_j = (a_best_j + 1 < ng) ? a_best_j + 1 : a_best_j;   /* nearest is left of XI  */
        else
            b_best_j = (a_best_j >= 1) ? a_best_j - 1 : a_best_j;       /* nearest is right of XI */
        b_best_err = fabs(xg[b_best_j] - v);
        size_t a = a_best_j, b = b_best_j;
        double ae = a_best_err, be = b_best_err;
        if (a > b) { size_t t = a; a = b; b = t; double e = ae; ae = be; be = e; }
        const double w = (ae > 0.0) ? ae / (ae + be) : 0.0;
        yi[i] = (1.0 - w) * yg[a] + w * yg[b];
    }
}

/* largest l with xg[l] <= q, assuming xg[0] <= q <= xg[ng-1] */
static inline size_t orc_bracket(const double* xg, size_t ng, double q)
{
    size_t lo = 0, hi = ng;                 /* invariant: xg[lo] <= q, (hi==ng or xg[hi] > q) */
    while (hi - lo > 1) {
        const size_t mid = lo + ((hi - lo) >> 1);
        if (xg[mid] <= q) lo = mid; else hi = mid;
    }
    return lo;
}

/*
 * Bracket formulation of the same semantics: l = largest index with
 * X[l] <= XI, r = min(l+1, NG-1).  Equal to the scan whenever fl(|X[j]-XI|) is
 * strictly decreasing along the scan (true for every grid whose spacing
 * exceeds the rounding granularity of the differences -- all BASELINE grids;
 * tests compare both on the fixtures).  Order-independent, so queries need no
 * sort and the loop threads trivially: this is the CPU baseline.
 */
void orc_interp1_bracket(const double* xg, const double* yg, size_t ng,
                         const double* xi, size_t ni, double extrap, double* yi,
                         int nthreads)
{
    const double xg_min = xg[0], xg_max = xg[ng - 1];
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (long long i = 0; i < (long long)ni; ++i) {
        const double v = xi[i];
        if (v != v) { yi[i] = NAN; continue; }
        if (v < xg_min || v > xg_max) { yi[i] = extrap; continue; }
        const size_t l = orc_bracket(xg, ng, v);
        const size_t r = (l + 1 < ng) ? l + 1 : l;
        yi[i] = orc_blend(xg[l], yg[l], xg[r], yg[r], v);
    }
}

/* ---- helpers for the full arma::interp1 front end ---------------------- */
typedef struct { double v; size_t i; } orc_pair;

static int orc_pair_cmp(const void* pa, const void* pb)
{
    const orc_pair* a = (const orc_pair*)pa;
    const orc_pair* b = (const orc_pair*)pb;
    if (a->v < b->v) return -1;
    if (a->v > b->v) return 1;
    return (a->i < b->i) ? -1 : (a->i > b->i);     /* stable: first occurrence first */
}

/*
 * arma::interp1(X, Y, XI, YI, "linear", extrap_val) front end
 * (interp1_helper): X is de-duplicated and sorted ascending (Y permuted
 * alike), XI is sorted if it is not already, the scan runs on the sorted
 * copies and the result is un-permuted.
 *
 * Documented deviations from upstream behaviour (none reachable from the
 * BASELINE configs):
 *   - duplicate X: upstream keeps whichever duplicate std::sort leaves first
 *     (unspecified); here the first occurrence in input order is kept.
 *   - NaN in XI: upstream sort_index() raises; here that query yields NaN.
 *   - NaN in X or fewer than 2 unique X: upstream raises; here ORC_ERR_GRID.
 */
int orc_interp1_arma(const double* x, const double* y, size_t n,
                     const double* xi, size_t ni, double extrap, double* yi)
{
    if (!x || !y || n < 2 || (ni && (!xi || !yi))) return ORC_ERR_ARG;
    orc_pair* gp = (orc_pair*)malloc(n * sizeof(orc_pair));
    double* xs = (double*)malloc(n * sizeof(double));
    double* ys = (double*)malloc(n * sizeof(double));
    if (!gp || !xs || !ys) { free(gp); free(xs); free(ys); return ORC_ERR_NOMEM; }
    for (size_t i = 0; i < n; ++i) {
        if (x[i] != x[i]) { free(gp); free(xs); free(ys); return ORC_ERR_GRID; }
        gp[i].v = x[i]; gp[i].i = i;
    }
    qsort(gp, n, sizeof(orc_pair), orc_pair_cmp);
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        if (m && gp[i].v == xs[m - 1]) continue;
        xs[m] = gp[i].v; ys[m] = y[gp[i].i]; ++m;
    }
    free(gp);
    if (m < 2) { free(xs); free(ys); return ORC_ERR_GRID; }

    int sorted = 1;
    for (size_t i = 1; i < ni; ++i) if (!(xi[i - 1] <= xi[i])) { sorted = 0; break; }
    if (sorted) {
        orc_interp1_scan_sorted(xs, ys, m, xi, ni, extrap, yi);
    } else {
        orc_pair* qp = (orc_pair*)malloc(ni * sizeof(orc_pair));
        double* qs = (double*)malloc(ni * sizeof(double));
        double* rs = (double*)malloc(ni * sizeof(double));
        if (!qp || !qs || !rs) { free(qp); free(qs); free(rs); free(xs); free(ys); return ORC_ERR_NOMEM; }
        size_t nn = 0;                                  /* NaN queries go to the tail */
        for (size_t i = 0; i < ni; ++i) if (xi[i] == xi[i]) { qp[nn].v = xi[i]; qp[nn].i = i; ++nn; }
        qsort(qp, nn, sizeof(orc_pair), orc_pair_cmp);
        for (size_t i = 0; i < nn; ++i) qs[i] = qp[i].v;
        orc_interp1_scan_sorted(xs, ys, m, qs, nn, extrap, rs);
        for (size_t i = 0; i < ni; ++i) yi[i] = NAN;
        for (size_t i = 0; i < nn; ++i) yi[qp[i].i] = rs[i];
        free(qp); free(qs); free(rs);
    }
    free(xs); free(ys);
    return ORC_OK;
}

/*
 * Implicit uniform grid X_i := fma(i, dx, x0), i = 0..ng-1 (single rounding;
 * this DEFINES the abscissae of the uniform entry point, see
 * include/mi355_interp.h mi_grid1_create_uniform).  Same bracket rule and the
 * same blend as above.
 */
static inline double orc_unode(double x0, double dx, long long i) { return fma((double)i, dx, x0); }

void orc_interp1_uniform(double x0, double dx, const double* yg, size_t ng,
                         const double* xi, size_t ni, double extrap, double* yi,
                         int nthreads)
{
    const long long last = (long long)ng - 1;
    const double xmin = x0, xmax = orc_unode(x0, dx, last);
    const double inv_dx = 1.0 / dx;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (long long i = 0; i < (long long)ni; ++i) {
        const double v = xi[i];
        if (v != v) { yi[i] = NAN; continue; }
        if (v < xmin || v > xmax) { yi[i] = extrap; continue; }
        long long l = (long long)((v - x0) * inv_dx);
        if (l < 0) l = 0;
        if (l > last) l = last;
        while (l > 0 && orc_unode(x0, dx, l) > v) --l;
        while (l < last && orc_unode(x0, dx, l + 1) <= v) ++l;
        const long long r = (l < last) ? l + 1 : l;
        yi[i] = orc_blend(orc_unode(x0, dx, l), yg[l], orc_unode(x0, dx, r), yg[r], v);
    }
}

/*
 * Scattered bilinear interpolation on a rectilinear grid (BASELINE.json
 * config 3; no counterpart in the reference or in Armadillo's gridded
 * interp2).  z is column-major ny x nx exactly like arma::mat(ny, nx):
 * z[i + j*ny] = Z(y_i, x_j).  For a query (xq, yq):
 *   - outside [xg0,xg_last] x [yg0,yg_last] -> extrap; NaN coordinate -> NaN
 *   - (lx, wx), (ly, wy) from the 1-D bracket rule / weight of interp1
 *   - blend along y (contiguous in memory) inside the two bracketing columns,
 *     then along x:
 *        c0 = (1-wy)*Z(ly,lx)   + wy*Z(ry,lx)
 *        c1 = (1-wy)*Z(ly,rx)   + wy*Z(ry,rx)
 *        zq = (1-wx)*c0 + wx*c1
 */
void orc_interp2_bilinear(const double* xg, size_t nx, const double* yg, size_t ny,
                          const double* z, const double* xq, const double* yq, size_t nq,
                          double extrap, double* zq, int nthreads)
{
    const double x_min = xg[0], x_max = xg[nx - 1], y_min = yg[0], y_max = yg[ny - 1];
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (long long i = 0; i < (long long)nq; ++i) {
        const double qx = xq[i], qy = yq[i];
        if (qx != qx || qy != qy) { zq[i] = NAN; continue; }
        if (qx < x_min || qx > x_max || qy < y_min || qy > y_max) { zq[i] = extrap; continue; }
        const size_t lx = orc_bracket(xg, nx, qx), rx = (lx + 1 < nx) ? lx + 1 : lx;
        const size_t ly = orc_bracket(yg, ny, qy), ry = (ly + 1 < ny) ? ly + 1 : ly;
        const double ax = qx - xg[lx], bx = xg[rx] - qx;
        const double ay = qy - yg[ly], by = yg[ry] - qy;
        const double wx = (ax > 0.0) ? ax / (ax + bx) : 0.0;
        const double wy = (ay > 0.0) ? ay / (ay + by) : 0.0;
        const double c0 = (1.0 - wy) * z[ly + lx * ny] + wy * z[ry + lx * ny];
        const double c1 = (1.0 - wy) * z[ly + rx * ny] + wy * z[ry + rx * ny];
        zq[i] = (1.0 - wx) * c0 + wx * c1;
    }
}

/* implicit uniform axes: x_j = fma(j, dx, x0), y_i = fma(i, dy, y0) */
void orc_interp2_bilinear_uniform(double x0, double dx, size_t nx, double y0, double dy, size_t ny,
                                  const double* z, const double* xq, const double* yq, size_t nq,
                                  double extrap, double* zq, int nthreads)
{
    const long long lastx = (long long)nx - 1, lasty = (long long)ny - 1;
    const double x_max = orc_unode(x0, dx, lastx), y_max = orc_unode(y0, dy, lasty);
    const double inv_dx = 1.0 / dx, inv_dy = 1.0 / dy;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (long long i = 0; i < (long long)nq; ++i) {
        const double qx = xq[i], qy = yq[i];
        if (qx != qx || qy != qy) { zq[i] = NAN; continue; }
        if (qx < x0 || qx > x_max || qy < y0 || qy > y_max) { zq[i] = extrap; continue; }
        long long lx = (long long)((qx - x0) * inv_dx), ly = (long long)((qy - y0) * inv_dy);
        if (lx < 0) lx = 0;
        if (lx > lastx) lx = lastx;
        if (ly < 0) ly = 0;
        if (ly > lasty) ly = lasty;
        while (lx > 0 && orc_unode(x0, dx, lx) > qx) --lx;
        while (lx < lastx && orc_unode(x0, dx, lx + 1) <= qx) ++lx;
        while (ly > 0 && orc_unode(y0, dy, ly) > qy) --ly;
        while (ly < lasty && orc_unode(y0, dy, ly + 1) <= qy) ++ly;
        const long long rx = (lx < lastx) ? lx + 1 : lx, ry = (ly < lasty) ? ly + 1 : ly;
        const double ax = qx - orc_unode(x0, dx, lx), bx = orc_unode(x0, dx, rx) - qx;
        const double ay = qy - orc_unode(y0, dy, ly), by = orc_unode(y0, dy, ry) - qy;
        const double wx = (ax > 0.0) ? ax / (ax + bx) : 0.0;
        const double wy = (ay > 0.0) ? ay / (ay + by) : 0.0;
        const double c0 = (1.0 - wy) * z[ly + lx * (long long)ny] + wy * z[ry + lx * (long long)ny];
        const double c1 = (1.0 - wy) * z[ly + rx * (long long)ny] + wy * z[ry + rx * (long long)ny];
        zq[i] = (1.0 - wx) * c0 + wx * c1;
    }
}

/*
 * RestrictKernel, EventDrivenMap.cu:769-785 (launch :205-206): in fp32
 *     x_k = -L + 2.0f*L/N * ind_k            (:781-782, N == blockDim.x)
 *     out = x0 + (T - t0)*(x1 - x0)/(t1 - t0) (:783)
 * Operation order: h = (2.0f*L)/(float)N; x_k = fmaf(h, (float)ind_k, -L)
 * (the reference is built with nvcc defaults, Makefile:3, i.e. -fmad=true,
 * which contracts `-L + h*ind` into one FFMA; for L = 3 and N a power of two
 * -- the reference's N = 1024 and 512 -- contracted and uncontracted results
 * are identical because every intermediate is exact); then the product
 * (T-t0)*(x1-x0), an IEEE division by (t1-t0) and a final add, each rounded.
 * No guard for t1 == t0 (inf/NaN), exactly like the reference.
 */
void orc_restrict_f32(const float* t0, const uint16_t* i0, const float* t1, const uint16_t* i1,
                      float T, float L, uint32_t ngrid, float* out, size_t n)
{
    const float h = (2.0f * L) / (float)ngrid;
    for (size_t k = 0; k < n; ++k) {
        const float x0 = fmaf(h, (float)i0[k], -L);
        const float x1 = fmaf(h, (float)i1[k], -L);
        const float num = (T - t0[k]) * (x1 - x0);
        const float q = num / (t1[k] - t0[k]);
        out[k] = x0 + q;
    }
}

/*
 * CountRealisationsKernel + realisationReductionKernelBlocks,
 * EventDrivenMap.cu:787-824: V[m] = sum_{r: accept[r]==1} x[m*R + r] / count,
 * count = sum_r accept[r].
 *
 * Documented decisions:
 *   - the reference overwrites accept[0] with the count before the mean reads
 *     the flags (:800-802 vs :817), so realisation 0 is dropped from the sum
 *     -- while still counted in the divisor (:822) -- unless count == 1, when
 *     it is summed whatever its own flag was.  `quirk != 0` reproduces exactly
 *     that (it is what mi_edm_params.mean_quirk = 1, the default of the
 *     EventDrivenMap pipeline, selects: results identical to the reference's
 *     come first); `quirk == 0` leaves the flags untouched and sums every
 *     accepted realisation (the true mean).
 *   - the reference accumulates in fp32 in a launch-shape-dependent order
 *     (strided partials + shuffle tree); the sum here is accumulated in fp64
 *     in index order, divided by the count in fp64 and rounded to fp32 ONCE
 *     (the reference divides two fp32 values, :822).  Rounding once makes the
 *     mean of identical realisations a function of the common value and R
 *     only, so a residual sharded over GPUs equals the unsharded one when
 *     sigma = 0.
 *     The HIP path produces the same fp64 sum bit-for-bit only when its
 *     partial sums are exact; tests therefore compare the mean with a
 *     1-ulp(fp32) tolerance and the count exactly.
 */
void orc_masked_mean_f32(const float* x, const uint32_t* accept, size_t nreal, size_t nspikes,
                         int quirk, float* mean, uint32_t* count_out)
{
    uint32_t count = 0;
    for (size_t r = 0; r < nreal; ++r) count += accept[r];
    for (size_t m = 0; m < nspikes; ++m) {
        double acc = 0.0;
        for (size_t r = 0; r < nreal; ++r) {
            uint32_t flag = accept[r];
            if (quirk && r == 0) flag = count;          /* accept[0] holds the count by then */
            if (flag == 1u) acc += (double)x[m * nreal + r];
        }
        mean[m] = (float)(acc / (double)count);
    }
    if (count_out) *count_out = count;
}

/* SplitMix64 -> U[0,1): (z >> 11) * 2^-53.  BASELINE.md section 2 seeds. */
void orc_splitmix_uniform(uint64_t seed, double* out, size_t n)
{
    uint64_t s = seed;
    for (size_t i = 0; i < n; ++i) {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        out[i] = (double)(z >> 11) * 0x1.0p-53;
    }
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
