// 1-D interpolation tables: host-side construction (closed-form detection, centred guess, bucket index) and the
// mi_grid1_* entry points.  Layouts and modes: mi_interp1_eval.hpp.
#include <algorithm>
#include <cmath>
#include <new>
#include <vector>

#include "mi_interp1_eval.hpp"

using mi_interp1::kMaxWalk;

namespace {

mi_status upload(mi_ctx* ctx, void** dev, const void* host, size_t bytes)
{
    hipError_t e = hipMalloc(dev, bytes);
    if (e != hipSuccess) return mi::fail(ctx, MI_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    MI_HIP(ctx, hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
    return MI_OK;
}

inline double host_unode(int formula, double x0, double dx, double span, double den, size_t i)
{
    if (formula == 0) return std::fma((double)i, dx, x0);
    if (formula == 1) return x0 + (double)i * dx;
    if (formula == 2) return x0 + span * ((double)i / den);
    const double rden = 1.0 / den, q0 = (double)i * rden;
    return x0 + span * std::fma(std::fma(-den, q0, (double)i), rden, q0);
}

// Does a closed form reproduce EVERY node of xs bit for bit (the last one may be pinned)?  Fills d on success.
bool detect_closed_form(const std::vector<double>& xs, G1Dev* d)
{
    const size_t n = xs.size();
    if (n < 3) return false;
    const double x0 = xs[0], xl = xs[n - 1], den = (double)(n - 1), span = xl - x0;
    const double dx_cands[2] = {span / den, xs[1] - xs[0]};
    const int order[4] = {0, 1, 3, 2};   // cheapest evaluation first
    for (int fi = 0; fi < 4; ++fi) {
        const int formula = order[fi];
        for (int c = 0; c < (formula >= 2 ? 1 : 2); ++c) {
            const double dx = dx_cands[c];
            if (!(dx > 0.0) || !std::isfinite(dx)) continue;
            bool ok = true;
            for (size_t i = 0; i + 1 < n && ok; ++i) ok = host_unode(formula, x0, dx, span, den, i) == xs[i];
            if (!ok) continue;
            const bool last_ok = host_unode(formula, x0, dx, span, den, n - 1) == xl;
            // the closed-form abscissae must be strictly increasing up to the (possibly pinned) last node
            if (!last_ok && !(host_unode(formula, x0, dx, span, den, n - 2) < xl)) continue;
            d->x0 = x0;
            d->dx = dx;
            d->span = span;
            d->den = den;
            d->rden = 1.0 / den;
            d->formula = formula;
            d->pin_last = last_ok ? 0 : 1;
            d->xmin = x0;
            d->xmax = xl;
            d->scale = 1.0 / dx;
            return true;
        }
    }
    return false;
}

// xs strictly increasing, finite, n >= 2
mi_status build_explicit(mi_ctx* ctx, const std::vector<double>& xs, const std::vector<double>& ys, mi_grid1** out)
{
    const size_t n = xs.size();
    if (n > 0x7ffffff0u) return mi::fail(ctx, MI_ERR_INVALID_ARG, "mi_grid1_create: n=%zu exceeds 2^31", n);
    mi_grid1* g = new (std::nothrow) mi_grid1();
    if (!g) return mi::fail(ctx, MI_ERR_NOMEM, "mi_grid1_create: out of host memory");
    g->ctx = ctx;
    g->device = ctx->device;
    g->n = n;
    g->dev_nodes = g->dev_s = nullptr;
    {   // linspace-like explicit grid: keep only Y, recompute X in registers (bit-exact by construction)
        G1Dev cf;
        memset(&cf, 0, sizeof(cf));
        if (detect_closed_form(xs, &cf)) {
            std::vector<double> yp(ys);
            yp.push_back(ys[n - 1]);
            mi_status st = upload(ctx, &g->dev_nodes, yp.data(), (n + 1) * sizeof(double));
            if (st != MI_OK) { delete g; return st; }
            cf.y = (const double*)g->dev_nodes;
            cf.n = (int)n;
            g->d = cf;
            g->mode = 0;
            g->table_bytes = (n + 1) * sizeof(double);
            *out = g;
            return MI_OK;
        }
    }
    std::vector<d2> nodes(n + 1);
    for (size_t i = 0; i < n; ++i) {
        nodes[i].x = xs[i];
        nodes[i].y = ys[i];
    }
    nodes[n] = nodes[n - 1];   // padding node: r = min(l+1, n-1)
    G1Dev& d = g->d;
    memset(&d, 0, sizeof(d));
    d.n = (int)n;
    d.xmin = xs[0];
    d.xmax = xs[n - 1];
    const double span = d.xmax - d.xmin;

    // Mode-1 test.  g(.) is monotone, so for a query q with bracket i
    // (X[i] <= q < X[i+1]):  g(X[i]) <= g(q) <= g(X[i+1]).  With
    // e_i = g(X[i]) - i in [e_lo, e_hi] for every node, the bracket lies in
    // [g(q) - 1 - e_hi, g(q) - e_lo]: at most e_hi - e_lo + 1 walk steps.
    const double scale1 = (double)(n - 1) / span;
    long e_lo = 0, e_hi = 0;
    bool finite_scale = std::isfinite(scale1) && scale1 > 0.0;
    if (finite_scale) {
        for (size_t i = 0; i < n; ++i) {
            const double t = (xs[i] - d.xmin) * scale1;
            long gi = (t >= 2147483000.0) ? 2147483000L : (long)(int)t;
            gi = std::min<long>(std::max<long>(gi, 0), (long)n - 2);
            const long e = gi - (long)i;
            e_lo = std::min(e_lo, e);
            e_hi = std::max(e_hi, e);
        }
    }
    // the clamp to n-2 makes e = -1 at the last node; harmless (walk-up handles it)
    if (finite_scale && (e_hi - e_lo + 1) <= kMaxWalk) {
        g->mode = 1;
        d.scale = scale1;
        // Centred guess: shift the origin so that every node's scaled abscissa sits inside its own unit cell,
        // t_i = (X_i - gorg) * scale in [i + m, i + 1 - m].  Holds whenever the grid deviates from a straight line by
        // less than one cell (jittered / mildly stretched grids).  Accepted only if the device expression gives
        // exactly i at every node; then eval_batch<3> needs no walk (see there).
        // Two candidate slopes: through the end nodes, and the least-squares line (end nodes of a jittered grid are
        // themselves jittered, which tilts the first one by up to a cell over the length of the table).
        double mi_ = 0.0, mx = 0.0;
        for (size_t i = 0; i < n; ++i) { mi_ += (double)i; mx += xs[i] - d.xmin; }
        mi_ /= (double)n;
        mx /= (double)n;
        double sxy = 0.0, sxx = 0.0;
        for (size_t i = 0; i < n; ++i) {
            const double di = (double)i - mi_;
            sxy += di * ((xs[i] - d.xmin) - mx);
            sxx += di * di;
        }
        const double cand[2] = {scale1, (sxy > 0.0) ? sxx / sxy : 0.0};
        for (int c = 0; c < 2 && !d.centred; ++c) {
            const double sc = cand[c];
            if (!(sc > 0.0) || !std::isfinite(sc) || n >= 0x7fffff00u) continue;
            double dlo = INFINITY, dhi = -INFINITY;
            for (size_t i = 0; i < n; ++i) {
                const double di = (xs[i] - d.xmin) * sc - (double)i;
                dlo = std::min(dlo, di);
                dhi = std::max(dhi, di);
            }
            const double m = 0.5 * (1.0 - (dhi - dlo));
            if (!(m > 1e-6)) continue;
            // Where inside its unit cell [i, i+1) a node's scaled abscissa sits decides how often the kernel needs the third
            // node: a query of cell l has G(q) = l+1 -- and then needs node G-1 on top of G and G+1 -- when it lies above
            // l+1, i.e. with probability E[t_{l+1} - (l+1)].  Centred (margin m on both sides) that is 1/2 for a jittered
            // grid; with the nodes as LOW in their cells as the verification allows (margin e below, 2m - e above) it is the
            // mean jitter, 1/4 for BASELINE's grid: fewer dependent gathers for unordered queries (round 4).  Same bracket,
            // same bits.  Tried from the smallest margin up; the centred origin is the last resort.
            const double margins[3] = {std::min(m, 1.0 / 4096.0), std::min(m, 1.0 / 64.0), m};
            for (int tr = 0; tr < 3 && !d.centred; ++tr) {
                const double gorg = d.xmin + (dlo - margins[tr]) / sc;
                bool ok = std::isfinite(gorg);
                for (size_t i = 0; i < n && ok; ++i) {
                    const double t = (xs[i] - gorg) * sc;
                    ok = t >= 0.0 && t < 2147483000.0 && (size_t)(int)t == i;
                }
                if (ok) {
                    d.gorg = gorg;
                    d.scale = sc;
                    d.centred = 1;
                }
            }
        }
    } else {
        g->mode = 2;
        size_t nb = n;
        double bscale = (double)nb / span;
        if (!std::isfinite(bscale) || !(bscale > 0.0)) {   // span underflow/overflow: one bucket
            nb = 1;
            bscale = 0.0;
        }
        d.nb = (int)nb;
        d.scale = bscale;
        // s[b] = largest node whose bucket is < b (0 if none); s[nb] = n-1
        std::vector<uint32_t> s(nb + 1, 0);
        auto bucket = [&](double x) {
            const double t = (x - d.xmin) * bscale;
            long b = (t >= 2147483000.0) ? 2147483000L : (long)(int)t;
            return (size_t)std::min<long>(std::max<long>(b, 0), (long)nb - 1);
        };
        size_t bprev = bucket(xs[0]);   // == 0
        for (size_t i = 1; i < n; ++i) {
            const size_t bi = bucket(xs[i]);
            for (size_t b = bprev + 1; b <= bi; ++b) s[b] = (uint32_t)(i - 1);
            bprev = bi;
        }
        for (size_t b = bprev + 1; b <= nb; ++b) s[b] = (uint32_t)(n - 1);
        mi_status st = upload(ctx, &g->dev_s, s.data(), (nb + 1) * sizeof(uint32_t));
        if (st != MI_OK) { delete g; return st; }
        d.s = (const uint32_t*)g->dev_s;
    }
    mi_status st = upload(ctx, &g->dev_nodes, nodes.data(), (n + 1) * sizeof(d2));
    if (st != MI_OK) {
        if (g->dev_s) (void)hipFree(g->dev_s);
        delete g;
        return st;
    }
    d.nodes = (const d2*)g->dev_nodes;
    g->table_bytes = (n + 1) * sizeof(d2) + (g->mode == 2 ? ((size_t)d.nb + 1) * 4 : 0);
    *out = g;
    return MI_OK;
}

mi_status fetch(mi_ctx* ctx, const double* p, size_t n, bool dev, std::vector<double>& v)
{
    v.resize(n);
    if (dev) {
        MI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        MI_HIP(ctx, hipMemcpy(v.data(), p, n * sizeof(double), hipMemcpyDeviceToHost));
    } else {
        memcpy(v.data(), p, n * sizeof(double));
    }
    return MI_OK;
}

}  // namespace

extern "C" {

mi_status mi_grid1_create(mi_ctx* ctx, const double* x, const double* y, size_t n, unsigned flags, mi_grid1** out)
{
    MI_REQUIRE(ctx, ctx && x && y && out, "mi_grid1_create: NULL argument");
    MI_REQUIRE(ctx, (flags & ~(MI_GRID_SANITISE | MI_GRID_DEVICE_PTRS)) == 0, "mi_grid1_create: unknown flags 0x%x", flags);
    *out = nullptr;
    if (n < 2) return mi::fail(ctx, MI_ERR_GRID, "mi_grid1_create: X must have at least two elements (n=%zu)", n);
    MI_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<double> xs, ys;
    mi_status st = fetch(ctx, x, n, flags & MI_GRID_DEVICE_PTRS, xs);
    if (st != MI_OK) return st;
    st = fetch(ctx, y, n, flags & MI_GRID_DEVICE_PTRS, ys);
    if (st != MI_OK) return st;
    for (size_t i = 0; i < n; ++i)
        if (!std::isfinite(xs[i]))
            return mi::fail(ctx, MI_ERR_GRID, "mi_grid1_create: X[%zu] is not finite", i);
    bool increasing = true;
    for (size_t i = 1; i < n && increasing; ++i) increasing = xs[i - 1] < xs[i];
    if ((flags & MI_GRID_SANITISE) && !increasing) {
        // arma::interp1 front end: unique + ascending sort of X, Y permuted
        // alike (first occurrence of a duplicate abscissa is kept).  A grid that is
        // already strictly increasing (the usual case) skips the 40 ms sort of 1e6 nodes.
        std::vector<size_t> idx(n);
        for (size_t i = 0; i < n; ++i) idx[i] = i;
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return xs[a] < xs[b]; });
        std::vector<double> x2, y2;
        x2.reserve(n);
        y2.reserve(n);
        for (size_t k = 0; k < n; ++k) {
            if (!x2.empty() && xs[idx[k]] == x2.back()) continue;
            x2.push_back(xs[idx[k]]);
            y2.push_back(ys[idx[k]]);
        }
        xs.swap(x2);
        ys.swap(y2);
        if (xs.size() < 2)
            return mi::fail(ctx, MI_ERR_GRID, "mi_grid1_create: X must have at least two unique elements");
    } else if (!increasing) {
        for (size_t i = 1; i < n; ++i)
            if (!(xs[i - 1] < xs[i]))
                return mi::fail(ctx, MI_ERR_GRID,
                                "mi_grid1_create: X not strictly increasing at %zu (pass MI_GRID_SANITISE)", i);
    }
    return build_explicit(ctx, xs, ys, out);
}

mi_status mi_grid1_create_uniform(mi_ctx* ctx, double x0, double dx, const double* y, size_t n, unsigned flags,
                                  mi_grid1** out)
{
    MI_REQUIRE(ctx, ctx && y && out, "mi_grid1_create_uniform: NULL argument");
    MI_REQUIRE(ctx, (flags & ~MI_GRID_DEVICE_PTRS) == 0, "mi_grid1_create_uniform: unknown flags 0x%x", flags);
    *out = nullptr;
    if (n < 2) return mi::fail(ctx, MI_ERR_GRID, "mi_grid1_create_uniform: need at least two nodes (n=%zu)", n);
    if (n > 0x7ffffff0u) return mi::fail(ctx, MI_ERR_INVALID_ARG, "mi_grid1_create_uniform: n=%zu exceeds 2^31", n);
    if (!(dx > 0.0) || !std::isfinite(dx) || !std::isfinite(x0) || !std::isfinite(std::fma((double)(n - 1), dx, x0)))
        return mi::fail(ctx, MI_ERR_GRID, "mi_grid1_create_uniform: need finite x0 and dx > 0");
    if (!(std::fma(1.0, dx, x0) > x0))
        return mi::fail(ctx, MI_ERR_GRID, "mi_grid1_create_uniform: dx too small relative to x0");
    MI_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<double> ys;
    mi_status st = fetch(ctx, y, n, flags & MI_GRID_DEVICE_PTRS, ys);
    if (st != MI_OK) return st;
    ys.push_back(ys[n - 1]);
    mi_grid1* g = new (std::nothrow) mi_grid1();
    if (!g) return mi::fail(ctx, MI_ERR_NOMEM, "mi_grid1_create_uniform: out of host memory");
    g->ctx = ctx;
    g->device = ctx->device;
    g->mode = 0;
    g->n = n;
    g->dev_s = nullptr;
    st = upload(ctx, &g->dev_nodes, ys.data(), (n + 1) * sizeof(double));
    if (st != MI_OK) { delete g; return st; }
    G1Dev& d = g->d;
    memset(&d, 0, sizeof(d));
    d.y = (const double*)g->dev_nodes;
    d.n = (int)n;
    d.x0 = x0;
    d.dx = dx;
    d.xmin = x0;
    d.xmax = std::fma((double)(n - 1), dx, x0);
    d.scale = 1.0 / dx;
    g->table_bytes = (n + 1) * sizeof(double);
    *out = g;
    return MI_OK;
}

mi_status mi_grid1_destroy(mi_grid1* g)
{
    if (!g) return MI_OK;
    (void)hipSetDevice(g->device);
    if (g->dev_nodes) (void)hipFree(g->dev_nodes);
    if (g->dev_s) (void)hipFree(g->dev_s);
    delete g;
    return MI_OK;
}

mi_status mi_grid1_info(const mi_grid1* g, size_t* n_nodes, int* mode, size_t* table_bytes)
{
    MI_REQUIRE(nullptr, g != nullptr, "mi_grid1_info: grid is NULL");
    if (n_nodes) *n_nodes = g->n;
    if (mode) *mode = (g->mode == 1 && g->d.centred) ? 3 : g->mode;
    if (table_bytes) *table_bytes = g->table_bytes;
    return MI_OK;
}

}  // extern "C"
