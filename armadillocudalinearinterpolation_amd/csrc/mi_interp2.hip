// Scattered bilinear interpolation over an HBM-resident table (input: column-major
// arma::mat layout) on MI355X.  Semantics: oracle/interp_oracle.c
// orc_interp2_bilinear[_uniform]; blend along y inside the two bracketing columns,
// then along x.  16 B in + 8 B out per query.
// Resident layouts (the table is far larger than L2, so every gather location is an
// L2 miss; both put the four corners of a cell in 32 contiguous bytes instead of the
// two separate 16-B column segments of the arma::mat layout -- measured 4.02 ms ->
// 2.27 / 2.08 ms for 1e8 queries on 4096^2):
//   column pairs  element (l,c) = {Z(l,c), Z(l,min(c+1,nx-1))}, 16 B; corners =
//                 elements (l,c),(l+1,c); 2x the input bytes
//   quad cells    element (l,c) = {Z(l,c), Z(l,rx), Z(ry,c), Z(ry,rx)}, 32 B, 32-B
//                 aligned = exactly one 64-B sector per query; 4x the input bytes
// HBM3E capacity (288 GB) is what makes spending 2-4x on a 128 MiB table sensible.
// Compiled with -ffp-contract=off (every product/sum rounds separately).
#include <algorithm>
#include <cmath>
#include <new>
#include <vector>

#include "mi_common.hpp"

typedef double d2 __attribute__((ext_vector_type(2)));

struct AxisDev {
    const double* nodes;   // explicit axis (null when implicit)
    int n;
    int use_guess;         // explicit: analytic guess + walk (1) or binary search (0)
    double xmin, xmax, scale;
    double x0, dx;         // implicit: node_i = fma(i, dx, x0)
};

typedef double d2v __attribute__((ext_vector_type(2)));

struct G2Dev {
    AxisDev ax, ay;
    const d2v* zp;         // (ny*nx + 1) column pairs {Z(l,c), Z(l,c+1)}, index l + c*ny
    int quads;             // 1: zp holds 2*ny*nx d2v: cell (l,c) = {Z(l,c), Z(l,rx)}, {Z(ry,c), Z(ry,rx)} (32 B)
};

struct mi_grid2 {
    mi_ctx* ctx;
    int device;            // copied at creation: destroy must not dereference a context that may be gone
    void* dev_x;
    void* dev_y;
    void* dev_z;
    G2Dev d;
    size_t table_bytes = 0;
};

namespace mi_interp2 {

constexpr int kBlock = 256;
constexpr int kMaxWalk = 4;

// IMPL: the axis is known to be implicit (uniform) at compile time -- no node loads, no search branches; the arithmetic
// is the one the general form takes for such an axis
template <bool IMPL = false>
__device__ __forceinline__ double axis_node(const AxisDev& a, int i)
{
    if constexpr (IMPL) return fma((double)i, a.dx, a.x0);
    return a.nodes ? a.nodes[i] : fma((double)i, a.dx, a.x0);
}

// largest l with node_l <= q (q inside [xmin, xmax])
template <bool IMPL = false>
__device__ __forceinline__ int axis_locate(const AxisDev& a, double q)
{
    if (!IMPL && a.nodes && !a.use_guess) {
        int lo = 0, hi = a.n;
        while (hi - lo > 1) {
            const int mid = lo + ((hi - lo) >> 1);
            if (a.nodes[mid] <= q) lo = mid; else hi = mid;
        }
        return lo;
    }
    int i = (int)((q - a.xmin) * a.scale);
    i = min(max(i, 0), a.n - 1);
    while (i > 0 && axis_node<IMPL>(a, i) > q) --i;
    while (i < a.n - 1 && axis_node<IMPL>(a, i + 1) <= q) ++i;
    return i;
}

__device__ __forceinline__ double weight(double xa, double xb, double q)
{
    const double a = q - xa, b = xb - q;
    return (a > 0.0) ? a / (a + b) : 0.0;
}

// eval2 in three steps, so that a kernel can put the cell loads of several queries in flight together
// (interp2_blocks_kernel); the direct kernel runs them back to back.  Same operations in the same order either way.
struct Loc2 {
    double sx, sy;       // the query, or the grid's origin for an out-of-range / NaN query (its result is replaced)
    int lx, ly, rx, ry;
    bool oor;
    const d2v* cell;     // two consecutive 16-B elements: {Z(ly,lx), Z(ly,rx)}, then the next row's pair / the quad's second half
};

template <bool IMPL = false>
__device__ __forceinline__ Loc2 locate2(const G2Dev& g, double qx, double qy)
{
    Loc2 L;
    L.oor = !(qx >= g.ax.xmin && qx <= g.ax.xmax && qy >= g.ay.xmin && qy <= g.ay.xmax);
    L.sx = L.oor ? g.ax.xmin : qx;
    L.sy = L.oor ? g.ay.xmin : qy;
    L.lx = axis_locate<IMPL>(g.ax, L.sx);
    L.ly = axis_locate<IMPL>(g.ay, L.sy);
    L.rx = min(L.lx + 1, g.ax.n - 1);
    L.ry = min(L.ly + 1, g.ay.n - 1);
    const size_t k = (size_t)L.lx * (size_t)g.ay.n + L.ly;
    L.cell = g.quads ? g.zp + 2 * k : g.zp + k;
    return L;
}

// lo = cell[0] = {Z(ly,lx), Z(ly,rx)}; hi = cell[1]: pairs: next row (padding past the last); quads: {Z(ry,lx), Z(ry,rx)}
template <bool IMPL = false>
__device__ __forceinline__ double blend2(const G2Dev& g, const Loc2& L, d2v lo, d2v hi, double qx, double qy, double extrap)
{
    const double wx = weight(axis_node<IMPL>(g.ax, L.lx), axis_node<IMPL>(g.ax, L.rx), L.sx);
    const double wy = weight(axis_node<IMPL>(g.ay, L.ly), axis_node<IMPL>(g.ay, L.ry), L.sy);
    const double z01 = (L.ry != L.ly) ? hi.x : lo.x;
    const double z11 = (L.ry != L.ly) ? hi.y : lo.y;
    const double c0 = (1.0 - wy) * lo.x + wy * z01;
    const double c1 = (1.0 - wy) * lo.y + wy * z11;
    const double r = (1.0 - wx) * c0 + wx * c1;
    if (L.oor) return (qx != qx || qy != qy) ? __builtin_nan("") : extrap;
    return r;
}

__device__ __forceinline__ double eval2(const G2Dev& g, double qx, double qy, double extrap)
{
    const Loc2 L = locate2(g, qx, qy);
    const d2v lo = L.cell[0], hi = L.cell[1];
    return blend2(g, L, lo, hi, qx, qy, extrap);
}

// One 16-B vector of each coordinate stream (two queries) per lane, one workgroup per 256 vectors: the
// launch shape that streams fastest on MI355X (see mi_interp1.hip).
template <bool VEC>
__global__ __launch_bounds__(kBlock) void interp2_kernel(G2Dev g, const double* __restrict__ xq,
                                                         const double* __restrict__ yq, double* __restrict__ zq,
                                                         size_t nq, double extrap)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if constexpr (VEC) {
        const size_t nvec = nq >> 1;
        if (i < nvec) {
            const d2 vx = __builtin_nontemporal_load(reinterpret_cast<const d2*>(xq) + i);
            const d2 vy = __builtin_nontemporal_load(reinterpret_cast<const d2*>(yq) + i);
            d2 o;
            o.x = eval2(g, vx.x, vy.x, extrap);
            o.y = eval2(g, vx.y, vy.y, extrap);
            __builtin_nontemporal_store(o, reinterpret_cast<d2*>(zq) + i);
        }
        if ((nq & 1) && i == nvec) zq[nq - 1] = eval2(g, xq[nq - 1], yq[nq - 1], extrap);
    } else {
        if (i < nq) zq[i] = eval2(g, xq[i], yq[i], extrap);
    }
}

}  // namespace mi_interp2

using namespace mi_interp2;

namespace {

// resident bytes of the table (mi_grid2_info)
void table_size(mi_grid2* g)
{
    g->table_bytes = (size_t)g->d.ax.n * (size_t)g->d.ay.n * (g->d.quads ? 32 : 16);
}

mi_status to_host(mi_ctx* ctx, const double* p, size_t n, bool dev, std::vector<double>& v)
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

mi_status make_explicit_axis(mi_ctx* ctx, const std::vector<double>& xs, const char* name, void** dev, AxisDev* a)
{
    const size_t n = xs.size();
    for (size_t i = 0; i < n; ++i)
        if (!std::isfinite(xs[i])) return mi::fail(ctx, MI_ERR_GRID, "mi_grid2_create: %s[%zu] is not finite", name, i);
    for (size_t i = 1; i < n; ++i)
        if (!(xs[i - 1] < xs[i]))
            return mi::fail(ctx, MI_ERR_GRID, "mi_grid2_create: %s not strictly increasing at %zu", name, i);
    memset(a, 0, sizeof(*a));
    a->n = (int)n;
    a->xmin = xs[0];
    a->xmax = xs[n - 1];
    const double scale = (double)(n - 1) / (a->xmax - a->xmin);
    long e_lo = 0, e_hi = 0;
    const bool ok = std::isfinite(scale) && scale > 0.0;
    if (ok) {
        for (size_t i = 0; i < n; ++i) {
            long gi = (long)(int)((xs[i] - a->xmin) * scale);
            gi = std::min<long>(std::max<long>(gi, 0), (long)n - 1);
            e_lo = std::min(e_lo, gi - (long)i);
            e_hi = std::max(e_hi, gi - (long)i);
        }
    }
    a->use_guess = ok && (e_hi - e_lo + 1) <= kMaxWalk;
    a->scale = ok ? scale : 0.0;
    hipError_t e = hipMalloc(dev, n * sizeof(double));
    if (e != hipSuccess) return mi::fail(ctx, MI_ERR_NOMEM, "hipMalloc axis failed: %s", hipGetErrorString(e));
    MI_HIP(ctx, hipMemcpy(*dev, xs.data(), n * sizeof(double), hipMemcpyHostToDevice));
    a->nodes = (const double*)*dev;
    return MI_OK;
}

mi_status make_uniform_axis(mi_ctx* ctx, double x0, double dx, size_t n, const char* name, AxisDev* a)
{
    if (!(dx > 0.0) || !std::isfinite(dx) || !std::isfinite(x0) || !std::isfinite(std::fma((double)(n - 1), dx, x0)) ||
        !(std::fma(1.0, dx, x0) > x0))
        return mi::fail(ctx, MI_ERR_GRID, "mi_grid2_create_uniform: bad %s axis (need finite origin, step > 0)", name);
    memset(a, 0, sizeof(*a));
    a->n = (int)n;
    a->x0 = x0;
    a->dx = dx;
    a->xmin = x0;
    a->xmax = std::fma((double)(n - 1), dx, x0);
    a->scale = 1.0 / dx;
    return MI_OK;
}

__global__ __launch_bounds__(kBlock) void quad_cells_kernel(const double* __restrict__ z, size_t ny, size_t nx,
                                                            d2v* __restrict__ zq)
{
    const size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x;      // k = l + c*ny
    if (k >= ny * nx) return;
    const size_t c = k / ny, l = k - c * ny;
    const size_t dr = (l + 1 < ny) ? 1 : 0, dc = (c + 1 < nx) ? ny : 0;
    d2v a, b;
    a.x = z[k];
    a.y = z[k + dc];
    b.x = z[k + dr];
    b.y = z[k + dr + dc];
    zq[2 * k] = a;
    zq[2 * k + 1] = b;
}

__global__ __launch_bounds__(kBlock) void pair_columns_kernel(const double* __restrict__ z, size_t ny, size_t nx,
                                                              d2v* __restrict__ zp)
{
    const size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x;      // k = l + c*ny
    const size_t count = ny * nx;
    if (k < count) {
        const size_t c = k / ny;
        d2v v;
        v.x = z[k];
        v.y = (c + 1 < nx) ? z[k + ny] : z[k];                       // rx = min(c+1, nx-1)
        zp[k] = v;
    } else if (k == count) {
        d2v v;
        v.x = 0.0;
        v.y = 0.0;
        zp[k] = v;                                                   // padding element (never selected)
    }
}

// column-major input (host or device) -> resident column-pair layout
mi_status upload_z(mi_ctx* ctx, const double* z, size_t ny, size_t nx, bool dev, void** out, bool quads)
{
    const size_t count = ny * nx;
    hipError_t e = hipMalloc(out, (quads ? 2 * count : count + 1) * sizeof(d2v));
    if (e != hipSuccess)
        return mi::fail(ctx, MI_ERR_NOMEM, "hipMalloc(%zu) for Z failed: %s", (count + 1) * sizeof(d2v), hipGetErrorString(e));
    const double* src = z;
    void* staging = nullptr;
    if (!dev) {
        e = hipMalloc(&staging, count * sizeof(double));
        if (e != hipSuccess) return mi::fail(ctx, MI_ERR_NOMEM, "hipMalloc staging for Z failed: %s", hipGetErrorString(e));
        MI_HIP(ctx, hipMemcpy(staging, z, count * sizeof(double), hipMemcpyHostToDevice));
        src = (const double*)staging;
    }
    const size_t grid = (count + 1 + kBlock - 1) / kBlock;
    if (quads)
        hipLaunchKernelGGL(quad_cells_kernel, dim3((unsigned)grid), dim3(kBlock), 0, ctx->stream, src, ny, nx, (d2v*)*out);
    else
        hipLaunchKernelGGL(pair_columns_kernel, dim3((unsigned)grid), dim3(kBlock), 0, ctx->stream, src, ny, nx, (d2v*)*out);
    hipError_t le = hipGetLastError();
    hipError_t se = hipStreamSynchronize(ctx->stream);
    if (staging) (void)hipFree(staging);
    if (le != hipSuccess || se != hipSuccess)
        return mi::fail(ctx, MI_ERR_HIP, "pair_columns_kernel failed: %s", hipGetErrorString(le != hipSuccess ? le : se));
    return MI_OK;
}

// Resident layout choice.  Quad cells (32 B per cell, exactly one 64-B sector per query) measured 8 % faster
// than column pairs (16 B per element, 1.25 sectors per query) on the 4096^2 / 1e8-query config at twice the
// bytes; they are the default while the table stays under 16 GiB, MI_GRID2_COMPACT forces pairs.
bool want_quads(unsigned flags, size_t ny, size_t nx)
{
    if (flags & MI_GRID2_COMPACT) return false;
    return (double)ny * (double)nx * 32.0 <= 16.0 * 1024.0 * 1024.0 * 1024.0;
}

void destroy(mi_grid2* g)
{
    if (!g) return;
    if (g->dev_x) (void)hipFree(g->dev_x);
    if (g->dev_y) (void)hipFree(g->dev_y);
    if (g->dev_z) (void)hipFree(g->dev_z);
    delete g;
}

}  // namespace

extern "C" {

mi_status mi_grid2_create(mi_ctx* ctx, const double* x, size_t nx, const double* y, size_t ny, const double* z,
                          unsigned flags, mi_grid2** out)
{
    MI_REQUIRE(ctx, ctx && x && y && z && out, "mi_grid2_create: NULL argument");
    MI_REQUIRE(ctx, (flags & ~(MI_GRID_DEVICE_PTRS | MI_GRID2_COMPACT)) == 0, "mi_grid2_create: unknown flags 0x%x", flags);
    *out = nullptr;
    if (nx < 2 || ny < 2) return mi::fail(ctx, MI_ERR_GRID, "mi_grid2_create: need at least 2x2 nodes");
    MI_REQUIRE(ctx, nx < 0x7ffffff0u && ny < 0x7ffffff0u, "mi_grid2_create: axis too long");
    MI_HIP(ctx, hipSetDevice(ctx->device));
    const bool dev = flags & MI_GRID_DEVICE_PTRS;
    std::vector<double> xs, ys;
    mi_status st = to_host(ctx, x, nx, dev, xs);
    if (st != MI_OK) return st;
    st = to_host(ctx, y, ny, dev, ys);
    if (st != MI_OK) return st;
    mi_grid2* g = new (std::nothrow) mi_grid2();
    if (!g) return mi::fail(ctx, MI_ERR_NOMEM, "mi_grid2_create: out of host memory");
    g->ctx = ctx;
    g->device = ctx->device;
    g->dev_x = g->dev_y = g->dev_z = nullptr;
    st = make_explicit_axis(ctx, xs, "X", &g->dev_x, &g->d.ax);
    if (st == MI_OK) st = make_explicit_axis(ctx, ys, "Y", &g->dev_y, &g->d.ay);
    const bool quads = want_quads(flags, ny, nx);
    if (st == MI_OK) st = upload_z(ctx, z, ny, nx, dev, &g->dev_z, quads);
    g->d.quads = quads;
    if (st != MI_OK) { destroy(g); return st; }
    g->d.zp = (const d2v*)g->dev_z;
    table_size(g);
    *out = g;
    return MI_OK;
}

mi_status mi_grid2_create_uniform(mi_ctx* ctx, double x0, double dx, size_t nx, double y0, double dy, size_t ny,
                                  const double* z, unsigned flags, mi_grid2** out)
{
    MI_REQUIRE(ctx, ctx && z && out, "mi_grid2_create_uniform: NULL argument");
    MI_REQUIRE(ctx, (flags & ~(MI_GRID_DEVICE_PTRS | MI_GRID2_COMPACT)) == 0, "mi_grid2_create_uniform: unknown flags 0x%x", flags);
    *out = nullptr;
    if (nx < 2 || ny < 2) return mi::fail(ctx, MI_ERR_GRID, "mi_grid2_create_uniform: need at least 2x2 nodes");
    MI_REQUIRE(ctx, nx < 0x7ffffff0u && ny < 0x7ffffff0u, "mi_grid2_create_uniform: axis too long");
    MI_HIP(ctx, hipSetDevice(ctx->device));
    mi_grid2* g = new (std::nothrow) mi_grid2();
    if (!g) return mi::fail(ctx, MI_ERR_NOMEM, "mi_grid2_create_uniform: out of host memory");
    g->ctx = ctx;
    g->device = ctx->device;
    g->dev_x = g->dev_y = g->dev_z = nullptr;
    mi_status st = make_uniform_axis(ctx, x0, dx, nx, "x", &g->d.ax);
    if (st == MI_OK) st = make_uniform_axis(ctx, y0, dy, ny, "y", &g->d.ay);
    const bool quads = want_quads(flags, ny, nx);
    if (st == MI_OK) st = upload_z(ctx, z, ny, nx, flags & MI_GRID_DEVICE_PTRS, &g->dev_z, quads);
    g->d.quads = quads;
    if (st != MI_OK) { destroy(g); return st; }
    g->d.zp = (const d2v*)g->dev_z;
    table_size(g);
    *out = g;
    return MI_OK;
}

mi_status mi_grid2_destroy(mi_grid2* g)
{
    if (g) (void)hipSetDevice(g->device);
    destroy(g);
    return MI_OK;
}

mi_status mi_grid2_info(const mi_grid2* g, size_t* table_bytes)
{
    MI_REQUIRE(nullptr, g != nullptr, "mi_grid2_info: grid is NULL");
    if (table_bytes) *table_bytes = g->table_bytes;
    return MI_OK;
}

mi_status mi_interp2_f64_dev(mi_ctx* ctx, const mi_grid2* g, const double* xq, const double* yq, double* zq, size_t nq,
                             double extrap)
{
    MI_REQUIRE(ctx, ctx && g, "mi_interp2_f64_dev: NULL context or grid");
    if (nq == 0) return MI_OK;
    MI_REQUIRE(ctx, xq && yq && zq, "mi_interp2_f64_dev: NULL query/result pointer");
    const uintptr_t a = reinterpret_cast<uintptr_t>(xq) | reinterpret_cast<uintptr_t>(yq) | reinterpret_cast<uintptr_t>(zq);
    MI_REQUIRE(ctx, (a & 7u) == 0, "mi_interp2_f64_dev: pointers must be 8-byte aligned");
    MI_HIP(ctx, hipSetDevice(ctx->device));   // a process may hold contexts on several devices (mi_group)
    const bool vec = (a & 15u) == 0;
    const size_t lanes = vec ? (nq >> 1) + (nq & 1) : nq;
    const size_t grid = (lanes + kBlock - 1) / kBlock;
    if (grid > 0x7fffffffull) return mi::fail(ctx, MI_ERR_INVALID_ARG, "mi_interp2_f64_dev: nq=%zu too large for one launch", nq);
    if (vec)
        hipLaunchKernelGGL((interp2_kernel<true>), dim3((unsigned)grid), dim3(kBlock), 0, ctx->stream, g->d, xq, yq, zq, nq, extrap);
    else
        hipLaunchKernelGGL((interp2_kernel<false>), dim3((unsigned)grid), dim3(kBlock), 0, ctx->stream, g->d, xq, yq, zq, nq, extrap);
    MI_LAUNCH_CHECK(ctx, "interp2 kernel");
    return MI_OK;
}

mi_status mi_interp2_f64_host(mi_ctx* ctx, const mi_grid2* g, const double* xq, const double* yq, double* zq, size_t nq,
                              double extrap)
{
    MI_REQUIRE(ctx, ctx && g, "mi_interp2_f64_host: NULL context or grid");
    if (nq == 0) return MI_OK;
    MI_REQUIRE(ctx, xq && yq && zq, "mi_interp2_f64_host: NULL query/result pointer");
    MI_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = nq * sizeof(double);
    mi_status st = MI_OK;
    for (int s = 0; s < 3 && st == MI_OK; ++s) st = mi::ensure_scratch(ctx, s, bytes);
    if (st != MI_OK) return st;
    const size_t chunk = (size_t)8 << 20;
    if (nq <= 2 * chunk) {
        MI_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], xq, bytes, hipMemcpyHostToDevice, ctx->stream));
        MI_HIP(ctx, hipMemcpyAsync(ctx->scratch[1], yq, bytes, hipMemcpyHostToDevice, ctx->stream));
        st = mi_interp2_f64_dev(ctx, g, (const double*)ctx->scratch[0], (const double*)ctx->scratch[1],
                                (double*)ctx->scratch[2], nq, extrap);
        if (st != MI_OK) return st;
        MI_HIP(ctx, hipMemcpyAsync(zq, ctx->scratch[2], bytes, hipMemcpyDeviceToHost, ctx->stream));
        MI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return MI_OK;
    }
    // chunked and pinned like mi_interp1_f64_host: the copy back of chunk k overlaps the uploads of chunk k+1
    st = mi::ensure_aux_stream(ctx);
    if (st != MI_OK) return st;
    const bool pin_x = mi::pin_host(xq, bytes), pin_y = mi::pin_host(yq, bytes), pin_z = mi::pin_host(zq, bytes);
    double *dx = (double*)ctx->scratch[0], *dy = (double*)ctx->scratch[1], *dz = (double*)ctx->scratch[2];
    // as in mi_interp1_f64_host: no early return before both streams are drained and the ranges released
    hipError_t herr = hipSuccess;
    const char* what = "";
    if (getenv("MI_TEST_FAIL_HOST_CHUNK")) { herr = hipErrorUnknown; what = "MI_TEST_FAIL_HOST_CHUNK (error-path test hook)"; }
    for (size_t off = 0; off < nq && herr == hipSuccess && st == MI_OK; off += chunk) {
        const size_t m = std::min(chunk, nq - off);
        herr = hipMemcpyAsync(dx + off, xq + off, m * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (herr == hipSuccess) herr = hipMemcpyAsync(dy + off, yq + off, m * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (herr != hipSuccess) { what = "upload of a query chunk"; break; }
        st = mi_interp2_f64_dev(ctx, g, dx + off, dy + off, dz + off, m, extrap);
        if (st != MI_OK) break;
        herr = hipEventRecord(ctx->aux_event, ctx->stream);
        if (herr == hipSuccess) herr = hipStreamWaitEvent(ctx->aux_stream, ctx->aux_event, 0);
        if (herr == hipSuccess) herr = hipMemcpyAsync(zq + off, dz + off, m * sizeof(double), hipMemcpyDeviceToHost, ctx->aux_stream);
        if (herr != hipSuccess) what = "download of a result chunk";
    }
    const hipError_t e1 = hipStreamSynchronize(ctx->stream), e2 = hipStreamSynchronize(ctx->aux_stream);
    if (pin_x) mi::unpin_host(xq);
    if (pin_y) mi::unpin_host(yq);
    if (pin_z) mi::unpin_host(zq);
    if (st != MI_OK) return st;
    if (herr != hipSuccess) return mi::fail(ctx, MI_ERR_HIP, "mi_interp2_f64_host: %s failed: %s", what, hipGetErrorString(herr));
    MI_HIP(ctx, e1);
    MI_HIP(ctx, e2);
    return MI_OK;
}

}  // extern "C"
