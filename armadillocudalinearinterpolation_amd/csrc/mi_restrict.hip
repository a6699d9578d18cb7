// The reference's own interpolation step on MI355X: the two-point "restrict"
// blend (EventDrivenMap.cu:769-785) and the masked mean over realisations that
// follows it (EventDrivenMap.cu:787-824), separately and fused in one pass.
//
// Launch shape: the reference launches S*R workgroups of N threads for S*R
// elements (:205); here one lane owns 4 consecutive elements and the grid is
// capped at 8 workgroups per CU.  HBM-bound: 12 B read + 4 B written per
// element (fused with the mean and no materialised output: 12 B + accept).
//
// fp32 operation order == oracle/interp_oracle.c orc_restrict_f32:
//   h = (2*L)/N ; x_k = fmaf(h, ind_k, -L) ; out = x0 + ((T-t0)*(x1-x0))/(t1-t0)
// Compiled with -ffp-contract=off and correctly rounded fp32 division.
#include <new>

#include "mi_common.hpp"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxSpikes = 8;
constexpr int kMaxPartialBlocks = 2048;

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float restrict_one(float t0, unsigned i0, float t1, unsigned i1, float T, float L, float h)
{
    const float x0 = fmaf(h, (float)i0, -L);
    const float x1 = fmaf(h, (float)i1, -L);
    const float num = (T - t0) * (x1 - x0);
    const float q = num / (t1 - t0);
    return x0 + q;
}

__global__ __launch_bounds__(kBlock) void restrict_vec_kernel(const float* t0, const unsigned short* __restrict__ i0,
                                                              const float* __restrict__ t1,
                                                              const unsigned short* __restrict__ i1, float T, float L,
                                                              float h, float* out, size_t n)
{
    const size_t nvec = n >> 2;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
        const f4 a = reinterpret_cast<const f4*>(t0)[i];
        const f4 b = reinterpret_cast<const f4*>(t1)[i];
        const us4 ia = reinterpret_cast<const us4*>(i0)[i];
        const us4 ib = reinterpret_cast<const us4*>(i1)[i];
        f4 r;
        r.x = restrict_one(a.x, ia.x, b.x, ib.x, T, L, h);
        r.y = restrict_one(a.y, ia.y, b.y, ib.y, T, L, h);
        r.z = restrict_one(a.z, ia.z, b.z, ib.z, T, L, h);
        r.w = restrict_one(a.w, ia.w, b.w, ib.w, T, L, h);
        reinterpret_cast<f4*>(out)[i] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t k = (nvec << 2) + threadIdx.x;
        out[k] = restrict_one(t0[k], i0[k], t1[k], i1[k], T, L, h);
    }
}

__global__ __launch_bounds__(kBlock) void restrict_scalar_kernel(const float* t0, const unsigned short* __restrict__ i0,
                                                                 const float* __restrict__ t1,
                                                                 const unsigned short* __restrict__ i1, float T,
                                                                 float L, float h, float* out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x; k < n; k += stride)
        out[k] = restrict_one(t0[k], i0[k], t1[k], i1[k], T, L, h);
}

// ---- masked mean ------------------------------------------------------------
// Stage 1: every workgroup reduces its grid-stride share of the realisations to
// nspikes fp64 partial sums + an accepted count (wave64 shuffles, then one LDS
// hop across the 4 waves) and stores them; stage 2: one workgroup adds the
// partials in a fixed order.  No float atomics: the result is bitwise
// reproducible from run to run.
struct Partials {
    double sum[kMaxPartialBlocks][kMaxSpikes];
    unsigned count[kMaxPartialBlocks];
    float x_real0[kMaxSpikes];   // restricted value of realisation 0 (quirk mode)
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ unsigned wave_sum_u(unsigned v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// FUSED: x is computed from the four event arrays; otherwise read from xin.
template <bool FUSED>
__global__ __launch_bounds__(kBlock) void mean_stage1_kernel(const float* __restrict__ xin,
                                                             const float* __restrict__ t0,
                                                             const unsigned short* __restrict__ i0,
                                                             const float* __restrict__ t1,
                                                             const unsigned short* __restrict__ i1,
                                                             const unsigned* __restrict__ accept, float T, float L,
                                                             float h, size_t nreal, int nspikes, int quirk,
                                                             float* restricted, Partials* part)
{
    double acc[kMaxSpikes];
#pragma unroll
    for (int m = 0; m < kMaxSpikes; ++m) acc[m] = 0.0;
    unsigned cnt = 0;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t r = (size_t)blockIdx.x * kBlock + threadIdx.x; r < nreal; r += stride) {
        const unsigned flag = accept[r];
        cnt += flag;
        const bool take = (flag == 1u) && !(quirk && r == 0);
#pragma unroll
        for (int m = 0; m < kMaxSpikes; ++m) {
            if (m < nspikes) {
                const size_t k = (size_t)m * nreal + r;
                float x;
                if constexpr (FUSED) {
                    x = restrict_one(t0[k], i0[k], t1[k], i1[k], T, L, h);
                    if (restricted) restricted[k] = x;
                } else {
                    x = xin[k];
                }
                if (take) acc[m] += (double)x;
                if (r == 0) part->x_real0[m] = x;
            }
        }
    }
    __shared__ double lds_sum[kBlock / 64][kMaxSpikes];
    __shared__ unsigned lds_cnt[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int m = 0; m < kMaxSpikes; ++m) {
        const double s = wave_sum(acc[m]);
        if (lane == 0) lds_sum[wave][m] = s;
    }
    const unsigned c = wave_sum_u(cnt);
    if (lane == 0) lds_cnt[wave] = c;
    __syncthreads();
    if (threadIdx.x < kMaxSpikes) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) s += lds_sum[w][threadIdx.x];
        part->sum[blockIdx.x][threadIdx.x] = s;
    }
    if (threadIdx.x == 0) {
        unsigned s = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) s += lds_cnt[w];
        part->count[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(kBlock) void mean_stage2_kernel(const Partials* __restrict__ part, int nblocks,
                                                             int nspikes, int quirk, float* mean, unsigned* count_out,
                                                             double* sums_out)
{
    __shared__ double lds_sum[kBlock / 64][kMaxSpikes];
    __shared__ unsigned lds_cnt[kBlock / 64];
    double acc[kMaxSpikes];
#pragma unroll
    for (int m = 0; m < kMaxSpikes; ++m) acc[m] = 0.0;
    unsigned cnt = 0;
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
#pragma unroll
        for (int m = 0; m < kMaxSpikes; ++m) acc[m] += part->sum[b][m];
        cnt += part->count[b];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int m = 0; m < kMaxSpikes; ++m) {
        const double s = wave_sum(acc[m]);
        if (lane == 0) lds_sum[wave][m] = s;
    }
    const unsigned c = wave_sum_u(cnt);
    if (lane == 0) lds_cnt[wave] = c;
    __syncthreads();
    if ((int)threadIdx.x < nspikes) {
        unsigned total = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) total += lds_cnt[w];
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) s += lds_sum[w][threadIdx.x];
        // reference quirk: accept[0] holds the count when the mean runs, so
        // realisation 0 is summed iff count == 1 (EventDrivenMap.cu:800-802,:817)
        const double x0 = quirk ? (double)part->x_real0[threadIdx.x] : 0.0;
        // partial block [sums | count | x0] (MI_EDM_PARTIAL_LEN): the sums WITHOUT that re-inclusion, so that the
        // blocks of several shards add up and the rule is applied once, to the total (mi_edm_residual_from_sums)
        if (sums_out) {
            sums_out[threadIdx.x] = s;
            sums_out[nspikes + 1 + threadIdx.x] = x0;
            if (threadIdx.x == 0) sums_out[nspikes] = (double)total;
        }
        if (quirk && total == 1u) s += x0;
        mean[threadIdx.x] = (float)(s / (double)total);    // one rounding: exact for identical realisations
        if (threadIdx.x == 0 && count_out) *count_out = total;
    }
}

mi_status mean_workspace(mi_ctx* ctx, Partials** out)
{
    static_assert(sizeof(Partials) <= mi_ctx::kReduceWsBytes, "reduction workspace too small");
    *out = (Partials*)ctx->reduce_ws;
    return MI_OK;
}

}  // namespace

extern "C" {

mi_status mi_restrict_f32_dev(mi_ctx* ctx, const float* t0, const uint16_t* i0, const float* t1, const uint16_t* i1,
                              float T, float L, uint32_t ngrid, float* out, size_t n)
{
    MI_REQUIRE(ctx, ctx != nullptr, "mi_restrict_f32_dev: ctx is NULL");
    if (n == 0) return MI_OK;
    MI_REQUIRE(ctx, t0 && i0 && t1 && i1 && out, "mi_restrict_f32_dev: NULL array pointer");
    MI_REQUIRE(ctx, ngrid > 0, "mi_restrict_f32_dev: ngrid must be positive");
    MI_HIP(ctx, hipSetDevice(ctx->device));
    const float h = (2.0f * L) / (float)ngrid;
    const uintptr_t a16 = reinterpret_cast<uintptr_t>(t0) | reinterpret_cast<uintptr_t>(t1) | reinterpret_cast<uintptr_t>(out);
    const uintptr_t a8 = reinterpret_cast<uintptr_t>(i0) | reinterpret_cast<uintptr_t>(i1);
    if (((a16 & 15u) | (a8 & 7u)) == 0) {
        const unsigned grid = mi::stream_grid(ctx, (n + 3) / 4, kBlock);
        hipLaunchKernelGGL(restrict_vec_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, t0, i0, t1, i1, T, L, h, out, n);
    } else {
        const unsigned grid = mi::stream_grid(ctx, n, kBlock);
        hipLaunchKernelGGL(restrict_scalar_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, t0, i0, t1, i1, T, L, h, out, n);
    }
    MI_LAUNCH_CHECK(ctx, "restrict kernel");
    return MI_OK;
}

mi_status mi_restrict_f32_host(mi_ctx* ctx, const float* t0, const uint16_t* i0, const float* t1, const uint16_t* i1,
                               float T, float L, uint32_t ngrid, float* out, size_t n)
{
    MI_REQUIRE(ctx, ctx != nullptr, "mi_restrict_f32_host: ctx is NULL");
    if (n == 0) return MI_OK;
    MI_REQUIRE(ctx, t0 && i0 && t1 && i1 && out, "mi_restrict_f32_host: NULL array pointer");
    MI_HIP(ctx, hipSetDevice(ctx->device));
    // one scratch slot: [t0 | t1 | i0 | i1], each section 16-byte aligned
    const size_t fb = (n * sizeof(float) + 15) & ~size_t(15), hb = (n * sizeof(uint16_t) + 15) & ~size_t(15);
    mi_status st = mi::ensure_scratch(ctx, 0, 2 * fb + 2 * hb);
    if (st != MI_OK) return st;
    char* base = (char*)ctx->scratch[0];
    float* d_t0 = (float*)base;
    float* d_t1 = (float*)(base + fb);
    uint16_t* d_i0 = (uint16_t*)(base + 2 * fb);
    uint16_t* d_i1 = (uint16_t*)(base + 2 * fb + hb);
    MI_HIP(ctx, hipMemcpyAsync(d_t0, t0, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(ctx, hipMemcpyAsync(d_t1, t1, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(ctx, hipMemcpyAsync(d_i0, i0, n * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(ctx, hipMemcpyAsync(d_i1, i1, n * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
    st = mi_restrict_f32_dev(ctx, d_t0, d_i0, d_t1, d_i1, T, L, ngrid, d_t0, n);       // in place, like the reference
    if (st != MI_OK) return st;
    MI_HIP(ctx, hipMemcpyAsync(out, d_t0, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MI_OK;
}

mi_status mi_masked_mean_f32_dev(mi_ctx* ctx, const float* x, const uint32_t* accept, size_t nreal, size_t nspikes,
                                 int quirk, float* mean_dev, uint32_t* count_dev, double* sums_dev)
{
    MI_REQUIRE(ctx, ctx != nullptr, "mi_masked_mean_f32_dev: ctx is NULL");
    MI_REQUIRE(ctx, x && accept && mean_dev, "mi_masked_mean_f32_dev: NULL pointer");
    MI_REQUIRE(ctx, nspikes >= 1 && nspikes <= (size_t)kMaxSpikes, "mi_masked_mean_f32_dev: nspikes=%zu not in [1,%d]", nspikes, kMaxSpikes);
    MI_REQUIRE(ctx, nreal >= 1, "mi_masked_mean_f32_dev: nreal must be positive");
    MI_HIP(ctx, hipSetDevice(ctx->device));
    Partials* part = nullptr;
    mi_status st = mean_workspace(ctx, &part);
    if (st != MI_OK) return st;
    unsigned grid = mi::stream_grid(ctx, nreal, kBlock);
    if (grid > (unsigned)kMaxPartialBlocks) grid = kMaxPartialBlocks;
    hipLaunchKernelGGL((mean_stage1_kernel<false>), dim3(grid), dim3(kBlock), 0, ctx->stream, x, nullptr, nullptr, nullptr,
                       nullptr, accept, 0.f, 0.f, 0.f, nreal, (int)nspikes, quirk, nullptr, part);
    MI_LAUNCH_CHECK(ctx, "masked-mean stage 1");
    hipLaunchKernelGGL(mean_stage2_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, part, (int)grid, (int)nspikes, quirk,
                       mean_dev, count_dev, sums_dev);
    MI_LAUNCH_CHECK(ctx, "masked-mean stage 2");
    return MI_OK;
}

mi_status mi_restrict_mean_f32_dev(mi_ctx* ctx, const float* t0, const uint16_t* i0, const float* t1, const uint16_t* i1,
                                   const uint32_t* accept, float T, float L, uint32_t ngrid, size_t nreal,
                                   size_t nspikes, int quirk, float* restricted_dev, float* mean_dev,
                                   uint32_t* count_dev, double* sums_dev)
{
    MI_REQUIRE(ctx, ctx != nullptr, "mi_restrict_mean_f32_dev: ctx is NULL");
    MI_REQUIRE(ctx, t0 && i0 && t1 && i1 && accept && mean_dev, "mi_restrict_mean_f32_dev: NULL pointer");
    MI_REQUIRE(ctx, nspikes >= 1 && nspikes <= (size_t)kMaxSpikes, "mi_restrict_mean_f32_dev: nspikes=%zu not in [1,%d]", nspikes, kMaxSpikes);
    MI_REQUIRE(ctx, nreal >= 1 && ngrid > 0, "mi_restrict_mean_f32_dev: nreal and ngrid must be positive");
    MI_HIP(ctx, hipSetDevice(ctx->device));
    Partials* part = nullptr;
    mi_status st = mean_workspace(ctx, &part);
    if (st != MI_OK) return st;
    const float h = (2.0f * L) / (float)ngrid;
    unsigned grid = mi::stream_grid(ctx, nreal, kBlock);
    if (grid > (unsigned)kMaxPartialBlocks) grid = kMaxPartialBlocks;
    hipLaunchKernelGGL((mean_stage1_kernel<true>), dim3(grid), dim3(kBlock), 0, ctx->stream, nullptr, t0, i0, t1, i1, accept,
                       T, L, h, nreal, (int)nspikes, quirk, restricted_dev, part);
    MI_LAUNCH_CHECK(ctx, "restrict+mean stage 1");
    hipLaunchKernelGGL(mean_stage2_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, part, (int)grid, (int)nspikes, quirk,
                       mean_dev, count_dev, sums_dev);
    MI_LAUNCH_CHECK(ctx, "restrict+mean stage 2");
    return MI_OK;
}

}  // extern "C"
