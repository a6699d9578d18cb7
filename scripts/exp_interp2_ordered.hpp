// RECORDED EXPERIMENT, not product code (moved out of csrc/ in round 4; last product form: commit 6d0f3e1 and before,
// where csrc/mi_interp2.hip included this file and include/mi355_interp.h exported mi_grid2_reserve /
// mi_ctx_set_interp2_path).  The call-wide cell ordering below was bit-identical to the direct bilinear kernel on every
// test and fuzz case (tests/test_interp_gpu.py::test_ordered_bilinear_path_* at that commit, 48 332 fuzz cases), moved
// 9.55 GB instead of 15.1 GB per 1e8 queries on the 4096^2 table -- and took 2.55 ms against 2.06 ms
// (profiles/r03_config3_ordered_*: its block pass is bound by the CUs' vector-memory request rate, four poorly coalesced
// lane-requests per query).  A pass 1 over 16 384-query tiles in SoA form (64-record runs) was priced on paper in round 4
// at 0.7 + 0.8 + 0.31 = 1.8-2.1 ms against the direct kernel's 2.06-2.09 ms: not built.
// It does not compile on its own: it needs the G2Dev / locate2 / blend2 definitions of csrc/mi_interp2.hip.
// Scattered bilinear interpolation, call-wide cell ordering (round 3; BASELINE.json configs[2]).
//
// The direct kernel (interp2_kernel, mi_interp2.hip) reads one random 32-B quad cell per query from a table far
// larger than L2, and every such miss moves a whole 128-B line between the fabric and L2: 15.1 GB per 1e8 queries on
// the 4096^2 table, 7.35 TB/s of fabric traffic for 2.5 GB of algorithmic bytes (profiles/r02_config3_interp2_*).
// What cuts it is more than one query per fetched line, i.e. the queries of the WHOLE call ordered by table block:
//   pass 1  interp2_order_kernel   tiles of 4096 queries: coarse block of every (x, y) (no bracket search: any block
//                                  assignment is correct, it only decides WHEN a query is evaluated), counting sort by
//                                  block in LDS, the sorted (x, y) pairs written tile by tile (coalesced 16-B stores),
//                                  each query's sorted position (u16) and the tile's block offsets beside them;
//   pass 2  interp2_blocks_kernel  block by block -- the workgroups of an XCD all on the same block at the same time,
//                                  so the block's cells (<= 2 MiB) are fetched into that XCD's L2 once and every further
//                                  query of the block is an L2 hit -- evaluate eval2() (the direct kernel's arithmetic:
//                                  results are bit-identical) on each tile's segment of the block, results written in
//                                  the same sorted positions;
//   pass 3  interp2_unsort_kernel  tile by tile: results back into query order through LDS (coalesced both ways).
// Traffic per query: 16 + 16 + 2 (pass 1), 16 + 8 (pass 2), 8 + 2 + 8 (pass 3) = 76 B of coalesced streams plus the
// table once per call, against 128 + 24 B for the direct kernel.  The workspace (26 B per query) belongs to the grid
// (mi_grid2_reserve), so the entry point itself allocates nothing.
//
// MEASURED (round 3, 1e8 queries, 4096^2 quad-cell table; profiles/r03_config3_ordered_*): fabric traffic 9.55 GB per call
// (pass 1: 1.60 GB read + 2.05 GB written, pass 2: 2.90 + 1.22, pass 3: 1.00 + 0.80) against 15.1 GB for the direct
// kernel -- and 2.55 ms against 2.06 ms: pass 1 0.64 ms, pass 2 1.60 ms, pass 3 0.31 ms.  Pass 2 is bound neither by
// memory bandwidth (102 M of its 138 M L2 requests hit; 83 G requests/s) nor by arithmetic (moving both fp64 divisions
// into pass 1 halved its vector-ALU work, 4.5e8 -> 2.0e8 wave-instructions, and left it at 1.75-2.05 ms with 45 % of the
// wave cycles waiting) nor by the queue atomics (eight items per atomic: 1.77 ms) nor by memory-channel camping (a tile
// pitch that is not a power of two: 1.75 ms): it is bound by the CU's vector memory path, which takes about one lane-request
// per ns per CU when the requests are not coalesced into long runs.  A query costs pass 2 four
// lane-requests -- its (x, y) record, the two 16-B halves of its cell, its result (five with weights + cell code records) --
// = 4e8 / (256 CUs x 1.05 per ns) = 1.5 ms; phase stamps show the 1024 record loads of a work item taking 7-25 us and its
// cell loads (L2 hits) 4-16 us behind the requests of the CU's other workgroups.  The direct kernel issues 3.5 lane-requests
// per query (1.3 ms) and hides them under its 2 ms of line fetches.  So the ordered path is NOT what mi_interp2_f64_dev
// takes by default (MI_INTERP2_ORDERED selects it); it stays as a tested alternative with bit-identical results.
#pragma once
#include "mi_common.hpp"

namespace mi_interp2 {

constexpr int kOrdTile = 4096;          // queries per tile (64 KiB of LDS for the sorted pairs)
constexpr int kOrdThreads = 512;        // pass 1: 8 queries per lane
constexpr int kOrdMaxBlocks = 1024;     // counting-sort bins (LDS histogram)
constexpr int kOrdSegs = 64;            // pass 2: tile segments scanned together (one wave-wide prefix)
constexpr int kOrdEvalThreads = 256;
constexpr int kOrdEvalUnroll = 4;       // pass 2: queries per lane in flight together

struct OrdGeom {
    int sx, sy;          // block of cell (lx, ly) = (lx >> sx) * nby + (ly >> sy)
    int nbx, nby, nb;    // nb = nbx * nby <= kOrdMaxBlocks
};

// approximate cell index along one axis (the guess the bracket search starts from): a linear map, clamped; NaN and
// out-of-range queries land in an end cell -- any block is a correct place to evaluate them
__device__ __forceinline__ int coarse_index(const AxisDev& a, double q)
{
    const int i = (int)((q - a.xmin) * a.scale);
    return min(max(i, 0), a.n - 1);
}

__device__ __forceinline__ int block_of(const G2Dev& g, const OrdGeom& o, double qx, double qy)
{
    return (coarse_index(g.ax, qx) >> o.sx) * o.nby + (coarse_index(g.ay, qy) >> o.sy);
}

// ---- pass 1 ------------------------------------------------------------------------------------------------
// off is block-major: off[b * off_stride + tile] = first sorted position of block b in that tile, row nb = kOrdTile
__global__ __launch_bounds__(kOrdThreads) void interp2_order_kernel(G2Dev g, OrdGeom o, const double* __restrict__ xq,
                                                                    const double* __restrict__ yq, size_t ntiles,
                                                                    d2* __restrict__ ws_xy, unsigned* __restrict__ ws_sp2,
                                                                    unsigned short* __restrict__ ws_off, size_t off_stride,
                                                                    unsigned* __restrict__ queues)
{
    __shared__ d2 sxy[kOrdTile];
    __shared__ unsigned hist[kOrdMaxBlocks];
    __shared__ unsigned wsum[kOrdThreads / 64];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid < 8) queues[tid] = 0;   // pass 2's per-XCD work queues (the kernel boundary orders this before pass 2)
    constexpr int V = kOrdTile / 2 / kOrdThreads;   // 16-B vectors per lane per coordinate (4)
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const d2* x2 = reinterpret_cast<const d2*>(xq + t * kOrdTile);
        const d2* y2 = reinterpret_cast<const d2*>(yq + t * kOrdTile);
        d2 vx[V], vy[V];
#pragma unroll
        for (int u = 0; u < V; ++u) {
            vx[u] = __builtin_nontemporal_load(x2 + tid + u * kOrdThreads);
            vy[u] = __builtin_nontemporal_load(y2 + tid + u * kOrdThreads);
        }
        for (int b = tid; b < kOrdMaxBlocks; b += kOrdThreads) hist[b] = 0;
        __syncthreads();
        unsigned short bin[2 * V], rank[2 * V];
#pragma unroll
        for (int u = 0; u < V; ++u) {
            const int b0 = block_of(g, o, vx[u].x, vy[u].x), b1 = block_of(g, o, vx[u].y, vy[u].y);
            bin[2 * u] = (unsigned short)b0;
            bin[2 * u + 1] = (unsigned short)b1;
            rank[2 * u] = (unsigned short)atomicAdd(&hist[b0], 1u);
            rank[2 * u + 1] = (unsigned short)atomicAdd(&hist[b1], 1u);
        }
        __syncthreads();
        {   // exclusive prefix over the kOrdMaxBlocks bins: two bins per lane, wave scan, wave totals through LDS
            const unsigned v0 = hist[2 * tid], v1 = hist[2 * tid + 1];
            unsigned incl = v0 + v1;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned up = __shfl_up(incl, off, 64);
                if ((tid & 63) >= off) incl += up;
            }
            if ((tid & 63) == 63) wsum[tid >> 6] = incl;
            __syncthreads();
            unsigned base = 0;
#pragma unroll
            for (int w = 0; w < kOrdThreads / 64; ++w) base += (w < (tid >> 6)) ? wsum[w] : 0u;
            const unsigned e0 = base + incl - (v0 + v1), e1 = e0 + v0;
            hist[2 * tid] = e0;
            hist[2 * tid + 1] = e1;
            if (2 * tid < o.nb) ws_off[(size_t)(2 * tid) * off_stride + t] = (unsigned short)e0;
            if (2 * tid + 1 < o.nb) ws_off[(size_t)(2 * tid + 1) * off_stride + t] = (unsigned short)e1;
            if (tid == 0) ws_off[(size_t)o.nb * off_stride + t] = (unsigned short)kOrdTile;
        }
        __syncthreads();
        unsigned* sp = ws_sp2 + t * (kOrdTile / 2);
#pragma unroll
        for (int u = 0; u < V; ++u) {
            const unsigned p0 = hist[bin[2 * u]] + rank[2 * u], p1 = hist[bin[2 * u + 1]] + rank[2 * u + 1];
            d2 a, b;
            a.x = vx[u].x; a.y = vy[u].x;
            b.x = vx[u].y; b.y = vy[u].y;
            sxy[p0] = a;
            sxy[p1] = b;
            sp[tid + u * kOrdThreads] = p0 | (p1 << 16);       // the query pair at vector (tid + u*512) of the tile
        }
        __syncthreads();
        d2* out = ws_xy + t * kOrdTile;
#pragma unroll
        for (int u = 0; u < kOrdTile / kOrdThreads; ++u) out[tid + u * kOrdThreads] = sxy[tid + u * kOrdThreads];
        __syncthreads();   // the next tile reuses sxy and hist
    }
}

// ---- pass 2 ------------------------------------------------------------------------------------------------
// value of the neighbouring lane of the pair (lane ^ 1): DPP quad_perm [1,0,3,2], no LDS traffic
__device__ __forceinline__ unsigned swap_pair32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ unsigned long long swap_pair64(unsigned long long v)
{
    return ((unsigned long long)swap_pair32((unsigned)(v >> 32)) << 32) | swap_pair32((unsigned)v);
}
__device__ __forceinline__ double swap_pair_f64(double v)
{
    return __longlong_as_double((long long)swap_pair64((unsigned long long)__double_as_longlong(v)));
}

// Work items are (block, group of kOrdSegs tiles).  Every XCD has its own queue -- the blocks b with b % 8 == its id, block
// after block, all groups of a block before the next -- and a workgroup pulls from the queue of the XCD it RUNS on
// (HW_REG_XCC_ID), one atomic per item.  So the workgroups of an XCD are always within one block of each other (its L2 holds
// one or two 2-MiB blocks, whatever the dispatcher did with the grid: the first version assigned blocks by blockIdx % 8 and
// by position in the grid, the workgroups drifted apart and 45 % of the cell reads missed L2), and a workgroup whose own
// queue is empty helps the others out (stolen items miss that XCD's L2 but nothing is left undone).  Correctness does not
// depend on where a workgroup runs or in which order items are taken.
__device__ __forceinline__ unsigned xcc_id()
{
    return (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;   // hwreg(HW_REG_XCC_ID, 0, 4)
}

template <bool IMPL>
__global__ __launch_bounds__(kOrdEvalThreads) void interp2_blocks_kernel(G2Dev g, OrdGeom o, size_t ntiles,
                                                                         const d2* __restrict__ ws_xy,
                                                                         double* __restrict__ ws_z,
                                                                         const unsigned short* __restrict__ ws_off,
                                                                         size_t off_stride, double extrap,
                                                                         unsigned* __restrict__ queues)
{
    __shared__ unsigned pre[kOrdSegs + 1];   // prefix of the segment lengths of this group of tiles
    __shared__ unsigned beg[kOrdSegs];       // first sorted position of the block in each tile
    __shared__ unsigned item_s;
    const int tid = threadIdx.x;
    const unsigned home = xcc_id();
    const unsigned ngroups = (unsigned)((ntiles + kOrdSegs - 1) / kOrdSegs);
    for (unsigned dq = 0; dq < 8; ++dq) {
        const unsigned qd = (home + dq) & 7u;
        const unsigned nbq = (unsigned)o.nb > qd ? ((unsigned)o.nb - qd + 7u) / 8u : 0u;
        const unsigned total = nbq * ngroups;
        for (;;) {
            if (tid == 0) item_s = atomicAdd(&queues[qd], 1u);
            __syncthreads();
            const unsigned item = item_s;
            if (item >= total) break;            // (uniform: every lane read the same word)
            const int b = (int)(qd + 8u * (item / ngroups));
            const size_t grp = item % ngroups;
            const size_t s0 = grp * kOrdSegs;
            if (tid < kOrdSegs) {
                const size_t s = s0 + tid;
                unsigned a = 0, e = 0;
                if (s < ntiles) {
                    a = ws_off[(size_t)b * off_stride + s];
                    e = ws_off[(size_t)(b + 1) * off_stride + s];
                }
                unsigned incl = e - a;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned up = __shfl_up(incl, off, 64);
                    if (tid >= off) incl += up;
                }
                pre[tid + 1] = incl;
                beg[tid] = a;
                if (tid == 0) pre[0] = 0;
            }
            __syncthreads();
            const unsigned M = pre[kOrdSegs];
            for (unsigned j0 = 0; j0 < M; j0 += kOrdEvalThreads * kOrdEvalUnroll) {
                unsigned idx[kOrdEvalUnroll];
                d2 q[kOrdEvalUnroll];
                bool on[kOrdEvalUnroll];
                const unsigned tile0 = (unsigned)s0 * kOrdTile;   // the workspace holds < 2^32 queries
                // lane l of pass u takes sorted query j0 + l + 256 u: neighbouring lanes read neighbouring 16-B pairs (a
                // lane taking four CONSECUTIVE queries instead was measured 2.3x slower: four times the vector-memory
                // transactions and divergent segment walks)
#pragma unroll
                for (int u = 0; u < kOrdEvalUnroll; ++u) {
                    const unsigned j = j0 + tid + u * kOrdEvalThreads;
                    on[u] = j < M;
                    int seg = 0;                                  // largest seg with pre[seg] <= j
#pragma unroll
                    for (int step = kOrdSegs / 2; step > 0; step >>= 1) seg += (pre[seg + step] <= j) ? step : 0;
                    idx[u] = tile0 + (unsigned)seg * kOrdTile + beg[seg] + (j - pre[seg]);
                    q[u].x = g.ax.xmin;                           // idle lanes evaluate the grid's origin (any valid cell)
                    q[u].y = g.ay.xmin;
                    if (on[u]) q[u] = __builtin_nontemporal_load(ws_xy + idx[u]);   // streams must not displace the block in L2
                }
                // The 32 bytes of a cell are two 16-B loads; issued by ONE lane they are two L2 requests (two
                // instructions), and the block pass is bound by the L2 request rate (every cell read is an L2 hit).
                // So the two lanes of a pair split them: in the first instruction both read the EVEN lane's cell (its two
                // halves: adjacent addresses in one instruction = one 64-B request), in the second the odd lane's; each
                // lane then takes the half it is missing from its neighbour (quad_perm swap, no LDS).
                Loc2 L[kOrdEvalUnroll];
                d2 ra[kOrdEvalUnroll], rb[kOrdEvalUnroll];
                const unsigned odd = tid & 1u;
#pragma unroll
                for (int u = 0; u < kOrdEvalUnroll; ++u) {
                    L[u] = locate2<IMPL>(g, q[u].x, q[u].y);
                    const unsigned long long mine = reinterpret_cast<unsigned long long>(L[u].cell);
                    const unsigned long long other = swap_pair64(mine);
                    const d2* pe = reinterpret_cast<const d2*>(odd ? other : mine);   // the even lane's cell
                    const d2* po = reinterpret_cast<const d2*>(odd ? mine : other);   // the odd lane's cell
                    ra[u] = pe[odd];
                    rb[u] = po[odd];
                }
#pragma unroll
                for (int u = 0; u < kOrdEvalUnroll; ++u) {
                    // even lane: has lo(own) in ra, lo(odd's) in rb -> sends rb, receives hi(own) = odd's ra
                    // odd lane:  has hi(even's) in ra, hi(own) in rb -> sends ra, receives lo(own) = even's rb
                    d2 send = odd ? ra[u] : rb[u], recv;
                    recv.x = swap_pair_f64(send.x);
                    recv.y = swap_pair_f64(send.y);
                    const d2 lo = odd ? recv : ra[u], hi = odd ? rb[u] : recv;
                    const double r = blend2<IMPL>(g, L[u], lo, hi, q[u].x, q[u].y, extrap);
                    if (on[u]) __builtin_nontemporal_store(r, ws_z + idx[u]);
                }
            }
            __syncthreads();   // pre / beg / item_s are rewritten for the next item
        }
        __syncthreads();       // every lane has read item_s before the next queue's first fetch overwrites it
    }
}

// ---- pass 3 ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kOrdEvalThreads) void interp2_unsort_kernel(const double* __restrict__ ws_z,
                                                                         const unsigned* __restrict__ ws_sp2,
                                                                         double* __restrict__ zq)
{
    __shared__ double sz[kOrdTile];
    const int tid = threadIdx.x;
    const size_t t = blockIdx.x;
    const d2* in = reinterpret_cast<const d2*>(ws_z + t * kOrdTile);
    constexpr int V = kOrdTile / 2 / kOrdEvalThreads;   // 8 vectors per lane
#pragma unroll
    for (int u = 0; u < V; ++u) {
        const d2 v = __builtin_nontemporal_load(in + tid + u * kOrdEvalThreads);
        reinterpret_cast<d2*>(sz)[tid + u * kOrdEvalThreads] = v;
    }
    __syncthreads();
    const unsigned* sp = ws_sp2 + t * (kOrdTile / 2);
    d2* out = reinterpret_cast<d2*>(zq + t * kOrdTile);
#pragma unroll
    for (int u = 0; u < V; ++u) {
        const unsigned pp = __builtin_nontemporal_load(sp + tid + u * kOrdEvalThreads);
        d2 r;
        r.x = sz[pp & 0xffffu];
        r.y = sz[pp >> 16];
        __builtin_nontemporal_store(r, out + tid + u * kOrdEvalThreads);
    }
}

}  // namespace mi_interp2
