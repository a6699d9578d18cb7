// NOT PRODUCT CODE.  The two eval_batch_from variants that were measured inside csrc/mi_interp1.hip in rounds 1-2 and
// found slower; moved here in round 3 so that the product translation unit carries no build-time experiment switches.
// To re-measure: paste the block into the MODE == 3 branch of csrc/mi_interp1_eval.hpp (it uses that function's locals).
//
// (a) MI_INTERP1_COOP -- wavefront reuse of shared abscissae through __shfl (north_star's suggestion; ordered queries,
//     streaming kernel): 0.304 -> 0.366 ms per 1e8 sorted queries on the jittered 1e6-node grid, 21 % slower.
// (b) MI_M3_TWO_LOOKUPS -- node G first, then ONE dependent gather of G-1 or G+1: 1.08 ms against 1.03 ms for the shipped
//     "G and G+1 eager, G-1 on demand" (profiles/r02_mode3_gather_variants.log).
#if 0
#if MI_INTERP1_COOP
        // Wavefront reuse of shared abscissae (ordered queries, streaming kernel): when every lane's guess lies within
        // +-30 nodes of lane 0's, the wave loads one 64-node window with a single coalesced gather (lane j: node
        // base + j) and each lane picks its nodes out of the other lanes' registers with __shfl (ds_bpermute) --
        // 1 + 8 cross-lane reads instead of three 16-B gathers per query.  Otherwise: the three gathers.
        // Measured (scripts/gpu_interp_timing.py, -DMI_INTERP1_COOP=1): 21 % SLOWER -- lanes that share a line are
        // already merged by the texture path, and eight ds_bpermute plus their waits cost more than what is left.
        bool coop[NQ];
        if constexpr (WIN && !LDSY) {
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const int ref = __builtin_amdgcn_readfirstlane(l[k]);
                coop[k] = __ballot(1) == ~0ull && __all(abs(l[k] - ref) <= 30) != 0;   // the window needs all 64 lanes
                if (coop[k]) {
                    const int base = ref - 31;                                   // window [ref-31, ref+32]
                    const int lane = (int)(threadIdx.x & 63u);
                    const d2 mine = load_node<false>(ytab, min(max(base + lane, 0), g.n));
                    const int r = l[k] - base;                                   // 1 .. 61
                    n0[k].x = __shfl(mine.x, r, 64);
                    n0[k].y = __shfl(mine.y, r, 64);
                    const bool down = qs[k] < n0[k].x;
                    const int ro = down ? r - 1 : r + 1;
                    const double ox = __shfl(mine.x, ro, 64), oy = __shfl(mine.y, ro, 64);
                    nm[k].x = ox; nm[k].y = oy;
                    n1[k].x = ox; n1[k].y = oy;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NQ; ++k) coop[k] = false;
        }
#else
        bool coop[NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) coop[k] = false;
#endif
// ---- (b)
#if MI_M3_TWO_LOOKUPS
#pragma unroll
            for (int k = 0; k < NQ; ++k) n0[k] = load_node<LDSY>(ytab, l[k]);
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const bool down = qs[k] < n0[k].x;
                n1[k] = load_node<LDSY>(ytab, down ? max(l[k] - 1, 0) : l[k] + 1);
            }
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const bool down = qs[k] < n0[k].x;
                const d2 a = down ? n1[k] : n0[k], b = down ? n0[k] : n1[k];
                out[k] = blend(a.x, a.y, b.x, b.y, qs[k]);
            }
#else
#endif
