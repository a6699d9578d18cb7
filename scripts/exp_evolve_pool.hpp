// NOT PRODUCT CODE.  The pooled-Newton form of the evolve kernel, built and measured in round 3, bit-identical to
// evolve_kernel on every stage tap (it ran in the EDM GPU tests and the pooled / unpooled A-B test) and SLOWER:
//   EXACT  N = 1024, R = 125 000: 163.3 -> 171.2 ms     N = 512: 55.7 -> 57.2 ms
//   FAST   N = 1024, R = 125 000:  93.3 -> 107.0 ms     N = 512: 27.8 -> 33.5 ms       (profiles/r03_edm_pooled_newton_timing.log)
// The Newton rounds are a quarter of a wave's time, but the kernel is bound by dependent latencies (59 % of the VALU issue
// peak), not by instruction count: running the Newton instruction stream once per workgroup instead of four times frees
// issue slots nobody was short of, and the two workgroup barriers per event make every wave wait for the slowest of four.
// To re-measure: paste the kernel into csrc/mi_edm.hip before evolve_wg_kernel and launch it from launch_evolve with
// LDS = the evolve_kernel's + kPoolFloats * 4 bytes, grid = min((R + 3) / 4, CUs * workgroups per CU).
#if 0
// ---- EvolveKernel, homogeneous model, Newton solves POOLED over the workgroup (round 3) ----------------------
// evolve_kernel spends a quarter of its time in the Newton rounds with five to ten of 64 lanes active: every realisation
// has only a handful of neurons about to fire.  Here the four waves of a workgroup (four realisations) put the
// candidates of an event -- (v, s) of every firing neuron -- into a pool in LDS, ONE wave (a different one every event)
// solves them all, each in a lane of its own, and the owners read their firing times back: the Newton instruction stream
// runs once per workgroup and event instead of four times, with four times the lanes busy.  Two workgroup barriers per
// event; a wave at the barrier leaves its SIMD to the other workgroups' waves.  newton_time() is a pure function of
// (v, s, beta), so WHICH lane evaluates it cannot change a bit: every stage tap stays bit-identical to evolve_kernel
// (tests/test_edm_gpu.py::test_pooled_newton_form_changes_nothing) and to the oracle.  Homogeneous model only (with a
// per-neuron beta the realisations of a workgroup need different numbers of events; the waves here run in step, the
// finished ones keep the barriers company).
constexpr unsigned kPoolFloats = 3u * 4u * 64u + 16u;   // candidates' v, s, the firing times; two copies of 4 counts + 4 flags

template <int MATH, int NS, bool UDIV>
__global__ __launch_bounds__(kEvolveBlock) void evolve_pool_kernel(edm::Model M, SpikeSeeds sd,
                                                                   const float* __restrict__ v0,
                                                                   const float* __restrict__ s0,
                                                                   const float* __restrict__ w,
                                                                   float* __restrict__ g_t0,
                                                                   unsigned short* __restrict__ g_i0,
                                                                   float* __restrict__ g_t1,
                                                                   unsigned short* __restrict__ g_i1,
                                                                   unsigned* __restrict__ g_accept)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned npl = (M.N + 63u) / 64u;
    const unsigned slots = npl * 64u;
    float* w_lds = lds;
    for (unsigned i = threadIdx.x; i < (unsigned)kMaxGrid; i += kEvolveBlock) {
        const float wi = (i < M.N) ? w[i] : 0.0f;
        w_lds[i] = M.beta_mean * wi;          // RN(beta * w[d]), as in evolve_kernel
    }
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* V = lds + kMaxGrid + (size_t)wave * 2u * slots;
    float* S = V + slots;
    float* cand_v = lds + kMaxGrid + 4u * 2u * slots;
    float* cand_s = cand_v + 256;
    float* tau_s = cand_s + 256;
    unsigned* meta = reinterpret_cast<unsigned*>(tau_s + 256);   // [parity][0..3 counts | 4..7 flags: bit 0 alive, bit 1 more pending]
    const unsigned full = (1u << M.S) - 1u;
    const float two_T = 2.0f * M.T;
    const float beta = M.beta_mean;
    unsigned par = 0;                         // parity of the meta copy: toggles at every collection (workgroup-uniform)

    for (unsigned rb = blockIdx.x * 4u; rb < M.R; rb += gridDim.x * 4u) {
        const unsigned r = rb + wave;
        const bool real = r < M.R;            // the last group of four may be short: its idle waves only keep the barriers
        unsigned skip = 0, nan_i = ~0u, valid = 0;
        for (unsigned k = 0; k < npl; ++k) {
            const unsigned i = k * 64u + lane;
            const bool act = real && i < M.N;
            const float vi = act ? v0[i] : 0.0f, si = act ? s0[i] : 0.0f;
            V[i] = vi;
            S[i] = si;
            if (__all(!act || (vi != vi && si != si)) && __any(act)) {
                skip |= 1u << k;
                if (act && nan_i == ~0u) nan_i = i;
            } else if (act) {
                valid |= 1u << k;
            }
        }
        float lt[NS], ct[NS];
        unsigned li[NS], ci[NS];
#pragma unroll
        for (int m = 0; m < NS; ++m) {
            lt[m] = 0.0f;
            ct[m] = 0.0f;
            ci[m] = 0u;
            li[m] = (m < (int)M.S) ? (unsigned)sd.ind[m] : 0u;
        }
        unsigned crossed = 0;
        float now = 0.0f;
        float base_t = INFINITY;
        unsigned base_i = 0;
        unsigned pend = 0;
        for (unsigned k = 0; k < npl; ++k) {
            if ((skip >> k) & 1u) continue;
            const unsigned i = k * 64u + lane;
            if (real && i < M.N) {
                if (edm::will_fire<MATH, UDIV>(M, V[i], S[i], beta)) pend |= (1u << k);
            }
        }
        auto lowest_quiet = [&]() {
            const unsigned quiet = ~pend & valid;
            base_t = INFINITY;
            base_i = 0;
            if (quiet != 0u) { base_t = edm::kNever; base_i = (unsigned)__builtin_ctz(quiet) * 64u + lane; }
            if (nan_i != ~0u && (base_t == INFINITY || nan_i < base_i)) { base_t = edm::kNever; base_i = nan_i; }
        };
        lowest_quiet();
        unsigned events = 0;
        for (unsigned ev = 0;; ++ev) {        // ev: events of the WORKGROUP (the same in its four waves)
            const bool alive = real && crossed < full && now < two_T && events < M.max_events;
            if (alive) ++events;
            float best = base_t;
            unsigned idx = base_i;
            bool any_alive = false;
            for (unsigned round = 0;; ++round) {
                // collect: every lane's lowest pending neuron goes into this wave's part of the pool
                const bool has = alive && pend != 0u;
                const unsigned long long mask = __ballot(has);
                unsigned my_pos = 0, my_i = 0;
                if (has) {
                    const unsigned k = (unsigned)__builtin_ctz(pend);
                    pend &= pend - 1u;
                    my_i = k * 64u + lane;
                    my_pos = (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
                    cand_v[wave * 64u + my_pos] = V[my_i];
                    cand_s[wave * 64u + my_pos] = S[my_i];
                }
                const bool more = __any(alive && pend != 0u) != 0;
                unsigned* mt = meta + par * 8u;
                par ^= 1u;
                if (lane == 0) {
                    mt[wave] = (unsigned)__popcll(mask);
                    mt[4u + wave] = (alive ? 1u : 0u) | (more ? 2u : 0u);
                }
                __syncthreads();
                const unsigned c0 = mt[0], c1 = mt[1], c2 = mt[2], c3 = mt[3];
                const unsigned fl = mt[4] | mt[5] | mt[6] | mt[7];
                if (round == 0) any_alive = (fl & 1u) != 0u;
                const unsigned total = c0 + c1 + c2 + c3;
                if (total != 0u) {            // (the same in every wave of the workgroup)
                    // solve: candidate `slot` of the pooled list; the wave that takes the first 64 changes every event
                    const unsigned slot = ((wave + 4u - (ev & 3u)) & 3u) * 64u + lane;
                    if (slot < total) {
                        const unsigned wq = (slot >= c0 ? 1u : 0u) + (slot >= c0 + c1 ? 1u : 0u) + (slot >= c0 + c1 + c2 ? 1u : 0u);
                        const unsigned cq = slot - (wq > 0u ? c0 : 0u) - (wq > 1u ? c1 : 0u) - (wq > 2u ? c2 : 0u);
                        tau_s[wq * 64u + cq] = edm::newton_time<MATH, UDIV>(M, cand_v[wq * 64u + cq], cand_s[wq * 64u + cq], beta);
                    }
                    __syncthreads();
                    if (has) {
                        const float tau = tau_s[wave * 64u + my_pos];
                        if (tau < best || (tau == best && my_i < idx)) { best = tau; idx = my_i; }
                    }
                }
                if (!(fl & 2u)) break;        // no wave has a second pending neuron in any lane
            }
            if (!any_alive) break;            // every realisation of the workgroup is through
            if (!alive) continue;             // this one is; the others are not: keep them company at the barriers
            wave_argmin(best, idx);
            idx = (unsigned)__builtin_amdgcn_readfirstlane((int)idx);
            const float dt = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, best)));
            // the three wave-uniform exponentials of the advance in one pass (lanes 0, 1, 2), as in evolve_kernel
            const float arg = (lane == 1u) ? (1.0f - beta) * dt : (lane == 2u) ? -beta * dt : -dt;
            const float ex = edm::expf_<MATH>(arg);
            const float e1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ex), 0));
            const float e2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ex), 1));
            const float e3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ex), 2));
            for (unsigned k = 0; k < npl; ++k) {
                if ((skip >> k) & 1u) continue;
                const unsigned i = k * 64u + lane;
                const float so = S[i];
                float vv = V[i] * e1;
                vv = vv + (M.I * (1.0f - e1) + edm::div_by<MATH, UDIV>(so * e1, 1.0f - beta) * (e2 - 1.0f));
                vv = (i == idx) ? vv * 0.0f : vv;
                float sn = so * e3;
                const unsigned dist = (unsigned)abs((int)i - (int)idx);
                sn = sn + w_lds[dist];
                V[i] = vv;
                S[i] = sn;
                if (edm::will_fire<MATH, UDIV>(M, vv, sn, beta)) pend |= (1u << k);
            }
            pend &= valid;
            if (events == 1u) {               // the all-NaN slice mask is final after the first update (see evolve_kernel)
                for (unsigned k = 0; k < npl; ++k) {
                    if ((skip >> k) & 1u) continue;
                    const unsigned i = k * 64u + lane;
                    const bool act = i < M.N;
                    const float vi = V[i], si = S[i];
                    if (__all(!act || (vi != vi && si != si)) && __any(act)) {
                        skip |= 1u << k;
                        valid &= ~(1u << k);
                        if (act && i < nan_i) nan_i = i;
                    }
                }
                pend &= valid;
            }
            lowest_quiet();
            now = now + dt;
            unsigned mi = 0;
#pragma unroll
            for (int m = 1; m < NS; ++m) {
                if (m < (int)M.S) {
                    unsigned lmi = li[0];
#pragma unroll
                    for (int j = 1; j < NS; ++j) lmi = (mi == (unsigned)j) ? li[j] : lmi;
                    const int dm = abs((int)idx - (int)li[m]);
                    const int d0 = abs((int)idx - (int)lmi);
                    mi += (dm < d0) ? 1u : 0u;
                }
            }
            if (!(crossed & (1u << mi))) {
                const bool after = now > M.T;
#pragma unroll
                for (int m = 0; m < NS; ++m) {
                    if (mi == (unsigned)m) {
                        if (after) { ct[m] = now; ci[m] = idx; }
                        else { lt[m] = now; li[m] = idx; }
                    }
                }
                if (after) crossed += (1u << mi);
            }
        }
        if (real) {
#pragma unroll
            for (int m = 0; m < NS; ++m) {
                if (lane == (unsigned)m && m < (int)M.S) {
                    const size_t k = (size_t)m * M.R + r;
                    g_t0[k] = lt[m];
                    g_i0[k] = (unsigned short)li[m];
                    g_t1[k] = ct[m];
                    g_i1[k] = (unsigned short)ci[m];
                }
            }
            if (lane == 0) g_accept[r] = (crossed == full) ? 1u : 0u;
        }
    }
}

#endif
