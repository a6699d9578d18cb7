// EventDrivenMap residual on MI355X: lift -> evolve -> restrict -> average
// (replaces EventDrivenMap.cu:57-404, 505-674 of the reference).
//
// MI355X-first structure (not the reference's launch shapes):
//   * Lift (EventDrivenMap.cu:505-542) does not depend on the realisation, so
//     it runs ONCE per ComputeF on N lanes (the reference runs it R times and
//     writes 8*R*N bytes); Evolve reads the N-point profile from L2.
//   * Evolve: ONE WAVE64 PER REALISATION, each lane owning N/64 neurons in
//     lane-private LDS slots (neuron i = k*64 + lane).  The reference uses one 1024-thread
//     block per realisation with two __syncthreads() and a shared-memory
//     arg-min per event (EventDrivenMap.cu:601-645, 855-881); here the
//     arg-min is 6 wave shuffles, there is no barrier in the event loop, and
//     the coupling table sits in LDS.  beta is generated on the fly from a
//     counter-based hash (no 4*R*N-byte beta array, no RNG library).
//   * Restrict + masked mean: the fused single-pass kernel of mi_restrict.hip.
//   * U and the seed indices travel as kernel arguments: no H2D copies; one
//     D2H copy of S+1 scalars per ComputeF (the reference: 3 small copies).
// Arithmetic: csrc/mi_edm_math.hpp == oracle/edm_oracle.c (bit-exact in
// MI_EDM_MATH_EXACT mode).  Documented decisions [D1]-[D7]: DESIGN.md.
#include <algorithm>
#include <cmath>
#include <new>
#include <type_traits>
#include <cstdlib>
#include <vector>

#include "mi_common.hpp"
#include "mi_edm_math.hpp"

namespace {

constexpr int kMaxSpikes = 8;
constexpr int kMaxGrid = 1024;
constexpr int kEvolveBlock = 256;   // 4 waves = 4 realisations per workgroup
// Hard bound on events per realisation (mi_edm_params.max_events, default 2^20, at most 2^24): the reference's
// loop (EventDrivenMap.cu:601) relies on time advancing, which degenerate parameters defeat (a strongly excitatory
// kernel fires at ever shorter intervals).  Every wave must reach an exit, so the loop also stops after
// max_events events (the realisation is then simply not accepted); the oracle applies the same rule.
constexpr unsigned kMaxEventsLimit = 1u << 24;
// Evolve's unguarded quotient (evolve_kernel, state_pass) wants |s| < 2^101 throughout.  With beta > 0 every event maps
// s -> s exp(-beta dt) + beta w with dt >= 0, so |s| grows by at most max |beta w| per event: if the lift profile and the
// coupling table stay below kBigS = 2^60, |s| < 2^60 (1 + 2^24) < 2^101 after any admissible number of events.  The lift
// kernel reports the first condition in bit 31 of its live-slice word, the host checks the second, and the launch hands
// the conjunction to the kernel in bit 31 of `store`.
constexpr float kBigS = 0x1.0p+60f;
constexpr unsigned kBigFlag = 0x80000000u, kSliceMask = 0xffffu;

struct SpikeSeeds {
    float U[kMaxSpikes + 1];        // (c, 0, Z1, ..), fp32 (EventDrivenMap.cu:172)
    unsigned short ind[kMaxSpikes]; // initialSpikeInd, EventDrivenMap.cu:361-372
};

// ---- LiftKernel (EventDrivenMap.cu:505-542), once per ComputeF -------------
// one grid point; returns the synaptic variable it stored
template <int MATH>
__device__ __forceinline__ float lift_point(const edm::Model& M, const SpikeSeeds& sd, unsigned i, float* __restrict__ v,
                                            float* __restrict__ s, bool store)
{
    const float c = sd.U[0], beta = M.beta_mean;
    const float a[2] = {M.a1, M.a2}, b[2] = {M.b1, M.b2};
    const float h = (2.0f * M.L) / (float)M.N;
    const float omb = 1.0f - beta;
    const float x = M.L - h * (float)i;
    const float xc = x / c;
    float sv = 0.0f, ss = 0.0f;
    for (unsigned m = 1; m <= M.S; ++m) {
        // every product, quotient and sum below is grouped as C groups the reference's expression (:522-534):
        // left to right, `-b1*c*U` = ((-b1)*c)*U, `a1*beta*c/(..)` = ((a1*beta)*c)/(..)  (== oracle/edm_oracle.c)
        const float Um = sd.U[m];
        const float cu = c * Um;
        const float d = x - cu;
        const float pos = (d > 0.0f) ? 1.0f : 0.0f;
        const float neg = (d <= 0.0f) ? 1.0f : 0.0f;
        const float ebu = edm::expf_<MATH>(beta * Um);
        const float exo = edm::expf_<MATH>(xc * omb);
        const float dxk[2] = {exo - edm::expf_<MATH>((cu / c) * omb),    // a1 term, :523: exp(((c*U)/c)*(1-beta))
                              exo - edm::expf_<MATH>(Um * omb)};         // a2 term, :524: exp((U)*(1-beta))
        const float ebc = edm::expf_<MATH>(-(beta / c) * d);
        float P[2], Q[2], R[2], B[2], sA[2], sB1[2], sB2[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float cb = c * b[k];
            const float abc = (a[k] * beta) * c;
            const float ep = (1.0f + cb) / c;
            const float em = (1.0f - cb) / c;
            const float Pk = abc / ((beta + cb) * (1.0f + cb));
            const float en = edm::expf_<MATH>(-(cb * Um));               // exp(-b*c*U) = exp(((-b)*c)*U)
            P[k] = (Pk * edm::expf_<MATH>(cu * ep)) * en;
            Q[k] = (((abc / omb) * ebu) * (1.0f / (beta + cb) + 1.0f / (cb - beta))) * dxk[k];
            R[k] = ((abc / ((cb - beta) * (1.0f - cb))) * edm::expf_<MATH>(cb * Um)) * (edm::expf_<MATH>(x * em) - edm::expf_<MATH>(cu * em));
            B[k] = (Pk * edm::expf_<MATH>(x * ep)) * en;
            // synaptic profile, :532-534
            sA[k] = ((beta * a[k]) * (c / (beta + cb))) * edm::expf_<MATH>(b[k] * d);
            sB1[k] = (((2.0f * a[k]) / b[k]) * (beta / (1.0f - (beta * beta) / (((c * c) * b[k]) * b[k])))) * ebc;
            sB2[k] = ((beta * a[k]) * (c / (cb - beta))) * edm::expf_<MATH>(b[k] * (cu - x));
        }
        // :522-527: P1 - P2 + Q1 - R1 - Q2 + R2, added left to right
        const float brA = ((((P[0] - P[1]) + Q[0]) - R[0]) - Q[1]) + R[1];
        const float brB = B[0] - B[1];
        const float dummyV = (pos * brA + neg * brB) * edm::expf_<MATH>(-xc);
        sv = sv + ((dummyV - pos * edm::expf_<MATH>(-d / c)) + neg * 0.0f);             // :530
        // (cu - x) > 0  <=>  d < 0 ; (cu - x) <= 0  <=>  d >= 0
        const float e = cu - x;
        const float posS = (e > 0.0f) ? 1.0f : 0.0f;
        const float negS = (e <= 0.0f) ? 1.0f : 0.0f;
        ss = ss + (posS * (sA[0] - sA[1]) + negS * (((sB1[0] - sB2[0]) - sB1[1]) + sB2[1]));
    }
    float vv = M.I + sv;
    vv = vv * ((vv < 1.0f) ? 1.0f : 0.0f);
    if (store) {
        v[i] = vv;
        s[i] = ss;
    }
    return ss;
}

// aux[0] <- bit k set iff the 64-neuron slice k holds a neuron whose synaptic variable is not NaN (the slices Evolve
// has to carry, see evolve_kernel); bit 31 set iff some synaptic variable is infinite or >= 2^60 in magnitude (kBigS)
template <int MATH>
__global__ __launch_bounds__(kMaxGrid) void lift_kernel(edm::Model M, SpikeSeeds sd, float* __restrict__ v,
                                                        float* __restrict__ s, unsigned* __restrict__ aux)
{
    __shared__ unsigned live;
    const unsigned i = threadIdx.x;
    if (i == 0) live = 0u;
    __syncthreads();
    const unsigned ic = (i < M.N) ? i : M.N - 1u;      // lanes past the grid recompute the last point and store nothing
    const float ss_ = lift_point<MATH>(M, sd, ic, v, s, i < M.N);
    if (__any(i < M.N && ss_ == ss_) && (i & 63u) == 0u) atomicOr(&live, 1u << (i >> 6));
    if (__any(i < M.N && ss_ == ss_ && !(fabsf(ss_) < kBigS)) && (i & 63u) == 0u) atomicOr(&live, kBigFlag);
    __syncthreads();
    if (i == 0) aux[0] = live;
}

// Wave64 unsigned minimum with DPP row shifts + row broadcasts (gfx9 family), result broadcast to every lane.
// No LDS traffic and no s_waitcnt, unlike __shfl_xor (ds_bpermute_b32).
__device__ __forceinline__ unsigned wave_umin(unsigned v)
{
#define MI_DPP_MIN(ctrl, row_mask) \
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, ctrl, row_mask, 0xf, false))
    MI_DPP_MIN(0x111, 0xf);   // row_shr:1
    MI_DPP_MIN(0x112, 0xf);   // row_shr:2
    MI_DPP_MIN(0x114, 0xf);   // row_shr:4
    MI_DPP_MIN(0x118, 0xf);   // row_shr:8   -> lane 15 of every row holds its row's minimum
    MI_DPP_MIN(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    MI_DPP_MIN(0x143, 0xc);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave minimum
#undef MI_DPP_MIN
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Wave64 unsigned maximum, same DPP pattern (identity 0).
__device__ __forceinline__ unsigned wave_umax(unsigned v)
{
#define MI_DPP_MAX(ctrl, row_mask) \
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, row_mask, 0xf, false))
    MI_DPP_MAX(0x111, 0xf);
    MI_DPP_MAX(0x112, 0xf);
    MI_DPP_MAX(0x114, 0xf);
    MI_DPP_MAX(0x118, 0xf);
    MI_DPP_MAX(0x142, 0xa);
    MI_DPP_MAX(0x143, 0xc);
#undef MI_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// ---- the reference's block arg-min, ties included (blockReduceMin / warpReduceMin, EventDrivenMap.cu:843-881) ----
// On 32-wide warps the reference's shuffle trees keep the lower lane's pair only when its time is STRICTLY smaller, so a
// tie goes to the higher lane of each step: among neurons of equal minimal time the winner is the one with the largest
//     key(i) = rev5(i >> 5) * 32 + rev5(i & 31)              (rev5 = the 5-bit number read backwards)
// and, in a block of fewer than 32 warps, the padding pairs (100.0f, 0) of :867-868 win whenever no time is below 100.0f
// (result: time 100.0f, index 0).  oracle/edm_oracle.c derives this (orc_edm_argmin) and checks it against a literal
// emulation of the trees.  `tree` = the block is made of whole warps (N a multiple of 32 -- the reference's 1024 and 512);
// otherwise ([D1], a shape the reference cannot run) ties go to the lowest index: key(i) = 1023 - i.  Neuron indices are
// < 1024.  A NaN time never wins ([D1]).
__device__ __forceinline__ unsigned rev5(unsigned x) { return __brev(x) >> 27; }
__device__ __forceinline__ unsigned tie_key(unsigned i, bool tree)
{
    return tree ? rev5(i >> 5) * 32u + rev5(i & 31u) : 1023u - i;
}
__device__ __forceinline__ unsigned tie_key_neuron(unsigned key, bool tree) { return tie_key(key, tree); }   // the map is an involution
__device__ __forceinline__ unsigned rev4(unsigned x) { return __brev(x) >> 28; }

// Minimum over the wave of (time, then LARGEST key).  Times are >= 0 (|t|, 100 or +inf, never NaN), so their bit
// patterns order like unsigned integers: two unsigned reductions give exactly the pair a comparison tree would.
__device__ __forceinline__ void wave_argmin(float& best, unsigned& key)
{
    const unsigned tb = __float_as_uint(best);
    const unsigned tmin = wave_umin(tb);
    const unsigned long long at_min = __ballot(tb == tmin);
    if (__builtin_popcountll(at_min) == 1)      // one lane holds the minimum (a firing time: every event in practice): its key, no second reduction
        key = (unsigned)__builtin_amdgcn_readlane((int)key, (int)__builtin_ctzll(at_min));
    else
        key = wave_umax(tb == tmin ? key + 1u : 0u) - 1u;
    best = __uint_as_float(tmin);
}

// ---- EvolveKernel (EventDrivenMap.cu:575-674): one wave64 per realisation ----
// Neuron state lives in LDS, [wave][array][slot*64 + lane]: every lane only ever touches its own slots, so the event
// loop needs no barrier and the per-neuron loop stays rolled (64 VGPRs for up to three bumps: eight waves per SIMD at N = 512.
// amdgpu_waves_per_eu(7, 8) is what gets the scheduler there WITHOUT scratch: left alone it takes 58 registers and a schedule
// that is 4 % slower at N = 512; asked for (8, 8) it spills two registers to scratch, which no product kernel should need).
//
// Dead slices.  A 64-neuron slice whose every neuron starts with a NaN synaptic variable (the stretch that the lift
// profile poisons through 0 * inf, LiftKernel :505-542 -- neurons 820..1023 at the reference's parameters, i.e. slices
// 13-15 of 16) can never matter again: s' = s e3 + beta w stays NaN, v' = v e1 + (.. + (s e1 / (1 - beta)) (e2 - 1)) is NaN
// after the first update, will_fire() is false for a NaN s, so such a neuron never fires, and as the arg-min's
// winner (only when NO neuron fires) only its index is used.  The lift kernel reports the live slices as a bit mask
// (`store`), the host sizes the LDS for those alone -- 13 slices: 4 + 4 * 6.5 KiB = 30 KiB per workgroup, FIVE
// workgroups (20 waves) per CU instead of four -- and the state pass never visits the others; their standing
// contribution to the arg-min, the time kNever with the largest tie key among the lane's dead neurons, is kept in nan_key.
//
// One workgroup per four realisations (launch_evolve): the hardware dispatcher hands them out as slots free up.
// LDS per workgroup: w[1024] + 4 waves * (2 or 3) arrays * popcount(store)*64 floats + 4 * 64 pending-neuron slots
// (evolve_lds_bytes).
// NS: compile-time bound of the per-bump loops (3 = the reference's noSpikes, else kMaxSpikes)
// UDIV: divisions by the homogeneous model's wave-uniform divisors through edm::div_by's exact quotient (three or five
//       operations, see ONE; in the state pass without its range guard when the tracked range of |s| allows it)
// TAPS: the same computation, also counting into taps[kTap*] how often it reaches the documented decisions
//       (mi_edm_debug_counters; never what ComputeF launches)
enum { kTapEvents = 0, kTapMaxEvents, kTapMaxNewton, kTapNewtonCap, kTapEventCap, kTapAccepted, kTapNoFiring, kTapTies, kTapCount };
// TREE: the block is made of whole warps (N a multiple of 32): arg-min ties as the reference breaks them (tie_key)
// GAP: the host has checked 0 < vth - I <= 1 (edm::gap_settles_sign; UDIV implies it)
// ONE: the host has proved that ONE correction step makes the quotient by 1 - beta exact (divisor_check_kernel; UDIV only)
template <int MATH, bool HETERO, int NS, bool UDIV, bool TAPS = false, bool TREE = true, bool GAP = UDIV, bool ONE = false>
__global__ __launch_bounds__(kEvolveBlock) __attribute__((amdgpu_waves_per_eu(NS <= 3 ? 7 : 4, 8))) void evolve_kernel(edm::Model M, SpikeSeeds sd, unsigned store_word,
                                                              unsigned long long* __restrict__ taps,
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
    const unsigned store = store_word & kSliceMask;            // the live slices
    const bool s_bounded = (store_word & kBigFlag) != 0u;      // |s| < 2^101 for the whole evolution (see kBigS)
    const unsigned npl = (M.N + 63u) / 64u;
    const unsigned slots = (unsigned)__builtin_popcount(store) * 64u;
    float* w_lds = lds;
    // homogeneous model: the table holds RN(beta * w[d]), the product every event would otherwise form again
    for (unsigned i = threadIdx.x; i < (unsigned)kMaxGrid; i += kEvolveBlock) {
        const float wi = (i < M.N) ? w[i] : 0.0f;
        w_lds[i] = HETERO ? wi : M.beta_mean * wi;
    }
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr unsigned kArrays = HETERO ? 3u : 2u;
    // per wave: the live slices' (v, s) side by side -- slice j holds v at [128 j, 128 j + 64) and s 64 floats further, so that one
    // ds_read2st64_b32 / ds_write2st64_b32 moves both -- then, for per-neuron beta only, the slices' beta values
    float* V = lds + kMaxGrid + (size_t)wave * kArrays * slots;
    float* S = V + 64;
    float* B = V + 2u * slots;   // only touched when HETERO; neuron at V[a] has its beta at B[bidx(a)]
    auto bidx = [&](unsigned a) { return ((a - lane) >> 1) + lane; };
    unsigned* list = reinterpret_cast<unsigned*>(lds + kMaxGrid + (size_t)(kEvolveBlock / 64) * kArrays * slots) + wave * 64u;   // this wave's pending neurons
    const unsigned full = (1u << M.S) - 1u;
    const float two_T = 2.0f * M.T;
    // arg-min ties as the reference breaks them (tie_key above); blocks that are not whole warps: lowest index
    constexpr bool tree = TREE;
    const bool padded = tree && (M.N >> 5) < 32u;          // the reference's second-stage padding lanes take part
    // Slice masks (pend, valid) are kept in KEY order: slice k sits at bit pos_of(k) = rev4(k) (whole warps) or 15 - k, so
    // that neuron k*64 + lane has the key  lane_base + (pos_of(k) << key_sh)  and the highest set bit of a mask is the
    // slice that wins a tie -- no per-event permutation.  (Check: i >> 5 = 2k + (lane >> 5), rev5 of it is
    // (lane >> 5) * 16 + rev4(k); and 1023 - i = (63 - lane) + 64 (15 - k).)
    constexpr unsigned key_sh = tree ? 5u : 6u;
    const unsigned lane_base = tree ? (lane >> 5) * 512u + rev5(lane & 31u) : 63u - lane;
    auto pos_of = [&](unsigned k) { return tree ? rev4(k) : 15u - k; };   // an involution: also slice-of-position
    // the same for every realisation: which of this lane's neurons exist in a live slice, and the largest tie key among
    // the lane's neurons in dead slices (they all stand at the time kNever)
    unsigned valid = 0, nan_key = ~0u;
    for (unsigned k = 0; k < npl; ++k) {
        const unsigned i = k * 64u + lane;
        if (i < M.N) {
            if ((store >> k) & 1u) valid |= 1u << pos_of(k);
            else {
                const unsigned key = lane_base + (pos_of(k) << key_sh);
                if (nan_key == ~0u || key > nan_key) nan_key = key;
            }
        }
    }

    // one realisation per wave: the grid is one workgroup per four realisations (launch_evolve), nothing loops over them here
    // (and nothing of the set-up above has to survive the event loop)
    const unsigned r = blockIdx.x * (kEvolveBlock / 64) + wave;
    if (r < M.R) {
        {
            unsigned a = lane;
            for (unsigned m = store; m != 0u; m &= m - 1u, a += 128u) {
                const unsigned i = (unsigned)__builtin_ctz(m) * 64u + lane;
                const bool act = i < M.N;
                V[a] = act ? v0[i] : 0.0f;
                S[a] = act ? s0[i] : 0.0f;
                if constexpr (HETERO) B[bidx(a)] = edm::beta_of<MATH>(M.beta_mean, M.beta_sigma, M.seed, M.N, (uint64_t)r + M.real_offset, act ? i : 0u);
            }
        }
        // per-bump event slots ([D2]: start at time 0 / index 0); wave-uniform values
        unsigned lt[NS], ct[NS];            // (times as bit patterns, so that the slots live in scalar registers like the indices)
        unsigned li[NS], ci[NS];
#pragma unroll
        for (int m = 0; m < NS; ++m) {
            lt[m] = 0u;                     // 0.0f
            ct[m] = 0u;
            ci[m] = 0u;
            li[m] = (m < (int)M.S) ? (unsigned)sd.ind[m] : 0u;
        }
        unsigned crossed = 0;
        float now = 0.0f;
        // smallest |s| over this lane's neurons as the last state pass left it (kTrack; the first pass runs guarded)
        constexpr bool kTrack = UDIV && !HETERO && MATH == 0;
        float s_lo = 0.0f;
        // Candidate bookkeeping.  Every event needs min over neurons of eventTime().  Neurons that will not
        // fire contribute exactly kNever; the few that will (the bump fronts) need a divergent Newton solve.
        // Instead of running that loop once per 64-neuron slice, each lane records its firing neurons in a
        // bitmask (bit k <-> neuron k*64+lane) during the state pass and the solves then run in compacted
        // rounds: one round handles one pending neuron of EVERY lane.  The minimum is taken lexicographically over
        // (time, largest tie key), so the result does not depend on evaluation order.
        unsigned base_key = ~0u;      // this lane's non-firing neurons all stand at kNever: the largest tie key among them (~0: none)
        unsigned pend = 0;
        {
            unsigned a = lane;
            for (unsigned m = store; m != 0u; m &= m - 1u, a += 128u) {
                const unsigned k = (unsigned)__builtin_ctz(m);
                const float bk = HETERO ? B[bidx(a)] : M.beta_mean;
                if (edm::will_fire<MATH, UDIV && !HETERO, (MI_EDM_FIRE_FILTER != 0), GAP>(M, V[a], S[a], bk)) pend |= (1u << pos_of(k));
            }
            pend &= valid;
        }
        // the lane's neuron that will not fire and would win a tie among those (all stand at kNever): the highest bit of the
        // quiet mask (no per-slice tracking); the dead slices' candidate folded in
        auto lowest_quiet = [&]() {
            const unsigned quiet = ~pend & valid;
            base_key = quiet != 0u ? lane_base + ((31u - (unsigned)__builtin_clz(quiet)) << key_sh) : ~0u;
            if (nan_key != ~0u && (base_key == ~0u || nan_key > base_key)) base_key = nan_key;
        };
        lowest_quiet();
        unsigned events = 0;
        unsigned tap_newton = 0, tap_cap = 0, tap_quiet = 0, tap_ties = 0;      // (TAPS only)
        while (crossed < full && now < two_T && events < M.max_events) {
            ++events;
            float best = base_key != ~0u ? edm::kNever : INFINITY;
            unsigned bkey = base_key;         // (~0 with best = +inf: wave_argmin reads key + 1 = 0 as "no candidate")
            // Firing-time solves, one round = the lowest pending neuron of every lane.  Which lane solves a neuron does not
            // matter: the minimum below is lexicographic in (time, tie key).
            __builtin_amdgcn_s_setprio(0);
            if constexpr (MATH == 0) {
            // EXACT math: the round's neurons are compacted into `list` (rank by v_mbcnt over the ballot) and each goes to a
            // lane PAIR (j, j + 32) that shares the two software exponentials and the two IEEE divisions of a Newton
            // iteration (edm::newton_time_paired): 138 instead of 149 ms per 125 000 x 1024 ComputeF.
            while (__any(pend != 0u)) {
                const bool has = pend != 0u;
                const unsigned long long bal = __ballot(has);
                const unsigned total = (unsigned)__builtin_popcountll(bal);
                if (has) {
                    const unsigned pos = (unsigned)__builtin_ctz(pend);
                    pend &= pend - 1u;
                    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                    list[rank] = ((pos_of(pos) * 64u + lane) << 16) | (lane_base + (pos << key_sh));   // neuron | its tie key
                }
                for (unsigned base = 0; base < total; base += 32u) {      // (wave-uniform; more than once only with > 32 solves)
                    const unsigned j = base + (lane & 31u);
                    if (j < total) {
                        const unsigned entry = list[j];
                        const unsigned i = entry >> 16, key = entry & 0xffffu;
                        const unsigned k = i >> 6;
                        const unsigned sl = (unsigned)__builtin_popcount(store & ((1u << k) - 1u));
                        const unsigned a = sl * 128u + (i & 63u);
                        const float bk = HETERO ? B[sl * 64u + (i & 63u)] : M.beta_mean;
                        uint32_t it = 0;
                        const float tau = edm::newton_time_paired<MATH, UDIV && !HETERO, ONE>(M, V[a], S[a], bk, lane >= 32u, TAPS ? &it : nullptr);
                        if (lane < 32u) {
                            if constexpr (TAPS) {
                                tap_newton = max(tap_newton, it);
                                tap_cap += (it >= M.max_iter) ? 1u : 0u;
                                tap_ties += (tau == best && tau < edm::kNever) ? 1u : 0u;      // two firing neurons met in one lane
                            }
                            if (tau < best || (tau == best && key > bkey)) { best = tau; bkey = key; }
                        }
                    }
                }
            }
            } else {
            // FAST math (hardware exp and reciprocal: nothing worth sharing, the pairing costs 10 %): every lane solves its own
            while (__any(pend != 0u)) {
                if (pend != 0u) {
                    const unsigned pos = (unsigned)__builtin_ctz(pend);
                    pend &= pend - 1u;
                    const unsigned k = pos_of(pos);
                    const unsigned a = (unsigned)__builtin_popcount(store & ((1u << k) - 1u)) * 128u + lane;
                    const float bk = HETERO ? B[bidx(a)] : M.beta_mean;
                    uint32_t it = 0;
                    const float tau = edm::newton_time<MATH, UDIV && !HETERO>(M, V[a], S[a], bk, TAPS ? &it : nullptr);
                    if constexpr (TAPS) {
                        tap_newton = max(tap_newton, it);
                        tap_cap += (it >= M.max_iter) ? 1u : 0u;
                        tap_ties += (tau == best && tau < edm::kNever) ? 1u : 0u;
                    }
                    const unsigned key = lane_base + (pos << key_sh);
                    if (tau < best || (tau == best && key > bkey)) { best = tau; bkey = key; }
                }
            }
            }
            if constexpr (TAPS) {
                const float mine = best;
                float bb = best;
                unsigned kk = bkey;
                wave_argmin(bb, kk);
                if (bb >= edm::kNever) tap_quiet += 1u;
                else if (__builtin_popcountll(__ballot(mine == bb)) > 1) tap_ties += 1u;   // ... or of two lanes
            }
            // From here to the next Newton rounds the wave runs at raised priority: the arg-min (two dependent DPP chains), the
            // exponentials and the state pass are what the other waves of the SIMD wait behind when they meet a wave that is in
            // its -- long, few-lane -- Newton rounds (priority 0): -2.3 % at N = 1024, -0.7 % at N = 512 (profiles/r04_ab_runs.log, block r04_ab27;
            // raising it for the state pass alone does nothing).
            __builtin_amdgcn_s_setprio(2);
            wave_argmin(best, bkey);
            // the winner is the same in every lane: in scalar registers the event bookkeeping below (nearest bump,
            // crossed mask) runs on the scalar unit
            unsigned idx = tie_key_neuron((unsigned)__builtin_amdgcn_readfirstlane((int)bkey), tree);
            float dt = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, best)));
            if (padded && !(dt < edm::kNever)) {       // no neuron fires before "never": the padding pair (100.0f, 0) wins (:867-868)
                dt = edm::kNever;
                idx = 0u;
            }
            // analytic state advance (EventDrivenMap.cu:612-617), fused with the firing test for the NEXT event
            float e1, e2u = 0.0f, e3u = 0.0f;
            if constexpr (!HETERO) {
                // the three wave-uniform exponentials of the advance in ONE pass of the software exp: lanes 0, 1, 2 take the
                // three arguments, the results come back through v_readlane (same routine, same inputs: same bits)
                const float arg = (lane == 1u) ? (1.0f - M.beta_mean) * dt : (lane == 2u) ? -M.beta_mean * dt : -dt;
                const float ex = edm::expf_<MATH>(arg);
                e1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ex), 0));
                e2u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ex), 1));
                e3u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ex), 2));
            } else {
                e1 = edm::expf_<MATH>(-dt);
            }
            const unsigned idx4 = idx * 4u;
            const bool sign_settles = edm::gap_settles_sign<GAP>(M);
            typedef __attribute__((address_space(3))) const float lds_cfloat;
            const unsigned w_base = (unsigned)(uintptr_t)(lds_cfloat*)w_lds;
            // guard_tag: std::true_type = the exact quotient checks its numerator's range in every slice (edm::div_by), false_type =
            // the range is known beforehand (below)
            auto state_pass = [&](auto guard_tag) {
            constexpr bool kGuard = decltype(guard_tag)::value;
            float lo = INFINITY;
            unsigned a = lane;
#pragma unroll 1                                                            // rolled: measured (DESIGN_HISTORY.md section 4)
            for (unsigned m = store; m != 0u; m &= m - 1u, a += 128u) {      // live slices only
                const unsigned k = (unsigned)__builtin_ctz(m);
                const float bk = HETERO ? B[bidx(a)] : M.beta_mean;
                const float e2 = HETERO ? edm::expf_<MATH>((1.0f - bk) * dt) : e2u;
                const float e3 = HETERO ? edm::expf_<MATH>(-bk * dt) : e3u;
                const float so = S[a];
                // the coupling value w[|i - idx|] (|i - idx| < kMaxGrid: i < npl*64 <= kMaxGrid, idx < N), addressed in bytes:
                // 4 |i - idx| = |4 i - 4 idx| in one v_sad_u32.  (Every LDS read of the slice before its arithmetic.)
                unsigned w_at;                                            // LDS address: the table's own goes in as v_sad_u32's addend
                asm("v_sad_u32 %0, %1, %2, %3" : "=v"(w_at) : "v"((lane << 2) | (k << 8)), "s"(idx4), "v"(w_base));
                const float wd = *reinterpret_cast<lds_cfloat*>((uintptr_t)w_at);
                float vv = V[a] * e1;
                vv = vv + (M.I * (1.0f - e1) + edm::div_by<MATH, UDIV && !HETERO, kGuard, ONE>(so * e1, 1.0f - bk) * (e2 - 1.0f));
                // reset of the neuron that fired (:615 multiplies every v by (tid != index)): x * 1 == x, so only that
                // neuron needs the multiply (by 0: NaN stays NaN, a finite value becomes a signed zero) -- and only its
                // slice looks for it (a scalar branch: idx and k are wave-uniform)
                if (k == (idx >> 6)) {
                    asm volatile("" : "+v"(vv));      // (keeps the compiler from turning the branch into a select on every slice)
                    vv = (lane == (idx & 63u)) ? vv * 0.0f : vv;
                }
                float sn = so * e3;
                sn = sn + (HETERO ? bk * wd : wd);
                V[a] = vv;
                S[a] = sn;
                if constexpr (kTrack) asm("v_min_f32 %0, |%1|, %0" : "+v"(lo) : "v"(sn));   // (a NaN leaves the bound alone)
                // With 0 < vth - I <= 1 no lane of the slice can fire unless some s is >= 0 (will_fire's first exit): most slices
                // of most events leave here with one scalar branch -- inside will_fire the same exit costs an exec-mask save /
                // restore and the two instructions that fold an all-false result into pend.
                if (!sign_settles || __any(sn >= 0.0f))
                pend |= (edm::will_fire<MATH, UDIV && !HETERO, (MI_EDM_FIRE_FILTER != 0), GAP>(M, vv, sn, bk) ? 1u : 0u) << pos_of(k);   // (padding lanes: masked below)
            }
            s_lo = lo;
            };
            if constexpr (kTrack) {
                // The exact quotient (so * e1) / (1 - beta) needs |so * e1| in [2^-100, 2^101) (or a NaN); its own test of that is
                // two compares and a scalar branch in every slice.  Above: |so * e1| <= |so| < 2^101 for the whole evolution when
                // the launch says so (s_bounded, see kBigS).  Below: the previous pass left the smallest |s| over this lane's
                // neurons, and RN(s_lo * e1) >= 2^-99 puts every |so * e1| above 2^-100 -- then the whole pass runs without the
                // per-slice test.  (First event, zeros, subnormals, a 100-unit "no neuron fires" step: the guarded pass.)
                const bool known = s_bounded && s_lo * e1 >= 0x1.0p-99f;
                if (__any(!known)) state_pass(std::true_type{});
                else state_pass(std::false_type{});
            } else {
                state_pass(std::true_type{});
            }
            pend &= valid;
            lowest_quiet();
            now = now + dt;
            // which bump does the event belong to ([D3]: the reference's increment rule, :625-629)
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
                const unsigned now_b = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(now));   // (wave-uniform)
#pragma unroll
                for (int m = 0; m < NS; ++m) {
                    if (mi == (unsigned)m) {
                        if (after) { ct[m] = now_b; ci[m] = idx; }
                        else { lt[m] = now_b; li[m] = idx; }
                    }
                }
                if (after) crossed += (1u << mi);
            }
        }
        // [spike][realisation] layout, EventDrivenMap.cu:661-668
#pragma unroll
        for (int m = 0; m < NS; ++m) {
            if (lane == (unsigned)m && m < (int)M.S) {
                const size_t k = (size_t)m * M.R + r;
                g_t0[k] = __uint_as_float(lt[m]);
                g_i0[k] = (unsigned short)li[m];
                g_t1[k] = __uint_as_float(ct[m]);
                g_i1[k] = (unsigned short)ci[m];
            }
        }
        if (lane == 0) g_accept[r] = (crossed == full) ? 1u : 0u;
        if constexpr (TAPS) {
            unsigned mx = tap_newton, cap = tap_cap, ties = tap_ties;
            for (int off = 32; off > 0; off >>= 1) {
                mx = max(mx, (unsigned)__shfl_xor((int)mx, off));
                cap += (unsigned)__shfl_xor((int)cap, off);
                ties = max(ties, (unsigned)__shfl_xor((int)ties, off));
            }
            if (lane == 0) {
                atomicAdd(&taps[kTapEvents], (unsigned long long)events);
                atomicMax(&taps[kTapMaxEvents], (unsigned long long)events);
                atomicMax(&taps[kTapMaxNewton], (unsigned long long)mx);
                atomicAdd(&taps[kTapNewtonCap], (unsigned long long)cap);
                atomicAdd(&taps[kTapEventCap], (crossed < full && events >= M.max_events) ? 1ull : 0ull);
                atomicAdd(&taps[kTapAccepted], (crossed == full) ? 1ull : 0ull);
                atomicAdd(&taps[kTapNoFiring], (unsigned long long)tap_quiet);
                atomicAdd(&taps[kTapTies], (unsigned long long)ties);
            }
        }
    }
}

// ---- EvolveKernel, latency form: one WORKGROUP of W waves per realisation ---------------------------------
// With few realisations (the reference's Driver.cu runs 1000, dedup_identical runs 1) the wave-per-realisation kernel
// leaves most SIMDs with at most one wave and every event pays the full latency of a 16-neuron-per-lane state pass
// (5.3 us per event at N = 1024).  Here the neurons of one realisation are spread over W waves (slice q = k*W + wave
// holds neurons q*64 .. q*64+63); every wave keeps its own copy of the (wave-uniform) event bookkeeping, and the
// only exchange per event is the lexicographic (time, index) minimum: each wave's DPP minimum goes through a
// double-buffered LDS slot and one barrier.  Arithmetic per neuron and the arg-min rule are those of evolve_kernel,
// so the results are bit-identical (tests/test_edm_gpu.py).
template <int MATH, bool HETERO, int NS, int W>
__global__ __launch_bounds__(64 * W) void evolve_wg_kernel(edm::Model M, SpikeSeeds sd, const float* __restrict__ v0,
                                                           const float* __restrict__ s0, const float* __restrict__ w,
                                                           float* __restrict__ g_t0, unsigned short* __restrict__ g_i0,
                                                           float* __restrict__ g_t1, unsigned short* __restrict__ g_i1,
                                                           unsigned* __restrict__ g_accept)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr unsigned kBlockT = 64u * W;
    const unsigned npl = (M.N + kBlockT - 1u) / kBlockT;      // slices per wave
    const unsigned slots = npl * kBlockT;
    float* w_lds = lds;
    for (unsigned i = threadIdx.x; i < (unsigned)kMaxGrid; i += kBlockT) {   // homogeneous model: RN(beta * w[d]), as in evolve_kernel
        const float wi = (i < M.N) ? w[i] : 0.0f;
        w_lds[i] = HETERO ? wi : M.beta_mean * wi;
    }
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float* V = lds + kMaxGrid;
    float* S = V + slots;
    float* B = S + slots;                                      // only touched when HETERO
    unsigned* red = reinterpret_cast<unsigned*>(S + slots + (HETERO ? slots : 0u));   // [2][W][2]: time bits, index
    const unsigned full = (1u << M.S) - 1u;
    const float two_T = 2.0f * M.T;
    const bool tree = (M.N & 31u) == 0u;                   // arg-min ties as the reference breaks them (tie_key)
    const bool padded = tree && (M.N >> 5) < 32u;
    unsigned qkey[kMaxGrid / kBlockT];                      // tie keys of this lane's neurons, slice by slice
#pragma unroll
    for (unsigned k = 0; k < kMaxGrid / kBlockT; ++k) qkey[k] = tie_key((k * W + wave) * 64u + lane, tree);
    __syncthreads();

    for (unsigned r = blockIdx.x; r < M.R; r += gridDim.x) {
        unsigned valid = 0;           // bit k: neuron (k*W + wave)*64 + lane exists
        for (unsigned k = 0; k < npl; ++k) {
            const unsigned i = (k * W + wave) * 64u + lane;
            const bool act = i < M.N;
            V[i] = act ? v0[i] : 0.0f;
            S[i] = act ? s0[i] : 0.0f;
            if constexpr (HETERO) B[i] = edm::beta_of<MATH>(M.beta_mean, M.beta_sigma, M.seed, M.N, (uint64_t)r + M.real_offset, act ? i : 0u);
            valid |= act ? (1u << k) : 0u;
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
        unsigned base_key = 0;
        unsigned pend = 0;
        for (unsigned k = 0; k < npl; ++k) {
            const unsigned i = (k * W + wave) * 64u + lane;
            if (i < M.N) {
                const float bk = HETERO ? B[i] : M.beta_mean;
                if (edm::will_fire<MATH>(M, V[i], S[i], bk)) pend |= (1u << k);
            }
        }
        // the lane's neuron that will not fire and would win a tie among those (at most 1024 / (64 W) = 4 slices per lane)
        auto lowest_quiet = [&]() {
            const unsigned quiet = ~pend & valid;
            base_t = quiet != 0u ? edm::kNever : INFINITY;
            base_key = 0;
#pragma unroll
            for (unsigned k = 0; k < kMaxGrid / kBlockT; ++k)
                if (((quiet >> k) & 1u) && qkey[k] > base_key) base_key = qkey[k];
        };
        lowest_quiet();
        unsigned events = 0;
        while (crossed < full && now < two_T && events < M.max_events) {
            ++events;
            float best = base_t;
            unsigned bkey = base_key;
            while (__any(pend != 0u)) {
                if (pend != 0u) {
                    const unsigned k = (unsigned)__builtin_ctz(pend);
                    pend &= pend - 1u;
                    const unsigned i = (k * W + wave) * 64u + lane;
                    const float bk = HETERO ? B[i] : M.beta_mean;
                    const float tau = edm::newton_time<MATH>(M, V[i], S[i], bk);
                    unsigned key = qkey[0];
#pragma unroll
                    for (unsigned q = 1; q < kMaxGrid / kBlockT; ++q) key = (k == q) ? qkey[q] : key;
                    if (tau < best || (tau == best && key > bkey)) { best = tau; bkey = key; }
                }
            }
            wave_argmin(best, bkey);
            // workgroup-wide minimum of (time, largest tie key) through LDS (double-buffered by event parity: one barrier per event)
            unsigned* slot = red + (events & 1u) * (2u * W);
            if (lane == 0) {
                slot[2u * wave] = __float_as_uint(best);
                slot[2u * wave + 1u] = bkey;
            }
            __syncthreads();
            unsigned idx;
            {
                unsigned tb = slot[0], kb = slot[1];
#pragma unroll
                for (int q = 1; q < W; ++q) {
                    const unsigned t2 = slot[2 * q], k2 = slot[2 * q + 1];
                    if (t2 < tb || (t2 == tb && k2 > kb)) { tb = t2; kb = k2; }
                }
                best = __uint_as_float(tb);
                idx = tie_key_neuron(kb, tree);
            }
            if (padded && !(best < edm::kNever)) {   // the padding pair (100.0f, 0) of the reference's second stage wins (:867-868)
                best = edm::kNever;
                idx = 0u;
            }
            const float dt = best;
            float e1, e2u = 0.0f, e3u = 0.0f;
            if constexpr (!HETERO) {         // the three uniform exponentials in one pass of the software exp (lanes 0, 1, 2)
                const float arg = (lane == 1u) ? (1.0f - M.beta_mean) * dt : (lane == 2u) ? -M.beta_mean * dt : -dt;
                const float ex = edm::expf_<MATH>(arg);
                e1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ex), 0));
                e2u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ex), 1));
                e3u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ex), 2));
            } else {
                e1 = edm::expf_<MATH>(-dt);
            }
            for (unsigned k = 0; k < npl; ++k) {
                const unsigned sl = k * W + wave;
                const unsigned i = sl * 64u + lane;
                const float bk = HETERO ? B[i] : M.beta_mean;
                const float e2 = HETERO ? edm::expf_<MATH>((1.0f - bk) * dt) : e2u;
                const float e3 = HETERO ? edm::expf_<MATH>(-bk * dt) : e3u;
                const float so = S[i];
                float vv = V[i] * e1;
                vv = vv + (M.I * (1.0f - e1) + edm::div_<MATH>(so * e1, 1.0f - bk) * (e2 - 1.0f));
                vv = (i == idx) ? vv * 0.0f : vv;             // reset of the neuron that fired
                float sn = so * e3;
                const unsigned dist = (unsigned)abs((int)i - (int)idx);   // < kMaxGrid
                sn = sn + (HETERO ? bk * w_lds[dist] : w_lds[dist]);
                V[i] = vv;
                S[i] = sn;
                if (edm::will_fire<MATH>(M, vv, sn, bk)) pend |= (1u << k);   // (padding lanes: masked below)
            }
            pend &= valid;
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
        if (wave == 0) {
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
        __syncthreads();   // the next realisation re-initialises V/S and reuses the reduction slots
    }
}

// dedup_identical: replicate the events of the one evolved realisation into every [spike][realisation] row
__global__ __launch_bounds__(256) void replicate_events_kernel(unsigned S, unsigned R, const float* __restrict__ one_t0,
                                                               const unsigned short* __restrict__ one_i0,
                                                               const float* __restrict__ one_t1,
                                                               const unsigned short* __restrict__ one_i1,
                                                               const unsigned* __restrict__ one_accept,
                                                               float* __restrict__ t0, unsigned short* __restrict__ i0,
                                                               float* __restrict__ t1, unsigned short* __restrict__ i1,
                                                               unsigned* __restrict__ accept)
{
    const unsigned acc = one_accept[0];
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (size_t)gridDim.x * blockDim.x) {
        for (unsigned m = 0; m < S; ++m) {
            const size_t k = (size_t)m * R + r;
            t0[k] = one_t0[m];
            i0[k] = one_i0[m];
            t1[k] = one_t1[m];
            i1[k] = one_i1[m];
        }
        accept[r] = acc;
    }
}

// Does ONE correction step give the IEEE quotient a / c for every a?  All 2^23 significands of a in [1, 2): inside the range
// edm::div_by guards, the quotient scales exactly with a's exponent, and both forms are odd in a and in c.  bad[0] counts
// the significands for which the two differ (none, for every divisor seen so far; the launch does not rely on that).
__global__ __launch_bounds__(256) void divisor_check_kernel(float c, unsigned* __restrict__ bad)
{
    const unsigned m = blockIdx.x * 256u + threadIdx.x;              // grid = 2^23 / 256 workgroups
    const float a = __uint_as_float(0x3f800000u | m);
    const bool differs = edm::div_by<0, true, false, true>(a, c) != a / c;
    const unsigned long long d = __ballot(differs);
    if (d != 0ull && (threadIdx.x & 63u) == 0u) atomicAdd(bad, (unsigned)__builtin_popcountll(d));
}

template <int MATH>
__global__ void math_probe_kernel(int op, const float* a, const float* b, float* out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (op) {
        case 0: r = edm::expf_<MATH>(a[i]); break;
        case 1: r = edm::logf_<MATH>(a[i]); break;
        case 2: r = edm::powf_<MATH>(a[i], b[i]); break;
        case 4:                                                   // quotient by a launch-uniform divisor (b[0]) as a launch takes it:
            r = (fabsf(b[0]) >= 0x1.0p-20f && fabsf(b[0]) <= 0x1.0p+20f)   // the five-operation form only for divisors the host gate
                    ? edm::div_by<MATH, true>(a[i], b[0]) : a[i] / b[0];  // admits (launch_evolve: uniform_divisors_ok), else IEEE
            break;
        case 5: r = a[i] / b[0]; break;                           // the IEEE expansion, for comparison
        case 6: case 7: case 8: case 9: case 10: case 11: {       // will_fire(v0 = a, s0 = b): exact path (even op) / with the
            edm::Model M = {};                                    // hardware pre-decision (odd op); beta by op pair
            M.vth = 1.0f;
            M.I = 0.9f;
            const float beta = op < 8 ? 13.0589f : op < 10 ? 1.5f : 0.7f;
            r = (op & 1) ? (edm::will_fire<MATH, false, true>(M, a[i], b[i], beta) ? 1.0f : 0.0f)
                         : (edm::will_fire<MATH, false, false>(M, a[i], b[i], beta) ? 1.0f : 0.0f);
            break;
        }
        case 12: r = edm::other_half(a[i], (threadIdx.x & 32u) != 0u); break;   // a[i ^ 32]: the lane-pair exchange of the paired solves
        case 13:                                                  // op 4 with ONE correction step (only exact for divisors that
            r = (fabsf(b[0]) >= 0x1.0p-20f && fabsf(b[0]) <= 0x1.0p+20f)   // pass mi_edm_divisor_check; the caller decides)
                    ? edm::div_by<MATH, true, true, true>(a[i], b[0]) : a[i] / b[0];
            break;
        default: r = edm::erfinvf_<MATH>(a[i]); break;
    }
    out[i] = r;
}

}  // namespace

struct mi_edm {
    mi_ctx* ctx;
    mi_edm_params p;
    edm::Model M;
    // device state
    float *d_v = nullptr, *d_s = nullptr, *d_w = nullptr;
    float *d_t0 = nullptr, *d_t1 = nullptr, *d_restricted = nullptr;
    uint16_t *d_i0 = nullptr, *d_i1 = nullptr;
    uint32_t* d_accept = nullptr;
    char* d_one = nullptr;         // dedup_identical: events of the one evolved realisation (kOneBytes)
    char* d_result = nullptr;      // mean f32[8] | count u32 (+pad) | sums f64[8]
    char* h_result = nullptr;      // pinned mirror
    unsigned* d_aux = nullptr;     // [0] live-slice mask written by the lift kernel
    unsigned* h_aux = nullptr;     // pinned mirror of d_aux[0]
    SpikeSeeds last_sd;            // seeds of the last evaluation (mi_edm_debug_counters re-runs Evolve with them)
    size_t alloc_real = 0;
    bool w_valid = false;
    bool have_run = false;
    int device = 0;                    // copied from the context at creation (destroy must not dereference the context)
    hipStream_t pending_stream = nullptr;   // stream of the evaluation in flight
    bool pending = false;              // mi_edm_compute_f_begin has enqueued an evaluation, _end has not collected it
    double pending_U0[kMaxSpikes + 1] = {0};
    uint16_t seed_ind[kMaxSpikes] = {0};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    float last_ms[4] = {0, 0, 0, 0};
    // test / tuning knobs (mi_edm_set_kernel_choice); every choice gives bit-identical results
    int waves_per_real = 0;            // 1 | 4 forces an evolve kernel form (0: by realisation count)
    bool no_uniform_div = false;       // never take the exact quotient by wave-uniform divisors
    // one-step quotient (edm::div_by ONE): the divisor 1 - beta it was last proved (or refuted) for; NaN = none yet
    float checked_divisor = NAN;
    float w_abs_max = INFINITY;        // max |w| of the coupling table on the device (NaN / inf entries: inf)
    bool one_step_exact = false;
};

namespace {

constexpr size_t kResultBytes = 8 * 4 + 8 + (2 * 8 + 1) * 8;   // mean f32[8] | count u32 (+pad) | partial block f64[2*8+1]
// evolve kernel choice by realisation count (launch_evolve): four waves per realisation (latency form) while every
// realisation can have a CU of its own and there are enough of them, else one wave per realisation (throughput form).
// scripts/gpu_edm_wpr.py on the round-4 kernels (profiles/r04_edm_waves_per_realisation.log), four waves against one:
// N = 1024: R = 64 .. 256 2.35 vs 2.80 ms, R = 8 .. 32 3.05 vs 2.79 ms, R = 300 .. 512 equal, R = 600 3.24 vs 2.83 ms;
// N = 512: R = 64 .. 256 1.10 vs 1.19 ms, otherwise the throughput form.  Two or sixteen waves per realisation never won.
constexpr unsigned kWgLow = 48;
// t0 f32[8] | t1 f32[8] | i0 u16[8] | i1 u16[8] | accept u32
constexpr size_t kOneT0 = 0, kOneT1 = 32, kOneI0 = 64, kOneI1 = 80, kOneAccept = 96, kOneBytes = 128;

mi_status validate(const mi_ctx* ctx, const mi_edm_params* p)
{
    if (!p) return mi::fail(ctx, MI_ERR_INVALID_ARG, "mi_edm: params is NULL");
    if (p->n_spikes < 1 || p->n_spikes > (uint32_t)kMaxSpikes)
        return mi::fail(ctx, MI_ERR_INVALID_ARG, "mi_edm: n_spikes=%u not in [1,%d]", p->n_spikes, kMaxSpikes);
    if (p->n_grid < 2 || p->n_grid > (uint32_t)kMaxGrid)
        return mi::fail(ctx, MI_ERR_INVALID_ARG, "mi_edm: n_grid=%u not in [2,%d]", p->n_grid, kMaxGrid);
    if (p->n_real < 1) return mi::fail(ctx, MI_ERR_INVALID_ARG, "mi_edm: n_real must be positive");
    if (!(p->time_horizon > 0.0f)) return mi::fail(ctx, MI_ERR_INVALID_ARG, "mi_edm: time_horizon must be > 0");
    if (!(p->beta_stddev >= 0.0f)) return mi::fail(ctx, MI_ERR_INVALID_ARG, "mi_edm: beta_stddev must be >= 0");
    if (!(p->newton_tol >= 0.0)) return mi::fail(ctx, MI_ERR_INVALID_ARG, "mi_edm: newton_tol must be >= 0");
    if (p->max_events < 1 || p->max_events > kMaxEventsLimit)
        return mi::fail(ctx, MI_ERR_INVALID_ARG, "mi_edm: max_events=%u not in [1, 2^24]", p->max_events);
    if (p->newton_max_iter > 100000u)
        return mi::fail(ctx, MI_ERR_INVALID_ARG, "mi_edm: newton_max_iter=%u exceeds 100000", p->newton_max_iter);
    if (p->math_mode != MI_EDM_MATH_EXACT && p->math_mode != MI_EDM_MATH_FAST)
        return mi::fail(ctx, MI_ERR_INVALID_ARG, "mi_edm: unknown math_mode %d", p->math_mode);
    if ((uint64_t)p->n_spikes * p->n_real > 0xfffffff0ull)
        return mi::fail(ctx, MI_ERR_INVALID_ARG, "mi_edm: n_spikes*n_real too large");
    return MI_OK;
}

void fill_model(const mi_edm_params& p, edm::Model* M)
{
    M->vth = p.vth; M->a1 = p.a1; M->a2 = p.a2; M->b1 = p.b1; M->b2 = p.b2; M->I = p.I; M->L = p.L;
    float tf = (float)p.newton_tol;
    if ((double)tf > p.newton_tol) tf = nextafterf(tf, -INFINITY);
    M->tol_f = tf;
    M->max_iter = p.newton_max_iter;
    M->S = p.n_spikes; M->N = p.n_grid; M->R = p.n_real;
    M->T = p.time_horizon;
    M->beta_mean = p.beta_mean; M->beta_sigma = p.beta_stddev;
    M->seed = p.seed;
    M->real_offset = p.real_offset;
    M->max_events = p.max_events;
}

void free_real_buffers(mi_edm* e)
{
    void* bufs[] = {e->d_t0, e->d_t1, e->d_restricted, e->d_i0, e->d_i1, e->d_accept};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    e->d_t0 = e->d_t1 = e->d_restricted = nullptr;
    e->d_i0 = e->d_i1 = nullptr;
    e->d_accept = nullptr;
    e->alloc_real = 0;
}

mi_status ensure_buffers(mi_edm* e)
{
    mi_ctx* ctx = e->ctx;
    const size_t SR = (size_t)kMaxSpikes * e->p.n_real;   // sized for any n_spikes
    if (e->alloc_real >= e->p.n_real) return MI_OK;
    free_real_buffers(e);
    MI_HIP(ctx, hipMalloc(&e->d_t0, SR * sizeof(float)));
    MI_HIP(ctx, hipMalloc(&e->d_t1, SR * sizeof(float)));
    MI_HIP(ctx, hipMalloc(&e->d_restricted, SR * sizeof(float)));
    MI_HIP(ctx, hipMalloc(&e->d_i0, SR * sizeof(uint16_t)));
    MI_HIP(ctx, hipMalloc(&e->d_i1, SR * sizeof(uint16_t)));
    MI_HIP(ctx, hipMalloc(&e->d_accept, (size_t)e->p.n_real * sizeof(uint32_t)));
    e->alloc_real = e->p.n_real;
    return MI_OK;
}

// BuildCouplingKernel + circshift (EventDrivenMap.cu:111-129, :826-841), on the host
template <int MATH>
void build_coupling(const mi_edm_params& p, float* w)
{
    const uint32_t N = p.n_grid;
    std::vector<float> tmp(N);
    const float h = (2.0f * p.L) / (float)N;
    for (uint32_t i = 0; i < N; ++i) {
        const float x = -p.L + h * (float)i;
        const float ax = fabsf(x);
        const float k = p.a1 * edm::expf_<MATH>(-p.b1 * ax) - p.a2 * edm::expf_<MATH>(-p.b2 * ax);
        tmp[i] = ((k * 2.0f) * p.L) / (float)N;
    }
    const uint32_t shift = N / 2;
    for (uint32_t i = 0; i < N; ++i) w[i] = tmp[(i + shift) % N];
}

// initialSpikeInd (EventDrivenMap.cu:361-372); [D5] stale entries persist in e->seed_ind
void seed_indices(const mi_edm_params& p, const double* Z, uint16_t* ind)
{
    const uint32_t N = p.n_grid, S = p.n_spikes;
    ind[0] = (uint16_t)(N / 2);
    for (uint32_t m = 1; m < S; ++m) {
        for (uint32_t i = ind[m - 1]; i > 0; --i) {
            const float xi = -p.L + ((float)(2u * i) * p.L) / (float)N;
            if ((double)xi < -Z[0] * Z[m]) {
                ind[m] = (uint16_t)i;
                break;
            }
        }
    }
}

// dynamic LDS of evolve_kernel: coupling table, the live slices' state per wave, the per-wave pending-neuron lists
size_t evolve_lds_bytes(bool hetero, unsigned live)
{
    return ((size_t)kMaxGrid + (size_t)(kEvolveBlock / 64) * ((hetero ? 3 : 2) * 64u * (unsigned)__builtin_popcount(live) + 64u)) * sizeof(float);
}

// which form of the evolve kernel a launch takes: 1 = one wave per realisation (throughput), 4 = one workgroup of four
// waves per realisation (latency).  mi_edm_set_kernel_choice overrides the choice.
int evolve_form(const mi_edm* e)
{
    const bool hetero = e->p.beta_stddev != 0.0f;
    const bool dedup = e->p.dedup_identical != 0 && !hetero && e->p.n_real > 1;
    const unsigned Reff = dedup ? 1u : e->p.n_real;
    if (e->waves_per_real) return e->waves_per_real;
    const unsigned cus = (unsigned)(e->ctx->compute_units > 0 ? e->ctx->compute_units : 256);
    return (Reff >= kWgLow && Reff <= cus) ? 4 : 1;
}

// Is the ONE-step quotient by c exact (edm::div_by ONE)?  Proved or refuted on the device over every significand, once per
// divisor value (about 10 us of kernel + a 4-byte read-back; the result is kept with the handle).
mi_status one_step_quotient_exact(mi_edm* e, float c, bool* exact)
{
    mi_ctx* ctx = e->ctx;
    if (!(e->checked_divisor == c)) {        // (NaN: nothing checked yet)
        MI_HIP(ctx, hipMemsetAsync(e->d_aux + 1, 0, sizeof(unsigned), ctx->stream));
        hipLaunchKernelGGL(divisor_check_kernel, dim3((1u << 23) / 256u), dim3(256), 0, ctx->stream, c, e->d_aux + 1);
        MI_LAUNCH_CHECK(ctx, "divisor check kernel");
        MI_HIP(ctx, hipMemcpyAsync(e->h_aux + 1, e->d_aux + 1, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        MI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        e->one_step_exact = e->h_aux[1] == 0u;
        e->checked_divisor = c;
    }
    *exact = e->one_step_exact;
    return MI_OK;
}

// live: the lift kernel's live-slice mask (wave-per-realisation form only; ignored by the latency form)
template <int MATH>
mi_status launch_evolve(mi_edm* e, const SpikeSeeds& sd, unsigned live_word)
{
    // the lift kernel's word: live slices + "some |s| is infinite or >= 2^60"; what goes to the kernel in the flag's place is
    // "|s| stays below 2^101 for the whole evolution" (kBigS): a bounded lift profile, beta > 0 and a bounded coupling table
    const bool s_bounded = (live_word & kBigFlag) == 0u && e->p.beta_mean > 0.0f && e->p.beta_mean * e->w_abs_max < 0.5f * kBigS;
    const unsigned live = (live_word & kSliceMask) | (s_bounded ? kBigFlag : 0u);
    mi_ctx* ctx = e->ctx;
    const unsigned N = e->p.n_grid, R = e->p.n_real;
    const bool hetero = e->p.beta_stddev != 0.0f;
    const size_t lds_bytes = evolve_lds_bytes(hetero, live & kSliceMask);
    // One workgroup per four realisations, however many that is: the hardware's workgroup dispatcher then IS the work
    // queue (a finished workgroup's slot goes to the next four realisations; per-workgroup set-up is the 4 KiB coupling
    // table, microseconds against milliseconds per realisation), so the launch ends within one realisation's time of the
    // ideal.  A grid capped at a multiple of the resident set left waves with 6 or 7 realisations each and a ragged end.
    // (profiles/r04_edm_grid_rounds.log: grid capped at 1 / 2 / 4 x the resident set 181 / 166 / 157 ms, uncapped 149 ms.)
    unsigned blocks = (R + 3) / 4;
    // dedup_identical: without heterogeneity the realisations are R copies of one computation
    const bool dedup = e->p.dedup_identical != 0 && !hetero && R > 1;
    edm::Model M = e->M;
    float *t0 = e->d_t0, *t1 = e->d_t1;
    uint16_t *i0 = e->d_i0, *i1 = e->d_i1;
    uint32_t* accept = e->d_accept;
    if (dedup) {
        M.R = 1;
        blocks = 1;
        t0 = (float*)(e->d_one + kOneT0);
        t1 = (float*)(e->d_one + kOneT1);
        i0 = (uint16_t*)(e->d_one + kOneI0);
        i1 = (uint16_t*)(e->d_one + kOneI1);
        accept = (uint32_t*)(e->d_one + kOneAccept);
    }
    // Few realisations: spread each one over a workgroup of 4 waves (evolve_wg_kernel, bit-identical results).
    const unsigned Reff = M.R;
    const int wpr = evolve_form(e);
    const bool three = e->p.n_spikes <= 3;
    if (wpr == 1) {
    // (N not a multiple of 32 -- no launch the reference could make -- runs the !TREE instantiation: ties to the lowest index)
    const bool whole_warps = (N & 31u) == 0u;
#define MI_EVOLVE_G(H, NS, UD, G, O)                                                                              \
    do {                                                                                                          \
        if (whole_warps)                                                                                          \
            hipLaunchKernelGGL((evolve_kernel<MATH, H, NS, UD, false, true, G, O>), dim3(blocks), dim3(kEvolveBlock), lds_bytes, ctx->stream, \
                               M, sd, live, nullptr, e->d_v, e->d_s, e->d_w, t0, i0, t1, i1, accept);             \
        else                                                                                                      \
            hipLaunchKernelGGL((evolve_kernel<MATH, H, NS, false, false, false, G, false>), dim3(blocks), dim3(kEvolveBlock), lds_bytes, ctx->stream, \
                               M, sd, live, nullptr, e->d_v, e->d_s, e->d_w, t0, i0, t1, i1, accept);             \
    } while (0)
    // (UD is only ever true together with gap_ok, and the one-step quotient only with UD)
#define MI_EVOLVE(H, NS, UD)                                                                                      \
    do {                                                                                                          \
        if (!gap_ok) MI_EVOLVE_G(H, NS, false, false, false);                                                     \
        else if (UD && one_step) MI_EVOLVE_G(H, NS, UD, true, UD);                                                \
        else MI_EVOLVE_G(H, NS, UD, true, false);                                                                 \
    } while (0)
        // The exact quotient by uniform divisors (edm::div_by): at every realisation count since the state pass runs it without
        // its guard (range tracking) -- R = 600 .. 3000: -4 .. -10 %, beyond: the kernel it always took (profiles/r04_ab_runs.log, block r04_ab20).
        // (Rounds 2-3 took it only from three waves per SIMD: with its guard it was slower on an underfilled device.)
        // edm::div_by<.., true> relies on its divisors -- 1 - beta, beta - 1 and vth - I -- lying in [2^-20, 2^20] in magnitude.
        auto in_range = [](float c) { return fabsf(c) >= 0x1.0p-20f && fabsf(c) <= 0x1.0p+20f; };
        const float gap = e->p.vth - e->p.I;               // 0 < vth - I <= 1 (edm::gap_settles_sign): a template flag of the kernels
        const bool gap_ok = gap > 0.0f && gap <= 1.0f;
        const bool uniform_divisors_ok = in_range(1.0f - e->p.beta_mean) && in_range(e->p.beta_mean - 1.0f) && in_range(gap) && gap_ok;
        const bool udiv = MATH == 0 && !hetero && !e->no_uniform_div && uniform_divisors_ok;
        bool one_step = false;                             // a single correction step of that quotient, when it is exact for 1 - beta
        if (udiv && whole_warps) {
            const mi_status st = one_step_quotient_exact(e, 1.0f - e->p.beta_mean, &one_step);
            if (st != MI_OK) return st;
        }
        if (hetero) { if (three) MI_EVOLVE(true, 3, false); else MI_EVOLVE(true, kMaxSpikes, false); }
        else if (udiv) { if (three) MI_EVOLVE(false, 3, true); else MI_EVOLVE(false, kMaxSpikes, true); }
        else { if (three) MI_EVOLVE(false, 3, false); else MI_EVOLVE(false, kMaxSpikes, false); }
#undef MI_EVOLVE
#undef MI_EVOLVE_G
    } else {
        const unsigned bt = 64u * (unsigned)wpr;
        (void)live;
        const unsigned wslots = ((N + bt - 1) / bt) * bt;
        const size_t wlds = ((size_t)kMaxGrid + (size_t)(hetero ? 3 : 2) * wslots) * sizeof(float) + (size_t)4 * wpr * sizeof(unsigned);
        const unsigned wper_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / wlds));
        const unsigned wblocks = std::min<unsigned>(Reff, (unsigned)(ctx->compute_units > 0 ? ctx->compute_units : 256) * wper_cu);
#define MI_EVOLVE_WG(H, NS, WV)                                                                                   \
    hipLaunchKernelGGL((evolve_wg_kernel<MATH, H, NS, WV>), dim3(wblocks), dim3(64 * WV), wlds, ctx->stream, M, sd,  \
                       e->d_v, e->d_s, e->d_w, t0, i0, t1, i1, accept)
        if (hetero) { if (three) MI_EVOLVE_WG(true, 3, 4); else MI_EVOLVE_WG(true, kMaxSpikes, 4); }
        else { if (three) MI_EVOLVE_WG(false, 3, 4); else MI_EVOLVE_WG(false, kMaxSpikes, 4); }
#undef MI_EVOLVE_WG
    }
    MI_LAUNCH_CHECK(ctx, "evolve kernel");
    if (dedup) {
        const unsigned grid = mi::stream_grid(ctx, R, 256);
        hipLaunchKernelGGL(replicate_events_kernel, dim3(grid), dim3(256), 0, ctx->stream, e->p.n_spikes, R, t0, i0, t1, i1,
                           accept, e->d_t0, e->d_i0, e->d_t1, e->d_i1, e->d_accept);
        MI_LAUNCH_CHECK(ctx, "replicate-events kernel");
    }
    return MI_OK;
}

template <int MATH>
mi_status run_pipeline(mi_edm* e, const SpikeSeeds& sd)
{
    mi_ctx* ctx = e->ctx;
    if (!e->w_valid) {
        std::vector<float> w(e->p.n_grid);
        build_coupling<MATH>(e->p, w.data());
        float wmax = 0.0f;
        for (float x : w) wmax = (fabsf(x) <= wmax) ? wmax : fabsf(x);      // (a NaN entry ends up as the maximum)
        e->w_abs_max = (wmax == wmax) ? wmax : INFINITY;
        MI_HIP(ctx, hipMemcpyAsync(e->d_w, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        MI_HIP(ctx, hipStreamSynchronize(ctx->stream));   // w is a stack-lifetime host buffer
        e->w_valid = true;
    }
    MI_HIP(ctx, hipEventRecord(e->ev[0], ctx->stream));
    hipLaunchKernelGGL((lift_kernel<MATH>), dim3(1), dim3(kMaxGrid), 0, ctx->stream, e->M, sd, e->d_v, e->d_s, e->d_aux);
    MI_LAUNCH_CHECK(ctx, "lift kernel");
    MI_HIP(ctx, hipEventRecord(e->ev[1], ctx->stream));
    // The throughput form sizes its LDS (hence its residency: five workgroups per CU instead of four at the reference's
    // parameters) by the number of live slices, which only the lift profile knows: one 4-byte read-back and a wait for the
    // lift kernel (tens of microseconds against an evolve of tens of milliseconds).  The latency form carries every slice
    // and does not wait.
    unsigned live = 0;
    if (evolve_form(e) == 1) {
        MI_HIP(ctx, hipMemcpyAsync(e->h_aux, e->d_aux, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        MI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        live = *e->h_aux;
    }
    mi_status st = launch_evolve<MATH>(e, sd, live);
    if (st != MI_OK) return st;
    MI_HIP(ctx, hipEventRecord(e->ev[2], ctx->stream));
    st = mi_restrict_mean_f32_dev(ctx, e->d_t0, e->d_i0, e->d_t1, e->d_i1, e->d_accept, e->p.time_horizon, e->p.L,
                                  e->p.n_grid, e->p.n_real, e->p.n_spikes,
                                  // realisation 0 of the WHOLE ensemble is the one the reference drops: only the shard that
                                  // holds it applies the rule locally (the others report plain sums)
                                  (e->p.mean_quirk != 0 && e->p.real_offset == 0) ? 1 : 0, nullptr,
                                  (float*)e->d_result, (uint32_t*)(e->d_result + 32), (double*)(e->d_result + 40));
    if (st != MI_OK) return st;
    MI_HIP(ctx, hipEventRecord(e->ev[3], ctx->stream));
    MI_HIP(ctx, hipMemcpyAsync(e->h_result, e->d_result, kResultBytes, hipMemcpyDeviceToHost, ctx->stream));
    return MI_OK;   // nothing waited for: mi_edm_compute_f_end synchronises
}

}  // namespace

extern "C" {

void mi_edm_default_params(mi_edm_params* p)
{
    if (!p) return;
    // parameters.hpp:1-15; Driver.cu:16,19; counterMax := 100 (undefined upstream, EventDrivenMap.cu:564)
    p->vth = 1.0f; p->a1 = 11.0f; p->a2 = 7.0f; p->b1 = 5.0f; p->b2 = 3.5f; p->I = 0.9f; p->L = 3.0f;
    p->newton_tol = 1e-6;
    p->newton_max_iter = 100;
    p->n_spikes = 3;
    p->time_horizon = 5.0f;
    p->n_grid = 1024;
    p->n_real = 1000;
    p->beta_mean = 13.0589f;
    p->beta_stddev = 0.0f;
    p->seed = 0x5EED0005ull;
    p->math_mode = MI_EDM_MATH_EXACT;
    p->mean_quirk = 1;   // the reference's averaging, as written (EventDrivenMap.cu:800-802,:817,:822); 0 = the true mean
    p->max_events = 1u << 20;
    p->real_offset = 0;
    p->dedup_identical = 0;
}

mi_status mi_edm_create(mi_ctx* ctx, const mi_edm_params* p, mi_edm** out)
{
    MI_REQUIRE(ctx, ctx && out, "mi_edm_create: NULL argument");
    *out = nullptr;
    mi_status st = validate(ctx, p);
    if (st != MI_OK) return st;
    MI_HIP(ctx, hipSetDevice(ctx->device));
    mi_edm* e = new (std::nothrow) mi_edm();
    if (!e) return mi::fail(ctx, MI_ERR_NOMEM, "mi_edm_create: out of host memory");
    e->ctx = ctx;
    e->device = ctx->device;
    e->p = *p;
    fill_model(e->p, &e->M);
    hipError_t err = hipSuccess;
    if (err == hipSuccess) err = hipMalloc(&e->d_v, kMaxGrid * sizeof(float));
    if (err == hipSuccess) err = hipMalloc(&e->d_s, kMaxGrid * sizeof(float));
    if (err == hipSuccess) err = hipMalloc(&e->d_w, kMaxGrid * sizeof(float));
    if (err == hipSuccess) err = hipMalloc(&e->d_result, kResultBytes);
    if (err == hipSuccess) err = hipMalloc(&e->d_one, kOneBytes);
    if (err == hipSuccess) err = hipMalloc(&e->d_aux, 2 * sizeof(unsigned));
    if (err == hipSuccess) err = hipHostMalloc(&e->h_result, kResultBytes);
    if (err == hipSuccess) err = hipHostMalloc(&e->h_aux, 2 * sizeof(unsigned));
    for (int i = 0; i < 4 && err == hipSuccess; ++i) err = hipEventCreate(&e->ev[i]);
    if (err != hipSuccess) {
        mi_edm_destroy(e);
        return mi::fail(ctx, MI_ERR_HIP, "mi_edm_create: allocation failed: %s", hipGetErrorString(err));
    }
    st = ensure_buffers(e);
    if (st != MI_OK) { mi_edm_destroy(e); return st; }
    *out = e;
    return MI_OK;
}

mi_status mi_edm_destroy(mi_edm* e)
{
    if (!e) return MI_OK;
    (void)hipSetDevice(e->device);
    // an evaluation begun and never collected still uses the buffers; the stream it was enqueued on is remembered here
    // (the context itself may already be gone: the Python layer closes handles in arbitrary order at interpreter exit)
    if (e->pending) (void)hipStreamSynchronize(e->pending_stream);
    free_real_buffers(e);
    void* bufs[] = {e->d_v, e->d_s, e->d_w, e->d_result, e->d_one, e->d_aux};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (e->h_result) (void)hipHostFree(e->h_result);
    if (e->h_aux) (void)hipHostFree(e->h_aux);
    for (int i = 0; i < 4; ++i)
        if (e->ev[i]) (void)hipEventDestroy(e->ev[i]);
    delete e;
    return MI_OK;
}

mi_status mi_edm_set_params(mi_edm* e, const mi_edm_params* p)
{
    MI_REQUIRE(nullptr, e != nullptr, "mi_edm_set_params: handle is NULL");
    mi_ctx* ctx = e->ctx;
    mi_status st = validate(ctx, p);
    if (st != MI_OK) return st;
    MI_HIP(ctx, hipSetDevice(ctx->device));
    const bool w_same = e->w_valid && p->n_grid == e->p.n_grid && p->L == e->p.L && p->a1 == e->p.a1 &&
                        p->a2 == e->p.a2 && p->b1 == e->p.b1 && p->b2 == e->p.b2 && p->math_mode == e->p.math_mode;
    MI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    e->pending = false;   // an uncollected evaluation is abandoned
    e->p = *p;
    fill_model(e->p, &e->M);
    e->w_valid = w_same;
    e->have_run = false;
    return ensure_buffers(e);
}

mi_status mi_edm_set_kernel_choice(mi_edm* e, int waves_per_realisation, int uniform_division)
{
    MI_REQUIRE(nullptr, e != nullptr, "mi_edm_set_kernel_choice: handle is NULL");
    MI_REQUIRE(e->ctx, waves_per_realisation == 0 || waves_per_realisation == 1 || waves_per_realisation == 4,
               "mi_edm_set_kernel_choice: waves_per_realisation must be 0 (automatic), 1 or 4");
    MI_REQUIRE(e->ctx, !e->pending, "mi_edm_set_kernel_choice: an evaluation is in flight");
    e->waves_per_real = waves_per_realisation;
    e->no_uniform_div = uniform_division == 0;
    return MI_OK;
}

mi_status mi_edm_compute_f_begin(mi_edm* e, const double* z)
{
    MI_REQUIRE(nullptr, e != nullptr, "mi_edm_compute_f_begin: handle is NULL");
    mi_ctx* ctx = e->ctx;
    MI_REQUIRE(ctx, z != nullptr, "mi_edm_compute_f_begin: NULL vector");
    MI_REQUIRE(ctx, !e->pending, "mi_edm_compute_f_begin: the previous evaluation has not been collected (call mi_edm_compute_f_end)");
    MI_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t S = e->p.n_spikes;
    // ZtoU (EventDrivenMap.cu:388-396) and the fp64 -> fp32 cast of :172
    double* U0 = e->pending_U0;
    U0[0] = z[0];
    U0[1] = 0.0;
    for (uint32_t i = 2; i <= S; ++i) U0[i] = z[i - 1];
    SpikeSeeds sd;
    memset(&sd, 0, sizeof(sd));
    for (uint32_t i = 0; i <= S; ++i) sd.U[i] = (float)U0[i];
    seed_indices(e->p, z, e->seed_ind);
    for (uint32_t m = 0; m < S; ++m) sd.ind[m] = e->seed_ind[m];
    e->last_sd = sd;
    // marked before the first launch: if run_pipeline fails half-way, kernels already enqueued still use the buffers,
    // and mi_edm_destroy / mi_edm_set_params synchronise only when they see work pending
    e->pending = true;
    e->pending_stream = ctx->stream;
    mi_status st = (e->p.math_mode == MI_EDM_MATH_FAST) ? run_pipeline<1>(e, sd) : run_pipeline<0>(e, sd);
    if (st != MI_OK) {
        (void)hipStreamSynchronize(ctx->stream);   // abandon the partial evaluation
        e->pending = false;
        return st;
    }
    return MI_OK;
}

mi_status mi_edm_compute_f_end(mi_edm* e, double* f, double* partial)
{
    MI_REQUIRE(nullptr, e != nullptr, "mi_edm_compute_f_end: handle is NULL");
    mi_ctx* ctx = e->ctx;
    MI_REQUIRE(ctx, f != nullptr, "mi_edm_compute_f_end: NULL vector");
    MI_REQUIRE(ctx, e->pending, "mi_edm_compute_f_end: no evaluation is in flight (call mi_edm_compute_f_begin)");
    MI_HIP(ctx, hipSetDevice(ctx->device));
    e->pending = false;
    MI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    e->have_run = true;
    const uint32_t S = e->p.n_spikes;
    const double* U0 = e->pending_U0;
    for (int i = 0; i < 3; ++i) (void)hipEventElapsedTime(&e->last_ms[i], e->ev[i], e->ev[i + 1]);
    (void)hipEventElapsedTime(&e->last_ms[3], e->ev[0], e->ev[3]);
    const float* mean = (const float*)e->h_result;
    const uint32_t count = *(const uint32_t*)(e->h_result + 32);
    const double* sums = (const double*)(e->h_result + 40);
    // host epilogue, EventDrivenMap.cu:237-239 (fp64)
    for (uint32_t m = 0; m < S; ++m) f[m] = (-U0[0] * U0[m + 1] - (double)mean[m]) + U0[0] * (double)e->p.time_horizon;
    (void)count;
    if (partial) {
        for (uint32_t m = 0; m < 2 * S + 1; ++m) partial[m] = sums[m];   // [sums | count | x0], MI_EDM_PARTIAL_LEN(S)
    }
    return MI_OK;
}

mi_status mi_edm_compute_f(mi_edm* e, const double* z, double* f, double* partial)
{
    MI_REQUIRE(nullptr, e != nullptr, "mi_edm_compute_f: handle is NULL");
    MI_REQUIRE(e->ctx, z && f, "mi_edm_compute_f: NULL vector");
    mi_status st = mi_edm_compute_f_begin(e, z);
    if (st != MI_OK) return st;
    return mi_edm_compute_f_end(e, f, partial);
}

mi_status mi_edm_residual_from_sums(const mi_edm_params* p, const double* z, const double* sc, double* f)
{
    if (!p || !z || !sc || !f) return mi::fail(nullptr, MI_ERR_INVALID_ARG, "mi_edm_residual_from_sums: NULL argument");
    const uint32_t S = p->n_spikes;
    if (S < 1 || S > (uint32_t)kMaxSpikes) return mi::fail(nullptr, MI_ERR_INVALID_ARG, "mi_edm_residual_from_sums: bad n_spikes");
    double U0[kMaxSpikes + 1];
    U0[0] = z[0];
    U0[1] = 0.0;
    for (uint32_t i = 2; i <= S; ++i) U0[i] = z[i - 1];
    const double count = sc[S];
    for (uint32_t m = 0; m < S; ++m) {
        // same rule and rounding as the single-device path (mean_stage2_kernel): realisation 0 re-enters the sum only
        // when exactly one realisation of the whole ensemble was accepted (EventDrivenMap.cu:800-802,:817), then
        // fp64 sum / count, rounded to fp32 once
        const double s = sc[m] + ((p->mean_quirk != 0 && count == 1.0) ? sc[S + 1 + m] : 0.0);
        const float mean = (float)(s / count);
        f[m] = (-U0[0] * U0[m + 1] - (double)mean) + U0[0] * (double)p->time_horizon;
    }
    return MI_OK;
}

mi_status mi_edm_debug_read(mi_edm* e, float* v, float* s, float* w, float* t0, uint16_t* i0, float* t1,
                            uint16_t* i1, uint32_t* accept, float* restricted, uint16_t* seed_ind)
{
    MI_REQUIRE(nullptr, e != nullptr, "mi_edm_debug_read: handle is NULL");
    mi_ctx* ctx = e->ctx;
    MI_REQUIRE(ctx, e->have_run, "mi_edm_debug_read: no ComputeF has run with the current parameters");
    MI_HIP(ctx, hipSetDevice(ctx->device));
    const size_t N = e->p.n_grid, SR = (size_t)e->p.n_spikes * e->p.n_real;
    if (restricted) {
        mi_status st = mi_restrict_f32_dev(ctx, e->d_t0, e->d_i0, e->d_t1, e->d_i1, e->p.time_horizon, e->p.L,
                                           e->p.n_grid, e->d_restricted, SR);
        if (st != MI_OK) return st;
    }
    MI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (v) MI_HIP(ctx, hipMemcpy(v, e->d_v, N * sizeof(float), hipMemcpyDeviceToHost));
    if (s) MI_HIP(ctx, hipMemcpy(s, e->d_s, N * sizeof(float), hipMemcpyDeviceToHost));
    if (w) MI_HIP(ctx, hipMemcpy(w, e->d_w, N * sizeof(float), hipMemcpyDeviceToHost));
    if (t0) MI_HIP(ctx, hipMemcpy(t0, e->d_t0, SR * sizeof(float), hipMemcpyDeviceToHost));
    if (i0) MI_HIP(ctx, hipMemcpy(i0, e->d_i0, SR * sizeof(uint16_t), hipMemcpyDeviceToHost));
    if (t1) MI_HIP(ctx, hipMemcpy(t1, e->d_t1, SR * sizeof(float), hipMemcpyDeviceToHost));
    if (i1) MI_HIP(ctx, hipMemcpy(i1, e->d_i1, SR * sizeof(uint16_t), hipMemcpyDeviceToHost));
    if (accept) MI_HIP(ctx, hipMemcpy(accept, e->d_accept, (size_t)e->p.n_real * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (restricted) MI_HIP(ctx, hipMemcpy(restricted, e->d_restricted, SR * sizeof(float), hipMemcpyDeviceToHost));
    if (seed_ind) memcpy(seed_ind, e->seed_ind, e->p.n_spikes * sizeof(uint16_t));
    return MI_OK;
}

// Decision-coverage taps (the device-side mirror of oracle/edm_oracle.c's orc_edm_counters): Evolve of the LAST ComputeF
// is run again by the tapped instantiation of the wave-per-realisation kernel (same arithmetic, same outputs, plus
// counters).  The product launch carries none of this.
mi_status mi_edm_debug_counters(mi_edm* e, uint64_t out[MI_EDM_N_COUNTERS])
{
    MI_REQUIRE(nullptr, e != nullptr, "mi_edm_debug_counters: handle is NULL");
    mi_ctx* ctx = e->ctx;
    MI_REQUIRE(ctx, out != nullptr, "mi_edm_debug_counters: NULL argument");
    MI_REQUIRE(ctx, e->have_run && !e->pending, "mi_edm_debug_counters: no ComputeF has completed with the current parameters");
    MI_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long* d_taps = nullptr;
    MI_HIP(ctx, hipMalloc(&d_taps, kTapCount * sizeof(unsigned long long)));
    mi_status st = MI_OK;
    hipError_t err = hipMemsetAsync(d_taps, 0, kTapCount * sizeof(unsigned long long), ctx->stream);
    unsigned live = 0;
    if (err == hipSuccess) err = hipMemcpyAsync(&live, e->d_aux, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
    if (err == hipSuccess) {
        live &= kSliceMask;
        const bool hetero = e->p.beta_stddev != 0.0f, three = e->p.n_spikes <= 3;
        const size_t lds_bytes = evolve_lds_bytes(hetero, live);
        const unsigned blocks = (e->p.n_real + 3) / 4;
        const bool whole_warps = (e->p.n_grid & 31u) == 0u;
#define MI_TAPPED(MATH, H, NS)                                                                                              \
    do {                                                                                                                    \
        if (whole_warps)                                                                                                    \
            hipLaunchKernelGGL((evolve_kernel<MATH, H, NS, false, true, true>), dim3(blocks), dim3(kEvolveBlock), lds_bytes, ctx->stream, \
                               e->M, e->last_sd, live, d_taps, e->d_v, e->d_s, e->d_w, e->d_t0, e->d_i0, e->d_t1, e->d_i1, e->d_accept); \
        else                                                                                                                \
            hipLaunchKernelGGL((evolve_kernel<MATH, H, NS, false, true, false>), dim3(blocks), dim3(kEvolveBlock), lds_bytes, ctx->stream, \
                               e->M, e->last_sd, live, d_taps, e->d_v, e->d_s, e->d_w, e->d_t0, e->d_i0, e->d_t1, e->d_i1, e->d_accept); \
    } while (0)
        if (e->p.math_mode == MI_EDM_MATH_FAST) {
            if (hetero) { if (three) MI_TAPPED(1, true, 3); else MI_TAPPED(1, true, kMaxSpikes); }
            else { if (three) MI_TAPPED(1, false, 3); else MI_TAPPED(1, false, kMaxSpikes); }
        } else {
            if (hetero) { if (three) MI_TAPPED(0, true, 3); else MI_TAPPED(0, true, kMaxSpikes); }
            else { if (three) MI_TAPPED(0, false, 3); else MI_TAPPED(0, false, kMaxSpikes); }
        }
#undef MI_TAPPED
        err = hipGetLastError();
    }
    unsigned long long h[kTapCount] = {0};
    if (err == hipSuccess) err = hipMemcpyAsync(h, d_taps, sizeof(h), hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_taps);
    if (err != hipSuccess) st = mi::fail(ctx, MI_ERR_HIP, "mi_edm_debug_counters failed: %s", hipGetErrorString(err));
    static_assert(kTapCount == MI_EDM_N_COUNTERS, "tap layout and ABI constant disagree");
    for (int i = 0; i < kTapCount; ++i) out[i] = (uint64_t)h[i];
    return st;
}

mi_status mi_edm_last_timings(mi_edm* e, float ms[4])
{
    MI_REQUIRE(nullptr, e && ms, "mi_edm_last_timings: NULL argument");
    for (int i = 0; i < 4; ++i) ms[i] = e->last_ms[i];
    return MI_OK;
}

// test hook: run the device math routines on arrays (op: 0 exp, 1 log, 2 pow, 3 erfinv, 4 a / b[0] as the kernels
// divide by a wave-uniform divisor, 5 a / b[0] as IEEE division, 6-11 the firing test will_fire(a, b) without / with its
// hardware pre-decision for beta = 13.0589, 1.5, 0.7)
mi_status mi_edm_math_probe(mi_ctx* ctx, int math_mode, int op, const float* a_dev, const float* b_dev,
                            float* out_dev, size_t n)
{
    MI_REQUIRE(ctx, ctx && a_dev && out_dev, "mi_edm_math_probe: NULL argument");
    if (n == 0) return MI_OK;
    MI_REQUIRE(ctx, op != 12 || n % 64 == 0, "mi_edm_math_probe: op 12 exchanges lane pairs and needs whole waves (n %% 64 == 0)");
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (math_mode == MI_EDM_MATH_FAST)
        hipLaunchKernelGGL((math_probe_kernel<1>), dim3(grid), dim3(256), 0, ctx->stream, op, a_dev, b_dev ? b_dev : a_dev, out_dev, n);
    else
        hipLaunchKernelGGL((math_probe_kernel<0>), dim3(grid), dim3(256), 0, ctx->stream, op, a_dev, b_dev ? b_dev : a_dev, out_dev, n);
    MI_LAUNCH_CHECK(ctx, "math probe kernel");
    return MI_OK;
}

}  // extern "C"

