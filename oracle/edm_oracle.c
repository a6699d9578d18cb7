/*
 * oracle/edm_oracle.c -- CPU ORACLE (test infrastructure, NOT the product) for
 * the EventDrivenMap residual: lift -> evolve -> restrict -> average.
 *
 * Restates, in plain scalar C, the fp32 device pipeline of the reference:
 *   BuildCouplingKernel/circshift  EventDrivenMap.cu:111-129, :826-841
 *   initialSpikeInd                EventDrivenMap.cu:361-376
 *   ZtoU                           EventDrivenMap.cu:388-396
 *   LiftKernel                     EventDrivenMap.cu:505-542
 *   fun/dfun/eventTime             EventDrivenMap.cu:544-573
 *   EvolveKernel (+blockReduceMin) EventDrivenMap.cu:575-674, :843-881
 *   RestrictKernel                 EventDrivenMap.cu:769-785   (interp_oracle.c)
 *   Count + mean                   EventDrivenMap.cu:787-824   (interp_oracle.c)
 *   host epilogue                  EventDrivenMap.cu:233-239
 *
 * PARITY STATUS: "parity unpinned".  The reference cannot be built or run here
 * (needs <armadillo>, curand.h, nvcc and an NVIDIA GPU; counterMax at
 * EventDrivenMap.cu:564 is undefined in its own parameters.hpp) and holds no
 * golden output.  Bit parity with the original is unreachable in principle
 * (CUDA's expf/powf and nvcc's FMA contraction are not reproducible here), so
 * this file DEFINES the arithmetic that the HIP path must match bit for bit in
 * MI_EDM_MATH_EXACT mode:
 *   - every fp32 operation rounds separately (-ffp-contract=off); the only
 *     fused operations are the fmaf() calls written out below;
 *   - exp/log/pow are the software routines in this file (Cephes-style
 *     polynomials, <= ~1 ulp), not libm;
 *   - the decisions where the reference is undefined or racy are listed in
 *     DESIGN.md "EventDrivenMap: documented decisions" and marked [D1]..[D8];
 *     orc_edm_compute_f_counted counts how often an evaluation reaches one of
 *     them (tests/test_edm_oracle_cpu.py: never, on the BASELINE inputs).
 * Two SENSITIVITY builds of this same file exist for tests only (oracle/Makefile):
 * liboracle_contract.so (-ffp-contract=fast -mfma: the compiler fuses a*b+c as
 * nvcc -fmad=true would, reference Makefile:3) and liboracle_libm.so
 * (-DORC_EDM_LIBM_MATH: libm expf/logf/powf instead of the routines below, a
 * stand-in for "some other <= 2 ulp implementation" such as CUDA's libdevice).
 * They bound how far the unreproducible parts of the reference's arithmetic can
 * move f; they are never what the HIP path is compared with.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* same field order as mi_edm_params in include/mi355_interp.h (restated) */
typedef struct orc_edm_params {
    float vth, a1, a2, b1, b2, I, L;
    double newton_tol;
    uint32_t newton_max_iter;
    uint32_t n_spikes;
    float time_horizon;
    uint32_t n_grid;
    uint32_t n_real;
    float beta_mean;
    float beta_stddev;
    uint64_t seed;
    int math_mode;
    int mean_quirk;
    uint32_t max_events;
    uint32_t real_offset;
} orc_edm_params;

#define ORC_MAX_SPIKES 8
#define ORC_MAX_GRID 1024

/* Optional decision-coverage counters (tests/test_edm_oracle_cpu.py): how often an evaluation reaches one of the points
 * where this restatement cannot follow the reference ([D0], [D1], [D2], [D5], [D8] of DESIGN.md) -- so that a test can
 * state that none of them is exercised on the BASELINE inputs -- plus plain workload counts.  All fields are totals over
 * the realisations of one orc_edm_compute_f_counted call, except the two maxima. */
typedef struct orc_edm_counters {
    uint64_t realisations, accepted, events;
    uint64_t argmin_ties;            /* [D1] events whose minimal firing time (a real one, < "never") is shared by >= 2 neurons */
    uint64_t no_firing_events;       /* [D1] events whose minimal time is the "never" value 100 (no neuron will fire)          */
    uint64_t nan_times;              /* [D1] candidate firing times that were NaN                                                */
    uint64_t argmin_tree_mismatch;   /* [D1] events at which the LITERAL reference reduction (orc_edm_argmin_reference_tree:
                                      *      32-wide shuffle tree + padded second stage, :843-881) picks another (time, index)
                                      *      than this file's rule; counted only when n_grid is a multiple of 32             */
    uint64_t unwritten_last_slots;   /* [D2] bumps of ACCEPTED realisations that never recorded a pre-T event (t0/i0 unset)   */
    uint64_t unwritten_last_slots_all; /*    the same over all realisations                                                     */
    uint64_t seed_scans_empty;       /* [D5] seed scans (one per bump m >= 1) that found no grid point                          */
    uint64_t newton_cap_hits;        /* [D0] firing-time solves that stopped at newton_max_iter                                 */
    uint64_t event_cap_hits;         /* [D8] realisations that left the event loop at max_events                                */
    uint64_t time_cap_exits;         /*      realisations that left it at t >= 2T without every bump crossing (:601)          */
    uint64_t newton_solves, newton_iters;
    uint64_t wave64_rounds;          /* sum over events of max over l of #{firing neurons i : i mod 64 == l} (the HIP kernel's rounds) */
    uint32_t max_newton_iter;        /* largest iteration count of any solve                                                    */
    uint32_t max_events_one;         /* largest event count of any realisation                                                  */
} orc_edm_counters;

void orc_restrict_f32(const float* t0, const uint16_t* i0, const float* t1, const uint16_t* i1,
                      float T, float L, uint32_t ngrid, float* out, size_t n);
void orc_masked_mean_f32(const float* x, const uint32_t* accept, size_t nreal, size_t nspikes,
                         int quirk, float* mean, uint32_t* count_out);

/* ---- deterministic fp32 math ------------------------------------------- */

/* exp(x): n = rint(x*log2e); r = x - n*ln2 (two-step fmaf); degree-5 polynomial
 * in r for (exp(r)-1-r)/r^2; scale with ldexpf. */
float orc_edm_expf(float x)
{
#ifdef ORC_EDM_LIBM_MATH   /* sensitivity probe only (oracle/Makefile: liboracle_libm.so), never the parity oracle */
    return expf(x);
#endif
    if (x != x) return x;
    if (x > 88.72283935546875f) return INFINITY;
    if (x < -103.97208404541015625f) return 0.0f;
    const float n = rintf(x * 0x1.715476p+0f);
    float r = fmaf(n, -0x1.62e4p-1f, x);           /* ln2 high part: 0.693145751953125 */
    r = fmaf(n, -0x1.7f7d1cp-20f, r);              /* ln2 low part: 1.42860677e-06     */
    const float z = r * r;
    float p = 0x1.a0d2bcp-13f;                     /* 1.9875691500e-4 */
    p = fmaf(p, r, 0x1.6e8716p-10f);               /* 1.3981999507e-3 */
    p = fmaf(p, r, 0x1.1112ep-7f);                 /* 8.3334519073e-3 */
    p = fmaf(p, r, 0x1.5554ep-5f);                 /* 4.1665795894e-2 */
    p = fmaf(p, r, 0x1.555554p-3f);                /* 1.6666665459e-1 */
    p = fmaf(p, r, 0x1.000002p-1f);                /* 5.0000001201e-1 */
    p = fmaf(p, z, r);
    p = p + 1.0f;
    return ldexpf(p, (int)n);
}

/* log(x): x = m*2^e with m in [sqrt(1/2), sqrt(2)); degree-8 polynomial in m-1. */
float orc_edm_logf(float x)
{
#ifdef ORC_EDM_LIBM_MATH
    return logf(x);
#endif
    if (x != x) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    int e;
    float m = frexpf(x, &e);                       /* m in [0.5, 1) */
    if (m < 0x1.6a09e6p-1f) { m = m + m; e -= 1; } /* sqrt(1/2) */
    const float f = m - 1.0f;
    const float z = f * f;
    float p = 0x1.203736p-4f;                      /*  7.0376836292e-2 */
    p = fmaf(p, f, -0x1.d7a37p-4f);                /* -1.1514610310e-1 */
    p = fmaf(p, f, 0x1.de4a34p-4f);                /*  1.1676998740e-1 */
    p = fmaf(p, f, -0x1.fcba9ap-4f);               /* -1.2420140846e-1 */
    p = fmaf(p, f, 0x1.23d37ep-3f);                /*  1.4249322787e-1 */
    p = fmaf(p, f, -0x1.555ca2p-3f);               /* -1.6668057665e-1 */
    p = fmaf(p, f, 0x1.999a2ep-3f);                /*  2.0000714765e-1 */
    p = fmaf(p, f, -0x1.fffffep-3f);               /* -2.4999993993e-1 */
    p = fmaf(p, f, 0x1.555554p-2f);                /*  3.3333331174e-1 */
    const float fe = (float)e;
    float y = (p * f) * z;
    y = fmaf(fe, -0x1.bd0106p-13f, y);             /* -2.12194440e-4 */
    y = fmaf(-0.5f, z, y);
    float r = f + y;
    r = fmaf(fe, 0x1.63p-1f, r);                   /* 0.693359375 */
    return r;
}

/* pow(a, b) for the firing test (a = s/(vth-I), b = 1/beta): exp(b*log(a)).
 * a < 0 -> NaN (as C pow for non-integer b), a == 0 -> exp(-inf*b). */
float orc_edm_powf(float a, float b)
{
#ifdef ORC_EDM_LIBM_MATH
    return powf(a, b);      /* the reference calls pow() itself (:559), not exp(b log a) */
#endif
    return orc_edm_expf(b * orc_edm_logf(a));
}

/* ---- per-neuron parameter heterogeneity ----------------------------------
 * The reference draws beta[r][i] ~ N(p0, sigma) with cuRAND XORWOW seeded from
 * clock() (EventDrivenMap.cu:103-104,179): reproducible only for sigma = 0.
 * [D6] Here z(r,i) is a counter-based draw: SplitMix64 hash of (seed, r*N+i)
 * -> u in (0,1) -> z = sqrt(2)*erfinv(2u-1) (Giles' single-precision erfinv),
 * beta = fmaf(sigma, z, p0); sigma = 0 gives beta = p0 exactly. */
static inline uint64_t orc_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

float orc_edm_erfinvf(float x)
{
    float w = -orc_edm_logf((1.0f - x) * (1.0f + x));
    float p;
    if (w < 5.0f) {
        w = w - 2.5f;
        p = 2.81022636e-08f;
        p = fmaf(p, w, 3.43273939e-07f);
        p = fmaf(p, w, -3.5233877e-06f);
        p = fmaf(p, w, -4.39150654e-06f);
        p = fmaf(p, w, 0.00021858087f);
        p = fmaf(p, w, -0.00125372503f);
        p = fmaf(p, w, -0.00417768164f);
        p = fmaf(p, w, 0.246640727f);
        p = fmaf(p, w, 1.50140941f);
    } else {
        w = sqrtf(w) - 3.0f;
        p = -0.000200214257f;
        p = fmaf(p, w, 0.000100950558f);
        p = fmaf(p, w, 0.00134934322f);
        p = fmaf(p, w, -0.00367342844f);
        p = fmaf(p, w, 0.00573950773f);
        p = fmaf(p, w, -0.0076224613f);
        p = fmaf(p, w, 0.00943887047f);
        p = fmaf(p, w, 1.00167406f);
        p = fmaf(p, w, 2.83297682f);
    }
    return p * x;
}

float orc_edm_beta(const orc_edm_params* P, uint32_t r, uint32_t i)
{
    if (P->beta_stddev == 0.0f) return P->beta_mean;
    const uint64_t ctr = ((uint64_t)r + P->real_offset) * P->n_grid + i;
    const uint64_t h = orc_mix64(P->seed + 0x9E3779B97F4A7C15ull * (ctr + 1));
    const float u = ((float)(uint32_t)(h >> 40) + 0.5f) * 0x1.0p-24f;      /* (0,1), 24 bits */
    const float z = 0x1.6a09e6p+0f * orc_edm_erfinvf(fmaf(2.0f, u, -1.0f)); /* sqrt(2)*erfinv */
    return fmaf(P->beta_stddev, z, P->beta_mean);
}

/* ---- BuildCouplingKernel + circshift (EventDrivenMap.cu:111-129,:826-841) -- */
void orc_edm_coupling(const orc_edm_params* P, float* w)
{
    const uint32_t N = P->n_grid;
    float tmp[ORC_MAX_GRID];
    const float h = (2.0f * P->L) / (float)N;
    for (uint32_t i = 0; i < N; ++i) {
        const float x = -P->L + h * (float)i;
        const float ax = fabsf(x);
        const float k = P->a1 * orc_edm_expf(-P->b1 * ax) - P->a2 * orc_edm_expf(-P->b2 * ax);
        tmp[i] = ((k * 2.0f) * P->L) / (float)N;
    }
    const uint32_t shift = N / 2;
    for (uint32_t i = 0; i < N; ++i) w[i] = tmp[(i + shift) % N];
}

/* ---- initialSpikeInd (EventDrivenMap.cu:361-372) ----------------------------
 * ind[0] = N/2; for m >= 1 scan i = ind[m-1] .. 1 downwards for the first grid
 * point left of -Z[0]*Z[m] (comparison in double: Z is an arma::vec).
 * [D5] When no point qualifies the reference leaves the entry stale (whatever
 * the previous call, or malloc, left there); here `ind` is in/out so the caller
 * carries the previous value, initially 0. */
static void seed_indices_impl(const orc_edm_params* P, const double* Z, uint16_t* ind, orc_edm_counters* C)
{
    const uint32_t N = P->n_grid, S = P->n_spikes;
    ind[0] = (uint16_t)(N / 2);
    for (uint32_t m = 1; m < S; ++m) {
        int found = 0;
        for (uint32_t i = ind[m - 1]; i > 0; --i) {
            const float xi = -P->L + ((float)(2u * i) * P->L) / (float)N;
            if ((double)xi < -Z[0] * Z[m]) { ind[m] = (uint16_t)i; found = 1; break; }
        }
        if (C && !found) C->seed_scans_empty += 1;
    }
}

void orc_edm_seed_indices(const orc_edm_params* P, const double* Z, uint16_t* ind) { seed_indices_impl(P, Z, ind, NULL); }

/* ---- LiftKernel (EventDrivenMap.cu:505-542) ----------------------------------
 * U = (c, 0, Z1, .., Z_{S-1}) in fp32.  Abscissa descends: x = L - (2L/N)*i.
 * Both branch expressions are always evaluated and multiplied by 0/1 flags, as
 * in the reference, so an overflowing exponential in the unselected branch
 * still poisons the result with NaN (0*inf).  The profile does not depend on
 * the realisation, so it is computed once (v[N], s[N]).
 * The six terms of dummyV and the four of dummyS are added left to right, as
 * written at :522-527 / :532-534; exp(((c*U)/c)*(1-beta)) of the a1 term (:523)
 * and exp((U)*(1-beta)) of the a2 term (:524) are both kept as written. */
void orc_edm_lift(const orc_edm_params* P, const float* U, float* v, float* s)
{
    const uint32_t N = P->n_grid, S = P->n_spikes;
    const float c = U[0], beta = P->beta_mean;
    const float a[2] = {P->a1, P->a2}, b[2] = {P->b1, P->b2};
    const float h = (2.0f * P->L) / (float)N;
    const float omb = 1.0f - beta;
    for (uint32_t i = 0; i < N; ++i) {
        const float x = P->L - h * (float)i;
        const float xc = x / c;
        float sv = 0.0f, ss = 0.0f;
        for (uint32_t m = 1; m <= S; ++m) {
            /* every product, quotient and sum below is grouped as C groups the reference's expression (:522-534):
             * left to right, `-b1*c*U` = ((-b1)*c)*U, `a1*beta*c/(..)` = ((a1*beta)*c)/(..) */
            const float Um = U[m];
            const float cu = c * Um;
            const float d = x - cu;
            const float pos = (d > 0.0f) ? 1.0f : 0.0f;
            const float neg = (d <= 0.0f) ? 1.0f : 0.0f;
            const float ebu = orc_edm_expf(beta * Um);
            const float exo = orc_edm_expf(xc * omb);
            const float dxk[2] = {exo - orc_edm_expf((cu / c) * omb),    /* a1 term, :523: exp(((c*U)/c)*(1-beta)) */
                                  exo - orc_edm_expf(Um * omb)};         /* a2 term, :524: exp((U)*(1-beta)) */
            const float ebc = orc_edm_expf(-(beta / c) * d);
            float P[2], Q[2], R[2], B[2], sA[2], sB1[2], sB2[2];
            for (int k = 0; k < 2; ++k) {
                const float cb = c * b[k];
                const float abc = (a[k] * beta) * c;
                const float ep = (1.0f + cb) / c;
                const float em = (1.0f - cb) / c;
                const float Pk = abc / ((beta + cb) * (1.0f + cb));
                const float en = orc_edm_expf(-(cb * Um));               /* exp(-b*c*U) = exp(((-b)*c)*U) */
                P[k] = (Pk * orc_edm_expf(cu * ep)) * en;
                Q[k] = (((abc / omb) * ebu) * (1.0f / (beta + cb) + 1.0f / (cb - beta))) * dxk[k];
                R[k] = ((abc / ((cb - beta) * (1.0f - cb))) * orc_edm_expf(cb * Um)) * (orc_edm_expf(x * em) - orc_edm_expf(cu * em));
                B[k] = (Pk * orc_edm_expf(x * ep)) * en;
                /* synaptic profile, :532-534 */
                sA[k] = ((beta * a[k]) * (c / (beta + cb))) * orc_edm_expf(b[k] * d);
                sB1[k] = (((2.0f * a[k]) / b[k]) * (beta / (1.0f - (beta * beta) / (((c * c) * b[k]) * b[k])))) * ebc;
                sB2[k] = ((beta * a[k]) * (c / (cb - beta))) * orc_edm_expf(b[k] * (cu - x));
            }
            /* :522-527: P1 - P2 + Q1 - R1 - Q2 + R2, added left to right */
            const float brA = ((((P[0] - P[1]) + Q[0]) - R[0]) - Q[1]) + R[1];
            const float brB = B[0] - B[1];
            const float dummyV = (pos * brA + neg * brB) * orc_edm_expf(-xc);
            sv = sv + ((dummyV - pos * orc_edm_expf(-d / c)) + neg * 0.0f);             /* :530 */
            /* (cu - x) > 0  <=>  d < 0 ; (cu - x) <= 0  <=>  d >= 0 */
            const float e = cu - x;
            const float posS = (e > 0.0f) ? 1.0f : 0.0f;
            const float negS = (e <= 0.0f) ? 1.0f : 0.0f;
            ss = ss + (posS * (sA[0] - sA[1]) + negS * (((sB1[0] - sB2[0]) - sB1[1]) + sB2[1]));
        }
        float vv = P->I + sv;
        vv = vv * ((vv < 1.0f) ? 1.0f : 0.0f);
        v[i] = vv;
        s[i] = ss;
    }
}

/* ---- fun / dfun / eventTime (EventDrivenMap.cu:544-573) -------------------- */
typedef struct { float f, df; } orc_fdf;

static inline orc_fdf orc_fun_dfun(const orc_edm_params* P, float t, float v, float s, float beta)
{
    const float e1 = orc_edm_expf(-t);
    const float e2 = orc_edm_expf((1.0f - beta) * t);
    const float se = s * e1;
    orc_fdf r;
    /* :546  v e^-t + I(1-e^-t) + s e^-t/(1-beta) (e^{(1-beta)t} - 1) - vth */
    r.f = ((v * e1 + P->I * (1.0f - e1)) + (se / (1.0f - beta)) * (e2 - 1.0f)) - P->vth;
    /* :551  I e^-t - v e^-t + s e^-t e^{-(beta-1)t} + s e^-t (e^{-(beta-1)t} - 1)/(beta-1) */
    r.df = ((P->I * e1 - v * e1) + se * e2) + (se * (e2 - 1.0f)) / (beta - 1.0f);
    return r;
}

/* iters (optional): number of Newton iterations taken; -1 when the neuron will never fire (decision false) */
static float event_time_impl(const orc_edm_params* P, float v0, float s0, float beta, int* iters)
{
    const float gap = P->vth - P->I;
    const float ratio = s0 / gap;
    const float pw = orc_edm_powf(ratio, 1.0f / beta);
    /* :559 */
    const float thr = (P->vth * pw + P->I * (1.0f - pw)) - (gap / (beta - 1.0f)) * (ratio - pw);
    const int decision = (v0 > thr) ? 1 : 0;
    float t = 0.0f;
    orc_fdf r = orc_fun_dfun(P, t, v0, s0, beta);
    float f = r.f * (float)decision, df = r.df;
    uint32_t counter = 0;
    while (((double)fabsf(f) > P->newton_tol) && (counter < P->newton_max_iter)) {
        t = t - f / df;
        r = orc_fun_dfun(P, t, v0, s0, beta);
        f = r.f;
        df = r.df;
        ++counter;
    }
    if (iters) *iters = decision ? (int)counter : -1;
    return fabsf(t) + 100.0f * (1.0f - (float)decision);
}

float orc_edm_event_time(const orc_edm_params* P, float v0, float s0, float beta) { return event_time_impl(P, v0, s0, beta, NULL); }

/* ---- blockReduceMin / warpReduceMin, literally (EventDrivenMap.cu:843-881) ----
 * What the reference's own reduction returns on a 32-wide-warp machine for a block of n = blockDim.x threads holding
 * (time[i], index i): `__shfl_down` trees in which a lane keeps its own pair only when its time is strictly smaller
 * (:849-850; on a tie the HIGHER lane's index moves down, a NaN coming from above displaces a number, a NaN already held
 * is displaced by anything), then a second tree over the per-warp results padded with (100.0f, 0) in the lanes
 * >= n/32 (:867-868).  `__shfl_down` past the end of the warp returns the lane's own value.  Only meaningful for
 * n a multiple of 32 (the reference launches 1024 or 512 threads); used by the decision-coverage counters, not by the
 * oracle's own event loop (see [D1]). */
static void warp_reduce_min_literal(float* t, uint32_t* ix)
{
    for (int offset = 16; offset > 0; offset /= 2) {
        float nt[32];
        uint32_t ni[32];
        for (int l = 0; l < 32; ++l) {
            const int src = (l + offset < 32) ? l + offset : l;
            const float dummy_t = t[src];
            const uint32_t dummy_i = ix[src];
            const float vt = (t[l] < dummy_t) ? t[l] : dummy_t;      /* :849 */
            nt[l] = vt;
            ni[l] = (vt < dummy_t) ? ix[l] : dummy_i;                /* :850, tested AFTER the update of val.time */
        }
        memcpy(t, nt, sizeof(nt));
        memcpy(ix, ni, sizeof(ni));
    }
}

void orc_edm_argmin_reference_tree(const float* time, uint32_t n, float* best, uint32_t* index)
{
    float st[32];
    uint32_t si[32];
    const uint32_t nwarps = n / 32u;                                 /* blockDim.x / warpSize, :866 */
    for (uint32_t l = 0; l < 32u; ++l) { st[l] = 100.0f; si[l] = 0u; }   /* :867-868 */
    for (uint32_t w = 0; w < nwarps && w < 32u; ++w) {
        float t[32];
        uint32_t ix[32];
        for (uint32_t l = 0; l < 32u; ++l) { t[l] = time[w * 32u + l]; ix[l] = w * 32u + l; }
        warp_reduce_min_literal(t, ix);
        st[w] = t[0];
        si[w] = ix[0];
    }
    warp_reduce_min_literal(st, si);
    *best = st[0];
    *index = si[0];
}

/* ---- the block arg-min the event loop uses ------------------------------------
 * The reference's reduction is deterministic on 32-wide warps, ties included, so it can be followed exactly whenever
 * the block is made of whole warps (n a multiple of 32, as with the reference's own 1024 and 512 threads).  Reading
 * :843-881: every step of a shuffle tree keeps the lower lane's pair only when its time is STRICTLY smaller, i.e. a tie
 * goes to the higher lane of the step; over the steps 16, 8, 4, 2, 1 that makes the winner among tied lanes the one
 * whose 5-bit lane number, read BACKWARDS, is largest.  The second tree does the same to the warp numbers, with the
 * padding pairs (100.0f, 0) in the lanes >= n/32 taking part (:867-868): when no neuron fires before the "never" time
 * 100.0f and the block has fewer than 32 warps, padding lane 31 wins and the result is (100.0f, 0).
 *   =>  winner = the neuron of minimal time with the largest key  rev5(i >> 5) * 32 + rev5(i & 31);
 *       if n/32 < 32 and that minimal time is not below 100.0f: (100.0f, index 0).
 * orc_edm_argmin_reference_tree (the literal emulation above) returns the same pair on every input without NaN
 * (tests/test_edm_oracle_cpu.py compares them on tie-laden random blocks).
 * [D1] What is left of the decision: a NaN time never wins (taken as +inf; the reference's tree lets a NaN from the
 * higher lane displace a number and be displaced in turn), and a block that is not made of whole warps (the reference
 * cannot run one meaningfully: lanes of the last warp would read exited threads, :866 drops the remainder) uses
 * "smallest time, ties to the lowest index". */
static inline uint32_t orc_rev5(uint32_t x)
{
    return ((x & 1u) << 4) | ((x & 2u) << 2) | (x & 4u) | ((x & 8u) >> 2) | ((x & 16u) >> 4);
}

void orc_edm_argmin(const float* time, uint32_t n, float* best, uint32_t* index)
{
    float tmin = INFINITY;
    uint32_t imin = 0;
    const int whole_warps = (n % 32u == 0u) && n >= 32u;
    uint32_t kmax = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const float t = (time[i] != time[i]) ? INFINITY : time[i];
        const uint32_t key = orc_rev5(i >> 5) * 32u + orc_rev5(i & 31u);
        if (t < tmin || i == 0u) { tmin = t; imin = i; kmax = key; }
        else if (whole_warps && t == tmin && key > kmax) { imin = i; kmax = key; }
    }
    if (whole_warps && n / 32u < 32u && !(tmin < 100.0f)) { tmin = 100.0f; imin = 0; }
    *best = tmin;
    *index = imin;
}

/* ---- EvolveKernel for one realisation (EventDrivenMap.cu:575-674) -----------
 * [D1] arg-min over neurons: orc_edm_argmin above -- the reference's own result, ties
 *      and padding lanes included, for blocks of whole warps; a NaN time never wins.
 * [D2] the per-bump event slots start as time 0 / index 0 (the reference
 *      leaves them uninitialised in shared memory, :580-583).
 * [D3] the bump-assignment rule `minIndex += (d_i < d_minIndex)` (:625-629) is
 *      reproduced as written (it is not a true arg-min for S >= 3).
 */
static void evolve_one_impl(const orc_edm_params* P, const float* v0, const float* s0, const float* w,
                            const uint16_t* seed_ind, uint32_t r,
                            float* last_t, uint16_t* last_i, float* cross_t, uint16_t* cross_i,
                            uint32_t* accept, uint32_t* n_events, orc_edm_counters* C)
{
    const uint32_t N = P->n_grid, S = P->n_spikes;
    const float T = P->time_horizon;
    float v[ORC_MAX_GRID], s[ORC_MAX_GRID], beta[ORC_MAX_GRID];
    for (uint32_t i = 0; i < N; ++i) { v[i] = v0[i]; s[i] = s0[i]; beta[i] = orc_edm_beta(P, r, i); }
    uint16_t li[ORC_MAX_SPIKES], ci[ORC_MAX_SPIKES];
    float lt[ORC_MAX_SPIKES], ct[ORC_MAX_SPIKES];
    for (uint32_t m = 0; m < S; ++m) { li[m] = seed_ind[m]; ci[m] = 0; lt[m] = 0.0f; ct[m] = 0.0f; }
    const uint32_t full = (1u << S) - 1u;
    uint32_t crossed = 0, events = 0, last_written = 0;
    float now = 0.0f;
    /* hard event cap (same rule as the HIP kernel): termination even when time cannot advance */
    while (crossed < full && now < 2.0f * T && events < P->max_events) {
        float best;
        uint32_t idx;
        uint32_t n_at_best = 0;
        uint8_t per_lane[64] = {0};
        float taus[ORC_MAX_GRID];
        for (uint32_t i = 0; i < N; ++i) {
            int it = -1;
            const float tau = event_time_impl(P, v[i], s[i], beta[i], C ? &it : NULL);
            taus[i] = tau;
            if (C) {
                if (tau != tau) C->nan_times += 1;
                if (it >= 0) {
                    C->newton_solves += 1;
                    C->newton_iters += (uint64_t)it;
                    if ((uint32_t)it > C->max_newton_iter) C->max_newton_iter = (uint32_t)it;
                    if ((uint32_t)it >= P->newton_max_iter) C->newton_cap_hits += 1;
                    per_lane[i & 63u] += 1;
                }
            }
        }
        orc_edm_argmin(taus, N, &best, &idx);
        if (C) {
            for (uint32_t i = 0; i < N; ++i) n_at_best += (taus[i] == best) ? 1u : 0u;
            if (best >= 100.0f) C->no_firing_events += 1;
            else if (n_at_best > 1) C->argmin_ties += 1;
            if (N % 32u == 0u) {
                float tb;
                uint32_t ti;
                orc_edm_argmin_reference_tree(taus, N, &tb, &ti);
                if (!(tb == best) || ti != idx) C->argmin_tree_mismatch += 1;
            }
            uint8_t rounds = 0;
            for (int l = 0; l < 64; ++l) if (per_lane[l] > rounds) rounds = per_lane[l];
            C->wave64_rounds += rounds;
        }
        const float dt = best;
        const float e1 = orc_edm_expf(-dt);
        for (uint32_t i = 0; i < N; ++i) {
            const float e2 = orc_edm_expf((1.0f - beta[i]) * dt);
            float vv = v[i] * e1;
            vv = vv + (P->I * (1.0f - e1) + ((s[i] * e1) / (1.0f - beta[i])) * (e2 - 1.0f));
            vv = vv * ((i != idx) ? 1.0f : 0.0f);
            float sn = s[i] * orc_edm_expf(-beta[i] * dt);
            const uint32_t dist = (i >= idx) ? (i - idx) : (idx - i);
            sn = sn + beta[i] * w[dist];
            v[i] = vv;
            s[i] = sn;
        }
        now = now + dt;
        ++events;
        uint32_t mi = 0;
        for (uint32_t m = 1; m < S; ++m) {
            const int dm = abs((int)idx - (int)li[m]);
            const int d0 = abs((int)idx - (int)li[mi]);
            mi += (dm < d0) ? 1u : 0u;
        }
        if (!(crossed & (1u << mi))) {
            if (now > T) { ct[mi] = now; ci[mi] = (uint16_t)idx; crossed += (1u << mi); }
            else { lt[mi] = now; li[mi] = (uint16_t)idx; last_written |= (1u << mi); }
        }
    }
    for (uint32_t m = 0; m < S; ++m) { last_t[m] = lt[m]; last_i[m] = li[m]; cross_t[m] = ct[m]; cross_i[m] = ci[m]; }
    *accept = (crossed == full) ? 1u : 0u;
    if (n_events) *n_events = events;
    if (C) {
        C->realisations += 1;
        C->accepted += *accept;
        C->events += events;
        if (events > C->max_events_one) C->max_events_one = events;
        const uint32_t unwritten = (uint32_t)__builtin_popcount(full & ~last_written);
        C->unwritten_last_slots_all += unwritten;
        if (*accept) C->unwritten_last_slots += unwritten;
        if (crossed < full) {
            if (events >= P->max_events) C->event_cap_hits += 1;
            else C->time_cap_exits += 1;
        }
    }
}

void orc_edm_evolve_one(const orc_edm_params* P, const float* v0, const float* s0, const float* w,
                        const uint16_t* seed_ind, uint32_t r,
                        float* last_t, uint16_t* last_i, float* cross_t, uint16_t* cross_i,
                        uint32_t* accept, uint32_t* n_events)
{
    evolve_one_impl(P, v0, s0, w, seed_ind, r, last_t, last_i, cross_t, cross_i, accept, n_events, NULL);
}

/* ---- ComputeF (EventDrivenMap.cu:154-240) --------------------------------------
 * Z: S doubles (c, Z1..); f: S doubles.  seed_ind: in/out uint16[S] ([D5]).
 * Optional debug outputs (any may be NULL): v,s,w float[N]; t0,t1,restricted
 * float[S*R]; i0,i1 uint16[S*R]; accept uint32[R]; sums double[S+1] receives
 * the accepted-realisation sums (fp64, index order) and the count.
 * [D4] accepted-realisation mean: interp_oracle.c orc_masked_mean_f32.
 */
static void counters_add(orc_edm_counters* a, const orc_edm_counters* b)
{
    a->realisations += b->realisations; a->accepted += b->accepted; a->events += b->events;
    a->argmin_ties += b->argmin_ties; a->no_firing_events += b->no_firing_events; a->nan_times += b->nan_times;
    a->argmin_tree_mismatch += b->argmin_tree_mismatch;
    a->unwritten_last_slots += b->unwritten_last_slots; a->unwritten_last_slots_all += b->unwritten_last_slots_all;
    a->seed_scans_empty += b->seed_scans_empty; a->newton_cap_hits += b->newton_cap_hits;
    a->event_cap_hits += b->event_cap_hits; a->time_cap_exits += b->time_cap_exits;
    a->newton_solves += b->newton_solves; a->newton_iters += b->newton_iters; a->wave64_rounds += b->wave64_rounds;
    if (b->max_newton_iter > a->max_newton_iter) a->max_newton_iter = b->max_newton_iter;
    if (b->max_events_one > a->max_events_one) a->max_events_one = b->max_events_one;
}

static int compute_f_impl(const orc_edm_params* P, const double* Z, double* f, uint16_t* seed_ind,
                          float* dv, float* ds, float* dw, float* dt0, uint16_t* di0, float* dt1,
                          uint16_t* di1, uint32_t* daccept, float* drestricted, double* dsums, int nthreads,
                          orc_edm_counters* C)
{
    const uint32_t N = P->n_grid, S = P->n_spikes, R = P->n_real;
    if (N < 2 || N > ORC_MAX_GRID || S < 1 || S > ORC_MAX_SPIKES || R < 1) return 1;
    float U[ORC_MAX_SPIKES + 1];
    double U0[ORC_MAX_SPIKES + 1];
    U0[0] = Z[0]; U0[1] = 0.0;
    for (uint32_t i = 2; i <= S; ++i) U0[i] = Z[i - 1];
    for (uint32_t i = 0; i <= S; ++i) U[i] = (float)U0[i];
    seed_indices_impl(P, Z, seed_ind, C);

    float* v = (float*)malloc(sizeof(float) * N);
    float* s = (float*)malloc(sizeof(float) * N);
    float* w = (float*)malloc(sizeof(float) * N);
    const size_t SR = (size_t)S * R;
    float* t0 = (float*)malloc(sizeof(float) * SR);
    float* t1 = (float*)malloc(sizeof(float) * SR);
    float* xr = (float*)malloc(sizeof(float) * SR);
    uint16_t* i0 = (uint16_t*)malloc(sizeof(uint16_t) * SR);
    uint16_t* i1 = (uint16_t*)malloc(sizeof(uint16_t) * SR);
    uint32_t* acc = (uint32_t*)malloc(sizeof(uint32_t) * R);
    orc_edm_coupling(P, w);
    orc_edm_lift(P, U, v, s);
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
    for (long long r = 0; r < (long long)R; ++r) {
        float lt[ORC_MAX_SPIKES], ct[ORC_MAX_SPIKES];
        uint16_t li[ORC_MAX_SPIKES], ci[ORC_MAX_SPIKES];
        uint32_t a;
        orc_edm_counters local;
        memset(&local, 0, sizeof(local));
        evolve_one_impl(P, v, s, w, seed_ind, (uint32_t)r, lt, li, ct, ci, &a, NULL, C ? &local : NULL);
        for (uint32_t m = 0; m < S; ++m) {
            const size_t k = (size_t)m * R + (size_t)r;     /* [spike][realisation], :661-668 */
            t0[k] = lt[m]; i0[k] = li[m]; t1[k] = ct[m]; i1[k] = ci[m];
        }
        acc[r] = a;
        if (C) {
#pragma omp critical(orc_edm_counters_merge)
            counters_add(C, &local);
        }
    }
    orc_restrict_f32(t0, i0, t1, i1, P->time_horizon, P->L, N, xr, SR);
    float mean[ORC_MAX_SPIKES];
    uint32_t count = 0;
    /* realisation 0 of the WHOLE ensemble is the one the reference drops (:800-802,:817): a shard that does not
     * start at realisation 0 (real_offset != 0) averages its own realisations plainly */
    const int quirk = (P->mean_quirk != 0 && P->real_offset == 0) ? 1 : 0;
    orc_masked_mean_f32(xr, acc, R, S, quirk, mean, &count);
    /* :237-239  f = -U0[0]*U0[1..S] - UT + U0[0]*T, in fp64 */
    for (uint32_t m = 0; m < S; ++m)
        f[m] = (-U0[0] * U0[m + 1] - (double)mean[m]) + U0[0] * (double)P->time_horizon;
    if (dsums) {
        /* the partial block of a shard, 2S+1 doubles: [sum_m over the accepted realisations -- without realisation 0
         * when the reference's rule applies | accepted count | x0_m = restricted position of realisation 0 (0 when
         * the rule does not apply)].  Blocks of several shards add element-wise; orc_edm_residual_from_sums applies
         * the "count == 1 re-includes realisation 0" rule to the total. */
        for (uint32_t m = 0; m < S; ++m) {
            double a = 0.0;
            for (uint32_t r = 0; r < R; ++r) {
                if (quirk && r == 0) continue;
                if (acc[r] == 1u) a += (double)xr[(size_t)m * R + r];
            }
            dsums[m] = a;
            dsums[S + 1 + m] = quirk ? (double)xr[(size_t)m * R] : 0.0;
        }
        dsums[S] = (double)count;
    }
    if (dv) memcpy(dv, v, sizeof(float) * N);
    if (ds) memcpy(ds, s, sizeof(float) * N);
    if (dw) memcpy(dw, w, sizeof(float) * N);
    if (dt0) memcpy(dt0, t0, sizeof(float) * SR);
    if (dt1) memcpy(dt1, t1, sizeof(float) * SR);
    if (di0) memcpy(di0, i0, sizeof(uint16_t) * SR);
    if (di1) memcpy(di1, i1, sizeof(uint16_t) * SR);
    if (daccept) memcpy(daccept, acc, sizeof(uint32_t) * R);
    if (drestricted) memcpy(drestricted, xr, sizeof(float) * SR);
    free(v); free(s); free(w); free(t0); free(t1); free(xr); free(i0); free(i1); free(acc);
    return 0;
}

int orc_edm_compute_f(const orc_edm_params* P, const double* Z, double* f, uint16_t* seed_ind,
                      float* dv, float* ds, float* dw, float* dt0, uint16_t* di0, float* dt1,
                      uint16_t* di1, uint32_t* daccept, float* drestricted, double* dsums, int nthreads)
{
    return compute_f_impl(P, Z, f, seed_ind, dv, ds, dw, dt0, di0, dt1, di1, daccept, drestricted, dsums, nthreads, NULL);
}

/* the same evaluation, also ADDING its decision-coverage counts to *C (the caller zeroes C before the first call) */
int orc_edm_compute_f_counted(const orc_edm_params* P, const double* Z, double* f, uint16_t* seed_ind,
                              float* dv, float* ds, float* dw, float* dt0, uint16_t* di0, float* dt1,
                              uint16_t* di1, uint32_t* daccept, float* drestricted, double* dsums, int nthreads,
                              orc_edm_counters* C)
{
    return compute_f_impl(P, Z, f, seed_ind, dv, ds, dw, dt0, di0, dt1, di1, daccept, drestricted, dsums, nthreads, C);
}

/* vectorised probes of the math routines (tests compare the GPU versions) */
void orc_edm_math_probe(int op, const float* a, const float* b, float* out, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        switch (op) {
            case 0: out[i] = orc_edm_expf(a[i]); break;
            case 1: out[i] = orc_edm_logf(a[i]); break;
            case 2: out[i] = orc_edm_powf(a[i], b[i]); break;
            case 3: out[i] = orc_edm_erfinvf(a[i]); break;
            default: out[i] = NAN;
        }
    }
}

/* f from the element-wise sum of the shards' partial blocks (see orc_edm_compute_f): the averaging rule applied to
 * the totals, one rounding to fp32, then EventDrivenMap.cu:237-239 in fp64 */
void orc_edm_residual_from_sums(const orc_edm_params* P, const double* Z, const double* sc, double* f)
{
    const uint32_t S = P->n_spikes;
    double U0[ORC_MAX_SPIKES + 1];
    U0[0] = Z[0]; U0[1] = 0.0;
    for (uint32_t i = 2; i <= S; ++i) U0[i] = Z[i - 1];
    const double count = sc[S];
    for (uint32_t m = 0; m < S; ++m) {
        const double s = sc[m] + ((P->mean_quirk != 0 && count == 1.0) ? sc[S + 1 + m] : 0.0);
        const float mean = (float)(s / count);
        f[m] = (-U0[0] * U0[m + 1] - (double)mean) + U0[0] * (double)P->time_horizon;
    }
}

void orc_edm_default_params(orc_edm_params* p)
{
    /* parameters.hpp:1-15, Driver.cu:16,19 ; counterMax := 100 (undefined upstream) */
    p->vth = 1.0f; p->a1 = 11.0f; p->a2 = 7.0f; p->b1 = 5.0f; p->b2 = 3.5f; p->I = 0.9f; p->L = 3.0f;
    p->newton_tol = 1e-6; p->newton_max_iter = 100; p->n_spikes = 3; p->time_horizon = 5.0f;
    p->n_grid = 1024; p->n_real = 1000; p->beta_mean = 13.0589f; p->beta_stddev = 0.0f;
    p->seed = 0x5EED0005ull; p->math_mode = 0; p->mean_quirk = 1; p->max_events = 1u << 20; p->real_offset = 0;
}
