// fp32 arithmetic of the EventDrivenMap pipeline on gfx950 (and on the host for
// the coupling table).  MATH = 0 (MI_EDM_MATH_EXACT): software exp/log built
// only from IEEE operations and explicit fmaf -- bit-identical on the GPU, on
// the host and in the CPU oracle's restatement of the same recipe.
// MATH = 1 (MI_EDM_MATH_FAST): v_exp_f32 / v_log_f32 hardware transcendentals.
// Compiled with -ffp-contract=off: nothing fuses unless written as fmaf.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#define MI_HD __host__ __device__ __forceinline__
#ifndef MI_EDM_FIRE_FILTER
#define MI_EDM_FIRE_FILTER 1   // will_fire: settle the clear cases with the hardware transcendentals (0: always the exact path)
#endif

namespace edm {

template <int MATH>
MI_HD float expf_(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (MATH == 1) return __expf(x);
#endif
    // Branch-free: the polynomial runs on a clamped argument and the special cases are selected at the
    // end (same values as the early-return form in oracle/edm_oracle.c for every input).
    const float xc = fminf(fmaxf(x, -104.0f), 89.0f);
    const float n = rintf(xc * 0x1.715476p+0f);
    float r = fmaf(n, -0x1.62e4p-1f, xc);
    r = fmaf(n, -0x1.7f7d1cp-20f, r);
    const float z = r * r;
    float p = 0x1.a0d2bcp-13f;
    p = fmaf(p, r, 0x1.6e8716p-10f);
    p = fmaf(p, r, 0x1.1112ep-7f);
    p = fmaf(p, r, 0x1.5554ep-5f);
    p = fmaf(p, r, 0x1.555554p-3f);
    p = fmaf(p, r, 0x1.000002p-1f);
    p = fmaf(p, z, r);
    p = p + 1.0f;
    float y = ldexpf(p, (int)n);
    // The oracle's two range cases need no select here: for x > 88.72283935546875 the clamped argument gives n = 128 and
    // p >= 1, so ldexpf overflows to +inf; for x < -103.97208404541015625 it gives n = -150 and p in [0.972, 1), i.e. less
    // than half of the smallest subnormal, which rounds to 0 (tests/test_edm_gpu.py walks every float around both
    // thresholds).  Only NaN has to be put back (fmaxf / fminf drop it).
    y = (x != x) ? x : y;
    return y;
}

template <int MATH>
MI_HD float logf_(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (MATH == 1) return __logf(x);
#endif
    const bool regular = (x > 0.0f) && (x < INFINITY);
    const float xs = regular ? x : 1.0f;
    int e;
    float m = frexpf(xs, &e);
    const bool low = m < 0x1.6a09e6p-1f;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const float f = m - 1.0f;
    const float z = f * f;
    float p = 0x1.203736p-4f;
    p = fmaf(p, f, -0x1.d7a37p-4f);
    p = fmaf(p, f, 0x1.de4a34p-4f);
    p = fmaf(p, f, -0x1.fcba9ap-4f);
    p = fmaf(p, f, 0x1.23d37ep-3f);
    p = fmaf(p, f, -0x1.555ca2p-3f);
    p = fmaf(p, f, 0x1.999a2ep-3f);
    p = fmaf(p, f, -0x1.fffffep-3f);
    p = fmaf(p, f, 0x1.555554p-2f);
    const float fe = (float)e;
    float y = (p * f) * z;
    y = fmaf(fe, -0x1.bd0106p-13f, y);
    y = fmaf(-0.5f, z, y);
    float r = f + y;
    r = fmaf(fe, 0x1.63p-1f, r);
    r = (x == INFINITY) ? x : r;
    r = (x == 0.0f) ? -INFINITY : r;
    r = (x < 0.0f) ? NAN : r;
    r = (x != x) ? x : r;
    return r;
}

template <int MATH>
MI_HD float powf_(float a, float b)
{
    return expf_<MATH>(b * logf_<MATH>(a));
}

template <int MATH>
MI_HD float erfinvf_(float x)
{
    float w = -logf_<MATH>((1.0f - x) * (1.0f + x));
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

MI_HD uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// per-neuron beta: counter-based draw (replaces cuRAND XORWOW seeded by clock(),
// EventDrivenMap.cu:103-104,179); sigma == 0 -> the mean exactly.
template <int MATH>
MI_HD float beta_of(float mean, float sigma, uint64_t seed, uint32_t n_grid, uint64_t r, uint32_t i)
{
    if (sigma == 0.0f) return mean;
    const uint64_t ctr = r * n_grid + i;
    const uint64_t h = mix64(seed + 0x9E3779B97F4A7C15ull * (ctr + 1));
    const float u = ((float)(uint32_t)(h >> 40) + 0.5f) * 0x1.0p-24f;
    const float z = 0x1.6a09e6p+0f * erfinvf_<MATH>(fmaf(2.0f, u, -1.0f));
    return fmaf(sigma, z, mean);
}

// model constants handed to the kernels by value
struct Model {
    float vth, a1, a2, b1, b2, I, L;
    float tol_f;          // largest float <= newton_tol: (double)|f| > tol  <=>  |f| > tol_f
    uint32_t max_iter;
    uint32_t S, N, R;
    float T;
    float beta_mean, beta_sigma;
    uint64_t seed;
    uint32_t real_offset;
    uint32_t max_events;
};

struct FdF {
    float f, df;
};

// a / b: IEEE division in EXACT mode; v_rcp_f32 * a (about 1 ulp) in FAST mode on the device
template <int MATH>
MI_HD float div_(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (MATH == 1) return a * __builtin_amdgcn_rcpf(b);
#endif
    return a / b;
}

// a / c for a divisor c that is the same in every lane and in every call of a launch (UNI; 1 - beta, beta - 1, vth - I of
// the homogeneous model), EXACT mode on the device: the IEEE quotient without the eleven-instruction expansion of `/`
// (v_div_scale x2, v_rcp, five fma, v_div_fmas, v_div_fixup; the pipeline has some 45 divisions per event, 26 of them by
// such a divisor).  With y = RN(1/c) -- an IEEE division that the compiler hoists out of every loop -- q0 = RN(a y) is
// within 1.5 ulp of a/c; one Newton step q1 = RN(q0 + RN(a - c q0) y) (the residual is exact by fma) is a faithful
// rounding of a/c, and by Markstein's theorem (Markstein 1990; Muller et al., Handbook of Floating-Point Arithmetic,
// sec. 4.7) a second step from a faithful quotient with y = RN(1/c) gives RN(a/c) itself: five multiply/fma.  The
// theorem wants no overflow and exact (possibly subnormal) residuals: |c| in [2^-20, 2^20] and |a| in [2^-100, 2^101).
// Every other numerator -- zero, subnormal, tiny, huge, infinite, NaN -- takes the IEEE expansion (the wave skips it when no
// lane needs it: the guard is a bit-field extract, a subtract and a compare), and so does a divisor outside its range.
// Checked against `/` bit for bit by tests/test_edm_gpu.py::test_uniform_divisor_quotient_is_the_ieee_quotient.
// Used where it pays: with eight waves per SIMD (N <= 512, the reference's Driver.cu) Evolve is 13 % faster with it; at
// N = 1024 (four waves per SIMD) the guard costs what the division saves (profiles/r02_edm_evolve_phases.log).
// GUARD = false: the caller vouches for |a| in [2^-100, 2^101) or a NaN (evolve_kernel's state pass tracks the range of the
// synaptic variables it divides), and the quotient is the five operations alone.
// ONE = true: a single correction step.  With rc = RN(1 / c) the first corrected quotient is already RN(a / c) for every
// divisor tried (and the second step then changes nothing); rather than lean on that, the host PROVES it for the launch's
// divisor before it picks a ONE kernel: divisor_check_kernel (mi_edm.hip) compares the one-step quotient with the IEEE
// quotient for all 2^23 significands of a (the quotient scales exactly with a's exponent inside the guarded range and is
// sign-symmetric), once per divisor value.
template <int MATH, bool UNI, bool GUARD = true, bool ONE = false>
MI_HD float div_by(float a, float c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (MATH == 1) return a * __builtin_amdgcn_rcpf(c);
    if constexpr (UNI) {
        {   // |c| in [2^-20, 2^20]: checked ONCE on the host before a UNI kernel is launched (uniform_divisors_ok, mi_edm.hip) --
            // as a test inside the loops the compiler kept it there, two scalar branches per division
            const float rc = 1.0f / c;
            const float q0 = a * rc;
            float q = fmaf(fmaf(-c, q0, a), rc, q0);
            if constexpr (!ONE) q = fmaf(fmaf(-c, q, a), rc, q);
            // |a| in [2^-100, 2^101): two compares on |a| (the sign comes off as an operand modifier).  Anything else -- zero,
            // subnormal, tiny, huge, infinite, NaN -- takes the IEEE expansion; the wave skips it when no lane needs it
            if constexpr (!GUARD) return q;
            const float aa = fabsf(a);
            const bool in_range = aa >= 0x1.0p-100f && aa < 0x1.0p+101f;
            if (__any(!in_range)) {
                asm volatile("");       // (a side effect: keeps the compiler from speculating the expansion on every call)
                const float qi = a / c;
                q = in_range ? q : qi;
            }
            return q;
        }
    }
#endif
    return a / c;
}

// fun/dfun, EventDrivenMap.cu:544-552, sharing e1 = exp(-t), e2 = exp((1-beta) t)
template <int MATH, bool UNI = false, bool ONE = false>
MI_HD FdF fun_dfun_e(const Model& M, float e1, float e2, float v, float s, float beta)
{
    const float se = s * e1;
    FdF r;
    r.f = ((v * e1 + M.I * (1.0f - e1)) + div_by<MATH, UNI, true, ONE>(se, 1.0f - beta) * (e2 - 1.0f)) - M.vth;
    r.df = ((M.I * e1 - v * e1) + se * e2) + div_by<MATH, UNI, true, ONE>(se * (e2 - 1.0f), beta - 1.0f);
    return r;
}

// eventTime, EventDrivenMap.cu:554-573, split in two so that callers can batch the Newton solves:
//   will_fire()    the closed-form test of :559 ("decision")
//   newton_time()  the Newton iteration of :561-571 for a neuron that will fire
// A neuron with decision false has f = fun(0)*0, i.e. +-0 or NaN: the reference's loop is not entered and
// eventTime returns exactly |0| + 100 (requires tol >= 0, validated) -> kNever.
// At t = 0 both exponentials are exactly 1 (expf_(+-0) == 1), so the first evaluation needs no exp.
constexpr float kNever = 100.0f;

// 0 < vth - I <= 1: the condition under which a negative (or NaN) synaptic variable alone settles will_fire (below).
// GAP = the host has found it true before the launch (mi_edm.hip launch_evolve): the test is then compile-time true -- as a
// run-time test inside the state pass it was scalar branches per slice.  (UNI kernels -- uniform divisors in range -- are
// only launched with it true.)
template <bool GAP>
MI_HD bool gap_settles_sign(const Model& M)
{
    const float gap = M.vth - M.I;
    return GAP ? true : (gap > 0.0f && gap <= 1.0f);
}

template <int MATH, bool UNI = false, bool FILTER = (MI_EDM_FIRE_FILTER != 0), bool GAP = UNI>
MI_HD bool will_fire(const Model& M, float v0, float s0, float beta)
{
    const float gap = M.vth - M.I;
    const bool gap_ok = gap_settles_sign<GAP>(M);
    // Exact shortcuts: a negative (or NaN) ratio makes log(ratio) NaN, hence pw, thr NaN and `v0 > thr` false.
    // With 0 < gap <= 1 the quotient s0/gap cannot underflow to -0, so s0 < 0 already decides it (no division), and so
    // does a NaN (the poisoned stretch of the lift profile: 3.8 of 16 slices per event at N = 1024); the inhibitory
    // surround puts most of the ring in the first case, 64 contiguous neurons per wave step.
    if (gap_ok && !(s0 >= 0.0f)) return false;   // s0 < 0, or NaN: NaN / gap is NaN and fails `ratio >= 0` below
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("");       // (a side effect: what follows stays behind the exit above -- the wave skips it when no lane has s0 >= 0)
    // EXACT mode, s0 >= 0: the test v0 > thr costs a software log, a software exp and three IEEE divisions (about 86
    // instructions on 3.2 slices per event) and is almost never close.  Evaluate thr first with the hardware transcendentals
    // (v_rcp_f32, v_log_f32, v_exp_f32: 1 ulp each) and settle every case that is clear of it by 1e-4 of the summed
    // magnitudes of its terms; close calls, NaN, zero and extreme arguments take the exact path below, so the decision
    // is the exact path's whenever the two evaluations could disagree.  Error budget: the two values of
    // pw = ratio^(1/beta) differ relatively by at most ln 2 |log2(ratio)/beta| 2^-22 (argument error of either
    // exponential) + a few ulp, i.e. < 3.5e-6 under the guard |log2(ratio)/beta| <= 16; thr is affine in pw and ratio
    // with coefficients bounded by the magnitudes summed in `mag` = |vth| pw + |I| (1 + pw) + |c| (ratio + pw), so
    // |thr_hw - thr_exact| < 4e-6 mag: a factor 25 inside the margin 1e-4 mag (the code uses a bound of mag that is cheaper
    // to form).  tests/test_edm_gpu.py compares the decisions of both forms on dense samples around the threshold (math
    // probe ops 6, 7) and every stage tap of ComputeF with the oracle.
    // Written without branches (one `decided` flag per lane): as nested ifs the pre-decision was five scalar branches per
    // slice, and the event loop is bound by those as much as by its vector instructions (profiles/r04_edm_evolve_steps.log).
    if constexpr (MATH == 0 && FILTER) {
        if (gap_ok) {                                           // (s0 >= 0 here)
            const float rf = s0 * __builtin_amdgcn_rcpf(gap);
            const float ex = __builtin_amdgcn_logf(rf) * __builtin_amdgcn_rcpf(beta);       // log2(ratio) / beta
            const float pwf = __builtin_amdgcn_exp2f(ex);
            const float c = gap * __builtin_amdgcn_rcpf(beta - 1.0f);
            // thr = vth pw + I (1 - pw) - c (ratio - pw) = I + gap pw - c (ratio - pw): two fused multiply-adds (this is the
            // approximate side: any form within the margin will do), and a margin from the slightly coarser bound
            // mag <= (|vth| + |I|) (1 + pw) + |c| (ratio + pw) whose coefficients are the same for every neuron of the launch
            const float thrf = fmaf(-c, rf - pwf, fmaf(gap, pwf, M.I));
            const float margin = fmaf(1.0e-4f * fabsf(c), rf + pwf, (1.0e-4f * (fabsf(M.vth) + fabsf(M.I))) * (1.0f + pwf));
            const bool usable = (rf >= 0x1.0p-40f) & (rf <= 0x1.0p+40f) & (fabsf(ex) <= 16.0f) & (margin < INFINITY);   // false for NaN anywhere
            const bool above = v0 > thrf + margin, below = v0 < thrf - margin;
            if (usable & (above | below)) return above;
        }
    }
#endif
    const float ratio = div_by<MATH, UNI>(s0, gap);   // (gap is always uniform; UNI only says whether the kernel opted in)
    if (!(ratio >= 0.0f)) return false;
    const float pw = powf_<MATH>(ratio, div_<MATH>(1.0f, beta));
    const float thr = (M.vth * pw + M.I * (1.0f - pw)) - div_<MATH>(gap, beta - 1.0f) * (ratio - pw);
    return v0 > thr;
}

// iters (optional, debug taps): receives the number of iterations taken
template <int MATH, bool UNI = false>
MI_HD float newton_time(const Model& M, float v0, float s0, float beta, uint32_t* iters = nullptr)
{
    float t = 0.0f;
    FdF r = fun_dfun_e<MATH, UNI>(M, 1.0f, 1.0f, v0, s0, beta);
    float f = r.f, df = r.df;
    uint32_t counter = 0;
    while ((fabsf(f) > M.tol_f) && (counter < M.max_iter)) {
        t = t - div_<MATH>(f, df);
        r = fun_dfun_e<MATH, UNI>(M, expf_<MATH>(-t), expf_<MATH>((1.0f - beta) * t), v0, s0, beta);
        f = r.f;
        df = r.df;
        ++counter;
    }
    if (iters) *iters = counter;
    return fabsf(t);
}

// v_permlane32_swap_b32 on (x, x): one VALU instruction, no LDS round trip.  It exchanges the upper half of its first
// operand with the lower half of its second, so with both operands = x the first comes back holding the LOWER lane's
// value in both lanes of every pair (l, l + 32) and the second the UPPER lane's value in both.  Both lanes of a pair
// must be active.
__device__ __forceinline__ void pair_values(float x, float& of_lower, float& of_upper)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // Written as inline assembly: with ROCm 7.2's clang the builtin's SECOND result is folded into its first at -O3 whenever
    // the two are used separately (`r[1]` becomes `extractvalue 0`; the -O0 lowering is right), which would silently hand
    // both lanes the lower lane's value.  s_nop 1: the wait states a VALU write of either operand needs before the swap
    // reads it (what the compiler inserts in front of its own v_permlane32_swap).
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    of_lower = a;
    of_upper = b;
#else
    of_lower = of_upper = x;      // (the host pass only parses the kernels)
#endif
}
// the value x holds in lane (lane ^ 32) of the wave (test probe).  upper = lane >= 32.
__device__ __forceinline__ float other_half(float x, bool upper)
{
    float lo, hi;
    pair_values(x, lo, hi);
    return upper ? lo : hi;
}

// newton_time() for a neuron given to the lane PAIR (l, l + 32), both holding the same (v0, s0, beta): the lower lane
// evaluates exp(-t) and f, the upper lane exp((1 - beta) t) and df, in ONE pass of the exponential and ONE of the
// division per iteration instead of two each (the rounds run with some ten of 64 lanes busy, so the partner is free),
// and the halves are exchanged by v_permlane32_swap.  Every operation a lane keeps is the one newton_time performs, on
// the same operands, so t is the same bit pattern in both lanes and equal to newton_time's: -t is (-1) t; the
// derivative's quotient a / (beta - 1) is formed as -(a / (1 - beta)), the same number (beta - 1 = -(1 - beta) exactly,
// and IEEE division is sign-symmetric; only the sign of a NaN can differ, and a NaN time never wins the arg-min).
template <int MATH, bool UNI = false, bool ONE = false>
__device__ __forceinline__ float newton_time_paired(const Model& M, float v0, float s0, float beta, bool upper,
                                                    uint32_t* iters = nullptr)
{
    const float omb = 1.0f - beta;
    float t = 0.0f;
    FdF r = fun_dfun_e<MATH, UNI, ONE>(M, 1.0f, 1.0f, v0, s0, beta);
    float f = r.f, df = r.df;
    const float cx = upper ? omb : -1.0f;
    uint32_t counter = 0;
    while ((fabsf(f) > M.tol_f) && (counter < M.max_iter)) {
        t = t - div_<MATH>(f, df);
        const float e = expf_<MATH>(cx * t);              // lower: exp(-t); upper: exp((1 - beta) t)
        float e1, e2;
        pair_values(e, e1, e2);                           // both lanes now hold both exponentials
        const float se = s0 * e1;
        const float em1 = e2 - 1.0f;
        float q = div_by<MATH, UNI, true, ONE>(upper ? se * em1 : se, omb);
        q = upper ? -q : q;
        const float ve = v0 * e1;
        const float fv = ((ve + M.I * (1.0f - e1)) + q * em1) - M.vth;     // :546 (meaningful in the lower lane)
        const float dv = ((M.I * e1 - ve) + se * e2) + q;                  // :551 (meaningful in the upper lane)
        pair_values(upper ? dv : fv, f, df);              // the lower lane's f and the upper lane's f' to both
        ++counter;
    }
    if (iters) *iters = counter;
    return fabsf(t);
}

template <int MATH>
MI_HD float event_time(const Model& M, float v0, float s0, float beta)
{
    return will_fire<MATH>(M, v0, s0, beta) ? newton_time<MATH>(M, v0, s0, beta) : kNever;
}

}  // namespace edm
