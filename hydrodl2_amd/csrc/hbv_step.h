// hbv_step.h -- one explicit HBV day and its adjoint, per lane (basin x member).
//
// Device math of the MI355X time-stepper.  Restates, in registers, the body of
// the reference's `for t in range(nsteps)` loop:
//   HBV 1.0   src/hydrodl2/models/hbv/hbv.py:428-494
//   HBV 1.1p  src/hydrodl2/models/hbv/hbv_1_1p.py:427-506   (+ parBETAET, capillary rise)
//   HBV 2.0   src/hydrodl2/models/hbv/hbv_2.py:471-556      (+ elevation TT switch, lateral flow)
// with the reference's association order and, in the adjoint,
// PyTorch autograd's sub-gradient conventions: minimum() ties
// split the gradient, clamp bounds are inclusive, comparison masks are constants.
//
// Everything is `static inline` on plain floats so the very same code compiles
// for gfx950 (hipcc) and for the host (tests/hosttest builds it with g++ to check
// the math against the oracle without a GPU).  Build with -ffp-contract=off:
// the reference rounds after every operator and thresholds such as T >= TT or
// min/clamp branch on those roundings.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#define HBVX_HD __host__ __device__ __forceinline__
#define HBVX_HDM __host__ __device__ __forceinline__
#else
#define HBVX_HD static inline
#define HBVX_HDM inline
#endif

namespace hbvx {

enum : int { MODEL_HBV10 = 0, MODEL_HBV11P = 1, MODEL_HBV20 = 2, MODEL_HBVADJ = 3, MODEL_HOURLY = 4 };
enum : int {
    P_BETA = 0, P_FC, P_K0, P_K1, P_K2, P_LP, P_PERC, P_UZL, P_TT, P_CFMAX, P_CFR, P_CWH,
    P_BETAET, P_C, P_RT, P_AC, P_F0, P_FMIN, P_ALPHA, NPARAM_MAX
};

HBVX_HD float fmax_(float a, float b) { return fmaxf(a, b); } // v_max_f32
HBVX_HD float fmin_(float a, float b) { return fminf(a, b); } // v_min_f32

// a / b for the forward quantities (SM/FC, SM/(LP*FC)).  The compiler's IEEE sequence is ten
// dependent instructions (div_scale x2, rcp, 4 fma, div_fmas, div_fixup).  On the device: reciprocal,
// ONE Newton step on the reciprocal, multiply, one fused correction of the quotient.  The numerator
// a is a storage -- it sits on the day-to-day dependency chain, where a dependent instruction costs a
// lone wave ~10 cycles (tools/micro/issue_rate.hip) -- while b is a parameter: its reciprocal and
// the Newton step do not depend on the state (static b: hoisted out of the time loop; dynamic b: issued
// beside the chain).  So only three dependent instructions follow a (mul, fma, fma) instead of the
// five of "rcp, q, then two corrections of q".  Same instruction count.  Correctly rounded on the path's
// operand range except in astronomically rare double-rounding cases, and exactly 1.0f when a == b (the
// inclusive clamp gradients at SM == FC depend on that): tests/test_gpu_parity.py::test_div_on_gpu.
// Operands here are O(1e-5 .. 1e3): no scaling needed.
HBVX_HD float div_(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float rc = __builtin_amdgcn_rcpf(b);
    rc = __builtin_fmaf(__builtin_fmaf(-b, rc, 1.0f), rc, rc);
    const float q = a * rc;
    const float r = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r, rc, q);
#else
    return a / b;
#endif
}

// A/B experiment (-DHBVX_Q_APPROX=1): SM2 / (LP FC) as one multiplication by the refined reciprocal -- one dependent
// instruction behind the storage instead of three, an ulp or two off the correctly rounded quotient.
#ifndef HBVX_Q_APPROX
#define HBVX_Q_APPROX 0
#endif
HBVX_HD float div_q_(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__) && HBVX_Q_APPROX
    float rc = __builtin_amdgcn_rcpf(b);
    rc = __builtin_fmaf(__builtin_fmaf(-b, rc, 1.0f), rc, rc);
    return a * rc;
#else
    return div_(a, b);
#endif
}

// a / b inside the adjoint only (gradients are compared at rtol 1e-3): rcp + multiply.
HBVX_HD float div_approx_(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return a * __builtin_amdgcn_rcpf(b);
#else
    return a / b;
#endif
}

// torch.sigmoid: 1/(1+exp(-x))  (hbv.py:201)
HBVX_HD float sigmoid_(float v) { return 1.0f / (1.0f + expf(-v)); }
// The same for the per-day values of DYNAMIC parameters: v_exp_f32 + v_rcp_f32, 4 instructions
// instead of ~25 (expf + IEEE division).  Every dynamic parameter costs one sigmoid per lane-day
// in the forward and two in the time-parallel adjoint; with 14 of them (config 3) the accurate
// version was more work than the rest of the adjoint step.  Error: a few ulp of the unit value,
// < 1e-6 relative on the de-scaled parameter; saturates to exactly 0 / 1.  Static values (read
// once per launch) keep sigmoid_, so thresholds on static parameters (TT, FC, PERC, UZL ...) see
// the library-accurate value.
HBVX_HD float sigmoid_dyn_(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504f));
#else
    return 1.0f / (1.0f + expf(-v));
#endif
}
// core/calc/utils.py:24: multiply, then add (no FMA)
HBVX_HD float descale_(float u, float lo, float hi) { return u * (hi - lo) + lo; }

// ---------------------------------------------------------------------------
// x**y for x > 0 (reference: `(SM / parFC) ** parBETA`, hbv.py:462; evapfactor ** parBETAET,
// hbv.py:476).  Three versions, fastest last:
//   pow_f64_   fp64 polynomials (~40 instructions): log2 via atanh series after frexp, exp2 via
//              Taylor after rint; 0.5 ulp.  The first fast version (ocml's correctly-rounded powf
//              is ~200 dependent fp32 instructions); kept for -DHBVX_POW_F64 and as the accuracy
//              yardstick in the tests.
//   pow_hw_    hardware log2 / exp2 on the reduced mantissa with hi+lo exponent arithmetic, all
//              special bases handled like powf; <= 2.1 ulp measured.  The general pow_pos_.
//   pow_step_  pow_hw_'s core for the time step: base clamped at FLT_MIN, no special-value
//              selects, no exponent clamp (see there).
// (tests/test_step_math_host.py::test_pow_accuracy, tests/test_gpu_parity.py::test_pow_on_gpu)
HBVX_HD double rcp64_(double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(v);       // v_rcp_f64 + one Newton step
    double e = __builtin_fma(-v, r, 1.0);
    return __builtin_fma(r, e, r);
#else
    return 1.0 / v;
#endif
}

// a*b + c in fp64.  On the device as one three-address v_fma_f64: hipcc otherwise turns each
// Horner step into v_mov_b64 (copy the coefficient) + v_fmac_f64.
HBVX_HD double fma64_(double a, double b, double c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return __builtin_fma(a, b, c);
#endif
}

HBVX_HD float pow_f64_(float x, float y)
{
    const double xd = (double)x;
#if defined(__HIP_DEVICE_COMPILE__)
    double m = __builtin_amdgcn_frexp_mant(xd);   // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(xd);
#else
    int k;
    double m = frexp(xd, &k);
#endif
    const int adj = (m < 0.70710678118654752440) ? 1 : 0; // m -> [sqrt(1/2), sqrt(2))
    m = adj ? m + m : m;
    k -= adj;
    const double s = (m - 1.0) * rcp64_(m + 1.0);
    const double s2 = s * s;
    // log2(m) = (2/ln2) (s + s^3/3 + ... + s^11/11), |s| <= 0.1716: next term < 3e-11.
    // Estrin form: the fp64 chain is latency-bound on one wave, depth 3 beats Horner's 5.
    const double s4 = s2 * s2;
    const double pa = fma64_(0.9617966939259756, s2, 2.8853900817779268);
    const double pb = fma64_(0.4121985831111324, s2, 0.5770780163555853);
    const double pc = fma64_(0.2623081892525388, s2, 0.3205988979753252);
    double pl = fma64_(fma64_(pc, s4, pb), s4, pa);
    const double lg = fma64_(pl, s, (double)k);
    double z = (double)y * lg;
    z = z > 130.0 ? 130.0 : (z < -160.0 ? -160.0 : z);
    const double n = __builtin_rint(z);
    const double f = z - n; // |f| <= 1/2 ; 2^f = sum (f ln2)^i / i!, i <= 8: next term < 2e-10
    const double f2 = f * f, f4 = f2 * f2;
    const double q0 = fma64_(6.9314718055994531e-01, f, 1.0);
    const double q1 = fma64_(5.5504108664821580e-02, f, 2.4022650695910071e-01);
    const double q2 = fma64_(1.3333558146428443e-03, f, 9.6181291076284772e-03);
    const double q3 = fma64_(1.5252733804059841e-05, f, 1.5403530393381610e-04);
    const double r0 = fma64_(q1, f2, q0), r1 = fma64_(q3, f2, q2);
    double e = fma64_(fma64_(1.3215486790144310e-06, f4, r1), f4, r0);
    float r = ldexpf((float)e, (int)n);
    const float inf = __builtin_inff();
    r = (x == 0.0f) ? (y > 0.0f ? 0.0f : (y == 0.0f ? 1.0f : inf)) : r;
    r = (x == inf) ? (y > 0.0f ? inf : (y == 0.0f ? 1.0f : 0.0f)) : r;
    r = (x < 0.0f || x != x || y != y) ? __builtin_nanf("") : r;
    return r;
}

// x**y for x >= 0 on the hardware transcendentals, arranged so that their 1-ulp errors stay 1-ulp
// errors of the result:
//   x = m 2^k, m in [sqrt(1/2), sqrt(2))  ->  log2 x = k + v_log_f32(m), |log2 m| <= 1/2: the
//   absolute error of the hardware log2 is <= 2^-25 whatever the magnitude of k;
//   k + log2 m and its product with y are carried as hi + lo pairs (Fast2Sum / fma residuals);
//   y log2 x = n + f, |f| <= 1/2  ->  x**y = v_exp_f32(f) 2^n.
// Error: <= (1 + 0.7 |y| / 4 ...) ~ 2-3 ulp for the exponents of this model (BETA <= 6, BETAET <= 5),
// measured by hbvx_selftest_pow (tests/test_gpu_parity.py); torch's own GPU pow is in the same class.
// pow_f64_ below is the 0.5-ulp fp64-polynomial version (-DHBVX_POW_F64 selects it): on the VALU-bound
// stepper waves it costs ~260 cycles per call against ~130 for this one.
// CLAMP: bound y log2 x before splitting it (needed when the product can overflow to +-inf:
// arbitrary y; inside the time step |y log2 x| < 800 and v_ldexp_f32 saturates by itself).
template <bool CLAMP>
HBVX_HD float pow_core_(float x, float y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float m = __builtin_amdgcn_frexp_mantf(x);     // [0.5, 1)
    int k = __builtin_amdgcn_frexp_expf(x);
#else
    int k;
    float m = frexpf(x, &k);
#endif
    const bool adj = m < 0.70710678f;
    m = adj ? m + m : m;
    k -= adj ? 1 : 0;
#if defined(__HIP_DEVICE_COMPILE__)
    const float l0 = __builtin_amdgcn_logf(m);     // v_log_f32: log2
#else
    const float l0 = log2f(m);
#endif
    const float kf = (float)k;
    const float lh = kf + l0;                      // Fast2Sum: |kf| >= |l0| or kf == 0
    const float ll = l0 - (lh - kf);
    float zh = y * lh;
    float zl = fmaf(y, lh, -zh);
    zl = fmaf(y, ll, zl);
    if (CLAMP) zh = zh > 130.0f ? 130.0f : (zh < -160.0f ? -160.0f : zh);
    const float n = rintf(zh);
    const float f = (zh - n) + zl;
#if defined(__HIP_DEVICE_COMPILE__)
    const float e = __builtin_amdgcn_exp2f(f);     // v_exp_f32: 2^f
#else
    const float e = exp2f(f);
#endif
    return ldexpf(e, (int)n);
}

HBVX_HD float pow_hw_(float x, float y)
{
    float r = pow_core_<true>(x, y);
    // special bases as flat selects of ready values (nested conditionals become EXEC-masked
    // branches when y is not loop-invariant)
    const float inf = __builtin_inff();
    const float at0 = (y > 0.0f) ? 0.0f : inf, atinf = (y > 0.0f) ? inf : 0.0f;
    const float r0 = (y == 0.0f) ? 1.0f : at0, ri = (y == 0.0f) ? 1.0f : atinf;
    r = (x == 0.0f) ? r0 : r;
    r = (x == inf) ? ri : r;
    const bool bad = (x < 0.0f) | (x != x) | (y != y);
    r = bad ? __builtin_nanf("") : r;
    return r;
}

HBVX_HD float pow_pos_(float x, float y)
{
#if defined(HBVX_POW_F64)
    return pow_f64_(x, y);
#else
    return pow_hw_(x, y);
#endif
}

// The power inside the time step: bases are finite and >= 0 by construction (storage ratios; the
// storages are clamped at nearzero), exponents finite and > 0 (parameter ranges).  The base is
// clamped at the smallest normal number instead of special-casing 0 -- 0**y becomes
// (1.2e-38)**y <= 4e-12 for y >= 0.3, far below the flux tolerance -- which removes the
// special-value selects from the VALU-bound stepper loop.
HBVX_HD float pow_step_(float x, float y)
{
#if defined(HBVX_POW_F64)
    return pow_f64_(x, y);
#else
    return pow_core_<false>(fmax_(x, 1.17549435e-38f), y);
#endif
}

// x**y as 2^(y log2 x) straight on the hardware transcendentals: three dependent instructions instead of the
// ~16 of pow_core_.  The error of v_log_f32 (1 ulp of log2 x) is multiplied by |y log2 x|, so the RELATIVE error
// grows with the exponent -- but every power of the implicit scheme is clamped to [0, 1] (hbv_adj.py:472-473,
// 484-485), where 2^z |z| 2^-23 <= 7e-8 ABSOLUTE whatever z <= 0, and results above 1 only need their side of 1
// (exact at x == 1).  Used inside the Newton iteration of AdjStaged::soil, whose acceptance test is 1e-3.
HBVX_HD float pow_fast_(float x, float y)   // x a normal positive number (the caller's clamp guarantees it)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
#else
    return exp2f(y * log2f(x));
#endif
}

// The powers of the time step whose RESULT is clamped to [0, 1] right away -- soil wetness (SM/FC)**BETA and the
// evaporation factor (hbv.py:462-465,474-477) -- or lies there by construction (the hourly infiltration decay).
// HBVX_POW_UNIT_FAST: 2^(y log2 x) straight on the hardware transcendentals (pow_fast_).  What the 2-ulp
// arrangement of pow_core_ buys is RELATIVE accuracy for results far from 1; on [0, 1] the plain form is as good in
// ABSOLUTE terms (|error| <= 2^z |z| 2^-23 <= 7e-8 for every z <= 0, and ~1 ulp near the clamp at 1 where |z| is
// small; exact at x == 1), and an absolute error is what the fluxes see: rech = (RAIN + tosoil) sw.
#ifndef HBVX_POW_UNIT_FAST
#define HBVX_POW_UNIT_FAST 1
#endif
HBVX_HD float pow_unit_(float x, float y)
{
#if HBVX_POW_UNIT_FAST && !defined(HBVX_POW_F64)
    // (no lower clamp on the base: the exponents of this model are positive -- BETA >= 1, BETAET >= 0.3, ALPHA >= 0.5 --
    // so x = 0 gives log2 = -inf, y * -inf = -inf, 2^-inf = 0, the limit; the storages behind x are >= nearzero anyway.
    // One dependent instruction less in front of every power.)
    return pow_fast_(x, y);
#else
    return pow_step_(x, y);
#endif
}

// HBVX_SAVE_POW: whether the forward keeps the two pre-clamp powers of a lane-day, (SM/FC)**BETA and the
// evaporation factor, next to the trajectory (hbvx_fwd_out.aux, 8 of 28 bytes per lane-day) for the adjoint.
// That paid while a power cost ~16 dependent instructions; with pow_unit_ it is log, multiply, exp, and the adjoint
// passes run beside a 3.8 GB gradient fill that wants the same HBM.  Default 0: nothing is
// written to or read from `aux`, the adjoint recomputes both powers with the forward's own instruction sequence
// -- the same bits, so every clamp predicate matches.  1 restores the saved rows (A/B builds only).
#ifndef HBVX_SAVE_POW
#define HBVX_SAVE_POW 0
#endif
constexpr bool SAVE_POW = HBVX_SAVE_POW != 0;

// HBVX_ADJ_FMA: the library is built with -ffp-contract=off because the FORWARD must round after every operator like the
// reference does (thresholds branch on those roundings, and the adjoint's recomputation of the forward must land on the
// same side of every one of them).  The adjoint's OWN arithmetic -- products of weights and incoming adjoints, the
// gradient sums, the chunk transfer maps -- decides no branch and is compared at rtol 1e-3: there a multiply-add is one
// fused instruction (a quarter fewer vector instructions in kernels whose waves are bound by their own issue rate; the fused form
// rounds once instead of twice, i.e. it is the more accurate of the two).  Clang scopes the pragma to the function
// body it opens; inlined callees keep their own setting.  g++ (the host build of this header) ignores it.
#ifndef HBVX_ADJ_FMA_OFF
#define HBVX_ADJ_FMA _Pragma("clang fp contract(fast)")
#else
#define HBVX_ADJ_FMA
#endif
#if !defined(__clang__)
#undef HBVX_ADJ_FMA
#define HBVX_ADJ_FMA
#endif

// natural log for the adjoint's d(x**y)/dy = x**y ln x: gradients are compared at rtol 1e-3,
// the hardware v_log_f32 (1 ulp on log2) is ample.
HBVX_HD float log_fast_(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __logf(v);
#else
    return logf(v);
#endif
}

// torch.minimum backward weights (ties: 1/2 each)
HBVX_HD void minw_(float a, float b, float &wa, float &wb)
{
    wa = (a < b) ? 1.0f : ((a > b) ? 0.0f : 0.5f);
    wb = 1.0f - wa;
}

// Incoming flux-series gradients of one lane (already scaled by 1/M or muwts).
struct FluxGrad {
    float gQ, gQ0, gQ1, gQ2, gET, gSWE, grech, gexc, gef, gtosoil, gPERC, gcap;
};

// All intermediates of one step.  The compiler keeps what is live in VGPRs.
template <int MODEL, bool BETAET>
struct Step {
    // inputs
    float SP, MW, SM, SUZ, SLZ;
    float P, Tf, PET;
    // snow
    float TTe, mlo, m_rain, m_snow, RAIN, SP1, dT, mp, mpc, melt, MW1, SP2;
    float cc, dT2, rp, rpc, refr, SP3, MW2, ts0, tosoil, MW3;
    // soil
    float r, sw0, sw, rt, rech, SM1, e0, exc, SM2, lpfc, q, ef0, ef, pe, ET, dd, SM3;
    // capillary
    float x1, cs, om, capp, cap, smc, SM4, slc, SLZ0;
    // groundwater
    float SUZ1, PERC, SUZ2, u0, u0c, Q0, SUZ3, Q1, SUZ4, SLZ1, SLZ1p, Q2, SLZ2, Q;
    // lateral flow
    float a0, a1, m1, m2, ee, sl;

    // The day step in its three feed-forward stages (snow -> soil -> groundwater).  fwd() runs
    // them back to back; the pipelined forward kernel (hbv_pipe.h) runs each stage on its own
    // wave.  The operations and their order are exactly those of the reference loop body.

    // snow: needs SP, MW, P, Tf; produces RAIN, tosoil, SP3, MW3      (hbv.py:429-459)
    HBVX_HDM void fwd_snow(const float *p, float elev)
    {
        const float TT = p[P_TT], CFMAX = p[P_CFMAX], CFR = p[P_CFR], CWH = p[P_CWH];
        if (MODEL == MODEL_HBV20) { // hbv_2.py:473-475
            float mhi = (elev >= 2000.0f) ? 1.0f : 0.0f;
            mlo = (elev < 2000.0f) ? 1.0f : 0.0f;
            TTe = mhi * 4.0f + mlo * TT;
        } else {
            mlo = 1.0f;
            TTe = TT;
        }
        m_rain = (Tf >= TTe) ? 1.0f : 0.0f;
        m_snow = (Tf < TTe) ? 1.0f : 0.0f;
        RAIN = P * m_rain;
        float SNOW = P * m_snow;
        SP1 = SP + SNOW;
        dT = Tf - TTe;
        mp = CFMAX * dT;
        mpc = fmax_(mp, 0.0f);
        melt = fmin_(mpc, SP1);
        MW1 = MW + melt;
        SP2 = SP1 - melt;
        cc = CFR * CFMAX;
        dT2 = TTe - Tf;
        rp = cc * dT2;
        rpc = fmax_(rp, 0.0f);
        refr = fmin_(rpc, MW1);
        SP3 = SP2 + refr;
        MW2 = MW1 - refr;
        ts0 = MW2 - CWH * SP3;
        tosoil = fmax_(ts0, 0.0f);
        MW3 = MW2 - tosoil;
    }

    // soil: needs SM, RAIN, tosoil, PET; produces rech, exc, ET, ef, SM3   (hbv.py:462-480)
    // USE_AUX: (SM/FC)^BETA and the pre-clamp evaporation factor come from the forward pass
    // (hbvx_fwd_out.aux) instead of two pow calls.
    // CHAIN (the forward time-steppers): the evaporation factor from SM1 instead of SM2 = SM1 - excess.  The
    // reference divides the storage AFTER the excess has left it (hbv.py:474); but the factor is clamped to [0, 1],
    // and whenever there is excess (SM1 > FC) both quotients are >= FC / (LP FC) >= 1 and clamp to exactly 1, while
    // without excess SM2 IS SM1 -- the same bits either way (except for LP within an ulp of 1 on a day with SM1 > 2 FC,
    // where SM1 - (SM1 - FC) can miss FC by an ulp).  What it buys: the quotient (three dependent instructions,
    // six with the BETAET power) no longer waits for the excess (three more) on the day-to-day chain of the soil wave,
    // it runs beside it.  The adjoint's recomputation keeps the reference's form (its masks test the quotient of SM2).
    template <bool USE_AUX, bool CHAIN = false>
    HBVX_HDM void fwd_soil(const float *p, float nz, float aux_sw0, float aux_ef0)
    {
        const float BETA = p[P_BETA], FC = p[P_FC], LP = p[P_LP];
        r = div_(SM, FC);
        sw0 = USE_AUX ? aux_sw0 : pow_unit_(r, BETA);
        sw = fmin_(fmax_(sw0, 0.0f), 1.0f);
        rt = RAIN + tosoil;
        rech = rt * sw;
        SM1 = ((SM + RAIN) + tosoil) - rech;
        e0 = SM1 - FC;
        exc = fmax_(e0, 0.0f);
        SM2 = SM1 - exc;
        lpfc = LP * FC;
        q = div_q_((CHAIN && !SAVE_POW) ? SM1 : SM2, lpfc);
        if (BETAET) ef0 = USE_AUX ? aux_ef0 : pow_unit_(q, p[P_BETAET]);
        else ef0 = q;
        ef = fmin_(fmax_(ef0, 0.0f), 1.0f);
        pe = PET * ef;
        ET = fmin_(SM2, pe);
        dd = SM2 - ET;
        // SM3 = max(SM2 - min(SM2, pe), nz) (hbv.py:479-480) without the minimum on the day-to-day chain: for
        // pe <= SM2 both forms subtract pe, otherwise the reference's difference is 0 and this one is negative, and
        // either way the lower clamp returns nz (nearzero >= 0, check_desc) -- the same bits, one dependent
        // instruction less per day on the soil wave.  ET and dd (the adjoint's predicate) stay as the reference's.
        SM3 = fmax_(SM2 - pe, nz);
    }

    // capillary rise (1.1p / 2.0): couples SM3 and SLZ                 (hbv_1_1p.py:482-490)
    HBVX_HDM void fwd_cap(const float *p, float nz)
    {
        if (MODEL != MODEL_HBV10) {
            const float C = p[P_C], FC = p[P_FC];
            x1 = div_(SM3, FC);
            float rc = fmin_(x1, 1.0f);
            cs = C * SLZ;
            om = 1.0f - rc;
            capp = cs * om;
            cap = fmin_(SLZ, capp);
            smc = SM3 + cap;
            SM4 = fmax_(smc, nz);
            slc = SLZ - cap;
            SLZ0 = fmax_(slc, nz);
        } else {
            cap = 0.0f;
            SM4 = SM3;
            SLZ0 = SLZ;
        }
    }

    // groundwater: needs SUZ, SLZ0, rech, exc; produces PERC, Q0, Q1, Q2, SUZ4, SLZ2   (hbv.py:483-494)
    HBVX_HDM void fwd_gw(const float *p, float ac)
    {
        const float K0 = p[P_K0], K1 = p[P_K1], K2 = p[P_K2], PERCp = p[P_PERC], UZL = p[P_UZL];
        SUZ1 = (SUZ + rech) + exc;
        PERC = fmin_(SUZ1, PERCp);
        SUZ2 = SUZ1 - PERC;
        u0 = SUZ2 - UZL;
        u0c = fmax_(u0, 0.0f);
        Q0 = K0 * u0c;
        SUZ3 = SUZ2 - Q0;
        Q1 = K1 * SUZ3;
        SUZ4 = SUZ3 - Q1;
        SLZ1 = SLZ0 + PERC;
        if (MODEL == MODEL_HBV20) { // hbv_2.py:545-550
            const float RT = p[P_RT], AC = p[P_AC];
            m1 = (ac < 2500.0f) ? 1.0f : 0.0f;
            m2 = (ac >= 2500.0f) ? 1.0f : 0.0f;
            a0 = (ac - AC) / 1000.0f;
            a1 = fmin_(fmax_(a0, -1.0f), 1.0f);
            float e0_ = -(ac - 2500.0f) / 50.0f;
            float e1_ = fmin_(fmax_(e0_, -10.0f), 0.0f);
            ee = expf(e1_);
            float LF = (a1 * RT) * m1 + (ee * RT) * m2;
            sl = SLZ1 + LF;
            SLZ1p = fmax_(sl, 0.0f);
        } else {
            SLZ1p = SLZ1;
        }
        Q2 = K2 * SLZ1p;
        SLZ2 = SLZ1p - Q2;
        Q = (Q0 + Q1) + Q2; // hbv.py:494
    }

    template <bool USE_AUX, bool CHAIN = false>
    HBVX_HDM void fwd(const float *p, float nz, float ac, float elev, float aux_sw0, float aux_ef0)
    {
        fwd_snow(p, elev);
        fwd_soil<USE_AUX, CHAIN>(p, nz, aux_sw0, aux_ef0);
        fwd_cap(p, nz);
        fwd_gw(p, ac);
    }

    // Adjoint of fwd().  a[5]: dL/d(new states) in, dL/d(old states) out.
    // gp[]: += dL/d(physical parameter).  gx[3] = dL/d(P, T, PET).
    HBVX_HDM void bwd(const float *p, float nz, const FluxGrad &g, float *a, float *gp,
                     float *gx) const
    {
        HBVX_ADJ_FMA
        const float BETA = p[P_BETA], FC = p[P_FC], K0 = p[P_K0], K1 = p[P_K1], K2 = p[P_K2],
                    LP = p[P_LP], PERCp = p[P_PERC], CFMAX = p[P_CFMAX], CFR = p[P_CFR],
                    CWH = p[P_CWH];
        float wa, wb;
        float aSP3 = a[0] + g.gSWE, aMW3 = a[1], aSMn = a[2], aSUZ4 = a[3], aSLZ2 = a[4];
        float aQ0 = g.gQ0 + g.gQ, aQ1 = g.gQ1 + g.gQ, aQ2 = g.gQ2 + g.gQ;

        aQ2 -= aSLZ2;
        float aSLZ1p = aSLZ2 + aQ2 * K2;
        gp[P_K2] += aQ2 * SLZ1p;
        float aSLZ1 = aSLZ1p;
        if (MODEL == MODEL_HBV20) {
            const float RT = p[P_RT];
            float as = (sl >= 0.0f) ? aSLZ1p : 0.0f;
            aSLZ1 = as;
            float at1 = as * m1, at2 = as * m2;
            gp[P_RT] += at1 * a1 + at2 * ee;
            float aa1 = at1 * RT;
            float aa0 = (a0 >= -1.0f && a0 <= 1.0f) ? aa1 : 0.0f;
            gp[P_AC] += -(aa0 * 0.001f);     // (the adjoint of a0 = (ac - AC) / 1000: no IEEE division for a gradient)
        }
        float aSLZ0 = aSLZ1;
        float aPERC = g.gPERC + aSLZ1;
        aQ1 -= aSUZ4;
        float aSUZ3 = aSUZ4 + aQ1 * K1;
        gp[P_K1] += aQ1 * SUZ3;
        aQ0 -= aSUZ3;
        float aSUZ2 = aSUZ3;
        gp[P_K0] += aQ0 * u0c;
        float au0 = (u0 >= 0.0f) ? aQ0 * K0 : 0.0f;
        aSUZ2 += au0;
        gp[P_UZL] -= au0;
        aPERC -= aSUZ2;
        float aSUZ1 = aSUZ2;
        minw_(SUZ1, PERCp, wa, wb);
        aSUZ1 += aPERC * wa;
        gp[P_PERC] += aPERC * wb;
        float aSUZ = aSUZ1;
        float arech = g.grech + aSUZ1;
        float aexc = g.gexc + aSUZ1;

        float aSLZ, aSM3;
        if (MODEL != MODEL_HBV10) {
            const float C = p[P_C];
            float az = (slc >= nz) ? aSLZ0 : 0.0f;
            aSLZ = az;
            float acap = g.gcap - az;
            float ay = (smc >= nz) ? aSMn : 0.0f;
            aSM3 = ay;
            acap += ay;
            minw_(SLZ, capp, wa, wb);
            aSLZ += acap * wa;
            float acapp = acap * wb;
            float acs = acapp * om;
            float arc = -(acapp * cs);
            gp[P_C] += acs * SLZ;
            aSLZ += acs * C;
            float ax1 = (x1 <= 1.0f) ? arc : 0.0f;
            aSM3 += div_approx_(ax1, FC);
            gp[P_FC] += -ax1 * div_approx_(x1, FC);
        } else {
            aSLZ = aSLZ0;
            aSM3 = aSMn;
        }
        float add = (dd >= nz) ? aSM3 : 0.0f;
        float aSM2 = add;
        float aET = g.gET - add;
        minw_(SM2, pe, wa, wb);
        aSM2 += aET * wa;
        float ape = aET * wb;
        float aef = g.gef + ape * PET;
        gx[2] = ape * ef;
        float aef0 = (ef0 >= 0.0f && ef0 <= 1.0f) ? aef : 0.0f;
        float aq;
        if (BETAET) {
            // d/dq q^b = b q^(b-1) = b * ef0 / q ; d/db = ef0 ln q   (q > 0: SM2 > 0)
            const float BE = p[P_BETAET];
            float dq = (q > 0.0f) ? BE * div_approx_(ef0, q) : 0.0f;
            float db = (q > 0.0f) ? ef0 * log_fast_(q) : 0.0f;
            aq = aef0 * dq;
            gp[P_BETAET] += aef0 * db;
        } else {
            aq = aef0;
        }
        aSM2 += div_approx_(aq, lpfc);
        float alpfc = -aq * div_approx_(q, lpfc);
        gp[P_LP] += alpfc * FC;
        gp[P_FC] += alpfc * LP;
        float aSM1 = aSM2;
        aexc -= aSM2;
        float ae0 = (e0 >= 0.0f) ? aexc : 0.0f;
        aSM1 += ae0;
        gp[P_FC] -= ae0;
        float aSM = aSM1;
        float aRAIN = aSM1;
        float atosoil = g.gtosoil + aSM1;
        arech -= aSM1;
        float art = arech * sw;
        float asw = arech * rt;
        aRAIN += art;
        atosoil += art;
        float asw0 = (sw0 >= 0.0f && sw0 <= 1.0f) ? asw : 0.0f;
        float dr = (r > 0.0f) ? BETA * div_approx_(sw0, r) : 0.0f;
        float db_ = (r > 0.0f) ? sw0 * log_fast_(r) : 0.0f;
        float ar = asw0 * dr;
        gp[P_BETA] += asw0 * db_;
        aSM += div_approx_(ar, FC);
        gp[P_FC] += -ar * div_approx_(r, FC);
        float aMW2 = aMW3;
        atosoil -= aMW3;
        float ats0 = (ts0 >= 0.0f) ? atosoil : 0.0f;
        aMW2 += ats0;
        gp[P_CWH] -= ats0 * SP3;
        aSP3 -= ats0 * CWH;
        float aSP2 = aSP3;
        float arefr = aSP3 - aMW2;
        float aMW1 = aMW2;
        minw_(rpc, MW1, wa, wb);
        float arpc = arefr * wa;
        aMW1 += arefr * wb;
        float arp = (rp >= 0.0f) ? arpc : 0.0f;
        float acc = arp * dT2;
        gp[P_CFR] += acc * CFMAX;
        gp[P_CFMAX] += acc * CFR;
        float aTTe = arp * cc;
        float aTf = -(arp * cc);
        float aMW = aMW1;
        float amelt = aMW1 - aSP2;
        float aSP1 = aSP2;
        minw_(mpc, SP1, wa, wb);
        float ampc = amelt * wa;
        aSP1 += amelt * wb;
        float amp = (mp >= 0.0f) ? ampc : 0.0f;
        gp[P_CFMAX] += amp * dT;
        aTTe -= amp * CFMAX;
        aTf += amp * CFMAX;
        float aSP = aSP1;
        gx[0] = aSP1 * m_snow + aRAIN * m_rain;
        gx[1] = aTf;
        gp[P_TT] += aTTe * mlo;

        a[0] = aSP; a[1] = aMW; a[2] = aSM; a[3] = aSUZ; a[4] = aSLZ;
    }

    // ---- transposed state Jacobian of one day, HBV 1.0 only --------------------------------
    // bwd() with zero flux adjoints is the linear map a -> J^T a.  In HBV 1.0 the day is a
    // feed-forward chain snow -> soil -> groundwater, so J^T is block upper-triangular:
    //   out[SUZ,SLZ] from a[SUZ,SLZ];  out[SM] from a[SM], a[SUZ,SLZ];  out[SP,MW] from all.
    // jt_coef() extracts the handful of coefficients once per day (same predicates and tie
    // weights as bwd()), jt_unit<LEVEL>() applies them to one adjoint vector:
    //   LEVEL 0: a[2..4] == 0 (stays in the snow block)   LEVEL 1: a[3..4] == 0   LEVEL 2: full.
    // Used by the time-parallel adjoint to propagate the five unit adjoints of a chunk at ~1/4
    // of the cost of five bwd() calls.
    struct JT {
        float cS, cU, wP;          // groundwater: out[SLZ] = cS a4; U = cU a3 + wP out[SLZ]
        float kap, me, rho, sw;    // soil
        float mts, cwh, wr, wm;    // snow
        float K1, K2, k1c, k0c, k0m, wac;   // groundwater factors of cU, for jt_affine
    };

    // snow block only (all daily variants share the snow stage): enough for jt_unit<0>
    HBVX_HDM JT jt_coef_snow(const float *p) const
    {
        JT c;
        float wa, wb;
        c.cS = c.cU = c.wP = c.kap = c.me = c.rho = c.sw = 0.0f;
        c.K1 = c.K2 = c.k1c = c.k0c = c.k0m = c.wac = 0.0f;
        c.mts = (ts0 >= 0.0f) ? 1.0f : 0.0f;
        c.cwh = p[P_CWH];
        minw_(rpc, MW1, wa, wb);
        c.wr = wb;
        minw_(mpc, SP1, wa, wb);
        c.wm = wb;
        return c;
    }

    HBVX_HDM JT jt_coef(const float *p, float nz) const
    {
        static_assert(MODEL == MODEL_HBV10, "jt_coef: HBV 1.0 only");
        const float BETA = p[P_BETA], FC = p[P_FC], K0 = p[P_K0], K1 = p[P_K1], K2 = p[P_K2];
        JT c;
        float wa, wb;
        c.cS = 1.0f - K2;
        minw_(SUZ1, p[P_PERC], wa, wb);
        const float m0k0 = (u0 >= 0.0f) ? K0 : 0.0f;
        c.cU = ((1.0f - K1) * (1.0f - m0k0)) * (1.0f - wa);
        c.wP = wa;
        c.K1 = K1; c.K2 = K2; c.k1c = 1.0f - K1; c.k0c = 1.0f - m0k0; c.k0m = m0k0; c.wac = 1.0f - wa;
        // soil
        minw_(SM2, pe, wa, wb);
        const float mef = (ef0 >= 0.0f && ef0 <= 1.0f) ? 1.0f : 0.0f;
        float dq = mef;
        if (BETAET) dq = (q > 0.0f) ? mef * (p[P_BETAET] * div_approx_(ef0, q)) : 0.0f;
        const float md = (dd >= nz) ? 1.0f : 0.0f;
        c.kap = md * ((1.0f - wa) - (wb * PET) * div_approx_(dq, lpfc));
        c.me = (e0 >= 0.0f) ? 1.0f : 0.0f;
        const float msw = (sw0 >= 0.0f && sw0 <= 1.0f) ? 1.0f : 0.0f;
        const float dr = (r > 0.0f) ? BETA * div_approx_(sw0, r) : 0.0f;
        c.rho = div_approx_((rt * msw) * dr, FC);
        c.sw = sw;
        // snow
        c.mts = (ts0 >= 0.0f) ? 1.0f : 0.0f;
        c.cwh = p[P_CWH];
        minw_(rpc, MW1, wa, wb);
        c.wr = wb;
        minw_(mpc, SP1, wa, wb);
        c.wm = wb;
        return c;
    }

    // ---- the same for the capillary models (HBV 1.1p / 2.0) ---------------------------------
    // Capillary rise feeds the lower zone back into the soil, so J^T couples the SM row with the
    // groundwater block: out[SLZ] takes a share of a[SM], out[SM] a share of a[SLZ].  The snow block
    // still stands alone (jt_unit<0> with jt_coef_snow / these coefficients).  From bwd() with zero
    // flux adjoints, line for line:
    //   sl = msl (1-K2) a4                       (HBV 2.0: msl = [SLZ + LF >= 0], else 1)
    //   U  = cU a3 + wP sl                       = out[SUZ]
    //   az = mz sl, ay = my a2                   (the two lower clamps after the capillary exchange)
    //   out[SLZ] = az + kc (ay - az)             kc = w_SLZ + w_cap (1 - min(SM/FC, 1)) C
    //   sm3      = ay - ks (ay - az)             ks = [SM/FC <= 1] w_cap C SLZ / FC
    //   then the soil and snow rows of jt_unit<2> with sm3 in place of a2.
    struct JTC : JT {
        float msl, mz, my, kc, ks;
    };

    HBVX_HDM JTC jt_coef_cap(const float *p, float nz) const
    {
        static_assert(MODEL == MODEL_HBV11P || MODEL == MODEL_HBV20, "jt_coef_cap: HBV 1.1p / 2.0");
        const float BETA = p[P_BETA], FC = p[P_FC], K0 = p[P_K0], K1 = p[P_K1], K2 = p[P_K2], C = p[P_C];
        JTC c;
        float wa, wb;
        c.cS = 1.0f - K2;
        c.msl = 1.0f;
        if (MODEL == MODEL_HBV20) c.msl = (sl >= 0.0f) ? 1.0f : 0.0f;
        minw_(SUZ1, p[P_PERC], wa, wb);
        const float m0k0 = (u0 >= 0.0f) ? K0 : 0.0f;
        c.cU = ((1.0f - K1) * (1.0f - m0k0)) * (1.0f - wa);
        c.wP = wa;
        c.K1 = K1; c.K2 = K2; c.k1c = 1.0f - K1; c.k0c = 1.0f - m0k0; c.k0m = m0k0; c.wac = 1.0f - wa;
        // capillary exchange
        c.mz = (slc >= nz) ? 1.0f : 0.0f;
        c.my = (smc >= nz) ? 1.0f : 0.0f;
        minw_(SLZ, capp, wa, wb);
        c.kc = wa + (wb * om) * C;
        c.ks = (x1 <= 1.0f) ? div_approx_(wb * cs, FC) : 0.0f;
        // soil (BETAET always on in these models)
        minw_(SM2, pe, wa, wb);
        const float mef = (ef0 >= 0.0f && ef0 <= 1.0f) ? 1.0f : 0.0f;
        const float dq = (q > 0.0f) ? mef * (p[P_BETAET] * div_approx_(ef0, q)) : 0.0f;
        const float md = (dd >= nz) ? 1.0f : 0.0f;
        c.kap = md * ((1.0f - wa) - (wb * PET) * div_approx_(dq, lpfc));
        c.me = (e0 >= 0.0f) ? 1.0f : 0.0f;
        const float msw = (sw0 >= 0.0f && sw0 <= 1.0f) ? 1.0f : 0.0f;
        const float dr = (r > 0.0f) ? BETA * div_approx_(sw0, r) : 0.0f;
        c.rho = div_approx_((rt * msw) * dr, FC);
        c.sw = sw;
        // snow
        c.mts = (ts0 >= 0.0f) ? 1.0f : 0.0f;
        c.cwh = p[P_CWH];
        minw_(rpc, MW1, wa, wb);
        c.wr = wb;
        minw_(mpc, SP1, wa, wb);
        c.wm = wb;
        return c;
    }

    // a <- J^T a (+ the runoff-series sources s0 = gQ0 + gQ, s1 = gQ1 + gQ, s2 = gQ2 + gQ; zeros for a unit)
    static HBVX_HDM void jt_cap(const JTC &c, float *a, float s0, float s1, float s2)
    {
        HBVX_ADJ_FMA
        const float sl = c.msl * (c.cS * a[4] + c.K2 * s2);
        const float z3 = c.k1c * a[3] + c.K1 * s1;
        const float z2 = z3 * c.k0c + c.k0m * s0;
        const float U = z2 * c.wac + c.wP * sl;
        const float az = c.mz * sl, ay = c.my * a[2];
        const float dc = ay - az;
        a[3] = U;
        a[4] = az + c.kc * dc;
        const float sm3 = ay - c.ks * dc;
        const float s2_ = c.kap * sm3;
        const float s1_ = s2_ + c.me * (U - s2_);
        const float w = U - s1_;
        a[2] = s1_ + c.rho * w;
        const float ats = s1_ + c.sw * w;
        const float t = c.mts * (ats - a[1]);
        const float mw2 = a[1] + t;
        const float sp2 = a[0] - t * c.cwh;
        const float mw1 = mw2 + (sp2 - mw2) * c.wr;
        a[1] = mw1;
        a[0] = sp2 + (mw1 - sp2) * c.wm;
    }

    // a <- J^T a + c(g) when only the runoff series carry gradient (loss on streamflow):
    // s0 = gQ0 + gQ, s1 = gQ1 + gQ, s2 = gQ2 + gQ enter through the groundwater block, the rest
    // is jt_unit<2>.  Replaces a full bwd() for the offset vector of the chunk maps.
    static HBVX_HDM void jt_affine(const JT &c, float *a, float s0, float s1, float s2)
    {
        HBVX_ADJ_FMA
        const float sl = c.cS * a[4] + c.K2 * s2;
        const float z3 = c.k1c * a[3] + c.K1 * s1;
        const float z2 = z3 * c.k0c + c.k0m * s0;
        const float U = z2 * c.wac + c.wP * sl;
        a[3] = U;
        a[4] = sl;
        const float s2_ = c.kap * a[2];
        const float s1_ = s2_ + c.me * (U - s2_);
        const float w = U - s1_;
        a[2] = s1_ + c.rho * w;
        const float ats = s1_ + c.sw * w;
        const float t = c.mts * (ats - a[1]);
        const float mw2 = a[1] + t;
        const float sp2 = a[0] - t * c.cwh;
        const float mw1 = mw2 + (sp2 - mw2) * c.wr;
        a[1] = mw1;
        a[0] = sp2 + (mw1 - sp2) * c.wm;
    }

    template <int LEVEL>
    static HBVX_HDM void jt_unit(const JT &c, float *a)
    {
        HBVX_ADJ_FMA
        float ats = 0.0f;
        if (LEVEL >= 1) {
            float U = 0.0f;
            if (LEVEL >= 2) {
                const float sl = c.cS * a[4];
                U = c.cU * a[3] + c.wP * sl;
                a[3] = U;
                a[4] = sl;
            }
            const float s2 = c.kap * a[2];
            const float s1 = s2 + c.me * (U - s2);
            const float w = U - s1;
            a[2] = s1 + c.rho * w;
            ats = s1 + c.sw * w;
        }
        const float t = c.mts * (ats - a[1]);
        const float mw2 = a[1] + t;
        const float sp2 = a[0] - t * c.cwh;
        const float mw1 = mw2 + (sp2 - mw2) * c.wr;
        a[1] = mw1;
        a[0] = sp2 + (mw1 - sp2) * c.wm;
    }
};

} // namespace hbvx

#include "hbv_step_hourly.h"
