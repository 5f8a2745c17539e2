// hbv_adj_step.h -- one implicit (backward-Euler) HBV day and its implicit-function adjoint.
//
// Restates the algorithm the reference's `HbvAdj` intends (the file itself is not runnable,
// SURVEY.md §2 #13):
//   RHS f(x, theta, t) in flux form from clamped storages        hbv_adj.py:385-431,444-498
//   residual G(x) = (x - x_t)/dt - f(x)                           hbv_adj.py:669-678
//   modified Newton, <= max_iter+1 updates, gtol on |G|_inf,
//   Jacobian refreshed only when res/res_old > 0.2                hbv_adj.py:516-581
//   adjoint (dG/dx)^T lam = dL/dx, dL/dtheta = -lam^T dG/dtheta   hbv_adj.py:617-633
//
// dG/dx = I/dt - df/dx is block lower-triangular (snow 2x2 -> soil -> upper -> lower box), so the
// 5x5 solve is a forward substitution in registers (the reference calls torch.linalg.solve on a
// batched dense matrix, hbv_adj.py:501,570); the transpose solve of the adjoint is the matching
// back substitution.  Derivatives are analytic with autograd's conventions (minimum() ties 1/2,
// inclusive clamps) where the reference uses an autograd Jacobian (hbv_adj.py:531,557) and
// finite differences for dG/dtheta (hbv_adj.py:606).
//
// Stopping rule: per LANE by default.  The reference's `torch.max(resnorm)` (hbv_adj.py:544) couples
// every basin x member of the batch through one global maximum, so all lanes take as many updates
// as the slowest; `vote` reproduces that over the 64 lanes of a wavefront (hbvx_desc.adj_stop = 1) --
// a batch-wide maximum would cost a grid barrier per Newton update and day.
#pragma once

#include "hbv_step.h"

namespace hbvx {

template <bool BETAET>
struct AdjStep {
    // inputs of the day
    float P, Tf, PET;
    // clamped storages and clamp masks (hbv_adj.py:387-391)
    float SP, MW, SM, SUZ, SLZ, c0, c1, c2, c3, c4;
    // fluxes
    float sf, rf, refr, melt, Isnow, sw0, sw, Peff, ex, ef0, ef, pe, et, perc, u0c, q0, q1, q2;
    // local partials / branch weights
    float rp, mp, war, wbr, wam, wbm, mI, msw, r, dsw, mex, qe, mef, wae, wbe, dE, wap, wbp, mq0;
    float dT, dT2, cc;
    // df/dy (sparse)
    float F00, F01, F10, F11, F20, F21, F22, F30, F31, F32, F33, F43, F44;
    float f[5];

    // f(y) and, if JAC, df/dy.  hbv_adj.py:385-431
    template <bool JAC>
    HBVX_HDM void eval(const float *y, const float *p)
    {
        const float BETA = p[P_BETA], FC = p[P_FC], K0 = p[P_K0], K1 = p[P_K1], K2 = p[P_K2],
                    LP = p[P_LP], PERCp = p[P_PERC], UZL = p[P_UZL], TT = p[P_TT],
                    CFMAX = p[P_CFMAX], CFR = p[P_CFR], CWH = p[P_CWH];
        SP = fmax_(y[0], 0.0f);  c0 = (y[0] >= 0.0f) ? 1.0f : 0.0f;
        MW = fmax_(y[1], 0.0f);  c1 = (y[1] >= 0.0f) ? 1.0f : 0.0f;
        SM = fmax_(y[2], 1e-8f); c2 = (y[2] >= 1e-8f) ? 1.0f : 0.0f;
        SUZ = fmax_(y[3], 0.0f); c3 = (y[3] >= 0.0f) ? 1.0f : 0.0f;
        SLZ = fmax_(y[4], 0.0f); c4 = (y[4] >= 0.0f) ? 1.0f : 0.0f;
        // :444-462 snowfall / rainfall
        sf = P * ((Tf < TT) ? 1.0f : 0.0f);
        rf = P * ((Tf >= TT) ? 1.0f : 0.0f);
        // :448-452 refreezing
        cc = CFR * CFMAX;
        dT2 = TT - Tf;
        rp = cc * dT2;
        const float rpc = fmax_(rp, 0.0f);
        refr = fmin_(rpc, MW);
        // :454-458 melt
        dT = Tf - TT;
        mp = CFMAX * dT;
        const float mpc = fmax_(mp, 0.0f);
        melt = fmin_(mpc, SP);
        // :464-468 meltwater to soil
        const float ts0 = MW - CWH * SP;
        Isnow = fmax_(ts0, 0.0f);
        // :470-474 effective precipitation
        r = div_(SM, FC);
        sw0 = pow_unit_(r, BETA);
        sw = fmin_(fmax_(sw0, 0.0f), 1.0f);
        const float rt = rf + Isnow;
        Peff = rt * sw;
        // :476-479 excess
        const float e0 = SM - FC;
        ex = fmax_(e0, 0.0f);
        // :481-486 evapotranspiration
        const float lpfc = LP * FC;
        qe = div_(SM, lpfc);
        ef0 = BETAET ? pow_unit_(qe, p[P_BETAET]) : qe;
        ef = fmin_(fmax_(ef0, 0.0f), 1.0f);
        pe = PET * ef;
        et = fmin_(SM, pe);
        // :488-498
        perc = fmin_(SUZ, PERCp);
        const float u0 = SUZ - UZL;
        u0c = fmax_(u0, 0.0f);
        q0 = K0 * u0c;
        q1 = K1 * SUZ;
        q2 = K2 * SLZ;
        // :425-429
        f[0] = (sf + refr) - melt;
        f[1] = (melt - refr) - Isnow;
        f[2] = (((Isnow + rf) - Peff) - ex) - et;
        f[3] = (((Peff + ex) - perc) - q0) - q1;
        f[4] = perc - q2;
        if (JAC) {
            minw_(rpc, MW, war, wbr);
            minw_(mpc, SP, wam, wbm);
            mI = (ts0 >= 0.0f) ? 1.0f : 0.0f;
            msw = (sw0 >= 0.0f && sw0 <= 1.0f) ? 1.0f : 0.0f;
            dsw = msw * (BETA * div_approx_(sw0, SM));       // d sw / d SM
            mex = (e0 >= 0.0f) ? 1.0f : 0.0f;
            mef = (ef0 >= 0.0f && ef0 <= 1.0f) ? 1.0f : 0.0f;
            minw_(SM, pe, wae, wbe);
            const float def = BETAET ? mef * (p[P_BETAET] * div_approx_(ef0, SM)) : mef * div_approx_(1.0f, lpfc);
            dE = wae + wbe * (PET * def);                      // d et / d SM
            minw_(SUZ, PERCp, wap, wbp);
            mq0 = (u0 >= 0.0f) ? 1.0f : 0.0f;
            const float dP = rt * dsw;                          // d Peff / d SM
            F00 = -wbm * c0;                 F01 = wbr * c1;
            F10 = (wbm + CWH * mI) * c0;     F11 = (-wbr - mI) * c1;
            F20 = (-(CWH * mI)) * (1.0f - sw) * c0;
            F21 = mI * (1.0f - sw) * c1;
            F22 = (-dP - mex - dE) * c2;
            F30 = (-(CWH * mI)) * sw * c0;
            F31 = (mI * sw) * c1;
            F32 = (dP + mex) * c2;
            F33 = (-wap - K0 * mq0 - K1) * c3;
            F43 = wap * c3;
            F44 = -K2 * c4;
        }
    }

    // Solve (I/dt - F) dx = g by forward substitution (block lower-triangular).
    HBVX_HDM void solve(float idt, const float *g, float *dx) const
    {
        const float J00 = idt - F00, J01 = -F01, J10 = -F10, J11 = idt - F11;
        const float det = J00 * J11 - J01 * J10;
        dx[0] = div_approx_(g[0] * J11 - J01 * g[1], det);
        dx[1] = div_approx_(J00 * g[1] - g[0] * J10, det);
        dx[2] = div_approx_(g[2] + F20 * dx[0] + F21 * dx[1], idt - F22);
        dx[3] = div_approx_(g[3] + F30 * dx[0] + F31 * dx[1] + F32 * dx[2], idt - F33);
        dx[4] = div_approx_(g[4] + F43 * dx[3], idt - F44);
    }

    // Solve (I/dt - F)^T lam = a by back substitution.
    HBVX_HDM void solve_t(float idt, const float *a, float *lam) const
    {
        HBVX_ADJ_FMA
        lam[4] = div_approx_(a[4], idt - F44);
        lam[3] = div_approx_(a[3] + F43 * lam[4], idt - F33);
        lam[2] = div_approx_(a[2] + F32 * lam[3], idt - F22);
        const float b0 = a[0] + F20 * lam[2] + F30 * lam[3];
        const float b1 = a[1] + F21 * lam[2] + F31 * lam[3];
        const float J00 = idt - F00, J01 = -F01, J10 = -F10, J11 = idt - F11;
        const float det = J00 * J11 - J01 * J10;
        // transpose of the 2x2 block: [J00 J10; J01 J11]
        lam[0] = div_approx_(b0 * J11 - J10 * b1, det);
        lam[1] = div_approx_(J00 * b1 - b0 * J01, det);
    }
};

// One implicit day: x <- solution of G(x) = (x - xt)/dt - f(x) = 0 starting from xt.
// Returns the number of Newton updates taken.  hbv_adj.py:516-581
HBVX_HD bool adj_any_(bool pred, bool vote)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return vote ? __builtin_amdgcn_ballot_w64(pred) != 0 : pred;
#else
    (void)vote;
    return pred;
#endif
}

template <bool BETAET>
HBVX_HD int adj_newton(AdjStep<BETAET> &s, const float *p, const float *xt, float idt, float gtol,
                       int max_iter, float *x, bool vote = false)
{
    float g[5], dx[5];
#pragma unroll
    for (int k = 0; k < 5; k++) x[k] = xt[k];
    s.template eval<true>(x, p);
    float res = 0.0f;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        g[k] = (x[k] - xt[k]) * idt - s.f[k];
        res = fmax_(res, fabsf(g[k]));
    }
    float res0 = 100.0f * res;
    // `e` always holds f and df/dx at the current iterate (one evaluation per iterate: the RHS costs
    // two pows); `s` holds the Jacobian the modified Newton step uses, refreshed from `e` only when
    // the residual ratio exceeds 0.2 (hbv_adj.py:546).  Same numbers as evaluating G and then, for a
    // refresh, the Jacobian again at the same point.
    AdjStep<BETAET> e = s;
    int it = 0;
    while (adj_any_(res > gtol, vote) && it <= max_iter) {
        it++;
        if (adj_any_(res > 0.2f * res0, vote)) {
            s.F00 = e.F00; s.F01 = e.F01; s.F10 = e.F10; s.F11 = e.F11; s.F20 = e.F20; s.F21 = e.F21;
            s.F22 = e.F22; s.F30 = e.F30; s.F31 = e.F31; s.F32 = e.F32; s.F33 = e.F33; s.F43 = e.F43;
            s.F44 = e.F44;
        }
        s.solve(idt, g, dx);
#pragma unroll
        for (int k = 0; k < 5; k++) x[k] = x[k] - dx[k];
        e.template eval<true>(x, p);
        res0 = res;
        res = 0.0f;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            g[k] = (x[k] - xt[k]) * idt - e.f[k];
            res = fmax_(res, fabsf(g[k]));
        }
    }
    return it;
}

// ---------------------------------------------------------------------------------------------
// Staged solve (hbvx_desc.adj_stop = 2): the same equations G(x) = 0, solved block by block.
//
// dG/dx is block lower-triangular because f is (hbv_adj.py:425-429: dS0, dS1 read SNOWPACK and MELTWATER
// only; dS2 adds SM; dS3 reads SM and SUZ; dS4 reads SUZ and SLZ).  The NONLINEAR system therefore
// decouples the same way: (y0, y1) -> y2 -> y3 -> y4, each block an equation in its own unknowns with
// the upstream blocks already solved.  Three of the four blocks are piecewise linear and monotone, so
// they have closed forms (the branch is chosen by the sign of the block residual at the kinks, the
// solution is then one linear solve); only the soil-moisture equation carries the two powers and keeps
// the reference's Newton policy (stop at |G2| <= gtol, at most max_iter + 1 updates, hbv_adj.py:518-
// 519,544 -- with the Jacobian always fresh: it is one scalar that the residual evaluation already
// produced).  What the reference's joint iteration returns is ANY iterate with |G|_inf <= gtol (or the
// fourth update); this solve returns a state with G0 = G1 = G3 = G4 = 0 up to rounding and |G2| <= gtol:
// the same acceptance test, reached along a different path, and the three waves of a workgroup can run
// the blocks of consecutive days as a pipeline (hbv_pipe.h) instead of one wave iterating on all five
// unknowns.  The joint modified Newton above stays as adj_stop = 0 / 1 (policy cross-check).
//
// Invariant: storages entering a day are >= 0 (zeros at the start, hbv_adj.py:254, and every closed
// form below returns >= 0 for inputs >= 0).  The flux expressions are those of AdjStep::eval, in the
// same association order, so the adjoint's re-evaluation at the solved state sees the same branches.
#ifndef ADJ_SOIL_POW
#define ADJ_SOIL_POW pow_fast_   // -DADJ_SOIL_POW=pow_step_: the 2-ulp power of the explicit models (A/B runs)
#endif
#ifndef ADJ_SOIL_HALLEY
#define ADJ_SOIL_HALLEY 1
#endif
#ifndef ADJ_SOIL_PEEL
#define ADJ_SOIL_PEEL 1
#endif
#ifndef ADJ_SOIL_CLAMP
#define ADJ_SOIL_CLAMP 3
#endif
#ifndef ADJ_SOIL_FREE
#define ADJ_SOIL_FREE 2      // updates taken as plain Halley steps before the bracket is built (0: safeguard every update)
#endif
template <bool BETAET>
struct AdjStaged {
    // ---- block 1: snow.  (y0 - y0t)/dt = sf + refr - melt,  (y1 - y1t)/dt = melt - refr - Isnow
    //      hbv_adj.py:425-426,444-468.  melt > 0 needs T > TT and refr > 0 needs T < TT: never both.
    static HBVX_HDM void snow(const float *p, float P, float Tf, float dt, float y0t, float y1t,
                              float &y0, float &y1, float &rf, float &Isnow)
    {
        const float TT = p[P_TT], CFMAX = p[P_CFMAX], CFR = p[P_CFR], CWH = p[P_CWH];
        const float sf = P * ((Tf < TT) ? 1.0f : 0.0f);
        rf = P * ((Tf >= TT) ? 1.0f : 0.0f);
        const float cc = CFR * CFMAX;
        const float rpc = fmax_(cc * (TT - Tf), 0.0f);
        const float mpc = fmax_(CFMAX * (Tf - TT), 0.0f);
        const float r1d = div_approx_(1.0f, 1.0f + dt);
        const float a0 = y0t + dt * sf;
        // warm day (refr = 0):  y0 + dt min(mpc, max(y0, 0)) = a0, increasing in y0: root <= k <=> value at k >= 0
        const float y0w = (a0 <= 0.0f) ? a0 : ((a0 <= mpc * (1.0f + dt)) ? a0 * r1d : a0 - dt * mpc);
        const float SPw = fmax_(y0w, 0.0f);
        const float meltw = fmin_(mpc, SPw);
        //   y1 + dt max(max(y1, 0) - CWH SP, 0) = b1
        const float b1 = y1t + dt * meltw, cw = CWH * SPw;
        const float y1w = (b1 <= cw) ? b1 : (b1 + dt * cw) * r1d;
        // cold day (melt = 0):  y0 = a0 + dt refr,  y1 + dt min(rpc, y1+) + dt max(y1+ - CWH max(y0, 0), 0) = y1t
        //   (increasing in y1 while dt CWH < 1).  Kinks: y1 = 0, y1 = rpc, and where meltwater starts to drain
        const float ap = fmax_(a0, 0.0f);
        const float S2 = fmax_(a0 + dt * rpc, 0.0f);                  // SNOWPACK once refr = rpc
        const bool rmode = ((rpc + dt * rpc) + dt * fmax_(rpc - CWH * S2, 0.0f)) - y1t >= 0.0f;   // root <= rpc: refr = y1
        const float k1 = div_approx_(CWH * ap, 1.0f - dt * CWH);        // y1 = CWH (a0 + dt y1)
        const bool act = rmode ? (k1 * (1.0f + dt) < y1t) : (CWH * S2 + dt * rpc < y1t);          // Isnow > 0 at the root
        const float y1ra = div_approx_(y1t + dt * (CWH * ap), (1.0f + 2.0f * dt) - (dt * dt) * CWH);
        const float y1r = act ? y1ra : y1t * r1d;
        const float y1f = act ? ((y1t - dt * rpc) + dt * (CWH * S2)) * r1d : y1t - dt * rpc;
        const float y1c = (y1t <= 0.0f) ? y1t : (rmode ? y1r : y1f);
        const float y0c = a0 + dt * fmin_(rpc, fmax_(y1c, 0.0f));
        const bool warm = mpc > 0.0f;
        y0 = warm ? y0w : y0c;
        y1 = warm ? y1w : y1c;
        Isnow = fmax_(fmax_(y1, 0.0f) - CWH * fmax_(y0, 0.0f), 0.0f);   // as AdjStep::eval (:464-468)
    }

    // ---- block 2: soil moisture.  (y2 - y2t)/dt = Isnow + rf - Peff - ex - et   (hbv_adj.py:427,470-486)
    //      G2 is increasing in y2 (Peff, ex and et all grow with SM), so the root is bracketed: it lies in
    //      [y2t - dt (ex + et)(y2t), y2t + dt (Isnow + rf)], and on the FC side that G2(FC) = (FC - y2t)/dt +
    //      min(FC, PET) names (at SM = FC the wetness is 1 and, LP <= 1, so is the evaporation factor: no power
    //      needed).  Newton from y2t with Halley's correction under the reference's policy (stop at |G2| <= gtol,
    //      at most max_iter + 1 updates).  Returns the update count; Peff, ex: at the accepted state.
    //
    //      The soil-moisture wave of the pipelined forward is a lone wave on its SIMD and pays ~8 cycles for EVERY
    //      instruction of this function, on or off the dependency chain (round 4: eleven builds timed against each
    //      other, profiles/r04_ab_soil_parts.txt) -- so the common day must be short, not clever:
    //        * the first ADJ_SOIL_FREE (2) updates are Halley steps with ONE safeguard, the one the hard days need: a
    //          step that would pass FC from a day whose root lies below it lands ON FC (a melt pulse of hundreds of
    //          mm into a small store: plain Newton cycles across that kink, slope 1 + In BETA / FC below and 2 above;
    //          below FC the residual is convex, so Newton from the right end then converges monotonically).  On
    //          7.5e6 lane-days of bench-shaped forcing (tools/micro/adj_soil_stats.cpp) 97.1 % of the lanes are done
    //          after one update and all but 0.0005 % after two;
    //        * a lane that is still unsolved after them finishes inside the full bracket -- built only then, from the
    //          outflow at y2t kept from the first evaluation and the side of y2t its residual named: every further
    //          update stays inside it (a step that leaves it becomes its midpoint) and the bracket shrinks with the
    //          sign of each residual.  tests/test_step_math_host.py: 200 000 adversarial days, |G2| <= gtol within
    //          the four updates on every one.
    //      Measured and not kept (same file): a third-derivative term in the update (Householder order 3), a
    //      Newton finish without the confirming evaluation for residuals just above gtol, and a start AT the kink
    //      of the evaporation factor for days whose root lies across it -- together they take the wave-days that
    //      need a second update from 54 % to 0.1 %, and each costs the common day as much as it saves.
    static HBVX_HDM int soil(const float *p, float rf, float Isnow, float PET, float idt, float y2t,
                             float gtol, int max_iter, float &y2, float &Peff, float &ex)
    {
        const float BETA = p[P_BETA], FC = p[P_FC], LP = p[P_LP];
        const float BE = BETAET ? p[P_BETAET] : 1.0f;
        const float lpfc = LP * FC;
        const float rt = rf + Isnow;
        // day constants (off the iteration's dependency chain)
        const float rFC = div_approx_(1.0f, FC), rLF = div_approx_(1.0f, lpfc);
        const float rtB = rt * BETA, B1 = BETA - 1.0f, PB = BETAET ? PET * BE : PET * rLF, E1 = BE - 1.0f;
        float x = y2t, g, J, H, et;
        // Residual, slope and curvature at x in ONE straight block.  Values as in AdjStep::eval up to the rounding of
        // the quotients (SM/FC by reciprocal: the iteration's acceptance test is 1e-3, and SM <= FC decides the
        // wetness clamp instead of the power's side of 1).
        auto eval = [&]() __attribute__((always_inline)) {
            const float SM = fmax_(x, 1e-8f);
            const float rSM = div_approx_(1.0f, SM);
            const float sw0 = ADJ_SOIL_POW(SM * rFC, BETA);
            const float qe = SM * rLF;
            const float ef0 = BETAET ? ADJ_SOIL_POW(qe, BE) : qe;
            const float e0 = SM - FC;
            Peff = rt * fmin_(sw0, 1.0f);
            ex = fmax_(e0, 0.0f);
            const float pe = PET * fmin_(ef0, 1.0f);
            et = fmin_(SM, pe);
            const float f2 = (((Isnow + rf) - Peff) - ex) - et;
            g = (x - y2t) * idt - f2;
            // G2' and G2'' between the kinks (the iteration needs a slope, not autograd's tie conventions)
            const float dP = (e0 <= 0.0f) ? rtB * (sw0 * rSM) : 0.0f;                  // d Peff / d SM
            const float dEp = (qe <= 1.0f) ? (BETAET ? PB * (ef0 * rSM) : PB) : 0.0f;  // d (PET ef) / d SM
            const bool epow = pe < SM;                                                 // et = PET ef
            const bool live = x >= 1e-8f;        // below the clamp of hbv_adj.py:389 the fluxes do not move
            const float Jf = (dP + ((e0 >= 0.0f) ? 1.0f : 0.0f)) + (epow ? dEp : 1.0f);
            J = idt + (live ? Jf : 0.0f);
            // sw'' = (BETA - 1) sw' / SM, ef'' likewise: four multiplications
            H = (ADJ_SOIL_HALLEY && live) ? rSM * (B1 * dP + ((BETAET && epow) ? E1 * dEp : 0.0f)) : 0.0f;
        };
        // Newton's step g / J with Halley's correction g / (J - g H / 2J) -- the powers make G2 smooth between its
        // kinks, and the correction saves the second update (a whole residual evaluation, two powers) on most days --
        // as ONE reciprocal: g J / (J^2 - g H / 2); Newton's step is kept where the correction is large
        // (denominator below J^2 / 2)
        auto halley = [&]() __attribute__((always_inline)) -> float {
            const float JJ = J * J, den = JJ - 0.5f * (g * H);
            const bool hal = den > 0.5f * JJ;
            return (hal ? g * J : g) * div_approx_(1.0f, hal ? den : J);
        };
        eval();
        const float out0 = ex + et;     // outflow at y2t: the lower end of the bracket, should it be needed
        const bool up0 = g < 0.0f;      // G2(y2t) < 0: the root lies above y2t
        const bool below = (FC - y2t) * idt + fmin_(FC, PET) > 0.0f;   // G2(FC) > 0: root < FC
        int it = 0;
        bool more = fabsf(g) > gtol;
        // free updates: the first without asking the wave whether any lane still needs it (a wave vote and its branch
        // cost ~100 cycles a day; a day that is already solved -- no input, no evaporation -- is rare), the others
        // behind a vote; lanes that have converged keep their state
#pragma unroll
        for (int k = 0; k < ADJ_SOIL_FREE; k++) {
            if (k > max_iter || (k >= ADJ_SOIL_PEEL && !adj_any_(more, true))) break;
            it++;
            float xn = x - halley();
            xn = (below && xn > FC) ? FC : xn;      // a step past FC lands ON FC (see above)
            x = more ? xn : x;
            eval();
            more = fabsf(g) > gtol;
        }
        if (adj_any_(more, true) && it <= max_iter) {
            // the safeguarded finish.  The bracket holds whatever the free updates did: its ends come from y2t alone,
            // and the current iterate only ever tightens it by the sign of its residual.
            const float dt = (idt == 1.0f) ? 1.0f : div_approx_(1.0f, idt);
            float lo = y2t - dt * out0;
            float hi = y2t + dt * rt;
            bool hi_new = below && FC < hi;   // hi is a bound that has not been an iterate yet
            hi = below ? fmin_(hi, FC) : hi;
            lo = below ? lo : fmax_(lo, FC);
            lo = up0 ? fmax_(lo, y2t) : lo;
            hi = up0 ? hi : fmin_(hi, y2t);
            do {
                it++;
                const float dx = halley();
                lo = (g < 0.0f) ? fmax_(lo, x) : lo;
                hi = (g < 0.0f) ? hi : fmin_(hi, x);
                float xn = x - dx;
                const bool past = xn > hi && hi_new;
                xn = (xn >= lo && xn <= hi) ? xn : (past ? hi : 0.5f * (lo + hi));
                hi_new = hi_new && !past && more;
                x = more ? xn : x;
                eval();
                more = fabsf(g) > gtol;
            } while (adj_any_(more, true) && it <= max_iter);
        }
        y2 = x;
        return it;
    }

    // ---- blocks 3, 4: upper and lower zone.  (y3 - y3t)/dt = Peff + ex - perc - q0 - q1,
    //      (y4 - y4t)/dt = perc - q2   (hbv_adj.py:428-429,488-498).  Q = q0 + q1 + q2 (:431)
    static HBVX_HDM void gw(const float *p, float Peff, float ex, float dt, float y3t, float y4t,
                            float &y3, float &y4, float &Q)
    {
        const float K0 = p[P_K0], K1 = p[P_K1], K2 = p[P_K2], PERCp = p[P_PERC], UZL = p[P_UZL];
        const float b3 = y3t + dt * (Peff + ex);
        // n(y) = y + dt (min(y+, PERC) + K0 max(y+ - UZL, 0) + K1 y+) - b3, increasing: root <= k <=> n(k) >= 0
        const float nP = ((PERCp + dt * PERCp) + dt * (K0 * fmax_(PERCp - UZL, 0.0f))) + (dt * K1) * PERCp - b3;
        const float nU = (UZL + dt * fmin_(UZL, PERCp)) + (dt * K1) * UZL - b3;
        const bool lin = nP >= 0.0f;     // SUZ <= PERC: perc = SUZ
        const bool q0a = nU < 0.0f;      // SUZ > UZL: interflow active
        const float den = ((1.0f + dt * K1) + (lin ? dt : 0.0f)) + (q0a ? dt * K0 : 0.0f);
        const float num = (b3 - (lin ? 0.0f : dt * PERCp)) + (q0a ? (dt * K0) * UZL : 0.0f);
        y3 = (b3 <= 0.0f) ? b3 : div_approx_(num, den);
        const float SUZ = fmax_(y3, 0.0f);
        const float perc = fmin_(SUZ, PERCp);
        const float b4 = y4t + dt * perc;
        y4 = (b4 <= 0.0f) ? b4 : div_approx_(b4, 1.0f + dt * K2);
        const float SLZ = fmax_(y4, 0.0f);
        const float q0 = K0 * fmax_(SUZ - UZL, 0.0f);
        Q = (q0 + K1 * SUZ) + K2 * SLZ;
    }

    // one day, all blocks (single-wave kernels, host build)
    static HBVX_HDM int day(const float *p, float P, float Tf, float PET, const float *xt, float gtol,
                            int max_iter, float *x, float &Q)
    {
        float rf, Isnow, Peff, ex;
        snow(p, P, Tf, 1.0f, xt[0], xt[1], x[0], x[1], rf, Isnow);
        const int it = soil(p, rf, Isnow, PET, 1.0f, xt[2], gtol, max_iter, x[2], Peff, ex);
        gw(p, Peff, ex, 1.0f, xt[3], xt[4], x[3], x[4], Q);
        return it;
    }
};

// Implicit-function adjoint of one day at the solved state x (= y_{t+1}).
//   a[5]  in : dL/dx from the future (lam_{t+1}/dt);  out: dL/dx_t = lam/dt
//   gQ       : dL/dQ_t for this lane (Q = q0+q1+q2 evaluated at x, hbv_adj.py:309-313,431)
//   gp[]     += dL/d(physical parameters of day t)
template <bool BETAET>
HBVX_HD void adj_backstep(AdjStep<BETAET> &s, const float *p, const float *x, float idt, float gQ,
                          float *a, float *gp)
{
    HBVX_ADJ_FMA
    const float BETA = p[P_BETA], FC = p[P_FC], K0 = p[P_K0], K1 = p[P_K1], K2 = p[P_K2],
                LP = p[P_LP], CFMAX = p[P_CFMAX], CFR = p[P_CFR];
    s.template eval<true>(x, p);
    // direct dependence of Q on the solved state
    float rhs[5] = {a[0], a[1], a[2], a[3] + gQ * (K0 * s.mq0 + K1) * s.c3, a[4] + gQ * K2 * s.c4};
    float lam[5];
    s.solve_t(idt, rhs, lam);
    // adjoints of the fluxes: lam^T df + gQ dQ   (signs from hbv_adj.py:425-431)
    const float a_refr = lam[0] - lam[1];
    const float a_melt = lam[1] - lam[0];
    float a_Isnow = lam[2] - lam[1];
    const float a_Peff = lam[3] - lam[2];
    const float a_ex = lam[3] - lam[2];
    const float a_et = -lam[2];
    const float a_perc = lam[4] - lam[3];
    const float a_q0 = gQ - lam[3];
    const float a_q1 = gQ - lam[3];
    const float a_q2 = gQ - lam[4];
    // groundwater
    gp[P_K0] += a_q0 * s.u0c;
    gp[P_UZL] -= a_q0 * K0 * s.mq0;
    gp[P_K1] += a_q1 * s.SUZ;
    gp[P_K2] += a_q2 * s.SLZ;
    gp[P_PERC] += a_perc * s.wbp;
    // evapotranspiration: et = min(SM, PET * clamp((SM/(LP FC))^BETAET))
    const float a_ef0 = a_et * s.wbe * s.PET * s.mef;
    float a_qe;
    if (BETAET) {
        const float BE = p[P_BETAET];
        a_qe = (s.qe > 0.0f) ? a_ef0 * (BE * div_approx_(s.ef0, s.qe)) : 0.0f;
        gp[P_BETAET] += (s.qe > 0.0f) ? a_ef0 * (s.ef0 * log_fast_(s.qe)) : 0.0f;
    } else {
        a_qe = a_ef0;
    }
    const float lpfc = LP * FC;
    const float a_lpfc = -a_qe * div_approx_(s.qe, lpfc);
    gp[P_LP] += a_lpfc * FC;
    gp[P_FC] += a_lpfc * LP;
    // excess
    gp[P_FC] -= a_ex * s.mex;
    // effective precipitation
    const float rt = s.rf + s.Isnow;
    const float a_sw0 = a_Peff * rt * s.msw;
    gp[P_BETA] += (s.r > 0.0f) ? a_sw0 * (s.sw0 * log_fast_(s.r)) : 0.0f;
    const float a_r = (s.r > 0.0f) ? a_sw0 * (BETA * div_approx_(s.sw0, s.r)) : 0.0f;
    gp[P_FC] += -a_r * div_approx_(s.r, FC);
    a_Isnow += a_Peff * s.sw;
    // meltwater to soil
    gp[P_CWH] -= a_Isnow * s.mI * s.SP;
    // melt
    const float a_mp = (s.mp >= 0.0f) ? a_melt * s.wam : 0.0f;
    gp[P_CFMAX] += a_mp * s.dT;
    gp[P_TT] -= a_mp * CFMAX;
    // refreezing
    const float a_rp = (s.rp >= 0.0f) ? a_refr * s.war : 0.0f;
    const float a_cc = a_rp * s.dT2;
    gp[P_CFR] += a_cc * CFMAX;
    gp[P_CFMAX] += a_cc * CFR;
    gp[P_TT] += a_rp * s.cc;
#pragma unroll
    for (int k = 0; k < 5; k++) a[k] = lam[k] * idt;
}

} // namespace hbvx
