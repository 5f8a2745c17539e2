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
        sw0 = pow_step_(r, BETA);
        sw = fmin_(fmax_(sw0, 0.0f), 1.0f);
        const float rt = rf + Isnow;
        Peff = rt * sw;
        // :476-479 excess
        const float e0 = SM - FC;
        ex = fmax_(e0, 0.0f);
        // :481-486 evapotranspiration
        const float lpfc = LP * FC;
        qe = div_(SM, lpfc);
        ef0 = BETAET ? pow_step_(qe, p[P_BETAET]) : qe;
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

// Implicit-function adjoint of one day at the solved state x (= y_{t+1}).
//   a[5]  in : dL/dx from the future (lam_{t+1}/dt);  out: dL/dx_t = lam/dt
//   gQ       : dL/dQ_t for this lane (Q = q0+q1+q2 evaluated at x, hbv_adj.py:309-313,431)
//   gp[]     += dL/d(physical parameters of day t)
template <bool BETAET>
HBVX_HD void adj_backstep(AdjStep<BETAET> &s, const float *p, const float *x, float idt, float gQ,
                          float *a, float *gp)
{
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
