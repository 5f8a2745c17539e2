// hbv_step_hourly.h -- Step<MODEL_HOURLY>: HBV 2.0 in rate form with dt = 1/24 day.
//
// Restates the loop body of the reference's `Hbv_2_hourly._PBM`
// (src/hydrodl2/models/hbv/hbv_2_hourly.py:527-675): storage guard-rails at the top of every step
// (:529-533), fluxes as rates per day integrated with `* dt`, Hortonian infiltration excess with a
// wetness-dependent capacity (:575-595), capillary rise (:616-628), lateral flow (:640-645), and
// streamflow = Q0 + Q1 + Q2 + IE (:652).  Forcings are per-step depths: P and PET are divided by
// dt first (:485-487).  Same member interface as the daily Step so that every kernel family
// (one-wave, tiled, time-parallel adjoint) instantiates it unchanged.  The adjoint keeps two saved
// pow results (soil wetness, evaporation factor) like the daily models and recomputes the third
// (infiltration capacity).  Included at the end of hbv_step.h.
#pragma once

namespace hbvx {

template <bool BETAET_UNUSED>
struct Step<MODEL_HOURLY, BETAET_UNUSED> {
    // inputs (storages before the guard rails; forcings as given: depth per step)
    float SP, MW, SM, SUZ, SLZ;
    float P, Tf, PET;
    // guard-railed storages + masks, rates
    float SPc, MWc, SMc, SUZc, SLZc, g0, g1, g2, g3, g4, Pr, PETr;
    float TTe, mlo, m_rain, m_snow, RAIN, SP1, dT, mp, mpcdt, melt, MW1, SP2;
    float cc, dT2, rp, rpcdt, refr, SP3, MW2, ts0, tosoil, MW3;
    float W, r, sc, oms, pw, fminF, fcap, infil, ie0, IE;
    float sw0, sw, rech, SM1, e0, exc, SM2, lpfc, q, ef0, ef, pe, pedt, ETm, ET, dd, SM3;
    float x1, cs, om, capp, capm, cap, smc, SM4, slc, SLZ0;
    float SUZ1, pdt, PERCm, PERC, SUZ2, u0, u0c, Q0, SUZ3, Q1, SUZ4, SLZ1, SLZ1p, Q2, SLZ2, Q;
    float a0, a1, m1, m2, ee, sl;

    static HBVX_HDM float dt_() { return (float)(1.0 / 24.0); } // self.dt (:58)

    // x / dt, correctly rounded like the reference's IEEE division: RN(1/dt) is exactly 24, so one
    // Newton step with the exact residual gives the correctly rounded quotient (Markstein; checked over
    // all 2^23 mantissas).  It has to be exact: the model divides by dt and multiplies by dt again all
    // over (tosoil, excess, ET, capillary rise, percolation), the round trips land within an ulp of the
    // thresholds the next hour compares against (storage >= 0, SM/FC <= 1), and a quotient one ulp off
    // flips those gradient masks -- isolated wrong forcing gradients in dry hours with the rcp-based div_.
    static HBVX_HDM float div_dt_(float a)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        const float q = a * 24.0f;
        const float r = __builtin_fmaf(-dt_(), q, a);
        return __builtin_fmaf(r, 24.0f, q);
#else
        return a / dt_();
#endif
    }

    // The step in two parts, split where the pipelined forward splits it (hbv_pipe.h): the snow
    // routine (needs SP, MW, P, Tf; produces RAIN, tosoil) and everything else (needs SM, SUZ, SLZ,
    // PET, RAIN, tosoil).  fwd() is the two in order -- the same operations in the same order as the
    // single routine it replaces.
    HBVX_HDM void fwd_snow(const float *p, float elev)
    {
        const float dt = dt_();
        const float TT = p[P_TT], CFMAX = p[P_CFMAX], CFR = p[P_CFR], CWH = p[P_CWH];
        Pr = div_dt_(P);      // :485
        // :529-533
        SPc = fmax_(SP, 0.0f);  g0 = (SP >= 0.0f) ? 1.0f : 0.0f;
        MWc = fmax_(MW, 0.0f);  g1 = (MW >= 0.0f) ? 1.0f : 0.0f;
        // :544-548
        const float mhi = (elev >= 2000.0f) ? 1.0f : 0.0f;
        mlo = (elev < 2000.0f) ? 1.0f : 0.0f;
        TTe = mhi * 4.0f + mlo * TT;
        m_rain = (Tf >= TTe) ? 1.0f : 0.0f;
        m_snow = (Tf < TTe) ? 1.0f : 0.0f;
        RAIN = Pr * m_rain;
        const float SNOW = Pr * m_snow;
        // :551-572
        SP1 = SPc + SNOW * dt;
        dT = Tf - TTe;
        mp = CFMAX * dT;
        mpcdt = fmax_(mp, 0.0f) * dt;
        melt = fmin_(mpcdt, SP1);
        MW1 = MWc + melt;
        SP2 = SP1 - melt;
        cc = CFR * CFMAX;
        dT2 = TTe - Tf;
        rp = cc * dT2;
        rpcdt = fmax_(rp, 0.0f) * dt;
        refr = fmin_(rpcdt, MW1);
        SP3 = SP2 + refr;
        MW2 = MW1 - refr;
        ts0 = div_dt_(MW2 - CWH * SP3);
        tosoil = fmax_(ts0, 0.0f);
        MW3 = MW2 - tosoil * dt;
    }

    template <bool USE_AUX>
    HBVX_HDM void fwd_rest(const float *p, float nz, float ac, float aux_sw0, float aux_ef0)
    {
        const float dt = dt_();
        const float BETA = p[P_BETA], FC = p[P_FC], K0 = p[P_K0], K1 = p[P_K1], K2 = p[P_K2],
                    LP = p[P_LP], PERCp = p[P_PERC], UZL = p[P_UZL], BE = p[P_BETAET], C = p[P_C],
                    RT = p[P_RT], AC = p[P_AC], F0 = p[P_F0], FMIN = p[P_FMIN], ALPHA = p[P_ALPHA];
        PETr = div_dt_(PET);  // :487
        SMc = fmax_(SM, nz);    g2 = (SM >= nz) ? 1.0f : 0.0f;   // :529-533
        SUZc = fmax_(SUZ, nz);  g3 = (SUZ >= nz) ? 1.0f : 0.0f;
        SLZc = fmax_(SLZ, nz);  g4 = (SLZ >= nz) ? 1.0f : 0.0f;
        // :577-595
        W = RAIN + tosoil;
        r = div_(SMc, FC);
        sc = fmin_(fmax_(r, 0.0f), (float)(1.0 - 0.01));
        fminF = FMIN * F0;
        oms = 1.0f - sc;
        pw = pow_unit_(oms, ALPHA);
        fcap = fminF + (F0 - fminF) * pw;
        infil = fmin_(W, fcap);
        ie0 = W - fcap;
        IE = fmax_(ie0, 0.0f);
        sw0 = USE_AUX ? aux_sw0 : pow_unit_(r, BETA);
        sw = fmin_(fmax_(sw0, 0.0f), 1.0f);
        rech = infil * sw;
        SM1 = SMc + (infil - rech) * dt;
        // :603-613
        e0 = div_dt_(SM1 - FC);
        exc = fmax_(e0, 0.0f);
        SM2 = SM1 - exc * dt;
        lpfc = LP * FC;
        q = div_(SM2, lpfc);
        ef0 = USE_AUX ? aux_ef0 : pow_unit_(q, BE);
        ef = fmin_(fmax_(ef0, 0.0f), 1.0f);
        pe = PETr * ef;
        pedt = pe * dt;
        ETm = fmin_(SM2, pedt);
        ET = div_dt_(ETm);
        dd = SM2 - ET * dt;
        SM3 = fmax_(dd, nz);
        // :616-628
        x1 = div_(SM3, FC);
        const float rc = fmin_(x1, 1.0f);
        cs = C * SLZc;
        om = 1.0f - rc;
        capp = (cs * om) * dt;
        capm = fmin_(SLZc, capp);
        cap = div_dt_(capm);
        smc = SM3 + cap * dt;
        SM4 = fmax_(smc, nz);
        slc = SLZc - cap * dt;
        SLZ0 = fmax_(slc, nz);
        // :631-648
        SUZ1 = SUZc + (rech + exc) * dt;
        pdt = PERCp * dt;
        PERCm = fmin_(SUZ1, pdt);
        PERC = div_dt_(PERCm);
        SUZ2 = SUZ1 - PERC * dt;
        u0 = SUZ2 - UZL;
        u0c = fmax_(u0, 0.0f);
        Q0 = K0 * u0c;
        SUZ3 = SUZ2 - Q0 * dt;
        Q1 = K1 * SUZ3;
        SUZ4 = SUZ3 - Q1 * dt;
        SLZ1 = SLZ0 + PERC * dt;
        m1 = (ac < 2500.0f) ? 1.0f : 0.0f;
        m2 = (ac >= 2500.0f) ? 1.0f : 0.0f;
        a0 = (ac - AC) / 1000.0f;
        a1 = fmin_(fmax_(a0, -1.0f), 1.0f);
        const float e0_ = -(ac - 2500.0f) / 50.0f;
        const float e1_ = fmin_(fmax_(e0_, -10.0f), 0.0f);
        ee = expf(e1_);
        const float LF = (a1 * RT) * m1 + (ee * RT) * m2;
        sl = SLZ1 + LF * dt;
        SLZ1p = fmax_(sl, 0.0f);
        Q2 = K2 * SLZ1p;
        SLZ2 = SLZ1p - Q2 * dt;
        Q = ((Q0 + Q1) + Q2) + IE; // :652
    }

    template <bool USE_AUX, bool CHAIN = false>     // (CHAIN: the daily models' forward form, hbv_step.h; no effect here)
    HBVX_HDM void fwd(const float *p, float nz, float ac, float elev, float aux_sw0, float aux_ef0)
    {
        fwd_snow(p, elev);
        fwd_rest<USE_AUX>(p, nz, ac, aux_sw0, aux_ef0);
    }

    HBVX_HDM void bwd(const float *p, float nz, const FluxGrad &g, float *a, float *gp, float *gx) const
    {
        HBVX_ADJ_FMA
        const float dt = dt_();
        const float idt = 24.0f; // adjoint-only reciprocal of dt (gradients: rtol 1e-3)
        const float BETA = p[P_BETA], FC = p[P_FC], K0 = p[P_K0], K1 = p[P_K1], K2 = p[P_K2],
                    LP = p[P_LP], CFMAX = p[P_CFMAX], CFR = p[P_CFR], CWH = p[P_CWH], BE = p[P_BETAET],
                    C = p[P_C], RT = p[P_RT], F0 = p[P_F0], FMIN = p[P_FMIN], ALPHA = p[P_ALPHA];
        float wa, wb;
        float aSP3 = a[0] + g.gSWE, aMW3 = a[1], aSM4 = a[2], aSUZ4 = a[3], aSLZ2 = a[4];
        float aQ0 = g.gQ0 + g.gQ, aQ1 = g.gQ1 + g.gQ, aQ2 = g.gQ2 + g.gQ;
        const float aIE = g.gQ;
        // lower box
        aQ2 -= aSLZ2 * dt;
        const float aSLZ1p = aSLZ2 + aQ2 * K2;
        gp[P_K2] += aQ2 * SLZ1p;
        const float as = (sl >= 0.0f) ? aSLZ1p : 0.0f;
        const float aLF = as * dt;
        const float at1 = aLF * m1, at2 = aLF * m2;
        gp[P_RT] += at1 * a1 + at2 * ee;
        const float aa1 = at1 * RT;
        const float aa0 = (a0 >= -1.0f && a0 <= 1.0f) ? aa1 : 0.0f;
        gp[P_AC] += -(aa0 * 0.001f);
        const float aSLZ0 = as;
        float aPERC = g.gPERC + as * dt;
        // upper box
        aQ1 -= aSUZ4 * dt;
        const float aSUZ3 = aSUZ4 + aQ1 * K1;
        gp[P_K1] += aQ1 * SUZ3;
        aQ0 -= aSUZ3 * dt;
        float aSUZ2 = aSUZ3;
        gp[P_K0] += aQ0 * u0c;
        const float au0 = (u0 >= 0.0f) ? aQ0 * K0 : 0.0f;
        aSUZ2 += au0;
        gp[P_UZL] -= au0;
        aPERC -= aSUZ2 * dt;
        float aSUZ1 = aSUZ2;
        const float aPERCm = aPERC * idt;
        minw_(SUZ1, pdt, wa, wb);
        aSUZ1 += aPERCm * wa;
        gp[P_PERC] += (aPERCm * wb) * dt;
        const float aSUZ = aSUZ1;
        const float are = aSUZ1 * dt;
        float arech = g.grech + are;
        float aexc = g.gexc + are;
        // capillary rise
        const float az = (slc >= nz) ? aSLZ0 : 0.0f;
        float aSLZ = az;
        float acap = g.gcap - az * dt;
        const float ay = (smc >= nz) ? aSM4 : 0.0f;
        float aSM3 = ay;
        acap += ay * dt;
        const float acapm = acap * idt;
        minw_(SLZc, capp, wa, wb);
        aSLZ += acapm * wa;
        const float acsom = (acapm * wb) * dt;
        const float acs = acsom * om;
        const float arc = -(acsom * cs);
        gp[P_C] += acs * SLZc;
        aSLZ += acs * C;
        const float ax1 = (x1 <= 1.0f) ? arc : 0.0f;
        aSM3 += div_approx_(ax1, FC);
        gp[P_FC] += -ax1 * div_approx_(x1, FC);
        // evapotranspiration
        const float add = (dd >= nz) ? aSM3 : 0.0f;
        float aSM2 = add;
        const float aET = g.gET - add * dt;
        const float aETm = aET * idt;
        minw_(SM2, pedt, wa, wb);
        aSM2 += aETm * wa;
        const float ape = (aETm * wb) * dt;
        const float aef = g.gef + ape * PETr;
        gx[2] = (ape * ef) * idt;
        const float aef0 = (ef0 >= 0.0f && ef0 <= 1.0f) ? aef : 0.0f;
        const float dq = (q > 0.0f) ? BE * div_approx_(ef0, q) : 0.0f;
        const float dbe = (q > 0.0f) ? ef0 * log_fast_(q) : 0.0f;
        const float aq = aef0 * dq;
        gp[P_BETAET] += aef0 * dbe;
        aSM2 += div_approx_(aq, lpfc);
        const float alpfc = -aq * div_approx_(q, lpfc);
        gp[P_LP] += alpfc * FC;
        gp[P_FC] += alpfc * LP;
        // excess
        float aSM1 = aSM2;
        aexc -= aSM2 * dt;
        const float ae0 = (e0 >= 0.0f) ? aexc : 0.0f;
        const float ae0d = ae0 * idt;
        aSM1 += ae0d;
        gp[P_FC] -= ae0d;
        // soil
        float aSM = aSM1;
        const float aird = aSM1 * dt;
        float ainfil = aird;
        arech -= aird;
        ainfil += arech * sw;
        const float asw = arech * infil;
        const float asw0 = (sw0 >= 0.0f && sw0 <= 1.0f) ? asw : 0.0f;
        const float dr = (r > 0.0f) ? BETA * div_approx_(sw0, r) : 0.0f;
        const float dbr = (r > 0.0f) ? sw0 * log_fast_(r) : 0.0f;
        float ar = asw0 * dr;
        gp[P_BETA] += asw0 * dbr;
        // infiltration capacity
        minw_(W, fcap, wa, wb);
        float aW = ainfil * wa;
        float afcap = ainfil * wb;
        const float aie0 = (ie0 >= 0.0f) ? aIE : 0.0f;
        aW += aie0;
        afcap -= aie0;
        float afmin = afcap;
        const float adiff = afcap * pw;
        const float apw = afcap * (F0 - fminF);
        gp[P_F0] += adiff;
        afmin -= adiff;
        gp[P_FMIN] += afmin * F0;
        gp[P_F0] += afmin * FMIN;
        const float aoms = apw * (ALPHA * div_approx_(pw, oms)); // oms >= 0.01
        gp[P_ALPHA] += apw * (pw * log_fast_(oms));
        ar += (r >= 0.0f && r <= (float)(1.0 - 0.01)) ? -aoms : 0.0f;
        aSM += div_approx_(ar, FC);
        gp[P_FC] += -ar * div_approx_(r, FC);
        const float aRAIN = aW;
        float atosoil = g.gtosoil + aW;
        // snow
        float aMW2 = aMW3;
        atosoil -= aMW3 * dt;
        const float ats0 = (ts0 >= 0.0f) ? atosoil : 0.0f;
        const float ats0d = ats0 * idt;
        aMW2 += ats0d;
        gp[P_CWH] -= ats0d * SP3;
        aSP3 -= ats0d * CWH;
        const float aSP2 = aSP3;
        const float arefr = aSP3 - aMW2;
        float aMW1 = aMW2;
        minw_(rpcdt, MW1, wa, wb);
        const float arpc = (arefr * wa) * dt;
        aMW1 += arefr * wb;
        const float arp = (rp >= 0.0f) ? arpc : 0.0f;
        const float acc = arp * dT2;
        gp[P_CFR] += acc * CFMAX;
        gp[P_CFMAX] += acc * CFR;
        float aTTe = arp * cc;
        float aTf = -(arp * cc);
        const float aMW = aMW1;
        const float amelt = aMW1 - aSP2;
        float aSP1 = aSP2;
        minw_(mpcdt, SP1, wa, wb);
        const float ampc = (amelt * wa) * dt;
        aSP1 += amelt * wb;
        const float amp = (mp >= 0.0f) ? ampc : 0.0f;
        gp[P_CFMAX] += amp * dT;
        aTTe -= amp * CFMAX;
        aTf += amp * CFMAX;
        const float aSNOW = aSP1 * dt;
        gx[0] = (aSNOW * m_snow + aRAIN * m_rain) * idt;
        gx[1] = aTf;
        gp[P_TT] += aTTe * mlo;
        a[0] = aSP1 * g0; a[1] = aMW * g1; a[2] = aSM * g2; a[3] = aSUZ * g3; a[4] = aSLZ * g4;
    }
};

} // namespace hbvx
