/*
 * hbv_oracle.c -- CPU ORACLE for the HBV time-stepper.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, scalar, fp32 restatement of the reference algorithm
 * (mhpi/hydrodl2, src/hydrodl2/models/hbv/{hbv,hbv_1_1p,hbv_2}.py and
 * src/hydrodl2/core/calc/{utils,uh_routing}.py), implementing the ABI of
 * include/hbvx.h on host memory.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product (hydrodl2_amd/) never does.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
 * outputs and autograd gradients of the reference itself (tests/golden/NAME.npz,
 * produced by tests/golden/make_golden.py in the authoring container).
 *
 * Build: oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp).  No FMA
 * contraction: the reference evaluates one rounded ATen op per Python operator.
 *
 * Every function cites the reference lines it restates.  The backward pass is a
 * hand-derived adjoint that reproduces PyTorch autograd's conventions
 * (SURVEY.md §8 a11): minimum() ties send grad/2 to each argument,
 * clamp(min=m) passes the gradient where x >= m (inclusive), clamp(0,1) where
 * 0 <= x <= 1, comparison masks carry no gradient.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/hbvx.h"

#ifdef _OPENMP
#include <omp.h>
#endif

static __thread char g_err[256] = "";

static int fail(int code, const char *msg)
{
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

int hbvx_version(void) { return HBVX_ABI_VERSION; }
/* the oracle re-materialises per lane and needs no scratch */
uint64_t hbvx_ckpt_workspace_bytes(const hbvx_desc *d, int32_t K) { (void)d; (void)K; return 0; }
/* the oracle keeps the plain row layout of the trajectory */
int hbvx_preferred_traj_layout(const hbvx_desc *d) { (void)d; return HBVX_TRAJ_ROWS; }
const char *hbvx_last_error(void) { return g_err; }
const char *hbvx_last_dispatch(int direction) { (void)direction; return "oracle"; }
int hbvx_zero_in_launch(void) { return 0; }
/* include/hbvx.h, ABI 10: the CPU forward writes no zeros (zero_state[0] stays 0), so the rest is everything */
int hbvx_zero_rest(void *ptr, uint64_t bytes, const void *zero_state, void *stream)
{
    (void)zero_state; (void)stream;
    if (bytes && !ptr) return fail(HBVX_E_NULL, "hbvx_zero_rest: buffer is NULL");
    if (bytes) memset(ptr, 0, (size_t)bytes);
    return HBVX_OK;
}
const char *hbvx_backend(void) { return "cpu-oracle"; }

uint64_t hbvx_sizeof(int which)
{
    switch (which) {
    case 0: return sizeof(hbvx_desc);
    case 1: return sizeof(hbvx_fwd_out);
    case 2: return sizeof(hbvx_bwd_io);
    case 3: return sizeof(hbvx_route_desc);
    case 4: return sizeof(hbvx_param_src);
    case 5: return sizeof(hbvx_param_grad);
    case 6: return sizeof(hbvx_gage_desc);
    default: return 0;
    }
}

/* Conditioning probe (tests/test_conditioning.py): the step's powers perturbed by +-`ulps` units in the last place,
 * the sign from a hash of the operands (the same operands give the same perturbation, so the adjoint's recomputed
 * forward sees what the forward saw).  0 = libm's powf, the oracle proper.  What a comparison of the perturbed run
 * with the plain one measures is the sensitivity of the reference's OWN float32 trajectory to the last bit of its
 * pow -- the noise floor any other pow implementation (torch's vectorised CPU pow, the GPU's) sits on. */
static int g_pow_noise = 0;
void hbvo_set_pow_noise(int ulps) { g_pow_noise = ulps; }
static inline float step_powf(float x, float y)
{
    float r = powf(x, y);
    if (g_pow_noise && r > 0.0f && r < 3.0e38f) {
        union { float f; uint32_t u; } a, b, c;
        a.f = x; b.f = y; c.f = r;
        uint32_t h = (a.u * 2654435761u) ^ (b.u * 2246822519u);
        h ^= h >> 15;
        if (h & 1u) c.u += (uint32_t)g_pow_noise; else c.u -= (uint32_t)g_pow_noise;
        r = c.f;
    }
    return r;
}

int hbvo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ */
/* parameter prep: hbv.py:201-208 (sigmoid), :242-246 (static row, dy_drop
 * blend), core/calc/utils.py:24 (p*(hi-lo)+lo, multiply THEN add).     */

static inline float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

typedef struct {
    float u;      /* unit value actually used (after sigmoid + blend) */
    int from_dyn; /* 1: came from the per-step tensor, 0: static row */
} unit_t;

static inline unit_t fetch_unit(const hbvx_desc *d, int i, int t, int b, int j)
{
    const hbvx_param_src *s = &d->p[i];
    unit_t r;
    float v;
    if (s->dyn && !(s->drop && s->drop[b])) {
        v = s->dyn[(int64_t)t * s->dyn_t_stride + (int64_t)b * s->dyn_b_stride + j];
        r.from_dyn = 1;
    } else {
        v = s->sta[(int64_t)b * s->sta_b_stride + j];
        r.from_dyn = 0;
    }
    r.u = d->raw_sigmoid ? sigmoidf_(v) : v;
    return r;
}

static inline float descale(float u, float lo, float hi) { return u * (hi - lo) + lo; }

/* ------------------------------------------------------------------ */
/* one explicit daily step, all intermediates kept for the adjoint.     */

typedef struct {
    /* inputs */
    float SP, MW, SM, SUZ, SLZ;
    float P, Tf, PET;
    /* snow */
    float TTe, mlo; /* effective threshold; d(TTe)/d(TT) */
    float m_rain, m_snow, RAIN, SNOW, SP1, dT, mp, mpc, melt, MW1, SP2;
    float cc, dT2, rp, rpc, refr, SP3, MW2, cwsp, ts0, tosoil, MW3;
    /* soil */
    float r, sw0, sw, rt, rech, SM1, e0, exc, SM2, lpfc, q, ef0, ef, pe, ET, dd, SM3;
    /* capillary (1.1p, 2.0) */
    float x1, rc, cs, om, capp, cap, smc, SM4, slc, SLZ0;
    /* groundwater */
    float SUZ1, PERC, SUZ2, u0, u0c, Q0, SUZ3, Q1, SUZ4, SLZ1, SLZ1p, Q2, SLZ2, Q;
    /* lateral flow (2.0) */
    float a0, a1, m1, m2, ee, LF, sl;
} step_t;

static inline float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

/* hbv.py:428-494 ; hbv_1_1p.py:427-506 ; hbv_2.py:471-556 */
static void step_fwd(int model, int has_betaet, float nz, const float *p, float ac, float elev,
                     step_t *s)
{
    const float BETA = p[HBVX_P_BETA], FC = p[HBVX_P_FC], K0 = p[HBVX_P_K0], K1 = p[HBVX_P_K1],
                K2 = p[HBVX_P_K2], LP = p[HBVX_P_LP], PERCp = p[HBVX_P_PERC],
                UZL = p[HBVX_P_UZL], TT = p[HBVX_P_TT], CFMAX = p[HBVX_P_CFMAX],
                CFR = p[HBVX_P_CFR], CWH = p[HBVX_P_CWH];
    /* hbv_2.py:473-475: parTT_new = (Elev>=2000)*4.0 + (Elev<2000)*parTT */
    if (model == HBVX_MODEL_HBV20) {
        float mhi = (elev >= 2000.0f) ? 1.0f : 0.0f;
        s->mlo = (elev < 2000.0f) ? 1.0f : 0.0f;
        s->TTe = mhi * 4.0f + s->mlo * TT;
    } else {
        s->mlo = 1.0f;
        s->TTe = TT;
    }
    /* hbv.py:429-435 */
    s->m_rain = (s->Tf >= s->TTe) ? 1.0f : 0.0f;
    s->m_snow = (s->Tf < s->TTe) ? 1.0f : 0.0f;
    s->RAIN = s->P * s->m_rain;
    s->SNOW = s->P * s->m_snow;
    /* hbv.py:438-445 */
    s->SP1 = s->SP + s->SNOW;
    s->dT = s->Tf - s->TTe;
    s->mp = CFMAX * s->dT;
    s->mpc = fmaxf(s->mp, 0.0f);
    s->melt = fminf(s->mpc, s->SP1);
    s->MW1 = s->MW + s->melt;
    s->SP2 = s->SP1 - s->melt;
    /* hbv.py:446-456 */
    s->cc = CFR * CFMAX;
    s->dT2 = s->TTe - s->Tf;
    s->rp = s->cc * s->dT2;
    s->rpc = fmaxf(s->rp, 0.0f);
    s->refr = fminf(s->rpc, s->MW1);
    s->SP3 = s->SP2 + s->refr;
    s->MW2 = s->MW1 - s->refr;
    /* hbv.py:457-459 */
    s->cwsp = CWH * s->SP3;
    s->ts0 = s->MW2 - s->cwsp;
    s->tosoil = fmaxf(s->ts0, 0.0f);
    s->MW3 = s->MW2 - s->tosoil;
    /* hbv.py:462-472 */
    s->r = s->SM / FC;
    s->sw0 = step_powf(s->r, BETA);
    s->sw = clamp01(s->sw0);
    s->rt = s->RAIN + s->tosoil;
    s->rech = s->rt * s->sw;
    s->SM1 = ((s->SM + s->RAIN) + s->tosoil) - s->rech;
    s->e0 = s->SM1 - FC;
    s->exc = fmaxf(s->e0, 0.0f);
    s->SM2 = s->SM1 - s->exc;
    /* hbv.py:474-480 ; hbv_1_1p.py:473-480 */
    s->lpfc = LP * FC;
    s->q = s->SM2 / s->lpfc;
    s->ef0 = has_betaet ? step_powf(s->q, p[HBVX_P_BETAET]) : s->q;
    s->ef = clamp01(s->ef0);
    s->pe = s->PET * s->ef;
    s->ET = fminf(s->SM2, s->pe);
    s->dd = s->SM2 - s->ET;
    s->SM3 = fmaxf(s->dd, nz);
    /* hbv_1_1p.py:482-490 ; hbv_2.py:525-533 */
    if (model != HBVX_MODEL_HBV10) {
        const float C = p[HBVX_P_C];
        s->x1 = s->SM3 / FC;
        s->rc = fminf(s->x1, 1.0f);
        s->cs = C * s->SLZ;
        s->om = 1.0f - s->rc;
        s->capp = s->cs * s->om;
        s->cap = fminf(s->SLZ, s->capp);
        s->smc = s->SM3 + s->cap;
        s->SM4 = fmaxf(s->smc, nz);
        s->slc = s->SLZ - s->cap;
        s->SLZ0 = fmaxf(s->slc, nz);
    } else {
        s->cap = 0.0f;
        s->SM4 = s->SM3;
        s->SLZ0 = s->SLZ;
    }
    /* hbv.py:483-492 */
    s->SUZ1 = (s->SUZ + s->rech) + s->exc;
    s->PERC = fminf(s->SUZ1, PERCp);
    s->SUZ2 = s->SUZ1 - s->PERC;
    s->u0 = s->SUZ2 - UZL;
    s->u0c = fmaxf(s->u0, 0.0f);
    s->Q0 = K0 * s->u0c;
    s->SUZ3 = s->SUZ2 - s->Q0;
    s->Q1 = K1 * s->SUZ3;
    s->SUZ4 = s->SUZ3 - s->Q1;
    s->SLZ1 = s->SLZ0 + s->PERC;
    /* hbv_2.py:545-550 */
    if (model == HBVX_MODEL_HBV20) {
        const float RT = p[HBVX_P_RT], AC = p[HBVX_P_AC];
        s->m1 = (ac < 2500.0f) ? 1.0f : 0.0f;
        s->m2 = (ac >= 2500.0f) ? 1.0f : 0.0f;
        s->a0 = (ac - AC) / 1000.0f;
        s->a1 = fminf(fmaxf(s->a0, -1.0f), 1.0f);
        float e0 = -(ac - 2500.0f) / 50.0f;
        float e1 = fminf(fmaxf(e0, -10.0f), 0.0f);
        s->ee = expf(e1);
        s->LF = (s->a1 * RT) * s->m1 + (s->ee * RT) * s->m2;
        s->sl = s->SLZ1 + s->LF;
        s->SLZ1p = fmaxf(s->sl, 0.0f);
    } else {
        s->SLZ1p = s->SLZ1;
    }
    s->Q2 = K2 * s->SLZ1p;
    s->SLZ2 = s->SLZ1p - s->Q2;
    /* hbv.py:494 */
    s->Q = (s->Q0 + s->Q1) + s->Q2;
}

/* torch.minimum backward: ties split. */
static inline void minw(float a, float b, float *wa, float *wb)
{
    if (a < b) { *wa = 1.0f; *wb = 0.0f; }
    else if (a > b) { *wa = 0.0f; *wb = 1.0f; }
    else { *wa = 0.5f; *wb = 0.5f; }
}

/* x**y backward (torch pow_backward_self / pow_backward_exponent). */
static inline float pow_dx(float x, float y) { return (y == 0.0f) ? 0.0f : y * powf(x, y - 1.0f); }
static inline float pow_dy(float x, float y, float res)
{
    return (x == 0.0f && y >= 0.0f) ? 0.0f : res * logf(x);
}

typedef struct {
    float gQ, gQ0, gQ1, gQ2, gET, gSWE, grech, gexc, gef, gtosoil, gPERC, gcap;
} fluxgrad_t;

/* Adjoint of step_fwd.  a[5] holds dL/d(new states) on entry and dL/d(old states)
 * on exit; gp[] accumulates dL/d(physical parameters); gx[3] = dL/d(P,T,PET). */
static void step_bwd(int model, int has_betaet, float nz, const float *p, const step_t *s,
                     const fluxgrad_t *g, float *a, float *gp, float *gx)
{
    const float BETA = p[HBVX_P_BETA], FC = p[HBVX_P_FC], K0 = p[HBVX_P_K0], K1 = p[HBVX_P_K1],
                K2 = p[HBVX_P_K2], LP = p[HBVX_P_LP], PERCp = p[HBVX_P_PERC],
                CFMAX = p[HBVX_P_CFMAX], CFR = p[HBVX_P_CFR], CWH = p[HBVX_P_CWH];
    float wa, wb;
    float aSP3 = a[0] + g->gSWE, aMW3 = a[1], aSMn = a[2], aSUZ4 = a[3], aSLZ2 = a[4];
    float aQ0 = g->gQ0 + g->gQ, aQ1 = g->gQ1 + g->gQ, aQ2 = g->gQ2 + g->gQ;

    /* SLZ2 = SLZ1p - Q2 ; Q2 = K2*SLZ1p */
    aQ2 -= aSLZ2;
    float aSLZ1p = aSLZ2 + aQ2 * K2;
    gp[HBVX_P_K2] += aQ2 * s->SLZ1p;
    float aSLZ1 = aSLZ1p;
    if (model == HBVX_MODEL_HBV20) {
        const float RT = p[HBVX_P_RT];
        float as = (s->sl >= 0.0f) ? aSLZ1p : 0.0f;
        aSLZ1 = as;
        float at1 = as * s->m1; /* adjoint of (a1*RT) */
        float at2 = as * s->m2; /* adjoint of (ee*RT) */
        gp[HBVX_P_RT] += at1 * s->a1 + at2 * s->ee;
        float aa1 = at1 * RT;
        float aa0 = (s->a0 >= -1.0f && s->a0 <= 1.0f) ? aa1 : 0.0f;
        gp[HBVX_P_AC] += -(aa0 / 1000.0f);
    }
    /* SLZ1 = SLZ0 + PERC */
    float aSLZ0 = aSLZ1;
    float aPERC = g->gPERC + aSLZ1;
    /* SUZ4 = SUZ3 - Q1 ; Q1 = K1*SUZ3 */
    aQ1 -= aSUZ4;
    float aSUZ3 = aSUZ4 + aQ1 * K1;
    gp[HBVX_P_K1] += aQ1 * s->SUZ3;
    /* SUZ3 = SUZ2 - Q0 ; Q0 = K0*max(u0,0) ; u0 = SUZ2 - UZL */
    aQ0 -= aSUZ3;
    float aSUZ2 = aSUZ3;
    gp[HBVX_P_K0] += aQ0 * s->u0c;
    float au0 = (s->u0 >= 0.0f) ? aQ0 * K0 : 0.0f;
    aSUZ2 += au0;
    gp[HBVX_P_UZL] -= au0;
    /* SUZ2 = SUZ1 - PERC ; PERC = min(SUZ1, PERCp) */
    aPERC -= aSUZ2;
    float aSUZ1 = aSUZ2;
    minw(s->SUZ1, PERCp, &wa, &wb);
    aSUZ1 += aPERC * wa;
    gp[HBVX_P_PERC] += aPERC * wb;
    /* SUZ1 = (SUZ + rech) + exc */
    float aSUZ = aSUZ1;
    float arech = g->grech + aSUZ1;
    float aexc = g->gexc + aSUZ1;

    float aSLZ, aSM3;
    if (model != HBVX_MODEL_HBV10) {
        const float C = p[HBVX_P_C];
        /* SLZ0 = max(SLZ - cap, nz) ; SM4 = max(SM3 + cap, nz) */
        float az = (s->slc >= nz) ? aSLZ0 : 0.0f;
        aSLZ = az;
        float acap = g->gcap - az;
        float ay = (s->smc >= nz) ? aSMn : 0.0f;
        aSM3 = ay;
        acap += ay;
        /* cap = min(SLZ, capp) */
        minw(s->SLZ, s->capp, &wa, &wb);
        aSLZ += acap * wa;
        float acapp = acap * wb;
        /* capp = (C*SLZ) * (1 - rc) */
        float acs = acapp * s->om;
        float arc = -(acapp * s->cs);
        gp[HBVX_P_C] += acs * s->SLZ;
        aSLZ += acs * C;
        /* rc = clamp(x1, max=1) ; x1 = SM3/FC */
        float ax1 = (s->x1 <= 1.0f) ? arc : 0.0f;
        aSM3 += ax1 / FC;
        gp[HBVX_P_FC] += -ax1 * ((s->SM3 / FC) / FC);
    } else {
        aSLZ = aSLZ0;
        aSM3 = aSMn;
    }
    /* SM3 = max(dd, nz) ; dd = SM2 - ET */
    float add = (s->dd >= nz) ? aSM3 : 0.0f;
    float aSM2 = add;
    float aET = g->gET - add;
    /* ET = min(SM2, pe) ; pe = PET*ef */
    minw(s->SM2, s->pe, &wa, &wb);
    aSM2 += aET * wa;
    float ape = aET * wb;
    float aef = g->gef + ape * s->PET;
    gx[2] = ape * s->ef;
    /* ef = clamp(ef0, 0, 1) */
    float aef0 = (s->ef0 >= 0.0f && s->ef0 <= 1.0f) ? aef : 0.0f;
    float aq;
    if (has_betaet) {
        const float BE = p[HBVX_P_BETAET];
        aq = aef0 * pow_dx(s->q, BE);
        gp[HBVX_P_BETAET] += aef0 * pow_dy(s->q, BE, s->ef0);
    } else {
        aq = aef0;
    }
    /* q = SM2 / (LP*FC) */
    aSM2 += aq / s->lpfc;
    float alpfc = -aq * ((s->SM2 / s->lpfc) / s->lpfc);
    gp[HBVX_P_LP] += alpfc * FC;
    gp[HBVX_P_FC] += alpfc * LP;
    /* SM2 = SM1 - exc ; exc = max(e0,0) ; e0 = SM1 - FC */
    float aSM1 = aSM2;
    aexc -= aSM2;
    float ae0 = (s->e0 >= 0.0f) ? aexc : 0.0f;
    aSM1 += ae0;
    gp[HBVX_P_FC] -= ae0;
    /* SM1 = ((SM + RAIN) + tosoil) - rech */
    float aSM = aSM1;
    float aRAIN = aSM1;
    float atosoil = g->gtosoil + aSM1;
    arech -= aSM1;
    /* rech = rt*sw ; rt = RAIN + tosoil */
    float art = arech * s->sw;
    float asw = arech * s->rt;
    aRAIN += art;
    atosoil += art;
    /* sw = clamp(sw0,0,1) ; sw0 = r**BETA ; r = SM/FC */
    float asw0 = (s->sw0 >= 0.0f && s->sw0 <= 1.0f) ? asw : 0.0f;
    float ar = asw0 * pow_dx(s->r, BETA);
    gp[HBVX_P_BETA] += asw0 * pow_dy(s->r, BETA, s->sw0);
    aSM += ar / FC;
    gp[HBVX_P_FC] += -ar * ((s->SM / FC) / FC);
    /* MW3 = MW2 - tosoil ; tosoil = max(ts0,0) ; ts0 = MW2 - CWH*SP3 */
    float aMW2 = aMW3;
    atosoil -= aMW3;
    float ats0 = (s->ts0 >= 0.0f) ? atosoil : 0.0f;
    aMW2 += ats0;
    gp[HBVX_P_CWH] -= ats0 * s->SP3;
    aSP3 -= ats0 * CWH;
    /* SP3 = SP2 + refr ; MW2 = MW1 - refr */
    float aSP2 = aSP3;
    float arefr = aSP3 - aMW2;
    float aMW1 = aMW2;
    /* refr = min(rpc, MW1) ; rpc = max(rp,0) ; rp = (CFR*CFMAX)*(TTe - T) */
    minw(s->rpc, s->MW1, &wa, &wb);
    float arpc = arefr * wa;
    aMW1 += arefr * wb;
    float arp = (s->rp >= 0.0f) ? arpc : 0.0f;
    float acc = arp * s->dT2;
    gp[HBVX_P_CFR] += acc * CFMAX;
    gp[HBVX_P_CFMAX] += acc * CFR;
    float aTTe = arp * s->cc;
    float aTf = -(arp * s->cc);
    /* MW1 = MW + melt ; SP2 = SP1 - melt */
    float aMW = aMW1;
    float amelt = aMW1 - aSP2;
    float aSP1 = aSP2;
    /* melt = min(mpc, SP1) ; mpc = max(mp,0) ; mp = CFMAX*(T - TTe) */
    minw(s->mpc, s->SP1, &wa, &wb);
    float ampc = amelt * wa;
    aSP1 += amelt * wb;
    float amp = (s->mp >= 0.0f) ? ampc : 0.0f;
    gp[HBVX_P_CFMAX] += amp * s->dT;
    aTTe -= amp * CFMAX;
    aTf += amp * CFMAX;
    /* SP1 = SP + SNOW ; SNOW = P*[T<TTe] ; RAIN = P*[T>=TTe] */
    float aSP = aSP1;
    gx[0] = aSP1 * s->m_snow + aRAIN * s->m_rain;
    gx[1] = aTf;
    gp[HBVX_P_TT] += aTTe * s->mlo;

    a[0] = aSP; a[1] = aMW; a[2] = aSM; a[3] = aSUZ; a[4] = aSLZ;
}

/* ------------------------------------------------------------------ */
/* Hbv_2_hourly: HBV 2.0 in rate form with dt = 1/24 day (hbv_2_hourly.py:527-675)            */

typedef struct {
    float SPi, MWi, SMi, SUZi, SLZi;          /* incoming storages (before the guard rails) */
    float SP, MW, SM, SUZ, SLZ, g0, g1, g2, g3, g4;
    float P, Tf, PET;                          /* rates: forcing / dt (:485-487) */
    float TTe, mlo, m_rain, m_snow, RAIN, SP1, dT, mp, mpcdt, melt, MW1, SP2;
    float cc, dT2, rp, rpcdt, refr, SP3, MW2, ts0, tosoil, MW3;
    float W, r, s, oms, pw, fmin_, fcap, infil, ie0, IE;
    float sw0, sw, rech, SM1, e0, exc, SM2, lpfc, q, ef0, ef, pe, pedt, ETm, ET, dd, SM3;
    float x1, rc, cs, om, capp, capm, cap, smc, SM4, slc, SLZ0;
    float SUZ1, pdt, PERCm, PERC, SUZ2, u0, u0c, Q0, SUZ3, Q1, SUZ4, SLZ1, SLZ1p, Q2, SLZ2, Q;
    float a0, a1, m1, m2, ee, sl;
} hstep_t;

#define HDT ((float)(1.0 / 24.0)) /* self.dt (hbv_2_hourly.py:58) */

static void hstep_fwd(float nz, const float *p, float ac, float elev, hstep_t *s)
{
    const float dt = HDT;
    const float BETA = p[HBVX_P_BETA], FC = p[HBVX_P_FC], K0 = p[HBVX_P_K0], K1 = p[HBVX_P_K1],
                K2 = p[HBVX_P_K2], LP = p[HBVX_P_LP], PERCp = p[HBVX_P_PERC], UZL = p[HBVX_P_UZL],
                TT = p[HBVX_P_TT], CFMAX = p[HBVX_P_CFMAX], CFR = p[HBVX_P_CFR], CWH = p[HBVX_P_CWH],
                BE = p[HBVX_P_BETAET], C = p[HBVX_P_C], RT = p[HBVX_P_RT], AC = p[HBVX_P_AC],
                F0 = p[HBVX_P_F0], FMIN = p[HBVX_P_FMIN], ALPHA = p[HBVX_P_ALPHA];
    /* :529-533 guard rails */
    s->SP = fmaxf(s->SPi, 0.0f);  s->g0 = (s->SPi >= 0.0f) ? 1.0f : 0.0f;
    s->MW = fmaxf(s->MWi, 0.0f);  s->g1 = (s->MWi >= 0.0f) ? 1.0f : 0.0f;
    s->SM = fmaxf(s->SMi, nz);    s->g2 = (s->SMi >= nz) ? 1.0f : 0.0f;
    s->SUZ = fmaxf(s->SUZi, nz);  s->g3 = (s->SUZi >= nz) ? 1.0f : 0.0f;
    s->SLZ = fmaxf(s->SLZi, nz);  s->g4 = (s->SLZi >= nz) ? 1.0f : 0.0f;
    /* :544-548 */
    float mhi = (elev >= 2000.0f) ? 1.0f : 0.0f;
    s->mlo = (elev < 2000.0f) ? 1.0f : 0.0f;
    s->TTe = mhi * 4.0f + s->mlo * TT;
    s->m_rain = (s->Tf >= s->TTe) ? 1.0f : 0.0f;
    s->m_snow = (s->Tf < s->TTe) ? 1.0f : 0.0f;
    s->RAIN = s->P * s->m_rain;
    float SNOW = s->P * s->m_snow;
    /* :551-572 */
    s->SP1 = s->SP + SNOW * dt;
    s->dT = s->Tf - s->TTe;
    s->mp = CFMAX * s->dT;
    s->mpcdt = fmaxf(s->mp, 0.0f) * dt;
    s->melt = fminf(s->mpcdt, s->SP1);
    s->MW1 = s->MW + s->melt;
    s->SP2 = s->SP1 - s->melt;
    s->cc = CFR * CFMAX;
    s->dT2 = s->TTe - s->Tf;
    s->rp = s->cc * s->dT2;
    s->rpcdt = fmaxf(s->rp, 0.0f) * dt;
    s->refr = fminf(s->rpcdt, s->MW1);
    s->SP3 = s->SP2 + s->refr;
    s->MW2 = s->MW1 - s->refr;
    s->ts0 = (s->MW2 - CWH * s->SP3) / dt;
    s->tosoil = fmaxf(s->ts0, 0.0f);
    s->MW3 = s->MW2 - s->tosoil * dt;
    /* :577-595 Hortonian infiltration excess */
    s->W = s->RAIN + s->tosoil;
    s->r = s->SM / FC;
    s->s = fminf(fmaxf(s->r, 0.0f), (float)(1.0 - 0.01));
    s->fmin_ = FMIN * F0;
    s->oms = 1.0f - s->s;
    s->pw = step_powf(s->oms, ALPHA);
    s->fcap = s->fmin_ + (F0 - s->fmin_) * s->pw;
    s->infil = fminf(s->W, s->fcap);
    s->ie0 = s->W - s->fcap;
    s->IE = fmaxf(s->ie0, 0.0f);
    s->sw0 = step_powf(s->r, BETA);
    s->sw = clamp01(s->sw0);
    s->rech = s->infil * s->sw;
    s->SM1 = s->SM + (s->infil - s->rech) * dt;
    /* :603-613 */
    s->e0 = (s->SM1 - FC) / dt;
    s->exc = fmaxf(s->e0, 0.0f);
    s->SM2 = s->SM1 - s->exc * dt;
    s->lpfc = LP * FC;
    s->q = s->SM2 / s->lpfc;
    s->ef0 = step_powf(s->q, BE);
    s->ef = clamp01(s->ef0);
    s->pe = s->PET * s->ef;
    s->pedt = s->pe * dt;
    s->ETm = fminf(s->SM2, s->pedt);
    s->ET = s->ETm / dt;
    s->dd = s->SM2 - s->ET * dt;
    s->SM3 = fmaxf(s->dd, nz);
    /* :616-628 capillary rise */
    s->x1 = s->SM3 / FC;
    s->rc = fminf(s->x1, 1.0f);
    s->cs = C * s->SLZ;
    s->om = 1.0f - s->rc;
    s->capp = (s->cs * s->om) * dt;
    s->capm = fminf(s->SLZ, s->capp);
    s->cap = s->capm / dt;
    s->smc = s->SM3 + s->cap * dt;
    s->SM4 = fmaxf(s->smc, nz);
    s->slc = s->SLZ - s->cap * dt;
    s->SLZ0 = fmaxf(s->slc, nz);
    /* :631-648 groundwater */
    s->SUZ1 = s->SUZ + (s->rech + s->exc) * dt;
    s->pdt = PERCp * dt;
    s->PERCm = fminf(s->SUZ1, s->pdt);
    s->PERC = s->PERCm / dt;
    s->SUZ2 = s->SUZ1 - s->PERC * dt;
    s->u0 = s->SUZ2 - UZL;
    s->u0c = fmaxf(s->u0, 0.0f);
    s->Q0 = K0 * s->u0c;
    s->SUZ3 = s->SUZ2 - s->Q0 * dt;
    s->Q1 = K1 * s->SUZ3;
    s->SUZ4 = s->SUZ3 - s->Q1 * dt;
    s->SLZ1 = s->SLZ0 + s->PERC * dt;
    s->m1 = (ac < 2500.0f) ? 1.0f : 0.0f;
    s->m2 = (ac >= 2500.0f) ? 1.0f : 0.0f;
    s->a0 = (ac - AC) / 1000.0f;
    s->a1 = fminf(fmaxf(s->a0, -1.0f), 1.0f);
    float e0 = -(ac - 2500.0f) / 50.0f;
    float e1 = fminf(fmaxf(e0, -10.0f), 0.0f);
    s->ee = expf(e1);
    float LF = (s->a1 * RT) * s->m1 + (s->ee * RT) * s->m2;
    s->sl = s->SLZ1 + LF * dt;
    s->SLZ1p = fmaxf(s->sl, 0.0f);
    s->Q2 = K2 * s->SLZ1p;
    s->SLZ2 = s->SLZ1p - s->Q2 * dt;
    /* :652 */
    s->Q = ((s->Q0 + s->Q1) + s->Q2) + s->IE;
}

static void hstep_bwd(float nz, const float *p, const hstep_t *s, const fluxgrad_t *g, float *a,
                      float *gp, float *gx)
{
    const float dt = HDT;
    const float BETA = p[HBVX_P_BETA], FC = p[HBVX_P_FC], K0 = p[HBVX_P_K0], K1 = p[HBVX_P_K1],
                K2 = p[HBVX_P_K2], LP = p[HBVX_P_LP], CFMAX = p[HBVX_P_CFMAX], CFR = p[HBVX_P_CFR],
                CWH = p[HBVX_P_CWH], BE = p[HBVX_P_BETAET], C = p[HBVX_P_C], RT = p[HBVX_P_RT],
                F0 = p[HBVX_P_F0], FMIN = p[HBVX_P_FMIN], ALPHA = p[HBVX_P_ALPHA];
    float wa, wb;
    float aSP3 = a[0] + g->gSWE, aMW3 = a[1], aSM4 = a[2], aSUZ4 = a[3], aSLZ2 = a[4];
    float aQ0 = g->gQ0 + g->gQ, aQ1 = g->gQ1 + g->gQ, aQ2 = g->gQ2 + g->gQ, aIE = g->gQ;
    /* lower box */
    aQ2 -= aSLZ2 * dt;
    float aSLZ1p = aSLZ2 + aQ2 * K2;
    gp[HBVX_P_K2] += aQ2 * s->SLZ1p;
    float as = (s->sl >= 0.0f) ? aSLZ1p : 0.0f;
    float aSLZ1 = as;
    float aLF = as * dt;
    float at1 = aLF * s->m1, at2 = aLF * s->m2;
    gp[HBVX_P_RT] += at1 * s->a1 + at2 * s->ee;
    float aa1 = at1 * RT;
    float aa0 = (s->a0 >= -1.0f && s->a0 <= 1.0f) ? aa1 : 0.0f;
    gp[HBVX_P_AC] += -(aa0 / 1000.0f);
    float aSLZ0 = aSLZ1;
    float aPERC = g->gPERC + aSLZ1 * dt;
    /* upper box */
    aQ1 -= aSUZ4 * dt;
    float aSUZ3 = aSUZ4 + aQ1 * K1;
    gp[HBVX_P_K1] += aQ1 * s->SUZ3;
    aQ0 -= aSUZ3 * dt;
    float aSUZ2 = aSUZ3;
    gp[HBVX_P_K0] += aQ0 * s->u0c;
    float au0 = (s->u0 >= 0.0f) ? aQ0 * K0 : 0.0f;
    aSUZ2 += au0;
    gp[HBVX_P_UZL] -= au0;
    aPERC -= aSUZ2 * dt;
    float aSUZ1 = aSUZ2;
    float aPERCm = aPERC / dt;
    minw(s->SUZ1, s->pdt, &wa, &wb);
    aSUZ1 += aPERCm * wa;
    gp[HBVX_P_PERC] += (aPERCm * wb) * dt;
    float aSUZ = aSUZ1;
    float are = aSUZ1 * dt;
    float arech = g->grech + are;
    float aexc = g->gexc + are;
    /* capillary rise */
    float az = (s->slc >= nz) ? aSLZ0 : 0.0f;
    float aSLZ = az;
    float acap = g->gcap - az * dt;
    float ay = (s->smc >= nz) ? aSM4 : 0.0f;
    float aSM3 = ay;
    acap += ay * dt;
    float acapm = acap / dt;
    minw(s->SLZ, s->capp, &wa, &wb);
    aSLZ += acapm * wa;
    float acsom = (acapm * wb) * dt;
    float acs = acsom * s->om;
    float arc = -(acsom * s->cs);
    gp[HBVX_P_C] += acs * s->SLZ;
    aSLZ += acs * C;
    float ax1 = (s->x1 <= 1.0f) ? arc : 0.0f;
    aSM3 += ax1 / FC;
    gp[HBVX_P_FC] += -ax1 * ((s->SM3 / FC) / FC);
    /* evapotranspiration */
    float add = (s->dd >= nz) ? aSM3 : 0.0f;
    float aSM2 = add;
    float aET = g->gET - add * dt;
    float aETm = aET / dt;
    minw(s->SM2, s->pedt, &wa, &wb);
    aSM2 += aETm * wa;
    float ape = (aETm * wb) * dt;
    float aef = g->gef + ape * s->PET;
    gx[2] = (ape * s->ef) / dt;
    float aef0 = (s->ef0 >= 0.0f && s->ef0 <= 1.0f) ? aef : 0.0f;
    float aq = aef0 * pow_dx(s->q, BE);
    gp[HBVX_P_BETAET] += aef0 * pow_dy(s->q, BE, s->ef0);
    aSM2 += aq / s->lpfc;
    float alpfc = -aq * ((s->SM2 / s->lpfc) / s->lpfc);
    gp[HBVX_P_LP] += alpfc * FC;
    gp[HBVX_P_FC] += alpfc * LP;
    /* excess */
    float aSM1 = aSM2;
    aexc -= aSM2 * dt;
    float ae0 = (s->e0 >= 0.0f) ? aexc : 0.0f;
    float ae0d = ae0 / dt;
    aSM1 += ae0d;
    gp[HBVX_P_FC] -= ae0d;
    /* soil: SM1 = SM + (infil - rech)*dt ; rech = infil*sw */
    float aSM = aSM1;
    float aird = aSM1 * dt;
    float ainfil = aird;
    arech -= aird;
    ainfil += arech * s->sw;
    float asw = arech * s->infil;
    float asw0 = (s->sw0 >= 0.0f && s->sw0 <= 1.0f) ? asw : 0.0f;
    float ar = asw0 * pow_dx(s->r, BETA);
    gp[HBVX_P_BETA] += asw0 * pow_dy(s->r, BETA, s->sw0);
    /* infiltration: infil = min(W, fcap) ; IE = max(W - fcap, 0) */
    minw(s->W, s->fcap, &wa, &wb);
    float aW = ainfil * wa;
    float afcap = ainfil * wb;
    float aie0 = (s->ie0 >= 0.0f) ? aIE : 0.0f;
    aW += aie0;
    afcap -= aie0;
    /* fcap = fmin + (F0 - fmin)*pw ; fmin = FMIN*F0 ; pw = (1-s)^ALPHA ; s = clamp(r, 0, 0.99) */
    float afmin = afcap;
    float adiff = afcap * s->pw;
    float apw = afcap * (F0 - s->fmin_);
    gp[HBVX_P_F0] += adiff;
    afmin -= adiff;
    gp[HBVX_P_FMIN] += afmin * F0;
    gp[HBVX_P_F0] += afmin * FMIN;
    float aoms = apw * pow_dx(s->oms, ALPHA);
    gp[HBVX_P_ALPHA] += apw * pow_dy(s->oms, ALPHA, s->pw);
    float as_ = -aoms;
    ar += (s->r >= 0.0f && s->r <= (float)(1.0 - 0.01)) ? as_ : 0.0f;
    aSM += ar / FC;
    gp[HBVX_P_FC] += -ar * ((s->SM / FC) / FC);
    float aRAIN = aW;
    float atosoil = g->gtosoil + aW;
    /* snow */
    float aMW2 = aMW3;
    atosoil -= aMW3 * dt;
    float ats0 = (s->ts0 >= 0.0f) ? atosoil : 0.0f;
    float ats0d = ats0 / dt;
    aMW2 += ats0d;
    gp[HBVX_P_CWH] -= ats0d * s->SP3;
    aSP3 -= ats0d * CWH;
    float aSP2 = aSP3;
    float arefr = aSP3 - aMW2;
    float aMW1 = aMW2;
    minw(s->rpcdt, s->MW1, &wa, &wb);
    float arpc = (arefr * wa) * dt;
    aMW1 += arefr * wb;
    float arp = (s->rp >= 0.0f) ? arpc : 0.0f;
    float acc = arp * s->dT2;
    gp[HBVX_P_CFR] += acc * CFMAX;
    gp[HBVX_P_CFMAX] += acc * CFR;
    float aTTe = arp * s->cc;
    float aTf = -(arp * s->cc);
    float aMW = aMW1;
    float amelt = aMW1 - aSP2;
    float aSP1 = aSP2;
    minw(s->mpcdt, s->SP1, &wa, &wb);
    float ampc = (amelt * wa) * dt;
    aSP1 += amelt * wb;
    float amp = (s->mp >= 0.0f) ? ampc : 0.0f;
    gp[HBVX_P_CFMAX] += amp * s->dT;
    aTTe -= amp * CFMAX;
    aTf += amp * CFMAX;
    float aSP = aSP1;
    float aSNOW = aSP1 * dt;
    gx[0] = (aSNOW * s->m_snow + aRAIN * s->m_rain) / dt;
    gx[1] = aTf;
    gp[HBVX_P_TT] += aTTe * s->mlo;
    a[0] = aSP * s->g0; a[1] = aMW * s->g1; a[2] = aSM * s->g2; a[3] = aSUZ * s->g3; a[4] = aSLZ * s->g4;
}

/* ------------------------------------------------------------------ */

static int check_desc(const hbvx_desc *d)
{
    if (!d) return fail(HBVX_E_NULL, "desc is NULL");
    if (d->abi_version != HBVX_ABI_VERSION) return fail(HBVX_E_ABI, "abi_version mismatch");
    if (d->T < 0 || d->B <= 0 || d->M <= 0 || d->M > 64) return fail(HBVX_E_SHAPE, "bad T/B/M");
    int ok = 0;
    if (d->model == HBVX_MODEL_HBV10) ok = (d->n_param == 12 || d->n_param == 13);
    else if (d->model == HBVX_MODEL_HBV11P) ok = (d->n_param == 14);
    else if (d->model == HBVX_MODEL_HBV20) ok = (d->n_param == 16);
    else if (d->model == HBVX_MODEL_HOURLY) ok = (d->n_param == 19);
    else if (d->model == HBVX_MODEL_HBVADJ) return fail(HBVX_E_UNSUPPORTED, "the implicit scheme's oracle is oracle/hbv_adj_oracle.py");
    else return fail(HBVX_E_UNSUPPORTED, "unknown model");
    if (!ok) return fail(HBVX_E_SHAPE, "n_param does not match model");
    if (!d->x && d->T > 0) return fail(HBVX_E_NULL, "forcing pointer is NULL");   /* an empty record has no forcings */
    if ((d->model == HBVX_MODEL_HBV20 || d->model == HBVX_MODEL_HOURLY) && (!d->ac || !d->elev))
        return fail(HBVX_E_NULL, "HBV 2.0 needs ac and elev");
    for (int i = 0; i < d->n_param; i++)
        if (!d->p[i].sta) return fail(HBVX_E_NULL, "static parameter pointer is NULL");
    return HBVX_OK;
}

static inline int has_betaet_(const hbvx_desc *d)
{
    return d->model != HBVX_MODEL_HBV10 || d->n_param == 13;
}

static inline void load_step_inputs(const hbvx_desc *d, int t, int b, int j, float *p,
                                    unit_t *u, step_t *s)
{
    for (int i = 0; i < d->n_param; i++) {
        u[i] = fetch_unit(d, i, t, b, j);
        p[i] = descale(u[i].u, d->p[i].lo, d->p[i].hi);
    }
    const float *xr = d->x + (int64_t)t * d->x_t_stride + (int64_t)b * d->x_b_stride;
    s->P = xr[d->ch_prcp];
    s->Tf = xr[d->ch_tmean];
    s->PET = xr[d->ch_pet];
}

/* hbv.py:423-511 with the parameter prep fused (see include/hbvx.h). */
int hbvx_forward(const hbvx_desc *d, const hbvx_fwd_out *out, void *stream)
{
    (void)stream;
    int rc = check_desc(d);
    if (rc) return rc;
    if (!out || !out->state_out) return fail(HBVX_E_NULL, "state_out is NULL");
    if (out->traj && HBVX_TRAJ_KIND(out->traj_layout) == HBVX_TRAJ_PACKED)
        return fail(HBVX_E_UNSUPPORTED, "the oracle keeps the trajectory in rows or checkpoints");
    /* HBVX_TRAJ_CKPT: storages entering days 0, K, 2K, ... only, [ceil(T/K), 5, N] */
    const int ckpt_k = (out->traj && HBVX_TRAJ_KIND(out->traj_layout) == HBVX_TRAJ_CKPT)
                           ? HBVX_TRAJ_CKPT_DAYS(out->traj_layout) : 0;
    if (out->traj && HBVX_TRAJ_KIND(out->traj_layout) == HBVX_TRAJ_CKPT && ckpt_k <= 0)
        return fail(HBVX_E_SHAPE, "checkpoint interval must be positive");
    const int T = d->T, B = d->B, M = d->M;
    const int64_t N = (int64_t)B * M;
    const int nf = out->n_flux;
    const int want_nf = (d->model == HBVX_MODEL_HBV10) ? 11 : 12;
    if (out->flux && nf != want_nf) return fail(HBVX_E_SHAPE, "n_flux does not match model");
    const int betaet = has_betaet_(d);

#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; b++) {
        float *acc = NULL;
        if (out->flux) acc = (float *)calloc((size_t)T * nf, sizeof(float));
        const float ac = d->ac ? d->ac[b] : 0.0f, elev = d->elev ? d->elev[b] : 0.0f;
        for (int j = 0; j < M; j++) {
            const int64_t n = (int64_t)b * M + j;
            float st[5];
            for (int k = 0; k < 5; k++)
                st[k] = d->state_in ? d->state_in[k * N + n] : 0.001f; /* hbv.py:131-134 */
            for (int t = 0; t < T; t++) {
                float p[HBVX_MAX_PARAM];
                unit_t u[HBVX_MAX_PARAM];
                step_t s;
                s.SP = st[0]; s.MW = st[1]; s.SM = st[2]; s.SUZ = st[3]; s.SLZ = st[4];
                load_step_inputs(d, t, b, j, p, u, &s);
                if (d->model == HBVX_MODEL_HOURLY) {
                    hstep_t h;
                    h.SPi = st[0]; h.MWi = st[1]; h.SMi = st[2]; h.SUZi = st[3]; h.SLZi = st[4];
                    h.P = s.P / HDT; h.Tf = s.Tf; h.PET = s.PET / HDT; /* hbv_2_hourly.py:485-487 */
                    hstep_fwd(d->nearzero, p, ac, elev, &h);
                    s.sw0 = h.sw0; s.ef0 = h.ef0;
                    s.SP3 = h.SP3; s.MW3 = h.MW3; s.SM4 = h.SM4; s.SUZ4 = h.SUZ4; s.SLZ2 = h.SLZ2;
                    s.Q = h.Q; s.Q0 = h.Q0; s.Q1 = h.Q1; s.Q2 = h.Q2; s.ET = h.ET; s.rech = h.rech;
                    s.exc = h.exc; s.ef = h.ef; s.tosoil = h.tosoil; s.PERC = h.PERC; s.cap = h.cap;
                } else {
                    step_fwd(d->model, betaet, d->nearzero, p, ac, elev, &s);
                }
                if (out->traj && ckpt_k) {
                    if (t % ckpt_k == 0)
                        for (int k = 0; k < 5; k++) out->traj[((int64_t)(t / ckpt_k) * 5 + k) * N + n] = st[k];
                } else if (out->traj) {
                    for (int k = 0; k < 5; k++) out->traj[((int64_t)k * (T + 1) + t) * N + n] = st[k];
                }
                if (out->aux) {
                    out->aux[((int64_t)0 * T + t) * N + n] = s.sw0;
                    out->aux[((int64_t)1 * T + t) * N + n] = s.ef0;
                }
                st[0] = s.SP3; st[1] = s.MW3; st[2] = s.SM4; st[3] = s.SUZ4; st[4] = s.SLZ2;
                if (acc) {
                    float *a = acc + (size_t)t * nf;
                    float wq = 1.0f;
                    if (d->muwts) /* hbv.py:511 (Qsimmu * muwts).sum(-1) */
                        wq = d->muwts[(int64_t)t * d->mu_t_stride + (int64_t)b * d->mu_b_stride + j];
                    a[HBVX_F_QSIM] += d->muwts ? s.Q * wq : s.Q;
                    a[HBVX_F_Q0] += s.Q0;
                    a[HBVX_F_Q1] += s.Q1;
                    a[HBVX_F_Q2] += s.Q2;
                    a[HBVX_F_AET] += s.ET;
                    a[HBVX_F_SWE] += s.SP3; /* hbv.py:499 stores the updated SNOWPACK */
                    a[HBVX_F_RECHARGE] += s.rech;
                    a[HBVX_F_EXCS] += s.exc;
                    a[HBVX_F_EVAPFACTOR] += s.ef;
                    a[HBVX_F_TOSOIL] += s.tosoil;
                    a[HBVX_F_PERC] += s.PERC;
                    if (nf > HBVX_F_CAPILLARY) a[HBVX_F_CAPILLARY] += s.cap;
                }
            }
            for (int k = 0; k < 5; k++) {
                out->state_out[k * N + n] = st[k];
                if (out->traj && !ckpt_k) out->traj[((int64_t)k * (T + 1) + T) * N + n] = st[k];
            }
        }
        if (acc) {
            for (int t = 0; t < T; t++)
                for (int k = 0; k < nf; k++) {
                    float v = acc[(size_t)t * nf + k];
                    /* hbv.py:509 mean(-1); the muwts form is a plain sum (hbv.py:511) */
                    if (!(k == HBVX_F_QSIM && d->muwts)) v = v / (float)M;
                    out->flux[((int64_t)k * T + t) * B + b] = v;
                }
            free(acc);
        }
    }
    return HBVX_OK;
}

/* The autograd tape of the same lines, by hand. */
int hbvx_backward(const hbvx_desc *d, const hbvx_bwd_io *io, void *stream)
{
    (void)stream;
    int rc = check_desc(d);
    if (rc) return rc;
    if (!io || !io->traj) return fail(HBVX_E_NULL, "traj is NULL");
    if (HBVX_TRAJ_KIND(io->traj_layout) == HBVX_TRAJ_PACKED)
        return fail(HBVX_E_UNSUPPORTED, "the oracle keeps the trajectory in rows or checkpoints");
    const int ckpt_k = HBVX_TRAJ_KIND(io->traj_layout) == HBVX_TRAJ_CKPT ? HBVX_TRAJ_CKPT_DAYS(io->traj_layout) : 0;
    if (HBVX_TRAJ_KIND(io->traj_layout) == HBVX_TRAJ_CKPT && ckpt_k <= 0)
        return fail(HBVX_E_SHAPE, "checkpoint interval must be positive");
    const int T = d->T, B = d->B, M = d->M;
    const int64_t N = (int64_t)B * M;
    const int nf = io->n_flux;
    const int betaet = has_betaet_(d);
    const float invM = 1.0f / (float)M;

#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; b++) {
        const float ac = d->ac ? d->ac[b] : 0.0f, elev = d->elev ? d->elev[b] : 0.0f;
        float *gxacc = NULL;
        if (io->grad_x) gxacc = (float *)calloc((size_t)T * 3, sizeof(float));
        for (int j = 0; j < M; j++) {
            const int64_t n = (int64_t)b * M + j;
            float a[5];
            for (int k = 0; k < 5; k++) a[k] = io->grad_state_out ? io->grad_state_out[k * N + n] : 0.0f;
            float gsta[HBVX_MAX_PARAM]; /* dL/d(unit static value), summed over t */
            float usta[HBVX_MAX_PARAM];
            for (int i = 0; i < HBVX_MAX_PARAM; i++) gsta[i] = 0.0f, usta[i] = 0.0f;
            /* checkpoints: re-materialise this lane's storages day by day, segment by segment */
            float *lane_traj = NULL;
            if (ckpt_k) {
                lane_traj = (float *)malloc((size_t)(T > 0 ? T : 1) * 5 * sizeof(float));
                for (int t0 = 0; t0 < T; t0 += ckpt_k) {
                    float st[5];
                    for (int k = 0; k < 5; k++) st[k] = io->traj[((int64_t)(t0 / ckpt_k) * 5 + k) * N + n];
                    for (int t = t0; t < T && t < t0 + ckpt_k; t++) {
                        float p[HBVX_MAX_PARAM];
                        unit_t u[HBVX_MAX_PARAM];
                        step_t s;
                        for (int k = 0; k < 5; k++) lane_traj[(size_t)t * 5 + k] = st[k];
                        s.SP = st[0]; s.MW = st[1]; s.SM = st[2]; s.SUZ = st[3]; s.SLZ = st[4];
                        load_step_inputs(d, t, b, j, p, u, &s);
                        if (d->model == HBVX_MODEL_HOURLY) {
                            hstep_t h;
                            h.SPi = st[0]; h.MWi = st[1]; h.SMi = st[2]; h.SUZi = st[3]; h.SLZi = st[4];
                            h.P = s.P / HDT; h.Tf = s.Tf; h.PET = s.PET / HDT;
                            hstep_fwd(d->nearzero, p, ac, elev, &h);
                            st[0] = h.SP3; st[1] = h.MW3; st[2] = h.SM4; st[3] = h.SUZ4; st[4] = h.SLZ2;
                        } else {
                            step_fwd(d->model, betaet, d->nearzero, p, ac, elev, &s);
                            st[0] = s.SP3; st[1] = s.MW3; st[2] = s.SM4; st[3] = s.SUZ4; st[4] = s.SLZ2;
                        }
                    }
                }
            }
            for (int t = T - 1; t >= 0; t--) {
                float p[HBVX_MAX_PARAM], gp[HBVX_MAX_PARAM], gx[3];
                unit_t u[HBVX_MAX_PARAM];
                step_t s;
                if (lane_traj) {
                    s.SP = lane_traj[(size_t)t * 5]; s.MW = lane_traj[(size_t)t * 5 + 1]; s.SM = lane_traj[(size_t)t * 5 + 2];
                    s.SUZ = lane_traj[(size_t)t * 5 + 3]; s.SLZ = lane_traj[(size_t)t * 5 + 4];
                } else {
                s.SP = io->traj[((int64_t)0 * (T + 1) + t) * N + n];
                s.MW = io->traj[((int64_t)1 * (T + 1) + t) * N + n];
                s.SM = io->traj[((int64_t)2 * (T + 1) + t) * N + n];
                s.SUZ = io->traj[((int64_t)3 * (T + 1) + t) * N + n];
                s.SLZ = io->traj[((int64_t)4 * (T + 1) + t) * N + n];
                }
                load_step_inputs(d, t, b, j, p, u, &s);
                hstep_t h;
                if (d->model == HBVX_MODEL_HOURLY) {
                    h.SPi = s.SP; h.MWi = s.MW; h.SMi = s.SM; h.SUZi = s.SUZ; h.SLZi = s.SLZ;
                    h.P = s.P / HDT; h.Tf = s.Tf; h.PET = s.PET / HDT;
                    hstep_fwd(d->nearzero, p, ac, elev, &h);
                    s.Q = h.Q;
                } else {
                    step_fwd(d->model, betaet, d->nearzero, p, ac, elev, &s);
                }
                for (int i = 0; i < HBVX_MAX_PARAM; i++) gp[i] = 0.0f;
                fluxgrad_t g;
                const float *gf = io->grad_flux;
                const float *gf4 = io->grad_flux4;
#define GF(k) ((gf ? gf[((int64_t)(k) * T + t) * B + b] : 0.0f) + \
               ((gf4 && (k) < 4) ? gf4[((int64_t)(k) * T + t) * B + b] : 0.0f))
                float wq = invM; /* mean(-1) backward: grad / M */
                if (d->muwts)
                    wq = d->muwts[(int64_t)t * d->mu_t_stride + (int64_t)b * d->mu_b_stride + j];
                g.gQ = GF(HBVX_F_QSIM) * wq;
                g.gQ0 = GF(HBVX_F_Q0) * invM;
                g.gQ1 = GF(HBVX_F_Q1) * invM;
                g.gQ2 = GF(HBVX_F_Q2) * invM;
                g.gET = GF(HBVX_F_AET) * invM;
                g.gSWE = GF(HBVX_F_SWE) * invM;
                g.grech = GF(HBVX_F_RECHARGE) * invM;
                g.gexc = GF(HBVX_F_EXCS) * invM;
                g.gef = GF(HBVX_F_EVAPFACTOR) * invM;
                g.gtosoil = GF(HBVX_F_TOSOIL) * invM;
                g.gPERC = GF(HBVX_F_PERC) * invM;
                g.gcap = (nf > HBVX_F_CAPILLARY) ? GF(HBVX_F_CAPILLARY) * invM : 0.0f;
                if (io->grad_muwts)
                    io->grad_muwts[((int64_t)t * B + b) * M + j] = GF(HBVX_F_QSIM) * s.Q;
#undef GF
                if (d->model == HBVX_MODEL_HOURLY) hstep_bwd(d->nearzero, p, &h, &g, a, gp, gx);
                else step_bwd(d->model, betaet, d->nearzero, p, &s, &g, a, gp, gx);
                if (gxacc) {
                    gxacc[t * 3 + 0] += gx[0];
                    gxacc[t * 3 + 1] += gx[1];
                    gxacc[t * 3 + 2] += gx[2];
                }
                /* de-scaling and sigmoid backward; dynamic rows are written, the
                 * static row's contributions are summed over time. */
                for (int i = 0; i < d->n_param; i++) {
                    float gu = gp[i] * (d->p[i].hi - d->p[i].lo);
                    if (u[i].from_dyn) {
                        float gr = d->raw_sigmoid ? gu * (u[i].u * (1.0f - u[i].u)) : gu;
                        if (io->g[i].dyn)
                            io->g[i].dyn[(int64_t)t * io->g[i].dyn_t_stride +
                                         (int64_t)b * io->g[i].dyn_b_stride + j] = gr;
                    } else {
                        gsta[i] += gu;
                        usta[i] = u[i].u;
                        if (d->p[i].dyn && io->g[i].dyn) /* dropped basin: zero dyn grad */
                            io->g[i].dyn[(int64_t)t * io->g[i].dyn_t_stride +
                                         (int64_t)b * io->g[i].dyn_b_stride + j] = 0.0f;
                    }
                }
            }
            for (int i = 0; i < d->n_param; i++) {
                if (!io->g[i].sta) continue;
                float gr = d->raw_sigmoid ? gsta[i] * (usta[i] * (1.0f - usta[i])) : gsta[i];
                io->g[i].sta[(int64_t)b * io->g[i].sta_b_stride + j] += gr;
            }
            if (io->grad_state_in)
                for (int k = 0; k < 5; k++) io->grad_state_in[k * N + n] = a[k];
            free(lane_traj);
        }
        if (gxacc) {
            for (int t = 0; t < T; t++) {
                float *gr = io->grad_x + (int64_t)t * d->x_t_stride + (int64_t)b * d->x_b_stride;
                gr[d->ch_prcp] = gxacc[t * 3 + 0];
                gr[d->ch_tmean] = gxacc[t * 3 + 1];
                gr[d->ch_pet] = gxacc[t * 3 + 2];
            }
            free(gxacc);
        }
    }
    return HBVX_OK;
}

/* ------------------------------------------------------------------ */
/* routing: core/calc/uh_routing.py:5-57                               */

static int check_route(const hbvx_route_desc *r)
{
    if (!r) return fail(HBVX_E_NULL, "route desc is NULL");
    if (r->abi_version != HBVX_ABI_VERSION) return fail(HBVX_E_ABI, "abi_version mismatch");
    if (r->T <= 0 || r->B <= 0 || r->S <= 0) return fail(HBVX_E_SHAPE, "bad T/B/S");
    int L = r->T < HBVX_UH_MAXLEN ? r->T : HBVX_UH_MAXLEN;
    if (r->L != L) return fail(HBVX_E_SHAPE, "L must be min(T, 15)");
    if (!r->ra || !r->rb) return fail(HBVX_E_NULL, "routing parameter pointer is NULL");
    return HBVX_OK;
}

/* uh_routing.py:5-22 for one basin. */
static void uh_gamma_one(const hbvx_route_desc *r, int b, float *w, float *ua, float *ub)
{
    float va = r->ra[(int64_t)b * r->r_stride], vb = r->rb[(int64_t)b * r->r_stride];
    *ua = r->raw_sigmoid ? sigmoidf_(va) : va;
    *ub = r->raw_sigmoid ? sigmoidf_(vb) : vb;
    float a = descale(*ua, r->a_lo, r->a_hi), bb = descale(*ub, r->b_lo, r->b_hi);
    float aa = fmaxf(a, 0.0f) + 0.1f;     /* :11-13 */
    float theta = fmaxf(bb, 0.0f) + 0.5f; /* :14 */
    float denom = expf(lgammaf(aa)) * powf(theta, aa); /* :17 */
    float sum = 0.0f;
    for (int k = 0; k < r->L; k++) {
        float t = (float)k + 0.5f;        /* :15 */
        float mid = powf(t, aa - 1.0f);   /* :18 */
        float right = expf(-t / theta);   /* :19 */
        w[k] = 1.0f / denom * mid * right; /* :20 */
        sum += w[k];
    }
    for (int k = 0; k < r->L; k++) w[k] = w[k] / sum; /* :21 */
}

int hbvx_route_forward(const hbvx_route_desc *r, const float *q, float *uh, float *q_rout,
                       void *stream)
{
    (void)stream;
    int rc = check_route(r);
    if (rc) return rc;
    if (!q || !uh || !q_rout) return fail(HBVX_E_NULL, "route buffer is NULL");
    const int T = r->T, B = r->B, L = r->L;
#pragma omp parallel for
    for (int b = 0; b < B; b++) {
        float w[HBVX_UH_MAXLEN], ua, ub;
        uh_gamma_one(r, b, w, &ua, &ub);
        for (int k = 0; k < L; k++) uh[(int64_t)b * L + k] = w[k];
        /* uh_routing.py:44-57: y[t] = sum_k UH[k] * x[t-k], zero history */
        for (int s = 0; s < r->S; s++)
            for (int t = 0; t < T; t++) {
                float y = 0.0f;
                for (int k = 0; k < L && k <= t; k++)
                    y += w[k] * q[((int64_t)s * T + (t - k)) * B + b];
                q_rout[((int64_t)s * T + t) * B + b] = y;
            }
    }
    return HBVX_OK;
}

uint64_t hbvx_route_workspace_bytes(const hbvx_route_desc *r)
{
    (void)r;
    return 0; /* the oracle reduces in place */
}

int hbvx_route_backward(const hbvx_route_desc *r, const float *q, const float *uh,
                        const float *grad_q_rout, float *grad_q, float *grad_ra, float *grad_rb,
                        void *workspace, uint64_t workspace_bytes, void *stream)
{
    (void)stream; (void)workspace; (void)workspace_bytes;
    int rc = check_route(r);
    if (rc) return rc;
    if (!q || !uh || !grad_q_rout || !grad_q) return fail(HBVX_E_NULL, "route buffer is NULL");
    const int T = r->T, B = r->B, L = r->L;
#pragma omp parallel for
    for (int b = 0; b < B; b++) {
        const float *w = uh + (int64_t)b * L;
        double gw[HBVX_UH_MAXLEN];
        for (int k = 0; k < L; k++) gw[k] = 0.0;
        for (int s = 0; s < r->S; s++)
            for (int t = 0; t < T; t++) {
                /* conv1d backward w.r.t. the input: correlation with the UH */
                float gx = 0.0f;
                for (int k = 0; k < L && t + k < T; k++)
                    gx += w[k] * grad_q_rout[((int64_t)s * T + (t + k)) * B + b];
                grad_q[((int64_t)s * T + t) * B + b] = gx;
                /* ... and w.r.t. the UH taps */
                float gy = grad_q_rout[((int64_t)s * T + t) * B + b];
                for (int k = 0; k < L && k <= t; k++)
                    gw[k] += (double)gy * (double)q[((int64_t)s * T + (t - k)) * B + b];
            }
        if (!grad_ra && !grad_rb) continue;
        /* w_k = u_k / sum(u), u_k = t_k^(aa-1) exp(-t_k/theta) / (Gamma(aa) theta^aa):
         * the Gamma and theta^aa factors cancel in the normalisation (uh_routing.py:17-21),
         * so dw_k/daa = w_k (ln t_k - sum_j w_j ln t_j),
         *    dw_k/dtheta = w_k (t_k - sum_j w_j t_j) / theta^2. */
        float va = r->ra[(int64_t)b * r->r_stride], vb = r->rb[(int64_t)b * r->r_stride];
        float ua = r->raw_sigmoid ? sigmoidf_(va) : va, ub = r->raw_sigmoid ? sigmoidf_(vb) : vb;
        float a = descale(ua, r->a_lo, r->a_hi), bb = descale(ub, r->b_lo, r->b_hi);
        double theta = (double)(fmaxf(bb, 0.0f) + 0.5f);
        double mlt = 0.0, mt = 0.0;
        for (int k = 0; k < L; k++) {
            double tk = k + 0.5;
            mlt += w[k] * log(tk);
            mt += w[k] * tk;
        }
        double gaa = 0.0, gth = 0.0;
        for (int k = 0; k < L; k++) {
            double tk = k + 0.5;
            gaa += gw[k] * w[k] * (log(tk) - mlt);
            gth += gw[k] * w[k] * (tk - mt) / (theta * theta);
        }
        double ga = (a > 0.0f) ? gaa : 0.0;  /* relu backward */
        double gb = (bb > 0.0f) ? gth : 0.0;
        double gua = ga * (double)(r->a_hi - r->a_lo), gub = gb * (double)(r->b_hi - r->b_lo);
        if (r->raw_sigmoid) {
            gua *= (double)ua * (1.0 - (double)ua);
            gub *= (double)ub * (1.0 - (double)ub);
        }
        if (grad_ra) grad_ra[(int64_t)b * r->r_stride] += (float)gua;
        if (grad_rb) grad_rb[(int64_t)b * r->r_stride] += (float)gub;
    }
    return HBVX_OK;
}

/* diagnostics entry of the ABI: the oracle's pow is libm's powf */
int hbvx_selftest_pow(const float *x, const float *y, float *out, int n, void *stream)
{
    (void)stream;
    for (int i = 0; i < n; i++) out[i] = powf(x[i], y[i]);
    return HBVX_OK;
}

int hbvx_selftest_div(const float *x, const float *y, float *out, int n, void *stream)
{
    (void)stream;
    for (int i = 0; i < n; i++) out[i] = x[i] / y[i];
    return HBVX_OK;
}

/* The implicit scheme (hbv_adj.py) is restated in oracle/hbv_adj_oracle.py (float64, autograd). */
int hbvx_adj_forward(const hbvx_desc *d, const hbvx_fwd_out *out, void *stream)
{
    (void)d; (void)out; (void)stream;
    return fail(HBVX_E_UNSUPPORTED, "see oracle/hbv_adj_oracle.py");
}
int hbvx_adj_backward(const hbvx_desc *d, const hbvx_bwd_io *io, void *stream)
{
    (void)d; (void)io; (void)stream;
    return fail(HBVX_E_UNSUPPORTED, "see oracle/hbv_adj_oracle.py");
}

uint64_t hbvx_backward_workspace_bytes(const hbvx_desc *d)
{
    (void)d;
    return 0; /* the oracle sweeps serially */
}

/* hbv.py:562-567 */
int hbvx_bfi(int32_t T, int32_t B, const float *qs, const float *q2, float nearzero, float *bfi,
             void *stream)
{
    (void)stream;
    for (int b = 0; b < B; b++) {
        float s0 = 0.0f, s2 = 0.0f;
        for (int t = 0; t < T; t++) {
            s0 += qs[(int64_t)t * B + b];
            s2 += q2[(int64_t)t * B + b];
        }
        bfi[b] = 100.0f * (s2 / (s0 + nearzero));
    }
    return HBVX_OK;
}

/* ------------------------------------------------------------------ */
/* gage routing: hbv_2_hourly.py:800-897                                 */

static int check_gage(const hbvx_gage_desc *r)
{
    if (!r) return fail(HBVX_E_NULL, "gage desc is NULL");
    if (r->abi_version != HBVX_ABI_VERSION) return fail(HBVX_E_ABI, "abi_version mismatch");
    if (r->T <= 0 || r->U <= 0 || r->G <= 0 || r->NPAIR < 0) return fail(HBVX_E_SHAPE, "bad T/U/G/NPAIR");
    int L = r->T < HBVX_GAGE_MAXLEN ? r->T : HBVX_GAGE_MAXLEN;
    if (r->L != L) return fail(HBVX_E_SHAPE, "L must be min(T, 72)");
    if (!r->pair_unit || !r->gage_ptr || !r->pair_gage || !r->unit_ptr || !r->unit_pairs || !r->areas ||
        !r->denom || !r->dp)
        return fail(HBVX_E_NULL, "gage routing pointer is NULL");
    return HBVX_OK;
}

/* uh_gamma (uh_routing.py:5-22) then _frac_shift1d (hbv_2_hourly.py:857-897) for one pair */
static void gage_uh_one(const hbvx_gage_desc *r, int p, float *w, float *us, float *kk_out, float *f_out)
{
    const int L = r->L;
    float a = descale(r->dp[p * 3 + 0], r->a_lo, r->a_hi);
    float b = descale(r->dp[p * 3 + 1], r->b_lo, r->b_hi);
    float tau = descale(r->dp[p * 3 + 2], r->tau_lo, r->tau_hi);
    float aa = fmaxf(a, 0.0f) + 0.1f, theta = fmaxf(b, 0.0f) + 0.5f;
    float denom = expf(lgammaf(aa)) * powf(theta, aa), sum = 0.0f;
    for (int k = 0; k < L; k++) {
        float t = (float)k + 0.5f;
        w[k] = 1.0f / denom * powf(t, aa - 1.0f) * expf(-t / theta);
        sum += w[k];
    }
    for (int k = 0; k < L; k++) w[k] = w[k] / sum;
    float kk = floorf(tau), f = tau - kk;
    if (!r->lag_uh) { kk = 0.0f; f = 0.0f; }
    for (int k = 0; k < L; k++) {
        float i0 = (float)k - kk, i1 = (float)k - (kk + 1.0f);
        float w0 = (i0 >= 0.0f && i0 <= (float)(L - 1)) ? w[(int)i0] : 0.0f;
        float w1 = (i1 >= 0.0f && i1 <= (float)(L - 1)) ? w[(int)i1] : 0.0f;
        us[k] = r->lag_uh ? (1.0f - f) * w0 + f * w1 : w[k];
    }
    *kk_out = kk;
    *f_out = f;
}

uint64_t hbvx_gage_route_workspace_bytes(const hbvx_gage_desc *r)
{
    (void)r;
    return 0;
}

int hbvx_gage_route_forward(const hbvx_gage_desc *r, const float *qs, float *uh, float *out,
                            void *workspace, uint64_t workspace_bytes, void *stream)
{
    (void)stream; (void)workspace; (void)workspace_bytes;
    int rc = check_gage(r);
    if (rc) return rc;
    if (!qs || !uh || !out) return fail(HBVX_E_NULL, "gage routing buffer is NULL");
    const int T = r->T, U = r->U, G = r->G, L = r->L;
    for (int p = 0; p < r->NPAIR; p++) {
        float w[HBVX_GAGE_MAXLEN], kk, f;
        gage_uh_one(r, p, w, uh + (int64_t)p * L, &kk, &f);
    }
    for (int g = 0; g < G; g++)
        for (int t = 0; t < T; t++) {
            float acc = 0.0f;
            for (int p = r->gage_ptr[g]; p < r->gage_ptr[g + 1]; p++) {
                const int u = r->pair_unit[p];
                float y = 0.0f; /* uh_conv of the area-weighted unit series (:819-837) */
                for (int k = 0; k < L && k <= t; k++)
                    y += uh[(int64_t)p * L + k] * (qs[(int64_t)(t - k) * U + u] * r->areas[u]);
                acc += y; /* scatter_add_ over the pairs of the gage (:841-846) */
            }
            out[(int64_t)t * G + g] = acc / r->denom[g]; /* :849-850 */
        }
    return HBVX_OK;
}

int hbvx_gage_route_backward(const hbvx_gage_desc *r, const float *qs, const float *uh,
                             const float *grad_out, float *grad_qs, float *grad_dp,
                             void *workspace, uint64_t workspace_bytes, void *stream)
{
    (void)stream; (void)workspace; (void)workspace_bytes;
    int rc = check_gage(r);
    if (rc) return rc;
    if (!qs || !uh || !grad_out || !grad_qs || !grad_dp) return fail(HBVX_E_NULL, "gage routing buffer is NULL");
    const int T = r->T, U = r->U, G = r->G, L = r->L;
    for (int64_t i = 0; i < (int64_t)T * U; i++) grad_qs[i] = 0.0f;
    for (int p = 0; p < r->NPAIR; p++) {
        const int u = r->pair_unit[p], g = r->pair_gage[p];
        double guh[HBVX_GAGE_MAXLEN];
        for (int k = 0; k < L; k++) guh[k] = 0.0;
        for (int t = 0; t < T; t++) {
            const float go = grad_out[(int64_t)t * G + g] / r->denom[g];
            for (int k = 0; k < L && k <= t; k++) {
                grad_qs[(int64_t)(t - k) * U + u] += go * uh[(int64_t)p * L + k] * r->areas[u];
                guh[k] += (double)go * (double)(qs[(int64_t)(t - k) * U + u] * r->areas[u]);
            }
        }
        /* shifted UH -> (w, f) -> (aa, theta, tau) */
        float w[HBVX_GAGE_MAXLEN], us[HBVX_GAGE_MAXLEN], kk, f;
        gage_uh_one(r, p, w, us, &kk, &f);
        double gw[HBVX_GAGE_MAXLEN], gf = 0.0;
        for (int j = 0; j < L; j++) gw[j] = 0.0;
        if (r->lag_uh) {
            for (int k = 0; k < L; k++) {
                int i0 = k - (int)kk, i1 = k - ((int)kk + 1);
                double w0 = (i0 >= 0 && i0 <= L - 1) ? w[i0] : 0.0, w1 = (i1 >= 0 && i1 <= L - 1) ? w[i1] : 0.0;
                gf += guh[k] * (w1 - w0);
                if (i0 >= 0 && i0 <= L - 1) gw[i0] += guh[k] * (1.0 - f);
                if (i1 >= 0 && i1 <= L - 1) gw[i1] += guh[k] * f;
            }
        } else {
            for (int k = 0; k < L; k++) gw[k] = guh[k];
        }
        float a = descale(r->dp[p * 3 + 0], r->a_lo, r->a_hi), b = descale(r->dp[p * 3 + 1], r->b_lo, r->b_hi);
        double theta = (double)(fmaxf(b, 0.0f) + 0.5f), mlt = 0.0, mt = 0.0;
        for (int k = 0; k < L; k++) { double tk = k + 0.5; mlt += w[k] * log(tk); mt += w[k] * tk; }
        double gaa = 0.0, gth = 0.0;
        for (int k = 0; k < L; k++) {
            double tk = k + 0.5;
            gaa += gw[k] * w[k] * (log(tk) - mlt);
            gth += gw[k] * w[k] * (tk - mt) / (theta * theta);
        }
        grad_dp[p * 3 + 0] = (float)(((a > 0.0f) ? gaa : 0.0) * (double)(r->a_hi - r->a_lo));
        grad_dp[p * 3 + 1] = (float)(((b > 0.0f) ? gth : 0.0) * (double)(r->b_hi - r->b_lo));
        grad_dp[p * 3 + 2] = (float)((r->lag_uh ? gf : 0.0) * (double)(r->tau_hi - r->tau_lo));
    }
    return HBVX_OK;
}

/* ------------------------------------------------------------------------- */
/* Sequence LSTM (include/hbvx_lstm.h): the caller side of the hot path
 * (not in the reference repository).  Semantics are
 * torch.nn.LSTM's, one layer, zero initial state:
 *   gates = x W_ih^T + b_ih + h_{t-1} W_hh^T + b_hh ; i, f, o = sigmoid, g = tanh ;
 *   c_t = f c_{t-1} + i g ; h_t = o tanh(c_t).
 * tests/test_lstm.py pins this restatement against torch.nn.LSTM on the CPU
 * (values and autograd gradients). */
#include "../include/hbvx_lstm.h"

static int check_lstm(const hbvx_lstm_desc *d)
{
    if (!d) return fail(HBVX_E_NULL, "lstm desc is NULL");
    if (d->abi_version != HBVX_LSTM_ABI_VERSION) return fail(HBVX_E_ABI, "lstm abi_version mismatch");
    if (d->T <= 0 || d->B <= 0 || d->H <= 0) return fail(HBVX_E_SHAPE, "lstm T/B/H out of range");
    return 0;
}

uint64_t hbvx_lstm_workspace_bytes(const hbvx_lstm_desc *d)
{
    (void)d;
    return 0;
}

static float lstm_sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

int hbvx_lstm_forward(const hbvx_lstm_desc *d, const float *w_hh, const float *gx, float *gates,
                      float *c_all, float *h_all, void *workspace, uint64_t workspace_bytes,
                      void *stream)
{
    (void)workspace; (void)workspace_bytes; (void)stream;
    int rc = check_lstm(d);
    if (rc) return rc;
    if (!w_hh || !gx || !gates || !c_all || !h_all) return fail(HBVX_E_NULL, "lstm buffer is NULL");
    const int T = d->T, B = d->B, H = d->H;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int t = 0; t < T; ++t) {
            const float *hp = t ? h_all + ((size_t)(t - 1) * B + b) * H : NULL;
            const float *cp = t ? c_all + ((size_t)(t - 1) * B + b) * H : NULL;
            for (int u = 0; u < H; ++u) {
                const size_t e = ((size_t)t * B + b) * H + u;
                float pre[4];
                for (int g = 0; g < 4; ++g) {
                    float acc = gx[e * 4 + g];
                    if (hp)
                        for (int k = 0; k < H; ++k) acc += w_hh[((size_t)g * H + u) * H + k] * hp[k];
                    pre[g] = acc;
                }
                const float ig = lstm_sigmoidf(pre[0]), fg = lstm_sigmoidf(pre[1]);
                const float gg = tanhf(pre[2]), og = lstm_sigmoidf(pre[3]);
                const float c = fg * (cp ? cp[u] : 0.0f) + ig * gg;
                gates[e * 4 + 0] = ig; gates[e * 4 + 1] = fg; gates[e * 4 + 2] = gg; gates[e * 4 + 3] = og;
                c_all[e] = c;
                h_all[e] = og * tanhf(c);
            }
        }
    }
    return 0;
}

int hbvx_lstm_backward(const hbvx_lstm_desc *d, const float *w_hh, const float *gates,
                       const float *c_all, const float *grad_h, float *grad_gates,
                       void *workspace, uint64_t workspace_bytes, void *stream)
{
    (void)workspace; (void)workspace_bytes; (void)stream;
    int rc = check_lstm(d);
    if (rc) return rc;
    if (!w_hh || !gates || !c_all || !grad_h || !grad_gates) return fail(HBVX_E_NULL, "lstm buffer is NULL");
    if (gates == grad_gates) return fail(HBVX_E_UNSUPPORTED, "lstm grad_gates must not alias gates");
    const int T = d->T, B = d->B, H = d->H;
    int bad = 0;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        float *dc = (float *)calloc((size_t)H, sizeof(float));
        float *dhr = (float *)calloc((size_t)H, sizeof(float));
        if (!dc || !dhr) {
            bad = 1;
            free(dc); free(dhr);
            continue;
        }
        for (int t = T - 1; t >= 0; --t) {
            for (int u = 0; u < H; ++u) {
                const size_t e = ((size_t)t * B + b) * H + u;
                const float ig = gates[e * 4], fg = gates[e * 4 + 1], gg = gates[e * 4 + 2], og = gates[e * 4 + 3];
                const float cp = t ? c_all[e - (size_t)B * H] : 0.0f;
                const float tc = tanhf(c_all[e]);
                const float dh = grad_h[e] + dhr[u];
                const float dcu = dc[u] + dh * og * (1.0f - tc * tc);
                grad_gates[e * 4 + 0] = dcu * gg * ig * (1.0f - ig);
                grad_gates[e * 4 + 1] = dcu * cp * fg * (1.0f - fg);
                grad_gates[e * 4 + 2] = dcu * ig * (1.0f - gg * gg);
                grad_gates[e * 4 + 3] = dh * tc * og * (1.0f - og);
                dc[u] = dcu * fg;
            }
            /* dh_{t-1} = dgates_t W_hh */
            for (int k = 0; k < H; ++k) dhr[k] = 0.0f;
            if (t)
                for (int u = 0; u < H; ++u)
                    for (int g = 0; g < 4; ++g) {
                        const float dg = grad_gates[(((size_t)t * B + b) * H + u) * 4 + g];
                        const float *wr = w_hh + ((size_t)g * H + u) * H;
                        for (int k = 0; k < H; ++k) dhr[k] += dg * wr[k];
                    }
        }
        free(dc); free(dhr);
    }
    return bad ? fail(HBVX_E_DEVICE, "lstm: out of memory") : 0;
}

int hbvx_lstm_check(const hbvx_lstm_desc *d, const void *workspace, void *stream)
{
    (void)workspace; (void)stream;
    return check_lstm(d);
}

/* include/hbvx.h: zero fill of the dense gradient buffers. */
int hbvx_zero(void *ptr, uint64_t bytes, void *stream)
{
    (void)stream;
    if (!ptr && bytes) return fail(HBVX_E_NULL, "hbvx_zero: buffer is NULL");
    if (bytes) memset(ptr, 0, (size_t)bytes);
    return 0;
}

/* include/hbvx.h: zero everything except the kept column groups on rows [r0, r1). */
int hbvx_zero_except(float *ptr, int64_t rows, int32_t width, int64_t r0, int64_t r1, int32_t group_w,
                     uint32_t keep_groups, void *stream)
{
    (void)stream;
    if (rows < 0 || width <= 0 || group_w <= 0) return fail(HBVX_E_SHAPE, "hbvx_zero_except: bad shape");
    if (!ptr && rows) return fail(HBVX_E_NULL, "hbvx_zero_except: buffer is NULL");
    for (int64_t r = 0; r < rows; r++)
        for (int32_t c = 0; c < width; c++) {
            const int32_t g = c / group_w;
            const int kept = r >= r0 && r < r1 && g < 32 && ((keep_groups >> g) & 1u);
            if (!kept) ptr[r * width + c] = 0.0f;
        }
    return 0;
}
