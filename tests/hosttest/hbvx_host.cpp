// hbvx_host.cpp -- TEST HARNESS: runs the product's device math (hydrodl2_amd/csrc/hbv_step.h)
// on the host behind the same C ABI, so the re-typed step/adjoint can be checked against the
// golden fixtures without a GPU.  Lives under tests/, is built only by
// tests/test_step_math_host.py and is never shipped or loaded by the package.  Routing is not
// part of hbv_step.h and is not provided here.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/hbvx.h"
#include "../../include/hbvx_lstm.h"
#include <dlfcn.h>

#include "../../hydrodl2_amd/csrc/hbv_step.h"
#include "../../hydrodl2_amd/csrc/hbv_adj_step.h"

using namespace hbvx;

static thread_local char g_err[256] = "";
extern "C" int hbvx_version(void) { return HBVX_ABI_VERSION; }
extern "C" int hbvx_preferred_traj_layout(const hbvx_desc *) { return HBVX_TRAJ_ROWS; }
extern "C" uint64_t hbvx_ckpt_workspace_bytes(const hbvx_desc *, int32_t) { return 0; }
extern "C" const char *hbvx_last_error(void) { return g_err; }
extern "C" const char *hbvx_last_dispatch(int) { return "host-steptest"; }
extern "C" int hbvx_zero_in_launch(void) { return 0; }
extern "C" int hbvx_zero_rest(void *ptr, uint64_t bytes, const void *, void *)
{
    if (bytes) memset(ptr, 0, (size_t)bytes);
    return 0;
}
extern "C" const char *hbvx_backend(void) { return "cpu-steptest"; }
extern "C" uint64_t hbvx_sizeof(int w)
{
    switch (w) {
    case 0: return sizeof(hbvx_desc);
    case 1: return sizeof(hbvx_fwd_out);
    case 2: return sizeof(hbvx_bwd_io);
    case 3: return sizeof(hbvx_route_desc);
    case 4: return sizeof(hbvx_param_src);
    case 5: return sizeof(hbvx_param_grad);
    case 6: return sizeof(hbvx_gage_desc);
    }
    return 0;
}

template <int MODEL, bool BETAET>
static void run_fwd(const hbvx_desc &d, const hbvx_fwd_out &o)
{
    const int T = d.T, B = d.B, M = d.M, NP = d.n_param;
    const int64_t N = (int64_t)B * M;
    std::vector<float> acc;
    for (int b = 0; b < B; b++) {
        if (o.flux) acc.assign((size_t)T * o.n_flux, 0.0f);
        for (int j = 0; j < M; j++) {
            const int64_t n = (int64_t)b * M + j;
            float st[5], p[NPARAM_MAX] = {0};
            for (int k = 0; k < 5; k++) st[k] = d.state_in ? d.state_in[k * N + n] : 0.001f;
            for (int t = 0; t < T; t++) {
                for (int i = 0; i < NP; i++) {
                    const hbvx_param_src &s = d.p[i];
                    bool ud = s.dyn && !(s.drop && s.drop[b]);
                    float v = ud ? s.dyn[(int64_t)t * s.dyn_t_stride + (int64_t)b * s.dyn_b_stride + j]
                                 : s.sta[(int64_t)b * s.sta_b_stride + j];
                    v = d.raw_sigmoid ? sigmoid_(v) : v;
                    p[i] = descale_(v, s.lo, s.hi);
                }
                Step<MODEL, BETAET> s;
                const float *xr = d.x + (int64_t)t * d.x_t_stride + (int64_t)b * d.x_b_stride;
                s.P = xr[d.ch_prcp]; s.Tf = xr[d.ch_tmean]; s.PET = xr[d.ch_pet];
                s.SP = st[0]; s.MW = st[1]; s.SM = st[2]; s.SUZ = st[3]; s.SLZ = st[4];
                s.template fwd<false, true>(p, d.nearzero, d.ac ? d.ac[b] : 0.f, d.elev ? d.elev[b] : 0.f,
                                            0.f, 0.f);     // the forward time-steppers' form of the soil stage
                if (o.traj)
                    for (int k = 0; k < 5; k++) o.traj[((int64_t)k * (T + 1) + t) * N + n] = st[k];
                if (o.aux) {
                    o.aux[((int64_t)0 * T + t) * N + n] = s.sw0;
                    o.aux[((int64_t)1 * T + t) * N + n] = s.ef0;
                }
                st[0] = s.SP3; st[1] = s.MW3; st[2] = s.SM4; st[3] = s.SUZ4; st[4] = s.SLZ2;
                if (o.flux) {
                    float *a = &acc[(size_t)t * o.n_flux];
                    float wq = d.muwts ? d.muwts[(int64_t)t * d.mu_t_stride + (int64_t)b * d.mu_b_stride + j] : 1.f;
                    a[HBVX_F_QSIM] += d.muwts ? s.Q * wq : s.Q;
                    a[HBVX_F_Q0] += s.Q0; a[HBVX_F_Q1] += s.Q1; a[HBVX_F_Q2] += s.Q2;
                    a[HBVX_F_AET] += s.ET; a[HBVX_F_SWE] += s.SP3; a[HBVX_F_RECHARGE] += s.rech;
                    a[HBVX_F_EXCS] += s.exc; a[HBVX_F_EVAPFACTOR] += s.ef; a[HBVX_F_TOSOIL] += s.tosoil;
                    a[HBVX_F_PERC] += s.PERC;
                    if (o.n_flux > HBVX_F_CAPILLARY) a[HBVX_F_CAPILLARY] += s.cap;
                }
            }
            for (int k = 0; k < 5; k++) {
                o.state_out[k * N + n] = st[k];
                if (o.traj) o.traj[((int64_t)k * (T + 1) + T) * N + n] = st[k];
            }
        }
        if (o.flux)
            for (int t = 0; t < T; t++)
                for (int k = 0; k < o.n_flux; k++) {
                    float v = acc[(size_t)t * o.n_flux + k];
                    if (!(k == HBVX_F_QSIM && d.muwts)) v = v * (1.0f / (float)M);
                    o.flux[((int64_t)k * T + t) * B + b] = v;
                }
    }
}

template <int MODEL, bool BETAET>
static void run_bwd(const hbvx_desc &d, const hbvx_bwd_io &io)
{
    const int T = d.T, B = d.B, M = d.M, NP = d.n_param, nf = io.n_flux;
    const int64_t N = (int64_t)B * M;
    const float invM = 1.0f / (float)M;
    const int64_t fs = (int64_t)T * B;
    std::vector<float> gxa;
    for (int b = 0; b < B; b++) {
        if (io.grad_x) gxa.assign((size_t)T * 3, 0.f);
        for (int j = 0; j < M; j++) {
            const int64_t n = (int64_t)b * M + j;
            float a[5], gsta[NPARAM_MAX] = {0}, usta[NPARAM_MAX] = {0};
            for (int k = 0; k < 5; k++) a[k] = io.grad_state_out ? io.grad_state_out[k * N + n] : 0.f;
            for (int t = T - 1; t >= 0; t--) {
                float p[NPARAM_MAX] = {0}, u[NPARAM_MAX] = {0};
                bool ud[NPARAM_MAX] = {false};
                for (int i = 0; i < NP; i++) {
                    const hbvx_param_src &s = d.p[i];
                    ud[i] = s.dyn && !(s.drop && s.drop[b]);
                    float v = ud[i] ? s.dyn[(int64_t)t * s.dyn_t_stride + (int64_t)b * s.dyn_b_stride + j]
                                    : s.sta[(int64_t)b * s.sta_b_stride + j];
                    u[i] = d.raw_sigmoid ? sigmoid_(v) : v;
                    p[i] = descale_(u[i], s.lo, s.hi);
                }
                Step<MODEL, BETAET> s;
                const float *xr = d.x + (int64_t)t * d.x_t_stride + (int64_t)b * d.x_b_stride;
                s.P = xr[d.ch_prcp]; s.Tf = xr[d.ch_tmean]; s.PET = xr[d.ch_pet];
                s.SP = io.traj[((int64_t)0 * (T + 1) + t) * N + n];
                s.MW = io.traj[((int64_t)1 * (T + 1) + t) * N + n];
                s.SM = io.traj[((int64_t)2 * (T + 1) + t) * N + n];
                s.SUZ = io.traj[((int64_t)3 * (T + 1) + t) * N + n];
                s.SLZ = io.traj[((int64_t)4 * (T + 1) + t) * N + n];
                // like the device adjoints: the two powers are recomputed unless the caller kept them
                if (io.aux)
                    s.template fwd<true>(p, d.nearzero, d.ac ? d.ac[b] : 0.f, d.elev ? d.elev[b] : 0.f,
                                         io.aux[((int64_t)0 * T + t) * N + n],
                                         io.aux[((int64_t)1 * T + t) * N + n]);
                else
                    s.template fwd<false>(p, d.nearzero, d.ac ? d.ac[b] : 0.f, d.elev ? d.elev[b] : 0.f, 0.f, 0.f);
                float gfv[HBVX_MAX_FLUX];
                for (int k = 0; k < nf; k++) {
                    gfv[k] = io.grad_flux ? io.grad_flux[(int64_t)k * fs + (int64_t)t * B + b] : 0.f;
                    if (io.grad_flux4 && k < 4) gfv[k] += io.grad_flux4[(int64_t)k * fs + (int64_t)t * B + b];
                }
                FluxGrad g;
                float wq = d.muwts ? d.muwts[(int64_t)t * d.mu_t_stride + (int64_t)b * d.mu_b_stride + j] : invM;
                g.gQ = gfv[HBVX_F_QSIM] * wq;
                g.gQ0 = gfv[HBVX_F_Q0] * invM; g.gQ1 = gfv[HBVX_F_Q1] * invM;
                g.gQ2 = gfv[HBVX_F_Q2] * invM; g.gET = gfv[HBVX_F_AET] * invM;
                g.gSWE = gfv[HBVX_F_SWE] * invM; g.grech = gfv[HBVX_F_RECHARGE] * invM;
                g.gexc = gfv[HBVX_F_EXCS] * invM; g.gef = gfv[HBVX_F_EVAPFACTOR] * invM;
                g.gtosoil = gfv[HBVX_F_TOSOIL] * invM; g.gPERC = gfv[HBVX_F_PERC] * invM;
                g.gcap = nf > HBVX_F_CAPILLARY ? gfv[HBVX_F_CAPILLARY] * invM : 0.f;
                if (io.grad_muwts) io.grad_muwts[((int64_t)t * B + b) * M + j] = gfv[HBVX_F_QSIM] * s.Q;
                float gp[NPARAM_MAX] = {0}, gx[3];
                s.bwd(p, d.nearzero, g, a, gp, gx);
                if (io.grad_x) { gxa[t * 3] += gx[0]; gxa[t * 3 + 1] += gx[1]; gxa[t * 3 + 2] += gx[2]; }
                for (int i = 0; i < NP; i++) {
                    float gu = gp[i] * (d.p[i].hi - d.p[i].lo);
                    if (d.p[i].dyn) {
                        float gr = d.raw_sigmoid ? gu * (u[i] * (1.0f - u[i])) : gu;
                        if (io.g[i].dyn)
                            io.g[i].dyn[(int64_t)t * io.g[i].dyn_t_stride + (int64_t)b * io.g[i].dyn_b_stride + j] =
                                ud[i] ? gr : 0.0f;
                        if (!ud[i]) { gsta[i] += gu; usta[i] = u[i]; }
                    } else {
                        gsta[i] += gu; usta[i] = u[i];
                    }
                }
            }
            for (int i = 0; i < NP; i++) {
                if (!io.g[i].sta) continue;
                float gr = d.raw_sigmoid ? gsta[i] * (usta[i] * (1.0f - usta[i])) : gsta[i];
                io.g[i].sta[(int64_t)b * io.g[i].sta_b_stride + j] += gr;
            }
            if (io.grad_state_in) for (int k = 0; k < 5; k++) io.grad_state_in[k * N + n] = a[k];
        }
        if (io.grad_x)
            for (int t = 0; t < T; t++) {
                float *gr = io.grad_x + (int64_t)t * d.x_t_stride + (int64_t)b * d.x_b_stride;
                gr[d.ch_prcp] = gxa[t * 3]; gr[d.ch_tmean] = gxa[t * 3 + 1]; gr[d.ch_pet] = gxa[t * 3 + 2];
            }
    }
}

#define DISPATCH(fn, ...)                                                                       \
    if (d->model == HBVX_MODEL_HBV10 && d->n_param == 12) fn<MODEL_HBV10, false>(__VA_ARGS__);   \
    else if (d->model == HBVX_MODEL_HBV10) fn<MODEL_HBV10, true>(__VA_ARGS__);                  \
    else if (d->model == HBVX_MODEL_HBV11P) fn<MODEL_HBV11P, true>(__VA_ARGS__);                \
    else if (d->model == HBVX_MODEL_HOURLY) fn<MODEL_HOURLY, true>(__VA_ARGS__);                \
    else fn<MODEL_HBV20, true>(__VA_ARGS__);

extern "C" int hbvx_forward(const hbvx_desc *d, const hbvx_fwd_out *o, void *)
{
    DISPATCH(run_fwd, *d, *o);
    return 0;
}
extern "C" uint64_t hbvx_backward_workspace_bytes(const hbvx_desc *) { return 0; }
extern "C" int hbvx_backward(const hbvx_desc *d, const hbvx_bwd_io *io, void *)
{
    DISPATCH(run_bwd, *d, *io);
    return 0;
}
// routing is not part of the step headers: forward it to the oracle library (HBVX_ORACLE_LIB)
static void *oracle_sym(const char *name)
{
    static void *h = nullptr;
    if (!h) {
        const char *path = getenv("HBVX_ORACLE_LIB");
        if (path) h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    }
    return h ? dlsym(h, name) : nullptr;
}
extern "C" int hbvx_route_forward(const hbvx_route_desc *r, const float *q, float *uh, float *qr, void *st)
{
    typedef int (*fn_t)(const hbvx_route_desc *, const float *, float *, float *, void *);
    fn_t fn = (fn_t)oracle_sym("hbvx_route_forward");
    if (!fn) { snprintf(g_err, sizeof g_err, "set HBVX_ORACLE_LIB for routing"); return HBVX_E_UNSUPPORTED; }
    return fn(r, q, uh, qr, st);
}
extern "C" uint64_t hbvx_route_workspace_bytes(const hbvx_route_desc *) { return 0; }
extern "C" int hbvx_route_backward(const hbvx_route_desc *r, const float *q, const float *uh,
                                   const float *gy, float *gq, float *ga, float *gb, void *ws,
                                   uint64_t wsb, void *st)
{
    typedef int (*fn_t)(const hbvx_route_desc *, const float *, const float *, const float *, float *,
                        float *, float *, void *, uint64_t, void *);
    fn_t fn = (fn_t)oracle_sym("hbvx_route_backward");
    if (!fn) { snprintf(g_err, sizeof g_err, "set HBVX_ORACLE_LIB for routing"); return HBVX_E_UNSUPPORTED; }
    return fn(r, q, uh, gy, gq, ga, gb, ws, wsb, st);
}

// implicit scheme through the product header hbv_adj_step.h
template <bool BETAET>
static void run_adj_fwd(const hbvx_desc &d, const hbvx_fwd_out &o)
{
    const int T = d.T, B = d.B, M = d.M, NP = d.n_param;
    const int64_t N = (int64_t)B * M;
    for (int b = 0; b < B; b++) {
        std::vector<float> acc((size_t)T, 0.f);
        for (int j = 0; j < M; j++) {
            const int64_t n = (int64_t)b * M + j;
            float x[5], xn[5], p[NPARAM_MAX] = {0};
            for (int k = 0; k < 5; k++) x[k] = d.state_in ? d.state_in[k * N + n] : 0.f;
            for (int t = 0; t < T; t++) {
                for (int i = 0; i < NP; i++) {
                    const hbvx_param_src &s = d.p[i];
                    bool ud = s.dyn && !(s.drop && s.drop[n]);
                    float v = ud ? s.dyn[(int64_t)t * s.dyn_t_stride + (int64_t)b * s.dyn_b_stride + j]
                                 : s.sta[(int64_t)b * s.sta_b_stride + j];
                    v = d.raw_sigmoid ? sigmoid_(v) : v;
                    p[i] = descale_(v, s.lo, s.hi);
                }
                AdjStep<BETAET> s;
                const float *xr = d.x + (int64_t)t * d.x_t_stride + (int64_t)b * d.x_b_stride;
                s.P = xr[d.ch_prcp]; s.Tf = xr[d.ch_tmean]; s.PET = xr[d.ch_pet];
                if (o.traj) for (int k = 0; k < 5; k++) o.traj[((int64_t)k * (T + 1) + t) * N + n] = x[k];
                if (d.adj_stop == 2) {
                    float Q;
                    AdjStaged<BETAET>::day(p, s.P, s.Tf, s.PET, x, d.adj_gtol, d.adj_max_iter, xn, Q);
                    for (int k = 0; k < 5; k++) x[k] = xn[k];
                    acc[t] += Q;
                    continue;
                }
                adj_newton<BETAET>(s, p, x, 1.0f, d.adj_gtol, d.adj_max_iter, xn);
                for (int k = 0; k < 5; k++) x[k] = xn[k];
                const float SUZ = fmaxf(x[3], 0.f), SLZ = fmaxf(x[4], 0.f);
                acc[t] += (p[P_K0] * fmaxf(SUZ - p[P_UZL], 0.f) + p[P_K1] * SUZ) + p[P_K2] * SLZ;
            }
            for (int k = 0; k < 5; k++) {
                o.state_out[k * N + n] = x[k];
                if (o.traj) o.traj[((int64_t)k * (T + 1) + T) * N + n] = x[k];
            }
        }
        if (o.flux) for (int t = 0; t < T; t++) o.flux[(int64_t)t * B + b] = acc[t] * (1.0f / (float)M);
    }
}

template <bool BETAET>
static void run_adj_bwd(const hbvx_desc &d, const hbvx_bwd_io &io)
{
    const int T = d.T, B = d.B, M = d.M, NP = d.n_param;
    const int64_t N = (int64_t)B * M;
    const float invM = 1.0f / (float)M;
    for (int b = 0; b < B; b++)
        for (int j = 0; j < M; j++) {
            const int64_t n = (int64_t)b * M + j;
            float a[5], gsta[NPARAM_MAX] = {0}, usta[NPARAM_MAX] = {0};
            for (int k = 0; k < 5; k++) a[k] = io.grad_state_out ? io.grad_state_out[k * N + n] : 0.f;
            for (int t = T - 1; t >= 0; t--) {
                float p[NPARAM_MAX] = {0}, u[NPARAM_MAX] = {0}, x[5], gp[NPARAM_MAX] = {0};
                bool ud[NPARAM_MAX] = {false};
                for (int i = 0; i < NP; i++) {
                    const hbvx_param_src &s = d.p[i];
                    ud[i] = s.dyn && !(s.drop && s.drop[n]);
                    float v = ud[i] ? s.dyn[(int64_t)t * s.dyn_t_stride + (int64_t)b * s.dyn_b_stride + j]
                                    : s.sta[(int64_t)b * s.sta_b_stride + j];
                    u[i] = d.raw_sigmoid ? sigmoid_(v) : v;
                    p[i] = descale_(u[i], s.lo, s.hi);
                }
                AdjStep<BETAET> s;
                const float *xr = d.x + (int64_t)t * d.x_t_stride + (int64_t)b * d.x_b_stride;
                s.P = xr[d.ch_prcp]; s.Tf = xr[d.ch_tmean]; s.PET = xr[d.ch_pet];
                for (int k = 0; k < 5; k++) x[k] = io.traj[((int64_t)k * (T + 1) + (t + 1)) * N + n];
                float gQ = io.grad_flux ? io.grad_flux[(int64_t)t * B + b] : 0.f;
                if (io.grad_flux4) gQ += io.grad_flux4[(int64_t)t * B + b];
                adj_backstep<BETAET>(s, p, x, 1.0f, gQ * invM, a, gp);
                for (int i = 0; i < NP; i++) {
                    float gu = gp[i] * (d.p[i].hi - d.p[i].lo);
                    if (d.p[i].dyn) {
                        float gr = d.raw_sigmoid ? gu * (u[i] * (1.0f - u[i])) : gu;
                        if (io.g[i].dyn)
                            io.g[i].dyn[(int64_t)t * io.g[i].dyn_t_stride + (int64_t)b * io.g[i].dyn_b_stride + j] =
                                ud[i] ? gr : 0.0f;
                        if (!ud[i]) { gsta[i] += gu; usta[i] = u[i]; }
                    } else {
                        gsta[i] += gu; usta[i] = u[i];
                    }
                }
            }
            for (int i = 0; i < NP; i++) {
                if (!io.g[i].sta) continue;
                float gr = d.raw_sigmoid ? gsta[i] * (usta[i] * (1.0f - usta[i])) : gsta[i];
                io.g[i].sta[(int64_t)b * io.g[i].sta_b_stride + j] += gr;
            }
            if (io.grad_state_in) for (int k = 0; k < 5; k++) io.grad_state_in[k * N + n] = a[k];
        }
}

extern "C" int hbvx_adj_forward(const hbvx_desc *d, const hbvx_fwd_out *o, void *)
{
    if (d->n_param == 13) run_adj_fwd<true>(*d, *o); else run_adj_fwd<false>(*d, *o);
    return 0;
}
extern "C" int hbvx_adj_backward(const hbvx_desc *d, const hbvx_bwd_io *io, void *)
{
    if (d->n_param == 13) run_adj_bwd<true>(*d, *io); else run_adj_bwd<false>(*d, *io);
    return 0;
}

extern "C" int hbvx_bfi(int32_t T, int32_t B, const float *qs, const float *q2, float nz, float *bfi, void *st)
{
    typedef int (*fn_t)(int32_t, int32_t, const float *, const float *, float, float *, void *);
    fn_t fn = (fn_t)oracle_sym("hbvx_bfi");
    if (!fn) { snprintf(g_err, sizeof g_err, "set HBVX_ORACLE_LIB for bfi"); return HBVX_E_UNSUPPORTED; }
    return fn(T, B, qs, q2, nz, bfi, st);
}

extern "C" uint64_t hbvx_gage_route_workspace_bytes(const hbvx_gage_desc *) { return 0; }
extern "C" int hbvx_gage_route_forward(const hbvx_gage_desc *r, const float *qs, float *uh, float *out, void *ws,
                                       uint64_t wsb, void *st)
{
    typedef int (*fn_t)(const hbvx_gage_desc *, const float *, float *, float *, void *, uint64_t, void *);
    fn_t fn = (fn_t)oracle_sym("hbvx_gage_route_forward");
    if (!fn) { snprintf(g_err, sizeof g_err, "set HBVX_ORACLE_LIB for gage routing"); return HBVX_E_UNSUPPORTED; }
    return fn(r, qs, uh, out, ws, wsb, st);
}
extern "C" int hbvx_gage_route_backward(const hbvx_gage_desc *r, const float *qs, const float *uh,
                                        const float *go, float *gqs, float *gdp, void *ws, uint64_t wsb, void *st)
{
    typedef int (*fn_t)(const hbvx_gage_desc *, const float *, const float *, const float *, float *, float *, void *,
                        uint64_t, void *);
    fn_t fn = (fn_t)oracle_sym("hbvx_gage_route_backward");
    if (!fn) { snprintf(g_err, sizeof g_err, "set HBVX_ORACLE_LIB for gage routing"); return HBVX_E_UNSUPPORTED; }
    return fn(r, qs, uh, go, gqs, gdp, ws, wsb, st);
}

// sequence LSTM (include/hbvx_lstm.h): forwarded to the oracle, like the routing entry points
extern "C" uint64_t hbvx_lstm_workspace_bytes(const hbvx_lstm_desc *) { return 0; }
extern "C" int hbvx_lstm_forward(const hbvx_lstm_desc *d, const float *w, const float *gx, float *g, float *c,
                                 float *h, void *ws, uint64_t wsb, void *st)
{
    typedef int (*fn_t)(const hbvx_lstm_desc *, const float *, const float *, float *, float *, float *, void *,
                        uint64_t, void *);
    fn_t fn = (fn_t)oracle_sym("hbvx_lstm_forward");
    if (!fn) { snprintf(g_err, sizeof g_err, "set HBVX_ORACLE_LIB for the lstm"); return HBVX_E_UNSUPPORTED; }
    return fn(d, w, gx, g, c, h, ws, wsb, st);
}
extern "C" int hbvx_lstm_backward(const hbvx_lstm_desc *d, const float *w, const float *g, const float *c,
                                  const float *gh, float *gg, void *ws, uint64_t wsb, void *st)
{
    typedef int (*fn_t)(const hbvx_lstm_desc *, const float *, const float *, const float *, const float *, float *,
                        void *, uint64_t, void *);
    fn_t fn = (fn_t)oracle_sym("hbvx_lstm_backward");
    if (!fn) { snprintf(g_err, sizeof g_err, "set HBVX_ORACLE_LIB for the lstm"); return HBVX_E_UNSUPPORTED; }
    return fn(d, w, g, c, gh, gg, ws, wsb, st);
}
extern "C" int hbvx_lstm_check(const hbvx_lstm_desc *, const void *, void *) { return 0; }
extern "C" int hbvx_zero(void *ptr, uint64_t bytes, void *) { if (bytes) memset(ptr, 0, (size_t)bytes); return 0; }
extern "C" int hbvx_zero_except(float *ptr, int64_t rows, int32_t width, int64_t r0, int64_t r1, int32_t group_w,
                                uint32_t keep, void *)
{
    for (int64_t r = 0; r < rows; r++)
        for (int32_t c = 0; c < width; c++) {
            const int32_t g = c / group_w;
            if (!(r >= r0 && r < r1 && g < 32 && ((keep >> g) & 1u))) ptr[r * width + c] = 0.0f;
        }
    return 0;
}

// Step::jt_unit against Step::bwd with zero flux adjoints (HBV 1.0): for n random days, the worst
// difference over the five unit adjoints, relative to the largest entry of J^T.
template <bool BETAET>
static float jt_check(const float *st, const float *f, const float *p, int n, float nz)
{
    float worst = 0.0f;
    for (int i = 0; i < n; i++) {
        Step<MODEL_HBV10, BETAET> s;
        s.SP = st[i * 5]; s.MW = st[i * 5 + 1]; s.SM = st[i * 5 + 2]; s.SUZ = st[i * 5 + 3]; s.SLZ = st[i * 5 + 4];
        s.P = f[i * 3]; s.Tf = f[i * 3 + 1]; s.PET = f[i * 3 + 2];
        const float *pp = p + i * NPARAM_MAX;
        s.template fwd<false>(pp, nz, 0.0f, 0.0f, 0.0f, 0.0f);
        FluxGrad g0;
        memset(&g0, 0, sizeof g0);
        const auto c = s.jt_coef(pp, nz);
        float scale = 1e-30f, err = 0.0f;
        for (int k = 0; k < 5; k++) {
            float a[5] = {0, 0, 0, 0, 0}, b[5] = {0, 0, 0, 0, 0}, gp[NPARAM_MAX] = {0}, gx[3];
            a[k] = b[k] = 1.0f;
            s.bwd(pp, nz, g0, a, gp, gx);
            if (k < 2) Step<MODEL_HBV10, BETAET>::template jt_unit<0>(c, b);
            else if (k == 2) Step<MODEL_HBV10, BETAET>::template jt_unit<1>(c, b);
            else Step<MODEL_HBV10, BETAET>::template jt_unit<2>(c, b);
            for (int j = 0; j < 5; j++) {
                scale = fmaxf(scale, fabsf(a[j]));
                err = fmaxf(err, fabsf(a[j] - b[j]));
            }
        }
        // affine form: random incoming adjoint + gradients on the runoff series only
        {
            float a[5], b[5], gp[NPARAM_MAX] = {0}, gx[3];
            for (int j = 0; j < 5; j++) a[j] = b[j] = 0.3f * (float)(((i * 7 + j * 13) % 11) - 5);
            FluxGrad g = g0;
            g.gQ = 0.01f * (float)((i % 7) - 3); g.gQ0 = 0.02f * (float)((i % 5) - 2);
            g.gQ1 = -0.015f * (float)((i % 3) - 1); g.gQ2 = 0.005f * (float)((i % 9) - 4);
            s.bwd(pp, nz, g, a, gp, gx);
            Step<MODEL_HBV10, BETAET>::jt_affine(c, b, g.gQ0 + g.gQ, g.gQ1 + g.gQ, g.gQ2 + g.gQ);
            for (int j = 0; j < 5; j++) {
                scale = fmaxf(scale, fabsf(a[j]));
                err = fmaxf(err, fabsf(a[j] - b[j]));
            }
        }
        worst = fmaxf(worst, err / scale);
    }
    return worst;
}
// AdjStaged::day against the residual it claims to zero: G(x) = (x - xt)/dt - f(x) with f from AdjStep::eval
// (the restatement of hbv_adj.py:385-431 the joint iteration uses).  out[k] = worst |G_k| over the n days,
// relative to max(1, |xt_k|, |x_k|); out[5] = the largest soil-moisture update count.
extern "C" void hbvx_test_staged_residual(const float *st, const float *f, const float *p, int n, float gtol,
                                          int max_iter, float *out, float *g2_each)
{
    for (int k = 0; k < 6; k++) out[k] = 0.0f;
    for (int i = 0; i < n; i++) {
        const float *xt = st + i * 5, *pp = p + i * NPARAM_MAX;
        float x[5], Q;
        const int it = AdjStaged<true>::day(pp, f[i * 3], f[i * 3 + 1], f[i * 3 + 2], xt, gtol, max_iter, x, Q);
        AdjStep<true> s;
        s.P = f[i * 3]; s.Tf = f[i * 3 + 1]; s.PET = f[i * 3 + 2];
        s.template eval<false>(x, pp);
        for (int k = 0; k < 5; k++) {
            const float g = (x[k] - xt[k]) - s.f[k];
            const float sc = fmaxf(1.0f, fmaxf(fabsf(xt[k]), fabsf(x[k])));
            out[k] = fmaxf(out[k], fabsf(g) / sc);
        }
        out[5] = fmaxf(out[5], (float)it);
        if (g2_each) { g2_each[2 * i] = (x[2] - xt[2]) - s.f[2]; g2_each[2 * i + 1] = x[2]; }
        const float Qe = (s.q0 + s.q1) + s.q2;
        if (fabsf(Q - Qe) > 1e-6f * fmaxf(1.0f, fabsf(Qe))) out[5] = 1e9f;   // Q must be the flux at the solved state
    }
}
// Diagnostics: how many soil-moisture updates the staged solve takes on a synthetic record.  N lanes (static
// parameters p [N][NPARAM_MAX], forcing f [T][N/M][3]), T days from zero storages; hist[k] counts WAVE-days
// (64 consecutive lanes) whose slowest lane took k updates, hist[16 + k] counts lane-days.
extern "C" void hbvx_test_staged_iters(const float *f, const float *p, int T, int N, int M, float gtol,
                                       int max_iter, long long *hist)
{
    for (int k = 0; k < 32; k++) hist[k] = 0;
    std::vector<float> x((size_t)N * 5, 0.0f);
    std::vector<int> wmax((size_t)(N + 63) / 64);
    for (int t = 0; t < T; t++) {
        std::fill(wmax.begin(), wmax.end(), 0);
        for (int n = 0; n < N; n++) {
            const float *ff = f + ((size_t)t * (N / M) + n / M) * 3;
            float xn[5], Q;
            const int it = AdjStaged<true>::day(p + (size_t)n * NPARAM_MAX, ff[0], ff[1], ff[2], &x[(size_t)n * 5],
                                                gtol, max_iter, xn, Q);
            for (int k = 0; k < 5; k++) x[(size_t)n * 5 + k] = xn[k];
            hist[16 + (it < 15 ? it : 15)]++;
            wmax[n / 64] = it > wmax[n / 64] ? it : wmax[n / 64];
        }
        for (int w : wmax) hist[w < 15 ? w : 15]++;
    }
}
// snow rows of the capillary models: jt_coef_snow + jt_unit<0> against bwd (HBV 2.0 instantiation)
extern "C" float hbvx_test_jt_snow(const float *st, const float *f, const float *p, int n, float nz)
{
    float worst = 0.0f;
    for (int i = 0; i < n; i++) {
        Step<MODEL_HBV20, true> s;
        s.SP = st[i * 5]; s.MW = st[i * 5 + 1]; s.SM = st[i * 5 + 2]; s.SUZ = st[i * 5 + 3]; s.SLZ = st[i * 5 + 4];
        s.P = f[i * 3]; s.Tf = f[i * 3 + 1]; s.PET = f[i * 3 + 2];
        const float *pp = p + i * NPARAM_MAX;
        s.template fwd<false>(pp, nz, 1500.0f, (i & 1) ? 2500.0f : 500.0f, 0.0f, 0.0f);
        FluxGrad g0;
        memset(&g0, 0, sizeof g0);
        const auto c = s.jt_coef_snow(pp);
        for (int k = 0; k < 2; k++) {
            float a[5] = {0, 0, 0, 0, 0}, b[5] = {0, 0, 0, 0, 0}, gp[NPARAM_MAX] = {0}, gx[3];
            a[k] = b[k] = 1.0f;
            s.bwd(pp, nz, g0, a, gp, gx);
            Step<MODEL_HBV20, true>::template jt_unit<0>(c, b);
            for (int j = 0; j < 5; j++) worst = fmaxf(worst, fabsf(a[j] - b[j]));
        }
    }
    return worst;
}
// capillary models: jt_coef_cap + jt_cap against bwd for the three coupled rows and for the affine form
// with gradients on the runoff series; worst difference relative to the largest entry of the day's J^T
template <int MODEL>
static float jt_cap_check(const float *st, const float *f, const float *p, int n, float nz)
{
    float worst = 0.0f;
    for (int i = 0; i < n; i++) {
        Step<MODEL, true> s;
        s.SP = st[i * 5]; s.MW = st[i * 5 + 1]; s.SM = st[i * 5 + 2]; s.SUZ = st[i * 5 + 3]; s.SLZ = st[i * 5 + 4];
        s.P = f[i * 3]; s.Tf = f[i * 3 + 1]; s.PET = f[i * 3 + 2];
        const float *pp = p + i * NPARAM_MAX;
        // drainage areas on both sides of parAC / 2500 km2 and elevations on both sides of 2000 m
        s.template fwd<false>(pp, nz, 40.0f + 3000.0f * (float)((i * 37) % 100) / 100.0f, (i & 1) ? 2500.0f : 500.0f, 0.0f, 0.0f);
        FluxGrad g0;
        memset(&g0, 0, sizeof g0);
        const auto c = s.jt_coef_cap(pp, nz);
        float scale = 1e-30f, err = 0.0f;
        for (int k = 0; k < 5; k++) {
            float a[5] = {0, 0, 0, 0, 0}, b[5] = {0, 0, 0, 0, 0}, gp[NPARAM_MAX] = {0}, gx[3];
            a[k] = b[k] = 1.0f;
            s.bwd(pp, nz, g0, a, gp, gx);
            if (k < 2) Step<MODEL, true>::template jt_unit<0>(c, b);
            else Step<MODEL, true>::jt_cap(c, b, 0.0f, 0.0f, 0.0f);
            for (int j = 0; j < 5; j++) {
                scale = fmaxf(scale, fabsf(a[j]));
                err = fmaxf(err, fabsf(a[j] - b[j]));
            }
        }
        {
            float a[5], b[5], gp[NPARAM_MAX] = {0}, gx[3];
            for (int j = 0; j < 5; j++) a[j] = b[j] = 0.3f * (float)(((i * 7 + j * 13) % 11) - 5);
            FluxGrad g = g0;
            g.gQ = 0.01f * (float)((i % 7) - 3); g.gQ0 = 0.02f * (float)((i % 5) - 2);
            g.gQ1 = -0.015f * (float)((i % 3) - 1); g.gQ2 = 0.005f * (float)((i % 9) - 4);
            s.bwd(pp, nz, g, a, gp, gx);
            Step<MODEL, true>::jt_cap(c, b, g.gQ0 + g.gQ, g.gQ1 + g.gQ, g.gQ2 + g.gQ);
            for (int j = 0; j < 5; j++) {
                scale = fmaxf(scale, fabsf(a[j]));
                err = fmaxf(err, fabsf(a[j] - b[j]));
            }
        }
        worst = fmaxf(worst, err / scale);
    }
    return worst;
}
extern "C" float hbvx_test_jt_cap(const float *st, const float *f, const float *p, int n, float nz, int model)
{
    return model == 2 ? jt_cap_check<MODEL_HBV20>(st, f, p, n, nz) : jt_cap_check<MODEL_HBV11P>(st, f, p, n, nz);
}
extern "C" float hbvx_test_jt(const float *st, const float *f, const float *p, int n, float nz, int betaet)
{
    return betaet ? jt_check<true>(st, f, p, n, nz) : jt_check<false>(st, f, p, n, nz);
}

// accuracy probe for hbvx::pow_pos_ (host build of the same source)
extern "C" void hbvx_test_pow(const float *x, const float *y, float *out, int n)
{
    for (int i = 0; i < n; i++) out[i] = pow_pos_(x[i], y[i]);
}
extern "C" void hbvx_test_pow_f64(const float *x, const float *y, float *out, int n)
{
    for (int i = 0; i < n; i++) out[i] = pow_f64_(x[i], y[i]);
}
