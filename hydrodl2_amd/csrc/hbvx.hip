// hbvx.hip -- libhbvx.so: HIP implementation of include/hbvx.h for gfx950 (MI355X).
//
// Lane mapping (all kernels): one wavefront lane per (basin, ensemble member).
// A 64-lane wave holds 64/Mp basins x Mp members, Mp = next power of two >= M,
// so one basin's ensemble sits in Mp adjacent lanes and the ensemble mean
// (reference hbv.py:507-511) is an intra-wave butterfly, never memory traffic.
// The time axis is strictly serial per lane (explicit Euler recurrence).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
#include "hbvx_host.h"
#include "hbv_step.h"
#include "hbv_gage.h"

using namespace hbvx;
using namespace hbvx_host;

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int hbvx_host::fail(int code, const char *msg)
{
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

int hbvx_host::hip_fail(hipError_t e, const char *what)
{
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return HBVX_E_DEVICE;
}

// which kernel family took the last forward / adjoint call of the process (diagnostic: the tests assert that the
// family they mean to check is the one that ran).  The adjoint runs on autograd's thread, the asking test on the
// main one: process-wide under a mutex, handed out through a per-thread copy.
static std::mutex g_disp_mu;
static char g_disp[2][48] = {"", ""};
void hbvx_host::note_dispatch(int dir, const char *family)
{
    std::lock_guard<std::mutex> lock(g_disp_mu);
    snprintf(g_disp[dir & 1], sizeof g_disp[0], "%s", family);
}
extern "C" int hbvx_zero_in_launch(void) { return hbvx_host::zero_taken() ? 1 : 0; }

extern "C" const char *hbvx_last_dispatch(int direction)
{
    static thread_local char out[48];
    std::lock_guard<std::mutex> lock(g_disp_mu);
    snprintf(out, sizeof out, "%s", g_disp[direction & 1]);
    return out;
}

extern "C" int hbvx_version(void) { return HBVX_ABI_VERSION; }
extern "C" const char *hbvx_last_error(void) { return g_err; }
extern "C" const char *hbvx_backend(void) { return hbvx::SAVE_POW ? "hip:gfx950+savepow" : "hip:gfx950"; }
extern "C" uint64_t hbvx_sizeof(int which)
{
    switch (which) {
    case 0: return sizeof(hbvx_desc);
    case 1: return sizeof(hbvx_fwd_out);
    case 2: return sizeof(hbvx_bwd_io);
    case 3: return sizeof(hbvx_route_desc);
    case 6: return sizeof(hbvx_gage_desc);
    case 4: return sizeof(hbvx_param_src);
    case 5: return sizeof(hbvx_param_grad);
    default: return 0;
    }
}

// ---------------------------------------------------------------------------
// shared device helpers
// ---------------------------------------------------------------------------
template <int MODEL, bool BETAET>
struct NParam {
    static constexpr int value = MODEL == MODEL_HBV10 ? (BETAET ? 13 : 12)
                               : MODEL == MODEL_HBV11P ? 14 : (MODEL == MODEL_HOURLY ? 19 : 16);
};

struct LaneId {
    int jm, b, j;   // padded member index, basin (clamped), member (clamped)
    bool active;    // lane maps to a real (basin, member)
    bool leader;    // first lane of a real basin
    int64_t n;      // b*M + j
};

__device__ __forceinline__ LaneId lane_id(const hbvx_desc &d, int lgMp)
{
    LaneId L;
    const int lane = threadIdx.x & 63;
    const int Mp = 1 << lgMp;
    L.jm = lane & (Mp - 1);
    int b = blockIdx.x * (64 >> lgMp) + (lane >> lgMp);
    L.active = (b < d.B) && (L.jm < d.M);
    L.leader = (b < d.B) && (L.jm == 0);
    L.b = b < d.B ? b : d.B - 1;
    L.j = L.jm < d.M ? L.jm : d.M - 1;
    L.n = (int64_t)L.b * d.M + L.j;
    return L;
}

// sum over the Mp lanes of one basin (xor butterfly; every lane gets the sum)
__device__ __forceinline__ float ens_sum(float v, int lgMp)
{
    for (int s = 0; s < lgMp; s++) v += __shfl_xor(v, 1 << s, 64);
    return v;
}

// ---------------------------------------------------------------------------
// forward recurrence (reference hbv.py:423-511 + parameter prep hbv.py:201-256)
// ---------------------------------------------------------------------------
struct FwdArgs {
    hbvx_desc d;
    hbvx_fwd_out o;
    int lgMp;
};

template <int MODEL, bool BETAET>
__global__ void __launch_bounds__(64) k_fwd(const FwdArgs A)
{
    constexpr int NP = NParam<MODEL, BETAET>::value;
    const hbvx_desc &d = A.d;
    const hbvx_fwd_out &o = A.o;
    const int lgMp = A.lgMp;
    const LaneId L = lane_id(d, lgMp);
    const int T = d.T;
    const int64_t N = (int64_t)d.B * d.M;
    const bool raw = d.raw_sigmoid != 0;
    const float nz = d.nearzero;
    const float ac = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.ac[L.b] : 0.0f;
    const float elev = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.elev[L.b] : 0.0f;

    float p[NPARAM_MAX];
    const float *dynp[NP];
    bool use_dyn[NP];
    unsigned dmask = 0;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const hbvx_param_src &s = d.p[i];
        float v = s.sta[(int64_t)L.b * s.sta_b_stride + L.j];
        v = raw ? sigmoid_(v) : v;
        p[i] = descale_(v, s.lo, s.hi);
        dynp[i] = s.dyn ? s.dyn + (int64_t)L.b * s.dyn_b_stride + L.j : s.sta;
        use_dyn[i] = s.dyn && !(s.drop && s.drop[L.b]);
        if (s.dyn) dmask |= 1u << i;
    }
#pragma unroll
    for (int i = NP; i < NPARAM_MAX; i++) p[i] = 0.0f;

    float st[5];
#pragma unroll
    for (int k = 0; k < 5; k++) st[k] = d.state_in ? d.state_in[k * N + L.n] : 0.001f;

    const float *xb = d.x + (int64_t)L.b * d.x_b_stride;
    const float *mu = d.muwts ? d.muwts + (int64_t)L.b * d.mu_b_stride + L.j : nullptr;
    const float invM = 1.0f / (float)d.M;
    const int nf = o.n_flux;
    // HBVX_TRAJ_CKPT: only the storages entering every K-th day are kept, [ceil(T/K), 5, N]
    const int ckpt_k = (o.traj && HBVX_TRAJ_KIND(o.traj_layout) == HBVX_TRAJ_CKPT) ? HBVX_TRAJ_CKPT_DAYS(o.traj_layout) : 0;

    // one-step-ahead register prefetch of the per-step inputs
    float nxf[3], nxd[NP];
    {
        const float *xr = xb;
        nxf[0] = T > 0 ? xr[d.ch_prcp] : 0.f;
        nxf[1] = T > 0 ? xr[d.ch_tmean] : 0.f;
        nxf[2] = T > 0 ? xr[d.ch_pet] : 0.f;
#pragma unroll
        for (int i = 0; i < NP; i++) nxd[i] = ((dmask >> i) & 1) && T > 0 ? dynp[i][0] : 0.f;
    }

    for (int t = 0; t < T; t++) {
        Step<MODEL, BETAET> s;
        s.P = nxf[0]; s.Tf = nxf[1]; s.PET = nxf[2];
        float cur[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) cur[i] = nxd[i];
        {
            const int tn = t + 1 < T ? t + 1 : t;
            const float *xr = xb + (int64_t)tn * d.x_t_stride;
            nxf[0] = xr[d.ch_prcp]; nxf[1] = xr[d.ch_tmean]; nxf[2] = xr[d.ch_pet];
#pragma unroll
            for (int i = 0; i < NP; i++)
                if ((dmask >> i) & 1) nxd[i] = dynp[i][(int64_t)tn * d.p[i].dyn_t_stride];
        }
#pragma unroll
        for (int i = 0; i < NP; i++)
            if ((dmask >> i) & 1) {
                float v = raw ? sigmoid_dyn_(cur[i]) : cur[i];
                float pv = descale_(v, d.p[i].lo, d.p[i].hi);
                if (use_dyn[i]) p[i] = pv;
            }
        s.SP = st[0]; s.MW = st[1]; s.SM = st[2]; s.SUZ = st[3]; s.SLZ = st[4];
        s.template fwd<false, true>(p, nz, ac, elev, 0.f, 0.f);

        if (ckpt_k) {
            if (t % ckpt_k == 0 && L.active) {
#pragma unroll
                for (int k = 0; k < 5; k++) o.traj[((int64_t)(t / ckpt_k) * 5 + k) * N + L.n] = st[k];
            }
        } else if (o.traj && L.active) {
#pragma unroll
            for (int k = 0; k < 5; k++) o.traj[((int64_t)k * (T + 1) + t) * N + L.n] = st[k];
        }
        if (SAVE_POW && o.aux && L.active) {
            o.aux[((int64_t)0 * T + t) * N + L.n] = s.sw0;
            o.aux[((int64_t)1 * T + t) * N + L.n] = s.ef0;
        }
        st[0] = s.SP3; st[1] = s.MW3; st[2] = s.SM4; st[3] = s.SUZ4; st[4] = s.SLZ2;

        if (o.flux) {
            const float act = L.active ? 1.0f : 0.0f;
            float wq = mu ? mu[(int64_t)t * d.mu_t_stride] : 1.0f;
            float f[HBVX_MAX_FLUX];
            f[HBVX_F_QSIM] = (mu ? s.Q * wq : s.Q) * act;
            f[HBVX_F_Q0] = s.Q0 * act;
            f[HBVX_F_Q1] = s.Q1 * act;
            f[HBVX_F_Q2] = s.Q2 * act;
            f[HBVX_F_AET] = s.ET * act;
            f[HBVX_F_SWE] = s.SP3 * act;
            f[HBVX_F_RECHARGE] = s.rech * act;
            f[HBVX_F_EXCS] = s.exc * act;
            f[HBVX_F_EVAPFACTOR] = s.ef * act;
            f[HBVX_F_TOSOIL] = s.tosoil * act;
            f[HBVX_F_PERC] = s.PERC * act;
            f[HBVX_F_CAPILLARY] = s.cap * act;
#pragma unroll
            for (int k = 0; k < HBVX_MAX_FLUX; k++) {
                if (k < nf) {
                    float v = ens_sum(f[k], lgMp);
                    if (!(k == HBVX_F_QSIM && mu)) v = v * invM;
                    if (L.leader) o.flux[((int64_t)k * T + t) * d.B + L.b] = v;
                }
            }
        }
    }
    if (L.active) {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            o.state_out[k * N + L.n] = st[k];
            if (o.traj && !ckpt_k) o.traj[((int64_t)k * (T + 1) + T) * N + L.n] = st[k];
        }
    }
}

// ---------------------------------------------------------------------------
// adjoint recurrence (the autograd tape of the same lines)
// ---------------------------------------------------------------------------
struct BwdArgs {
    hbvx_desc d;
    hbvx_bwd_io io;
    int lgMp;
};

template <int MODEL, bool BETAET>
__global__ void __launch_bounds__(64) k_bwd(const BwdArgs A)
{
    constexpr int NP = NParam<MODEL, BETAET>::value;
    const hbvx_desc &d = A.d;
    const hbvx_bwd_io &io = A.io;
    const int lgMp = A.lgMp;
    const LaneId L = lane_id(d, lgMp);
    const int T = d.T;
    const int64_t N = (int64_t)d.B * d.M;
    const bool raw = d.raw_sigmoid != 0;
    const float nz = d.nearzero;
    const float ac = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.ac[L.b] : 0.0f;
    const float elev = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.elev[L.b] : 0.0f;
    const int nf = io.n_flux;

    float p[NPARAM_MAX], usta[NP], gsta[NP];
    const float *dynp[NP];
    float *gdyn[NP];
    bool use_dyn[NP];
    unsigned dmask = 0;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const hbvx_param_src &s = d.p[i];
        float v = s.sta[(int64_t)L.b * s.sta_b_stride + L.j];
        usta[i] = raw ? sigmoid_(v) : v;
        p[i] = descale_(usta[i], s.lo, s.hi);
        dynp[i] = s.dyn ? s.dyn + (int64_t)L.b * s.dyn_b_stride + L.j : s.sta;
        gdyn[i] = io.g[i].dyn ? io.g[i].dyn + (int64_t)L.b * io.g[i].dyn_b_stride + L.j : nullptr;
        use_dyn[i] = s.dyn && !(s.drop && s.drop[L.b]);
        gsta[i] = 0.0f;
        if (s.dyn) dmask |= 1u << i;
    }
#pragma unroll
    for (int i = NP; i < NPARAM_MAX; i++) p[i] = 0.0f;

    const float *xb = d.x + (int64_t)L.b * d.x_b_stride;
    const float *mu = d.muwts ? d.muwts + (int64_t)L.b * d.mu_b_stride + L.j : nullptr;
    const float invM = 1.0f / (float)d.M;
    float a[5];
#pragma unroll
    for (int k = 0; k < 5; k++) a[k] = io.grad_state_out ? io.grad_state_out[k * N + L.n] : 0.0f;

    for (int t = T - 1; t >= 0; t--) {
        Step<MODEL, BETAET> s;
        const float *xr = xb + (int64_t)t * d.x_t_stride;
        s.P = xr[d.ch_prcp]; s.Tf = xr[d.ch_tmean]; s.PET = xr[d.ch_pet];
        s.SP = io.traj[((int64_t)0 * (T + 1) + t) * N + L.n];
        s.MW = io.traj[((int64_t)1 * (T + 1) + t) * N + L.n];
        s.SM = io.traj[((int64_t)2 * (T + 1) + t) * N + L.n];
        s.SUZ = io.traj[((int64_t)3 * (T + 1) + t) * N + L.n];
        s.SLZ = io.traj[((int64_t)4 * (T + 1) + t) * N + L.n];
        const float sw0 = SAVE_POW ? io.aux[((int64_t)0 * T + t) * N + L.n] : 0.0f;
        const float ef0 = SAVE_POW ? io.aux[((int64_t)1 * T + t) * N + L.n] : 0.0f;
        float ud[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            ud[i] = usta[i];
            if ((dmask >> i) & 1) {
                float v = dynp[i][(int64_t)t * d.p[i].dyn_t_stride];
                v = raw ? sigmoid_dyn_(v) : v;
                if (use_dyn[i]) {
                    ud[i] = v;
                    p[i] = descale_(v, d.p[i].lo, d.p[i].hi);
                }
            }
        }
        s.template fwd<SAVE_POW>(p, nz, ac, elev, sw0, ef0);

        FluxGrad g;
        const int64_t fs = (int64_t)T * d.B;
        const int64_t go = (int64_t)t * d.B + L.b;
        auto GF = [&](int k) -> float {
            float v = io.grad_flux ? io.grad_flux[k * fs + go] : 0.0f;
            if (io.grad_flux4 && k < 4) v += io.grad_flux4[k * fs + go];
            return v;
        };
        const float gq = GF(HBVX_F_QSIM);
        const float wq = mu ? mu[(int64_t)t * d.mu_t_stride] : invM;
        g.gQ = gq * wq;
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
        if (io.grad_muwts && L.active) io.grad_muwts[((int64_t)t * d.B + L.b) * d.M + L.j] = gq * s.Q;

        float gp[NPARAM_MAX], gx[3];
#pragma unroll
        for (int i = 0; i < NPARAM_MAX; i++) gp[i] = 0.0f;
        s.bwd(p, nz, g, a, gp, gx);

#pragma unroll
        for (int i = 0; i < NP; i++) {
            float gu = gp[i] * (d.p[i].hi - d.p[i].lo);
            if ((dmask >> i) & 1) {
                float gr = raw ? gu * (ud[i] * (1.0f - ud[i])) : gu;
                if (gdyn[i] && L.active)
                    gdyn[i][(int64_t)t * io.g[i].dyn_t_stride] = use_dyn[i] ? gr : 0.0f;
                gsta[i] += use_dyn[i] ? 0.0f : gu;
            } else {
                gsta[i] += gu;
            }
        }
        if (io.grad_x) {
            const float act = L.active ? 1.0f : 0.0f;
            float g0 = ens_sum(gx[0] * act, lgMp), g1 = ens_sum(gx[1] * act, lgMp),
                  g2 = ens_sum(gx[2] * act, lgMp);
            if (L.leader) {
                float *gr = io.grad_x + (int64_t)t * d.x_t_stride + (int64_t)L.b * d.x_b_stride;
                gr[d.ch_prcp] = g0; gr[d.ch_tmean] = g1; gr[d.ch_pet] = g2;
            }
        }
    }
    if (L.active) {
#pragma unroll
        for (int i = 0; i < NP; i++) {
            if (!io.g[i].sta) continue;
            float gr = raw ? gsta[i] * (usta[i] * (1.0f - usta[i])) : gsta[i];
            float *dst = io.g[i].sta + (int64_t)L.b * io.g[i].sta_b_stride + L.j;
            *dst += gr;
        }
        if (io.grad_state_in) {
#pragma unroll
            for (int k = 0; k < 5; k++) io.grad_state_in[k * N + L.n] = a[k];
        }
    }
}

// ---------------------------------------------------------------------------
// unit-hydrograph routing (reference core/calc/uh_routing.py:5-57)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void route_ab(const hbvx_route_desc &r, int b, float &ua, float &ub,
                                         float &a, float &bb)
{
    float va = r.ra[(int64_t)b * r.r_stride], vb = r.rb[(int64_t)b * r.r_stride];
    ua = r.raw_sigmoid ? sigmoid_(va) : va;
    ub = r.raw_sigmoid ? sigmoid_(vb) : vb;
    a = descale_(ua, r.a_lo, r.a_hi);
    bb = descale_(ub, r.b_lo, r.b_hi);
}

// uh_gamma (uh_routing.py:5-22): one thread per basin
__global__ void k_uh_gamma(const hbvx_route_desc r, float *__restrict__ uh)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= r.B) return;
    float ua, ub, a, bb;
    route_ab(r, b, ua, ub, a, bb);
    float aa = fmaxf(a, 0.0f) + 0.1f;
    float theta = fmaxf(bb, 0.0f) + 0.5f;
    float denom = expf(lgammaf(aa)) * powf(theta, aa);
    float w[HBVX_UH_MAXLEN];
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < HBVX_UH_MAXLEN; k++) {
        float t = (float)k + 0.5f;
        float mid = powf(t, aa - 1.0f);
        float right = expf(-t / theta);
        w[k] = (k < r.L) ? 1.0f / denom * mid * right : 0.0f;
        sum += w[k];
    }
#pragma unroll
    for (int k = 0; k < HBVX_UH_MAXLEN; k++)
        if (k < r.L) uh[(int64_t)b * r.L + k] = w[k] / sum;
}

// Routing kernels: thread = (basin, chunk of ROUTE_CHUNK days); the 64 lanes of a wave are 64
// consecutive basins (256-byte coalesced accesses), the 15-tap window slides through registers,
// so every input element is read once (plus a 14-day halo per chunk).
#ifndef ROUTE_CHUNK
#define ROUTE_CHUNK 32
#endif
// The adjoint's chunk: its tap-gradient partials are per chunk, so longer chunks halve the second stage's input; measured at
// the headline shape (round 5, whole-library variants): forward 32 / 16 / 64 days: 81 / 78 / 82 us; adjoint: 68 / 84 / 57 us.
#ifndef ROUTE_CHUNK_BWD
#define ROUTE_CHUNK_BWD 64
#endif

__device__ __forceinline__ void load_uh(const float *__restrict__ uh, int b, int L, float *w)
{
#pragma unroll
    for (int k = 0; k < HBVX_UH_MAXLEN; k++) w[k] = (k < L) ? uh[(int64_t)b * L + k] : 0.0f;
}

// uh_conv (uh_routing.py:25-57): y[s,t,b] = sum_k UH[b,k] * q[s,t-k,b], zero history.
// grid.z = series; the chunk is consumed in groups of ROUTE_GROUP days whose loads are all issued
// before the first is used (a thread would otherwise have one load in flight at a time).
#ifndef ROUTE_GROUP
#define ROUTE_GROUP 8
#endif
__global__ void __launch_bounds__(256) k_route_fwd(int T, int B, int S, int L,
                                                   const float *__restrict__ q,
                                                   const float *__restrict__ uh,
                                                   float *__restrict__ y)
{
    const int b = blockIdx.x * 64 + (threadIdx.x & 63);
    const int chunk = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int t0 = chunk * ROUTE_CHUNK;
    if (b >= B || t0 >= T) return;
    const int t1 = min(T, t0 + ROUTE_CHUNK);
    const int s = blockIdx.z;
    float w[HBVX_UH_MAXLEN];
    load_uh(uh, b, L, w);
    const float *qs = q + (int64_t)s * T * B + b;
    float *ys = y + (int64_t)s * T * B + b;
    float win[HBVX_UH_MAXLEN];
#pragma unroll
    for (int k = 1; k < HBVX_UH_MAXLEN; k++) win[k] = (t0 - k >= 0) ? qs[(int64_t)(t0 - k) * B] : 0.0f;
    for (int tg = t0; tg < t1; tg += ROUTE_GROUP) {
        float v[ROUTE_GROUP];
#pragma unroll
        for (int j = 0; j < ROUTE_GROUP; j++) v[j] = (tg + j < t1) ? qs[(int64_t)(tg + j) * B] : 0.0f;
#pragma unroll
        for (int j = 0; j < ROUTE_GROUP; j++) {
            win[0] = v[j];
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < HBVX_UH_MAXLEN; k++) acc += w[k] * win[k];
            if (tg + j < t1) ys[(int64_t)(tg + j) * B] = acc;
#pragma unroll
            for (int k = HBVX_UH_MAXLEN - 1; k > 0; k--) win[k] = win[k - 1];
        }
    }
}

// conv1d backward in one pass over the chunk:
//   gq[s,t,b] = sum_k UH[b,k] * gy[s,t+k,b]                      (w.r.t. the input)
//   ws[(chunk*L+k)*B + b] = sum_{s, t in chunk} gy[s,t,b] * q[s,t-k,b]   (partial tap gradients)
__global__ void __launch_bounds__(256) k_route_bwd(int T, int B, int S, int L,
                                                   const float *__restrict__ q,
                                                   const float *__restrict__ uh,
                                                   const float *__restrict__ gy,
                                                   float *__restrict__ gq, double *__restrict__ ws)
{
    const int b = blockIdx.x * 64 + (threadIdx.x & 63);
    const int chunk = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int t0 = chunk * ROUTE_CHUNK_BWD;
    if (b >= B || t0 >= T) return;
    const int t1 = min(T, t0 + ROUTE_CHUNK_BWD);
    // The tap gradients are summed in DOUBLE: what leaves this path is sum_k gw[k] w_k (ln t_k - <ln t>), whose weights
    // sum to zero (the hydrograph is normalised), so the common part of the gw[k] -- nearly all of them when the
    // inflow is smooth -- cancels and float32 sums (eps x ~1e+2) left 1e-6 where the result is 1e-4: 0.7 % off the
    // float64 value on hbv_ties' route_b, twenty times the reference's own error (round 5).
    float w[HBVX_UH_MAXLEN];
    double gw[HBVX_UH_MAXLEN];
    load_uh(uh, b, L, w);
#pragma unroll
    for (int k = 0; k < HBVX_UH_MAXLEN; k++) gw[k] = 0.0;
    for (int s = 0; s < S; s++) {
    const float *qs = q + (int64_t)s * T * B + b;
    const float *gs = gy + (int64_t)s * T * B + b;
    float *gqs = gq + (int64_t)s * T * B + b;
    float qwin[HBVX_UH_MAXLEN], gwin[HBVX_UH_MAXLEN]; // q[t-k], gy[t+k]
#pragma unroll
    for (int k = 1; k < HBVX_UH_MAXLEN; k++) qwin[k] = (t0 - k >= 0) ? qs[(int64_t)(t0 - k) * B] : 0.0f;
#pragma unroll
    for (int k = 0; k < HBVX_UH_MAXLEN - 1; k++) gwin[k + 1] = (t0 + k < T) ? gs[(int64_t)(t0 + k) * B] : 0.0f;
    for (int tg = t0; tg < t1; tg += ROUTE_GROUP) {
        float vq[ROUTE_GROUP], vg[ROUTE_GROUP];
#pragma unroll
        for (int j = 0; j < ROUTE_GROUP; j++) {
            const int t = tg + j;
            vq[j] = (t < t1) ? qs[(int64_t)t * B] : 0.0f;
            vg[j] = (t + HBVX_UH_MAXLEN - 1 < T) ? gs[(int64_t)(t + HBVX_UH_MAXLEN - 1) * B] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < ROUTE_GROUP; j++) {
            const bool on = tg + j < t1;
#pragma unroll
            for (int k = 0; k < HBVX_UH_MAXLEN - 1; k++) gwin[k] = gwin[k + 1];
            gwin[HBVX_UH_MAXLEN - 1] = vg[j];
            qwin[0] = vq[j];
            float acc = 0.0f;
            const double g0 = on ? (double)gwin[0] : 0.0;
#pragma unroll
            for (int k = 0; k < HBVX_UH_MAXLEN; k++) {
                acc += w[k] * gwin[k];
                gw[k] = fma(g0, (double)qwin[k], gw[k]);
            }
            if (on) gqs[(int64_t)(tg + j) * B] = acc;
#pragma unroll
            for (int k = HBVX_UH_MAXLEN - 1; k > 0; k--) qwin[k] = qwin[k - 1];
        }
    }
    }
    if (ws) {
#pragma unroll
        for (int k = 0; k < HBVX_UH_MAXLEN; k++)
            if (k < L) ws[((int64_t)chunk * L + k) * B + b] = gw[k];
    }
}

// stage 2: reduce the chunks (fixed order: deterministic) and go through the normalised
// gamma UH to the routing inputs.  Block = 16 basins x 16 tap-threads x 4 parts: thread (bl, k, part) sums tap k over
// the chunks c = part (mod 4) in ascending order (part 0 also the last nchunk mod 4 ones), the parts are combined as
// (p0 + p1) + (p2 + p3) -- the association the one-thread-per-tap version used, on four times the threads and blocks
// (11 blocks of 1 024 at config 2 were a 35 us latency chain of 229 dependent loads) -- then the k == 0 threads finish
// per basin.
// d w_k / d aa    = w_k (ln t_k - sum_j w_j ln t_j)      (Gamma(aa) and theta^aa cancel
// d w_k / d theta = w_k (t_k - sum_j w_j t_j) / theta^2    in the normalisation)
#define ROUTE_PB 16      // basins per block of k_route_bwd_params
__global__ void __launch_bounds__(1024) k_route_bwd_params(const hbvx_route_desc r, int nchunk,
                                                           const float *__restrict__ uh,
                                                           const double *__restrict__ ws,
                                                           float *grad_ra, float *grad_rb)
{
    __shared__ double prt[4][16][ROUTE_PB];
    __shared__ double red[16][ROUTE_PB];
    const int bl = threadIdx.x & (ROUTE_PB - 1), k = (threadIdx.x >> 4) & 15, part = threadIdx.x >> 8;
    const int b = blockIdx.x * ROUTE_PB + bl;
    const int B = r.B, L = r.L;
    double acc = 0.0;
    if (b < B && k < L) {
        const double *src = ws + (int64_t)k * B + b;
        const int64_t cs = (int64_t)L * B;
        const int n4 = nchunk & ~3;
        for (int c = part; c < n4; c += 4) acc += src[(int64_t)c * cs];
        if (part == 0)
            for (int c = n4; c < nchunk; c++) acc += src[(int64_t)c * cs];
    }
    prt[part][k][bl] = acc;
    __syncthreads();
    if (part == 0) red[k][bl] = (prt[0][k][bl] + prt[1][k][bl]) + (prt[2][k][bl] + prt[3][k][bl]);
    __syncthreads();
    if (part != 0 || k != 0 || b >= B) return;
    float ua, ub, a, bb;
    route_ab(r, b, ua, ub, a, bb);
    const float theta = fmaxf(bb, 0.0f) + 0.5f;
    // the finish in double as well: the cancellation happens HERE (sum_j c_j = 0), on sums that are now exact to 1e-16
    double w[HBVX_UH_MAXLEN], gw[HBVX_UH_MAXLEN];
    double mlt = 0.0, mt = 0.0;
#pragma unroll
    for (int j = 0; j < HBVX_UH_MAXLEN; j++) {
        gw[j] = (j < L) ? red[j][bl] : 0.0;
        w[j] = (j < L) ? (double)uh[(int64_t)b * L + j] : 0.0;
        const double tk = (double)j + 0.5;
        mlt += w[j] * log(tk);
        mt += w[j] * tk;
    }
    double gaa = 0.0, gth = 0.0;
    const double th2 = (double)theta * (double)theta;
#pragma unroll
    for (int j = 0; j < HBVX_UH_MAXLEN; j++) {
        const double tk = (double)j + 0.5;
        gaa += gw[j] * w[j] * (log(tk) - mlt);
        gth += gw[j] * w[j] * ((tk - mt) / th2);
    }
    float ga = (a > 0.0f) ? (float)gaa : 0.0f;   // relu backward (uh_routing.py:11-14)
    float gb = (bb > 0.0f) ? (float)gth : 0.0f;
    float gua = ga * (r.a_hi - r.a_lo), gub = gb * (r.b_hi - r.b_lo);
    if (r.raw_sigmoid) {
        gua *= ua * (1.0f - ua);
        gub *= ub * (1.0f - ub);
    }
    if (grad_ra) grad_ra[(int64_t)b * r.r_stride] += gua;
    if (grad_rb) grad_rb[(int64_t)b * r.r_stride] += gub;
}

// ---------------------------------------------------------------------------
// host side of the C ABI
// ---------------------------------------------------------------------------
int hbvx_host::check_desc(const hbvx_desc *d)
{
    if (!d) return fail(HBVX_E_NULL, "desc is NULL");
    if (d->abi_version != HBVX_ABI_VERSION) return fail(HBVX_E_ABI, "abi_version mismatch");
    if (d->T < 0 || d->B <= 0 || d->M <= 0 || d->M > 64) return fail(HBVX_E_SHAPE, "bad T/B/M");
    bool ok = false;
    if (d->model == HBVX_MODEL_HBV10) ok = (d->n_param == 12 || d->n_param == 13);
    else if (d->model == HBVX_MODEL_HBV11P) ok = (d->n_param == 14);
    else if (d->model == HBVX_MODEL_HBV20) ok = (d->n_param == 16);
    else if (d->model == HBVX_MODEL_HBVADJ) ok = (d->n_param == 12 || d->n_param == 13);
    else if (d->model == HBVX_MODEL_HOURLY) ok = (d->n_param == 19);
    else return fail(HBVX_E_UNSUPPORTED, "unknown model");
    if (!ok) return fail(HBVX_E_SHAPE, "n_param does not match model");
    if (!d->x && d->T > 0) return fail(HBVX_E_NULL, "forcing pointer is NULL");   /* an empty record has no forcings */
    if (d->ch_prcp < 0 || d->ch_tmean < 0 || d->ch_pet < 0)
        return fail(HBVX_E_SHAPE, "negative forcing channel");
    if (!(d->nearzero >= 0.0f)) return fail(HBVX_E_SHAPE, "nearzero must be >= 0 (the storages' lower clamp, hbv.py:54)");
    if ((d->model == HBVX_MODEL_HBV20 || d->model == HBVX_MODEL_HOURLY) && (!d->ac || !d->elev))
        return fail(HBVX_E_NULL, "HBV 2.0 needs ac and elev");
    for (int i = 0; i < d->n_param; i++)
        if (!d->p[i].sta) return fail(HBVX_E_NULL, "static parameter pointer is NULL");
    // Bounds come from the caller (a model's parameter_bounds table may be edited).  Two short forms of the forward
    // equal the reference only inside the reference's own tables (hbv.py:88-101, hbv_2_hourly.py:97-119) and are
    // refused outside them instead of silently computing something else (hbv_step.h):
    //  * the evaporation factor is taken from the storage BEFORE the excess leaves it (fwd_soil, CHAIN): equal to
    //    hbv.py:474-477 iff LP <= 1 (with LP > 1 an excess day gives (1 / LP)**BETAET < 1 there, 1 here);
    //  * the powers run on 2^(y log2 x) without a lower clamp of the base (pow_unit_): x = 0 needs y > 0
    //    (0 * -inf is NaN where torch's 0**0 is 1), i.e. positive lower bounds of BETA, BETAET, ALPHA.
    if (d->model != HBVX_MODEL_HBVADJ) {
        if (!(d->p[P_LP].hi <= 1.0f && d->p[P_LP].lo <= 1.0f))
            return fail(HBVX_E_UNSUPPORTED, "parLP bounds above 1 are not supported (evaporation factor on an excess day, hbv.py:474-477)");
        if (!(d->p[P_BETA].lo > 0.0f && d->p[P_BETA].hi > 0.0f))
            return fail(HBVX_E_UNSUPPORTED, "parBETA bounds must be positive (soil wetness power, hbv.py:462)");
        if (d->n_param > P_BETAET && !(d->p[P_BETAET].lo > 0.0f && d->p[P_BETAET].hi > 0.0f))
            return fail(HBVX_E_UNSUPPORTED, "parBETAET bounds must be positive (evaporation factor power, hbv.py:476)");
        if (d->model == HBVX_MODEL_HOURLY && !(d->p[P_ALPHA].lo > 0.0f && d->p[P_ALPHA].hi > 0.0f))
            return fail(HBVX_E_UNSUPPORTED, "parALPHA bounds must be positive (infiltration power, hbv_2_hourly.py:590-600)");
    }
    return HBVX_OK;
}

int hbvx_host::lg_members(int M)
{
    int lg = 0;
    while ((1 << lg) < M) lg++;
    return lg;
}

template <typename Args, typename K0, typename K1, typename K2, typename K3, typename K4>
static hipError_t launch_variant(const hbvx_desc *d, const Args &a, dim3 grid, hipStream_t st,
                                 K0 k0, K1 k1, K2 k2, K3 k3, K4 k4)
{
    if (d->model == HBVX_MODEL_HOURLY) {
        hipLaunchKernelGGL(k4, grid, dim3(64), 0, st, a);
        return hipGetLastError();
    }
    if (d->model == HBVX_MODEL_HBV10 && d->n_param == 12) hipLaunchKernelGGL(k0, grid, dim3(64), 0, st, a);
    else if (d->model == HBVX_MODEL_HBV10) hipLaunchKernelGGL(k1, grid, dim3(64), 0, st, a);
    else if (d->model == HBVX_MODEL_HBV11P) hipLaunchKernelGGL(k2, grid, dim3(64), 0, st, a);
    else hipLaunchKernelGGL(k3, grid, dim3(64), 0, st, a);
    return hipGetLastError();
}


int hbvx_host::env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// HBVX_KERNEL=simple selects the one-wave kernels (kept as a cross-check); default is the
// wave-specialised / streaming / time-parallel families.
bool hbvx_host::use_tiled(const hbvx_desc *d)
{
    const char *v = getenv("HBVX_KERNEL");
    if (v && !strcmp(v, "simple")) return false;
    return d->T > 0;
}

hipError_t hbvx_host::set_dynamic_lds(const void *kern, int lds)
{
    struct Entry { const void *k; int dev, lds; };
    static std::mutex mu;
    static Entry seen[256];
    static int n_seen = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    Entry *e = nullptr;
    for (int i = 0; i < n_seen && !e; i++)
        if (seen[i].k == kern && seen[i].dev == dev) e = &seen[i];
    if (e && lds <= e->lds) return hipSuccess;
    const hipError_t rc = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (rc != hipSuccess) return rc;
    if (e) e->lds = lds;
    else if (n_seen < 256) seen[n_seen++] = Entry{kern, dev, lds};
    // (a full table only costs the driver round trip again: the attribute is still raised before the launch)
    return hipSuccess;
}

int hbvx_host::count_dyn(const hbvx_desc *d)
{
    int nd = 0;
    for (int i = 0; i < d->n_param; i++) nd += d->p[i].dyn ? 1 : 0;
    return nd;
}

int hbvx_host::device_cu_count()
{
    static int n_cu[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (n_cu[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n_cu[dev] = v;
    }
    return n_cu[dev];
}

bool &hbvx_host::zero_taken()
{
    static thread_local bool taken = false;
    return taken;
}

static int forward_dispatch(const hbvx_desc *d, const hbvx_fwd_out *out, void *stream);

extern "C" int hbvx_forward(const hbvx_desc *d, const hbvx_fwd_out *out, void *stream)
{
    zero_taken() = false;
    if (out && out->zero_ptr && !out->zero_state) return fail(HBVX_E_NULL, "zero_ptr needs zero_state");
    return forward_dispatch(d, out, stream);
}

static int forward_dispatch(const hbvx_desc *d, const hbvx_fwd_out *out, void *stream)
{
    int rc = check_desc(d);
    if (rc) return rc;
    if (d->model == HBVX_MODEL_HBVADJ) return fail(HBVX_E_UNSUPPORTED, "use hbvx_adj_forward for HBVADJ");
    if (!out || !out->state_out) return fail(HBVX_E_NULL, "state_out is NULL");
    const int want_nf = (d->model == HBVX_MODEL_HBV10) ? 11 : 12;
    if (out->flux && out->n_flux != want_nf) return fail(HBVX_E_SHAPE, "n_flux does not match model");
    if (out->traj && out->traj_layout != HBVX_TRAJ_ROWS) {
        const int kind = HBVX_TRAJ_KIND(out->traj_layout);
        if (kind == HBVX_TRAJ_CKPT) {
            // K-day checkpoints: the streaming kernel where it has an instance, else the generic one
            const int K = HBVX_TRAJ_CKPT_DAYS(out->traj_layout);
            if (K != 4 && K != 8 && K != 16) return fail(HBVX_E_SHAPE, "checkpoint interval must be 4, 8 or 16");
            if (out->aux) return fail(HBVX_E_SHAPE, "checkpoints: aux must be NULL");
            if (try_fwd_stream(d, out, stream, &rc)) return rc;
            if (try_fwd_pipe(d, out, stream, &rc)) return rc;
            if (try_fwd_stream(d, out, stream, &rc, true)) return rc;
            FwdArgs ca;
            ca.d = *d;
            ca.o = *out;
            ca.lgMp = lg_members(d->M);
            const int bpw_c = 64 >> ca.lgMp;
            note_dispatch(0, "simple");
            hipError_t ec = launch_variant(d, ca, dim3((d->B + bpw_c - 1) / bpw_c), (hipStream_t)stream,
                                           k_fwd<MODEL_HBV10, false>, k_fwd<MODEL_HBV10, true>,
                                           k_fwd<MODEL_HBV11P, true>, k_fwd<MODEL_HBV20, true>, k_fwd<MODEL_HOURLY, true>);
            return ec == hipSuccess ? HBVX_OK : hip_fail(ec, "hbvx_forward (checkpoints) launch");
        }
        // packed trajectory: the streaming family only (hbvx_preferred_traj_layout said so)
        if (out->traj_layout != HBVX_TRAJ_PACKED) return fail(HBVX_E_SHAPE, "unknown traj_layout");
        if (try_fwd_stream(d, out, stream, &rc, true)) return rc;
        return fail(HBVX_E_UNSUPPORTED, "packed trajectory: no forward kernel for this call");
    }
    // kernel families in order of preference ("Which kernel runs when"): the streaming
    // kernels take the grids above their measured cross-over, the pipelined forward the rest of what
    // it can hold, the tiled forward whatever is left
    if (try_fwd_stream(d, out, stream, &rc)) return rc;
    if (try_fwd_pipe(d, out, stream, &rc)) return rc;
    if (try_fwd_tiled(d, out, stream, &rc)) return rc;
    FwdArgs a;
    a.d = *d;
    a.o = *out;
    a.lgMp = lg_members(d->M);
    const int bpw = 64 >> a.lgMp;
    dim3 grid((d->B + bpw - 1) / bpw);
    note_dispatch(0, "simple");
    hipError_t e = launch_variant(d, a, grid, (hipStream_t)stream,
                                  k_fwd<MODEL_HBV10, false>, k_fwd<MODEL_HBV10, true>,
                                  k_fwd<MODEL_HBV11P, true>, k_fwd<MODEL_HBV20, true>, k_fwd<MODEL_HOURLY, true>);
    if (e != hipSuccess) return hip_fail(e, "hbvx_forward launch");
    return HBVX_OK;
}

extern "C" int hbvx_backward(const hbvx_desc *d, const hbvx_bwd_io *io, void *stream)
{
    int rc = check_desc(d);
    if (rc) return rc;
    if (d->model == HBVX_MODEL_HBVADJ) return fail(HBVX_E_UNSUPPORTED, "use hbvx_adj_backward for HBVADJ");
    if (!io || !io->traj) return fail(HBVX_E_NULL, "traj is NULL");
    if (SAVE_POW && !io->aux && HBVX_TRAJ_KIND(io->traj_layout) != HBVX_TRAJ_CKPT) return fail(HBVX_E_NULL, "aux is NULL");
    const int want_nf = (d->model == HBVX_MODEL_HBV10) ? 11 : 12;
    if (io->n_flux != want_nf) return fail(HBVX_E_SHAPE, "n_flux does not match model");
    if (d->T == 0) return HBVX_OK;
    if (io->traj_layout != HBVX_TRAJ_ROWS) {
        if (HBVX_TRAJ_KIND(io->traj_layout) == HBVX_TRAJ_CKPT) {
            if (try_bwd_stream_ckpt(d, io, stream, &rc)) return rc;
            if (try_bwd_ckpt(d, io, stream, &rc)) return rc;
            return fail(HBVX_E_UNSUPPORTED, "checkpoints: no adjoint kernel for this call");
        }
        if (io->traj_layout != HBVX_TRAJ_PACKED) return fail(HBVX_E_SHAPE, "unknown traj_layout");
        if (try_bwd_stream(d, io, stream, &rc)) return rc;
        return fail(HBVX_E_UNSUPPORTED, "packed trajectory: no adjoint kernel for this call");
    }
    if (try_bwd_stream(d, io, stream, &rc)) return rc;
    if (try_bwd_chunked(d, io, stream, &rc)) return rc;
    if (try_bwd_tiled(d, io, stream, &rc)) return rc;
    BwdArgs a;
    a.d = *d;
    a.io = *io;
    a.lgMp = lg_members(d->M);
    const int bpw = 64 >> a.lgMp;
    dim3 grid((d->B + bpw - 1) / bpw);
    store_gate(io, (hipStream_t)stream);
    note_dispatch(1, "simple");
    hipError_t e = launch_variant(d, a, grid, (hipStream_t)stream,
                                  k_bwd<MODEL_HBV10, false>, k_bwd<MODEL_HBV10, true>,
                                  k_bwd<MODEL_HBV11P, true>, k_bwd<MODEL_HBV20, true>, k_bwd<MODEL_HOURLY, true>);
    if (e != hipSuccess) return hip_fail(e, "hbvx_backward launch");
    return HBVX_OK;
}

static int check_route(const hbvx_route_desc *r)
{
    if (!r) return fail(HBVX_E_NULL, "route desc is NULL");
    if (r->abi_version != HBVX_ABI_VERSION) return fail(HBVX_E_ABI, "abi_version mismatch");
    if (r->T <= 0 || r->B <= 0 || r->S <= 0) return fail(HBVX_E_SHAPE, "bad T/B/S");
    const int L = r->T < HBVX_UH_MAXLEN ? r->T : HBVX_UH_MAXLEN;
    if (r->L != L) return fail(HBVX_E_SHAPE, "L must be min(T, 15)");
    if (!r->ra || !r->rb) return fail(HBVX_E_NULL, "routing parameter pointer is NULL");
    return HBVX_OK;
}

extern "C" int hbvx_route_forward(const hbvx_route_desc *r, const float *q, float *uh,
                                  float *q_rout, void *stream)
{
    int rc = check_route(r);
    if (rc) return rc;
    if (!q || !uh || !q_rout) return fail(HBVX_E_NULL, "route buffer is NULL");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_uh_gamma, dim3((r->B + 63) / 64), dim3(64), 0, st, *r, uh);
    const int nchunk = (r->T + ROUTE_CHUNK - 1) / ROUTE_CHUNK;
    hipLaunchKernelGGL(k_route_fwd, dim3((r->B + 63) / 64, (nchunk + 3) / 4, r->S), dim3(256), 0, st, r->T,
                       r->B, r->S, r->L, q, uh, q_rout);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "hbvx_route_forward launch");
    return HBVX_OK;
}

static int route_chunks(const hbvx_route_desc *r) { return (r->T + ROUTE_CHUNK_BWD - 1) / ROUTE_CHUNK_BWD; }   // the adjoint's

extern "C" uint64_t hbvx_route_workspace_bytes(const hbvx_route_desc *r)
{
    if (!r || r->T <= 0 || r->B <= 0) return 0;
    return (uint64_t)route_chunks(r) * (uint64_t)r->L * (uint64_t)r->B * sizeof(double);    // (8-byte aligned: torch's allocations are)
}

extern "C" int hbvx_route_backward(const hbvx_route_desc *r, const float *q, const float *uh,
                                   const float *grad_q_rout, float *grad_q, float *grad_ra,
                                   float *grad_rb, void *workspace, uint64_t workspace_bytes,
                                   void *stream)
{
    int rc = check_route(r);
    if (rc) return rc;
    if (!q || !uh || !grad_q_rout || !grad_q) return fail(HBVX_E_NULL, "route buffer is NULL");
    hipStream_t st = (hipStream_t)stream;
    const bool want_p = grad_ra || grad_rb;
    if (want_p && (!workspace || workspace_bytes < hbvx_route_workspace_bytes(r)))
        return fail(HBVX_E_NULL, "route workspace missing or too small");
    const int nchunk = route_chunks(r);
    hipLaunchKernelGGL(k_route_bwd, dim3((r->B + 63) / 64, (nchunk + 3) / 4), dim3(256), 0, st, r->T,
                       r->B, r->S, r->L, q, uh, grad_q_rout, grad_q, want_p ? (double *)workspace : nullptr);
    if (want_p)
        hipLaunchKernelGGL(k_route_bwd_params, dim3((r->B + ROUTE_PB - 1) / ROUTE_PB), dim3(1024), 0, st, *r, nchunk,
                           uh, (const double *)workspace, grad_ra, grad_rb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "hbvx_route_backward launch");
    return HBVX_OK;
}

// ---------------------------------------------------------------------------
// diagnostics: device-side accuracy probe of hbvx::pow_pos_ (tests only)
// ---------------------------------------------------------------------------
__global__ void k_selftest_pow(const float *x, const float *y, float *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pow_pos_(x[i], y[i]);
}

extern "C" int hbvx_selftest_pow(const float *x, const float *y, float *out, int n, void *stream)
{
    if (!x || !y || !out || n <= 0) return fail(HBVX_E_NULL, "selftest_pow: bad arguments");
    hipLaunchKernelGGL(k_selftest_pow, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y,
                       out, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "selftest_pow launch");
    return HBVX_OK;
}

__global__ void k_selftest_div(const float *x, const float *y, float *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = div_(x[i], y[i]);
}

extern "C" int hbvx_selftest_div(const float *x, const float *y, float *out, int n, void *stream)
{
    if (!x || !y || !out || n <= 0) return fail(HBVX_E_NULL, "selftest_div: bad arguments");
    hipLaunchKernelGGL(k_selftest_div, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y,
                       out, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "selftest_div launch");
    return HBVX_OK;
}

// ---------------------------------------------------------------------------
// baseflow index (hbv.py:562-567): block = 16 basins x 64 time slices (64-byte row segments; four
// times the blocks of a 64-basin tile: B / 16 instead of B / 64 CUs busy), fixed-order sums
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_bfi(int T, int B, const float *__restrict__ qs,
                                              const float *__restrict__ q2, float nz,
                                              float *__restrict__ bfi)
{
    __shared__ float r0[64][16], r2[64][16];
    const int bl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int b = blockIdx.x * 16 + bl;
    float a0 = 0.0f, a2 = 0.0f, c0 = 0.0f, c2 = 0.0f;
    if (b < B) {
        int t = sl;
        for (; t + 64 < T; t += 128) {
            a0 += qs[(int64_t)t * B + b];
            a2 += q2[(int64_t)t * B + b];
            c0 += qs[(int64_t)(t + 64) * B + b];
            c2 += q2[(int64_t)(t + 64) * B + b];
        }
        if (t < T) {
            a0 += qs[(int64_t)t * B + b];
            a2 += q2[(int64_t)t * B + b];
        }
    }
    r0[sl][bl] = a0 + c0;
    r2[sl][bl] = a2 + c2;
    __syncthreads();
    if (sl == 0 && b < B) {
        float s0 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int k = 0; k < 64; k++) {
            s0 += r0[k][bl];
            s2 += r2[k][bl];
        }
        bfi[b] = 100.0f * (s2 / (s0 + nz));
    }
}

extern "C" int hbvx_bfi(int32_t T, int32_t B, const float *qs, const float *q2, float nearzero,
                        float *bfi, void *stream)
{
    if (!qs || !q2 || !bfi || T <= 0 || B <= 0) return fail(HBVX_E_NULL, "hbvx_bfi: bad arguments");
    hipLaunchKernelGGL(k_bfi, dim3((B + 15) / 16), dim3(1024), 0, (hipStream_t)stream, T, B, qs, q2,
                       nearzero, bfi);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "hbvx_bfi launch");
    return HBVX_OK;
}

// ---------------------------------------------------------------------------
// gage routing (hbv_2_hourly.py:800-897)
// ---------------------------------------------------------------------------
static int check_gage(const hbvx_gage_desc *r)
{
    if (!r) return fail(HBVX_E_NULL, "gage desc is NULL");
    if (r->abi_version != HBVX_ABI_VERSION) return fail(HBVX_E_ABI, "abi_version mismatch");
    if (r->T <= 0 || r->U <= 0 || r->G <= 0 || r->NPAIR < 0) return fail(HBVX_E_SHAPE, "bad T/U/G/NPAIR");
    int L = r->T < HBVX_GAGE_MAXLEN ? r->T : HBVX_GAGE_MAXLEN;
    if (r->L != L) return fail(HBVX_E_SHAPE, "L must be min(T, 72)");
    if (r->U > 65535 || r->G > 65535) return fail(HBVX_E_SHAPE, "more than 65535 units or gages per call");
    if (!r->pair_unit || !r->gage_ptr || !r->pair_gage || !r->unit_ptr || !r->unit_pairs || !r->areas ||
        !r->denom || !r->dp)
        return fail(HBVX_E_NULL, "gage routing pointer is NULL");
    return HBVX_OK;
}

extern "C" uint64_t hbvx_gage_route_workspace_bytes(const hbvx_gage_desc *r)
{
    if (!r || r->T <= 0 || r->U <= 0 || r->G <= 0) return 0;
    // transposed runoff [U,T], transposed gradient [G,T], transposed input gradient [U,T],
    // per-pair series [NPAIR,T]
    return ((uint64_t)2 * r->U + r->G + r->NPAIR) * (uint64_t)r->T * sizeof(float);
}

static void launch_transpose(int R, int C, const float *in, float *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_transpose, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, st, R, C, in, out);
}

extern "C" int hbvx_gage_route_forward(const hbvx_gage_desc *r, const float *qs, float *uh, float *out,
                                       void *workspace, uint64_t workspace_bytes, void *stream)
{
    int rc = check_gage(r);
    if (rc) return rc;
    if (!qs || !uh || !out) return fail(HBVX_E_NULL, "gage routing buffer is NULL");
    if (!workspace || workspace_bytes < hbvx_gage_route_workspace_bytes(r))
        return fail(HBVX_E_NULL, "gage routing workspace missing or too small");
    hipStream_t st = (hipStream_t)stream;
    float *qsT = (float *)workspace;
    launch_transpose(r->T, r->U, qs, qsT, st);
    if (r->NPAIR > 0) hipLaunchKernelGGL(k_gage_uh, dim3(r->NPAIR), dim3(128), 0, st, *r, uh);
    float *lag = qsT + ((int64_t)2 * r->U + r->G) * r->T;
    if (r->NPAIR > 0)
        hipLaunchKernelGGL(k_gage_lag_fwd, dim3(r->NPAIR, (r->T + GAGE_TILE4 - 1) / GAGE_TILE4), dim3(GAGE_TILE), 0,
                           st, *r, qsT, uh, lag);
    hipLaunchKernelGGL(k_gage_sum_fwd, dim3((r->T + 255) / 256, r->G), dim3(256), 0, st, *r, lag, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "hbvx_gage_route_forward launch");
    return HBVX_OK;
}

extern "C" int hbvx_gage_route_backward(const hbvx_gage_desc *r, const float *qs, const float *uh,
                                        const float *grad_out, float *grad_qs, float *grad_dp,
                                        void *workspace, uint64_t workspace_bytes, void *stream)
{
    int rc = check_gage(r);
    if (rc) return rc;
    if (!qs || !uh || !grad_out || !grad_qs || !grad_dp) return fail(HBVX_E_NULL, "gage routing buffer is NULL");
    if (!workspace || workspace_bytes < hbvx_gage_route_workspace_bytes(r))
        return fail(HBVX_E_NULL, "gage routing workspace missing or too small");
    hipStream_t st = (hipStream_t)stream;
    float *qsT = (float *)workspace;
    float *goT = qsT + (int64_t)r->U * r->T;
    float *gqsT = goT + (int64_t)r->G * r->T;
    launch_transpose(r->T, r->U, qs, qsT, st);
    launch_transpose(r->T, r->G, grad_out, goT, st);
    float *lag = gqsT + (int64_t)r->U * r->T;
    if (r->NPAIR > 0)
        hipLaunchKernelGGL(k_gage_lag_bwd, dim3(r->NPAIR, (r->T + GAGE_TILE4 - 1) / GAGE_TILE4), dim3(GAGE_TILE), 0,
                           st, *r, uh, goT, lag);
    hipLaunchKernelGGL(k_gage_sum_bwd, dim3((r->T + 255) / 256, r->U), dim3(256), 0, st, *r, lag, gqsT);
    launch_transpose(r->U, r->T, gqsT, grad_qs, st);
    if (r->NPAIR > 0)
        hipLaunchKernelGGL(k_gage_bwd_p, dim3(r->NPAIR), dim3(GAGE_TILE), 0, st, *r, qsT, goT, grad_dp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "hbvx_gage_route_backward launch");
    return HBVX_OK;
}

// ---------------------------------------------------------------------------
// zero fill of the dense gradient buffers (the autograd contract wants grad tensors shaped like the raw
// parameter tensor [T,B,ny], of which static parameters touch one row): streaming 16-byte non-temporal
// stores, grid-stride, no read-for-ownership
// ---------------------------------------------------------------------------
typedef float zero_f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_zero_nt(zero_f4 *__restrict__ p, uint64_t n16)
{
    const zero_f4 z = {0.f, 0.f, 0.f, 0.f};
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256)
        __builtin_nontemporal_store(z, &p[i]);
}

// Small and ragged pieces go through a kernel too, never hipMemsetAsync: captured into a HIP graph (graph=True), the
// memset node of a 1 944-byte static-parameter gradient replayed as a no-op on this stack -- the adjoint then added
// its sums to stale memory (round 5: tests/test_graphed.py, Hbv_2's [B, ws] gradient).  A kernel node replays.
__global__ void __launch_bounds__(256) k_zero_bytes(unsigned char *__restrict__ p, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0;
}
static hipError_t zero_bytes(void *ptr, uint64_t n, hipStream_t st)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_zero_bytes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (unsigned char *)ptr, n);
    return hipGetLastError();
}

extern "C" int hbvx_zero(void *ptr, uint64_t bytes, void *stream)
{
    if (!ptr && bytes) return fail(HBVX_E_NULL, "hbvx_zero: buffer is NULL");
    hipStream_t st = (hipStream_t)stream;
    const uint64_t head = ((uintptr_t)ptr & 15) ? 16 - ((uintptr_t)ptr & 15) : 0;
    if (bytes < 4096 || head >= bytes) {
        hipError_t e = zero_bytes(ptr, bytes, st);
        return e == hipSuccess ? 0 : hip_fail(e, "hbvx_zero");
    }
    hipError_t e = hipSuccess;
    if (head) e = zero_bytes(ptr, head, st);
    const uint64_t n16 = (bytes - head) / 16, tail = (bytes - head) - n16 * 16;
    // one 16-byte store per thread: on MI355X 6.6 TB/s against 5.3 for a grid-stride loop on a few thousand
    // workgroups (3.8 GB: 0.58 ms against 0.72 ms for torch's fill)
    const uint64_t want = n16 / 256 + 1;
    const int blocks = (int)(want < 0x7fffffffull ? want : 0x7fffffffull);
    hipLaunchKernelGGL(k_zero_nt, dim3(blocks), dim3(256), 0, st, (zero_f4 *)((char *)ptr + head), n16);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess && tail) e = zero_bytes((char *)ptr + head + n16 * 16, tail, st);
    return e == hipSuccess ? 0 : hip_fail(e, "hbvx_zero");
}

// hbvx_zero_rest (include/hbvx.h): what the forward's fill workgroups left of hbvx_fwd_out.zero_ptr.  The first missing piece
// is a device-side number (zero_state[0]), so the grid covers the whole buffer and the threads in front of it leave: eight
// 16-byte stores per thread, a wave instruction = 1 KB contiguous.
__global__ void __launch_bounds__(256) k_zero_rest(zero_f4 *__restrict__ p, uint64_t n16, const unsigned *__restrict__ state)
{
    const uint64_t npiece = (n16 + HBVX_ZERO_PIECE / 16 - 1) / (HBVX_ZERO_PIECE / 16);
    const uint64_t claimed = state[0];
    const uint64_t start = (claimed < npiece ? claimed : npiece) * (HBVX_ZERO_PIECE / 16);
    const uint64_t i0 = (uint64_t)blockIdx.x * 2048 + threadIdx.x;
    if (i0 + 2048 <= start) return;
    const zero_f4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint64_t i = i0 + (uint64_t)k * 256;
        if (i >= start && i < n16) __builtin_nontemporal_store(z, &p[i]);
    }
}

extern "C" int hbvx_zero_rest(void *ptr, uint64_t bytes, const void *zero_state, void *stream)
{
    if (!bytes) return HBVX_OK;
    if (!ptr || !zero_state) return fail(HBVX_E_NULL, "hbvx_zero_rest: buffer or state is NULL");
    if ((uintptr_t)ptr & 15) return fail(HBVX_E_SHAPE, "hbvx_zero_rest: buffer must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const uint64_t n16 = bytes / 16, tail = bytes - n16 * 16;
    hipError_t e = hipSuccess;
    if (n16) {
        const uint64_t blocks = (n16 + 2047) / 2048;
        if (blocks > 0x7fffffffull) return fail(HBVX_E_SHAPE, "hbvx_zero_rest: buffer too large");
        hipLaunchKernelGGL(k_zero_rest, dim3((unsigned)blocks), dim3(256), 0, st, (zero_f4 *)ptr, n16, (const unsigned *)zero_state);
        e = hipGetLastError();
    }
    if (e == hipSuccess && tail) e = zero_bytes((char *)ptr + n16 * 16, tail, st);
    return e == hipSuccess ? HBVX_OK : hip_fail(e, "hbvx_zero_rest");
}

// Zero fill that leaves out what the adjoint overwrites (include/hbvx.h: hbvx_zero_except).
// Rows outside [r0, r1) are contiguous memory: plain hbvx_zero.  Inside, only the gaps between
// the kept column groups are written: 16 lanes per row walk the (few, short) gap ranges.  When less
// than half of a row is kept the dense fill is cheaper than the gap walk and is used instead (the
// adjoint's stores then land on zeros).
struct ZeroGaps {
    int n;
    int start[34], len[34];
};

__global__ void __launch_bounds__(256) k_zero_gaps(float *__restrict__ p, int64_t r0, int64_t r1, int width,
                                                    const ZeroGaps G)
{
    const int64_t row = r0 + (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= r1) return;
    float *q = p + row * width;
    const int lane = threadIdx.x & 15;
    for (int g = 0; g < G.n; g++)
        for (int c = lane; c < G.len[g]; c += 16) q[G.start[g] + c] = 0.0f;
}

extern "C" int hbvx_zero_except(float *ptr, int64_t rows, int32_t width, int64_t r0, int64_t r1, int32_t group_w,
                                uint32_t keep_groups, void *stream)
{
    if (rows < 0 || width <= 0 || group_w <= 0) return fail(HBVX_E_SHAPE, "hbvx_zero_except: bad shape");
    if (!ptr && rows) return fail(HBVX_E_NULL, "hbvx_zero_except: buffer is NULL");
    if (rows == 0) return 0;
    r0 = r0 < 0 ? 0 : r0;
    r1 = r1 > rows ? rows : r1;
    // gap ranges = complement of the kept groups inside [0, width)
    ZeroGaps G;
    G.n = 0;
    int kept_cols = 0, c = 0;
    while (c < width) {
        const int g = c / group_w;
        const bool kept = g < 32 && ((keep_groups >> g) & 1u);
        int e = (g + 1) * group_w;
        e = e > width ? width : e;
        if (kept) kept_cols += e - c;
        else if (G.n > 0 && G.start[G.n - 1] + G.len[G.n - 1] == c) G.len[G.n - 1] += e - c;
        else if (G.n < 34) { G.start[G.n] = c; G.len[G.n] = e - c; G.n++; }
        c = e;
    }
    if (r1 <= r0 || 2 * kept_cols < width)
        return hbvx_zero(ptr, (uint64_t)rows * (uint64_t)width * sizeof(float), stream);
    int rc = 0;
    if (r0 > 0) rc = hbvx_zero(ptr, (uint64_t)r0 * (uint64_t)width * sizeof(float), stream);
    if (!rc && r1 < rows)
        rc = hbvx_zero(ptr + r1 * width, (uint64_t)(rows - r1) * (uint64_t)width * sizeof(float), stream);
    if (rc || G.n == 0) return rc;
    const int64_t nblk = (r1 - r0 + 15) / 16;
    if (nblk > 0x7fffffff) return hbvx_zero(ptr, (uint64_t)rows * (uint64_t)width * sizeof(float), stream);
    hipLaunchKernelGGL(k_zero_gaps, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, ptr, r0, r1, (int)width, G);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "hbvx_zero_except");
}

