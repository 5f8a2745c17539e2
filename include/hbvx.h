/*
 * hbvx.h -- C ABI of the MI355X-native HBV time-stepper (libhbvx.so).
 *
 * The reference (mhpi/hydrodl2) has no FFI of its own: its plug-in contract is
 * the Python class `Hbv(config, device).forward(x_dict, parameters)`
 * (src/hydrodl2/models/hbv/hbv.py:284-361).  This header declares the entry
 * points a binding for that path needs; each one names the reference code it
 * replaces.  Plain pointers and sizes only -- no torch types.
 *
 *   hbvx_forward         replaces the `for t in range(nsteps)` recurrence and the
 *                        ensemble mean: hbv.py:423-511, hbv_1_1p.py:422-521,
 *                        hbv_2.py:464-581, fused with the parameter prep
 *                        (sigmoid, static-row pick, dy_drop blend, de-scaling:
 *                        hbv.py:201-208,236-256, core/calc/utils.py:9-24).
 *   hbvx_backward        replaces the autograd tape of the same lines
 *                        (SURVEY.md §3.4 / §8 a11): hand-written adjoint.
 *   hbvx_route_forward   replaces uh_gamma + uh_conv on the ensemble means:
 *                        core/calc/uh_routing.py:5-57, called at hbv.py:523-538.
 *   hbvx_route_backward  replaces conv1d/lgamma/pow autograd of those lines.
 *
 * Ownership: the caller allocates and owns every buffer; the library keeps no
 * state between calls and allocates nothing persistent.  All device work is
 * enqueued on `stream` (a hipStream_t passed as void*); no call synchronises.
 * Errors: 0 on success, a negative HBVX_E_* otherwise; hbvx_last_error() returns
 * a thread-local message.  Nothing throws across the ABI.
 *
 * The CPU oracle (oracle/hbv_oracle.c) implements the same ABI on host memory
 * (stream ignored) so that tests can drive both with identical descriptors.
 */
#ifndef HBVX_H
#define HBVX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HBVX_ABI_VERSION 10
#define HBVX_MAX_PARAM 20
#define HBVX_NSTATE 5   /* SNOWPACK, MELTWATER, SM, SUZ, SLZ  (hbv.py:61-67) */
#define HBVX_MAX_FLUX 12
#define HBVX_UH_MAXLEN 15 /* lenF (hbv.py:526, hbv_2.py:52) */

/* Parameter slots: the order of `parameter_bounds` in every variant
 * (hbv.py:88-101,124-125; hbv_1_1p.py:87-102; hbv_2.py:90-107). */
enum hbvx_param_slot {
    HBVX_P_BETA = 0, HBVX_P_FC, HBVX_P_K0, HBVX_P_K1, HBVX_P_K2, HBVX_P_LP,
    HBVX_P_PERC, HBVX_P_UZL, HBVX_P_TT, HBVX_P_CFMAX, HBVX_P_CFR, HBVX_P_CWH,
    HBVX_P_BETAET, HBVX_P_C, HBVX_P_RT, HBVX_P_AC,
    HBVX_P_F0, HBVX_P_FMIN, HBVX_P_ALPHA /* hourly Hortonian infiltration (hbv_2_hourly.py:108-114) */
};

enum hbvx_model {
    HBVX_MODEL_HBV10 = 0,  /* n_param 12, or 13 when parBETAET is present */
    HBVX_MODEL_HBV11P = 1, /* n_param 14: + parBETAET always, capillary rise */
    HBVX_MODEL_HBV20 = 2,  /* n_param 16: + elevation TT switch, lateral flow */
    HBVX_MODEL_HBVADJ = 3, /* implicit (backward-Euler) HBV, hbvx_adj_* only; n_param 12 or 13 */
    HBVX_MODEL_HOURLY = 4  /* n_param 19: HBV 2.0 in rate form with dt = 1/24 day, storage guard-rails,
                              Hortonian infiltration excess (hbv_2_hourly.py:527-675); forcings are
                              per-step depths, fluxes are rates per day */
};

/* Ensemble-mean series written by hbvx_forward, flux[k][t][b]
 * (the buffers of hbv.py:402-420 after mean(-1): hbv.py:509,575-588). */
enum hbvx_flux {
    HBVX_F_QSIM = 0, /* Q0+Q1+Q2; weighted by muwts when given (hbv.py:508-511) */
    HBVX_F_Q0, HBVX_F_Q1, HBVX_F_Q2, HBVX_F_AET, HBVX_F_SWE, HBVX_F_RECHARGE,
    HBVX_F_EXCS, HBVX_F_EVAPFACTOR, HBVX_F_TOSOIL, HBVX_F_PERC,
    HBVX_F_CAPILLARY /* 1.1p / 2.0 only */
};

enum hbvx_error {
    HBVX_OK = 0,
    HBVX_E_NULL = -1,        /* required pointer missing */
    HBVX_E_SHAPE = -2,       /* T/B/M/n_param out of range */
    HBVX_E_UNSUPPORTED = -3, /* variant / option not built */
    HBVX_E_ABI = -4,         /* abi_version mismatch */
    HBVX_E_DEVICE = -5       /* HIP runtime error (message has the hipError) */
};

/* Where one physical parameter comes from.  Values are "unit" parameters
 * u in [0,1] (after the optional sigmoid); physical value = u*(hi-lo)+lo. */
typedef struct hbvx_param_src {
    const float *dyn;     /* per-step values or NULL (static parameter):
                             (t,b,j) at dyn[t*dyn_t_stride + b*dyn_b_stride + j] */
    const float *sta;     /* static values, required: (b,j) at sta[b*sta_b_stride + j]
                             (the reference's "last row": hbv.py:242) */
    const uint8_t *drop;  /* optional [B]; 1 = basin uses the static value although
                             the parameter is dynamic (dy_drop mask, hbv.py:245-246) */
    int64_t dyn_t_stride;
    int64_t dyn_b_stride;
    int64_t sta_b_stride;
    float lo, hi;         /* bounds (parameter_bounds) */
} hbvx_param_src;

/* Where the gradient w.r.t. the *input* values (raw if raw_sigmoid, else unit)
 * goes; same addressing as hbvx_param_src.  `dyn` rows are overwritten for
 * every (t,b,j) of the call; `sta` is accumulated into (+=) after the dyn rows,
 * so both may alias the same tensor (the reference's static row T-1 lies inside
 * the dynamic tensor).  NULL pointers skip that gradient. */
typedef struct hbvx_param_grad {
    float *dyn;
    float *sta;
    int64_t dyn_t_stride;
    int64_t dyn_b_stride;
    int64_t sta_b_stride;
} hbvx_param_grad;

typedef struct hbvx_desc {
    int32_t abi_version;  /* HBVX_ABI_VERSION */
    int32_t model;        /* enum hbvx_model */
    int32_t T;            /* steps in this call */
    int32_t B;            /* basins */
    int32_t M;            /* ensemble members per basin (nmul), 1..64 */
    int32_t n_param;      /* 12, 13, 14 or 16 */
    int32_t raw_sigmoid;  /* 1: inputs are raw NN outputs, apply sigmoid first
                             (hbv.py:201); 0: inputs already in [0,1] (hbv_2.py:211) */
    int32_t ch_prcp, ch_tmean, ch_pet; /* channel of each forcing (hbv.py:388-390) */
    float nearzero;       /* hbv.py:54 */
    int32_t adj_stop;     /* hbvx_adj_* only: how a day's system G(x) = 0 is solved and who decides that the
                             iteration stops.
                             0: joint modified Newton on the five storages (hbv_adj.py:507-581), every lane
                                stops for itself.
                             1: the same, the slowest lane of the WAVEFRONT (64 lanes = 64/Mp basins x Mp
                                members) decides -- the reference's rule, one torch.max over the batch
                                (hbv_adj.py:544,546), restricted to the lanes that share a wavefront: a
                                batch-wide maximum would need a grid barrier per Newton update.
                             2: staged solve along the block lower-triangular structure of dG/dx
                                (hbv_adj.py:425-429): snow, upper zone and lower zone in closed form
                                (piecewise linear, monotone), soil moisture by scalar Newton under
                                adj_gtol / adj_max_iter, per lane.  Same acceptance test |G|_inf <= gtol;
                                the blocks of consecutive days run as a wave pipeline. */
    const float *x;       /* forcings: (t,b,c) at x[t*x_t_stride + b*x_b_stride + c] */
    int64_t x_t_stride, x_b_stride;
    const float *ac;      /* [B] HBV 2.0 `ac_all`  (hbv_2.py:345), else NULL */
    const float *elev;    /* [B] HBV 2.0 `elev_all` (hbv_2.py:346), else NULL */
    const float *muwts;   /* optional ensemble weights (t,b,j) at
                             muwts[t*mu_t_stride + b*mu_b_stride + j] (strides may be 0) */
    int64_t mu_t_stride, mu_b_stride;
    const float *state_in; /* [5,B,M] or NULL = every storage 0.001 (hbv.py:128-136);
                              hbvx_adj_*: NULL = 0 (hbv_adj.py:254) */
    hbvx_param_src p[HBVX_MAX_PARAM];
    /* hbvx_adj_* only (ignored elsewhere): Newton policy of hbv_adj.py:518-519,544 */
    float adj_gtol;        /* stop when max_i |G_i| <= gtol; reference 1e-3 */
    int32_t adj_max_iter;  /* updates allowed = max_iter + 1; reference max_iter = 3 */
} hbvx_desc;

/* Layout of the saved trajectory (hbvx_fwd_out.traj, same buffer and size either way):
 *   ROWS    traj [5,T+1,N] (storage k entering day t at traj[(k*(T+1)+t)*N + n])
 *   PACKED  traj = records [T+1,N,4] (SNOWPACK, MELTWATER, SM, SUZ) followed by SLZ rows [T+1,N].
 *           Two wide stores / loads per lane-day instead of five: what the streaming kernels for
 *           large grids use (a vector-memory instruction costs the same issue time whatever its
 *           width).  N = B*M, lane n = b*M + j.
 * `aux` (ABI <= 8: the two pre-clamp powers of every lane-day, [2,T,N] / records [T,N,2]) is no longer
 * part of the trajectory: since ABI 9 the adjoint recomputes both powers from the saved storages with
 * the forward's own instruction sequence (20 instead of 28 bytes per lane-day), and the library neither
 * writes nor reads the pointer -- pass NULL.  (Development builds with -DHBVX_SAVE_POW=1 restore the
 * old behaviour for A/B measurements; hbvx_backend() then ends in "+savepow".) */
enum hbvx_traj_layout { HBVX_TRAJ_ROWS = 0, HBVX_TRAJ_PACKED = 1, HBVX_TRAJ_CKPT = 2 };
/* traj_layout = kind | (K << 8).  HBVX_TRAJ_CKPT with K in {4, 8, 16}: `traj` holds only the five
 * storages entering days 0, K, 2K, ... as [ceil(T/K), 5, N] (20/K bytes per lane-day instead of 20);
 * hbvx_backward re-materialises each K-day segment from its checkpoint (one extra
 * forward step per day).  The caller chooses it when the full trajectory does not fit
 * (100 000 basins x 16 x 7 300 days: 327 GB of trajectory against 29 GB of checkpoints at K = 8, 15 GB at
 * K = 16; the block-wise adjoint re-materialises one block of days at a time into caller scratch:
 * at most HBVX_CKPT_SCRATCH_MB (default 2048 MB) of trajectory -- or 2 K days of it, 40 K N bytes, if that
 * is more -- plus the block's gradient-series windows, two 20 N-byte carries and the inner adjoint's own
 * workspace; hbvx_ckpt_workspace_bytes returns the exact sum). */
#define HBVX_TRAJ_KIND(layout) ((layout) & 0xFF)
#define HBVX_TRAJ_CKPT_DAYS(layout) ((layout) >> 8)

typedef struct hbvx_fwd_out {
    float *flux;      /* [n_flux,T,B] or NULL (state warm-up: hbv.py:557-559) */
    float *state_out; /* [5,B,M] storages after the last step, required */
    float *traj;      /* optional [5,T+1,B*M]: storages entering step t; row T = final.
                         Needed by hbvx_backward; rows 1..T are HBV 2.0's state series
                         (hbv_2.py:571-575) */
    float *aux;       /* unused since ABI 9 (see above): NULL */
    int32_t n_flux;   /* 11 (HBV 1.0) or 12 */
    int32_t traj_layout; /* enum hbvx_traj_layout: how traj / aux are laid out.  Must be what
                            hbvx_preferred_traj_layout() returns for this desc, or HBVX_TRAJ_ROWS */
    void *zero_ptr;      /* ABI 10, optional: a caller-owned buffer (16-byte aligned) the caller needs ZEROED later -- its
                            dense gradient tensor ([T,B,ny]: 3.8 GB at config 2).  The pipelined forward is a latency
                            chain on 168 of 256 CUs; surplus workgroups of the SAME launch, on the CUs it leaves idle,
                            write zeros into this buffer in 256 KB pieces, front to back, for as long as the recurrence
                            runs, and stop with it (same stream, same launch: no cross-stream ownership of the buffer,
                            and the forward is never longer than its recurrence).  How many pieces they wrote is left in
                            zero_state[0]; hbvx_zero_rest() then fills what is missing -- all of it when another
                            kernel family took the call (zero_state[0] stays 0).  NULL: nothing.  The CPU oracle
                            leaves everything to hbvx_zero_rest. */
    uint64_t zero_bytes; /* multiple of 16 */
    void *zero_state;    /* device memory, two uint32 words, ZERO on entry ([0]: pieces claimed, [1]: workgroups of the
                            recurrence that have finished); required with zero_ptr */
} hbvx_fwd_out;

typedef struct hbvx_bwd_io {
    const float *traj;       /* from hbvx_forward, required */
    const float *aux;        /* unused since ABI 9: NULL */
    const float *grad_flux;  /* [n_flux,T,B] dL/d(flux series) or NULL (= zeros) */
    const float *grad_flux4; /* optional [4,T,B]: extra gradient of series 0..3 (Qsim,Q0,Q1,Q2),
                                i.e. grad_q of hbvx_route_backward; added to grad_flux */
    const float *grad_state_out; /* optional [5,B,M]: dL/d(final storages) (adjoint seed; NULL = 0) */
    float *grad_x;           /* optional, addressed like desc->x (overwritten) */
    float *grad_muwts;       /* optional [T,B,M] contiguous (overwritten) */
    float *grad_state_in;    /* optional [5,B,M] (overwritten) */
    int32_t n_flux;
    int32_t traj_layout;     /* the layout the forward call wrote traj / aux in */
    hbvx_param_grad g[HBVX_MAX_PARAM];
    void *workspace;          /* optional caller-owned scratch of hbvx_backward_workspace_bytes();
                                 enables the time-parallel (chunked) adjoint */
    uint64_t workspace_bytes;
    void *store_gate;         /* optional hipEvent_t: `stream` waits for it before the first kernel that STORES
                                 dynamic-parameter / forcing / muwts gradients is launched (kernels that only read
                                 -- the transfer-map and scan passes of the time-parallel adjoint -- run ahead of
                                 it).  Lets the caller fill the gradient tensor ([T,B,ny]: 3.8 GB at config 2) on
                                 another stream BESIDE the first half of the adjoint instead of in front of it.
                                 NULL: no wait.  The CPU oracle ignores it. */
} hbvx_bwd_io;

/* Unit-hydrograph routing of S series that share one UH per basin. */
typedef struct hbvx_route_desc {
    int32_t abi_version;
    int32_t T, B, S;      /* steps, basins, series (4: Qsim,Q0,Q1,Q2 at hbv.py:530-538) */
    int32_t L;            /* UH length = min(T, 15) (uh_routing.py:8) */
    int32_t raw_sigmoid;  /* routing parameters are raw (hbv.py:212) or unit (hbv_2.py:228) */
    const float *ra;      /* route_a input of basin b at ra[b*r_stride] */
    const float *rb;      /* route_b input of basin b at rb[b*r_stride] */
    int64_t r_stride;
    float a_lo, a_hi, b_lo, b_hi; /* routing_parameter_bounds (hbv.py:102-105) */
} hbvx_route_desc;

int hbvx_version(void);                 /* HBVX_ABI_VERSION */
const char *hbvx_last_error(void);
const char *hbvx_backend(void);         /* "hip:gfx950" or "cpu-oracle" */
uint64_t hbvx_sizeof(int which);        /* 0 desc, 1 fwd_out, 2 bwd_io, 3 route_desc,
                                           4 param_src, 5 param_grad, 6 gage_desc: layout check */
/* Diagnostic: the kernel family that took the process's last hbvx_forward (direction 0) / hbvx_backward (1) call --
 * "pipe", "stream2", "stream", "tiled", "simple", "chunked", "ckpt-block:<family>", "ckpt-lds"; "oracle" in the CPU
 * restatement.  The parity tests assert that the family they mean to pin against the reference is the one that ran. */
const char *hbvx_last_dispatch(int direction);
/* 1 when the last hbvx_forward / hbvx_adj_forward of this thread carried fill workgroups for hbvx_fwd_out.zero_ptr in its
 * launch (the pipelined forward with idle CUs), else 0 (diagnostic, like hbvx_last_dispatch). */
int hbvx_zero_in_launch(void);
/* Zero what the forward's launch left of hbvx_fwd_out.zero_ptr: pieces zero_state[0] .. end (HBVX_ZERO_PIECE bytes each).
 * Asynchronous on `stream`, which must be ordered behind the forward call. */
#define HBVX_ZERO_PIECE (256u * 1024u)
int hbvx_zero_rest(void *ptr, uint64_t bytes, const void *zero_state, void *stream);

/* The trajectory layout this library wants for the problem (grid size, dynamic set): pass the
 * value in hbvx_fwd_out.traj_layout and hbvx_bwd_io.traj_layout.  HBVX_TRAJ_ROWS is always accepted. */
int hbvx_preferred_traj_layout(const hbvx_desc *d);

int hbvx_forward(const hbvx_desc *d, const hbvx_fwd_out *out, void *stream);
int hbvx_backward(const hbvx_desc *d, const hbvx_bwd_io *io, void *stream);
/* Scratch bytes the time-parallel adjoint wants for this problem (0: not applicable). */
uint64_t hbvx_backward_workspace_bytes(const hbvx_desc *d);
/* Scratch bytes hbvx_backward wants with HBVX_TRAJ_CKPT and interval K (hbvx_bwd_io.workspace): one
 * block of re-materialised trajectory + the block's gradient series + what the regular adjoint needs
 * for a block.  With it the adjoint runs block-wise on the fast kernels; without, a serial fallback. */
uint64_t hbvx_ckpt_workspace_bytes(const hbvx_desc *d, int32_t K);

/* q [S,T,B] -> uh [B,L] (normalised gamma UH) and q_rout [S,T,B]. */
int hbvx_route_forward(const hbvx_route_desc *r, const float *q, float *uh, float *q_rout,
                       void *stream);
/* Baseflow index (hbv.py:562-567): bfi[b] = 100 * sum_t q2[t,b] / (sum_t qs[t,b] + nearzero), with
 * qs, q2 [T,B] (routed streamflow and routed groundwater flow); fixed-order sums (deterministic). */
int hbvx_bfi(int32_t T, int32_t B, const float *qs, const float *q2, float nearzero, float *bfi,
             void *stream);

/* Scratch bytes hbvx_route_backward needs (per-time-chunk partial tap gradients). */
uint64_t hbvx_route_workspace_bytes(const hbvx_route_desc *r);
/* grad_q_rout [S,T,B] -> grad_q [S,T,B] (overwritten) and the gradient w.r.t. the
 * routing inputs, accumulated (+=) at grad_ra[b*r_stride], grad_rb[b*r_stride].
 * `workspace` is caller-owned scratch of at least hbvx_route_workspace_bytes(r). */
int hbvx_route_backward(const hbvx_route_desc *r, const float *q, const float *uh,
                        const float *grad_q_rout, float *grad_q, float *grad_ra,
                        float *grad_rb, void *workspace, uint64_t workspace_bytes,
                        void *stream);

/* Gage routing of unit runoff (hbv_2_hourly.py:800-897 `distr_routing` + `_frac_shift1d`): for every
 * (gage, unit) pair with outlet_topo == 1 a normalised gamma unit hydrograph of L = min(T, 72)
 * taps (core/calc/uh_routing.py:5-22), shifted by tau = k + f as (1-f) w[t-k] + f w[t-k-1] with zero
 * padding when lag_uh; out[t,g] = sum_{pairs of g} sum_k uh[p,k] qs[t-k,unit(p)] areas[unit(p)] / denom[g].
 * With the identity topology (one pair per unit, areas = denom = 1, lag_uh = 0) this is the
 * 72-tap per-unit routing of hbv_2_hourly.py:693-700.
 * Index arrays are built by the caller from outlet_topo: pairs sorted by gage (gage_ptr: CSR over
 * gages) and, for the backward gather, the pair indices grouped by unit (unit_ptr / unit_pairs). */
#define HBVX_GAGE_MAXLEN 72
typedef struct hbvx_gage_desc {
    int32_t abi_version;
    int32_t T, U, G;           /* steps, units, gages */
    int32_t NPAIR;             /* (gage, unit) pairs */
    int32_t L;                 /* min(T, 72) */
    int32_t lag_uh;            /* apply the fractional shift by route_tau */
    int32_t reserved0;
    const int32_t *pair_unit;  /* [NPAIR] unit of each pair, pairs ordered by gage */
    const int32_t *gage_ptr;   /* [G+1] pairs of gage g: gage_ptr[g] .. gage_ptr[g+1]-1 */
    const int32_t *pair_gage;  /* [NPAIR] gage of each pair */
    const int32_t *unit_ptr;   /* [U+1] CSR over units into unit_pairs */
    const int32_t *unit_pairs; /* [NPAIR] pair indices grouped by unit */
    const float *areas;        /* [U] */
    const float *denom;        /* [G] upstream area of each gage, clamped at 1e-6 (:849) */
    const float *dp;           /* [NPAIR,3] unit-interval route_a, route_b, route_tau (contiguous) */
    float a_lo, a_hi, b_lo, b_hi, tau_lo, tau_hi; /* distr_parameter_bounds (:120-124) */
} hbvx_gage_desc;

/* Scratch for the two calls below (bytes; caller-owned, contents undefined afterwards): the
 * time-major runoff / gradient are transposed once per call so that a unit's (gage's) series is
 * contiguous when the kernels stage it. */
uint64_t hbvx_gage_route_workspace_bytes(const hbvx_gage_desc *r);
/* qs [T,U] -> uh [NPAIR,L] (the lagged unit hydrographs, kept for the backward) and out [T,G]. */
int hbvx_gage_route_forward(const hbvx_gage_desc *r, const float *qs, float *uh, float *out,
                            void *workspace, uint64_t workspace_bytes, void *stream);
/* grad_out [T,G] -> grad_qs [T,U] and grad_dp [NPAIR,3] (both overwritten). */
int hbvx_gage_route_backward(const hbvx_gage_desc *r, const float *qs, const float *uh,
                             const float *grad_out, float *grad_qs, float *grad_dp,
                             void *workspace, uint64_t workspace_bytes, void *stream);

/* Implicit HBV ("HBV adjoint", hbv_adj.py): per day solve G(x) = (x - x_t)/dt - f(x, theta_t, t) = 0
 * (hbv_adj.py:669-687) by modified Newton (hbv_adj.py:507-581) with the analytic 5x5 Jacobian;
 * flux[0][t][b] = ensemble mean of q0+q1+q2 at the solved state (hbv_adj.py:309-317).
 * Uses hbvx_desc (model HBVX_MODEL_HBVADJ; `drop` is per LANE [B*M], hbv_adj.py:182-189),
 * hbvx_fwd_out with n_flux = 1 (traj rows 1..T = solved states) and hbvx_bwd_io.
 * The backward is the implicit-function adjoint the reference intends (hbv_adj.py:617-633):
 * (dG/dx)^T lambda = dL/dx, dL/dtheta = -lambda^T dG/dtheta, dL/dx_t = lambda/dt. */
int hbvx_adj_forward(const hbvx_desc *d, const hbvx_fwd_out *out, void *stream);
int hbvx_adj_backward(const hbvx_desc *d, const hbvx_bwd_io *io, void *stream);

/* Zero `bytes` bytes at `ptr` (streaming non-temporal stores).  The autograd contract of the plug-in wants
 * gradient tensors shaped like the raw parameter tensor [T,B,ny] (hbv.py:211-246: static parameters read
 * one row of it), so the dense zero fill is part of every backward step. */
int hbvx_zero(void *ptr, uint64_t bytes, void *stream);

/* The same for a gradient buffer of which hbvx_backward will overwrite a part: `ptr` is a dense
 * [rows, width] float matrix (rows = T*B of the raw parameter tensor, width = ny); every element is zeroed
 * EXCEPT those with r0 <= row < r1 and column in a kept group -- column c belongs to group c / group_w,
 * group g (< 32) is kept when bit g of `keep_groups` is set.  Those are exactly the elements the
 * adjoint stores (never accumulates) for a dynamic parameter: rows = the days of the call, group =
 * the parameter's nmul columns (hbv.py:201-208).  With every parameter dynamic (BASELINE config 3)
 * this removes a 4.4 GB fill per step.  keep_groups == 0 is hbvx_zero. */
int hbvx_zero_except(float *ptr, int64_t rows, int32_t width, int64_t r0, int64_t r1, int32_t group_w,
                     uint32_t keep_groups, void *stream);

/* Diagnostics (tests only): out[i] = the device pow used for (SM/FC)**BETA on x[i], y[i]. */
int hbvx_selftest_pow(const float *x, const float *y, float *out, int n, void *stream);
/* Diagnostics (tests only): out[i] = the device quotient used for SM/FC on x[i] / y[i]. */
int hbvx_selftest_div(const float *x, const float *y, float *out, int n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HBVX_H */
