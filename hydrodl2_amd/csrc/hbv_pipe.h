// hbv_pipe.h -- pipelined forward for HBV 1.0 with static parameters.
//
// In HBV 1.0 one day is a feed-forward chain of three sub-recurrences
//     snow (SNOWPACK, MELTWATER) -> soil (SM) -> groundwater (SUZ, SLZ)
// (hbv.py:429-459 -> :462-480 -> :483-494; no storage feeds back into an earlier stage; HBV 1.1p
// and 2.0 break this with capillary rise SLZ -> SM).  Each sub-recurrence is serial in time, but
// stage s of tile k only needs stage s-1 of the same tile.  So three waves of one workgroup run the
// three stages one tile apart -- a software pipeline through LDS -- and the per-day latency of the
// workgroup becomes the LONGEST stage (soil: the pow and two divisions) instead of the sum.
//
//   wave 0  snow         tile it      reads forcings, writes RAIN/tosoil for the soil wave
//   wave 1  soil         tile it-1    reads RAIN/tosoil + PET, writes recharge/excess for wave 2
//   wave 2  groundwater  tile it-2
//   waves 3,4  fillers   stage the forcings of tile it+2 (loads only)
//   waves 5..11 drainers outputs of tiles it-1/it-2/it-3: ensemble means -> flux series,
//                        storages -> trajectory, pow results -> aux (stores only)
//
// One raw s_barrier per iteration; every interface is double-buffered (forcings: 4 slots, filled
// two tiles ahead and read by two stages).  The arithmetic is Step::fwd_snow / fwd_soil / fwd_gw of hbv_step.h, i.e.
// operation for operation what the tiled kernel computes; ensemble sums use the same member order,
// so both kernels give bit-identical results.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/hbvx.h"
#include "hbv_step.h"
#include "hbv_tiled.h"

namespace hbvx {

struct PipeArgs {
    hbvx_desc d;
    hbvx_fwd_out o;
    int lgMp;
    int Kt;
};

// LDS layout in floats for Kt days per tile
struct PipeLds {
    int xin, ab, bc, oa, ob, oc, total;
    __host__ __device__ explicit PipeLds(int Kt)
    {
        xin = 0;                    // [4][Kt][64][4]
        ab = xin + 4 * Kt * 256;    // [2][Kt][2][64]
        bc = ab + 2 * Kt * 128;     // [2][Kt][2][64]
        oa = bc + 2 * Kt * 128;     // [2][Kt][4][64]
        ob = oa + 2 * Kt * 256;     // [2][Kt][7][64]
        oc = ob + 2 * Kt * 448;     // [2][Kt][7][64]
        total = oc + 2 * Kt * 448;
    }
};

// Drain one stage's output tile (helper waves).  The tile holds NSER series per day: NFS flux
// series first (global flux index = nibble i of FMAP), then NTS storage series (nibble i of KIND:
// < 5 -> trajectory row of that storage, 5/6 -> aux 0/1).  Everything that indexes the layout is a
// compile-time constant (no integer division in the loop).
template <int NSER, int NFS, unsigned FMAP, int NTS, unsigned KIND>
__device__ __forceinline__ void pipe_drain(const hbvx_desc &d, const hbvx_fwd_out &o, const LaneT &L,
                                           const float *buf, int t0, int nt, int w, int NHW, int hid,
                                           int nhid, int lgMp, int b0, bool has_traj, bool has_aux)
{
    const int T = d.T;
    const int64_t N = (int64_t)d.B * d.M;
    const int bpw = 64 >> lgMp, M = d.M;
    const float invM = 1.0f / (float)M;
    if (has_traj) {
        for (int tt = w; tt < nt; tt += NHW) {
            const int t = t0 + tt;
            const float *src = buf + (tt * NSER + NFS) * 64 + L.lane;
            float v[NTS];
#pragma unroll
            for (int i = 0; i < NTS; i++) v[i] = src[i * 64];
            if (L.active) {
#pragma unroll
                for (int i = 0; i < NTS; i++) {
                    constexpr unsigned kinds = KIND;
                    const int kd = (int)((kinds >> (4 * i)) & 0xFu);
                    if (kd < 5) o.traj[((int64_t)kd * (T + 1) + t) * N + L.n] = v[i];
                    else if (has_aux) o.aux[((int64_t)(kd - 5) * T + t) * N + L.n] = v[i];
                }
            }
        }
    }
    ens_reduce_tile<NSER, NFS>(buf, nt * NFS * bpw, L.lane, w, NHW, M, lgMp,
                               [&](int tt, int ks, int bl, float acc) {
                                   const int kk = (int)((FMAP >> (4 * ks)) & 0xFu);
                                   if (b0 + bl < d.B)
                                       o.flux[((int64_t)kk * T + (t0 + tt)) * d.B + b0 + bl] = acc * invM;
                               });
}

template <bool BETAET>
__global__ void __launch_bounds__(1024) k_fwd_pipe(const PipeArgs A)
{
    constexpr int NP = BETAET ? 13 : 12;
    constexpr int NF = 11;
    extern __shared__ __align__(16) float lds[];
    const hbvx_desc &d = A.d;
    const hbvx_fwd_out &o = A.o;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const LaneT L = lane_t(d, A.lgMp);
    const int lane = L.lane;
    const int T = d.T, Kt = A.Kt;
    const int nT = (T + Kt - 1) / Kt;
    const int nIt = nT + 3;
    const int64_t N = (int64_t)d.B * d.M;
    const bool raw = d.raw_sigmoid != 0;
    const bool has_traj = o.traj != nullptr, has_aux = o.aux != nullptr;
    const PipeLds P(Kt);
    const float nz = d.nearzero;

    float p[NPARAM_MAX];
#pragma unroll
    for (int i = 0; i < NPARAM_MAX; i++) p[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const hbvx_param_src &s = d.p[i];
        float v = s.sta[(int64_t)L.b * s.sta_b_stride + L.j];
        v = raw ? sigmoid_(v) : v;
        p[i] = descale_(v, s.lo, s.hi);
    }
    auto tile_nt = [&](int tile) { return min(Kt, T - tile * Kt); };

    if (wave == 0) {
        // ------------------------------ snow ------------------------------
        __builtin_amdgcn_s_setprio(3);
        float SP = d.state_in ? d.state_in[0 * N + L.n] : 0.001f;
        float MW = d.state_in ? d.state_in[1 * N + L.n] : 0.001f;
        lds_barrier();
        for (int it = 0; it < nIt; it++) {
            const int tile = it;
            if (tile < nT) {
                const int nt = tile_nt(tile);
                const float4 *in4 = reinterpret_cast<const float4 *>(lds + P.xin + (tile & 3) * Kt * 256);
                float *ab = lds + P.ab + (tile & 1) * Kt * 128 + lane;
                float *oa = lds + P.oa + (tile & 1) * Kt * 256 + lane;
                float4 fn = in4[lane];
                for (int tt = 0; tt < nt; tt++) {
                    Step<MODEL_HBV10, BETAET> s;
                    const float4 f = fn;
                    if (tt + 1 < nt) fn = in4[(tt + 1) * 64 + lane];
                    s.P = f.x; s.Tf = f.y;
                    s.SP = SP; s.MW = MW;
                    s.fwd_snow(p, 0.0f);
                    ab[tt * 128] = s.RAIN;
                    ab[tt * 128 + 64] = s.tosoil;
                    float *q = oa + tt * 256;
                    q[0] = s.SP3; q[64] = s.tosoil; q[128] = SP; q[192] = MW;
                    SP = s.SP3; MW = s.MW3;
                }
            }
            lds_barrier();
        }
        if (L.active) {
            o.state_out[0 * N + L.n] = SP;
            o.state_out[1 * N + L.n] = MW;
            if (has_traj) {
                o.traj[((int64_t)0 * (T + 1) + T) * N + L.n] = SP;
                o.traj[((int64_t)1 * (T + 1) + T) * N + L.n] = MW;
            }
        }
    } else if (wave == 1) {
        // ------------------------------ soil ------------------------------
        __builtin_amdgcn_s_setprio(3);
        float SM = d.state_in ? d.state_in[2 * N + L.n] : 0.001f;
        lds_barrier();
        for (int it = 0; it < nIt; it++) {
            const int tile = it - 1;
            if (tile >= 0 && tile < nT) {
                const int nt = tile_nt(tile);
                const float4 *in4 = reinterpret_cast<const float4 *>(lds + P.xin + (tile & 3) * Kt * 256);
                const float *ab = lds + P.ab + (tile & 1) * Kt * 128 + lane;
                float *bc = lds + P.bc + (tile & 1) * Kt * 128 + lane;
                float *ob = lds + P.ob + (tile & 1) * Kt * 448 + lane;
                float npet = in4[lane].z, nrain = ab[0], nts = ab[64];
                for (int tt = 0; tt < nt; tt++) {
                    Step<MODEL_HBV10, BETAET> s;
                    s.PET = npet; s.RAIN = nrain; s.tosoil = nts;
                    if (tt + 1 < nt) {
                        npet = in4[(tt + 1) * 64 + lane].z;
                        nrain = ab[(tt + 1) * 128];
                        nts = ab[(tt + 1) * 128 + 64];
                    }
                    s.SM = SM;
                    s.template fwd_soil<false>(p, nz, 0.0f, 0.0f);
                    bc[tt * 128] = s.rech;
                    bc[tt * 128 + 64] = s.exc;
                    float *q = ob + tt * 448;
                    q[0] = s.ET; q[64] = s.rech; q[128] = s.exc; q[192] = s.ef;
                    q[256] = SM; q[320] = s.sw0; q[384] = s.ef0;
                    SM = s.SM3;
                }
            }
            lds_barrier();
        }
        if (L.active) {
            o.state_out[2 * N + L.n] = SM;
            if (has_traj) o.traj[((int64_t)2 * (T + 1) + T) * N + L.n] = SM;
        }
    } else if (wave == 2) {
        // --------------------------- groundwater ---------------------------
        __builtin_amdgcn_s_setprio(3);
        float SUZ = d.state_in ? d.state_in[3 * N + L.n] : 0.001f;
        float SLZ = d.state_in ? d.state_in[4 * N + L.n] : 0.001f;
        lds_barrier();
        for (int it = 0; it < nIt; it++) {
            const int tile = it - 2;
            if (tile >= 0 && tile < nT) {
                const int nt = tile_nt(tile);
                const float *bc = lds + P.bc + (tile & 1) * Kt * 128 + lane;
                float *oc = lds + P.oc + (tile & 1) * Kt * 448 + lane;
                float nrech = bc[0], nexc = bc[64];
                for (int tt = 0; tt < nt; tt++) {
                    Step<MODEL_HBV10, BETAET> s;
                    s.rech = nrech; s.exc = nexc;
                    if (tt + 1 < nt) {
                        nrech = bc[(tt + 1) * 128];
                        nexc = bc[(tt + 1) * 128 + 64];
                    }
                    s.SUZ = SUZ; s.SLZ0 = SLZ;
                    s.fwd_gw(p, 0.0f);
                    float *q = oc + tt * 448;
                    q[0] = s.Q; q[64] = s.Q0; q[128] = s.Q1; q[192] = s.Q2; q[256] = s.PERC;
                    q[320] = SUZ; q[384] = SLZ;
                    SUZ = s.SUZ4; SLZ = s.SLZ2;
                }
            }
            lds_barrier();
        }
        if (L.active) {
            o.state_out[3 * N + L.n] = SUZ;
            o.state_out[4 * N + L.n] = SLZ;
            if (has_traj) {
                o.traj[((int64_t)3 * (T + 1) + T) * N + L.n] = SUZ;
                o.traj[((int64_t)4 * (T + 1) + T) * N + L.n] = SLZ;
            }
        }
    } else {
        // ------------------------------ helpers ------------------------------
        // Two kinds, so that no wave waits for its own stores (loads and stores share the in-order
        // vmcnt counter): waves 3,4 "fillers" only LOAD (forcings, two tiles ahead); waves 5.. only
        // STORE (drains).
        const int lgMp = A.lgMp, bpw = 64 >> lgMp;
        const int b0 = blockIdx.x * bpw;
        constexpr unsigned fmapA = HBVX_F_SWE | (HBVX_F_TOSOIL << 4), kindA = 0u | (1u << 4);
        constexpr unsigned fmapB = HBVX_F_AET | (HBVX_F_RECHARGE << 4) | (HBVX_F_EXCS << 8) |
                                   (HBVX_F_EVAPFACTOR << 12),
                           kindB = 2u | (5u << 4) | (6u << 8);
        constexpr unsigned fmapC = HBVX_F_QSIM | (HBVX_F_Q0 << 4) | (HBVX_F_Q1 << 8) | (HBVX_F_Q2 << 12) |
                                   (HBVX_F_PERC << 16),
                           kindC = 3u | (4u << 4);
        if (wave < 5) {
            const int w = wave - 3;            // 0 or 1
            const float *xb = d.x + (int64_t)L.b * d.x_b_stride;
            auto fill = [&](int tile) {
                float4 *in4 = reinterpret_cast<float4 *>(lds + P.xin + (tile & 3) * Kt * 256);
                const int t0 = tile * Kt, nt = tile_nt(tile);
                for (int tt = w; tt < nt; tt += 2) {
                    const float *xr = xb + (int64_t)(t0 + tt) * d.x_t_stride;
                    float4 f;
                    f.x = xr[d.ch_prcp]; f.y = xr[d.ch_tmean]; f.z = xr[d.ch_pet]; f.w = 0.0f;
                    in4[tt * 64 + lane] = f;
                }
            };
            fill(0);
            if (nT > 1) fill(1);
            lds_barrier();
            for (int it = 0; it < nIt; it++) {
                if (it + 2 < nT) fill(it + 2);
                lds_barrier();
            }
        } else {
            const int NDR = (blockDim.x >> 6) - 5;
            const int w = wave - 5;
            const int hid = w * 64 + lane, nhid = NDR * 64;
            lds_barrier();
            for (int it = 0; it < nIt; it++) {
                int tile = it - 1;
                if (tile >= 0 && tile < nT)
                    pipe_drain<4, 2, fmapA, 2, kindA>(d, o, L, lds + P.oa + (tile & 1) * Kt * 256, tile * Kt,
                                                      tile_nt(tile), w, NDR, hid, nhid, lgMp, b0, has_traj, has_aux);
                tile = it - 2;
                if (tile >= 0 && tile < nT)
                    pipe_drain<7, 4, fmapB, 3, kindB>(d, o, L, lds + P.ob + (tile & 1) * Kt * 448, tile * Kt,
                                                      tile_nt(tile), w, NDR, hid, nhid, lgMp, b0, has_traj, has_aux);
                tile = it - 3;
                if (tile >= 0 && tile < nT)
                    pipe_drain<7, 5, fmapC, 2, kindC>(d, o, L, lds + P.oc + (tile & 1) * Kt * 448, tile * Kt,
                                                      tile_nt(tile), w, NDR, hid, nhid, lgMp, b0, has_traj, has_aux);
                lds_barrier();
            }
        }
        (void)NF;
    }
}

} // namespace hbvx
