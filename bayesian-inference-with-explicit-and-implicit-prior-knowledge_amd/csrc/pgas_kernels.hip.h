// pgas_kernels.hip.h -- gfx950 device code of the conditional-SMC engine.
//
// Kernel inventory (DESIGN.md section 5 has the byte accounting and the roofline of each):
//   k_pack        A (nx,M) -> coefficient tensor on the dense frequency grid
//   k_init        x_0 ~ N(m0,P0)                                   src/PGAS.py:155-174,194
//   k_front       per particle: basis, A phi, log-weights, propagate, per-segment softmax scans
//                                                                   src/PGAS.py:45-77,90-118,130-134
//   k_upper       cross-segment CDF (2 blocks: resampling CDF, ancestor CDF + ancestor search)
//                                                                   src/PGAS.py:102,118,121-127
//   k_back        systematic resampling search + weight update      src/Filtering.py:28-35, src/PGAS.py:137-147
//   k_propagate   every particle through a range of time steps, state in registers (no synchronisation)
//   k_resample    per step: resampling search of step t-1, weight update, softmax scans of step t
//   k_segscan     softmax scan of a weight vector (final index)     src/PGAS.py:224
//   k_backtrace   ancestor chase                                    src/Filtering.py:40-55
//   k_basis_eval  phi(x) in reference order (test hook)             src/BasisFunctions.py:77-80
//
// All arithmetic follows the canonical order of DESIGN.md section 4 so that results are bit
// identical to oracle/pgas_canon.c.  Built with -ffp-contract=off: every fused multiply-add
// below is an explicit PGAS_FMA.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pgas_canon.h"

#define PG_BLK 256
#define PG_PPT 4
static_assert(PG_BLK * PG_PPT == PGAS_SEG && PGAS_SEG == 1024, "one workgroup owns one canonical segment of 1024 particles");
#define PG_UPPER_THREADS 256
// W (template parameter of k_propagate) = workgroups resident per CU (= waves per SIMD it is compiled for):
// N = 2^20 is exactly 4 segments per CU, so the 1- and 2-dimensional bases are built for W = 4.
#define PG_MAX_NSEG 8192

// Diagnostic build only (-DPG_STAMPS): per-workgroup wall-clock stamps (100 MHz s_memrealtime) at phase boundaries of
// k_resample_fast, read back with pgas_debug_stamps.  No stamp executes in the product build.
#ifdef PG_STAMPS
__device__ unsigned long long g_stamps[2048 * 16];
#define PG_STAMP(id)                                                                                \
    do {                                                                                            \
        if (threadIdx.x == 0) g_stamps[blockIdx.x * 16 + (id)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define PG_STAMP(id) do { } while (0)
#endif

struct DevModel {
    int32_t N, T, nx, ny, nu, D, M;
    int32_t J[PGAS_MAX_D], j0[PGAS_MAX_D], jstep[PGAS_MAX_D], sel[PGAS_MAX_D];
    double alpha[PGAS_MAX_D], beta[PGAS_MAX_D];
    double nrm;
    double H[PGAS_MAX_NY * 2];
    double LRinv[PGAS_MAX_NY * PGAS_MAX_NY];
    double cR;
    int32_t JP;       // padded innermost grid extent
    int32_t nseg;     // ceil(N / PGAS_SEG): segments of THIS device
    // particle sharding (pgas_shard_setup; single device: p0 = 0, Ng = N, nseg_g = nseg)
    int64_t p0;       // global index of local particle 0
    int32_t Ng;       // global particle count
    int32_t nseg_g;   // global segment count
    const double* y;  // device (T,ny)
    const double* u;  // device (T,nu)
};

struct TransParams {   // transition parameters, by value
    double LS[4], LSinv[4], cS;
    const double* G;   // packed coefficient tensor
};

struct UpperHdr {      // written by k_upper
    double S[2];
    int32_t valid[2];
    int32_t ref_idx;
    int32_t final_idx;
    unsigned long long ref_granule;  // k_resample_fast: {launch tag : 32, ancestor of the conditioned particle : 32}, one 8-byte store
};

#define PG_MAX_RANKS 8
struct Peers {         // device pointers of every rank's buffers (xGMI peer mappings); world == 1: this device only
    const uint64_t* c1[PG_MAX_RANKS];   // the c1 buffer being READ this launch
    const uint64_t* c2[PG_MAX_RANKS];
    const double* laux[PG_MAX_RANKS];   // la_buf bases (row offset added in the kernel)
    const double* x[PG_MAX_RANKS];      // x_trace bases
    const int32_t* anc[PG_MAX_RANKS];   // anc_trace bases
    int32_t world, nseg_l, Nl;          // ranks, segments per rank, particles per rank
};

struct ScanBufs {      // per-step scan scratch (device)
    double* laux;      // (nseg*SEG)
    uint64_t* c1;      // (nseg*SEG) quantised inclusive cumsum of the resampling weights
    uint64_t* c2;      // (nseg*SEG) same for the ancestor weights
    double* segm;      // segment maxima as READ by the cross-segment scan: (ranks, 2, nsegp) after the all-gather
    uint64_t* segs;    // segment totals, same layout
    double* segm_w;    // (2, nsegp) where this device's segment scans WRITE (== segm on a single device)
    uint64_t* segs_w;
    int32_t nseg_l;    // segments per rank in segm/segs (single device: >= nseg, so every segment maps to "rank" 0)
    int32_t rank_stride;  // words between two ranks' blocks in segm/segs (2 * nsegp_local)
    double* excl;      // (2, nsegp)
    double* scale;     // (2, nsegp)
    double* cm;        // (2, nsegp) running max of the segment-end CDF numerators
    UpperHdr* hdr;
    int32_t nsegp;     // padded LOCAL nseg (multiple of 64): stride between the two CDFs in segm/segs
    int32_t nsegp_g;   // padded global nseg: stride between the two CDFs in excl/scale/cm
};

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
// ---- wave-level primitives on DPP (data-parallel primitives: VALU-latency lane exchange, no LDS round trip) ----------
// row_shr:n shifts inside each row of 16 lanes, row_bcast:15 / row_bcast:31 feed a row's / half-wave's last lane to the
// following row(s).  Lanes without a source read 0 (`old` = 0 with bound_ctrl / row masks), and adding 0.0 is exact.
#define PG_DPP_ROW_SHR(n) (0x110 + (n))
#define PG_DPP_ROW_BCAST15 0x142
#define PG_DPP_ROW_BCAST31 0x143
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t v) {
    int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ double readlane_f64(double v, int lane_uniform) {
    const int l = __builtin_amdgcn_readfirstlane(lane_uniform);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// Canonical inclusive scan of one group of 64 doubles ("R16 tree"): Kogge-Stone inside each row of
// 16 lanes (offsets 1,2,4,8), then rows 1,3 add the total of the row before, then rows 2,3 add lane 31's value.
__device__ __forceinline__ double wave_scan_add(double v) {
    v = v + dpp_f64<PG_DPP_ROW_SHR(1), 0xf>(v);
    v = v + dpp_f64<PG_DPP_ROW_SHR(2), 0xf>(v);
    v = v + dpp_f64<PG_DPP_ROW_SHR(4), 0xf>(v);
    v = v + dpp_f64<PG_DPP_ROW_SHR(8), 0xf>(v);
    v = v + dpp_f64<PG_DPP_ROW_BCAST15, 0xa>(v);
    v = v + dpp_f64<PG_DPP_ROW_BCAST31, 0xc>(v);
    return v;
}
// inclusive running maximum of non-negative values (exact in any order; same exchange pattern)
__device__ __forceinline__ double wave_scan_max(double v) {
    v = __builtin_fmax(v, dpp_f64<PG_DPP_ROW_SHR(1), 0xf>(v));
    v = __builtin_fmax(v, dpp_f64<PG_DPP_ROW_SHR(2), 0xf>(v));
    v = __builtin_fmax(v, dpp_f64<PG_DPP_ROW_SHR(4), 0xf>(v));
    v = __builtin_fmax(v, dpp_f64<PG_DPP_ROW_SHR(8), 0xf>(v));
    v = __builtin_fmax(v, dpp_f64<PG_DPP_ROW_BCAST15, 0xa>(v));
    v = __builtin_fmax(v, dpp_f64<PG_DPP_ROW_BCAST31, 0xc>(v));
    return v;
}
// inclusive integer scan (exact in any order)
__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v) {
    v += dpp_u64<PG_DPP_ROW_SHR(1), 0xf>(v);
    v += dpp_u64<PG_DPP_ROW_SHR(2), 0xf>(v);
    v += dpp_u64<PG_DPP_ROW_SHR(4), 0xf>(v);
    v += dpp_u64<PG_DPP_ROW_SHR(8), 0xf>(v);
    v += dpp_u64<PG_DPP_ROW_BCAST15, 0xa>(v);
    v += dpp_u64<PG_DPP_ROW_BCAST31, 0xc>(v);
    return v;
}
// maximum over the wave (NaN ignored, any sign).  Lanes without a DPP source keep their own value (`old` = own), so
// negative values and -inf are handled; the result is read from lane 63 (scalar broadcast).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_keep_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max(double v) {
    v = __builtin_fmax(v, dpp_keep_f64<PG_DPP_ROW_SHR(1), 0xf>(v));
    v = __builtin_fmax(v, dpp_keep_f64<PG_DPP_ROW_SHR(2), 0xf>(v));
    v = __builtin_fmax(v, dpp_keep_f64<PG_DPP_ROW_SHR(4), 0xf>(v));
    v = __builtin_fmax(v, dpp_keep_f64<PG_DPP_ROW_SHR(8), 0xf>(v));
    v = __builtin_fmax(v, dpp_keep_f64<PG_DPP_ROW_BCAST15, 0xa>(v));
    v = __builtin_fmax(v, dpp_keep_f64<PG_DPP_ROW_BCAST31, 0xc>(v));
    return readlane_f64(v, 63);
}
__device__ __forceinline__ int wave_sum_i(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, PG_DPP_ROW_SHR(1), 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, PG_DPP_ROW_SHR(2), 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, PG_DPP_ROW_SHR(4), 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, PG_DPP_ROW_SHR(8), 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, PG_DPP_ROW_BCAST15, 0xa, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, PG_DPP_ROW_BCAST31, 0xc, 0xf, true);
    return __builtin_amdgcn_readlane(v, 63);
}

template <int NX>
__device__ __forceinline__ double pick_input(const DevModel& md, int d, const double (&x)[NX], const double* __restrict__ ut) {
    const int s = md.sel[d];
    if (s == 0) return x[0];
    if (NX > 1 && s == 1) return x[NX > 1 ? 1 : 0];
    return ut[s - NX];
}

// log N(y_t; H x, R)   (likelihood_fcn, src/Toy_Example.py:142-144 / src/EMPS.py:250-252)
template <int NX>
__device__ __forceinline__ double loglik(const DevModel& md, const double* __restrict__ yt, const double (&xv)[NX]) {
    double e[PGAS_MAX_NY];
    double quad = 0.0;
#pragma unroll
    for (int j = 0; j < PGAS_MAX_NY; ++j) {
        e[j] = 0.0;
        if (j < md.ny) {
            e[j] = yt[j];
#pragma unroll
            for (int k = 0; k < NX; ++k) e[j] = PGAS_FMA(-md.H[j * NX + k], xv[k], e[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < PGAS_MAX_NY; ++j) {
        if (j < md.ny) {
            double w = 0.0;
#pragma unroll
            for (int l = 0; l <= j; ++l) w = PGAS_FMA(md.LRinv[j * md.ny + l], e[l], w);
            quad = PGAS_FMA(w, w, quad);
        }
    }
    return PGAS_FMA(-0.5, quad, md.cR);
}

// ---- per-dimension sines sin(pi (j0 + q step) r), q = 0.. (src/BasisFunctions.py:77-80) ------------------
// canonical recurrences:
//   D == 1 : rotation   (s,c) <- (s cd + c sd, c cd - s sd)          (stable for the 40-frequency Toy basis)
//   D >= 2 : Chebyshev  s_{q+1} = 2 cd s_q - s_{q-1},  s_{-1} = sin(pi (j0 - step) r)  (= 0 when j0 == step)
struct DimStart {
    double s0, c0, sd, cd;
};

// start / step sines of dimension d for P particles at once (batch form keeps polynomial coefficients in registers)
template <int P>
__device__ __forceinline__ void dim_start_n(const DevModel& md, int d, const double (&r)[P], DimStart (&ds)[P]) {
    double a[P], sv[P], cv[P];
#pragma unroll
    for (int p = 0; p < P; ++p) a[p] = (double)md.j0[d] * r[p];
    pgas_sincospi_n(a, sv, cv, P);
#pragma unroll
    for (int p = 0; p < P; ++p) {
        ds[p].s0 = sv[p];
        ds[p].c0 = cv[p];
        ds[p].sd = sv[p];
        ds[p].cd = cv[p];
    }
    if (md.jstep[d] != md.j0[d]) {  // uniform
#pragma unroll
        for (int p = 0; p < P; ++p) a[p] = (double)md.jstep[d] * r[p];
        pgas_sincospi_n(a, sv, cv, P);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            ds[p].sd = sv[p];
            ds[p].cd = cv[p];
        }
    }
}
__device__ __forceinline__ void rotate(double& sc, double& cc, double sd, double cd) {
    double sn = PGAS_FMA(sc, cd, cc * sd);
    double cn = PGAS_FMA(cc, cd, -(sc * sd));
    sc = sn;
    cc = cn;
}
// Chebyshev state of one dimension: cur = s_q, prev = s_{q-1}, tw = 2 cd
struct Cheb {
    double cur, prev, tw;
};
__device__ __forceinline__ Cheb cheb_init(const DevModel& md, int d, const DimStart& ds) {
    Cheb c;
    c.cur = ds.s0;
    c.prev = (md.jstep[d] == md.j0[d]) ? 0.0 : PGAS_FMA(ds.s0, ds.cd, -(ds.c0 * ds.sd));
    c.tw = ds.cd + ds.cd;
    return c;
}
__device__ __forceinline__ void cheb_next(Cheb& c) {
    const double nx = PGAS_FMA(c.tw, c.cur, -c.prev);
    c.prev = c.cur;
    c.cur = nx;
}

// ------------------------------------------------------------------------------------------
// aux = A phi(x, u_t) for P particles at once (src/PGAS.py:52-55).  The basis is separable
// (src/BasisFunctions.py:77-80), so the M products collapse into a nested contraction over the
// dense frequency grid; the coefficients are wave-uniform (scalar loads), the outer dimensions'
// sines come from a recurrence, only the innermost dimension's table lives in registers.
// ------------------------------------------------------------------------------------------
template <int NX, int D, int JIN, int P>
__device__ __forceinline__ void eval_mean(const DevModel& md, const double* __restrict__ G, const double* __restrict__ ut,
                                          const double (&x)[P][NX], double (&aux)[P][NX]) {
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int k = 0; k < NX; ++k) aux[p][k] = 0.0;
    if constexpr (D == 1) {
        double r[P];
        DimStart ds[P];
#pragma unroll
        for (int p = 0; p < P; ++p) r[p] = PGAS_FMA(pick_input<NX>(md, 0, x[p], ut), md.alpha[0], md.beta[0]);
        dim_start_n<P>(md, 0, r, ds);
        const int J0 = md.J[0];
        for (int a = 0; a < J0; ++a) {
            double g[NX];
#pragma unroll
            for (int k = 0; k < NX; ++k) g[k] = G[a * NX + k];
#pragma unroll
            for (int p = 0; p < P; ++p) {
#pragma unroll
                for (int k = 0; k < NX; ++k) aux[p][k] = PGAS_FMA(g[k], ds[p].s0, aux[p][k]);
                rotate(ds[p].s0, ds[p].c0, ds[p].sd, ds[p].cd);
            }
        }
    } else {
        // innermost dimension table
        constexpr int DI = D - 1;
        double tab[P][JIN];
        Cheb c0[P];
        {
            double rin[P], rout[P];
            DimStart din[P], dout[P];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                rin[p] = PGAS_FMA(pick_input<NX>(md, DI, x[p], ut), md.alpha[DI], md.beta[DI]);
                rout[p] = PGAS_FMA(pick_input<NX>(md, 0, x[p], ut), md.alpha[0], md.beta[0]);
            }
            dim_start_n<P>(md, DI, rin, din);
            dim_start_n<P>(md, 0, rout, dout);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                Cheb ci = cheb_init(md, DI, din[p]);
#pragma unroll
                for (int q = 0; q < JIN; ++q) {
                    tab[p][q] = ci.cur;
                    cheb_next(ci);
                }
                c0[p] = cheb_init(md, 0, dout[p]);
            }
        }
        const int J0 = md.J[0];
        if constexpr (D == 2) {
            for (int a = 0; a < J0; ++a) {
                double in[P][NX];
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int k = 0; k < NX; ++k) in[p][k] = 0.0;
                const double* __restrict__ Ga = G + (size_t)a * JIN * NX;
#pragma unroll
                for (int q = 0; q < JIN; ++q) {
#pragma unroll
                    for (int k = 0; k < NX; ++k) {
                        const double g = Ga[q * NX + k];
#pragma unroll
                        for (int p = 0; p < P; ++p) in[p][k] = PGAS_FMA(g, tab[p][q], in[p][k]);
                    }
                }
#pragma unroll
                for (int p = 0; p < P; ++p) {
#pragma unroll
                    for (int k = 0; k < NX; ++k) aux[p][k] = PGAS_FMA(c0[p].cur, in[p][k], aux[p][k]);
                    cheb_next(c0[p]);
                }
            }
        } else {
            Cheb c1s[P];
            {
                double r1[P];
                DimStart d1[P];
#pragma unroll
                for (int p = 0; p < P; ++p) r1[p] = PGAS_FMA(pick_input<NX>(md, 1, x[p], ut), md.alpha[1], md.beta[1]);
                dim_start_n<P>(md, 1, r1, d1);
#pragma unroll
                for (int p = 0; p < P; ++p) c1s[p] = cheb_init(md, 1, d1[p]);
            }
            const int J1 = md.J[1];
            for (int a = 0; a < J0; ++a) {
                double mid[P][NX];
                Cheb c1[P];
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    c1[p] = c1s[p];
#pragma unroll
                    for (int k = 0; k < NX; ++k) mid[p][k] = 0.0;
                }
                for (int b = 0; b < J1; ++b) {
                    double in[P][NX];
#pragma unroll
                    for (int p = 0; p < P; ++p)
#pragma unroll
                        for (int k = 0; k < NX; ++k) in[p][k] = 0.0;
                    const double* __restrict__ Gab = G + ((size_t)a * J1 + b) * JIN * NX;
#pragma unroll
                    for (int q = 0; q < JIN; ++q) {
#pragma unroll
                        for (int k = 0; k < NX; ++k) {
                            const double g = Gab[q * NX + k];
#pragma unroll
                            for (int p = 0; p < P; ++p) in[p][k] = PGAS_FMA(g, tab[p][q], in[p][k]);
                        }
                    }
#pragma unroll
                    for (int p = 0; p < P; ++p) {
#pragma unroll
                        for (int k = 0; k < NX; ++k) mid[p][k] = PGAS_FMA(c1[p].cur, in[p][k], mid[p][k]);
                        cheb_next(c1[p]);
                    }
                }
#pragma unroll
                for (int p = 0; p < P; ++p) {
#pragma unroll
                    for (int k = 0; k < NX; ++k) aux[p][k] = PGAS_FMA(c0[p].cur, mid[p][k], aux[p][k]);
                    cheb_next(c0[p]);
                }
            }
        }
    }
}

// sines of every frequency of dimension d for ONE point (test hook / trajectory basis): s[q], q < J[d]
template <int NX>
__device__ __forceinline__ void dim_sines_point(const DevModel& md, int d, const double (&xv)[NX], const double* __restrict__ ut,
                                                double* __restrict__ s) {
    double r[1] = {PGAS_FMA(pick_input<NX>(md, d, xv, ut), md.alpha[d], md.beta[d])};
    DimStart ds[1];
    dim_start_n<1>(md, d, r, ds);
    if (md.D == 1) {
        for (int q = 0; q < md.J[d]; ++q) {
            s[q] = ds[0].s0;
            rotate(ds[0].s0, ds[0].c0, ds[0].sd, ds[0].cd);
        }
    } else {
        Cheb c = cheb_init(md, d, ds[0]);
        for (int q = 0; q < md.J[d]; ++q) {
            s[q] = c.cur;
            cheb_next(c);
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_pack: G[pos[m]][k] = A[k][m] * nrm, zero elsewhere
// ------------------------------------------------------------------------------------------
__global__ void k_pack(const double* __restrict__ A, const int32_t* __restrict__ pos, int M, int nx, double nrm,
                       double* __restrict__ G, int64_t gtotal) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // phase split by launch: caller memsets G first
    if (i < (int64_t)M * nx) {
        const int m = (int)(i / nx), k = (int)(i % nx);
        G[(int64_t)pos[m] * nx + k] = A[(int64_t)k * M + m] * nrm;
    }
    (void)gtotal;
}

// ------------------------------------------------------------------------------------------
// k_init: x_0 = m0 + L0 z, conditioned particle last
// ------------------------------------------------------------------------------------------
template <int NX>
__global__ __launch_bounds__(PG_BLK) void k_init(DevModel md, uint64_t seed, const double* __restrict__ m0L0 /* m0[nx], L0[nx*nx] */,
                                                  const double* __restrict__ ref0, double* __restrict__ x0) {
    const int64_t p = (int64_t)blockIdx.x * PG_BLK + threadIdx.x;
    if (p >= md.N) return;
    double z[2];
    pgas_rng_normals(seed, PGAS_STREAM_INIT, 0u, (uint64_t)(md.p0 + p), NX, z);
    double xv[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        double v = m0L0[k];
#pragma unroll
        for (int l = 0; l <= k; ++l) v = PGAS_FMA(m0L0[NX + k * NX + l], z[l], v);
        xv[k] = v;
    }
    if (md.p0 + p == md.Ng - 1) {
#pragma unroll
        for (int k = 0; k < NX; ++k) xv[k] = ref0[k];
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) x0[p * NX + k] = xv[k];
}

// ------------------------------------------------------------------------------------------
// segment softmax scan shared by k_front / k_resample / k_segscan.
// lw[r] is the log-weight of particle seg*SEG + r*BLK + tid (-inf when past N).  Writes the
// quantised inclusive cumsum (index order) and the segment (max, total).
// ------------------------------------------------------------------------------------------
struct ScanSmem {
    uint64_t q[2][PGAS_SEG];
    double red[2][PG_BLK / 64];
    uint64_t wtot[2][PG_BLK / 64];
};

template <int NW, bool STORE_B = true>  // NW: weight vectors scanned together (1 or 2); STORE_B = false: the second one only leaves its
                                         // segment partial (max, total) -- its per-particle cumsum is recomputed where it is needed
__device__ __forceinline__ void segment_scan(ScanSmem& sm, const double (&lw)[NW][PG_PPT], int seg, int nsegp,
                                             uint64_t* __restrict__ cA, uint64_t* __restrict__ cB, double* __restrict__ segm,
                                             uint64_t* __restrict__ segs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double mx[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        double m = -__builtin_inf();
#pragma unroll
        for (int r = 0; r < PG_PPT; ++r) m = __builtin_fmax(m, lw[w][r]);  // fmax ignores NaN
        m = wave_max(m);
        if (lane == 0) sm.red[w][wave] = m;
    }
    __syncthreads();
    uint64_t qsum_b = 0;
    {
        double arg[NW * PG_PPT], ev[NW * PG_PPT];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            double m = sm.red[w][0];
#pragma unroll
            for (int v = 1; v < PG_BLK / 64; ++v) m = __builtin_fmax(m, sm.red[w][v]);
            const double kref = pgas_seg_ref(m);  // power-of-two reference of the segment (include/pgas_canon.h)
            mx[w] = kref;
#pragma unroll
            for (int r = 0; r < PG_PPT; ++r) arg[w * PG_PPT + r] = pgas_seg_arg(lw[w][r], kref);
        }
        pgas_exp_n(arg, ev, NW * PG_PPT);
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int r = 0; r < PG_PPT; ++r) {
                const double e = ev[w * PG_PPT + r];
                const uint64_t q = (e > 0.0) ? pgas_double_to_u64(__builtin_rint(e * PGAS_FIX_SCALE)) : 0ull;
                if (w == 0 || STORE_B) sm.q[w][r * PG_BLK + tid] = q;  // strided -> contiguous particle order for the prefix
                else qsum_b += q;                                      // only the segment total is wanted: any order will do
            }
    }
    __syncthreads();
    uint64_t loc[NW][PG_PPT], incl[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        uint64_t run = 0;
        if (w == 0 || STORE_B) {
#pragma unroll
            for (int j = 0; j < PG_PPT; ++j) {
                run += sm.q[w][PG_PPT * tid + j];
                loc[w][j] = run;
            }
        } else {
            run = qsum_b;
        }
        incl[w] = wave_incl_scan_u64(run);
        if (lane == 63) sm.wtot[w][wave] = incl[w];
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        uint64_t off = 0, tot = 0;
#pragma unroll
        for (int v = 0; v < PG_BLK / 64; ++v) {
            const uint64_t t = sm.wtot[w][v];
            if (v < wave) off += t;
            tot += t;
        }
        if (w == 0 || STORE_B) {
            const uint64_t base = off + incl[w] - loc[w][PG_PPT - 1];
            ulonglong2* dst = reinterpret_cast<ulonglong2*>((w == 0 ? cA : cB) + (size_t)seg * PGAS_SEG + PG_PPT * tid);
            dst[0] = make_ulonglong2(base + loc[w][0], base + loc[w][1]);
            dst[1] = make_ulonglong2(base + loc[w][2], base + loc[w][3]);
        }
        if (tid == 0) {
            segm[(size_t)w * nsegp + seg] = mx[w];
            segs[(size_t)w * nsegp + seg] = tot;
        }
    }
}

// ------------------------------------------------------------------------------------------
// per-particle part of a step for the PG_PPT particles of this thread (src/PGAS.py:90-100,109-116,130-134):
//   la = log p(y_t | aux),  h = log N(ref_t; aux, S),  x_new = aux + L_S z  (conditioned particle = ref_t).
// Reads only the particle's own state: in the reference the propagation does NOT depend on the
// resampled ancestors (quirk Q1), which is what lets the sweep run this part for all time steps
// without any synchronisation (k_propagate).
// ------------------------------------------------------------------------------------------
template <int NX, int D, int JIN, int P>
__device__ __forceinline__ void propagate_particles(const DevModel& md, const TransParams& tp, int t, uint64_t seed,
                                                    const double* __restrict__ ref_t, int seg, const double (&xprev)[PG_PPT][NX],
                                                    double (&xnew)[PG_PPT][NX], double (&la)[PG_PPT], double (&h)[PG_PPT],
                                                    double (&aux)[PG_PPT][NX]) {
    const int tid = threadIdx.x;
    const double* __restrict__ yt = md.y + (size_t)t * md.ny;
    const double* __restrict__ ut = md.u + (size_t)t * md.nu;
    double rf[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) rf[k] = ref_t[k];
#pragma unroll
    for (int r0 = 0; r0 < PG_PPT; r0 += P) {
        double xin[P][NX], ax[P][NX];
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int k = 0; k < NX; ++k) xin[p][k] = xprev[r0 + p][k];
        eval_mean<NX, D, JIN, P>(md, tp.G, ut, xin, ax);
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int k = 0; k < NX; ++k) aux[r0 + p][k] = ax[p][k];
    }
    // propagation noise for the four particles at once (Philox -> Box-Muller)
    double z0[PG_PPT], z1[PG_PPT];
    {
        pgas_u32x4 w[PG_PPT];
#pragma unroll
        for (int r = 0; r < PG_PPT; ++r) {
            const int64_t pi = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
            w[r] = pgas_rng_block(seed, PGAS_STREAM_PROP, 0u, (uint32_t)t, (uint64_t)(md.p0 + pi));
        }
        pgas_normal_pair_n(w, z0, z1, PG_PPT);
    }
#pragma unroll
    for (int r = 0; r < PG_PPT; ++r) {
        const int64_t pi = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
        la[r] = loglik<NX>(md, yt, aux[r]);
        double quad = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            double w = 0.0;
#pragma unroll
            for (int l = 0; l <= k; ++l) w = PGAS_FMA(tp.LSinv[k * NX + l], rf[l] - aux[r][l], w);
            quad = PGAS_FMA(w, w, quad);
        }
        h[r] = PGAS_FMA(-0.5, quad, tp.cS);
        const double z[2] = {z0[r], z1[r]};
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            double v = aux[r][k];
#pragma unroll
            for (int l = 0; l <= k; ++l) v = PGAS_FMA(tp.LS[k * NX + l], z[l], v);
            xnew[r][k] = (md.p0 + pi == md.Ng - 1) ? rf[k] : v;
        }
    }
}

template <int NX>
__device__ __forceinline__ void store_particles(const DevModel& md, double* __restrict__ x, int seg, const double (&xv)[PG_PPT][NX]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int r = 0; r < PG_PPT; ++r) {
        const int64_t pi = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
        if (pi < md.N) {
            if constexpr (NX == 2) {
                reinterpret_cast<double2*>(x)[pi] = make_double2(xv[r][0], xv[r][1]);
            } else {
#pragma unroll
                for (int k = 0; k < NX; ++k) x[pi * NX + k] = xv[r][k];
            }
        }
    }
}

// front half of one step as pgas_step needs it: propagate, store, log-weights of both softmaxes
template <int NX, int D, int JIN, int P>
__device__ __forceinline__ void front_particles(const DevModel& md, const TransParams& tp, int t, uint64_t seed,
                                                const double* __restrict__ ref_t, int seg, const double (&xprev)[PG_PPT][NX],
                                                const double (&logw)[PG_PPT], int corrected, double* __restrict__ x_new,
                                                double* __restrict__ laux_out, double (&lw)[2][PG_PPT]) {
    const int tid = threadIdx.x;
    double xn[PG_PPT][NX], la[PG_PPT], h[PG_PPT], aux[PG_PPT][NX];
    propagate_particles<NX, D, JIN, P>(md, tp, t, seed, ref_t, seg, xprev, xn, la, h, aux);
    // corrected mode (resample before propagate): hand the transition means to k_back_corrected instead of new states
    if (corrected) store_particles<NX>(md, x_new, seg, aux);
    else store_particles<NX>(md, x_new, seg, xn);
#pragma unroll
    for (int r = 0; r < PG_PPT; ++r) {
        const int64_t pi = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
        const bool valid = pi < md.N;
        if (valid) laux_out[pi] = la[r];
        const double l1 = la[r] + logw[r];
        lw[0][r] = valid ? l1 : -__builtin_inf();
        lw[1][r] = valid ? l1 + h[r] : -__builtin_inf();
    }
}

template <int NX>
__device__ __forceinline__ void load_particles(const DevModel& md, const double* __restrict__ x, int seg, double (&xv)[PG_PPT][NX]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int r = 0; r < PG_PPT; ++r) {
        int64_t pi = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
        if (pi >= md.N) pi = md.N - 1;
        if constexpr (NX == 2) {
            const double2 v = reinterpret_cast<const double2*>(x)[pi];
            xv[r][0] = v.x;
            xv[r][1] = v.y;
        } else {
#pragma unroll
            for (int k = 0; k < NX; ++k) xv[r][k] = x[pi * NX + k];
        }
    }
}

// k_front: one workgroup per segment.
template <int NX, int D, int JIN, int P>
__global__ __launch_bounds__(PG_BLK) void k_front(DevModel md, TransParams tp, int t, uint64_t seed,
                                                   const double* __restrict__ x_prev, const double* __restrict__ logw_prev,
                                                   const double* __restrict__ ref_t, int corrected, double* __restrict__ x_new, ScanBufs sb) {
    __shared__ ScanSmem sm;
    const int seg = blockIdx.x, tid = threadIdx.x;
    double xv[PG_PPT][NX], lwp[PG_PPT], lw[2][PG_PPT];
    load_particles<NX>(md, x_prev, seg, xv);
#pragma unroll
    for (int r = 0; r < PG_PPT; ++r) {
        const int64_t pi = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
        lwp[r] = (logw_prev != nullptr && pi < md.N) ? logw_prev[pi] : 0.0;
    }
    front_particles<NX, D, JIN, P>(md, tp, t, seed, ref_t, seg, xv, lwp, corrected, x_new, sb.laux, lw);
    segment_scan<2>(sm, lw, seg, sb.nsegp, sb.c1, sb.c2, sb.segm_w, sb.segs_w);
}

// ------------------------------------------------------------------------------------------
// k_upper: cross-segment CDF.  Block w (0: resampling weights, 1: ancestor weights) turns the
// segment (max, total) pairs into exclusive prefixes in the canonical KS64-tree order, the
// running maximum of the segment-end numerators and the normaliser S; block `search_block`
// additionally counts #{k : W_k < u S} (ancestor of the reference particle, src/PGAS.py:121-124;
// or the final index, :225).
//
// Latency is what matters here (one small workgroup per CDF on the sweep's critical path), so
// the kernel is organised around few workgroup barriers: wave v owns the level-0 groups
// g = v, v + 4, ... (64 consecutive segments each, Kogge-Stone by shuffles), the level-1/2 scans
// are tiny and recomputed by every wave from LDS, and the running maximum needs one more barrier.
// ------------------------------------------------------------------------------------------
#define PG_UPPER_WAVES (PG_UPPER_THREADS / 64)
#define PG_MAX_GROUPS (PG_MAX_NSEG / 64)
template <int MG>  // capacity in level-0 groups
struct UpperSmemT {
    double ga[2][MG];                // level-0 group totals (up to two CDFs scanned together)
    double gmax[2][MG];              // per-group maximum of the segment-end numerators
    double red[2][PG_UPPER_WAVES];
    double par[3];                   // (excl, scale, carry) of one segment, broadcast for cdf_count_block
    int cnt[2];
    unsigned long long wsum[PG_UPPER_WAVES];  // wave totals of cdf_count_block_recompute's integer scan
};
typedef UpperSmemT<PG_MAX_GROUPS> UpperSmem;

// Cross-segment scan of NC (1 or 2) CDFs by a 256-thread workgroup, sharing the three workgroup barriers.
// Thread (wave v, lane l) owns segments b = ((v + 4 e) << 6) + l, e < GPW, and gets their exclusive prefix, scale
// and running maximum in registers; S[c] is the normaliser of CDF c.  CDF c reads segm/segs + c * stride.
template <int GPW, int NC, class SM>  // level-0 groups (of 64 segments) per wave: nseg <= 64 * PG_UPPER_WAVES * GPW
__device__ __forceinline__ void upper_core(SM& sm, const double* __restrict__ segm, const uint64_t* __restrict__ segs, int stride,
                                           int nseg, double (&excl)[NC][GPW], double (&scale)[NC][GPW], double (&cmx)[NC][GPW],
                                           double (&S)[NC], int nseg_l = 0x40000000, int rank_stride = 0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n1 = (nseg + 63) >> 6;  // level-0 groups
    const int n2 = (n1 + 63) >> 6;    // <= 2 for nseg <= 8192

    // 1. global max of the segment maxima (group g of wave v: g = v + PG_UPPER_WAVES * e)
    double mv[NC][GPW];
    uint64_t sv[NC][GPW];
    double g[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        g[c] = -__builtin_inf();
#pragma unroll
        for (int e = 0; e < GPW; ++e) {
            const int b = ((wave + PG_UPPER_WAVES * e) << 6) + lane;
            mv[c][e] = -__builtin_inf();
            sv[c][e] = 0;
            if (b < nseg) {
                const size_t at = (size_t)c * stride + (size_t)(b / nseg_l) * rank_stride + (b % nseg_l);  // (rank, cdf, local segment)
                mv[c][e] = segm[at];
                sv[c][e] = segs[at];
            }
            g[c] = __builtin_fmax(g[c], mv[c][e]);
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        g[c] = wave_max(g[c]);
        if (lane == 0) sm.red[c][wave] = g[c];
    }
    if (tid < 2) sm.cnt[tid] = 0;
    PG_STAMP(8);
    __syncthreads();
    PG_STAMP(9);
    // 2. scaled totals and level-0 Kogge-Stone scans
    double tot[NC][GPW], incA[NC][GPW];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        g[c] = sm.red[c][0];
#pragma unroll
        for (int v = 1; v < PG_UPPER_WAVES; ++v) g[c] = __builtin_fmax(g[c], sm.red[c][v]);
        double ev[GPW];
#pragma unroll
        for (int e = 0; e < GPW; ++e) ev[e] = pgas_seg_scale(mv[c][e], g[c]);  // exact power of two: no exp across segments
#pragma unroll
        for (int e = 0; e < GPW; ++e) {
            const int gi = wave + PG_UPPER_WAVES * e;
            const int b = (gi << 6) + lane;
            tot[c][e] = 0.0;
            incA[c][e] = 0.0;
            scale[c][e] = 0.0;
            if (gi < n1) {  // wave-uniform
                double sc = ev[e];
                if (!(sc >= 0.0)) sc = 0.0;
                if (b < nseg) {
                    scale[c][e] = sc;
                    tot[c][e] = sc * (pgas_u64_to_double(sv[c][e]) * PGAS_FIX_INV);
                }
                incA[c][e] = wave_scan_add(tot[c][e]);
                if (lane == 63) sm.ga[c][gi] = incA[c][e];
            }
        }
    }
    PG_STAMP(10);
    __syncthreads();
    PG_STAMP(11);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        // 3. levels 1 and 2, recomputed by every wave: incB over the group totals (groups of 64), incC over those
        double incB0 = wave_scan_add(lane < n1 ? sm.ga[c][lane] : 0.0);                            // level-1 group 0
        double incB1 = n2 > 1 ? wave_scan_add(64 + lane < n1 ? sm.ga[c][64 + lane] : 0.0) : 0.0;   // level-1 group 1
        const double GB0 = readlane_f64(incB0, 63);  // level 2 (zero padded): incC[0] = GB0
        // 4. exclusive prefixes, segment-end numerators, per-group running maxima
#pragma unroll
        for (int e = 0; e < GPW; ++e) {
            const int gi = wave + PG_UPPER_WAVES * e;
            const int b = (gi << 6) + lane;
            const int h = gi >> 6, l1 = gi & 63;
            const double eC = h ? GB0 : 0.0;
            const double srcB = h ? incB1 : incB0;
            const double prevB = readlane_f64(srcB, l1 ? l1 - 1 : 0);
            const double eB = l1 ? prevB : 0.0;
            const double exB = eC + eB;
            double wend = 0.0;
            cmx[c][e] = 0.0;
            excl[c][e] = 0.0;
            if (gi < n1) {
                const double up = __shfl_up(incA[c][e], 1);
                const double eA = lane ? up : 0.0;
                const double ex = exB + eA;
                if (b < nseg) {
                    excl[c][e] = ex;
                    wend = ex + tot[c][e];
                }
                cmx[c][e] = wave_scan_max(wend);
                if (lane == 63) sm.gmax[c][gi] = cmx[c][e];
            }
        }
    }
    PG_STAMP(12);
    __syncthreads();
    PG_STAMP(13);
    // 5. running maximum across groups (exact, any order): carry of group gi = max of gmax[0..gi)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        double m0 = wave_scan_max(lane < n1 ? sm.gmax[c][lane] : 0.0);
        double m1 = n2 > 1 ? wave_scan_max(64 + lane < n1 ? sm.gmax[c][64 + lane] : 0.0) : 0.0;
        const double M0 = readlane_f64(m0, 63), M1 = readlane_f64(m1, 63);
#pragma unroll
        for (int e = 0; e < GPW; ++e) {
            const int gi = wave + PG_UPPER_WAVES * e;
            if (gi < n1) {
                const int h = gi >> 6, l1 = gi & 63;
                const double src = h ? m1 : m0;
                const double prev = readlane_f64(src, l1 ? l1 - 1 : 0);
                double carry = l1 ? prev : 0.0;
                if (h) carry = __builtin_fmax(carry, M0);
                cmx[c][e] = __builtin_fmax(cmx[c][e], carry);
            }
        }
        S[c] = __builtin_fmax(M0, M1);
    }
}

// #{k : W_k < tau} for one CDF whose cross-segment scan sits in registers -- the searchsorted of
// src/PGAS.py:122-124 / :225.  Three workgroup barriers; sm.cnt must be zero on entry (upper_core leaves it so).
template <int GPW, class SM>
__device__ __forceinline__ int cdf_count_block(SM& sm, const double (&ex)[GPW], const double (&sc)[GPW], const double (&cmx)[GPW],
                                               int nseg, int N, double tau, const uint64_t* __restrict__ cbuf_local,
                                               const uint64_t* const* cbuf_ranks = nullptr, int nseg_l = 0x40000000) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int c = 0;
#pragma unroll
    for (int e = 0; e < GPW; ++e) {
        const int b = ((wave + PG_UPPER_WAVES * e) << 6) + lane;
        c += (b < nseg && cmx[e] < tau) ? 1 : 0;
    }
    c = wave_sum_i(c);
    if (lane == 0 && c) atomicAdd(&sm.cnt[0], c);
    if (tid == 0) sm.par[2] = 0.0;
    __syncthreads();
    const int bs = sm.cnt[0];
    if (bs >= nseg) return N - 1;
#pragma unroll
    for (int e = 0; e < GPW; ++e) {  // the owners of segments bs and bs-1 publish (excl, scale) and the carry
        const int b = ((wave + PG_UPPER_WAVES * e) << 6) + lane;
        if (b == bs) {
            sm.par[0] = ex[e];
            sm.par[1] = sc[e];
        }
        if (b == bs - 1) sm.par[2] = cmx[e];
    }
    __syncthreads();
    const int64_t base = (int64_t)bs * PGAS_SEG;
    const int n = (N - base) < PGAS_SEG ? (int)(N - base) : PGAS_SEG;
    const double e0 = sm.par[0], s0 = sm.par[1], cy = sm.par[2];
    const uint64_t* __restrict__ cseg = cbuf_ranks ? cbuf_ranks[bs / nseg_l] + (size_t)(bs % nseg_l) * PGAS_SEG : cbuf_local + base;
    int k = 0;
    for (int i = tid; i < n; i += PG_UPPER_THREADS) {
        double num = e0 + s0 * (pgas_u64_to_double(cseg[i]) * PGAS_FIX_INV);
        num = __builtin_fmax(num, cy);
        k += (num < tau) ? 1 : 0;
    }
    k = wave_sum_i(k);
    if (lane == 0 && k) atomicAdd(&sm.cnt[1], k);
    __syncthreads();
    const int64_t r = base + sm.cnt[1];
    return r > N - 1 ? N - 1 : (int)r;
}

// The ancestor CDF of a step (src/PGAS.py:117-124) is read in ONE segment only: the one the reference particle's uniform falls
// into.  k_resample_fast therefore never stores its per-particle cumsum; the workgroup that draws the ancestor rebuilds that
// segment from what the sweep already keeps in HBM -- lw2_i = (la_s[i] + logw_{s-1}[i]) + h_s[i] with
// logw_{s-1}[i] = ln_{s-1}[i] - la_{s-1}[a_{s-1}[i]] (0 for s = 1), the same expressions, the segment's stored reference k,
// the same fixed-point numerators and an exact integer cumsum -- and counts against it.  Bit-identical to the stored version.
struct AncInputs {
    const double* la_s;       // (np) log p(y_s | aux_s)
    const double* h_s;        // (np) log N(ref_s; aux_s, S)
    const double* ln_p;       // (np) log p(y_{s-1} | x_{s-1}), or nullptr for s = 1
    const double* la_p;       // (np) log p(y_{s-1} | aux_{s-1})
    const int32_t* anc_p;     // (N)  ancestors of step s-1
    const double* kref;       // (nseg) segment references of the ancestor CDF of step s
};
template <int GPW, class SM>
__device__ __forceinline__ int cdf_count_block_recompute(SM& sm, const double (&ex)[GPW], const double (&sc)[GPW], const double (&cmx)[GPW],
                                                         int nseg, int N, double tau, const AncInputs& in) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int c = 0;
#pragma unroll
    for (int e = 0; e < GPW; ++e) {
        const int b = ((wave + PG_UPPER_WAVES * e) << 6) + lane;
        c += (b < nseg && cmx[e] < tau) ? 1 : 0;
    }
    c = wave_sum_i(c);
    if (lane == 0 && c) atomicAdd(&sm.cnt[0], c);
    if (tid == 0) sm.par[2] = 0.0;
    __syncthreads();
    const int bs = sm.cnt[0];
    if (bs >= nseg) return N - 1;
#pragma unroll
    for (int e = 0; e < GPW; ++e) {
        const int b = ((wave + PG_UPPER_WAVES * e) << 6) + lane;
        if (b == bs) {
            sm.par[0] = ex[e];
            sm.par[1] = sc[e];
        }
        if (b == bs - 1) sm.par[2] = cmx[e];
    }
    const int64_t base = (int64_t)bs * PGAS_SEG;
    const int n = (N - base) < PGAS_SEG ? (int)(N - base) : PGAS_SEG;
    const double kref = in.kref[bs];
    double arg[PG_PPT], ev[PG_PPT];
#pragma unroll
    for (int j = 0; j < PG_PPT; ++j) {
        const int i = PG_PPT * tid + j;
        double lw2 = -__builtin_inf();
        if (i < n) {
            const int64_t gi = base + i;
            double logw = 0.0;
            if (in.ln_p) logw = in.ln_p[gi] - in.la_p[in.anc_p[gi]];
            const double l1 = in.la_s[gi] + logw;
            lw2 = l1 + in.h_s[gi];
        }
        arg[j] = pgas_seg_arg(lw2, kref);
    }
    pgas_exp_n(arg, ev, PG_PPT);
    uint64_t loc[PG_PPT], run = 0;
#pragma unroll
    for (int j = 0; j < PG_PPT; ++j) {
        run += (ev[j] > 0.0) ? pgas_double_to_u64(__builtin_rint(ev[j] * PGAS_FIX_SCALE)) : 0ull;
        loc[j] = run;
    }
    const uint64_t incl = wave_incl_scan_u64(run);
    if (lane == 63) sm.wsum[wave] = incl;
    __syncthreads();
    uint64_t off = incl - run;
#pragma unroll
    for (int v = 0; v < PG_UPPER_WAVES; ++v)
        if (v < wave) off += sm.wsum[v];
    const double e0 = sm.par[0], s0 = sm.par[1], cy = sm.par[2];
    int k = 0;
#pragma unroll
    for (int j = 0; j < PG_PPT; ++j) {
        double num = e0 + s0 * (pgas_u64_to_double(off + loc[j]) * PGAS_FIX_INV);
        num = __builtin_fmax(num, cy);
        k += (PG_PPT * tid + j < n && num < tau) ? 1 : 0;
    }
    k = wave_sum_i(k);
    if (lane == 0 && k) atomicAdd(&sm.cnt[1], k);
    __syncthreads();
    const int64_t r = base + sm.cnt[1];
    return r > N - 1 ? N - 1 : (int)r;
}

template <int GPW>
__global__ __launch_bounds__(PG_UPPER_THREADS) void k_upper(int N, int nseg, ScanBufs sb, Peers pr, int search_block, double u_search,
                                                             int final_mode) {
    // N, nseg: GLOBAL particle / segment counts (every rank of a sharded sweep runs this kernel on the gathered partials
    // and gets identical results)
    __shared__ UpperSmem sm;
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* __restrict__ excl = sb.excl + (size_t)w * sb.nsegp_g;
    double* __restrict__ scale = sb.scale + (size_t)w * sb.nsegp_g;
    double* __restrict__ cm = sb.cm + (size_t)w * sb.nsegp_g;
    double ex[1][GPW], sc[1][GPW], cmx[1][GPW], Sv[1];
    upper_core<GPW, 1>(sm, sb.segm + (size_t)w * sb.nsegp, sb.segs + (size_t)w * sb.nsegp, 0, nseg, ex, sc, cmx, Sv, sb.nseg_l, sb.rank_stride);
    const double S = Sv[0];
#pragma unroll
    for (int e = 0; e < GPW; ++e) {
        const int b = ((wave + PG_UPPER_WAVES * e) << 6) + lane;
        if (b < nseg) {
            excl[b] = ex[0][e];
            scale[b] = sc[0][e];
            cm[b] = cmx[0][e];
        }
    }
    const bool valid = (S > 0.0) && (S < __builtin_inf());
    if (tid == 0) {
        sb.hdr->S[w] = S;
        sb.hdr->valid[w] = valid ? 1 : 0;
    }
    if (w != search_block) return;
    int result = N - 1;
    if (valid)
        result = cdf_count_block<GPW>(sm, ex[0], sc[0], cmx[0], nseg, N, u_search * S, w == 0 ? sb.c1 : sb.c2,
                                      pr.world > 1 ? (w == 0 ? pr.c1 : pr.c2) : nullptr, pr.nseg_l);
    if (tid == 0) {
        if (final_mode)
            sb.hdr->final_idx = result;
        else
            sb.hdr->ref_idx = result;
    }
}

// ------------------------------------------------------------------------------------------
// back half: systematic resampling search (src/Filtering.py:28-35) + weight update
// (src/PGAS.py:137-147).
//
// Slot i = seg*SEG + 4*tid + j (four CONSECUTIVE slots per thread for the search; the thresholds
// tau_i = U_i * S grow with i).  The workgroup finds the range of source segments its slots fall
// into (two counts over the running-max array cm), stages the CDF numerators of up to
// PG_STAGE segments at a time in LDS as doubles -- num_k = max(carry, excl + scale * c_k 2^-51),
// evaluated once per source -- and every slot does a lower-bound search with plain double
// compares.  Ancestors then go through LDS to the particle-major (strided) layout the front
// half uses.
// ------------------------------------------------------------------------------------------
#define PG_STAGE 3
struct BackSmem {
    double num[PG_STAGE][PGAS_SEG];
    int a[PGAS_SEG];
    int red[2][PG_BLK / 64];
};

// Resampling slot j of thread tid inside its workgroup's 1024 slots: lanes of a wave take consecutive slots (for every j),
// so that the lower-bound probes of a wave fall on consecutive LDS words (no bank conflicts).
__device__ __forceinline__ int slot_of(int tid, int j) { return ((tid >> 6) << 8) + (j << 6) + (tid & 63); }

__device__ __forceinline__ double slot_tau(double u1, int64_t i, int N, double invN, bool pow2, double S) {
    const double x = u1 + (double)i;
    const double Ui = pow2 ? x * invN : x / (double)N;  // exact either way when N is a power of two
    return Ui * S;
}

// c1 data of GLOBAL source segment bs (this device's buffer, or a peer's through its xGMI mapping)
__device__ __forceinline__ const uint64_t* c1_segment(const ScanBufs& sb, const Peers& pr, int bs) {
    return pr.world > 1 ? pr.c1[bs / pr.nseg_l] + (size_t)(bs % pr.nseg_l) * PGAS_SEG : sb.c1 + (size_t)bs * PGAS_SEG;
}

// Search for the 1024 slots of local segment `seg`.  All indices that leave this function are GLOBAL particle indices;
// md.p0 / md.Ng / md.nseg_g place the device's shard in the global particle range (single device: 0 / N / nseg).
__device__ __forceinline__ void resample_slots(const DevModel& md, BackSmem& sm, double u1, const ScanBufs& sb, const Peers& pr, int seg,
                                               int32_t* __restrict__ anc_out, int (&anc_pm)[PG_PPT], bool conditioned = true) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nseg = md.nseg_g, N = md.Ng;
    const double S = sb.hdr->S[0];
    const bool valid = sb.hdr->valid[0] != 0;
    const bool pow2 = (N & (N - 1)) == 0;
    const double invN = 1.0 / (double)N;
    const double* __restrict__ cm = sb.cm;
    const int64_t loc_i = (int64_t)seg * PGAS_SEG;    // local index of slot 0 of this workgroup
    const int64_t base_i = md.p0 + loc_i;             // its global index
    const int nslots = (md.N - loc_i) < PGAS_SEG ? (int)(md.N - loc_i) : PGAS_SEG;

    double tau[PG_PPT];
    int a[PG_PPT];
#pragma unroll
    for (int j = 0; j < PG_PPT; ++j) {
        const int64_t i = base_i + slot_of(tid, j);
        tau[j] = slot_tau(u1, i, N, invN, pow2, S);
        a[j] = valid ? N - 1 : (int)(i < N ? i : N - 1);
    }
    if (valid) {  // uniform
        // source-segment range of this workgroup: b_lo = #{b: cm[b] < tau_first}, b_hi = #{b: cm[b] < tau_last}
        const double tau_first = slot_tau(u1, base_i, N, invN, pow2, S);
        const double tau_last = slot_tau(u1, base_i + nslots - 1, N, invN, pow2, S);
        int clo = 0, chi = 0;
        for (int b = tid; b < nseg; b += PG_BLK) {
            const double v = cm[b];
            clo += (v < tau_first) ? 1 : 0;
            chi += (v < tau_last) ? 1 : 0;
        }
        clo = wave_sum_i(clo);
        chi = wave_sum_i(chi);
        if (lane == 0) {
            sm.red[0][wave] = clo;
            sm.red[1][wave] = chi;
        }
        __syncthreads();
        int b_lo = 0, b_hi = 0;
#pragma unroll
        for (int v = 0; v < PG_BLK / 64; ++v) {
            b_lo += sm.red[0][v];
            b_hi += sm.red[1][v];
        }
        if (b_hi > nseg - 1) b_hi = nseg - 1;  // slots beyond the last segment keep a = N-1
        // non-empty segments of [b_lo, b_hi] (running max moved), found by bisection jumps over cm: the next one after
        // carry c is #{b : cm[b] <= c}.  Bounded work however long the run of empty segments is.
        int sb_idx[PG_STAGE];
        double g_carry = 0.0;
        int ns = 0, b = b_lo;
        while (ns <= PG_STAGE) {
            const double c0 = b ? cm[b - 1] : 0.0;
            int lo = b, hi = nseg;  // first index >= b with cm > c0
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cm[mid] <= c0) lo = mid + 1; else hi = mid;
            }
            if (lo > b_hi) break;
            if (ns == 0) g_carry = c0;
            if (ns < PG_STAGE) sb_idx[ns] = lo;
            ++ns;
            b = lo + 1;
        }
        if (ns > 0 && ns <= PG_STAGE) {
            // ---- common case: the workgroup's slots fall into at most PG_STAGE source segments.  Stage their
            // numerators back to back; they are non-decreasing across the whole window (running-max carry),
            // so one branch-free lower-bound search per slot settles every slot.
#pragma unroll
            for (int g = 0; g < PG_STAGE; ++g) {
                double4 v = make_double4(__builtin_inf(), __builtin_inf(), __builtin_inf(), __builtin_inf());
                if (g < ns) {
                    const int bs = sb_idx[g];
                    const double ex = sb.excl[bs], sc = sb.scale[bs], cy = bs ? cm[bs - 1] : 0.0;
                    const int64_t base_k = (int64_t)bs * PGAS_SEG;
                    const int n = (N - base_k) < PGAS_SEG ? (int)(N - base_k) : PGAS_SEG;
                    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(c1_segment(sb, pr, bs)) + 2 * tid;
                    const ulonglong2 c01 = src[0], c23 = src[1];
                    const int k0 = PG_PPT * tid;
                    if (k0 + 0 < n) v.x = __builtin_fmax(ex + sc * (pgas_u64_to_double(c01.x) * PGAS_FIX_INV), cy);
                    if (k0 + 1 < n) v.y = __builtin_fmax(ex + sc * (pgas_u64_to_double(c01.y) * PGAS_FIX_INV), cy);
                    if (k0 + 2 < n) v.z = __builtin_fmax(ex + sc * (pgas_u64_to_double(c23.x) * PGAS_FIX_INV), cy);
                    if (k0 + 3 < n) v.w = __builtin_fmax(ex + sc * (pgas_u64_to_double(c23.y) * PGAS_FIX_INV), cy);
                }
                reinterpret_cast<double4*>(sm.num[g])[tid] = v;
            }
            __syncthreads();
            const double* __restrict__ num = &sm.num[0][0];
            int pos[PG_PPT] = {0, 0, 0, 0};
#pragma unroll
            for (int step = 2048; step >= 1; step >>= 1) {
#pragma unroll
                for (int j = 0; j < PG_PPT; ++j) {  // loads are unconditional so the four chains advance in lock step
                    const int q = pos[j] + step;
                    const int qc = q <= PG_STAGE * PGAS_SEG ? q : PG_STAGE * PGAS_SEG;
                    const double v = num[qc - 1];
                    pos[j] = (q <= PG_STAGE * PGAS_SEG && v < tau[j]) ? q : pos[j];
                }
            }
#pragma unroll
            for (int j = 0; j < PG_PPT; ++j) {
                if (g_carry < tau[j] && pos[j] < ns * PGAS_SEG) {
                    const int g = pos[j] >> 10, off = pos[j] & (PGAS_SEG - 1);
                    const int64_t ai = (int64_t)(g == 0 ? sb_idx[0] : g == 1 ? sb_idx[1] : sb_idx[2]) * PGAS_SEG + off;
                    a[j] = ai > N - 1 ? N - 1 : (int)ai;
                }
            }
        } else if (ns > PG_STAGE) {
            // ---- degenerate weights: the slots of this workgroup spread over many source segments.  Staging them
            // all would make this workgroup the straggler of the launch, so every slot searches for itself:
            // segment by bisection over cm (global, cache resident), then bisection inside the segment.
            int sbi[PG_PPT];
#pragma unroll
            for (int j = 0; j < PG_PPT; ++j) {
                int lo = b_lo, hi = b_hi + 1;  // b_i = #{b : cm[b] < tau}
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (cm[mid] < tau[j]) lo = mid + 1; else hi = mid;
                }
                sbi[j] = lo;
            }
#pragma unroll
            for (int j = 0; j < PG_PPT; ++j) {
                const int bs = sbi[j];
                if (bs < nseg) {
                    const double ex = sb.excl[bs], sc = sb.scale[bs], cy = bs ? cm[bs - 1] : 0.0;
                    const int64_t base_k = (int64_t)bs * PGAS_SEG;
                    const int n = (N - base_k) < PGAS_SEG ? (int)(N - base_k) : PGAS_SEG;
                    const uint64_t* __restrict__ c = c1_segment(sb, pr, bs);
                    int lo = 0, hi = n;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        const double v = __builtin_fmax(ex + sc * (pgas_u64_to_double(c[mid]) * PGAS_FIX_INV), cy);
                        if (v < tau[j]) lo = mid + 1; else hi = mid;
                    }
                    const int64_t ai = base_k + lo;
                    a[j] = ai > N - 1 ? N - 1 : (int)ai;
                }
            }
        }
    }
    // slot-major -> particle-major through LDS; the conditioned particle takes the ancestor drawn by k_upper
#pragma unroll
    for (int j = 0; j < PG_PPT; ++j) {
        const int64_t i = base_i + slot_of(tid, j);
        if (conditioned && i == N - 1) a[j] = sb.hdr->ref_idx;  // src/PGAS.py:127
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PG_PPT; ++j) {
        sm.a[slot_of(tid, j)] = a[j];
        if (loc_i + slot_of(tid, j) < md.N) anc_out[loc_i + slot_of(tid, j)] = a[j];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PG_PPT; ++r) anc_pm[r] = sm.a[r * PG_BLK + tid];  // ancestor of local particle loc_i + r*BLK + tid
}

template <int NX>
__device__ __forceinline__ void back_slots(const DevModel& md, BackSmem& sm, int t, double u1, const ScanBufs& sb, int seg,
                                           const double (&xcur)[PG_PPT][NX], int32_t* __restrict__ anc_out, double (&logw_new)[PG_PPT]) {
    const int tid = threadIdx.x;
    int anc[PG_PPT];
    Peers none;
    none.world = 1;
    resample_slots(md, sm, u1, sb, none, seg, anc_out, anc);
    const double* __restrict__ yt = md.y + (size_t)t * md.ny;
#pragma unroll
    for (int r = 0; r < PG_PPT; ++r) {
        const int64_t i = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
        logw_new[r] = 0.0;
        if (i < md.N) logw_new[r] = loglik<NX>(md, yt, xcur[r]) - sb.laux[anc[r]];
    }
}

template <int NX>
__global__ __launch_bounds__(PG_BLK) void k_back(DevModel md, int t, double u1, const double* __restrict__ x_cur, ScanBufs sb,
                                                  int32_t* __restrict__ anc_out, double* __restrict__ logw_out) {
    __shared__ BackSmem sm;
    const int seg = blockIdx.x, tid = threadIdx.x;
    double xv[PG_PPT][NX], lwn[PG_PPT];
    load_particles<NX>(md, x_cur, seg, xv);
    back_slots<NX>(md, sm, t, u1, sb, seg, xv, anc_out, lwn);
#pragma unroll
    for (int r = 0; r < PG_PPT; ++r) {
        const int64_t i = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
        if (i < md.N) logw_out[i] = lwn[r];
    }
}

// k_back_corrected: the second half of a step in the CORRECTED mode (resample_before_propagate; quirk Q1 removed):
//   a = systematic resampling search,  x_new_i = aux[a_i] + L_S z_i  (conditioned particle = ref_t),
//   logw_new_i = log p(y_t | x_new_i) - l_aux[a_i].
// aux holds the transition means k_front stored; the noise z_i is the same Philox draw the default mode uses for
// particle i at time t, so the two modes differ only in which mean the noise is added to.
template <int NX>
__global__ __launch_bounds__(PG_BLK) void k_back_corrected(DevModel md, TransParams tp, int t, uint64_t seed, double u1,
                                                            const double* __restrict__ aux, const double* __restrict__ ref_t, ScanBufs sb,
                                                            int32_t* __restrict__ anc_out, double* __restrict__ x_new,
                                                            double* __restrict__ logw_out) {
    __shared__ BackSmem sm;
    const int seg = blockIdx.x, tid = threadIdx.x;
    int anc[PG_PPT];
    Peers none;
    none.world = 1;
    resample_slots(md, sm, u1, sb, none, seg, anc_out, anc);
    const double* __restrict__ yt = md.y + (size_t)t * md.ny;
    double z0[PG_PPT], z1[PG_PPT];
    {
        pgas_u32x4 w[PG_PPT];
#pragma unroll
        for (int r = 0; r < PG_PPT; ++r) {
            const int64_t pi = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
            w[r] = pgas_rng_block(seed, PGAS_STREAM_PROP, 0u, (uint32_t)t, (uint64_t)(md.p0 + pi));
        }
        pgas_normal_pair_n(w, z0, z1, PG_PPT);
    }
    double xn[PG_PPT][NX];
#pragma unroll
    for (int r = 0; r < PG_PPT; ++r) {
        const int64_t pi = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
        const int src = pi < md.N ? anc[r] : 0;
        const double z[2] = {z0[r], z1[r]};
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            double v = aux[(size_t)src * NX + k];
#pragma unroll
            for (int l = 0; l <= k; ++l) v = PGAS_FMA(tp.LS[k * NX + l], z[l], v);
            xn[r][k] = (md.p0 + pi == md.Ng - 1) ? ref_t[k] : v;
        }
        if (pi < md.N) logw_out[pi] = loglik<NX>(md, yt, xn[r]) - sb.laux[src];
    }
    store_particles<NX>(md, x_new, seg, xn);
}

// ------------------------------------------------------------------------------------------
// The sweep (condSequentialMonteCarlo.__call__, src/PGAS.py:176-228) as two decoupled pipelines.
//
// k_propagate  advances every particle through time steps [t0, t1) with its state in registers:
//              no dependence on the weights or ancestors (quirk Q1), hence no barrier, no LDS, no
//              inter-workgroup traffic.  Per particle-step it writes x_t (trace), la_t = log p(y_t|aux_t),
//              h_t = log N(ref_t; aux_t, S) and ln_t = log p(y_t | x_t): everything the weight
//              recursion needs later.
// k_resample   one launch per time step: systematic-resampling search of step t-1 (needs k_upper(t-1)),
//              logw_{t-1} = ln_{t-1} - la_{t-1}[a], then both softmax scans of step t.
// ------------------------------------------------------------------------------------------
template <int NX, int D, int JIN, int P, int W>
__global__ __launch_bounds__(PG_BLK, W) void k_propagate(DevModel md, TransParams tp, uint64_t seed, int t0, int t1,
                                                          double* __restrict__ x_trace, const double* __restrict__ ref,
                                                          double* __restrict__ la_buf, double* __restrict__ h_buf,
                                                          double* __restrict__ ln_buf) {
    const int seg = blockIdx.x, tid = threadIdx.x;
    const size_t row = (size_t)md.N * NX, np = (size_t)md.nseg * PGAS_SEG;
    double xv[PG_PPT][NX];
    load_particles<NX>(md, x_trace + (size_t)(t0 - 1) * row, seg, xv);
    for (int t = t0; t < t1; ++t) {
        double xn[PG_PPT][NX], la[PG_PPT], h[PG_PPT], aux[PG_PPT][NX];
        propagate_particles<NX, D, JIN, P>(md, tp, t, seed, ref + (size_t)t * NX, seg, xv, xn, la, h, aux);
        store_particles<NX>(md, x_trace + (size_t)t * row, seg, xn);
        const double* __restrict__ yt = md.y + (size_t)t * md.ny;
#pragma unroll
        for (int r = 0; r < PG_PPT; ++r) {
            const size_t pi = (size_t)seg * PGAS_SEG + r * PG_BLK + tid;  // buffers are padded to nseg*SEG
            la_buf[(size_t)t * np + pi] = la[r];
            h_buf[(size_t)t * np + pi] = h[r];
            ln_buf[(size_t)t * np + pi] = loglik<NX>(md, yt, xn[r]);
#pragma unroll
            for (int k = 0; k < NX; ++k) xv[r][k] = xn[r][k];
        }
    }
}

// mode bits of k_resample
#define PG_RS_SEARCH 1  // resample step t-1 (sb_prev valid) and form logw_{t-1}; otherwise logw_{t-1} = 0 (t = 1)
#define PG_RS_SCAN 2    // scan step t's weights into sb_next; otherwise only emit logw_{t-1} (after the last step)
__global__ __launch_bounds__(PG_BLK) void k_resample(DevModel md, int t, int mode, double u1_prev, const double* __restrict__ la_t,
                                                      const double* __restrict__ h_t, const double* __restrict__ ln_prev,
                                                      ScanBufs sb_prev, ScanBufs sb_next, Peers pr, int64_t laux_row_off,
                                                      int32_t* __restrict__ anc_out, double* __restrict__ logw_out) {
    __shared__ union {
        BackSmem b;
        ScanSmem s;
    } sm;
    const int seg = blockIdx.x, tid = threadIdx.x;
    double lwp[PG_PPT] = {0.0, 0.0, 0.0, 0.0};
    if (mode & PG_RS_SEARCH) {
        double lnv[PG_PPT];
#pragma unroll
        for (int r = 0; r < PG_PPT; ++r) lnv[r] = ln_prev[(size_t)seg * PGAS_SEG + r * PG_BLK + tid];
        int anc[PG_PPT];
        resample_slots(md, sm.b, u1_prev, sb_prev, pr, seg, anc_out, anc);
#pragma unroll
        for (int r = 0; r < PG_PPT; ++r) {
            const int64_t i = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
            if (i < md.N) {
                // log p(y_{t-1} | aux_{t-1}) of the ancestor: this device's row, or the owning peer's (src/PGAS.py:146)
                const double la_anc = pr.world > 1 ? pr.laux[anc[r] / pr.Nl][laux_row_off + anc[r] % pr.Nl] : sb_prev.laux[anc[r]];
                lwp[r] = lnv[r] - la_anc;
            }
        }
        if (logw_out != nullptr) {
#pragma unroll
            for (int r = 0; r < PG_PPT; ++r) {
                const int64_t i = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
                if (i < md.N) logw_out[i] = lwp[r];
            }
        }
        __syncthreads();  // BackSmem -> ScanSmem reuse
    }
    if (mode & PG_RS_SCAN) {
        double lw[2][PG_PPT];
#pragma unroll
        for (int r = 0; r < PG_PPT; ++r) {
            const size_t pi = (size_t)seg * PGAS_SEG + r * PG_BLK + tid;
            const bool valid = pi < (size_t)md.N;
            const double l1 = la_t[pi] + lwp[r];
            lw[0][r] = valid ? l1 : -__builtin_inf();
            lw[1][r] = valid ? l1 + h_t[pi] : -__builtin_inf();
        }
        segment_scan<2>(sm.s, lw, seg, sb_next.nsegp, sb_next.c1, sb_next.c2, sb_next.segm_w, sb_next.segs_w);
    }
}

// ------------------------------------------------------------------------------------------
// k_resample_fast: k_resample with the cross-segment scan of step t-1 folded in (no k_upper launch on
// the critical path).  Every workgroup recomputes the scan of the resampling CDF from the nseg <= 1024
// segment (max,total) pairs -- 16 KB from L2, three barriers -- keeps its running maximum in LDS, and goes
// straight on to the search.  The workgroup that owns the conditioned particle N-1 also scans the
// ancestor CDF and draws the reference particle's ancestor (src/PGAS.py:121-127).  Bit-identical to the
// k_upper + k_resample pair (same upper_core, same search).
// ------------------------------------------------------------------------------------------
#define PG_FAST_GPW 4
#define PG_FGROUPS 4  // at most this many staged windows per workgroup before falling back to per-slot bisection
#define PG_FSTAGE 2   // source segments staged at once by k_resample_fast (LDS budget: five workgroups per CU)
#define PG_FAST_NSEG (64 * PG_UPPER_WAVES * PG_FAST_GPW)
struct FastSmem {
    double cm[PG_FAST_NSEG];
    union {
        double num[PG_FSTAGE][PGAS_SEG];
        struct {
            double excl[PG_FAST_NSEG], scale[PG_FAST_NSEG];
        } tab;
        ScanSmem scan;
        int a[PGAS_SEG];  // ancestors, slot-major -> particle-major exchange (after the search is done with `num`)
    } u;
    // candidate source segments of this workgroup (index, excl, scale, carry): copied out of `tab` before staging reuses it
    int cand_b[PG_FGROUPS * PG_FSTAGE];
    double cand_ex[PG_FGROUPS * PG_FSTAGE], cand_sc[PG_FGROUPS * PG_FSTAGE], cand_cy[PG_FGROUPS * PG_FSTAGE];
    UpperSmemT<PG_UPPER_WAVES * PG_FAST_GPW> up;
};

// Grid = nseg + 1 workgroups.  Workgroup 0 only draws the reference particle's ancestor (scan of the ancestor CDF + one
// count, src/PGAS.py:121-127) and publishes it as one 8-byte {launch tag, index} word; workgroup b + 1 owns segment b.
// The workgroup that owns the conditioned particle reads that word late (after its own search).  Workgroup 0 never
// waits for anyone, so the hand-off cannot deadlock whatever the dispatch order; if the word has not arrived within the
// spin budget the owner computes the ancestor itself (same code, same result).
__global__ __launch_bounds__(PG_BLK, 5) void k_resample_fast(DevModel md, int t, int mode, unsigned tag, double u1_prev, double u2_prev, AncInputs anc_in,
                                                           const double* __restrict__ la_t, const double* __restrict__ h_t,
                                                           const double* __restrict__ ln_prev, ScanBufs sb_prev, ScanBufs sb_next,
                                                           int32_t* __restrict__ anc_out, double* __restrict__ logw_out) {
    __shared__ FastSmem sm;
    constexpr int GPW = PG_FAST_GPW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nseg = md.nseg, N = md.N;
    if (blockIdx.x == 0) {  // ---- ancestor workgroup
        if (!(mode & PG_RS_SEARCH)) return;
        double ex2[1][GPW], sc2[1][GPW], cm2[1][GPW], S2[1];
        upper_core<GPW, 1>(sm.up, sb_prev.segm + sb_prev.nsegp, sb_prev.segs + sb_prev.nsegp, 0, nseg, ex2, sc2, cm2, S2);
        int r = N - 1;
        if ((S2[0] > 0.0) && (S2[0] < __builtin_inf()))
            r = cdf_count_block_recompute<GPW>(sm.up, ex2[0], sc2[0], cm2[0], nseg, N, u2_prev * S2[0], anc_in);
        if (tid == 0)
            __hip_atomic_store(&sb_prev.hdr->ref_granule, ((unsigned long long)tag << 32) | (unsigned)r, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int seg = blockIdx.x - 1;
    const int64_t base_i = (int64_t)seg * PGAS_SEG;
    PG_STAMP(0);
    // own-particle inputs first: their latency overlaps the scan below
    double lnv[PG_PPT];
#pragma unroll
    for (int r = 0; r < PG_PPT; ++r) lnv[r] = (mode & PG_RS_SEARCH) ? ln_prev[(size_t)base_i + r * PG_BLK + tid] : 0.0;
    double lwp[PG_PPT] = {0.0, 0.0, 0.0, 0.0};
    if (mode & PG_RS_SEARCH) {
        const int nsegp = sb_prev.nsegp;
        // ---- cross-segment scan of the resampling CDF of step t-1; the workgroup that owns the conditioned particle
        // scans the ancestor CDF in the same pass (shared barriers) and draws the reference particle's ancestor
        // (src/PGAS.py:121-127)
        const bool last_wg = base_i + PGAS_SEG >= N;  // uniform
        double ex[GPW], sc[GPW], cmx[GPW], S;
        {
            double ex1[1][GPW], sc1[1][GPW], cm1[1][GPW], S1[1];
            upper_core<GPW, 1>(sm.up, sb_prev.segm, sb_prev.segs, 0, nseg, ex1, sc1, cm1, S1);
#pragma unroll
            for (int e = 0; e < GPW; ++e) {
                ex[e] = ex1[0][e];
                sc[e] = sc1[0][e];
                cmx[e] = cm1[0][e];
            }
            S = S1[0];
        }
        const bool valid = (S > 0.0) && (S < __builtin_inf());
        PG_STAMP(1);
#pragma unroll
        for (int e = 0; e < GPW; ++e) {
            const int b = ((wave + PG_UPPER_WAVES * e) << 6) + lane;
            sm.cm[b] = b < nseg ? cmx[e] : __builtin_inf();
            sm.u.tab.excl[b] = ex[e];
            sm.u.tab.scale[b] = sc[e];
        }
        __syncthreads();
        const bool pow2 = (N & (N - 1)) == 0;
        const double invN = 1.0 / (double)N;
        const int nslots = (N - base_i) < PGAS_SEG ? (int)(N - base_i) : PGAS_SEG;
        double tau[PG_PPT];
        int a[PG_PPT];
#pragma unroll
        for (int j = 0; j < PG_PPT; ++j) {
            const int64_t i = base_i + slot_of(tid, j);
            tau[j] = slot_tau(u1_prev, i, N, invN, pow2, S);
            a[j] = valid ? N - 1 : (int)(i < N ? i : N - 1);
        }
        int ns = 0;  // non-empty source segments in [b_lo, b_hi], counted up to PG_FGROUPS * PG_FSTAGE + 1
        int b_lo = 0, b_hi = 0;
        if (valid) {  // uniform
            // source-segment range of this workgroup: lower bounds over the running maxima (branch-free bisection, LDS)
            const double tau_first = slot_tau(u1_prev, base_i, N, invN, pow2, S);
            const double tau_last = slot_tau(u1_prev, base_i + nslots - 1, N, invN, pow2, S);
#pragma unroll
            for (int step = PG_FAST_NSEG / 2; step >= 1; step >>= 1) {
                if (sm.cm[b_lo + step - 1] < tau_first) b_lo += step;
                if (sm.cm[b_hi + step - 1] < tau_last) b_hi += step;
            }
            // (cm is padded with +inf, so both counts are <= nseg; a count of nseg means "past the last segment")
            if (b_hi > nseg - 1) b_hi = nseg - 1;
            // enumerate the non-empty source segments of [b_lo, b_hi] (a segment whose running max did not move owns no
            // slot) by bisection jumps: the next one after carry c is #{b : cm[b] <= c}.  Bounded work however long the
            // run of empty segments between two heavy particles is.
            int b = b_lo;
            while (ns <= PG_FGROUPS * PG_FSTAGE) {
                const double c0 = b ? sm.cm[b - 1] : 0.0;
                int nb = 0;
#pragma unroll
                for (int step = PG_FAST_NSEG / 2; step >= 1; step >>= 1)
                    if (sm.cm[nb + step - 1] <= c0) nb += step;
                if (nb + 1 <= PG_FAST_NSEG && sm.cm[nb] <= c0) ++nb;  // all PG_FAST_NSEG entries <= c0 cannot happen (cm pads with +inf or ends at S > c0)
                if (nb > b_hi) break;
                if (ns < PG_FGROUPS * PG_FSTAGE && tid == 0) {
                    sm.cand_b[ns] = nb;
                    sm.cand_ex[ns] = sm.u.tab.excl[nb];
                    sm.cand_sc[ns] = sm.u.tab.scale[nb];
                    sm.cand_cy[ns] = c0;
                }
                ++ns;
                b = nb + 1;
            }
        }
        PG_STAMP(2);
        if (ns > 0 && ns <= PG_FGROUPS * PG_FSTAGE) {
            // ---- common case: stage the numerators of PG_FSTAGE source segments at a time, back to back; they are
            // non-decreasing across the window (running-max carry), so one branch-free lower bound per slot settles
            // every slot that falls into the window
            __syncthreads();
            for (int k0w = 0; k0w < ns; k0w += PG_FSTAGE) {  // uniform
                const int ng = ns - k0w < PG_FSTAGE ? ns - k0w : PG_FSTAGE;
                int sb_idx[PG_FSTAGE] = {0, 0};
                double sex[PG_FSTAGE] = {0.0, 0.0}, ssc[PG_FSTAGE] = {0.0, 0.0}, scy[PG_FSTAGE] = {0.0, 0.0};
#pragma unroll
                for (int g = 0; g < PG_FSTAGE; ++g) {
                    if (g < ng) {
                        sb_idx[g] = sm.cand_b[k0w + g];
                        sex[g] = sm.cand_ex[k0w + g];
                        ssc[g] = sm.cand_sc[k0w + g];
                        scy[g] = sm.cand_cy[k0w + g];
                    }
                }
                const double g_carry = scy[0];
                // global loads first, then the barrier that frees the table / the previous window
                ulonglong2 c01[PG_FSTAGE], c23[PG_FSTAGE];
#pragma unroll
                for (int g = 0; g < PG_FSTAGE; ++g) {
                    if (g < ng) {
                        const ulonglong2* src = reinterpret_cast<const ulonglong2*>(sb_prev.c1 + (int64_t)sb_idx[g] * PGAS_SEG) + 2 * tid;
                        c01[g] = src[0];
                        c23[g] = src[1];
                    }
                }
                __syncthreads();
#pragma unroll
                for (int g = 0; g < PG_FSTAGE; ++g) {
                    double4 v = make_double4(__builtin_inf(), __builtin_inf(), __builtin_inf(), __builtin_inf());
                    if (g < ng) {
                        const int64_t base_k = (int64_t)sb_idx[g] * PGAS_SEG;
                        const int n = (N - base_k) < PGAS_SEG ? (int)(N - base_k) : PGAS_SEG;
                        const int k0 = PG_PPT * tid;
                        const double e0 = sex[g], s0 = ssc[g], cy = scy[g];
                        if (k0 + 0 < n) v.x = __builtin_fmax(e0 + s0 * (pgas_u64_to_double(c01[g].x) * PGAS_FIX_INV), cy);
                        if (k0 + 1 < n) v.y = __builtin_fmax(e0 + s0 * (pgas_u64_to_double(c01[g].y) * PGAS_FIX_INV), cy);
                        if (k0 + 2 < n) v.z = __builtin_fmax(e0 + s0 * (pgas_u64_to_double(c23[g].x) * PGAS_FIX_INV), cy);
                        if (k0 + 3 < n) v.w = __builtin_fmax(e0 + s0 * (pgas_u64_to_double(c23[g].y) * PGAS_FIX_INV), cy);
                    }
                    reinterpret_cast<double4*>(sm.u.num[g])[tid] = v;
                }
                __syncthreads();
                PG_STAMP(3);
                const double* __restrict__ num = &sm.u.num[0][0];
                int pos[PG_PPT] = {0, 0, 0, 0};
#pragma unroll
                for (int step = PG_FSTAGE * PGAS_SEG / 2; step >= 1; step >>= 1) {
#pragma unroll
                    for (int j = 0; j < PG_PPT; ++j) {  // loads are unconditional so the four chains advance in lock step
                        const int q = pos[j] + step;
                        const int qc = q <= PG_FSTAGE * PGAS_SEG ? q : PG_FSTAGE * PGAS_SEG;
                        const double v = num[qc - 1];
                        pos[j] = (q <= PG_FSTAGE * PGAS_SEG && v < tau[j]) ? q : pos[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < PG_PPT; ++j) {
                    if (g_carry < tau[j] && pos[j] < ng * PGAS_SEG) {
                        const int g = pos[j] >> 10, off = pos[j] & (PGAS_SEG - 1);
                        const int64_t ai = (int64_t)(g == 0 ? sb_idx[0] : sb_idx[1]) * PGAS_SEG + off;
                        a[j] = ai > N - 1 ? N - 1 : (int)ai;
                    }
                }
            }
        } else if (ns > PG_FGROUPS * PG_FSTAGE) {
            // degenerate weights: per-slot bisection, segment level in LDS, particle level in global memory
#pragma unroll
            for (int j = 0; j < PG_PPT; ++j) {
                int bs = 0;
#pragma unroll
                for (int step = PG_FAST_NSEG / 2; step >= 1; step >>= 1)
                    if (sm.cm[bs + step - 1] < tau[j]) bs += step;
                if (bs < nseg) {
                    const double e0 = sm.u.tab.excl[bs], s0 = sm.u.tab.scale[bs], cy = bs ? sm.cm[bs - 1] : 0.0;
                    const int64_t base_k = (int64_t)bs * PGAS_SEG;
                    const int n = (N - base_k) < PGAS_SEG ? (int)(N - base_k) : PGAS_SEG;
                    const uint64_t* __restrict__ c = sb_prev.c1 + base_k;
                    int lo = 0, hi = n;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        const double v = __builtin_fmax(e0 + s0 * (pgas_u64_to_double(c[mid]) * PGAS_FIX_INV), cy);
                        if (v < tau[j]) lo = mid + 1; else hi = mid;
                    }
                    const int64_t ai = base_k + lo;
                    a[j] = ai > N - 1 ? N - 1 : (int)ai;
                }
            }
        }
        if (last_wg) {
            // ancestor of the conditioned particle, published by workgroup 0 (src/PGAS.py:127)
            __syncthreads();
            if (tid == 0) {
                unsigned long long g = 0;
                int spins = 0;
                for (; spins < (1 << 16); ++spins) {
                    g = __hip_atomic_load(&sb_prev.hdr->ref_granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(g >> 32) == tag) break;
                    __builtin_amdgcn_s_sleep(8);
                }
                sm.up.cnt[0] = ((unsigned)(g >> 32) == tag) ? (int)(unsigned)g : -1;
            }
            __syncthreads();
            int ref_idx = sm.up.cnt[0];
            if (ref_idx < 0) {  // uniform: the word never arrived -- draw the ancestor here
                __syncthreads();
                double ex2[1][GPW], sc2[1][GPW], cm2[1][GPW], S2[1];
                upper_core<GPW, 1>(sm.up, sb_prev.segm + nsegp, sb_prev.segs + nsegp, 0, nseg, ex2, sc2, cm2, S2);
                ref_idx = N - 1;
                if ((S2[0] > 0.0) && (S2[0] < __builtin_inf()))
                    ref_idx = cdf_count_block_recompute<GPW>(sm.up, ex2[0], sc2[0], cm2[0], nseg, N, u2_prev * S2[0], anc_in);
            }
#pragma unroll
            for (int j = 0; j < PG_PPT; ++j)
                if (base_i + slot_of(tid, j) == N - 1) a[j] = ref_idx;
        }
        PG_STAMP(4);
        // ---- slot-major -> particle-major through LDS, ancestor trace, weight update
        __syncthreads();
    #pragma unroll
        for (int j = 0; j < PG_PPT; ++j) {
            sm.u.a[slot_of(tid, j)] = a[j];
            if (base_i + slot_of(tid, j) < N) anc_out[base_i + slot_of(tid, j)] = a[j];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PG_PPT; ++r) {
            const int64_t i = base_i + r * PG_BLK + tid;
            if (i < N) lwp[r] = lnv[r] - sb_prev.laux[sm.u.a[r * PG_BLK + tid]];
        }
        PG_STAMP(5);
        if (logw_out != nullptr) {
#pragma unroll
            for (int r = 0; r < PG_PPT; ++r) {
                const int64_t i = base_i + r * PG_BLK + tid;
                if (i < N) logw_out[i] = lwp[r];
            }
        }
        __syncthreads();  // staging area -> ScanSmem reuse
    }
    if (mode & PG_RS_SCAN) {
        double lw[2][PG_PPT];
#pragma unroll
        for (int r = 0; r < PG_PPT; ++r) {
            const size_t pi = (size_t)base_i + r * PG_BLK + tid;
            const bool valid_p = pi < (size_t)N;
            const double l1 = la_t[pi] + lwp[r];
            lw[0][r] = valid_p ? l1 : -__builtin_inf();
            lw[1][r] = valid_p ? l1 + h_t[pi] : -__builtin_inf();
        }
        PG_STAMP(6);
        segment_scan<2, false>(sm.u.scan, lw, seg, sb_next.nsegp, sb_next.c1, sb_next.c2, sb_next.segm_w, sb_next.segs_w);
    }
    PG_STAMP(7);
}

// ------------------------------------------------------------------------------------------
// k_segscan: softmax scan of a plain weight vector (final index draw, src/PGAS.py:224)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PG_BLK) void k_segscan(int N, const double* __restrict__ logw, ScanBufs sb) {
    __shared__ ScanSmem sm;
    const int seg = blockIdx.x, tid = threadIdx.x;
    double lw[1][PG_PPT];
#pragma unroll
    for (int r = 0; r < PG_PPT; ++r) {
        const int64_t i = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
        lw[0][r] = i < N ? logw[i] : -__builtin_inf();
    }
    segment_scan<1>(sm, lw, seg, sb.nsegp, sb.c1, nullptr, sb.segm_w, sb.segs_w);
}

// ------------------------------------------------------------------------------------------
// k_backtrace: reconstruct_trajectory (src/Filtering.py:40-55), one lane chases the ancestors
// ------------------------------------------------------------------------------------------
__global__ void k_backtrace(int Nl, int T, int nx, const double* __restrict__ x_trace, const int32_t* __restrict__ anc_trace,
                            Peers pr, const UpperHdr* __restrict__ hdr, double* __restrict__ traj) {
    // b is a GLOBAL particle index; its row lives on rank b / Nl (single device: rank 0, Nl = N)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int64_t b = hdr->final_idx;
    for (int i = T - 1; i >= 0; --i) {
        const int r = pr.world > 1 ? (int)(b / Nl) : 0;
        const int64_t bl = b - (int64_t)r * Nl;
        const double* __restrict__ xr = pr.world > 1 ? pr.x[r] : x_trace;
        for (int k = 0; k < nx; ++k) traj[(size_t)i * nx + k] = xr[((size_t)i * Nl + bl) * nx + k];
        if (i > 0) {
            const int32_t* __restrict__ ar = pr.world > 1 ? pr.anc[r] : anc_trace;
            // ancestor of particle b of time i, stored in row i-1 ... but row i-1 is indexed by the CHILD (time i) particle
            b = ar[(size_t)(i - 1) * Nl + bl];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Standalone Filtering entry points (reference src/Filtering.py): systematic_SISR on a weight vector whose softmax scan
// (k_segscan + k_upper) is already in sb, and reconstruct_trajectory from caller-owned traces.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PG_BLK) void k_systematic(DevModel md, double u, ScanBufs sb, int32_t* __restrict__ idx_out) {
    __shared__ BackSmem sm;
    int anc[PG_PPT];
    Peers none;
    none.world = 1;
    resample_slots(md, sm, u, sb, none, blockIdx.x, idx_out, anc, false);
}

__global__ void k_backtrace_idx(int N, int T, int nx, const double* __restrict__ x_trace, const int32_t* __restrict__ anc_trace,
                                int64_t idx, double* __restrict__ traj) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int64_t b = idx;
    for (int i = T - 1; i >= 0; --i) {
        for (int k = 0; k < nx; ++k) traj[(size_t)i * nx + k] = x_trace[((size_t)i * N + b) * nx + k];
        if (i > 0) b = anc_trace[(size_t)(i - 1) * N + b];
    }
}

// ------------------------------------------------------------------------------------------
// k_basis_eval (test hook): phi (np,M) in reference order; k_aux (test hook): aux (N,nx)
// ------------------------------------------------------------------------------------------
template <int NX>
__global__ __launch_bounds__(PG_BLK) void k_basis_eval(DevModel md, const int32_t* __restrict__ idx, const double* __restrict__ x,
                                                        int64_t np, int t, double* __restrict__ phi) {
    const int64_t p = (int64_t)blockIdx.x * PG_BLK + threadIdx.x;
    if (p >= np) return;
    const double* __restrict__ ut = md.u + (size_t)t * md.nu;
    double xv[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) xv[k] = x[p * NX + k];
    double sv[PGAS_MAX_D][PGAS_MAX_J];
    for (int d = 0; d < md.D; ++d) dim_sines_point<NX>(md, d, xv, ut, sv[d]);
    for (int m = 0; m < md.M; ++m) {
        double f = md.nrm;
        for (int d = 0; d < md.D; ++d) f = f * sv[d][(idx[m * md.D + d] - md.j0[d]) / md.jstep[d]];
        phi[p * md.M + m] = f;
    }
}

template <int NX, int D, int JIN, int P>
__global__ __launch_bounds__(PG_BLK) void k_aux(DevModel md, TransParams tp, int t, const double* __restrict__ x, double* __restrict__ aux_out) {
    const int seg = blockIdx.x, tid = threadIdx.x;
    double xv[PG_PPT][NX];
    load_particles<NX>(md, x, seg, xv);
    const double* __restrict__ ut = md.u + (size_t)t * md.nu;
#pragma unroll
    for (int r0 = 0; r0 < PG_PPT; r0 += P) {
        double xin[P][NX], aux[P][NX];
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int k = 0; k < NX; ++k) xin[p][k] = xv[r0 + p][k];
        eval_mean<NX, D, JIN, P>(md, tp.G, ut, xin, aux);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int64_t pi = (int64_t)seg * PGAS_SEG + (r0 + p) * PG_BLK + tid;
            if (pi < md.N)
#pragma unroll
                for (int k = 0; k < NX; ++k) aux_out[pi * NX + k] = aux[p][k];
        }
    }
}
