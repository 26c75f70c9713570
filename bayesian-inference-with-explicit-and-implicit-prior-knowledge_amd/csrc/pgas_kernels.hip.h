// pgas_kernels.hip.h -- gfx950 device code of the conditional-SMC engine.
//
// Kernel inventory (DESIGN.md section 5 has the byte accounting and the roofline of each):
//   k_pack        A (nx,M) -> coefficient tensor on the dense frequency grid
//   k_init        x_0 ~ N(m0,P0)                                   src/PGAS.py:155-174,194
//   k_front       per particle: basis, A phi, log-weights, propagate, per-segment softmax scans
//                                                                   src/PGAS.py:45-77,90-118,130-134
//   k_propagate   every particle through a range of time steps (one by default), no synchronisation
//   (pgas_resample.hip.h: k_groups, k_step, k_count, k_back, k_systematic -- the hierarchical CDF, the resampling search,
//    the weight update and the ancestor of the conditioned particle, src/Filtering.py:28-35, src/PGAS.py:101-127,137-147)
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
// k_step, read back with pgas_debug_stamps (tools/stamps_probe.py).  No stamp executes in the product build.
#ifdef PG_STAMPS
__device__ unsigned long long g_stamps[2048 * 16];
#define PG_STAMP(id)                                                                                \
    do {                                                                                            \
        if (threadIdx.x == 0) g_stamps[blockIdx.x * 16 + (id)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define PG_STAMP(id) do { } while (0)
#endif


// Streaming hints: data written once and read much later (x_t, the la / h / ln hand-off rows, the ancestor trace) is stored
// non-temporally so that it does not occupy this XCD's L2: 84.9 -> 81.2 ms per SMO sweep.  Measured and NOT adopted: the same for the
// fixed-point CDF (82.2 ms; the next launch's neighbours re-read it) and non-temporal LOADS of the hand-off rows / x_{t-1} (86 ms).
#ifndef PG_NO_STREAM_STORES
#define PG_NT_X
#define PG_NT_H
#define PG_NT_STORES
#endif
typedef double pg_nt_d2 __attribute__((ext_vector_type(2)));
typedef unsigned long long pg_nt_u2 __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ void st_stream(T* p, T v) {
#ifdef PG_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
template <typename T>
__device__ __forceinline__ T ld_stream(const T* p) {
#ifdef PG_NT_LOADS
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// load through the constant address space: for memory nothing writes while the kernel runs (device-resident parameters)
template <typename T>
__device__ __forceinline__ T ld_const(const T* p) {
    return *(const __attribute__((address_space(4))) T*)p;
}

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
    const uint64_t* qdesc;  // device (J[0]) or NULL, 3-D bases: per outermost frequency a, the count of leading innermost frequencies
                            // that carry a basis function in row (a, b), 5 bits per b (J[1] <= 12) -- the selected index sets are
                            // balls, not boxes (src/BasisFunctions.py:33-57), so about half of the dense grid is zeros the contraction
                            // can skip (exact: adding 0 * s changes nothing)
};

struct TransParams {   // transition parameters; ONE copy lives in device memory (pgas_ctx::d_tp) and every kernel reads it from there, so
                       // that a captured sweep (HIP graph) picks up new parameters on replay and pgas_set_params_dev can produce
                       // them without a host round trip
    double LS[4], LSinv[4], cS;
    const double* G;   // packed coefficient tensor
};

#define PGAS_LOG_2PI 0x1.d67f1c864beb4p+0   /* log(2 pi) rounded to nearest */

struct SweepParams {   // device-resident per-sweep scalars, written by k_sweep_begin before the time loop: the kernels of a sweep take
                       // nothing seed-dependent by value, so the whole sweep can be captured once in a HIP graph and replayed
    uint64_t seed;
    uint32_t epoch;    // sweeps run on this context so far: makes the ancestor hand-off tag of launch t unique across replays
    uint32_t pad;
    double u_final;    // uniform of the final index (src/PGAS.py:225)
};

struct UpperHdr {      // small per-scan-buffer results
    int32_t ref_idx;     // ancestor of the conditioned particle (k_count, step API)
    int32_t final_idx;   // index of the final draw (k_count)
    unsigned long long ref_granule;  // k_step: {launch tag : 32, ancestor of the conditioned particle : 32}, one 8-byte store
};

#define PG_MAX_RANKS 8
struct Peers {         // device pointers of every rank's scan buffers (xGMI peer mappings); world == 1: this device only
    const uint64_t* c1[PG_MAX_RANKS];   // the c1 buffer being READ this launch
    const uint64_t* c2[PG_MAX_RANKS];   // step API only (pgas_step keeps the ancestor cumsum)
    int32_t world, nseg_l, Nl;          // ranks, segments per rank, particles per rank
};

// The traces (x, la, h, ln, ancestors) are stored as ROW BLOCKS: consecutive time rows in allocations of at most 1 GiB, so that every
// block can be exported to the other ranks' processes with hipIpcGetMemHandle (the HIP runtime bundled with PyTorch never returns
// from hipIpcOpenMemHandle for an allocation of 2 GiB or more).  Kernels that touch a handful of rows get the row pointers from
// the host, per launch (StepArgs / AncIn, k_propagate's arguments); only the ancestor chase walks every row of every rank and
// reads the block table itself.
#define PG_RB_X 0
#define PG_RB_LA 1
#define PG_RB_H 2
#define PG_RB_LN 3
#define PG_RB_ANC 4
#define PG_RB_NKIND 5
struct BtTab {         // k_backtrace: blk[(rank * 2 + (0: x, 1: anc)) * nblk_max + b] = base of block b (rows b << shift ...) on that rank
    const void* const* blk;
    int32_t nblk_max, shift_x, shift_anc, entries;
};

struct ScanBufs {      // per-step scan scratch (device)
    double* laux;      // (nseg*SEG) step API: log p(y_t | aux_t) of k_front
    uint64_t* c1;      // (nseg*SEG) quantised inclusive cumsum of the resampling weights, per segment
    uint64_t* c2;      // (nseg*SEG) same for the ancestor weights (step API only)
    double* segk;      // segment references kref as READ by the group scans: (ranks, 2, nsegp) after the all-gather
    uint64_t* segs;    // segment totals, same layout
    double* segk_w;    // (2, nsegp) where this device's segment scans WRITE (== segk on a single device)
    uint64_t* segs_w;
    int32_t nseg_l;    // segments per rank in segk/segs (single device: >= nseg, so every segment maps to "rank" 0)
    int32_t rank_stride;  // words between two ranks' blocks in segk/segs (2 * nsegp)
    int32_t nsegp;     // padded LOCAL nseg (multiple of 64): stride between the two CDFs in segk/segs
    int32_t nsegp_g;   // padded GLOBAL nseg: stride between the two CDFs in tab_*
    double* tab_e;     // (2, nsegp_g) per-segment records written by k_groups: exclusive prefix inside the group,
    double* tab_sc;    //              scale 2^(kref - KG),
    double* tab_m;     //              running maximum of the segment-end values inside the group
    double* grp_K;     // (2, PG_MAX_GRP) group references KG
    double* grp_T;     // (2, PG_MAX_GRP) group totals TG
    unsigned* grp_cnt; // (PG_MAX_GRP) arrival counters of k_step's in-launch group scans (zero between uses)
    UpperHdr* hdr;
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

// FAST instantiations (template parameter J0T > 0 of eval_mean / k_propagate): the model's shape is known at compile time --
// basis input d is component d of concat(state, input) (sel[d] == d), every dimension's frequencies are j0, 2 j0, 3 j0, ...
// (jstep == j0: one sincospi per dimension, Chebyshev recurrence starting from 0), the outermost dimension has exactly J0T
// frequencies.  Same arithmetic, fewer branches, moves and scalar registers.  Every configuration of the reference has this shape.
template <int NX, int DSEL>
__device__ __forceinline__ double pick_input_fixed(const double (&x)[NX], const double* __restrict__ ut) {
    if constexpr (DSEL < NX) return x[DSEL];
    else return ut[DSEL - NX];
}

template <int NX>
__device__ __forceinline__ double pick_input(const DevModel& md, int d, const double (&x)[NX], const double* __restrict__ ut);
// r_d = v[sel[d]] alpha_d + beta_d, the argument of dimension d's sines in units of pi
template <int NX, int DSEL, bool FAST>
__device__ __forceinline__ double basis_arg(const DevModel& md, const double (&x)[NX], const double* __restrict__ ut) {
    double v;
    if constexpr (FAST) v = pick_input_fixed<NX, DSEL>(x, ut);
    else v = pick_input<NX>(md, DSEL, x, ut);
    return PGAS_FMA(v, md.alpha[DSEL], md.beta[DSEL]);
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
template <int P, bool FAST = false>
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
    if (!FAST && md.jstep[d] != md.j0[d]) {  // uniform
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
#ifndef PG_QCHUNK
#define PG_QCHUNK 4   // innermost frequencies per uniform branch of the 3-D contraction
#endif
struct Cheb {
    double cur, prev, tw;
};
template <bool FAST = false>
__device__ __forceinline__ Cheb cheb_init(const DevModel& md, int d, const DimStart& ds) {
    Cheb c;
    c.cur = ds.s0;
    c.prev = (FAST || md.jstep[d] == md.j0[d]) ? 0.0 : PGAS_FMA(ds.s0, ds.cd, -(ds.c0 * ds.sd));
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
// WIDE (k_sweep_pipe: one wave per SIMD, the coefficients in LDS, registers to spare): the rows of the 2-D FAST contraction fully
// unrolled, so that the seven row sums are independent instruction streams instead of one rolled, latency-bound loop.  Every
// accumulation keeps its order (q ascending inside a row, a ascending across rows): the same bits.
template <int NX, int D, int JIN, int P, int J0T = 0, bool WIDE = false>
__device__ __forceinline__ void eval_mean(const DevModel& md, const double* __restrict__ G, const double* __restrict__ ut,
                                          const double (&x)[P][NX], double (&aux)[P][NX]) {
    constexpr bool FAST = J0T > 0;
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int k = 0; k < NX; ++k) aux[p][k] = 0.0;
    if constexpr (D == 1) {
        double r[P];
        DimStart ds[P];
#pragma unroll
        for (int p = 0; p < P; ++p) r[p] = basis_arg<NX, 0, FAST>(md, x[p], ut);
        dim_start_n<P, FAST>(md, 0, r, ds);
        const int J0 = FAST ? J0T : md.J[0];
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
                rin[p] = basis_arg<NX, DI, FAST>(md, x[p], ut);
                rout[p] = basis_arg<NX, 0, FAST>(md, x[p], ut);
            }
            dim_start_n<P, FAST>(md, DI, rin, din);
            dim_start_n<P, FAST>(md, 0, rout, dout);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                Cheb ci = cheb_init<FAST>(md, DI, din[p]);
#pragma unroll
                for (int q = 0; q < JIN; ++q) {
                    tab[p][q] = ci.cur;
                    cheb_next(ci);
                }
                c0[p] = cheb_init<FAST>(md, 0, dout[p]);
            }
        }
        const int J0 = FAST ? J0T : md.J[0];
        if constexpr (D == 2) {
            auto body = [&](int a) {
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
            };
            if constexpr (FAST && WIDE) {
                double in[J0T][P][NX];
#pragma unroll
                for (int a = 0; a < J0T; ++a) {
                    const double* __restrict__ Ga = G + (size_t)a * JIN * NX;
#pragma unroll
                    for (int p = 0; p < P; ++p)
#pragma unroll
                        for (int k = 0; k < NX; ++k) in[a][p][k] = 0.0;
#pragma unroll
                    for (int q = 0; q < JIN; ++q)
#pragma unroll
                        for (int k = 0; k < NX; ++k)
#pragma unroll
                            for (int p = 0; p < P; ++p) in[a][p][k] = PGAS_FMA(Ga[q * NX + k], tab[p][q], in[a][p][k]);
                }
#pragma unroll
                for (int a = 0; a < J0T; ++a) {
#pragma unroll
                    for (int p = 0; p < P; ++p) {
#pragma unroll
                        for (int k = 0; k < NX; ++k) aux[p][k] = PGAS_FMA(c0[p].cur, in[a][p][k], aux[p][k]);
                        cheb_next(c0[p]);
                    }
                }
            } else if constexpr (FAST) {
                // Kept rolled (unrolled, the compiler hoists all J0T x JIN x NX coefficient loads to the top and spills ~290 SGPRs)
                // and software-pipelined by hand: the scalar loads of row a + 1 are issued before the FMAs of row a, so their
                // latency hides under 38 vector instructions instead of stalling every iteration.
                double g[JIN * NX];
#pragma unroll
                for (int i = 0; i < JIN * NX; ++i) g[i] = G[i];
#pragma unroll 1
                for (int a = 0; a < J0T; ++a) {
                    const double* __restrict__ Gn = G + (size_t)(a + 1 < J0T ? a + 1 : a) * JIN * NX;
                    double gn[JIN * NX];
#pragma unroll
                    for (int i = 0; i < JIN * NX; ++i) gn[i] = Gn[i];
                    double in[P][NX];
#pragma unroll
                    for (int p = 0; p < P; ++p)
#pragma unroll
                        for (int k = 0; k < NX; ++k) in[p][k] = 0.0;
#pragma unroll
                    for (int q = 0; q < JIN; ++q)
#pragma unroll
                        for (int k = 0; k < NX; ++k)
#pragma unroll
                            for (int p = 0; p < P; ++p) in[p][k] = PGAS_FMA(g[q * NX + k], tab[p][q], in[p][k]);
#pragma unroll
                    for (int p = 0; p < P; ++p) {
#pragma unroll
                        for (int k = 0; k < NX; ++k) aux[p][k] = PGAS_FMA(c0[p].cur, in[p][k], aux[p][k]);
                        cheb_next(c0[p]);
                    }
#pragma unroll
                    for (int i = 0; i < JIN * NX; ++i) g[i] = gn[i];
                }
            } else {
                for (int a = 0; a < J0; ++a) body(a);
            }
        } else {
            Cheb c1s[P];
            {
                double r1[P];
                DimStart d1[P];
#pragma unroll
                for (int p = 0; p < P; ++p) r1[p] = basis_arg<NX, 1, FAST>(md, x[p], ut);
                dim_start_n<P, FAST>(md, 1, r1, d1);
#pragma unroll
                for (int p = 0; p < P; ++p) c1s[p] = cheb_init<FAST>(md, 1, d1[p]);
            }
            const int J1 = md.J[1];
            // one row (a, b) of the grid: in = sum_q G[a][b][q] tab[q], mid += s1[b] in.  Rows without coefficients are skipped whole
            // (mid + s 0 = mid bit for bit: mid starts at +0 and +0 + (-0) = +0), the first chunk initialises `in` (no zeroing), and
            // the b loop is unrolled by two so that the recurrence s1[b+1] = 2c s1[b] - s1[b-1] alternates between two registers
            // instead of moving them (3 244 -> about 2 750 vector instructions per particle-step at M = 729).
            auto row = [&](int a, int b, const uint64_t desc, const double (&s1)[P], double (&mid)[P][NX]) {
                const int qn = md.qdesc ? (int)((desc >> (5 * b)) & 31u) : JIN;   // uniform: the row's coefficients beyond qn are zeros
                if (qn == 0) return;
                const double* __restrict__ Gab = G + ((size_t)a * J1 + b) * JIN * NX;
                double in[P][NX];
#pragma unroll
                for (int q0 = 0; q0 < JIN; q0 += PG_QCHUNK) {
                    if (q0 < qn) {   // uniform branch per chunk of PG_QCHUNK frequencies
#pragma unroll
                        for (int q = q0; q < (q0 + PG_QCHUNK < JIN ? q0 + PG_QCHUNK : JIN); ++q) {
#pragma unroll
                            for (int k = 0; k < NX; ++k) {
                                const double g = Gab[q * NX + k];
#pragma unroll
                                for (int p = 0; p < P; ++p) in[p][k] = q == 0 ? g * tab[p][q] : PGAS_FMA(g, tab[p][q], in[p][k]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int k = 0; k < NX; ++k) mid[p][k] = PGAS_FMA(s1[p], in[p][k], mid[p][k]);
            };
            for (int a = 0; a < J0; ++a) {
                double mid[P][NX];
                double sa[P], sb[P];   // s1[b-1] / s1[b], alternating roles
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    sa[p] = c1s[p].prev;
                    sb[p] = c1s[p].cur;
#pragma unroll
                    for (int k = 0; k < NX; ++k) mid[p][k] = 0.0;
                }
                const uint64_t desc = md.qdesc ? md.qdesc[a] : 0ull;   // one scalar load per outermost frequency
                int b = 0;
                for (; b + 1 < J1; b += 2) {
                    row(a, b, desc, sb, mid);
#pragma unroll
                    for (int p = 0; p < P; ++p) sa[p] = PGAS_FMA(c1s[p].tw, sb[p], -sa[p]);
                    row(a, b + 1, desc, sa, mid);
#pragma unroll
                    for (int p = 0; p < P; ++p) sb[p] = PGAS_FMA(c1s[p].tw, sa[p], -sb[p]);
                }
                if (b < J1) row(a, b, desc, sb, mid);
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

// ------------------------------------------------------------------------------------------
// The 3-D contraction on the f64 matrix cores (k_propagate<..., MX>; nx = 2, innermost extent 12, J[0], J[1] <= 12: the
// 729-function bases of EMPS / Vehicle).  v_mfma_f64_16x16x4_f64 accumulates its four products as the ascending chain
// acc = fma(a_k, b_k, acc), k = 0..3 (tools/probes/mfma_f64_probe.hip: 200 of 200 random trials bit-identical on gfx950), which IS
// the canonical order of the innermost sum in = sum_q G[a][b][q][k] tab[q] (q ascending) -- up to the sign of an
// exact zero (the vector form starts with a product, the matrix form from +0), which cannot reach `mid` (mid starts at +0 and
// +0 + (-0) = +0).  So stage 1 runs as MFMA tiles and stages 2 / 3 (the sums over b and a, with their sine recurrences) stay
// scalar chains in the canonical order: the same bits as eval_mean.
//
// Geometry of one MFMA (A: lane l holds A[l % 16][l / 16], B: lane l holds B[l / 16][l % 16], D: lane l holds D[4 r + l / 16][l % 16]):
//   columns = 16 particles (sub-batch s of the wave's 64), K = 4 innermost frequencies (step ks of 3), rows = 16 (a, b, k) triples.
//   Tile (ci, tau) carries rows i = 4 (b - 4 tau) + g with lane group g = l / 16 of the RESULT owning (a = 2 ci + g / 2, k = g % 2):
//   after the tiles tau = 0, 1, 2 of pair ci, lane (g, n) holds in[a][b][k], b = 0..11, of particle 16 s + n in 12 accumulator
//   registers with constant indices -- exactly one b-chain.  The ball-shaped index set (MxInfo::desc: K steps per tile) lets most
//   tiles stop after 1 or 2 K steps and some vanish: 35 MFMAs per 16 particles at M = 729 instead of the dense 54.
// The operands: A from an image of the coefficient tensor in LDS (k_pack_mx writes it per parameter set, one 512-byte row per MFMA),
// B from a per-wave transposed table of the innermost sines in LDS; the recurrence starts of the two outer dimensions travel by
// ds_bpermute; stage 3 pairs lane groups g and g + 2 (a = 2 ci, 2 ci + 1: ascending a).
#define PG_MX_MAXSLOTS 54   // 6 pairs x 3 tiles x 3 K steps
#define PG_MX_TSTRIDE 13    // doubles per particle of the transposed sine table (12 + 1: conflict-free columns)
struct MxInfo {
    uint64_t desc;        // 2 bits per (pair ci, tile tau) at bit 6 ci + 2 tau: K steps of that tile (0..3)
    int32_t nslots, pad;  // MFMAs per sub-batch = rows of the image
    int32_t idx[PG_MX_MAXSLOTS * 64];   // element of the packed tensor G behind image entry (slot, lane); -1: zero
};
typedef double pg_d4 __attribute__((ext_vector_type(4)));

__global__ void k_pack_mx(const double* __restrict__ G, const MxInfo* __restrict__ mi, double* __restrict__ img) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < mi->nslots * 64) {
        const int j = mi->idx[i];
        img[i] = j >= 0 ? G[j] : 0.0;
    }
}

// K steps of tile i = 3 ci + tau under descriptor `desc`, and the image row of its first MFMA
__host__ __device__ constexpr int mx_nk(uint64_t desc, int i) { return (int)((desc >> (2 * i)) & 3u); }
__host__ __device__ constexpr int mx_slot(uint64_t desc, int i) {
    int n = 0;
    for (int j = 0; j < i; ++j) n += mx_nk(desc, j);
    return n;
}
// The tile descriptor is a COMPILE-TIME constant (like the FAST shapes above): with the K-step counts known, a sub-batch is straight-line
// code -- 35 MFMAs at M = 729 with the image rows at immediate offsets, the b-chains between them -- instead of 54 uniform branches
// (measured: the branchy form ran at 179 us per step against 114 us for the vector form).
#define PG_MX_DESC_BALL729 0x4a6efbefull   // the 729-function ball of the 11 x 11 x 11 grid (EMPS and Vehicle: src/EMPS.py:101-113)
template <int NX, int JIN, uint64_t DESC>
__device__ __forceinline__ void eval_mean_mx(const DevModel& md, const double* __restrict__ aimg, double* __restrict__ tabT,
                                             const double* __restrict__ ut, const double (&x)[NX], double (&aux)[NX]) {
    static_assert(NX == 2 && JIN == 12, "instantiated for the 729-function bases");
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    Cheb c0, c1;
    {
        double r2[1] = {basis_arg<NX, 2, false>(md, x, ut)}, r0[1] = {basis_arg<NX, 0, false>(md, x, ut)}, r1[1] = {basis_arg<NX, 1, false>(md, x, ut)};
        DimStart d2[1], d0[1], d1[1];
        dim_start_n<1>(md, 2, r2, d2);
        dim_start_n<1>(md, 0, r0, d0);
        dim_start_n<1>(md, 1, r1, d1);
        Cheb ci = cheb_init(md, 2, d2[0]);
#pragma unroll
        for (int q = 0; q < JIN; ++q) {
            tabT[lane * PG_MX_TSTRIDE + q] = ci.cur;
            cheb_next(ci);
        }
        c0 = cheb_init(md, 0, d0[0]);
        c1 = cheb_init(md, 1, d1[0]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the table is read by other lanes of this wave only: program order is enough
    __builtin_amdgcn_wave_barrier();
    const int J0 = md.J[0];
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
        const int src = 16 * s + n;   // the particle (lane of this wave) whose column this lane holds in sub-batch s
        double B[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) B[ks] = tabT[src * PG_MX_TSTRIDE + 4 * ks + g];
        Cheb e0, e1;
        e0.cur = __shfl(c0.cur, src); e0.prev = __shfl(c0.prev, src); e0.tw = __shfl(c0.tw, src);
        e1.cur = __shfl(c1.cur, src); e1.prev = __shfl(c1.prev, src); e1.tw = __shfl(c1.tw, src);
        // The MFMAs of a sub-batch are one stream over the image's rows (slot = 0, 1, ...): row slot + 1 is fetched while MFMA `slot`
        // runs; the tiles come in (ci, tau) order, each with its own K-step count, and the b-chain of tile i is issued AFTER the
        // MFMAs of tile i + 1 so that the matrix core works while the vector ALU adds.
        const double* __restrict__ ap = aimg + lane;
        pg_d4 acc[2];
        auto tile = [&](const int i, pg_d4& d) {   // i is a constant after unrolling: bit position 6 ci + 2 tau = 2 (3 ci + tau)
            const int nk = mx_nk(DESC, i), s0 = mx_slot(DESC, i);
            d = pg_d4{0.0, 0.0, 0.0, 0.0};
            if (nk > 0) d = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[(s0 + 0) * 64], B[0], d, 0, 0, 0);
            if (nk > 1) d = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[(s0 + 1) * 64], B[1], d, 0, 0, 0);
            if (nk > 2) d = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[(s0 + 2) * 64], B[2], d, 0, 0, 0);
        };
        double midv[6];
        double sa = 0.0, sb = 0.0, mid = 0.0;
        tile(0, acc[0]);
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int ci = i / 3, tau = i % 3;
            if (i + 1 < 18) tile(i + 1, acc[(i + 1) & 1]);
            // stage 2: mid = sum_b s1[b] in[b], b ascending, the recurrence of eval_mean (s1[b+1] = tw s1[b] - s1[b-1]); tiles behind
            // the last one with coefficients are skipped (their rows are zeros: mid + s 0 = mid)
            if (tau == 0) {
                sa = e1.prev; sb = e1.cur; mid = 0.0;
            }
            if ((((unsigned)(DESC >> (6 * ci)) & 63u) >> (2 * tau)) != 0u) {
                const pg_d4 d = acc[i & 1];
                mid = PGAS_FMA(sb, d[0], mid);
                sa = PGAS_FMA(e1.tw, sb, -sa);
                mid = PGAS_FMA(sa, d[1], mid);
                sb = PGAS_FMA(e1.tw, sa, -sb);
                mid = PGAS_FMA(sb, d[2], mid);
                sa = PGAS_FMA(e1.tw, sb, -sa);
                mid = PGAS_FMA(sa, d[3], mid);
                sb = PGAS_FMA(e1.tw, sa, -sb);
            }
            if (tau == 2) midv[ci] = mid;
        }
        // stage 3 on lane groups 0 / 1 (k = g): aux = sum_a s0[a] mid[a], a ascending = own pair member, then group g + 2's
        double al = 0.0;
#pragma unroll
        for (int ci = 0; ci < 6; ++ci) {
            const double other = __shfl_xor(midv[ci], 32);
            if (2 * ci < J0) {
                al = PGAS_FMA(e0.cur, midv[ci], al);
                cheb_next(e0);
            }
            if (2 * ci + 1 < J0) {
                al = PGAS_FMA(e0.cur, other, al);
                cheb_next(e0);
            }
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const double v = __shfl(al, k * 16 + n);
            if (g == s) aux[k] = v;
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
                       double* __restrict__ G, int64_t gtotal, TransParams tp_host, const double* __restrict__ S_dev, TransParams* __restrict__ tp_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // phase split by launch: caller memsets G first
    if (i < (int64_t)M * nx) {
        const int m = (int)(i / nx), k = (int)(i % nx);
        G[(int64_t)pos[m] * nx + k] = A[(int64_t)k * M + m] * nrm;
    }
    (void)gtotal;
    if (i == 0) {
        TransParams tp = tp_host;   // pgas_set_params: factor, inverse and constant computed by the caller
        if (S_dev != nullptr) {
            // pgas_set_params_dev: error_cov (nx, nx) is on the device; its Cholesky factor, the factor's inverse and the normalising
            // constant in IEEE operations only (sqrt, /, *, -, pgas_log), in this order -- oracle/canon.py chol_parts_dev mirrors it
            for (int q = 0; q < 4; ++q) tp.LS[q] = 0.0, tp.LSinv[q] = 0.0;
            if (nx == 1) {
                const double l = __builtin_sqrt(S_dev[0]);
                tp.LS[0] = l;
                tp.LSinv[0] = 1.0 / l;
                tp.cS = -0.5 * PGAS_LOG_2PI - pgas_log(l);
            } else {
                const double l00 = __builtin_sqrt(S_dev[0]);
                const double l10 = S_dev[2] / l00;
                const double l11 = __builtin_sqrt(S_dev[3] - l10 * l10);
                const double i00 = 1.0 / l00, i11 = 1.0 / l11;
                tp.LS[0] = l00; tp.LS[2] = l10; tp.LS[3] = l11;
                tp.LSinv[0] = i00; tp.LSinv[2] = -(l10 * i00) * i11; tp.LSinv[3] = i11;
                tp.cS = -PGAS_LOG_2PI - (pgas_log(l00) + pgas_log(l11));
            }
        }
        *tp_out = tp;
    }
}

// ------------------------------------------------------------------------------------------
// k_sweep_begin: the sweep's scalars and its T + 1 resampling / ancestor uniforms (src/Filtering.py:19, src/PGAS.py:123) into device memory
// ------------------------------------------------------------------------------------------
__global__ void k_sweep_begin(uint64_t seed, uint32_t epoch, int T, SweepParams* __restrict__ sp, double* __restrict__ u_res,
                              double* __restrict__ u_anc) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t <= T) {
        u_res[t] = pgas_rng_uniform(seed, PGAS_STREAM_RESAMPLE, (uint32_t)t);
        u_anc[t] = pgas_rng_uniform(seed, PGAS_STREAM_ANCESTOR, (uint32_t)t);
    }
    if (t == 0) {
        sp->seed = seed;
        sp->epoch = epoch;
        sp->pad = 0;
        sp->u_final = pgas_rng_uniform(seed, PGAS_STREAM_FINAL, 0u);
    }
}

// ------------------------------------------------------------------------------------------
// k_init: x_0 = m0 + L0 z, conditioned particle last
// ------------------------------------------------------------------------------------------
template <int NX>
__global__ __launch_bounds__(PG_BLK) void k_init(DevModel md, uint64_t seed_val, const SweepParams* __restrict__ sp /* non-NULL: seed from there */,
                                                  const double* __restrict__ m0L0 /* m0[nx], L0[nx*nx] */,
                                                  const double* __restrict__ ref0, double* __restrict__ x0) {
    const int64_t p = (int64_t)blockIdx.x * PG_BLK + threadIdx.x;
    if (p >= md.N) return;
    const uint64_t seed = sp ? sp->seed : seed_val;
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
// segment softmax scan shared by k_front / k_step / k_segscan.
// lw[r] is the log-weight of particle seg*SEG + r*BLK + tid (-inf when past N).  Writes the
// quantised inclusive cumsum (index order) and the segment (max, total).
// ------------------------------------------------------------------------------------------
// q = rint(exp(x) 2^51) as an integer, for x <= 0.25: the softmax numerators of DESIGN.md 4.3 (x = lw - kref ln 2 <= 0 up to rounding).
// Same values as pgas_double_to_u64(rint(pgas_exp(x) * 2^51)) with "0 unless exp(x) > 0" (include/pgas_detmath.h) on that domain, in
// fewer instructions: the argument is clamped at -708 instead of selected afterwards (exp(-708) 2^51 = 7e-293 rounds to 0, as the
// flushed 0 does; NaN and -inf take the same way), the factor 2^51 rides in the exponent that scales the polynomial, and the
// round-to-nearest-even integer is read from the mantissa of v + 2^52 (0 <= v < 2^52: exact, same rounding as rint).
template <int NB>
__device__ __forceinline__ void dev_exp_q51_n(const double (&x)[NB], uint64_t (&q)[NB]) {
    const double LOG2E = 0x1.71547652b82fep+0;
    const double LN2_HI = 0x1.62e42fefa39efp-1;
    const double LN2_LO = 0x1.abc9e3b39803fp-56;
    const double C[13] = {0x1.1eed8eff8d898p-29, 0x1.ae64567f544e4p-26, 0x1.27e4fb7789f5cp-22, 0x1.71de3a556c734p-19,
                          0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-13, 0x1.6c16c16c16c17p-10, 0x1.1111111111111p-7,
                          0x1.5555555555555p-5,  0x1.5555555555555p-3,  0.5, 1.0, 1.0};
    double k[NB], r[NB], p[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const double xc = __builtin_fmax(x[i], -708.0);   // NaN -> -708 as well (fmax ignores NaN): q = 0 like the reference expression
        k[i] = __builtin_rint(xc * LOG2E);
        r[i] = PGAS_FMA(-k[i], LN2_HI, xc);
        r[i] = PGAS_FMA(-k[i], LN2_LO, r[i]);
        p[i] = 0x1.6124613a86d09p-33; /* 1/13! */
    }
#pragma unroll
    for (int j = 0; j < 13; ++j)
#pragma unroll
        for (int i = 0; i < NB; ++i) p[i] = PGAS_FMA(p[i], r[i], C[j]);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        // p in (0.70, 1.42), k in [-1022, 0]: v = p 2^(k + 51) is a normal number below 2^52
        const int ki = (int)k[i] + PGAS_FIX_BITS;
        const double v = __hiloint2double(__double2hiint(p[i]) + (int)((unsigned)ki << 20), __double2loint(p[i]));
        const double y = v + 0x1p52;
        q[i] = ((uint64_t)(uint32_t)(__double2hiint(y) & 0x000fffff) << 32) | (uint32_t)__double2loint(y);
    }
}

// Transposition array of the segment scans: thread t WRITES the numerators of particles r * 256 + t (r = 0..3) and READS those of
// particles 4 t .. 4 t + 3 for the prefix sums.  Element i lives at plane (i & 3), slot (i >> 2), planes PG_QPLANE = 256 + 8 words
// apart: the reads (lane stride 8 bytes inside a plane) and the writes (planes 16 banks apart, 8 consecutive slots per plane in each
// half-wave) are both free of LDS bank conflicts.  Laid out linearly, the four 8-byte reads of a lane sat 32 bytes from its
// neighbour's: four lanes per bank (most of the 38 % conflict cycles round 2 measured in k_step, profiles/r02_pmc_sq_lds.txt).
#define PG_QPLANE (PG_BLK + 8)
__device__ __forceinline__ int q_slot(int i) { return (i & 3) * PG_QPLANE + (i >> 2); }
struct ScanSmem {
    uint64_t q[2][4 * PG_QPLANE];
    double red[2][PG_BLK / 64];
    uint64_t wtot[2][PG_BLK / 64];
};

// HANDOFF: the segment partials are stored write-through (agent-scope 8-byte stores) because another workgroup of the SAME launch
// reads them (the last arriver of the group, k_step); a plain store otherwise (the next launch reads them).
template <int NW, bool STORE_B = true, bool HANDOFF = false>  // NW: weight vectors scanned together (1 or 2); STORE_B = false: the second one only leaves its
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
#pragma unroll
        for (int w = 0; w < NW; ++w) {   // one batch of four numerators per weight vector: half the live registers of one batch of eight
            double m = sm.red[w][0];
#pragma unroll
            for (int v = 1; v < PG_BLK / 64; ++v) m = __builtin_fmax(m, sm.red[w][v]);
            const double kref = pgas_seg_ref(m);  // power-of-two reference of the segment (include/pgas_canon.h)
            mx[w] = kref;
            double arg[PG_PPT];
            uint64_t qv[PG_PPT];
#pragma unroll
            for (int r = 0; r < PG_PPT; ++r) arg[r] = pgas_seg_arg(lw[w][r], kref);
            dev_exp_q51_n<PG_PPT>(arg, qv);
#pragma unroll
            for (int r = 0; r < PG_PPT; ++r) {
                if (w == 0 || STORE_B) sm.q[w][q_slot(r * PG_BLK + tid)] = qv[r];  // strided -> contiguous particle order for the prefix
                else qsum_b += qv[r];                                      // only the segment total is wanted: any order will do
            }
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
                run += sm.q[w][j * PG_QPLANE + tid];   // = q_slot(4 tid + j)
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
#ifdef PG_NT_C1
            pg_nt_u2* dn = reinterpret_cast<pg_nt_u2*>(dst);
            __builtin_nontemporal_store(pg_nt_u2{base + loc[w][0], base + loc[w][1]}, dn);
            __builtin_nontemporal_store(pg_nt_u2{base + loc[w][2], base + loc[w][3]}, dn + 1);
#else
            dst[0] = make_ulonglong2(base + loc[w][0], base + loc[w][1]);
            dst[1] = make_ulonglong2(base + loc[w][2], base + loc[w][3]);
#endif
        }
        if (tid == 0) {
            if constexpr (HANDOFF) {
                __hip_atomic_store(&segm[(size_t)w * nsegp + seg], mx[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&segs[(size_t)w * nsegp + seg], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                segm[(size_t)w * nsegp + seg] = mx[w];
                segs[(size_t)w * nsegp + seg] = tot;
            }
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
template <int NX, int D, int JIN, int P, int J0T = 0>
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
        eval_mean<NX, D, JIN, P, J0T>(md, tp.G, ut, xin, ax);
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
__global__ __launch_bounds__(PG_BLK) void k_front(DevModel md, const TransParams* __restrict__ tpp, int t, uint64_t seed,
                                                   const double* __restrict__ x_prev, const double* __restrict__ logw_prev,
                                                   const double* __restrict__ ref_t, int corrected, double* __restrict__ x_new, ScanBufs sb) {
    __shared__ ScanSmem sm;
    const TransParams tp = *tpp;
    const int seg = blockIdx.x, tid = threadIdx.x;
    double xv[PG_PPT][NX], lwp[PG_PPT], lw[2][PG_PPT];
    load_particles<NX>(md, x_prev, seg, xv);
#pragma unroll
    for (int r = 0; r < PG_PPT; ++r) {
        const int64_t pi = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
        lwp[r] = (logw_prev != nullptr && pi < md.N) ? logw_prev[pi] : 0.0;
    }
    front_particles<NX, D, JIN, P>(md, tp, t, seed, ref_t, seg, xv, lwp, corrected, x_new, sb.laux, lw);
    segment_scan<2>(sm, lw, seg, sb.nsegp, sb.c1, sb.c2, sb.segk_w, sb.segs_w);
}

// ------------------------------------------------------------------------------------------
// The sweep (condSequentialMonteCarlo.__call__, src/PGAS.py:176-228) as two decoupled pipelines.
//
// k_propagate  advances every particle through time steps [t0, t1) with its state in registers:
//              no dependence on the weights or ancestors (quirk Q1), hence no barrier, no LDS, no
//              inter-workgroup traffic.  Per particle-step it writes x_t (trace), la_t = log p(y_t|aux_t),
//              h_t = log N(ref_t; aux_t, S) and ln_t = log p(y_t | x_t): everything the weight
//              recursion needs later.
//              The weight recursion itself is k_step + k_groups (pgas_resample.hip.h).
// ------------------------------------------------------------------------------------------
// One group of P particles of this thread through one time step, start to finish (basis, transition mean, noise, the three
// log-densities, new state): the FAST k_propagate walks its PG_PPT particles group by group so that only one group's
// intermediates are live at a time (the all-at-once form needs ~165 VGPRs, this one fits four waves per SIMD).
// everything of a group's step after the transition means: noise, the three log-densities, the new states
template <int NX, int P>
__device__ __forceinline__ void propagate_finish(const DevModel& md, const TransParams& tp, int t, uint64_t seed, const double (&rf)[NX],
                                                 const double* __restrict__ yt, int seg, int r0, const double (&aux)[P][NX],
                                                 double (&xn)[P][NX], double (&la)[P], double (&h)[P], double (&ln)[P]) {
    const int tid = threadIdx.x;
    double z0[P], z1[P];
    {
        pgas_u32x4 w[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int64_t pi = (int64_t)seg * PGAS_SEG + (r0 + p) * PG_BLK + tid;
            w[p] = pgas_rng_block(seed, PGAS_STREAM_PROP, 0u, (uint32_t)t, (uint64_t)(md.p0 + pi));
        }
        pgas_normal_pair_n(w, z0, z1, P);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int64_t pi = (int64_t)seg * PGAS_SEG + (r0 + p) * PG_BLK + tid;
        la[p] = loglik<NX>(md, yt, aux[p]);
        double quad = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            double w = 0.0;
#pragma unroll
            for (int l = 0; l <= k; ++l) w = PGAS_FMA(tp.LSinv[k * NX + l], rf[l] - aux[p][l], w);
            quad = PGAS_FMA(w, w, quad);
        }
        h[p] = PGAS_FMA(-0.5, quad, tp.cS);
        const double z[2] = {z0[p], z1[p]};
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            double v = aux[p][k];
#pragma unroll
            for (int l = 0; l <= k; ++l) v = PGAS_FMA(tp.LS[k * NX + l], z[l], v);
            xn[p][k] = (md.p0 + pi == md.Ng - 1) ? rf[k] : v;
        }
        ln[p] = loglik<NX>(md, yt, xn[p]);
    }
}
template <int NX, int D, int JIN, int P, int J0T>
__device__ __forceinline__ void propagate_group(const DevModel& md, const TransParams& tp, const double* G, int t, uint64_t seed, const double (&rf)[NX],
                                                const double* __restrict__ yt, const double* __restrict__ ut, int seg, int r0,
                                                const double (&xin)[P][NX], double (&xn)[P][NX], double (&la)[P], double (&h)[P], double (&ln)[P]) {
    double aux[P][NX];
    eval_mean<NX, D, JIN, P, J0T>(md, G, ut, xin, aux);
    propagate_finish<NX, P>(md, tp, t, seed, rf, yt, seg, r0, aux, xn, la, h, ln);
}

// PPT = particles per thread: PPT / 4 segments per workgroup, grid = ceil(nseg / (PPT / 4)).  A thread walks its particles in
// groups of P (propagate_group).  Measured on the SingleMassOscillator sweep: PPT = 8 (512 workgroups at N = 2^20, all resident
// at once) beats 4 by 6-9 % when the weight recursion runs beside it.
// ONE = the launch covers exactly one time step (t1 == t0 + 1; what the sweep uses for the cheap bases): a group's state is then
// loaded right before its pass instead of being held for the whole launch (8 particles x 2 doubles = 32 VGPRs less).
// MX: the 3-D contraction on the matrix cores (eval_mean_mx above; P == 1: a wave's 64 particles per pass): `mxi` = the tile
// descriptor, `gimg` = the MFMA-operand image of the coefficient tensor (k_pack_mx).
template <int NX, int D, int JIN, int P, int J0T, int PPT, bool ONE, uint64_t MXD = 0>
__device__ __forceinline__ void propagate_kernel(const DevModel& md, const TransParams* __restrict__ tpp, const double* __restrict__ G_arg, const SweepParams* __restrict__ swp, int t0, int t1,
                                                 const double* __restrict__ x_prev /* row t0-1 */, double* __restrict__ x_rows /* rows t0 ... */,
                                                 const double* __restrict__ ref,
                                                 double* __restrict__ la_rows, double* __restrict__ h_rows,
                                                 double* __restrict__ ln_rows /* rows t0 ... of the hand-off buffers */,
                                                 const MxInfo* __restrict__ mxi = nullptr, const double* __restrict__ gimg = nullptr) {
    const int tid = threadIdx.x;
    // Device-resident parameters (a replayed graph sees the current ones), read through the constant address space: nothing writes them
    // while the kernel runs.  The coefficient tensor's ADDRESS stays a kernel argument: taken from memory (tp.G), the tensor's scalar
    // loads lose their kernarg provenance and the schedule of this kernel went from 140 to 193 registers -- one k_step wave per SIMD
    // beside it instead of two, 83 instead of 69 ms per sweep.
    TransParams tp;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        tp.LS[q] = ld_const(&tpp->LS[q]);
        tp.LSinv[q] = ld_const(&tpp->LSinv[q]);
    }
    tp.cS = ld_const(&tpp->cS);
    tp.G = G_arg;
    const uint64_t seed = ld_const(&swp->seed);
    const size_t row = (size_t)md.N * NX, np = (size_t)md.nseg * PGAS_SEG;
    // 3-D bases: the coefficient tensor (J0 x J1 x JIN x NX doubles, 23 KB for the 729-function bases) is copied to LDS once per launch
    // and read from there with wave-uniform (broadcast) addresses: per row of the frequency grid the scalar-load path would stall
    // on a fresh s_load (24 coefficients for 48 FMAs), LDS reads pipeline behind the FMAs.
    const double* Guse = tp.G;
    double* mx_tab = nullptr;
    constexpr bool MX = MXD != 0;
    if constexpr (MX) {
        static_assert(D == 3 && P == 1, "matrix-core contraction: 3-D bases, one particle per lane and pass");
        extern __shared__ __attribute__((aligned(16))) double pg_g_lds[];
        const int nimg = ld_const(&mxi->nslots) * 64;
        for (int i = tid; i < nimg; i += PG_BLK) pg_g_lds[i] = gimg[i];
        __syncthreads();
        Guse = pg_g_lds;
        mx_tab = pg_g_lds + nimg + (tid >> 6) * (64 * PG_MX_TSTRIDE);
    } else if constexpr (D == 3) {
        extern __shared__ __attribute__((aligned(16))) double pg_g_lds[];
        const int gtot = md.J[0] * md.J[1] * JIN * NX;
        for (int i = tid; i < gtot; i += PG_BLK) pg_g_lds[i] = tp.G[i];
        __syncthreads();
        Guse = pg_g_lds;
    }
    {
        static_assert(PPT % PG_PPT == 0 && PPT % P == 0, "whole segments per workgroup, whole groups per thread");
        const int seg0 = blockIdx.x * (PPT / PG_PPT);
        // particle r of this thread: segment seg0 + r / 4, row (r % 4) * 256 + tid inside it (the layout every other kernel uses)
        auto particle = [&](int r) { return (size_t)(seg0 + r / PG_PPT) * PGAS_SEG + (size_t)(r % PG_PPT) * PG_BLK + tid; };
        auto load_state = [&](int r, double (&xr)[NX]) {
            size_t pi = particle(r);
            if (pi >= (size_t)md.N) pi = md.N - 1;
            if constexpr (NX == 2) {
#ifdef PG_NT_LOADS
                const pg_nt_d2 v = __builtin_nontemporal_load(reinterpret_cast<const pg_nt_d2*>(x_prev) + pi);
#else
                const double2 v = reinterpret_cast<const double2*>(x_prev)[pi];
#endif
                xr[0] = v.x;
                xr[1] = v.y;
            } else {
#pragma unroll
                for (int k = 0; k < NX; ++k) xr[k] = x_prev[pi * NX + k];
            }
        };
        double xv[ONE ? 1 : PPT][NX];
        if constexpr (!ONE) {
#pragma unroll
            for (int r = 0; r < PPT; ++r) load_state(r, xv[r]);
        }
        for (int t = t0; t < (ONE ? t0 + 1 : t1); ++t) {
            const double* __restrict__ yt = md.y + (size_t)t * md.ny;
            const double* __restrict__ ut = md.u + (size_t)t * md.nu;
            double rf[NX];
#pragma unroll
            for (int k = 0; k < NX; ++k) rf[k] = ref[(size_t)t * NX + k];
            double* __restrict__ xt = x_rows + (size_t)(t - t0) * row;
            const size_t hrow = (size_t)(t - t0) * np;
            // ONE: kept rolled -- four unrolled copies of the pass are 35 KB of code, and k_step's 60 KB run beside it on an
            // instruction cache of 64 KB per CU pair
            constexpr int kUnroll = ONE ? 1 : PPT / P;
            double xnext[P][NX];   // ONE: state of the NEXT group, loaded while this one computes (its HBM latency would otherwise
                                    // be exposed once per group: nothing else in this kernel waits on memory)
            if constexpr (ONE) {
#pragma unroll
                for (int p = 0; p < P; ++p) load_state(p, xnext[p]);
            }
#pragma unroll kUnroll
            for (int r0 = 0; r0 < PPT; r0 += P) {
                if (seg0 + r0 / PG_PPT >= md.nseg) break;   // uniform: odd segment count, second half of the last workgroup
                double xin[P][NX], xn[P][NX], la[P], h[P], ln[P];
#pragma unroll
                for (int p = 0; p < P; ++p) {
#pragma unroll
                    for (int k = 0; k < NX; ++k) xin[p][k] = ONE ? xnext[p][k] : xv[ONE ? 0 : r0 + p][k];
                }
                if constexpr (ONE) {
                    const int rn = r0 + P < PPT ? r0 + P : r0;   // the last group re-reads its own rows (harmless, keeps the loop uniform)
#pragma unroll
                    for (int p = 0; p < P; ++p) load_state(rn + p, xnext[p]);
                }
                if constexpr (MX) {
                    double aux[P][NX];
                    eval_mean_mx<NX, JIN, MXD>(md, Guse, mx_tab, ut, xin[0], aux[0]);
                    propagate_finish<NX, P>(md, tp, t, seed, rf, yt, seg0 + r0 / PG_PPT, r0 % PG_PPT, aux, xn, la, h, ln);
                } else {
                    propagate_group<NX, D, JIN, P, J0T>(md, tp, Guse, t, seed, rf, yt, ut, seg0 + r0 / PG_PPT, r0 % PG_PPT, xin, xn, la, h, ln);
                }
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const size_t pi = particle(r0 + p);   // la / h / ln are padded to nseg*SEG
                    if (pi < (size_t)md.N) {
                        if constexpr (NX == 2) {
#ifdef PG_NT_X
                            typedef double pg_d2 __attribute__((ext_vector_type(2)));
                            pg_d2 v2 = {xn[p][0], xn[p][1]};
                            __builtin_nontemporal_store(v2, reinterpret_cast<pg_d2*>(xt) + pi);
#else
                            reinterpret_cast<double2*>(xt)[pi] = make_double2(xn[p][0], xn[p][1]);
#endif
                        } else {
#pragma unroll
                            for (int k = 0; k < NX; ++k) xt[pi * NX + k] = xn[p][k];
                        }
                    }
#ifdef PG_NT_H
                    __builtin_nontemporal_store(la[p], &la_rows[hrow + pi]);
                    __builtin_nontemporal_store(h[p], &h_rows[hrow + pi]);
                    __builtin_nontemporal_store(ln[p], &ln_rows[hrow + pi]);
#else
                    la_rows[hrow + pi] = la[p];
                    h_rows[hrow + pi] = h[p];
                    ln_rows[hrow + pi] = ln[p];
#endif
                    if constexpr (!ONE) {
#pragma unroll
                        for (int k = 0; k < NX; ++k) xv[r0 + p][k] = xn[p][k];
                    }
                }
            }
        }
    }
}

template <int NX, int D, int JIN, int P, int W, int J0T = 0, int PPT = PG_PPT, bool ONE = false>
__global__ __launch_bounds__(PG_BLK, W) void k_propagate(DevModel md, const TransParams* __restrict__ tpp, const double* __restrict__ G, const SweepParams* __restrict__ swp, int t0, int t1,
                                                          const double* __restrict__ x_prev, double* __restrict__ x_rows, const double* __restrict__ ref,
                                                          double* __restrict__ la_rows, double* __restrict__ h_rows, double* __restrict__ ln_rows) {
    propagate_kernel<NX, D, JIN, P, J0T, PPT, ONE>(md, tpp, G, swp, t0, t1, x_prev, x_rows, ref, la_rows, h_rows, ln_rows);
}
template <int NX, int JIN, int W, int PPT, bool ONE, uint64_t MXD>
__global__ __launch_bounds__(PG_BLK, W) void k_propagate_mx(DevModel md, const TransParams* __restrict__ tpp, const double* __restrict__ G, const SweepParams* __restrict__ swp, int t0, int t1,
                                                             const double* __restrict__ x_prev, double* __restrict__ x_rows, const double* __restrict__ ref,
                                                             double* __restrict__ la_rows, double* __restrict__ h_rows, double* __restrict__ ln_rows,
                                                             const MxInfo* __restrict__ mxi, const double* __restrict__ gimg) {
    propagate_kernel<NX, 3, JIN, 1, 0, PPT, ONE, MXD>(md, tpp, G, swp, t0, t1, x_prev, x_rows, ref, la_rows, h_rows, ln_rows, mxi, gimg);
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
    segment_scan<1>(sm, lw, seg, sb.nsegp, sb.c1, nullptr, sb.segk_w, sb.segs_w);
}

// ------------------------------------------------------------------------------------------
// k_backtrace: reconstruct_trajectory (src/Filtering.py:40-55), one lane chases the ancestors
// ------------------------------------------------------------------------------------------
__global__ void k_backtrace(int Nl, int T, int nx, BtTab tab, int world, const UpperHdr* __restrict__ hdr, double* __restrict__ traj) {
    // b is a GLOBAL particle index; its row lives on rank b / Nl (single device: rank 0, Nl = N).  The block table goes to LDS
    // first when it fits (it is on the dependent chain of every hop), then one lane chases.
    extern __shared__ const void* pg_bt_lds[];
    const bool in_lds = tab.entries <= 2048;
    if (in_lds) {
        for (int i = threadIdx.x; i < tab.entries; i += blockDim.x) pg_bt_lds[i] = tab.blk[i];
        __syncthreads();
    }
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const void* const* blk = in_lds ? pg_bt_lds : tab.blk;
    const int mx = (1 << tab.shift_x) - 1, ma = (1 << tab.shift_anc) - 1;
    int64_t b = hdr->final_idx;
    for (int i = T - 1; i >= 0; --i) {
        const int r = world > 1 ? (int)(b / Nl) : 0;
        const int64_t bl = b - (int64_t)r * Nl;
        const double* __restrict__ xr = (const double*)blk[(size_t)(r * 2) * tab.nblk_max + (i >> tab.shift_x)] + (size_t)(i & mx) * Nl * nx;
        for (int k = 0; k < nx; ++k) traj[(size_t)i * nx + k] = xr[(size_t)bl * nx + k];
        if (i > 0) {
            // ancestor of particle b of time i: row i-1 of the ancestor trace, indexed by the CHILD (time i) particle
            const int32_t* __restrict__ ar = (const int32_t*)blk[(size_t)(r * 2 + 1) * tab.nblk_max + ((i - 1) >> tab.shift_anc)] + (size_t)((i - 1) & ma) * Nl;
            b = ar[bl];
        }
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
__global__ __launch_bounds__(PG_BLK) void k_aux(DevModel md, const TransParams* __restrict__ tpp, int t, const double* __restrict__ x, double* __restrict__ aux_out) {
    const int seg = blockIdx.x, tid = threadIdx.x;
    const TransParams tp = *tpp;
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
