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
//   k_fused       k_back(t-1) + k_front(t) in one launch (the sweep's steady state)
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
static_assert(PG_BLK * PG_PPT == PGAS_SEG, "one workgroup owns one canonical segment");
#define PG_UPPER_THREADS 1024
#define PG_MAX_NSEG 8192

struct DevModel {
    int32_t N, T, nx, ny, nu, D, M;
    int32_t J[PGAS_MAX_D], j0[PGAS_MAX_D], jstep[PGAS_MAX_D], sel[PGAS_MAX_D];
    double alpha[PGAS_MAX_D], beta[PGAS_MAX_D];
    double nrm;
    double H[PGAS_MAX_NY * 2];
    double LRinv[PGAS_MAX_NY * PGAS_MAX_NY];
    double cR;
    int32_t JP;       // padded innermost grid extent
    int32_t nseg;     // ceil(N / PGAS_SEG)
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
};

struct ScanBufs {      // per-step scan scratch (device)
    double* laux;      // (nseg*SEG)
    uint64_t* c1;      // (nseg*SEG) quantised inclusive cumsum of the resampling weights
    uint64_t* c2;      // (nseg*SEG) same for the ancestor weights
    double* segm;      // (2, nsegp) segment maxima
    uint64_t* segs;    // (2, nsegp) segment totals
    double* excl;      // (2, nsegp)
    double* scale;     // (2, nsegp)
    double* cm;        // (2, nsegp) running max of the segment-end CDF numerators
    UpperHdr* hdr;
    int32_t nsegp;     // padded nseg (multiple of 64)
};

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = __builtin_fmax(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        int o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint64_t o = __shfl_up((unsigned long long)v, off);
        if (lane >= off) v += o;
    }
    return v;
}
// canonical Kogge-Stone inclusive scan of one group of 64
__device__ __forceinline__ double wave_ks_add(double v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        double o = __shfl_up(v, off);
        if (lane >= off) v = v + o;
    }
    return v;
}
__device__ __forceinline__ double wave_ks_max(double v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        double o = __shfl_up(v, off);
        if (lane >= off) v = __builtin_fmax(v, o);
    }
    return v;
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

// sin/cos of the first frequency and of the frequency step of basis dimension d
__device__ __forceinline__ void dim_start(const DevModel& md, int d, double r, double& sc, double& cc, double& sd, double& cd) {
    pgas_sincospi((double)md.j0[d] * r, &sc, &cc);
    pgas_sincospi((double)md.jstep[d] * r, &sd, &cd);
}
__device__ __forceinline__ void rotate(double& sc, double& cc, double sd, double cd) {
    double sn = PGAS_FMA(sc, cd, cc * sd);
    double cn = PGAS_FMA(cc, cd, -(sc * sd));
    sc = sn;
    cc = cn;
}

// ------------------------------------------------------------------------------------------
// aux = A phi(x, u_t) for P particles at once (src/PGAS.py:52-55).  The basis is separable
// (src/BasisFunctions.py:77-80), so the M products collapse into a nested contraction over the
// dense frequency grid; the coefficients are wave-uniform (scalar loads), the outer dimensions'
// sines come from a rotation recurrence, only the innermost dimension's table lives in registers.
// ------------------------------------------------------------------------------------------
template <int NX, int D, int JIN, int P>
__device__ __forceinline__ void eval_mean(const DevModel& md, const double* __restrict__ G, const double* __restrict__ ut,
                                          const double (&x)[P][NX], double (&aux)[P][NX]) {
    if constexpr (D == 1) {
        double sc[P], cc[P], sd[P], cd[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            double r = PGAS_FMA(pick_input<NX>(md, 0, x[p], ut), md.alpha[0], md.beta[0]);
            dim_start(md, 0, r, sc[p], cc[p], sd[p], cd[p]);
#pragma unroll
            for (int k = 0; k < NX; ++k) aux[p][k] = 0.0;
        }
        const int J0 = md.J[0];
        for (int a = 0; a < J0; ++a) {
            double g[NX];
#pragma unroll
            for (int k = 0; k < NX; ++k) g[k] = G[a * NX + k];
#pragma unroll
            for (int p = 0; p < P; ++p) {
#pragma unroll
                for (int k = 0; k < NX; ++k) aux[p][k] = PGAS_FMA(g[k], sc[p], aux[p][k]);
                rotate(sc[p], cc[p], sd[p], cd[p]);
            }
        }
    } else {
        // innermost dimension table
        constexpr int DI = D - 1;
        double tab[P][JIN];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            double r = PGAS_FMA(pick_input<NX>(md, DI, x[p], ut), md.alpha[DI], md.beta[DI]);
            double sc, cc, sd, cd;
            dim_start(md, DI, r, sc, cc, sd, cd);
#pragma unroll
            for (int q = 0; q < JIN; ++q) {
                tab[p][q] = sc;
                rotate(sc, cc, sd, cd);
            }
        }
        double s0[P], c0[P], sd0[P], cd0[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            double r = PGAS_FMA(pick_input<NX>(md, 0, x[p], ut), md.alpha[0], md.beta[0]);
            dim_start(md, 0, r, s0[p], c0[p], sd0[p], cd0[p]);
#pragma unroll
            for (int k = 0; k < NX; ++k) aux[p][k] = 0.0;
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
                    for (int k = 0; k < NX; ++k) aux[p][k] = PGAS_FMA(s0[p], in[p][k], aux[p][k]);
                    rotate(s0[p], c0[p], sd0[p], cd0[p]);
                }
            }
        } else {
            double s1s[P], c1s[P], sd1[P], cd1[P];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                double r = PGAS_FMA(pick_input<NX>(md, 1, x[p], ut), md.alpha[1], md.beta[1]);
                dim_start(md, 1, r, s1s[p], c1s[p], sd1[p], cd1[p]);
            }
            const int J1 = md.J[1];
            for (int a = 0; a < J0; ++a) {
                double mid[P][NX], s1[P], c1[P];
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    s1[p] = s1s[p];
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
                        for (int k = 0; k < NX; ++k) mid[p][k] = PGAS_FMA(s1[p], in[p][k], mid[p][k]);
                        rotate(s1[p], c1[p], sd1[p], cd1[p]);
                    }
                }
#pragma unroll
                for (int p = 0; p < P; ++p) {
#pragma unroll
                    for (int k = 0; k < NX; ++k) aux[p][k] = PGAS_FMA(s0[p], mid[p][k], aux[p][k]);
                    rotate(s0[p], c0[p], sd0[p], cd0[p]);
                }
            }
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
    pgas_rng_normals(seed, PGAS_STREAM_INIT, 0u, (uint64_t)p, NX, z);
    double xv[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        double v = m0L0[k];
#pragma unroll
        for (int l = 0; l <= k; ++l) v = PGAS_FMA(m0L0[NX + k * NX + l], z[l], v);
        xv[k] = v;
    }
    if (p == md.N - 1) {
#pragma unroll
        for (int k = 0; k < NX; ++k) xv[k] = ref0[k];
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) x0[p * NX + k] = xv[k];
}

// ------------------------------------------------------------------------------------------
// segment softmax scan shared by k_front / k_fused / k_segscan.
// lw[r] is the log-weight of particle seg*SEG + r*BLK + tid (-inf when past N).  Writes the
// quantised inclusive cumsum (index order) and the segment (max, total).
// ------------------------------------------------------------------------------------------
struct ScanSmem {
    uint64_t q[2][PGAS_SEG];
    double red[2][PG_BLK / 64];
    uint64_t wtot[2][PG_BLK / 64];
};

template <int NW>  // number of weight vectors scanned together (1 or 2)
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
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        double m = sm.red[w][0];
#pragma unroll
        for (int v = 1; v < PG_BLK / 64; ++v) m = __builtin_fmax(m, sm.red[w][v]);
        mx[w] = m;
#pragma unroll
        for (int r = 0; r < PG_PPT; ++r) {
            const double e = pgas_exp(lw[w][r] - m);
            const uint64_t q = (e > 0.0) ? pgas_double_to_u64(__builtin_rint(e * PGAS_FIX_SCALE)) : 0ull;
            sm.q[w][r * PG_BLK + tid] = q;
        }
    }
    __syncthreads();
    uint64_t loc[NW][PG_PPT], incl[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        uint64_t run = 0;
#pragma unroll
        for (int j = 0; j < PG_PPT; ++j) {
            run += sm.q[w][PG_PPT * tid + j];
            loc[w][j] = run;
        }
        incl[w] = wave_incl_scan_u64(run, lane);
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
        const uint64_t base = off + incl[w] - loc[w][PG_PPT - 1];
        ulonglong2* dst = reinterpret_cast<ulonglong2*>((w == 0 ? cA : cB) + (size_t)seg * PGAS_SEG + PG_PPT * tid);
        dst[0] = make_ulonglong2(base + loc[w][0], base + loc[w][1]);
        dst[1] = make_ulonglong2(base + loc[w][2], base + loc[w][3]);
        if (tid == 0) {
            segm[(size_t)w * nsegp + seg] = mx[w];
            segs[(size_t)w * nsegp + seg] = tot;
        }
    }
}

// ------------------------------------------------------------------------------------------
// front half of a step for the PG_PPT particles of this thread (src/PGAS.py:90-118,130-134)
// ------------------------------------------------------------------------------------------
template <int NX, int D, int JIN, int P>
__device__ __forceinline__ void front_particles(const DevModel& md, const TransParams& tp, int t, uint64_t seed,
                                                const double* __restrict__ ref_t, int seg, const double (&xprev)[PG_PPT][NX],
                                                const double (&logw)[PG_PPT], double* __restrict__ x_new,
                                                double* __restrict__ laux_out, double (&lw)[2][PG_PPT]) {
    const int tid = threadIdx.x;
    const double* __restrict__ yt = md.y + (size_t)t * md.ny;
    const double* __restrict__ ut = md.u + (size_t)t * md.nu;
    double rf[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) rf[k] = ref_t[k];
#pragma unroll
    for (int r0 = 0; r0 < PG_PPT; r0 += P) {
        double xin[P][NX], aux[P][NX];
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int k = 0; k < NX; ++k) xin[p][k] = xprev[r0 + p][k];
        eval_mean<NX, D, JIN, P>(md, tp.G, ut, xin, aux);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int r = r0 + p;
            const int64_t pi = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
            const bool valid = pi < md.N;
            const double la = loglik<NX>(md, yt, aux[p]);
            const double l1 = la + logw[r];
            double quad = 0.0;
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                double w = 0.0;
#pragma unroll
                for (int l = 0; l <= k; ++l) w = PGAS_FMA(tp.LSinv[k * NX + l], rf[l] - aux[p][l], w);
                quad = PGAS_FMA(w, w, quad);
            }
            const double l2 = l1 + PGAS_FMA(-0.5, quad, tp.cS);
            double z[2];
            pgas_rng_normals(seed, PGAS_STREAM_PROP, (uint32_t)t, (uint64_t)pi, NX, z);
            double xn[NX];
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                double v = aux[p][k];
#pragma unroll
                for (int l = 0; l <= k; ++l) v = PGAS_FMA(tp.LS[k * NX + l], z[l], v);
                xn[k] = (pi == md.N - 1) ? rf[k] : v;
            }
            if (valid) {
                if constexpr (NX == 2) {
                    reinterpret_cast<double2*>(x_new)[pi] = make_double2(xn[0], xn[1]);
                } else {
#pragma unroll
                    for (int k = 0; k < NX; ++k) x_new[pi * NX + k] = xn[k];
                }
                laux_out[pi] = la;
            }
            lw[0][r] = valid ? l1 : -__builtin_inf();
            lw[1][r] = valid ? l2 : -__builtin_inf();
        }
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
                                                   const double* __restrict__ ref_t, double* __restrict__ x_new, ScanBufs sb) {
    __shared__ ScanSmem sm;
    const int seg = blockIdx.x, tid = threadIdx.x;
    double xv[PG_PPT][NX], lwp[PG_PPT], lw[2][PG_PPT];
    load_particles<NX>(md, x_prev, seg, xv);
#pragma unroll
    for (int r = 0; r < PG_PPT; ++r) {
        const int64_t pi = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
        lwp[r] = (logw_prev != nullptr && pi < md.N) ? logw_prev[pi] : 0.0;
    }
    front_particles<NX, D, JIN, P>(md, tp, t, seed, ref_t, seg, xv, lwp, x_new, sb.laux, lw);
    segment_scan<2>(sm, lw, seg, sb.nsegp, sb.c1, sb.c2, sb.segm, sb.segs);
}

// ------------------------------------------------------------------------------------------
// k_upper: cross-segment CDF.  Block w (0: resampling weights, 1: ancestor weights) turns the
// segment (max, total) pairs into exclusive prefixes in the canonical KS64-tree order, the
// running maximum of the segment-end numerators and the normaliser S; block `search_block`
// additionally counts #{k : W_k < u S} (ancestor of the reference particle, src/PGAS.py:121-124;
// or the final index, :225).
// ------------------------------------------------------------------------------------------
struct UpperSmem {                 // carved from dynamic LDS: inc[ninc] first (ninc = nseg rounded up to 1024)
    double* inc;                  // level-0 inclusive values, later the running max cm
    double* ga;                   // [PG_MAX_NSEG/64] level-0 group totals -> level-1 inclusive
    double* gb;                   // [64] level-1 group totals -> level-2 inclusive
    double* red;                  // [PG_UPPER_THREADS/64]
    int* cnt;
};
static inline size_t upper_smem_bytes(int nseg) {
    const size_t ninc = ((size_t)nseg + PG_UPPER_THREADS - 1) / PG_UPPER_THREADS * PG_UPPER_THREADS;
    return (ninc + PG_MAX_NSEG / 64 + 64 + PG_UPPER_THREADS / 64) * sizeof(double) + 16;
}

__global__ __launch_bounds__(PG_UPPER_THREADS) void k_upper(int N, int nseg, ScanBufs sb, int search_block, double u_search,
                                                             int final_mode) {
    extern __shared__ __attribute__((aligned(16))) char upper_raw[];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    UpperSmem sm;
    {
        const size_t ninc = ((size_t)nseg + PG_UPPER_THREADS - 1) / PG_UPPER_THREADS * PG_UPPER_THREADS;
        sm.inc = reinterpret_cast<double*>(upper_raw);
        sm.ga = sm.inc + ninc;
        sm.gb = sm.ga + PG_MAX_NSEG / 64;
        sm.red = sm.gb + 64;
        sm.cnt = reinterpret_cast<int*>(sm.red + PG_UPPER_THREADS / 64);
    }
    const int nsegp = sb.nsegp;
    const double* __restrict__ segm = sb.segm + (size_t)w * nsegp;
    const uint64_t* __restrict__ segs = sb.segs + (size_t)w * nsegp;
    double* __restrict__ excl = sb.excl + (size_t)w * nsegp;
    double* __restrict__ scale = sb.scale + (size_t)w * nsegp;
    double* __restrict__ cm = sb.cm + (size_t)w * nsegp;
    const int nchunk = (nseg + PG_UPPER_THREADS - 1) / PG_UPPER_THREADS;

    // 1. global max of the segment maxima
    double g = -__builtin_inf();
    for (int b = tid; b < nseg; b += PG_UPPER_THREADS) g = __builtin_fmax(g, segm[b]);
    g = wave_max(g);
    if (lane == 0) sm.red[wave] = g;
    if (tid == 0) *sm.cnt = 0;
    __syncthreads();
    g = sm.red[0];
    for (int v = 1; v < PG_UPPER_THREADS / 64; ++v) g = __builtin_fmax(g, sm.red[v]);
    __syncthreads();

    // 2. scaled totals and level-0 scans (one wave = one group of 64 consecutive segments)
    double tot[PG_MAX_NSEG / PG_UPPER_THREADS], incA[PG_MAX_NSEG / PG_UPPER_THREADS];
#pragma unroll
    for (int e = 0; e < PG_MAX_NSEG / PG_UPPER_THREADS; ++e) {
        tot[e] = 0.0;
        incA[e] = 0.0;
        if (e < nchunk) {
            const int b = e * PG_UPPER_THREADS + tid;
            if (b < nseg) {
                double sc = pgas_exp(segm[b] - g);
                if (!(sc >= 0.0)) sc = 0.0;
                scale[b] = sc;
                tot[e] = sc * (pgas_u64_to_double(segs[b]) * PGAS_FIX_INV);
            }
            incA[e] = wave_ks_add(tot[e], lane);
            sm.inc[b] = incA[e];
            if (lane == 63) sm.ga[b >> 6] = incA[e];
        }
    }
    __syncthreads();
    // 3. level 1: groups of 64 level-0 totals
    const int n1 = (nseg + 63) >> 6, n2 = (n1 + 63) >> 6;
    if (wave < n2) {
        const int gi = wave * 64 + lane;
        double v = gi < n1 ? sm.ga[gi] : 0.0;
        v = wave_ks_add(v, lane);
        __builtin_amdgcn_wave_barrier();
        sm.ga[gi] = v;  // ga is sized for n2*64 entries
        if (lane == 63) sm.gb[wave] = v;
    }
    __syncthreads();
    // 4. level 2
    if (wave == 0) {
        double v = lane < n2 ? sm.gb[lane] : 0.0;
        v = wave_ks_add(v, lane);
        sm.gb[lane] = v;
    }
    __syncthreads();
    // 5. exclusive prefixes, segment-end numerators
    double wend[PG_MAX_NSEG / PG_UPPER_THREADS];
#pragma unroll
    for (int e = 0; e < PG_MAX_NSEG / PG_UPPER_THREADS; ++e) {
        wend[e] = 0.0;
        if (e < nchunk) {
            const int b = e * PG_UPPER_THREADS + tid;
            const int gq = b >> 6, h = gq >> 6;
            const double eC = (h & 63) ? sm.gb[h - 1] : 0.0;
            const double eB = (gq & 63) ? sm.ga[gq - 1] : 0.0;
            const double eA = (b & 63) ? sm.inc[b - 1] : 0.0;
            const double ex = (eC + eB) + eA;
            if (b < nseg) {
                excl[b] = ex;
                wend[e] = ex + tot[e];
            }
        }
    }
    __syncthreads();
    // 6. running maximum of W_end (exact, any order)
    double carry = 0.0;
#pragma unroll
    for (int e = 0; e < PG_MAX_NSEG / PG_UPPER_THREADS; ++e) {
        if (e < nchunk) {  // uniform
            const int b = e * PG_UPPER_THREADS + tid;
            double v = wave_ks_max(wend[e], lane);
            if (lane == 63) sm.red[wave] = v;
            __syncthreads();
            double pre = carry, all = carry;
            for (int q = 0; q < PG_UPPER_THREADS / 64; ++q) {
                const double tq = sm.red[q];
                if (q < wave) pre = __builtin_fmax(pre, tq);
                all = __builtin_fmax(all, tq);
            }
            v = __builtin_fmax(v, pre);
            if (b < nseg) {
                cm[b] = v;
                sm.inc[b] = v;
            }
            carry = all;
            __syncthreads();
        }
    }
    const double S = carry;
    const bool valid = (S > 0.0) && (S < __builtin_inf());
    if (tid == 0) {
        sb.hdr->S[w] = S;
        sb.hdr->valid[w] = valid ? 1 : 0;
    }
    if (w != search_block) return;

    // 7. #{k : W_k < tau}
    int result = N - 1;
    if (valid) {
        const double tau = u_search * S;
        int c = 0;
        for (int b = tid; b < nseg; b += PG_UPPER_THREADS) c += (sm.inc[b] < tau) ? 1 : 0;
        for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
        if (lane == 0 && c) atomicAdd(sm.cnt, c);
        __syncthreads();
        const int bs = *sm.cnt;
        __syncthreads();
        if (bs < nseg) {
            const int64_t base = (int64_t)bs * PGAS_SEG;
            const int n = (N - base) < PGAS_SEG ? (int)(N - base) : PGAS_SEG;
            const uint64_t* __restrict__ c = (w == 0 ? sb.c1 : sb.c2) + base;
            const double ex = excl[bs], sc = scale[bs], cy = bs ? sm.inc[bs - 1] : 0.0;
            if (tid == 0) *sm.cnt = 0;
            __syncthreads();
            int k = 0;
            for (int i = tid; i < n; i += PG_UPPER_THREADS) {
                double num = ex + sc * (pgas_u64_to_double(c[i]) * PGAS_FIX_INV);
                num = __builtin_fmax(num, cy);
                k += (num < tau) ? 1 : 0;
            }
            for (int off = 32; off >= 1; off >>= 1) k += __shfl_xor(k, off);
            if (lane == 0 && k) atomicAdd(sm.cnt, k);
            __syncthreads();
            const int64_t r = base + *sm.cnt;
            result = r > N - 1 ? N - 1 : (int)r;
        }
    }
    if (tid == 0) {
        if (final_mode)
            sb.hdr->final_idx = result;
        else
            sb.hdr->ref_idx = result;
    }
}

// ------------------------------------------------------------------------------------------
// back half: systematic resampling search (src/Filtering.py:28-35) + weight update
// (src/PGAS.py:137-147) for the PG_PPT slots of this thread.
// ------------------------------------------------------------------------------------------
struct BackSmem {
    uint64_t c[PGAS_SEG];
    int red[PG_BLK / 64];
};

// cmS: cm array of the resampling CDF staged in LDS (nseg doubles)
template <int NX>
__device__ __forceinline__ void back_slots(const DevModel& md, BackSmem& sm, const double* __restrict__ cmS, int t, double u1,
                                           const ScanBufs& sb, int seg, const double (&xcur)[PG_PPT][NX], int32_t* __restrict__ anc_out,
                                           double (&logw_new)[PG_PPT]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nseg = md.nseg, N = md.N;
    const double S = sb.hdr->S[0];
    const bool valid = sb.hdr->valid[0] != 0;
    double tau[PG_PPT];
    int bi[PG_PPT], a[PG_PPT];
#pragma unroll
    for (int r = 0; r < PG_PPT; ++r) {
        const int64_t i = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
        const double Ui = (u1 + (double)i) / (double)N;
        tau[r] = Ui * S;
        // b_i = #{b : cm[b] < tau}
        int lo = 0, hi = nseg;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cmS[mid] < tau[r]) lo = mid + 1; else hi = mid;
        }
        bi[r] = (i < N && valid) ? lo : 0x7fffffff;
        a[r] = (i < N) ? (valid ? N - 1 : (int)i) : 0;
    }
    // visit the distinct segments referenced by this workgroup's slots in increasing order
    int cur = -1;
    while (true) {
        int nxt = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < PG_PPT; ++r) nxt = (bi[r] > cur && bi[r] < nxt) ? bi[r] : nxt;
        nxt = wave_min_i(nxt);
        __syncthreads();  // previous iteration's readers are done with sm.c / sm.red
        if (lane == 0) sm.red[wave] = nxt;
        __syncthreads();
        nxt = sm.red[0];
#pragma unroll
        for (int v = 1; v < PG_BLK / 64; ++v) nxt = sm.red[v] < nxt ? sm.red[v] : nxt;
        if (nxt >= nseg) break;  // uniform: all remaining slots map past the last segment (a = N-1) or are done
        cur = nxt;
        const int64_t base = (int64_t)cur * PGAS_SEG;
        const int n = (N - base) < PGAS_SEG ? (int)(N - base) : PGAS_SEG;
        {
            const ulonglong2* src = reinterpret_cast<const ulonglong2*>(sb.c1 + base);
            ulonglong2* dst = reinterpret_cast<ulonglong2*>(sm.c);
            dst[tid] = src[tid];
            dst[tid + PG_BLK] = src[tid + PG_BLK];
        }
        __syncthreads();
        const double ex = sb.excl[cur], sc = sb.scale[cur], cy = cur ? cmS[cur - 1] : 0.0;
#pragma unroll
        for (int r = 0; r < PG_PPT; ++r) {
            if (bi[r] == cur) {
                int lo = 0, hi = n;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    double num = ex + sc * (pgas_u64_to_double(sm.c[mid]) * PGAS_FIX_INV);
                    num = __builtin_fmax(num, cy);
                    if (num < tau[r]) lo = mid + 1; else hi = mid;
                }
                const int64_t ai = base + lo;
                a[r] = ai > N - 1 ? N - 1 : (int)ai;
            }
        }
    }
    const double* __restrict__ yt = md.y + (size_t)t * md.ny;
#pragma unroll
    for (int r = 0; r < PG_PPT; ++r) {
        const int64_t i = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
        logw_new[r] = 0.0;
        if (i < N) {
            if (i == N - 1) a[r] = sb.hdr->ref_idx;  // src/PGAS.py:127
            anc_out[i] = a[r];
            logw_new[r] = loglik<NX>(md, yt, xcur[r]) - sb.laux[a[r]];
        }
    }
}

__device__ __forceinline__ void stage_cm(double* cmS, const double* __restrict__ cm, int nseg) {
    for (int b = threadIdx.x; b < nseg; b += PG_BLK) cmS[b] = cm[b];
    __syncthreads();
}

template <int NX>
__global__ __launch_bounds__(PG_BLK) void k_back(DevModel md, int t, double u1, const double* __restrict__ x_cur, ScanBufs sb,
                                                  int32_t* __restrict__ anc_out, double* __restrict__ logw_out) {
    __shared__ BackSmem sm;
    extern __shared__ __attribute__((aligned(16))) char dyn_raw[];
    double* cmS = reinterpret_cast<double*>(dyn_raw);
    const int seg = blockIdx.x, tid = threadIdx.x;
    stage_cm(cmS, sb.cm, md.nseg);
    double xv[PG_PPT][NX], lwn[PG_PPT];
    load_particles<NX>(md, x_cur, seg, xv);
    back_slots<NX>(md, sm, cmS, t, u1, sb, seg, xv, anc_out, lwn);
#pragma unroll
    for (int r = 0; r < PG_PPT; ++r) {
        const int64_t i = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
        if (i < md.N) logw_out[i] = lwn[r];
    }
}

// ------------------------------------------------------------------------------------------
// k_fused: back half of step t-1 followed by the front half of step t for the same particles
// (the log-weight never leaves registers).  sb_prev holds step t-1's scan results, sb_next
// receives step t's.
// ------------------------------------------------------------------------------------------
template <int NX, int D, int JIN, int P>
__global__ __launch_bounds__(PG_BLK) void k_fused(DevModel md, TransParams tp, int t, uint64_t seed, double u1_prev,
                                                   const double* __restrict__ x_prev, const double* __restrict__ ref_t,
                                                   double* __restrict__ x_new, ScanBufs sb_prev, ScanBufs sb_next,
                                                   int32_t* __restrict__ anc_out, double* __restrict__ logw_trace_row) {
    __shared__ union {
        BackSmem b;
        ScanSmem s;
    } sm;
    extern __shared__ __attribute__((aligned(16))) char dyn_raw[];
    double* cmS = reinterpret_cast<double*>(dyn_raw);
    const int seg = blockIdx.x, tid = threadIdx.x;
    stage_cm(cmS, sb_prev.cm, md.nseg);
    double xv[PG_PPT][NX], lwp[PG_PPT], lw[2][PG_PPT];
    load_particles<NX>(md, x_prev, seg, xv);
    back_slots<NX>(md, sm.b, cmS, t - 1, u1_prev, sb_prev, seg, xv, anc_out, lwp);
    if (logw_trace_row != nullptr) {
#pragma unroll
        for (int r = 0; r < PG_PPT; ++r) {
            const int64_t i = (int64_t)seg * PGAS_SEG + r * PG_BLK + tid;
            if (i < md.N) logw_trace_row[i] = lwp[r];
        }
    }
    front_particles<NX, D, JIN, P>(md, tp, t, seed, ref_t, seg, xv, lwp, x_new, sb_next.laux, lw);
    __syncthreads();  // BackSmem -> ScanSmem reuse
    segment_scan<2>(sm.s, lw, seg, sb_next.nsegp, sb_next.c1, sb_next.c2, sb_next.segm, sb_next.segs);
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
    segment_scan<1>(sm, lw, seg, sb.nsegp, sb.c1, nullptr, sb.segm, sb.segs);
}

// ------------------------------------------------------------------------------------------
// k_backtrace: reconstruct_trajectory (src/Filtering.py:40-55), one lane chases the ancestors
// ------------------------------------------------------------------------------------------
__global__ void k_backtrace(int N, int T, int nx, const double* __restrict__ x_trace, const int32_t* __restrict__ anc_trace,
                            const UpperHdr* __restrict__ hdr, double* __restrict__ traj) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int64_t b = hdr->final_idx;
    for (int k = 0; k < nx; ++k) traj[(size_t)(T - 1) * nx + k] = x_trace[((size_t)(T - 1) * N + b) * nx + k];
    for (int i = T - 2; i >= 0; --i) {
        b = anc_trace[(size_t)i * N + b];
        for (int k = 0; k < nx; ++k) traj[(size_t)i * nx + k] = x_trace[((size_t)i * N + b) * nx + k];
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
    double sc[PGAS_MAX_D], cc[PGAS_MAX_D], sd[PGAS_MAX_D], cd[PGAS_MAX_D];
    for (int d = 0; d < md.D; ++d) {
        const double r = PGAS_FMA(pick_input<NX>(md, d, xv, ut), md.alpha[d], md.beta[d]);
        dim_start(md, d, r, sc[d], cc[d], sd[d], cd[d]);
    }
    for (int m = 0; m < md.M; ++m) {
        double f = md.nrm;
        for (int d = 0; d < md.D; ++d) {
            const int q = (idx[m * md.D + d] - md.j0[d]) / md.jstep[d];
            double s = sc[d], c = cc[d];
            for (int i = 0; i < q; ++i) rotate(s, c, sd[d], cd[d]);
            f = f * s;
        }
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
