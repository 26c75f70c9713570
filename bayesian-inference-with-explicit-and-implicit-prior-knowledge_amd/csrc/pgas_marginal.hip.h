// pgas_marginal.hip.h -- device kernels of the MARGINALISED family (reference src/Algorithm1.py, src/Algorithm3.py):
// every particle carries MNIW sufficient statistics (T0 (M), T1 (M,M), T2, T3) of a latent function with a scalar
// interface variable (n = 1 in every instantiation of the reference: SingleMassOscillator.py, Vehicle.py, EMPS.py, Toy_Example.py).
//
//   k_rng_normal / k_rng_student_t   per-particle N(0,1) and Student-t variates from the Philox streams of pgas_canon.h
//   k_mniw_solve                     one wave per particle: eta1 = P1 + s T1_{a_i} (+ R1), Cholesky in LDS, two triangular solves ->
//                                    m = eta0^T eta1^-1 phi   (BI:48-50 with Algorithm1.py:228-231, and BI:81),
//                                    c = phi^T eta1^-1 phi    (BI:84),  q = eta0^T eta1^-1 eta0 (BI:42, :115),  log det eta1 (BI:119)
//   k_stats_gather_update            T_out[i] = s T_in[a_i] + (phi_i xi_i, phi_i phi_i^T, xi_i^2, 1)   (Algorithm1.py:317-320,358-377)
//
// HBM-bound by construction: 8 (M^2 + M + 2) bytes per particle and pass (13.8 KB for M = 41).
#pragma once

#include "pgas_kernels.hip.h"

// t_dev (nullable): the time index of the Philox counters read from device memory instead of the argument -- what lets a whole
// filter step be captured in a graph and replayed for every t (pgas_m_set_time_source)
__global__ __launch_bounds__(256) void k_rng_normal(uint64_t seed, uint32_t stream, uint32_t t, const uint32_t* __restrict__ t_dev, int64_t p0,
                                                     int64_t n, int ncol, double* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    if (t_dev) t = *t_dev;
    double z[8];
    pgas_rng_normals(seed, stream, t, (uint64_t)(p0 + p), ncol, z);
    for (int k = 0; k < ncol; ++k) out[p * ncol + k] = z[k];
}

// nu_anc (nullable) / nu0 / nu_scale: degrees of freedom nu0 + nu_scale * nu[anc[p]] formed here (df = P3 + lambda T3[a], BI:45) instead of by two
// torch launches in front of the kernel; the defaults (NULL, 0, 1) read nu[p] as before
__global__ __launch_bounds__(256) void k_rng_student_t(uint64_t seed, uint32_t stream, uint32_t t, const uint32_t* __restrict__ t_dev, int64_t p0,
                                                        int64_t n, const double* __restrict__ nu, double* __restrict__ out,
                                                        const int32_t* __restrict__ nu_anc = nullptr, double nu0 = 0.0, double nu_scale = 1.0) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    if (t_dev) t = *t_dev;
    const double v = nu_anc ? nu0 + nu_scale * nu[nu_anc[p]] : nu[p];
    out[p] = pgas_rng_student_t(seed, stream, t, (uint64_t)(p0 + p), v);
}

// chi^2(nu_p) = 2 Gamma(nu_p / 2): the Bartlett diagonal of PGAS.sample_params (src/PGAS.py:323-327), drawn where it is consumed
// one uniform of (seed, stream, t) into device memory: the u of systematic resampling when t comes from t_dev
__global__ void k_rng_uniform_dev(uint64_t seed, uint32_t stream, uint32_t t, const uint32_t* __restrict__ t_dev, double* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = pgas_rng_uniform(seed, stream, t_dev ? *t_dev : t);
}

__global__ __launch_bounds__(256) void k_rng_chi2(uint64_t seed, uint32_t stream, uint32_t t, int64_t p0, int64_t n, const double* __restrict__ nu,
                                                   double* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    out[p] = 2.0 * pgas_rng_gamma(seed, stream, t, (uint64_t)(p0 + p), 0.5 * nu[p]);
}

#define PGAS_PI_D 3.141592653589793238462643383279502884
#define PG_MN_MAXM 62      // one matrix row per lane plus the two right-hand-side rows; particles per workgroup = blockDim.x / 64

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// k_mniw_solve: one wave per particle, lane = matrix row, the row held in registers (MT = compile-time row capacity).
//
// The two right-hand sides ride along as two EXTRA ROWS of the matrix: with
//        [ eta1   .   . ]                      [ L            ]
//    B = [ phi^T  0   . ]   elimination of the first M columns (right-looking Cholesky) leaves rows M, M+1 = [ v^T ; w^T ],
//        [ eta0^T 0   0 ]   v = L^-1 phi, w = L^-1 eta0, and the Schur complement of the corner = -[ v.v  . ; w.v  w.w ],
// i.e. c, m and q fall out of the same FMAs that factorise eta1 -- no separate substitution, no reductions.
// Only the lower triangle is touched: row r is read by lanes 0..r (contiguous in memory), parked in LDS in packed triangular
// order, and every lane picks up its own row.  Column k of the factor is broadcast lane by lane through LDS (wave-uniform
// ds_read of the packed factor), so the trailing update costs the VALU one FMA per matrix element.  1/sqrt(pivot) comes from
// v_rsq_f64 plus two Newton steps (relative error ~1e-16) instead of an IEEE sqrt and an IEEE division: that chain of ~60
// dependent instructions per column was the critical path of the first version.
__device__ __forceinline__ double rsqrt_newton(double a) {
    double y = __builtin_amdgcn_rsq(a);
    const double h = 0.5 * a;
    y = y * PGAS_FMA(-h * y, y, 1.5);
    y = y * PGAS_FMA(-h * y, y, 1.5);
    return y;
}
// Column k of the factorisation, k a template parameter: the recursion unrolls the outer loop at compile time whatever the
// optimiser's unroll budget is (with `#pragma unroll` the larger instantiations fell back to a runtime loop and their
// register-resident rows to scratch memory).
template <int MT, int K>
struct CholColumn {
    static __device__ __forceinline__ void run(double (&row)[MT], int M, double* __restrict__ A, int tl, int lane) {
        constexpr int k = K;
        if (k < M) {  // wave-uniform: columns M, M+1 (the right-hand sides' corner) and the padding are not pivots
            const double inv = rsqrt_newton(readlane_f64(row[k], k));
            const double lk = row[k] * inv;  // L[lane][k] for lane > k (lanes <= k carry values nobody reads)
            // column k of the factor goes to LDS in packed row order (L below the diagonal, 1/L_kk on it)
            if (lane < MT && lane >= k) A[tl + k] = lk;
            if (lane == k) A[tl + k] = inv;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // A[lane][j] -= L[lane][k] L[j][k] (only lanes >= j are read later)
            const double nlk = -lk;
#pragma unroll
            for (int j0 = k + 1; j0 < MT; j0 += 8) {   // groups of 8 bound the registers the broadcast values occupy
                double lj[8];
#pragma unroll
                for (int j = j0; j < j0 + 8 && j < MT; ++j) lj[j - j0] = A[j * (j + 1) / 2 + k];   // eight reads in flight
#pragma unroll
                for (int j = j0; j < j0 + 8 && j < MT; ++j) {
                    row[j] = PGAS_FMA(nlk, lj[j - j0], row[j]);
                    asm volatile("" : "+v"(row[j]));   // consume the broadcast here: otherwise every update of column j is deferred to iteration j
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        CholColumn<MT, K + 1>::run(row, M, A, tl, lane);
    }
};
template <int MT>
struct CholColumn<MT, MT> {
    static __device__ __forceinline__ void run(double (&)[MT], int, double* __restrict__, int, int) {}
};

// Lfac_out (n, (M+2)(M+3)/2): the packed factor INCLUDING the two extra rows (row M = v, row M+1 = w = L^-1 eta0), for k_mniw_trisolve.
template <int MT>
__global__ __launch_bounds__(256) void k_mniw_solve(int64_t n, int M, double scale, const int32_t* __restrict__ anc, const double* __restrict__ P0,
                                                     const double* __restrict__ P1, const double* __restrict__ T0,
                                                     const double* __restrict__ T1, const double* __restrict__ R0,
                                                     const double* __restrict__ R1, const double* __restrict__ phi,
                                                     double* __restrict__ m_out, double* __restrict__ c_out,
                                                     double* __restrict__ q_out, double* __restrict__ logdet_out,
                                                     double* __restrict__ Lfac_out, int32_t* __restrict__ fail_out) {
    extern __shared__ double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (p >= n) return;  // whole wave leaves together; no workgroup barrier below
    const int64_t src = anc ? (int64_t)anc[p] : p;   // statistics of the resampled ancestor (src/Algorithm1.py:358-361)
    double* __restrict__ A = smem + (size_t)wave * (MT * (MT + 1) / 2);
    const double* __restrict__ T1p = T1 + (size_t)src * M * M;
#pragma unroll 6
    for (int r = 0; r < M; ++r) {
        if (lane <= r) {
            double v = P1[r * M + lane] + scale * T1p[r * M + lane];
            if (R1) v += R1[r * M + lane];
            A[r * (r + 1) / 2 + lane] = v;
        }
    }
    const int tM = M * (M + 1) / 2, tM1 = (M + 1) * (M + 2) / 2;   // starts of rows M and M+1
    if (lane < M) {
        double w = P0[lane] + scale * T0[(size_t)src * M + lane];
        if (R0) w += R0[lane];
        A[tM + lane] = phi ? phi[(size_t)p * M + lane] : 0.0;
        A[tM1 + lane] = w;
    } else if (lane == M) {   // the corner: (M,M), (M+1,M), (M+1,M+1)
        A[tM + M] = 0.0;
        A[tM1 + M] = 0.0;
        A[tM1 + M + 1] = 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int tl = lane < MT ? lane * (lane + 1) / 2 : 0;
    double row[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const bool have = lane < M + 2 && j <= lane;
        const double v = A[have ? tl + j : 0];
        row[j] = have ? v : 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    CholColumn<MT, 0>::run(row, M, A, tl, lane);
    // rows M and M+1 now hold [v^T, -v.v] and [w^T, -w.v, -w.w]; put their tails (columns >= M) next to the factor in LDS
#pragma unroll
    for (int j = 0; j < MT; ++j)
        if ((lane == M || lane == M + 1) && j >= M && j <= lane) A[tl + j] = row[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (Lfac_out) {  // hand the factor (with w in row M+1) to k_mniw_trisolve: the children of this particle reuse it
        const int tri_out = (M + 2) * (M + 3) / 2;
        double* __restrict__ dst = Lfac_out + (size_t)p * tri_out;
        for (int e = lane; e < tri_out; e += 64) dst[e] = A[e];
    }
    const double dinv = lane < M ? A[tl + lane] : 1.0;                    // 1 / L_kk
    const double ld = -2.0 * wave_sum_f64(lane < M ? pgas_log(dinv) : 0.0);  // log det eta1 = sum log L_kk^2
    if (lane == 0) {
        if (c_out) c_out[p] = -A[tM + M];
        if (m_out) m_out[p] = -A[tM1 + M];
        if (q_out) q_out[p] = -A[tM1 + M + 1];
        if (logdet_out) logdet_out[p] = ld;
        if (!(ld - ld == 0.0) && fail_out) atomicAdd(fail_out, 1);   // a non-positive pivot leaves NaN / inf in the factor
    }
}

// ------------------------------------------------------------------------------------------
// k_mniw_solve_mfma: the same factorisation, blocked by panels of four columns, with the trailing update
//     A22 -= L21 L21^T      (a rank-4 update of every remaining 16 x 16 tile)
// on v_mfma_f64_16x16x4_f64.  The (M+2)-row augmented matrix (see k_mniw_solve) lives in the MFMA accumulator layout,
// lower tiles only: tile (tr, tc), register i of lane l = element (16 tr + (l>>4) + 4 i, 16 tc + (l&15)).  Per panel:
//   (a) the four panel columns go from the accumulators to LDS,  (b) come back with lane = row,
//   (c) are factorised in registers (pivot broadcast by v_readlane, 1/sqrt by v_rsq_f64 + Newton),
//   (d) the scaled columns go to LDS as the MFMA operand image P[k][row] (rows above the trailing block and non-pivot
//       columns zeroed) and, packed, as the output factor,  (e) each lane picks up its A/B operand element
//       P[l>>4][16 r + (l&15)] per tile row,  (f) one MFMA per remaining tile.
// c, q and m (= the Schur complement of the two right-hand-side rows) are accumulated column by column on lanes M, M+1.
// ------------------------------------------------------------------------------------------
typedef double pg_mf4 __attribute__((ext_vector_type(4)));
__host__ __device__ constexpr int pg_tile_id(int tr, int tc) { return tr * (tr + 1) / 2 + tc; }

template <int NT, int P>
struct CholPanel {
    static __device__ __forceinline__ void run(pg_mf4 (&acc)[NT * (NT + 1) / 2], int M, double* __restrict__ colbuf, double* __restrict__ Lpack,
                                               int lane, double& dinv, double& nrm, double& crs) {
        constexpr int R = 16 * NT, k0 = 4 * P, tc = k0 / 16, c0 = k0 % 16, tcur = (k0 + 4) / 16;
        if (k0 < M) {  // wave-uniform: this panel holds at least one pivot column
            const int lc = (lane & 15) - c0;  // which panel column this lane's accumulator column is, if 0..3
            if (lc >= 0 && lc < 4) {
#pragma unroll
                for (int tr = tc; tr < NT; ++tr)
#pragma unroll
                    for (int i = 0; i < 4; ++i) colbuf[lc * R + 16 * tr + (lane >> 4) + 4 * i] = acc[pg_tile_id(tr, tc)][i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            double a[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) a[c] = colbuf[c * R + (lane < R ? lane : 0)];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                constexpr int kbase = k0;
                const int k = kbase + c;
                if (k < M) {  // wave-uniform: columns M, M+1 (right-hand sides) and the padding are not pivots
                    const double inv = rsqrt_newton(readlane_f64(a[c], k));
                    a[c] = a[c] * inv;  // L[lane][k] for lane > k
                    if (lane == k) dinv = inv;
#pragma unroll
                    for (int c2 = c + 1; c2 < 4; ++c2) a[c2] = PGAS_FMA(-a[c], readlane_f64(a[c], kbase + c2), a[c2]);
                    nrm = PGAS_FMA(a[c], a[c], nrm);                     // lane M: v.v, lane M+1: w.w
                    crs = PGAS_FMA(a[c], readlane_f64(a[c], M), crs);    // lane M+1: w.v
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = k0 + c;
                const bool piv = k < M;
                if (lane < R) colbuf[c * R + lane] = (piv && lane >= k0 + 4) ? a[c] : 0.0;
                if (piv && lane >= k && lane < M + 2) Lpack[lane * (lane + 1) / 2 + k] = lane == k ? dinv : a[c];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if constexpr (tcur < NT) {
                double op[NT];
#pragma unroll
                for (int r = tcur; r < NT; ++r) op[r] = colbuf[(lane >> 4) * R + 16 * r + (lane & 15)];
#pragma unroll
                for (int tr = tcur; tr < NT; ++tr)
#pragma unroll
                    for (int t2 = tcur; t2 <= tr; ++t2)
                        acc[pg_tile_id(tr, t2)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-op[tr], op[t2], acc[pg_tile_id(tr, t2)], 0, 0, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        CholPanel<NT, P + 1>::run(acc, M, colbuf, Lpack, lane, dinv, nrm, crs);
    }
};
template <int NT>
struct CholPanel<NT, 4 * NT> {
    static __device__ __forceinline__ void run(pg_mf4 (&)[NT * (NT + 1) / 2], int, double* __restrict__, double* __restrict__, int, double&, double&,
                                               double&) {}
};

template <int NT>
__global__ __launch_bounds__(256) void k_mniw_solve_mfma(int64_t n, int M, double scale, const int32_t* __restrict__ anc, const double* __restrict__ P0,
                                                          const double* __restrict__ P1, const double* __restrict__ T0,
                                                          const double* __restrict__ T1, const double* __restrict__ R0,
                                                          const double* __restrict__ R1, const double* __restrict__ phi,
                                                          double* __restrict__ m_out, double* __restrict__ c_out,
                                                          double* __restrict__ q_out, double* __restrict__ logdet_out,
                                                          double* __restrict__ Lfac_out, int32_t* __restrict__ fail_out) {
    extern __shared__ double smem[];
    constexpr int R = 16 * NT, PERW = 4 * R + R * (R + 1) / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (p >= n) return;  // whole wave leaves together; no workgroup barrier below
    const int64_t src = anc ? (int64_t)anc[p] : p;
    double* __restrict__ colbuf = smem + (size_t)wave * PERW;
    double* __restrict__ Lpack = colbuf + 4 * R;
    const double* __restrict__ T1p = T1 + (size_t)src * M * M;
    pg_mf4 acc[NT * (NT + 1) / 2];
    // Branch-free loader straight into the accumulator layout: every element is read from a clamped (always valid) address and
    // then selected, so all loads of a matrix are in flight together (staging rows through LDS first, as k_mniw_solve does,
    // issues fewer instructions but measured 30 % slower here: its row loop serialises the memory latency).
    // Column values of the two right-hand-side rows first (one per tile column and lane) ...
    const int lr = lane >> 4, lcol = lane & 15;
    double phic[NT], etac[NT];
#pragma unroll
    for (int t2 = 0; t2 < NT; ++t2) {
        const int col = 16 * t2 + lcol, cc = col < M ? col : 0;
        double e = P0[cc] + scale * T0[(size_t)src * M + cc];
        if (R0) e += R0[cc];
        phic[t2] = phi ? phi[(size_t)p * M + cc] : 0.0;
        etac[t2] = e;
    }
    // ... then the lower tiles of eta1 = P1 + scale T1 (+ R1)
#pragma unroll
    for (int tr = 0; tr < NT; ++tr)
#pragma unroll
        for (int t2 = 0; t2 <= tr; ++t2)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * tr + lr + 4 * i, col = 16 * t2 + lcol;
                const bool inmat = row < M && col <= row;
                const int e = inmat ? row * M + col : 0;
                double v = P1[e] + scale * T1p[e];
                if (R1) v += R1[e];
                const double rhs = row == M ? phic[t2] : etac[t2];
                acc[pg_tile_id(tr, t2)][i] = inmat ? v : ((col < M && (row == M || row == M + 1)) ? rhs : 0.0);
            }
    double dinv = 1.0, nrm = 0.0, crs = 0.0;
    CholPanel<NT, 0>::run(acc, M, colbuf, Lpack, lane, dinv, nrm, crs);
    if (lane == M) {  // the corner of the packed output (not read by k_mniw_trisolve)
        const int tM = M * (M + 1) / 2, tM1 = (M + 1) * (M + 2) / 2;
        Lpack[tM + M] = 0.0;
        Lpack[tM1 + M] = 0.0;
        Lpack[tM1 + M + 1] = 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (Lfac_out) {
        const int tri_out = (M + 2) * (M + 3) / 2;
        double* __restrict__ dst = Lfac_out + (size_t)p * tri_out;
        for (int e = lane; e < tri_out; e += 64) dst[e] = Lpack[e];
    }
    const double ld = -2.0 * wave_sum_f64(lane < M ? pgas_log(dinv) : 0.0);
    const double cc = readlane_f64(nrm, M), qq = readlane_f64(nrm, M + 1), mm = readlane_f64(crs, M + 1);
    if (lane == 0) {
        if (c_out) c_out[p] = cc;
        if (m_out) m_out[p] = mm;
        if (q_out) q_out[p] = qq;
        if (logdet_out) logdet_out[p] = ld;
        if (!(ld - ld == 0.0) && fail_out) atomicAdd(fail_out, 1);
    }
}

// Solve with a stored factor: v = L^-1 phi for the factor of particle anc[p] (written by k_mniw_solve), then
// m = w . v and c = v . v with w = L^-1 eta0 of the same ancestor (row M+1 of the stored triangle).
// Streaming: 8 ((M+2)(M+3)/2 + M) bytes per particle.
__global__ __launch_bounds__(256) void k_mniw_trisolve(int64_t n, int M, const int32_t* __restrict__ anc, const double* __restrict__ Lfac,
                                                        const double* __restrict__ phi, double* __restrict__ m_out, double* __restrict__ c_out) {
    extern __shared__ double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (p >= n) return;
    const int64_t src = anc ? (int64_t)anc[p] : p;
    const int tri_n = (M + 2) * (M + 3) / 2;
    double* __restrict__ A = smem + (size_t)wave * tri_n;
    const double* __restrict__ Ls = Lfac + (size_t)src * tri_n;
    for (int e = lane; e < tri_n; e += 64) A[e] = Ls[e];
    double b = lane < M ? phi[(size_t)p * M + lane] : 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int tl = lane < M ? lane * (lane + 1) / 2 : 0;
    const double w = lane < M ? A[(M + 1) * (M + 2) / 2 + lane] : 0.0;
    const double dinv = lane < M ? A[tl + lane] : 1.0;
    for (int k = 0; k < M; ++k) {
        const double bk = readlane_f64(b, k) * readlane_f64(dinv, k);
        const double lk = (lane > k && lane < M) ? A[tl + k] : 0.0;
        if (lane == k) b = bk;
        else if (lane > k) b = PGAS_FMA(-lk, bk, b);
    }
    const double mm = wave_sum_f64(lane < M ? w * b : 0.0);
    const double cc = wave_sum_f64(lane < M ? b * b : 0.0);
    if (lane == 0) {
        if (m_out) m_out[p] = mm;
        if (c_out) c_out[p] = cc;
    }
}

// ------------------------------------------------------------------------------------------
// k_hilbert_batch: basis_fcn(state, input) for every particle (src/BasisFunctions.py:77-80 under the vmap of src/Algorithm1.py:220-225,
// :243-248): phi[p][m] = prod_d amp_d sin(pi j[m][d] (v_d / div_d - center_d + L_d) / size_d), v = concat(state[p], input)[sel] -- one
// launch instead of the ten elementwise launches of the torch expression.  Thread = (particle, function).
// k_mniw_draw: the scalar matrix-t draw of src/Algorithm1.py:251-262 / BI:64-108 behind the stored-factor solve:
//   xi = m + sqrt((P2 + s T2[a] - q[a]) / (P3 + s T3[a])) t sqrt(c + 1).
// ------------------------------------------------------------------------------------------
// k_lbm_diff: g[p] = log base measure of (prior + statistics of particle p) minus that of (prior + statistics + the reference trajectory's
// remaining statistics) -- the two vmap(BI.prior_mniw_log_base_measure) terms of the conditional filter's ancestor weights
// (src/Algorithm3.py:93-108; BI:111-124 with n = 1: multigammaln(a, 1) = lgamma(a)) -- from the solves' q and log det:
//   lbm(nu, Psi, logdet) = -M/2 log(2 pi) + logdet / 2 - nu / 2 log 2 - lgamma(nu / 2) + nu / 2 log Psi.
__global__ __launch_bounds__(256) void k_lbm_diff(int64_t n, int M, const double* __restrict__ T2, const double* __restrict__ T3, const double* __restrict__ q1,
                                                   const double* __restrict__ ld1, const double* __restrict__ q2, const double* __restrict__ ld2, double P2,
                                                   double P3, const double* __restrict__ r2, const double* __restrict__ r3, double* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const double c0 = -0.5 * (double)M * 1.8378770664093454835606594728112353, ln2 = 0.6931471805599453094172321214581766;
    const double nu1 = P3 + T3[p], psi1 = P2 + T2[p] - q1[p];
    const double nu2 = P3 + T3[p] + r3[0], psi2 = P2 + T2[p] + r2[0] - q2[p];
    const double a = c0 + 0.5 * ld1[p] - 0.5 * nu1 * ln2 - lgamma(nu1 / 2) + log(psi1) * nu1 / 2;
    const double b = c0 + 0.5 * ld2[p] - 0.5 * nu2 * ln2 - lgamma(nu2 / 2) + log(psi2) * nu2 / 2;
    out[p] = a - b;
}
#define PG_HB_MAXD 4
struct HilbertArgs {
    int32_t D, nx, nu, sel[PG_HB_MAXD];
    double div[PG_HB_MAXD], center[PG_HB_MAXD], L[PG_HB_MAXD], size[PG_HB_MAXD], amp[PG_HB_MAXD];
};
__global__ __launch_bounds__(256) void k_hilbert_batch(int64_t n, int M, HilbertArgs h, const double* __restrict__ state, const double* __restrict__ input,
                                                        const int32_t* __restrict__ idx, double* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * M) return;
    const int64_t p = e / M;
    const int m = (int)(e - p * M);
    double prod = 1.0;
    for (int d = 0; d < h.D; ++d) {
        const int sd = h.sel[d];
        const double v = sd < h.nx ? state[p * h.nx + sd] : input[sd - h.nx];
        const double ang = PGAS_PI_D * (double)idx[m * h.D + d] * ((v / h.div[d] - h.center[d] + h.L[d]) / h.size[d]);
        const double f = h.amp[d] * sin(ang);
        prod = d == 0 ? f : prod * f;
    }
    out[e] = prod;
}
__global__ __launch_bounds__(256) void k_mniw_draw(int64_t n, double scale, const int32_t* __restrict__ anc, const double* __restrict__ m,
                                                    const double* __restrict__ c, const double* __restrict__ q, const double* __restrict__ T2,
                                                    const double* __restrict__ T3, double P2, double P3, const double* __restrict__ t,
                                                    double* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int64_t a = anc ? (int64_t)anc[p] : p;
    const double df = P3 + scale * T3[a];
    const double row = (P2 + scale * T2[a] - q[a]) / df;
    out[p] = m[p] + sqrt(row) * t[p] * sqrt(c[p] + 1.0);
}

// ------------------------------------------------------------------------------------------
// k_expr: a state-space model's transition / output function as DATA.  pgas_amd/exprs.py traces the model callable once into a register
// program over per-particle scalars (opcode, destination, two sources; inputs and constants preloaded); this kernel runs it for every
// particle in one launch -- an RK4 transition is ~20 instructions here and ~45 elementwise launches in torch (src/StateSpaceModel.py:32-87
// evaluates the callables under jax.vmap).  Three tails share the program's result v (n_out values per particle):
//   mode 0  out[p][k] = v_k                                                  transition_mdl / output_mdl          (:32-54)
//   mode 1  out[p][k] = v_k + sum_l z[p][l] Qc[k][l]                         draw_state, z standard normals       (:56-74)
//   mode 2  out[p]    = cR - 1/2 |LRinv (y - v)|^2                           log_likelihood                        (:76-87)
// `anc` (nullable) gathers the state and the interface variables of particle anc[p] (the resampled parents, Algorithm1.py:350-353).
// ------------------------------------------------------------------------------------------
#define PG_EX_MAXREG 96
#define PG_EX_MAXIV 4
#define PG_EX_MAXOUT 8
struct ExprArgs {
    const double* state; const int32_t* anc; const double* u; const double* iv[PG_EX_MAXIV];
    int32_t ivw[PG_EX_MAXIV], n_iv, nx, nu;
    const int32_t* code; const double* consts; int32_t ninstr, nconst, n_in;
    int32_t out_reg[PG_EX_MAXOUT], nout, mode;
    const double* aux; const double* mat; double cR; double* out;
};
__global__ __launch_bounds__(256) void k_expr(int64_t n, ExprArgs a) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int64_t src = a.anc ? (int64_t)a.anc[p] : p;
    double r[PG_EX_MAXREG];
    int k = 0;
    for (int j = 0; j < a.nx; ++j) r[k++] = a.state[src * a.nx + j];
    for (int j = 0; j < a.nu; ++j) r[k++] = a.u[j];
    for (int i = 0; i < a.n_iv; ++i)
        for (int j = 0; j < a.ivw[i]; ++j) r[k++] = a.iv[i][src * a.ivw[i] + j];
    for (int j = 0; j < a.nconst; ++j) r[a.n_in + j] = a.consts[j];
    for (int i = 0; i < a.ninstr; ++i) {   // uniform control flow: every lane runs the same program
        const int op = a.code[4 * i], d = a.code[4 * i + 1];
        const double x = r[a.code[4 * i + 2]], y = r[a.code[4 * i + 3]];
        double v;
        switch (op) {
            case 1: v = x + y; break;
            case 2: v = x - y; break;
            case 3: v = x * y; break;
            case 4: v = x / y; break;
            case 5: v = -x; break;
            case 6: v = cos(x); break;
            case 7: v = sin(x); break;
            case 8: v = tan(x); break;
            case 9: v = tanh(x); break;
            case 10: v = atan(x); break;
            case 11: v = sqrt(x); break;
            case 12: v = exp(x); break;
            case 13: v = (double)((x > 0.0) - (x < 0.0)); break;
            default: v = x; break;
        }
        r[d] = v;
    }
    double val[PG_EX_MAXOUT];
    for (int j = 0; j < a.nout; ++j) val[j] = r[a.out_reg[j]];
    if (a.mode == 0) {
        for (int j = 0; j < a.nout; ++j) a.out[p * a.nout + j] = val[j];
    } else if (a.mode == 1) {
        for (int j = 0; j < a.nout; ++j) {
            double acc = val[j];
            for (int l = 0; l < a.nout; ++l) acc += a.aux[p * a.nout + l] * a.mat[j * a.nout + l];
            a.out[p * a.nout + j] = acc;
        }
    } else {
        double q = 0.0;
        for (int j = 0; j < a.nout; ++j) {
            double e = 0.0;
            for (int l = 0; l < a.nout; ++l) e += (a.aux[l] - val[l]) * a.mat[j * a.nout + l];
            q += e * e;
        }
        a.out[p] = a.cR - 0.5 * q;
    }
}

// ------------------------------------------------------------------------------------------
// Bases of 63 ... 126 functions (the reference's formulas, BI:48-50,64-124, are general in M): M + 2 rows no longer fit one per lane.
// k_mniw_solve_wide / k_mniw_trisolve_wide: one wave per particle, TWO rows per lane (rows lane and lane + 64), the packed triangle
// in LDS (66 KB at M = 126) and updated in place, column by column -- the SAME operations in the same order as k_mniw_solve (pivot
// 1 / sqrt by v_rsq_f64 + two Newton steps, A[r][j] = fma(-L[r][k], L[j][k], A[r][j]) for k ascending, right-hand sides as two extra
// rows), so that for M <= 62 the results are bit-identical to it (tests/test_gpu_marginal.py).  A generality path, not a fast one:
// every FMA costs two LDS reads and a write.
// The same kernels carry interface variables with nv > 1 components (eta0 is (M, nv): BI:18-50 are general in n): the nv columns of
// eta0 are nv extra rows of the augmented matrix, M + 1 + nv <= 128 rows in all; the Schur complement of the corner then holds
// c = v.v, m_j = w_j.v and Q_jk = w_j.w_k (row_scale = eta2 - Q, BI:42).  nv = 1 is the scalar case above, bit for bit.
// ------------------------------------------------------------------------------------------
#define PG_MN_MAXM_WIDE 126
#define PG_MN_MAXROWS_WIDE 128
__global__ __launch_bounds__(64) void k_mniw_solve_wide(int64_t n, int M, int nv, double scale, const int32_t* __restrict__ anc, const double* __restrict__ P0,
                                                         const double* __restrict__ P1, const double* __restrict__ T0,
                                                         const double* __restrict__ T1, const double* __restrict__ R0,
                                                         const double* __restrict__ R1, const double* __restrict__ phi,
                                                         double* __restrict__ m_out, double* __restrict__ c_out,
                                                         double* __restrict__ q_out, double* __restrict__ logdet_out,
                                                         double* __restrict__ Lfac_out, int32_t* __restrict__ fail_out) {
    extern __shared__ double smem[];
    const int lane = threadIdx.x;
    const int64_t p = blockIdx.x;
    if (p >= n) return;
    const int64_t src = anc ? (int64_t)anc[p] : p;
    double* __restrict__ A = smem;
    const double* __restrict__ T1p = T1 + (size_t)src * M * M;
    const int R = M + 1 + nv;
    for (int r = 0; r < M; ++r)
        for (int j = lane; j <= r; j += 64) {
            double v = P1[r * M + j] + scale * T1p[r * M + j];
            if (R1) v += R1[r * M + j];
            A[r * (r + 1) / 2 + j] = v;
        }
    const int tM = M * (M + 1) / 2;   // start of row M (phi); row M + 1 + v holds column v of eta0
    for (int j = lane; j < M; j += 64) {
        A[tM + j] = phi ? phi[(size_t)p * M + j] : 0.0;
        for (int v = 0; v < nv; ++v) {
            double w = P0[j * nv + v] + scale * T0[((size_t)src * M + j) * nv + v];
            if (R0) w += R0[j * nv + v];
            A[(M + 1 + v) * (M + 2 + v) / 2 + j] = w;
        }
    }
    if (lane == 0)   // the corner: rows M ... R-1, columns M ... row
        for (int r = M; r < R; ++r)
            for (int cc = M; cc <= r; ++cc) A[r * (r + 1) / 2 + cc] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int r0 = lane, r1 = lane + 64;
    const int t0 = r0 * (r0 + 1) / 2, t1 = r1 * (r1 + 1) / 2;
    for (int k = 0; k < M; ++k) {
        const int tk = k * (k + 1) / 2;
        const double inv = rsqrt_newton(A[tk + k]);
        double l0 = 0.0, l1 = 0.0;
        if (r0 > k && r0 < R) l0 = A[t0 + k] * inv;
        if (r1 > k && r1 < R) l1 = A[t1 + k] * inv;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (r0 > k && r0 < R) A[t0 + k] = l0;
        if (r1 > k && r1 < R) A[t1 + k] = l1;
        if (r0 == k) A[t0 + k] = inv;        // 1 / L_kk on the diagonal, the packed format of k_mniw_solve
        if (r1 == k) A[t1 + k] = inv;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (r0 > k && r0 < R) {
            const double nl = -l0;
            for (int j = k + 1; j <= r0; ++j) A[t0 + j] = PGAS_FMA(nl, A[j * (j + 1) / 2 + k], A[t0 + j]);
        }
        if (r1 > k && r1 < R) {
            const double nl = -l1;
            for (int j = k + 1; j <= r1; ++j) A[t1 + j] = PGAS_FMA(nl, A[j * (j + 1) / 2 + k], A[t1 + j]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (Lfac_out) {
        const int tri_out = R * (R + 1) / 2;
        double* __restrict__ dst = Lfac_out + (size_t)p * tri_out;
        for (int e = lane; e < tri_out; e += 64) dst[e] = A[e];
    }
    const double d0 = r0 < M ? A[t0 + r0] : 1.0, d1 = r1 < M ? A[t1 + r1] : 1.0;
    const double ld = -2.0 * wave_sum_f64((r0 < M ? pgas_log(d0) : 0.0) + (r1 < M ? pgas_log(d1) : 0.0));
    if (lane == 0) {
        if (c_out) c_out[p] = -A[tM + M];
        for (int v = 0; v < nv; ++v) {
            const int tv = (M + 1 + v) * (M + 2 + v) / 2;
            if (m_out) m_out[(size_t)p * nv + v] = -A[tv + M];
            if (q_out)
                for (int u = 0; u <= v; ++u) {
                    const double q = -A[tv + M + 1 + u];
                    q_out[((size_t)p * nv + v) * nv + u] = q;
                    q_out[((size_t)p * nv + u) * nv + v] = q;
                }
        }
        if (logdet_out) logdet_out[p] = ld;
        if (!(ld - ld == 0.0) && fail_out) atomicAdd(fail_out, 1);
    }
}

__global__ __launch_bounds__(64) void k_mniw_trisolve_wide(int64_t n, int M, int nv, const int32_t* __restrict__ anc, const double* __restrict__ Lfac,
                                                            const double* __restrict__ phi, double* __restrict__ m_out, double* __restrict__ c_out) {
    extern __shared__ double smem[];
    const int lane = threadIdx.x;
    const int64_t p = blockIdx.x;
    if (p >= n) return;
    const int64_t src = anc ? (int64_t)anc[p] : p;
    const int tri_n = (M + 1 + nv) * (M + 2 + nv) / 2;
    double* __restrict__ A = smem;
    const double* __restrict__ Ls = Lfac + (size_t)src * tri_n;
    for (int e = lane; e < tri_n; e += 64) A[e] = Ls[e];
    const int r0 = lane, r1 = lane + 64;
    double b0 = r0 < M ? phi[(size_t)p * M + r0] : 0.0, b1 = r1 < M ? phi[(size_t)p * M + r1] : 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int t0 = r0 < M ? r0 * (r0 + 1) / 2 : 0, t1 = r1 < M ? r1 * (r1 + 1) / 2 : 0;
    const double di0 = r0 < M ? A[t0 + r0] : 1.0, di1 = r1 < M ? A[t1 + r1] : 1.0;
    for (int k = 0; k < M; ++k) {
        const int kl = k & 63;
        const double bk = k < 64 ? readlane_f64(b0, kl) * readlane_f64(di0, kl) : readlane_f64(b1, kl) * readlane_f64(di1, kl);   // uniform
        const double lk0 = (r0 > k && r0 < M) ? A[t0 + k] : 0.0, lk1 = (r1 > k && r1 < M) ? A[t1 + k] : 0.0;
        if (r0 == k) b0 = bk;
        else if (r0 > k) b0 = PGAS_FMA(-lk0, bk, b0);
        if (r1 == k) b1 = bk;
        else if (r1 > k) b1 = PGAS_FMA(-lk1, bk, b1);
    }
    for (int v = 0; v < nv; ++v) {   // m_v = w_v . v with w_v = row M + 1 + v of the stored triangle
        const int tw = (M + 1 + v) * (M + 2 + v) / 2;
        const double w0 = r0 < M ? A[tw + r0] : 0.0, w1 = r1 < M ? A[tw + r1] : 0.0;
        const double mm = wave_sum_f64((r0 < M ? w0 * b0 : 0.0) + (r1 < M ? w1 * b1 : 0.0));
        if (lane == 0 && m_out) m_out[(size_t)p * nv + v] = mm;
    }
    const double cc = wave_sum_f64((r0 < M ? b0 * b0 : 0.0) + (r1 < M ? b1 * b1 : 0.0));
    if (lane == 0 && c_out) c_out[p] = cc;
}

// one workgroup per particle: T_out[p] = scale * T_in[a_p] + update
// nv = components of the interface variable: T0 (n, M, nv), T2 (n, nv, nv), xi (n, nv); nv = 1 is the scalar layout
__global__ __launch_bounds__(256) void k_stats_gather_update(int64_t n, int M, int nv, double scale, const int32_t* __restrict__ anc,
                                                              const double* __restrict__ T0i, const double* __restrict__ T1i,
                                                              const double* __restrict__ T2i, const double* __restrict__ T3i,
                                                              const double* __restrict__ phi, const double* __restrict__ xi,
                                                              double* __restrict__ T0o, double* __restrict__ T1o, double* __restrict__ T2o,
                                                              double* __restrict__ T3o) {
    __shared__ double sphi[128];
    const int64_t p = blockIdx.x;
    const int64_t a = anc ? (int64_t)anc[p] : p;
    const int tid = threadIdx.x;
    for (int e = tid; e < M; e += 256) {
        const double f = phi[(size_t)p * M + e];
        sphi[e] = f;
        for (int v = 0; v < nv; ++v) T0o[((size_t)p * M + e) * nv + v] = scale * T0i[((size_t)a * M + e) * nv + v] + f * xi[(size_t)p * nv + v];
    }
    __syncthreads();
    const double* __restrict__ src = T1i + (size_t)a * M * M;
    double* __restrict__ dst = T1o + (size_t)p * M * M;
    for (int e = tid; e < M * M; e += 256) {
        const int r = e / M, c = e - r * M;
        dst[e] = scale * src[e] + sphi[r] * sphi[c];
    }
    if (tid < nv * nv) {
        const int v = tid / nv, u = tid - v * nv;
        T2o[(size_t)p * nv * nv + tid] = scale * T2i[(size_t)a * nv * nv + tid] + xi[(size_t)p * nv + v] * xi[(size_t)p * nv + u];
    }
    if (tid == 0) T3o[p] = scale * T3i[a] + 1.0;
}

// ------------------------------------------------------------------------------------------
// Weighted reduction over the particle axis (src/Algorithm1.py:166-170, :445-457):
//   S = sum_p w_p (T0_p, T1_p, T2_p, T3_p)
// a (1 x n) by (n x (M^2 + M + 2)) contraction that is pure streaming: every statistic is read exactly once.
// Thread = one column of the concatenated record [T1 | T0 | T2 | T3], workgroup = 256 columns x one chunk of particles;
// partial sums per chunk, then a second pass adds the chunks in index order (deterministic, no atomics).
// ------------------------------------------------------------------------------------------
#define PG_WS_CHUNK 512
__device__ __forceinline__ double ws_column(int col, int M, int nv, int64_t p, const double* __restrict__ T0, const double* __restrict__ T1,
                                            const double* __restrict__ T2, const double* __restrict__ T3) {
    const int mm = M * M, m0 = M * nv, m2 = nv * nv;   // record = [T1 | T0 | T2 | T3]
    if (col < mm) return T1[(size_t)p * mm + col];
    if (col < mm + m0) return T0[(size_t)p * m0 + (col - mm)];
    return col < mm + m0 + m2 ? T2[(size_t)p * m2 + (col - mm - m0)] : T3[p];
}
__global__ __launch_bounds__(256) void k_weighted_stats_partial(int64_t n, int M, int nv, const double* __restrict__ w, const double* __restrict__ T0,
                                                                 const double* __restrict__ T1, const double* __restrict__ T2,
                                                                 const double* __restrict__ T3, double* __restrict__ partial) {
    const int ncol = M * M + M * nv + nv * nv + 1;
    const int col = blockIdx.x * 256 + threadIdx.x;
    const int64_t p0 = (int64_t)blockIdx.y * PG_WS_CHUNK;
    const int64_t p1 = p0 + PG_WS_CHUNK < n ? p0 + PG_WS_CHUNK : n;
    if (col >= ncol) return;
    double acc0 = 0.0, acc1 = 0.0;  // two chains: the loads of consecutive particles overlap
    int64_t p = p0;
    for (; p + 1 < p1; p += 2) {
        acc0 = PGAS_FMA(w[p], ws_column(col, M, nv, p, T0, T1, T2, T3), acc0);
        acc1 = PGAS_FMA(w[p + 1], ws_column(col, M, nv, p + 1, T0, T1, T2, T3), acc1);
    }
    if (p < p1) acc0 = PGAS_FMA(w[p], ws_column(col, M, nv, p, T0, T1, T2, T3), acc0);
    partial[(size_t)blockIdx.y * ncol + col] = acc0 + acc1;
}
__global__ __launch_bounds__(256) void k_weighted_stats_final(int nchunk, int M, int nv, const double* __restrict__ partial, double* __restrict__ S0,
                                                               double* __restrict__ S1, double* __restrict__ S2, double* __restrict__ S3) {
    const int mm = M * M, m0 = M * nv, m2 = nv * nv, ncol = mm + m0 + m2 + 1;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= ncol) return;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;   // four chains in a fixed interleaving: deterministic, and the loads overlap
    int c = 0;
    for (; c + 3 < nchunk; c += 4) {
        a0 += partial[(size_t)c * ncol + col];
        a1 += partial[(size_t)(c + 1) * ncol + col];
        a2 += partial[(size_t)(c + 2) * ncol + col];
        a3 += partial[(size_t)(c + 3) * ncol + col];
    }
    for (; c < nchunk; ++c) a0 += partial[(size_t)c * ncol + col];
    const double acc = (a0 + a1) + (a2 + a3);
    if (col < mm) S1[col] = acc;
    else if (col < mm + m0) S0[col - mm] = acc;
    else if (col < mm + m0 + m2) S2[col - mm - m0] = acc;
    else S3[0] = acc;
}
