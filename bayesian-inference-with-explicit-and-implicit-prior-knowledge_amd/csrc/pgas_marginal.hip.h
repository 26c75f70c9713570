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

__global__ __launch_bounds__(256) void k_rng_normal(uint64_t seed, uint32_t stream, uint32_t t, int64_t p0, int64_t n, int ncol,
                                                     double* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    double z[8];
    pgas_rng_normals(seed, stream, t, (uint64_t)(p0 + p), ncol, z);
    for (int k = 0; k < ncol; ++k) out[p * ncol + k] = z[k];
}

__global__ __launch_bounds__(256) void k_rng_student_t(uint64_t seed, uint32_t stream, uint32_t t, int64_t p0, int64_t n,
                                                        const double* __restrict__ nu, double* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    out[p] = pgas_rng_student_t(seed, stream, t, (uint64_t)(p0 + p), nu[p]);
}

#define PG_MN_MAXM 64      // one matrix row per lane; particles per workgroup = blockDim.x / 64 (one wave each)

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Dynamic LDS: blockDim.x / 64 matrices of M x LD doubles, LD = M | 1 (odd stride: column reads hit distinct banks).
__global__ __launch_bounds__(256) void k_mniw_solve(int64_t n, int M, double scale, const int32_t* __restrict__ anc, const double* __restrict__ P0,
                                                                   const double* __restrict__ P1, const double* __restrict__ T0,
                                                                   const double* __restrict__ T1, const double* __restrict__ R0,
                                                                   const double* __restrict__ R1, const double* __restrict__ phi,
                                                                   double* __restrict__ m_out, double* __restrict__ c_out,
                                                                   double* __restrict__ q_out, double* __restrict__ logdet_out,
                                                                   int32_t* __restrict__ fail_out) {
    extern __shared__ double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (p >= n) return;  // whole wave leaves together; no workgroup barrier below
    const int LD = M | 1;
    double* __restrict__ A = smem + (size_t)wave * M * LD;
    // eta1 = P1 + scale * T1_p (+ R1): coalesced over the flattened matrix
    const int64_t src = anc ? (int64_t)anc[p] : p;   // statistics of the resampled ancestor (src/Algorithm1.py:358-361)
    const double* __restrict__ T1p = T1 + (size_t)src * M * M;
    for (int e = lane; e < M * M; e += 64) {
        const int r = e / M, cc = e - r * M;
        double v = P1[e] + scale * T1p[e];
        if (R1) v += R1[e];
        A[r * LD + cc] = v;
    }
    // right-hand sides: lane l holds b_l = phi_l and w_l = eta0_l
    double b = 0.0, w = 0.0;
    if (lane < M) {
        w = P0[lane] + scale * T0[(size_t)src * M + lane];
        if (R0) w += R0[lane];
        if (phi) b = phi[(size_t)p * M + lane];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // right-looking Cholesky, lane = row; the forward substitutions ride along column by column
    double logdet = 0.0;
    int bad = 0;
    for (int k = 0; k < M; ++k) {
        const double akk = A[k * LD + k];
        if (!(akk > 0.0)) bad = 1;
        const double d = sqrt(akk);
        logdet += pgas_log(akk);  // log det = sum log L_kk^2
        double lk = 0.0;          // L[lane][k]
        if (lane > k && lane < M) {
            lk = A[lane * LD + k] / d;
            A[lane * LD + k] = lk;
        }
        // forward substitution step k for both right-hand sides
        const double bk = __shfl(b, k) / d, wk = __shfl(w, k) / d;
        if (lane == k) {
            b = bk;
            w = wk;
        } else if (lane > k) {
            b -= lk * bk;
            w -= lk * wk;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // trailing update of row `lane`: A[lane][j] -= L[lane][k] L[j][k], k < j <= lane
        if (lane > k && lane < M) {
            for (int j = k + 1; j <= lane; ++j) A[lane * LD + j] -= lk * A[j * LD + k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const double mm = wave_sum_f64(lane < M ? w * b : 0.0);
    const double cc = wave_sum_f64(lane < M ? b * b : 0.0);
    const double qq = wave_sum_f64(lane < M ? w * w : 0.0);
    if (lane == 0) {
        if (m_out) m_out[p] = mm;
        if (c_out) c_out[p] = cc;
        if (q_out) q_out[p] = qq;
        if (logdet_out) logdet_out[p] = logdet;
        if (bad && fail_out) atomicAdd(fail_out, 1);
    }
}

// one workgroup per particle: T_out[p] = scale * T_in[a_p] + update
__global__ __launch_bounds__(256) void k_stats_gather_update(int64_t n, int M, double scale, const int32_t* __restrict__ anc,
                                                              const double* __restrict__ T0i, const double* __restrict__ T1i,
                                                              const double* __restrict__ T2i, const double* __restrict__ T3i,
                                                              const double* __restrict__ phi, const double* __restrict__ xi,
                                                              double* __restrict__ T0o, double* __restrict__ T1o, double* __restrict__ T2o,
                                                              double* __restrict__ T3o) {
    __shared__ double sphi[PG_MN_MAXM * 2];
    const int64_t p = blockIdx.x;
    const int64_t a = anc ? (int64_t)anc[p] : p;
    const int tid = threadIdx.x;
    const double x = xi[p];
    for (int e = tid; e < M; e += 256) {
        const double f = phi[(size_t)p * M + e];
        sphi[e] = f;
        T0o[(size_t)p * M + e] = scale * T0i[(size_t)a * M + e] + f * x;
    }
    __syncthreads();
    const double* __restrict__ src = T1i + (size_t)a * M * M;
    double* __restrict__ dst = T1o + (size_t)p * M * M;
    for (int e = tid; e < M * M; e += 256) {
        const int r = e / M, c = e - r * M;
        dst[e] = scale * src[e] + sphi[r] * sphi[c];
    }
    if (tid == 0) {
        T2o[p] = scale * T2i[a] + x * x;
        T3o[p] = scale * T3i[a] + 1.0;
    }
}
