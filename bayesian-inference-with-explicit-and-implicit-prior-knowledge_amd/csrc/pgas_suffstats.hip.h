// pgas_suffstats.hip.h -- MNIW sufficient statistics of one trajectory
// (PGAS.sample_params, reference src/PGAS.py:294-303; prior_mniw_calcStatistics, BI:53-61).
//
// The reference materialises T-1 outer products (T-1,M,M) and sums them; here
//   k_traj_basis : Phi (R,M) = basis(traj[r], inputs[r]),  R = T-1            (quirk Q3 pairing)
//   k_syrk_mfma  : T1 = Phi^T Phi on v_mfma_f64_16x16x4_f64 (the one real dense contraction on the path)
//   k_t0t2       : T0 = Phi^T X+, T2 = X+^T X+  (skinny, VALU)
// Floating-point summation order differs from the reference's sum over t; parity is asserted
// to 1e-12 relative (tests/test_gpu_suffstats.py), not bit for bit.
#pragma once

#include "pgas_kernels.hip.h"

template <int NX>
__global__ __launch_bounds__(64) void k_traj_basis(DevModel md, const int32_t* __restrict__ idx, const double* __restrict__ traj, int R,
                                                    int Mp, double* __restrict__ phi) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= R) return;
    const double* __restrict__ ut = md.u + (size_t)r * md.nu;
    double xv[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) xv[k] = traj[(size_t)r * NX + k];
    double sv[PGAS_MAX_D][PGAS_MAX_J];
    for (int d = 0; d < md.D; ++d) dim_sines_point<NX>(md, d, xv, ut, sv[d]);
    for (int m = 0; m < md.M; ++m) {
        double f = md.nrm;
        for (int d = 0; d < md.D; ++d) f = f * sv[d][(idx[m * md.D + d] - md.j0[d]) / md.jstep[d]];
        phi[(size_t)r * Mp + m] = f;
    }
}

typedef double pg_double4 __attribute__((ext_vector_type(4)));

// One wave per 16x16 tile of T1.  A operand of lane l: Phi[r0 + (l>>4)][i0 + (l&15)], B operand:
// Phi[r0 + (l>>4)][j0 + (l&15)]; accumulator register i of lane l: row (l>>4) + 4 i, column l&15
// (f64 MFMA layout, cdna_hip_programming.md section 3).
__global__ __launch_bounds__(64) void k_syrk_mfma(const double* __restrict__ phi, int Rp, int Mp, int M, double* __restrict__ T1) {
    const int lane = threadIdx.x;
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    const int kk = lane >> 4, cc = lane & 15;
    pg_double4 acc = {0.0, 0.0, 0.0, 0.0};
    const double* __restrict__ pa = phi + (size_t)kk * Mp + i0 + cc;
    const double* __restrict__ pb = phi + (size_t)kk * Mp + j0 + cc;
    for (int r0 = 0; r0 < Rp; r0 += 4) {
        const double a = pa[(size_t)r0 * Mp];
        const double b = pb[(size_t)r0 * Mp];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i0 + kk + 4 * i, col = j0 + cc;
        if (row < M && col < M) T1[(size_t)row * M + col] = acc[i];
    }
}

// blocks 0..ceil(M/64)-1: T0[m][k] = sum_r Phi[r][m] X+[r][k];  last block: T2 = X+^T X+
__global__ __launch_bounds__(64) void k_t0t2(const double* __restrict__ phi, const double* __restrict__ traj, int R, int Mp, int M, int nx,
                                              double* __restrict__ T0, double* __restrict__ T2) {
    const int nb0 = (M + 63) / 64;
    if ((int)blockIdx.x < nb0) {
        const int m = blockIdx.x * 64 + threadIdx.x;
        if (m >= M) return;
        double acc[2] = {0.0, 0.0};
        for (int r = 0; r < R; ++r) {
            const double f = phi[(size_t)r * Mp + m];
            for (int k = 0; k < nx; ++k) acc[k] = PGAS_FMA(f, traj[(size_t)(r + 1) * nx + k], acc[k]);
        }
        for (int k = 0; k < nx; ++k) T0[(size_t)m * nx + k] = acc[k];
    } else {
        const int e = threadIdx.x;
        if (e >= nx * nx) return;
        const int a = e / nx, b = e % nx;
        double acc = 0.0;
        for (int r = 0; r < R; ++r) acc = PGAS_FMA(traj[(size_t)(r + 1) * nx + a], traj[(size_t)(r + 1) * nx + b], acc);
        T2[e] = acc;
    }
}
