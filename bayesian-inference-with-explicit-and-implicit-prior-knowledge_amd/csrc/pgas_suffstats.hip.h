// pgas_suffstats.hip.h -- MNIW sufficient statistics of one trajectory
// (PGAS.sample_params, reference src/PGAS.py:294-303; prior_mniw_calcStatistics, BI:53-61).
//
// The reference materialises T-1 outer products (T-1,M,M) and sums them; here
//   k_traj_basis  : Z (Rp,Mp) = [ Phi | X+ | 0 ],  Phi[r] = basis(traj[r], inputs[r]),  X+[r] = traj[r+1],  R = T-1   (quirk Q3 pairing)
//   k_syrk_lds    : Z^T Z = [[T1, T0], [T0^T, T2]] on v_mfma_f64_16x16x4_f64: 64x64 blocks of the lower triangle, 16-row panels of Z staged
//                   through LDS (double-buffered, register-prefetched), the T-1 rows split over gridDim.y into partial slabs
//   k_syrk_reduce : sums the slabs in split order (deterministic) and scatters T1 (both triangles), T0 and T2
// Floating-point summation order differs from the reference's sum over t; parity is asserted
// to 1e-12 relative (tests/test_gpu_suffstats.py), not bit for bit.
#pragma once

#include "pgas_kernels.hip.h"

#define SY_RB 16   // rows of Z per k_traj_basis block

// One block per SY_RB rows: threads 0..SY_RB-1 evaluate the per-dimension sines of their row into LDS, then all threads write the rows
// with m fastest (coalesced).  Columns M..M+nx-1 carry X+ = traj[r+1]; pad columns and pad rows are written as zeros.
template <int NX>
__global__ __launch_bounds__(256) void k_traj_basis(DevModel md, const int32_t* __restrict__ idx, const double* __restrict__ traj, int R, int Rp,
                                                     int Mp, double* __restrict__ phi) {
    __shared__ double sv[SY_RB][PGAS_MAX_D][PGAS_MAX_J];
    const int tid = threadIdx.x, rb = blockIdx.x * SY_RB;
    if (tid < SY_RB && rb + tid < R) {
        const int r = rb + tid;
        const double* __restrict__ ut = md.u + (size_t)r * md.nu;
        double xv[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) xv[k] = traj[(size_t)r * NX + k];
        for (int d = 0; d < md.D; ++d) dim_sines_point<NX>(md, d, xv, ut, sv[tid][d]);
    }
    __syncthreads();
    for (int m = tid; m < Mp; m += 256) {
        int q[PGAS_MAX_D] = {0, 0, 0};
        if (m < md.M)
            for (int d = 0; d < md.D; ++d) q[d] = (idx[m * md.D + d] - md.j0[d]) / md.jstep[d];
        for (int i = 0; i < SY_RB; ++i) {
            const int r = rb + i;
            if (r >= Rp) break;
            double f = 0.0;
            if (r < R) {
                if (m < md.M) {
                    f = md.nrm;
                    for (int d = 0; d < md.D; ++d) f = f * sv[i][d][q[d]];
                } else if (m < md.M + NX) {
                    f = traj[(size_t)(r + 1) * NX + (m - md.M)];
                }
            }
            phi[(size_t)r * Mp + m] = f;
        }
    }
}

typedef double pg_double4 __attribute__((ext_vector_type(4)));
typedef double pg_double2 __attribute__((ext_vector_type(2)));

#define SY_BM 64   // block of Z^T Z per workgroup (4 waves, 32x32 each = 2x2 MFMA tiles)
#define SY_KB 16   // rows of Z per LDS panel
#define SY_LD 80   // LDS row stride in doubles: rows k and k+1 of a ds_read_b64 fragment fall into disjoint halves of the 64 banks

// Lower-triangle block b -> (bi, bj), bi >= bj.
__device__ __forceinline__ void tri_block(int b, int& bi, int& bj) {
    int i = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= b) ++i;
    while (i * (i + 1) / 2 > b) --i;
    bi = i;
    bj = b - i * (i + 1) / 2;
}

// MFMA operand layout (f64 16x16x4, cdna_hip_programming.md section 3): A operand of lane l = A[row l&15][k l>>4], B operand = B[k l>>4][col l&15],
// accumulator register i of lane l = C[row (l>>4) + 4 i][col l&15].  Here A[i][k] = Z[k][I+i] and B[k][j] = Z[k][J+j], so both operands are read
// from a panel row k at 16 consecutive columns.
__global__ __launch_bounds__(256) void k_syrk_lds(const double* __restrict__ phi, int Rp, int Mp, int kb_per_split, double* __restrict__ ws) {
    __shared__ double pa[2][SY_KB][SY_LD];
    __shared__ double pb[2][SY_KB][SY_LD];
    int bi, bj;
    tri_block(blockIdx.x, bi, bj);
    const bool diag = bi == bj;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = tid >> 4, lc = (tid & 15) * 4;                // this thread's 4 doubles of a 16 x 64 panel
    const int nkb = Rp / SY_KB;
    const int kb0 = blockIdx.y * kb_per_split, kb1 = min(kb0 + kb_per_split, nkb);
    const double* __restrict__ ga = phi + (size_t)lr * Mp + bi * SY_BM + lc;
    const double* __restrict__ gb = phi + (size_t)lr * Mp + bj * SY_BM + lc;
    pg_double4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = pg_double4{0.0, 0.0, 0.0, 0.0};
    // register prefetch two panels ahead: the global-load latency (> 1 us under load) is longer than one panel's 16 MFMAs per wave
    pg_double2 ra[2][2], rb[2][2];
    auto fetch = [&](int kb, int slot) {
        if (kb < kb1) {
            const size_t o = (size_t)kb * SY_KB * Mp;
            ra[slot][0] = *(const pg_double2*)(ga + o); ra[slot][1] = *(const pg_double2*)(ga + o + 2);
            if (!diag) { rb[slot][0] = *(const pg_double2*)(gb + o); rb[slot][1] = *(const pg_double2*)(gb + o + 2); }
        }
    };
    fetch(kb0, 0);
    fetch(kb0 + 1, 1);
    const int wr = (w >> 1) * 32 + (lane & 15), wc = (w & 1) * 32 + (lane & 15), kq = lane >> 4;
    int cur = 0;
    for (int kb = kb0; kb < kb1; ++kb) {
        *(pg_double2*)&pa[cur][lr][lc] = ra[0][0]; *(pg_double2*)&pa[cur][lr][lc + 2] = ra[0][1];
        if (!diag) { *(pg_double2*)&pb[cur][lr][lc] = rb[0][0]; *(pg_double2*)&pb[cur][lr][lc + 2] = rb[0][1]; }
        __syncthreads();   // one barrier per panel: the buffer written here was last read two iterations ago, before the previous barrier
        ra[0][0] = ra[1][0]; ra[0][1] = ra[1][1]; rb[0][0] = rb[1][0]; rb[0][1] = rb[1][1];
        fetch(kb + 2, 1);
        const double (*A)[SY_LD] = pa[cur];
        const double (*B)[SY_LD] = diag ? pa[cur] : pb[cur];
#pragma unroll
        for (int kk = 0; kk < SY_KB / 4; ++kk) {
            const int k = kk * 4 + kq;
            const double a0 = A[k][wr], a1 = A[k][wr + 16], b0 = B[k][wc], b1 = B[k][wc + 16];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        cur ^= 1;
    }
    double* __restrict__ out = ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (SY_BM * SY_BM);
    const int r0 = (w >> 1) * 32 + (lane >> 4), c0 = (w & 1) * 32 + (lane & 15);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) out[(r0 + 16 * i + 4 * e) * SY_BM + c0 + 16 * j] = acc[i][j][e];
}

// Z^T Z element (gi, gj) of block (bi, bj): rows/columns < M belong to T1, M..M+nx-1 to X+.
__global__ __launch_bounds__(256) void k_syrk_reduce(const double* __restrict__ ws, int ntri, int S, int M, int nx, double* __restrict__ T0,
                                                      double* __restrict__ T1, double* __restrict__ T2) {
    int bi, bj;
    tri_block(blockIdx.x, bi, bj);
    const int e = blockIdx.y * 256 + threadIdx.x;
    const int gi = bi * SY_BM + e / SY_BM, gj = bj * SY_BM + e % SY_BM;
    if (gi >= M + nx || gj >= M + nx) return;
    double v = 0.0;
    for (int z = 0; z < S; ++z) v += ws[((size_t)z * ntri + blockIdx.x) * (SY_BM * SY_BM) + e];
    const bool mirror = bi != bj;   // a diagonal block holds both triangles itself
    if (gi < M && gj < M) {
        T1[(size_t)gi * M + gj] = v;
        if (mirror) T1[(size_t)gj * M + gi] = v;
    } else if (gi >= M && gj < M) {
        T0[(size_t)gj * nx + (gi - M)] = v;
    } else if (gi < M && gj >= M) {
        if (mirror) T0[(size_t)gi * nx + (gj - M)] = v;   // only when the X+ columns straddle a block boundary
    } else {
        T2[(gi - M) * nx + (gj - M)] = v;
        if (mirror) T2[(gj - M) * nx + (gi - M)] = v;
    }
}
