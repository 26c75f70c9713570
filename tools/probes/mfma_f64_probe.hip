// Probe: is v_mfma_f64_16x16x4_f64 (D = A B + C, K = 4) bit-identical to the sequential chain
//   d = fma(a[i][3], b[3][j], fma(a[i][2], b[2][j], fma(a[i][1], b[1][j], fma(a[i][0], b[0][j], c))))   (k ascending)
// or to the descending one, or to neither?  Decides whether a matrix-core contraction can keep the canonical order of DESIGN.md 4.2.
// Layout (gfx950, one wave): A: lane l holds A[i = l % 16][k = l / 16]; B: lane l holds B[k = l / 16][j = l % 16];
// C/D: 4 values per lane, D[i = 4 * r + l / 16][j = l % 16], r = 0..3 (the layout csrc/pgas_suffstats.hip.h stores its accumulators with).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, const double* C, double* D) {
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16], b = B[(l / 16) * 16 + l % 16];
    d4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[(4 * r + l / 16) * 16 + l % 16];
    d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * r + l / 16) * 16 + l % 16] = d[r];
}
int main() {
    std::mt19937_64 g(7);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    int asc = 0, desc = 0, neither = 0, trials = 200;
    for (int t = 0; t < trials; ++t) {
        std::vector<double> A(64), B(64), C(256), D(256);
        for (auto& v : A) v = u(g) * std::ldexp(1.0, (int)(u(g) * 20));
        for (auto& v : B) v = u(g) * std::ldexp(1.0, (int)(u(g) * 20));
        for (auto& v : C) v = (t % 2) ? 0.0 : u(g);
        double *dA, *dB, *dC, *dD;
        hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048);
        hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
        bool a_ok = true, d_ok = true;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double up = C[i * 16 + j], dn = C[i * 16 + j];
                for (int kk = 0; kk < 4; ++kk) up = std::fma(A[i * 4 + kk], B[kk * 16 + j], up);
                for (int kk = 3; kk >= 0; --kk) dn = std::fma(A[i * 4 + kk], B[kk * 16 + j], dn);
                a_ok = a_ok && up == D[i * 16 + j];
                d_ok = d_ok && dn == D[i * 16 + j];
            }
        asc += a_ok; desc += d_ok; neither += !a_ok && !d_ok;
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
    }
    printf("mfma_f64_16x16x4 over %d random trials: == ascending fma chain in %d, == descending chain in %d, neither in %d\n", trials, asc, desc, neither);
    return 0;
}
