// Probe: do v_mfma_f64_16x16x4_f64 and fp64 vector FMAs of two waves on the SAME SIMD run concurrently on gfx950, or do they share
// the fp64 datapath?  One workgroup of 512 threads per CU = 8 waves = 2 per SIMD (waves w and w + 4 share SIMD w % 4).
//   mode 0: waves 0-3 run NM dependent-free MFMAs each, waves 4-7 exit        -> T_mfma
//   mode 1: waves 4-7 run NV fp64 FMAs each (8 independent chains), 0-3 exit  -> T_valu
//   mode 2: both at once                                                       -> max(T_mfma, T_valu) if the units are separate, the sum if shared
//   mode 3: waves 0-7 all MFMA; mode 4: waves 0-7 all VALU (two waves of the same kind per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(int nm, int nv, double* out, double seed) {
    const int w = threadIdx.x >> 6;
    const bool do_m = MODE == 0 ? w < 4 : MODE == 2 ? w < 4 : MODE == 3;
    const bool do_v = MODE == 1 ? w >= 4 : MODE == 2 ? w >= 4 : MODE == 4;
    double r = 0.0;
    if (do_m) {
        d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        const double a = seed + threadIdx.x, b = seed * 0.5;
        for (int i = 0; i < nm; i += 4) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a3, 0, 0, 0);
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    }
    if (do_v) {
        double c[8];
        for (int j = 0; j < 8; ++j) c[j] = seed + j;
        const double m = 1.0 + seed * 1e-9, d = seed * 1e-3;
        for (int i = 0; i < nv; i += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = __builtin_fma(c[j], m, d);
        for (int j = 0; j < 8; ++j) r += c[j];
    }
    if (r == 12345.678) out[threadIdx.x] = r;
}
template <int MODE>
float run(int nm, int nv, double* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, nm, nv, out, 1.25);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, nm, nv, out, 1.25);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    double* out;
    hipMalloc(&out, 4096);
    const int nm = 1 << 16, nv = 1 << 20;   // 65536 MFMAs x 64 cycles = 4.2 M cycles; 1 M FMAs x 4 cycles = 4.2 M cycles
    const float t0 = run<0>(nm, nv, out), t1 = run<1>(nm, nv, out), t2 = run<2>(nm, nv, out), t3 = run<3>(nm, nv, out), t4 = run<4>(nm, nv, out);
    printf("one MFMA wave per SIMD: %.3f ms (%.1f cycles per MFMA at 2.4 GHz)\n", t0, t0 * 2.4e6 / nm);
    printf("one VALU wave per SIMD: %.3f ms (%.2f cycles per fp64 FMA)\n", t1, t1 * 2.4e6 / nv);
    printf("MFMA wave + VALU wave on the same SIMD: %.3f ms  (max = %.3f, sum = %.3f)\n", t2, t0 > t1 ? t0 : t1, t0 + t1);
    printf("two MFMA waves per SIMD: %.3f ms; two VALU waves per SIMD: %.3f ms\n", t3, t4);
    return 0;
}
