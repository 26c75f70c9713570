// Probe: sustained rate of v_fma_f64 on one gfx950 SIMD by operand kind and waves per SIMD (1 workgroup per CU, 256 CUs busy).
//   KIND 0: c = fma(c, m, d) with m, d in VGPRs (three VGPR sources)      KIND 1: m in an SGPR (two VGPR sources)
//   KIND 2: v_mul_f64 c = c * m (two VGPR sources)                        KIND 3: v_add_f64
// Prints cycles per instruction per SIMD at a nominal 2.4 GHz -- 4.0 would be the 78.6 TFLOP/s vector peak.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void k(int n, double* out, double seed, const double* mv) {
    double c[12];
    for (int j = 0; j < 12; ++j) c[j] = seed + j + threadIdx.x;
    double m = KIND == 1 ? seed * 1.0000001 : mv[threadIdx.x & 1], d = mv[2 + (threadIdx.x & 1)];
    for (int i = 0; i < n; i += 12)
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            if (KIND == 0 || KIND == 1) c[j] = __builtin_fma(c[j], m, d);
            if (KIND == 2) c[j] = c[j] * m;
            if (KIND == 3) c[j] = c[j] + d;
        }
    double r = 0;
    for (int j = 0; j < 12; ++j) r += c[j];
    if (r == 12345.678) out[threadIdx.x] = r;
}
template <int KIND>
float run(int threads, int n, double* out, const double* mv) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, n, out, 1.25, mv);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, n, out, 1.25, mv);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    double *out, *mv;
    (void)hipMalloc(&out, 8192);
    (void)hipMalloc(&mv, 64);
    const double h[4] = {1.0000001, 1.0000002, 1e-3, 2e-3};
    (void)hipMemcpy(mv, h, 32, hipMemcpyHostToDevice);
    const int n = 12 << 16;
    const char* names[4] = {"v_fma_f64 (3 VGPR sources)", "v_fma_f64 (SGPR multiplier)", "v_mul_f64", "v_add_f64"};
    for (int kind = 0; kind < 4; ++kind)
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int threads = 256 * wps;
            const float ms = kind == 0 ? run<0>(threads, n, out, mv) : kind == 1 ? run<1>(threads, n, out, mv) : kind == 2 ? run<2>(threads, n, out, mv) : run<3>(threads, n, out, mv);
            printf("%-30s %d wave(s) per SIMD, 12 independent chains each: %.3f ms = %.2f cycles per instruction per SIMD\n", names[kind], wps, ms, ms * 2.4e6 / ((double)n * wps));
        }
    return 0;
}
