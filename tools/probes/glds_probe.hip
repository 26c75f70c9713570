// Probe: semantics of __builtin_amdgcn_global_load_lds (16-byte form) on gfx950 as k_step uses it:
// per-lane global source, wave-uniform LDS base, lane l lands at base + 16 l.  Prints "glds ok" or the first mismatch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(const double* __restrict__ src, double* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) double buf[4][1024];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int g = 0; g < 4; ++g)
        for (int i = 0; i < 2; ++i) {
            // wave w moves elements [w*256 + i*128, +128) of segment g: lane l the pair (2l, 2l+1)
            const int e0 = wave * 256 + i * 128;
            const double* gp = src + (size_t)g * 1024 + e0 + 2 * lane;
            __builtin_amdgcn_global_load_lds(gp, (__attribute__((address_space(3))) void*)&buf[g][e0], 16, 0, 0);
        }
    __syncthreads();
    for (int g = 0; g < 4; ++g)
        for (int j = 0; j < 4; ++j) out[g * 1024 + j * 256 + tid] = buf[g][j * 256 + tid] * 2.0;
}
int main() {
    std::vector<double> h(4096), o(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i + 0.5;
    double *d, *r;
    hipMalloc(&d, 4096 * 8); hipMalloc(&r, 4096 * 8);
    hipMemcpy(d, h.data(), 4096 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, r);
    hipMemcpy(o.data(), r, 4096 * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4096; ++i)
        if (o[i] != 2.0 * h[i]) { printf("glds MISMATCH at %d: %g vs %g\n", i, o[i], 2.0 * h[i]); return 1; }
    printf("glds ok\n");
    return 0;
}
