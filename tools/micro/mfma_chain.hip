// micro-benchmark: v_mfma_f32_32x32x16_f16 issued (a) three times in a row into the SAME accumulator, tile after tile (the
// order of the policy kernel's layer-2 loop) against (b) the three terms interleaved across two tiles, and (c) ten independent
// accumulators round-robin; one wave per SIMD, 10 accumulator tiles = 160 AGPRs, cycles from s_memtime / clock64.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float *out, long long *cyc, int iters, float seed) {
    f32x16 acc[10];
    for (int i = 0; i < 10; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    f16x8 a0, a1, b0, b1;
    for (int j = 0; j < 8; ++j) { a0[j] = (_Float16)(seed + threadIdx.x * 1e-3f); a1[j] = (_Float16)(seed * 0.5f); b0[j] = (_Float16)1.f; b1[j] = (_Float16)0.25f; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[u], 0, 0, 0);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 10; u += 2) {
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[u], 0, 0, 0);
                acc[u + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[u + 1], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[u], 0, 0, 0);
                acc[u + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[u + 1], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[u], 0, 0, 0);
                acc[u + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[u + 1], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int u = 0; u < 10; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(t == 1 ? a1 : a0, t == 0 ? b1 : b0, acc[u], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 10; ++i) for (int v = 0; v < 16; ++v) s += acc[i][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
    float *out; long long *cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64);
    const int iters = 2000;
    const char *names[3] = {"3 in a row per tile", "interleaved across 2 tiles", "term-major over 10 tiles"};
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters, 1.f);
        hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters, 1.f);
        hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters, 1.f);
        hipDeviceSynchronize();
    }
    long long h[3]; hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost);
    for (int m = 0; m < 3; ++m) printf("%-28s: %.2f shader cycles per MFMA\n", names[m], (double)h[m] / (iters * 30.0));
    return 0;
}
