// micro-benchmark: achievable f32-input MFMA rate (v_mfma_f32_16x16x4_f32), 1 and 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float av = a + threadIdx.x * 1e-6f, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 1024}) {
        const int iters = 4000;
        hipLaunchKernelGGL(k<20>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.f, 1.f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<20>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)blocks * 4 * iters * 20 * 16 * 16 * 4 * 2;
        printf("blocks %d (%.1f waves/SIMD): %.3f ms, %.1f TFLOP/s\n", blocks, blocks / 256.0, ms, flop / ms / 1e9);
    }
    return 0;
}
