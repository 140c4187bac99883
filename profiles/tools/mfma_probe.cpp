// Probe (profiling tool): pure v_mfma_f32_32x32x2_f32 issue rate with 4 accumulators per wave, 4 waves per workgroup,
// optionally with the LDS-read pattern of fz_cor_gemm_kernel.  hipcc --offload-arch=gfx950 -O3 mfma_probe.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void probe(float *out, int iters)
{
    __shared__ __attribute__((aligned(16))) float sA[128 * 36], sB[128 * 36];
    for (int i = threadIdx.x; i < 128 * 36; i += 256) { sA[i] = 0.001f * (i & 63); sB[i] = 0.002f * (i & 31); }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1, lm = lane & 31, lh = lane >> 5;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float av0[4] = {1.f, 2.f, 3.f, 4.f}, av1[4] = {.5f, .25f, .125f, .1f}, bv0[4] = {1.f, 1.f, 2.f, 2.f}, bv1[4] = {3.f, 1.f, 4.f, 1.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODE >= 1) {
                float4 a0 = *reinterpret_cast<const float4 *>(&sA[(wm * 64 + lm) * 36 + lh * 16 + q * 4]);
                float4 a1 = *reinterpret_cast<const float4 *>(&sA[(wm * 64 + 32 + lm) * 36 + lh * 16 + q * 4]);
                float4 b0 = *reinterpret_cast<const float4 *>(&sB[(wn * 64 + lm) * 36 + lh * 16 + q * 4]);
                float4 b1 = *reinterpret_cast<const float4 *>(&sB[(wn * 64 + 32 + lm) * 36 + lh * 16 + q * 4]);
                av0[0] = a0.x; av0[1] = a0.y; av0[2] = a0.z; av0[3] = a0.w; av1[0] = a1.x; av1[1] = a1.y; av1[2] = a1.z; av1[3] = a1.w;
                bv0[0] = b0.x; bv0[1] = b0.y; bv0[2] = b0.z; bv0[3] = b0.w; bv1[0] = b1.x; bv1[1] = b1.y; bv1[2] = b1.z; bv1[3] = b1.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv0[e], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv1[e], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv0[e], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv1[e], acc[1][1], 0, 0, 0);
            }
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, int blocks) {
    float *d; hipMalloc(&d, (size_t)blocks * 256 * 4); const int iters = 2000;
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10); hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); hipDeviceSynchronize();
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double fl = (double)blocks * 4 * 64 * 4096.0 * iters;
    printf("%s blocks=%d: %.1f TFLOP/s\n", name, blocks, fl / dt / 1e12); hipFree(d);
}
int main() {
    for (int blocks : {256, 512, 768, 3072}) { run<0>("regs only       ", blocks); run<1>("+ LDS reads     ", blocks); run<2>("+ LDS + barrier ", blocks); }
    return 0;
}
