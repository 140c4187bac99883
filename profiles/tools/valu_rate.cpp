// Micro-benchmark (profiling tool, not product): issue cost of the vector instructions the Fisher-z segment kernel is made of, in
// shader cycles per wave64 instruction and SIMD, with 4 wavefronts per SIMD and 8 independent dependency chains per lane (so that
// latency is hidden and the figure is the pipe's throughput).  hipcc --offload-arch=gfx950 -O3 valu_rate.cpp -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_IT 4096
#define CHAINS 8
template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double seed)
{
    double a[CHAINS];
    float f[CHAINS];
    for (int c = 0; c < CHAINS; ++c) {
        a[c] = seed + 1e-3 * (threadIdx.x + 64 * c);
        f[c] = (float)a[c];
    }
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == 0) a[c] = __builtin_fma(a[c], 1.0000001, 1e-9);
            if (OP == 1) a[c] = a[c] * 1.0000001;
            if (OP == 2) a[c] = a[c] + 1e-9;
            if (OP == 3) a[c] = __builtin_amdgcn_rcp(a[c]);
            if (OP == 4) a[c] = __builtin_amdgcn_rsq(a[c]);
            if (OP == 5) a[c] = __builtin_rint(a[c]);
            if (OP == 6) a[c] = __builtin_amdgcn_div_fixup(a[c], 1.5, a[c]);
            if (OP == 7) a[c] = __builtin_fmax(a[c], 0.25);
            if (OP == 8) f[c] = __builtin_fmaf(f[c], 1.0000001f, 1e-9f);
            if (OP == 9) f[c] = __builtin_amdgcn_rcpf(f[c]);
            if (OP == 10) f[c] = __builtin_sqrtf(f[c]) ;
            if (OP == 11) a[c] = (double)(float)a[c];                  // two conversions
            if (OP == 12) a[c] = (a[c] > 0.5) ? a[c] : 0.75;           // compare + 2 x cndmask
            if (OP == 13) a[c] = 1.0 / a[c];                           // the compiler's IEEE division
            if (OP == 14) a[c] = sqrt(a[c]);                           // the compiler's IEEE square root
        }
    }
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += a[c] + f[c];
    if (s == 12345.678) out[0] = s;
}
template <int OP>
void run(const char *name, int per)
{
    double *d;
    hipMalloc(&d, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * 4;  // 4 workgroups of 4 wavefronts per CU: 4 wavefronts per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    // wave-instructions per SIMD: 4 waves x N_IT x CHAINS x per
    const double insts = 4.0 * N_IT * CHAINS * per;
    printf("%-28s %8.3f ms  %6.2f ns per wave-instruction and SIMD  = %5.1f cycles at %.2f GHz (nominal)\n", name, ms, 1e6 * ms / insts,
           1e6 * ms / insts * clk_khz * 1e-6, clk_khz * 1e-6);
    hipFree(d);
}
int main()
{
    run<0>("v_fma_f64", 1);
    run<1>("v_mul_f64", 1);
    run<2>("v_add_f64", 1);
    run<3>("v_rcp_f64", 1);
    run<4>("v_rsq_f64", 1);
    run<5>("v_rndne_f64", 1);
    run<6>("v_div_fixup_f64", 1);
    run<7>("v_max_f64", 1);
    run<8>("v_fma_f32", 1);
    run<9>("v_rcp_f32", 1);
    run<10>("sqrtf (IEEE seq.)", 1);
    run<11>("cvt f64->f32->f64 (2 insts)", 2);
    run<12>("cmp_f64 + 2 cndmask (3 insts)", 3);
    run<13>("1.0 / x  f64 (IEEE seq.)", 1);
    run<14>("sqrt f64 (IEEE seq.)", 1);
    return 0;
}
