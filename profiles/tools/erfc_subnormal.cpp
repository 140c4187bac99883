// Probe (profiling tool): device erfc / exp in the subnormal output range vs host libm.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const double *x, double *e, double *ex, int n) {
    int i = threadIdx.x; if (i < n) { e[i] = erfc(x[i]); ex[i] = exp(-x[i] * x[i]); }
}
int main() {
    const int n = 12; double hx[n], he[n], hex[n], *dx, *de, *dex;
    for (int i = 0; i < n; ++i) hx[i] = 26.0 + 0.12 * i;
    hipMalloc(&dx, n * 8); hipMalloc(&de, n * 8); hipMalloc(&dex, n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, de, dex, n);
    hipMemcpy(he, de, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hex, dex, n * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("x=%.2f dev erfc=%.6g host erfc=%.6g | dev exp(-x^2)=%.6g host=%.6g\n", hx[i], he[i], erfc(hx[i]), hex[i], exp(-hx[i]*hx[i]));
    return 0;
}
