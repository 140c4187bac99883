// Micro-benchmark (profiling tool, not product): host cost of one small round trip in different staging schemes.
// hipcc --offload-arch=gfx950 -O2 launch_latency.cpp -o launch_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k(const int *in, int *out, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = in[i] + 1; }
int main() {
    const int n = 4096, iters = 3000;
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int *hin, *hout, *din, *dout; hipHostMalloc(&hin, n * 4); hipHostMalloc(&hout, n * 4); hipMalloc(&din, n * 4); hipMalloc(&dout, n * 4);
    memset(hin, 0, n * 4);
    int *hin_d, *hout_d; hipHostGetDevicePointer((void **)&hin_d, hin, 0); hipHostGetDevicePointer((void **)&hout_d, hout, 0);
    for (int mode = 0; mode < 5; ++mode) {
        for (int w = 0; w < 2; ++w) {
            double t0 = now();
            for (int it = 0; it < iters; ++it) {
                if (mode == 0) {  // copies + events + sync (current)
                    hipMemcpyAsync(din, hin, n * 4, hipMemcpyHostToDevice, s); hipEventRecord(e0, s);
                    hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, s, din, dout, n); hipEventRecord(e1, s);
                    hipMemcpyAsync(hout, dout, n * 4, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                } else if (mode == 1) {  // copies, no events
                    hipMemcpyAsync(din, hin, n * 4, hipMemcpyHostToDevice, s);
                    hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, s, din, dout, n);
                    hipMemcpyAsync(hout, dout, n * 4, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
                } else if (mode == 2) {  // zero-copy in and out, no events
                    hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, s, hin_d, hout_d, n); hipStreamSynchronize(s);
                } else if (mode == 3) {  // zero-copy + events
                    hipEventRecord(e0, s); hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, s, hin_d, hout_d, n); hipEventRecord(e1, s);
                    hipStreamSynchronize(s); float ms; hipEventElapsedTime(&ms, e0, e1);
                } else {  // zero-copy out, H2D copy in, no events
                    hipMemcpyAsync(din, hin, n * 4, hipMemcpyHostToDevice, s);
                    hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, s, din, hout_d, n); hipStreamSynchronize(s);
                }
            }
            double dt = now() - t0;
            if (w == 1) printf("mode %d: %.2f us per round trip\n", mode, 1e6 * dt / iters);
        }
    }
    return 0;
}
