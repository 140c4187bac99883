// Probe (r04): binary Gram products on the MX-fp4 matrix instruction of gfx950.  One wavefront multiplies a 32 x 64 bit matrix A with a
// 64 x 32 bit matrix B through v_mfma_scale_f32_32x32x64_f8f6f4 (both operands E2M1, block scale 2^0): every lane expands the 64 bits
// of its row (lane & 31; K half lane >> 5 -> 32 samples) into 32 nibbles with code 0b0010 = 1.0, register q = bit q of each nibble.
// Prints the number of accumulator entries that differ from the popcounts computed on the host.
// build: hipcc --offload-arch=gfx950 -O2 profiles/tools/mfma_fp4_probe.cpp -o profiles/tools/mfma_fp4_probe.bin   (*.bin is git-ignored; never under the package directory)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void probe(const unsigned long long *A, const unsigned long long *B, float *C /* [32][32] row-major */, int variant)
{
    const int lane = threadIdx.x & 63, row = lane & 31, half = lane >> 5;
    const unsigned wa = (unsigned)(A[row] >> (32 * half)), wb = (unsigned)(B[row] >> (32 * half));
    v8i a, b;
    for (int q = 0; q < 8; ++q) a[q] = b[q] = 0;
    if (variant == 0) {  // code 0b0010 (1.0), scale 2^0
        a[0] = (int)((wa << 1) & 0x22222222u); a[1] = (int)(wa & 0x22222222u); a[2] = (int)((wa >> 1) & 0x22222222u); a[3] = (int)((wa >> 2) & 0x22222222u);
        b[0] = (int)((wb << 1) & 0x22222222u); b[1] = (int)(wb & 0x22222222u); b[2] = (int)((wb >> 1) & 0x22222222u); b[3] = (int)((wb >> 2) & 0x22222222u);
    } else {  // code 0b0001 (0.5): results are counts / 4
        a[0] = (int)(wa & 0x11111111u); a[1] = (int)((wa >> 1) & 0x11111111u); a[2] = (int)((wa >> 2) & 0x11111111u); a[3] = (int)((wa >> 3) & 0x11111111u);
        b[0] = (int)(wb & 0x11111111u); b[1] = (int)((wb >> 1) & 0x11111111u); b[2] = (int)((wb >> 2) & 0x11111111u); b[3] = (int)((wb >> 3) & 0x11111111u);
    }
    v16f acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, 127, 0, 127);
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * half, j = lane & 31;  // C layout of the 32x32 forms: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
        C[i * 32 + j] = acc[r];
    }
}

int main()
{
    unsigned long long hA[32], hB[32];
    srand(7);
    for (int i = 0; i < 32; ++i) {
        hA[i] = ((unsigned long long)rand() << 33) ^ ((unsigned long long)rand() << 11) ^ (unsigned long long)rand();
        hB[i] = ((unsigned long long)rand() << 33) ^ ((unsigned long long)rand() << 11) ^ (unsigned long long)rand();
    }
    unsigned long long *dA, *dB;
    float *dC, hC[1024];
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    for (int variant = 0; variant < 2; ++variant) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, variant);
        if (hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 1; }
        int bad = 0, badT = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                const float e = (float)__builtin_popcountll(hA[i] & hB[j]) * (variant ? 0.25f : 1.0f);
                bad += hC[i * 32 + j] != e;
                badT += hC[j * 32 + i] != e;
            }
        printf("variant %d: %d of 1024 entries differ (transposed reading: %d); C[0][0..3] = %g %g %g %g, expected %d %d %d %d\n", variant, bad, badT,
               hC[0], hC[1], hC[2], hC[3], __builtin_popcountll(hA[0] & hB[0]), __builtin_popcountll(hA[0] & hB[1]), __builtin_popcountll(hA[0] & hB[2]),
               __builtin_popcountll(hA[0] & hB[3]));
    }
    return 0;
}
