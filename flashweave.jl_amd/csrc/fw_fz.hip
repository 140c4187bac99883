// FlashWeave-S (Fisher-z) device path for gfx950: level-0 Pearson matrix on fp32 MFMA, level-0 pair tests,
// and the per-(T, candidate) test_subsets batch (recursive partial correlation evaluated as a DP).
//
// Reference semantics (file:line into /root/reference/src):
//   cor(data_dense) -> Float32          learning.jl:42-45 (Statistics.cor: centre, X'X, cov2cor! + clamp)
//   univariate FzTest                    tests.jl:108-160 (branch :149-153), fz_pval statfuns.jl:3-17
//   conditional FzTestCond               tests.jl:250-265, pcor_rec statfuns.jl:23-75 (len_z = 0 at tests.jl:256)
//   test_subsets                         tests.jl:281-346
// Compiled with -ffp-contract=off: the reference never fuses a*b+c and the partial-correlation value must be
// reproducible to the bit (only +,-,*,/,sqrt,rint are involved).
#include "fw_internal.h"
#include "fw_unrank.h"

#include <algorithm>
#include <cmath>

// ------------------------------------------------------------------------------------------------
// 1. centring + column norms
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fz_center_kernel(const float *__restrict__ data, float *__restrict__ xc,
                                                        float *__restrict__ sd, int n, int p, int n_pad)
{
    const int v = blockIdx.x;
    float *dst = xc + (size_t)v * n_pad;
    __shared__ double s_red[4];
    if (v >= p) {
        for (int i = threadIdx.x; i < n_pad; i += 256) dst[i] = 0.0f;
        if (threadIdx.x == 0) sd[v] = 0.0f;
        return;
    }
    const float *src = data + (size_t)v * n;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)src[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
    __syncthreads();
    const double tot = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    const float mean = (float)(tot / (double)n);
    __syncthreads();
    double ss = 0.0;
    for (int i = threadIdx.x; i < n_pad; i += 256) {
        float d = 0.0f;
        if (i < n) {
            d = src[i] - mean;
            ss += (double)d * (double)d;
        }
        dst[i] = d;
    }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) sd[v] = sqrtf((float)(s_red[0] + s_red[1] + s_red[2] + s_red[3]));
}

// ------------------------------------------------------------------------------------------------
// 2. C = Xc' Xc on v_mfma_f32_32x32x2_f32 (exact fp32), upper-triangular 128x128 tiles, fused cov2cor epilogue
//    Xc is stored [variable][n_pad] (k contiguous), so both operands are k-contiguous ("TN" GEMM).
//    LDS tile layout s[128][BK + 4]: 16-byte aligned rows for ds_write_b128 / ds_read_b128, and the row stride
//    of 36 floats keeps the 16-lane b128 read groups conflict free.
//    k is permuted inside a tile: lanes 0-31 take k in [0,16), lanes 32-63 k in [16,32) -- both operands use the
//    same permutation, so the dot product is unchanged while each lane reads 4 consecutive k as one b128.
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GEMM_BM 128
#define GEMM_BK 32
#define GEMM_LD (GEMM_BK + 4)

__device__ __forceinline__ void tri_decode(int b, int T, int &bi, int &bj)
{
    // rows of the upper triangle have T, T-1, ... tiles; small T (<= ~1000) -> a loop is fine but use closed form
    // bi = floor(((2T+1) - sqrt((2T+1)^2 - 8b)) / 2)
    double t = 2.0 * T + 1.0;
    int r = (int)floor((t - sqrt(t * t - 8.0 * (double)b)) * 0.5);
    // fix-up for rounding
    while (r > 0 && (long long)r * T - (long long)r * (r - 1) / 2 > b) --r;
    while ((long long)(r + 1) * T - (long long)(r + 1) * r / 2 <= b) ++r;
    bi = r;
    bj = r + (b - (int)((long long)r * T - (long long)r * (r - 1) / 2));
}

__global__ __launch_bounds__(256) void fz_cor_gemm_kernel(const float *__restrict__ xc, const float *__restrict__ sd,
                                                          float *__restrict__ cor, int p, int n_pad, int T)
{
    // two LDS stages (73.7 KB): while a tile is being multiplied, the next one is already in registers and is written
    // to the other stage right after the MFMA block -- one barrier per k-tile
    __shared__ __attribute__((aligned(16))) float sA[2][GEMM_BM * GEMM_LD];
    __shared__ __attribute__((aligned(16))) float sB[2][GEMM_BM * GEMM_LD];
    int bi, bj;
    tri_decode(blockIdx.x, T, bi, bj);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lm = lane & 31, lh = lane >> 5;

    const float *gA = xc + (size_t)bi * GEMM_BM * n_pad;
    const float *gB = xc + (size_t)bj * GEMM_BM * n_pad;
    // global->LDS staging: 128 columns x 32 k = 1024 float4, 4 per thread; 8 consecutive lanes cover one 128-B row
    const int ld_col = tid >> 3;  // 0..31 (+32 per step)
    const int ld_k4 = tid & 7;    // float4 index within the 32-k row

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    // Staging registers as eight named float4 values: indexing an array from inside a lambda made the compiler keep
    // them in scratch memory (scratch_store/scratch_load around every prefetch, visible in the r01 ISA) and turned the
    // prefetch into a synchronous load.
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    const float *pA = gA + (size_t)ld_col * n_pad + ld_k4 * 4, *pB = gB + (size_t)ld_col * n_pad + ld_k4 * 4;
    const size_t cstep = (size_t)32 * n_pad;
#define GEMM_GLOAD(k0)                                                     \
    do {                                                                   \
        ra0 = *reinterpret_cast<const float4 *>(pA + (k0));                \
        ra1 = *reinterpret_cast<const float4 *>(pA + cstep + (k0));        \
        ra2 = *reinterpret_cast<const float4 *>(pA + 2 * cstep + (k0));    \
        ra3 = *reinterpret_cast<const float4 *>(pA + 3 * cstep + (k0));    \
        rb0 = *reinterpret_cast<const float4 *>(pB + (k0));                \
        rb1 = *reinterpret_cast<const float4 *>(pB + cstep + (k0));        \
        rb2 = *reinterpret_cast<const float4 *>(pB + 2 * cstep + (k0));    \
        rb3 = *reinterpret_cast<const float4 *>(pB + 3 * cstep + (k0));    \
    } while (0)
#define GEMM_SSTORE(st)                                                                              \
    do {                                                                                             \
        float *wa = &sA[st][ld_col * GEMM_LD + ld_k4 * 4], *wb = &sB[st][ld_col * GEMM_LD + ld_k4 * 4]; \
        *reinterpret_cast<float4 *>(wa) = ra0;                                                       \
        *reinterpret_cast<float4 *>(wa + 32 * GEMM_LD) = ra1;                                        \
        *reinterpret_cast<float4 *>(wa + 64 * GEMM_LD) = ra2;                                        \
        *reinterpret_cast<float4 *>(wa + 96 * GEMM_LD) = ra3;                                        \
        *reinterpret_cast<float4 *>(wb) = rb0;                                                       \
        *reinterpret_cast<float4 *>(wb + 32 * GEMM_LD) = rb1;                                        \
        *reinterpret_cast<float4 *>(wb + 64 * GEMM_LD) = rb2;                                        \
        *reinterpret_cast<float4 *>(wb + 96 * GEMM_LD) = rb3;                                        \
    } while (0)

    GEMM_GLOAD(0);
    GEMM_SSTORE(0);
    __syncthreads();
    int st = 0;
    for (int k0 = 0; k0 < n_pad; k0 += GEMM_BK, st ^= 1) {
        const bool more = k0 + GEMM_BK < n_pad;
        if (more) GEMM_GLOAD(k0 + GEMM_BK);  // next tile -> registers (latency hidden behind the MFMA block)
        const float *cA = sA[st], *cB = sB[st];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 a0 = *reinterpret_cast<const float4 *>(&cA[(wm * 64 + lm) * GEMM_LD + lh * 16 + q * 4]);
            float4 a1 = *reinterpret_cast<const float4 *>(&cA[(wm * 64 + 32 + lm) * GEMM_LD + lh * 16 + q * 4]);
            float4 b0 = *reinterpret_cast<const float4 *>(&cB[(wn * 64 + lm) * GEMM_LD + lh * 16 + q * 4]);
            float4 b1 = *reinterpret_cast<const float4 *>(&cB[(wn * 64 + 32 + lm) * GEMM_LD + lh * 16 + q * 4]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b1.x, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b1.y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b0.y, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b1.z, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b0.z, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b1.w, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b0.w, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc[1][1], 0, 0, 0);
        }
        if (more) GEMM_SSTORE(st ^ 1);  // the other stage was last read before the previous barrier
        __syncthreads();
    }
#undef GEMM_GLOAD
#undef GEMM_SSTORE
    // epilogue: cov2cor! (C[i,j] / (xsd[i] * xsd[j]), clampcor, unit diagonal), write (i,j) and the mirror (j,i)
    const bool vec_ok = (p & 3) == 0;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int j = bj * GEMM_BM + wn * 64 + tn * 32 + lm;
            const float sdj = (j < p) ? sd[j] : 0.0f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float vals[4];
                const int i0 = bi * GEMM_BM + wm * 64 + tm * 32 + 8 * g + 4 * lh;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = i0 + e;
                    float c = acc[tm][tn][4 * g + e];
                    const float sdi = (i < p) ? sd[i] : 0.0f;
                    float r = c / (sdi * sdj);
                    r = r > 1.0f ? 1.0f : (r < -1.0f ? -1.0f : r);  // NaN stays NaN
                    if (i == j) r = 1.0f;
                    vals[e] = r;
                    if (i < p && j < p) cor[(size_t)i * p + j] = r;
                }
                if (bi != bj && j < p) {
                    if (vec_ok && i0 + 3 < p) {
                        *reinterpret_cast<float4 *>(&cor[(size_t)j * p + i0]) = make_float4(vals[0], vals[1], vals[2], vals[3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (i0 + e < p) cor[(size_t)j * p + i0 + e] = vals[e];
                    }
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// 3. shared device math
// ------------------------------------------------------------------------------------------------
// out of line for the segment kernel: inlined, the ~70 polynomial coefficients of log / erfc are hoisted into VGPRs
// across the hot loop, which only reaches this code for tests inside the significance guard band
__device__ __noinline__ double fz_pval_slow(double r, double zscale)
{
    double z = (zscale > 0.0) ? zscale * log((1.0 + r) / (1.0 - r)) : 0.0;
    double cc = erfc(fabs(z) * 0.7071067811865476) / 2.0;
    // p-values in the subnormal range (< 2.2e-308, |r| > 0.68 at n = 2000) are flushed to zero.  There erfc implementations
    // differ in the last unit of the subnormal grid (device library vs glibc vs Julia's openlibm: 0 vs 9.9e-324 observed),
    // and since candidates are ordered by p (hiton.jl:212-215) such a difference reorders the conditioning sets of a target.
    // With the flush the order among them is the reference's rule for EQUAL p-values (stable: ascending variable index) --
    // a tie-break the reference leaves to the bits of its libm; the oracle does the same (fwo_fz_pval).
    const double p = cc * 2.0;
    return p < 2.2250738585072014e-308 ? 0.0 : p;
}
// x-key of the max-p tracking: |z| / sqrt2 (see FZ_X_SUB)
__device__ __noinline__ double fz_xkey_slow(double r, double zscale)
{
    return fabs(zscale * log((1.0 + r) / (1.0 - r))) * 0.7071067811865476;
}

__device__ __forceinline__ double fz_pval_dev(double r, double zscale /* sqrt(n-3)/2, 0 if n <= 3 */)
{
    // statfuns.jl:3-17; ccdf(Normal(), x) = erfc(x / sqrt2) / 2 (StatsFuns.normccdf)
    double z = (zscale > 0.0) ? zscale * log((1.0 + r) / (1.0 - r)) : 0.0;
    double cc = erfc(fabs(z) * 0.7071067811865476) / 2.0;
    const double p = cc * 2.0;
    return p < 2.2250738585072014e-308 ? 0.0 : p;  // subnormal p-values flushed: see fz_pval_slow
}

// round(x, digits = 5) = rint(x * 1e5) / 1e5 in the value's own type.  The division of the integer n = rint(x * 1e5)
// by 1e5 is done as q0 = n * c, r = fma(-q0, 1e5, n), q = fma(r, c, q0) with c = RN(1e-5): this is the correctly
// rounded quotient for every integer |n| <= 400000 in both Float32 and Float64 -- verified exhaustively on the host
// (tests/test_oracle_golden.py::test_fast_division_by_1e5_is_exact) -- and costs 3 instructions instead of an IEEE
// division sequence.  Every argument here is a - b * c with a, b, c in [-1, 1] (matrix entries are validated /
// clamped to that range, every recursion level clamps its result), so |n| <= 200000; NaN propagates as in the
// plain division.
__device__ __forceinline__ float round5_f32(float x)
{
    const float n = rintf(x * 100000.0f);
    const float q0 = n * 1e-5f;
    const float r = fmaf(-q0, 100000.0f, n);
    const float y = fmaf(r, 1e-5f, q0);
    return isfinite(y) ? y : x;
}
__device__ __forceinline__ double round5_f64(double x)
{
    const double n = rint(x * 100000.0);
    const double q0 = n * 1e-5;
    const double r = fma(-q0, 100000.0, n);
    const double y = fma(r, 1e-5, q0);
    return isfinite(y) ? y : x;
}

// Partial correlation rho(X, Y | z[0..K-1]) with the reference's peel order (last element first) and its mixed
// Float32/Float64 arithmetic, evaluated bottom-up: U = [X, Y, z_K, ..., z_1]; level j conditions
// every remaining pair (a before b in U) on z_j.  (The recursion of statfuns.jl:44-53 touches exactly these
// pairs in exactly these argument orders; level-1 values are symmetric.)
template <int K>
__device__ __forceinline__ double fz_pcor_dp(const float *__restrict__ cor, int p, int X, int Y, const int *z)
{
    constexpr int M = K + 2;
    int U[M];
    U[0] = X;
    U[1] = Y;
#pragma unroll
    for (int j = 0; j < K; ++j) U[2 + j] = z[K - 1 - j];
    double R[M][M];
    bool is32[M][M];
    float C0[M][M];
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
        for (int b = a + 1; b < M; ++b) C0[a][b] = cor[(size_t)U[a] * p + U[b]];
    // level 1 (statfuns.jl:32-41), ContType = Float32
    {
        constexpr int last = M - 1;
#pragma unroll
        for (int a = 0; a < last; ++a)
#pragma unroll
            for (int b = a + 1; b < last; ++b) {
                const float xy = C0[a][b], xz = C0[a][last], yz = C0[b][last];
                const float prod = xz * yz;
                float e = xy - prod;
                e = round5_f32(e);
                const float s1 = 1.0f - xz * xz, s2 = 1.0f - yz * yz;
                const float d = sqrtf(s1) * sqrtf(s2);
                double v;
                bool f32;
                if (d == 0.0f) {
                    v = 0.0;
                    f32 = false;
                } else {
                    v = (double)(e / d);
                    f32 = true;
                }
                if (v < -1.0) {
                    v = -1.0;
                    f32 = false;
                } else if (v >= 1.0) {
                    v = 1.0;
                    f32 = false;
                }
                R[a][b] = v;
                is32[a][b] = f32;
            }
    }
#pragma unroll
    for (int j = 2; j <= K; ++j) {
        const int last = M - j;
#pragma unroll
        for (int a = 0; a < M; ++a)
#pragma unroll
            for (int b = a + 1; b < M; ++b) {
                if (b < last) {
                    const double va = R[a][b], vb = R[a][last], vc = R[b][last];
                    double ev, d1;
                    if (j == 2) {
                        const bool a32 = is32[a][b], b32 = is32[a][last], c32 = is32[b][last];
                        double prod;
                        bool p32;
                        if (b32 && c32) {
                            prod = (double)((float)vb * (float)vc);
                            p32 = true;
                        } else {
                            prod = vb * vc;
                            p32 = false;
                        }
                        if (a32 && p32)
                            ev = (double)round5_f32((float)va - (float)prod);
                        else
                            ev = round5_f64(va - prod);
                        if (b32) {
                            const float bb = (float)vb * (float)vb;
                            d1 = (double)sqrtf(1.0f - bb);
                        } else {
                            d1 = sqrt(1.0 - vb * vb);
                        }
                    } else {
                        ev = round5_f64(va - vb * vc);
                        d1 = sqrt(1.0 - vb * vb);
                    }
                    const double d2 = sqrt(1.0 - vc * vc);
                    const double denom = d1 * d2;
                    double v = (denom == 0.0) ? 0.0 : ev / denom;
                    if (v < -1.0)
                        v = -1.0;
                    else if (v >= 1.0)
                        v = 1.0;
                    R[a][b] = v;
                    is32[a][b] = false;
                }
            }
    }
    return R[0][1];
}

__device__ __forceinline__ double fz_pcor_any(const float *__restrict__ cor, int p, int X, int Y, const int *z, int k)
{
    switch (k) {
        case 1: return fz_pcor_dp<1>(cor, p, X, Y, z);
        case 2: return fz_pcor_dp<2>(cor, p, X, Y, z);
        case 3: return fz_pcor_dp<3>(cor, p, X, Y, z);
        case 4: return fz_pcor_dp<4>(cor, p, X, Y, z);
        case 5: return fz_pcor_dp<5>(cor, p, X, Y, z);
        default: return (double)cor[(size_t)X * p + Y];
    }
}

// ------------------------------------------------------------------------------------------------
// 4. level 0: all pairs i < j of the resident matrix (tests.jl:149-159 + the NaN/m rule of :397-398,522-526)
// ------------------------------------------------------------------------------------------------
struct FzL0Counters {
    unsigned long long n_sig;  // pairs with p < alpha (raw)
    unsigned long long n_nan;  // pairs with NaN p (excluded from m)
};

// Kernel 1 (screen): p < alpha  <=>  |r| beyond the exact thresholds of fz_thresholds_kernel (lower edge of the
// guard band); only those pairs (3 % at cfg3) go on to the Float64 log / erfc of kernel 2, densely packed, instead of
// every wavefront paying for them.  NaN correlations are counted (they are excluded from m, tests.jl:397-398).
#define FZ_L0_ROWS 8     // rows per workgroup
#define FZ_L0_COLS 1024  // columns per workgroup (4 per thread, one float4 load)
#define FZ_L0_QCAP 2048  // per-workgroup queue of screened pairs (overflow: direct append)
__global__ __launch_bounds__(256) void fz_level0_kernel(const float *__restrict__ cor, int p, const double *__restrict__ thr,
                                                        FzL0Counters *cnt, unsigned long long cap, int32_t *out_i,
                                                        int32_t *out_j, float *out_r)
{
    // 8 x 1024 pairs per workgroup (one row x 256 columns per workgroup was bound by workgroup dispatch), screened
    // pairs queued in LDS and appended with ONE atomic per workgroup (one atomic per wavefront step on the single
    // counter was 4.7 of the kernel's 5 ms: ~700 000 same-address atomics)
    __shared__ int s_qi[FZ_L0_QCAP], s_qj[FZ_L0_QCAP];
    __shared__ float s_qr[FZ_L0_QCAP];
    __shared__ int s_qn;
    __shared__ unsigned long long s_qbase;
    const int i0 = blockIdx.y * FZ_L0_ROWS;
    const int jb = blockIdx.x * FZ_L0_COLS;
    if (jb + FZ_L0_COLS - 1 <= i0) return;  // tile entirely on/below the diagonal
    if (threadIdx.x == 0) s_qn = 0;
    __syncthreads();
    const int j0 = jb + threadIdx.x * 4;
    const int lane = threadIdx.x & 63;
    const float lo_pos = (float)thr[0], lo_neg = (float)thr[2];
    // Float32 screen against thresholds lowered by 1e-6 relative (>> the rounding of the conversion): it can only let a
    // few more pairs through to the exact kernel, never drop one
    const float flo_pos = lo_pos * 0.999999f, flo_neg = lo_neg * 0.999999f;
    unsigned int n_nan = 0;
    for (int ii = 0; ii < FZ_L0_ROWS; ++ii) {
        const int i = i0 + ii;
        if (i >= p) break;
        float rv[4] = {0.f, 0.f, 0.f, 0.f};
        const float *row = cor + (size_t)i * p;
        if (j0 + 3 < p && ((((size_t)i * p + j0) & 3) == 0)) {
            const float4 q = *(const float4 *)(row + j0);
            rv[0] = q.x;
            rv[1] = q.y;
            rv[2] = q.z;
            rv[3] = q.w;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j0 + u < p) rv[u] = row[j0 + u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u;
            const bool in = j > i && j < p;
            const float r = rv[u];
            const bool isn = in && isnan(r);
            n_nan += isn;
            if (in && !isn && fabsf(r) >= (r < 0.0f ? flo_neg : flo_pos)) {
                const int q = atomicAdd(&s_qn, 1);  // LDS
                if (q < FZ_L0_QCAP) {
                    s_qi[q] = i;
                    s_qj[q] = j;
                    s_qr[q] = r;
                } else {
                    const unsigned long long slot = atomicAdd(&cnt->n_sig, 1ull);
                    if (slot < cap) {
                        out_i[slot] = i;
                        out_j[slot] = j;
                        out_r[slot] = r;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_nan += __shfl_xor(n_nan, o);
    if (lane == 0 && n_nan) atomicAdd(&cnt->n_nan, (unsigned long long)n_nan);
    __syncthreads();
    const int nq = s_qn < FZ_L0_QCAP ? s_qn : FZ_L0_QCAP;
    if (threadIdx.x == 0 && nq > 0) s_qbase = atomicAdd(&cnt->n_sig, (unsigned long long)nq);
    __syncthreads();
    for (int q = threadIdx.x; q < nq; q += 256) {
        const unsigned long long slot = s_qbase + (unsigned long long)q;
        if (slot < cap) {
            out_i[slot] = s_qi[q];
            out_j[slot] = s_qj[q];
            out_r[slot] = s_qr[q];
        }
    }
}

// Kernel 2 (exact): Fisher-z p-value of the screened pairs (tests.jl:149-159); keeps p < alpha.
__global__ __launch_bounds__(256) void fz_level0_exact_kernel(const int32_t *__restrict__ ci, const int32_t *__restrict__ cj,
                                                              const float *__restrict__ cr, unsigned long long ncand,
                                                              double alpha, double zscale, FzL0Counters *cnt,
                                                              unsigned long long cap, int32_t *out_i, int32_t *out_j,
                                                              float *out_r, double *out_p)
{
    const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    bool sig = false;
    float r = 0.0f;
    double pv = 1.0;
    if (t < ncand) {
        r = cr[t];
        pv = fz_pval_dev((double)r, zscale);
        sig = pv < alpha;
    }
    const unsigned long long ms = __ballot(sig);
    const int lane = threadIdx.x & 63;
    if (ms) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&cnt->n_sig, (unsigned long long)__popcll(ms));
        base = __shfl(base, 0);
        if (sig) {
            const unsigned long long slot = base + __popcll(ms & ((1ull << lane) - 1ull));
            if (slot < cap) {
                out_i[slot] = ci[t];
                out_j[slot] = cj[t];
                out_r[slot] = r;
                out_p[slot] = pv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 5. batch of single tests (tests.jl:108-160 / 250-265), one lane per test
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fz_test_batch_kernel(const float *__restrict__ cor, int p, long long m,
                                                            const int32_t *__restrict__ X, const int32_t *__restrict__ Y,
                                                            const long long *__restrict__ zoff,
                                                            const int32_t *__restrict__ zflat, double zscale,
                                                            fw_test_result *__restrict__ out)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= m) return;
    const int k = (int)(zoff[t + 1] - zoff[t]);
    int z[FW_MAX_K];
    for (int q = 0; q < FW_MAX_K; ++q) z[q] = (q < k) ? zflat[zoff[t] + q] : 0;
    const double r = fz_pcor_any(cor, p, X[t], Y[t], z, k);
    fw_test_result o;
    o.stat = r;
    o.pval = fz_pval_dev(r, zscale);
    o.df = 0;
    o.suff_power = 1;
    out[t] = o;
}

// ------------------------------------------------------------------------------------------------
// 6. test_subsets: one 256-lane workgroup per (T, candidate, accepted) job.  Subsets are enumerated in the
//    reference order (sizes max_k..1, lexicographic over positions); lane l of a chunk evaluates rank base + l;
//    the first non-significant rank (or the max_tests stop) ends the job, otherwise the (p, rank) maximum with
//    "later wins ties" is carried across chunks (tests.jl:311-345).
// ------------------------------------------------------------------------------------------------
#define FW_ACC_LDS 2048

// C(m, t) and the lexicographic unranking of subset ranks: fw_unrank.h (shared with the host-side exhaustive check)
#define binom_u64 fw_binom_u64
#define unrank_comb fw_unrank_comb

// ---- tagged scalar forms of the pcor_rec levels (same arithmetic as fz_pcor_dp, used by the run-based kernel) ----
struct TV {
    double v;
    bool f32;
};

// sqrt(x) for x in {0} U [2^-52, 1] (here: 1 - v^2 with |v| <= 1 in Float64): the library's sequence (v_rsq_f64 + two
// Goldschmidt / Newton steps, correctly rounded) without its scaling for arguments below 2^-767 and its
// infinity check -- the same instructions on the same values, hence the same bits, 6 instructions less per root.
__device__ __forceinline__ double fz_sqrt_unit(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    g = fma(d, h, g);
    return x == 0.0 ? 0.0 : g;  // rsq(0) = inf would poison g; NaN propagates as in sqrt
}

// statfuns.jl:32-41 with ContType = Float32
__device__ __forceinline__ TV pc_l1(float xy, float xz, float yz)
{
    const float prod = xz * yz;
    float e = xy - prod;
    e = round5_f32(e);
    const float s1 = 1.0f - xz * xz, s2 = 1.0f - yz * yz;
    const float d = sqrtf(s1) * sqrtf(s2);
    TV r;
    if (d == 0.0f) {
        r.v = 0.0;
        r.f32 = false;
    } else {
        r.v = (double)(e / d);
        r.f32 = true;
    }
    if (r.v < -1.0) {
        r.v = -1.0;
        r.f32 = false;
    } else if (r.v >= 1.0) {
        r.v = 1.0;
        r.f32 = false;
    }
    return r;
}

// statfuns.jl:44-62, children from level 1 (Float32 unless they were replaced by a Float64 literal).
// d2c = sqrt(1 - c^2) in Float64 (statfuns.jl:52, `^2.0`) is passed in so that callers can share it.
__device__ __forceinline__ double pc_l2_d2(TV a, TV b, TV c, double d2c)
{
    double prod, ev, d1;
    bool p32;
    if (b.f32 && c.f32) {
        prod = (double)((float)b.v * (float)c.v);
        p32 = true;
    } else {
        prod = b.v * c.v;
        p32 = false;
    }
    if (a.f32 && p32)
        ev = (double)round5_f32((float)a.v - (float)prod);
    else
        ev = round5_f64(a.v - prod);
    if (b.f32) {
        const float bb = (float)b.v * (float)b.v;
        d1 = (double)sqrtf(1.0f - bb);
    } else {
        d1 = sqrt(1.0 - b.v * b.v);
    }
    const double denom = d1 * d2c;
    double v = (denom == 0.0) ? 0.0 : ev / denom;
    v = v < -1.0 ? -1.0 : v;  // two selects, no branch (NaN stays NaN)
    v = v >= 1.0 ? 1.0 : v;
    return v;
}
__device__ __forceinline__ double pc_l2(TV a, TV b, TV c) { return pc_l2_d2(a, b, c, sqrt(1.0 - c.v * c.v)); }

// pc_l2_d2 specialised for the overwhelmingly common case that all three children are Float32 values (no Float64
// literal 0 / +-1 among them): identical arithmetic, no per-flag branches.
__device__ __forceinline__ double pc_l2_all32(float a, float b, float c, double d2c)
{
    const float prod = b * c;
    const double ev = (double)round5_f32(a - prod);
    const float bb = b * b;
    const double d1 = (double)sqrtf(1.0f - bb);
    const double denom = d1 * d2c;
    double v = (denom == 0.0) ? 0.0 : ev / denom;
    v = v < -1.0 ? -1.0 : v;
    v = v >= 1.0 ? 1.0 : v;
    return v;
}

// the same with d1 = Float64(sqrt(1f0 - b^2)) taken from the LDS table (it only depends on the (z1, z2) entry)
__device__ __forceinline__ double pc_l2_all32_d1(float a, float b, float c, double d1, double d2c)
{
    const float prod = b * c;
    const double ev = (double)round5_f32(a - prod);
    const double denom = d1 * d2c;
    double v = (denom == 0.0) ? 0.0 : ev / denom;
    v = v < -1.0 ? -1.0 : v;
    v = v >= 1.0 ? 1.0 : v;
    return v;
}

// pc_l1 with the two square roots sqrt(1 - xz^2), sqrt(1 - yz^2) taken from the LDS table
__device__ __forceinline__ TV pc_l1_r(float xy, float xz, float yz, float rxz, float ryz)
{
    const float prod = xz * yz;
    const float e = round5_f32(xy - prod);
    const float d = rxz * ryz;
    const bool nz = d != 0.0f;
    const float q = e / (nz ? d : 1.0f);
    TV r;
    r.v = nz ? (double)q : 0.0;
    r.f32 = nz;
    const bool lo = r.v < -1.0, hi = r.v >= 1.0;
    r.v = lo ? -1.0 : r.v;
    r.v = hi ? 1.0 : r.v;
    r.f32 = r.f32 && !lo && !hi;
    return r;
}

// statfuns.jl:44-62, all-Float64 children (level >= 3)
__device__ __forceinline__ double pc_l3(double a, double b, double c)
{
    const double ev = round5_f64(a - b * c);
    const double denom = fz_sqrt_unit(1.0 - b * b) * fz_sqrt_unit(1.0 - c * c);
    double v = (denom == 0.0) ? 0.0 : ev / denom;
    v = v < -1.0 ? -1.0 : v;
    v = v >= 1.0 ? 1.0 : v;
    return v;
}

// Run-based segment kernel.  A segment [start, end) is processed in chunks of 256 * R ranks; lane l evaluates the R
// consecutive ranks [cbase + l*R, cbase + (l+1)*R): it unranks once, then steps the combination lexicographically and
// reuses every partial correlation that does not involve the position that changed (for max_k = 3 that is 4 of the
// 10 formula evaluations and 6 of the 10 matrix entries).  Rank order is preserved: a lane stops at its first
// stopping rank, the workgroup takes the minimum over lanes.
#ifndef FW_RUN_MAX
#define FW_RUN_MAX 32  // chunk = 8192 ranks: fewer table builds / unrankings per test (16 -> 32: -9 % kernel time at cfg3)
#endif
// Table path of the size-3 enumeration (accepted sets of up to FZ_TAB_A variables, max_k <= 3).  With
// (z1, z2, z3) = accepted[(i, j, k)], i < j < k, the recursion of statfuns.jl:44-53 needs
//   rho(X,Y|z1,z2)   = l2(A1(i), LX(i,j), LY(i,j))          -- depends on (i, j) only
//   rho(X,z3|z1,z2)  = l2(LX(i,k), LX(i,j), F1(i,j,k))      -- LX(i,v) = rho(X,v|z1), LY(i,v) = rho(Y,v|z1)
//   rho(Y,z3|z1,z2)  = l2(LY(i,k), LY(i,j), F1(i,j,k))      -- F1 = rho(z3,z2|z1)
// so per chunk the workgroup first builds, for every z1-block i the chunk touches, one LDS entry per later position
// v: {LX, LY, cor[v][z1], variable id + Float32 flags, rho(X,Y|z1,v)}.  A test then costs one matrix gather, one
// level-1, two level-2 and one level-3 evaluation instead of 3 + 2 + 1 evaluations and 4 gathers, and -- more
// importantly -- no lane ever recomputes a prefix while the other 63 wait (the divergence of the in-lane caching
// path).  A chunk of 256 * FW_RUN_MAX = 8192 ranks touches at most 1021 entries for every |accepted| <= 512
// (profiles/tools/tab_bound.py 512 8192).
#define FZ_TAB_A FW_TAB_A
#define FZ_TAB_CAP 1024
#define FZ_TAB_ZMASK 0x1FFFFFFF
// entries of blocks [i0, i): block t holds a - 1 - t entries
__device__ __forceinline__ int fz_tab_off(int i, int i0, int a)
{
    return (i - i0) * (a - 1) - (i * (i - 1) - i0 * (i0 - 1)) / 2;
}
#define FZ_X_NONE 1.0e308    // "no candidate yet"
#define FZ_X_SUB 26.0        // beyond this x = |z|/sqrt2, erfc(x)/2*2 leaves the normal range (ties become possible)
#define FZ_X_SUBKEY 1.0e300  // common x-key of the underflow regime (ordered by exact p there)
#define FZ_X_LAZY (-1.0)     // lane best taken by |r| alone; its x-key is computed at the end of the run

// (xa, pa, ra) strictly better than (xb, pb, rb)?  smaller x-key = larger p; equal keys: larger exact p, then later rank
__device__ __forceinline__ bool fz_key_better(double xa, double pa, unsigned long long ra, double xb, double pb,
                                              unsigned long long rb)
{
    if (xa != xb) return xa < xb;
    if (xa == FZ_X_NONE) return false;
    if (pa != pb) return pa > pb;
    return ra > rb;
}

// |r| thresholds for `p < alpha`: bisection on the exact device p-value, then a +-1e-9 relative guard band.
// thr = {lo_pos, hi_pos, lo_neg, hi_neg}: |r| > hi -> significant for sure, |r| < lo -> not significant for sure.
__device__ void fz_thresholds_dev(double alpha, double zscale, double *thr)
{
    for (int sgn = 0; sgn < 2; ++sgn) {
        double lo = 0.0, hi = 1.0;  // p(lo) >= alpha (not sig), p(hi) < alpha (sig) unless nothing is ever significant
        const double sg = sgn ? -1.0 : 1.0;
        if (!(fz_pval_dev(sg * 1.0, zscale) < alpha)) {
            thr[2 * sgn] = 2.0;
            thr[2 * sgn + 1] = 2.0;
            continue;
        }
        for (int it = 0; it < 200; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (fz_pval_dev(sg * mid, zscale) < alpha)
                hi = mid;
            else
                lo = mid;
        }
        thr[2 * sgn] = lo * (1.0 - 1e-9);
        thr[2 * sgn + 1] = hi * (1.0 + 1e-9);
    }
}

__global__ void fz_thresholds_kernel(double alpha, double zscale, double *thr)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    fz_thresholds_dev(alpha, zscale, thr);
    // thr[4]: |r| below which x = |z|/sqrt2 < FZ_X_SUB for sure (see fz_seg_body); once here instead of once per thread
    thr[4] = zscale > 0.0 ? tanh(FZ_X_SUB * 0.7071067811865476 / zscale) * (1.0 - 1e-9) : 2.0;
}

// HIGHK: subsets of size 4-5 possible; LOCAL: per-job matrices (fz_nz); TAB: size-3 subsets through the LDS table
// (the host routes only segments of jobs with |accepted| <= FZ_TAB_A to a TAB launch; never together with HIGHK)
template <bool HIGHK, bool LOCAL, bool TAB>
__device__ __forceinline__ void fz_seg_body(const float *__restrict__ cor_g, int p_g, const FwSeg *__restrict__ segs,
                                            const int32_t *__restrict__ accflat, FwSegOut *__restrict__ out, int max_k,
                                            double alpha, double zscale_g, long long max_tests,
                                            const double *__restrict__ thr_g, const FwNzJob *__restrict__ recs,
                                            long long n_obs_min, const unsigned sidx /* segment this workgroup evaluates */)
{
    __shared__ int s_acc[TAB ? FZ_TAB_A : FW_ACC_LDS];  // TAB: |accepted| <= FZ_TAB_A by the host's routing
    __shared__ unsigned long long s_stop[4];
    __shared__ double s_bx[4], s_bps[4];
    __shared__ unsigned long long s_br[4];
    __shared__ unsigned int s_evc[4];  // tests really executed by each wavefront in the current chunk
    __shared__ double s_best_x, s_best_ps, s_best_stat;
    __shared__ unsigned long long s_best_rank;
    __shared__ float4 s_tab[TAB ? FZ_TAB_CAP : 1];    // {LX, LY, cor[v][z1], variable id | Float32 flags}
    __shared__ float s_tab_r1[TAB ? FZ_TAB_CAP : 1];   // sqrt(1 - cor[v][z1]^2)                       (Float32 roots)
    __shared__ float2 s_tab_r2[TAB ? FZ_TAB_CAP : 1];  // {sqrt(1 - LX^2), sqrt(1 - LY^2)}
    __shared__ double s_tab_a2[TAB ? FZ_TAB_CAP : 1]; // rho(X, Y | z1, v)
    __shared__ int s_blk[2];

    const FwSeg seg = segs[sidx];
    const int a = seg.acc_len;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t *gacc = accflat + seg.acc_off;
    const bool in_lds = a <= (TAB ? FZ_TAB_A : FW_ACC_LDS);
    const float *cor = cor_g;
    int p = p_g;
    double zscale = zscale_g;
    const double *thr = thr_g;
    if (LOCAL) {
        const FwNzJob *rec = recs + seg.pad;
        cor = cor_g + rec->cor_off;
        p = rec->m;
        zscale = rec->zscale;
        thr = rec->thr;
        if ((long long)rec->nR < n_obs_min) {  // tests.jl:294-296: (0, 1, 0, false) with zero tests
            if (tid == 0) {
                FwSegOut o;
                o.stop_rank = 0;
                o.stop_stat = 0.0;
                o.stop_pval = 1.0;
                o.best_rank = 0;
                o.best_stat = 0.0;
                o.best_pval = -1.0;
                o.stop_df = -2;  // marker: no test was executed
                o.stop_power = 0;
                o.best_df = 0;
                o.pad = 0;
                o.evaluated = 0;
                out[sidx] = o;
            }
            return;
        }
    } else if (in_lds) {
        for (int i = tid; i < a; i += 256) s_acc[i] = gacc[i];
    }
    if (tid == 0) {
        s_best_x = FZ_X_NONE;
        s_best_ps = 0.0;
        s_best_stat = 0.0;
        s_best_rank = 0;
    }
    unsigned long long cnt[FW_MAX_K + 1];
#pragma unroll
    for (int s = FW_MAX_K; s >= 1; --s) cnt[s] = (s <= max_k) ? binom_u64(a, s) : 0ull;
    // significance thresholds on |r| (see fz_thresholds_kernel): outside [lo, hi] the verdict of p < alpha is certain
    const double rlo_pos = thr[0], rhi_pos = thr[1], rlo_neg = thr[2], rhi_neg = thr[3];
    // |r| below which x = |z|/sqrt2 < FZ_X_SUB for sure: x = zscale * log((1+r)/(1-r)) / sqrt2  <=>  r = tanh(x / (sqrt2 zscale))
    const double rsub_lo = !LOCAL ? thr[4] : (zscale > 0.0 ? tanh(FZ_X_SUB * 0.7071067811865476 / zscale) * (1.0 - 1e-9) : 2.0);
    __syncthreads();
#define ACCV(i) (LOCAL ? ((i) + 2) : (in_lds ? s_acc[(i)] : gacc[(i)]))
#define CORV(u, v) cor[(size_t)(u) * p + (v)]

    const int X = LOCAL ? 0 : seg.X, Y = LOCAL ? 1 : seg.Y;
    const float cXY = CORV(X, Y);
    const unsigned long long NONE = FW_RANK_NONE;
    const unsigned long long len = seg.end - seg.start;
    const int R = (int)((len + 255) / 256 < FW_RUN_MAX ? (len + 255) / 256 : FW_RUN_MAX);
    unsigned long long evaluated = 0;

    for (unsigned long long cbase = seg.start; cbase < seg.end; cbase += 256ull * R) {
        const unsigned long long r0 = cbase + (unsigned long long)tid * R;
        unsigned long long r1 = r0 + R;
        if (r1 > seg.end) r1 = seg.end;
        const bool any = r0 < seg.end;
        // ---- table of the z1-blocks this chunk touches (see FZ_TAB_A) ----
        bool tab_ok = false;
        int tb_i0 = 0;
        if (TAB) {
            const unsigned long long c3 = (max_k >= 3) ? cnt[3] : 0ull;
            if (cbase < c3) {  // workgroup-uniform; a <= FZ_TAB_A by the host's routing
                unsigned long long last3 = cbase + 256ull * R;
                last3 = last3 < seg.end ? last3 : seg.end;
                last3 = (last3 < c3 ? last3 : c3) - 1ull;
                if (tid == 0 || tid == 64) {
                    int q[FW_MAX_K];
                    unrank_comb(tid == 0 ? cbase : last3, a, 3, q);
                    s_blk[tid == 0 ? 0 : 1] = q[0];
                }
                __syncthreads();
                const int i0 = s_blk[0], i1 = s_blk[1];
                const int E = fz_tab_off(i1 + 1, i0, a);  // <= 1021 for a <= 512 and chunks of 8192 ranks
                if (E > FZ_TAB_CAP) __builtin_trap();      // would be a routing bug on the host side: fail loudly
                if (E <= FZ_TAB_CAP) {
                    tab_ok = true;
                    tb_i0 = i0;
                    for (int e = tid; e < E; e += 256) {
                        int i = i0, rem = e;
                        while (rem >= a - 1 - i) {
                            rem -= a - 1 - i;
                            ++i;
                        }
                        const int z1 = ACCV(i), zv = ACCV(i + 1 + rem);
                        const float cXz1 = CORV(X, z1), cYz1 = CORV(Y, z1), cvz1 = CORV(zv, z1);
                        const TV A1 = pc_l1(cXY, cXz1, cYz1);
                        const TV LX = pc_l1(CORV(X, zv), cXz1, cvz1);
                        const TV LY = pc_l1(CORV(Y, zv), cYz1, cvz1);
                        s_tab_a2[e] = pc_l2(A1, LX, LY);
                        // square roots the level-1 / level-2 formulas take of this entry's values (statfuns.jl:36,52);
                        // for a Float64-literal LX / LY (0, +-1) the Float32 root is the exact one as well
                        const float fx = (float)LX.v, fy = (float)LY.v;
                        s_tab_r1[e] = sqrtf(1.0f - cvz1 * cvz1);
                        s_tab_r2[e] = make_float2(sqrtf(1.0f - fx * fx), sqrtf(1.0f - fy * fy));
                        s_tab[e] = make_float4((float)LX.v, (float)LY.v, cvz1,
                                               __int_as_float(zv | (LX.f32 ? (1 << 30) : 0) | (LY.f32 ? (1 << 29) : 0)));
                    }
                }
                __syncthreads();
            }
        }
        // lane-local results
        unsigned long long my_stop = NONE, my_br = 0;
        double stop_stat = 0.0, stop_p = 0.0, my_bstat = 0.0;
        // lane best: ordered by x = |z|/sqrt2 ascending (= p descending); in the underflow regime (x > FZ_X_SUB, where
        // different x can give the same subnormal/zero p) by the exact p instead; later rank wins ties (tests.jl:338)
        double my_bx = FZ_X_NONE, my_bps = 0.0, my_ba = 0.0;
        unsigned int my_done = 0;  // tests this lane executes in this chunk (it leaves its run at its first stop)
        if (any) {
            // unrank the first rank of the run
            unsigned long long rem = r0;
            int s = max_k;
            while (s > 1 && rem >= cnt[s]) {
                rem -= cnt[s];
                --s;
            }
            int pos[FW_MAX_K];
#pragma unroll
            for (int q = 0; q < FW_MAX_K; ++q) pos[q] = 0;
            unrank_comb(rem, a, s, pos);
            int chg = 0;  // lowest position index that changed since the previous test of this lane (0 = everything)
            // cached state for s <= 3
            int z1 = 0, z2 = 0;
            float cXz1 = 0.f, cYz1 = 0.f, cXz2 = 0.f, cYz2 = 0.f, cz2z1 = 0.f;
            TV A1{0.0, false}, B1{0.0, false}, C1{0.0, false};
            double A2 = 0.0;
            int boff = 0;
            for (unsigned long long r = r0; r < r1; ++r) {
                double stat;
                ++my_done;
                if (TAB && s == 3 && tab_ok) {
                    const int pi = pos[0];
                    if (chg <= 0) boff = fz_tab_off(pi, tb_i0, a) - pi - 1;
                    const int ej = boff + pos[1], ek = boff + pos[2];
                    const float4 tj = s_tab[ej], tk = s_tab[ek];
                    const float rj1 = s_tab_r1[ej], rk1 = s_tab_r1[ek];
                    const float2 rj2 = s_tab_r2[ej];
                    const double A2j = s_tab_a2[ej];
                    const int fj = __float_as_int(tj.w), fk = __float_as_int(tk.w);
                    const float c32 = CORV(fk & FZ_TAB_ZMASK, fj & FZ_TAB_ZMASK);
                    const TV F1 = pc_l1_r(c32, tk.z, tj.z, rk1, rj1);
                    const double dF = fz_sqrt_unit(1.0 - F1.v * F1.v);  // shared by the two level-2 values below
                    double D2, E2;
                    if (__all((((fj & fk) >> 29) & 3) == 3 && F1.f32)) {  // wave-uniform fast path: no Float64 literal
                        D2 = pc_l2_all32_d1(tk.x, tj.x, (float)F1.v, (double)rj2.x, dF);
                        E2 = pc_l2_all32_d1(tk.y, tj.y, (float)F1.v, (double)rj2.y, dF);
                    } else {
                        const TV D1{(double)tk.x, ((fk >> 30) & 1) != 0}, Bj{(double)tj.x, ((fj >> 30) & 1) != 0};
                        const TV E1{(double)tk.y, ((fk >> 29) & 1) != 0}, Cj{(double)tj.y, ((fj >> 29) & 1) != 0};
                        D2 = pc_l2_d2(D1, Bj, F1, dF);
                        E2 = pc_l2_d2(E1, Cj, F1, dF);
                    }
                    stat = pc_l3(A2j, D2, E2);
                } else if (!TAB && s == 3) {
                    if (chg <= 0) {
                        z1 = ACCV(pos[0]);
                        cXz1 = CORV(X, z1);
                        cYz1 = CORV(Y, z1);
                        A1 = pc_l1(cXY, cXz1, cYz1);
                    }
                    if (chg <= 1) {
                        z2 = ACCV(pos[1]);
                        cXz2 = CORV(X, z2);
                        cYz2 = CORV(Y, z2);
                        cz2z1 = CORV(z2, z1);
                        B1 = pc_l1(cXz2, cXz1, cz2z1);
                        C1 = pc_l1(cYz2, cYz1, cz2z1);
                        A2 = pc_l2(A1, B1, C1);
                    }
                    const int z3 = ACCV(pos[2]);
                    const float cXz3 = CORV(X, z3), cYz3 = CORV(Y, z3), cz3z1 = CORV(z3, z1), cz3z2 = CORV(z3, z2);
                    const TV D1 = pc_l1(cXz3, cXz1, cz3z1);
                    const TV E1 = pc_l1(cYz3, cYz1, cz3z1);
                    const TV F1 = pc_l1(cz3z2, cz3z1, cz2z1);
                    const double dF = fz_sqrt_unit(1.0 - F1.v * F1.v);  // shared by the two level-2 values below
                    double D2, E2;
                    if (__all(D1.f32 && E1.f32 && F1.f32 && B1.f32 && C1.f32)) {  // wave-uniform fast path
                        D2 = pc_l2_all32((float)D1.v, (float)B1.v, (float)F1.v, dF);
                        E2 = pc_l2_all32((float)E1.v, (float)C1.v, (float)F1.v, dF);
                    } else {
                        D2 = pc_l2_d2(D1, B1, F1, dF);
                        E2 = pc_l2_d2(E1, C1, F1, dF);
                    }
                    stat = pc_l3(A2, D2, E2);
                } else if (s == 2) {
                    if (chg <= 0) {
                        z1 = ACCV(pos[0]);
                        cXz1 = CORV(X, z1);
                        cYz1 = CORV(Y, z1);
                        A1 = pc_l1(cXY, cXz1, cYz1);
                    }
                    z2 = ACCV(pos[1]);
                    cXz2 = CORV(X, z2);
                    cYz2 = CORV(Y, z2);
                    cz2z1 = CORV(z2, z1);
                    B1 = pc_l1(cXz2, cXz1, cz2z1);
                    C1 = pc_l1(cYz2, cYz1, cz2z1);
                    stat = pc_l2(A1, B1, C1);
                } else if (s == 1) {
                    z1 = ACCV(pos[0]);
                    stat = pc_l1(cXY, CORV(X, z1), CORV(Y, z1)).v;
                } else if (HIGHK) {
                    int zs[FW_MAX_K];
#pragma unroll
                    for (int q = 0; q < FW_MAX_K; ++q) zs[q] = (q < s) ? ACCV(pos[q]) : 0;
                    stat = fz_pcor_any(cor, p, X, Y, zs, s);
                } else {
                    stat = 0.0;
                }
                const double av = fabs(stat);
                const bool negr = stat < 0.0;
                bool sig;
                if (av > (negr ? rhi_neg : rhi_pos))
                    sig = true;
                else if (av < (negr ? rlo_neg : rlo_pos))
                    sig = false;
                else
                    sig = fz_pval_slow(stat, zscale) < alpha;  // inside the guard band (or NaN): exact
                if (!sig || (max_tests > 0 && r + 1 >= (unsigned long long)max_tests)) {
                    my_stop = r;
                    stop_stat = stat;
                    stop_p = fz_pval_slow(stat, zscale);
                    break;
                }
                // tests.jl:338 `pval >= lowest.pval`, sequential within the run, without evaluating p (or even z) for
                // every test.  In the normal regime (|r| < rsub_lo, i.e. x < FZ_X_SUB) p is strictly decreasing in |r|
                // once two values differ by more than rounding noise, so a clearly smaller |r| replaces the lane best
                // "lazily" (x is computed once per run, below), a clearly larger one is skipped, and only near-ties and
                // the underflow regime take the exact path.
                bool exact = false, lazy_take = false;
                if (av < rsub_lo) {
                    if (my_bx == FZ_X_NONE || my_bx > FZ_X_SUB)
                        lazy_take = true;  // nothing yet, or the best so far sits in the underflow regime (smaller p)
                    else if (av < my_ba * (1.0 - 1e-12))
                        lazy_take = true;
                    else
                        exact = av <= my_ba * (1.0 + 1e-12);
                } else {
                    exact = true;
                }
                if (lazy_take) {
                    my_bx = FZ_X_LAZY;
                    my_bps = 0.0;
                    my_ba = av;
                    my_br = r;
                    my_bstat = stat;
                }
                if (exact) {
                    if (my_bx == FZ_X_LAZY)
                        my_bx = fz_xkey_slow(my_bstat, zscale);
                    const double xz = fz_xkey_slow(stat, zscale);
                    bool take;
                    double ps = 0.0;
                    if (xz > FZ_X_SUB) {
                        ps = fz_pval_slow(stat, zscale);  // exact (possibly subnormal / zero) p
                        take = (my_bx == FZ_X_NONE) || (my_bx > FZ_X_SUB && ps >= my_bps);
                    } else {
                        take = (my_bx == FZ_X_NONE) || (my_bx > FZ_X_SUB) || (xz <= my_bx);
                    }
                    if (take) {
                        my_bx = xz;
                        my_bps = ps;
                        my_ba = av;
                        my_br = r;
                        my_bstat = stat;
                    }
                }
                // next combination in lexicographic order (sizes descend when one is exhausted)
                if (s == 3 && pos[2] < a - 1) {  // by far the most frequent step, with static register indices (the
                    ++pos[2];                     // generic code below indexes pos[] dynamically: ~60 instructions)
                    chg = 2;
                    continue;
                }
                int i = s - 1;
                while (i >= 0 && pos[i] == a - s + i) --i;
                if (i < 0) {
                    --s;
#pragma unroll
                    for (int q = 0; q < FW_MAX_K; ++q) pos[q] = q;
                    chg = 0;
                    if (s < 1) break;  // end of the enumeration (r1 never exceeds it)
                } else {
                    ++pos[i];
                    for (int j = i + 1; j < s; ++j) pos[j] = pos[j - 1] + 1;
                    chg = i;
                }
            }
        }
        if (my_bx == FZ_X_LAZY)  // resolve the lazily kept lane best: its x-key
            my_bx = fz_xkey_slow(my_bstat, zscale);
        // first stopping rank in the workgroup
        unsigned long long ws = my_stop;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long t = __shfl_xor(ws, o);
            ws = t < ws ? t : ws;
        }
        // key of the lane best: (xk, ps, rank); xk = x in the normal regime, FZ_X_SUBKEY in the underflow regime
        double bx = (my_bx == FZ_X_NONE) ? FZ_X_NONE : (my_bx > FZ_X_SUB ? FZ_X_SUBKEY : my_bx);
        double bps = my_bps;
        unsigned long long br = my_br;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ox = __shfl_xor(bx, o), ops = __shfl_xor(bps, o);
            const unsigned long long orr = __shfl_xor(br, o);
            if (fz_key_better(ox, ops, orr, bx, bps, br)) {
                bx = ox;
                bps = ops;
                br = orr;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) my_done += __shfl_xor(my_done, o);
        if (lane == 0) {
            s_evc[wave] = my_done;
            s_stop[wave] = ws;
            s_bx[wave] = bx;
            s_bps[wave] = bps;
            s_br[wave] = br;
        }
        __syncthreads();
        unsigned long long first = s_stop[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) first = s_stop[w] < first ? s_stop[w] : first;
        evaluated += (unsigned long long)(s_evc[0] + s_evc[1] + s_evc[2] + s_evc[3]);  // executed tests, not chunk sizes
        if (first != NONE) {
            if (my_stop == first) {
                FwSegOut o;
                o.stop_rank = first;
                o.stop_stat = stop_stat;
                o.stop_pval = stop_p;
                o.best_rank = 0;
                o.best_stat = 0.0;
                o.best_pval = -1.0;
                o.stop_df = 0;
                o.stop_power = 1;
                o.best_df = 0;
                o.pad = 0;
                o.evaluated = evaluated;
                out[sidx] = o;
            }
            return;
        }
        double cbx = s_bx[0], cbps = s_bps[0];
        unsigned long long cbr = s_br[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (fz_key_better(s_bx[w], s_bps[w], s_br[w], cbx, cbps, cbr)) {
                cbx = s_bx[w];
                cbps = s_bps[w];
                cbr = s_br[w];
            }
        // chunks hold increasing ranks: the newer chunk wins ties against the running best of the segment
        if (my_bx != FZ_X_NONE && my_br == cbr && cbx != FZ_X_NONE &&
            (s_best_x == FZ_X_NONE || !fz_key_better(s_best_x, s_best_ps, 0ull, cbx, cbps, 1ull))) {
            s_best_x = cbx;
            s_best_ps = cbps;
            s_best_stat = my_bstat;
            s_best_rank = my_br;
        }
        __syncthreads();
    }
#undef ACCV
#undef CORV
    if (tid == 0) {
        FwSegOut o;
        o.stop_rank = NONE;
        o.stop_stat = 0.0;
        o.stop_pval = 0.0;
        o.best_rank = s_best_rank;
        o.best_stat = s_best_stat;
        o.best_pval = (s_best_x == FZ_X_NONE) ? -1.0 : fz_pval_slow(s_best_stat, zscale);
        o.stop_df = 0;
        o.stop_power = 1;
        o.best_df = 0;
        o.pad = 0;
        o.evaluated = evaluated;
        out[sidx] = o;
    }
}

// Host-driven rounds: one workgroup per segment (ns_dev == nullptr).  Device-driven rounds (fw_devhiton.hip): a fixed
// grid strides over an unsorted segment list whose live length sits in device memory; the table / in-lane variants
// each pick their own segments.
template <bool HIGHK, bool LOCAL, bool TAB>
__global__ __launch_bounds__(256, 4) void fz_subsets_seg_kernel(const float *__restrict__ cor_g, int p_g,
                                                             const FwSeg *__restrict__ segs,
                                                             const int32_t *__restrict__ accflat,
                                                             FwSegOut *__restrict__ out, int max_k, double alpha,
                                                             double zscale_g, long long max_tests,
                                                             const double *__restrict__ thr_g,
                                                             const FwNzJob *__restrict__ recs, long long n_obs_min,
                                                             const unsigned *__restrict__ ns_dev,
                                                             const unsigned *__restrict__ big_dev)
{
    // device rounds launch the in-lane variant next to the table variant whenever a long accepted list is POSSIBLE (with
    // whitelists that is nearly always); the fill kernel knows whether one EXISTS in this launch -- without one, leave
    // before walking the segment list (r02 profile, feed-forward rounds: 26 us per launch for nothing, 40 ms per pass)
    if (big_dev && !TAB && !HIGHK && *big_dev == 0u) return;
    // one instance of the body for both modes: host-driven = exactly one iteration, every segment of the launch is ours
    const unsigned ns = ns_dev ? *ns_dev : gridDim.x;
    for (unsigned s = blockIdx.x; s < ns; s += gridDim.x) {
        if (ns_dev && !HIGHK && ((segs[s].acc_len <= FZ_TAB_A) != TAB)) continue;  // workgroup-uniform
        fz_seg_body<HIGHK, LOCAL, TAB>(cor_g, p_g, segs, accflat, out, max_k, alpha, zscale_g, max_tests, thr_g, recs, n_obs_min, s);
        __syncthreads();  // the LDS state of the body is reused by the next segment
    }
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
static int fz_ensure_thresholds(fw_ctx *ctx, hipStream_t stream);

static double fz_zscale(const fw_ctx *ctx)
{
    const long long sf = (long long)ctx->P.n - 3;  // len_z = 0 always (tests.jl:156,256)
    return sf > 0 ? std::sqrt((double)sf) / 2.0 : 0.0;
}

int fwi_fz_compute_cor(fw_ctx *ctx)
{
    if (!ctx->have_data) return fw_fail(ctx, FW_ERR_STATE, "fw_compute_cor_mat: no data uploaded (fw_set_data_dense_f32)");
    const int n = ctx->P.n, p = ctx->P.p;
    ctx->n_pad = (n + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
    ctx->p_pad = (p + GEMM_BM - 1) / GEMM_BM * GEMM_BM;
    if (!ctx->d_xc) FW_HIP(ctx, hipMalloc(&ctx->d_xc, sizeof(float) * (size_t)ctx->n_pad * ctx->p_pad));
    if (!ctx->d_sd) FW_HIP(ctx, hipMalloc(&ctx->d_sd, sizeof(float) * (size_t)ctx->p_pad));
    if (!ctx->d_cor) FW_HIP(ctx, hipMalloc(&ctx->d_cor, sizeof(float) * (size_t)p * p));
    hipLaunchKernelGGL(fz_center_kernel, dim3(ctx->p_pad), dim3(256), 0, ctx->stream, ctx->d_data, ctx->d_xc, ctx->d_sd, n,
                       p, ctx->n_pad);
    const int T = ctx->p_pad / GEMM_BM;
    const int nblk = T * (T + 1) / 2;
    hipLaunchKernelGGL(fz_cor_gemm_kernel, dim3(nblk), dim3(256), 0, ctx->stream, ctx->d_xc, ctx->d_sd, ctx->d_cor, p,
                       ctx->n_pad, T);
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cnt.kernel_launches += 2;
    ctx->have_cor = true;
    return FW_OK;
}

int fwi_fz_level0(fw_ctx *ctx, std::vector<int32_t> &pi, std::vector<int32_t> &pj, std::vector<double> &stat,
                  std::vector<double> &pval, int64_t *m_reliable, FwL0Dev *dev)
{
    if (dev) *dev = FwL0Dev{};
    const int p = ctx->P.p;
    const long long npairs = (long long)p * (p - 1) / 2;
    if (ctx->P.n < ctx->n_obs_min_eff) {  // tests.jl:11 -> every test lacks power -> all NaN
        pi.clear();
        pj.clear();
        stat.clear();
        pval.clear();
        *m_reliable = 0;
        return FW_OK;
    }
    {
        int rc0 = fz_ensure_thresholds(ctx, ctx->stream);
        if (rc0) return rc0;
    }
    unsigned long long cap = (unsigned long long)std::min<long long>(npairs, 4ll << 20);
    if (cap < ctx->l0_cap_hint) cap = ctx->l0_cap_hint;  // a repeated call does not overflow (and re-run the kernel) again
    if (cap == 0) cap = 1;
    FzL0Counters h{};
    for (int attempt = 0; attempt < 2; ++attempt) {
        int rc;
        if ((rc = fw_dev_reserve(ctx, ctx->d_tmp0, 2 * sizeof(FzL0Counters)))) return rc;
        if ((rc = fw_dev_reserve(ctx, ctx->d_tmp1, cap * (2 * sizeof(int32_t) + sizeof(float))))) return rc;
        if ((rc = fw_dev_reserve(ctx, ctx->d_tmp2, cap * sizeof(double)))) return rc;
        if ((rc = fw_dev_reserve(ctx, ctx->d_jobs, cap * (2 * sizeof(int32_t) + sizeof(float))))) return rc;
        FW_HIP(ctx, hipMemsetAsync(ctx->d_tmp0.ptr, 0, 2 * sizeof(FzL0Counters), ctx->stream));
        FzL0Counters *d_c1 = (FzL0Counters *)ctx->d_tmp0.ptr, *d_c2 = d_c1 + 1;
        int32_t *oi = (int32_t *)ctx->d_tmp1.ptr;
        int32_t *oj = oi + cap;
        float *orr = (float *)(oj + cap);
        double *op = (double *)ctx->d_tmp2.ptr;
        int32_t *ci = (int32_t *)ctx->d_jobs.ptr;
        int32_t *cj = ci + cap;
        float *cr = (float *)(cj + cap);
        dim3 grid((p + FZ_L0_COLS - 1) / FZ_L0_COLS, (p + FZ_L0_ROWS - 1) / FZ_L0_ROWS);
        hipLaunchKernelGGL(fz_level0_kernel, grid, dim3(256), 0, ctx->stream, ctx->d_cor, p, (const double *)ctx->d_thr, d_c1, cap, ci,
                           cj, cr);
        FW_HIP(ctx, hipGetLastError());
        FzL0Counters h1{};
        FW_HIP(ctx, hipMemcpyAsync(&h1, d_c1, sizeof(h1), hipMemcpyDeviceToHost, ctx->stream));
        FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (h1.n_sig > ctx->l0_cap_hint) ctx->l0_cap_hint = h1.n_sig;
        if (h1.n_sig > cap) {  // more screened pairs than the buffer holds: retry with the exact count
            cap = h1.n_sig;
            continue;
        }
        if (h1.n_sig)
            hipLaunchKernelGGL(fz_level0_exact_kernel, dim3((unsigned)((h1.n_sig + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const int32_t *)ci, (const int32_t *)cj, (const float *)cr, h1.n_sig, ctx->P.alpha, fz_zscale(ctx),
                               d_c2, cap, oi, oj, orr, op);
        FW_HIP(ctx, hipGetLastError());
        FW_HIP(ctx, hipMemcpyAsync(&h, d_c2, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
        h.n_nan = h1.n_nan;
        ctx->cnt.kernel_launches += 2;
        if (h.n_sig <= cap) {
            const size_t k = (size_t)h.n_sig;
            if (dev) {  // results stay on the device for fwi_bh_csr_device
                dev->i = oi;
                dev->j = oj;
                dev->stat32 = orr;
                dev->pval = op;
                dev->k = k;
                *m_reliable = npairs - (long long)h.n_nan;
                return FW_OK;
            }
            pi.resize(k);
            pj.resize(k);
            stat.resize(k);
            pval.resize(k);
            std::vector<float> rr(k);
            if (k) {
                FW_HIP(ctx, hipMemcpy(pi.data(), oi, k * sizeof(int32_t), hipMemcpyDeviceToHost));
                FW_HIP(ctx, hipMemcpy(pj.data(), oj, k * sizeof(int32_t), hipMemcpyDeviceToHost));
                FW_HIP(ctx, hipMemcpy(rr.data(), orr, k * sizeof(float), hipMemcpyDeviceToHost));
                FW_HIP(ctx, hipMemcpy(pval.data(), op, k * sizeof(double), hipMemcpyDeviceToHost));
            }
            for (size_t t = 0; t < k; ++t) stat[t] = (double)rr[t];
            *m_reliable = npairs - (long long)h.n_nan;
            return FW_OK;
        }
        cap = h.n_sig;
    }
    return fw_fail(ctx, FW_ERR_DEVICE, "fz level-0: compaction buffer overflow twice");
}

int fwi_fz_test_batch(fw_ctx *ctx, int64_t m, const int32_t *X, const int32_t *Y, const int64_t *zoff,
                      const int32_t *zflat, fw_test_result *out)
{
    if (m == 0) return FW_OK;
    const int64_t nz = zoff[m];
    int rc;
    if ((rc = fw_dev_reserve(ctx, ctx->d_jobs, (size_t)m * 2 * sizeof(int32_t) + (size_t)(m + 1) * sizeof(int64_t)))) return rc;
    if ((rc = fw_dev_reserve(ctx, ctx->d_acc, (size_t)(nz > 0 ? nz : 1) * sizeof(int32_t)))) return rc;
    if ((rc = fw_dev_reserve(ctx, ctx->d_out, (size_t)m * sizeof(fw_test_result)))) return rc;
    long long *dz = (long long *)ctx->d_jobs.ptr;
    int32_t *dX = (int32_t *)(dz + m + 1);
    int32_t *dY = dX + m;
    FW_HIP(ctx, hipMemcpyAsync(dz, zoff, (size_t)(m + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    FW_HIP(ctx, hipMemcpyAsync(dX, X, (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    FW_HIP(ctx, hipMemcpyAsync(dY, Y, (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    if (nz > 0)
        FW_HIP(ctx, hipMemcpyAsync(ctx->d_acc.ptr, zflat, (size_t)nz * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(fz_test_batch_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_cor,
                       ctx->P.p, (long long)m, dX, dY, dz, (const int32_t *)ctx->d_acc.ptr, fz_zscale(ctx),
                       (fw_test_result *)ctx->d_out.ptr);
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipMemcpyAsync(out, ctx->d_out.ptr, (size_t)m * sizeof(fw_test_result), hipMemcpyDeviceToHost, ctx->stream));
    FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cnt.kernel_launches += 1;
    const bool power = ctx->P.n >= ctx->n_obs_min_eff;  // tests.jl:254 / :111
    if (!power)
        for (int64_t t = 0; t < m; ++t) {
            out[t].stat = 0.0;
            out[t].pval = 1.0;
            out[t].df = 0;
            out[t].suff_power = 0;
        }
    return FW_OK;
}

static int fz_ensure_thresholds(fw_ctx *ctx, hipStream_t stream)
{
    if (!ctx->d_thr) {
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_thr, 8 * sizeof(double)));
        hipLaunchKernelGGL(fz_thresholds_kernel, dim3(1), dim3(64), 0, stream, ctx->P.alpha, fz_zscale(ctx), ctx->d_thr);
        FW_HIP(ctx, hipGetLastError());
        FW_HIP(ctx, hipStreamSynchronize(stream));  // another stream may use it next
    }
    return FW_OK;
}

// Device-driven rounds (fw_devhiton.hip): a fixed grid over an unsorted segment list whose live length is *d_ns.
int fwi_fz_segments_dev(fw_ctx *ctx, unsigned grid, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out, const unsigned *d_ns,
                        bool any_big, const unsigned *d_big, hipStream_t stream)
{
    int rc = fz_ensure_thresholds(ctx, stream);
    if (rc) return rc;
    const unsigned grid_big = grid < 512u ? grid : 512u;  // accepted sets beyond FZ_TAB_A are rare: few striding workgroups
    if (ctx->P.max_k > 3) {
        hipLaunchKernelGGL((fz_subsets_seg_kernel<true, false, false>), dim3(grid), dim3(256), 0, stream, ctx->d_cor, ctx->P.p, d_segs,
                           d_acc, d_out, ctx->P.max_k, ctx->P.alpha, fz_zscale(ctx), (long long)ctx->P.max_tests, ctx->d_thr,
                           (const FwNzJob *)nullptr, 0ll, d_ns, d_big);
    } else {
        hipLaunchKernelGGL((fz_subsets_seg_kernel<false, false, true>), dim3(grid), dim3(256), 0, stream, ctx->d_cor, ctx->P.p, d_segs,
                           d_acc, d_out, ctx->P.max_k, ctx->P.alpha, fz_zscale(ctx), (long long)ctx->P.max_tests, ctx->d_thr,
                           (const FwNzJob *)nullptr, 0ll, d_ns, d_big);
        if (any_big)  // some accepted set may exceed FZ_TAB_A
            hipLaunchKernelGGL((fz_subsets_seg_kernel<false, false, false>), dim3(grid_big), dim3(256), 0, stream, ctx->d_cor, ctx->P.p,
                               d_segs, d_acc, d_out, ctx->P.max_k, ctx->P.alpha, fz_zscale(ctx), (long long)ctx->P.max_tests,
                               ctx->d_thr, (const FwNzJob *)nullptr, 0ll, d_ns, d_big);
    }
    FW_HIP(ctx, hipGetLastError());
    return FW_OK;
}

int fwi_fz_segments(fw_ctx *ctx, int64_t nseg, int64_t nseg_tab, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out,
                    FwPoolBuf &pb)
{
    if (nseg == 0) return FW_OK;
    FW_HIP(ctx, hipEventRecord(pb.ev0, pb.launch_stream));
    {
        int rc = fz_ensure_thresholds(ctx, pb.launch_stream);
        if (rc) return rc;
    }
    if (ctx->P.max_k > 3)
        hipLaunchKernelGGL((fz_subsets_seg_kernel<true, false, false>), dim3((unsigned)nseg), dim3(256), 0, pb.launch_stream, ctx->d_cor,
                           ctx->P.p, d_segs, d_acc, d_out, ctx->P.max_k, ctx->P.alpha, fz_zscale(ctx),
                           (long long)ctx->P.max_tests, ctx->d_thr, (const FwNzJob *)nullptr, 0ll, (const unsigned *)nullptr, (const unsigned *)nullptr);
    else {
        // segments [0, nseg_tab) belong to jobs with |accepted| <= FZ_TAB_A: table kernel; the rest: in-lane caching
        if (nseg_tab > 0)
            hipLaunchKernelGGL((fz_subsets_seg_kernel<false, false, true>), dim3((unsigned)nseg_tab), dim3(256), 0, pb.launch_stream,
                               ctx->d_cor, ctx->P.p, d_segs, d_acc, d_out, ctx->P.max_k, ctx->P.alpha, fz_zscale(ctx),
                               (long long)ctx->P.max_tests, ctx->d_thr, (const FwNzJob *)nullptr, 0ll, (const unsigned *)nullptr, (const unsigned *)nullptr);
        if (nseg > nseg_tab)
            hipLaunchKernelGGL((fz_subsets_seg_kernel<false, false, false>), dim3((unsigned)(nseg - nseg_tab)), dim3(256), 0,
                               pb.launch_stream, ctx->d_cor, ctx->P.p, d_segs + nseg_tab, d_acc, d_out + nseg_tab, ctx->P.max_k,
                               ctx->P.alpha, fz_zscale(ctx), (long long)ctx->P.max_tests, ctx->d_thr, (const FwNzJob *)nullptr,
                               0ll, (const unsigned *)nullptr, (const unsigned *)nullptr);
    }
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipEventRecord(pb.ev1, pb.launch_stream));
    return FW_OK;
}


// ================================================================================================
// HE-S ("fz_nz"): zero-ignoring Fisher-z tests.  Reference: tests.jl:108-160 (branch :120-125), statfuns.jl:91-123
// (pair correlation over the rows where both are non-zero), tests.jl:293-308 + statfuns.jl:138-155 (cor_subset! per
// (T, candidate) job), hiton.jl:41-50,85 (row views).  Data layout: dense Float32 [p][n] column-major exactly as
// uploaded (zeros = absences) + one nz bit plane [p][W].  All sums run sequentially over the rows in Float64 in row
// order -- the same operation sequence as the oracle -- so correlations are reproducible to the bit.
// ================================================================================================

// ---- level 0: one thread per pair (16 x 16 pair tiles), two passes over the samples ----
__global__ __launch_bounds__(256) void fznz_level0_kernel(const float *__restrict__ data, const unsigned long long *__restrict__ nz,
                                                          int n, int p, int W, double alpha, long long n_obs_min,
                                                          FzL0Counters *cnt, unsigned long long cap, int32_t *out_i,
                                                          int32_t *out_j, double *out_s, double *out_p)
{
    const int X = blockIdx.y * 16 + (threadIdx.x >> 4), Y = blockIdx.x * 16 + (threadIdx.x & 15);
    if (blockIdx.x * 16 + 15 <= blockIdx.y * 16) return;  // tile entirely on/below the diagonal
    const bool valid = X < Y && Y < p;
    const float *cx = data + (size_t)(valid ? X : 0) * n, *cy = data + (size_t)(valid ? Y : 0) * n;
    const unsigned long long *mx = nz + (size_t)(valid ? X : 0) * W, *my = nz + (size_t)(valid ? Y : 0) * W;
    double sum_x = 0.0, sum_y = 0.0;
    long long nn = 0;
    for (int w = 0; w < W; ++w) {
        unsigned long long m = valid ? (mx[w] & my[w]) : 0ull;
        nn += __popcll(m);
        while (m) {
            const int row = w * 64 + __builtin_ctzll(m);
            m &= m - 1;
            sum_x += (double)cx[row];
            sum_y += (double)cy[row];
        }
    }
    double stat = 0.0, pval = 1.0;
    bool reliable = false;
    if (valid && (long long)n >= n_obs_min) {
        double pc = 0.0;
        if (nn > 0) {
            const double mean_x = sum_x / (double)nn, mean_y = sum_y / (double)nn;
            double cov = 0.0, vx = 0.0, vy = 0.0;
            for (int w = 0; w < W; ++w) {
                unsigned long long m = mx[w] & my[w];
                while (m) {
                    const int row = w * 64 + __builtin_ctzll(m);
                    m &= m - 1;
                    const double dx = (double)cx[row] - mean_x, dy = (double)cy[row] - mean_y;
                    cov += dx * dy;
                    vx += dx * dx;
                    vy += dy * dy;
                }
            }
            pc = cov / sqrt(vx * vy);
            if (pc > 1.0)
                pc = 1.0;
            else if (pc < -1.0)
                pc = -1.0;
        }
        if (nn < n_obs_min) pc = 0.0;  // tests.jl:123-125
        const long long sf = nn - 3;
        stat = pc;
        pval = fz_pval_dev(pc, sf > 0 ? sqrt((double)sf) / 2.0 : 0.0);
        reliable = nn >= n_obs_min;
    }
    const bool isn = valid && (!reliable || isnan(pval));  // NaN in the reference's condensed arrays (tests.jl:397-398)
    const bool sig = valid && reliable && pval < alpha;
    // one atomic per wavefront for each counter (per-pair atomics on one address serialise the kernel)
    const int lane = threadIdx.x & 63;
    const unsigned long long mn = __ballot(isn), ms = __ballot(sig);
    if (mn) {
        const int leader = __ffsll((long long)mn) - 1;
        if (lane == leader) atomicAdd(&cnt->n_nan, (unsigned long long)__popcll(mn));
    }
    if (ms) {
        const int leader = __ffsll((long long)ms) - 1;
        unsigned long long base = 0;
        if (lane == leader) base = atomicAdd(&cnt->n_sig, (unsigned long long)__popcll(ms));
        base = __shfl(base, leader);
        if (sig) {
            const unsigned long long slot = base + (unsigned long long)__popcll(ms & ((1ull << lane) - 1ull));
            if (slot < cap) {
                out_i[slot] = X;
                out_j[slot] = Y;
                out_s[slot] = stat;
                out_p[slot] = pval;
            }
        }
    }
}

// ---- per-job correlation sub-matrix (Statistics.cor of the row view restricted to {T, cand} + accepted) ----
// Summation order ("tree64", mirrored by oracle/fw_oracle.c fz_nz_tree64): the rows R of the view are numbered q = 0, 1, ...
// in ascending row order; lane l of a wavefront adds the terms q = l, l + 64, ... sequentially in Float64, and the 64
// partials are combined in a fixed tree: pairs (l, l^1), then (l, l^2), then inside every group of 16 lanes
// (Q3 + Q2) + (Q1 + Q0), then (R3 + R2) + (R1 + R0) over the four groups -- the DPP network's natural order.  One
// wavefront per sum (a column mean, a column norm, a pair's dot product), the four wavefronts of the workgroup take
// the sums of a job in turn.  r01 walked the rows sequentially in one lane per pair: serial by construction, 113 of the
// 125 ms of a pass at 3 000 variables.
#define FZNZ_MAXM_LDS 2050
#define FZNZ_ROWS_LDS 16384  // rows of a view kept as an LDS list (n beyond that: the sequential form)

__device__ __forceinline__ double fznz_tree64(double v)  // every lane returns the total
{
#define FZNZ_DPP_ADDD(ctrl, rmask)                                                                                         \
    {                                                                                                                      \
        const long long b = __double_as_longlong(v);                                                                       \
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, ctrl, rmask, 0xf, false);           \
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), ctrl, rmask, 0xf, false);   \
        v += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));                                       \
    }
    FZNZ_DPP_ADDD(0xb1, 0xf)   // quad_perm:[1,0,3,2]
    FZNZ_DPP_ADDD(0x4e, 0xf)   // quad_perm:[2,3,0,1]
    FZNZ_DPP_ADDD(0x114, 0xf)  // row_shr:4
    FZNZ_DPP_ADDD(0x118, 0xf)  // row_shr:8
    FZNZ_DPP_ADDD(0x142, 0xa)  // row_bcast:15
    FZNZ_DPP_ADDD(0x143, 0xc)  // row_bcast:31
#undef FZNZ_DPP_ADDD
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__global__ __launch_bounds__(256) void fznz_submat_kernel(const float *__restrict__ data, const unsigned long long *__restrict__ nz,
                                                          int n, int W, FwNzJob *__restrict__ recs,
                                                          const int32_t *__restrict__ accflat, float *__restrict__ arena,
                                                          double alpha)
{
    __shared__ double s_mean[FZNZ_MAXM_LDS], s_sd[FZNZ_MAXM_LDS], s_ss01[2];
    __shared__ int s_var[FZNZ_MAXM_LDS];
    __shared__ unsigned short s_rows[FZNZ_ROWS_LDS];
    __shared__ int s_woff[257];
    FwNzJob *rec = recs + blockIdx.x;
    const int m = rec->m, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long *mx = nz + (size_t)rec->X * W, *my = nz + (size_t)rec->Y * W;
    for (int t = tid; t < m; t += 256) s_var[t] = t == 0 ? rec->X : (t == 1 ? rec->Y : accflat[rec->acc_off + t - 2]);
    long long nR = 0;
    for (int w = 0; w < W; ++w) nR += __popcll(mx[w] & my[w]);
    float *local = arena + rec->cor_off;
    const long long npairs = (long long)m * (m - 1) / 2;
    const bool tree = rec->acc_len > 0 && n <= FZNZ_ROWS_LDS && W <= 256;  // wave-uniform (workgroup-uniform)
    if (tree) {
        // row list of the view, ascending: word w of the AND mask expands behind the rows of the words before it
        if (tid < W) s_woff[tid + 1] = __popcll(mx[tid] & my[tid]);
        if (tid == 0) s_woff[0] = 0;
        __syncthreads();
        if (tid == 0)
            for (int w = 0; w < W; ++w) s_woff[w + 1] += s_woff[w];
        __syncthreads();
        if (tid < W) {
            unsigned long long mk = mx[tid] & my[tid];
            int o = s_woff[tid];
            while (mk) {
                s_rows[o++] = (unsigned short)(tid * 64 + __builtin_ctzll(mk));
                mk &= mk - 1;
            }
        }
        __syncthreads();
        const int nr = (int)nR;
        for (int t = wave; t < m; t += 4) {  // column means and norms
            const float *col = data + (size_t)s_var[t] * n;
            double sacc = 0.0;
            for (int q = lane; q < nr; q += 64) sacc += (double)col[s_rows[q]];
            const double mean = fznz_tree64(sacc) / (double)nR;
            double ss = 0.0;
            for (int q = lane; q < nr; q += 64) {
                const double d = (double)col[s_rows[q]] - mean;
                ss += d * d;
            }
            ss = fznz_tree64(ss);
            if (lane == 0) {
                s_mean[t] = mean;
                s_sd[t] = sqrt(ss);
            }
        }
        __syncthreads();
        for (long long q0 = wave; q0 < npairs; q0 += 4) {  // one wavefront per pair
            int a = (int)(((2.0 * m - 1.0) - sqrt((2.0 * m - 1.0) * (2.0 * m - 1.0) - 8.0 * (double)q0)) * 0.5);
            while (a > 0 && (long long)a * (2 * m - a - 1) / 2 > q0) --a;
            while ((long long)(a + 1) * (2 * m - a - 2) / 2 <= q0) ++a;
            const int b = a + 1 + (int)(q0 - (long long)a * (2 * m - a - 1) / 2);
            const float *ca = data + (size_t)s_var[a] * n, *cb = data + (size_t)s_var[b] * n;
            const double ma = s_mean[a], mb = s_mean[b];
            double sacc = 0.0;
            for (int q = lane; q < nr; q += 64) {
                const int row = s_rows[q];
                sacc += ((double)ca[row] - ma) * ((double)cb[row] - mb);
            }
            sacc = fznz_tree64(sacc);
            double r = sacc / (s_sd[a] * s_sd[b]);
            if (r > 1.0) r = 1.0;
            if (r < -1.0) r = -1.0;
            if (isnan(r)) r = 0.0;  // statfuns.jl:150
            const float rf = (float)r;  // the scratch matrix of the reference is Float32 (learning.jl:127-129)
            if (lane == 0) {
                local[(size_t)a * m + b] = rf;
                local[(size_t)b * m + a] = rf;
            }
        }
        for (int t = tid; t < m; t += 256) local[(size_t)t * m + t] = 1.0f;
        if (tid == 0) {
            rec->rxy = 0.0;  // (only univariate jobs read it: they take the sequential form below)
            rec->nR = (int32_t)nR;
            const long long sf = nR - 3;
            rec->zscale = sf > 0 ? sqrt((double)sf) / 2.0 : 0.0;
            fz_thresholds_dev(alpha, rec->zscale, rec->thr);
        }
        return;
    }
    __syncthreads();
    // sequential form (univariate jobs -- their pair statistic must equal level 0's, statfuns.jl:91-123 in row order -- and
    // views beyond the LDS row list): column means and norms over R, sequential Float64 sums in row order
    for (int t = tid; t < m; t += 256) {
        const float *col = data + (size_t)s_var[t] * n;
        double sacc = 0.0;
        for (int w = 0; w < W; ++w) {
            unsigned long long mk = mx[w] & my[w];
            while (mk) {
                const int row = w * 64 + __builtin_ctzll(mk);
                mk &= mk - 1;
                sacc += (double)col[row];
            }
        }
        const double mean = sacc / (double)nR;
        double ss = 0.0;
        for (int w = 0; w < W; ++w) {
            unsigned long long mk = mx[w] & my[w];
            while (mk) {
                const int row = w * 64 + __builtin_ctzll(mk);
                mk &= mk - 1;
                const double d = (double)col[row] - mean;
                ss += d * d;
            }
        }
        s_mean[t] = mean;
        s_sd[t] = sqrt(ss);
        if (t < 2) s_ss01[t] = ss;
    }
    __syncthreads();
    for (long long q = tid; q < npairs; q += 256) {
        // pair index -> (a, b), a < b, row-major over the upper triangle
        int a = (int)(((2.0 * m - 1.0) - sqrt((2.0 * m - 1.0) * (2.0 * m - 1.0) - 8.0 * (double)q)) * 0.5);
        while (a > 0 && (long long)a * (2 * m - a - 1) / 2 > q) --a;
        while ((long long)(a + 1) * (2 * m - a - 2) / 2 <= q) ++a;
        const int b = a + 1 + (int)(q - (long long)a * (2 * m - a - 1) / 2);
        const float *ca = data + (size_t)s_var[a] * n, *cb = data + (size_t)s_var[b] * n;
        const double ma = s_mean[a], mb = s_mean[b];
        double sacc = 0.0;
        for (int w = 0; w < W; ++w) {
            unsigned long long mk = mx[w] & my[w];
            while (mk) {
                const int row = w * 64 + __builtin_ctzll(mk);
                mk &= mk - 1;
                sacc += ((double)ca[row] - ma) * ((double)cb[row] - mb);
            }
        }
        if (q == 0) {  // (a, b) = (0, 1): the pair statistic of statfuns.jl:114-120, unrounded
            double pc = sacc / sqrt(s_ss01[0] * s_ss01[1]);
            if (pc > 1.0)
                pc = 1.0;
            else if (pc < -1.0)
                pc = -1.0;
            rec->rxy = nR > 0 ? pc : 0.0;
        }
        double r = sacc / (s_sd[a] * s_sd[b]);
        if (r > 1.0) r = 1.0;
        if (r < -1.0) r = -1.0;
        if (isnan(r)) r = 0.0;  // statfuns.jl:150
        const float rf = (float)r;  // the scratch matrix of the reference is Float32 (learning.jl:127-129)
        local[(size_t)a * m + b] = rf;
        local[(size_t)b * m + a] = rf;
    }
    for (int t = tid; t < m; t += 256) local[(size_t)t * m + t] = 1.0f;
    if (tid == 0) {
        rec->nR = (int32_t)nR;
        const long long sf = nR - 3;
        rec->zscale = sf > 0 ? sqrt((double)sf) / 2.0 : 0.0;
        fz_thresholds_dev(alpha, rec->zscale, rec->thr);
    }
}

// ---- explicit single tests (fw_test_batch): one thread per test on the job's local matrix ----
__global__ __launch_bounds__(64) void fznz_single_kernel(const FwNzJob *__restrict__ recs, const float *__restrict__ arena,
                                                         long long m_tests, long long n_obs_min,
                                                         fw_test_result *__restrict__ out)
{
    const long long t = (long long)blockIdx.x * 64 + threadIdx.x;
    if (t >= m_tests) return;
    const FwNzJob rec = recs[t];
    fw_test_result o;
    if (rec.acc_len == 0) {  // univariate: tests.jl:120-125,155-159
        const double ps = (long long)rec.nR < n_obs_min ? 0.0 : rec.rxy;
        o.stat = ps;
        o.pval = fz_pval_dev(ps, rec.zscale);
        o.df = 0;
        o.suff_power = (long long)rec.nR >= n_obs_min ? 1 : 0;
    } else if ((long long)rec.nR < n_obs_min) {
        o.stat = 0.0;
        o.pval = 1.0;
        o.df = 0;
        o.suff_power = 0;
    } else {
        int z[FW_MAX_K];
        for (int q = 0; q < FW_MAX_K; ++q) z[q] = 2 + q;
        const double r = fz_pcor_any(arena + rec.cor_off, rec.m, 0, 1, z, rec.acc_len);
        o.stat = r;
        o.pval = fz_pval_dev(r, rec.zscale);
        o.df = 0;
        o.suff_power = 1;
    }
    out[t] = o;
}

int fwi_fznz_upload(fw_ctx *ctx, const float *data)
{
    const int n = ctx->P.n, p = ctx->P.p, W = (n + 63) / 64;
    const size_t bytes = sizeof(float) * (size_t)n * p;
    if (!ctx->d_data) FW_HIP(ctx, hipMalloc(&ctx->d_data, bytes));
    FW_HIP(ctx, hipMemcpy(ctx->d_data, data, bytes, hipMemcpyHostToDevice));
    std::vector<uint64_t> nzb((size_t)p * W, 0);
    for (int v = 0; v < p; ++v)
        for (int i = 0; i < n; ++i)
            if (data[(size_t)v * n + i] != 0.0f) nzb[(size_t)v * W + (i >> 6)] |= 1ull << (i & 63);
    if (ctx->d_nzbits) (void)hipFree(ctx->d_nzbits);
    ctx->d_nzbits = nullptr;
    FW_HIP(ctx, hipMalloc((void **)&ctx->d_nzbits, sizeof(uint64_t) * nzb.size()));
    FW_HIP(ctx, hipMemcpy(ctx->d_nzbits, nzb.data(), sizeof(uint64_t) * nzb.size(), hipMemcpyHostToDevice));
    ctx->W = W;
    return FW_OK;
}

int fwi_fznz_level0(fw_ctx *ctx, std::vector<int32_t> &pi, std::vector<int32_t> &pj, std::vector<double> &stat,
                    std::vector<double> &pval, int64_t *m_reliable, FwL0Dev *dev)
{
    if (dev) *dev = FwL0Dev{};
    const int p = ctx->P.p;
    const long long npairs = (long long)p * (p - 1) / 2;
    unsigned long long cap = (unsigned long long)std::min<long long>(npairs, 4ll << 20);
    if (cap < ctx->l0_cap_hint) cap = ctx->l0_cap_hint;
    if (cap == 0) cap = 1;
    FzL0Counters h{};
    for (int attempt = 0; attempt < 2; ++attempt) {
        int rc;
        if ((rc = fw_dev_reserve(ctx, ctx->d_tmp0, sizeof(FzL0Counters)))) return rc;
        if ((rc = fw_dev_reserve(ctx, ctx->d_tmp1, cap * 2 * sizeof(int32_t)))) return rc;
        if ((rc = fw_dev_reserve(ctx, ctx->d_tmp2, cap * 2 * sizeof(double)))) return rc;
        FW_HIP(ctx, hipMemsetAsync(ctx->d_tmp0.ptr, 0, sizeof(FzL0Counters), ctx->stream));
        int32_t *oi = (int32_t *)ctx->d_tmp1.ptr, *oj = oi + cap;
        double *os = (double *)ctx->d_tmp2.ptr, *op = os + cap;
        dim3 grid((p + 15) / 16, (p + 15) / 16);
        hipLaunchKernelGGL(fznz_level0_kernel, grid, dim3(256), 0, ctx->stream, ctx->d_data,
                           (const unsigned long long *)ctx->d_nzbits, ctx->P.n, p, ctx->W, ctx->P.alpha,
                           (long long)ctx->n_obs_min_eff, (FzL0Counters *)ctx->d_tmp0.ptr, cap, oi, oj, os, op);
        FW_HIP(ctx, hipGetLastError());
        FW_HIP(ctx, hipMemcpyAsync(&h, ctx->d_tmp0.ptr, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->cnt.kernel_launches += 1;
        if (h.n_sig > ctx->l0_cap_hint) ctx->l0_cap_hint = h.n_sig;
        if (h.n_sig <= cap) {
            const size_t k = (size_t)h.n_sig;
            if (dev) {
                dev->i = oi;
                dev->j = oj;
                dev->stat64 = os;
                dev->pval = op;
                dev->k = k;
                *m_reliable = npairs - (long long)h.n_nan;
                return FW_OK;
            }
            pi.resize(k);
            pj.resize(k);
            stat.resize(k);
            pval.resize(k);
            if (k) {
                FW_HIP(ctx, hipMemcpy(pi.data(), oi, k * sizeof(int32_t), hipMemcpyDeviceToHost));
                FW_HIP(ctx, hipMemcpy(pj.data(), oj, k * sizeof(int32_t), hipMemcpyDeviceToHost));
                FW_HIP(ctx, hipMemcpy(stat.data(), os, k * sizeof(double), hipMemcpyDeviceToHost));
                FW_HIP(ctx, hipMemcpy(pval.data(), op, k * sizeof(double), hipMemcpyDeviceToHost));
            }
            *m_reliable = npairs - (long long)h.n_nan;
            return FW_OK;
        }
        cap = h.n_sig;
    }
    return fw_fail(ctx, FW_ERR_DEVICE, "fz_nz level-0: compaction buffer overflow twice");
}

// recs_host: one record per job of this launch (X, Y, acc_off, acc_len, m, cor_off filled); d_acc: flat accepted ints
int fwi_fznz_submatrices(fw_ctx *ctx, int64_t njobs, const FwNzJob *recs_host, size_t arena_floats, const int32_t *d_acc,
                         hipStream_t stream)
{
    int rc;
    if ((rc = fw_dev_reserve(ctx, ctx->d_nzrecs, (size_t)njobs * sizeof(FwNzJob)))) return rc;
    if ((rc = fw_dev_reserve(ctx, ctx->d_arena, std::max<size_t>(arena_floats, 1) * sizeof(float)))) return rc;
    FW_HIP(ctx, hipMemcpyAsync(ctx->d_nzrecs.ptr, recs_host, (size_t)njobs * sizeof(FwNzJob), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(fznz_submat_kernel, dim3((unsigned)njobs), dim3(256), 0, stream, ctx->d_data,
                       (const unsigned long long *)ctx->d_nzbits, ctx->P.n, ctx->W, (FwNzJob *)ctx->d_nzrecs.ptr, d_acc,
                       (float *)ctx->d_arena.ptr, ctx->P.alpha);
    FW_HIP(ctx, hipGetLastError());
    ctx->cnt.kernel_launches += 1;
    return FW_OK;
}

int fwi_fznz_segments(fw_ctx *ctx, int64_t nseg, int64_t nseg_tab, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out,
                      FwPoolBuf &pb)
{
    if (nseg == 0) return FW_OK;
    FW_HIP(ctx, hipEventRecord(pb.ev0, pb.launch_stream));
    if (ctx->P.max_k > 3)
        hipLaunchKernelGGL((fz_subsets_seg_kernel<true, true, false>), dim3((unsigned)nseg), dim3(256), 0, pb.launch_stream,
                           (const float *)ctx->d_arena.ptr, 0, d_segs, d_acc, d_out, ctx->P.max_k, ctx->P.alpha, 0.0,
                           (long long)ctx->P.max_tests, (const double *)nullptr, (const FwNzJob *)ctx->d_nzrecs.ptr,
                           (long long)ctx->n_obs_min_eff, (const unsigned *)nullptr, (const unsigned *)nullptr);
    else {
        if (nseg_tab > 0)
            hipLaunchKernelGGL((fz_subsets_seg_kernel<false, true, true>), dim3((unsigned)nseg_tab), dim3(256), 0, pb.launch_stream,
                               (const float *)ctx->d_arena.ptr, 0, d_segs, d_acc, d_out, ctx->P.max_k, ctx->P.alpha, 0.0,
                               (long long)ctx->P.max_tests, (const double *)nullptr, (const FwNzJob *)ctx->d_nzrecs.ptr,
                               (long long)ctx->n_obs_min_eff, (const unsigned *)nullptr, (const unsigned *)nullptr);
        if (nseg > nseg_tab)
            hipLaunchKernelGGL((fz_subsets_seg_kernel<false, true, false>), dim3((unsigned)(nseg - nseg_tab)), dim3(256), 0,
                               pb.launch_stream, (const float *)ctx->d_arena.ptr, 0, d_segs + nseg_tab, d_acc, d_out + nseg_tab,
                               ctx->P.max_k, ctx->P.alpha, 0.0, (long long)ctx->P.max_tests, (const double *)nullptr,
                               (const FwNzJob *)ctx->d_nzrecs.ptr, (long long)ctx->n_obs_min_eff, (const unsigned *)nullptr, (const unsigned *)nullptr);
    }
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipEventRecord(pb.ev1, pb.launch_stream));
    return FW_OK;
}

int fwi_fznz_test_batch(fw_ctx *ctx, int64_t m, const int32_t *X, const int32_t *Y, const int64_t *zoff,
                        const int32_t *zflat, fw_test_result *out)
{
    if (m == 0) return FW_OK;
    std::vector<FwNzJob> recs((size_t)m);
    size_t arena = 0;
    for (int64_t t = 0; t < m; ++t) {
        FwNzJob r{};
        r.X = X[t];
        r.Y = Y[t];
        r.acc_off = zoff[t];
        r.acc_len = (int32_t)(zoff[t + 1] - zoff[t]);
        r.m = r.acc_len + 2;
        r.cor_off = (long long)arena;
        arena += (size_t)r.m * r.m;
        recs[(size_t)t] = r;
    }
    const int64_t nz = zoff[m];
    int rc;
    if ((rc = fw_dev_reserve(ctx, ctx->d_acc, (size_t)std::max<int64_t>(nz, 1) * sizeof(int32_t)))) return rc;
    if ((rc = fw_dev_reserve(ctx, ctx->d_out, (size_t)m * sizeof(fw_test_result)))) return rc;
    if (nz > 0)
        FW_HIP(ctx, hipMemcpyAsync(ctx->d_acc.ptr, zflat, (size_t)nz * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = fwi_fznz_submatrices(ctx, m, recs.data(), arena, (const int32_t *)ctx->d_acc.ptr, ctx->stream))) return rc;
    hipLaunchKernelGGL(fznz_single_kernel, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, ctx->stream,
                       (const FwNzJob *)ctx->d_nzrecs.ptr, (const float *)ctx->d_arena.ptr, (long long)m,
                       (long long)ctx->n_obs_min_eff, (fw_test_result *)ctx->d_out.ptr);
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipMemcpyAsync(out, ctx->d_out.ptr, (size_t)m * sizeof(fw_test_result), hipMemcpyDeviceToHost, ctx->stream));
    FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cnt.kernel_launches += 1;
    return FW_OK;
}
