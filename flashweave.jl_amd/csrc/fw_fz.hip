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

struct FzL0Counters {
    unsigned long long n_sig;  // pairs with p < alpha (raw)
    unsigned long long n_nan;  // pairs with NaN p (excluded from m)
};

// SCREEN (fw_params.no_cor_mat, the reference's dense_cor = false): the tile of correlations is never written -- the epilogue runs the
// level-0 screen of fz_level0_kernel on the accumulators (same Float32 values, same thresholds) and appends the pairs that pass to
// the candidate list of fz_level0_exact_kernel.  No p x p matrix exists at any time: p is bounded by the data, not by p^2 floats.
#define GEMM_QCAP (2 * GEMM_BM * GEMM_LD / 3)  // screened pairs a workgroup queues in LDS (the first operand stage, free after the k loop)
template <bool SCREEN>
__global__ __launch_bounds__(256) void fz_cor_gemm_kernel(const float *__restrict__ xc, const float *__restrict__ sd,
                                                          float *__restrict__ cor, int p, int n_pad, int T,
                                                          int row_mode /* 1: whole tile rows from bi0 on, no mirrored writes (row-block sharding) */,
                                                          int bi0, const double *__restrict__ thr, FzL0Counters *cnt, unsigned long long cap,
                                                          int32_t *__restrict__ out_i, int32_t *__restrict__ out_j, float *__restrict__ out_r)
{
    // two LDS stages (73.7 KB): while a tile is being multiplied, the next one is already in registers and is written
    // to the other stage right after the MFMA block -- one barrier per k-tile
    __shared__ __attribute__((aligned(16))) float sA[2][GEMM_BM * GEMM_LD];
    __shared__ __attribute__((aligned(16))) float sB[2][GEMM_BM * GEMM_LD];
    int bi, bj;
    if (row_mode) {
        bi = bi0 + (int)blockIdx.x / T;
        bj = (int)blockIdx.x % T;
    } else {
        tri_decode(blockIdx.x, T, bi, bj);
    }
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
    __shared__ int s_qn;
    __shared__ unsigned long long s_qbase;
    int *q_i = (int *)&sA[0][0], *q_j = q_i + GEMM_QCAP;
    float *q_r = (float *)(q_j + GEMM_QCAP);
    float flo_pos = 0.0f, flo_neg = 0.0f;
    unsigned int n_nan = 0;
    if (SCREEN) {
        if (tid == 0) s_qn = 0;
        // (fz_level0_kernel: thresholds lowered by 1e-6 relative -- it can only let a few more pairs through to the exact kernel)
        flo_pos = (float)thr[0] * 0.999999f;
        flo_neg = (float)thr[2] * 0.999999f;
        __syncthreads();
    }
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
                    if (SCREEN) {
                        const bool in = i < j && j < p;
                        const bool isn = in && isnan(r);
                        n_nan += isn;
                        if (in && !isn && fabsf(r) >= (r < 0.0f ? flo_neg : flo_pos)) {
                            const int q = atomicAdd(&s_qn, 1);  // LDS
                            if (q < GEMM_QCAP) {
                                q_i[q] = i;
                                q_j[q] = j;
                                q_r[q] = r;
                            } else {
                                const unsigned long long slot = atomicAdd(&cnt->n_sig, 1ull);
                                if (slot < cap) {
                                    out_i[slot] = i;
                                    out_j[slot] = j;
                                    out_r[slot] = r;
                                }
                            }
                        }
                    } else if (i < p && j < p) {
                        cor[(size_t)i * p + j] = r;
                    }
                }
                if (!SCREEN && bi != bj && j < p && !row_mode) {
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
    if (SCREEN) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n_nan += __shfl_xor(n_nan, o);
        if (lane == 0 && n_nan) atomicAdd(&cnt->n_nan, (unsigned long long)n_nan);
        __syncthreads();
        const int nq = s_qn < GEMM_QCAP ? s_qn : GEMM_QCAP;
        if (tid == 0 && nq > 0) s_qbase = atomicAdd(&cnt->n_sig, (unsigned long long)nq);
        __syncthreads();
        for (int q = tid; q < nq; q += 256) {
            const unsigned long long slot = s_qbase + (unsigned long long)q;
            if (slot < cap) {
                out_i[slot] = q_i[q];
                out_j[slot] = q_j[q];
                out_r[slot] = q_r[q];
            }
        }
    }
}

#include "fw_fz_core.h"

__global__ void fz_thresholds_kernel(double alpha, double zscale, double *thr)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    fz_thresholds_dev(alpha, zscale, thr);
    // thr[4]: |r| below which x = |z|/sqrt2 < FZ_X_SUB for sure (see fz_seg_body); once here instead of once per thread
    thr[4] = zscale > 0.0 ? tanh(FZ_X_SUB * 0.7071067811865476 / zscale) * (1.0 - 1e-9) : 2.0;
    // thr[5..7]: the squared thresholds of the screened size-3 test (fz_seg_body: h2_pos, h2_neg, s2), once here so that the segment
    // kernel holds them in scalar registers
    thr[5] = thr[1] * thr[1] * (1.0 + 1e-12);
    thr[6] = thr[3] * thr[3] * (1.0 + 1e-12);
    thr[7] = (thr[4] < 1.0 ? thr[4] * thr[4] : 1.0) * (1.0 - 1e-12);
}

// ------------------------------------------------------------------------------------------------
// 4. level 0: all pairs i < j of the resident matrix (tests.jl:149-159 + the NaN/m rule of :397-398,522-526)
// ------------------------------------------------------------------------------------------------

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
// conditioning sets of 6 and 7 variables (r05; tests.jl:311-343 has no cap): the same bottom-up form, instantiated for K = 6, 7 -- only in
// the general-form kernels below (a 9 x 9 Float64 work matrix per lane: nothing for the table kernels' register budgets)
__device__ __forceinline__ double fz_pcor_any7(const float *__restrict__ cor, int p, int X, int Y, const int *z, int k)
{
    switch (k) {
        case 6: return fz_pcor_dp<6>(cor, p, X, Y, z);
        case 7: return fz_pcor_dp<7>(cor, p, X, Y, z);
        default: return fz_pcor_any(cor, p, X, Y, z, k);
    }
}

template <bool K7>
__global__ __launch_bounds__(256) void fz_test_batch_kernel(const float *__restrict__ cor, int p, long long m,
                                                            const int32_t *__restrict__ X, const int32_t *__restrict__ Y,
                                                            const long long *__restrict__ zoff,
                                                            const int32_t *__restrict__ zflat, double zscale,
                                                            fw_test_result *__restrict__ out)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= m) return;
    const int k = (int)(zoff[t + 1] - zoff[t]);
    constexpr int KM = K7 ? FW_MAX_K : FW_MAX_K_FAST;
    int z[KM];
    for (int q = 0; q < KM; ++q) z[q] = (q < k) ? zflat[zoff[t] + q] : 0;
    const double r = K7 ? fz_pcor_any7(cor, p, X[t], Y[t], z, k) : fz_pcor_any(cor, p, X[t], Y[t], z, k);
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

// Host-driven rounds: one workgroup per segment (ns_dev == nullptr).  Device-driven rounds (fw_devhiton.hip): a fixed
// grid strides over an unsorted segment list whose live length sits in device memory; the table / in-lane variants
// each pick their own segments.
#ifndef FW_HIGHK_OCC
#define FW_HIGHK_OCC 4  // workgroups per CU the size-4/5 variants are compiled for (4: 128 VGPRs, 2: 256 VGPRs)
#endif
template <bool HIGHK, bool LOCAL, bool TAB>
#ifndef FW_HIGHK_OCC_LONG
#define FW_HIGHK_OCC_LONG 3  // ... and the long-list variant with its level-3 position tables (48 KB of LDS, 168 VGPRs)
#endif
__global__ __launch_bounds__(256, HIGHK ? ((TAB || LOCAL) ? FW_HIGHK_OCC : FW_HIGHK_OCC_LONG) : 4) void fz_subsets_seg_kernel(const float *__restrict__ cor_g, int p_g,
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
        // routing by list length (workgroup-uniform); HIGHK without the flag word: the generic variant takes every segment
        // (per-job matrices, device rounds of fz_nz: the size-3 table / in-lane pair routes the same way; max_k 4-5 has one variant)
        if (ns_dev && (!LOCAL || (!HIGHK && big_dev)) && (!HIGHK || big_dev) && ((segs[s].acc_len <= (HIGHK ? FZ_HK_A : FZ_TAB_A)) != TAB)) continue;
        fz_seg_body<HIGHK, LOCAL, TAB>(cor_g, p_g, segs[s], accflat + segs[s].acc_off, false, out + s, max_k, alpha, zscale_g, max_tests, thr_g, recs,
                                       n_obs_min);
        __syncthreads();  // the LDS state of the body is reused by the next segment
    }
}

// ------------------------------------------------------------------------------------------------
// 6b. test_subsets with max_k = 6, 7 (r05): the GENERAL FORM.  One workgroup per segment, every thread a run of consecutive ranks
//     (unranked once by a linear scan with saturating binomials, then stepped lexicographically), every test the plain bottom-up
//     pcor_rec (fz_pcor_any7) and its exact p-value -- no tables, no thresholds, no lazy maximum: first stop = smallest stopping rank,
//     otherwise the `>=` maximum of the p-values with the later rank winning ties (tests.jl:326-341), reduced through LDS.  Same record
//     as the table kernels write, merged by the same host code.  A slow path by design: max_k beyond 5 is rare (the reference's
//     default is 3) and the host job pool drives it.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long fz_binom_sat7(int m, int t)  // C(m, t), t <= 7, saturating at 2^62
{
    if (t < 0 || m < t) return 0ull;
    double est = 1.0;
    for (int i = 1; i <= t; ++i) est = est * (double)(m - t + i) / (double)i;
    if (est > 2.0e18) return 1ull << 62;  // (the exact product below holds t * C(m, t) at most)
    unsigned long long v = 1ull;
    for (int i = 1; i <= t; ++i) v = v * (unsigned long long)(m - t + i) / (unsigned long long)i;
    return v;
}

template <bool LOCAL>
__global__ __launch_bounds__(256) void fz_subsets_slow_kernel(const float *__restrict__ cor_g, int p_g, const FwSeg *__restrict__ segs,
                                                              const int32_t *__restrict__ accflat, FwSegOut *__restrict__ out, int max_k,
                                                              double alpha, double zscale_g, long long max_tests,
                                                              const FwNzJob *__restrict__ recs, long long n_obs_min)
{
    __shared__ unsigned long long s_stop[256], s_br[256];
    __shared__ double s_bp[256];
    __shared__ unsigned int s_done[256];
    const FwSeg seg = segs[blockIdx.x];
    FwSegOut *out_rec = out + blockIdx.x;
    const int tid = threadIdx.x, a = seg.acc_len;
    const int32_t *gacc = accflat + seg.acc_off;
    const float *cor = cor_g;
    int p = p_g;
    double zscale = zscale_g;
    if (LOCAL) {
        const FwNzJob *rec = recs + seg.pad;
        cor = cor_g + rec->cor_off;
        p = rec->m;
        zscale = rec->zscale;
        if ((long long)rec->nR < n_obs_min) {  // tests.jl:294-296: (0, 1, 0, false) with zero tests (the marker of fz_seg_body)
            if (tid == 0) {
                FwSegOut o;
                o.stop_rank = 0;
                o.stop_stat = 0.0;
                o.stop_pval = 1.0;
                o.best_rank = 0;
                o.best_stat = 0.0;
                o.best_pval = -1.0;
                o.stop_df = -2;
                o.stop_power = 0;
                o.best_df = 0;
                o.pad = 0;
                o.evaluated = 0;
                *out_rec = o;
            }
            return;
        }
    }
    const int X = LOCAL ? 0 : seg.X, Y = LOCAL ? 1 : seg.Y;
    unsigned long long cnt[FW_MAX_K + 1];
    for (int s = FW_MAX_K; s >= 1; --s) cnt[s] = (s <= max_k) ? fz_binom_sat7(a, s) : 0ull;
    const unsigned long long len = seg.end - seg.start, R = (len + 255ull) / 256ull;
    const unsigned long long r0 = seg.start + (unsigned long long)tid * R;
    unsigned long long r1 = r0 + R;
    if (r1 > seg.end) r1 = seg.end;
    unsigned long long my_stop = FW_RANK_NONE, my_br = 0ull;
    double stop_stat = 0.0, stop_p = 0.0, my_bp = -1.0, my_bstat = 0.0;
    unsigned int my_done = 0u;
    if (r0 < seg.end) {
        unsigned long long rem = r0;
        int s = max_k;
        while (s > 1 && rem >= cnt[s]) {
            rem -= cnt[s];
            --s;
        }
        int pos[FW_MAX_K];
        for (int q = 0; q < FW_MAX_K; ++q) pos[q] = 0;
        {   // position d = the first c whose block of C(a - 1 - c, s - d - 1) subsets holds the rank (the scan of unrank_host)
            int prev = -1;
            for (int d = 0; d < s; ++d) {
                int c = prev + 1;
                for (;;) {
                    const unsigned long long with_c = fz_binom_sat7(a - 1 - c, s - d - 1);
                    if (rem < with_c) break;
                    rem -= with_c;
                    ++c;
                }
                pos[d] = c;
                prev = c;
            }
        }
        for (unsigned long long r = r0; r < r1; ++r) {
            int zs[FW_MAX_K];
            for (int q = 0; q < FW_MAX_K; ++q) zs[q] = (q < s) ? (LOCAL ? pos[q] + 2 : gacc[pos[q]]) : 0;
            const double stat = fz_pcor_any7(cor, p, X, Y, zs, s);
            const double pv = fz_pval_slow(stat, zscale);
            ++my_done;
            if (!(pv < alpha) || (max_tests > 0 && r + 1ull >= (unsigned long long)max_tests)) {
                my_stop = r;
                stop_stat = stat;
                stop_p = pv;
                break;
            }
            if (pv >= my_bp) {  // tests.jl:338 `>=`: the later rank wins ties
                my_bp = pv;
                my_bstat = stat;
                my_br = r;
            }
            int i = s - 1;
            while (i >= 0 && pos[i] == a - s + i) --i;
            if (i < 0) {
                --s;
                for (int q = 0; q < FW_MAX_K; ++q) pos[q] = q;
                if (s < 1) break;
            } else {
                ++pos[i];
                for (int j = i + 1; j < s; ++j) pos[j] = pos[j - 1] + 1;
            }
        }
    }
    s_stop[tid] = my_stop;
    s_bp[tid] = my_bp;
    s_br[tid] = my_br;
    s_done[tid] = my_done;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            if (s_stop[tid + o] < s_stop[tid]) s_stop[tid] = s_stop[tid + o];
            if (s_bp[tid + o] > s_bp[tid] || (s_bp[tid + o] == s_bp[tid] && s_br[tid + o] > s_br[tid])) {
                s_bp[tid] = s_bp[tid + o];
                s_br[tid] = s_br[tid + o];
            }
            s_done[tid] += s_done[tid + o];
        }
        __syncthreads();
    }
    const unsigned long long first = s_stop[0];
    FwSegOut o;
    o.stop_df = 0;
    o.best_df = 0;
    o.pad = 0;
    o.evaluated = s_done[0];
    if (first != FW_RANK_NONE) {
        if (my_stop != first) return;
        o.stop_rank = first;
        o.stop_stat = stop_stat;
        o.stop_pval = stop_p;
        o.best_rank = 0;
        o.best_stat = 0.0;
        o.best_pval = -1.0;
        o.stop_power = 1;
        *out_rec = o;
        return;
    }
    if (s_bp[0] < 0.0 ? tid != 0 : !(my_bp == s_bp[0] && my_br == s_br[0] && my_done > 0u)) return;  // the owner of the maximum writes (no test at all: thread 0)
    o.stop_rank = FW_RANK_NONE;
    o.stop_stat = 0.0;
    o.stop_pval = 0.0;
    o.best_rank = s_bp[0] < 0.0 ? 0ull : my_br;
    o.best_stat = s_bp[0] < 0.0 ? 0.0 : my_bstat;
    o.best_pval = s_bp[0] < 0.0 ? -1.0 : my_bp;
    o.stop_power = 1;
    *out_rec = o;
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
    hipLaunchKernelGGL(fz_cor_gemm_kernel<false>, dim3(nblk), dim3(256), 0, ctx->stream, ctx->d_xc, ctx->d_sd, ctx->d_cor, p,
                       ctx->n_pad, T, 0, 0, (const double *)nullptr, (FzL0Counters *)nullptr, 0ull, (int32_t *)nullptr, (int32_t *)nullptr,
                       (float *)nullptr);
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cnt.kernel_launches += 2;
    ctx->have_cor = true;
    return FW_OK;
}

// Row-block share of the matrix (fw_compute_cor_mat_rows): tile rows [t0, t1) against every tile column, written as whole rows.
// An element computed here and its mirror image computed by another rank are the same bits: the two MFMA operands swap roles, the
// products and their order over k do not change.
int fwi_fz_compute_cor_rows(fw_ctx *ctx, int rank, int world, int64_t *row0, int64_t *rows_per_rank)
{
    if (!ctx->have_data) return fw_fail(ctx, FW_ERR_STATE, "fw_compute_cor_mat_rows: no data uploaded (fw_set_data_dense_f32)");
    const int n = ctx->P.n, p = ctx->P.p;
    ctx->n_pad = (n + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
    ctx->p_pad = (p + GEMM_BM - 1) / GEMM_BM * GEMM_BM;
    const int T = ctx->p_pad / GEMM_BM;
    const int tpr = (T + world - 1) / world;
    *rows_per_rank = (int64_t)tpr * GEMM_BM;
    *row0 = (int64_t)rank * tpr * GEMM_BM;
    if (!ctx->d_cor || ctx->cor_capacity < (int64_t)world * tpr * GEMM_BM * p)
        return fw_fail(ctx, FW_ERR_STATE, "fw_compute_cor_mat_rows: needs a caller-owned matrix of at least %lld floats (fw_use_cor_buffer)",
                       (long long)world * tpr * GEMM_BM * p);
    if (!ctx->d_xc) FW_HIP(ctx, hipMalloc(&ctx->d_xc, sizeof(float) * (size_t)ctx->n_pad * ctx->p_pad));
    if (!ctx->d_sd) FW_HIP(ctx, hipMalloc(&ctx->d_sd, sizeof(float) * (size_t)ctx->p_pad));
    hipLaunchKernelGGL(fz_center_kernel, dim3(ctx->p_pad), dim3(256), 0, ctx->stream, ctx->d_data, ctx->d_xc, ctx->d_sd, n,
                       p, ctx->n_pad);
    const int t0 = std::min(rank * tpr, T), t1 = std::min(t0 + tpr, T);
    if (t1 > t0)
        hipLaunchKernelGGL(fz_cor_gemm_kernel<false>, dim3((unsigned)((t1 - t0) * T)), dim3(256), 0, ctx->stream, ctx->d_xc, ctx->d_sd, ctx->d_cor,
                           p, ctx->n_pad, T, 1, t0, (const double *)nullptr, (FzL0Counters *)nullptr, 0ull, (int32_t *)nullptr, (int32_t *)nullptr,
                           (float *)nullptr);
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cnt.kernel_launches += 2;
    ctx->have_cor = false;  // until the caller has gathered the other ranks' rows (fw_cor_mat_ready)
    ctx->cor_rows_rank = rank;
    ctx->cor_rows_world = world;
    ctx->cor_rows_per_rank = *rows_per_rank;
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
        if (ctx->P.no_cor_mat) {  // dense_cor = false: centred columns -> MFMA tiles -> screen, no matrix
            const int n = ctx->P.n;
            ctx->n_pad = (n + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
            ctx->p_pad = (p + GEMM_BM - 1) / GEMM_BM * GEMM_BM;
            if (!ctx->d_xc) FW_HIP(ctx, hipMalloc(&ctx->d_xc, sizeof(float) * (size_t)ctx->n_pad * ctx->p_pad));
            if (!ctx->d_sd) FW_HIP(ctx, hipMalloc(&ctx->d_sd, sizeof(float) * (size_t)ctx->p_pad));
            if (attempt == 0)
                hipLaunchKernelGGL(fz_center_kernel, dim3(ctx->p_pad), dim3(256), 0, ctx->stream, ctx->d_data, ctx->d_xc, ctx->d_sd, n, p,
                                   ctx->n_pad);
            const int T = ctx->p_pad / GEMM_BM;
            hipLaunchKernelGGL(fz_cor_gemm_kernel<true>, dim3((unsigned)(T * (T + 1) / 2)), dim3(256), 0, ctx->stream, ctx->d_xc, ctx->d_sd,
                               (float *)nullptr, p, ctx->n_pad, T, 0, 0, (const double *)ctx->d_thr, d_c1, cap, ci, cj, cr);
        } else {
            dim3 grid((p + FZ_L0_COLS - 1) / FZ_L0_COLS, (p + FZ_L0_ROWS - 1) / FZ_L0_ROWS);
            hipLaunchKernelGGL(fz_level0_kernel, grid, dim3(256), 0, ctx->stream, ctx->d_cor, p, (const double *)ctx->d_thr, d_c1, cap, ci,
                               cj, cr);
        }
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
    int kmax = 0;
    for (int64_t t = 0; t < m; ++t) kmax = std::max<int>(kmax, (int)(zoff[t + 1] - zoff[t]));
    if (kmax > FW_MAX_K_FAST)
        hipLaunchKernelGGL(fz_test_batch_kernel<true>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_cor,
                           ctx->P.p, (long long)m, dX, dY, dz, (const int32_t *)ctx->d_acc.ptr, fz_zscale(ctx),
                           (fw_test_result *)ctx->d_out.ptr);
    else
        hipLaunchKernelGGL(fz_test_batch_kernel<false>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_cor,
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

#ifdef FW_FZ_FASTDBG
extern "C" void fwi_fz_fastdbg_print()
{
    unsigned long long c[24] = {0};
    if (hipMemcpyFromSymbol(c, HIP_SYMBOL(fz_fast_cnt), sizeof(c)) == hipSuccess && c[19])
        fprintf(stderr, "[fw] cheap-screen potential: of %llu fast-loop wave-iterations, every lane clearly (relative margin on |stat|^2: 2e-4 / 2e-3 / 2e-2) above the wavefront's minimum so far and inside the sure range in %llu / %llu / %llu\n",
                c[19], c[16], c[17], c[18]);
    if (c[23])
        fprintf(stderr, "[fw] cheap screen (validation build: decides nothing): lanes it would skip %llu, of them VIOLATIONS (exact value not above the bound, or not sure) %llu; wave-iterations it would skip whole %llu of %llu\n",
                c[20], c[21], c[22], c[23]);
    if (hipMemcpyFromSymbol(c, HIP_SYMBOL(fz_fast_cnt), sizeof(c)) == hipSuccess)
        fprintf(stderr, "[fw] fast loop: wave-iterations %llu, lane-tests %llu; waves with a lane not clean %llu, not significant-for-sure %llu, beyond the normal range of p %llu, behind a stop %llu, tie %llu; lanes beyond the normal range %llu\n",
                c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
    if (c[8])
        fprintf(stderr, "[fw] segments (table variant) %llu, evaluated %llu; 100 MHz ticks per segment (first wavefront): prologue %.1f, table build %.1f, lane loop %.1f, reductions %.1f, whole body %.1f\n",
                c[8], c[14], (double)c[9] / c[8], (double)c[10] / c[8], (double)c[11] / c[8], (double)c[12] / c[8], (double)c[13] / c[8]);
}
#endif
static int fz_ensure_thresholds(fw_ctx *ctx, hipStream_t stream)
{
    if (!ctx->d_thr) {
        const char *dbg = fw_knob("FW_FZ_DBG");
        const int flags = dbg ? atoi(dbg) : 0;
        FW_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(fz_dbg_flags), &flags, sizeof(int)));
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_thr, 12 * sizeof(double)));
        hipLaunchKernelGGL(fz_thresholds_kernel, dim3(1), dim3(64), 0, stream, ctx->P.alpha, fz_zscale(ctx), ctx->d_thr);
        FW_HIP(ctx, hipGetLastError());
        FW_HIP(ctx, hipStreamSynchronize(stream));  // another stream may use it next
    }
    return FW_OK;
}

// Device-driven rounds (fw_devhiton.hip): a fixed grid over an unsorted segment list whose live length is *d_ns.
int fwi_fz_thresholds(fw_ctx *ctx, hipStream_t stream, double *zscale)
{
    *zscale = fz_zscale(ctx);
    return fz_ensure_thresholds(ctx, stream);
}

int fwi_fz_segments_dev(fw_ctx *ctx, unsigned grid, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out, const unsigned *d_ns,
                        bool any_big, const unsigned *d_big, hipStream_t stream)
{
    int rc = fz_ensure_thresholds(ctx, stream);
    if (rc) return rc;
    const unsigned grid_big = grid < 512u ? grid : 512u;  // accepted sets beyond FZ_TAB_A are rare: few striding workgroups
    if (ctx->P.max_k > 3) {
        // level-2 table variant for |accepted| <= FZ_HK_A, generic variant for the longer lists (both stride the list)
        const bool no_hk = fw_knob("FW_NO_HK") != nullptr;  // profiling / test knob (read per call)
        if (!no_hk)
            hipLaunchKernelGGL((fz_subsets_seg_kernel<true, false, true>), dim3(grid), dim3(256), 0, stream, ctx->d_cor, ctx->P.p, d_segs,
                               d_acc, d_out, ctx->P.max_k, ctx->P.alpha, fz_zscale(ctx), (long long)ctx->P.max_tests, ctx->d_thr,
                               (const FwNzJob *)nullptr, 0ll, d_ns, d_big);
        hipLaunchKernelGGL((fz_subsets_seg_kernel<true, false, false>), dim3(grid), dim3(256), 0, stream, ctx->d_cor,
                           ctx->P.p, d_segs, d_acc, d_out, ctx->P.max_k, ctx->P.alpha, fz_zscale(ctx), (long long)ctx->P.max_tests,
                           ctx->d_thr, (const FwNzJob *)nullptr, 0ll, d_ns, no_hk ? (const unsigned *)nullptr : d_big);
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
    if (ctx->P.max_k > FW_MAX_K_FAST) {  // conditioning sets of 6 and 7 variables: the general form for every segment
        hipLaunchKernelGGL(fz_subsets_slow_kernel<false>, dim3((unsigned)nseg), dim3(256), 0, pb.launch_stream, ctx->d_cor, ctx->P.p, d_segs, d_acc,
                           d_out, ctx->P.max_k, ctx->P.alpha, fz_zscale(ctx), (long long)ctx->P.max_tests, (const FwNzJob *)nullptr, 0ll);
    } else if (ctx->P.max_k > 3) {
        // segments [0, nseg_tab) belong to jobs with |accepted| <= FZ_HK_A: level-2 table kernel; the rest: generic form
        if (nseg_tab > 0)
            hipLaunchKernelGGL((fz_subsets_seg_kernel<true, false, true>), dim3((unsigned)nseg_tab), dim3(256), 0, pb.launch_stream,
                               ctx->d_cor, ctx->P.p, d_segs, d_acc, d_out, ctx->P.max_k, ctx->P.alpha, fz_zscale(ctx),
                               (long long)ctx->P.max_tests, ctx->d_thr, (const FwNzJob *)nullptr, 0ll, (const unsigned *)nullptr, (const unsigned *)nullptr);
        if (nseg > nseg_tab)
            hipLaunchKernelGGL((fz_subsets_seg_kernel<true, false, false>), dim3((unsigned)(nseg - nseg_tab)), dim3(256), 0,
                               pb.launch_stream, ctx->d_cor, ctx->P.p, d_segs + nseg_tab, d_acc, d_out + nseg_tab, ctx->P.max_k,
                               ctx->P.alpha, fz_zscale(ctx), (long long)ctx->P.max_tests, ctx->d_thr, (const FwNzJob *)nullptr,
                               0ll, (const unsigned *)nullptr, (const unsigned *)nullptr);
    } else {
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

// Per-job |r| thresholds for `p < alpha` (every job has its own n_R, hence its own Fisher-z scale).  p = erfc(|z| / sqrt2)
// with z = zscale * log((1 + r) / (1 - r)) is below alpha exactly when |z| / sqrt2 exceeds xcrit = erfc^-1(alpha) -- one
// number per context, found on the host -- so the threshold is r* = tanh(xcrit / (sqrt2 zscale)) in closed form; the
// +-1e-9 guard band (inside which the exact p-value decides, see fz_seg_body) is six orders of magnitude wider than the
// rounding of tanh / erfc.  The r01 form bisected the device p-value 400 times per job in ONE lane: 0.4 ms, the whole
// duration of a sub-matrix launch.  xcrit < 0 (alpha >= 1: no root) keeps the bisection.
__device__ __forceinline__ void fznz_thresholds(double alpha, double xcrit, double zscale, double *thr)
{
    if (!(xcrit >= 0.0)) {
        fz_thresholds_dev(alpha, zscale, thr);
        return;
    }
    if (!(zscale > 0.0)) {  // p = 1 for every r: never significant
        thr[0] = thr[1] = thr[2] = thr[3] = 2.0;
        return;
    }
    const double rs = tanh(xcrit * 0.7071067811865476 / zscale);  // |z| / sqrt2 = xcrit  <=>  atanh(r) = xcrit / (sqrt2 zscale)
    thr[0] = thr[2] = rs * (1.0 - 1e-9);
    thr[1] = thr[3] = rs * (1.0 + 1e-9);
}

#define FZNZ_NT 256                // threads per job workgroup (4 wavefronts)
#define FZNZ_NW (FZNZ_NT / 64)
#define FZNZ_RED_STRIDE (8 * 72)  // per-wavefront reduction scratch: 8 accumulators x (8 x 9) padded partials

// dynamic LDS of a launch (bytes): reduction scratch, then per-variable arrays for the largest job, then the row list
static inline size_t fznz_lds_bytes(int m_cap, int n)
{
    const size_t rows = n <= FZNZ_ROWS_LDS ? (size_t)((n + 3) & ~3) * sizeof(unsigned short) : 0;
    return sizeof(double) * FZNZ_NW * FZNZ_RED_STRIDE + (size_t)m_cap * (2 * sizeof(double) + sizeof(int)) + 264 * sizeof(int) + rows;
}

__global__ __launch_bounds__(FZNZ_NT) void fznz_submat_kernel(const float *__restrict__ data, const unsigned long long *__restrict__ nz,
                                                              int n, int W, FwNzJob *__restrict__ recs,
                                                              const int32_t *__restrict__ accflat, float *__restrict__ arena,
                                                              double alpha, int m_cap, double xcrit,
                                                              double *__restrict__ arena64 /* recursive_pcor = 0: the UNROUNDED Float64
                                                              correlations go here (same element offsets) instead of the Float32 matrix:
                                                              no clamp, no NaN -> 0 (that is cor_subset!'s), a variable listed twice
                                                              correlates with itself exactly 1 (StatsBase sums the same numbers) */,
                                                              int m_lo /* jobs with m <= m_lo belong to another launch (host pool: 0) */)
{
    extern __shared__ double s_dyn[];
    __shared__ double s_ss01[2];
    double *s_red = s_dyn;                                   // [FZNZ_NW][FZNZ_RED_STRIDE]
    double *s_mean = s_red + FZNZ_NW * FZNZ_RED_STRIDE;      // [m_cap]
    double *s_sd = s_mean + m_cap;                           // [m_cap]
    int *s_var = (int *)(s_sd + m_cap);                      // [m_cap]  (m_cap is even: 8-byte alignment holds)
    int *s_woff = s_var + m_cap;                             // [257] (+ padding)
    unsigned short *s_rows = (unsigned short *)(s_woff + 264);  // [n] when n <= FZNZ_ROWS_LDS
    FwNzJob *rec = recs + blockIdx.x;
    // device rounds (fw_devhiton.hip: one record slot per target, every slot a workgroup): pad bit 0 = nothing to compute (no job, or a
    // job in a later window whose matrix is still in its slot); m_lo < m <= m_cap selects the launch's share when the round launches
    // this kernel twice (small jobs with small LDS arrays at four workgroups per CU, the few long lists with arrays for the longest)
    if ((rec->pad & 1) || rec->m <= m_lo || rec->m > m_cap) return;
    const int m = rec->m, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long *mx = nz + (size_t)rec->X * W, *my = nz + (size_t)rec->Y * W;
    for (int t = tid; t < m; t += FZNZ_NT) s_var[t] = t == 0 ? rec->X : (t == 1 ? rec->Y : accflat[rec->acc_off + t - 2]);
    long long nR = 0;
    for (int w = 0; w < W; ++w) nR += __popcll(mx[w] & my[w]);
    float *local = arena + rec->cor_off;
    double *local64 = arena64 ? arena64 + rec->cor_off : nullptr;
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
        for (int t = wave; t < m; t += FZNZ_NW) {  // column means and norms
            const float *col = data + (size_t)s_var[t] * n;
            double sacc = 0.0;
#pragma unroll 4
            for (int q = lane; q < nr; q += 64) sacc += (double)col[s_rows[q]];
            const double mean = fznz_tree64(sacc) / (double)nR;
            double ss = 0.0;
#pragma unroll 4
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
        // Dot products in register tiles: one wavefront per 4 x 4 block of pairs (a in 4A.., b in 4B.., A <= B) -- every
        // row costs 8 gathered values for 16 products instead of 32, and the per-pair sum is the same sequence of
        // operations as before (lane l adds the rows l, l + 64, ... of ITS pair in Float64, then tree64 over the lanes).
        // The 64 x 16 partials meet through LDS, eight accumulators at a time: lane (c, g) = (lane >> 3, lane & 7) adds
        // the partials of accumulator c held by lanes 8 g .. 8 g + 7 as the balanced tree of fznz_tree64 --
        // ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7)) -- and three shuffles add the eight groups pairwise; every
        // node of that tree has the same two operands as in the DPP form (addition commutes bit for bit), so the
        // oracle's mirrored order still applies.
        const int nb = (m + 3) >> 2;
        const int ntiles = nb * (nb + 1) / 2;
        double *red = s_red + wave * FZNZ_RED_STRIDE;
        for (int t = wave; t < ntiles; t += FZNZ_NW) {
            int A = (int)(((2.0 * nb + 1.0) - sqrt((2.0 * nb + 1.0) * (2.0 * nb + 1.0) - 8.0 * (double)t)) * 0.5);
            while (A > 0 && A * (2 * nb - A + 1) / 2 > t) --A;
            while ((A + 1) * (2 * nb - A) / 2 <= t) ++A;
            const int B = A + (t - A * (2 * nb - A + 1) / 2);
            const float *ca[4], *cb[4];
            double ma[4], mb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ia = 4 * A + i < m ? 4 * A + i : m - 1, ib = 4 * B + i < m ? 4 * B + i : m - 1;
                ca[i] = data + (size_t)s_var[ia] * n;
                cb[i] = data + (size_t)s_var[ib] * n;
                ma[i] = s_mean[ia];
                mb[i] = s_mean[ib];
            }
            double acc[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] = 0.0;
            for (int q = lane; q < nr; q += 64) {
                const int row = s_rows[q];
                double xa[4], xb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xa[i] = (double)ca[i][row] - ma[i];
                    xb[i] = (double)cb[i][row] - mb[i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[4 * i + j] += xa[i] * xb[j];
            }
            const int c = lane >> 3, g = lane & 7;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                {
                    double *dst = red + 9 * (lane >> 3) + (lane & 7);
#pragma unroll
                    for (int k = 0; k < 8; ++k) dst[72 * k] = acc[8 * h + k];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const double *src = red + 72 * c + 9 * g;
                double x[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = src[i];
#pragma unroll
                for (int w = 1; w < 8; w <<= 1)
#pragma unroll
                    for (int i = 0; i < 8; i += 2 * w) x[i] = x[i] + x[i + w];
                double tot = x[0];
                tot = tot + __shfl_xor(tot, 1);
                tot = tot + __shfl_xor(tot, 2);
                tot = tot + __shfl_xor(tot, 4);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int k16 = 8 * h + c;
                const int a = 4 * A + (k16 >> 2), b = 4 * B + (k16 & 3);
                if (g == 0 && a < b && b < m && local64) {
                    const double r64 = s_var[a] == s_var[b] ? 1.0 : tot / (s_sd[a] * s_sd[b]);
                    local64[(size_t)a * m + b] = r64;
                    local64[(size_t)b * m + a] = r64;
                } else if (g == 0 && a < b && b < m) {
                    double r = tot / (s_sd[a] * s_sd[b]);
                    if (r > 1.0) r = 1.0;
                    if (r < -1.0) r = -1.0;
                    if (isnan(r)) r = 0.0;  // statfuns.jl:150
                    const float rf = (float)r;  // the scratch matrix of the reference is Float32 (learning.jl:127-129)
                    local[(size_t)a * m + b] = rf;
                    local[(size_t)b * m + a] = rf;
                }
            }
        }
        for (int t = tid; t < m; t += FZNZ_NT) {
            if (local64)
                local64[(size_t)t * m + t] = 1.0;
            else
                local[(size_t)t * m + t] = 1.0f;
        }
        if (tid == 0) {
            rec->rxy = 0.0;  // (only univariate jobs read it: they take the sequential form below)
            rec->nR = (int32_t)nR;
            const long long sf = nR - 3;
            rec->zscale = sf > 0 ? sqrt((double)sf) / 2.0 : 0.0;
            fznz_thresholds(alpha, xcrit, rec->zscale, rec->thr);
        }
        return;
    }
    __syncthreads();
    // sequential form (univariate jobs -- their pair statistic must equal level 0's, statfuns.jl:91-123 in row order -- and
    // views beyond the LDS row list): column means and norms over R, sequential Float64 sums in row order
    for (int t = tid; t < m; t += FZNZ_NT) {
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
    for (long long q = tid; q < npairs; q += FZNZ_NT) {
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
        if (local64) {
            const double r64 = s_var[a] == s_var[b] ? 1.0 : sacc / (s_sd[a] * s_sd[b]);
            local64[(size_t)a * m + b] = r64;
            local64[(size_t)b * m + a] = r64;
            continue;
        }
        double r = sacc / (s_sd[a] * s_sd[b]);
        if (r > 1.0) r = 1.0;
        if (r < -1.0) r = -1.0;
        if (isnan(r)) r = 0.0;  // statfuns.jl:150
        const float rf = (float)r;  // the scratch matrix of the reference is Float32 (learning.jl:127-129)
        local[(size_t)a * m + b] = rf;
        local[(size_t)b * m + a] = rf;
    }
    for (int t = tid; t < m; t += FZNZ_NT) {
        if (local64)
            local64[(size_t)t * m + t] = 1.0;
        else
            local[(size_t)t * m + t] = 1.0f;
    }
    if (tid == 0) {
        rec->nR = (int32_t)nR;
        const long long sf = nR - 3;
        rec->zscale = sf > 0 ? sqrt((double)sf) / 2.0 : 0.0;
        fznz_thresholds(alpha, xcrit, rec->zscale, rec->thr);
        // the matrix is in its slot: nothing to compute until the slot gets a fresh record.  (Device rounds: a target that finishes and is
        // compacted off the active list in the same round never gets its "pad = 1" from dh_nz_recs_kernel; r05 recomputed the matrix of
        // every such target in every later round of the run, from a list buffer the state machine had reused -- never read, but computed.)
        rec->pad |= 1;
    }
}

// ---- explicit single tests (fw_test_batch): one thread per test on the job's local matrix ----
// StatsBase.partialcor on a job's Float64 correlation matrix (statfuns.jl:19-21): condition on Z_k, ..., Z_1 in turn, clamp at the end
// (the recursion of fw_fzs.hip / the oracle's fwo_pcor, one thread, m <= FW_MAX_K + 2)
__device__ double fznz_partialcor64(const double *__restrict__ C, int m)
{
    double R[FW_MAX_K + 2][FW_MAX_K + 2];
    for (int a = 0; a < m; ++a)
        for (int b = 0; b < m; ++b) R[a][b] = C[a * m + b];
    for (int t = m - 1; t >= 2; --t)
        for (int a = 0; a < t; ++a)
            for (int b = a + 1; b < t; ++b) {
                const double r = (R[a][b] - R[a][t] * R[b][t]) / (sqrt(1.0 - R[a][t] * R[a][t]) * sqrt(1.0 - R[b][t] * R[b][t]));
                R[a][b] = R[b][a] = r;
            }
    double r = R[0][1];
    if (r < -1.0) r = -1.0;
    if (r > 1.0) r = 1.0;
    return r;
}

__global__ __launch_bounds__(64) void fznz_single_kernel(const FwNzJob *__restrict__ recs, const float *__restrict__ arena,
                                                         long long m_tests, long long n_obs_min,
                                                         fw_test_result *__restrict__ out, const double *__restrict__ arena64)
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
        const double r = arena64 ? fznz_partialcor64(arena64 + rec.cor_off, rec.m) : fz_pcor_any7(arena + rec.cor_off, rec.m, 0, 1, z, rec.acc_len);
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

// erfc^-1(alpha) by bisection on the host's erfc (-1: alpha >= 1 or not a probability, the kernel bisects per job instead)
static double fznz_xcrit(double alpha)
{
    if (!(alpha > 0.0) || !(alpha < 1.0)) return -1.0;
    double lo = 0.0, hi = 40.0;  // erfc(0) = 1 >= alpha, erfc(40) = 0 < alpha
    for (int it = 0; it < 200; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (std::erfc(mid) < alpha)
            hi = mid;
        else
            lo = mid;
    }
    return 0.5 * (lo + hi);
}

// recs_host: one record per job of this launch (X, Y, acc_off, acc_len, m, cor_off filled); d_acc: flat accepted ints
int fwi_fznz_submatrices(fw_ctx *ctx, int64_t njobs, const FwNzJob *recs_host, size_t arena_floats, const int32_t *d_acc,
                         hipStream_t stream, bool f64)
{
    int rc;
    static const bool nz_trace = fw_knob("FW_NZ_TRACE") != nullptr;  // profiling: shape of every sub-matrix launch
    if (nz_trace) {
        long long mmax = 0, pairs = 0, uni = 0;
        for (int64_t j = 0; j < njobs; ++j) {
            mmax = std::max<long long>(mmax, recs_host[j].m);
            pairs += (long long)recs_host[j].m * (recs_host[j].m - 1) / 2;
            uni += recs_host[j].acc_len == 0;
        }
        fprintf(stderr, "[fw] fz_nz sub-matrices: %lld jobs (%lld univariate), largest m %lld, %lld pairs\n", (long long)njobs, uni, mmax, pairs);
    }
    if ((rc = fw_dev_reserve(ctx, ctx->d_nzrecs, (size_t)njobs * sizeof(FwNzJob)))) return rc;
    if ((rc = fw_dev_reserve(ctx, ctx->d_arena, std::max<size_t>(arena_floats, 1) * (f64 ? sizeof(double) : sizeof(float))))) return rc;
    FW_HIP(ctx, hipMemcpyAsync(ctx->d_nzrecs.ptr, recs_host, (size_t)njobs * sizeof(FwNzJob), hipMemcpyHostToDevice, stream));
    int m_cap = 4;
    for (int64_t j = 0; j < njobs; ++j) m_cap = std::max(m_cap, (int)recs_host[j].m);
    m_cap = (m_cap + 15) & ~15;  // LDS arrays of the launch are sized for its largest job
    const size_t lds = fznz_lds_bytes(m_cap, ctx->P.n);
    if (lds > 160u * 1024u - 64u)
        return fw_fail(ctx, FW_ERR_LIMIT, "fz_nz: a job with %d variables does not fit the LDS of one workgroup", m_cap);
    if (!ctx->fznz_lds_raised) {  // raise the kernel's dynamic-LDS limit (above the 64 KB default) once per context, i.e. per device
        FW_HIP(ctx, hipFuncSetAttribute((const void *)fznz_submat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160u * 1024u - 64u)));
        ctx->fznz_lds_raised = true;
    }
    hipLaunchKernelGGL(fznz_submat_kernel, dim3((unsigned)njobs), dim3(FZNZ_NT), lds, stream, ctx->d_data,
                       (const unsigned long long *)ctx->d_nzbits, ctx->P.n, ctx->W, (FwNzJob *)ctx->d_nzrecs.ptr, d_acc,
                       (float *)ctx->d_arena.ptr, ctx->P.alpha, m_cap, fznz_xcrit(ctx->P.alpha), f64 ? (double *)ctx->d_arena.ptr : (double *)nullptr, 0);
    FW_HIP(ctx, hipGetLastError());
    ctx->cnt.kernel_launches += 1;
    return FW_OK;
}

// Device rounds of fz_nz (fw_devhiton.hip, r05): one record slot per target of the run, the records written on the device
// (dh_nz_recs_kernel); one workgroup per slot, idle slots leave at once.  Two launches when the run can hold long lists: jobs of up to
// FZNZ_DEV_SMALL variables with LDS arrays for that many (four workgroups per CU), longer ones with arrays for the longest list possible.
#define FZNZ_DEV_SMALL 64
int fwi_fznz_dev_limits(fw_ctx *ctx, int m_max)
{
    const int m_cap = (std::max(m_max, 4) + 15) & ~15;
    return fznz_lds_bytes(m_cap, ctx->P.n) <= 160u * 1024u - 64u ? FW_OK : FW_ERR_LIMIT;
}
int fwi_fznz_submatrices_dev(fw_ctx *ctx, int nslots, FwNzJob *d_recs, const int32_t *d_acc, float *d_arena, int m_max, bool any_long, hipStream_t stream)
{
    if (!ctx->fznz_lds_raised) {
        FW_HIP(ctx, hipFuncSetAttribute((const void *)fznz_submat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160u * 1024u - 64u)));
        ctx->fznz_lds_raised = true;
    }
    const int m_small = std::min(FZNZ_DEV_SMALL, (std::max(m_max, 4) + 15) & ~15);
    hipLaunchKernelGGL(fznz_submat_kernel, dim3((unsigned)nslots), dim3(FZNZ_NT), fznz_lds_bytes(m_small, ctx->P.n), stream, ctx->d_data,
                       (const unsigned long long *)ctx->d_nzbits, ctx->P.n, ctx->W, d_recs, d_acc, d_arena, ctx->P.alpha, m_small,
                       fznz_xcrit(ctx->P.alpha), (double *)nullptr, 0);
    if (any_long && m_max > m_small) {
        const int m_cap = (m_max + 15) & ~15;
        hipLaunchKernelGGL(fznz_submat_kernel, dim3((unsigned)nslots), dim3(FZNZ_NT), fznz_lds_bytes(m_cap, ctx->P.n), stream, ctx->d_data,
                           (const unsigned long long *)ctx->d_nzbits, ctx->P.n, ctx->W, d_recs, d_acc, d_arena, ctx->P.alpha, m_cap,
                           fznz_xcrit(ctx->P.alpha), (double *)nullptr, m_small);
    }
    FW_HIP(ctx, hipGetLastError());
    return FW_OK;
}

// ... and the enumeration on the job-local matrices: the segment list is the device-built one (`*d_ns` records), seg.pad = record slot
int fwi_fznz_segments_dev(fw_ctx *ctx, unsigned grid, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out, const unsigned *d_ns,
                          bool any_big, const unsigned *d_big, const FwNzJob *d_recs, const float *d_arena, hipStream_t stream)
{
    if (ctx->P.max_k > 3) {
        hipLaunchKernelGGL((fz_subsets_seg_kernel<true, true, false>), dim3(grid), dim3(256), 0, stream, d_arena, 0, d_segs, d_acc, d_out, ctx->P.max_k,
                           ctx->P.alpha, 0.0, (long long)ctx->P.max_tests, (const double *)nullptr, d_recs, (long long)ctx->n_obs_min_eff, d_ns,
                           (const unsigned *)nullptr);
    } else {
        hipLaunchKernelGGL((fz_subsets_seg_kernel<false, true, true>), dim3(grid), dim3(256), 0, stream, d_arena, 0, d_segs, d_acc, d_out, ctx->P.max_k,
                           ctx->P.alpha, 0.0, (long long)ctx->P.max_tests, (const double *)nullptr, d_recs, (long long)ctx->n_obs_min_eff, d_ns, d_big);
        if (any_big)
            hipLaunchKernelGGL((fz_subsets_seg_kernel<false, true, false>), dim3(grid < 512u ? grid : 512u), dim3(256), 0, stream, d_arena, 0, d_segs,
                               d_acc, d_out, ctx->P.max_k, ctx->P.alpha, 0.0, (long long)ctx->P.max_tests, (const double *)nullptr, d_recs,
                               (long long)ctx->n_obs_min_eff, d_ns, d_big);
    }
    FW_HIP(ctx, hipGetLastError());
    return FW_OK;
}

int fwi_fznz_segments(fw_ctx *ctx, int64_t nseg, int64_t nseg_tab, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out,
                      FwPoolBuf &pb)
{
    if (nseg == 0) return FW_OK;
    FW_HIP(ctx, hipEventRecord(pb.ev0, pb.launch_stream));
    if (ctx->P.max_k > FW_MAX_K_FAST)  // conditioning sets of 6 and 7 variables: the general form on the job-local matrices
        hipLaunchKernelGGL(fz_subsets_slow_kernel<true>, dim3((unsigned)nseg), dim3(256), 0, pb.launch_stream, (const float *)ctx->d_arena.ptr, 0, d_segs,
                           d_acc, d_out, ctx->P.max_k, ctx->P.alpha, 0.0, (long long)ctx->P.max_tests, (const FwNzJob *)ctx->d_nzrecs.ptr,
                           (long long)ctx->n_obs_min_eff);
    else if (ctx->P.max_k > 3)
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
    const bool f64 = !ctx->P.recursive_pcor;  // no cor_mat: StatsBase.partialcor on the view (tests.jl:253)
    if ((rc = fwi_fznz_submatrices(ctx, m, recs.data(), arena, (const int32_t *)ctx->d_acc.ptr, ctx->stream, f64))) return rc;
    hipLaunchKernelGGL(fznz_single_kernel, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, ctx->stream,
                       (const FwNzJob *)ctx->d_nzrecs.ptr, (const float *)ctx->d_arena.ptr, (long long)m,
                       (long long)ctx->n_obs_min_eff, (fw_test_result *)ctx->d_out.ptr, f64 ? (const double *)ctx->d_arena.ptr : (const double *)nullptr);
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipMemcpyAsync(out, ctx->d_out.ptr, (size_t)m * sizeof(fw_test_result), hipMemcpyDeviceToHost, ctx->stream));
    FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cnt.kernel_launches += 1;
    return FW_OK;
}


// ------------------------------------------------------------------------------------------------
// device self-test of the hand-written arithmetic sequences (fw_selftest, include/flashweave_amd.h)
// ------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ unsigned long long st_mix(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// FW_SELFTEST_DIV: fz_div_nn(n, d) against the compiler's IEEE division, bit for bit, on the operand ranges of the NaN-free
// partial-correlation path: numerators = round5 values (|k| <= 200 000 multiples of 1e-5 as Float32-converted and as Float64
// values, zero and -0 included), denominators = products of square roots of 1 - v^2 (Float32 roots converted, Float64 roots;
// log-uniform in [2^-54, 1], the end points and powers of two included)
__global__ void fz_selftest_div_kernel(unsigned long long cases, unsigned long long seed, unsigned long long *mismatches)
{
    unsigned long long bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < cases;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long h = st_mix(seed + i), h2 = st_mix(h);
        const int k = (int)(h % 400001ull) - 200000;
        double n;
        if ((h >> 32) & 1ull)
            n = (double)round5_f32_nn((float)k * 1e-5f);
        else
            n = round5_f64_nn((double)k * 1e-5);
        if (((h >> 33) & 1023ull) == 0ull) n = -0.0;
        // two factors in (0, 1]: mantissa from the hash, exponent log-uniform
        const double m1 = 1.0 + (double)(h2 & 0xFFFFFFFFFFFFFull) * 2.220446049250313e-16;         // [1, 2)
        const double m2 = 1.0 + (double)((h2 >> 11) & 0xFFFFFull) * 9.5367431640625e-07;            // coarse mantissa (Float32-like roots)
        const int e1 = -(int)((h >> 43) % 28ull), e2 = -(int)((h >> 48) % 28ull);
        double f1 = ldexp(m1, e1 - 1), f2 = (double)(float)ldexp(m2, e2 - 1);
        const unsigned sel = (unsigned)((h >> 53) & 63ull);
        if (sel == 0u) f1 = 1.0;
        if (sel == 1u) f2 = 1.0;
        if (sel == 2u) f1 = ldexp(1.0, e1);
        if (sel == 3u) f2 = fz_sqrt_unit_raw(1.0 - (1.0 - ldexp(1.0, -52)) * (1.0 - ldexp(1.0, -52)));  // the smallest root of 1 - v^2, |v| < 1
        const double d = f1 * f2;
        const double a = fz_div_nn(n, d), b = n / d;
        bad += __double_as_longlong(a) != __double_as_longlong(b);
    }
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(mismatches, bad);
}
}  // namespace

int fwi_selftest_div(fw_ctx *ctx, unsigned long long cases, unsigned long long seed, unsigned long long *mismatches)
{
    unsigned long long *d = nullptr;
    FW_HIP(ctx, hipMalloc((void **)&d, 8));
    FW_HIP(ctx, hipMemsetAsync(d, 0, 8, ctx->stream));
    hipLaunchKernelGGL(fz_selftest_div_kernel, dim3(4096), dim3(256), 0, ctx->stream, cases, seed, d);
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipMemcpyAsync(mismatches, d, 8, hipMemcpyDeviceToHost, ctx->stream));
    FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(d);
    return FW_OK;
}
