// FlashWeave-S without a correlation matrix: conditional Fisher-z tests straight from the sample columns
// (fw_params.recursive_pcor = 0 -- the reference's FzTestCond with an empty cor_mat: tests.jl:253 -> pcor, statfuns.jl:19-21 ->
// StatsBase.partialcor; learn_network(recursive_pcor = false), learning.jl:127,211).
//
// One wavefront = one test (X, Y | Z_1..Z_k): the k + 2 normalised columns (Float32, n x p column-major exactly as uploaded)
// are STREAMED from HBM / L2, lane l reads elements l, l + 64, ... of every column (coalesced 256-byte rows), twice:
//   pass 1  column means (Float64 partial sums per lane, DPP wave reduction);
//   pass 2  the (k+2)(k+3)/2 centred cross products (Float64 FMAs per element and lane, DPP wave reduction);
// then every lane conditions the (k+2) x (k+2) correlation matrix on Z_k, Z_{k-1}, ..., Z_1 (the unrolled recursion of
// StatsBase._partialcor, oracle/fw_oracle.c fwo_pcor), clamps and takes the Fisher-z p-value with len_z = 0 (tests.jl:256).
// Algorithmic bytes: B_fzS(k, n) = (k + 2) * n * 4 + 32 per test (variant S) -- this kernel really is
// bandwidth-bound: ~30 flops per 4 bytes at k = 3.
#include <cmath>

#include "fw_internal.h"
#include "fw_unrank.h"

#define FZS_MAXM (FW_MAX_K + 2)

namespace {

struct FzsDev {
    const float *data;  // n x p column-major
    int n, p;
    double zscale;      // sqrt(n - 3) / 2, 0 if n <= 3
    long long n_obs_min;
};

__device__ __forceinline__ double fzs_pval(double r, double zscale)
{
    // statfuns.jl:3-17; ccdf(Normal(), x) = erfc(x / sqrt2) / 2; subnormal p-values flushed (fz_pval_dev, fw_fz.hip)
    const double z = (zscale > 0.0) ? zscale * log((1.0 + r) / (1.0 - r)) : 0.0;
    const double p = (erfc(fabs(z) * 0.7071067811865476) / 2.0) * 2.0;
    return p < 2.2250738585072014e-308 ? 0.0 : p;
}

__device__ __forceinline__ double fzs_wave_sum(double v)
{
#define FZS_DPP_ADDD(ctrl, rmask)                                                                                          \
    {                                                                                                                      \
        const long long b = __double_as_longlong(v);                                                                       \
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, ctrl, rmask, 0xf, false);           \
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), ctrl, rmask, 0xf, false);   \
        v += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));                                       \
    }
    FZS_DPP_ADDD(0xb1, 0xf)
    FZS_DPP_ADDD(0x4e, 0xf)
    FZS_DPP_ADDD(0x114, 0xf)
    FZS_DPP_ADDD(0x118, 0xf)
    FZS_DPP_ADDD(0x142, 0xa)
    FZS_DPP_ADDD(0x143, 0xc)
#undef FZS_DPP_ADDD
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

struct FzsRes {
    double stat, pval;
    int power;
};

// K = compile-time bound of the conditioning-set size (register arrays); every lane returns the same result
template <int K>
__device__ __forceinline__ FzsRes fzs_test_wave(const FzsDev &P, int X, int Y, const int *zs, int k)
{
    constexpr int M = K + 2;
    FzsRes res;
    if ((long long)P.n < P.n_obs_min) {  // sufficient_power(X, Y, data, test_obj, n_obs_min), tests.jl:9-12,252
        res.stat = 0.0;
        res.pval = 1.0;
        res.power = 0;
        return res;
    }
    const int lane = threadIdx.x & 63;
    const int m = k + 2;
    const float *col[M];
    col[0] = P.data + (size_t)X * P.n;
    col[1] = P.data + (size_t)Y * P.n;
#pragma unroll
    for (int j = 0; j < K; ++j) col[2 + j] = P.data + (size_t)(j < k ? zs[j] : X) * P.n;
    // pass 1: means
    double mu[M];
#pragma unroll
    for (int a = 0; a < M; ++a) mu[a] = 0.0;
    for (int i = lane; i < P.n; i += 64) {
#pragma unroll
        for (int a = 0; a < M; ++a)
            if (a < m) mu[a] += (double)col[a][i];
    }
#pragma unroll
    for (int a = 0; a < M; ++a) mu[a] = (a < m) ? fzs_wave_sum(mu[a]) / (double)P.n : 0.0;
    // pass 2: centred cross products (upper triangle)
    double S[M][M];
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
        for (int b = 0; b < M; ++b) S[a][b] = 0.0;
    for (int i = lane; i < P.n; i += 64) {
        double d[M];
#pragma unroll
        for (int a = 0; a < M; ++a) d[a] = (a < m) ? (double)col[a][i] - mu[a] : 0.0;
#pragma unroll
        for (int a = 0; a < M; ++a)
#pragma unroll
            for (int b = a; b < M; ++b)
                if (b < m) S[a][b] = fma(d[a], d[b], S[a][b]);
    }
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
        for (int b = a; b < M; ++b)
            if (b < m) S[a][b] = fzs_wave_sum(S[a][b]);
    // pairwise correlations, then condition on Z_k, ..., Z_1
    double R[M][M];
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
        for (int b = a + 1; b < M; ++b) R[a][b] = (b < m) ? S[a][b] / sqrt(S[a][a] * S[b][b]) : 0.0;
#pragma unroll
    for (int t = M - 1; t >= 2; --t)
        if (t < m) {
#pragma unroll
            for (int a = 0; a < t; ++a)
#pragma unroll
                for (int b = a + 1; b < t; ++b)
                    R[a][b] = (R[a][b] - R[a][t] * R[b][t]) / (sqrt(1.0 - R[a][t] * R[a][t]) * sqrt(1.0 - R[b][t] * R[b][t]));
        }
    double r = R[0][1];
    if (r < -1.0) r = -1.0;  // Statistics.clampcor
    if (r > 1.0) r = 1.0;
    res.stat = r;
    res.pval = fzs_pval(r, P.zscale);
    res.power = 1;
    return res;
}

template <int K>
__global__ __launch_bounds__(256) void fzs_test_batch_kernel(FzsDev P, long long m, const int32_t *__restrict__ X,
                                                             const int32_t *__restrict__ Y, const long long *__restrict__ zoff,
                                                             const int32_t *__restrict__ zflat, fw_test_result *__restrict__ out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + wave;
    if (t >= m) return;
    const int k = __builtin_amdgcn_readfirstlane((int)(zoff[t + 1] - zoff[t]));
    int zs[K > 0 ? K : 1];
#pragma unroll
    for (int q = 0; q < K; ++q) zs[q] = __builtin_amdgcn_readfirstlane((q < k) ? zflat[zoff[t] + q] : 0);
    const FzsRes r = fzs_test_wave<K>(P, __builtin_amdgcn_readfirstlane(X[t]), __builtin_amdgcn_readfirstlane(Y[t]), zs, k);
    if (lane == 0) {
        fw_test_result o;
        o.stat = r.stat;
        o.pval = r.pval;
        o.df = 0;
        o.suff_power = r.power;
        out[t] = o;
    }
}

// test_subsets segments: 4 wavefronts per workgroup, wavefront w evaluates a run of consecutive ranks (tests.jl:281-346;
// same segment / merge protocol as the other kinds: first stop, else the (p, rank) maximum with "later wins ties")
#define FZS_RUN 4
template <int K>
__global__ __launch_bounds__(256) void fzs_subsets_seg_kernel(FzsDev P, const FwSeg *__restrict__ segs,
                                                              const int32_t *__restrict__ accflat, FwSegOut *__restrict__ out,
                                                              int max_k, double alpha, long long max_tests)
{
    __shared__ unsigned long long s_stop[4], s_br[4];
    __shared__ double s_sstat[4], s_sp[4], s_bp[4], s_bstat[4];
    __shared__ int s_spow[4];
    __shared__ unsigned int s_evc[4];
    const FwSeg seg = segs[blockIdx.x];
    const int a = seg.acc_len;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int32_t *gacc = accflat + seg.acc_off;
    const unsigned long long NONE = FW_RANK_NONE;
    double best_p = -1.0, best_stat = 0.0;
    unsigned long long best_rank = 0, evaluated = 0;
    const unsigned long long len = seg.end - seg.start;
    const int R = (int)((len + 3) / 4 < FZS_RUN ? (len + 3) / 4 : FZS_RUN);
    for (unsigned long long cbase = seg.start; cbase < seg.end; cbase += 4ull * R) {
        const unsigned long long r0 = cbase + (unsigned long long)wave * R;
        unsigned long long r1 = r0 + R;
        if (r1 > seg.end) r1 = seg.end;
        unsigned long long my_stop = NONE, my_br = 0;
        double stop_stat = 0.0, stop_p = 0.0, my_bp = -1.0, my_bstat = 0.0;
        int stop_pow = 0;
        unsigned int my_done = 0;
        if (r0 < seg.end) {
            unsigned long long rem = r0;
            int s = max_k;
            while (s > 1 && rem >= fw_binom_u64(a, s)) {
                rem -= fw_binom_u64(a, s);
                --s;
            }
            int pos[K > 0 ? K : 1];
#pragma unroll
            for (int q = 0; q < K; ++q) pos[q] = 0;
            fw_unrank_comb(rem, a, s, pos);
            for (unsigned long long r = r0; r < r1; ++r) {
                int zs[K > 0 ? K : 1];
#pragma unroll
                for (int q = 0; q < K; ++q) zs[q] = __builtin_amdgcn_readfirstlane((q < s) ? gacc[pos[q]] : 0);
                const FzsRes t = fzs_test_wave<K>(P, seg.X, seg.Y, zs, s);
                ++my_done;
                const bool sig = (t.pval < alpha) && t.power;
                if (!sig || (max_tests > 0 && r + 1 >= (unsigned long long)max_tests)) {
                    my_stop = r;
                    stop_stat = t.stat;
                    stop_p = t.pval;
                    stop_pow = t.power;
                    break;
                }
                if (t.pval >= my_bp) {
                    my_bp = t.pval;
                    my_br = r;
                    my_bstat = t.stat;
                }
                int i = s - 1;
                while (i >= 0 && pos[i] == a - s + i) --i;
                if (i < 0) {
                    --s;
#pragma unroll
                    for (int q = 0; q < K; ++q) pos[q] = q;
                    if (s < 1) break;
                } else {
                    ++pos[i];
                    for (int j = i + 1; j < s; ++j) pos[j] = pos[j - 1] + 1;
                }
            }
        }
        if (lane == 0) {
            s_evc[wave] = my_done;
            s_stop[wave] = my_stop;
            s_sstat[wave] = stop_stat;
            s_sp[wave] = stop_p;
            s_spow[wave] = stop_pow;
            s_bp[wave] = my_bp;
            s_br[wave] = my_br;
            s_bstat[wave] = my_bstat;
        }
        __syncthreads();
        evaluated += (unsigned long long)(s_evc[0] + s_evc[1] + s_evc[2] + s_evc[3]);
        int fw = -1;
        unsigned long long first = NONE;
#pragma unroll
        for (int w = 0; w < 4; ++w)
            if (s_stop[w] < first) {
                first = s_stop[w];
                fw = w;
            }
        if (fw >= 0) {
            if (threadIdx.x == 0) {
                FwSegOut o;
                o.stop_rank = first;
                o.stop_stat = s_sstat[fw];
                o.stop_pval = s_sp[fw];
                o.best_rank = 0;
                o.best_stat = 0.0;
                o.best_pval = -1.0;
                o.stop_df = 0;
                o.stop_power = s_spow[fw];
                o.best_df = 0;
                o.pad = 0;
                o.evaluated = evaluated;
                out[blockIdx.x] = o;
            }
            return;
        }
#pragma unroll
        for (int w = 0; w < 4; ++w)  // waves hold increasing ranks: sequential `>=` merge (tests.jl:338)
            if (s_bp[w] >= 0.0 && s_bp[w] >= best_p) {
                best_p = s_bp[w];
                best_stat = s_bstat[w];
                best_rank = s_br[w];
            }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        FwSegOut o;
        o.stop_rank = NONE;
        o.stop_stat = 0.0;
        o.stop_pval = 0.0;
        o.best_rank = best_rank;
        o.best_stat = best_stat;
        o.best_pval = best_p;
        o.stop_df = 0;
        o.stop_power = 1;
        o.best_df = 0;
        o.pad = 0;
        o.evaluated = evaluated;
        out[blockIdx.x] = o;
    }
}

FzsDev fzs_dev(const fw_ctx *ctx)
{
    FzsDev P;
    P.data = ctx->d_data;
    P.n = ctx->P.n;
    P.p = ctx->P.p;
    P.zscale = ctx->P.n > 3 ? std::sqrt((double)(ctx->P.n - 3)) / 2.0 : 0.0;
    P.n_obs_min = ctx->n_obs_min_eff;
    return P;
}

}  // namespace

int fwi_fzs_test_batch(fw_ctx *ctx, int64_t m, const int32_t *X, const int32_t *Y, const int64_t *zoff, const int32_t *zflat,
                       fw_test_result *out)
{
    if (m == 0) return FW_OK;
    if (!ctx->d_data) return fw_fail(ctx, FW_ERR_STATE, "recursive_pcor = 0 needs the data matrix on the device (fw_set_data_dense_f32)");
    const int64_t nz = zoff[m];
    int kmax = 0;
    for (int64_t t = 0; t < m; ++t) kmax = std::max<int>(kmax, (int)(zoff[t + 1] - zoff[t]));
    int rc;
    if ((rc = fw_dev_reserve(ctx, ctx->d_jobs, (size_t)m * 2 * sizeof(int32_t) + (size_t)(m + 1) * sizeof(int64_t)))) return rc;
    if ((rc = fw_dev_reserve(ctx, ctx->d_acc, (size_t)(nz > 0 ? nz : 1) * sizeof(int32_t)))) return rc;
    if ((rc = fw_dev_reserve(ctx, ctx->d_out, (size_t)m * sizeof(fw_test_result)))) return rc;
    long long *dz = (long long *)ctx->d_jobs.ptr;
    int32_t *dX = (int32_t *)(dz + m + 1), *dY = dX + m;
    FW_HIP(ctx, hipMemcpyAsync(dz, zoff, (size_t)(m + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    FW_HIP(ctx, hipMemcpyAsync(dX, X, (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    FW_HIP(ctx, hipMemcpyAsync(dY, Y, (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    if (nz > 0) FW_HIP(ctx, hipMemcpyAsync(ctx->d_acc.ptr, zflat, (size_t)nz * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    const dim3 grid((unsigned)((m + 3) / 4));
    const FzsDev P = fzs_dev(ctx);
    if (kmax <= 3)
        hipLaunchKernelGGL((fzs_test_batch_kernel<3>), grid, dim3(256), 0, ctx->stream, P, (long long)m, dX, dY, dz,
                           (const int32_t *)ctx->d_acc.ptr, (fw_test_result *)ctx->d_out.ptr);
    else
        hipLaunchKernelGGL((fzs_test_batch_kernel<FW_MAX_K>), grid, dim3(256), 0, ctx->stream, P, (long long)m, dX, dY, dz,
                           (const int32_t *)ctx->d_acc.ptr, (fw_test_result *)ctx->d_out.ptr);
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipMemcpyAsync(out, ctx->d_out.ptr, (size_t)m * sizeof(fw_test_result), hipMemcpyDeviceToHost, ctx->stream));
    FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cnt.kernel_launches += 1;
    return FW_OK;
}

int fwi_fzs_segments(fw_ctx *ctx, int64_t nseg, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out, FwPoolBuf &pb)
{
    if (nseg == 0) return FW_OK;
    if (!ctx->d_data) return fw_fail(ctx, FW_ERR_STATE, "recursive_pcor = 0 needs the data matrix on the device (fw_set_data_dense_f32)");
    FW_HIP(ctx, hipEventRecord(pb.ev0, pb.launch_stream));
    const FzsDev P = fzs_dev(ctx);
    if (ctx->P.max_k <= 3)
        hipLaunchKernelGGL((fzs_subsets_seg_kernel<3>), dim3((unsigned)nseg), dim3(256), 0, pb.launch_stream, P, d_segs, d_acc, d_out,
                           ctx->P.max_k, ctx->P.alpha, (long long)ctx->P.max_tests);
    else
        hipLaunchKernelGGL((fzs_subsets_seg_kernel<FW_MAX_K>), dim3((unsigned)nseg), dim3(256), 0, pb.launch_stream, P, d_segs, d_acc,
                           d_out, ctx->P.max_k, ctx->P.alpha, (long long)ctx->P.max_tests);
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipEventRecord(pb.ev1, pb.launch_stream));
    return FW_OK;
}
