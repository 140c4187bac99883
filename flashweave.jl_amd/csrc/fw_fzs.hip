// FlashWeave-S without a correlation matrix: conditional Fisher-z tests straight from the sample columns
// (fw_params.recursive_pcor = 0 -- the reference's FzTestCond with an empty cor_mat: tests.jl:253 -> pcor, statfuns.jl:19-21 ->
// StatsBase.partialcor; learn_network(recursive_pcor = false), learning.jl:127,211).
//
// One wavefront = one test (X, Y | Z_1..Z_k): the normalised columns (Float32, n x p column-major exactly as uploaded) are STREAMED
// from HBM / L2 in ONE pass of 16-byte loads (lane l holds samples 4l .. 4l+3 of every 256-sample row).  r03 rewrite of the r02
// kernel (two passes of 4-byte loads, 213-244 VGPRs, X and Y re-read for every subset of a job):
//   * column means and sums of squared deviations (Float64) are computed ONCE per data upload (fzs_colstat_kernel), so a test
//     needs only the cross products, and  sum (x_a - mu_a)(x_b - mu_b) = sum x_a (x_b - mu_b)  (the dropped term is mu_a times the
//     rounding residue of a centred sum: < 1e-14 relative) -- one conversion + one FMA per element and pair;
//   * inside a test_subsets job the wavefront keeps the X and Y columns in registers (n <= 2048: 2 x 8 float4 per lane) and their
//     cross product for all the subsets it evaluates: a test streams its k conditioning columns only;
//   * then every lane conditions the (k+2) x (k+2) correlation matrix on Z_k, ..., Z_1 (the unrolled recursion of
//     StatsBase._partialcor, oracle/fw_oracle.c fwo_pcor), clamps and takes the Fisher-z p-value with len_z = 0 (tests.jl:256).
// Algorithmic bytes: B_fzS(k, n) = (k + 2) * n * 4 + 32 per test (variant S): what a test that shares nothing
// with its neighbours streams.  With X / Y held, the bytes really requested are k * n * 4 per test.
#include <cmath>

#include "fw_internal.h"
#include "fw_unrank.h"

#define FZS_MAXM (FW_MAX_K + 2)

namespace {

struct FzsDev {
    const float *data;  // n x p column-major
    const double *st;   // per column {mean, sum of squared deviations}
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

// per column: {mean, sum of squared deviations} in Float64, one wavefront per column (two passes over the column: it is read from
// L2 the second time).  Once per data upload.
__global__ __launch_bounds__(256) void fzs_colstat_kernel(const float *__restrict__ data, int n, int p, double *__restrict__ st)
{
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (col >= p) return;
    const float *c = data + (size_t)col * n;
    double s = 0.0;
    for (int i = lane; i < n; i += 64) s += (double)c[i];
    const double mu = fzs_wave_sum(s) / (double)n;
    double q = 0.0;
    for (int i = lane; i < n; i += 64) {
        const double d = (double)c[i] - mu;
        q = fma(d, d, q);
    }
    q = fzs_wave_sum(q);
    if (lane == 0) {
        st[2 * (size_t)col] = mu;
        st[2 * (size_t)col + 1] = q;
    }
}

// X and Y of a job, as the wavefront holds them across its subsets.  T > 0: float4 registers (n % 4 == 0, n <= 256 T);
// T == 0: nothing is held, the columns are streamed with every test (any n)
template <int T>
struct FzsXY {
    float4 x[T > 0 ? T : 1], y[T > 0 ? T : 1];
    double sxy;            // sum x (y - mu_y)
    const float *cx, *cy;  // the columns
    double mux, muy, ssx, ssy;
};

template <int T>
__device__ __forceinline__ void fzs_load_xy(const FzsDev &P, int X, int Y, FzsXY<T> &H)
{
    const int lane = threadIdx.x & 63;
    H.cx = P.data + (size_t)X * P.n;
    H.cy = P.data + (size_t)Y * P.n;
    H.mux = P.st[2 * (size_t)X];
    H.ssx = P.st[2 * (size_t)X + 1];
    H.muy = P.st[2 * (size_t)Y];
    H.ssy = P.st[2 * (size_t)Y + 1];
    double s = 0.0;
    if (T > 0) {
        const int n4 = P.n >> 2;
#pragma unroll
        for (int t = 0; t < (T > 0 ? T : 1); ++t) {
            const int q = lane + 64 * t;
            const bool in = q < n4;
            const float4 z0 = make_float4(0.f, 0.f, 0.f, 0.f);
            H.x[t] = in ? ((const float4 *)H.cx)[q] : z0;
            H.y[t] = in ? ((const float4 *)H.cy)[q] : z0;
            if (in) {  // (masked lanes: x = 0 contributes nothing whatever the deviation of y)
                s = fma((double)H.x[t].x, (double)H.y[t].x - H.muy, s);
                s = fma((double)H.x[t].y, (double)H.y[t].y - H.muy, s);
                s = fma((double)H.x[t].z, (double)H.y[t].z - H.muy, s);
                s = fma((double)H.x[t].w, (double)H.y[t].w - H.muy, s);
            }
        }
    } else {
        for (int i = lane; i < P.n; i += 64) s = fma((double)H.cx[i], (double)H.cy[i] - H.muy, s);
    }
    H.sxy = fzs_wave_sum(s);
}

// K = compile-time bound of the conditioning-set size (register arrays); every lane returns the same result
template <int K, int T>
__device__ __forceinline__ FzsRes fzs_test_wave(const FzsDev &P, const FzsXY<T> &H, const int *zs, int k)
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
    const float *col[K > 0 ? K : 1];
    double mu[K > 0 ? K : 1], ss[M];
    ss[0] = H.ssx;
    ss[1] = H.ssy;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int v = j < k ? zs[j] : zs[0];
        col[j] = P.data + (size_t)v * P.n;
        mu[j] = P.st[2 * (size_t)v];
        ss[2 + j] = P.st[2 * (size_t)v + 1];
    }
    // cross products S[a][b], a < b, b >= 2:  sum x_a (z_b - mu_b)   (S[0][1] is the job's)
    double S[M][M];
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
        for (int b = 0; b < M; ++b) S[a][b] = 0.0;
#define FZS_ACC(xv, yv, zsel)                                                             \
    {                                                                                     \
        double zr[K > 0 ? K : 1], zd[K > 0 ? K : 1];                                      \
        _Pragma("unroll") for (int j = 0; j < K; ++j)                                    \
        {                                                                                 \
            zr[j] = (double)(zsel(j));                                                    \
            zd[j] = zr[j] - mu[j];                                                        \
        }                                                                                 \
        const double xd = (double)(xv), yd = (double)(yv);                                \
        _Pragma("unroll") for (int j = 0; j < K; ++j) if (j < k)                         \
        {                                                                                 \
            S[0][2 + j] = fma(xd, zd[j], S[0][2 + j]);                                    \
            S[1][2 + j] = fma(yd, zd[j], S[1][2 + j]);                                    \
            _Pragma("unroll") for (int i = 0; i < K; ++i) if (i < j) S[2 + i][2 + j] = fma(zr[i], zd[j], S[2 + i][2 + j]); \
        }                                                                                 \
    }
    if (T > 0) {
        const int n4 = P.n >> 2;
#pragma unroll
        for (int t = 0; t < (T > 0 ? T : 1); ++t) {
            const int q = lane + 64 * t;
            if (q < n4) {
                float4 z4[K > 0 ? K : 1];
#pragma unroll
                for (int j = 0; j < K; ++j) z4[j] = ((const float4 *)col[j])[q];  // all loads of the row in flight together
#define ZX(j) z4[j].x
#define ZY(j) z4[j].y
#define ZZ(j) z4[j].z
#define ZW(j) z4[j].w
                FZS_ACC(H.x[t].x, H.y[t].x, ZX)
                FZS_ACC(H.x[t].y, H.y[t].y, ZY)
                FZS_ACC(H.x[t].z, H.y[t].z, ZZ)
                FZS_ACC(H.x[t].w, H.y[t].w, ZW)
#undef ZX
#undef ZY
#undef ZZ
#undef ZW
            }
        }
    } else {
        for (int i = lane; i < P.n; i += 64) {
            float zf[K > 0 ? K : 1];
#pragma unroll
            for (int j = 0; j < K; ++j) zf[j] = col[j][i];
#define ZS(j) zf[j]
            FZS_ACC(H.cx[i], H.cy[i], ZS)
#undef ZS
        }
    }
#undef FZS_ACC
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
        for (int b = a + 1; b < M; ++b)
            if (b >= 2 && b < m) S[a][b] = fzs_wave_sum(S[a][b]);
    S[0][1] = H.sxy;
    // pairwise correlations, then condition on Z_k, ..., Z_1
    double R[M][M];
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
        for (int b = a + 1; b < M; ++b) R[a][b] = (b < m) ? S[a][b] / sqrt(ss[a] * ss[b]) : 0.0;
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

template <int K, int T>
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
    FzsXY<T> H;
    fzs_load_xy<T>(P, __builtin_amdgcn_readfirstlane(X[t]), __builtin_amdgcn_readfirstlane(Y[t]), H);
    const FzsRes r = fzs_test_wave<K, T>(P, H, zs, k);
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
#define FZS_RUN 16  // (r02: 4; the wavefront now pays for X and Y once per run)
template <int K, int T>
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
    FzsXY<T> H;
    bool xy_loaded = false;
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
            if (!xy_loaded) {  // X and Y of the job: once per wavefront and segment
                fzs_load_xy<T>(P, seg.X, seg.Y, H);
                xy_loaded = true;
            }
            for (unsigned long long r = r0; r < r1; ++r) {
                int zs[K > 0 ? K : 1];
#pragma unroll
                for (int q = 0; q < K; ++q) zs[q] = __builtin_amdgcn_readfirstlane((q < s) ? gacc[pos[q]] : 0);
                const FzsRes t = fzs_test_wave<K, T>(P, H, zs, s);
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
    P.st = ctx->d_fzs_stat;
    P.n = ctx->P.n;
    P.p = ctx->P.p;
    P.zscale = ctx->P.n > 3 ? std::sqrt((double)(ctx->P.n - 3)) / 2.0 : 0.0;
    P.n_obs_min = ctx->n_obs_min_eff;
    return P;
}

// column means / sums of squared deviations: once per data upload
int fzs_ensure_stat(fw_ctx *ctx, hipStream_t st)
{
    if (ctx->have_fzs_stat) return FW_OK;
    if (!ctx->d_fzs_stat) FW_HIP(ctx, hipMalloc((void **)&ctx->d_fzs_stat, sizeof(double) * 2 * (size_t)ctx->P.p));
    hipLaunchKernelGGL(fzs_colstat_kernel, dim3((unsigned)((ctx->P.p + 3) / 4)), dim3(256), 0, st, (const float *)ctx->d_data, ctx->P.n, ctx->P.p,
                       ctx->d_fzs_stat);
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipStreamSynchronize(st));  // (other streams use it next)
    ctx->have_fzs_stat = true;
    return FW_OK;
}

// X / Y columns in registers: n a multiple of 4 (16-byte rows) and at most 2048 samples (8 float4 per lane and column)
bool fzs_hold_xy(const fw_ctx *ctx) { return ctx->P.n % 4 == 0 && ctx->P.n <= 2048; }

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
    if ((rc = fzs_ensure_stat(ctx, ctx->stream))) return rc;
    const FzsDev P = fzs_dev(ctx);
#define FZS_BATCH(KK, TT)                                                                                                    \
    hipLaunchKernelGGL((fzs_test_batch_kernel<KK, TT>), grid, dim3(256), 0, ctx->stream, P, (long long)m, dX, dY, dz,        \
                       (const int32_t *)ctx->d_acc.ptr, (fw_test_result *)ctx->d_out.ptr)
    if (fzs_hold_xy(ctx)) {
        if (kmax <= 3) FZS_BATCH(3, 8); else FZS_BATCH(FW_MAX_K, 8);
    } else {
        if (kmax <= 3) FZS_BATCH(3, 0); else FZS_BATCH(FW_MAX_K, 0);
    }
#undef FZS_BATCH
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
    if (int rc = fzs_ensure_stat(ctx, pb.launch_stream)) return rc;
    FW_HIP(ctx, hipEventRecord(pb.ev0, pb.launch_stream));
    const FzsDev P = fzs_dev(ctx);
#define FZS_SEG(KK, TT)                                                                                                               \
    hipLaunchKernelGGL((fzs_subsets_seg_kernel<KK, TT>), dim3((unsigned)nseg), dim3(256), 0, pb.launch_stream, P, d_segs, d_acc, d_out, \
                       ctx->P.max_k, ctx->P.alpha, (long long)ctx->P.max_tests)
    if (fzs_hold_xy(ctx)) {
        if (ctx->P.max_k <= 3) FZS_SEG(3, 8); else FZS_SEG(FW_MAX_K, 8);
    } else {
        if (ctx->P.max_k <= 3) FZS_SEG(3, 0); else FZS_SEG(FW_MAX_K, 0);
    }
#undef FZS_SEG
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipEventRecord(pb.ev1, pb.launch_stream));
    return FW_OK;
}
