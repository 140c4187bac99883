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
//   * inside a test_subsets job the workgroup keeps the X and Y columns in LDS (n <= 2048: 16 KB) and every wavefront their cross
//     product for all the subsets it evaluates: a test streams its k conditioning columns only;
//   * the (k+2) x (k+2) correlation matrix is then conditioned on Z_k, ..., Z_1 (the unrolled recursion of StatsBase._partialcor,
//     oracle/fw_oracle.c fwo_pcor), clamped, and the Fisher-z p-value taken with len_z = 0 (tests.jl:256) -- for up to four tests
//     of a wavefront at once, one lane per matrix entry (fzs_finish below);
//   * bookkeeping of a run (ranks, positions, best / stop records) is wave-uniform and lives in scalar registers.
// 128 VGPRs, four workgroups per CU.  History on the micro-benchmark (|accepted| = 40, n = 2000; tests/s in the kernel): r02 form
// not timed; one pass + X / Y in registers 1.36e8; X / Y in LDS + arithmetic out of line 1.79e8; lane-parallel arithmetic 2.72e8;
// four tests finished together 3.67e8 (= 1.8 x the 8 TB/s nominal B_fzS rate: X / Y come from LDS and the Z columns of a job from L2).
// Algorithmic bytes: B_fzS(k, n) = (k + 2) * n * 4 + 32 per test (variant S): what a test that shares nothing
// with its neighbours streams.  With X / Y held, the bytes really requested are k * n * 4 per test.
#include <cmath>

#include "fw_internal.h"
#include "fw_unrank.h"

#define FZS_MAXM (FW_MAX_K_FAST + 2)

namespace {

struct FzsDev {
    const float *data;  // n x p column-major
    const double *st;   // per column {mean, sum of squared deviations}
    int n, p;
    double zscale;      // sqrt(n - 3) / 2, 0 if n <= 3
    long long n_obs_min;
    int nzjobs;         // fz_nz without a matrix (r04): a job's matrix is the correlation of its ROW VIEW (fznz_submat_kernel, Float64), its
                        // sample size and z scale come from its record (GRAM kernels only)
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

// a value every lane holds -> scalar registers (the bookkeeping around the tests then stays out of the vector file)
__device__ __forceinline__ double fzs_uniform(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

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

// X and Y of a job.  T > 0 (n % 4 == 0, n <= 256 T): the workgroup keeps both columns in LDS for all the subsets its four
// wavefronts evaluate (16 KB at n = 2000; the first r03 version held them in 64 registers per lane, which together with the
// ~150 registers of the then sequential arithmetic left one wavefront per SIMD and 700 B of scratch); T == 0: streamed with every test.
template <int T>
struct FzsXY {
    const float4 *lx, *ly;  // LDS copies (T > 0)
    double sxy;             // sum x (y - mu_y)
    const float *cx, *cy;   // the columns in global memory
    double mux, muy, ssx, ssy;
    int vx, vy;             // the variables
};

#define FZS_LDS_N 2048  // samples per column the LDS copy holds

// wavefront-level: statistics + the X-Y cross product; lds != nullptr: the columns are already there (filled by the workgroup)
template <int T>
__device__ __forceinline__ void fzs_load_xy(const FzsDev &P, int X, int Y, FzsXY<T> &H, const float *lds)
{
    const int lane = threadIdx.x & 63;
    H.vx = X;
    H.vy = Y;
    H.cx = P.data + (size_t)X * P.n;
    H.cy = P.data + (size_t)Y * P.n;
    H.mux = P.st[2 * (size_t)X];
    H.ssx = P.st[2 * (size_t)X + 1];
    H.muy = P.st[2 * (size_t)Y];
    H.ssy = P.st[2 * (size_t)Y + 1];
    H.lx = (const float4 *)lds;
    H.ly = (const float4 *)(lds + FZS_LDS_N);
    double s = 0.0;
    if (T > 0) {
        const int n4 = P.n >> 2;
        for (int q = lane; q < n4; q += 64) {
            const float4 x = H.lx[q], y = H.ly[q];
            s = fma((double)x.x, (double)y.x - H.muy, s);
            s = fma((double)x.y, (double)y.y - H.muy, s);
            s = fma((double)x.z, (double)y.z - H.muy, s);
            s = fma((double)x.w, (double)y.w - H.muy, s);
        }
    } else {
        for (int i = lane; i < P.n; i += 64) s = fma((double)H.cx[i], (double)H.cy[i] - H.muy, s);
    }
    H.sxy = fzs_uniform(fzs_wave_sum(s));
}

// ---- from the sums to the p-value ---------------------------------------------------------------------------------------------
// pairwise correlations from the cross products, conditioned on Z_k, ..., Z_1 (StatsBase._partialcor unrolled), clampcor, p-value.
// A wavefront first STREAMS up to G tests (fzs_stream: sums into the test's LDS slot), then FINISHES them together (fzs_finish):
// one lane per pair (a, b) of a test, LPT lanes per test.  The M (M - 1) / 2 correlations of a conditioning step are independent,
// so the lanes of a test update them side by side (K = 3: 3 steps of one division + two square roots instead of 10 such updates one
// after the other on every lane), the pivot column comes out of LDS, the working set is one register instead of ~150, and the
// log / erfc sequence of the p-value is paid once per G tests.  Same formula and order per element as the sequential form (the
// first r03 version: 1.8e8 tests/s; this one: see profiles/r03_fzs_micro.json).
#ifndef FZS_GMAX
#define FZS_GMAX 4
#endif
template <int K>
struct FzsLay {
    static constexpr int M = K + 2, PAIRS = M * (M - 1) / 2;
    static constexpr int LPT = PAIRS <= 4 ? 4 : (PAIRS <= 16 ? 16 : 32);           // lanes per test
    static constexpr int G = (64 / LPT) < FZS_GMAX ? (64 / LPT) : FZS_GMAX;        // tests finished together
    // slot of a test (doubles): centred cross products s[a * M + b] (a < b), sums of squared deviations, working correlations,
    // {statistic, p-value}.  (By value through the stack the sums were 15 KB of scratch written and read back per test: 266 GB of
    // HBM writes in the micro-benchmark, a third of the algorithmic bytes.)
    static constexpr int SS = M * M, WORK = M * M + M, RES = 2 * M * M + M, QD = 2 * M * M + M + 2;
};
#define FZS_WAVE_SYNC()                                        \
    {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
    }

template <int K, bool GRAM>
__device__ __noinline__ void fzs_finish(double *q, const int *mm, int ng, double zscale, const double *cjob, int mj, const int *idx)
{
    using L = FzsLay<K>;
    constexpr int M = L::M;
    const int lane = threadIdx.x & 63;
    const int g = lane / L::LPT, pi = lane % L::LPT;
    int la = 0, off = pi;  // pair index -> (a, b), a < b, row by row
#pragma unroll
    for (int i = 0; i < M - 1; ++i)
        if (la == i && off >= M - 1 - i) {
            off -= M - 1 - i;
            la = i + 1;
        }
    const int lb = la + 1 + off;
    const bool live = g < ng && g < L::G && pi < L::PAIRS;
    double *qs = q + (live ? g : 0) * L::QD;
    const int m = live ? mm[g] : 0;
    int mmax = 2;
    for (int i = 0; i < ng; ++i) mmax = mm[i] > mmax ? mm[i] : mmax;
    mmax = __builtin_amdgcn_readfirstlane(mmax);
    const bool pair = live && lb < m;
    // GRAM: the pair's correlation is an entry of the job's matrix (fzs_gram_kernel; LDS or L2), addressed by the job-local indices
    // of the test's variables (0 = X, 1 = Y, 2 + position in the accepted list)
    double R = 0.0;
    if (GRAM) {
        if (pair) R = cjob[idx[(live ? g : 0) * M + la] * mj + idx[(live ? g : 0) * M + lb]];
    } else {
        R = pair ? qs[la * M + lb] / sqrt(qs[L::SS + la] * qs[L::SS + lb]) : 0.0;
    }
    for (int t = mmax - 1; t >= 2; --t) {
        if (pair) qs[L::WORK + la * M + lb] = R;
        FZS_WAVE_SYNC()
        if (pair && lb < t && t < m) {
            const double rat = qs[L::WORK + la * M + t], rbt = qs[L::WORK + lb * M + t];
            R = (R - rat * rbt) / (sqrt(1.0 - rat * rat) * sqrt(1.0 - rbt * rbt));
        }
        FZS_WAVE_SYNC()
    }
    if (R < -1.0) R = -1.0;  // Statistics.clampcor
    if (R > 1.0) R = 1.0;
    const double pv = fzs_pval(R, zscale);
    if (live && pi == 0) {  // pair (0, 1)
        qs[L::RES] = R;
        qs[L::RES + 1] = pv;
    }
    FZS_WAVE_SYNC()
}

// K = compile-time bound of the conditioning-set size (register arrays); the sums of the test land in its LDS slot qs
template <int K, int T>
__device__ __forceinline__ void fzs_stream(const FzsDev &P, const FzsXY<T> &H, const int *zs, int k, double *qs)
{
    using L = FzsLay<K>;
    constexpr int M = K + 2;
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
#pragma unroll 2
        for (int q = lane; q < n4; q += 64) {
            float4 z4[K > 0 ? K : 1];
#pragma unroll
            for (int j = 0; j < K; ++j) z4[j] = ((const float4 *)col[j])[q];  // the k loads of a row in flight together
            const float4 x4 = H.lx[q], y4 = H.ly[q];
#define ZX(j) z4[j].x
#define ZY(j) z4[j].y
#define ZZ(j) z4[j].z
#define ZW(j) z4[j].w
            FZS_ACC(x4.x, y4.x, ZX)
            FZS_ACC(x4.y, y4.y, ZY)
            FZS_ACC(x4.z, y4.z, ZZ)
            FZS_ACC(x4.w, y4.w, ZW)
#undef ZX
#undef ZY
#undef ZZ
#undef ZW
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
    {   // the same variable twice in a test (feed-forward whitelists can repeat a member of the elimination pool, hiton.jl:24-26):
        // its cross product with itself is its sum of squares exactly -> correlation exactly 1, as StatsBase's identical sums give
        int id[M];
        id[0] = H.vx;
        id[1] = H.vy;
#pragma unroll
        for (int j = 0; j < K; ++j) id[2 + j] = j < k ? zs[j] : -1 - j;
#pragma unroll
        for (int a = 0; a < M; ++a)
#pragma unroll
            for (int b = a + 1; b < M; ++b)
                if (id[a] == id[b]) S[a][b] = ss[a];
    }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < M; ++a) {
            qs[L::SS + a] = ss[a];
#pragma unroll
            for (int b = a + 1; b < M; ++b) qs[a * M + b] = S[a][b];
        }
    }
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
    // explicit tests: every wavefront has its own X and Y -> its own LDS slot (T > 0)
    __shared__ __attribute__((aligned(16))) float s_xy[T > 0 ? 4 : 1][T > 0 ? 2 * FZS_LDS_N : 4];
    const int Xu = __builtin_amdgcn_readfirstlane(X[t]), Yu = __builtin_amdgcn_readfirstlane(Y[t]);
    if (T > 0) {
        const int n4 = P.n >> 2;
        float4 *dx = (float4 *)s_xy[wave], *dy = (float4 *)(s_xy[wave] + FZS_LDS_N);
        const float4 *gx = (const float4 *)(P.data + (size_t)Xu * P.n), *gy = (const float4 *)(P.data + (size_t)Yu * P.n);
        for (int q = lane; q < n4; q += 64) {
            dx[q] = gx[q];
            dy[q] = gy[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    FzsXY<T> H;
    fzs_load_xy<T>(P, Xu, Yu, H, s_xy[T > 0 ? wave : 0]);
    __shared__ double s_q[4][FzsLay<K>::QD];
    __shared__ int s_m[4][FZS_GMAX];
    const bool power = (long long)P.n >= P.n_obs_min;  // sufficient_power(X, Y, data, test_obj, n_obs_min), tests.jl:9-12,252
    if (power) {
        fzs_stream<K, T>(P, H, zs, k, s_q[wave]);
        if (lane == 0) s_m[wave][0] = k + 2;
        FZS_WAVE_SYNC()
        fzs_finish<K, false>(s_q[wave], s_m[wave], 1, P.zscale, nullptr, 0, nullptr);
    }
    if (lane == 0) {
        fw_test_result o;
        o.stat = power ? s_q[wave][FzsLay<K>::RES] : 0.0;
        o.pval = power ? s_q[wave][FzsLay<K>::RES + 1] : 1.0;
        o.df = 0;
        o.suff_power = power ? 1 : 0;
        out[t] = o;
    }
}

// test_subsets segments: 4 wavefronts per workgroup, wavefront w evaluates a run of consecutive ranks (tests.jl:281-346;
// same segment / merge protocol as the other kinds: first stop, else the (p, rank) maximum with "later wins ties")
#define FZS_RUN 16  // (r02: 4; the wavefront now pays for X and Y once per run)
#ifndef FZS_SEG_OCC
#define FZS_SEG_OCC 4  // workgroups per CU (= waves per SIMD) the segment kernel is compiled for; 3 / 4 / 5: 3.31 / 3.67 / 1.65 e8 tests/s (5: spills inside the loop)
#endif
// GRAM: the correlations of a job are not streamed per test but read from the job's (a + 2) x (a + 2) Float64 matrix, computed once
// per job and pool round by fzs_gram_kernel (see there); the workgroup stages it in LDS (dynamic, lds_m <= 80 variables) when it fits.
template <int K, int T, bool GRAM>
__global__ __launch_bounds__(256, FZS_SEG_OCC) void fzs_subsets_seg_kernel(FzsDev P, const FwSeg *__restrict__ segs,
                                                              const int32_t *__restrict__ accflat, FwSegOut *__restrict__ out,
                                                              int max_k, double alpha, long long max_tests,
                                                              const FwNzJob *__restrict__ recs, const double *__restrict__ arena, int lds_m)
{
    extern __shared__ double s_cjob[];
    __shared__ unsigned long long s_stop[4], s_br[4];
    __shared__ double s_sstat[4], s_sp[4], s_bp[4], s_bstat[4];
    __shared__ int s_spow[4];
    __shared__ unsigned int s_evc[4];
    const FwSeg seg = segs[blockIdx.x];
    const int a = seg.acc_len;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int32_t *gacc = accflat + seg.acc_off;
    const unsigned long long NONE = FW_RANK_NONE;
    double best_p = -1.0, best_stat = 0.0;
    unsigned long long best_rank = 0, evaluated = 0;
    const unsigned long long len = seg.end - seg.start;
    const int R = (int)((len + 3) / 4 < FZS_RUN ? (len + 3) / 4 : FZS_RUN);
    // X and Y of the segment's job: one LDS copy for the four wavefronts
    __shared__ __attribute__((aligned(16))) float s_xy[T > 0 ? 2 * FZS_LDS_N : 4];
    using L = FzsLay<K>;
    __shared__ double s_q[4][L::G * L::QD];
    __shared__ int s_m[4][FZS_GMAX];
    __shared__ int s_idx[GRAM ? 4 : 1][GRAM ? FZS_GMAX : 1][GRAM ? K + 2 : 1];
    const double *cjob = nullptr;
    int mjob = 0;
    double zs_job = P.zscale;
    bool power = (long long)P.n >= P.n_obs_min;  // sufficient_power(X, Y, data, test_obj, n_obs_min), tests.jl:9-12,252
    if (GRAM) {
        const FwNzJob rec = recs[seg.pad];
        if (P.nzjobs) {  // the data of the job is its row view: size(data, 1) = rec.nR (tests.jl:293-296, 252-256)
            if ((long long)rec.nR < P.n_obs_min) {  // (0, 1, 0, false) with zero tests
                if (threadIdx.x == 0) {
                    FwSegOut o;
                    o.stop_rank = 0;
                    o.stop_stat = 0.0;
                    o.stop_pval = 1.0;
                    o.best_rank = 0;
                    o.best_stat = 0.0;
                    o.best_pval = -1.0;
                    o.stop_df = -2;  // marker: no test was executed (fwi_pool_collect)
                    o.stop_power = 0;
                    o.best_df = 0;
                    o.pad = 0;
                    o.evaluated = 0;
                    out[blockIdx.x] = o;
                }
                return;
            }
            zs_job = rec.zscale;
            power = true;
        }
        mjob = rec.m;
        cjob = arena + rec.cor_off;
        if (mjob <= lds_m) {
            for (int i = threadIdx.x; i < mjob * mjob; i += 256) s_cjob[i] = cjob[i];
            __syncthreads();
            cjob = s_cjob;
        }
    }
    if (T > 0) {
        const int n4 = P.n >> 2;
        const float4 *gx = (const float4 *)(P.data + (size_t)seg.X * P.n), *gy = (const float4 *)(P.data + (size_t)seg.Y * P.n);
        for (int q = threadIdx.x; q < n4; q += 256) {
            ((float4 *)s_xy)[q] = gx[q];
            ((float4 *)(s_xy + FZS_LDS_N))[q] = gy[q];
        }
        __syncthreads();
    }
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
            if (!GRAM && !xy_loaded) {  // X and Y of the job: once per wavefront and segment
                fzs_load_xy<T>(P, seg.X, seg.Y, H, s_xy);
                xy_loaded = true;
            }
            for (unsigned long long r = r0; r < r1;) {
                // stream up to G consecutive ranks, finish them together
                int ng = 0;
                unsigned long long rg = r;
                while (ng < L::G && rg < r1 && s >= 1) {
                    if (power && GRAM) {
                        if (lane < K + 2) {
                            int v = lane;  // job-local index: 0 = X, 1 = Y, 2 + position
#pragma unroll
                            for (int q = 0; q < K; ++q)
                                if (lane == 2 + q) v = q < s ? 2 + pos[q] : 0;
                            s_idx[wave][ng][lane] = v;
                        }
                        if (lane == 0) s_m[wave][ng] = s + 2;
                    } else if (power) {
                        int zs[K > 0 ? K : 1];
#pragma unroll
                        for (int q = 0; q < K; ++q) zs[q] = __builtin_amdgcn_readfirstlane((q < s) ? gacc[pos[q]] : 0);
                        fzs_stream<K, T>(P, H, zs, s, s_q[wave] + ng * L::QD);  // (s uniform: scalar branches on k)
                        if (lane == 0) s_m[wave][ng] = s + 2;
                    }
                    ++ng;
                    ++rg;
                    int i = s - 1;  // next subset (sizes max_k .. 1, lexicographic inside a size)
                    while (i >= 0 && pos[i] == a - s + i) --i;
                    if (i < 0) {
                        --s;
#pragma unroll
                        for (int q = 0; q < K; ++q) pos[q] = q;
                    } else {
                        ++pos[i];
                        for (int j = i + 1; j < s; ++j) pos[j] = pos[j - 1] + 1;
                    }
                }
                if (power) {
                    FZS_WAVE_SYNC()
                    fzs_finish<K, GRAM>(s_q[wave], s_m[wave], ng, zs_job, cjob, mjob, GRAM ? &s_idx[wave][0][0] : nullptr);
                }
                my_done += (unsigned int)ng;
                bool stopped = false;
                for (int g = 0; g < ng; ++g) {
                    const double t_stat = power ? fzs_uniform(s_q[wave][g * L::QD + L::RES]) : 0.0;
                    const double t_pval = power ? fzs_uniform(s_q[wave][g * L::QD + L::RES + 1]) : 1.0;
                    const unsigned long long rr = r + (unsigned long long)g;
                    const bool sig = (t_pval < alpha) && power;
                    if (!sig || (max_tests > 0 && rr + 1 >= (unsigned long long)max_tests)) {
                        my_stop = rr;
                        stop_stat = t_stat;
                        stop_p = t_pval;
                        stop_pow = power ? 1 : 0;
                        stopped = true;
                        break;
                    }
                    if (t_pval >= my_bp) {
                        my_bp = t_pval;
                        my_br = rr;
                        my_bstat = t_stat;
                    }
                }
                if (stopped || s < 1) break;
                FZS_WAVE_SYNC()  // (the slots are rewritten by the next group)
                r = rg;
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

// ---- job-local correlation matrices ---------------------------------------------------------------------------------------------
// Every test of a (X, Y | subsets of the accepted list) job conditions a sub-matrix of ONE matrix: the Float64 correlations of
// {X, Y} u accepted.  Streaming the columns per test (above) computes each of them C(a, k - 1) times over; this kernel computes
// them once per job (the matrix stays in the launch arena while the job's pool lives) -- one workgroup per job, one wavefront per 4 x 4 tile of pairs (8 columns in 16-byte loads feed
// 16 sums x 4 samples), the same sum  x_i (x_j - mu_j), i < j in job order (X, Y, accepted positions), as the streamed form, the
// same DPP wave sums -- and the segment kernel (GRAM) then only gathers and conditions: tests/s go from 3.7e8 (streamed, itself
// 1.8 x the nominal HBM rate) to the rate of the conditioning arithmetic.  Row-major m x m doubles per job in the launch's arena.
__global__ __launch_bounds__(256) void fzs_gram_kernel(FzsDev P, const FwNzJob *__restrict__ recs, const int32_t *__restrict__ accflat,
                                                       double *__restrict__ arena)
{
    const FwNzJob J = recs[blockIdx.x];
    if (J.nR == 0) return;  // the job's matrix is still in the arena from an earlier round of its pool (fwi_pool_launch)
    const int m = J.m;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int32_t *acc = accflat + J.acc_off;
    double *C = arena + J.cor_off;
    const int T4 = (m + 3) >> 2, ntiles = T4 * (T4 + 1) / 2;
    for (int tile = wave; tile < ntiles; tile += 4) {
        int ti = 0, rem = tile;  // tile -> (ti, tj), ti <= tj, row by row over the upper triangle
        while (rem >= T4 - ti) {
            rem -= T4 - ti;
            ++ti;
        }
        const int tj = ti + rem;
        const float *ci[4], *cj[4];
        double muj[4], ssi[4], ssj[4];
        int idi[4], idj[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int li = 4 * ti + q, lj = 4 * tj + q;
            li = li < m ? li : m - 1;
            lj = lj < m ? lj : m - 1;
            const int vi = __builtin_amdgcn_readfirstlane(li == 0 ? J.X : (li == 1 ? J.Y : acc[li - 2]));
            const int vj = __builtin_amdgcn_readfirstlane(lj == 0 ? J.X : (lj == 1 ? J.Y : acc[lj - 2]));
            idi[q] = vi;
            idj[q] = vj;
            ci[q] = P.data + (size_t)vi * P.n;
            cj[q] = P.data + (size_t)vj * P.n;
            ssi[q] = P.st[2 * (size_t)vi + 1];
            muj[q] = P.st[2 * (size_t)vj];
            ssj[q] = P.st[2 * (size_t)vj + 1];
        }
        double S[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) S[a][b] = 0.0;
#define FZS_GACC(XS, YS)                                                     \
    {                                                                        \
        double d[4];                                                         \
        _Pragma("unroll") for (int b = 0; b < 4; ++b) d[b] = (double)(YS(b)) - muj[b]; \
        _Pragma("unroll") for (int a = 0; a < 4; ++a)                       \
        {                                                                    \
            const double xa = (double)(XS(a));                               \
            _Pragma("unroll") for (int b = 0; b < 4; ++b) S[a][b] = fma(xa, d[b], S[a][b]); \
        }                                                                    \
    }
        if ((P.n & 3) == 0) {
            const int n4 = P.n >> 2;
            for (int q = lane; q < n4; q += 64) {
                float4 xi[4], xj[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    xi[a] = ((const float4 *)ci[a])[q];
                    xj[a] = ((const float4 *)cj[a])[q];
                }
#define GX(a) xi[a].x
#define GY(b) xj[b].x
                FZS_GACC(GX, GY)
#undef GX
#undef GY
#define GX(a) xi[a].y
#define GY(b) xj[b].y
                FZS_GACC(GX, GY)
#undef GX
#undef GY
#define GX(a) xi[a].z
#define GY(b) xj[b].z
                FZS_GACC(GX, GY)
#undef GX
#undef GY
#define GX(a) xi[a].w
#define GY(b) xj[b].w
                FZS_GACC(GX, GY)
#undef GX
#undef GY
            }
        } else {
            for (int i = lane; i < P.n; i += 64) {
                float xi[4], xj[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    xi[a] = ci[a][i];
                    xj[a] = cj[a][i];
                }
#define GX(a) xi[a]
#define GY(b) xj[b]
                FZS_GACC(GX, GY)
#undef GX
#undef GY
            }
        }
#undef FZS_GACC
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int li = 4 * ti + a, lj = 4 * tj + b;
                if (li < lj && lj < m) {  // (uniform)
                    const double v = fzs_wave_sum(S[a][b]);
                    if (lane == 0) {
                        // the same variable twice in a job (feed-forward whitelists: hiton.jl:24-26 pushes a whitelisted member of the
                        // elimination pool again): its correlation with itself is exactly 1, as StatsBase's identical sums give it
                        // (S / sqrt(S S)), and the conditioning then divides by zero exactly as the reference does -- not by the
                        // rounding residue of x (x - mu) against (x - mu)^2
                        const double r = idi[a] == idj[b] ? 1.0 : v / sqrt(ssi[a] * ssj[b]);
                        C[(size_t)li * m + lj] = r;
                        C[(size_t)lj * m + li] = r;
                    }
                }
            }
    }
    for (int i = threadIdx.x; i < m; i += 256) C[(size_t)i * m + i] = 1.0;
}

FzsDev fzs_dev(const fw_ctx *ctx)
{
    FzsDev P;
    P.data = ctx->d_data;
    P.st = ctx->d_fzs_stat;
    P.n = ctx->P.n;
    P.p = ctx->P.p;
    P.zscale = ctx->P.n > 3 ? std::sqrt((double)(ctx->P.n - 3)) / 2.0 : 0.0;
    P.nzjobs = 0;
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

// X / Y columns in LDS: n a multiple of 4 (16-byte rows) and at most FZS_LDS_N samples
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
        if (kmax <= 3) FZS_BATCH(3, 8); else FZS_BATCH(FW_MAX_K_FAST, 8);
    } else {
        if (kmax <= 3) FZS_BATCH(3, 0); else FZS_BATCH(FW_MAX_K_FAST, 0);
    }
#undef FZS_BATCH
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipMemcpyAsync(out, ctx->d_out.ptr, (size_t)m * sizeof(fw_test_result), hipMemcpyDeviceToHost, ctx->stream));
    FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cnt.kernel_launches += 1;
    return FW_OK;
}

int fwi_fzs_segments(fw_ctx *ctx, int64_t nseg, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out, FwPoolBuf &pb,
                     const FwNzJob *recs_host, int64_t njobs, size_t arena_doubles)
{
    if (nseg == 0) return FW_OK;
    if (!ctx->d_data) return fw_fail(ctx, FW_ERR_STATE, "recursive_pcor = 0 needs the data matrix on the device (fw_set_data_dense_f32)");
    if (int rc = fzs_ensure_stat(ctx, pb.launch_stream)) return rc;
    // job-local correlation matrices (fzs_gram_kernel) unless FW_FZS_GRAM=0 (profiling / parity knob: every test streams its columns)
    // or the launch's matrices would not fit a sensible arena
    const bool gram_env = !(fw_knob("FW_FZS_GRAM") && atoi(fw_knob("FW_FZS_GRAM")) == 0);  // (read per launch: the tests switch it)
    const bool gram = gram_env && recs_host && njobs > 0 && arena_doubles * sizeof(double) <= ((size_t)8 << 30);
    int lds_m = 0;
    if (gram) {
        int rc;
        if ((rc = fw_dev_reserve(ctx, ctx->d_nzrecs, (size_t)njobs * sizeof(FwNzJob)))) return rc;
        if ((rc = fw_dev_reserve(ctx, ctx->d_arena, std::max<size_t>(arena_doubles, 1) * sizeof(double)))) return rc;
        FW_HIP(ctx, hipMemcpyAsync(ctx->d_nzrecs.ptr, recs_host, (size_t)njobs * sizeof(FwNzJob), hipMemcpyHostToDevice, pb.launch_stream));
        for (int64_t j = 0; j < njobs; ++j) lds_m = std::max(lds_m, (int)recs_host[j].m);
        lds_m = std::min(lds_m, 80);  // 80 x 80 doubles = 50 KB + 9 KB of static LDS: inside the 64 KB a workgroup gets without asking; longer jobs read their matrix through L2
    }
    FW_HIP(ctx, hipEventRecord(pb.ev0, pb.launch_stream));
    const FzsDev P = fzs_dev(ctx);
    if (gram) {
        hipLaunchKernelGGL(fzs_gram_kernel, dim3((unsigned)njobs), dim3(256), 0, pb.launch_stream, P, (const FwNzJob *)ctx->d_nzrecs.ptr, d_acc,
                           (double *)ctx->d_arena.ptr);
        ctx->cnt.kernel_launches += 1;
    }
    const size_t lds = gram ? (size_t)lds_m * lds_m * sizeof(double) : 0;
#define FZS_SEG(KK, TT, GG)                                                                                                        \
    hipLaunchKernelGGL((fzs_subsets_seg_kernel<KK, TT, GG>), dim3((unsigned)nseg), dim3(256), lds, pb.launch_stream, P, d_segs, d_acc, \
                       d_out, ctx->P.max_k, ctx->P.alpha, (long long)ctx->P.max_tests, (const FwNzJob *)ctx->d_nzrecs.ptr,         \
                       (const double *)ctx->d_arena.ptr, lds_m)
    if (gram) {
        if (ctx->P.max_k <= 3) FZS_SEG(3, 0, true); else FZS_SEG(FW_MAX_K_FAST, 0, true);
    } else if (fzs_hold_xy(ctx)) {
        if (ctx->P.max_k <= 3) FZS_SEG(3, 8, false); else FZS_SEG(FW_MAX_K_FAST, 8, false);
    } else {
        if (ctx->P.max_k <= 3) FZS_SEG(3, 0, false); else FZS_SEG(FW_MAX_K_FAST, 0, false);
    }
#undef FZS_SEG
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipEventRecord(pb.ev1, pb.launch_stream));
    return FW_OK;
}

// fz_nz without a matrix (fw_params.recursive_pcor = 0 with FW_FZ_NZ; the reference's FzTestCond with an empty cor_mat on the row views
// of hiton.jl:85, tests.jl:253 -> statfuns.jl:19-21): the job records and their Float64 view matrices are already on the device
// (fwi_fznz_submatrices with f64 = true wrote them into ctx->d_arena); the GRAM segment kernel conditions them as StatsBase.partialcor
// does, with each job's own sample size.
int fwi_fzs_segments_nz(fw_ctx *ctx, int64_t nseg, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out, FwPoolBuf &pb, int m_max)
{
    if (nseg == 0) return FW_OK;
    FW_HIP(ctx, hipEventRecord(pb.ev0, pb.launch_stream));
    FzsDev P;
    P.data = (const float *)ctx->d_data;
    P.st = nullptr;
    P.n = ctx->P.n;
    P.p = ctx->P.p;
    P.zscale = 0.0;
    P.n_obs_min = ctx->n_obs_min_eff;
    P.nzjobs = 1;
    const int lds_m = std::min(m_max, 80);
    const size_t lds = (size_t)lds_m * lds_m * sizeof(double);
    if (ctx->P.max_k <= 3)
        hipLaunchKernelGGL((fzs_subsets_seg_kernel<3, 0, true>), dim3((unsigned)nseg), dim3(256), lds, pb.launch_stream, P, d_segs, d_acc, d_out,
                           ctx->P.max_k, ctx->P.alpha, (long long)ctx->P.max_tests, (const FwNzJob *)ctx->d_nzrecs.ptr,
                           (const double *)ctx->d_arena.ptr, lds_m);
    else
        hipLaunchKernelGGL((fzs_subsets_seg_kernel<FW_MAX_K_FAST, 0, true>), dim3((unsigned)nseg), dim3(256), lds, pb.launch_stream, P, d_segs, d_acc,
                           d_out, ctx->P.max_k, ctx->P.alpha, (long long)ctx->P.max_tests, (const FwNzJob *)ctx->d_nzrecs.ptr,
                           (const double *)ctx->d_arena.ptr, lds_m);
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipEventRecord(pb.ev1, pb.launch_stream));
    return FW_OK;
}
