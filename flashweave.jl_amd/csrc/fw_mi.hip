// FlashWeave-F / FlashWeaveHE-F (discrete contingency-table tests) device path for gfx950.
//
// Reference semantics (file:line into /root/reference/src), SPARSE-path rules (default make_sparse = true):
//   univariate test                      tests.jl:28-77 (+ vector rule :80-92 at level 0)
//   conditional test                     tests.jl:184-229
//   2-way / 3-way tables                 contingency.jl:80-123 (2-way), :182-258 (k = 1 HE special case), :300-480 (generic)
//   mutual information / df / p          statfuns.jl:157-305
//   test_subsets                         tests.jl:281-346
// Data layout in HBM: every variable is a pair of bit planes over the samples, [p][W] 64-bit words each:
//   nz plane: value != 0;  hi plane: value == 2   (values are 0..2: presence/absence, or 0 + two non-zero bins).
// Information-theoretic width: 1 bit (mi) / 2 bits (mi_nz) per value -- the (k+2)*n*b/8 algorithmic bytes of SURVEY 8d.
//
// A conditional test is ONE WAVEFRONT (fw_mi_core.h): lanes <-> 32-row words of the bit planes, a table cell is an
// AND + popcount, a DPP reduction sums the lanes, then lanes <-> (stratum, cell) pairs compute marginals, MI terms (fp64
// log), df, and a wave reduction yields G2 / p.  Level 0 (all pairs) is a tiled AND+popcount kernel.
#include "fw_internal.h"

#include <algorithm>
#include <cmath>

#include "fw_mi_core.h"
#include "fw_unrank.h"

static MiDev mi_dev(const fw_ctx *ctx);
static double host_igamc(double a, double x);

// L = 0 selects the generic form (values above 2: mi_test_core_gen, one byte per value, 32-bit LDS table) in the kernels below
template <int L, int NXY, bool PRE, bool WIDE>
static __device__ __forceinline__ MiRes mi_test_any(const MiDev &P, int X, int Y, const MiZs &zs, int k, unsigned short *tab)
{
    if constexpr (L == 0)
        return mi_test_core_gen(P, X, Y, zs, k, P.gtab ? P.gtab + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * P.gtab_words : (unsigned *)tab);
    else
        return mi_test_core<L, NXY, PRE, WIDE>(P, X, Y, zs, k, tab);
}
#define MI_TAB_U16(L, WIDE) ((L) == 0 ? 2 * MIG_TAB32 : ((WIDE) ? 2 * MI_TAB16 : MI_TAB16))
// conditioning sets of 6 and 7 variables (r05, BIG = true): 3^7 strata x 6 entries (2 x 2 sub-table + total) -- 26 KB per wavefront;
// a 3 x 3 sub-table fits up to 3^6 strata (fwi_mi_big_limits)
#define MI_TAB16_BIG 13312
#define MI_TAB_SEL(L, WIDE, BIG) ((BIG) ? MI_TAB16_BIG : MI_TAB_U16(L, WIDE))

// ------------------------------------------------------------------------------------------------
// batch of single tests: one wave per test
// ------------------------------------------------------------------------------------------------
template <int L, int NXY, bool PRE, bool WIDE, bool BIG = false>
__global__ __launch_bounds__(256) void mi_test_batch_kernel(MiDev P, long long m, const int32_t *__restrict__ X,
                                                            const int32_t *__restrict__ Y,
                                                            const long long *__restrict__ zoff,
                                                            const int32_t *__restrict__ zflat,
                                                            fw_test_result *__restrict__ out)
{
    __shared__ unsigned short s_tab[4][MI_TAB_SEL(L, WIDE, BIG)];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + wave;
    if (t >= m) return;
    const int k = (int)(zoff[t + 1] - zoff[t]);
    MiZs zs;
#pragma unroll
    for (int q = 0; q < MI_MAX_K; ++q) zs.v[q] = (q < k) ? zflat[zoff[t] + q] : 0;
    if (P.prof) P.prof += 8 * t;  // one record per test
    MiRes r = mi_test_any<L, NXY, PRE, WIDE>(P, X[t], Y[t], zs, k, s_tab[wave]);
    const unsigned long long pt = P.prof ? __builtin_readcyclecounter() : 0ull;
    (void)mi_res_pval(r);
    if (P.prof && lane == 0) P.prof[3] = __builtin_readcyclecounter() - pt;
    if (lane == 0) {
        fw_test_result o;
        o.stat = r.stat;
        o.pval = r.pval;
        o.df = r.df;
        o.suff_power = r.power;
        out[t] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// test_subsets segments: 4 waves per workgroup, wave w evaluates ranks cbase + w*R .. (run of R consecutive ranks)
// ------------------------------------------------------------------------------------------------
#define MI_RUN 4

template <int L, int NXY, bool PRE, bool WIDE, bool BIG = false>
__device__ __forceinline__ void mi_seg_body(const MiDev &P, const FwSeg *__restrict__ segs, const int32_t *__restrict__ accflat,
                                            FwSegOut *__restrict__ out, int max_k, double alpha, long long max_tests,
                                            const unsigned sidx /* segment this workgroup evaluates */, int need_p)
{
    __shared__ unsigned short s_tab[4][MI_TAB_SEL(L, WIDE, BIG)];
    __shared__ unsigned long long s_stop[4], s_br[4];
    __shared__ double s_sstat[4], s_sp[4], s_bp[4], s_bstat[4];
    __shared__ int s_sdf[4], s_spow[4], s_bdf[4];
    __shared__ unsigned int s_evc[4];  // tests really executed by each wavefront in the current chunk
    const FwSeg seg = segs[sidx];
    const int a = seg.acc_len;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int32_t *gacc = accflat + seg.acc_off;
    unsigned long long cnt[MI_MAX_K + 1];
#pragma unroll
    for (int s = MI_MAX_K; s >= 1; --s) cnt[s] = (s <= max_k) ? (BIG ? fw_binom_any(a, s) : fw_binom_u64(a, s)) : 0ull;
    const unsigned long long NONE = FW_RANK_NONE;
    // running best of the segment (kept redundantly by every thread: values come from LDS broadcasts)
    double best_p = -1.0, best_stat = 0.0;
    unsigned long long best_rank = 0, evaluated = 0;
    int best_df = 0;
    const unsigned long long len = seg.end - seg.start;
    const int R = (int)((len + 3) / 4 < MI_RUN ? (len + 3) / 4 : MI_RUN);
    for (unsigned long long cbase = seg.start; cbase < seg.end; cbase += 4ull * R) {
        const unsigned long long r0 = cbase + (unsigned long long)wave * R;
        unsigned long long r1 = r0 + R;
        if (r1 > seg.end) r1 = seg.end;
        unsigned long long my_stop = NONE, my_br = 0;
        double stop_stat = 0.0, stop_p = 0.0, my_bp = -1.0, my_bstat = 0.0;
        int stop_df = 0, stop_pow = 0, my_bdf = 0;
        unsigned int my_done = 0;
        if (r0 < seg.end) {
            unsigned long long rem = r0;
            int s = max_k;
            while (s > 1 && rem >= cnt[s]) {
                rem -= cnt[s];
                --s;
            }
            int pos[MI_MAX_K];
#pragma unroll
            for (int q = 0; q < MI_MAX_K; ++q) pos[q] = 0;
            if (BIG)
                fw_unrank_scan(rem, a, s, pos);
            else
                fw_unrank_comb(rem, a, s, pos);
            MiBest mb;
            mb.p = -3.0;
            mb.stat = mb.g = 0.0;
            mb.df = 0;
            for (unsigned long long r = r0; r < r1; ++r) {
                MiZs zs;
#pragma unroll
                for (int q = 0; q < MI_MAX_K; ++q) zs.v[q] = (q < s) ? gacc[pos[q]] : 0;
                MiRes t = mi_test_any<L, NXY, PRE, WIDE>(P, seg.X, seg.Y, zs, s, s_tab[wave]);
                ++my_done;
                const int ev = mi_account(P, t, max_tests > 0 && r + 1 >= (unsigned long long)max_tests, mb, need_p != 0);
                if (ev == 1) {
                    my_stop = r;
                    stop_stat = t.stat;
                    stop_p = t.pval;
                    stop_df = t.df;
                    stop_pow = t.power;
                    break;
                }
                if (ev == 2) {
                    my_bp = mb.p;
                    my_br = r;
                    my_bstat = mb.stat;
                    my_bdf = mb.df;
                }
                int i = s - 1;
                while (i >= 0 && pos[i] == a - s + i) --i;
                if (i < 0) {
                    --s;
#pragma unroll
                    for (int q = 0; q < MI_MAX_K; ++q) pos[q] = q;
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
            s_sdf[wave] = stop_df;
            s_spow[wave] = stop_pow;
            s_bp[wave] = my_bp;
            s_br[wave] = my_br;
            s_bstat[wave] = my_bstat;
            s_bdf[wave] = my_bdf;
        }
        __syncthreads();
        evaluated += (unsigned long long)(s_evc[0] + s_evc[1] + s_evc[2] + s_evc[3]);  // executed tests, not chunk sizes
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
                o.stop_df = s_sdf[fw];
                o.stop_power = s_spow[fw];
                o.best_df = 0;
                o.pad = 0;
                o.evaluated = evaluated;
                out[sidx] = o;
            }
            return;
        }
#pragma unroll
        for (int w = 0; w < 4; ++w)  // waves hold increasing ranks: sequential `>=` merge (tests.jl:338)
            if (s_bp[w] >= 0.0 && s_bp[w] >= best_p) {
                best_p = s_bp[w];
                best_stat = s_bstat[w];
                best_rank = s_br[w];
                best_df = s_bdf[w];
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
        o.best_df = best_df;
        o.pad = 0;
        o.evaluated = evaluated;
        out[sidx] = o;
    }
}

// Host-driven rounds: one workgroup per segment (ns_dev == nullptr); device-driven rounds (fw_devhiton.hip): a fixed
// grid covers the device-built segment list whose live length is *ns_dev.
template <int L, int NXY, bool PRE, bool WIDE, bool BIG = false>
__global__ __launch_bounds__(256) void mi_subsets_seg_kernel(MiDev P, const FwSeg *__restrict__ segs,
                                                             const int32_t *__restrict__ accflat,
                                                             FwSegOut *__restrict__ out, int max_k, double alpha,
                                                             long long max_tests, const unsigned *__restrict__ ns_dev, int need_p)
{
    // no grid-stride loop here: with the body inside a loop the compiler hoists its invariants and needs 254 VGPRs
    // (occupancy 1 instead of 3); the device-driven grid covers the whole segment list and surplus workgroups leave
    if (ns_dev && blockIdx.x >= *ns_dev) return;
    mi_seg_body<L, NXY, PRE, WIDE, BIG>(P, segs, accflat, out, max_k, alpha, max_tests, blockIdx.x, need_p);
}

// ------------------------------------------------------------------------------------------------
// level 0: all pairs X < Y.  64 x 64 pair tiles, AND + popcount over the bit planes, 4 x 4 pairs per thread.
// The joint 3 x 3 table follows from A = |nzX & nzY|, B = |hiX & nzY|, C = |nzX & hiY|, D = |hiX & hiY| and the
// per-column totals.  p-values (igamc) are only evaluated for pairs whose G statistic is near/above the alpha
// quantile: BH only looks at p < alpha and at the count of reliable tests (statfuns.jl:331, tests.jl:522-526).
// ------------------------------------------------------------------------------------------------
struct MiL0Counters {
    unsigned long long n_sig;
    unsigned long long n_unreliable;  // tests without power (NaN in the reference's condensed arrays)
};

#define L0_T 64
#define L0_WC 8
#define L0_QCAP 512  // per-workgroup candidate queue of the level-0 screen

// -> bit 0: significant (p < alpha), bit 1: unreliable; statistic and p-value of a significant pair in stat_out / pval_out
__device__ __forceinline__ int mi_pair_epilogue(const MiDev &P, int X, int Y, int A, int B, int C, int D, const int32_t *cnt_nz,
                                                const int32_t *cnt_hi, double alpha, const double *gthr, double &stat_out, double &pval_out)
{
    const int L = P.L;
    const int nzX = cnt_nz[X], nzY = cnt_nz[Y], hiX = cnt_hi[X], hiY = cnt_hi[Y];
    // joint table t[x][y], x,y in {0,1,2}: 2 = hi, 1 = nz & !hi
    long long t[3][3];
    t[2][2] = D;
    t[2][1] = B - D;
    t[1][2] = C - D;
    t[1][1] = A - B - C + D;
    t[2][0] = hiX - B;
    t[1][0] = (nzX - hiX) - (A - B);
    t[0][2] = hiY - C;
    t[0][1] = (nzY - hiY) - (A - C);
    t[0][0] = (long long)P.n - nzX - nzY + A;
    bool unreliable = false;
    double stat = 0.0, pval = 1.0;
    // vector rule (tests.jl:86-88): everything fails if levels[X] < 2; then the scalar test (tests.jl:28-77)
    const long long vx = P.levels[X], vy = P.levels[Y];
    const long long ox = vx > 1 ? 2 : 1, oy = vy > 1 ? 2 : 1;
    bool pre = vx >= 2 && (long long)P.n >= P.n_obs_min && (((double)P.n / (double)((vx - ox) * (vy - oy))) > (double)P.hps);
    if (!pre) {
        unreliable = true;
    } else {
        const bool flagX = P.nzmode && P.maxv[X] > 1, flagY = P.nzmode && P.maxv[Y] > 1;
        const int sx = flagX ? 1 : 0, sy = flagY ? 1 : 0;
        int lx, ly;
        if (P.nzmode) {
            lx = L - sx;
            ly = L - sy;
        } else {
            lx = (int)vx;
            ly = (int)vy;
        }
        long long n_obs = 0;
        for (int i = sx; i < L; ++i)
            for (int j = sy; j < L; ++j) n_obs += t[i][j];
        if (n_obs < P.n_obs_min || !(((double)n_obs / (double)((long long)lx * ly)) > (double)P.hps)) {
            unreliable = true;
        } else {
            long long mi_[3] = {0, 0, 0}, mj_[3] = {0, 0, 0};
            for (int i = 0; i < lx; ++i)
                for (int j = 0; j < ly; ++j) {
                    mi_[i] += t[i + sx][j + sy];
                    mj_[j] += t[i + sx][j + sy];
                }
            double pos = 0.0, neg = 0.0;
            long long npos = 0, nneg = 0;
            for (int i = 0; i < lx; ++i)
                for (int j = 0; j < ly; ++j) {
                    const long long c = t[i + sx][j + sy];
                    if (c != 0 && mi_[i] != 0 && mj_[j] != 0) {
                        const double term = (double)c * log((double)(n_obs * c) / (double)(mi_[i] * mj_[j]));
                        if (i == j) {
                            pos += term;
                            npos += c;
                        } else {
                            neg += term;
                            nneg += c;
                        }
                    }
                }
            double mi = (pos + neg) / (double)n_obs;
            if (neg * ((double)nneg / (double)n_obs) > pos * ((double)npos / (double)n_obs)) mi *= -1.0;
            int alx = 0, aly = 0;
            for (int i = 0; i < lx; ++i) alx += mi_[i] > 0;
            for (int j = 0; j < ly; ++j) aly += mj_[j] > 0;
            alx = alx < 1 ? 1 : alx;
            aly = aly < 1 ? 1 : aly;
            const int df = (alx - 1) * (aly - 1);
            stat = mi;
            const double g = 2.0 * fabs(mi) * (double)n_obs;
            if (df > 0 && g >= gthr[df])  // below the (slightly lowered) alpha quantile p >= alpha for sure
                pval = mi_igamc(0.5 * (double)df, 0.5 * g);
            else
                pval = 1.0;  // any value >= alpha: never looked at again
        }
    }
    stat_out = stat;
    pval_out = pval;
    return ((!unreliable && pval < alpha) ? 1 : 0) | (unreliable ? 2 : 0);
}

// ---- level-0 screening (kernel 1) -----------------------------------------------------------------------
// Integer logic (reliability, df) is exact; the G statistic is evaluated in Float32 only to decide whether the pair
// can possibly reach the alpha quantile.  |G32 - G| <= 2 * sum_c c * (2e-7 |log| + 1e-7) <= n * 4e-6, so a margin of
// 0.5 + 1e-4 n below 0.99 * quantile is safe by more than an order of magnitude.  Candidates go to kernel 2, which
// evaluates the statistic and the p-value in Float64 (mi_pair_epilogue).  Everything is unrolled over the 3 x 3
// table so that nothing lives in scratch memory.
struct MiCand {
    int32_t X, Y, A, B, C, D;
};

// returns 1 if the pair is unreliable (no power / too few observations), else 0.
// mX / mY = {nz count, hi count, levels, max value} of the two columns (staged in LDS by the caller).
// G / 2 = sum_c c ln(c n / (m_i m_j)) = sum_c T[c] + S ln(n_obs) - sum_i T[m_i] - sum_j T[m_j] with T[x] = x ln x and
// S = sum of the cells inside the level ranges: 15 lookups in a Float32 table of T plus one of ln instead of nine
// logarithms (the power rules n / d > hps are evaluated as the equivalent integer comparison n > hps * d).
// The integer / Float32 part of mi_pair_screen for the pair every HE table is made of (see the comment inside it), as its own function
// for callers that screen in two passes (mi_level0_mfma_kernel): 1 = unreliable, 0 = cannot be significant, -1 = needs the full screen.
__device__ __forceinline__ int mi_pair_prescreen(const MiDev &P, const int4 mX, const int4 mY, int A, int B, int C, int D, const double *gthr)
{
    const bool flagX = P.nzmode && mX.w > 1, flagY = P.nzmode && mY.w > 1;
    if (!(flagX && flagY && P.L == 3)) return -1;
    const int vx = mX.z, vy = mY.z;
    const int ox = vx > 1 ? 2 : 1, oy = vy > 1 ? 2 : 1;
    bool reliable = vx >= 2 && (long long)P.n >= P.n_obs_min && (long long)P.n > (long long)P.hps * (vx - ox) * (vy - oy);
    reliable = reliable && (long long)A >= P.n_obs_min && (long long)A > (long long)P.hps * 4;
    if (!reliable) return 1;
    const int r1 = A - B, c1 = A - C;
    if (r1 == 0 || B == 0 || c1 == 0 || C == 0) return 0;
    const float det = (float)((long long)A * D - (long long)B * C);
    const float lhs = 2.0f * (float)A * det * det, rhs = ((float)r1 * (float)B) * ((float)c1 * (float)C);
    return (lhs < 0.98f * (float)gthr[1] * rhs) ? 0 : -1;
}

__device__ __forceinline__ int mi_pair_screen(const MiDev &P, const int4 mX, const int4 mY, int X, int Y, int A, int B, int C, int D,
                                              const float *__restrict__ xlnx, const float *__restrict__ lnx, const double *gthr,
                                              MiL0Counters *cnt, unsigned long long cap_c, MiCand *__restrict__ cands,
                                              MiCand *s_q, int *s_qn, int qcap = L0_QCAP)
{
    const int L = P.L;
    const int nzX = mX.x, nzY = mY.x, hiX = mX.y, hiY = mY.y;
    int t00, t01, t02, t10, t11, t12, t20, t21, t22;
    t22 = D;
    t21 = B - D;
    t12 = C - D;
    t11 = A - B - C + D;
    t20 = hiX - B;
    t10 = (nzX - hiX) - (A - B);
    t02 = hiY - C;
    t01 = (nzY - hiY) - (A - C);
    t00 = P.n - nzX - nzY + A;
    const int vx = mX.z, vy = mY.z;
    const int ox = vx > 1 ? 2 : 1, oy = vy > 1 ? 2 : 1;
    bool reliable = vx >= 2 && (long long)P.n >= P.n_obs_min && (long long)P.n > (long long)P.hps * (vx - ox) * (vy - oy);
    const bool flagX = P.nzmode && mX.w > 1, flagY = P.nzmode && mY.w > 1;
    // Fast verdict for the pair every HE table is made of: both columns hold the value 2 in nz mode, so the adjusted table is
    // the 2 x 2 table of the levels {1, 2} on the rows where both are non-zero -- cells (A-B-C+D, C-D; B-D, D), n_obs = A, row
    // sums (A-B, B), column sums (A-C, C) -- and everything the general code below derives cell by cell is a handful of
    // integers: reliability (tests.jl:50-56), df = 1 unless a marginal is empty, and for G the bound
    //     G = 2 n KL(P_xy || P_x P_y) <= 2 n chi^2-divergence = 2 X^2,   X^2 = A (A D - B C)^2 / ((A-B) B (A-C) C)
    // (ln x <= x - 1).  A pair with 2 X^2 below the alpha quantile cannot be significant: no table look-up at all (r03: the
    // 16 look-ups of every reliable pair were 19 of cfg4's 64 ms).  Float32 products of five counts <= 2^17 stay below 2^85;
    // their relative error (< 1e-6) is covered by the factor 0.98 on a quantile that is already lowered (gthr).
    if (flagX && flagY && L == 3) {
        reliable = reliable && (long long)A >= P.n_obs_min && (long long)A > (long long)P.hps * 4;
        if (!reliable) return 1;
        const int r1 = A - B, c1 = A - C;
        if (r1 == 0 || B == 0 || c1 == 0 || C == 0) return 0;  // an empty marginal: df = 0, p = 1
        const float det = (float)((long long)A * D - (long long)B * C);
        const float lhs = 2.0f * (float)A * det * det, rhs = ((float)r1 * (float)B) * ((float)c1 * (float)C);
        if (lhs < 0.98f * (float)gthr[1] * rhs) return 0;
    }
    const int sx = flagX ? 1 : 0, sy = flagY ? 1 : 0;
    const int lx = P.nzmode ? L - sx : vx, ly = P.nzmode ? L - sy : vy;
    const int tt[3][3] = {{t00, t01, t02}, {t10, t11, t12}, {t20, t21, t22}};
    int n_obs = 0, S = 0;
    int mi_[3] = {0, 0, 0}, mj_[3] = {0, 0, 0};  // indexed by sub-table row / column
    int tv[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const bool insub = i >= sx && j >= sy && i < L && j < L;
            n_obs += insub ? tt[i][j] : 0;
            const bool inlev = insub && (i - sx) < lx && (j - sy) < ly;
            const int v = inlev ? tt[i][j] : 0;
            // sub-table indices i - sx, j - sy in {0, 1, 2}; sx, sy in {0, 1}
            if (sx == 0) mi_[i] += v; else if (i >= 1) mi_[i - 1] += v;
            if (sy == 0) mj_[j] += v; else if (j >= 1) mj_[j - 1] += v;
            S += v;
            tv[i][j] = v;
        }
    // integer verdicts first: an unreliable pair or one without degrees of freedom needs none of the 16 table lookups below
    // (r03: they were interleaved with the counting loop and paid for by every pair)
    reliable = reliable && (long long)n_obs >= P.n_obs_min && (long long)n_obs > (long long)P.hps * lx * ly;
    if (!reliable) return 1;  // counted per workgroup by the caller (one atomic per pair serialised the whole kernel)
    int alx = (mi_[0] > 0) + (mi_[1] > 0) + (mi_[2] > 0), aly = (mj_[0] > 0) + (mj_[1] > 0) + (mj_[2] > 0);
    alx = alx < 1 ? 1 : alx;
    aly = aly < 1 ? 1 : aly;
    const int df = (alx - 1) * (aly - 1);
    if (df == 0) return 0;  // p = 1
    double g32;
    if (xlnx) {
        float g = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) g += xlnx[tv[i][j]];
        g += (float)S * lnx[n_obs];
        g -= (xlnx[mi_[0]] + xlnx[mi_[1]] + xlnx[mi_[2]]) + (xlnx[mj_[0]] + xlnx[mj_[1]] + xlnx[mj_[2]]);
        // |g - G/2| <= 16 table roundings of <= 0.003 each at n <= 65536: far inside the margin below
        g32 = 2.0 * fabs((double)g);
    } else {
        // no tables (mi_level0_mfma_kernel: one wavefront per SIMD, where 16 look-ups in global memory are 16 exposed round trips):
        // x log2 x from the hardware logarithm (v_log_f32, 1 ulp), summed in Float64, times ln 2 at the end.  A term is at most
        // n log2 n with a relative error below 2^-22: 0.03 at n = 5 000, 0.5 at n = 65 535 -- 16 of them stay inside the margin below
        // (1.0 resp. 7.0 in G).
        auto T2 = [](int x) -> double { return x > 0 ? (double)((float)x * __log2f((float)x)) : 0.0; };
        double g2 = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) g2 += T2(tv[i][j]);
        g2 += (double)((float)S * __log2f((float)n_obs));
        g2 -= (T2(mi_[0]) + T2(mi_[1]) + T2(mi_[2])) + (T2(mj_[0]) + T2(mj_[1]) + T2(mj_[2]));
        g32 = 2.0 * 0.6931471805599453 * fabs(g2);
        // tighter margin than the table form's: |g32 - G| <= 2 ln 2 x 16 terms x n log2 n x 2^-22 (0.33 at n = 5 000, 5.6 at 65 535)
        const double fn = (double)P.n;
        const double err = 2.0 * 0.6931471805599453 * 16.0 * fn * (double)__log2f((float)P.n) * 2.384185791015625e-07;
        if (g32 < 0.99 * gthr[df] - (0.05 + 1.5 * err)) return 0;
    }
    if (xlnx && g32 < 0.99 * gthr[df] - (0.5 + 1e-4 * (double)P.n)) return 0;  // cannot reach the alpha quantile
    // candidates are queued per workgroup in LDS and appended to the global list with ONE atomic per workgroup (millions
    // of atomics on the one counter serialised the kernel: 46 of 66 ms at cfg4); overflow falls back to the direct append
    MiCand cd;
    cd.X = X;
    cd.Y = Y;
    cd.A = A;
    cd.B = B;
    cd.C = C;
    cd.D = D;
    const int qs = atomicAdd(s_qn, 1);
    if (qs < qcap) {
        s_q[qs] = cd;
    } else {
        const unsigned long long slot = atomicAdd(&cnt->n_sig, 1ull);  // n_sig doubles as the candidate counter in kernel 1
        if (slot < cap_c) cands[slot] = cd;
    }
    return 0;
}

// The full screen of mi_pair_screen for a pair the first pass of mi_level0_mfma_kernel has already found standard (both variables
// nz-adjusted with three levels: the 2 x 2 table of the non-zero levels), reliable and with four non-empty marginals (df = 1): the
// nine terms of G that are not zero, in the order and with the roundings of the general form's table-free branch -- same value, same
// verdict, a tenth of its instructions (r05: the second pass was 10 000 of the kernel's 85 000 cycles per tile).
// thr1 = 0.99 gthr[1] - (0.05 + 1.5 err), see mi_pair_screen.
__device__ __forceinline__ void mi_pair_screen_std(int X, int Y, int A, int B, int C, int D, double thr1, MiL0Counters *cnt,
                                                   unsigned long long cap_c, MiCand *__restrict__ cands, MiCand *s_q, int *s_qn, int qcap)
{
    auto T2 = [](int x) -> double { return x > 0 ? (double)((float)x * __log2f((float)x)) : 0.0; };
    double g2 = 0.0;
    g2 += T2(A - B - C + D);
    g2 += T2(C - D);
    g2 += T2(B - D);
    g2 += T2(D);
    g2 += (double)((float)A * __log2f((float)A));
    g2 -= (T2(A - B) + T2(B) + 0.0) + (T2(A - C) + T2(C) + 0.0);
    const double g32 = 2.0 * 0.6931471805599453 * fabs(g2);
    if (g32 < thr1) return;
    MiCand cd;
    cd.X = X, cd.Y = Y, cd.A = A, cd.B = B, cd.C = C, cd.D = D;
    const int qs = atomicAdd(s_qn, 1);
    if (qs < qcap) {
        s_q[qs] = cd;
    } else {
        const unsigned long long slot = atomicAdd(&cnt->n_sig, 1ull);
        if (slot < cap_c) cands[slot] = cd;
    }
}

// kernel 2: exact Float64 statistic + p-value for the screened candidates, one thread each (L0X_IT x 256 candidates per workgroup)
#define L0X_IT 4
__global__ __launch_bounds__(256) void mi_level0_exact_kernel(MiDev P, const MiCand *__restrict__ cands, unsigned long long ncand,
                                                              const int32_t *__restrict__ cnt_nz, const int32_t *__restrict__ cnt_hi,
                                                              double alpha, const double *gthr, MiL0Counters *cnt,
                                                              unsigned long long cap, int32_t *out_i, int32_t *out_j,
                                                              double *out_s, double *out_p);

template <bool HAS_HI>
__global__ __launch_bounds__(256) void mi_level0_kernel(MiDev P, int p, int T, const int32_t *__restrict__ cnt_nz,
                                                        const int32_t *__restrict__ cnt_hi, const double *gthr,
                                                        MiL0Counters *cnt, unsigned long long cap_c, MiCand *__restrict__ cands,
                                                        const float *__restrict__ xlnx, const float *__restrict__ lnx,
                                                        int dbg /* profiling only: 1 no epilogue, 2 no loads, 4 no popcounts */,
                                                        int b_off /* first tile of this launch (rank's share, fw_level0_sharded) */)
{
    // word-major staging: lanes of a wave read consecutive 8-byte elements of a word row (Y columns are dealt
    // tx + 16 v), X rows are wave broadcasts -> no LDS bank conflicts in the popcount loop
    __shared__ unsigned long long sXn[L0_WC][L0_T], sYn[L0_WC][L0_T];
    __shared__ unsigned long long sXh[HAS_HI ? L0_WC : 1][L0_T], sYh[HAS_HI ? L0_WC : 1][L0_T];
    __shared__ double s_gthr[8];
    __shared__ int s_cnt[256 * 16];
    __shared__ int4 s_meta[2 * L0_T];  // {nz count, hi count, levels, max value} of the tile's X and Y columns
    __shared__ MiCand s_q[L0_QCAP];
    __shared__ int s_qn;
    __shared__ unsigned long long s_qbase;
    // triangular tile decode
    int b = blockIdx.x + b_off, bi = 0;
    while (b >= T - bi) {
        b -= T - bi;
        ++bi;
    }
    const int bj = bi + b;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    if (tid < 8) s_gthr[tid] = gthr[tid];
    if (tid == 0) s_qn = 0;
    if (tid < 2 * L0_T) {
        const int g = (tid < L0_T ? bi : bj) * L0_T + (tid & (L0_T - 1));
        s_meta[tid] = g < p ? make_int4(cnt_nz[g], cnt_hi[g], P.levels[g], P.maxv[g]) : make_int4(0, 0, 0, 0);
    }
    int A[4][4], B[4][4], C[4][4], D[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) A[u][v] = B[u][v] = C[u][v] = D[u][v] = 0;
    // Staging is software-pipelined (r04): the words of chunk c + 1 are requested into registers (two elements per thread and
    // plane: 16 VGPRs) BEFORE the popcount loop of chunk c runs and written to LDS after it, so the L2 round trip of a chunk
    // hides behind ~2 000 VALU instructions instead of standing between two barriers.  (r03 ablation: staging alone 12.6 ms,
    // popcount loop ~35 ms, and the kernel took their SUM -- three workgroups per CU did not overlap them.)
    constexpr int L0_EPT = L0_T * L0_WC / 256;  // elements per thread and array
    unsigned long long rXn[L0_EPT], rYn[L0_EPT], rXh[L0_EPT], rYh[L0_EPT];
    auto l0_fetch = [&](int w0) {
#pragma unroll
        for (int q = 0; q < L0_EPT; ++q) {
            const int e = tid + 256 * q;
            const int col = e / L0_WC, w = e % L0_WC;
            const int gx = bi * L0_T + col, gy = bj * L0_T + col;
            const bool wv = w0 + w < P.W && !(dbg & 2);
            rXn[q] = (gx < p && wv) ? P.nz[(size_t)gx * P.W + w0 + w] : 0ull;
            rYn[q] = (gy < p && wv) ? P.nz[(size_t)gy * P.W + w0 + w] : 0ull;
            if (HAS_HI) {
                rXh[q] = (gx < p && wv) ? P.hi[(size_t)gx * P.W + w0 + w] : 0ull;
                rYh[q] = (gy < p && wv) ? P.hi[(size_t)gy * P.W + w0 + w] : 0ull;
            } else {
                rXh[q] = rYh[q] = 0ull;
            }
        }
    };
    l0_fetch(0);
    for (int w0 = 0; w0 < P.W; w0 += L0_WC) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < L0_EPT; ++q) {
            const int e = tid + 256 * q;
            const int col = e / L0_WC, w = e % L0_WC;
            sXn[w][col] = rXn[q];
            sYn[w][col] = rYn[q];
            if (HAS_HI) {
                sXh[w][col] = rXh[q];
                sYh[w][col] = rYh[q];
            }
        }
        __syncthreads();
        if (w0 + L0_WC < P.W) l0_fetch(w0 + L0_WC);  // in flight while this chunk is counted
#pragma unroll 1  // unrolling this loop made the compiler hoist all 8 x 16 LDS reads: 256 VGPRs + scratch (r01 ISA)
        for (int w = 0; w < ((dbg & 4) ? 0 : L0_WC); ++w) {
            unsigned long long xn[4], yn[4], xh[4], yh[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xn[u] = sXn[w][ty * 4 + u];
                yn[u] = sYn[w][tx + 16 * u];
                if (HAS_HI) {
                    xh[u] = sXh[w][ty * 4 + u];
                    yh[u] = sYh[w][tx + 16 * u];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    // acc += popcount(a & b): v_bcnt_u32_b32 adds its second operand, so a 64-bit word costs 2 ANDs + 2 BCNTs.
                    // Written as inline asm: the compiler expands __popcll / __builtin_popcount to bcnt(x, 0) and adds the
                    // results with a separate v_add / v_add3 per counter -- 64 of the loop's 323 instructions (r02 ISA)
#define L0_ACC(acc, a, b)                                                                                   \
    {                                                                                                       \
        const unsigned lo_ = (unsigned)(a) & (unsigned)(b), hi_ = (unsigned)((a) >> 32) & (unsigned)((b) >> 32); \
        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(lo_));                                           \
        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(hi_));                                           \
    }
                    L0_ACC(A[u][v], xn[u], yn[v]);
                    if (HAS_HI) {
                        L0_ACC(B[u][v], xh[u], yn[v]);
                        L0_ACC(C[u][v], xn[u], yh[v]);
                        L0_ACC(D[u][v], xh[u], yh[v]);
                    }
#undef L0_ACC
                }
        }
    }
    __syncthreads();
    if (dbg & 1) {
        if (A[0][0] + B[1][1] + C[2][2] + D[3][3] == -12345) cnt->n_sig = 1;  // keep the loop alive
        return;
    }
    // The 64 counters of a thread are parked in LDS row by row so that the screening code can run in a rolled loop
    // with a dynamic index: inlining it 16 times with the counters live cost 256 VGPRs + scratch (r01 ISA).
    int *mine = s_cnt + tid * 16;
    int n_unrel = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            mine[v * 4 + 0] = A[u][v];
            mine[v * 4 + 1] = B[u][v];
            mine[v * 4 + 2] = C[u][v];
            mine[v * 4 + 3] = D[u][v];
        }
        const int X = bi * L0_T + ty * 4 + u;
        const int4 mX = s_meta[ty * 4 + u];
#pragma unroll 1
        for (int v = 0; v < 4; ++v) {
            const int Y = bj * L0_T + tx + 16 * v;
            if (X < Y && Y < p)
                n_unrel += mi_pair_screen(P, mX, s_meta[L0_T + tx + 16 * v], X, Y, mine[v * 4 + 0], mine[v * 4 + 1], mine[v * 4 + 2],
                                          mine[v * 4 + 3], xlnx, lnx, s_gthr, cnt, cap_c, cands, s_q, &s_qn);
        }
    }
    n_unrel = wave_sum_i(n_unrel);
    if ((tid & 63) == 0 && n_unrel) atomicAdd(&cnt->n_unreliable, (unsigned long long)n_unrel);
    __syncthreads();
    const int nq = s_qn < L0_QCAP ? s_qn : L0_QCAP;
    if (tid == 0 && nq > 0) s_qbase = atomicAdd(&cnt->n_sig, (unsigned long long)nq);
    __syncthreads();
    for (int q = tid; q < nq; q += 256)
        if (s_qbase + (unsigned long long)q < cap_c) cands[s_qbase + q] = s_q[q];
}

// ------------------------------------------------------------------------------------------------
// kernel 1, matrix-core form (r04 / r05; three-valued data, n <= 65 535).  The four counts of a pair are four entries of the Gram matrix
// of the 2p bit planes over the n samples: A = <nzX, nzY>, B = <hiX, nzY>, C = <nzX, hiY>, D = <hiX, hiY> -- a binary GEMM that
// the popcount form above runs at the integer-VALU peak (16 instructions per pair and 64-sample word: ~40 ms at cfg4 whatever the
// tiling).  Here: the MX-fp4 instruction v_mfma_scale_f32_32x32x64_f8f6f4 (a set bit is a power-of-two E2M1 value, block scale 2^0, every
// product of two set bits is exactly 1, counts exact in the Float32 accumulators; 65 536 multiply-adds per 32 cycles and SIMD).
// (The int8 form v_mfma_i32_32x32x32_i8 was measured first in r04 -- 19.8 against 14.8 ms -- and is gone.)  Workgroup tile 128 x 128
// variables, 512 threads = eight wavefronts, TWO PER SIMD (a lone wavefront issues one instruction per ~5.5 cycles whatever it
// executes); wavefront (wx, wy) owns 64 X x 32 Y variables = 2 blocks of 32 x 32 variables x 4 plane pairs = 8 accumulator tiles (128
// registers).  ANY assignment of samples to the K index of the instruction is right as long as both operands use the same one -- the
// sum over samples does not depend on their order -- so no layout table is involved beyond "row = lane & 31, K group = lane >> 5" for
// both operands and the documented C layout (col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5));
// profiles/tools/mfma_fp4_probe.cpp checks exactly that against host popcounts.
//
// r05, from two measurements (profiles/r05_level0_matrix_loop.json): (i) profiles/tools/mfma_overlap_gen.py -- the SIMD issues up to
// five vector instructions under every matrix instruction for free with two wavefronts resident, the sixth costs, a lone wavefront keeps
// the pipe busy only up to ~3; (ii) ablation builds of this kernel -- the operand reads from LDS cost nothing, staging cost 2.1 of the
// loop's 10.2 ms: the compiler had put the "value or zero" selects of the staging loads right behind the loads, so every wavefront sat
// out the full memory latency at the head of every stage, behind a dependent load of the plane pointer (a per-lane choice between two
// kernel arguments becomes a load from the argument segment).  Now:
//  * staging map with the plane fixed per load (q & 1) and one 32-bit lane offset for all loads of a tile (scalar base + offset);
//    interior tiles load unconditionally, the last tile row / column selects at the LDS write, a whole stage after the load;
//  * pre-permuted operands: the thread that stages a 64-sample word rewrites it ONCE per workgroup so that a lane's four
//    operand registers are three ANDs and a copy instead of seven instructions in each of the wavefronts that share the row (3 x
//    duplicated: 42 -> 18 vector instructions per wavefront and word).  X side: register q = the samples at bit q of every nibble, as
//    the E2M1 values 0.5 / 1 / 2 (bits 0, 1, 2 in place) and, for bit 3, 2 (moved to bit 2, second word); Y side: the same samples as
//    2 / 1 / 0.5 / 0.5 -- every product of two set bits is 1;
//  * the barrier of a stage sits in front of the LAST two words' matrix instructions, so the first operands of the next stage are read
//    and expanded under 16 matrix instructions instead of in a bubble behind the barrier.
typedef int l0m_v8i __attribute__((ext_vector_type(8)));
typedef float l0m_v16f __attribute__((ext_vector_type(16)));
#define L0M_T 128
#define L0M_WC 8      // 64-sample words per stage
static_assert(L0M_T == 128 && L0M_WC == 8, "the staging map of mi_level0_mfma_kernel is written for 128-variable tiles and 8-word stages");
#define L0M_S 8       // tiles per side of a super-tile (the unit of the XCD-aware order and of the sharded forms)
#define L0M_QCAP 1024 // per-workgroup candidate queue
#define L0M_SCAP 2048 // pairs per tile that pass the integer / Float32 verdicts of the first pass (an eighth per wavefront; more: to the exact kernel unscreened)
#define L0M_CAPL 10   // survivors per lane in the packed first pass (a lane's list; beyond: the wavefront takes the general form)
#define L0M_RS 34                          // 32-bit words per staged row (side, plane, variable): 8 words x {main, bit-3 word} x 2 halves + 2 pad (conflict-free operand reads)
#define L0M_BUF (2 * 2 * L0M_T * L0M_RS)   // 32-bit words per stage buffer
static_assert(2 * L0M_BUF * 4 >= (int)(L0M_QCAP * 24 + L0M_SCAP * 16 + L0M_CAPL * 512 * 16), "the epilogue's queues and lane lists live in the stage buffers");

typedef uint2 l0m_word;  // {main, bit-3 word} of this lane's 32 samples
// X side: bits 0..2 of every nibble stay where they are (E2M1 0.5 / 1 / 2), bit 3 moves to bit 2 of the second word (2)
__device__ __forceinline__ void l0m_perm_x(unsigned h, unsigned &m, unsigned &b)
{
    m = h;
    b = (h >> 1) & 0x44444444u;
}
// Y side: the sample at bit 0 becomes 2, at bit 1 stays 1, at bit 2 becomes 0.5; bit 3 -> 0.5 in the second word
__device__ __forceinline__ void l0m_perm_y(unsigned h, unsigned &m, unsigned &b)
{
    m = ((h & 0x11111111u) << 2) | (h & 0x22222222u) | ((h >> 2) & 0x11111111u);
    b = (h >> 3) & 0x11111111u;
}
template <bool YSIDE>
__device__ __forceinline__ l0m_v8i l0m_expand_fp4(l0m_word w)
{
    l0m_v8i r;
    r[0] = (int)(w.x & (YSIDE ? 0x44444444u : 0x11111111u));
    r[1] = (int)(w.x & 0x22222222u);
    r[2] = (int)(w.x & (YSIDE ? 0x11111111u : 0x44444444u));
    r[3] = (int)w.y;
    r[4] = r[5] = r[6] = r[7] = 0;
    return r;
}

__global__ __launch_bounds__(512) void mi_level0_mfma_kernel(MiDev P, int p, int T, const int32_t *__restrict__ cnt_nz,
                                                            const int32_t *__restrict__ cnt_hi, const double *gthr,
                                                            MiL0Counters *cnt, unsigned long long cap_c, MiCand *__restrict__ cands,
                                                            int dbg, int slot_off, int slot_end /* this launch's share of the tile list, see below */,
                                                            unsigned long long *prof /* FW_L0_VERBOSE: shader cycles per phase, else null */)
{
    // two stage buffers [side][plane][variable][L0M_RS]; the epilogue's queues reuse them (nothing reads a stage after the loop)
    __shared__ __attribute__((aligned(16))) unsigned s_raw[2 * L0M_BUF];
    __shared__ double s_gthr[8];
    __shared__ int4 s_meta[2 * L0M_T];
    __shared__ int s_qn, s_nsw[8], s_nun[8];
    __shared__ unsigned char s_std[2 * L0M_T];
    __shared__ unsigned long long s_qbase;
    MiCand *const s_q = (MiCand *)s_raw;                            // [L0M_QCAP]
    uint4 *const s_surv = (uint4 *)(s_raw + (L0M_QCAP * 24) / 4);  // [L0M_SCAP] {local X | local Y << 8, A | B << 16, C | D << 16, -}: one eighth per wavefront
    static_assert(sizeof(MiCand) == 24, "queue offsets");
    // XCD-aware tile order: consecutive workgroups go round-robin to the eight XCDs, each with its own 4 MB L2.  The tile list is cut
    // into SUPER-TILES of L0M_S x L0M_S tiles (16 x 128 variables x 2 planes x n / 8 bytes = 2.6 MB at n = 5 000: L2-resident) and the
    // workgroups of one XCD (blockIdx & 7) work through the tiles of one super-tile after the other, so a column of bit planes crosses
    // the fabric once per super-tile instead of once per tile (r04: the staging alone moved 25 GB at 2 TB/s, 12 of the kernel's 31 ms).
    // A launch covers the SLOTS [slot_off, slot_end) of the list (slot = super-tile x 64 + tile inside it, row-major; a rank of a
    // sharded run gets a contiguous slot range holding its share of the real tiles).
    const int st = slot_off / (L0M_S * L0M_S) + (int)(blockIdx.x >> 3) / (L0M_S * L0M_S) * 8 + (int)(blockIdx.x & 7);
    const int slot = st * (L0M_S * L0M_S) + (int)(blockIdx.x >> 3) % (L0M_S * L0M_S);
    if (slot < slot_off || slot >= slot_end) return;
    const int TS = (T + L0M_S - 1) / L0M_S;
    int sj = st, si = 0;
    while (sj >= TS - si) {
        sj -= TS - si;
        ++si;
    }
    sj += si;
    const int tin = (int)(blockIdx.x >> 3) % (L0M_S * L0M_S);
    const int bi = si * L0M_S + tin / L0M_S, bj = sj * L0M_S + tin % L0M_S;
    if (bi >= T || bj >= T || bi > bj) return;
    // eight wavefronts, two per SIMD: wavefront (wx, wy) owns X variables [64 wx, +64) x Y variables [32 wy, +32)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wx = wave >> 2, wy = wave & 3;
    const unsigned long long pt0 = prof ? __builtin_readcyclecounter() : 0ull;
    if (tid < 8) s_gthr[tid] = gthr[tid];
    if (tid == 0) s_qn = 0;
    if (tid < 2 * L0M_T) {
        const int g = (tid < L0M_T ? bi : bj) * L0M_T + (tid & (L0M_T - 1));
        const int4 m = g < p ? make_int4(cnt_nz[g], cnt_hi[g], P.levels[g], P.maxv[g]) : make_int4(0, 0, 0, 0);
        s_meta[tid] = m;
        s_std[tid] = (g < p && P.nzmode && P.L == 3 && m.w > 1 && m.z == 3) ? 1 : 0;  // nz-adjusted, three levels: see the epilogue
    }
    l0m_v16f acc[2][2][2];  // [X block][X plane][Y plane]: counts as Float32 (exact)
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q >> 2][(q >> 1) & 1][q & 1][r] = 0;
    // staging: a stage is 2 sides x 2 planes x 128 variables x L0M_WC words = 4 096 words, 8 per thread: load q of thread t is (side
    // q >> 2, plane q & 1, variable 64 ((q >> 1) & 1) + (t >> 3), word t & 7) -- eight consecutive lanes read the 64 contiguous bytes of
    // one (variable, plane); the plane pointer and the row base are the same for the whole wavefront (scalar base) and the lane's offset
    // (t >> 3) W + (t & 7) the same for all loads of the kernel.  No clamps and no selects behind the loads: the planes are allocated
    // with their rows padded to whole tiles and eight words of slack, all zero (fwi_mi_upload), so a tile of the last tile row / column
    // reads zeros; the words beyond W of the last stage belong to the next row and are zeroed WHEN THE STAGE IS WRITTEN TO LDS, a stage
    // after the load.
    const int srow = tid >> 3, sw = tid & 7;
    const unsigned loff8 = (unsigned)(srow * P.W + sw) * 8u;  // (bytes, 32 bits: scalar base + lane offset addressing)
    unsigned long long rr[8];
    auto fetch1 = [&](int q, int w0) {
        // (the base through readfirstlane: a scalar the loop optimiser cannot fold into eight per-lane 64-bit induction pointers)
        const unsigned long long ba = (unsigned long long)(((q & 1) ? P.hi : P.nz) + ((size_t)(((q >> 2) ? bj : bi) * L0M_T + ((q >> 1) & 1) * 64) * P.W + w0));
        // (a GLOBAL-address-space pointer: a flat load counts on the LDS counter as well, and every wait for operand words would wait for it)
        typedef const char __attribute__((address_space(1))) *l0m_gptr;
        const l0m_gptr b = (l0m_gptr)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ba >> 32)) << 32) |
                                      (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)ba));
        rr[q] = *(const unsigned long long __attribute__((address_space(1))) *)(b + loff8);
    };
    // rr[q] (a word of the stage that starts at word w0) -> stage buffer `bufsel`
    unsigned *const sdst = s_raw + srow * L0M_RS + sw * 4;
    auto store1 = [&](int q, int bufsel, int w0) {
        unsigned long long v = rr[q];
        // (the last stage's words beyond W belong to the next row: zeroed on the Y side only -- a product with a zero is zero)
        // (FW_L0_DBG = 2, "no loads": zeros on the Y side make every count zero)
        if (q >> 2) v = (w0 + sw < P.W && !(dbg & 2)) ? v : 0ull;
        unsigned *dst = sdst + bufsel * L0M_BUF + (((q >> 2) * 2 + (q & 1)) * L0M_T + ((q >> 1) & 1) * 64) * L0M_RS;
        unsigned m0, b0, m1, b1;
        if (q >> 2) {
            l0m_perm_y((unsigned)v, m0, b0);
            l0m_perm_y((unsigned)(v >> 32), m1, b1);
        } else {
            l0m_perm_x((unsigned)v, m0, b0);
            l0m_perm_x((unsigned)(v >> 32), m1, b1);
        }
        *(uint2 *)dst = make_uint2(m0, b0);        // half 0: samples 0..31 of the word
        *(uint2 *)(dst + 2) = make_uint2(m1, b1);  // half 1
    };
    const int rowX = wx * 64 + (lane & 31), rowY = wy * 32 + (lane & 31), half = lane >> 5;
    // this lane's operand rows inside a stage buffer: X block 0, X block 1, Y block (plane 0; plane 1 is L0M_T rows further)
    const int offX0 = rowX * L0M_RS, offX1 = (rowX + 32) * L0M_RS, offY = (2 * L0M_T + rowY) * L0M_RS;
#define L0M_LD(buf_, off_, w_) (*(const uint2 *)((buf_) + (off_) + (w_) * 4 + half * 2))
#define L0M_WORDS(dst, buf_, w_)                                                           \
    {                                                                                      \
        _Pragma("unroll") for (int pl_ = 0; pl_ < 2; ++pl_)                                \
        {                                                                                  \
            dst[0][pl_] = L0M_LD(buf_, offX0 + pl_ * L0M_T * L0M_RS, w_);                  \
            dst[1][pl_] = L0M_LD(buf_, offX1 + pl_ * L0M_T * L0M_RS, w_);                  \
            dst[2][pl_] = L0M_LD(buf_, offY + pl_ * L0M_T * L0M_RS, w_);                   \
        }                                                                                  \
    }
#define L0M_EXPAND4(f_, src)                                                               \
    {                                                                                      \
        _Pragma("unroll") for (int pl_ = 0; pl_ < 2; ++pl_)                                \
        {                                                                                  \
            f_[0][pl_] = l0m_expand_fp4<false>(src[0][pl_]);                               \
            f_[1][pl_] = l0m_expand_fp4<false>(src[1][pl_]);                               \
            f_[2][pl_] = l0m_expand_fp4<true>(src[2][pl_]);                                \
        }                                                                                  \
    }
#define L0M_MFMA8F(f_)                                                                     \
    {                                                                                      \
        _Pragma("unroll") for (int a_ = 0; a_ < 2; ++a_) _Pragma("unroll") for (int px_ = 0; px_ < 2; ++px_) \
            _Pragma("unroll") for (int py_ = 0; py_ < 2; ++py_)                            \
                acc[a_][px_][py_] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(f_[a_][px_], f_[2][py_], acc[a_][px_][py_], 4, 4, 0, 127, 0, 127); \
    }
// One word (L0M_HALF): request the operand words of the word after next, multiply the current operands (8 matrix instructions) and,
// under them, expand the words requested one word EARLIER (two sets of raw words: no wait for LDS inside a word) and do the word's
// staging piece -- one of the thread's eight staged words permuted and written to the other buffer, its successor requested.
#ifndef L0M_SCHED
#define L0M_SCHED 5  // vector instructions under one matrix instruction (the probe's free budget with two wavefronts per SIMD)
#endif
#if L0M_SCHED > 0
#define L0M_SCHED8()                                                                       \
    {                                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);                                 \
        _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_)                                   \
        {                                                                                  \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                             \
            __builtin_amdgcn_sched_group_barrier(0x002, L0M_SCHED, 0);                     \
            if (q_ == 3) __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);                \
            if (q_ == 4) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                \
        }                                                                                  \
    }
#else
#define L0M_SCHED8()
#endif
#define L0M_HALF(fcur_, fnext_, wexp_, wload_, buf_, widx_, piece_)                        \
    {                                                                                      \
        L0M_WORDS(wload_, buf_, widx_);                                                    \
        L0M_MFMA8F(fcur_);                                                                 \
        piece_;                                                                            \
        L0M_EXPAND4(fnext_, wexp_);                                                        \
        L0M_SCHED8();                                                                      \
    }
    // Two stage buffers.  Behind the barrier of stage s - 1 nobody reads the other buffer any more, so stage s + 1 is written to it
    // PIECE BY PIECE during the first six words of stage s -- permute, write, request the word of stage s + 2 into the same
    // registers (a whole stage of latency) -- instead of in one lump in front of the barrier, where all eight wavefronts would do
    // nothing but staging at the same time.  The barrier of stage s sits in front of its LAST TWO words: their matrix instructions
    // cover the requests for the first two words of stage s + 1 (word j requests word j + 2, and the last request of stage s into its own
    // buffer -- word 7 -- is made by word 5, in front of the barrier).  The stage body is straight-line code (behind the last stage the
    // pieces write a buffer nobody reads and request the last stage again): with branches around the pieces the compiler sank all 64
    // matrix instructions of a stage behind them.
#pragma unroll
    for (int q = 0; q < 8; ++q) fetch1(q, 0);
#pragma unroll
    for (int q = 0; q < 8; ++q) store1(q, 0, 0);
    __syncthreads();
    const int w0_last = (P.W - 1) / L0M_WC * L0M_WC;
#pragma unroll
    for (int q = 0; q < 8; ++q) fetch1(q, L0M_WC < P.W ? L0M_WC : 0);
    {
        l0m_word wA[3][2], wB[3][2];  // [X block 0, X block 1, Y block][plane]: this lane's 32 samples of its operand rows, two words in flight
        l0m_v8i f0[3][2], f1[3][2];
        int bufsel = 0;
        const unsigned *cur = s_raw;
        L0M_WORDS(wA, cur, 0);
        L0M_WORDS(wB, cur, 1);
        L0M_EXPAND4(f0, wA);
#pragma unroll 1
        for (int w0 = 0; w0 < P.W; w0 += L0M_WC) {
            const unsigned *const nxt = s_raw + (bufsel ^ 1) * L0M_BUF;
            const int w0n = w0 + L0M_WC, w0nn = w0 + 2 * L0M_WC <= w0_last ? w0 + 2 * L0M_WC : w0_last;
#define L0M_PIECE(q_) { store1(q_, bufsel ^ 1, w0n); fetch1(q_, w0nn); }
            L0M_HALF(f0, f1, wB, wA, cur, 2, L0M_PIECE(0));                     // word 0 multiplied, word 1 expanded, word 2 requested
            L0M_HALF(f1, f0, wA, wB, cur, 3, { L0M_PIECE(1); L0M_PIECE(2); });
            L0M_HALF(f0, f1, wB, wA, cur, 4, L0M_PIECE(3));
            L0M_HALF(f1, f0, wA, wB, cur, 5, { L0M_PIECE(4); L0M_PIECE(5); });
            L0M_HALF(f0, f1, wB, wA, cur, 6, L0M_PIECE(6));
            L0M_HALF(f1, f0, wA, wB, cur, 7, L0M_PIECE(7));
#undef L0M_PIECE
            __syncthreads();
            L0M_HALF(f0, f1, wB, wA, nxt, 0, {});  // word 6 multiplied, word 7 expanded, word 0 of the next stage requested
            L0M_HALF(f1, f0, wA, wB, nxt, 1, {});  // word 7 multiplied, word 0 of the next stage expanded, its word 1 requested
            cur = nxt;
            bufsel ^= 1;
        }
    }
#undef L0M_HALF
#undef L0M_SCHED8
#undef L0M_EXPAND4
#undef L0M_MFMA8F
#undef L0M_LD
#undef L0M_WORDS
    __syncthreads();
    const unsigned long long pt1 = prof ? __builtin_readcyclecounter() : 0ull;
    if (dbg & 1) {
        if ((int)acc[0][0][0][0] + (int)acc[1][1][1][15] == -12345) cnt->n_sig = 1;
        return;
    }
    // Epilogue in two passes.  Pass 1 decides the pair every HE table is made of -- both variables nz-adjusted with three levels
    // ("standard", one flag per variable) -- on integers and one Float32 inequality, BRANCH-FREE (the rules of mi_pair_prescreen as
    // selects), reading the accumulators in place (fully unrolled).  Pairs that need the full screen -- and every pair with a
    // non-standard variable -- go to a survivor list in LDS: each wavefront fills its own eighth (ballot + lane rank: no atomics, no
    // waits).  Pass 2 walks the lists densely with the full screen (hardware logarithms instead of table look-ups).  A pair that finds
    // its wavefront's list full goes to the exact kernel unscreened (that kernel counts unreliable pairs among its input itself).
    // (One pass, pair by pair: every wavefront waited for the global-memory look-ups of its slowest lane in each of its steps --
    // 31 ms at cfg4.)
    int n_unrel = 0;
    const bool pre_ok = (long long)P.n >= P.n_obs_min && (long long)P.n > (long long)P.hps;  // tests.jl:9-20 for two three-level variables
    const long long thrA64 = P.n_obs_min > (long long)P.hps * 4 + 1 ? P.n_obs_min : (long long)P.hps * 4 + 1;
    const int thrA = thrA64 > 0x7fffffffll ? 0x7fffffff : (int)thrA64;  // reliable <=> A >= thrA
    const float kthr = 0.98f * (float)s_gthr[1];
    int my_ns = 0;  // survivors of this wavefront so far (wave-uniform)
    const int seg0 = wave * (L0M_SCAP / 8);
    int n_unrel_w = 0;  // wave-uniform count of unreliable pairs (popcounts of lane masks: no vector adds)
    // FAST tiles -- off the diagonal, inside the table, every variable standard: all but the last tile row / column and the
    // diagonal -- need neither the index tests nor the flags.
    const bool fast_tile = __syncthreads_and(tid >= 2 * L0M_T || s_std[tid & (2 * L0M_T - 1)] != 0) && bi < bj && (bj + 1) * L0M_T <= p;
    const int lY = wy * 32 + (lane & 31);
    const int Y = bj * L0M_T + lY;
    auto pass1 = [&](auto fast_c) {
        constexpr bool FAST = decltype(fast_c)::value;
        const bool stdY = FAST || s_std[L0M_T + lY] != 0;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            unsigned stdX = 0u;  // bit r: the X variable of accumulator register r is standard
            if (!FAST) {
#pragma unroll
                for (int r = 0; r < 16; ++r) stdX |= (unsigned)s_std[wx * 64 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * half] << r;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lX = wx * 64 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int X = bi * L0M_T + lX;
                const int cA = (int)acc[a][0][0][r], cB = (int)acc[a][1][0][r], cC = (int)acc[a][0][1][r], cD = (int)acc[a][1][1][r];  // A = <nzX, nzY>, B = <hiX, nzY>, C = <nzX, hiY>, D = <hiX, hiY>
                const bool valid = FAST || (X < Y && Y < p);
                const bool stdp = FAST || (stdY && ((stdX >> r) & 1u));
                const bool rel = pre_ok && cA >= thrA;
                // |A D - B C| from one fused multiply-add: A D enters exactly, B C rounded to 24 bits -- the bound adds that
                // rounding (<= 2^-24 B C) and the result's own (2^-24 |det|), so a pair is only dropped if its exact 2 X^2 is
                // below the threshold (counts <= 65 535 are exact Float32 values).  An empty marginal (df = 0, p = 1) makes the
                // right-hand side zero.
                const float fA = (float)cA, fB = (float)cB, fC = (float)cC, fD = (float)cD;
                const float bc = fB * fC;
                const float det = fabsf(__builtin_fmaf(fA, fD, -bc)) * 1.0000003f + 6.0e-8f * bc;
                const float lhs = 2.0f * fA * det * det, rhs = ((fA - fB) * fB) * ((fA - fC) * fC);
                const bool pass = rhs > 0.0f && !(lhs < kthr * rhs);
                const bool surv = valid && (!stdp || (rel && pass));
                n_unrel_w += __builtin_popcountll(__builtin_amdgcn_ballot_w64(valid && stdp && !rel));
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(surv);
                if (bal) {
                    const int slot = my_ns + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                    if (surv) {
                        if (slot < L0M_SCAP / 8) {
                            s_surv[seg0 + slot] = make_uint4((unsigned)lX | ((unsigned)lY << 8), (unsigned)cA | ((unsigned)cB << 16), (unsigned)cC | ((unsigned)cD << 16), 0u);
                        } else {  // list full (rare): unscreened to the exact kernel
                            MiCand cd;
                            cd.X = X, cd.Y = Y, cd.A = cA, cd.B = cB, cd.C = cC, cd.D = cD;
                            const int qs = atomicAdd(&s_qn, 1);
                            if (qs < L0M_QCAP) {
                                s_q[qs] = cd;
                            } else {
                                const unsigned long long gs = atomicAdd(&cnt->n_sig, 1ull);
                                if (gs < cap_c) cands[gs] = cd;
                            }
                        }
                    }
                    my_ns += __builtin_popcountll(bal);
                }
            }
        }
    };
    // FAST tiles, r05: the same verdicts on PAIRS of accumulator registers with packed Float32 instructions (v_pk_mul / v_pk_fma /
    // v_pk_add: the counts ARE Float32 values, no conversions), and survivors appended to a list per LANE (no ballot, no lane rank, no
    // wave-uniform bookkeeping per accumulator register: one compare-and-add for the unreliable count, an exec-masked 16-byte LDS
    // write for a survivor -- the accumulator register's index rides in the five low mantissa bits of the count A, which are zero for
    // an integer below 2^19); the lanes' lists are compacted into the wavefront's segment of s_surv afterwards.  The bound is the
    // r04 one with the cross term of det^2 = (|x| k1 + k2 bc)^2 split by 2 |x| bc <= x^2 + bc^2 (no absolute value, which the packed
    // instructions lack): 2 A det^2 <= A (K1 x^2 + K2 bc^2).  A lane with more than L0M_CAPL survivors (a hub column) sends its whole
    // wavefront through the general form above.
    bool packed_done = false;
    if (fast_tile) {
        typedef float l0m_f2 __attribute__((ext_vector_type(2)));
        float *const s_lane = (float *)(s_raw + (L0M_QCAP * 24 + L0M_SCAP * 16) / 4);  // [4 values][L0M_CAPL][512]: four 4-byte writes with one address (a 16-byte write wants its values in consecutive registers: three copies per survivor)
        constexpr int PL = L0M_CAPL * 512;
        const float fthrA = pre_ok ? (float)(thrA < (1 << 24) ? thrA : (1 << 24)) : 3.0e38f;
        const l0m_f2 K1 = {2.000003f, 2.000003f}, K2 = {1.3e-7f, 1.3e-7f}, KT = {kthr, kthr};
        int lcnt = 0, unrel = 0;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r2 = 0; r2 < 8; ++r2) {
                const l0m_f2 A = {acc[a][0][0][2 * r2], acc[a][0][0][2 * r2 + 1]}, B = {acc[a][1][0][2 * r2], acc[a][1][0][2 * r2 + 1]};
                const l0m_f2 C = {acc[a][0][1][2 * r2], acc[a][0][1][2 * r2 + 1]}, D = {acc[a][1][1][2 * r2], acc[a][1][1][2 * r2 + 1]};
                const l0m_f2 bc = B * C;
                const l0m_f2 x = __builtin_elementwise_fma(A, D, -bc);
                const l0m_f2 lhs = A * __builtin_elementwise_fma(x * x, K1, (bc * bc) * K2);
                const l0m_f2 rhs = ((A - B) * B) * ((A - C) * C);
                const l0m_f2 kr = rhs * KT;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const bool rel = A[e] >= fthrA;
                    const bool surv = rel && rhs[e] > 0.0f && !(lhs[e] < kr[e]);
                    unrel += rel ? 0 : 1;
                    if (surv) {
                        if (lcnt < L0M_CAPL) {
                            float *d = s_lane + lcnt * 512 + tid;
                            d[0] = __uint_as_float(__float_as_uint(A[e]) | (unsigned)(a * 16 + 2 * r2 + e));
                            d[PL] = B[e];
                            d[2 * PL] = C[e];
                            d[3 * PL] = D[e];
                        }
                        ++lcnt;
                    }
                }
            }
        if (__builtin_amdgcn_ballot_w64(lcnt > L0M_CAPL) == 0ull) {
            packed_done = true;
            n_unrel += unrel;
            int incl = lcnt;  // inclusive scan of the lanes' counts
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            my_ns = __builtin_amdgcn_readlane(incl, 63);
            const int off = incl - lcnt;
            for (int k = 0; k < lcnt; ++k) {
                const float *en = s_lane + k * 512 + tid;
                const unsigned ab = __float_as_uint(en[0]), tag = ab & 31u;
                const int cA = (int)__uint_as_float(ab & ~31u), cB = (int)en[PL], cC = (int)en[2 * PL], cD = (int)en[3 * PL];
                const int r = (int)(tag & 15u), lX = wx * 64 + 32 * (int)(tag >> 4) + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (off + k < L0M_SCAP / 8) {
                    s_surv[seg0 + off + k] = make_uint4((unsigned)lX | ((unsigned)lY << 8), (unsigned)cA | ((unsigned)cB << 16), (unsigned)cC | ((unsigned)cD << 16), 0u);
                } else {  // segment full (rare): unscreened to the exact kernel
                    MiCand cd;
                    cd.X = bi * L0M_T + lX, cd.Y = Y, cd.A = cA, cd.B = cB, cd.C = cC, cd.D = cD;
                    const int qs = atomicAdd(&s_qn, 1);
                    if (qs < L0M_QCAP) {
                        s_q[qs] = cd;
                    } else {
                        const unsigned long long gs = atomicAdd(&cnt->n_sig, 1ull);
                        if (gs < cap_c) cands[gs] = cd;
                    }
                }
            }
        }
    }
    if (!packed_done) pass1(std::false_type{});
    if (lane == 0) n_unrel += n_unrel_w;
    if (lane == 0) s_nsw[wave] = my_ns < L0M_SCAP / 8 ? my_ns : L0M_SCAP / 8;
    const unsigned long long pt2 = prof ? __builtin_readcyclecounter() : 0ull;
    __syncthreads();
    {   // the eight lists as one: thread t takes entries t, t + 512, ... of their concatenation
        int cum[9];
        cum[0] = 0;
#pragma unroll
        for (int sg = 0; sg < 8; ++sg) cum[sg + 1] = cum[sg] + s_nsw[sg];
        const int tot = (dbg & 8) ? 0 : cum[8];
        const double thr1_std = 0.99 * s_gthr[1] - (0.05 + 1.5 * (2.0 * 0.6931471805599453 * 16.0 * (double)P.n * (double)__log2f((float)P.n) * 2.384185791015625e-07));
        for (int g = tid; g < tot; g += 512) {
            int sg = 0;
#pragma unroll
            for (int q = 1; q < 8; ++q) sg += g >= cum[q];
            int base = 0;
#pragma unroll
            for (int q = 1; q < 8; ++q) base = (q == sg) ? cum[q] : base;
            const uint4 e = s_surv[sg * (L0M_SCAP / 8) + (g - base)];
            const int lX = (int)(e.x & 0xffu), lYq = (int)(e.x >> 8);
            if (fast_tile)  // (workgroup-uniform) every entry is a standard, reliable pair with df = 1
                mi_pair_screen_std(bi * L0M_T + lX, bj * L0M_T + lYq, (int)(e.y & 0xffffu), (int)(e.y >> 16), (int)(e.z & 0xffffu), (int)(e.z >> 16), thr1_std,
                                   cnt, cap_c, cands, s_q, &s_qn, L0M_QCAP);
            else
                n_unrel += mi_pair_screen(P, s_meta[lX], s_meta[L0M_T + lYq], bi * L0M_T + lX, bj * L0M_T + lYq, (int)(e.y & 0xffffu), (int)(e.y >> 16),
                                          (int)(e.z & 0xffffu), (int)(e.z >> 16), (const float *)nullptr, (const float *)nullptr, s_gthr, cnt, cap_c, cands,
                                          s_q, &s_qn, L0M_QCAP);
        }
    }
    const unsigned long long pt3 = prof ? __builtin_readcyclecounter() : 0ull;
    n_unrel = wave_sum_i(n_unrel);
    if (lane == 0) s_nun[wave] = n_unrel;
    __syncthreads();
    const int nq = s_qn < L0M_QCAP ? s_qn : L0M_QCAP;
    if (tid == 0) {  // one atomic per tile and counter (eight wavefronts x 76 000 tiles on one address queued up behind each other)
        int nu = 0;
        for (int q = 0; q < 8; ++q) nu += s_nun[q];
        if (nu) atomicAdd(&cnt->n_unreliable, (unsigned long long)nu);
        if (nq > 0) s_qbase = atomicAdd(&cnt->n_sig, (unsigned long long)nq);
    }
    __syncthreads();
    for (int q = tid; q < nq; q += 512)
        if (s_qbase + (unsigned long long)q < cap_c) cands[s_qbase + q] = s_q[q];
    if (prof && tid == 0) {
        atomicAdd(prof + 0, pt1 - pt0);
        atomicAdd(prof + 1, pt2 - pt1);
        atomicAdd(prof + 2, pt3 - pt2);
        atomicAdd(prof + 3, __builtin_readcyclecounter() - pt3);
        atomicAdd(prof + 4, 1ull);
        int nsv = 0;
        for (int q = 0; q < 8; ++q) nsv += s_nsw[q];
        atomicAdd(prof + 5, (unsigned long long)nsv);
    }
}

// Kernel 2.  A workgroup takes L0X_IT x 256 consecutive candidates, gathers the significant ones in LDS and appends them to the output
// with ONE atomic (r04 / early r05: one same-address atomic per wavefront -- 220 000 of them at cfg4 queued up behind each other in
// the L2 and were most of the kernel's 5.3 ms).  Unreliable pairs are counted by the screening kernel -- except those it hands over
// UNSCREENED (mi_level0_mfma_kernel when a survivor list is full): a screened candidate is reliable by construction, so whatever is
// unreliable here has not been counted.  The order of the output does not matter (BH sorts, the neighbour lists are keyed).
struct MiL0Rec {
    int32_t X, Y;
    double stat, pval;
};
__global__ __launch_bounds__(256) void mi_level0_exact_kernel(MiDev P, const MiCand *__restrict__ cands, unsigned long long ncand,
                                                              const int32_t *__restrict__ cnt_nz, const int32_t *__restrict__ cnt_hi,
                                                              double alpha, const double *gthr, MiL0Counters *cnt,
                                                              unsigned long long cap, int32_t *out_i, int32_t *out_j,
                                                              double *out_s, double *out_p)
{
    __shared__ MiL0Rec s_rec[L0X_IT * 256];
    __shared__ unsigned int s_n, s_unrel;
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) s_n = 0u, s_unrel = 0u;
    __syncthreads();
    const unsigned long long t0 = (unsigned long long)blockIdx.x * (L0X_IT * 256);
    unsigned int my_unrel = 0u;
#pragma unroll 1
    for (int it = 0; it < L0X_IT; ++it) {
        const unsigned long long t = t0 + (unsigned long long)it * 256 + tid;
        int verdict = 0;
        MiL0Rec r;
        r.X = r.Y = 0;
        r.stat = 0.0, r.pval = 1.0;
        if (t < ncand) {
            const MiCand c = cands[t];
            r.X = c.X, r.Y = c.Y;
            verdict = mi_pair_epilogue(P, c.X, c.Y, c.A, c.B, c.C, c.D, cnt_nz, cnt_hi, alpha, gthr, r.stat, r.pval);
        }
        my_unrel += (unsigned)(verdict >> 1);
        const unsigned long long km = __ballot(verdict & 1);  // one LDS atomic per wavefront
        if (km != 0ull) {
            unsigned int base = 0u;
            const int leader = __ffsll((long long)km) - 1;
            if (lane == leader) base = atomicAdd(&s_n, (unsigned int)__popcll(km));
            base = __shfl(base, leader);
            if (verdict & 1) s_rec[base + (unsigned int)__popcll(km & ((1ull << lane) - 1ull))] = r;
        }
    }
    my_unrel = (unsigned)wave_sum_i((int)my_unrel);
    if (lane == 0 && my_unrel) atomicAdd(&s_unrel, my_unrel);
    __syncthreads();
    const unsigned int n = s_n;
    if (tid == 0) {
        if (n) s_base = atomicAdd(&cnt->n_sig, (unsigned long long)n);
        if (s_unrel) atomicAdd(&cnt->n_unreliable, (unsigned long long)s_unrel);
    }
    __syncthreads();
    const unsigned long long base = s_base;
    for (unsigned int k = tid; k < n; k += 256) {
        const unsigned long long slot = base + k;
        if (slot < cap) {
            const MiL0Rec r = s_rec[k];
            out_i[slot] = r.X;
            out_j[slot] = r.Y;
            out_s[slot] = r.stat;
            out_p[slot] = r.pval;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
MiDev fwi_mi_dev(const fw_ctx *ctx) { return mi_dev(ctx); }

// test_subsets paths (the call edge hiton.jl:100): under the dense rules the reference hands them a row view of the data
static MiDev mi_dev_subsets(const fw_ctx *ctx)
{
    MiDev P = mi_dev(ctx);
    P.view = P.dense && P.nzmode && ctx->mi_view;
    return P;
}

static MiDev mi_dev(const fw_ctx *ctx)
{
    MiDev P;
    P.nz = (const unsigned long long *)ctx->d_nzbits;
    P.hi = (const unsigned long long *)ctx->d_hibits;
    P.levels = ctx->d_levels;
    P.maxv = ctx->d_maxvals;
    P.W = ctx->W;
    P.n = ctx->P.n;
    P.L = ctx->L;
    P.nzmode = ctx->P.kind == FW_MI_NZ;
    P.hps = ctx->P.hps;
    P.dense = ctx->P.dense_rules != 0;
    P.n_obs_min = ctx->n_obs_min_eff;
    P.gthr = ctx->d_gthr;
    P.gthr_n = ctx->gthr_n;
    P.alpha = ctx->P.alpha;
    P.prof = nullptr;
    P.view = 0;
    P.vals = ctx->mi_generic ? ctx->d_vals : nullptr;
    P.gtab = nullptr;
    P.gtab_words = 0ull;
    static const int rowk = fw_knob("FW_MI_ROWK") ? atoi(fw_knob("FW_MI_ROWK")) : 2;  // 99: popcount form only (A/B)
    P.rowk = rowk;
    return P;
}

// ---- generic form: a value above 2 somewhere (mi_test_core_gen) -------------------------------------------------------------
// One byte per (variable, sample); levels / max_vals as misc.jl:64-97 computes them (distinct values of the column, the implicit
// zero included); L = maximum(max_vals) + 1 for every table of the context (types.jl:89,110).  The conditional tests run through the
// same batch / segment kernels as the bit-plane forms (template value L = 0), HITON-PC through the host job pool, level 0 through
// mig_level0_kernel (one wavefront per pair).  Limits: values <= 7, L^max_k (L^2 + 1) <= MIG_TAB32 words of LDS per wavefront.
static double host_igamc(double a, double x);
#define MIG_GTAB_MAX (64ll << 20)          // words of one device-memory table
#define MIG_GTAB_BYTES (1ll << 30)         // device memory the tables of ONE launch may take: launches are cut to fit
// workgroups (of four wavefronts, one table each) a launch of the generic form may hold: all of them with LDS tables, else what fits
static int64_t mig_launch_wgs(fw_ctx *ctx, int64_t want)
{
    if (!ctx->mi_generic || ctx->mig_gtab_words == 0) return want;
    const int64_t fit = MIG_GTAB_BYTES / (4 * 4 * ctx->mig_gtab_words);
    return std::max<int64_t>(1, std::min<int64_t>(want, fit));
}
static int mig_dev_tables(fw_ctx *ctx, MiDev &P, int64_t wgs, int which /* 0: the engine's stream, 1 / 2: pool 0 / 1 */)
{
    if (!ctx->mi_generic || ctx->mig_gtab_words == 0) return FW_OK;
    if (int rc = fw_dev_reserve(ctx, ctx->d_mig_tab[which], (size_t)wgs * 4 * (size_t)ctx->mig_gtab_words * sizeof(unsigned))) return rc;
    P.gtab = (unsigned *)ctx->d_mig_tab[which].ptr;
    P.gtab_words = (unsigned long long)ctx->mig_gtab_words;
    return FW_OK;
}
static int mig_upload(fw_ctx *ctx, const int64_t *colptr, const int32_t *rowval, const int32_t *nzval)
{
    const int n = ctx->P.n, p = ctx->P.p;
    std::vector<unsigned char> vals((size_t)p * n, 0);
    int maxv_all = 0;
    for (int v = 0; v < p; ++v) {
        bool seen[MIG_MAX_L] = {false};
        int mx = 0;
        for (int64_t j = colptr[v]; j < colptr[v + 1]; ++j) {
            const int32_t x = nzval[j];
            vals[(size_t)v * n + rowval[j]] = (unsigned char)x;
            seen[x] = true;
            mx = std::max(mx, (int)x);
        }
        int lev = n > colptr[v + 1] - colptr[v] ? 1 : 0;
        for (int q = 1; q < MIG_MAX_L; ++q) lev += seen[q] ? 1 : 0;
        ctx->levels[v] = lev;
        ctx->max_vals[v] = mx;
        maxv_all = std::max(maxv_all, mx);
    }
    ctx->L = maxv_all + 1;
    ctx->mi_nxy = ctx->L;
    ctx->W = (n + 63) / 64;
    long long strata = 1;
    for (int j = 0; j < ctx->P.max_k; ++j) strata *= ctx->L;
    // tables beyond the LDS of a wavefront live in device memory (r05); beyond MIG_GTAB_MAX words (256 MB) per table: the limit
    const long long tab_words = strata * ((long long)ctx->L * ctx->L + 1);
    if (ctx->L > MIG_MAX_L || tab_words > MIG_GTAB_MAX)
        return fw_fail(ctx, FW_ERR_LIMIT, "discrete data with %d levels and max_k = %d needs %lld table words per test (limits: %d levels, %lld words): lower max_k or merge levels",
                       ctx->L, ctx->P.max_k, tab_words, MIG_MAX_L, (long long)MIG_GTAB_MAX);
    ctx->mig_gtab_words = tab_words > MIG_TAB32 ? ((tab_words + 63) & ~63ll) : 0;
    void **ptrs[] = {(void **)&ctx->d_nzbits, (void **)&ctx->d_hibits, (void **)&ctx->d_levels, (void **)&ctx->d_maxvals, (void **)&ctx->d_firstnz, (void **)&ctx->d_vals,
                     (void **)&ctx->d_gthr};
    for (void **q : ptrs)
        if (*q) {
            (void)hipFree(*q);
            *q = nullptr;
        }
    FW_HIP(ctx, hipMalloc((void **)&ctx->d_vals, vals.size()));
    FW_HIP(ctx, hipMemcpy(ctx->d_vals, vals.data(), vals.size(), hipMemcpyHostToDevice));
    FW_HIP(ctx, hipMalloc((void **)&ctx->d_levels, sizeof(int32_t) * p));
    FW_HIP(ctx, hipMalloc((void **)&ctx->d_maxvals, sizeof(int32_t) * p));
    FW_HIP(ctx, hipMemcpy(ctx->d_levels, ctx->levels.data(), sizeof(int32_t) * p, hipMemcpyHostToDevice));
    FW_HIP(ctx, hipMemcpy(ctx->d_maxvals, ctx->max_vals.data(), sizeof(int32_t) * p, hipMemcpyHostToDevice));
    {  // alpha quantiles of G^2 per df: df <= (L - 1)^2 per stratum
        const int ndf = (int)std::min<long long>((long long)(ctx->L - 1) * (ctx->L - 1) * strata + 1, 4096);
        std::vector<double> q((size_t)ndf, 1e300);
        for (int df = 1; df < ndf; ++df) {
            double lo = 0.0, hi = 16.0 + 4.0 * df;
            while (host_igamc(0.5 * df, 0.5 * hi) >= ctx->P.alpha) hi *= 2.0;
            for (int it = 0; it < 64; ++it) {
                const double mid = 0.5 * (lo + hi);
                if (host_igamc(0.5 * df, 0.5 * mid) < ctx->P.alpha)
                    hi = mid;
                else
                    lo = mid;
            }
            q[df] = 0.5 * (lo + hi);
        }
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_gthr, sizeof(double) * (size_t)ndf));
        FW_HIP(ctx, hipMemcpy(ctx->d_gthr, q.data(), sizeof(double) * (size_t)ndf, hipMemcpyHostToDevice));
        ctx->gthr_n = ndf;
    }
    ctx->mi_generic = true;
    return FW_OK;
}

// level 0 of the generic form: wavefront w of the launch tests pair q0 + w (pairs linearised row by row over the upper triangle);
// tests.jl:80-92 (everything fails if levels[X] < 2) + the scalar test, kept if reliable and p < alpha
__global__ __launch_bounds__(256) void mig_level0_kernel(MiDev P, int p, long long q0, long long q1, double alpha, MiL0Counters *cnt,
                                                         unsigned long long cap, int32_t *out_i, int32_t *out_j, double *out_s, double *out_p)
{
    __shared__ unsigned short s_tab[4][2 * MIG_TAB32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long q = q0 + (long long)blockIdx.x * 4 + wave;
    if (q >= q1) return;
    // q -> (i, j), i < j: row i starts at offset i p - i (i + 1) / 2
    const double pd = (double)p - 0.5;
    long long i = (long long)(pd - sqrt(pd * pd - 2.0 * (double)q));
    if (i < 0) i = 0;
    while (i > 0 && i * (long long)p - i * (i + 1) / 2 > q) --i;
    while ((i + 1) * (long long)p - (i + 1) * (i + 2) / 2 <= q) ++i;
    const long long j = q - (i * (long long)p - i * (i + 1) / 2) + i + 1;
    MiRes r;
    r.stat = 0.0;
    r.pval = 1.0;
    r.df = 0;
    r.power = 0;
    r.g = 0.0;
    r.n_obs = 0;
    if (P.levels[(int)i] >= 2) {
        MiZs zs;
#pragma unroll
        for (int t = 0; t < MI_MAX_K; ++t) zs.v[t] = 0;
        r = mi_test_core_gen(P, (int)i, (int)j, zs, 0, (unsigned *)s_tab[wave]);
    }
    double pv = 1.0;
    if (r.power) pv = mi_res_pval(r);
    if (lane == 0) {
        if (!r.power) {
            atomicAdd(&cnt->n_unreliable, 1ull);
        } else if (pv < alpha) {
            const unsigned long long slot = atomicAdd(&cnt->n_sig, 1ull);
            if (slot < cap) {
                out_i[slot] = (int32_t)i;
                out_j[slot] = (int32_t)j;
                out_s[slot] = r.stat;
                out_p[slot] = pv;
            }
        }
    }
}

static int mig_level0(fw_ctx *ctx, std::vector<int32_t> &pi, std::vector<int32_t> &pj, std::vector<double> &stat, std::vector<double> &pval,
                      int64_t *m_reliable, FwL0Dev *dev)
{
    const int p = ctx->P.p;
    const long long npairs = (long long)p * (p - 1) / 2;
    const long long q0 = npairs * ctx->l0_rank / ctx->l0_world, q1 = npairs * (ctx->l0_rank + 1) / ctx->l0_world;
    const MiDev P = mi_dev(ctx);
    unsigned long long cap = (unsigned long long)std::max<long long>(std::min<long long>(q1 - q0, 1ll << 22), 1);
    if (cap < ctx->l0_cap_hint) cap = ctx->l0_cap_hint;
    MiL0Counters h{};
    int rc;
    int32_t *oi = nullptr, *oj = nullptr;
    double *os = nullptr, *op = nullptr;
    for (int attempt = 0;; ++attempt) {
        if ((rc = fw_dev_reserve(ctx, ctx->d_tmp0, sizeof(MiL0Counters)))) return rc;
        if ((rc = fw_dev_reserve(ctx, ctx->d_tmp1, cap * 2 * sizeof(int32_t)))) return rc;
        if ((rc = fw_dev_reserve(ctx, ctx->d_tmp2, cap * 2 * sizeof(double)))) return rc;
        oi = (int32_t *)ctx->d_tmp1.ptr;
        oj = oi + cap;
        os = (double *)ctx->d_tmp2.ptr;
        op = os + cap;
        FW_HIP(ctx, hipMemsetAsync(ctx->d_tmp0.ptr, 0, sizeof(MiL0Counters), ctx->stream));
        if (q1 > q0)
            hipLaunchKernelGGL(mig_level0_kernel, dim3((unsigned)((q1 - q0 + 3) / 4)), dim3(256), 0, ctx->stream, P, p, q0, q1, ctx->P.alpha,
                               (MiL0Counters *)ctx->d_tmp0.ptr, cap, oi, oj, os, op);
        FW_HIP(ctx, hipGetLastError());
        FW_HIP(ctx, hipMemcpyAsync(&h, ctx->d_tmp0.ptr, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->cnt.kernel_launches += 1;
        if (h.n_sig > ctx->l0_cap_hint) ctx->l0_cap_hint = h.n_sig;
        if (h.n_sig <= cap) break;
        if (attempt == 1) return fw_fail(ctx, FW_ERR_DEVICE, "discrete level-0 (generic form): result buffer overflow twice");
        cap = h.n_sig;
    }
    *m_reliable = npairs - (long long)h.n_unreliable;  // (sharded runs: this rank's share; fw_level0_sharded adds the ranks' counts up)
    const size_t k = (size_t)h.n_sig;
    pi.clear();
    pj.clear();
    stat.clear();
    pval.clear();
    if (dev) {
        *dev = FwL0Dev{};
        dev->i = oi;
        dev->j = oj;
        dev->stat64 = os;
        dev->pval = op;
        dev->k = k;
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
    return FW_OK;
}

int fwi_mi_upload(fw_ctx *ctx, const int64_t *colptr, const int32_t *rowval, const int32_t *nzval)
{
    const int n = ctx->P.n, p = ctx->P.p;
    if (ctx->P.max_k > MI_MAX_K) return fw_fail(ctx, FW_ERR_LIMIT, "discrete tests support max_k <= %d (got %d)", MI_MAX_K, ctx->P.max_k);
    const int W = (n + 63) / 64;
    std::vector<uint64_t> nzb((size_t)p * W, 0), hib((size_t)p * W, 0);
    std::vector<int32_t> cnt_nz(p, 0), cnt_hi(p, 0);
    ctx->levels.assign(p, 0);
    ctx->max_vals.assign(p, 0);
    int maxv_all = 0;
    bool generic = false;
    ctx->mi_generic = false;
    for (int v = 0; v < p; ++v) {
        if (colptr[v + 1] < colptr[v]) return fw_fail(ctx, FW_ERR_ARG, "colptr not monotone at column %d", v);
        bool seen[4] = {false, false, false, false};
        int32_t mx = 0;
        int64_t prev_row = -1;
        for (int64_t j = colptr[v]; j < colptr[v + 1]; ++j) {
            const int32_t r = rowval[j], x = nzval[j];
            if (r < 0 || r >= n || r <= prev_row) return fw_fail(ctx, FW_ERR_ARG, "row indices of column %d are not sorted / in range", v);
            prev_row = r;
            if (x < 1 || x >= MIG_MAX_L)
                return fw_fail(ctx, FW_ERR_LIMIT, "discrete values must be 0 .. %d (column %d holds %d); stored zeros are not allowed", MIG_MAX_L - 1, v, x);
            if (x > 2) {  // more than three levels somewhere: the generic form below
                generic = true;
                continue;
            }
            seen[x] = true;
            mx = std::max(mx, x);
            nzb[(size_t)v * W + (r >> 6)] |= 1ull << (r & 63);
            ++cnt_nz[v];
            if (x == 2) {
                hib[(size_t)v * W + (r >> 6)] |= 1ull << (r & 63);
                ++cnt_hi[v];
            }
        }
        const int64_t nnz = colptr[v + 1] - colptr[v];
        // misc.jl:64-72 / :84-87
        ctx->levels[v] = (int32_t)((seen[1] ? 1 : 0) + (seen[2] ? 1 : 0) + (n > nnz ? 1 : 0));
        ctx->max_vals[v] = mx;
        maxv_all = std::max(maxv_all, (int)mx);
    }
    if (generic) return mig_upload(ctx, colptr, rowval, nzval);
    ctx->L = maxv_all + 1;  // types.jl:89,110
    if (ctx->L < 2) ctx->L = 2;
    ctx->mi_nxy = (ctx->L == 3 && ctx->P.kind == FW_MI) ? 3 : 2;
    if (int rcb = fwi_mi_big_limits(ctx, ctx->P.max_k)) return rcb;
    // 9-cell tables only where X / Y can take three values inside the sub-table: "mi" on data that holds the value 2
    // (nz-adjusted tests drop the zero level of such a variable, presence / absence data has two values anyway)
    ctx->mi_nxy = (ctx->L == 3 && ctx->P.kind == FW_MI) ? 3 : 2;
    ctx->W = W;
    const size_t pb = sizeof(uint64_t) * (size_t)p * W;
    void **ptrs[] = {(void **)&ctx->d_nzbits, (void **)&ctx->d_hibits, (void **)&ctx->d_levels, (void **)&ctx->d_maxvals, (void **)&ctx->d_firstnz};
    for (void **q : ptrs)
        if (*q) {
            (void)hipFree(*q);
            *q = nullptr;
        }
    // (rows padded to whole level-0 tiles plus one stage of slack words, zero: mi_level0_mfma_kernel stages without clamps)
    const size_t pb_alloc = sizeof(uint64_t) * ((size_t)((p + L0M_T - 1) / L0M_T * L0M_T) * W + L0M_WC);
    FW_HIP(ctx, hipMalloc((void **)&ctx->d_nzbits, pb_alloc));
    FW_HIP(ctx, hipMemset(ctx->d_nzbits, 0, pb_alloc));
    FW_HIP(ctx, hipMemcpy(ctx->d_nzbits, nzb.data(), pb, hipMemcpyHostToDevice));
    if (ctx->L > 2) {
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_hibits, pb_alloc));
        FW_HIP(ctx, hipMemset(ctx->d_hibits, 0, pb_alloc));
        FW_HIP(ctx, hipMemcpy(ctx->d_hibits, hib.data(), pb, hipMemcpyHostToDevice));
    }
    FW_HIP(ctx, hipMalloc((void **)&ctx->d_levels, sizeof(int32_t) * p));
    FW_HIP(ctx, hipMalloc((void **)&ctx->d_maxvals, sizeof(int32_t) * p));
    FW_HIP(ctx, hipMemcpy(ctx->d_levels, ctx->levels.data(), sizeof(int32_t) * p, hipMemcpyHostToDevice));
    FW_HIP(ctx, hipMemcpy(ctx->d_maxvals, ctx->max_vals.data(), sizeof(int32_t) * p, hipMemcpyHostToDevice));
    {  // Float32 tables T[x] = x ln x and ln x, x = 0..n, for the level-0 screen (mi_pair_screen)
        std::vector<float> tab(2 * ((size_t)n + 1), 0.0f);
        for (int x = 1; x <= n; ++x) {
            tab[x] = (float)((double)x * std::log((double)x));
            tab[(size_t)n + 1 + x] = (float)std::log((double)x);
        }
        if (ctx->d_xlnx) (void)hipFree(ctx->d_xlnx);
        ctx->d_xlnx = nullptr;
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_xlnx, tab.size() * sizeof(float)));
        FW_HIP(ctx, hipMemcpy(ctx->d_xlnx, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    {  // alpha quantiles of G^2 per df (mi_issig): df <= (levels - 1)^2 per stratum, L^max_k strata
        int strata = 1;
        for (int j = 0; j < ctx->P.max_k; ++j) strata *= ctx->L;
        const int ndf = (ctx->mi_nxy == 3 ? 4 : 1) * strata + 1;
        std::vector<double> q((size_t)ndf, 1e300);
        for (int df = 1; df < ndf; ++df) {
            double lo = 0.0, hi = 16.0 + 4.0 * df;
            while (host_igamc(0.5 * df, 0.5 * hi) >= ctx->P.alpha) hi *= 2.0;
            for (int it = 0; it < 64; ++it) {
                const double mid = 0.5 * (lo + hi);
                if (host_igamc(0.5 * df, 0.5 * mid) < ctx->P.alpha)
                    hi = mid;
                else
                    lo = mid;
            }
            q[df] = 0.5 * (lo + hi);
        }
        if (ctx->d_gthr) (void)hipFree(ctx->d_gthr);
        ctx->d_gthr = nullptr;
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_gthr, sizeof(double) * (size_t)ndf));
        FW_HIP(ctx, hipMemcpy(ctx->d_gthr, q.data(), sizeof(double) * (size_t)ndf, hipMemcpyHostToDevice));
        ctx->gthr_n = ndf;
    }
    // d_firstnz doubles as storage for the per-column totals [cnt_nz | cnt_hi]
    FW_HIP(ctx, hipMalloc((void **)&ctx->d_firstnz, sizeof(int32_t) * 2 * (size_t)p));
    FW_HIP(ctx, hipMemcpy(ctx->d_firstnz, cnt_nz.data(), sizeof(int32_t) * p, hipMemcpyHostToDevice));
    FW_HIP(ctx, hipMemcpy(ctx->d_firstnz + p, cnt_hi.data(), sizeof(int32_t) * p, hipMemcpyHostToDevice));
    return FW_OK;
}

// chi-square quantile by bisection on the host (only used as a conservative skip threshold, see mi_pair_epilogue)
static double host_igamc(double a, double x)
{
    if (x <= 0.0 || a <= 0.0) return 1.0;
    double ax = a * std::log(x) - x - std::lgamma(a);
    if (x < 1.0 || x < a) {
        ax = std::exp(ax);
        double r = a, c = 1.0, ans = 1.0;
        do {
            r += 1.0;
            c *= x / r;
            ans += c;
        } while (c / ans > 1e-16);
        return 1.0 - ans * ax / a;
    }
    ax = std::exp(ax);
    double y = 1.0 - a, z = x + y + 1.0, c = 0.0, pkm2 = 1.0, qkm2 = x, pkm1 = x + 1.0, qkm1 = z * x, ans = pkm1 / qkm1, t;
    do {
        c += 1.0;
        y += 1.0;
        z += 2.0;
        double yc = y * c, pk = pkm1 * z - pkm2 * yc, qk = qkm1 * z - qkm2 * yc;
        if (qk != 0.0) {
            double r = pk / qk;
            t = std::fabs((ans - r) / r);
            ans = r;
        } else
            t = 1.0;
        pkm2 = pkm1;
        pkm1 = pk;
        qkm2 = qkm1;
        qkm1 = qk;
        if (std::fabs(pk) > 4503599627370496.0) {
            pkm2 *= 2.22044604925031308085e-16;
            pkm1 *= 2.22044604925031308085e-16;
            qkm2 *= 2.22044604925031308085e-16;
            qkm1 *= 2.22044604925031308085e-16;
        }
    } while (t > 1e-16);
    return ans * ax;
}

int fwi_mi_level0(fw_ctx *ctx, std::vector<int32_t> &pi, std::vector<int32_t> &pj, std::vector<double> &stat,
                  std::vector<double> &pval, int64_t *m_reliable, FwL0Dev *dev)
{
    if (dev) *dev = FwL0Dev{};
    if (ctx->mi_generic) return mig_level0(ctx, pi, pj, stat, pval, m_reliable, dev);
    const int p = ctx->P.p;
    const long long npairs = (long long)p * (p - 1) / 2;
    // G thresholds per df (df <= 4 at level 0): 0.999 * the alpha quantile -> everything below has p > alpha
    double gthr[8];
    for (int df = 0; df < 8; ++df) {
        if (df == 0) {
            gthr[df] = 1e300;
            continue;
        }
        double lo = 0.0, hi = 1e4;
        for (int it = 0; it < 200; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (host_igamc(0.5 * df, 0.5 * mid) < ctx->P.alpha)
                hi = mid;
            else
                lo = mid;
        }
        gthr[df] = 0.999 * lo;
    }
    int rc;
    // matrix-core form of kernel 1: three-valued data whose counts fit 16 bits (FW_L0_MFMA=0: the popcount form, for A/B runs)
    // Default: the nz-adjusted kind (its pairs are decided by the branch-free first pass of the epilogue) from 1 024 variables on.
    // FW_L0_MFMA=2 forces it wherever it is defined (tests: small tables, the plain three-valued kind through the overflow path).
    const int l0_mfma_knob = fw_knob("FW_L0_MFMA") ? atoi(fw_knob("FW_L0_MFMA")) : 1;
    const bool l0_mfma = l0_mfma_knob != 0 && ctx->d_hibits && ctx->P.n <= 65535 &&
                         (l0_mfma_knob == 2 || (ctx->P.kind == FW_MI_NZ && p >= 1024));
    const int l0_tile = l0_mfma ? L0M_T : L0_T;
    const int T = (p + l0_tile - 1) / l0_tile;
    const int TS = (T + L0M_S - 1) / L0M_S;
    const int nblk_all = (int)((long long)T * (T + 1) / 2);
    // target-sharded runs: every rank screens a contiguous range of the linearised upper-triangular tile list (tiles cost
    // the same: the list is balanced) and the significant pairs are all-gathered afterwards (fw_level0_sharded)
    const int b_off = (int)((long long)nblk_all * ctx->l0_rank / ctx->l0_world);
    const int nblk = (int)((long long)nblk_all * (ctx->l0_rank + 1) / ctx->l0_world) - b_off;
    // matrix-core form: the list is walked super-tile by super-tile (slot = super-tile x 64 + tile inside it; slots below the diagonal
    // or beyond the table are empty).  This rank's slots: the contiguous range that holds the real tiles b_off .. b_off + nblk - 1.
    int slot_off = 0, slot_end = 0;
    if (l0_mfma) {
        long long seen = 0;
        bool open = false;
        slot_off = slot_end = (int)((long long)TS * (TS + 1) / 2 * L0M_S * L0M_S);
        for (int si = 0, st = 0; si < TS && !(open && seen >= (long long)b_off + nblk); ++si)
            for (int sj = si; sj < TS && !(open && seen >= (long long)b_off + nblk); ++sj, ++st)
                for (int tin = 0; tin < L0M_S * L0M_S; ++tin) {
                    const int bi_ = si * L0M_S + tin / L0M_S, bj_ = sj * L0M_S + tin % L0M_S;
                    const bool real = bi_ < T && bj_ < T && bi_ <= bj_;
                    if (real && !open && seen == b_off && nblk > 0) {
                        slot_off = st * L0M_S * L0M_S + tin;
                        open = true;
                    }
                    if (real) ++seen;
                    if (open && seen >= (long long)b_off + nblk) {
                        slot_end = st * L0M_S * L0M_S + tin + 1;
                        break;
                    }
                }
        if (!open) slot_off = slot_end = 0;
    }
    const MiDev P = mi_dev(ctx);
    // ---- kernel 1: popcounts + exact reliability/df + Float32 screen -> candidate records ----
    static const int l0_dbg = fw_knob("FW_L0_DBG") ? atoi(fw_knob("FW_L0_DBG")) : 0;  // profiling only (invalid results)
    unsigned long long cap_c = (unsigned long long)std::min<long long>(npairs, 8ll << 20);
    if (cap_c < ctx->l0_cap_hint) cap_c = ctx->l0_cap_hint;  // a repeated call does not overflow (and re-run the kernel) again
    if (cap_c == 0) cap_c = 1;
    MiL0Counters h1{};
    double *d_gthr = nullptr;
    for (int attempt = 0;; ++attempt) {
        if ((rc = fw_dev_reserve(ctx, ctx->d_tmp0, 2 * sizeof(MiL0Counters) + 8 * sizeof(double) + 8 * sizeof(unsigned long long)))) return rc;
        if ((rc = fw_dev_reserve(ctx, ctx->d_jobs, cap_c * sizeof(MiCand)))) return rc;
        FW_HIP(ctx, hipMemsetAsync(ctx->d_tmp0.ptr, 0, 2 * sizeof(MiL0Counters), ctx->stream));
        d_gthr = (double *)((char *)ctx->d_tmp0.ptr + 2 * sizeof(MiL0Counters));
        unsigned long long *d_prof = (unsigned long long *)(d_gthr + 8);
        const bool l0_prof = l0_mfma && fw_knob("FW_L0_VERBOSE");
        if (l0_prof) FW_HIP(ctx, hipMemsetAsync(d_prof, 0, 8 * sizeof(unsigned long long), ctx->stream));
        FW_HIP(ctx, hipMemcpyAsync(d_gthr, gthr, sizeof(gthr), hipMemcpyHostToDevice, ctx->stream));
        if (nblk == 0)
            ;  // more ranks than tiles: nothing to screen here
        else if (l0_mfma) {
            FW_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
            hipLaunchKernelGGL(mi_level0_mfma_kernel,
                               dim3((unsigned)(8 * (((slot_end - 1) / (L0M_S * L0M_S) - slot_off / (L0M_S * L0M_S) + 1 + 7) / 8) * L0M_S * L0M_S)), dim3(512), 0,
                               ctx->stream, P, p, T, ctx->d_firstnz, ctx->d_firstnz + p, d_gthr, (MiL0Counters *)ctx->d_tmp0.ptr, cap_c,
                               (MiCand *)ctx->d_jobs.ptr, l0_dbg, slot_off, slot_end, l0_prof ? d_prof : nullptr);
            FW_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        } else if (ctx->d_hibits)
            hipLaunchKernelGGL(mi_level0_kernel<true>, dim3(nblk), dim3(256), 0, ctx->stream, P, p, T, ctx->d_firstnz,
                               ctx->d_firstnz + p, d_gthr, (MiL0Counters *)ctx->d_tmp0.ptr, cap_c, (MiCand *)ctx->d_jobs.ptr, ctx->d_xlnx, ctx->d_xlnx + (ctx->P.n + 1), l0_dbg, b_off);
        else
            hipLaunchKernelGGL(mi_level0_kernel<false>, dim3(nblk), dim3(256), 0, ctx->stream, P, p, T, ctx->d_firstnz,
                               ctx->d_firstnz + p, d_gthr, (MiL0Counters *)ctx->d_tmp0.ptr, cap_c, (MiCand *)ctx->d_jobs.ptr, ctx->d_xlnx, ctx->d_xlnx + (ctx->P.n + 1), l0_dbg, b_off);
        FW_HIP(ctx, hipGetLastError());
        FW_HIP(ctx, hipMemcpyAsync(&h1, ctx->d_tmp0.ptr, sizeof(h1), hipMemcpyDeviceToHost, ctx->stream));
        FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->cnt.kernel_launches += 1;
        if (l0_mfma && nblk > 0) {  // the Gram product's rate: this launch's tiles x (2 x 128 plane rows)^2 x 64 W multiply-adds
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ctx->cnt.t_l0_mfma_s += 1e-3 * (double)ms;
            ctx->cnt.l0_mfma_flops += 2.0 * (double)nblk * 256.0 * 256.0 * 64.0 * (double)ctx->W;
        }
        if (l0_prof) {
            unsigned long long hp[8];
            FW_HIP(ctx, hipMemcpy(hp, d_prof, sizeof(hp), hipMemcpyDeviceToHost));
            const double nt = (double)std::max<unsigned long long>(hp[4], 1);
            fprintf(stderr, "[fw] mi_level0_mfma_kernel, shader cycles per tile (%llu tiles): staging + matrix loop %.0f, first pass %.0f, second pass %.0f, queue %.0f; pairs for the second pass per tile %.0f\n",
                    hp[4], hp[0] / nt, hp[1] / nt, hp[2] / nt, hp[3] / nt, hp[5] / nt);
        }
        if (h1.n_sig > ctx->l0_cap_hint) ctx->l0_cap_hint = h1.n_sig;
        if (h1.n_sig <= cap_c) break;
        if (attempt == 1) return fw_fail(ctx, FW_ERR_DEVICE, "discrete level-0: candidate buffer overflow twice");
        cap_c = h1.n_sig;
    }
    const unsigned long long ncand = h1.n_sig;
    *m_reliable = npairs - (long long)h1.n_unreliable;
    if (fw_knob("FW_L0_VERBOSE")) fprintf(stderr, "[fw] discrete level-0: pairs %lld reliable %lld candidates %llu\n", npairs, (long long)*m_reliable, ncand);
    pi.clear();
    pj.clear();
    stat.clear();
    pval.clear();
    if (ncand == 0) return FW_OK;
    // ---- kernel 2: exact Float64 statistic + p-value of the candidates ----
    MiL0Counters *d_cnt2 = (MiL0Counters *)ctx->d_tmp0.ptr + 1;
    if ((rc = fw_dev_reserve(ctx, ctx->d_tmp1, ncand * 2 * sizeof(int32_t)))) return rc;
    if ((rc = fw_dev_reserve(ctx, ctx->d_tmp2, ncand * 2 * sizeof(double)))) return rc;
    int32_t *oi = (int32_t *)ctx->d_tmp1.ptr, *oj = oi + ncand;
    double *os = (double *)ctx->d_tmp2.ptr, *op = os + ncand;
    hipLaunchKernelGGL(mi_level0_exact_kernel, dim3((unsigned)((ncand + L0X_IT * 256 - 1) / (L0X_IT * 256))), dim3(256), 0, ctx->stream, P,
                       (const MiCand *)ctx->d_jobs.ptr, ncand, ctx->d_firstnz, ctx->d_firstnz + p, ctx->P.alpha, d_gthr, d_cnt2,
                       ncand, oi, oj, os, op);
    FW_HIP(ctx, hipGetLastError());
    MiL0Counters h2{};
    FW_HIP(ctx, hipMemcpyAsync(&h2, d_cnt2, sizeof(h2), hipMemcpyDeviceToHost, ctx->stream));
    FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cnt.kernel_launches += 1;
    *m_reliable -= (long long)h2.n_unreliable;  // pairs the screening kernel handed over unscreened (mi_pair_epilogue)
    const size_t k = (size_t)h2.n_sig;
    if (dev) {  // results stay on the device for fwi_bh_csr_device
        dev->i = oi;
        dev->j = oj;
        dev->stat64 = os;
        dev->pval = op;
        dev->k = k;
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
    return FW_OK;
}

// Device-driven rounds (fw_devhiton.hip)
int fwi_mi_segments_dev(fw_ctx *ctx, unsigned grid, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out, const unsigned *d_ns,
                        hipStream_t stream)
{
#define MI_SEG_LAUNCH(LL, NN, PP, WW)                                                                                                  \
    hipLaunchKernelGGL((mi_subsets_seg_kernel<LL, NN, PP, WW>), dim3(grid), dim3(256), 0, stream, mi_dev_subsets(ctx), d_segs, d_acc, d_out, \
                       ctx->P.max_k, ctx->P.alpha, (long long)ctx->P.max_tests, d_ns, 0 /* HITON-PC never reads a rejected test's p */)
    const bool pre = ctx->P.n <= MI_PRE_N && ctx->P.max_k <= MI_PRE_K;
    const bool wide = ctx->P.n > 65535;  // 32-bit cell counts and tables (fw_mi_core.h)
    if (ctx->mi_generic) {
        MI_SEG_LAUNCH(0, 2, false, false);
    } else if (ctx->L == 2) {
        if (wide) MI_SEG_LAUNCH(2, 2, false, true); else if (pre) MI_SEG_LAUNCH(2, 2, true, false); else MI_SEG_LAUNCH(2, 2, false, false);
    } else if (ctx->mi_nxy == 2) {
        if (wide) MI_SEG_LAUNCH(3, 2, false, true); else if (pre) MI_SEG_LAUNCH(3, 2, true, false); else MI_SEG_LAUNCH(3, 2, false, false);
    } else {
        if (wide) MI_SEG_LAUNCH(3, 3, false, true); else if (pre) MI_SEG_LAUNCH(3, 3, true, false); else MI_SEG_LAUNCH(3, 3, false, false);
    }
#undef MI_SEG_LAUNCH
    FW_HIP(ctx, hipGetLastError());
    return FW_OK;
}

// conditioning sets of 6 and 7 variables (r05): L^k strata x (sub-table cells + total, whole words) 16-bit entries must fit MI_TAB16_BIG,
// and the large table has no 32-bit-count form
int fwi_mi_big_limits(fw_ctx *ctx, int k)
{
    if (k <= FW_MAX_K_FAST || ctx->mi_generic) return FW_OK;
    if (ctx->P.n > 65535) return fw_fail(ctx, FW_ERR_LIMIT, "discrete tests with %d conditioning variables need n <= 65535 (got %d)", k, ctx->P.n);
    long long strata = 1;
    for (int j = 0; j < k; ++j) strata *= ctx->L;
    const int nct16 = (ctx->mi_nxy * ctx->mi_nxy + 1 + 1) & ~1;
    if (strata * nct16 > MI_TAB16_BIG)
        return fw_fail(ctx, FW_ERR_LIMIT, "discrete tests with %d conditioning variables on %d-level data with a %d x %d sub-table need %lld table entries (limit %d)", k,
                       ctx->L, ctx->mi_nxy, ctx->mi_nxy, strata * nct16, MI_TAB16_BIG);
    return FW_OK;
}

int fwi_mi_test_batch(fw_ctx *ctx, int64_t m, const int32_t *X, const int32_t *Y, const int64_t *zoff,
                      const int32_t *zflat, fw_test_result *out)
{
    if (m == 0) return FW_OK;
    for (int64_t t = 0; t < m; ++t)
        if (zoff[t + 1] - zoff[t] > MI_MAX_K)
            return fw_fail(ctx, FW_ERR_LIMIT, "discrete tests support at most %d conditioning variables", MI_MAX_K);
    const int64_t nz = zoff[m];
    int rc;
    if ((rc = fw_dev_reserve(ctx, ctx->d_jobs, (size_t)m * 2 * sizeof(int32_t) + (size_t)(m + 1) * sizeof(int64_t)))) return rc;
    if ((rc = fw_dev_reserve(ctx, ctx->d_acc, (size_t)(nz > 0 ? nz : 1) * sizeof(int32_t)))) return rc;
    if ((rc = fw_dev_reserve(ctx, ctx->d_out, (size_t)m * sizeof(fw_test_result)))) return rc;
    long long *dz = (long long *)ctx->d_jobs.ptr;
    int32_t *dX = (int32_t *)(dz + m + 1), *dY = dX + m;
    FW_HIP(ctx, hipMemcpyAsync(dz, zoff, (size_t)(m + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    FW_HIP(ctx, hipMemcpyAsync(dX, X, (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    FW_HIP(ctx, hipMemcpyAsync(dY, Y, (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    if (nz > 0)
        FW_HIP(ctx, hipMemcpyAsync(ctx->d_acc.ptr, zflat, (size_t)nz * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    int kmax = 0;
    for (int64_t t = 0; t < m; ++t) kmax = std::max<int>(kmax, (int)(zoff[t + 1] - zoff[t]));
    MiDev Pd = mi_dev(ctx);
    static const bool prof = fw_knob("FW_MI_PROF") != nullptr;
    if (prof) {
        if ((rc = fw_dev_reserve(ctx, ctx->d_tmp0, (size_t)m * 8 * sizeof(unsigned long long)))) return rc;
        FW_HIP(ctx, hipMemsetAsync(ctx->d_tmp0.ptr, 0, (size_t)m * 8 * sizeof(unsigned long long), ctx->stream));
        Pd.prof = (unsigned long long *)ctx->d_tmp0.ptr;
    }
#define MI_TB_LAUNCH(LL, NN, PP, WW)                                                                                                         \
    hipLaunchKernelGGL((mi_test_batch_kernel<LL, NN, PP, WW>), dim3((unsigned)((m + 3) / 4)), dim3(256), 0, ctx->stream, Pd, \
                       (long long)m, dX, dY, dz, (const int32_t *)ctx->d_acc.ptr, (fw_test_result *)ctx->d_out.ptr)
    const bool pre = ctx->P.n <= MI_PRE_N && kmax <= MI_PRE_K;  // the batch's own largest conditioning set decides here
    const bool wide = ctx->P.n > 65535;
    if (kmax > FW_MAX_K_FAST && !ctx->mi_generic) {  // conditioning sets of 6 and 7 variables: the large table
        if (int rcb = fwi_mi_big_limits(ctx, kmax)) return rcb;
#define MI_TB_LAUNCH_BIG(LL, NN)                                                                                                             \
    hipLaunchKernelGGL((mi_test_batch_kernel<LL, NN, false, false, true>), dim3((unsigned)((m + 3) / 4)), dim3(256), 0, ctx->stream, Pd, \
                       (long long)m, dX, dY, dz, (const int32_t *)ctx->d_acc.ptr, (fw_test_result *)ctx->d_out.ptr)
        if (ctx->L == 2) MI_TB_LAUNCH_BIG(2, 2); else if (ctx->mi_nxy == 2) MI_TB_LAUNCH_BIG(3, 2); else MI_TB_LAUNCH_BIG(3, 3);
#undef MI_TB_LAUNCH_BIG
    } else if (ctx->mi_generic) {
        // (tables in device memory: launches of as many workgroups as fit the table budget, one after the other on the stream)
        const int64_t wgs_all = (m + 3) / 4, wgs_max = mig_launch_wgs(ctx, wgs_all);
        if ((rc = mig_dev_tables(ctx, Pd, wgs_max, 0))) return rc;
        for (int64_t w0 = 0; w0 < wgs_all; w0 += wgs_max) {
            const int64_t t0 = 4 * w0, mc = std::min<int64_t>(m - t0, 4 * wgs_max);
            hipLaunchKernelGGL((mi_test_batch_kernel<0, 2, false, false>), dim3((unsigned)((mc + 3) / 4)), dim3(256), 0, ctx->stream, Pd, (long long)mc, dX + t0,
                               dY + t0, dz + t0, (const int32_t *)ctx->d_acc.ptr, (fw_test_result *)ctx->d_out.ptr + t0);
        }
    } else if (ctx->L == 2) {
        if (wide) MI_TB_LAUNCH(2, 2, false, true); else if (pre) MI_TB_LAUNCH(2, 2, true, false); else MI_TB_LAUNCH(2, 2, false, false);
    } else if (ctx->mi_nxy == 2) {
        if (wide) MI_TB_LAUNCH(3, 2, false, true); else if (pre) MI_TB_LAUNCH(3, 2, true, false); else MI_TB_LAUNCH(3, 2, false, false);
    } else {
        if (wide) MI_TB_LAUNCH(3, 3, false, true); else if (pre) MI_TB_LAUNCH(3, 3, true, false); else MI_TB_LAUNCH(3, 3, false, false);
    }
#undef MI_TB_LAUNCH
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipMemcpyAsync(out, ctx->d_out.ptr, (size_t)m * sizeof(fw_test_result), hipMemcpyDeviceToHost, ctx->stream));
    FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (prof) {
        std::vector<unsigned long long> hv((size_t)m * 8);
        FW_HIP(ctx, hipMemcpy(hv.data(), ctx->d_tmp0.ptr, hv.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int64_t t = 0; t < m; ++t)
            for (int q = 0; q < 8; ++q) h[q] += hv[(size_t)t * 8 + q];
        const double nt = (double)std::max<unsigned long long>(h[4], 1), nm = (double)std::max<unsigned long long>(h[5], 1);
        fprintf(stderr, "[fw] mi test phases, shader cycles per test: counting %.0f, occupancy/power %.0f, MI terms %.0f (of %.0f%% with power), p-value %.0f\n",
                h[0] / nt, h[1] / nt, h[2] / nm, 100.0 * nm / nt, h[3] / nt);
    }
    ctx->cnt.kernel_launches += 1;
    return FW_OK;
}

int fwi_mi_segments(fw_ctx *ctx, int64_t nseg, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out, FwPoolBuf &pb)
{
    if (nseg == 0) return FW_OK;
    FW_HIP(ctx, hipEventRecord(pb.ev0, pb.launch_stream));
#define MI_SEG_LAUNCH(LL, NN, PP, WW)                                                                                                       \
    hipLaunchKernelGGL((mi_subsets_seg_kernel<LL, NN, PP, WW>), dim3((unsigned)nseg), dim3(256), 0, pb.launch_stream, mi_dev_subsets(ctx), d_segs, \
                       d_acc, d_out, ctx->P.max_k, ctx->P.alpha, (long long)ctx->P.max_tests, (const unsigned *)nullptr, 1)
    const bool pre = ctx->P.n <= MI_PRE_N && ctx->P.max_k <= MI_PRE_K;
    const bool wide = ctx->P.n > 65535;  // 32-bit cell counts and tables (fw_mi_core.h)
    if (ctx->P.max_k > FW_MAX_K_FAST && !ctx->mi_generic) {  // conditioning sets of 6 and 7 variables: the large table (fwi_mi_big_limits holds)
#define MI_SEG_LAUNCH_BIG(LL, NN)                                                                                                           \
    hipLaunchKernelGGL((mi_subsets_seg_kernel<LL, NN, false, false, true>), dim3((unsigned)nseg), dim3(256), 0, pb.launch_stream, mi_dev_subsets(ctx), \
                       d_segs, d_acc, d_out, ctx->P.max_k, ctx->P.alpha, (long long)ctx->P.max_tests, (const unsigned *)nullptr, 1)
        if (ctx->L == 2) MI_SEG_LAUNCH_BIG(2, 2); else if (ctx->mi_nxy == 2) MI_SEG_LAUNCH_BIG(3, 2); else MI_SEG_LAUNCH_BIG(3, 3);
#undef MI_SEG_LAUNCH_BIG
    } else if (ctx->mi_generic) {
        MiDev Pg = mi_dev_subsets(ctx);
        const int64_t wgs_max = mig_launch_wgs(ctx, nseg);
        if (int rcg = mig_dev_tables(ctx, Pg, wgs_max, &pb == &ctx->pb[1] ? 2 : 1)) return rcg;
        for (int64_t s0 = 0; s0 < nseg; s0 += wgs_max)
            hipLaunchKernelGGL((mi_subsets_seg_kernel<0, 2, false, false>), dim3((unsigned)std::min<int64_t>(wgs_max, nseg - s0)), dim3(256), 0, pb.launch_stream, Pg,
                               d_segs + s0, d_acc, d_out + s0, ctx->P.max_k, ctx->P.alpha, (long long)ctx->P.max_tests, (const unsigned *)nullptr, 1);
    } else if (ctx->L == 2) {
        if (wide) MI_SEG_LAUNCH(2, 2, false, true); else if (pre) MI_SEG_LAUNCH(2, 2, true, false); else MI_SEG_LAUNCH(2, 2, false, false);
    } else if (ctx->mi_nxy == 2) {
        if (wide) MI_SEG_LAUNCH(3, 2, false, true); else if (pre) MI_SEG_LAUNCH(3, 2, true, false); else MI_SEG_LAUNCH(3, 2, false, false);
    } else {
        if (wide) MI_SEG_LAUNCH(3, 3, false, true); else if (pre) MI_SEG_LAUNCH(3, 3, true, false); else MI_SEG_LAUNCH(3, 3, false, false);
    }
#undef MI_SEG_LAUNCH
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipEventRecord(pb.ev1, pb.launch_stream));
    return FW_OK;
}
