// Wave-level conditional contingency-table test of the discrete kinds (FW_MI / FW_MI_NZ), popcount formulation.
// Included by fw_mi.hip (single tests, test_subsets segments) and fw_devhiton.hip (per-target HITON-PC kernel).
//
// Reference semantics (file:line into /root/reference/src), SPARSE-path rules unless MiDev::dense:
//   conditional test                     tests.jl:184-229      univariate form tests.jl:28-77
//   2-way / 3-way tables                 contingency.jl:80-123 (2-way), :182-258 (k = 1 HE special case), :300-480 (generic)
//   mutual information / df / p          statfuns.jl:157-305
//
// One wavefront = one test (X, Y | Z_1..Z_k).  Every variable is two bit planes over the samples ([p][W] 64-bit words:
// nz = value != 0, hi = value == 2), read here as 32-bit words: lane l owns words l, l + 64, ... of every plane.
// A table cell is a popcount: count(x, y, z_1..z_k) = |rows & X_x & Y_y & Z1_z1 & ... & Zk_zk| (v_and + v_bcnt per 32 rows
// and lane), summed over the lanes with a DPP reduction.  Only the cells of the nz-adjusted sub-table are counted (rows
// and columns the reference's nz_adjust_cont_tab drops are never looked at); strata are processed in batches of L^2
// (the two fastest digits of the key), so any k fits the registers.  The reduced counts land in a small LDS table
// [stratum][cell]; marginals, G^2 terms and df are then computed with lanes <-> (stratum, cell) pairs.
// (r01 binned one row per lane and step through LDS atomics: ~6 000 wave instructions per test at n = 5 000 whatever k;
// this form needs ~0.15k (k = 1) to ~1.4k (k = 3, 27 strata) and never touches an atomic.)
#pragma once
#include "fw_internal.h"

#ifndef MI_MAX_K
#define MI_MAX_K FW_MAX_K  // (fw_devhiton.hip: FW_MAX_K_FAST -- the persistent kernel serves max_k <= 5)
#endif
#define MI_PRE_N 6144  // samples up to which a lane holds every word of a column in registers (3 x 64 x 32 rows)
#define MI_PRE_K 3     // ... and the largest conditioning set that form serves
#define MI_TAB16 2432  // u16 entries of one wave's LDS table: 3^5 strata x 10 (9 cells + stratum total)

struct MiDev {
    const unsigned long long *nz;
    const unsigned long long *hi;  // may be null when L == 2
    const int32_t *levels;
    const int32_t *maxv;
    int W, n, L, nzmode, hps;
    int dense;  // dense-matrix table rules (contingency.jl:7-56): every row counted, levels_z = distinct Z keys over all rows
    int view;   // dense rules inside HITON-PC: the rows of a (T, candidate) test are those where T / the candidate are non-zero if
                // they have more than two levels (prepare_nzdata, hiton.jl:41-50,85 -> needs_nz_view, misc.jl:103-107)
    long long n_obs_min;
    const double *gthr;  // [df] -> G^2 with ccdf(Chisq(df), G^2) = alpha (host bisection), df = 0 .. gthr_n - 1; may be null
    int gthr_n;
    double alpha;
    unsigned long long *prof;  // profiling only (FW_MI_PROF=1, fw_test_batch): shader-clock cycles per phase, summed over tests
    const unsigned char *vals;  // generic form (data with a value above 2, r04): one byte per (variable, sample), [p][n]; else null
    int rowk;  // smallest conditioning-set size for which the ROW form of the counting phase is considered (mi_bin_rows; 99: never)
    unsigned *gtab;               // generic form with tables beyond LDS (r05): one table of gtab_words words per wavefront of the launch in
    unsigned long long gtab_words;  // device memory (wavefront w of workgroup b: table 4 b + w); null / 0: the table lives in LDS
};

struct MiRes {
    double stat, pval;  // pval: NaN until mi_res_pval() has been asked for it (Q(a, x) costs as much as a small test)
    int df, power;
    double g;           // G^2 = 2 |MI| n_obs (statfuns.jl:157-161)
    long long n_obs;
};

struct MiZs {
    int v[MI_MAX_K];
};

// ------------------------------------------------------------------------------------------------
// special functions: regularised upper incomplete gamma Q(a, x) (series / continued fraction, Cephes structure)
// stands in for ccdf(Chisq(df), g) = Q(df/2, g/2)  (statfuns.jl:157-161)
// ------------------------------------------------------------------------------------------------
static __device__ __noinline__ double mi_igamc(double a, double x)
{
    if (isnan(a) || isnan(x)) return NAN;
    if (x <= 0.0 || a <= 0.0) return 1.0;
    if (isinf(x)) return 0.0;
    // one degree of freedom (every 2 x 2 table: the pair tests of the HE kinds, r05): Q(1/2, x) = erfc(sqrt(x)) -- one library call
    // instead of a series / continued fraction of up to a few hundred terms (the level-0 exact kernel spent most of its 6.7 ms at cfg4
    // here).  Same function, other rounding: relative difference to the expansion <= 2 x 1.1e-16 (the rounding of the root, amplified
    // by d log erfc / d log s = 2 s^2), far inside the 1e-10 of DESIGN.md section 2.  Only in the normal range of p: the subnormal
    // tail keeps the expansion (and its "ax < -745.2 -> 0" cut), so that p-values down there stay what they were.
    if (a == 0.5 && x < 690.0) return erfc(sqrt(x));
    double ax = a * log(x) - x - lgamma(a);
    if (x < 1.0 || x < a) {
        if (ax < -745.2) return 1.0;
        ax = exp(ax);
        double r = a, c = 1.0, ans = 1.0;
        do {
            r += 1.0;
            c *= x / r;
            ans += c;
        } while (c / ans > 1.1102230246251565e-16);
        return 1.0 - ans * ax / a;
    }
    if (ax < -745.2) return 0.0;
    ax = exp(ax);
    const double big = 4503599627370496.0, biginv = 2.22044604925031308085e-16;
    double y = 1.0 - a, z = x + y + 1.0, c = 0.0;
    double pkm2 = 1.0, qkm2 = x, pkm1 = x + 1.0, qkm1 = z * x;
    double ans = pkm1 / qkm1, t;
    do {
        c += 1.0;
        y += 1.0;
        z += 2.0;
        const double yc = y * c;
        const double pk = pkm1 * z - pkm2 * yc;
        const double qk = qkm1 * z - qkm2 * yc;
        if (qk != 0.0) {
            const double r = pk / qk;
            t = fabs((ans - r) / r);
            ans = r;
        } else {
            t = 1.0;
        }
        pkm2 = pkm1;
        pkm1 = pk;
        qkm2 = qkm1;
        qkm1 = qk;
        if (fabs(pk) > big) {
            pkm2 *= biginv;
            pkm1 *= biginv;
            qkm2 *= biginv;
            qkm1 *= biginv;
        }
    } while (t > 1.1102230246251565e-16);
    return ans * ax;
}

static __device__ __forceinline__ double mi_pval_dev(double mi_abs, int df, long long n_obs)
{
    const double g = 2.0 * mi_abs * (double)n_obs;
    return df > 0 ? mi_igamc(0.5 * (double)df, 0.5 * g) : 1.0;
}

// Wave reductions on the DPP network (row shifts / broadcasts between VGPR lanes: no LDS round trip, ~8 cycles a step; the
// ds_bpermute shuffles they replace cost ~100 cycles each, and a test ends with ten of these reductions in a row).
// sum over the 64 lanes, valid in lane 63: four DPP adds inside each row of 16 lanes, two row broadcasts (gfx9 DPP)
static __device__ __forceinline__ unsigned mi_dpp_sum63(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false);   // quad_perm:[1,0,3,2]
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false);   // quad_perm:[2,3,0,1]
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}
// every lane gets the sum (values below 2^32: counts of rows, df)
static __device__ __forceinline__ int wave_sum_i(int v) { return __builtin_amdgcn_readlane((int)mi_dpp_sum63((unsigned)v), 63); }
static __device__ __forceinline__ long long wave_sum_ll(long long v) { return (long long)(unsigned)wave_sum_i((int)v); }
static __device__ __forceinline__ int wave_max_i(int v)
{
#define MI_DPP_MAX(ctrl, rmask)                                                                   \
    {                                                                                             \
        const int t = __builtin_amdgcn_update_dpp(v, v, ctrl, rmask, 0xf, false); /* invalid source: own value */ \
        v = t > v ? t : v;                                                                        \
    }
    MI_DPP_MAX(0xb1, 0xf)
    MI_DPP_MAX(0x4e, 0xf)
    MI_DPP_MAX(0x114, 0xf)
    MI_DPP_MAX(0x118, 0xf)
    MI_DPP_MAX(0x142, 0xa)
    MI_DPP_MAX(0x143, 0xc)
#undef MI_DPP_MAX
    return __builtin_amdgcn_readlane(v, 63);
}
static __device__ __forceinline__ unsigned mi_wave_max_u(unsigned v) { return (unsigned)wave_max_i((int)v); }
static __device__ __forceinline__ double wave_sum_d(double v)
{
#define MI_DPP_ADDD(ctrl, rmask)                                                                               \
    {                                                                                                          \
        const long long b = __double_as_longlong(v);                                                           \
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, ctrl, rmask, 0xf, false);           \
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), ctrl, rmask, 0xf, false);   \
        v += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); /* invalid source: + 0.0 */ \
    }
    MI_DPP_ADDD(0xb1, 0xf)
    MI_DPP_ADDD(0x4e, 0xf)
    MI_DPP_ADDD(0x114, 0xf)
    MI_DPP_ADDD(0x118, 0xf)
    MI_DPP_ADDD(0x142, 0xa)
    MI_DPP_ADDD(0x143, 0xc)
#undef MI_DPP_ADDD
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// level mask of one 32-row word: value == d  (planes: zn = value != 0, zh = value == 2)
template <int L>
static __device__ __forceinline__ unsigned mi_level_mask(unsigned zn, unsigned zh, int d)
{
    if (L == 2) return d == 0 ? ~zn : zn;
    return d == 0 ? ~zn : (d == 1 ? (zn & ~zh) : zh);
}

// the 32-row words one lane holds of the k + 2 columns of a test (word index dw of every plane)
template <int KM>
struct MiWords {
    unsigned xn, xh, yn, yh, vm;  // vm: rows of this word that exist (< n)
    unsigned zn[KM], zh[KM];
};

template <int L, int KM>
static __device__ __forceinline__ void mi_load_words(MiWords<KM> &w, const unsigned *pn, const unsigned *ph, const unsigned *xn,
                                                     const unsigned *xh, const unsigned *yn, const unsigned *yh, size_t W2,
                                                     const MiZs &zs, int k, int dw, int nd, int n)
{
    const bool ok = dw < nd;
    w.xn = ok ? xn[dw] : 0u;
    w.yn = ok ? yn[dw] : 0u;
    w.xh = (L == 3 && ok && xh) ? xh[dw] : 0u;
    w.yh = (L == 3 && ok && yh) ? yh[dw] : 0u;
#pragma unroll
    for (int j = 0; j < KM; ++j) {
        w.zn[j] = (j < k && ok) ? pn[(size_t)zs.v[j] * W2 + dw] : 0u;
        w.zh[j] = (L == 3 && j < k && ok && ph) ? ph[(size_t)zs.v[j] * W2 + dw] : 0u;
    }
    const int left = n - dw * 32;
    w.vm = !ok ? 0u : (left >= 32 ? 0xffffffffu : ((1u << left) - 1u));
}

// byte form of the loader (n <= 512: lane l holds rows 8 l .. 8 l + 7 of every plane -- 64 lanes share the rows of a small table instead
// of 16 lanes holding a 32-row word each; the fields of MiWords then carry 8 bits)
template <int L, int KM>
static __device__ __forceinline__ void mi_load_bytes(MiWords<KM> &w, const unsigned *pn, const unsigned *ph, const unsigned *xn,
                                                     const unsigned *xh, const unsigned *yn, const unsigned *yh, size_t W2,
                                                     const MiZs &zs, int k, int lane, int n)
{
    const bool ok = 8 * lane < n;
    w.xn = ok ? ((const unsigned char *)xn)[lane] : 0u;
    w.yn = ok ? ((const unsigned char *)yn)[lane] : 0u;
    w.xh = (L == 3 && ok && xh) ? ((const unsigned char *)xh)[lane] : 0u;
    w.yh = (L == 3 && ok && yh) ? ((const unsigned char *)yh)[lane] : 0u;
#pragma unroll
    for (int j = 0; j < KM; ++j) {
        w.zn[j] = (j < k && ok) ? ((const unsigned char *)(pn + (size_t)zs.v[j] * W2))[lane] : 0u;
        w.zh[j] = (L == 3 && j < k && ok && ph) ? ((const unsigned char *)(ph + (size_t)zs.v[j] * W2))[lane] : 0u;
    }
    const int left = n - 8 * lane;
    w.vm = !ok ? 0u : (left >= 8 ? 0xffu : ((1u << left) - 1u));
}

// adds the rows of one word to the cell counters of batch b (strata b * SB .. b * SB + SB - 1)
template <int L, int NXY, int KM>
static __device__ __forceinline__ void mi_count_words(const MiWords<KM> &w, unsigned (&acc)[L * L][NXY * NXY + 1], int b, int k, int SB,
                                                      bool flagX, bool flagY, bool dense, bool tot_sep, bool viewX, bool viewY)
{
    constexpr int NC = NXY * NXY;
    constexpr int SBMAX = L * L;
    unsigned msub = w.vm;  // rows of the (nz-adjusted) sub-table
    if (flagX) msub &= w.xn;
    if (flagY) msub &= w.yn;
    unsigned mtab = msub;  // rows of the table (stratum occupancy)
    if (dense) {
        mtab = w.vm;  // every row of the data handed to the test ...
        if (viewX) mtab &= w.xn;  // ... which inside HITON-PC is a row view (MiDev::view)
        if (viewY) mtab &= w.yn;
    }
    unsigned xs[NXY], ys[NXY];
    if (NXY == 2) {
        const unsigned xu = flagX ? w.xh : w.xn, yu = flagY ? w.yh : w.yn;
        xs[0] = ~xu;
        xs[1] = xu;
        ys[0] = ~yu;
        ys[1] = yu;
    } else {
#pragma unroll
        for (int i = 0; i < NXY; ++i) {
            xs[i] = mi_level_mask<3>(w.xn, w.xh, i);
            ys[i] = mi_level_mask<3>(w.yn, w.yh, i);
        }
    }
    unsigned cm[NC];
#pragma unroll
    for (int j = 0; j < NXY; ++j)
#pragma unroll
        for (int i = 0; i < NXY; ++i) cm[i + NXY * j] = msub & xs[i] & ys[j];
    // digits of Z_3.. are fixed inside a batch (key = sum_j z_j L^j, types.jl:32-39 cum_levels)
    unsigned hm = 0xffffffffu;
    {
        int bb = b;
#pragma unroll
        for (int j = 2; j < KM; ++j)
            if (j < k) {
                const int d = bb % L;
                bb /= L;
                hm &= mi_level_mask<L>(w.zn[j], w.zh[j], d);
            }
    }
    unsigned z0[L], z1[L];
#pragma unroll
    for (int d = 0; d < L; ++d) {
        z0[d] = (k >= 1) ? mi_level_mask<L>(w.zn[0], w.zh[0], d) : (d == 0 ? 0xffffffffu : 0u);
        z1[d] = (k >= 2) ? (mi_level_mask<L>(w.zn[1], w.zh[1], d) & hm) : (d == 0 ? hm : 0u);
    }
#pragma unroll
    for (int s = 0; s < SBMAX; ++s)
        if (s < SB) {
            const unsigned zk = z0[s % L] & z1[s / L];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[s][c] += (unsigned)__builtin_popcount(cm[c] & zk);
            if (tot_sep) acc[s][NC] += (unsigned)__builtin_popcount(mtab & zk);
        }
}

// ROW form of the counting phase (r04).  The popcount form above costs strata x cells x words whatever the data holds: 27 x 4 x 3
// AND + popcount pairs and 54 six-step DPP reductions for a k = 3 test at n = 5000 (~2 100 of its ~2 900 VALU instructions).  A lane
// can instead walk the SET BITS of its own sub-table row mask -- in the nz-adjusted kinds only the rows where X and Y are both
// non-zero (cfg4: a few hundred of 5 000) -- extract the k + 2 values of each row from the words it already holds and add 1 to the
// [stratum][cell] entry of the wavefront's LDS table (ds_add_u32; two 16-bit counts per word unless WIDE).  Cost: ~26 instructions
// per step, steps = the largest popcount of a lane's word.  Same table, same integers: everything after the counting is unchanged.
template <int L, int NXY, int KM, bool WIDE>
static __device__ __forceinline__ void mi_bin_rows(const MiWords<KM> &w, unsigned *tab32, int k, bool flagX, bool flagY)
{
    constexpr int NC = NXY * NXY;
    constexpr int NCT16 = (NC + 2) & ~1;
    unsigned m = w.vm;
    if (flagX) m &= w.xn;
    if (flagY) m &= w.yn;
    const unsigned xu = flagX ? w.xh : w.xn, yu = flagY ? w.yh : w.yn;  // NXY == 2: the bit that tells the sub-table's two values apart
    while (m) {
        const int b = __builtin_ctz(m);
        m &= m - 1u;
        unsigned c;
        if (NXY == 2)
            c = ((xu >> b) & 1u) + 2u * ((yu >> b) & 1u);
        else
            c = ((w.xn >> b) & 1u) + ((w.xh >> b) & 1u) + 3u * (((w.yn >> b) & 1u) + ((w.yh >> b) & 1u));
        unsigned key = 0u, mul = 1u;
#pragma unroll
        for (int j = 0; j < KM; ++j)
            if (j < k) {
                unsigned d = (w.zn[j] >> b) & 1u;
                if (L == 3) d += (w.zh[j] >> b) & 1u;
                key += d * mul;
                mul *= (unsigned)L;
            }
        const unsigned idx = key * (unsigned)NCT16 + c;
        if (WIDE)
            __hip_atomic_fetch_add(tab32 + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        else
            __hip_atomic_fetch_add(tab32 + (idx >> 1), 1u << ((idx & 1u) << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}
template <int KM>
static __device__ __forceinline__ int mi_row_steps(const MiWords<KM> &w, bool flagX, bool flagY)
{
    unsigned m = w.vm;
    if (flagX) m &= w.xn;
    if (flagY) m &= w.yn;
    return (int)mi_wave_max_u((unsigned)__builtin_popcount(m));
}

// ------------------------------------------------------------------------------------------------
// L = number of levels of the context (2 or 3); NXY = 2: X and Y take two values inside the sub-table (presence /
// absence, or the two non-zero bins of an nz-adjusted variable) -> 4 cells per stratum; NXY = 3: 9 cells (mi on 3-valued data).
// tab: this wave's LDS table (MI_TAB16 u16).  Every lane returns the same result.
// ------------------------------------------------------------------------------------------------
// WIDE (n > 65 535; never with PRE): the cell counts no longer fit 16 bits -- one count per register through the reduction and a
// 32-bit table (the caller's table slot is twice as large); everything else is the same code.
template <bool WIDE>
struct MiTabT {
    typedef unsigned short type;
};
template <>
struct MiTabT<true> {
    typedef unsigned type;
};
template <int L, int NXY, bool PRE, bool WIDE = false>
static __device__ __forceinline__ MiRes mi_test_core(const MiDev &P, const int X_in, const int Y_in, const MiZs &zs_in, const int k_in,
                                                     unsigned short *tab_raw)
{
    typedef typename MiTabT<WIDE>::type TabT;
    TabT *tab = (TabT *)tab_raw;
    // the test is the same in every lane: say so (scalar registers, scalar branches -- the compiler cannot prove that values
    // loaded through a wave-indexed pointer are uniform, and predicated every stratum of the unrolled loops instead)
    const int X = __builtin_amdgcn_readfirstlane(X_in), Y = __builtin_amdgcn_readfirstlane(Y_in);
    const int k = __builtin_amdgcn_readfirstlane(k_in);
    MiZs zs;
#pragma unroll
    for (int q = 0; q < MI_MAX_K; ++q) zs.v[q] = __builtin_amdgcn_readfirstlane(zs_in.v[q]);
    constexpr int NC = NXY * NXY;
    constexpr int NCT = NC + 1;            // + the stratum total over the table rows
    constexpr int NCT16 = (NCT + 1) & ~1;  // u16 entries per stratum (whole 32-bit words)
    constexpr int SBMAX = L * L;
    const int lane = threadIdx.x & 63;
    const bool flagX = P.nzmode && P.maxv[X] > 1, flagY = P.nzmode && P.maxv[Y] > 1;
    const bool any_flag = flagX || flagY;
    const bool special_k1 = (k == 1) && any_flag && !P.dense;  // contingency.jl:250-253 (sparse dispatch only)
    int lx, ly;
    if (P.nzmode) {  // tests.jl:200-203: levels of the nz-adjusted sub-table
        lx = L - (flagX ? 1 : 0);
        ly = L - (flagY ? 1 : 0);
    } else {
        lx = P.levels[X];
        ly = P.levels[Y];
    }
    MiRes res;
    res.stat = 0.0;
    res.pval = 1.0;
    res.df = 0;
    res.power = 0;
    res.g = 0.0;
    res.n_obs = 0;
    if (k == 0) {  // tests.jl:36 sufficient_power(X, Y, data, ...) pre-check (tests.jl:9-20)
        bool ok = (long long)P.n >= P.n_obs_min;
        if (ok) {
            const long long vx = P.levels[X], vy = P.levels[Y];
            const long long ox = vx > 1 ? 2 : 1, oy = vy > 1 ? 2 : 1;
            ok = ((double)P.n / (double)((vx - ox) * (vy - oy))) > (double)P.hps;
        }
        if (!ok) return res;
    }
    int S = 1;
    for (int j = 0; j < k; ++j) S *= L;
    const int SB = k >= 2 ? SBMAX : S;  // strata per batch: the digits of Z_1 and Z_2 vary inside a batch
    const int nbatch = S / SB;
    const bool tot_sep = P.dense && any_flag;  // dense rule: the table holds every row, the sub-table only the non-zero ones
    const bool viewX = P.dense && P.view && P.nzmode && P.levels[X] > 2, viewY = P.dense && P.view && P.nzmode && P.levels[Y] > 2;
    const unsigned *pn = (const unsigned *)P.nz, *ph = (const unsigned *)P.hi;
    const size_t W2 = 2 * (size_t)P.W;
    const unsigned *xn = pn + (size_t)X * W2, *yn = pn + (size_t)Y * W2;
    const unsigned *xh = ph ? ph + (size_t)X * W2 : nullptr, *yh = ph ? ph + (size_t)Y * W2 : nullptr;
    const int nd = (P.n + 31) >> 5;
    unsigned *tab32 = (unsigned *)tab_raw;  // (!WIDE: two 16-bit counts per word)
    const unsigned long long pt0 = P.prof ? __builtin_readcyclecounter() : 0ull;
    // ---- counting ----
    // PRE (n <= 6144: at most three 32-row words per lane and plane): every word of the k + 2 columns is loaded once, up
    // front, with all loads in flight together -- one memory round trip per test.  (Loading inside the batch / word loops
    // made a k = 3 test at n = 5000 nine dependent round trips: 16 us per test for a lone wavefront, r02 trace.)
    constexpr int NIT = PRE ? 3 : 1;
    constexpr int KM = PRE ? 3 : MI_MAX_K;  // the register-resident form serves k <= 3 (the host picks PRE only for max_k <= 3)
    MiWords<KM> wd[NIT];
    // small tables (n <= 512, k >= 2): the row form on BYTES -- every lane walks at most eight rows whatever k, where the popcount
    // form keeps 16 lanes busy with 9 .. 27 strata x cells each (cfg2: n = 500)
    const bool byteform = P.n <= 512 && !tot_sep && k >= P.rowk && k >= 2;
    if (PRE && !byteform) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) mi_load_words<L, KM>(wd[it], pn, ph, xn, xh, yn, yh, W2, zs, k, it * 64 + lane, nd, P.n);
    }
    // row form or popcount form?  Both fill the same table; an estimate of their instruction counts decides (wave-uniform).
    bool rowform = false;
    if (byteform) {
        rowform = true;
        const int nw = (S * NCT16 * (int)sizeof(TabT)) >> 2;
        for (int q = lane; q < nw; q += 64) tab32[q] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        mi_load_bytes<L, KM>(wd[0], pn, ph, xn, xh, yn, yh, W2, zs, k, lane, P.n);
        mi_bin_rows<L, NXY, KM, WIDE>(wd[0], tab32, k, flagX, flagY);
    } else if (!tot_sep && any_flag && k >= P.rowk) {  // (sub-tables of the nz-adjusted kinds: few rows; a dense sub-table walks 32 bits per word)
        const int nslot = (nd + 63) >> 6;
        int steps;
        if (PRE) {
            steps = 0;
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if (it * 64 < nd) steps += mi_row_steps<KM>(wd[it], flagX, flagY);
        } else {
            mi_load_words<L, KM>(wd[0], pn, ph, xn, xh, yn, yh, W2, zs, 0, lane, nd, P.n);  // X and Y only: the first 2 048 rows stand for the rest
            steps = mi_row_steps<KM>(wd[0], flagX, flagY) * nslot;
        }
        const int cost_rows = 40 + steps * (14 + 4 * k);
        const int cost_pop = nbatch * (nslot * (SB * NC * 2 + 20) + SB * ((NC + 1) / 2) * 7);
        rowform = cost_rows < cost_pop;
    }
    if (rowform && !byteform) {
        const int nw = (S * NCT16 * (int)sizeof(TabT)) >> 2;
        for (int q = lane; q < nw; q += 64) tab32[q] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (PRE) {
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if (it * 64 < nd) mi_bin_rows<L, NXY, KM, WIDE>(wd[it], tab32, k, flagX, flagY);
        } else {
            for (int d0 = 0; d0 < nd; d0 += 64) {
                mi_load_words<L, KM>(wd[0], pn, ph, xn, xh, yn, yh, W2, zs, k, d0 + lane, nd, P.n);
                mi_bin_rows<L, NXY, KM, WIDE>(wd[0], tab32, k, flagX, flagY);
            }
        }
    }
    for (int b = 0; b < (rowform ? 0 : nbatch); ++b) {
        unsigned acc[SBMAX][NCT];
#pragma unroll
        for (int s = 0; s < SBMAX; ++s)
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc[s][c] = 0u;
        if (PRE) {
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if (it * 64 < nd) mi_count_words<L, NXY, KM>(wd[it], acc, b, k, SB, flagX, flagY, P.dense != 0, tot_sep, viewX, viewY);
        } else {
            for (int d0 = 0; d0 < nd; d0 += 64) {
                mi_load_words<L, KM>(wd[0], pn, ph, xn, xh, yn, yh, W2, zs, k, d0 + lane, nd, P.n);
                mi_count_words<L, NXY, KM>(wd[0], acc, b, k, SB, flagX, flagY, P.dense != 0, tot_sep, viewX, viewY);
            }
        }
        // two 16-bit counts per register (a count never exceeds n <= 65535), six DPP adds each (independent chains: the
        // compiler interleaves them and the DPP wait states disappear), then lane 63 files the totals in one go
        if constexpr (WIDE) {
#pragma unroll
            for (int s = 0; s < SBMAX; ++s)
#pragma unroll
                for (int c = 0; c < NCT; ++c)
                    if (c < NC || tot_sep) {
                        const unsigned r = mi_dpp_sum63(acc[s][c]);
                        if (lane == 63 && s < SB) tab[(b * SB + s) * NCT16 + c] = r;
                    }
        } else {
            unsigned red[SBMAX][NCT16 / 2];
    #pragma unroll
            for (int s = 0; s < SBMAX; ++s)
    #pragma unroll
                for (int q = 0; q < NCT16 / 2; ++q) {
                    const unsigned lo = acc[s][2 * q], hi = (2 * q + 1 < NCT) ? acc[s][2 * q + 1] : 0u;
                    // unconditional for the cell pairs (strata beyond SB hold zeros): straight-line code, chains interleave
                    red[s][q] = (2 * q < NC) ? mi_dpp_sum63(lo | (hi << 16)) : 0u;
                }
            if (tot_sep) {
    #pragma unroll
                for (int s = 0; s < SBMAX; ++s)
    #pragma unroll
                    for (int q = 0; q < NCT16 / 2; ++q)
                        if (2 * q >= NC) red[s][q] = mi_dpp_sum63(acc[s][2 * q] | ((2 * q + 1 < NCT ? acc[s][2 * q + 1] : 0u) << 16));
            }
            if (lane == 63) {
    #pragma unroll
                for (int s = 0; s < SBMAX; ++s)
                    if (s < SB) {
    #pragma unroll
                        for (int q = 0; q < NCT16 / 2; ++q)
                            if (2 * q < NC || tot_sep) tab32[((b * SB + s) * NCT16) / 2 + q] = red[s][q];
                    }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long pt1 = P.prof ? __builtin_readcyclecounter() : 0ull;
    // ---- lanes <-> strata: occupancy, n_obs ----
    int n_nonempty = 0, zmax = -1, key0_seen = 0;
    long long n_obs = 0, n_counted = 0;
    for (int base = 0; base < S; base += 64) {
        const int key = base + lane;
        if (key < S) {
            const TabT *t = tab + key * NCT16;
            int sub = 0;
#pragma unroll
            for (int c = 0; c < NC; ++c) sub += (int)t[c];
            const int tot = tot_sep ? (int)t[NC] : sub;
            if (tot > 0) {
                ++n_nonempty;
                zmax = key;  // k = 1: the key is the Z value itself
                if (key == 0) key0_seen = 1;
            }
            n_obs += sub;       // sum(sub_ctab)
            n_counted += tot;   // rows that entered the table
        }
    }
    n_nonempty = wave_sum_i(n_nonempty);
    key0_seen = __builtin_amdgcn_readlane(key0_seen, 0);
    n_obs = wave_sum_ll(n_obs);
    // levels_z
    int levels_z;
    if (k == 0) {
        levels_z = 1;
    } else if (special_k1) {
        const int zm = wave_max_i(zmax);
        levels_z = zm < 0 ? 1 : zm + 1;  // contingency.jl:168-176,186,229
    } else if (any_flag && !P.dense) {
        // distinct keys among counted rows, +1 if uncounted rows exist and the all-zero key was not among them
        n_counted = wave_sum_ll(n_counted);
        levels_z = n_nonempty + ((P.n - n_counted > 0 && !key0_seen) ? 1 : 0);
    } else {
        levels_z = n_nonempty;  // all rows are in the table (dense rule: level_map! misc.jl:162-184)
    }
    // power (tests.jl:58 / :210)
    bool power;
    if (k == 0)
        power = (n_obs >= P.n_obs_min) && (((double)n_obs / (double)((long long)lx * ly)) > (double)P.hps);
    else
        power = ((double)n_obs / (double)((long long)lx * ly * levels_z)) > (double)P.hps;
    const unsigned long long pt2 = P.prof ? __builtin_readcyclecounter() : 0ull;
    if (P.prof && lane == 0) {
        P.prof[0] = pt1 - pt0;
        P.prof[1] = pt2 - pt1;
        P.prof[4] = 1ull;
    }
    if (!power) return res;
    // ---- mutual information (statfuns.jl:163-254): lanes <-> (stratum, cell) pairs ----
    double pos = 0.0, neg = 0.0;
    long long npos = 0, nneg = 0;
    int df_part = 0;
    const int npairs = S * NC;
    for (int base = 0; base < npairs; base += 64) {
        const int q = base + lane;
        if (q < npairs) {
            const int key = q / NC, c = q - key * NC;
            const int i = c % NXY, j = c / NXY;
            const TabT *t = tab + key * NCT16;
            long long mi_[NXY], mj_[NXY], mk = 0;
#pragma unroll
            for (int u = 0; u < NXY; ++u) mi_[u] = mj_[u] = 0;
            long long mine = 0;
#pragma unroll
            for (int jj = 0; jj < NXY; ++jj)
#pragma unroll
                for (int ii = 0; ii < NXY; ++ii) {
                    const long long v = (ii < lx && jj < ly) ? (long long)t[ii + NXY * jj] : 0;  // marginals over 1:levels_x, 1:levels_y
                    mi_[ii] += v;
                    mj_[jj] += v;
                    mk += v;
                    if (ii == i && jj == j) mine = v;
                }
            long long my_mi = 0, my_mj = 0;
#pragma unroll
            for (int u = 0; u < NXY; ++u) {
                if (u == i) my_mi = mi_[u];
                if (u == j) my_mj = mj_[u];
            }
            if (mine != 0 && my_mi != 0 && my_mj != 0) {
                const double denom_k = (k == 0) ? (double)n_obs : (double)mk;  // 2-D form uses n_obs = sum(ctab)
                const double term = log((denom_k * (double)mine) / (double)(my_mi * my_mj)) * (double)mine;
                if (i == j) {
                    pos += term;
                    npos += mine;
                } else {
                    neg += term;
                    nneg += mine;
                }
            }
            if (c == 0) {  // statfuns.jl:281-297 adjust_df, once per stratum
                int alx = 0, aly = 0;
#pragma unroll
                for (int u = 0; u < NXY; ++u) {
                    alx += mi_[u] > 0;
                    aly += mj_[u] > 0;
                }
                alx = alx < 1 ? 1 : alx;
                aly = aly < 1 ? 1 : aly;
                df_part += (alx - 1) * (aly - 1);
            }
        }
    }
    pos = wave_sum_d(pos);
    neg = wave_sum_d(neg);
    npos = wave_sum_ll(npos);
    nneg = wave_sum_ll(nneg);
    const int df = wave_sum_i(df_part);
    const double nd_ = (k == 0) ? (double)n_obs : (double)(npos + nneg);
    double mi = (pos + neg) / nd_;
    if (neg * ((double)nneg / nd_) > pos * ((double)npos / nd_)) mi *= -1.0;
    if (P.prof && lane == 0) {
        P.prof[2] = __builtin_readcyclecounter() - pt2;
        P.prof[5] = 1ull;
    }
    res.stat = mi;
    res.pval = NAN;
    res.df = df;
    res.power = 1;
    res.g = 2.0 * fabs(mi) * (double)n_obs;
    res.n_obs = n_obs;
    return res;
}

// ------------------------------------------------------------------------------------------------
// GENERIC form (r04): discrete variables with MORE THAN THREE LEVELS (values 0 .. L - 1, L <= 8; the reference sizes its tables
// L x L x L^max_k for any L, types.jl:98-117, misc.jl:64-97; reachable with make_onehot = false meta data, preprocessing.jl:42-117).
// The bit-plane forms above hold two planes per variable; here the data is one byte per value and the wavefront bins row by row:
// lane l takes rows l, l + 64, ... and adds 1 to cell [key][x][y] of a 32-bit LDS table (ds_add_u32; key = sum_j z_j L^j).  Everything
// after the counting -- stratum occupancy, levels_z, power, mutual information, adjust_df -- is the same rule set as
// mi_test_core, written with run-time loop bounds (sub-table = levels sx.. of X, sy.. of Y).  A slow path by design (meta-variable
// scale problems): ~n / 64 LDS atomics per lane and test instead of a handful of popcounts.  tab: MIG_TAB32 32-bit words.
// ------------------------------------------------------------------------------------------------
#define MIG_MAX_L 62  // values 0 .. 61: the pair table L^2 + 1 of level 0 fits MIG_TAB32 words (r01-r04: 8)
#define MIG_TAB32 3840  // words of one wavefront's table: L^k strata x (L^2 cells + the stratum total) must fit (host check)
static __device__ __forceinline__ MiRes mi_test_core_gen(const MiDev &P, const int X_in, const int Y_in, const MiZs &zs_in, const int k_in,
                                                         unsigned *tab)
{
    const int X = __builtin_amdgcn_readfirstlane(X_in), Y = __builtin_amdgcn_readfirstlane(Y_in);
    const int k = __builtin_amdgcn_readfirstlane(k_in);
    int zv[MI_MAX_K];
#pragma unroll
    for (int q = 0; q < MI_MAX_K; ++q) zv[q] = __builtin_amdgcn_readfirstlane(zs_in.v[q]);
    const int lane = threadIdx.x & 63;
    const int L = P.L, LL = L * L, NCT = LL + 1;
    const bool flagX = P.nzmode && P.maxv[X] > 1, flagY = P.nzmode && P.maxv[Y] > 1;
    const bool any_flag = flagX || flagY;
    const bool special_k1 = (k == 1) && any_flag && !P.dense;  // contingency.jl:250-253 (sparse dispatch only)
    const int sx = flagX ? 1 : 0, sy = flagY ? 1 : 0;
    int lx, ly;
    if (P.nzmode) {  // tests.jl:200-203
        lx = L - sx;
        ly = L - sy;
    } else {
        lx = P.levels[X];
        ly = P.levels[Y];
    }
    MiRes res;
    res.stat = 0.0;
    res.pval = 1.0;
    res.df = 0;
    res.power = 0;
    res.g = 0.0;
    res.n_obs = 0;
    if (k == 0) {  // tests.jl:36 sufficient_power(X, Y, data, ...) pre-check (tests.jl:9-20)
        bool ok = (long long)P.n >= P.n_obs_min;
        if (ok) {
            const long long vx = P.levels[X], vy = P.levels[Y];
            const long long ox = vx > 1 ? 2 : 1, oy = vy > 1 ? 2 : 1;
            ok = ((double)P.n / (double)((vx - ox) * (vy - oy))) > (double)P.hps;
        }
        if (!ok) return res;
    }
    int S = 1;
    for (int j = 0; j < k; ++j) S *= L;
    const bool tot_sep = P.dense && any_flag;
    const bool viewX = P.dense && P.view && P.nzmode && P.levels[X] > 2, viewY = P.dense && P.view && P.nzmode && P.levels[Y] > 2;
    // ---- counting ----
    // (table in device memory, r05: the counting atomics are performed in the L2, the reads below must not come from this CU's vector
    // cache -- agent-scope fences instead of wavefront-scope ones)
    for (int i = lane; i < S * NCT; i += 64) tab[i] = 0u;
    if (P.gtab)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    else
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const size_t n = (size_t)P.n;
    const unsigned char *cx = P.vals + (size_t)X * n, *cy = P.vals + (size_t)Y * n;
    for (int r = lane; r < P.n; r += 64) {
        const int x = cx[r], y = cy[r];
        int key = 0, mul = 1;
#pragma unroll
        for (int j = 0; j < MI_MAX_K; ++j)
            if (j < k) {
                key += (int)P.vals[(size_t)zv[j] * n + r] * mul;
                mul *= L;
            }
        const bool in_sub = (!flagX || x != 0) && (!flagY || y != 0);
        const bool in_tab = P.dense ? ((!viewX || x != 0) && (!viewY || y != 0)) : in_sub;
        if (in_sub) atomicAdd(&tab[key * NCT + x + L * y], 1u);
        if (tot_sep && in_tab) atomicAdd(&tab[key * NCT + LL], 1u);
    }
    if (P.gtab)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    else
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- lanes <-> strata: occupancy, n_obs ----
    int n_nonempty = 0, zmax = -1, key0_seen = 0;
    long long n_obs = 0, n_counted = 0;
    for (int base = 0; base < S; base += 64) {
        const int key = base + lane;
        if (key < S) {
            const unsigned *t = tab + key * NCT;
            int sub = 0;
            for (int yy = sy; yy < L; ++yy)
                for (int xx = sx; xx < L; ++xx) sub += (int)t[xx + L * yy];
            const int tot = tot_sep ? (int)t[LL] : sub;
            if (tot > 0) {
                ++n_nonempty;
                zmax = key;
                if (key == 0) key0_seen = 1;
            }
            n_obs += sub;
            n_counted += tot;
        }
    }
    n_nonempty = wave_sum_i(n_nonempty);
    key0_seen = __builtin_amdgcn_readlane(key0_seen, 0);
    n_obs = wave_sum_ll(n_obs);
    int levels_z;
    if (k == 0) {
        levels_z = 1;
    } else if (special_k1) {
        const int zm = wave_max_i(zmax);
        levels_z = zm < 0 ? 1 : zm + 1;
    } else if (any_flag && !P.dense) {
        n_counted = wave_sum_ll(n_counted);
        levels_z = n_nonempty + ((P.n - n_counted > 0 && !key0_seen) ? 1 : 0);
    } else {
        levels_z = n_nonempty;
    }
    bool power;
    if (k == 0)
        power = (n_obs >= P.n_obs_min) && (((double)n_obs / (double)((long long)lx * ly)) > (double)P.hps);
    else
        power = ((double)n_obs / (double)((long long)lx * ly * levels_z)) > (double)P.hps;
    if (!power) return res;
    // ---- mutual information: lanes <-> (stratum, cell) pairs; cell (i, j) = sub-table indices, table entry (sx + i, sy + j) ----
    double pos = 0.0, neg = 0.0;
    long long npos = 0, nneg = 0;
    int df_part = 0;
    const int npairs = S * LL;
#define MIG_V(ii, jj) (((ii) < lx && (jj) < ly && sx + (ii) < L && sy + (jj) < L) ? (long long)t[(sx + (ii)) + L * (sy + (jj))] : 0ll)
    for (int base = 0; base < npairs; base += 64) {
        const int q = base + lane;
        if (q < npairs) {
            const int key = q / LL, c = q - key * LL;
            const int i = c % L, j = c / L;
            const unsigned *t = tab + key * NCT;
            const long long mine = MIG_V(i, j);
            if (mine != 0 || c == 0) {
                long long my_mi = 0, my_mj = 0, mk = 0;
                for (int u = 0; u < L; ++u) {
                    my_mi += MIG_V(i, u);
                    my_mj += MIG_V(u, j);
                }
                int alx = 0, aly = 0;
                for (int u = 0; u < L; ++u) {
                    long long ri = 0, cj = 0;
                    for (int w = 0; w < L; ++w) {
                        ri += MIG_V(u, w);
                        cj += MIG_V(w, u);
                    }
                    mk += ri;
                    alx += ri > 0;
                    aly += cj > 0;
                }
                if (mine != 0 && my_mi != 0 && my_mj != 0) {
                    const double denom_k = (k == 0) ? (double)n_obs : (double)mk;
                    const double term = log((denom_k * (double)mine) / (double)(my_mi * my_mj)) * (double)mine;
                    if (i == j) {
                        pos += term;
                        npos += mine;
                    } else {
                        neg += term;
                        nneg += mine;
                    }
                }
                if (c == 0) {  // adjust_df (statfuns.jl:281-297), once per stratum
                    alx = alx < 1 ? 1 : alx;
                    aly = aly < 1 ? 1 : aly;
                    df_part += (alx - 1) * (aly - 1);
                }
            }
        }
    }
#undef MIG_V
    pos = wave_sum_d(pos);
    neg = wave_sum_d(neg);
    npos = wave_sum_ll(npos);
    nneg = wave_sum_ll(nneg);
    const int df = wave_sum_i(df_part);
    const double nd_ = (k == 0) ? (double)n_obs : (double)(npos + nneg);
    double mi = (pos + neg) / nd_;
    if (neg * ((double)nneg / nd_) > pos * ((double)npos / nd_)) mi *= -1.0;
    res.stat = mi;
    res.pval = NAN;
    res.df = df;
    res.power = 1;
    res.g = 2.0 * fabs(mi) * (double)n_obs;
    res.n_obs = n_obs;
    return res;
}

// ------------------------------------------------------------------------------------------------
// FOUR tests per wavefront (r03): row r of 16 lanes (one DPP row) evaluates (X, Y | Zs_r) -- four consecutive subsets of one
// test_subsets job, same X, Y and k.  n <= 5120 (lane i of a row owns the 32-row words i, i + 16, ..., i + 144 of every plane),
// k <= 3, NXY = 2.  Why: at n = 5000 a lane of the one-test form holds 2.45 words per plane, and the wavefront spends more
// instructions reducing 64 lanes (six DPP steps per packed counter, ~490 per k = 3 test) and waiting for its one memory round trip
// than counting; here a reduction ends inside the row (four DPP steps for four tests at once), the lanes are 98 % occupied, and
// four tests share one memory round trip -- the sequential prefix of a job (the chain that bounds the discrete stage) advances
// four subsets per step.  Same tables, same rules, same arithmetic per test as mi_test_core (the cell counts are integers; the
// Float64 sums over (stratum, cell) pairs run over 16 lanes instead of 64, in a different order: within the 1e-12 of
// DESIGN.md section 2).  Every lane of a row returns that row's result.
// ------------------------------------------------------------------------------------------------
#define MI4_NW 10      // words per lane and plane
#define MI4_N 5120     // samples
#define MI4_TAB16 192  // u16 entries of one row's table: 27 strata x 6 (4 cells + total + pad), rounded up

// sum over the 16 lanes of a DPP row, every lane gets it
static __device__ __forceinline__ unsigned mi_row_sum_u(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false);   // quad_perm:[1,0,3,2]
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false);   // quad_perm:[2,3,0,1]
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);  // row_ror:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);  // row_ror:8
    return v;
}
static __device__ __forceinline__ int mi_row_max_i(int v)
{
#define MI_ROW_MAX(ctrl)                                                         \
    {                                                                            \
        const int t = __builtin_amdgcn_update_dpp(v, v, ctrl, 0xf, 0xf, false);  \
        v = t > v ? t : v;                                                       \
    }
    MI_ROW_MAX(0xb1)
    MI_ROW_MAX(0x4e)
    MI_ROW_MAX(0x124)
    MI_ROW_MAX(0x128)
#undef MI_ROW_MAX
    return v;
}
static __device__ __forceinline__ double mi_row_sum_d(double v)
{
#define MI_ROW_ADDD(ctrl)                                                                                           \
    {                                                                                                               \
        const long long b = __double_as_longlong(v);                                                                \
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, ctrl, 0xf, 0xf, false);        \
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), ctrl, 0xf, 0xf, false); \
        v += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));                                \
    }
    MI_ROW_ADDD(0xb1)
    MI_ROW_ADDD(0x4e)
    MI_ROW_ADDD(0x124)
    MI_ROW_ADDD(0x128)
#undef MI_ROW_ADDD
    return v;
}

// zrow: the conditioning variables of THIS lane's row; k (1..3) and X, Y are wave-uniform.  tab4: MI4_TAB16 u16 per row.
template <int L>
// y_lane: the Y variable of THIS lane's row.  The four rows of a step are four subsets of one (X, Y) job (y_lane = Y_in) or the first
// subsets of four candidates of one target (mi_first4: y_lane differs per row; Y_in is then any of them -- the caller makes sure they
// agree in everything the uniform decisions below read: levels, maxv > 1).
static __device__ __forceinline__ MiRes mi_test_core4(const MiDev &P, const int X_in, const int Y_in, const int (&zrow)[3], const int k_in,
                                                      unsigned short *tab4, const int y_lane)
{
    constexpr int NXY = 2, NC = 4, NCT = 5, NCT16 = 6, SBMAX = L * L;
    const int X = __builtin_amdgcn_readfirstlane(X_in), Y = __builtin_amdgcn_readfirstlane(Y_in);
    const int k = __builtin_amdgcn_readfirstlane(k_in);
    const int lane = threadIdx.x & 63, li = lane & 15, row = lane >> 4;
    const bool flagX = P.nzmode && P.maxv[X] > 1, flagY = P.nzmode && P.maxv[Y] > 1;
    const bool any_flag = flagX || flagY;
    const bool special_k1 = (k == 1) && any_flag && !P.dense;  // contingency.jl:250-253 (sparse dispatch only)
    int lx, ly;
    if (P.nzmode) {  // tests.jl:200-203
        lx = L - (flagX ? 1 : 0);
        ly = L - (flagY ? 1 : 0);
    } else {
        lx = P.levels[X];
        ly = P.levels[Y];
    }
    MiRes res;
    res.stat = 0.0;
    res.pval = 1.0;
    res.df = 0;
    res.power = 0;
    res.g = 0.0;
    res.n_obs = 0;
    int S = 1;
    for (int j = 0; j < k; ++j) S *= L;
    const int SB = k >= 2 ? SBMAX : S;
    const int nbatch = S / SB;
    const bool tot_sep = P.dense && any_flag;
    const bool viewX = P.dense && P.view && P.nzmode && P.levels[X] > 2, viewY = P.dense && P.view && P.nzmode && P.levels[Y] > 2;
    const unsigned *pn = (const unsigned *)P.nz, *ph = (const unsigned *)P.hi;
    const size_t W2 = 2 * (size_t)P.W;
    const unsigned *xn = pn + (size_t)X * W2, *yn = pn + (size_t)y_lane * W2;
    const unsigned *xh = (L == 3 && ph) ? ph + (size_t)X * W2 : nullptr, *yh = (L == 3 && ph) ? ph + (size_t)y_lane * W2 : nullptr;
    const int nd = (P.n + 31) >> 5;
    const int nw = (nd + 15) >> 4;  // words per lane (uniform)
    // ---- X / Y part of every cell mask of this lane's words (the same for the four rows; batch independent) ----
    unsigned cm[MI4_NW][NC], mt[MI4_NW];
#pragma unroll
    for (int w = 0; w < MI4_NW; ++w) {
        const int dw = li + 16 * w;
        const bool ok = w < nw && dw < nd;
        const unsigned wxn = ok ? xn[dw] : 0u, wyn = ok ? yn[dw] : 0u;
        const unsigned wxh = (ok && xh) ? xh[dw] : 0u, wyh = (ok && yh) ? yh[dw] : 0u;
        const int left = P.n - dw * 32;
        const unsigned vm = !ok ? 0u : (left >= 32 ? 0xffffffffu : ((1u << left) - 1u));
        unsigned msub = vm;
        if (flagX) msub &= wxn;
        if (flagY) msub &= wyn;
        unsigned mtab = msub;
        if (P.dense) {
            mtab = vm;
            if (viewX) mtab &= wxn;
            if (viewY) mtab &= wyn;
        }
        const unsigned xu = flagX ? wxh : wxn, yu = flagY ? wyh : wyn;
        cm[w][0] = msub & ~xu & ~yu;
        cm[w][1] = msub & xu & ~yu;
        cm[w][2] = msub & ~xu & yu;
        cm[w][3] = msub & xu & yu;
        mt[w] = mtab;
    }
    const unsigned *z0n = pn + (size_t)zrow[0] * W2, *z1n = pn + (size_t)zrow[k >= 2 ? 1 : 0] * W2, *z2n = pn + (size_t)zrow[k >= 3 ? 2 : 0] * W2;
    const unsigned *z0h = (L == 3 && ph) ? ph + (size_t)zrow[0] * W2 : nullptr;
    const unsigned *z1h = (L == 3 && ph) ? ph + (size_t)zrow[k >= 2 ? 1 : 0] * W2 : nullptr;
    const unsigned *z2h = (L == 3 && ph) ? ph + (size_t)zrow[k >= 3 ? 2 : 0] * W2 : nullptr;
    unsigned short *tab = tab4 + row * MI4_TAB16;
    unsigned *tab32 = (unsigned *)tab;
    // every word of the conditioning columns up front, all loads in flight together: ONE memory round trip per step (loading
    // inside the batch / word loops made a step 30 dependent round trips: 38 us, as long as four one-test steps)
    unsigned zwn[3][MI4_NW], zwh[3][MI4_NW];
#pragma unroll
    for (int w = 0; w < MI4_NW; ++w) {
        const int dw = li + 16 * w;
        const bool ok = w < nw && dw < nd;
        zwn[0][w] = ok ? z0n[dw] : 0u;
        zwh[0][w] = (ok && z0h) ? z0h[dw] : 0u;
        zwn[1][w] = (ok && k >= 2) ? z1n[dw] : 0u;
        zwh[1][w] = (ok && k >= 2 && z1h) ? z1h[dw] : 0u;
        zwn[2][w] = (ok && k >= 3) ? z2n[dw] : 0u;
        zwh[2][w] = (ok && k >= 3 && z2h) ? z2h[dw] : 0u;
    }
    for (int b = 0; b < nbatch; ++b) {  // the digit of Z_3 is fixed inside a batch
        unsigned acc[SBMAX][NCT];
#pragma unroll
        for (int s = 0; s < SBMAX; ++s)
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc[s][c] = 0u;
#pragma unroll
        for (int w = 0; w < MI4_NW; ++w)
            if (w < nw) {
                const unsigned a0n = zwn[0][w], a0h = zwh[0][w];
                const unsigned a1n = zwn[1][w], a1h = zwh[1][w];
                unsigned hm = 0xffffffffu;
                if (k >= 3) hm = mi_level_mask<L>(zwn[2][w], zwh[2][w], b);
                unsigned z0[L], z1[L];
#pragma unroll
                for (int d = 0; d < L; ++d) {
                    z0[d] = mi_level_mask<L>(a0n, a0h, d);
                    z1[d] = (k >= 2) ? (mi_level_mask<L>(a1n, a1h, d) & hm) : (d == 0 ? hm : 0u);
                }
#pragma unroll
                for (int s = 0; s < SBMAX; ++s)
                    if (s < SB) {
                        const unsigned zk = z0[s % L] & z1[s / L];
#pragma unroll
                        for (int c = 0; c < NC; ++c) acc[s][c] += (unsigned)__builtin_popcount(cm[w][c] & zk);
                        if (tot_sep) acc[s][NC] += (unsigned)__builtin_popcount(mt[w] & zk);
                    }
            }
        // two 16-bit counts per register (a count never exceeds n <= 5120), four DPP adds inside the row
#pragma unroll
        for (int s = 0; s < SBMAX; ++s)
            if (s < SB) {
                const unsigned r0 = mi_row_sum_u(acc[s][0] | (acc[s][1] << 16));
                const unsigned r1 = mi_row_sum_u(acc[s][2] | (acc[s][3] << 16));
                const unsigned r2 = tot_sep ? mi_row_sum_u(acc[s][4]) : 0u;
                if (li == 0) {
                    unsigned *t = tab32 + ((b * SB + s) * NCT16) / 2;
                    t[0] = r0;
                    t[1] = r1;
                    t[2] = r2;
                }
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- lanes of the row <-> strata: occupancy, n_obs ----
    int n_nonempty = 0, zmax = -1, key0_seen = 0;
    unsigned n_obs_u = 0u, n_counted_u = 0u;
    for (int base = 0; base < S; base += 16) {
        const int key = base + li;
        if (key < S) {
            const unsigned short *t = tab + key * NCT16;
            int sub = 0;
#pragma unroll
            for (int c = 0; c < NC; ++c) sub += (int)t[c];
            const int tot = tot_sep ? (int)t[NC] : sub;
            if (tot > 0) {
                ++n_nonempty;
                zmax = key;
                if (key == 0) key0_seen = 1;
            }
            n_obs_u += (unsigned)sub;
            n_counted_u += (unsigned)tot;
        }
    }
    n_nonempty = (int)mi_row_sum_u((unsigned)n_nonempty);
    key0_seen = (int)mi_row_sum_u((unsigned)key0_seen);  // only the lane with key 0 can have set it
    const long long n_obs = (long long)mi_row_sum_u(n_obs_u);
    int levels_z;
    if (special_k1) {
        const int zm = mi_row_max_i(zmax);
        levels_z = zm < 0 ? 1 : zm + 1;
    } else if (any_flag && !P.dense) {
        const long long n_counted = (long long)mi_row_sum_u(n_counted_u);
        levels_z = n_nonempty + ((P.n - n_counted > 0 && !key0_seen) ? 1 : 0);
    } else {
        levels_z = n_nonempty;
    }
    const bool power = ((double)n_obs / (double)((long long)lx * ly * levels_z)) > (double)P.hps;  // tests.jl:210
    // ---- mutual information: lanes of the row <-> (stratum, cell) pairs; rows without power ride along (masked) ----
    double pos = 0.0, neg = 0.0;
    unsigned npos = 0u, nneg = 0u;
    int df_part = 0;
    const int npairs = S * NC;
    for (int base = 0; base < npairs; base += 16) {
        const int q = base + li;
        if (q < npairs && power) {
            const int key = q >> 2, c = q & 3;
            const int i = c & 1, j = c >> 1;
            const unsigned short *t = tab + key * NCT16;
            long long mi_[NXY] = {0, 0}, mj_[NXY] = {0, 0}, mk = 0, mine = 0;
#pragma unroll
            for (int jj = 0; jj < NXY; ++jj)
#pragma unroll
                for (int ii = 0; ii < NXY; ++ii) {
                    const long long v = (ii < lx && jj < ly) ? (long long)t[ii + NXY * jj] : 0;
                    mi_[ii] += v;
                    mj_[jj] += v;
                    mk += v;
                    if (ii == i && jj == j) mine = v;
                }
            const long long my_mi = i == 0 ? mi_[0] : mi_[1], my_mj = j == 0 ? mj_[0] : mj_[1];
            if (mine != 0 && my_mi != 0 && my_mj != 0) {
                const double term = log(((double)mk * (double)mine) / (double)(my_mi * my_mj)) * (double)mine;
                if (i == j) {
                    pos += term;
                    npos += (unsigned)mine;
                } else {
                    neg += term;
                    nneg += (unsigned)mine;
                }
            }
            if (c == 0) {  // adjust_df, once per stratum
                int alx = (mi_[0] > 0) + (mi_[1] > 0), aly = (mj_[0] > 0) + (mj_[1] > 0);
                alx = alx < 1 ? 1 : alx;
                aly = aly < 1 ? 1 : aly;
                df_part += (alx - 1) * (aly - 1);
            }
        }
    }
    pos = mi_row_sum_d(pos);
    neg = mi_row_sum_d(neg);
    const long long np_ = (long long)mi_row_sum_u(npos), nn_ = (long long)mi_row_sum_u(nneg);
    const int df = (int)mi_row_sum_u((unsigned)df_part);
    if (!power) return res;
    const double nd_ = (double)(np_ + nn_);
    double mi = (pos + neg) / nd_;
    if (neg * ((double)nn_ / nd_) > pos * ((double)np_ / nd_)) mi *= -1.0;
    res.stat = mi;
    res.pval = NAN;
    res.df = df;
    res.power = 1;
    res.g = 2.0 * fabs(mi) * (double)n_obs;
    res.n_obs = n_obs;
    return res;
}

// p-value of a finished test (statfuns.jl:157-161); tests without power keep (0, 1)
static __device__ __forceinline__ double mi_res_pval(MiRes &r)
{
    if (isnan(r.pval)) r.pval = r.df > 0 ? mi_igamc(0.5 * (double)r.df, 0.5 * r.g) : 1.0;
    return r.pval;
}

// issig (tests.jl:1-3) without the p-value where G^2 is clearly on one side of the alpha quantile of its df: p is monotone
// in G^2, the table holds the quantile to ~1e-15, and inside a relative band of 1e-9 around it the exact Q(a, x) decides --
// the verdict is the one `p < alpha` gives, the incomplete gamma function is just not evaluated for it.
static __device__ __forceinline__ bool mi_issig(const MiDev &P, MiRes &r)
{
    if (!r.power) return false;
    if (!isnan(r.pval)) return r.pval < P.alpha;
    if (r.df <= 0) {
        r.pval = 1.0;
        return false;
    }
    if (P.gthr && r.df < P.gthr_n) {
        const double q = P.gthr[r.df];
        if (r.g > q * (1.0 + 1e-9)) return true;
        if (r.g < q * (1.0 - 1e-9)) return false;
    }
    return mi_res_pval(r) < P.alpha;
}

// every field through v_readfirstlane: inside a function that is called (not inlined) the arguments arrive in vector
// registers and the compiler no longer knows that they are the same in every lane
static __device__ __forceinline__ unsigned long long mi_rfl64(unsigned long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
static __device__ __forceinline__ MiDev mi_uniform(const MiDev &P)
{
    MiDev U;
    U.nz = (const unsigned long long *)mi_rfl64((unsigned long long)P.nz);
    U.hi = (const unsigned long long *)mi_rfl64((unsigned long long)P.hi);
    U.levels = (const int32_t *)mi_rfl64((unsigned long long)P.levels);
    U.maxv = (const int32_t *)mi_rfl64((unsigned long long)P.maxv);
    U.W = __builtin_amdgcn_readfirstlane(P.W);
    U.n = __builtin_amdgcn_readfirstlane(P.n);
    U.L = __builtin_amdgcn_readfirstlane(P.L);
    U.nzmode = __builtin_amdgcn_readfirstlane(P.nzmode);
    U.hps = __builtin_amdgcn_readfirstlane(P.hps);
    U.dense = __builtin_amdgcn_readfirstlane(P.dense);
    U.view = __builtin_amdgcn_readfirstlane(P.view);
    U.n_obs_min = (long long)mi_rfl64((unsigned long long)P.n_obs_min);
    U.gthr = (const double *)mi_rfl64((unsigned long long)P.gthr);
    U.gthr_n = __builtin_amdgcn_readfirstlane(P.gthr_n);
    U.alpha = __longlong_as_double((long long)mi_rfl64((unsigned long long)__double_as_longlong(P.alpha)));
    U.prof = (unsigned long long *)mi_rfl64((unsigned long long)P.prof);
    U.vals = (const unsigned char *)mi_rfl64((unsigned long long)P.vals);
    U.rowk = __builtin_amdgcn_readfirstlane(P.rowk);
    return U;
}

// Bookkeeping of one test inside a test_subsets enumeration (tests.jl:326-341).  Returns 1: the job stops at this test (not
// significant, or `last`: the max_tests-th test); 2: significant and the new maximum-p result ("later wins ties"); 0:
// significant, smaller p than the current maximum.  The p-value is evaluated only where its value matters: for the
// maximum-p candidate unless it is provably smaller than the current one -- Q(a, x) grows with a and falls with x, so a
// test with df <= df_best and G^2 > G^2_best (by a margin far above the rounding of Q) cannot tie or beat it -- and for
// the stopping test if the caller reports it (need_stop_p: the ABI returns it; HITON-PC drops a rejected candidate
// without looking at its p, hiton.jl:67-70).
struct MiBest {
    double p, stat, g;  // p = -3: none yet
    int df;
};
static __device__ __forceinline__ int mi_account(const MiDev &P, MiRes &t, bool last, MiBest &b, bool need_stop_p)
{
    const bool sig = mi_issig(P, t);
    if (!sig || last) {
        if (need_stop_p || sig)
            (void)mi_res_pval(t);
        else if (isnan(t.pval))
            t.pval = 1.0;  // any value >= alpha: a rejection
        return 1;
    }
    if (b.p > 1e-290 && t.df <= b.df && t.g > b.g * (1.0 + 1e-6)) return 0;
    const double p = mi_res_pval(t);
    if (p >= b.p) {
        b.p = p;
        b.stat = t.stat;
        b.g = t.g;
        b.df = t.df;
        return 2;
    }
    return 0;
}

// Kernels that call mi_test_core are templates on <L, NXY, PRE>; the host picks the instantiation once per context
// (fw_ctx::L, fw_ctx::mi_nxy, n <= MI_PRE_N): NXY = 3 only for FW_MI on data that holds the value 2 -- every test of such a context uses
// the 9-cell form, which is also correct for two-valued pairs (their extra cells stay empty).
MiDev fwi_mi_dev(const fw_ctx *ctx);
