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
// A conditional test is ONE WAVEFRONT: the k+2 column words are wave-uniform loads, lane l bins sample 64w+l into a
// per-wave LDS table (cell = x + L*y + L^2*key) with ds_add, then lanes <-> strata compute marginals, MI terms
// (fp64 log), df, and a wave reduction yields G2 / p.  Level 0 (all pairs) is a tiled AND+popcount kernel.
#include "fw_internal.h"

#include <algorithm>
#include <cmath>

#define MI_MAXCELL 256  // L*L*L^k <= 3*3*27 = 243
#define MI_MAX_K 3

struct MiDev {
    const unsigned long long *nz;
    const unsigned long long *hi;  // may be null when L == 2
    const int32_t *levels;
    const int32_t *maxv;
    int W, n, L, nzmode, hps;
    int dense;  // dense-matrix table rules (contingency.jl:7-56): every row counted, levels_z = distinct Z keys over all rows
    long long n_obs_min;
};

// ------------------------------------------------------------------------------------------------
// special functions: regularised upper incomplete gamma Q(a, x) (series / continued fraction, Cephes structure)
// stands in for ccdf(Chisq(df), g) = Q(df/2, g/2)  (statfuns.jl:157-161)
// ------------------------------------------------------------------------------------------------
__device__ double mi_igamc(double a, double x)
{
    if (isnan(a) || isnan(x)) return NAN;
    if (x <= 0.0 || a <= 0.0) return 1.0;
    if (isinf(x)) return 0.0;
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

__device__ __forceinline__ double mi_pval_dev(double mi_abs, int df, long long n_obs)
{
    const double g = 2.0 * mi_abs * (double)n_obs;
    return df > 0 ? mi_igamc(0.5 * (double)df, 0.5 * g) : 1.0;
}

__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ long long wave_sum_ll(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int t = __shfl_xor(v, o);
        v = t > v ? t : v;
    }
    return v;
}

// ------------------------------------------------------------------------------------------------
// one wavefront = one test (X, Y | zs[0..k-1]); tab = this wave's LDS table (MI_MAXCELL ints)
// ------------------------------------------------------------------------------------------------
struct MiRes {
    double stat, pval;
    int df, power;
};

// word held by lane i (wave-uniform i) -> every lane
__device__ __forceinline__ unsigned long long mi_rl64(unsigned long long v, int i)
{
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, i);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), i);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ MiRes mi_test_wave(const MiDev &P, int X, int Y, const int *zs, int k, int *tab)
{
    const int lane = threadIdx.x & 63;
    const int L = P.L, L2 = L * L;
    int nkeys = 1;
    for (int j = 0; j < k; ++j) nkeys *= L;
    for (int c = lane; c < MI_MAXCELL; c += 64) tab[c] = 0;
    const bool flagX = P.nzmode && P.maxv[X] > 1, flagY = P.nzmode && P.maxv[Y] > 1;
    const bool any_flag = flagX || flagY;
    const bool special_k1 = (k == 1) && any_flag && !P.dense;  // contingency.jl:250-253 (sparse dispatch only)
    const int sx = flagX ? 1 : 0, sy = flagY ? 1 : 0;
    int lx, ly;
    if (P.nzmode) {  // tests.jl:200-203: levels of the nz-adjusted sub-table
        lx = L - sx;
        ly = L - sy;
    } else {
        lx = P.levels[X];
        ly = P.levels[Y];
    }
    MiRes res;
    if (k == 0) {  // tests.jl:36 sufficient_power(X, Y, data, ...) pre-check (tests.jl:9-20)
        bool ok = (long long)P.n >= P.n_obs_min;
        if (ok) {
            const long long vx = P.levels[X], vy = P.levels[Y];
            const long long ox = vx > 1 ? 2 : 1, oy = vy > 1 ? 2 : 1;
            ok = ((double)P.n / (double)((vx - ox) * (vy - oy))) > (double)P.hps;
        }
        if (!ok) {
            res.stat = 0.0;
            res.pval = 1.0;
            res.df = 0;
            res.power = 0;
            return res;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- binning ----
    const unsigned long long *cx = P.nz + (size_t)X * P.W, *cy = P.nz + (size_t)Y * P.W;
    const unsigned long long *hx = P.hi ? P.hi + (size_t)X * P.W : nullptr, *hy = P.hi ? P.hi + (size_t)Y * P.W : nullptr;
    const unsigned long long *cz[MI_MAX_K], *hz[MI_MAX_K];
#pragma unroll
    for (int j = 0; j < MI_MAX_K; ++j) {
        cz[j] = (j < k) ? P.nz + (size_t)zs[j] * P.W : nullptr;
        hz[j] = (j < k && P.hi) ? P.hi + (size_t)zs[j] * P.W : nullptr;
    }
    int my_counted = 0;
    // Lane l first loads word w0 + l of every column (coalesced, all loads in flight at once); the words are then
    // handed to the whole wavefront one at a time with v_readlane.  (One dependent round of ~10 global loads per
    // word was ~1.5 us x 79 words = 120 us per test at cfg4: the kernel was bound by L2 latency, not by anything else.)
    for (int w0 = 0; w0 < P.W; w0 += 64) {
      const int wl = w0 + lane;
      const bool wok = wl < P.W;
      const unsigned long long rxn = wok ? cx[wl] : 0ull, ryn = wok ? cy[wl] : 0ull;
      const unsigned long long rxh = (wok && hx) ? hx[wl] : 0ull, ryh = (wok && hy) ? hy[wl] : 0ull;
      unsigned long long rzn[MI_MAX_K], rzh[MI_MAX_K];
#pragma unroll
      for (int j = 0; j < MI_MAX_K; ++j) {
          rzn[j] = (j < k && wok) ? cz[j][wl] : 0ull;
          rzh[j] = (j < k && wok && hz[j]) ? hz[j][wl] : 0ull;
      }
      const int nw = (P.W - w0) < 64 ? (P.W - w0) : 64;
      for (int wi = 0; wi < nw; ++wi) {
        const int w = w0 + wi;
        const int row = w * 64 + lane;
        const unsigned long long xn = mi_rl64(rxn, wi), yn = mi_rl64(ryn, wi);
        const unsigned long long xh = mi_rl64(rxh, wi), yh = mi_rl64(ryh, wi);
        const int xv = (int)((xn >> lane) & 1ull) + (int)((xh >> lane) & 1ull);
        const int yv = (int)((yn >> lane) & 1ull) + (int)((yh >> lane) & 1ull);
        int key = 0, mul = 1, anyz = 0;
#pragma unroll
        for (int j = 0; j < MI_MAX_K; ++j)
            if (j < k) {
                const unsigned long long zn = mi_rl64(rzn[j], wi);
                const unsigned long long zh = mi_rl64(rzh[j], wi);
                const int zv = (int)((zn >> lane) & 1ull) + (int)((zh >> lane) & 1ull);
                key += zv * mul;  // key = sum_j z_j * L^j (types.jl:32-39 cum_levels)
                mul *= L;
                anyz |= zv;
            }
        bool counted = row < P.n;
        if (P.dense)
            ;  // contingency.jl:42-56: the dense form visits every row; nz_adjust_cont_tab drops row/col 0 afterwards
        else if (any_flag)
            counted = counted && (!flagX || xv != 0) && (!flagY || yv != 0);  // rows the merge does not skip
        else
            counted = counted && ((xv | yv | anyz) != 0);  // rows the merge visits; the rest is added below
        if (counted) {
            atomicAdd(&tab[xv + L * yv + L2 * key], 1);
            ++my_counted;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int n_counted = wave_sum_i(my_counted);
    if (!any_flag && !P.dense) {
        // contingency.jl:462-476: never-visited (all-zero) rows go to cell (0, 0, stratum of the all-zero key)
        if (lane == 0) tab[0] += P.n - n_counted;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- lanes <-> strata ----
    long long nobs_part = 0, npos = 0, nneg = 0;
    int df_part = 0, nonempty = 0, zmax = 0;
    long long mk = 0;
    long long mi_[3] = {0, 0, 0}, mj_[3] = {0, 0, 0};
    int cell[3][3];
    bool key0_seen = false;
    const bool act = lane < nkeys;
    if (act) {
        const int *t = tab + L2 * lane;
        long long stratum_total = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int v = (i < L && j < L) ? t[i + L * j] : 0;
                cell[i][j] = v;
                stratum_total += v;
                if (i >= sx && j >= sy) nobs_part += v;                     // sum(sub_ctab)
                if (i >= sx && j >= sy && i - sx < lx && j - sy < ly) {    // marginals over 1:levels_x, 1:levels_y
                    mi_[i] += v;
                    mj_[j] += v;
                    mk += v;
                }
            }
        if (stratum_total > 0) {
            nonempty = 1;
            zmax = lane;  // k = 1: the key is the Z value itself
        }
        int alx = 0, aly = 0;  // statfuns.jl:281-297
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            alx += mi_[i] > 0;
            aly += mj_[i] > 0;
        }
        alx = alx < 1 ? 1 : alx;
        aly = aly < 1 ? 1 : aly;
        df_part = (alx - 1) * (aly - 1);
    }
    const int n_nonempty = wave_sum_i(nonempty);
    key0_seen = __shfl(nonempty, 0) != 0;
    const long long n_obs = wave_sum_ll(nobs_part);
    // levels_z
    int levels_z;
    if (k == 0) {
        levels_z = 1;
    } else if (special_k1) {
        const int zm = wave_max_i(nonempty ? zmax : -1);
        levels_z = zm < 0 ? 1 : zm + 1;  // contingency.jl:168-176,186,229
    } else if (any_flag && !P.dense) {
        // distinct keys among counted rows, +1 if uncounted rows exist and the all-zero key was not among them
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
    if (!power) {
        res.stat = 0.0;
        res.pval = 1.0;
        res.df = 0;
        res.power = 0;
        return res;
    }
    // ---- mutual information (statfuns.jl:163-254) ----
    double pos = 0.0, neg = 0.0;
    if (act) {
        const double denom_k = (k == 0) ? (double)n_obs : (double)mk;  // 2-D form uses n_obs = sum(ctab)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const bool inside = i >= sx && j >= sy && i - sx < lx && j - sy < ly;
                const long long c = cell[i][j];
                if (inside && c != 0 && mi_[i] != 0 && mj_[j] != 0) {
                    const double term = log((denom_k * (double)c) / (double)(mi_[i] * mj_[j])) * (double)c;
                    if (i - sx == j - sy) {
                        pos += term;
                        npos += c;
                    } else {
                        neg += term;
                        nneg += c;
                    }
                }
            }
    }
    pos = wave_sum_d(pos);
    neg = wave_sum_d(neg);
    npos = wave_sum_ll(npos);
    nneg = wave_sum_ll(nneg);
    const int df = wave_sum_i(df_part);
    const double nd = (k == 0) ? (double)n_obs : (double)(npos + nneg);
    double mi = (pos + neg) / nd;
    if (neg * ((double)nneg / nd) > pos * ((double)npos / nd)) mi *= -1.0;
    res.stat = mi;
    res.pval = mi_pval_dev(fabs(mi), df, n_obs);
    res.df = df;
    res.power = 1;
    return res;
}

// ------------------------------------------------------------------------------------------------
// batch of single tests: one wave per test
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mi_test_batch_kernel(MiDev P, long long m, const int32_t *__restrict__ X,
                                                            const int32_t *__restrict__ Y,
                                                            const long long *__restrict__ zoff,
                                                            const int32_t *__restrict__ zflat,
                                                            fw_test_result *__restrict__ out)
{
    __shared__ int s_tab[4][MI_MAXCELL];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + wave;
    if (t >= m) return;
    const int k = (int)(zoff[t + 1] - zoff[t]);
    int zs[MI_MAX_K];
    for (int q = 0; q < MI_MAX_K; ++q) zs[q] = (q < k) ? zflat[zoff[t] + q] : 0;
    const MiRes r = mi_test_wave(P, X[t], Y[t], zs, k, s_tab[wave]);
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
__device__ __forceinline__ unsigned long long mi_binom(long long m, int t)
{
    if (m < t) return 0ull;
    const unsigned long long u = (unsigned long long)m;
    switch (t) {
        case 0: return 1ull;
        case 1: return u;
        case 2: return u * (u - 1) / 2ull;
        default: return (u * (u - 1) / 2ull) * (u - 2) / 3ull;
    }
}

__device__ void mi_unrank(unsigned long long rem, int a, int s, int *pos)
{
    int prev = -1;
    for (int d = 0; d < s; ++d) {
        const int t = s - d;
        const unsigned long long tot = mi_binom(a - 1 - prev, t);
        int lo = prev + 1, hi = a - t;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tot - mi_binom(a - mid, t) <= rem)
                lo = mid;
            else
                hi = mid - 1;
        }
        rem -= tot - mi_binom(a - lo, t);
        pos[d] = lo;
        prev = lo;
    }
}

#define MI_RUN 4

__device__ __forceinline__ void mi_seg_body(const MiDev &P, const FwSeg *__restrict__ segs, const int32_t *__restrict__ accflat,
                                            FwSegOut *__restrict__ out, int max_k, double alpha, long long max_tests,
                                            const unsigned sidx /* segment this workgroup evaluates */)
{
    __shared__ int s_tab[4][MI_MAXCELL];
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
    for (int s = MI_MAX_K; s >= 1; --s) cnt[s] = (s <= max_k) ? mi_binom(a, s) : 0ull;
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
            int pos[MI_MAX_K] = {0, 0, 0};
            mi_unrank(rem, a, s, pos);
            for (unsigned long long r = r0; r < r1; ++r) {
                int zs[MI_MAX_K];
#pragma unroll
                for (int q = 0; q < MI_MAX_K; ++q) zs[q] = (q < s) ? gacc[pos[q]] : 0;
                const MiRes t = mi_test_wave(P, seg.X, seg.Y, zs, s, s_tab[wave]);
                ++my_done;
                const bool sig = (t.pval < alpha) && t.power;
                if (!sig || (max_tests > 0 && r + 1 >= (unsigned long long)max_tests)) {
                    my_stop = r;
                    stop_stat = t.stat;
                    stop_p = t.pval;
                    stop_df = t.df;
                    stop_pow = t.power;
                    break;
                }
                if (t.pval >= my_bp) {
                    my_bp = t.pval;
                    my_br = r;
                    my_bstat = t.stat;
                    my_bdf = t.df;
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
__global__ __launch_bounds__(256) void mi_subsets_seg_kernel(MiDev P, const FwSeg *__restrict__ segs,
                                                             const int32_t *__restrict__ accflat,
                                                             FwSegOut *__restrict__ out, int max_k, double alpha,
                                                             long long max_tests, const unsigned *__restrict__ ns_dev)
{
    // no grid-stride loop here: with the body inside a loop the compiler hoists its invariants and needs 254 VGPRs
    // (occupancy 1 instead of 3); the device-driven grid covers the whole segment list and surplus workgroups leave
    if (ns_dev && blockIdx.x >= *ns_dev) return;
    mi_seg_body(P, segs, accflat, out, max_k, alpha, max_tests, blockIdx.x);
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

__device__ __forceinline__ void mi_pair_epilogue(const MiDev &P, int X, int Y, int A, int B, int C, int D, const int32_t *cnt_nz,
                                                 const int32_t *cnt_hi, double alpha, const double *gthr, MiL0Counters *cnt,
                                                 unsigned long long cap, int32_t *out_i, int32_t *out_j, double *out_s,
                                                 double *out_p)
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
    // (unreliable pairs were already counted by the screening kernel)
    const bool keep = !unreliable && pval < alpha;
    const unsigned long long km = __ballot(keep);  // one atomic per wavefront
    unsigned long long base = 0;
    const int lane = threadIdx.x & 63;
    if (km != 0ull) {
        const int leader = __ffsll((long long)km) - 1;
        if (lane == leader) base = atomicAdd(&cnt->n_sig, (unsigned long long)__popcll(km));
        base = __shfl(base, leader);
    }
    if (keep) {
        const unsigned long long slot = base + (unsigned long long)__popcll(km & ((1ull << lane) - 1ull));
        if (slot < cap) {
            out_i[slot] = X;
            out_j[slot] = Y;
            out_s[slot] = stat;
            out_p[slot] = pval;
        }
    }
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
__device__ __forceinline__ int mi_pair_screen(const MiDev &P, const int4 mX, const int4 mY, int X, int Y, int A, int B, int C, int D,
                                              const float *__restrict__ xlnx, const float *__restrict__ lnx, const double *gthr,
                                              MiL0Counters *cnt, unsigned long long cap_c, MiCand *__restrict__ cands,
                                              MiCand *s_q, int *s_qn)
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
    const int sx = flagX ? 1 : 0, sy = flagY ? 1 : 0;
    const int lx = P.nzmode ? L - sx : vx, ly = P.nzmode ? L - sy : vy;
    const int tt[3][3] = {{t00, t01, t02}, {t10, t11, t12}, {t20, t21, t22}};
    int n_obs = 0, S = 0;
    int mi_[3] = {0, 0, 0}, mj_[3] = {0, 0, 0};  // indexed by sub-table row / column
    float g = 0.0f;
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
            g += xlnx[v];
        }
    reliable = reliable && (long long)n_obs >= P.n_obs_min && (long long)n_obs > (long long)P.hps * lx * ly;
    if (!reliable) return 1;  // counted per workgroup by the caller (one atomic per pair serialised the whole kernel)
    int alx = (mi_[0] > 0) + (mi_[1] > 0) + (mi_[2] > 0), aly = (mj_[0] > 0) + (mj_[1] > 0) + (mj_[2] > 0);
    alx = alx < 1 ? 1 : alx;
    aly = aly < 1 ? 1 : aly;
    const int df = (alx - 1) * (aly - 1);
    if (df == 0) return 0;  // p = 1
    g += (float)S * lnx[n_obs];
    g -= (xlnx[mi_[0]] + xlnx[mi_[1]] + xlnx[mi_[2]]) + (xlnx[mj_[0]] + xlnx[mj_[1]] + xlnx[mj_[2]]);
    // |g - G/2| <= 16 table roundings of <= 0.003 each at n <= 65536: far inside the margin below
    const double g32 = 2.0 * fabs((double)g);
    if (g32 < 0.99 * gthr[df] - (0.5 + 1e-4 * (double)P.n)) return 0;  // cannot reach the alpha quantile
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
    if (qs < L0_QCAP) {
        s_q[qs] = cd;
    } else {
        const unsigned long long slot = atomicAdd(&cnt->n_sig, 1ull);  // n_sig doubles as the candidate counter in kernel 1
        if (slot < cap_c) cands[slot] = cd;
    }
    return 0;
}

// kernel 2: exact Float64 statistic + p-value for the screened candidates, one thread each
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
                                                        int dbg /* profiling only: 1 no epilogue, 2 no loads, 4 no popcounts */)
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
    int b = blockIdx.x, bi = 0;
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
    for (int w0 = 0; w0 < P.W; w0 += L0_WC) {
        __syncthreads();
        for (int e = tid; e < L0_T * L0_WC; e += 256) {
            const int col = e / L0_WC, w = e % L0_WC;
            const int gx = bi * L0_T + col, gy = bj * L0_T + col;
            const bool wv = w0 + w < P.W && !(dbg & 2);
            sXn[w][col] = (gx < p && wv) ? P.nz[(size_t)gx * P.W + w0 + w] : 0ull;
            sYn[w][col] = (gy < p && wv) ? P.nz[(size_t)gy * P.W + w0 + w] : 0ull;
            if (HAS_HI) {
                sXh[w][col] = (gx < p && wv) ? P.hi[(size_t)gx * P.W + w0 + w] : 0ull;
                sYh[w][col] = (gy < p && wv) ? P.hi[(size_t)gy * P.W + w0 + w] : 0ull;
            }
        }
        __syncthreads();
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
                    A[u][v] += __popcll(xn[u] & yn[v]);
                    if (HAS_HI) {
                        B[u][v] += __popcll(xh[u] & yn[v]);
                        C[u][v] += __popcll(xn[u] & yh[v]);
                        D[u][v] += __popcll(xh[u] & yh[v]);
                    }
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

__global__ __launch_bounds__(256) void mi_level0_exact_kernel(MiDev P, const MiCand *__restrict__ cands, unsigned long long ncand,
                                                              const int32_t *__restrict__ cnt_nz, const int32_t *__restrict__ cnt_hi,
                                                              double alpha, const double *gthr, MiL0Counters *cnt,
                                                              unsigned long long cap, int32_t *out_i, int32_t *out_j,
                                                              double *out_s, double *out_p)
{
    const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= ncand) return;
    const MiCand c = cands[t];
    mi_pair_epilogue(P, c.X, c.Y, c.A, c.B, c.C, c.D, cnt_nz, cnt_hi, alpha, gthr, cnt, cap, out_i, out_j, out_s, out_p);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
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
    return P;
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
    for (int v = 0; v < p; ++v) {
        if (colptr[v + 1] < colptr[v]) return fw_fail(ctx, FW_ERR_ARG, "colptr not monotone at column %d", v);
        bool seen[4] = {false, false, false, false};
        int32_t mx = 0;
        int64_t prev_row = -1;
        for (int64_t j = colptr[v]; j < colptr[v + 1]; ++j) {
            const int32_t r = rowval[j], x = nzval[j];
            if (r < 0 || r >= n || r <= prev_row) return fw_fail(ctx, FW_ERR_ARG, "row indices of column %d are not sorted / in range", v);
            prev_row = r;
            if (x < 1 || x > 2)
                return fw_fail(ctx, FW_ERR_LIMIT, "discrete values must be 0, 1 or 2 (column %d holds %d); stored zeros are not allowed", v, x);
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
    ctx->L = maxv_all + 1;  // types.jl:89,110
    if (ctx->L < 2) ctx->L = 2;
    ctx->W = W;
    const size_t pb = sizeof(uint64_t) * (size_t)p * W;
    void **ptrs[] = {(void **)&ctx->d_nzbits, (void **)&ctx->d_hibits, (void **)&ctx->d_levels, (void **)&ctx->d_maxvals, (void **)&ctx->d_firstnz};
    for (void **q : ptrs)
        if (*q) {
            (void)hipFree(*q);
            *q = nullptr;
        }
    FW_HIP(ctx, hipMalloc((void **)&ctx->d_nzbits, pb));
    FW_HIP(ctx, hipMemcpy(ctx->d_nzbits, nzb.data(), pb, hipMemcpyHostToDevice));
    if (ctx->L > 2) {
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_hibits, pb));
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
    const int T = (p + L0_T - 1) / L0_T;
    const int nblk = T * (T + 1) / 2;
    const MiDev P = mi_dev(ctx);
    // ---- kernel 1: popcounts + exact reliability/df + Float32 screen -> candidate records ----
    static const int l0_dbg = getenv("FW_L0_DBG") ? atoi(getenv("FW_L0_DBG")) : 0;  // profiling only (invalid results)
    unsigned long long cap_c = (unsigned long long)std::min<long long>(npairs, 8ll << 20);
    if (cap_c < ctx->l0_cap_hint) cap_c = ctx->l0_cap_hint;  // a repeated call does not overflow (and re-run the kernel) again
    if (cap_c == 0) cap_c = 1;
    MiL0Counters h1{};
    double *d_gthr = nullptr;
    for (int attempt = 0;; ++attempt) {
        if ((rc = fw_dev_reserve(ctx, ctx->d_tmp0, 2 * sizeof(MiL0Counters) + 8 * sizeof(double)))) return rc;
        if ((rc = fw_dev_reserve(ctx, ctx->d_jobs, cap_c * sizeof(MiCand)))) return rc;
        FW_HIP(ctx, hipMemsetAsync(ctx->d_tmp0.ptr, 0, 2 * sizeof(MiL0Counters), ctx->stream));
        d_gthr = (double *)((char *)ctx->d_tmp0.ptr + 2 * sizeof(MiL0Counters));
        FW_HIP(ctx, hipMemcpyAsync(d_gthr, gthr, sizeof(gthr), hipMemcpyHostToDevice, ctx->stream));
        if (ctx->d_hibits)
            hipLaunchKernelGGL(mi_level0_kernel<true>, dim3(nblk), dim3(256), 0, ctx->stream, P, p, T, ctx->d_firstnz,
                               ctx->d_firstnz + p, d_gthr, (MiL0Counters *)ctx->d_tmp0.ptr, cap_c, (MiCand *)ctx->d_jobs.ptr, ctx->d_xlnx, ctx->d_xlnx + (ctx->P.n + 1), l0_dbg);
        else
            hipLaunchKernelGGL(mi_level0_kernel<false>, dim3(nblk), dim3(256), 0, ctx->stream, P, p, T, ctx->d_firstnz,
                               ctx->d_firstnz + p, d_gthr, (MiL0Counters *)ctx->d_tmp0.ptr, cap_c, (MiCand *)ctx->d_jobs.ptr, ctx->d_xlnx, ctx->d_xlnx + (ctx->P.n + 1), l0_dbg);
        FW_HIP(ctx, hipGetLastError());
        FW_HIP(ctx, hipMemcpyAsync(&h1, ctx->d_tmp0.ptr, sizeof(h1), hipMemcpyDeviceToHost, ctx->stream));
        FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->cnt.kernel_launches += 1;
        if (h1.n_sig > ctx->l0_cap_hint) ctx->l0_cap_hint = h1.n_sig;
        if (h1.n_sig <= cap_c) break;
        if (attempt == 1) return fw_fail(ctx, FW_ERR_DEVICE, "discrete level-0: candidate buffer overflow twice");
        cap_c = h1.n_sig;
    }
    const unsigned long long ncand = h1.n_sig;
    *m_reliable = npairs - (long long)h1.n_unreliable;
    if (getenv("FW_L0_VERBOSE")) fprintf(stderr, "[fw] discrete level-0: pairs %lld reliable %lld candidates %llu\n", npairs, (long long)*m_reliable, ncand);
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
    hipLaunchKernelGGL(mi_level0_exact_kernel, dim3((unsigned)((ncand + 255) / 256)), dim3(256), 0, ctx->stream, P,
                       (const MiCand *)ctx->d_jobs.ptr, ncand, ctx->d_firstnz, ctx->d_firstnz + p, ctx->P.alpha, d_gthr, d_cnt2,
                       ncand, oi, oj, os, op);
    FW_HIP(ctx, hipGetLastError());
    MiL0Counters h2{};
    FW_HIP(ctx, hipMemcpyAsync(&h2, d_cnt2, sizeof(h2), hipMemcpyDeviceToHost, ctx->stream));
    FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cnt.kernel_launches += 1;
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
    hipLaunchKernelGGL(mi_subsets_seg_kernel, dim3(grid), dim3(256), 0, stream, mi_dev(ctx), d_segs, d_acc, d_out, ctx->P.max_k,
                       ctx->P.alpha, (long long)ctx->P.max_tests, d_ns);
    FW_HIP(ctx, hipGetLastError());
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
    hipLaunchKernelGGL(mi_test_batch_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, ctx->stream, mi_dev(ctx), (long long)m,
                       dX, dY, dz, (const int32_t *)ctx->d_acc.ptr, (fw_test_result *)ctx->d_out.ptr);
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipMemcpyAsync(out, ctx->d_out.ptr, (size_t)m * sizeof(fw_test_result), hipMemcpyDeviceToHost, ctx->stream));
    FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cnt.kernel_launches += 1;
    return FW_OK;
}

int fwi_mi_segments(fw_ctx *ctx, int64_t nseg, const FwSeg *d_segs, const int32_t *d_acc, FwSegOut *d_out, FwPoolBuf &pb)
{
    if (nseg == 0) return FW_OK;
    FW_HIP(ctx, hipEventRecord(pb.ev0, pb.launch_stream));
    hipLaunchKernelGGL(mi_subsets_seg_kernel, dim3((unsigned)nseg), dim3(256), 0, pb.launch_stream, mi_dev(ctx), d_segs, d_acc, d_out,
                       ctx->P.max_k, ctx->P.alpha, (long long)ctx->P.max_tests, (const unsigned *)nullptr);
    FW_HIP(ctx, hipGetLastError());
    FW_HIP(ctx, hipEventRecord(pb.ev1, pb.launch_stream));
    return FW_OK;
}
