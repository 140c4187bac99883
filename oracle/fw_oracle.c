/*
 * fw_oracle.c -- CPU ORACLE for the FlashWeave conditional-independence hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libflashweave_amd.so) never
 * links, calls or falls back to anything in this directory.
 *
 * What it is: a plain-C restatement of the reference's ALGORITHMS (Julia, /root/reference,
 * FlashWeave.jl v0.19.2+master) for the path named by BASELINE.json.north_star.  Every
 * function cites the reference file:line it follows.  The reference cannot be executed
 * here (pure Julia, no julia binary, no network), so this restatement is pinned against
 * the reference's own golden vectors instead (tests/test_oracle_golden.py):
 *   test/data/tests_expected.tsv, test/contingency.jl:5-24, test/statfuns.jl:24-71,
 *   test/data/learning_expected/ (eight .edgelist files).
 *
 * Third-party arithmetic restated here (absent from /root/reference, Project.toml:29-48):
 *   Distributions.ccdf(Chisq(df), g) -> igamc(df/2, g/2)   (Cephes-style series / continued fraction)
 *   Distributions.ccdf(Normal(), z)  -> erfc(z * invsqrt2) / 2  (StatsFuns.normccdf)
 *   Combinatorics.combinations       -> lexicographic k-subsets of positions
 *
 * Conventions: variable indices are 0-based; data is n samples x p variables;
 * sparse data is CSC with 0-based row indices; dense matrices are column-major.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: Julia never contracts a*b+c).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define FWO_MI 0
#define FWO_MI_NZ 1
#define FWO_FZ 2
#define FWO_FZ_NZ 3
#define FWO_IS_CONT(c) ((c)->kind >= FWO_FZ)

typedef struct {
    double stat;
    double pval;
    int64_t df;
    int32_t suff_power;
    int32_t pad;
} fwo_result; /* types.jl:140-145 TestResult */

typedef struct fwo_ctx {
    int kind; /* FWO_MI / FWO_MI_NZ / FWO_FZ */
    int nz;   /* is_zero_adjusted, types.jl:61-64 */
    int n, p;
    /* discrete data */
    int sparse;
    const int64_t *colptr; /* p+1 */
    const int32_t *rowval; /* 0-based, sorted within a column */
    const int32_t *nzval;
    const int32_t *dense; /* n x p column-major */
    int32_t *levels;      /* misc.jl:64-81 */
    int32_t *max_vals;    /* misc.jl:84-97 */
    int L;                /* maximum(max_vals)+1, types.jl:89,110 */
    /* continuous */
    const float *cor32; /* p x p, ContType = Float32 (learning.jl:44) */
    const double *cor64; /* p x p, ContType = Float64 (test convenience wrapper, tests.jl:272) */
    int n_obs;          /* size(data, 1) for fz tests */
    /* fz_nz (HE-S): the normalised matrix itself (clr_nz, zeros = absences), n x p column-major */
    const double *fdata; /* values widened to double (exact) */
    int fdata_f32;       /* 1: the reference's element type is Float32 (prec = 32), 0: Float64 */
    const uint8_t *rowmask; /* dense discrete data: rows of the current view (hiton.jl:41-50 prepare_nzdata), NULL = all rows */
    int fz_stream;       /* "fz" with fdata attached: conditional tests through pcor (StatsBase.partialcor), not pcor_rec */
    /* scratch (one MiTestCond sized for max_k, hiton.jl:192) */
    int max_k;
    int64_t nstrata_cap; /* L^max_k (+1 slack) */
    int64_t *ctab;       /* L x L x nstrata_cap, column-major */
    int64_t *marg_i, *marg_j, *marg_k;
    int32_t *zmap;   /* z_map_arr, types.jl:26-46 */
    int64_t zmap_len;
    int32_t *zrow; /* dense level_map! scratch z[i] */
    int64_t cum_levels[16];
} fwo_ctx;

/* ------------------------------------------------------------------------------------------
 * Special functions
 * ---------------------------------------------------------------------------------------- */

/* Regularised upper incomplete gamma Q(a, x); Cephes igamc/igam structure.
 * Stands in for Distributions.ccdf(Chisq(df), g) = Q(df/2, g/2) (statfuns.jl:159). */
static double fwo_igam_series(double a, double x)
{
    double ax = a * log(x) - x - lgamma(a);
    if (ax < -745.2) return 0.0;
    ax = exp(ax);
    double r = a, c = 1.0, ans = 1.0;
    do {
        r += 1.0;
        c *= x / r;
        ans += c;
    } while (c / ans > 1.1102230246251565e-16);
    return ans * ax / a;
}

static double fwo_igamc(double a, double x)
{
    if (isnan(a) || isnan(x)) return NAN;
    if (x <= 0.0 || a <= 0.0) return 1.0;
    if (isinf(x)) return 0.0;
    if (x < 1.0 || x < a) return 1.0 - fwo_igam_series(a, x);
    double ax = a * log(x) - x - lgamma(a);
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
        double yc = y * c;
        double pk = pkm1 * z - pkm2 * yc;
        double qk = qkm1 * z - qkm2 * yc;
        if (qk != 0.0) {
            double r = pk / qk;
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

double fwo_chisq_ccdf(int64_t df, double g) { return fwo_igamc(0.5 * (double)df, 0.5 * g); }

/* statfuns.jl:157-161 mi_pval */
double fwo_mi_pval(double mi, int64_t df, int64_t n_obs)
{
    double g_stat = 2.0 * mi * (double)n_obs;
    return df > 0 ? fwo_chisq_ccdf(df, g_stat) : 1.0;
}

/* statfuns.jl:3-11 fisher_z_transform */
static double fwo_fisher_z(double p, int64_t n, int64_t len_z)
{
    int64_t sample_factor = n - len_z - 3;
    if (sample_factor > 0) return (sqrt((double)sample_factor) / 2.0) * log((1.0 + p) / (1.0 - p));
    return 0.0;
}

/* statfuns.jl:13-17 fz_pval; ccdf(Normal(), x) = erfc(x * invsqrt2) / 2 (StatsFuns.normccdf) */
double fwo_fz_pval(double stat, int64_t n, int64_t len_z)
{
    double fz = fwo_fisher_z(stat, n, len_z);
    double cc = erfc(fabs(fz) * 0.7071067811865476) / 2.0;
    /* subnormal p-values are flushed to zero on both sides of the parity check: their last bit depends on the erfc
     * implementation (glibc / device library / openlibm), and candidates are ordered by p (hiton.jl:212-215); flushed, they
     * tie and the reference's stable index order applies -- see fz_pval_dev in csrc/fw_fz.hip and DESIGN.md section 2 */
    double p = cc * 2.0;
    return p < 2.2250738585072014e-308 ? 0.0 : p;
}

/* ------------------------------------------------------------------------------------------
 * Context
 * ---------------------------------------------------------------------------------------- */

static int64_t ipow64(int64_t b, int e)
{
    int64_t r = 1;
    while (e-- > 0) r *= b;
    return r;
}

static int cmp_i32(const void *a, const void *b)
{
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

/* misc.jl:64-97 get_levels / get_max_vals (sparse and dense forms) */
static void fwo_compute_levels(fwo_ctx *c)
{
    c->levels = (int32_t *)calloc((size_t)c->p, sizeof(int32_t));
    c->max_vals = (int32_t *)calloc((size_t)c->p, sizeof(int32_t));
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(c->n > 0 ? c->n : 1));
    for (int v = 0; v < c->p; ++v) {
        int64_t m = 0;
        if (c->sparse) {
            for (int64_t j = c->colptr[v]; j < c->colptr[v + 1]; ++j) tmp[m++] = c->nzval[j];
        } else {
            for (int i = 0; i < c->n; ++i) tmp[m++] = c->dense[(int64_t)v * c->n + i];
        }
        qsort(tmp, (size_t)m, sizeof(int32_t), cmp_i32);
        int32_t uniq = 0, mx = 0;
        for (int64_t j = 0; j < m; ++j) {
            if (j == 0 || tmp[j] != tmp[j - 1]) ++uniq;
        }
        if (m > 0) mx = tmp[m - 1];
        if (c->sparse) {
            /* misc.jl:70-71: implicit zero counts as a level when some row is empty */
            if ((int64_t)c->n > m) ++uniq;
            /* misc.jl:86: zero if the column is empty */
            if (m == 0) mx = 0;
        }
        c->levels[v] = uniq;
        c->max_vals[v] = mx;
    }
    free(tmp);
    int mx = 0;
    for (int v = 0; v < c->p; ++v)
        if (c->max_vals[v] > mx) mx = c->max_vals[v];
    c->L = mx + 1;
}

static void fwo_alloc_scratch(fwo_ctx *c, int max_k)
{
    c->max_k = max_k;
    int64_t L = c->L;
    c->nstrata_cap = ipow64(L, max_k > 0 ? max_k : 0) + 2;
    c->ctab = (int64_t *)calloc((size_t)(L * L * c->nstrata_cap), sizeof(int64_t));
    c->marg_i = (int64_t *)calloc((size_t)(L * c->nstrata_cap), sizeof(int64_t));
    c->marg_j = (int64_t *)calloc((size_t)(L * c->nstrata_cap), sizeof(int64_t));
    c->marg_k = (int64_t *)calloc((size_t)c->nstrata_cap, sizeof(int64_t));
    /* types.jl:32-46 ZMapper: cum_levels[j] = L^(j-1); len = L + sum_j L*cum_levels[j] */
    int64_t len = L;
    for (int j = 0; j < (max_k > 0 ? max_k : 1); ++j) {
        c->cum_levels[j] = ipow64(L, j);
        len += L * c->cum_levels[j];
    }
    c->zmap_len = len + 2;
    c->zmap = (int32_t *)malloc(sizeof(int32_t) * (size_t)c->zmap_len);
    c->zrow = (int32_t *)malloc(sizeof(int32_t) * (size_t)(c->n > 0 ? c->n : 1));
}

fwo_ctx *fwo_create_discrete_sparse(int n, int p, const int64_t *colptr, const int32_t *rowval, const int32_t *nzval,
                                    int nz, int max_k)
{
    fwo_ctx *c = (fwo_ctx *)calloc(1, sizeof(fwo_ctx));
    c->kind = nz ? FWO_MI_NZ : FWO_MI;
    c->nz = nz;
    c->n = n;
    c->p = p;
    c->sparse = 1;
    c->colptr = colptr;
    c->rowval = rowval;
    c->nzval = nzval;
    fwo_compute_levels(c);
    fwo_alloc_scratch(c, max_k);
    return c;
}

fwo_ctx *fwo_create_discrete_dense(int n, int p, const int32_t *data_colmajor, int nz, int max_k)
{
    fwo_ctx *c = (fwo_ctx *)calloc(1, sizeof(fwo_ctx));
    c->kind = nz ? FWO_MI_NZ : FWO_MI;
    c->nz = nz;
    c->n = n;
    c->p = p;
    c->sparse = 0;
    c->dense = data_colmajor;
    fwo_compute_levels(c);
    fwo_alloc_scratch(c, max_k);
    return c;
}

/* cor32 XOR cor64 is non-NULL; n_obs = number of samples the matrix was computed from */
fwo_ctx *fwo_create_fz(int n_obs, int p, const float *cor32, const double *cor64)
{
    fwo_ctx *c = (fwo_ctx *)calloc(1, sizeof(fwo_ctx));
    c->kind = FWO_FZ;
    c->nz = 0;
    c->n = n_obs;
    c->n_obs = n_obs;
    c->p = p;
    c->cor32 = cor32;
    c->cor64 = cor64;
    return c;
}

/* fz_nz: data = n x p column-major values (Float32 values widened exactly if is_f32) */
fwo_ctx *fwo_create_fz_nz(int n, int p, const double *data, int is_f32)
{
    fwo_ctx *c = (fwo_ctx *)calloc(1, sizeof(fwo_ctx));
    c->kind = FWO_FZ_NZ;
    c->nz = 1;
    c->n = n;
    c->n_obs = n;
    c->p = p;
    c->fdata = data;
    c->fdata_f32 = is_f32;
    return c;
}

void fwo_destroy(fwo_ctx *c)
{
    if (!c) return;
    free(c->levels);
    free(c->max_vals);
    free(c->ctab);
    free(c->marg_i);
    free(c->marg_j);
    free(c->marg_k);
    free(c->zmap);
    free(c->zrow);
    free(c);
}

int fwo_L(const fwo_ctx *c) { return c->L; }
void fwo_get_levels(const fwo_ctx *c, int32_t *levels, int32_t *max_vals)
{
    memcpy(levels, c->levels, sizeof(int32_t) * (size_t)c->p);
    memcpy(max_vals, c->max_vals, sizeof(int32_t) * (size_t)c->p);
}

/* ------------------------------------------------------------------------------------------
 * Contingency tables
 * ---------------------------------------------------------------------------------------- */

#define CT(c, i, j, k) ((c)->ctab[(int64_t)(i) + (c)->L * ((int64_t)(j) + (c)->L * (int64_t)(k))])

static void ctab_reset(fwo_ctx *c, int64_t nstrata)
{
    memset(c->ctab, 0, sizeof(int64_t) * (size_t)(c->L * c->L * nstrata));
}

static inline int32_t dense_at(const fwo_ctx *c, int i, int v) { return c->dense[(int64_t)v * c->n + i]; }

/* contingency.jl:7-17 dense 2-way */
static void ctab2_dense(fwo_ctx *c, int X, int Y)
{
    ctab_reset(c, 1);
    for (int i = 0; i < c->n; ++i) {
        if (c->rowmask && !c->rowmask[i]) continue; /* rows outside the @view (hiton.jl:41-50) do not exist for this test */
        CT(c, dense_at(c, i, X), dense_at(c, i, Y), 0) += 1;
    }
}

/* contingency.jl:42-56 dense 3-way + misc.jl:162-184 level_map! */
static int ctab3_dense(fwo_ctx *c, int X, int Y, const int *Zs, int k)
{
    ctab_reset(c, c->nstrata_cap);
    for (int64_t t = 0; t < c->zmap_len; ++t) c->zmap[t] = -1;
    int32_t levels_z = 0;
    for (int i = 0; i < c->n; ++i) {
        if (c->rowmask && !c->rowmask[i]) continue;
        int64_t gfp = 0; /* reference is 1-based: gfp_map = 1 + sum */
        for (int j = 0; j < k; ++j) gfp += (int64_t)dense_at(c, i, Zs[j]) * c->cum_levels[j];
        int32_t lv = c->zmap[gfp];
        if (lv != -1) {
            c->zrow[i] = lv;
        } else {
            c->zmap[gfp] = levels_z;
            c->zrow[i] = levels_z;
            ++levels_z;
        }
    }
    for (int i = 0; i < c->n; ++i) {
        if (c->rowmask && !c->rowmask[i]) continue;
        CT(c, dense_at(c, i, X), dense_at(c, i, Y), c->zrow[i]) += 1;
    }
    return levels_z;
}

/* contingency.jl:80-106 contingency_table_2d_optim! (both X and Y zero-adjusted) */
static void ctab2_sparse_optim(fwo_ctx *c, int X, int Y)
{
    ctab_reset(c, 1);
    int64_t px = c->colptr[X], py = c->colptr[Y];
    const int64_t ex = c->colptr[X + 1], ey = c->colptr[Y + 1];
    while (px < ex && py < ey) {
        int32_t rx = c->rowval[px], ry = c->rowval[py];
        if (rx == ry) {
            CT(c, c->nzval[px], c->nzval[py], 0) += 1;
            ++px;
            ++py;
        } else if (rx < ry) {
            ++px;
        } else {
            ++py;
        }
    }
}

/* contingency.jl:300-480 sparse_ctab_backend! for N = 2 + k columns.
 * Row indices are handled 1-based exactly like the reference (sentinel n_rows + 1,
 * initial min_ind = n_rows).  Returns zmap.levels_total for the 3-D case. */
static int ctab_sparse_backend(fwo_ctx *c, const int *cols, int N, int X_nz, int Y_nz)
{
    const int nzmode = c->nz; /* T <: Nz is a property of the test type */
    const int64_t n_rows = c->n;
    const int three_d = N > 2;
    int64_t ptr[2 + 16], bound[2 + 16], rowind[2 + 16];
    int32_t val[2 + 16];
    int64_t n_oob = 0, min_ind = n_rows;
    int break_loop = 0;
    int32_t levels_total = 0;

    if (three_d) {
        ctab_reset(c, c->nstrata_cap); /* types.jl:119-122 reset! */
        for (int64_t t = 0; t < c->zmap_len; ++t) c->zmap[t] = -1;
    } else {
        ctab_reset(c, 1);
    }
    /* contingency.jl:323-350 init */
    for (int i = 0; i < N; ++i) {
        ptr[i] = c->colptr[cols[i]];
        bound[i] = c->colptr[cols[i] + 1];
        val[i] = 0;
        if (ptr[i] < bound[i]) {
            rowind[i] = (int64_t)c->rowval[ptr[i]] + 1;
            if (rowind[i] < min_ind) min_ind = rowind[i];
        } else {
            if (nzmode && i < 2 && (i == 0 ? X_nz : Y_nz)) break_loop = 1; /* :286-298 */
            rowind[i] = n_rows + 1;
            ++n_oob;
        }
    }
    /* contingency.jl:444-457 main loop */
    for (;;) {
        int skip_row = 0;
        int64_t next_min = n_rows;
        for (int i = 0; i < N; ++i) {
            if (nzmode && i >= 2 && skip_row) {
                /* :394-411 fast-forward Z pointers on skipped rows */
                while (rowind[i] < next_min) {
                    ++ptr[i];
                    if (ptr[i] >= bound[i]) {
                        ++n_oob;
                        rowind[i] = n_rows + 1;
                    } else {
                        rowind[i] = (int64_t)c->rowval[ptr[i]] + 1;
                    }
                }
                continue;
            }
            if (rowind[i] == min_ind) { /* :373-383 */
                val[i] = c->nzval[ptr[i]];
                ++ptr[i];
                if (ptr[i] >= bound[i]) {
                    if (nzmode && i < 2 && (i == 0 ? X_nz : Y_nz)) break_loop = 1;
                    ++n_oob;
                    rowind[i] = n_rows + 1;
                } else {
                    rowind[i] = (int64_t)c->rowval[ptr[i]] + 1;
                }
            } else { /* :384-387 */
                val[i] = 0;
                if (nzmode && i < 2 && (i == 0 ? X_nz : Y_nz)) skip_row = 1;
            }
            if (rowind[i] < next_min) next_min = rowind[i]; /* :389-391 */
        }
        if (!(nzmode && skip_row)) {
            if (!three_d) {
                CT(c, val[0], val[1], 0) += 1; /* :417-419 */
            } else {
                /* :262-284 make_zmap_expression (0-based key here) */
                int64_t gfp = 0;
                for (int i = 2; i < N; ++i) gfp += (int64_t)val[i] * c->cum_levels[i - 2];
                int32_t z = c->zmap[gfp];
                if (z == -1) {
                    z = levels_total;
                    c->zmap[gfp] = z;
                    ++levels_total;
                }
                CT(c, val[0], val[1], z) += 1;
            }
        }
        if (nzmode && break_loop) break; /* :438-440 */
        if (n_oob >= N) break;           /* :451-453 */
        min_ind = next_min;
    }
    /* :461-477 rows never visited go to cell (0, 0[, all-zero stratum]) */
    if (!three_d) {
        int64_t s = 0;
        for (int64_t t = 0; t < (int64_t)c->L * c->L; ++t) s += c->ctab[t];
        CT(c, 0, 0, 0) += n_rows - s;
        return 0;
    }
    int64_t s = 0;
    for (int64_t t = 0; t < (int64_t)c->L * c->L * (levels_total > 0 ? levels_total : 1); ++t) s += c->ctab[t];
    int64_t all_zero_obs = n_rows - s;
    if (all_zero_obs > 0) {
        int32_t zi;
        if (c->zmap[0] != -1) {
            zi = c->zmap[0];
        } else {
            zi = levels_total;
            ++levels_total;
        }
        CT(c, 0, 0, zi) += all_zero_obs;
    }
    return levels_total;
}

/* contingency.jl:182-237 k = 1 HE special case (+ helper expressions :128-179).
 * X_nz / Y_nz: whether X / Y are zero-adjusted (Nz type parameters). */
static int ctab3_sparse_k1(fwo_ctx *c, int X, int Y, int Z, int X_nz, int Y_nz)
{
    ctab_reset(c, c->nstrata_cap);
    int32_t levels_z = 1; /* 1-based max stratum seen */
    int64_t px = c->colptr[X], py = c->colptr[Y], pz = c->colptr[Z];
    const int64_t ex = c->colptr[X + 1], ey = c->colptr[Y + 1], ez = c->colptr[Z + 1];
    int64_t row_Z = pz < ez ? (int64_t)c->rowval[pz] : (int64_t)c->n; /* 0-based analogue of n+1 */
#define ZUPD(row)                                                                    \
    do {                                                                             \
        while (pz < ez - 1 && row_Z < (row)) {                                       \
            ++pz;                                                                    \
            row_Z = c->rowval[pz];                                                   \
        }                                                                            \
        if (row_Z == (row)) {                                                        \
            val_Z = c->nzval[pz] + 1;                                                \
            if (val_Z > levels_z) levels_z = val_Z;                                  \
        } else {                                                                     \
            val_Z = 1;                                                               \
        }                                                                            \
    } while (0)
    int32_t val_Z = 1;
    while (px < ex && py < ey) {
        int64_t rx = c->rowval[px], ry = c->rowval[py];
        if (rx == ry) {
            ZUPD(rx);
            CT(c, c->nzval[px], c->nzval[py], val_Z - 1) += 1;
            ++px;
            ++py;
        } else if (rx < ry) {
            if (!Y_nz) { /* X_zeroupd_expr exists iff Y is NoNz (:199) */
                ZUPD(rx);
                CT(c, c->nzval[px], 0, val_Z - 1) += 1;
            }
            ++px;
        } else {
            if (!X_nz) { /* Y_zeroupd_expr exists iff X is NoNz (:200) */
                ZUPD(ry);
                CT(c, 0, c->nzval[py], val_Z - 1) += 1;
            }
            ++py;
        }
    }
    if (!Y_nz) { /* :223 X_zerofinish */
        while (px < ex) {
            int64_t rx = c->rowval[px];
            ZUPD(rx);
            CT(c, c->nzval[px], 0, val_Z - 1) += 1;
            ++px;
        }
    }
    if (!X_nz) { /* :224 Y_zerofinish */
        while (py < ey) {
            int64_t ry = c->rowval[py];
            ZUPD(ry);
            CT(c, 0, c->nzval[py], val_Z - 1) += 1;
            ++py;
        }
    }
#undef ZUPD
    return levels_z;
}

/* contingency.jl:109-123 sparse 2-way dispatch */
static void ctab2_sparse(fwo_ctx *c, int X, int Y)
{
    int X_nz = 0, Y_nz = 0;
    if (c->nz) {
        X_nz = c->max_vals[X] > 1;
        Y_nz = c->max_vals[Y] > 1;
    }
    if (X_nz && Y_nz) {
        ctab2_sparse_optim(c, X, Y);
    } else {
        int cols[2] = {X, Y};
        ctab_sparse_backend(c, cols, 2, X_nz, Y_nz);
    }
}

/* contingency.jl:240-258 sparse 3-way dispatch */
static int ctab3_sparse(fwo_ctx *c, int X, int Y, const int *Zs, int k)
{
    int X_nz = 0, Y_nz = 0;
    if (c->nz) {
        X_nz = c->max_vals[X] > 1;
        Y_nz = c->max_vals[Y] > 1;
    }
    if (k == 1 && (X_nz || Y_nz)) return ctab3_sparse_k1(c, X, Y, Zs[0], X_nz, Y_nz);
    int cols[2 + 16];
    cols[0] = X;
    cols[1] = Y;
    for (int j = 0; j < k; ++j) cols[2 + j] = Zs[j];
    return ctab_sparse_backend(c, cols, 2 + k, X_nz, Y_nz);
}

/* Expose tables for the known-answer tests of test/contingency.jl:55-69.
 * out must hold L*L*nslots int64 (column-major [x][y][z]). Returns levels_z (0 for k = 0). */
int fwo_contingency_table(fwo_ctx *c, int X, int Y, const int *Zs, int k, int64_t *out, int64_t nslots)
{
    int lz = 0;
    if (k == 0) {
        if (c->sparse)
            ctab2_sparse(c, X, Y);
        else
            ctab2_dense(c, X, Y);
        memcpy(out, c->ctab, sizeof(int64_t) * (size_t)(c->L * c->L));
        return 0;
    }
    lz = c->sparse ? ctab3_sparse(c, X, Y, Zs, k) : ctab3_dense(c, X, Y, Zs, k);
    int64_t ncopy = nslots < c->nstrata_cap ? nslots : c->nstrata_cap;
    memcpy(out, c->ctab, sizeof(int64_t) * (size_t)(c->L * c->L * ncopy));
    return lz;
}

/* ------------------------------------------------------------------------------------------
 * Mutual information, df  (operate on the nz-adjusted sub-view: rows >= sx, cols >= sy)
 * ---------------------------------------------------------------------------------------- */

#define SUB(c, i, j, k) CT(c, (i) + sx, (j) + sy, k)

/* statfuns.jl:209-254 */
static double mi_2d(fwo_ctx *c, int sx, int sy, int levels_x, int levels_y, int64_t *marg_i, int64_t *marg_j)
{
    const int dimx = c->L - sx, dimy = c->L - sy;
    for (int i = 0; i < c->L; ++i) marg_i[i] = marg_j[i] = 0;
    for (int i = 0; i < levels_x; ++i)
        for (int j = 0; j < levels_y; ++j) {
            marg_i[i] += SUB(c, i, j, 0);
            marg_j[j] += SUB(c, i, j, 0);
        }
    int64_t n_obs = 0;
    for (int j = 0; j < dimy; ++j)
        for (int i = 0; i < dimx; ++i) n_obs += SUB(c, i, j, 0);
    double pos = 0.0, neg = 0.0;
    int64_t npos = 0, nneg = 0;
    for (int i = 0; i < levels_x; ++i) {
        int64_t mi_ = marg_i[i];
        for (int j = 0; j < levels_y; ++j) {
            int64_t cell = SUB(c, i, j, 0), mj = marg_j[j];
            if (cell != 0 && mi_ != 0 && mj != 0) {
                double cell_mi = (double)cell * log((double)(n_obs * cell) / (double)(mi_ * mj));
                if (i == j) {
                    pos += cell_mi;
                    npos += cell;
                } else {
                    neg += cell_mi;
                    nneg += cell;
                }
            }
        }
    }
    double mi = (pos + neg) / (double)n_obs;
    if (neg * ((double)nneg / (double)n_obs) > pos * ((double)npos / (double)n_obs)) mi *= -1.0;
    return mi;
}

/* statfuns.jl:163-207 */
static double mi_3d(fwo_ctx *c, int sx, int sy, int levels_x, int levels_y, int levels_z)
{
    const int dimx = c->L - sx, dimy = c->L - sy;
    const int64_t cap = c->nstrata_cap;
    int64_t *marg_i = c->marg_i, *marg_j = c->marg_j, *marg_k = c->marg_k;
    memset(marg_i, 0, sizeof(int64_t) * (size_t)(c->L * cap));
    memset(marg_j, 0, sizeof(int64_t) * (size_t)(c->L * cap));
    memset(marg_k, 0, sizeof(int64_t) * (size_t)cap);
    for (int i = 0; i < levels_x; ++i)
        for (int j = 0; j < levels_y; ++j)
            for (int k = 0; k < levels_z; ++k) {
                int64_t v = SUB(c, i, j, k);
                marg_i[i + (int64_t)c->L * k] += v;
                marg_j[j + (int64_t)c->L * k] += v;
                marg_k[k] += v;
            }
    double pos = 0.0, neg = 0.0;
    int64_t npos = 0, nneg = 0;
    /* reference loops k over the whole third dimension; strata >= levels_z have zero marginals */
    for (int i = 0; i < dimx; ++i)
        for (int j = 0; j < dimy; ++j)
            for (int k = 0; k < levels_z; ++k) {
                int64_t cell = SUB(c, i, j, k);
                int64_t mik = marg_i[i + (int64_t)c->L * k], mjk = marg_j[j + (int64_t)c->L * k];
                if (cell != 0 && mik != 0 && mjk != 0) {
                    double inner = log((double)(marg_k[k] * cell) / (double)(mik * mjk)) * (double)cell;
                    if (i == j) {
                        pos += inner;
                        npos += cell;
                    } else {
                        neg += inner;
                        nneg += cell;
                    }
                }
            }
    int64_t n_obs = npos + nneg;
    double mi = (pos + neg) / (double)n_obs;
    if (neg * ((double)nneg / (double)n_obs) > pos * ((double)npos / (double)n_obs)) mi *= -1.0;
    return mi;
}

static inline int isign(int64_t v) { return (v > 0) - (v < 0); }

/* statfuns.jl:281-297 */
static int64_t adjust_df_vec(const int64_t *marg_i, const int64_t *marg_j, int levels_x, int levels_y)
{
    int64_t alx = 0, aly = 0;
    for (int i = 0; i < levels_x; ++i) alx += isign(marg_i[i]);
    for (int j = 0; j < levels_y; ++j) aly += isign(marg_j[j]);
    if (alx < 1) alx = 1;
    if (aly < 1) aly = 1;
    return (alx - 1) * (aly - 1);
}

/* Convenience forms used by test/statfuns.jl:44-55: MI of a free-standing table */
double fwo_mutual_information(const int64_t *ctab, int lx, int ly, int lz)
{
    fwo_ctx c;
    memset(&c, 0, sizeof(c));
    c.L = lx > ly ? lx : ly;
    c.nstrata_cap = lz > 0 ? lz : 1;
    int64_t L = c.L;
    c.ctab = (int64_t *)calloc((size_t)(L * L * c.nstrata_cap), sizeof(int64_t));
    c.marg_i = (int64_t *)calloc((size_t)(L * c.nstrata_cap), sizeof(int64_t));
    c.marg_j = (int64_t *)calloc((size_t)(L * c.nstrata_cap), sizeof(int64_t));
    c.marg_k = (int64_t *)calloc((size_t)c.nstrata_cap, sizeof(int64_t));
    for (int k = 0; k < (lz > 0 ? lz : 1); ++k)
        for (int j = 0; j < ly; ++j)
            for (int i = 0; i < lx; ++i) CT(&c, i, j, k) = ctab[i + (int64_t)lx * (j + (int64_t)ly * k)];
    double r = lz > 0 ? mi_3d(&c, 0, 0, lx, ly, lz) : mi_2d(&c, 0, 0, lx, ly, c.marg_i, c.marg_j);
    free(c.ctab);
    free(c.marg_i);
    free(c.marg_j);
    free(c.marg_k);
    return r;
}

/* ------------------------------------------------------------------------------------------
 * Single tests
 * ---------------------------------------------------------------------------------------- */

static inline void set_result(fwo_result *r, double stat, double pval, int64_t df, int pw)
{
    r->stat = stat;
    r->pval = pval;
    r->df = df;
    r->suff_power = pw;
    r->pad = 0;
}

/* tests.jl:5-6 */
static inline int suff_power3(int64_t lx, int64_t ly, int64_t n_obs, int hps)
{
    return ((double)n_obs / (double)(lx * ly)) > (double)hps;
}
static inline int suff_power4(int64_t lx, int64_t ly, int64_t lz, int64_t n_obs, int hps)
{
    return ((double)n_obs / (double)(lx * ly * lz)) > (double)hps;
}

/* tests.jl:9-20 data form (the discrete check is applied to levels - offset_levels(levels)) */
static int suff_power_data(const fwo_ctx *c, int X, int Y, int64_t n_rows, int64_t n_obs_min, int hps)
{
    if (n_rows < n_obs_min) return 0;
    if (!FWO_IS_CONT(c)) {
        int64_t lx = c->levels[X], ly = c->levels[Y];
        int64_t ox = lx > 1 ? 2 : 1, oy = ly > 1 ? 2 : 1; /* statfuns.jl:307-311 called with levels */
        if (!suff_power3(lx - ox, ly - oy, n_rows, hps)) return 0;
    }
    return 1;
}

/* tests.jl:28-77 univariate discrete test */
static void disc_test_uni(fwo_ctx *c, int X, int Y, int hps, int64_t n_obs_min, fwo_result *out)
{
    int levels_x = c->levels[X], levels_y = c->levels[Y];
    if (!suff_power_data(c, X, Y, c->n, n_obs_min, hps)) {
        set_result(out, 0.0, 1.0, 0, 0);
        return;
    }
    if (!c->sparse)
        ctab2_dense(c, X, Y);
    else
        ctab2_sparse(c, X, Y);
    int sx = 0, sy = 0;
    if (c->nz) { /* statfuns.jl:313-323 nz_adjust_cont_tab; tests.jl:48-51 */
        sx = c->max_vals[X] > 1 ? 1 : 0;
        sy = c->max_vals[Y] > 1 ? 1 : 0;
        levels_x = c->L - sx;
        levels_y = c->L - sy;
    }
    int64_t n_obs = 0;
    for (int j = sy; j < c->L; ++j)
        for (int i = sx; i < c->L; ++i) n_obs += CT(c, i, j, 0);
    if (n_obs < n_obs_min || !suff_power3(levels_x, levels_y, n_obs, hps)) {
        set_result(out, 0.0, 1.0, 0, 0);
        return;
    }
    double mi = mi_2d(c, sx, sy, levels_x, levels_y, c->marg_i, c->marg_j);
    int64_t df = adjust_df_vec(c->marg_i, c->marg_j, levels_x, levels_y);
    double pval = fwo_mi_pval(fabs(mi), df, n_obs);
    set_result(out, mi, pval, df, 1);
}

/* tests.jl:184-229 conditional discrete test */
static void disc_test_cond(fwo_ctx *c, int X, int Y, const int *Zs, int k, int hps, fwo_result *out)
{
    int levels_x = c->levels[X], levels_y = c->levels[Y];
    int levels_z = c->sparse ? ctab3_sparse(c, X, Y, Zs, k) : ctab3_dense(c, X, Y, Zs, k);
    int sx = 0, sy = 0;
    if (c->nz) {
        sx = c->max_vals[X] > 1 ? 1 : 0;
        sy = c->max_vals[Y] > 1 ? 1 : 0;
        levels_x = c->L - sx;
        levels_y = c->L - sy;
    }
    int64_t n_obs = 0;
    for (int kk = 0; kk < levels_z; ++kk)
        for (int j = sy; j < c->L; ++j)
            for (int i = sx; i < c->L; ++i) n_obs += CT(c, i, j, kk);
    if (!suff_power4(levels_x, levels_y, levels_z, n_obs, hps)) {
        set_result(out, 0.0, 1.0, 0, 0);
        return;
    }
    double mi = mi_3d(c, sx, sy, levels_x, levels_y, levels_z);
    int64_t df = 0; /* statfuns.jl:299-305 */
    for (int kk = 0; kk < levels_z; ++kk)
        df += adjust_df_vec(c->marg_i + (int64_t)c->L * kk, c->marg_j + (int64_t)c->L * kk, levels_x, levels_y);
    double pval = fwo_mi_pval(fabs(mi), df, n_obs);
    set_result(out, mi, pval, df, 1);
}

/* --- continuous ---------------------------------------------------------------------------- */

/* A Julia value that is either Float32 or Float64 at run time (pcor_rec is type-unstable for
 * Float32 matrices: literal 0.0 / -1.0 / 1.0 are Float64, statfuns.jl:41,53,58-62). */
typedef struct {
    double v;
    int is32;
} tval;

static inline tval tv32(float x)
{
    tval t = {(double)x, 1};
    return t;
}
static inline tval tv64(double x)
{
    tval t = {x, 0};
    return t;
}

/* round(x, digits=5): Base._round_digits -> round(x * 10^5) / 10^5 in the value's own type,
 * RoundNearest (ties to even); returns x if the result is not finite. */
static inline float round5_f32(float x)
{
    float y = rintf(x * 100000.0f) / 100000.0f;
    return isfinite(y) ? y : x;
}
static inline double round5_f64(double x)
{
    double y = rint(x * 100000.0) / 100000.0;
    return isfinite(y) ? y : x;
}

static inline tval cor_at(const fwo_ctx *c, int a, int b)
{
    if (c->cor32) return tv32(c->cor32[(int64_t)b * c->p + a]);
    return tv64(c->cor64[(int64_t)b * c->p + a]);
}

static inline tval clamp_p(tval p)
{ /* statfuns.jl:58-62 */
    if (p.v < -1.0) return tv64(-1.0);
    if (p.v >= 1.0) return tv64(1.0);
    return p;
}

/* statfuns.jl:23-75 pcor_rec (cache_result = false path; caching does not change values) */
static tval pcor_rec(const fwo_ctx *c, int X, int Y, const int *Zs, int k)
{
    const int one32 = c->cor32 != NULL; /* one(ContType) is Float32 */
    tval p;
    if (k == 1) {
        tval pXY = cor_at(c, X, Y), pXZ = cor_at(c, X, Zs[0]), pYZ = cor_at(c, Y, Zs[0]);
        if (one32) {
            float xy = (float)pXY.v, xz = (float)pXZ.v, yz = (float)pYZ.v;
            float prod = xz * yz;
            float e = xy - prod;
            e = round5_f32(e);
            float s1 = 1.0f - xz * xz, s2 = 1.0f - yz * yz;
            float d = sqrtf(s1) * sqrtf(s2);
            p = (d == 0.0f) ? tv64(0.0) : tv32(e / d);
        } else {
            double xy = pXY.v, xz = pXZ.v, yz = pYZ.v;
            double prod = xz * yz;
            double e = round5_f64(xy - prod);
            double d = sqrt(1.0 - xz * xz) * sqrt(1.0 - yz * yz);
            p = (d == 0.0) ? tv64(0.0) : tv64(e / d);
        }
    } else {
        int Z0 = Zs[k - 1];
        tval a = pcor_rec(c, X, Y, Zs, k - 1);
        tval b = pcor_rec(c, X, Z0, Zs, k - 1);
        tval cc = pcor_rec(c, Y, Z0, Zs, k - 1);
        /* enum_term = a - b * c with Julia promotion */
        tval prod;
        if (b.is32 && cc.is32) {
            float t = (float)b.v * (float)cc.v;
            prod = tv32(t);
        } else {
            double t = b.v * cc.v;
            prod = tv64(t);
        }
        tval e;
        if (a.is32 && prod.is32) {
            float t = (float)a.v - (float)prod.v;
            e = tv32(round5_f32(t));
        } else {
            double t = a.v - prod.v;
            e = tv64(round5_f64(t));
        }
        /* sqrt(one(ContType) - b^2): b^2 = b*b in b's type */
        tval d1;
        if (b.is32 && one32) {
            float bb = (float)b.v * (float)b.v;
            float s = 1.0f - bb;
            d1 = tv32(sqrtf(s));
        } else if (b.is32) {
            float bb = (float)b.v * (float)b.v;
            double s = 1.0 - (double)bb;
            d1 = tv64(sqrt(s));
        } else {
            double bb = b.v * b.v;
            double s = 1.0 - bb;
            d1 = tv64(sqrt(s));
        }
        /* sqrt(one(ContType) - c^2.0): c^2.0 is Float64 */
        double c2 = cc.v * cc.v;
        double d2 = sqrt(1.0 - c2);
        double denom = d1.v * d2;
        p = (denom == 0.0) ? tv64(0.0) : tv64(e.v / denom);
    }
    return clamp_p(p);
}

double fwo_pcor_rec(const fwo_ctx *c, int X, int Y, const int *Zs, int k) { return pcor_rec(c, X, Y, Zs, k).v; }

/* tests.jl:108-160, branch with a precomputed cor_mat (:149-153) */
static void fz_test_uni(const fwo_ctx *c, int X, int Y, int64_t n_obs_min, fwo_result *out)
{
    if (c->n_obs < n_obs_min) { /* sufficient_power, tests.jl:11; n_obs = 0 -> power = (0 >= n_obs_min) */
        set_result(out, 0.0, 1.0, 0, 0 >= n_obs_min);
        return;
    }
    int64_t n_obs = c->n_obs;
    double p_stat = cor_at(c, X, Y).v;
    double pval = fwo_fz_pval(p_stat, n_obs, 0);
    set_result(out, p_stat, pval, 0, n_obs >= n_obs_min);
}

/* statfuns.jl:19-21 pcor(X, Y, Zs, data) = StatsBase.partialcor(data[:, X], data[:, Y], data[:, collect(Zs)]) -- the path
 * of tests.jl:253 when the test object holds no cor_mat (recursive_pcor = false / dense_cor = false, learning.jl:127,211).
 * StatsBase is a registered dependency (Project.toml compat "0.32, 0.33"), not vendored in the reference; its published
 * algorithm (src/partialcor.jl of those releases), restated:
 *   _partialcor(x, y, z::Vector): ONE pass of centred sums Sxx, Syy, Szz, Sxy, Sxz, Szy (means first), pairwise
 *       r = S_ab / sqrt(S_aa S_bb), result (rxy - rxz rzy) / (sqrt(1 - rxz^2) sqrt(1 - rzy^2));
 *   _partialcor(x, y, Z::Matrix): z0 = first column, Zrest = the others;
 *       (r(x,y|Zrest) - r(x,z0|Zrest) r(z0,y|Zrest)) / (sqrt(1 - r(x,z0|Zrest)^2) sqrt(1 - r(z0,y|Zrest)^2));
 *   partialcor = clampcor(_partialcor(...)) (clamped to [-1, 1]).
 * No 5-digit rounding here (that is pcor_rec's).  The recursion unrolls into: condition the correlation matrix of
 * {X, Y, Z_1..Z_k} on Z_k, then Z_{k-1}, ..., finally Z_1.  Sums in Float64 (prec = 64 semantics; with Float32 data the
 * reference accumulates in Float32 under @simd in an unknowable order: tolerance, like `cor`).
 * Pinned on test/statfuns.jl:24-37 (exp_pcor_Z1 / exp_pcor_Z3, rtol 1e-6). */
double fwo_pcor(const fwo_ctx *c, int X, int Y, const int *Zs, int k)
{
    enum { MAXM = 18 };
    const int m = k + 2;
    if (!c->fdata || m > MAXM) return NAN;
    int v[MAXM];
    double mu[MAXM], R[MAXM][MAXM];
    v[0] = X;
    v[1] = Y;
    for (int j = 0; j < k; ++j) v[2 + j] = Zs[j];
    for (int a = 0; a < m; ++a) {
        double s = 0.0;
        for (int i = 0; i < c->n; ++i) s += c->fdata[(int64_t)v[a] * c->n + i];
        mu[a] = s / (double)c->n;
    }
    for (int a = 0; a < m; ++a)
        for (int b = a; b < m; ++b) {
            double s = 0.0;
            for (int i = 0; i < c->n; ++i)
                s += (c->fdata[(int64_t)v[a] * c->n + i] - mu[a]) * (c->fdata[(int64_t)v[b] * c->n + i] - mu[b]);
            R[a][b] = R[b][a] = s;
        }
    double sd[MAXM];
    for (int a = 0; a < m; ++a) sd[a] = R[a][a];
    for (int a = 0; a < m; ++a)
        for (int b = 0; b < m; ++b) R[a][b] = R[a][b] / sqrt(sd[a] * sd[b]);
    for (int t = m - 1; t >= 2; --t)
        for (int a = 0; a < t; ++a)
            for (int b = a + 1; b < t; ++b) {
                const double r = (R[a][b] - R[a][t] * R[b][t]) / (sqrt(1.0 - R[a][t] * R[a][t]) * sqrt(1.0 - R[b][t] * R[b][t]));
                R[a][b] = R[b][a] = r;
            }
    double r = R[0][1];
    if (r < -1.0) r = -1.0; /* Statistics.clampcor */
    if (r > 1.0) r = 1.0;
    return r;
}

/* attach the normalised data (n x p column-major, widened to double) to a "fz" context; stream != 0: conditional tests use
 * pcor instead of pcor_rec (the reference's FzTestCond with an empty cor_mat) */
void fwo_fz_set_data(fwo_ctx *c, const double *data, int stream)
{
    c->fdata = data;
    c->fz_stream = stream;
}

/* tests.jl:250-265 */
static void fz_test_cond(const fwo_ctx *c, int X, int Y, const int *Zs, int k, int64_t n_obs_min, fwo_result *out)
{
    if (c->n_obs >= n_obs_min) {
        double p_stat = c->fz_stream ? fwo_pcor(c, X, Y, Zs, k) : pcor_rec(c, X, Y, Zs, k).v;
        double pval = fwo_fz_pval(p_stat, c->n_obs, 0); /* len_z hard-wired to 0, tests.jl:256 */
        set_result(out, p_stat, pval, 0, 1);
    } else {
        set_result(out, 0.0, 1.0, 0, 0);
    }
}

/* ---- HE-S ("fz_nz") ------------------------------------------------------------------------ */

#define FD(c, i, v) ((c)->fdata[(int64_t)(v) * (c)->n + (i)])

/* statfuns.jl:91-123 cor(X, Y, data::SparseMatrixCSC, nz = true): two passes over the rows where both X and Y are
 * non-zero (misc.jl:275-364 iter_apply_sparse_rows! with x_nzadj = y_nzadj = true), Float64 accumulators. */
static double fz_nz_pair_cor(const fwo_ctx *c, int X, int Y, int64_t *n_obs_out)
{
    double sum_x = 0.0, sum_y = 0.0;
    int64_t nn = 0;
    for (int i = 0; i < c->n; ++i) {
        const double x = FD(c, i, X), y = FD(c, i, Y);
        if (x != 0.0 && y != 0.0) {
            sum_x += x;
            sum_y += y;
            ++nn;
        }
    }
    *n_obs_out = nn;
    if (nn == 0) return 0.0;
    const double mean_x = sum_x / (double)nn, mean_y = sum_y / (double)nn;
    double cov = 0.0, vx = 0.0, vy = 0.0;
    for (int i = 0; i < c->n; ++i) {
        const double x = FD(c, i, X), y = FD(c, i, Y);
        if (x != 0.0 && y != 0.0) {
            const double dx = x - mean_x, dy = y - mean_y;
            cov += dx * dy;
            vx += dx * dx;
            vy += dy * dy;
        }
    }
    double p = cov / sqrt(vx * vy);
    if (p > 1.0)
        p = 1.0;
    else if (p < -1.0)
        p = -1.0;
    return p;
}

/* tests.jl:108-160 with a sparse matrix and an empty cor_mat (branch :120-125) */
static void fz_nz_test_uni(const fwo_ctx *c, int X, int Y, int64_t n_obs_min, fwo_result *out)
{
    if (c->n < n_obs_min) { /* sufficient_power(X, Y, data, ...) on the full row count, tests.jl:11 */
        set_result(out, 0.0, 1.0, 0, 0 >= n_obs_min);
        return;
    }
    int64_t n_obs = 0;
    double p_stat = fz_nz_pair_cor(c, X, Y, &n_obs);
    if (n_obs < n_obs_min) p_stat = 0.0;
    set_result(out, p_stat, fwo_fz_pval(p_stat, n_obs, 0), 0, n_obs >= n_obs_min);
}

static double fz_nz_seq(const double *t, int64_t n)
{
    double s = 0.0;
    for (int64_t q = 0; q < n; ++q) s += t[q];
    return s;
}
/* "tree64": lane l adds the terms q = l, l + 64, ... in order; the 64 partials are combined as pairs (l, l^1), then (l, l^2),
 * then inside every group of 16 lanes (Q3 + Q2) + (Q1 + Q0), then (R3 + R2) + (R1 + R0) -- the order of the device's DPP
 * reduction (fznz_tree64 in csrc/fw_fz.hip) */
static double fz_nz_tree64(const double *t, int64_t n)
{
    double part[64], Q[16], R[4];
    for (int l = 0; l < 64; ++l) {
        double s = 0.0;
        for (int64_t q = l; q < n; q += 64) s += t[q];
        part[l] = s;
    }
    for (int j = 0; j < 16; ++j) Q[j] = (part[4 * j] + part[4 * j + 1]) + (part[4 * j + 2] + part[4 * j + 3]);
    for (int r = 0; r < 4; ++r) R[r] = (Q[4 * r + 3] + Q[4 * r + 2]) + (Q[4 * r + 1] + Q[4 * r]);
    return (R[3] + R[2]) + (R[1] + R[0]);
}

/* statfuns.jl:138-155 cor_subset!: Statistics.cor of the rows R (both X and Y non-zero, hiton.jl:41-50,85) restricted
 * to vars; NaN -> 0; stored in a Float32 matrix (learning.jl:127-129, cont_type = Float32).  local: m x m floats. */
static int64_t fz_nz_cor_subset(const fwo_ctx *c, int X, int Y, const int *vars, int m, float *local)
{
    int *rows = (int *)malloc(sizeof(int) * (size_t)(c->n > 0 ? c->n : 1));
    int64_t nR = 0;
    for (int i = 0; i < c->n; ++i)
        if (FD(c, i, X) != 0.0 && FD(c, i, Y) != 0.0) rows[nR++] = i;
    if (local) {
        const int64_t nr = nR > 0 ? nR : 1;
        if (c->fdata_f32) { /* Statistics.cor in Float32: mean, centring, x'x and cov2cor! all in Float32 */
            float *xc = (float *)malloc(sizeof(float) * (size_t)(nr * m));
            float *sd = (float *)malloc(sizeof(float) * (size_t)m);
            for (int a = 0; a < m; ++a) {
                float s = 0.0f;
                for (int64_t q = 0; q < nR; ++q) s += (float)FD(c, rows[q], vars[a]);
                const float mean = s / (float)nR;
                float ss = 0.0f;
                for (int64_t q = 0; q < nR; ++q) {
                    const float d = (float)FD(c, rows[q], vars[a]) - mean;
                    xc[(int64_t)a * nr + q] = d;
                    ss += d * d;
                }
                sd[a] = sqrtf(ss);
            }
            for (int a = 0; a < m; ++a)
                for (int b = a + 1; b < m; ++b) {
                    float s = 0.0f;
                    for (int64_t q = 0; q < nR; ++q) s += xc[(int64_t)a * nr + q] * xc[(int64_t)b * nr + q];
                    float r = s / (sd[a] * sd[b]);
                    if (r > 1.0f) r = 1.0f;
                    if (r < -1.0f) r = -1.0f;
                    if (isnan(r)) r = 0.0f;
                    local[(int64_t)b * m + a] = local[(int64_t)a * m + b] = r;
                }
            free(xc);
            free(sd);
        } else {
            /* Float64 sums in the device's order (csrc/fw_fz.hip fznz_submat_kernel, "tree64"): 64 interleaved partials over
             * the rows of the view in ascending order, combined in a fixed tree.  Univariate jobs (m == 2: the pair statistic
             * must equal level 0's sequential fz_nz_pair_cor) and views beyond the device's LDS row list keep the sequential
             * order.  The reference's own order (Statistics.cor on a view, BLAS) is not knowable: a tolerance either way. */
            const int tree = m > 2 && c->n <= 16384;
            double *xc = (double *)malloc(sizeof(double) * (size_t)(nr * m));
            double *sd = (double *)malloc(sizeof(double) * (size_t)m);
            double *term = (double *)malloc(sizeof(double) * (size_t)nr);
            for (int a = 0; a < m; ++a) {
                for (int64_t q = 0; q < nR; ++q) term[q] = FD(c, rows[q], vars[a]);
                const double mean = (tree ? fz_nz_tree64(term, nR) : fz_nz_seq(term, nR)) / (double)nR;
                for (int64_t q = 0; q < nR; ++q) {
                    const double d = FD(c, rows[q], vars[a]) - mean;
                    xc[(int64_t)a * nr + q] = d;
                    term[q] = d * d;
                }
                sd[a] = sqrt(tree ? fz_nz_tree64(term, nR) : fz_nz_seq(term, nR));
            }
            for (int a = 0; a < m; ++a)
                for (int b = a + 1; b < m; ++b) {
                    for (int64_t q = 0; q < nR; ++q) term[q] = xc[(int64_t)a * nr + q] * xc[(int64_t)b * nr + q];
                    const double s = tree ? fz_nz_tree64(term, nR) : fz_nz_seq(term, nR);
                    double r = s / (sd[a] * sd[b]);
                    if (r > 1.0) r = 1.0;
                    if (r < -1.0) r = -1.0;
                    if (isnan(r)) r = 0.0;
                    local[(int64_t)b * m + a] = local[(int64_t)a * m + b] = (float)r;
                }
            free(xc);
            free(sd);
            free(term);
        }
        for (int a = 0; a < m; ++a) local[(int64_t)a * m + a] = 1.0f;
    }
    free(rows);
    return nR;
}

/* fz_nz with recursive_pcor = false (tests.jl:253 on the row view of hiton.jl:85: FzTestCond with an empty cor_mat): the conditional
 * tests are StatsBase.partialcor of the view's columns.  stream != 0 switches a FWO_FZ_NZ context to that form. */
void fwo_fz_nz_set_stream(fwo_ctx *c, int stream) { c->fz_stream = stream; }

/* the rows R (X != 0 and Y != 0) of the columns `vars` as an nR x m column-major Float64 matrix (caller frees) */
static double *fz_nz_view(const fwo_ctx *c, int X, int Y, const int *vars, int m, int64_t *nR_out)
{
    int64_t nR = 0;
    for (int i = 0; i < c->n; ++i)
        if (FD(c, i, X) != 0.0 && FD(c, i, Y) != 0.0) ++nR;
    double *v = (double *)malloc(sizeof(double) * (size_t)((nR > 0 ? nR : 1) * m));
    for (int a = 0; a < m; ++a) {
        int64_t q = 0;
        for (int i = 0; i < c->n; ++i)
            if (FD(c, i, X) != 0.0 && FD(c, i, Y) != 0.0) v[(int64_t)a * nR + q++] = FD(c, i, vars[a]);
    }
    *nR_out = nR;
    return v;
}

/* conditional fz_nz test of one explicit subset (tests.jl:250-265 on the row view of hiton.jl:85) */
static void fz_nz_test_cond(const fwo_ctx *c, int X, int Y, const int *Zs, int k, int64_t n_obs_min, fwo_result *out)
{
    int vars[2 + 16], loc[16];
    vars[0] = X;
    vars[1] = Y;
    for (int j = 0; j < k; ++j) {
        vars[2 + j] = Zs[j];
        loc[j] = 2 + j;
    }
    const int m = k + 2;
    float local[18 * 18];
    const int64_t nR = fz_nz_cor_subset(c, X, Y, vars, m, local);
    if (nR < n_obs_min) {
        set_result(out, 0.0, 1.0, 0, 0);
        return;
    }
    fwo_ctx tmp;
    memset(&tmp, 0, sizeof(tmp));
    tmp.kind = FWO_FZ;
    tmp.p = m;
    tmp.n = tmp.n_obs = (int)nR;
    tmp.cor32 = local;
    if (c->fz_stream) { /* no cor_mat: pcor on the view's columns */
        int64_t nv;
        double *view = fz_nz_view(c, X, Y, vars, m, &nv);
        tmp.fdata = view;
        const double ps = fwo_pcor(&tmp, 0, 1, loc, k);
        free(view);
        set_result(out, ps, fwo_fz_pval(ps, nR, 0), 0, 1);
        return;
    }
    const double p_stat = pcor_rec(&tmp, 0, 1, loc, k).v;
    set_result(out, p_stat, fwo_fz_pval(p_stat, nR, 0), 0, 1);
}

/* Public single-test entry: k = 0 univariate, k >= 1 conditional.
 * hps is used by discrete tests, n_obs_min by univariate discrete and all fz tests (Q2). */
void fwo_test(fwo_ctx *c, int X, int Y, const int *Zs, int k, int hps, int64_t n_obs_min, fwo_result *out)
{
    if (c->kind == FWO_FZ_NZ) {
        if (k == 0)
            fz_nz_test_uni(c, X, Y, n_obs_min, out);
        else
            fz_nz_test_cond(c, X, Y, Zs, k, n_obs_min, out);
    } else if (c->kind == FWO_FZ) {
        if (k == 0)
            fz_test_uni(c, X, Y, n_obs_min, out);
        else
            fz_test_cond(c, X, Y, Zs, k, n_obs_min, out);
    } else {
        if (k == 0)
            disc_test_uni(c, X, Y, hps, n_obs_min, out);
        else
            disc_test_cond(c, X, Y, Zs, k, hps, out);
    }
}

/* tests.jl:1-3 */
static double hiton_now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static __thread double fwo_tls_deadline = 0.0; /* set by fwo_learn for bench.py's bounded baseline sample (max_seconds) */

static inline int issig(const fwo_result *r, double alpha) { return r->pval < alpha && r->suff_power; }

/* ------------------------------------------------------------------------------------------
 * test_subsets  (tests.jl:281-346)
 * ---------------------------------------------------------------------------------------- */

static double binom_d(int n, int k)
{
    if (k < 0 || k > n) return 0.0;
    double r = 1.0;
    for (int i = 1; i <= k; ++i) r = r * (double)(n - k + i) / (double)i;
    return floor(r + 0.5);
}

/* status: 0 = sentinel (empty Z_total), 1 = returned at a non-significant / max_tests stop, 2 = all significant.
 * Zs_out: the conditioning set of the returned result (variable ids), nZs_out its length. */
int fwo_test_subsets(fwo_ctx *c, int X, int Y, const int *Z_total, int nZ, int max_k, double alpha, int hps,
                     int64_t n_obs_min, int64_t max_tests, fwo_result *out, int *Zs_out, int *nZs_out,
                     int64_t *num_tests_out, double *frac_out)
{
    if (nZ == 0) { /* :285 */
        set_result(out, NAN, NAN, -1, 1);
        *nZs_out = 0;
        *num_tests_out = -1;
        *frac_out = NAN;
        return 0;
    }
    if (c->kind == FWO_FZ_NZ) {
        /* tests.jl:293-308: rows with X != 0 and Y != 0 (hiton.jl:41-50,85); too few -> (0, 1, 0, false), 0 tests;
         * else cor_subset! over {X, Y} + Z_total, then the ordinary enumeration on that Float32 matrix */
        const int m = nZ + 2;
        int *vars = (int *)malloc(sizeof(int) * (size_t)m);
        int *locZ = (int *)malloc(sizeof(int) * (size_t)(nZ > 0 ? nZ : 1));
        vars[0] = X;
        vars[1] = Y;
        for (int j = 0; j < nZ; ++j) {
            vars[2 + j] = Z_total[j];
            locZ[j] = 2 + j;
        }
        int64_t nR = fz_nz_cor_subset(c, X, Y, vars, m, NULL);
        int status;
        if (n_obs_min > nR) {
            set_result(out, 0.0, 1.0, 0, 0);
            *nZs_out = 0;
            *num_tests_out = 0;
            *frac_out = 0.0;
            status = 1;
        } else {
            float *local = (float *)malloc(sizeof(float) * (size_t)m * (size_t)m);
            fz_nz_cor_subset(c, X, Y, vars, m, local);
            fwo_ctx tmp;
            memset(&tmp, 0, sizeof(tmp));
            tmp.kind = FWO_FZ;
            tmp.p = m;
            tmp.n = tmp.n_obs = (int)nR;
            tmp.cor32 = local;
            double *view = NULL;
            if (c->fz_stream) { /* no cor_mat (recursive_pcor = false): every test is pcor on the view's columns, tests.jl:253 */
                int64_t nv;
                view = fz_nz_view(c, X, Y, vars, m, &nv);
                tmp.fdata = view;
                tmp.fz_stream = 1;
            }
            int zl[16];
            status = fwo_test_subsets(&tmp, 0, 1, locZ, nZ, max_k, alpha, hps, n_obs_min, max_tests, out, zl, nZs_out,
                                      num_tests_out, frac_out);
            free(view);
            for (int j = 0; j < *nZs_out; ++j) Zs_out[j] = Z_total[zl[j] - 2];
            free(local);
        }
        free(vars);
        free(locZ);
        return status;
    }
    fwo_result lowest;
    set_result(&lowest, 0.0, 0.0, 0, 1); /* :287 */
    int lowest_Zs[16], lowest_n = 0;
    int64_t num_tests = 0;
    double num_tests_total = 0.0;
    int idx[16], Zs[16];
    for (int s = max_k; s >= 1; --s) {
        num_tests_total += binom_d(nZ, s);
        if (s > nZ) continue;
        for (int i = 0; i < s; ++i) idx[i] = i;
        for (;;) {
            for (int i = 0; i < s; ++i) Zs[i] = Z_total[idx[i]];
            fwo_result r;
            fwo_test(c, X, Y, Zs, s, hps, n_obs_min, &r);
            ++num_tests;
            /* baseline sampling only: a single enumeration can hold 10^9 subsets (cfg5); past the sample's deadline it is cut
             * short like a max_tests stop (its result is not used, only the tests counted so far) */
            const int timed_out = fwo_tls_deadline > 0.0 && (num_tests & 1023) == 0 && hiton_now_s() > fwo_tls_deadline;
            if (!issig(&r, alpha) || (max_tests > 0 && num_tests >= max_tests) || timed_out) { /* :326 */
                for (int rs = s - 1; rs >= 1; --rs) num_tests_total += binom_d(nZ, rs);
                *out = r;
                memcpy(Zs_out, Zs, sizeof(int) * (size_t)s);
                *nZs_out = s;
                *num_tests_out = num_tests;
                *frac_out = (double)num_tests / num_tests_total;
                return 1;
            } else if (r.pval >= lowest.pval) { /* :338 */
                lowest = r;
                memcpy(lowest_Zs, Zs, sizeof(int) * (size_t)s);
                lowest_n = s;
            }
            /* next lexicographic combination of positions */
            int i = s - 1;
            while (i >= 0 && idx[i] == nZ - s + i) --i;
            if (i < 0) break;
            ++idx[i];
            for (int j = i + 1; j < s; ++j) idx[j] = idx[j - 1] + 1;
        }
    }
    *out = lowest;
    memcpy(Zs_out, lowest_Zs, sizeof(int) * (size_t)lowest_n);
    *nZs_out = lowest_n;
    *num_tests_out = num_tests;
    *frac_out = (double)num_tests / num_tests_total;
    return 2;
}

/* ------------------------------------------------------------------------------------------
 * Level 0: pw_univar_neighbors (tests.jl:436-532) + benjamini_hochberg! (statfuns.jl:326-350)
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    int64_t pair; /* condensed index (X ascending, then Y) */
    int32_t X, Y;
    double stat, pval;
} fwo_pairrec;

typedef struct {
    int32_t *off; /* p+1 */
    int32_t *idx; /* partner ids ascending */
    double *stat;
    double *pval; /* BH-adjusted when FDR */
    int64_t n_tests;
    int64_t m_reliable;
} fwo_nbrs;

static int cmp_pairrec_pval(const void *a, const void *b)
{
    const fwo_pairrec *x = (const fwo_pairrec *)a, *y = (const fwo_pairrec *)b;
    if (x->pval < y->pval) return -1;
    if (x->pval > y->pval) return 1;
    return (x->pair > y->pair) - (x->pair < y->pair); /* stable */
}
static int cmp_pairrec_pair(const void *a, const void *b)
{
    const fwo_pairrec *x = (const fwo_pairrec *)a, *y = (const fwo_pairrec *)b;
    return (x->pair > y->pair) - (x->pair < y->pair);
}

/* statfuns.jl:326-350 on a free-standing vector (for test/statfuns.jl:61-71) */
void fwo_benjamini_hochberg(double *pvals, int64_t len, double alpha, int64_t m)
{
    if (len == 0) return;
    fwo_pairrec *f = (fwo_pairrec *)malloc(sizeof(fwo_pairrec) * (size_t)len);
    int64_t nf = 0;
    for (int64_t i = 0; i < len; ++i)
        if (pvals[i] < alpha) {
            f[nf].pair = i;
            f[nf].pval = pvals[i];
            ++nf;
        }
    if (nf == 0) {
        free(f);
        return;
    }
    qsort(f, (size_t)nf, sizeof(fwo_pairrec), cmp_pairrec_pval);
    double last = f[nf - 1].pval * (double)m / (double)nf;
    f[nf - 1].pval = last < 1.0 ? last : 1.0;
    for (int64_t i = nf - 2; i >= 0; --i) {
        double next_adj = f[i + 1].pval;
        double new_adj = f[i].pval * (double)m / (double)(i + 1);
        f[i].pval = next_adj < new_adj ? next_adj : new_adj;
    }
    for (int64_t i = 0; i < len; ++i) pvals[i] = NAN;
    for (int64_t i = 0; i < nf; ++i) pvals[f[i].pair] = f[i].pval;
    free(f);
}

void fwo_nbrs_free(fwo_nbrs *nb)
{
    if (!nb) return;
    free(nb->off);
    free(nb->idx);
    free(nb->stat);
    free(nb->pval);
    free(nb);
}

/* tests.jl:436-532.  The condensed NaN-initialised arrays of the reference are not materialised:
 * only p < alpha survives BH and only the count of non-NaN p-values (m) enters it. */
fwo_nbrs *fwo_level0(fwo_ctx *c, double alpha, int hps, int64_t n_obs_min, int FDR, int correct_reliable_only)
{
    const int p = c->p;
    int64_t cap = 1024, nf = 0;
    fwo_pairrec *f = (fwo_pairrec *)malloc(sizeof(fwo_pairrec) * (size_t)cap);
    int64_t n_tests = 0, m = 0, pair = 0;
    for (int X = 0; X < p - 1; ++X) {
        /* tests.jl:80-92: all tests fail if levels[X] < 2 */
        int x_fail = !FWO_IS_CONT(c) && c->levels[X] < 2;
        for (int Y = X + 1; Y < p; ++Y, ++pair) {
            fwo_result r;
            if (x_fail)
                set_result(&r, 0.0, 1.0, 0, 0);
            else
                fwo_test(c, X, Y, NULL, 0, hps, n_obs_min, &r);
            ++n_tests;
            double stat = r.stat, pval = r.pval;
            if (correct_reliable_only && !r.suff_power) stat = pval = NAN; /* :397-398 */
            if (!isnan(pval)) ++m;                                        /* :522-526 */
            if (pval < alpha) {
                if (nf == cap) {
                    cap *= 2;
                    f = (fwo_pairrec *)realloc(f, sizeof(fwo_pairrec) * (size_t)cap);
                }
                f[nf].pair = pair;
                f[nf].X = X;
                f[nf].Y = Y;
                f[nf].stat = stat;
                f[nf].pval = pval;
                ++nf;
            }
        }
    }
    if (!correct_reliable_only) m = (int64_t)p * (p - 1) / 2;
    if (FDR && nf > 0) {
        qsort(f, (size_t)nf, sizeof(fwo_pairrec), cmp_pairrec_pval);
        double last = f[nf - 1].pval * (double)m / (double)nf;
        f[nf - 1].pval = last < 1.0 ? last : 1.0;
        for (int64_t i = nf - 2; i >= 0; --i) {
            double next_adj = f[i + 1].pval;
            double new_adj = f[i].pval * (double)m / (double)(i + 1);
            f[i].pval = next_adj < new_adj ? next_adj : new_adj;
        }
        qsort(f, (size_t)nf, sizeof(fwo_pairrec), cmp_pairrec_pair);
    }
    /* tests.jl:372-388 condensed_stats_to_dict: keep adj p < alpha */
    fwo_nbrs *nb = (fwo_nbrs *)calloc(1, sizeof(fwo_nbrs));
    nb->off = (int32_t *)calloc((size_t)p + 1, sizeof(int32_t));
    for (int64_t i = 0; i < nf; ++i)
        if (f[i].pval < alpha) {
            nb->off[f[i].X + 1]++;
            nb->off[f[i].Y + 1]++;
        }
    for (int v = 0; v < p; ++v) nb->off[v + 1] += nb->off[v];
    int64_t tot = nb->off[p];
    nb->idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tot > 0 ? tot : 1));
    nb->stat = (double *)malloc(sizeof(double) * (size_t)(tot > 0 ? tot : 1));
    nb->pval = (double *)malloc(sizeof(double) * (size_t)(tot > 0 ? tot : 1));
    int32_t *fill = (int32_t *)calloc((size_t)p, sizeof(int32_t));
    for (int64_t i = 0; i < nf; ++i)
        if (f[i].pval < alpha) {
            int X = f[i].X, Y = f[i].Y;
            int64_t a = nb->off[X] + fill[X]++, b = nb->off[Y] + fill[Y]++;
            nb->idx[a] = Y;
            nb->stat[a] = f[i].stat;
            nb->pval[a] = f[i].pval;
            nb->idx[b] = X;
            nb->stat[b] = f[i].stat;
            nb->pval[b] = f[i].pval;
        }
    free(fill);
    free(f);
    nb->n_tests = n_tests;
    nb->m_reliable = m;
    return nb;
}

/* Baseline-sampling helpers (bench.py cpu_baseline leg; not part of any parity check).
 * fwo_nbrs_from_csr: wrap level-0 neighbour lists computed elsewhere so that fwo_learn can time the conditional
 * stage of sampled targets without a full CPU level-0 pass (4.7e8 pair tests at cfg4 take minutes on one core). */
fwo_nbrs *fwo_nbrs_from_csr(int p, const int64_t *off, const int32_t *idx, const double *stat, const double *pval,
                            int64_t n_tests)
{
    fwo_nbrs *nb = (fwo_nbrs *)calloc(1, sizeof(fwo_nbrs));
    int64_t tot = off[p];
    nb->off = (int32_t *)malloc(sizeof(int32_t) * ((size_t)p + 1));
    for (int v = 0; v <= p; ++v) nb->off[v] = (int32_t)off[v];
    nb->idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tot > 0 ? tot : 1));
    nb->stat = (double *)malloc(sizeof(double) * (size_t)(tot > 0 ? tot : 1));
    nb->pval = (double *)malloc(sizeof(double) * (size_t)(tot > 0 ? tot : 1));
    memcpy(nb->idx, idx, sizeof(int32_t) * (size_t)tot);
    memcpy(nb->stat, stat, sizeof(double) * (size_t)tot);
    memcpy(nb->pval, pval, sizeof(double) * (size_t)tot);
    nb->n_tests = n_tests;
    return nb;
}

/* Level-0 pair tests of the rows X = x_start, x_start + x_stride, ... (every Y > X), as fwo_level0 runs them, until
 * max_seconds have passed; returns the number of tests executed, *seconds = time spent. */
int64_t fwo_level0_sample(fwo_ctx *c, int hps, int64_t n_obs_min, int x_start, int x_stride, double max_seconds,
                          double *seconds)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    const double t0 = (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    double t1 = t0;
    int64_t n_tests = 0;
    volatile double sink = 0.0;
    for (int X = x_start; X < c->p - 1; X += (x_stride > 0 ? x_stride : 1)) {
        int x_fail = !FWO_IS_CONT(c) && c->levels[X] < 2;
        for (int Y = X + 1; Y < c->p; ++Y) {
            fwo_result r;
            if (x_fail)
                set_result(&r, 0.0, 1.0, 0, 0);
            else
                fwo_test(c, X, Y, NULL, 0, hps, n_obs_min, &r);
            sink += r.pval;
            ++n_tests;
        }
        clock_gettime(CLOCK_MONOTONIC, &ts);
        t1 = (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
        if (max_seconds > 0 && t1 - t0 > max_seconds) break;
    }
    (void)sink;
    *seconds = t1 - t0;
    return n_tests;
}

/* Level-0 pair tests of the rows X = x_start, x_start + x_stride, ... (at most max_rows of them; every Y > X) WITH their
 * results: the pairs whose raw p-value is below alpha (after the reliability rule of tests.jl:397-398), in (X, Y) order --
 * what fwo_level0 would hand to Benjamini-Hochberg from these rows.  For the full-size level-0 parity tests (the device run
 * with FDR = false keeps exactly these pairs).  Returns the number of pairs written (at most cap; -1 if cap was too small);
 * *n_tests = pair tests executed, *m = those with a non-NaN p-value. */
int64_t fwo_level0_rows(fwo_ctx *c, double alpha, int hps, int64_t n_obs_min, int x_start, int x_stride, int max_rows, int64_t cap,
                        int32_t *out_x, int32_t *out_y, double *out_stat, double *out_pval, int64_t *n_tests, int64_t *m)
{
    int64_t nf = 0, nt = 0, mm = 0;
    int rows = 0;
    for (int X = x_start; X < c->p - 1 && rows < max_rows; X += (x_stride > 0 ? x_stride : 1), ++rows) {
        int x_fail = !FWO_IS_CONT(c) && c->levels[X] < 2;
        for (int Y = X + 1; Y < c->p; ++Y) {
            fwo_result r;
            if (x_fail)
                set_result(&r, 0.0, 1.0, 0, 0);
            else
                fwo_test(c, X, Y, NULL, 0, hps, n_obs_min, &r);
            ++nt;
            double stat = r.stat, pval = r.pval;
            if (!r.suff_power) stat = pval = NAN;
            if (!isnan(pval)) ++mm;
            if (pval < alpha) {
                if (nf == cap) return -1;
                out_x[nf] = X;
                out_y[nf] = Y;
                out_stat[nf] = stat;
                out_pval[nf] = pval;
                ++nf;
            }
        }
    }
    *n_tests = nt;
    *m = mm;
    return nf;
}

int64_t fwo_nbrs_total(const fwo_nbrs *nb, int p) { return nb->off[p]; }
int64_t fwo_nbrs_ntests(const fwo_nbrs *nb) { return nb->n_tests; }
void fwo_nbrs_copy(const fwo_nbrs *nb, int p, int32_t *off, int32_t *idx, double *stat, double *pval)
{
    memcpy(off, nb->off, sizeof(int32_t) * ((size_t)p + 1));
    int64_t tot = nb->off[p];
    memcpy(idx, nb->idx, sizeof(int32_t) * (size_t)tot);
    memcpy(stat, nb->stat, sizeof(double) * (size_t)tot);
    memcpy(pval, nb->pval, sizeof(double) * (size_t)tot);
}

/* ------------------------------------------------------------------------------------------
 * HITON-PC (hiton.jl:109-149, 283-400) and the LGL driver (learning.jl:84-117, 203-279;
 * feed-forward order of interleaved.jl:112-183; misc.jl:137-272 post-processing)
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    int32_t *key;
    double *stat, *pval;
    int n, cap;
} odict; /* OrderedDict{Int,Tuple{Float64,Float64}} with insertion order; re-assignment keeps position */

static void od_init(odict *d)
{
    d->n = 0;
    d->cap = 16;
    d->key = (int32_t *)malloc(sizeof(int32_t) * 16);
    d->stat = (double *)malloc(sizeof(double) * 16);
    d->pval = (double *)malloc(sizeof(double) * 16);
}
static void od_free(odict *d)
{
    free(d->key);
    free(d->stat);
    free(d->pval);
}
static int od_find(const odict *d, int k)
{
    for (int i = 0; i < d->n; ++i)
        if (d->key[i] == k) return i;
    return -1;
}
static void od_set(odict *d, int k, double s, double pv)
{
    int i = od_find(d, k);
    if (i < 0) {
        if (d->n == d->cap) {
            d->cap *= 2;
            d->key = (int32_t *)realloc(d->key, sizeof(int32_t) * (size_t)d->cap);
            d->stat = (double *)realloc(d->stat, sizeof(double) * (size_t)d->cap);
            d->pval = (double *)realloc(d->pval, sizeof(double) * (size_t)d->cap);
        }
        i = d->n++;
        d->key[i] = k;
    }
    d->stat[i] = s;
    d->pval[i] = pv;
}

typedef struct {
    double alpha;
    int hps;
    int64_t n_obs_min;
    int max_k;
    int64_t max_tests;
    int FDR;
    int feed_forward;
    int round_size;  /* whitelist snapshot refresh interval in targets; 1 = reference single_il */
    int max_targets; /* > 0: stop after this many targets of the schedule (baseline sampling) */
    int target_stride; /* > 1 (only with feed_forward = 0): process every stride-th target of the schedule */
    double max_seconds; /* > 0: stop the conditional stage after this many seconds (baseline sampling) */
    int target_offset;  /* first schedule position to process (with target_stride: worker w of W takes w, w + W, ...) */
    double deadline;    /* internal: absolute time at which the baseline sample stops (0 = none); set by fwo_learn */
} fwo_params;

typedef struct {
    int *v;
    int n, cap;
} ivec;
static void iv_push(ivec *a, int x)
{
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 16;
        a->v = (int *)realloc(a->v, sizeof(int) * (size_t)a->cap);
    }
    a->v[a->n++] = x;
}

/* hiton.jl:109-149 hiton_backend for one phase. whitelist: byte mask over variables (may be NULL). */
static void hiton_phase(fwo_ctx *c, int T, const int *cands, int ncands, char phase, const fwo_params *P,
                        const uint8_t *whitelist, const odict *support, odict *accepted_dict, int64_t *n_tests)
{
    ivec acc = {0, 0, 0};
    if (phase == 'E')
        for (int i = 0; i < ncands; ++i) iv_push(&acc, cands[i]); /* :124 */
    for (int ci = 0; ci < ncands; ++ci) {
        int cand = cands[ci];
        /* baseline sampling only (max_seconds): a target with thousands of candidates must not overrun the time budget by
         * minutes -- the tests counted so far and the time spent still give the rate; results of such a run are not used */
        if (P->deadline > 0.0 && hiton_now_s() > P->deadline) break;
        if (whitelist && whitelist[cand]) { /* hiton.jl:20-30 */
            iv_push(&acc, cand);
            od_set(accepted_dict, cand, NAN, NAN);
            continue;
        }
        if (phase == 'E') { /* :134-136 deleteat!(accepted, findall(in(candidate), accepted)) */
            int w = 0;
            for (int i = 0; i < acc.n; ++i)
                if (acc.v[i] != cand) acc.v[w++] = acc.v[i];
            acc.n = w;
        }
        fwo_result r;
        int Zs[16], nZs;
        int64_t nt;
        double frac;
        /* hiton.jl:193 + :85 prepare_nzdata(T, .) then prepare_nzdata(candidate, .): with a dense matrix and a zero-adjusted
         * discrete test, the tests of this (T, candidate) pair see only the rows where T (if levels[T] > 2) and the candidate
         * (if levels[candidate] > 2) are non-zero (needs_nz_view, misc.jl:103-107) */
        uint8_t *view = NULL;
        if (!FWO_IS_CONT(c) && c->nz && !c->sparse && (c->levels[T] > 2 || c->levels[cand] > 2)) {
            view = (uint8_t *)malloc((size_t)(c->n > 0 ? c->n : 1));
            for (int i = 0; i < c->n; ++i)
                view[i] = (c->levels[T] <= 2 || dense_at(c, i, T) != 0) && (c->levels[cand] <= 2 || dense_at(c, i, cand) != 0);
            c->rowmask = view;
        }
        fwo_test_subsets(c, T, cand, acc.v, acc.n, P->max_k, P->alpha, P->hps, P->n_obs_min, P->max_tests, &r, Zs,
                         &nZs, &nt, &frac);
        if (view) {
            c->rowmask = NULL;
            free(view);
        }
        if (nt > 0) *n_tests += nt;
        /* hiton.jl:53-78 update_sig_result! (fast_elim = true) */
        if (acc.n == 0) {
            int si = od_find(support, cand);
            iv_push(&acc, cand);
            od_set(accepted_dict, cand, support->stat[si], support->pval[si]);
        } else if (issig(&r, P->alpha)) {
            iv_push(&acc, cand);
            od_set(accepted_dict, cand, r.stat, r.pval);
        }
    }
    free(acc.v);
}

typedef struct {
    double a_pval;
    int idx;
    int order;
} cand_rec;
static int cmp_cand(const void *a, const void *b)
{
    const cand_rec *x = (const cand_rec *)a, *y = (const cand_rec *)b;
    if (x->a_pval < y->a_pval) return -1;
    if (x->a_pval > y->a_pval) return 1;
    return (x->order > y->order) - (x->order < y->order);
}

/* hiton.jl:283-400 si_HITON_PC (prev_state 'S', no time limit). PC receives state_results. */
static void si_hiton_pc(fwo_ctx *c, int T, const fwo_nbrs *nb, const fwo_params *P, const uint8_t *whitelist,
                        odict *PC, int64_t *n_tests)
{
    const int o = nb->off[T], deg = nb->off[T + 1] - nb->off[T];
    if (P->max_k == 0) { /* learning.jl:171-172: nbr_dict = all_univar_nbrs, si_HITON_PC is not called */
        for (int i = 0; i < deg; ++i) od_set(PC, nb->idx[o + i], nb->stat[o + i], nb->pval[o + i]);
        return;
    }
    if (!FWO_IS_CONT(c) && c->levels[T] < 2) return; /* hiton.jl:182-184 */
    odict univar;
    od_init(&univar);
    for (int i = 0; i < deg; ++i) od_set(&univar, nb->idx[o + i], nb->stat[o + i], nb->pval[o + i]);
    /* hiton.jl:211-217 candidates sorted by (adjusted) p, stable */
    cand_rec *cr = (cand_rec *)malloc(sizeof(cand_rec) * (size_t)(deg > 0 ? deg : 1));
    int nc = 0;
    for (int i = 0; i < deg; ++i)
        if (univar.pval[i] < P->alpha) {
            cr[nc].a_pval = univar.pval[i];
            cr[nc].idx = univar.key[i];
            cr[nc].order = i;
            ++nc;
        }
    if (nc == 0) { /* :336-338 */
        free(cr);
        od_free(&univar);
        return;
    }
    qsort(cr, (size_t)nc, sizeof(cand_rec), cmp_cand);
    int *cands = (int *)malloc(sizeof(int) * (size_t)nc);
    for (int i = 0; i < nc; ++i) cands[i] = cr[i].idx;
    free(cr);
    odict TPC;
    od_init(&TPC);
    hiton_phase(c, T, cands, nc, 'I', P, whitelist, &univar, &TPC, n_tests);
    /* elimination: candidates = keys(TPC) in insertion order (:242) */
    int *pc_cands = (int *)malloc(sizeof(int) * (size_t)(TPC.n > 0 ? TPC.n : 1));
    for (int i = 0; i < TPC.n; ++i) pc_cands[i] = TPC.key[i];
    hiton_phase(c, T, pc_cands, TPC.n, 'E', P, whitelist, &TPC, PC, n_tests);
    /* hiton.jl:249-256 update_PC_dict! */
    for (int i = 0; i < PC->n; ++i) {
        int ti = od_find(&TPC, PC->key[i]);
        if (ti >= 0 && (TPC.pval[ti] > PC->pval[i] || isnan(PC->pval[i]))) {
            PC->stat[i] = TPC.stat[ti];
            PC->pval[i] = TPC.pval[ti];
        }
    }
    free(cands);
    free(pc_cands);
    od_free(&TPC);
    od_free(&univar);
}

typedef struct {
    int64_t n_edges;
    int32_t *src, *dst; /* src < dst */
    double *w;
    int64_t n_level0_tests, n_cond_tests;
    double t_level0, t_cond;
    int n_targets_done;
    /* directed per-target results (CSR) for fine-grained comparison */
    int32_t *pc_off, *pc_idx;
    double *pc_stat, *pc_pval;
} fwo_network;

void fwo_network_free(fwo_network *g)
{
    if (!g) return;
    free(g->src);
    free(g->dst);
    free(g->w);
    free(g->pc_off);
    free(g->pc_idx);
    free(g->pc_stat);
    free(g->pc_pval);
    free(g);
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
    int deg, idx;
} degrec;
static int cmp_deg(const void *a, const void *b)
{
    const degrec *x = (const degrec *)a, *y = (const degrec *)b;
    if (x->deg != y->deg) return (x->deg > y->deg) - (x->deg < y->deg);
    return (x->idx > y->idx) - (x->idx < y->idx);
}

/* learning.jl:51-64 automatic n_obs_min (fires for every test type: `<` binds looser than `&`) */
int64_t fwo_auto_n_obs_min(const fwo_ctx *c, int64_t n_obs_min, int hps, int max_k)
{
    if (n_obs_min >= 0) return n_obs_min;
    if (FWO_IS_CONT(c)) return 20;
    int64_t max_level = 0;
    for (int v = 0; v < c->p; ++v)
        if (c->levels[v] > max_level) max_level = c->levels[v];
    int64_t n_strata = ipow64(max_level, max_k);
    if (n_strata > 8) n_strata = 8;
    return (int64_t)hps * 2 * 2 * n_strata;
}

/* misc.jl:201-218 maxweight */
static double maxweight(double w1, double w2)
{
    if (isnan(w1)) return w2;
    if (isnan(w2)) return w1;
    double s1 = (w1 > 0) - (w1 < 0), s2 = (w2 > 0) - (w2 < 0);
    if (s1 * s2 < 0) return w1; /* "Arbitrarily choosing one": here the direction listed first */
    double a1 = fabs(w1), a2 = fabs(w2);
    return (a1 > a2 ? a1 : a2) * s1;
}

/* LGL (learning.jl:203-279) with the deterministic feed-forward schedule.  nb_in: optional precomputed
 * level-0 result (NULL -> computed here). */
/* Worker pool of fwo_learn_mt: the targets between two whitelist snapshots are independent of each other
 * (interleaved.jl:124-183: a worker's whitelist is the graph as the master knew it when the job was queued), so
 * one oracle context per thread runs them in any order; the master part (graph update) stays in schedule order. */
typedef struct {
    fwo_ctx *c;
    const fwo_nbrs *nb;
    const fwo_params *P;
    const degrec *order;
    const ivec *adj;
    const int *snap;
    odict *PCs;
    int lo, hi;       /* schedule positions [lo, hi) of the block */
    int *next;        /* shared ticket: positions are handed out from hi - 1 downwards (heaviest first) */
    int64_t n_tests;
    uint8_t *wl;
} mt_job;

static void *mt_worker(void *arg)
{
    mt_job *J = (mt_job *)arg;
    for (;;) {
        int k = __atomic_fetch_add(J->next, 1, __ATOMIC_RELAXED);
        int ti = J->hi - 1 - k;
        if (ti < J->lo) break;
        int T = J->order[ti].idx;
        const uint8_t *wlp = NULL;
        if (J->P->feed_forward && J->P->max_k > 0 && J->snap[T] > 0) {
            for (int i = 0; i < J->snap[T]; ++i) J->wl[J->adj[T].v[i]] = 1;
            wlp = J->wl;
        }
        si_hiton_pc(J->c, T, J->nb, J->P, wlp, &J->PCs[T], &J->n_tests);
        if (wlp)
            for (int i = 0; i < J->snap[T]; ++i) J->wl[J->adj[T].v[i]] = 0;
    }
    return NULL;
}

fwo_network *fwo_learn_mt(fwo_ctx **cs, int nthr, const fwo_params *P_in, const fwo_nbrs *nb_in);

fwo_network *fwo_learn(fwo_ctx *c, const fwo_params *P_in, const fwo_nbrs *nb_in) { return fwo_learn_mt(&c, 1, P_in, nb_in); }

/* nthr > 1: cs[0..nthr) are contexts over the same read-only inputs (one per thread: a context holds scratch);
 * needs round_size > 1 or feed_forward = 0, ignores target_stride / target_offset / max_seconds.  Same network,
 * directed lists and test count as the sequential loop (tests/test_oracle_golden.py). */
fwo_network *fwo_learn_mt(fwo_ctx **cs, int nthr, const fwo_params *P_in, const fwo_nbrs *nb_in)
{
    fwo_ctx *c = cs[0];
    fwo_params P = *P_in;
    const int p = c->p;
    P.n_obs_min = fwo_auto_n_obs_min(c, P.n_obs_min, P.hps, P.max_k);
    fwo_network *g = (fwo_network *)calloc(1, sizeof(fwo_network));
    if (P.n_obs_min > c->n) { /* learning.jl:66-73 error */
        g->n_edges = -1;
        return g;
    }
    double t0 = now_s();
    fwo_nbrs *nb_own = NULL;
    const fwo_nbrs *nb = nb_in;
    if (!nb) {
        nb_own = fwo_level0(c, P.alpha, P.hps, P.n_obs_min, P.FDR, 1);
        nb = nb_own;
    }
    g->n_level0_tests = nb->n_tests;
    g->t_level0 = now_s() - t0;
    /* learning.jl:97-98 targets by ascending univariate degree (stable) */
    degrec *order = (degrec *)malloc(sizeof(degrec) * (size_t)p);
    for (int v = 0; v < p; ++v) {
        order[v].deg = nb->off[v + 1] - nb->off[v];
        order[v].idx = v;
    }
    qsort(order, (size_t)p, sizeof(degrec), cmp_deg);

    odict *PCs = (odict *)malloc(sizeof(odict) * (size_t)p);
    for (int v = 0; v < p; ++v) od_init(&PCs[v]);
    /* running graph (interleaved.jl:102,136-140) as adjacency lists */
    ivec *adj = (ivec *)calloc((size_t)p, sizeof(ivec));
    uint8_t *wl = (uint8_t *)calloc((size_t)p, 1);
    /* snapshot of adjacency sizes at the last round boundary */
    int *snap = (int *)calloc((size_t)p, sizeof(int));
    int rs = P.round_size > 0 ? P.round_size : 1;
    int nt = (P.max_targets > 0 && P.max_targets < p) ? P.max_targets : p;
    double t1 = now_s();
    P.deadline = P.max_seconds > 0 ? t1 + P.max_seconds : 0.0;
    fwo_tls_deadline = P.deadline;
    int stride = (P.target_stride > 1 && !P.feed_forward) ? P.target_stride : 1;
    int n_done = 0;
    const int ti0 = (P.target_offset > 0 && !P.feed_forward) ? P.target_offset : 0;
    if (nthr > 1 && (rs > 1 || !P.feed_forward)) {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthr);
        mt_job *jobs = (mt_job *)calloc((size_t)nthr, sizeof(mt_job));
        for (int w = 0; w < nthr; ++w) jobs[w].wl = (uint8_t *)calloc((size_t)p, 1);
        const int blk = P.feed_forward ? rs : nt;
        for (int lo = 0; lo < nt; lo += blk) {
            int hi = lo + blk < nt ? lo + blk : nt, next = 0;
            for (int v = 0; v < p; ++v) snap[v] = adj[v].n;
            for (int w = 0; w < nthr; ++w) {
                mt_job *J = &jobs[w];
                J->c = cs[w]; J->nb = nb; J->P = &P; J->order = order; J->adj = adj; J->snap = snap; J->PCs = PCs;
                J->lo = lo; J->hi = hi; J->next = &next;
                pthread_create(&th[w], NULL, mt_worker, J);
            }
            for (int w = 0; w < nthr; ++w) pthread_join(th[w], NULL);
            for (int ti = lo; ti < hi; ++ti) { /* the master's graph update, in schedule order */
                int T = order[ti].idx;
                ++n_done;
                for (int i = 0; i < PCs[T].n; ++i) {
                    int u = PCs[T].key[i], dup = 0;
                    for (int j = 0; j < adj[T].n; ++j)
                        if (adj[T].v[j] == u) {
                            dup = 1;
                            break;
                        }
                    if (!dup) {
                        iv_push(&adj[T], u);
                        iv_push(&adj[u], T);
                    }
                }
            }
        }
        for (int w = 0; w < nthr; ++w) {
            g->n_cond_tests += jobs[w].n_tests;
            free(jobs[w].wl);
        }
        free(jobs);
        free(th);
    } else
    for (int ti = ti0; ti < nt; ti += stride) {
        int T = order[ti].idx;
        if (P.max_seconds > 0 && now_s() - t1 > P.max_seconds) break;
        ++n_done;
        /* rs = 1 is the reference's single_il master: job_q_buff_size = 1, so the first TWO targets of the schedule are
         * enqueued up front with an empty whitelist (interleaved.jl:62,76-86) -- no snapshot refresh before target 1 */
        if (ti % rs == 0 && !(rs == 1 && ti == 1))
            for (int v = 0; v < p; ++v) snap[v] = adj[v].n;
        const uint8_t *wlp = NULL;
        if (P.feed_forward && P.max_k > 0 && snap[T] > 0) {
            for (int i = 0; i < snap[T]; ++i) wl[adj[T].v[i]] = 1;
            wlp = wl;
        }
        si_hiton_pc(c, T, nb, &P, wlp, &PCs[T], &g->n_cond_tests);
        if (wlp)
            for (int i = 0; i < snap[T]; ++i) wl[adj[T].v[i]] = 0;
        for (int i = 0; i < PCs[T].n; ++i) { /* add_edge! is idempotent */
            int u = PCs[T].key[i], dup = 0;
            for (int j = 0; j < adj[T].n; ++j)
                if (adj[T].v[j] == u) {
                    dup = 1;
                    break;
                }
            if (!dup) {
                iv_push(&adj[T], u);
                iv_push(&adj[u], T);
            }
        }
    }
    g->t_cond = now_s() - t1;
    g->n_targets_done = n_done;
    fwo_tls_deadline = 0.0;

    /* misc.jl:137-159 make_weights ("cond_stat") -> per-direction weight, stored in PCs[].stat */
    for (int T = 0; T < p; ++T) {
        if (FWO_IS_CONT(c)) continue;
        const int o = nb->off[T], deg = nb->off[T + 1] - nb->off[T];
        for (int i = 0; i < PCs[T].n; ++i) {
            double us = NAN;
            for (int j = 0; j < deg; ++j)
                if (nb->idx[o + j] == PCs[T].key[i]) {
                    us = nb->stat[o + j];
                    break;
                }
            double sg = (us > 0) - (us < 0);
            if (isnan(us)) sg = NAN;
            PCs[T].stat[i] = sg * fabs(PCs[T].stat[i]);
        }
    }
    /* directed CSR dump */
    g->pc_off = (int32_t *)calloc((size_t)p + 1, sizeof(int32_t));
    for (int T = 0; T < p; ++T) g->pc_off[T + 1] = g->pc_off[T] + PCs[T].n;
    int64_t tot = g->pc_off[p];
    g->pc_idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tot > 0 ? tot : 1));
    g->pc_stat = (double *)malloc(sizeof(double) * (size_t)(tot > 0 ? tot : 1));
    g->pc_pval = (double *)malloc(sizeof(double) * (size_t)(tot > 0 ? tot : 1));
    for (int T = 0; T < p; ++T)
        for (int i = 0; i < PCs[T].n; ++i) {
            g->pc_idx[g->pc_off[T] + i] = PCs[T].key[i];
            g->pc_stat[g->pc_off[T] + i] = PCs[T].stat[i];
            g->pc_pval[g->pc_off[T] + i] = PCs[T].pval[i];
        }
    /* misc.jl:230-272 make_symmetric_graph, OR rule; lower-index endpoint's direction is "weight1" */
    int64_t cap = 1024;
    g->src = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    g->dst = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    g->w = (double *)malloc(sizeof(double) * (size_t)cap);
    for (int a = 0; a < p; ++a) {
        /* neighbours b > a reachable from either direction */
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 0) {
                for (int i = 0; i < PCs[a].n; ++i) {
                    int b = PCs[a].key[i];
                    if (b <= a) continue;
                    int ri = od_find(&PCs[b], a);
                    double w = maxweight(PCs[a].stat[i], ri >= 0 ? PCs[b].stat[ri] : NAN);
                    if (isnan(w)) continue;
                    if (g->n_edges == cap) {
                        cap *= 2;
                        g->src = (int32_t *)realloc(g->src, sizeof(int32_t) * (size_t)cap);
                        g->dst = (int32_t *)realloc(g->dst, sizeof(int32_t) * (size_t)cap);
                        g->w = (double *)realloc(g->w, sizeof(double) * (size_t)cap);
                    }
                    g->src[g->n_edges] = a;
                    g->dst[g->n_edges] = b;
                    g->w[g->n_edges] = w;
                    ++g->n_edges;
                }
            } else {
                for (int i = 0; i < adj[a].n; ++i) {
                    int b = adj[a].v[i];
                    if (b <= a) continue;
                    if (od_find(&PCs[a], b) >= 0) continue; /* handled in pass 0 */
                    int ri = od_find(&PCs[b], a);
                    if (ri < 0) continue;
                    double w = maxweight(PCs[b].stat[ri], NAN);
                    if (isnan(w)) continue;
                    if (g->n_edges == cap) {
                        cap *= 2;
                        g->src = (int32_t *)realloc(g->src, sizeof(int32_t) * (size_t)cap);
                        g->dst = (int32_t *)realloc(g->dst, sizeof(int32_t) * (size_t)cap);
                        g->w = (double *)realloc(g->w, sizeof(double) * (size_t)cap);
                    }
                    g->src[g->n_edges] = a;
                    g->dst[g->n_edges] = b;
                    g->w[g->n_edges] = w;
                    ++g->n_edges;
                }
            }
        }
    }
    for (int v = 0; v < p; ++v) {
        od_free(&PCs[v]);
        free(adj[v].v);
    }
    free(PCs);
    free(adj);
    free(wl);
    free(snap);
    free(order);
    if (nb_own) fwo_nbrs_free(nb_own);
    return g;
}

/* accessors for ctypes */
int64_t fwo_network_nedges(const fwo_network *g) { return g->n_edges; }
void fwo_network_copy(const fwo_network *g, int32_t *src, int32_t *dst, double *w)
{
    memcpy(src, g->src, sizeof(int32_t) * (size_t)g->n_edges);
    memcpy(dst, g->dst, sizeof(int32_t) * (size_t)g->n_edges);
    memcpy(w, g->w, sizeof(double) * (size_t)g->n_edges);
}
void fwo_network_stats(const fwo_network *g, int64_t *n_level0, int64_t *n_cond, double *t_level0, double *t_cond,
                       int *n_targets)
{
    *n_level0 = g->n_level0_tests;
    *n_cond = g->n_cond_tests;
    *t_level0 = g->t_level0;
    *t_cond = g->t_cond;
    *n_targets = g->n_targets_done;
}
int64_t fwo_network_pc_total(const fwo_network *g, int p) { return g->pc_off[p]; }
void fwo_network_pc_copy(const fwo_network *g, int p, int32_t *off, int32_t *idx, double *stat, double *pval)
{
    memcpy(off, g->pc_off, sizeof(int32_t) * ((size_t)p + 1));
    int64_t tot = g->pc_off[p];
    memcpy(idx, g->pc_idx, sizeof(int32_t) * (size_t)tot);
    memcpy(stat, g->pc_stat, sizeof(double) * (size_t)tot);
    memcpy(pval, g->pc_pval, sizeof(double) * (size_t)tot);
}

/* ------------------------------------------------------------------------------------------
 * Level-0 Pearson matrix (learning.jl:42-45 -> Statistics.cor: centre, X'X, cov2cor! with clamp and
 * unit diagonal).  Accumulation in Float64, result rounded to Float32: the `prec=64` path the goldens
 * pin (cor in Float64, then convert(Matrix{Float32}, .)).  The prec=32 path of the reference sums in
 * Float32 inside BLAS in an unknowable order: parity for it is a tolerance.
 * data: n x p column-major Float64.  out32 / out64: p x p (either may be NULL).
 * ---------------------------------------------------------------------------------------- */
void fwo_cor(const double *data, int n, int p, float *out32, double *out64)
{
    double *xc = (double *)malloc(sizeof(double) * (size_t)n * (size_t)p);
    double *sd = (double *)malloc(sizeof(double) * (size_t)p);
    for (int v = 0; v < p; ++v) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += data[(int64_t)v * n + i];
        double mean = s / (double)n;
        double ss = 0.0;
        for (int i = 0; i < n; ++i) {
            double d = data[(int64_t)v * n + i] - mean;
            xc[(int64_t)v * n + i] = d;
            ss += d * d;
        }
        sd[v] = sqrt(ss);
    }
    for (int a = 0; a < p; ++a) {
        for (int b = a; b < p; ++b) {
            double r;
            if (a == b) {
                r = 1.0;
            } else {
                double s = 0.0;
                const double *xa = xc + (int64_t)a * n, *xb = xc + (int64_t)b * n;
                for (int i = 0; i < n; ++i) s += xa[i] * xb[i];
                r = s / (sd[a] * sd[b]);
                if (r > 1.0) r = 1.0; /* clampcor */
                if (r < -1.0) r = -1.0;
            }
            if (out64) out64[(int64_t)b * p + a] = out64[(int64_t)a * p + b] = r;
            if (out32) out32[(int64_t)b * p + a] = out32[(int64_t)a * p + b] = (float)r;
        }
    }
    free(xc);
    free(sd);
}

/* Numerical identity used by the device kernels (csrc/fw_fz.hip round5_*): for every integer |n| <= limit,
 * q0 = n*c, r = fma(-q0, 1e5, n), q = fma(r, c, q0) with c = RN(1e-5) equals the correctly rounded n / 1e5,
 * in Float32 and in Float64.  Returns the number of counter-examples (expected 0). */
int64_t fwo_check_fast_div1e5(int64_t limit)
{
    int64_t bad = 0;
    const float c32 = 1e-5f;
    const double c64 = 1e-5;
    for (int64_t n = -limit; n <= limit; ++n) {
        const float fn = (float)n;
        const float q0 = fn * c32, r = fmaf(-q0, 100000.0f, fn), q1 = fmaf(r, c32, q0);
        if (q1 != fn / 100000.0f) ++bad;
        const double dn = (double)n;
        const double p0 = dn * c64, rr = fma(-p0, 100000.0, dn), p1 = fma(rr, c64, p0);
        if (p1 != dn / 100000.0) ++bad;
    }
    return bad;
}
