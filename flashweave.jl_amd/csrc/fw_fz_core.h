// Device-side core of the Fisher-z path shared by the segment kernels (fw_fz.hip) and the persistent per-target kernel
// (fw_devhiton.hip): p-values, the pcor_rec levels (statfuns.jl:23-75) and the run-based body of test_subsets
// (tests.jl:281-346).  Compiled with -ffp-contract=off in every unit that includes it.
#pragma once
#include "fw_internal.h"
#include "fw_unrank.h"

// ------------------------------------------------------------------------------------------------
// 3. shared device math
// ------------------------------------------------------------------------------------------------
// out of line for the segment kernel: inlined, the ~70 polynomial coefficients of log / erfc are hoisted into VGPRs
// across the hot loop, which only reaches this code for tests inside the significance guard band
static __device__ __noinline__ double fz_pval_slow(double r, double zscale)
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
static __device__ __noinline__ double fz_xkey_slow(double r, double zscale)
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

// ---- NaN-free forms for the table kernel's wave-uniform fast path ----
// There every input is known not to be NaN (table entries carry a "clean" flag, the gathered entry is checked), so no
// intermediate can be NaN either (sums / products of values in [-1, 1], square roots of 1 - v^2 >= 0, divisions by a
// non-zero finite denominator) and `v < -1 ? -1 : v`, `v >= 1 ? 1 : v` (six instructions in Float64: the selects keep a
// NaN) are v_max_f64 / v_min_f64: same values for every non-NaN v.
__device__ __forceinline__ double fz_clamp_unit_nn(double v) { return __builtin_fmin(__builtin_fmax(v, -1.0), 1.0); }
// round5 without the "not finite: hand the argument back" select (finite arguments give finite results; a NaN comes out
// as a NaN either way)
__device__ __forceinline__ float round5_f32_nn(float x)
{
    const float n = rintf(x * 100000.0f);
    const float q0 = n * 1e-5f;
    const float r = fmaf(-q0, 100000.0f, n);
    return fmaf(r, 1e-5f, q0);
}
__device__ __forceinline__ double round5_f64_nn(double x)
{
    const double n = rint(x * 100000.0);
    const double q0 = n * 1e-5;
    const double r = fma(-q0, 100000.0, n);
    return fma(r, 1e-5, q0);
}
// Partial correlation rho(X, Y | z[0..K-1]) with the reference's peel order (last element first) and its mixed
// Float32/Float64 arithmetic, evaluated bottom-up: U = [X, Y, z_K, ..., z_1]; level j conditions
// every remaining pair (a before b in U) on z_j.  (The recursion of statfuns.jl:44-53 touches exactly these
// pairs in exactly these argument orders; level-1 values are symmetric.)
// levels 2..K of the bottom-up form, in place on the level-1 values R / is32 (pairs a < b < M - 1 of
// U = [X, Y, z_K, ..., z_2]; z_1 has been conditioned on): returns rho(X, Y | z_1..z_K)
// NN: no NaN among the level-1 values (checked by the caller), hence none anywhere below -- the selects that keep a NaN
// alive (round5's isfinite, the two clamps) become plain arithmetic / v_max + v_min: same values for every non-NaN input
template <int K, bool NN = false>
__device__ __forceinline__ double fz_pcor_levels(double (&R)[K + 2][K + 2], bool (&is32)[K + 2][K + 2])
{
    constexpr int M = K + 2;
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
                            ev = (double)(NN ? round5_f32_nn((float)va - (float)prod) : round5_f32((float)va - (float)prod));
                        else
                            ev = NN ? round5_f64_nn(va - prod) : round5_f64(va - prod);
                        if (b32) {
                            const float bb = (float)vb * (float)vb;
                            d1 = (double)sqrtf(1.0f - bb);
                        } else {
                            d1 = fz_sqrt_unit(1.0 - vb * vb);
                        }
                    } else {
                        ev = NN ? round5_f64_nn(va - vb * vc) : round5_f64(va - vb * vc);
                        d1 = fz_sqrt_unit(1.0 - vb * vb);
                    }
                    const double d2 = fz_sqrt_unit(1.0 - vc * vc);
                    const double denom = d1 * d2;
                    double v = (denom == 0.0) ? 0.0 : ev / denom;
                    if (NN) {
                        v = fz_clamp_unit_nn(v);
                    } else {
                        if (v < -1.0)
                            v = -1.0;
                        else if (v >= 1.0)
                            v = 1.0;
                    }
                    R[a][b] = v;
                    is32[a][b] = false;
                }
            }
    }
    return R[0][1];
}

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
    return fz_pcor_levels<K>(R, is32);
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


#define FW_ACC_LDS 2048

// C(m, t) and the lexicographic unranking of subset ranks: fw_unrank.h (shared with the host-side exhaustive check)
#define binom_u64 fw_binom_u64
#define unrank_comb fw_unrank_comb

// ---- tagged scalar forms of the pcor_rec levels (same arithmetic as fz_pcor_dp, used by the run-based kernel) ----
struct TV {
    double v;
    bool f32;
};

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

// fz_sqrt_unit without its x == 0 select: NaN for x = 0 (the caller tests x itself), the same bits otherwise
__device__ __forceinline__ double fz_sqrt_unit_raw(double x)
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
    return fma(d, h, g);
}

// pc_l1_r with the clamps done in Float32 (the conversion to Float64 is exact, so the same values) and the "still a
// Float32 value" flag false for a NaN as well (a NaN is a NaN in either arithmetic: the slow path decides nothing else)
__device__ __forceinline__ float pc_l1_rf(float xy, float xz, float yz, float rxz, float ryz, bool &f32ok)
{
    const float prod = xz * yz;
    const float e = round5_f32_nn(xy - prod);
    const float d = rxz * ryz;
    const bool nz = d != 0.0f;
    float q = e / (nz ? d : 1.0f);
    q = nz ? q : 0.0f;
    const bool lo = q < -1.0f, hi = q >= 1.0f;
    q = lo ? -1.0f : q;
    q = hi ? 1.0f : q;
    f32ok = nz && !lo && !hi && q == q;
    return q;
}

// n / d, correctly rounded, for the operands of the NaN-free path: |n| <= 2 a multiple of 1e-5 (or zero), d in [2^-60, 1] -- far from
// every range in which v_div_scale_f64 rescales (operands with extreme exponents, quotients near the ends of the exponent range),
// so the compiler's IEEE sequence (2 x v_div_scale, v_rcp_f64, two Newton steps, quotient + residual, v_div_fmas, v_div_fixup)
// degenerates to the same instructions on the same values without the two scalings and with a plain fma for v_div_fmas:
// same bits, two instructions and a VCC dependency less per division.  v_div_fixup stays: it gives 0 / d the sign of the
// numerator (the residual step turns -0 into +0) and d = 0 its inf / NaN (selected away by the callers).
// fw_selftest(FW_SELFTEST_DIV) compares it with the compiler's division on the device over these operand ranges.
__device__ __forceinline__ double fz_div_nn(double n, double d)
{
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    const double q = n * y;
    const double r = __builtin_fma(-d, q, n);
    return __builtin_amdgcn_div_fixup(__builtin_fma(r, y, q), d, n);
}

// pc_l2_d2 for NaN-free children of which some may be Float64 literals (0, +-1: exact in Float32 as well, so the product b c and the
// root d1 = sqrt(1 - b^2) are the same values in either arithmetic): only `a - b c` and its rounding are done in Float32 when all three
// are Float32 values and in Float64 otherwise
__device__ __forceinline__ double pc_l2_mix_d1_nn(float a, float b, float c, bool all32, double d1, double d2c)
{
    const float prod = b * c;
    const double ev32 = (double)round5_f32_nn(a - prod);
    const double ev64 = round5_f64_nn((double)a - (double)prod);
    const double ev = all32 ? ev32 : ev64;
    const double denom = d1 * d2c;
    const double v = (denom == 0.0) ? 0.0 : fz_div_nn(ev, denom);
    return fz_clamp_unit_nn(v);
}

__device__ __forceinline__ double pc_l2_all32_d1_nn(float a, float b, float c, double d1, double d2c)
{
    const float prod = b * c;
    const double ev = (double)round5_f32_nn(a - prod);
    const double denom = d1 * d2c;
    const double v = (denom == 0.0) ? 0.0 : fz_div_nn(ev, denom);
    return fz_clamp_unit_nn(v);
}

// the product of the two roots is zero exactly when one of the arguments is (roots of values >= 2^-53 are >= 2^-27: no
// underflow in the product), so one test of the arguments replaces the two selects inside fz_sqrt_unit and `denom == 0`
__device__ __forceinline__ double pc_l3_nn(double a, double b, double c)
{
    const double ev = round5_f64_nn(a - b * c);
    const double xb = 1.0 - b * b, xc = 1.0 - c * c;
    const double denom = fz_sqrt_unit_raw(xb) * fz_sqrt_unit_raw(xc);
    const double v = (xb == 0.0 || xc == 0.0) ? 0.0 : fz_div_nn(ev, denom);
    return fz_clamp_unit_nn(v);
}

// pc_l3_nn in two halves for the screened size-3 test of the table kernel: the test first looks at ev = round5(a - b c) and the two
// radicands xb = 1 - b^2, xc = 1 - c^2 (fz_seg_body: ev^2 against thr^2 xb xc decides "significant for sure" and "clearly not the
// lane's maximum-p test" without the two square roots and the division); the quotient itself -- the SAME operations on the same
// values as pc_l3_nn, hence the same bits -- is only taken where its value is needed: a stopping test, a test inside a guard band,
// the lane's maximum-p candidate once per run.  Out of line: rare, and the hot loop should not carry its registers.
static __device__ __noinline__ double fz_l3_finish(double ev, double xb, double xc)
{
    const double denom = fz_sqrt_unit_raw(xb) * fz_sqrt_unit_raw(xc);
    const double v = (xb == 0.0 || xc == 0.0) ? 0.0 : fz_div_nn(ev, denom);
    return fz_clamp_unit_nn(v);
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
#ifndef FW_FZ_FASTLOOP
#define FW_FZ_FASTLOOP 1  // table kernel: the common size-3 test behind one wave-uniform branch (0: the general form only; A/B knob)
#endif
#ifndef FW_L3_ENTRY_NN
#define FW_L3_ENTRY_NN 1  // long-list kernel: NaN-free arithmetic decided per test (its two table entries) where the chunk's tables are not clean (0: per chunk only; A/B knob)
#endif
#ifndef FW_FZ_INTERLEAVE
#define FW_FZ_INTERLEAVE 1  // table kernel: interleaved lane <-> rank mapping for size-3 chunks (0: runs everywhere; A/B knob)
#endif
#ifndef FW_RUN_MAX
#define FW_RUN_MAX 32  // chunk = 8192 ranks: fewer table builds / unrankings per test (16 -> 32: -9 % kernel time at cfg3)
#endif
#ifndef FW_FZ_SCREEN
#define FW_FZ_SCREEN 1  // r06: size-3 fast loop -- a conservative Float32 screen in front of the exact test (see "cheap screen" in fz_seg_body; 0: off, A/B knob)
#endif
#ifndef FW_FZ_CHEAPRED
#define FW_FZ_CHEAPRED 1  // r06: the end-of-chunk reductions take their common case first (0: the r05 form; A/B knob)
#endif
#ifndef FW_RUN_MAX3
#define FW_RUN_MAX3 64  // size-3 table kernel (r06): chunks of up to 16 384 ranks -- a segment of the big launches (9 000-14 000 ranks) is ONE chunk, one table build, one
                        // round of reductions instead of two; a chunk whose z1-blocks would not fit the table (1 127 entries at worst, tab_bound.py 512 16384) is halved
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
#define FZ_TAB_ZMASK 0x0FFFFFFF
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
static __device__ void fz_thresholds_dev(double alpha, double zscale, double *thr)
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


// ---- level-2 tables for subsets of 4 and 5 variables (HIGHK && TAB, |accepted| <= FZ_HK_A) ----
// With (z1, z2) = accepted[(i, j)] fixed, the bottom-up form of statfuns.jl:44-53 (fz_pcor_dp) conditions every pair of
// {X, Y, later positions} on z1 and then on z2 before the remaining positions are looked at: those level-2 values
// depend on (i, j) and the pair only.  Per chunk the workgroup builds, for every (i, j) sub-block the chunk touches, the
// table {rho(X,Y|z1,z2); rho(X,v|..), rho(Y,v|..) for every later position v; rho(p,q|..) for later positions p > q}
// in LDS (Float64: level-2 results are never Float32 values), and a test of size 5 is 6 + 3 + 1 level >= 3 formulas on
// ten table entries instead of 15 + 10 + 6 + 3 + 1 formulas on 21 gathered matrix entries (size 4: 3 + 1 instead of
// 10 + 6 + 3 + 1).  Same formulas on the same values in the same order: results are bit-identical to fz_pcor_dp.
// The square root of 1 - v^2 is taken once per conditioning value and shared by the formulas that use it.
#define FZ_HK_A FW_HK_A
#define FZ_HK_CAP 4096  // table entries (doubles) per workgroup: 32 KB
#define FZ_HK_DIR 96    // sub-blocks per chunk
__device__ __forceinline__ double pc_l3s(double a, double b, double c, double sb, double sc)
{
    const double ev = round5_f64(a - b * c);
    const double denom = sb * sc;
    double v = (denom == 0.0) ? 0.0 : ev / denom;
    v = v < -1.0 ? -1.0 : v;
    v = v >= 1.0 ? 1.0 : v;
    return v;
}
__device__ __forceinline__ double pc_l3s_nn(double a, double b, double c, double sb, double sc)  // no NaN among the inputs
{
    const double ev = round5_f64_nn(a - b * c);
    const double denom = sb * sc;
    const double v = (denom == 0.0) ? 0.0 : fz_div_nn(ev, denom);
    return fz_clamp_unit_nn(v);
}
__device__ __forceinline__ double fz_sq1(double v) { return fz_sqrt_unit(1.0 - v * v); }
// entries of one sub-block table with n later positions
__device__ __forceinline__ int fz_hk_entries(int n) { return 1 + 2 * n + n * (n - 1) / 2; }
// tests of one sub-block: C(n, t) for t = 2, 3 (n <= FZ_HK_A)
__device__ __forceinline__ unsigned int fz_hk_tests(int n, int t)
{
    const unsigned int u = (unsigned int)n;
    if (n < t) return 0u;
    const unsigned int h = u * (u - 1u) / 2u;
    return t == 2 ? h : h * (u - 2u) / 3u;
}

template <bool NN>
__device__ __forceinline__ double fz_hk_stat(const double *__restrict__ tb, int n, int s, int l3, int l4, int l5)
{
#define HK_L3(a, b, c, sb, sc) (NN ? pc_l3s_nn(a, b, c, sb, sc) : pc_l3s(a, b, c, sb, sc))
    const double *tx = tb + 1, *ty = tb + 1 + n, *tp = tb + 1 + 2 * n;
    const double XY = tb[0], X3 = tx[l3], Y3 = ty[l3], X4 = tx[l4], Y4 = ty[l4];
    const double P43 = tp[l4 * (l4 - 1) / 2 + l3];
    const double sX3 = fz_sq1(X3), sY3 = fz_sq1(Y3), s43 = fz_sq1(P43);
    // level 3: condition on z3
    const double T_XY = HK_L3(XY, X3, Y3, sX3, sY3);
    const double T_X4 = HK_L3(X4, X3, P43, sX3, s43);
    const double T_Y4 = HK_L3(Y4, Y3, P43, sY3, s43);
    const double sX4 = fz_sq1(T_X4), sY4 = fz_sq1(T_Y4);
    if (s == 4) return HK_L3(T_XY, T_X4, T_Y4, sX4, sY4);  // level 4: condition on z4
    const int o5 = l5 * (l5 - 1) / 2;
    const double X5 = tx[l5], Y5 = ty[l5], P54 = tp[o5 + l4], P53 = tp[o5 + l3];
    const double s53 = fz_sq1(P53);
    const double T_X5 = HK_L3(X5, X3, P53, sX3, s53);
    const double T_Y5 = HK_L3(Y5, Y3, P53, sY3, s53);
    const double T_54 = HK_L3(P54, P53, P43, s53, s43);
    const double s54 = fz_sq1(T_54);
    // level 4: condition on z4
    const double Q_XY = HK_L3(T_XY, T_X4, T_Y4, sX4, sY4);
    const double Q_X5 = HK_L3(T_X5, T_X4, T_54, sX4, s54);
    const double Q_Y5 = HK_L3(T_Y5, T_Y4, T_54, sY4, s54);
    // level 5: condition on z5
    return HK_L3(Q_XY, Q_X5, Q_Y5, fz_sq1(Q_X5), fz_sq1(Q_Y5));
#undef HK_L3
}

// ---- level-1 table for subsets of 4 and 5 variables over LONG accepted lists (HIGHK && !TAB, FZ_HK_A < |accepted| <= FZ_L1_A) ----
// A (z1, z2) table does not fit LDS there, but a chunk of 8192 ranks nearly always lies inside ONE z1-block (C(a-1-i, s-1)
// subsets), and everything level 1 needs about a later position v given z1 -- rho(X,v|z1), rho(Y,v|z1), cor[v][z1] and its
// Float32 root, the entries of the size-3 table -- is one LDS entry per position.  A test then gathers the C(s-1, 2)
// matrix entries among its own later positions (6 instead of 21 for size 5) and evaluates 6 level-1 formulas with
// ready-made roots instead of 15 full ones; levels 2..K are fz_pcor_levels as before (same values, same order).
#define FZ_L1_A 512  // (r03: 1024 -> 512: with the level-3 tables LDS bounds the occupancy of this variant; cfg5's longest list is 480)
#ifdef FW_FZ_FASTDBG
static __device__ unsigned long long fz_fast_cnt[24];
#endif
static __device__ int fz_dbg_flags;  // profiling knob (FW_FZ_DBG, set by fz_ensure_thresholds): bit 0 = no level-1 table
template <int K>
__device__ __forceinline__ double fz_l1t_stat(const float *__restrict__ cor, int p, const float4 *__restrict__ tab,
                                              const unsigned char *__restrict__ tabf, const int *__restrict__ acc, const int *pos,
                                              double a1v, bool a1f, bool clean /* no NaN in the table (whole chunk) */)
{
    constexpr int M = K + 2;
    double R[M][M];
    bool is32[M][M];
    float c[K - 1], r[K - 1];
    int zid[K - 1];
    R[0][1] = a1v;
    is32[0][1] = a1f;
#pragma unroll
    for (int t = 0; t < K - 1; ++t) {  // U[2 + t] = z_{K - t} = accepted[pos[K - 1 - t]]
        const int pp = pos[K - 1 - t];
        const float4 e = tab[pp];
        const int f = tabf[pp];
        R[0][2 + t] = (double)e.x;
        is32[0][2 + t] = (f & 1) != 0;
        R[1][2 + t] = (double)e.y;
        is32[1][2 + t] = (f & 2) != 0;
        c[t] = e.z;
        r[t] = e.w;
        zid[t] = acc[pp];
    }
#pragma unroll
    for (int t = 0; t < K - 1; ++t)
#pragma unroll
        for (int u = t + 1; u < K - 1; ++u) {
#ifdef FW_CORT_TRANSPOSED
            const TV v = pc_l1_r(cor[(size_t)zid[u] * p + zid[t]], c[t], c[u], r[t], r[u]);
#else
            const TV v = pc_l1_r(cor[(size_t)zid[t] * p + zid[u]], c[t], c[u], r[t], r[u]);
#endif
            R[2 + t][2 + u] = v.v;
            is32[2 + t][2 + u] = v.f32;
            clean = clean && v.v == v.v;
        }
#ifndef FW_L1T_NO_NN  // (A/B knob)
    if (__all(clean)) return fz_pcor_levels<K, true>(R, is32);  // wave-uniform: no NaN among the level-1 values
#endif
    return fz_pcor_levels<K, false>(R, is32);
}

// ---- level-3 position tables for subsets of 4 and 5 variables over long lists (r03; on top of the level-1 table) ----
// Inside one z1-block, fix (z2, z3) = accepted[(j, k)] as well.  Everything the bottom-up form of statfuns.jl:44-53 needs about a
// later position w given (z1, z2, z3) is a handful of numbers per POSITION (not per pair):
//     P2[w] = rho(w,z2|z1)            Q3[w] = rho(w,z3|z1,z2)
//     X3[w] = rho(X,w|z1,z2,z3)       Y3[w] = rho(Y,w|z1,z2,z3)       A4[w] = rho(X,Y|z1,z2,z3,w)
// and the square roots the next level takes of them.  Per chunk the workgroup builds them for the (z2, z3) sub-blocks the chunk
// touches (8 formulas + 2 matrix gathers per position; a sub-block of n later positions holds C(n, 2) size-5 tests).  A size-5
// test (z4, v) = positions (l, m) is then: ONE gather cor[v][z4], the chain of that pair -- rho(v,z4|z1) (level-1 table),
// rho(v,z4|z1,z2) from P2[m], P2[l], rho(v,z4|z1,z2,z3) from Q3[m], Q3[l] -- then rho(X,v|z1..z4), rho(Y,v|z1..z4) from X3 / Y3
// and the final value from A4[l]: 6 formulas and 3 Float64 square roots instead of the 6 + 20 formulas of the level-1 form
// (r02 PMC: 21.7 VALU wave-instructions per executed cfg5 test).  A size-4 test (z4 = position l) IS A4[l].  Same formulas on
// the same values with the same argument roles as fz_pcor_dp (level 2 is not symmetric in its two conditioning values: the
// later position is the first argument, as in U = [X, Y, z_K, ..., z_2]): bit-identical, checked by
// tests/test_gpu_fz.py::test_size_4_5_table_kernels_value_at_random_ranks and the cfg5 long-list oracle test.
#ifndef FZ_L3_CAP
#define FZ_L3_CAP 448
#endif
// FZ_L3_CAP: positions (over all sub-blocks of a chunk): 31 KB; with the level-1 table and the list 48 KB -> three workgroups per CU
#define FZ_L3_DIR 64   // sub-blocks per chunk

// HIGHK: subsets of size 4-5 possible; LOCAL: per-job matrices (fz_nz); TAB: size-3 subsets through the LDS table
// (the host routes only segments of jobs with |accepted| <= FZ_TAB_A to a TAB launch); HIGHK && TAB: subsets of 4 and 5
// variables through the level-2 tables above (jobs with |accepted| <= FZ_HK_A), sizes <= 3 through the in-lane forms
template <bool HIGHK, bool LOCAL, bool TAB>
__device__ __forceinline__ void fz_seg_body(const float *__restrict__ cor_g, int p_g, const FwSeg seg /* workgroup-uniform */,
                                            const int32_t *__restrict__ gacc /* the job's accepted list */,
                                            const bool remote_acc /* list written through by another workgroup: sc1 loads */,
                                            FwSegOut *out_rec, int max_k,
                                            double alpha, double zscale_g, long long max_tests,
                                            const double *__restrict__ thr_g, const FwNzJob *__restrict__ recs,
                                            long long n_obs_min)
{
    constexpr bool TAB3 = TAB && !HIGHK;          // size-3 table (max_k <= 3)
    constexpr bool HK = TAB && HIGHK && !LOCAL;   // level-2 tables for sizes 4 and 5
    static_assert(!(TAB && HIGHK && LOCAL), "no level-2 table variant for per-job matrices");
    constexpr bool L1T = HIGHK && !TAB && !LOCAL;  // level-1 table for sizes 4 and 5 over long lists
#ifndef FW_FZ_L1T_ROW32
#define FW_FZ_L1T_ROW32 1  // r06: the size-5 gather of a local matrix with a 32-bit row base (cfg5 54.6 -> 54.0 s)
#endif
#ifndef FW_FZ_TMAT_HIGHK
#define FW_FZ_TMAT_HIGHK 1  // r06: the max_k 4-5 kernels take a target's local matrix too (0: the p x p matrix, A/B knob)
#endif
#ifndef FW_L3_SCREEN
#define FW_L3_SCREEN 0  // (A/B knob; r05: built, bit-identical, cfg5 58.72 s with it against 58.63 s without -- the tests of cfg5 live where p underflows to 0 (a p = 0 branch of the maximum-p bookkeeping was tried too: neutral there, -2 % at cfg3, removed) -- so the r03 / r04 form stays: every size-5 test of the position tables takes its quotient)
#endif
    // SCR: the test loop keeps a statistic as (numerator, radicands) where a screen on the squares decides -- the size-3 table kernel
    // (r04) and, r05, the size-5 tests of the level-3 position tables (the last formula of a test: two square roots and a division)
    constexpr bool SCR = TAB3 || (L1T && FW_L3_SCREEN != 0);
    // the accepted list in LDS.  TAB: |accepted| bounded by the host's routing; long-list variant: up to FZ_L1_A entries (the table
    // forms), longer lists are read from global memory by the generic gather form -- LDS is what bounds this variant's occupancy
    constexpr int ACC_LDS = TAB3 ? FZ_TAB_A : (HK ? FZ_HK_A + 8 : (L1T ? FZ_L1_A : FW_ACC_LDS));
    __shared__ int s_acc[ACC_LDS];
    __shared__ float4 s_l1[L1T ? FZ_L1_A : 1];         // {rho(X,v|z1), rho(Y,v|z1), cor[v][z1], sqrt(1 - cor[v][z1]^2)} per position v
    __shared__ unsigned char s_l1f[L1T ? FZ_L1_A : 1];  // Float32 flags of the first two
    __shared__ double s_l1a;                            // rho(X,Y|z1)
    __shared__ int s_l1af, s_l1nan;
    // level-3 position tables (L1T chunks, see "level-3 position tables" below)
    __shared__ float s3_p2[L1T ? FZ_L3_CAP : 1], s3_r2[L1T ? FZ_L3_CAP : 1];  // rho(w,z2|z1) and sqrt(1 - .^2) in Float32
    __shared__ unsigned char s3_fl[L1T ? FZ_L3_CAP : 1];
    // (one block: its 28 KB also stage the sorted ids of a target's local matrix in the prologue, before any table exists)
    __shared__ double s3_blk[L1T ? 8 * FZ_L3_CAP : 1];
    double *const s3_d2c = s3_blk, *const s3_q3 = s3_blk + (L1T ? FZ_L3_CAP : 0), *const s3_sq3 = s3_blk + (L1T ? 2 * FZ_L3_CAP : 0),
                  *const s3_x3 = s3_blk + (L1T ? 3 * FZ_L3_CAP : 0), *const s3_sx3 = s3_blk + (L1T ? 4 * FZ_L3_CAP : 0), *const s3_y3 = s3_blk + (L1T ? 5 * FZ_L3_CAP : 0),
                  *const s3_sy3 = s3_blk + (L1T ? 6 * FZ_L3_CAP : 0), *const s3_a4 = s3_blk + (L1T ? 7 * FZ_L3_CAP : 0);
    __shared__ int s3_off[L1T ? FZ_L3_DIR + 1 : 2], s3_low[L1T ? FZ_L3_DIR : 1], s3_jk[L1T ? FZ_L3_DIR : 1];
    __shared__ float s3_bp2[L1T ? FZ_L3_DIR : 1];  // block scalars of sub-block (z2, z3): rho(z3,z2|z1) ...
    __shared__ unsigned char s3_bfl[L1T ? FZ_L3_DIR : 1];
    __shared__ double s3_bd2c[L1T ? FZ_L3_DIR : 1], s3_bx2[L1T ? FZ_L3_DIR : 1], s3_bsx2[L1T ? FZ_L3_DIR : 1], s3_by2[L1T ? FZ_L3_DIR : 1],
        s3_bsy2[L1T ? FZ_L3_DIR : 1], s3_ba3[L1T ? FZ_L3_DIR : 1];
    __shared__ int s3_n, s3_lin0, s3_okf, s3_dirty;  // s3_dirty: a NaN or a Float64-literal level-1 value somewhere in the chunk's tables
    __shared__ unsigned long long s3_end;
    __shared__ double s_hk[HK ? FZ_HK_CAP : 1];
    __shared__ int s_hk_off[HK ? FZ_HK_DIR + 1 : 2];
    __shared__ unsigned short s_hk_ij[HK ? FZ_HK_DIR : 1];
    __shared__ unsigned long long s_hk_end;
    __shared__ int s_hk_n, s_hk_lin0, s_hk_nan, s_hk_skip0, s_hk_prev;
    __shared__ unsigned int s_cstop;  // smallest stopping rank of the current chunk so far (relative to the chunk), 0xffffffff = none
    __shared__ unsigned int s_scrq;   // cheap screen: bits of a Float32 upper bound of the smallest |stat|^2 this SEGMENT has evaluated so far in the normal range of p (FLT_MAX: none)
    __shared__ unsigned long long s_stop[4];
    __shared__ double s_bx[4], s_bps[4];
    __shared__ unsigned long long s_br[4];
    __shared__ unsigned int s_evc[4];  // tests really executed by each wavefront in the current chunk
    __shared__ double s_best_x, s_best_ps, s_best_stat;
    __shared__ unsigned long long s_best_rank;
    __shared__ float4 s_tab[TAB3 ? FZ_TAB_CAP : 1];    // {LX, LY, cor[v][z1], variable id | Float32 flags}
    __shared__ float s_tab_r1[TAB3 ? FZ_TAB_CAP : 1];   // sqrt(1 - cor[v][z1]^2)                       (Float32 roots)
    __shared__ float2 s_tab_r2[TAB3 ? FZ_TAB_CAP : 1];  // {sqrt(1 - LX^2), sqrt(1 - LY^2)}
    __shared__ double s_tab_a2[TAB3 ? FZ_TAB_CAP : 1]; // rho(X, Y | z1, v)
    __shared__ int s_blk[2];

#ifdef FW_FZ_FASTDBG
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long dbg_t1 = 0, dbg_tab = 0, dbg_loop = 0, dbg_red = 0;
#endif
    const int a = seg.acc_len;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool in_lds = a <= (TAB3 ? FZ_TAB_A : (HK ? FZ_HK_A : (L1T ? FZ_L1_A : FW_ACC_LDS)));
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
                *out_rec = o;
            }
            return;
        }
    } else if (in_lds) {
        if (remote_acc)
            for (int i = tid; i < a; i += 256) s_acc[i] = __hip_atomic_load(gacc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            for (int i = tid; i < a; i += 256) s_acc[i] = gacc[i];
    }
    // r06: the target's LOCAL correlation matrix (FwSeg::tm, fw_devhiton.hip: dh_tmat_build_kernel): every variable id this body uses -- X, Y,
    // the accepted list -- is replaced by its index in [T, T's level-0 neighbours ascending], `cor` / `p` by the (tm_m x tm_m) copy: the same
    // Float32 entries, gathered out of a few hundred KB that stay in L2 instead of the 400 MB matrix.  The sorted ids are staged in the
    // (not yet used) table array for the binary searches.  An id that is not on the list would be a bug of the caller: fail loudly.
    int Xl = -1, Yl = -1;
    bool tloc = false;  // this segment reads its target's local matrix (workgroup-uniform)
    if (!LOCAL && seg.tm != 0ull && (TAB3 || (FW_FZ_TMAT_HIGHK && ((HK && FW_FZ_TMAT_HIGHK != 2) || L1T) && in_lds))) {
        // staging: FZ_TAB_CAP float4 = 4096 ints / the level-2 tables' 32 KB / the level-3 tables' 28 KB (the host gives no matrix to a target with more than 4 095 neighbours)
        int32_t *s_ids = TAB3 ? (int32_t *)s_tab : (HK ? (int32_t *)s_hk : (int32_t *)s3_blk);
        if (!TAB3) tloc = true;
        const int32_t *ids = (const int32_t *)seg.tm_ids;
        const int nid = seg.tm_m - 1;
        for (int i = tid; i < nid; i += 256) s_ids[i] = ids[i];
        __syncthreads();
        auto local_of = [&](int v) -> int {
            if (v == seg.X) return 0;
            int lo = 0, hi = nid - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_ids[mid] < v)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            if (nid <= 0 || s_ids[lo] != v) __builtin_trap();
            return lo + 1;
        };
        for (int i = tid; i < a; i += 256) s_acc[i] = local_of(s_acc[i]);
        Xl = 0;
        Yl = local_of(seg.Y);
        cor = cor_g + (((long long)seg.tm - (long long)(unsigned long long)cor_g) >> 2);  // (= (const float *)seg.tm, derived from the kernel's global pointer: global_load, not flat_load)
        p = seg.tm_m;
        __syncthreads();  // (the table array is free again; the barrier below publishes s_acc)
    }
    if (tid == 0) {
        s_best_x = FZ_X_NONE;
        s_best_ps = 0.0;
        s_best_stat = 0.0;
        s_best_rank = 0;
        s_cstop = 0xffffffffu;
        s_scrq = 0x7f7fffffu;
        if (HK) s_hk_prev = -1;
    }
    unsigned long long cnt[FW_MAX_K_FAST + 1];
#pragma unroll
    for (int s = FW_MAX_K_FAST; s >= 1; --s)  // (size-3 table variant: |accepted| <= 512, sizes <= 3 -- 32-bit binomials, no 64-bit division in every workgroup's prologue)
        cnt[s] = (s <= max_k) ? (TAB3 ? (unsigned long long)fw_binom32(a, s) : binom_u64(a, s)) : 0ull;
    // significance thresholds on |r| (see fz_thresholds_kernel): outside [lo, hi] the verdict of p < alpha is certain
    const double rlo_pos = thr[0], rhi_pos = thr[1], rlo_neg = thr[2], rhi_neg = thr[3];
    // |r| below which x = |z|/sqrt2 < FZ_X_SUB for sure: x = zscale * log((1+r)/(1-r)) / sqrt2  <=>  r = tanh(x / (sqrt2 zscale))
    const double rsub_lo = !LOCAL ? thr[4] : (zscale > 0.0 ? tanh(FZ_X_SUB * 0.7071067811865476 / zscale) * (1.0 - 1e-9) : 2.0);
    // screened size-3 test (TAB3, see fz_l3_finish): with |stat| = |ev| / (sqrt(xb) sqrt(xc)) up to 1e-15 relative,
    //   ev^2 > h2 xb xc  =>  |stat| > rhi (significant for sure: rhi already sits 1e-9 above the true threshold),
    //   ev^2 < s2 xb xc  =>  |stat| < min(rsub_lo, 1): not clamped, p in the normal range (maximum-p tracking by |stat| alone)
    // (the context-wide thresholds carry them ready-made in thr[5..7]: fz_thresholds_kernel; per-job thresholds of fz_nz: computed here)
    const double h2_pos = !LOCAL ? thr[5] : rhi_pos * rhi_pos * (1.0 + 1e-12), h2_neg = !LOCAL ? thr[6] : rhi_neg * rhi_neg * (1.0 + 1e-12);
    const double s2 = !LOCAL ? thr[7] : (rsub_lo < 1.0 ? rsub_lo * rsub_lo : 1.0) * (1.0 - 1e-12);
    const float scr_h2 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)(h2_pos > h2_neg ? h2_pos : h2_neg) * 1.00001f)));  // cheap screen: the significance bound on |stat|^2 of either sign, rounded up
    __syncthreads();
#define ACCV(i) (LOCAL ? ((i) + 2) : (in_lds ? s_acc[(i)] : gacc[(i)]))
#define CORV(u, v) cor[(size_t)(u) * p + (v)]
// CORT(u, v): the entry a lane gathers where u is the index that moves from lane to lane.  r04 measured the transposed read
// cor[v][u] here (the matrix is exactly symmetric, and the 64 gathers of a wavefront then fall into one row instead of 64 rows):
// neutral at cfg3 (193.2 against 191.8 ms), 3.7 % SLOWER at cfg5 (65.3 against 63.0 s on the same box, p = 100 000: a row is 400 KB,
// the lines of 64 scattered columns of one row conflict in the L2 sets where 64 rows at one column do not) -- kept off.
// -DFW_CORT_TRANSPOSED builds it.
// r06: the size-3 table kernel reads the transposed entry: with a target's LOCAL matrix (FwSeg::tm) a row is at most a few hundred floats, so the
// 64 gathers of a wavefront step -- one z2, 64 different z3 -- fall into a dozen cache lines of ONE row instead of 64 lines of 64 rows, and the
// step stops being bound by the rate at which the CU's texture path looks up distinct lines (FW_CORT_TAB3=0: the r05 order; on the p x p
// matrix, where a row is 40 KB, either order touches ~64 lines)
#ifndef FW_CORT_TAB3
#define FW_CORT_TAB3 1
#endif
#ifdef FW_CORT_TRANSPOSED
#define CORT(u, v) (LOCAL ? CORV(u, v) : CORV(v, u))
#else
// r06, max_k 4-5 with a local matrix (tloc, workgroup-uniform): transposed as well -- a lane's run of tests walks ALONG a row of a few hundred floats
#define CORT(u, v) ((FW_CORT_TAB3 && TAB3 && !LOCAL) ? CORV(v, u) : ((FW_FZ_TMAT_HIGHK && HIGHK && !LOCAL) ? cor[(size_t)(tloc ? (v) : (u)) * p + (tloc ? (u) : (v))] : CORV(u, v)))
#endif

    const int X = LOCAL ? 0 : (Xl >= 0 ? Xl : seg.X), Y = LOCAL ? 1 : (Yl >= 0 ? Yl : seg.Y);
    const float cXY = CORV(X, Y);
    const unsigned long long NONE = FW_RANK_NONE;
    const unsigned long long len = seg.end - seg.start;
    constexpr int RMAX = (TAB && !HIGHK) ? FW_RUN_MAX3 : FW_RUN_MAX;
    const int R = (int)((len + 255) / 256 < RMAX ? (len + 255) / 256 : RMAX);
    unsigned long long evaluated = 0;

    unsigned long long cnext = seg.start;
    for (unsigned long long cbase = seg.start; cbase < seg.end; cbase = cnext) {
#ifdef FW_FZ_FASTDBG
        const unsigned long long dbg_c0 = __builtin_amdgcn_s_memtime();
        if (!dbg_t1) dbg_t1 = dbg_c0;
#endif
        unsigned long long cend = cbase + 256ull * R;
        cend = cend < seg.end ? cend : seg.end;
        int Rc = R;
        // ---- level-2 tables of the (z1, z2) sub-blocks this chunk touches (see FZ_HK_A); the chunk ends where the tables
        // are full, where the size changes, or after 256 * FW_RUN_MAX ranks ----
        bool hk_ok = false, hk_nn = false;
        int hk_s = 0, hk_lin0 = 0;
        if (HK) {
            unsigned long long rem0 = cbase;
            int s0 = max_k;
            while (s0 > 1 && rem0 >= cnt[s0]) {  // workgroup-uniform
                rem0 -= cnt[s0];
                --s0;
            }
            if (s0 >= 4) {
                if (tid == 0) {
                    int q[FW_MAX_K_FAST];
                    fw_unrank_comb32((uint32_t)rem0, a, s0, q);  // a <= FZ_HK_A: everything fits 32 bits
                    int i = q[0], j = q[1];
                    // rank (inside the size-s0 enumeration) behind the last subset of sub-block (i, j)
                    unsigned long long cur_end = (unsigned long long)(fw_binom32(a, s0) - fw_binom32(a - i, s0)) +
                                                 (unsigned long long)(fw_binom32(a - 1 - i, s0 - 1) - fw_binom32(a - j, s0 - 1)) +
                                                 (unsigned long long)fz_hk_tests(a - 1 - j, s0 - 2);
                    unsigned long long lim = rem0 + 256ull * FW_RUN_MAX;  // chunk end (inside the size-s0 enumeration)
                    if (lim > rem0 + (seg.end - cbase)) lim = rem0 + (seg.end - cbase);
                    if (lim > cnt[s0]) lim = cnt[s0];
                    const int ij0 = (s0 << 16) | (i << 8) | j;
                    s_hk_skip0 = (ij0 == s_hk_prev) ? 1 : 0;  // the previous chunk's first table is this chunk's first table
                    s_hk_prev = ij0;
                    s_hk_lin0 = i * (a - s0 + 1) - i * (i - 1) / 2 + (j - i - 1);
                    int nd = 0, eoff = 0;
                    for (;;) {
                        const int e = fz_hk_entries(a - 1 - j);
                        if (nd > 0 && eoff + e > FZ_HK_CAP) {  // tables full: the chunk ends in front of this sub-block
                            lim = cur_end - (unsigned long long)fz_hk_tests(a - 1 - j, s0 - 2);
                            break;
                        }
                        s_hk_ij[nd] = (unsigned short)((i << 8) | j);
                        s_hk_off[nd] = eoff;
                        eoff += e;
                        ++nd;
                        if (cur_end >= lim) break;
                        if (nd == FZ_HK_DIR) {
                            lim = cur_end;
                            break;
                        }
                        if (++j > a - s0 + 1) {  // first sub-block of the next z1-block
                            ++i;
                            j = i + 1;
                        }
                        cur_end += (unsigned long long)fz_hk_tests(a - 1 - j, s0 - 2);
                    }
                    s_hk_off[nd] = eoff;
                    s_hk_n = nd;
                    s_hk_end = cbase + (lim - rem0);
                    if (!s_hk_skip0) s_hk_nan = 0;
                }
                __syncthreads();
                const int nd = s_hk_n, etot = s_hk_off[nd];
                cend = s_hk_end;
                hk_ok = true;
                hk_s = s0;
                hk_lin0 = s_hk_lin0;
                for (int e = (s_hk_skip0 ? s_hk_off[1] : 0) + tid; e < etot; e += 256) {
                    int d = 0;
                    for (int w = 64; w > 0; w >>= 1)  // largest d with off[d] <= e (FZ_HK_DIR <= 127)
                        if (d + w < nd && s_hk_off[d + w] <= e) d += w;
                    const int ij = s_hk_ij[d], i = ij >> 8, j = ij & 255, n = a - 1 - j;
                    int l = e - s_hk_off[d];
                    int A, B;  // the pair, A before B in U = [X, Y, later positions descending]
                    if (l == 0) {
                        A = X;
                        B = Y;
                    } else if (l <= 2 * n) {
                        A = l <= n ? X : Y;
                        B = ACCV(j + 1 + (l <= n ? l - 1 : l - 1 - n));
                    } else {
                        l -= 1 + 2 * n;
                        int pl = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)l)) * 0.5f);
                        while (pl * (pl - 1) / 2 > l) --pl;
                        while ((pl + 1) * pl / 2 <= l) ++pl;
                        A = ACCV(j + 1 + pl);
                        B = ACCV(j + 1 + (l - pl * (pl - 1) / 2));
                    }
                    const int z1 = ACCV(i), z2 = ACCV(j);
                    const float cAz1 = CORV(A, z1), cBz1 = CORV(B, z1), cz2z1 = CORV(z2, z1);
                    const TV RAB = pc_l1(CORV(A, B), cAz1, cBz1);
                    const TV RA2 = pc_l1(CORV(A, z2), cAz1, cz2z1);
                    const TV RB2 = pc_l1(CORV(B, z2), cBz1, cz2z1);
                    const double v = pc_l2(RAB, RA2, RB2);
                    s_hk[e] = v;
                    if (!(v == v)) s_hk_nan = 1;
                }
                __syncthreads();
                hk_nn = s_hk_nan == 0;
                const unsigned long long clen = cend - cbase;
                Rc = (int)((clen + 255ull) / 256ull);
            }
        }
        bool l1_ok = false, l1_clean = false, l3_ok = false, l3_nn = false;
        int l1_s = 0, l3_lin0 = 0, l3_i = 0;
        if (L1T && in_lds && a <= FZ_L1_A && !(fz_dbg_flags & 1)) {
            unsigned long long rem0 = cbase;
            int s0 = max_k;
            while (s0 > 1 && rem0 >= cnt[s0]) {  // workgroup-uniform
                rem0 -= cnt[s0];
                --s0;
            }
            if (s0 >= 4 && rem0 + (cend - cbase) <= cnt[s0]) {  // the whole chunk holds subsets of s0 variables
                if (tid == 0) s_l1nan = 0;
                if (tid == 0 || tid == 64) {
                    const unsigned long long rr = tid == 0 ? rem0 : rem0 + (cend - cbase) - 1ull;
                    s_blk[tid == 0 ? 0 : 1] = a - fw_inv_binom(cnt[s0] - rr, s0, a);  // first position of that rank
                }
                __syncthreads();
                const int i0 = s_blk[0], i1 = s_blk[1];
                if (i0 == i1) {  // ... inside one z1-block
                    l1_ok = true;
                    l1_s = s0;
                    const int z1 = ACCV(i0);
                    const float cXz1 = CORV(X, z1), cYz1 = CORV(Y, z1);
                    for (int v = i0 + 1 + tid; v < a; v += 256) {
                        const int zv = ACCV(v);
                        const float cvz1 = CORT(zv, z1);
                        const TV LX = pc_l1(CORV(X, zv), cXz1, cvz1);
                        const TV LY = pc_l1(CORV(Y, zv), cYz1, cvz1);
                        s_l1[v] = make_float4((float)LX.v, (float)LY.v, cvz1, sqrtf(1.0f - cvz1 * cvz1));
                        s_l1f[v] = (unsigned char)((LX.f32 ? 1 : 0) | (LY.f32 ? 2 : 0));
                        if (!(LX.v == LX.v && LY.v == LY.v && cvz1 == cvz1)) s_l1nan = 1;
                    }
                    if (tid == 0) {
                        const TV A1 = pc_l1(cXY, cXz1, cYz1);
                        s_l1a = A1.v;
                        s_l1af = A1.f32 ? 1 : 0;
                        if (!(A1.v == A1.v)) s_l1nan = 1;
                    }
                }
                __syncthreads();
                l1_clean = s_l1nan == 0;
                // ---- level-3 position tables of the (z2, z3) sub-blocks this chunk touches; the chunk ends where they are full ----
                if (l1_ok && !(fz_dbg_flags & 2)) {
                    const int i = i0;
                    if (tid == 0) {
                        int q[FW_MAX_K_FAST];
                        unrank_comb(rem0, a, s0, q);
                        int j = q[1], k = q[2];
                        const int l0 = q[3];
                        const int t3 = s0 - 3;  // positions behind z3: C(n, t3) subsets per sub-block
                        unsigned long long cur_end = (binom_u64(a, s0) - binom_u64(a - i, s0)) + (binom_u64(a - 1 - i, s0 - 1) - binom_u64(a - j, s0 - 1)) +
                                                     (binom_u64(a - 1 - j, s0 - 2) - binom_u64(a - k, s0 - 2)) + binom_u64(a - 1 - k, t3);
                        unsigned long long lim = rem0 + (cend - cbase);
                        int nd = 0, eoff = 0, okf = 1;
                        const int jfirst = i + 1;
                        s3_lin0 = (j - jfirst) * (a - s0 + 1 - i) - (j - jfirst) * (j - jfirst - 1) / 2 + (k - j - 1);
                        for (;;) {
                            const int lo_w = nd == 0 ? l0 : k + 1;
                            const int ne = a - lo_w;
                            if (eoff + ne > FZ_L3_CAP) {
                                if (nd == 0) {  // one sub-block alone does not fit: this chunk takes the level-1 form
                                    okf = 0;
                                    break;
                                }
                                lim = cur_end - binom_u64(a - 1 - k, t3);  // the chunk ends in front of this sub-block
                                break;
                            }
                            s3_jk[nd] = (j << 16) | k;
                            s3_off[nd] = eoff;
                            s3_low[nd] = lo_w;
                            eoff += ne;
                            ++nd;
                            if (cur_end >= lim) break;
                            if (nd == FZ_L3_DIR) {
                                lim = cur_end;
                                break;
                            }
                            if (++k > a - s0 + 2) {  // first sub-block of the next z2 (still inside the z1-block: lim is)
                                ++j;
                                k = j + 1;
                            }
                            cur_end += binom_u64(a - 1 - k, t3);
                        }
                        s3_off[nd] = eoff;
                        s3_n = nd;
                        s3_okf = okf;
                        s3_dirty = 0;
                        s3_end = cbase + (lim - rem0);
                    }
                    __syncthreads();
                    if (s3_okf) {
                        l3_ok = true;
                        l3_lin0 = s3_lin0;
                        l3_i = i;
                        cend = s3_end;
                        Rc = (int)((cend - cbase + 255ull) / 256ull);
                        const int nd = s3_n, etot = s3_off[nd];
                        const int z1 = ACCV(i);
                        (void)z1;
                        const TV A1{s_l1a, s_l1af != 0};
                        // block scalars: one thread per sub-block
                        for (int d = tid; d < nd; d += 256) {
                            const int j = s3_jk[d] >> 16, k = s3_jk[d] & 0xffff;
                            const float4 e2 = s_l1[j], e3 = s_l1[k];
                            const int f2 = s_l1f[j], f3 = s_l1f[k];
                            const TV LXz2{(double)e2.x, (f2 & 1) != 0}, LYz2{(double)e2.y, (f2 & 2) != 0};
                            const TV LXz3{(double)e3.x, (f3 & 1) != 0}, LYz3{(double)e3.y, (f3 & 2) != 0};
                            const TV P2z3 = pc_l1_r(CORV(ACCV(k), ACCV(j)), e3.z, e2.z, e3.w, e2.w);  // rho(z3,z2|z1)
                            const double dP = fz_sq1(P2z3.v);
                            const double A2 = pc_l2(A1, LXz2, LYz2);           // rho(X,Y|z1,z2)
                            const double X2 = pc_l2_d2(LXz3, LXz2, P2z3, dP);  // rho(X,z3|z1,z2)
                            const double Y2 = pc_l2_d2(LYz3, LYz2, P2z3, dP);  // rho(Y,z3|z1,z2)
                            const double sx = fz_sq1(X2), sy = fz_sq1(Y2);
                            s3_bp2[d] = (float)P2z3.v;
                            s3_bfl[d] = P2z3.f32 ? 1 : 0;
                            s3_bd2c[d] = dP;
                            s3_bx2[d] = X2;
                            s3_bsx2[d] = sx;
                            s3_by2[d] = Y2;
                            s3_bsy2[d] = sy;
                            const double A3 = pc_l3s(A2, X2, Y2, sx, sy);      // rho(X,Y|z1,z2,z3)
                            s3_ba3[d] = A3;
                            if (!(P2z3.f32 && X2 == X2 && Y2 == Y2 && A3 == A3)) s3_dirty = 1;
                        }
                        __syncthreads();
                        for (int e = tid; e < etot; e += 256) {
                            int d = 0;
                            for (int w = 32; w > 0; w >>= 1)  // largest d with off[d] <= e (FZ_L3_DIR <= 64)
                                if (d + w < nd && s3_off[d + w] <= e) d += w;
                            const int j = s3_jk[d] >> 16, k = s3_jk[d] & 0xffff;
                            const int wv = s3_low[d] + (e - s3_off[d]);
                            const int zw = ACCV(wv), z2 = ACCV(j), z3 = ACCV(k);
                            const float4 ew = s_l1[wv], e2 = s_l1[j], e3 = s_l1[k];
                            const int fw_ = s_l1f[wv], f2 = s_l1f[j];
                            const TV LXw{(double)ew.x, (fw_ & 1) != 0}, LYw{(double)ew.y, (fw_ & 2) != 0};
                            const TV LXz2{(double)e2.x, (f2 & 1) != 0}, LYz2{(double)e2.y, (f2 & 2) != 0};
                            const TV P2z3{(double)s3_bp2[d], s3_bfl[d] != 0};
                            const TV P2w = pc_l1_r(CORT(zw, z2), ew.z, e2.z, ew.w, e2.w);  // rho(w,z2|z1)
                            const TV P3w = pc_l1_r(CORT(zw, z3), ew.z, e3.z, ew.w, e3.w);  // rho(w,z3|z1)
                            const double dw = fz_sq1(P2w.v);
                            const double Q3 = pc_l2_d2(P3w, P2w, P2z3, s3_bd2c[d]);  // rho(w,z3|z1,z2): w is the first of the pair
                            const double X2w = pc_l2_d2(LXw, LXz2, P2w, dw);        // rho(X,w|z1,z2)
                            const double Y2w = pc_l2_d2(LYw, LYz2, P2w, dw);
                            const double sq = fz_sq1(Q3);
                            const double X3 = pc_l3s(X2w, s3_bx2[d], Q3, s3_bsx2[d], sq);  // rho(X,w|z1,z2,z3)
                            const double Y3 = pc_l3s(Y2w, s3_by2[d], Q3, s3_bsy2[d], sq);
                            const double sx = fz_sq1(X3), sy = fz_sq1(Y3);
                            const float p2f = (float)P2w.v;
                            s3_p2[e] = p2f;
                            s3_r2[e] = sqrtf(1.0f - p2f * p2f);  // the d1 of a level-2 formula whose first conditioning value is this one
                            s3_d2c[e] = dw;
                            s3_q3[e] = Q3;
                            s3_sq3[e] = sq;
                            s3_x3[e] = X3;
                            s3_sx3[e] = sx;
                            s3_y3[e] = Y3;
                            s3_sy3[e] = sy;
                            const double A4 = pc_l3s(s3_ba3[d], X3, Y3, sx, sy);  // rho(X,Y|z1,z2,z3,w): the size-4 statistic
                            s3_a4[e] = A4;
                            // bit 0: rho(w,z2|z1) is still a Float32 value; bit 1 (r05): this ENTRY is clean -- Float32-typed and no NaN in it.  A
                            // size-5 test whose two entries are clean takes the NaN-free arithmetic also in a chunk that holds an unclean
                            // entry somewhere else (cfg5: 40 % of the size-5 tests sat in such chunks and took the NaN-preserving form
                            // with its IEEE divisions for one entry in ~900)
                            const bool e_nonan = P2w.v == P2w.v && Q3 == Q3 && X3 == X3 && Y3 == Y3 && A4 == A4;  // bit 2: no NaN in it (Float32-typed or not)
                            const bool e_clean = P2w.f32 && e_nonan;
                            s3_fl[e] = (unsigned char)((P2w.f32 ? 1 : 0) | (e_clean ? 2 : 0) | (e_nonan ? 4 : 0));
                            if (!e_clean) s3_dirty = 1;
                        }
                        __syncthreads();
                        l3_nn = s3_dirty == 0 && l1_clean;
                    }
                }
            }
        }
        // ---- size-3 table: the z1-blocks [i0, i1] this chunk touches; a chunk whose blocks do not fit the table is halved (FW_RUN_MAX3) ----
        bool tab_ok = false;
        int tb_i0 = 0, tb_E = 0;
        if (TAB3) {
            const unsigned long long c3 = (max_k >= 3) ? cnt[3] : 0ull;
            if (cbase < c3) {  // workgroup-uniform; a <= FZ_TAB_A by the host's routing
                for (;;) {
                    unsigned long long last3 = cend;
                    last3 = (last3 < c3 ? last3 : c3) - 1ull;
                    if (tid == 0 || tid == 64) {
                        int q[FW_MAX_K_FAST];
                        fw_unrank_comb32((uint32_t)(tid == 0 ? cbase : last3), a, 3, q);  // a <= FZ_TAB_A: 32-bit form
                        s_blk[tid == 0 ? 0 : 1] = q[0];
                    }
                    __syncthreads();
                    const int i0 = s_blk[0], i1 = s_blk[1];
                    tb_E = fz_tab_off(i1 + 1, i0, a);  // <= 1021 for a <= 512 and chunks of 8192 ranks
                    tb_i0 = i0;
                    if (tb_E <= FZ_TAB_CAP) break;
                    if (Rc <= FW_RUN_MAX) __builtin_trap();  // would be a routing bug on the host side: fail loudly
                    __syncthreads();                         // (s_blk is rewritten)
                    Rc = (Rc + 1) / 2;
                    cend = cbase + 256ull * Rc;
                }
                tab_ok = true;
            }
        }
        cnext = cend;
        // Lane <-> rank mapping.  Runs: lane l of the workgroup takes Rc consecutive ranks (one unranking, unit steps).
        // Interleaved (table kernel, chunk entirely inside the size-3 enumeration): wavefront w owns the 64 Rc consecutive
        // ranks behind cbase + w 64 Rc and lane l takes every 64th of them -- in step t the wavefront tests 64 consecutive
        // ranks, so a stop in step t ends the whole wavefront (all later steps hold later ranks): no lanes running on in
        // a half-empty wavefront behind a stop, at the price of a step of 64 ranks (a short carry loop) instead of 1.
        const bool ilv = TAB3 && FW_FZ_INTERLEAVE && max_k >= 3 && cend <= cnt[3];  // workgroup-uniform
        const unsigned long long r0 = ilv ? cbase + (unsigned long long)wave * 64ull * Rc + (unsigned long long)lane
                                          : cbase + (unsigned long long)tid * Rc;
        unsigned long long r1 = ilv ? cbase + (unsigned long long)(wave + 1) * 64ull * Rc : r0 + Rc;
        if (r1 > cend) r1 = cend;
        const unsigned long long rstep = ilv ? 64ull : 1ull;
        const bool any = r0 < r1;
        // ---- table of the z1-blocks this chunk touches (see FZ_TAB_A) ----
        if (TAB3 && tab_ok) {
            {
                {
                    const int i0 = tb_i0, E = tb_E;
                    for (int e = tid; e < E; e += 256) {
                        int i = i0, rem = e;
                        while (rem >= a - 1 - i) {
                            rem -= a - 1 - i;
                            ++i;
                        }
                        const int z1 = ACCV(i), zv = ACCV(i + 1 + rem);
                        const float cXz1 = CORV(X, z1), cYz1 = CORV(Y, z1), cvz1 = CORT(zv, z1);
                        const TV A1 = pc_l1(cXY, cXz1, cYz1);
                        const TV LX = pc_l1(CORV(X, zv), cXz1, cvz1);
                        const TV LY = pc_l1(CORV(Y, zv), cYz1, cvz1);
                        const double a2 = pc_l2(A1, LX, LY);
                        s_tab_a2[e] = a2;
                        const bool clean = LX.v == LX.v && LY.v == LY.v && a2 == a2;  // no NaN in this entry
                        // square roots the level-1 / level-2 formulas take of this entry's values (statfuns.jl:36,52);
                        // for a Float64-literal LX / LY (0, +-1) the Float32 root is the exact one as well
                        const float fx = (float)LX.v, fy = (float)LY.v;
                        s_tab_r1[e] = sqrtf(1.0f - cvz1 * cvz1);
                        s_tab_r2[e] = make_float2(sqrtf(1.0f - fx * fx), sqrtf(1.0f - fy * fy));
                        s_tab[e] = make_float4((float)LX.v, (float)LY.v, cvz1,
                                               __int_as_float(zv | (LX.f32 ? (1 << 30) : 0) | (LY.f32 ? (1 << 29) : 0) |
                                                              (clean ? (1 << 28) : 0)));
                    }
                }
                __syncthreads();
            }
        }
#ifdef FW_FZ_FASTDBG
        const unsigned long long dbg_c1 = __builtin_amdgcn_s_memtime();
#endif
        // lane-local results
        unsigned long long my_stop = NONE, my_br = 0;
        double stop_stat = 0.0, stop_p = 0.0, my_bstat = 0.0;
        // lane best: ordered by x = |z|/sqrt2 ascending (= p descending); in the underflow regime (x > FZ_X_SUB, where
        // different x can give the same subnormal/zero p) by the exact p instead; later rank wins ties (tests.jl:338)
        double my_bx = FZ_X_NONE, my_bps = 0.0, my_ba = 0.0;
        // TAB3: the lane best as (numerator, radicands) -- |stat|^2 = my_bev^2 / (my_bxb my_bxc); its statistic is
        // fz_l3_finish(my_bev, my_bxb, my_bxc), or my_bev itself when my_bxb = my_bxc = 1 (tests whose quotient was taken)
        double my_bev = 0.0, my_bxb = 1.0, my_bxc = 1.0;
        unsigned int my_done = 0;  // tests this lane executes in this chunk (it leaves its run at its first stop)
#if defined(FW_FZ_FASTDBG) && FW_FZ_FASTDBG >= 4
        double dbg_wmin = 1.0e300;
#endif
#if defined(FW_FZ_FASTDBG) && FW_FZ_FASTDBG >= 5
        bool dbg_sok = false;
        float dbg_sT = 0.0f;
#endif
        if (any) {
            // unrank the first rank of the run
            unsigned long long rem = r0;
            int s = max_k;
            while (s > 1 && rem >= cnt[s]) {
                rem -= cnt[s];
                --s;
            }
            int pos[FW_MAX_K_FAST];
#pragma unroll
            for (int q = 0; q < FW_MAX_K_FAST; ++q) pos[q] = 0;
            if (TAB || (L1T && a <= FW_UNRANK32_A5))  // |accepted| <= FZ_TAB_A with max_k <= 3, or <= 128 with max_k <= 5: the 32-bit unranking (fw_unrank.h)
                fw_unrank_comb32((uint32_t)rem, a, s, pos);
            else
                unrank_comb(rem, a, s, pos);
            int chg = 0;  // lowest position index that changed since the previous test of this lane (0 = everything)
            // cached state for s <= 3
            int z1 = 0, z2 = 0;
            float cXz1 = 0.f, cYz1 = 0.f, cXz2 = 0.f, cYz2 = 0.f, cz2z1 = 0.f;
            TV A1{0.0, false}, B1{0.0, false}, C1{0.0, false};
            double A2 = 0.0;
            int boff = 0;
            float pf_c32 = 0.0f;  // fast loop: the matrix entry of THIS iteration's test, fetched during the previous one
            const unsigned int cbase32 = (unsigned int)cbase, r1rel = (unsigned int)r1 - (unsigned int)cbase;
            const bool chunk_capped = max_tests > 0 && cend >= (unsigned long long)max_tests;  // workgroup-uniform
            const bool p32 = p <= 46000;                                                      // p^2 < 2^31: 32-bit element index
            bool pf_ok = false;
            const double *hk_tb = s_hk;
            int hk_j = 0;
            float4 tj = make_float4(0.f, 0.f, 0.f, 0.f);
            float rj1 = 0.f;
            float2 rj2 = make_float2(0.f, 0.f);
            double A2j = 0.0;
            // level-3 position tables: the z4 entry of the running sub-block
            int l3_base = 0, zl = 0;
#if FW_FZ_L1T_ROW32
            int zl_row = 0;
#endif
            float4 t1l = make_float4(0.f, 0.f, 0.f, 0.f);
            float p2l = 0.f;
            bool fl3 = false, cl3 = false, nn3 = false;
            double d2cl = 0.0, q3l = 0.0, sq3l = 0.0, x3l = 0.0, sx3l = 0.0, y3l = 0.0, sy3l = 0.0, a4l = 0.0;
            for (unsigned long long r = r0; r < r1; r += rstep) {
                double stat;
                double l3_ev = 0.0, l3_xb = 1.0, l3_xc = 1.0;
                bool screened = false;  // TAB3 fast path: stat is still (l3_ev, l3_xb, l3_xc)
                ++my_done;
                if (TAB3 && FW_FZ_FASTLOOP && ilv && tab_ok) {
                    // (workgroup-uniform: the whole chunk lies in the size-3 enumeration and has its table.)  r05: the common test in one
                    // piece -- every entry clean, the test "significant for sure" on the squares, no stop in front of it, and the
                    // maximum-p bookkeeping decided without a quotient (clearly larger: nothing; clearly smaller or nothing yet: taken
                    // by selects).  When that holds for EVERY lane of the wavefront the iteration ends here behind ONE scalar branch;
                    // otherwise nothing has been committed and the general code below runs the test again with all its cases (same
                    // values: the same functions on the same operands).  In the general form each of those cases is a v_cmp + exec
                    // mask + branch per test: ~170 of the ~290 vector instructions of a test (profiles/r05_cfg3_pmc_summary.json:
                    // 2.06 of 4.50 VALU per test are moves / compares / selects).
                    // (Two tests per iteration -- this rank and the one 64 further, two independent chains -- measured slower: 175.6 against
                    // 167.8 ms on one box, profiles/r05_cfg3_fast_loop.txt.)
                    const int pi = pos[0];
                    if (chg <= 0) boff = fz_tab_off(pi, tb_i0, a) - pi - 1;
                    const int ej = boff + pos[1];
                    tj = s_tab[ej];
                    rj1 = s_tab_r1[ej];
                    rj2 = s_tab_r2[ej];
                    A2j = s_tab_a2[ej];
                    const int ek = boff + pos[2];
                    const float4 tk = s_tab[ek];
                    const float rk1 = s_tab_r1[ek];
                    const int fj = __float_as_int(tj.w), fk = __float_as_int(tk.w);
                    // (the gathers of the fast loop index the matrix with 32 bits where p^2 allows it -- always with a local matrix: one multiply-add
                    // and a scalar base instead of a 64-bit multiply and two 64-bit adds per gather)
#define CORT32(u, v) (p32 ? ((FW_CORT_TAB3 && !LOCAL) ? cor[(unsigned int)(v) * (unsigned int)p + (unsigned int)(u)] : cor[(unsigned int)(u) * (unsigned int)p + (unsigned int)(v)]) : CORT(u, v))
                    float c32 = pf_c32;  // fetched during the previous iteration (below); a lane's first iteration behind the general form: now
                    if (!pf_ok) {
                        c32 = CORT32(fk & FZ_TAB_ZMASK, fj & FZ_TAB_ZMASK);
                        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) HERE: behind the join the compiler would otherwise wait for everything in flight, the next entry included
                    }
                    // the next test of this lane (64 ranks ahead): its position, and its matrix entry on the way while this test is
                    // evaluated -- the gather (two table words, then the entry) is otherwise the head of every test's dependent chain
                    const unsigned int rrel = (unsigned int)r - cbase32;  // the rank relative to the chunk (a chunk holds at most 16 384 ranks)
                    const bool hasN = rrel + 64u < r1rel;
                    int ni = pos[0], nj = pos[1], nk = pos[2] + (hasN ? 64 : 0), nchg = 2;
                    while (nk > a - 1) {  // row (i, j) holds k = j + 1 .. a - 1: carry the overflow into the next rows
                        const int over = nk - a;
                        if (++nj > a - 2) {
                            ++ni;
                            nj = ni + 1;
                            nchg = 0;
                        } else if (nchg > 1) {
                            nchg = 1;
                        }
                        nk = nj + 1 + over;
                    }
                    int nboff = boff;
                    if (nchg <= 0) nboff = fz_tab_off(ni, tb_i0, a) - ni - 1;
                    const int nfj = __float_as_int(s_tab[nboff + nj].w), nfk = __float_as_int(s_tab[nboff + nk].w);  // (in range also without a next test: this test's row)
                    if (hasN) pf_c32 = CORT32(nfk & FZ_TAB_ZMASK, nfj & FZ_TAB_ZMASK);  // straight into the loop-carried register: a copy at the end of the iteration would wait for it
#undef CORT32
                    const bool f_nostop = !(__hip_atomic_load(&s_cstop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < rrel);
                    bool f_capped = false;
                    if (chunk_capped) f_capped = r + 1 >= (unsigned long long)max_tests;  // (wave-uniform branch: only the chunk that holds the cap looks)
#if FW_FZ_SCREEN
                    // ---- cheap screen (r06) ----
                    // A test that is (a) significant for sure, (b) not the last test of a capped job, (c) not behind a stop and (d) clearly
                    // LARGER in |stat| than a test this segment has already evaluated in the normal range of p can neither end the job nor be
                    // (or tie) its maximum-p test: its value is never needed (profiles/r06_cfg3_cheap_screen_potential.txt: true of every lane in
                    // 77-81 % of cfg3's fast-loop iterations, whatever the margin between 1e-4 and 1e-2).  Here that is DECIDED, conservatively,
                    // from a Float32 evaluation without the three round5 steps, the square root and the three correctly rounded divisions of the
                    // exact sequence, with an error bound that covers every difference between the two.  With
                    //   F = e1 / d1,  g = 1 - F^2,  nD = LXk - LXj F,  nE = LYk - LYj F        (e1 = c32 - ck cj, d1 = rk1 rj1: level 1, statfuns.jl:36)
                    //   Pb = rjx^2 g - nD^2,  Pc = rjy^2 g - nE^2,  Num = A2j rjx rjy g - nD nE    (rjx = sqrt(1 - LXj^2), ...: level 2 multiplied out)
                    // the exact sequence's |stat|^2 = ev^2 / (xb xc) equals Num^2 / (Pb Pc) in real arithmetic when no clamp acts.  Differences: round5 moves
                    // e1, nD, nE and ev by <= 5e-6 each; rcp and the Float32 operations by < 3e-7 relative each.  Propagated (|values| <= 1, |nD|, |nE| <= 2):
                    //   |dF| <= aF = 8e-6 / d1 + 4e-6,  |dg| <= 2 aF + 2e-6,  |dnD|, |dnE| <= 1e-5 + aF,  |dPb|, |dPc|, |dNum| <= dl = 6 aF + 5e-5 = 4.8e-5 / d1 + 7.4e-5
                    // (guards: d1 > 0.01 so that aF < 1e-3; g > 0.02 so that F is not clamped; Pb, Pc > dl so that D2, E2 are not clamped and the radicands positive).
                    // Then |stat|^2 >= (|Num| - dl)^2 / ((Pb + dl)(Pc + dl)), compared -- without a division -- with the larger of the significance bound and the
                    // segment's smallest |stat|^2 so far (s_scrq, an upper bound of it, published by the exact path below), times 1.0002 for the roundings of the
                    // comparison itself.  Anything else -- and every NaN: all comparisons are written to fail on one -- takes the exact path.
                    {
                        const unsigned int sq_bits = __hip_atomic_load(&s_scrq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (sq_bits != 0x7f7fffffu) {  // (workgroup-uniform up to timing; wave-uniform: one LDS word)
                            const float scr_T = fmaxf(__uint_as_float(sq_bits), scr_h2) * 1.0002f;
                            const float e1s = c32 - tk.z * tj.z;
                            const float d1s = rk1 * rj1;
                            const float inv1 = __builtin_amdgcn_rcpf(d1s);
                            const float Fs = e1s * inv1;
                            const float dl = fmaf(4.8e-5f, inv1, 7.4e-5f);
                            const float gs = fmaf(-Fs, Fs, 1.0f);
                            const float nDs = fmaf(-tj.x, Fs, tk.x), nEs = fmaf(-tj.y, Fs, tk.y);
                            const float rg = rj2.x * gs;
                            const float Pb = fmaf(-nDs, nDs, rj2.x * rg), Pc = fmaf(-nEs, nEs, rj2.y * rj2.y * gs);
                            const float Nm = fmaf(-nDs, nEs, (float)A2j * rg * rj2.y);
                            const float Nlo = fabsf(Nm) - dl;
                            const bool s_ok = (((fj & fk) >> 28) & 7) == 7 && d1s > 0.01f && gs > 0.02f && Pb > dl && Pc > dl && Nlo > 0.0f &&
                                              Nlo * Nlo > scr_T * ((Pb + dl) * (Pc + dl)) && !f_capped && f_nostop;
#if defined(FW_FZ_FASTDBG) && FW_FZ_FASTDBG >= 5
                            dbg_sok = s_ok;
                            dbg_sT = scr_T;
#else
                            if (__all(s_ok)) {
                                if (!hasN) break;
                                pos[0] = ni;
                                pos[1] = nj;
                                pos[2] = nk;
                                chg = nchg;
                                boff = nboff;
                                pf_ok = true;
                                continue;
                            }
#endif
                        }
                    }
#endif
                    bool f1ok;
                    const float F1f = pc_l1_rf(c32, tk.z, tj.z, rk1, rj1, f1ok);
                    const double F1v = (double)F1f;
                    const double dF = fz_sqrt_unit(1.0 - F1v * F1v);
                    const bool clean = (((fj & fk) >> 28) & 7) == 7 && f1ok;
                    const double D2 = pc_l2_all32_d1_nn(tk.x, tj.x, F1f, (double)rj2.x, dF);
                    const double E2 = pc_l2_all32_d1_nn(tk.y, tj.y, F1f, (double)rj2.y, dF);
                    const double f_ev = round5_f64_nn(A2j - D2 * E2), f_xb = 1.0 - D2 * D2, f_xc = 1.0 - E2 * E2;
                    const double f_m2 = f_xb * f_xc, f_e2 = f_ev * f_ev;
                    const bool f_sure = f_e2 > (f_ev < 0.0 ? h2_neg : h2_pos) * f_m2 && f_e2 < s2 * f_m2 && !f_capped;
                    const double f_lhs = f_e2 * (my_bxb * my_bxc), f_rhs = (my_bev * my_bev) * f_m2;
                    const bool f_take = my_bx > FZ_X_SUB || f_lhs < f_rhs * (1.0 - 1e-11);
                    const bool f_tie = !f_take && f_lhs <= f_rhs * (1.0 + 1e-11);
#if defined(FW_FZ_FASTDBG) && FW_FZ_FASTDBG >= 2
                    {
                        const bool sig = f_e2 > (f_ev < 0.0 ? h2_neg : h2_pos) * f_m2, nrm = f_e2 < s2 * f_m2;
                        const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
                        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (unsigned)__builtin_ctzll(act)) {
                            atomicAdd(&fz_fast_cnt[0], 1ull);
                            atomicAdd(&fz_fast_cnt[1], (unsigned long long)__builtin_popcountll(act));
                        }
                        const unsigned long long b_clean = __builtin_amdgcn_ballot_w64(!clean), b_sig = __builtin_amdgcn_ballot_w64(!sig), b_nrm = __builtin_amdgcn_ballot_w64(!nrm),
                                                 b_stop = __builtin_amdgcn_ballot_w64(!f_nostop), b_tie = __builtin_amdgcn_ballot_w64(f_tie);
                        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (unsigned)__builtin_ctzll(act)) {
                            if (b_clean) atomicAdd(&fz_fast_cnt[2], 1ull);
                            if (b_sig) atomicAdd(&fz_fast_cnt[3], 1ull);
                            if (b_nrm) atomicAdd(&fz_fast_cnt[4], 1ull);
                            if (b_stop) atomicAdd(&fz_fast_cnt[5], 1ull);
                            if (b_tie) atomicAdd(&fz_fast_cnt[6], 1ull);
                            atomicAdd(&fz_fast_cnt[7], (unsigned long long)__builtin_popcountll(b_nrm));
                        }
                    }
#endif
#if defined(FW_FZ_FASTDBG) && FW_FZ_FASTDBG >= 4
                    {   // potential of a cheap conservative screen (r06): iterations in which EVERY lane's |stat|^2 = e2 / m2 lies clearly above the
                        // smallest value this wavefront has seen so far in the chunk and clearly inside (significant, normal range)
                        const double q = f_e2 / f_m2;
                        double wq = q;
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) wq = __builtin_fmin(wq, __shfl_xor(wq, o));
                        const double hq = f_ev < 0.0 ? h2_neg : h2_pos;
                        const bool p4 = __all(clean && q > dbg_wmin * 1.0002 && q > hq * 1.0002 && q < s2 * 0.9998);
                        const bool p3 = __all(clean && q > dbg_wmin * 1.002 && q > hq * 1.002 && q < s2 * 0.998);
                        const bool p2 = __all(clean && q > dbg_wmin * 1.02 && q > hq * 1.02 && q < s2 * 0.98);
                        if (lane == 0) {
                            atomicAdd(&fz_fast_cnt[19], 1ull);
                            if (p4) atomicAdd(&fz_fast_cnt[16], 1ull);
                            if (p3) atomicAdd(&fz_fast_cnt[17], 1ull);
                            if (p2) atomicAdd(&fz_fast_cnt[18], 1ull);
                        }
                        dbg_wmin = __builtin_fmin(dbg_wmin, wq);
                    }
#endif
#if FW_FZ_SCREEN && defined(FW_FZ_FASTDBG) && FW_FZ_FASTDBG >= 5
                    {   // validation build: the screen decides nothing; counted: lanes it would have skipped, and among them those whose exact value the
                        // bookkeeping needed after all (not sure, tie, or a new lane best BELOW the bound it was screened against) -- must stay 0
                        // (a skipped test beyond the normal range of p is fine: its p is below the bound test's, which lies in the normal range)
                        const bool f_sig = f_e2 > (f_ev < 0.0 ? h2_neg : h2_pos) * f_m2;
                        const bool viol = dbg_sok && (!clean || !f_sig || f_capped || (f_e2 <= (double)dbg_sT / 1.0002 * f_m2));
                        const unsigned long long bs = __ballot(dbg_sok), bv = __ballot(viol);
                        const bool all_ok = __all(dbg_sok);
                        if (lane == (int)__builtin_ctzll(__ballot(true))) {
                            atomicAdd(&fz_fast_cnt[20], (unsigned long long)__popcll(bs));
                            atomicAdd(&fz_fast_cnt[21], (unsigned long long)__popcll(bv));
                            if (all_ok) atomicAdd(&fz_fast_cnt[22], 1ull);
                            atomicAdd(&fz_fast_cnt[23], 1ull);
                        }
                        dbg_sok = false;
                    }
#endif
                    if (__all(clean && f_sure && f_nostop && !f_tie)) {
                        if (f_take) {
                            my_bx = FZ_X_LAZY;
                            my_bps = 0.0;
                            my_br = r;
                            my_bev = f_ev;
                            my_bxb = f_xb;
                            my_bxc = f_xc;
#if FW_FZ_SCREEN
                            // the screen's bound: this test's |stat|^2 = ev^2 / (xb xc), rounded UP into Float32 (f_sure: xb xc > 0, normal range of p);
                            // positive floats order like their bit patterns.  Any wavefront of the workgroup may read it a little late: an older, larger
                            // value is a bound as well.
                            (void)__hip_atomic_fetch_min(&s_scrq, __float_as_uint(((float)f_e2 / ((float)f_xb * (float)f_xc)) * 1.00001f), __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
                        }
                        if (!hasN) break;
                        pos[0] = ni;
                        pos[1] = nj;
                        pos[2] = nk;
                        chg = nchg;
                        boff = nboff;
                        pf_ok = true;
                        continue;
                    }
                    pf_ok = false;
                }
                if (HK && s >= 4) {  // (every chunk of this variant that holds subsets of 4 or 5 variables has its tables)
                    if (!hk_ok || s != hk_s) __builtin_trap();  // would be a chunking bug: fail loudly
                    if (chg <= 1) {  // (z1, z2) changed: this sub-block's table
                        const int pi = pos[0], pj = pos[1];
                        const int d = pi * (a - s + 1) - pi * (pi - 1) / 2 + (pj - pi - 1) - hk_lin0;
                        hk_tb = s_hk + s_hk_off[d];
                        hk_j = pj;
                    }
                    const int nn = a - 1 - hk_j;
                    stat = hk_nn ? fz_hk_stat<true>(hk_tb, nn, s, pos[2] - hk_j - 1, pos[3] - hk_j - 1, pos[4] - hk_j - 1)
                                 : fz_hk_stat<false>(hk_tb, nn, s, pos[2] - hk_j - 1, pos[3] - hk_j - 1, pos[4] - hk_j - 1);
                } else if (TAB3 && s == 3 && tab_ok) {
                    const int pi = pos[0];
                    if (chg <= 0) boff = fz_tab_off(pi, tb_i0, a) - pi - 1;
                    if (chg <= 1) {  // the (z1, z2) entry stays in registers while only the last position moves
                        const int ej = boff + pos[1];
                        tj = s_tab[ej];
                        rj1 = s_tab_r1[ej];
                        rj2 = s_tab_r2[ej];
                        A2j = s_tab_a2[ej];
                    }
                    const int ek = boff + pos[2];
                    const float4 tk = s_tab[ek];
                    const float rk1 = s_tab_r1[ek];
                    const int fj = __float_as_int(tj.w), fk = __float_as_int(tk.w);
                    const float c32 = CORT(fk & FZ_TAB_ZMASK, fj & FZ_TAB_ZMASK);
                    bool f1ok;
                    const float F1f = pc_l1_rf(c32, tk.z, tj.z, rk1, rj1, f1ok);
                    const double F1v = (double)F1f;
                    const double dF = fz_sqrt_unit(1.0 - F1v * F1v);  // shared by the two level-2 values below
                    // wave-uniform fast path: no Float64 literal and no NaN among the inputs (bit 28 = clean entry)
                    if (__all((((fj & fk) >> 28) & 7) == 7 && f1ok)) {
                        const double D2 = pc_l2_all32_d1_nn(tk.x, tj.x, F1f, (double)rj2.x, dF);
                        const double E2 = pc_l2_all32_d1_nn(tk.y, tj.y, F1f, (double)rj2.y, dF);
                        // pc_l3_nn(A2j, D2, E2), first half: numerator and radicands (the quotient follows where it is needed)
                        l3_ev = round5_f64_nn(A2j - D2 * E2);
                        l3_xb = 1.0 - D2 * D2;
                        l3_xc = 1.0 - E2 * E2;
                        screened = true;
                        stat = 0.0;
                    } else {
                        const TV F1 = pc_l1_r(c32, tk.z, tj.z, rk1, rj1);
                        const TV D1{(double)tk.x, ((fk >> 30) & 1) != 0}, Bj{(double)tj.x, ((fj >> 30) & 1) != 0};
                        const TV E1{(double)tk.y, ((fk >> 29) & 1) != 0}, Cj{(double)tj.y, ((fj >> 29) & 1) != 0};
                        const double D2 = pc_l2_d2(D1, Bj, F1, dF);
                        const double E2 = pc_l2_d2(E1, Cj, F1, dF);
                        stat = pc_l3(A2j, D2, E2);
                    }
                } else if (!TAB3 && s == 3) {
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
                    const float cXz3 = CORV(X, z3), cYz3 = CORV(Y, z3), cz3z1 = CORT(z3, z1), cz3z2 = CORT(z3, z2);
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
                } else if (L1T && l3_ok && s == l1_s) {  // level-3 position tables (see FZ_L3_CAP)
                    if (chg <= 2) {  // (z2, z3) changed: this sub-block's entries
                        const int pj = pos[1], pk = pos[2], u = pj - (l3_i + 1);
                        const int d = u * (a - s + 1 - l3_i) - u * (u - 1) / 2 + (pk - pj - 1) - l3_lin0;
                        l3_base = s3_off[d] - s3_low[d];
                    }
                    if (s == 4) {
                        stat = s3_a4[l3_base + pos[3]];
                    } else {
                        if (chg <= 3) {  // z4 changed: its entry stays in registers while only the last position moves
                            const int pl = pos[3], el = l3_base + pl;
                            t1l = s_l1[pl];
                            zl = s_acc[pl];
#if FW_FZ_L1T_ROW32
                            zl_row = zl * p;
#endif
                            p2l = s3_p2[el];
                            fl3 = (s3_fl[el] & 1) != 0;
                            cl3 = (s3_fl[el] & 2) != 0;
                            nn3 = (s3_fl[el] & 4) != 0;
                            d2cl = s3_d2c[el];
                            q3l = s3_q3[el];
                            sq3l = s3_sq3[el];
                            x3l = s3_x3[el];
                            sx3l = s3_sx3[el];
                            y3l = s3_y3[el];
                            sy3l = s3_sy3[el];
                            a4l = s3_a4[el];
                        }
                        const int pm = pos[4], em = l3_base + pm;
                        const float4 t1m = s_l1[pm];
#if FW_FZ_L1T_ROW32
                        // (local matrix: m^2 < 2^24 -- the row base of z4 is computed where z4 changes, the test adds its column)
                        const float c45 = tloc ? cor[(unsigned int)(zl_row + s_acc[pm])] : CORV(s_acc[pm], zl);
#else
                        const float c45 = CORT(s_acc[pm], zl);
#endif
                        bool f1ok;
                        const float R1f = pc_l1_rf(c45, t1m.z, t1l.z, t1m.w, t1l.w, f1ok);     // rho(v,z4|z1)
                        const int flm = (l3_nn || !FW_L3_ENTRY_NN) ? 7 : (int)s3_fl[em];
                        const bool t_nn = l3_nn || (FW_L3_ENTRY_NN && cl3 && (flm & 2) != 0);  // the chunk's tables clean, or this test's two entries
                        const bool nn_all32 = __all(t_nn && f1ok);
                        // r05: no NaN among the inputs but a Float64 literal (a clamped or zero-denominator level-1 value: 0.5 % of cfg5's
                        // size-5 tests, one lane in every fourth wavefront): only the numerator of the level-2 formula depends on the
                        // types (Float32 arithmetic when all three children are Float32 values: pc_l2_d2) -- a select, not the
                        // NaN-preserving form
                        const bool nn_mix = FW_L3_ENTRY_NN && !nn_all32 && __all(nn3 && (flm & 4) != 0 && R1f == R1f);
                        if (nn_all32 || nn_mix) {
                            // wave-uniform fast path: no NaN (and, nn_all32, no Float64 literal) among the inputs -- the NaN-preserving selects
                            // of the clamps / round5 become v_max + v_min and plain arithmetic: the same values for these inputs
                            const double R2 = nn_all32 ? pc_l2_all32_d1_nn(R1f, s3_p2[em], p2l, (double)s3_r2[em], d2cl)  // rho(v,z4|z1,z2)
                                                       : pc_l2_mix_d1_nn(R1f, s3_p2[em], p2l, f1ok && (flm & 1) != 0 && fl3, (double)s3_r2[em], d2cl);
                            const double R3 = pc_l3s_nn(R2, s3_q3[em], q3l, s3_sq3[em], sq3l);                  // rho(v,z4|z1,z2,z3)
                            const double s45 = fz_sq1(R3);
                            const double X4 = pc_l3s_nn(s3_x3[em], x3l, R3, sx3l, s45);                         // rho(X,v|z1..z4)
                            const double Y4 = pc_l3s_nn(s3_y3[em], y3l, R3, sy3l, s45);                         // rho(Y,v|z1..z4)
                            if (SCR) {
                                // rho(X,Y|z1..z4,v) = pc_l3s_nn(a4l, X4, Y4, sqrt(1 - X4^2), sqrt(1 - Y4^2)), first half: numerator and
                                // radicands; fz_l3_finish takes the same quotient (same operations, same values: a root of 0 makes the
                                // denominator 0 in either form) only where the value itself is needed -- see the screen below
                                l3_ev = round5_f64_nn(a4l - X4 * Y4);
                                l3_xb = 1.0 - X4 * X4;
                                l3_xc = 1.0 - Y4 * Y4;
                                screened = true;
                                stat = 0.0;
                            } else {
                                stat = pc_l3s_nn(a4l, X4, Y4, fz_sq1(X4), fz_sq1(Y4));                          // rho(X,Y|z1..z4,v)
                            }
                        } else {
                            const TV R1 = pc_l1_r(c45, t1m.z, t1l.z, t1m.w, t1l.w);
                            const TV Bm{(double)s3_p2[em], (s3_fl[em] & 1) != 0}, Cl{(double)p2l, fl3};
                            const double R2 = pc_l2_d2(R1, Bm, Cl, d2cl);                      // v is the first of the pair
                            const double R3 = pc_l3s(R2, s3_q3[em], q3l, s3_sq3[em], sq3l);
                            const double s45 = fz_sq1(R3);
                            const double X4 = pc_l3s(s3_x3[em], x3l, R3, sx3l, s45);
                            const double Y4 = pc_l3s(s3_y3[em], y3l, R3, sy3l, s45);
                            stat = pc_l3s(a4l, X4, Y4, fz_sq1(X4), fz_sq1(Y4));
                        }
                    }
                } else if (L1T && l1_ok && s == l1_s) {
                    stat = s == 5 ? fz_l1t_stat<5>(cor, p, s_l1, s_l1f, s_acc, pos, s_l1a, s_l1af != 0, l1_clean)
                                  : fz_l1t_stat<4>(cor, p, s_l1, s_l1f, s_acc, pos, s_l1a, s_l1af != 0, l1_clean);
                } else if (HIGHK && !HK) {
                    int zs[FW_MAX_K_FAST];
#pragma unroll
                    for (int q = 0; q < FW_MAX_K_FAST; ++q) zs[q] = (q < s) ? ACCV(pos[q]) : 0;
                    stat = fz_pcor_any(cor, p, X, Y, zs, s);
                } else {
                    stat = 0.0;
                }
                // |stat|^2 = e2 / m2 (TAB3); `sure`: significant, in the normal range of p, not the last test of a capped job -- decided
                // on the squares, no quotient taken (fz_l3_finish)
                double e2 = 0.0, m2 = 1.0;
                bool sure = false;
                if (SCR) {
                    if (screened) {
                        m2 = l3_xb * l3_xc;
                        e2 = l3_ev * l3_ev;
                        sure = e2 > (l3_ev < 0.0 ? h2_neg : h2_pos) * m2 && e2 < s2 * m2 &&
                               !(max_tests > 0 && r + 1 >= (unsigned long long)max_tests);
                        if (!sure) stat = fz_l3_finish(l3_ev, l3_xb, l3_xc);
                    } else {
                        e2 = stat * stat;
                    }
                }
                const double av = fabs(stat);
                if (!sure) {
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
                        (void)__hip_atomic_fetch_min(&s_cstop, (unsigned int)(r - cbase), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        break;
                    }
                }
                // a lane of this workgroup has stopped at an earlier rank of the chunk: nothing behind it matters any more
                // (the merge takes the first stop) -- r01/r02 profile: lanes running on behind the stop were the 18 % of
                // speculative tests
                if (__hip_atomic_load(&s_cstop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned int)(r - cbase)) break;
                // tests.jl:338 `pval >= lowest.pval`, sequential within the run, without evaluating p (or even z) for
                // every test.  In the normal regime (|r| < rsub_lo, i.e. x < FZ_X_SUB) p is strictly decreasing in |r|
                // once two values differ by more than rounding noise, so a clearly smaller |r| replaces the lane best
                // "lazily" (x is computed once per run, below), a clearly larger one is skipped, and only near-ties and
                // the underflow regime take the exact path.
                bool exact = false, lazy_take = false;
                if (SCR) {
                    // the same three-way decision on the squares e2 / m2 against my_bev^2 / (my_bxb my_bxc), cross-multiplied (no division)
                    if (sure || av < rsub_lo) {
                        if (my_bx == FZ_X_NONE || my_bx > FZ_X_SUB) {
                            lazy_take = true;
                        } else {
                            const double lhs = e2 * (my_bxb * my_bxc), rhs = (my_bev * my_bev) * m2;
                            if (lhs < rhs * (1.0 - 1e-11))
                                lazy_take = true;
                            else
                                exact = lhs <= rhs * (1.0 + 1e-11);
                        }
                    } else {
                        exact = true;
                    }
                    if (lazy_take) {
                        my_bx = FZ_X_LAZY;
                        my_bps = 0.0;
                        my_br = r;
                        my_bev = sure ? l3_ev : stat;
                        my_bxb = sure ? l3_xb : 1.0;
                        my_bxc = sure ? l3_xc : 1.0;
                    }
                    if (exact) {
                        if (sure) stat = fz_l3_finish(l3_ev, l3_xb, l3_xc);
                        if (my_bx == FZ_X_LAZY) {
                            if (my_bxb != 1.0 || my_bxc != 1.0) {
                                my_bev = fz_l3_finish(my_bev, my_bxb, my_bxc);
                                my_bxb = my_bxc = 1.0;
                            }
                            my_bx = fz_xkey_slow(my_bev, zscale);
                        }
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
                            my_br = r;
                            my_bev = stat;
                            my_bxb = my_bxc = 1.0;
                        }
                    }
                } else {
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
                }
                // next combination in lexicographic order (sizes descend when one is exhausted)
                if (TAB3 && ilv) {  // 64 ranks ahead, only if that rank is still this lane's (it is then inside the size-3
                    if (r + 64ull >= r1) break;  // enumeration and the carry loop below ends)
                    int i = pos[0], j = pos[1], k = pos[2] + 64;
                    chg = 2;
                    while (k > a - 1) {  // row (i, j) holds k = j + 1 .. a - 1: carry the overflow into the next rows
                        const int over = k - a;
                        if (++j > a - 2) {
                            ++i;
                            j = i + 1;
                            chg = 0;
                        } else if (chg > 1) {
                            chg = 1;
                        }
                        k = j + 1 + over;
                    }
                    pos[0] = i;
                    pos[1] = j;
                    pos[2] = k;
                    continue;
                }
                if (s == 3 && pos[2] < a - 1) {  // by far the most frequent step, with static register indices (the
                    ++pos[2];                     // generic code below indexes pos[] dynamically: ~60 instructions)
                    chg = 2;
                    continue;
                }
                if (HIGHK && s == 5 && pos[4] < a - 1) {
                    ++pos[4];
                    chg = 4;
                    continue;
                }
                if (HIGHK && s == 4 && pos[3] < a - 1) {
                    ++pos[3];
                    chg = 3;
                    continue;
                }
                int i = s - 1;
                while (i >= 0 && pos[i] == a - s + i) --i;
                if (i < 0) {
                    --s;
#pragma unroll
                    for (int q = 0; q < FW_MAX_K_FAST; ++q) pos[q] = q;
                    chg = 0;
                    if (s < 1) break;  // end of the enumeration (r1 never exceeds it)
                } else {
                    ++pos[i];
                    for (int j = i + 1; j < s; ++j) pos[j] = pos[j - 1] + 1;
                    chg = i;
                }
            }
        }
#ifdef FW_FZ_FASTDBG
        const unsigned long long dbg_c2 = __builtin_amdgcn_s_memtime();
#endif
        if (SCR) {  // the lane best's statistic (its quotient, if it has not been taken yet)
            if (my_bx != FZ_X_NONE && (my_bxb != 1.0 || my_bxc != 1.0)) my_bev = fz_l3_finish(my_bev, my_bxb, my_bxc);
            my_bstat = my_bev;
        }
        if (my_bx == FZ_X_LAZY)  // resolve the lazily kept lane best: its x-key
            my_bx = fz_xkey_slow(my_bstat, zscale);
        // first stopping rank in the workgroup
        unsigned long long ws = my_stop;
        if (!FW_FZ_CHEAPRED || __any(my_stop != NONE)) {  // (r06: no lane stopped -- nearly every chunk of cfg3 -- costs one ballot instead of six 64-bit shuffle steps)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned long long t = __shfl_xor(ws, o);
                ws = t < ws ? t : ws;
            }
        }
        // key of the lane best: (xk, ps, rank); xk = x in the normal regime, FZ_X_SUBKEY in the underflow regime
        double bx = (my_bx == FZ_X_NONE) ? FZ_X_NONE : (my_bx > FZ_X_SUB ? FZ_X_SUBKEY : my_bx);
        double bps = my_bps;
        unsigned long long br = my_br;
        bool reduced = false;
        if (FW_FZ_CHEAPRED) {
            // r06: the common case -- one lane holds the smallest x-key -- as a minimum over ONE double and a ballot; equal smallest keys
            // (ties: the exact p, then the later rank decide) and the underflow regime take the full lexicographic reduction below
            double mn = bx;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mn = __builtin_fmin(mn, __shfl_xor(mn, o));
            const unsigned long long eq = __ballot(bx == mn);
            if (mn == FZ_X_NONE) {  // no lane has a candidate: (NONE, ., .) -- what the full reduction returns is never read (fz_key_better)
                reduced = true;
            } else if (mn != FZ_X_SUBKEY && __popcll(eq) == 1) {
                const int src = __ffsll((long long)eq) - 1;
                bx = mn;
                bps = __shfl(bps, src);
                br = __shfl(br, src);
                reduced = true;
            }
        }
        if (!reduced) {
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
                *out_rec = o;
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
#ifdef FW_FZ_FASTDBG
        {
            const unsigned long long dbg_c3 = __builtin_amdgcn_s_memtime();
            dbg_tab += dbg_c1 - dbg_c0;
            dbg_loop += dbg_c2 - dbg_c1;
            dbg_red += dbg_c3 - dbg_c2;
        }
#endif
    }
#undef ACCV
#undef CORV
#undef CORT
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
        *out_rec = o;
#ifdef FW_FZ_FASTDBG
        if (TAB3) {
            atomicAdd(&fz_fast_cnt[8], 1ull);
            atomicAdd(&fz_fast_cnt[9], dbg_t1 - dbg_t0);
            atomicAdd(&fz_fast_cnt[10], dbg_tab);
            atomicAdd(&fz_fast_cnt[11], dbg_loop);
            atomicAdd(&fz_fast_cnt[12], dbg_red);
            atomicAdd(&fz_fast_cnt[13], __builtin_amdgcn_s_memtime() - dbg_t0);
            atomicAdd(&fz_fast_cnt[14], evaluated);
        }
#endif
    }
}
