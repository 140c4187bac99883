// Lexicographic unranking of subset ranks (the enumeration order of tests.jl:281-346 over the POSITIONS of the accepted
// vector).  Shared by the device kernels (fw_fz.hip) and a host-side exhaustive check (tests/native/unrank_check.cpp,
// tests/test_abi_cpu.py): everything here is integer-exact, the floating-point root is only a starting guess.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define FW_HD __host__ __device__ __forceinline__
#else
#define FW_HD inline
#endif

// C(m, t) for t <= 5, saturating at 2^62
FW_HD unsigned long long fw_binom_u64(long long m, int t)
{
    if (m < t) return 0ull;
    const unsigned long long SAT = 1ull << 62;
    const double est = (t == 0) ? 1.0
                                : (t == 1) ? (double)m
                                           : (t == 2) ? 0.5 * m * (m - 1)
                                                      : (t == 3) ? (double)m * (m - 1) * (m - 2) / 6.0
                                                                 : (t == 4) ? (double)m * (m - 1) * (m - 2) * (m - 3) / 24.0
                                                                            : (double)m * (m - 1) * (m - 2) * (m - 3) * (m - 4) / 120.0;
    // the exact path multiplies by up to (m - 4) before the division by t <= 5: the intermediate is t * C(m, t), so saturate
    // where that product would leave 64 bits (2^64 / 5 = 3.69e18), not where C(m, t) itself would
    if (est > 3.6e18) return SAT;
    const unsigned long long u = (unsigned long long)m;
    switch (t) {
        case 0: return 1ull;
        case 1: return u;
        case 2: return u * (u - 1) / 2ull;
        case 3: return (u * (u - 1) / 2ull) * (u - 2) / 3ull;
        case 4: return ((u * (u - 1) / 2ull) * (u - 2) / 3ull) * (u - 3) / 4ull;
        default: return (((u * (u - 1) / 2ull) * (u - 2) / 3ull) * (u - 3) / 4ull) * (u - 4) / 5ull;
    }
}

// general forms for conditioning sets of 6 and 7 variables (r05; slow paths only): C(m, t) by the running product, saturating at 2^62
// (the intermediate holds t * C(m, t)), and the unranking by a linear scan over the positions
FW_HD unsigned long long fw_binom_any(long long m, int t)
{
    if (t < 0 || m < t) return 0ull;
    double est = 1.0;
    for (int i = 1; i <= t; ++i) est = est * (double)(m - t + i) / (double)i;
    if (est > 2.0e18) return 1ull << 62;
    unsigned long long v = 1ull;
    for (int i = 1; i <= t; ++i) v = v * (unsigned long long)(m - t + i) / (unsigned long long)i;
    return v;
}
FW_HD void fw_unrank_scan(unsigned long long rem, int a, int s, int *pos)
{
    int prev = -1;
    for (int d = 0; d < s; ++d) {
        int c = prev + 1;
        for (;;) {
            const unsigned long long with_c = fw_binom_any(a - 1 - c, s - d - 1);
            if (rem < with_c) break;
            rem -= with_c;
            ++c;
        }
        pos[d] = c;
        prev = c;
    }
}

// reference form: binary search per position (3 x log2(a) binomials; kept for the host-side check)
FW_HD void fw_unrank_bsearch(unsigned long long rem, int a, int s, int *pos)
{
    int prev = -1;
    for (int d = 0; d < s; ++d) {
        const int t = s - d;
        const unsigned long long tot = fw_binom_u64(a - 1 - prev, t);
        int lo = prev + 1, hi = a - t;
        while (lo < hi) {  // largest c with tot - C(a - c, t) <= rem
            const int mid = (lo + hi + 1) >> 1;
            const unsigned long long g = tot - fw_binom_u64(a - mid, t);
            if (g <= rem)
                lo = mid;
            else
                hi = mid - 1;
        }
        rem -= tot - fw_binom_u64(a - lo, t);
        pos[d] = lo;
        prev = lo;
    }
}

// smallest m in [t, mmax] with C(m, t) >= R  (1 <= R <= C(mmax, t)): t-th root as the guess, exact fix-up
FW_HD int fw_inv_binom(unsigned long long R, int t, int mmax)
{
    if (t == 1) return (int)R;
    const float x = (float)R;  // single precision is plenty for a guess (and keeps the kernels' register count down)
    int m;
    if (t == 2)
        m = (int)((1.0f + sqrtf(1.0f + 8.0f * x)) * 0.5f);
    else if (t == 3)
        m = (int)cbrtf(6.0f * x) + 1;
    else if (t == 4)
        m = (int)sqrtf(sqrtf(24.0f * x)) + 2;
    else
        m = (int)exp2f(0.2f * log2f(120.0f * x)) + 2;
    m = m < t ? t : (m > mmax ? mmax : m);
    while (m > t && fw_binom_u64(m - 1, t) >= R) --m;
    while (fw_binom_u64(m, t) < R) ++m;
    return m;
}

// position d of the subset = a - (smallest m with C(m, t) >= number of subsets from this rank to the end of the
// block that starts at the previous position): the same answer as fw_unrank_bsearch with ~3 binomials per position
FW_HD void fw_unrank_comb(unsigned long long rem, int a, int s, int *pos)
{
    int prev = -1;
    for (int d = 0; d < s; ++d) {
        const int t = s - d, n = a - 1 - prev;
        const unsigned long long tot = fw_binom_u64(n, t);
        const int m = fw_inv_binom(tot - rem, t, n);
        rem -= tot - fw_binom_u64(m, t);
        pos[d] = a - m;
        prev = a - m;
    }
}

// ---- 32-bit forms for subsets of at most 3 positions out of a <= FW_UNRANK32_A (the LDS-table kernel: a <= 512) ----
// C(1024, 3) < 2^28 and m (m - 1) / 2 * (m - 2) < 2^29: everything fits 32-bit integers, the divisions are by constants
// and the root guesses are single hardware instructions -- ~100 instructions per unranking instead of ~1 500 for the
// 64-bit general form (which pays for saturating binomials up to t = 5).  Same exact integer fix-up, same answers
// (tests/native/unrank_check.cpp compares the two on every rank).
#define FW_UNRANK32_A 1024
#define FW_UNRANK32_A5 128  // positions 4 and 5: C(m, 4) * (m - 4) < 2^32 up to m = 130
FW_HD uint32_t fw_binom32(int m, int t)  // t in 1..5, 0 <= m <= FW_UNRANK32_A (t <= 3) / FW_UNRANK32_A5 (t = 4, 5)
{
    if (m < t) return 0u;
    const uint32_t u = (uint32_t)m;
    if (t == 1) return u;
    const uint32_t h = (u * (u - 1u)) >> 1;
    if (t == 2) return h;
    const uint32_t c3 = h * (u - 2u) / 3u;
    if (t == 3) return c3;
    const uint32_t c4 = c3 * (u - 3u) / 4u;
    return t == 4 ? c4 : c4 * (u - 4u) / 5u;
}

// smallest m in [t, mmax] with C(m, t) >= R  (1 <= R <= C(mmax, t))
FW_HD int fw_inv_binom32(uint32_t R, int t, int mmax)
{
    if (t == 1) return (int)R;
    const float x = (float)R;
    int m;
    if (t == 2) {
        m = (int)((1.0f + sqrtf(1.0f + 8.0f * x)) * 0.5f);
    } else {
        const float f = t == 3 ? 6.0f : (t == 4 ? 24.0f : 120.0f);
#if defined(__HIP_DEVICE_COMPILE__)
        // v_log_f32 / v_exp_f32: a guess, the integer fix-up below makes it exact
        m = (int)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf(f * x) * (t == 3 ? 0.33333334f : (t == 4 ? 0.25f : 0.2f))) + (t + 1) / 2;
#else
        m = (int)(t == 3 ? cbrtf(f * x) : (t == 4 ? sqrtf(sqrtf(f * x)) : exp2f(0.2f * log2f(f * x)))) + (t + 1) / 2;
#endif
    }
    m = m < t ? t : (m > mmax ? mmax : m);
    while (m > t && fw_binom32(m - 1, t) >= R) --m;
    while (fw_binom32(m, t) < R) ++m;
    return m;
}

FW_HD void fw_unrank_comb32(uint32_t rem, int a, int s, int *pos)  // s <= 3: a <= FW_UNRANK32_A; s <= 5: a <= FW_UNRANK32_A5
{
    int prev = -1;
    for (int d = 0; d < s; ++d) {
        const int t = s - d, n = a - 1 - prev;
        const uint32_t tot = fw_binom32(n, t);
        const int m = fw_inv_binom32(tot - rem, t, n);
        rem -= tot - fw_binom32(m, t);
        pos[d] = a - m;
        prev = a - m;
    }
}
