// Host-side exhaustive check of csrc/fw_unrank.h: the root-guess unranking equals the binary-search form and the plain
// enumeration for every rank of every (a, s) with a <= A_FULL, and on random ranks for large a.  Prints "ok <count>".
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../flashweave.jl_amd/csrc/fw_unrank.h"

int main(int argc, char **argv)
{
    const int A_FULL = argc > 1 ? atoi(argv[1]) : 40;
    unsigned long long checked = 0;
    for (int s = 1; s <= 5; ++s)
        for (int a = s; a <= A_FULL; ++a) {
            int pos[5], ref[5], enumr[5];
            for (int q = 0; q < s; ++q) enumr[q] = q;
            const unsigned long long N = fw_binom_u64(a, s);
            for (unsigned long long r = 0; r < N; ++r) {
                fw_unrank_comb(r, a, s, pos);
                fw_unrank_bsearch(r, a, s, ref);
                if (memcmp(pos, ref, sizeof(int) * s) || memcmp(pos, enumr, sizeof(int) * s)) {
                    printf("mismatch a %d s %d rank %llu\n", a, s, r);
                    return 1;
                }
                ++checked;
                int q = s - 1;  // next subset in lexicographic order
                while (q >= 0 && enumr[q] == a - s + q) --q;
                if (q >= 0) {
                    ++enumr[q];
                    for (int z = q + 1; z < s; ++z) enumr[z] = enumr[z - 1] + 1;
                }
            }
        }
    unsigned long long x = 88172645463325252ull;
    const int big[] = {41, 63, 64, 65, 120, 255, 256, 511, 512, 513, 1000, 2047, 2048, 5000};
    for (int s = 1; s <= 5; ++s)
        for (int a : big) {
            const unsigned long long N = fw_binom_u64(a, s);
            if (N >= (1ull << 62)) continue;
            for (int it = 0; it < 20000; ++it) {
                x ^= x << 13;
                x ^= x >> 7;
                x ^= x << 17;
                unsigned long long r = it < 4 ? (it == 0 ? 0 : it == 1 ? N - 1 : it == 2 ? N / 2 : 1 % N) : x % N;
                int pos[5], ref[5];
                fw_unrank_comb(r, a, s, pos);
                fw_unrank_bsearch(r, a, s, ref);
                if (memcmp(pos, ref, sizeof(int) * s)) {
                    printf("mismatch a %d s %d rank %llu\n", a, s, r);
                    return 1;
                }
                ++checked;
            }
        }
    // the 32-bit forms of the LDS-table kernel (s <= 3, a <= FW_UNRANK32_A): every rank of every a up to A_FULL32 and of
    // the boundary sizes, against the general form
    const int A_FULL32 = argc > 2 ? atoi(argv[2]) : 96;
    const int edge32[] = {127, 128, 129, 255, 256, 257, 400, 511, 512};
    for (int s = 1; s <= 3; ++s) {
        for (int pass = 0; pass < 2; ++pass) {
            const int na = pass == 0 ? A_FULL32 - s + 1 : (int)(sizeof(edge32) / sizeof(edge32[0]));
            for (int ia = 0; ia < na; ++ia) {
                const int a = pass == 0 ? s + ia : edge32[ia];
                const unsigned long long N = fw_binom_u64(a, s);
                if ((unsigned long long)fw_binom32(a, s) != N) {
                    printf("binom32 mismatch a %d s %d\n", a, s);
                    return 1;
                }
                int enumr[3];
                for (int q = 0; q < s; ++q) enumr[q] = q;
                for (unsigned long long r = 0; r < N; ++r) {
                    int pos[3];
                    fw_unrank_comb32((uint32_t)r, a, s, pos);
                    if (memcmp(pos, enumr, sizeof(int) * s)) {
                        printf("mismatch32 a %d s %d rank %llu\n", a, s, r);
                        return 1;
                    }
                    ++checked;
                    int q = s - 1;
                    while (q >= 0 && enumr[q] == a - s + q) --q;
                    if (q >= 0) {
                        ++enumr[q];
                        for (int z = q + 1; z < s; ++z) enumr[z] = enumr[z - 1] + 1;
                    }
                }
            }
        }
    }
    // sizes 4 and 5 in 32 bits (a <= FW_UNRANK32_A5): every rank up to a = 36, random ranks beyond
    for (int s = 4; s <= 5; ++s) {
        for (int a = s; a <= 36; ++a) {
            const unsigned long long N = fw_binom_u64(a, s);
            if ((unsigned long long)fw_binom32(a, s) != N) {
                printf("binom32 mismatch a %d s %d\n", a, s);
                return 1;
            }
            for (unsigned long long r = 0; r < N; ++r) {
                int pos[5], ref[5];
                fw_unrank_comb32((uint32_t)r, a, s, pos);
                fw_unrank_comb(r, a, s, ref);
                if (memcmp(pos, ref, sizeof(int) * s)) {
                    printf("mismatch32 a %d s %d rank %llu\n", a, s, r);
                    return 1;
                }
                ++checked;
            }
        }
        const int mid32[] = {37, 50, 64, 87, 88, 89, 100, 127, 128};
        for (int a : mid32) {
            const unsigned long long N = fw_binom_u64(a, s);
            if ((unsigned long long)fw_binom32(a, s) != N) {
                printf("binom32 mismatch a %d s %d\n", a, s);
                return 1;
            }
            for (int it = 0; it < 200000; ++it) {
                x ^= x << 13;
                x ^= x >> 7;
                x ^= x << 17;
                const unsigned long long r = it < 3 ? (it == 0 ? 0 : it == 1 ? N - 1 : N / 2) : x % N;
                int pos[5], ref[5];
                fw_unrank_comb32((uint32_t)r, a, s, pos);
                fw_unrank_bsearch(r, a, s, ref);
                if (memcmp(pos, ref, sizeof(int) * s)) {
                    printf("mismatch32 a %d s %d rank %llu\n", a, s, r);
                    return 1;
                }
                ++checked;
            }
        }
    }
    const int big32[] = {513, 777, 1023, 1024};
    for (int s = 1; s <= 3; ++s)
        for (int a : big32) {
            const unsigned long long N = fw_binom_u64(a, s);
            for (int it = 0; it < 200000; ++it) {
                x ^= x << 13;
                x ^= x >> 7;
                x ^= x << 17;
                const unsigned long long r = it < 3 ? (it == 0 ? 0 : it == 1 ? N - 1 : N / 2) : x % N;
                int pos[5], ref[5];
                fw_unrank_comb32((uint32_t)r, a, s, pos);
                fw_unrank_bsearch(r, a, s, ref);
                if (memcmp(pos, ref, sizeof(int) * s)) {
                    printf("mismatch32 a %d s %d rank %llu\n", a, s, r);
                    return 1;
                }
                ++checked;
            }
        }
    printf("ok %llu\n", checked);
    return 0;
}
