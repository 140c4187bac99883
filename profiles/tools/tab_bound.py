"""Worst-case number of LDS table entries (FZ_TAB_CAP, csrc/fw_fz.hip) a chunk of C consecutive ranks of the size-3
subset enumeration over `a` accepted variables can touch.  Block i (first position) holds n = a-1-i entries and
n(n-1)/2 ranks.  Prints the maximum over a <= A for C = 4096 (256 lanes x FW_RUN_MAX = 16)."""
import sys


def worst(a, C):
    ns = [a - 1 - i for i in range(a - 2)]
    starts = [0]
    for n in ns:
        starts.append(starts[-1] + n * (n - 1) // 2)
    tot, best = starts[-1], 0
    for b in range(len(ns)):  # worst case: the chunk begins on the last rank of a block
        s0 = starts[b + 1] - 1
        e = min(tot, s0 + C)
        E, i = 0, b
        while i < len(ns) and starts[i] < e:
            E += ns[i]
            i += 1
        best = max(best, E)
    return best


if __name__ == "__main__":
    A = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    print(max(worst(a, C) for a in range(3, A + 1)))
