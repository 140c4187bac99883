"""Shared helpers for the test-suite: golden-file readers (formats of the reference's test/data)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def read_tests_expected():
    """tests_expected.tsv (reference test/tests.jl:16-39): key -> list of (stat, pval, df, suff_power)."""
    exp = {}
    with open(os.path.join(GOLDEN, "tests_expected.tsv")) as f:
        lines = f.read().strip().split("\n")[1:]
    for line in lines:
        k, s, p, df, pw = line.split("\t")
        exp.setdefault(k, []).append((float(s), float(p), int(df), pw == "true"))
    return exp


def read_edgelist(path):
    """.edgelist format of the reference (src/io.jl:338-389): two header lines, then id<TAB>id<TAB>weight.
    Returns {(i, j): w} with 0-based i < j."""
    with open(path) as f:
        lines = f.read().strip().split("\n")
    hdr = lines[0].split("\t")[-1].split(",")
    inv = {h: i for i, h in enumerate(hdr)}
    e = {}
    for line in lines[2:]:
        a, b, w = line.split("\t")
        i, j = inv[a], inv[b]
        e[(min(i, j), max(i, j))] = float(w)
    return e


def load_norm(name, dtype):
    return np.loadtxt(os.path.join(GOLDEN, name + ".tsv"), dtype=dtype)


def rel(a, b):
    if a == b:
        return 0.0
    return abs(a - b) / max(abs(a), abs(b), 1e-300)
