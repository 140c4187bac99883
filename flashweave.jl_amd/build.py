"""Builds libflashweave_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build_library(force=False, jobs=4):
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, "-j%d" % jobs]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    so = os.path.join(_HERE, "libflashweave_amd.so")
    if not os.path.exists(so):
        raise RuntimeError("build did not produce " + so)
    return so
