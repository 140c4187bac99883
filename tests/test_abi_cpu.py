"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/flashweave_amd.h declares; without a GPU the product fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

import flashweave_jl_amd as fw
from tests.util import ROOT


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(fw.lib_path()):
        fw.build_library()
    return fw.load_library()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "flashweave_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(fw_[a-z0-9_]+)\s*\(", hdr))
    names -= {"fw_allgather_fn", "fw_dev_exchange"}
    assert len(names) >= 20
    raw = ctypes.CDLL(fw.lib_path())
    missing = [n for n in sorted(names) if not hasattr(raw, n)]
    assert not missing, missing


def test_abi_version_and_defaults(lib):
    from flashweave_jl_amd.engine import _Params
    assert lib.fw_abi_version() == 6  # 6: FW_MAX_K 7; 3: fw_dev_exchange, fw_level0_sharded_dev, fw_use_cor_buffer / fw_compute_cor_mat_rows / fw_cor_mat_ready; 4: fw_params.no_cor_mat
    P = _Params()
    lib.fw_params_default(ctypes.byref(P), fw.FW_FZ, 100, 10)
    # learn_network defaults, reference src/learning.jl:466-473
    assert (P.max_k, P.hps, P.fdr, P.n_obs_min, P.max_tests, P.alpha) == (3, 5, 1, -1, 10_000_000, 0.01)


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(fw.FlashWeaveError) as ei:
        fw.Engine("fz", 100, 10)
    assert ei.value.code == -2 and "no CPU fallback" in str(ei.value)


def test_argument_validation(lib):
    from flashweave_jl_amd.engine import _Params
    P = _Params()
    h = ctypes.c_void_p()
    lib.fw_params_default(ctypes.byref(P), 7, 100, 10)
    assert lib.fw_ctx_create(ctypes.byref(P), ctypes.byref(h)) == -1  # unknown kind
    lib.fw_params_default(ctypes.byref(P), fw.FW_FZ, 100, 10)
    P.max_k = 9
    assert lib.fw_ctx_create(ctypes.byref(P), ctypes.byref(h)) == -5  # FW_ERR_LIMIT
    assert b"max_k" in lib.fw_last_error(None)


def test_unranking_root_guess_equals_enumeration(tmp_path):
    # csrc/fw_unrank.h (shared by the device kernels): inverse-binomial unranking with a floating-point root as the guess
    # and an exact integer fix-up == binary-search unranking == plain lexicographic enumeration, exhaustively for
    # |accepted| <= 32 and subset sizes 1..5, and on random ranks for lists of up to 5 000 entries
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "unrank_check")
    subprocess.run(["g++", "-O2", "-o", exe, os.path.join(root, "tests", "native", "unrank_check.cpp")], check=True)
    out = subprocess.run([exe, "32"], check=True, capture_output=True, text=True).stdout
    assert out.startswith("ok "), out


def test_graft_entry_build_passes():
    # the driver's "does it build" check: compiles (or finds up to date) the HIP library + the oracle and asserts the ABI version
    import __graft_entry__ as g
    assert g.build().endswith("libflashweave_amd.so")


def test_struct_layouts_match_header_ctypes_and_this_table(tmp_path):
    # sizeof / offsetof of every struct that crosses the boundary, three ways: the header compiled by gcc, the ctypes mirrors the
    # tests and bench.py call through (engine.py), and the table INTEGRATION.md gives a maintainer who writes the Julia mirrors
    # (r05's stub declared zs::NTuple{5,Int32} against an 80-byte fw_subsets_result: nothing checked the documented layout)
    import subprocess
    from flashweave_jl_amd import engine as E
    mirrors = {"fw_params": E._Params, "fw_test_result": E._TestResult, "fw_subsets_result": E._SubsetsResult,
               "fw_learn_opts": E._LearnOpts, "fw_counters": E._Counters, "fw_dev_exchange": E._DevExchange}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "flashweave_amd.h"', 'int main(void) {']
    for name, cls in mirrors.items():
        src.append('printf("%s sizeof %%zu\\n", sizeof(%s));' % (name, name))
        for f, _ in cls._fields_:
            src.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (name, f, name, f))
    src.append('return 0; }')
    cfile = tmp_path / "layout.c"
    cfile.write_text("\n".join(src))
    exe = str(tmp_path / "layout")
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", exe, str(cfile)], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    hdr = {}
    for ln in out.splitlines():
        s, f, v = ln.split()
        hdr.setdefault(s, {})[f] = int(v)
    assert hdr["fw_subsets_result"]["sizeof"] == 80 and hdr["fw_params"]["sizeof"] == 64  # ABI 6
    # (1) ctypes mirrors == header
    for name, cls in mirrors.items():
        assert ctypes.sizeof(cls) == hdr[name]["sizeof"], name
        for f, _ in cls._fields_:
            assert getattr(cls, f).offset == hdr[name][f], (name, f)
    # (2) the table of INTEGRATION.md == header: every struct, every field
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    rows = re.findall(r"^\| `(fw_[a-z_]+)` \| (\d+) \| (.+?) \|$", doc, flags=re.M)
    seen = {}
    for name, size, fields in rows:
        d = {"sizeof": int(size)}
        for item in fields.split(","):
            f, off = item.split()
            d[f] = int(off)
        seen[name] = d
    assert seen == hdr
    # (3) the Julia mirrors of the stub say the same sizes, and no struct is described twice
    for jl, name in (("FwParams", "fw_params"), ("FwTestResult", "fw_test_result"), ("FwSubsetsResult", "fw_subsets_result"),
                     ("FwLearnOpts", "fw_learn_opts"), ("FwCounters", "fw_counters")):
        m = re.findall(r"^struct %s\s+# mirrors %s.*?: (\d+) bytes" % (jl, name), doc, flags=re.M)
        assert m == [str(hdr[name]["sizeof"])], (jl, m)
    assert "NTuple{%d,Int32}" % fw.engine.FW_MAX_K in doc and "zs::NTuple{5" not in doc.split("### Struct layout")[0]
