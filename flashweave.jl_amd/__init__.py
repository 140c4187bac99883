"""flashweave.jl_amd -- host-side mirror of the reference interface for the CI-test hot path, over the
C ABI of libflashweave_amd.so (include/flashweave_amd.h).

Only the path BASELINE.json names lives here: level-0 all-pairs tests and the per-pair conditional test
batch, plus the thin caller (`lgl`) that the reference keeps in Julia (hiton.jl / learning.jl).  Names and
argument meaning follow the reference (src/tests.jl, src/learning.jl); indices are 0-based in Python.

There is no CPU fallback: creating an Engine without a gfx950 device raises FlashWeaveError.
"""
from .engine import (FW_FZ, FW_FZ_NZ, FW_MI, FW_MI_NZ, Engine, FlashWeaveError, TestResult, lib_path, load_library,
                     normalize_counts)  # noqa: F401
from .build import build_library  # noqa: F401
from .api import FWResult, learn_network  # noqa: F401,E402
