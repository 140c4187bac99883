"""Import shim: the package directory is named `flashweave.jl_amd` (after the reference, FlashWeave.jl), which is
not a valid Python identifier; `import flashweave_jl_amd` loads it under this name."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "flashweave.jl_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_pkg_dir, "__init__.py"),
                                               submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
