"""Import shim: `import ngs_barcode_count_amd` loads the package in ./ngs-barcode-count_amd/
(that directory name is not a Python identifier)."""
import importlib.util
import os
import sys

_d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ngs-barcode-count_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_d, "__init__.py"),
                                               submodule_search_locations=[_d])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
