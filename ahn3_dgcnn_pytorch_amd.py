"""Import shim: exposes the package directory `ahn3-dgcnn.pytorch_amd/` (whose name is not a
valid Python identifier) as the module `ahn3_dgcnn_pytorch_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ahn3-dgcnn.pytorch_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
