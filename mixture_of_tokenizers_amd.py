"""Import shim: the package directory is ``mixture-of-tokenizers_amd/`` (a hyphen cannot appear in
a Python identifier), so ``import mixture_of_tokenizers_amd`` loads that directory under this name."""
import importlib.util
import sys
from pathlib import Path

_dir = Path(__file__).resolve().parent / "mixture-of-tokenizers_amd"
_spec = importlib.util.spec_from_file_location(__name__, _dir / "__init__.py", submodule_search_locations=[str(_dir)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
