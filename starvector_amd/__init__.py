"""Import shim: ``star-vector_amd/`` (hyphen, repo layout) is not an importable name, so this package
re-homes ``starvector_amd`` onto that directory.  No code lives here."""
import importlib.util as _u
import os as _os
import sys as _sys

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "star-vector_amd")
_spec = _u.spec_from_file_location("starvector_amd", _os.path.join(_real, "__init__.py"),
                                   submodule_search_locations=[_real])
_mod = _u.module_from_spec(_spec)
_sys.modules["starvector_amd"] = _mod
_spec.loader.exec_module(_mod)
