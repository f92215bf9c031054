"""Import shim: the package directory is `tera-mind_amd/` (hyphen, per the repo layout
contract), which is not a legal Python identifier.  `import teramind_amd` loads that
directory as the package `teramind_amd` (sub-modules resolve normally afterwards)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tera-mind_amd")
_spec = _ilu.spec_from_file_location("teramind_amd", _os.path.join(_dir, "__init__.py"),
                                     submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["teramind_amd"] = _mod
_spec.loader.exec_module(_mod)
