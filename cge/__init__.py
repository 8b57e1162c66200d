"""Import alias: the product package lives in the directory ``cge.jl_amd/`` (the name the
build contract fixes).  A dot is not legal in a Python package directory name, so this
tiny top-level package registers that directory as the submodule ``cge.jl_amd``:

    import cge.jl_amd as cgeh
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cge.jl_amd")
_spec = _ilu.spec_from_file_location(
    "cge.jl_amd", _os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
jl_amd = _ilu.module_from_spec(_spec)
_sys.modules["cge.jl_amd"] = jl_amd
_spec.loader.exec_module(jl_amd)
