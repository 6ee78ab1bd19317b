"""Import alias.  The product package lives in the directory ``conan-fgw_amd/`` (the repository's
required layout); a hyphen is not importable, so this one-file shim exposes it as ``conan_fgw_amd``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "conan-fgw_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
