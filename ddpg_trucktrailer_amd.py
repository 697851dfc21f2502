"""Import shim: the package directory is `ddpg-trucktrailer_amd/` (not a Python identifier), so
this module gives it an importable name.  `import ddpg_trucktrailer_amd.vec_env` etc. resolve
inside that directory through `__path__`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "ddpg-trucktrailer_amd")]
_init = _os.path.join(__path__[0], "__init__.py")
with open(_init) as _f:
    exec(compile(_f.read(), _init, "exec"))
