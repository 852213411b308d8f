"""Import shim: the package directory is named `p3d-raytracer_amd` (hyphen), which Python
cannot import by name.  `import p3d_amd` loads it under the module name p3d_raytracer_amd
and re-exports its public names."""
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.join(_HERE, "p3d-raytracer_amd")

if "p3d_raytracer_amd" not in sys.modules:
    _spec = importlib.util.spec_from_file_location("p3d_raytracer_amd", os.path.join(_PKG, "__init__.py"),
                                                   submodule_search_locations=[_PKG])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules["p3d_raytracer_amd"] = _mod
    _spec.loader.exec_module(_mod)
_mod = sys.modules["p3d_raytracer_amd"]
globals().update({k: v for k, v in vars(_mod).items() if not k.startswith("__")})
