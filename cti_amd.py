"""Import shim: the package directory is named `iccv19_vqa-cti_amd` (hyphen), which Python cannot import by name.
`import cti_amd` loads it under the module name `iccv19_vqa_cti_amd` and re-exports its public names."""
import importlib.util
import os
import sys

_NAME = "iccv19_vqa_cti_amd"


def _load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "iccv19_vqa-cti_amd")
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(root, "__init__.py"), submodule_search_locations=[root])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod


pkg = _load()
globals().update({k: getattr(pkg, k) for k in pkg.__all__})
__all__ = list(pkg.__all__) + ["pkg"]
