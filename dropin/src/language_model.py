"""Drop-in replacement of the reference's src/language_model.py: re-exports the MI355X-native implementation
(see INTEGRATION.md).  Put `dropin/` (or a copy of this file inside the reference tree) ahead on sys.path."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _root not in sys.path:
    sys.path.insert(0, _root)
import cti_amd as _c  # noqa: E402

_m = __import__("iccv19_vqa_cti_amd.language_model", fromlist=["*"])
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
