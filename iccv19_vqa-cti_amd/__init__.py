"""iccv19_vqa-cti_amd -- MI355X-native Compact Trilinear Interaction hot path.

Drop-in `nn.Module`s for the reference's TCNet / BCNet / BiAttention / TriAttention / FCNet / ModeProduct whose
forward runs in hand-written gfx950 HIP kernels (csrc/) reached through the C ABI of include/cti_hip.h.
The directory name carries a hyphen, so import it through the repo-root `cti_amd` module (or `dropin/src/*`)."""
from . import _lib, ops, autograd                              # noqa: F401
from .fc import FCNet, WNLinear                                # noqa: F401
from .tc import TCNet                                          # noqa: F401
from .bc import BCNet                                          # noqa: F401
from .attention import BiAttention, TriAttention, StackedAttention   # noqa: F401
from .Tensor import ModeProduct                                # noqa: F401
from .ops import set_precision, get_precision, invalidate_caches, set_range_check, f16f6_range_status, run_concurrently   # noqa: F401
from ._lib import CtiError                                     # noqa: F401
from .dp import FlatAdamaxDP                                   # noqa: F401
from .graph import GraphedTrainStep, GraphedForward                            # noqa: F401
from .language_model import WordEmbedding, QuestionEmbedding   # noqa: F401
from .classifier import SimpleClassifier                       # noqa: F401
from .loss_function import BCEWithLogitsSum, Distillation_Loss # noqa: F401
from .teacher_logits import make_json_with_logits, dump_teacher_logits, load_teacher_logits, teacher_logit_batch   # noqa: F401
from . import base_model                                       # noqa: F401
from .base_model import BanModel, CTIModel, TanModel, MCBanModel, build_ban, build_cti, build_mc_cti, build_mc_ban   # noqa: F401

__all__ = ["FCNet", "WNLinear", "TCNet", "BCNet", "BiAttention", "TriAttention", "StackedAttention", "ModeProduct",
           "ops", "set_precision", "get_precision", "invalidate_caches", "set_range_check", "f16f6_range_status", "CtiError", "FlatAdamaxDP", "GraphedTrainStep", "GraphedForward",
           "WordEmbedding", "QuestionEmbedding", "SimpleClassifier", "BCEWithLogitsSum", "Distillation_Loss", "base_model",
           "BanModel", "CTIModel", "TanModel", "MCBanModel", "build_ban", "build_cti", "build_mc_cti", "build_mc_ban"]
