"""Drop-in for the reference's src/classifier.py: re-exports the MI355X implementation."""
import cti_amd  # noqa: F401  (repo root on sys.path)
from iccv19_vqa_cti_amd.classifier import *  # noqa: F401,F403
