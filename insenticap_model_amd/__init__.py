"""insenticap_model_amd - MI355X (gfx950) native caption-decoder hot path of InSentiCap.

    from insenticap_model_amd import Captioner, XECriterion, clip_gradient
"""
from .captioner import Captioner, XECriterion  # noqa: F401
from .optim import FusedClampAdam, clip_gradient  # noqa: F401

__all__ = ['Captioner', 'XECriterion', 'FusedClampAdam', 'clip_gradient']
