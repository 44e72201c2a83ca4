"""insenticap_model_amd - MI355X (gfx950) native caption-decoder hot path of InSentiCap.

    from insenticap_model_amd import Captioner, XECriterion
"""
from .captioner import Captioner, XECriterion  # noqa: F401

__all__ = ['Captioner', 'XECriterion']
