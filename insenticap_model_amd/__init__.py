"""insenticap_model_amd - MI355X (gfx950) native caption-decoder hot path of InSentiCap.

    from insenticap_model_amd import Captioner, XECriterion, Detector, clip_gradient
"""
from .captioner import Captioner, XECriterion  # noqa: F401
from .detector import Detector  # noqa: F401
from .optim import FusedClampAdam, clip_gradient  # noqa: F401
from .rewards import RewardCriterion, get_ciderd_scorer, get_cls_reward, get_self_critical_reward  # noqa: F401

__all__ = ['Captioner', 'XECriterion', 'Detector', 'FusedClampAdam', 'clip_gradient', 'RewardCriterion',
           'get_ciderd_scorer', 'get_self_critical_reward', 'get_cls_reward']
