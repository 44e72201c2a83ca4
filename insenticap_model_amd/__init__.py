"""insenticap_model_amd - MI355X (gfx950) native caption-decoder hot path of InSentiCap.

    from insenticap_model_amd import Captioner, XECriterion, Detector, clip_gradient
"""
import os as _os

# Kernel arguments in device memory instead of host memory: the GEMM / step entry points take ~1 KB
# argument blocks whose first (dependent) scalar loads otherwise cross PCIe - measured 18.5 -> 11 us per
# small launch on MI355X.  Read by the HIP runtime when it initialises (first HIP call), so it has to be
# in the environment before then; an explicit user setting wins.
_os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

from .captioner import Captioner, XECriterion  # noqa: F401,E402
from .detector import Detector  # noqa: F401,E402
from .optim import FusedClampAdam, clip_gradient  # noqa: F401,E402
from .rewards import RewardCriterion, get_ciderd_scorer, get_cls_reward, get_self_critical_reward  # noqa: F401,E402

__all__ = ['Captioner', 'XECriterion', 'Detector', 'FusedClampAdam', 'clip_gradient', 'RewardCriterion',
           'get_ciderd_scorer', 'get_self_critical_reward', 'get_cls_reward']
