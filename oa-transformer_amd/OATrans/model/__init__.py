"""`from OATrans import model as module_arch` (train.py:4, train_dist_multi.py:5) then
`config.initialize('arch', module_arch)` -> getattr(module, 'FrozenInTime'): the names the entry
points resolve by reflection live here (the reference's own __init__ is empty)."""
from .layers import sim_matrix
from .loss import NormSoftmaxLoss
from .metric import t2v_metrics, v2t_metrics
from .oa_model import FrozenInTime
from .video_transformer import SpaceTimeTransformer
from . import oa_model_global_local, oa_model_region_mem  # noqa: F401  (module_arch = model.oa_model_global_local, train_dist_multi_global_local.py:7)

__all__ = ["FrozenInTime", "NormSoftmaxLoss", "SpaceTimeTransformer", "sim_matrix", "t2v_metrics", "v2t_metrics"]
