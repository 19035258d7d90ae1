"""Entry point of the region_mem variant (/root/reference/OATrans/train_dist_region_mem.py): identical skeleton to
train_dist_multi.py with module_arch = model.oa_model_region_mem and the matching trainer (:4-11 there)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from OATrans import train_dist_multi as _base  # noqa: E402
from OATrans.model import oa_model_region_mem as module_arch  # noqa: E402
from OATrans.trainer.trainer_region_mem import Multi_Trainer_dist  # noqa: E402

_base.module_arch = module_arch
_base.Multi_Trainer_dist = Multi_Trainer_dist

if __name__ == '__main__':
    config = _base.parse_cli()
    _base.run(config, config.args)
