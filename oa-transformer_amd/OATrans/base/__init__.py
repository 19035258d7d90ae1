from .base_trainer import BaseTrainer, Multi_BaseTrainer_dist  # noqa: F401
