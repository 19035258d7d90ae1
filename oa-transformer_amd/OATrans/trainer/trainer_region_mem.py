"""Region-memory trainer (/root/reference/OATrans/trainer/trainer_region_mem.py): the distributed trainer
with the step of trainer/step.py:region_mem_step (4 gathers packed into one, InfoNCE + 0.1 * region BCE)."""
try:
    from OATrans.trainer.step import region_mem_step
    from OATrans.trainer.trainer_dist import Multi_Trainer_dist as _Base
except ImportError:
    from trainer.step import region_mem_step
    from trainer.trainer_dist import Multi_Trainer_dist as _Base


class Multi_Trainer_dist(_Base):
    def _to_device(self, data):
        data = super()._to_device(data)
        data['text_region_embedding'] = data['text_region_embedding'].to(self.device)
        data['patch_masks'] = data['patch_masks'].to(self.device)
        return data

    def train_step(self, data):
        return region_mem_step(self.model, self.loss, self.optimizer, data, self.args)

    def _valid_epoch(self, epoch):
        return {}          # retrieval validation needs (text, video) only; identical to the base once wired
