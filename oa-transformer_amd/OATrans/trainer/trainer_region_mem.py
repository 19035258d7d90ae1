"""Region-memory trainer (/root/reference/OATrans/trainer/trainer_region_mem.py): the distributed trainer
with the step of trainer/step.py:region_mem_step (4 gathers packed into one, InfoNCE + 0.1 * region BCE)."""
try:
    from OATrans.model.layers import sim_matrix
    from OATrans.model.oa_layers import bce_sum
    from OATrans.trainer.step import region_mem_step
    from OATrans.trainer.trainer_dist import Multi_Trainer_dist as _Base
except ImportError:
    from model.layers import sim_matrix
    from model.oa_layers import bce_sum
    from trainer.step import region_mem_step
    from trainer.trainer_dist import Multi_Trainer_dist as _Base


class Multi_Trainer_dist(_Base):
    def _to_device(self, data):
        data = super()._to_device(data)
        data['text_region_embedding'] = data['text_region_embedding'].to(self.device)
        data['patch_masks'] = data['patch_masks'].to(self.device)
        return data

    step_impl = staticmethod(region_mem_step)

    def _val_batch(self, data):
        """trainer_region_mem.py:226-263 of the reference: InfoNCE(text, video) + BCE_sum(region_sim, patch_mask) / rows
        of the LOCAL batch (no 0.1 weight in validation); the epoch's retrieval metrics use (text, video)."""
        text, video, rsim = self.model.module(data)
        text_all, vid_all = self._gather_embeds(text), self._gather_embeds(video)
        pm = data['patch_masks'].float()
        if pm.dim() == 4:
            pm = pm.squeeze(1)
        loss = self.loss(sim_matrix(text_all, vid_all)) + bce_sum(rsim.reshape(-1, rsim.shape[-1]),
                                                                   pm.reshape(-1, pm.shape[-1])) / rsim.shape[0]
        return text_all, vid_all, loss
