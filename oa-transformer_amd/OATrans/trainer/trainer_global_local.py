"""Global+local trainer (/root/reference/OATrans/trainer/trainer_global_local.py): three InfoNCE terms over
six gathered tensors (one packed collective here).  The reference leaves torch.autograd.set_detect_anomaly
on at import (:16) - a debugging leftover that is not reproduced."""
try:
    from OATrans.model.layers import sim_matrix
    from OATrans.trainer.step import global_local_step
    from OATrans.trainer.trainer_dist import Multi_Trainer_dist as _Base
except ImportError:
    from model.layers import sim_matrix
    from trainer.step import global_local_step
    from trainer.trainer_dist import Multi_Trainer_dist as _Base


class Multi_Trainer_dist(_Base):
    def _to_device(self, data):
        data = super()._to_device(data)
        if self.tokenizer is not None and not isinstance(data['pad_text'], dict):
            data['pad_text'] = self.tokenizer(data['pad_text'], return_tensors='pt', padding=True, truncation=True)
        data['pad_text'] = {k: v.to(self.device) for k, v in data['pad_text'].items()}
        for k in ('patch_masks', 'object_token_masks', 'object_token_len'):
            data[k] = data[k].to(self.device)
        return data

    step_impl = staticmethod(global_local_step)

    def _val_batch(self, data):
        """trainer_global_local.py:296-362 of the reference: short-text and tag-padded-text losses against the video
        embedding; the retrieval metrics of the epoch are computed on (short text, video)."""
        text, pad_text, video, _pad_video, _extra = self.model.module(data, return_embeds=True)
        text_all, pad_all, vid_all = (self._gather_embeds(t) for t in (text, pad_text, video))
        loss = self.loss(sim_matrix(text_all, vid_all)) + self.loss(sim_matrix(pad_all, vid_all))
        return text_all, vid_all, loss
