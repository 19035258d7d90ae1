"""Single-process trainer used by train.py (/root/reference/OATrans/trainer/trainer.py:9-113):
no gather, optional per-epoch lr_scheduler.step()."""
import numpy as np
import torch

try:
    from OATrans.base import BaseTrainer
    from OATrans.model.layers import sim_matrix
except ImportError:
    from base import BaseTrainer
    from model.layers import sim_matrix


class Trainer(BaseTrainer):
    def __init__(self, model, loss, metrics, optimizer, config, data_loader, valid_data_loader=None, lr_scheduler=None,
                 len_epoch=None, writer=None, visualizer=None, tokenizer=None, max_samples_per_epoch=50000):
        super().__init__(model, loss, metrics, optimizer, config, writer)
        self.data_loader = data_loader
        self.len_epoch = len_epoch if len_epoch is not None else min(len(x) for x in data_loader)
        self.valid_data_loader = valid_data_loader
        self.do_validation = self.valid_data_loader is not None
        self.lr_scheduler = lr_scheduler
        self.batch_size = self.data_loader[0].batch_size
        self.log_step = max(1, int(np.sqrt(self.batch_size)))
        self.total_batch_sum = sum(x.batch_size for x in self.data_loader)
        self.tokenizer = tokenizer
        self.max_samples_per_epoch = max_samples_per_epoch

    def _to_device(self, data):
        if self.tokenizer is not None and not isinstance(data['text'], dict):
            data['text'] = self.tokenizer(data['text'], return_tensors='pt', padding=True, truncation=True)
        data['text'] = {k: v.to(self.device) for k, v in data['text'].items()}
        data['video'] = data['video'].to(self.device)
        return data

    def _train_epoch(self, epoch):
        self.model.train()
        total = [0.0] * len(self.data_loader)
        n_iter = 0
        for batch_idx, data_li in enumerate(zip(*self.data_loader)):
            if (batch_idx + 1) * self.total_batch_sum > self.max_samples_per_epoch:
                break
            for dl_idx, data in enumerate(data_li):
                data = self._to_device(data)
                if hasattr(self.model, 'begin_step'):
                    self.model.begin_step()
                self.optimizer.zero_grad()
                text_embeds, video_embeds = self.model(data)
                loss = self.loss(sim_matrix(text_embeds, video_embeds))
                loss.backward()
                self.optimizer.step()
                total[dl_idx] += loss.detach().item()
                if batch_idx % self.log_step == 0:
                    self.logger.debug('Train Epoch: {} dl{} [{}] Loss: {:.6f}'.format(epoch, dl_idx, batch_idx, total[dl_idx] / (n_iter + 1)))
            n_iter += 1
            if batch_idx == self.len_epoch:
                break
        log = {f'loss_{i}': total[i] / max(1, n_iter) for i in range(len(self.data_loader))}
        if self.do_validation:
            log.update(self._valid_epoch(epoch))
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        return log

    def _valid_epoch(self, epoch):
        """trainer.py:117-190 of the reference: per-batch loss, retrieval metrics over the whole validation set."""
        def val_batch(data):
            t, v = self.model(self._to_device(data), return_embeds=True)
            return t, v, self.loss(sim_matrix(t, v))
        return self._run_validation(epoch, val_batch)
