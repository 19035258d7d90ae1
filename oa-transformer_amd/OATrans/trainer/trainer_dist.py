"""Distributed training step loop - the hot path's outer loop.

Contract of /root/reference/OATrans/trainer/trainer_dist.py:
  Multi_Trainer_dist(args, model, loss, metrics, optimizer, config, data_loader, valid_data_loader,
                     lr_scheduler, len_epoch, writer, visualizer, tokenizer, max_samples_per_epoch)  (:65-67)
  _train_epoch (:124-199): one optimiser step per loader per iteration; tokenise -> H2D -> forward ->
  all-gather video/text embeddings -> sim_matrix -> loss -> backward -> step; LR = lr1 * 0.1^k at the
  schedule milestones (:117-122), set per epoch.
Differences by construction: a single packed all-gather, gradient all-reduce over flat buffers on a side
stream, no per-step `.item()` host syncs (losses are accumulated on the device, read once per log step).
"""
import os
import time

import numpy as np
import torch

try:
    from OATrans.base import Multi_BaseTrainer_dist
    from OATrans.model.layers import sim_matrix
    from OATrans.parallel import AllGather_multi
    from OATrans.trainer.step import hot_step
    from OATrans.utils.util import inf_loop
except ImportError:
    from base import Multi_BaseTrainer_dist
    from model.layers import sim_matrix
    from parallel import AllGather_multi
    from trainer.step import hot_step
    from utils.util import inf_loop


class Multi_Trainer_dist(Multi_BaseTrainer_dist):
    def __init__(self, args, model, loss, metrics, optimizer, config, data_loader, valid_data_loader=None,
                 lr_scheduler=None, len_epoch=None, writer=None, visualizer=None, tokenizer=None,
                 max_samples_per_epoch=50000):
        super().__init__(args, model, loss, metrics, optimizer, config, writer)
        self.data_loader = data_loader
        if len_epoch is None:
            self.len_epoch = min(len(x) for x in data_loader)
        else:
            self.data_loader = inf_loop(data_loader)
            self.len_epoch = len_epoch
        self.valid_data_loader = valid_data_loader
        self.do_validation = self.valid_data_loader is not None
        self.lr_scheduler = lr_scheduler
        self.visualizer = visualizer
        self.batch_size = self.data_loader[0].batch_size
        self.log_step = max(1, int(np.sqrt(self.batch_size)))
        self.total_batch_sum = sum(x.batch_size for x in self.data_loader)
        self.tokenizer = tokenizer
        self.max_samples_per_epoch = max_samples_per_epoch
        self.n_gpu = self.args.world_size
        self.allgather = AllGather_multi.apply
        if hasattr(optimizer, 'grad_scale'):
            optimizer.grad_scale = 1.0          # GradSync averages; keep the optimiser neutral

    def _adjust_learning_rate(self, optimizer, epoch, args):
        lr = args.learning_rate1
        for milestone in args.schedule:
            lr *= 0.1 if epoch >= milestone else 1.
        for group in optimizer.param_groups:
            group['lr'] = lr

    def _to_device(self, data):
        if self.tokenizer is not None and not isinstance(data['text'], dict):
            data['text'] = self.tokenizer(data['text'], return_tensors='pt', padding=True, truncation=True)
        data['text'] = {k: v.to(self.device, non_blocking=True) for k, v in data['text'].items()}
        data['video'] = data['video'].to(self.device, non_blocking=True)
        return data

    step_impl = staticmethod(hot_step)

    def train_step(self, data):
        """forward -> gather -> sim -> loss -> backward -> grad sync -> step; returns the device loss.
        OAT_GRAPH_STEP=1 (one rank): the step is captured per input signature and replayed as one hipGraph launch
        (trainer/graph_step.py); meant for fixed-shape batches (captions padded to a fixed length)."""
        if self.args.world_size == 1 and os.environ.get("OAT_GRAPH_STEP", "0") == "1":
            if getattr(self, "_graphed", None) is None:
                try:
                    from OATrans.trainer.graph_step import GraphedStep
                except ImportError:
                    from trainer.graph_step import GraphedStep
                self._graphed = GraphedStep(type(self).step_impl, self.model, self.loss, self.optimizer, self.args)
            return self._graphed(data)
        return type(self).step_impl(self.model, self.loss, self.optimizer, data, self.args)

    def _train_epoch(self, epoch):
        self.model.train()
        total_loss = [torch.zeros((), device=self.device) for _ in self.data_loader]
        for loader in self.data_loader:
            loader.train_sampler.set_epoch(epoch)
        begin = time.time()
        n_iter = 0
        for batch_idx, data_li in enumerate(zip(*self.data_loader)):
            if (batch_idx + 1) * self.total_batch_sum > self.max_samples_per_epoch:
                break
            for dl_idx, data in enumerate(data_li):
                loss = self.train_step(self._to_device(data))
                total_loss[dl_idx] += loss
                if batch_idx % self.log_step == 0 and self.args.rank == 0:
                    val = loss.item()                                   # the only host sync, on log steps
                    self.logger.debug('Train Epoch: {} dl{} {} Loss: {:.6f} ({:.2f}s)'.format(
                        epoch, dl_idx, self._progress(batch_idx, dl_idx), val, time.time() - begin))
                    if self.writer is not None:
                        self.writer.log_scalar(f'loss_train_{dl_idx}', val)
                    begin = time.time()
            n_iter += 1
            if batch_idx == self.len_epoch:
                break
        log = {f'loss_{i}': (total_loss[i].item() / max(1, self.len_epoch)) for i in range(len(self.data_loader))}   # reference :187-189
        if self.do_validation:
            val_log = self._valid_epoch(epoch)
            if self.args.rank == 0:
                log.update(val_log)
        self._adjust_learning_rate(self.optimizer, epoch, self.args)
        return log

    def _gather_embeds(self, t):
        """Raw (no-grad) all_gather in rank order, as trainer_dist.py:230-237 of the reference."""
        if self.n_gpu <= 1:
            return t
        out = torch.empty((self.n_gpu * t.shape[0], t.shape[1]), dtype=t.dtype, device=t.device)
        torch.distributed.all_gather_into_tensor(out, t.contiguous())
        return out

    def _val_batch(self, data):
        """(gathered text, gathered video, loss) of one validation batch (trainer_dist.py:226-246); the object-aware
        trainers override this with their own forward and loss terms."""
        text_embed, vid_embed = self.model.module(data, return_embeds=True)
        text_all, vid_all = self._gather_embeds(text_embed), self._gather_embeds(vid_embed)
        return text_all, vid_all, self.loss(sim_matrix(text_all, vid_all))

    def _valid_epoch(self, epoch):
        """trainer_dist.py:201-281 of the reference (see _TrainerCore._run_validation)."""
        return self._run_validation(epoch, lambda data: self._val_batch(self._to_device(data)))

    def _progress(self, batch_idx, dl_idx):
        dl = self.data_loader[dl_idx]
        if hasattr(dl, 'n_samples'):
            current, total = batch_idx * dl.batch_size, int(dl.n_samples / self.n_gpu)
        else:
            current, total = batch_idx, self.len_epoch
        return '[{}/{} ({:.0f}%)]'.format(current, total, 100.0 * current / max(1, total))
