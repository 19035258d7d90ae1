"""Epoch loop, monitoring, checkpoint save / resume shared by the trainers.

Mirrors the behaviour of /root/reference/OATrans/base/base_trainer.py:
  Multi_BaseTrainer_dist (:9-244) - one process per GPU, rank-0 logging / checkpointing
  BaseTrainer (:247-475)          - single process (train.py)
Checkpoint layout is the reference's (:163-186): {'arch','epoch','state_dict','optimizer',
'monitor_best','config'} in checkpoint-epoch{N}.pth / model_best.pth, so files interchange.
"""
from abc import abstractmethod

import torch
from numpy import inf

try:
    from OATrans.parallel import HipDataParallel
except ImportError:
    from parallel import HipDataParallel


class _TrainerCore:
    def _setup(self, model, loss, metrics, optimizer, config, writer, is_main):
        self.config = config
        self.logger = config.get_logger('trainer', config['trainer']['verbosity'])
        self.loss = loss
        self.metrics = metrics
        self.optimizer = optimizer
        self.writer = writer
        self.is_main = is_main
        cfg = config['trainer']
        self.epochs = cfg['epochs']
        self.save_period = cfg['save_period']
        self.monitor = cfg.get('monitor', 'off')
        self.init_val = cfg.get('init_val', True)
        if self.monitor == 'off':
            self.mnt_mode, self.mnt_best = 'off', 0
        else:
            self.mnt_mode, self.mnt_metric = self.monitor.split()
            assert self.mnt_mode in ('min', 'max')
            self.mnt_best = inf if self.mnt_mode == 'min' else -inf
            self.early_stop = cfg.get('early_stop', inf)
        self.start_epoch = 1
        self.checkpoint_dir = config.save_dir
        if config.resume is not None:
            self._resume_checkpoint(config.resume)

    @abstractmethod
    def _train_epoch(self, epoch):
        raise NotImplementedError

    def _valid_epoch(self, epoch):
        return {}

    def _run_validation(self, epoch, val_batch):
        """Shared body of the trainers' _valid_epoch (trainer_dist.py:201-281, trainer.py:117-190 of the reference):
        `val_batch(data) -> (text embeddings, video embeddings, loss)` for every batch of every validation loader, then -
        over the WHOLE set - one sim_matrix and every configured metric (t2v_metrics / v2t_metrics), returned as
        `nested_val_metrics` (train() flattens it to val_{loader}_{metric}_{R1...} for the monitor).  Embeddings stay on
        the device until a loader is exhausted: one host transfer per loader instead of one per batch."""
        try:
            from OATrans.model.layers import sim_matrix
        except ImportError:
            from model.layers import sim_matrix
        self.model.eval()
        loaders = self.valid_data_loader
        n_dl = len(loaders)
        total = [torch.zeros((), device=self.device) for _ in range(n_dl)]
        nested_metrics = {x: {} for x in range(n_dl)}
        res_dict = {}
        with torch.no_grad():
            for dl_idx, dl in enumerate(loaders):
                text_arr, vid_arr = [], []
                for data in dl:
                    text_all, vid_all, loss = val_batch(data)
                    text_arr.append(text_all)
                    vid_arr.append(vid_all)
                    total[dl_idx] += loss
                res_dict[f'val_loss_{dl_idx}'] = total[dl_idx].item() / max(1, len(dl))
                if self.writer is not None:
                    self.writer.log_scalar(f'loss_val_{dl_idx}', res_dict[f'val_loss_{dl_idx}'])
                if not text_arr:
                    continue
                sims = sim_matrix(torch.cat(text_arr), torch.cat(vid_arr)).detach().cpu().numpy()
                for metric in (self.metrics or []):
                    res = metric(sims)
                    nested_metrics[dl_idx][metric.__name__] = res
                    if self.is_main:
                        name = getattr(dl, 'dataset_name', str(dl_idx))
                        self.logger.info('[{}] {} epoch {}: R@1 {:.1f} R@5 {:.1f} R@10 {:.1f} R@50 {:.1f} MedR {:g} MeanR {:.1f}'.format(
                            metric.__name__, name, epoch, res['R1'], res['R5'], res['R10'], res['R50'], res['MedR'], res['MeanR']))
                    if self.writer is not None:
                        for key, val in res.items():
                            self.writer.log_scalar(f'{metric.__name__}_{key}_{dl_idx}', val)
        res_dict['nested_val_metrics'] = nested_metrics
        self.model.train()
        return res_dict

    def train(self):
        not_improved = 0
        if self.init_val:
            self._valid_epoch(-1)
        for epoch in range(self.start_epoch, self.epochs + 1):
            result = self._train_epoch(epoch)
            log = {'epoch': epoch}
            for key, value in result.items():
                if key == 'nested_val_metrics':
                    for subkey, subval in value.items():
                        for subsubkey, subsubval in subval.items():
                            for name, v in subsubval.items():
                                log[f"val_{subkey}_{subsubkey}_{name}"] = v
                else:
                    log[key] = value
            if self.is_main:
                for key, value in log.items():
                    self.logger.info('    {:15s}: {}'.format(str(key), value))
            best = False
            if self.mnt_mode != 'off' and self.mnt_metric in log:
                improved = (log[self.mnt_metric] <= self.mnt_best) if self.mnt_mode == 'min' else \
                    (log[self.mnt_metric] >= self.mnt_best)
                if improved:
                    self.mnt_best, not_improved, best = log[self.mnt_metric], 0, True
                else:
                    not_improved += 1
                if not_improved > self.early_stop:
                    if self.is_main:
                        self.logger.info(f"Validation performance didn't improve for {self.early_stop} epochs. Training stops.")
                    break
            if self.is_main and (epoch % self.save_period == 0 or best):
                self._save_checkpoint(epoch, save_best=best)

    def _unwrapped(self):
        return self.model.module if hasattr(self.model, 'module') else self.model

    def _save_checkpoint(self, epoch, save_best=False):
        state = {'arch': type(self._unwrapped()).__name__, 'epoch': epoch, 'state_dict': self.model.state_dict(),
                 'optimizer': self.optimizer.state_dict(), 'monitor_best': self.mnt_best,
                 'config': self.config.config}
        filename = str(self.checkpoint_dir / 'checkpoint-epoch{}.pth'.format(epoch))
        torch.save(state, filename)
        self.logger.info("Saving checkpoint: {} ...".format(filename))
        if save_best:
            torch.save(state, str(self.checkpoint_dir / 'model_best.pth'))
            self.logger.info("Saving current best: model_best.pth ...")

    def _resume_checkpoint(self, resume_path):
        self.logger.info("Loading checkpoint: {} ...".format(resume_path))
        checkpoint = torch.load(str(resume_path), map_location='cpu', weights_only=False)
        self.start_epoch = checkpoint['epoch'] + 1
        self.mnt_best = checkpoint['monitor_best']
        if checkpoint['config']['arch'] != self.config['arch']:
            self.logger.warning("Warning: Architecture configuration given in config file is different from that of "
                                "checkpoint. This may yield an exception while state_dict is being loaded.")
        sd = checkpoint['state_dict']
        ours = list(self.model.state_dict().keys())
        theirs = list(sd.keys())
        if ours and theirs:
            if theirs[0].startswith('module.') and not ours[0].startswith('module.'):
                sd = {k[len('module.'):]: v for k, v in sd.items()}
            elif ours[0].startswith('module.') and not theirs[0].startswith('module.'):
                sd = {'module.' + k: v for k, v in sd.items()}
        self.model.load_state_dict(sd)
        if checkpoint['config']['optimizer']['type'] != self.config['optimizer']['type']:
            self.logger.warning("Warning: Optimizer type given in config file is different from that of checkpoint. "
                                "Optimizer parameters not being resumed.")
        else:
            try:
                self.optimizer.load_state_dict(checkpoint['optimizer'])
            except (ValueError, KeyError) as e:
                self.logger.warning(f"optimizer state not resumed: {e}")
        if hasattr(self.model, 'broadcast_parameters'):
            self.model.broadcast_parameters()
        self.logger.info("Checkpoint loaded. Resume training from epoch {}".format(self.start_epoch))


class Multi_BaseTrainer_dist(_TrainerCore):
    """Base for the torch.distributed trainers: ctor (args, model, loss, metrics, optimizer, config, writer)."""

    def __init__(self, args, model, loss, metrics, optimizer, config, writer=None, init_val=False):
        self.args = args
        self.device = torch.device(f'cuda:{args.local_rank}' if torch.cuda.is_available() else 'cpu')
        model = model.to(self.device)
        if hasattr(model, 'set_device'):
            model.set_device(self.device)                     # base_trainer.py:19
        for m in model.modules():
            if hasattr(m, 'flatten_parameters') and m is not model and hasattr(m, '_engine'):
                m.flatten_parameters()
        # the reference wraps only when n_gpu > 1 yet dereferences .module in validation
        # (trainer_dist.py:227); wrapping always keeps both paths valid.
        self.model = HipDataParallel(model)
        self._setup(self.model, loss, metrics, optimizer, config, writer, is_main=(args.rank == 0))


class BaseTrainer(_TrainerCore):
    """Single-process trainer base: ctor (model, loss, metrics, optimizer, config, writer)."""

    def __init__(self, model, loss, metrics, optimizer, config, writer=None, init_val=False):
        self.device = torch.device('cuda:0' if torch.cuda.is_available() else 'cpu')
        self.model = model.to(self.device)
        if hasattr(model, 'set_device'):
            model.set_device(self.device)
        self._setup(self.model, loss, metrics, optimizer, config, writer, is_main=True)
