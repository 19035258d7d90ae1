"""Single-process entry point (/root/reference/OATrans/train.py): same config-driven construction,
Trainer without the embedding gather.  The HIP engine needs an MI355X; on a CPU-only host this
script stops with a clear error instead of silently running a different code path."""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from OATrans import model as module_arch, model as module_loss, model as module_metric  # noqa: E402
from OATrans import optim as module_optim  # noqa: E402
from OATrans.data_loader import data_loader as module_data  # noqa: E402
from OATrans.parse_config import ConfigParser  # noqa: E402
from OATrans.trainer.trainer import Trainer  # noqa: E402
from OATrans.train_dist_multi import build_tokenizer, init_dataloaders  # noqa: E402


def run(config):
    if not torch.cuda.is_available():
        raise SystemExit("train.py: no MI355X visible - the training hot path has no CPU fallback "
                         "(use oracle/ for CPU arithmetic)")
    logger = config.get_logger('train')
    tokenizer = build_tokenizer(config)
    data_loader, valid_data_loader = init_dataloaders(config, module_data)
    model = config.initialize('arch', module_arch)
    logger.info(model)
    loss = config.initialize(name="loss", module=module_loss)
    metrics = [getattr(module_metric, met) for met in config['metrics']]
    model = model.cuda()
    for m in (model.video_model, model.text_model):
        m.flatten_parameters()
        m._grad_views()
    optimizer = config.initialize('optimizer', module_optim, [p for p in model.parameters() if p.requires_grad])
    trainer = Trainer(model, loss, metrics, optimizer, config=config, data_loader=data_loader,
                      valid_data_loader=valid_data_loader, lr_scheduler=None, tokenizer=tokenizer,
                      max_samples_per_epoch=config['trainer']['max_samples_per_epoch'])
    trainer.train()


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='OA-Transformer on MI355X')
    parser.add_argument('-c', '--config', default=None, type=str)
    parser.add_argument('-r', '--resume', default=None, type=str)
    parser.add_argument('-d', '--device', default=None, type=str)
    parser.add_argument('-o', '--observe', action='store_true')
    CustomArgs = collections.namedtuple('CustomArgs', 'flags type target')
    options = [
        CustomArgs(['--lr', '--learning_rate'], type=float, target=('optimizer', 'args', 'lr')),
        CustomArgs(['--bs', '--batch_size'], type=int, target=('data_loader', 'args', 'batch_size')),
    ]
    run(ConfigParser(parser, options))
