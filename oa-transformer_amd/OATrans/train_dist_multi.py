"""Distributed pre-training / fine-tuning entry point (one process per GPU).

    python -m torch.distributed.run --nproc_per_node 8 train_dist_multi.py -c configs/pt/...json

Same flags, environment rendezvous (MASTER_ADDR / MASTER_PORT / WORLD_SIZE / RANK / LOCAL_RANK) and
config-driven construction as /root/reference/OATrans/train_dist_multi.py:20-162.  backend 'nccl' on
PyTorch-ROCm IS RCCL; xGMI is used automatically inside a node.  Experiment tracking (sacred /
neptune) is optional and reads credentials from the environment only.
"""
import argparse
import collections
import os
import sys

# dmabuf IPC (the host driver supports nothing else): without it RCCL's intra-node transport fails with
# `hipIpcGetMemHandle: invalid argument`.  Read when the HIP runtime initialises, so set it before torch touches the GPU.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from OATrans import model as module_arch, model as module_loss, model as module_metric  # noqa: E402
from OATrans import optim as module_optim  # noqa: E402
from OATrans.data_loader import data_loader as module_data  # noqa: E402
from OATrans.parse_config_dist_multi import ConfigParser  # noqa: E402
from OATrans.trainer.trainer_dist import Multi_Trainer_dist  # noqa: E402
from OATrans.utils.util import replace_nested_dict_item  # noqa: E402


def init_dataloaders(config, module_data):
    """Build the train loaders, then the same configs with split='val' (reference :89-111)."""
    dl_cfg = config["data_loader"]
    if isinstance(dl_cfg, dict) and "type" in dl_cfg and "args" in dl_cfg:
        train = [config.initialize("data_loader", module_data)]
        dl_cfg['args'] = replace_nested_dict_item(dl_cfg['args'], 'split', 'val')
        return train, [config.initialize("data_loader", module_data)]
    if isinstance(dl_cfg, list):
        train = [config.initialize('data_loader', module_data, index=i) for i in range(len(dl_cfg))]
        for c in dl_cfg:
            c['args'] = replace_nested_dict_item(c['args'], 'split', 'val')
        return train, [config.initialize('data_loader', module_data, index=i) for i in range(len(dl_cfg))]
    raise ValueError("Check data_loader config, not correct format.")


def build_tokenizer(config):
    path = config['arch']['args']['text_params']['model']
    if os.path.isdir(path):
        import transformers
        return transformers.AutoTokenizer.from_pretrained(path)
    return None            # offline: loaders emit pre-tokenised captions


def run(config, args):
    logger = config.get_logger('train')
    os.environ['TOKENIZERS_PARALLELISM'] = "false"
    os.environ['TRANSFORMERS_OFFLINE'] = "1"
    if os.environ.get('OAT_ONE_DEVICE') == '1':      # dry run of the multi-rank path on a one-GPU box (with OAT_DIST_BACKEND=gloo)
        args.local_rank = 0
    torch.cuda.set_device(args.local_rank)
    torch.distributed.init_process_group(backend=os.environ.get('OAT_DIST_BACKEND', 'nccl'), init_method='tcp://{}:{}'.format(args.master_address, args.master_port),
                                         rank=args.rank, world_size=args.world_size)
    if args.rank == 0:
        print('world_size', args.world_size, 'local_rank', args.local_rank, flush=True)
    tokenizer = build_tokenizer(config)
    data_loader, valid_data_loader = init_dataloaders(config, module_data)
    model = config.initialize('arch', module_arch)
    if args.rank == 0:
        logger.info(model)
    loss = config.initialize(name="loss", module=module_loss)
    metrics = [getattr(module_metric, met) for met in config['metrics']]
    if config.get('linear_evaluation', False):
        for name, p in model.named_parameters():
            p.requires_grad = name.startswith(('txt_proj', 'vid_proj'))
    model = model.to(torch.device(f'cuda:{args.local_rank}'))
    for m in (model.video_model, model.text_model):
        m.flatten_parameters()
    trainable = [p for p in model.parameters() if p.requires_grad]
    for m in (model.video_model, model.text_model):
        m._grad_views()                      # persistent .grad buffers exist before the optimiser looks
    optimizer = config.initialize('optimizer', module_optim, trainable)
    trainer = Multi_Trainer_dist(args, model, loss, metrics, optimizer, config=config, data_loader=data_loader,
                                 valid_data_loader=valid_data_loader, lr_scheduler=None, visualizer=None,
                                 writer=None, tokenizer=tokenizer,
                                 max_samples_per_epoch=config['trainer']['max_samples_per_epoch'])
    trainer.train()
    torch.distributed.destroy_process_group()


def parse_cli():
    parser = argparse.ArgumentParser(description='OA-Transformer on MI355X')
    parser.add_argument('-c', '--config', default=None, type=str, help='config file path (default: None)')
    parser.add_argument('-r', '--resume', default=None, type=str, help='path to latest checkpoint (default: None)')
    parser.add_argument('-d', '--device', default=None, type=str, help='indices of GPUs to enable (default: all)')
    parser.add_argument('-o', '--observe', action='store_true', help='Whether to observe (neptune)')
    parser.add_argument('-l', '--launcher', choices=['none', 'pytorch'], default='none', help='job launcher')
    parser.add_argument('-k', '--local_rank', type=int, default=int(os.environ.get('LOCAL_RANK', 0)))
    parser.add_argument('-ma', '--master_address', default=os.environ.get('MASTER_ADDR', '127.0.0.1'))
    parser.add_argument('-mp', '--master_port', type=int, default=int(os.environ.get('MASTER_PORT', 29500)))
    parser.add_argument('-ws', '--world_size', type=int, default=int(os.environ.get('WORLD_SIZE', 1)))
    parser.add_argument('-rk', '--rank', type=int, default=int(os.environ.get('RANK', 0)))
    parser.add_argument('-lr1', '--learning_rate1', type=float, default=2e-4)
    parser.add_argument('-sc', '--schedule', default=[60, 80])
    parser.add_argument('-le', '--linear_evaluation', default=False)
    CustomArgs = collections.namedtuple('CustomArgs', 'flags type target')
    options = [
        CustomArgs(['--lr', '--learning_rate'], type=float, target=('optimizer', 'args', 'lr')),
        CustomArgs(['--bs', '--batch_size'], type=int, target=('data_loader', 'args', 'batch_size')),
    ]
    return ConfigParser(parser, options)


if __name__ == '__main__':
    config = parse_cli()
    run(config, config.args)
