"""ConfigParser - JSON experiment config + CLI overrides + reflection factory.

Same contract as /root/reference/OATrans/parse_config_dist_multi.py:13-150 (and parse_config.py):
  * `-c config.json` or `-r checkpoint` (reads the sibling config.json), `--lr/--bs`-style overrides
    through CustomArgs(flags, type, target)
  * save_dir/{models,log,web}/<name>/<timestamp>, config.json written next to the checkpoints
  * initialize(name, module, *args, index=None, **kwargs): config[name]['type'] is looked up on
    `module` and built with config[name]['args']; constructor parameters missing there are filled
    from top-level config keys; the parsed CLI namespace is injected as `args` for the three classes
    that take it (:93-98).
A `.yaml` file with the identical schema is accepted as a convenience (north_star wording).
"""
import inspect
import logging
import os
from datetime import datetime
from functools import reduce
from operator import getitem
from pathlib import Path

try:
    from OATrans.logger import setup_logging
    from OATrans.utils.util import read_json, write_json
except ImportError:                      # run with cwd = OATrans/ like the reference
    from logger import setup_logging
    from utils.util import read_json, write_json

_ARGS_INJECTED = ("FrozenInTime", "MultiDistTextObjectVideoDataLoader", "TextObjectVideoDataLoader")


def _read_config(path):
    path = Path(path)
    if path.suffix in (".yaml", ".yml"):
        import yaml
        with path.open() as fh:
            return yaml.safe_load(fh)
    return read_json(path)


class ConfigParser:
    def __init__(self, args, options='', timestamp=True, test=False):
        for opt in options:
            args.add_argument(*opt.flags, default=None, type=opt.type)
        args = args.parse_args()
        self.args = args
        if getattr(args, "device", None):
            os.environ["CUDA_VISIBLE_DEVICES"] = args.device
        if args.resume is None:
            assert args.config is not None, "Configuration file need to be specified. Add '-c config.json', for example."
            self.cfg_fname = Path(args.config)
            config = _read_config(self.cfg_fname)
            self.resume = None
        else:
            self.resume = Path(args.resume)
            config = read_json(self.resume.parent / 'config.json')
            if args.config is not None:
                config.update(_read_config(args.config))
        self._config = _update_config(config, options, args)
        save_dir = Path(self.config['trainer']['save_dir'])
        stamp = datetime.now().strftime(r'%m%d_%H%M%S') if timestamp else ''
        name = self.config['name']
        self._save_dir = save_dir / 'models' / name / stamp
        self._web_log_dir = save_dir / 'web' / name / stamp
        self._log_dir = save_dir / 'log' / name / stamp
        self.log_levels = {0: logging.WARNING, 1: logging.INFO, 2: logging.DEBUG}
        if not test:
            self.save_dir.mkdir(parents=True, exist_ok=True)
            self.log_dir.mkdir(parents=True, exist_ok=True)
            write_json(self.config, self.save_dir / 'config.json')
            setup_logging(self.log_dir)

    def initialize(self, name, module, *args, index=None, **kwargs):
        node = self[name] if index is None else self[name][index]
        module_name = node['type']
        module_args = dict(node['args'])
        if index is None:
            assert all(k not in module_args for k in kwargs), 'Overwriting kwargs given in config file is not allowed'
            module_args.update(kwargs)
        cls = getattr(module, module_name)
        for param in inspect.signature(cls.__init__).parameters:
            if param not in module_args and param in self.config:
                module_args[param] = self[param]
            if param == 'args' and module_name in _ARGS_INJECTED:
                module_args[param] = self.args
        return cls(*args, **module_args)

    def __getitem__(self, name):
        return self.config[name]

    def get(self, name, default=None):
        return self.config.get(name, default)

    def get_logger(self, name, verbosity=2):
        assert verbosity in self.log_levels, f'verbosity option {verbosity} is invalid. Valid options are {self.log_levels.keys()}.'
        logger = logging.getLogger(name)
        logger.setLevel(self.log_levels[verbosity])
        return logger

    @property
    def config(self):
        return self._config

    @property
    def save_dir(self):
        return self._save_dir

    @property
    def log_dir(self):
        return self._log_dir


def _update_config(config, options, args):
    for opt in options:
        value = getattr(args, _get_opt_name(opt.flags))
        if value is not None:
            _set_by_path(config, opt.target, value)
    return config


def _get_opt_name(flags):
    for flg in flags:
        if flg.startswith('--'):
            return flg.replace('--', '')
    return flags[0].replace('--', '')


def _set_by_path(tree, keys, value):
    reduce(getitem, keys[:-1], tree)[keys[-1]] = value
