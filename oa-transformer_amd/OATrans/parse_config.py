"""Experiment configuration for the entry points: JSON (or YAML) file + command-line overrides + a factory that builds
the objects a config section names.

Written from the CONTRACT of /root/reference/OATrans/parse_config_dist_multi.py:13-150 (what train*.py and the
trainers rely on), not from its text:

  ConfigParser(parser, options=(), timestamp=True, test=False)
      parser    an argparse.ArgumentParser carrying at least -c/--config, -r/--resume, -d/--device (train*.py)
      options   CustomArgs-like records (flags, type, target): each adds a CLI flag whose value, when given,
                replaces config[target[0]][target[1]]...
      -r <ckpt> re-reads the config.json stored beside the checkpoint; a -c file given as well is laid over it
      run directories  <trainer.save_dir>/{models,log,web}/<name>/<MMDD_HHMMSS>/ ; config.json is written into
                the models directory and logging is configured on the log directory (skipped with test=True)
  cfg[name], cfg.get(name), cfg.config, cfg.save_dir, cfg.log_dir, cfg.resume, cfg.args, cfg.get_logger(name, verbosity)
  cfg.initialize(name, module, *args, index=None, **kwargs)
      builds getattr(module, section['type'])(*args, **section['args']); section = cfg[name] or cfg[name][index];
      constructor parameters the section does not give are taken from top-level config keys of the same name; the
      parsed command line is passed as `args` to the three classes whose constructors take it; kwargs may add to a
      section's args but never override them.

A ready-made dict may be passed instead of an argparse parser (tests, notebooks): ConfigParser(config_dict).
"""
import inspect
import logging
import os
from datetime import datetime
from pathlib import Path
from types import SimpleNamespace

try:
    from OATrans.logger import setup_logging
    from OATrans.utils.util import read_json, write_json
except ImportError:                      # run with cwd = OATrans/ like the reference's entry points
    from logger import setup_logging
    from utils.util import read_json, write_json

# classes whose constructors receive the parsed command line as `args` (parse_config_dist_multi.py:93-98)
_TAKES_CLI_ARGS = frozenset({"FrozenInTime", "MultiDistTextObjectVideoDataLoader", "TextObjectVideoDataLoader"})
_LOG_LEVELS = {0: logging.WARNING, 1: logging.INFO, 2: logging.DEBUG}


def _load(path):
    path = Path(path)
    if path.suffix.lower() in (".yaml", ".yml"):
        import yaml
        with path.open() as fh:
            return yaml.safe_load(fh)
    return read_json(path)


def _dest_of(flags):
    """Attribute name argparse gives an option declared with `flags`: the first long flag, else the first flag."""
    long_flags = [f for f in flags if f.startswith("--")]
    return (long_flags[0] if long_flags else flags[0]).lstrip("-").replace("-", "_")


def _assign(tree, keys, value):
    node = tree
    for key in keys[:-1]:
        node = node[key]
    node[keys[-1]] = value


class ConfigParser:
    def __init__(self, args, options='', timestamp=True, test=False):
        options = list(options or ())
        if isinstance(args, dict):                         # programmatic use: the config itself
            self.args = SimpleNamespace(config=None, resume=None, device=None)
            self.resume, config = None, args
        else:
            for opt in options:
                args.add_argument(*opt.flags, default=None, type=opt.type)
            self.args = args.parse_args()
            if getattr(self.args, "device", None):
                os.environ["CUDA_VISIBLE_DEVICES"] = self.args.device
            if self.args.resume is not None:
                self.resume = Path(self.args.resume)
                config = read_json(self.resume.parent / "config.json")
                if self.args.config is not None:
                    config.update(_load(self.args.config))
            else:
                if self.args.config is None:
                    raise AssertionError("Configuration file need to be specified. Add '-c config.json', for example.")
                self.resume = None
                self.cfg_fname = Path(self.args.config)
                config = _load(self.cfg_fname)
            for opt in options:                            # command-line overrides of single config entries
                value = getattr(self.args, _dest_of(opt.flags), None)
                if value is not None:
                    _assign(config, opt.target, value)
        self._config = config

        root = Path(config["trainer"]["save_dir"])
        run = datetime.now().strftime("%m%d_%H%M%S") if timestamp else ""
        self._save_dir, self._log_dir, self._web_log_dir = (root / kind / config["name"] / run for kind in ("models", "log", "web"))
        self.log_levels = dict(_LOG_LEVELS)
        if not test:
            self._save_dir.mkdir(parents=True, exist_ok=True)
            self._log_dir.mkdir(parents=True, exist_ok=True)
            write_json(config, self._save_dir / "config.json")
            setup_logging(self._log_dir)

    # ---- factory
    def initialize(self, name, module, *args, index=None, **kwargs):
        section = self[name] if index is None else self[name][index]
        cls = getattr(module, section["type"])
        ctor_args = dict(section["args"])
        if index is None:
            clash = [k for k in kwargs if k in ctor_args]
            assert not clash, "Overwriting kwargs given in config file is not allowed"
            ctor_args.update(kwargs)
        for param in inspect.signature(cls.__init__).parameters:
            if param == "args" and section["type"] in _TAKES_CLI_ARGS:
                ctor_args["args"] = self.args
            elif param not in ctor_args and param in self._config:
                ctor_args[param] = self._config[param]
        return cls(*args, **ctor_args)

    # ---- read access
    def __getitem__(self, name):
        return self._config[name]

    def get(self, name, default=None):
        return self._config.get(name, default)

    def get_logger(self, name, verbosity=2):
        if verbosity not in self.log_levels:
            raise AssertionError(f"verbosity option {verbosity} is invalid. Valid options are {list(self.log_levels)}.")
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
