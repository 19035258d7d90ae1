"""The helpers the entry points and the model import from OATrans.utils
(/root/reference/OATrans/utils/util.py:14-50, :75-95)."""
import json
from collections import OrderedDict
from itertools import repeat
from pathlib import Path


def replace_nested_dict_item(obj, key, replace_value):
    for k, v in obj.items():
        if isinstance(v, dict):
            obj[k] = replace_nested_dict_item(v, key, replace_value)
    if key in obj:
        obj[key] = replace_value
    return obj


def state_dict_data_parallel_fix(load_state_dict, curr_state_dict):
    """Strip / add the `module.` prefix so (Distributed)DataParallel checkpoints load either way."""
    load_keys, curr_keys = list(load_state_dict.keys()), list(curr_state_dict.keys())
    if not curr_keys or not load_keys:
        return load_state_dict
    cur_dp, load_dp = curr_keys[0].startswith('module.'), load_keys[0].startswith('module.')
    if load_dp and not cur_dp:
        return OrderedDict((k[len('module.'):], v) for k, v in load_state_dict.items())
    if cur_dp and not load_dp:
        return OrderedDict(('module.' + k, v) for k, v in load_state_dict.items())
    return load_state_dict


def read_json(fname):
    with Path(fname).open('rt') as handle:
        return json.load(handle, object_hook=OrderedDict)


def write_json(content, fname):
    with Path(fname).open('wt') as handle:
        json.dump(content, handle, indent=4, sort_keys=False)


def inf_loop(data_loader):
    for loader in repeat(data_loader):
        yield from loader
