"""Golden vectors for checkpoint interop, produced by the reference's own
`FrozenInTime._inflate_positional_embeds` (model/oa_model.py:148-189, lifted with `ast` and run on a stand-in `self`)
and `state_dict_data_parallel_fix` (utils/util.py:24-50).

    python tests/golden/make_golden_ckpt.py      # needs /root/reference; writes tests/golden/ckpt_interop.pt"""
import ast
import os
from collections import OrderedDict

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 20240917


def lift(path, name, ns):
    with open(path) as fh:
        tree = ast.parse(fh.read())
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == name:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
            return ns[name]
    raise KeyError(name)


class Self:
    def __init__(self, frames, fix, sd):
        self.video_params = {"model": "SpaceTimeTransformer", "num_frames": frames}
        self.load_temporal_fix = fix
        self._sd = sd

    def state_dict(self):
        return self._sd


def main():
    inflate = lift("/root/reference/OATrans/model/oa_model.py", "_inflate_positional_embeds", {"torch": torch, "F": F})
    fix = lift("/root/reference/OATrans/utils/util.py", "state_dict_data_parallel_fix", {"OrderedDict": OrderedDict})
    g = torch.Generator().manual_seed(SEED)
    D = 32
    cases = []
    for frames, n_load, mode in ((4, 2, "zeros"), (4, 6, "zeros"), (8, 3, "interp"), (8, 3, "bilinear"), (4, 4, "zeros"), (5, 1, "interp")):
        te = torch.randn(1, n_load, D, generator=g)
        cur = {"video_model.temporal_embed": torch.zeros(1, frames, D), "video_model.pos_embed": torch.zeros(1, 10, D)}
        out = inflate(Self(frames, mode, cur), {"video_model.temporal_embed": te.clone(), "video_model.pos_embed": torch.zeros(1, 10, D)})
        cases.append(dict(frames=frames, mode=mode, load=te, out=out["video_model.temporal_embed"].clone()))
    fixes = []
    for load_keys, cur_keys in ((["module.a.w", "module.b.w"], ["a.w", "b.w"]), (["a.w", "b.w"], ["module.a.w", "module.b.w"]),
                                (["a.w"], ["a.w", "b.w"]), (["module.a.w"], ["module.a.w"])):
        out = fix(OrderedDict((k, i) for i, k in enumerate(load_keys)), OrderedDict((k, 0) for k in cur_keys))
        fixes.append(dict(load=load_keys, cur=cur_keys, out=list(out.items())))
    torch.save(dict(inflate=cases, prefix_fix=fixes), os.path.join(HERE, "ckpt_interop.pt"))
    print([(c["frames"], c["mode"], tuple(c["out"].shape)) for c in cases], fixes)


if __name__ == "__main__":
    main()
