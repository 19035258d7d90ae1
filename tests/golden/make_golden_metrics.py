"""Golden vectors for the retrieval metrics, produced by the reference's own
OATrans/model/metric.py (imported by file path; shims: empty `ipdb`, `np.bool = bool` - the reference predates
numpy 1.24).

    python tests/golden/make_golden_metrics.py      # needs /root/reference; writes tests/golden/metrics.pt"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 20240917


def main():
    sys.modules.setdefault("ipdb", types.ModuleType("ipdb"))
    if not hasattr(np, "bool"):
        np.bool = bool
    spec = importlib.util.spec_from_file_location("ref_metric", "/root/reference/OATrans/model/metric.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.RandomState(SEED)
    cases = []

    def add(name, sims, masks=None):
        t2v = ref.t2v_metrics(sims.copy(), None if masks is None else masks.copy())
        v2t = ref.v2t_metrics(sims.copy(), None if masks is None else masks.copy())
        cases.append(dict(name=name, sims=torch.from_numpy(sims.copy()), masks=None if masks is None else torch.from_numpy(masks.copy()),
                          t2v={k: float(v) for k, v in t2v.items()}, v2t={k: float(v) for k, v in v2t.items()}))

    add("random_12", rng.randn(12, 12))
    add("diag_strong_64", np.eye(64) * 3 + rng.randn(64, 64))
    add("ties_int_10", rng.randint(0, 3, size=(10, 10)).astype(np.float64))
    add("constant_6", np.zeros((6, 6)))
    add("multi_caption_20x5", rng.randn(20, 5))
    m = np.ones((5, 4))
    m[1, 3] = 0
    m[4, 2:] = 0
    add("multi_caption_masked_20x5", rng.randn(20, 5), m)
    torch.save(cases, os.path.join(HERE, "metrics.pt"))
    for c in cases:
        print(c["name"], c["t2v"], c["v2t"])


if __name__ == "__main__":
    main()
