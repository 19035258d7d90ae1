"""Retrieval metrics on a similarity matrix (numpy, CPU; validation side of the path).
Behaviour of /root/reference/OATrans/model/metric.py:16-120,281-291 for the one-caption-per-video
case used by the shipped configs: rows = text queries, columns = videos, diagonal = ground truth."""
import numpy as np


def cols2metrics(cols, num_queries):
    m = {}
    for k in (1, 5, 10, 50):
        m[f"R{k}"] = 100 * float(np.sum(cols < k)) / num_queries
    m["MedR"] = float(np.median(cols) + 1)
    m["MeanR"] = float(np.mean(cols) + 1)
    m["geometric_mean_R1-R5-R10"] = float(np.cbrt(m["R1"] * m["R5"] * m["R10"]))
    return m


def _ranks(sims):
    """Rank (0 = best) of the diagonal entry in every row; ties broken optimistically-averaged like
    the reference's 'break ties by averaging' branch."""
    sims = np.asarray(sims, dtype=np.float64)
    gt = np.diag(sims)[:, None]
    better = (sims > gt).sum(axis=1)
    ties = (sims == gt).sum(axis=1) - 1
    return better + ties / 2.0


def t2v_metrics(sims, query_masks=None):
    assert sims.ndim == 2 and sims.shape[0] == sims.shape[1], "one caption per video expected"
    return cols2metrics(_ranks(sims), sims.shape[0])


def v2t_metrics(sims, query_masks=None):
    assert sims.ndim == 2 and sims.shape[0] == sims.shape[1], "one caption per video expected"
    return cols2metrics(_ranks(sims.T), sims.shape[0])
