"""Retrieval metrics on a similarity matrix (numpy, CPU; validation side of the path).

Same results as /root/reference/OATrans/model/metric.py for every input its t2v_metrics (:16-120), v2t_metrics
(:123-212) and cols2metrics (:281-291) accept, including their two DIFFERENT tie rules, several captions per video
and query masks; pinned by tests/golden/metrics.pt (generated from the reference module itself).
Rows of `sims` are text queries, columns are videos; with q captions per video, rows [j*q, (j+1)*q) belong to video j."""
import numpy as np


def cols2metrics(cols, num_queries):
    cols = np.asarray(cols, dtype=np.float64)
    m = {"R1": 100 * float(np.sum(cols == 0)) / num_queries}          # an averaged tie rank of 0.5 is NOT a hit
    for k in (5, 10, 50):
        m[f"R{k}"] = 100 * float(np.sum(cols < k)) / num_queries
    m["MedR"] = float(np.median(cols) + 1)
    m["MeanR"] = float(np.mean(cols) + 1)
    with np.errstate(divide="ignore"):
        m["geometric_mean_R1-R5-R10"] = float(np.exp(np.mean(np.log([m["R1"], m["R5"], m["R10"]]))))   # scipy gmean
    return m


def t2v_metrics(sims, query_masks=None):
    """Text-to-video: rank of the ground-truth video among all videos, ties broken OPTIMISTICALLY (:62)."""
    sims = np.asarray(sims)
    assert sims.ndim == 2, "expected a matrix"
    num_queries, num_vids = sims.shape
    per_video = num_queries // num_vids
    dists = -sims
    gt = dists[np.arange(num_queries), np.arange(num_queries) // per_video][:, None]
    cols = (dists < gt).sum(axis=1)                    # first position of the GT distance in the sorted row
    if query_masks is not None:
        query_masks = np.asarray(query_masks)
        assert query_masks.size == num_queries, "invalid query mask shape"
        cols = cols[query_masks.reshape(-1).astype(bool)]
        num_queries = query_masks.sum()
    return cols2metrics(cols, num_queries)


def v2t_metrics(sims, query_masks=None):
    """Video-to-text: best rank over the video's own captions, ties broken by AVERAGING (:152); masked captions are
    pushed to the end of every ranking (:161)."""
    sims = np.asarray(sims).T
    assert sims.ndim == 2, "expected a matrix"
    num_queries, num_caps = sims.shape
    per_video = num_caps // num_queries
    dists = -sims.astype(np.float64)
    MISSING_VAL = 1E8
    if query_masks is not None:
        dists[:, np.logical_not(np.asarray(query_masks).reshape(-1))] = MISSING_VAL
    ranks = np.full(num_queries, np.inf)
    for ii in range(num_queries):
        row = dists[ii]
        for jj in range(ii * per_video, (ii + 1) * per_video):
            if row[jj] == MISSING_VAL:
                continue
            rank = (row < row[jj]).sum() + ((row == row[jj]).sum() - 1) / 2.0      # mean position of the tied block
            ranks[ii] = min(ranks[ii], rank)
    return cols2metrics(ranks, num_queries)
