"""Losses of the hot path.  Only NormSoftmaxLoss is used by the shipped configs
(/root/reference/OATrans/model/loss.py:7-25); the margin / cross-entropy variants of the
reference are dead code there and are not provided."""
from torch import nn

from ..ops import hip
from .layers import _NormSoftmaxFn


class NormSoftmaxLoss(nn.Module):
    def __init__(self, temperature=0.05):
        super().__init__()
        self.temperature = temperature

    def forward(self, x):
        """x: square cosine-similarity matrix; symmetric InfoNCE over rows and columns."""
        if x.shape[0] != x.shape[1]:
            raise ValueError("NormSoftmaxLoss needs a square similarity matrix (diagonal = positives)")
        if not x.is_cuda:
            raise hip.OatError("NormSoftmaxLoss runs on MI355X only (no CPU path); use the oracle for CPU")
        return _NormSoftmaxFn.apply(x, self.temperature)
