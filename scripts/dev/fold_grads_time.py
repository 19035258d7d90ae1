"""Dev: oat_ln_fold_grads alone at a ViT block's three folded layers (qkv, qkv, fc1 at D = 768)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
torch.manual_seed(0)
ent = []
for N in (2304, 2304, 3072):
    K = 768
    r = lambda *s: torch.randn(*s, device="cuda")
    dWp, dbp, W, g, b = r(N, K), r(N), r(N, K), r(K), r(K)
    ent.append((dWp, dbp, W, g, b, torch.empty_like(dWp), dbp, torch.zeros(K, device="cuda"), torch.zeros(K, device="cuda"), False))
tab = hip.FoldGradTable(ent)
for _ in range(5): tab.run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50): tab.run()
e.record(); torch.cuda.synchronize()
print("ln_fold_grads: %.1f us per launch" % (s.elapsed_time(e) / 50 * 1e3))
for dWp, dbp, W, g, b, dW, db, dg, dbt, _ in ent:
    assert torch.allclose(dW, dWp * g + dbp[:, None] * b, rtol=1e-5, atol=1e-5)
    assert torch.allclose(dg, (W * dWp).sum(0), rtol=1e-4, atol=1e-2) and torch.allclose(dbt, (W * dbp[:, None]).sum(0), rtol=1e-4, atol=1e-2)
print("ok")
