import os, sys
sys.path.insert(0, "oa-transformer_amd"); sys.path.insert(0, ".")
import torch
from OATrans.ops import hip
def t(fn, n=50):
    fn(); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e3
for (G, R, D) in ((8, 196, 768), (1, 196, 768), (1, 32, 768), (16, 441, 768)):
    x = torch.randn(G * R, D, device="cuda"); o = torch.zeros(G, D, device="cuda")
    print(G, R, D, f"{t(lambda: hip.grouped_rowsum(x, G, R, D, o)):.1f} us")
