"""Dev: what streaming kernels reach on this part: torch copy (1 read + 1 write), fill (write only), sum (read only)."""
import torch
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / n * 1e-3
for mb in (154, 462, 1024, 4096):
    n = mb * (1 << 20) // 4
    a = torch.randn(n, device="cuda"); b = torch.empty_like(a)
    tc = t(lambda: b.copy_(a)); tf = t(lambda: b.fill_(1.0)); ts = t(lambda: a.sum())
    print(f"{mb:5d} MB: copy {2 * n * 4 / tc / 1e12:5.2f} TB/s   fill {n * 4 / tf / 1e12:5.2f} TB/s   sum(read) {n * 4 / ts / 1e12:5.2f} TB/s")
