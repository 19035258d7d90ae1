"""Reference point only (NOT used by the product): what the vendor BLAS reaches on the hot-path shapes."""
import torch
M = 50208
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for (n, k) in [(2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304)]:
    x = torch.randn(M, k, device="cuda").bfloat16(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).bfloat16(); b = torch.randn(n, device="cuda").bfloat16()
    t = timeit(lambda: torch.nn.functional.linear(x, w, b))
    dy = torch.randn(M, n, device="cuda").bfloat16()
    t2 = timeit(lambda: dy.t() @ x)
    print(f"hipBLASLt NT M={M} N={n} K={k}: {2*M*n*k/t/1e12:7.1f} TF/s ({t*1e6:.1f} us) | TN wgrad: {2*M*n*k/t2/1e12:7.1f} TF/s ({t2*1e6:.1f} us)")
