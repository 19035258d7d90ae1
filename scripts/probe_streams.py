"""Dev probe: does work enqueued on the default stream AFTER a long chain of tiny kernels on a side stream wait for that chain?"""
import time, torch
x0 = torch.zeros(1024, device="cuda"); x1 = torch.zeros(1024, device="cuda")
big = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
def run(n_side, wait_first, label):
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t_host = time.perf_counter()
    es = torch.cuda.Event(enable_timing=True)
    for _ in range(20): _ = big @ big   # keep the GPU busy ~20 ms so that the host finishes every enqueue below first
    e0.record(main)
    if wait_first: side.wait_stream(main)
    with torch.cuda.stream(side):
        for _ in range(n_side): x1.add_(1.0)
        es.record(side)
    x0.add_(1.0)                       # main-stream kernel enqueued AFTER the side chain
    e1.record(main)
    th = time.perf_counter() - t_host
    torch.cuda.synchronize()
    print(f"host enqueue {th*1e3:6.2f} ms | " f"{label:40s} side chain done at {e0.elapsed_time(es)*1e3:8.1f} us, main kernel done at {e0.elapsed_time(e1)*1e3:8.1f} us")
for rep in range(2):
    run(300, False, "300 tiny side kernels, no wait")
    run(300, True, "300 tiny side kernels, side waits main")
    run(0, True, "no side kernels")
