"""Dev (GPU): does a CU-masked HIP stream (hipExtStreamCreateWithCUMask) confine kernels on this part, and how do mask
bits map to CUs?  Times one bandwidth-bound kernel (LayerNorm, 6276 workgroups) on masked streams of various widths."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
rt = ctypes.CDLL("libamdhip64.so")
rt.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
def masked_stream(words):
    h = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = rt.hipExtStreamCreateWithCUMask(ctypes.byref(h), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value)
M, D = 50208, 768
x = torch.randn(M, D, device="cuda").bfloat16(); y = torch.empty_like(x); mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
A = torch.randn(4096, 768, device="cuda").bfloat16(); W = torch.randn(768, 768, device="cuda").bfloat16(); o = torch.empty(4096, 768, device="cuda", dtype=torch.bfloat16)
def timeit(stream, fn, n=10):
    with torch.cuda.stream(stream):
        for _ in range(3): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(stream)
        for _ in range(n): fn()
        e.record(stream)
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
ln = lambda: hip.layernorm_fwd_r16(x, M, D, 1e-6, y=y, mean=mean, rstd=rstd)
print("unmasked stream: LN %.1f us" % timeit(torch.cuda.Stream(), ln))
for name, words in [("all 256 bits", [0xffffffff] * 8), ("bits 0-127", [0xffffffff] * 4 + [0] * 4), ("bits 0-31", [0xffffffff] + [0] * 7),
                    ("bits 0-15", [0xffff] + [0] * 7), ("bits 0-7", [0xff] + [0] * 7), ("bit 0", [1] + [0] * 7),
                    ("bits 0,8,16,..,120 (16 bits)", [0x01010101] * 4 + [0] * 4), ("bits 240-255", [0] * 7 + [0xffff0000])]:
    try:
        st = masked_stream(words)
        print(f"{name:32s}: LN {timeit(st, ln):8.1f} us")
    except Exception as ex:
        print(name, "failed:", ex)
