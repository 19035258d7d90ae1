"""Dev experiment: does ln_bwd (118 VGPRs, 32 KB LDS) co-reside with a CU-filling kernel of another stream?"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/libspin.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "spin.hip"), "-o", so])
spin = ctypes.CDLL(so)
M, D = 50208, 768
x = torch.randn(M, D, device="cuda"); g = torch.ones(D, device="cuda")
mean = torch.zeros(M, device="cuda"); rstd = torch.ones(M, device="cuda")
dy = torch.randn(M, D, device="cuda").bfloat16(); G = torch.randn(M, D, device="cuda"); dx16 = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda"); out = torch.zeros(16, device="cuda")
side = torch.cuda.Stream()
def ln(): hip.layernorm_bwd(dy, x, mean, rstd, g, M, D, dx=G, dx16=dx16, dres=G, dgamma=dg, dbeta=db, accumulate=False)
def timed(fn):
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record(); fn(); en.record(); torch.cuda.synchronize(); return st.elapsed_time(en) * 1e3
for _ in range(3): ln()
torch.cuda.synchronize()
print("ln_bwd alone us", min(timed(ln) for _ in range(5)))
for vg, blocks, lds in ((192, 256, 128 * 1024), (232, 256, 128 * 1024), (128, 256, 128 * 1024), (192, 192, 128 * 1024), (232, 192, 128 * 1024)):
    iters = 40000
    def spin_only():
        spin.spin_launch(vg, blocks, lds, iters, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(side.cuda_stream))
    spin_only(); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record(side); spin_only(); en.record(side); torch.cuda.synchronize()
    t_spin = st.elapsed_time(en) * 1e3
    res = []
    for _ in range(3):
        spin_only()
        torch.cuda._sleep(200000)           # let the spin kernel occupy the CUs first
        res.append(timed(ln))
        torch.cuda.synchronize()
    print(f"spin vgpr={vg} blocks={blocks}: spin alone {t_spin:.0f} us; ln_bwd beside it {min(res):.0f} .. {max(res):.0f} us")
