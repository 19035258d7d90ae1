"""Dev probe: operand layout and scale encoding of v_mfma_scale_f32_16x16x128_f8f6f4 (OCP e4m3), and the packed
f32 -> fp8 conversion, against torch.float8_e4m3fn."""
import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libprobe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "probe.hip"), "-o", so])
lib = ctypes.CDLL(so)
torch.manual_seed(0)
A = (torch.randn(16, 128, device="cuda")).to(torch.float8_e4m3fn)
B = (torch.randn(16, 128, device="cuda") * torch.arange(1, 17, device="cuda")[:, None] / 8).to(torch.float8_e4m3fn)   # asymmetric
C = torch.zeros(16, 16, device="cuda")
ref = A.float() @ B.float().t()
for sa, sb in ((127, 127), (128, 127), (127, 126), (0x7f7f7f7f, 0x7f7f7f7f)):
    lib.run(ctypes.c_void_p(A.data_ptr()), ctypes.c_void_p(B.data_ptr()), ctypes.c_void_p(C.data_ptr()), sa, sb)
    torch.cuda.synchronize()
    print(f"scale_a={sa:#x} scale_b={sb:#x}: max|C - ref| = {(C - ref).abs().max().item():.4g}, max|C - 2ref| = {(C - 2 * ref).abs().max().item():.4g}, "
          f"max|C - ref/2| = {(C - ref / 2).abs().max().item():.4g}, max|C-ref^T| = {(C - ref.t()).abs().max().item():.4g}  |ref|max {ref.abs().max().item():.3g}")
x = torch.randn(4096, device="cuda") * 3
y = torch.empty(4096, dtype=torch.uint8, device="cuda")
lib.run_cvt(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), 4096)
torch.cuda.synchronize()
want = x.to(torch.float8_e4m3fn).view(torch.uint8)
print("cvt_pk_fp8_f32 == torch e4m3fn:", torch.equal(y, want), (y != want).sum().item())
big = torch.tensor([1000.0, -1000.0, 448.0, 449.0, 464.0, 480.0, float("inf"), 1e-10], device="cuda")
yb = torch.empty(8, dtype=torch.uint8, device="cuda")
lib.run_cvt(ctypes.c_void_p(big.data_ptr()), ctypes.c_void_p(yb.data_ptr()), 8)
torch.cuda.synchronize()
print("saturation:", yb.view(torch.float8_e4m3fn).float().tolist(), "torch:", big.to(torch.float8_e4m3fn).float().tolist())
