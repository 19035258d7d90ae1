"""Dev: what does a kernel BOUNDARY cost on the GPU?  Sequences replayed from a launch tape (host cost out of the way):
A = N768/K768 GEMM, L = add+LayerNorm forward over the same rows, E = a stream edge pair (main -> side -> main, the side
stream gets a tiny kernel).  time(seq) - sum of the members' own back-to-back times = the cost of switching."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip

M = 50208; Mp = (M + 255) // 256 * 256; D = 768
dev = "cuda"
A = torch.randn(Mp, D, device=dev).bfloat16(); W = (torch.randn(D, D, device=dev) * D ** -0.5).bfloat16()
bias = torch.randn(D, device=dev); out = torch.zeros(Mp, D, device=dev, dtype=torch.bfloat16)
x = torch.randn(Mp, D, device=dev); s32 = torch.empty_like(x); y = torch.empty(Mp, D, device=dev, dtype=torch.bfloat16)
g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev); mean = torch.empty(Mp, device=dev); rstd = torch.empty(Mp, device=dev)
small = torch.zeros(1024, device=dev); small2 = torch.zeros(1024, device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()

def opA(): hip.gemm_nt(A, W, M, D, D, hip.EPI_BF16, out, bias=bias)
def opL(): hip.add_layernorm_fwd(x, out, s32, g, b, M, D, 1e-6, y=y, mean=mean, rstd=rstd)
def opS(): hip.axpby(small, small, small2, 1.0, 0.0)          # a tiny kernel
def opE():
    hip.stream_edge(main, side)
    with torch.cuda.stream(side):
        opS()
    hip.stream_edge(side, main)
def opH():                                                     # half an edge: the side stream waits for main, main never waits
    hip.stream_edge(main, side)
    with torch.cuda.stream(side):
        opS()

OPS = {"A": opA, "L": opL, "S": opS, "E": opE, "H": opH}

def tape_of(seq, reps):
    for c in seq: OPS[c]()                 # warm
    torch.cuda.synchronize()
    hip.tape_begin()
    for _ in range(reps):
        for c in seq: OPS[c]()
    return hip.tape_end()

def time_seq(seq, reps=20, rounds=5):
    t = tape_of(seq, reps)
    ts = []
    for _ in range(rounds):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        st.record(); hip.tape_replay(t); en.record(); torch.cuda.synchronize()
        ts.append(st.elapsed_time(en) / reps * 1e3)
    hip.tape_free(t)
    return sorted(ts)[len(ts) // 2]

res = {}
for seq in ["A", "L", "S", "AL", "AS", "LS", "ALS", "AE", "AEL", "AH", "AHL", "AAL", "ALL"]:
    res[seq] = time_seq(seq)
    own = sum(res[c] for c in seq if c in res and c in "ALS")
    print(f"{seq:5s} {res[seq]:8.2f} us per repetition" + (f"   members alone {own:8.2f}" if len(seq) > 1 and all(c in "ALS" for c in seq) else ""), flush=True)

# ---- does a second ACTIVE queue lengthen the boundaries of the main queue?  The side stream runs a chain of tiny kernels
# (no edges to main) while main replays A L A L ...
def busy_side(n):
    with torch.cuda.stream(side):
        for _ in range(n): opS()
for seq in ["AL", "A", "ALS"]:
    t = tape_of(seq, 20)
    ts = {0: [], 1: []}
    for r in range(5):
        for withside in (0, 1):
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            if withside: busy_side(1500)          # ~1500 x (3 us kernel + launch) keeps the side queue busy for the whole replay
            st.record(); hip.tape_replay(t); en.record(); torch.cuda.synchronize()
            ts[withside].append(st.elapsed_time(en) / 20 * 1e3)
    hip.tape_free(t)
    print(f"{seq:4s} main alone {sorted(ts[0])[2]:8.2f}   with a busy side queue {sorted(ts[1])[2]:8.2f}", flush=True)
