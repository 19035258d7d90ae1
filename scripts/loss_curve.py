import sys, os, argparse, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oa-transformer_amd"))
import bench
args = argparse.Namespace(variant="frozen", frames=8, batch=32, res=224, lr=float(os.environ.get("LR", "2e-5")))
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
dp, opt, loss_fn = bench.build(args, dev)
data = bench.synthetic_batch(args, 0, dev)
from OATrans.trainer.step import hot_step
sa = argparse.Namespace(world_size=1, rank=0, local_rank=0)
print([round(hot_step(dp, loss_fn, opt, data, sa).item(), 3) for _ in range(16)])
