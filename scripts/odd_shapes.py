import sys, os, argparse, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oa-transformer_amd"))
from OATrans import model as module_arch
from OATrans.optim import AdamW
from OATrans.parallel import HipDataParallel
from OATrans.trainer.step import hot_step
torch.manual_seed(0)
m = module_arch.FrozenInTime(
    video_params=dict(model="SpaceTimeTransformer", arch_config="base_patch16_224", num_frames=3, pretrained=True, time_init="rand", arch_kwargs=dict(depth=2)),
    object_params=dict(model="", input_objects=False),
    text_params=dict(model="pretrained/distilbert-base-uncased", pretrained=True, input="text", config=dict(n_layers=1)),
    projection="minimal", load_checkpoint="").cuda()
m.set_device(torch.device("cuda"))
for sub in (m.video_model, m.text_model): sub.flatten_parameters()
dp = HipDataParallel(m); opt = AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-5)
sa = argparse.Namespace(world_size=1, rank=0, local_rank=0)
for B, T, L in ((1, 3, 5), (3, 2, 9), (5, 1, 32), (2, 3, 1)):
    g = torch.Generator().manual_seed(B)
    data = {"video": torch.randn(B, T, 3, 224, 224, generator=g).cuda(),
            "text": {"input_ids": torch.randint(1000, 30000, (B, L), generator=g).cuda(), "attention_mask": torch.ones(B, L, dtype=torch.int64).cuda()}}
    ls = [hot_step(dp, module_arch.NormSoftmaxLoss(), opt, data, sa).item() for _ in range(2)]
    print(B, T, L, ls, all(torch.isfinite(p).all().item() for p in m.parameters()))
