"""One tiny hot-path step on cuda:0, checked against the CPU oracle (driver smoke test).
Imports oracle/ as the CHECKER only (allowed for smoke()); the step itself is the HIP path."""
import argparse

import torch


def smoke_step():
    from OATrans.model.layers import HipLinear, ReLULinear
    from OATrans.model.loss import NormSoftmaxLoss
    from OATrans.model.text_transformer import DistilBertHIP
    from OATrans.model.video_transformer import SpaceTimeTransformer
    from OATrans.model.layers import sim_matrix
    from OATrans.optim import AdamW
    from OATrans.utils import seeded_init as si
    from oracle import oatrans_oracle as orc

    seed = 31337
    vshape = dict(embed_dim=128, depth=2, mlp_ratio=4, num_frames=2, patches_per_frame=4, patch=16)
    tshape = dict(dim=128, n_layers=2, hidden_dim=512, vocab=500, max_pos=32)
    sd = si.frozen_state_dict(seed, vshape, tshape, proj_dim=256)
    dev = torch.device("cuda:0")
    txt = DistilBertHIP(dict(vocab_size=500, max_position_embeddings=32, n_layers=2, n_heads=2, dim=128, hidden_dim=512))
    txt.eval()                      # parity against the (eval-mode) oracle: training-mode dropout off
    txt.load_state_dict({k[11:]: v for k, v in sd.items() if k.startswith("text_model.")})
    vid = SpaceTimeTransformer(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2, num_frames=2, time_init="rand")
    vid.head = torch.nn.Identity()
    vid.load_state_dict({k[12:]: v for k, v in sd.items() if k.startswith("video_model.")}, strict=False)
    vid.need_patch_tokens = False
    tp, vp = ReLULinear(128, 256), HipLinear(128, 256)      # 256-d projection as in the shipped configs
    tp[1].load_state_dict({"weight": sd["txt_proj.1.weight"], "bias": sd["txt_proj.1.bias"]})
    vp.load_state_dict({"weight": sd["vid_proj.0.weight"], "bias": sd["vid_proj.0.bias"]})
    txt, vid, tp, vp = txt.to(dev), vid.to(dev), tp.to(dev), vp.to(dev)
    video = si.seeded_tensor(seed, "smoke.video", (4, 2, 3, 32, 32))
    ids = si.seeded_ints(seed, "smoke.ids", (4, 6), 1, 500)
    mask = torch.ones(4, 6, dtype=torch.int64)
    mask[2, 4:] = 0
    params = [p for m in (txt, vid, tp, vp) for p in m.parameters()]
    txt.begin_step()
    t = tp(txt(input_ids=ids.to(dev), attention_mask=mask.to(dev)).last_hidden_state[:, 0])
    v = vp(vid(video.to(dev))[0])
    sim = sim_matrix(t, v)
    loss = NormSoftmaxLoss()(sim)
    loss.backward()
    opt = AdamW(params, lr=1e-3)
    opt.step()
    torch.cuda.synchronize()
    oloss, osim, _, _ = orc.train_step_loss(sd, video, ids, mask, num_heads=2, text_heads=2)
    err = (sim.detach().cpu() - osim).abs().max().item()
    assert err <= 1e-3, f"smoke: sim-matrix error {err} vs CPU oracle (stated bound 1e-3)"
    assert abs(loss.item() - oloss.item()) <= 5e-2 * max(1.0, abs(oloss.item())), (loss.item(), oloss.item())
    assert all(torch.isfinite(p).all() for p in params)
    print(f"smoke ok: loss {loss.item():.4f} (oracle {oloss.item():.4f}), sim max-abs err {err:.2e}")
