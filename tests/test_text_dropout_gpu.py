"""Training-mode dropout of the text tower (HF DistilBERT's three nn.Dropout sites; the reference leaves the tower in
train mode, oa_model.py:56) on a real MI355X.  torch's own mask stream cannot be reproduced, so parity is established
in two steps: (1) the device generator is Philox4x32-10 bit for bit (Random123 known answers + the numpy restatement
in the oracle) and the masks have the Bernoulli(1 - p) / (1 - p) distribution; (2) with the SAME masks handed to the
oracle, forward and backward of the tower match it (hidden states rel-L2 <= 2e-3, gradients norm <= 5e-2 / cosine >= 0.99)."""
import numpy as np
import pytest
import torch

from oracle import oatrans_oracle as orc

pytestmark = pytest.mark.gpu

KAT = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),                 # Random123 kat_vectors
       ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
       ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def test_device_philox_known_answers_and_oracle():
    from OATrans.ops import hip
    words = torch.tensor([list(c) + list(k) for c, k, _ in KAT], dtype=torch.int64, device="cuda")
    got = hip.philox4x32_10(words).cpu()
    assert got.tolist() == [list(w) for _, _, w in KAT]
    g = torch.Generator().manual_seed(3)
    rnd = torch.randint(0, 2 ** 32, (4096, 6), generator=g, dtype=torch.int64)
    want = orc.philox4x32_10(rnd[:, :4].numpy().astype(np.uint64), rnd[:, 4:].numpy().astype(np.uint64))
    assert np.array_equal(hip.philox4x32_10(rnd.cuda()).cpu().numpy().astype(np.uint64), want)


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_masks_equal_the_oracle_and_are_bernoulli(p):
    from OATrans.ops import hip
    n = 1 << 20
    st = hip.new_rng_state(0x1234567890abcdef, "cuda")
    hip.rng_tick(st)
    hip.rng_tick(st)
    assert st.tolist() == [0x1234567890abcdef, 2]
    m = hip.dropout_mask(n, p, st, site=5).cpu()
    assert torch.equal(m, orc.dropout_multipliers(n, p, 0x1234567890abcdef, 2, 5))
    kept = (m != 0)
    assert torch.all(m[kept] == np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
    sigma = (p * (1 - p) / n) ** 0.5
    assert abs((~kept).float().mean().item() - p) < 5 * sigma
    other = hip.dropout_mask(n, p, st, site=6).cpu()                     # another site: independent draws
    both = ((m == 0) & (other == 0)).float().mean().item()
    assert abs(both - p * p) < 5 * (p * p * (1 - p * p) / n) ** 0.5
    x = torch.randn(64, 768, device="cuda")
    r = torch.randn(64, 768, device="cuda")
    o32 = torch.empty_like(x)
    o16 = torch.empty(64, 768, device="cuda", dtype=torch.bfloat16)
    hip.dropout(x, 64, 768, p, st, 5, resid=r, out32=o32, out16=o16)
    want = x.cpu() * m[:64 * 768].view(64, 768) + r.cpu()
    assert torch.equal(o32.cpu(), want) and torch.equal(o16.cpu(), want.bfloat16())


def _text_tower(n_layers=2):
    from OATrans.model.text_transformer import DistilBertHIP
    torch.manual_seed(0)
    txt = DistilBertHIP(dict(vocab_size=1000, max_position_embeddings=128, n_layers=n_layers, n_heads=12, dim=768, hidden_dim=3072))
    with torch.no_grad():
        for n, prm in txt.named_parameters():
            if "LayerNorm.weight" in n or "layer_norm.weight" in n:
                prm.normal_(1.0, 0.1)
            elif prm.dim() == 1:
                prm.normal_(0, 0.05)
    return txt.cuda()


@pytest.mark.parametrize("B,L,pad", [(4, 9, 6), (2, 70, 41), (3, 33, 32)])
def test_training_mode_tower_matches_oracle_given_the_same_masks(B, L, pad):
    """L = 9: one workgroup per (b, h), a sequence length that is no multiple of the four elements a Philox draw covers;
    L = 70: three 32-row workgroups per (b, h) and two LDS chunks of keys / queries (64 + 6); L = 33: a one-row last chunk"""
    txt = _text_tower()
    txt.train()
    txt.set_dropout_seed(777)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1, 1000, (B, L), generator=g)
    mask = torch.ones(B, L, dtype=torch.int64)
    mask[1, pad:] = 0
    dout = torch.randn(B, L, 768, generator=g)
    dout[1, pad:] = 0                                     # padded positions: don't-care rows
    txt.begin_step()
    h = txt(input_ids=ids.cuda(), attention_mask=mask.cuda()).last_hidden_state
    (h * dout.cuda()).sum().backward()
    torch.cuda.synchronize()
    state = txt._rng_state.tolist()
    assert state == [777, 1]
    sd = {"text_model." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in txt.state_dict().items()}
    masks = orc.distilbert_dropout_masks(B, L, 768, 12, 2, 0.1, 0.1, state[0], state[1])
    oh = orc.distilbert(ids, mask, sd, n_heads=12, dropout=masks)
    (oh * dout).sum().backward()
    keep = mask.bool()
    rel = ((h.detach().cpu()[keep] - oh.detach()[keep]).norm() / oh.detach()[keep].norm()).item()
    eval_h = orc.distilbert(ids, mask, {k: v.detach() for k, v in sd.items()}, n_heads=12)
    moved = ((oh.detach()[keep] - eval_h[keep]).norm() / eval_h[keep].norm()).item()
    print("train-mode hidden rel err", rel, "; dropout moved the output by", moved)
    assert moved > 0.05 and rel < 2e-3
    bad = []
    for n, prm in txt.named_parameters():
        r = sd["text_model." + n].grad
        if r is None or r.norm() < 1e-6 or n.endswith("k_lin.bias"):
            continue                     # a key bias shifts every score of a row alike: its true gradient is 0 (noise both sides)
        gg = prm.grad.detach().float().cpu().flatten()
        r = r.flatten()
        nerr = abs(gg.norm() - r.norm()).item() / r.norm().item()
        cos = torch.dot(gg, r).item() / (gg.norm().item() * r.norm().item())
        if nerr > 5e-2 or cos < 0.99:
            bad.append((n, nerr, cos))
    assert not bad, bad[:8]


def test_eval_is_identity_masks_advance_and_reseed_repeats():
    txt = _text_tower(n_layers=1)
    ids = torch.randint(1, 1000, (3, 8), device="cuda")
    mask = torch.ones(3, 8, dtype=torch.int64, device="cuda")

    def run():
        with torch.no_grad():
            return txt(input_ids=ids, attention_mask=mask).last_hidden_state.clone()
    txt.eval()
    e1, e2 = run(), run()
    assert torch.equal(e1, e2) and txt._rng_state is None          # eval mode never touches the generator
    txt.train()
    txt.set_dropout_seed(5)
    a, b = run(), run()
    assert not torch.equal(a, b) and not torch.equal(a, e1)        # new masks on every forward
    txt.set_dropout_seed(5)
    a2, b2 = run(), run()
    assert torch.equal(a, a2) and torch.equal(b, b2)               # same seed, same call sequence: same masks
    txt.config.dropout = txt.config.attention_dropout = 0.0
    assert torch.equal(run(), e1)                                  # rates 0 in train mode: identity as well


def test_graph_replays_draw_new_masks():
    """The captured training step reads the rng state from device memory, so every replay ticks it and draws new masks."""
    import argparse
    from OATrans import model as module_arch
    from OATrans.optim import AdamW
    from OATrans.parallel import HipDataParallel
    from OATrans.trainer.graph_step import GraphedStep
    from OATrans.trainer.step import hot_step
    from tests.test_model_gpu import _batch, _small_frozen
    m = _small_frozen(seed=3, depth=2)
    m.train()
    m.text_model.set_dropout_seed(11)
    for sub in (m.video_model, m.text_model):
        sub.flatten_parameters()
    dp = HipDataParallel(m)
    opt = AdamW([p for p in m.parameters() if p.requires_grad], lr=0.0)          # lr 0: the only thing that changes is the mask
    sa = argparse.Namespace(world_size=1, rank=0, local_rank=0)
    step = GraphedStep(hot_step, dp, module_arch.NormSoftmaxLoss(), opt, sa, warmup=2)
    b = _batch(seed=21)
    losses = [step(b).item() for _ in range(6)]
    torch.cuda.synchronize()
    assert step.replays >= 3
    assert m.text_model._rng_state.tolist() == [11, 6]
    assert len(set(losses)) == 6, losses
