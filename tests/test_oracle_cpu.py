"""CPU checks of oracle pieces that have no reference golden of their own: the Philox generator (Random123's published
known-answer vectors) and the dropout masks built on it."""
import torch

from oracle import oatrans_oracle as orc


def test_philox_known_answers_and_dropout_multipliers():
    """The oracle's Philox4x32-10 against Random123's published known-answer vectors, and the dropout multipliers built
    on it: values in {0, 1/(1-p)}, drop rate p, all-ones masks reproduce the eval-mode DistilBERT."""
    import numpy as np
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, w in kat:
        got = orc.philox4x32_10(np.array(c, dtype=np.uint64), np.array(k, dtype=np.uint64))
        assert [int(x) for x in got] == list(w)
    n = 200000
    m = orc.dropout_multipliers(n, 0.1, 99, 3, 2)
    assert set(m.unique().tolist()) == {0.0, float(np.float32(1.0) / (np.float32(1.0) - np.float32(0.1)))}
    assert abs((m == 0).float().mean().item() - 0.1) < 5 * (0.09 / n) ** 0.5
    assert not torch.equal(m, orc.dropout_multipliers(n, 0.1, 99, 4, 2))          # next forward call: new masks
    torch.manual_seed(0)
    D, H, L, B = 32, 2, 5, 2
    p = {"text_model.embeddings.word_embeddings.weight": torch.randn(50, D), "text_model.embeddings.position_embeddings.weight": torch.randn(8, D),
         "text_model.embeddings.LayerNorm.weight": torch.ones(D), "text_model.embeddings.LayerNorm.bias": torch.zeros(D)}
    b = "text_model.transformer.layer.0."
    for l, (o, i) in {"attention.q_lin": (D, D), "attention.k_lin": (D, D), "attention.v_lin": (D, D), "attention.out_lin": (D, D),
                      "ffn.lin1": (4 * D, D), "ffn.lin2": (D, 4 * D)}.items():
        p[b + l + ".weight"], p[b + l + ".bias"] = torch.randn(o, i) * 0.1, torch.randn(o) * 0.1
    for l in ("sa_layer_norm", "output_layer_norm"):
        p[b + l + ".weight"], p[b + l + ".bias"] = torch.ones(D), torch.zeros(D)
    ids, mask = torch.randint(0, 50, (B, L)), torch.ones(B, L, dtype=torch.int64)
    ones = {"emb": torch.ones(B, L, D), ("attn", 0): torch.ones(B, H, L, L), ("ffn", 0): torch.ones(B, L, D)}
    assert torch.equal(orc.distilbert(ids, mask, p, n_heads=H, dropout=ones), orc.distilbert(ids, mask, p, n_heads=H))
    masks = orc.distilbert_dropout_masks(B, L, D, H, 1, 0.1, 0.1, 7, 1)
    assert not torch.equal(orc.distilbert(ids, mask, p, n_heads=H, dropout=masks), orc.distilbert(ids, mask, p, n_heads=H))
