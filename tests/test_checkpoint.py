"""Checkpoint compatibility: a checkpoint written by the reference itself (whole-object pickle of its
``model.OmniBioTA`` after ``.to(bfloat16)``, tests/golden/ref_checkpoint_bf16.pt, made by oracle/gen_golden.py) loads into
this package's classes; this package's own checkpoints round-trip."""
import os
import pickle

import numpy as np
import pytest
import torch

import omnibiote_ref as R
from omnibiote_amd import checkpoint as CK
from omnibiote_amd import model as M
from omnibiote_amd.mup_compat import set_base_shapes


def _cfg(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_checkpoint_io.npz"))
    bs, V, Lyr, H, C, _ = [int(v) for v in g["cfg"]]
    return g, R.RefConfig(block_size=bs, vocab_size=V, n_layer=Lyr, n_head=H, n_embd=C)


def test_reference_checkpoint_names_reference_classes(golden_dir):
    """The fixture really is a reference-format pickle: it names model.OmniBioTA / mup.layer.MuReadout, not our classes."""
    import zipfile
    with zipfile.ZipFile(os.path.join(golden_dir, "ref_checkpoint_bf16.pt")) as z:
        blob = z.read([n for n in z.namelist() if n.endswith("data.pkl")][0])
    assert b"OmniBioTA" in blob and b"mup.layer" in blob and b"omnibiote_amd" not in blob


def test_load_reference_checkpoint_into_product_classes(golden_dir):
    g, cfg = _cfg(golden_dir)
    m = CK.load_checkpoint(os.path.join(golden_dir, "ref_checkpoint_bf16.pt"))
    assert type(m) is M.OmniBioTA and type(m.transformer.h[0]) is M.Block and type(m.lm_head) is M.MuReadout
    assert type(m.transformer.h[0].attn) is M.SelfAttention and type(m.transformer.ln_f) is M.LayerNorm
    sd = m.state_dict()
    want = R.hash_weights(cfg)
    assert [k for k in sd if "freqs_cis" not in k] == list(want)
    for k, v in want.items():
        assert sd[k].dtype == torch.bfloat16 and torch.equal(sd[k], v.to(torch.bfloat16)), k
    f = sd["transformer.h.0.attn.freqs_cis"]
    assert f.dtype == torch.bfloat16 and not f.is_complex()        # the reference's .to(bfloat16) already degraded it
    assert m.config.n_embd == cfg.n_embd and m.config.flash is True
    assert m.get_num_params() == sum(v.numel() for v in want.values()) - want["transformer.wte.weight"].numel()
    with pytest.raises(RuntimeError, match="no CPU fallback"):     # loads on CPU, computes only on the GPU
        m(torch.from_numpy(g["tokens"]))


def test_own_checkpoint_round_trip(tmp_path):
    c = M.OmniBioTAConfig(); c.flash = True
    c.block_size, c.vocab_size, c.n_layer, c.n_head, c.n_embd, c.dropout = 32, 64, 1, 2, 128, 0.1
    m = M.OmniBioTA(c)
    path = os.path.join(tmp_path, "own.pt")
    CK.save_checkpoint(m, path)
    m2 = CK.load_checkpoint(path)
    assert type(m2) is M.OmniBioTA
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    m3 = torch.load(path, weights_only=False)                       # plain torch.load works too (the evals' way)
    assert type(m3) is M.OmniBioTA


@pytest.mark.gpu
def test_loaded_reference_checkpoint_runs_on_hip_and_matches_reference_output(golden_dir):
    g, cfg = _cfg(golden_dir)
    m = CK.load_checkpoint(os.path.join(golden_dir, "ref_checkpoint_bf16.pt")).to("cuda").eval()
    with torch.no_grad():
        emb = m(torch.from_numpy(g["tokens"]).to("cuda"), return_embeddings=True)
    d = (emb.float().cpu() - torch.from_numpy(g["emb"])).abs()
    assert d.max().item() <= 0.10 and d.mean().item() <= 5e-3, (d.max().item(), d.mean().item())
    # fine-tuning as the evals do it: deepcopy, parameter groups by name, a backward pass
    import copy
    ft = copy.deepcopy(m).train()
    for mod in ft.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    for blk in ft.transformer.h:
        blk.attn.dropout = 0.0
    out = ft(torch.from_numpy(g["tokens"]).to("cuda"), return_embeddings=True)[:, 0]
    out.float().pow(2).sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in ft.named_parameters() if "lm_head" not in n)
