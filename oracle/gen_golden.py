"""Golden-vector generator.  TEST INFRASTRUCTURE ONLY — runs in the build container, never on the GPU box.

Imports the *reference's own* ``training/model.py`` and ``training/train_encoder.py`` from
/root/reference (read-only; nothing is copied) and records inputs and outputs as small ``.npz`` fixtures
under ``tests/golden/``.  Weights come from ``omnibiote_ref.hash_weights`` (closed form), so fixtures
hold only tokens, masks and outputs.

The reference imports ``mup`` (README.md:16 pins mup==1.0.0; not vendored, not installable here).  A
minimal in-process stand-in is registered for the import to succeed: ``MuReadout`` is restated from its
published algorithm (``Linear(output_mult * x / width_mult)``, width_mult = fan_in / base fan_in with the
base width 24 of train_encoder.py:158).  Everything up to ``emb`` is genuine reference code; ``logits``,
``loss`` and gradients additionally depend on that restated readout ("parity unpinned" for mup).

Usage:  python oracle/gen_golden.py   (writes tests/golden/*.npz)
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/training"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)
import omnibiote_ref as R  # noqa: E402


class MuReadout(nn.Linear):
    """In-process stand-in for mup.layer.MuReadout (restated; see the module docstring).  It presents itself as
    ``mup.layer.MuReadout`` so that a pickled model names the class the way the real package would."""

    def __init__(self, *a, readout_zero_init=False, output_mult=1.0, **kw):
        self.output_mult = output_mult
        super().__init__(*a, **kw)

    def width_mult(self):
        return self.in_features / R.MUP_BASE_WIDTH

    def forward(self, x):
        return super().forward(self.output_mult * x / self.width_mult())


MuReadout.__module__ = "mup.layer"


def _install_mup_standin():
    mup = types.ModuleType("mup")
    layer = types.ModuleType("mup.layer")
    layer.MuReadout = MuReadout
    mup.layer = layer
    mup.MuReadout = MuReadout
    mup.set_base_shapes = lambda *a, **k: None
    mup.MuAdamW = None
    sys.modules["mup"] = mup
    sys.modules["mup.layer"] = layer


def _import_reference():
    _install_mup_standin()
    for name in ("wandb",):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)
    import model as ref_model  # noqa
    import train_encoder as ref_train  # noqa
    return ref_model, ref_train


def build_ref(ref_model, cfg: R.RefConfig, dtype):
    c = ref_model.OmniBioTAConfig()
    c.block_size, c.vocab_size, c.n_layer, c.n_head, c.n_embd = cfg.block_size, cfg.vocab_size, cfg.n_layer, cfg.n_head, cfg.n_embd
    c.dropout = 0.0
    c.flash = cfg.flash
    m = ref_model.OmniBioTA(c)
    w = R.hash_weights(cfg)
    sd = m.state_dict()
    for k, v in w.items():
        assert sd[k].shape == v.shape, k
        sd[k].copy_(v)
    if dtype != torch.float32:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m.to(dtype)
    m.train()
    return m


def synth_tokens(rng, B, T, V, eos_rows):
    """Packed rows: tag, body..., EOS ; eos_rows[b] lists interior EOS positions."""
    tok = rng.integers(20, V, size=(B, T)).astype(np.int64)
    tok[:, 0] = 4
    for b, pos in enumerate(eos_rows):
        for p in pos:
            tok[b, p] = R.EOS_TOKEN
            if p + 1 < T:
                tok[b, p + 1] = 18 if (p % 2) else 4
    return tok


def sample(t: torch.Tensor, stride=5):
    return t.detach().float().flatten()[::stride].numpy().copy()


def run_case(ref_model, ref_train, name, cfg, dtype, B, T, eos_rows, with_mask, seed, n_accum=2, want_grads=True):
    rng = np.random.default_rng(seed)
    tokens = synth_tokens(rng, B, T, cfg.vocab_size, eos_rows)
    bern = rng.random((B, T)) < 0.15
    tok_t = torch.from_numpy(tokens)
    masked_ids, mlm = R.mlm_corrupt(tok_t, torch.from_numpy(bern))
    m = build_ref(ref_model, cfg, dtype)
    out = {"tokens": tokens, "mlm_mask": mlm.numpy(), "masked_ids": masked_ids.numpy(),
           "cfg": np.array([cfg.block_size, cfg.vocab_size, cfg.n_layer, cfg.n_head, cfg.n_embd, int(cfg.flash)], dtype=np.int64),
           "n_accum": np.int64(n_accum), "grad_stride": np.int64(5)}
    attn = None
    if with_mask:
        # train_encoder.py:290-292
        am = torch.ones((B, T, T), dtype=dtype) * -1e9
        am = ref_train.create_attention_mask(am, tok_t, padding=False)
        out["allowed"] = (am == 0).numpy()
        attn = am.unsqueeze(1).expand(-1, cfg.n_head, -1, -1)
    emb = m(masked_ids, attn_mask=attn, return_embeddings=True)
    out["emb"] = emb.detach().float().numpy()
    m.zero_grad(set_to_none=True)
    if with_mask:  # the manual path adds the mask in place (model.py:142): rebuild
        am = torch.ones((B, T, T), dtype=dtype) * -1e9
        am = ref_train.create_attention_mask(am, tok_t, padding=False)
        attn = am.unsqueeze(1).expand(-1, cfg.n_head, -1, -1)
        if not cfg.flash:
            attn = attn.contiguous()
    logits = m.forward(masked_ids, attn_mask=attn)
    out["logits"] = logits.detach().float().numpy()
    # train_encoder.py:301-305
    loss = F.cross_entropy(logits.view(-1, logits.size(-1)), tok_t.view(-1), reduction="none") / n_accum
    loss *= mlm.view(-1).float()
    loss = loss.sum() / mlm.view(-1).sum()
    out["loss"] = np.float32(loss.item())
    if want_grads:
        loss.backward()
        for k, p in m.named_parameters():
            g = p.grad
            out["grad_sample/" + k] = sample(g)
            out["grad_sum/" + k] = np.float64(g.double().sum().item())
            out["grad_abs/" + k] = np.float64(g.double().abs().sum().item())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss {out['loss']:.6f} emb|max| {np.abs(out['emb']).max():.4f}")


def mask_cases(ref_train):
    """create_attention_mask known answers, incl. SURVEY fact 4's example (row 0 EOS [3,8]; row 1 EOS [2,6,9])."""
    cases = {}
    T = 12
    tok = np.full((2, T), 30, dtype=np.int64)
    tok[0, [3, 8]] = 3
    tok[1, [2, 6, 9]] = 3
    cases["fact4"] = (tok, False)
    rng = np.random.default_rng(7)
    tok = rng.integers(20, 100, size=(4, 24)).astype(np.int64)
    tok[0, [5]] = 3
    tok[2, [0, 1, 13, 23]] = 3
    tok[3, [10, 11, 12]] = 3
    cases["ragged"] = (tok, False)   # row 1: no interior EOS
    tok = rng.integers(20, 100, size=(3, 16)).astype(np.int64)
    tok[1, [7]] = 3
    tok[1, 8:] = 1
    tok[2, [3, 9]] = 3
    tok[2, 10:] = 1
    cases["padding"] = (tok, True)   # row 0 has no EOS at all -> attends everywhere
    tok = rng.integers(20, 100, size=(2, 8)).astype(np.int64)
    cases["no_eos_padding"] = (tok, True)
    tok = rng.integers(20, 100, size=(1, 8)).astype(np.int64)
    tok[0, 7] = 3
    cases["eos_last"] = (tok, False)
    out = {}
    for name, (tok, padding) in cases.items():
        B, T = tok.shape
        am = torch.ones((B, T, T), dtype=torch.float32) * -1e9
        am = ref_train.create_attention_mask(am, torch.from_numpy(tok), padding=padding)
        vals = set(np.unique(am.numpy()).tolist())
        assert vals <= {0.0, -1e9}, vals
        out[name + "/tokens"] = tok
        out[name + "/padding"] = np.bool_(padding)
        out[name + "/allowed"] = (am == 0).numpy()
    np.savez_compressed(os.path.join(OUT, "attention_masks.npz"), **out)
    print("attention_masks:", list(cases))


def encode_case(ref_model):
    cfg = R.RefConfig(block_size=32, vocab_size=256, n_layer=1, n_head=2, n_embd=128)
    m = build_ref(ref_model, cfg, torch.float32)
    m.eval()
    rng = np.random.default_rng(11)
    tok = rng.integers(20, 256, size=(3, 20)).astype(np.int64)
    out = {"tokens": tok, "cfg": np.array([32, 256, 1, 2, 128, 1], dtype=np.int64)}
    with torch.no_grad():
        for method in ("mean", "first", "last", "max", "all"):
            out[method] = m.encode(torch.from_numpy(tok), method=method).numpy()
    out["num_params"] = np.int64(m.get_num_params())
    out["num_params_all"] = np.int64(m.get_num_params(non_embedding=False))
    np.savez_compressed(os.path.join(OUT, "encode.npz"), **out)
    print("encode: ok")


def rope_case(ref_model):
    """The helper functions in isolation, both buffer dtypes."""
    torch.manual_seed(3)
    q = torch.randn(2, 16, 2, 64)
    k = torch.randn(2, 16, 2, 64)
    tab = ref_model.precompute_freqs_cis(64, 32)
    oq, ok = ref_model.apply_rotary_emb(q, k, tab)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tab_b = tab.to(torch.bfloat16)  # what Module.to does to the buffer
    qb, kb = q.bfloat16(), k.bfloat16()
    oqb, okb = ref_model.apply_rotary_emb(qb, kb, tab_b)
    x = torch.linspace(-6, 6, 97)
    np.savez_compressed(os.path.join(OUT, "rope_gelu.npz"), q=q.numpy(), k=k.numpy(),
                        table_real=tab.real.numpy(), table_imag=tab.imag.numpy(),
                        oq=oq.numpy(), ok=ok.numpy(), oq_bf16=oqb.float().numpy(), ok_bf16=okb.float().numpy(),
                        gelu_x=x.numpy(), gelu_y=ref_model.fused_gelu(x).numpy(),
                        gelu_y_bf16=ref_model.fused_gelu(x.bfloat16()).float().numpy())
    print("rope_gelu: ok")


def checkpoint_case(ref_model):
    """A checkpoint exactly as the reference trainer writes it (train_encoder.py:170,413): the whole model object, after
    ``.to(bfloat16)``, pickled with torch.save.  Data produced BY the reference (class names inside the pickle point at its
    ``model`` module and at ``mup``), used to test omnibiote_amd.checkpoint.load_checkpoint."""
    cfg = R.RefConfig(block_size=64, vocab_size=64, n_layer=1, n_head=2, n_embd=128)
    m = build_ref(ref_model, cfg, torch.bfloat16)
    m.eval()
    torch.save(m, os.path.join(OUT, "ref_checkpoint_bf16.pt"))
    rng = np.random.default_rng(21)
    tok = rng.integers(4, 64, size=(2, 40)).astype(np.int64)
    with torch.no_grad():
        emb = m(torch.from_numpy(tok), return_embeddings=True)
    np.savez_compressed(os.path.join(OUT, "ref_checkpoint_io.npz"), tokens=tok, emb=emb.float().numpy(),
                        cfg=np.array([64, 64, 1, 2, 128, 1], dtype=np.int64))
    print("checkpoint: ok", os.path.getsize(os.path.join(OUT, "ref_checkpoint_bf16.pt")), "bytes")


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(4)
    ref_model, ref_train = _import_reference()
    tiny = dict(block_size=64, vocab_size=512, n_layer=2, n_head=2, n_embd=128)   # hs = 64
    wide = dict(block_size=128, vocab_size=256, n_layer=1, n_head=2, n_embd=256)  # hs = 128
    eos2 = [[20, 41], [9, 30, 50]]
    run_case(ref_model, ref_train, "tiny_fp32_nomask", R.RefConfig(**tiny), torch.float32, 2, 64, [[], []], False, 1)
    run_case(ref_model, ref_train, "tiny_fp32_mask", R.RefConfig(**tiny), torch.float32, 2, 64, eos2, True, 2)
    run_case(ref_model, ref_train, "tiny_fp32_mask_manual", R.RefConfig(**tiny, flash=False), torch.float32, 2, 64, eos2, True, 2, want_grads=False)
    run_case(ref_model, ref_train, "tiny_bf16_mask", R.RefConfig(**tiny), torch.bfloat16, 2, 64, eos2, True, 2)
    run_case(ref_model, ref_train, "tiny_bf16_nomask", R.RefConfig(**tiny), torch.bfloat16, 2, 64, [[], []], False, 1, want_grads=False)
    eos3 = [[40, 100], [15, 64, 65], [127]]
    run_case(ref_model, ref_train, "wide_fp32_mask", R.RefConfig(**wide), torch.float32, 3, 128, eos3, True, 3)
    run_case(ref_model, ref_train, "wide_bf16_mask", R.RefConfig(**wide), torch.bfloat16, 3, 128, eos3, True, 3)
    # ragged: T shorter than block_size and not a multiple of any tile
    run_case(ref_model, ref_train, "wide_fp32_ragged", R.RefConfig(**wide), torch.float32, 2, 77, [[30], []], True, 4, want_grads=False)
    mask_cases(ref_train)
    encode_case(ref_model)
    rope_case(ref_model)
    checkpoint_case(ref_model)


if __name__ == "__main__":
    main()
