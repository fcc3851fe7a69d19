"""GPU parity at the sizes of BASELINE configs 2, 4 and 5 (SURVEY §5 long-context row, §8 rows a5-a12):

  config 4  small ctx = 4096      attention fwd/bwd at T = 4096 vs the oracle (no mask / multi-document key ranges / the
                                  reference's dense additive mask), and — at the full B = 8, H = 8 — sampled heads vs the
                                  oracle plus size-independent properties (softmax rows sum to 1; V = 1 makes dQ = dK = 0
                                  and column sums of dV equal the query count; backward linear in dO);
  config 5  large 24L/2048d/16h   LayerNorm at 2048 columns, every GEMM shape of the large block (N, K in 2048 / 6144 /
                                  8192) vs a sampled fp64 reference and a full fp32 one, one block fwd/bwd at
                                  C = 2048 / H = 16 and a 4-layer model at that width vs the oracle;
  config 2  small 8L/1024d/8h     ONE full-depth micro-batch (8 layers, T = 1024, full 65 536-way logits, loss, every
                                  gradient) vs the oracle — the end-to-end check at the benchmark's own depth and width.

Tolerances as in test_hip_ops.py: bf16 inputs, fp32 accumulation, one bf16 rounding per stored tensor."""
import warnings

import numpy as np
import pytest
import torch

import omnibiote_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
RTOL = 2.0 ** -7


def ops():
    from omnibiote_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def close(got, ref, atol, rtol=RTOL, what=""):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - ref).abs()
    bad = err > rtol * ref.abs() + atol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off; max err {err.max():.4g} (ref max {ref.abs().max():.4g})"


def seeded_weights(cfg, seed=0):
    """Same amplitudes as R.hash_weights, from torch's CPU generator (the closed-form hash takes minutes at these sizes;
    nothing here is compared with a stored fixture, only HIP vs oracle on the same tensors)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in R.param_shapes(cfg).items():
        u = torch.rand(shape, generator=g) - 0.5
        out[name] = u * 2.0 if name.endswith("wte.weight") else (1.0 + u * 0.5 if "ln_" in name else u * (2.0 / np.sqrt(shape[-1])))
    return out


def multi_document_tokens(B, T, seed, n_docs=5):
    rng = np.random.default_rng(seed)
    tok = rng.integers(20, 1000, size=(B, T))
    for b in range(B):
        tok[b, np.sort(rng.choice(np.arange(8, T - 8), size=n_docs - 1, replace=False))] = R.EOS_TOKEN
    return tok


def masks_from_tokens(tok, T):
    blocks = R.document_blocks(tok)
    return R.dense_mask_from_blocks(blocks, T), torch.from_numpy(R.key_ranges_from_blocks(blocks, T))


# ----------------------------------------------------------------------------------------- config 4: ctx = 4096
@pytest.mark.parametrize("mode", ["none", "ranges", "dense"])
def test_attention_ctx4096_vs_oracle(mode):
    B, H, T, hs = 1, 2, 4096, 128
    C = H * hs
    scale = 8.0 / 1024          # the small config's 8 / n_embd
    qkv = rnd(B, T, 3 * C, seed=7)
    q, k, v = [t.reshape(B, T, H, hs).transpose(1, 2).float().requires_grad_(True) for t in qkv.split(C, dim=2)]
    dense, ranges = masks_from_tokens(multi_document_tokens(B, T, seed=1), T)
    o = ops()
    mask_add, spec = None, None
    if mode == "ranges":
        mask_add, spec = dense.unsqueeze(1), o.MaskSpec(ranges=ranges.to(DEV))
    elif mode == "dense":
        mask_add = dense.unsqueeze(1)
        spec = o.MaskSpec.from_user(dense.to(BF).to(DEV).unsqueeze(1).expand(B, H, T, T), B, T, H, DEV)   # train_encoder.py:292
    ref = R.attention(q, k, v, scale, mask_add)
    d_o = rnd(B, T, C, seed=99)
    ref.backward(d_o.reshape(B, T, H, hs).transpose(1, 2).float())
    got, lse = o.attn_fwd(qkv.to(DEV), B, T, H, hs, scale, spec)
    close(got, ref.transpose(1, 2).reshape(B, T, C), atol=6e-3, what=f"attn fwd T=4096 {mode}")
    dqkv = o.attn_bwd(qkv.to(DEV), got, d_o.to(DEV), lse, B, T, H, hs, scale, spec)
    dref = torch.cat([g.transpose(1, 2).reshape(B, T, C) for g in (q.grad, k.grad, v.grad)], dim=2)
    close(dqkv, dref, atol=1.5e-2, rtol=2.0 ** -6, what=f"attn bwd T=4096 {mode}")


def test_attention_full_size_ctx4096_sampled_heads_and_properties():
    """B = 8, T = 4096, H = 8, hs = 128 (config 4's micro-batch).  (a) two sampled (batch, head) pairs against the
    oracle; (b) V = 1: outputs are 1, dQ = dK = 0, sum_k dV = #queries x dO-column; (c) backward is linear in dO."""
    B, T, H, hs = 8, 4096, 8, 128
    C = H * hs
    scale = 8.0 / C
    g = torch.Generator(device=DEV).manual_seed(4)
    qkv = torch.randn(B, T, 3 * C, device=DEV, generator=g).to(BF)
    d_o = (torch.randn(B, T, C, device=DEV, generator=g) * 0.5).to(BF)
    tok = multi_document_tokens(B, T, seed=2, n_docs=4)
    dense, ranges = masks_from_tokens(tok, T)
    o = ops()
    spec = o.MaskSpec(ranges=ranges.to(DEV))
    out, lse = o.attn_fwd(qkv, B, T, H, hs, scale, spec)
    dqkv = o.attn_bwd(qkv, out, d_o, lse, B, T, H, hs, scale, spec)
    assert torch.isfinite(lse).all() and torch.isfinite(dqkv.float()).all()
    for (b, h) in ((3, 5), (7, 0)):
        sl = slice(h * hs, (h + 1) * hs)
        q, k, v = [qkv[b, :, i * C:(i + 1) * C][:, sl].float().cpu().reshape(1, 1, T, hs).requires_grad_(True) for i in range(3)]
        ref = R.attention(q, k, v, scale, dense[b].reshape(1, 1, T, T))
        ref.backward(d_o[b, :, sl].float().cpu().reshape(1, 1, T, hs))
        close(out[b, :, sl], ref.reshape(T, hs), atol=6e-3, what=f"full-size fwd (b={b},h={h})")
        for i, gr in enumerate((q.grad, k.grad, v.grad)):
            close(dqkv[b, :, i * C:(i + 1) * C][:, sl], gr.reshape(T, hs), atol=1.5e-2, rtol=2.0 ** -6, what=f"full-size bwd part {i} (b={b},h={h})")
    # (c) linearity in dO: bwd(dO) + bwd(dO2) == bwd(dO + dO2) up to the bf16 rounding of the three results
    d_o2 = (torch.randn(B, T, C, device=DEV, generator=g) * 0.5).to(BF)
    both = (d_o.float() + d_o2.float()).to(BF)
    lhs = o.attn_bwd(qkv, out, both, lse, B, T, H, hs, scale, spec).float()
    rhs = dqkv.float() + o.attn_bwd(qkv, out, d_o2, lse, B, T, H, hs, scale, spec).float()
    rel = (lhs - rhs).norm() / rhs.norm()
    assert rel < 1.5e-2, rel.item()            # dO + dO2 itself is rounded to bf16 (2^-9 relative per element)
    # (b) V = 1
    qkv1 = qkv.clone()
    qkv1[..., 2 * C:] = 1.0
    out1, lse1 = o.attn_fwd(qkv1, B, T, H, hs, scale, spec)
    assert (out1.float() - 1.0).abs().max().item() <= 2.0 ** -7
    ones = torch.ones_like(out1)               # use the exact O = 1 so that delta = rowsum(dO) exactly
    dq1 = o.attn_bwd(qkv1, ones, d_o, lse1, B, T, H, hs, scale, spec).float()
    assert dq1[..., :2 * C].abs().max().item() <= 2e-2 * max(1.0, d_o.float().abs().max().item())      # dS = P (dP - delta) = 0
    dv = dq1[..., 2 * C:].reshape(B, T, H, hs)
    want = d_o.float().reshape(B, T, H, hs).sum(dim=1)            # sum_k dV[k] = sum_q dO[q] (every P row sums to 1)
    got = dv.sum(dim=1)
    assert ((got - want).abs() <= 0.02 * want.abs() + 2.0).all(), (got - want).abs().max().item()


# ------------------------------------------------------------------------------ config 5: large (2048d / 16 heads)
@pytest.mark.parametrize("rows,cols", [(8192, 2048), (100, 2048), (8192, 1024)])
def test_layernorm_full_rows_and_large_width(rows, cols):
    x, dy = rnd(rows, cols, seed=1, scale=2.0), rnd(rows, cols, seed=2)
    w = (1.0 + 0.3 * rnd(cols, seed=3).float()).to(BF)
    xf, wf = x.float().requires_grad_(True), w.float().requires_grad_(True)
    ref = R.layer_norm(xf, wf)
    ref.backward(dy.float())
    o = ops()
    y, mean, rstd = o.layernorm_fwd(x.to(DEV), w.to(DEV))
    close(y, ref, atol=2e-2, what="ln fwd")
    dx, dw = o.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd)
    close(dx, xf.grad, atol=2e-2, rtol=2.0 ** -6, what="ln dx")
    close(dw, wf.grad, atol=0.02 * wf.grad.abs().max().item() + 1e-2, rtol=2.0 ** -6, what="ln dw")


LARGE_GEMMS = [  # (M, N, K, a_kmajor, b_kmajor, epilogue-name)  — the large block at 8192 rows
    (8192, 6144, 2048, True, True, "none"), (8192, 2048, 2048, True, True, "add"), (8192, 8192, 2048, True, True, "gelu"),
    (8192, 2048, 8192, True, True, "add"), (8192, 8192, 2048, True, False, "gelu_bwd"), (8192, 2048, 8192, True, False, "none"),
    (8192, 2048, 6144, True, False, "none"), (2048, 8192, 8192, False, False, "none"), (6144, 2048, 8192, False, False, "none"),
]


@pytest.mark.parametrize("M,N,K,ak,bk,epi", LARGE_GEMMS)
def test_large_config_gemm_shapes(M, N, K, ak, bk, epi):
    """Every projection shape of the large block: the whole output against torch's fp32 matmul of the same bf16
    operands (on the GPU), and 48 sampled rows against an fp64 product on the CPU."""
    from omnibiote_amd import _lib as L
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = (torch.randn((M, K) if ak else (K, M), device=DEV, generator=g) * 0.5).to(BF)
    b = (torch.randn((N, K) if bk else (K, N), device=DEV, generator=g) * 0.5).to(BF)
    A = a.float() if ak else a.float().t()
    Bm = b.float() if bk else b.float().t()
    acc = A @ Bm.t()
    aux = (torch.randn(M, N, device=DEV, generator=g)).to(BF) if epi in ("add", "gelu_bwd") else None
    code = {"none": L.EPI_NONE, "add": L.EPI_ADD, "gelu": L.EPI_GELU, "gelu_bwd": L.EPI_GELU_BWD}[epi]
    got = ops().gemm(a, b, M, N, K, ak, bk, code, aux)
    atol = 0.02 * np.sqrt(K) * 0.25
    rows = torch.from_numpy(np.random.default_rng(0).choice(M, size=48, replace=False)).to(DEV)
    acc64 = (A[rows].double().cpu() @ Bm.double().cpu().t())
    assert (acc[rows].cpu().double() - acc64).abs().max().item() <= 1e-3 * np.sqrt(K)     # the fp32 reference itself
    if epi == "none":
        close(got, acc, atol=atol, what="gemm")
    elif epi == "add":
        close(got, aux.float() + acc.to(BF).float(), atol=atol, what="gemm+add")
    elif epi == "gelu":
        der, act = got
        h = acc.to(BF).float().cpu()
        close(act, R.gelu_erf(h), atol=atol, what="gelu act")
        hh = h.clone().requires_grad_(True)
        R.gelu_erf(hh).sum().backward()
        close(der, hh.grad, atol=2e-2, what="gelu derivative")
    else:
        close(got, acc.to(BF).float() * aux.float(), atol=atol * 1.5, what="gemm*aux")


@pytest.mark.parametrize("grouped", ["0", "1"])
def test_block_fwd_bwd_large_width_vs_oracle(monkeypatch, grouped):
    """One block of the large config (C = 2048, H = 16, hs = 128) at B = 2, T = 256 against R.block_forward."""
    monkeypatch.setenv("OBTE_GROUPED_WGRAD", grouped)
    B, T, C, H = 2, 256, 2048, 16
    hs = C // H
    cfg = R.RefConfig(block_size=T, vocab_size=256, n_layer=1, n_head=H, n_embd=C)
    w = {k: v.to(BF) for k, v in seeded_weights(cfg).items()}
    pre = "transformer.h.0."
    names = ["ln_1.weight", "attn.c_attn.weight", "attn.c_proj.weight", "ln_2.weight", "mlp.c_fc.weight", "mlp.c_proj.weight"]
    x, dy = rnd(B, T, C, seed=1), rnd(B, T, C, seed=2, scale=0.1)
    dense, ranges = masks_from_tokens(multi_document_tokens(B, T, seed=3, n_docs=3), T)
    tab = R.cast_rope_table(R.rope_table(hs, T), BF)          # the training regime: cos-only
    wf = {k: v.float().requires_grad_(True) for k, v in w.items()}
    xf = x.float().requires_grad_(True)
    ref = R.block_forward(xf, wf, pre, cfg, tab, dense.unsqueeze(1))
    ref.backward(dy.float())
    from omnibiote_amd.model import rope_tables
    o = ops()
    params = tuple(w[pre + n].to(DEV) for n in names)
    rope = rope_tables(tab.to(DEV))
    spec = o.MaskSpec(ranges=ranges.to(DEV))
    y, act = o.block_fwd(x.to(DEV), params, rope, H, spec)
    close(y, ref, atol=3e-2, rtol=2.0 ** -6, what="block fwd C=2048")
    dx, grads = o.block_bwd(x.to(DEV), dy.to(DEV), act, params, rope, H, spec)
    close(dx, xf.grad, atol=2e-2, rtol=2.0 ** -5, what="block dx C=2048")
    for n, gq in zip(names, grads):
        gr = wf[pre + n].grad
        close(gq, gr, atol=0.03 * gr.abs().max().item() + 1e-3, rtol=2.0 ** -5, what="block d" + n)


def _hip_model(cfg, w, T):
    from omnibiote_amd.model import OmniBioTA, OmniBioTAConfig
    from omnibiote_amd.mup_compat import set_base_shapes
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        c = OmniBioTAConfig()
        c.block_size, c.vocab_size, c.n_layer, c.n_head, c.n_embd, c.dropout, c.flash = T, cfg.vocab_size, cfg.n_layer, cfg.n_head, cfg.n_embd, 0.0, True
        m = OmniBioTA(c)
        cb = OmniBioTAConfig(); cb.block_size, cb.vocab_size, cb.n_layer, cb.dropout, cb.flash = T, cfg.vocab_size, cfg.n_layer, 0.0, True
        cb.n_embd, cb.n_head = 24, 3
        base = OmniBioTA(cb)
        cb.n_embd, cb.n_head = 48, 12
        delta = OmniBioTA(cb)
    set_base_shapes(m, base, delta=delta, rescale_params=False)
    m.load_state_dict(w, strict=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.to(BF)
    return m.to(DEV)


def _model_vs_oracle(cfg, B, T, seed, n_docs, emb_bar, logit_bar, grad_cos, grad_rel):
    """One micro-batch: forward (emb, logits, loss) and every parameter gradient, HIP vs the oracle in fp32 arithmetic
    on the same bf16-valued weights (cos-only RoPE, as the bf16 module holds it)."""
    w = seeded_weights(cfg, seed)
    m = _hip_model(cfg, w, T)
    tok_np = multi_document_tokens(B, T, seed=seed + 1, n_docs=n_docs)
    tok_np = np.where(tok_np == R.EOS_TOKEN, tok_np, tok_np % cfg.vocab_size)
    tok = torch.from_numpy(tok_np)
    rng = np.random.default_rng(seed + 2)
    masked, mlm = R.mlm_corrupt(tok, torch.from_numpy(rng.random((B, T)) < 0.15))
    from omnibiote_amd.masks import RangeMask
    rm = RangeMask.from_tokens(tok.to(DEV))
    emb = m(masked.to(DEV), attn_mask=rm, return_embeddings=True)
    logits = m(masked.to(DEV), attn_mask=rm)
    loss, dlogits = ops().masked_ce(logits, tok.to(DEV), mlm.to(DEV), 1)
    logits.backward(dlogits)
    torch.cuda.synchronize()
    wb = {k: v.to(BF).float().requires_grad_(True) for k, v in w.items()}
    dense = rm.dense(torch.float32).cpu().unsqueeze(1)
    rope = R.cast_rope_table(R.rope_table(cfg.n_embd // cfg.n_head, T), BF)
    ref_emb = R.model_forward(wb, cfg, masked, dense, return_embeddings=True, rope=rope)
    ref_logits = R.readout(ref_emb, wb["lm_head.weight"], cfg.n_embd / cfg.mup_base_width)
    ref_loss = R.masked_lm_loss(ref_logits, tok, mlm, 1)
    ref_loss.backward()
    d = (emb.float().cpu() - ref_emb.detach()).abs()
    print(f"[{cfg.n_layer}L/{cfg.n_embd}d T={T}] emb max {d.max().item():.4f} mean {d.mean().item():.5f}; ", end="")
    assert d.max().item() <= emb_bar[0] and d.mean().item() <= emb_bar[1], ("emb", d.max().item(), d.mean().item())
    d = (logits.float().cpu() - ref_logits.detach()).abs()
    print(f"logits max {d.max().item():.4f} mean {d.mean().item():.5f}; loss {loss.item():.4f} vs {ref_loss.item():.4f}")
    assert d.max().item() <= logit_bar[0] and d.mean().item() <= logit_bar[1], ("logits", d.max().item(), d.mean().item())
    assert abs(loss.item() - ref_loss.item()) <= 0.02, (loss.item(), ref_loss.item())
    worst = (1.0, 0.0, "")
    for k, p in m.named_parameters():
        got, want = p.grad.float().cpu().flatten(), wb[k].grad.flatten()
        assert torch.isfinite(got).all(), k
        cos = (torch.dot(got, want) / (got.norm() * want.norm() + 1e-30)).item()
        rel = ((got - want).norm() / (want.norm() + 1e-30)).item()
        if cos < worst[0]:
            worst = (cos, rel, k)
        assert cos >= grad_cos and rel <= grad_rel, (k, cos, rel)
    print(f"worst gradient: {worst[2]} cos {worst[0]:.5f} rel {worst[1]:.4f}")
    return worst


def test_large_width_four_layer_model_vs_oracle():
    """The large config's width and head count (2048d / 16 heads) through the whole model path (embedding, 4 blocks,
    ln_f, muP readout with width_mult 85.33, loss, all gradients) at T = 256."""
    cfg = R.RefConfig(block_size=256, vocab_size=8192, n_layer=4, n_head=16, n_embd=2048)
    _model_vs_oracle(cfg, B=2, T=256, seed=11, n_docs=3, emb_bar=(0.10, 8e-3), logit_bar=(2e-3, 2e-4), grad_cos=0.9995, grad_rel=0.03)   # measured: 0.045/4.3e-3, 2e-4/3e-5, 0.99997/0.008


# ------------------------------------------------------------------------ config 2: the benchmark's own model, full depth
def test_small_config_full_depth_one_microbatch_vs_oracle():
    """small = 8L / 1024d / 8h, T = 1024, full 65 536-way logits: one micro-batch row end to end against the oracle."""
    cfg = R.RefConfig(block_size=1024, vocab_size=65536, n_layer=8, n_head=8, n_embd=1024)
    _model_vs_oracle(cfg, B=1, T=1024, seed=21, n_docs=3, emb_bar=(0.15, 1e-2), logit_bar=(3e-3, 4e-4), grad_cos=0.9995, grad_rel=0.04)   # measured: 0.076/5.7e-3, 6e-4/8e-5, 0.99992/0.013


def test_small_config_ctx4096_one_row_vs_oracle():
    """BASELINE config 4 end to end: small = 8L / 1024d / 8h at T = 4096 (block_size 4096, RoPE table of 4096 positions),
    one multi-document row through embedding, 8 blocks, ln_f, the 65 536-way readout, the loss and every gradient."""
    cfg = R.RefConfig(block_size=4096, vocab_size=65536, n_layer=8, n_head=8, n_embd=1024)
    _model_vs_oracle(cfg, B=1, T=4096, seed=31, n_docs=6, emb_bar=(0.15, 1e-2), logit_bar=(3e-3, 4e-4), grad_cos=0.9995, grad_rel=0.04)
