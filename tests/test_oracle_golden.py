"""The oracle (oracle/omnibiote_ref.py) pinned against vectors captured from the reference itself
(oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import omnibiote_ref as R


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def cfg_of(g):
    bs, V, L, H, C, flash = [int(v) for v in g["cfg"]]
    return R.RefConfig(block_size=bs, vocab_size=V, n_layer=L, n_head=H, n_embd=C, flash=bool(flash))


def mask_of(g, dtype):
    if "allowed" not in g.files:
        return None
    allowed = torch.from_numpy(g["allowed"])
    m = torch.where(allowed, torch.zeros((), dtype=torch.float32), torch.full((), R.MASKED_VALUE)).to(dtype)
    return m.unsqueeze(1)


FP32_CASES = ["tiny_fp32_nomask", "tiny_fp32_mask", "tiny_fp32_mask_manual", "wide_fp32_mask", "wide_fp32_ragged"]
BF16_CASES = ["tiny_bf16_mask", "tiny_bf16_nomask", "wide_bf16_mask"]


@pytest.mark.parametrize("name", FP32_CASES)
def test_forward_fp32(golden_dir, name):
    g = load(golden_dir, name)
    cfg = cfg_of(g)
    w = R.hash_weights(cfg)
    idx = torch.from_numpy(g["masked_ids"])
    mask = mask_of(g, torch.float32)
    emb = R.model_forward(w, cfg, idx, mask, return_embeddings=True)
    np.testing.assert_allclose(emb.numpy(), g["emb"], rtol=0, atol=2e-5)
    logits = R.model_forward(w, cfg, idx, mask)
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=0, atol=2e-5)
    loss = R.masked_lm_loss(logits, torch.from_numpy(g["tokens"]), torch.from_numpy(g["mlm_mask"]), int(g["n_accum"]))
    assert abs(loss.item() - float(g["loss"])) < 2e-6


@pytest.mark.parametrize("name", ["tiny_fp32_nomask", "tiny_fp32_mask", "wide_fp32_mask"])
def test_backward_fp32(golden_dir, name):
    g = load(golden_dir, name)
    cfg = cfg_of(g)
    w = {k: v.requires_grad_(True) for k, v in R.hash_weights(cfg).items()}
    logits = R.model_forward(w, cfg, torch.from_numpy(g["masked_ids"]), mask_of(g, torch.float32))
    loss = R.masked_lm_loss(logits, torch.from_numpy(g["tokens"]), torch.from_numpy(g["mlm_mask"]), int(g["n_accum"]))
    loss.backward()
    stride = int(g["grad_stride"])
    for k, p in w.items():
        got = p.grad.flatten()[::stride].numpy()
        want = g["grad_sample/" + k]
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-6 + 1e-4 * np.abs(want).max(), err_msg=k)
        assert abs(p.grad.double().sum().item() - float(g["grad_sum/" + k])) <= 1e-4 * float(g["grad_abs/" + k]) + 1e-7, k


@pytest.mark.parametrize("name", BF16_CASES)
def test_forward_bf16_cos_only_rope(golden_dir, name):
    """bf16 run: the module's complex RoPE buffer degenerates to bf16 cos.  The oracle,
    run in bf16 with the same torch ops, must agree to bf16 rounding noise (a few ulp of the output scale);
    true rotation instead would be off by O(1)."""
    g = load(golden_dir, name)
    cfg = cfg_of(g)
    w = {k: v.bfloat16() for k, v in R.hash_weights(cfg).items()}
    idx = torch.from_numpy(g["masked_ids"])
    mask = mask_of(g, torch.bfloat16)
    emb = R.model_forward(w, cfg, idx, mask, return_embeddings=True).float().numpy()
    assert np.abs(emb - g["emb"]).max() <= 0.07, np.abs(emb - g["emb"]).max()
    assert np.abs(emb - g["emb"]).mean() <= 4e-3
    # and it is NOT what complex RoPE would give
    emb_c = R.model_forward(w, cfg, idx, mask, return_embeddings=True,
                            rope=R.rope_table(cfg.n_embd // cfg.n_head, cfg.block_size)).float().numpy()
    assert np.abs(emb_c - g["emb"]).mean() > 2 * np.abs(emb - g["emb"]).mean()
    logits = R.model_forward(w, cfg, idx, mask)
    loss = R.masked_lm_loss(logits, torch.from_numpy(g["tokens"]), torch.from_numpy(g["mlm_mask"]), int(g["n_accum"]))
    assert abs(loss.item() - float(g["loss"])) < 0.05


def test_flash_and_manual_paths_agree(golden_dir):
    a, b = load(golden_dir, "tiny_fp32_mask"), load(golden_dir, "tiny_fp32_mask_manual")
    np.testing.assert_allclose(a["emb"], b["emb"], atol=5e-6)


def test_attention_mask_builder(golden_dir):
    g = load(golden_dir, "attention_masks")
    names = sorted({k.split("/")[0] for k in g.files})
    assert "fact4" in names
    for n in names:
        tok, padding, allowed = g[n + "/tokens"], bool(g[n + "/padding"]), g[n + "/allowed"]
        blocks = R.document_blocks(tok, padding=padding)
        T = tok.shape[1]
        dense = R.dense_mask_from_blocks(blocks, T)
        np.testing.assert_array_equal((dense == 0).numpy(), allowed, err_msg=n)
        rng = R.key_ranges_from_blocks(blocks, T)
        rebuilt = np.zeros_like(allowed)
        for b in range(tok.shape[0]):
            for t in range(T):
                rebuilt[b, t, rng[b, t, 0]:rng[b, t, 1]] = True
        np.testing.assert_array_equal(rebuilt, allowed, err_msg=n)
    # SURVEY fact 4 known answer
    blocks = R.document_blocks(g["fact4/tokens"])
    rng = R.key_ranges_from_blocks(blocks, 12)
    assert [tuple(r) for r in rng[0, [0, 4, 9]]] == [(0, 4), (4, 9), (9, 12)]
    assert [tuple(r) for r in rng[1, [0, 7, 10]]] == [(0, 7), (7, 10), (10, 12)]


def test_encode_pooling_and_param_count(golden_dir):
    g = load(golden_dir, "encode")
    cfg = cfg_of(g)
    w = R.hash_weights(cfg)
    emb = R.model_forward(w, cfg, torch.from_numpy(g["tokens"]), None, return_embeddings=True)
    for method in ("mean", "first", "last", "max", "all"):
        np.testing.assert_allclose(R.encode_pool(emb, method).numpy(), g[method], atol=2e-5)
    with pytest.raises(AssertionError):
        R.encode_pool(emb, "median")
    enc = R.OracleEncoder(cfg)
    assert enc.get_num_params() == int(g["num_params"])
    assert enc.get_num_params(non_embedding=False) == int(g["num_params_all"])


def test_rope_and_gelu_helpers(golden_dir):
    g = load(golden_dir, "rope_gelu")
    q, k = torch.from_numpy(g["q"]), torch.from_numpy(g["k"])
    tab = R.rope_table(64, 32)
    np.testing.assert_allclose(tab.real.numpy(), g["table_real"], atol=1e-7)
    np.testing.assert_allclose(tab.imag.numpy(), g["table_imag"], atol=1e-7)
    np.testing.assert_allclose(R.apply_rope(q, tab).numpy(), g["oq"], atol=1e-6)
    np.testing.assert_allclose(R.apply_rope(k, tab).numpy(), g["ok"], atol=1e-6)
    tb = R.cast_rope_table(tab, torch.bfloat16)
    assert tb.dtype == torch.bfloat16 and not tb.is_complex()
    np.testing.assert_array_equal(R.apply_rope(q.bfloat16(), tb).float().numpy(), g["oq_bf16"])
    np.testing.assert_array_equal(R.apply_rope(k.bfloat16(), tb).float().numpy(), g["ok_bf16"])
    x = torch.from_numpy(g["gelu_x"])
    np.testing.assert_allclose(R.gelu_erf(x).numpy(), g["gelu_y"], atol=1e-7)
    np.testing.assert_array_equal(R.gelu_erf(x.bfloat16()).float().numpy(), g["gelu_y_bf16"])


def _trajectory_setup(golden_dir):
    g = np.load(os.path.join(golden_dir, "trajectory_tiny_bf16.npz"))
    bs, V, Lyr, H, C, _ = [int(v) for v in g["cfg"]]
    hyper = dict(zip(["lr", "wd", "b1", "b2", "eps", "total_iters", "rows", "mini", "T", "steps", "n_batches"], g["hyper"]))
    for k in ("total_iters", "rows", "mini", "T", "steps", "n_batches"):
        hyper[k] = int(hyper[k])
    return g, R.RefConfig(block_size=bs, vocab_size=V, n_layer=Lyr, n_head=H, n_embd=C), hyper


def test_oracle_training_trajectory_tracks_the_reference_run(golden_dir):
    """tests/golden/trajectory_tiny_bf16.npz holds 30 optimizer steps of the IMPORTED reference model in its own regime
    (bf16 parameters and moments, torch.optim.AdamW with the muP groups passed explicitly, clip 1.0, LinearLR, the
    reference's loss lines and mask builder; oracle/gen_golden_trajectory.py), once through SDPA and once through its
    manual attention path.  The two reference runs differ by at most one bf16 ulp of a micro-batch loss (2^-6 = 0.0156);
    the oracle, run the same way, must stay within that spread of the reference at every step."""
    from omnibiote_amd import train_encoder as TE
    g, cfg, h = _trajectory_setup(golden_dir)
    spread = float(np.abs(g["losses_flash"] - g["losses_manual"]).max())
    assert spread <= 2.0 ** -6 + 1e-9
    w = {k: v.to(torch.bfloat16) for k, v in R.hash_weights(cfg).items()}
    enc = R.OracleEncoder(cfg, w)
    enc.rope = R.cast_rope_table(R.rope_table(cfg.n_embd // cfg.n_head, cfg.block_size), torch.bfloat16)
    named = enc.named_weights()
    mats = [p for n, p in named.items() if p.dim() == 2 and "wte" not in n and "lm_head" not in n]
    vecs = [p for n, p in named.items() if not (p.dim() == 2 and "wte" not in n and "lm_head" not in n)]
    wm = cfg.n_embd / R.MUP_BASE_WIDTH
    opt = torch.optim.AdamW([{"params": mats, "lr": h["lr"] / wm, "weight_decay": h["wd"] * wm}, {"params": vecs, "lr": h["lr"], "weight_decay": h["wd"]}],
                            lr=h["lr"], betas=(h["b1"], h["b2"]), eps=h["eps"], weight_decay=h["wd"])
    sched = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1.0, end_factor=0.0, total_iters=h["total_iters"])
    step = TE.TrainStep(enc, opt, sched, mini_batch_size=h["mini"], n_head=cfg.n_head, loss_impl="torch", mask_impl="dense")
    losses = []
    for i in range(h["steps"]):
        ids = torch.from_numpy(g["tokens"][i % h["n_batches"]])
        losses.append(step(ids, mlm_mask=torch.from_numpy(g["mlm_draw"][i]))["loss"].item())
    losses = np.array(losses)
    assert np.abs(losses - g["losses_flash"]).max() <= 2.0 ** -6 + 1e-6, (losses, g["losses_flash"])
    assert losses[-5:].mean() < losses[:5].mean() - 0.1                       # it trains
    stride = int(g["param_stride"])
    for k, p in named.items():
        d = np.abs(p.detach().float().flatten()[::stride].numpy() - g["final_flash/" + k])
        dd = np.abs(g["final_manual/" + k] - g["final_flash/" + k])
        # bar: the spread between the reference's own two runs (dd), plus two bf16 ulps at 1.0 spread over the tensor
        assert d.mean() <= 3.0 * dd.mean() + 2e-3 and d.max() <= 0.06, (k, d.mean(), d.max(), dd.mean(), dd.max())


def test_dropout_generator_statistics():
    """The counter-based dropout generator (restated in the oracle bit for bit from csrc/common.h: one hash per row, one per
    pair of columns, 16 bits per element) must behave like independent Bernoulli draws at the sizes the path uses it:
    overall rate, per-row and per-column rates, no correlation between the two elements of a pair, between neighbouring
    pairs, neighbouring rows, sites or seeds."""
    p, rows, cols = 0.1, 2048, 1024
    keep = R.dropout_keep(np.arange(rows, dtype=np.uint64)[:, None], np.arange(cols, dtype=np.uint64)[None, :], p, 0x1234_5678_9ABC, 1)
    k = keep.astype(np.float64)
    n = rows * cols
    sd = np.sqrt(p * (1 - p))
    assert abs(k.mean() - (1 - p)) < 4 * sd / np.sqrt(n)
    assert np.abs(k.mean(axis=1) - (1 - p)).max() < 5.5 * sd / np.sqrt(cols)       # every row
    assert np.abs(k.mean(axis=0) - (1 - p)).max() < 5.5 * sd / np.sqrt(rows)       # every column
    z = (k - k.mean()) / k.std()

    def corr(a, b):
        return float((a * b).mean())
    bound = 5.0 / np.sqrt(n / 2)
    assert abs(corr(z[:, 0::2], z[:, 1::2])) < bound          # the two halves of one 32-bit hash
    assert abs(corr(z[:, 1:-1:2], z[:, 2::2])) < bound        # neighbouring pairs
    assert abs(corr(z[:-1], z[1:])) < bound                   # neighbouring rows
    other_site = R.dropout_keep(np.arange(rows, dtype=np.uint64)[:, None], np.arange(cols, dtype=np.uint64)[None, :], p, 0x1234_5678_9ABC, 2)
    other_seed = R.dropout_keep(np.arange(rows, dtype=np.uint64)[:, None], np.arange(cols, dtype=np.uint64)[None, :], p, 0x1234_5678_9ABD, 1)
    for o in (other_site, other_seed):
        zo = (o - o.mean()) / o.std()
        assert abs(corr(z, zo)) < bound
    # rows beyond 2^32 use the high word
    hi = R.dropout_keep(np.array([[5], [5 + (1 << 32)]], dtype=np.uint64), np.arange(4096, dtype=np.uint64)[None, :], p, 7, 0)
    assert (hi[0] != hi[1]).mean() > 0.1
