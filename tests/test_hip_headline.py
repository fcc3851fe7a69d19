"""GPU parity of the path ``bench.py`` times, at the size it times it.

  * ``ops.masked_ce_rows`` (the compact CE forward+backward over the MLM-masked rows) at V = 65 536 against the oracle's
    ``R.masked_lm_loss`` (training/train_encoder.py:301-305), ragged row counts {1, 63, 65, 1167, 1312}, with and without
    per-row weights;
  * ``TrainStep(lm_head_impl="masked")`` — the headline readout since round 3: embeddings of the masked
    positions gathered, logits [~1229, 65 536] for those rows only, compact CE, split-K dgrad over K = 65 536 and
    accumulate-wgrad over K ~ 1229, d emb rows scattered back — and ``lm_head_impl="dense"`` (rounds 1-2's headline: logits of
    every position in the forward, the same backward) — on the small config (8L / 1024d / 8h, V = 65 536, T = 1024) at the
    REAL micro-batch of 8 rows, four micro-batches, through the host-side MLM prelude bench.py uses, on two streams and with
    the committed tuned plan table (all GEMM structures and the nearest-plan lookup for the ragged row counts), against the
    oracle run micro-batch by micro-batch in fp32 (train_encoder.py:284-311), which computes every position's logits and
    multiplies the unmasked ones by zero as the reference does; and the same steps with two micro-batches per pass;
  * BASELINE config 5 at depth: 24L / 2048d / 16h, T = 1024, one row end to end against the oracle.

The fp32 oracle costs ~12 s per 8-row micro-batch on the GPU box's 16 cores; it runs once per module."""
import os
import warnings

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import omnibiote_ref as R

from test_hip_configs45 import _hip_model, _model_vs_oracle, seeded_weights

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------ masked_ce_rows at V = 65 536
@pytest.mark.parametrize("n_rows", [1, 63, 65, 1167, 1312])
@pytest.mark.parametrize("weighted", [False, True])
def test_masked_ce_rows_full_vocabulary_vs_oracle(n_rows, weighted):
    from omnibiote_amd import ops
    V, M, n_accum = 65536, 2048, 16
    g = torch.Generator().manual_seed(100 + n_rows)
    logits = (torch.randn(M, V, generator=g) * 1.5).to(BF)
    targets = torch.randint(0, V, (M,), generator=g)
    rows = torch.sort(torch.randperm(M, generator=g)[:n_rows]).values
    logits[rows[0], targets[rows[0]]] = 9.0                        # one confident row: p(target) ~ 1, gradient row ~ 0
    w = (torch.rand(n_rows, generator=g) * 0.01 + 1e-4) if weighted else None
    # oracle, fp32: train_encoder.py:301-305 on the listed rows (every other row is multiplied by zero there)
    lg = logits[rows].float().requires_grad_(True)
    if weighted:   # a pass that covers several micro-batches: each row carries 1 / (masked count of its own micro-batch)
        ref_loss = (F.cross_entropy(lg, targets[rows], reduction="none") / n_accum * w).sum()
    else:
        ref_loss = R.masked_lm_loss(lg, targets[rows], torch.ones(n_rows, dtype=torch.bool), n_accum)
    ref_loss.backward()
    loss, dl = ops.masked_ce_rows(logits.to(DEV), targets.to(DEV), rows.to(DEV), n_accum,
                                  row_weights=None if w is None else w.to(DEV))
    assert tuple(dl.shape) == (n_rows, V) and dl.dtype == BF
    # the form without a row list (logits and targets hold the listed rows alone: the masked-rows readout) — bitwise the same
    loss_c, dl_c = ops.masked_ce_rows(logits[rows].contiguous().to(DEV), targets[rows].to(DEV), None, n_accum,
                                      row_weights=None if w is None else w.to(DEV))
    assert loss_c.item() == loss.item() and torch.equal(dl_c, dl)
    assert abs(loss.item() - ref_loss.item()) <= 2e-5 * abs(ref_loss.item()) + 1e-7, (loss.item(), ref_loss.item())
    got, want = dl.float().cpu(), lg.grad
    err = (got - want).abs()
    bad = err > 2.0 ** -7 * want.abs() + 1e-9            # one bf16 rounding of an fp32 value
    assert not bad.any(), (int(bad.sum()), err.max().item(), want.abs().max().item())
    # the target column carries (p - 1) * scale: the only negative entry of its row
    tcol = got[torch.arange(n_rows), targets[rows]]
    assert (tcol <= 0).all() and (got.sum(dim=1).abs() <= 2e-2 * got.abs().sum(dim=1) + 1e-12).all()


# ------------------------------------------------------------------------- the headline step at its own micro-batch size
SMALL = dict(n_layer=8, n_embd=1024, n_head=8, T=1024, V=65536, mini=8, n_accum=4, seed=41)
_cache = {}


def _small_problem():
    """Weights, one optimizer step's batch (32 multi-document rows) and the oracle's loss + gradients for it."""
    if "ref" in _cache:
        return _cache["ref"]
    from omnibiote_amd import train_encoder as TE
    s = SMALL
    cfg = R.RefConfig(block_size=s["T"], vocab_size=s["V"], n_layer=s["n_layer"], n_head=s["n_head"], n_embd=s["n_embd"])
    w = seeded_weights(cfg, s["seed"])
    rows = s["mini"] * s["n_accum"]
    ids = torch.from_numpy(TE.synthetic_rows(rows, s["T"], s["V"], np.random.default_rng(s["seed"]), single_document=False))
    np.random.seed(s["seed"])                                   # the stream TrainStep._host_prelude will draw from
    draw = torch.from_numpy(np.random.binomial(1, 0.15, (rows, s["T"])) != 0)
    masked_ids, mlm = R.mlm_corrupt(ids, draw)                  # train_encoder.py:273-279
    wb = {k: v.to(BF).float().requires_grad_(True) for k, v in w.items()}
    rope = R.cast_rope_table(R.rope_table(cfg.n_embd // cfg.n_head, s["T"]), BF)
    threads = torch.get_num_threads()
    ref_loss, per_mb = 0.0, []
    for j in range(s["n_accum"]):                               # train_encoder.py:284-311, micro-batch by micro-batch
        sl = slice(j * s["mini"], (j + 1) * s["mini"])
        # the reference builds one mask per mini-batch (train_encoder.py:290-292; its row-0 quirk applies per mini-batch)
        dense = R.dense_mask_from_blocks(R.document_blocks(ids[sl].numpy()), s["T"]).unsqueeze(1)
        logits = R.model_forward(wb, cfg, masked_ids[sl], dense, rope=rope)
        lj = R.masked_lm_loss(logits, ids[sl], mlm[sl], s["n_accum"])
        lj.backward()
        per_mb.append(lj.item())
        ref_loss += lj.item()
        del logits, lj
    torch.set_num_threads(threads)
    grads = {k: v.grad.clone() for k, v in wb.items()}
    _cache["ref"] = (cfg, w, ids, mlm, ref_loss, grads)
    return _cache["ref"]


def _run_headline_step(per_pass, pipeline_streams, tuned, impl="masked"):
    from omnibiote_amd import train_encoder as TE
    from omnibiote_amd import _lib as L
    from omnibiote_amd import tune
    s = SMALL
    cfg, w, ids, mlm, ref_loss, ref_grads = _small_problem()
    m = _hip_model(cfg, w, s["T"])
    if tuned:   # the plan table committed with the profiles: structures 1/2/3, split-K 12 for the compact dgrad, nearest-plan lookup
        tune.load_plans(os.path.join(ROOT, "profiles", "r03_gemm_plans_small.json"))
    try:
        opt = torch.optim.SGD(m.parameters(), lr=0.0)
        step = TE.TrainStep(m, opt, None, mini_batch_size=s["mini"], n_head=s["n_head"], lm_head_impl=impl, max_grad_norm=1e9,
                            pipeline_streams=pipeline_streams, micro_batches_per_pass=per_pass)
        np.random.seed(s["seed"])                               # the same Bernoulli draw as the oracle's
        out = step(ids.to(DEV), input_ids_host=ids.numpy())     # bench.py's calling convention: host copy -> no device round trip
        torch.cuda.synchronize()
        lists = step._mask_rows_host                            # the per-micro-batch masked-row lists the prelude built
        assert [int(r.numel()) for r in lists] == [int(mlm[j * s["mini"]:(j + 1) * s["mini"]].sum()) for j in range(s["n_accum"])]
    finally:
        if tuned:
            L.lib().obte_gemm_plan_clear()
            tune._done.clear()
    loss = out["loss"].item()
    assert abs(loss - ref_loss) <= 0.02, (loss, ref_loss)
    errs = []
    for k, p in m.named_parameters():
        got, want = p.grad.float().cpu().flatten(), ref_grads[k].flatten()
        assert torch.isfinite(got).all(), k
        cos = (torch.dot(got, want) / (got.norm() * want.norm() + 1e-30)).item()
        rel = ((got - want).norm() / (want.norm() + 1e-30)).item()
        errs.append((cos, rel, k))
    errs.sort()
    worst = errs[0]
    print("[headline] least-aligned gradients: " + "; ".join(f"{k} cos {c:.5f} rel {r:.4f}" for c, r, k in errs[:4]))
    bad = [(k, c, r) for c, r, k in errs if not (c >= 0.9995 and r <= 0.04)]   # the bars of test_hip_configs45.py's full-size tests
    assert not bad, bad
    # rows of the embedding no (masked) token selected receive no gradient at all
    touched = torch.zeros(s["V"], dtype=torch.bool)
    touched[R.mlm_corrupt(ids, mlm)[0].reshape(-1)] = True
    assert not m.transformer.wte.weight.grad[~touched.to(DEV)].any()
    print(f"[headline readout={impl} per_pass={per_pass} streams={pipeline_streams} tuned={tuned}] loss {loss:.4f} vs {ref_loss:.4f}; "
          f"worst gradient {worst[2]} cos {worst[0]:.5f} rel {worst[1]:.4f}")
    return {k: p.grad.clone() for k, p in m.named_parameters()}, loss


@pytest.mark.parametrize("impl", ["masked", "dense"])
def test_headline_readout_step_small_config_vs_oracle(impl):
    """pipeline_streams=2, tuned plans, B = 8 x 4 micro-batches: what BENCH times ("masked"), and rounds 1-2's form ("dense")."""
    _cache["g1", impl] = _run_headline_step(per_pass=1, pipeline_streams=2, tuned=True, impl=impl)
    if ("g1", "masked") in _cache and ("g1", "dense") in _cache:   # the two readouts: the same mathematics, far inside the bar
        (ga, la), (gb, lb) = _cache["g1", "masked"], _cache["g1", "dense"]
        assert abs(la - lb) <= 2e-3
        for k in ga:
            a, b = ga[k].float().flatten(), gb[k].float().flatten()
            assert ((a - b).norm() / (a.norm() + 1e-30)).item() <= 0.02, k


@pytest.mark.parametrize("impl", ["masked", "dense"])
def test_headline_step_two_micro_batches_per_pass_vs_oracle(impl):
    """micro_batches_per_pass=2 (16 rows per launch, every masked row weighted by its own micro-batch's count)."""
    g2, loss2 = _run_headline_step(per_pass=2, pipeline_streams=1, tuned=False, impl=impl)
    if ("g1", impl) in _cache:    # and the two executions of the same mathematics agree far inside the bar against the oracle
        g1, loss1 = _cache["g1", impl]
        assert abs(loss1 - loss2) <= 2e-3
        for k in g1:
            a, b = g1[k].float().flatten(), g2[k].float().flatten()
            assert ((a - b).norm() / (a.norm() + 1e-30)).item() <= 0.02, k


# --------------------------------------------------------------------------------------- config 5 at its depth and length
def test_large_config_full_depth_one_row_vs_oracle():
    """BASELINE config 5's model: 24L / 2048d / 16h, T = 1024, the 65 536-way readout (width_mult 85.33), loss and every
    gradient for one multi-document row against the oracle (~9 TFLOP of fp32 on the host: about a minute)."""
    _cache.clear()      # the small problem's fp32 tensors (3 GB) are no longer needed
    cfg = R.RefConfig(block_size=1024, vocab_size=65536, n_layer=24, n_head=16, n_embd=2048)
    _model_vs_oracle(cfg, B=1, T=1024, seed=51, n_docs=3, emb_bar=(0.25, 2e-2), logit_bar=(3e-3, 4e-4), grad_cos=0.999, grad_rel=0.05)
