"""GPU parity of the path ``bench.py`` times, at the size it times it.

  * ``ops.masked_ce_rows`` (the compact CE forward+backward over the MLM-masked rows) at V = 65 536 against the oracle's
    ``R.masked_lm_loss`` (training/train_encoder.py:301-305), ragged row counts {1, 63, 65, 1167, 1312}, with and without
    per-row weights;
  * (round 5) the step EXACTLY as BENCH executes it — four micro-batches per pass (32-row launches), two streams, this round's plan
    table and the table start-up tuning gives on the box — on multi-document and on single-document rows; and one training-mode
    step at the reference's default dropout 0.1 at 8L / 1024d against the oracle with the restated masks;
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
SMALL = dict(n_layer=8, n_embd=1024, n_head=8, T=1024, V=65536, mini=8, seed=41)
_cache = {}


def _small_problem(single_document=False):
    """Weights, one optimizer step's rows and the oracle's loss + gradients for them.  Multi-document rows (the mask path, the
    row >= 1 merge quirk per mini-batch): 64 rows = 8 micro-batches, with the oracle's running sums snapshotted after 4 and after
    8 micro-batches — the first 32 rows are the 4-micro-batch problem of rounds 2-4 (its losses carry 1/4 instead of 1/8: an
    exact factor of 2), all 64 the two-passes-of-four problem.  Single-document rows (what bench.py times): 32 rows."""
    key = ("ref", single_document)
    if key in _cache:
        return _cache[key]
    from omnibiote_amd import train_encoder as TE
    s = SMALL
    n_mb = 4 if single_document else 8
    cfg = R.RefConfig(block_size=s["T"], vocab_size=s["V"], n_layer=s["n_layer"], n_head=s["n_head"], n_embd=s["n_embd"])
    w = seeded_weights(cfg, s["seed"])
    rows = s["mini"] * n_mb
    ids = torch.from_numpy(TE.synthetic_rows(rows, s["T"], s["V"], np.random.default_rng(s["seed"] + (7 if single_document else 0)),
                                             single_document=single_document))
    np.random.seed(s["seed"])                                   # the stream TrainStep._host_prelude will draw from
    draw = torch.from_numpy(np.random.binomial(1, 0.15, (rows, s["T"])) != 0)
    masked_ids, mlm = R.mlm_corrupt(ids, draw)                  # train_encoder.py:273-279
    wb = {k: v.to(BF).float().requires_grad_(True) for k, v in w.items()}
    rope = R.cast_rope_table(R.rope_table(cfg.n_embd // cfg.n_head, s["T"]), BF)
    threads = torch.get_num_threads()
    run, snaps = 0.0, {}
    for j in range(n_mb):                                       # train_encoder.py:284-311, micro-batch by micro-batch
        sl = slice(j * s["mini"], (j + 1) * s["mini"])
        # the reference builds one mask per mini-batch (train_encoder.py:290-292; its row-0 quirk applies per mini-batch)
        dense = R.dense_mask_from_blocks(R.document_blocks(ids[sl].numpy()), s["T"]).unsqueeze(1)
        logits = R.model_forward(wb, cfg, masked_ids[sl], dense, rope=rope)
        lj = R.masked_lm_loss(logits, ids[sl], mlm[sl], n_mb)
        lj.backward()
        run += lj.item()
        del logits, lj
        if j + 1 in (4, 8):
            f = float(n_mb) / (j + 1)                           # n_mb / n_accum: 2.0 (exact) for the 4-of-8 snapshot, 1.0 otherwise
            snaps[j + 1] = (run * f, {k: v.grad * f for k, v in wb.items()})
    torch.set_num_threads(threads)
    _cache[key] = (cfg, w, ids, mlm, snaps)
    return _cache[key]


def _load_plans(which, per_pass):
    """The GEMM plan table the step runs with: a committed table of a bench run ("r03": rounds 3-4's tests; "r05": this round's,
    72+ shapes incl. the 32-row launches), or "startup": tuned on THIS box the way bench.py tunes before it times anything."""
    from omnibiote_amd import tune
    s = SMALL
    if which == "startup":
        for k in sorted({1, per_pass}):
            tune.tune_model_shapes(k * s["mini"] * s["T"], s["n_embd"], s["V"], device=DEV)
        return
    path = os.path.join(ROOT, "profiles", f"{which}_gemm_plans_small.json")
    assert os.path.exists(path), path
    tune.load_plans(path)


def _run_headline_step(per_pass, pipeline_streams, tuned, impl="masked", n_accum=4, single_document=False, plans="r03"):
    from omnibiote_amd import train_encoder as TE
    from omnibiote_amd import _lib as L
    from omnibiote_amd import tune
    s = SMALL
    cfg, w, ids, mlm, snaps = _small_problem(single_document)
    ref_loss, ref_grads = snaps[n_accum]
    rows = s["mini"] * n_accum
    ids, mlm = ids[:rows], mlm[:rows]
    m = _hip_model(cfg, w, s["T"])
    if tuned:   # a committed plan table (all GEMM structures, split-K for the compact dgrad, nearest-plan lookup) or this box's own
        _load_plans(plans, per_pass)
    try:
        opt = torch.optim.SGD(m.parameters(), lr=0.0)
        step = TE.TrainStep(m, opt, None, mini_batch_size=s["mini"], n_head=s["n_head"], lm_head_impl=impl, max_grad_norm=1e9,
                            pipeline_streams=pipeline_streams, micro_batches_per_pass=per_pass)
        np.random.seed(s["seed"])                               # the same Bernoulli draw as the oracle's (its first `rows` rows)
        out = step(ids.to(DEV), input_ids_host=ids.numpy())     # bench.py's calling convention: host copy -> no device round trip
        torch.cuda.synchronize()
        L.check_device_status("headline step")
        lists = step._mask_rows_host                            # the per-micro-batch masked-row lists the prelude built
        assert [int(r.numel()) for r in lists] == [int(mlm[j * s["mini"]:(j + 1) * s["mini"]].sum()) for j in range(n_accum)]
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
    print(f"[headline readout={impl} per_pass={per_pass} streams={pipeline_streams} tuned={plans if tuned else False} micro-batches={n_accum} "
          f"{'single' if single_document else 'multi'}-document] loss {loss:.4f} vs {ref_loss:.4f}; worst gradient {worst[2]} cos {worst[0]:.5f} rel {worst[1]:.4f}")
    return {k: p.grad.clone() for k, p in m.named_parameters()}, loss


@pytest.mark.parametrize("impl", ["masked", "dense"])
def test_headline_readout_step_small_config_vs_oracle(impl):
    """pipeline_streams=2, tuned plans, B = 8 x 4 micro-batches, one per pass: rounds 3's headline form ("masked"), and rounds 1-2's ("dense")."""
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


@pytest.mark.parametrize("plans", ["r05", "startup"])
def test_headline_step_as_bench_runs_it_four_per_pass_two_streams_vs_oracle(plans):
    """EXACTLY the execution BENCH times: micro_batches_per_pass = 4 (32-row launches, M = 32 768 in every
    GEMM), two streams with the per-layer backward order, the masked-positions readout with the rows-form last block, and the plan
    table of this round's bench run (profiles/r05_gemm_plans_small.json — every structure the tuner picked at these shapes) or the
    table bench.py's own start-up tuning produces on this box; two passes of four micro-batches (64 multi-document rows), so both
    streams carry a pass and the second accumulates in place behind the first."""
    g, loss = _run_headline_step(per_pass=4, pipeline_streams=2, tuned=True, impl="masked", n_accum=8, plans=plans)
    key = ("g4", "multi")
    if key in _cache:   # two plan tables, the same mathematics
        g0, loss0 = _cache[key]
        assert abs(loss0 - loss) <= 2e-3
        for k in g0:
            a, b = g0[k].float().flatten(), g[k].float().flatten()
            assert ((a - b).norm() / (a.norm() + 1e-30)).item() <= 0.02, k
    _cache[key] = (g, loss)


def test_headline_step_single_document_rows_four_per_pass_vs_oracle():
    """The rows bench.py times are single-document rows (the FLOP formula assumes full attention): no interior EOS, every
    key range [0, T), the attention kernels' no-mask-test fast path in every tile.  One pass of four micro-batches, this round's plans."""
    _run_headline_step(per_pass=4, pipeline_streams=2, tuned=True, impl="masked", n_accum=4, single_document=True, plans="r05")


def test_small_config_step_with_dropout_vs_oracle_with_the_restated_masks():
    """The reference's default regime (--dropout 0.1, train_encoder.py:445) at the small config's width and depth through the block
    path: 8L / 1024d / 8h, T = 1024, two multi-document rows, training mode — embedding, attention-probability and both
    residual-projection masks on in every block, the forward's keep bits feeding the backward, the masked gradient handed from
    block to block — against the oracle given the product's restated masks (oracle/omnibiote_ref.py dropout_scale_mask: PyTorch's
    own stream cannot be reproduced): logits, loss and every parameter's gradient at the bars of the dropout-free step."""
    from omnibiote_amd import model as M
    from omnibiote_amd import ops
    from omnibiote_amd import train_encoder as TE
    from omnibiote_amd import _lib as L
    from omnibiote_amd.masks import RangeMask
    s, p_drop, B = SMALL, 0.1, 2
    cfg = R.RefConfig(block_size=s["T"], vocab_size=s["V"], n_layer=s["n_layer"], n_head=s["n_head"], n_embd=s["n_embd"])
    w = seeded_weights(cfg, s["seed"] + 1)
    ids = torch.from_numpy(TE.synthetic_rows(B, s["T"], s["V"], np.random.default_rng(5), single_document=False))
    draw = torch.from_numpy(np.random.default_rng(6).random((B, s["T"])) < 0.15)
    masked_ids, mlm = R.mlm_corrupt(ids, draw)
    m = _hip_model(cfg, w, s["T"])
    TE.set_dropout(m, p_drop)
    m.train()
    torch.manual_seed(77)
    seeds = [M._new_seed() for _ in range(1 + s["n_layer"])]     # the embedding's, then one per block, in the order forward() draws them
    torch.manual_seed(77)
    logits = m(masked_ids.to(DEV), attn_mask=RangeMask.from_tokens(ids.to(DEV)))
    loss, dlogits = ops.masked_ce(logits, ids.to(DEV), mlm.to(DEV), 1)
    logits.backward(dlogits)
    torch.cuda.synchronize()
    L.check_device_status("dropout step")
    wb = {k: v.to(BF).float().requires_grad_(True) for k, v in w.items()}
    rope = R.cast_rope_table(R.rope_table(cfg.n_embd // cfg.n_head, s["T"]), BF)
    dense = R.dense_mask_from_blocks(R.document_blocks(ids.numpy()), s["T"]).unsqueeze(1)
    ref_logits = R.model_forward(wb, cfg, masked_ids, dense, rope=rope, dropout=(p_drop, seeds))
    ref_loss = R.masked_lm_loss(ref_logits, ids, mlm, 1)
    ref_loss.backward()
    rows = mlm.reshape(-1).nonzero().squeeze(1)
    got_l, want_l = logits.float().cpu().reshape(-1, s["V"])[rows], ref_logits.detach().reshape(-1, s["V"])[rows]
    assert (got_l - want_l).abs().max().item() <= 2e-2 and (got_l - want_l).abs().mean().item() <= 2e-3
    assert abs(loss.item() - ref_loss.item()) <= 0.02, (loss.item(), ref_loss.item())
    errs = []
    for k, p in m.named_parameters():
        got, want = p.grad.float().cpu().flatten(), wb[k].grad.flatten()
        assert torch.isfinite(got).all(), k
        errs.append(((torch.dot(got, want) / (got.norm() * want.norm() + 1e-30)).item(), ((got - want).norm() / (want.norm() + 1e-30)).item(), k))
    errs.sort()
    print(f"[dropout {p_drop} 8L/1024d] loss {loss.item():.4f} vs {ref_loss.item():.4f}; least-aligned gradients: "
          + "; ".join(f"{k} cos {c:.5f} rel {r:.4f}" for c, r, k in errs[:3]))
    bad = [(k, c, r) for c, r, k in errs if not (c >= 0.9995 and r <= 0.04)]
    assert not bad, bad


# --------------------------------------------------------------------------------------- config 5 at its depth and length
def test_large_config_full_depth_one_row_vs_oracle():
    """BASELINE config 5's model: 24L / 2048d / 16h, T = 1024, the 65 536-way readout (width_mult 85.33), loss and every
    gradient for one multi-document row against the oracle (~9 TFLOP of fp32 on the host: about a minute)."""
    _cache.clear()      # the small problem's fp32 tensors (3 GB) are no longer needed
    cfg = R.RefConfig(block_size=1024, vocab_size=65536, n_layer=24, n_head=16, n_embd=2048)
    _model_vs_oracle(cfg, B=1, T=1024, seed=51, n_docs=3, emb_bar=(0.25, 2e-2), logit_bar=(3e-3, 4e-4), grad_cos=0.999, grad_rel=0.05)
