"""Is the one-kernel attention backward reproducible when other work shares the card?

  (1) op level: the backward of one (B, H, T) problem alone, then the same call repeated while a second stream runs GEMMs and
      attention forwards — every repeat must equal the quiet result bit for bit (the hand-off chain fixes the summation order);
  (2) step level: the small config's training step (multi-document rows, 8 x 4 micro-batches) with one and with two streams,
      in both backward forms, gradients compared run to run and form to form.

    python tools/fused_race.py [--reps 40] [--skip_step]"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnibiote_amd import ops, masks, _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--skip_step", action="store_true")
ap.add_argument("--skip_op", action="store_true")
a = ap.parse_args()
dev = "cuda"


def op_level(B, H, T, hs, multi):
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(B, T, 3 * H * hs, device=dev, generator=g).to(torch.bfloat16)
    d_o = torch.randn(B, T, H * hs, device=dev, generator=g).to(torch.bfloat16)
    tok = torch.randint(20, 100, (B, T), device=dev)
    if multi:
        for b in range(B):
            tok[b, torch.randint(8, T - 8, (3,))] = 3
    spec = ops.MaskSpec(ranges=masks.RangeMask.from_tokens(tok).key_ranges)
    scale = 8.0 / (H * hs)
    o, lse = ops.attn_fwd(qkv, B, T, H, hs, scale, spec)
    quiet = ops.attn_bwd(qkv, o, d_o, lse, B, T, H, hs, scale, spec, one_kernel=True).clone()
    pair = ops.attn_bwd(qkv, o, d_o, lse, B, T, H, hs, scale, spec, one_kernel=False)
    torch.cuda.synchronize()
    print(f"[op B{B} H{H} T{T} multi={multi}] one-kernel vs pair: max abs diff {(quiet.float() - pair.float()).abs().max().item():.4g}", flush=True)
    side = torch.cuda.Stream()
    x = torch.randn(8192, 1024, device=dev).to(torch.bfloat16)
    w = torch.randn(4096, 1024, device=dev).to(torch.bfloat16)
    for kind, one in (("gemm", False), ("gemm", True), ("attn_fwd", True), ("both", True)):
        bad = 0
        ref = quiet if one else pair
        for r in range(a.reps):
            with torch.cuda.stream(side):
                for _ in range(3):
                    if kind in ("gemm", "both"):
                        ops.linear_fwd(x, w)
                    if kind in ("attn_fwd", "both"):
                        ops.attn_fwd(qkv, B, T, H, hs, scale, spec)
            got = ops.attn_bwd(qkv, o, d_o, lse, B, T, H, hs, scale, spec, one_kernel=one)
            torch.cuda.synchronize()
            if not torch.equal(got, ref):
                d = (got.float() - ref.float()).abs().reshape(B, T, 3, H, hs)
                bad += 1
                if bad <= 3:
                    print(f"   rep {r} beside {kind}: {int((d > 0).sum())} elements differ, max {d.max().item():.4g} (|ref| max per part "
                          f"{[round(ref.float().reshape(B, T, 3, H, hs)[:, :, i].abs().max().item(), 4) for i in range(3)]})", flush=True)
                    for bb, hh in torch.nonzero(d.amax(dim=(1, 2, 4)) > 0).tolist()[:6]:
                        per = d[bb, :, :, hh].reshape(T // 32, 32, 3, hs).amax(dim=(1, 3))   # (slice of 32 rows, part)
                        print(f"      b {bb} h {hh}: 32-row slices that differ, dQ {torch.nonzero(per[:, 0] > 0).flatten().tolist()} "
                              f"dK {torch.nonzero(per[:, 1] > 0).flatten().tolist()} dV {torch.nonzero(per[:, 2] > 0).flatten().tolist()}; "
                              f"max {per.amax(dim=0).tolist()}", flush=True)
        print(f"[op B{B} H{H} T{T} multi={multi}] {'one-kernel' if one else 'kernel pair'} beside {kind}: {bad}/{a.reps} repeats differ from the quiet result", flush=True)


def step_level():
    from omnibiote_amd import train_encoder as TE
    from types import SimpleNamespace
    s = dict(n_layer=8, n_embd=1024, n_head=8, T=1024, V=65536, mini=8, n_accum=4, seed=41)
    rows = s["mini"] * s["n_accum"]
    ids = torch.from_numpy(TE.synthetic_rows(rows, s["T"], s["V"], np.random.default_rng(s["seed"]), single_document=False))
    torch.manual_seed(0)
    m = TE.build_model(SimpleNamespace(dropout=0.0, ctx_len=s["T"], n_embd=s["n_embd"], n_layer=s["n_layer"], n_head=s["n_head"],
                                       disable_flash=False, checkpoint_freq=0), dev)
    res = {}
    for form in ("two", "one"):
        L.lib().obte_attn_bwd_select(1 if form == "two" else 0)
        for streams in (1, 2):
            runs = []
            for r in range(3):
                for p in m.parameters():
                    p.grad = None
                opt = torch.optim.SGD(m.parameters(), lr=0.0)
                step = TE.TrainStep(m, opt, None, mini_batch_size=s["mini"], n_head=s["n_head"], lm_head_impl="masked", max_grad_norm=1e9,
                                    pipeline_streams=streams, micro_batches_per_pass=1)
                np.random.seed(s["seed"])
                step(ids.to(dev), input_ids_host=ids.numpy())
                torch.cuda.synchronize()
                runs.append({k: p.grad.clone() for k, p in m.named_parameters()})
            res[form, streams] = runs
            for r in (1, 2):
                diff = [(k, ((runs[r][k].float() - runs[0][k].float()).norm() / (runs[0][k].float().norm() + 1e-30)).item()) for k in runs[0]]
                diff = sorted([d for d in diff if d[1] > 0], key=lambda t: -t[1])
                print(f"[step form={form} streams={streams}] run {r} vs run 0: {len(diff)} tensors differ; worst {diff[:3]}", flush=True)
    L.lib().obte_attn_bwd_select(0)
    base = res["two", 1][0]
    for key, runs in res.items():
        for r, g in enumerate(runs):
            diff = sorted([(((g[k].float() - base[k].float()).norm() / (base[k].float().norm() + 1e-30)).item(), k) for k in g], reverse=True)
            print(f"[step {key} run {r}] vs two-kernel/one-stream run 0: worst rel {diff[0][0]:.4f} {diff[0][1]}; next {diff[1][0]:.4f} {diff[1][1]}", flush=True)


if not a.skip_op:
    op_level(8, 8, 1024, 128, True)
    op_level(8, 8, 1024, 128, False)
if not a.skip_step:
    step_level()
