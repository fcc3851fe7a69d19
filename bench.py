"""bench.py — masked-LM training throughput of the OmniBioTE encoder hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]          (N > 1: launched by torchrun, one rank per GPU)

A *step* is one optimizer step of the reference's training loop (training/train_encoder.py:241-323) on this rank's
128 rows of 1024 synthetic tokens: 16 accumulated micro-batches of mini_batch_size 8 through the drop-in
``OmniBioTA`` (small: 8L/1024d/8h, bf16, dropout 0), masked-LM loss over the full 65 536-way logits, backward,
global-norm clip, MuAdamW-grouped AdamW, LinearLR — BASELINE.json configs[1] at N=1 and configs[2]
(batch_size 1024 over 8 ranks) at N=8.  Per-rank work is fixed as N grows (weak scaling).  Inputs are resident in
HBM before the timed region.  Rank 0 prints ONE JSON line.

Extra objects: ``roofline`` for the dominant kernel family (bf16 MFMA GEMM; per-launch durations from HIP events
recorded on the launch stream by the library's opt-in profiler during one extra step right after the timed region:
bracketing every launch of the timed steps themselves with events was measured to cost 6 % of ``value`` — the events keep
the tail of one kernel from overlapping the head of the next — so the timed region runs un-instrumented), and, at N=1, ``cpu_baseline`` — the CPU oracle (oracle/omnibiote_ref.py, the reference's
arithmetic restated in plain torch) timed on this host on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
METRIC = "MLM train tokens/sec, small (8L/1024d) ctx=1024 at 1/2/4/8 MI355X"

CONFIGS = {
    "small": dict(n_layer=8, n_embd=1024, n_head=8, ctx_len=1024),
    "small4k": dict(n_layer=8, n_embd=1024, n_head=8, ctx_len=4096),
    "large": dict(n_layer=24, n_embd=2048, n_head=16, ctx_len=1024),
    "tiny": dict(n_layer=2, n_embd=128, n_head=2, ctx_len=128),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--config", default="small", choices=sorted(CONFIGS))
    p.add_argument("--rows_per_rank", type=int, default=128, help="rows per rank per optimizer step (batch_size / world)")
    p.add_argument("--mini_batch_size", type=int, default=8)
    p.add_argument("--multi_document", action="store_true", help="rows with interior EOS (block-diagonal masks)")
    p.add_argument("--masked_lm_head", action="store_true",
                   help="readout + CE on the MLM-masked rows only (SURVEY §8f rank 1; same loss/gradients, not the headline)")
    p.add_argument("--dropout", type=float, default=0.0, help="0.0 = the parity regime (default); 0.1 = the reference's default")
    p.add_argument("--no_cpu_baseline", action="store_true")
    p.add_argument("--no_roofline", action="store_true")
    p.add_argument("--dense_mask", action="store_true",
                   help="pass the reference's dense additive (B,H,T,T) mask (expand view) instead of key ranges")
    p.add_argument("--no_variants", action="store_true", help="skip the extra masked-rows-readout measurement")
    p.add_argument("--pipeline_streams", type=int, default=2, choices=[1, 2],
                   help="2: forward of micro-batch j+1 beside the backward of micro-batch j on a second HIP stream (bitwise the same results)")
    p.add_argument("--plan_cache", default="", help="JSON file of tuned GEMM plans: loaded if present (no tuning launches), else tuned and written")
    p.add_argument("--shapes_out", default="", help="write the per-shape launch table (from the profiler step) to this file")
    return p.parse_args()


def harness_args(cfg, a, world):
    from omnibiote_amd.train_encoder import parse_args
    h = parse_args([])
    h.batch_size = a.rows_per_rank * world
    h.mini_batch_size = a.mini_batch_size
    h.n_layer, h.n_embd, h.n_head, h.ctx_len = cfg["n_layer"], cfg["n_embd"], cfg["n_head"], cfg["ctx_len"]
    h.dropout = a.dropout
    h.token_budget = 20e9
    return h


KIND_NAMES = {12: "gemm_fwd(NT)", 13: "gemm_fwd(NT)+gelu", 14: "gemm_fwd(NT)+residual", 16: "gemm_fwd(NT)+residual+dropout",
              17: "gemm_fwd(NT)+rope", 8: "gemm_dgrad(NN)", 11: "gemm_dgrad(NN)+gelu_bwd", 0: "gemm_wgrad(TN)",
              2: "gemm_wgrad(TN)+accumulate", 32: "gemm_grouped(wgrads)", 34: "gemm_grouped(wgrads)+accumulate",
              33: "gemm_grouped(wgrads+dgrad)", 35: "gemm_grouped(wgrads+dgrad)+accumulate", 100: "attn_fwd", 101: "attn_bwd"}


def collect_profile(cap=200000):
    from omnibiote_amd import _lib
    ms = np.zeros(cap, dtype=np.float64)
    dims = np.zeros(3 * cap, dtype=np.int64)
    kind = np.zeros(cap, dtype=np.int32)
    n = _lib.lib().obte_profile_collect(ms.ctypes.data_as(ctypes.c_void_p), dims.ctypes.data_as(ctypes.c_void_p),
                                        kind.ctypes.data_as(ctypes.c_void_p), cap)
    return ms[:n], dims[:3 * n].reshape(n, 3), kind[:n]


def roofline_from_profile(ms, dims, kind, n_steps):
    """Group launches by kernel family; the dominant family (largest total time) becomes ``roofline``."""
    fam = {}
    for t, (d0, d1, d2), k in zip(ms, dims, kind):
        if k >= 100:
            # attention: fwd 4*T*T*D per (b,h) ; bwd 2.5x that (five products) — algorithmic, recompute not counted
            flops = 4.0 * d0 * d1 * d1 * d2 * (1.0 if k == 100 else 2.5)
            name = KIND_NAMES[int(k)]
        else:
            flops = 2.0 * d0 * d1 * d2
            name = "gemm_bf16_kernel"
        f = fam.setdefault(name, dict(time_ms=0.0, flops=0.0, launches=0))
        f["time_ms"] += float(t); f["flops"] += flops; f["launches"] += 1
    by_kind = {}
    for t, (d0, d1, d2), k in zip(ms, dims, kind):
        e = by_kind.setdefault(KIND_NAMES.get(int(k), str(int(k))), dict(time_ms=0.0, flops=0.0, launches=0))
        e["time_ms"] += float(t); e["launches"] += 1
        e["flops"] += 2.0 * d0 * d1 * d2 if k < 100 else 4.0 * d0 * d1 * d1 * d2 * (1.0 if k == 100 else 2.5)
    dom = max(fam, key=lambda n: fam[n]["time_ms"])
    f = fam[dom]
    achieved = f["flops"] / (f["time_ms"] * 1e-3) / 1e12
    roof = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": None,
            "launches_per_step": f["launches"] // max(n_steps, 1),
            "avg_launch_ms": round(f["time_ms"] / f["launches"], 4),
            "avg_launch_gflop": round(f["flops"] / f["launches"] / 1e9, 3),
            "share_of_profiled_time": round(f["time_ms"] / sum(x["time_ms"] for x in fam.values()), 3),
            "breakdown": {n: {"ms_per_step": round(e["time_ms"] / max(n_steps, 1), 3),
                               "tflops": round(e["flops"] / (e["time_ms"] * 1e-3) / 1e12, 1),
                               "launches_per_step": e["launches"] // max(n_steps, 1)} for n, e in sorted(by_kind.items())}}
    return roof


def cpu_baseline(cfg, mini_rows=1, steps=3, warmup=1, device="cpu", rows=None):
    """The oracle's train step (fwd + masked CE + bwd + clip + AdamW), bf16 like the GPU run: on the host cores (the
    `cpu_baseline` object), or — same module, same step, `device="cuda"` — as eager PyTorch-ROCm ops on the GPU the
    HIP path just ran on (`eager_gpu_baseline`: what the reference's own op set costs on this hardware)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import omnibiote_ref as R
    from omnibiote_amd import train_encoder as TE
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    rc = R.RefConfig(block_size=cfg["ctx_len"], vocab_size=2 ** 16, n_layer=cfg["n_layer"], n_head=cfg["n_head"], n_embd=cfg["n_embd"])
    torch.manual_seed(0)
    shapes = R.param_shapes(rc)
    w = {k: (torch.randn(s) * (1.0 if "wte" in k else 0.02) + (1.0 if "ln_" in k else 0.0)).bfloat16().to(device) for k, s in shapes.items()}
    enc = R.OracleEncoder(rc, w)
    enc.rope = R.cast_rope_table(R.rope_table(rc.n_embd // rc.n_head, rc.block_size), torch.bfloat16).to(device)
    opt = torch.optim.AdamW(enc.parameters(), lr=1e-3)
    step = TE.TrainStep(enc, opt, None, mini_batch_size=mini_rows, n_head=rc.n_head, loss_impl="torch", mask_impl="dense")
    rng = np.random.default_rng(0)
    rows = rows or mini_rows
    ids = torch.from_numpy(TE.synthetic_rows(rows, cfg["ctx_len"], 2 ** 16, rng)).to(device)
    times = []
    for i in range(warmup + steps):
        if device != "cpu":
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(ids)
        if device != "cpu":
            torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[warmup:]))
    where = f"host cores ({threads} threads; host has {os.cpu_count()} logical CPUs)" if device == "cpu" else "eager PyTorch-ROCm ops on cuda:0"
    return {"value": round(rows * cfg["ctx_len"] / t, 1), "unit": "tokens/s", "cores": threads if device == "cpu" else 0, "kind": "port",
            "sample": f"oracle train step (fwd+masked CE+bwd+clip+AdamW) on {where}, {rows} row(s) x {cfg['ctx_len']} tokens in "
                      f"micro-batches of {mini_rows}, bf16, dense additive masks, median of {steps} steps after {warmup} warm-up"}


def main():
    a = parse()
    cfg = CONFIGS[a.config]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        dist.init_process_group("nccl")   # RCCL
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from omnibiote_amd import _lib
    from omnibiote_amd import train_encoder as TE
    _lib.lib()   # fail loudly, before any timing, if the HIP library is missing

    h = harness_args(cfg, a, world)
    torch.manual_seed(1234)
    np.random.seed(1234 + rank)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        m = TE.build_model(h, dev)
    n_params = m.get_num_params()
    from omnibiote_amd import tune
    if a.plan_cache and os.path.exists(a.plan_cache):
        tune.load_plans(a.plan_cache)
    else:
        tune.tune_model_shapes(a.mini_batch_size * cfg["ctx_len"], cfg["n_embd"], 2 ** 16, device=dev, verbose=(rank == 0 and bool(a.shapes_out)))
        if a.plan_cache and rank == 0:
            tune.save_plans(a.plan_cache)
    force_ddp = os.environ.get("OBTE_FORCE_DDP") == "1"   # rehearse the N>1 code path on one GPU
    if force_ddp and world == 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1)
    model = TE.wrap_ddp(m, local) if (world > 1 or force_ddp) else m
    total_iters = 1000
    opt, sched = TE.build_optimizer(m, h, total_iters)
    step = TE.TrainStep(model, opt, sched, mini_batch_size=a.mini_batch_size, n_head=cfg["n_head"],
                        lm_head_impl="masked" if a.masked_lm_head else "dense", pipeline_streams=a.pipeline_streams,
                        mask_impl="dense" if a.dense_mask else "ranges")
    rng = np.random.default_rng(1234 + rank)
    T = cfg["ctx_len"]
    # synthetic batches resident in HBM before timing; a fresh one per step
    batches = [torch.from_numpy(TE.synthetic_rows(a.rows_per_rank, T, 2 ** 16, rng, single_document=not a.multi_document)).to(dev)
               for _ in range(min(a.steps + a.warmup, 4))]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    losses = []
    for i in range(a.warmup):
        losses.append(step(batches[i % len(batches)])["loss"])
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        losses.append(step(batches[(a.warmup + i) % len(batches)])["loss"])
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    tokens_per_step = a.rows_per_rank * T * world   # no PAD in the synthetic rows: all tokens count (train_encoder.py:350)
    value = tokens_per_step * a.steps / elapsed
    fpt = TE.flops_per_token(n_params, cfg["n_layer"], cfg["n_embd"], T)

    roofline = None
    if not a.no_roofline and rank == 0:
        # per-launch durations are only meaningful when launches do not share the chip: the profiled step runs on one
        # stream (the timed steps above overlap two micro-batches, which stretches every kernel that has company)
        step.pipeline_streams = 1
        _lib.lib().obte_profile_enable(1)
        step(batches[0])
        torch.cuda.synchronize()
        ms, dims, kind = collect_profile()
        _lib.lib().obte_profile_enable(0)
        step.pipeline_streams = a.pipeline_streams
        if len(ms):
            roofline = roofline_from_profile(ms, dims, kind, 1)
            # HBM bytes per launch of the GEMM family: PMC counters cannot be read from inside this process, so the
            # figure comes from the committed rocprofv3 --pmc passes over this same workload (tools/pmc_traffic.py)
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_gemm_family_traffic.json")
            default_workload = (a.config == "small" and not a.masked_lm_head and a.dropout == 0.0 and not a.multi_document
                                and a.rows_per_rank == 128 and a.mini_batch_size == 8)
            if default_workload and os.path.exists(pmc):
                with open(pmc) as f:
                    t = json.load(f)
                roofline["traffic"] = round(t["traffic_bytes_per_launch"])
                roofline["traffic_unit"] = "bytes/launch (2 x FETCH_SIZE + WRITE_SIZE, profiles/r01_pmc_gemm_family_traffic.json)"
            if a.shapes_out:
                tab = {}
                for t, d, k in zip(ms, dims, kind):
                    key = (KIND_NAMES.get(int(k), str(int(k))), int(d[0]), int(d[1]), int(d[2]))
                    e = tab.setdefault(key, [0, 0.0])
                    e[0] += 1; e[1] += float(t)
                with open(a.shapes_out, "w") as f:
                    f.write(f"{'kernel':28s} {'d0':>7s} {'d1':>7s} {'d2':>7s} {'calls':>6s} {'avg_us':>9s} {'TFLOP/s':>8s} {'ms/step':>8s}\n")
                    for (name, d0, d1, d2), (n, tt) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
                        fl = 2.0 * d0 * d1 * d2 if not name.startswith("attn") else 4.0 * d0 * d1 * d1 * d2 * (1.0 if name == "attn_fwd" else 2.5)
                        f.write(f"{name:28s} {d0:7d} {d1:7d} {d2:7d} {n:6d} {tt / n * 1e3:9.1f} {fl * n / (tt * 1e-3) / 1e12:8.1f} {tt:8.2f}\n")
    # Not the headline: the same step with the readout and the cross entropy restricted to the ~15 % MLM-masked rows
    # (TrainStep lm_head_impl="masked": same loss and gradients, rows outside the mask contribute exact zeros).  Reported
    # beside `value`, which keeps the reference's full logits.
    variants = None
    if not a.masked_lm_head and not a.no_variants:
        step.lm_head_impl = "masked"
        for i in range(2):
            step(batches[i % len(batches)])
        sync()
        t0 = time.perf_counter()
        for i in range(3):
            step(batches[i % len(batches)])
        sync()
        el = time.perf_counter() - t0
        if world > 1:
            te = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el = float(te.item())
        step.lm_head_impl = "dense"
        variants = {"masked_rows_readout": {"value": round(tokens_per_step * 3 / el, 1), "unit": "tokens/s", "steps": 3,
                                            "note": "readout + CE on the MLM-masked rows only; identical loss and gradients; not the headline"}}
    if world > 1:
        dist.barrier()

    if rank == 0:
        out = {
            "metric": METRIC if a.config == "small" else f"MLM train tokens/sec, {a.config} ctx={T}", "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"OmniBioTA {a.config} ({cfg['n_layer']}L/{cfg['n_embd']}d/{cfg['n_head']}h) ctx={T} MLM train step: "
                                   f"{a.rows_per_rank} rows/rank = {a.rows_per_rank // a.mini_batch_size} micro-batches of {a.mini_batch_size}, "
                                   f"{'masked-rows-only' if a.masked_lm_head else 'full'} 65536-way logits, dropout {a.dropout:g}, {'multi' if a.multi_document else 'single'}-document rows",
                       "global_batch_rows": a.rows_per_rank * world, "mini_batch_size": a.mini_batch_size, "seq_len": T,
                       "parallelism": f"dp{world}", "dropout": a.dropout, "vocab": 65536},
            "flops_per_token": fpt,
            "mfma_fraction_whole_step": round(value * fpt / (PEAK_BF16_TFLOPS * 1e12 * world), 4),
            "final_loss": round(float(losses[-1].item()), 4),
            "roofline": roofline,
        }
        if variants:
            out["variants"] = variants
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
            try:   # same oracle step as eager torch ops on this GPU (informational; never the product path)
                del step, opt, model, m
                torch.cuda.empty_cache()
                out["eager_gpu_baseline"] = cpu_baseline(cfg, mini_rows=a.mini_batch_size, steps=2, warmup=1, device="cuda", rows=a.rows_per_rank)
            except Exception as e:   # e.g. out of memory on a large config: the headline does not depend on it
                out["eager_gpu_baseline"] = {"value": None, "note": repr(e)[:200]}
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
