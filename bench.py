"""bench.py — masked-LM training throughput of the OmniBioTE encoder hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]          (N > 1: launched by torchrun, one rank per GPU)

A *step* is one optimizer step of the reference's training loop (training/train_encoder.py:241-323) on this rank's
128 rows of 1024 synthetic tokens: 16 accumulated micro-batches of mini_batch_size 8 (k of them per forward/backward pass,
k in {1, 2, 4} chosen by measurement at start-up and named in ``config``: masks and loss normalisation stay per micro-batch) through the drop-in
``OmniBioTA`` (small: 8L/1024d/8h, bf16, dropout 0), the 65 536-way readout and masked-LM cross entropy evaluated on the
MLM-masked positions only (``--readout masked``, the default since round 3: the loss multiplies every other position by
zero, train_encoder.py:304, so loss and gradients are the reference's; the last block's MLP half, its attention's queries with
the attention projection (round 5; keys and values of every position) and ln_f run on those positions too — ``model.forward(rows=...)``; ``--readout dense`` / ``dense_full`` compute every position's logits as
rounds 1-2 did and are reported as variants), backward, global-norm clip, MuAdamW-grouped AdamW, LinearLR — BASELINE.json
configs[1] at N=1 and configs[2] (batch_size 1024 over 8 ranks) at N=8.  At N=1 the line also carries ``other_configs``:
driver-timed steps of BASELINE configs 4 (small, ctx 4096) and 5 (large 24L/2048d/16h, its single-GPU leg).  Per-rank work is fixed as N grows (weak scaling).  Inputs are resident in
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

READOUT_TEXT = {"dense": "full 65536-way logits for every position in the forward, readout backward over the MLM-masked rows (the other rows of d(logits) are exact zeros)",
                "dense_full": "full 65536-way logits and dense d(logits)",
                "masked": "65536-way readout + CE on the MLM-masked positions only (SURVEY §8f rank 1: the loss multiplies every other position by zero, "
                          "train_encoder.py:304 — same loss, same gradients); the positions are handed to model.forward(rows=...), so the last block's MLP half, "
                          "its attention's queries (keys and values of every position) with the attention projection, and ln_f run on them alone as well"}
# leads config.workload (the driver keeps the first 128 characters of it): which readout ran
READOUT_LEAD = {"masked": "masked-positions readout+CE, rows-form last block", "masked_full": "masked-positions readout+CE, full last block",
                "dense": "every-position logits fwd, masked-rows readout bwd", "dense_full": "reference-literal dense logits + dense dlogits"}
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
    p.add_argument("--readout", default="masked", choices=["dense", "dense_full", "masked"],
                   help="masked (default, SURVEY §8f rank 1): readout + CE on the MLM-masked positions only — the loss multiplies "
                        "every other position by zero (train_encoder.py:304), so loss and gradients are the reference's.  dense: logits "
                        "of every position in the forward, the backward contracts over the masked rows only.  dense_full: also the "
                        "dense [M,V] d(logits) and full-size backward products (the reference's literal graph)")
    p.add_argument("--masked_lm_head", action="store_true", help="alias of --readout masked")
    p.add_argument("--dropout", type=float, default=0.0, help="0.0 = the parity regime (default); 0.1 = the reference's default")
    p.add_argument("--no_cpu_baseline", action="store_true")
    p.add_argument("--no_roofline", action="store_true")
    p.add_argument("--dense_mask", action="store_true",
                   help="pass the reference's dense additive (B,H,T,T) mask (expand view) instead of key ranges")
    p.add_argument("--no_other_configs", action="store_true", help="skip the driver-timed steps of BASELINE configs 4 (small4k) and 5 (large)")
    p.add_argument("--no_variants", action="store_true", help="skip the extra measurements (masked-rows readout, dropout 0.1, dense-mask calling convention)")
    p.add_argument("--pipeline_streams", type=int, default=2, choices=[1, 2, 3],
                   help="2: forward of micro-batch j+1 beside the backward of micro-batch j on a second HIP stream (bitwise the same results)")
    p.add_argument("--full_last_block", action="store_true",
                   help="readout masked: run the last block's MLP half and ln_f on every position (model.forward(return_embeddings=True), rows "
                        "gathered afterwards) instead of handing the masked positions to model.forward(rows=...)")
    p.add_argument("--backward_order", default="layer", choices=["layer", "pass"],
                   help="two streams: order the backward passes per parameter group (the next backward follows one layer behind) or per pass; "
                        "bitwise the same results")
    p.add_argument("--micro_batches_per_pass", type=int, default=0,
                   help="k micro-batches per forward/backward pass (k * mini_batch_size rows per launch; masks and the loss normalisation stay "
                        "per micro-batch, so --mini_batch_size keeps the reference's meaning and loss / gradients are those of separate passes). "
                        "An execution option like the GEMM plans.  0 (default): k in {1, 2, 4} timed at start-up over two steps each, the fastest kept")
    p.add_argument("--grad_exchange", default="allreduce", choices=["allreduce", "all_links"],
                   help="N > 1: DDP's bucketed all-reduce (the reference's), or the direct reduce-scatter + all-gather over all xGMI links "
                        "(omnibiote_amd/comm.py); either way per-bucket device events are recorded and reported in config.collectives")
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
              33: "gemm_grouped(wgrads+dgrad)", 35: "gemm_grouped(wgrads+dgrad)+accumulate", 100: "attn_fwd", 101: "attn_bwd",
              102: "attn_fwd(rows: last block)", 103: "attn_bwd(rows: last block)",
              110: "ln_fwd", 111: "ln_bwd", 112: "masked_ce_rows", 113: "adamw"}
HBM_KINDS = {110: lambda d0, d1, d2: 4.0 * d0 * d1, 111: lambda d0, d1, d2: (6.0 + 2.0 * d2) * d0 * d1,
             112: lambda d0, d1, d2: 4.0 * d0 * d1, 113: lambda d0, d1, d2: 14.0 * d0}   # algorithmic bytes per launch
PEAK_HBM_TBS = 8.0   # MI355X HBM3E (MI355X_MICROARCH.md)
# profiler kind = code above + 1000 * kernel structure: the name rocprofv3 --kernel-trace shows for that launch
STRUCT_NAMES = {1: "gemm_bf16_kernel", 2: "gemm_v2_kernel", 3: "gemm_v3_kernel", 4: "gemm_v4_kernel", 7: "gemm_v7_kernel"}
GEMM_FAMILY = "bf16 MFMA GEMM family: " + ", ".join(sorted(set(STRUCT_NAMES.values()))) + ", gemm_v3_group_kernel"


def rocprof_name(k: int) -> str:
    code, struct = int(k) % 1000, int(k) // 1000
    if code >= 100:
        return KIND_NAMES.get(code, str(code))
    if 32 <= code < 40:
        return "gemm_v3_group_kernel"
    return STRUCT_NAMES.get(struct, "gemm_v2_kernel")


def launch_flops(code: int, d0, d1, d2) -> float:
    if code in (100, 101):   # attention: fwd 4*T*T*D per (b,h); bwd 2.5x (five products) — algorithmic, recompute not counted
        return 4.0 * d0 * d1 * d1 * d2 * (1.0 if code == 100 else 2.5)
    if code in (102, 103):   # queries at listed rows only (the last block): d0 = queries x heads, d1 = keys
        return 4.0 * d0 * d1 * d2 * (1.0 if code == 102 else 2.5)
    if code in HBM_KINDS:
        return 0.0
    return 2.0 * d0 * d1 * d2


def collect_profile(cap=200000):
    from omnibiote_amd import _lib
    ms = np.zeros(cap, dtype=np.float64)
    dims = np.zeros(3 * cap, dtype=np.int64)
    kind = np.zeros(cap, dtype=np.int32)
    n = _lib.lib().obte_profile_collect(ms.ctypes.data_as(ctypes.c_void_p), dims.ctypes.data_as(ctypes.c_void_p),
                                        kind.ctypes.data_as(ctypes.c_void_p), cap)
    return ms[:n], dims[:3 * n].reshape(n, 3), kind[:n]


def roofline_from_profile(ms, dims, kind, n_steps):
    """Group launches by kernel family; the dominant family (largest total time) becomes ``roofline``; inside it every
    kernel is also listed under the name rocprofv3 shows for it, with the dominant single kernel named."""
    fam, by_kind, by_kernel = {}, {}, {}
    hbm = {}
    for t, (d0, d1, d2), k in zip(ms, dims, kind):
        code = int(k) % 1000
        if code in HBM_KINDS:   # HBM-bound kernels: algorithmic bytes / time against the 8 TB/s roof, reported beside the MFMA family
            e = hbm.setdefault(KIND_NAMES[code], dict(time_ms=0.0, bytes=0.0, launches=0))
            e["time_ms"] += float(t); e["bytes"] += HBM_KINDS[code](float(d0), float(d1), float(d2)); e["launches"] += 1
            continue
        flops = launch_flops(code, d0, d1, d2)
        name = GEMM_FAMILY if code < 100 else KIND_NAMES.get(code, str(code))
        for table, key in ((fam, name), (by_kind, KIND_NAMES.get(code, str(code))), (by_kernel, rocprof_name(k))):
            e = table.setdefault(key, dict(time_ms=0.0, flops=0.0, launches=0))
            e["time_ms"] += float(t); e["flops"] += flops; e["launches"] += 1
    dom = max(fam, key=lambda n: fam[n]["time_ms"])
    f = fam[dom]
    achieved = f["flops"] / (f["time_ms"] * 1e-3) / 1e12
    n_steps = max(n_steps, 1)

    def row(e):
        tf = e["flops"] / (e["time_ms"] * 1e-3) / 1e12
        return {"ms_per_step": round(e["time_ms"] / n_steps, 3), "tflops": round(tf, 1), "frac": round(tf / PEAK_BF16_TFLOPS, 4),
                "launches_per_step": e["launches"] // n_steps, "avg_launch_us": round(e["time_ms"] / e["launches"] * 1e3, 1)}
    kernels = {n: row(e) for n, e in sorted(by_kernel.items(), key=lambda kv: -kv[1]["time_ms"])}
    gemm_kernels = [n for n in kernels if n.startswith("gemm")]
    roof = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": None,
            "launches_per_step": f["launches"] // n_steps,
            "avg_launch_ms": round(f["time_ms"] / f["launches"], 4),
            "avg_launch_gflop": round(f["flops"] / f["launches"] / 1e9, 3),
            "share_of_profiled_time": round(f["time_ms"] / sum(x["time_ms"] for x in fam.values()), 3),
            # demangled BASE name (template arguments dropped): every instantiation of a kernel template — one per operand layout and
            # epilogue — is summed under it, here and in profiles/rNN_bench_small_families.txt (tools/ktrace_summary.py).  rocprofv3's
            # own kernel_stats.csv lists instantiations separately, so its top single ROW can be another kernel (e.g. the
            # non-template gemm_v3_group_kernel): compare like with like.
            "dominant_kernel_by_rocprof_name": gemm_kernels[0] if (dom == GEMM_FAMILY and gemm_kernels) else dom,
            "dominant_kernel_naming": "demangled base name, template instantiations summed (kernel_stats.csv rows are per instantiation)",
            "by_rocprof_kernel": kernels,
            "breakdown": {n: row(e) for n, e in sorted(by_kind.items())},
            "hbm_bound_kernels": {n: {"ms_per_step": round(e["time_ms"] / n_steps, 3), "launches_per_step": e["launches"] // n_steps,
                                      "achieved_TBps": round(e["bytes"] / (e["time_ms"] * 1e-3) / 1e12, 2),
                                      "frac_of_hbm_peak": round(e["bytes"] / (e["time_ms"] * 1e-3) / 1e12 / PEAK_HBM_TBS, 3)}
                                  for n, e in sorted(hbm.items())}}
    return roof


def oracle_step_rate(cfg, mini_rows, steps, warmup, device, rows, dtype, threads, budget_s=45.0, dropout=0.0):
    """tokens/s of the oracle's train step (fwd + masked CE + bwd + clip + AdamW) on ``device``; median of the timed steps.
    ``budget_s`` bounds the sample: timing stops early (never below 3 timed steps) once that much wall time is spent."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import omnibiote_ref as R
    from omnibiote_amd import train_encoder as TE
    torch.set_num_threads(threads)
    rc = R.RefConfig(block_size=cfg["ctx_len"], vocab_size=2 ** 16, n_layer=cfg["n_layer"], n_head=cfg["n_head"], n_embd=cfg["n_embd"])
    torch.manual_seed(0)
    shapes = R.param_shapes(rc)
    w = {k: (torch.randn(s) * (1.0 if "wte" in k else 0.02) + (1.0 if "ln_" in k else 0.0)).to(dtype).to(device) for k, s in shapes.items()}
    enc = R.OracleEncoder(rc, w)
    enc.torch_dropout_p = float(dropout)   # > 0: the reference's four dropout sites with torch's generator (its default regime)
    enc.train()
    enc.rope = R.cast_rope_table(R.rope_table(rc.n_embd // rc.n_head, rc.block_size), dtype).to(device)
    opt = torch.optim.AdamW(enc.parameters(), lr=1e-3)
    step = TE.TrainStep(enc, opt, None, mini_batch_size=mini_rows, n_head=rc.n_head, loss_impl="torch", mask_impl="dense")
    rng = np.random.default_rng(0)
    ids = torch.from_numpy(TE.synthetic_rows(rows, cfg["ctx_len"], 2 ** 16, rng)).to(device)
    times, t_begin = [], time.perf_counter()
    for i in range(warmup + steps):
        if device != "cpu":
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(ids)
        if device != "cpu":
            torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        if len(times) - warmup >= 3 and time.perf_counter() - t_begin > budget_s:
            break
    timed = times[warmup:]
    return rows * cfg["ctx_len"] / float(np.median(timed)), len(timed)


def attention_work_fraction(ids_host, mini: int) -> dict:
    """What share of the full T x T attention a batch of packed rows needs: `pair_fraction` = allowed (query, key) pairs / T^2 (the
    algorithmic share), `tile_fraction` = the share of tiles the kernels visit — forward: per block of 256 queries the 64-key tiles
    that meet the union of its rows' key ranges; backward: per block of 256 keys the 32-query slices inside the union of its keys'
    query ranges (symmetric masks: a key's queries are its own range); weighted 1 : 2 like the 4 L C T + 8 L C T of the reference's
    formula.  Key ranges from the product's own mask builder (the reference's quirk per mini-batch group)."""
    from omnibiote_amd import masks
    kr = masks.RangeMask.from_tokens(torch.from_numpy(np.asarray(ids_host)), group=mini).key_ranges.numpy().astype(np.int64)
    rows, T, _ = kr.shape
    s, e = kr[..., 0], kr[..., 1]
    pair = float((e - s).clip(min=0).sum()) / (rows * T * T)
    fwd_tiles = bwd_tiles = 0
    nq, nk = (T + 255) // 256, (T + 63) // 64
    for b0 in range(nq):
        lo = s[:, b0 * 256:(b0 + 1) * 256].min(axis=1)
        hi = e[:, b0 * 256:(b0 + 1) * 256].max(axis=1)
        fwd_tiles += int(((hi + 63) // 64 - lo // 64).clip(min=0).sum())
    for k0 in range((T + 255) // 256):
        lo = s[:, k0 * 256:(k0 + 1) * 256].min(axis=1)
        hi = e[:, k0 * 256:(k0 + 1) * 256].max(axis=1)
        bwd_tiles += int(((hi + 31) // 32 - lo // 32).clip(min=0).sum())
    f_fwd = fwd_tiles / float(rows * nq * nk)
    f_bwd = bwd_tiles / float(rows * ((T + 255) // 256) * ((T + 31) // 32))
    return {"pair_fraction": round(pair, 4), "forward_tile_fraction": round(f_fwd, 4), "backward_tile_fraction": round(f_bwd, 4),
            "tile_fraction": round((f_fwd + 2.0 * f_bwd) / 3.0, 4)}


def usable_cores() -> int:
    """Cores this process may actually run on: the affinity mask, capped by the cgroup CPU quota (a container sees all
    of the host's logical CPUs in os.cpu_count() but is throttled to its share; oversubscribing that share with one
    thread per logical CPU is an order of magnitude slower than matching it)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(float(parts[0]) / float(parts[1]) + 0.5)))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, int(q / int(f2.read().split()[0]) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def log(msg: str) -> None:
    """Progress on stderr (rank 0): the JSON line stays the only thing on stdout."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(cfg, mini_rows=8, steps=5, warmup=2, device="cpu", rows=None):
    """SURVEY.md §8(d) protocol: the oracle's train step on the host cores at the workload's own micro-batch (the
    largest B the step ever sees), every core this process may use, median of >= 5 steps after 2 warm-ups (fewer, never
    below 3, if the time budget runs out), in bf16 (the regime of the GPU run: `value`) and in fp32 (`fp32`).
    With device="cuda" the same module runs as eager PyTorch-ROCm ops (`eager_gpu_baseline`, informational)."""
    rows = rows or mini_rows
    if device != "cpu":
        v, n = oracle_step_rate(cfg, mini_rows, steps, warmup, device, rows, torch.bfloat16, torch.get_num_threads())
        return {"value": round(v, 1), "unit": "tokens/s", "cores": 0, "kind": "port",
                "sample": f"oracle train step as eager PyTorch-ROCm ops on cuda:0, {rows} rows x {cfg['ctx_len']} tokens in micro-batches of "
                          f"{mini_rows}, bf16, dense additive masks, median of {n} steps after {warmup} warm-up"}
    usable = usable_cores()
    threads = max(1, usable)
    v16, n16 = oracle_step_rate(cfg, mini_rows, steps, warmup, "cpu", rows, torch.bfloat16, threads)
    v32, n32 = oracle_step_rate(cfg, mini_rows, steps, warmup, "cpu", rows, torch.float32, threads, budget_s=30.0)
    vd, nd = oracle_step_rate(cfg, mini_rows, 3, 1, "cpu", rows, torch.bfloat16, threads, budget_s=25.0, dropout=0.1)
    return {"value": round(v16, 1), "unit": "tokens/s", "cores": threads, "kind": "port", "dtype": "bf16",
            "fp32": {"value": round(v32, 1), "unit": "tokens/s", "steps": n32},
            # "dropout 0 and 0.1" — the reference's default regime (train_encoder.py:445): Bernoulli masks from torch's
            # generator at its four sites (23 % of the reference's own CPU step)
            "dropout_0.1": {"value": round(vd, 1), "unit": "tokens/s", "dtype": "bf16", "steps": nd},
            "host": {"logical_cpus": os.cpu_count(), "usable_by_this_process": usable, "torch_threads": threads},
            "sample": f"oracle (oracle/omnibiote_ref.py) train step fwd+masked CE+bwd+clip+AdamW on the host cores, one micro-batch of "
                      f"{mini_rows} rows x {cfg['ctx_len']} tokens per step, dense additive masks, dropout 0; bf16: median of {n16} steps, "
                      f"fp32: median of {n32} steps, each after {warmup} warm-ups"}


def measure_other_config(name: str, rows: int, a, dev, steps: int = 2, warmup: int = 1):
    """One more BASELINE config on this GPU, driver-timed inside the same run (N = 1 only; never the headline): its own model,
    its own tuned plans, `warmup` + `steps` optimizer steps of `rows` rows through the same TrainStep as the headline, then one
    single-stream step under the library's launch profiler for the kernel-family fractions."""
    from omnibiote_amd import _lib, tune
    from omnibiote_amd import train_encoder as TE
    import contextlib
    import io
    cfg = CONFIGS[name]
    T = cfg["ctx_len"]
    t_begin = time.perf_counter()
    saved = (a.rows_per_rank,)
    a.rows_per_rank = rows
    try:
        h = harness_args(cfg, a, 1)
    finally:
        (a.rows_per_rank,) = saved
    torch.manual_seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        m = TE.build_model(h, dev)
    n_params = m.get_num_params()
    tune.tune_model_shapes(a.mini_batch_size * T, cfg["n_embd"], 2 ** 16, device=dev)
    opt, sched = TE.build_optimizer(m, h, 1000)
    ts = TE.TrainStep(m, opt, sched, mini_batch_size=a.mini_batch_size, n_head=cfg["n_head"], lm_head_impl=a.readout,
                      pipeline_streams=a.pipeline_streams, backward_order=a.backward_order, rows_forward=not a.full_last_block)
    rng = np.random.default_rng(4321)
    host = [TE.synthetic_rows(rows, T, 2 ** 16, rng, single_document=True) for _ in range(2)]
    batches = [torch.from_numpy(hb).to(dev) for hb in host]
    set_up_s = time.perf_counter() - t_begin
    loss = None
    for i in range(warmup):
        loss = ts(batches[i % 2], input_ids_host=host[i % 2])["loss"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = ts(batches[i % 2], input_ids_host=host[i % 2])["loss"]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    value = rows * T * steps / el
    fpt = TE.flops_per_token(n_params, cfg["n_layer"], cfg["n_embd"], T)
    fpt_exec = TE.flops_per_token_executed(n_params, cfg["n_layer"], cfg["n_embd"], T, a.readout, not a.full_last_block,
                                           rows_attention=(not a.full_last_block) and not a.dense_mask)
    out = {"workload": f"{READOUT_LEAD['masked_full' if (a.readout == 'masked' and a.full_last_block) else a.readout]}; OmniBioTA {name} "
                       f"({cfg['n_layer']}L/{cfg['n_embd']}d/{cfg['n_head']}h) ctx={T}, {rows} rows = {rows // a.mini_batch_size} micro-batches of "
                       f"{a.mini_batch_size}, dropout {a.dropout:g}, single-document rows, one GPU",
           "value": round(value, 1), "unit": "tokens/s", "steps": steps, "warmup": warmup, "ms_per_step": round(el / steps * 1e3, 2),
           "flops_per_token_executed": fpt_exec, "mfma_fraction_whole_step_executed": round(value * fpt_exec / (PEAK_BF16_TFLOPS * 1e12), 4),
           "reference_formula": {"flops_per_token": fpt, "model_flops_fraction": round(value * fpt / (PEAK_BF16_TFLOPS * 1e12), 4)},
           "final_loss": round(float(loss.item()), 4), "set_up_s": round(set_up_s, 1)}
    if not bool(torch.isfinite(loss).item()):
        raise SystemExit(f"bench.py: other_configs[{name}]: non-finite loss")
    if not a.no_roofline:
        ts.pipeline_streams = 1
        _lib.lib().obte_profile_enable(1)
        ts(batches[0], input_ids_host=host[0])
        torch.cuda.synchronize()
        ms, dims, kind = collect_profile()
        _lib.lib().obte_profile_enable(0)
        if len(ms):
            r = roofline_from_profile(ms, dims, kind, 1)
            fam = {"gemm_family": {k: r[k] for k in ("achieved", "frac", "launches_per_step", "avg_launch_ms", "share_of_profiled_time")}}
            for k in ("attn_fwd", "attn_bwd"):
                if k in r["breakdown"]:
                    fam[k] = r["breakdown"][k]
            fam["gemm_family"]["unit"] = "TFLOP/s"
            out["families_single_stream_profiled_step"] = fam
    del ts, opt, sched, m, batches
    torch.cuda.empty_cache()
    out["wall_s"] = round(time.perf_counter() - t_begin, 1)
    return out


class Watchdog:
    """Fail loudly instead of hanging: if the guarded phase (process-group set-up, the first collective, the timed steps)
    has not finished after ``seconds`` of wall time, say where on stderr and end THIS process with a non-zero code —
    torchrun then stops the other ranks and reports failure (the parent of `self_launch` passes that exit code on).  A rank
    stuck inside a collective cannot be interrupted from Python, hence os._exit."""

    def __init__(self, seconds: float, what: str):
        import threading
        self.what, self.seconds = what, seconds
        self.t = threading.Timer(seconds, self._fire)
        self.t.daemon = True

    def _fire(self):
        print(f"[bench] FATAL rank {os.environ.get('RANK', '0')}: '{self.what}' not finished after {self.seconds:.0f} s — "
              "giving up (non-zero exit) instead of hanging", file=sys.stderr, flush=True)
        os._exit(97)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a CHILD process group (one rank per GPU,
    `python -m torch.distributed.run`, rendezvous on 127.0.0.1) before this process has made any HIP call, pass the
    child's output through (rank 0 prints the JSON line) and return its exit code.  No exec: the parent only waits."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's intra-node transport needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(n_gpus, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.masked_lm_head:
        a.readout = "masked"
    cfg = CONFIGS[a.config]
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or os.environ.get("OBTE_BENCH_FORCE_LAUNCH") == "1"):
        sys.exit(self_launch(a.gpus))   # (OBTE_BENCH_FORCE_LAUNCH=1: rehearse the child launch with one rank on a one-GPU box)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; drop WORLD_SIZE (bench.py then starts its own ranks) or "
                         f"launch with torch.distributed.run --nproc-per-node {a.gpus}")
    # OBTE_BENCH_REHEARSE=1: every rank on cuda:0 over gloo — rehearses the N > 1 control flow (collective order, DDP, the
    # per-rank variants) on a one-GPU box, where RCCL refuses two ranks on one device.  The numbers it prints mean nothing.
    rehearse = os.environ.get("OBTE_BENCH_REHEARSE") == "1" and world > 1
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    init_s = float(os.environ.get("OBTE_BENCH_INIT_TIMEOUT_S", "240"))
    if world > 1:
        import datetime
        n_dev = torch.cuda.device_count()
        if not rehearse and local >= n_dev:
            raise SystemExit(f"bench.py: rank {rank} wants cuda:{local} but this node shows {n_dev} GPU(s)")
        with Watchdog(init_s, f"init_process_group + first all-reduce over {world} ranks"):
            if rehearse:
                dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=init_s))
            else:   # RCCL; device_id binds the communicator to this rank's GPU up front (eager init: no lazy connect inside the timed region)
                dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=max(init_s, 600.0)))
            # first contact: every rank contributes 1 to a device all-reduce; anything but `world` back means the ranks are
            # not the job we think they are (mis-launch, wrong communicator) — stop before timing nonsense
            probe = torch.ones(1, dtype=torch.float32, device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            if int(probe.item()) != world or dist.get_world_size() != world:
                raise SystemExit(f"bench.py: rank {rank}: first all-reduce returned {probe.item()} over a group of "
                                 f"{dist.get_world_size()}, expected {world}")
        log(f"process group up: backend {dist.get_backend()}, world {dist.get_world_size()}, first all-reduce ok")
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
    plans_loaded = bool(a.plan_cache and os.path.exists(a.plan_cache))
    if plans_loaded:
        tune.load_plans(a.plan_cache)
    else:
        # every rank times the candidates on its own GPU (in parallel), then all adopt rank 0's table, so that the N
        # replicas run the same kernels (same arithmetic, same speed)
        tune.tune_model_shapes(max(1, a.micro_batches_per_pass) * a.mini_batch_size * cfg["ctx_len"], cfg["n_embd"], 2 ** 16, device=dev,
                               verbose=(rank == 0 and bool(a.shapes_out)))
        if world > 1:
            box = [tune.export_plans() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            if rank != 0:
                tune.import_plans(box[0])
    force_ddp = os.environ.get("OBTE_FORCE_DDP") == "1"   # rehearse the N>1 code path on one GPU
    if force_ddp and world == 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1)
    model = TE.wrap_ddp(m, local, grad_exchange=a.grad_exchange, timed=True) if (world > 1 or force_ddp) else m
    total_iters = 1000
    opt, sched = TE.build_optimizer(m, h, total_iters)
    step = TE.TrainStep(model, opt, sched, mini_batch_size=a.mini_batch_size, n_head=cfg["n_head"],
                        lm_head_impl=a.readout, pipeline_streams=a.pipeline_streams, micro_batches_per_pass=max(1, a.micro_batches_per_pass), backward_order=a.backward_order, rows_forward=not a.full_last_block,
                        mask_impl="dense" if a.dense_mask else "ranges")
    rng = np.random.default_rng(1234 + rank)
    T = cfg["ctx_len"]
    # synthetic batches resident in HBM before timing; a fresh one per step
    host_batches = [TE.synthetic_rows(a.rows_per_rank, T, 2 ** 16, rng, single_document=not a.multi_document) for _ in range(min(a.steps + a.warmup, 4))]
    batches = [torch.from_numpy(hb).to(dev) for hb in host_batches]
    host_of = {id(b): hb for b, hb in zip(batches, host_batches)}   # the loader's host copy of each resident batch (the MLM mask is
                                                                    # drawn on the host, train_encoder.py:273: no device round trip)
    _step = step   # the TrainStep object: attribute switches below go to it

    def step(batch):   # noqa: F811
        return _step(batch, input_ids_host=host_of.get(id(batch)))

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- micro-batches per pass: an execution option chosen by measurement, like the GEMM plans ---------------------------------
    # k micro-batches of mini_batch_size rows go through the model in ONE pass of k * mini_batch_size rows (TrainStep: the mask
    # builder's per-micro-batch behaviour and the reference's per-micro-batch loss normalisation are kept, so loss and gradients are
    # those of k separate passes — tests/test_hip_headline.py, test_hip_model.py); every kernel sees k times the rows per launch.
    # Which k is fastest depends on the shapes (more tile rounds per launch against fewer passes to pipeline): each candidate is
    # timed over two steps after one warm-up, before the warm-up steps of the contract; all ranks adopt rank 0's choice.
    per_pass_selection = None
    n_accum_rank = a.rows_per_rank // a.mini_batch_size
    if a.micro_batches_per_pass == 0:
        cands = [k for k in (1, 2, 4) if a.readout in ("dense", "masked") and n_accum_rank % k == 0 and n_accum_rank // k >= 2] or [1]
        per_pass_selection = {}
        if len(cands) > 1:
            with Watchdog(600.0, "choosing micro-batches per pass"):
                for k in cands:
                    if not plans_loaded and k > 1:
                        tune.tune_model_shapes(k * a.mini_batch_size * cfg["ctx_len"], cfg["n_embd"], 2 ** 16, device=dev)
                        if world > 1:
                            box = [tune.export_plans() if rank == 0 else None]
                            dist.broadcast_object_list(box, src=0)
                            if rank != 0:
                                tune.import_plans(box[0])
                    _step.per_pass = k
                    step(batches[0])
                    sync()
                    t0 = time.perf_counter()
                    for i in range(2):
                        step(batches[(1 + i) % len(batches)])
                    sync()
                    tk = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
                    if world > 1:
                        dist.all_reduce(tk, op=dist.ReduceOp.MAX)
                    per_pass_selection[k] = round(float(tk.item()) / 2 * 1e3, 3)
            best = min(per_pass_selection, key=per_pass_selection.get)
            if world > 1:
                box = [best if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                best = box[0]
        else:
            best = cands[0]
        a.micro_batches_per_pass = best
        log(f"micro-batches per pass: {best} (ms per step over two steps: {per_pass_selection})")
    _step.per_pass = a.micro_batches_per_pass
    if a.plan_cache and rank == 0 and not plans_loaded:
        tune.save_plans(a.plan_cache)
    log(f"model built, plans ready; timing {a.warmup}+{a.steps} steps")
    losses = []
    step_budget_s = float(os.environ.get("OBTE_BENCH_STEP_TIMEOUT_S", "30")) * (4.0 if rehearse else 1.0)
    with Watchdog(120.0 + step_budget_s * (a.warmup + a.steps), f"{a.warmup}+{a.steps} train steps"):
        for i in range(a.warmup):
            losses.append(step(batches[i % len(batches)])["loss"])
        sync()
        t0 = time.perf_counter()
        for i in range(a.steps):
            losses.append(step(batches[(a.warmup + i) % len(batches)])["loss"])
        sync()
        elapsed = time.perf_counter() - t0
    _lib.check_device_status("timed steps")   # a kernel that found its own results invalid (device status word): a failed run, not a number
    exchange = None
    th = getattr(model, "_obte_timed_hook", None)
    if th is not None:   # per-bucket device events of the gradient exchange; the LAST timed step's buckets are reported
        ex = th.collect(getattr(_step, "backward_end_event", None))
        if ex:
            nb = max(1, len(ex["buckets"]) // (a.warmup + a.steps + (3 * len(per_pass_selection) if per_pass_selection and len(per_pass_selection) > 1 else 0)))
            ex["buckets"] = ex["buckets"][-nb:]
            ex["note"] = ("milliseconds from a bucket's gradients being complete on the backward's stream to its exchange having finished; "
                          "ms_exposed_after_backward = how long the exchange ran past the end of the step's last backward pass")
            exchange = ex
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    tokens_per_step = a.rows_per_rank * T * world   # no PAD in the synthetic rows: all tokens count (train_encoder.py:350)
    value = tokens_per_step * a.steps / elapsed
    fpt = TE.flops_per_token(n_params, cfg["n_layer"], cfg["n_embd"], T)
    # 15 % of the positions are MLM-masked (train_encoder.py:271); their share of the readout products (and of the last block's MLP half) remains
    fpt_exec = TE.flops_per_token_executed(n_params, cfg["n_layer"], cfg["n_embd"], T, a.readout, not a.full_last_block,
                                           rows_attention=(not a.full_last_block) and not a.dense_mask)

    log(f"timed region done: {value:,.0f} tokens/s")
    tail_guard = Watchdog(900.0, "profiled step + variants (they contain collectives at N > 1)")
    tail_guard.__enter__()
    roofline = None
    if not a.no_roofline:
        # per-launch durations are only meaningful when launches do not share the chip: the profiled step runs on one
        # stream (the timed steps above overlap two micro-batches, which stretches every kernel that has company).
        # EVERY rank runs this step (it contains the gradient all-reduce); only rank 0 records and reports.
        _step.pipeline_streams = 1
        if rank == 0:
            _lib.lib().obte_profile_enable(1)
        step(batches[0])
        torch.cuda.synchronize()
        _lib.check_device_status("profiled step")
        ms, dims, kind = collect_profile() if rank == 0 else ([], [], [])
        if rank == 0:
            _lib.lib().obte_profile_enable(0)
        _step.pipeline_streams = a.pipeline_streams
        if rank == 0 and len(ms):
            roofline = roofline_from_profile(ms, dims, kind, 1)
            # HBM bytes per launch of the GEMM family: PMC counters cannot be read from inside this process, so the
            # figure comes from the committed rocprofv3 --pmc passes over this same workload (tools/pmc_traffic.py)
            # (a pass recorded at another number of micro-batches per pass has other launches: only a matching one is quoted)
            def _pmc_matches(q):
                try:
                    with open(q) as f:
                        return json.load(f).get("micro_batches_per_pass", 1) == a.micro_batches_per_pass
                except Exception:
                    return False
            pmc = next((q for q in (os.path.join(ROOT, "profiles", n) for n in ("r05_pmc_gemm_family_traffic.json", "r04_pmc_gemm_family_traffic.json", "r03_pmc_gemm_family_traffic.json", "r02_pmc_gemm_family_traffic.json"))
                        if os.path.exists(q) and _pmc_matches(q)), "")
            default_workload = (a.config == "small" and a.readout == "masked" and a.dropout == 0.0 and not a.multi_document
                                and a.rows_per_rank == 128 and a.mini_batch_size == 8)
            if default_workload and pmc:
                with open(pmc) as f:
                    t = json.load(f)
                roofline["traffic"] = round(t["traffic_bytes_per_launch"])
                roofline["traffic_unit"] = "bytes/launch (2 x FETCH_SIZE + WRITE_SIZE)"
                roofline["traffic_source"] = f"committed PMC pass ({os.path.basename(pmc)}): a static, pre-recorded figure, not measured in this run"
                by_shape = os.path.join(ROOT, "profiles", "r05_pmc_gemm_by_shape.json")
                if os.path.exists(by_shape):   # which launches read more than their operands, and by how much (tools/pmc_gemm_by_shape.py)
                    with open(by_shape) as f:
                        roofline["traffic_by_shape"] = {"source": os.path.basename(by_shape), "read_over_operands": {
                            q["name"]: q["read_ratio"] for q in json.load(f)["shapes"]}}
            if a.shapes_out:
                tab = {}
                for t, d, k in zip(ms, dims, kind):
                    if int(k) % 1000 in HBM_KINDS:
                        continue
                    key = (KIND_NAMES.get(int(k) % 1000, str(int(k))) + " [" + rocprof_name(k) + "]", int(d[0]), int(d[1]), int(d[2]), int(k) % 1000)
                    e = tab.setdefault(key, [0, 0.0])
                    e[0] += 1; e[1] += float(t)
                with open(a.shapes_out, "w") as f:
                    f.write(f"{'launch [rocprof kernel name]':62s} {'d0':>7s} {'d1':>7s} {'d2':>7s} {'calls':>6s} {'avg_us':>9s} {'TFLOP/s':>8s} {'ms/step':>8s}\n")
                    for (name, d0, d1, d2, code), (n, tt) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
                        fl = launch_flops(code, d0, d1, d2)
                        f.write(f"{name:62s} {d0:7d} {d1:7d} {d2:7d} {n:6d} {tt / n * 1e3:9.1f} {fl * n / (tt * 1e-3) / 1e12:8.1f} {tt:8.2f}\n")
    # Not the headline — the same step in the other regimes a user of the reference meets, each timed over 10 steps after
    # 2 warm-ups and reported beside `value`:
    #   dense_logits_forward / dense_dlogits_full_backward   the readout computed for every position in the forward, and also
    #                        with the dense d(logits) backward (the reference's literal graph) — same loss and gradients as the
    #                        headline's masked-rows readout, rows outside the mask contribute exact zeros;
    #   dropout_0.1          the reference's default --dropout (train_encoder.py:445); `value` is quoted at dropout 0,
    #                        the parity regime;
    #   dense_mask_calling_convention   the mask handed over as the reference's additive (B,H,T,T) expand() view
    #                        (train_encoder.py:290-292) instead of key ranges.
    variants = None

    def timed_variant(n=10, w=2, bs=None):
        bs = bs or batches
        for i in range(w):
            step(bs[i % len(bs)])
        sync()
        t0 = time.perf_counter()
        for i in range(n):
            step(bs[i % len(bs)])
        sync()
        el = time.perf_counter() - t0
        _lib.check_device_status("variant")
        if world > 1:
            te = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el = float(te.item())
        return round(tokens_per_step * n / el, 1)

    if not a.no_variants:
        log("variants")
        variants = {}
        if a.readout != "masked":
            _step.lm_head_impl = "masked"
            variants["masked_rows_readout"] = {"value": timed_variant(), "unit": "tokens/s", "steps": 10,
                                               "note": "readout + CE on the MLM-masked rows only, forward included; identical loss and gradients"}
            _step.lm_head_impl = a.readout
        if a.readout == "masked" and _step.rows_forward:
            _step.rows_forward = False
            variants["masked_readout_full_last_block"] = {"value": timed_variant(), "unit": "tokens/s", "steps": 10,
                                                          "note": "the masked-positions readout on model.forward(return_embeddings=True) of every position: the last "
                                                                  "block's MLP half and ln_f computed for all positions, rows gathered afterwards (the headline hands "
                                                                  "the positions to model.forward(rows=...)); identical loss and gradients"}
            _step.rows_forward = True
        if a.readout != "dense":
            _step.lm_head_impl = "dense"
            variants["dense_logits_forward"] = {"value": timed_variant(), "unit": "tokens/s", "steps": 10,
                                                "note": "logits of EVERY position in the forward (model.py:253 as the reference calls it), readout backward over "
                                                        "the masked rows; identical loss and gradients (rounds 1-2 quoted this form)"}
            _step.lm_head_impl = a.readout
        if a.readout != "dense_full":
            _step.lm_head_impl = "dense_full"
            variants["dense_dlogits_full_backward"] = {"value": timed_variant(), "unit": "tokens/s", "steps": 10,
                                                       "note": "the reference's literal graph: dense [M,V] d(logits) and full-size readout backward products "
                                                               "(85 % of those rows are exact zeros); identical loss and gradients"}
            _step.lm_head_impl = a.readout
        if a.readout in ("dense", "masked"):
            # the headline's k was chosen at start-up; here the neighbouring choices over 10 steps each (k = 1 is one pass per
            # micro-batch, the form rounds 1-3 quoted)
            for k_alt in (1, 2, 4):
                if k_alt == a.micro_batches_per_pass or n_accum_rank % k_alt != 0 or n_accum_rank // k_alt < 2:
                    continue
                _step.per_pass = k_alt
                if not plans_loaded:
                    tune.tune_model_shapes(k_alt * a.mini_batch_size * cfg["ctx_len"], cfg["n_embd"], 2 ** 16, device=dev)   # plans for its shapes
                    if world > 1:
                        box = [tune.export_plans() if rank == 0 else None]
                        dist.broadcast_object_list(box, src=0)
                        if rank != 0:
                            tune.import_plans(box[0])
                variants[f"micro_batches_per_pass_{k_alt}"] = {"value": timed_variant(), "unit": "tokens/s", "steps": 10,
                                                               "note": f"{k_alt} micro-batch{'es' if k_alt > 1 else ''} of {a.mini_batch_size} rows per forward/backward pass "
                                                                       f"(the headline runs {a.micro_batches_per_pass}); same loss and gradients (per-micro-batch masks and loss "
                                                                       "normalisation are kept), every kernel sees that many times the rows per launch"}
            _step.per_pass = a.micro_batches_per_pass
        if a.dropout == 0.0:
            TE.set_dropout(m, 0.1)
            variants["dropout_0.1"] = {"value": timed_variant(), "unit": "tokens/s", "steps": 10,
                                       "note": "the reference's default --dropout 0.1 (fused counter-based masks at all four sites)"}
            TE.set_dropout(m, 0.0)
        if not a.multi_document:
            # "the multi-document variant is reported second" — rows that pack several documents (loader.py:118-163), masked
            # block-diagonally (train_encoder.py:25-57, its row >= 1 merge quirk per mini-batch): the attention kernels skip the key
            # tiles no query of a workgroup may see.  The FLOP figure counts only the tiles the kernels visit.
            md_host = [TE.synthetic_rows(a.rows_per_rank, T, 2 ** 16, rng, single_document=False) for _ in range(2)]
            md = [torch.from_numpy(hb).to(dev) for hb in md_host]
            for b_, hb in zip(md, md_host):
                host_of[id(b_)] = hb
            work = attention_work_fraction(md_host[0], a.mini_batch_size)
            v = timed_variant(bs=md)
            fpt_md = TE.flops_per_token_executed(n_params, cfg["n_layer"], cfg["n_embd"], T, a.readout, not a.full_last_block,
                                                 attention_fraction=work["tile_fraction"],
                                                 rows_attention=(not a.full_last_block) and not a.dense_mask)
            variants["multi_document"] = {"value": v, "unit": "tokens/s", "steps": 10,
                                          "attention_work": work, "flops_per_token_executed": fpt_md,
                                          "mfma_fraction_whole_step_executed": round(v * fpt_md / (PEAK_BF16_TFLOPS * 1e12 * world), 4),
                                          "note": "rows packing several documents (80 % nucleotide / 20 % peptide length mix), block-diagonal key ranges with the "
                                                  "reference builder's row >= 1 quirk; executed FLOP = the headline's minus the attention tiles outside every "
                                                  "workgroup's key range (tile_fraction of 12 L C T: forward 256 queries x 64 keys, backward 256 keys x 32 queries)"}
            del md
        if not a.dense_mask:
            _step.mask_impl = "dense"
            variants["dense_mask_calling_convention"] = {"value": timed_variant(), "unit": "tokens/s", "steps": 10,
                                                         "note": "attn_mask passed as the reference's dense additive (B,H,T,T) expand() view"}
            _step.mask_impl = "ranges"
    # the headline's plan table, before other configs add their shapes to the library's
    plans = tune.export_plans()
    other_configs = None
    if world == 1 and not a.no_other_configs and a.config == "small":
        # BASELINE configs 4 and 5 (their single-GPU legs), driver-timed in this same run: never the headline
        other_configs = {}
        for oname, orows in (("small4k", 32), ("large", 32)):
            log(f"other config {oname}")
            try:
                other_configs[oname] = measure_other_config(oname, orows, a, dev)
            except SystemExit:
                raise
            except Exception as e:   # noqa: BLE001 — the headline does not depend on it; say what happened
                other_configs[oname] = {"value": None, "note": repr(e)[:300]}
    if world > 1:
        dist.barrier()
    tail_guard.__exit__()

    for x in losses:    # every rank: a non-finite loss is a failed run, not a number to report
        if not bool(torch.isfinite(x).item()):
            raise SystemExit(f"bench.py: rank {rank}: non-finite loss {float(x.item())}")
    if rank == 0:
        import hashlib
        plan_rows = [f"{'k' if r['a_kmajor'] else 'm'}{'k' if r['b_kmajor'] else 'n'} epi{r['epilogue']} {r['M']}x{r['N']}x{r['K']}: "
                     f"{STRUCT_NAMES.get(r['variant'], r['variant'])} bn{r['bn']} split{r['splits']}" for r in plans]
        plan_hash = hashlib.sha256("\n".join(plan_rows).encode()).hexdigest()[:16]
        if world > 1:
            rccl = None
            if dist.get_backend() == "nccl":
                try:   # informational: nothing about the measurement depends on it
                    rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
                except Exception as e:   # noqa: BLE001
                    rccl = "unknown (" + type(e).__name__ + ")"
            collectives = {"backend": "RCCL (torch.distributed backend nccl)" if dist.get_backend() == "nccl" else dist.get_backend(),
                           "rccl_version": rccl, "world_size": dist.get_world_size(), "ranks_in_first_all_reduce": world,
                           "grad_exchange": a.grad_exchange,
                           "pattern": ("one bucketed gradient all-reduce" if a.grad_exchange == "allreduce" else
                                       "one all_to_all + one all_gather per bucket (direct reduce-scatter + all-gather over all links, fp32 fixed-order reduction)")
                                      + " per optimizer step (DDP no_sync on all but the last micro-batch, 100-MB buckets) + one scalar all-reduce",
                           "exchange_timing_last_step": exchange}
        else:
            collectives = "none (single rank)"
        out = {
            "metric": METRIC if a.config == "small" else f"MLM train tokens/sec, {a.config} ctx={T}", "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic" if not rehearse else "synthetic (REHEARSAL: all ranks on one GPU over gloo; not a measurement)",
            "config": {"workload": f"{READOUT_LEAD['masked_full' if (a.readout == 'masked' and a.full_last_block) else a.readout]}; "
                                   f"OmniBioTA {a.config} ({cfg['n_layer']}L/{cfg['n_embd']}d/{cfg['n_head']}h) ctx={T} MLM train step: "
                                   f"{a.rows_per_rank} rows/rank = {a.rows_per_rank // a.mini_batch_size} micro-batches of {a.mini_batch_size} ({a.micro_batches_per_pass} per pass), "
                                   f"{READOUT_TEXT[a.readout]}, dropout {a.dropout:g}, {'multi' if a.multi_document else 'single'}-document rows",
                       "global_batch_rows": a.rows_per_rank * world, "mini_batch_size": a.mini_batch_size, "seq_len": T,
                       "micro_batches_per_pass": a.micro_batches_per_pass,
                       "micro_batches_per_pass_selection_ms_per_step": per_pass_selection,
                       "parallelism": f"dp{world}", "dropout": a.dropout, "vocab": 65536,
                       "collectives": collectives},
            # FLOP the step actually executes per token: the reference's 6N + 12LCT (train_encoder.py:360) minus the part of the
            # readout the default path does not perform (its three products run over the ~15 % MLM-masked rows only; the other
            # rows are multiplied by zero in the loss).  This is the fraction of the bf16 MFMA peak the whole step sustains.
            "flops_per_token_executed": fpt_exec,
            "mfma_fraction_whole_step_executed": round(value * fpt_exec / (PEAK_BF16_TFLOPS * 1e12 * world), 4),
            # the reference's own formula, for comparison with its MFU log line only: it counts FLOP this step does not execute
            "reference_formula": {"flops_per_token": fpt, "model_flops_fraction": round(value * fpt / (PEAK_BF16_TFLOPS * 1e12 * world), 4)},
            "final_loss": round(float(losses[-1].item()), 4),
            # which kernel structure / tile width / split-K each GEMM shape ran with (tuned at start-up or loaded from
            # --plan_cache): lets a rocprof summary under profiles/ be matched to this run
            "gemm_plans": {"sha16": plan_hash, "source": ("cache " + a.plan_cache) if (a.plan_cache and plans_loaded) else "tuned at start-up", "table": plan_rows},
            "roofline": roofline,
        }
        if variants:
            out["variants"] = variants
        if other_configs:
            out["other_configs"] = other_configs
        if world == 1 and not a.no_cpu_baseline:
            log(f"cpu_baseline on {usable_cores()} cores")
            out["cpu_baseline"] = cpu_baseline(cfg)
            log("eager_gpu_baseline")
            try:   # same oracle step as eager torch ops on this GPU (informational; never the product path)
                del step, _step, opt, model, m
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
