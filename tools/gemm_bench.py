"""Micro-benchmark of the GEMM entry point on the hot-path shapes (random bf16 data).  Usage:
    python tools/gemm_bench.py [--reps 20] [--shapes small|all] [--only NAME]
Prints TFLOP/s per shape (median of reps, HIP events on the launch stream)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnibiote_amd import _lib as L, ops  # noqa: E402

M = int(os.environ.get("OBTE_BENCH_M", "8192"))   # rows per launch (8192 = one micro-batch of 8 x 1024 tokens)
C, V = 1024, 65536
SHAPES = {
    # name: (kind, M, N, K)
    "fwd_qkv": ("nt", M, 3 * C, C), "fwd_proj": ("nt_add", M, C, C), "fwd_fc": ("nt_gelu", M, 4 * C, C),
    "fwd_mlp": ("nt_add", M, C, 4 * C), "fwd_lm": ("nt", M, V, C),
    "dg_mlp": ("nn_gelubwd", M, 4 * C, C), "dg_fc": ("nn", M, C, 4 * C), "dg_proj": ("nn", M, C, C),
    "dg_qkv": ("nn", M, C, 3 * C), "dg_lm": ("nn", M, C, V),
    "wg_mlp": ("tn", C, 4 * C, M), "wg_fc": ("tn", 4 * C, C, M), "wg_proj": ("tn", C, C, M), "wg_qkv": ("tn", 3 * C, C, M),
    "wg_lm": ("tn", V, C, M),
}


def run(name, reps):
    kind, m, n, k = SHAPES[name]
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn(m * k, device=dev, generator=g).to(torch.bfloat16)
    b = torch.randn(n * k, device=dev, generator=g).to(torch.bfloat16)
    aux = torch.randn(m * n, device=dev, generator=g).to(torch.bfloat16) if ("add" in kind or "gelubwd" in kind) else None
    ak, bk = kind.startswith("nt") or kind.startswith("nn"), kind.startswith("nt")
    epi = {"nt": L.EPI_NONE, "nn": L.EPI_NONE, "tn": L.EPI_NONE, "nt_add": L.EPI_ADD, "nt_gelu": L.EPI_GELU, "nn_gelubwd": L.EPI_GELU_BWD}[kind]
    out = torch.empty(m * n, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, b, m, n, k, ak, bk, epi, aux, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm(a, b, m, n, k, ak, bk, epi, aux, out=out)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    med = ts[len(ts) // 2]
    print(f"{name:9s} {kind:11s} M={m:6d} N={n:6d} K={k:6d}  {med * 1e3:9.1f} us  {2.0 * m * n * k / (med * 1e-3) / 1e12:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    for name in SHAPES:
        if a.only and name not in a.only.split(","):
            continue
        run(name, a.reps)
