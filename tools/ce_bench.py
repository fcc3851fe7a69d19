"""masked_ce_rows micro-benchmark at the hot-path shape (1229 masked rows of 8192, V = 65536), logits flushed from the caches
between repetitions.   python tools/ce_bench.py [--rows 1229] [--reps 15]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnibiote_amd import ops
ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, default=1229); ap.add_argument("--reps", type=int, default=15)
a = ap.parse_args()
M, V = 8192, 65536
g = torch.Generator(device="cuda").manual_seed(0)
logits = (torch.randn(M, V, device="cuda", generator=g) * 1.5).to(torch.bfloat16)
tgt = torch.randint(0, V, (M,), device="cuda")
rows = torch.sort(torch.randperm(M, device="cuda")[:a.rows]).values
big = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
ts = []
for _ in range(a.reps):
    big.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.masked_ce_rows(logits, tgt, rows, 16); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort(); t = ts[len(ts) // 2] * 1e3
print(f"masked_ce_rows {a.rows} rows: {t:7.1f} us  {4.0 * a.rows * V / t / 1e6:5.2f} TB/s (algorithmic read + write; includes the torch glue of the wrapper)")
