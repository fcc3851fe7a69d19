"""LayerNorm micro-benchmark at the hot-path shape (8192 rows x 1024 columns; --cols 2048 for the large config).
    python tools/ln_bench.py [--rows 8192] [--cols 1024] [--reps 30]       (OBTE_LIB_PATH=... for an A/B build)"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnibiote_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=8192); ap.add_argument("--cols", type=int, default=1024); ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--warm", action="store_true", help="no flush between repetitions: operands as the previous launch left them in L2 / the Infinity Cache")
a = ap.parse_args()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(a.rows, a.cols, device=dev, generator=g).to(torch.bfloat16)
dy = torch.randn(a.rows, a.cols, device=dev, generator=g).to(torch.bfloat16)
dr = torch.randn(a.rows, a.cols, device=dev, generator=g).to(torch.bfloat16)
w = torch.ones(a.cols, device=dev, dtype=torch.bfloat16)
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)   # flushed between repetitions: operands come from HBM, as in the step


def timeit(fn):
    ts = []
    for _ in range(a.reps):
        if not a.warm:
            big.zero_()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(); fn(); e1.record(); fn(); e2.record(); e2.synchronize()
        ts.append(e1.elapsed_time(e2))     # the second of two back-to-back calls: no launch latency in it
    ts.sort()
    return ts[len(ts) // 2] * 1e3


y, mean, rstd = ops.layernorm_fwd(x, w)
tf = timeit(lambda: ops.layernorm_fwd(x, w))
tb = timeit(lambda: ops.layernorm_bwd(dy, x, w, mean, rstd, dresid=dr))
by = a.rows * a.cols * 2
print(("warm " if a.warm else "cold ") + f"ln fwd {tf:7.1f} us {2 * by / tf / 1e6:6.2f} TB/s | ln bwd(+resid, +dw reduce) {tb:7.1f} us {4 * by / tb / 1e6:6.2f} TB/s", flush=True)
