"""Attention micro-benchmark on the hot-path shape (B=8, H=8, T=1024, hs=128, single-document key ranges).
    python tools/attn_bench.py [--reps 10] [--T 1024] [--hs 128]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnibiote_amd import ops, masks

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--B", type=int, default=8); ap.add_argument("--H", type=int, default=8)
ap.add_argument("--T", type=int, default=1024); ap.add_argument("--hs", type=int, default=128)
ap.add_argument("--dense", action="store_true", help="pass the mask as the reference's dense additive (B,H,T,T) expand() view")
ap.add_argument("--nomask", action="store_true", help="no mask at all (rows without EOS attend everywhere)")
ap.add_argument("--two_kernel", action="store_true", help="backward as the dQ + dK/dV kernel pair (default: the one-kernel form where it applies)")
ap.add_argument("--multi", action="store_true", help="multi-document rows (block-diagonal mask) instead of one document per row")
a = ap.parse_args()
B, H, T, hs = a.B, a.H, a.T, a.hs
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B, T, 3 * H * hs, device=dev, generator=g).to(torch.bfloat16)
d_o = torch.randn(B, T, H * hs, device=dev, generator=g).to(torch.bfloat16)
tok = torch.randint(20, 100, (B, T), device=dev)
if a.multi:
    for b in range(B):
        tok[b, torch.randint(8, T - 8, (3,))] = 3
rm = masks.RangeMask.from_tokens(tok)
spec = ops.MaskSpec(ranges=rm.key_ranges)
if a.nomask:
    spec = None
if a.dense:
    spec = ops.MaskSpec.from_user(rm.dense(torch.bfloat16).unsqueeze(1).expand(-1, H, -1, -1), B, T, H, dev)
scale = 8.0 / (H * hs)
def timeit(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2] * 1e3
o, lse = ops.attn_fwd(qkv, B, T, H, hs, scale, spec)
tf = timeit(lambda: ops.attn_fwd(qkv, B, T, H, hs, scale, spec))
tb = timeit(lambda: ops.attn_bwd(qkv, o, d_o, lse, B, T, H, hs, scale, spec, one_kernel=not a.two_kernel))
fl = 4.0 * B * H * T * T * hs
print(f"attn fwd {tf:8.1f} us {fl / tf / 1e6:7.1f} TFLOP/s | bwd {tb:8.1f} us {2.5 * fl / tb / 1e6:7.1f} TFLOP/s (algorithmic)", flush=True)
