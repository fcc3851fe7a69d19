"""Where the one-kernel attention backward spends its slice loop: the debug library (make -C omnibiote_amd/csrc debug) run with parts of
the loop compiled out by OBTE_ATTN_SKIP bits (results are wrong, timing only).  Interleaved rounds in one process, median per mask.
    OBTE_LIB_PATH=omnibiote_amd/libomnibiote_hip_debug.so python tools/attn_fused_skip.py [--T 1024] [--masks 0,1,2,...]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnibiote_amd import ops, masks

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=1024)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--masks", default="0,1,2,4,8,16,32,64,128,3,35,39,103,255")
a = ap.parse_args()
B, H, T, hs = 8, 8, a.T, 128
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B, T, 3 * H * hs, device=dev, generator=g).to(torch.bfloat16)
d_o = torch.randn(B, T, H * hs, device=dev, generator=g).to(torch.bfloat16)
tok = torch.randint(20, 100, (B, T), device=dev)
spec = ops.MaskSpec(ranges=masks.RangeMask.from_tokens(tok).key_ranges)
scale = 8.0 / (H * hs)
os.environ["OBTE_ATTN_SKIP"] = "0"
o, lse = ops.attn_fwd(qkv, B, T, H, hs, scale, spec)
ms = [int(x) for x in a.masks.split(",")]
res = {m: [] for m in ms}
for r in range(a.rounds + 1):
    for m in ms:
        os.environ["OBTE_ATTN_SKIP"] = str(m)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ops.attn_bwd(qkv, o, d_o, lse, B, T, H, hs, scale, spec)
        e1.record(); e1.synchronize()
        if r:
            res[m].append(e0.elapsed_time(e1) / 3 * 1e3)
NAMES = {1: "D reads", 2: "A reads", 4: "softmax", 8: "barrier", 16: "dma wait", 32: "C reads", 64: "dS image + dQ stores", 128: "DMA issue"}
base = sorted(res[0])[len(res[0]) // 2] if 0 in res else None
for m in ms:
    v = sorted(res[m]); med = v[len(v) // 2]
    what = " + ".join(n for b, n in NAMES.items() if m & b) or "nothing removed"
    print(f"skip {m:4d}: {med:8.1f} us (min {v[0]:7.1f})  {'' if base is None else f'{med - base:+7.1f}'}   without: {what}", flush=True)
