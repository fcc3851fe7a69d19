"""Attention micro-benchmark with attention-probability dropout on (the reference's default regime, p = 0.1)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnibiote_amd import ops, masks
B, H, T, hs, p = int(os.environ.get('OBTE_BENCH_B', '8')), 8, 1024, 128, 0.1
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B, T, 3 * H * hs, device=dev, generator=g).to(torch.bfloat16)
d_o = torch.randn(B, T, H * hs, device=dev, generator=g).to(torch.bfloat16)
tok = torch.randint(20, 100, (B, T), device=dev)
spec = ops.MaskSpec(ranges=masks.RangeMask.from_tokens(tok).key_ranges)
scale = 8.0 / (H * hs)
def timeit(fn, reps=20):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(); fn(); e1.record(); fn(); e2.record(); e2.synchronize(); ts.append(e1.elapsed_time(e2))
    ts.sort(); return ts[len(ts) // 2] * 1e3
for pp in (0.0, p):
    o, lse = ops.attn_fwd(qkv, B, T, H, hs, scale, spec, pp, 1234)
    tf = timeit(lambda: ops.attn_fwd(qkv, B, T, H, hs, scale, spec, pp, 1234))
    tb = timeit(lambda: ops.attn_bwd(qkv, o, d_o, lse, B, T, H, hs, scale, spec, dropout_p=pp, dropout_seed=1234))
    print(f"dropout {pp:g}: attn fwd {tf:7.1f} us | bwd {tb:7.1f} us", flush=True)
    if pp > 0:   # the forward leaving its keep bits for the key-major backward kernel
        o2, lse2, bits = ops.attn_fwd(qkv, B, T, H, hs, scale, spec, pp, 1234, keep_bits=True)
        tf2 = timeit(lambda: ops.attn_fwd(qkv, B, T, H, hs, scale, spec, pp, 1234, keep_bits=True))
        tb2 = timeit(lambda: ops.attn_bwd(qkv, o, d_o, lse, B, T, H, hs, scale, spec, dropout_p=pp, dropout_seed=1234, drop_bits=bits))
        tb3 = timeit(lambda: ops.attn_bwd(qkv, o, d_o, lse, B, T, H, hs, scale, spec, dropout_p=pp, dropout_seed=1234, drop_bits=bits, one_kernel=False))
        print(f"dropout {pp:g} with keep bits: attn fwd {tf2:7.1f} us | bwd one kernel {tb2:7.1f} us | bwd kernel pair {tb3:7.1f} us", flush=True)
