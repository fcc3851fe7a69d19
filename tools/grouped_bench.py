"""The block backward's grouped launch (four weight gradients, K = tokens, + the c_attn input gradient) with and without the
evenly divided split form (structure 8): python tools/grouped_bench.py [tokens] [C]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnibiote_amd import ops
K = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
shapes = [(4 * C, C), (C, 4 * C), (3 * C, C), (C, C)]
data = [((torch.randn(K, M, device=dev, generator=g) * 0.5).to(torch.bfloat16), (torch.randn(K, N, device=dev, generator=g) * 0.5).to(torch.bfloat16),
         torch.zeros(M, N, device=dev, dtype=torch.bfloat16)) for M, N in shapes]
dy5 = (torch.randn(K, 3 * C, device=dev, generator=g) * 0.5).to(torch.bfloat16)
w5 = (torch.randn(3 * C, C, device=dev, generator=g) * 0.2).to(torch.bfloat16)
out5 = torch.empty(K, C, device=dev, dtype=torch.bfloat16)
flop = sum(2.0 * M * N * K for M, N in shapes) + 2.0 * K * C * 3 * C
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for acc in (False, True):
    probs = [dict(a=a, b=b, M=M, N=N, K=K, out=o, accumulate=acc) for (M, N), (a, b, o) in zip(shapes, data)]
    probs.append(dict(a=dy5, b=w5, M=K, N=C, K=3 * C, out=out5, a_kmajor=True))
    for split in (False, True):
        res = {}
        for mode in ("warm", "cold"):
            ts = []
            for _ in range(8):
                if mode == "cold":
                    big.zero_()
                else:
                    ops.gemm_grouped(probs, split=split)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.gemm_grouped(probs, split=split); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort(); res[mode] = ts[len(ts) // 2]
        print(f"tokens {K} C {C} accumulate {acc} split {split}: warm {res['warm']:8.1f} us ({flop / res['warm'] / 1e6:6.0f} TF)  cold {res['cold']:8.1f} us", flush=True)
