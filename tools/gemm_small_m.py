"""The few-tile products of the rows form (the last block's MLP half on ~1229 positions) under each GEMM structure.
    OBTE_GEMM=v1|v2|v3|v4 [OBTE_GEMM_BN=128|256] python tools/gemm_small_m.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnibiote_amd import _lib as L, ops
M, C = 1232, 1024
SH = [("fc+gelu", M, 4 * C, C, True, True, L.EPI_GELU), ("mlp+res", M, C, 4 * C, True, True, L.EPI_ADD), ("dg+gelu'", M, 4 * C, C, True, False, L.EPI_GELU_BWD),
      ("dg fc", M, C, 4 * C, True, False, L.EPI_NONE), ("wg mlp", C, 4 * C, M, False, False, L.EPI_NONE), ("wg fc", 4 * C, C, M, False, False, L.EPI_NONE)]
g = torch.Generator(device="cuda").manual_seed(0)
for name, m, n, k, ak, bk, epi in SH:
    a = torch.randn(m * k, device="cuda", generator=g).to(torch.bfloat16); b = torch.randn(n * k, device="cuda", generator=g).to(torch.bfloat16)
    aux = torch.randn(m * n, device="cuda", generator=g).to(torch.bfloat16) if epi in (L.EPI_ADD, L.EPI_GELU_BWD) else None
    out = torch.empty(m * n, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): ops.gemm(a, b, m, n, k, ak, bk, epi, aux, out=out)
    ts = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm(a, b, m, n, k, ak, bk, epi, aux, out=out); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); t = ts[len(ts) // 2]
    print(f"{name:9s} {m:5d}x{n:5d}x{k:5d}  {t * 1e3:7.1f} us  {2.0 * m * n * k / (t * 1e-3) / 1e12:6.1f} TFLOP/s", flush=True)
