"""One GEMM shape under every structure the plan table admits, warm (back-to-back launches) and cold (512-MB flush between launches).
    python tools/gemm_variants.py M N K [epi]      epi: 0 none, 1 gelu, 2 add, 3 gelu_bwd(NN), 5 rope, 8 = plain dy W (NN)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnibiote_amd import ops, _lib as L
M, N, K = (int(x) for x in sys.argv[1:4])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
nn = epi in (3, 8)
epi = 0 if epi == 8 else epi
ak, bk = True, not nn
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randn(M * K, device=dev, generator=g).to(torch.bfloat16)
b = torch.randn(N * K, device=dev, generator=g).to(torch.bfloat16)
aux = torch.randn(M * N, device=dev, generator=g).to(torch.bfloat16) if epi in (2, 3) else None
out = torch.empty(M * N, device=dev, dtype=torch.bfloat16)
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
lib = L.lib()
rope = None
if epi == 5:
    tab = torch.randn(1024, 64, device=dev, generator=g)
    rope = (torch.cos(tab), torch.sin(tab), 1024, 128)
for variant, bn in ((2, 128), (4, 128), (2, 256), (3, 256), (7, 256)):
    if lib.obte_gemm_plan_set(int(ak), int(bk), epi, M, N, K, variant, bn, 1) != 0:
        continue
    res = {}
    for mode in ("warm", "cold"):
        ts = []
        for _ in range(12):
            if mode == "cold":
                big.zero_()
            else:
                ops.gemm(a, b, M, N, K, ak, bk, epi, aux, out=out, rope=rope)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gemm(a, b, M, N, K, ak, bk, epi, aux, out=out, rope=rope); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort(); res[mode] = ts[len(ts) // 2]
    print(f"structure {variant} bn {bn}: warm {res['warm']:7.1f} us ({2.0 * M * N * K / res['warm'] / 1e6:6.0f} TF)   cold {res['cold']:7.1f} us", flush=True)
