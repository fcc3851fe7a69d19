"""Per-workgroup phase stamps of one GEMM shape under structures 3 and 7 (debug library: make -C omnibiote_amd/csrc debug).
    OBTE_LIB_PATH=omnibiote_amd/libomnibiote_hip_debug.so OBTE_GEMM_TIMES=1 python tools/gemm_stamps.py M N K [epi]
epi: 0 none, 1 gelu, 2 add, 3 gelu_bwd(NN), 5 rope, 8 = plain dy W (NN).  The library prints one line per launch on stderr."""
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
rope = None
if epi == 5:
    tab = torch.randn(1024, 64, device=dev, generator=g)
    rope = (torch.cos(tab), torch.sin(tab), 1024, 128)
lib = L.lib()
for variant in (3, 7):
    if lib.obte_gemm_plan_set(int(ak), int(bk), epi, M, N, K, variant, 256, 1) != 0:
        continue
    for _ in range(3):
        ops.gemm(a, b, M, N, K, ak, bk, epi, aux, out=out, rope=rope)
    torch.cuda.synchronize()
