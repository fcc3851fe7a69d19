"""One GEMM shape, every plan candidate, operands from HBM (512-MiB flush between repetitions, as the tuner times them):
    python tools/gemm_cold.py fwd_fc|dg_mlp|fwd_qkv|fwd_proj|dg_proj   [--reps 7]      (OBTE_GEMM_NT=1 etc. via the environment)"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnibiote_amd import _lib as L, ops, tune

M, C = 8192, 1024
SHAPES = {"fwd_lm": (M, 65536, C, True, True, L.EPI_NONE), "fwd_fc": (M, 4 * C, C, True, True, L.EPI_GELU), "dg_mlp": (M, 4 * C, C, True, False, L.EPI_GELU_BWD),
          "fwd_qkv": (M, 3 * C, C, True, True, L.EPI_NONE), "fwd_qkv_rope": (M, 3 * C, C, True, True, L.EPI_ROPE_QK), "fwd_proj": (M, C, C, True, True, L.EPI_ADD), "dg_proj": (M, C, C, True, False, L.EPI_NONE),
          "fwd_mlp": (M, C, 4 * C, True, True, L.EPI_ADD), "dg_fc": (M, C, 4 * C, True, False, L.EPI_NONE)}
ap = argparse.ArgumentParser(); ap.add_argument("names", nargs="+"); ap.add_argument("--reps", type=int, default=7)
a = ap.parse_args()
lib = L.lib()
for name in a.names:
    m, n, k, ak, bk, epi = SHAPES[name]
    g = torch.Generator(device="cuda").manual_seed(0)
    A = torch.randn(m * k, device="cuda", generator=g).to(torch.bfloat16)
    B = torch.randn(n * k, device="cuda", generator=g).to(torch.bfloat16)
    aux = torch.randn(m * n, device="cuda", generator=g).to(torch.bfloat16) if epi in (L.EPI_ADD, L.EPI_GELU_BWD) else None
    out = torch.empty(m * n, device="cuda", dtype=torch.bfloat16)
    rope = None
    if epi == L.EPI_ROPE_QK:
        tab = torch.randn(1024, 64, device="cuda", generator=g)
        rope = (torch.cos(tab), torch.sin(tab), 1024, 128)
    row = []
    for (v, bn, sp) in [(1, 128, 1), (2, 128, 1), (2, 256, 1), (3, 256, 1), (4, 128, 1)]:
        L.check(lib.obte_gemm_plan_set(int(ak), int(bk), epi, m, n, k, v, bn, sp), "plan")
        t = tune._time_once(A, B, m, n, k, ak, bk, epi, aux, out, reps=a.reps, rope=rope)
        row.append(f"v{v}/{bn}: {t * 1e3:6.1f} us {2.0 * m * n * k / (t * 1e-3) / 1e12:6.0f} TF")
    print(f"{name:9s} [{os.environ.get('OBTE_GEMM_NT', '0')}] " + " | ".join(row), flush=True)
