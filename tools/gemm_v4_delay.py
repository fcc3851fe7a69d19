"""Timing experiment (debug library): the v4 structure with every second workgroup of a CU started late.
    OBTE_LIB_PATH=.../libomnibiote_hip_debug.so OBTE_GEMM_V4_DELAY=n python tools/gemm_v4_delay.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnibiote_amd import _lib as L, tune
M, C = 8192, 1024
lib = L.lib()
for name, (m, n, k, ak, bk, epi) in {"fwd_fc": (M, 4 * C, C, True, True, L.EPI_GELU), "dg_mlp": (M, 4 * C, C, True, False, L.EPI_GELU_BWD),
                                      "fwd_qkv": (M, 3 * C, C, True, True, L.EPI_NONE)}.items():
    g = torch.Generator(device="cuda").manual_seed(0)
    A = torch.randn(m * k, device="cuda", generator=g).to(torch.bfloat16)
    B = torch.randn(n * k, device="cuda", generator=g).to(torch.bfloat16)
    aux = torch.randn(m * n, device="cuda", generator=g).to(torch.bfloat16) if epi in (L.EPI_ADD, L.EPI_GELU_BWD) else None
    out = torch.empty(m * n, device="cuda", dtype=torch.bfloat16)
    L.check(lib.obte_gemm_plan_set(int(ak), int(bk), epi, m, n, k, 4, 128, 1), "plan")
    t = tune._time_once(A, B, m, n, k, ak, bk, epi, aux, out, reps=7)
    print(f"delay {os.environ.get('OBTE_GEMM_V4_DELAY', '0')}: {name} v4/128 {t * 1e3:6.1f} us", flush=True)
