"""How does this repo's GEMM compare with the vendor library (torch.mm -> hipBLASLt/rocBLAS) on the hot-path shapes?
Reference point only: the product path never calls the vendor library.  python tools/blas_compare.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnibiote_amd import ops, tune, _lib as L
M, C, V = 8192, 1024, 65536
dev = "cuda"
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2] * 1e3
g = torch.Generator(device=dev).manual_seed(0)
def rnd(*s): return torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
cases = [("fwd_lm  x[M,C] @ W[V,C]^T", (M, V, C), "nt"), ("fwd_fc  x[M,C] @ W[4C,C]^T", (M, 4 * C, C), "nt"), ("fwd_qkv", (M, 3 * C, C), "nt"),
         ("fwd_mlp x[M,4C] @ W[C,4C]^T", (M, C, 4 * C), "nt"), ("dg_lm   dy[M,V] @ W[V,C]", (M, C, V), "nn"), ("dg_fc   dy[M,4C] @ W[4C,C]", (M, C, 4 * C), "nn"),
         ("wg_lm   dy[M,V]^T @ x[M,C]", (V, C, M), "tn"), ("wg_fc   dy[M,4C]^T @ x[M,C]", (4 * C, C, M), "tn")]
for name, (m, n, k), kind in cases:
    if kind == "nt":
        a, b = rnd(m, k), rnd(n, k)
        t_blas = timeit(lambda: torch.mm(a, b.t()))
        tune.tune_gemm(m, n, k, True, True, L.EPI_NONE)
        t_own = timeit(lambda: ops.linear_fwd(a, b))
    elif kind == "nn":
        a, b = rnd(m, k), rnd(k, n)
        t_blas = timeit(lambda: torch.mm(a, b))
        tune.tune_gemm(m, n, k, True, False, L.EPI_NONE)
        t_own = timeit(lambda: ops.linear_dgrad(a, b))
    else:
        a, b = rnd(k, m), rnd(k, n)
        t_blas = timeit(lambda: torch.mm(a.t(), b))
        tune.tune_gemm(m, n, k, False, False, L.EPI_NONE)
        t_own = timeit(lambda: ops.linear_wgrad(a, b))
    fl = 2.0 * m * n * k
    print(f"{name:30s} vendor {t_blas:8.1f} us {fl / t_blas / 1e6:7.1f} TF | this repo {t_own:8.1f} us {fl / t_own / 1e6:7.1f} TF", flush=True)
