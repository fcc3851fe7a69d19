import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnibiote_amd import _lib as L, ops
lib = L.lib()
dev='cuda'
def t(M,N,K,ak,bk,epi,variant,bn,splits,reps=7):
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn(M*K, device=dev, generator=g).to(torch.bfloat16)
    b = torch.randn(N*K, device=dev, generator=g).to(torch.bfloat16)
    aux = torch.randn(M*N, device=dev, generator=g).to(torch.bfloat16) if epi in (L.EPI_ADD, L.EPI_GELU_BWD) else None
    out = torch.empty(M*N, device=dev, dtype=torch.bfloat16)
    L.check(lib.obte_gemm_plan_set(int(ak),int(bk),epi,M,N,K,variant,bn,splits),"plan")
    for _ in range(2): ops.gemm(a,b,M,N,K,ak,bk,epi,aux,out=out)
    torch.cuda.synchronize()
    ts=[]
    for _ in range(reps):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm(a,b,M,N,K,ak,bk,epi,aux,out=out); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts)//2]*1e3
print("mode", os.environ.get("OBTE_GEMM_DEBUG", "normal"))
for (M,N) in [(8192,4096)]:
    for K in (64, 256,1024,4096):
        row=[]
        for (variant,bn) in [(2,256),(1,128)]:
            for epi in (L.EPI_NONE, L.EPI_GELU):
                us=t(M,N,K,True,True,epi,variant,bn,1)
                row.append(f"v{variant}/{bn}/{'gelu' if epi else 'none'}={us:6.1f}")
        print(f"M={M} N={N} K={K}: "+"  ".join(row), flush=True)
