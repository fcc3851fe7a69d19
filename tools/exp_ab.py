import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnibiote_amd import _lib as L, ops
lib = L.lib()
dev='cuda'
def t(M,N,K,ak,bk,epi,variant,bn,splits,reps=9):
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
print("lib", os.environ.get("OBTE_LIB_PATH", "default"))
C=1024; M=8192
cases=[("dg_mlp+gelubwd",(M,4*C,C,True,False,L.EPI_GELU_BWD)),("fwd_fc+gelu",(M,4*C,C,True,True,L.EPI_GELU)),("fwd_qkv",(M,3*C,C,True,True,L.EPI_NONE)),
       ("dg_fc",(M,C,4*C,True,False,L.EPI_NONE)),("fwd_mlp+add",(M,C,4*C,True,True,L.EPI_ADD)),("fwd_lm",(M,65536,C,True,True,L.EPI_NONE))]
for name,(m,n,k,ak,bk,epi) in cases:
    row=[]
    for (v,bn) in [(2,256),(3,256),(2,128)]:
        row.append(f"v{v}/{bn}={t(m,n,k,ak,bk,epi,v,bn,1):7.1f}")
    print(f"{name:16s} "+"  ".join(row), flush=True)
