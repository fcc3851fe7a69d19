import sys; sys.path.insert(0, '/root/repo')
from omnibiote_amd import tune, _lib as L
for (M,N,K,epi) in [(8192,3072,1024,L.EPI_ROPE_QK),(8192,3072,1024,L.EPI_NONE),(8192,4096,1024,L.EPI_GELU),(8192,6144,2048,L.EPI_ROPE_QK),(32768,3072,1024,L.EPI_ROPE_QK)]:
    tune.tune_gemm(M,N,K,True,True,epi,verbose=True)
