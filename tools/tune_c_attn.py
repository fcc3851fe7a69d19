"""Verbose tuning of a few block GEMM shapes (which structure wins and by how much).   python tools/tune_c_attn.py [rows]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnibiote_amd import tune, _lib as L
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
for (N, K, epi) in [(3072, 1024, L.EPI_ROPE_QK), (3072, 1024, L.EPI_NONE), (4096, 1024, L.EPI_GELU), (1024, 1024, L.EPI_ADD), (1024, 4096, L.EPI_ADD)]:
    tune.tune_gemm(M, N, K, True, True, epi, verbose=True)
