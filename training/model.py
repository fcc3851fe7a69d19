"""Import shim: ``from model import OmniBioTA, OmniBioTAConfig`` (train_encoder.py:6 and the evals'
``sys.path.insert(0, '../training')``) resolves to the MI355X implementation in ``omnibiote_amd.model``."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnibiote_amd.model import (  # noqa: F401,E402
    Block, LayerNorm, MLP, MuReadout, OmniBioTA, OmniBioTAConfig, SelfAttention, apply_rotary_emb, fused_gelu,
    precompute_freqs_cis, reshape_for_broadcast,
)
