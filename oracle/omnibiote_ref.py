"""CPU oracle for the OmniBioTE encoder hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-torch, CPU-only restatement of the arithmetic the reference performs on its
encoder-training hot path.  It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  Nothing under ``omnibiote_amd/`` imports it, and the
product path fails loudly when the HIP library is missing rather than routing through here.

Parity status
-------------
* Pinned: every function below is checked against golden vectors produced by importing the reference's
  ``training/model.py`` and ``training/train_encoder.py`` in the build container
  (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).
* **parity unpinned** for the three ``mup==1.0.0`` touch points (``MuReadout``, ``set_base_shapes``,
  ``MuAdamW``): the package is pinned by the reference's README (README.md:16) but is neither vendored
  under /root/reference nor installable here, and the reference holds no test for it.  The readout is
  restated from its published algorithm, ``Linear(output_mult * x / width_mult)`` with
  ``width_mult = n_embd / base_n_embd`` (base 24, train_encoder.py:158), and every fixture passes
  ``width_mult`` explicitly.

Each function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

EOS_TOKEN = 3   # training/loader.py:4
MASK_TOKEN = 2  # training/loader.py:5, training/train_encoder.py:20
PAD_TOKEN = 1   # training/loader.py:6
MASKED_VALUE = -1e9  # training/train_encoder.py:40,290
GELU_DIVISOR = 1.41421  # training/model.py:25 (not sqrt(2))
LN_EPS = 1e-5  # training/model.py:72
MUP_BASE_WIDTH = 24  # training/train_encoder.py:158


# --------------------------------------------------------------------------------------------------
# elementary ops
# --------------------------------------------------------------------------------------------------
def layer_norm(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """Weight-only LayerNorm over the last dim, eps 1e-5 (training/model.py:63-72)."""
    return F.layer_norm(x, (x.shape[-1],), weight, None, LN_EPS)


def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    """erf-GELU with the reference's truncated constant (training/model.py:23-25).

    Op order kept (x*0.5, then times (1+erf(x/1.41421))) so that low-precision dtypes round the same way
    the un-fused TorchScript function does on CPU."""
    return x * 0.5 * (1.0 + torch.erf(x / GELU_DIVISOR))


def rope_angles(head_dim: int, n_pos: int, theta: float = 10000.0) -> torch.Tensor:
    """(n_pos, head_dim/2) fp32 angles t * theta^(-2j/head_dim) (training/model.py:53-58)."""
    j = torch.arange(0, head_dim, 2)[: head_dim // 2].float()
    inv = 1.0 / (theta ** (j / head_dim))
    t = torch.arange(n_pos)
    return torch.outer(t, inv).float()


def rope_table(head_dim: int, n_pos: int, theta: float = 10000.0) -> torch.Tensor:
    """complex64 (n_pos, head_dim/2) table, unit modulus (training/model.py:59)."""
    ang = rope_angles(head_dim, n_pos, theta)
    return torch.polar(torch.ones_like(ang), ang)


def cast_rope_table(table: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """What ``nn.Module.to(dtype)`` does to the persistent complex buffer:
    a real floating dtype keeps only the real part (cos), rounded to that dtype."""
    if table.is_complex() and dtype.is_floating_point and not dtype.is_complex:
        return table.real.to(dtype)
    return table


def apply_rope(x: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """Rotate adjacent pairs (x[2j], x[2j+1]) of the last dim of x (B, T, H, hs) by table[t, j]
    (training/model.py:39-50).  ``table`` complex -> true rotation; real -> both components scaled by it
    (the degenerate mode the reference runs in after ``.to(bfloat16)``).  Arithmetic in fp32, result cast
    back to x.dtype, as the reference does."""
    B, T, H, hs = x.shape
    xf = x.float().reshape(B, T, H, hs // 2, 2)
    xe, xo = xf[..., 0], xf[..., 1]
    tab = table[:T]
    if tab.is_complex():
        c = tab.real.float().view(1, T, 1, hs // 2)
        s = tab.imag.float().view(1, T, 1, hs // 2)
    else:
        c = tab.float().view(1, T, 1, hs // 2)
        s = torch.zeros_like(c)
    oe = xe * c - xo * s
    oo = xe * s + xo * c
    return torch.stack((oe, oo), dim=-1).reshape(B, T, H, hs).to(x.dtype)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float,
              mask_add: Optional[torch.Tensor] = None, drop_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(q k^T * scale + mask) v with q,k,v (B, H, T, hs); non-causal
    (training/model.py:125-145, the 'manual' path; the SDPA path :118-122,134-138 is the same math).
    ``drop_mask`` (optional, (B, H, T, T), 1/(1-p) where kept and 0 where dropped): attn_dropout on the probabilities
    (model.py:121,129,137,144)."""
    att = (q @ k.transpose(-2, -1)) * scale
    if mask_add is not None:
        att = att + mask_add
    att = torch.softmax(att, dim=-1)
    if drop_mask is not None:
        att = att * drop_mask
    return att @ v


def readout(emb: torch.Tensor, weight: torch.Tensor, width_mult: float, output_mult: float = 1.0) -> torch.Tensor:
    """MuReadout: Linear(output_mult * x / width_mult), no bias (training/model.py:208,253; mup 1.0.0)."""
    return F.linear(output_mult * emb / width_mult, weight)


# --------------------------------------------------------------------------------------------------
# dropout masks of the HIP path, restated (include/omnibiote_hip.h "dropout"; csrc/common.h drop_keep).
# The reference uses PyTorch's generator (model.py:83-84,160,204), whose stream no other implementation can
# reproduce; the product instead derives every mask from a counter-based hash of (seed, site, row, column),
# which this restatement mirrors bit for bit so that dropout-on parity tests can hand the oracle the same mask.
# --------------------------------------------------------------------------------------------------
_M32 = np.uint64(0xFFFFFFFF)


def _hash32(x: np.ndarray) -> np.ndarray:
    x = x & _M32
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def dropout_keep(rows: np.ndarray, cols: np.ndarray, p: float, seed: int, site: int) -> np.ndarray:
    """Boolean keep decision for the elements (row, col) of dropout site ``site`` (csrc/common.h drop_rowkey / drop_keep):
    rowkey = hash32(hash32(lo32(row) ^ s0) + hi32(row) * 0x9E3779B1 + s1); bits = hash32(rowkey ^ (col >> 1)); the element
    owns the low (even col) or high (odd col) 16 bits, kept iff they are >= round(p * 2^16).  ``rows`` / ``cols`` broadcast."""
    rows = np.asarray(rows, dtype=np.uint64)
    cols = np.asarray(cols, dtype=np.uint64)
    s0 = np.uint64(((seed & 0xFFFFFFFF) ^ ((site * 0x632BE5AB) & 0xFFFFFFFF)) & 0xFFFFFFFF)
    s1 = np.uint64(((seed >> 32) + site * 0x9E3779B9) & 0xFFFFFFFF)
    thresh = np.uint64(max(int(float(np.float32(p)) * 65536.0 + 0.5), 1)) if p > 0 else np.uint64(0)
    rk = _hash32((rows & _M32) ^ s0)
    rk = _hash32((rk + (((rows >> np.uint64(32)) * np.uint64(0x9E3779B1)) & _M32) + s1) & _M32)
    bits = _hash32(rk ^ (cols >> np.uint64(1)))
    own = np.where((cols & np.uint64(1)) == 1, bits >> np.uint64(16), bits & np.uint64(0xFFFF))
    return own >= thresh


def dropout_scale_mask(shape, p: float, seed: int, site: int) -> torch.Tensor:
    """fp32 tensor of ``shape`` holding 1/(1-p) where kept and 0 where dropped.  The tensor is read as a matrix: col = the
    last dimension, row = the flat index of the leading dimensions — (token, feature) for activations, ((b*H + h)*T + q,
    key) for attention probabilities of shape (B, H, T, T)."""
    shape = tuple(int(v) for v in shape)
    ncol = shape[-1]
    nrow = int(np.prod(shape[:-1])) if len(shape) > 1 else 1
    keep = dropout_keep(np.arange(nrow, dtype=np.uint64)[:, None], np.arange(ncol, dtype=np.uint64)[None, :], p, seed, site)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return torch.from_numpy(np.where(keep, scale, np.float32(0.0)).astype(np.float32).reshape(shape))


def dropout_apply(x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """dropout as the product rounds it: bf16(x * scale) where kept (x already carries its own dtype)."""
    return (x.float() * mask).to(x.dtype)


# --------------------------------------------------------------------------------------------------
# attention-mask builder (training/train_encoder.py:25-57) restated over numpy
# --------------------------------------------------------------------------------------------------
def document_blocks(tokens: np.ndarray, eos: int = EOS_TOKEN, padding: bool = False) -> List[List[Tuple[int, int]]]:
    """Blocks [start, end) of mutually-attending positions, per batch row, exactly as the reference's
    ``create_attention_mask`` ends up zeroing them — including its quirk that in every row after the
    first one found, the first EOS does not advance ``prev_index`` (train_encoder.py:48-51), so the first
    two documents of those rows share one block.  Ends are clipped to T."""
    tokens = np.asarray(tokens)
    B, T = tokens.shape
    if not padding:
        ext = np.concatenate([tokens, np.full((B, 1), eos, dtype=tokens.dtype)], axis=1)
    else:
        ext = tokens
    rows, cols = np.nonzero(ext == eos)  # row-major order == torch.nonzero order
    blocks: List[List[Tuple[int, int]]] = [[] for _ in range(B)]
    prev_index = 0
    prev_row = 0
    for r, c in zip(rows.tolist(), cols.tolist()):
        if r == prev_row:
            blocks[prev_row].append((prev_index, c + 1))
            prev_index = c + 1
        else:
            prev_row = r
            prev_index = 0
            blocks[prev_row].append((prev_index, c + 1))
            # prev_index deliberately NOT advanced: reference quirk
    seen = set(rows.tolist())
    for r in range(B):
        if r not in seen:
            blocks[r] = [(0, T)]  # train_encoder.py:53-55: rows without EOS attend everywhere
    return [[(s, min(e, T)) for (s, e) in row] for row in blocks]


def dense_mask_from_blocks(blocks: List[List[Tuple[int, int]]], T: int, dtype=torch.float32) -> torch.Tensor:
    """(B, T, T) additive mask with 0 inside blocks and -1e9 elsewhere.  Blocks are applied in order, like
    the reference's successive slice assignments (later blocks may overlap earlier ones)."""
    B = len(blocks)
    m = torch.full((B, T, T), MASKED_VALUE, dtype=torch.float32)
    for b, row in enumerate(blocks):
        for (s, e) in row:
            m[b, s:e, s:e] = 0.0
    return m.to(dtype)


def key_ranges_from_blocks(blocks: List[List[Tuple[int, int]]], T: int) -> np.ndarray:
    """(B, T, 2) int32 [k_start, k_end) per query.  Valid because the union of overlapping reference
    blocks (the quirk merges block 0 and 1 into [0, e1)) is still a set of disjoint contiguous ranges:
    after painting, query t attends the union of blocks containing it."""
    B = len(blocks)
    out = np.zeros((B, T, 2), dtype=np.int32)
    for b, row in enumerate(blocks):
        allowed = np.zeros((T, T), dtype=bool)
        for (s, e) in row:
            allowed[s:e, s:e] = True
        for t in range(T):
            ks = np.nonzero(allowed[t])[0]
            if len(ks) == 0:
                out[b, t] = (0, 0)
            else:
                assert ks[-1] - ks[0] + 1 == len(ks), "non-contiguous key set"
                out[b, t] = (ks[0], ks[-1] + 1)
    return out


# --------------------------------------------------------------------------------------------------
# model-level restatement
# --------------------------------------------------------------------------------------------------
@dataclass
class RefConfig:
    """Field names/defaults of OmniBioTAConfig (training/model.py:183-193) + the ad-hoc ``flash``
    (train_encoder.py:152) + the muP base width used for the readout multiplier."""
    block_size: int = 2048
    vocab_size: int = 2 ** 16
    n_layer: int = 12
    n_head: int = 12
    n_embd: int = 1024
    dropout: float = 0.1
    bias: bool = False
    autoregressive: bool = False
    checkpoint_freq: int = 0
    flash: bool = True
    mup_base_width: int = MUP_BASE_WIDTH


def param_names(n_layer: int) -> List[str]:
    """state_dict parameter keys in registration order (training/model.py:202-208, :173-176)."""
    names = ["transformer.wte.weight"]
    for i in range(n_layer):
        p = f"transformer.h.{i}."
        names += [p + "ln_1.weight", p + "attn.c_attn.weight", p + "attn.c_proj.weight",
                  p + "ln_2.weight", p + "mlp.c_fc.weight", p + "mlp.c_proj.weight"]
    names += ["transformer.ln_f.weight", "lm_head.weight"]
    return names


def param_shapes(cfg: RefConfig) -> Dict[str, Tuple[int, ...]]:
    C, V = cfg.n_embd, cfg.vocab_size
    shapes: Dict[str, Tuple[int, ...]] = {"transformer.wte.weight": (V, C)}
    for i in range(cfg.n_layer):
        p = f"transformer.h.{i}."
        shapes[p + "ln_1.weight"] = (C,)
        shapes[p + "attn.c_attn.weight"] = (3 * C, C)
        shapes[p + "attn.c_proj.weight"] = (C, C)
        shapes[p + "ln_2.weight"] = (C,)
        shapes[p + "mlp.c_fc.weight"] = (4 * C, C)
        shapes[p + "mlp.c_proj.weight"] = (C, 4 * C)
    shapes["transformer.ln_f.weight"] = (C,)
    shapes["lm_head.weight"] = (V, C)
    return shapes


def hash_weights(cfg: RefConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Closed-form, torch-version-independent fp32 weights: an integer hash of (tensor ordinal, flat index)
    mapped to a uniform in [-a, a).  Amplitudes mimic the reference's default init scale
    (U(+-1/sqrt(fan_in)) for Linear, unit-ish for wte, around 1 for LayerNorm) so activations are
    realistically sized.  Used by the golden-vector generator and by every parity test, so that only
    inputs and outputs need to be stored."""
    out: Dict[str, torch.Tensor] = {}
    for ordinal, (name, shape) in enumerate(param_shapes(cfg).items()):
        n = int(np.prod(shape))
        i = np.arange(n, dtype=np.uint64)
        h = (i * np.uint64(2654435761) + np.uint64((ordinal + 1) * 40503 + seed * 7919)) & np.uint64(0xFFFFFFFF)
        h ^= h >> np.uint64(15)
        h = (h * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
        h ^= h >> np.uint64(13)
        u = (h & np.uint64(0xFFFF)).astype(np.float64) / 65536.0 - 0.5  # exact in fp32: k/65536 - 0.5
        if name.endswith("wte.weight"):
            w = u * 2.0
        elif "ln_" in name:
            w = 1.0 + u * 0.5
        else:
            w = u * (2.0 / math.sqrt(shape[-1]))
        out[name] = torch.from_numpy(w.astype(np.float32).reshape(shape))
    return out


def block_forward(x: torch.Tensor, p: Dict[str, torch.Tensor], prefix: str, cfg: RefConfig,
                  rope: torch.Tensor, mask_add: Optional[torch.Tensor], drop: Optional[Tuple[float, int]] = None,
                  drop_masks=None) -> torch.Tensor:
    """Pre-LN residual block (training/model.py:170-181) with SelfAttention (:98-152) and MLP (:162-168).
    Dropout (model.py:83-84,121,151,160,167) is off unless asked for: ``drop = (p, seed)`` applies the product's restated
    counter-based masks of this block (dropout_scale_mask: sites 1 attention probabilities, 2 attention projection, 3 MLP
    projection) — PyTorch's own generator stream cannot be reproduced by anybody, so dropout-on parity is stated against the
    same masks; ``drop_masks = (attn, resid, mlp)`` hands ready-made scale masks over instead (any of them None = none;
    the CPU baseline of the dropout regime draws them with torch's generator, as the reference does)."""
    B, T, C = x.shape
    m_attn = m_res = m_mlp = None
    if drop is not None and drop[0] > 0:
        m_attn = dropout_scale_mask((B, cfg.n_head, T, T), drop[0], drop[1], 1)
        m_res = dropout_scale_mask((B, T, C), drop[0], drop[1], 2)
        m_mlp = dropout_scale_mask((B, T, C), drop[0], drop[1], 3)
    elif drop_masks is not None:
        m_attn, m_res, m_mlp = drop_masks
    H = cfg.n_head
    hs = C // H
    h1 = layer_norm(x, p[prefix + "ln_1.weight"])
    qkv = F.linear(h1, p[prefix + "attn.c_attn.weight"])
    q, k, v = qkv.split(C, dim=2)
    q = apply_rope(q.reshape(B, T, H, hs), rope).transpose(1, 2)
    k = apply_rope(k.reshape(B, T, H, hs), rope).transpose(1, 2)
    v = v.reshape(B, T, H, hs).transpose(1, 2)
    y = attention(q, k, v, 8.0 / C, mask_add, m_attn)  # scale 8/n_embd: training/model.py:119
    y = y.transpose(1, 2).contiguous().view(B, T, C)
    yo = F.linear(y, p[prefix + "attn.c_proj.weight"])
    x = x + (yo if m_res is None else yo * m_res)
    h2 = layer_norm(x, p[prefix + "ln_2.weight"])
    a = gelu_erf(F.linear(h2, p[prefix + "mlp.c_fc.weight"]))
    mo = F.linear(a, p[prefix + "mlp.c_proj.weight"])
    x = x + (mo if m_mlp is None else mo * m_mlp)
    return x


def model_forward(p: Dict[str, torch.Tensor], cfg: RefConfig, idx: torch.Tensor,
                  mask_add: Optional[torch.Tensor] = None, return_embeddings: bool = False,
                  rope: Optional[torch.Tensor] = None, dropout: Optional[Tuple[float, List[int]]] = None,
                  torch_dropout_p: float = 0.0) -> torch.Tensor:
    """OmniBioTA.forward (training/model.py:225-254).  ``mask_add`` is the additive (B, 1|H, T, T) mask.
    ``rope`` defaults to what the reference's module would hold: the complex table for fp32 parameters (the
    reference never calls ``.to(float32)``), the real cos-only table after ``.to(bfloat16)``/``.to(half)``.
    ``dropout = (p, [embedding seed, block 0's seed, block 1's seed, ...])``: training-mode dropout under the product's restated
    masks (block_forward; the embedding's is site 0 of its own seed: model.py:204,242).  ``torch_dropout_p``: training-mode dropout
    with masks from torch's generator at the reference's four sites — what the reference's own CPU step does; used for timing
    (bench.py's cpu_baseline at the reference's default --dropout 0.1), not for parity."""
    B, T = idx.shape
    assert T <= cfg.block_size
    wte = p["transformer.wte.weight"]
    if rope is None:
        rope = rope_table(cfg.n_embd // cfg.n_head, cfg.block_size)
        if wte.dtype != torch.float32:
            rope = cast_rope_table(rope, wte.dtype)
    x = F.embedding(idx, wte)
    if dropout is not None and dropout[0] > 0:
        x = x * dropout_scale_mask(tuple(x.shape), dropout[0], dropout[1][0], 0).to(x.device)
    elif torch_dropout_p > 0:
        x = F.dropout(x, torch_dropout_p, True)
    for i in range(cfg.n_layer):
        if torch_dropout_p > 0:   # the reference's bernoulli_ calls (23 % of its CPU step), scale masks of the same shapes
            keep = 1.0 - torch_dropout_p
            mk = lambda shape: (torch.empty(shape, dtype=x.dtype, device=x.device).bernoulli_(keep) / keep)   # noqa: E731
            x = block_forward(x, p, f"transformer.h.{i}.", cfg, rope, mask_add,
                              drop_masks=(mk((B, cfg.n_head, T, T)), mk(tuple(x.shape)), mk(tuple(x.shape))))
        else:
            x = block_forward(x, p, f"transformer.h.{i}.", cfg, rope, mask_add,
                              drop=None if dropout is None else (dropout[0], dropout[1][1 + i]))
    emb = layer_norm(x, p["transformer.ln_f.weight"])
    if return_embeddings:
        return emb
    return readout(emb, p["lm_head.weight"], cfg.n_embd / cfg.mup_base_width)


def encode_pool(emb: torch.Tensor, method: str) -> torch.Tensor:
    """Pooling of OmniBioTA.encode (training/model.py:256-277)."""
    assert method in ("mean", "first", "last", "max", "all"), f"Unknown pooling method {method}"
    if method == "mean":
        return emb.mean(dim=1)
    if method == "first":
        return emb[:, 0]
    if method == "last":
        return emb[:, -1]
    if method == "max":
        return emb.max(dim=1)[0]
    return emb


def masked_lm_loss(logits: torch.Tensor, targets: torch.Tensor, mlm_mask: torch.Tensor, n_accum: int) -> torch.Tensor:
    """The reference's micro-batch loss (training/train_encoder.py:301-305): per-token CE divided by the
    number of accumulation steps, zeroed outside the MLM mask, summed, divided by the mask count.  The
    in-place multiply keeps the loss in the logits' dtype, as the reference's ``loss *= mask.float()``."""
    loss = F.cross_entropy(logits.view(-1, logits.size(-1)), targets.reshape(-1), reduction="none") / n_accum
    loss *= mlm_mask.reshape(-1).float()
    return loss.sum() / mlm_mask.reshape(-1).sum()


def mlm_corrupt(tokens: torch.Tensor, bern: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """MLM corruption (training/train_encoder.py:276-279): Bernoulli draw AND not PAD AND not EOS; every
    selected token becomes MASK_TOKEN (no 80/10/10)."""
    mask = bern.bool() & (tokens != PAD_TOKEN) & (tokens != EOS_TOKEN)
    return tokens.masked_fill(mask, MASK_TOKEN), mask


# --------------------------------------------------------------------------------------------------
# nn.Module wrapper: same parameter tree as the reference, forward = the restatement above.  Used by the
# CPU harness tests (gloo, world_size 2) and by bench.py's cpu_baseline leg.
# --------------------------------------------------------------------------------------------------
class OracleEncoder(nn.Module):
    def __init__(self, cfg: RefConfig, weights: Optional[Dict[str, torch.Tensor]] = None):
        super().__init__()
        self.cfg = cfg
        shapes = param_shapes(cfg)
        if weights is None:
            weights = hash_weights(cfg)
        self._names = list(shapes.keys())
        self.params = nn.ParameterList([nn.Parameter(weights[n].clone()) for n in self._names])
        self.register_buffer("rope", rope_table(cfg.n_embd // cfg.n_head, cfg.block_size))

    def named_weights(self) -> Dict[str, torch.Tensor]:
        return {n: w for n, w in zip(self._names, self.params)}

    def get_num_params(self, non_embedding: bool = True) -> int:
        n = sum(w.numel() for w in self.params)
        if non_embedding:
            n -= self.params[0].numel()
        return n

    torch_dropout_p = 0.0   # > 0 (bench.py's cpu_baseline at the reference's default): training-mode dropout, torch's generator

    def forward(self, idx, attn_mask=None, return_embeddings=False):
        return model_forward(self.named_weights(), self.cfg, idx, attn_mask, return_embeddings, rope=self.rope,
                             torch_dropout_p=self.torch_dropout_p if self.training else 0.0)


def flops_per_token(cfg: RefConfig, T: int) -> float:
    """6N + 12 L C T with N = all parameters minus wte (training/train_encoder.py:360, model.py:213-223)."""
    C, L, V = cfg.n_embd, cfg.n_layer, cfg.vocab_size
    n = 12 * C * C * L + C * V + (2 * L + 1) * C
    return 6.0 * n + 12.0 * L * C * T
