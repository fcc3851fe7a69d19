"""Attention masks of the OmniBioTE trainer as per-query key ranges.

Every mask the reference builds (training/train_encoder.py:25-57 ``create_attention_mask``) is
block-diagonal with contiguous blocks and values {0, -1e9}, shipped as a dense (B, n_head, T, T) tensor.  The
fused attention kernels take the same information as int32 ``[k_start, k_end)`` per query — 8 bytes per token
instead of 2*T — and skip KV tiles outside a workgroup's range.

``RangeMask.from_tokens`` is the builder: pure tensor ops on whatever device the tokens live on (no ``nonzero()``,
no host sync, no Python loop over EOS positions) and reproduces the reference builder's output exactly, including
its quirk: in every batch row except row 0 the first EOS does not advance the block start (train_encoder.py:48-51),
so the first two documents of those rows share one block.

One documented difference: positions that the reference leaves fully masked (the PAD tail after the last EOS in
``--use_padding`` mode) get the empty range here.  The reference's softmax over an all -1e9 row degenerates to a
uniform average of V there; the range kernels return 0.  Those positions are PAD: they are excluded from the loss
(train_encoder.py:278) and no other position attends to them, so neither the loss nor any gradient changes.
"""
from __future__ import annotations

import torch

EOS_TOKEN = 3   # training/loader.py:4
MASKED_VALUE = -1e9


class RangeMask:
    """key_ranges: int32 (B, T, 2), [k_start, k_end) of the keys each query may attend to."""

    def __init__(self, key_ranges: torch.Tensor):
        assert key_ranges.dtype == torch.int32 and key_ranges.dim() == 3 and key_ranges.shape[-1] == 2
        self.key_ranges = key_ranges.contiguous()

    @property
    def shape(self):
        B, T, _ = self.key_ranges.shape
        return (B, T, T)

    def to(self, device):
        return RangeMask(self.key_ranges.to(device))

    @staticmethod
    def from_tokens(input_ids: torch.Tensor, eos_token: int = EOS_TOKEN, padding: bool = False, group: int = 0) -> "RangeMask":
        """``group`` > 0: ``input_ids`` stacks several mini-batches of ``group`` rows each (the reference builds one
        mask per mini-batch, and its row-0 exception applies to the first row of EACH mini-batch) — one pass of tensor
        ops for a whole optimizer step's rows instead of one per micro-step."""
        B, T = input_ids.shape
        dev = input_ids.device
        is_eos = input_ids == eos_token
        if not padding:  # the reference appends an EOS column (train_encoder.py:33-37)
            is_eos = torch.cat([is_eos, torch.ones(B, 1, dtype=torch.bool, device=dev)], dim=1)
        Tx = is_eos.shape[1]
        pos = torch.arange(Tx, device=dev).expand(B, Tx)
        BIG = Tx + 1
        # next EOS at or after t (BIG if none), last EOS strictly before t (-1 if none)
        nxt = torch.where(is_eos, pos, torch.full_like(pos, BIG))
        nxt = torch.flip(torch.cummin(torch.flip(nxt, dims=[1]), dim=1).values, dims=[1])
        prv = torch.cummax(torch.where(is_eos, pos, torch.full_like(pos, -1)), dim=1).values
        prv = torch.cat([torch.full((B, 1), -1, dtype=prv.dtype, device=dev), prv[:, :-1]], dim=1)
        start = prv + 1
        end = nxt + 1
        # the quirk: rows b >= 1 — the first EOS of the row does not advance the block start
        n_before = torch.cumsum(is_eos.long(), dim=1) - is_eos.long()
        c1 = nxt[:, 0]                                                    # first EOS of the row (BIG if none)
        after_c1 = torch.clamp(c1 + 1, max=Tx - 1)
        c2 = torch.where(c1 + 1 < Tx, nxt.gather(1, after_c1.unsqueeze(1)).squeeze(1), torch.full_like(c1, BIG))
        has_c2 = (c2 < BIG).unsqueeze(1)
        row = torch.arange(B, device=dev)
        quirk_row = ((row % group if group > 0 else row) >= 1).unsqueeze(1)
        end = torch.where(quirk_row & (n_before == 0) & has_c2, (c2 + 1).unsqueeze(1).expand(B, Tx), end)
        start = torch.where(quirk_row & (n_before == 1), torch.zeros_like(start), start)
        # positions with no EOS at or after them are never painted: empty range ...
        empty = nxt >= BIG
        start = torch.where(empty, torch.zeros_like(start), start)
        end = torch.where(empty, torch.zeros_like(end), end)
        # ... except rows without any EOS, which attend everywhere (train_encoder.py:53-55)
        no_eos = ~is_eos.any(dim=1, keepdim=True)
        start = torch.where(no_eos, torch.zeros_like(start), start)
        end = torch.where(no_eos, torch.full_like(end, T), end)
        rng = torch.stack([start[:, :T], torch.clamp(end[:, :T], max=T)], dim=2).to(torch.int32)
        return RangeMask(rng)

    def dense(self, dtype=torch.bfloat16) -> torch.Tensor:
        """The reference's additive (B, T, T) tensor: 0 where attention happens, -1e9 elsewhere."""
        B, T, _ = self.key_ranges.shape
        k = torch.arange(T, device=self.key_ranges.device).view(1, 1, T)
        allowed = (k >= self.key_ranges[..., 0:1]) & (k < self.key_ranges[..., 1:2])
        out = torch.full((B, T, T), MASKED_VALUE, dtype=torch.float32, device=allowed.device)
        return out.masked_fill(allowed, 0.0).to(dtype)

    @staticmethod
    def from_dense(attn_mask: torch.Tensor) -> "RangeMask":
        """Convert a reference-style additive mask (B, [H,] T, T) to ranges.  Validates (one host sync) that each
        row's zero set is one contiguous run and that all heads share the mask; raises ValueError otherwise."""
        m = attn_mask
        if m.dim() == 4:
            if m.shape[1] > 1 and m.stride(1) != 0 and not bool((m == m[:, :1]).all()):
                raise ValueError("per-head masks cannot be expressed as key ranges")
            m = m[:, 0]
        B, T, _ = m.shape
        allowed = m == 0
        if not bool(((m == 0) | (m <= -1e8)).all()):
            raise ValueError("mask values other than 0 / -1e9 cannot be expressed as key ranges")
        cnt = allowed.sum(dim=2)
        k = torch.arange(T, device=m.device).view(1, 1, T)
        first = torch.where(allowed, k, torch.full_like(k, T)).min(dim=2).values
        last = torch.where(allowed, k, torch.full_like(k, -1)).max(dim=2).values
        if not bool(((last - first + 1 == cnt) | (cnt == 0)).all()):
            raise ValueError("a mask row with a non-contiguous key set cannot be expressed as key ranges")
        start = torch.where(cnt > 0, first, torch.zeros_like(first))
        end = torch.where(cnt > 0, last + 1, torch.zeros_like(last))
        return RangeMask(torch.stack([start, end], dim=2).to(torch.int32))


def create_attention_mask(attn_mask: torch.Tensor, input_ids: torch.Tensor, EOS_TOKEN: int = EOS_TOKEN,
                          padding: bool = False) -> torch.Tensor:
    """Signature-compatible with the reference's TorchScript builder (train_encoder.py:31-57): fills ``attn_mask``
    (B, T, T) in place with 0 / -1e9 and returns it — built from ranges, without the per-EOS Python loop."""
    dense = RangeMask.from_tokens(input_ids, EOS_TOKEN, padding).dense(attn_mask.dtype)
    attn_mask.copy_(dense)
    return attn_mask
