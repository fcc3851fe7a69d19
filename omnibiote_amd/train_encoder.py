"""Data-parallel masked-LM training harness: the counterpart of the reference's ``training/train_encoder.py``
for the MI355X path (one process per GPU under torchrun, DDP over RCCL).

Kept from the reference (file:line = training/train_encoder.py):
  * flags --batch_size / --mini_batch_size / --n_head / --n_embd / --n_layer / --ctx_len / --dropout / --lr / betas /
    --epsilon / --weight_decay / --token_budget / --disable_flash / --force_lr / --checkpoint_freq / --use_padding /
    --batch_ramp / --warmup_period (:438-467), global batch split evenly over ranks (:115-118);
  * model + muP set-up: target, base (n_embd 24 / n_head 3) and delta (48 / 12) models, set_base_shapes, bf16 (:145-170);
  * lr = args.lr * sqrt(batch_size) / 32, MuAdamW (or AdamW with --force_lr), LinearLR 1 -> 0 (:195-201);
  * MLM corruption with the host NumPy RNG, 15 %, every selected token -> MASK (:273-279);
  * gradient accumulation over rows/mini_batch_size micro-batches with loss / n_accum, masked-only CE averaged over
    the masked tokens of the micro-batch (:284-305); clip_grad_norm_(1.0), optimizer, scheduler (:316-318);
  * tokens/s = non-PAD tokens of all ranks / wall time, MFU formula 6N + 12 L C T (:350-364).
Changed on purpose (same mathematics, MI355X-first mechanics):
  * the attention mask travels as per-query key ranges built on the device (masks.RangeMask) instead of a dense
    (B, H, T, T) tensor built by a Python loop behind a nonzero() host sync (:290-292);
  * gradients are all-reduced once per optimizer step — DDP ``no_sync()`` on all but the last micro-batch — instead
    of after every micro-batch (the reference never calls no_sync, :284-311); averaging already-averaged sums is
    idempotent, so the result is the same while 1/n_accum of the bytes cross xGMI, in buckets that overlap with the
    last micro-batch's backward;
  * loss: fused CE forward+backward kernel (one pass over the logits); the per-step scalars (loss, token count) are
    reduced with one small device all-reduce instead of two pickled gloo all_gather_object calls (:335,354);
  * clip + AdamW run as fused kernels without a host sync (the clip coefficient stays on the device).
Data: with ``--base_dir`` pointing at the reference's directory layout (``genbank/train``, ``uniref100/train`` ... of
``.npy`` token shards, train_encoder.py:70-99) batches come from ``omnibiote_amd.loader`` (same packing and mixing as the
reference loader, loader thread + bounded queue, pinned-memory copies on their own stream); otherwise from synthetic rows
of the same contract.  Checkpoints follow the reference's convention (whole-object model pickle every
``--save_freq`` tokens, previous one removed, train_encoder.py:412-423); optimizer and scheduler are saved as state_dicts.
wandb logging is replaced by one line per step on rank 0.
"""
from __future__ import annotations

import argparse
import contextlib
import math
import os
import time
from typing import Callable, Dict, Iterable, List, Optional

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

EOS_TOKEN, MASK_TOKEN, PAD_TOKEN = 3, 2, 1   # training/loader.py:4-6
DNA_TAG, PROTEIN_TAG = 4, 18                   # tokenizer ids
BANNED_MIXED = 65533                           # train_encoder.py:62-67


# ----------------------------------------------------------------------------------------------- synthetic data
def synthetic_rows(rows: int, ctx_len: int, vocab: int, rng: np.random.Generator, single_document: bool = True,
                   nucleotide_fraction: float = 0.8) -> np.ndarray:
    """int64 (rows, ctx_len) in the packed, un-padded format get_sequence(..., USE_PADDING=False) yields
    (loader.py:118-163): documents ``tag, body..., EOS`` concatenated and truncated at ctx_len; 80 % of the rows
    nucleotide (tag 4), 20 % peptide (tag 18) as the 'mixed' split (train_encoder.py:82-86); body ids uniform over
    [20, vocab) minus the banned id.  single_document=True makes every row one document longer than ctx_len (no
    interior EOS -> full attention), the variant the FLOP formula assumes."""
    out = np.empty((rows, ctx_len), dtype=np.int64)
    hi = min(vocab, 65536)
    for r in range(rows):
        nucleotide = rng.random() < nucleotide_fraction
        tag = DNA_TAG if nucleotide else PROTEIN_TAG
        pos = 0
        while pos < ctx_len:
            if single_document:
                n = ctx_len
            else:
                lo_len, hi_len = (256, 8192) if nucleotide else (32, 512)
                n = int(math.exp(rng.uniform(math.log(lo_len), math.log(hi_len))))
            doc = rng.integers(20, hi, size=n + 2)
            doc[doc == BANNED_MIXED] = 20
            doc[0] = tag
            doc[-1] = EOS_TOKEN
            take = min(len(doc), ctx_len - pos)
            out[r, pos:pos + take] = doc[:take]
            pos += take
    rng.shuffle(out, axis=0)   # loader.py:176
    return out


def mlm_corrupt(input_ids: torch.Tensor, mask_prob: float = 0.15):
    """train_encoder.py:273-279: host NumPy Bernoulli, minus PAD and EOS positions, all selected -> MASK_TOKEN."""
    draw = np.random.binomial(1, mask_prob, tuple(input_ids.shape))
    mask = torch.as_tensor(draw, dtype=torch.bool, device=input_ids.device)
    mask = mask & (input_ids != PAD_TOKEN) & (input_ids != EOS_TOKEN)
    return input_ids.masked_fill(mask, MASK_TOKEN), mask


# ---------------------------------------------------------------------------------------------------- optimizer
class FusedAdamW(torch.optim.Optimizer):
    """AdamW over bf16 parameters with bf16 moments (the reference's pure-bf16 regime) on the fused HIP kernels.
    ``step(max_norm=...)`` also applies clip_grad_norm_ semantics (train_encoder.py:316) without a host sync: the
    squared norm is accumulated on the device and the coefficient min(1, max_norm/(norm+1e-6)) is consumed by the
    update kernel directly."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, rounding: str = "reference"):
        """rounding="reference" (default): every tensor op of torch.optim.AdamW's bf16 update is rounded to bf16 exactly
        where the reference's optimizer rounds it (obte_adamw_multi_bf16_ref), so trajectories track the reference's;
        "single": fp32 arithmetic per element with one rounding per state (more accurate, not the reference's numbers)."""
        assert rounding in ("reference", "single")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.rounding = rounding
        self._norm_sq = None

    @torch.no_grad()
    def step(self, max_norm: Optional[float] = None):
        """Multi-tensor launches: <= 32 tensors per kernel (per parameter group, since betas/eps are per group)."""
        import ctypes as C
        from . import _lib as L
        from . import ops
        lib = L.lib()
        stream = torch.cuda.current_stream().cuda_stream
        batches = []   # (MtArgs, betas, eps)
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            for i in range(0, len(ps), L.MT_MAX):
                chunk = ps[i:i + L.MT_MAX]
                a = L.MtArgs()
                a.count = len(chunk)
                for j, p in enumerate(chunk):
                    st = self.state[p]
                    if not st:
                        st["step"] = 0
                        st["exp_avg"] = torch.zeros_like(p)
                        st["exp_avg_sq"] = torch.zeros_like(p)
                    st["step"] += 1
                    if p.dtype != torch.bfloat16 or p.numel() % 8 or not p.is_cuda:
                        raise RuntimeError("FusedAdamW needs bf16 GPU parameters with numel % 8 == 0")
                    a.p[j], a.g[j], a.m[j], a.v[j] = p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                    a.n[j], a.lr[j], a.weight_decay[j], a.step[j] = p.numel(), float(group["lr"]), float(group["weight_decay"]), st["step"]
                    a.lr64[j], a.weight_decay64[j] = float(group["lr"]), float(group["weight_decay"])   # un-rounded, for the _ref kernel's double-formed scalars
                batches.append((a, group["betas"], group["eps"]))
        if not batches:
            return None
        clip = None
        if max_norm is not None and self.rounding == "reference":
            # clip_grad_norm_ on bf16 gradients (train_encoder.py:316): per-tensor norms rounded to bf16, their 2-norm
            # rounded to bf16, the coefficient max_norm / (total + 1e-6) formed and clamped in bf16
            dev = torch.device("cuda", torch.cuda.current_device())
            n_tensors = sum(a.count for a, _, _ in batches)
            if self._norm_sq is None or self._norm_sq.device != dev or self._norm_sq.numel() != n_tensors:
                self._norm_sq = torch.zeros(n_tensors, dtype=torch.float32, device=dev)
            self._norm_sq.zero_()
            off = 0
            for a, _, _ in batches:
                L.check(lib.obte_sumsq_multi_bf16_each(C.byref(a), self._norm_sq.data_ptr() + 4 * off, stream), "obte_sumsq_multi_bf16_each")
                off += a.count
            total = torch.linalg.vector_norm(self._norm_sq.sqrt().to(torch.bfloat16))
            clip = torch.clamp(max_norm / (total + 1e-6), max=1.0).float().reshape(1)
        elif max_norm is not None:
            dev = torch.device("cuda", torch.cuda.current_device())
            if self._norm_sq is None or self._norm_sq.device != dev or self._norm_sq.numel() != 1:
                self._norm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
            self._norm_sq.zero_()
            for a, _, _ in batches:
                L.check(lib.obte_sumsq_multi_bf16(C.byref(a), self._norm_sq.data_ptr(), stream), "obte_sumsq_multi_bf16")
            clip = torch.clamp(max_norm / (self._norm_sq.sqrt() + 1e-6), max=1.0)
        fn = lib.obte_adamw_multi_bf16_ref if self.rounding == "reference" else lib.obte_adamw_multi_bf16
        for a, (b1, b2), eps in batches:
            L.check(fn(C.byref(a), b1, b2, eps, None if clip is None else clip.data_ptr(), stream), "obte_adamw_multi_bf16")
        return None if clip is None else self._norm_sq.sum()


# --------------------------------------------------------------------------------------------------- train step
class TrainStep:
    """One optimizer step of train_encoder.py:241-323 over a (rows, ctx_len) batch of this rank's rows."""

    def __init__(self, model, optimizer, scheduler=None, *, mini_batch_size: int, n_head: int, use_padding: bool = False,
                 loss_impl: str = "fused", mask_impl: str = "ranges", sync_every_micro_step: bool = False,
                 max_grad_norm: float = 1.0, lm_head_impl: str = "masked", pipeline_streams: int = 1,
                 fused_loss_fn: Optional[Callable] = None, micro_batches_per_pass: int = 1, backward_order: str = "layer",
                 rows_forward: bool = True):
        """fused_loss_fn: ``(logits, targets, mlm_mask, n_accum) -> (loss, dlogits)`` used by loss_impl="fused" instead of
        the HIP kernel (ops.masked_ce) — lets the CPU multi-process tests drive the product scheduling (in-place
        accumulation, no_sync, hand-delivered d(logits)) with a stub model and a torch loss."""
        self.fused_loss_fn = fused_loss_fn
        # micro_batches_per_pass = k > 1 (the two row-compact readout paths): k consecutive micro-batches go through the model in ONE
        # forward/backward of k * mini_batch_size rows.  Rows never interact across a batch (attention is per row, the mask
        # builder's per-micro-batch quirk is kept), and the loss keeps the reference's normalisation — every masked row is
        # weighted 1 / (n_accum * masked tokens of ITS micro-batch), train_encoder.py:301-305 — so loss and gradients are those
        # of k separate passes up to summation order; every kernel simply sees k times the rows per launch.  An execution
        # option like pipeline_streams, off by default: --mini_batch_size keeps its meaning either way.
        self.per_pass = max(1, int(micro_batches_per_pass))
        self.model, self.optimizer, self.scheduler = model, optimizer, scheduler
        self.mini, self.n_head, self.use_padding = mini_batch_size, n_head, use_padding
        self.loss_impl, self.mask_impl = loss_impl, mask_impl
        self.sync_every = sync_every_micro_step
        self.max_grad_norm = max_grad_norm
        # "masked" (default): the loss multiplies the unmasked ~85 % of the positions by zero
        #     (train_encoder.py:304), so the readout (model.py:253, reached through forward(return_embeddings=True)) and the
        #     cross entropy run on the MLM-masked positions alone: logits [n_masked, V] instead of [B*T, V], CE over those rows
        #     (ops.masked_ce_rows), the two backward products over those rows (model._ReadoutRowsGradFn).  Same loss, same
        #     gradients — the positions left out contribute exact zeros — for 1/6.7 of the lm_head work.
        # "dense": the forward computes the logits of EVERY position, as the reference does (train_encoder.py:296); d(logits)
        #     has exact-zero rows outside the mask, so the readout's backward contracts over the masked rows only.
        # "dense_full": the same forward, and the backward through the dense [M, V] d(logits) tensor — the literal
        #     translation of the reference's graph (logits.backward(dlogits)); kept for A/B and for callers of model(x).
        assert lm_head_impl in ("dense", "dense_full", "masked")
        self.lm_head_impl = lm_head_impl
        self._dlogits = {}
        self._all_ranges = None
        # pipeline_streams = 2: micro-batches alternate between two HIP streams so that the forward of micro-batch j+1
        # runs beside the backward of micro-batch j (kernel tails and half-filled grids of one fill with work of the
        # other).  The backward passes stay strictly ordered (an event per micro-batch), so every gradient buffer sees
        # the same read-modify-write sequence as on one stream: results are bitwise those of pipeline_streams = 1.
        assert pipeline_streams in (1, 2, 3)
        self.pipeline_streams = pipeline_streams
        # backward_order (pipeline_streams = 2): "pass" — the backward of micro-batch j+1 starts after the LAST kernel of micro-batch
        # j's backward (one event per pass); "layer" (default) — every parameter group's update waits for the same group's update
        # of the previous micro-batch only (model.BackwardOrder: one event per block / LayerNorm / embedding / readout), so the
        # next backward follows one layer behind and both streams stay busy.  Each gradient buffer sees the same sequence of
        # read-modify-writes either way: bitwise the same results (tested against pipeline_streams = 1).
        assert backward_order in ("layer", "pass")
        self.backward_order = backward_order
        # lm_head_impl="masked": hand the list of masked positions to model.forward(rows=...) so that everything after the last
        # block's attention (its MLP half, ln_f, the readout) is computed for those positions only; False: forward(return_
        # embeddings=True) on every position, rows gathered afterwards (the reference's forward contract to the letter)
        self.rows_forward = bool(rows_forward) and hasattr(model.module if hasattr(model, "module") else model, "transformer")
        self._order = None           # the BackwardOrder of the pass being built
        self._prev_order_events = None
        self._streams = None
        self._slot = 0
        self._prev_bwd_done = None
        self._host_bufs = {}
        self._ln_store = None

    @staticmethod
    def _no_inplace() -> bool:
        """OBTE_NO_INPLACE_ACCUM=1 (A/B switch): every gradient is delivered through autograd's AccumulateGrad (`grad += new`,
        issued by the engine AFTER a node's backward returned), so no per-group event a node records can cover it: pipelined
        passes are then ordered as wholes (backward_order="pass": one event after the entire backward)."""
        return os.environ.get("OBTE_NO_INPLACE_ACCUM") == "1"

    def _inplace(self, enabled: bool, ln_partial_mode: int = 0):
        if self.loss_impl != "fused" or self._no_inplace():   # CPU-oracle tests / A-B switch (whole-pass ordering then: __call__)
            return contextlib.nullcontext()
        from .model import LnPartialStore, accumulate_grads_inplace
        if self.fused_loss_fn is not None or os.environ.get("OBTE_NO_LN_PARTIALS") == "1":   # stub models / A-B switch
            ln_partial_mode = 0
        if self._ln_store is None:
            self._ln_store = LnPartialStore()     # this step object's own fp32 LayerNorm partial sums
        return accumulate_grads_inplace(enabled, ln_partial_mode, store=self._ln_store, order=self._order)

    def _mask(self, tokens: torch.Tensor, dtype, j: int = -1, k: int = 1):
        from . import masks
        if j >= 0 and self._all_ranges is not None:   # built once per optimizer step for every micro-batch
            rm = masks.RangeMask(self._all_ranges[j * k * self.mini:(j + 1) * k * self.mini])
        else:
            rm = masks.RangeMask.from_tokens(tokens, padding=self.use_padding)
        if self.mask_impl == "ranges":
            return rm
        B, T = tokens.shape
        return rm.dense(dtype).unsqueeze(1).expand(-1, self.n_head, -1, -1)   # train_encoder.py:292

    def _loss_backward(self, logits, targets, mask, n_accum):
        if self.loss_impl == "fused" and self.fused_loss_fn is not None:
            loss, dlogits = self.fused_loss_fn(logits, targets, mask, n_accum)
            if logits.is_cuda:
                self._order_backward()
            logits.backward(dlogits)
            return loss.detach()
        if self.loss_impl == "fused":
            from . import ops
            if self._slot not in self._dlogits:   # one reusable d(logits) buffer per stream in flight
                self._dlogits[self._slot] = ops.DLogitsBuffer()
            loss, dlogits = ops.masked_ce(logits, targets, mask, n_accum, reuse=self._dlogits[self._slot])
            self._order_backward()
            logits.backward(dlogits)
            return loss.detach()
        # the reference's own three lines (train_encoder.py:301-305)
        loss = F.cross_entropy(logits.view(-1, logits.size(-1)), targets.reshape(-1), reduction="none") / n_accum
        loss *= mask.reshape(-1).float()
        loss = loss.sum() / mask.reshape(-1).sum()
        loss.backward()
        return loss.detach().float()

    def _order_backward(self, whole_pass: bool = False):
        """Pipelined micro-batches: this backward may start only after the previous micro-batch's backward finished.
        whole_pass: wait for the previous backward as a whole even under per-group ordering — for a graph whose gradients
        reach the readout weight through AccumulateGrad instead of a node that carries the BackwardOrder (the zero
        gradients of a pass with nothing masked)."""
        if getattr(self, "_join_side_streams", False):   # the isolated last pass of a DDP-wrapped model: everything before it is done
            self._join_side_streams = False
            for st in self._streams:
                torch.cuda.current_stream().wait_stream(st)
            return
        if self._order is not None and not whole_pass:   # per-group events instead (model.BackwardOrder); _prev_bwd_done is its fallback
            return
        if self._prev_bwd_done is not None:
            torch.cuda.current_stream().wait_event(self._prev_bwd_done)

    def _pass_targets(self, j: int, k: int):
        """The labels of the masked positions of pass j in _pass_rows' order, when the host prelude shipped them; else None."""
        lists = getattr(self, "_mask_targets_host", None)
        if lists is None:
            return None
        lists = lists[j * k:(j + 1) * k]
        if k == 1:
            return lists[0]
        keep = [t for t in lists if t.numel() > 0]
        return torch.cat(keep) if keep else lists[0]

    def _pass_rows(self, j: int, k: int, rows_per_mb: int):
        """Masked positions of pass j (micro-batches j*k .. j*k + k-1) as row indices into the pass's k * mini rows, and —
        for k > 1 — the weight 1 / (masked tokens of its own micro-batch) of each."""
        lists = self._mask_rows_host[j * k:(j + 1) * k]
        if k == 1:
            return lists[0], None
        keep = [(i, r) for i, r in enumerate(lists) if r.numel() > 0]
        if not keep:
            return lists[0], None
        rows = torch.cat([r + i * rows_per_mb for i, r in keep])
        w = torch.cat([torch.full((r.numel(),), 1.0 / r.numel(), dtype=torch.float32, device=r.device) for _, r in keep])
        return rows, w

    def _dense_logits_sparse_backward(self, x, y, mk, attn_mask, n_accum, k: int = 1):
        """lm_head_impl="dense": full logits in the forward, backward over the masked rows (see __init__)."""
        from . import ops
        from .model import _ReadoutRowsGradFn
        emb = self.model(x, attn_mask=attn_mask, return_embeddings=True)
        core = self.model.module if hasattr(self.model, "module") else self.model
        rows, weights = self._pass_rows(self._mb, k, self.mini * x.shape[1])
        with torch.no_grad():
            logits = core.lm_head(emb)                     # (B, T, V): every position, as model.py:253 computes them
            if rows.numel() == 0:
                loss, dl = None, None
            else:
                loss, dl = ops.masked_ce_rows(logits, y.reshape(-1), rows, n_accum, row_weights=weights)
        del logits
        self._order_backward(whole_pass=dl is None)
        if dl is None:    # nothing masked: zero gradients for every parameter (the reference would produce 0/0 = NaN here)
            (emb.sum() * 0 + core.lm_head.weight.sum() * 0).backward()
            return torch.zeros((), dtype=torch.float32, device=x.device)
        emb_rows = emb.reshape(-1, emb.shape[-1]).index_select(0, rows)
        wm = float(core.lm_head.output_mult) / float(core.lm_head.width_mult())
        _ReadoutRowsGradFn.apply(emb_rows, core.lm_head.weight, wm, dl).backward()
        return loss.detach()

    def _masked_rows_loss_backward(self, x, y, mk, attn_mask, n_accum, k: int = 1):
        """lm_head_impl="masked" (SURVEY.md §8f rank 1): readout + CE on the masked rows only, forward included.  The row
        indices come from the host-side MLM draw (no device sync); logits exist for the listed rows alone ([n_masked, V]),
        the CE kernel turns them into d(logits) rows, and the two backward products are those of the "dense" path."""
        from . import ops
        from .model import _ReadoutRowsGradFn
        core = self.model.module if hasattr(self.model, "module") else self.model
        rows, weights = self._pass_rows(self._mb, k, self.mini * x.shape[1])
        if rows.numel() == 0:
            # nothing masked in this pass: still hand EVERY parameter a (zero) gradient — lm_head included — or
            # DDP's reducer would wait for it forever when this is the synchronising micro-batch
            emb = self.model(x, attn_mask=attn_mask, return_embeddings=True)
            self._order_backward(whole_pass=True)
            (emb.sum() * 0 + core.lm_head.weight.sum() * 0).backward()
            return torch.zeros((), dtype=torch.float32, device=x.device)
        if self.rows_forward:
            # the model is told which positions are wanted: the last block's MLP half and ln_f run on them alone (model.forward(rows=))
            emb_rows = self.model(x, attn_mask=attn_mask, return_embeddings=True, rows=rows)
        else:
            emb = self.model(x, attn_mask=attn_mask, return_embeddings=True)
            emb_rows = emb.reshape(-1, emb.shape[-1]).index_select(0, rows)
        with torch.no_grad():
            logits = core.lm_head(emb_rows)                # (n_masked, V): model.py:253 on the rows the loss keeps (:304)
            tg = self._pass_targets(self._mb, k)
            loss, dl = ops.masked_ce_rows(logits, tg if tg is not None else y.reshape(-1).index_select(0, rows), None, n_accum, row_weights=weights)
        del logits
        self._order_backward()
        wm = float(core.lm_head.output_mult) / float(core.lm_head.width_mult())
        _ReadoutRowsGradFn.apply(emb_rows, core.lm_head.weight, wm, dl).backward()
        return loss.detach()

    def _host_prelude(self, input_ids, ids_host, rows, n_accum, want_rows):
        """The MLM corruption with every host-side input at hand (train_encoder.py:273-279): the Bernoulli draw AND the
        PAD/EOS exclusions are evaluated on the host copy of the batch, so the final mask — and from it the per-micro-batch
        lists of masked positions — is known without a device round trip.  Mask and lists go to the GPU through reused
        pinned buffers as asynchronous copies; the host never waits for the device here (the `.cpu()` of the device-side
        form stalls the host until the previous optimizer step has drained: 1.5-2 ms of idle GPU per step)."""
        dev = input_ids.device
        T = input_ids.shape[1]
        ids_h = np.asarray(ids_host)[:rows]
        draw = np.random.binomial(1, 0.15, (rows, T))                                  # same call, same stream as mlm_corrupt
        mask_h = (draw != 0) & (ids_h != PAD_TOKEN) & (ids_h != EOS_TOKEN)
        st = self._host_bufs.get((rows, T, dev))
        if st is None:
            st = dict(mask_pin=torch.empty((rows, T), dtype=torch.bool).pin_memory(), rows_pin=torch.empty(rows * T, dtype=torch.int64).pin_memory(),
                      mask_dev=torch.empty((rows, T), dtype=torch.bool, device=dev), rows_dev=torch.empty(rows * T, dtype=torch.int64, device=dev),
                      tgt_pin=torch.empty(rows * T, dtype=torch.int64).pin_memory(), tgt_dev=torch.empty(rows * T, dtype=torch.int64, device=dev),
                      done=None)
            if len(self._host_bufs) >= 4:          # --batch_ramp walks through many row counts: keep a few
                self._host_bufs.pop(next(iter(self._host_bufs)))
            self._host_bufs[(rows, T, dev)] = st
        if st["done"] is not None:
            st["done"].synchronize()        # the previous step's copies out of these pinned buffers (long finished)
        st["mask_pin"].numpy()[...] = mask_h
        st["mask_dev"].copy_(st["mask_pin"], non_blocking=True)
        lists = None
        if want_rows:
            per = mask_h.reshape(n_accum, -1)
            idx = [np.flatnonzero(per[j]) for j in range(n_accum)]
            sizes = [len(v) for v in idx]
            total = int(sum(sizes))
            if total:
                st["rows_pin"].numpy()[:total] = np.concatenate(idx)
                st["rows_dev"][:total].copy_(st["rows_pin"][:total], non_blocking=True)
                # the labels of those positions (the uncorrupted ids), in the same order: the compact CE needs no device-side gather
                per_ids = ids_h.reshape(n_accum, -1)
                st["tgt_pin"].numpy()[:total] = np.concatenate([per_ids[j][idx[j]] for j in range(n_accum)])
                st["tgt_dev"][:total].copy_(st["tgt_pin"][:total], non_blocking=True)
            off = np.concatenate([[0], np.cumsum(sizes)])
            lists = [st["rows_dev"][int(off[j]):int(off[j + 1])] for j in range(n_accum)]
            self._mask_targets_host = [st["tgt_dev"][int(off[j]):int(off[j + 1])] for j in range(n_accum)]
        st["done"] = torch.cuda.current_stream().record_event()
        mask = st["mask_dev"]
        return input_ids.masked_fill(mask, MASK_TOKEN), mask, lists

    def __call__(self, input_ids: torch.Tensor, mlm_mask: Optional[torch.Tensor] = None, input_ids_host=None) -> Dict[str, torch.Tensor]:
        """mlm_mask (optional, bool (rows, T)): the positions to corrupt instead of the host Bernoulli draw of
        train_encoder.py:273-274 (PAD/EOS are still excluded) — lets tests hand two runs the same corruption.
        input_ids_host (optional, the same batch as a host array/tensor, e.g. what the loader produced before its H2D copy):
        lets the step form the MLM mask without waiting for the device (see _host_prelude); results are identical.
        NEEDED for the sync-free path of the default readout (lm_head_impl="dense" / "masked" list the masked rows per
        micro-batch): without it the final mask comes back from the device once per optimizer step — a blocking D2H copy that
        waits for the previous step to drain (a one-time warning says so)."""
        rows = input_ids.shape[0] // self.mini * self.mini
        input_ids = input_ids[:rows]
        n_accum = rows // self.mini
        self.optimizer.zero_grad(set_to_none=True)
        # the two row-compact readouts need the HIP kernels; a stub model / torch loss (CPU tests) takes the generic graph
        sparse_rows = self.lm_head_impl in ("dense", "masked") and self.loss_impl == "fused" and self.fused_loss_fn is None
        self._mask_targets_host = None   # set by the host prelude when it runs (labels of the masked positions)
        if mlm_mask is None and input_ids_host is not None and input_ids.is_cuda:
            masked_ids, mask, lists = self._host_prelude(input_ids, input_ids_host, rows, n_accum, sparse_rows)
            if sparse_rows:
                self._mask_rows_host = lists
        else:
            if mlm_mask is None:
                masked_ids, mask = mlm_corrupt(input_ids)
            else:
                mask = mlm_mask[:rows].to(input_ids.device) & (input_ids != PAD_TOKEN) & (input_ids != EOS_TOKEN)
                masked_ids = input_ids.masked_fill(mask, MASK_TOKEN)
            if sparse_rows:
                # per-micro-batch row indices of the masked positions; mlm_corrupt drew the mask on the host, but PAD/EOS
                # exclusions were applied on the device, so fetch the final mask once per optimizer step (one small D2H copy)
                if input_ids.is_cuda and mlm_mask is None and not getattr(TrainStep, "_warned_no_host_copy", False):
                    TrainStep._warned_no_host_copy = True
                    import warnings
                    warnings.warn("TrainStep: no input_ids_host given — the masked-row lists of the default readout path need a "
                                  "blocking device-to-host copy per optimizer step (pass the loader's host copy of the batch to avoid it)")
                mh = mask.reshape(rows // self.mini, -1).cpu()
                self._mask_rows_host = [torch.nonzero(mh[j], as_tuple=False).reshape(-1).to(input_ids.device) for j in range(mh.shape[0])]
        dtype = next(self.model.parameters()).dtype
        core_model = self.model.module if hasattr(self.model, "module") else self.model
        k = self.per_pass if (sparse_rows and n_accum % self.per_pass == 0) else 1
        n_pass, span = n_accum // k, k * self.mini          # passes through the model, rows per pass
        emb_orders = None
        if input_ids.is_cuda and self.loss_impl == "fused" and self.fused_loss_fn is None and hasattr(core_model, "transformer"):
            # the embedding backward sums gradient rows in sorted-token order: ONE segmented sort for all micro-batches of
            # the step instead of a radix sort (four launches) per micro-batch
            emb_orders = torch.sort(masked_ids.reshape(n_pass, -1), dim=1, stable=True).indices.to(torch.int32)
        cum_loss = torch.zeros((), dtype=torch.float32, device=input_ids.device)
        from . import masks
        self._all_ranges = masks.RangeMask.from_tokens(input_ids, padding=self.use_padding, group=self.mini).key_ranges
        pipelined = (self.pipeline_streams >= 2 and input_ids.is_cuda and self.loss_impl == "fused" and n_pass > 2
                     and not self.sync_every)
        main = torch.cuda.current_stream() if input_ids.is_cuda else None
        if pipelined:
            if self._streams is None or len(self._streams) != self.pipeline_streams:
                self._streams = [torch.cuda.Stream() for _ in range(self.pipeline_streams)]
                # gradients are produced on the side streams by design; the engine's cross-stream sync is what we want
                warn_off = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
                if warn_off is not None:
                    warn_off(False)
            for st in self._streams:
                st.wait_stream(main)
        ns = self.pipeline_streams if pipelined else 1
        partial = [cum_loss] + [torch.zeros_like(cum_loss) for _ in range(ns - 1)]
        self._prev_bwd_done = None
        self._prev_order_events = None
        for j in range(n_pass):
            self._mb = j
            x = masked_ids[j * span:(j + 1) * span]
            y = input_ids[j * span:(j + 1) * span]
            last = j == n_pass - 1
            # the last pass of a DDP-wrapped model (the reducer's hooks and bucket all-reduces) runs on the caller's stream, after
            # everything; without a reducer it is a pass like the others and the caller's stream joins the side streams afterwards
            isolate_last = last and hasattr(self.model, "no_sync")
            side = pipelined and not isolate_last
            if pipelined and isolate_last:
                # its forward still runs beside the previous backward passes (it only reads weights); the caller's stream joins
                # the side streams right before its backward (_order_backward)
                self._join_side_streams = True
                self._prev_bwd_done = None
            self._slot = (j % ns) if side else 0
            self._order = None
            if side and self.backward_order == "layer" and self.fused_loss_fn is None and not self._no_inplace():
                from .model import BackwardOrder
                self._order = BackwardOrder(self._prev_order_events, self._prev_bwd_done)
            with (torch.cuda.stream(self._streams[j % ns]) if side else contextlib.nullcontext()):
                attn_mask = self._mask(y, dtype, j, k)
                ctx = contextlib.nullcontext()
                if hasattr(self.model, "no_sync") and not last and not self.sync_every:
                    ctx = self.model.no_sync()
                if emb_orders is not None:    # this pass's slice of the step-wide id sort, for its embedding backward
                    from .model import embedding_order
                    ctx_order = embedding_order(emb_orders[j])
                else:
                    ctx_order = contextlib.nullcontext()
                # all but the last pass: nobody observes the per-micro-batch gradients, so the big matrices are
                # accumulated by the wgrad epilogues themselves (model.accumulate_grads_inplace)
                # LayerNorm weight gradients: pass 0 delivers through autograd (there is no .grad yet), 1 .. n-2 carry
                # fp32 partial sums (first / more), the last one folds them in and delivers the total through autograd
                ln_mode = 0
                if n_pass > 2 and not self.sync_every and j >= 1:
                    ln_mode = 1 if j == 1 else (3 if last else 2)
                # (the last pass too when no reducer is attached: nothing observes its gradients before the optimizer does)
                with ctx, ctx_order, self._inplace(not (last and hasattr(self.model, "no_sync")) and not self.sync_every, ln_mode):
                    mk = mask[j * span:(j + 1) * span]
                    if sparse_rows and self.lm_head_impl == "masked":
                        partial[self._slot] += self._masked_rows_loss_backward(x, y, mk, attn_mask, n_accum, k)
                    elif sparse_rows:
                        partial[self._slot] += self._dense_logits_sparse_backward(x, y, mk, attn_mask, n_accum, k)
                    else:
                        logits = self.model(x, attn_mask=attn_mask)
                        partial[self._slot] += self._loss_backward(logits, y, mk, n_accum)
                        del logits
                if side:
                    self._prev_bwd_done = torch.cuda.current_stream().record_event()
                    self._prev_order_events = self._order.events if self._order is not None else None
        if input_ids.is_cuda and hasattr(self.model, "_obte_timed_hook"):   # measurement only (comm.TimedHook): the step's last backward ends here
            self.backward_end_event = torch.cuda.Event(enable_timing=True)
            self.backward_end_event.record()
        if pipelined:
            for st in self._streams:   # (a no-op after an isolated last pass: that one already waited for them)
                main.wait_stream(st)
            cum_loss = partial[0] + partial[1]
            for extra in partial[2:]:
                cum_loss = cum_loss + extra
        if isinstance(self.optimizer, FusedAdamW):
            self.optimizer.step(max_norm=self.max_grad_norm)
        else:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.max_grad_norm)
            self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()
        tokens = (input_ids != PAD_TOKEN).sum()
        return {"loss": cum_loss, "tokens": tokens}


# ------------------------------------------------------------------------------------------------- model set-up
def set_dropout(model, p: float) -> None:
    """Change the dropout probability of a built model (config.dropout feeds four sites: model.py:83-84,160,204)."""
    core = model.module if hasattr(model, "module") else model
    core.config.dropout = p
    core.transformer.drop.p = p
    for blk in core.transformer.h:
        blk.attn.dropout = p
        blk.attn.attn_dropout.p = p
        blk.attn.resid_dropout.p = p
        blk.mlp.dropout.p = p


def build_model(args, device, dtype=torch.bfloat16, vocab_size: int = 2 ** 16):
    """train_encoder.py:145-170: target model, muP base/delta models sharing ONE mutated config object,
    set_base_shapes, cast, move."""
    from .model import OmniBioTA, OmniBioTAConfig
    try:
        from mup import set_base_shapes  # type: ignore
    except Exception:
        from .mup_compat import set_base_shapes
    config = OmniBioTAConfig()
    config.vocab_size = vocab_size
    config.dropout = args.dropout
    config.block_size = args.ctx_len
    config.n_embd = args.n_embd
    config.n_layer = args.n_layer
    config.n_head = args.n_head
    config.flash = not args.disable_flash
    config.checkpoint_freq = args.checkpoint_freq
    m = OmniBioTA(config)
    config.n_embd, config.n_head = 24, 3
    base_model = OmniBioTA(config)
    config.n_embd, config.n_head = 48, 12
    delta_model = OmniBioTA(config)
    set_base_shapes(m, base_model, delta=delta_model)
    del base_model, delta_model
    config.n_embd, config.n_head = args.n_embd, args.n_head   # the reference leaves 48/12 behind; nothing reads it
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")   # "Casting complex values to real": the reference's cos-only RoPE regime
        m.to(dtype)
    m.to(device)
    return m


def build_optimizer(model, args, total_iters: int, fused: bool = True):
    """train_encoder.py:195-201."""
    lr = args.lr * np.sqrt(args.batch_size) / 32
    params = list(model.parameters())
    betas = (args.beta1, args.beta2)
    if args.force_lr:
        groups = [{"params": params}]
    else:
        from .mup_compat import mu_param_groups
        groups = mu_param_groups(params, lr, args.weight_decay)
    if fused:
        opt = FusedAdamW(groups, lr=lr, betas=betas, eps=args.epsilon, weight_decay=args.weight_decay)
    else:
        opt = torch.optim.AdamW(groups, lr=lr, betas=betas, eps=args.epsilon, weight_decay=args.weight_decay)
    sched = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1.0, end_factor=0.0, total_iters=max(total_iters, 1))
    return opt, sched


def flops_per_token(num_model_params: int, n_layer: int, n_embd: int, ctx_len: int) -> float:
    return 6.0 * num_model_params + 12.0 * n_layer * n_embd * ctx_len   # train_encoder.py:360


def flops_per_token_executed(num_model_params: int, n_layer: int, n_embd: int, ctx_len: int, lm_head_impl: str = "masked",
                             rows_forward: bool = True, vocab: int = 2 ** 16, masked_fraction: float = 0.15,
                             attention_fraction: float = 1.0, rows_attention: Optional[bool] = None) -> float:
    """FLOP per token the step actually EXECUTES: the reference's 6N + 12LCT minus the products the readout form leaves out on
    the (1 - masked_fraction) of the positions the loss multiplies by zero (train_encoder.py:304) — "dense": the two backward
    products of the readout; "masked": all three, and with rows_forward the last block's MLP half (8 C^2 parameters) as well;
    rows_attention (default: as rows_forward; the library takes this form without a dense mask, csrc/block.cpp rows_attn): also the
    last block's attention projection and the q third of its c_attn (C^2 parameters each) and its attention for the queries at
    those positions — keys and values of every position are still formed."""
    skip = 1.0 - masked_fraction
    skipped = {"dense": 4.0 * n_embd * vocab * skip, "dense_full": 0.0, "masked": 6.0 * n_embd * vocab * skip}[lm_head_impl]
    if lm_head_impl == "masked" and rows_forward:
        skipped += 6.0 * 8.0 * n_embd ** 2 * skip
        if rows_attention is None or rows_attention:
            skipped += (2 * 6.0 * n_embd ** 2 + 12.0 * n_embd * ctx_len * attention_fraction) * skip   # c_proj and c_attn's q third: C^2 parameters each
    # attention_fraction < 1 (rows that pack several documents): the share of the 12 L C T attention term the kernels visit
    skipped += 12.0 * n_layer * n_embd * ctx_len * (1.0 - attention_fraction)
    return flops_per_token(num_model_params, n_layer, n_embd, ctx_len) - skipped


def wrap_ddp(model, device_index: Optional[int], bucket_cap_mb: int = 100, grad_exchange: str = "allreduce", timed: bool = False):
    """DDP over RCCL.  Buckets of ~100 MB (a few per all-reduce of small's 470 MB) keep each ring/tree step long
    enough to run at xGMI link rate while still overlapping the tail of backward; gradients are views into the
    buckets, so the reducer never copies.
    grad_exchange: "allreduce" — DDP's own bucketed all-reduce (train_encoder.py:185 as the reference runs it); "all_links" — the
    direct reduce-scatter + all-gather of comm.AllLinksHook (every xGMI link of the mesh at once, fp32 fixed-order reduction: SURVEY
    section 5).  timed: wrap the exchange in comm.TimedHook (per-bucket device events; the wrapper hangs on the returned module as
    ``_obte_timed_hook``) — measurement only."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    from . import comm
    assert grad_exchange in ("allreduce", "all_links"), grad_exchange
    kw = dict(bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True, broadcast_buffers=False)
    ddp = DDP(model, **kw) if device_index is None else DDP(model, device_ids=[device_index], **kw)
    hook = comm.AllLinksHook() if grad_exchange == "all_links" else (comm.allreduce_mean_hook() if timed else None)
    if timed and hook is not None:
        hook = comm.TimedHook(hook)
        ddp._obte_timed_hook = hook
    if hook is not None:
        ddp.register_comm_hook(None, comm.as_ddp_hook(hook, f"obte_{grad_exchange}{'_timed' if timed else ''}_hook"))
    return ddp


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="OmniBioTE masked-LM training on MI355X (synthetic data)")
    p.add_argument("--batch_size", type=int, default=1024)
    p.add_argument("--mini_batch_size", type=int, default=8)
    p.add_argument("--n_head", type=int, default=8)
    p.add_argument("--n_embd", type=int, default=1024)
    p.add_argument("--n_layer", type=int, default=8)
    p.add_argument("--ctx_len", type=int, default=2048)
    p.add_argument("--dropout", type=float, default=0.1)
    p.add_argument("--lr", type=float, default=1e-2)
    p.add_argument("--beta1", type=float, default=0.9)
    p.add_argument("--beta2", type=float, default=0.999)
    p.add_argument("--epsilon", type=float, default=1e-8)
    p.add_argument("--weight_decay", type=float, default=1e-2)
    p.add_argument("--token_budget", type=float, default=20e9)
    p.add_argument("--test_freq", type=int, default=int(1e7))
    p.add_argument("--save_freq", type=int, default=int(1e9))
    p.add_argument("--save_name", type=str, default="omnibiota")
    p.add_argument("--disable_flash", action="store_true", default=False)
    p.add_argument("--wandb_project_name", type=str, default="omnibiota")
    p.add_argument("--base_dir", type=str, default="")
    p.add_argument("--force_lr", action="store_true", default=False)
    p.add_argument("--checkpoint_freq", type=int, default=0)
    p.add_argument("--banned_token", type=int, default=BANNED_MIXED)
    p.add_argument("--warmup_period", type=float, default=0.05)
    p.add_argument("--batch_ramp", action="store_true", default=False)
    p.add_argument("--train_type", type=str, default="mixed")
    p.add_argument("--FSDP", action="store_true", default=False)
    p.add_argument("--use_padding", action="store_true", default=False)
    p.add_argument("--resume_from", type=int, default=0)
    # additions
    p.add_argument("--max_steps", type=int, default=0, help="stop after this many optimizer steps (0 = token budget)")
    p.add_argument("--multi_document", action="store_true", default=False, help="synthetic rows with interior EOS")
    p.add_argument("--pipeline_streams", type=int, default=2, choices=[1, 2, 3],
                   help="2: overlap the forward of micro-batch j+1 with the backward of micro-batch j (same results)")
    p.add_argument("--micro_batches_per_pass", type=int, default=1,
                   help="k > 1: k micro-batches of --mini_batch_size rows per forward/backward pass (masks and loss normalisation stay per "
                        "micro-batch: same loss and gradients; +4-6 %% at k = 2-4 on the small config, bench.py chooses it by measurement)")
    p.add_argument("--grad_exchange", default="allreduce", choices=["allreduce", "all_links"],
                   help="allreduce: DDP's bucketed all-reduce as the reference runs it; all_links: direct reduce-scatter + all-gather over "
                        "every xGMI link of the node at once (comm.AllLinksHook; fp32 fixed-order reduction, one rounding)")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="torch.distributed backend: nccl (= RCCL over xGMI, the production path) or gloo (plumbing runs: "
                        "BASELINE config 1; gradients then cross the host)")
    p.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                   help="cpu only exercises the harness plumbing: the model itself has no CPU path and says so at the first forward")
    return p.parse_args(argv)


def effective_batch(i: int, total_iters: int, args, batch_size: int) -> int:
    """Batch-size ramp of train_encoder.py:245-255."""
    mini = args.mini_batch_size
    if args.batch_ramp:
        e = min((int(i / (total_iters * args.warmup_period) * batch_size) // mini) * mini + mini, batch_size)
    else:
        e = batch_size
    return e // mini * mini


TRAIN_TYPES = {   # train_encoder.py:72-93
    "protein": (["uniref100/train"], [1.0]),
    "nucleotide": (["genbank/train"], [1.0]),
    "mixed": (["genbank/train", "uniref100/train"], [0.80, 0.20]),
    "halfnhalf": (["genbank/train", "uniref100/train"], [0.50, 0.50]),
}
TEST_SETS = {     # train_encoder.py:72-93: (test_dirs, test_names)
    "protein": (["uniref100/val"], ["uniref100"]),
    "nucleotide": (["genbank/val"], ["genbank"]),
    "mixed": (["genbank/val", "uniref100/val"], ["genbank", "uniref100"]),
    "halfnhalf": (["genbank/val", "uniref100/val"], ["genbank", "uniref100"]),
}


def make_batch_source(args, batch_size: int, device, rng):
    """``(next_batch, description, close)``: ``next_batch(rows) -> LongTensor (rows, ctx_len)`` on ``device``; real shards
    when --base_dir has them, else synthetic.  ``close()`` stops and joins the loader thread (call it before tearing the
    process group down)."""
    if args.train_type not in TRAIN_TYPES:
        raise ValueError("Invalid train_type. Must be one of 'protein', 'nucleotide', 'mixed', or 'halfnhalf'")
    dirs, props = TRAIN_TYPES[args.train_type]
    dirs = [os.path.join(args.base_dir, d) for d in dirs] if args.base_dir else []
    if dirs and all(os.path.isdir(d) and os.listdir(d) for d in dirs):
        import queue
        import threading
        from . import loader as LD
        files = [sorted(os.path.join(d, f) for f in os.listdir(d)) for d in dirs]
        readers = [LD.line_reader(f, banned_tokens=[args.banned_token]) for f in files]
        gens = [LD.get_sequence(r, args.ctx_len, args.use_padding) for r in readers]
        batches = LD.get_batch(gens, LD.batch_split(batch_size, props), return_pt=True)
        q = queue.Queue(maxsize=2)                                   # train_encoder.py:140-142
        stop = threading.Event()
        th = threading.Thread(target=LD.data_loader_parallel, args=(q, batches, device, stop), daemon=True)
        th.start()
        pool = [q.get(block=True)]

        def real(rows):
            while sum(b.shape[0] for b in pool) < rows:              # the reference's grand_batch top-up (:258-261)
                pool.append(q.get(block=True))
            hosts = [getattr(b, "_obte_host_copy", None) for b in pool]
            cat = torch.cat(pool, dim=0) if len(pool) > 1 else pool[0]
            host = None if any(h is None for h in hosts) else (np.concatenate(hosts) if len(hosts) > 1 else hosts[0])
            rest, out = cat[rows:], cat[:rows]
            if host is not None:      # the host copies travel with the rows they belong to
                rest._obte_host_copy, out._obte_host_copy = host[rows:], host[:rows]
            pool[:] = [rest]
            return out

        def close():
            stop.set()
            while th.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.1)
        return real, "shards under " + args.base_dir, close
    if args.base_dir:
        print(f"note: no token shards under {args.base_dir}; training on synthetic rows")

    def synth(rows):
        host = synthetic_rows(rows, args.ctx_len, 2 ** 16, rng, single_document=not args.multi_document)
        dev_t = torch.from_numpy(host).to(device)
        dev_t._obte_host_copy = host      # lets TrainStep form the MLM mask without a device round trip
        return dev_t
    return synth, "synthetic rows", (lambda: None)


def make_test_sources(args, device, rng):
    """Held-out generators for the periodic evaluation (train_encoder.py:131-133): ``[(name, next_rows)]`` where
    ``next_rows(n) -> LongTensor (n, ctx_len)``.  The ``*/val`` shard directories when --base_dir has them, else one
    synthetic stream from its own RNG."""
    dirs, names = TEST_SETS[args.train_type]
    dirs = [os.path.join(args.base_dir, d) for d in dirs] if args.base_dir else []
    out = []
    if dirs and all(os.path.isdir(d) and os.listdir(d) for d in dirs):
        from . import loader as LD
        for d, name in zip(dirs, names):
            files = sorted(os.path.join(d, f) for f in os.listdir(d))
            gen = LD.get_sequence(LD.line_reader(files, banned_tokens=[args.banned_token]), args.ctx_len, args.use_padding)

            def rows_from(n, gen=gen):
                arr = np.concatenate([np.asarray(next(gen)).reshape(1, -1) for _ in range(n)])   # :376-379
                return torch.as_tensor(arr, dtype=torch.long, device=device)
            out.append((name, rows_from))
        return out
    trng = np.random.default_rng(int(rng.integers(1 << 31)) + 7919)

    def synth(n):
        return torch.from_numpy(synthetic_rows(n, args.ctx_len, 2 ** 16, trng, single_document=not args.multi_document)).to(device)
    return [("synthetic", synth)]


@torch.no_grad()
def evaluate(model, test_sources, args, world: int, device) -> Dict[str, float]:
    """The held-out pass of train_encoder.py:371-410: model.eval(), one mini-batch per test set, the same MLM corruption
    and attention mask as training, loss = sum_masked(CE) / n_masked / world on every rank, summed over ranks."""
    from . import masks, ops
    core = model.module if hasattr(model, "module") else model
    was_training = core.training
    model.eval()
    out = {}
    for name, next_rows in test_sources:
        test_batch = next_rows(args.mini_batch_size)
        masked_ids, mask = mlm_corrupt(test_batch)
        attn = masks.RangeMask.from_tokens(test_batch, padding=args.use_padding)
        logits = model(masked_ids, attn_mask=attn)
        if logits.is_cuda and logits.dtype == torch.bfloat16:
            loss, _ = ops.masked_ce(logits, test_batch, mask, 1)
        else:
            ce = F.cross_entropy(logits.view(-1, logits.size(-1)).float(), test_batch.view(-1), reduction="none")
            loss = (ce * mask.view(-1).float()).sum() / mask.view(-1).sum()
        loss = loss.float() / world
        if world > 1:
            dist.all_reduce(loss)
        out[name] = float(loss.item())
    model.train(was_training)
    return out


def run(args):
    if args.FSDP:
        raise NotImplementedError("--FSDP is outside this build's scope (the north star names DDP)")
    backend = getattr(args, "backend", "nccl")
    on_gpu = getattr(args, "device", "cuda") == "cuda"
    if backend == "nccl" and not on_gpu:
        raise SystemExit("--backend nccl (RCCL) needs --device cuda; use --backend gloo for a CPU plumbing run")
    if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:   # a bare `python train_encoder.py` is a world of one;
        os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "1"         # a launcher that set only one of the two is a mis-launch
    elif "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        raise SystemExit("train_encoder: RANK and WORLD_SIZE must both be set by the launcher (or neither, for a single process)")
    for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29511")):
        os.environ.setdefault(k, v)
    # torchrun sets LOCAL_RANK; srun / mpirun style launchers set only RANK: then bind like the reference does,
    # rank % GPUs on the node (train_encoder.py:110-111)
    local = int(os.environ.get("LOCAL_RANK", int(os.environ["RANK"]) % max(torch.cuda.device_count(), 1) if on_gpu else 0))
    if on_gpu:
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    dist.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))   # nccl = RCCL on ROCm
    rank, world = dist.get_rank(), dist.get_world_size()
    assert args.batch_size % world == 0, "Batch size must be divisible by the number of processes."
    batch_size = args.batch_size // world
    np.random.seed(1234 + rank)
    torch.manual_seed(1234)   # identical initial weights on every rank (DDP broadcasts rank 0's anyway)
    m = build_model(args, device)
    if args.resume_from > 0:   # train_encoder.py:174-178
        from .checkpoint import load_checkpoint
        m = load_checkpoint(f"{args.save_name}_{args.resume_from}.pt", map_location=device)
        m.to(torch.bfloat16).to(device)
        print(f"Loaded model from {args.resume_from} token checkpoint")
    torch.manual_seed(1234 + rank)   # per-rank dropout streams from here on
    n_params = m.get_num_params()
    if on_gpu:
        from . import tune
        tune.tune_model_shapes(max(1, getattr(args, "micro_batches_per_pass", 1)) * args.mini_batch_size * args.ctx_len, args.n_embd, 2 ** 16,
                               device=device, verbose=(rank == 0))
        if world > 1:   # every rank adopts rank 0's plan table: the replicas then run the same kernels
            box = [tune.export_plans() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            if rank != 0:
                tune.import_plans(box[0])
    model = wrap_ddp(m, local if on_gpu else None, grad_exchange=getattr(args, "grad_exchange", "allreduce")) if world > 1 else m
    total_iters = int(args.token_budget / (world * batch_size * args.ctx_len))
    opt, sched = build_optimizer(m, args, total_iters, fused=on_gpu)
    step = TrainStep(model, opt, sched, mini_batch_size=args.mini_batch_size, n_head=args.n_head, use_padding=args.use_padding,
                     pipeline_streams=getattr(args, "pipeline_streams", 1), micro_batches_per_pass=getattr(args, "micro_batches_per_pass", 1))
    rng = np.random.default_rng(1234 + rank)
    next_batch, source, close_source = make_batch_source(args, batch_size, device, rng)
    test_sources = make_test_sources(args, device, rng)
    if rank == 0:
        print(f"data: {source}; backend {backend}, world {world}, device {device}")
    trained, last_save, last_test, start = 0, 0, 0, 0
    if args.resume_from > 0:   # train_encoder.py:210-223 (optimizer/scheduler as state_dicts, not whole objects)
        trained = last_save = last_test = args.resume_from
        start = int(total_iters * (trained / args.token_budget))
        st = torch.load(f"{args.save_name}_optimizer_{args.resume_from}.pt", map_location=device, weights_only=False)
        opt.load_state_dict(st["optimizer"]); sched.load_state_dict(st["scheduler"])
    fpt = flops_per_token(n_params, args.n_layer, args.n_embd, args.ctx_len)   # the reference's MFU formula (:360)
    # what this step executes: its default readout runs on the ~15 % MLM-masked positions only (fewer FLOP, same gradients)
    fpt_exec = flops_per_token_executed(n_params, args.n_layer, args.n_embd, args.ctx_len, step.lm_head_impl, step.rows_forward,
                                        rows_attention=step.rows_forward and step.mask_impl == "ranges")
    n_steps = total_iters if args.max_steps <= 0 else min(total_iters, start + args.max_steps)

    def save(tag):
        from .checkpoint import save_checkpoint
        save_checkpoint(model, f"{args.save_name}{tag}.pt")
        torch.save({"optimizer": opt.state_dict(), "scheduler": sched.state_dict()}, f"{args.save_name}_optimizer{tag}.pt")

    history = []
    try:
        for i in range(start, n_steps):
            t0 = time.time()
            rows = effective_batch(i, total_iters, args, batch_size)
            ids = next_batch(rows)
            out = step(ids, input_ids_host=getattr(ids, "_obte_host_copy", None))
            stats = torch.stack([out["loss"], out["tokens"].float()])
            if world > 1:
                dist.all_reduce(stats)
            if on_gpu:
                torch.cuda.synchronize()
                from . import _lib
                _lib.check_device_status(f"step {i}")   # a kernel that found its own results invalid says so here, not never
            dt = time.time() - t0
            loss, toks = stats[0].item() / world, int(stats[1].item())
            trained += toks
            history.append(loss)
            if rank == 0:
                lrs = [g["lr"] for g in opt.param_groups]
                print(f"step {i} loss {loss:.4f} lr {lrs[0]:.5f}|{lrs[-1]:.5f} tokens/s {toks / dt:,.0f} "
                      f"MFMA-frac(executed FLOP) {toks / dt * fpt_exec / (2.5e15 * world) * 100:.1f}% "
                      f"[reference formula 6N+12LCT: {toks / dt * fpt / (2.5e15 * world) * 100:.1f}%] trained {trained / 1e6:.2f}M", flush=True)
            if trained - last_test > args.test_freq:            # train_encoder.py:371-410
                for name, v in evaluate(model, test_sources, args, world, device).items():
                    if rank == 0:
                        print(f"test_loss/{name} {v:.4f} at {trained / 1e6:.2f}M tokens", flush=True)
                last_test = trained
            if rank == 0 and trained - last_save > args.save_freq:   # train_encoder.py:412-423: keep only the newest
                save(f"_{trained}")
                if last_save > 0:
                    for f in (f"{args.save_name}_{last_save}.pt", f"{args.save_name}_optimizer_{last_save}.pt"):
                        if os.path.exists(f):
                            os.remove(f)
                last_save = trained
        if rank == 0 and args.save_name:
            save("")                                        # train_encoder.py:429-432
    finally:
        close_source()
        dist.destroy_process_group()
    return history


if __name__ == "__main__":
    run(parse_args())
