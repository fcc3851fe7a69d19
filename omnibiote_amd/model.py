"""Drop-in replacement for the reference's ``training/model.py`` class surface, computed by hand-written HIP.

Same names, constructor RNG order, attribute tree and state_dict keys as the reference
(training/model.py:63-277), so ``train_encoder.py`` and the eval scripts can ``from model import OmniBioTA,
OmniBioTAConfig`` unchanged (``training/model.py`` in this repo re-exports this module).  What differs is what
runs underneath: every block forward/backward is one call into ``libomnibiote_hip.so``.

Contract notes (each mirrors a reference behaviour):
  * ``OmniBioTAConfig`` has no ``flash`` field; callers assign it ad hoc (train_encoder.py:152).  Reading it
    when absent raises AttributeError exactly like the reference.
  * ``freqs_cis`` is a persistent complex64 buffer.  ``module.to(torch.bfloat16)`` turns it into a real bf16
    tensor holding cos only (PyTorch semantics), and the reference then *scales* pairs instead of rotating
    them.  This module reproduces both modes from the buffer's dtype: complex -> rotation, real -> cos-only.
  * softmax scale is 8 / n_embd (model.py:119), GELU uses erf(x / 1.41421) (model.py:25), LayerNorm eps 1e-5.
  * The HIP path computes in bf16 with fp32 accumulation — the regime the reference trains and evaluates in
    (train_encoder.py:21,170).  Parameters must be bf16 on a GPU at forward time; anything else raises.
  * dropout (embedding, attention probabilities, both residual projections; model.py:83-84,160,204) is fused
    into the kernels with a counter-based mask: same distribution and scaling as nn.Dropout, but necessarily a
    different random stream than PyTorch's generator.  Each forward draws its seeds from torch's CPU generator, so
    ``torch.manual_seed`` makes runs reproducible and activation checkpointing recomputes identical masks.
"""
from __future__ import annotations

import weakref
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from . import _lib as L
from . import ops

try:  # the real package, if the environment has it (README.md:16 pins mup==1.0.0)
    from mup import MuReadout as _MuReadoutBase  # type: ignore
    _HAVE_MUP = True
except Exception:  # not installed here: restated semantics, see mup_compat.py
    from .mup_compat import MuReadout as _MuReadoutBase
    _HAVE_MUP = False


# ------------------------------------------------------------------------------------------------ helper functions
def fused_gelu(x: torch.Tensor) -> torch.Tensor:
    """x * 0.5 * (1 + erf(x / 1.41421))  (model.py:23-25).  Tensor-level helper kept for API parity; inside the
    block the same formula runs in the c_fc GEMM epilogue."""
    return x * 0.5 * (1.0 + torch.erf(x / 1.41421))


def precompute_freqs_cis(dim: int, end: int, theta: float = 10000.0) -> torch.Tensor:
    """complex64 (end, dim/2) table of unit phasors exp(i t theta^(-2j/dim))  (model.py:53-61)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
    angles = torch.outer(torch.arange(end), inv_freq).float()
    return torch.polar(torch.ones_like(angles), angles)


def reshape_for_broadcast(freqs_cis: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """(model.py:28-35) view the first x.shape[1] rows of the table as (1, T, 1, hs/2)."""
    assert freqs_cis.shape[-1] == x.shape[-1]
    f = freqs_cis[: x.shape[1]]
    return f.view(*[d if i in (1, x.ndim - 1) else 1 for i, d in enumerate(x.shape)])


def rope_tables(freqs_cis: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """fp32 (cos, sin) tables the kernels consume.  A real-valued buffer (after ``.to(bfloat16)``) yields
    sin = 0: the reference's degenerate cos-only scaling, including the bf16 rounding of cos."""
    if freqs_cis.is_complex():
        return freqs_cis.real.float().contiguous(), freqs_cis.imag.float().contiguous()
    c = freqs_cis.float().contiguous()
    return c, torch.zeros_like(c)


def apply_rotary_emb(xq: torch.Tensor, xk: torch.Tensor, freqs_cis: torch.Tensor):
    """(model.py:39-50) on (B, T, H, hs) tensors; runs the HIP RoPE kernel on a packed copy."""
    B, T, H, hs = xq.shape
    cos, sin = rope_tables(freqs_cis.to(xq.device))
    packed = torch.cat([xq.reshape(B, T, H * hs), xk.reshape(B, T, H * hs), torch.zeros_like(xq).reshape(B, T, H * hs)], dim=2)
    packed = packed.to(torch.bfloat16).contiguous()
    ops.rope_qk_(packed, cos, sin, B, T, H, hs)
    q, k, _ = packed.split(H * hs, dim=2)
    return q.reshape(B, T, H, hs).type_as(xq), k.reshape(B, T, H, hs).type_as(xk)


def _require_hip(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: the OmniBioTE MI355X path needs GPU tensors (got {t.device}); there is no CPU "
                           "fallback in this package. Move the model and inputs to the GPU, e.g. m.to(torch.bfloat16).to('cuda').")
    if t.dtype != torch.bfloat16:
        raise RuntimeError(f"{what}: parameters/activations must be torch.bfloat16 (the reference trains and evaluates "
                           f"in bf16: train_encoder.py:21,170); got {t.dtype}. Call model.to(torch.bfloat16).")
    L.lib()  # raises HipLibraryError if the shared library is missing


# ------------------------------------------------------------------------------------------ in-place grad accumulation
# How the backward of a micro-batch delivers its weight gradients is a property of the TRAINING STEP that built the graph,
# not of the process: each autograd node below captures the active ``GradPolicy`` snapshot at FORWARD time (``ctx.pol``) and
# its backward — which PyTorch runs on an autograd-engine thread — reads only that.  Two TrainSteps on two models in one
# process, or forward passes issued from different threads, therefore cannot see each other's switches (a context variable
# selects the snapshot; nothing here is a mutable module global).
#   accumulate: the backward of the big matrices adds straight into an existing ``param.grad`` (the wgrad GEMM's epilogue /
#       the embedding scatter do the read-modify-write) and hands autograd ``None`` for them, which removes the separate
#       ``grad += new`` pass over 470 MB per micro-batch.  The arithmetic is the one autograd would do (bf16(old +
#       bf16(new))).  Only valid while nothing needs to observe per-micro-batch gradients: the harness sets it for the
#       micro-batches that run under DDP's no_sync(), never for the last one (whose AccumulateGrad hooks feed the reducer).
#   ln_mode: LayerNorm weight gradients over micro-batches: instead of one reduction launch + one bf16 ``grad += new`` per
#       LayerNorm and micro-batch (17 LayerNorms x 16 micro-batches on the small config), the per-workgroup partial sums are
#       carried in a persistent fp32 buffer per weight (L.LN_PARTIAL_FIRST on the first such micro-batch, _MORE after it) and
#       reduced ONCE, by the micro-batch that runs with L.LN_PARTIAL_LAST and hands the total to autograd.  0 = off.  The
#       total is bf16(sum in fp32) — closer to the exact gradient than autograd's running bf16 sum, not bitwise equal to it.
#   store: who owns those fp32 buffers (one ``LnPartialStore`` per TrainStep; a process-wide default for bare uses of the
#       context manager).
import contextvars
import os


class LnPartialStore:
    """fp32 partial-sum buffers of LayerNorm weights, keyed by the parameter's identity (not an attribute of the parameter:
    whole-object pickles of the model — the reference's checkpoint format — must not carry 2 MB of scratch per weight)."""

    def __init__(self):
        self._bufs = {}   # id(weight) -> (weak reference to it, buffer)

    def get(self, param):
        key = id(param)
        ent = self._bufs.get(key)
        if ent is not None and ent[0]() is param and ent[1].device == param.device and \
                ent[1].numel() == L.lib().obte_layernorm_bwd_ws_rows() * param.numel():
            return ent[1]
        buf = ops.ln_partials_buffer(param.numel(), param.device)
        bufs = self._bufs
        self._bufs[key] = (weakref.ref(param, lambda _r, k=key: bufs.pop(k, None)), buf)
        return buf


class BackwardOrder:
    """Cross-stream ordering of the backward passes of consecutive micro-batches, PER PARAMETER GROUP instead of per pass.
    Backward passes that run on different HIP streams read-modify-write the same gradient buffers (in-place accumulation, the
    LayerNorm partial sums); what has to be ordered is each buffer's own sequence of updates, not the passes as wholes.  Every
    autograd node that owns parameters waits — on the stream it runs on — for the event the SAME node of the previous
    micro-batch recorded (``wait``), does its work, and records its own (``done``).  The backward of micro-batch j+1 can then
    follow that of micro-batch j one layer behind instead of starting after its last kernel; every buffer still sees the
    updates in micro-batch order, so results are bitwise those of one stream.  ``prev_events``: the ``events`` of the previous
    micro-batch's object (complete on the host by the time this pass's backward is issued: the harness issues backward passes
    in order); ``fallback``: an event that covers the whole previous backward, for a node the previous pass did not run."""
    __slots__ = ("prev_events", "fallback", "events")

    def __init__(self, prev_events=None, fallback=None):
        self.prev_events, self.fallback, self.events = prev_events, fallback, {}

    def wait(self, key):
        ev = None if self.prev_events is None else self.prev_events.get(key)
        if ev is None:
            ev = self.fallback
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def done(self, key):
        self.events[key] = torch.cuda.current_stream().record_event()


class GradPolicy:
    """Immutable snapshot of the switches above; what an autograd node keeps in ``ctx.pol``."""
    __slots__ = ("accumulate", "ln_mode", "store", "order")

    def __init__(self, accumulate: bool = False, ln_mode: int = 0, store: Optional[LnPartialStore] = None, order: Optional[BackwardOrder] = None):
        object.__setattr__(self, "accumulate", bool(accumulate))
        object.__setattr__(self, "ln_mode", int(ln_mode))
        object.__setattr__(self, "store", store)
        object.__setattr__(self, "order", order)

    def __setattr__(self, *a):
        raise AttributeError("GradPolicy is immutable: enter accumulate_grads_inplace(...) for different switches")


_NO_POLICY = GradPolicy()
_policy = contextvars.ContextVar("obte_grad_policy", default=_NO_POLICY)
_default_store = LnPartialStore()
_embedding_order = contextvars.ContextVar("obte_embedding_order", default=None)


def current_grad_policy() -> GradPolicy:
    """The snapshot an autograd node should capture in its forward (``ctx.pol = current_grad_policy()``)."""
    return _policy.get()


class accumulate_grads_inplace:
    """``with accumulate_grads_inplace(enabled, ln_partial_mode, store=...)``: graphs BUILT inside deliver their gradients
    that way when they run backward (inside the block or later, on whatever thread)."""

    def __init__(self, enabled: bool = True, ln_partial_mode: int = 0, store: Optional[LnPartialStore] = None, order: Optional[BackwardOrder] = None):
        self.pol = GradPolicy(enabled, ln_partial_mode, store if store is not None else _default_store, order)

    def __enter__(self):
        self.token = _policy.set(self.pol)
        return self

    def __exit__(self, *exc):
        _policy.reset(self.token)
        return False


class embedding_order:
    """``with embedding_order(order)``: the next OmniBioTA.forward in this context takes ``order`` (int32, a stable argsort of
    its flattened token ids) for its embedding backward instead of sorting the ids itself — a harness that sorts a whole
    optimizer step's ids in one call hands each micro-batch its slice this way.  Consumed by that forward."""

    def __init__(self, order):
        self.box = [order]

    def __enter__(self):
        self.token = _embedding_order.set(self.box)
        return self

    def __exit__(self, *exc):
        _embedding_order.reset(self.token)
        return False


def _ln_partials(param, pol: GradPolicy):
    """The persistent fp32 partial-sum buffer of a LayerNorm weight in the policy's store (created on first use)."""
    return (pol.store if pol.store is not None else _default_store).get(param)


def _grad_slot(param, pol: Optional[GradPolicy] = None):
    """The existing gradient of ``param`` if the node's policy says accumulate in place and that is applicable, else None.
    (The HIP entry points that receive it insist on bf16 themselves; the protocol is dtype-agnostic so that the CPU
    multi-process tests can drive it with a stub model.)"""
    if pol is None or not pol.accumulate:
        return None
    g = getattr(param, "grad", None)
    if g is None or g.dtype != param.dtype or not g.is_contiguous() or g.shape != param.shape:
        return None
    return g


def _ord_wait(pol: Optional[GradPolicy], key) -> None:
    if pol is not None and pol.order is not None:
        pol.order.wait(key)


def _ord_done(pol: Optional[GradPolicy], key) -> None:
    if pol is not None and pol.order is not None:
        pol.order.done(key)


# ------------------------------------------------------------------------------------------------- autograd glue
class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        y, mean, rstd = ops.layernorm_fwd(x.contiguous(), w)
        ctx.save_for_backward(x, w, mean, rstd)
        ctx.w_param = w
        ctx.pol = current_grad_policy()
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        pol = ctx.pol
        _ord_wait(pol, id(ctx.w_param))
        if pol.ln_mode:
            dx, dw = ops.layernorm_bwd(dy.contiguous(), x.contiguous(), w, mean, rstd, partials=_ln_partials(ctx.w_param, pol),
                                       partial_mode=pol.ln_mode)
        else:
            dx, dw = ops.layernorm_bwd(dy.contiguous(), x.contiguous(), w, mean, rstd, accumulate_into=_grad_slot(ctx.w_param, pol))
        _ord_done(pol, id(ctx.w_param))
        return dx, dw


class _LinearFn(torch.autograd.Function):
    """y = alpha * x W^T (bias-free nn.Linear; alpha = 1/width_mult for the muP readout)."""

    @staticmethod
    def forward(ctx, x, w, alpha):
        ctx.save_for_backward(x, w)
        ctx.alpha = alpha
        ctx.w_param = w
        ctx.pol = current_grad_policy()
        x2 = x.reshape(-1, x.shape[-1])
        y = ops.linear_fwd(x2.contiguous(), w, alpha=alpha)
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        dx = dw = None
        _ord_wait(ctx.pol, id(ctx.w_param))
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            dx, dw = ops.linear_bwd(dy2, x2, w, alpha=ctx.alpha, accumulate_into=_grad_slot(ctx.w_param, ctx.pol))
            _ord_done(ctx.pol, id(ctx.w_param))
            return dx.view_as(x), dw, None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_dgrad(dy2, w, alpha=ctx.alpha).view_as(x)
        if ctx.needs_input_grad[1]:
            slot = _grad_slot(ctx.w_param, ctx.pol)
            dw = ops.linear_wgrad(dy2, x2, alpha=ctx.alpha, accumulate_into=slot)
            if slot is not None:
                dw = None
        _ord_done(ctx.pol, id(ctx.w_param))
        return dx, dw, None


class _LinearGeluFn(torch.autograd.Function):
    """gelu_erf_1.41421(x W^T) (model.py:163-165) through the c_fc GEMM's fused epilogue — the same kernel, rounding and
    saved derivative as inside ``Block`` (one erf evaluation yields the activation and gelu'(h))."""

    @staticmethod
    def forward(ctx, x, w):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        der, act = ops.linear_fwd(x2, w, epilogue=L.EPI_GELU)
        ctx.save_for_backward(x, w, der)
        ctx.w_param = w
        ctx.pol = current_grad_policy()
        return act.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dact):
        x, w, der = ctx.saved_tensors
        dh = dact.reshape(-1, dact.shape[-1]) * der            # bf16(dact * gelu'(h)), as OBTE_EPI_GELU_BWD forms it
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        _ord_wait(ctx.pol, id(ctx.w_param))
        dx, dw = ops.linear_bwd(dh.contiguous(), x2, w, accumulate_into=_grad_slot(ctx.w_param, ctx.pol))
        _ord_done(ctx.pol, id(ctx.w_param))
        return dx.view_as(x), dw


class _ReadoutRowsGradFn(torch.autograd.Function):
    """Backward-only node of the readout for a loss that looks at a subset of the positions (the masked-LM loss:
    train_encoder.py:301-305).  ``emb_rows`` [n, C] are the final embeddings of those positions, ``dlogits_rows`` [n, V]
    the rows of d(loss)/d(logits) that are not exact zeros (ops.masked_ce_rows on the DENSE logits).  forward() returns a
    zero scalar that stands for the loss in the graph; backward() contracts over the n rows only:
    d emb_rows = alpha dlogits_rows W, dW = alpha dlogits_rows^T emb_rows — exactly what the dense products give, the
    rows left out being all zero."""

    @staticmethod
    def forward(ctx, emb_rows, w, alpha, dlogits_rows):
        ctx.save_for_backward(emb_rows, w, dlogits_rows)
        ctx.alpha = alpha
        ctx.w_param = w
        ctx.pol = current_grad_policy()
        return torch.zeros((), dtype=torch.float32, device=emb_rows.device)

    @staticmethod
    def backward(ctx, dloss):
        emb_rows, w, dl = ctx.saved_tensors
        _ord_wait(ctx.pol, id(ctx.w_param))
        slot = _grad_slot(ctx.w_param, ctx.pol)
        dx, dw = ops.linear_bwd(dl, emb_rows.contiguous(), w, alpha=ctx.alpha, accumulate_into=slot)
        _ord_done(ctx.pol, id(ctx.w_param))
        return dx, dw, None, None


class _EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, wte, dropout_p, dropout_seed, order):
        ctx.save_for_backward(idx)
        ctx.vocab = wte.shape[0]
        ctx.w_param = wte
        ctx.drop = (dropout_p, dropout_seed)
        ctx.order = order   # optional: a stable argsort of idx.reshape(-1) (int32) the caller already has
        ctx.pol = current_grad_policy()
        return ops.embedding_fwd(idx.contiguous(), wte, dropout_p, dropout_seed)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        _ord_wait(ctx.pol, id(ctx.w_param))
        dw = ops.embedding_bwd(idx.contiguous(), dout.contiguous(), ctx.vocab, accumulate_into=_grad_slot(ctx.w_param, ctx.pol),
                               dropout_p=ctx.drop[0], dropout_seed=ctx.drop[1], order=ctx.order)
        _ord_done(ctx.pol, id(ctx.w_param))
        return None, dw, None, None, None


class _GradHandOff:
    """The hand-off of the dropout-masked gradient between consecutive blocks' backward passes (ops.block_bwd, dy_masked): block
    i + 1's backward leaves dropout(dx) under block i's MLP-projection mask here, block i's backward takes it instead of spending
    a pass on masking dy.  ONE object per forward call (OmniBioTA.forward creates it and every _BlockFn node of that call holds
    it on its ctx), slots keyed by block index: nothing process-wide, two models or two passes never see each other's entries.
    The taker uses the entry only if the gradient it received IS the dx the entry was made for and nobody touched it since —
    same storage and shape, the same version counter (autograd's in-place accumulation of a second contribution, a tensor hook
    that edits the gradient: both bump it), its own probability and seed; otherwise it masks dy itself."""

    __slots__ = ("slots",)

    def __init__(self):
        self.slots = {}

    def put(self, index, dx, dx_masked, p, seed):
        if dx_masked is not None:
            self.slots[index] = (dx_masked, dx, dx._version, float(p), int(seed))

    def take(self, index, dy, drop):
        ent = self.slots.pop(index, None)
        if ent is None or os.environ.get("OBTE_DROPOUT_HANDOFF") == "0":   # (A/B and tests: every block masks its own dy)
            return None
        dx_masked, dx, version, p, seed = ent
        if (dy.data_ptr() != dx.data_ptr() or dy.shape != dx.shape or dy._version != version or p != float(drop[0])
                or seed != int(drop[1])):
            return None
        return dx_masked


class _BlockFn(torch.autograd.Function):
    """One pre-LN transformer block (model.py:170-181): a single C call per pass."""

    @staticmethod
    def forward(ctx, x, ln1, attn_w, proj_w, ln2, fc_w, mlp_w, rope_cos, rope_sin, n_head, mask, dropout_p, dropout_seed, out_rows=None,
                below_seed=None, handoff=None, index=0):
        x = x.contiguous()
        ctx.below_seed = below_seed if (dropout_p > 0 and handoff is not None) else None   # the dropout seed of the block below (None: no block there)
        ctx.handoff, ctx.index = handoff, index   # this forward call's _GradHandOff and the block's place in it
        params = (ln1, attn_w, proj_w, ln2, fc_w, mlp_w)
        y, act = ops.block_fwd(x, params, (rope_cos, rope_sin), n_head, mask, dropout_p, dropout_seed, out_rows=out_rows)
        ctx.save_for_backward(x, act, rope_cos, rope_sin, *params)
        ctx.n_head, ctx.mask = n_head, mask
        ctx.out_rows = out_rows   # the rows form: y is [n, C], the positions the caller wants (ops.block_fwd)
        ctx.drop = (dropout_p, dropout_seed)
        ctx.w_params = params
        ctx.pol = current_grad_policy()
        return y

    @staticmethod
    def backward(ctx, dy):
        x, act, rope_cos, rope_sin, *params = ctx.saved_tensors
        pol = ctx.pol
        _ord_wait(pol, id(ctx.w_params[0]))   # one group per block: its six weights are updated by this one call
        slots = [_grad_slot(w, pol) for w in ctx.w_params]
        lnp = (_ln_partials(ctx.w_params[0], pol), _ln_partials(ctx.w_params[3], pol)) if pol.ln_mode else None
        dy = dy.contiguous()
        # dropout: the block above may have left dropout(dy) under this block's MLP-projection mask (one pass less per block)
        dy_masked = ctx.handoff.take(ctx.index, dy, ctx.drop) if (ctx.drop[0] > 0 and ctx.handoff is not None) else None
        res = ops.block_bwd(x, dy, act, tuple(params), (rope_cos, rope_sin), ctx.n_head, ctx.mask,
                            accumulate_into=slots, dropout_p=ctx.drop[0], dropout_seed=ctx.drop[1], ln_partials=lnp,
                            ln_partial_mode=pol.ln_mode, out_rows=ctx.out_rows, dy_masked=dy_masked, dx_mask_seed=ctx.below_seed)
        dx, grads = res[0], res[1]
        if ctx.below_seed is not None:
            ctx.handoff.put(ctx.index - 1, dx, res[2], ctx.drop[0], ctx.below_seed)
        _ord_done(pol, id(ctx.w_params[0]))
        return (dx, *grads, None, None, None, None, None, None, None, None, None, None)


class _AttnCoreFn(torch.autograd.Function):
    """rope + fused attention on a packed qkv activation (used by the standalone SelfAttention module)."""

    @staticmethod
    def forward(ctx, qkv, rope_cos, rope_sin, n_head, scale, mask, dropout_p, dropout_seed):
        B, T, C3 = qkv.shape
        hs = C3 // 3 // n_head
        qkv = qkv.contiguous().clone()
        ops.rope_qk_(qkv, rope_cos, rope_sin, B, T, n_head, hs)
        o, lse = ops.attn_fwd(qkv, B, T, n_head, hs, scale, mask, dropout_p, dropout_seed)
        ctx.save_for_backward(qkv, o, lse, rope_cos, rope_sin)
        ctx.meta = (B, T, n_head, hs, scale, mask, dropout_p, dropout_seed)
        return o

    @staticmethod
    def backward(ctx, d_o):
        qkv, o, lse, rope_cos, rope_sin = ctx.saved_tensors
        B, T, H, hs, scale, mask, dp, dseed = ctx.meta
        dqkv = ops.attn_bwd(qkv, o, d_o.contiguous(), lse, B, T, H, hs, scale, mask, rope=(rope_cos, rope_sin),
                            dropout_p=dp, dropout_seed=dseed)
        return dqkv, None, None, None, None, None, None, None


class _DropoutFn(torch.autograd.Function):
    """Elementwise dropout with the library's counter-based mask (used by the standalone SelfAttention / MLP modules;
    inside Block the same masks are applied in the GEMM epilogues)."""

    @staticmethod
    def forward(ctx, x, p, seed, site):
        ctx.cfg = (p, seed, site)
        return ops.dropout(x.contiguous(), p, seed, site)

    @staticmethod
    def backward(ctx, dy):
        p, seed, site = ctx.cfg
        return ops.dropout(dy.contiguous(), p, seed, site), None, None, None


def _new_seed() -> int:
    """62-bit seed from torch's CPU generator: reproducible under torch.manual_seed, restored by checkpoint()'s RNG
    preservation, and no GPU synchronisation."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def _active_p(module: nn.Module, p: float) -> float:
    p = float(p)
    if not 0.0 <= p < 1.0:
        raise ValueError(f"dropout probability has to be in [0, 1), got {p}")
    return p if module.training else 0.0


# ------------------------------------------------------------------------------------------------------- modules
class LayerNorm(nn.Module):
    """LayerNorm with optional bias (model.py:63-72).  The reference always builds it with bias=False."""

    def __init__(self, ndim, bias):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(ndim))
        self.bias = nn.Parameter(torch.zeros(ndim)) if bias else None

    def forward(self, input):
        if self.bias is not None:
            raise NotImplementedError("bias=True is not used by the reference (model.py:191) and not implemented")
        _require_hip(input, "LayerNorm")
        return _LayerNormFn.apply(input, self.weight)


class SelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        assert config.n_embd % config.n_head == 0
        self.c_attn = nn.Linear(config.n_embd, 3 * config.n_embd, bias=config.bias)
        self.c_proj = nn.Linear(config.n_embd, config.n_embd, bias=config.bias)
        self.attn_dropout = nn.Dropout(config.dropout, inplace=True)
        self.resid_dropout = nn.Dropout(config.dropout, inplace=True)
        self.n_head = config.n_head
        self.n_embd = config.n_embd
        self.dropout = config.dropout
        self.autoregressive = config.autoregressive
        self.flash = config.flash  # AttributeError if the caller never set it, as in the reference (model.py:89)
        self.register_buffer("freqs_cis", precompute_freqs_cis(self.n_embd // self.n_head, config.block_size))
        if not self.flash:
            # state_dict parity with the reference's non-flash modules (model.py:92-96); the buffer is unused here:
            # both settings run the same fused kernel, which is exact attention either way.
            self.register_buffer("bias", torch.tril(torch.ones(config.block_size, config.block_size))
                                 .view(1, 1, config.block_size, config.block_size))
        self._rope_key = None
        self._rope_val = None

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_rope_key"] = None
        state["_rope_val"] = None
        return state

    def rope(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """fp32 cos/sin tables derived from the (possibly dtype-cast) ``freqs_cis`` buffer, cached."""
        f = self.freqs_cis
        key = (f.data_ptr(), f.dtype, f.device, f._version)
        if self._rope_key != key:
            self._rope_val = rope_tables(f)
            self._rope_key = key
        return self._rope_val

    def forward(self, x, attn_mask=None):
        _require_hip(x, "SelfAttention")
        if self.autoregressive:
            raise NotImplementedError("autoregressive=True is not used by the encoder (model.py:192) and not implemented")
        B, T, C = x.size()
        mask = ops.MaskSpec.from_user(attn_mask, B, T, self.n_head, x.device)
        cos, sin = self.rope()
        p = _active_p(self, self.dropout)
        seed = _new_seed() if p > 0 else 0
        qkv = _LinearFn.apply(x, self.c_attn.weight, 1.0)
        y = _AttnCoreFn.apply(qkv, cos, sin, self.n_head, 8.0 / self.n_embd, mask, p, seed)
        y = _LinearFn.apply(y, self.c_proj.weight, 1.0)
        return _DropoutFn.apply(y, p, seed, L.SITE_RESID) if p > 0 else y


class MLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.c_fc = nn.Linear(config.n_embd, 4 * config.n_embd, bias=config.bias)
        self.c_proj = nn.Linear(4 * config.n_embd, config.n_embd, bias=config.bias)
        self.dropout = nn.Dropout(config.dropout, inplace=True)

    def forward(self, x):
        _require_hip(x, "MLP")
        p = _active_p(self, self.dropout.p)
        y = _LinearFn.apply(_LinearGeluFn.apply(x, self.c_fc.weight), self.c_proj.weight, 1.0)
        return _DropoutFn.apply(y, p, _new_seed(), L.SITE_MLP) if p > 0 else y


class Block(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.ln_1 = LayerNorm(config.n_embd, bias=config.bias)
        self.attn = SelfAttention(config)
        self.ln_2 = LayerNorm(config.n_embd, bias=config.bias)
        self.mlp = MLP(config)
        if config.bias:
            raise NotImplementedError("bias=True is not used by the reference (model.py:191) and not implemented")

    def forward(self, x, attn_mask=None, out_rows=None, below_seed=None, seed=None, handoff=None, index=0):
        """below_seed, seed, handoff, index (internal, OmniBioTA.forward): the dropout seed the block BELOW uses in this forward —
        this block's backward then also writes its dx under that block's MLP-projection mask (ops.block_bwd) and leaves it in
        `handoff` (the forward call's _GradHandOff) for block index - 1; `seed` is this block's own, drawn by the caller so that it
        can hand it on (None: drawn here).
        out_rows (optional, int64 (n,), ascending distinct rows of the flattened (b*t, n_embd) activation): the caller
        wants the block's output at those positions alone and gets it as (n, n_embd) — the attention half runs on every
        position, the MLP half on the listed ones (per-position arithmetic: the same values there; in training mode the
        dropout mask of the MLP projection is drawn for the (n, n_embd) output)."""
        _require_hip(x, "Block")
        _require_hip(self.attn.c_attn.weight, "Block parameters")
        if self.attn.autoregressive:
            raise NotImplementedError("autoregressive=True is not used by the encoder and not implemented")
        B, T, C = x.shape
        if out_rows is not None:
            _check_rows(out_rows, B * T, 1)
        mask = ops.MaskSpec.from_user(attn_mask, B, T, self.attn.n_head, x.device)
        cos, sin = self.attn.rope()
        # one dropout probability per block, as in the reference (config.dropout feeds all three nn.Dropout modules)
        p = _active_p(self, self.attn.dropout)
        if p > 0:
            seed = _new_seed() if seed is None else seed
        else:
            seed = 0
        return _BlockFn.apply(x, self.ln_1.weight, self.attn.c_attn.weight, self.attn.c_proj.weight, self.ln_2.weight,
                              self.mlp.c_fc.weight, self.mlp.c_proj.weight, cos, sin, self.attn.n_head, mask, p, seed, out_rows,
                              below_seed if p > 0 else None, handoff if p > 0 else None, index)


def _check_rows(rows, total: int, n_blocks: int) -> None:
    """The contract of ``OmniBioTA.forward(rows=)`` / ``Block.forward(out_rows=)``: a non-empty int64 vector on the GPU of
    ASCENDING, DISTINCT positions in [0, total).  Shape, dtype and device are checked on every call (no device read).  The
    values are the caller's responsibility — duplicates are unsupported (the scatter of the block backward would keep one of
    the contributions instead of their sum) and the gather clamps an out-of-range index instead of faulting — unless
    OBTE_CHECK_ROWS=1, which verifies bounds and strict monotonicity at the price of a device round trip per call."""
    if not torch.is_tensor(rows) or rows.dtype != torch.int64 or rows.dim() != 1 or not rows.is_cuda:
        raise ValueError("OmniBioTA.forward: rows must be a 1-D int64 tensor on the GPU")
    if rows.numel() == 0 or n_blocks == 0:
        raise ValueError("OmniBioTA.forward: rows must list at least one position (and the model needs a block)")
    if rows.numel() > total:
        raise ValueError(f"OmniBioTA.forward: {rows.numel()} rows listed, the batch has {total} positions")
    if os.environ.get("OBTE_CHECK_ROWS") == "1":
        r = rows.detach().cpu()
        if int(r[0]) < 0 or int(r[-1]) >= total or (r.numel() > 1 and not bool((r[1:] > r[:-1]).all())):
            raise ValueError(f"OmniBioTA.forward: rows must be strictly ascending positions in [0, {total})")


@dataclass
class OmniBioTAConfig:
    block_size: int = 2048
    vocab_size: int = 2 ** 16
    n_layer: int = 12
    n_head: int = 12
    n_embd: int = 1024
    dropout: float = 0.1
    bias: bool = False
    autoregressive: bool = False
    checkpoint_freq: int = 0


class MuReadout(_MuReadoutBase):
    """The readout layer (model.py:208).  With the real ``mup`` package this is its MuReadout with the matmul
    swapped for the HIP GEMM; without it, the restated class from mup_compat."""

    def forward(self, x):
        _require_hip(x, "MuReadout")
        wm = self.width_mult()
        return _LinearFn.apply(x, self.weight, float(self.output_mult) / float(wm))


class OmniBioTA(nn.Module):
    def __init__(self, config):
        super().__init__()
        assert config.vocab_size is not None
        assert config.block_size is not None
        self.config = config
        self.transformer = nn.ModuleDict(dict(
            wte=nn.Embedding(config.vocab_size, config.n_embd),
            drop=nn.Dropout(config.dropout, inplace=True),
            h=nn.ModuleList([Block(config) for _ in range(config.n_layer)]),
            ln_f=LayerNorm(config.n_embd, bias=config.bias),
        ))
        self.lm_head = MuReadout(config.n_embd, config.vocab_size, bias=False)
        print("number of parameters: %.2fM" % (self.get_num_params() / 1e6,))

    def get_num_params(self, non_embedding=True):
        """All parameters, minus the token embedding when non_embedding (model.py:213-223; lm_head is counted)."""
        n_params = sum(p.numel() for p in self.parameters())
        if non_embedding:
            n_params -= self.transformer.wte.weight.numel()
        return n_params

    def forward(self, idx, attn_mask=None, return_embeddings=False, rows=None):
        """idx (b, t) int64 -> logits (b, t, vocab) or, with return_embeddings, emb (b, t, n_embd)
        (model.py:225-254).  ``attn_mask``: None, the reference's additive (b, n_head, t, t) tensor (any strides,
        expand() views included), or a ``masks.RangeMask`` (per-query key ranges; the fast path).
        ``rows`` (an extension, not in the reference; int64 (n,), ascending positions of the flattened (b*t) batch): the
        caller needs the result at those positions only — a masked-LM loss looks at ~15 % of them (train_encoder.py:304) —
        and gets (n, n_embd) embeddings or (n, vocab) logits.  Nothing after the last block's attention mixes positions, so
        that block's MLP half, ln_f and the readout run on the listed positions alone; every value returned is the one the full
        forward computes there (the same arithmetic per position; few-tile projections may sum their K range in split-K order)."""
        _, t = idx.size()
        assert t <= self.config.block_size, f"Cannot forward sequence of length {t}, block size is only {self.config.block_size}"
        wte = self.transformer.wte.weight
        _require_hip(wte, "OmniBioTA")
        if not idx.is_cuda:
            raise RuntimeError("OmniBioTA.forward: idx must be on the GPU")
        b = idx.shape[0]
        mask = ops.MaskSpec.from_user(attn_mask, b, t, self.config.n_head, idx.device)
        p = _active_p(self, self.transformer.drop.p)
        # a training harness that sorts the token ids of a whole optimizer step in one call (the embedding backward sums
        # gradient rows in sorted-id order) hands this micro-batch's order through `with embedding_order(...)`; consumed once
        box = _embedding_order.get()
        order = None
        if box is not None:
            order, box[0] = box[0], None
        if order is not None and (order.numel() != idx.numel() or order.dtype != torch.int32 or order.device != idx.device):
            order = None
        x = _EmbeddingFn.apply(idx, wte, p, _new_seed() if p > 0 else 0, order)
        n_blocks = len(self.transformer.h)
        if rows is not None:
            _check_rows(rows, b * t, n_blocks)
        # dropout: the seed of the block just run is handed to the next one, whose backward masks its dx for the block below and
        # leaves it in this call's hand-off object (nothing of this lives on the modules or in the process)
        below = None
        handoff = _GradHandOff()
        for i, block in enumerate(self.transformer.h):
            last_rows = rows if (rows is not None and i == n_blocks - 1) else None
            if self.config.checkpoint_freq > 0 and i % self.config.checkpoint_freq == 0:
                x = checkpoint(block, x, mask, last_rows, use_reentrant=False)
                below = None   # (a recomputed block redraws nothing, but keep the hand-off out of checkpointed segments)
            else:
                bp = _active_p(block, block.attn.dropout)
                seed = _new_seed() if bp > 0 else None
                x = block(x, attn_mask=mask, out_rows=last_rows, below_seed=below, seed=seed, handoff=handoff, index=i)
                below = seed
        emb = self.transformer.ln_f(x)
        if return_embeddings:
            return emb
        return self.lm_head(emb)

    def encode(self, idx, method="mean"):
        """Pooled sequence embedding (model.py:256-277)."""
        assert method in ["mean", "first", "last", "max", "all"], f"Unknown pooling method {method}"
        emb = self.forward(idx, return_embeddings=True)
        if method == "mean":
            return emb.mean(dim=1)
        elif method == "first":
            return emb[:, 0]
        elif method == "last":
            return emb[:, -1]
        elif method == "max":
            return emb.max(dim=1)[0]
        return emb
