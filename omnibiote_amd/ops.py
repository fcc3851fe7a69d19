"""Tensor-level wrappers over the C ABI (include/omnibiote_hip.h).

Each function validates shapes/dtypes/devices on the host (the kernels assume them), borrows raw device
pointers from PyTorch-owned tensors for the duration of the call, and enqueues on PyTorch's *current* HIP
stream.  Nothing here synchronises the host.  PyTorch is plumbing only: allocation, streams, autograd glue.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib as L

bf16 = torch.bfloat16


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need(t: torch.Tensor, name: str, dtype=bf16, contiguous: bool = True) -> None:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the OmniBioTE HIP path runs on MI355X only; got a {t.device} tensor "
                           "(there is no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x: torch.Tensor, w: torch.Tensor, eps: float = 1e-5):
    _need(x, "x"); _need(w, "weight")
    cols = x.shape[-1]
    rows = x.numel() // cols
    assert w.numel() == cols
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    L.check(L.lib().obte_layernorm_fwd(_ptr(x), _ptr(w), _ptr(y), _ptr(mean), _ptr(rstd), rows, cols, eps, _stream()),
            "obte_layernorm_fwd")
    return y, mean, rstd


def ln_partials_buffer(cols: int, device) -> torch.Tensor:
    """A persistent fp32 partial-sum buffer for obte_layernorm_bwd_partial (one per LayerNorm weight)."""
    return torch.empty(L.lib().obte_layernorm_bwd_ws_rows() * cols, dtype=torch.float32, device=device)


def layernorm_bwd(dy, x, w, mean, rstd, dresid=None, accumulate_into=None, partials=None, partial_mode=0, dropout=None):
    """accumulate_into: an existing bf16 weight gradient to add into in place (then the returned dw is None).
    partials + partial_mode (L.LN_PARTIAL_*): accumulate the weight gradient over calls in the caller's fp32 buffer; dw is
    returned only by LN_PARTIAL_LAST (None otherwise).
    dropout=(p, seed, site): also return dropout(dx) under that mask as a third value (obte_layernorm_bwd_dropout)."""
    _need(dy, "dy"); _need(x, "x"); _need(w, "weight")
    _need(mean, "mean", torch.float32); _need(rstd, "rstd", torch.float32)
    cols = x.shape[-1]
    rows = x.numel() // cols
    assert dy.shape == x.shape and mean.numel() == rows and rstd.numel() == rows
    if dresid is not None:
        _need(dresid, "dresid"); assert dresid.shape == x.shape
    dx = torch.empty_like(x)
    if dropout is not None:
        dp, dseed, dsite = dropout
        dxd = torch.empty_like(x)
        if partial_mode:
            _need(partials, "partials", torch.float32)
            assert accumulate_into is None and partials.numel() == L.lib().obte_layernorm_bwd_ws_rows() * cols
            dw = torch.empty_like(w) if partial_mode == L.LN_PARTIAL_LAST else None
            buf = partials
        else:
            dw = accumulate_into if accumulate_into is not None else torch.empty_like(w)
            buf = torch.empty(L.lib().obte_layernorm_bwd_ws_rows() * cols, dtype=torch.float32, device=x.device)
        L.check(L.lib().obte_layernorm_bwd_dropout(_ptr(dy), _ptr(x), _ptr(w), _ptr(mean), _ptr(rstd), _ptr(dresid), _ptr(dx), _ptr(dxd), _ptr(dw),
                                                    _ptr(buf), rows, cols, int(partial_mode), int(accumulate_into is not None), float(dp), int(dseed),
                                                    int(dsite), _stream()), "obte_layernorm_bwd_dropout")
        return dx, (None if (accumulate_into is not None and not partial_mode) else dw), dxd
    if partial_mode:
        _need(partials, "partials", torch.float32)
        assert accumulate_into is None and partials.numel() == L.lib().obte_layernorm_bwd_ws_rows() * cols
        dw = torch.empty_like(w) if partial_mode == L.LN_PARTIAL_LAST else None
        L.check(L.lib().obte_layernorm_bwd_partial(_ptr(dy), _ptr(x), _ptr(w), _ptr(mean), _ptr(rstd), _ptr(dresid), _ptr(dx), _ptr(dw),
                                                    _ptr(partials), rows, cols, int(partial_mode), _stream()), "obte_layernorm_bwd_partial")
        return dx, dw
    if accumulate_into is not None:
        _need(accumulate_into, "grad"); assert accumulate_into.shape == w.shape
    dw = accumulate_into if accumulate_into is not None else torch.empty_like(w)
    ws = torch.empty(L.lib().obte_layernorm_bwd_ws_rows() * cols, dtype=torch.float32, device=x.device)
    L.check(L.lib().obte_layernorm_bwd_acc(_ptr(dy), _ptr(x), _ptr(w), _ptr(mean), _ptr(rstd), _ptr(dresid), _ptr(dx),
                                            _ptr(dw), _ptr(ws), rows, cols, int(accumulate_into is not None), _stream()),
            "obte_layernorm_bwd")
    return dx, (None if accumulate_into is not None else dw)


# --------------------------------------------------------------------------------------------------------- GEMM
def gemm(a, b, M, N, K, a_kmajor=True, b_kmajor=True, epilogue=L.EPI_NONE, aux=None, alpha=1.0, out=None, dropout=None,
         rope=None):
    """D[M,N] = epilogue(alpha * sum_k A(m,k) B(n,k)); see include/omnibiote_hip.h.  Returns d, or (d, d2) for
    the GELU epilogue."""
    _need(a, "a"); _need(b, "b")
    lda = K if a_kmajor else M
    ldb = K if b_kmajor else N
    assert a.numel() == M * K, (a.shape, M, K)
    assert b.numel() == N * K, (b.shape, N, K)
    d = out if out is not None else torch.empty((M, N), dtype=bf16, device=a.device)
    _need(d, "d"); assert d.numel() == M * N
    d2 = None
    if epilogue == L.EPI_GELU:
        d2 = torch.empty((M, N), dtype=bf16, device=a.device)
    if epilogue in (L.EPI_ADD, L.EPI_GELU_BWD, L.EPI_ADD_DROPOUT):
        _need(aux, "aux"); assert aux.numel() == M * N
    dp, dseed, dsite = dropout if dropout is not None else (0.0, 0, 0)   # (p, seed, site) for EPI_ADD_DROPOUT
    g = L.GemmArgs(_ptr(a), _ptr(b), _ptr(d), _ptr(aux), _ptr(d2), M, N, K, lda, ldb, N,
                   int(a_kmajor), int(b_kmajor), epilogue, float(alpha), float(dp), int(dsite), int(dseed))
    if epilogue == L.EPI_ROPE_QK:   # rope = (cos, sin, T, head_dim)
        cos, sin, rT, rhs = rope
        _need(cos, "cos", torch.float32); _need(sin, "sin", torch.float32)
        assert cos.shape[0] >= rT and cos.shape[-1] == rhs // 2
        g.rope_cos, g.rope_sin, g.rope_T, g.rope_head_dim = _ptr(cos), _ptr(sin), rT, rhs
    ws_bytes = int(L.lib().obte_gemm_workspace_bytes(M, N, K)) if (epilogue in (L.EPI_NONE, L.EPI_ADD) and M * N <= (1 << 23)) else 0
    if ws_bytes > 0:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device)
        L.check(L.lib().obte_gemm_bf16_ws(C.byref(g), _ptr(ws), ws_bytes, _stream()), "obte_gemm_bf16_ws")
    else:
        L.check(L.lib().obte_gemm_bf16(C.byref(g), _stream()), "obte_gemm_bf16")
    return (d, d2) if d2 is not None else d


def gemm_grouped(problems):
    """Up to six GEMMs in a single launch (obte_gemm_grouped_bf16).  ``problems`` is a list of dicts with keys
    a, b, M, N, K, out and optionally a_kmajor / b_kmajor (default False: the weight-gradient layout) and accumulate
    (out += alpha A B in place) / alpha.  Put the problems with the longest K first.  Returns the outs."""
    assert 1 <= len(problems) <= 6
    arr = (L.GemmArgs * len(problems))()
    for i, q in enumerate(problems):
        a, b, M, N, K, out = q["a"], q["b"], q["M"], q["N"], q["K"], q["out"]
        ak, bk, acc = bool(q.get("a_kmajor", False)), bool(q.get("b_kmajor", False)), bool(q.get("accumulate", False))
        _need(a, "a"); _need(b, "b"); _need(out, "out")
        assert a.numel() == M * K and b.numel() == N * K and out.numel() == M * N
        arr[i] = L.GemmArgs(_ptr(a), _ptr(b), _ptr(out), _ptr(out) if acc else None, None, M, N, K,
                            K if ak else M, K if bk else N, N, int(ak), int(bk),
                            L.EPI_ADD if acc else L.EPI_NONE, float(q.get("alpha", 1.0)), 0.0, 0, 0)
    L.check(L.lib().obte_gemm_grouped_bf16(arr, len(problems), _stream()), "obte_gemm_grouped_bf16")
    return [q["out"] for q in problems]


def linear_fwd(x2d, w, epilogue=L.EPI_NONE, aux=None, alpha=1.0, dropout=None):
    """y = x W^T for x [M,K], W [N,K] (nn.Linear forward)."""
    M, K = x2d.shape
    N = w.shape[0]
    assert w.shape[1] == K
    return gemm(x2d, w, M, N, K, True, True, epilogue, aux, alpha, dropout=dropout)


def dropout(x, p, seed, site=L.SITE_USER, out=None):
    """out = dropout(x) with the library's counter-based mask: x is read as a matrix [numel / last dim, last dim] and
    element (row, col) decides (include/omnibiote_hip.h, dropout); x may alias out."""
    _need(x, "x")
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().obte_dropout_bf16(_ptr(x), _ptr(out), x.numel(), int(x.shape[-1]), float(p), int(seed), int(site), _stream()),
            "obte_dropout_bf16")
    return out


def linear_dgrad(dy2d, w, epilogue=L.EPI_NONE, aux=None, alpha=1.0):
    """dx = dy W for dy [M,N], W [N,K]."""
    M, N = dy2d.shape
    K = w.shape[1]
    assert w.shape[0] == N
    return gemm(dy2d, w, M, K, N, True, False, epilogue, aux, alpha)


def linear_wgrad(dy2d, x2d, alpha=1.0, accumulate_into=None):
    """dW = dy^T x for dy [M,N], x [M,K] -> [N,K].  accumulate_into: add into this tensor in place (alpha must be 1)."""
    M, N = dy2d.shape
    K = x2d.shape[1]
    assert x2d.shape[0] == M
    if accumulate_into is not None:
        return gemm(dy2d, x2d, N, K, M, False, False, L.EPI_ADD, accumulate_into, alpha, out=accumulate_into)
    return gemm(dy2d, x2d, N, K, M, False, False, L.EPI_NONE, None, alpha)


def _tiles256(m, n):
    return ((m + 255) // 256) * ((n + 255) // 256)


def linear_bwd_pair_is_grouped(M, N, K) -> bool:
    """Whether linear_bwd (dy [M,N], x [M,K], W [N,K]) takes the single grouped launch.  Only on request (OBTE_GROUPED_LM=1):
    the readout's pair — an input gradient of few 256x256 tiles with a long K (= vocabulary) beside a weight gradient of many
    tiles with a short K (= rows) — was grouped by default in rounds 3-4, when the single launches had no good split-K plans;
    with the tuned plans the two launches are faster (round 5, masked rows 4 740-4 915: 1 030-1 080 us against 1 440-1 460 us
    grouped; the step with every pass grouped 108.3 ms, with none 105.9 ms), so the default is the two launches."""
    import os
    return os.environ.get("OBTE_GROUPED_LM", "") == "1" and M >= 128 and N >= 128


def linear_bwd(dy2d, x2d, w, alpha=1.0, accumulate_into=None):
    """Both gradients of y = alpha x W^T: dx = alpha dy W, dW = alpha dy^T x (added into ``accumulate_into`` when given).
    Returns (dx, dW or None).  One grouped launch when linear_bwd_pair_is_grouped, else the two single launches."""
    M, N = dy2d.shape
    K = x2d.shape[1]
    if not linear_bwd_pair_is_grouped(M, N, K):
        dx = linear_dgrad(dy2d, w, alpha=alpha)
        dw = linear_wgrad(dy2d, x2d, alpha=alpha, accumulate_into=accumulate_into)
        return dx, (None if accumulate_into is not None else dw)
    dx = torch.empty((M, K), dtype=bf16, device=dy2d.device)
    dw = accumulate_into if accumulate_into is not None else torch.empty((N, K), dtype=bf16, device=dy2d.device)
    gemm_grouped([dict(a=dy2d, b=w, M=M, N=K, K=N, out=dx, a_kmajor=True, alpha=alpha),
                  dict(a=dy2d, b=x2d, M=N, N=K, K=M, out=dw, accumulate=accumulate_into is not None, alpha=alpha)])
    return dx, (None if accumulate_into is not None else dw)


# --------------------------------------------------------------------------------------------------------- RoPE
def rope_qk_(qkv, cos, sin, B, T, H, hs, inverse=False):
    _need(qkv, "qkv"); _need(cos, "cos", torch.float32); _need(sin, "sin", torch.float32)
    assert qkv.numel() == B * T * 3 * H * hs
    assert cos.shape[-1] == hs // 2 and cos.shape[0] >= T and cos.shape == sin.shape
    L.check(L.lib().obte_rope_qk_inplace(_ptr(qkv), _ptr(cos), _ptr(sin), B, T, H, hs, int(inverse), _stream()),
            "obte_rope_qk_inplace")
    return qkv


# ---------------------------------------------------------------------------------------------------- attention
class MaskSpec:
    """How a mask reaches the kernels: per-query key ranges (int32 [B,T,2]) or a dense additive bf16 tensor
    (B, H|1, T, T), possibly an expand() view with stride-0 heads, last dim contiguous."""

    __slots__ = ("ranges", "dense", "sb", "sh", "sq", "qbounds", "exact")

    def __init__(self, ranges=None, dense=None, qbounds=None, exact=None):
        """With ``dense``: ``ranges`` / ``qbounds`` are the optional loop bounds of obte_mask_bounds (per query / per key);
        the dense values still decide every weight — unless ``exact`` (obte_mask_bounds' device flag) says the mask IS a
        range mask, in which case the range kernels serve it (include/omnibiote_hip.h, obte_mask_bounds)."""
        self.ranges, self.dense, self.qbounds, self.exact = ranges, dense, qbounds, exact
        self.sb = self.sh = self.sq = 0
        if dense is not None:
            self.sb, self.sh, self.sq = dense.stride(0), dense.stride(1), dense.stride(2)

    @staticmethod
    def dense_with_bounds(m):
        """A dense additive mask (B, H, T, T) plus the conservative bounds that let the kernels skip the key tiles every
        row masks out (one pass over the mask per model forward; shared by all layers)."""
        B, H, T, _ = m.shape
        kb = torch.empty((B, T, 2), dtype=torch.int32, device=m.device)
        qb = torch.empty((B, T, 2), dtype=torch.int32, device=m.device)
        scratch = torch.empty((B * T,), dtype=torch.uint8, device=m.device)
        exact = torch.empty((1,), dtype=torch.int32, device=m.device)
        counts = torch.empty((B * T,), dtype=torch.int32, device=m.device)
        L.check(L.lib().obte_mask_bounds(_ptr(m), m.stride(0), m.stride(1), m.stride(2), B, H, T, _ptr(kb), _ptr(qb), _ptr(scratch),
                                         _ptr(exact), _ptr(counts), _stream()), "obte_mask_bounds")
        return MaskSpec(ranges=kb, dense=m, qbounds=qb, exact=exact)

    @staticmethod
    def from_user(attn_mask, B, T, H, device):
        if attn_mask is None:
            return MaskSpec()
        if isinstance(attn_mask, MaskSpec):
            return attn_mask
        if hasattr(attn_mask, "key_ranges"):  # masks.RangeMask
            r = attn_mask.key_ranges
            _need(r, "key_ranges", torch.int32)
            if tuple(r.shape) != (B, T, 2):
                raise RuntimeError(f"key_ranges must be (B,T,2)=({B},{T},2), got {tuple(r.shape)}")
            return MaskSpec(ranges=r)
        m = attn_mask
        if not isinstance(m, torch.Tensor):
            raise TypeError("attn_mask must be None, a tensor or a RangeMask")
        if m.dim() == 3:
            m = m.unsqueeze(1)
        if m.dim() != 4 or m.shape[0] != B or m.shape[2] != T or m.shape[3] != T or m.shape[1] not in (1, H):
            raise RuntimeError(f"attn_mask must be (B, n_head, T, T)=({B},{H},{T},{T}), got {tuple(attn_mask.shape)}")
        if not m.is_cuda:
            raise RuntimeError("attn_mask must be on the GPU")
        if m.dtype == torch.bool:
            raise RuntimeError("boolean masks are not part of the reference's contract; pass an additive float mask")
        if m.dtype != bf16 or m.stride(3) != 1:
            # convert without materialising H copies of an expand()ed mask
            if m.shape[1] == 1 or m.stride(1) == 0:
                m = m[:, :1].to(bf16).contiguous()
            else:
                m = m.to(bf16).contiguous()
        if m.shape[1] == 1:
            m = m.expand(B, H, T, T)
        return MaskSpec.dense_with_bounds(m)


def attn_fwd(qkv, B, T, H, hs, scale, mask: Optional[MaskSpec] = None, dropout_p=0.0, dropout_seed=0, keep_bits=False):
    """keep_bits (dropout only): also return the forward's keep decisions in key-major order (include/omnibiote_hip.h,
    obte_attn_fwd_args::drop_bits) for attn_bwd(drop_bits=...): (o, lse, bits)."""
    _need(qkv, "qkv"); assert qkv.numel() == B * T * 3 * H * hs
    mask = mask or MaskSpec()
    o = torch.empty((B, T, H * hs), dtype=bf16, device=qkv.device)
    lse = torch.empty((B, H, T), dtype=torch.float32, device=qkv.device)
    bits = None
    if keep_bits and dropout_p > 0.0:   # (zeroed: words of key tiles the forward skips stay defined; they belong to masked pairs)
        bits = torch.zeros(int(L.lib().obte_attn_drop_bits_bytes(B, T, H)) // 4, dtype=torch.int32, device=qkv.device)
    a = L.AttnFwdArgs(_ptr(qkv), _ptr(o), _ptr(lse), _ptr(mask.ranges), _ptr(mask.dense), mask.sb, mask.sh, mask.sq,
                      B, T, H, hs, float(scale), float(dropout_p), int(dropout_seed), _ptr(mask.exact), _ptr(bits))
    L.check(L.lib().obte_attn_fwd(C.byref(a), _stream()), "obte_attn_fwd")
    return (o, lse, bits) if keep_bits else (o, lse)


def attn_bwd(qkv, o, d_o, lse, B, T, H, hs, scale, mask: Optional[MaskSpec] = None, rope=None, dropout_p=0.0, dropout_seed=0,
             one_kernel: bool = True, drop_bits=None):
    """one_kernel: hand the library the scratch that lets it run the one-kernel backward where that form applies (head size
    128, no dense mask; with dropout: only from the forward's keep bits, drop_bits — include/omnibiote_hip.h,
    obte_attn_bwd_args::ws); False: the dQ + dK/dV kernel pair."""
    _need(qkv, "qkv"); _need(o, "o"); _need(d_o, "d_o"); _need(lse, "lse", torch.float32)
    mask = mask or MaskSpec()
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, H, T), dtype=torch.float32, device=qkv.device)
    cos, sin = rope if rope is not None else (None, None)
    ws_bytes = int(L.lib().obte_attn_bwd_ws_bytes(B, T, H, hs)) if (one_kernel and (dropout_p == 0.0 or drop_bits is not None) and mask.dense is None) else 0
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=qkv.device) if ws_bytes > 0 else None
    a = L.AttnBwdArgs(_ptr(qkv), _ptr(o), _ptr(d_o), _ptr(lse), _ptr(delta), _ptr(dqkv), _ptr(cos), _ptr(sin),
                      _ptr(mask.ranges), _ptr(mask.dense), mask.sb, mask.sh, mask.sq, B, T, H, hs, float(scale),
                      float(dropout_p), int(dropout_seed), _ptr(mask.qbounds), _ptr(mask.exact), _ptr(ws), ws_bytes, _ptr(drop_bits))
    L.check(L.lib().obte_attn_bwd(C.byref(a), _stream()), "obte_attn_bwd")
    return dqkv


# ---------------------------------------------------------------------------------------------------- embedding
def embedding_fwd(idx, wte, dropout_p=0.0, dropout_seed=0):
    _need(idx, "idx", torch.int64); _need(wte, "wte")
    V, Cc = wte.shape
    out = torch.empty(tuple(idx.shape) + (Cc,), dtype=bf16, device=wte.device)
    L.check(L.lib().obte_embedding_fwd_dropout(_ptr(idx), _ptr(wte), _ptr(out), idx.numel(), Cc, V, float(dropout_p),
                                                int(dropout_seed), _stream()), "obte_embedding_fwd")
    return out


def embedding_bwd(idx, dout, vocab, accumulate_into=None, dropout_p=0.0, dropout_seed=0, order=None):
    """accumulate_into: an existing dense (vocab, C) gradient; the touched rows are updated in place
    (bf16(old + bf16(sum))), nothing else is read or written.  order: a stable argsort of idx.reshape(-1) as int32, if
    the caller has one already (the harness sorts a whole optimizer step's ids in one call)."""
    _need(idx, "idx", torch.int64); _need(dout, "dout")
    rows, Cc = idx.numel(), dout.shape[-1]
    if order is None:
        # the order is plumbing (a stable sort of <= a few 10k token ids); the summation itself is ours
        order = torch.sort(idx.reshape(-1), stable=True).indices.to(torch.int32)
    else:
        _need(order, "order", torch.int32); assert order.numel() == rows
    ws = torch.empty(max(int(L.lib().obte_embedding_bwd_ws_bytes(rows, Cc)), 16), dtype=torch.uint8, device=dout.device)
    if accumulate_into is not None:
        _need(accumulate_into, "grad"); assert tuple(accumulate_into.shape) == (vocab, Cc)
        L.check(L.lib().obte_embedding_bwd_dropout(_ptr(idx), _ptr(order), _ptr(dout), _ptr(accumulate_into), _ptr(ws), rows, Cc,
                                                    vocab, 1, float(dropout_p), int(dropout_seed), _stream()), "obte_embedding_bwd")
        return None
    dwte = torch.empty((vocab, Cc), dtype=bf16, device=dout.device)
    L.check(L.lib().obte_embedding_bwd_dropout(_ptr(idx), _ptr(order), _ptr(dout), _ptr(dwte), _ptr(ws), rows, Cc, vocab, 0,
                                                float(dropout_p), int(dropout_seed), _stream()), "obte_embedding_bwd")
    return dwte


# ---------------------------------------------------------------------------------------------- loss, optimizer
class DLogitsBuffer:
    """A gradient buffer reused across micro-batches by ``masked_ce``: only the rows whose masked-ness changed since
    the previous call are written (the other ~85 % already hold zeros).  The caller must be done with the previous
    contents (i.e. have run the backward that consumes them) before the next ``masked_ce`` on the same buffer — true for
    a training loop, where forward/loss/backward of one micro-batch are enqueued before the next begins."""

    def __init__(self):
        self.buf = None
        self.prev_mask = None

    def get(self, like: torch.Tensor):
        if self.buf is None or self.buf.shape != like.shape or self.buf.device != like.device:
            self.buf = torch.empty_like(like)
            self.prev_mask = None      # fresh memory: the next call writes every row
        return self.buf


def masked_ce(logits, targets, mlm_mask, n_accum: int, reuse: Optional[DLogitsBuffer] = None):
    """Returns (loss scalar fp32 tensor, dlogits bf16) with the reference's micro-batch normalisation
    (train_encoder.py:301-305): loss = sum_masked(CE)/n_accum / count."""
    _need(logits, "logits"); _need(targets, "targets", torch.int64)
    V = logits.shape[-1]
    rows = logits.numel() // V
    assert targets.numel() == rows and mlm_mask.numel() == rows
    m8 = mlm_mask.reshape(-1).to(torch.uint8).contiguous()
    inv_count = (1.0 / m8.sum(dtype=torch.float32)).reshape(1)
    row_loss = torch.empty(rows, dtype=torch.float32, device=logits.device)
    if reuse is None:
        dlogits, prev = torch.empty_like(logits), None
    else:
        dlogits, prev = reuse.get(logits), reuse.prev_mask
    L.check(L.lib().obte_masked_ce_fwd_bwd_reuse(_ptr(logits), _ptr(targets), _ptr(m8), _ptr(prev), _ptr(inv_count), 1.0 / n_accum,
                                                  _ptr(row_loss), _ptr(dlogits), rows, V, _stream()), "obte_masked_ce_fwd_bwd")
    if reuse is not None:
        reuse.prev_mask = m8
    loss = row_loss.sum() * inv_count[0]
    return loss, dlogits


def masked_ce_rows(logits, targets, rows, n_accum: int, row_weights: Optional[torch.Tensor] = None):
    """The same loss on a compact list of masked positions: ``rows`` int64 (n,) ascending row indices into the dense
    logits (viewed [M, V]) — or None when logits and targets already hold the listed rows alone ([n, V] and (n,): the
    readout that computes the masked rows only).  Returns (loss, dlogits_rows bf16 [n, V]) — the rows of d(logits) that are
    not exact zeros.
    row_weights (fp32 (n,), optional): replaces the 1/n normalisation by a weight per listed row — a call that covers
    several micro-batches passes 1 / (masked tokens of the row's own micro-batch)."""
    _need(logits, "logits"); _need(targets, "targets", torch.int64)
    V = logits.shape[-1]
    M = logits.numel() // V
    if rows is not None:
        _need(rows, "rows", torch.int64)
    n = rows.numel() if rows is not None else M
    assert targets.numel() == M and 0 < n <= M
    if row_weights is not None:
        _need(row_weights, "row_weights", torch.float32); assert row_weights.numel() == n
    # the 1 / n of the mean is known on the host here (the list's length): it travels in the by-value row scale, and the device-side
    # gradient scale is a constant 1 (a NULL grad_scale: no persistent [1.0] tensor whose fill another stream would have to be
    # ordered after) — no fill launch per call, and the loss is the plain sum of the row losses
    row_scale = (1.0 if row_weights is not None else 1.0 / n) / n_accum
    row_loss = torch.empty(n, dtype=torch.float32, device=logits.device)
    dl = torch.empty((n, V), dtype=bf16, device=logits.device)
    L.check(L.lib().obte_masked_ce_rows(_ptr(logits), _ptr(targets), _ptr(rows), None, row_scale, _ptr(row_weights),
                                        _ptr(row_loss), _ptr(dl), n, M, V, _stream()), "obte_masked_ce_rows")
    return row_loss.sum(), dl


def adamw_step_(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, clip_coef=None):
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _need(t, n)
    assert p.numel() == g.numel() == m.numel() == v.numel()
    L.check(L.lib().obte_adamw_bf16(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, beta1, beta2, eps, weight_decay,
                                     int(step), _ptr(clip_coef), _stream()), "obte_adamw_bf16")


def sumsq_(g, out):
    _need(g, "g"); _need(out, "out", torch.float32)
    L.check(L.lib().obte_sumsq_bf16(_ptr(g), g.numel(), _ptr(out), _stream()), "obte_sumsq_bf16")


# -------------------------------------------------------------------------------------------------------- block
def _block_desc(B, T, Cc, H, params, rope, mask: MaskSpec, dropout_p=0.0, dropout_seed=0, ln_partials=None, ln_partial_mode=0, out_rows=None,
                dy_masked=None, dx_masked=None, dx_mask_seed=0):
    ln1, attn_w, proj_w, ln2, fc_w, mlp_w = params
    p1, p2 = ln_partials if ln_partials is not None else (None, None)
    if out_rows is not None:
        _need(out_rows, "out_rows", torch.int64)
        assert 0 < out_rows.numel() <= B * T, "the rows form of the block: a non-empty list of positions"
    return L.BlockDesc(B, T, Cc, H, _ptr(ln1), _ptr(attn_w), _ptr(proj_w), _ptr(ln2), _ptr(fc_w), _ptr(mlp_w),
                       _ptr(rope[0]), _ptr(rope[1]), _ptr(mask.ranges), _ptr(mask.dense), mask.sb, mask.sh, mask.sq,
                       float(dropout_p), int(dropout_seed), _ptr(mask.qbounds), _ptr(p1), _ptr(p2), int(ln_partial_mode), _ptr(mask.exact),
                       _ptr(out_rows), 0 if out_rows is None else out_rows.numel(), _ptr(dy_masked), _ptr(dx_masked), int(dx_mask_seed))


def block_fwd(x, params, rope, H, mask: MaskSpec, dropout_p=0.0, dropout_seed=0, out_rows=None):
    """One transformer block forward.  Returns (y, act) where act is the opaque saved-activation buffer.
    out_rows (int64 (n,), ascending distinct rows of the [B*T, C] activation): only those positions of the output are
    wanted — y is [n, C], the MLP half runs on them alone (include/omnibiote_hip.h, obte_block_desc::out_rows).  With dropout
    the mask of the MLP projection (site 3) is that of the [n, C] output — element (i, c) — in forward and backward alike."""
    _need(x, "x")
    B, T, Cc = x.shape
    for i, w in enumerate(params):
        _need(w, f"param{i}")
    _need(rope[0], "rope_cos", torch.float32); _need(rope[1], "rope_sin", torch.float32)
    assert rope[0].shape[0] >= T
    y = torch.empty_like(x) if out_rows is None else torch.empty((out_rows.numel(), Cc), dtype=bf16, device=x.device)
    act = torch.empty(int(L.lib().obte_block_act_bytes_p(B, T, Cc, H, float(dropout_p))), dtype=torch.uint8, device=x.device)
    d = _block_desc(B, T, Cc, H, params, rope, mask, dropout_p, dropout_seed, out_rows=out_rows)
    L.check(L.lib().obte_block_fwd(C.byref(d), _ptr(x), _ptr(y), _ptr(act), _stream()), "obte_block_fwd")
    return y, act


def block_bwd(x, dy, act, params, rope, H, mask: MaskSpec, accumulate_into=None, dropout_p=0.0, dropout_seed=0, ln_partials=None,
              ln_partial_mode=0, out_rows=None, dy_masked=None, dx_mask_seed=None):
    """accumulate_into: optional list of 6 tensors-or-None (same order as params).  When the four matrix entries are all
    given, their gradients are added into those tensors in place and the corresponding returned grads are None; the
    same, independently, for the two LayerNorm weights (entries 0 and 3).
    ln_partials=(buf1, buf2) + ln_partial_mode (L.LN_PARTIAL_*): the two LayerNorm weight gradients are accumulated over
    calls in those fp32 buffers instead (see layernorm_bwd); their returned grads are None except with LN_PARTIAL_LAST.
    Dropout hand-off between blocks (include/omnibiote_hip.h, obte_block_desc::dy_masked): dy_masked = dropout(dy) under THIS
    block's (seed, site 3) mask if the block above already wrote it; dx_mask_seed = the seed of the block BELOW: then
    dropout(dx) under that block's mask is written too and returned as a third value (else None)."""
    _need(x, "x"); _need(dy, "dy")
    B, T, Cc = x.shape
    assert dy.numel() == (B * T if out_rows is None else out_rows.numel()) * Cc, "dy: one row per (wanted) position"
    ws = torch.empty(int(L.lib().obte_block_bwd_ws_bytes(B, T, Cc, H)), dtype=torch.uint8, device=x.device)
    dx = torch.empty_like(x)
    acc = accumulate_into is not None and all(accumulate_into[i] is not None for i in (1, 2, 4, 5))
    acc_ln = (not ln_partial_mode) and accumulate_into is not None and all(accumulate_into[i] is not None for i in (0, 3))
    if ln_partial_mode:
        for t in ln_partials:
            _need(t, "ln_partials", torch.float32)
            assert t.numel() == L.lib().obte_layernorm_bwd_ws_rows() * Cc
    grads = []
    for i, w in enumerate(params):
        if (acc and i in (1, 2, 4, 5)) or (acc_ln and i in (0, 3)):
            g = accumulate_into[i]
            _need(g, "grad"); assert g.shape == w.shape
            grads.append(g)
        else:
            grads.append(torch.empty_like(w))
    dx_masked = torch.empty_like(x) if (dx_mask_seed is not None and dropout_p > 0.0) else None
    if dy_masked is not None:
        _need(dy_masked, "dy_masked"); assert dy_masked.numel() == dy.numel() and dropout_p > 0.0
    d = _block_desc(B, T, Cc, H, params, rope, mask, dropout_p, dropout_seed, ln_partials, ln_partial_mode, out_rows=out_rows,
                    dy_masked=dy_masked, dx_masked=dx_masked, dx_mask_seed=dx_mask_seed or 0)
    L.check(L.lib().obte_block_bwd_acc(C.byref(d), _ptr(x), _ptr(dy), _ptr(act), _ptr(ws), _ptr(dx), *[_ptr(g) for g in grads],
                                        int(acc) + 2 * int(acc_ln), _stream()), "obte_block_bwd")
    ln_none = ln_partial_mode in (L.LN_PARTIAL_FIRST, L.LN_PARTIAL_MORE)
    grads = [None if ((acc and i in (1, 2, 4, 5)) or ((acc_ln or ln_none) and i in (0, 3))) else g for i, g in enumerate(grads)]
    if dx_mask_seed is not None:
        return dx, grads, dx_masked
    return dx, grads
