"""GPU parity tests, kernel level: every C-ABI entry point against the CPU oracle / a torch fp32 reference of the
same op on the same seeded inputs.  Floating-point path, so the bar is a stated tolerance: inputs are bf16,
accumulation fp32, outputs rounded to bf16 once — |got - ref| <= RTOL*|ref| + ATOL with RTOL = 2^-7 (two bf16
ulps) unless a test says otherwise; ATOL scales with the magnitude of the reduction."""
import math

import numpy as np
import pytest
import torch

import omnibiote_ref as R

pytestmark = pytest.mark.gpu

DEV = "cuda"
BF = torch.bfloat16
RTOL = 2.0 ** -7


def ops():
    from omnibiote_amd import ops as o
    return o


def L():
    from omnibiote_amd import _lib
    return _lib


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def close(got, ref, atol, rtol=RTOL, what=""):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - ref).abs()
    bound = rtol * ref.abs() + atol
    bad = err > bound
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off; max err {err.max():.4g} (ref max {ref.abs().max():.4g})"


# ------------------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(256, 384, 128), (200, 136, 192), (128, 128, 64), (1024, 1024, 1024), (77, 512, 256)])
def test_gemm_forward_nt(M, N, K):
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2)
    ref = x.float() @ w.float().t()
    got = ops().linear_fwd(x.to(DEV), w.to(DEV))
    close(got, ref, atol=0.02 * math.sqrt(K), what="gemm NT")


def test_gemm_identity_asymmetric():
    """A = I against an asymmetric B catches a transposed C write or a wrong operand lane map exactly."""
    K = 128
    eye = torch.eye(K).to(BF)
    b = (torch.arange(256 * K).reshape(256, K) % 251 - 125).float().to(BF)  # exactly representable
    got = ops().linear_fwd(eye.to(DEV), b.to(DEV))  # [128, 256] = I @ b^T
    assert torch.equal(got.cpu().float(), b.float().t())
    got = ops().linear_fwd(b.to(DEV), eye.to(DEV))  # [256, 128] = b @ I
    assert torch.equal(got.cpu().float(), b.float())


@pytest.mark.parametrize("M,N,K", [(256, 128, 384), (300, 256, 384), (130, 512, 128), (64, 1024, 2048)])
def test_gemm_dgrad_nn(M, N, K):
    """dx[M,N] = dy[M,K] W[K,N]: B is read k-strided through the transposing LDS reads."""
    dy, w = rnd(M, K, seed=3), rnd(K, N, seed=4)
    ref = dy.float() @ w.float()
    got = ops().linear_dgrad(dy.to(DEV), w.to(DEV))
    close(got, ref, atol=0.02 * math.sqrt(K), what="gemm NN")


def test_gemm_dgrad_identity():
    N = 256
    w = (torch.arange(128 * N).reshape(128, N) % 241 - 120).float().to(BF)
    eye = torch.eye(128).to(BF)
    got = ops().linear_dgrad(eye.to(DEV), w.to(DEV))
    assert torch.equal(got.cpu().float(), w.float())


@pytest.mark.parametrize("M,N,K", [(384, 256, 256), (384, 256, 300), (128, 136, 77), (1024, 256, 2048), (512, 128, 154)])
def test_gemm_wgrad_tn(M, N, K):
    """dW[M,N] = dy[K,M]^T x[K,N]: both operands k-strided; K (tokens) may be ragged — the tail is zero-filled
    by the buffer bounds check."""
    dy, x = rnd(K, M, seed=5), rnd(K, N, seed=6)
    ref = dy.float().t() @ x.float()
    got = ops().linear_wgrad(dy.to(DEV), x.to(DEV))
    close(got, ref, atol=0.02 * math.sqrt(K), what="gemm TN")


def test_gemm_wgrad_identity():
    x = (torch.arange(128 * 256).reshape(128, 256) % 239 - 119).float().to(BF)
    eye = torch.eye(128).to(BF)
    got = ops().linear_wgrad(eye.to(DEV), x.to(DEV))  # I^T x
    assert torch.equal(got.cpu().float(), x.float())


def test_gemm_epilogues_and_alpha():
    M, N, K = 200, 256, 128
    x, w, r = rnd(M, K, seed=7), rnd(N, K, seed=8, scale=0.2), rnd(M, N, seed=9)
    acc = x.float() @ w.float().t()
    o = ops()
    # alpha
    close(o.linear_fwd(x.to(DEV), w.to(DEV), alpha=1 / 42.0), acc / 42.0, atol=2e-3, what="alpha")
    # residual add: bf16(aux + bf16(acc))
    ref = (r.float() + acc.to(BF).float())
    close(o.linear_fwd(x.to(DEV), w.to(DEV), epilogue=L().EPI_ADD, aux=r.to(DEV)), ref, atol=0.03, what="add")
    # GELU: h = bf16(acc); d = gelu'(h), d2 = gelu(h)
    d, d2 = o.linear_fwd(x.to(DEV), w.to(DEV), epilogue=L().EPI_GELU)
    hb = acc.to(BF).float().requires_grad_(True)
    act = R.gelu_erf(hb)
    act.sum().backward()
    close(d2, act, atol=0.03, what="gelu act")       # atol covers an ulp of difference in the bf16 rounding of acc
    close(d, hb.grad, atol=0.02, what="gelu derivative")
    # GELU backward on dgrad: multiply by the stored derivative
    dy, w2, h = rnd(M, K, seed=10), rnd(K, N, seed=11, scale=0.2), rnd(M, N, seed=12)
    ref = (dy.float() @ w2.float()).to(BF).float() * h.float()
    got = o.linear_dgrad(dy.to(DEV), w2.to(DEV), epilogue=L().EPI_GELU_BWD, aux=h.to(DEV))
    close(got, ref, atol=0.03, what="gelu bwd")


def test_gemm_rejects_bad_shapes():
    o = ops()
    with pytest.raises(RuntimeError):
        o.linear_fwd(rnd(64, 100).to(DEV), rnd(64, 100).to(DEV))  # k-contiguous K % 64 != 0
    with pytest.raises(RuntimeError):
        o.linear_fwd(rnd(64, 64), rnd(64, 64))  # CPU tensors: no fallback


# -------------------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("rows,cols", [(37, 128), (64, 256), (513, 1024), (9, 2048), (5, 1032)])
def test_layernorm_fwd_bwd(rows, cols):
    x, w, dy, dres = rnd(rows, cols, seed=1, scale=2.0), (1 + 0.3 * rnd(cols, seed=2).float()).to(BF), rnd(rows, cols, seed=3), rnd(rows, cols, seed=4)
    xf, wf = x.float().requires_grad_(True), w.float().requires_grad_(True)
    yref = R.layer_norm(xf, wf)
    yref.backward(dy.float())
    o = ops()
    y, mean, rstd = o.layernorm_fwd(x.to(DEV), w.to(DEV))
    close(y, yref, atol=2e-3, what="ln fwd")
    close(mean, x.float().mean(1), atol=1e-5, rtol=1e-5, what="ln mean")
    dx, dw = o.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd)
    close(dx, xf.grad, atol=4e-3, what="ln dx")
    close(dw, wf.grad, atol=0.02 * math.sqrt(rows), what="ln dw")
    dx2, _ = o.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd, dresid=dres.to(DEV))
    close(dx2, xf.grad + dres.float(), atol=8e-3, what="ln dx+resid")
    # in-place accumulation: exactly autograd's bf16(old + new)
    old = rnd(cols, seed=5)
    slot = old.clone().to(DEV)
    _, none = o.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd, accumulate_into=slot)
    assert none is None
    assert torch.equal(slot.cpu(), (old.float() + dw.cpu().float()).to(BF))


@pytest.mark.parametrize("rows,cols", [(8192, 1024), (8197, 1024), (4101, 2048), (2051, 256)])
def test_layernorm_pipelined_rows_and_dropped_gradient(rows, cols):
    """Row counts at which every wave of the LayerNorm kernels walks several rows (the pipelined forms: next row's loads in
    flight under the current row's reductions), ragged tails included, against the oracle; and the backward's second output,
    dropout(dx) under the library's counter-based mask, bit for bit what the stand-alone dropout pass gives."""
    x, w, dy, dres = rnd(rows, cols, seed=1, scale=2.0), (1 + 0.3 * rnd(cols, seed=2).float()).to(BF), rnd(rows, cols, seed=3), rnd(rows, cols, seed=4)
    xf, wf = x.float().requires_grad_(True), w.float().requires_grad_(True)
    yref = R.layer_norm(xf, wf)
    yref.backward(dy.float())
    o = ops()
    y, mean, rstd = o.layernorm_fwd(x.to(DEV), w.to(DEV))
    close(y, yref, atol=2e-3, what="ln fwd")
    close(mean, x.float().mean(1), atol=1e-5, rtol=1e-5, what="ln mean")
    close(rstd, 1.0 / torch.sqrt(x.float().var(1, unbiased=False) + 1e-5), atol=1e-5, rtol=1e-4, what="ln rstd")
    dx, dw = o.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd, dresid=dres.to(DEV))
    close(dx, xf.grad + dres.float(), atol=8e-3, what="ln dx+resid")
    close(dw, wf.grad, atol=0.02 * math.sqrt(rows), what="ln dw")
    p, seed, site = 0.1, 1234567, 2
    dx2, dw2, dxd = o.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd, dresid=dres.to(DEV), dropout=(p, seed, site))
    assert torch.equal(dx2, dx) and torch.equal(dw2, dw)
    assert torch.equal(dxd, o.dropout(dx, p, seed, site))
    kept = (dxd != 0).float().mean().item()
    assert abs(kept - 0.9) < 0.01
    # partial-sum mode with the dropped output: same dx / dropped dx, weight gradient from the fp32 partials
    part = o.ln_partials_buffer(cols, DEV)
    from omnibiote_amd import _lib as L
    dx3, none, dxd3 = o.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd, dresid=dres.to(DEV), partials=part, partial_mode=L.LN_PARTIAL_FIRST,
                                      dropout=(p, seed, site))
    assert none is None and torch.equal(dx3, dx) and torch.equal(dxd3, dxd)
    _, dw3, _ = o.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd, dresid=dres.to(DEV), partials=part, partial_mode=L.LN_PARTIAL_LAST,
                                dropout=(p, seed, site))
    close(dw3, 2.0 * wf.grad, atol=0.04 * math.sqrt(rows), what="ln dw from partials")


# ------------------------------------------------------------------------------------------------------- RoPE
@pytest.mark.parametrize("hs", [64, 128])
@pytest.mark.parametrize("mode", ["complex", "cos_only"])
def test_rope(hs, mode):
    B, T, H = 2, 37, 2
    C = H * hs
    qkv = rnd(B, T, 3 * C, seed=5)
    tab = R.rope_table(hs, 64)
    if mode == "cos_only":
        tab = R.cast_rope_table(tab, BF)
    from omnibiote_amd.model import rope_tables
    cos, sin = rope_tables(tab.to(DEV))
    got = ops().rope_qk_(qkv.clone().to(DEV), cos, sin, B, T, H, hs).cpu()
    q, k, v = qkv.split(C, dim=2)
    rq = R.apply_rope(q.reshape(B, T, H, hs), tab).reshape(B, T, C)
    rk = R.apply_rope(k.reshape(B, T, H, hs), tab).reshape(B, T, C)
    ref = torch.cat([rq, rk, v], dim=2)
    # fp32 math on both sides, one bf16 rounding: allow 1 bf16 ulp (2^-7 relative at the bottom of a binade) for
    # fma-contraction differences
    close(got, ref, atol=1e-6, rtol=2.0 ** -7, what="rope")
    assert torch.equal(got[..., 2 * C:], v)
    # inverse is the transpose: <R x, y> == <x, R^T y>
    y = rnd(B, T, 3 * C, seed=6)
    ry = ops().rope_qk_(y.clone().to(DEV), cos, sin, B, T, H, hs, inverse=True).cpu()
    lhs = (got.float()[..., :2 * C] * y.float()[..., :2 * C]).sum()
    rhs = (qkv.float()[..., :2 * C] * ry.float()[..., :2 * C]).sum()
    assert abs(lhs - rhs) <= 2e-2 * abs(lhs) + 1.0


# -------------------------------------------------------------------------------------------------- attention
def _blocks_to_masks(tokens, T):
    blocks = R.document_blocks(tokens)
    dense = R.dense_mask_from_blocks(blocks, T)
    ranges = torch.from_numpy(R.key_ranges_from_blocks(blocks, T))
    return dense, ranges


def _attn_case(B, T, H, hs, seed):
    C = H * hs
    qkv = rnd(B, T, 3 * C, seed=seed)
    q, k, v = [t.reshape(B, T, H, hs).transpose(1, 2).float() for t in qkv.split(C, dim=2)]
    return qkv, q, k, v


@pytest.mark.parametrize("hs", [64, 128])
@pytest.mark.parametrize("T", [64, 77, 130, 256])
@pytest.mark.parametrize("mode", ["none", "ranges", "dense"])
def test_attention_fwd_bwd(hs, T, mode):
    B, H = 2, 2
    C = H * hs
    scale = 8.0 / C
    qkv, q, k, v = _attn_case(B, T, H, hs, seed=hs + T)
    rng = np.random.default_rng(T)
    tokens = rng.integers(20, 100, size=(B, T))
    tokens[0, [T // 3, T // 2]] = R.EOS_TOKEN
    tokens[1, [5, T - 10]] = R.EOS_TOKEN
    dense, ranges = _blocks_to_masks(tokens, T)
    o = ops()
    mask_add, spec = None, None
    if mode == "ranges":
        mask_add, spec = dense.unsqueeze(1), o.MaskSpec(ranges=ranges.to(DEV))
    elif mode == "dense":
        mask_add = dense.unsqueeze(1)
        spec = o.MaskSpec.from_user(dense.to(BF).to(DEV).unsqueeze(1).expand(B, H, T, T), B, T, H, DEV)
    qf, kf, vf = q.requires_grad_(True), k.requires_grad_(True), v.requires_grad_(True)
    ref = R.attention(qf, kf, vf, scale, mask_add)  # (B,H,T,hs)
    d_o = rnd(B, T, C, seed=99)
    ref.backward(d_o.reshape(B, T, H, hs).transpose(1, 2).float())
    got, lse = o.attn_fwd(qkv.to(DEV), B, T, H, hs, scale, spec)
    close(got, ref.transpose(1, 2).reshape(B, T, C), atol=6e-3, what=f"attn fwd {mode}")
    att = (q @ k.transpose(-2, -1)) * scale
    if mask_add is not None:
        att = att + mask_add
    close(lse, torch.logsumexp(att, dim=-1), atol=2e-3, rtol=1e-3, what="lse")
    dqkv = o.attn_bwd(qkv.to(DEV), got, d_o.to(DEV), lse, B, T, H, hs, scale, spec)
    dref = torch.cat([g.transpose(1, 2).reshape(B, T, C) for g in (qf.grad, kf.grad, vf.grad)], dim=2)
    close(dqkv, dref, atol=1.5e-2, rtol=2.0 ** -6, what=f"attn bwd {mode}")


def test_attention_backward_hand_off_failure_is_reported_not_silent():
    """A wait of the one-kernel backward's dQ hand-off chain that gives up must be a hard failure the host sees: the kernel ORs
    OBTE_STATUS_ATTN_BWD_HANDOFF into the library's device status word (pinned host memory) and obte_device_status / the Python
    check raise at the next synchronising point.  Forced here through the test hook obte_fault_inject(1): the counter of slice 0
    of (batch 0, head 0) is never added to and waits give up after 2^10 polls instead of 2^20.  Before and after, the same call
    is clean and bitwise reproducible (the word is sticky until read, then cleared; a failure leaves no state behind)."""
    from omnibiote_amd import _lib as L
    B, H, T, hs = 1, 2, 1024, 128
    C = H * hs
    scale = 8.0 / C
    qkv, q, k, v = _attn_case(B, T, H, hs, seed=11)
    d_o = rnd(B, T, C, seed=12)
    o = ops()
    got, lse = o.attn_fwd(qkv.to(DEV), B, T, H, hs, scale, None)
    good = o.attn_bwd(qkv.to(DEV), got, d_o.to(DEV), lse, B, T, H, hs, scale, None)
    torch.cuda.synchronize()
    assert L.lib().obte_device_status(0) == 0
    prev = L.lib().obte_fault_inject(1)
    try:
        o.attn_bwd(qkv.to(DEV), got, d_o.to(DEV), lse, B, T, H, hs, scale, None)
        torch.cuda.synchronize()
    finally:
        L.lib().obte_fault_inject(prev)
    assert L.lib().obte_device_status(0) & L.STATUS_ATTN_BWD_HANDOFF, "the timed-out hand-off went unreported"
    with pytest.raises(L.DeviceStatusError, match="hand-off"):
        L.check_device_status("forced failure")
    assert L.lib().obte_device_status(0) == 0                     # read once, cleared
    again = o.attn_bwd(qkv.to(DEV), got, d_o.to(DEV), lse, B, T, H, hs, scale, None)
    torch.cuda.synchronize()
    L.check_device_status("after the forced failure")
    assert torch.equal(good, again)


def test_attention_backward_one_kernel_under_contention_keeps_its_chain():
    """The regime the hand-off's forward progress was questioned in: 32 rows x 8 heads at T = 1024 = 1024
    workgroups of one (4 x the CUs), launched on two streams at once while a third stream holds CUs with a long-running
    kernel of its own.  Every launch must finish with a clean status word and give, bit for bit, what it gives alone."""
    from omnibiote_amd import _lib as L
    B, H, T, hs = 32, 8, 1024, 128
    C = H * hs
    scale = 8.0 / C
    g = torch.Generator(device=DEV).manual_seed(3)
    qkv = torch.randn(B, T, 3 * C, device=DEV, generator=g).to(BF)
    d_o = (torch.randn(B, T, C, device=DEV, generator=g) * 0.1).to(BF)
    o = ops()
    out, lse = o.attn_fwd(qkv, B, T, H, hs, scale, None)
    alone = o.attn_bwd(qkv, out, d_o, lse, B, T, H, hs, scale, None)
    torch.cuda.synchronize()
    L.check_device_status("alone")
    s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    big = torch.randn(8192, 8192, device=DEV, dtype=torch.float32)
    res = {}
    for rep in range(2):
        with torch.cuda.stream(s3):                              # a CU hog of another kind (fp32 matmuls: long-running workgroups)
            for _ in range(6):
                big = big @ big * 1e-4
        for name, st in (("a", s1), ("b", s2)):
            with torch.cuda.stream(st):
                for _ in range(3):
                    res[name] = o.attn_bwd(qkv, out, d_o, lse, B, T, H, hs, scale, None)
        torch.cuda.synchronize()
        L.check_device_status(f"contended launches, round {rep}")
        assert torch.equal(res["a"], alone) and torch.equal(res["b"], alone)


@pytest.mark.parametrize("T,mode", [(600, "ranges"), (600, "none"), (1024, "ranges"), (333, "ranges")])
def test_attention_backward_one_kernel_form(T, mode):
    """The one-kernel backward (head size 128, key ranges or no mask, no dropout; csrc/attention_bwd_fused.hip) over several
    256-key blocks, ragged lengths and multi-document rows — against the oracle (model.py:115-146 differentiated by autograd),
    against the dQ + dK/dV kernel pair (the two forms sum in different orders: equal to rounding, not bitwise), and against
    itself run twice (no atomics anywhere: bitwise)."""
    B, H, hs = 2, 3, 128
    C = H * hs
    scale = 8.0 / C
    qkv, q, k, v = _attn_case(B, T, H, hs, seed=T)
    rng = np.random.default_rng(T)
    tokens = rng.integers(20, 100, size=(B, T))
    tokens[0, [T // 5, T // 2, T // 2 + 40]] = R.EOS_TOKEN      # documents that start and end inside / across key blocks
    tokens[1, [3, T - 7]] = R.EOS_TOKEN
    dense, ranges = _blocks_to_masks(tokens, T)
    o = ops()
    mask_add, spec = (dense.unsqueeze(1), o.MaskSpec(ranges=ranges.to(DEV))) if mode == "ranges" else (None, None)
    qf, kf, vf = q.requires_grad_(True), k.requires_grad_(True), v.requires_grad_(True)
    ref = R.attention(qf, kf, vf, scale, mask_add)
    d_o = rnd(B, T, C, seed=5)
    ref.backward(d_o.reshape(B, T, H, hs).transpose(1, 2).float())
    dref = torch.cat([g.transpose(1, 2).reshape(B, T, C) for g in (qf.grad, kf.grad, vf.grad)], dim=2)
    got, lse = o.attn_fwd(qkv.to(DEV), B, T, H, hs, scale, spec)
    cos, sin = None, None
    one = o.attn_bwd(qkv.to(DEV), got, d_o.to(DEV), lse, B, T, H, hs, scale, spec)
    two = o.attn_bwd(qkv.to(DEV), got, d_o.to(DEV), lse, B, T, H, hs, scale, spec, one_kernel=False)
    close(one, dref, atol=1.5e-2, rtol=2.0 ** -6, what=f"one-kernel bwd {mode} T={T}")
    close(one, two.float().cpu(), atol=4e-3, rtol=2.0 ** -7, what="one-kernel vs kernel pair")
    again = o.attn_bwd(qkv.to(DEV), got, d_o.to(DEV), lse, B, T, H, hs, scale, spec)
    assert torch.equal(one, again), "one-kernel backward is not run-to-run bitwise"
    # with the inverse RoPE of the epilogues (what the block passes): the same rotation of the dq and dk thirds in both forms
    tab = torch.randn(T, hs // 2, generator=torch.Generator().manual_seed(1))
    rope = (torch.cos(tab).to(DEV), torch.sin(tab).to(DEV))
    one_r = o.attn_bwd(qkv.to(DEV), got, d_o.to(DEV), lse, B, T, H, hs, scale, spec, rope=rope)
    two_r = o.attn_bwd(qkv.to(DEV), got, d_o.to(DEV), lse, B, T, H, hs, scale, spec, rope=rope, one_kernel=False)
    close(one_r, two_r.float().cpu(), atol=4e-3, rtol=2.0 ** -7, what="one-kernel vs kernel pair, inverse RoPE")


@pytest.mark.parametrize("T,mode", [(600, "ranges"), (333, "none"), (1024, "ranges")])
def test_attention_dropout_keep_bits_from_the_forward(T, mode):
    """With dropout on, the forward can leave its keep decisions behind in key-major order (one word per key and 32-query
    slice); the key-major backward kernel then reads them instead of hashing every (query, key) pair again.  Same decisions,
    same arithmetic: the backward with the bits equals the backward that hashes bit for bit, and the forward's outputs do not
    depend on whether the bits are written."""
    B, H, hs, p, seed = 2, 3, 128, 0.1, 0x5EED
    C = H * hs
    scale = 8.0 / C
    qkv, _, _, _ = _attn_case(B, T, H, hs, seed=T + 1)
    tokens = np.random.default_rng(T).integers(20, 100, size=(B, T))
    tokens[0, [T // 5, T // 2]] = R.EOS_TOKEN
    tokens[1, [7, T - 9]] = R.EOS_TOKEN
    _, ranges = _blocks_to_masks(tokens, T)
    o = ops()
    spec = o.MaskSpec(ranges=ranges.to(DEV)) if mode == "ranges" else None
    d_o = rnd(B, T, C, seed=6).to(DEV)
    qd = qkv.to(DEV)
    out0, lse0 = o.attn_fwd(qd, B, T, H, hs, scale, spec, dropout_p=p, dropout_seed=seed)
    out1, lse1, bits = o.attn_fwd(qd, B, T, H, hs, scale, spec, dropout_p=p, dropout_seed=seed, keep_bits=True)
    assert torch.equal(out0, out1) and torch.equal(lse0, lse1) and bits is not None
    hashed = o.attn_bwd(qd, out0, d_o, lse0, B, T, H, hs, scale, spec, dropout_p=p, dropout_seed=seed)
    from_bits = o.attn_bwd(qd, out0, d_o, lse0, B, T, H, hs, scale, spec, dropout_p=p, dropout_seed=seed, drop_bits=bits, one_kernel=False)
    assert torch.equal(hashed, from_bits)
    # the ONE-kernel backward takes the dropout path from the same bits (round 5): the kernel pair's gradients to rounding (the two
    # forms sum in different orders), bitwise run to run, inverse RoPE included
    fused = o.attn_bwd(qd, out0, d_o, lse0, B, T, H, hs, scale, spec, dropout_p=p, dropout_seed=seed, drop_bits=bits)
    close(fused, from_bits.float().cpu(), atol=4e-3, rtol=2.0 ** -7, what=f"one-kernel dropout backward vs kernel pair, T={T} {mode}")
    assert torch.equal(fused, o.attn_bwd(qd, out0, d_o, lse0, B, T, H, hs, scale, spec, dropout_p=p, dropout_seed=seed, drop_bits=bits))
    tab = torch.randn(T, hs // 2, generator=torch.Generator().manual_seed(2))
    rope = (torch.cos(tab).to(DEV), torch.sin(tab).to(DEV))
    fused_r = o.attn_bwd(qd, out0, d_o, lse0, B, T, H, hs, scale, spec, rope=rope, dropout_p=p, dropout_seed=seed, drop_bits=bits)
    pair_r = o.attn_bwd(qd, out0, d_o, lse0, B, T, H, hs, scale, spec, rope=rope, dropout_p=p, dropout_seed=seed, drop_bits=bits, one_kernel=False)
    close(fused_r, pair_r.float().cpu(), atol=4e-3, rtol=2.0 ** -7, what="one-kernel dropout backward vs kernel pair, inverse RoPE")
    # the bits are the oracle's mask: word (b*H + h, t, key) bit i = keep(row (b, h, 32 t + i), key)
    nsl = (T + 31) // 32
    w = bits.reshape(B * H, nsl, T).cpu().numpy().astype(np.uint32)
    rows = (np.arange(B * H, dtype=np.uint64)[:, None] * np.uint64(T) + np.arange(T, dtype=np.uint64)[None, :]).reshape(-1)
    keep = R.dropout_keep(rows[:, None], np.arange(T, dtype=np.uint64)[None, :], p, seed, 1).reshape(B * H, T, T)   # [bh, q, key]
    got = ((w[:, np.arange(T) // 32, :] >> (np.arange(T) % 32).astype(np.uint32)[None, :, None]) & 1).astype(bool)  # [bh, q, key]
    lo, hi = ranges.numpy()[..., 0], ranges.numpy()[..., 1]
    for b in range(B):
        for q in range(0, T, 37):
            ks, ke = (int(lo[b, q]), int(hi[b, q])) if mode == "ranges" else (0, T)
            for hh in range(H):
                assert np.array_equal(got[b * H + hh, q, ks:ke], keep[b * H + hh, q, ks:ke]), (b, hh, q)


def test_attention_backward_one_kernel_beside_another_stream():
    """The one-kernel backward at the benchmark's shape (B = 8, H = 8, T = 1024, multi-document rows) repeated while a second
    stream keeps the memory system busy with GEMMs: every repeat equals the quiet result bit for bit.  (Round 4: the loop's
    lse / delta load was an asm load with a register destination that hipcc copied in front of the wait covering it — the
    previous slice's row constants whenever the load was late, i.e. only beside other work: tools/fused_race.py.)"""
    B, H, T, hs = 8, 8, 1024, 128
    g = torch.Generator(device=DEV).manual_seed(0)
    qkv = torch.randn(B, T, 3 * H * hs, device=DEV, generator=g).to(BF)
    d_o = torch.randn(B, T, H * hs, device=DEV, generator=g).to(BF)
    tok = np.random.default_rng(3).integers(20, 100, size=(B, T))
    for b in range(B):
        tok[b, np.random.default_rng(10 + b).choice(np.arange(8, T - 8), size=3, replace=False)] = R.EOS_TOKEN
    _, ranges = _blocks_to_masks(tok, T)
    o = ops()
    spec = o.MaskSpec(ranges=ranges.to(DEV))
    scale = 8.0 / (H * hs)
    out, lse = o.attn_fwd(qkv, B, T, H, hs, scale, spec)
    quiet = o.attn_bwd(qkv, out, d_o, lse, B, T, H, hs, scale, spec).clone()
    pair = o.attn_bwd(qkv, out, d_o, lse, B, T, H, hs, scale, spec, one_kernel=False)
    close(quiet, pair.float().cpu(), atol=4e-3, rtol=2.0 ** -7, what="one-kernel vs kernel pair")
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    x, w = torch.randn(8192, 1024, device=DEV).to(BF), torch.randn(4096, 1024, device=DEV).to(BF)
    differ = 0
    for r in range(24):
        with torch.cuda.stream(side):
            for _ in range(3):
                o.linear_fwd(x, w)
                if r % 2:
                    o.attn_fwd(qkv, B, T, H, hs, scale, spec)
        got = o.attn_bwd(qkv, out, d_o, lse, B, T, H, hs, scale, spec)
        torch.cuda.synchronize()
        differ += int(not torch.equal(got, quiet))
    assert differ == 0, f"{differ}/24 repeats beside another stream differ from the quiet result"


@pytest.mark.parametrize("hs", [64, 128])
def test_attention_dense_mask_bounds_edge_cases(hs):
    """Dense additive masks reach the kernels with conservative loop bounds (obte_mask_bounds).  The cases the bounds
    must not break: a row that masks every key (the reference's softmax then runs over the raw scores), an asymmetric
    mask (key-side bounds differ from query-side), masks that differ per head, a soft bias (-5) that is not a mask, and
    large fully-masked regions that do get skipped."""
    B, H, T = 2, 2, 320
    C = H * hs
    scale = 8.0 / C
    qkv, q, k, v = _attn_case(B, T, H, hs, seed=7 * hs)
    neg = -1e9
    m = torch.full((B, H, T, T), neg)
    # batch 0: two blocks [0,100) and [100,320) on both heads, but head 1 additionally lets queries 10..19 see keys 250..259
    m[0, :, :100, :100] = 0; m[0, :, 100:, 100:] = 0
    m[0, 1, 10:20, 250:260] = 0
    m[0, 0, 40, :] = neg                       # a fully masked row on head 0 only
    # batch 1: asymmetric band (query q sees keys q-30 .. q), a soft bias region, and a fully masked row on both heads
    for qq in range(T):
        m[1, :, qq, max(0, qq - 30):qq + 1] = 0
    m[1, :, 200:210, 0:5] = -5.0
    m[1, :, 77, :] = neg
    mb = m.to(BF)
    o = ops()
    spec = o.MaskSpec.from_user(mb.to(DEV), B, T, H, DEV)
    assert spec.dense is not None and spec.ranges is not None and spec.qbounds is not None
    kb, qb = spec.ranges.cpu(), spec.qbounds.cpu()
    assert kb[0, 40].tolist() == [0, T] and kb[1, 77].tolist() == [0, T]          # fully masked rows: nothing may be skipped
    assert kb[0, 15].tolist() == [0, 260] and kb[0, 150].tolist() == [100, T]      # union over heads
    assert kb[1, 300].tolist() == [270, 301] and kb[1, 205].tolist() == [0, 206]   # the -5 bias is not a mask
    assert qb[0, 255].tolist() == [10, T] and qb[1, 0].tolist()[0] == 0 and qb[1, 0].tolist()[1] == 210
    assert qb[1, 319].tolist() == [77, 320]                                         # only the fully masked row and itself
    qf, kf, vf = q.requires_grad_(True), k.requires_grad_(True), v.requires_grad_(True)
    ref = R.attention(qf, kf, vf, scale, mb.float())
    d_o = rnd(B, T, C, seed=98)
    # fully masked rows: the forward matches the reference (uniform softmax over scores that vanish beside -1e9); the
    # backward gives them zero weight by contract (their lse is +inf), so the reference's upstream gradient is zeroed there
    d_ref = d_o.reshape(B, T, H, hs).transpose(1, 2).float().clone()
    d_ref[0, 0, 40] = 0; d_ref[1, :, 77] = 0
    ref.backward(d_ref)
    got, lse = o.attn_fwd(qkv.to(DEV), B, T, H, hs, scale, spec)
    close(got, ref.transpose(1, 2).reshape(B, T, C), atol=6e-3, what="dense+bounds fwd")
    assert torch.isinf(lse[0, 0, 40]) and torch.isinf(lse[1, :, 77]).all() and torch.isfinite(lse[0, 1, 40])
    dqkv = o.attn_bwd(qkv.to(DEV), got, d_o.to(DEV), lse, B, T, H, hs, scale, spec)
    dref = torch.cat([g.transpose(1, 2).reshape(B, T, C) for g in (qf.grad, kf.grad, vf.grad)], dim=2)
    close(dqkv, dref, atol=1.5e-2, rtol=2.0 ** -6, what="dense+bounds bwd")
    # and the bounds change nothing: bitwise the same as the dense path without them
    plain = o.MaskSpec(dense=spec.dense)
    got2, lse2 = o.attn_fwd(qkv.to(DEV), B, T, H, hs, scale, plain)
    assert torch.equal(got, got2) and torch.equal(lse, lse2)
    assert torch.equal(dqkv, o.attn_bwd(qkv.to(DEV), got2, d_o.to(DEV), lse2, B, T, H, hs, scale, plain))


def test_attention_softmax_rescale_branch():
    """Spike one key late in the sequence so the running max jumps at a later KV tile: the online-softmax
    rescale of O and l must be exact (cdna guide rule 26)."""
    B, T, H, hs = 1, 256, 1, 128
    C = H * hs
    qkv = rnd(B, T, 3 * C, seed=123, scale=0.5)
    qkv[0, 200, C:2 * C] = 6.0 * torch.sign(qkv[0, 17, :C].float()).to(BF)  # key 200 aligns with query 17
    q, k, v = [t.reshape(B, T, H, hs).transpose(1, 2).float() for t in qkv.split(C, dim=2)]
    scale = 0.25
    ref = R.attention(q, k, v, scale)
    got, _ = ops().attn_fwd(qkv.to(DEV), B, T, H, hs, scale)
    close(got, ref.transpose(1, 2).reshape(B, T, C), atol=6e-3, what="rescale branch")


def test_attention_rows_sum_to_one_full_size():
    """BASELINE config 2 size (B=8,T=1024,H=8,hs=128): with V = 1 every output element must be 1 (softmax rows
    sum to 1) whatever Q, K and the mask are — a size-independent property."""
    B, T, H, hs = 8, 1024, 8, 128
    C = H * hs
    g = torch.Generator(device=DEV).manual_seed(0)
    qkv = torch.randn(B, T, 3 * C, device=DEV, generator=g).to(BF)
    qkv[..., 2 * C:] = 1.0
    starts = torch.arange(T, device=DEV) // 300 * 300
    ranges = torch.stack([starts, torch.clamp(starts + 300, max=T)], dim=1).to(torch.int32).unsqueeze(0).expand(B, T, 2).contiguous()
    for spec in (None, ops().MaskSpec(ranges=ranges)):
        o, lse = ops().attn_fwd(qkv, B, T, H, hs, 8.0 / C, spec)
        assert torch.isfinite(lse).all()
        assert (o.float() - 1.0).abs().max().item() <= 2.0 ** -7


# -------------------------------------------------------------------------------------------------- embedding
def test_embedding_fwd_bwd_with_heavy_duplicates():
    V, C, rows = 512, 256, 1000
    rng = np.random.default_rng(0)
    idx = rng.integers(0, V, size=rows)
    idx[rng.random(rows) < 0.3] = 2      # a MASK-like token hit ~300 times: spans many 32-row chunks
    idx[100:140] = 7                     # a run that starts and ends inside chunk boundaries after sorting
    idx = torch.from_numpy(idx)
    wte, dout = rnd(V, C, seed=1), rnd(rows, C, seed=2)
    o = ops()
    got = o.embedding_fwd(idx.to(DEV), wte.to(DEV))
    assert torch.equal(got.cpu(), wte[idx])
    dw = o.embedding_bwd(idx.to(DEV), dout.to(DEV), V)
    ref = torch.zeros(V, C).index_add_(0, idx, dout.float())
    close(dw, ref, atol=2e-2, what="embedding bwd")
    untouched = torch.ones(V, dtype=torch.bool); untouched[idx] = False
    assert (dw.cpu()[untouched] == 0).all()
    dw2 = o.embedding_bwd(idx.to(DEV), dout.to(DEV), V)
    assert torch.equal(dw, dw2), "embedding backward must be bitwise reproducible"


@pytest.mark.parametrize("rows", [1, 31, 32, 33, 64])
def test_embedding_bwd_edge_sizes(rows):
    V, C = 64, 128
    idx = torch.full((rows,), 5, dtype=torch.int64)
    if rows > 2:
        idx[-1] = 9
    dout = rnd(rows, C, seed=rows)
    dw = ops().embedding_bwd(idx.to(DEV), dout.to(DEV), V)
    ref = torch.zeros(V, C).index_add_(0, idx, dout.float())
    close(dw, ref, atol=2e-2, what="embedding bwd edge")


# --------------------------------------------------------------------------------------------- loss / optimizer
@pytest.mark.parametrize("rows,V", [(64, 512), (50, 65536), (33, 1000)])
def test_masked_ce(rows, V):
    logits = rnd(rows, V, seed=1, scale=2.0)
    tgt = torch.from_numpy(np.random.default_rng(1).integers(0, V, size=rows))
    mask = torch.from_numpy(np.random.default_rng(2).random(rows) < 0.3)
    mask[0] = True
    lf = logits.float().requires_grad_(True)
    ref = R.masked_lm_loss(lf, tgt, mask, n_accum=4)
    ref.backward()
    loss, dl = ops().masked_ce(logits.to(DEV), tgt.to(DEV), mask.to(DEV), 4)
    assert abs(loss.item() - ref.item()) <= 1e-4 * abs(ref.item()) + 1e-6
    close(dl, lf.grad, atol=2e-7, rtol=2.0 ** -7, what="dlogits")
    assert (dl.cpu()[~mask] == 0).all()
    # gradient rows sum to ~0 for masked rows (softmax - onehot)
    assert dl.float().sum(1).abs().max().item() < 1e-3


@pytest.mark.parametrize("V", [512, 65536, 65536 + 4096])   # register kernel <8>, <32>, and the two-pass kernel beyond 65 536
def test_masked_ce_rows_with_and_without_a_row_list(V):
    """obte_masked_ce_rows: a list of positions into dense logits, or (list = NULL) logits that already are the listed rows —
    against the oracle's masked loss on those rows, and the two forms against each other (bitwise)."""
    M, n, n_accum = 24, 9, 4
    logits = rnd(M, V, seed=3, scale=2.0)
    tgt = torch.from_numpy(np.random.default_rng(3).integers(0, V, size=M))
    rows = torch.tensor([0, 2, 3, 7, 11, 12, 20, 22, 23])
    lf = logits[rows].float().requires_grad_(True)
    ref = R.masked_lm_loss(lf, tgt[rows], torch.ones(n, dtype=torch.bool), n_accum)
    ref.backward()
    loss, dl = ops().masked_ce_rows(logits.to(DEV), tgt.to(DEV), rows.to(DEV), n_accum)
    loss_c, dl_c = ops().masked_ce_rows(logits[rows].contiguous().to(DEV), tgt[rows].to(DEV), None, n_accum)
    assert abs(loss.item() - ref.item()) <= 1e-4 * abs(ref.item()) + 1e-6
    close(dl, lf.grad, atol=2e-7, rtol=2.0 ** -7, what="dlogits rows")
    assert loss_c.item() == loss.item() and torch.equal(dl_c, dl)
    from omnibiote_amd import _lib as L
    with pytest.raises(RuntimeError):   # no list: the logits must be exactly the listed rows
        bad = torch.empty(1, dtype=torch.float32, device=DEV)
        L.check(L.lib().obte_masked_ce_rows(logits.to(DEV).data_ptr(), tgt.to(DEV).data_ptr(), None, bad.data_ptr(), 1.0, None,
                                            bad.data_ptr(), dl.data_ptr(), n, M, V, 0), "obte_masked_ce_rows")


def test_full_size_properties_of_the_memory_bound_kernels():
    """BASELINE config 2 sizes (8192 rows, 1024 features, 65536 classes), size-independent properties:
    LayerNorm rows have zero mean / unit variance for w = 1 and its input gradient is orthogonal to the all-ones vector
    and to x_hat; cross-entropy gradient rows vanish off the mask and sum to zero on it, and the loss of uniform logits is
    log V; an embedding round trip conserves mass."""
    o = ops()
    M, Cc, V = 8192, 1024, 65536
    g = torch.Generator(device=DEV).manual_seed(5)
    x = (torch.randn(M, Cc, device=DEV, generator=g) * 3 + 1).to(BF)
    w = torch.ones(Cc, device=DEV, dtype=BF)
    y, mean, rstd = o.layernorm_fwd(x, w)
    yf = y.float()
    assert yf.mean(1).abs().max().item() < 2e-2 and (yf.var(1, unbiased=False) - 1).abs().max().item() < 3e-2
    dy = torch.randn(M, Cc, device=DEV, generator=g).to(BF)
    dx, dw = o.layernorm_bwd(dy, x, w, mean, rstd)
    xhat = (x.float() - mean[:, None]) * rstd[:, None]
    assert dx.float().sum(1).abs().max().item() < 0.3 and (dx.float() * xhat).sum(1).abs().max().item() < 1.0   # bf16 rows of ~1e3 terms
    close(dw, (dy.float() * xhat).sum(0), atol=2.5, rtol=2.0 ** -6, what="ln dw full size")
    # cross entropy: uniform logits -> loss = log V exactly; gradient structure
    tgt = torch.randint(0, V, (M,), device=DEV, generator=g)
    mask = torch.rand(M, device=DEV, generator=g) < 0.15
    logits = torch.zeros(M, V, device=DEV, dtype=BF)
    loss, dl = o.masked_ce(logits, tgt, mask, 1)
    assert abs(loss.item() - math.log(V)) < 1e-3
    assert (dl[~mask] == 0).all()
    rows = dl[mask].float()
    assert rows.sum(1).abs().max().item() < 1e-4 and (rows.min(1).values < 0).all()     # one negative entry (the target) per row
    del logits, dl, rows
    # embedding: scatter-add of all-ones rows gives the token histogram
    idx = torch.randint(0, V, (M,), device=DEV, generator=g)
    dwte = o.embedding_bwd(idx, torch.ones(M, Cc, device=DEV, dtype=BF), V)
    hist = torch.bincount(idx, minlength=V).float()
    assert torch.equal(dwte[:, 0].float(), hist) and torch.equal(dwte[:, -1].float(), hist)


def test_adamw_matches_fp32_formula():
    n = 4096 + 8
    p, g = rnd(n, seed=1), rnd(n, seed=2, scale=0.1)
    m, v = rnd(n, seed=3, scale=0.01), (rnd(n, seed=4, scale=0.01).float() ** 2).to(BF)
    lr, b1, b2, eps, wd, step, clip = 1e-2, 0.9, 0.999, 1e-8, 1e-2, 3, 0.5
    gf = g.float() * clip
    pf = p.float() * (1 - lr * wd)
    mf = m.float() + (gf - m.float()) * (1 - b1)
    vf = v.float() * b2 + gf * gf * (1 - b2)
    denom = vf.sqrt() / math.sqrt(1 - b2 ** step) + eps
    pf = pf - (lr / (1 - b1 ** step)) * mf / denom
    pd, md, vd = p.clone().to(DEV), m.clone().to(DEV), v.clone().to(DEV)
    cc = torch.tensor([clip], device=DEV)
    ops().adamw_step_(pd, g.to(DEV), md, vd, lr, b1, b2, eps, wd, step, cc)
    close(pd, pf, atol=1e-6, rtol=2.0 ** -8, what="adamw p")
    close(md, mf, atol=1e-7, rtol=2.0 ** -8, what="adamw m")
    close(vd, vf, atol=1e-9, rtol=2.0 ** -8, what="adamw v")
    out = torch.zeros(1, device=DEV)
    ops().sumsq_(g.to(DEV), out)
    assert abs(out.item() - (g.float() ** 2).sum().item()) <= 1e-4 * out.item()


# ------------------------------------------------------------------------------------------------------ block
@pytest.mark.parametrize("C,H,T", [(128, 2, 64), (256, 2, 77), (1024, 8, 128)])
@pytest.mark.parametrize("mode", ["complex", "cos_only"])
@pytest.mark.parametrize("grouped", ["0", "1"])
def test_block_fwd_bwd_vs_oracle(monkeypatch, C, H, T, mode, grouped):
    monkeypatch.setenv("OBTE_GROUPED_WGRAD", grouped)   # per-matrix split-K launches / one grouped weight-gradient launch
    B = 2
    hs = C // H
    cfg = R.RefConfig(block_size=T, vocab_size=256, n_layer=1, n_head=H, n_embd=C)
    w = {k: v.to(BF) for k, v in R.hash_weights(cfg).items()}
    pre = "transformer.h.0."
    names = ["ln_1.weight", "attn.c_attn.weight", "attn.c_proj.weight", "ln_2.weight", "mlp.c_fc.weight", "mlp.c_proj.weight"]
    x, dy = rnd(B, T, C, seed=1), rnd(B, T, C, seed=2, scale=0.1)
    tokens = np.random.default_rng(3).integers(20, 100, size=(B, T))
    tokens[0, T // 2] = R.EOS_TOKEN
    dense, ranges = _blocks_to_masks(tokens, T)
    tab = R.rope_table(hs, T)
    if mode == "cos_only":
        tab = R.cast_rope_table(tab, BF)
    # oracle in fp32 arithmetic on the bf16-valued parameters
    wf = {k: v.float().requires_grad_(True) for k, v in w.items()}
    xf = x.float().requires_grad_(True)
    ref = R.block_forward(xf, wf, pre, cfg, tab, dense.unsqueeze(1))
    ref.backward(dy.float())
    from omnibiote_amd.model import rope_tables
    o = ops()
    params = tuple(w[pre + n].to(DEV) for n in names)
    rope = rope_tables(tab.to(DEV))
    spec = o.MaskSpec(ranges=ranges.to(DEV))
    y, act = o.block_fwd(x.to(DEV), params, rope, H, spec)
    close(y, ref, atol=3e-2, rtol=2.0 ** -6, what="block fwd")
    dx, grads = o.block_bwd(x.to(DEV), dy.to(DEV), act, params, rope, H, spec)
    close(dx, xf.grad, atol=2e-2, rtol=2.0 ** -5, what="block dx")
    for n, g in zip(names, grads):
        gr = wf[pre + n].grad
        tol = 0.03 * gr.abs().max().item() + 1e-3
        close(g, gr, atol=tol, rtol=2.0 ** -5, what="block d" + n)


@pytest.mark.parametrize("C,H,T,n_rows", [(128, 2, 64, 1), (256, 2, 77, 23), (1024, 8, 128, 40), (1024, 8, 1024, 300)])
@pytest.mark.parametrize("grouped", ["0", "1"])
@pytest.mark.parametrize("accumulate", [False, True])
def test_block_rows_form_vs_oracle_and_vs_the_full_block(monkeypatch, C, H, T, n_rows, grouped, accumulate):
    """obte_block_desc::out_rows (the last block of a masked-LM step): the block's output at the listed positions only, its
    MLP half computed on those positions alone.  Against the oracle — the full block, output rows picked, gradient fed at those
    rows and zero elsewhere (what a loss that multiplies the other positions by zero produces) — and against the HIP full
    block run the same way: the same forward rows (to one bf16 rounding: the few-tile projections of the rows form are split-K),
    gradients within the bar; with in-place accumulation as the harness uses it."""
    monkeypatch.setenv("OBTE_GROUPED_WGRAD", grouped)
    B = 2
    hs = C // H
    cfg = R.RefConfig(block_size=T, vocab_size=256, n_layer=1, n_head=H, n_embd=C)
    w = {k: v.to(BF) for k, v in R.hash_weights(cfg).items()}
    pre = "transformer.h.0."
    names = ["ln_1.weight", "attn.c_attn.weight", "attn.c_proj.weight", "ln_2.weight", "mlp.c_fc.weight", "mlp.c_proj.weight"]
    g = torch.Generator().manual_seed(5)
    rows = torch.sort(torch.randperm(B * T, generator=g)[:n_rows]).values
    x, dy_r = rnd(B, T, C, seed=1), rnd(n_rows, C, seed=2, scale=0.1)
    tokens = np.random.default_rng(3).integers(20, 100, size=(B, T))
    tokens[0, T // 2] = R.EOS_TOKEN
    dense, ranges = _blocks_to_masks(tokens, T)
    tab = R.cast_rope_table(R.rope_table(hs, T), BF)
    wf = {k: v.float().requires_grad_(True) for k, v in w.items()}
    xf = x.float().requires_grad_(True)
    ref = R.block_forward(xf, wf, pre, cfg, tab, dense.unsqueeze(1)).reshape(-1, C)[rows]
    ref.backward(dy_r.float())
    from omnibiote_amd.model import rope_tables
    o = ops()
    params = tuple(w[pre + n].to(DEV) for n in names)
    rope = rope_tables(tab.to(DEV))
    spec = o.MaskSpec(ranges=ranges.to(DEV))
    rows_d = rows.to(DEV)
    y_r, act = o.block_fwd(x.to(DEV), params, rope, H, spec, out_rows=rows_d)
    assert tuple(y_r.shape) == (n_rows, C)
    y_full, act_full = o.block_fwd(x.to(DEV), params, rope, H, spec)
    # the listed rows of the full block: the same arithmetic per position, up to the summation order of a split-K projection and —
    # the attention with its queries at the listed rows only — the moments at which a wave of other queries moves its running maximum
    # (the attention output differs by single bf16 roundings, which the MLP half carries on)
    close(y_r, y_full.reshape(-1, C)[rows_d], atol=8e-3, rtol=2.0 ** -7, what="block rows vs full block")
    close(y_r, ref, atol=3e-2, rtol=2.0 ** -6, what="block rows fwd")
    acc = None
    old = None
    if accumulate:   # gradients added into existing buffers by the epilogues, LayerNorm weights included
        old = [rnd(*p.shape, seed=30 + i, scale=0.05).to(DEV) for i, p in enumerate(params)]
        acc = [t.clone() for t in old]
    dx, grads = o.block_bwd(x.to(DEV), dy_r.to(DEV), act, params, rope, H, spec, accumulate_into=acc, out_rows=rows_d)
    close(dx, xf.grad, atol=2e-2, rtol=2.0 ** -5, what="block rows dx")
    dy_full = torch.zeros(B * T, C, dtype=BF, device=DEV)
    dy_full[rows_d] = dy_r.to(DEV)
    dx_full, grads_full = o.block_bwd(x.to(DEV), dy_full.reshape(B, T, C), act_full, params, rope, H, spec)
    assert (dx.float() - dx_full.float()).norm().item() <= 0.01 * dx_full.float().norm().item() + 1e-6
    for i, (n, gr_) in enumerate(zip(names, grads)):
        want = wf[pre + n].grad
        got = (acc[i].float() - old[i].float()) if accumulate else gr_.float()
        tol = 0.03 * want.abs().max().item() + 1e-3 + (2.0 ** -7 * old[i].float().abs().max().item() if accumulate else 0.0)
        close(got, want, atol=tol, rtol=2.0 ** -5, what="block rows d" + n)
        if not accumulate:
            assert (gr_.float() - grads_full[i].float()).norm().item() <= 0.02 * grads_full[i].float().norm().item() + 1e-5, n


@pytest.mark.parametrize("C,H,T,n_rows", [(256, 2, 77, 23), (1024, 8, 512, 700)])
def test_block_rows_form_without_a_mask_and_with_whole_sequences_left_out(monkeypatch, C, H, T, n_rows):
    """The rows form's attention runs its QUERIES at the listed positions only (csrc/block.cpp rows_attn; keys and values of every
    position): here without any mask (the kernels' no-mask mode over a gathered query set), with a batch element that has NO listed
    row at all and one whose rows fill more than one 256-query block — against the oracle's full block picked at those rows, and
    against the same call with the full attention (OBTE_ROWS_ATTN=0 is read once per process, so the comparison is to the oracle and
    to the HIP full block)."""
    B = 3
    hs = C // H
    cfg = R.RefConfig(block_size=T, vocab_size=256, n_layer=1, n_head=H, n_embd=C)
    w = {k: v.to(BF) for k, v in R.hash_weights(cfg).items()}
    pre = "transformer.h.0."
    names = ["ln_1.weight", "attn.c_attn.weight", "attn.c_proj.weight", "ln_2.weight", "mlp.c_fc.weight", "mlp.c_proj.weight"]
    g = torch.Generator().manual_seed(9)
    # batch element 1 contributes nothing; element 0 few rows, element 2 the rest (more than 256 at the larger size)
    n0 = min(7, n_rows - 1)
    rows0 = torch.randperm(T, generator=g)[:n0]
    rows2 = 2 * T + torch.randperm(T, generator=g)[:min(T, n_rows - n0)]
    rows = torch.sort(torch.cat([rows0, rows2])).values
    n_rows = rows.numel()
    x, dy_r = rnd(B, T, C, seed=1), rnd(n_rows, C, seed=2, scale=0.1)
    tab = R.cast_rope_table(R.rope_table(hs, T), BF)
    wf = {k: v.float().requires_grad_(True) for k, v in w.items()}
    xf = x.float().requires_grad_(True)
    ref = R.block_forward(xf, wf, pre, cfg, tab, None).reshape(-1, C)[rows]
    ref.backward(dy_r.float())
    from omnibiote_amd.model import rope_tables
    o = ops()
    params = tuple(w[pre + n].to(DEV) for n in names)
    rope = rope_tables(tab.to(DEV))
    rows_d = rows.to(DEV)
    spec = o.MaskSpec()
    y_r, act = o.block_fwd(x.to(DEV), params, rope, H, spec, out_rows=rows_d)
    close(y_r, ref, atol=3e-2, rtol=2.0 ** -6, what="block rows fwd, no mask")
    y_full, act_full = o.block_fwd(x.to(DEV), params, rope, H, spec)
    close(y_r, y_full.reshape(-1, C)[rows_d], atol=8e-3, rtol=2.0 ** -7, what="block rows vs full block, no mask")
    dx, grads = o.block_bwd(x.to(DEV), dy_r.to(DEV), act, params, rope, H, spec, out_rows=rows_d)
    close(dx, xf.grad, atol=2e-2, rtol=2.0 ** -5, what="block rows dx, no mask")
    assert not dx[1].isnan().any()
    for n, gr_ in zip(names, grads):
        want = wf[pre + n].grad
        close(gr_.float(), want, atol=0.03 * want.abs().max().item() + 1e-3, rtol=2.0 ** -5, what="block rows d" + n + ", no mask")


# ------------------------------------------------------------------------------ GEMM: split-K and big-tile paths
@pytest.mark.parametrize("M,N,K", [(1024, 1024, 2048), (384, 256, 4096), (256, 128, 8192), (3072, 1024, 1100)])
def test_gemm_wgrad_split_k(M, N, K):
    """Weight-gradient shapes with few output tiles and a long token dimension take the split-K path (fp32 slabs
    + fixed-order reduce): same answer, bitwise reproducible."""
    assert L().lib().obte_gemm_workspace_bytes(M, N, K) > 0
    dy, x = rnd(K, M, seed=21, scale=0.5), rnd(K, N, seed=22, scale=0.5)
    ref = dy.float().t() @ x.float()
    got = ops().linear_wgrad(dy.to(DEV), x.to(DEV))
    close(got, ref, atol=0.01 * math.sqrt(K), what="gemm TN split-K")
    assert torch.equal(got, ops().linear_wgrad(dy.to(DEV), x.to(DEV)))


def test_gemm_long_k_and_many_tiles():
    """Ring wrap-around over many K-tiles (3-stage ring, loads two tiles ahead) and a multi-tile grid."""
    M, N, K = 520, 392, 4096
    x, w = rnd(M, K, seed=31, scale=0.3), rnd(N, K, seed=32, scale=0.3)
    close(ops().linear_fwd(x.to(DEV), w.to(DEV)), x.float() @ w.float().t(), atol=0.02 * math.sqrt(K) * 0.09, what="long K NT")
    dy, w2 = rnd(M, K, seed=33, scale=0.3), rnd(K, N, seed=34, scale=0.3)
    close(ops().linear_dgrad(dy.to(DEV), w2.to(DEV)), dy.float() @ w2.float(), atol=0.02 * math.sqrt(K) * 0.09, what="long K NN")


def test_gemm_linearity_full_size():
    """BASELINE config-2 sized forward GEMM (8192 x 3072 x 1024): f(x1 + x2) == f(x1) + f(x2) up to bf16 rounding of
    the outputs, and exact agreement with a row subsample of the fp32 reference."""
    M, N, K = 8192, 3072, 1024
    g = torch.Generator(device=DEV).manual_seed(0)
    x1 = torch.randn(M, K, device=DEV, generator=g).to(BF)
    x2 = (torch.randn(M, K, device=DEV, generator=g) * 0.5).to(BF)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.03).to(BF)
    xs = (x1.float() + x2.float()).to(BF)
    y = ops().linear_fwd(xs, w).float()
    y12 = ops().linear_fwd(x1, w).float() + ops().linear_fwd(x2, w).float()
    assert (y - y12).abs().max().item() <= 0.06
    rows = torch.arange(0, M, 257, device=DEV)
    ref = xs[rows].float() @ w.float().t()
    close(y[rows], ref, atol=0.02, what="full-size rows")


def _force_gemm(monkeypatch, bn):
    """'128' / '256': tile width of the K-tile ring; '256h': the 256-wide half-tile ring (structure 3); '128q': the 128-wide
    half-tile ring with four waves and two workgroups per CU (structure 4)."""
    monkeypatch.setenv("OBTE_GEMM_BN", bn[:3])
    if bn.endswith("h"):
        monkeypatch.setenv("OBTE_GEMM", "v3")
    if bn.endswith("q"):
        monkeypatch.setenv("OBTE_GEMM", "v4")


@pytest.mark.parametrize("bn", ["128", "256", "256h", "128q"])
def test_gemm_both_tile_widths(monkeypatch, bn):
    """Every layout, epilogue and the split-K path on both tile widths and both ring structures (the library picks
    per shape; here forced)."""
    _force_gemm(monkeypatch, bn)
    o = ops()
    for (M, N, K) in [(520, 392, 256), (256, 256, 128), (300, 1024, 1024)]:
        x, w = rnd(M, K, seed=41), rnd(N, K, seed=42, scale=0.2)
        close(o.linear_fwd(x.to(DEV), w.to(DEV)), x.float() @ w.float().t(), atol=0.02 * math.sqrt(K) * 0.2, what=f"NT bn{bn}")
        dy, w2 = rnd(M, K, seed=43), rnd(K, N, seed=44, scale=0.2)
        close(o.linear_dgrad(dy.to(DEV), w2.to(DEV)), dy.float() @ w2.float(), atol=0.02 * math.sqrt(K) * 0.2, what=f"NN bn{bn}")
    for (M, N, K) in [(384, 520, 300), (1024, 1024, 2048), (256, 264, 77)]:
        dy, xx = rnd(K, M, seed=45, scale=0.5), rnd(K, N, seed=46, scale=0.5)
        close(o.linear_wgrad(dy.to(DEV), xx.to(DEV)), dy.float().t() @ xx.float(), atol=0.01 * math.sqrt(K), what=f"TN bn{bn}")
    # identity checks: exact
    eye = torch.eye(256).to(BF)
    b = (torch.arange(512 * 256).reshape(512, 256) % 251 - 125).float().to(BF)
    assert torch.equal(o.linear_fwd(eye.to(DEV), b.to(DEV)).cpu().float(), b.float().t())
    assert torch.equal(o.linear_dgrad(eye.to(DEV), b.t().contiguous().to(DEV)).cpu().float(), b.float().t())
    assert torch.equal(o.linear_wgrad(eye.to(DEV), b.t().contiguous().to(DEV)).cpu().float(), b.float().t())
    # epilogues
    M, N, K = 300, 512, 128
    x, w, r = rnd(M, K, seed=7), rnd(N, K, seed=8, scale=0.2), rnd(M, N, seed=9)
    acc = x.float() @ w.float().t()
    close(o.linear_fwd(x.to(DEV), w.to(DEV), epilogue=L().EPI_ADD, aux=r.to(DEV)), r.float() + acc.to(BF).float(), atol=0.03, what="add")
    d, d2 = o.linear_fwd(x.to(DEV), w.to(DEV), epilogue=L().EPI_GELU)
    close(d2, R.gelu_erf(acc.to(BF).float()), atol=0.03, what="gelu act")
    close(o.linear_fwd(x.to(DEV), w.to(DEV), alpha=1 / 42.0), acc / 42.0, atol=2e-3, what="alpha")


def test_gemm_row_dot_epilogue_forms_the_softmax_backward_delta():
    """The block backward's d(attention output) = dx1 W_proj with the row-dot epilogue of structure 7 (csrc/common.h
    obte_gemm_rowdot_bf16): the product itself bit for bit the plain launch's, and rowdot[(b H + h) T + t] = sum over head h's 128
    columns of D[b T + t, c] * other[b T + t, c] — the softmax backward's delta, which the attention backward's prep launch otherwise
    forms by reading both tensors again; shapes structure 7 does not take answer 1 and launch nothing."""
    import ctypes as C
    o, Lm = ops(), L()
    lib = Lm.lib()
    lib.obte_gemm_rowdot_bf16.restype = C.c_int
    lib.obte_gemm_rowdot_bf16.argtypes = [C.POINTER(Lm.GemmArgs), C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    for (B, T, Cc) in [(16, 1024, 1024), (64, 256, 2048)]:
        M, H = B * T, Cc // 128
        dx1, w, y = rnd(M, Cc, seed=3, scale=0.5), rnd(Cc, Cc, seed=4, scale=0.05), rnd(M, Cc, seed=5)
        dx1d, wd, yd = dx1.to(DEV), w.to(DEV), y.to(DEV)
        plain = o.gemm(dx1d, wd, M, Cc, Cc, True, False, Lm.EPI_NONE, None)
        out = torch.empty(M * Cc, dtype=BF, device=DEV)
        dot = torch.full((B * H * T,), float("nan"), dtype=torch.float32, device=DEV)
        g = Lm.GemmArgs()
        g.a, g.b, g.d = dx1d.data_ptr(), wd.data_ptr(), out.data_ptr()
        g.M, g.N, g.K, g.lda, g.ldb, g.ldd = M, Cc, Cc, Cc, Cc, Cc
        g.a_kmajor, g.b_kmajor, g.epilogue, g.alpha = 1, 0, Lm.EPI_NONE, 1.0
        rc = lib.obte_gemm_rowdot_bf16(C.byref(g), yd.data_ptr(), dot.data_ptr(), T, 128, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, (rc, Lm.last_error() if hasattr(Lm, "last_error") else "")
        assert torch.equal(out.reshape(M, Cc), plain.reshape(M, Cc)), "the product itself"
        want = (out.reshape(B, T, H, 128).float() * yd.reshape(B, T, H, 128).float()).sum(-1).permute(0, 2, 1).reshape(-1)
        close(dot, want.cpu(), atol=2e-3 * math.sqrt(128), rtol=1e-4, what=f"row dot B={B} T={T} C={Cc}")
    # fewer than 256 tiles: not structure 7's shape
    M, Cc = 2048, 1024
    g = Lm.GemmArgs()
    a_, b_, d_ = rnd(M, Cc, seed=1).to(DEV), rnd(Cc, Cc, seed=2).to(DEV), torch.empty(M * Cc, dtype=BF, device=DEV)
    g.a, g.b, g.d = a_.data_ptr(), b_.data_ptr(), d_.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldd = M, Cc, Cc, Cc, Cc, Cc
    g.a_kmajor, g.b_kmajor, g.epilogue, g.alpha = 1, 0, Lm.EPI_NONE, 1.0
    dot = torch.zeros(2 * 8 * 1024, dtype=torch.float32, device=DEV)
    assert lib.obte_gemm_rowdot_bf16(C.byref(g), a_.data_ptr(), dot.data_ptr(), 1024, 128, torch.cuda.current_stream().cuda_stream) == 1


def test_gemm_persistent_continuous_ring_structure():
    """Structure 7 (csrc/gemm_bf16_v7.hip: the 256 x 256 half-tile ring as a persistent kernel — the LDS-DMA stream runs on across
    tiles, each wave's epilogue goes through 4 KiB of staging of its own, stores drain under the next tile's loop): every layout and
    epilogue it is built for — x W^T plain / GELU pair / residual add / residual add + dropout / RoPE on the q, k thirds; dy W plain /
    GELU' — at 1, 2, 2.5 and 8 tiles per workgroup and K from 256 to 4096, bit for bit against structure 3 (the same main loop
    and per-tile arithmetic, one tile per workgroup), against the fp32 reference, and with an exact identity product; shapes it
    does not take (ragged, fewer tiles than CUs, short K) fall back; the plan table refuses layouts it has no form for."""
    o, lib, Lm = ops(), L().lib(), L()
    def plan(ak, bk, epi, M, N, K, variant):
        Lm.check(lib.obte_gemm_plan_set(ak, bk, epi, M, N, K, variant, 256, 1), "obte_gemm_plan_set")
    try:
        # x W^T, plain: tile counts around the persistent walk's edges (256 = one each, 320 = some two, 512, 2048)
        for (M, N, K) in [(8192, 2048, 256), (8192, 2560, 512), (16384, 2048, 1024), (32768, 4096, 1024), (8192, 2048, 4096)]:
            x, w = rnd(M, K, seed=61).to(DEV), rnd(N, K, seed=62, scale=0.2).to(DEV)
            outs = {}
            for variant in (7, 3):
                plan(1, 1, Lm.EPI_NONE, M, N, K, variant)
                outs[variant] = o.linear_fwd(x, w)
            assert torch.equal(outs[7], outs[3]), f"structure 7 vs 3, NT {M}x{N}x{K}"
            rows = torch.arange(0, M, 257)
            close(outs[7][rows.to(DEV)], x[rows.to(DEV)].float().cpu() @ w.float().cpu().t(), atol=0.02 * math.sqrt(K) * 0.2, what=f"NT structure 7 {M}x{N}x{K}")
        # epilogues of the forward
        M, N, K = 16384, 3072, 1024
        x, w, r = rnd(M, K, seed=7).to(DEV), rnd(N, K, seed=8, scale=0.2).to(DEV), rnd(M, N, seed=9).to(DEV)
        T, hs = 1024, 128
        tab = torch.randn(T, hs // 2, generator=torch.Generator().manual_seed(1))
        rope = (torch.cos(tab).to(DEV), torch.sin(tab).to(DEV))
        for epi in (Lm.EPI_GELU, Lm.EPI_ADD, Lm.EPI_ADD_DROPOUT, Lm.EPI_ROPE_QK):
            outs = {}
            for variant in (7, 3):
                plan(1, 1, epi, M, N, K, variant)
                kw = {}
                if epi in (Lm.EPI_ADD, Lm.EPI_ADD_DROPOUT):
                    kw["aux"] = r
                if epi == Lm.EPI_ADD_DROPOUT:
                    kw["dropout"] = (0.1, SEED, 2)
                if epi == Lm.EPI_ROPE_QK:
                    outs[variant] = o.gemm(x.reshape(-1), w.reshape(-1), M, N, K, True, True, epi, rope=(rope[0], rope[1], T, hs))
                else:
                    outs[variant] = o.linear_fwd(x, w, epilogue=epi, **kw)
            if epi == Lm.EPI_GELU:
                assert torch.equal(outs[7][0], outs[3][0]) and torch.equal(outs[7][1], outs[3][1]), "GELU pair"
                acc = (x[:256].float().cpu() @ w.float().cpu().t())
                close(outs[7][1][:256], R.gelu_erf(acc.to(BF).float()), atol=0.03, what="gelu act, structure 7")
            else:
                assert torch.equal(outs[7].reshape(-1), outs[3].reshape(-1)), f"epilogue {epi}"
        # dy W (B k-strided): plain and GELU'
        M, N, K = 16384, 4096, 1024
        dy, w, h = rnd(M, K, seed=11).to(DEV), rnd(K, N, seed=12, scale=0.2).to(DEV), rnd(M, N, seed=13).to(DEV)
        for epi in (Lm.EPI_NONE, Lm.EPI_GELU_BWD):
            outs = {}
            for variant in (7, 3):
                plan(1, 0, epi, M, N, K, variant)
                outs[variant] = o.gemm(dy.reshape(-1), w.reshape(-1), M, N, K, True, False, epi, h.reshape(-1) if epi == Lm.EPI_GELU_BWD else None)
            assert torch.equal(outs[7], outs[3]), f"NN epilogue {epi}"
        close(outs[7].reshape(M, N)[:256], (dy[:256].float().cpu() @ w.float().cpu()).to(BF).float() * h[:256].float().cpu(), atol=0.05, what="dgrad + GELU', structure 7")
        # exact identity product with an asymmetric B, two tiles per workgroup
        M, N, K = 16384, 2048, 1024
        plan(1, 1, Lm.EPI_NONE, M, N, K, 7)
        eye = torch.zeros(M, K); eye[torch.arange(K), torch.arange(K)] = 1.0; eye[8192 + torch.arange(K), torch.arange(K)] = 1.0
        b = (torch.arange(N * K).reshape(N, K) % 251 - 125).float().to(BF)
        got = o.linear_fwd(eye.to(BF).to(DEV), b.to(DEV)).cpu().float()
        assert torch.equal(got[:K], b.float().t()) and torch.equal(got[8192:8192 + K], b.float().t()) and not got[K:8192].any() and not got[8192 + K:].any()
        # shapes the structure does not take fall back (ragged; fewer tiles than CUs; K of two K-tiles) instead of failing
        for (M, N, K) in [(300, 520, 128), (2048, 2048, 1024), (16384, 4096, 128)]:
            plan(1, 1, 0, M, N, K, 7)
            xs, ws = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2)
            close(o.linear_fwd(xs.to(DEV), ws.to(DEV))[:300], xs[:300].float() @ ws.float().t(), atol=0.02 * math.sqrt(K) * 0.2, what="fallback")
        with pytest.raises(RuntimeError, match="continuous-ring structure"):
            Lm.check(lib.obte_gemm_plan_set(0, 0, 0, 8192, 4096, 8192, 7, 256, 1), "obte_gemm_plan_set")
    finally:
        Lm.check(lib.obte_gemm_plan_clear(), "obte_gemm_plan_clear")


def test_gemm_192_wide_tile():
    """The 256 x 192 tile of the K-tile ring (N = 3072 is sixteen of them: two full rounds of 256 workgroups instead of
    one and a half of the 256-wide tile), installed through the plan table for x @ W^T: plain, GELU pair, residual add,
    RoPE epilogue, ragged M/N/K edges, and an exact identity product."""
    o, lib = ops(), L().lib()
    def plan(epi, M, N, K):
        L().check(lib.obte_gemm_plan_set(1, 1, epi, M, N, K, 2, 192, 1), "obte_gemm_plan_set")
    try:
        for (M, N, K) in [(520, 392, 256), (256, 192, 128), (300, 1000, 1024), (1024, 3072, 1024), (77, 576, 192)]:
            plan(L().EPI_NONE, M, N, K)
            x, w = rnd(M, K, seed=41), rnd(N, K, seed=42, scale=0.2)
            close(o.linear_fwd(x.to(DEV), w.to(DEV)), x.float() @ w.float().t(), atol=0.02 * math.sqrt(K) * 0.2, what=f"NT 192 {M}x{N}x{K}")
        plan(L().EPI_NONE, 256, 512, 256)
        eye = torch.eye(256).to(BF)
        b = (torch.arange(512 * 256).reshape(512, 256) % 251 - 125).float().to(BF)
        assert torch.equal(o.linear_fwd(eye.to(DEV), b.to(DEV)).cpu().float(), b.float().t())
        M, N, K = 300, 520, 128
        x, w, r = rnd(M, K, seed=7), rnd(N, K, seed=8, scale=0.2), rnd(M, N, seed=9)
        acc = x.float() @ w.float().t()
        plan(L().EPI_ADD, M, N, K)
        close(o.linear_fwd(x.to(DEV), w.to(DEV), epilogue=L().EPI_ADD, aux=r.to(DEV)), r.float() + acc.to(BF).float(), atol=0.03, what="add 192")
        plan(L().EPI_GELU, M, N, K)
        d, d2 = o.linear_fwd(x.to(DEV), w.to(DEV), epilogue=L().EPI_GELU)
        pre = acc.to(BF).float().requires_grad_(True)
        R.gelu_erf(pre).sum().backward()
        close(d, pre.grad, atol=0.03, what="gelu derivative 192")   # (d carries GELU'(pre-activation), what the backward multiplies by)
        close(d2, R.gelu_erf(acc.to(BF).float()), atol=0.03, what="gelu act 192")
        # the plan may not be installed for layouts or epilogues the tile does not implement
        with pytest.raises(RuntimeError, match="192-wide tile"):
            L().check(lib.obte_gemm_plan_set(0, 0, 0, M, N, K, 2, 192, 1), "obte_gemm_plan_set")
        with pytest.raises(RuntimeError, match="192-wide tile"):
            L().check(lib.obte_gemm_plan_set(1, 1, L().EPI_GELU_BWD, M, N, K, 2, 192, 1), "obte_gemm_plan_set")
        # RoPE epilogue: bit-equal to the projection under the same tile followed by the stand-alone RoPE kernel
        from omnibiote_amd.model import rope_tables
        for hs, B, T, H in [(128, 2, 77, 3), (64, 2, 130, 6)]:
            C = H * hs
            x, w = rnd(B * T, C, seed=61), rnd(3 * C, C, seed=62, scale=0.1)
            cos, sin = rope_tables(R.cast_rope_table(R.rope_table(hs, 256), BF).to(DEV))
            plan(L().EPI_ROPE_QK, B * T, 3 * C, C)
            plan(L().EPI_NONE, B * T, 3 * C, C)
            fused = o.gemm(x.to(DEV), w.to(DEV), B * T, 3 * C, C, True, True, L().EPI_ROPE_QK, rope=(cos, sin, T, hs))
            plain = o.linear_fwd(x.to(DEV), w.to(DEV))
            o.rope_qk_(plain, cos, sin, B, T, H, hs)
            assert torch.equal(fused, plain)
    finally:
        L().check(lib.obte_gemm_plan_clear(), "obte_gemm_plan_clear")


@pytest.mark.parametrize("accumulate", [False, True])
def test_gemm_grouped_matches_single_launches(accumulate):
    """One grouped launch of four weight-gradient shaped products (ragged M/N, K not a multiple of 64) plus a shorter
    input-gradient shaped product in another layout (the mix a block's backward issues), against fp32 references and
    against the single-problem entry point."""
    o = ops()
    K = 1000
    shapes = [(520, 264), (256, 1024), (776, 256), (256, 256)]
    probs, refs, singles = [], [], []
    for i, (M, N) in enumerate(shapes):
        a, b = rnd(K, M, seed=60 + i, scale=0.5), rnd(K, N, seed=70 + i, scale=0.5)
        base = rnd(M, N, seed=80 + i)
        probs.append(dict(a=a.to(DEV), b=b.to(DEV), M=M, N=N, K=K, out=base.clone().to(DEV), accumulate=accumulate))
        acc = a.float().t() @ b.float()
        refs.append(base.float() + acc.to(BF).float() if accumulate else acc)
        singles.append(o.linear_wgrad(a.to(DEV), b.to(DEV), accumulate_into=base.clone().to(DEV) if accumulate else None))
    # dX = dY W: A k-contiguous [M, K2], B k-strided [K2, N]; never accumulates
    M5, N5, K5 = 1000, 392, 320
    dy, w = rnd(M5, K5, seed=90, scale=0.5), rnd(K5, N5, seed=91, scale=0.2)
    probs.append(dict(a=dy.to(DEV), b=w.to(DEV), M=M5, N=N5, K=K5, out=torch.full((M5, N5), 7.0, dtype=BF, device=DEV), a_kmajor=True))
    refs.append(dy.float() @ w.float())
    singles.append(o.linear_dgrad(dy.to(DEV), w.to(DEV)))
    outs = o.gemm_grouped(probs)
    for got, ref, single in zip(outs, refs, singles):
        close(got, ref, atol=0.01 * math.sqrt(K) + 0.02, what="grouped")
        close(got, single.float().cpu(), atol=0.07, what="grouped vs single launch")   # fp32 summation order differs (split-K)
    with pytest.raises(RuntimeError, match="epilogue must be NONE or ADD"):
        q = probs[0]
        bad = (L().GemmArgs * 1)()
        bad[0] = L().GemmArgs(q["a"].data_ptr(), q["b"].data_ptr(), q["out"].data_ptr(), None, None, q["M"], q["N"], K, q["M"], q["N"], q["N"],
                              0, 0, L().EPI_GELU, 1.0, 0.0, 0, 0)
        L().check(L().lib().obte_gemm_grouped_bf16(bad, 1, None), "obte_gemm_grouped_bf16")


@pytest.mark.parametrize("accumulate", [False, True])
def test_linear_bwd_grouped_pair_matches_two_launches(monkeypatch, accumulate):
    """The readout's backward as one grouped launch (input gradient with K = vocabulary beside the weight gradient with
    K = tokens, alpha = 1/width_mult on both) against the two single launches and fp32 references."""
    o = ops()
    M, N, K, alpha = 384, 2048, 256, 0.25
    dy, x, w = rnd(M, N, seed=11, scale=0.3), rnd(M, K, seed=12), rnd(N, K, seed=13, scale=0.2)
    base = rnd(N, K, seed=14)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("OBTE_GROUPED_LM", mode)
        assert o.linear_bwd_pair_is_grouped(M, N, K) == (mode == "1")
        slot = base.clone().to(DEV) if accumulate else None
        dx, dw = o.linear_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), alpha=alpha, accumulate_into=slot)
        res[mode] = (dx.cpu(), (slot if accumulate else dw).cpu())
    dx_ref = alpha * (dy.float() @ w.float())
    dw_new = alpha * (dy.float().t() @ x.float())
    dw_ref = base.float() + dw_new.to(BF).float() if accumulate else dw_new
    for mode in ("0", "1"):
        close(res[mode][0], dx_ref, atol=0.05, what=f"dx grouped={mode}")
        close(res[mode][1], dw_ref, atol=0.05, what=f"dW grouped={mode}")
    close(res["1"][0], res["0"][0].float(), atol=0.03, what="dx grouped vs single")
    close(res["1"][1], res["0"][1].float(), atol=0.03, what="dW grouped vs single")
    monkeypatch.delenv("OBTE_GROUPED_LM")
    assert not o.linear_bwd_pair_is_grouped(8192, 65536, 1024) and not o.linear_bwd_pair_is_grouped(4915, 65536, 1024)   # on request only


def test_gemm_plan_cache_round_trip(tmp_path):
    """tune -> save_plans -> plan_clear -> load_plans: the same plans are installed again without a launch, and a GEMM
    that uses them still matches the fp32 reference."""
    import json
    from omnibiote_amd import tune
    M, N, K = 512, 1024, 2048
    v, bn, splits, ms = tune.tune_gemm(M, N, K, False, False, L().EPI_NONE)
    path = str(tmp_path / "plans.json")
    tune.save_plans(path)
    rows = json.load(open(path))
    mine = [r for r in rows if (r["M"], r["N"], r["K"], r["a_kmajor"], r["b_kmajor"]) == (M, N, K, False, False)]
    assert len(mine) == 1 and (mine[0]["variant"], mine[0]["bn"], mine[0]["splits"]) == (v, bn, splits)
    L().check(L().lib().obte_gemm_plan_clear(), "obte_gemm_plan_clear")
    assert tune.load_plans(path) == len(rows)
    dy, x = rnd(K, M, seed=3, scale=0.5), rnd(K, N, seed=4, scale=0.5)
    close(ops().linear_wgrad(dy.to(DEV), x.to(DEV)), dy.float().t() @ x.float(), atol=0.01 * math.sqrt(K), what="wgrad under a re-loaded plan")
    with pytest.raises(RuntimeError, match="bad plan"):
        L().check(L().lib().obte_gemm_plan_set(0, 0, 0, M, N, K, 9, 128, 1), "obte_gemm_plan_set")


def test_multi_tensor_adamw_matches_single_tensor_kernel():
    """The multi-tensor launch (<= 32 tensors per kernel, per-tensor lr / weight decay / step) must reproduce the
    single-tensor kernel bit for bit, and the multi-tensor sum of squares must equal the sum of the single ones."""
    import ctypes as C
    from omnibiote_amd import _lib as LL
    sizes = [8, 1024, 16384, 16392, 50000 * 8, 1024, 24]
    lrs = [1e-2, 2e-3, 5e-4, 1e-2, 3e-3, 1e-2, 7e-3]
    wds = [0.0, 1e-2, 0.4, 1e-2, 1e-2, 0.1, 1e-2]
    steps = [1, 2, 3, 4, 5, 6, 7]
    b1, b2, eps, clip = 0.9, 0.999, 1e-8, 0.7
    P = [rnd(n, seed=10 + i).to(DEV) for i, n in enumerate(sizes)]
    G = [rnd(n, seed=30 + i, scale=0.1).to(DEV) for i, n in enumerate(sizes)]
    Mo = [rnd(n, seed=50 + i, scale=0.01).to(DEV) for i, n in enumerate(sizes)]
    Vo = [(rnd(n, seed=70 + i, scale=0.01).float() ** 2).to(BF).to(DEV) for i, n in enumerate(sizes)]
    cc = torch.tensor([clip], device=DEV)
    ref = [(p.clone(), m.clone(), v.clone()) for p, m, v in zip(P, Mo, Vo)]
    for (p, m, v), g, lr, wd, st in zip(ref, G, lrs, wds, steps):
        ops().adamw_step_(p, g, m, v, lr, b1, b2, eps, wd, st, cc)
    a = LL.MtArgs()
    a.count = len(sizes)
    for j in range(len(sizes)):
        a.p[j], a.g[j], a.m[j], a.v[j] = P[j].data_ptr(), G[j].data_ptr(), Mo[j].data_ptr(), Vo[j].data_ptr()
        a.n[j], a.lr[j], a.weight_decay[j], a.step[j] = sizes[j], lrs[j], wds[j], steps[j]
    stream = torch.cuda.current_stream().cuda_stream
    tot = torch.zeros(1, device=DEV)
    LL.check(LL.lib().obte_sumsq_multi_bf16(C.byref(a), tot.data_ptr(), stream))
    LL.check(LL.lib().obte_adamw_multi_bf16(C.byref(a), b1, b2, eps, cc.data_ptr(), stream))
    for (p, m, v), pp, mm, vv in zip(ref, P, Mo, Vo):
        assert torch.equal(p, pp) and torch.equal(m, mm) and torch.equal(v, vv)
    want = sum((g.float() ** 2).sum().item() for g in G)
    assert abs(tot.item() - want) <= 1e-4 * want


def test_wgrad_and_embedding_accumulate_in_place():
    o = ops()
    for (M, N, K) in [(1024, 1024, 2048), (384, 520, 300), (4096, 256, 512)]:   # split-K and plain plans
        dy, x, old = rnd(K, M, seed=51, scale=0.5), rnd(K, N, seed=52, scale=0.5), rnd(M, N, seed=53)
        fresh = o.linear_wgrad(dy.to(DEV), x.to(DEV))
        want = (old.to(DEV).float() + fresh.float()).to(BF)            # autograd's accumulation
        acc = old.clone().to(DEV)
        o.linear_wgrad(dy.to(DEV), x.to(DEV), accumulate_into=acc)
        assert torch.equal(acc, want), (M, N, K)
        acc2 = old.clone().to(DEV)
        o.linear_wgrad(dy.to(DEV), x.to(DEV), alpha=0.25, accumulate_into=acc2)
        want2 = (old.to(DEV).float() + o.linear_wgrad(dy.to(DEV), x.to(DEV), alpha=0.25).float()).to(BF)
        assert torch.equal(acc2, want2)
    V, C, rows = 300, 128, 500
    idx = torch.from_numpy(np.random.default_rng(3).integers(0, V, size=rows)); idx[:200] = 2
    dout, old = rnd(rows, C, seed=54), rnd(V, C, seed=55)
    fresh = o.embedding_bwd(idx.to(DEV), dout.to(DEV), V)
    acc = old.clone().to(DEV)
    assert o.embedding_bwd(idx.to(DEV), dout.to(DEV), V, accumulate_into=acc) is None
    assert torch.equal(acc, (old.to(DEV).float() + fresh.float()).to(BF))


# ---------------------------------------------------------------------------------------------------- dropout
SEED = 0x1234_5678_9ABC


def test_dropout_mask_matches_restatement_and_rate():
    n, p = 1 << 20, 0.1
    x = torch.ones(n, dtype=BF, device=DEV)
    y = ops().dropout(x, p, SEED, site=7).float().cpu()
    mask = R.dropout_scale_mask((n,), p, SEED, 7)           # a 1-D tensor is one row of n columns
    assert torch.equal(y, mask.to(BF).float())
    y2 = ops().dropout(x.view(1 << 10, 1 << 10), p, SEED, site=7).float().cpu()      # the same buffer as a 1024 x 1024 matrix
    assert torch.equal(y2, R.dropout_scale_mask((1 << 10, 1 << 10), p, SEED, 7).to(BF).float())
    assert not torch.equal(y2.flatten(), y)
    assert abs((y != 0).float().mean().item() - 0.9) < 2e-3
    z = ops().dropout(x, p, SEED + 1, site=7).float().cpu()
    assert not torch.equal(y, z)
    assert torch.equal(ops().dropout(x, 0.0, SEED, site=7), x)
    with pytest.raises(RuntimeError):
        ops().dropout(x, 1.0, SEED)


def test_embedding_dropout_fwd_bwd():
    V, C, rows, p = 300, 128, 500, 0.3
    idx = torch.from_numpy(np.random.default_rng(3).integers(0, V, size=rows)); idx[:150] = 2
    wte, dout = rnd(V, C, seed=1), rnd(rows, C, seed=2)
    mask = R.dropout_scale_mask((rows, C), p, SEED, 0)
    got = ops().embedding_fwd(idx.to(DEV), wte.to(DEV), p, SEED)
    assert torch.equal(got.cpu(), R.dropout_apply(wte[idx], mask))
    dw = ops().embedding_bwd(idx.to(DEV), dout.to(DEV), V, dropout_p=p, dropout_seed=SEED)
    ref = torch.zeros(V, C).index_add_(0, idx, R.dropout_apply(dout, mask).float())
    close(dw, ref, atol=3e-2, what="embedding bwd with dropout")


@pytest.mark.parametrize("bn", ["128", "256", "256h"])
def test_gemm_residual_dropout_epilogue(monkeypatch, bn):
    _force_gemm(monkeypatch, bn)
    M, N, K, p = 300, 512, 128, 0.2
    x, w, r = rnd(M, K, seed=7), rnd(N, K, seed=8, scale=0.2), rnd(M, N, seed=9)
    acc = (x.float() @ w.float().t()).to(BF)
    mask = R.dropout_scale_mask((M, N), p, SEED, 2)
    ref = r.float() + R.dropout_apply(acc, mask).float()
    got = ops().linear_fwd(x.to(DEV), w.to(DEV), epilogue=L().EPI_ADD_DROPOUT, aux=r.to(DEV), dropout=(p, SEED, 2))
    # the bf16 rounding of acc can differ by an ulp from the fp32 reference matmul: compare with tolerance, but the
    # dropped positions must be exactly the residual
    close(got, ref, atol=0.04, what="residual dropout")
    assert torch.equal(got.cpu()[mask == 0], r[mask == 0])


@pytest.mark.parametrize("hs", [64, 128])
@pytest.mark.parametrize("mode", ["none", "ranges", "dense"])
def test_attention_dropout_fwd_bwd(hs, mode):
    B, H, T, p = 2, 2, 130, 0.2
    C = H * hs
    scale = 8.0 / C
    qkv, q, k, v = _attn_case(B, T, H, hs, seed=hs)
    tokens = np.random.default_rng(T).integers(20, 100, size=(B, T))
    tokens[0, [T // 3]] = R.EOS_TOKEN
    tokens[1, [5, T - 10]] = R.EOS_TOKEN
    dense, ranges = _blocks_to_masks(tokens, T)
    o = ops()
    mask_add, spec = None, None
    if mode == "ranges":
        mask_add, spec = dense.unsqueeze(1), o.MaskSpec(ranges=ranges.to(DEV))
    elif mode == "dense":
        mask_add = dense.unsqueeze(1)
        spec = o.MaskSpec.from_user(dense.to(BF).to(DEV).unsqueeze(1).expand(B, H, T, T), B, T, H, DEV)
    dmask = R.dropout_scale_mask((B, H, T, T), p, SEED, 1)
    qf, kf, vf = q.requires_grad_(True), k.requires_grad_(True), v.requires_grad_(True)
    att = (qf @ kf.transpose(-2, -1)) * scale
    if mask_add is not None:
        att = att + mask_add
    ref = (torch.softmax(att, dim=-1) * dmask) @ vf
    d_o = rnd(B, T, C, seed=99)
    ref.backward(d_o.reshape(B, T, H, hs).transpose(1, 2).float())
    got, lse = o.attn_fwd(qkv.to(DEV), B, T, H, hs, scale, spec, p, SEED)
    close(got, ref.transpose(1, 2).reshape(B, T, C), atol=8e-3, what=f"attn dropout fwd {mode}")
    close(lse, torch.logsumexp(att.detach(), dim=-1), atol=2e-3, rtol=1e-3, what="lse is of the un-dropped scores")
    dqkv = o.attn_bwd(qkv.to(DEV), got, d_o.to(DEV), lse, B, T, H, hs, scale, spec, dropout_p=p, dropout_seed=SEED)
    dref = torch.cat([t.transpose(1, 2).reshape(B, T, C) for t in (qf.grad, kf.grad, vf.grad)], dim=2)
    close(dqkv, dref, atol=2e-2, rtol=2.0 ** -6, what=f"attn dropout bwd {mode}")


@pytest.mark.parametrize("grouped", ["0", "1"])
def test_block_dropout_fwd_bwd_vs_oracle(monkeypatch, grouped):
    """A whole block with all three dropout sites on, against the oracle fed the restated masks."""
    monkeypatch.setenv("OBTE_GROUPED_WGRAD", grouped)
    B, T, C, H, p = 2, 64, 128, 2, 0.15
    hs = C // H
    cfg = R.RefConfig(block_size=T, vocab_size=256, n_layer=1, n_head=H, n_embd=C)
    w = {k: v.to(BF) for k, v in R.hash_weights(cfg).items()}
    pre = "transformer.h.0."
    names = ["ln_1.weight", "attn.c_attn.weight", "attn.c_proj.weight", "ln_2.weight", "mlp.c_fc.weight", "mlp.c_proj.weight"]
    x, dy = rnd(B, T, C, seed=1), rnd(B, T, C, seed=2, scale=0.1)
    tab = R.cast_rope_table(R.rope_table(hs, T), BF)
    m_attn = R.dropout_scale_mask((B, H, T, T), p, SEED, 1)
    m_res = R.dropout_scale_mask((B, T, C), p, SEED, 2)
    m_mlp = R.dropout_scale_mask((B, T, C), p, SEED, 3)
    wf = {k: v.float().requires_grad_(True) for k, v in w.items()}
    xf = x.float().requires_grad_(True)
    import torch.nn.functional as F
    h1 = R.layer_norm(xf, wf[pre + "ln_1.weight"])
    qkv = F.linear(h1, wf[pre + "attn.c_attn.weight"])
    q, k, v = qkv.split(C, dim=2)
    q = R.apply_rope(q.reshape(B, T, H, hs), tab).transpose(1, 2)
    k = R.apply_rope(k.reshape(B, T, H, hs), tab).transpose(1, 2)
    v = v.reshape(B, T, H, hs).transpose(1, 2)
    y = (torch.softmax((q @ k.transpose(-2, -1)) * (8.0 / C), dim=-1) * m_attn) @ v
    y = y.transpose(1, 2).contiguous().view(B, T, C)
    x1 = xf + F.linear(y, wf[pre + "attn.c_proj.weight"]) * m_res
    a = R.gelu_erf(F.linear(R.layer_norm(x1, wf[pre + "ln_2.weight"]), wf[pre + "mlp.c_fc.weight"]))
    ref = x1 + F.linear(a, wf[pre + "mlp.c_proj.weight"]) * m_mlp
    ref.backward(dy.float())
    from omnibiote_amd.model import rope_tables
    o = ops()
    params = tuple(w[pre + n].to(DEV) for n in names)
    rope = rope_tables(tab.to(DEV))
    spec = o.MaskSpec()
    yg, act = o.block_fwd(x.to(DEV), params, rope, H, spec, p, SEED)
    close(yg, ref, atol=4e-2, rtol=2.0 ** -6, what="block dropout fwd")
    dx, grads = o.block_bwd(x.to(DEV), dy.to(DEV), act, params, rope, H, spec, dropout_p=p, dropout_seed=SEED)
    close(dx, xf.grad, atol=2e-2, rtol=2.0 ** -5, what="block dropout dx")
    for n, gg in zip(names, grads):
        gr = wf[pre + n].grad
        close(gg, gr, atol=0.03 * gr.abs().max().item() + 1e-3, rtol=2.0 ** -5, what="block dropout d" + n)


@pytest.mark.parametrize("grouped", ["0", "1"])
def test_block_rows_form_with_dropout_vs_oracle(monkeypatch, grouped):
    """The rows form of the block (obte_block_desc::out_rows) with all three dropout sites on: sites 1 and 2 (attention
    probabilities, attention projection) as in the full block, site 3 (MLP projection) on the [n, C] output — against the oracle
    fed the restated masks, forward and every gradient."""
    monkeypatch.setenv("OBTE_GROUPED_WGRAD", grouped)
    B, T, C, H, p, n_rows = 2, 64, 128, 2, 0.15, 29
    hs = C // H
    cfg = R.RefConfig(block_size=T, vocab_size=256, n_layer=1, n_head=H, n_embd=C)
    w = {k: v.to(BF) for k, v in R.hash_weights(cfg).items()}
    pre = "transformer.h.0."
    names = ["ln_1.weight", "attn.c_attn.weight", "attn.c_proj.weight", "ln_2.weight", "mlp.c_fc.weight", "mlp.c_proj.weight"]
    rows = torch.sort(torch.randperm(B * T, generator=torch.Generator().manual_seed(8))[:n_rows]).values
    x, dy = rnd(B, T, C, seed=1), rnd(n_rows, C, seed=2, scale=0.1)
    tab = R.cast_rope_table(R.rope_table(hs, T), BF)
    m_attn = R.dropout_scale_mask((B, H, T, T), p, SEED, 1)
    m_res = R.dropout_scale_mask((B, T, C), p, SEED, 2)
    m_mlp = R.dropout_scale_mask((n_rows, C), p, SEED, 3)
    wf = {k: v.float().requires_grad_(True) for k, v in w.items()}
    xf = x.float().requires_grad_(True)
    import torch.nn.functional as F
    h1 = R.layer_norm(xf, wf[pre + "ln_1.weight"])
    qkv = F.linear(h1, wf[pre + "attn.c_attn.weight"])
    q, k, v = qkv.split(C, dim=2)
    q = R.apply_rope(q.reshape(B, T, H, hs), tab).transpose(1, 2)
    k = R.apply_rope(k.reshape(B, T, H, hs), tab).transpose(1, 2)
    v = v.reshape(B, T, H, hs).transpose(1, 2)
    y = (torch.softmax((q @ k.transpose(-2, -1)) * (8.0 / C), dim=-1) * m_attn) @ v
    y = y.transpose(1, 2).contiguous().view(B, T, C)
    x1 = (xf + F.linear(y, wf[pre + "attn.c_proj.weight"]) * m_res).reshape(-1, C)[rows]
    a = R.gelu_erf(F.linear(R.layer_norm(x1, wf[pre + "ln_2.weight"]), wf[pre + "mlp.c_fc.weight"]))
    ref = x1 + F.linear(a, wf[pre + "mlp.c_proj.weight"]) * m_mlp
    ref.backward(dy.float())
    from omnibiote_amd.model import rope_tables
    o = ops()
    params = tuple(w[pre + n].to(DEV) for n in names)
    rope = rope_tables(tab.to(DEV))
    spec = o.MaskSpec()
    rows_d = rows.to(DEV)
    yg, act = o.block_fwd(x.to(DEV), params, rope, H, spec, p, SEED, out_rows=rows_d)
    close(yg, ref, atol=4e-2, rtol=2.0 ** -6, what="block rows dropout fwd")
    dx, grads = o.block_bwd(x.to(DEV), dy.to(DEV), act, params, rope, H, spec, dropout_p=p, dropout_seed=SEED, out_rows=rows_d)
    close(dx, xf.grad, atol=2e-2, rtol=2.0 ** -5, what="block rows dropout dx")
    for n, gg in zip(names, grads):
        gr = wf[pre + n].grad
        close(gg, gr, atol=0.03 * gr.abs().max().item() + 1e-3, rtol=2.0 ** -5, what="block rows dropout d" + n)


@pytest.mark.parametrize("hs,mode", [(64, "complex"), (128, "cos_only")])
def test_gemm_rope_epilogue_equals_projection_then_rope(hs, mode):
    """c_attn with RoPE fused in its epilogue == plain projection followed by the stand-alone RoPE kernel."""
    B, T, H = 2, 77, 2
    C = H * hs
    x, w = rnd(B * T, C, seed=61), rnd(3 * C, C, seed=62, scale=0.1)
    tab = R.rope_table(hs, 128)
    if mode == "cos_only":
        tab = R.cast_rope_table(tab, BF)
    from omnibiote_amd.model import rope_tables
    cos, sin = rope_tables(tab.to(DEV))
    o = ops()
    fused = o.gemm(x.to(DEV), w.to(DEV), B * T, 3 * C, C, True, True, L().EPI_ROPE_QK, rope=(cos, sin, T, hs))
    plain = o.linear_fwd(x.to(DEV), w.to(DEV))
    o.rope_qk_(plain, cos, sin, B, T, H, hs)
    assert torch.equal(fused, plain)


def test_masked_ce_reused_gradient_buffer():
    """A dlogits buffer reused across calls: rows unmasked in consecutive calls are skipped, yet the buffer always equals
    what a fresh full write would produce."""
    rows, V = 96, 1024
    o = ops()
    buf = o.DLogitsBuffer()
    for it in range(4):
        logits = rnd(rows, V, seed=10 + it, scale=2.0).to(DEV)
        tgt = torch.from_numpy(np.random.default_rng(it).integers(0, V, size=rows)).to(DEV)
        mask = torch.from_numpy(np.random.default_rng(100 + it).random(rows) < 0.2).to(DEV)
        mask[it] = True
        loss_a, fresh = o.masked_ce(logits, tgt, mask, 2)
        loss_b, reused = o.masked_ce(logits, tgt, mask, 2, reuse=buf)
        assert torch.equal(fresh, reused) and loss_a.item() == loss_b.item()


def test_fused_adamw_reference_rounding_is_torch_adamw_on_bf16_tensors():
    """What the reference literally runs: torch.optim.AdamW (MuAdamW underneath, train_encoder.py:195-199) on bf16
    parameters with bf16 moments, preceded by clip_grad_norm_ (:316).  FusedAdamW(rounding="reference") must reproduce it
    over several steps — not a formula typed into the test, the optimizer itself on CPU bf16 tensors.  Integer-grade bar:
    after 12 steps (clipping active, LinearLR, two parameter groups) at least 99.9 % of all parameter and moment elements
    are BIT-identical (measured: all of them) and none is further than 2 bf16 ulps away."""
    from omnibiote_amd import train_encoder as TE
    shapes = [(256, 128), (1024,), (64, 512), (8,)]
    steps = 12
    gen = torch.Generator().manual_seed(5)
    p0 = [torch.randn(s, generator=gen).to(BF) for s in shapes]
    grads = [[(torch.randn(s, generator=gen) * (0.3 if t % 3 else 3.0)).to(BF) for s in shapes] for t in range(steps)]   # some steps clip, some do not
    lr, wd, betas, eps = 3e-3, 1e-2, (0.9, 0.999), 1e-8
    cpu = [torch.nn.Parameter(x.clone()) for x in p0]
    gpu = [torch.nn.Parameter(x.clone().to(DEV)) for x in p0]
    groups = lambda ps: [{"params": ps[:2], "lr": lr / 4, "weight_decay": wd * 4}, {"params": ps[2:], "lr": lr, "weight_decay": wd}]
    ref = torch.optim.AdamW(groups(cpu), lr=lr, betas=betas, eps=eps, weight_decay=wd)
    fused = TE.FusedAdamW(groups(gpu), lr=lr, betas=betas, eps=eps, weight_decay=wd, rounding="reference")
    sched_r = torch.optim.lr_scheduler.LinearLR(ref, start_factor=1.0, end_factor=0.0, total_iters=40)
    sched_f = torch.optim.lr_scheduler.LinearLR(fused, start_factor=1.0, end_factor=0.0, total_iters=40)
    for t in range(steps):
        for q, r, gq in zip(cpu, gpu, grads[t]):
            q.grad = gq.clone()
            r.grad = gq.clone().to(DEV)
        torch.nn.utils.clip_grad_norm_(cpu, 1.0)
        ref.step(); sched_r.step()
        fused.step(max_norm=1.0); sched_f.step()
    total = same = 0
    worst = {"p": 0.0, "exp_avg": 0.0, "exp_avg_sq": 0.0}
    for q, r in zip(cpu, gpu):
        states = [("p", q.data, r.data), ("exp_avg", ref.state[q]["exp_avg"], fused.state[r]["exp_avg"]),
                  ("exp_avg_sq", ref.state[q]["exp_avg_sq"], fused.state[r]["exp_avg_sq"])]
        for kind, a, b in states:
            a, b = a.float(), b.float().cpu()
            total += a.numel()
            same += int((a == b).sum())
            ulp = torch.maximum(a.abs(), b.abs()) * 2.0 ** -7 + 1e-30
            # a parameter is the running sum of updates of size ~lr: near zero its own ulp is far finer than one ulp of an
            # update, so parameters get an absolute allowance of 2 % of one lr-sized step on top of the two ulps
            slack = 0.02 * lr if kind == "p" else 0.0
            worst[kind] = max(worst[kind], ((a - b).abs() / (2.0 * ulp + slack)).max().item())
    print("fused AdamW (reference rounding) vs torch.optim.AdamW on bf16 CPU tensors:", same, "of", total, "elements identical; worst / bar:", worst)
    assert max(worst.values()) <= 1.0, worst
    assert same >= 0.999 * total, (same, total)


@pytest.mark.parametrize("kind,expect", [("block_diagonal", 1), ("key_padding", 1), ("additive_values", 0), ("per_head", 0),
                                         ("hole", 0), ("fully_masked_row", 0)])
def test_dense_mask_that_is_a_range_mask_takes_the_range_kernels(kind, expect):
    """obte_mask_bounds decides on the device whether a dense additive mask IS a range mask (contiguous runs of exact zeros,
    all heads alike, every key's queries contiguous too); if so the range kernels serve it, else the dense ones — both are
    launched and one set returns at once.  Whatever the route, forward and backward must match the oracle, and the flag must
    say what the mask is.  key_padding is NOT symmetric: it exercises the per-key query ranges of the dK/dV kernel."""
    B, H, T, hs = 2, 2, 192, 64
    C = H * hs
    scale = 8.0 / C
    qkv, q, k, v = _attn_case(B, T, H, hs, seed=31)
    NEG = -1e9
    m = torch.full((B, H, T, T), NEG)
    if kind == "block_diagonal":
        tokens = np.random.default_rng(1).integers(20, 100, size=(B, T))
        tokens[0, [50, 120]] = R.EOS_TOKEN
        tokens[1, [7, 100, 150]] = R.EOS_TOKEN
        m = _blocks_to_masks(tokens, T)[0].unsqueeze(1).expand(B, H, T, T).clone()
    elif kind == "key_padding":
        for b, n in enumerate((150, 77)):
            m[b, :, :, :n] = 0.0
    elif kind == "additive_values":
        m[:] = 0.0
        m[:, :, :, 100:] = NEG
        m[:, :, 5, 10] = -0.5
    elif kind == "per_head":
        m[:, 0, :, :100] = 0.0
        m[:, 1, :, :120] = 0.0
    elif kind == "hole":
        m[:, :, :, :150] = 0.0
        m[:, :, 30, 60:70] = NEG
    elif kind == "fully_masked_row":
        m[:, :, :, :150] = 0.0
        m[0, :, 3, :] = NEG
    o = ops()
    md = m.to(BF).to(DEV)
    if kind in ("block_diagonal", "key_padding", "hole"):
        md = md[:, :1].expand(B, H, T, T)            # the reference's stride-0 head view (train_encoder.py:292)
    spec = o.MaskSpec.from_user(md, B, T, H, DEV)
    assert int(spec.exact.item()) == expect, kind
    qf, kf, vf = q.requires_grad_(True), k.requires_grad_(True), v.requires_grad_(True)
    ref = R.attention(qf, kf, vf, scale, m.to(BF).float())
    d_o = rnd(B, T, C, seed=98)
    ref.backward(d_o.reshape(B, T, H, hs).transpose(1, 2).float())
    got, lse = o.attn_fwd(qkv.to(DEV), B, T, H, hs, scale, spec)
    rows_ok = torch.ones(B, T, dtype=torch.bool)
    if kind == "fully_masked_row":
        rows_ok[0, 3] = False                         # uniform softmax over raw scores there: forward checked, backward row excluded
    close(got, ref.transpose(1, 2).reshape(B, T, C), atol=6e-3, what=f"{kind} fwd")
    dqkv = o.attn_bwd(qkv.to(DEV), got, d_o.to(DEV), lse, B, T, H, hs, scale, spec)
    dref = torch.cat([g.transpose(1, 2).reshape(B, T, C) for g in (qf.grad, kf.grad, vf.grad)], dim=2)
    if kind != "fully_masked_row":
        close(dqkv, dref, atol=1.5e-2, rtol=2.0 ** -6, what=f"{kind} bwd")
