"""CPU tests of the host-side logic: the mask builder against the reference's golden masks and the oracle, the
drop-in class surface, the muP stand-ins, and the C-ABI library's symbol table."""
import ctypes
import io
import os
import re

import numpy as np
import pytest
import torch

import omnibiote_ref as R
from omnibiote_amd import _lib, masks, mup_compat
from omnibiote_amd.model import OmniBioTA, OmniBioTAConfig, precompute_freqs_cis, rope_tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_range_mask_builder_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "attention_masks.npz"))
    for n in sorted({k.split("/")[0] for k in g.files}):
        tok, padding, allowed = torch.from_numpy(g[n + "/tokens"]), bool(g[n + "/padding"]), g[n + "/allowed"]
        rm = masks.RangeMask.from_tokens(tok, padding=padding)
        dense = rm.dense(torch.float32)
        np.testing.assert_array_equal((dense == 0).numpy(), allowed, err_msg=n)
        assert set(np.unique(dense.numpy()).tolist()) <= {0.0, -1e9}
        # in-place, signature-compatible builder
        am = torch.ones(allowed.shape, dtype=torch.float32) * -1e9
        out = masks.create_attention_mask(am, tok, padding=padding)
        assert out is am
        np.testing.assert_array_equal((am == 0).numpy(), allowed, err_msg=n)
        # round trip dense -> ranges -> dense
        back = masks.RangeMask.from_dense(dense.unsqueeze(1).expand(-1, 3, -1, -1))
        np.testing.assert_array_equal((back.dense(torch.float32) == 0).numpy(), allowed, err_msg=n)


@pytest.mark.parametrize("seed", range(8))
def test_range_mask_builder_random_vs_oracle(seed):
    rng = np.random.default_rng(seed)
    B, T = int(rng.integers(1, 6)), int(rng.integers(4, 70))
    tok = rng.integers(4, 30, size=(B, T))
    tok[rng.random((B, T)) < 0.12] = R.EOS_TOKEN
    for padding in (False, True):
        blocks = R.document_blocks(tok, padding=padding)
        want = (R.dense_mask_from_blocks(blocks, T) == 0).numpy()
        got = (masks.RangeMask.from_tokens(torch.from_numpy(tok), padding=padding).dense(torch.float32) == 0).numpy()
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("seed", range(4))
def test_range_mask_builder_grouped_equals_per_mini_batch(seed):
    """One pass over all rows of an optimizer step (group = mini-batch size) must give exactly the ranges of building
    each mini-batch on its own — including the reference's first-row exception, which restarts with every mini-batch."""
    rng = np.random.default_rng(100 + seed)
    mini, n, T = int(rng.integers(1, 5)), int(rng.integers(1, 5)), int(rng.integers(4, 70))
    tok = rng.integers(4, 30, size=(mini * n, T))
    tok[rng.random(tok.shape) < 0.15] = R.EOS_TOKEN
    for padding in (False, True):
        allr = masks.RangeMask.from_tokens(torch.from_numpy(tok), padding=padding, group=mini).key_ranges
        for j in range(n):
            one = masks.RangeMask.from_tokens(torch.from_numpy(tok[j * mini:(j + 1) * mini]), padding=padding).key_ranges
            assert torch.equal(allr[j * mini:(j + 1) * mini], one)


def test_from_dense_rejects_non_block_masks():
    m = torch.full((1, 4, 4), -1e9)
    m[0, 0, 0] = 0; m[0, 0, 2] = 0
    with pytest.raises(ValueError):
        masks.RangeMask.from_dense(m)


def make_cfg(**kw):
    c = OmniBioTAConfig()
    c.flash = True
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def test_class_surface_and_state_dict_keys():
    c = OmniBioTAConfig()
    assert (c.block_size, c.vocab_size, c.n_layer, c.n_head, c.n_embd, c.dropout, c.bias, c.autoregressive,
            c.checkpoint_freq) == (2048, 65536, 12, 12, 1024, 0.1, False, False, 0)
    c.n_head = 8
    with pytest.raises(AttributeError):   # no `flash` field: callers set it ad hoc (train_encoder.py:152)
        OmniBioTA(c)
    cfg = make_cfg(block_size=32, vocab_size=128, n_layer=2, n_head=2, n_embd=128, dropout=0.0)
    m = OmniBioTA(cfg)
    want = ["transformer.wte.weight"]
    for i in range(2):
        p = f"transformer.h.{i}."
        want += [p + "ln_1.weight", p + "attn.freqs_cis", p + "attn.c_attn.weight", p + "attn.c_proj.weight",
                 p + "ln_2.weight", p + "mlp.c_fc.weight", p + "mlp.c_proj.weight"]
    want += ["transformer.ln_f.weight", "lm_head.weight"]
    assert list(m.state_dict().keys()) == want
    assert [n for n, _ in m.named_parameters()] == R.param_names(2)
    assert m.state_dict()["transformer.h.0.attn.freqs_cis"].dtype == torch.complex64
    assert m.transformer.h[0].attn.n_head == 2 and m.transformer.h[0].attn.n_embd == 128
    assert m.get_num_params() == sum(p.numel() for p in m.parameters()) - 128 * 128
    assert m.get_num_params(non_embedding=False) == sum(p.numel() for p in m.parameters())
    cfg.flash = False
    m2 = OmniBioTA(cfg)
    assert "transformer.h.0.attn.bias" in m2.state_dict()   # the reference's extra tril buffer (model.py:95)
    # mutate-and-reuse the same config object for the muP base/delta models (train_encoder.py:158-164)
    cfg.flash = True
    cfg.n_embd, cfg.n_head = 24, 3
    base = OmniBioTA(cfg)
    cfg.n_embd, cfg.n_head = 48, 12
    delta = OmniBioTA(cfg)
    assert base.lm_head.weight.shape == (128, 24) and delta.transformer.h[0].attn.c_attn.weight.shape == (144, 48)


def test_constructor_rng_order_matches_reference_layout():
    """Same parameter creation order as the reference => same weights from the same seed (wte N(0,1), then per
    block c_attn, attn.c_proj, c_fc, mlp.c_proj kaiming-uniform, then lm_head)."""
    cfg = make_cfg(block_size=16, vocab_size=64, n_layer=1, n_head=2, n_embd=128, dropout=0.0)
    torch.manual_seed(5)
    m = OmniBioTA(cfg)
    torch.manual_seed(5)
    wte = torch.nn.Embedding(64, 128)
    c_attn = torch.nn.Linear(128, 384, bias=False)
    assert torch.equal(m.transformer.wte.weight, wte.weight)
    assert torch.equal(m.transformer.h[0].attn.c_attn.weight, c_attn.weight)


def test_module_behaviour_to_deepcopy_pickle_and_loud_cpu_failure():
    import copy
    cfg = make_cfg(block_size=16, vocab_size=64, n_layer=1, n_head=2, n_embd=128, dropout=0.0)
    m = OmniBioTA(cfg)
    m2 = copy.deepcopy(m)
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    assert torch.equal(m3.lm_head.weight, m.lm_head.weight) and torch.equal(m2.lm_head.weight, m.lm_head.weight)
    m3.load_state_dict(m.state_dict())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 4, dtype=torch.long))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.encode(torch.zeros(1, 4, dtype=torch.long))
    with pytest.raises(AssertionError):
        m(torch.zeros(1, 17, dtype=torch.long))   # T > block_size
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.to(torch.bfloat16)
    f = m.transformer.h[0].attn.freqs_cis
    assert f.dtype == torch.bfloat16 and not f.is_complex()   # cos only after the cast
    cos, sin = rope_tables(f)
    assert float(sin.abs().max()) == 0.0
    want = R.cast_rope_table(R.rope_table(64, 16), torch.bfloat16).float()
    assert torch.equal(cos, want)
    cos_c, sin_c = rope_tables(precompute_freqs_cis(64, 16))
    assert torch.equal(cos_c, R.rope_table(64, 16).real) and torch.equal(sin_c, R.rope_table(64, 16).imag)


def test_mup_standins_width_mult_and_groups():
    cfg = make_cfg(block_size=16, vocab_size=64, n_layer=1, n_head=2, n_embd=128, dropout=0.0)
    m = OmniBioTA(cfg)
    w0 = m.lm_head.weight.detach().clone()
    with pytest.raises(AssertionError):
        m.lm_head.width_mult()   # set_base_shapes not called yet
    cfg.n_embd, cfg.n_head = 24, 3
    base = OmniBioTA(cfg)
    cfg.n_embd, cfg.n_head = 48, 12
    delta = OmniBioTA(cfg)
    mup_compat.set_base_shapes(m, base, delta=delta)
    wm = 128 / 24
    assert abs(m.lm_head.width_mult() - wm) < 1e-9
    torch.testing.assert_close(m.lm_head.weight, w0 * wm ** 0.5)
    opt = mup_compat.MuAdamW(m.parameters(), lr=0.01, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8)
    assert len(opt.param_groups) == 2
    mat, vec = opt.param_groups
    assert abs(mat["lr"] - 0.01 / wm) < 1e-12 and abs(mat["weight_decay"] - 0.01 * wm) < 1e-12
    assert vec["lr"] == 0.01 and vec["weight_decay"] == 0.01
    assert {tuple(p.shape) for p in mat["params"]} == {(384, 128), (128, 128), (512, 128), (128, 512)}
    assert sum(p.numel() for p in vec["params"]) == 64 * 128 * 2 + 3 * 128


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    lib = _lib.lib()
    assert lib.obte_abi_version() == 1
    header = open(os.path.join(ROOT, "include", "omnibiote_hip.h")).read()
    declared = set(re.findall(r"\b(obte_[a-z0-9_]+)\s*\(", header))
    declared -= {"obte_stream"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in the header but not exported"
    # argument validation happens on the host, before any launch (no GPU needed): NULL descriptor -> EINVAL
    assert lib.obte_gemm_bf16(None, None) == -1
    assert b"null" in lib.obte_last_error()


def test_config1_cpu_gloo_harness_is_launchable_and_fails_loudly_at_the_model(tmp_path, monkeypatch):
    """BASELINE config 1 (tiny 2L/128d/2h ctx=128, CPU torchrun world_size=1, --disable_flash): the harness itself —
    process group on gloo, muP model set-up, optimizer groups, scheduler, data source, step loop — must start on a CPU
    world; the model has no CPU path by design and must say so at its first forward instead of computing something."""
    import torch.distributed as dist
    from omnibiote_amd import train_encoder as TE
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29533")
    args = TE.parse_args(["--n_layer", "2", "--n_embd", "128", "--n_head", "2", "--ctx_len", "128", "--batch_size", "16", "--mini_batch_size", "8",
                          "--disable_flash", "--backend", "gloo", "--device", "cpu", "--max_steps", "2", "--save_name", ""])
    with pytest.raises(RuntimeError, match="no CPU\\s+fallback|needs GPU tensors"):
        TE.run(args)
    assert not dist.is_initialized()      # torn down on the way out


def test_grad_policy_is_captured_per_graph_not_per_process():
    """model.GradPolicy: the switches of the backward pass are a snapshot taken where the graph is built (a context
    variable), immutable afterwards; another thread starts from the default and never sees this thread's setting."""
    import threading
    from omnibiote_amd import model as M
    assert M.current_grad_policy().accumulate is False and M.current_grad_policy().ln_mode == 0
    store = M.LnPartialStore()
    seen = {}
    with M.accumulate_grads_inplace(True, 2, store=store):
        inner = M.current_grad_policy()
        assert inner.accumulate and inner.ln_mode == 2 and inner.store is store
        t = threading.Thread(target=lambda: seen.setdefault("other", M.current_grad_policy()))
        t.start(); t.join()
        with M.accumulate_grads_inplace(False):
            assert not M.current_grad_policy().accumulate
        assert M.current_grad_policy() is inner
    assert seen["other"].accumulate is False and seen["other"].ln_mode == 0
    assert M.current_grad_policy().accumulate is False
    with pytest.raises(AttributeError):
        inner.accumulate = False
    p = torch.nn.Parameter(torch.zeros(4, 4))
    p.grad = torch.ones(4, 4)
    assert M._grad_slot(p, inner) is p.grad and M._grad_slot(p, M.current_grad_policy()) is None and M._grad_slot(p) is None
    box_order = torch.arange(3, dtype=torch.int32)
    with M.embedding_order(box_order):
        assert M._embedding_order.get()[0] is box_order
    assert M._embedding_order.get() is None


def test_tuner_ranking_breaks_ties_towards_the_deeper_ring_and_lists_the_masked_readout_shapes():
    """tune.rank_candidates: candidates within 3 % of the fastest are a tie, taken in the order half-tile ring (3), K-tile
    ring (2), two-workgroups-per-CU ring (4), first structure (1); outside the band the fastest wins.  tune.model_gemm_shapes
    lists the three products of the masked-positions readout (about 15 % of the rows) beside the block's shapes."""
    from omnibiote_amd import tune
    r = tune.rank_candidates({(2, 256, 12): 0.2174, (3, 256, 12): 0.2196, (2, 128, 8): 0.2500, (1, 128, 1): 0.9})
    assert r[0][1:] == (3, 256, 12) and [x[1:] for x in r[1:]] == [(2, 256, 12), (2, 128, 8), (1, 128, 1)]
    r = tune.rank_candidates({(2, 256, 1): 0.100, (3, 256, 1): 0.104, (4, 128, 1): 0.1005})
    assert r[0][1:] == (2, 256, 1)                       # 4 % slower is not a tie; among the tied, 2 goes before 4
    r = tune.rank_candidates({(3, 256, 2): 0.1029, (3, 256, 4): 0.1021, (1, 128, 1): 0.1000})
    assert r[0][1:] == (3, 256, 4)                       # the faster of the preferred structure's own candidates
    shapes = tune.model_gemm_shapes(8192, 1024, 65536)
    mm = [s for s in shapes if 1232 in s[:3]]
    assert sorted((s[0], s[1], s[2], s[3], s[4]) for s in mm) == sorted([
        (1232, 1024, 65536, True, False), (65536, 1024, 1232, False, False), (1232, 65536, 1024, True, True),      # the readout
        (1232, 4096, 1024, True, True), (1232, 1024, 4096, True, True), (1232, 4096, 1024, True, False), (1232, 1024, 4096, True, False),
        (1024, 4096, 1232, False, False), (4096, 1024, 1232, False, False),                                          # the last block's MLP half
        (1232, 1024, 1024, True, True), (1232, 1024, 1024, True, False), (1024, 1024, 1232, False, False),           # and its attention projection
        (1232, 1024, 1024, True, True)])                                                                             # and its q rows (epilogue NONE)


def test_tuner_lists_the_192_wide_and_the_persistent_structure_only_where_they_apply():
    """tune.candidates: the 256 x 192 tile (structure 2) for x W^T products whose N is a multiple of 192 and whose epilogue it
    implements; the persistent continuous-ring structure 7 for whole 256 x 256 tiles, at least one per CU, of the x W^T and dy W
    layouts with the epilogues it is built for."""
    from omnibiote_amd import tune
    L = _lib
    c = tune.candidates(4096, 3072, 1024, L.EPI_ROPE_QK, True, True)
    assert (2, 192, 1) in c and (7, 256, 1) not in c                      # 16 x 12 = 192 tiles: fewer than the CUs
    c = tune.candidates(32768, 3072, 1024, L.EPI_ROPE_QK, True, True)
    assert (2, 192, 1) in c and (7, 256, 1) in c
    assert (2, 192, 1) not in tune.candidates(8192, 3072, 1024, L.EPI_NONE, True, False)       # x W layouts only
    assert (2, 192, 1) not in tune.candidates(8192, 3072, 1024, L.EPI_GELU_BWD, True, True)
    assert (2, 192, 1) not in tune.candidates(8192, 4096, 1024, L.EPI_GELU, True, True)        # 4096 is not a multiple of 192
    assert (7, 256, 1) in tune.candidates(8192, 4096, 1024, L.EPI_GELU, True, True)            # 512 tiles
    assert (7, 256, 1) not in tune.candidates(8192, 1024, 1024, L.EPI_ADD, True, True)         # 128 tiles: fewer than the CUs
    assert (7, 256, 1) in tune.candidates(32768, 4096, 1024, L.EPI_GELU_BWD, True, False)
    assert (7, 256, 1) not in tune.candidates(32768, 4096, 1024, L.EPI_GELU_BWD, True, True)   # GELU' exists for dy W only
    assert (7, 256, 1) not in tune.candidates(32768, 1024, 1024, L.EPI_NONE, False, False)     # weight-gradient layout: not built
    assert (7, 256, 1) not in tune.candidates(8200, 4096, 1024, L.EPI_NONE, True, True)        # ragged rows
    assert (7, 256, 1) not in tune.candidates(8192, 4096, 192, L.EPI_NONE, True, True)         # three K-tiles: too short
    assert not any(v in (5, 6) for v, _, _ in tune.candidates(32768, 4096, 1024, L.EPI_NONE, True, True))   # structures removed


def test_masked_gradient_handoff_checks_storage_shape_version_probability_and_seed(monkeypatch):
    """model._GradHandOff (dropout: a block's dx under the mask of the block below; one object per forward call, nothing global):
    an entry is taken only by the gradient tensor it was stored for, untouched since (its version counter: an in-place
    accumulation of a second contribution or a hook editing the gradient bump it), with the taker's own probability and seed;
    anything else gets None (and the block masks its own gradient); OBTE_DROPOUT_HANDOFF=0 turns the hand-off off; two
    objects never see each other's entries."""
    from omnibiote_amd import model as M
    assert not hasattr(M, "_masked_grad")            # the process-wide table of rounds 3-4 is gone
    h = M._GradHandOff()
    dx, dxm = torch.zeros(4, 8), torch.ones(4, 8)
    h.put(3, dx, dxm, 0.1, 77)
    assert h.take(2, dx, (0.1, 77)) is None                                  # another block's slot
    assert h.take(3, torch.zeros(4, 8), (0.1, 77)) is None                   # another tensor (and the entry is consumed)
    assert h.take(3, dx, (0.1, 77)) is None
    h.put(3, dx, dxm, 0.1, 77)
    assert h.take(3, dx, (0.1, 78)) is None                                  # another seed
    h.put(3, dx, dxm, 0.1, 77)
    assert h.take(3, dx, (0.2, 77)) is None                                  # another probability
    h.put(3, dx, dxm, 0.1, 77)
    assert h.take(3, dx.view(8, 4), (0.1, 77)) is None                       # same storage, another shape
    h.put(3, dx, dxm, 0.1, 77)
    dx.add_(1.0)                                                             # what autograd's InputBuffer does with a second contribution
    assert h.take(3, dx, (0.1, 77)) is None
    h.put(3, dx, dxm, 0.1, 77)
    assert h.take(3, dx, (0.1, 77)) is dxm and h.take(3, dx, (0.1, 77)) is None   # consumed once
    h.put(3, dx, None, 0.1, 77)
    assert not h.slots
    other = M._GradHandOff()
    h.put(1, dx, dxm, 0.1, 5)
    assert other.take(1, dx, (0.1, 5)) is None and h.take(1, dx, (0.1, 5)) is dxm   # per forward call, not per process
    monkeypatch.setenv("OBTE_DROPOUT_HANDOFF", "0")
    h.put(3, dx, dxm, 0.1, 77)
    assert h.take(3, dx, (0.1, 77)) is None


def test_forward_rows_contract_is_checked_on_the_host():
    """OmniBioTA.forward(rows=) / Block.forward(out_rows=): shape, dtype, device and count of the list are refused on the host
    before anything is launched (the values — ascending, distinct, in range — are the caller's contract, verified only under
    OBTE_CHECK_ROWS=1, which needs the device)."""
    import pytest
    import torch
    from omnibiote_amd.model import _check_rows
    for bad in (torch.zeros(3, dtype=torch.int32), torch.zeros((2, 2), dtype=torch.int64), [0, 1], torch.zeros(3, dtype=torch.int64)):
        with pytest.raises(ValueError):
            _check_rows(bad, 16, 1)   # the last one: a CPU tensor (the model lives on the GPU)


def test_attention_backward_isa_keeps_its_hand_counted_waits_honest(tmp_path):
    """The one-kernel attention backward counts its own vector-memory operations (s_waitcnt vmcnt(N) by hand).  That only holds
    while hipcc puts nothing of its own into the slice loop's queue and leaves the registers of the asm-issued loads alone until the
    wait that releases them; tools/fused_audit.py checks both on the ISA of every instantiation (no scratch access inside the loop;
    no instruction touching an in-flight destination).  Compiled here exactly as csrc/Makefile compiles the file."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "omnibiote_amd", "csrc", "attention_bwd_fused.hip")
    out = str(tmp_path / "fused.s")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only",
                        src, "-o", out], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    a = subprocess.run([sys.executable, os.path.join(root, "tools", "fused_audit.py"), out], capture_output=True, text=True, timeout=300)
    assert a.returncode == 0 and "audit ok" in a.stdout, a.stdout[-3000:]
    assert a.stdout.count("scratch accesses 0") == 4, a.stdout   # no mask / key ranges, each with and without dropout


def test_attention_backward_workspace_is_zero_where_the_one_kernel_form_does_not_apply():
    """obte_attn_bwd_ws_bytes answers 0 — "allocate nothing, the two-kernel form runs" — wherever the one-kernel backward does not
    apply (other head sizes, more than 256 query slices, rows whose byte offsets pass 32 bits), never a sentinel a caller would
    hand to an allocator; the block's backward workspace, which contains it, stays a sane number at long contexts (ctx_len is a
    free argument of the reference: train_encoder.py:444)."""
    lib = _lib.lib()
    assert lib.obte_attn_bwd_ws_bytes(8, 1024, 8, 128) > 0
    assert lib.obte_attn_bwd_ws_bytes(8, 8192, 8, 128) > 0
    assert lib.obte_attn_bwd_ws_bytes(1, 16384, 8, 128) == 0        # 512 slices: the pair of kernels
    assert lib.obte_attn_bwd_ws_bytes(1, 8224, 8, 128) == 0
    assert lib.obte_attn_bwd_ws_bytes(8, 1024, 8, 64) == 0
    assert lib.obte_attn_bwd_ws_bytes(0, 1024, 8, 128) == 0
    small = lib.obte_block_bwd_ws_bytes(1, 8192, 1024, 8)
    for T in (16384, 32768):
        b = lib.obte_block_bwd_ws_bytes(1, T, 1024, 8)
        assert 0 < b < (1 << 36), b
        assert b < 8 * small * (T // 8192)                          # grows with the activations, not with a sentinel


def test_block_activation_buffer_leaves_out_the_dropout_keep_bits_without_dropout():
    """obte_block_act_bytes_p(p = 0) ends before the attention dropout's keep bits (B H ceil(T/32) T words — 134 MB per block at
    B = 8, T = 4096); with p > 0 it is obte_block_act_bytes."""
    lib = _lib.lib()
    for B, T, C, H in ((8, 1024, 1024, 8), (8, 4096, 1024, 8), (2, 1024, 2048, 16)):
        full, bits = lib.obte_block_act_bytes(B, T, C, H), lib.obte_attn_drop_bits_bytes(B, T, H)
        assert lib.obte_block_act_bytes_p(B, T, C, H, 0.1) == full
        lean = lib.obte_block_act_bytes_p(B, T, C, H, 0.0)
        assert full - lean == (bits + 255) // 256 * 256 and lean > 0
