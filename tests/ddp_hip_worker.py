"""Worker process of tests/test_hip_ddp.py (not collected by pytest): one data-parallel rank of the PRODUCT train step —
the HIP model, the fused CE kernel, in-place gradient accumulation under DDP.no_sync(), two-stream micro-batch
pipelining, FusedAdamW — under torch.distributed with ``--backend gloo`` (both ranks may share one GPU; gradients cross
the host) or ``--backend nccl`` (RCCL, one GPU per rank).  Rank 0 saves losses, final gradients and weights.

    RANK/WORLD_SIZE/MASTER_ADDR/MASTER_PORT from the environment;  argv: backend out_path gpus_available
"""
import os
import sys
import warnings

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))   # hash_weights: the closed-form initial weights the fixtures use

C, H, LYR, V, T, MINI = 128, 2, 2, 512, 64, 2
ROWS_TOTAL, STEPS = 16, 2          # world 2: 8 rows per rank = 4 micro-batches (> 2, so the two-stream pipeline engages)


def build_model(dev):
    import omnibiote_ref as R
    from omnibiote_amd.model import OmniBioTA, OmniBioTAConfig
    from omnibiote_amd.mup_compat import set_base_shapes
    c = OmniBioTAConfig(); c.block_size, c.vocab_size, c.n_layer, c.n_head, c.n_embd, c.dropout, c.flash = T, V, LYR, H, C, 0.0, True
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        m = OmniBioTA(c)
        cb = OmniBioTAConfig(); cb.block_size, cb.vocab_size, cb.n_layer, cb.dropout, cb.flash = T, V, LYR, 0.0, True
        cb.n_embd, cb.n_head = 24, 3
        base = OmniBioTA(cb)
        cb.n_embd, cb.n_head = 48, 12
        delta = OmniBioTA(cb)
    set_base_shapes(m, base, delta=delta, rescale_params=False)
    m.load_state_dict(R.hash_weights(R.RefConfig(block_size=T, vocab_size=V, n_layer=LYR, n_head=H, n_embd=C)), strict=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.to(torch.bfloat16)
    return m.to(dev)


def data():
    from omnibiote_amd import train_encoder as TE
    rng = np.random.default_rng(5)
    ids = torch.from_numpy(TE.synthetic_rows(ROWS_TOTAL, T, V, rng, single_document=False))
    ids[:, T // 2] = 3
    mlm = torch.from_numpy(rng.random((ROWS_TOTAL, T)) < 0.15)
    mlm[:, 1] = True
    return ids, mlm


def run(backend, out_path, gpus):
    from omnibiote_amd import train_encoder as TE
    from omnibiote_amd.mup_compat import mu_param_groups
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = rank % max(gpus, 1) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    m = build_model(dev)
    model = TE.wrap_ddp(m, local, bucket_cap_mb=1) if world > 1 else m     # 1 MB buckets: several buckets even at this size
    lr, wd = 1e-2, 1e-2
    opt = TE.FusedAdamW(mu_param_groups(list(m.parameters()), lr, wd), lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    step = TE.TrainStep(model, opt, None, mini_batch_size=MINI, n_head=H, pipeline_streams=2)
    ids, mlm = data()
    per = ROWS_TOTAL // world
    mine, mine_mask = ids[rank * per:(rank + 1) * per].to(dev), mlm[rank * per:(rank + 1) * per].to(dev)
    losses = []
    for s in range(STEPS):
        out = step(mine, mlm_mask=mine_mask)
        t = out["loss"].clone()
        if world > 1:
            dist.all_reduce(t)
        losses.append(t.item() / world)
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"losses": losses, "g": {k: p.grad.detach().float().cpu() for k, p in m.named_parameters()},
                    "w": {k: p.detach().float().cpu() for k, p in m.named_parameters()}}, out_path)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    run(sys.argv[1], sys.argv[2], int(sys.argv[3]))
