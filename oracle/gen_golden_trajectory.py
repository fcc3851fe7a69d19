"""Golden TRAINING TRAJECTORY generator.  TEST INFRASTRUCTURE ONLY — runs in the build container (needs /root/reference).

Runs the reference's own training arithmetic for a few optimizer steps and records the loss curve:
  * the model is the *imported reference* ``OmniBioTA`` (training/model.py) in its training regime: built in fp32,
    ``.to(torch.bfloat16)`` (train_encoder.py:170 — parameters bf16, RoPE buffer degraded to cos-only);
  * the step is the reference's loop body restated line for line from train_encoder.py:268-318 — host Bernoulli MLM
    corruption (:273-279), the TorchScript ``create_attention_mask`` imported from the reference (:288-292), the three
    loss lines (:301-305), ``clip_grad_norm_(1.0)``, optimizer, LinearLR (:316-318);
  * the optimizer is ``torch.optim.AdamW`` on the bf16 parameters (bf16 moments, updated op by op in bf16: what the
    reference's MuAdamW does underneath) with the two parameter groups MuAdamW would build passed explicitly (matrix-like
    parameters lr / width_mult and weight_decay * width_mult — ``mup`` itself is not installable here: "parity
    unpinned" for that grouping; with --force_lr the reference uses exactly torch.optim.AdamW, :196-197).
Two runs are recorded: the flash path (SDPA) and the reference's manual attention path (--disable_flash).  They are the
same mathematics in a different bf16 rounding order, so their difference is the reference's OWN run-to-run spread; the
tests quote it as the bar the HIP path has to meet.

Usage:  python oracle/gen_golden_trajectory.py    (writes tests/golden/trajectory_tiny_bf16.npz)
"""
import os
import sys
import warnings

import numpy as np
import torch
import torch.nn.functional as F

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)
import omnibiote_ref as R  # noqa: E402
from gen_golden import _import_reference, build_ref  # noqa: E402

CFG = dict(block_size=64, vocab_size=512, n_layer=2, n_head=2, n_embd=128)
ROWS, MINI, T, STEPS, N_BATCHES = 8, 4, 64, 30, 2
LR, WD, BETAS, EPS, TOTAL_ITERS = 1e-2, 1e-2, (0.9, 0.999), 1e-8, 60


def token_stream(seed=0):
    """N_BATCHES fixed batches, cycled: multi-document rows (interior EOS), so the block-diagonal masks matter."""
    rng = np.random.default_rng(seed)
    out = []
    for b in range(N_BATCHES):
        tok = rng.integers(20, CFG["vocab_size"], size=(ROWS, T)).astype(np.int64)
        tok[:, 0] = 4
        for r in range(ROWS):
            for p in rng.choice(np.arange(6, T - 6), size=int(rng.integers(0, 3)), replace=False):
                tok[r, p] = R.EOS_TOKEN
        out.append(tok)
    return np.stack(out)


def mu_groups(named_params, lr, wd, width_mult):
    mats = [p for n, p in named_params if p.dim() == 2 and "wte" not in n and "lm_head" not in n]
    vecs = [p for n, p in named_params if not (p.dim() == 2 and "wte" not in n and "lm_head" not in n)]
    return [{"params": mats, "lr": lr / width_mult, "weight_decay": wd * width_mult}, {"params": vecs, "lr": lr, "weight_decay": wd}]


def run(ref_model, ref_train, flash, tokens, masks):
    cfg = R.RefConfig(**CFG, flash=flash)
    m = build_ref(ref_model, cfg, torch.bfloat16)
    dtype = torch.bfloat16
    opt = torch.optim.AdamW(mu_groups(list(m.named_parameters()), LR, WD, CFG["n_embd"] / R.MUP_BASE_WIDTH), lr=LR, betas=BETAS, eps=EPS,
                            weight_decay=WD)
    sched = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1.0, end_factor=0.0, total_iters=TOTAL_ITERS)
    losses, norms = [], []
    for i in range(STEPS):
        input_ids = torch.from_numpy(tokens[i % N_BATCHES])
        cum_loss = 0
        opt.zero_grad(set_to_none=True)
        mask = torch.as_tensor(masks[i], dtype=torch.bool)                                         # :273-275 (draw recorded)
        mask = mask & (input_ids != 1) & (input_ids != R.EOS_TOKEN)
        masked_ids = input_ids.masked_fill(mask, 2)
        n_accum = ROWS // MINI
        for j in range(n_accum):
            x = masked_ids[j * MINI:(j + 1) * MINI]
            y = input_ids[j * MINI:(j + 1) * MINI]
            attn_mask = torch.ones((MINI, T, T), dtype=dtype) * -1e9                                 # :289-292
            attn_mask = ref_train.create_attention_mask(attn_mask, y, padding=False)
            attn_mask = attn_mask.unsqueeze(1).expand(-1, CFG["n_head"], -1, -1)
            if not flash:
                attn_mask = attn_mask.contiguous()      # the manual path adds the mask in place (model.py:142)
            logits = m.forward(x.view(MINI, -1), attn_mask=attn_mask)
            loss = F.cross_entropy(logits.view(-1, logits.size(-1)), y.reshape(-1), reduction="none") / n_accum   # :301
            loss *= mask[j * MINI:(j + 1) * MINI].view(-1).float()                                  # :304
            loss = loss.sum() / mask[j * MINI:(j + 1) * MINI].view(-1).sum()                          # :305
            loss.backward()
            cum_loss += loss.item()
        norms.append(float(torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)))                      # :316
        opt.step()
        sched.step()
        losses.append(cum_loss)
    final = {k: p.detach().float().flatten()[::7].numpy().copy() for k, p in m.named_parameters()}
    return np.array(losses, dtype=np.float64), np.array(norms, dtype=np.float64), final


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    ref_model, ref_train = _import_reference()
    tokens = token_stream()
    np.random.seed(77)
    masks = np.stack([np.random.binomial(1, 0.15, (ROWS, T)) for _ in range(STEPS)]).astype(bool)     # :273
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        la, na, fa = run(ref_model, ref_train, True, tokens, masks)
        lb, nb, fb = run(ref_model, ref_train, False, tokens, masks)
    out = {"tokens": tokens, "mlm_draw": masks, "losses_flash": la, "losses_manual": lb, "grad_norms_flash": na, "grad_norms_manual": nb,
           "cfg": np.array([CFG["block_size"], CFG["vocab_size"], CFG["n_layer"], CFG["n_head"], CFG["n_embd"], 1], dtype=np.int64),
           "hyper": np.array([LR, WD, BETAS[0], BETAS[1], EPS, TOTAL_ITERS, ROWS, MINI, T, STEPS, N_BATCHES], dtype=np.float64),
           "param_stride": np.int64(7)}
    for k, v in fa.items():
        out["final_flash/" + k] = v
        out["final_manual/" + k] = fb[k]
    np.savez_compressed(os.path.join(OUT, "trajectory_tiny_bf16.npz"), **out)
    print("losses (flash): ", np.round(la, 4))
    print("losses (manual):", np.round(lb, 4))
    print("max |flash - manual| per step:", np.abs(la - lb).max(), " grad norms:", np.round(na[:4], 3), np.round(nb[:4], 3))
    d = max(np.abs(fa[k] - fb[k]).max() for k in fa)
    print("max final-parameter difference between the two reference runs:", d)


if __name__ == "__main__":
    main()
