"""Not collected by pytest: a longer (40-step) version of test_loss_curve_tracks_oracle_step_for_step, printed rather than
asserted — HIP path (bf16, fused optimizer, two-stream micro-batches) against the CPU oracle with an fp32 master copy.
    python tests/long_curve_check.py     (on a GPU box)"""
import sys, os, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import omnibiote_ref as R
from omnibiote_amd import train_encoder as TE
from omnibiote_amd.mup_compat import mu_param_groups, set_base_shapes
from omnibiote_amd.model import OmniBioTA, OmniBioTAConfig
BF = torch.bfloat16; DEV = "cuda"
C, H, Lyr, V, T, rows, mini, steps = 256, 2, 2, 1024, 128, 16, 4, 40
cfg = R.RefConfig(block_size=T, vocab_size=V, n_layer=Lyr, n_head=H, n_embd=C)
w = R.hash_weights(cfg)
c = OmniBioTAConfig(); c.block_size, c.vocab_size, c.n_layer, c.n_head, c.n_embd, c.dropout, c.flash = T, V, Lyr, H, C, 0.0, True
m = OmniBioTA(c)
cb = OmniBioTAConfig(); cb.block_size, cb.vocab_size, cb.n_layer, cb.dropout, cb.flash = T, V, Lyr, 0.0, True
cb.n_embd, cb.n_head = 24, 3; base = OmniBioTA(cb); cb.n_embd, cb.n_head = 48, 12; delta = OmniBioTA(cb)
set_base_shapes(m, base, delta=delta, rescale_params=False)
m.load_state_dict(w, strict=False)
with warnings.catch_warnings():
    warnings.simplefilter("ignore"); m.to(BF)
m.to(DEV)
lr, wd = 3e-3, 1e-2
opt = TE.FusedAdamW(mu_param_groups(list(m.parameters()), lr, wd), lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
step = TE.TrainStep(m, opt, None, mini_batch_size=mini, n_head=H, pipeline_streams=2)
enc = R.OracleEncoder(cfg, {k: v.to(BF).float() for k, v in w.items()})
enc.rope = R.cast_rope_table(R.rope_table(C // H, T), BF)
named = enc.named_weights()
mats = [p for n, p in named.items() if p.dim() == 2 and "wte" not in n and "lm_head" not in n]
vecs = [p for n, p in named.items() if not (p.dim() == 2 and "wte" not in n and "lm_head" not in n)]
wm = C / 24
ref_opt = torch.optim.AdamW([{"params": mats, "lr": lr / wm, "weight_decay": wd * wm}, {"params": vecs, "lr": lr, "weight_decay": wd}], lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
ref_step = TE.TrainStep(enc, ref_opt, None, mini_batch_size=mini, n_head=H, loss_impl="torch", mask_impl="dense")
rng = np.random.default_rng(0)
ids = torch.from_numpy(TE.synthetic_rows(rows, T, V, rng, single_document=False)); ids[:, T // 2] = R.EOS_TOKEN
for s in range(steps):
    np.random.seed(100 + s % 4); a = step(ids.to(DEV))["loss"].item()
    np.random.seed(100 + s % 4); b = ref_step(ids)["loss"].item()
    if s % 4 == 0 or s == steps - 1: print(f"step {s:2d}  hip {a:.4f}  oracle(fp32 master) {b:.4f}  rel {abs(a-b)/b:.4f}", flush=True)
