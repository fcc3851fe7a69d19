"""What the HIP path measures on the tiny golden fixtures (tests/test_hip_model.py forward/backward parity tests): the worst
statistic per case, so that the bars in those tests can be set to what is measured plus margin.   python tools/measure_bars.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_model as TM   # noqa: E402
from omnibiote_amd import ops   # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
DEV = "cuda"
for name, rope_mode in TM.CASES:
    for mk in ("ranges", "dense"):
        g = TM.load(G, name)
        m = TM.build(g, rope_mode)
        H = int(g["cfg"][3])
        idx = torch.from_numpy(g["masked_ids"]).to(DEV)
        mask = TM.masks_for(g, mk, H)
        e = TM.stats(m(idx, attn_mask=mask, return_embeddings=True), g["emb"])
        logits = m(idx, attn_mask=mask)
        l = TM.stats(logits, g["logits"])
        loss, _ = ops.masked_ce(logits, torch.from_numpy(g["tokens"]).to(DEV), torch.from_numpy(g["mlm_mask"]).to(DEV), int(g["n_accum"]))
        print(f"fwd {name:18s} {mk:6s} emb max {e[0]:.4f} mean {e[1]:.5f} | logits max {l[0]:.4f} mean {l[1]:.5f} | dloss {abs(loss.item() - float(g['loss'])):.5f}")
for name, rope_mode in [("tiny_fp32_mask", "complex"), ("wide_fp32_mask", "complex"), ("tiny_bf16_mask", "cos_only")]:
    g = TM.load(G, name)
    m = TM.build(g, rope_mode)
    idx = torch.from_numpy(g["masked_ids"]).to(DEV)
    logits = m(idx, attn_mask=TM.masks_for(g, "ranges", int(g["cfg"][3])))
    loss, dlogits = ops.masked_ce(logits, torch.from_numpy(g["tokens"]).to(DEV), torch.from_numpy(g["mlm_mask"]).to(DEV), int(g["n_accum"]))
    logits.backward(dlogits)
    stride = int(g["grad_stride"])
    worst = (0.0, 1.0, "", "")
    for k, p in m.named_parameters():
        want = torch.from_numpy(g["grad_sample/" + k])
        got = p.grad.float().flatten()[::stride].cpu()
        denom = want.norm().item() + 1e-12
        rel = (got - want).norm().item() / denom
        cos = torch.dot(got, want).item() / (got.norm().item() * denom + 1e-30)
        if rel > worst[0]:
            worst = (rel, worst[1], k, worst[3])
        if cos < worst[1]:
            worst = (worst[0], cos, worst[2], k)
    print(f"bwd {name:18s} worst rel {worst[0]:.4f} ({worst[2]})  worst cos {worst[1]:.5f} ({worst[3]})")
