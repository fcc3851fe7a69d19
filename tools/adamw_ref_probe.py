"""Diagnostic: FusedAdamW(rounding="reference") vs torch.optim.AdamW on bf16 CPU tensors, identical-element fraction per step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnibiote_amd import train_encoder as TE
BF = torch.bfloat16
shapes = [(256, 128), (1024,), (64, 512), (8,)]
steps = 6
gen = torch.Generator().manual_seed(5)
p0 = [torch.randn(s, generator=gen).to(BF) for s in shapes]
grads = [[(torch.randn(s, generator=gen) * (0.3 if t % 3 else 3.0)).to(BF) for s in shapes] for t in range(steps)]
lr, wd, betas, eps = 3e-3, 1e-2, (0.9, 0.999), 1e-8
cpu = [torch.nn.Parameter(x.clone()) for x in p0]
gpu = [torch.nn.Parameter(x.clone().cuda()) for x in p0]
groups = lambda ps: [{"params": ps[:2], "lr": lr / 4, "weight_decay": wd * 4}, {"params": ps[2:], "lr": lr, "weight_decay": wd}]
ref = torch.optim.AdamW(groups(cpu), lr=lr, betas=betas, eps=eps, weight_decay=wd)
fused = TE.FusedAdamW(groups(gpu), lr=lr, betas=betas, eps=eps, weight_decay=wd, rounding="reference")
for t in range(steps):
    for q, r, gq in zip(cpu, gpu, grads[t]):
        q.grad = gq.clone(); r.grad = gq.clone().cuda()
    tn = torch.nn.utils.clip_grad_norm_(cpu, 1.0)
    gc = [q.grad.clone() for q in cpu]
    ref.step()
    fused.step(max_norm=1.0)
    tot_g = torch.linalg.vector_norm(fused._norm_sq.sqrt().to(BF))
    print(f"step {t}: torch total_norm {tn.item()} fused {tot_g.item()}  per-tensor {[round(float(x),3) for x in fused._norm_sq.sqrt().to(BF).float().cpu()]} vs {[round(float(torch.linalg.vector_norm(g).float()),3) for g in grads[t]]}")
    for i, (q, r) in enumerate(zip(cpu, gpu)):
        f = lambda a, b: float((a.float() == b.float().cpu()).float().mean())
        print(f"   tensor {i}: p {f(q.data, r.data):.5f} m {f(ref.state[q]['exp_avg'], fused.state[r]['exp_avg']):.5f} v {f(ref.state[q]['exp_avg_sq'], fused.state[r]['exp_avg_sq']):.5f}")
