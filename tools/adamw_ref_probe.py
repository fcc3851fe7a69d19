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

# ---- one more step with full diagnostics on tensor 0
t = steps
gq = (torch.randn(shapes[0], generator=gen) * 0.3).to(BF)
m_old = ref.state[cpu[0]]["exp_avg"].clone(); m_old_g = fused.state[gpu[0]]["exp_avg"].clone()
print("m_old identical before:", float((m_old.float() == m_old_g.float().cpu()).float().mean()))
fused.state[gpu[0]]["exp_avg"].copy_(m_old.cuda()); fused.state[gpu[0]]["exp_avg_sq"].copy_(ref.state[cpu[0]]["exp_avg_sq"].cuda()); gpu[0].data.copy_(cpu[0].data.cuda())
for i, (q, r) in enumerate(zip(cpu, gpu)):
    q.grad = (gq.clone() if i == 0 else torch.zeros_like(q)); r.grad = q.grad.clone().cuda()
tn = torch.nn.utils.clip_grad_norm_(cpu, 1.0)
g_clipped = cpu[0].grad.clone()
ref.step(); fused.step(max_norm=1.0)
m_c, m_g = ref.state[cpu[0]]["exp_avg"].float(), fused.state[gpu[0]]["exp_avg"].float().cpu()
bad = (m_c != m_g).nonzero()
print("after resync, one step: m identical", float((m_c == m_g).float().mean()), "n bad", len(bad))
emu = (m_old.float() + torch.tensor(0.1, dtype=torch.float32) * (g_clipped.float() - m_old.float())).to(BF).float()
print("emu vs cpu", float((emu == m_c).float().mean()), "emu vs gpu", float((emu == m_g).float().mean()))
for ix in bad[:6]:
    ix = tuple(ix.tolist())
    print(ix, "m_old", m_old[ix].item(), "g_clipped(cpu)", g_clipped[ix].item(), "g", gq[ix].item(), "cpu", m_c[ix].item(), "gpu", m_g[ix].item(),
          "exact", m_old[ix].double().item() + 0.1 * (g_clipped[ix].double().item() - m_old[ix].double().item()))
