"""How long the HOST needs to enqueue one optimizer step (no synchronisation inside the loop) against the GPU's time for it:
the step is GPU-bound only while the first stays well below the second."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnibiote_amd import train_encoder as TE, tune
import contextlib, io
h = TE.parse_args([]); h.batch_size, h.mini_batch_size, h.n_layer, h.n_embd, h.n_head, h.ctx_len, h.dropout = 128, 8, 8, 1024, 8, 1024, 0.0
dev = torch.device("cuda", 0)
with contextlib.redirect_stdout(io.StringIO()):
    m = TE.build_model(h, dev)
tune.tune_model_shapes(8 * 1024, 1024, 2 ** 16, device=dev)
opt, sched = TE.build_optimizer(m, h, 1000)
step = TE.TrainStep(m, opt, sched, mini_batch_size=8, n_head=8, pipeline_streams=2)
rng = np.random.default_rng(0)
host = TE.synthetic_rows(128, 1024, 2 ** 16, rng)
ids = torch.from_numpy(host).to(dev)
for _ in range(2):
    step(ids, input_ids_host=host)
torch.cuda.synchronize()
t0 = time.perf_counter(); hs = []
for _ in range(6):
    a = time.perf_counter(); step(ids, input_ids_host=host); hs.append(time.perf_counter() - a)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue per step: {np.mean(hs) * 1e3:.1f} ms (first {hs[0] * 1e3:.1f}, last {hs[-1] * 1e3:.1f}); wall per step incl. GPU: {t_all / 6 * 1e3:.1f} ms; host done after {t_enq * 1e3:.0f} ms of {t_all * 1e3:.0f} ms")
