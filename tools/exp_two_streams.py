"""Experiment: do two independent training replicas on two HIP streams of ONE GPU finish faster than back to back?
(Upper bound for running two micro-batches concurrently.)  python tools/exp_two_streams.py"""
import contextlib, io, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from omnibiote_amd import train_encoder as TE, tune, _lib

class A: pass
a = A(); a.rows_per_rank = 128; a.mini_batch_size = 8; a.dropout = 0.0
cfg = bench.CONFIGS["small"]
dev = torch.device("cuda", 0)
_lib.lib()
h = bench.harness_args(cfg, a, 1)
tune.tune_model_shapes(8 * 1024, 1024, 2 ** 16, device=dev)
reps = []
NREP = int(os.environ.get('NREP', '2'))
for r in range(NREP):
    torch.manual_seed(1234 + r)
    with contextlib.redirect_stdout(io.StringIO()):
        m = TE.build_model(h, dev)
    opt, sched = TE.build_optimizer(m, h, 1000)
    reps.append(TE.TrainStep(m, opt, sched, mini_batch_size=8, n_head=8, pipeline_streams=1))
rng = np.random.default_rng(0)
batch = torch.from_numpy(TE.synthetic_rows(128, 1024, 2 ** 16, rng, single_document=True)).to(dev)
streams = [torch.cuda.Stream() for _ in range(NREP)]
def run(concurrent, steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        for r in range(NREP):
            s = streams[r] if concurrent else streams[0]
            with torch.cuda.stream(s):
                reps[r](batch)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps
run(False, 1); run(True, 1)
for name, c in (("sequential", False), ("concurrent", True), ("sequential", False), ("concurrent", True)):
    t = run(c, 3)
    print(f"{name}: {t * 1e3:.1f} ms per {NREP} steps -> {NREP * 128 * 1024 / t:.0f} tokens/s", flush=True)
