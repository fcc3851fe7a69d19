"""bench.py's N > 1 control flow under test on the GPU box: `python bench.py --gpus 2` starts its two
ranks as a child `torch.distributed.run` (a fresh process tree — never a re-exec of a process that touched the GPU), every
rank goes through process-group set-up, the first-contact all-reduce, plan broadcast, DDP wrap, the timed steps (no_sync +
bucketed all-reduce on the last micro-batch, two-stream pipeline), the every-rank profiled step, the per-rank variants and the
final barrier, and rank 0 prints ONE JSON line.  On a one-GPU box both ranks share cuda:0 over gloo
(OBTE_BENCH_REHEARSE=1: RCCL refuses two ranks on one device); with two or more GPUs the same test runs over RCCL.
The numbers mean nothing here — the flow, the line's fields and the exit codes are what is checked.
Reference counterpart: training/train_encoder.py:105-118,185,284-311 under torchrun."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(args, env_extra, tmp_path, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out, err = open(tmp_path / "bench.out", "w+"), open(tmp_path / "bench.err", "w+")
    try:
        rc = subprocess.call([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=out, stderr=err, cwd=ROOT, timeout=timeout)
    finally:
        out.seek(0); err.seek(0)
        so, se = out.read(), err.read()
        out.close(); err.close()
    return rc, so, se


@pytest.mark.timeout(900)
def test_bench_two_ranks_prints_one_line(tmp_path):
    rccl = torch.cuda.device_count() >= 2
    # the tiny config (BASELINE configs[0]: 2L / 128d / 2h, ctx 128) keeps the gloo rehearsal short — 34 MB of gradients cross the
    # host per step instead of 470 MB; the control flow under test does not depend on the model size
    rc, so, se = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--rows_per_rank", "24", "--no_cpu_baseline", "--config", "tiny"],
                            {} if rccl else {"OBTE_BENCH_REHEARSE": "1"}, tmp_path)
    assert rc == 0, se[-4000:]
    lines = [l for l in so.splitlines() if l.startswith("{")]
    assert len(lines) == 1, so[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["scaling"] == "weak" and line["unit"] == "tokens/s"
    assert line["value"] > 0 and line["ms_per_step"] > 0
    col = line["config"]["collectives"]
    assert col["world_size"] == 2 and col["ranks_in_first_all_reduce"] == 2
    assert (col["rccl_version"] is not None) == rccl and ("RCCL" in col["backend"]) == rccl
    assert line["config"]["global_batch_rows"] == 48 and line["config"]["parallelism"] == "dp2"
    assert 0.0 < line["final_loss"] < 20.0
    assert "mfma_fraction_whole_step" not in line and line["mfma_fraction_whole_step_executed"] >= 0 and line["flops_per_token_executed"] > 0
    assert len(line["gemm_plans"]["table"]) >= 15 and len(line["gemm_plans"]["sha16"]) == 16
    assert line["roofline"]["bound"] == "mfma" and line["roofline"]["frac"] > 0
    assert set(line["variants"]) >= {"dense_logits_forward", "masked_readout_full_last_block", "dense_dlogits_full_backward", "dropout_0.1", "dense_mask_calling_convention"}
    assert ("REHEARSAL" in line["data"]) == (not rccl)
    assert "first all-reduce ok" in se


@pytest.mark.timeout(300)
def test_bench_fails_loudly_when_a_rank_never_arrives(tmp_path):
    """A world of two with one rank missing: the present rank must give up with a non-zero exit code and a message, not hang
    (the set-up watchdog; 8 GPUs of a driver-launched job would otherwise sit in init_process_group until killed)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               OBTE_BENCH_INIT_TIMEOUT_S="20", OBTE_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no_cpu_baseline"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert "FATAL" in p.stderr or "Timed out" in p.stderr or "timeout" in p.stderr.lower(), p.stderr[-2000:]
