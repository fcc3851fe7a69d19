"""Gradient exchange of the data-parallel step (training/train_encoder.py:105-109,185: DDP over NCCL) for MI355X nodes.

The reference relies on DDP's default bucketed all-reduce.  On an 8-GPU xGMI mesh every GPU has a point-to-point link to
each of the other seven, and a ring all-reduce moves every byte over ONE of them per step: a 470-MB gradient (small config)
is bound by a single link (~10.7 ms), while the direct algorithm — every rank sends shard j of the bucket
straight to rank j (all seven links at once), each rank sums the W shards it received, then every rank broadcasts its reduced
shard to all the others (again all links) — moves 1/W of the bytes per link (~1.5 ms).  ``AllLinksHook`` is that algorithm as
a DDP communication hook: one ``all_to_all_single`` + one ``all_gather_into_tensor`` per bucket (RCCL issues both as
concurrent point-to-point transfers), with the reduction done HERE, in fp32, in rank order, and rounded to the gradients'
dtype once — so the result does not depend on the collective library's reduction tree, is identical on every rank, and is
bitwise reproducible.  (A ring all-reduce on bf16 rounds after every hop; at world size 2 the two agree bit for bit, beyond
that this form is the more accurate one — tests/test_distributed_cpu.py.)

``TimedHook`` wraps either exchange (the default all-reduce or the all-links form) and records, per bucket, when it became
ready on the backward's stream and when its exchange finished, so that the first multi-GPU run can say how much of the
communication the last pass's backward covered (bench.py: ``config.collectives``).
"""
from typing import List, Optional

import torch
import torch.distributed as dist


def _shard_elems(n: int, world: int, align: int = 64) -> int:
    """Elements per rank: ceil(n / world), rounded up to `align` so that every shard starts on a 128-byte line (bf16)."""
    s = (n + world - 1) // world
    return (s + align - 1) // align * align


def reduce_shards_fixed_order(recv: torch.Tensor, world: int, out_dtype: torch.dtype) -> torch.Tensor:
    """recv: [world, shard] — row r is rank r's contribution to this rank's shard.  fp32 sum in rank order, mean, ONE rounding."""
    acc = recv[0].float()
    for r in range(1, world):
        acc = acc + recv[r].float()
    return (acc / world).to(out_dtype)


class AllLinksHook:
    """DDP comm hook: mean of the bucket over the group by all_to_all (reduce-scatter, reduced here) + all_gather.

        model = DistributedDataParallel(...); model.register_comm_hook(None, AllLinksHook(group))

    ``via_host``: stage the exchange through host memory — ONLY for rehearsing the hook with several ranks on one GPU over
    gloo, whose device-tensor support stops at broadcast / all-reduce (set automatically for that combination)."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None, via_host: Optional[bool] = None):
        self.group = group
        self.via_host = via_host
        self._bufs = {}

    def _scratch(self, key, numel, like: torch.Tensor) -> torch.Tensor:
        t = self._bufs.get(key)
        if t is None or t.numel() < numel or t.device != like.device or t.dtype != like.dtype:
            t = torch.empty(numel, dtype=like.dtype, device=like.device)
            self._bufs[key] = t
        return t[:numel]

    def __call__(self, state, bucket: dist.GradBucket) -> torch.futures.Future[torch.Tensor]:
        group = self.group if self.group is not None else dist.group.WORLD
        world = dist.get_world_size(group)
        buf = bucket.buffer()
        n = buf.numel()
        if world == 1:
            fut = torch.futures.Future()
            fut.set_result(buf)
            return fut
        via_host = self.via_host
        if via_host is None:
            via_host = buf.is_cuda and dist.get_backend(group) == "gloo"
        shard = _shard_elems(n, world)
        idx = bucket.index()
        if via_host:   # rehearsal path (see the class docstring): the same arithmetic on host copies
            send = torch.zeros(world * shard, dtype=buf.dtype)
            send[:n].copy_(buf)
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send, group=group)
            mine = reduce_shards_fixed_order(recv.view(world, shard), world, buf.dtype)
            out = torch.empty_like(send)
            dist.all_gather_into_tensor(out, mine, group=group)
            buf.copy_(out[:n])
            fut = torch.futures.Future()
            fut.set_result(buf)
            return fut
        send = self._scratch(("send", idx), world * shard, buf)
        send[:n].copy_(buf)
        if world * shard > n:
            send[n:].zero_()
        recv = self._scratch(("recv", idx), world * shard, buf)
        out = self._scratch(("out", idx), world * shard, buf)
        work = dist.all_to_all_single(recv, send, group=group, async_op=True)

        def reduce_and_gather(fut):
            # runs once the exchange has completed (for RCCL: on a stream ordered behind it); the all-gather is enqueued from
            # here and awaited by the stream, not by the host
            mine = reduce_shards_fixed_order(recv.view(world, shard), world, buf.dtype)
            dist.all_gather_into_tensor(out, mine, group=group)
            buf.copy_(out[:n])
            return buf

        return work.get_future().then(reduce_and_gather)


def allreduce_mean_hook(group: Optional[dist.ProcessGroup] = None):
    """DDP's default exchange as an explicit hook (so that it can be wrapped by TimedHook): divide, all-reduce."""
    def hook(state, bucket: dist.GradBucket) -> torch.futures.Future[torch.Tensor]:
        g = group if group is not None else dist.group.WORLD
        buf = bucket.buffer()
        buf.div_(dist.get_world_size(g))
        return dist.all_reduce(buf, group=g, async_op=True).get_future().then(lambda f: f.value()[0])
    return hook


class TimedHook:
    """Wraps a comm hook and records, per bucket and per optimizer step, two device events: `ready` on the stream the hook was called
    on (the backward's: the bucket's gradients are complete) and `done` where the exchange's future resolves.  Nothing is
    synchronised here; ``collect()`` (call it after the step has been synchronised with) turns the events into milliseconds."""

    def __init__(self, inner):
        self.inner = inner
        self.records: List[dict] = []

    def __call__(self, state, bucket: dist.GradBucket) -> torch.futures.Future[torch.Tensor]:
        buf = bucket.buffer()
        if not buf.is_cuda:
            return self.inner(state, bucket)
        rec = {"index": bucket.index(), "bytes": buf.numel() * buf.element_size(), "last": bucket.is_last(),
               "ready": torch.cuda.Event(enable_timing=True), "done": torch.cuda.Event(enable_timing=True)}
        rec["ready"].record()
        self.records.append(rec)

        def stamp(fut):
            rec["done"].record()
            v = fut.value()
            return v[0] if isinstance(v, (list, tuple)) else v

        return self.inner(state, bucket).then(stamp)

    def collect(self, backward_end: Optional[torch.cuda.Event] = None) -> dict:
        """Per-bucket milliseconds from `ready` to `done`, the span from the first `ready` to the last `done`, and — given the
        event recorded when the step's last backward finished — how long the exchange ran past it (the exposed part)."""
        recs, self.records = self.records, []
        if not recs:
            return {}
        out = {"buckets": [{"index": r["index"], "MB": round(r["bytes"] / 1e6, 1), "ms_ready_to_done": round(r["ready"].elapsed_time(r["done"]), 3)}
                           for r in recs]}
        first = recs[0]["ready"]
        out["ms_first_ready_to_last_done"] = round(max(first.elapsed_time(r["done"]) for r in recs), 3)
        out["MB_total"] = round(sum(r["bytes"] for r in recs) / 1e6, 1)
        if backward_end is not None:
            out["ms_exposed_after_backward"] = round(max(0.0, max(backward_end.elapsed_time(r["done"]) for r in recs)), 3)
        return out


def as_ddp_hook(obj, name: str):
    """DDP's register_comm_hook wants a plain function (it reads __name__ / __qualname__ and the annotations): wrap a hook object."""
    def hook(state, bucket: dist.GradBucket) -> torch.futures.Future[torch.Tensor]:
        return obj(state, bucket)
    hook.__name__ = hook.__qualname__ = name
    return hook
