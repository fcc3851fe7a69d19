"""Token-shard reader and sequence packer: the counterpart of the reference's ``training/loader.py``.

Same names, arguments and output contract as the reference's live functions — ``line_reader`` (loader.py:25-59),
``get_sequence`` (:118-163), ``get_batch`` (:165-181), ``data_loader_parallel`` (:8-23) — and, given the same global NumPy
seed, the same tokens in the same order (tests/test_loader.py checks this against vectors recorded from the reference,
bit for bit).  That includes its behaviours one would not invent:
  * ``np.random.shuffle(filenames)`` shuffles the caller's list in place, every pass;
  * files are read in groups of 10 and documents are shuffled within a group only;
  * the empty tail after a shard's final EOS takes part in the shuffle (it consumes RNG draws) and is then skipped;
  * when a packed row is exactly full, the document just pulled from the reader is discarded;
  * in truncation mode the overflowing document's remainder is discarded; in padding mode the whole overflowing document is
    (and a document longer than ctx_len can never be emitted).
What is different is mechanics: rows are packed into preallocated int32 arrays (no per-token Python lists), batches are
stacked arrays, and the loader thread stages batches in pinned memory and copies them on its own HIP stream so the copy
overlaps the training step.
"""
from __future__ import annotations

import queue
from typing import Iterable, Iterator, List, Sequence

import numpy as np
import torch

EOS_TOKEN = 3
MASK_TOKEN = 2
PAD_TOKEN = 1

FILES_PER_GROUP = 10   # loader.py:34: shards loaded (and shuffled) together


def line_reader(filenames, banned_tokens) -> Iterator[np.ndarray]:
    """Endless stream of documents (int32 arrays, final EOS included, banned ids removed) from ``.npy`` token shards."""
    banned = np.asarray(list(banned_tokens), dtype=np.int64)
    while True:
        np.random.shuffle(filenames)                                   # in place, like the reference
        for g in range(0, len(filenames), FILES_PER_GROUP):
            block = np.concatenate([np.load(f) for f in filenames[g:g + FILES_PER_GROUP]])
            cut = np.flatnonzero(block == EOS_TOKEN) + 1               # one past each EOS
            starts = np.concatenate(([0], cut))
            ends = np.concatenate((cut, [len(block)]))                 # the last piece may be empty: it still joins the shuffle
            order = np.arange(len(starts))
            np.random.shuffle(order)
            for i in order:
                doc = block[starts[i]:ends[i]]
                if len(doc) == 0:
                    continue
                if len(banned) == 1:
                    doc = doc[doc != banned[0]]
                elif len(banned) > 1:
                    doc = doc[~np.isin(doc, banned)]
                yield doc.astype(np.int32, copy=False)


def get_sequence(reader, ctx_len: int, USE_PADDING: bool = False) -> Iterator[np.ndarray]:
    """Pack documents into rows of exactly ``ctx_len`` tokens (int32 arrays)."""
    row = np.empty(ctx_len, dtype=np.int32)
    fill = 0
    while True:
        doc = next(reader)
        if fill == ctx_len:              # exactly full: emit; the document just pulled is dropped (reference behaviour)
            yield row.copy()
            fill = 0
            continue
        if fill + len(doc) > ctx_len:
            if USE_PADDING:
                if fill == 0:            # a document longer than a row: skipped
                    continue
                row[fill:] = PAD_TOKEN
            else:
                row[fill:] = doc[:ctx_len - fill]
            yield row.copy()
            fill = 0
            continue
        row[fill:fill + len(doc)] = doc
        fill += len(doc)


def get_batch(generators: Sequence[Iterator[np.ndarray]], train_ints: Sequence[int], return_pt: bool = False, device="cpu"):
    """``train_ints[i]`` rows from ``generators[i]``, rows shuffled; int64 (rows, ctx_len)."""
    while True:
        rows: List[np.ndarray] = []
        for gen, n in zip(generators, train_ints):
            for _ in range(n):
                rows.append(np.asarray(next(gen)))
        np.random.shuffle(rows)          # a list of the same length as the reference's: same permutation
        batch = np.stack(rows).astype(np.int64)
        yield torch.from_numpy(batch).to(device) if return_pt else batch


def data_loader_parallel(batch_queue: "queue.Queue", batch_generator, device, stop=None) -> None:
    """Thread target: move batches to ``device`` ahead of the training loop.  On a GPU the batch is staged in pinned memory
    and copied on a dedicated stream; it is published only after that copy has completed, so the consumer may use it on
    any stream.  ``stop`` (a threading.Event, optional): set it and drain the queue to make the thread return, so that it
    can be joined before the process group and the HIP context are torn down."""
    import queue as _queue
    dev = torch.device(device)
    if dev.type == "cuda":
        torch.cuda.set_device(dev)   # the current device is per thread and defaults to 0: without this every rank would
                                     # create a context (and pin memory) on GPU 0
    copy_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
    while stop is None or not stop.is_set():
        try:
            data = next(batch_generator)
        except StopIteration:
            break
        host = data
        if copy_stream is not None:
            pinned = data.pin_memory()
            with torch.cuda.stream(copy_stream):
                data = pinned.to(dev, non_blocking=True)
            copy_stream.synchronize()
        else:
            data = data.to(dev)
        data._obte_host_copy = host.numpy()   # the batch as it was on the host: the trainer draws its MLM mask there (no device round trip)
        while True:   # bounded queue (train_encoder.py:141): wait for room, but notice a stop request
            try:
                batch_queue.put(data, timeout=0.2)
                break
            except _queue.Full:
                if stop is not None and stop.is_set():
                    return


def batch_split(batch_size: int, proportions: Sequence[float]) -> List[int]:
    """Rows per dataset for one batch (train_encoder.py:120-124): floor of each share, remainder to the last dataset."""
    split = [int(x * batch_size) for x in proportions]
    split[-1] += batch_size - sum(split)
    return split
