"""The shard reader / sequence packer against vectors recorded from the reference's own loader
(oracle/gen_golden_loader.py -> tests/golden/loader.npz).  Integer work: bit-exact.  CPU only."""
import os
import queue
import sys
import threading

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from gen_golden_loader import make_shards  # the closed-form shard recipe (no reference import happens at import time)

from omnibiote_amd import loader as LD


@pytest.fixture()
def shards(tmp_path):
    return make_shards(str(tmp_path), 13, seed=1), make_shards(str(tmp_path), 4, seed=2)


def take(gen, n):
    return [next(gen) for _ in range(n)]


def test_constants(golden_dir):
    g = np.load(os.path.join(golden_dir, "loader.npz"))
    assert [LD.EOS_TOKEN, LD.MASK_TOKEN, LD.PAD_TOKEN] == g["consts"].tolist()


def test_line_reader_matches_reference(golden_dir, shards):
    g = np.load(os.path.join(golden_dir, "loader.npz"))
    files_a, files_b = shards
    np.random.seed(11)
    mine = list(files_a)
    docs = take(LD.line_reader(mine, banned_tokens=[199]), 60)
    assert [len(d) for d in docs] == g["docs_len"].tolist()
    np.testing.assert_array_equal(np.concatenate(docs), g["docs_cat"])
    assert all(d.dtype == np.int32 for d in docs) and not any((d == 199).any() for d in docs)
    assert mine != list(files_a) and sorted(mine) == sorted(files_a)       # shuffled in place, like the reference
    np.random.seed(12)
    docs2 = take(LD.line_reader(list(files_b), banned_tokens=[199, 198]), 25)
    assert [len(d) for d in docs2] == g["docs2_len"].tolist()
    np.testing.assert_array_equal(np.concatenate(docs2), g["docs2_cat"])


@pytest.mark.parametrize("mode,pad", [("trunc", False), ("pad", True)])
@pytest.mark.parametrize("ctx", [32, 50])
def test_get_sequence_matches_reference(golden_dir, shards, mode, pad, ctx):
    g = np.load(os.path.join(golden_dir, "loader.npz"))
    np.random.seed(13)
    seqs = take(LD.get_sequence(LD.line_reader(list(shards[0]), banned_tokens=[199]), ctx, pad), 30)
    np.testing.assert_array_equal(np.stack(seqs), g[f"seq_{mode}_{ctx}"])
    assert all(len(s) == ctx for s in seqs)
    if pad:
        assert any((s == LD.PAD_TOKEN).any() for s in seqs)


def test_get_batch_matches_reference_and_loader_thread(golden_dir, shards):
    g = np.load(os.path.join(golden_dir, "loader.npz"))
    files_a, files_b = shards

    def make():
        np.random.seed(14)
        gens = [LD.get_sequence(LD.line_reader(list(files_a), banned_tokens=[199]), 40, False),
                LD.get_sequence(LD.line_reader(list(files_b), banned_tokens=[199]), 40, False)]
        return LD.get_batch(gens, [3, 1], return_pt=True)

    batches = take(make(), 5)
    assert all(b.dtype == torch.int64 and tuple(b.shape) == (4, 40) for b in batches)
    np.testing.assert_array_equal(np.stack([b.numpy() for b in batches]), g["batches"])
    # the loader thread delivers the same batches in order through a bounded queue (train_encoder.py:140-142)
    q = queue.Queue(maxsize=2)

    def finite(gen, n):
        for _ in range(n):
            yield next(gen)

    t = threading.Thread(target=LD.data_loader_parallel, args=(q, finite(make(), 5), "cpu"))
    t.start()
    got = [q.get(timeout=30) for _ in range(5)]
    t.join(timeout=30)
    np.testing.assert_array_equal(np.stack([b.numpy() for b in got]), g["batches"])


def test_batch_split():
    assert LD.batch_split(128, [0.8, 0.2]) == [102, 26]       # train_encoder.py:120-124
    assert LD.batch_split(7, [0.5, 0.5]) == [3, 4]
    assert LD.batch_split(16, [1.0]) == [16]
