"""Golden vectors for the data loader / sequence packer.  TEST INFRASTRUCTURE ONLY — build container.

Imports the reference's own ``training/loader.py`` (numpy + torch only; nothing copied), runs its live functions
(line_reader :25-59, get_sequence :118-163, get_batch :165-181) over small synthetic token shards with the global NumPy RNG
seeded, and records what they yield.  The shards are a closed-form function of a seed (``make_shards``), so the fixture
holds only outputs; tests/test_loader.py rebuilds the shards and requires the product loader to reproduce every token.

Usage: python oracle/gen_golden_loader.py   (writes tests/golden/loader.npz)
"""
import os
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden", "loader.npz")


def make_shards(directory, n_files, seed, vocab=200, banned=199):
    """n_files .npy int32 shards of concatenated documents ``tag, body..., EOS(3)``; some bodies contain the banned id;
    includes an empty document (two EOS in a row) and documents longer than any ctx_len used."""
    rng = np.random.default_rng(seed)
    names = []
    for f in range(n_files):
        toks = []
        for d in range(int(rng.integers(6, 14))):
            n = int(rng.choice([0, 3, 7, 20, 45, 130]))
            body = rng.integers(20, vocab, size=n)
            if n > 4:
                body[int(rng.integers(0, n))] = banned
            toks.extend([4 if d % 3 else 18] if n else [])
            toks.extend(body.tolist())
            toks.append(3)
        path = os.path.join(directory, f"shard_{seed}_{f:03d}.npy")
        np.save(path, np.asarray(toks, dtype=np.int32))
        names.append(path)
    return names


def take(gen, n):
    return [next(gen) for _ in range(n)]


def main():
    sys.path.insert(0, "/root/reference/training")
    import loader as ref  # the reference's own module
    out = {}
    with tempfile.TemporaryDirectory() as d:
        files_a = make_shards(d, 13, seed=1)   # > 10 files: exercises the chunk-of-10 split
        files_b = make_shards(d, 4, seed=2)
        # 1) line_reader: the first 60 documents (two passes over the 13 files happen within them or not, either way)
        np.random.seed(11)
        docs = take(ref.line_reader(list(files_a), banned_tokens=[199]), 60)
        out["docs_len"] = np.array([len(x) for x in docs], dtype=np.int64)
        out["docs_cat"] = np.concatenate(docs).astype(np.int64)
        np.random.seed(12)
        docs2 = take(ref.line_reader(list(files_b), banned_tokens=[199, 198]), 25)   # several banned ids: np.isin branch
        out["docs2_len"] = np.array([len(x) for x in docs2], dtype=np.int64)
        out["docs2_cat"] = np.concatenate(docs2).astype(np.int64)
        # 2) get_sequence, truncation and padding modes, ctx_len that hits the "exactly full" quirk too
        for mode, pad in (("trunc", False), ("pad", True)):
            for ctx in (32, 50):
                np.random.seed(13)
                seqs = take(ref.get_sequence(ref.line_reader(list(files_a), banned_tokens=[199]), ctx, pad), 30)
                out[f"seq_{mode}_{ctx}"] = np.asarray(seqs, dtype=np.int64)
        # 3) get_batch: two generators mixed 3:1, shuffled rows
        np.random.seed(14)
        gens = [ref.get_sequence(ref.line_reader(list(files_a), banned_tokens=[199]), 40, False),
                ref.get_sequence(ref.line_reader(list(files_b), banned_tokens=[199]), 40, False)]
        batches = take(ref.get_batch(gens, [3, 1], return_pt=True), 5)
        out["batches"] = np.stack([b.numpy() for b in batches])
        out["consts"] = np.array([ref.EOS_TOKEN, ref.MASK_TOKEN, ref.PAD_TOKEN], dtype=np.int64)
    np.savez_compressed(OUT, **out)
    print("loader golden:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
