"""The N>1 path (difacto_amd/sharded.py) with world_size 2 over gloo on CPU: the
key-range partition, the three all_to_all_v exchanges and the sequential
source-rank-order updates must reproduce, bit for bit, a single store that
receives the same pushes in the same order."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HYPER = dict(l1=0.02, l2=0.01, lr=0.3, V_lr=0.05, V_l2=0.02, V_threshold=0, V_init_scale=0.2, seed=5)
V_DIM = 5
STEPS = 4
WORLD = 2


def make_batches(rank):
    from conftest import random_batch
    rng = np.random.default_rng(100 + rank)
    return [random_batch(rng, 60, 2 ** 64 - 1 if i % 2 else 400, 12, binary=(i % 2 == 0)) for i in range(STEPS)]


def _splits(kind):
    """None = uniform ranges; "balanced" = split keys balanced on the keys the batches hold"""
    if kind != "balanced":
        return None
    from difacto_amd.sharded import balanced_splits
    from difacto_amd.synth import reverse_bytes_np
    ids = np.concatenate([b["index"] for r in range(WORLD) for b in make_batches(r)])
    return balanced_splits(reverse_bytes_np(ids), WORLD)


def _worker(rank, world, port, out_dir, kind, exchange):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sharded_harness import ShardedWorker
    from sharded_testlib import OracleBackend
    be = OracleBackend(V_DIM, HYPER)
    splits = _splits(kind)
    w = ShardedWorker(be, splits=splits, exchange=exchange)
    ahead = 2 if exchange == "overlap" else 1
    preds, infos = [], []
    batches = make_batches(rank)
    for i in range(min(ahead, len(batches))):
        w.submit(batches[i], True, i < 2)
    for i in range(len(batches)):
        if i + ahead < len(batches):  # later minibatches are localized (and their counts exchanged) during this step
            w.submit(batches[i + ahead], True, i + ahead < 2)
        infos.append(w.step())
        preds.append(be.pred().copy())
    # every rank owns a disjoint key range
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), preds=np.concatenate(preds), loss=be.loss,
             nkeys=be.store.size(), sent=np.array([x["sent"] for x in infos]), recv=np.array([x["received"] for x in infos]))
    # final owned model, key by key
    allkeys = np.unique(np.concatenate([be.o.localize(b["offset"], b["index"])["feaids"]
                                        for r in range(world) for b in make_batches(r)]))
    from difacto_amd.sharded import owner_of, uniform_splits
    mine = allkeys[owner_of(allkeys, uniform_splits(world) if splits is None else splits) == rank]
    vals, lens = be.store.pull(mine)
    np.savez(os.path.join(out_dir, "model%d.npz" % rank), keys=mine, vals=vals, lens=lens)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,exchange", [("uniform", "sync"), ("balanced", "sync"), ("balanced", "overlap")])
def test_sharded_world2_matches_single_store(tmp_path, oracle, kind, exchange):
    from oracle import bindings as ob
    port = 29600 + (os.getpid() % 300) + (7 if kind == "balanced" else 0) + (13 if exchange == "overlap" else 0)
    mp.spawn(_worker, args=(WORLD, port, str(tmp_path), kind, exchange), nprocs=WORLD, join=True)

    from sharded_testlib import emulate_single_store
    batches = [make_batches(r) for r in range(WORLD)]
    store, preds, loss = emulate_single_store(oracle, batches, V_DIM, HYPER, push_cnt_steps=2, overlap=(exchange == "overlap"))

    total_keys = 0
    for r in range(WORLD):
        got = np.load(os.path.join(tmp_path, "rank%d.npz" % r))
        assert np.array_equal(got["preds"], np.concatenate(preds[r])), "rank %d predictions" % r
        assert float(got["loss"]) == pytest.approx(loss[r], rel=1e-6)
        total_keys += int(got["nkeys"])
        # the exchange really crossed ranks
        assert got["sent"][:, 1 - r].sum() > 0 and got["recv"][:, 1 - r].sum() > 0
        m = np.load(os.path.join(tmp_path, "model%d.npz" % r))
        vals, lens = store.pull(m["keys"])
        assert np.array_equal(m["lens"], lens)
        assert np.array_equal(m["vals"], vals), "rank %d owned model" % r
        assert np.any(lens > 1)
    assert total_keys == store.size()


def test_key_span_partition_is_contiguous_and_total():
    from difacto_amd.sharded import key_span
    for world in (1, 2, 3, 4, 8):
        span = key_span(world)
        assert span * world >= 2 ** 64 - 1
        keys = np.sort(np.random.default_rng(world).integers(0, 2 ** 64 - 1, size=1000, dtype=np.uint64))
        owner = (keys // np.uint64(span)).astype(np.int64) if world > 1 else np.zeros(len(keys), np.int64)
        assert owner.min() >= 0 and owner.max() < world
        assert np.all(np.diff(owner) >= 0)  # ascending keys -> contiguous slices per owner


def test_balanced_splits_even_out_a_grouped_id_space():
    """ids carrying a feature-group id in their low bits (EncodeFeaGrpID, base.h:60-63): after
    ReverseBytes the group sits in the top bits and uniform ranges are skewed; balanced splits are not"""
    from difacto_amd.sharded import balanced_splits, owner_of, uniform_splits
    from difacto_amd.synth import reverse_bytes_np
    rng = np.random.default_rng(3)
    sizes = rng.integers(1000, 60000, size=39)
    ids = np.concatenate([(rng.integers(0, 2 ** 52, size=n, dtype=np.uint64) << np.uint64(12)) | np.uint64(g)
                          for g, n in enumerate(sizes)])
    keys = reverse_bytes_np(ids)
    for world in (2, 4, 8):
        uni = np.bincount(owner_of(keys, uniform_splits(world)), minlength=world)
        bal = np.bincount(owner_of(keys, balanced_splits(keys[::7], world)), minlength=world)
        assert bal.sum() == uni.sum() == len(keys)
        assert bal.max() / bal.mean() < 1.05
        assert uni.max() / uni.mean() > bal.max() / bal.mean()
