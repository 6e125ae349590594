"""The loop-back transport (dfh_comm_create_loopback, include/difacto_hip.h) — what `bench.py --emulate-world` measures
with — computes what a real rank computes: ONE GPU plays rank r of a W-rank job whose other ranks exist only as the
messages they would send.  Those messages are taken from a replay of the WHOLE job on one oracle store
(tests/test_shard_native.py: emulate / emulate_overlap, the documented order of the sharded step): the peers' keys and
counts inside r's range, the rows the other owners answer r's pulls with, the gradient rows the peers push for r's
keys.  Fed with them, the emulated rank's logits must be rank r's logits of the replay and its shard of the model the
replay's, at the tolerance of the real multi-rank tests (rtol 1e-5 / 2e-5)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HYPER = dict(l1=0.02, l2=0.01, lr=0.3, V_lr=0.05, V_l2=0.02, V_threshold=1, V_init_scale=0.2, seed=5)
V_DIM = 8
ROWS = 300
STEPS = 4


def dense_rows(vals, lens, k, stride):
    """ragged SGDUpdater::Get / gradient layout -> exchange rows [x, has_V, 0, 0 | V]"""
    n = len(lens)
    out = np.zeros((n, stride), np.float32)
    ends = np.cumsum(lens)
    begs = ends - lens
    out[:, 0] = vals[begs] if n else 0
    hv = lens > 1
    out[hv, 1] = 1.0
    for j in np.flatnonzero(hv):
        out[j, 4:4 + k] = vals[begs[j] + 1:ends[j]]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("W,r,mode,per_key", [(3, 1, "sync", 0), (4, 0, "sync", 0), (3, 2, "overlap", 0), (8, 5, "overlap", 0),
                                              (4, 0, "sync", 1), (8, 5, "overlap", 1)])
def test_loopback_rank_computes_what_the_real_rank_computes(oracle, W, r, mode, per_key):
    # per_key: ctx option "owner_per_key" — the owner side per distinct key (dfh_shard_count_pull_multi / _push_grad_listed)
    import test_shard_native as T
    from conftest import random_batch
    from difacto_amd import capi, sharded
    from difacto_amd.synth import reverse_bytes_np
    T.DENSE_IDS = mode == "overlap"
    old = T.PUSH_CNT_STEPS
    try:
        batches = []
        for p in range(W):
            rng = np.random.default_rng(4200 + p)
            batches.append([random_batch(rng, ROWS, 3000, 30, binary=(i % 2 == 0)) for i in range(STEPS)])
        ids = np.concatenate([b["index"] for bs in batches for b in bs])
        splits = sharded.balanced_splits(reverse_bytes_np(ids), W)
        rec = []
        T.PUSH_CNT_STEPS = 10 ** 9   # counts are pushed in every step (the flag is a property of the job)
        store, preds, loss = (T.emulate_overlap if mode == "overlap" else T.emulate)(oracle, batches, V_DIM, HYPER, splits, rec)
    finally:
        T.DENSE_IDS = False
        T.PUSH_CNT_STEPS = old
    lo = 0 if r == 0 else int(splits[r - 1])
    hi = 2 ** 64 if r == W - 1 else int(splits[r])
    stride = capi.row_stride(V_DIM)

    ctx = capi.Context(0)
    ctx.set_option("owner_per_key", per_key)
    comm = capi.Comm.loopback(ctx, r, W)
    assert "loop-back" in comm.info()
    tb = capi.Table(ctx, 1 << 16, V_dim=V_DIM, init_mode=capi.INIT_HASH, **HYPER)
    sh = capi.Shard(tb, comm, splits)
    if mode == "overlap":
        sh.set_exchange("overlap")
    keep = []

    def dev(a):
        d = capi.DeviceBuffer.from_numpy(ctx, np.ascontiguousarray(a))
        keep.append(d)
        return d.ptr

    for i in range(STEPS):   # every exchange of the job, in minibatch order (a FIFO per kind of exchange)
        st = rec[i]
        cnt_words = np.zeros((W, 2), np.int64)
        ks, cs, gs = [], [], []
        for p in range(W):
            if p == r:
                continue
            fk = st["locs"][p]["feaids"]
            m = (fk >= np.uint64(lo)) & (fk <= np.uint64(hi - 1))
            ks.append(fk[m])
            cs.append(st["locs"][p]["feacnt"][m])
            gs.append(dense_rows(st["grads"][p], st["pulled"][p][1], V_DIM, stride)[m])
            cnt_words[p] = (int(m.sum()), 1)
        comm.feed(capi.XCHG_COUNTS, dev(cnt_words))
        comm.feed(capi.XCHG_KEYS, dev(np.concatenate(ks)))
        comm.feed(capi.XCHG_CNT, dev(np.concatenate(cs).astype(np.float32)))
        comm.feed(capi.XCHG_ROWS, dev(dense_rows(st["pulled"][r][0], st["pulled"][r][1], V_DIM, stride)))
        comm.feed(capi.XCHG_GRADS, dev(np.concatenate(gs)))
    max_nnz = max(int(b["offset"][-1]) for b in batches[r])
    bts = [capi.Batch(ctx, ROWS, max_nnz) for _ in range(2)]
    ctx.set_pipeline(1)

    def prepare(j):
        if j >= STEPS:
            return None
        b = batches[r][j]
        bts[j % 2].load_host(b["offset"], b["index"], b["value"], b["label"])
        bts[j % 2].localize()
        return bts[j % 2]

    cur = prepare(0)
    for i in range(STEPS):
        nxt = prepare(i + 1)
        if nxt is not None:
            sh.prefetch_counts(nxt)   # its counts (and, overlapped, its keys and rows) travel inside this step
        assert sh.step(cur, is_train=True, push_cnt=True)
        np.testing.assert_allclose(cur.pred(), preds[r][i], rtol=1e-5, atol=1e-6, err_msg="step %d logits" % i)
        cur = nxt
    tb.check()
    allkeys = np.unique(np.concatenate([st["locs"][p]["feaids"] for st in rec for p in st["locs"]]))
    mine = allkeys[(allkeys >= np.uint64(lo)) & (allkeys <= np.uint64(hi - 1))]
    got_v, got_l = tb.pull(mine)
    want_v, want_l = store.pull(mine)
    assert np.array_equal(got_l, want_l) and np.any(want_l > 1)
    np.testing.assert_allclose(got_v, want_v, rtol=2e-5, atol=1e-6, err_msg="the emulated rank's shard of the model")
    assert tb.size() == len(mine)
    sent, recv, groups = comm.stats()
    assert sent > 0 and recv > 0 and groups >= 4 * STEPS
    for o_ in [sh] + bts + [tb, comm] + keep:
        o_.close()
    ctx.close()
