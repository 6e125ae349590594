"""The sharded store behind the C ABI (dfh_comm / dfh_shard / dfh_shard_step, csrc/dfh_shard.hip):
the exchange runs inside libdifacto_hip.so — RCCL ncclSend / ncclRecv in production — and every rank
calls one entry point per minibatch.

  * world 1 over the real RCCL transport: the step must equal the fused single-GPU step;
  * 2 and 4 ranks sharing the one GPU of the test box through the host-callback transport (gloo
    carries the bytes; RCCL cannot put two ranks on one device), with UNEVEN numbers of minibatches
    per rank (a rank whose data is exhausted keeps serving its shard), uniform and balanced key
    ranges: per-rank logits, loss, penalty and the union of the shards against ONE oracle store that
    receives the same requests in the documented order (count pushes, pulls, then gradient pushes: a
    key's owner first — its own keys never leave it and are updated in place by its backward pass —
    then the other sources in ascending rank order).
"""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HYPER = dict(l1=0.02, l2=0.01, lr=0.3, V_lr=0.05, V_l2=0.02, V_threshold=1, V_init_scale=0.2, seed=5)
V_DIM = 8
ROWS = 300
PUSH_CNT_STEPS = 2


def nsteps(rank):
    return 5 - (rank % 2) * 2  # ranks 0, 2: five minibatches; ranks 1, 3: three


def push_cnt_steps():
    """steps that push feature counts.  The overlapped exchange pulls the NEXT minibatch inside the current call, with the
    current call's push_cnt: the flag is a property of the job (epoch 0 or not, sgd_learner.cc:201-202), so those cases
    keep it constant"""
    return 10 ** 9 if DENSE_IDS else PUSH_CNT_STEPS


DENSE_IDS = False   # the overlapped-exchange cases: every step draws from the same 3000 ids, so that consecutive minibatches
                    # of all ranks share keys and the order of pulls and pushes across steps shows in the logits


def make_batches(rank):
    from conftest import random_batch
    rng = np.random.default_rng(900 + rank)
    return [random_batch(rng, ROWS, 2 ** 64 - 1 if (i % 2 and not DENSE_IDS) else 3000, 30, binary=(i % 2 == 0))
            for i in range(nsteps(rank))]


def subset_push(store, kind, keys, grads, lens, mask):
    """push the entries of a ragged gradient vector selected by mask (one bool per key)"""
    ends = np.cumsum(lens)
    begs = ends - lens
    sel = np.flatnonzero(mask)
    if len(sel) == 0:
        return
    g = np.concatenate([grads[begs[j]:ends[j]] for j in sel])
    store.push(keys[sel], kind, g, lens[sel])


def emulate(oracle, batches, V_dim, hyper, splits, rec=None):
    """ONE store receiving the ranks' requests of every step: count pushes, pulls, then the gradient pushes —
    per key the owner's own push first, then the other ranks' in ascending rank order.  rec (a list): per step what every
    rank localized, pulled and pushed (tests/test_loopback_emulation.py plays one rank's peers back from it)"""
    from oracle import bindings as ob
    from difacto_amd import sharded
    world = len(batches)
    steps = max(len(b) for b in batches)
    store = oracle.store_create(init_mode=ob.INIT_HASH, V_dim=V_dim, **hyper)
    preds = [[] for _ in range(world)]
    loss = [0.0] * world
    for i in range(steps):
        act = [r for r in range(world) if i < len(batches[r])]
        locs = {r: oracle.localize(batches[r][i]["offset"], batches[r][i]["index"]) for r in act}
        if i < PUSH_CNT_STEPS:
            for r in act:
                store.push(locs[r]["feaids"], ob.FEA_COUNT, locs[r]["feacnt"])
        pulled = {r: store.pull(locs[r]["feaids"]) for r in act}
        grads = {}
        for r in act:
            b, loc = batches[r][i], locs[r]
            vals, lens = pulled[r]
            wp, vp = oracle.get_pos(lens)
            p = oracle.fm_predict(V_dim, loc["offset"], loc["index"], b["value"], vals, wp, vp)
            preds[r].append(p)
            loss[r] += oracle.loss_evaluate(b["label"], p)
            grads[r] = oracle.fm_calcgrad(V_dim, loc["offset"], loc["index"], b["value"], b["label"], vals, p, wp, vp)
        owner = {r: sharded.owner_of(locs[r]["feaids"], splits) for r in act}
        if rec is not None:
            rec.append(dict(locs=locs, pulled=pulled, grads=grads))
        for r in act:   # every owner's own keys first
            subset_push(store, ob.GRADIENT, locs[r]["feaids"], grads[r], pulled[r][1], owner[r] == r)
        for r in act:   # then the keys other ranks own, source rank after source rank
            subset_push(store, ob.GRADIENT, locs[r]["feaids"], grads[r], pulled[r][1], owner[r] != r)
    return store, preds, loss


def merge_ragged(mask_a, ra, rb):
    """(vals, lens) of a key list from the pulls of its two halves: the keys with mask_a from ra, the others from rb"""
    (va, la), (vb, lb) = ra, rb
    lens = np.zeros(len(mask_a), np.int32)
    lens[mask_a] = la
    lens[~mask_a] = lb
    ends = np.cumsum(lens)
    begs = ends - lens
    vals = np.zeros(int(ends[-1]) if len(lens) else 0, np.float32)
    ea, eb = np.cumsum(la), np.cumsum(lb)
    ia = ib = 0
    for j in range(len(lens)):
        if mask_a[j]:
            vals[begs[j]:ends[j]] = va[ea[ia] - la[ia]:ea[ia]]
            ia += 1
        else:
            vals[begs[j]:ends[j]] = vb[eb[ib] - lb[ib]:eb[ib]]
            ib += 1
    return vals, lens


def emulate_overlap(oracle, batches, V_dim, hyper, splits, rec=None):
    """ONE store receiving the requests of the OVERLAPPED exchange (dfh_shard_set_exchange(s, 1), include/difacto_hip.h)
    in its documented order.  Per step t:  L(t): every rank count-pushes and reads the keys it owns itself (zero
    staleness);  [first step only: the other owners' keys are count-pushed and pulled now];  F(t);  the own keys'
    gradients are applied;  R(t+1): the keys of step t+1 that other ranks own are count-pushed and pulled — BEFORE —
    P(t): the gradients of step t for other owners' keys are applied, source rank after source rank."""
    from oracle import bindings as ob
    from difacto_amd import sharded
    world = len(batches)
    steps = max(len(b) for b in batches)
    store = oracle.store_create(init_mode=ob.INIT_HASH, V_dim=V_dim, **hyper)
    preds = [[] for _ in range(world)]
    loss = [0.0] * world
    locs = [{r: oracle.localize(batches[r][i]["offset"], batches[r][i]["index"]) for r in range(world) if i < len(batches[r])}
            for i in range(steps)]
    mine = [{r: sharded.owner_of(l["feaids"], splits) == r for r, l in locs[i].items()} for i in range(steps)]

    def remote_pull(i):
        if i < push_cnt_steps():
            for r, l in locs[i].items():
                m = ~mine[i][r]
                store.push(l["feaids"][m], ob.FEA_COUNT, l["feacnt"][m])
        return {r: store.pull(l["feaids"][~mine[i][r]]) for r, l in locs[i].items()}

    remote = None
    for i in range(steps):
        act = sorted(locs[i])
        own = {}
        for r in act:   # L: own keys
            l, m = locs[i][r], mine[i][r]
            if i < push_cnt_steps():
                store.push(l["feaids"][m], ob.FEA_COUNT, l["feacnt"][m])
            own[r] = store.pull(l["feaids"][m])
        if i == 0:
            remote = remote_pull(0)
        grads, lens_of = {}, {}
        for r in act:   # F
            b, l = batches[r][i], locs[i][r]
            vals, lens = merge_ragged(mine[i][r], own[r], remote[r])
            wp, vp = oracle.get_pos(lens)
            p = oracle.fm_predict(V_dim, l["offset"], l["index"], b["value"], vals, wp, vp)
            preds[r].append(p)
            loss[r] += oracle.loss_evaluate(b["label"], p)
            grads[r] = oracle.fm_calcgrad(V_dim, l["offset"], l["index"], b["value"], b["label"], vals, p, wp, vp)
            lens_of[r] = lens
            if rec is not None:
                if len(rec) <= i:
                    rec.append(dict(locs=locs[i], pulled={}, grads=grads))
                rec[i]["pulled"][r] = (vals, lens)
        for r in act:   # the own keys' update (fused, in place)
            subset_push(store, ob.GRADIENT, locs[i][r]["feaids"], grads[r], lens_of[r], mine[i][r])
        nxt = remote_pull(i + 1) if i + 1 < steps else None   # R(t+1) ahead of P(t)
        for r in act:   # P(t): the other owners' keys, source rank after source rank
            subset_push(store, ob.GRADIENT, locs[i][r]["feaids"], grads[r], lens_of[r], ~mine[i][r])
        remote = nxt
    return store, preds, loss


def _worker(rank, world, port, out_dir, balanced, prefetch, mode="sync"):
    global DENSE_IDS
    DENSE_IDS = mode == "overlap"
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from difacto_amd import capi, sharded
    from difacto_amd.synth import reverse_bytes_np
    ctx = capi.Context(0)

    def exchange(send, sb, recv, rb):  # bytes in, bytes out: gloo all_to_all_v on the host
        out = torch.empty(sum(rb), dtype=torch.uint8)
        dist.all_to_all_single(out, torch.from_numpy(np.array(send, copy=True)), output_split_sizes=rb, input_split_sizes=sb)
        recv[:] = out.numpy()

    comm = capi.Comm.callback(ctx, rank, world, exchange)
    comm.selfcheck(30.0)   # collective: every rank's id arrives everywhere, or an error instead of a hang
    # the wire probe bench.py --gpus N runs before its timed region (dfh_comm_wire_probe): collective, a positive time per
    # grouped exchange, and the job's own statistics untouched by it
    st0 = comm.stats(reset=False)
    assert comm.wire_probe(4096, reps=2) > 0.0
    assert comm.stats(reset=False) == st0
    assert "callback" in comm.info()
    comm.stats(reset=True)
    splits = None
    if balanced:
        ids = np.concatenate([b["index"] for r in range(world) for b in make_batches(r)])
        splits = sharded.balanced_splits(reverse_bytes_np(ids), world)
    tb = capi.Table(ctx, 1 << 16, V_dim=V_DIM, init_mode=capi.INIT_HASH, **HYPER)
    sh = capi.Shard(tb, comm, splits)
    if mode == "overlap":
        sh.set_exchange("overlap")   # two minibatches in flight: needs the next one announced (prefetch)
        sh.set_timing(True)
    if balanced:   # exchange buffers up front (dfh_shard_reserve: no re-allocation inside a step); the other cases let them grow
        sh.reserve(ROWS * 30, 2 * ROWS * 30 * world)
    batches = make_batches(rank)
    max_nnz = max(int(b["offset"][-1]) for b in batches)
    bts = [capi.Batch(ctx, ROWS, max_nnz) for _ in range(2)]
    bt = bts[0]
    if prefetch:
        ctx.set_pipeline(1)   # the next minibatch is localized on the preparation stream while this one steps

    def prepare(j):
        if j >= len(batches):
            return None
        b = batches[j]
        bts[j % 2].load_host(b["offset"], b["index"], b["value"], b["label"])
        bts[j % 2].localize()
        return bts[j % 2]

    preds, i = [], 0
    cur = prepare(0)
    while True:
        if prefetch:
            # the following step's per-owner counts travel inside this step (dfh_shard_prefetch_counts)
            nxt = prepare(i + 1)
            sh.prefetch_counts(nxt)
        active = sh.step(cur, is_train=True, push_cnt=i < push_cnt_steps())
        if not active:
            break
        if cur is not None:
            preds.append(cur.pred().copy())
        i += 1
        cur = nxt if prefetch else prepare(i)
    assert i == max(nsteps(r) for r in range(world))
    tb.check()
    if mode == "overlap":
        ms, n = sh.get_timing()
        assert n == i + 1 and ms["F"] > 0 and ms["K"] > 0 and ms["G"] > 0, (ms, n)
    progs = [x.progress() for x in bts]
    import types
    prog = types.SimpleNamespace(loss=sum(p.loss for p in progs), nrows=sum(p.nrows for p in progs),
                                 penalty=sum(p.penalty for p in progs))
    # dfh_comm_stats: bytes that left this rank for OTHER ranks / arrived from them.  Summed over the job they are equal,
    # and with every minibatch's keys spread over all owners nobody's exchange is empty
    sent, recv, groups = comm.stats()
    assert groups >= i and sent > 0 and recv > 0
    tot = comm.allreduce_sum([prog.loss, prog.nrows, 1.0, float(sent), float(recv)])
    assert tot[2] == world and tot[3] == tot[4]
    from oracle import bindings as ob
    o = ob.Oracle()
    allkeys = np.unique(np.concatenate([o.localize(b["offset"], b["index"])["feaids"]
                                        for r in range(world) for b in make_batches(r)]))
    sp = sharded.uniform_splits(world) if splits is None else splits
    mine = allkeys[sharded.owner_of(allkeys, sp) == rank]
    lo, hi = sh.owned_range()
    assert (len(mine) == 0 or int(mine.min()) >= lo) and (hi == 0 or len(mine) == 0 or int(mine.max()) < hi)
    vals, lens = tb.pull(mine)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), preds=np.concatenate(preds), loss=prog.loss, penalty=prog.penalty,
             nrows=prog.nrows, total_loss=tot[0], total_rows=tot[1], nkeys=tb.size(), keys=mine, vals=vals, lens=lens)
    dist.barrier()
    for o_ in [sh] + bts + [tb, comm]:
        o_.close()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("WORLD,balanced,prefetch,exchange", [(2, False, False, "sync"), (2, True, True, "sync"), (4, True, False, "sync"),
                                                              (4, False, True, "sync"), (2, True, True, "overlap"),
                                                              (4, False, True, "overlap"), (8, True, True, "overlap"),
                                                              (8, False, False, "sync")])
def test_shard_step_ranks_share_one_gpu(tmp_path, oracle, WORLD, balanced, prefetch, exchange):
    port = 29700 + (os.getpid() % 80) + WORLD + (10 if prefetch else 0) + (20 if exchange == "overlap" else 0)
    mp.spawn(_worker, args=(WORLD, port, str(tmp_path), balanced, prefetch, exchange), nprocs=WORLD, join=True)
    global DENSE_IDS
    DENSE_IDS = exchange == "overlap"
    if os.environ.get("DFH_DUMP"):   # debugging aid: keep the ranks' outputs
        import shutil
        dst = os.path.join(os.environ["DFH_DUMP"], "w%d_%s_%s" % (WORLD, "bal" if balanced else "uni", exchange))
        shutil.copytree(str(tmp_path), dst, dirs_exist_ok=True)
    batches = [make_batches(r) for r in range(WORLD)]
    from difacto_amd import sharded
    from difacto_amd.synth import reverse_bytes_np
    if balanced:
        ids = np.concatenate([b["index"] for r in range(WORLD) for b in batches[r]])
        splits = sharded.balanced_splits(reverse_bytes_np(ids), WORLD)
    else:
        splits = sharded.uniform_splits(WORLD)
    store, preds, loss = (emulate_overlap if exchange == "overlap" else emulate)(oracle, batches, V_DIM, HYPER, splits)
    total, total_loss = 0, 0.0
    for r in range(WORLD):
        got = np.load(os.path.join(tmp_path, "rank%d.npz" % r))
        np.testing.assert_allclose(got["preds"], np.concatenate(preds[r]), rtol=1e-5, atol=1e-6, err_msg="rank %d preds" % r)
        assert float(got["loss"]) == pytest.approx(loss[r], rel=1e-5)
        assert float(got["nrows"]) == ROWS * nsteps(r)
        assert float(got["penalty"]) > 0
        vals, lens = store.pull(got["keys"])
        assert np.array_equal(got["lens"], lens)
        np.testing.assert_allclose(got["vals"], vals, rtol=2e-5, atol=1e-6, err_msg="rank %d owned model" % r)
        assert np.any(lens > 1)
        total += int(got["nkeys"])
        total_loss += loss[r]
        assert float(got["total_rows"]) == sum(ROWS * nsteps(q) for q in range(WORLD))
    assert float(got["total_loss"]) == pytest.approx(total_loss, rel=1e-5)
    assert total == store.size()
    if exchange == "overlap":   # the replayed order matters: the sync order gives other logits on these minibatches
        _, preds_sync, _ = emulate(oracle, batches, V_DIM, HYPER, splits)
        assert max(np.abs(np.concatenate(preds_sync[r]) - np.concatenate(preds[r])).max() for r in range(WORLD)) > 1e-3
    DENSE_IDS = False


@pytest.mark.gpu
def test_shard_step_world1_over_rccl_matches_fused():
    """one rank over the real transport (RCCL loaded at run time): every key is the rank's own, so the step
    is the fused step (k_lookup -> k_forward<MIXED> on the table -> k_backward_all with the in-place
    update) and must give what dfh_sgd_step gives"""
    from conftest import random_batch
    from difacto_amd import capi
    ctx = capi.Context(0)
    comm = capi.Comm.rccl(ctx, 0, 1, capi.Comm.unique_id())
    comm.selfcheck(30.0)
    assert comm.wire_probe(1 << 20, reps=2) == 0.0   # one rank: no wires
    info = comm.info()   # which RCCL: version + the file it was bound from (the bench line logs it)
    assert info.startswith("rccl ") and "librccl" in info and int(info.split()[1]) > 20000, info
    kw = dict(l1=0.02, l2=0.01, lr=0.3, V_lr=0.05, V_l2=0.02, V_threshold=0, V_init_scale=0.2, seed=2)
    rng = np.random.default_rng(77)
    batches = [random_batch(rng, 200, 4000, 25, binary=(i == 1)) for i in range(3)]
    ta = capi.Table(ctx, 1 << 15, V_dim=16, **kw)
    tb = capi.Table(ctx, 1 << 15, V_dim=16, **kw)
    sh = capi.Shard(tb, comm)
    max_nnz = max(int(b["offset"][-1]) for b in batches)
    ba, bb = capi.Batch(ctx, 200, max_nnz), capi.Batch(ctx, 200, max_nnz)
    for epoch in range(2):
        for b in batches:
            for bt in (ba, bb):
                bt.load_host(b["offset"], b["index"], b["value"], b["label"])
                bt.localize()
            ba.sgd_step(ta, is_train=True, push_cnt=(epoch == 0))
            assert sh.step(bb, is_train=True, push_cnt=(epoch == 0))
            np.testing.assert_allclose(bb.pred(), ba.pred(), rtol=1e-5, atol=1e-6)
            pa, pb = ba.progress(), bb.progress()
            assert pb.loss == pytest.approx(pa.loss, rel=1e-6) and pb.penalty == pytest.approx(pa.penalty, rel=1e-5)
    assert not sh.step(None)  # nobody has data: the epoch is over
    ea, eb = ta.export(), tb.export()
    oa, ob_ = np.argsort(ea["keys"]), np.argsort(eb["keys"])
    assert np.array_equal(ea["keys"][oa], eb["keys"][ob_]) and np.array_equal(ea["has_V"][oa], eb["has_V"][ob_])
    np.testing.assert_allclose(eb["scal"][ob_], ea["scal"][oa], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(eb["V"][ob_], ea["V"][oa], rtol=2e-5, atol=1e-6)
    assert comm.allreduce_sum([1.5, 2.0]).tolist() == [1.5, 2.0]
    sent, recv, groups = comm.stats()
    assert sent == 0 and recv == 0 and groups > 0   # one rank: nothing leaves it, the exchanges still ran
    for o in (sh, ba, bb, ta, tb, comm):
        o.close()
    ctx.close()


# ------------------------------------------------------------------ the literal Store::Push / Pull on the sharded model
def _literal_requests(rank, world):
    """per call and rank: ascending unique keys (rank 1 of 3 asks nothing in the middle call) and what it pushes"""
    rng = np.random.default_rng(4000 + rank)
    calls = []
    for c in range(3):
        n = 0 if (c == 1 and rank == 1) else int(rng.integers(50, 400))
        keys = np.unique(rng.integers(1, 2 ** 64 - 1 if c == 2 else 900, size=n, dtype=np.uint64))
        calls.append(dict(keys=keys, cnt=rng.integers(1, 6, size=len(keys)).astype(np.float32), seed=int(rng.integers(1 << 30))))
    return calls


def _literal_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from difacto_amd import capi

    def exchange(send, sb, recv, rb):
        out = torch.empty(sum(rb), dtype=torch.uint8)
        dist.all_to_all_single(out, torch.from_numpy(np.array(send, copy=True)), output_split_sizes=rb, input_split_sizes=sb)
        recv[:] = out.numpy()

    ctx = capi.Context(0)
    comm = capi.Comm.callback(ctx, rank, world, exchange)
    assert np.array_equal(comm.allgather(np.arange(3, dtype=np.int64) + 10 * rank),
                          np.stack([np.arange(3, dtype=np.int64) + 10 * r for r in range(world)]))
    tb = capi.Table(ctx, 1 << 14, V_dim=V_DIM, init_mode=capi.INIT_HASH, **HYPER)
    sh = capi.Shard(tb, comm)
    out = {}
    for c, q in enumerate(_literal_requests(rank, world)):
        keys = q["keys"]
        sh.push_host(keys, capi.FEA_COUNT, q["cnt"])
        vals, lens = sh.pull_host(keys)
        g = np.random.default_rng(q["seed"]).normal(size=len(vals)).astype(np.float32)
        sh.push_host(keys, capi.GRADIENT, g, lens)
        v2, l2 = sh.pull_host(keys)
        out["vals%d" % c], out["lens%d" % c], out["after%d" % c], out["lens_after%d" % c] = vals, lens, v2, l2
    tb.check()
    np.savez(os.path.join(out_dir, "lit%d.npz" % rank), **out)
    dist.barrier()
    for o_ in (sh, tb, comm):
        o_.close()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("WORLD", [1, 3])
def test_literal_push_pull_on_the_sharded_store(tmp_path, oracle, WORLD):
    """dfh_shard_push_host / dfh_shard_pull_host = Store::Push / Pull with host arrays, collective over the ranks (a rank
    may ask nothing): every call against ONE oracle store that receives the ranks' requests in ascending rank order"""
    from oracle import bindings as ob
    port = 29650 + (os.getpid() % 40) + WORLD
    mp.spawn(_literal_worker, args=(WORLD, port, str(tmp_path)), nprocs=WORLD, join=True)
    store = oracle.store_create(init_mode=ob.INIT_HASH, V_dim=V_DIM, **HYPER)
    reqs = [_literal_requests(r, WORLD) for r in range(WORLD)]
    got = [np.load(os.path.join(tmp_path, "lit%d.npz" % r)) for r in range(WORLD)]
    for c in range(3):
        for r in range(WORLD):
            store.push(reqs[r][c]["keys"], ob.FEA_COUNT, reqs[r][c]["cnt"])
        pulled = [store.pull(reqs[r][c]["keys"]) for r in range(WORLD)]
        for r in range(WORLD):
            assert np.array_equal(got[r]["lens%d" % c], pulled[r][1]) and np.array_equal(got[r]["vals%d" % c], pulled[r][0]), (c, r)
        for r in range(WORLD):   # ascending rank order
            vals, lens = pulled[r]
            g = np.random.default_rng(reqs[r][c]["seed"]).normal(size=len(vals)).astype(np.float32)
            store.push(reqs[r][c]["keys"], ob.GRADIENT, g, lens)
        for r in range(WORLD):
            v, l = store.pull(reqs[r][c]["keys"])
            assert np.array_equal(got[r]["lens_after%d" % c], l)
            np.testing.assert_allclose(got[r]["after%d" % c], v, rtol=2e-6, atol=1e-7, err_msg="call %d rank %d" % (c, r))
