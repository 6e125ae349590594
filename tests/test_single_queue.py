"""GPU: the single-queue step (difacto_amd/csrc/dfh_riders.hip) against the serial step.

In the single-queue step the four stages of the device Localizer (Localizer::Compact, src/data/localizer.cc:11-103) of the
NEXT minibatches ride as extra blocks of the launches the current step makes anyway; the reference gets the same overlap
from its reader thread and batch tracker (src/sgd/sgd_learner.cc:196-224).  The stages are the same block functions on
the same arguments wherever they run, so everything must come out BIT FOR BIT as in the serial order: the Localizer's
outputs (integer work), every step's logits, the model.
"""
import numpy as np
import pytest

from conftest import random_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from difacto_amd import capi as m
    m.lib()
    return m


def criteo_like(rng, rows, slots, ids):
    """rows x slots binary features, one per slot, Zipf-ish ids (many keys occur once, a few very often)"""
    off = (np.arange(rows + 1) * slots).astype(np.uint64)
    r = (rng.zipf(1.3, size=rows * slots) % ids).astype(np.uint64)
    slot = np.tile(np.arange(slots, dtype=np.uint64), rows)
    idx = (r << np.uint64(12)) | slot
    lab = np.where(rng.random(rows) < 0.25, 1.0, 0.0).astype(np.float32)
    return dict(offset=off, index=idx, value=None, label=lab)


# (options, minibatches prepared ahead): which launch carries which stage — the default (scatter in the lookup pass, sort in the
# forward, emit + the next count in the update), everything in the update launch four deep, stages alone, riders first
POLICIES = [
    ({}, 2),
    ({}, 1),
    ({}, 3),
    ({"rider_slot_count": 2, "rider_slot_scatter": 2, "rider_slot_sort": 2, "rider_slot_emit": 2}, 4),
    ({"rider_slot_count": 2 + 4, "rider_slot_scatter": 0, "rider_slot_sort": 1 + 4, "rider_slot_emit": 2}, 2),
    ({"rider_slot_count": 0, "rider_slot_scatter": 1, "rider_slot_sort": 2, "rider_slot_emit": 0}, 2),
    ({"rider_period_lookup": 1, "rider_period_forward": 1, "rider_period_update": 1}, 2),
    ({"rider_period_lookup": 64, "rider_period_forward": 64, "rider_period_update": 64}, 2),
    ({"rider_start_lookup": 100, "rider_start_forward": 50, "rider_start_update": 80, "rider_period_update": 1}, 2),
    ({"rider_slot_count": 2, "rider_slot_scatter": 2, "rider_slot_sort": 2, "rider_slot_emit": 2, "rider_start_update": 50,
      "rider_period_update": 1}, 4),
]


def run_stream(capi, batches, kw, V_dim, max_rows, single, opts, ahead, nsteps, is_train_of, inspect_every=0):
    ctx = capi.Context(0)
    if single:
        ctx.set_option("single_queue", 1)
        for name, val in opts.items():
            ctx.set_option(name, val)
    max_nnz = max(int(b["offset"][-1]) for b in batches)
    tb = capi.Table(ctx, 1 << 18, V_dim=V_dim, **kw)
    bts = [capi.Batch(ctx, max_rows, max_nnz) for _ in range(ahead + 1)]
    for b_ in bts:
        b_.set_option("compute_auc", 1)

    def prep(i):
        b = batches[i % len(batches)]
        bt = bts[i % len(bts)]
        bt.load_host(b["offset"], b["index"], b["value"], b["label"])
        bt.localize()
        bt.lookup(tb)

    preds, locs = [], []
    for i in range(min(ahead, nsteps)):
        prep(i)
    for i in range(nsteps):
        if i + ahead < nsteps:
            prep(i + ahead)
        bt = bts[i % len(bts)]
        if inspect_every and i % inspect_every == 0:
            locs.append(bt.get_localized())      # a consumer other than the step: the stages still noted are queued first
        bt.sgd_step(tb, is_train=is_train_of(i), push_cnt=(i < len(batches)))
        preds.append(bt.pred())
    prog = [b_.progress(reset=True) for b_ in bts]
    tot = (sum(p.loss for p in prog), sum(p.auc for p in prog), sum(p.penalty for p in prog), sum(p.nrows for p in prog))
    keys = np.unique(np.concatenate([capi.reverse_bytes_np(b["index"]) if hasattr(capi, "reverse_bytes_np") else
                                     _rev(b["index"]) for b in batches]))
    model = tb.pull(keys)
    for o in bts + [tb]:
        o.close()
    ctx.close()
    return preds, locs, tot, model


def _rev(ids):
    from difacto_amd import synth
    return synth.reverse_bytes_np(np.ascontiguousarray(ids, np.uint64))


def _same(a, b, what):
    assert len(a) == len(b), what
    for i, (x, y) in enumerate(zip(a, b)):
        if isinstance(x, (tuple, list)):
            _same(x, y, "%s[%d]" % (what, i))
        elif isinstance(x, dict):
            for k_ in x:
                assert np.array_equal(np.asarray(x[k_]), np.asarray(y[k_])), "%s[%d].%s" % (what, i, k_)
        else:
            assert np.array_equal(np.asarray(x), np.asarray(y)), "%s[%d]" % (what, i)


def _sums(got, ref):
    """progress sums: dfh_progress holds fp32 sums per batch object, added up here over a different number of objects — equal to fp32 rounding"""
    for g, r in zip(got, ref):
        assert g == pytest.approx(r, rel=1e-6), "progress sums"


@pytest.mark.parametrize("opts,ahead", POLICIES)
def test_single_queue_bit_identical_to_serial(capi, opts, ahead):
    """a stream of Criteo-shaped minibatches of one size class (the steady state: splitters stored, every stage rides) —
    logits of every step, progress sums and the model bit for bit the serial step's, whatever launch carries which stage"""
    rng = np.random.default_rng(7)
    batches = [criteo_like(rng, 1500, 39, 40_000) for _ in range(6)]
    kw = dict(l1=0.02, l2=0.01, lr=0.3, V_lr=0.05, V_l2=0.02, V_threshold=2, V_init_scale=0.2, seed=5)
    nsteps = 17
    train = lambda i: i % 7 != 6   # a validation step now and then: its launches carry riders too (or the stages run alone)
    ref = run_stream(capi, batches, kw, 8, 1500, False, {}, 1, nsteps, train)
    got = run_stream(capi, batches, kw, 8, 1500, True, opts, ahead, nsteps, train)
    _same(got[0], ref[0], "pred")
    _sums(got[2], ref[2])
    _same(list(got[3]), list(ref[3]), "model")


@pytest.mark.parametrize("V_dim,binary", [(0, True), (5, False), (64, True)])
def test_single_queue_ragged_sizes_and_consumers(capi, V_dim, binary):
    """ragged minibatches that change size class (the splitters are bootstrapped again: such a call is queued at once, not
    noted), empty rows, real values, a minibatch read back (dfh_batch_get_localized) before its stages have all found a
    carrier: the Localizer's outputs, the logits and the model bit for bit the serial step's"""
    rng = np.random.default_rng(11 + V_dim)
    batches = [random_batch(rng, int(rng.integers(20, 400)), 3000, 40, binary=binary, empty_rows=(i % 3 == 0)) for i in range(7)]
    kw = dict(l1=0.01, l2=0.0, lr=0.2, V_lr=0.05, V_l2=0.01, V_threshold=0, V_init_scale=0.1, seed=3)
    nsteps = 19
    train = lambda i: i % 5 != 4
    ref = run_stream(capi, batches, kw, V_dim, 400, False, {}, 1, nsteps, train, inspect_every=3)
    got = run_stream(capi, batches, kw, V_dim, 400, True, {}, 2, nsteps, train, inspect_every=3)
    _same(got[1], ref[1], "localized")
    _same(got[0], ref[0], "pred")
    _sums(got[2], ref[2])
    _same(list(got[3]), list(ref[3]), "model")


def test_single_queue_against_the_oracle(capi, oracle):
    """the riding Localizer against Localizer::Compact itself (oracle, bit-exact) at a size where every stage rides, and the
    step's logits against FMLoss::Predict on the weights the device pulls (rtol 1e-5 + the summation floor)"""
    from oracle import tolerance as T
    rng = np.random.default_rng(3)
    batches = [criteo_like(rng, 2000, 39, 100_000) for _ in range(5)]
    kw = dict(l1=0.02, l2=0.01, lr=0.3, V_lr=0.05, V_l2=0.02, V_threshold=0, V_init_scale=0.2, seed=5)
    k = 8
    ctx = capi.Context(0)
    ctx.set_option("single_queue", 1)
    tb = capi.Table(ctx, 1 << 18, V_dim=k, **kw)
    max_nnz = max(int(b["offset"][-1]) for b in batches)
    bts = [capi.Batch(ctx, 2000, max_nnz) for _ in range(3)]

    def prep(i):
        b = batches[i % len(batches)]
        bts[i % 3].load_host(b["offset"], b["index"], b["value"], b["label"])
        bts[i % 3].localize()

    prep(0)
    prep(1)
    for i in range(12):
        if i + 2 < 12:
            prep(i + 2)
        b = batches[i % len(batches)]
        bt = bts[i % 3]
        loc = oracle.localize(b["offset"], b["index"])
        if i >= 6:   # (from the third round on every stage of this minibatch rode in an earlier step's launches)
            got = bt.get_localized()
            assert np.array_equal(got["feaids"], loc["feaids"])
            assert np.array_equal(got["index"], loc["index"])
            assert np.array_equal(got["feacnt"], loc["feacnt"])
        if i < 5:   # the step's own count push precedes its pull (sgd_learner.cc:214-217)
            tb.push(loc["feaids"], capi.FEA_COUNT, loc["feacnt"])
        vals, lens = tb.pull(loc["feaids"])   # the weights this step sees
        w_pos, V_pos = oracle.get_pos(lens)
        want = oracle.fm_predict(k, loc["offset"], loc["index"], None, vals, w_pos, V_pos)
        bt.sgd_step(tb, is_train=True, push_cnt=False)
        w64, V64, _has = T.dense_rows(vals, lens, k)
        _, floor_p = T.predict_bound(T.design(loc["offset"], loc["index"], None, loc["U"]), w64, V64)
        T.check(bt.pred(), want, floor_p, "logits of step %d" % i)
    for o in bts + [tb]:
        o.close()
    ctx.close()
