"""Direct parity of the kernels bench.py times — k_forward<L,D> and k_backward_all<L,...> (hot / mid /
short-segment roles, fused and non-fused) — against the oracle's FMLoss::Predict / CalcGrad
(/root/reference/src/loss/fm_loss.h:67-119, 148-199) on IDENTICAL minibatches and IDENTICAL,
non-trivial weights (mixed has-V / no-V keys):

  (i)  packed path:  rows pulled from the oracle's model -> exchange layout -> dfh_batch_forward +
       dfh_batch_backward; per-example logits and per-key gradients at rtol 1e-5;
  (ii) fused path:   the oracle's model imported with dfh_table_import, ONE dfh_sgd_step; the step's
       logits at rtol 1e-5, and the model it leaves against SGDUpdater::Update
       (src/sgd/sgd_updater.cc:58-148) applied by the oracle to the oracle's own gradients.

Shapes: the reference's rcv1 fixture with V_dim 8 (BASELINE config C2), ragged random batches with
V_dim 0 / 4 / 5 / 64 / 128, a hot-key batch (segments of ~1000 occurrences), and the full-size C3
minibatch (10 000 rows x 39 slots from the 33 M id space) with V_dim 64 and 128.

Tolerance (oracle/tolerance.py): |got - ref| <= 1e-5 |ref| + floor, floor = C_SIGMA (= 1) * 2^-24 * sqrt(n) *
sum|terms| of every fp32 sum involved, computed in float64 from the same inputs.  On top of that
the tests require that most values agree at the pure rtol 1e-5 with no floor at all.
"""
import os
import zlib

import numpy as np
import pytest

from conftest import random_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from difacto_amd import capi as m
    m.lib()
    return m


@pytest.fixture(scope="module")
def ctx(capi):
    c = capi.Context(0)
    yield c
    c.close()


def _c3_batch(nrows=10000, seed=42):
    from difacto_amd import synth
    return synth.CriteoSynth(total_ids=33_000_000, seed=seed).batch(nrows)


def _hot_batch(rng, nrows=3000):
    """39 binary features per row over a tiny id space: segments of 1 .. ~2500 occurrences"""
    s = 39
    z = rng.zipf(1.3, size=nrows * s).astype(np.uint64) % np.uint64(5000)
    off = (np.arange(nrows + 1) * s).astype(np.uint64)
    lab = np.where(rng.random(nrows) < 0.3, 1.0, -1.0).astype(np.float32)
    return dict(offset=off, index=z, value=None, label=lab)


def _case(name, rcv1):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    if name == "c2_rcv1_k8":
        return rcv1, 8, 0.1
    if name.startswith("ragged_k"):
        k = int(name[len("ragged_k"):])
        return random_batch(rng, 500, 3000, 40, binary=(k % 2 == 1)), k, 0.1
    if name == "hot_k64":
        return _hot_batch(rng), 64, 0.05
    if name == "adversarial_k16":
        # SURVEY 7 "hard parts": |x| over six decades (1e-3 .. 1e3, random signs) on top of few ids, so that every sum is a
        # heavy cancellation of terms of very different size and part of the logits reach the +-20 clamp
        b = random_batch(rng, 400, 600, 40)
        nnz = len(b["index"])
        b["value"] = (np.sign(rng.normal(size=nnz)) * 10.0 ** rng.uniform(-3, 3, size=nnz)).astype(np.float32)
        return b, 16, 0.004
    if name == "tall_ragged_k4":
        # 30 000 rows of 0 .. 20 nonzeros (many empty): ~300 k pairs, so the Localizer's tags keep 13 row bits and
        # k_loc_emit has to pick every pair's row among four candidates (rows congruent modulo 8192)
        return random_batch(rng, 30000, 50000, 20), 4, 0.1
    if name == "c3_full_k64":
        return _c3_batch(), 64, 0.1
    if name == "c3_full_k128":
        return _c3_batch(seed=7), 128, 0.07
    raise KeyError(name)


CASES = ["c2_rcv1_k8", "ragged_k0", "ragged_k4", "ragged_k5", "ragged_k64", "ragged_k128", "hot_k64", "adversarial_k16",
         "tall_ragged_k4", "c3_full_k64", "c3_full_k128"]

# The tolerance of every comparison here is rtol 1e-5 |ref| + the summation-noise floor of oracle/tolerance.py; the
# backstop is the fraction of elements inside the PURE rtol 1e-5 (no floor).  PURE_MIN holds, per case and quantity, the
# fraction measured on MI355X (profiles/r04zz_parity.json, C_SIGMA = 1; newer records: profiles/r05*_parity.json) minus 0.01: a kernel change that pushes more elements onto the
# floor fails here even while every element is still inside the model's tolerance.  DFH_PARITY_RECORD=<file> re-records.
PURE_MIN = {
    'fused/adversarial_k16/l1=0': {'logits': 0.971},
    'fused/adversarial_k16/l1=0.05': {'logits': 0.971},
    'fused/c2_rcv1_k8/l1=0': {'logits': 0.98},
    'fused/c2_rcv1_k8/l1=0.05': {'logits': 0.98},
    'fused/c3_full_k128/l1=0': {'logits': 0.974},
    'fused/c3_full_k64/l1=0': {'logits': 0.976},
    'fused/hot_k64/l1=0': {'logits': 0.979},
    'fused/hot_k64/l1=0.05': {'logits': 0.979},
    'fused/ragged_k0/l1=0': {'logits': 0.981},
    'fused/ragged_k0/l1=0.05': {'logits': 0.981},
    'fused/ragged_k5/l1=0': {'logits': 0.987},
    'fused/ragged_k5/l1=0.05': {'logits': 0.987},
    'fused/ragged_k64/l1=0': {'logits': 0.979},
    'fused/ragged_k64/l1=0.05': {'logits': 0.979},
    'packed/adversarial_k16': {'grad_V': 0.984, 'grad_w': 0.99, 'logits': 0.977},
    'packed/c2_rcv1_k8': {'grad_V': 0.982, 'grad_w': 0.989, 'logits': 0.96},
    'packed/c3_full_k128': {'grad_V': 0.984, 'grad_w': 0.989, 'logits': 0.972},
    'packed/c3_full_k64': {'grad_V': 0.984, 'grad_w': 0.989, 'logits': 0.971},
    'packed/hot_k64': {'grad_V': 0.986, 'grad_w': 0.989, 'logits': 0.988},
    'packed/ragged_k0': {'grad_w': 0.989, 'logits': 0.987},
    'packed/ragged_k128': {'grad_V': 0.984, 'grad_w': 0.988, 'logits': 0.979},
    'packed/ragged_k4': {'grad_V': 0.983, 'grad_w': 0.988, 'logits': 0.979},
    'packed/ragged_k5': {'grad_V': 0.983, 'grad_w': 0.989, 'logits': 0.987},
    'packed/ragged_k64': {'grad_V': 0.984, 'grad_w': 0.988, 'logits': 0.981},
}
_RECORD = {}


def _pure(case, what, got, ref):
    f = _pure_rtol_fraction(got, ref)
    _RECORD.setdefault(case, {})["pure_rtol_" + what] = f
    lo = PURE_MIN.get(case, {}).get(what, 0.9)
    assert f >= lo, "%s %s: only %.4f of the elements within the pure rtol 1e-5 (recorded minimum %.4f)" % (case, what, f, lo)
    return f


def _worst(case, what, ratio):
    _RECORD.setdefault(case, {})["worst_err_over_tol_" + what] = ratio
    return ratio


@pytest.fixture(scope="module", autouse=True)
def _write_record():
    yield
    path = os.environ.get("DFH_PARITY_RECORD")
    if path and _RECORD:
        import json
        json.dump(_RECORD, open(path, "w"), indent=1, sort_keys=True)


def _weights(rng, U, k, scale, frac_no_v=0.3):
    """ragged (vals, lens) as Store::Pull returns them: w for every key, V for ~70 % of them"""
    if k == 0:
        return (rng.normal(size=U) * scale).astype(np.float32), np.zeros(0, np.int32)
    lens = np.where(rng.random(U) < frac_no_v, 1, 1 + k).astype(np.int32)
    return (rng.normal(size=int(lens.sum())) * scale).astype(np.float32), lens


def _pure_rtol_fraction(got, ref, rtol=1e-5):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    live = np.abs(ref) > 0
    return float((np.abs(got - ref)[live] <= rtol * np.abs(ref)[live]).mean()) if live.any() else 1.0


# ------------------------------------------------------------------ (i) packed rows
@pytest.mark.parametrize("name", CASES)
def test_hot_kernels_packed_rows_vs_oracle(capi, ctx, oracle, rcv1, name):
    """k_forward<L,5> and k_backward_all<L,false,false,false> on rows in the exchange layout"""
    from oracle import tolerance as T
    b, k, scale = _case(name, rcv1)
    rng = np.random.default_rng(17)
    loc = oracle.localize(b["offset"], b["index"])
    U = loc["U"]
    W, lens = _weights(rng, U, k, scale)
    wp, vp = (None, None) if k == 0 else oracle.get_pos(lens)
    # reference side
    po = oracle.fm_predict(k, loc["offset"], loc["index"], b["value"], W, wp, vp)
    # device side: the same minibatch through the device Localizer, the same rows in the exchange layout
    stride = capi.row_stride(k)
    rows = T.packed_rows(W, lens, k, stride)
    nnz = int(b["offset"][-1])
    bt = capi.Batch(ctx, len(b["label"]), nnz)
    bt.load_host(b["offset"], b["index"], b["value"], b["label"])
    bt.localize()
    got = bt.get_localized()
    assert np.array_equal(got["feaids"], loc["feaids"]) and np.array_equal(got["index"], loc["index"])
    d_rows = capi.DeviceBuffer.from_numpy(ctx, rows)
    d_grads = capi.DeviceBuffer(ctx, max(rows.nbytes, 16))
    bt.forward(k, d_rows.ptr)
    pg = bt.pred()
    X = T.design(loc["offset"], loc["index"], b["value"], U)
    w64, V64, has = T.dense_rows(W, lens, k)
    p64, floor_p = T.predict_bound(X, w64, V64)
    worst = _worst("packed/" + name, "logits", T.check(pg, po, floor_p, "%s logits vs oracle" % name))
    T.check(pg, p64, floor_p, "%s logits vs float64" % name)
    _pure("packed/" + name, "logits", pg, po)
    # gradients: FMLoss::CalcGrad takes the logits as an input (fm_loss.h:130-146): give the oracle
    # the device's, so that the comparison isolates CalcGrad itself
    go = oracle.fm_calcgrad(k, loc["offset"], loc["index"], b["value"], b["label"], W, pg, wp, vp)
    bt.backward(k, d_rows.ptr, d_grads.ptr)
    gg = d_grads.to_numpy(np.float32, rows.size).reshape(U, stride)
    gw_o, gV_o, _ = T.dense_rows(go, lens, k)
    gw64, gV64, floor_w, floor_V = T.calcgrad_bound(X, b["label"], pg, w64, V64, has)
    worst_w = _worst("packed/" + name, "grad_w", T.check(gg[:, 0], gw_o, floor_w, "%s grad_w vs oracle" % name))
    T.check(gg[:, 0], gw64, floor_w, "%s grad_w vs float64" % name)
    assert np.array_equal(gg[:, 1] != 0, has), "has_V flag of the gradient rows"
    assert not gg[:, 2:4].any()
    if k:
        worst_V = _worst("packed/" + name, "grad_V", T.check(gg[:, 4:4 + k], gV_o, floor_V, "%s grad_V vs oracle" % name))
        T.check(gg[:, 4:4 + k], gV64, floor_V, "%s grad_V vs float64" % name)
        assert not gg[~has][:, 4:].any(), "keys pulled without V get no V gradient (fm_loss.h:181)"
        assert not gg[:, 4 + k:].any(), "padding stays zero"
        _pure("packed/" + name, "grad_V", gg[:, 4:4 + k][has], gV_o[has])
    else:
        worst_V = 0.0
    _pure("packed/" + name, "grad_w", gg[:, 0], gw_o)
    print("%s: U=%d worst err/tol logits %.3f grad_w %.3f grad_V %.3f" % (name, U, worst, worst_w, worst_V))
    for o in (bt, d_rows, d_grads):
        o.close()


# ------------------------------------------------------------------ (ii) the fused step
def _oracle_model(oracle, rng, keys, k, kw, scale, init_mode):
    """an oracle store holding a non-trivial model over `keys`: random FTRL state, V (+ AdaGrad
    accumulators) for ~60 % of the keys, w == 0 for ~20 % (some of which are due for lazy InitV)"""
    so = oracle.store_create(init_mode=init_mode, V_dim=k, **kw)
    U = len(keys)
    w = (rng.normal(size=U) * scale).astype(np.float32)
    w[rng.random(U) < 0.2] = 0.0
    sqrt_g = np.abs(rng.normal(size=U)).astype(np.float32)
    z = (rng.normal(size=U) * 0.5).astype(np.float32)
    cnt = rng.integers(0, 2 * max(kw.get("V_threshold", 0), 1) + 2, size=U).astype(np.float32)
    has = (rng.random(U) < 0.6) & (w != 0) if k else np.zeros(U, bool)
    V = np.zeros((U, 2 * max(k, 1)), np.float32)
    if k:
        V[:, :k] = rng.normal(size=(U, k)) * scale
        V[:, k:] = np.abs(rng.normal(size=(U, k))) * 0.3
        V[~has] = 0
    for i in range(U):
        so.poke(int(keys[i]), cnt[i], w[i], sqrt_g[i], z[i], V[i, :2 * k] if has[i] else None)
    scal = np.stack([cnt, w, sqrt_g, z], 1)
    return so, scal, has.astype(np.int32), V[:, :2 * k] if k else None


FUSED_CASES = ["c2_rcv1_k8", "ragged_k0", "ragged_k5", "ragged_k64", "hot_k64", "adversarial_k16", "tall_ragged_k4", "c3_full_k64",
               "c3_full_k128"]


@pytest.mark.parametrize("name", FUSED_CASES)
@pytest.mark.parametrize("l1", [0.0, 0.05])
def test_fused_first_step_vs_oracle(capi, ctx, oracle, rcv1, name, l1):
    """k_lookup + k_forward<L,5> (table rows) + k_backward_all<L,true,true,*> (in-place FTRL / AdaGrad /
    lazy InitV): the first step on an imported model"""
    from oracle import bindings as ob, tolerance as T
    if name.startswith("c3_full") and l1 != 0.0:
        pytest.skip("full-size case runs once")
    b, k, scale = _case(name, rcv1)
    rng = np.random.default_rng(23)
    kw = dict(l1=l1, l2=0.01, lr=0.05, lr_beta=1.0, V_lr=0.02, V_lr_beta=1.0, V_l2=0.02, V_threshold=3, V_init_scale=0.1, seed=5)
    loc = oracle.localize(b["offset"], b["index"])
    keys, U = loc["feaids"], loc["U"]
    so, scal, has_v, V = _oracle_model(oracle, rng, keys, k, kw, scale, ob.INIT_HASH)
    tb = capi.Table(ctx, max(2 * U, 1 << 12), V_dim=k, init_mode=capi.INIT_HASH, **kw)
    tb.import_(keys, scal, has_v, V)
    nnz = int(b["offset"][-1])
    bt = capi.Batch(ctx, len(b["label"]), nnz)
    bt.load_host(b["offset"], b["index"], b["value"], b["label"])
    bt.localize()
    # the weights both sides see
    vals, lens = so.pull(keys)
    vg, lg = tb.pull(keys)
    assert np.array_equal(vals, vg) and np.array_equal(lens, lg), "imported model differs from the oracle's"
    bt.sgd_step(tb, is_train=True, push_cnt=False)
    pg = bt.pred()
    wp, vp = (None, None) if k == 0 else oracle.get_pos(lens)
    po = oracle.fm_predict(k, loc["offset"], loc["index"], b["value"], vals, wp, vp)
    X = T.design(loc["offset"], loc["index"], b["value"], U)
    w64, V64, has = T.dense_rows(vals, lens, k)
    _, floor_p = T.predict_bound(X, w64, V64)
    case = "fused/%s/l1=%g" % (name, l1)
    worst = _worst(case, "logits", T.check(pg, po, floor_p, "%s fused logits vs oracle" % name))
    _pure(case, "logits", pg, po)
    # the reference's update applied to the reference's gradient of the same logits
    go = oracle.fm_calcgrad(k, loc["offset"], loc["index"], b["value"], b["label"], vals, pg, wp, vp)
    so.push(keys, ob.GRADIENT, go, lens)
    _, _, floor_w, floor_V = T.calcgrad_bound(X, b["label"], pg, w64, V64, has)
    gw_o, gV_o, _ = T.dense_rows(go, lens, k)
    tol_gw = 1e-5 * np.abs(gw_o) + floor_w
    # device model after the step
    ex = tb.export()
    order = np.argsort(ex["keys"])
    assert np.array_equal(ex["keys"][order], keys)
    sc, hv = ex["scal"][order], ex["has_V"][order]
    Vd = ex["V"][order] if k else None
    ref = [so.peek(int(key)) for key in keys]
    r_scal = np.array([[r["fea_cnt"], r["w"], r["sqrt_g"], r["z"]] for r in ref], np.float64)
    r_has = np.array([r["V"] is not None for r in ref])
    assert np.array_equal(hv != 0, r_has), "lazy InitV decisions differ"
    assert np.array_equal(sc[:, 0], r_scal[:, 0]), "fea_cnt must be untouched by a gradient push"
    # d(state)/d(gradient): sqrt_g 1; z 1 + |w|/lr; w' (lr (1 + |w|/lr) + |w'|) / lr_beta  (sgd_updater.cc:104-120)
    w_old = np.abs(scal[:, 1].astype(np.float64))
    amp_z = 1.0 + w_old / kw["lr"]
    _worst(case, "sqrt_g", T.check(sc[:, 2], r_scal[:, 2], 2 * tol_gw, "%s sqrt_g after the step" % name))
    _worst(case, "z", T.check(sc[:, 3], r_scal[:, 3], 2 * amp_z * tol_gw, "%s z after the step" % name))
    tol_w = 2 * (kw["lr"] * amp_z + np.abs(r_scal[:, 1])) * tol_gw
    near_l1 = np.abs(np.abs(r_scal[:, 3]) - l1) <= 2 * amp_z * tol_gw  # the |z| <= l1 test may fall either way
    assert near_l1.mean() < 0.01
    _worst(case, "w", T.check(sc[~near_l1, 1], r_scal[~near_l1, 1], tol_w[~near_l1], "%s w after the step" % name))
    if k:
        r_V = np.array([r["V"] if r["V"] is not None else np.zeros(2 * k, np.float32) for r in ref], np.float64)
        fresh = r_has & ~has  # initialised by this step's update: hash init, bit for bit
        assert np.array_equal(Vd[fresh], r_V[fresh].astype(np.float32))
        upd = has
        tol_gV = 1e-5 * np.abs(gV_o) + floor_V
        amp_V = 1.0 + kw["V_l2"]
        _worst(case, "acc", T.check(Vd[upd][:, k:], r_V[upd][:, k:], 2 * amp_V * tol_gV[upd],
                                    "%s AdaGrad accumulators after the step" % name))
        _worst(case, "V", T.check(Vd[upd][:, :k], r_V[upd][:, :k], 2 * amp_V * kw["V_lr"] * 3 * tol_gV[upd] + 1e-7 * np.abs(r_V[upd][:, :k]),
                                  "%s V after the step" % name))
        assert not Vd[~r_has].any()
    print("%s l1=%g: U=%d worst logits err/tol %.3f, %d keys got V" % (name, l1, U, worst, int((r_has & ~has).sum())))
    tb.close()
    bt.close()


def test_fused_step_with_count_push_vs_oracle(capi, ctx, oracle):
    """the epoch-0 form of the step (k_lookup applies Push(kFeaCount) first, sgd_learner.cc:214-217): rows
    whose count crosses V_threshold get their V before the pull, as in the reference"""
    from oracle import bindings as ob, tolerance as T
    rng = np.random.default_rng(31)
    b = random_batch(rng, 400, 2000, 30, binary=True)
    k = 8
    kw = dict(l1=0.0, l2=0.0, lr=0.05, V_lr=0.02, V_l2=0.01, V_threshold=3, V_init_scale=0.1, seed=5)
    loc = oracle.localize(b["offset"], b["index"])
    keys, U = loc["feaids"], loc["U"]
    so, scal, has_v, V = _oracle_model(oracle, rng, keys, k, kw, 0.1, ob.INIT_HASH)
    tb = capi.Table(ctx, 1 << 13, V_dim=k, init_mode=capi.INIT_HASH, **kw)
    tb.import_(keys, scal, has_v, V)
    bt = capi.Batch(ctx, 400, int(b["offset"][-1]))
    bt.load_host(b["offset"], b["index"], b["value"], b["label"])
    bt.localize()
    bt.sgd_step(tb, is_train=False, push_cnt=True)
    so.push(keys, ob.FEA_COUNT, loc["feacnt"])
    vals, lens = so.pull(keys)
    vg, lg = tb.pull(keys)
    assert np.array_equal(lens, lg) and np.array_equal(vals, vg)
    assert (lens > 1).sum() > has_v.sum(), "the count push should have initialised some V rows"
    wp, vp = oracle.get_pos(lens)
    po = oracle.fm_predict(k, loc["offset"], loc["index"], None, vals, wp, vp)
    X = T.design(loc["offset"], loc["index"], None, U)
    w64, V64, _ = T.dense_rows(vals, lens, k)
    _, floor_p = T.predict_bound(X, w64, V64)
    T.check(bt.pred(), po, floor_p, "logits after the count push")
    tb.close()
    bt.close()
