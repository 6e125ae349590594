"""CPU check of oracle/tolerance.py, the tolerance the GPU kernel-parity tests use
(tests/test_kernel_parity.py): |got - ref| <= 1e-5 |ref| + C_SIGMA * 2^-24 * sqrt(n) * sum|terms|  (C_SIGMA = 1 since round 4; 4 before).

  * it admits what it must: the reference's own fp32 results against exact float64 arithmetic,
    and the reference's results when the same terms are summed in a different order
    (nonzeros of every row reversed, rows reversed) — the situation of the device kernels;
  * it is not vacuous: a perturbation of 1e-4 relative in one weight, or one dropped term, fails.
"""
import numpy as np
import pytest

from conftest import random_batch


def _setup(oracle, k, seed, binary):
    from oracle import tolerance as T
    rng = np.random.default_rng(seed)
    b = random_batch(rng, 300, 2000, 45, binary=binary)
    loc = oracle.localize(b["offset"], b["index"])
    U = loc["U"]
    lens = np.where(rng.random(U) < 0.3, 1, 1 + k).astype(np.int32) if k else np.zeros(0, np.int32)
    W = (rng.normal(size=int(lens.sum()) if k else U) * 0.1).astype(np.float32)
    wp, vp = oracle.get_pos(lens) if k else (None, None)
    D = T.design(loc["offset"], loc["index"], b["value"], U)
    w64, V64, has = T.dense_rows(W, lens, k)
    return T, b, loc, lens, W, wp, vp, D, w64, V64, has


def _reversed_rows(loc, value):
    """the same minibatch with the nonzeros of every row in reverse order"""
    off = np.asarray(loc["offset"], np.int64)
    idx = loc["index"].copy()
    val = None if value is None else value.copy()
    for i in range(len(off) - 1):
        idx[off[i]:off[i + 1]] = idx[off[i]:off[i + 1]][::-1]
        if val is not None:
            val[off[i]:off[i + 1]] = val[off[i]:off[i + 1]][::-1]
    return idx, val


@pytest.mark.parametrize("k", [0, 5, 64])
@pytest.mark.parametrize("binary", [False, True])
def test_floor_admits_reference_rounding_and_reordering(oracle, k, binary):
    T, b, loc, lens, W, wp, vp, D, w64, V64, has = _setup(oracle, k, 100 + k, binary)
    po = oracle.fm_predict(k, loc["offset"], loc["index"], b["value"], W, wp, vp)
    p64, floor_p = T.predict_bound(D, w64, V64)
    assert T.check(po, p64, floor_p, "oracle logits vs float64") < 1.0
    idx_r, val_r = _reversed_rows(loc, b["value"])
    pr = oracle.fm_predict(k, loc["offset"], idx_r, val_r, W, wp, vp)
    assert T.check(pr, po, floor_p, "logits, nonzeros reversed") < 1.0
    go = oracle.fm_calcgrad(k, loc["offset"], loc["index"], b["value"], b["label"], W, po, wp, vp)
    gw64, gV64, floor_w, floor_V = T.calcgrad_bound(D, b["label"], po, w64, V64, has)
    gw_o, gV_o, _ = T.dense_rows(go, lens, k)
    assert T.check(gw_o, gw64, floor_w, "oracle grad_w vs float64") < 1.0
    if k:
        assert T.check(gV_o, gV64, floor_V, "oracle grad_V vs float64") < 1.0
    # rows in reverse order: every per-key sum runs backwards (spmv.h:152-168 sums in row order)
    n = len(b["label"])
    off = np.asarray(loc["offset"], np.int64)
    rows = [np.arange(off[i], off[i + 1]) for i in range(n)][::-1]
    perm = np.concatenate(rows) if rows else np.zeros(0, np.int64)
    off_r = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.uint64)
    val_p = None if b["value"] is None else b["value"][perm]
    gr = oracle.fm_calcgrad(k, off_r, loc["index"][perm], val_p, b["label"][::-1].copy(), W, po[::-1].copy(), wp, vp)
    gw_r, gV_r, _ = T.dense_rows(gr, lens, k)
    assert T.check(gw_r, gw_o, floor_w, "grad_w, rows reversed") < 1.0
    if k:
        assert T.check(gV_r, gV_o, floor_V, "grad_V, rows reversed") < 1.0


def test_floor_is_not_vacuous(oracle):
    k = 8
    T, b, loc, lens, W, wp, vp, D, w64, V64, has = _setup(oracle, k, 7, False)
    po = oracle.fm_predict(k, loc["offset"], loc["index"], b["value"], W, wp, vp)
    _, floor_p = T.predict_bound(D, w64, V64)
    go = oracle.fm_calcgrad(k, loc["offset"], loc["index"], b["value"], b["label"], W, po, wp, vp)
    _, _, floor_w, floor_V = T.calcgrad_bound(D, b["label"], po, w64, V64, has)
    gw_o, gV_o, _ = T.dense_rows(go, lens, k)
    # every weight off by 1e-4 relative: logits and gradients must be rejected
    W2 = (W.astype(np.float64) * (1 + 1e-4)).astype(np.float32)
    p2 = oracle.fm_predict(k, loc["offset"], loc["index"], b["value"], W2, wp, vp)
    with pytest.raises(AssertionError):
        T.check(p2, po, floor_p, "perturbed logits")
    g2 = oracle.fm_calcgrad(k, loc["offset"], loc["index"], b["value"], b["label"], W2, po, wp, vp)
    _, gV_2, _ = T.dense_rows(g2, lens, k)
    with pytest.raises(AssertionError):
        T.check(gV_2, gV_o, floor_V, "perturbed grad_V")
    # one nonzero dropped from the longest row
    off = np.asarray(loc["offset"], np.int64).copy()
    i = int(np.argmax(np.diff(off)))
    keep = np.ones(int(off[-1]), bool)
    keep[off[i]] = False
    off2 = off.copy()
    off2[i + 1:] -= 1
    p3 = oracle.fm_predict(k, off2.astype(np.uint64), loc["index"][keep], b["value"][keep], W, wp, vp)
    with pytest.raises(AssertionError):
        T.check(p3, po, floor_p, "logits with one term dropped")
    g3 = oracle.fm_calcgrad(k, off2.astype(np.uint64), loc["index"][keep], b["value"][keep], b["label"], W, po, wp, vp)
    gw_3, _, _ = T.dense_rows(g3, lens, k)
    with pytest.raises(AssertionError):
        T.check(gw_3, gw_o, floor_w, "grad_w with one term dropped")
