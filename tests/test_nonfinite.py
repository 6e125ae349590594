"""SURVEY.md 8(a) trap (3): SpMV skips exact-zero operands (/root/reference/src/common/spmv.h:125, :155), SpMM skips keys
without V (spmm.h:108, V_pos = -1) — numerically neutral, except that a skipped operand never meets an Inf / NaN
feature value.  What must hold on every path (reference, C restatement, HIP):

  * a key whose pulled weight is exactly 0 and that has no V contributes NOTHING to the logit, whatever its value:
    the logit stays finite where every other term is finite;
  * a nonzero weight (or an allocated V row) times Inf / NaN makes the logit non-finite (and, for V_dim > 0, a logit
    of +-Inf is clamped to +-20 while NaN stays NaN, fm_loss.h:118);
  * a slope that is exactly 0 (V_dim = 0, |logit| > ~89: exp overflows, -y / inf = -+0) is skipped by TransTimes:
    the gradient of a key does not see that example's Inf / NaN value.

CPU: the C restatement against the reference itself (oracle/_ref) on such inputs.  GPU: k_forward / k_backward_all
(exchange layout) and the fused step (k_lookup + k_forward + k_update_fused on an imported model) against the
restatement: the same finite / non-finite pattern, and rtol 1e-5 (+ floor) where finite.  The device does not
distinguish +-Inf from NaN in a sum that is non-finite on both sides (padding lanes multiply by 0), and it has one
documented deviation: four consecutive V coordinates (one lane's 16 B slice) that are all exactly 0.0 inside an
ALLOCATED row are treated like a key without V (dfh_kernels.hip k_forward) — NaN in the reference,
neutral here.  Observable only when the WHOLE allocated row is zero (any nonzero slice meets the same value and makes
Inf - Inf = NaN on both sides): constructed in test_allocated_all_zero_V_row_times_nonfinite_value, which asserts the
documented outcome on both sides.
"""
import numpy as np
import pytest


def _batch(rng, k, nrows=160, nkeys=300, s=12):
    """rows of `s` real-valued features; special keys (ranks after localisation = raw id order, ids are < 2^20 so the
    nibble reversal keeps no simple order: the roles are assigned AFTER localisation, by rank)"""
    off = (np.arange(nrows + 1) * s).astype(np.uint64)
    idx = rng.integers(1, nkeys, size=nrows * s).astype(np.uint64)
    val = rng.normal(size=nrows * s).astype(np.float32)
    lab = np.where(rng.random(nrows) < 0.4, 1.0, -1.0).astype(np.float32)
    return dict(offset=off, index=idx, value=val, label=lab)


def _scenario(oracle, k, seed):
    rng = np.random.default_rng(seed)
    b = _batch(rng, k)
    loc = oracle.localize(b["offset"], b["index"])
    U = loc["U"]
    ranks = rng.permutation(U)
    zero_w_nov = ranks[:12]        # w == 0, no V: skipped entirely
    live_nov = ranks[12:20]        # w != 0, no V
    zero_w_v = ranks[20:28]        # w == 0 but V allocated (V_dim > 0 only)
    w = (rng.normal(size=U) * 0.1).astype(np.float32)
    w[w == 0] = 0.01
    w[zero_w_nov] = 0.0
    w[zero_w_v] = 0.0
    has = np.ones(U, bool) if k else np.zeros(U, bool)
    has[zero_w_nov] = False
    has[live_nov] = False
    lens = np.where(has, 1 + k, 1).astype(np.int32) if k else np.zeros(0, np.int32)
    vals = []
    for u in range(U):
        vals.append(w[u:u + 1])
        if k and has[u]:
            vals.append((rng.normal(size=k) * 0.1).astype(np.float32))
    W = np.concatenate(vals).astype(np.float32)
    # poison the values: per nonzero, by the role of its key
    val = b["value"].copy()
    ix = loc["index"]
    nnz = len(ix)
    kind = np.zeros(nnz, np.int8)
    poison = np.array([np.inf, -np.inf, np.nan], np.float32)
    for j in range(nnz):
        u = ix[j]
        r = rng.random()
        if u in zero_w_nov and r < 0.6:
            val[j] = poison[rng.integers(0, 3)]
            kind[j] = 1            # harmless: must be skipped
        elif (u in live_nov or u in zero_w_v) and r < 0.15:
            val[j] = poison[rng.integers(0, 3)]
            kind[j] = 2            # poisons the example
        elif r < 0.002:
            val[j] = poison[rng.integers(0, 3)]
            kind[j] = 2
    b["value"] = val
    rows = np.repeat(np.arange(len(b["label"])), np.diff(b["offset"]).astype(np.int64))
    bad_rows = np.zeros(len(b["label"]), bool)
    bad_rows[rows[kind == 2]] = True
    assert (kind == 1).sum() > 20 and bad_rows.sum() > 3 and (~bad_rows).sum() > 20
    return b, loc, W, lens, bad_rows, kind


def _mask_check(got, ref, what, rtol=1e-5, atol=1e-5):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    fg, fr = np.isfinite(got), np.isfinite(ref)
    assert np.array_equal(fg, fr), "%s: finite pattern differs at %r (got %r, want %r)" % (
        what, np.nonzero(fg != fr)[0][:8], got[fg != fr][:8], ref[fg != fr][:8])
    np.testing.assert_allclose(got[fr], ref[fr], rtol=rtol, atol=atol, err_msg=what)


@pytest.mark.parametrize("k", [0, 4, 64])
def test_restatement_matches_reference_on_nonfinite_values(oracle, ref, k):
    b, loc, W, lens, bad_rows, kind = _scenario(oracle, k, 100 + k)
    wp, vp = (None, None) if k == 0 else oracle.get_pos(lens)
    po = oracle.fm_predict(k, loc["offset"], loc["index"], b["value"], W, wp, vp)
    pr = ref.fm_predict(k, loc["offset"], loc["index"], b["value"], W, wp, vp)
    # the skip: every example whose only poison sits on zero-weight keys without V has a finite logit
    assert np.all(np.isfinite(pr[~bad_rows])) and not np.all(np.isfinite(pr[bad_rows]))
    assert np.array_equal(np.isnan(po), np.isnan(pr))
    fin = np.isfinite(pr)
    assert np.array_equal(po[fin], pr[fin])
    pr2, gr = ref.fm_predict_calcgrad(k, loc["offset"], loc["index"], b["value"], b["label"], W, wp, vp)
    go = oracle.fm_calcgrad(k, loc["offset"], loc["index"], b["value"], b["label"], W, pr2, wp, vp)
    assert np.array_equal(np.isfinite(go), np.isfinite(gr))
    f = np.isfinite(gr)
    assert np.array_equal(go[f], gr[f])


def _zero_slope_scenario(oracle):
    """V_dim = 0 (no clamp): a logit of +1000 on a positive example makes its slope exactly -0 (exp overflows); the row also
    carries an Inf value on a key of zero weight"""
    rng = np.random.default_rng(5)
    b = _batch(rng, 0, nrows=80, nkeys=120, s=6)
    loc = oracle.localize(b["offset"], b["index"])
    U = loc["U"]
    W = (rng.normal(size=U) * 0.1).astype(np.float32)
    big = loc["index"][0]          # first nonzero of row 0: give its key a huge weight and a finite value
    W[big] = 1000.0
    val = b["value"].copy()
    off = b["offset"].astype(np.int64)
    val[off[0]] = 1.0
    b["label"][0] = 1.0            # y pred = +1000 -> exp = inf -> p = -0
    # poison another nonzero of row 0 whose key has zero weight
    j2 = off[0] + 1 if loc["index"][off[0] + 1] != big else off[0] + 2
    u2 = loc["index"][j2]
    val[j2] = np.inf
    W[u2] = 0.0
    for j in range(off[0], off[1]):   # no other nonzero of row 0 may share the two special keys
        if j not in (off[0], j2) and loc["index"][j] in (big, u2):
            val[j] = 0.0
    b["value"] = val
    return b, loc, W, u2


def test_zero_slope_is_skipped_like_the_reference(oracle, ref):
    """TransTimes skips a slope that is exactly 0 (spmv.h:155), so an Inf value in such a row does not reach the gradient"""
    b, loc, W, u2 = _zero_slope_scenario(oracle)
    pr, gr = ref.fm_predict_calcgrad(0, loc["offset"], loc["index"], b["value"], b["label"], W)
    assert np.isfinite(pr[0]) and pr[0] > 500
    go = oracle.fm_calcgrad(0, loc["offset"], loc["index"], b["value"], b["label"], W, pr)
    assert np.array_equal(np.isfinite(go), np.isfinite(gr)) and np.isfinite(gr[u2])
    f = np.isfinite(gr)
    assert np.array_equal(go[f], gr[f])


# --------------------------------------------------------------------------- GPU
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


@pytest.mark.gpu
@pytest.mark.parametrize("k", [0, 4, 64])
def test_hip_kernels_on_nonfinite_values(capi, ctx, oracle, k):
    """exchange layout: k_forward<L,5> + k_backward_all<L,false,...>"""
    from oracle import tolerance as T
    b, loc, W, lens, bad_rows, kind = _scenario(oracle, k, 100 + k)
    wp, vp = (None, None) if k == 0 else oracle.get_pos(lens)
    po = oracle.fm_predict(k, loc["offset"], loc["index"], b["value"], W, wp, vp)
    stride = capi.row_stride(k)
    rows = T.packed_rows(W, lens, k, stride)
    U = loc["U"]
    bt = capi.Batch(ctx, len(b["label"]), int(b["offset"][-1]))
    bt.load_host(b["offset"], b["index"], b["value"], b["label"])
    bt.localize()
    d_rows = capi.DeviceBuffer.from_numpy(ctx, rows)
    d_grads = capi.DeviceBuffer(ctx, max(rows.nbytes, 16))
    bt.forward(k, d_rows.ptr)
    pg = bt.pred()
    assert np.all(np.isfinite(pg[~bad_rows])), "a zero weight without V met a non-finite value"
    _mask_check(pg, po, "logits (V_dim %d)" % k)
    go = oracle.fm_calcgrad(k, loc["offset"], loc["index"], b["value"], b["label"], W, pg, wp, vp)
    bt.backward(k, d_rows.ptr, d_grads.ptr)
    gg = d_grads.to_numpy(np.float32, rows.size).reshape(U, stride)
    gw_o, gV_o, has = T.dense_rows(go, lens, k)
    _mask_check(gg[:, 0], gw_o, "grad_w (V_dim %d)" % k, rtol=1e-4)
    if k:
        _mask_check(gg[:, 4:4 + k][has], gV_o[has], "grad_V (V_dim %d)" % k, rtol=1e-4)
    for o in (bt, d_rows, d_grads):
        o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [0, 4, 64])
def test_hip_fused_step_on_nonfinite_values(capi, ctx, oracle, k):
    """the resident-table path: k_lookup + k_forward (uw words, speculative V rows) + k_update_fused; the model after
    the step must be non-finite exactly where the reference's is"""
    from oracle import bindings as ob, tolerance as T
    b, loc, W, lens, bad_rows, kind = _scenario(oracle, k, 100 + k)
    keys, U = loc["feaids"], loc["U"]
    kw = dict(l1=0.0, l2=0.01, lr=0.05, V_lr=0.02, V_l2=0.02, V_threshold=1000, V_init_scale=0.1, seed=5)
    so = oracle.store_create(init_mode=ob.INIT_HASH, V_dim=k, **kw)
    w64, V64, has = T.dense_rows(W, lens, k)
    rng = np.random.default_rng(9)
    scal = np.stack([np.ones(U), w64, np.abs(rng.normal(size=U)), rng.normal(size=U) * 0.1], 1).astype(np.float32)
    V = np.zeros((U, 2 * max(k, 1)), np.float32)
    if k:
        V[:, :k] = V64
        V[:, k:] = np.abs(rng.normal(size=(U, k))) * 0.3
        V[~has] = 0
    for i in range(U):
        so.poke(int(keys[i]), *scal[i], V[i, :2 * k] if (k and has[i]) else None)
    tb = capi.Table(ctx, 1 << 12, V_dim=k, init_mode=capi.INIT_HASH, **kw)
    tb.import_(keys, scal, has.astype(np.int32), V[:, :2 * k] if k else None)
    bt = capi.Batch(ctx, len(b["label"]), int(b["offset"][-1]))
    bt.load_host(b["offset"], b["index"], b["value"], b["label"])
    bt.localize()
    vals, ln = so.pull(keys)
    assert np.array_equal(vals, W)
    bt.sgd_step(tb, is_train=True, push_cnt=False)
    pg = bt.pred()
    wp, vp = (None, None) if k == 0 else oracle.get_pos(ln)
    po = oracle.fm_predict(k, loc["offset"], loc["index"], b["value"], vals, wp, vp)
    assert np.all(np.isfinite(pg[~bad_rows]))
    _mask_check(pg, po, "fused logits (V_dim %d)" % k)
    go = oracle.fm_calcgrad(k, loc["offset"], loc["index"], b["value"], b["label"], vals, pg, wp, vp)
    so.push(keys, ob.GRADIENT, go, ln)
    vo, lo = so.pull(keys)
    vg, lg = tb.pull(keys)
    assert np.array_equal(lg, lo)
    _mask_check(vg, vo, "model after the step (V_dim %d)" % k, rtol=2e-4, atol=1e-6)
    assert np.isfinite(vo).sum() > 0.3 * len(vo)
    tb.close()
    bt.close()


def _zero_row_scenario(k):
    """3 examples x 3 keys; the key of rank 1 has an ALLOCATED V row that is exactly zero and w = 0; example 1 carries it with x = Inf,
    example 2 with x = NaN, example 0 with a finite value"""
    off = np.array([0, 3, 6, 9], np.uint64)
    idx = np.array([0, 1, 2] * 3, np.uint32)   # localized: ranks of the three keys
    val = np.array([0.5, 1.0, -1.0, 0.25, np.inf, 2.0, 1.5, np.nan, 0.5], np.float32)
    lab = np.array([1, -1, 1], np.float32)
    rng = np.random.default_rng(5)
    rows = [np.concatenate([[0.3], rng.normal(size=k) * 0.1]), np.concatenate([[0.0], np.zeros(k)]),
            np.concatenate([[-0.2], rng.normal(size=k) * 0.1])]
    W = np.concatenate(rows).astype(np.float32)
    lens = np.full(3, 1 + k, np.int32)
    return off, idx, val, lab, W, lens


@pytest.mark.parametrize("k", [4, 64])
def test_allocated_all_zero_V_row_times_nonfinite_value_reference_side(oracle, ref, k):
    """the reference's side of what was, through round 5, the one documented deviation: SpMM has no zero skip (spmm.h:108-118 tests V_pos only), so an
    allocated row of zeros times Inf / NaN is NaN, and so is the logit (fm_loss.h:118 clamps +-Inf, never NaN)"""
    off, idx, val, lab, W, lens = _zero_row_scenario(k)
    wp, vp = oracle.get_pos(lens)
    po = oracle.fm_predict(k, off, idx, val, W, wp, vp)
    pr = ref.fm_predict(k, off, idx, val, W, wp, vp)
    assert np.isfinite(po[0]) and np.isnan(po[1]) and np.isnan(po[2])
    assert np.array_equal(np.isnan(pr), np.isnan(po)) and pr[0] == po[0]


@pytest.mark.gpu
@pytest.mark.parametrize("k", [4, 64])
def test_allocated_all_zero_V_row_times_nonfinite_value(capi, ctx, oracle, k):
    """... and the device's — since round 6 the reference's.  Through round 5 k_forward took a 16 B slice of a speculatively
    loaded V row that was all zero for "no V" and this state (reachable by import / load only) came out neutral where the
    reference gives NaN: the one documented deviation.  Now a key's has-V flag travels with its row word (kHasV, left by
    k_lookup / k_uw_remote; the header flag of packed rows) and decides, as V_pos = -1 does in SpMM::Times (spmm.h:108-118):
    both forward paths — packed rows (dfh_batch_forward) and the fused step's row words (dfh_sgd_step) — give the
    reference's NaN pattern and its finite logits."""
    from oracle import tolerance as T
    off, idx, val, lab, W, lens = _zero_row_scenario(k)
    wp, vp = oracle.get_pos(lens)
    po = oracle.fm_predict(k, off, idx, val, W, wp, vp)
    assert np.isfinite(po[0]) and np.isnan(po[1]) and np.isnan(po[2])
    stride = capi.row_stride(k)
    rows = T.packed_rows(W, lens, k, stride)
    bt = capi.Batch(ctx, 3, 9)
    raw = np.array([5, 9, 12], np.uint64)   # any three distinct ids; ranks follow the reversed order
    keys = np.array([capi.reverse_bytes(int(x)) for x in raw], dtype=np.uint64)
    order = np.argsort(keys)
    bt.load_host(off, raw[order][idx], val, lab)   # raw[order][r]: the raw id whose reversed key has rank r
    bt.localize()
    d_rows = capi.DeviceBuffer.from_numpy(ctx, rows)
    bt.forward(k, d_rows.ptr)
    pg = bt.pred()
    assert np.array_equal(np.isnan(pg), np.isnan(po)), (pg, po)
    np.testing.assert_allclose(pg[0], po[0], rtol=1e-5, atol=1e-6)
    # the fused step: the same model imported (every key WITH V, the middle one's row exactly zero), a prediction step
    tb = capi.Table(ctx, 1 << 10, V_dim=k, init_mode=capi.INIT_HASH, l1=0.0, l2=0.0, V_threshold=0, seed=1)
    Wm = W.reshape(3, 1 + k)
    scal = np.stack([np.full(3, 5.0), Wm[:, 0], np.full(3, 0.5), np.zeros(3)], 1).astype(np.float32)   # fea_cnt, w, sqrt_g, z
    V = np.concatenate([Wm[:, 1:], np.full((3, k), 0.1)], 1).astype(np.float32)                        # V | accumulators
    tb.import_(keys[order], scal, np.ones(3, np.int32), V)
    bt.load_host(off, raw[order][idx], val, lab)
    bt.localize()
    bt.sgd_step(tb, is_train=False, push_cnt=False)
    pf = bt.pred()
    assert np.array_equal(np.isnan(pf), np.isnan(po)), (pf, po)
    np.testing.assert_allclose(pf[0], po[0], rtol=1e-5, atol=1e-6)
    # ... and a key WITHOUT V stays skipped whatever its value: has_V = 0 for the middle key -> all three logits finite
    tb2 = capi.Table(ctx, 1 << 10, V_dim=k, init_mode=capi.INIT_HASH, l1=0.0, l2=0.0, V_threshold=1000, seed=1)
    has = np.array([1, 0, 1], np.int32)
    V2 = V.copy()
    V2[1] = 0
    tb2.import_(keys[order], scal, has, V2)
    bt.load_host(off, raw[order][idx], val, lab)
    bt.localize()
    bt.sgd_step(tb2, is_train=False, push_cnt=False)
    pn = bt.pred()
    lens2 = np.array([1 + k, 1, 1 + k], np.int32)
    W2 = np.concatenate([Wm[0], Wm[1, :1], Wm[2]]).astype(np.float32)
    wp2, vp2 = oracle.get_pos(lens2)
    po2 = oracle.fm_predict(k, off, idx, val, W2, wp2, vp2)
    assert np.all(np.isfinite(po2)) and np.all(np.isfinite(pn)), (pn, po2)
    np.testing.assert_allclose(pn, po2, rtol=1e-5, atol=1e-6)
    for o in (bt, d_rows, tb, tb2):
        o.close()


@pytest.mark.gpu
def test_hip_zero_slope_is_skipped(capi, ctx, oracle):
    """k_backward_all (gradient rows) and the fused update, V_dim = 0: the example whose slope underflowed to 0 does not
    hand its Inf value to the gradient of a zero-weight key"""
    from oracle import bindings as ob, tolerance as T
    b, loc, W, u2 = _zero_slope_scenario(oracle)
    U = loc["U"]
    stride = capi.row_stride(0)
    rows = T.packed_rows(W, np.zeros(0, np.int32), 0, stride)
    bt = capi.Batch(ctx, len(b["label"]), int(b["offset"][-1]))
    bt.load_host(b["offset"], b["index"], b["value"], b["label"])
    bt.localize()
    d_rows = capi.DeviceBuffer.from_numpy(ctx, rows)
    d_grads = capi.DeviceBuffer(ctx, max(rows.nbytes, 16))
    bt.forward(0, d_rows.ptr)
    pg = bt.pred()
    assert np.isfinite(pg[0]) and pg[0] > 500
    go = oracle.fm_calcgrad(0, loc["offset"], loc["index"], b["value"], b["label"], W, pg)
    bt.backward(0, d_rows.ptr, d_grads.ptr)
    gg = d_grads.to_numpy(np.float32, rows.size).reshape(U, stride)
    assert np.isfinite(gg[u2, 0])
    _mask_check(gg[:, 0], go, "grad_w with a zero slope", rtol=1e-4)
    # the fused step on the same model
    kw = dict(l1=0.0, l2=0.0, lr=0.05, V_threshold=1000, seed=1)
    so = oracle.store_create(init_mode=ob.INIT_HASH, V_dim=0, **kw)
    tb = capi.Table(ctx, 1 << 10, V_dim=0, init_mode=capi.INIT_HASH, **kw)
    scal = np.stack([np.ones(U), W, np.full(U, 0.5), np.zeros(U)], 1).astype(np.float32)
    for i in range(U):
        so.poke(int(loc["feaids"][i]), *scal[i], None)
    tb.import_(loc["feaids"], scal, np.zeros(U, np.int32), None)
    bt.load_host(b["offset"], b["index"], b["value"], b["label"])
    bt.localize()
    bt.sgd_step(tb, is_train=True, push_cnt=False)
    g2 = oracle.fm_calcgrad(0, loc["offset"], loc["index"], b["value"], b["label"], W, bt.pred())
    so.push(loc["feaids"], ob.GRADIENT, g2)
    vg, _ = tb.pull(loc["feaids"])
    vo, _ = so.pull(loc["feaids"])
    assert np.isfinite(vo[u2])
    _mask_check(vg, vo, "model after the step with a zero slope", rtol=2e-4, atol=1e-6)
    for o in (bt, d_rows, d_grads, tb):
        o.close()
